#!/usr/bin/env python3
"""Randomised exactness stress of the cell index behind GL_ASSOC_BRUTE: many maps (sizes, anisotropies up to the
cond <= 1e8 admission limit and beyond, clustered / spread means, huge and tiny components) x point clouds
(on-surface, uniform, far outside, exactly on means / cell boundaries); idx AND chi2 must be bit-identical to the
plain N x K sweep.    python tools/stress_index.py [n_cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gmmloc_amd
from gmmloc_amd import api


def haar(rng, K):
    q, r = np.linalg.qr(rng.standard_normal((K, 3, 3)))
    return q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ctx = gmmloc_amd.Context(0)
    ctx.set_option("assoc_index_min", 0)
    bad = 0
    for case in range(ncase):
        rng = np.random.default_rng(1000 + case)
        K = int(rng.choice([3, 17, 200, 1500, 4096, 9000, 20000]))
        ext = rng.choice([0.5, 5.0, 40.0])
        mean = rng.uniform(-ext, ext, (K, 3)) if rng.uniform() < 0.7 else rng.standard_normal((K, 3)) * ext * 0.2
        lo_e, hi_e = rng.choice([-8.0, -6.0, -4.0]), rng.choice([-2.0, 0.0, 1.0])
        lam = 10.0 ** rng.uniform(lo_e, hi_e, (K, 3))
        if rng.uniform() < 0.3:
            lam[rng.integers(0, K, max(1, K // 50))] *= 1e4  # a few huge components
        R = haar(rng, K)
        cov = np.einsum("kij,kj,klj->kil", R, lam, R)
        cov = 0.5 * (cov + cov.transpose(0, 2, 1))
        g = api.GMM(ctx, mean, cov.reshape(K, 9))
        N = int(rng.choice([1000, 20000, 100000]))
        comp = rng.integers(0, K, N)
        L = np.linalg.cholesky(cov[comp] + 1e-300 * np.eye(3))
        pts = mean[comp] + np.einsum("nij,nj->ni", L, rng.standard_normal((N, 3))) * rng.choice([0.5, 1.0, 3.0])
        info = g.index_info()
        extra = [rng.uniform(-ext * 1.5, ext * 1.5, (N // 4, 3)), mean[rng.integers(0, K, min(K, 500))],
                 rng.uniform(-ext, ext, (2000, 3)).round(2)]
        if info["enabled"]:  # points on cell boundaries
            h = info["cell"]
            extra.append(np.round(rng.uniform(-ext, ext, (2000, 3)) / h) * h)
        pts = np.ascontiguousarray(np.concatenate([pts] + extra))
        t = torch.from_numpy(pts).cuda()
        i1, d1 = g.associate3d(t, api.ASSOC_BRUTE)
        i2, d2 = g.associate3d(t, api.ASSOC_EXHAUSTIVE)
        ok = bool(torch.equal(i1, i2)) and bool(torch.equal(d1, d2))
        work = g.index_work(t) / len(pts) if info["enabled"] else float("nan")
        print("case %2d K=%5d N=%6d ext=%4.1f index=%s cell=%.3g always=%d cand/pt=%.1f resolved=%.2f %s" % (
            case, K, len(pts), ext, info["enabled"], info["cell"], info["always"], work,
            float((d1 <= 9.0).float().mean()), "OK" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        g.close()
    print("%d / %d cases identical" % (ncase - bad, ncase))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
