"""gl_track_frames on frames with one slot per FEATURE (1 200) of which a share holds a map point - the reference's frame - against the
share: what the refine costs per slot and per point (round 6: k_ba1_prep puts the empty slots last).   python tools/track_sparse.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api
M = 1200
mean, cov, cam, frames = bench.make_workload(2048)
prm = api.Params(); ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda k: torch.from_numpy(np.stack([f[k][:M] if f[k].ndim else f[k] for f in frames])).cuda()
pose0, Xw0, obs = T("pose_init"), T("Xw"), T("obs")
rng = np.random.default_rng(3)
for keep in (1.0, 0.6, 0.35, 0.15):
    oc = np.stack([np.where(rng.uniform(size=M) < keep, f["octave"][:M], -1) for f in frames]).astype(np.int32)
    octv = torch.from_numpy(oc).cuda()
    for B in (2048, 1):
        p0, x0, ob, o = pose0[:B].contiguous(), Xw0[:B].contiguous(), obs[:B].contiguous(), octv[:B].contiguous()
        ts = []
        for it in range(12 if B > 1 else 40):
            p, x = p0.clone(), x0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, ob, o, want_d2=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts[3:]))
        print("%4d slots, %3.0f %% with a map point, B = %4d: %.3f ms per call (%.1f k frames/s), pose checksum %.12f"
              % (M, 100 * keep, B, ms, B / ms, float(p.sum().item())), flush=True)
