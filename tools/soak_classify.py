"""Three-way comparison of the deviations tools/soak.py dumped: HIP (saved GPU outputs) vs the C++ oracle vs the
independent numpy restatement (oracle/numpy_ref.py), plus the oracle's OWN sensitivity to changes that leave the
mathematics untouched - 1-ulp perturbations of one input, or a re-ordering of the points (summation order).  A
deviation is called ILL-CONDITIONED only if the oracle itself moves by more than the tolerance under such a change;
otherwise it is a REAL mismatch of the HIP path.  CPU only (no GPU needed): the inputs are regenerated from the
(map, round) label by tools/soak_cases.py.
    python tools/soak_classify.py DUMP_DIR [--numpy]      (--numpy also runs the slow dense numpy LM on track / ba)"""
import argparse
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import numpy_ref as nr
from gmmloc_amd import api
from tests import oracle_lib
from tests.test_gpu_pose import pose_err
from tools import soak_cases as sc

orc = oracle_lib.load()
cam, gts = api.Camera(), sc.load_gt()
ncam = nr.Cam(cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.width, cam.height)
nprm = nr.Prm()
_maps = {}


def get_map(name):
    if name not in _maps:
        mean, cov = sc.load_map(name)
        _maps[name] = (mean, cov, orc.gmm_create(mean, cov), nr.build_components(mean, cov))
    return _maps[name]


def ulp_variants(x, n=6, seed=0):
    """n copies of x with ONE random coordinate moved by one ulp."""
    rng = np.random.default_rng(seed)
    out = []
    flat = np.asarray(x, float)
    for _ in range(n):
        y = flat.copy().ravel()
        j = int(rng.integers(0, y.size))
        y[j] = np.nextafter(y[j], np.inf if rng.integers(0, 2) else -np.inf)
        out.append(y.reshape(flat.shape))
    return out


def classify_chain_fallback(mapname, r, dump):
    mean, cov, h, comps = get_map(mapname)
    ch = sc.gen(mapname, r, mean, cov, gts, cam)["chain"]
    keep = ch["octave"] >= 0
    pts, obs, octv = ch["pts"][keep], ch["obs"][keep], ch["octave"][keep]
    c_ref, n_ref = orc.search_correspondence_for_pose(h, cam, ch["pose"], ch["obs"][:, :2].copy(), 5)
    c_ref, n_ref = c_ref[keep], n_ref[keep]
    o_ref, p_ref = orc.check_map_association(h, cam, ch["pose"], pts, obs, octv, c_ref, n_ref)
    pg = dump["pts_gpu"]
    rows = []
    for j in np.nonzero(np.abs(pg - p_ref).max(1) > 1e-9)[0]:
        # numpy restatement of the same feature (needs the neighbour rows of its candidates only)
        need = sorted(set(int(c) for c in c_ref[j] if c >= 0))
        nb_rows = dict(zip(need, [jj for jj, _ in nr.neighbour_rows(mean, cov, comps["det"], need)])) if need else {}
        nbs = [nb_rows.get(k, np.zeros(0, int)) for k in range(mean.shape[0])]
        c_np, p_np = nr.check_map_association(pts[j], obs[j], int(octv[j]), ch["pose"], c_ref[j], comps, mean, nbs, ncam, nprm)
        spread = 0.0
        for v in ulp_variants(pts[j]):
            _, pv = orc.check_map_association(h, cam, ch["pose"], v[None], obs[j][None], octv[j][None], c_ref[j][None], n_ref[j][None])
            spread = max(spread, np.abs(pv[0] - p_ref[j]).max())
        T = nr.SE3.from7(ch["pose"])
        rows.append(dict(feature=int(j), hip_vs_oracle=np.abs(pg[j] - p_ref[j]).max(), numpy_vs_oracle=np.abs(p_np - p_ref[j]).max(),
                         oracle_1ulp_spread=spread, moved=np.linalg.norm(p_ref[j] - pts[j]), depth=T.map(pts[j])[2],
                         obs=obs[j].tolist(), comp_oracle=int(o_ref[j]), comp_numpy=int(c_np)))
    return rows


def classify_tri(mapname, r, dump):
    mean, cov, h, comps = get_map(mapname)
    m = sc.gen(mapname, r, mean, cov, gts, cam)["tri"]
    x_ref, t_ref, c_ref = orc.create_map_points(h, cam, **m)
    xg, tg, cg = dump["x_gpu"], dump["type_gpu"], dump["comp_gpu"]
    with np.errstate(invalid="ignore"):
        bad = (tg != t_ref) | (cg != c_ref) | ((t_ref > 0) & ~(np.abs(xg - x_ref).max(1) <= 1e-8))
    rows = []
    for j in np.nonzero(bad)[0]:
        one = {k: m[k][j:j + 1] for k in m}
        pt_np, t_np, c_np = nr.create_map_point(m["pose1"][j], m["uvr1"][j], m["depth1"][j], int(m["oct1"][j]), m["pose2"][j], m["uvr2"][j],
                                                m["depth2"][j], int(m["oct2"][j]), m["cand1"][j][:m["n1"][j]], m["cand2"][j][:m["n2"][j]],
                                                comps, mean, ncam, nprm)
        spread, flips = 0.0, 0
        for v in ulp_variants(m["uvr1"][j]):
            o2 = dict(one)
            o2["uvr1"] = v[None]
            xv, tv, cv = orc.create_map_points(h, cam, **o2)
            flips += int(tv[0] != t_ref[j] or cv[0] != c_ref[j])
            with np.errstate(invalid="ignore"):
                spread = max(spread, np.nanmax(np.abs(xv[0] - x_ref[j])))
        rows.append(dict(match=int(j), type=(int(tg[j]), int(t_ref[j]), int(t_np)), comp=(int(cg[j]), int(c_ref[j]), int(c_np)),
                         dist_oracle=float(np.linalg.norm(x_ref[j])), hip_vs_oracle=float(np.nanmax(np.abs(xg[j] - x_ref[j]))),
                         numpy_vs_oracle=float(np.nanmax(np.abs(pt_np - x_ref[j]))) if pt_np is not None else float("nan"),
                         oracle_1ulp_spread=spread, oracle_1ulp_decision_flips=flips))
    return rows


def classify_tri_rejected(mapname, r, dump):
    """a match all sides REJECT (type 0): what optimizeTriangulationVec leaves behind (the candidate it settled on last) is no
    output of the reference; the oracle's own by-product under 1-ulp changes of one input"""
    mean, cov, h, comps = get_map(mapname)
    m = sc.gen(mapname, r, mean, cov, gts, cam)["tri"]
    x_ref, t_ref, c_ref = orc.create_map_points(h, cam, **m)
    tg, cg = dump["type_gpu"], dump["comp_gpu"]
    rows = []
    for j in np.nonzero((tg == 0) & (t_ref == 0) & (cg != c_ref))[0]:
        one = {k: m[k][j:j + 1] for k in m}
        seen, probes = {int(c_ref[j])}, 0
        for key in ("uvr1", "uvr2", "pose1", "pose2"):
            for v in ulp_variants(m[key][j], 12):
                _, tv, cv = orc.create_map_points(h, cam, **dict(one, **{key: v[None]}))
                seen.add(int(cv[0]) if tv[0] == 0 else -2)
                probes += 1
        rows.append(dict(match=int(j), comp_hip=int(cg[j]), comp_oracle=int(c_ref[j]), oracle_by_products_under_1ulp=sorted(seen), probes=probes,
                         hip_among_them=int(cg[j]) in seen))
    return rows


def classify_track(mapname, r, dump, with_numpy, prior=False):
    mean, cov, h, comps = get_map(mapname)
    f = sc.gen(mapname, r, mean, cov, gts, cam)["track"]
    keep, p_ref, pts_ref, a_ref, idx0, d20 = sc.track_oracle(orc, h, cam, f, prior=prior)
    dt, dr = pose_err(dump["pose_gpu"], p_ref)
    row = dict(M=len(keep), hip_vs_oracle=(dt, dr), oracle_vs_gt=pose_err(p_ref, f["pose_gt"]), init_vs_gt=pose_err(f["pose_init"], f["pose_gt"]))
    # the oracle's own sensitivity: (1) its points re-ordered (summation order), (2) every observation moved by a
    # relative 1e-16 .. 1e-15 (0 or 1 ulp).  A Levenberg schedule that stops before convergence can sit next to an
    # accept / reject flip: then a FEW of the probes jump by the full distance between the two branches, the others
    # do not move at all - so many probes, and the count of jumps, not just a maximum over a handful
    rng = np.random.default_rng(0)
    d_perm, d_ulp = [], []
    for _ in range(12):
        perm = rng.permutation(len(keep))
        _, p1, _, a1, _, _ = sc.track_oracle(orc, h, cam, f, perm, prior=prior)
        d_perm.append(max(pose_err(p1, p_ref)))
    for _ in range(36):
        g = dict(f)
        g["obs"] = f["obs"] * (1 + 3e-16 * rng.standard_normal(f["obs"].shape))
        g["obs"][f["obs"] < 0] = f["obs"][f["obs"] < 0]
        _, p1, _, a1, _, _ = sc.track_oracle(orc, h, cam, g, prior=prior)
        d_ulp.append(max(pose_err(p1, p_ref)))
    d_all = np.array(d_perm + d_ulp)
    row["oracle_probes"] = len(d_all)
    row["oracle_probe_median"] = float(np.median(d_all))
    row["oracle_probe_max"] = float(d_all.max())
    row["oracle_probes_above_1e-6"] = int((d_all > 1e-6).sum())
    if with_numpy and not prior:
        L = len(keep)
        assoc = np.where(d20 <= 9.0, idx0, -1).astype(np.int32)
        res = nr.joint_optimization(1, 0, f["pose_init"][None], np.zeros(1, np.uint8), f["Xw"][keep], assoc, np.arange(L + 1),
                                    np.zeros(L, int), f["obs"][keep], f["octave"][keep], comps, mean, ncam, nprm)
        row["numpy_vs_oracle"] = pose_err(res[0][0], p_ref)
        row["numpy_vs_hip"] = pose_err(res[0][0], dump["pose_gpu"])
    return [row]


def classify_ba(r, dump, with_numpy):
    mean, cov, h, comps = get_map("map_v1")
    b = sc.gen_ba(r + 1, mean, cov, gts, cam)[r]
    p = b["problem"]
    idx, d2 = orc.associate3d(h, p["points"])
    a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    args = (p["obs_ptr"], p["obs_pose"], p["obs_uvr"], p["obs_oct"])
    ref = orc.joint_optimization(h, cam, p["P"], p["F"], p["poses"], p["prior"], p["points"], a, *args)
    row = dict(P=p["P"], F=p["F"], L=len(p["points"]), nb=b["nb"], prior=b["prior"],
               hip_vs_oracle=max(pose_err(dump["poses_gpu"][j], ref[0][j]) for j in range(p["P"])))
    stereo = np.array([(p["obs_uvr"][p["obs_ptr"][l]:p["obs_ptr"][l + 1], 2] >= 0).any() for l in range(len(p["points"]))])
    if "points_gpu" in dump and stereo.any():
        dp = np.abs(dump["points_gpu"] - ref[1]).max(1)
        row["hip_vs_oracle_stereo_points"] = float(dp[stereo].max())
        row["worst_point"] = int(np.nonzero(stereo)[0][np.argmax(dp[stereo])])
    for k in ("dropped_gpu", "erase_gpu"):
        if k in dump:
            row[k + "_differs"] = int((dump[k][:len(ref[2 if k[0] == "d" else 3])] != ref[2 if k[0] == "d" else 3]).sum())
    # the oracle on the same problem with its points (and their observations) re-ordered
    rng = np.random.default_rng(0)
    sp = (0.0, 0.0)
    psp = 0.0
    L = len(p["points"])
    for _ in range(8):
        perm = rng.permutation(L)
        cnt = np.diff(p["obs_ptr"])[perm]
        optr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        sel = np.concatenate([np.arange(p["obs_ptr"][l], p["obs_ptr"][l + 1]) for l in perm]).astype(int)
        r2 = orc.joint_optimization(h, cam, p["P"], p["F"], p["poses"], p["prior"], p["points"][perm], a[perm], optr,
                                    p["obs_pose"][sel], p["obs_uvr"][sel], p["obs_oct"][sel])
        e = max(pose_err(r2[0][j], ref[0][j]) for j in range(p["P"]))
        sp = (max(sp[0], e[0]), max(sp[1], e[1]))
        back = np.empty_like(r2[1])
        back[perm] = r2[1]  # the re-ordered run's points in the original order
        if stereo.any():
            psp = max(psp, float(np.abs(back - ref[1]).max(1)[stereo].max()))
    row["oracle_reorder_spread"] = sp
    row["oracle_reorder_spread_stereo_points"] = psp
    if with_numpy:
        res = nr.joint_optimization(p["P"], p["F"], p["poses"], p["prior"], p["points"], a, *args, comps, mean, ncam, nprm)
        row["numpy_vs_oracle"] = max(pose_err(res[0][j], ref[0][j]) for j in range(p["P"]))
        row["numpy_vs_hip"] = max(pose_err(res[0][j], dump["poses_gpu"][j]) for j in range(p["P"]))
    return [row]


def fmt(v):
    if isinstance(v, (tuple, list)) and v and isinstance(v[0], (float, np.floating)):
        return "(" + ", ".join("%.3g" % x for x in v) + ")"
    if isinstance(v, (float, np.floating)):
        return "%.3g" % v
    return str(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("--numpy", action="store_true")
    a = ap.parse_args()
    # the oracle's searchCorrespondence works on the view rendered last: one helper that renders first
    def scfp(h, cam_, pose, uv, k):
        orc.render_view(h, cam_, pose)
        return orc.search_correspondence(h, uv, k)
    orc.search_correspondence_for_pose = scfp
    for path in sorted(glob.glob(os.path.join(a.dump, "*.npz"))):
        m = re.match(r"(chain_fallback|chain|tri_far|tri_rejected|tri|track_prior|track|pose|ba)_(map_v[12])_r(\d+)\.npz", os.path.basename(path))
        if not m:
            continue
        kind, mapname, r = m.group(1), m.group(2), int(m.group(3))
        dump = np.load(path)
        if kind == "chain_fallback":
            rows = classify_chain_fallback(mapname, r, dump)
        elif kind in ("tri", "tri_far"):
            rows = classify_tri(mapname, r, dump)
        elif kind == "tri_rejected":
            rows = classify_tri_rejected(mapname, r, dump)
        elif kind in ("track", "track_prior"):
            rows = classify_track(mapname, r, dump, a.numpy, prior=kind == "track_prior")
        elif kind == "ba":
            rows = classify_ba(r, dump, a.numpy)
        else:
            rows = [dict(note="no classifier for this kind")]
        for row in rows:
            print("%-14s %s r%-5d " % (kind, mapname, r) + "  ".join("%s=%s" % (k, fmt(v)) for k, v in row.items()), flush=True)


if __name__ == "__main__":
    main()
