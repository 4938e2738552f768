"""Randomised parity soak on the GPU box: many seeds of the key-frame chain (search2d -> checkMapAssociation,
createMapPoints), of the per-frame path and of the local BA against the oracle; prints the first mismatch and a summary.
    python tools/soak.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
from tests import oracle_lib
from tests.test_gpu_pose import pose_err
from tests.test_gpu_track import oracle_track

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
orc = oracle_lib.load()
ctx = gmmloc_amd.Context(0)
cam, prm = api.Camera(), api.Params()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
gts = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))
bad = 0
soft = 0  # numerical deviations on ill-conditioned inputs (reported, not failures)
t0 = time.time()
for mapname, seqs in (("map_v1", ["V1_01_easy", "V1_02_medium", "V1_03_difficult"]), ("map_v2", ["V2_01_easy", "V2_02_medium"])):
    d = np.load(os.path.join(ROOT, "tests", "golden", mapname + ".npz")); mean, cov = d["mean"], d["cov"]
    g = gmmloc_amd.GMM(ctx, mean, cov, prm); h = orc.gmm_create(mean, cov)
    for r in range(rounds):
        rng = np.random.default_rng(1000 * len(mapname) + r)
        gt = gts[seqs[r % len(seqs)]]
        while True:  # a key-frame pair with a baseline: the caller of createMapPoints skips the others (the sequences
            ia = int(rng.integers(0, gt.shape[0] - 40)); ib = ia + int(rng.integers(3, 30))  # start with a standing robot)
            if np.linalg.norm(gt[ia][1:4] - gt[ib][1:4]) > 0.05:
                break
        p1, p2 = synth.gt_row_to_Tcw(gt[ia]), synth.gt_row_to_Tcw(gt[ib])
        N = int(rng.integers(50, 900))
        # ---- key-frame association chain
        f = synth.synth_frame(mean, cov, p1, cam, N, 7000 + r, mono_frac=0.0, outlier_frac=0.1)
        pts = f["Xw"] + rng.standard_normal((N, 3)) * 0.02
        octv = f["octave"].copy(); octv[rng.uniform(size=N) < 0.05] = -1
        cand, ncand, vids, nview = g.search2d(cam, T(p1[None]), T(f["obs"][None, :, :2].copy()), None, k=5, view_cap=4096)
        pd = T(pts[None])
        out = api.check_map_association(ctx, g, cam, prm, T(p1[None]), pd, T(f["obs"][None]), T(octv[None]), cand, ncand)
        torch.cuda.synchronize()
        ids, _, _, _ = orc.render_view(h, cam, p1)
        c_ref, n_ref = orc.search_correspondence(h, f["obs"][:, :2].copy(), 5)
        ok = int(nview[0]) == len(ids) and np.array_equal(vids[0].cpu().numpy()[:len(ids)], ids) and \
            np.array_equal(cand[0].cpu().numpy(), c_ref) and np.array_equal(ncand[0].cpu().numpy(), n_ref)
        keep = octv >= 0
        o_ref, p_ref = orc.check_map_association(h, cam, p1, pts[keep], f["obs"][keep], octv[keep], c_ref[keep], n_ref[keep])
        pg = pd[0].cpu().numpy()[keep]
        hit = o_ref >= 0  # associated points to 1e-9; the fallback branch (no association, point moved towards the nearest
        ok = ok and np.array_equal(out[0].cpu().numpy()[keep], o_ref) and np.allclose(pg[hit], p_ref[hit], rtol=0, atol=1e-9)
            # mean) also runs on inconsistent outliers (u < 0, negative disparity): 5 GN steps on garbage, reported only
        soft += int((np.abs(pg[~hit] - p_ref[~hit]).max(1) > 1e-4).sum()) if (~hit).any() else 0
        # ---- createMapPoints
        m = synth.synth_tri_matches(mean, cov, p1, p2, cam, int(rng.integers(20, 500)), 9000 + r)
        x_ref, t_ref, cc_ref = orc.create_map_points(h, cam, **m)
        keys = ("pose1", "uvr1", "depth1", "oct1", "pose2", "uvr2", "depth2", "oct2", "cand1", "n1", "cand2", "n2")
        x, t, c = api.create_map_points(ctx, g, cam, prm, *[T(m[k]) for k in keys])
        torch.cuda.synchronize()
        # near-parallel rays triangulate to points 10^3 .. 10^8 m away where Gauss-Newton is chaotic (the numpy
        # restatement disagrees with both there): neither the coordinates nor the sign-dependent checks compare
        xg = x.cpu().numpy()
        with np.errstate(invalid="ignore"):
            sane = (np.linalg.norm(x_ref, axis=1) < 100.0) & (np.linalg.norm(np.nan_to_num(xg, nan=1e9), axis=1) < 100.0)
        acc = t_ref > 0
        ok2 = np.array_equal(t.cpu().numpy()[sane], t_ref[sane]) and np.array_equal(c.cpu().numpy()[sane], cc_ref[sane]) and \
            np.allclose(xg[sane & acc], x_ref[sane & acc], rtol=0, atol=1e-8)
        soft += int((~sane).sum())
        # ---- per-frame path (every 4th round: the oracle's joint_optimization is slow)
        ok3 = True
        if r % 4 == 0:
            M = int(rng.integers(30, 700))
            ft = synth.synth_frame(mean, cov, p1, cam, M, 11000 + r, outlier_frac=0.05)
            pose, Xw = T(ft["pose_init"][None]), T(ft["Xw"][None])
            assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, T(ft["obs"][None]), T(ft["octave"][None]))
            torch.cuda.synchronize()
            keep, p_ref, pts_ref, a_ref, idx0, d20 = oracle_track(orc, h, cam, ft)
            dt, dr = pose_err(pose.cpu().numpy()[0], p_ref)
            # the Levenberg schedule (5 / 5 / 40 iterations) can stop before convergence on small, outlier-ridden frames:
            # the end point then depends on the summation order (it does between the library's own launch shapes): reported
            ok3 = dt < 1e-2 and dr < 1e-2 and np.array_equal(assoc.cpu().numpy()[0][keep], a_ref) and np.array_equal(d2.cpu().numpy()[0][keep], d20)
            soft += 0 if (dt < 1e-6 and dr < 1e-6) else 1
        # ---- optimizeCurrentPose (random size: every launch shape of the kernel over the rounds)
        Mp = int(rng.integers(5, 1300))
        fp = synth.synth_frame(mean, cov, p1, cam, Mp, 13000 + r, outlier_frac=0.08)
        fp["octave"][rng.uniform(size=Mp) < 0.1] = -1
        pose = T(fp["pose_init"][None])
        outl, nin = api.optimize_current_pose(ctx, cam, prm, pose, T(fp["Xw"][None]), T(fp["obs"][None]), T(fp["octave"][None]))
        torch.cuda.synchronize()
        pr, orf, nr_ = orc.optimize_current_pose(cam, fp["pose_init"], fp["Xw"], fp["obs"], fp["octave"])
        dt, dr = pose_err(pose.cpu().numpy()[0], pr)
        ok3 = ok3 and dt < 1e-6 and dr < 1e-6 and np.array_equal(outl.cpu().numpy()[0], orf) and int(nin[0]) == nr_
        if not (ok and ok2 and ok3):
            bad += 1
            print("MISMATCH", mapname, "round", r, "frames", ia, ib, "N", N, "chain", ok, "createMapPoints", ok2, "track+pose", ok3, flush=True)
    orc.gmm_destroy(h)
# ---- local BA: random window sizes and forced workgroup counts (lanes per point / waves per block combinations)
from tests.test_gpu_ba import make_ba_problem, run_gpu, check
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean, cov = d["mean"], d["cov"]
g = gmmloc_amd.GMM(ctx, mean, cov, prm); h = orc.gmm_create(mean, cov)
rng = np.random.default_rng(5)
for r in range(max(10, rounds // 4)):
    P, F, L = int(rng.integers(1, 8)), int(rng.integers(0, 4)), int(rng.integers(20, 400))
    nb = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
    if nb:
        os.environ["GMMLOC_BAGEN_NB"] = str(nb)
    else:
        os.environ.pop("GMMLOC_BAGEN_NB", None)
    p = make_ba_problem(mean, cov, gts["V1_01_easy"], cam, P, F, L, 500 + r, bool(rng.integers(0, 2)))
    idx, d2 = orc.associate3d(h, p["points"])
    a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    try:
        check([p], [a], run_gpu((torch, ctx), g, cam, prm, [p], [a]), orc, h, cam)
    except AssertionError as e:  # windows without a fixed pose or prior have gauge freedom; small ones stop unconverged
        soft += 1
        print("deviation local BA round", r, "P F L NB", P, F, L, nb, str(e)[:120], flush=True)
os.environ.pop("GMMLOC_BAGEN_NB", None)
orc.gmm_destroy(h)
print("soak: %d rounds per map, %d decision mismatches, %d numerical deviations on ill-conditioned inputs, %.0f s" % (rounds, bad, soft, time.time() - t0))
sys.exit(1 if bad else 0)
