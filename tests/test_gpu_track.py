"""GPU parity: gl_track_frames (exact all-component association + single-pose jointOptimization with
Schur-marginalised points) vs the oracle's associate3d + joint_optimization.
Tolerance (north_star): pose within 1e-6 m / 1e-6 rad; indices exact."""
import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import synth, api
from tests.test_gpu_pose import make_frames, pose_err

pytestmark = pytest.mark.gpu


def oracle_track(oracle, h, cam, f):
    keep = np.nonzero(f["octave"] >= 0)[0]
    Xw = f["Xw"][keep]
    idx, d2 = oracle.associate3d(h, Xw)
    assoc = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    L = len(keep)
    poses, pts, dropped, erase, it = oracle.joint_optimization(
        h, cam, 1, 0, f["pose_init"][None], np.zeros(1, np.uint8), Xw, assoc, np.arange(L + 1, dtype=np.int32),
        np.zeros(L, np.int32), f["obs"][keep], f["octave"][keep])
    final = np.where(dropped == 1, -1, assoc)
    return keep, poses[0], pts, final, idx, d2


@pytest.mark.parametrize("shape", [0, -1, 1])  # one workgroup per frame / by batch size / one point per thread
@pytest.mark.parametrize("mapname,M,seed", [("v1", 400, 10), ("v1", 2000, 20), ("synth", 1000, 30), ("v1", 2100, 40)])
def test_track_frames_matches_oracle(gpu, oracle, map_v1, gt_sync, opt, mapname, M, seed, shape):
    torch, ctx = gpu
    opt("ba_shape", shape)
    mean, cov = map_v1 if mapname == "v1" else synth.synth_gmm(4096, 1)
    cam, prm = api.Camera(), api.Params()
    gt = gt_sync["V1_03_difficult"]
    frames = make_frames(mean, cov, gt, cam, 4, M, seed, outlier_frac=0.05)
    frames[1]["octave"][::7] = -1
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    pose = torch.from_numpy(np.stack([f["pose_init"] for f in frames])).cuda()
    Xw = torch.from_numpy(np.stack([f["Xw"] for f in frames])).cuda()
    obs = torch.from_numpy(np.stack([f["obs"] for f in frames])).cuda()
    octv = torch.from_numpy(np.stack([f["octave"] for f in frames])).cuda()
    pose_b, Xw_b = pose.clone(), Xw.clone()
    assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, obs, octv)
    # without the chi2 output the cell index does not resolve the points above the gate (they are
    # dropped either way): the results must be bit-identical
    assoc_b, _ = gmmloc_amd.track_frames(ctx, g, cam, prm, pose_b, Xw_b, obs, octv, want_d2=False)
    torch.cuda.synchronize()
    assert torch.equal(assoc, assoc_b) and torch.equal(pose, pose_b) and torch.equal(Xw, Xw_b)
    pose, Xw, assoc, d2 = pose.cpu().numpy(), Xw.cpu().numpy(), assoc.cpu().numpy(), d2.cpu().numpy()
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, idx0, d20 = oracle_track(oracle, h, cam, f)
        assert np.array_equal(d2[i][keep], d20)
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < 1e-6 and dr < 1e-6, (i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref), (i, int((assoc[i][keep] != a_ref).sum()))
        # points: stereo observations to 1e-6 m; a monocular point has its depth from the GMM edge alone (a plane that may
        # graze the ray) or from the LM damping: measured <= 1.0e-6 m on these frames (stereo <= 1.1e-7), held to 5e-6
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        stereo = f["obs"][keep][:, 2] >= 0
        worst = int(np.argmax(err))
        assert err[stereo].max() < 1e-6, (i, worst, err[worst], f["obs"][keep][worst], a_ref[worst])
        assert err.max() < 5e-6, (i, worst, err[worst])
        assert (assoc[i][f["octave"] < 0] == -1).all()
        # structure actually constrains the pose: close to the generating pose
        gdt, gdr = pose_err(pose[i], f["pose_gt"])
        assert gdt < 0.1 and gdr < 0.05
    oracle.gmm_destroy(h)


def test_track_frames_edge_cases(gpu, oracle, map_v1, gt_sync):
    """Frames the tracker can hand over: no map point at all, a handful of points, nothing near the map (no GMM
    edge survives the gate: plain reprojection refinement), every observation monocular, one frame of many sizes
    sharing a padded batch."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    gt = gt_sync["V1_01_easy"]
    M = 300
    frames = make_frames(mean, cov, gt, cam, 6, M, 500, outlier_frac=0.02)
    frames[0]["octave"][:] = -1                       # nothing to optimise: pose must come back unchanged
    frames[1]["octave"][5:] = -1                      # 5 points
    frames[2]["Xw"] += np.array([40.0, -35.0, 20.0])  # far from every component; consistent observations are kept
    frames[3]["obs"][:, 2] = -1.0                     # monocular only
    frames[4]["octave"][100:] = -1                    # padded short frame ...
    frames[4]["Xw"][100:200] = np.nan                 # ... whose padding rows hold garbage
    frames[4]["Xw"][200:] = np.inf
    frames[4]["obs"][100:] = -np.inf
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
    pose, Xw, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
    assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, obs, octv)
    torch.cuda.synchronize()
    pose, Xw, assoc = pose.cpu().numpy(), Xw.cpu().numpy(), assoc.cpu().numpy()
    assert np.array_equal(pose[0], frames[0]["pose_init"]) and (assoc[0] == -1).all()
    assert np.array_equal(Xw[0], frames[0]["Xw"])
    assert (assoc[2] == -1).all()
    for i in range(1, 6):
        f = frames[i]
        keep, p_ref, pts_ref, a_ref, idx0, d20 = oracle_track(oracle, h, cam, f)
        dt, dr = pose_err(pose[i], p_ref)
        assert np.isfinite(pose[i]).all() and dt < 1e-6 and dr < 1e-6, (i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref), i
        assert (assoc[i][f["octave"] < 0] == -1).all()
        untouched = f["octave"] < 0
        assert np.array_equal(Xw[i][untouched], f["Xw"][untouched], equal_nan=True)  # padding rows are not written
    oracle.gmm_destroy(h)


def test_track_frames_full_size_known_answer(gpu):
    """BASELINE configs[1] shape (2 000 points x 4 096 Gaussians per frame), too big for the oracle in a test,
    checked through properties: with noise-free observations of points that sit exactly on their Gaussians'
    means every residual vanishes at the generating pose, so (1) that pose is the answer, from a perturbed
    start, to 1e-9 (measured: 1e-13); (2) on perturbed points the reported chi2 / association are the
    exhaustive N x K sweep's, bit for bit, and only gated-out points lose their component; (3) a second run
    from the answer does not move it (idempotence)."""
    torch, ctx = gpu
    B, M, K = 12, 2000, 4096
    cam, prm = api.Camera(), api.Params()
    mean, cov = synth.synth_gmm(K, 1)
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(2024)
    pose_gt, pose0, Xgt, obs = [], [], [], []
    for b in range(B):
        eye = rng.uniform([-3.5, -2.5, 0.8], [2.5, 3.5, 2.2])
        a = rng.uniform(0, 2 * np.pi)
        T = synth.look_at_pose(eye, eye + 3.0 * np.array([np.cos(a), np.sin(a), rng.uniform(-0.3, 0.3)]))
        R, t = synth.quat_to_R(T[:4]), T[4:]
        pc = mean @ R.T + t
        u = cam.fx * pc[:, 0] / pc[:, 2] + cam.cx
        v = cam.fy * pc[:, 1] / pc[:, 2] + cam.cy
        vis = np.nonzero((pc[:, 2] > 0.5) & (pc[:, 2] < 9.0) & (u >= 0) & (u < cam.width) & (v >= 0) & (v < cam.height))[0]
        assert vis.size > 30
        sel = vis[rng.integers(0, vis.size, M)]
        mono = rng.uniform(size=M) < 0.2
        obs.append(np.stack([u[sel], v[sel], np.where(mono, -1.0, u[sel] - cam.bf / pc[sel, 2])], 1))
        Xgt.append(mean[sel])
        pose_gt.append(T)
        pose0.append(synth.perturb_pose(T, rng, 0.005, 0.01))
    Tn = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).cuda()
    obs_t = Tn(obs)
    octv = torch.from_numpy(rng.integers(0, 8, (B, M)).astype(np.int32)).cuda()
    # (2) association at perturbed points (2 mm = two sigma of the thinnest axis: part of them is gated out)
    Xn = Tn(Xgt) + torch.from_numpy(rng.normal(0, 0.002, (B, M, 3))).cuda()
    idx_x, d2_x = g.associate3d(Xn.reshape(-1, 3), api.ASSOC_EXHAUSTIVE)
    assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, Tn(pose0), Xn.clone(), obs_t, octv)
    torch.cuda.synchronize()
    assert torch.equal(d2.reshape(-1), d2_x)
    kept = (d2_x <= 9.0).reshape(B, M).cpu().numpy()
    assert 0.3 < kept.mean() < 0.98
    a_np, ix = assoc.cpu().numpy(), idx_x.reshape(B, M).cpu().numpy()
    still = a_np >= 0
    assert np.array_equal(a_np[still], ix[still]) and not (still & ~kept).any()
    # (1) the known answer
    pose, Xw = Tn(pose0), Tn(Xgt)
    assoc, _ = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, obs_t, octv, want_d2=False)
    torch.cuda.synchronize()
    p_np, X_np = pose.cpu().numpy(), Xw.cpu().numpy()
    assert (assoc.cpu().numpy() >= 0).all()
    for b in range(B):
        dt, dr = pose_err(p_np[b], pose_gt[b])
        assert dt < 1e-9 and dr < 1e-9, (b, dt, dr)
        assert np.abs(X_np[b] - Xgt[b]).max() < 1e-6
    # (3) idempotence
    pose2, Xw2 = pose.clone(), Xw.clone()
    gmmloc_amd.track_frames(ctx, g, cam, prm, pose2, Xw2, obs_t, octv, want_d2=False)
    torch.cuda.synchronize()
    for b in range(B):
        dt, dr = pose_err(pose2.cpu().numpy()[b], p_np[b])
        assert dt < 1e-9 and dr < 1e-9, (b, dt, dr)


def _run_track(torch, ctx, g, cam, prm, frames, sel=None):
    fr = frames if sel is None else [frames[i] for i in sel]
    T = lambda k: torch.from_numpy(np.stack([f[k] for f in fr])).cuda()
    pose, Xw = T("pose_init"), T("Xw")
    assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, T("obs"), T("octave"))
    torch.cuda.synchronize()
    return pose.cpu().numpy(), Xw.cpu().numpy(), assoc.cpu().numpy(), d2.cpu().numpy()


def test_track_frames_bit_identical_across_shapes(gpu, map_v1, gt_sync, opt):
    """ONE summation order (gl_ba_fast_impl.hpp): the frame-at-a-time shape (one point per thread, a frame dealt to
    1..4 co-resident workgroups, sums exchanged between them) and the batch shape (one workgroup of G waves per
    frame) must return the same BITS - poses, points, associations - on odd sizes: every chunks-per-group count,
    last groups with one chunk or a few points, padding rows, all three LDS classes."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(77)
    for M, B in [(37, 3), (64, 2), (130, 4), (300, 5), (496, 2), (513, 3), (640, 2), (777, 7), (1000, 2), (1025, 2),
                 (1537, 5), (1999, 1), (2000, 3)]:
        frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, B, M, 1000 + M, outlier_frac=0.05)
        for f in frames:
            f["octave"][rng.uniform(size=M) < 0.1] = -1
        res = {}
        for shape in (0, 1):
            opt("ba_shape", shape)
            res[shape] = _run_track(torch, ctx, g, cam, prm, frames)
        for a, b, what in zip(res[0], res[1], ("pose", "points", "assoc", "chi2")):
            assert np.array_equal(a, b, equal_nan=True), (M, B, what, np.abs(a - b).max())
        opt("ba_same_xcd", 1)  # the latency shape with the opt-in same-XCD form of its exchange (the default is device scope)
        for a, b, what in zip(res[1], _run_track(torch, ctx, g, cam, prm, frames), ("pose", "points", "assoc", "chi2")):
            assert np.array_equal(a, b, equal_nan=True), (M, B, what)
        opt("ba_same_xcd", 0)


def test_track_frames_two_frames_per_cu_same_bits(gpu, map_v1, gt_sync, opt):
    """Option ba_two_frames: the 2 000-point class of the plain batch refine as two frames per CU (bafd2000x - a wave owns TWO
    groups of the canonical order, the points' hand-over slots live in global memory).  Same summation order, so the same BITS as
    the eight-wave shape: every group count of the class (G = 4 ... 8: the wave's second group absent, short, full), a last group
    of a few points, padding rows, and a batch with more frames than the device holds workgroups (the persistent frame queue)."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(78)
    opt("ba_shape", 0)
    for M, B in [(1001, 3), (1024, 2), (1025, 2), (1280, 3), (1300, 2), (1537, 5), (1793, 2), (1999, 3), (2000, 4), (2000, 1100)]:
        U = min(B, 6)
        uniq = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, U, M, 5000 + M, outlier_frac=0.05)
        for f in uniq:
            f["octave"][rng.uniform(size=M) < 0.1] = -1
        frames = [uniq[b % U] for b in range(B)]
        res = {}
        for two in (0, 1):
            opt("ba_two_frames", two)
            res[two] = _run_track(torch, ctx, g, cam, prm, frames)
        opt("ba_two_frames", 0)
        for a, b, what in zip(res[0], res[1], ("pose", "points", "assoc", "chi2")):
            assert np.array_equal(a, b, equal_nan=True), (M, B, what, np.abs(a - b).max())


def test_track_frames_rendezvous_gives_up_cleanly(gpu, map_v1, gt_sync, opt):
    """The latency shape is launched plainly (a cooperative launch costs 31 us per call): every exchange between the
    workgroups of a frame has a time limit, the workgroups write to a staging area, and the one-workgroup kernel that
    always follows copies the staged result of a complete frame and recomputes the others from the untouched inputs.
    With the limit at 0 (a workgroup that does not find all its siblings' words at its first look gives up) most frames
    take the recompute path at the rendezvous; with ba_test_abort_seq = n the last workgroup of EVERY frame gives up at its
    n-th exchange - in the middle of the schedule, after thousands of LDS updates, or at the very last reduction while its
    siblings have already staged their share.  The caller must get the same bits every time."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    for M, B in [(300, 5), (1000, 2), (1999, 1), (2000, 9)]:
        frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, B, M, 3000 + M, outlier_frac=0.05)
        opt("ba_shape", 0)
        ref = _run_track(torch, ctx, g, cam, prm, frames)
        opt("ba_shape", 1)
        opt("ba_rendezvous_us", 50000)  # (a generous limit: whether 200 us always suffice is a property of the box, not of the code)
        ctx.counter_read(0)
        res = _run_track(torch, ctx, g, cam, prm, frames)
        assert ctx.counter_read(0) == 0  # an undisturbed launch completes on the latency shape
        for a, b, what in zip(ref, res, ("pose", "points", "assoc", "chi2")):
            assert np.array_equal(a, b, equal_nan=True), (M, B, what)
        for limit_us, abort_seq in ((0, 0), (200, 1), (200, 2), (200, 9), (200, 40), (50000, 0)):
            opt("ba_rendezvous_us", limit_us)
            opt("ba_test_abort_seq", abort_seq)
            for _ in range(3):  # the outcome of the rendezvous at limit 0 varies from launch to launch
                res = _run_track(torch, ctx, g, cam, prm, frames)
                for a, b, what in zip(ref, res, ("pose", "points", "assoc", "chi2")):
                    assert np.array_equal(a, b, equal_nan=True), (M, B, limit_us, abort_seq, what)
            redone = ctx.counter_read(0)
            if abort_seq:
                assert redone == 3 * B  # every frame gave up and was recomputed by the follow-up kernel
            elif limit_us == 50000:
                assert redone == 0
        opt("ba_rendezvous_us", 200)
        opt("ba_test_abort_seq", 0)
    # the last exchange of a frame: find it (the trial count is data dependent) and abort exactly there
    frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 1, 1000, 4000, outlier_frac=0.05)
    opt("ba_shape", 0)
    ref = _run_track(torch, ctx, g, cam, prm, frames)
    opt("ba_shape", 1)
    last = None
    for seq in range(60, 220):
        opt("ba_test_abort_seq", seq)
        res = _run_track(torch, ctx, g, cam, prm, frames)
        for a, b, what in zip(ref, res, ("pose", "points", "assoc", "chi2")):
            assert np.array_equal(a, b, equal_nan=True), (seq, what)
        if ctx.counter_read(0) == 0:  # the schedule has fewer exchanges than `seq`: the abort never fired
            last = seq - 1
            break
    assert last is not None and last >= 60


def test_track_frames_result_independent_of_batch(gpu, map_v1, gt_sync, opt):
    """A drop-in caller gets the same bits for a frame whatever rides with it: alone (B = 1: latency shape), in a
    handful, and inside a batch larger than the chip (B > CUs: batch shape)."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    opt("ba_shape", -1)
    for M, B in [(300, 700), (1200, 300), (2000, 280)]:
        frames = make_frames(mean, cov, gt_sync["V1_03_difficult"], cam, 6, M, 5000 + M, outlier_frac=0.08)
        frames = [frames[i % 6] for i in range(B)]
        big = _run_track(torch, ctx, g, cam, prm, frames)
        for sel in ([0], [1], [0, 1, 2, 3, 4, 5], list(range(40))):
            small = _run_track(torch, ctx, g, cam, prm, frames, sel)
            for a, b, what in zip(big, small, ("pose", "points", "assoc", "chi2")):
                assert np.array_equal(a[sel], b, equal_nan=True), (M, len(sel), what)


def test_track_frames_step32_option(gpu, oracle, opt):
    """The refine's point step: exact fp64 (default, what bench.py times) against the fp32-cached step of round 1
    (option ba_step32) on 8 of the actual bench frames (bench.py seeds, 4 096-Gaussian synthetic map, M = 2 000,
    noisy observations, 10 % outliers), both against the oracle.  The default must hold the north-star tolerance
    with margin; the fp32 variant is only required to stay close (it is an opt-in speed switch)."""
    import bench
    torch, ctx = gpu
    mean, cov, cam, frames = bench.make_workload(8)
    prm = api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    ref = [oracle_track(oracle, h, cam, f) for f in frames]
    worst = {}
    for s32 in (0, 1):
        opt("ba_step32", s32)
        pose, Xw, assoc, d2 = _run_track(torch, ctx, g, cam, prm, frames)
        w = (0.0, 0.0)
        for i, (keep, p_ref, pts_ref, a_ref, idx0, d20) in enumerate(ref):
            dt, dr = pose_err(pose[i], p_ref)
            w = (max(w[0], dt), max(w[1], dr))
            assert np.array_equal(d2[i][keep], d20)
            if s32 == 0:
                assert np.array_equal(assoc[i][keep], a_ref), i
        worst[s32] = w
    oracle.gmm_destroy(h)
    assert worst[0][0] < 1e-6 and worst[0][1] < 1e-6, worst   # exact step: the north-star tolerance (measured 7e-8)
    assert worst[1][0] < 1e-4 and worst[1][1] < 1e-4, worst   # fp32-cached step: close, not bound to 1e-6


@pytest.mark.parametrize("M", [450, 900, 1500])
def test_track_frames_lds_classes(gpu, oracle, map_v1, gt_sync, opt, M):
    """The batch shape on the three LDS classes (496 / 1 000 / 2 000 points: 4 / 2 / 1 frames per CU); small test
    batches would default to the latency shape, so the batch shape is forced here."""
    torch, ctx = gpu
    opt("ba_shape", 0)
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 3, M, 3000 + M, outlier_frac=0.05)
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    pose, Xw, assoc, d2 = _run_track(torch, ctx, g, cam, prm, frames)
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, idx0, d20 = oracle_track(oracle, h, cam, f)
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < 1e-6 and dr < 1e-6, (i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref)
    oracle.gmm_destroy(h)
