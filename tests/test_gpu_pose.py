"""GPU parity: Tracking::optimizeCurrentPose (B3) vs the oracle.
Tolerance (north_star): pose within 1e-6 m / 1e-6 rad; outlier masks and inlier counts equal."""
import numpy as np
import pytest

from gmmloc_amd import synth, api
import gmmloc_amd

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-6, 1e-6


def pose_err(a, b):
    Ra, Rb = synth.quat_to_R(a[:4]), synth.quat_to_R(b[:4])
    dR = Ra @ Rb.T
    # rotation angle from the skew part (|sin| ~ angle): arccos of the trace has a 1e-8 noise floor at zero
    sk = 0.5 * np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    ang = np.arctan2(np.linalg.norm(sk), (np.trace(dR) - 1) / 2)
    return np.linalg.norm(a[4:] - b[4:]), ang


def make_frames(mean, cov, gt, cam, B, M, seed0, **kw):
    fr = []
    for i in range(B):
        row = gt[(i * 37) % gt.shape[0]]
        fr.append(synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(row), cam, M, seed0 + i, **kw))
    return fr


def run_gpu(gpu, cam, prm, frames):
    torch, ctx = gpu
    pose = torch.from_numpy(np.stack([f["pose_init"] for f in frames])).cuda()
    Xw = torch.from_numpy(np.stack([f["Xw"] for f in frames])).cuda()
    obs = torch.from_numpy(np.stack([f["obs"] for f in frames])).cuda()
    octv = torch.from_numpy(np.stack([f["octave"] for f in frames])).cuda()
    outl, nin = gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose, Xw, obs, octv)
    torch.cuda.synchronize()
    return pose.cpu().numpy(), outl.cpu().numpy(), nin.cpu().numpy()


@pytest.mark.parametrize("kernel", ["1", "4", "8"])  # waves per frame: 1 = large batches, 4 / 8 = few frames
@pytest.mark.parametrize("M,seed", [(300, 100), (1200, 200), (2000, 300), (37, 400)])
def test_optimize_current_pose_matches_oracle(gpu, oracle, map_v1, gt_sync, opt, M, seed, kernel):
    opt("pose_waves", int(kernel))
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 6, M, seed)
    pose, outl, nin = run_gpu(gpu, cam, prm, frames)
    for i, f in enumerate(frames):
        p_ref, o_ref, n_ref = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < TOL_T and dr < TOL_R, (i, dt, dr)
        assert np.array_equal(outl[i], o_ref), (i, int((outl[i] != o_ref).sum()))
        assert nin[i] == n_ref
        # and the optimiser actually recovers the generating pose to noise level
        gt_dt, gt_dr = pose_err(pose[i], f["pose_gt"])
        if M >= 300:
            assert gt_dt < 0.05 and gt_dr < 0.02


@pytest.mark.parametrize("kernel", ["1", "4", "8"])
def test_optimize_current_pose_edge_cases(gpu, oracle, map_v1, gt_sync, opt, kernel):
    opt("pose_waves", int(kernel))
    """< 3 correspondences -> returns 0 and leaves the pose; < 10 -> single round; features
    without map point (octave < 0) are skipped; noise-free input recovers the pose."""
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 4, 64, 900, outlier_frac=0.0)
    frames[0]["octave"][2:] = -1  # 2 edges
    frames[1]["octave"][7:] = -1  # 7 edges
    frames[2]["octave"][::3] = -1
    # frame 3: noise-free
    f = frames[3]
    R, t = synth.quat_to_R(f["pose_gt"][:4]), f["pose_gt"][4:]
    pc = f["Xw"] @ R.T + t
    f["obs"][:, 0] = cam.fx * pc[:, 0] / pc[:, 2] + cam.cx
    f["obs"][:, 1] = cam.fy * pc[:, 1] / pc[:, 2] + cam.cy
    f["obs"][:, 2] = f["obs"][:, 0] - cam.bf / pc[:, 2]
    pose, outl, nin = run_gpu(gpu, cam, prm, frames)
    for i, f in enumerate(frames):
        p_ref, o_ref, n_ref = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < TOL_T and dr < TOL_R, (i, dt, dr)
        assert nin[i] == n_ref
        assert np.array_equal(outl[i], o_ref)
    assert nin[0] == 0 and np.allclose(pose[0], frames[0]["pose_init"] / np.r_[np.ones(4) * np.linalg.norm(frames[0]["pose_init"][:4]), 1, 1, 1])
    dt, dr = pose_err(pose[3], frames[3]["pose_gt"])
    assert dt < 1e-8 and dr < 1e-8


def test_optimize_current_pose_keeps_flags_without_map_point(gpu, oracle, map_v1, gt_sync):
    """is_outlier_[i] is reset only where mappoints_[i] exists (tracking_opt.cpp:63-69): the flags the host holds
    for the other features must survive the call; the flags of the features with a map point are rewritten."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    f = make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 1, 400, 900)[0]
    f["octave"][::3] = -1
    preset = np.random.default_rng(1).integers(0, 2, 400).astype(np.uint8)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).cuda()
    pose, outl = T(f["pose_init"]), T(preset)
    _, nin = gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose, T(f["Xw"]), T(f["obs"]), T(f["octave"]), outlier=outl)
    torch.cuda.synchronize()
    p_ref, o_ref, n_ref = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
    got, none = outl.cpu().numpy()[0], f["octave"] < 0
    assert np.array_equal(got[none], preset[none]) and np.array_equal(got[~none], o_ref[~none]) and int(nin[0]) == n_ref


@pytest.mark.parametrize("regs", [0, 1])
@pytest.mark.parametrize("waves", [0, 1, 4, 8])
@pytest.mark.parametrize("npts", [1, 2])
def test_optimize_current_pose_resets_flags_before_the_too_few_edges_return(gpu, oracle, map_v1, gt_sync, opt, regs, waves, npts):
    """Frames with 1-2 correspondences return 0 at tracking_opt.cpp:139 - AFTER is_outlier_ was reset for the features
    with a map point (:63-69).  The flags the host passes in must come back 0 for those and untouched for the others,
    whichever launch shape answers (the on-chip shapes hold the flags in registers until the end of the kernel)."""
    torch, ctx = gpu
    opt("pose_regs", regs)
    opt("pose_waves", waves)
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    f = make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 1, 200, 901)[0]
    keep = np.array([17, 150])[:npts]
    f["octave"][np.setdiff1d(np.arange(200), keep)] = -1
    preset = np.ones(200, np.uint8)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).cuda()
    pose, outl = T(f["pose_init"]), T(preset)
    _, nin = gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose, T(f["Xw"]), T(f["obs"]), T(f["octave"]), outlier=outl)
    torch.cuda.synchronize()
    p_ref, o_ref, n_ref = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
    got = outl.cpu().numpy()[0]
    want = preset.copy()
    want[keep] = 0
    assert n_ref == 0 and int(nin[0]) == 0 and np.array_equal(o_ref[keep], np.zeros(npts, np.uint8))
    assert np.array_equal(got, want), (got[keep], int((got != want).sum()))


@pytest.mark.parametrize("mode,M", [(-1, 1200), (1, 1200), (1, 1000), (1, 700), (-1, 1025), (-1, 2500), (1, 300)])
def test_optimize_current_pose_compacted_problems(gpu, oracle, map_v1, gt_sync, opt, mode, M):
    """option pose_compact (round 6): a problem with one slot per FEATURE (the reference's frame: 1 200, a few hundred with a map
    point) is compacted to a stride of at most 1 024 where its edges fit, its edge list dealt over the waves.  A batch that mixes
    sparse frames (compacted), full frames (more than 1 024 edges: the full-stride problem, on the device's own decision) and a frame
    without an edge: every frame within 1e-6 of the oracle with equal masks and counts, within 1e-8 of the uncompacted run - bit-equal
    to it where the full-stride problem was kept -, untouched flags where there is no edge, the same bits alone as in the batch."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 6, M, 9100 + M)
    rng = np.random.default_rng(M)
    for b, keep in enumerate((0.3, 1.0, 0.45, 0.0, 0.7, 0.15)):  # share of the slots that keep their edge
        drop = rng.uniform(size=M) >= keep
        frames[b]["octave"] = np.where(drop, -1, frames[b]["octave"]).astype(np.int32)
    n_edges = [int((f["octave"] >= 0).sum()) for f in frames]

    def run(fr, flags_in):
        pose = torch.from_numpy(np.stack([f["pose_init"] for f in fr])).cuda()
        Xw = torch.from_numpy(np.stack([f["Xw"] for f in fr])).cuda()
        obs = torch.from_numpy(np.stack([f["obs"] for f in fr])).cuda()
        octv = torch.from_numpy(np.stack([f["octave"] for f in fr])).cuda()
        outl = torch.from_numpy(flags_in.copy()).cuda()
        nin = torch.zeros(len(fr), dtype=torch.int32).cuda()
        ctx._enter()
        try:
            api._check(ctx.lib.gl_optimize_current_pose(ctx.h, api.C.byref(cam.c()), api.C.byref(prm.c()), len(fr), M, api._ptr(pose), api._ptr(Xw),
                                                        api._ptr(obs), api._ptr(octv), api._ptr(outl), api._ptr(nin)))
        finally:
            ctx._exit()
        torch.cuda.synchronize()
        return pose.cpu().numpy(), outl.cpu().numpy(), nin.cpu().numpy()

    flags = rng.integers(0, 2, (6, M)).astype(np.uint8)  # the caller's is_outlier_: kept where a feature has no map point
    opt("pose_compact", 0)
    p0, o0, n0 = run(frames, flags)
    opt("pose_compact", mode)
    p1, o1, n1 = run(frames, flags)
    assert np.array_equal(o0, o1) and np.array_equal(n0, n1)
    for b, f in enumerate(frames):
        no_edge = f["octave"] < 0
        assert np.array_equal(o1[b][no_edge], flags[b][no_edge])
        assert np.abs(p1[b] - p0[b]).max() < 1e-8
        if n_edges[b] > 1024 or n_edges[b] < 3:
            assert np.array_equal(p1[b], p0[b]), b  # the full-stride problem / no optimisation at all
        pr, outl_r, nin_r = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
        et, er = pose_err(p1[b], pr)
        assert et < TOL_T and er < TOL_R and n1[b] == nin_r and np.array_equal(o1[b][~no_edge], outl_r[~no_edge])
    ps, os_, ns = run(frames[2:3], flags[2:3])
    assert np.array_equal(ps[0], p1[2]) and np.array_equal(os_[0], o1[2])
    if (mode, M) in ((-1, 1200), (1, 700)):
        # a batch large enough for the one-wave-per-frame shapes (edges and flags in memory: the full-stride problems of 1 200 slots,
        # the compacted ones of stride 768; the flags come back through k_pose_scatter): the same bits as in the batch of six
        rep = 260
        pb, ob, nb = run(frames * rep, np.tile(flags, (rep, 1)))
        for r in (0, 131, rep - 1):
            assert np.array_equal(pb[6 * r:6 * r + 6], p1) and np.array_equal(ob[6 * r:6 * r + 6], o1) and np.array_equal(nb[6 * r:6 * r + 6], n1), r
    if M == 1200:
        assert max(n_edges) > 1024 and 3 <= min(n for n in n_edges if n) < 400, n_edges


def test_optimize_current_pose_bit_identical_across_shapes_and_batches(gpu, map_v1, gt_sync, opt):
    """One canonical summation order (gl_refine_pose.hip): the refined pose, the outlier mask and the inlier count of a
    frame are the same BITS on one wave (batch shape), on a wave per group (few frames), and whatever rides in the call."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    for M in (37, 64, 300, 513, 1000, 1200, 1999, 2500):
        frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 6, M, 7000 + M)
        frames[1]["octave"][::5] = -1
        res = {}
        for nw in (1, 4, 8):
            opt("pose_waves", nw)
            res[nw] = run_gpu(gpu, cam, prm, frames)
        for nw in (4, 8):
            for a, b in zip(res[1], res[nw]):
                assert np.array_equal(a, b), (M, nw)
        opt("pose_waves", 0)
        big = run_gpu(gpu, cam, prm, [frames[i % 6] for i in range(1700)])  # > 1536 frames: the one-wave shape
        one = run_gpu(gpu, cam, prm, frames[:1])                             # the frame-at-a-time shape
        for a, b, c in zip(res[1], big, one):
            assert np.array_equal(a, b[:6]) and np.array_equal(a[:1], c), M
        opt("pose_regs", 0)  # the same shapes with the edges re-read from global memory every trial
        for a, b in zip(res[1], run_gpu(gpu, cam, prm, frames)):
            assert np.array_equal(a, b), M
        opt("pose_regs", 1)
