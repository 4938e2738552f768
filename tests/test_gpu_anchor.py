"""GPU parity: gl_track_frames_anchored - the per-frame associate + structure-constrained refine WITH the reference's
gauge anchors (the prior edge / fixed first key-frame of localization_opt.cpp:556-581 + factors.cpp:19-53, the fixed
observer key-frames of :491-516) - against the oracle's joint_optimization(P = 1, prior = 1) and (P = 1, F fixed).
Tolerance (north_star): pose within 1e-6 m / 1e-6 rad, decisions exact; bits equal across launch shapes and batches."""
import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import synth, api
from tests.test_gpu_pose import make_frames, pose_err

pytestmark = pytest.mark.gpu


def project(cam, T, X, rng, sig, mono_frac=0.2):
    R, t = synth.quat_to_R(T[:4]), T[4:]
    pc = X @ R.T + t
    z = np.maximum(pc[:, 2], 0.2)
    u = cam.fx * pc[:, 0] / z + cam.cx + rng.standard_normal(len(X)) * sig
    v = cam.fy * pc[:, 1] / z + cam.cy + rng.standard_normal(len(X)) * sig
    ur = u - cam.bf / z + rng.standard_normal(len(X)) * sig * 0.5
    ur = np.where(rng.uniform(size=len(X)) < mono_frac, -1.0, np.maximum(ur, 0.0)).astype(np.float32).astype(np.float64)
    vis = (pc[:, 2] > 0.3) & (u >= 0) & (u < cam.width) & (v >= 0) & (v < cam.height)
    return np.stack([u, v, ur], 1), vis


def add_fixed(f, cam, F, seed):
    """F fixed observer key-frames near the frame's pose, each seeing a random 70 % of the frame's points."""
    rng = np.random.default_rng(seed)
    M = f["Xw"].shape[0]
    fp = np.stack([synth.perturb_pose(f["pose_gt"], rng, 0.05, 0.15) for _ in range(F)])
    fobs, foct = np.zeros((M, F, 3)), np.full((M, F), -1, np.int32)
    for j in range(F):
        oc = rng.integers(0, 8, M).astype(np.int32)
        o, vis = project(cam, fp[j], f["Xw"], rng, 1.2 ** oc)
        seen = vis & (rng.uniform(size=M) < 0.7)
        fobs[:, j], foct[:, j] = o, np.where(seen, oc, -1)
        # a few gross outliers among the fixed observations: the reprojection gating must erase them
        bad = seen & (rng.uniform(size=M) < 0.04)
        fobs[bad, j, 0] += 25.0
    f.update(fixed_pose=fp, fixed_obs=fobs, fixed_oct=foct)
    return f


def oracle_anchored(oracle, h, cam, f, prior, F, prm=None):
    keep = np.nonzero(f["octave"] >= 0)[0]
    Xw = f["Xw"][keep]
    idx, d2 = oracle.associate3d(h, Xw)
    assoc = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    ptr, opose, ouvr, ooct, slot = [0], [], [], [], []
    for n, l in enumerate(keep):
        opose.append(0); ouvr.append(f["obs"][l]); ooct.append(f["octave"][l]); slot.append((n, -1))
        for j in range(F):
            if f["fixed_oct"][l, j] >= 0:
                opose.append(1 + j); ouvr.append(f["fixed_obs"][l, j]); ooct.append(f["fixed_oct"][l, j]); slot.append((n, j))
        ptr.append(len(opose))
    poses = np.concatenate([f["pose_init"][None], f["fixed_pose"][:F]]) if F else f["pose_init"][None]
    p, pts, dropped, erase, it = oracle.joint_optimization(h, cam, 1, F, poses, np.array([1 if prior else 0], np.uint8), Xw, assoc,
                                                           np.array(ptr, np.int32), np.array(opose, np.int32), np.array(ouvr),
                                                           np.array(ooct, np.int32), prm=prm)
    ferase = np.zeros((len(keep), max(F, 1)), np.uint8)
    for o, (n, j) in enumerate(slot):
        if j >= 0:
            ferase[n, j] = erase[o]
    return keep, p[0], pts, np.where(dropped == 1, -1, assoc), ferase[:, :F]


def dev(torch, frames, key):
    return torch.from_numpy(np.ascontiguousarray(np.stack([f[key] for f in frames]))).cuda()


@pytest.mark.parametrize("shape", [0, -1, 1])  # one workgroup per frame / by batch size / one point per thread
@pytest.mark.parametrize("mapname,M,seed", [("v1", 300, 11), ("v1", 1000, 21), ("synth", 2000, 31), ("v1", 2100, 41)])
def test_track_frames_prior_matches_oracle(gpu, oracle, map_v1, gt_sync, opt, mapname, M, seed, shape):
    torch, ctx = gpu
    opt("ba_shape", shape)
    mean, cov = map_v1 if mapname == "v1" else synth.synth_gmm(4096, 1)
    cam, prm = api.Camera(), api.Params()
    frames = make_frames(mean, cov, gt_sync["V1_03_difficult"], cam, 4, M, seed, outlier_frac=0.05)
    frames[1]["octave"][::7] = -1
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    pose, Xw = dev(torch, frames, "pose_init"), dev(torch, frames, "Xw")
    obs, octv = dev(torch, frames, "obs"), dev(torch, frames, "octave")
    prior = torch.tensor([1, 1, 0, 1], dtype=torch.uint8).cuda()  # frame 2 rides unanchored in the same call
    assoc, d2, _ = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, obs, octv, prior=prior)
    torch.cuda.synchronize()
    pose, Xw, assoc = pose.cpu().numpy(), Xw.cpu().numpy(), assoc.cpu().numpy()
    moved = []
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, _ = oracle_anchored(oracle, h, cam, f, bool(prior[i]), 0)
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < 1e-6 and dr < 1e-6, (i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref), (i, int((assoc[i][keep] != a_ref).sum()))
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        assert err[f["obs"][keep][:, 2] >= 0].max() < 1e-6 and err.max() < 1e-5
        # the anchor does something: the oracle's unanchored answer is another pose
        _, p_free, _, _, _ = oracle_anchored(oracle, h, cam, f, False, 0)
        moved.append(pose_err(p_ref, p_free)[0])
    assert max(moved[0], moved[1], moved[3]) > 1e-5 and moved[2] == 0.0
    oracle.gmm_destroy(h)


def test_track_frames_fixed_first_keyframe(gpu, oracle, map_v1, gt_sync):
    """!ba_first_as_prior: key-frame 0 is FIXED (vSE3->setFixed, localization_opt.cpp:578-580) - the pose must come back
    bit for bit and only the points move."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params(ba_first_as_prior=0)
    oprm = oracle_params(oracle, ba_first_as_prior=0)
    frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 3, 500, 77, outlier_frac=0.05)
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    pose, Xw = dev(torch, frames, "pose_init"), dev(torch, frames, "Xw")
    pose0 = pose.clone()
    prior = torch.ones(3, dtype=torch.uint8).cuda()
    assoc, _, _ = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, dev(torch, frames, "obs"), dev(torch, frames, "octave"), prior=prior)
    torch.cuda.synchronize()
    Xw, assoc = Xw.cpu().numpy(), assoc.cpu().numpy()
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, _ = oracle_anchored(oracle, h, cam, f, True, 0, prm=oprm)
        assert max(pose_err(p_ref, f["pose_init"])) < 1e-12  # the oracle leaves the fixed vertex where it was
        assert torch.equal(pose[i], pose0[i])
        assert np.array_equal(assoc[i][keep], a_ref)
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        assert err[f["obs"][keep][:, 2] >= 0].max() < 1e-6 and err.max() < 1e-5
    oracle.gmm_destroy(h)


def oracle_params(oracle, **kw):
    p = type(oracle.prm).from_buffer_copy(oracle.prm)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_track_frames_prior_bit_identical_across_shapes_and_batches(gpu, map_v1, gt_sync, opt):
    """The anchored refine keeps the property of the plain one: a frame's result is the same bits on the batch shape, on
    the latency shape (device-scope and same-XCD exchange), on the general kernel's order-independent parts, and
    whatever else rides in the call."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    for M in (37, 300, 513, 1000, 1999):
        frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 5, M, 8100 + M, outlier_frac=0.05)
        obs, octv = dev(torch, frames, "obs"), dev(torch, frames, "octave")
        prior = torch.ones(5, dtype=torch.uint8).cuda()

        def run(sel=slice(None)):
            pose, Xw = dev(torch, frames, "pose_init")[sel].contiguous(), dev(torch, frames, "Xw")[sel].contiguous()
            a, _, _ = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, obs[sel].contiguous(), octv[sel].contiguous(),
                                                       prior=prior[sel].contiguous(), want_d2=False)
            torch.cuda.synchronize()
            return pose.cpu().numpy(), Xw.cpu().numpy(), a.cpu().numpy()
        opt("ba_shape", 0)
        ref = run()
        opt("ba_shape", 1)
        for same in (1, 0):
            opt("ba_same_xcd", same)
            for a, b in zip(ref, run()):
                assert np.array_equal(a, b, equal_nan=True), (M, same)
        opt("ba_same_xcd", 0)
        opt("ba_shape", -1)
        one = run(slice(2, 3))
        for a, b in zip(ref, one):
            assert np.array_equal(a[2:3], b, equal_nan=True), M


@pytest.mark.parametrize("prior", [0, 1])
@pytest.mark.parametrize("M,F,seed", [(300, 2, 5), (700, 3, 6), (1200, 1, 7)])
def test_track_frames_fixed_observers_match_oracle(gpu, oracle, map_v1, gt_sync, M, F, seed, prior):
    """F fixed observer key-frames per frame (+ optionally the prior): = jointOptimization with P = 1, F fixed poses."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = [add_fixed(f, cam, F, 900 + seed + i) for i, f in
              enumerate(make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 3, M, 40 + seed, outlier_frac=0.05))]
    frames[2]["octave"][::5] = -1
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    pose, Xw = dev(torch, frames, "pose_init"), dev(torch, frames, "Xw")
    pr = torch.full((3,), prior, dtype=torch.uint8).cuda()
    assoc, d2, ferase = gmmloc_amd.track_frames_anchored(
        ctx, g, cam, prm, pose, Xw, dev(torch, frames, "obs"), dev(torch, frames, "octave"), prior=pr,
        fixed_pose=dev(torch, frames, "fixed_pose"), fixed_obs=dev(torch, frames, "fixed_obs"), fixed_oct=dev(torch, frames, "fixed_oct"),
        want_erase=True)
    torch.cuda.synchronize()
    pose, Xw, assoc, ferase = pose.cpu().numpy(), Xw.cpu().numpy(), assoc.cpu().numpy(), ferase.cpu().numpy()
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, fe_ref = oracle_anchored(oracle, h, cam, f, bool(prior), F)
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < 1e-6 and dr < 1e-6, (i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref), (i, int((assoc[i][keep] != a_ref).sum()))
        assert np.array_equal(ferase[i][keep], fe_ref), (i, int((ferase[i][keep] != fe_ref).sum()))
        assert fe_ref.sum() > 0  # (the planted outliers are found)
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        well = (f["obs"][keep][:, 2] >= 0) | ((f["fixed_oct"][keep] >= 0) & ~fe_ref.astype(bool)).any(1)  # stereo, or a second view
        assert err[well].max() < 1e-6 and err.max() < 1e-5, (i, err[well].max(), err.max())
        assert (assoc[i][f["octave"] < 0] == -1).all() and not ferase[i][f["octave"] < 0].any()
        untouched = f["octave"] < 0
        assert np.array_equal(Xw[i][untouched], f["Xw"][untouched])
        # fixed observers pin the gauge: the pose lands on the generating one
        gdt, gdr = pose_err(pose[i], f["pose_gt"])
        assert gdt < 0.03 and gdr < 0.02, (i, gdt, gdr)
    oracle.gmm_destroy(h)


def test_track_frame_host_anchored(gpu, map_v1, gt_sync):
    """gl_track_frame_host_anchored (host buffers in / out) = gl_track_frames_anchored(prior = 1) on device buffers, bit for bit."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    f = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 1, 700, 4242, outlier_frac=0.05)[0]
    pose, Xw = dev(torch, [f], "pose_init"), dev(torch, [f], "Xw")
    a, _, _ = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, dev(torch, [f], "obs"), dev(torch, [f], "octave"),
                                               prior=torch.ones(1, dtype=torch.uint8).cuda(), want_d2=False)
    torch.cuda.synchronize()
    hp = api.HostFramePath(ctx, g, cam, prm)
    hpose, hX = f["pose_init"].copy(), f["Xw"].copy()
    ha = hp.track_frame(hpose, hX, f["obs"], f["octave"], anchored=True)
    assert np.array_equal(hpose, pose.cpu().numpy()[0]) and np.array_equal(hX, Xw.cpu().numpy()[0]) and np.array_equal(ha, a.cpu().numpy()[0])
    upose, uX = f["pose_init"].copy(), f["Xw"].copy()
    hp.track_frame(upose, uX, f["obs"], f["octave"])
    assert not np.array_equal(upose, hpose)  # the unanchored call is another problem
    with pytest.raises(TypeError):
        hp.track_frame(f["pose_init"].copy(), f["Xw"].astype(np.float32), f["obs"], f["octave"])
    with pytest.raises(TypeError):
        hp.track_frame(f["pose_init"].copy(), f["Xw"].copy(), f["obs"], f["octave"].astype(np.int64))


@pytest.mark.parametrize("M,F,seed", [(300, 2, 5), (900, 4, 9), (1500, 3, 10)])
def test_track_frames_fixed_observers_on_chip_route(gpu, oracle, map_v1, gt_sync, opt, M, F, seed):
    """Round 4: up to 4 fixed observers run inside the on-chip per-frame refine (the kFixed instances of k_ba1_fast) instead of
    k_track_pack -> k_ba_gen -> k_track_unpack.  Held to the oracle (above, every F <= 3 case) and here: (a) the packed route
    (option ba_fixed_pack = 1, still the route of F > 4) gives the same decisions and the same pose to 1e-6; (b) a frame's bits
    do not depend on the batch it rides in (alone, among 3, among 40 with other sizes of LDS class excluded by the common
    stride); (c) F = 4 against the oracle."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    frames = [add_fixed(f, cam, F, 300 + seed + i) for i, f in
              enumerate(make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 3, M, 70 + seed, outlier_frac=0.05))]
    frames[1]["octave"][::6] = -1
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)

    def run(fr, prior):
        pose, Xw = dev(torch, fr, "pose_init"), dev(torch, fr, "Xw")
        pr = torch.full((len(fr),), prior, dtype=torch.uint8).cuda()
        a, d2, fe = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, dev(torch, fr, "obs"), dev(torch, fr, "octave"), prior=pr,
                                                     fixed_pose=dev(torch, fr, "fixed_pose"), fixed_obs=dev(torch, fr, "fixed_obs"),
                                                     fixed_oct=dev(torch, fr, "fixed_oct"), want_erase=True)
        torch.cuda.synchronize()
        return pose.cpu().numpy(), Xw.cpu().numpy(), a.cpu().numpy(), fe.cpu().numpy()
    for prior in (0, 1):
        chip = run(frames, prior)
        opt("ba_fixed_pack", 1)
        pack = run(frames, prior)
        opt("ba_fixed_pack", 0)
        for i, f in enumerate(frames):
            keep, p_ref, pts_ref, a_ref, fe_ref = oracle_anchored(oracle, h, cam, f, bool(prior), F)
            for name, out in (("on chip", chip), ("packed", pack)):
                dt, dr = pose_err(out[0][i], p_ref)
                assert dt < 1e-6 and dr < 1e-6, (name, prior, i, dt, dr)
                assert np.array_equal(out[2][i][keep], a_ref) and np.array_equal(out[3][i][keep], fe_ref), (name, prior, i)
            assert not chip[3][i][f["octave"] < 0].any() and (chip[2][i][f["octave"] < 0] == -1).all()
        one = run(frames[1:2], prior)
        many = run(frames + frames * 12 + frames[:1], prior)
        for k in range(4):
            assert np.array_equal(one[k][0], chip[k][1], equal_nan=True), (prior, k)
            assert np.array_equal(many[k][1], chip[k][1], equal_nan=True) and np.array_equal(many[k][39], chip[k][0], equal_nan=True), (prior, k)
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("M", [37, 496, 497, 984, 985, 2000])
def test_track_frames_fixed_observers_lds_class_boundaries_and_mixed_anchors(gpu, oracle, map_v1, gt_sync, M):
    """The on-chip fixed-observer instances at the edges of their three LDS classes (496 / 984 / 2 000 points), with the prior
    flag MIXED over the frames of one call (it is a per-frame byte: prior edge on frame 0 and 2, none on frame 1) and one key-frame
    that observes nothing; then the fixed-pose variant (!ba_first_as_prior) with fixed observers: the flagged poses come back
    bit for bit, the points are held by the observers."""
    torch, ctx = gpu
    mean, cov = map_v1
    F = 2
    cam, prm = api.Camera(), api.Params()
    frames = [add_fixed(f, cam, F, 800 + M + i) for i, f in
              enumerate(make_frames(mean, cov, gt_sync["V1_01_easy"], cam, 3, M, 500 + M, outlier_frac=0.05))]
    frames[1]["fixed_oct"][:, 1] = -1  # the second key-frame of frame 1 sees none of its points
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    flags = np.array([1, 0, 1], np.uint8)

    def run(p):
        pose, Xw = dev(torch, frames, "pose_init"), dev(torch, frames, "Xw")
        a, _, fe = gmmloc_amd.track_frames_anchored(ctx, g, cam, p, pose, Xw, dev(torch, frames, "obs"), dev(torch, frames, "octave"),
                                                    prior=torch.from_numpy(flags).cuda(), fixed_pose=dev(torch, frames, "fixed_pose"),
                                                    fixed_obs=dev(torch, frames, "fixed_obs"), fixed_oct=dev(torch, frames, "fixed_oct"), want_erase=True)
        torch.cuda.synchronize()
        return pose.cpu().numpy(), Xw.cpu().numpy(), a.cpu().numpy(), fe.cpu().numpy()
    pose, Xw, assoc, fe = run(prm)
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, fe_ref = oracle_anchored(oracle, h, cam, f, bool(flags[i]), F)
        dt, dr = pose_err(pose[i], p_ref)
        assert dt < 1e-6 and dr < 1e-6, (M, i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref) and np.array_equal(fe[i][keep], fe_ref), (M, i)
    assert not fe[1][:, 1].any()
    prm0 = api.Params(ba_first_as_prior=0)
    oprm = oracle_params(oracle, ba_first_as_prior=0)
    pose, Xw, assoc, fe = run(prm0)
    for i, f in enumerate(frames):
        keep, p_ref, pts_ref, a_ref, fe_ref = oracle_anchored(oracle, h, cam, f, bool(flags[i]), F, prm=oprm)
        if flags[i]:
            assert np.array_equal(pose[i], f["pose_init"])  # a fixed vertex keeps the caller's bits
        else:
            dt, dr = pose_err(pose[i], p_ref)
            assert dt < 1e-6 and dr < 1e-6, (M, i, dt, dr)
        assert np.array_equal(assoc[i][keep], a_ref) and np.array_equal(fe[i][keep], fe_ref), (M, i)
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        well = (f["obs"][keep][:, 2] >= 0) | ((f["fixed_oct"][keep] >= 0) & ~fe_ref.astype(bool)).any(1)
        # (a handful of points per frame are badly conditioned on their own - outliers that slide along their plane: the decisions
        # above are exact for them too, their coordinates are compared loosely)
        assert np.quantile(err[well], 0.98) < 1e-6 and err[well].max() < 1e-3, (M, i, np.quantile(err[well], 0.98), err[well].max())
    oracle.gmm_destroy(h)
