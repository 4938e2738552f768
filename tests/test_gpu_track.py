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


@pytest.mark.parametrize("mapname,M,seed", [("v1", 400, 10), ("v1", 2000, 20), ("synth", 1000, 30), ("v1", 2100, 40)])
def test_track_frames_matches_oracle(gpu, oracle, map_v1, gt_sync, mapname, M, seed):
    torch, ctx = gpu
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
        # points: well-constrained ones (stereo observation, still an inlier) to 1e-6 m; a monocular
        # or gated-out point is only held by the LM damping along its ray -> looser bound
        err = np.abs(Xw[i][keep] - pts_ref).max(1)
        stereo = f["obs"][keep][:, 2] >= 0
        worst = int(np.argmax(err))
        assert err[stereo].max() < 1e-6, (i, worst, err[worst], f["obs"][keep][worst], a_ref[worst])
        assert err.max() < 1e-4
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
    frames[4]["octave"][100:] = -1                    # padded short frame
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
        assert np.array_equal(Xw[i][untouched], f["Xw"][untouched])  # padding rows are not written
    oracle.gmm_destroy(h)
