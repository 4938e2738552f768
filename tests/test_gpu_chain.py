"""gl_track_frame_chain (Tracking::trackWithMotionModel -> searchLocalPoints -> trackLocalMap, tracking.cpp:210-376, device resident)
against the oracle's four functions run in sequence with the host's glue in numpy.

The stages feed each other, and the pose of an optimisation agrees with the oracle's to 1e-6, not to the bit - a map point that
projects onto the very edge of a search window could then go either way.  Every stage is therefore checked EXACTLY on the inputs
the device really gave it (the oracle's stage 3 runs from the device's stage-2 pose and associations, its stage 4 from the device's
stage-3 matches), and the free-running oracle chain beside it has to agree on the poses to 1e-6 and on (nearly) all matches."""
import numpy as np
import pytest

from gmmloc_amd import api, synth
from tests.test_gpu_match import CamF

pytestmark = pytest.mark.gpu

TH_MM, TH_LOCAL = 7.0, 3.0


def pose_inputs(f, m_last, m_local=None):
    has_l = m_last >= 0
    has_m = (m_local >= 0) if m_local is not None else np.zeros_like(has_l)
    Xw = np.where(has_l[:, None], f["last_pt"][np.maximum(m_last, 0)],
                  np.where(has_m[:, None], f["mp_pos"][np.maximum(m_local, 0)] if m_local is not None else 0.0, 0.0))
    obs = np.concatenate([f["feat_uv"], f["feat_ur"][:, None].astype(np.float64)], 1)
    oc = np.where(has_l | has_m, f["feat_oct"], -1).astype(np.int32)
    return Xw, obs, oc


def oracle_stage1(o, f):
    k = ("pose_cw", "pose_lw", "feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "last_pt", "last_valid", "last_oct",
         "last_angle", "last_desc")
    m, n = o.search_by_projection_frame(CamF, *[f[x] for x in k], th=TH_MM, mono=False, check_orientation=True)
    if n < 20:  # tracking.cpp:335-342
        m, n = o.search_by_projection_frame(CamF, *[f[x] for x in k], th=2 * TH_MM, mono=False, check_orientation=True)
    return m, n


def oracle_stage3(o, cam, f, pose_mm, m_last_before, m_last_kept):
    """searchLocalPoints from pose_mm: candidates minus the local map points the frame saw in stage 1, kept features taken"""
    NP = len(f["mp_cand"])
    seen = np.zeros(NP, bool)
    l = f["last_to_local"][m_last_before[m_last_before >= 0]]
    seen[l[l >= 0]] = True
    cand = (f["mp_cand"] != 0) & ~seen
    taken = (f["feat_taken"] != 0) | (m_last_kept >= 0)
    twc = o.pose_twc(pose_mm)
    uvr, lvl, vc, dd, iv, n = o.project_map_points(cam, pose_mm, twc, f["mp_pos"], f["mp_normal"], f["mp_max_dist"], f["mp_min_dist"], cand.astype(np.uint8))
    m, nm = o.search_by_projection(cam.width, cam.height, f["feat_uv"], f["feat_ur"], f["feat_oct"], f["feat_desc"], taken.astype(np.uint8), uvr, lvl, vc,
                                   iv, f["mp_desc"], th=TH_LOCAL, nn_ratio=0.8)
    return m, nm, iv


def run_chain(torch, ctx, frames):
    cam, prm = api.Camera(), api.Params()
    a = {k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in frames]).astype(api.CHAIN_DTYPES[k]))).cuda() for k in api.CHAIN_DTYPES}
    out = api.track_frame_chain(ctx, cam, prm, a, th_mm=TH_MM, th_local=TH_LOCAL, nn_ratio=0.8, mono=False)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("NF,NL,NP", [(1200, 1000, 3000), (600, 500, 1200), (300, 1500, 700)])
def test_track_frame_chain_matches_oracle_sequence(gpu, oracle, NF, NL, NP):
    torch, ctx = gpu
    cam = api.Camera()
    frames = [synth.synth_chain_frame(NF, NL, NP, 4000 + 13 * NF + b, cam) for b in range(4)]
    out = run_chain(torch, ctx, frames)
    n_local_total = 0
    for b, f in enumerate(frames):
        # stage 1: bit-exact matches
        m1, n1 = oracle_stage1(oracle, f)
        assert out["counts"][b, 0] == n1
        # stage 2 on those matches: pose 1e-6, the same outliers
        Xw, obs, oc = pose_inputs(f, m1)
        pose2, outl2, ninl2 = oracle.optimize_current_pose(cam, f["pose_cw"], Xw, obs, oc)
        assert np.abs(out["pose_mm"][b] - pose2).max() < 1e-6
        assert out["counts"][b, 1] == ninl2
        kept = np.where(outl2 != 0, -1, m1)
        assert np.array_equal(out["match_last"][b], kept)
        # stage 3 from the DEVICE's stage-2 pose: bit-exact matches and in-view flags
        m3, n3, iv = oracle_stage3(oracle, cam, f, out["pose_mm"][b], m1, kept)
        assert out["counts"][b, 2] == n3 and np.array_equal(out["match_local"][b], m3)
        assert np.array_equal(out["inview"][b], iv)
        n_local_total += n3
        # stage 4 from the device's associations and stage-2 pose: pose 1e-6, the same outliers
        Xw, obs, oc = pose_inputs(f, kept, m3)
        pose4, outl4, ninl4 = oracle.optimize_current_pose(cam, out["pose_mm"][b], Xw, obs, oc)
        assert np.abs(out["pose"][b] - pose4).max() < 1e-6
        assert out["counts"][b, 3] == ninl4
        assert np.array_equal(out["outlier"][b][oc >= 0], outl4[oc >= 0])
        # the free-running oracle chain (its own stage-2 pose): poses within 1e-6, matches all but identical
        m3f, n3f, _ = oracle_stage3(oracle, cam, f, pose2, m1, kept)
        assert (m3f != m3).mean() < 0.01
        Xwf, obsf, ocf = pose_inputs(f, kept, m3f)
        pose4f, _, _ = oracle.optimize_current_pose(cam, pose2, Xwf, obsf, ocf)
        if np.array_equal(m3f, m3):
            assert np.abs(out["pose"][b] - pose4f).max() < 1e-6
        # and the chain tracks: the final pose is closer to the generating pose than the prediction was
        assert np.abs(out["pose"][b] - f["pose_true"]).max() < np.abs(f["pose_cw"] - f["pose_true"]).max()
    assert n_local_total > 0


def test_track_frame_chain_wide_retry(gpu, oracle):
    """a prediction so far off that th = 7 finds fewer than 20 matches: the frame is searched again with th = 14 (tracking.cpp:335-342),
    the frames beside it are not"""
    torch, ctx = gpu
    cam = api.Camera()
    frames = [synth.synth_chain_frame(800, 700, 1500, 5100 + b, cam) for b in range(3)]
    bad = frames[1]
    a = np.deg2rad(3.0)  # 3 degrees of yaw = ~23 px: outside every th = 7 window but the top octaves', inside th = 14 from octave 3
    dq, q0 = np.array([0, np.sin(a / 2), 0, np.cos(a / 2)]), bad["pose_cw"][:4]
    qp = np.concatenate([dq[3] * q0[:3] + q0[3] * dq[:3] + np.cross(dq[:3], q0[:3]), [dq[3] * q0[3] - dq[:3] @ q0[:3]]])
    bad["pose_cw"] = np.concatenate([qp, synth.quat_to_R(dq) @ bad["pose_cw"][4:]])
    out = run_chain(torch, ctx, frames)
    retried = 0
    for b, f in enumerate(frames):
        k = ("pose_cw", "pose_lw", "feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "last_pt", "last_valid", "last_oct",
             "last_angle", "last_desc")
        m7, n7 = oracle.search_by_projection_frame(CamF, *[f[x] for x in k], th=TH_MM, mono=False, check_orientation=True)
        m1, n1 = oracle_stage1(oracle, f)
        retried += int(n7 < 20)
        assert out["counts"][b, 0] == n1
        Xw, obs, oc = pose_inputs(f, m1)
        pose2, outl2, _ = oracle.optimize_current_pose(cam, f["pose_cw"], Xw, obs, oc)
        assert np.array_equal(out["match_last"][b], np.where(outl2 != 0, -1, m1))
    assert retried == 1
