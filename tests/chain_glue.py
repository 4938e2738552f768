"""The host glue of Tracking::track (tracking.cpp:34-118) in numpy around the ORACLE's functions: what gl_track_frame_chain does on the
device, stage by stage, for ONE frame (a dict of synth.synth_chain_frame).  Test infrastructure (tests/test_gpu_chain.py,
tools/soak_chain.py); nothing in the product imports it.

The stages feed each other, and the pose of an optimisation agrees with the oracle's to 1e-6, not to the bit - a map point that
projects onto the very edge of a search window could then go either way.  `check_chain` therefore checks every stage EXACTLY on
the inputs the device really gave it (stage 3 runs from the device's stage-2 pose, stage 4 from the device's stage-3 matches)."""
import numpy as np

from tests.test_gpu_match import CamF

TH_MM, TH_LOCAL = 7.0, 3.0
S1_KEYS = ("pose_cw", "pose_lw", "feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "last_pt", "last_valid", "last_oct",
           "last_angle", "last_desc")


def has_fallback(f):
    return "kf_desc" in f


def pose_inputs(f, m_last, m_local=None, m_kf=None):
    """the pose problem of optimizeCurrentPose (tracking_opt.cpp:63-133): a feature's map point is its local-map match
    (orb_matcher.cpp:104 overwrites), else its last-frame match, else its key-frame match"""
    NF = len(f["feat_oct"])
    none = -np.ones(NF, np.int64)
    m_local = none if m_local is None else m_local
    m_kf = none if m_kf is None else m_kf
    has_m, has_l, has_k = m_local >= 0, m_last >= 0, m_kf >= 0
    Xw = np.zeros((NF, 3))
    if has_k.any():
        Xw = np.where(has_k[:, None], f["kf_pt"][np.maximum(m_kf, 0)], Xw)
    Xw = np.where(has_l[:, None], f["last_pt"][np.maximum(m_last, 0)], Xw)
    Xw = np.where(has_m[:, None], f["mp_pos"][np.maximum(m_local, 0)], Xw)
    obs = np.concatenate([f["feat_uv"], f["feat_ur"][:, None].astype(np.float64)], 1)
    oc = np.where(has_l | has_m | has_k, f["feat_oct"], -1).astype(np.int32)
    return Xw, obs, oc


def oracle_stage1(o, f):
    m, n = o.search_by_projection_frame(CamF, *[f[x] for x in S1_KEYS], th=TH_MM, mono=False, check_orientation=True)
    if n < 20:  # tracking.cpp:340-346
        m, n = o.search_by_projection_frame(CamF, *[f[x] for x in S1_KEYS], th=2 * TH_MM, mono=False, check_orientation=True)
    return m, n


def oracle_front(o, cam, f):
    """trackWithMotionModel (+ trackKeyFrame when it returns less than 10 and the frame has a key-frame) -> dict"""
    NF = len(f["feat_oct"])
    m1, n1 = oracle_stage1(o, f)
    Xw, obs, oc = pose_inputs(f, m1)
    pose2, outl2, ninl2 = o.optimize_current_pose(cam, f["pose_cw"], Xw, obs, oc)
    drop = np.where((outl2 != 0) & (m1 >= 0), m1, -1)
    kept = np.where(outl2 != 0, -1, m1)
    observed = f["last_observed"] if "last_observed" in f else np.ones(len(f["last_oct"]), np.uint8)
    ret = 0 if n1 < 20 else int(((kept >= 0) & (observed[np.maximum(kept, 0)] != 0)).sum())
    r = dict(m1=m1, n1=n1, pose=pose2, ninl=ninl2, match_last=kept, drop_src=drop, ret_mm=ret, mode=0, nbow=0, ret_kf=0,
             match_kf=-np.ones(NF, np.int64), drop_kf=-np.ones(NF, np.int64))
    if has_fallback(f) and ret < 10:  # tracking.cpp:50-72
        kf = dict(angle=f["kf_angle"], desc=f["kf_desc"], has_mp=f["kf_has_mp"], node_id=f["kf_node_id"], node_ptr=f["kf_node_ptr"], node_idx=f["kf_node_idx"])
        fr = dict(angle=f["feat_angle"], desc=f["feat_desc"], node_id=f["feat_node_id"], node_ptr=f["feat_node_ptr"], node_idx=f["feat_node_idx"])
        mk, nbow = o.search_by_bow(kf, fr, 0.7, True)
        Xw, obs, oc = pose_inputs(f, -np.ones(NF, np.int64), None, mk)
        pose_k, outl_k, ninl_k = o.optimize_current_pose(cam, f["pose_lw"], Xw, obs, oc)
        kept_k = np.where(outl_k != 0, -1, mk)
        nk = int((kept_k >= 0).sum())
        r.update(pose=pose_k, ninl=ninl_k, match_last=-np.ones(NF, np.int64), match_kf=kept_k, drop_kf=np.where((outl_k != 0) & (mk >= 0), mk, -1),
                 nbow=nbow, ret_kf=nk, mode=2 if nk < 10 else 1)
    return r


def oracle_stage3(o, cam, f, pose, match_last, match_kf, drop_src, drop_kf):
    """searchLocalPoints from `pose`: candidates minus the local map points the frame holds or dropped, features with an OBSERVED map
    point taken (orb_matcher.cpp:74-76)"""
    NP = len(f["mp_cand"])
    seen = np.zeros(NP, bool)
    for idx, tab in ((match_last, "last_to_local"), (drop_src, "last_to_local"), (match_kf, "kf_to_local"), (drop_kf, "kf_to_local")):
        if tab in f and (idx >= 0).any():
            l = f[tab][idx[idx >= 0]]
            seen[l[(l >= 0) & (l < NP)]] = True
    cand = (f["mp_cand"] != 0) & ~seen
    observed = f["last_observed"] if "last_observed" in f else np.ones(len(f["last_oct"]), np.uint8)
    taken = (f["feat_taken"] != 0) | ((match_last >= 0) & (observed[np.maximum(match_last, 0)] != 0)) | (match_kf >= 0)
    twc = o.pose_twc(pose)
    uvr, lvl, vc, dd, iv, n = o.project_map_points(cam, pose, twc, f["mp_pos"], f["mp_normal"], f["mp_max_dist"], f["mp_min_dist"], cand.astype(np.uint8))
    m, nm = o.search_by_projection(cam.width, cam.height, f["feat_uv"], f["feat_ur"], f["feat_oct"], f["feat_desc"], taken.astype(np.uint8), uvr, lvl, vc,
                                   iv, f["mp_desc"], th=TH_LOCAL, nn_ratio=0.8)
    return m, nm, iv


def check_chain(o, cam, f, out, b, split=False):
    """every stage of frame b of the device's outputs `out` against the oracle; returns dict(front=..., replaced, pose4, m3) and raises
    AssertionError with the stage's name on a mismatch"""
    r = oracle_front(o, cam, f)
    assert out["counts"][b, 0] == r["n1"], "stage 1: matches"
    assert out["counts2"][b, 0] == r["ret_mm"], "stage 2: return value of trackWithMotionModel"
    assert out["counts2"][b, 3] == r["mode"], "mode"
    assert np.abs(out["pose_mm"][b] - r["pose"]).max() < 1e-6, "stage 2 / 2b: pose"
    assert out["counts"][b, 1] == r["ninl"], "stage 2 / 2b: inliers"
    assert np.array_equal(out["drop_src"][b], r["drop_src"]), "stage 2: dropped matches"
    if has_fallback(f):
        assert out["counts2"][b, 1] == r["nbow"] and out["counts2"][b, 2] == r["ret_kf"], "stage 2b: counts"
        assert np.array_equal(out["match_kf"][b], r["match_kf"]) and np.array_equal(out["drop_kf"][b], r["drop_kf"]), "stage 2b: matches"
    # stage 3 from the DEVICE's pose: bit-exact matches and in-view flags
    m3, n3, iv = oracle_stage3(o, cam, f, out["pose_mm"][b], r["match_last"], r["match_kf"], r["drop_src"], r["drop_kf"])
    assert out["counts"][b, 2] == n3 and np.array_equal(out["match_local"][b], m3), "stage 3: matches"
    assert np.array_equal(out["inview"][b], iv), "stage 3: in-view flags"
    replaced = (m3 >= 0) & (r["match_last"] >= 0)  # a temporal point's feature took the local map point
    assert np.array_equal(out["match_last"][b], np.where(replaced, -1, r["match_last"])), "final last-frame associations"
    # stage 4 from the device's associations and pose
    Xw, obs, oc = pose_inputs(f, r["match_last"], m3, r["match_kf"])
    pose4, outl4, ninl4 = o.optimize_current_pose(cam, out["pose_mm"][b], Xw, obs, oc)
    assert np.abs(out["pose"][b] - pose4).max() < 1e-6, "stage 4: pose"
    assert out["counts"][b, 3] == ninl4, "stage 4: inliers"
    assert np.array_equal(out["outlier"][b][oc >= 0], outl4[oc >= 0]), "stage 4: outliers"
    return dict(front=r, m3=m3, n3=n3, replaced=int(replaced.sum()), pose4=pose4)
