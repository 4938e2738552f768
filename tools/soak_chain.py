"""Randomised parity soak of gl_track_frame_chain (Tracking::trackWithMotionModel -> searchLocalPoints -> trackLocalMap,
tracking.cpp:210-376) against the oracle's four functions: every stage is checked on the inputs the device really gave it
(tests/test_gpu_chain.py: matches / in-view flags / outlier masks / counts equal, poses within 1e-6).
    python tools/soak_chain.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gmmloc_amd
from gmmloc_amd import api, synth
from tests import oracle_lib
from tests.test_gpu_chain import oracle_stage1, oracle_stage3, pose_inputs, run_chain
from tests.test_gpu_match import CamF

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
orc = oracle_lib.load()
ctx = gmmloc_amd.Context(0)
cam = api.Camera()
bad = dict(stage1=0, pose2=0, mask2=0, stage3=0, pose4=0, mask4=0)
worst = dict(pose2=0.0, pose4=0.0)
frames_checked = retried = 0
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(770000 + r)
    NF, NL, NP = int(rng.integers(150, 1600)), int(rng.integers(150, 1800)), int(rng.integers(200, 4000))
    B = int(rng.integers(1, 5))
    frames = [synth.synth_chain_frame(NF, NL, NP, 6007 * r + b, cam) for b in range(B)]
    if r % 5 == 0:  # a prediction far enough off for the wide retry (tracking.cpp:335-342) on one frame of the call
        f = frames[0]
        a = np.deg2rad(float(rng.uniform(2.0, 4.0)))
        dq, q0 = np.array([0, np.sin(a / 2), 0, np.cos(a / 2)]), f["pose_cw"][:4]
        qp = np.concatenate([dq[3] * q0[:3] + q0[3] * dq[:3] + np.cross(dq[:3], q0[:3]), [dq[3] * q0[3] - dq[:3] @ q0[:3]]])
        f["pose_cw"] = np.concatenate([qp, synth.quat_to_R(dq) @ f["pose_cw"][4:]])
    out = run_chain(torch, ctx, frames)
    for b, f in enumerate(frames):
        frames_checked += 1
        tag = "round %d frame %d NF %d NL %d NP %d" % (r, b, NF, NL, NP)
        k = ("pose_cw", "pose_lw", "feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "last_pt", "last_valid", "last_oct",
             "last_angle", "last_desc")
        _, n7 = orc.search_by_projection_frame(CamF, *[f[x] for x in k], th=7.0, mono=False, check_orientation=True)
        retried += int(n7 < 20)
        m1, n1 = oracle_stage1(orc, f)
        if out["counts"][b, 0] != n1:
            bad["stage1"] += 1
            print("MISMATCH stage 1 (count)", tag, int(out["counts"][b, 0]), n1, flush=True)
            continue
        Xw, obs, oc = pose_inputs(f, m1)
        pose2, outl2, ninl2 = orc.optimize_current_pose(cam, f["pose_cw"], Xw, obs, oc)
        d2 = float(np.abs(out["pose_mm"][b] - pose2).max())
        worst["pose2"] = max(worst["pose2"], d2)
        kept = np.where(outl2 != 0, -1, m1)
        if d2 >= 1e-6:
            bad["pose2"] += 1
            print("DEVIATION pose 2", tag, d2, flush=True)
        if out["counts"][b, 1] != ninl2 or not np.array_equal(out["match_last"][b], kept):
            bad["mask2"] += 1
            print("MISMATCH stage 2 (outliers)", tag, flush=True)
            continue
        m3, n3, iv = oracle_stage3(orc, cam, f, out["pose_mm"][b], m1, kept)
        if out["counts"][b, 2] != n3 or not np.array_equal(out["match_local"][b], m3) or not np.array_equal(out["inview"][b], iv):
            bad["stage3"] += 1
            print("MISMATCH stage 3", tag, int(out["counts"][b, 2]), n3, int((out["match_local"][b] != m3).sum()), flush=True)
            continue
        Xw, obs, oc = pose_inputs(f, kept, m3)
        pose4, outl4, ninl4 = orc.optimize_current_pose(cam, out["pose_mm"][b], Xw, obs, oc)
        d4 = float(np.abs(out["pose"][b] - pose4).max())
        worst["pose4"] = max(worst["pose4"], d4)
        if d4 >= 1e-6:
            bad["pose4"] += 1
            print("DEVIATION pose 4", tag, d4, flush=True)
        if out["counts"][b, 3] != ninl4 or not np.array_equal(out["outlier"][b][oc >= 0], outl4[oc >= 0]):
            bad["mask4"] += 1
            print("MISMATCH stage 4 (outliers)", tag, flush=True)
print("chain soak: %d rounds, %d frames (%d took the wide retry): %s; worst pose deviation stage 2 %.2e, stage 4 %.2e; %.0f s"
      % (rounds, frames_checked, retried, bad, worst["pose2"], worst["pose4"], time.time() - t0))
