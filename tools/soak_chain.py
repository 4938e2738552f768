"""Randomised parity soak of gl_track_frame_chain (Tracking::track's device half, tracking.cpp:34-118: trackWithMotionModel ->
trackKeyFrame where it fails -> searchLocalPoints -> trackLocalMap) against the oracle's functions in sequence: every stage is checked
on the inputs the device really gave it (tests/chain_glue.py: matches / in-view flags / outlier masks / counts equal, poses within 1e-6).
Rounds mix plain frames, wide retries, temporal points, and batches WITH a reference key-frame in which some predictions are 10 degrees
off (the fallback) or every last-frame point is temporal (20+ matches that count for nothing).
    python tools/soak_chain.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gmmloc_amd
from gmmloc_amd import api, synth
from tests import chain_glue as G
from tests import oracle_lib
from tests.test_gpu_chain import run_chain

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
orc = oracle_lib.load()
ctx = gmmloc_amd.Context(0)
cam = api.Camera()
bad = {}
modes = [0, 0, 0]
frames_checked = replaced = 0
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(770000 + r)
    NF, NL, NP = int(rng.integers(150, 1600)), int(rng.integers(150, 1800)), int(rng.integers(200, 4000))
    B = int(rng.integers(1, 5))
    with_kf = r % 3 == 1
    NK = int(rng.integers(60, 1500)) if with_kf else 0
    frames = []
    for b in range(B):
        tf = float(rng.choice([0.0, 0.0, 0.25, 1.0])) if r % 2 else 0.0
        rot = 10.0 if (with_kf and rng.uniform() < 0.4) else None
        frames.append(synth.synth_chain_frame(NF, NL, NP, 6007 * r + b, cam, temporal_frac=tf, NK=NK, pred_rot_deg=rot))
    if r % 5 == 0:  # a prediction far enough off for the wide retry (tracking.cpp:340-346) on one frame of the call
        f = frames[0]
        a = np.deg2rad(float(rng.uniform(2.0, 4.0)))
        dq, q0 = np.array([0, np.sin(a / 2), 0, np.cos(a / 2)]), f["pose_cw"][:4]
        qp = np.concatenate([dq[3] * q0[:3] + q0[3] * dq[:3] + np.cross(dq[:3], q0[:3]), [dq[3] * q0[3] - dq[:3] @ q0[:3]]])
        f["pose_cw"] = np.concatenate([qp, synth.quat_to_R(dq) @ f["pose_cw"][4:]])
    out = run_chain(torch, ctx, frames)
    for b, f in enumerate(frames):
        frames_checked += 1
        try:
            c = G.check_chain(orc, cam, f, out, b)
            modes[c["front"]["mode"]] += 1
            replaced += c["replaced"]
        except AssertionError as e:
            bad[str(e)] = bad.get(str(e), 0) + 1
            print("MISMATCH %s: round %d frame %d NF %d NL %d NP %d NK %d" % (e, r, b, NF, NL, NP, NK), flush=True)
print("chain soak: %d rounds, %d frames (modes motion model / key-frame / lost: %s; %d temporal points replaced): mismatches %s; %.0f s"
      % (rounds, frames_checked, modes, replaced, bad or "none", time.time() - t0))
