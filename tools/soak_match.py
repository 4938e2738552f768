"""Randomised parity soak of the two matchers (gl_search_by_projection: ORBmatcher::searchByProjection of
searchLocalPoints, orb_matcher.cpp:27-110; gl_search_by_projection_frame: the trackWithMotionModel overload,
orb_matcher.cpp:410-542) against the oracle's sequential restatement: integer work, every index and count must be equal.
    python tools/soak_match.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gmmloc_amd
from gmmloc_amd import api, synth
from tests import oracle_lib
from tests.test_gpu_match import CamF, run_gpu, run_gpu_frame

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
orc = oracle_lib.load()
ctx = gmmloc_amd.Context(0)
bad = dict(local=0, frame=0)
n_checked = dict(local=0, frame=0)
matched = dict(local=0, frame=0)
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(90000 + r)
    # searchLocalPoints overload
    NF, NP = int(rng.integers(1, 2200)), int(rng.integers(1, 3000))
    th, dup = float(rng.choice([1.0, 3.0, 5.0])), float(rng.uniform(0, 0.9))
    ctx.set_option("match_desc_lds", int(rng.integers(0, 2)))
    B = int(rng.integers(1, 5))
    frames = [synth.synth_match_frame(NF, NP, 7919 * r + b, dup_frac=dup) for b in range(B)]
    m, n = run_gpu(torch, ctx, frames, th)
    for b, f in enumerate(frames):
        m_ref, n_ref = orc.search_by_projection(th=th, **f)
        n_checked["local"] += 1
        matched["local"] += int(n_ref)
        if n[b] != n_ref or not np.array_equal(m[b], m_ref):
            bad["local"] += 1
            print("MISMATCH local  round %d frame %d NF %d NP %d th %.0f dup %.2f: %d vs %d matches, %d indices differ"
                  % (r, b, NF, NP, th, dup, int(n[b]), n_ref, int((m[b] != m_ref).sum())), flush=True)
    # trackWithMotionModel overload
    NF, NL = int(rng.integers(50, 2000)), int(rng.integers(50, 2500))
    th = float(rng.choice([7.0, 14.0, 15.0]))
    motion = str(rng.choice(["none", "forward", "backward"]))
    mono, chk = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    frames = [synth.synth_motion_frames(NF, NL, 104729 * r + b, CamF, motion) for b in range(B)]
    m, n = run_gpu_frame(torch, ctx, frames, th, mono, chk)
    for b, f in enumerate(frames):
        m_ref, n_ref = orc.search_by_projection_frame(CamF, th=th, mono=mono, check_orientation=chk, **f)
        n_checked["frame"] += 1
        matched["frame"] += int(n_ref)
        if n[b] != n_ref or not np.array_equal(m[b], m_ref):
            bad["frame"] += 1
            print("MISMATCH frame  round %d frame %d NF %d NL %d th %.0f %s mono %s chk %s: %d vs %d matches, %d indices differ"
                  % (r, b, NF, NL, th, motion, mono, chk, int(n[b]), n_ref, int((m[b] != m_ref).sum())), flush=True)
ctx.set_option("match_desc_lds", -1)
print("matcher soak: %d rounds; frames checked %s, matches compared %s; mismatching frames %s; %.0f s"
      % (rounds, n_checked, matched, bad, time.time() - t0))
sys.exit(1 if sum(bad.values()) else 0)
