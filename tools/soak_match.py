"""Randomised parity soak of the five matchers (gl_fuse_search: the matching half of Localization::fuseObservations, localization.cpp:226-318;
gl_search_by_projection: ORBmatcher::searchByProjection of
searchLocalPoints, orb_matcher.cpp:27-110; gl_search_by_projection_frame: the trackWithMotionModel overload,
orb_matcher.cpp:410-542; gl_search_for_triangulation :141-293; gl_search_by_bow :295-408) against the oracle's sequential
restatement: integer work, every index and count must be equal.
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
from tests.test_gpu_match import CamF, FUSE_KEYS, run_gpu, run_gpu_frame, _pack_pairs, _pack_bow, _pack_fuse

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
orc = oracle_lib.load()
ctx = gmmloc_amd.Context(0)
bad = dict(local=0, frame=0, tri=0, bow=0, fuse=0)
n_checked = dict(local=0, frame=0, tri=0, bow=0, fuse=0)
matched = dict(local=0, frame=0, tri=0, bow=0, fuse=0)
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(90000 + r)
    # searchLocalPoints overload
    NF, NP = int(rng.integers(1, 2200)), int(rng.integers(1, 3000))
    th, dup = float(rng.choice([1.0, 3.0, 5.0])), float(rng.uniform(0, 0.9))
    ctx.set_option("match_desc_lds", int(rng.integers(0, 2)))
    B = int(rng.integers(1, 5))
    frames = [synth.synth_match_frame(NF, NP, 7919 * r + b, dup_frac=dup, float_uv=(r + b) % 4 != 0) for b in range(B)]
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
    frames = [synth.synth_motion_frames(NF, NL, 104729 * r + b, CamF, motion, float_uv=(r + b) % 4 != 0) for b in range(B)]
    m, n = run_gpu_frame(torch, ctx, frames, th, mono, chk)
    for b, f in enumerate(frames):
        m_ref, n_ref = orc.search_by_projection_frame(CamF, th=th, mono=mono, check_orientation=chk, **f)
        n_checked["frame"] += 1
        matched["frame"] += int(n_ref)
        if n[b] != n_ref or not np.array_equal(m[b], m_ref):
            bad["frame"] += 1
            print("MISMATCH frame  round %d frame %d NF %d NL %d th %.0f %s mono %s chk %s: %d vs %d matches, %d indices differ"
                  % (r, b, NF, NL, th, motion, mono, chk, int(n[b]), n_ref, int((m[b] != m_ref).sum())), flush=True)
    # searchForTriangulation and searchByBoW on the same kind of key-frame pairs
    cam = api.Camera()
    pairs = [synth.synth_tri_search_pair(int(rng.integers(20, 1500)), int(rng.integers(20, 1500)), 31337 * r + b, cam,
                                         n_nodes=int(rng.integers(3, 260)), only_stereo_frac=float(rng.uniform(0, 1)), pad=int(rng.integers(0, 2)))
             for b in range(B)]
    k1, k2, fm, ep = _pack_pairs(torch, pairs)
    only_stereo, chk = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    m, n = api.search_for_triangulation(ctx, k1, k2, fm, ep, only_stereo, chk)
    torch.cuda.synchronize()
    m, n = m.cpu().numpy(), n.cpu().numpy()
    for b, p in enumerate(pairs):
        m_ref, n_ref = orc.search_for_triangulation(p["kf1"], p["kf2"], p["fmat"], p["epipole"], only_stereo, chk)
        n_checked["tri"] += 1
        matched["tri"] += int(n_ref)
        if n[b] != n_ref or not np.array_equal(m[b, :len(m_ref)], m_ref):
            bad["tri"] += 1
            print("MISMATCH tri    round %d pair %d: %d vs %d matches" % (r, b, int(n[b]), n_ref), flush=True)
    bows = [synth.synth_bow_pair(int(rng.integers(20, 1500)), int(rng.integers(20, 1500)), 27644437 + 977 * r + b, cam,
                                 n_nodes=int(rng.integers(3, 260)), mp_frac=float(rng.uniform(0.05, 1))) for b in range(B)]
    kf, fr = _pack_bow(torch, bows)
    ratio = float(rng.choice([0.6, 0.7, 0.8, 0.9]))
    m, n = api.search_by_bow(ctx, kf, fr, ratio, chk)
    torch.cuda.synchronize()
    m, n = m.cpu().numpy(), n.cpu().numpy()
    for b, p in enumerate(bows):
        m_ref, n_ref = orc.search_by_bow(p[0], p[1], ratio, chk)
        n_checked["bow"] += 1
        matched["bow"] += int(n_ref)
        if n[b] != n_ref or not np.array_equal(m[b, :len(m_ref)], m_ref):
            bad["bow"] += 1
            print("MISMATCH bow    round %d pair %d ratio %.1f chk %s: %d vs %d matches" % (r, b, ratio, chk, int(n[b]), n_ref), flush=True)
    # the matching half of fuseObservations: float coordinates (the record walk), arbitrary doubles (the walk from global memory), mixed batches
    camf = api.Camera()
    camf.width, camf.height = 752, 480
    th = float(rng.choice([3.0, 5.0]))
    fuses = [synth.synth_fuse_frame(int(rng.integers(5, 2400)), int(rng.integers(5, 3200)), 611953 * r + b, float_coords=(r + b) % 3 != 0) for b in range(B)]
    bi, bd = api.fuse_search(ctx, camf, *_pack_fuse(torch, fuses), th=th)
    torch.cuda.synchronize()
    bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    for b, f in enumerate(fuses):
        ri, rd, n_ref = orc.fuse_search(f["width"], f["height"], *[f[k] for k in FUSE_KEYS], th=th)
        n_checked["fuse"] += 1
        matched["fuse"] += int(n_ref)
        if not (np.array_equal(bi[b, :len(ri)], ri) and np.array_equal(bd[b, :len(ri)], rd)):
            bad["fuse"] += 1
            print("MISMATCH fuse   round %d key-frame %d th %.0f: %d indices differ" % (r, b, th, int((bi[b, :len(ri)] != ri).sum())), flush=True)
ctx.set_option("match_desc_lds", -1)
print("matcher soak: %d rounds; frames checked %s, matches compared %s; mismatching frames %s; %.0f s"
      % (rounds, n_checked, matched, bad, time.time() - t0))
sys.exit(1 if sum(bad.values()) else 0)
