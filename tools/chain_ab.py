#!/usr/bin/env python3
"""Device time of gl_track_frame_chain and gl_optimize_current_pose, calls enqueued back to back between two HIP events on the context's
stream (the host enqueues faster than the device runs: what the events see is the device).  For A/Bs between libraries / options:
    [GMMLOC_HIP_LIB=...] [GMMLOC_POSE_DUAL=0] python tools/chain_ab.py [label]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gmmloc_amd
from gmmloc_amd import api, synth

label = sys.argv[1] if len(sys.argv) > 1 else "lib"
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)
NF, NL, NP = 1200, 1000, 3000


def pack(frames, keys):
    return {k: torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f[k]) for f in frames]).astype(keys[k]))).cuda() for k in keys if k in frames[0]}


def ev_us(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.stream):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0.record(ctx.stream)
        for _ in range(n):
            fn()
        e1.record(ctx.stream)
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


res = {"label": label}
frames = [synth.synth_chain_frame(NF, NL, NP, 7000 + b, cam) for b in range(16)]
per = []
for f in frames:
    one = pack([f], api.CHAIN_DTYPES)
    out = api.track_frame_chain(ctx, cam, prm, one)
    per.append(ev_us(lambda: api.track_frame_chain(ctx, cam, prm, one, out=out), 60))
res["chain_one_frame_us_mean_min_max_of_16"] = [float(np.mean(per)), float(np.min(per)), float(np.max(per))]
from tests.test_gpu_chain import pack as pack_kf  # noqa: E402

fk = pack_kf(torch, [synth.synth_chain_frame(NF, NL, NP, 7000, cam, NK=NL, pred_rot_deg=10.0)])
outk = api.track_frame_chain(ctx, cam, prm, fk)
res["chain_through_fallback_us"] = ev_us(lambda: api.track_frame_chain(ctx, cam, prm, fk, out=outk), 60)
big = pack([frames[b % 16] for b in range(2048)], api.CHAIN_DTYPES)
outb = api.track_frame_chain(ctx, cam, prm, big)
t = ev_us(lambda: api.track_frame_chain(ctx, cam, prm, big, out=outb), 5)
res["chain_batch_2048_ms"] = t / 1e3
res["chain_batch_frames_per_s"] = 2048 / (t * 1e-6)
# the public pose entry point: one problem of 1 200 slots, 420 edges
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz"))
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
f = synth.synth_frame(d["mean"], d["cov"], synth.gt_row_to_Tcw(gt[100]), cam, 1280, 50)
rng = np.random.default_rng(1)
for M, n_act in ((1200, 420), (1200, 1200), (1024, 1024)):
    oc = np.full(M, -1, np.int32)
    act = np.sort(rng.choice(M, n_act, replace=False)) if n_act < M else np.arange(M)
    oc[act] = f["octave"][:n_act]
    Xw, ob = np.zeros((M, 3)), np.zeros((M, 3))
    Xw[act], ob[act] = f["Xw"][:n_act], f["obs"][:n_act]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).cuda()
    p0, xw, o, o_c = T(f["pose_init"]), T(Xw), T(ob), T(oc)
    pw = p0.clone()
    outl = torch.zeros((1, M), dtype=torch.uint8, device="cuda")

    def call():
        pw.copy_(p0)
        api.optimize_current_pose(ctx, cam, prm, pw, xw, o, o_c, outl)

    res["pose_%d_slots_%d_edges_us" % (M, n_act)] = ev_us(call, 100)
print(json.dumps(res))
