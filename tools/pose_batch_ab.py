#!/usr/bin/env python3
"""gl_optimize_current_pose in batches, device time by HIP events: B frames of M slots of which a share holds an edge.
    [GMMLOC_POSE_REGS=2] python tools/pose_batch_ab.py [label]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gmmloc_amd
from gmmloc_amd import api, synth

label = sys.argv[1] if len(sys.argv) > 1 else "lib"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz"))
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)
res = {"label": label}
U = 32
for B, M, frac in ((2048, 1200, 0.35), (2048, 1200, 0.6), (2048, 1000, 1.0), (4096, 1000, 1.0), (2048, 512, 0.8), (300, 1200, 0.35)):
    rng = np.random.default_rng(M + int(100 * frac))
    fr = []
    for u in range(U):
        f = synth.synth_frame(d["mean"], d["cov"], synth.gt_row_to_Tcw(gt[100 + 7 * u]), cam, M, 50 + u)
        oc = f["octave"].copy()
        if frac < 1.0:
            oc[rng.uniform(size=M) >= frac] = -1
        fr.append((f["pose_init"], f["Xw"], f["obs"], oc.astype(np.int32)))
    T = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([fr[b % U][k] for b in range(B)]))).cuda()
    p0, xw, ob, oc = T(0), T(1), T(2), T(3)
    pw = p0.clone()
    outl = torch.zeros((B, M), dtype=torch.uint8, device="cuda")

    def call():
        pw.copy_(p0)
        api.optimize_current_pose(ctx, cam, prm, pw, xw, ob, oc, outl)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.stream):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0.record(ctx.stream)
        for _ in range(5):
            call()
        e1.record(ctx.stream)
    torch.cuda.synchronize()
    res["B%d_M%d_%d%%_ms" % (B, M, int(100 * frac))] = e0.elapsed_time(e1) / 5
    res["B%d_M%d_%d%%_sum" % (B, M, int(100 * frac))] = float(pw.double().sum().item())
print(json.dumps(res))
