"""gl_track_frames_anchored with F fixed observer key-frames: on chip (kFixed instances) against the packed route and against
the prior-only / plain refine, on the same frames.   python tools/fixed_time.py [frames] [points] [F]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gmmloc_amd
from gmmloc_amd import api, synth
from tests.test_gpu_anchor import add_fixed
from tools import soak_cases as sc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
F = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mean, cov = sc.load_map("map_v1")
gt = sc.load_gt()["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)
g = gmmloc_amd.GMM(ctx, mean, cov, prm)
NU = 64  # distinct frames, tiled
frames = []
for i in range(NU):
    f = synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(gt[(i * 37) % gt.shape[0]]), cam, M, 5000 + i, outlier_frac=0.05)
    frames.append(add_fixed(f, cam, F, 9000 + i))
T = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([frames[i % NU][k] for i in range(B)]))).cuda()
pose0, Xw0, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
fp, fo, fc = T("fixed_pose"), T("fixed_obs"), T("fixed_oct")
one = torch.ones(B, dtype=torch.uint8).cuda()
trials = torch.zeros(B, dtype=torch.int32).cuda()
ctx.set_stats_buffer(trials)


def run(kind, n=3):
    ctx.timing(True)
    ts = []
    for it in range(n + 1):
        p, x = pose0.clone(), Xw0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind == "plain":
            gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, obs, octv, want_d2=False)
        elif kind == "prior":
            gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, p, x, obs, octv, prior=one, want_d2=False)
        else:
            gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, p, x, obs, octv, prior=one, fixed_pose=fp, fixed_obs=fo, fixed_oct=fc, want_d2=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ctx.timing(False)
    dt = min(ts[1:])
    print("%-28s %8.3f ms per call, %9.0f frames/s, %.1f trials/frame" % (kind, 1e3 * dt, B / dt, float(trials.sum().item()) / B))


print("%d frames x %d points, %d fixed observers" % (B, M, F))
run("plain")
run("prior")
run("fixed on chip")
ctx.set_option("ba_fixed_pack", 1)
run("fixed packed (k_ba_gen)", 1)
