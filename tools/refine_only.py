"""The bench step's refine alone (gl_track_frames on the bench frames), a few launches: the workload of the PC-sampling /
counter runs that look INSIDE k_ba1_fast (tools/pc_sample.sh).   python tools/refine_only.py [frames] [launches] [prior]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import gmmloc_amd
from gmmloc_amd import api

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PRIOR = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = int(sys.argv[4]) if len(sys.argv) > 4 else bench.N_PTS
mean, cov, cam, frames = bench.make_workload(NF)
ctx = gmmloc_amd.Context(0)
prm = api.Params()
g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose0, Xw0, obs, octv = T("pose_init"), T("Xw")[:, :M].contiguous(), T("obs")[:, :M].contiguous(), T("octave")[:, :M].contiguous()
one = torch.ones(NF, dtype=torch.uint8).cuda()
trials = torch.zeros(NF, dtype=torch.int32).cuda()
ctx.set_stats_buffer(trials)
ctx.timing(True)
for it in range(NL + 1):
    p, x = pose0.clone(), Xw0.clone()
    if it == 1:
        ctx.timing_read(api.TIMER_BA, reset=True)
    if PRIOR:
        gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, p, x, obs, octv, prior=one, want_d2=False)
    else:
        gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, obs, octv, want_d2=False)
    torch.cuda.synchronize()
ms, n = ctx.timing_read(api.TIMER_BA)
ntr = float(trials.sum().item())
ks = ms / 1e3 / max(n, 1)
tf = bench.FLOP_PER_POINT_TRIAL * M * ntr / ks / 1e12
import hashlib
fp = hashlib.sha256(p.cpu().numpy().tobytes() + x.cpu().numpy().tobytes()).hexdigest()[:16]  # bits of the last launch's poses + points
print("refine: %d frames x %d points, %.3f ms per launch, %.1f trials/frame, %.2f TFLOP/s = %.4f of the fp64 VALU peak, result bits %s"
      % (NF, M, 1e3 * ks, ntr / NF, tf, tf / bench.PEAK_FP64_VALU_TFLOPS, fp))
if os.environ.get("REFINE_SAVE"):  # (two builds whose canonical orders differ: compare the poses instead of the bits)
    np.save(os.environ["REFINE_SAVE"], p.cpu().numpy())
if os.environ.get("REFINE_COMPARE"):
    q = np.load(os.environ["REFINE_COMPARE"])
    print("max |pose - %s| = %.3e" % (os.environ["REFINE_COMPARE"], np.abs(p.cpu().numpy() - q).max()))
