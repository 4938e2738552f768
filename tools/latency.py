"""Single-frame / small-batch latency of gl_track_frames on the bench workload (2000 points x 4096 Gaussians):
the north star quotes a latency target (>= 50x the CPU associate+optimize time per frame) next to frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
import bench

Bmax = 256
mean, cov, cam, frames = bench.make_workload(Bmax)
prm = api.Params(); ctx = gmmloc_amd.Context(0); gmm = gmmloc_amd.GMM(ctx, mean, cov, prm)
dev = torch.device("cuda", 0)
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).to(dev)
pose0, Xw0, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
for B in [1, 2, 4, 8, 16, 64, 256]:
    p0, x0, o, oc = pose0[:B].contiguous(), Xw0[:B].contiguous(), obs[:B].contiguous(), octv[:B].contiguous()
    pose, Xw = p0.clone(), x0.clone()
    with torch.cuda.stream(ctx.stream):
        reps = 30
        ctx.timing(True)
        for it in range(reps + 3):
            if it == 3:
                torch.cuda.synchronize(); ctx.timing_read(api.TIMER_ASSOC, reset=True); ctx.timing_read(api.TIMER_BA, reset=True)
                t0 = time.perf_counter()
            pose.copy_(p0); Xw.copy_(x0)
            gmmloc_amd.track_frames(ctx, gmm, cam, prm, pose, Xw, o, oc, want_d2=False)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        a, na = ctx.timing_read(api.TIMER_ASSOC); b, nb = ctx.timing_read(api.TIMER_BA)
        ctx.timing(False)
    print("B=%4d  wall %.3f ms/call   assoc kernel %.3f ms   refine kernel %.3f ms   (%.3f ms/frame)" % (B, wall, a / na, b / nb, wall / B), flush=True)
