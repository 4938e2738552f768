"""Persistent kernel (bagen_mode 1) against the pipelined shape (2) of gl_joint_optimization over window sizes and small batches (ms per call):
   python tools/ba_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def timed(P, F, L, B, mode, reps=4):
    ctx.set_option("bagen_mode", mode)
    p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
    idx, d2 = g.associate3d(T(p["points"]))
    assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).repeat(B, 1).contiguous()
    rep = lambda a: T(np.repeat(a[None], B, 0))
    args = [rep(p["prior"]), assoc, rep(p["obs_ptr"]), rep(p["obs_pose"]), rep(p["obs_uvr"]), rep(p["obs_oct"])]
    buf = torch.zeros(B, dtype=torch.int32).cuda()
    ctx.set_stats_buffer(buf)

    def run():
        poses, pts = rep(p["poses"]), rep(p["points"])
        return api.joint_optimization(ctx, g, cam, prm, P, F, poses, args[0], pts, *args[1:])
    for _ in range(2):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, int(buf[0].item()), len(p["obs_pose"])


for (P, F, L) in ((1, 1, 200), (2, 2, 400), (3, 2, 600), (4, 2, 800), (6, 3, 1200), (8, 4, 1500), (12, 4, 2000), (16, 6, 2500), (20, 8, 3000)):
    for B in (1, 2, 4, 8):
        m1, t1, nobs = timed(P, F, L, B, 1)
        m2, t2, _ = timed(P, F, L, B, 2)
        print("P%-2d F%d L%-4d obs %-6d B %d: persistent %.3f ms (%d trials), pipelined %.3f ms (%d trials)  -> %s" %
              (P, F, L, nobs, B, m1, t1, m2, t2, "pipelined" if m2 < m1 else "persistent"), flush=True)
