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
for B in [1, 2, 4, 8, 16, 32, 64, 256]:
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

# ---- B = 1 latency of the other per-frame entry points on a real-map frame (v1.gmm, M = 1000) ----------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from gmmloc_amd import synth
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean1, cov1 = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
g1 = gmmloc_amd.GMM(ctx, mean1, cov1, prm)
fr = [synth.synth_frame(mean1, cov1, synth.gt_row_to_Tcw(gt[(100 + 17 * i) % gt.shape[0]]), cam, 1000, 50 + i) for i in range(64)]
fr = fr * 64
T1 = lambda k: torch.from_numpy(np.stack([f[k] for f in fr])).to(dev)
p0, x0, o, oc = T1("pose_init"), T1("Xw"), T1("obs"), T1("octave")
uv = o[:, :, :2].contiguous()


def timed(name, fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("%-58s %.3f ms/call" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)


with torch.cuda.stream(ctx.stream):
    for B in (1, 16, 64, 128, 1024, 4096):
        pb, xb, ob, ocb, uvb = p0[:B].contiguous(), x0[:B].contiguous(), o[:B].contiguous(), oc[:B].contiguous(), uv[:B].contiguous()
        for env in ("1", "4", "8"):
            ctx.set_option("pose_waves", int(env))
            timed("optimizeCurrentPose B=%d M=1000 (%s waves per frame)" % (B, env),
                  lambda: api.optimize_current_pose(ctx, cam, prm, pb.clone(), xb, ob, ocb))
        ctx.set_option("pose_waves", 0)
        ctx.set_option("pose_regs", 0)
        timed("optimizeCurrentPose B=%d M=1000 (edges re-read from global memory)" % B,
              lambda: api.optimize_current_pose(ctx, cam, prm, pb.clone(), xb, ob, ocb))
        ctx.set_option("pose_regs", 1)
        timed("track_frames B=%d M=1000 K=%d" % (B, mean1.shape[0]),
              lambda: gmmloc_amd.track_frames(ctx, g1, cam, prm, pb.clone(), xb.clone(), ob, ocb, want_d2=False))
        timed("search2d B=%d N=1000" % B, lambda: g1.search2d(cam, pb, uvb, None, k=5))
