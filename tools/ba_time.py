"""Single-problem and batch time of gl_joint_optimization at the library's own shape choice (ms):
   python tools/ba_time.py            P8/F4/L1500, P12/F4/L2000, P20/F8/L3000 single; 64 and 256 problems of P8/F4/L1500"""
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
if os.environ.get("BAGEN_MODE"):
    ctx.set_option("bagen_mode", int(os.environ["BAGEN_MODE"]))  # 1 persistent kernel, 2 pipelined shape
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def timed(P, F, L, B, reps=5):
    p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
    idx, d2 = g.associate3d(T(p["points"]))
    assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).repeat(B, 1).contiguous()
    rep = lambda a: T(np.repeat(a[None], B, 0))
    args = [rep(p["prior"]), assoc, rep(p["obs_ptr"]), rep(p["obs_pose"]), rep(p["obs_uvr"]), rep(p["obs_oct"])]

    stats = torch.zeros(B, dtype=torch.int32).cuda()
    ctx.set_stats_buffer(stats)

    def run():
        poses, pts = rep(p["poses"]), rep(p["points"])
        r = api.joint_optimization(ctx, g, cam, prm, P, F, poses, args[0], pts, *args[1:])
        return poses, r
    for _ in range(2):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        poses, r = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    tr = int(stats[0].item())
    print("P%d F%d L%d obs %d x %d problem(s): %.3f ms per launch, %.3f ms per problem, iters %s, %d Levenberg trials (%.1f us per trial)" %
          (P, F, L, len(p["obs_pose"]), B, ms, ms / B, r[2][:3].cpu().numpy(), tr, ms * 1e3 / max(tr, 1)), flush=True)


for P, F, L in ((8, 4, 1500), (12, 4, 2000), (20, 8, 3000)):
    timed(P, F, L, 1)
timed(8, 4, 1500, 64, 3)
timed(8, 4, 1500, 256, 2)
