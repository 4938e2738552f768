"""B copies of one gl_joint_optimization window, a few calls (for rocprofv3 --kernel-trace --stats and for timing the modes on a batch):
   python tools/ba_batch_prof.py P F L B [bagen_mode] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem
d = np.load(os.path.join(ROOT, "tests/golden/map_v1.npz")); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests/golden/gt_sync.npz"))["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P, F, L, B = [int(x) for x in sys.argv[1:5]]
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
ctx.set_option("bagen_mode", mode)
p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
idx, d2 = g.associate3d(T(p["points"]))
assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).repeat(B, 1).contiguous()
rep = lambda a: T(np.repeat(a[None], B, 0))
args = [rep(p["prior"]), assoc, rep(p["obs_ptr"]), rep(p["obs_pose"]), rep(p["obs_uvr"]), rep(p["obs_oct"])]
stats = torch.zeros(B, dtype=torch.int32).cuda()
ctx.set_stats_buffer(stats)
def run():
    poses, pts = rep(p["poses"]), rep(p["points"])
    api.joint_optimization(ctx, g, cam, prm, P, F, poses, args[0], pts, *args[1:])
    return poses
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    poses = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print("P%d F%d L%d B %d mode %d: %.3f ms per launch, %.4f ms per window, %d trials, pose fingerprint %s" %
      (P, F, L, B, mode, ms, ms / B, int(stats[0].item()), poses[0, 0].cpu().numpy().tobytes().hex()[:32]))
