"""One gl_joint_optimization window three times (for rocprofv3 --kernel-trace --stats): python tools/ba_one.py P F L [bagen_mode]"""
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
P, F, L = [int(x) for x in sys.argv[1:4]]
if len(sys.argv) > 4:
    ctx.set_option("bagen_mode", int(sys.argv[4]))
p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
idx, d2 = g.associate3d(T(p["points"]))
assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).contiguous()
for _ in range(3):
    api.joint_optimization(ctx, g, cam, prm, P, F, T(p["poses"][None]), T(p["prior"][None]), T(p["points"][None]), assoc, T(p["obs_ptr"][None]), T(p["obs_pose"][None]), T(p["obs_uvr"][None]), T(p["obs_oct"][None]))
torch.cuda.synchronize()
