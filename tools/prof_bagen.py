"""Phase timing of k_ba_gen (general local BA).  Needs: make -C gmmloc_amd/csrc clean all EXTRA=-DGL_BAGEN_PROF"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
P, F, L = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 4, 1500))]
p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
idx, d2 = g.associate3d(T(p["points"]))
assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).contiguous()
poses = T(p["poses"][None])
api.joint_optimization(ctx, g, cam, prm, P, F, poses, T(p["prior"][None]), T(p["points"][None]), assoc, T(p["obs_ptr"][None]),
                       T(p["obs_pose"][None]), T(p["obs_uvr"][None]), T(p["obs_oct"][None]))
torch.cuda.synchronize()
c = poses[0, :2].reshape(-1).cpu().numpy()  # 14 doubles
names = ["P1 points", "P2 blocks", "(gap)", "solve", "P3+accept", "trials"]
tot = c[:5].sum()
for n, v in zip(names, c):
    print("%-10s %12.0f cycles %5.1f %%" % (n, v, 100 * v / tot if n != "trials" else 0))
print("per trial: %.0f cycles, observations %d" % (tot / max(c[5], 1), len(p["obs_pose"])))
print("  inside solve phase, per trial: diagonal terms / right-hand side %.0f  LDL^T solve %.0f  (rest = trial poses + barrier)" % (c[6] / max(c[5], 1), c[8] / max(c[5], 1)))
print("  inside the solve, per trial: panel columns out + diagonal block %.0f  trailing update (registers) %.0f  (rest = load, panel rows, backward substitution)" % (c[9] / max(c[5], 1), c[11] / max(c[5], 1)))
print("  barriers of workgroup 0, per trial (all phases, setup included): own workgroup %.0f  other workgroups + fences %.0f" % (c[7] / max(c[5], 1), c[10] / max(c[5], 1)))
