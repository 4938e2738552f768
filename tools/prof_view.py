"""Phase cycles of k_search2d (debug build: make -C gmmloc_amd/csrc clean all EXTRA=-DGL_VIEW_PROF)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
ctx = gmmloc_amd.Context(0); cam = api.Camera(); g = gmmloc_amd.GMM(ctx, mean, cov)
B, N = 64, 1000
poses = torch.from_numpy(np.stack([synth.gt_row_to_Tcw(gt[(11 + i * 13) % gt.shape[0]]) for i in range(B)])).cuda()
uv = torch.from_numpy(np.random.default_rng(0).uniform([0, 0], [752, 480], (B, N, 2))).cuda()
cand, ncand, ids, nv = g.search2d(cam, poses, uv, None, k=5, view_cap=64)
torch.cuda.synchronize()
v = ids.cpu().numpy()[:, :6].astype(np.float64)
v[:, :4] *= 16
print("mean cycles: project+cull %.0f  merge %.0f  sort %.0f  knn %.0f   candidates %.0f  accepted %.0f" % tuple(v.mean(0)))
