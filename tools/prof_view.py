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
v = ids.cpu().numpy()[:, :18].astype(np.float64)
v[:, :4] *= 16
v[:, 6:9] *= 16
v[:, 11:14] *= 16
print("knn (thread 0): pass-1 cycles %.0f, pass-2 cycles %.0f, tail %.0f; set bits of lane 0: %.0f, wave-0 loop iterations %.0f, with insertion %.0f" % tuple(v.mean(0)[11:17]))
print("mean cycles: project+cull %.0f  merge %.0f  sort %.0f  knn %.0f   candidates %.0f  accepted %.0f" % tuple(v.mean(0)[:6]))
print("merge: screen %.0f  exact %.0f  resolve %.0f cycles;  rounds %.1f  listed pairs %.0f" % tuple(v.mean(0)[6:11]), " order-free rounds %.1f" % v.mean(0)[17])

# wall time per call (normal build too): both block shapes
import time
for Bt in (1, 16, 64, 128, 512, 2048):
    pb = poses[:1].repeat(Bt, 1).contiguous() if Bt > B else poses[:Bt].contiguous()
    ub = uv[:1].repeat(Bt, 1, 1).contiguous() if Bt > B else uv[:Bt].contiguous()
    line = "B=%5d:" % Bt
    for th in ("256", "1024"):
        ctx.set_option("view_threads", int(th))
        for _ in range(3):
            g.search2d(cam, pb, ub, None, k=5, view_cap=64)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            g.search2d(cam, pb, ub, None, k=5, view_cap=64)
        torch.cuda.synchronize()
        line += "  T=%s %.3f ms/call" % (th, (time.perf_counter() - t0) / 10 * 1e3)
    print(line, flush=True)
