"""k_assoc_cells on the bench points (8.19 M points x 4 096 Gaussians), ms per launch; GMMLOC_ASSOC_PACK_MB=0: CSR path (no packed cell table)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api
B = 4096
mean, cov, cam, frames = bench.make_workload(B)
prm = api.Params(); ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
pts = torch.from_numpy(np.concatenate([f["Xw"] for f in frames])).cuda()
for want in (False, True):
    g.associate3d(pts, api.ASSOC_BRUTE, want_d2=want)
    ctx.timing(True); ctx.timing_read(api.TIMER_ASSOC, reset=True)
    for _ in range(5): idx, d2 = g.associate3d(pts, api.ASSOC_BRUTE, want_d2=want)
    torch.cuda.synchronize()
    ms, n = ctx.timing_read(api.TIMER_ASSOC); ctx.timing(False)
    print("pack" if not os.environ.get("GMMLOC_ASSOC_PACK_MB") == "0" else "csr ", "want_d2", want, "%.3f ms per call (%d timed launches)" % (ms / 5, n), "checksum", int(idx.sum().item()))
