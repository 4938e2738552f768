"""Phase timing of k_ba1_fast.  Needs the debug build:
   make -C gmmloc_amd/csrc clean all EXTRA=-DGL_BA_PROF
(thread 0 of block 0 accumulates clock64() deltas and returns them in pose[0])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gmmloc_amd
from gmmloc_amd import api
import bench
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mean, cov, cam, frames = bench.make_workload(NF)
ctx = gmmloc_amd.Context(0); prm = api.Params(); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose, Xw, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
p2, x2 = pose.clone(), Xw.clone()
gmmloc_amd.track_frames(ctx, g, cam, prm, p2, x2, obs, octv)
torch.cuda.synchronize()
c = p2[0].cpu().numpy()
names = ["passA", "reduceA", "solve+bcast", "passB", "reduceB", "accept", "trials"]
tot = c[:6].sum()
for n, v in zip(names, c):
    print("%-12s %12.0f cycles  %5.1f %%" % (n, v, 100 * v / tot if n != "trials" else 0))
print("per trial: %.0f cycles" % (tot / max(c[6], 1)))
