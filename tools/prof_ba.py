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
SHAPE = int(sys.argv[2]) if len(sys.argv) > 2 else -1
PRIOR = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 2:
    pass
mean, cov, cam, frames = bench.make_workload(NF)
ctx = gmmloc_amd.Context(0); prm = api.Params(); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
ctx.set_option("ba_shape", SHAPE)
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose, Xw, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
p2, x2 = pose.clone(), Xw.clone()
if PRIOR:
    gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, p2, x2, obs, octv, prior=torch.ones(NF, dtype=torch.uint8).cuda())
else:
    gmmloc_amd.track_frames(ctx, g, cam, prm, p2, x2, obs, octv)
torch.cuda.synchronize()
c = p2[0].cpu().numpy()
names = ["passA", "reduceA", "solve+bcast", "passB", "reduceB", "accept", "trials"]
tot = c[:6].sum()
for n, v in zip(names, c):
    print("%-12s %12.0f cycles  %5.1f %%" % (n, v, 100 * v / tot if n != "trials" else 0))
print("per trial: %.0f cycles" % (tot / max(c[6], 1)))
w = x2[0].cpu().numpy().reshape(-1)[:64 * 8 * 4].reshape(64, 8, 4)
w = w[4:40]  # steady-state trials of the first phases
names2 = ["pass A end", "after reduce A", "pass B start", "pass B end"]
for k, n in enumerate(names2):
    sk = w[:, :, k].max(1) - w[:, :, k].min(1)
    print("%-16s skew over the 8 waves: mean %6.0f  max %6.0f cycles" % (n, sk.mean(), sk.max()))
dA = (w[:, :, 1].min(1) - w[:, :, 0].max(1))
print("last wave done with pass A -> first wave past reduce A: mean %.0f cycles" % dA.mean())
dS = (w[:, :, 2].min(1) - w[:, :, 1].max(1))
print("reduce A done -> pass B start (solve + broadcast): mean %.0f cycles" % dS.mean())
dur = w[:, :, 0] - np.roll(w[:, :, 3], 1, axis=0)
print("pass A duration per wave (from the previous pass-B end, incl. reduce B + accept): mean per wave", np.round(dur[1:].mean(0)))
durB = w[:, :, 3] - w[:, :, 2]
print("pass B duration per wave: mean per wave", np.round(durB.mean(0)))
