"""Where a pass of k_optimize_current_pose goes (frame-at-a-time shapes, edges on chip).  Needs the diagnosis build:
   bash tools/build_variant.sh poseprof "-DGL_POSE_PROF" gl_refine_pose.hip
   GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py [M] [B]
Thread 0 of every frame returns its clock64() deltas per phase in the frame's pose, the number of passes in ninlier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gmmloc_amd
from gmmloc_amd import api, synth

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = gmmloc_amd.Context(0)
cam, prm = api.Camera(), api.Params()
mean, cov = synth.synth_gmm(512, 3)
fr = [synth.synth_frame(mean, cov, synth.look_at_pose([0.0, 0.0, 1.5], [3.0, 1.0, 1.2]), cam, M, 5 + b) for b in range(B)]
T = lambda k: torch.from_numpy(np.stack([f[k] for f in fr])).cuda()
pose, Xw, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
for _ in range(3):
    p2 = pose.clone()
    outl, nin = gmmloc_amd.optimize_current_pose(ctx, cam, prm, p2, Xw, obs, octv)
torch.cuda.synchronize()
c = p2.cpu().numpy()
n = nin.cpu().numpy().astype(np.float64)
names = ["entry barrier", "edges", "reduce-scatter", "barriers + blocks", "6x6 solve", "pose update"]
tot = c[:, 6].mean()
print("M = %d, B = %d: %.0f passes per frame, %.0f cycles per frame = %.0f per pass" % (M, B, n.mean(), tot, tot / n.mean()))
for i, nm in enumerate(names):
    print("  %-18s %8.0f cycles per pass  %5.1f %%" % (nm, c[:, i].mean() / n.mean(), 100 * c[:, i].mean() / tot))
print("  %-18s %8.0f cycles per pass  %5.1f %%" % ("everything else", (tot - c[:, :6].sum(1).mean()) / n.mean(), 100 * (tot - c[:, :6].sum(1).mean()) / tot))
