"""optimizeCurrentPose, large batches: one wave per frame against the default launch rule (a wave per group with the edges
on chip where that fills the CU, gl_refine_pose.hip) and against the same shapes reading the edges from global memory
(pose_regs = 0); full frames of M edges and ragged ones (M_f ~ U{150..M}).   python tools/pose_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)
dev = torch.device("cuda", 0)
for M in (300, 600, 1000, 1200, 1600, 2000):
    fr = [synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(gt[(100 + 17 * i) % gt.shape[0]]), cam, M, 50 + i) for i in range(64)]
    for ragged in (False, True):
        rng = np.random.default_rng(1)
        for B in (2048, 4096):
            fs = [fr[i % 64] for i in range(B)]
            T = lambda k: torch.from_numpy(np.stack([f[k] for f in fs])).to(dev)
            p0, x0, o, oc = T("pose_init"), T("Xw"), T("obs"), T("octave").clone()
            if ragged:
                for b in range(B):
                    oc[b, int(rng.integers(150, M + 1)):] = -1
            out = []
            with torch.cuda.stream(ctx.stream):
                for nw, regs in ((1, 1), (0, 1), (0, 0)):
                    ctx.set_option("pose_waves", nw)
                    ctx.set_option("pose_regs", regs)
                    for _ in range(2):
                        api.optimize_current_pose(ctx, cam, prm, p0.clone(), x0, o, oc)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(5):
                        api.optimize_current_pose(ctx, cam, prm, p0.clone(), x0, o, oc)
                    torch.cuda.synchronize()
                    out.append((time.perf_counter() - t0) / 5 * 1e3)
            print("M %4d %s B %4d: one wave per frame %.3f ms, default %.3f ms, default with pose_regs = 0 %.3f ms" % (M, "ragged" if ragged else "full  ", B, out[0], out[1], out[2]), flush=True)
