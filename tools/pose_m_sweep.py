"""optimizeCurrentPose, ONE frame: latency against the number of edge SLOTS M (the stride of the problem), with ~35 % of the slots
holding an edge (a tracked frame: 1 200 features, ~400 with a map point) and with the same edges COMPACTED to the front of a smaller
stride - what gl_track_frame_chain's compaction buys (round 6).   python tools/pose_m_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz")); mean1, cov1 = d["mean"], d["cov"]
gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
cam, prm = api.Camera(), api.Params(); ctx = gmmloc_amd.Context(0)
f = synth.synth_frame(mean1, cov1, synth.gt_row_to_Tcw(gt[100]), cam, 1280, 50)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).cuda()
rng = np.random.default_rng(1)


def timed(fn, reps=40):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


with torch.cuda.stream(ctx.stream):
    for M, frac in ((1200, 1.0), (1200, 0.35), (1024, 1.0), (1024, 0.41), (768, 0.55), (512, 0.82), (448, 0.94), (256, 1.0)):
        n_act = int(round(M * frac))
        oc = np.full(M, -1, np.int32)
        act = np.sort(rng.choice(M, n_act, replace=False)) if frac < 1.0 and M == 1200 else np.arange(n_act)
        oc[act] = f["octave"][:n_act]
        Xw, ob = np.zeros((M, 3)), np.zeros((M, 3))
        Xw[act], ob[act] = f["Xw"][:n_act], f["obs"][:n_act]
        p0, xw, o, o_c = T(f["pose_init"]), T(Xw), T(ob), T(oc)
        ms = timed(lambda: api.optimize_current_pose(ctx, cam, prm, p0.clone(), xw, o, o_c))
        print("M = %4d slots, %4d edges: %.3f ms per call" % (M, n_act, ms), flush=True)
