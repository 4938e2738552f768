"""The bench step on TWO contexts (two streams, two scratch blocks) taking the steps in turn, against one context: does the tail of
step k's refine overlap the association / set-up kernels of step k + 1?   python tools/two_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api
B = 4096
mean, cov, cam, frames = bench.make_workload(B)
prm = api.Params()
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose0, Xw0, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
ctxs = [gmmloc_amd.Context(0) for _ in range(2)]
gmms = [gmmloc_amd.GMM(c, mean, cov, prm) for c in ctxs]
bufs = [(pose0.clone(), Xw0.clone()) for _ in range(2)]


def run(nctx, steps):
    for i in range(steps):
        k = i % nctx
        with torch.cuda.stream(ctxs[k].stream):
            bufs[k][0].copy_(pose0)
            bufs[k][1].copy_(Xw0)
            gmmloc_amd.track_frames(ctxs[k], gmms[k], cam, prm, bufs[k][0], bufs[k][1], obs, octv, want_d2=False)


for nctx in (1, 2, 1, 2):
    run(nctx, 4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(nctx, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%d context(s): %.3f ms per step, %.1f k frames/s" % (nctx, dt * 1e3, B / dt / 1e3), flush=True)
same = all(torch.equal(bufs[0][i], bufs[1][i]) for i in range(2))
print("results of the two contexts bit-identical:", same)
