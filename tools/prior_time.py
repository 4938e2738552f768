import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api
B = 1024
mean, cov, cam, frames = bench.make_workload(B)
prm = api.Params(); ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose0, Xw0, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
pr = torch.ones(B, dtype=torch.uint8).cuda()
for prior in (False, True):
    def st():
        p, x = pose0.clone(), Xw0.clone()
        if prior: gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, p, x, obs, octv, prior=pr, want_d2=False)
        else: gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, obs, octv, want_d2=False)
    st(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): st()
    torch.cuda.synchronize(); print(os.environ.get("GMMLOC_HIP_LIB", "default")[-12:], "prior" if prior else "plain", "%.3f ms per 1024 frames" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
