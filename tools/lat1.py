"""single-frame latency of gl_track_frames (bench frame 0 and a 700-point frame) + the bench step, quick A/B helper"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api, synth
mean, cov, cam, frames = bench.make_workload(1024)
prm = api.Params(); ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda k, fr: torch.from_numpy(np.stack([f[k] for f in fr])).cuda()
f700 = synth.synth_frame(mean, cov, synth.look_at_pose(np.array([-1.0, 0.5, 1.5]), np.array([1.5, 2.0, 1.4])), cam, 700, 20200901)
for name, fr in (("2000 pts", frames[:1]), ("700 pts", [f700])):
    p0, x0, ob, oc = T("pose_init", fr), T("Xw", fr), T("obs", fr), T("octave", fr)
    p, x = p0.clone(), x0.clone()
    lat = []
    for it in range(60):
        p.copy_(p0); x.copy_(x0); torch.cuda.synchronize()
        t = time.perf_counter(); gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, ob, oc, want_d2=False); torch.cuda.synchronize()
        lat.append(time.perf_counter() - t)
    print(os.environ.get("GMMLOC_HIP_LIB", "default")[-10:], name, "single frame %.4f ms (median of 50)" % (1e3 * float(np.median(lat[10:]))), flush=True)
p0, x0, ob, oc = T("pose_init", frames), T("Xw", frames), T("obs", frames), T("octave", frames)
def st():
    p, x = p0.clone(), x0.clone(); gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, ob, oc, want_d2=False)
st(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(8): st()
torch.cuda.synchronize(); print("1024-frame step %.3f ms" % ((time.perf_counter() - t) / 8 * 1e3))
