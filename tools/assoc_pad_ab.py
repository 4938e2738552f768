"""The indexed association's cooperative gather from the 96-byte records (assoc_rec_pad = 0) against their one-per-128-byte-line copy
(1, the default), and with the lists of more than three candidates walked by their lane alone (assoc_coop_long = 0) against gathered
cooperatively like the short ones (1, the default), on the bench points: ms per launch of 8.19 M points as gl_track_frames runs it (no sweep of the unresolved points),
interleaved, and the results of gl_associate3d (indices and chi2) compared bit for bit.   python tools/assoc_pad_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gmmloc_amd, bench
from gmmloc_amd import api
B = 4096
mean, cov, cam, frames = bench.make_workload(B)
pts = torch.from_numpy(np.concatenate([f["Xw"] for f in frames])).cuda()
T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
pose0, Xw0, obs, octv = T("pose_init"), T("Xw"), T("obs"), T("octave")
prm = api.Params()
ctxs, gs = {}, {}
# (assoc_rec_pad, assoc_coop_long, assoc_coop_bal); "old" = the kernel of rounds 3 - 5h, the last = the defaults
VARIANTS = {"old": (0, 0, 0), "pad": (1, 0, 0), "long": (0, 1, 0), "pad+long": (1, 1, 0), "pad+long+bal": (1, 1, 1)}
if os.environ.get("AB_ONLY"):  # (variant libraries)
    VARIANTS = {n: VARIANTS[n] for n in ("old", "pad+long", "pad+long+bal")}
DEFAULT = "pad+long+bal"
for name, (pad, lng, bal) in VARIANTS.items():
    ctxs[name] = gmmloc_amd.Context(0)
    ctxs[name].set_option("assoc_rec_pad", pad)
    ctxs[name].set_option("assoc_coop_long", lng)
    ctxs[name].set_option("assoc_coop_bal", bal)
    gs[name] = gmmloc_amd.GMM(ctxs[name], mean, cov, prm)
res = {}
for name in VARIANTS:
    idx, d2 = gs[name].associate3d(pts, api.ASSOC_BRUTE, want_d2=True)
    res[name] = (idx.clone(), d2.clone())
same = all(bool((res["old"][0] == res[n][0]).all().item()) and bool((res["old"][1].view(torch.int64) == res[n][1].view(torch.int64)).all().item())
           for n in VARIANTS)
print("gl_associate3d, %d points: indices and chi2 bit-equal between the variants: %s (%d associated)" %
      (pts.shape[0], same, int((res["old"][0] >= 0).sum().item())), flush=True)
def run(name, reps):
    ctx, g = ctxs[name], gs[name]
    def step():
        p, x = pose0.clone(), Xw0.clone()
        return gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, obs, octv, want_d2=False)
    step()
    ctx.timing(True); ctx.timing_read(api.TIMER_ASSOC, reset=True)
    for _ in range(reps): step()
    torch.cuda.synchronize()
    ms, n = ctx.timing_read(api.TIMER_ASSOC); ctx.timing(False)
    return ms / max(n, 1)
for rnd in range(3):
    t = {n: run(n, 4) for n in VARIANTS}
    print("round %d, ms per launch of %d points: " % (rnd, pts.shape[0]) + ", ".join("%s %.4f" % (n, t[n]) for n in VARIANTS) +
          " (default against old: %.1f %%)" % (100 * (t[DEFAULT] / t["old"] - 1)), flush=True)
