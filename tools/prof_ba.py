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

# slot markers: [trial < 16][wave][pass][slot 0..3 start, pass end]
flat = x2[0].cpu().numpy().reshape(-1)
sm = flat[2048:2048 + 16 * 8 * 2 * 5].reshape(16, 8, 2, 5)
for ps, nm in ((0, "pass A"), (1, "pass B")):
    d = np.diff(sm[6:16, :, ps, :], axis=2)  # slot durations [trial][wave][4]
    print("%s slot durations (cycles), mean over trials 6..15; rows = waves 0..7, columns = slots 0..3 (the last column runs to the pass end)" % nm)
    for w_ in range(8):
        dd = d[:, w_, :]
        ok = (np.abs(dd) < 1e8).all(0)  # (a wave with fewer slots leaves the later markers unwritten)
        tot = sm[6:16, w_, ps, 4] - sm[6:16, w_, ps, 0]
        print("   wave %d: %s   pass %6.0f" % (w_, " ".join(("%6.0f" % v) if o else "     -" for v, o in zip(dd.mean(0), ok)), tot.mean()))
    st = sm[6:16, :, ps, 0]
    print("   first-slot start skew over waves: mean %.0f" % (st.max(1) - st.min(1)).mean())
# inside a slot: [trial 6..15][wave][pass][slot][loads arrived, step half done (B), slot done], against the slot's start
q = flat[2048 + 1280:2048 + 1280 + 10 * 8 * 2 * 4 * 3].reshape(10, 8, 2, 4, 3)
st = sm[6:16, :, :, :4]
for ps, nm in ((0, "pass A"), (1, "pass B")):
    print("%s: cycles from the slot's start until its global loads have arrived | (B: step half done) | slot done; mean over trials and slots" % nm)
    for w_ in range(8):
        d = q[:, w_, ps, :, :] - st[:, w_, ps, :, None]
        print("   wave %d: loads %6.0f   step %6.0f   done %6.0f      per slot loads: %s" % (w_, d[:, :, 0].mean(), d[:, :, 1].mean() if ps else 0, d[:, :, 2].mean(),
              " ".join("%5.0f" % v for v in d[:, :, 0].mean(0))))

if NF > 1024:
    k = np.concatenate([p2[1024].cpu().numpy(), x2[1024].cpu().numpy().reshape(-1)[:6]])
    names3 = ["kernel entry -> set-up done (points into LDS, flags, tables)", "optimize(5) #1", "gate 1", "optimize(5) #2", "gate 2", "optimize(40)", "-> outputs ready"]
    print("frame 1024 (middle of the launch), stations of the kernel in cycles; %d trials" % k[9])
    for i, n in enumerate(names3):
        print("   %-64s %9.0f" % (n, k[i + 1] - k[i] if i < 6 else k[8] - k[6]))
    print("   %-64s %9.0f" % ("whole kernel (entry -> outputs ready)", k[8] - k[0]))
    print("   %-64s %9.0f" % ("of which the three lambda initialisations", k[12]))

if NF > 1024:
    P_ = p2.cpu().numpy()
    sel = np.array([f for f in range(NF) if f not in (0, 1024)])
    t0, t1, hw, tr = P_[sel, 0], P_[sel, 1], P_[sel, 2].astype(np.int64), P_[sel, 3]
    cu = ((hw >> 24) & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15)  # xcc, se, sh, cu
    T0, T1 = t0.min(), t1.max()
    ids = np.unique(cu)
    busy, gaps, nper = [], [], []
    for c_ in ids:
        m = cu == c_
        o = np.argsort(t0[m])
        a, b = t0[m][o], t1[m][o]
        busy.append((b - a).sum())
        nper.append(m.sum())
        gaps += list(a[1:] - b[:-1])
    busy = np.array(busy)
    print("workgroup timeline (wall_clock64 = 100 MHz ticks): launch span %.0f us on %d distinct CUs, %.1f frames per CU (min %d max %d)" % ((T1 - T0) / 100, len(ids), np.mean(nper), min(nper), max(nper)))
    print("   mean workgroup duration %.1f us (%.2f us per trial), CU busy share of the span: mean %.3f min %.3f" % ((t1 - t0).mean() / 100, ((t1 - t0) / tr).mean() / 100, (busy / (T1 - T0)).mean(), (busy / (T1 - T0)).min()))
    g = np.array(gaps)
    print("   gap between consecutive workgroups of a CU: median %.1f us, mean %.1f us; first start after launch start: mean %.1f us; last end before span end: mean %.1f us" %
          (np.median(g) / 100, g.mean() / 100, np.mean([t0[cu == c_].min() - T0 for c_ in ids]) / 100, np.mean([T1 - t1[cu == c_].max() for c_ in ids]) / 100))
