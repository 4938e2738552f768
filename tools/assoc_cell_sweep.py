"""The indexed association on the bench points against the cell size of the index (option assoc_cell, metres; 0 = automatic): ms per
launch of 8.19 M points, the size of the packed cell table and the candidates evaluated per point.   python tools/assoc_cell_sweep.py"""
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
ref = None
for cell in [0.0] + [float(v) for v in sys.argv[1:]]:
    ctx = gmmloc_amd.Context(0)
    ctx.set_option("assoc_cell", cell)
    ctx.set_option("assoc_pack_mb", 4096)
    g = gmmloc_amd.GMM(ctx, mean, cov, prm)
    info = g.index_info()
    # the association as gl_track_frames runs it (points whose best chi2 is above the gate stay unassociated: no sweep of the rest)
    def step():
        p, x = pose0.clone(), Xw0.clone()
        return gmmloc_amd.track_frames(ctx, g, cam, prm, p, x, obs, octv, want_d2=False)
    step()
    ctx.timing(True); ctx.timing_read(api.TIMER_ASSOC, reset=True)
    for _ in range(3): idx = step()
    torch.cuda.synchronize()
    ms, n = ctx.timing_read(api.TIMER_ASSOC); ctx.timing(False)
    ms = ms * 5 / max(n, 1)
    idx = idx[0] if isinstance(idx, (tuple, list)) else idx
    work = g.index_work(pts)
    cs = int(idx.sum().item())
    ref = cs if ref is None else ref
    print("assoc_cell %-6s cell %.4f m dims %s  %.3f ms per launch, %.2f candidates per point, index %s, same result %s" %
          (cell or "auto", info["cell"], info["dims"], ms / 5, work / pts.shape[0], info["bytes"], cs == ref), flush=True)
    g.close(); ctx.close()
