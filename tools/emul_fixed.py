"""The host replay of k_ba1_fast's arithmetic (tools/emul_ba1.cpp) WITH fixed observer key-frames against the oracle's
joint_optimization(P = 1, F fixed): the design check of the on-chip fixed-observer path before it went into the kernel.
    python tools/emul_fixed.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402
from tests.test_gpu_anchor import add_fixed, oracle_anchored  # noqa: E402
from tests.test_gpu_pose import make_frames, pose_err  # noqa: E402
from gmmloc_amd import api, synth  # noqa: E402
from tools import soak_cases as sc  # noqa: E402
from tools.emul_ba1 import load_emul  # noqa: E402


def run_case(orc, lib, mean, cov, gg, h, cam, prm, f, F, prior, variant=16):
    keep, p_ref, pts_ref, a_ref, fe_ref = oracle_anchored(orc, h, cam, f, bool(prior), F)
    L = len(keep)
    Xw = f["Xw"][keep]
    idx0, d20 = orc.associate3d(h, Xw)
    a0 = np.where(d20 <= 9.0, idx0, -1).astype(np.int32)
    obs, octv = f["obs"][keep], f["octave"][keep].astype(np.int32)
    deg = (gg["flags"] & 1).astype(np.int32)
    fl = (1 | ((octv & 7) << 8) | np.where(obs[:, 2] < 0, 0, 2) | np.where(a0 >= 0, 4 | np.where(deg[np.maximum(a0, 0)] != 0, 8, 0), 0)).astype(np.int32)
    nrm = lambda o: np.stack([(o[..., 0] - cam.cx) / cam.fx, (o[..., 1] - cam.cy) / cam.fy, np.where(o[..., 2] < 0, -1e30, (o[..., 2] - cam.cx) / cam.fx)], -1)
    obn = np.ascontiguousarray(nrm(obs))
    obn[:, 2] = (obs[:, 2] - cam.cx) / cam.fx  # (the frame's own edge takes its stereo flag from the flag word)
    axis = gg["axis"].reshape(-1, 3, 3)
    n = axis[:, :, 0]
    plane4 = np.concatenate([n, (n * mean).sum(1, keepdims=True)], 1).copy()
    Lc = gg["sqrt_info"].reshape(-1, 3, 3)
    LLt = Lc @ Lc.transpose(0, 2, 1)
    hgw = np.stack([LLt[:, 0, 0], LLt[:, 0, 1], LLt[:, 0, 2], LLt[:, 1, 1], LLt[:, 1, 2], LLt[:, 2, 2]], 1).copy()
    s2 = prm.sigma2_inv.astype(np.float64)
    sx, sy = (s2 * cam.fx * cam.fx).copy(), (s2 * cam.fy * cam.fy).copy()
    lm = float(prm.c().ba_lambda2)
    str_thresh = float(np.float32(prm.c().tri_str_thresh) * np.float32(prm.c().ba_lambda2))
    dm, ds = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    fRt = np.zeros((max(F, 1), 12))
    for j in range(F):
        fRt[j, :9] = synth.quat_to_R(f["fixed_pose"][j][:4]).ravel()
        fRt[j, 9:] = f["fixed_pose"][j][4:]
    fobn = np.ascontiguousarray(nrm(f["fixed_obs"][keep][:, :F]))
    foct = np.ascontiguousarray(f["fixed_oct"][keep][:, :F].astype(np.int32))
    ferase = np.zeros((L, max(F, 1)), np.uint8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.emul_set_fixed(F, P(fRt), P(fobn), P(foct), P(ferase))
    pose, pts = f["pose_init"].copy(), Xw.copy()
    trace, flo = np.zeros((128, 10)), np.zeros(L, np.int32)
    nt = lib.emul_track(L, P(pose), P(pts), P(obn), P(fl), P(a0), P(plane4), P(hgw), P(np.ascontiguousarray(mean)), P(sx), P(sy),
                        C.c_double(cam.bf / cam.fx), C.c_double(lm), C.c_double(str_thresh), C.c_double(dm), C.c_double(ds),
                        1 if prior else 0, int(variant), P(trace), 128, P(flo))
    lib.emul_set_fixed(0, None, None, None, None)
    dt, dr = pose_err(pose, p_ref)
    return dict(dt=dt, dr=dr, erase_equal=bool(np.array_equal(ferase[:, :F], fe_ref)), n_erase=int(fe_ref.sum()), trials=nt,
                dpts=float(np.abs(pts - pts_ref).max()))


if __name__ == "__main__":
    orc = oracle_lib.load()
    lib = load_emul()
    mean, cov = sc.load_map("map_v1")
    gts = sc.load_gt()
    cam, prm = api.Camera(), api.Params()
    h = orc.gmm_create(mean, cov)
    gg = orc.gmm_get(h)
    for M, F, seed in ((300, 2, 5), (700, 3, 6), (1200, 1, 7), (500, 4, 8)):
        for prior in (0, 1):
            frames = [add_fixed(f, cam, F, 900 + seed + i) for i, f in
                      enumerate(make_frames(mean, cov, gts["V1_01_easy"], cam, 3, M, 40 + seed, outlier_frac=0.05))]
            frames[2]["octave"][::5] = -1
            for i, f in enumerate(frames):
                r = run_case(orc, lib, mean, cov, gg, h, cam, prm, f, F, prior)
                print("M %4d F %d prior %d frame %d: pose %.2e m %.2e rad, fixed-erase equal %s (%d), points %.2e, %d trials"
                      % (M, F, prior, i, r["dt"], r["dr"], r["erase_equal"], r["n_erase"], r["dpts"], r["trials"]))
