"""Drives tools/emul_ba1.cpp (host emulation of k_ba1_fast's arithmetic) against the oracle's Levenberg trace on one
soak frame.  Debugging aid: python tools/emul_ba1.py map_v1 63072 [variant bitmask ...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import soak_cases as sc  # noqa: E402
from tests import oracle_lib  # noqa: E402
from gmmloc_amd import api  # noqa: E402


def pose_err(a, b):
    from gmmloc_amd import synth
    Ra, Rb = synth.quat_to_R(a[:4]), synth.quat_to_R(b[:4])
    dR = Ra @ Rb.T
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    ca, cb = -Ra.T @ a[4:], -Rb.T @ b[4:]
    return float(np.linalg.norm(ca - cb)), float(ang)


def load_emul():
    so = os.path.join(ROOT, "build_tmp", "libemul_ba1.so")
    src = os.path.join(ROOT, "tools", "emul_ba1.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    return C.CDLL(so)


def oracle_trace(orc, h, cam, f, prior=True):
    """runs the oracle in a child process with OG_TRACE=1 and parses the trace"""
    raise NotImplementedError


def run(mapname, r, variants, prior=True):
    orc = oracle_lib.load()
    mean, cov = sc.load_map(mapname)
    gts = sc.load_gt()
    cam, prm = api.Camera(), api.Params()
    h = orc.gmm_create(mean, cov)
    gg = orc.gmm_get(h)
    f = sc.gen(mapname, r, mean, cov, gts, cam)["track"]
    keep, p_ref, pts_ref, a_ref, idx0, d20 = sc.track_oracle(orc, h, cam, f, prior=prior)
    L = len(keep)
    a0 = np.where(d20 <= 9.0, idx0, -1).astype(np.int32)
    obs, octv = f["obs"][keep], f["octave"][keep].astype(np.int32)
    deg = (gg["flags"] & 1).astype(np.int32)
    fl = (1 | ((octv & 7) << 8) | np.where(obs[:, 2] < 0, 0, 2) | np.where(a0 >= 0, 4 | np.where(deg[np.maximum(a0, 0)] != 0, 8, 0), 0)).astype(np.int32)
    obn = np.stack([(obs[:, 0] - cam.cx) / cam.fx, (obs[:, 1] - cam.cy) / cam.fy, (obs[:, 2] - cam.cx) / cam.fx], 1).copy()
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
    lib = load_emul()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {}
    for v in variants:
        pose = f["pose_init"].copy()
        pts = f["Xw"][keep].copy()
        trace = np.zeros((128, 10))
        flo = np.zeros(L, np.int32)
        nt = lib.emul_track(L, P(pose), P(pts), P(obn), P(fl), P(a0), P(plane4), P(hgw), P(np.ascontiguousarray(mean)), P(sx), P(sy),
                            C.c_double(cam.bf / cam.fx), C.c_double(lm), C.c_double(str_thresh), C.c_double(dm), C.c_double(ds),
                            1 if prior else 0, int(v), P(trace), 128, P(flo))
        out[v] = dict(pose=pose, pts=pts, trace=trace[:nt], err=pose_err(pose, p_ref))
    return out, p_ref


if __name__ == "__main__":
    mapname, r = sys.argv[1], int(sys.argv[2])
    variants = [int(x) for x in sys.argv[3:]] or [0]
    out, p_ref = run(mapname, r, variants)
    for v, o in out.items():
        print("variant", v, "trials", len(o["trace"]), "pose err (m, rad) vs oracle", o["err"])
        if os.environ.get("EMUL_TRACE"):
            for i, t in enumerate(o["trace"]):
                print(i, " ".join("%.17g" % x for x in t[:4]), "|", " ".join("%.6g" % x for x in t[4:]))
