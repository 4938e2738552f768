"""Trajectory output and evaluation on the far side of the hot path:
the TUM writer of Map::summarize (map.cpp:162-188, through the C-ABI) and the numbers of
scripts/evo_euroc.py:28-57 -- timestamp association (evo sync.associate_trajectories, max_diff 0.01 s),
Umeyama alignment with scale (evo trajectory.align_trajectory(correct_scale=True)) and the
translation-part APE statistics.  numpy only; evo itself is not needed."""
import ctypes as C

import numpy as np

from . import _lib


def write_tum(path, stamps, pose_wc):
    """stamps (N,), pose_wc (N,7) = (qx qy qz qw tx ty tz) of T_wc."""
    lib = _lib.load()
    stamps = np.ascontiguousarray(stamps, np.float64)
    pose_wc = np.ascontiguousarray(pose_wc, np.float64)
    rc = lib.gl_write_tum_trajectory(str(path).encode(), stamps.ctypes.data, pose_wc.ctypes.data, len(stamps))
    if rc != 0:
        raise RuntimeError(lib.gl_last_error_string().decode())


def read_tum(path):
    """-> stamps (N,), xyz (N,3), quat_xyzw (N,4)."""
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def associate(stamps_ref, stamps_est, max_diff=0.01):
    """Greedy nearest-timestamp matching like evo's sync.associate_trajectories: for every stamp of the
    shorter trajectory the closest one of the longer, kept when |dt| <= max_diff.  -> (idx_ref, idx_est)."""
    stamps_ref, stamps_est = np.asarray(stamps_ref), np.asarray(stamps_est)
    swap = len(stamps_est) > len(stamps_ref)
    a, b = (stamps_ref, stamps_est) if swap else (stamps_est, stamps_ref)  # a: shorter
    order = np.argsort(b)
    pos = np.clip(np.searchsorted(b[order], a), 1, len(b) - 1)
    left, right = order[pos - 1], order[pos]
    j = np.where(np.abs(b[left] - a) <= np.abs(b[right] - a), left, right)
    ok = np.abs(b[j] - a) <= max_diff
    ia, ib = np.nonzero(ok)[0], j[ok]
    return (ia, ib) if swap else (ib, ia)


def umeyama(src, dst, with_scale=True):
    """Least-squares similarity dst ~ s R src + t (Umeyama 1991), as evo's geometry.umeyama_alignment."""
    src, dst = np.asarray(src, float), np.asarray(dst, float)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / xs.var(0).sum()) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ape_translation(stamps_gt, xyz_gt, stamps_est, xyz_est, max_diff=0.01, correct_scale=True):
    """evo_euroc.py:40-54 -> dict(mean, rmse, median, std, min, max, n, scale)."""
    ig, ie = associate(stamps_gt, stamps_est, max_diff)
    g, e = np.asarray(xyz_gt)[ig], np.asarray(xyz_est)[ie]
    if len(ig) < 3:  # nothing to align
        nan = float("nan")
        return {"mean": nan, "rmse": nan, "median": nan, "std": nan, "min": nan, "max": nan, "n": int(len(ig)), "scale": nan}
    s, R, t = umeyama(e, g, correct_scale)
    err = np.linalg.norm(g - (s * (R @ e.T).T + t), axis=1)
    return {"mean": float(err.mean()), "rmse": float(np.sqrt((err ** 2).mean())), "median": float(np.median(err)),
            "std": float(err.std()), "min": float(err.min()), "max": float(err.max()), "n": int(len(err)), "scale": s}
