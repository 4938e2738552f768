"""Deterministic inputs of the randomised parity soak (tools/soak.py), by (map, round): shared by the soak on the
GPU box, by the CPU-side three-way classifier (tools/soak_classify.py) and by tests/test_gpu_soak_cases.py, so a
deviation found by the soak can be reproduced anywhere from its (map, round) label alone."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from gmmloc_amd import api, synth  # noqa: E402

MAPS = (("map_v1", ["V1_01_easy", "V1_02_medium", "V1_03_difficult"]), ("map_v2", ["V2_01_easy", "V2_02_medium"]))
TRI_KEYS = ("pose1", "uvr1", "depth1", "oct1", "pose2", "uvr2", "depth2", "oct2", "cand1", "n1", "cand2", "n2")


def load_map(mapname):
    d = np.load(os.path.join(ROOT, "tests", "golden", mapname + ".npz"))
    return d["mean"], d["cov"]


def load_gt():
    return np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))


def gen(mapname, r, mean, cov, gts, cam):
    """All sub-problems of soak round r on `mapname` (same random streams as the round-1 soak)."""
    seqs = dict(MAPS)[mapname]
    rng = np.random.default_rng(1000 * len(mapname) + r)
    gt = gts[seqs[r % len(seqs)]]
    while True:  # a key-frame pair with a baseline: the caller of createMapPoints skips the others (the sequences
        ia = int(rng.integers(0, gt.shape[0] - 40))  # start with a standing robot)
        ib = ia + int(rng.integers(3, 30))
        if np.linalg.norm(gt[ia][1:4] - gt[ib][1:4]) > 0.05:
            break
    p1, p2 = synth.gt_row_to_Tcw(gt[ia]), synth.gt_row_to_Tcw(gt[ib])
    N = int(rng.integers(50, 900))
    out = dict(ia=ia, ib=ib, p1=p1, p2=p2, N=N)
    # key-frame association chain
    f = synth.synth_frame(mean, cov, p1, cam, N, 7000 + r, mono_frac=0.0, outlier_frac=0.1)
    pts = f["Xw"] + rng.standard_normal((N, 3)) * 0.02
    octv = f["octave"].copy()
    octv[rng.uniform(size=N) < 0.05] = -1
    out["chain"] = dict(pose=p1, obs=f["obs"], pts=pts, octave=octv)
    # createMapPoints
    out["tri"] = synth.synth_tri_matches(mean, cov, p1, p2, cam, int(rng.integers(20, 500)), 9000 + r)
    # per-frame path (every 4th round: the oracle's joint_optimization is slow)
    out["track"] = None
    if r % 4 == 0:
        M = int(rng.integers(30, 700))
        out["track"] = synth.synth_frame(mean, cov, p1, cam, M, 11000 + r, outlier_frac=0.05)
    # optimizeCurrentPose (random size: every launch shape of the kernel over the rounds)
    Mp = int(rng.integers(5, 1300))
    fp = synth.synth_frame(mean, cov, p1, cam, Mp, 13000 + r, outlier_frac=0.08)
    fp["octave"][rng.uniform(size=Mp) < 0.1] = -1
    out["pose"] = fp
    return out


def gen_ba(nrounds, mean, cov, gts, cam):
    """Local-BA problems of the soak: random window sizes and forced workgroup counts (one shared stream)."""
    from tests.test_gpu_ba import make_ba_problem
    rng = np.random.default_rng(5)
    out = []
    for r in range(nrounds):
        P, F, L = int(rng.integers(1, 8)), int(rng.integers(0, 4)), int(rng.integers(20, 400))
        nb = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
        prior = bool(rng.integers(0, 2))
        out.append(dict(r=r, P=P, F=F, L=L, nb=nb, prior=prior,
                        problem=make_ba_problem(mean, cov, gts["V1_01_easy"], cam, P, F, L, 500 + r, prior)))
    return out


def track_oracle(orc, h, cam, f, perm=None, prior=False):
    """oracle associate3d + single-pose joint_optimization of frame f; `perm` re-orders the points first
    (the mathematics is order-free, the floating-point sums are not); prior: with the prior edge on the pose
    (gl_track_frames_anchored)."""
    keep = np.nonzero(f["octave"] >= 0)[0]
    if perm is not None:
        keep = keep[perm]
    Xw = f["Xw"][keep]
    idx, d2 = orc.associate3d(h, Xw)
    assoc = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    L = len(keep)
    poses, pts, dropped, erase, it = orc.joint_optimization(
        h, cam, 1, 0, f["pose_init"][None], np.full(1, 1 if prior else 0, np.uint8), Xw, assoc, np.arange(L + 1, dtype=np.int32),
        np.zeros(L, np.int32), f["obs"][keep], f["octave"][keep])
    final = np.where(dropped == 1, -1, assoc)
    return keep, poses[0], pts, final, idx, d2
