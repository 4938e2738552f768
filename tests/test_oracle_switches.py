"""CPU: the oracle's DECLARED deviations from Eigen / g2o internals do not decide its results.

The reference's arithmetic sits in un-vendored, unpinned Eigen and g2o: parity with the real binary stays
unpinned.  Four choices of the restatement are known to differ from what those libraries run internally - un-pivoted
LDL^T (g2o::LinearSolverDense = pivoted Eigen::LDLT, LinearSolverEigen = SimplicialLDLT with a fill-reducing ordering),
cyclic-Jacobi eigen-decomposition (Eigen: tridiagonalisation + shifted QR/QL), no `!isfinite(lambda)` break in the
Levenberg retry loop (newer g2o has one), insertion order where optimizeTriangulationVec iterates an unordered_set.
oracle/og_math.hpp carries an alternative for each behind a switch; here every golden problem and 200 soak problems
(tools/soak_cases.py: the inputs of the GPU soak) are run under every alternative: decisions must be EQUAL and poses /
points within 1e-8 of the default build, at least 95 % of them within 1e-9 (measured: the pose refine moves by at most
3.8e-9 m on 8 of 203 frames - the smallest ones, a handful of edges - under either alternative factorisation and not at all
under the lambda break; two orders of magnitude inside the north-star 1e-6).  This does not pin g2o - it shows the answers
do not hang on what could not be."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.test_oracle_golden import Cam, g
from tests.test_gpu_pose import pose_err
from tools import soak_cases

SW_LDLT, SW_EIG, SW_LAMBDA, SW_TRI = 0, 1, 2, 3
TOL, TOL_MOST = 1e-8, 1e-9


@pytest.fixture
def switch(oracle):
    def set_(which, value):
        oracle.lib.orc_set_switch(C.c_int(which), C.c_int(value))
    yield set_
    for w in range(4):
        oracle.lib.orc_set_switch(C.c_int(w), C.c_int(0))


def soak_rounds(n_per_map=100):
    gts = soak_cases.load_gt()
    for mapname, _ in soak_cases.MAPS:
        mean, cov = soak_cases.load_map(mapname)
        for r in range(n_per_map):
            yield mapname, r, mean, cov, soak_cases.gen(mapname, r, mean, cov, gts, Cam)


def run_pose(oracle, f):
    return oracle.optimize_current_pose(Cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])


def same_pose(a, b, what):
    dt, dr = pose_err(a, b)
    assert dt < TOL and dr < TOL, (what, dt, dr)
    return max(dt, dr)


def test_pose_refine_under_solver_switches(oracle, switch):
    """optimizeCurrentPose (LinearSolverDense): golden frames + the pose problems of 200 soak rounds."""
    d = g("golden_pose.npz")
    frames = [dict(pose_init=d["f%d_pose_init" % j], Xw=d["f%d_Xw" % j], obs=d["f%d_obs" % j], octave=d["f%d_octave" % j]) for j in range(3)]
    frames += [c["pose"] for _, _, _, _, c in soak_rounds()]
    assert len(frames) == 203
    ref = [run_pose(oracle, f) for f in frames]
    for which, value in ((SW_LDLT, 1), (SW_LDLT, 2), (SW_LAMBDA, 1)):
        switch(which, value)
        errs = []
        for i, f in enumerate(frames):
            p, o, n = run_pose(oracle, f)
            errs.append(same_pose(p, ref[i][0], (which, value, i)))
            assert np.array_equal(o, ref[i][1]) and n == ref[i][2], (which, value, i)
        assert (np.array(errs) < TOL_MOST).mean() >= 0.95, (which, value, np.sort(errs)[-12:])
        switch(which, 0)


def test_structure_refine_and_local_ba_under_solver_switches(oracle, switch, map_v1):
    """jointOptimization (LinearSolverEigen = SimplicialLDLT): the golden windows, 30 random windows of the soak and the
    per-frame problems of the soak rounds (every 4th round has one: 50 of them)."""
    gts = soak_cases.load_gt()
    maps = {m: soak_cases.load_map(m) for m, _ in soak_cases.MAPS}
    hs = {m: oracle.gmm_create(*maps[m]) for m in maps}
    d = g("golden_ba.npz")
    jobs = []
    for k in ("a", "b"):
        jobs.append(("map_v1", int(d[k + "_P"]), int(d[k + "_F"]), d[k + "_poses"], d[k + "_prior"], d[k + "_points"], d[k + "_assoc"],
                     d[k + "_obs_ptr"], d[k + "_obs_pose"], d[k + "_obs_uvr"], d[k + "_obs_oct"]))
    for w in soak_cases.gen_ba(30, *maps["map_v1"], gts, Cam):
        p = w["problem"]
        idx, d2 = oracle.associate3d(hs["map_v1"], p["points"])
        a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
        jobs.append(("map_v1", w["P"], w["F"], p["poses"], p["prior"], p["points"], a, p["obs_ptr"], p["obs_pose"], p["obs_uvr"], p["obs_oct"]))
    for mapname, r, mean, cov, c in soak_rounds():
        f = c["track"]
        if f is None:
            continue
        keep = np.nonzero(f["octave"] >= 0)[0]
        idx, d2 = oracle.associate3d(hs[mapname], f["Xw"][keep])
        a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
        L = len(keep)
        jobs.append((mapname, 1, 0, f["pose_init"][None], np.zeros(1, np.uint8), f["Xw"][keep], a, np.arange(L + 1, dtype=np.int32),
                     np.zeros(L, np.int32), f["obs"][keep], f["octave"][keep]))
        # ... and anchored by the prior edge (gl_track_frames_anchored)
        jobs.append((mapname, 1, 0, f["pose_init"][None], np.ones(1, np.uint8), f["Xw"][keep], a, np.arange(L + 1, dtype=np.int32),
                     np.zeros(L, np.int32), f["obs"][keep], f["octave"][keep]))
    assert len(jobs) == 2 + 30 + 100

    def run(j):
        return oracle.joint_optimization(hs[j[0]], Cam, *j[1:])
    ref = [run(j) for j in jobs]
    loose = 0
    for which, value in ((SW_LDLT, 1), (SW_LDLT, 2), (SW_LAMBDA, 1)):
        switch(which, value)
        for i, j in enumerate(jobs):
            r = run(j)
            assert np.array_equal(r[2], ref[i][2]) and np.array_equal(r[3], ref[i][3]), (which, value, i)  # dropped / erased: decisions
            worst = max(max(pose_err(r[0][q], ref[i][0][q])) for q in range(j[1]))
            if worst >= TOL:
                # an unanchored single free pose (no prior, no fixed key-frame) leaves a gauge direction to the damping alone:
                # the soak of round 2 found the ORACLE ITSELF moving by 1e-4 m on such frames when one input moves by an ulp
                # (DESIGN.md 2).  Tolerated only there, counted, and bounded.
                assert j[1] == 1 and j[2] == 0 and not j[4].any() and worst < 1e-3, (which, value, i, worst)
                loose += 1
        switch(which, 0)
    assert loose <= 6, loose  # (3 switch settings x 50 unanchored frames: a handful at most)
    for h in hs.values():
        oracle.gmm_destroy(h)


def test_components_and_their_consumers_under_the_eigen_solver_switch(oracle, switch):
    """SelfAdjointEigenSolver: eigenvalues to rounding, eigenvectors up to sign; thresholds (is_degenerated, is_salient),
    renderView's culls and every consumer of the plane normal must decide the same."""
    d = g("golden_components.npz")
    maps = {m: soak_cases.load_map(m) for m, _ in soak_cases.MAPS}

    def build():
        out = {}
        for name, (mean, cov) in list(maps.items()) + [("golden", (d["mean"], d["cov"]))]:
            h = oracle.gmm_create(mean, cov)
            out[name] = (h, oracle.gmm_get(h))
        return out
    ref = build()
    switch(SW_EIG, 1)
    alt = build()
    for name in ref:
        a, b = ref[name][1], alt[name][1]
        assert np.array_equal(a["flags"], b["flags"]), name
        rel = np.abs(a["scale"] - b["scale"]) / np.maximum(np.abs(a["scale"]).max(1, keepdims=True), 1e-300)
        assert rel.max() < 1e-12, (name, rel.max())
        na, nb = a["axis"].reshape(-1, 3, 3), b["axis"].reshape(-1, 3, 3)
        # a well-separated eigenvector agrees up to sign (the plane normal of a degenerate component always is)
        gap = (a["scale"][:, 1] - a["scale"][:, 0]) / np.maximum(a["scale"][:, 2], 1e-300)
        dots = np.abs((na[:, :, 0] * nb[:, :, 0]).sum(1))
        assert (1 - dots[gap > 1e-6]).max() < 1e-9, name
    # consumers of the normal on the real map: view rendering, point optimisation, key-frame association, triangulation
    gd, gc, gt_ = g("golden_view.npz"), g("golden_cma.npz"), g("golden_tri.npz")
    N = gt_["x3d"].shape[0]
    n1, n2 = (gt_["cand1"] >= 0).sum(1).astype(np.int32), (gt_["cand2"] >= 0).sum(1).astype(np.int32)

    def consumers(h):
        out = [oracle.render_view(h, Cam, p)[0] for p in gd["poses"]]
        out.append(oracle.check_map_association(h, Cam, gc["pose"], gc["pts"], gc["uvr"], gc["octave"], gc["cand"], gc["ncand"]))
        out.append(oracle.optimize_triangulation(h, Cam, gt_["x3d"], np.tile(gt_["pose1"], (N, 1)), gt_["uvr1"], gt_["oct1"],
                                                 np.tile(gt_["pose2"], (N, 1)), gt_["uvr2"], gt_["oct1"], gt_["cand1"], n1, gt_["cand2"], n2))
        return out
    switch(SW_EIG, 0)
    ra = consumers(ref["map_v1"][0])   # components built with Jacobi, consumers with Jacobi (2-D eigen in renderView)
    switch(SW_EIG, 1)
    rb = consumers(alt["map_v1"][0])   # everything with QL
    for x, y in zip(ra[:-2], rb[:-2]):
        assert np.array_equal(x, y)  # rendered component lists
    for (ca, pa), (cb, pb) in zip(ra[-2:], rb[-2:]):
        assert np.array_equal(ca, cb) and np.abs(pa - pb).max() < TOL
    for name in ref:
        oracle.gmm_destroy(ref[name][0])
        oracle.gmm_destroy(alt[name][0])


def test_triangulation_under_candidate_order_switch(oracle, switch, map_v1):
    """optimizeTriangulationVec walks an unordered_set<GaussianComponent*> (localization_opt.cpp:144-158): whatever order the
    hash gives.  Golden matches + the createMapPoints batches of 200 soak rounds under four orders: the map points that
    are created (type, component, position) must be the same; for REJECTED matches the by-product candidate is not an
    output of the reference (round 2's soak: it flips with one ulp) and is not compared."""
    maps = {m: soak_cases.load_map(m) for m, _ in soak_cases.MAPS}
    hs = {m: oracle.gmm_create(*maps[m]) for m in maps}
    gc = g("golden_cmp.npz")
    batches = [("map_v1", {k: gc[k] for k in soak_cases.TRI_KEYS})]
    batches += [(mapname, {k: c["tri"][k] for k in soak_cases.TRI_KEYS}) for mapname, r, _, _, c in soak_rounds()]
    assert len(batches) == 201
    ref = [oracle.create_map_points(hs[m], Cam, **kw) for m, kw in batches]
    created = sum(int((t > 0).sum()) for _, t, _ in ref)
    assert created > 5000
    for order in (1, 2, 3):
        switch(SW_TRI, order)
        for i, (m, kw) in enumerate(batches):
            x, t, c = oracle.create_map_points(hs[m], Cam, **kw)
            assert np.array_equal(t, ref[i][1]), (order, i)
            made = ref[i][1] > 0
            assert np.array_equal(c[made], ref[i][2][made]), (order, i)
            near = made & (np.abs(ref[i][0]).max(1) < 100.0)  # (points beyond 100 m: near-parallel rays, ill-conditioned by construction)
            assert not near.any() or np.abs(x[near] - ref[i][0][near]).max() < TOL, (order, i)
        switch(SW_TRI, 0)
    for h in hs.values():
        oracle.gmm_destroy(h)
