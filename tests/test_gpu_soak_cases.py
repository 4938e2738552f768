"""The deviations the strict randomised soak found (tools/soak.py, 2 000 rounds per map; profiles/history/r2_soak_*.txt),
each held as a three-way test: HIP vs the C++ oracle vs the independent numpy restatement
(tests/golden/soak_numpy_ref.npz, tools/make_soak_golden.py), together with the oracle's OWN sensitivity to changes
that leave the mathematics untouched (inputs moved by an ulp, points re-ordered).  A case counts as ill-conditioned
only because the ORACLE ITSELF moves by more than the parity tolerance under such a change - and the HIP result must
then still lie within the spread the oracle and the numpy restatement show among themselves; everything else in the
same problem (the other features / matches, every decision) is held to the strict tolerance.
Inputs are regenerated from the (map, round) label by tools/soak_cases.py."""
import os

import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import api
from tests.conftest import GOLDEN
from tests.test_gpu_pose import pose_err
from tools import soak_cases as sc
from tools.make_soak_golden import BA, FALLBACK, TRACK, TRI, TRI_REJECTED

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def env(gpu, oracle):
    torch, ctx = gpu
    cam, prm, gts = api.Camera(), api.Params(), sc.load_gt()
    maps = {}
    for name in ("map_v1", "map_v2"):
        mean, cov = sc.load_map(name)
        maps[name] = (mean, cov, api.GMM(ctx, mean, cov, prm), oracle.gmm_create(mean, cov))
    ref = np.load(os.path.join(GOLDEN, "soak_numpy_ref.npz"))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return dict(torch=torch, ctx=ctx, cam=cam, prm=prm, gts=gts, maps=maps, ref=ref, T=T)


def ulp_variants(x, n, rng):
    out = []
    for _ in range(n):
        y = np.array(x, float).ravel().copy()
        j = int(rng.integers(0, y.size))
        y[j] = np.nextafter(y[j], np.inf if rng.integers(0, 2) else -np.inf)
        out.append(y.reshape(np.shape(x)))
    return out


@pytest.mark.parametrize("mapname,r", TRACK)
def test_soak_track_bifurcation(env, oracle, opt, mapname, r):
    """gl_track_frames on small, outlier-ridden frames (124 / 504 points, optimum 5 - 14 cm off the generating pose):
    the 5 / 5 / 40 Levenberg schedule stops before convergence next to an accept / reject flip.  Moving the
    observations by <= 1 ulp makes the ORACLE jump by 0.2 - 1 mm in some probes (2 of 48 on the first frame, half of
    them on the second) and by < 1e-6 in the others; HIP and the numpy restatement each land on one of the branches.
    Which frames sit on such a flip depends on the summation order, so the soak's single track deviation moved from
    the first frame to the second when the refine's order changed; both are held here.  Associations, chi2 and the
    bits across launch shapes are exact."""
    e = env
    mean, cov, g, h = e["maps"][mapname]
    f = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["track"]
    keep, p_ref, pts_ref, a_ref, idx0, d20 = sc.track_oracle(oracle, h, e["cam"], f)
    res = {}
    for shape in (0, 1):
        opt("ba_shape", shape)
        pose, Xw = e["T"](f["pose_init"][None]), e["T"](f["Xw"][None])
        assoc, d2 = gmmloc_amd.track_frames(e["ctx"], g, e["cam"], e["prm"], pose, Xw, e["T"](f["obs"][None]), e["T"](f["octave"][None]))
        e["torch"].cuda.synchronize()
        res[shape] = (pose.cpu().numpy()[0], Xw.cpu().numpy()[0], assoc.cpu().numpy()[0], d2.cpu().numpy()[0])
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)  # one summation order: the same bits whatever the launch shape
    pose_hip, _, assoc, d2 = res[0]
    assert np.array_equal(d2[keep], d20) and np.array_equal(assoc[keep], a_ref)  # decisions are exact
    hip = max(pose_err(pose_hip, p_ref))
    # the oracle's own sensitivity: points re-ordered / observations moved by <= 1 ulp
    rng = np.random.default_rng(0)
    probes = []
    for _ in range(12):
        _, p1, _, _, _, _ = sc.track_oracle(oracle, h, e["cam"], f, rng.permutation(len(keep)))
        probes.append(max(pose_err(p1, p_ref)))
    for _ in range(36):
        gobs = f["obs"] * (1 + 3e-16 * rng.standard_normal(f["obs"].shape))
        gobs[f["obs"] < 0] = f["obs"][f["obs"] < 0]
        _, p1, _, _, _, _ = sc.track_oracle(oracle, h, e["cam"], dict(f, obs=gobs))
        probes.append(max(pose_err(p1, p_ref)))
    probes = np.array(probes)
    numpy_vs_oracle = max(pose_err(e["ref"]["track_%s_r%d_pose" % (mapname, r)], p_ref))
    spread = max(probes.max(), numpy_vs_oracle)
    assert (probes > TOL).sum() >= 1, "the oracle no longer jumps: this frame must then pass the strict tolerance"
    assert probes.min() < TOL    # ... and stays inside the strict tolerance in other probes: a flip between branches
    assert hip < TOL or hip < 10 * spread, (hip, spread, numpy_vs_oracle)


def test_soak_track_near_singular_reduced_system(env, oracle, opt):
    """Frame v1 r19656 of the 20 000-round soak (402 points): the reduced 6 x 6 system of the last optimize(40) is near
    singular (the gauge direction is held by lambda alone, cond ~1e11), so the Levenberg path hangs on the last digits
    of the Schur complement.  With the point contributions evaluated as A - A D^-1 A (a subtraction that keeps 5 digits
    for a point held by its reprojection alone) HIP rejected a step the oracle accepts at trial 36, took a 12-trial
    detour and ran out of the iteration budget 1.75e-5 m from an optimum that the oracle, its FMA-contracted build, the
    numpy restatement and 400 perturbed oracle runs all reach - a real weakness, not an ill-conditioned input.  Evaluated
    as M D^-1 A (gl_ba_fast_impl.hpp, gl_ba.hip) every kernel follows the oracle's path: strict tolerance here, for the
    batch shape, the latency shape and the general kernel.  profiles/history/r2f_track_v1_r19656_trace*.txt"""
    e = env
    mean, cov, g, h = e["maps"]["map_v1"]
    f = sc.gen("map_v1", 19656, mean, cov, e["gts"], e["cam"])["track"]
    keep, p_ref, pts_ref, a_ref, idx0, d20 = sc.track_oracle(oracle, h, e["cam"], f)
    for name, value in (("ba_shape", 0), ("ba_shape", 1), ("ba_slow", 1)):
        opt(name, value)
        pose, Xw = e["T"](f["pose_init"][None]), e["T"](f["Xw"][None])
        assoc, d2 = gmmloc_amd.track_frames(e["ctx"], g, e["cam"], e["prm"], pose, Xw, e["T"](f["obs"][None]), e["T"](f["octave"][None]))
        e["torch"].cuda.synchronize()
        assert np.array_equal(assoc.cpu().numpy()[0][keep], a_ref)
        assert max(pose_err(pose.cpu().numpy()[0], p_ref)) < TOL, name
        opt(name, -1 if name == "ba_shape" else 0)


@pytest.mark.parametrize("r", BA)
def test_soak_ba_gauge_free_window(env, oracle, r):
    """gl_joint_optimization on a window with ONE free key-frame, no fixed one and no prior: the gauge is free, the
    normal equations are singular up to the damping, and the oracle's own answer moves by 0.4 mm when its points are
    re-ordered.  Decisions (dropped associations, erased observations) are exact."""
    from tests.test_gpu_ba import run_gpu
    e = env
    mean, cov, g, h = e["maps"]["map_v1"]
    b = sc.gen_ba(r + 1, mean, cov, e["gts"], e["cam"])[r]
    p = b["problem"]
    assert p["P"] == 1 and p["F"] == 0 and not b["prior"]
    idx, d2 = oracle.associate3d(h, p["points"])
    a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    e["ctx"].set_option("bagen_nb", b["nb"])
    try:
        out = run_gpu((e["torch"], e["ctx"]), g, e["cam"], e["prm"], [p], [a])
    finally:
        e["ctx"].set_option("bagen_nb", 0)
    args = (p["obs_ptr"], p["obs_pose"], p["obs_uvr"], p["obs_oct"])
    ref = oracle.joint_optimization(h, e["cam"], 1, 0, p["poses"], p["prior"], p["points"], a, *args)
    nobs = len(p["obs_pose"])
    assert np.array_equal(out[2][0], ref[2]) and np.array_equal(out[3][0][:nobs], ref[3])
    hip = max(pose_err(out[0][0][0], ref[0][0]))
    rng = np.random.default_rng(0)
    L = len(p["points"])
    spread = 0.0
    for _ in range(6):
        perm = rng.permutation(L)
        optr = np.concatenate([[0], np.cumsum(np.diff(p["obs_ptr"])[perm])]).astype(np.int32)
        sel = np.concatenate([np.arange(p["obs_ptr"][l], p["obs_ptr"][l + 1]) for l in perm]).astype(int)
        r2 = oracle.joint_optimization(h, e["cam"], 1, 0, p["poses"], p["prior"], p["points"][perm], a[perm], optr, p["obs_pose"][sel],
                                       p["obs_uvr"][sel], p["obs_oct"][sel])
        spread = max(spread, max(pose_err(r2[0][0], ref[0][0])))
    numpy_vs_oracle = max(pose_err(e["ref"]["ba_r%d_poses" % r][0], ref[0][0]))
    assert spread > TOL, "the oracle is stable under re-ordering: this window must then pass the strict tolerance"
    assert hip < 10 * max(spread, numpy_vs_oracle), (hip, spread, numpy_vs_oracle)


@pytest.mark.parametrize("mapname,r,j", FALLBACK)
def test_soak_check_map_association_fallback(env, oracle, mapname, r, j):
    """checkMapAssociation's no-association fallback (gmmloc_opt.cpp:237-256) also "refines" features whose
    observation is inconsistent (u < 0, negative disparity): five Gauss-Newton steps that throw the point tens of metres
    to kilometres away - a result that the oracle itself changes in the same digits when the point moves by one ulp.
    The association decisions of ALL features and every other point of the key-frame are exact / within 1e-9."""
    e = env
    mean, cov, g, h = e["maps"][mapname]
    ch = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["chain"]
    T = e["T"]
    cand, ncand, vids, nview = g.search2d(e["cam"], T(ch["pose"][None]), T(ch["obs"][None, :, :2].copy()), None, k=5, view_cap=4096)
    pd = T(ch["pts"][None])
    out = api.check_map_association(e["ctx"], g, e["cam"], e["prm"], T(ch["pose"][None]), pd, T(ch["obs"][None]), T(ch["octave"][None]), cand, ncand)
    e["torch"].cuda.synchronize()
    ids, _, _, _ = oracle.render_view(h, e["cam"], ch["pose"])
    c_ref, n_ref = oracle.search_correspondence(h, ch["obs"][:, :2].copy(), 5)
    assert np.array_equal(vids[0].cpu().numpy()[:len(ids)], ids) and np.array_equal(cand[0].cpu().numpy(), c_ref)
    keep = ch["octave"] >= 0
    pts, obs, octv, c_k, n_k = ch["pts"][keep], ch["obs"][keep], ch["octave"][keep], c_ref[keep], n_ref[keep]
    o_ref, p_ref = oracle.check_map_association(h, e["cam"], ch["pose"], pts, obs, octv, c_k, n_k)
    og, pg = out[0].cpu().numpy()[keep], pd[0].cpu().numpy()[keep]
    assert np.array_equal(og, o_ref)
    dev = np.abs(pg - p_ref).max(1)
    others = np.arange(len(dev)) != j
    assert dev[others].max() <= 1e-9
    assert o_ref[j] == -1 and np.linalg.norm(p_ref[j] - pts[j]) > 10.0  # unassociated, thrown > 10 m away
    spread = 0.0
    for v in ulp_variants(pts[j], 12, np.random.default_rng(0)):
        _, pv = oracle.check_map_association(h, e["cam"], ch["pose"], v[None], obs[j][None], octv[j][None], c_k[j][None], n_k[j][None])
        spread = max(spread, np.abs(pv[0] - p_ref[j]).max())
    rec = e["ref"]["fallback_%s_r%d_f%d" % (mapname, r, j)]
    numpy_vs_oracle = np.abs(rec[1:] - p_ref[j]).max()
    assert int(rec[0]) == -1
    assert dev[j] <= 1e-9 or (spread > 1e-9 and dev[j] < 10 * max(spread, numpy_vs_oracle)), (dev[j], spread, numpy_vs_oracle)


@pytest.mark.parametrize("mapname,r,j", TRI)
def test_soak_create_map_points_far(env, oracle, mapname, r, j):
    """createMapPoints on an epipolar match with near-parallel rays: the linear triangulation lands 10^4 .. 10^9 m
    away, where 20 Gauss-Newton steps are chaotic - one ulp on a key-point moves the oracle's point by as much as HIP
    and the numpy restatement differ from it, and flips its sign-dependent checks.  All other matches are exact."""
    e = env
    mean, cov, g, h = e["maps"][mapname]
    m = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["tri"]
    x_ref, t_ref, c_ref = oracle.create_map_points(h, e["cam"], **m)
    x, t, c = api.create_map_points(e["ctx"], g, e["cam"], e["prm"], *[e["T"](m[k]) for k in sc.TRI_KEYS])
    e["torch"].cuda.synchronize()
    xg, tg, cg = x.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy()
    others = np.arange(len(t_ref)) != j
    with np.errstate(invalid="ignore"):
        sane = others & (np.linalg.norm(x_ref, axis=1) < 100.0)
    assert np.array_equal(tg[sane], t_ref[sane]) and np.array_equal(cg[sane], c_ref[sane])
    acc = sane & (t_ref > 0)
    assert np.abs(xg[acc] - x_ref[acc]).max() <= 1e-8
    assert np.linalg.norm(x_ref[j]) > 1e3  # the match in question: far beyond the 10 m room
    one = {k: m[k][j:j + 1] for k in m}
    spread, flips = 0.0, 0
    for v in ulp_variants(m["uvr1"][j], 12, np.random.default_rng(0)):
        xv, tv, cv = oracle.create_map_points(h, e["cam"], **dict(one, uvr1=v[None]))
        flips += int(tv[0] != t_ref[j] or cv[0] != c_ref[j])
        with np.errstate(invalid="ignore"):
            spread = max(spread, np.nanmax(np.abs(xv[0] - x_ref[j])))
    rec = e["ref"]["tri_%s_r%d_m%d" % (mapname, r, j)]
    numpy_vs_oracle = np.nanmax(np.abs(rec[2:] - x_ref[j]))
    dev = np.nanmax(np.abs(xg[j] - x_ref[j]))
    decision_equal = tg[j] == t_ref[j] and cg[j] == c_ref[j]
    assert spread > 1e-8, "the oracle is stable under an ulp: this match must then pass the strict tolerance"
    assert dev < 10 * max(spread, numpy_vs_oracle), (dev, spread, numpy_vs_oracle)
    assert decision_equal or flips > 0 or int(rec[0]) != t_ref[j], (tg[j], t_ref[j], rec[0], flips)


@pytest.mark.parametrize("mapname,r,j", TRI_REJECTED)
def test_soak_create_map_points_rejected_match(env, oracle, mapname, r, j):
    """createMapPoints on a mono-mono match that HIP, the oracle and the numpy restatement all REJECT (type 0: no map
    point is created, localization_opt.cpp:286-420 `continue`s): what optimizeTriangulationVec leaves behind - the last
    candidate it settled on and the point it moved - is no output of the reference, and it sits on a knife edge: the
    oracle's own by-product flips between two candidates when ONE input moves by an ulp (21 of 48 / 7 of 48 probes).
    The 6 000-round soak found the two matches; every created point and every decision of the two batches is exact."""
    e = env
    mean, cov, g, h = e["maps"][mapname]
    m = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["tri"]
    x_ref, t_ref, c_ref = oracle.create_map_points(h, e["cam"], **m)
    x, t, c = api.create_map_points(e["ctx"], g, e["cam"], e["prm"], *[e["T"](m[k]) for k in sc.TRI_KEYS])
    e["torch"].cuda.synchronize()
    xg, tg, cg = x.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy()
    with np.errstate(invalid="ignore"):
        sane = (np.arange(len(t_ref)) != j) & (np.linalg.norm(x_ref, axis=1) < 100.0)
    assert np.array_equal(tg[sane], t_ref[sane]) and np.array_equal(cg[sane], c_ref[sane])
    acc = sane & (t_ref > 0)
    assert np.abs(xg[acc] - x_ref[acc]).max() <= 1e-8
    rec = e["ref"]["tri_%s_r%d_m%d" % (mapname, r, j)]
    assert tg[j] == 0 and t_ref[j] == 0 and int(rec[0]) == 0  # rejected three ways: the decision is the same
    one = {k: m[k][j:j + 1] for k in m}
    seen = {int(c_ref[j])}
    for key in ("uvr1", "uvr2", "pose1", "pose2"):
        for v in ulp_variants(m[key][j], 12, np.random.default_rng(1)):
            _, tv, cv = oracle.create_map_points(h, e["cam"], **dict(one, **{key: v[None]}))
            assert tv[0] == 0
            seen.add(int(cv[0]))
    assert len(seen) > 1, "the oracle's by-product is stable: this match must then compare equal"
    assert int(cg[j]) in seen and int(rec[1]) in seen


# ---- round 3: the deviations of the 20 000-round soak on the anchored per-frame path and on the pipelined local BA ----------
TRACK_PRIOR = [("map_v1", 16708), ("map_v2", 292), ("map_v2", 4632)]   # profiles/history/r3_soak_strict_20000.txt: 3 of 10 000 frames
BA_PIPE = [333, 3893]                                                    # 2 of 2 500 windows on the pipelined shape


def _reorder_spread(fn, n, rng, runs=12):
    """largest pose change of the ORACLE when its points are re-ordered (the mathematics is order-free, the sums are not)"""
    ref = fn(None)
    worst, moved = 0.0, 0
    for _ in range(runs):
        d = max(pose_err(fn(rng.permutation(n)), ref))
        worst = max(worst, d)
        moved += d > TOL
    return ref, worst, moved


@pytest.mark.parametrize("mapname,r", TRACK_PRIOR)
def test_soak_track_prior_bifurcation(env, oracle, opt, mapname, r):
    """gl_track_frames_anchored, 3 frames of 10 000: the 5 / 5 / 40 Levenberg schedule ends next to an accept / reject flip.
    The ORACLE'S OWN answer moves by more than the tolerance when its points are re-ordered (in 23 - 24 of 24 re-orderings,
    by 1.4e-5 ... 9.8e-5 m), HIP lies within that spread; associations are exact, and both launch shapes return the same bits."""
    e = env
    mean, cov, g, h = e["maps"][mapname]
    f = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["track"]
    keep = np.nonzero(f["octave"] >= 0)[0]
    a_ref = sc.track_oracle(oracle, h, e["cam"], f, prior=True)[3]
    ref, spread, moved = _reorder_spread(lambda perm: sc.track_oracle(oracle, h, e["cam"], f, perm=perm, prior=True)[1], len(keep),
                                         np.random.default_rng(0))
    one = e["torch"].ones(1, dtype=e["torch"].uint8).cuda()
    res = []
    for shape in (0, 1):
        opt("ba_shape", shape)
        pose, Xw = e["T"](f["pose_init"][None]), e["T"](f["Xw"][None])
        assoc, _, _ = gmmloc_amd.track_frames_anchored(e["ctx"], g, e["cam"], e["prm"], pose, Xw, e["T"](f["obs"][None]),
                                                       e["T"](f["octave"][None]), prior=one)
        e["torch"].cuda.synchronize()
        res.append((pose.cpu().numpy()[0], Xw.cpu().numpy()[0], assoc.cpu().numpy()[0]))
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.array_equal(res[0][2][keep], a_ref)
    hip = max(pose_err(res[0][0], ref))
    assert moved >= 6 and spread > TOL, "the oracle is stable under re-ordering: this frame must then pass the strict tolerance"
    assert hip < 10 * spread, (hip, spread)


@pytest.mark.parametrize("r", BA_PIPE)
def test_soak_ba_pipelined_gauge_free_window(env, oracle, opt, r):
    """The two windows of 2 500 on which the PIPELINED shape of gl_joint_optimization left the strict tolerance: free
    key-frames only, no fixed one, no prior (r333: one free pose, the window of round 2; r3893: two).  The oracle's own pose
    moves by 4.4e-4 / 2.5e-6 m under re-ordering and - r333 - its own decisions change in 7 of 24 re-orderings; HIP lies
    within the oracle's spread, and where the oracle's decisions are stable HIP's are equal to them."""
    from tests.test_gpu_ba import run_gpu
    e = env
    mean, cov, g, h = e["maps"]["map_v1"]
    b = sc.gen_ba(r + 1, mean, cov, e["gts"], e["cam"])[r]
    p = b["problem"]
    assert p["F"] == 0 and not b["prior"]
    idx, d2 = oracle.associate3d(h, p["points"])
    a = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    L, P = len(p["points"]), p["P"]

    def run_oracle(perm):
        if perm is None:
            return oracle.joint_optimization(h, e["cam"], P, 0, p["poses"], p["prior"], p["points"], a, p["obs_ptr"], p["obs_pose"], p["obs_uvr"], p["obs_oct"])
        optr = np.concatenate([[0], np.cumsum(np.diff(p["obs_ptr"])[perm])]).astype(np.int32)
        sel = np.concatenate([np.arange(p["obs_ptr"][l], p["obs_ptr"][l + 1]) for l in perm]).astype(int)
        q = oracle.joint_optimization(h, e["cam"], P, 0, p["poses"], p["prior"], p["points"][perm], a[perm], optr, p["obs_pose"][sel],
                                      p["obs_uvr"][sel], p["obs_oct"][sel])
        dr, er = np.zeros(L, np.uint8), np.zeros(len(sel), np.uint8)
        dr[perm], er[sel] = q[2], q[3]
        return q[0], None, dr, er, q[4]
    ref = run_oracle(None)
    rng = np.random.default_rng(0)
    spread, flips = 0.0, 0
    for _ in range(12):
        q = run_oracle(rng.permutation(L))
        spread = max(spread, max(max(pose_err(q[0][j], ref[0][j])) for j in range(P)))
        flips += int(not (np.array_equal(q[2], ref[2]) and np.array_equal(q[3], ref[3])))
    opt("bagen_mode", 2)
    out = run_gpu((e["torch"], e["ctx"]), g, e["cam"], e["prm"], [p], [a])
    nobs = len(p["obs_pose"])
    hip = max(max(pose_err(out[0][0][j], ref[0][j])) for j in range(P))
    assert spread > TOL and hip < 10 * spread, (hip, spread)
    if flips == 0:
        assert np.array_equal(out[2][0], ref[2]) and np.array_equal(out[3][0][:nobs], ref[3])


# ---- round 3, 50 000-round soak (profiles/history/r3c_soak_strict_50000.txt): ONE decision of createMapPoints in 48 M matches ------------
def test_soak_create_map_points_parallax_knife_edge(env, oracle):
    """createMapPoints decides between the two-view triangulation and the stereo un-projection with
    `cosParallaxRays < cos(2 atan2(mb / 2, depth))`, both sides FLOAT (localization_opt.cpp:306-321).  On match 270 of
    map_v2 round 33994 the exact value of the right-hand side lies 0.5015 float ulps above the float cosParallaxRays: whether
    the comparison holds is the last bit of a float cosine of a float arc tangent, i.e. of the math library (glibc in the oracle,
    the device library on the GPU) - with the device library's cosf / atan2f HIP took the stereo branch, the oracle the two-view
    one.  The created point was the SAME (both initial points converge in optimizeTriangulationVec: 8.9e-16), so was its
    component; only the MapPoint type recorded the branch (2 against 4).  k_tri_pre evaluates the two functions correctly
    rounded since (double, rounded once: what glibc returns on all but 0.036 % of the depths, against 0.34 % of last-bit
    differences for the device library) and agrees on this match; the test holds the knife edge and allows either branch.
    Every other match of the batch is exact."""
    e = env
    mapname, r, j = "map_v2", 33994, 270
    mean, cov, g, h = e["maps"][mapname]
    m = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["tri"]
    x_ref, t_ref, c_ref = oracle.create_map_points(h, e["cam"], **m)
    x, t, c = api.create_map_points(e["ctx"], g, e["cam"], e["prm"], *[e["T"](m[k]) for k in sc.TRI_KEYS])
    e["torch"].cuda.synchronize()
    xg, tg, cg = x.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy()
    with np.errstate(invalid="ignore"):
        sane = (np.arange(len(t_ref)) != j) & (np.linalg.norm(x_ref, axis=1) < 100.0)
    assert np.array_equal(tg[sane], t_ref[sane]) and np.array_equal(cg[sane], c_ref[sane])
    acc = sane & (t_ref > 0)
    assert np.abs(xg[acc] - x_ref[acc]).max() <= 1e-8
    # the match in question: same point, same component, the type differs at most by the branch (1 <-> 3, 2 <-> 4)
    assert cg[j] == c_ref[j] and np.abs(xg[j] - x_ref[j]).max() <= 1e-8
    assert tg[j] == t_ref[j] or {int(tg[j]), int(t_ref[j])} in ({1, 3}, {2, 4})
    # ... and the threshold really is a knife edge: the exact right-hand side within one float ulp of the float left-hand side
    cam = e["cam"]
    k1, k2, p1, p2 = m["uvr1"][j], m["uvr2"][j], m["pose1"][j], m["pose2"][j]

    def ray(p, k):  # world direction of the key-point's viewing ray (pose = qx qy qz qw t: T_cw)
        qx, qy, qz, qw = -p[0], -p[1], -p[2], p[3]
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                      [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                      [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        return R @ np.array([(k[0] - cam.cx) / cam.fx, (k[1] - cam.cy) / cam.fy, 1.0])
    r1, r2 = ray(p1, k1), ray(p2, k2)
    cos_rays = np.float32(r1 @ r2 / np.linalg.norm(r1) / np.linalg.norm(r2))
    depth = np.float64(m["depth1"][j] if k1[2] >= 0 else m["depth2"][j])
    cps_exact = np.cos(2.0 * np.arctan2(np.float64(np.float32(cam.bf / cam.fx / 2)), depth))
    assert abs(float(cos_rays) - cps_exact) < float(np.spacing(np.float32(cps_exact)))


# ---- round 3, rounds 50 000 .. 86 000 of the soak: ONE anchored frame in 43 000 that is NOT ill-conditioned ----------------------------
def test_soak_track_prior_single_pose_kernels_real_deviation(env, oracle, opt):
    """map_v1 round 63072 (138 points, prior edge): in round 3 gl_track_frames_anchored ended 3.6e-5 rad / 5.2e-6 m from the oracle
    (the general single-pose kernel k_ba1, option ba_slow, 6.2e-5 rad) although the ORACLE IS STABLE here - it moves by < 4e-8 under
    re-orderings and under relative changes of its inputs from 3e-16 to 1e-12, the numpy restatement agrees with it to 1.6e-8,
    and HIP's own general local-BA kernels solve the same problem to 2e-8 of it.  A REAL deviation (1 of 43 000 anchored frames in
    86 000 soak rounds).  Cause, found in round 4 by replaying the kernel's arithmetic on the host (tools/emul_ba1.py): point 83
    of the frame, an outlier with a plane edge, runs away along its plane to 5 km from the camera; its damped 3 x 3 block then
    has the eigenvalues 400 / 7e-3 / lambda, the determinant of the cofactor inverse is rounding noise once lambda < 1e-6, and the
    product form of the Schur term carried that noise - times |q|^2 = 2.6e7 through G = [-[q]x | I] - into the reduced pose system:
    pose steps of 1e-4 .. 1e-3 where the oracle's are 1e-6.  The single-pose kernels factorise the block (L Delta L^T, backward
    stable whatever its scaling) since; every launch shape and the general kernel hold the contract's 1e-6 here."""
    e = env
    mapname, r = "map_v1", 63072
    mean, cov, g, h = e["maps"][mapname]
    f = sc.gen(mapname, r, mean, cov, e["gts"], e["cam"])["track"]
    keep, p_ref, pts_ref, a_ref, idx0, d20 = sc.track_oracle(oracle, h, e["cam"], f, prior=True)
    rng = np.random.default_rng(0)
    for _ in range(6):  # the oracle itself: stable under re-ordering and input noise
        _, p1, _, _, _, _ = sc.track_oracle(oracle, h, e["cam"], f, rng.permutation(len(keep)), prior=True)
        assert max(pose_err(p1, p_ref)) < 2e-7
        q = dict(f)
        q["obs"] = np.where(f["obs"] < 0, f["obs"], f["obs"] * (1 + 1e-13 * rng.standard_normal(f["obs"].shape)))
        _, p1, _, _, _, _ = sc.track_oracle(oracle, h, e["cam"], q, prior=True)
        assert max(pose_err(p1, p_ref)) < 2e-7
    T, torch, ctx = e["T"], e["torch"], e["ctx"]
    one = torch.ones(1, dtype=torch.uint8).cuda()
    bits = {}
    for name, options in (("batch", {"ba_shape": 0}), ("latency", {"ba_shape": 1}), ("general", {"ba_slow": 1})):
        for k, v in options.items():
            opt(k, v)
        pose, Xw = T(f["pose_init"][None]), T(f["Xw"][None])
        assoc = gmmloc_amd.track_frames_anchored(ctx, g, e["cam"], e["prm"], pose, Xw, T(f["obs"][None]), T(f["octave"][None]), prior=one)[0]
        torch.cuda.synchronize()
        for k in options:
            opt(k, {"ba_shape": -1, "ba_slow": 0}[k])
        dt, dr = pose_err(pose.cpu().numpy()[0], p_ref)
        assert np.array_equal(assoc.cpu().numpy()[0][keep], a_ref), name
        assert dt < 1e-6 and dr < 1e-6, (name, dt, dr)  # the north-star tolerance (round 3: 5.2e-6 m / 3.6e-5 rad, asserted 2e-4)
        bits[name] = pose.cpu().numpy().tobytes()
    assert bits["batch"] == bits["latency"]
    # the same problem through gl_joint_optimization: one free pose with the prior edge, both launch shapes
    L = len(keep)
    a0 = np.where(d20 <= 9.0, idx0, -1).astype(np.int32)
    for mode in (1, 2):
        opt("bagen_mode", mode)
        poses, pts = T(f["pose_init"][None, None].copy()), T(f["Xw"][keep][None].copy())
        api.joint_optimization(ctx, g, e["cam"], e["prm"], 1, 0, poses, T(np.ones((1, 1), np.uint8)), pts, T(a0[None]),
                               T(np.arange(L + 1, dtype=np.int32)[None]), T(np.zeros((1, L), np.int32)), T(f["obs"][keep][None].copy()),
                               T(f["octave"][keep][None].astype(np.int32)))
        torch.cuda.synchronize()
        assert max(pose_err(poses.cpu().numpy()[0, 0], p_ref)) < 1e-6
