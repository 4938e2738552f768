"""GPU parity: GMM construction (A0, A2) and association (A1 exhaustive, A6) vs the oracle.
Bit-exact for indices / flags; the Mahalanobis values themselves are bit-identical because
both sides evaluate the same canonical fp64 expression."""
import os

import numpy as np
import pytest

from gmmloc_amd import synth
from gmmloc_amd import api

pytestmark = pytest.mark.gpu


def _mk(gpu, oracle, mean, cov):
    torch, ctx = gpu
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    return g, h


@pytest.mark.parametrize("which", ["v1", "v2", "synth4096"])
def test_component_build_matches_oracle(gpu, oracle, map_v1, map_v2, which):
    mean, cov = {"v1": map_v1, "v2": map_v2, "synth4096": synth.synth_gmm(4096, 1)}[which]
    g, h = _mk(gpu, oracle, mean, cov)
    ref = oracle.gmm_get(h)
    assert g.K == mean.shape[0]
    # cov_inv feeds the bit-exact indices: must be bit-identical
    assert np.array_equal(g.get(api.F_COV_INV), ref["cov_inv"])
    assert np.array_equal(g.get(api.F_DET), ref["det"])
    assert np.array_equal(g.get(api.F_FLAGS), ref["flags"])
    np.testing.assert_allclose(g.get(api.F_SCALE), ref["scale"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(g.get(api.F_SQRT_INFO), ref["sqrt_info"], rtol=1e-13, atol=0)
    # eigenvectors up to sign: compare n n^T of the plane normal (axis_.col(0)) where it is used
    A, B = g.get(api.F_AXIS).reshape(-1, 3, 3), ref["axis"].reshape(-1, 3, 3)
    deg = (ref["flags"] & 1).astype(bool)
    n0, n1 = A[deg][:, :, 0], B[deg][:, :, 0]
    assert np.abs(np.abs(np.einsum("ki,ki->k", n0, n1)) - 1).max() < 1e-12
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("which", ["v1", "v2"])
def test_neighbour_graph_matches_oracle(gpu, oracle, map_v1, map_v2, which):
    mean, cov = {"v1": map_v1, "v2": map_v2}[which]
    g, h = _mk(gpu, oracle, mean, cov)
    ptr, col, dist = oracle.neighbours(h)
    assert np.array_equal(g.get(api.F_NBS_PTR), ptr)
    assert np.array_equal(g.get(api.F_NBS_IDX), col)
    np.testing.assert_allclose(g.get(api.F_NBS_DIST), dist, rtol=0, atol=1e-12)
    if which == "v1":  # 16 048 directed edges, max degree 29
        assert len(col) == 16048 and np.diff(ptr).max() == 29
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("which,N,seed", [("v1", 2000, 1), ("v2", 2000, 2), ("synth4096", 2000, 1),
                                           ("synth4096", 2000, 2), ("synth4096", 2000, 3), ("synth4096", 1, 5),
                                           ("synth4096", 63, 6), ("synth4096", 257, 7), ("synth300", 5000, 8)])
def test_associate3d_brute_bit_exact(gpu, oracle, map_v1, map_v2, which, N, seed):
    torch, ctx = gpu
    mean, cov = {"v1": map_v1, "v2": map_v2, "synth4096": synth.synth_gmm(4096, seed),
                 "synth300": synth.synth_gmm(300, seed)}[which]
    g, h = _mk(gpu, oracle, mean, cov)
    pts = synth.synth_points(mean, cov, N, seed)
    idx_ref, d2_ref = oracle.associate3d(h, pts)
    idx, d2 = g.associate3d(torch.from_numpy(pts).cuda())
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(d2.cpu().numpy(), d2_ref)  # same expression, same rounding
    oracle.gmm_destroy(h)


def test_associate3d_ties_and_exact_hits(gpu, oracle):
    """Duplicate components: the lowest index must win; a point at a mean has d2 == 0."""
    torch, ctx = gpu
    mean, cov = synth.synth_gmm(200, 11)
    mean = np.concatenate([mean, mean[:50]])
    cov = np.concatenate([cov, cov[:50]])
    g, h = _mk(gpu, oracle, mean, cov)
    pts = mean[:50].copy()
    idx, d2 = g.associate3d(torch.from_numpy(pts).cuda())
    idx_ref, d2_ref = oracle.associate3d(h, pts)
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert (d2.cpu().numpy() == 0).all() and (idx.cpu().numpy() < 200).all()
    oracle.gmm_destroy(h)


def test_associate3d_empty(gpu, map_v1):
    torch, ctx = gpu
    g = api.GMM(ctx, *map_v1)
    idx, d2 = g.associate3d(torch.empty((0, 3), dtype=torch.float64, device="cuda"))
    assert idx.numel() == 0


def test_associate3d_order_invariance(gpu, map_v1):
    """argmin must not depend on the order / batching of the points."""
    torch, ctx = gpu
    mean, cov = map_v1
    g = api.GMM(ctx, mean, cov)
    pts = synth.synth_points(mean, cov, 3000, 4)
    perm = np.random.default_rng(0).permutation(3000)
    a, _ = g.associate3d(torch.from_numpy(pts).cuda())
    b, _ = g.associate3d(torch.from_numpy(np.ascontiguousarray(pts[perm])).cuda())
    assert np.array_equal(a.cpu().numpy()[perm], b.cpu().numpy())


@pytest.mark.parametrize("which", ["v1", "v2"])
def test_knn3d_and_query_point(gpu, oracle, map_v1, map_v2, which):
    torch, ctx = gpu
    mean, cov = {"v1": map_v1, "v2": map_v2}[which]
    g, h = _mk(gpu, oracle, mean, cov)
    pts = synth.synth_points(mean, cov, 1500, 9)
    ki, kd, _ = oracle.knn3d(h, pts, 5)
    gi, gd = g.knn3d(torch.from_numpy(pts).cuda(), 5)
    assert np.array_equal(gi.cpu().numpy(), ki)
    assert np.array_equal(gd.cpu().numpy(), kd)
    if oracle.nf is not None:  # the reference's own vendored nanoflann
        ni, nd, _ = oracle.nanoflann_knn(mean, pts, 5)
        assert np.array_equal(gi.cpu().numpy(), ni)
    q = g.queryPoint(torch.from_numpy(pts).cuda())
    assert np.array_equal(q.cpu().numpy(), ki[:, 0])
    # few queries run a wave per query, many a thread per query: both kernels, other k, ties through duplicated means
    big = np.concatenate([synth.synth_points(mean, cov, 9000, 10), mean[:200], 0.5 * (mean[:100] + mean[1:101])])
    for k in (1, 3, 8):
        ki2, kd2, _ = oracle.knn3d(h, big, k)
        for sl in (slice(0, 9300), slice(9000, 9300), slice(9299, 9300)):  # 9 300 -> thread kernel, 300 / 1 -> wave kernel
            gi2, gd2 = g.knn3d(torch.from_numpy(big[sl]).cuda(), k)
            assert np.array_equal(gi2.cpu().numpy(), ki2[sl]) and np.array_equal(gd2.cpu().numpy(), kd2[sl])
    oracle.gmm_destroy(h)


def test_gmm_file_roundtrip(gpu, map_v1, tmp_path):
    torch, ctx = gpu
    mean, cov = map_v1
    g = api.GMM(ctx, mean, cov)
    p = tmp_path / "m.gmm"
    g.save(p)
    g2 = api.GMM.load(ctx, p)
    assert g2.K == g.K
    assert np.array_equal(g2.get(api.F_MEAN), mean)
    # saveGMMModel writes cov(i) column-major and loadGMMModel reads row-major (gmm_utils.cpp:54-59,102-104)
    assert np.array_equal(g2.get(api.F_COV).reshape(-1, 3, 3), cov.reshape(-1, 3, 3).transpose(0, 2, 1))
    with pytest.raises(api.GLError):
        api.GMM.load(ctx, tmp_path / "does_not_exist.gmm")
    (tmp_path / "bad.gmm").write_bytes(b"\x02\x05\x08")
    with pytest.raises(api.GLError):
        api.GMM.load(ctx, tmp_path / "bad.gmm")


# ---- the exact cell index behind ASSOC_BRUTE vs the plain N x K sweep (ASSOC_EXHAUSTIVE) -------

def _both(torch, g, pts):
    ctx = g.ctx
    t = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    ctx.set_option("assoc_index_min", 0)  # small problems default to the sweep; force the index
    try:
        a = g.associate3d(t, api.ASSOC_BRUTE)
    finally:
        ctx.set_option("assoc_index_min", -1)
    b = g.associate3d(t, api.ASSOC_EXHAUSTIVE)
    return [x.cpu().numpy() for x in a], [x.cpu().numpy() for x in b]


@pytest.mark.parametrize("which,N,seed", [("v1", 20000, 21), ("v2", 20000, 22), ("synth4096", 50000, 23),
                                           ("synth300", 5000, 24), ("synth65536", 4000, 25)])
@pytest.mark.parametrize("coop", [1, 0])  # wave-cooperative record gather (default) / a lane per record
def test_cell_index_equals_exhaustive(gpu, oracle, map_v1, map_v2, opt, which, N, seed, coop):
    """idx and chi2 bit-identical to the all-pairs sweep, for inliers, outliers (swept again) and
    points far outside the map."""
    torch, ctx = gpu
    opt("assoc_coop", coop)
    mean, cov = {"v1": map_v1, "v2": map_v2, "synth4096": synth.synth_gmm(4096, seed),
                 "synth300": synth.synth_gmm(300, seed), "synth65536": synth.synth_gmm(65536, seed)}[which]
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(seed)
    pts = synth.synth_points(mean, cov, N, seed)
    # + uniform points in and far around the bounding box, + points exactly on means and cell-ish lattices
    lo, hi = mean.min(0), mean.max(0)
    pts = np.concatenate([pts, rng.uniform(lo - 5, hi + 5, (N // 4, 3)), rng.uniform(lo, hi, (N // 4, 3)),
                          mean[rng.integers(0, len(mean), 200)],
                          np.round(rng.uniform(lo, hi, (500, 3)) * 8) / 8, np.array([[1e6, -1e6, 3.0], [0.0, 0.0, 1e12]])])
    (i1, d1), (i2, d2) = _both(torch, g, pts)
    assert np.array_equal(i1, i2)
    assert np.array_equal(d1, d2)
    # and against the oracle on a slice (the sweep itself is pinned by test_associate3d_brute_bit_exact)
    h = oracle.gmm_create(mean, cov)
    ir, dr = oracle.associate3d(h, pts[:1500])
    assert np.array_equal(i1[:1500], ir) and np.array_equal(d1[:1500], dr)
    oracle.gmm_destroy(h)


def test_cell_index_record_copy_option_same_bits(gpu, opt):
    """assoc_rec_pad: the cooperative gather from the one-record-per-128-byte-line copy (default) and from the 96-byte records
    give the same indices and the same chi2 bits (and both equal the all-pairs sweep)."""
    torch, ctx = gpu
    mean, cov = synth.synth_gmm(4096, 41)
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(41)
    lo, hi = mean.min(0), mean.max(0)
    pts = np.concatenate([synth.synth_points(mean, cov, 60000, 41), rng.uniform(lo - 2, hi + 2, (8000, 3))])
    out = {}
    for pad, lng, bal in ((1, 1, 1), (0, 1, 1), (1, 1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0)):
        opt("assoc_rec_pad", pad)
        opt("assoc_coop_long", lng)   # lists of more than three candidates through the cooperative gather (default) / by their lane
        opt("assoc_coop_bal", bal)    # a pair per lane and round (default) / the pairs of a point by the lane that owns it
        out[pad, lng, bal] = _both(torch, g, pts)
    (i1, d1), (ie, de) = out[1, 1, 1]
    assert np.array_equal(i1, ie) and np.array_equal(d1, de)
    for key in ((0, 1, 1), (1, 1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0)):
        (i0, d0), _ = out[key]
        assert np.array_equal(i1, i0) and np.array_equal(d1, d0), key


def test_cell_index_8_byte_cells_same_bits(gpu, opt):
    """assoc_cell8 (round 6): the packed cell table in 8 bytes per cell (count + three 20-bit component indices, or the place of a longer
    list) against the 16-byte cells of rounds 3 - 5: half the table, the same indices and chi2 bits, both walks (cooperative gather and a
    lane per record), short and long lists, and both equal the all-pairs sweep."""
    torch, ctx = gpu
    rng = np.random.default_rng(43)
    cases = []
    mean, cov = synth.synth_gmm(4096, 43)
    lo, hi = mean.min(0), mean.max(0)
    cases.append((mean, cov, np.concatenate([synth.synth_points(mean, cov, 60000, 43), rng.uniform(lo - 2, hi + 2, (8000, 3))])))
    K = 600  # many overlapping components: cells with tens of candidates
    m2 = rng.uniform(-0.5, 0.5, (K, 3))
    c2 = np.tile((np.eye(3) * 0.04).reshape(1, 9), (K, 1)) * rng.uniform(0.5, 1.5, (K, 1))
    cases.append((m2, c2, rng.uniform(-0.7, 0.7, (30000, 3))))
    for case, (mean, cov, pts) in enumerate(cases):
        res, size = {}, {}
        for c8 in (1, 0):
            opt("assoc_cell8", c8)  # read when the GMM's index is built
            g = api.GMM(ctx, mean, cov)
            size[c8] = g.index_info()["bytes"]["packed_cells"]
            for coop in (1, 0):
                opt("assoc_coop", coop)
                res[c8, coop] = _both(torch, g, pts)
        (i1, d1), (ie, de) = res[1, 1]
        assert np.array_equal(i1, ie) and np.array_equal(d1, de)
        for key in ((1, 0), (0, 1), (0, 0)):
            (i0, d0), _ = res[key]
            assert np.array_equal(i1, i0) and np.array_equal(d1, d0), key
        assert size[1] * 2 == size[0] and (size[1] > 0 or case > 0), (case, size)  # (the small map's index may come without a packed table)


def test_cell_index_long_lists_in_chunks(gpu):
    """Many overlapping components: cells with tens of candidates, more than one chunk of the wave's candidate table (384) per wave."""
    torch, ctx = gpu
    rng = np.random.default_rng(77)
    K = 600
    mean = rng.uniform(-0.5, 0.5, (K, 3))
    cov = np.tile((np.eye(3) * 0.04).reshape(1, 9), (K, 1)) * rng.uniform(0.5, 1.5, (K, 1))
    g = api.GMM(ctx, mean, cov)
    pts = rng.uniform(-0.7, 0.7, (30000, 3))
    (i1, d1), (i2, d2) = _both(torch, g, pts)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert (i1 >= 0).sum() > 10000


def test_cell_index_adversarial_components(gpu):
    """Components the index cannot bound (singular, indefinite, huge, needle-like with cond > 1e8,
    non-finite) must not change the result; duplicates must keep the lowest index."""
    torch, ctx = gpu
    mean, cov = synth.synth_gmm(500, 31)
    pts0 = synth.synth_points(mean, cov, 20000, 3)
    cov = cov.reshape(-1, 3, 3).copy()
    mean = mean.copy()
    cov[3] = np.diag([1e-12, 1.0, 1.0])            # cond 1e12
    cov[7] = np.diag([100.0, 100.0, 100.0])        # covers the whole map
    cov[11] = np.diag([1.0, -1.0, 1.0])            # indefinite
    cov[13] = np.zeros((3, 3))                     # singular
    cov[17] = np.array([[1, 2, 0], [0.5, 1, 0], [0, 0, 1.0]])  # asymmetric
    mean[19] = np.nan                              # never selected
    cov[23][0, 0] = np.inf
    mean = np.concatenate([mean, mean[100:140]])   # duplicates (ties)
    cov = np.concatenate([cov, cov[100:140]]).reshape(-1, 9)
    g = api.GMM(ctx, mean, cov)
    rng = np.random.default_rng(5)
    pts = np.concatenate([pts0, mean[100:140], rng.uniform(-20, 20, (3000, 3))])
    with np.errstate(all="ignore"):
        (i1, d1), (i2, d2) = _both(torch, g, pts)
    assert np.array_equal(i1, i2)
    assert np.array_equal(d1, d2, equal_nan=True)


def test_cell_index_single_component_and_tiny_maps(gpu):
    torch, ctx = gpu
    for K in (1, 2, 5):
        mean, cov = synth.synth_gmm(K, 40 + K)
        g = api.GMM(ctx, mean, cov)
        pts = np.concatenate([synth.synth_points(mean, cov, 500, K), np.random.default_rng(K).uniform(-9, 9, (500, 3))])
        (i1, d1), (i2, d2) = _both(torch, g, pts)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


def test_kshard_association_is_bit_identical(gpu, map_v2):
    """K-sharding: the chi2 of a (point, Gaussian) pair does not depend on the rest of the map, so
    associating against the shards and merging (value, then lowest global index) reproduces the unsharded
    result exactly - through the cell index of each shard as well as through the sweep."""
    torch, ctx = gpu
    from gmmloc_amd import replay
    mean, cov = map_v2
    rng = np.random.default_rng(21)
    pts = mean[rng.integers(0, mean.shape[0], 5000)] + rng.normal(0, 0.08, (5000, 3))
    t = torch.from_numpy(pts).cuda()
    full = api.GMM(ctx, mean, cov)
    for mode in (api.ASSOC_BRUTE, api.ASSOC_EXHAUSTIVE):
        idx_f, d2_f = full.associate3d(t, mode)
        world = 3
        parts = []
        for r in range(world):
            k0, kr = replay.shard_components(mean.shape[0], r, world)
            g = api.GMM(ctx, mean[k0:k0 + kr], cov[k0:k0 + kr])
            idx, d2 = g.associate3d(t, mode)
            parts.append((idx.to(torch.int64) + k0, d2))
        d2m = torch.stack([p[1] for p in parts]).min(0).values
        gi = torch.stack([torch.where(p[1] == d2m, p[0], torch.full_like(p[0], replay.INDEX_NONE)) for p in parts]).min(0).values
        assert torch.equal(d2m, d2_f) and torch.equal(gi, idx_f.to(torch.int64))
        # the library-side merge on one rank is the identity
        gi1, d21 = replay.merge_sharded_association(d2_f, idx_f, 0)
        assert torch.equal(gi1, idx_f.to(torch.int64)) and torch.equal(d21, d2_f)


def test_association_full_stress_size_properties(gpu):
    """BASELINE configs[4] shape, 50 000 points x 65 536 Gaussians (3.3 G pairs; the oracle would need minutes):
    (1) the cell index and the plain sweep agree bit for bit; (2) known answer: a point on a component's mean has
    chi2 exactly 0 there, so the argmin is that component (or a lower-index one that also gives 0); (3) permuting
    the points permutes the result; (4) every reported chi2 is the canonical expression of its own pair."""
    torch, ctx = gpu
    K, N = 65536, 50000
    mean, cov = synth.synth_gmm(K, 7)
    g = api.GMM(ctx, mean, cov)
    assert g.index_info()["enabled"]
    rng = np.random.default_rng(8)
    comp = rng.integers(0, K, N)
    pts = synth.synth_points(mean, cov, N, seed=9)
    pts[: N // 5] = mean[comp[: N // 5]]  # a fifth of the points exactly on a mean
    t = torch.from_numpy(pts).cuda()
    idx_i, d2_i = g.associate3d(t, api.ASSOC_BRUTE)
    idx_x, d2_x = g.associate3d(t, api.ASSOC_EXHAUSTIVE)
    assert torch.equal(idx_i, idx_x) and torch.equal(d2_i, d2_x)
    ii, dd = idx_x.cpu().numpy(), d2_x.cpu().numpy()
    on = slice(0, N // 5)
    assert (dd[on] == 0.0).all() and (ii[on] <= comp[on]).all()
    perm = torch.from_numpy(rng.permutation(N)).cuda()
    idx_p, d2_p = g.associate3d(t[perm].contiguous(), api.ASSOC_BRUTE)
    assert torch.equal(idx_p, idx_x[perm]) and torch.equal(d2_p, d2_x[perm])
    # (4) re-evaluate the winning pair of 4 096 points on the host with the library's own records:
    #     r = (x - mu)^T L, chi2 = r.r in extended precision (agreement to the cancellation level of r, not to the ulp)
    sub = rng.integers(0, N, 4096)
    A = g.get(api.F_SQRT_INFO).reshape(K, 3, 3)[ii[sub]] if hasattr(api, "F_SQRT_INFO") else None
    if A is not None:
        d = (pts[sub] - mean[ii[sub]]).astype(np.longdouble)
        r = np.einsum("ni,nij->nj", d, A.astype(np.longdouble))
        ref = (r * r).sum(1).astype(np.float64)
        assert np.allclose(dd[sub], ref, rtol=1e-7, atol=1e-9)
