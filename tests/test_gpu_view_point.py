"""GPU parity: renderView + searchCorrespondence (A3-A5), optimizePoint (B1),
checkMapAssociation (A8), optimizeTriangulationVec (B2) vs the oracle."""
import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import synth, api

pytestmark = pytest.mark.gpu


def _poses(gt, n, step=61):
    return np.stack([synth.gt_row_to_Tcw(gt[(7 + i * step) % gt.shape[0]]) for i in range(n)])


@pytest.mark.parametrize("threads", ["256", "1024"])  # the batch / the few-views block shape
@pytest.mark.parametrize("mapname,seq", [("v1", "V1_01_easy"), ("v1", "V1_03_difficult"), ("v2", "V2_02_medium")])
def test_search2d_matches_oracle(gpu, oracle, map_v1, map_v2, gt_sync, opt, mapname, seq, threads):
    torch, ctx = gpu
    opt("view_threads", int(threads))
    mean, cov = map_v1 if mapname == "v1" else map_v2
    cam = api.Camera()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    B, N = 6, 600
    poses = _poses(gt_sync[seq], B)
    rng = np.random.default_rng(3)
    uv = np.stack([rng.uniform(0, 752, (B, N)), rng.uniform(0, 480, (B, N))], 2)
    nfeat = np.array([N, N - 7, N, 1, N, N], np.int32)
    cand, ncand, vids, nview = g.search2d(cam, torch.from_numpy(poses).cuda(), torch.from_numpy(uv).cuda(),
                                          torch.from_numpy(nfeat).cuda(), k=5, view_cap=2048)
    torch.cuda.synchronize()
    cand, ncand, vids, nview = cand.cpu().numpy(), ncand.cpu().numpy(), vids.cpu().numpy(), nview.cpu().numpy()
    tot = 0
    for b in range(B):
        ids, m2, c2, dep = oracle.render_view(h, cam, poses[b])
        assert nview[b] == len(ids), (b, nview[b], len(ids))
        assert np.array_equal(vids[b][:len(ids)], ids)  # same set AND same (depth-sorted) order
        assert (vids[b][len(ids):] == -1).all()
        c_ref, n_ref = oracle.search_correspondence(h, uv[b], 5)
        nf = nfeat[b]
        assert np.array_equal(ncand[b][:nf], n_ref[:nf])
        assert np.array_equal(cand[b][:nf], c_ref[:nf])
        assert (ncand[b][nf:] == 0).all() and (cand[b][nf:] == -1).all()
        tot += len(ids)
    assert tot > 100 * B / 2  # views are not trivially empty
    oracle.gmm_destroy(h)


def _check_search2d(torch, ctx, oracle, mean, cov, poses, uv, k):
    cam = api.Camera()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    B = poses.shape[0]
    cand, ncand, vids, nview = g.search2d(cam, torch.from_numpy(poses).cuda(), torch.from_numpy(uv).cuda(), None, k=k,
                                          view_cap=4096)
    torch.cuda.synchronize()
    cand, ncand, vids, nview = cand.cpu().numpy(), ncand.cpu().numpy(), vids.cpu().numpy(), nview.cpu().numpy()
    tot = 0
    for b in range(B):
        ids, m2, c2, dep = oracle.render_view(h, cam, poses[b])
        assert nview[b] == len(ids), (b, nview[b], len(ids))
        assert np.array_equal(vids[b][:len(ids)], ids)
        c_ref, n_ref = oracle.search_correspondence(h, uv[b], k)
        assert np.array_equal(ncand[b], n_ref)
        assert np.array_equal(cand[b], c_ref)
        tot += len(ids)
    oracle.gmm_destroy(h)
    return tot / B


@pytest.mark.parametrize("k", [1, 3, 8])
def test_search2d_knn_sizes(gpu, oracle, map_v1, gt_sync, k):
    """searchCorrespondence with k != 5 (the run-time-k instance of the selection network)."""
    torch, ctx = gpu
    mean, cov = map_v1
    poses = _poses(gt_sync["V1_02_medium"], 4, step=97)
    rng = np.random.default_rng(10 + k)
    uv = np.stack([rng.uniform(0, 752, (4, 333)), rng.uniform(0, 480, (4, 333))], 2)
    assert _check_search2d(torch, ctx, oracle, mean, cov, poses, uv, k) > 50


@pytest.mark.parametrize("threads,slot_lds", [("256", None), ("1024", None), ("256", "24"), ("1024", "24")])
def test_search2d_interacting_candidates(gpu, oracle, map_v1, gt_sync, opt, threads, slot_lds):
    """Every component followed by a jittered copy: consecutive candidates fall below the merge threshold of each
    other, replace each other's slots and share old argmins - the ordered part of the merge rounds, in both
    block shapes, with the accepted list in LDS and spilled to the global scratch."""
    torch, ctx = gpu
    opt("view_threads", int(threads))
    if slot_lds:
        opt("view_slot_lds", int(slot_lds))
    mean, cov = map_v1
    rng = np.random.default_rng(77)
    K = mean.shape[0]
    cov = cov.reshape(K, 3, 3)
    m2 = np.empty((3 * K, 3)); c2 = np.empty((3 * K, 3, 3))
    m2[0::3] = mean; m2[1::3] = mean + rng.normal(0, 0.02, (K, 3)); m2[2::3] = mean + rng.normal(0, 0.05, (K, 3))
    c2[0::3] = cov; c2[1::3] = cov * rng.uniform(0.7, 1.4, (K, 1, 1)); c2[2::3] = cov * rng.uniform(0.5, 2.0, (K, 1, 1))
    poses = _poses(gt_sync["V1_01_easy"], 5, step=131)
    uv = np.stack([rng.uniform(0, 752, (5, 200)), rng.uniform(0, 480, (5, 200))], 2)
    assert _check_search2d(torch, ctx, oracle, m2, c2, poses, uv, 5) > 50


@pytest.mark.parametrize("threads", ["256", "1024"])
def test_search2d_needle_fan(gpu, oracle, opt, threads):
    """160 thin components fanned around one centre: every pair has Mahalanobis part 0 (nothing is screened
    out, the pair list of a round overflows and the exhaustive path runs) while the log part keeps most of
    them apart."""
    torch, ctx = gpu
    opt("view_threads", int(threads))
    n = 160
    mean = np.tile(np.array([[0.0, 0.0, 3.0]]), (n, 1)) + np.random.default_rng(5).normal(0, 1e-4, (n, 3))
    cov = np.empty((n, 3, 3))
    for i in range(n):
        a = np.pi * i / n
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        cov[i] = R @ np.diag([0.5 ** 2, 0.002 ** 2, 0.0005 ** 2]) @ R.T
    poses = np.array([[0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0.1, -0.05, 0.2]], np.float64)  # qx qy qz qw tx ty tz (Tcw)
    rng = np.random.default_rng(6)
    uv = np.stack([rng.uniform(250, 500, (2, 300)), rng.uniform(150, 330, (2, 300))], 2)
    assert _check_search2d(torch, ctx, oracle, mean, cov, poses, uv, 5) > 100


def _frames(mean, cov, gt, cam, B, M, seed, **kw):
    return [synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(gt[(11 + i * 43) % gt.shape[0]]), cam, M, seed + i, **kw)
            for i in range(B)]


def test_optimize_point_matches_oracle(gpu, oracle, map_v1, gt_sync):
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    flags = g.get(api.F_FLAGS)
    deg = np.nonzero(flags & 1)[0]
    f = _frames(mean, cov, gt_sync["V1_01_easy"], cam, 1, 500, 5, mono_frac=0.0, outlier_frac=0.2)[0]
    N = 500
    comp = f["comp"].copy()
    comp[(flags[comp] & 1) == 0] = deg[3]
    pose = np.tile(f["pose_gt"], (N, 1))
    R, t = synth.quat_to_R(f["pose_gt"][:4]), f["pose_gt"][4:]
    pz2 = np.minimum(1.0, (f["Xw"] @ R.T + t)[:, 2]) ** 2
    X0 = f["Xw"] + np.random.default_rng(1).standard_normal((N, 3)) * 0.03
    r_ref = oracle.optimize_point(h, cam, X0, f["obs"], f["octave"], pose, comp, pz2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    res, c2p, c2s, est = gmmloc_amd.api.optimize_point(ctx, g, cam, prm, T(X0), T(f["obs"]), T(f["octave"]), T(pose),
                                                       T(comp.astype(np.int32)), T(pz2))
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), r_ref[0])
    np.testing.assert_allclose(est.cpu().numpy(), r_ref[3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(c2p.cpu().numpy(), r_ref[1], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(c2s.cpu().numpy(), r_ref[2], rtol=1e-8, atol=1e-10)
    assert 0 < r_ref[0].sum() < N
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("mapname", ["v1", "v2"])
def test_check_map_association_matches_oracle(gpu, oracle, map_v1, map_v2, gt_sync, mapname):
    """Full key-frame association chain: search2d -> checkMapAssociation."""
    torch, ctx = gpu
    mean, cov = map_v1 if mapname == "v1" else map_v2
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    B, N = 3, 700
    fr = _frames(mean, cov, gt_sync["V1_02_medium" if mapname == "v1" else "V2_01_easy"], cam, B, N, 50,
                 mono_frac=0.0, outlier_frac=0.1)
    poses = np.stack([f["pose_gt"] for f in fr])
    # stereo feature -> unprojected map point (frame.cpp:27-35), perturbed in depth
    pts = np.stack([f["Xw"] for f in fr]) + np.random.default_rng(2).standard_normal((B, N, 3)) * 0.02
    uvr = np.stack([f["obs"] for f in fr])
    octv = np.stack([f["octave"] for f in fr])
    octv[0, ::9] = -1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cand, ncand, _, _ = g.search2d(cam, T(poses), T(uvr[:, :, :2].copy()), None, k=5)
    pts_d = T(pts)
    out = gmmloc_amd.api.check_map_association(ctx, g, cam, prm, T(poses), pts_d, T(uvr), T(octv), cand, ncand)
    torch.cuda.synchronize()
    out, pts_o, cand, ncand = out.cpu().numpy(), pts_d.cpu().numpy(), cand.cpu().numpy(), ncand.cpu().numpy()
    n_assoc = 0
    for b in range(B):
        oracle.render_view(h, cam, poses[b])
        c_ref, n_ref = oracle.search_correspondence(h, uvr[b][:, :2].copy(), 5)
        assert np.array_equal(cand[b], c_ref) and np.array_equal(ncand[b], n_ref)
        keep = octv[b] >= 0
        o_ref, p_ref = oracle.check_map_association(h, cam, poses[b], pts[b][keep], uvr[b][keep],
                                                    octv[b][keep], c_ref[keep], n_ref[keep])
        assert np.array_equal(out[b][keep], o_ref), int((out[b][keep] != o_ref).sum())
        np.testing.assert_allclose(pts_o[b][keep], p_ref, rtol=0, atol=1e-9)
        assert (out[b][~keep] == -1).all() and np.array_equal(pts_o[b][~keep], pts[b][~keep])
        n_assoc += int((o_ref >= 0).sum())
    assert n_assoc > 50
    oracle.gmm_destroy(h)


def test_optimize_triangulation_matches_oracle(gpu, oracle, map_v1, gt_sync):
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    gt = gt_sync["V1_01_easy"]
    N = 400
    p1 = synth.gt_row_to_Tcw(gt[900])
    p2 = synth.gt_row_to_Tcw(gt[912])
    f = synth.synth_frame(mean, cov, p1, cam, N, 9, outlier_frac=0.05, mono_frac=0.5)
    rng = np.random.default_rng(4)
    R2, t2 = synth.quat_to_R(p2[:4]), p2[4:]
    pc2 = f["Xw"] @ R2.T + t2
    u2 = cam.fx * pc2[:, 0] / pc2[:, 2] + cam.cx + rng.standard_normal(N) * 0.7
    v2 = cam.fy * pc2[:, 1] / pc2[:, 2] + cam.cy + rng.standard_normal(N) * 0.7
    ur2 = np.where(rng.uniform(size=N) < 0.5, -1.0, u2 - cam.bf / pc2[:, 2])
    uvr2 = np.stack([u2, v2, ur2], 1)
    oct2 = rng.integers(0, 8, N).astype(np.int32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    poses = np.stack([p1, p2])
    uv = np.stack([f["obs"][:, :2], uvr2[:, :2]])
    cand, ncand, _, _ = g.search2d(cam, T(poses), T(uv), None, k=5)
    torch.cuda.synchronize()
    c, n = cand.cpu().numpy(), ncand.cpu().numpy()
    x0 = f["Xw"] + rng.standard_normal((N, 3)) * 0.03
    xd = T(x0)
    out = gmmloc_amd.api.optimize_triangulation(ctx, g, cam, prm, xd, T(np.tile(p1, (N, 1))), T(f["obs"]),
                                                T(f["octave"]), T(np.tile(p2, (N, 1))), T(uvr2), T(oct2),
                                                T(c[0]), T(n[0]), T(c[1]), T(n[1]))
    torch.cuda.synchronize()
    o_ref, x_ref = oracle.optimize_triangulation(h, cam, x0, np.tile(p1, (N, 1)), f["obs"], f["octave"],
                                                 np.tile(p2, (N, 1)), uvr2, oct2, c[0], n[0], c[1], n[1])
    assert np.array_equal(out.cpu().numpy(), o_ref), int((out.cpu().numpy() != o_ref).sum())
    np.testing.assert_allclose(xd.cpu().numpy(), x_ref, rtol=0, atol=1e-9)
    assert (o_ref >= 0).sum() > 20
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("ia,ib,seed", [(900, 915, 5), (300, 306, 6), (1500, 1530, 7), (2000, 2012, 8)])
def test_create_map_points_matches_oracle(gpu, oracle, map_v1, gt_sync, ia, ib, seed):
    """createMapPoints per-match block (parallax test, SVD triangulation / stereo unprojection, B2, checks)."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    gt = gt_sync["V1_01_easy"]
    N = 600
    m = synth.synth_tri_matches(mean, cov, synth.gt_row_to_Tcw(gt[ia]), synth.gt_row_to_Tcw(gt[ib]), cam, N, seed)
    x_ref, t_ref, c_ref = oracle.create_map_points(h, cam, **m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    keys = ("pose1", "uvr1", "depth1", "oct1", "pose2", "uvr2", "depth2", "oct2", "cand1", "n1", "cand2", "n2")
    x, t, c = gmmloc_amd.api.create_map_points(ctx, g, cam, prm, *[T(m[k]) for k in keys])
    torch.cuda.synchronize()
    x, t, c = x.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy()
    assert np.array_equal(t, t_ref), int((t != t_ref).sum())
    assert np.array_equal(c, c_ref), int((c != c_ref).sum())
    # a handful of matches triangulate to points kilometres away (parallel rays that still pass every
    # check): Gauss-Newton is chaotic there and only the decision, not the coordinates, is comparable
    sane = np.linalg.norm(x_ref, axis=1) < 100.0
    assert sane.mean() > 0.98
    np.testing.assert_allclose(x[sane], x_ref[sane], rtol=0, atol=1e-8)
    assert (t_ref > 0).sum() > 50 and (t_ref == 0).sum() > 20 and len(set(t_ref.tolist())) >= 4
    oracle.gmm_destroy(h)


@pytest.mark.parametrize("K", [1, 2, 5, 17])
def test_search2d_tiny_maps(gpu, oracle, K):
    """Fewer components than a merge round holds, views with zero or one rendered component, features = 0."""
    torch, ctx = gpu
    rng = np.random.default_rng(40 + K)
    mean = np.array([0.0, 0.0, 3.0]) + rng.normal(0, 0.6, (K, 3))
    cov = np.empty((K, 3, 3))
    for i in range(K):
        A = rng.normal(0, 1, (3, 3))
        Q, _ = np.linalg.qr(A)
        cov[i] = Q @ np.diag([rng.uniform(1e-6, 1e-4), rng.uniform(0.01, 0.2), rng.uniform(0.01, 0.5)]) @ Q.T
        cov[i] = 0.5 * (cov[i] + cov[i].T)
    poses = np.array([[0, 0, 0, 1, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0.3, 0.2, -0.5]], np.float64)  # 2nd looks away
    uv = np.stack([rng.uniform(0, 752, (3, 50)), rng.uniform(0, 480, (3, 50))], 2)
    _check_search2d(torch, ctx, oracle, mean, cov, poses, uv, 5)
    # no features at all: only the rendered list is produced
    g = api.GMM(ctx, mean, cov)
    cand, ncand, vids, nview = g.search2d(api.Camera(), torch.from_numpy(poses).cuda(),
                                          torch.zeros((3, 0, 2), dtype=torch.float64).cuda(), None, k=5, view_cap=32)
    torch.cuda.synchronize()
    h = oracle.gmm_create(mean, cov)
    for b in range(3):
        ids, _, _, _ = oracle.render_view(h, api.Camera(), poses[b])
        assert int(nview[b]) == len(ids) and np.array_equal(vids[b].cpu().numpy()[:len(ids)], ids)
    oracle.gmm_destroy(h)
