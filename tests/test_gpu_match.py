"""GPU parity: gl_search_by_projection (ORBmatcher::searchByProjection, orb_matcher.cpp:27-110) vs the
oracle's sequential restatement.  Integer work: indices and counts must be bit-exact, including the
order-dependent hand-over of features between map points."""
import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import synth, api

pytestmark = pytest.mark.gpu

KEYS = ("feat_uv", "feat_ur", "feat_oct", "feat_desc", "feat_taken", "mp_uvr", "mp_level", "mp_viewcos", "mp_valid",
        "mp_desc")


def run_gpu(torch, ctx, frames, th, nn_ratio=0.8):
    cam = api.Camera()
    cam.width, cam.height = frames[0]["width"], frames[0]["height"]
    T = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in frames]))).cuda()
    a = {k: T(k) for k in KEYS}
    a["mp_level"] = a["mp_level"].to(torch.int32)
    m, n = api.search_by_projection(ctx, cam, *[a[k] for k in KEYS], th=th, nn_ratio=nn_ratio)
    torch.cuda.synchronize()
    return m.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("NF,NP,th,dup", [(300, 200, 3.0, 0.25), (1200, 800, 3.0, 0.25), (2000, 2500, 5.0, 0.5),
                                          (64, 4000, 3.0, 0.9), (1000, 1000, 1.0, 0.3), (1, 1, 3.0, 0.0)])
@pytest.mark.parametrize("shape", ["0", "1"])  # batch shape (512 threads) / few-frames shape (1 024, descriptors in LDS)
@pytest.mark.parametrize("fuv", [True, False])  # key-point coordinates that are float values (the reference's: the 32-bit window walk) / arbitrary doubles (general walk)
def test_search_by_projection_matches_oracle(gpu, oracle, opt, NF, NP, th, dup, shape, fuv):
    torch, ctx = gpu
    opt("match_desc_lds", int(shape))
    frames = [synth.synth_match_frame(NF, NP, 1000 * NF + 7 * b, dup_frac=dup, float_uv=fuv) for b in range(5)]
    m, n = run_gpu(torch, ctx, frames, th)
    tot = 0
    for b, f in enumerate(frames):
        m_ref, n_ref = oracle.search_by_projection(th=th, **f)
        assert n[b] == n_ref, (b, int(n[b]), n_ref)
        assert np.array_equal(m[b], m_ref), (b, int((m[b] != m_ref).sum()))
        tot += n_ref
    assert tot > 0 or NF == 1


@pytest.mark.parametrize("fuv", [True, False])
def test_search_by_projection_conflict_chain(gpu, oracle, fuv):
    """Worst case for the fixed-point iteration: every map point wants the same features, so each one
    only settles after all earlier ones have (as many rounds as map points).  Round 5: every map point has all 40 features as
    candidates and keeps only its three best of round 1 - from round 3 on the later ones have lost those and walk again."""
    torch, ctx = gpu
    rng = np.random.default_rng(3)
    NF, NP = 40, 60
    desc0 = rng.integers(0, 256, 32, dtype=np.uint8)
    feat_desc = np.tile(desc0, (NF, 1))
    for i in range(NF):  # feature i differs from the common descriptor in i bits: strict preference order
        bits = rng.choice(256, i, replace=False)
        np.bitwise_xor.at(feat_desc[i], bits // 8, (1 << (bits % 8)).astype(np.uint8))
    f = dict(width=752, height=480, feat_uv=np.tile([[300.0, 200.0]], (NF, 1)) + rng.uniform(-3, 3, (NF, 2)),
             feat_ur=-np.ones(NF, np.float32), feat_oct=np.zeros(NF, np.int32), feat_desc=feat_desc,
             feat_taken=np.zeros(NF, np.uint8), mp_uvr=np.tile([[300.0, 200.0, 250.0]], (NP, 1)),
             mp_level=np.zeros(NP), mp_viewcos=np.full(NP, 0.5), mp_valid=np.ones(NP, np.uint8),
             mp_desc=np.tile(desc0, (NP, 1)))
    if fuv:
        f["feat_uv"] = f["feat_uv"].astype(np.float32).astype(np.float64)
    m, n = run_gpu(torch, ctx, [f], 3.0, nn_ratio=1.1)  # ratio test off: pure hand-over
    m_ref, n_ref = oracle.search_by_projection(th=3.0, nn_ratio=1.1, **f)
    assert n_ref == NF and n[0] == n_ref and np.array_equal(m[0], m_ref)
    assert m_ref.tolist() == list(range(NF))  # map point k ends up with its k-th choice


def test_search_by_projection_degenerate_inputs(gpu, oracle):
    torch, ctx = gpu
    f = synth.synth_match_frame(500, 400, 9)
    g = dict(f)
    g["mp_valid"] = np.zeros_like(f["mp_valid"])            # nothing in view
    h = dict(f)
    h["feat_taken"] = np.ones_like(f["feat_taken"])          # every feature already has a map point
    k = dict(f)
    k["mp_uvr"] = f["mp_uvr"] + 5000.0                       # all projections far outside the image
    l = dict(f)
    l["feat_oct"] = -np.ones_like(f["feat_oct"])             # no features at all
    frames = [f, g, h, k, l]
    m, n = run_gpu(torch, ctx, frames, 3.0)
    for b, fr in enumerate(frames):
        m_ref, n_ref = oracle.search_by_projection(th=3.0, **fr)
        assert n[b] == n_ref and np.array_equal(m[b], m_ref), b
    assert n[0] > 0 and (n[1:] == 0).all()


FKEYS = ("pose_cw", "pose_lw", "feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "last_pt",
         "last_valid", "last_oct", "last_angle", "last_desc")


class CamF:  # cfg/v1.yaml intrinsics as the float config scalars (config.h:38-48)
    f32 = staticmethod(lambda x: float(np.float32(x)))
    fx = fy = f32.__func__(435.2046959714599)
    cx = f32.__func__(367.4517211914062)
    cy = f32.__func__(252.2008514404297)
    bf = f32.__func__(47.90639384423901)
    width, height = 752, 480


def run_gpu_frame(torch, ctx, frames, th, mono=False, chk=True):
    cam = api.Camera()
    T = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in frames]))).cuda()
    a = [T(k) for k in FKEYS]
    m, n = api.search_by_projection_frame(ctx, cam, *a, th=th, mono=mono, check_orientation=chk)
    torch.cuda.synchronize()
    return m.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("NF,NL,th,motion,mono,chk", [(300, 250, 7.0, "none", False, True), (1200, 900, 7.0, "forward", False, True),
                                                     (1500, 1200, 14.0, "backward", False, True),
                                                     (1000, 1000, 7.0, "forward", True, True),
                                                     (800, 800, 7.0, "none", False, False), (2000, 3000, 14.0, "none", False, True)])
def test_search_by_projection_frame_matches_oracle(gpu, oracle, NF, NL, th, motion, mono, chk):
    torch, ctx = gpu
    c = api.Camera()
    assert (c.fx, c.cx, c.bf, c.width, c.height) == (CamF.fx, CamF.cx, CamF.bf, CamF.width, CamF.height)
    frames = [synth.synth_motion_frames(NF, NL, 77 * NF + b, CamF, motion, float_uv=b % 2 == 0) for b in range(4)]  # both walks
    m, n = run_gpu_frame(torch, ctx, frames, th, mono, chk)
    tot = 0
    for b, f in enumerate(frames):
        m_ref, n_ref = oracle.search_by_projection_frame(CamF, th=th, mono=mono, check_orientation=chk, **f)
        assert n[b] == n_ref, (b, int(n[b]), n_ref)
        assert np.array_equal(m[b], m_ref), (b, int((m[b] != m_ref).sum()))
        tot += n_ref
    assert tot > 50


def _pack_pairs(torch, pairs):
    """list of synth_tri_search_pair dicts -> the batched CUDA tensors of api.search_for_triangulation (strides = the maxima)"""
    N1 = max(len(p["kf1"]["oct"]) for p in pairs)
    N2 = max(len(p["kf2"]["oct"]) for p in pairs)
    out = []
    for key, N in (("kf1", N1), ("kf2", N2)):
        NN = max(len(p[key]["node_id"]) for p in pairs)
        B = len(pairs)
        t = dict(uv=np.zeros((B, N, 2)), ur=np.full((B, N), -1.0, np.float32), oct=np.full((B, N), -1, np.int32), angle=np.zeros((B, N), np.float32),
                 desc=np.zeros((B, N, 32), np.uint8), has_mp=np.zeros((B, N), np.uint8), nnode=np.zeros(B, np.int32),
                 node_id=np.zeros((B, NN), np.int32), node_ptr=np.zeros((B, NN + 1), np.int32), node_idx=np.zeros((B, N), np.int32))
        for b, p in enumerate(pairs):
            k = p[key]
            n, nn = len(k["oct"]), len(k["node_id"])
            for name in ("uv", "ur", "oct", "angle", "desc", "has_mp"):
                t[name][b, :n] = k[name]
            t["nnode"][b] = nn
            t["node_id"][b, :nn] = k["node_id"]
            t["node_ptr"][b, :nn + 1] = k["node_ptr"]
            t["node_ptr"][b, nn + 1:] = k["node_ptr"][-1]
            t["node_idx"][b, :len(k["node_idx"])] = k["node_idx"]
        out.append({name: torch.from_numpy(v).cuda() for name, v in t.items()})
    fm = torch.from_numpy(np.stack([p["fmat"].reshape(9) for p in pairs])).cuda()
    ep = torch.from_numpy(np.stack([p["epipole"] for p in pairs]).astype(np.float32)).cuda()
    return out[0], out[1], fm, ep


@pytest.mark.parametrize("only_stereo,check_orientation", [(False, True), (True, True), (False, False)])
def test_search_for_triangulation_matches_oracle(gpu, oracle, only_stereo, check_orientation):
    """gl_search_for_triangulation (ORBmatcher::searchForTriangulation, orb_matcher.cpp:141-293) against the sequential oracle:
    matches12 bit for bit - crowded nodes with rival features (the order-dependent hand-over), equal descriptor distances
    (last one wins), mono / stereo mixes, padding slots, pairs of different sizes in one batch, the rotation histogram."""
    torch, ctx = gpu
    cam = api.Camera()
    pairs = [synth.synth_tri_search_pair(N1, N2, 400 + i, cam, n_nodes=nodes, pad=1)
             for i, (N1, N2, nodes) in enumerate(((300, 350, 60), (1200, 1100, 200), (700, 900, 25), (64, 70, 5), (2000, 1900, 300), (500, 40, 80)))]
    k1, k2, fm, ep = _pack_pairs(torch, pairs)
    match, nm = api.search_for_triangulation(ctx, k1, k2, fm, ep, only_stereo, check_orientation)
    torch.cuda.synchronize()
    match, nm = match.cpu().numpy(), nm.cpu().numpy()
    total = 0
    for b, p in enumerate(pairs):
        m_ref, n_ref = oracle.search_for_triangulation(p["kf1"], p["kf2"], p["fmat"], p["epipole"], only_stereo, check_orientation)
        n1 = len(m_ref)
        assert np.array_equal(match[b, :n1], m_ref), (b, int((match[b, :n1] != m_ref).sum()))
        assert (match[b, n1:] == -1).all() and nm[b] == n_ref
        total += n_ref
    assert total > 100


def test_search_for_triangulation_soak(gpu, oracle):
    """200 random key-frame pairs (sizes, node counts, stereo fractions): every match equal to the oracle's."""
    torch, ctx = gpu
    cam = api.Camera()
    rng = np.random.default_rng(77)
    pairs = [synth.synth_tri_search_pair(int(rng.integers(20, 1400)), int(rng.integers(20, 1400)), 5000 + i, cam, n_nodes=int(rng.integers(3, 250)),
                                         only_stereo_frac=float(rng.uniform(0.0, 1.0)), pad=int(rng.integers(0, 2))) for i in range(200)]
    k1, k2, fm, ep = _pack_pairs(torch, pairs)
    checked = 0
    for only_stereo, chk in ((False, True), (True, False)):
        match, nm = api.search_for_triangulation(ctx, k1, k2, fm, ep, only_stereo, chk)
        torch.cuda.synchronize()
        match, nm = match.cpu().numpy(), nm.cpu().numpy()
        for b, p in enumerate(pairs):
            m_ref, n_ref = oracle.search_for_triangulation(p["kf1"], p["kf2"], p["fmat"], p["epipole"], only_stereo, chk)
            assert np.array_equal(match[b, :len(m_ref)], m_ref) and nm[b] == n_ref, (b, only_stereo, chk)
            checked += n_ref
    assert checked > 5000


def test_search_gather_create_map_points_chain_on_device(gpu, oracle, map_v1):
    """searchForTriangulation -> matched_pairs -> createMapPoints without leaving the device: gl_search_for_triangulation,
    gl_gather_triangulation_matches, gl_create_map_points on 3 key-frame pairs, against the oracle's matcher + the same gather in
    numpy + the oracle's create_map_points (types and components exact, points 1e-8 where the triangulation is sane)."""
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx, mean, cov)
    h = oracle.gmm_create(mean, cov)
    rng = np.random.default_rng(3)
    pairs = [synth.synth_tri_search_pair(600, 650, 700 + i, cam, n_nodes=80) for i in range(3)]
    K = 5
    sides = []
    for p in pairs:  # depth, candidate components per feature (kf->comps_), the two poses
        for key, pose in (("kf1", p["pose1"]), ("kf2", p["pose2"])):
            k = p[key]
            n = len(k["oct"])
            k["depth"] = np.where(k["ur"] >= 0, cam.bf / np.maximum(k["uv"][:, 0] - k["ur"], 1e-3), -1.0).astype(np.float32)
            k["cand"] = rng.integers(0, mean.shape[0], (n, K)).astype(np.int32)
            k["ncand"] = rng.integers(0, K + 1, n).astype(np.int32)
            k["pose"] = pose
    k1, k2, fm, ep = _pack_pairs(torch, pairs)
    match, nm = api.search_for_triangulation(ctx, k1, k2, fm, ep, False, True)

    def side(key, N):
        B = len(pairs)
        t = dict(pose=np.zeros((B, 7)), uv=np.zeros((B, N, 2)), ur=np.full((B, N), -1.0, np.float32), depth=np.full((B, N), -1.0, np.float32),
                 oct=np.zeros((B, N), np.int32), cand=np.full((B, N, K), -1, np.int32), ncand=np.zeros((B, N), np.int32))
        for b, p in enumerate(pairs):
            k = p[key]
            n = len(k["oct"])
            t["pose"][b] = k["pose"]
            for name in ("uv", "ur", "depth", "oct", "cand", "ncand"):
                t[name][b, :n] = k[name]
        return {q: torch.from_numpy(v).cuda() for q, v in t.items()}
    s1, s2 = side("kf1", match.shape[1]), side("kf2", k2["oct"].shape[1])
    off, m = api.gather_triangulation_matches(ctx, match, nm, s1, s2)
    torch.cuda.synchronize()
    off = off.cpu().numpy()
    total = int(off[-1])
    assert total == int(nm.sum().item()) and total > 100
    keys = ("pose1", "uvr1", "depth1", "oct1", "pose2", "uvr2", "depth2", "oct2", "cand1", "n1", "cand2", "n2")
    x, t, c = api.create_map_points(ctx, g, cam, prm, *[m[q][:total].contiguous() for q in keys])
    torch.cuda.synchronize()
    x, t, c = x.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy()
    pos = 0
    for b, p in enumerate(pairs):  # the reference's chain on the host
        m_ref, n_ref = oracle.search_for_triangulation(p["kf1"], p["kf2"], p["fmat"], p["epipole"], False, True)
        i1 = np.nonzero(m_ref >= 0)[0]
        i2 = m_ref[i1]
        assert off[b] == pos and np.array_equal(m["idx1"][pos:pos + n_ref].cpu().numpy(), i1) and np.array_equal(m["idx2"][pos:pos + n_ref].cpu().numpy(), i2)
        a, bb = p["kf1"], p["kf2"]
        args = dict(pose1=np.tile(a["pose"], (n_ref, 1)), uvr1=np.concatenate([a["uv"][i1], a["ur"][i1, None].astype(np.float64)], 1), depth1=a["depth"][i1],
                    oct1=a["oct"][i1], pose2=np.tile(bb["pose"], (n_ref, 1)), uvr2=np.concatenate([bb["uv"][i2], bb["ur"][i2, None].astype(np.float64)], 1),
                    depth2=bb["depth"][i2], oct2=bb["oct"][i2], cand1=a["cand"][i1], n1=a["ncand"][i1], cand2=bb["cand"][i2], n2=bb["ncand"][i2])
        x_ref, t_ref, c_ref = oracle.create_map_points(h, cam, **args)
        assert np.array_equal(t[pos:pos + n_ref], t_ref) and np.array_equal(c[pos:pos + n_ref], c_ref), b
        sane = (t_ref > 0) & (np.linalg.norm(x_ref, axis=1) < 100.0)
        if sane.any():
            np.testing.assert_allclose(x[pos:pos + n_ref][sane], x_ref[sane], rtol=0, atol=1e-8)
        pos += n_ref
    oracle.gmm_destroy(h)


def _pack_bow(torch, pairs):
    """list of (kf, fr) dicts -> the batched CUDA tensors of api.search_by_bow (strides = the maxima, padding slots: no map point /
    not in any list)"""
    out = []
    for side in (0, 1):
        N = max(len(p[side]["angle"]) for p in pairs)
        NN = max(len(p[side]["node_id"]) for p in pairs)
        B = len(pairs)
        t = dict(angle=np.zeros((B, N), np.float32), desc=np.zeros((B, N, 32), np.uint8), has_mp=np.zeros((B, N), np.uint8),
                 nnode=np.zeros(B, np.int32), node_id=np.zeros((B, NN), np.int32), node_ptr=np.zeros((B, NN + 1), np.int32),
                 node_idx=np.zeros((B, N), np.int32))
        for b, p in enumerate(pairs):
            k = p[side]
            n, nn = len(k["angle"]), len(k["node_id"])
            t["angle"][b, :n] = k["angle"]
            t["desc"][b, :n] = k["desc"]
            if side == 0:
                t["has_mp"][b, :n] = k["has_mp"]
            t["nnode"][b] = nn
            t["node_id"][b, :nn] = k["node_id"]
            t["node_ptr"][b, :nn + 1] = k["node_ptr"]
            t["node_ptr"][b, nn + 1:] = k["node_ptr"][-1]
            t["node_idx"][b, :len(k["node_idx"])] = k["node_idx"]
        out.append({name: torch.from_numpy(v).cuda() for name, v in t.items()})
    return out[0], out[1]


@pytest.mark.parametrize("nn_ratio,check_orientation", [(0.7, True), (0.9, True), (0.6, False)])
def test_search_by_bow_matches_oracle(gpu, oracle, nn_ratio, check_orientation):
    """gl_search_by_bow (ORBmatcher::searchByBoW, orb_matcher.cpp:295-408) against the sequential oracle: matches bit for bit -
    crowded nodes with rival key-frame features (the order-dependent hand-over, which also changes a later feature's second-best
    distance), equal best distances (the ratio test fails), features without a map point, pairs of different sizes in one batch,
    the rotation histogram."""
    torch, ctx = gpu
    cam = api.Camera()
    pairs = [synth.synth_bow_pair(N1, N2, 600 + i, cam, n_nodes=nodes)
             for i, (N1, N2, nodes) in enumerate(((300, 350, 60), (1200, 1100, 200), (700, 900, 25), (64, 70, 5), (2000, 1900, 300), (500, 40, 80)))]
    kf, fr = _pack_bow(torch, pairs)
    match, nm = api.search_by_bow(ctx, kf, fr, nn_ratio, check_orientation)
    torch.cuda.synchronize()
    match, nm = match.cpu().numpy(), nm.cpu().numpy()
    total = 0
    for b, p in enumerate(pairs):
        m_ref, n_ref = oracle.search_by_bow(p[0], p[1], nn_ratio, check_orientation)
        n2 = len(m_ref)
        assert np.array_equal(match[b, :n2], m_ref), (b, int((match[b, :n2] != m_ref).sum()))
        assert (match[b, n2:] == -1).all() and nm[b] == n_ref
        total += n_ref
    assert total > 100


def test_search_by_bow_soak(gpu, oracle):
    """200 random key-frame / frame pairs (sizes, node counts, map-point fractions): every match equal to the oracle's."""
    torch, ctx = gpu
    cam = api.Camera()
    rng = np.random.default_rng(78)
    pairs = [synth.synth_bow_pair(int(rng.integers(20, 1400)), int(rng.integers(20, 1400)), 7000 + i, cam, n_nodes=int(rng.integers(3, 250)),
                                  mp_frac=float(rng.uniform(0.1, 1.0))) for i in range(200)]
    kf, fr = _pack_bow(torch, pairs)
    checked = 0
    for ratio, chk in ((0.7, True), (0.85, False)):
        match, nm = api.search_by_bow(ctx, kf, fr, ratio, chk)
        torch.cuda.synchronize()
        match, nm = match.cpu().numpy(), nm.cpu().numpy()
        for b, p in enumerate(pairs):
            m_ref, n_ref = oracle.search_by_bow(p[0], p[1], ratio, chk)
            assert np.array_equal(match[b, :len(m_ref)], m_ref) and nm[b] == n_ref, (b, ratio, chk)
            checked += n_ref
    assert checked > 3000


FUSE_KEYS = ("feat_uv", "feat_ur", "feat_oct", "feat_desc", "mp_uvr", "mp_level", "mp_valid", "mp_desc")


def _pack_fuse(torch, frames):
    NF = max(len(f["feat_oct"]) for f in frames)
    NP = max(len(f["mp_level"]) for f in frames)
    B = len(frames)
    t = dict(feat_uv=np.zeros((B, NF, 2)), feat_ur=np.full((B, NF), -1.0, np.float32), feat_oct=np.full((B, NF), -1, np.int32),
             feat_desc=np.zeros((B, NF, 32), np.uint8), mp_uvr=np.zeros((B, NP, 3)), mp_level=np.zeros((B, NP), np.int32),
             mp_valid=np.zeros((B, NP), np.uint8), mp_desc=np.zeros((B, NP, 32), np.uint8))
    for b, f in enumerate(frames):
        nf, npn = len(f["feat_oct"]), len(f["mp_level"])
        for k in ("feat_uv", "feat_ur", "feat_oct", "feat_desc"):
            t[k][b, :nf] = f[k]
        for k in ("mp_uvr", "mp_level", "mp_valid", "mp_desc"):
            t[k][b, :npn] = f[k]
    return [torch.from_numpy(t[k]).cuda() for k in FUSE_KEYS]


@pytest.mark.parametrize("coords", ["double", "float", "mixed", "float_records_off"])
def test_fuse_search_matches_oracle(gpu, oracle, opt, coords):
    """gl_fuse_search (Localization::fuseObservations, localization.cpp:226-318, the matching half) against the sequential oracle:
    best feature and best distance of every map point bit for bit - crowded windows, equal distances (the first in the grid's
    visiting order), the level band, the chi2 gates, points outside the image, padding slots, key-frames of different sizes in one
    batch; then 120 random key-frames.  coords: feature coordinates that are float values (cv::KeyPoint's: the record walk of round 6),
    arbitrary doubles (the walk from global memory, chosen per key-frame on the device), both in one batch, and the option
    fuse_records = 0."""
    torch, ctx = gpu
    cam = api.Camera()
    cam.width, cam.height = 752, 480
    if coords == "float_records_off":
        opt("fuse_records", 0)
    fc = lambda i: coords.startswith("float") or (coords == "mixed" and i % 2 == 0)
    frames = [synth.synth_fuse_frame(NF, NP, 800 + i, float_coords=fc(i))
              for i, (NF, NP) in enumerate(((300, 260), (1200, 1500), (2000, 3000), (40, 900), (700, 30), (5, 5)))]
    for th in (3.0, 5.0):
        bi, bd = api.fuse_search(ctx, cam, *_pack_fuse(torch, frames), th=th)
        torch.cuda.synchronize()
        bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
        tot = 0
        for b, f in enumerate(frames):
            ri, rd, n = oracle.fuse_search(f["width"], f["height"], *[f[k] for k in FUSE_KEYS], th=th)
            npn = len(ri)
            assert np.array_equal(bi[b, :npn], ri) and np.array_equal(bd[b, :npn], rd), (b, th, int((bi[b, :npn] != ri).sum()))
            assert (bi[b, npn:] == -1).all()
            tot += n
        assert tot > 1000
    rng = np.random.default_rng(79)
    frames = [synth.synth_fuse_frame(int(rng.integers(5, 1800)), int(rng.integers(5, 2200)), 9000 + i, float_coords=fc(i)) for i in range(120)]
    bi, bd = api.fuse_search(ctx, cam, *_pack_fuse(torch, frames), th=3.0)
    torch.cuda.synchronize()
    bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    tot = 0
    for b, f in enumerate(frames):
        ri, rd, n = oracle.fuse_search(f["width"], f["height"], *[f[k] for k in FUSE_KEYS], th=3.0)
        assert np.array_equal(bi[b, :len(ri)], ri) and np.array_equal(bd[b, :len(ri)], rd), b
        tot += n
    assert tot > 20000


def _pack_project(torch, frames):
    B, NP = len(frames), max(len(f["cand"]) for f in frames)
    t = dict(pose_cw=np.zeros((B, 7)), t_wc=np.zeros((B, 3)), pos=np.zeros((B, NP, 3)), normal=np.zeros((B, NP, 3)),
             max_dist=np.zeros((B, NP), np.float32), min_dist=np.zeros((B, NP), np.float32), cand=np.zeros((B, NP), np.uint8))
    for b, f in enumerate(frames):
        n = len(f["cand"])
        t["pose_cw"][b], t["t_wc"][b] = f["pose_cw"], f["t_wc"]
        for k in ("pos", "normal", "max_dist", "min_dist", "cand"):
            t[k][b, :n] = f[k]
    return [torch.from_numpy(t[k]).cuda() for k in ("pose_cw", "t_wc", "pos", "normal", "max_dist", "min_dist", "cand")]


def test_project_map_points_matches_oracle(gpu, oracle):
    """gl_project_map_points (Frame::project3, frame.cpp:98-119 + MapPoint::checkScaleAndVisible, mappoint.cpp:257-303; the loop of
    tracking.cpp:233-256) against the sequential oracle, EVERY output bit for bit (uvr, view_cos, dist as doubles; level; flag):
    frames of different sizes in one batch (padding = non-candidates), points behind the camera / outside the image / outside the
    distance band / seen too obliquely; the known answers; the level at its seven steps (ratios 40 float steps either side of
    1.2f ^ k: the oracle's level is the host libm's, the device compares against the steps the host found with the same libm);
    then 150 random frames; then the output feeds gl_fuse_search as it is."""
    torch, ctx = gpu
    cam = api.Camera()

    def check(frames, scale_factor=1.2):
        out = api.project_map_points(ctx, cam, *_pack_project(torch, frames), scale_factor=scale_factor)
        torch.cuda.synchronize()
        out = [o.cpu().numpy() for o in out]
        tot = 0
        for b, f in enumerate(frames):
            ref = oracle.project_map_points(cam, scale_factor=scale_factor, **f)
            n = len(f["cand"])
            for o, r, name in zip(out, ref[:5], ("uvr", "level", "viewcos", "dist", "inview")):
                assert np.array_equal(o[b, :n], r), (b, name, int((o[b, :n] != r).sum()))
                assert (o[b, n:] == 0).all()
            tot += ref[5]
        return tot, out

    frames = [synth.synth_project_frame(NP, 600 + i, cam) for i, NP in enumerate((400, 3000, 1500, 40, 1, 7000, 2500, 64))]
    assert check(frames)[0] > 3000
    assert check(frames[:3], scale_factor=1.1)[0] > 800
    # known answers + the level steps
    from tests.test_oracle_golden import _project_kat_inputs
    _, pose, pos, nrm, mx, mn, cand = _project_kat_inputs()
    kat = dict(pose_cw=pose, t_wc=np.zeros(3), pos=pos, normal=nrm, max_dist=mx, min_dist=mn, cand=cand)
    r = []
    for k in range(0, 9):
        x = np.float32(np.float32(1.2) ** k)
        for _ in range(40):
            x = np.nextafter(x, np.float32(0))
        for _ in range(81):
            r.append(x)
            x = np.nextafter(x, np.float32(100))
    r = np.array(r, np.float32)
    P = np.tile([0.0, 0.0, 1.0], (len(r), 1))
    steps = dict(pose_cw=pose, t_wc=np.zeros(3), pos=P, normal=P, max_dist=r, min_dist=np.full(len(r), 0.01, np.float32), cand=np.ones(len(r), np.uint8))
    tot, out = check([kat, steps])
    assert out[4][0, :11].tolist() == [1, 0, 0, 0, 0, 1, 0, 1, 0, 1, 1] and out[1][0, :11].tolist() == [0, 0, 0, 0, 0, 0, 0, 7, 0, 4, 3]
    assert len(np.nonzero(np.diff(out[1][1]))[0]) == 7
    rng = np.random.default_rng(83)
    frames = [synth.synth_project_frame(int(rng.integers(1, 6000)), 12000 + i, cam) for i in range(150)]
    assert check(frames)[0] > 60000
    # the outputs are the matcher's inputs: B x NP x 3 uvr, int32 level, uint8 flag
    f = synth.synth_fuse_frame(500, 400, 77)
    pr = synth.synth_project_frame(400, 78, cam)
    uvr, level, viewcos, dist, inview = api.project_map_points(ctx, cam, *_pack_project(torch, [pr]))
    fe = [torch.from_numpy(np.ascontiguousarray(f[k][None])).cuda() for k in ("feat_uv", "feat_ur", "feat_oct", "feat_desc")]
    bi, bd = api.fuse_search(ctx, cam, *fe, uvr, level, inview, torch.from_numpy(f["mp_desc"][None]).cuda(), th=3.0)
    torch.cuda.synchronize()
    ref = oracle.project_map_points(cam, **pr)
    ri, rd, n = oracle.fuse_search(752, 480, f["feat_uv"], f["feat_ur"], f["feat_oct"], f["feat_desc"], ref[0], ref[1], ref[4], f["mp_desc"], th=3.0)
    assert np.array_equal(bi.cpu().numpy()[0], ri) and np.array_equal(bd.cpu().numpy()[0], rd)


def test_search_local_points_chain_matches_oracle(gpu, oracle):
    """gl_search_local_points = Tracking::searchLocalPoints on the device (tracking.cpp:213-270): the projection / visibility loop and
    ORBmatcher(0.8).searchByProjection on its outputs in one call, against the oracle's two functions one after the other: matches,
    counts and in-view flags bit for bit, frames of different sizes in a batch, th 3 and 5 (the first two frames)."""
    torch, ctx = gpu
    cam = api.Camera()
    PK = ("pose_cw", "t_wc", "pos", "normal", "max_dist", "min_dist", "cand")
    rng = np.random.default_rng(91)
    frames = [synth.synth_local_points_frame(NF, NP, 3000 + i, cam) for i, (NF, NP) in enumerate(((900, 2500), (1200, 4000), (300, 700), (50, 3000), (1000, 60)))]
    frames += [synth.synth_local_points_frame(int(rng.integers(20, 1500)), int(rng.integers(20, 4096)), 3100 + i, cam, float_uv=i % 3 != 0) for i in range(40)]
    B, NF, NP = len(frames), max(len(f["feat_oct"]) for f in frames), max(len(f["cand"]) for f in frames)
    t = dict(feat_uv=np.zeros((B, NF, 2)), feat_ur=np.full((B, NF), -1.0, np.float32), feat_oct=np.full((B, NF), -1, np.int32),
             feat_desc=np.zeros((B, NF, 32), np.uint8), feat_taken=np.zeros((B, NF), np.uint8), mp_desc=np.zeros((B, NP, 32), np.uint8))
    for b, f in enumerate(frames):
        for k in ("feat_uv", "feat_ur", "feat_oct", "feat_desc", "feat_taken"):
            t[k][b, :len(f["feat_oct"])] = f[k]
        t["mp_desc"][b, :len(f["cand"])] = f["mp_desc"]
    T = lambda a: torch.from_numpy(a).cuda()
    proj = _pack_project(torch, frames)
    tot = 0
    for th in (3.0, 5.0):
        match, nm, inview = api.search_local_points(ctx, cam, T(t["feat_uv"]), T(t["feat_ur"]), T(t["feat_oct"]), T(t["feat_desc"]), T(t["feat_taken"]),
                                                    *proj, T(t["mp_desc"]), th=th)
        torch.cuda.synchronize()
        match, nm, inview = match.cpu().numpy(), nm.cpu().numpy(), inview.cpu().numpy()
        for b, f in enumerate(frames):
            uvr, lvl, vc, dd, iv, n = oracle.project_map_points(cam, **{k: f[k] for k in PK})
            ref, nref = oracle.search_by_projection(cam.width, cam.height, f["feat_uv"], f["feat_ur"], f["feat_oct"], f["feat_desc"], f["feat_taken"],
                                                    uvr, lvl, vc, iv, f["mp_desc"], th=th)
            nf, npn = len(ref), len(iv)
            assert np.array_equal(inview[b, :npn], iv) and (inview[b, npn:] == 0).all(), b
            assert np.array_equal(match[b, :nf], ref) and nm[b] == nref, (b, th, int((match[b, :nf] != ref).sum()))
            assert (match[b, nf:] == -1).all()
            tot += nref
    assert tot > 5000, tot
