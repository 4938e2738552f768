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
def test_search_by_projection_matches_oracle(gpu, oracle, opt, NF, NP, th, dup, shape):
    torch, ctx = gpu
    opt("match_desc_lds", int(shape))
    frames = [synth.synth_match_frame(NF, NP, 1000 * NF + 7 * b, dup_frac=dup) for b in range(5)]
    m, n = run_gpu(torch, ctx, frames, th)
    tot = 0
    for b, f in enumerate(frames):
        m_ref, n_ref = oracle.search_by_projection(th=th, **f)
        assert n[b] == n_ref, (b, int(n[b]), n_ref)
        assert np.array_equal(m[b], m_ref), (b, int((m[b] != m_ref).sum()))
        tot += n_ref
    assert tot > 0 or NF == 1


def test_search_by_projection_conflict_chain(gpu, oracle):
    """Worst case for the fixed-point iteration: every map point wants the same features, so each one
    only settles after all earlier ones have (as many rounds as map points)."""
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
    frames = [synth.synth_motion_frames(NF, NL, 77 * NF + b, CamF, motion) for b in range(4)]
    m, n = run_gpu_frame(torch, ctx, frames, th, mono, chk)
    tot = 0
    for b, f in enumerate(frames):
        m_ref, n_ref = oracle.search_by_projection_frame(CamF, th=th, mono=mono, check_orientation=chk, **f)
        assert n[b] == n_ref, (b, int(n[b]), n_ref)
        assert np.array_equal(m[b], m_ref), (b, int((m[b] != m_ref).sum()))
        tot += n_ref
    assert tot > 50
