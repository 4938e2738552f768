"""Synthetic workloads for the tests and the benchmark.

No EuRoC imagery exists in this environment (the reference's .MISSING_LARGE_BLOBS),
so frame problems are synthesised from a GMM map + camera poses and labelled so.
Pure numpy, deterministic per seed; nothing here is measured or shipped.
"""
import numpy as np

V1_BBOX = np.array([[-5.0, 4.0], [-4.0, 5.0], [0.0, 3.5]])  # V1 map extent


def _haar(rng, n):
    q, r = np.linalg.qr(rng.standard_normal((n, 3, 3)))
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    det = np.linalg.det(q)
    q[:, :, 0] *= det[:, None]
    return q


def synth_gmm(K, seed=1, planar_frac=0.95):
    """Config 2 map: means ~U(bbox); Sigma = R diag(l) R^T, 95 % planar
    l=(1e-6, LogU[1e-3,0.1], LogU[1e-3,2]), 5 % volumetric l~LogU[1e-3,0.1]^3."""
    rng = np.random.default_rng(seed)
    mean = rng.uniform(V1_BBOX[:, 0], V1_BBOX[:, 1], size=(K, 3))
    R = _haar(rng, K)
    planar = rng.uniform(size=K) < planar_frac
    lam = np.empty((K, 3))
    lam[:, 0] = np.where(planar, 1e-6, np.exp(rng.uniform(np.log(1e-3), np.log(0.1), K)))
    lam[:, 1] = np.exp(rng.uniform(np.log(1e-3), np.log(0.1), K))
    lam[:, 2] = np.where(planar, np.exp(rng.uniform(np.log(1e-3), np.log(2.0), K)),
                         np.exp(rng.uniform(np.log(1e-3), np.log(0.1), K)))
    cov = np.einsum("kij,kj,klj->kil", R, lam, R)
    cov = 0.5 * (cov + cov.transpose(0, 2, 1))  # bit-symmetric
    return mean, cov.reshape(K, 9)


def synth_points(mean, cov, N, seed=1, frac_on=0.8):
    """80 % drawn from a random component, 20 % uniform in the bbox."""
    rng = np.random.default_rng(seed + 1000)
    K = mean.shape[0]
    comp = rng.integers(0, K, N)
    L = np.linalg.cholesky(cov.reshape(K, 3, 3)[comp] + 1e-15 * np.eye(3))
    pts = mean[comp] + np.einsum("nij,nj->ni", L, rng.standard_normal((N, 3)))
    uni = rng.uniform(size=N) >= frac_on
    lo, hi = mean.min(0) - 0.5, mean.max(0) + 0.5
    pts[uni] = rng.uniform(lo, hi, size=(int(uni.sum()), 3))
    return np.ascontiguousarray(pts)


# ---- SE3 helpers (numpy, independent of oracle/ and of the kernels) --------
def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W


def gt_row_to_Tcw(row):
    """gt_sync row (t x y z qx qy qz qw) = T_wc  ->  pose7 of T_cw (qx qy qz qw tx ty tz)."""
    Rwc = quat_to_R(row[4:8] / np.linalg.norm(row[4:8]))
    Rcw = Rwc.T
    tcw = -Rcw @ row[1:4]
    return np.concatenate([R_to_quat(Rcw), tcw])


def perturb_pose(pose7, rng, sig_rot=0.01, sig_t=0.02):
    """init = exp(xi) * T_cw, xi ~ N(0, diag(sig_rot, sig_t)^2)."""
    R = quat_to_R(pose7[:4])
    dR = so3_exp(rng.standard_normal(3) * sig_rot)
    return np.concatenate([R_to_quat(dR @ R), dR @ pose7[4:] + rng.standard_normal(3) * sig_t])


def synth_frame(mean, cov, pose7, cam, M, seed, outlier_frac=0.1, mono_frac=0.15, sigma2_inv=None,
                min_vis=20):
    """One frame problem from a map and a camera pose.
    Returns dict(pose_init, pose_gt, Xw (M,3), obs (M,3: u v u_right), octave (M,), comp (M,))."""
    rng = np.random.default_rng(seed)
    K = mean.shape[0]
    R, t = quat_to_R(pose7[:4]), pose7[4:]
    mc = mean @ R.T + t
    z = mc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = cam.fx * mc[:, 0] / z + cam.cx
        v = cam.fy * mc[:, 1] / z + cam.cy
    vis = np.nonzero((z > 0.3) & (z < 8.0) & (u >= 0) & (u < cam.width) & (v >= 0) & (v < cam.height))[0]
    if vis.size < min_vis:  # camera looks away from the map: fall back to the nearest components
        vis = np.argsort(np.abs(z - 2.0))[:max(min_vis, 64)]
    comp = vis[rng.integers(0, vis.size, M)]
    Lc = np.linalg.cholesky(cov.reshape(K, 3, 3)[comp] + 1e-15 * np.eye(3))
    Xw = mean[comp] + np.einsum("nij,nj->ni", Lc, rng.standard_normal((M, 3)))
    pc = Xw @ R.T + t
    pc[:, 2] = np.maximum(pc[:, 2], 0.2)
    octave = rng.integers(0, 8, M).astype(np.int32)
    sig = 1.2 ** octave
    uo = cam.fx * pc[:, 0] / pc[:, 2] + cam.cx + rng.standard_normal(M) * sig
    vo = cam.fy * pc[:, 1] / pc[:, 2] + cam.cy + rng.standard_normal(M) * sig
    ur = uo - cam.bf / pc[:, 2] + rng.standard_normal(M) * sig * 0.5
    out = rng.uniform(size=M) < outlier_frac
    uo[out] += rng.choice([-30.0, 30.0], int(out.sum()))
    vo[out] += rng.choice([-30.0, 30.0], int(out.sum()))
    mono = rng.uniform(size=M) < mono_frac
    ur[mono] = -1.0
    ur[~mono] = np.maximum(ur[~mono], 0.0)
    # the reference stores u_right as float (types/feature.h:37)
    ur = ur.astype(np.float32).astype(np.float64)
    obs = np.stack([uo, vo, ur], 1)
    return dict(pose_init=perturb_pose(pose7, rng), pose_gt=pose7.copy(), Xw=np.ascontiguousarray(Xw),
                obs=np.ascontiguousarray(obs), octave=octave, comp=comp.astype(np.int32))


def look_at_pose(eye, target, up=(0.0, 0.0, 1.0)):
    """T_cw of a camera at `eye` looking at `target` (z forward, x right, y down)."""
    zc = np.asarray(target, float) - np.asarray(eye, float)
    zc /= np.linalg.norm(zc)
    xc = np.cross(zc, np.asarray(up, float))
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    Rcw = np.stack([xc, yc, zc], 0)
    return np.concatenate([R_to_quat(Rcw), -Rcw @ np.asarray(eye, float)])


def _key_point_uv(uv, float_uv):
    """The reference's Feature::uv is a Vector2d built from a cv::KeyPoint's FLOAT coordinates (types/feature.h:11): realistic
    inputs are float values held in doubles (float_uv, the default; what the matchers' 32-bit window walk is exact for); arbitrary
    doubles (float_uv = False) take the matchers' general path."""
    return uv.astype(np.float32).astype(np.float64) if float_uv else uv


def synth_match_frame(NF, NP, seed, width=752, height=480, scale_factor=1.2, dup_frac=0.25, float_uv=True):
    """Inputs of ORBmatcher::searchByProjection for one frame: NF ORB-like features (uv, u_right, octave,
    256-bit descriptor, taken flag) and NP projected map points (uvr, predicted level, viewing cosine, valid,
    descriptor).  70 % of the map points are generated from a feature (projection within the search window,
    descriptor = the feature's with 0..70 flipped bits), `dup_frac` of those share their feature with another
    map point (conflicts: the earlier one must win), the rest are distractors."""
    rng = np.random.default_rng(seed)
    uv = np.stack([rng.uniform(-5, width + 5, NF), rng.uniform(-5, height + 5, NF)], 1)
    octv = rng.integers(0, 8, NF).astype(np.int32)
    octv[rng.uniform(size=NF) < 0.03] = -1  # padding slots
    ur = np.where(rng.uniform(size=NF) < 0.7, uv[:, 0] - rng.uniform(2, 60, NF), -1.0).astype(np.float32)
    desc = rng.integers(0, 256, (NF, 32), dtype=np.uint8)
    taken = (rng.uniform(size=NF) < 0.05).astype(np.uint8)
    sf = 1.2 ** np.arange(8)
    src = rng.integers(0, NF, NP)
    n_dup = int(dup_frac * NP)
    src[rng.integers(0, NP, n_dup)] = src[rng.integers(0, NP, n_dup)]
    from_feat = rng.uniform(size=NP) < 0.7
    level = np.clip(octv[src] + rng.integers(0, 2, NP), 0, 7).astype(np.float64)
    level = np.where(from_feat, level, rng.integers(0, 8, NP)).astype(np.float64)
    win = 4.0 * 3.0 * sf[level.astype(int)]
    mp_uv = np.where(from_feat[:, None], uv[src] + rng.uniform(-1.2, 1.2, (NP, 2)) * win[:, None],
                     np.stack([rng.uniform(-30, width + 30, NP), rng.uniform(-30, height + 30, NP)], 1))
    mp_ur = np.where(ur[src] > 0, ur[src] + rng.uniform(-1.2, 1.2, NP) * win, mp_uv[:, 0] - rng.uniform(2, 60, NP))
    mp_uvr = np.concatenate([mp_uv, mp_ur[:, None]], 1)
    viewcos = np.where(rng.uniform(size=NP) < 0.5, rng.uniform(0.9981, 1.0, NP), rng.uniform(0.5, 0.9979, NP))
    valid = (rng.uniform(size=NP) < 0.9).astype(np.uint8)
    mp_desc = desc[src].copy()
    nflip = rng.integers(0, 71, NP)
    for m in range(NP):
        if from_feat[m]:
            bits = rng.choice(256, nflip[m], replace=False)
            np.bitwise_xor.at(mp_desc[m], bits // 8, (1 << (bits % 8)).astype(np.uint8))
        else:
            mp_desc[m] = rng.integers(0, 256, 32, dtype=np.uint8)
    uv = _key_point_uv(uv, float_uv)
    return dict(width=width, height=height, feat_uv=uv, feat_ur=ur, feat_oct=octv, feat_desc=desc, feat_taken=taken,
                mp_uvr=mp_uvr, mp_level=level, mp_viewcos=viewcos, mp_valid=valid, mp_desc=mp_desc)


def synth_motion_frames(NF, NL, seed, cam, motion="none", rot_deg=8.0, float_uv=True):
    """Inputs of ORBmatcher::searchByProjection(CurrentFrame, LastFrame, th, bMono): a last frame with NL
    features carrying map points and a current frame with NF features that re-observes ~70 % of them
    (pixel noise, descriptor bit flips, in-plane rotation `rot_deg` so that the orientation histogram has a
    dominant bin plus outliers).  motion: "none" | "forward" | "backward" selects the level window."""
    rng = np.random.default_rng(seed)
    W, H = cam.width, cam.height
    pose_lw = np.array([0, 0, 0, 1.0, 0, 0, 0])
    dz = {"none": 0.0, "forward": -0.4, "backward": 0.4}[motion]  # t_cw.z of the current frame
    ang = np.deg2rad(1.5)
    pose_cw = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, dz])
    depth = rng.uniform(1.5, 8.0, NL)
    u_l, v_l = rng.uniform(20, W - 20, NL), rng.uniform(20, H - 20, NL)
    last_pt = np.stack([(u_l - cam.cx) / cam.fx * depth, (v_l - cam.cy) / cam.fy * depth, depth], 1)
    last_valid = (rng.uniform(size=NL) < 0.85).astype(np.uint8)
    last_oct = rng.integers(0, 8, NL).astype(np.int32)
    last_angle = rng.uniform(0, 360, NL).astype(np.float32)
    last_desc = rng.integers(0, 256, (NL, 32), dtype=np.uint8)
    R = quat_to_R(pose_cw[:4])
    pc = last_pt @ R.T + pose_cw[4:]
    u_c = cam.fx * pc[:, 0] / pc[:, 2] + cam.cx
    v_c = cam.fy * pc[:, 1] / pc[:, 2] + cam.cy
    src = rng.integers(0, NL, NF)
    reobs = rng.uniform(size=NF) < 0.7
    sf = 1.2 ** np.arange(8)
    noise = rng.uniform(-1.1, 1.1, (NF, 2)) * (7.0 * sf[last_oct[src]])[:, None]
    uv = np.where(reobs[:, None], np.stack([u_c[src], v_c[src]], 1) + noise,
                  np.stack([rng.uniform(-5, W + 5, NF), rng.uniform(-5, H + 5, NF)], 1))
    octv = np.clip(last_oct[src] + rng.integers(-1, 2, NF), 0, 7).astype(np.int32)
    octv[rng.uniform(size=NF) < 0.03] = -1
    ur_true = uv[:, 0] - cam.bf / np.maximum(pc[src, 2], 0.1)
    ur = np.where(rng.uniform(size=NF) < 0.7, ur_true + rng.uniform(-1.1, 1.1, NF) * 7.0 * sf[last_oct[src]], -1.0)
    ur = ur.astype(np.float32)
    out_rot = rng.uniform(size=NF) < 0.15
    angle = np.where(out_rot, rng.uniform(0, 360, NF), (last_angle[src] - rot_deg + rng.normal(0, 3, NF)) % 360.0)
    desc = last_desc[src].copy()
    nflip = rng.integers(0, 71, NF)
    for i in range(NF):
        if reobs[i]:
            bits = rng.choice(256, nflip[i], replace=False)
            np.bitwise_xor.at(desc[i], bits // 8, (1 << (bits % 8)).astype(np.uint8))
        else:
            desc[i] = rng.integers(0, 256, 32, dtype=np.uint8)
    taken = (rng.uniform(size=NF) < 0.05).astype(np.uint8)
    uv = _key_point_uv(uv, float_uv)
    return dict(pose_cw=pose_cw, pose_lw=pose_lw, feat_uv=uv, feat_ur=ur, feat_oct=octv,
                feat_angle=angle.astype(np.float32), feat_desc=desc, feat_taken=taken, last_pt=last_pt,
                last_valid=last_valid, last_oct=last_oct, last_angle=last_angle, last_desc=last_desc)


def synth_tri_matches(mean, cov, pose1, pose2, cam, N, seed, K_cand=5, allowed=None):
    """Epipolar matches between two key-frames for Localization::createMapPoints: points drawn from map
    components seen by both, observed with pixel noise; a random mix of stereo (u_right, depth) and mono
    (u_right = depth = -1) key-points, 10 % gross mismatches, candidate component lists (restricted to
    `allowed` components when given: needle-like components have no unique plane normal, so two eigen-solvers
    legitimately disagree on them)."""
    rng = np.random.default_rng(seed)
    f1 = synth_frame(mean, cov, pose1, cam, N, seed, outlier_frac=0.0, mono_frac=0.0)
    X = f1["Xw"]
    R2, t2 = quat_to_R(pose2[:4]), pose2[4:]
    pc2 = X @ R2.T + t2
    R1, t1 = quat_to_R(pose1[:4]), pose1[4:]
    pc1 = X @ R1.T + t1

    def kps(pc):
        u = cam.fx * pc[:, 0] / pc[:, 2] + cam.cx + rng.standard_normal(N) * 0.6
        v = cam.fy * pc[:, 1] / pc[:, 2] + cam.cy + rng.standard_normal(N) * 0.6
        st = rng.uniform(size=N) < 0.5
        depth = np.where(st, pc[:, 2] * (1 + rng.standard_normal(N) * 0.01), -1.0).astype(np.float32)
        ur = np.where(st, u - cam.bf / np.maximum(pc[:, 2], 0.05) + rng.standard_normal(N) * 0.6, -1.0)
        lost = st & (ur < 0)  # a real stereo match has u_right >= 0; otherwise the key-point is monocular
        ur[lost] = -1.0
        depth[lost] = -1.0
        return np.stack([u, v, ur], 1), depth
    k1, d1 = kps(pc1)
    k2, d2 = kps(pc2)
    bad = rng.uniform(size=N) < 0.1
    k2[bad, :2] += rng.uniform(-12, 12, (int(bad.sum()), 2))
    o1 = rng.integers(0, 8, N).astype(np.int32)
    o2 = np.clip(o1 + rng.integers(-1, 2, N), 0, 7).astype(np.int32)
    pool = np.arange(mean.shape[0]) if allowed is None else np.nonzero(allowed)[0]
    c1 = -np.ones((N, K_cand), np.int32)
    c2 = -np.ones((N, K_cand), np.int32)
    for i in range(N):
        n = int(rng.integers(0, K_cand))
        own = int(f1["comp"][i])
        c = ([own] if allowed is None or allowed[own] else []) + [int(x) for x in rng.choice(pool, K_cand - 1)]
        rng.shuffle(c)
        n = min(n, len(c))
        c1[i, :n] = c[:n]
        m = int(rng.integers(0, K_cand - 1))
        c2[i, :m] = rng.choice(pool, m)
    return dict(pose1=np.tile(pose1, (N, 1)), uvr1=k1, depth1=d1, oct1=o1, pose2=np.tile(pose2, (N, 1)), uvr2=k2,
                depth2=d2, oct2=o2, cand1=c1, n1=(c1 >= 0).sum(1).astype(np.int32), cand2=c2,
                n2=(c2 >= 0).sum(1).astype(np.int32))


def fundamental_and_epipole(pose1, pose2, cam):
    """MathUtils::computeFundamentalMatrix(Tcw1, K1, Tcw2, K2) (math_utils.cpp:16-43) and the epipole of
    searchForTriangulation (orb_matcher.cpp:155-160; the reference maps Tcw1.translation(), not kf1's camera centre, into
    kf2 - kept) - the host's part of gl_search_for_triangulation's inputs (plain fp64 here; Eigen in the reference host)."""
    R1, t1 = quat_to_R(pose1[:4]), pose1[4:]
    R2, t2 = quat_to_R(pose2[:4]), pose2[4:]
    R12 = R1 @ R2.T
    t12 = -(R12 @ t2) + t1
    sk = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    E = sk @ R12
    fx, fy, cx, cy = (float(np.float32(v)) for v in (cam.fx, cam.fy, cam.cx, cam.cy))  # (config.h: float scalars)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    F = np.linalg.inv(K.T) @ E @ np.linalg.inv(K)
    C2 = R2 @ t1 + t2
    invz = np.float32(1.0) / np.float32(C2[2])
    ex = np.float32(fx * C2[0] * invz + cx)
    ey = np.float32(fy * C2[1] * invz + cy)
    return F, np.array([ex, ey], np.float32)


def synth_tri_search_pair(N1, N2, seed, cam, n_nodes=160, only_stereo_frac=0.6, pad=0):
    """Inputs of ORBmatcher::searchForTriangulation for one key-frame pair: two key-frames a short baseline apart that see the
    same 3-D points (pixel noise, descriptor bit flips, mostly the same vocabulary node: DBoW2 words of the same patch agree
    at the feature-vector level most of the time), distractor features, features that already have map points, a dominant
    rotation bin with outliers; the two feature vectors as CSR (node ids ascending, feature indices ascending inside a node:
    DBoW2 appends them in index order).  Nodes are few and crowded, so that features compete for the same partner (the
    order-dependent part of the reference loop) and equal descriptor distances occur."""
    rng = np.random.default_rng(seed)
    W, H = cam.width, cam.height
    pose1 = np.array([0, 0, 0, 1.0, 0, 0, 0])
    ang = np.deg2rad(rng.uniform(-3, 3))
    pose2 = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), rng.uniform(0.1, 0.3), rng.uniform(-0.05, 0.05), rng.uniform(-0.1, 0.1)])
    NP = max(N1, N2)
    depth = rng.uniform(1.5, 9.0, NP)
    u1, v1 = rng.uniform(10, W - 10, NP), rng.uniform(10, H - 10, NP)
    X = np.stack([(u1 - cam.cx) / cam.fx * depth, (v1 - cam.cy) / cam.fy * depth, depth], 1)
    R2, t2 = quat_to_R(pose2[:4]), pose2[4:]
    pc2 = X @ R2.T + t2
    u2 = cam.fx * pc2[:, 0] / pc2[:, 2] + cam.cx
    v2 = cam.fy * pc2[:, 1] / pc2[:, 2] + cam.cy
    base_desc = rng.integers(0, 256, (NP, 32), dtype=np.uint8)
    base_node = rng.integers(0, n_nodes, NP) * 7 + 3  # sparse node ids
    base_oct = rng.integers(0, 8, NP)
    base_angle = rng.uniform(0, 360, NP)
    sf = 1.2 ** np.arange(8)

    def view(N, u, v, z, rot, sd):
        r = np.random.default_rng(sd)
        src = r.permutation(NP)[:N] if N <= NP else r.integers(0, NP, N)
        ndup = int(0.15 * N)  # several features of the SAME point (neighbouring pyramid levels do that): rivals for one partner
        src[r.integers(0, N, ndup)] = src[r.integers(0, N, ndup)]
        real = r.uniform(size=N) < 0.75
        octv = np.clip(base_oct[src] + r.integers(-1, 2, N), 0, 7).astype(np.int32)
        noise = r.normal(0, 0.6, (N, 2)) * sf[octv][:, None]
        gross = r.uniform(size=N) < 0.08  # off the epipolar line
        noise[gross] += r.uniform(-25, 25, (int(gross.sum()), 2))
        uv = np.where(real[:, None], np.stack([u[src], v[src]], 1) + noise, np.stack([r.uniform(0, W, N), r.uniform(0, H, N)], 1))
        stereo = r.uniform(size=N) < only_stereo_frac
        ur = np.where(stereo, uv[:, 0] - cam.bf / np.maximum(z[src], 0.1), -1.0).astype(np.float32)
        desc = base_desc[src].copy()
        nflip = r.choice([0, 4, 8, 8, 12, 16, 16, 24, 40, 60], N)  # few distinct values: equal distances happen
        for i in range(N):
            if real[i]:
                bits = r.choice(256, nflip[i], replace=False)
                np.bitwise_xor.at(desc[i], bits // 8, (1 << (bits % 8)).astype(np.uint8))
            else:
                desc[i] = r.integers(0, 256, 32, dtype=np.uint8)
        node = np.where(real & (r.uniform(size=N) < 0.85), base_node[src], r.integers(0, n_nodes, N) * 7 + 3)
        out_rot = r.uniform(size=N) < 0.12
        angle = np.where(out_rot, r.uniform(0, 360, N), (base_angle[src] + rot + r.normal(0, 3, N)) % 360.0).astype(np.float32)
        has_mp = (r.uniform(size=N) < 0.3).astype(np.uint8)
        if pad:
            octv[r.uniform(size=N) < 0.02] = -1
        ids = np.unique(node[octv >= 0])
        ptr, idx = [0], []
        for n in ids:
            members = np.nonzero((node == n) & (octv >= 0))[0]
            idx.extend(members.tolist())
            ptr.append(len(idx))
        return dict(uv=uv, ur=ur, oct=octv, angle=angle, desc=desc, has_mp=has_mp, node_id=ids.astype(np.int32),
                    node_ptr=np.array(ptr, np.int32), node_idx=np.array(idx, np.int32))

    kf1 = view(N1, u1, v1, depth, 0.0, seed * 2 + 1)
    kf2 = view(N2, u2, v2, pc2[:, 2], -14.0, seed * 2 + 2)
    fmat, epi = fundamental_and_epipole(pose1, pose2, cam)
    return dict(kf1=kf1, kf2=kf2, fmat=fmat, epipole=epi, pose1=pose1, pose2=pose2)


def synth_bow_pair(N1, N2, seed, cam, n_nodes=160, mp_frac=0.7):
    """Inputs of ORBmatcher::searchByBoW for one reference key-frame / current frame pair: the two views of synth_tri_search_pair
    (same points seen twice, descriptor bit flips, crowded vocabulary nodes, rivals for one partner, equal distances), the
    key-frame's features holding a valid map point with probability mp_frac.  -> (kf, fr) dicts."""
    pr = synth_tri_search_pair(N1, N2, seed, cam, n_nodes=n_nodes)
    kf, fr = dict(pr["kf1"]), dict(pr["kf2"])
    rng = np.random.default_rng(seed + 77777)
    kf["has_mp"] = (rng.uniform(size=len(kf["oct"])) < mp_frac).astype(np.uint8)
    return kf, fr


def synth_fuse_frame(NF, NP, seed, width=752, height=480, scale_factor=1.2, float_coords=False):
    """Inputs of the matching half of Localization::fuseObservations for one key-frame: NF features (clustered, so that windows
    hold several candidates; some with the SAME descriptor: ties) and NP projected map points - 75 % made from a feature with a
    pixel error around the chi2 gates (5.99 / 7.8 at the feature's level, so that some pass and some do not), level = the feature's
    octave or one above, descriptor with 0 .. 70 flipped bits (TH_LOW = 50 in between), the rest distractors; padding slots, points
    outside the image, invalid points.  float_coords: the feature coordinates are float values, as cv::KeyPoint's are (the record walk of
    gl_fuse_search, round 6); the default keeps the frames of the earlier rounds (arbitrary doubles: the walk from global memory)."""
    rng = np.random.default_rng(seed)
    sf = scale_factor ** np.arange(8)
    ncl = max(1, NF // 6)
    centres = np.stack([rng.uniform(0, width, ncl), rng.uniform(0, height, ncl)], 1)
    uv = centres[rng.integers(0, ncl, NF)] + rng.normal(0, 4.0, (NF, 2))
    if float_coords:
        uv = uv.astype(np.float32).astype(np.float64)
    octv = rng.integers(0, 8, NF).astype(np.int32)
    octv[rng.uniform(size=NF) < 0.03] = -1
    ur = np.where(rng.uniform(size=NF) < 0.7, uv[:, 0] - rng.uniform(2, 60, NF), -1.0).astype(np.float32)
    desc = rng.integers(0, 256, (NF, 32), dtype=np.uint8)
    twin = rng.integers(0, NF, NF // 10)  # equal descriptors on neighbouring features: equal distances
    desc[twin] = desc[(twin + 1) % NF]
    src = rng.integers(0, NF, NP)
    from_feat = rng.uniform(size=NP) < 0.75
    level = np.clip(np.maximum(octv[src], 0) + rng.integers(0, 2, NP), 0, 7).astype(np.int32)
    level = np.where(from_feat, level, rng.integers(0, 8, NP)).astype(np.int32)
    sig = sf[np.maximum(octv[src], 0)]
    err = rng.normal(0, 1.2, (NP, 3)) * sig[:, None]
    mp_uv = np.where(from_feat[:, None], uv[src] + err[:, :2],
                     np.stack([rng.uniform(-30, width + 30, NP), rng.uniform(-30, height + 30, NP)], 1))
    mp_ur = np.where(ur[src] >= 0, ur[src] + err[:, 2], mp_uv[:, 0] - rng.uniform(2, 60, NP))
    mp_uvr = np.concatenate([mp_uv, mp_ur[:, None]], 1)
    valid = (rng.uniform(size=NP) < 0.9).astype(np.uint8)
    mp_desc = desc[src].copy()
    nflip = rng.integers(0, 71, NP)
    for m in range(NP):
        if from_feat[m]:
            bits = rng.choice(256, nflip[m], replace=False)
            np.bitwise_xor.at(mp_desc[m], bits // 8, (1 << (bits % 8)).astype(np.uint8))
        else:
            mp_desc[m] = rng.integers(0, 256, 32, dtype=np.uint8)
    return dict(width=width, height=height, feat_uv=uv, feat_ur=ur, feat_oct=octv, feat_desc=desc, mp_uvr=mp_uvr, mp_level=level,
                mp_valid=valid, mp_desc=mp_desc)


def synth_project_frame(NP, seed, cam, scale_factor=1.2):
    """Inputs of the projection / visibility loop of Tracking::searchLocalPoints for one frame: a camera somewhere in a room, NP map
    points around it - most in front (a share behind it, outside the image, exactly on the image border), each with the normal and the
    distance band of a reference key-frame that saw it: normal = direction from that key-frame (so that view_cos lands on both sides
    of 0.5), max_dist_ = its distance x 1.2 ^ level, min_dist_ = max_dist_ / 1.2 ^ 7 (ORB-SLAM's update of normal and depth) - the
    reference key-frame at 0.3 .. 4 times the camera's distance, so that the band test rejects some and the predicted level covers
    0 .. 7 with clamping at both ends; a share of non-candidates."""
    rng = np.random.default_rng(seed)
    eye = rng.uniform(-3, 3, 3)
    unit = lambda n: (lambda g: g / np.linalg.norm(g, axis=1)[:, None])(rng.standard_normal((n, 3)))
    target = eye + unit(1)[0] * 3.0
    pose = look_at_pose(eye, target, up=unit(1)[0])
    R = quat_to_R(pose[:4])
    t_wc = -R.T @ pose[4:]
    depth = rng.uniform(0.3, 12.0, NP)
    depth[rng.uniform(size=NP) < 0.08] *= -1.0
    u = rng.uniform(-0.15 * cam.width, 1.15 * cam.width, NP)
    v = rng.uniform(-0.15 * cam.height, 1.15 * cam.height, NP)
    pc = np.stack([(u - cam.cx) / cam.fx * depth, (v - cam.cy) / cam.fy * depth, depth], 1)
    pos = (pc - pose[4:]) @ R  # R^T (pc - t)
    ref_dir = -(pos - t_wc) / np.linalg.norm(pos - t_wc, axis=1)[:, None]
    ang = rng.uniform(0, np.deg2rad(80), NP)  # the key-frame that saw the point: 0 .. 80 degrees away from the camera's ray
    axis = np.cross(ref_dir, unit(NP))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    ref_ray = ref_dir * np.cos(ang)[:, None] + np.cross(axis, ref_dir) * np.sin(ang)[:, None]
    normal = -ref_ray  # from the key-frame towards the point
    ref_dist = np.abs(depth) * np.exp(rng.uniform(np.log(0.3), np.log(4.0), NP))
    level = rng.integers(0, 8, NP)
    max_dist = (ref_dist * scale_factor ** level).astype(np.float32)
    min_dist = (max_dist / np.float32(scale_factor ** 7)).astype(np.float32)
    cand = (rng.uniform(size=NP) < 0.9).astype(np.uint8)
    return dict(pose_cw=pose, t_wc=t_wc, pos=pos, normal=normal, max_dist=max_dist, min_dist=min_dist, cand=cand)


def synth_local_points_frame(NF, NP, seed, cam, scale_factor=1.2, float_uv=True):
    """One frame of Tracking::searchLocalPoints: the map points of synth_project_frame and NF ORB-like features, 70 % of them at the
    (approximately) projected position of a map point in front of the camera - a few pixels off, at the octave of the point's
    predicted level or one below, descriptor = the point's with 0 .. 70 flipped bits - so that the chain project -> searchByProjection
    has matches, contested features, ratio-test failures and points whose window holds nothing.  The rest of the features are clutter."""
    rng = np.random.default_rng(seed)
    f = synth_project_frame(NP, seed, cam, scale_factor)
    R, t = quat_to_R(f["pose_cw"][:4]), f["pose_cw"][4:]
    pc = f["pos"] @ R.T + t
    z = np.where(np.abs(pc[:, 2]) < 1e-6, 1e-6, pc[:, 2])
    uv = np.stack([cam.fx * pc[:, 0] / z + cam.cx, cam.fy * pc[:, 1] / z + cam.cy], 1)
    dist = np.linalg.norm(f["pos"] - f["t_wc"], axis=1)
    lvl = np.clip(np.ceil(np.log(np.maximum(f["max_dist"] / np.maximum(dist, 1e-9), 1e-9)) / np.log(scale_factor)), 0, 7).astype(np.int32)
    front = np.nonzero((z > 0) & (uv[:, 0] > 0) & (uv[:, 0] < cam.width) & (uv[:, 1] > 0) & (uv[:, 1] < cam.height))[0]
    mp_desc = rng.integers(0, 256, (NP, 32), dtype=np.uint8)
    feat_uv = np.stack([rng.uniform(0, cam.width, NF), rng.uniform(0, cam.height, NF)], 1)
    feat_oct = rng.integers(0, 8, NF).astype(np.int32)
    feat_desc = rng.integers(0, 256, (NF, 32), dtype=np.uint8)
    if len(front):
        src = front[rng.integers(0, len(front), NF)]
        near = rng.uniform(size=NF) < 0.7
        sig = scale_factor ** lvl[src]
        feat_uv = np.where(near[:, None], uv[src] + rng.normal(0, 1.5, (NF, 2)) * sig[:, None], feat_uv)
        feat_oct = np.where(near, np.clip(lvl[src] - rng.integers(0, 2, NF), 0, 7), feat_oct).astype(np.int32)
        nflip = rng.integers(0, 71, NF)
        for i in np.nonzero(near)[0]:
            d = mp_desc[src[i]].copy()
            bits = rng.choice(256, nflip[i], replace=False)
            np.bitwise_xor.at(d, bits // 8, (1 << (bits % 8)).astype(np.uint8))
            feat_desc[i] = d
    feat_ur = np.where(rng.uniform(size=NF) < 0.7, feat_uv[:, 0] - rng.uniform(2, 60, NF), -1.0).astype(np.float32)
    feat_taken = (rng.uniform(size=NF) < 0.05).astype(np.uint8)
    f.update(feat_uv=_key_point_uv(feat_uv, float_uv), feat_ur=feat_ur, feat_oct=feat_oct, feat_desc=feat_desc, feat_taken=feat_taken, mp_desc=mp_desc)
    return f


def synth_chain_frame(NF, NL, NP, seed, cam, scale_factor=1.2, temporal_frac=0.0, NK=0, n_nodes=120, pred_rot_deg=None):
    """One tracked frame for gl_track_frame_chain (trackWithMotionModel -> searchLocalPoints -> trackLocalMap), geometrically
    CONSISTENT so that both pose optimisations have inliers: a last frame at the identity with NL map points, a current frame at a
    small true motion whose features re-observe 60 % of them (sub-pixel noise scaled by the octave, 8 % gross outliers), a
    motion-model prediction a fraction of a degree / a centimetre off, and a local map of NP points = the last frame's valid map
    points (last_to_local) + points seen by half of the remaining features (position from the TRUE pose, descriptor = the feature's
    with flipped bits, distance bounds that predict the feature's octave) + distractors.
    Round 6 (drawn from a generator of their own: the frames of the earlier seeds are what they were):
    temporal_frac > 0: that share of the last frame's map points are TEMPORAL points (createTemporalPoints, tracking.cpp:44-46:
    last_observed = 0, not in the local map through last_to_local - but the local map still holds a real point at the same place, so
    that searchLocalPoints can replace them); NK > 0: a reference key-frame that sees the last frame's map points (features, DBoW2
    feature vector as CSR, kf_pt / kf_to_local) and the frame's own feature vector - the inputs of the trackKeyFrame fallback;
    pred_rot_deg: the motion-model prediction is off by that many degrees (10: no th = 14 window holds a match any more)."""
    rng = np.random.default_rng(seed)
    W, H = cam.width, cam.height
    sf = scale_factor ** np.arange(8)
    rotv = lambda ax, a: np.concatenate([np.asarray(ax, float) / np.linalg.norm(ax) * np.sin(a / 2), [np.cos(a / 2)]])
    pose_lw = np.array([0, 0, 0, 1.0, 0, 0, 0])
    pose_true = np.concatenate([rotv(rng.standard_normal(3), np.deg2rad(rng.uniform(0.5, 2.0))), rng.uniform(-0.04, 0.04, 3)])
    dq = rotv(rng.standard_normal(3), np.deg2rad(rng.uniform(0.05, 0.3)))
    q0, q1 = pose_true[:4], dq  # prediction = dq * true (Hamilton product on (x, y, z, w))
    qp = np.concatenate([q1[3] * q0[:3] + q0[3] * q1[:3] + np.cross(q1[:3], q0[:3]), [q1[3] * q0[3] - q1[:3] @ q0[:3]]])
    pose_pred = np.concatenate([qp / np.linalg.norm(qp), quat_to_R(dq) @ pose_true[4:] + rng.uniform(-0.01, 0.01, 3)])
    if pred_rot_deg is not None:
        dq2 = rotv([0.0, 1.0, 0.0], np.deg2rad(pred_rot_deg))
        qp = np.concatenate([dq2[3] * q0[:3] + q0[3] * dq2[:3] + np.cross(dq2[:3], q0[:3]), [dq2[3] * q0[3] - dq2[:3] @ q0[:3]]])
        pose_pred = np.concatenate([qp / np.linalg.norm(qp), quat_to_R(dq2) @ pose_true[4:]])
    R, t = quat_to_R(pose_true[:4]), pose_true[4:]
    proj = lambda X: (lambda pc: (cam.fx * pc[:, 0] / pc[:, 2] + cam.cx, cam.fy * pc[:, 1] / pc[:, 2] + cam.cy, pc[:, 2]))(X @ R.T + t)
    # last frame
    depth = rng.uniform(1.5, 8.0, NL)
    u_l, v_l = rng.uniform(20, W - 20, NL), rng.uniform(20, H - 20, NL)
    last_pt = np.stack([(u_l - cam.cx) / cam.fx * depth, (v_l - cam.cy) / cam.fy * depth, depth], 1)
    last_valid = (rng.uniform(size=NL) < 0.85).astype(np.uint8)
    last_oct = rng.integers(0, 8, NL).astype(np.int32)
    last_angle = rng.uniform(0, 360, NL).astype(np.float32)
    last_desc = rng.integers(0, 256, (NL, 32), dtype=np.uint8)
    u_c, v_c, z_c = proj(last_pt)
    # current frame
    src = rng.permutation(NL)[:min(NF, NL)] if NL >= NF else rng.integers(0, NL, NF)
    src = np.resize(src, NF)
    reobs = rng.uniform(size=NF) < 0.6
    octv = np.clip(last_oct[src] + rng.integers(-1, 2, NF), 0, 7).astype(np.int32)
    gross = rng.uniform(size=NF) < 0.08
    noise = rng.normal(0, 0.6, (NF, 2)) * sf[octv][:, None] + np.where(gross[:, None], rng.uniform(-15, 15, (NF, 2)), 0.0)
    uv = np.where(reobs[:, None], np.stack([u_c[src], v_c[src]], 1) + noise, np.stack([rng.uniform(5, W - 5, NF), rng.uniform(5, H - 5, NF)], 1))
    zf = np.where(reobs, z_c[src], rng.uniform(1.5, 8.0, NF))
    ur = np.where(rng.uniform(size=NF) < 0.7, uv[:, 0] - cam.bf / zf + rng.normal(0, 0.6, NF) * sf[octv], -1.0).astype(np.float32)
    angle = np.where(rng.uniform(size=NF) < 0.1, rng.uniform(0, 360, NF), (last_angle[src] - 6.0 + rng.normal(0, 3, NF)) % 360.0).astype(np.float32)
    desc = np.where(reobs[:, None], last_desc[src], rng.integers(0, 256, (NF, 32), dtype=np.uint8)).astype(np.uint8)
    flips = rng.integers(0, 51, NF)
    for i in range(NF):
        bits = rng.choice(256, flips[i], replace=False)
        np.bitwise_xor.at(desc[i], bits // 8, (1 << (bits % 8)).astype(np.uint8))
    octv[rng.uniform(size=NF) < 0.02] = -1
    uv = _key_point_uv(uv, True)
    taken = (rng.uniform(size=NF) < 0.03).astype(np.uint8)
    # local map: the last frame's valid map points first
    t_wc = -R.T @ t
    shared = np.nonzero(last_valid & (rng.uniform(size=NL) < 0.9))[0][:max(NP // 2, 1)]
    last_to_local = -np.ones(NL, np.int32)
    last_to_local[shared] = np.arange(len(shared))
    mp_pos, mp_desc, mp_lvl = [last_pt[shared]], [last_desc[shared]], [last_oct[shared]]
    free = np.nonzero(~reobs & (octv >= 0))[0]
    free = free[rng.uniform(size=len(free)) < 0.5][:max(NP - len(shared) - 1, 0)]
    zf2 = rng.uniform(1.5, 8.0, len(free))
    uvn = uv[free] + rng.normal(0, 0.5, (len(free), 2)) * sf[octv[free]][:, None]
    pc = np.stack([(uvn[:, 0] - cam.cx) / cam.fx * zf2, (uvn[:, 1] - cam.cy) / cam.fy * zf2, zf2], 1)
    mp_pos.append((pc - t) @ R)
    d2 = desc[free].copy()
    for i in range(len(free)):
        bits = rng.choice(256, int(rng.integers(0, 41)), replace=False)
        np.bitwise_xor.at(d2[i], bits // 8, (1 << (bits % 8)).astype(np.uint8))
    mp_desc.append(d2)
    mp_lvl.append(np.clip(octv[free] + rng.integers(0, 2, len(free)), 0, 7))
    nrest = NP - len(shared) - len(free)
    zr = rng.uniform(0.5, 10.0, nrest)
    pr = np.stack([(rng.uniform(0, W, nrest) - cam.cx) / cam.fx * zr, (rng.uniform(0, H, nrest) - cam.cy) / cam.fy * zr, zr], 1)
    mp_pos.append((pr - t) @ R)
    mp_desc.append(rng.integers(0, 256, (nrest, 32), dtype=np.uint8))
    mp_lvl.append(rng.integers(0, 8, nrest))
    mp_pos, mp_desc, mp_lvl = np.concatenate(mp_pos), np.concatenate(mp_desc).astype(np.uint8), np.concatenate(mp_lvl).astype(int)
    dist = np.linalg.norm(mp_pos - t_wc, axis=1)
    max_dist = (dist * scale_factor ** (mp_lvl - 0.5)).astype(np.float32)  # predicted level = ceil(log(max / dist) / log(1.2)) = the level
    min_dist = (max_dist / np.float32(scale_factor ** 7) * 0.5).astype(np.float32)
    ray = (mp_pos - t_wc) / dist[:, None]
    tilt = rng.standard_normal((NP, 3)) * 0.25
    normal = ray + tilt
    normal /= np.linalg.norm(normal, axis=1)[:, None]
    cand = (rng.uniform(size=NP) < 0.95).astype(np.uint8)
    f = dict(feat_uv=uv, feat_ur=ur, feat_oct=octv, feat_angle=angle, feat_desc=desc, feat_taken=taken, pose_lw=pose_lw, last_pt=last_pt,
             last_valid=last_valid, last_oct=last_oct, last_angle=last_angle, last_desc=last_desc, last_to_local=last_to_local, mp_pos=mp_pos,
             mp_normal=normal, mp_max_dist=max_dist, mp_min_dist=min_dist, mp_cand=cand, mp_desc=mp_desc, pose_cw=pose_pred, pose_true=pose_true)
    r2 = np.random.default_rng(seed + 424242)
    f["last_observed"] = (r2.uniform(size=NL) >= temporal_frac).astype(np.uint8)
    if temporal_frac > 0:
        f["last_to_local"] = np.where(f["last_observed"] != 0, last_to_local, -1).astype(np.int32)
    if NK > 0:
        flip = lambda d, nmax: [np.bitwise_xor.at(d[i], b // 8, (1 << (b % 8)).astype(np.uint8)) for i in range(len(d))
                                for b in [r2.choice(256, int(r2.integers(0, nmax + 1)), replace=False)]]
        node_pt = r2.integers(0, n_nodes, NL) * 5 + 2  # the vocabulary node of a map point's patch (sparse ids)
        lk = r2.choice(np.nonzero(last_valid)[0], NK, replace=True)
        kf_desc = last_desc[lk].copy()
        flip(kf_desc, 30)
        kf_angle = np.where(r2.uniform(size=NK) < 0.1, r2.uniform(0, 360, NK), (last_angle[lk] + 4.0 + r2.normal(0, 3, NK)) % 360.0).astype(np.float32)
        kf_node = np.where(r2.uniform(size=NK) < 0.9, node_pt[lk], r2.integers(0, n_nodes, NK) * 5 + 2)
        fr_node = np.where(reobs & (r2.uniform(size=NF) < 0.85), node_pt[src], r2.integers(0, n_nodes, NF) * 5 + 2)

        def csr(node, ok):
            ids = np.unique(node[ok])
            ptr, idx = [0], []
            for n in ids:
                idx.extend(np.nonzero((node == n) & ok)[0].tolist())
                ptr.append(len(idx))
            return ids.astype(np.int32), np.array(ptr, np.int32), np.array(idx, np.int32)
        kid, kptr, kidx = csr(kf_node, np.ones(NK, bool))
        fid, fptr, fidx = csr(fr_node, octv >= 0)
        f.update(kf_angle=kf_angle, kf_desc=kf_desc, kf_has_mp=(r2.uniform(size=NK) < 0.9).astype(np.uint8), kf_node_id=kid, kf_node_ptr=kptr,
                 kf_node_idx=kidx, kf_pt=last_pt[lk].copy(), kf_to_local=f["last_to_local"][lk].astype(np.int32), feat_node_id=fid, feat_node_ptr=fptr,
                 feat_node_idx=fidx)
    return f
