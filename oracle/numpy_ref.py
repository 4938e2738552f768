"""TEST INFRASTRUCTURE ONLY -- an independent numpy restatement of the gmmloc hot path.

Purpose: pin the C++ oracle (oracle/gmmloc_oracle.cpp).  The reference ships no tests
and cannot be compiled here (Eigen / g2o / OpenCV / ROS absent), so this second,
differently-structured implementation (LAPACK inverses / eigh / cholesky, vectorised
pair sweeps, a dense *un-reduced* Levenberg system instead of the Schur complement,
geometric Jacobians J = -Jpi [ -[p]x | I ]) is what the golden vectors under
tests/golden/ are generated from (tools/make_golden.py).  Reference lines are cited per
function; g2o semantics per SURVEY.md Appendix A.
"""
import numpy as np

F32 = np.float32


def default_sigma2_inv():
    # init_config.hpp:60-79 -- float arithmetic
    sf = F32(1.0)
    out = [F32(1.0)]
    for _ in range(1, 8):
        sf = F32(sf * F32(1.2))
        out.append(F32(F32(1.0) / F32(sf * sf)))
    return np.array(out, dtype=np.float32)


class Cam:
    def __init__(self, fx, fy, cx, cy, bf, width, height):
        self.fx, self.fy, self.cx, self.cy, self.bf, self.width, self.height = fx, fy, cx, cy, bf, width, height


class Prm:
    def __init__(self):
        self.neighbor_dist_thresh = 2.5
        self.tri_lambda2 = F32(400.0)
        self.tri_str_thresh = F32(0.0064)
        self.ba_lambda2 = F32(400.0)
        self.tri_check_str_chi2 = True
        self.ba_first_as_prior = True
        self.sigma2_inv = default_sigma2_inv()


# ------------------------------------------------------------------ A0 / A1 / A2
def build_components(mean, cov):
    """gaussian.h:30-39, gaussian.cpp:36-63."""
    C = cov.reshape(-1, 3, 3)
    inv = np.linalg.inv(C)
    det = np.linalg.det(C)
    w, V = np.linalg.eigh(C)
    L = np.linalg.cholesky(0.5 * (inv + inv.transpose(0, 2, 1)))
    return dict(cov_inv=inv, det=det, scale=w, axis=V, sqrt_info=L, is_deg=w[:, 0] < 1e-4,
                is_salient=(w[:, 1] > 0.2) & (w[:, 2] > 0.2))


def chi2_all(mean, cov_inv, pts):
    """gaussian.cpp:65-70 for every (point, component): (N, K)."""
    d = pts[:, None, :] - mean[None, :, :]
    return np.einsum("nki,kij,nkj->nk", d, cov_inv, d)


def bh(mean0, cov0, det0, mean1, cov1, det1):
    """gmm_utils.h:30-52, broadcasting over leading dims (any dimension 2 or 3)."""
    c = (cov0 + cov1) / 2.0
    d = mean1 - mean0
    d0 = np.einsum("...i,...ij,...j->...", d, np.linalg.inv(c), d) / 8.0
    d1 = np.log(np.linalg.det(c) / np.sqrt(det0 * det1)) / 2.0
    return d0 + d1


def neighbour_rows(mean, cov, det, rows, thresh=2.5):
    """gaussian_mixture.cpp:61-78 for the given rows -> list of (idx array, dist array)."""
    C = cov.reshape(-1, 3, 3)
    out = []
    for i in rows:
        dist = bh(mean[i], C[i], det[i], mean, C, det)
        dist[i] = np.inf
        j = np.nonzero(dist < thresh)[0]
        out.append((j, dist[j]))
    return out


# ------------------------------------------------------------------ SE3 (g2o se3quat.h)
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R2q(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        q = np.array([(R[2, 1] - R[1, 2]) * 0.5 / s, (R[0, 2] - R[2, 0]) * 0.5 / s, (R[1, 0] - R[0, 1]) * 0.5 / s,
                      0.5 * s])
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = 0.5 * s
        q[3] = (R[k, j] - R[j, k]) * 0.5 / s
        q[j] = (R[j, i] + R[i, j]) * 0.5 / s
        q[k] = (R[k, i] + R[i, k]) * 0.5 / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


class SE3:
    """T = (R, t), x -> R x + t; kept as a rotation matrix (the C++ oracle keeps quaternions)."""

    def __init__(self, R, t):
        self.R, self.t = np.array(R, float), np.array(t, float)

    @staticmethod
    def from7(p):
        q = np.asarray(p[:4], float)
        q = q / np.linalg.norm(q)
        return SE3(q2R(q), p[4:7])

    def to7(self):
        return np.concatenate([R2q(self.R), self.t])

    def map(self, x):
        return x @ self.R.T + self.t

    def mul(self, o):
        return SE3(self.R @ o.R, self.R @ o.t + self.t)

    def inv(self):
        return SE3(self.R.T, -self.R.T @ self.t)

    @staticmethod
    def exp(u):
        w, v = u[:3], u[3:]
        th = np.linalg.norm(w)
        W = skew(w)
        if th < 1e-5:
            R = np.eye(3) + W + 0.5 * W @ W
            V = np.eye(3) + 0.5 * W + W @ W / 6.0
        else:
            R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
            V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
        # g2o re-normalises through a quaternion
        return SE3(q2R(R2q(R)), V @ v)

    def log(self):
        R = self.R
        d = 0.5 * (np.trace(R) - 1)
        dR = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        if abs(d) > 0.99999:
            w = 0.5 * dR
            W = skew(w)
            Vi = np.eye(3) - 0.5 * W + W @ W / 12.0
        else:
            th = np.arccos(d)
            w = th / (2 * np.sqrt(1 - d * d)) * dR
            W = skew(w)
            Vi = np.eye(3) - 0.5 * W + (1 - th / (2 * np.tan(th / 2))) / th ** 2 * W @ W
        return np.concatenate([w, Vi @ self.t])

    def adj(self):
        A = np.zeros((6, 6))
        A[:3, :3] = self.R
        A[3:, 3:] = self.R
        A[3:, :3] = skew(self.t) @ self.R
        return A


# ------------------------------------------------------------------ A3 / A4 / A5
def project_gaussian(mu, C, cam, R, t):
    """gmm_utils.cpp:121-146, pinhole_camera.cpp:68-150 -> (mean2d, cov2d, depth) or None."""
    pc = R @ mu + t
    x, y, z = pc
    if not z > 0:
        return None
    u, v = cam.fx * x / z + cam.cx, cam.fy * y / z + cam.cy
    if not (u >= 0 and v >= 0 and u < cam.width and v < cam.height):
        return None
    J = np.array([[cam.fx / z, 0, -cam.fx * x / z ** 2], [0, cam.fy / z, -cam.fy * y / z ** 2]])
    return np.array([u, v]), J @ R @ C @ R.T @ J.T, z


def render_view(mean, cov, comps, cam, pose7):
    """gaussian_mixture.cpp:271-371 -> list of dict(id, mean, cov, det, depth), depth-descending."""
    T = SE3.from7(pose7)
    R, t = T.R, T.t
    twc = -R.T @ t
    C3 = cov.reshape(-1, 3, 3)
    out = []
    cos_thr = np.cos(78.0 * np.pi / 180.0)
    for k in range(mean.shape[0]):
        if comps["is_deg"][k]:
            po = mean[k] - twc
            po = po / np.linalg.norm(po)
            if abs(po @ comps["axis"][k][:, 0]) < cos_thr:
                continue
        pg = project_gaussian(mean[k], C3[k], cam, R, t)
        if pg is None:
            continue
        m2, c2, z = pg
        w = np.linalg.eigvalsh(0.5 * (c2 + c2.T))
        if w[0] < 4.0 and w[1] < 4.0:
            continue
        g = dict(id=k, mean=m2, cov=c2, det=np.linalg.det(c2), depth=z)
        if out:
            d = np.array([bh(o["mean"], o["cov"], o["det"], g["mean"], g["cov"], g["det"]) for o in out])
            d = np.where(np.isnan(d), np.inf, d)
            j = int(np.argmin(d))
            if d[j] < 0.8:
                if g["depth"] < out[j]["depth"]:
                    out[j] = g
            else:
                out.append(g)
        else:
            out.append(g)
    order = sorted(range(len(out)), key=lambda i: -out[i]["depth"])  # stable
    return [out[i] for i in order]


def search_correspondence(view, uv, k=5):
    """gaussian_mixture.cpp:484-534 -> (cand (N,k) parent ids / -1, ncand)."""
    N = uv.shape[0]
    cand = -np.ones((N, k), np.int32)
    ncand = np.zeros(N, np.int32)
    if not view:
        return cand, ncand
    m = np.stack([g["mean"] for g in view])
    inv = np.stack([np.linalg.inv(g["cov"]) for g in view])
    for n in range(N):
        d = ((uv[n] - m) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:k]
        c = 0
        for j in order:
            dd = uv[n] - m[j]
            if dd @ inv[j] @ dd < 9.0:
                cand[n, c] = view[j]["id"]
                c += 1
        ncand[n] = c
    return cand, ncand


# ------------------------------------------------------------------ edges
def proj_stereo(pc, cam):
    iz = 1.0 / pc[2]
    u = pc[0] * iz * cam.fx + cam.cx
    return np.array([u, pc[1] * iz * cam.fy + cam.cy, u - cam.bf * iz])


def dproj(pc, cam, stereo):
    x, y, z = pc
    J = np.array([[cam.fx / z, 0, -cam.fx * x / z ** 2], [0, cam.fy / z, -cam.fy * y / z ** 2]])
    if stereo:
        J = np.vstack([J, J[0] + np.array([0, 0, cam.bf / z ** 2])])
    return J


def huber(e, delta):
    d2 = delta * delta
    if e <= d2:
        return e, 1.0
    s = np.sqrt(e)
    return 2 * s * delta - d2, delta / s


def fdelta(c):
    return float(F32(np.sqrt(c)))


# ------------------------------------------------------------------ B1
def optimize_point(pt, uvr, octave, pose7, normal, mu, proj_z2, cam, prm):
    """gmmloc_opt.cpp:260-342: 5 Gauss-Newton steps; chi2 values are those of the LAST
    computeActiveErrors(), i.e. evaluated before the 5th update (g2o does not re-evaluate)."""
    T = SE3.from7(pose7)
    s = float(prm.sigma2_inv[octave])
    lam = float(prm.tri_lambda2) * proj_z2
    x = np.array(pt, float)
    for _ in range(5):
        pc = T.map(x)
        e = uvr - proj_stereo(pc, cam)
        J = -dproj(pc, cam, True) @ T.R
        es = normal @ (x - mu)
        H = s * J.T @ J + lam * np.outer(normal, normal)
        b = -(s * J.T @ e + lam * normal * es)
        chi2_proj, chi2_str = s * e @ e, lam * es * es
        x = x + np.linalg.solve(H, b)
    res = True
    if chi2_proj > 7.815:
        res = False
    if prm.tri_check_str_chi2 and chi2_str > float(F32(prm.tri_str_thresh * prm.tri_lambda2)):
        res = False
    return res, chi2_proj, chi2_str, x


# ------------------------------------------------------------------ A8
def check_map_association(pt, uvr, octave, pose7, cands, comps, mean, nbs, cam, prm):
    """GMMLoc::checkMapAssociation (gmmloc_opt.cpp:156-258).  cands: parent component indices of the feature
    (kf->comps_[idx]); comps: build_components() dict; nbs: list of neighbour index lists (nbs_).
    Returns (component or -1, point) -- the point is moved exactly where the reference writes it."""
    pt0 = np.array(pt, float)
    cands = [int(c) for c in cands if c >= 0]
    if not cands:
        return -1, pt0
    T = SE3.from7(pose7)
    z = min(1.0, T.map(pt0)[2])
    pz2 = z * z
    run = lambda k: optimize_point(pt0, uvr, octave, pose7, comps["axis"][k][:, 0], mean[k], pz2, cam, prm)
    chi2 = lambda k, x: float((x - mean[k]) @ comps["cov_inv"][k] @ (x - mean[k]))
    best, best_v, best_x = -1, np.inf, None
    for k in cands:
        ok, c2p, _, x = run(k)
        if ok and c2p < best_v:
            best, best_v, best_x = k, c2p, x
    if best >= 0:
        ll, sel = chi2(best, best_x), best
        for n in nbs[best]:
            ln = chi2(n, best_x)
            if ln < ll:
                ll, sel = ln, n
        if sel != best:
            ok, _, _, x = run(sel)
            if ok:
                best_x = x
            else:
                sel, ll = best, chi2(best, best_x)
        if ll > 9.0:
            return -1, pt0
        return sel, best_x
    # fallback: nearest mean of the 5-NN (queryPoint); only a degenerate component is tried, and the
    # association is NOT returned even when the point is moved (:237-256)
    k = int(np.argmin(((mean - pt0) ** 2).sum(1)))
    if not comps["is_deg"][k]:
        return -1, pt0
    ok, _, _, x = run(k)
    return -1, (x if ok else pt0)


# ------------------------------------------------------------------ B2
def optimize_triangulation(x3d, pose1, uvr1, oct1, pose2, uvr2, cands1, cands2, comps, mean, cam, prm):
    """Localization::optimizeTriangulationVec (localization_opt.cpp:27-204): for every DEGENERATE candidate
    (comps1 then comps2, de-duplicated) 20 Gauss-Newton steps from the initial point on two fixed-pose
    reprojection edges (both weighted with kp1's 1/sigma^2, :132,135) + the point-to-plane edge
    (tri_lambda2); chi2 values are those of the last computeActiveErrors (before the 20th update)."""
    x0 = np.array(x3d, float)
    T1, T2 = SE3.from7(pose1), SE3.from7(pose2)
    st1, st2 = not (uvr1[2] < 0), not (uvr2[2] < 0)
    s = float(prm.sigma2_inv[oct1])
    lam = float(prm.tri_lambda2)
    th1, th2 = (7.8 if st1 else 5.991), (7.8 if st2 else 5.991)
    order = []
    for c in list(cands1) + list(cands2):
        if c >= 0 and int(c) not in order:
            order.append(int(c))
    best, best_v, best_x = -1, np.inf, None
    for k in order:
        if not comps["is_deg"][k]:
            continue
        n, mu = comps["axis"][k][:, 0], mean[k]
        x = x0.copy()
        for _ in range(20):
            H, b = lam * np.outer(n, n), -lam * n * (n @ (x - mu))
            errs = []
            for T, uvr, st in ((T1, uvr1, st1), (T2, uvr2, st2)):
                pc = T.map(x)
                d = 3 if st else 2
                e = uvr[:d] - proj_stereo(pc, cam)[:d]
                J = -dproj(pc, cam, st) @ T.R
                H = H + s * J.T @ J
                b = b - s * J.T @ e
                errs.append(s * e @ e)
            es = lam * (n @ (x - mu)) ** 2
            try:
                x = x + np.linalg.solve(H, b)
            except np.linalg.LinAlgError:  # solver failure ends optimize(); the errors above stay
                break
        ok = not (prm.tri_check_str_chi2 and es > float(F32(prm.tri_str_thresh * prm.tri_lambda2)))
        if errs[0] > th1 or errs[1] > th2:
            ok = False
        if ok and errs[0] + errs[1] < best_v:
            best, best_v, best_x = k, errs[0] + errs[1], x
    return best, (best_x if best >= 0 else x0)


# ------------------------------------------------------------------ createMapPoints (per match)
def create_map_point(pose1, kp1, depth1, oct1, pose2, kp2, depth2, oct2, cands1, cands2, comps, mean, cam, prm,
                     scale_factor=1.2):
    """Localization::createMapPoints, one epipolar match (localization_opt.cpp:286-420): parallax test,
    np.linalg.svd triangulation or stereo unprojection, B2, reprojection and scale checks.
    Returns (point or None, type 0..4, component)."""
    f32 = np.float32
    sf = [f32(1.0)]
    for _ in range(7):
        sf.append(f32(sf[-1] * f32(scale_factor)))
    fx, fy, cx, cy = f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy)
    ifx, ify = f32(1.0) / fx, f32(1.0) / fy
    mbf = f32(cam.bf)
    mb = f32(mbf / fx)
    T1, T2 = SE3.from7(pose1), SE3.from7(pose2)
    W1, W2 = T1.inv(), T2.inv()
    ur1, ur2 = f32(kp1[2]), f32(kp2[2])
    st1, st2 = bool(ur1 >= 0), bool(ur2 >= 0)
    xn1 = np.array([(kp1[0] - np.float64(cx)) * np.float64(ifx), (kp1[1] - np.float64(cy)) * np.float64(ify), 1.0])
    xn2 = np.array([(kp2[0] - np.float64(cx)) * np.float64(ifx), (kp2[1] - np.float64(cy)) * np.float64(ify), 1.0])
    r1, r2 = W1.R @ xn1, W2.R @ xn2
    cos_rays = f32(r1 @ r2 / (np.linalg.norm(r1) * np.linalg.norm(r2)))
    c1 = c2 = f32(cos_rays + f32(1))
    if st1:
        c1 = f32(np.cos(f32(f32(2) * np.arctan2(f32(mb / f32(2)), f32(depth1)))))
    elif st2:
        c2 = f32(np.cos(f32(f32(2) * np.arctan2(f32(mb / f32(2)), f32(depth2)))))
    cs = min(c1, c2)
    from_mono = False
    if cos_rays < cs and cos_rays > 0 and (st1 or st2 or np.float64(cos_rays) < 0.9998):
        M1 = np.hstack([T1.R, T1.t[:, None]])
        M2 = np.hstack([T2.R, T2.t[:, None]])
        A = np.stack([xn1[0] * M1[2] - M1[0], xn1[1] * M1[2] - M1[1], xn2[0] * M2[2] - M2[0], xn2[1] * M2[2] - M2[1]])
        vt = np.linalg.svd(A)[2][3]
        pt = vt[:3] / vt[3]
        from_mono = True
    elif st1 and c1 < c2:
        z = np.float64(f32(depth1))
        pt = W1.map(np.array([z * (kp1[0] - cam.cx) / cam.fx, z * (kp1[1] - cam.cy) / cam.fy, z]))
    elif st2 and c2 < c1:
        z = np.float64(f32(depth2))
        pt = W2.map(np.array([z * (kp2[0] - cam.cx) / cam.fx, z * (kp2[1] - cam.cy) / cam.fy, z]))
    else:
        return None, 0, -1
    b1 = np.array([kp1[0], kp1[1], kp1[2] if depth1 > 0 else -1.0])
    b2 = np.array([kp2[0], kp2[1], kp2[2] if depth2 > 0 else -1.0])
    comp, pt = optimize_triangulation(pt, pose1, b1, oct1, pose2, b2, cands1, cands2, comps, mean, cam, prm)

    def proj(T):
        pc = T.map(pt)
        if pc[2] < 0.0:
            return None
        u, v = cam.fx * (pc[0] / pc[2]) + cam.cx, cam.fy * (pc[1] / pc[2]) + cam.cy
        if not (0.0 <= u < cam.width and 0.0 <= v < cam.height and pc[2] > 0.0):
            return None
        return np.array([u, v, u - np.float64(mbf) / pc[2]])

    p1, p2 = proj(T1), proj(T2)
    if p1 is None or p2 is None:
        return pt, 0, comp
    err = lambda kp, ur, o: ((kp[:2] - o[:2]) ** 2).sum() + (0.0 if ur < 0 else (np.float64(ur) - o[2]) ** 2)
    s2 = np.float64(f32(sf[oct1] * sf[oct1]))
    if err(kp1, ur1, p1) > (7.8 if st1 else 5.991) * s2 or err(kp2, ur2, p2) > (7.8 if st2 else 5.991) * s2:
        return pt, 0, comp
    d1, d2 = f32(np.linalg.norm(pt - W1.t)), f32(np.linalg.norm(pt - W2.t))
    eps = np.finfo(np.float32).eps
    if d1 <= eps or d2 <= eps:
        return pt, 0, comp
    ratio_d, ratio_o, rf = f32(d2 / d1), f32(sf[oct1] / sf[oct2]), f32(f32(1.5) * f32(scale_factor))
    if f32(ratio_d * rf) < ratio_o or ratio_d > f32(ratio_o * rf):
        return pt, 0, comp
    return pt, (2 if comp >= 0 else 1) if from_mono else (4 if comp >= 0 else 3), comp


# ------------------------------------------------------------------ generic dense LM (B3 / B4)
class Problem:
    """Un-reduced dense Levenberg-Marquardt over free poses (6) and free points (3) with the
    g2o control flow (OptimizationAlgorithmLevenberg::solve).  Edges are dicts."""

    def __init__(self, cam, prm):
        self.cam, self.prm = cam, prm
        self.poses, self.pose_fixed = [], []
        self.points = []
        self.edges = []

    def residual(self, e):
        cam = self.cam
        k = e["kind"]
        if k in ("pose_mono", "pose_stereo", "ba_mono", "ba_stereo"):
            T = self.poses[e["pose"]]
            X = e["Xw"] if k.startswith("pose") else self.points[e["pt"]]
            pc = T.map(X)
            pr = proj_stereo(pc, cam)
            st = k.endswith("stereo")
            r = e["meas"][:3 if st else 2] - pr[:3 if st else 2]
            Jp = -dproj(pc, cam, st) @ np.hstack([-skew(pc), np.eye(3)])
            Jx = -dproj(pc, cam, st) @ T.R
            return r, Jp, Jx, pc
        if k == "deg":
            x = self.points[e["pt"]]
            return np.array([e["normal"] @ (x - e["mean"])]), None, e["normal"][None, :], None
        if k == "gauss":
            x = self.points[e["pt"]]
            Lt = e["sqrt_info"].T
            return Lt @ (x - e["mean"]), None, Lt, None
        if k == "prior":
            T = self.poses[e["pose"]]
            d = e["inv_meas"].mul(T)
            dv = d.log()
            Jr = np.zeros((6, 6))
            Jr[:3, :3] = skew(dv[:3])
            Jr[3:, 3:] = skew(dv[:3])
            Jr[:3, 3:] = skew(dv[3:])
            Jr = np.eye(6) + 0.5 * Jr
            return dv, Jr @ T.inv().adj(), None, None
        raise ValueError(k)

    def chi2(self, e):
        r = self.residual(e)[0]
        return float(r @ e["info"] @ r)

    def activate(self, level):
        self.active = [e for e in self.edges if e["level"] == level and not self._all_fixed(e)]
        fp = sorted({e["pose"] for e in self.active if "pose" in e and not self.pose_fixed[e["pose"]]})
        fx = sorted({e["pt"] for e in self.active if "pt" in e})
        self.ip = {p: 6 * i for i, p in enumerate(fp)}
        self.ix = {x: 6 * len(fp) + 3 * i for i, x in enumerate(fx)}
        self.n = 6 * len(fp) + 3 * len(fx)

    def _all_fixed(self, e):
        if "pt" in e:
            return False
        return self.pose_fixed[e["pose"]]

    def compute_errors(self):
        for e in self.active:
            e["chi2"] = self.chi2(e)

    def robust_chi2(self):
        return sum(huber(e["chi2"], e["delta"])[0] if e["robust"] else e["chi2"] for e in self.active)

    def build(self):
        H, b = np.zeros((self.n, self.n)), np.zeros(self.n)
        for e in self.active:
            r, Jp, Jx, _ = self.residual(e)
            w = huber(e["chi2"], e["delta"])[1] if e["robust"] else 1.0
            blocks = []
            if Jp is not None and not self.pose_fixed[e["pose"]]:
                blocks.append((self.ip[e["pose"]], Jp))
            if Jx is not None and "pt" in e:
                blocks.append((self.ix[e["pt"]], Jx))
            for (o1, J1) in blocks:
                b[o1:o1 + J1.shape[1]] -= w * J1.T @ e["info"] @ r
                for (o2, J2) in blocks:
                    H[o1:o1 + J1.shape[1], o2:o2 + J2.shape[1]] += w * J1.T @ e["info"] @ J2
        return H, b

    def state(self):
        return [SE3(T.R.copy(), T.t.copy()) for T in self.poses], [x.copy() for x in self.points]

    def restore(self, s):
        self.poses, self.points = s[0], s[1]

    def apply(self, dx):
        for p, o in self.ip.items():
            self.poses[p] = SE3.exp(dx[o:o + 6]).mul(self.poses[p])
        for x, o in self.ix.items():
            self.points[x] = self.points[x] + dx[o:o + 3]

    def optimize(self, iters):
        if self.n == 0:
            return -1
        done = 0
        for it in range(iters):
            self.compute_errors()
            cur = self.robust_chi2()
            H, b = self.build()
            if it == 0:
                self.lam = 1e-5 * np.max(np.abs(np.diag(H)))
                self.ni = 2.0
            rho, q = 0.0, 0
            while True:
                saved = self.state()
                try:
                    Hl = H + self.lam * np.eye(self.n)
                    np.linalg.cholesky(Hl)
                    dx = np.linalg.solve(Hl, b)
                    ok = True
                except np.linalg.LinAlgError:
                    dx, ok = np.zeros(self.n), False
                self.apply(dx)
                self.compute_errors()
                tmp = self.robust_chi2() if ok else np.finfo(float).max
                rho = (cur - tmp) / (dx @ (self.lam * dx + b) + 1e-3)
                if rho > 0 and np.isfinite(tmp):
                    a = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    self.lam *= max(1.0 / 3.0, a)
                    self.ni = 2.0
                    cur = tmp
                else:
                    self.lam *= self.ni
                    self.ni *= 2
                    self.restore(saved)
                q += 1
                if not (rho < 0 and q < 10):
                    break
            done += 1
            if q == 10 or rho == 0:
                break
        return done


def optimize_current_pose(pose7, Xw, obs, octave, cam, prm):
    """tracking_opt.cpp:21-217 -> (pose7, is_outlier, n_inliers)."""
    N = Xw.shape[0]
    pb = Problem(cam, prm)
    T0 = SE3.from7(pose7)
    pb.poses, pb.pose_fixed = [T0], [False]
    idx = []
    for i in range(N):
        if octave[i] < 0:
            continue
        mono = obs[i, 2] < 0
        s = float(prm.sigma2_inv[octave[i]])
        pb.edges.append(dict(kind="pose_mono" if mono else "pose_stereo", pose=0, Xw=Xw[i], meas=obs[i],
                             info=s * np.eye(2 if mono else 3), robust=True,
                             delta=fdelta(5.991 if mono else 7.815), level=0, chi2=0.0))
        idx.append(i)
    outl = np.zeros(N, np.uint8)
    if len(idx) < 3:
        return pose7.copy(), outl, 0
    nbad = 0
    for rnd in range(4):
        pb.poses = [SE3(T0.R.copy(), T0.t.copy())]
        pb.activate(0)
        pb.optimize(10)
        nbad = 0
        for e, i in zip(pb.edges, idx):
            if outl[i]:
                e["chi2"] = pb.chi2(e)
            thr = F32(5.991) if e["kind"] == "pose_mono" else F32(7.815)
            if F32(e["chi2"]) > thr:
                outl[i], e["level"] = 1, 1
                nbad += 1
            else:
                outl[i], e["level"] = 0, 0
            if rnd == 2:
                e["robust"] = False
        if len(pb.edges) < 10:
            break
    return pb.poses[0].to7(), outl, len(idx) - nbad


def joint_optimization(P, F, poses7, has_prior, points, assoc, obs_ptr, obs_pose, obs_uvr, obs_oct, comps, mean,
                       cam, prm):
    """localization_opt.cpp:456-925 on the flat problem (see oracle/gmmloc_oracle.cpp)."""
    pb = Problem(cam, prm)
    pb.poses = [SE3.from7(p) for p in poses7]
    pb.pose_fixed = [False] * P + [True] * F
    pb.points = [np.array(x, float) for x in points]
    L = len(points)
    gmm_deg = [None] * L
    eobs = []
    for i in range(P):
        if has_prior[i]:
            if prm.ba_first_as_prior:
                sr = 1.0 / (2.0 * np.pi / 180.0) ** 2
                info = np.diag([sr] * 3 + [1.0 / 0.01 ** 2] * 3)
                pb.edges.append(dict(kind="prior", pose=i, inv_meas=pb.poses[i].inv(), info=info, robust=False,
                                     delta=0.0, level=0, chi2=0.0))
            else:
                pb.pose_fixed[i] = True
    for l in range(L):
        a = assoc[l]
        if a >= 0:
            if comps["is_deg"][a]:
                e = dict(kind="deg", pt=l, normal=comps["axis"][a][:, 0].copy(), mean=mean[a].copy(),
                         info=float(prm.ba_lambda2) * np.eye(1), robust=False, delta=0.0, level=0, chi2=0.0)
                gmm_deg[l] = e
            else:
                e = dict(kind="gauss", pt=l, sqrt_info=comps["sqrt_info"][a], mean=mean[a].copy(), info=np.eye(3),
                         robust=False, delta=0.0, level=0, chi2=0.0)
            pb.edges.append(e)
        for o in range(obs_ptr[l], obs_ptr[l + 1]):
            mono = obs_uvr[o, 2] < 0
            s = float(prm.sigma2_inv[obs_oct[o]])
            e = dict(kind="ba_mono" if mono else "ba_stereo", pt=l, pose=int(obs_pose[o]), meas=obs_uvr[o],
                     info=s * np.eye(2 if mono else 3), robust=True, delta=fdelta(5.991 if mono else 7.815),
                     level=0, chi2=0.0)
            pb.edges.append(e)
            eobs.append(e)
    thr_str = float(F32(prm.tri_str_thresh * prm.ba_lambda2))
    pb.activate(0)
    pb.optimize(5)
    for e in gmm_deg:
        if e is None:
            continue
        e["chi2"] = pb.chi2(e)
        if e["chi2"] > thr_str:
            e["level"] = 1
        e["robust"] = False
    pb.activate(0)
    pb.optimize(5)
    for e in eobs:
        th = 5.991 if e["kind"] == "ba_mono" else 7.815
        pc = pb.poses[e["pose"]].map(pb.points[e["pt"]])
        if e["chi2"] > th or not pc[2] > 0:
            e["level"] = 1
        e["robust"] = False
    pb.activate(0)
    iters = pb.optimize(40)
    dropped = np.zeros(L, np.uint8)
    for l, e in enumerate(gmm_deg):
        if e is not None and pb.chi2(e) > thr_str:
            dropped[l] = 1
    erase = np.zeros(len(eobs), np.uint8)
    for o, e in enumerate(eobs):
        th = 5.991 if e["kind"] == "ba_mono" else 7.815
        pc = pb.poses[e["pose"]].map(pb.points[e["pt"]])
        erase[o] = 1 if (e["chi2"] > th or not pc[2] > 0) else 0
    return (np.stack([T.to7() for T in pb.poses[:P]]), np.stack(pb.points), dropped, erase, iters)


# ---- ORBmatcher::searchByProjection (orb_matcher.cpp:27-110), independent restatement ----------------
_POP8 = np.array([bin(i).count("1") for i in range(256)], np.int32)


def search_by_projection(width, height, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, mp_uvr, mp_level, mp_viewcos,
                         mp_valid, mp_desc, th=3.0, nn_ratio=0.8, scale_factor=1.2):
    """Brute force over ALL features per map point instead of the 64x48 grid walk: a feature passes
    getFeaturesInArea (frame.cpp:121-177) iff it is registered in the grid (frame.cpp:54-79), its octave is in
    [level-1, level] and |du|, |dv| < r (float); the grid only fixes the visiting ORDER (cell column, cell
    row, feature index), which decides ties of `dist < bestDist`.  Hamming distance by byte popcount table."""
    f32 = np.float32
    col_inv, row_inv = f32(64) / f32(width), f32(48) / f32(height)
    sf = [f32(1.0)]
    for _ in range(7):
        sf.append(f32(sf[-1] * f32(scale_factor)))
    NF, NP = len(feat_oct), len(mp_valid)
    # std::round = half away from zero
    rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)
    px = rnd(feat_uv[:, 0] * np.float64(col_inv))
    py = rnd(feat_uv[:, 1] * np.float64(row_inv))
    in_grid = (feat_oct >= 0) & (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    order = np.lexsort((np.arange(NF), py, px))  # primary px, then py, then index
    order = order[in_grid[order]]
    taken = np.array(feat_taken, bool).copy()
    match = -np.ones(NF, np.int32)
    n = 0
    th = f32(th)
    for m in range(NP):
        if not mp_valid[m]:
            continue
        lvl = int(mp_level[m])
        r = f32(2.5) if np.float64(f32(mp_viewcos[m])) > 0.998 else f32(4.0)
        if np.float64(th) != 1.0:
            r = f32(r * th)
        rr = f32(r * sf[lvl])
        x, y = f32(mp_uvr[m, 0]), f32(mp_uvr[m, 1])
        # empty-range early outs of getFeaturesInArea (only matter for points far outside the image)
        if int(np.floor(f32(f32(x - rr) * col_inv))) >= 64 or int(np.ceil(f32(f32(x + rr) * col_inv))) < 0:
            continue
        if int(np.floor(f32(f32(y - rr) * row_inv))) >= 48 or int(np.ceil(f32(f32(y + rr) * row_inv))) < 0:
            continue
        best, best2, lvl1, lvl2, bidx = 256, 256, -1, -1, -1
        for idx in order:
            oc = int(feat_oct[idx])
            if oc < lvl - 1 or (lvl >= 0 and oc > lvl):
                continue
            dx, dy = f32(feat_uv[idx, 0] - np.float64(x)), f32(feat_uv[idx, 1] - np.float64(y))
            if not (abs(dx) < rr and abs(dy) < rr):
                continue
            if taken[idx]:
                continue
            if feat_ur[idx] > 0:
                er = f32(abs(mp_uvr[m, 2] - np.float64(feat_ur[idx])))
                if er > rr:
                    continue
            d = int(_POP8[np.bitwise_xor(mp_desc[m], feat_desc[idx])].sum())
            if d < best:
                best2, best, lvl2, lvl1, bidx = best, d, lvl1, oc, idx
            elif d < best2:
                lvl2, best2 = oc, d
        if best <= 100:
            if lvl1 == lvl2 and f32(best) > f32(f32(nn_ratio) * f32(best2)):
                continue
            match[bidx] = m
            taken[bidx] = True
            n += 1
    return match, n


def _qrot(q, v):
    """Eigen Quaternion * Vector3 (uv = 2 q.vec x v; v + w uv + q.vec x uv)."""
    qv, w = np.asarray(q[:3], float), float(q[3])
    uv = np.cross(qv, v)
    uv = uv + uv
    return v + w * uv + np.cross(qv, uv)


def search_by_projection_frame(cam, pose_cw, pose_lw, feat_uv, feat_ur, feat_oct, feat_angle, feat_desc, feat_taken,
                               last_pt, last_valid, last_oct, last_angle, last_desc, th=7.0, mono=False,
                               check_orientation=True, scale_factor=1.2):
    """ORBmatcher::searchByProjection(CurrentFrame, LastFrame, th, bMono) (orb_matcher.cpp:410-542),
    independent restatement: brute force over all features in grid order, table popcount, histogram by
    np.bincount."""
    f32 = np.float32
    W, H = int(cam.width), int(cam.height)
    col_inv, row_inv = f32(64) / f32(W), f32(48) / f32(H)
    sf = [f32(1.0)]
    for _ in range(7):
        sf.append(f32(sf[-1] * f32(scale_factor)))
    fx, fy, cx, cy = (np.float64(f32(v)) for v in (cam.fx, cam.fy, cam.cx, cam.cy))
    mbf = f32(cam.bf)
    mb = f32(mbf / f32(cam.fx))
    NF, NL = len(feat_oct), len(last_valid)
    rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)
    px = rnd(feat_uv[:, 0] * np.float64(col_inv))
    py = rnd(feat_uv[:, 1] * np.float64(row_inv))
    in_grid = (feat_oct >= 0) & (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    order = np.lexsort((np.arange(NF), py, px))
    order = order[in_grid[order]]
    qc, tc = pose_cw[:4], pose_cw[4:]
    twc = _qrot(np.array([-qc[0], -qc[1], -qc[2], qc[3]]), tc * -1.0)
    tlc = _qrot(pose_lw[:4], twc) + pose_lw[4:]
    fwd = bool(tlc[2] > np.float64(mb)) and not mono
    bwd = bool(-tlc[2] > np.float64(mb)) and not mono
    taken = np.array(feat_taken, bool).copy()
    match = -np.ones(NF, np.int32)
    bins = {}
    th = f32(th)
    for i in range(NL):
        if not last_valid[i]:
            continue
        ptc = _qrot(qc, last_pt[i]) + tc
        xc, yc, invzc = f32(ptc[0]), f32(ptc[1]), f32(1.0 / ptc[2])
        if invzc < 0:
            continue
        u = f32(fx * np.float64(xc) * np.float64(invzc) + cx)
        v = f32(fy * np.float64(yc) * np.float64(invzc) + cy)
        if u < 0 or u > f32(W) or v < 0 or v > f32(H):
            continue
        oc0 = int(last_oct[i])
        rr = f32(th * sf[oc0])
        if fwd:
            lo, hi = oc0, -1
        elif bwd:
            lo, hi = 0, oc0
        else:
            lo, hi = oc0 - 1, oc0 + 1
        check = lo > 0 or hi >= 0
        if int(np.floor(f32(f32(u - rr) * col_inv))) >= 64 or int(np.ceil(f32(f32(u + rr) * col_inv))) < 0:
            continue
        if int(np.floor(f32(f32(v - rr) * row_inv))) >= 48 or int(np.ceil(f32(f32(v + rr) * row_inv))) < 0:
            continue
        best, bidx = 256, -1
        for idx in order:
            oc = int(feat_oct[idx])
            if check and (oc < lo or (hi >= 0 and oc > hi)):
                continue
            dx, dy = f32(feat_uv[idx, 0] - np.float64(u)), f32(feat_uv[idx, 1] - np.float64(v))
            if not (abs(dx) < rr and abs(dy) < rr):
                continue
            if taken[idx]:
                continue
            if feat_ur[idx] > 0:
                ur = f32(u - f32(mbf * invzc))
                if abs(f32(ur - f32(feat_ur[idx]))) > rr:
                    continue
            d = int(_POP8[np.bitwise_xor(last_desc[i], feat_desc[idx])].sum())
            if d < best:
                best, bidx = d, idx
        if best <= 100:
            match[bidx] = i
            taken[bidx] = True
            if check_orientation:
                rot = f32(f32(last_angle[i]) - f32(feat_angle[bidx]))
                if rot < 0.0:
                    rot = f32(rot + f32(360.0))
                b = int(rnd(np.float64(f32(rot * f32(f32(30) / f32(360.0))))))
                if b == 30:
                    b = 0
                bins[bidx] = b
    if check_orientation and bins:
        cnt = np.bincount(np.array(list(bins.values())), minlength=30)
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for b in range(30):
            c = int(cnt[b])
            if c > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, c, i2, i1, b
            elif c > m2:
                m3, m2, i3, i2 = m2, c, i2, b
            elif c > m3:
                m3, i3 = c, b
        if m2 < f32(0.1) * f32(m1):
            i2 = i3 = -1
        elif m3 < f32(0.1) * f32(m1):
            i3 = -1
        for idx, b in bins.items():
            if b not in (i1, i2, i3):
                match[idx] = -1
    return match, int((match >= 0).sum())


def search_for_triangulation(kf1, kf2, fmat, epipole, only_stereo=False, check_orientation=True, scale_factor=1.2):
    """ORBmatcher::searchForTriangulation (orb_matcher.cpp:141-293) + checkEpipolarDist (:119-139), independent restatement:
    the two feature vectors as python dicts (node -> list), table popcount, every candidate's geometric tests evaluated as
    vectors up front (they do not depend on the state of the loop), the selection itself as the literal double loop."""
    f32 = np.float32
    sf = [f32(1.0)]
    for _ in range(7):
        sf.append(f32(sf[-1] * f32(scale_factor)))
    sf = np.array(sf, f32)
    sigma2 = (sf * sf).astype(f32)
    fv1 = {int(n): kf1["node_idx"][kf1["node_ptr"][i]:kf1["node_ptr"][i + 1]] for i, n in enumerate(kf1["node_id"])}
    fv2 = {int(n): kf2["node_idx"][kf2["node_ptr"][i]:kf2["node_ptr"][i + 1]] for i, n in enumerate(kf2["node_id"])}
    N1, N2 = len(kf1["oct"]), len(kf2["oct"])
    st1, st2 = kf1["ur"] >= 0, kf2["ur"] >= 0
    F = np.asarray(fmat, np.float64).reshape(3, 3)
    ex, ey = f32(epipole[0]), f32(epipole[1])
    # per kf2 feature: squared distance to the epipole (float), its threshold, and the chi2 bound of the epipolar test
    dex = (np.float64(ex) - kf2["uv"][:, 0]).astype(f32)
    dey = (np.float64(ey) - kf2["uv"][:, 1]).astype(f32)
    oc2 = np.maximum(kf2["oct"], 0)
    near_epipole = (dex * dex + dey * dey) < (f32(100) * sf[oc2])
    bound = np.float64(3.84) * sigma2[oc2].astype(np.float64)
    matched2 = np.zeros(N2, bool)
    match = -np.ones(N1, np.int32)
    rot_bin = {}
    rnd = lambda v: float(np.floor(v + 0.5)) if v >= 0 else float(np.ceil(v - 0.5))
    for node in sorted(set(fv1) & set(fv2)):
        for idx1 in fv1[node]:
            idx1 = int(idx1)
            if kf1["has_mp"][idx1] or (only_stereo and not st1[idx1]):
                continue
            u1, v1 = kf1["uv"][idx1]
            a = u1 * F[0, 0] + v1 * F[1, 0] + F[2, 0]
            b = u1 * F[0, 1] + v1 * F[1, 1] + F[2, 1]
            c = u1 * F[0, 2] + v1 * F[1, 2] + F[2, 2]
            den = f32(a * a + b * b)
            best, bidx = 50, -1
            for idx2 in fv2[node]:
                idx2 = int(idx2)
                if matched2[idx2] or kf2["has_mp"][idx2] or (only_stereo and not st2[idx2]):
                    continue
                d = int(_POP8[np.bitwise_xor(kf1["desc"][idx1], kf2["desc"][idx2])].sum())
                if d > 50 or d > best:
                    continue
                if not st1[idx1] and not st2[idx2] and near_epipole[idx2]:
                    continue
                num = f32(a * kf2["uv"][idx2, 0] + b * kf2["uv"][idx2, 1] + c)
                if den == 0:
                    continue
                with np.errstate(over="ignore"):
                    dsqr = f32(f32(num * num) / den)
                if not (np.float64(dsqr) < bound[idx2]):
                    continue
                best, bidx = d, idx2
            if bidx >= 0:
                match[idx1] = bidx
                matched2[bidx] = True
                if check_orientation:
                    rot = f32(f32(kf1["angle"][idx1]) - f32(kf2["angle"][bidx]))
                    if rot < 0.0:
                        rot = f32(rot + f32(360.0))
                    bb = int(rnd(float(f32(rot * f32(f32(30) / f32(360.0))))))
                    rot_bin[idx1] = 0 if bb == 30 else bb
    if check_orientation and rot_bin:
        cnt = np.bincount(np.array(list(rot_bin.values())), minlength=30)
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for bb in range(30):
            cc = int(cnt[bb])
            if cc > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, cc, i2, i1, bb
            elif cc > m2:
                m3, m2, i3, i2 = m2, cc, i2, bb
            elif cc > m3:
                m3, i3 = cc, bb
        if m2 < f32(0.1) * f32(m1):
            i2 = i3 = -1
        elif m3 < f32(0.1) * f32(m1):
            i3 = -1
        for idx, bb in rot_bin.items():
            if bb not in (i1, i2, i3):
                match[idx] = -1
    return match, int((match >= 0).sum())


def search_by_bow(kf, fr, nn_ratio=0.7, check_orientation=True):
    """ORBmatcher::searchByBoW (orb_matcher.cpp:295-408), independent restatement: the two feature vectors as python dicts,
    the Hamming distances of a node as ONE table-popcount matrix, the claimed features masked out of it, best / second best by a
    stable argsort (the reference keeps the FIRST of equal distances as best and lets a later equal one become second best),
    acceptance and the rotation histogram as in the source.  -> (match21 [N2]: key-frame feature or -1, nmatches)."""
    f32 = np.float32
    fv1 = {int(n): kf["node_idx"][kf["node_ptr"][i]:kf["node_ptr"][i + 1]] for i, n in enumerate(kf["node_id"])}
    fv2 = {int(n): fr["node_idx"][fr["node_ptr"][i]:fr["node_ptr"][i + 1]] for i, n in enumerate(fr["node_id"])}
    N2 = len(fr["angle"])
    match = -np.ones(N2, np.int32)
    rot_bin = {}
    rnd = lambda v: float(np.floor(v + 0.5)) if v >= 0 else float(np.ceil(v - 0.5))
    for node in sorted(set(fv1) & set(fv2)):
        cand = np.asarray(fv2[node], np.int64)
        if len(cand) == 0:
            continue
        dmat = None
        for q in fv1[node]:
            q = int(q)
            if not kf["has_mp"][q]:
                continue
            if dmat is None:  # distances of every key-frame feature of the node to every frame feature of the node
                qs = np.asarray(fv1[node], np.int64)
                dmat = _POP8[np.bitwise_xor(kf["desc"][qs][:, None, :], fr["desc"][cand][None, :, :])].sum(2).astype(np.int64)
                row_of = {int(x): i for i, x in enumerate(qs)}
            d = dmat[row_of[q]]
            free = match[cand] < 0
            if not free.any():
                continue
            dd, cc = d[free], cand[free]
            order = np.argsort(dd, kind="stable")
            b1 = int(dd[order[0]])
            b2 = int(dd[order[1]]) if len(order) > 1 else 256
            bidx = int(cc[order[0]])
            if b1 <= 50 and f32(b1) < f32(nn_ratio) * f32(b2):
                match[bidx] = q
                if check_orientation:
                    rot = f32(f32(kf["angle"][q]) - f32(fr["angle"][bidx]))
                    if rot < 0.0:
                        rot = f32(rot + f32(360.0))
                    bb = int(rnd(float(f32(rot * f32(f32(30) / f32(360.0))))))
                    rot_bin[bidx] = 0 if bb == 30 else bb
    if check_orientation and rot_bin:
        cnt = np.bincount(np.array(list(rot_bin.values())), minlength=30)
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for bb in range(30):
            c = int(cnt[bb])
            if c > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, c, i2, i1, bb
            elif c > m2:
                m3, m2, i3, i2 = m2, c, i2, bb
            elif c > m3:
                m3, i3 = c, bb
        if m2 < f32(0.1) * f32(m1):
            i2 = i3 = -1
        elif m3 < f32(0.1) * f32(m1):
            i3 = -1
        for idx, bb in rot_bin.items():
            if bb not in (i1, i2, i3):
                match[idx] = -1
    return match, int((match >= 0).sum())


def fuse_search(width, height, feat_uv, feat_ur, feat_oct, feat_desc, mp_uvr, mp_level, mp_valid, mp_desc, th=3.0, scale_factor=1.2):
    """Localization::fuseObservations (localization.cpp:226-318), the matching half, independent restatement: all features at once
    as vectors (window test in float, level band, the pixel / disparity chi2 against 5.99 / 7.8, byte-table popcount); the grid of
    the reference only fixes the visiting ORDER (cell column, cell row, feature index), which decides ties: the minimum is taken
    over the candidates in that order with a stable argmin.  -> (best_idx [NP] or -1, best_dist [NP], matched)."""
    f32 = np.float32
    col_inv, row_inv = f32(64) / f32(width), f32(48) / f32(height)
    sf = [f32(1.0)]
    for _ in range(7):
        sf.append(f32(sf[-1] * f32(scale_factor)))
    s2i = np.array([f32(1.0)] + [f32(f32(1.0) / f32(v * v)) for v in sf[1:]], f32)
    NF, NP = len(feat_oct), len(mp_valid)
    rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)
    px = rnd(feat_uv[:, 0] * np.float64(col_inv))
    py = rnd(feat_uv[:, 1] * np.float64(row_inv))
    in_grid = (feat_oct >= 0) & (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    order = np.lexsort((np.arange(NF), py, px))
    order = order[in_grid[order]]
    fu, fv, fr, fo = feat_uv[order, 0], feat_uv[order, 1], feat_ur[order].astype(f32), feat_oct[order]
    best_idx = -np.ones(NP, np.int32)
    best_dist = np.full(NP, 256, np.int32)
    n = 0
    for m in range(NP):
        if not mp_valid[m]:
            continue
        lvl = int(mp_level[m])
        rr = f32(f32(th) * sf[lvl])
        x, y = f32(mp_uvr[m, 0]), f32(mp_uvr[m, 1])
        if int(np.floor(f32(f32(x - rr) * col_inv))) >= 64 or int(np.ceil(f32(f32(x + rr) * col_inv))) < 0:
            continue
        if int(np.floor(f32(f32(y - rr) * row_inv))) >= 48 or int(np.ceil(f32(f32(y + rr) * row_inv))) < 0:
            continue
        # the cell window of getFeaturesInArea (a feature outside it cannot pass the float window test: its cell is its rounded position)
        cx0 = max(0, int(np.floor(f32(f32(x - rr) * col_inv)))); cx1 = min(63, int(np.ceil(f32(f32(x + rr) * col_inv))))
        cy0 = max(0, int(np.floor(f32(f32(y - rr) * row_inv)))); cy1 = min(47, int(np.ceil(f32(f32(y + rr) * row_inv))))
        pxo, pyo = px[order], py[order]
        win = (pxo >= cx0) & (pxo <= cx1) & (pyo >= cy0) & (pyo <= cy1)
        dxf = (fu - np.float64(x)).astype(f32)
        dyf = (fv - np.float64(y)).astype(f32)
        win &= (np.abs(dxf) < rr) & (np.abs(dyf) < rr)
        win &= (fo >= lvl - 1) & (fo <= lvl)
        dx, dy, dz = fu - mp_uvr[m, 0], fv - mp_uvr[m, 1], fr.astype(np.float64) - mp_uvr[m, 2]
        err = np.where(fr < 0, dx * dx + dy * dy, dx * dx + dy * dy + dz * dz) * s2i[np.clip(fo, 0, 7)].astype(np.float64)
        win &= ~(err > np.where(fr >= 0, 7.8, 5.99))
        if not win.any():
            continue
        cand = np.nonzero(win)[0]
        d = _POP8[np.bitwise_xor(mp_desc[m][None, :], feat_desc[order[cand]])].sum(1)
        k = int(np.argmin(d))  # first minimum in visiting order
        best_dist[m] = int(d[k])
        if d[k] <= 50:
            best_idx[m] = int(order[cand[k]])
            n += 1
    return best_idx, best_dist, n


def _libm_logf():
    import ctypes
    import ctypes.util
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    m.logf.restype = ctypes.c_float
    m.logf.argtypes = [ctypes.c_float]
    return m.logf


def project_map_points(cam, pose_cw, t_wc, pos, normal, max_dist, min_dist, cand, scale_factor=1.2):
    """Frame::project3 (frame.cpp:98-119, pinhole_camera.cpp:46-66, 128-150) + MapPoint::checkScaleAndVisible (mappoint.cpp:257-303),
    independent restatement: every test as a mask over all points at once; the float logarithm of the level is the host libm's logf
    (what std::log(float) calls), one call per surviving point.  -> (uvr [NP,3], level [NP], view_cos [NP], dist [NP], in_view [NP])."""
    f32 = np.float32
    fx, fy, cx, cy = (np.float64(f32(v)) for v in (cam.fx, cam.fy, cam.cx, cam.cy))
    mbf = np.float64(f32(cam.bf))
    logf = _libm_logf()
    sfl = f32(logf(f32(scale_factor)))
    pos, normal = np.asarray(pos, float), np.asarray(normal, float)
    NP = len(cand)
    q, t = np.asarray(pose_cw[:4], float), np.asarray(pose_cw[4:], float)
    uv = np.cross(q[None, :3], pos)
    uv = uv + uv
    ptc = pos + q[3] * uv + np.cross(q[None, :3], uv) + t[None, :]
    with np.errstate(all="ignore"):
        z = ptc[:, 2]
        rz = 1.0 / z
        u = fx * (ptc[:, 0] * rz) + cx
        v = fy * (ptc[:, 1] * rz) + cy
        ok = np.asarray(cand, bool) & ~(z < 0) & (u >= 0) & (v >= 0) & (u < float(cam.width)) & (v < float(cam.height)) & (z > 0)
        ur = u - mbf / z
        vec = pos - np.asarray(t_wc, float)[None, :]
        dist = np.sqrt(vec[:, 0] * vec[:, 0] + vec[:, 1] * vec[:, 1] + vec[:, 2] * vec[:, 2]).astype(f32)
        mx, mn = np.asarray(max_dist, f32), np.asarray(min_dist, f32)
        ok &= ~((dist < f32(0.8) * mn) | (dist > f32(1.2) * mx))
        vcos = ((vec[:, 0] * normal[:, 0] + vec[:, 1] * normal[:, 1] + vec[:, 2] * normal[:, 2]) / dist.astype(np.float64)).astype(f32)
        ok &= ~(vcos < f32(0.5))
        ratio = mx / dist
    level = np.zeros(NP, np.int32)
    for m in np.nonzero(ok)[0]:
        lq = np.ceil(f32(f32(logf(ratio[m])) / sfl))
        level[m] = int(min(max(lq, 0), 7)) if np.isfinite(lq) else 0
    uvr = np.where(ok[:, None], np.stack([u, v, ur], 1), 0.0)
    return uvr, level, np.where(ok, vcos.astype(np.float64), 0.0), np.where(ok, dist.astype(np.float64), 0.0), ok.astype(np.uint8)
