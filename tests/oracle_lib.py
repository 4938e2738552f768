"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class orc_camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("bf", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class orc_params(C.Structure):
    _fields_ = [("neighbor_dist_thresh", C.c_double), ("tri_lambda2", C.c_float),
                ("tri_str_thresh", C.c_float), ("ba_lambda2", C.c_float),
                ("tri_check_str_chi2", C.c_int32), ("ba_first_as_prior", C.c_int32),
                ("sigma2_inv", C.c_float * 8)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self, lib, nf):
        self.lib, self.nf = lib, nf
        lib.orc_gmm_create.restype = C.c_void_p
        self.prm = orc_params()
        lib.orc_default_params(C.byref(self.prm))

    def camera(self, cam):
        return orc_camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.width, cam.height)

    # ---- gmm ----
    def gmm_create(self, mean, cov):
        mean, cov = _f64(mean), _f64(cov)
        return C.c_void_p(self.lib.orc_gmm_create(_p(mean), _p(cov), mean.shape[0]))

    def gmm_destroy(self, h):
        self.lib.orc_gmm_destroy(h)

    def gmm_get(self, h):
        K = self.lib.orc_gmm_count(h)
        out = dict(cov_inv=np.zeros((K, 9)), det=np.zeros(K), scale=np.zeros((K, 3)), axis=np.zeros((K, 9)),
                   sqrt_info=np.zeros((K, 9)), flags=np.zeros(K, np.uint8))
        self.lib.orc_gmm_get(h, _p(out["cov_inv"]), _p(out["det"]), _p(out["scale"]), _p(out["axis"]),
                             _p(out["sqrt_info"]), _p(out["flags"]))
        return out

    def neighbours(self, h, thresh=2.5):
        K = self.lib.orc_gmm_count(h)
        ptr = np.zeros(K + 1, np.int32)
        nnz = self.lib.orc_gmm_neighbours(h, C.c_double(thresh), _p(ptr), None, None)
        col, dist = np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.orc_gmm_neighbours(h, C.c_double(thresh), _p(ptr), _p(col), _p(dist))
        return ptr, col, dist

    def neighbour_rows(self, h, r0, r1, thresh=2.5, cap=1 << 16):
        ptr = np.zeros(r1 - r0 + 1, np.int32)
        col, dist = np.zeros(cap, np.int32), np.zeros(cap)
        nnz = self.lib.orc_gmm_neighbour_rows(h, C.c_double(thresh), r0, r1, _p(ptr), _p(col), _p(dist), cap)
        assert nnz <= cap
        return ptr, col[:nnz], dist[:nnz]

    def associate3d(self, h, pts):
        pts = _f64(pts)
        N = pts.shape[0]
        idx, d2 = np.zeros(N, np.int32), np.zeros(N)
        self.lib.orc_associate3d(h, _p(pts), N, _p(idx), _p(d2))
        return idx, d2

    def chi2(self, h, comp, pts):
        pts, comp = _f64(pts), _i32(comp)
        out = np.zeros(pts.shape[0])
        self.lib.orc_chi2(h, _p(comp), _p(pts), pts.shape[0], _p(out))
        return out

    def knn3d(self, h, pts, k=5):
        pts = _f64(pts)
        N = pts.shape[0]
        idx, dist, cnt = np.zeros((N, k), np.int32), np.zeros((N, k)), np.zeros(N, np.int32)
        self.lib.orc_knn3d(h, _p(pts), N, k, _p(idx), _p(dist), _p(cnt))
        return idx, dist, cnt

    def render_view(self, h, cam, pose, cap=8192):
        pose = _f64(pose)
        ids, m2, c2, dep = np.zeros(cap, np.int32), np.zeros((cap, 2)), np.zeros((cap, 4)), np.zeros(cap)
        c = self.camera(cam)
        V = self.lib.orc_render_view(h, C.byref(c), _p(pose), cap, _p(ids), _p(m2), _p(c2), _p(dep))
        return ids[:V].copy(), m2[:V].copy(), c2[:V].copy(), dep[:V].copy()

    def search_correspondence(self, h, uv, k=5):
        uv = _f64(uv)
        N = uv.shape[0]
        cand, ncand = np.zeros((N, k), np.int32), np.zeros(N, np.int32)
        self.lib.orc_search_correspondence(h, _p(uv), N, k, _p(cand), _p(ncand))
        return cand, ncand

    def optimize_point(self, h, cam, pts, uvr, octave, poses, comp, proj_z2, prm=None):
        pts, uvr, poses, proj_z2 = _f64(pts), _f64(uvr), _f64(poses), _f64(proj_z2)
        octave, comp = _i32(octave), _i32(comp)
        N = pts.shape[0]
        res, c2p, c2s, est = np.zeros(N, np.uint8), np.zeros(N), np.zeros(N), np.zeros((N, 3))
        c = self.camera(cam)
        self.lib.orc_optimize_point(h, C.byref(c), C.byref(prm or self.prm), N, _p(pts), _p(uvr), _p(octave),
                                    _p(poses), _p(comp), _p(proj_z2), _p(res), _p(c2p), _p(c2s), _p(est))
        return res, c2p, c2s, est

    def check_map_association(self, h, cam, pose, pts, uvr, octave, cand, ncand, prm=None):
        pts = _f64(pts).copy()
        uvr, pose = _f64(uvr), _f64(pose)
        octave, cand, ncand = _i32(octave), _i32(cand), _i32(ncand)
        N, k = cand.shape
        out = np.zeros(N, np.int32)
        c = self.camera(cam)
        self.lib.orc_check_map_association(h, C.byref(c), C.byref(prm or self.prm), _p(pose), N, _p(pts), _p(uvr),
                                           _p(octave), _p(cand), _p(ncand), k, _p(out))
        return out, pts

    def optimize_triangulation(self, h, cam, x3d, pose1, uvr1, oct1, pose2, uvr2, oct2, cand1, n1, cand2, n2,
                               prm=None):
        x3d = _f64(x3d).copy()
        pose1, uvr1, pose2, uvr2 = _f64(pose1), _f64(uvr1), _f64(pose2), _f64(uvr2)
        oct1, oct2, cand1, n1, cand2, n2 = _i32(oct1), _i32(oct2), _i32(cand1), _i32(n1), _i32(cand2), _i32(n2)
        N, k = cand1.shape
        out = np.zeros(N, np.int32)
        c = self.camera(cam)
        self.lib.orc_optimize_triangulation(h, C.byref(c), C.byref(prm or self.prm), N, _p(x3d), _p(pose1), _p(uvr1),
                                            _p(oct1), _p(pose2), _p(uvr2), _p(oct2), _p(cand1), _p(n1), _p(cand2),
                                            _p(n2), k, _p(out))
        return out, x3d

    def create_map_points(self, h, cam, pose1, uvr1, depth1, oct1, pose2, uvr2, depth2, oct2, cand1, n1, cand2, n2,
                          scale_factor=1.2, prm=None):
        """Localization::createMapPoints per-match block -> (x3d [N,3], type [N], comp [N])."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        pose1, uvr1, pose2, uvr2 = _f64(pose1), _f64(uvr1), _f64(pose2), _f64(uvr2)
        depth1, depth2 = f32(depth1), f32(depth2)
        oct1, oct2, cand1, n1, cand2, n2 = _i32(oct1), _i32(oct2), _i32(cand1), _i32(n1), _i32(cand2), _i32(n2)
        N, k = cand1.shape
        x3d, typ, comp = np.zeros((N, 3)), np.zeros(N, np.int32), np.zeros(N, np.int32)
        c = self.camera(cam)
        self.lib.orc_create_map_points(h, C.byref(c), C.byref(prm or self.prm), C.c_float(scale_factor), N, _p(pose1),
                                       _p(uvr1), _p(depth1), _p(oct1), _p(pose2), _p(uvr2), _p(depth2), _p(oct2),
                                       _p(cand1), _p(n1), _p(cand2), _p(n2), k, _p(x3d), _p(typ), _p(comp))
        return x3d, typ, comp

    def optimize_current_pose(self, cam, pose, Xw, obs, octave, prm=None):
        pose = _f64(pose).copy()
        Xw, obs = _f64(Xw), _f64(obs)
        octave = _i32(octave)
        N = Xw.shape[0]
        has = (octave >= 0).astype(np.uint8)
        oc = np.maximum(octave, 0).astype(np.int32)
        outl = np.zeros(N, np.uint8)
        c = self.camera(cam)
        n = self.lib.orc_optimize_current_pose(C.byref(c), C.byref(prm or self.prm), _p(pose), N, _p(Xw), _p(obs),
                                               _p(oc), _p(has), _p(outl))
        return pose, outl, n

    def joint_optimization(self, h, cam, P, F, poses, has_prior, points, assoc, obs_ptr, obs_pose, obs_uvr,
                           obs_oct, prm=None, stop=0):
        poses, points, obs_uvr = _f64(poses).copy(), _f64(points).copy(), _f64(obs_uvr)
        has_prior = np.ascontiguousarray(has_prior, np.uint8)
        assoc, obs_ptr, obs_pose, obs_oct = _i32(assoc), _i32(obs_ptr), _i32(obs_pose), _i32(obs_oct)
        L, nobs = points.shape[0], obs_pose.shape[0]
        dropped, erase = np.zeros(L, np.uint8), np.zeros(max(nobs, 1), np.uint8)
        c = self.camera(cam)
        it = self.lib.orc_joint_optimization_stop(h, C.byref(c), C.byref(prm or self.prm), P, F, L, nobs, _p(poses),
                                                  _p(has_prior), _p(points), _p(assoc), _p(obs_ptr), _p(obs_pose),
                                                  _p(obs_uvr), _p(obs_oct), _p(dropped), _p(erase), int(stop))
        return poses, points, dropped, erase[:nobs], it

    def search_by_projection(self, width, height, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, mp_uvr, mp_level,
                             mp_viewcos, mp_valid, mp_desc, th=3.0, nn_ratio=0.8, scale_factor=1.2):
        """ORBmatcher::searchByProjection for one frame -> (feat_match int32 [NF], nmatches)."""
        feat_uv, mp_uvr = _f64(feat_uv), _f64(mp_uvr)
        feat_ur = np.ascontiguousarray(feat_ur, dtype=np.float32)
        feat_oct = _i32(feat_oct)
        feat_desc = np.ascontiguousarray(feat_desc, dtype=np.uint8)
        mp_desc = np.ascontiguousarray(mp_desc, dtype=np.uint8)
        feat_taken = np.ascontiguousarray(feat_taken, dtype=np.uint8)
        mp_valid = np.ascontiguousarray(mp_valid, dtype=np.uint8)
        mp_level, mp_viewcos = _f64(mp_level), _f64(mp_viewcos)
        NF, NP = feat_uv.shape[0], mp_uvr.shape[0]
        out = np.zeros(NF, np.int32)
        self.lib.orc_search_by_projection.restype = C.c_int
        n = self.lib.orc_search_by_projection(C.c_int(width), C.c_int(height), C.c_float(scale_factor), NF, _p(feat_uv),
                                              _p(feat_ur), _p(feat_oct), _p(feat_desc), _p(feat_taken), NP, _p(mp_uvr),
                                              _p(mp_level), _p(mp_viewcos), _p(mp_valid), _p(mp_desc), C.c_float(th),
                                              C.c_float(nn_ratio), _p(out))
        return out, int(n)

    def search_by_projection_frame(self, cam, pose_cw, pose_lw, feat_uv, feat_ur, feat_oct, feat_angle, feat_desc,
                                   feat_taken, last_pt, last_valid, last_oct, last_angle, last_desc, th=7.0, mono=False,
                                   check_orientation=True, scale_factor=1.2):
        """ORBmatcher::searchByProjection(CurrentFrame, LastFrame, th, bMono) -> (feat_match [NF], nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        feat_uv, last_pt, pose_cw, pose_lw = _f64(feat_uv), _f64(last_pt), _f64(pose_cw), _f64(pose_lw)
        feat_ur, feat_angle, last_angle = f32(feat_ur), f32(feat_angle), f32(last_angle)
        feat_oct, last_oct = _i32(feat_oct), _i32(last_oct)
        feat_desc, feat_taken, last_valid, last_desc = u8(feat_desc), u8(feat_taken), u8(last_valid), u8(last_desc)
        NF, NL = feat_uv.shape[0], last_pt.shape[0]
        out = np.zeros(NF, np.int32)
        c = self.camera(cam)
        self.lib.orc_search_by_projection_frame.restype = C.c_int
        n = self.lib.orc_search_by_projection_frame(C.byref(c), C.c_float(scale_factor), _p(pose_cw), _p(pose_lw), NF,
                                                    _p(feat_uv), _p(feat_ur), _p(feat_oct), _p(feat_angle), _p(feat_desc),
                                                    _p(feat_taken), NL, _p(last_pt), _p(last_valid), _p(last_oct),
                                                    _p(last_angle), _p(last_desc), C.c_float(th), int(bool(mono)),
                                                    int(bool(check_orientation)), _p(out))
        return out, int(n)

    def search_for_triangulation(self, kf1, kf2, fmat, epipole, only_stereo=False, check_orientation=True, scale_factor=1.2):
        """ORBmatcher::searchForTriangulation on one key-frame pair (dicts: uv, ur, oct, angle, desc, has_mp, node_id,
        node_ptr, node_idx) -> (match12 [N1], nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        args = []
        keep = []
        for kf in (kf1, kf2):
            arrs = [_f64(kf["uv"]), f32(kf["ur"]), _i32(kf["oct"]), f32(kf["angle"]), u8(kf["desc"]), u8(kf["has_mp"]),
                    _i32(kf["node_id"]), _i32(kf["node_ptr"]), _i32(kf["node_idx"])]
            keep.append(arrs)
            args += [arrs[0].shape[0]] + [_p(a) for a in arrs[:6]] + [len(arrs[6])] + [_p(a) for a in arrs[6:]]
        fm, ep = _f64(fmat), f32(epipole)
        out = np.zeros(keep[0][0].shape[0], np.int32)
        self.lib.orc_search_for_triangulation.restype = C.c_int
        n = self.lib.orc_search_for_triangulation(C.c_float(scale_factor), *args, _p(fm), _p(ep), int(bool(only_stereo)),
                                                  int(bool(check_orientation)), _p(out))
        return out, int(n)

    def search_by_bow(self, kf, fr, nn_ratio=0.7, check_orientation=True):
        """ORBmatcher::searchByBoW on one key-frame / frame pair (dicts: angle, desc, has_mp (key-frame: valid map point),
        node_id, node_ptr, node_idx) -> (match21 [N2]: key-frame feature or -1, nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        a1 = [f32(kf["angle"]), u8(kf["desc"]), u8(kf["has_mp"]), _i32(kf["node_id"]), _i32(kf["node_ptr"]), _i32(kf["node_idx"])]
        a2 = [f32(fr["angle"]), u8(fr["desc"]), _i32(fr["node_id"]), _i32(fr["node_ptr"]), _i32(fr["node_idx"])]
        out = np.zeros(len(a2[0]), np.int32)
        self.lib.orc_search_by_bow.restype = C.c_int
        n = self.lib.orc_search_by_bow(C.c_float(nn_ratio), int(bool(check_orientation)), len(a1[0]), _p(a1[0]), _p(a1[1]), _p(a1[2]),
                                       len(a1[3]), _p(a1[3]), _p(a1[4]), _p(a1[5]), len(a2[0]), _p(a2[0]), _p(a2[1]), len(a2[2]),
                                       _p(a2[2]), _p(a2[3]), _p(a2[4]), _p(out))
        return out, int(n)

    def fuse_search(self, width, height, feat_uv, feat_ur, feat_oct, feat_desc, mp_uvr, mp_level, mp_valid, mp_desc, th=3.0, scale_factor=1.2):
        """Localization::fuseObservations, matching half, one key-frame -> (best_idx [NP], best_dist [NP], matched)."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        a = [_f64(feat_uv), f32(feat_ur), _i32(feat_oct), u8(feat_desc), _f64(mp_uvr), _i32(mp_level), u8(mp_valid), u8(mp_desc)]
        NF, NP = len(a[2]), len(a[5])
        bi, bd = np.zeros(NP, np.int32), np.zeros(NP, np.int32)
        self.lib.orc_fuse_search.restype = C.c_int
        n = self.lib.orc_fuse_search(int(width), int(height), C.c_float(scale_factor), NF, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), NP,
                                     _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), C.c_float(th), _p(bi), _p(bd))
        return bi, bd, int(n)

    def project_map_points(self, cam, pose_cw, t_wc, pos, normal, max_dist, min_dist, cand, scale_factor=1.2):
        """Frame::project3 + MapPoint::checkScaleAndVisible, one frame -> (uvr, level, view_cos, dist, in_view, count)."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        a = [_f64(pose_cw), _f64(t_wc), _f64(pos), _f64(normal), f32(max_dist), f32(min_dist), np.ascontiguousarray(cand, dtype=np.uint8)]
        NP = len(a[6])
        uvr, lvl, vc, dd, iv = np.zeros((NP, 3)), np.zeros(NP, np.int32), np.zeros(NP), np.zeros(NP), np.zeros(NP, np.uint8)
        self.lib.orc_project_map_points.restype = C.c_int
        n = self.lib.orc_project_map_points(C.byref(self.camera(cam)), C.c_float(scale_factor), _p(a[0]), _p(a[1]), NP, _p(a[2]), _p(a[3]), _p(a[4]),
                                            _p(a[5]), _p(a[6]), _p(uvr), _p(lvl), _p(vc), _p(dd), _p(iv))
        return uvr, lvl, vc, dd, iv, int(n)

    def pose_twc(self, pose_cw):
        out = np.zeros(3)
        self.lib.orc_pose_twc(_p(_f64(pose_cw)), _p(out))
        return out

    def se3_exp(self, u):
        out = np.zeros(7)
        self.lib.orc_se3_exp(_p(_f64(u)), _p(out))
        return out

    def se3_log(self, pose):
        out = np.zeros(6)
        self.lib.orc_se3_log(_p(_f64(pose)), _p(out))
        return out

    # ---- real nanoflann (oracle/_ref) ----
    def nanoflann_tree3d(self, pts):
        """Persistent kd-tree over 3-D points (as the reference builds once per map) -> handle for nanoflann_tree_knn."""
        assert self.nf is not None, "oracle/_ref/libnanoflann_ref.so missing"
        self.nf.nfref_tree3d_create.restype = C.c_void_p
        pts = _f64(pts)
        return C.c_void_p(self.nf.nfref_tree3d_create(_p(pts), pts.shape[0]))

    def nanoflann_tree_knn(self, tree, q, k):
        q = _f64(q)
        idx, dist = np.zeros((q.shape[0], k), np.int32), np.zeros((q.shape[0], k))
        self.nf.nfref_tree3d_knn(tree, _p(q), q.shape[0], k, _p(idx), _p(dist))
        return idx, dist

    def nanoflann_tree_destroy(self, tree):
        self.nf.nfref_tree3d_destroy(tree)

    def nanoflann_knn(self, pts, q, k):
        assert self.nf is not None, "oracle/_ref/libnanoflann_ref.so missing"
        pts, q = _f64(pts), _f64(q)
        dim = pts.shape[1]
        nq = q.shape[0]
        idx, dist, cnt = np.zeros((nq, k), np.int32), np.zeros((nq, k)), np.zeros(nq, np.int32)
        fn = self.nf.nfref_knn2d if dim == 2 else self.nf.nfref_knn3d
        fn(_p(pts), pts.shape[0], _p(q), nq, k, _p(idx), _p(dist), _p(cnt))
        return idx, dist, cnt


_cached = None


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR])


def load_native():
    """The same oracle source built on THIS machine with the reference's flags (-O3 -march=native,
    gmmloc/CMakeLists.txt:7) for the CPU-baseline timing of bench.py; None when it cannot be built here."""
    so = os.path.join(ODIR, "_native", "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", ODIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(so)
    except (subprocess.CalledProcessError, OSError):
        return None
    nfp = os.path.join(ODIR, "_ref", "libnanoflann_ref.so")
    return Oracle(lib, C.CDLL(nfp) if os.path.exists(nfp) else None)


def load():
    global _cached
    if _cached is not None:
        return _cached
    so = os.path.join(ODIR, "liboracle.so")
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    nfp = os.path.join(ODIR, "_ref", "libnanoflann_ref.so")
    nf = C.CDLL(nfp) if os.path.exists(nfp) else None
    _cached = Oracle(lib, nf)
    return _cached
