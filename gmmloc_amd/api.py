"""Host-side mirror of the reference's interface for the hot path, over the C-ABI.

Names follow the reference (``GMM.renderView`` + ``searchCorrespondence`` ->
``GMM.search2d``, ``GMM.queryPoint``, ``optimize_current_pose`` =
``Tracking::optimizeCurrentPose`` ...).  Device buffers are torch CUDA tensors
(PyTorch-ROCm is used for memory, streams and torch.distributed only); every
function launches hand-written HIP kernels through ``libgmmloc_hip.so``.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

ASSOC_BRUTE = 0
ASSOC_KNN5_EUCLID = 1
ASSOC_EXHAUSTIVE = 2

F_MEAN, F_COV, F_COV_INV, F_DET, F_SCALE, F_AXIS, F_SQRT_INFO, F_FLAGS, F_NBS_PTR, F_NBS_IDX, F_NBS_DIST = range(11)
TIMER_ASSOC, TIMER_REFINE_POSE, TIMER_BA, TIMER_BA_PREP = 0, 1, 2, 3
COUNTER_BA_REDONE, COUNTER_MATCH_ROUNDS, COUNTER_MATCH_UNITS = 0, 1, 2


class GLError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise GLError("libgmmloc_hip error %d: %s" % (rc, _lib.load().gl_last_error_string().decode()))


@dataclass
class Camera:
    """PinholeCamera + camera::bf.  Defaults: gmmloc_ros/cfg/v1.yaml:5-17 read as float
    (config.h:38: `extern float fx, fy, cx, cy`)."""
    fx: float = float(np.float32(435.2046959714599))
    fy: float = float(np.float32(435.2046959714599))
    cx: float = float(np.float32(367.4517211914062))
    cy: float = float(np.float32(252.2008514404297))
    bf: float = float(np.float32(47.90639384423901))
    width: int = 752
    height: int = 480

    def c(self):
        return _lib.gl_camera(self.fx, self.fy, self.cx, self.cy, self.bf, self.width, self.height)


class Params:
    """Hot-path config values (config.h:31-89); defaults = cfg/v1.yaml."""

    def __init__(self, **kw):
        self._c = _lib.gl_params()
        _lib.load().gl_default_params(C.byref(self._c))
        for k, v in kw.items():
            setattr(self._c, k, v)

    def c(self):
        return self._c

    @property
    def sigma2_inv(self):
        return np.array(list(self._c.sigma2_inv), dtype=np.float32)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device buffers must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


class Context:
    """gl_ctx_t: device + HIP stream + scratch.

    The context owns a dedicated torch stream (``ctx.stream``) and launches every kernel on
    it.  (torch's default stream is NOT ordered with kernels an external library puts on
    HIP's null stream -- measured on the MI355X box -- so the null stream is never used.)
    The wrappers below order the context stream against torch's current stream with
    events (``_enter`` / ``_exit``); under ``with torch.cuda.stream(ctx.stream):`` both are
    no-ops and calls are fully asynchronous.
    """

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.device = device
        with torch.cuda.device(device):
            self.stream = torch.cuda.Stream(device)
        h = C.c_void_p()
        _check(self.lib.gl_ctx_create(device, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h

    def _enter(self):
        cur = self.torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)

    def _exit(self):
        cur = self.torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def synchronize(self):
        _check(self.lib.gl_ctx_synchronize(self.h))

    def set_option(self, name, value):
        """gl_ctx_set_option: tuning / test knob of this context (see include/gmmloc_hip.h)."""
        _check(self.lib.gl_ctx_set_option(self.h, name.encode(), float(value)))

    def get_option(self, name):
        v = C.c_double()
        _check(self.lib.gl_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def timing(self, on):
        _check(self.lib.gl_ctx_timing_enable(self.h, 1 if on else 0))

    def timing_read(self, timer, reset=True):
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.gl_ctx_timing_read(self.h, timer, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def counter_read(self, counter=0, reset=True):
        """gl_ctx_counter_read: 0 = GL_COUNTER_BA_REDONE (frames of latency-shape launches redone by the follow-up kernel),
        1 / 2 = GL_COUNTER_MATCH_ROUNDS / _UNITS (rounds of the matchers' owner fixed point, and the frames / pairs they ran on)."""
        v = C.c_int64(0)
        _check(self.lib.gl_ctx_counter_read(self.h, counter, C.byref(v), 1 if reset else 0))
        return v.value

    def set_stats_buffer(self, trials, iters=None):
        """Register (or clear with None) an int32 CUDA tensor that receives per-frame LM trial counts; `iters` (same size,
        optional) receives the outer Levenberg iterations of the per-frame refine."""
        self._stats = (trials, iters)
        n = trials.numel() if trials is not None else 0
        assert iters is None or iters.numel() >= n
        _check(self.lib.gl_ctx_set_stats_buffers(self.h, _ptr(trials), _ptr(iters), n))

    def set_edge_stats_buffer(self, edges):
        """Register (or clear with None) an int32 CUDA tensor (n, 2): per frame of the on-chip refine the level-0 reprojection edges
        summed over its Levenberg trials / over its outer iterations (gl_ctx_set_edge_stats_buffer)."""
        self._edge_stats = edges
        n = edges.shape[0] if edges is not None else 0
        assert edges is None or (edges.dim() == 2 and edges.shape[1] == 2 and edges.is_contiguous())
        _check(self.lib.gl_ctx_set_edge_stats_buffer(self.h, _ptr(edges), n))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GMM:
    """gmmloc::GMM (gaussian_mixture.h:98-170) on the GPU."""

    def __init__(self, ctx, mean, cov, params=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.params = params or Params()
        mean = np.ascontiguousarray(mean, dtype=np.float64).reshape(-1, 3)
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9)
        assert mean.shape[0] == cov.shape[0]
        h = C.c_void_p()
        _check(self.lib.gl_gmm_create(ctx.h, mean.ctypes.data, cov.ctypes.data, mean.shape[0],
                                      C.byref(self.params.c()), C.byref(h)))
        self.h = h

    @classmethod
    def load(cls, ctx, path, params=None):
        """GMMUtility::loadGMMModel (gmm_utils.cpp:9-67)."""
        self = cls.__new__(cls)
        self.ctx, self.lib = ctx, ctx.lib
        self.params = params or Params()
        h = C.c_void_p()
        _check(self.lib.gl_gmm_load_file(ctx.h, str(path).encode(), C.byref(self.params.c()), C.byref(h)))
        self.h = h
        return self

    def save(self, path):
        _check(self.lib.gl_gmm_save_file(self.h, str(path).encode()))

    def countComponents(self):
        return self.lib.gl_gmm_count(self.h)

    K = property(countComponents)

    def get(self, field):
        K = self.K
        nnz = self.lib.gl_gmm_nbs_count(self.h)
        shape, dt = {
            F_MEAN: ((K, 3), np.float64), F_COV: ((K, 9), np.float64), F_COV_INV: ((K, 9), np.float64),
            F_DET: ((K,), np.float64), F_SCALE: ((K, 3), np.float64), F_AXIS: ((K, 9), np.float64),
            F_SQRT_INFO: ((K, 9), np.float64), F_FLAGS: ((K,), np.uint8), F_NBS_PTR: ((K + 1,), np.int32),
            F_NBS_IDX: ((nnz,), np.int32), F_NBS_DIST: ((nnz,), np.float64)}[field]
        out = np.zeros(shape, dtype=dt)
        _check(self.lib.gl_gmm_get(self.h, field, out.ctypes.data, out.nbytes))
        return out

    # ---- association ----------------------------------------------------------
    def associate3d(self, pts, mode=ASSOC_BRUTE, want_d2=True):
        """pts: (N,3) float64 CUDA tensor -> (idx int32 (N,), d2 float64 (N,))."""
        import torch
        N = pts.shape[0]
        idx = torch.empty(N, dtype=torch.int32, device=pts.device)
        d2 = torch.empty(N, dtype=torch.float64, device=pts.device) if want_d2 else None
        self.ctx._enter()
        try:
            _check(self.lib.gl_associate3d(self.ctx.h, self.h, _ptr(pts), N, mode, _ptr(idx), _ptr(d2)))
        finally:
            self.ctx._exit()
        return idx, d2

    def index_info(self):
        """Cell index behind ASSOC_BRUTE: dict(enabled, cell, dims, entries, always, t_resolve)."""
        import ctypes as C
        info = (C.c_double * 8)()
        _check(self.lib.gl_gmm_index_info(self.h, info))
        b = (C.c_double * 3)()
        _check(self.lib.gl_gmm_index_bytes(self.h, b))
        return {"enabled": bool(info[0]), "cell": info[1], "dims": (int(info[2]), int(info[3]), int(info[4])),
                "entries": int(info[5]), "always": int(info[6]), "t_resolve": info[7],
                "bytes": {"cell_pointers": int(b[0]), "lists": int(b[1]), "packed_cells": int(b[2])}}

    def index_work(self, pts):
        """Number of (point, component) evaluations the cell index performs for these points."""
        import torch
        out = torch.zeros(1, dtype=torch.int64, device=pts.device)
        self.ctx._enter()
        try:
            _check(self.lib.gl_assoc_index_work(self.ctx.h, self.h, _ptr(pts), pts.shape[0], _ptr(out)))
        finally:
            self.ctx._exit()
        return int(out.item())

    def knn3d(self, pts, k=5):
        import torch
        N = pts.shape[0]
        idx = torch.empty((N, k), dtype=torch.int32, device=pts.device)
        dist = torch.empty((N, k), dtype=torch.float64, device=pts.device)
        self.ctx._enter()
        try:
            _check(self.lib.gl_knn3d(self.ctx.h, self.h, _ptr(pts), N, k, _ptr(idx), _ptr(dist)))
        finally:
            self.ctx._exit()
        return idx, dist

    def queryPoint(self, pts):
        """GMM::queryPoint (gaussian_mixture.cpp:545-576), batched."""
        return self.associate3d(pts, ASSOC_KNN5_EUCLID)[0]

    def close(self):
        if getattr(self, "h", None):
            self.lib.gl_gmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def optimize_current_pose(ctx, cam, prm, pose, Xw, obs, octave, outlier=None):
    """Tracking::optimizeCurrentPose (tracking_opt.cpp:21-217) for B frames.
    pose (B,7) in/out, Xw (B,M,3), obs (B,M,3), octave (B,M) int32 (<0: no map point); outlier (B,M) uint8
    in/out (is_outlier_: rewritten where a map point exists, untouched elsewhere; default: zeros).
    Returns (outlier uint8 (B,M), ninlier int32 (B,))."""
    import torch
    B, M = octave.shape
    if outlier is None:
        outlier = torch.zeros((B, M), dtype=torch.uint8, device=pose.device)
    nin = torch.zeros(B, dtype=torch.int32, device=pose.device)
    ctx._enter()
    try:
        _check(ctx.lib.gl_optimize_current_pose(ctx.h, C.byref(cam.c()), C.byref(prm.c()), B, M, _ptr(pose), _ptr(Xw),
                                                _ptr(obs), _ptr(octave), _ptr(outlier), _ptr(nin)))
    finally:
        ctx._exit()
    return outlier, nin


def track_frames(ctx, gmm, cam, prm, pose, Xw, obs, octave, want_d2=True):
    """North-star per-frame path (gl_track_frames): exhaustive Mahalanobis association of the
    M map points of each of the B frames against the K Gaussians + structure-constrained
    refinement (jointOptimization with one free pose).  pose (B,7) and Xw (B,M,3) are updated
    in place.  Returns (assoc int32 (B,M), d2 float64 (B,M) or None)."""
    import torch
    B, M = octave.shape
    assoc = torch.empty((B, M), dtype=torch.int32, device=pose.device)
    d2 = torch.empty((B, M), dtype=torch.float64, device=pose.device) if want_d2 else None
    ctx._enter()
    try:
        _check(ctx.lib.gl_track_frames(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), B, M, _ptr(pose), _ptr(Xw),
                                       _ptr(obs), _ptr(octave), _ptr(assoc), _ptr(d2)))
    finally:
        ctx._exit()
    return assoc, d2


def track_frames_anchored(ctx, gmm, cam, prm, pose, Xw, obs, octave, prior=None, fixed_pose=None, fixed_obs=None,
                          fixed_oct=None, want_d2=True, want_erase=False):
    """gl_track_frames_anchored: the per-frame path with the reference's gauge anchors (localization_opt.cpp:491-516,
    556-581).  prior (B,) uint8 or None: EdgeSE3QuatPrior on the input pose (or the pose fixed when
    prm.ba_first_as_prior == 0); fixed_pose (B,F,7), fixed_obs (B,M,F,3), fixed_oct (B,M,F) int32 (<0: not observed):
    F fixed observer key-frames.  pose and Xw are updated in place.
    Returns (assoc int32 (B,M), d2 or None, fixed_erase uint8 (B,M,F) or None)."""
    import torch
    B, M = octave.shape
    F = 0 if fixed_pose is None else fixed_pose.shape[1]
    assoc = torch.empty((B, M), dtype=torch.int32, device=pose.device)
    d2 = torch.empty((B, M), dtype=torch.float64, device=pose.device) if want_d2 else None
    erase = torch.zeros((B, M, F), dtype=torch.uint8, device=pose.device) if (want_erase and F) else None
    an = _lib.gl_track_anchor(_ptr(prior), F, _ptr(fixed_pose), _ptr(fixed_obs), _ptr(fixed_oct), _ptr(erase))
    ctx._enter()
    try:
        _check(ctx.lib.gl_track_frames_anchored(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), B, M, _ptr(pose), _ptr(Xw),
                                                _ptr(obs), _ptr(octave), _ptr(assoc), _ptr(d2), C.byref(an)))
    finally:
        ctx._exit()
    return assoc, d2, erase


class HostFramePath:
    """The frame-at-a-time caller with HOST buffers in and out - gl_track_frame_host, what
    include/gmmloc_hip/gmm_adapter.hpp::trackFrame does for the reference host (tracking.cpp calls the path once per
    frame): the context's page-locked staging buffer and its device mirror, one enqueued transfer each way and ONE
    synchronize per frame."""

    def __init__(self, ctx, gmm, cam, prm, max_points=0):
        self.ctx, self.gmm, self.cam, self.prm = ctx, gmm, cam.c(), prm.c()

    @staticmethod
    def _arr(a, dtype, shape, name, writable=False):
        # the C entry point reads raw host memory: a float32 Xw or an int64 octave would be read out of bounds
        if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.shape == shape
                and (a.flags.writeable or not writable)):
            raise TypeError("HostFramePath: %s must be a C-contiguous %s array of shape %s%s"
                            % (name, np.dtype(dtype).name, shape, " (writable: updated in place)" if writable else ""))
        return a

    def track_frame(self, pose, Xw, obs, octave, anchored=False):
        """pose (7,), Xw (M,3) float64 are updated in place; obs (M,3) float64, octave (M,) int32; returns assoc (M,) int32.
        anchored: with the prior edge on the input pose (gl_track_frame_host_anchored)."""
        M = octave.shape[0]
        self._arr(pose, np.float64, (7,), "pose", True)
        self._arr(Xw, np.float64, (M, 3), "Xw", True)
        self._arr(obs, np.float64, (M, 3), "obs")
        self._arr(octave, np.int32, (M,), "octave")
        assoc = np.empty(M, np.int32)
        fn = self.ctx.lib.gl_track_frame_host_anchored if anchored else self.ctx.lib.gl_track_frame_host
        _check(fn(self.ctx.h, self.gmm.h, C.byref(self.cam), C.byref(self.prm), M, pose.ctypes.data,
                  Xw.ctypes.data, obs.ctypes.data, octave.ctypes.data, assoc.ctypes.data))
        return assoc

    def close(self):
        pass


def read_gmm_file(path):
    """Host-only .gmm reader (gl_gmm_file_read) -> (mean (K,3), cov (K,9))."""
    lib = _lib.load()
    K = C.c_int32()
    _check(lib.gl_gmm_file_read(str(path).encode(), None, None, 0, C.byref(K)))
    mean, cov = np.zeros((K.value, 3)), np.zeros((K.value, 9))
    _check(lib.gl_gmm_file_read(str(path).encode(), mean.ctypes.data, cov.ctypes.data, K.value, C.byref(K)))
    return mean, cov


def write_gmm_file(path, mean, cov, flags):
    lib = _lib.load()
    mean = np.ascontiguousarray(mean, np.float64)
    cov = np.ascontiguousarray(cov, np.float64)
    flags = np.ascontiguousarray(flags, np.uint8)
    _check(lib.gl_gmm_file_write(str(path).encode(), mean.ctypes.data, cov.ctypes.data, flags.ctypes.data,
                                 mean.shape[0]))


def _gmm_search2d(self, cam, pose, uv, nfeat=None, k=5, view_cap=0):
    """GMMLoc::associateMapElements = GMM::renderView + GMM::searchCorrespondence for B key-frames.
    pose (B,7), uv (B,N,2) -> (cand int32 (B,N,k), ncand int32 (B,N), view_ids (B,view_cap) | None,
    nview (B,) | None)."""
    import torch
    B, N = uv.shape[0], uv.shape[1]
    cand = torch.empty((B, N, k), dtype=torch.int32, device=uv.device)
    ncand = torch.empty((B, N), dtype=torch.int32, device=uv.device)
    vids = torch.empty((B, view_cap), dtype=torch.int32, device=uv.device) if view_cap else None
    nview = torch.empty(B, dtype=torch.int32, device=uv.device) if view_cap else None
    self.ctx._enter()
    try:
        _check(self.lib.gl_search2d(self.ctx.h, self.h, C.byref(cam.c()), B, _ptr(pose), N, _ptr(uv), _ptr(nfeat), k,
                                    _ptr(cand), _ptr(ncand), view_cap, _ptr(vids), _ptr(nview)))
    finally:
        self.ctx._exit()
    return cand, ncand, vids, nview


GMM.search2d = _gmm_search2d


def optimize_point(ctx, gmm, cam, prm, pts, uvr, octave, pose, comp, proj_z2):
    """GMMLoc::optimizePoint (gmmloc_opt.cpp:260-342), N problems -> (res, chi2_proj, chi2_str, pt_est)."""
    import torch
    N = pts.shape[0]
    dev = pts.device
    res = torch.empty(N, dtype=torch.uint8, device=dev)
    c2p = torch.empty(N, dtype=torch.float64, device=dev)
    c2s = torch.empty(N, dtype=torch.float64, device=dev)
    est = torch.empty((N, 3), dtype=torch.float64, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_optimize_point(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), N, _ptr(pts), _ptr(uvr),
                                         _ptr(octave), _ptr(pose), _ptr(comp), _ptr(proj_z2), _ptr(res), _ptr(c2p),
                                         _ptr(c2s), _ptr(est)))
    finally:
        ctx._exit()
    return res, c2p, c2s, est


def check_map_association(ctx, gmm, cam, prm, pose, pts, uvr, octave, cand, ncand):
    """GMMLoc::checkMapAssociation (gmmloc_opt.cpp:156-258): pose (B,7), pts (B,N,3) in/out, uvr (B,N,3),
    octave (B,N), cand (B,N,k), ncand (B,N) -> out_comp (B,N)."""
    import torch
    B, N, k = cand.shape
    out = torch.empty((B, N), dtype=torch.int32, device=pts.device)
    ctx._enter()
    try:
        _check(ctx.lib.gl_check_map_association(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), B, N, _ptr(pose),
                                                _ptr(pts), _ptr(uvr), _ptr(octave), _ptr(cand), _ptr(ncand), k, _ptr(out)))
    finally:
        ctx._exit()
    return out


def optimize_triangulation(ctx, gmm, cam, prm, x3d, pose1, uvr1, oct1, pose2, uvr2, oct2, cand1, n1, cand2, n2):
    """Localization::optimizeTriangulationVec (localization_opt.cpp:27-204), N problems; x3d in/out."""
    import torch
    N, k = cand1.shape
    out = torch.empty(N, dtype=torch.int32, device=x3d.device)
    ctx._enter()
    try:
        _check(ctx.lib.gl_optimize_triangulation(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), N, _ptr(x3d),
                                                 _ptr(pose1), _ptr(uvr1), _ptr(oct1), _ptr(pose2), _ptr(uvr2), _ptr(oct2),
                                                 _ptr(cand1), _ptr(n1), _ptr(cand2), _ptr(n2), k, _ptr(out)))
    finally:
        ctx._exit()
    return out


def joint_optimization(ctx, gmm, cam, prm, P, F, poses, prior, points, assoc, obs_ptr, obs_pose, obs_uvr, obs_oct, stop_flag=None):
    """Localization::jointOptimization (localization_opt.cpp:456-925) on B flat problems of one shape.
    poses (B,P+F,7) and points (B,L,3) are updated in place.  stop_flag: int32 tensor of one element (device or
    pinned host memory) = the reference's pbStopFlag (gl_joint_optimization_stoppable: > 0 stop, < 0 iteration budget).
    Returns (assoc_dropped uint8 (B,L), obs_erase uint8 (B,NOBS), iters int32 (B,))."""
    import torch
    B, L = points.shape[0], points.shape[1]
    NOBS = obs_pose.shape[1]
    dev = points.device
    dropped = torch.zeros((B, L), dtype=torch.uint8, device=dev)
    erase = torch.zeros((B, NOBS), dtype=torch.uint8, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    ctx._enter()
    try:
        if stop_flag is None:
            _check(ctx.lib.gl_joint_optimization(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), B, P, F, L, NOBS,
                                                 _ptr(poses), _ptr(prior), _ptr(points), _ptr(assoc), _ptr(obs_ptr),
                                                 _ptr(obs_pose), _ptr(obs_uvr), _ptr(obs_oct), _ptr(dropped), _ptr(erase),
                                                 _ptr(iters)))
        else:
            _check(ctx.lib.gl_joint_optimization_stoppable(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), B, P, F, L, NOBS,
                                                           _ptr(poses), _ptr(prior), _ptr(points), _ptr(assoc), _ptr(obs_ptr),
                                                           _ptr(obs_pose), _ptr(obs_uvr), _ptr(obs_oct), _ptr(dropped), _ptr(erase),
                                                           _ptr(iters), _ptr(stop_flag)))
    finally:
        ctx._exit()
    return dropped, erase, iters


def search_by_projection(ctx, cam, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, mp_uvr, mp_level, mp_viewcos,
                         mp_valid, mp_desc, th=3.0, nn_ratio=0.8, scale_factor=1.2):
    """ORBmatcher::searchByProjection (orb_matcher.cpp:27-110) for B frames.  feat_* (B,NF,...), mp_* (B,NP,...)
    device tensors -> (feat_match int32 (B,NF), nmatches int32 (B,))."""
    import torch
    B, NF = feat_oct.shape
    NP = mp_level.shape[1]
    match = torch.empty((B, NF), dtype=torch.int32, device=feat_oct.device)
    nm = torch.empty(B, dtype=torch.int32, device=feat_oct.device)
    ctx._enter()
    try:
        _check(ctx.lib.gl_search_by_projection(ctx.h, C.byref(cam.c()), float(scale_factor), B, NF, NP, _ptr(feat_uv),
                                               _ptr(feat_ur), _ptr(feat_oct), _ptr(feat_desc), _ptr(feat_taken), _ptr(mp_uvr),
                                               _ptr(mp_level), _ptr(mp_viewcos), _ptr(mp_valid), _ptr(mp_desc), float(th),
                                               float(nn_ratio), _ptr(match), _ptr(nm)))
    finally:
        ctx._exit()
    return match, nm


def search_local_points(ctx, cam, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, pose_cw, t_wc, mp_pos, mp_normal, mp_max_dist,
                        mp_min_dist, mp_cand, mp_desc, th=3.0, nn_ratio=0.8, scale_factor=1.2):
    """Tracking::searchLocalPoints (tracking.cpp:213-270), the device part in one call: project_map_points, then
    search_by_projection on its outputs -> (feat_match int32 (B,NF), nmatches int32 (B,), inview u8 (B,NP))."""
    import torch
    B, NF = feat_oct.shape
    NP = mp_cand.shape[1]
    dev = feat_oct.device
    match = torch.empty((B, NF), dtype=torch.int32, device=dev)
    nm = torch.empty(B, dtype=torch.int32, device=dev)
    inview = torch.empty((B, NP), dtype=torch.uint8, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_search_local_points(ctx.h, C.byref(cam.c()), float(scale_factor), B, NF, NP, _ptr(feat_uv), _ptr(feat_ur), _ptr(feat_oct),
                                              _ptr(feat_desc), _ptr(feat_taken), _ptr(pose_cw), _ptr(t_wc), _ptr(mp_pos), _ptr(mp_normal),
                                              _ptr(mp_max_dist), _ptr(mp_min_dist), _ptr(mp_cand), _ptr(mp_desc), float(th), float(nn_ratio),
                                              _ptr(match), _ptr(nm), _ptr(inview)))
    finally:
        ctx._exit()
    return match, nm, inview


CHAIN_FIELDS = ("feat_uv", "feat_ur", "feat_oct", "feat_angle", "feat_desc", "feat_taken", "pose_lw", "last_pt", "last_valid", "last_oct",
                "last_angle", "last_desc", "last_to_local", "mp_pos", "mp_normal", "mp_max_dist", "mp_min_dist", "mp_cand", "mp_desc",
                "pose_cw", "pose_mm", "match_last", "match_local", "outlier", "counts", "inview")
CHAIN_DTYPES = {"feat_uv": "float64", "feat_ur": "float32", "feat_oct": "int32", "feat_angle": "float32", "feat_desc": "uint8",
                "feat_taken": "uint8", "pose_lw": "float64", "last_pt": "float64", "last_valid": "uint8", "last_oct": "int32",
                "last_angle": "float32", "last_desc": "uint8", "last_to_local": "int32", "mp_pos": "float64", "mp_normal": "float64",
                "mp_max_dist": "float32", "mp_min_dist": "float32", "mp_cand": "uint8", "mp_desc": "uint8", "pose_cw": "float64"}


# round 6: optional inputs (a key absent from `a` = NULL).  last_observed: countObservations() > 0 of the last frame's map points; the
# kf_* / feat_node_* buffers switch the trackKeyFrame fallback on.
CHAIN_OPT_DTYPES = {"last_observed": "uint8", "kf_angle": "float32", "kf_desc": "uint8", "kf_has_mp": "uint8", "kf_nnode": "int32",
                    "kf_node_id": "int32", "kf_node_ptr": "int32", "kf_node_idx": "int32", "kf_pt": "float64", "kf_to_local": "int32",
                    "feat_nnode": "int32", "feat_node_id": "int32", "feat_node_ptr": "int32", "feat_node_idx": "int32"}
CHAIN_FIELDS2A = ("last_observed", "drop_src", "counts2")
CHAIN_FIELDS2B = ("kf_angle", "kf_desc", "kf_has_mp", "kf_nnode", "kf_node_id", "kf_node_ptr", "kf_node_idx", "kf_pt", "kf_to_local",
                  "feat_nnode", "feat_node_id", "feat_node_ptr", "feat_node_idx", "match_kf", "drop_kf")


class _ChainIO(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in CHAIN_FIELDS] + [(k, C.c_void_p) for k in CHAIN_FIELDS2A] +
                [("NK", C.c_int32), ("NNK", C.c_int32), ("NNF", C.c_int32), ("reserved_", C.c_int32)] + [(k, C.c_void_p) for k in CHAIN_FIELDS2B])


def _chain_io(a, out):
    import torch
    for k, dt in CHAIN_DTYPES.items():
        t = a[k]
        assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch." + dt, (k, t.dtype, t.is_contiguous())
    for k, dt in CHAIN_OPT_DTYPES.items():
        if k in a and a[k] is not None:
            t = a[k]
            assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch." + dt, (k, t.dtype, t.is_contiguous())
    io = _ChainIO()
    for k in CHAIN_FIELDS + CHAIN_FIELDS2A + CHAIN_FIELDS2B:
        t = out["pose"] if k == "pose_cw" else (out[k] if k in out else a.get(k))
        setattr(io, k, _ptr(t) if t is not None else None)
    if a.get("kf_desc") is not None:
        io.NK, io.NNK, io.NNF = a["kf_desc"].shape[1], a["kf_node_id"].shape[1], a["feat_node_id"].shape[1]
    return io


def _chain_out(a, front_only=False):
    import torch
    B, NF = a["feat_oct"].shape
    NP = a["mp_cand"].shape[1]
    dev = a["feat_oct"].device
    i32 = lambda *sh: torch.full(sh, -1, dtype=torch.int32, device=dev)
    out = dict(pose=a["pose_cw"].clone(), pose_mm=torch.empty((B, 7), dtype=torch.float64, device=dev), match_last=i32(B, NF),
               match_local=i32(B, NF), outlier=torch.zeros((B, NF), dtype=torch.uint8, device=dev),
               counts=torch.zeros((B, 4), dtype=torch.int32, device=dev), inview=torch.zeros((B, NP), dtype=torch.uint8, device=dev),
               drop_src=i32(B, NF), counts2=torch.zeros((B, 4), dtype=torch.int32, device=dev))
    if a.get("kf_desc") is not None:
        out["match_kf"] = i32(B, NF)
        out["drop_kf"] = i32(B, NF)
    return out


def track_frame_chain(ctx, cam, prm, a, th_mm=7.0, th_local=3.0, nn_ratio=0.8, mono=False, scale_factor=1.2, out=None):
    """gl_track_frame_chain: trackWithMotionModel (-> trackKeyFrame where it fails, if `a` holds the key-frame buffers) ->
    searchLocalPoints -> trackLocalMap for B frames, device resident.  `a`: dict of CUDA tensors with the input keys of CHAIN_DTYPES
    (+ optionally those of CHAIN_OPT_DTYPES); pose_cw (B,7): the motion-model prediction, NOT modified - the result comes back as a
    new tensor.  Returns dict(pose, pose_mm, match_last, match_local, outlier, counts (B,4), inview, drop_src, counts2 (B,4)
    [, match_kf, drop_kf]).  `out`: the dict of an earlier call, to be used again (a host that tracks frame after frame keeps its buffers:
    0.05 ms of allocations less per call); its pose is reset to a["pose_cw"] first."""
    B, NF = a["feat_oct"].shape
    NL, NP = a["last_oct"].shape[1], a["mp_cand"].shape[1]
    if out is None:
        out = _chain_out(a)
    else:
        out["pose"].copy_(a["pose_cw"])
    io = _chain_io(a, out)
    ctx._enter()
    try:
        _check(ctx.lib.gl_track_frame_chain(ctx.h, C.byref(cam.c()), C.byref(prm.c()), float(scale_factor), B, NF, NL, NP, C.byref(io),
                                            float(th_mm), float(th_local), float(nn_ratio), int(bool(mono))))
    finally:
        ctx._exit()
    return out


def track_frame_chain_front(ctx, cam, prm, a, th_mm=7.0, mono=False, scale_factor=1.2):
    """gl_track_frame_chain_front: stages 1, 2 (and the trackKeyFrame fallback) of the chain.  Returns the same dict as
    track_frame_chain (pose = the pose after stage 2 / 2b; match_local / inview / counts[2:] untouched); hand it to
    track_frame_chain_back together with the local map Tracking::updateLocalMap made from it."""
    B, NF = a["feat_oct"].shape
    NL, NP = a["last_oct"].shape[1], a["mp_cand"].shape[1]
    out = _chain_out(a)
    io = _chain_io(a, out)
    ctx._enter()
    try:
        _check(ctx.lib.gl_track_frame_chain_front(ctx.h, C.byref(cam.c()), C.byref(prm.c()), float(scale_factor), B, NF, NL, NP, C.byref(io),
                                                  float(th_mm), int(bool(mono))))
    finally:
        ctx._exit()
    return out


def track_frame_chain_back(ctx, cam, prm, a, out, th_local=3.0, nn_ratio=0.8, scale_factor=1.2):
    """gl_track_frame_chain_back: stages 3, 4 on the associations `out` (of track_frame_chain_front) and the local map in `a`
    (mp_*, last_to_local, kf_to_local: indices into THIS local map).  `out` is updated in place and returned."""
    B, NF = a["feat_oct"].shape
    NL, NP = a["last_oct"].shape[1], a["mp_cand"].shape[1]
    import torch
    if out["inview"].shape[1] != NP:
        out["inview"] = torch.zeros((B, NP), dtype=torch.uint8, device=a["feat_oct"].device)
    io = _chain_io(a, out)
    ctx._enter()
    try:
        _check(ctx.lib.gl_track_frame_chain_back(ctx.h, C.byref(cam.c()), C.byref(prm.c()), float(scale_factor), B, NF, NL, NP, C.byref(io),
                                                 float(th_local), float(nn_ratio)))
    finally:
        ctx._exit()
    return out


def search_by_projection_frame(ctx, cam, pose_cw, pose_lw, feat_uv, feat_ur, feat_oct, feat_angle, feat_desc, feat_taken,
                               last_pt, last_valid, last_oct, last_angle, last_desc, th=7.0, mono=False,
                               check_orientation=True, scale_factor=1.2):
    """ORBmatcher::searchByProjection(CurrentFrame, LastFrame, th, bMono) (orb_matcher.cpp:410-542) for B frame
    pairs -> (feat_match int32 (B,NF): index of the last-frame feature, nmatches int32 (B,))."""
    import torch
    B, NF = feat_oct.shape
    NL = last_oct.shape[1]
    match = torch.empty((B, NF), dtype=torch.int32, device=feat_oct.device)
    nm = torch.empty(B, dtype=torch.int32, device=feat_oct.device)
    ctx._enter()
    _check(ctx.lib.gl_search_by_projection_frame(ctx.h, C.byref(cam.c()), float(scale_factor), B, NF, NL, _ptr(pose_cw),
                                                 _ptr(pose_lw), _ptr(feat_uv), _ptr(feat_ur), _ptr(feat_oct),
                                                 _ptr(feat_angle), _ptr(feat_desc), _ptr(feat_taken), _ptr(last_pt),
                                                 _ptr(last_valid), _ptr(last_oct), _ptr(last_angle), _ptr(last_desc),
                                                 float(th), int(bool(mono)), int(bool(check_orientation)), _ptr(match),
                                                 _ptr(nm)))
    ctx._exit()
    return match, nm


KF_KEYS = ("uv", "ur", "oct", "angle", "desc", "has_mp", "nnode", "node_id", "node_ptr", "node_idx")
KF_DTYPES = {"uv": "float64", "ur": "float32", "oct": "int32", "angle": "float32", "desc": "uint8", "has_mp": "uint8", "nnode": "int32",
             "node_id": "int32", "node_ptr": "int32", "node_idx": "int32"}


def _check_kf(kf, name, keys=KF_KEYS):
    """the key-frame tensors go to the library as raw pointers: a wrong dtype or a strided view would be read as garbage"""
    for k in keys:
        t, dt = kf[k], KF_DTYPES[k]
        assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch." + dt, "%s[%r]: expected a contiguous CUDA %s tensor, got %s" % (name, k, dt, t.dtype)


def search_for_triangulation(ctx, kf1, kf2, fmat, epipole, only_stereo=False, check_orientation=True, scale_factor=1.2):
    """ORBmatcher::searchForTriangulation (orb_matcher.cpp:141-293) for B key-frame pairs.  kf1 / kf2: dicts of CUDA tensors
    with the keys KF_KEYS - uv (B,N,2) f64, ur (B,N) f32, oct (B,N) i32, angle (B,N) f32, desc (B,N,32) u8, has_mp (B,N) u8 and
    the DBoW2 feature vector as CSR: nnode (B,) i32, node_id (B,NN) i32 ascending, node_ptr (B,NN+1) i32, node_idx (B,N) i32;
    fmat (B,9) f64, epipole (B,2) f32 -> (match12 int32 (B,N1): feature of key-frame 2 or -1, nmatches int32 (B,))."""
    import torch
    B, N1 = kf1["oct"].shape
    N2 = kf2["oct"].shape[1]
    NN1, NN2 = kf1["node_id"].shape[1], kf2["node_id"].shape[1]
    assert kf1["node_ptr"].shape[1] == NN1 + 1 and kf2["node_ptr"].shape[1] == NN2 + 1
    _check_kf(kf1, "kf1")
    _check_kf(kf2, "kf2")
    assert str(fmat.dtype) == "torch.float64" and str(epipole.dtype) == "torch.float32"
    dev = kf1["oct"].device
    match = torch.empty((B, N1), dtype=torch.int32, device=dev)
    nm = torch.empty(B, dtype=torch.int32, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_search_for_triangulation(ctx.h, float(scale_factor), B, N1, N2, NN1, NN2, *[_ptr(kf1[k]) for k in KF_KEYS],
                                                   *[_ptr(kf2[k]) for k in KF_KEYS], _ptr(fmat), _ptr(epipole), int(bool(only_stereo)),
                                                   int(bool(check_orientation)), _ptr(match), _ptr(nm)))
    finally:
        ctx._exit()
    return match, nm


def level_steps(scale_factor=1.2):
    """The seven steps of the predicted level (mappoint.cpp:289-293) with the host's logf: step[L] = largest ratio with level <= L."""
    out = np.zeros(7, np.float32)
    _check(_lib.load().gl_level_steps(float(scale_factor), out.ctypes.data_as(C.c_void_p)))
    return out


def project_map_points(ctx, cam, pose_cw, t_wc, pos, normal, max_dist, min_dist, cand, scale_factor=1.2):
    """Frame::project3 + MapPoint::checkScaleAndVisible for B frames x NP map points (tracking.cpp:233-256): CUDA tensors pose_cw (B,7),
    t_wc (B,3), pos / normal (B,NP,3) f64, max_dist / min_dist (B,NP) f32, cand (B,NP) u8 -> (uvr (B,NP,3), level int32 (B,NP),
    viewcos (B,NP), dist (B,NP), inview u8 (B,NP)): the mp_* inputs of search_by_projection / fuse_search."""
    import torch
    B, NP = cand.shape
    dev = cand.device
    uvr = torch.empty((B, NP, 3), dtype=torch.float64, device=dev)
    level = torch.empty((B, NP), dtype=torch.int32, device=dev)
    viewcos = torch.empty((B, NP), dtype=torch.float64, device=dev)
    dist = torch.empty((B, NP), dtype=torch.float64, device=dev)
    inview = torch.empty((B, NP), dtype=torch.uint8, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_project_map_points(ctx.h, C.byref(cam.c()), float(scale_factor), B, NP, _ptr(pose_cw), _ptr(t_wc), _ptr(pos), _ptr(normal),
                                             _ptr(max_dist), _ptr(min_dist), _ptr(cand), _ptr(uvr), _ptr(level), _ptr(viewcos), _ptr(dist),
                                             _ptr(inview)))
    finally:
        ctx._exit()
    return uvr, level, viewcos, dist, inview


def fuse_search(ctx, cam, feat_uv, feat_ur, feat_oct, feat_desc, mp_uvr, mp_level, mp_valid, mp_desc, th=3.0, scale_factor=1.2):
    """Localization::fuseObservations (localization.cpp:226-318), the matching half, for B key-frames: CUDA tensors feat_uv (B,NF,2)
    f64, feat_ur (B,NF) f32, feat_oct (B,NF) i32, feat_desc (B,NF,32) u8; mp_uvr (B,NP,3) f64, mp_level (B,NP) i32, mp_valid (B,NP)
    u8, mp_desc (B,NP,32) u8 -> (best_idx int32 (B,NP) or -1, best_dist int32 (B,NP))."""
    import torch
    B, NF = feat_oct.shape
    NP = mp_level.shape[1]
    bi = torch.empty((B, NP), dtype=torch.int32, device=feat_oct.device)
    bd = torch.empty((B, NP), dtype=torch.int32, device=feat_oct.device)
    ctx._enter()
    try:
        _check(ctx.lib.gl_fuse_search(ctx.h, C.byref(cam.c()), float(scale_factor), B, NF, NP, _ptr(feat_uv), _ptr(feat_ur), _ptr(feat_oct),
                                      _ptr(feat_desc), _ptr(mp_uvr), _ptr(mp_level), _ptr(mp_valid), _ptr(mp_desc), float(th), _ptr(bi), _ptr(bd)))
    finally:
        ctx._exit()
    return bi, bd


def search_by_bow(ctx, kf, fr, nn_ratio=0.7, check_orientation=True):
    """ORBmatcher::searchByBoW (orb_matcher.cpp:295-408) for B key-frame / frame pairs.  kf: dict of CUDA tensors angle (B,N1) f32,
    desc (B,N1,32) u8, has_mp (B,N1) u8 (valid map point) and the feature vector as CSR (nnode, node_id, node_ptr, node_idx, as in
    search_for_triangulation); fr: angle, desc and the feature vector of the frame -> (match21 int32 (B,N2): the key-frame feature
    whose map point the frame feature gets, or -1; nmatches int32 (B,))."""
    import torch
    B, N1 = kf["angle"].shape
    N2 = fr["angle"].shape[1]
    NN1, NN2 = kf["node_id"].shape[1], fr["node_id"].shape[1]
    assert kf["node_ptr"].shape[1] == NN1 + 1 and fr["node_ptr"].shape[1] == NN2 + 1
    _check_kf(kf, "kf", ("angle", "desc", "has_mp", "nnode", "node_id", "node_ptr", "node_idx"))
    _check_kf(fr, "fr", ("angle", "desc", "nnode", "node_id", "node_ptr", "node_idx"))
    dev = kf["angle"].device
    match = torch.empty((B, N2), dtype=torch.int32, device=dev)
    nm = torch.empty(B, dtype=torch.int32, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_search_by_bow(ctx.h, float(nn_ratio), int(bool(check_orientation)), B, N1, N2, NN1, NN2,
                                        *[_ptr(kf[k]) for k in ("angle", "desc", "has_mp", "nnode", "node_id", "node_ptr", "node_idx")],
                                        *[_ptr(fr[k]) for k in ("angle", "desc", "nnode", "node_id", "node_ptr", "node_idx")],
                                        _ptr(match), _ptr(nm)))
    finally:
        ctx._exit()
    return match, nm


def gather_triangulation_matches(ctx, match12, nmatches, side1, side2, cap=None):
    """gl_gather_triangulation_matches: the matches of search_for_triangulation as the per-match arrays of create_map_points.
    side1 / side2: dicts of CUDA tensors pose (B,7), uv (B,N,2), ur (B,N) f32, depth (B,N) f32, oct (B,N) i32, cand (B,N,k) i32,
    ncand (B,N) i32.  Returns (pair_off (B+1,), dict of the per-match tensors with `cap` rows)."""
    import torch
    B, N1 = match12.shape
    N2 = side2["oct"].shape[1]
    k = side1["cand"].shape[2]
    cap = cap if cap is not None else B * min(N1, N2)
    dev = match12.device
    off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    m = dict(pose1=torch.zeros((cap, 7), dtype=torch.float64, device=dev), uvr1=torch.zeros((cap, 3), dtype=torch.float64, device=dev),
             depth1=torch.zeros(cap, dtype=torch.float32, device=dev), oct1=torch.zeros(cap, dtype=torch.int32, device=dev),
             cand1=torch.zeros((cap, k), dtype=torch.int32, device=dev), n1=torch.zeros(cap, dtype=torch.int32, device=dev),
             pose2=torch.zeros((cap, 7), dtype=torch.float64, device=dev), uvr2=torch.zeros((cap, 3), dtype=torch.float64, device=dev),
             depth2=torch.zeros(cap, dtype=torch.float32, device=dev), oct2=torch.zeros(cap, dtype=torch.int32, device=dev),
             cand2=torch.zeros((cap, k), dtype=torch.int32, device=dev), n2=torch.zeros(cap, dtype=torch.int32, device=dev),
             pair=torch.full((cap,), -1, dtype=torch.int32, device=dev), idx1=torch.full((cap,), -1, dtype=torch.int32, device=dev),
             idx2=torch.full((cap,), -1, dtype=torch.int32, device=dev))
    keys = ("pose", "uv", "ur", "depth", "oct", "cand", "ncand")
    ctx._enter()
    try:
        _check(ctx.lib.gl_gather_triangulation_matches(
            ctx.h, B, N1, N2, k, cap, _ptr(match12), _ptr(nmatches), *[_ptr(side1[q]) for q in keys], *[_ptr(side2[q]) for q in keys], _ptr(off),
            *[_ptr(m[q]) for q in ("pose1", "uvr1", "depth1", "oct1", "cand1", "n1", "pose2", "uvr2", "depth2", "oct2", "cand2", "n2", "pair", "idx1", "idx2")]))
    finally:
        ctx._exit()
    return off, m


def create_map_points(ctx, gmm, cam, prm, pose1, uvr1, depth1, oct1, pose2, uvr2, depth2, oct2, cand1, n1, cand2, n2,
                      scale_factor=1.2):
    """Localization::createMapPoints per-match block (localization_opt.cpp:286-420), N matches ->
    (x3d float64 (N,3), type int32 (N,), comp int32 (N,))."""
    import torch
    N, k = cand1.shape
    dev = pose1.device
    x3d = torch.empty((N, 3), dtype=torch.float64, device=dev)
    typ = torch.empty(N, dtype=torch.int32, device=dev)
    comp = torch.empty(N, dtype=torch.int32, device=dev)
    ctx._enter()
    try:
        _check(ctx.lib.gl_create_map_points(ctx.h, gmm.h, C.byref(cam.c()), C.byref(prm.c()), float(scale_factor), N,
                                            _ptr(pose1), _ptr(uvr1), _ptr(depth1), _ptr(oct1), _ptr(pose2), _ptr(uvr2),
                                            _ptr(depth2), _ptr(oct2), _ptr(cand1), _ptr(n1), _ptr(cand2), _ptr(n2), k,
                                            _ptr(x3d), _ptr(typ), _ptr(comp)))
    finally:
        ctx._exit()
    return x3d, typ, comp
