/*
 * gmmloc_hip.h -- C-ABI of libgmmloc_hip.so: the MI355X (gfx950) drop-in for the
 * GMM association + structure-constrained refinement hot path of
 * HyHuang1995/gmmloc.  extern "C", plain pointers and sizes, no C++ types, no
 * exceptions across the boundary.
 *
 * The reference has NO plugin / FFI interface for this path: it is ordinary C++
 * methods on heap objects.  Each entry point below names the
 * reference function it replaces (paths relative to the reference root); the
 * adapter a maintainer would add to the reference host is in INTEGRATION.md and
 * include/gmmloc_hip/gmm_adapter.hpp.
 *
 * Conventions
 *   - every call returns int: 0 = GL_OK, <0 = error (gl_last_error_string()).
 *     "not converged" / "no association" are results, not errors.
 *   - components are identified by int32 index = order in the .gmm file
 *     (GMM::getComponent3d(idx), gaussian_mixture.h:147-149); -1 = none.
 *   - poses are 7 doubles  (qx qy qz qw tx ty tz)  = g2o::SE3Quat T_cw
 *     (world -> camera), Eigen coefficient order.
 *   - pointers suffixed _dev are DEVICE pointers valid on the context's device;
 *     those calls are asynchronous on the context's HIP stream.  Pointers
 *     without the suffix are host pointers and the call is synchronous.
 *   - all arithmetic is IEEE fp64 (the reference's scalar_t = double,
 *     common/eigen_types.h:6); float config scalars stay float (config.h:38-89).
 *   - a gl_gmm_t is immutable after creation and may be shared by contexts /
 *     host threads; a gl_ctx_t (stream + scratch) belongs to one host thread.
 */
#ifndef GMMLOC_HIP_H_
#define GMMLOC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gl_gmm gl_gmm_t;
typedef struct gl_ctx gl_ctx_t;

enum gl_status {
  GL_OK = 0,
  GL_ERR_ARG = -1,     /* bad argument (reference: CHECK / CHECK_NOTNULL aborts) */
  GL_ERR_IO = -2,      /* file cannot be opened (gmm_utils.cpp:19-22 -> false)   */
  GL_ERR_FORMAT = -3,  /* malformed .gmm stream (gmm_utils.cpp:27-35,44-51)      */
  GL_ERR_DEVICE = -4,  /* HIP runtime error                                      */
  GL_ERR_NOMEM = -5
};

/* PinholeCamera (cv/pinhole_camera.h) + camera::bf (config.h:38-52). */
typedef struct gl_camera {
  double fx, fy, cx, cy, bf;
  int32_t width, height;
} gl_camera;

/* Hot-path configuration values (config.h:31-89, cfg/v1.yaml). */
typedef struct gl_params {
  double neighbor_dist_thresh; /* gmmmap::neighbor_dist_thresh (2.5)          */
  float tri_lambda2;           /* loc::tri_lambda2   (400)                    */
  float tri_str_thresh;        /* loc::tri_str_thresh (0.0064)                */
  float ba_lambda2;            /* loc::ba_lambda2    (400)                    */
  int32_t tri_check_str_chi2;  /* loc::tri_check_str_chi2 (true)              */
  int32_t ba_first_as_prior;   /* loc::ba_first_as_prior  (true)              */
  float sigma2_inv[8];         /* frame::sigma2_inv, init_config.hpp:60-79    */
} gl_params;

/* Fills the values of gmmloc_ros/cfg/v1.yaml and the float 1.2^(-2l) table. */
void gl_default_params(gl_params* p);

const char* gl_last_error_string(void); /* thread-local */
int gl_device_count(void);

/* ---- context: device + stream + scratch --------------------------------- */
/* hip_stream: the hipStream_t to launch on (e.g. torch's current stream); NULL = the
 * device's default (null) stream. */
int gl_ctx_create(int device, void* hip_stream, gl_ctx_t** out);
int gl_ctx_destroy(gl_ctx_t* ctx);
int gl_ctx_synchronize(gl_ctx_t* ctx);
void* gl_ctx_stream(gl_ctx_t* ctx);
/* Tuning / test options of a context (launch shapes, A/B switches; never the results' meaning).  Each option is
 * initialised ONCE at gl_ctx_create from the environment variable GMMLOC_<NAME IN CAPITALS> and changed only by
 * this call afterwards - no entry point reads the environment.  Names:
 *   ba_shape (-1 auto | 0 one workgroup per frame | 1 one point per thread; same bits either way),
 *   ba_persist (1; 0: the batch-shaped refine of gl_track_frames as one block per frame instead of persistent workgroups drawing frames
 *     from a queue - same bits; A/B),
 *   ba_two_frames (0; 1: the 2 000-point class of the plain batch refine as two frames per CU - two groups of the summation order per
 *     wave, the points' hand-over slots in global memory; same bits, measured 21 % slower: an experiment kept with its test),
 *   ba_step32 (1: fp32-cached point step in gl_track_frames, faster, NOT bit-compatible with the default),
 *   assoc_grid (0: every association is the plain N x K sweep, never the cell index),
 *   assoc_coop (1; 0: the indexed association gathers a record per lane instead of per six lanes - A/B),
 *   assoc_rec_pad (1; 0: the cooperative gather reads the 96-byte records instead of their one-per-128-byte-line copy - A/B),
 *   assoc_coop_long (1; 0: a point whose cell lists more than three candidates walks that list alone after the cooperative gather - A/B),
 *   assoc_coop_bal (1; 0: the pairs of the cooperative gather are evaluated by the lane that owns the point instead of one per lane and round - A/B),
 *   ba_rendezvous_us (200): time limit of every exchange between the workgroups of a frame on the latency shape of
 *     gl_track_frames (one point per thread, up to 8 workgroups per frame, launched plainly).  A frame whose workgroups do
 *     not find each other in time - another launch holds the CUs - or lose each other later gives up; its results only
 *     ever reach the caller's buffers through the one-workgroup kernel that always follows, which copies the staged result
 *     of a complete frame and recomputes the others from the untouched inputs: same bits, <= ~0.5 ms more.
 *     GL_COUNTER_BA_REDONE counts those frames,
 *   ba_same_xcd (0): 1 lets the exchange of that shape use its same-XCD form - workgroup-scope atomic stores that stay in
 *     the XCD's L2, polled by the siblings with agent-scope (L1-bypassing) loads: 0.34 instead of 0.37 ms per frame of 2 000
 *     points.  OPT-IN, because it rests on a HARDWARE ASSUMPTION outside the HIP memory model: a workgroup-scope store becomes
 *     visible to an agent-scope load of ANOTHER workgroup on the same XCD because the vector L1 of gfx942 / gfx950 is
 *     write-through and the XCD's workgroups share one L2.  Even when enabled it is used only if (a) a probe at gl_ctx_create
 *     found block b on XCC id b % 8 and (b) the frame's workgroups reported one and the same id in the launch's first
 *     (device-scope) exchange; a word that did not become visible would time the exchange out (ba_rendezvous_us) and send the
 *     frame to the follow-up kernel.  The default (0) uses device-scope stores only: the model-conforming path; same bits,
 *   bagen_mode (0): launch shape of gl_joint_optimization.  A window's RESULT BITS and the call's BLOCKING BEHAVIOUR are a
 *     function of the window shape (P, F, L, NOBS) and this option alone - never of B:
 *       0  by window: the pipelined shape (a kernel per phase, cycles enqueued ahead; 1.4 - 2 x less per Levenberg trial) for
 *          windows of >= 3 000 observations (NOBS), the persistent cooperative kernel below that;
 *       1  the persistent kernel for every window: ASYNCHRONOUS on the context's stream (stream-capturable; the caller
 *          synchronises).  Its workgroup count per window follows NOBS; a batch too large to keep B x that many workgroups
 *          co-resident is launched in sub-batches, in stream order;
 *       2  the pipelined shape for every window that fits it (P <= 22, P + F <= 32): BLOCKING - the call returns with the work
 *          complete (it reads the count of unfinished windows back between chunks of cycles; not stream-capturable);
 *       3  the persistent kernel with the whole batch in ONE launch and as many workgroups per window as stay co-resident:
 *          the fastest form of large batches (round 3's default); asynchronous; the ONE mode in which a window's bits depend
 *          on the batch size (the workgroup count decides the order of its partial sums).
 *     Same arithmetic everywhere, every mode held to the oracle; the shapes add their partial sums in different orders, so a
 *     window may take a different number of trials in each,
 *   assoc_pack_mb (512; memory budget in MB of the packed cell table of a GMM created with this context, 0 = none),
 *   assoc_cell8 (1; that table in 8 bytes per cell where K < 2^20, 0 = 16 bytes per cell; same results),
 *   pose_compact (-1; gl_optimize_current_pose moves the edges of a problem of more than 1 024 slots - one slot per feature, 1 200 in the
 *     reference - to the front of a problem of 1 024 where they fit: one frame of 420 edges 0.31 -> 0.21 ms; 1: every problem of more than
 *     256 slots; 0: never.  Same decisions, poses within 1e-9 of the uncompacted problem's),
 *   assoc_cell, assoc_globcells (> 0: cell size in metres / cell-count threshold of the index instead of the automatic ones),
 *   ba_fixed_pack (1: fixed observers of gl_track_frames_anchored always through the general kernel),
 *   pipe_lanes, pipe_judge, schur_kper (-1 automatic; A/B switches of the pipelined local BA in batches: streams a call is split over,
 *     the verdict on a trial as a kernel of its own, chunks per wave of the Schur pass - none changes a bit),
 *   ba_slow, ba_test_abort_seq, pose_waves, pose_regs, bagen_nb, view_slot_lds, view_threads, assoc_index_min, match_desc_lds, fuse_records, pose_compact_cap. */
int gl_ctx_set_option(gl_ctx_t* ctx, const char* name, double value);
int gl_ctx_get_option(gl_ctx_t* ctx, const char* name, double* value);
/* Kernel timing with HIP events on the context's stream: while enabled, every
 * launch of the named hot kernel class is bracketed by events.  Returns the
 * accumulated milliseconds / launch count since the last reset. */
enum gl_timer {
  GL_TIMER_ASSOC = 0,       /* association kernels                                             */
  GL_TIMER_REFINE_POSE = 1, /* k_optimize_current_pose                                         */
  GL_TIMER_BA = 2,          /* the refine kernel proper (k_ba1_fast / k_ba1 / k_ba_gen)        */
  GL_TIMER_BA_PREP = 3,     /* k_ba1_prep: set-up launch of gl_track_frames' refine            */
  GL_TIMER_COUNT = 8
};
int gl_ctx_timing_enable(gl_ctx_t* ctx, int on);
int gl_ctx_timing_read(gl_ctx_t* ctx, int timer, double* total_ms, int64_t* launches, int reset);
/* Event counters of a context (device-side, read with one small synchronous copy on the context's stream). */
enum gl_counter {
  GL_COUNTER_BA_REDONE = 0, /* frames of latency-shape launches of gl_track_frames that gave up and were recomputed by the follow-up kernel */
  GL_COUNTER_MATCH_ROUNDS = 1, /* rounds of the owner fixed point, summed over the frames / pairs of gl_search_by_projection{,_frame},
                                  gl_search_for_triangulation, gl_search_by_bow (what the reference's order-dependent loop costs here) */
  GL_COUNTER_MATCH_UNITS = 2,  /* ... and the number of those frames / pairs */
  GL_COUNTER_BA_COOP_FALLBACK = 3, /* windows of gl_joint_optimization (persistent kernel) that ran with ONE workgroup because even a single
                                      window's cooperative launch was refused (another context holds the CUs): same arithmetic, but the
                                      window's partial sums are added in another order - the one exception to "a window's bits depend on
                                      its shape and bagen_mode only" (a refused sub-batch is first halved at the same workgroup count) */
  GL_COUNTER_COUNT = 4
};
int gl_ctx_counter_read(gl_ctx_t* ctx, int counter, int64_t* value, int reset);
/* Optional statistics: while a device buffer of n int32 is registered, gl_track_frames (and gl_track_frames_anchored,
 * gl_joint_optimization) write the number of Levenberg trials (linearise + solve + evaluate) each frame / problem
 * b < n spent, so that the algorithmic work of a launch can be reported.  NULL / 0 unregisters. */
int gl_ctx_set_stats_buffer(gl_ctx_t* ctx, int32_t* trials_dev, int n);
/* ... and, for the per-frame refine (gl_track_frames / gl_track_frames_anchored), the number of OUTER Levenberg iterations
 * (g2o's optimize() iterations: one linearisation each; a trial beyond the first of an iteration re-solves the same
 * linearisation with a larger lambda) into iters_dev (n int32, may be NULL): the two counts bracket the algorithmic work. */
int gl_ctx_set_stats_buffers(gl_ctx_t* ctx, int32_t* trials_dev, int32_t* iters_dev, int n);
/* ... and the ACTIVE part of that work (gl_track_frames / gl_track_frames_anchored on the on-chip refine, M <= 2000): per frame
 * b < n two int32 {sum over its Levenberg trials, sum over its outer iterations} of the number of level-0 reprojection edges the
 * trial / iteration ran on.  The reference puts gated-out edges at level 1 (localization_opt.cpp:799-825): they are in no
 * linearisation after that, so a flop model prices these sums, not points x trials.  edges_dev: n x 2 int32; NULL / 0 unregisters. */
int gl_ctx_set_edge_stats_buffer(gl_ctx_t* ctx, int32_t* edges_dev, int n);

/* ---- GMM map: replaces GMMUtility::loadGMMModel (gmm_utils.cpp:9-67),
 *      GaussianComponent ctor + decompose (gaussian.h:30-39, gaussian.cpp:36-63)
 *      and the GMM::GMM neighbour graph (gaussian_mixture.cpp:43-91) ---------- */
/* mean: K x 3, cov: K x 9 row-major, host pointers. Builds the device-resident
 * SoA (inverse, det, eigen axes/scales, chol(cov^-1), flags) and the
 * Bhattacharyya neighbour graph (CSR) on the GPU. */
int gl_gmm_create(gl_ctx_t* ctx, const double* mean, const double* cov, int K, const gl_params* prm,
                  gl_gmm_t** out);
/* .gmm stream reader: varint32 count, then count x {varint32 size, ComponentProto}
 * (protobuf_utils.cpp:12-29,42-80; GMM.proto:5-14). */
int gl_gmm_load_file(gl_ctx_t* ctx, const char* path, const gl_params* prm, gl_gmm_t** out);
/* GMMUtility::saveGMMModel (gmm_utils.cpp:69-119): same stream, byte-compatible. */
int gl_gmm_save_file(const gl_gmm_t* gmm, const char* path);
int gl_gmm_destroy(gl_gmm_t* gmm);
int gl_gmm_count(const gl_gmm_t* gmm);
/* Host-only halves of the two calls above (no device needed): parse a .gmm stream into
 * caller arrays (mean cap x 3, cov cap x 9 row-major; either may be NULL to query *K_out),
 * and write one (flags: bit0 is_degenerated, bit1 is_salient per component). */
int gl_gmm_file_read(const char* path, double* mean, double* cov, int cap, int* K_out);
int gl_gmm_file_write(const char* path, const double* mean, const double* cov, const uint8_t* flags, int K);

/* Map::summarize (map.cpp:162-188), host only: writes `timestamp tx ty tz qx qy qz qw` (T_wc, TUM format,
 * fixed notation, 6 / 9 digits) for N frames; pose_wc N x 7 in the library's (qx qy qz qw tx ty tz) order. */
int gl_write_tum_trajectory(const char* path, const double* stamps, const double* pose_wc, int N);

enum gl_gmm_field {
  GL_F_MEAN = 0,      /* K x 3 double */
  GL_F_COV = 1,       /* K x 9 double */
  GL_F_COV_INV = 2,   /* K x 9 double   cov_inv_                         */
  GL_F_DET = 3,       /* K double       det_                             */
  GL_F_SCALE = 4,     /* K x 3 double   scale_ (ascending eigenvalues)   */
  GL_F_AXIS = 5,      /* K x 9 double   axis_ row-major, column = vector */
  GL_F_SQRT_INFO = 6, /* K x 9 double   sqrt_info_ = chol_L(cov_inv_)    */
  GL_F_FLAGS = 7,     /* K uint8        bit0 is_degenerated, bit1 is_salient */
  GL_F_NBS_PTR = 8,   /* (K+1) int32    CSR row pointer of nbs_          */
  GL_F_NBS_IDX = 9,   /* nnz int32      neighbour component index        */
  GL_F_NBS_DIST = 10  /* nnz double     NeighbourInfo::dist              */
};
/* Copies a derived array to host memory (bytes = capacity of host_out). */
int gl_gmm_get(const gl_gmm_t* gmm, int field, void* host_out, size_t bytes);
int gl_gmm_nbs_count(const gl_gmm_t* gmm);

/* ---- association --------------------------------------------------------- */
enum gl_assoc_mode {
  GL_ASSOC_BRUTE = 0,       /* argmin_k GaussianComponent::chi2 (gaussian.cpp:65-70) over ALL K:
                               the north-star `associate`; first index wins ties.  Served by an
                               exact cell index built at gl_gmm_create (candidates whose chi2 <= 9
                               ellipsoid can reach the point's cell) + an all-pairs sweep of the
                               points it cannot resolve: identical output to GL_ASSOC_EXHAUSTIVE */
  GL_ASSOC_KNN5_EUCLID = 1, /* GMM::queryPoint (gaussian_mixture.cpp:545-576): nearest mean
                               of the exact 5-NN; d2 = its chi2                          */
  GL_ASSOC_EXHAUSTIVE = 2   /* the same argmin by the plain N x K sweep (no index)       */
};
/* pts_dev: N x 3; idx_dev: N int32; d2_dev: N double (may be NULL). */
int gl_associate3d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int mode, int32_t* idx_dev,
                   double* d2_dev);
/* Diagnostics of the cell index behind GL_ASSOC_BRUTE (gl_grid.hip).
 * info[8] = {enabled, cell size [m], dim x, dim y, dim z, entries (sum of list lengths),
 *            always-evaluated components, resolve threshold (chi2)}. */
int gl_gmm_index_info(const gl_gmm_t* gmm, double info[8]);
/* Device memory of that index: bytes[3] = {cell pointers, candidate lists, packed cell table}.  The packed table (16 bytes per
 * cell: one random read per point instead of two; 176 MB on the 4 096-Gaussian bench map) is built only within the memory
 * budget of the creating context's option assoc_pack_mb (default 512, 0 = never). */
int gl_gmm_index_bytes(const gl_gmm_t* gmm, double bytes[3]);
/* Number of (point, component) chi2 evaluations the index performs for these N points (its
 * algorithmic work, excluding the exhaustive sweep of unresolved points): *pairs_dev (device
 * int64) is overwritten. */
int gl_assoc_index_work(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int64_t* pairs_dev);

/* exact k-NN (k <= 8) on the 3-D means, ascending squared L2 (queryPoint's knnSearch).
 * idx_dev: N x k (-1 padded); dist_dev: N x k (may be NULL). */
int gl_knn3d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int k, int32_t* idx_dev,
             double* dist_dev);

/* GMM::renderView (gaussian_mixture.cpp:271-371) + GMM::searchCorrespondence
 * (gaussian_mixture.cpp:484-534) for B key-frames at once
 * (= GMMLoc::associateMapElements, gmmloc_opt.cpp:115-135).
 *  pose_dev: B x 7; uv_dev: B x N x 2; nfeat_dev: B int32 (<= N) or NULL (= N);
 *  cand_dev: B x N x k int32 parent indices in kNN order, gated by 2-D MDist2 < 9, -1 padded;
 *  ncand_dev: B x N int32;
 *  view_ids_dev (optional): B x view_cap int32 rendered parent ids sorted by depth
 *  descending (components2d_ order), -1 padded; nview_dev (optional): B int32. */
int gl_search2d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, int B, const double* pose_dev, int N,
                const double* uv_dev, const int32_t* nfeat_dev, int k, int32_t* cand_dev, int32_t* ncand_dev,
                int view_cap, int32_t* view_ids_dev, int32_t* nview_dev);

/* ---- feature matching (the producer of optimizeCurrentPose's correspondences) ---- */
/* ORBmatcher::searchByProjection(Frame&, mappts, stats, th) (orb_matcher.cpp:27-110) with
 * Frame::assignFeaturesToGrid / getFeaturesInArea (frame.cpp:54-79, 121-177),
 * ORBmatcher::DescriptorDistance (orb_matcher.cpp:580-596) and computeRadiusByViewingCos (:112-117),
 * for B frames of NF <= 3072 feature slots and NP <= 4096 projected map points.
 *  cam: width / height size the 64 x 48 feature grid (init_config.hpp:50-54); scale_factor: ORB pyramid
 *  factor (frame::scale_factor, 1.2).
 *  feat_uv B x NF x 2 double; feat_ur B x NF float (u_right, <= 0: none); feat_oct B x NF int32
 *  (< 0: empty slot); feat_desc B x NF x 32 bytes; feat_taken B x NF uint8 (1 = F.mappoints_[i] is set and
 *  has observations on entry);
 *  mp_uvr B x NP x 3 double (ProjStat::uvr); mp_level B x NP int32 (scale_pred, 0..7); mp_viewcos B x NP
 *  double; mp_valid B x NP uint8 (is_in_view_ && !not_valid_); mp_desc B x NP x 32 bytes;
 *  th (3 / 5 in searchLocalPoints, tracking.cpp:258-266), nn_ratio (0.8).
 *  out: feat_match B x NF int32 = index of the map point this call assigned to the feature
 *  (F.mappoints_[bestIdx] = mappt), -1 none; nmatches B int32 (the return value).
 *  The map points are processed in index order exactly like the reference loop: a feature assigned to a
 *  map point is not offered to the later ones. */
int gl_search_by_projection(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP,
                            const double* feat_uv_dev, const float* feat_ur_dev, const int32_t* feat_oct_dev,
                            const uint8_t* feat_desc_dev, const uint8_t* feat_taken_dev, const double* mp_uvr_dev,
                            const int32_t* mp_level_dev, const double* mp_viewcos_dev, const uint8_t* mp_valid_dev,
                            const uint8_t* mp_desc_dev, float th, float nn_ratio, int32_t* feat_match_dev,
                            int32_t* nmatches_dev);

/* ORBmatcher::searchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (orb_matcher.cpp:410-542) + computeThreeMaxima (:544-578): the matcher of
 * Tracking::trackWithMotionModel (tracking.cpp:334-350).  Per frame pair: pose_cw / pose_lw = getTcw() of the
 * current / last frame (B x 7); the current frame's features as above plus feat_angle B x NF float
 * (cv::KeyPoint::angle); the last frame's NL features: last_pt B x NL x 3 (position of its map point),
 * last_valid B x NL uint8 (mappoints_[i] && !is_outlier_[i]), last_oct B x NL int32, last_angle B x NL float,
 * last_desc B x NL x 32 (the map point's descriptor).  th = 7 or 14; mono = bMono; check_orientation =
 * ORBmatcher::check_orientation_.  cam supplies fx fy cx cy bf (as the float config scalars) and width /
 * height.  out: feat_match B x NF int32 = index i of the last-frame feature whose map point the feature
 * received, -1 none (after the rotation-consistency filter); nmatches B int32. */
int gl_search_by_projection_frame(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NL,
                                  const double* pose_cw_dev, const double* pose_lw_dev, const double* feat_uv_dev,
                                  const float* feat_ur_dev, const int32_t* feat_oct_dev, const float* feat_angle_dev,
                                  const uint8_t* feat_desc_dev, const uint8_t* feat_taken_dev, const double* last_pt_dev,
                                  const uint8_t* last_valid_dev, const int32_t* last_oct_dev,
                                  const float* last_angle_dev, const uint8_t* last_desc_dev, float th, int mono,
                                  int check_orientation, int32_t* feat_match_dev, int32_t* nmatches_dev);
/* ORBmatcher::searchForTriangulation (orb_matcher.cpp:141-293) with checkEpipolarDist (:119-139) and the rotation histogram
 * (computeThreeMaxima, :544-578) for B key-frame pairs: the matches Localization::createMapPoints triangulates
 * (localization_opt.cpp:266).  Per pair and key-frame k = 1, 2 (strides N1 / N2 features, NN1 / NN2 vocabulary nodes):
 *   uv{k}_dev B x N x 2 double, ur{k}_dev B x N float (u_right, < 0: mono), oct{k}_dev B x N int32 (< 0: padding slot),
 *   angle{k}_dev B x N float (key-point angle, degrees), desc{k}_dev B x N x 32 uint8, has_mp{k}_dev B x N uint8 (the feature
 *   already has a map point: getMapPoint(idx) != nullptr);
 *   the DBoW2::FeatureVector of the key-frame as CSR: nnode{k}_dev B int32 (nodes in use), node_id{k}_dev B x NN int32
 *   ASCENDING (std::map order), node_ptr{k}_dev B x (NN + 1) int32, node_idx{k}_dev B x N int32 (feature indices, list order);
 *   fmat_dev B x 9 double: MathUtils::computeFundamentalMatrix(Tcw1, K1, Tcw2, K2), row-major; epipole_dev B x 2 float:
 *   (ex, ey) of :155-160 - both built by the host (Eigen there), see INTEGRATION.md;
 *   scale_factor: frame::scale_factor (the level tables of init_config.hpp:63-79 are rebuilt from it);
 *   only_stereo: bOnlyStereo; check_orientation: ORBmatcher::check_orientation_.
 * out: match12_dev B x N1 int32 = matches12 (feature of key-frame 2 or -1; `matched_pairs` = its non-negative entries in
 * index order), nmatches_dev B int32 (the return value).  Order-exact: the same pairs as the sequential loop. */
int gl_search_for_triangulation(gl_ctx_t* ctx, float scale_factor, int B, int N1, int N2, int NN1, int NN2,
                                const double* uv1_dev, const float* ur1_dev, const int32_t* oct1_dev, const float* angle1_dev,
                                const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev,
                                const int32_t* node_id1_dev, const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev,
                                const double* uv2_dev, const float* ur2_dev, const int32_t* oct2_dev, const float* angle2_dev,
                                const uint8_t* desc2_dev, const uint8_t* has_mp2_dev, const int32_t* nnode2_dev,
                                const int32_t* node_id2_dev, const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev,
                                const double* fmat_dev, const float* epipole_dev, int only_stereo, int check_orientation,
                                int32_t* match12_dev, int32_t* nmatches_dev);
/* The projection / visibility loop in front of gl_search_by_projection and gl_fuse_search - Tracking::searchLocalPoints
 * (tracking.cpp:233-256), Localization::fuseObservations (localization.cpp:242-254): per map point Frame::project3 (frame.cpp:98-119,
 * pinhole_camera.cpp:46-66, 128-150) and MapPoint::checkScaleAndVisible (mappoint.cpp:257-303).  B frames x NP map points: pose_cw
 * B x 7 (getTcw), t_wc B x 3 (T_w_c_->translation()), pos / normal B x NP x 3 (getPosition, normal_), max_dist / min_dist B x NP float
 * (max_dist_, min_dist_), cand B x NP uint8 (the host's tests in front: non-null, valid, not seen / observed already).  Out: uvr
 * B x NP x 3, level B x NP int32 (ProjStat::scale_pred), viewcos / dist B x NP double (ProjStat), inview B x NP uint8 (is_in_view_ =
 * project3 && checkScaleAndVisible): exactly the mp_* inputs of the two matchers.  cam: fx fy cx cy bf as the float config scalars,
 * width / height. */
/* Host only (no device work): the seven steps of the predicted level of MapPoint::checkScaleAndVisible (mappoint.cpp:289-293) as
 * the HOST's libm gives them - step7[L] = the largest float ratio with ceil(logf(ratio) / logf(scale_factor)) <= L - which
 * gl_project_map_points compares against on the device.  scale_factor in (1, 1.7]. */
int gl_level_steps(float scale_factor, float* step7);
int gl_project_map_points(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NP, const double* pose_cw_dev,
                          const double* t_wc_dev, const double* pos_dev, const double* normal_dev, const float* max_dist_dev,
                          const float* min_dist_dev, const uint8_t* cand_dev, double* uvr_dev, int32_t* level_dev,
                          double* viewcos_dev, double* dist_dev, uint8_t* inview_dev);
/* Tracking::searchLocalPoints (tracking.cpp:213-270), the device part in one call: gl_project_map_points followed by
 * gl_search_by_projection (ORBmatcher(0.8), th 3 - or 5 for the first two frames) on its outputs, which stay in the context's
 * scratch.  Arguments as in the two calls; inview_dev (B x NP uint8, may be null): is_in_view_ per map point for the host's
 * num_visible_++ (tracking.cpp:251). */
int gl_search_local_points(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP, const double* feat_uv_dev,
                           const float* feat_ur_dev, const int32_t* feat_oct_dev, const uint8_t* feat_desc_dev,
                           const uint8_t* feat_taken_dev, const double* pose_cw_dev, const double* t_wc_dev, const double* mp_pos_dev,
                           const double* mp_normal_dev, const float* mp_max_dist_dev, const float* mp_min_dist_dev,
                           const uint8_t* mp_cand_dev, const uint8_t* mp_desc_dev, float th, float nn_ratio, int32_t* feat_match_dev,
                           int32_t* nmatches_dev, uint8_t* inview_dev);

/* One tracked frame, device resident (round 5; round 6: the fallback, temporal points, the two halves): what Tracking::track
 * (tracking.cpp:34-118) runs per frame - trackWithMotionModel (:333-376), trackKeyFrame when that fails (:297-331),
 * searchLocalPoints (:210-270), trackLocalMap (:272-299) - for B frames as ONE enqueued sequence on the context's stream, every
 * intermediate array in the context's scratch:
 *   1  gl_search_by_projection_frame(th_mm, check_orientation = 1): ORBmatcher(0.9, true).searchByProjection(curr, last, 7); a frame
 *      with fewer than 20 matches is searched again with 2 x th_mm (:340-346; the second launch skips the other frames)
 *   2  gl_optimize_current_pose on the matched features (Xw = the last frame's map point), then the outliers lose their map point
 *      and their is_outlier_ flag (:360-371; drop_src remembers the map point: it has been seen, :367).  counts2[0] = what
 *      trackWithMotionModel returns: the kept matches whose map point has observations (last_observed), 0 below 20 matches.
 *   2b (only with the key-frame buffers, kf_desc != NULL) a frame with counts2[0] < 10 (:50-58) goes through trackKeyFrame instead:
 *      gl_search_by_bow(0.7, check_orientation = 1) against the reference key-frame, pose = the LAST frame's, gl_optimize_current_pose,
 *      outliers dropped (drop_kf); its associations are then the key-frame's alone (match_last = -1, match_kf).  The other frames'
 *      workgroups return at once.  counts2[3] = 1 (tracked through the key-frame) or 2 (fewer than 10 kept matches: the reference
 *      reports a tracking failure, :66-71, and the later outputs of the frame mean nothing).
 *   3  gl_search_local_points from the refined pose (t_wc = -R^T t computed on the device): candidates = mp_cand minus the local map
 *      points the frame holds or dropped (last_visible_idx_ == idx, :243); a feature is taken if its map point has observations -
 *      a TEMPORAL point (createTemporalPoints, :44-46: no observation; last_observed = 0) stays matchable and is REPLACED by the
 *      local map point found for its feature (orb_matcher.cpp:74-76, 104: match_last of that feature ends as -1)
 *   4  gl_optimize_current_pose on all features with a map point (trackLocalMap, :276); its outliers are reported, not cleared.
 * THE LOCAL MAP: the reference rebuilds local_mappoints_ between 2 and 3 (Tracking::updateLocalMap, :119-207: the key-frames that
 * observe the frame's CURRENT map points and their neighbours) - host code on host containers.  gl_track_frame_chain takes ONE
 * local map, fixed before stage 1: it reproduces Tracking::track only if that list is the one updateLocalMap would produce
 * (e.g. the previous frame's local map when the covisibility set did not change); stage 3's matches are order-exact for the
 * list it is given, not for a list it never saw.  A host that wants the reference's sequence exactly calls the two halves -
 *      gl_track_frame_chain_front (1, 2, 2b)  ->  its own updateLocalMap  ->  gl_track_frame_chain_back (3, 4)
 * - one round trip instead of three; front needs drop_src (and drop_kf with the fallback) as OUTPUT buffers, back reads match_last /
 * match_kf / drop_src / drop_kf and the to_local maps against the NEW local map.
 * What the host keeps doing: num_visible_ / num_found_ / countObservations bookkeeping, the decision on counts2[0] / counts2[3]
 * (without the fallback buffers: on counts[0] < 20 as before).  All pointers are DEVICE pointers; layouts as in the single calls. */
typedef struct gl_track_chain_io {
  /* current frame: B x NF (x 2 / x 32) */
  const double* feat_uv;
  const float* feat_ur;
  const int32_t* feat_oct;
  const float* feat_angle;
  const uint8_t* feat_desc;
  const uint8_t* feat_taken; /* features that may not be matched at all (normally zeros) */
  /* last frame: B x NL; last_to_local: index of the feature's map point in the local map below, or -1 */
  const double* pose_lw;
  const double* last_pt;
  const uint8_t* last_valid;
  const int32_t* last_oct;
  const float* last_angle;
  const uint8_t* last_desc;
  const int32_t* last_to_local;
  /* local map: B x NP */
  const double* mp_pos;
  const double* mp_normal;
  const float* mp_max_dist;
  const float* mp_min_dist;
  const uint8_t* mp_cand;
  const uint8_t* mp_desc;
  /* in / out */
  double* pose_cw;      /* B x 7: in the motion-model prediction, out the pose after trackLocalMap (front: after stage 2 / 2b)       */
  double* pose_mm;      /* B x 7 out (may be NULL): the pose after stage 2 / 2b                                                     */
  int32_t* match_last;  /* B x NF out: last-frame feature whose map point the feature holds, or -1                                  */
  int32_t* match_local; /* B x NF out: local map point found in stage 3, or -1                                                      */
  uint8_t* outlier;     /* B x NF out: is_outlier_ after stage 4                                                                    */
  int32_t* counts;      /* B x 4 out: matches of stage 1 (after the retry), inliers of stage 2 (2b), matches of stage 3, inliers of stage 4 */
  uint8_t* inview;      /* B x NP out (may be NULL): is_in_view_ of stage 3                                                         */
  /* ---- round 6; every pointer below may be NULL (a zero-initialised struct behaves as in round 5) ---- */
  const uint8_t* last_observed; /* B x NL: countObservations() > 0 of the last-frame feature's map point (NULL: all of them)        */
  int32_t* drop_src;    /* B x NF out: the last-frame feature whose map point stage 2 dropped as an outlier, or -1                  */
  int32_t* counts2;     /* B x 4 out: {return value of trackWithMotionModel, searchByBoW matches, return value of trackKeyFrame,
                           mode 0 motion model / 1 key-frame / 2 lost}; required with the fallback                                  */
  /* the fallback (trackKeyFrame): the reference key-frame, side 1 of gl_search_by_bow, B x NK; kf_desc == NULL: no fallback        */
  int32_t NK, NNK, NNF; /* key-frame features; node capacities of the key-frame's and the frame's feature vectors                   */
  int32_t reserved_;
  const float* kf_angle;
  const uint8_t* kf_desc;
  const uint8_t* kf_has_mp;
  const int32_t* kf_nnode;
  const int32_t* kf_node_id;
  const int32_t* kf_node_ptr;
  const int32_t* kf_node_idx;
  const double* kf_pt;          /* B x NK x 3: position of the key-frame feature's map point                                         */
  const int32_t* kf_to_local;   /* B x NK: index of that map point in the local map, or -1                                           */
  const int32_t* feat_nnode;    /* the frame's DBoW2::FeatureVector as CSR (ORBVocabulary::transform, :298): B, B x NNF, B x (NNF+1), B x NF */
  const int32_t* feat_node_id;
  const int32_t* feat_node_ptr;
  const int32_t* feat_node_idx;
  int32_t* match_kf;    /* B x NF out: key-frame feature whose map point the feature holds (mode 1), or -1                          */
  int32_t* drop_kf;     /* B x NF out: the key-frame feature whose map point stage 2b dropped, or -1                                */
} gl_track_chain_io;
int gl_track_frame_chain(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                         const gl_track_chain_io* io, float th_mm, float th_local, float nn_ratio, int mono);
int gl_track_frame_chain_front(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                               const gl_track_chain_io* io, float th_mm, int mono);
int gl_track_frame_chain_back(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                              const gl_track_chain_io* io, float th_local, float nn_ratio);

/* Localization::fuseObservations (localization.cpp:226-318), the matching half, for B key-frames: per candidate map point the most
 * similar feature inside Frame::getFeaturesInArea(u, v, th * scale_factors[level]) (frame.cpp:121-177) with octave level - 1 or
 * level and Feature::error(uvr) * sigma2_inv[octave] within 5.99 (mono) / 7.8 (stereo).  Features as in gl_search_by_projection
 * (feat_uv B x NF x 2, feat_ur B x NF float (< 0: mono), feat_oct B x NF (< 0: padding slot), feat_desc B x NF x 32); map points:
 * mp_uvr B x NP x 3 = Frame::project3's (u, v, u_right), mp_level B x NP int32 = ProjStat::scale_pred, mp_valid B x NP uint8 = the
 * host's tests of :238-254 (non-null, valid, not observed by the key-frame, project3 and checkScaleAndVisible passed), mp_desc
 * B x NP x 32.  Out: best_idx B x NP int32 (the feature, if its distance is <= TH_LOW = 50, else -1) and best_dist B x NP int32 (256:
 * no candidate).  The map points do not interact in this loop; what the reference does with a match (:296-312: addObservation, or
 * replaceMapPoint by observation count) stays with the host, in list order.  cam supplies width / height (the 64 x 48 grid). */
int gl_fuse_search(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP, const double* feat_uv_dev,
                   const float* feat_ur_dev, const int32_t* feat_oct_dev, const uint8_t* feat_desc_dev, const double* mp_uvr_dev,
                   const int32_t* mp_level_dev, const uint8_t* mp_valid_dev, const uint8_t* mp_desc_dev, float th,
                   int32_t* best_idx_dev, int32_t* best_dist_dev);
/* ORBmatcher::searchByBoW (orb_matcher.cpp:295-408; computeThreeMaxima :544-578) for B key-frame / frame pairs: the matcher of
 * Tracking::trackReferenceKeyFrame (tracking.cpp:303), the last function of the reference's ORBmatcher.  Side 1 = the reference
 * key-frame: angle B x N1 float, desc B x N1 x 32, has_mp B x N1 uint8 (the feature holds a map point that is valid: `pMP &&
 * !pMP->not_valid_`), its DBoW2::FeatureVector as CSR like gl_search_for_triangulation's (nnode B; node_id B x NN1 ascending;
 * node_ptr B x (NN1 + 1); node_idx B x N1 in list order); side 2 = the current frame: angle, desc, feature vector.  nn_ratio /
 * check_orientation: the matcher's constructor arguments (tracking.cpp:300: 0.7, true).  Out: match21 B x N2 int32 - the
 * KEY-FRAME FEATURE whose map point the reference stores in matches[realIdxF], or -1 - and nmatches B.  Equal to the sequential
 * loop bit for bit (the first of equal best distances wins, a later equal one becomes second best). */
int gl_search_by_bow(gl_ctx_t* ctx, float nn_ratio, int check_orientation, int B, int N1, int N2, int NN1, int NN2,
                     const float* angle1_dev, const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev,
                     const int32_t* node_id1_dev, const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev,
                     const float* angle2_dev, const uint8_t* desc2_dev, const int32_t* nnode2_dev, const int32_t* node_id2_dev,
                     const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev, int32_t* match21_dev, int32_t* nmatches_dev);
/* The matches of gl_search_for_triangulation as the per-match arrays of gl_create_map_points, without leaving the device:
 * what Localization::createMapPoints reads per matched pair (localization_opt.cpp:286-420) - the two key-frames' poses,
 * key-points, depths, octaves and candidate components (kf->comps_[idx]).  Inputs per pair and key-frame: pose B x 7, uv B x N x 2,
 * ur / depth B x N float, oct B x N, cand B x N x k int32 (+ ncand B x N).  Output: the matches of all pairs, pair after pair, inside
 * a pair in ascending feature index of key-frame 1 (= matched_pairs), compacted: pair_off_dev B + 1 int32 (exclusive scan of the
 * counts; [B] = total), m_* arrays of `cap` entries (entries beyond cap are dropped: size cap >= sum of nmatches, e.g. B x
 * min(N1, N2)), m_pair / m_idx1 / m_idx2: the pair and the two feature indices of every match. */
int gl_gather_triangulation_matches(gl_ctx_t* ctx, int B, int N1, int N2, int k, int cap, const int32_t* match12_dev,
                                    const int32_t* nmatches_dev, const double* pose1_dev, const double* uv1_dev, const float* ur1_dev,
                                    const float* depth1_dev, const int32_t* oct1_dev, const int32_t* cand1_dev, const int32_t* ncand1_dev,
                                    const double* pose2_dev, const double* uv2_dev, const float* ur2_dev, const float* depth2_dev,
                                    const int32_t* oct2_dev, const int32_t* cand2_dev, const int32_t* ncand2_dev, int32_t* pair_off_dev,
                                    double* m_pose1_dev, double* m_uvr1_dev, float* m_depth1_dev, int32_t* m_oct1_dev, int32_t* m_cand1_dev,
                                    int32_t* m_n1_dev, double* m_pose2_dev, double* m_uvr2_dev, float* m_depth2_dev, int32_t* m_oct2_dev,
                                    int32_t* m_cand2_dev, int32_t* m_n2_dev, int32_t* m_pair_dev, int32_t* m_idx1_dev, int32_t* m_idx2_dev);

/* ---- point refinement ----------------------------------------------------- */
/* GMMLoc::optimizePoint (gmmloc_opt.cpp:260-342), N independent problems.
 * pts N x 3, uvr N x 3 (u, v, u_right), octave N, pose N x 7, comp N, proj_z2 N.
 * out: res N uint8, chi2_proj N, chi2_str N, pt_est N x 3.  A problem without a component (comp < 0 or
 * >= K) or with an octave outside 0..7 is not solved: res = 0, chi2 = 0, pt_est = pts. */
int gl_optimize_point(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int N,
                      const double* pts_dev, const double* uvr_dev, const int32_t* octave_dev,
                      const double* pose_dev, const int32_t* comp_dev, const double* proj_z2_dev,
                      uint8_t* res_dev, double* chi2_proj_dev, double* chi2_str_dev, double* pt_est_dev);

/* GMMLoc::checkMapAssociation (gmmloc_opt.cpp:156-258) for the N features of B
 * key-frames: pose_dev B x 7; pts_dev B x N x 3 in/out (written where the reference
 * writes pt3d); uvr B x N x 3; octave B x N (<0 = skip feature); cand B x N x k,
 * ncand B x N (from gl_search2d); out_comp B x N (component or -1). */
int gl_check_map_association(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm,
                             int B, int N, const double* pose_dev, double* pts_dev, const double* uvr_dev,
                             const int32_t* octave_dev, const int32_t* cand_dev, const int32_t* ncand_dev, int k,
                             int32_t* out_comp_dev);

/* Localization::optimizeTriangulationVec (localization_opt.cpp:27-204), N problems.
 * x3d N x 3 in/out; pose1/pose2 N x 7; uvr1/uvr2 N x 3 (u_right < 0 => mono edge);
 * oct1/oct2 N; cand1/cand2 N x k with n1/n2 N; out_comp N (an octave outside 0..7: -1, x3d untouched). */
int gl_optimize_triangulation(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm,
                              int N, double* x3d_dev, const double* pose1_dev, const double* uvr1_dev,
                              const int32_t* oct1_dev, const double* pose2_dev, const double* uvr2_dev,
                              const int32_t* oct2_dev, const int32_t* cand1_dev, const int32_t* n1_dev,
                              const int32_t* cand2_dev, const int32_t* n2_dev, int k, int32_t* out_comp_dev);

/* Localization::createMapPoints, the per-match block (localization_opt.cpp:286-420), for
 * N epipolar matches: parallax test -> linear triangulation (smallest right singular vector of the 4 x 4
 * system, the reference's JacobiSVD) or stereo unprojection (frame.cpp:27-35) -> optimizeTriangulationVec
 * (= gl_optimize_triangulation, with u_right taken as -1 unless depth > 0, :116-137) -> project3 into both
 * key-frames, reprojection checks (both with kp1's sigma^2, :370-391) and scale consistency (:393-404).
 *  pose1 / pose2 N x 7 (getTcw of the two key-frames); uvr1 / uvr2 N x 3 (u, v, u_right; < 0 = monocular);
 *  depth1 / depth2 N float (Feature::depth, -1 = none); oct1 / oct2 N; candidate tables as in
 *  gl_optimize_triangulation; scale_factor = frame::scale_factor (1.2).
 *  out: x3d N x 3 (the point after the structure optimisation; zeros when no point was formed),
 *  type N int32: 0 = rejected, 1 FromTriMono, 2 FromTriMonoGMM, 3 FromTriStereo, 4 FromTriStereoGMM
 *  (MapPoint::type_, :407-419), comp N int32 (str_ptr as component index, -1 = none). */
int gl_create_map_points(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, float scale_factor,
                         int N, const double* pose1_dev, const double* uvr1_dev, const float* depth1_dev,
                         const int32_t* oct1_dev, const double* pose2_dev, const double* uvr2_dev,
                         const float* depth2_dev, const int32_t* oct2_dev, const int32_t* cand1_dev,
                         const int32_t* n1_dev, const int32_t* cand2_dev, const int32_t* n2_dev, int k, double* x3d_dev,
                         int32_t* type_dev, int32_t* comp_dev);

/* ---- pose refinement ------------------------------------------------------ */
/* Tracking::optimizeCurrentPose (tracking_opt.cpp:21-217) for B frames.
 *  pose_dev B x 7 in/out; Xw_dev B x M x 3; obs_dev B x M x 3 (u, v, u_right; u_right < 0
 *  => monocular edge); octave_dev B x M int32 (< 0 => feature has no map point);
 *  outlier_dev B x M uint8 (is_outlier_), in/out: rewritten for the features with a map point, left untouched
 *  for the others (the reference resets is_outlier_[i] only where mappoints_[i] exists, :63-69);
 *  ninlier_dev B int32 (return value). */
int gl_optimize_current_pose(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M,
                             double* pose_dev, const double* Xw_dev, const double* obs_dev,
                             const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev);

/* Localization::jointOptimization (localization_opt.cpp:456-925) on B flat problems
 * sharing one shape.  Per problem: P free poses (local key-frames, [0,P)), F fixed
 * poses ([P,P+F)), L points (all marginalised) each with <= 1 GMM association, and
 * observations in CSR order by point.
 *  poses_dev   B x (P+F) x 7   in/out for [0,P)
 *  prior_dev   B x P uint8     1 = key-frame idx_ 0 (prior edge / fixed, :556-581)
 *  points_dev  B x L x 3       in/out
 *  assoc_dev   B x L int32     component or -1
 *  obs_ptr_dev B x (L+1) int32; obs_pose_dev B x NOBS int32; obs_uvr_dev B x NOBS x 3;
 *  obs_oct_dev B x NOBS int32  (NOBS = stride; only obs_ptr[L] entries are used)
 *  out: assoc_dropped_dev B x L uint8 (:837-853), obs_erase_dev B x NOBS uint8 (:855-879),
 *       iters_dev B int32 (actual_iter of the last optimize(40), :827-828). */
int gl_joint_optimization(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B,
                          int P, int F, int L, int NOBS, double* poses_dev, const uint8_t* prior_dev,
                          double* points_dev, const int32_t* assoc_dev, const int32_t* obs_ptr_dev,
                          const int32_t* obs_pose_dev, const double* obs_uvr_dev, const int32_t* obs_oct_dev,
                          uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev);

/* The same with the reference's stop word (`pbStopFlag`, Localization::jointOptimization's abort: localization_opt.cpp:541-542
 * setForceStopFlag, :765-767, :792-796).  stop_flag: one int32 in device or host-mapped memory (gl_malloc / gl_malloc_host),
 * read with system scope once per Levenberg trial and acted on where g2o tests terminate() - before an outer iteration:
 *   > 0 on entry      the call returns at once, nothing is written (the reference's `return` at :765-767; iters = 0);
 *   > 0 later         the running outer iteration finishes, no further one starts; the reprojection gating and optimize(40)
 *                     are skipped (bDoMore = false), outputs are those of the last accepted step;
 *   < 0               a BUDGET of -value outer iterations in total (deterministic; what the parity tests use);
 *   0 / NULL          gl_joint_optimization. */
int gl_joint_optimization_stoppable(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B,
                                    int P, int F, int L, int NOBS, double* poses_dev, const uint8_t* prior_dev,
                                    double* points_dev, const int32_t* assoc_dev, const int32_t* obs_ptr_dev,
                                    const int32_t* obs_pose_dev, const double* obs_uvr_dev, const int32_t* obs_oct_dev,
                                    uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                                    const int32_t* stop_flag);

/* North-star per-frame path: associate + structure-constrained pose refinement for B
 * frames of M map points each:
 *   1. idx = argmin_k chi2_k(Xw)  (GL_ASSOC_BRUTE), association kept iff chi2 <= 9
 *      (the gate of checkMapAssociation, gmmloc_opt.cpp:230-232);
 *   2. jointOptimization restricted to the frame: 1 free pose, M free marginalised points,
 *      one reprojection edge per point (mono / stereo, Huber) + its GMM edge
 *      (EdgePt2GaussianDeg x ba_lambda2 or EdgePt2Gaussian), schedule 5 / 5 / 40.
 *  pose_dev B x 7 in/out; Xw_dev B x M x 3 in/out; obs_dev B x M x 3; octave_dev B x M
 *  (<0 = no point); assoc_dev B x M int32 out (association after the final gate, -1 = none);
 *  d2_dev B x M double out (chi2 of the argmin at the INPUT point, may be NULL). */
int gl_track_frames(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B, int M,
                    double* pose_dev, double* Xw_dev, const double* obs_dev, const int32_t* octave_dev,
                    int32_t* assoc_dev, double* d2_dev);
/* The same with a gauge anchor.  The reference never runs its structure BA without one: the other observers of the
 * local map points enter as FIXED key-frames (localization_opt.cpp:491-516) and key-frame 0 carries an EdgeSE3QuatPrior
 * (factors.cpp:19-53; measurement = its pose on entry, sigma_rot 2 deg, sigma_t 1 cm) or is fixed itself when
 * !ba_first_as_prior (:556-581).  Per frame:
 *   prior_dev       B uint8 or NULL; 1 = the frame's pose is "key-frame 0": prior edge on pose_dev's input value
 *                   (gl_params.ba_first_as_prior != 0) or fixed pose (== 0: only the points move);
 *   F               fixed observer key-frames per frame, 0 .. GL_TRACK_MAX_FIXED;
 *   fixed_pose_dev  B x F x 7; fixed_obs_dev B x M x F x 3 (u, v, u_right; u_right < 0 => mono);
 *   fixed_oct_dev   B x M x F int32 (< 0: point not observed by that key-frame);
 *   fixed_erase_dev B x M x F uint8 out or NULL (observations the reference would erase, :855-879).
 * F == 0 runs on the on-chip refine of gl_track_frames (the prior costs one 6x6 block per Levenberg trial); F > 0 is
 * packed into flat problems (P = 1) and solved by the general kernel of gl_joint_optimization.  Same arithmetic as
 * jointOptimization with P = 1: parity tests against the oracle's joint_optimization(P = 1, prior / F fixed). */
#define GL_TRACK_MAX_FIXED 8
typedef struct gl_track_anchor {
  const uint8_t* prior_dev;
  int32_t F;
  const double* fixed_pose_dev;
  const double* fixed_obs_dev;
  const int32_t* fixed_oct_dev;
  uint8_t* fixed_erase_dev;
} gl_track_anchor;
int gl_track_frames_anchored(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B, int M,
                             double* pose_dev, double* Xw_dev, const double* obs_dev, const int32_t* octave_dev,
                             int32_t* assoc_dev, double* d2_dev, const gl_track_anchor* anchor);
/* The same for ONE frame with HOST buffers in and out - what the reference's tracking thread would call once per frame
 * (tracking.cpp:274,312,356).  The context keeps a page-locked staging buffer and its device mirror (grown on demand),
 * enqueues one copy each way around gl_track_frames(B = 1) on its stream and synchronises once; pose_host (7) and
 * Xw_host (M x 3) are updated in place, assoc_host (M) receives the associations.  Blocking. */
int gl_track_frame_host(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int M,
                        double* pose_host, double* Xw_host, const double* obs_host, const int32_t* octave_host,
                        int32_t* assoc_host);
/* ... anchored by the prior edge on the input pose (gl_track_frames_anchored with prior = 1, F = 0) */
int gl_track_frame_host_anchored(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int M,
                                 double* pose_host, double* Xw_host, const double* obs_host, const int32_t* octave_host,
                                 int32_t* assoc_host);

/* ---- device memory helpers for hosts without their own HIP allocator ------ */
int gl_malloc(gl_ctx_t* ctx, size_t bytes, void** dev_out);
int gl_free(gl_ctx_t* ctx, void* dev);
int gl_memcpy_h2d(gl_ctx_t* ctx, void* dst_dev, const void* src, size_t bytes); /* synchronous */
int gl_memcpy_d2h(gl_ctx_t* ctx, void* dst, const void* src_dev, size_t bytes); /* synchronous */
/* The frame-at-a-time host path (tracking.cpp:274,312,356 call once per frame): page-locked staging memory and
 * copies that are only ENQUEUED on the context's stream, so a frame costs one gl_ctx_synchronize instead of one per
 * transfer.  The host side of an _async copy must come from gl_malloc_host and stay untouched until the
 * synchronize. */
int gl_malloc_host(gl_ctx_t* ctx, size_t bytes, void** host_out);
int gl_free_host(gl_ctx_t* ctx, void* host);
int gl_memcpy_h2d_async(gl_ctx_t* ctx, void* dst_dev, const void* src_pinned, size_t bytes);
int gl_memcpy_d2h_async(gl_ctx_t* ctx, void* dst_pinned, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GMMLOC_HIP_H_ */
