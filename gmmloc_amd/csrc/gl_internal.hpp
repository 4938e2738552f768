// Internal host-side structures of libgmmloc_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "gmmloc_hip.h"

namespace gl {

void set_error(const char* fmt, ...);

#define GL_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      gl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return GL_ERR_DEVICE;                                                           \
    }                                                                                 \
  } while (0)

#define GL_REQUIRE(cond, msg)              \
  do {                                     \
    if (!(cond)) {                         \
      gl::set_error("%s: %s", __func__, msg); \
      return GL_ERR_ARG;                   \
    }                                      \
  } while (0)

// the documented size limits of the matcher-side entry points assume gfx950's 160 KB of LDS: on a device with less the REAL limit is
// the LDS the shape needs, reported as such instead of as a raw HIP error of the launch
#define GL_REQUIRE_LDS(c, bytes)                                                                                              \
  do {                                                                                                                        \
    if ((size_t)(bytes) > (size_t)(c)->lds_max) {                                                                             \
      gl::set_error("%s: this shape needs %zu bytes of LDS per workgroup, the device has %d", __func__, (size_t)(bytes), (c)->lds_max); \
      return GL_ERR_ARG;                                                                                                      \
    }                                                                                                                         \
  } while (0)

// Device-resident, immutable GMM (SoA).  Layout (all fp64 unless noted):
//   rec12   K x 12   {mean[3], cov_inv[9]}  -- the association record, 96 B
//   cov     K x 9    row-major covariance
//   det     K
//   scale   K x 3    ascending eigenvalues
//   axis    K x 9    row-major, column c = eigenvector c
//   sqrt_info K x 9  lower Cholesky factor of cov_inv
//   flags   K uint8  bit0 degenerated, bit1 salient
//   plane   K x 8    {normal[3] (= axis col 0), mean[3], pad, pad}
//   nbs_ptr K+1 int32, nbs_idx nnz int32, nbs_dist nnz  (CSR of nbs_)
// Exact cell index of the Mahalanobis argmin (gl_grid.hip): CSR lists of candidate components per
// grid cell + a short list of components every point must evaluate.
struct CellIndex {
  bool enabled = false;
  double lo[3] = {0, 0, 0};
  double h = 0, inv_h = 0, t_resolve = 0;
  int dim[3] = {0, 0, 0};
  int nglob = 0;
  size_t nnz = 0, ncell = 0;
  int32_t* ptr = nullptr;   // ncell + 1
  int32_t* idx = nullptr;   // nnz, ascending within a cell
  int32_t* glob = nullptr;  // nglob, ascending
  // packed cells (round 3): {count, then the list itself when count <= 3, else the offset into idx}: ONE random 16-byte
  // read per point where ptr[c], ptr[c + 1] and idx[...] are two (the kernel is bound by the fabric's random sectors)
  void* cell4 = nullptr;    // ncell x int4, or null (table above the memory budget, or the 8-byte form in use)
  void* cell8 = nullptr;    // ncell x 8 bytes (round 6: option assoc_cell8, K < 2^20), or null
  // the records again, one per 128-byte line (K x 16 doubles, the last four unused): a 96-byte record of rec12 lies across
  // two lines three times out of four, and the cooperative gather is bound by line requests (round 5; null without cell4)
  double* rec16 = nullptr;
};

struct Gmm {
  int device = 0;
  int K = 0;
  std::vector<double> h_mean, h_cov;  // host copies (save_file / get)
  double* rec12 = nullptr;
  double* mean = nullptr;  // K x 3 (kNN kernels)
  double* cov = nullptr;
  double* det = nullptr;
  double* scale = nullptr;
  double* axis = nullptr;
  double* sqrt_info = nullptr;
  double* hgw = nullptr;  // K x 6 sym: sqrt_info * sqrt_info^T (EdgePt2Gaussian J^T J, world frame)
  double* plane4 = nullptr;  // K x 4: axis_.col(0) and its dot product with the mean (EdgePt2GaussianDeg's plane)
  uint8_t* flags = nullptr;
  int32_t* nbs_ptr = nullptr;
  int32_t* nbs_idx = nullptr;
  double* nbs_dist = nullptr;
  int nnz = 0;
  CellIndex grid;
  gl_params prm;
};

// Tuning / test knobs of a context.  Read ONCE from the environment (GMMLOC_<NAME>) at gl_ctx_create and
// changed afterwards only through gl_ctx_set_option: no libc environment scan on the call paths.
struct Options {
  double ba_shape = -1;         // gl_track_frames refine: -1 auto, 0 one workgroup per frame, 1 one point per thread
  double ba_step32 = 0;         // 1: point step from an fp32 cache of the pass-A solve (faster, not the default)
  double ba_persist = 1;        // 1: the batch-shaped refine runs as persistent workgroups that draw their frames from a queue (0: one block per frame; A/B)
  double ba_two_frames = 0;     // 1: 2 000-point class of the plain batch refine as TWO frames per CU (bafd2000x: two groups per wave, hand-over slots in global memory; same bits)
  double ba_slow = 0;           // 1: general kernel k_ba1 also for M <= 2000 (A/B)
  double ba_fixed_pack = 0;     // 1: gl_track_frames_anchored with fixed observers always through the general kernel (k_track_pack -> k_ba_gen; A/B, tests)
  double pose_waves = 0;        // gl_optimize_current_pose: 0 auto, 1 / 4 / 8 waves per frame
  double ba_same_xcd = 0;          //   1: the latency shape may use the same-XCD form of its exchange (outside the HIP memory model: opt-in, gmmloc_hip.h)
  double ba_rendezvous_us = 200;   // time limit of EVERY exchange of the latency shape (the fallback it protects costs ~0.5 ms; 0: a workgroup
                                   // gives up at its first unsuccessful look -> follow-up kernel; tests)
  double ba_test_abort_seq = 0;    // tests: n > 0 makes the last workgroup of every frame give up at its n-th exchange
  double pose_regs = 1;         //   0: the frame-at-a-time shapes read their edges from global memory every trial (A/B)
  double bagen_nb = 0;          // gl_joint_optimization, persistent kernel: 0 auto, n workgroups per problem
  double bagen_mode = 0;        //   0: by WINDOW size (never by B), 1: the persistent kernel k_ba_gen, 2: the pipelined shape, 3: persistent, whole batch in one launch (gmmloc_hip.h)
  double view_slot_lds = 0;     // gl_search2d: accepted-list records kept in LDS (0 = all that fit)
  double view_threads = 0;      //   0 auto, 256 / 1024
  double assoc_index_min = -1;  // pairs below which GL_ASSOC_BRUTE stays on the sweep (-1 = built-in)
  double assoc_grid = -1;       // 0: never use the cell index (every association is the N x K sweep); A/B and bench
  double assoc_coop_bal = 1;    // 1: the cooperative gather's pairs are evaluated one per lane and round (needs assoc_coop_long), 0: by the lane that owns the point (A/B; same results)
  double assoc_coop_long = 1;   // 1: lists of more than three candidates go through the cooperative gather too, 0: the lane walks them alone (A/B; same results)
  double assoc_rec_pad = 1;     // 1: k_assoc_cells_coop gathers from the one-line-per-record copy (CellIndex::rec16), 0: from rec12 (A/B; same results)
  double assoc_coop = 1;        // 1: wave-cooperative record gather in the indexed association (k_assoc_cells_coop), 0: a lane per record
  double pipe_fuse_asm = -1;    // pipelined local BA: the solve kernel assembles the system itself (-1: calls of a few windows, 0 never, 1 always; same bits)
  double pose_compact_cap = 1024;  // the largest stride of a compacted pose problem (tests: 512 / 256 drive frames into the full-stride problem)
  double fuse_records = 1;      // gl_fuse_search: 0 = the walk that reads the features from global memory (A/B)
  double pose_compact = -1;     // gl_optimize_current_pose: problems of more than 1 024 slots compacted to 1 024 where their edges fit (-1, default); 1: every problem of more than 256 slots; 0: never
  double assoc_cell8 = 1;       // the packed cell table in 8 bytes per cell where the component indices fit 20 bits (0: 16 bytes per cell as in rounds 3 - 5)
  double assoc_pack_mb = 512;   // memory budget (MB) of the packed cell table a GMM built with this context may add to its cell index (0: none)
  double assoc_cell = 0;        // > 0: cell size (m) of the index instead of the automatic one (tuning)
  double assoc_globcells = 0;   // > 0: components whose box covers more cells than this are evaluated for every point (tuning)
  double match_desc_lds = -1;   // gl_search_by_projection: descriptors in LDS (-1 auto, 0 / 1)
  double pipe_lanes = -1;       // pipelined local BA in batches: streams a call's windows are split over (-1 auto: 2 from 16 windows; 1 .. 4; A/B)
  double pipe_judge = -1;       //   the verdict on a trial as a kernel of its own (-1 auto: from 32 windows per lane; 0 / 1; A/B)
  double schur_kper = -1;       //   chunks of a block one wave of the Schur pass takes (-1 auto; 1 / 2 / 4 / 8; A/B)
};
// name -> member; nullptr if unknown
double* option_slot(Options& o, const char* name);

struct Ctx {
  int device = 0;
  int ncu = 256;  // compute units of the device (shape decisions: frames vs CUs)
  int lds_max = 160 * 1024;  // LDS a workgroup may have on this device (hipDeviceAttributeMaxSharedMemoryPerBlock: 160 KB on gfx950, 64 KB on gfx942)
  Options opt;
  // per-context (= per device, per host thread) caches of driver queries: the dynamic-LDS limit already set
  // for a kernel (hipFuncSetAttribute is per device) and occupancy answers
  std::map<const void*, size_t> lds_limit;
  std::map<std::pair<const void*, size_t>, int> occupancy;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // the device reports its XCC id per workgroup and places block b on XCD b % 8 (probed once at gl_ctx_create): the
  // latency-shape kernels may then use the same-XCD form of their exchange - after checking the ids again at run time
  bool xcc_ids_trusted = false;
  // scratch (grown on demand)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* scratch_b = nullptr;  // second block (ctx_scratch_b): the matchers' candidate cache
  size_t scratch_b_bytes = 0;
  void* scratch_c = nullptr;  // third block (ctx_scratch_c): the intermediates of gl_track_frame_chain, whose stages use the other two
  size_t scratch_c_bytes = 0;
  // staging of the frame-at-a-time host entry point (gl_track_frame_host): page-locked + device mirror, grown on demand
  void* host_stage = nullptr;
  void* dev_stage = nullptr;
  size_t stage_bytes = 0;
  // device counters (gl_ctx_counter_read): [0] frames of latency-shape launches redone by the follow-up kernel
  int32_t* counters = nullptr;
  long coop_fallbacks = 0;  // host side (GL_COUNTER_BA_COOP_FALLBACK): windows of gl_joint_optimization run with one workgroup after a refused cooperative launch
  int* host_word = nullptr;  // page-locked words (the pipelined local BA's counts of unfinished problems, per lane)
  hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};  // further lanes of the pipelined local BA in batches (launch_ba_pipe)
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int pipe_hint = 0;         // cycles the last pipelined local BA needed (the next call enqueues that many + 2 ahead)
  // optional statistics buffer (gl_ctx_set_stats_buffer)
  int32_t* stats = nullptr;
  int32_t* stats_iters = nullptr;  // (gl_ctx_set_stats_buffers) outer Levenberg iterations per frame
  int stats_n = 0;
  int32_t* stats_edges = nullptr;  // (gl_ctx_set_edge_stats_buffer) n x 2: level-0 reprojection edges x trials / x outer iterations
  int stats_edges_n = 0;
  // timing
  bool timing = false;
  double timer_ms[GL_TIMER_COUNT] = {0};
  int64_t timer_n[GL_TIMER_COUNT] = {0};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};

int ctx_scratch(Ctx* c, size_t bytes, void** out);
int ctx_scratch_b(Ctx* c, size_t bytes, void** out);
int ctx_scratch_c(Ctx* c, size_t bytes, void** out);
bool probe_xcc_ids(Ctx* c);  // gl_ba_fast.hip
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE and costs a driver call: a context (one device,
// one host thread) remembers the limit it has set for each kernel and raises it only when it grows.  The caller
// has made the context's device current.
inline hipError_t ensure_dynamic_lds(Ctx* c, const void* kernel, size_t bytes) {
  size_t& have = c->lds_limit[kernel];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
// bracket a launch region with events when timing is enabled
struct TimerScope {
  Ctx* c;
  int id;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  TimerScope(Ctx* ctx, int timer);
  ~TimerScope();
};

inline gl::Gmm* G(const gl_gmm_t* g) { return (gl::Gmm*)g; }
inline gl::Ctx* C(gl_ctx_t* c) { return (gl::Ctx*)c; }

// .gmm stream (gl_io.cpp)
int read_gmm_file(const char* path, std::vector<double>& mean, std::vector<double>& cov);
int write_gmm_file(const char* path, const double* mean, const double* cov, const uint8_t* flags, int K);

// fixed observer key-frames of gl_track_frames_anchored (the caller's device arrays: gmmloc_hip.h, gl_track_anchor)
struct TrackFixed {
  int F;
  const double* pose;   // B x F x 7
  const double* obs;    // B x M x F x 3
  const int32_t* oct;   // B x M x F
  uint8_t* erase;       // B x M x F out, or null
};
size_t ba1_scratch_bytes(int B, int L, int F);

// launchers implemented across the .hip files
int launch_build_components(Ctx* c, Gmm* g);
int launch_build_neighbours(Ctx* c, Gmm* g);
int build_cell_index(Ctx* c, Gmm* g);
void free_cell_index(Gmm* g);
// gl_search_by_projection_frame with a per-frame gate (gl_match.hip): frames with gate_nm[f] >= gate_min are left untouched
struct PoseCompacted;  // gl_pose_compact.hpp
// (gl_refine_pose.hip) problems whose compacted form exists: see there
int pose_compacted_launch(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M, double* pose_dev, const double* Xw_dev,
                          const double* obs_dev, const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev, int nin_stride,
                          const PoseCompacted& pc);
int optimize_current_pose_plain(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M, double* pose_dev, const double* Xw_dev,
                                const double* obs_dev, const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev, int nin_stride);
int launch_bow_gated(gl_ctx_t* ctx, float nn_ratio, int check_orientation, int B, int N1, int N2, int NN1, int NN2, const float* angle1_dev,
                     const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev, const int32_t* node_id1_dev,
                     const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev, const float* angle2_dev, const uint8_t* desc2_dev,
                     const int32_t* nnode2_dev, const int32_t* node_id2_dev, const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev,
                     int32_t* match21_dev, int32_t* nmatches_dev, const int32_t* run_flag);
int launch_match_frame_gated(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NL, const double* pose_cw, const double* pose_lw,
                             const double* feat_uv, const float* feat_ur, const int32_t* feat_oct, const float* feat_angle, const uint8_t* feat_desc,
                             const uint8_t* feat_taken, const double* last_pt, const uint8_t* last_valid, const int32_t* last_oct, const float* last_angle,
                             const uint8_t* last_desc, float th, int mono, int check_orientation, int32_t* feat_match, int32_t* nmatches,
                             const int32_t* gate_nm, int gate_min);
// association launchers (gl_assoc.hip: all-pairs sweep; gl_grid.hip: cell index + sweep of the rest)
int launch_assoc_brute(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2);
int launch_assoc_sweep(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2, const int32_t* list,
                       const int32_t* count_dev, void* scratch);
size_t assoc_scratch_bytes(int K, int N, bool listed = false);
int launch_assoc_index(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2, bool resolve_all,
                       void* scratch);
size_t assoc_index_scratch_bytes(int K, int N, bool resolve_all);

}  // namespace gl
