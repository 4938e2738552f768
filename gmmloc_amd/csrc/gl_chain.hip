// gl_track_frame_chain: one tracked frame without host round trips ("Next" #6; round 6: the trackKeyFrame fallback,
// temporal points, and the chain in two halves around the host's updateLocalMap).
// Tracking::trackWithMotionModel (tracking.cpp:326-376) -> Tracking::searchLocalPoints (:210-270) -> Tracking::trackLocalMap
// (:272-299) as ONE sequence of launches on the context's stream: the four device stages that existed as entry points of their
// own (gl_search_by_projection_frame, gl_optimize_current_pose, gl_search_local_points, gl_optimize_current_pose) and the glue
// the host did between them - gathering a matched feature's map point into the pose problem, dropping the outliers of the first
// optimisation, marking what the frame has already seen - as four small kernels.  Every intermediate array lives in the
// context's third scratch block (the stages themselves use the first two).
#include "gl_internal.hpp"
#include "gl_pose_compact.hpp"

namespace {

// Eigen Quaternion * Vector3 (as gl_match.hip / the oracle: uv = 2 q.vec x v;  v + w uv + q.vec x uv)
__device__ __forceinline__ void quat_rot_c(const double* q, const double* v, double* o) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  o[0] = v[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]);
  o[1] = v[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]);
  o[2] = v[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]);
}

// the pose problem of Tracking::optimizeCurrentPose from the frame's associations (tracking_opt.cpp:63-133): a feature with a map
// point contributes {Xw = the point's position, obs = (u, v, u_right), octave}; the others get octave -1.  A feature's map point is
// its local-map match (stage 3: ORBmatcher::searchByProjection writes F.mappoints_[bestIdx] = mappt over whatever the feature held,
// orb_matcher.cpp:104 - it only ever picks features without a map point or with an UNOBSERVED one, :74-76), else its last-frame
// match (stage 1), else its key-frame match (the trackKeyFrame fallback).
struct ChainSrc {  // edge i of ONE frame (pointers at the frame's rows; match_* may be null)
  const double* feat_uv;
  const float* feat_ur;
  const int32_t* feat_oct;
  const int32_t* match_last;
  const double* last_pt;
  const int32_t* match_local;
  const double* mp_pos;
  const int32_t* match_kf;
  const double* kf_pt;
  __device__ const double* point(int i) const {
    const int k = match_local ? match_local[i] : -1;
    if (k >= 0) return mp_pos + (size_t)k * 3;
    const int j = match_last ? match_last[i] : -1;
    if (j >= 0) return last_pt + (size_t)j * 3;
    const int q = match_kf ? match_kf[i] : -1;
    return q >= 0 ? kf_pt + (size_t)q * 3 : nullptr;
  }
  __device__ int octave(int i) const { return point(i) ? feat_oct[i] : -1; }
  __device__ void load(int i, double* X, double* O) const {
    const double* P = point(i);
    X[0] = P ? P[0] : 0.0;
    X[1] = P ? P[1] : 0.0;
    X[2] = P ? P[2] : 0.0;
    O[0] = feat_uv[(size_t)i * 2];
    O[1] = feat_uv[(size_t)i * 2 + 1];
    O[2] = (double)feat_ur[i];
  }
};
// One workgroup per frame: the frame's pose problem at full stride (Xw, obs, oct: what a frame with more edges than the compacted
// stride runs on) AND compacted (gl_pose_compact.hpp; pc.MC == 0: not compacted), plus the small things that would each be a launch of
// their own (a dependent launch costs the device ~5 us whatever it does, profiles/r6_chain_trace.txt): the flags cleared
// (outlier_clear), a last-frame match that stage 3 replaced cleared (`finalise`: match_last / match_local / match_kf name the
// feature's ONE map point when the chain returns), the preceding search's count copied into counts[., nm_slot], the fallback's start
// pose (pose_dst := pose_src).  gate: only the frames with gate[b] != 0 get edges (trackKeyFrame: the flagged frames).
constexpr int T_GATHER = 1024;  // (a frame's 1 200 features in two rounds: every pass is a chain of dependent loads)
__global__ __launch_bounds__(T_GATHER) void k_chain_gather(int B, int NF, int NL, int NP, int NK, const double* __restrict__ feat_uv,
                                                     const float* __restrict__ feat_ur, const int32_t* __restrict__ feat_oct,
                                                     int32_t* match_last, const double* __restrict__ last_pt, const int32_t* __restrict__ match_local,
                                                     const double* __restrict__ mp_pos, const int32_t* __restrict__ match_kf,
                                                     const double* __restrict__ kf_pt, const int32_t* __restrict__ gate, double* __restrict__ Xw,
                                                     double* __restrict__ obs, int32_t* __restrict__ oct, gl::PoseCompacted pc, int finalise,
                                                     uint8_t* __restrict__ outlier_clear, const int32_t* __restrict__ nm, int32_t* __restrict__ counts,
                                                     int nm_slot, const double* __restrict__ pose_src, double* __restrict__ pose_dst) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) return;
  const bool on = !gate || gate[b] != 0;
  const size_t fb = (size_t)b * NF;
  const ChainSrc src = {feat_uv + fb * 2, feat_ur + fb, feat_oct + fb, (on && match_last) ? match_last + fb : nullptr, last_pt + (size_t)b * NL * 3,
                        (on && match_local) ? match_local + fb : nullptr, mp_pos + (size_t)b * NP * 3, (on && match_kf) ? match_kf + fb : nullptr,
                        kf_pt ? kf_pt + (size_t)b * NK * 3 : nullptr};
  // (the full-stride problem is what a frame with more edges than the compacted stride runs on: only such a frame writes it - 62 KB of
  //  the 190 KB a frame of 1 200 features moved here)
  const bool full = pc.MC > 0 ? gl::pose_compact_frame<T_GATHER>(src, b, NF, pc) : true;
  for (int i = tid; i < NF; i += T_GATHER) {
    const size_t g = fb + i;
    if (full) {
      double X[3], O[3];
      src.load(i, X, O);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Xw[g * 3 + k] = X[k];
        obs[g * 3 + k] = O[k];
      }
      oct[g] = src.octave(i);
    }
    if (outlier_clear) outlier_clear[g] = 0;
  }
  if (finalise && match_last && match_local) {  // (after the last read of the frame's associations; the local match wins either way)
    for (int i = tid; i < NF; i += T_GATHER)
      if (match_local[fb + i] >= 0 && match_last[fb + i] >= 0) match_last[fb + i] = -1;
  }
  if (tid == 0 && nm) counts[(size_t)b * 4 + nm_slot] = nm[b];
  if (tid < 7 && pose_dst) pose_dst[(size_t)b * 7 + tid] = pose_src[(size_t)b * 7 + tid];
}

// The glue between the stages, one workgroup per frame, the parts a call needs in ONE launch:
//  GLUE_AFTER_MM - after the first optimisation (tracking.cpp:360-374): an outlier's feature loses its map point and its flag (drop_src
//    remembers which last-frame feature it had: that map point has been SEEN by this frame, last_visible_idx_ = idx, :367); what
//    trackWithMotionModel returns is the number of kept matches whose map point has observations (:370) - 0 when the search found fewer
//    than 20 (:344-345).  Tracking::track (:50-58) falls back to trackKeyFrame when that number is below 10: fb_flag.
//  GLUE_AFTER_FB - what trackKeyFrame does with its optimisation's result (:313-330), for the flagged frames: the outliers lose their map
//    point and their flag (seen: drop_kf), the pose is the optimised one, and the frame's associations are the key-frame's alone
//    (`curr_frame_->mappoints_ = mappts` dropped every last-frame match).  mode 1: tracked through the key-frame; 2: fewer than 10 kept
//    matches, the reference returns "tracking failure" (:66-71) and the later stages' outputs of the frame mean nothing.
//  GLUE_POSE_MM - the pose after stage 2 / 2b copied out (pose_mm).
//  GLUE_BEFORE_LOCAL - before searchLocalPoints (tracking.cpp:213-243): every map point the frame holds or has dropped as an outlier has
//    been seen (last_visible_idx_ == idx: no candidate, :243 - the outliers of BOTH optimisations of a frame that went through the
//    fallback); a feature is taken if its map point has observations (orb_matcher.cpp:74-76: a TEMPORAL point - createTemporalPoints,
//    tracking.cpp:44-46: no observation - stays replaceable); T_w_c.translation() of the refined pose (Frame::setTcw; SE3Quat::inverse:
//    r = conj(q), t = r * (-t)).
// Every per-feature loop walks i = thread, thread + 256, ...: a feature's words are written and read again by the SAME thread; the
// frame-wide things (cand, the pose) are ordered by the barriers between the parts.
enum { GLUE_AFTER_MM = 1, GLUE_AFTER_FB = 2, GLUE_POSE_MM = 4, GLUE_BEFORE_LOCAL = 8 };
struct GlueArgs {
  int B, NF, NL, NP, NK, has_fallback;
  int32_t* match_last;
  int32_t* match_kf;  // null: no key-frame buffers
  uint8_t* outlier;
  const uint8_t* last_observed;
  const int32_t* nm_all;
  int32_t* drop_src;
  int32_t* drop_kf;
  int32_t* counts;
  int32_t* counts2;
  int32_t* fb_flag;
  // after_fb
  const double* pose_fb;
  const uint8_t* outl_fb;
  const int32_t* nm_bow;
  const int32_t* ninl_fb;
  double* pose_cw;
  double* pose_mm;
  // before_local
  const int32_t* last_to_local;
  const int32_t* kf_to_local;
  const uint8_t* feat_taken0;
  const uint8_t* mp_cand0;
  uint8_t* taken;
  uint8_t* cand;
  double* t_wc;
};
template <int PARTS>
__global__ __launch_bounds__(256) void k_chain_glue(GlueArgs a) {
  __shared__ int s_cnt[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int B = a.B, NF = a.NF, NL = a.NL, NP = a.NP, NK = a.NK;
  if (b >= B) return;
  if (PARTS & GLUE_AFTER_MM) {
    int nmap = 0;
    for (int i = tid; i < NF; i += 256) {
      const size_t g = (size_t)b * NF + i;
      int j = a.match_last[g], d = -1;
      if (j >= 0) {
        if (a.outlier[g]) {
          d = j;
          j = -1;
          a.match_last[g] = -1;
          a.outlier[g] = 0;
        } else if (!a.last_observed || a.last_observed[(size_t)b * NL + j]) {
          ++nmap;
        }
      }
      a.drop_src[g] = d;
      if (a.match_kf) {
        a.match_kf[g] = -1;
        a.drop_kf[g] = -1;
      }
    }
    for (int o = 32; o > 0; o >>= 1) nmap += __shfl_xor(nmap, o);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = nmap;
    __syncthreads();
    if (tid == 0) {
      const int nm = a.nm_all[b];
      const int ret = nm < 20 ? 0 : s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      const int fb = (a.has_fallback && ret < 10) ? 1 : 0;
      if (a.counts2) {
        a.counts2[(size_t)b * 4] = ret;
        a.counts2[(size_t)b * 4 + 1] = 0;
        a.counts2[(size_t)b * 4 + 2] = 0;
        a.counts2[(size_t)b * 4 + 3] = fb;
      }
      a.fb_flag[b] = fb;
    }
    __syncthreads();
  }
  if ((PARTS & GLUE_AFTER_FB) && a.fb_flag[b] != 0) {  // (workgroup-uniform)
    int kept = 0;
    for (int i = tid; i < NF; i += 256) {
      const size_t g = (size_t)b * NF + i;
      int q = a.match_kf[g], d = -1;
      if (q >= 0 && a.outl_fb[g]) {
        d = q;
        q = -1;
      }
      a.match_kf[g] = q;
      a.match_last[g] = -1;
      a.outlier[g] = 0;
      a.drop_kf[g] = d;
      kept += q >= 0;  // (a key-frame's map point is observed by that key-frame: countObservations() > 0, :327)
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = kept;
    __syncthreads();
    if (tid < 7) a.pose_cw[(size_t)b * 7 + tid] = a.pose_fb[(size_t)b * 7 + tid];
    if (tid == 0) {
      const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      a.counts[(size_t)b * 4 + 1] = a.ninl_fb[b];
      a.counts2[(size_t)b * 4 + 1] = a.nm_bow[b];
      a.counts2[(size_t)b * 4 + 2] = n;
      a.counts2[(size_t)b * 4 + 3] = n < 10 ? 2 : 1;
    }
    __syncthreads();
  }
  if ((PARTS & GLUE_POSE_MM) && a.pose_mm && tid < 7) a.pose_mm[(size_t)b * 7 + tid] = a.pose_cw[(size_t)b * 7 + tid];
  if (PARTS & GLUE_BEFORE_LOCAL) {
    for (int m = tid; m < NP; m += 256) a.cand[(size_t)b * NP + m] = a.mp_cand0[(size_t)b * NP + m];
    __syncthreads();
    for (int i = tid; i < NF; i += 256) {
      const size_t g = (size_t)b * NF + i;
      const int j = a.match_last[g], q = a.match_kf ? a.match_kf[g] : -1, d = a.drop_src ? a.drop_src[g] : -1, dk = (a.match_kf && a.drop_kf) ? a.drop_kf[g] : -1;
      bool tk = a.feat_taken0[g] != 0;
      if (j >= 0) {
        const int l = a.last_to_local[(size_t)b * NL + j];
        if (l >= 0 && l < NP) a.cand[(size_t)b * NP + l] = 0;
        tk = tk || !a.last_observed || a.last_observed[(size_t)b * NL + j] != 0;
      }
      if (q >= 0) {
        const int l = a.kf_to_local ? a.kf_to_local[(size_t)b * NK + q] : -1;
        if (l >= 0 && l < NP) a.cand[(size_t)b * NP + l] = 0;
        tk = true;
      }
      if (d >= 0) {
        const int l = a.last_to_local[(size_t)b * NL + d];
        if (l >= 0 && l < NP) a.cand[(size_t)b * NP + l] = 0;
      }
      if (dk >= 0) {
        const int l = a.kf_to_local ? a.kf_to_local[(size_t)b * NK + dk] : -1;
        if (l >= 0 && l < NP) a.cand[(size_t)b * NP + l] = 0;
      }
      a.taken[g] = tk ? 1 : 0;
    }
    if (tid == 0) {
      const double* p = a.pose_cw + (size_t)b * 7;
      const double qi[4] = {-p[0], -p[1], -p[2], p[3]};
      const double mt[3] = {p[4] * -1., p[5] * -1., p[6] * -1.};
      double o[3];
      quat_rot_c(qi, mt, o);
      a.t_wc[(size_t)b * 3] = o[0];
      a.t_wc[(size_t)b * 3 + 1] = o[1];
      a.t_wc[(size_t)b * 3 + 2] = o[2];
    }
  }
}

// the chain's intermediates in the context's third scratch block
struct ChainScratch {
  double *Xw, *obs, *t_wc, *pose_fb;
  int32_t *oct, *nm, *ninl, *fb_flag, *nm_bow, *drop_src, *drop_kf;
  uint8_t *taken, *cand, *outl_fb;
  gl::PoseCompacted pc;  // pc.MC == 0: the pose problems are not compacted (option pose_compact, NF)
};
int chain_scratch(gl::Ctx* c, int B, int NF, int NP, ChainScratch* S) {
  const size_t nf = (size_t)B * NF, np = (size_t)B * NP;
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  const int MC = gl::pose_compact_stride((int)c->opt.pose_compact, NF, (int)c->opt.pose_compact_cap);
  void* scratch = nullptr;
  const int rc = gl::ctx_scratch_c(c, 2 * up(nf * 24) + 3 * up(nf * 4) + 2 * up(nf) + up(np) + up((size_t)B * 24) + up((size_t)B * 56) + 4 * up((size_t)B * 4) +
                                       (MC ? gl::pose_compacted_bytes(B, NF, MC) : 0), &scratch);
  if (rc != GL_OK) return rc;
  char* s = (char*)scratch;
  auto take = [&](size_t bytes) {
    char* p = s;
    s += up(bytes);
    return (void*)p;
  };
  S->Xw = (double*)take(nf * 24);
  S->obs = (double*)take(nf * 24);
  S->oct = (int32_t*)take(nf * 4);
  S->drop_src = (int32_t*)take(nf * 4);
  S->drop_kf = (int32_t*)take(nf * 4);
  S->taken = (uint8_t*)take(nf);
  S->outl_fb = (uint8_t*)take(nf);
  S->cand = (uint8_t*)take(np);
  S->t_wc = (double*)take((size_t)B * 24);
  S->pose_fb = (double*)take((size_t)B * 56);
  S->nm = (int32_t*)take((size_t)B * 4);
  S->ninl = (int32_t*)take((size_t)B * 4);
  S->fb_flag = (int32_t*)take((size_t)B * 4);
  S->nm_bow = (int32_t*)take((size_t)B * 4);
  S->pc = gl::PoseCompacted{};
  if (MC) gl::pose_compacted_place(s, B, NF, MC, &S->pc);
  return GL_OK;
}
// optimizeCurrentPose of the chain in TWO launches: the gather (the problem of the frame's current associations, compacted: the frame's
// 1 200 feature slots hold a few hundred edges) and the optimisation, which writes the outlier flags in feature order and the inlier
// count where they belong itself.  match_last / match_local / match_kf may be null; nm_slot >= 0: the preceding search's count S.nm goes
// to counts[., nm_slot] on the way.
struct ChainOpt {
  double* pose;                // in / out
  int32_t* match_last;         // or null
  const int32_t* match_local;  // or null
  const int32_t* match_kf;     // or null
  const int32_t* gate;         // or null: only the frames with gate[b] != 0 have edges
  int finalise;
  uint8_t* outlier;            // flags out
  bool clear_outlier;
  int32_t* ninl;               // inlier counts out, ninl[b * nin_stride]
  int nin_stride;
  int nm_slot;
  const double* pose_src;      // pose := pose_src first (the fallback starts from the last frame's pose), or null
};
int chain_optimise(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int NF, int NL, int NP, int NK, const gl_track_chain_io* io,
                   const ChainScratch& S, const ChainOpt& o) {
  gl::Ctx* c = gl::C(ctx);
  k_chain_gather<<<B, T_GATHER, 0, c->stream>>>(B, NF, NL, NP, NK, io->feat_uv, io->feat_ur, io->feat_oct, o.match_last, io->last_pt, o.match_local, io->mp_pos,
                                           o.match_kf, io->kf_pt, o.gate, S.Xw, S.obs, S.oct, S.pc, o.finalise, o.clear_outlier ? o.outlier : nullptr,
                                           o.nm_slot >= 0 ? S.nm : nullptr, io->counts, o.nm_slot, o.pose_src, o.pose_src ? o.pose : nullptr);
  GL_HIP(hipGetLastError());
  if (S.pc.MC > 0) return gl::pose_compacted_launch(ctx, cam, prm, B, NF, o.pose, S.Xw, S.obs, S.oct, o.outlier, o.ninl, o.nin_stride, S.pc);
  return gl::optimize_current_pose_plain(ctx, cam, prm, B, NF, o.pose, S.Xw, S.obs, S.oct, o.outlier, o.ninl, o.nin_stride);
}

bool chain_has_fallback(const gl_track_chain_io* io) { return io->kf_desc != nullptr; }

int chain_check(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int NF, int NL, int NP, const gl_track_chain_io* io) {
  GL_REQUIRE(ctx && cam && prm && io, "null argument");
  GL_REQUIRE(B >= 0 && NF >= 1 && NL >= 1 && NP >= 1, "bad B / NF / NL / NP");
  GL_REQUIRE(io->feat_uv && io->feat_ur && io->feat_oct && io->feat_angle && io->feat_desc && io->feat_taken && io->pose_lw && io->last_pt &&
                 io->last_valid && io->last_oct && io->last_angle && io->last_desc && io->last_to_local && io->mp_pos && io->mp_normal &&
                 io->mp_max_dist && io->mp_min_dist && io->mp_cand && io->mp_desc && io->pose_cw && io->match_last && io->match_local &&
                 io->outlier && io->counts,
             "null buffer");
  if (chain_has_fallback(io)) {
    GL_REQUIRE(io->NK >= 1 && io->NNK >= 1 && io->NNF >= 1, "bad NK / NNK / NNF of the key-frame fallback");
    GL_REQUIRE(io->kf_angle && io->kf_has_mp && io->kf_nnode && io->kf_node_id && io->kf_node_ptr && io->kf_node_idx && io->kf_pt && io->kf_to_local &&
                   io->feat_nnode && io->feat_node_id && io->feat_node_ptr && io->feat_node_idx && io->match_kf && io->counts2,
               "null buffer of the key-frame fallback");
  }
  return GL_OK;
}

GlueArgs glue_args(int B, int NF, int NL, int NP, const gl_track_chain_io* io, const ChainScratch& S, int32_t* drop_src, int32_t* drop_kf) {
  const bool fbk = chain_has_fallback(io);
  GlueArgs a;
  a.B = B;
  a.NF = NF;
  a.NL = NL;
  a.NP = NP;
  a.NK = fbk ? io->NK : 1;
  a.has_fallback = fbk ? 1 : 0;
  a.match_last = io->match_last;
  a.match_kf = fbk ? io->match_kf : nullptr;
  a.outlier = io->outlier;
  a.last_observed = io->last_observed;
  a.nm_all = S.nm;
  a.drop_src = drop_src;
  a.drop_kf = drop_kf;
  a.counts = io->counts;
  a.counts2 = io->counts2;
  a.fb_flag = S.fb_flag;
  a.pose_fb = S.pose_fb;
  a.outl_fb = S.outl_fb;
  a.nm_bow = S.nm_bow;
  a.ninl_fb = S.ninl;
  a.pose_cw = io->pose_cw;
  a.pose_mm = io->pose_mm;
  a.last_to_local = io->last_to_local;
  a.kf_to_local = fbk ? io->kf_to_local : nullptr;
  a.feat_taken0 = io->feat_taken;
  a.mp_cand0 = io->mp_cand;
  a.taken = S.taken;
  a.cand = S.cand;
  a.t_wc = S.t_wc;
  return a;
}
template <int PARTS>
int glue_launch(gl::Ctx* c, const GlueArgs& a) {
  k_chain_glue<PARTS><<<a.B, 256, 0, c->stream>>>(a);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// stages 1 - 2 (+ the fallback): trackWithMotionModel, trackKeyFrame.  through_local: the glue in front of stage 3 rides in the front's
// last launch (the one-call chain); else the front ends with the pose copied out.
int chain_front(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP, const gl_track_chain_io* io,
                float th_mm, int mono, const ChainScratch& S, int32_t* drop_src, int32_t* drop_kf, bool through_local) {
  gl::Ctx* c = gl::C(ctx);
  const bool fbk = chain_has_fallback(io);
  const int NK = fbk ? io->NK : 1;
  int rc;
  // ---- stage 1: ORBmatcher(0.9, true).searchByProjection(curr, last, th), again with 2 th where fewer than 20 matches (tracking.cpp:330-342)
  rc = gl_search_by_projection_frame(ctx, cam, scale_factor, B, NF, NL, io->pose_cw, io->pose_lw, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_angle,
                                     io->feat_desc, io->feat_taken, io->last_pt, io->last_valid, io->last_oct, io->last_angle, io->last_desc, th_mm, mono, 1,
                                     io->match_last, S.nm);
  if (rc != GL_OK) return rc;
  rc = gl::launch_match_frame_gated(ctx, cam, scale_factor, B, NF, NL, io->pose_cw, io->pose_lw, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_angle,
                                    io->feat_desc, io->feat_taken, io->last_pt, io->last_valid, io->last_oct, io->last_angle, io->last_desc, 2 * th_mm, mono, 1,
                                    io->match_last, S.nm, S.nm, 20);
  if (rc != GL_OK) return rc;
  // ---- stage 2: optimizeCurrentPose on the matched features, outliers dropped (:348-371)
  ChainOpt o2 = {io->pose_cw, io->match_last, nullptr, nullptr, nullptr, 0, io->outlier, true, io->counts + 1, 4, 0, nullptr};
  rc = chain_optimise(ctx, cam, prm, B, NF, NL, NP, NK, io, S, o2);
  if (rc != GL_OK) return rc;
  const GlueArgs ga = glue_args(B, NF, NL, NP, io, S, drop_src, drop_kf);
  if (!fbk) return through_local ? glue_launch<GLUE_AFTER_MM | GLUE_POSE_MM | GLUE_BEFORE_LOCAL>(c, ga) : glue_launch<GLUE_AFTER_MM | GLUE_POSE_MM>(c, ga);
  rc = glue_launch<GLUE_AFTER_MM>(c, ga);
  if (rc != GL_OK) return rc;
  // ---- Tracking::trackKeyFrame (:297-331) for the flagged frames: ORBmatcher(0.7, true).searchByBoW(ref_keyframe_, curr), the last
  // frame's pose, optimizeCurrentPose, outliers dropped.  The workgroups of the other frames return at once (no edge: their
  // optimisation returns at `edges().size() < 10`, on a scratch pose).
  rc = gl::launch_bow_gated(ctx, 0.7f, 1, B, io->NK, NF, io->NNK, io->NNF, io->kf_angle, io->kf_desc, io->kf_has_mp, io->kf_nnode, io->kf_node_id,
                            io->kf_node_ptr, io->kf_node_idx, io->feat_angle, io->feat_desc, io->feat_nnode, io->feat_node_id, io->feat_node_ptr,
                            io->feat_node_idx, io->match_kf, S.nm_bow, S.fb_flag);
  if (rc != GL_OK) return rc;
  ChainOpt ofb = {S.pose_fb, nullptr, nullptr, io->match_kf, S.fb_flag, 0, S.outl_fb, true, S.ninl, 1, -1, io->pose_lw};
  rc = chain_optimise(ctx, cam, prm, B, NF, NL, NP, NK, io, S, ofb);
  if (rc != GL_OK) return rc;
  return through_local ? glue_launch<GLUE_AFTER_FB | GLUE_POSE_MM | GLUE_BEFORE_LOCAL>(c, ga) : glue_launch<GLUE_AFTER_FB | GLUE_POSE_MM>(c, ga);
}

// stages 3 - 4: searchLocalPoints, trackLocalMap (glue_done: the front's last launch already prepared stage 3)
int chain_back(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP, const gl_track_chain_io* io,
               float th_local, float nn_ratio, const ChainScratch& S, int32_t* drop_src, int32_t* drop_kf, bool glue_done) {
  gl::Ctx* c = gl::C(ctx);
  const bool fbk = chain_has_fallback(io);
  const int NK = fbk ? io->NK : 1;
  int rc;
  if (!glue_done) {
    rc = glue_launch<GLUE_BEFORE_LOCAL>(c, glue_args(B, NF, NL, NP, io, S, drop_src, drop_kf));
    if (rc != GL_OK) return rc;
  }
  // ---- stage 3: searchLocalPoints from the refined pose (:210-270)
  rc = gl_search_local_points(ctx, cam, scale_factor, B, NF, NP, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_desc, S.taken, io->pose_cw, S.t_wc, io->mp_pos,
                              io->mp_normal, io->mp_max_dist, io->mp_min_dist, S.cand, io->mp_desc, th_local, nn_ratio, io->match_local, S.nm, io->inview);
  if (rc != GL_OK) return rc;
  // ---- stage 4: trackLocalMap's optimizeCurrentPose on every feature with a map point (:272-299)
  ChainOpt o4 = {io->pose_cw, io->match_last, io->match_local, fbk ? io->match_kf : nullptr, nullptr, 1, io->outlier, false, io->counts + 3, 4, 2, nullptr};
  return chain_optimise(ctx, cam, prm, B, NF, NL, NP, NK, io, S, o4);
}

}  // namespace

extern "C" int gl_track_frame_chain(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                                    const gl_track_chain_io* io, float th_mm, float th_local, float nn_ratio, int mono) {
  int rc = chain_check(ctx, cam, prm, B, NF, NL, NP, io);
  if (rc != GL_OK || B == 0) return rc;
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  ChainScratch S;
  rc = chain_scratch(c, B, NF, NP, &S);
  if (rc != GL_OK) return rc;
  int32_t* drop = io->drop_src ? io->drop_src : S.drop_src;
  int32_t* dropk = io->drop_kf ? io->drop_kf : S.drop_kf;
  rc = chain_front(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_mm, mono, S, drop, dropk, true);
  if (rc != GL_OK) return rc;
  return chain_back(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_local, nn_ratio, S, drop, dropk, true);
}

extern "C" int gl_track_frame_chain_front(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                                          const gl_track_chain_io* io, float th_mm, int mono) {
  int rc = chain_check(ctx, cam, prm, B, NF, NL, NP, io);
  if (rc != GL_OK || B == 0) return rc;
  GL_REQUIRE(io->drop_src && (!chain_has_fallback(io) || io->drop_kf),
             "null drop_src / drop_kf (the split chain hands the dropped matches to the host and to gl_track_frame_chain_back)");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  ChainScratch S;
  rc = chain_scratch(c, B, NF, NP, &S);
  if (rc != GL_OK) return rc;
  return chain_front(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_mm, mono, S, io->drop_src, io->drop_kf ? io->drop_kf : S.drop_kf, false);
}

extern "C" int gl_track_frame_chain_back(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                                         const gl_track_chain_io* io, float th_local, float nn_ratio) {
  int rc = chain_check(ctx, cam, prm, B, NF, NL, NP, io);
  if (rc != GL_OK || B == 0) return rc;
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  ChainScratch S;
  rc = chain_scratch(c, B, NF, NP, &S);
  if (rc != GL_OK) return rc;
  return chain_back(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_local, nn_ratio, S, io->drop_src, io->drop_kf, false);
}
