// gl_track_frame_chain: one tracked frame without host round trips ("Next" #6; round 6: the trackKeyFrame fallback,
// temporal points, and the chain in two halves around the host's updateLocalMap).
// Tracking::trackWithMotionModel (tracking.cpp:326-376) -> Tracking::searchLocalPoints (:210-270) -> Tracking::trackLocalMap
// (:272-299) as ONE sequence of launches on the context's stream: the four device stages that existed as entry points of their
// own (gl_search_by_projection_frame, gl_optimize_current_pose, gl_search_local_points, gl_optimize_current_pose) and the glue
// the host did between them - gathering a matched feature's map point into the pose problem, dropping the outliers of the first
// optimisation, marking what the frame has already seen - as four small kernels.  Every intermediate array lives in the
// context's third scratch block (the stages themselves use the first two).
#include "gl_internal.hpp"

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
// match (stage 1), else its key-frame match (the trackKeyFrame fallback).  `finalise`: a last-frame match that stage 3 replaced is
// cleared, so that match_last / match_local / match_kf name the feature's ONE map point when the chain returns.
__global__ __launch_bounds__(256) void k_chain_pose_inputs(int B, int NF, int NL, int NP, int NK, const double* __restrict__ feat_uv,
                                                          const float* __restrict__ feat_ur, const int32_t* __restrict__ feat_oct,
                                                          int32_t* __restrict__ match_last, const double* __restrict__ last_pt,
                                                          const int32_t* __restrict__ match_local, const double* __restrict__ mp_pos,
                                                          const int32_t* __restrict__ match_kf, const double* __restrict__ kf_pt,
                                                          double* __restrict__ Xw, double* __restrict__ obs, int32_t* __restrict__ oct, int finalise) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)B * NF) return;
  const size_t b = g / NF;
  const int j = match_last[g], k = match_local ? match_local[g] : -1, q = match_kf ? match_kf[g] : -1;
  const double* X = k >= 0 ? mp_pos + (b * NP + k) * 3 : (j >= 0 ? last_pt + (b * NL + j) * 3 : (q >= 0 ? kf_pt + (b * NK + q) * 3 : nullptr));
  if (finalise && k >= 0 && j >= 0) match_last[g] = -1;
  Xw[g * 3] = X ? X[0] : 0.0;
  Xw[g * 3 + 1] = X ? X[1] : 0.0;
  Xw[g * 3 + 2] = X ? X[2] : 0.0;
  obs[g * 3] = feat_uv[g * 2];
  obs[g * 3 + 1] = feat_uv[g * 2 + 1];
  obs[g * 3 + 2] = (double)feat_ur[g];
  oct[g] = X ? feat_oct[g] : -1;
}

// after the first optimisation (tracking.cpp:360-374): an outlier's feature loses its map point and its flag (drop_src remembers which
// last-frame feature it had: that map point has been SEEN by this frame, last_visible_idx_ = idx, :367); what trackWithMotionModel
// returns is the number of kept matches whose map point has observations (:370) - 0 when the search found fewer than 20 (:344-345).
// Tracking::track (:50-58) falls back to trackKeyFrame when that number is below 10: fb_flag.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_chain_after_mm(int B, int NF, int NL, int32_t* __restrict__ match_last, uint8_t* __restrict__ outlier,
                                                       const uint8_t* __restrict__ last_observed, const int32_t* __restrict__ nm_all,
                                                       int32_t* __restrict__ drop_src, int32_t* __restrict__ match_kf, int32_t* __restrict__ drop_kf,
                                                       int32_t* __restrict__ counts2, int32_t* __restrict__ fb_flag, int has_fallback) {
  __shared__ int s_cnt[4];
  const int b = blockIdx.x;
  if (b >= B) return;
  int nmap = 0;
  for (int i = threadIdx.x; i < NF; i += 256) {
    const size_t g = (size_t)b * NF + i;
    int j = match_last[g], d = -1;
    if (j >= 0) {
      if (outlier[g]) {
        d = j;
        j = -1;
        match_last[g] = -1;
        outlier[g] = 0;
      } else if (!last_observed || last_observed[(size_t)b * NL + j]) {
        ++nmap;
      }
    }
    drop_src[g] = d;
    if (match_kf) {
      match_kf[g] = -1;
      drop_kf[g] = -1;
    }
  }
  for (int o = 32; o > 0; o >>= 1) nmap += __shfl_xor(nmap, o);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = nmap;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nm = nm_all[b];
    const int ret = nm < 20 ? 0 : s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const int fb = (has_fallback && ret < 10) ? 1 : 0;
    if (counts2) {
      counts2[(size_t)b * 4] = ret;
      counts2[(size_t)b * 4 + 1] = 0;
      counts2[(size_t)b * 4 + 2] = 0;
      counts2[(size_t)b * 4 + 3] = fb;
    }
    fb_flag[b] = fb;
  }
}

// ---- the fallback: Tracking::trackKeyFrame (tracking.cpp:297-331) for the frames k_chain_after_mm flagged ------------------------------
// the pose problem of its optimizeCurrentPose: curr_frame_->mappoints_ = the searchByBoW matches (:308), Tcw = the LAST frame's (:309).
// The other frames get no edge at all (their optimisation returns at `edges().size() < 10`) and a scratch pose.
__global__ __launch_bounds__(256) void k_chain_fb_inputs(int B, int NF, int NK, const int32_t* __restrict__ fb_flag, const double* __restrict__ feat_uv,
                                                        const float* __restrict__ feat_ur, const int32_t* __restrict__ feat_oct,
                                                        const int32_t* __restrict__ match_kf, const double* __restrict__ kf_pt,
                                                        const double* __restrict__ pose_lw, double* __restrict__ pose_fb, double* __restrict__ Xw,
                                                        double* __restrict__ obs, int32_t* __restrict__ oct) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)B * NF) return;
  const size_t b = g / NF;
  const int i = (int)(g - b * NF);
  const bool fb = fb_flag[b] != 0;
  const int q = fb ? match_kf[g] : -1;
  const double* X = q >= 0 ? kf_pt + (b * NK + q) * 3 : nullptr;
  Xw[g * 3] = X ? X[0] : 0.0;
  Xw[g * 3 + 1] = X ? X[1] : 0.0;
  Xw[g * 3 + 2] = X ? X[2] : 0.0;
  obs[g * 3] = feat_uv[g * 2];
  obs[g * 3 + 1] = feat_uv[g * 2 + 1];
  obs[g * 3 + 2] = (double)feat_ur[g];
  oct[g] = X ? feat_oct[g] : -1;
  if (i < 7) pose_fb[b * 7 + i] = pose_lw[b * 7 + i];
}
// ... and what trackKeyFrame does with the result (:313-330): the outliers lose their map point and their flag (seen: drop_src), the
// pose is the optimised one, and the frame's associations are the key-frame's alone (`curr_frame_->mappoints_ = mappts` dropped every
// last-frame match).  mode 1: tracked through the key-frame; 2: fewer than 10 kept matches, the reference returns "tracking failure"
// (:66-71) and the later stages' outputs of the frame mean nothing.
__global__ __launch_bounds__(256) void k_chain_after_fb(int B, int NF, const int32_t* __restrict__ fb_flag, const double* __restrict__ pose_fb,
                                                       const uint8_t* __restrict__ outl_fb, const int32_t* __restrict__ nm_bow,
                                                       const int32_t* __restrict__ ninl_fb, double* __restrict__ pose_cw,
                                                       int32_t* __restrict__ match_last, int32_t* __restrict__ match_kf, uint8_t* __restrict__ outlier,
                                                       int32_t* __restrict__ drop_kf, int32_t* __restrict__ counts, int32_t* __restrict__ counts2) {
  __shared__ int s_cnt[4];
  const int b = blockIdx.x;
  if (b >= B || fb_flag[b] == 0) return;
  int kept = 0;
  for (int i = threadIdx.x; i < NF; i += 256) {
    const size_t g = (size_t)b * NF + i;
    int q = match_kf[g], d = -1;
    if (q >= 0 && outl_fb[g]) {
      d = q;
      q = -1;
    }
    match_kf[g] = q;
    match_last[g] = -1;
    outlier[g] = 0;
    drop_kf[g] = d;
    kept += q >= 0;  // (a key-frame's map point is observed by that key-frame: countObservations() > 0, :327)
  }
  for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x < 7) pose_cw[(size_t)b * 7 + threadIdx.x] = pose_fb[(size_t)b * 7 + threadIdx.x];
  if (threadIdx.x == 0) {
    const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    counts[(size_t)b * 4 + 1] = ninl_fb[b];
    counts2[(size_t)b * 4 + 1] = nm_bow[b];
    counts2[(size_t)b * 4 + 2] = n;
    counts2[(size_t)b * 4 + 3] = n < 10 ? 2 : 1;
  }
}

// before searchLocalPoints (tracking.cpp:213-243): every map point the frame holds or has dropped as an outlier has been seen
// (last_visible_idx_ == idx: no candidate, :243 - the outliers of BOTH optimisations of a frame that went through the fallback); a feature is taken if its map point has observations (orb_matcher.cpp:74-76: a
// TEMPORAL point - createTemporalPoints, tracking.cpp:44-46: no observation - stays replaceable).  One workgroup per frame.
__global__ __launch_bounds__(256) void k_chain_before_local(int B, int NF, int NL, int NP, int NK, const int32_t* __restrict__ match_last,
                                                           const int32_t* __restrict__ match_kf, const int32_t* __restrict__ drop_src,
                                                           const int32_t* __restrict__ drop_kf, const int32_t* __restrict__ last_to_local,
                                                           const int32_t* __restrict__ kf_to_local, const uint8_t* __restrict__ last_observed,
                                                           const uint8_t* __restrict__ feat_taken0, const uint8_t* __restrict__ mp_cand0,
                                                           uint8_t* __restrict__ taken, uint8_t* __restrict__ cand) {
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int m = threadIdx.x; m < NP; m += 256) cand[(size_t)b * NP + m] = mp_cand0[(size_t)b * NP + m];
  __syncthreads();
  for (int i = threadIdx.x; i < NF; i += 256) {
    const size_t g = (size_t)b * NF + i;
    const int j = match_last[g], q = match_kf ? match_kf[g] : -1, d = drop_src ? drop_src[g] : -1, dk = drop_kf ? drop_kf[g] : -1;
    bool tk = feat_taken0[g] != 0;
    if (j >= 0) {
      const int l = last_to_local[(size_t)b * NL + j];
      if (l >= 0 && l < NP) cand[(size_t)b * NP + l] = 0;
      tk = tk || !last_observed || last_observed[(size_t)b * NL + j] != 0;
    }
    if (q >= 0) {
      const int l = kf_to_local ? kf_to_local[(size_t)b * NK + q] : -1;
      if (l >= 0 && l < NP) cand[(size_t)b * NP + l] = 0;
      tk = true;
    }
    if (d >= 0) {
      const int l = last_to_local[(size_t)b * NL + d];
      if (l >= 0 && l < NP) cand[(size_t)b * NP + l] = 0;
    }
    if (dk >= 0) {
      const int l = kf_to_local ? kf_to_local[(size_t)b * NK + dk] : -1;
      if (l >= 0 && l < NP) cand[(size_t)b * NP + l] = 0;
    }
    taken[g] = tk ? 1 : 0;
  }
}

// T_w_c.translation() of the refined pose (Frame::setTcw; SE3Quat::inverse: r = conj(q), t = r * (-t)), and optional copies
__global__ void k_chain_twc(int B, const double* __restrict__ pose, double* __restrict__ t_wc, double* __restrict__ pose_copy) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* p = pose + (size_t)b * 7;
  const double qi[4] = {-p[0], -p[1], -p[2], p[3]};
  const double mt[3] = {p[4] * -1., p[5] * -1., p[6] * -1.};
  double o[3];
  quat_rot_c(qi, mt, o);
  if (t_wc) {
    t_wc[(size_t)b * 3] = o[0];
    t_wc[(size_t)b * 3 + 1] = o[1];
    t_wc[(size_t)b * 3 + 2] = o[2];
  }
  if (pose_copy) {
    for (int i = 0; i < 7; ++i) pose_copy[(size_t)b * 7 + i] = p[i];
  }
}

__global__ void k_chain_counts(int B, const int32_t* __restrict__ src, int32_t* __restrict__ counts, int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) counts[(size_t)b * 4 + slot] = src[b];
}

// the chain's intermediates in the context's third scratch block
struct ChainScratch {
  double *Xw, *obs, *t_wc, *pose_fb;
  int32_t *oct, *nm, *ninl, *fb_flag, *nm_bow, *drop_src, *drop_kf;
  uint8_t *taken, *cand, *outl_fb;
};
int chain_scratch(gl::Ctx* c, int B, int NF, int NP, ChainScratch* S) {
  const size_t nf = (size_t)B * NF, np = (size_t)B * NP;
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  void* scratch = nullptr;
  const int rc = gl::ctx_scratch_c(c, 2 * up(nf * 24) + 3 * up(nf * 4) + 2 * up(nf) + up(np) + up((size_t)B * 24) + up((size_t)B * 56) + 4 * up((size_t)B * 4), &scratch);
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
  return GL_OK;
}
// optimizeCurrentPose of the chain: the problem of the frame's current associations (match_local / match_kf may be null), outlier flags in
// feature order, the inlier count to counts[., slot]
int chain_optimise(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int NF, int NL, int NP, int NK, const gl_track_chain_io* io,
                   const ChainScratch& S, double* pose, const int32_t* match_local, const int32_t* match_kf, int finalise, int count_slot) {
  gl::Ctx* c = gl::C(ctx);
  const size_t nf = (size_t)B * NF;
  const unsigned gf = (unsigned)((nf + 255) / 256), gb = (unsigned)((B + 63) / 64);
  // (gl_optimize_current_pose compacts a problem of more than 1 024 slots itself - option pose_compact: the frame's 1 200 feature slots
  // hold a few hundred edges)
  k_chain_pose_inputs<<<gf, 256, 0, c->stream>>>(B, NF, NL, NP, NK, io->feat_uv, io->feat_ur, io->feat_oct, io->match_last, io->last_pt, match_local, io->mp_pos,
                                                match_kf, io->kf_pt, S.Xw, S.obs, S.oct, finalise);
  GL_HIP(hipGetLastError());
  const int rc = gl_optimize_current_pose(ctx, cam, prm, B, NF, pose, S.Xw, S.obs, S.oct, io->outlier, S.ninl);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, S.ninl, io->counts, count_slot);
  GL_HIP(hipGetLastError());
  return GL_OK;
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

// stages 1 - 2 (+ the fallback): trackWithMotionModel, trackKeyFrame
int chain_front(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP, const gl_track_chain_io* io,
                float th_mm, int mono, const ChainScratch& S, int32_t* drop_src, int32_t* drop_kf) {
  gl::Ctx* c = gl::C(ctx);
  const size_t nf = (size_t)B * NF;
  const unsigned gf = (unsigned)((nf + 255) / 256), gb = (unsigned)((B + 63) / 64);
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
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, S.nm, io->counts, 0);
  // ---- stage 2: optimizeCurrentPose on the matched features, outliers dropped (:348-371)
  GL_HIP(hipMemsetAsync(io->outlier, 0, nf, c->stream));
  rc = chain_optimise(ctx, cam, prm, B, NF, NL, NP, NK, io, S, io->pose_cw, nullptr, nullptr, 0, 1);
  if (rc != GL_OK) return rc;
  k_chain_after_mm<<<B, 256, 0, c->stream>>>(B, NF, NL, io->match_last, io->outlier, io->last_observed, S.nm, drop_src, fbk ? io->match_kf : nullptr, drop_kf, io->counts2,
                                             S.fb_flag, fbk ? 1 : 0);
  GL_HIP(hipGetLastError());
  if (fbk) {
    // ---- Tracking::trackKeyFrame (:297-331) for the flagged frames: ORBmatcher(0.7, true).searchByBoW(ref_keyframe_, curr), the last
    // frame's pose, optimizeCurrentPose, outliers dropped.  The workgroups of the other frames return at once.
    rc = gl::launch_bow_gated(ctx, 0.7f, 1, B, io->NK, NF, io->NNK, io->NNF, io->kf_angle, io->kf_desc, io->kf_has_mp, io->kf_nnode, io->kf_node_id,
                              io->kf_node_ptr, io->kf_node_idx, io->feat_angle, io->feat_desc, io->feat_nnode, io->feat_node_id, io->feat_node_ptr,
                              io->feat_node_idx, io->match_kf, S.nm_bow, S.fb_flag);
    if (rc != GL_OK) return rc;
    GL_HIP(hipMemsetAsync(S.outl_fb, 0, nf, c->stream));
    k_chain_fb_inputs<<<gf, 256, 0, c->stream>>>(B, NF, io->NK, S.fb_flag, io->feat_uv, io->feat_ur, io->feat_oct, io->match_kf, io->kf_pt, io->pose_lw, S.pose_fb,
                                                S.Xw, S.obs, S.oct);
    GL_HIP(hipGetLastError());
    rc = gl_optimize_current_pose(ctx, cam, prm, B, NF, S.pose_fb, S.Xw, S.obs, S.oct, S.outl_fb, S.ninl);
    if (rc != GL_OK) return rc;
    k_chain_after_fb<<<B, 256, 0, c->stream>>>(B, NF, S.fb_flag, S.pose_fb, S.outl_fb, S.nm_bow, S.ninl, io->pose_cw, io->match_last, io->match_kf, io->outlier,
                                               drop_kf, io->counts, io->counts2);
    GL_HIP(hipGetLastError());
  }
  if (io->pose_mm) k_chain_twc<<<gb, 64, 0, c->stream>>>(B, io->pose_cw, nullptr, io->pose_mm);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// stages 3 - 4: searchLocalPoints, trackLocalMap
int chain_back(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP, const gl_track_chain_io* io,
               float th_local, float nn_ratio, const ChainScratch& S, const int32_t* drop_src, const int32_t* drop_kf) {
  gl::Ctx* c = gl::C(ctx);
  const unsigned gb = (unsigned)((B + 63) / 64);
  const bool fbk = chain_has_fallback(io);
  const int NK = fbk ? io->NK : 1;
  int rc;
  k_chain_before_local<<<B, 256, 0, c->stream>>>(B, NF, NL, NP, NK, io->match_last, fbk ? io->match_kf : nullptr, drop_src, fbk ? drop_kf : nullptr,
                                                 io->last_to_local, fbk ? io->kf_to_local : nullptr, io->last_observed, io->feat_taken, io->mp_cand, S.taken,
                                                 S.cand);
  k_chain_twc<<<gb, 64, 0, c->stream>>>(B, io->pose_cw, S.t_wc, nullptr);
  GL_HIP(hipGetLastError());
  // ---- stage 3: searchLocalPoints from the refined pose (:210-270)
  rc = gl_search_local_points(ctx, cam, scale_factor, B, NF, NP, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_desc, S.taken, io->pose_cw, S.t_wc, io->mp_pos,
                              io->mp_normal, io->mp_max_dist, io->mp_min_dist, S.cand, io->mp_desc, th_local, nn_ratio, io->match_local, S.nm, io->inview);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, S.nm, io->counts, 2);
  // ---- stage 4: trackLocalMap's optimizeCurrentPose on every feature with a map point (:272-299)
  return chain_optimise(ctx, cam, prm, B, NF, NL, NP, NK, io, S, io->pose_cw, io->match_local, fbk ? io->match_kf : nullptr, 1, 3);
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
  rc = chain_front(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_mm, mono, S, drop, dropk);
  if (rc != GL_OK) return rc;
  return chain_back(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_local, nn_ratio, S, drop, dropk);
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
  return chain_front(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_mm, mono, S, io->drop_src, io->drop_kf ? io->drop_kf : S.drop_kf);
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
  return chain_back(ctx, cam, prm, scale_factor, B, NF, NL, NP, io, th_local, nn_ratio, S, io->drop_src, io->drop_kf);
}
