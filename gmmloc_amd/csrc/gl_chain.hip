// gl_track_frame_chain: one tracked frame without host round trips ("Next" #6).
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
// its last-frame match (stage 1) or, failing that, its local-map match (stage 3).
__global__ __launch_bounds__(256) void k_chain_pose_inputs(int B, int NF, int NL, int NP, const double* __restrict__ feat_uv,
                                                          const float* __restrict__ feat_ur, const int32_t* __restrict__ feat_oct,
                                                          const int32_t* __restrict__ match_last, const double* __restrict__ last_pt,
                                                          const int32_t* __restrict__ match_local, const double* __restrict__ mp_pos,
                                                          double* __restrict__ Xw, double* __restrict__ obs, int32_t* __restrict__ oct) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)B * NF) return;
  const size_t b = g / NF;
  const int j = match_last[g], k = match_local ? match_local[g] : -1;
  const double* X = j >= 0 ? last_pt + (b * NL + j) * 3 : (k >= 0 ? mp_pos + (b * NP + k) * 3 : nullptr);
  Xw[g * 3] = X ? X[0] : 0.0;
  Xw[g * 3 + 1] = X ? X[1] : 0.0;
  Xw[g * 3 + 2] = X ? X[2] : 0.0;
  obs[g * 3] = feat_uv[g * 2];
  obs[g * 3 + 1] = feat_uv[g * 2 + 1];
  obs[g * 3 + 2] = (double)feat_ur[g];
  oct[g] = X ? feat_oct[g] : -1;
}

// after the first optimisation (tracking.cpp:360-371): every map point matched in stage 1 has been seen by this frame
// (last_visible_idx_ = idx: no candidate of searchLocalPoints, :243) - inliers and outliers alike; an outlier's feature loses its map
// point and its flag.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_chain_after_mm(int B, int NF, int NL, int NP, int32_t* __restrict__ match_last,
                                                       uint8_t* __restrict__ outlier, const int32_t* __restrict__ last_to_local,
                                                       const uint8_t* __restrict__ feat_taken0, const uint8_t* __restrict__ mp_cand0,
                                                       uint8_t* __restrict__ taken, uint8_t* __restrict__ cand) {
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int m = threadIdx.x; m < NP; m += 256) cand[(size_t)b * NP + m] = mp_cand0[(size_t)b * NP + m];
  __syncthreads();
  for (int i = threadIdx.x; i < NF; i += 256) {
    const size_t g = (size_t)b * NF + i;
    int j = match_last[g];
    if (j >= 0) {
      const int l = last_to_local[(size_t)b * NL + j];
      if (l >= 0 && l < NP) cand[(size_t)b * NP + l] = 0;
      if (outlier[g]) {
        j = -1;
        match_last[g] = -1;
        outlier[g] = 0;
      }
    }
    taken[g] = (feat_taken0[g] || j >= 0) ? 1 : 0;
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
  t_wc[(size_t)b * 3] = o[0];
  t_wc[(size_t)b * 3 + 1] = o[1];
  t_wc[(size_t)b * 3 + 2] = o[2];
  if (pose_copy) {
    for (int i = 0; i < 7; ++i) pose_copy[(size_t)b * 7 + i] = p[i];
  }
}

__global__ void k_chain_counts(int B, const int32_t* __restrict__ src, int32_t* __restrict__ counts, int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) counts[(size_t)b * 4 + slot] = src[b];
}

}  // namespace

extern "C" int gl_track_frame_chain(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, float scale_factor, int B, int NF, int NL, int NP,
                                    const gl_track_chain_io* io, float th_mm, float th_local, float nn_ratio, int mono) {
  GL_REQUIRE(ctx && cam && prm && io, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NF >= 1 && NL >= 1 && NP >= 1, "bad B / NF / NL / NP");
  GL_REQUIRE(io->feat_uv && io->feat_ur && io->feat_oct && io->feat_angle && io->feat_desc && io->feat_taken && io->pose_lw && io->last_pt &&
                 io->last_valid && io->last_oct && io->last_angle && io->last_desc && io->last_to_local && io->mp_pos && io->mp_normal &&
                 io->mp_max_dist && io->mp_min_dist && io->mp_cand && io->mp_desc && io->pose_cw && io->match_last && io->match_local &&
                 io->outlier && io->counts,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  const size_t nf = (size_t)B * NF, np = (size_t)B * NP;
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  void* scratch = nullptr;
  {
    const int rc = gl::ctx_scratch_c(c, 2 * up(nf * 24) + up(nf * 4) + up(nf) + up(np) + up((size_t)B * 24) + 2 * up((size_t)B * 4), &scratch);
    if (rc != GL_OK) return rc;
  }
  char* s = (char*)scratch;
  double* Xw = (double*)s;
  s += up(nf * 24);
  double* obs = (double*)s;
  s += up(nf * 24);
  int32_t* oct = (int32_t*)s;
  s += up(nf * 4);
  uint8_t* taken = (uint8_t*)s;
  s += up(nf);
  uint8_t* cand = (uint8_t*)s;
  s += up(np);
  double* t_wc = (double*)s;
  s += up((size_t)B * 24);
  int32_t* nm = (int32_t*)s;
  s += up((size_t)B * 4);
  int32_t* ninl = (int32_t*)s;
  const unsigned gf = (unsigned)((nf + 255) / 256), gb = (unsigned)((B + 63) / 64);
  int rc;
  // ---- stage 1: ORBmatcher(0.9, true).searchByProjection(curr, last, th), again with 2 th where fewer than 20 matches (tracking.cpp:330-342)
  rc = gl_search_by_projection_frame(ctx, cam, scale_factor, B, NF, NL, io->pose_cw, io->pose_lw, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_angle,
                                     io->feat_desc, io->feat_taken, io->last_pt, io->last_valid, io->last_oct, io->last_angle, io->last_desc, th_mm, mono, 1,
                                     io->match_last, nm);
  if (rc != GL_OK) return rc;
  rc = gl::launch_match_frame_gated(ctx, cam, scale_factor, B, NF, NL, io->pose_cw, io->pose_lw, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_angle,
                                    io->feat_desc, io->feat_taken, io->last_pt, io->last_valid, io->last_oct, io->last_angle, io->last_desc, 2 * th_mm, mono, 1,
                                    io->match_last, nm, nm, 20);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, nm, io->counts, 0);
  // ---- stage 2: optimizeCurrentPose on the matched features, outliers dropped (:348-371)
  GL_HIP(hipMemsetAsync(io->outlier, 0, nf, c->stream));
  k_chain_pose_inputs<<<gf, 256, 0, c->stream>>>(B, NF, NL, NP, io->feat_uv, io->feat_ur, io->feat_oct, io->match_last, io->last_pt, nullptr, nullptr, Xw, obs, oct);
  GL_HIP(hipGetLastError());
  rc = gl_optimize_current_pose(ctx, cam, prm, B, NF, io->pose_cw, Xw, obs, oct, io->outlier, ninl);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, ninl, io->counts, 1);
  k_chain_after_mm<<<B, 256, 0, c->stream>>>(B, NF, NL, NP, io->match_last, io->outlier, io->last_to_local, io->feat_taken, io->mp_cand, taken, cand);
  k_chain_twc<<<gb, 64, 0, c->stream>>>(B, io->pose_cw, t_wc, io->pose_mm);
  GL_HIP(hipGetLastError());
  // ---- stage 3: searchLocalPoints from the refined pose (:210-270)
  rc = gl_search_local_points(ctx, cam, scale_factor, B, NF, NP, io->feat_uv, io->feat_ur, io->feat_oct, io->feat_desc, taken, io->pose_cw, t_wc, io->mp_pos,
                              io->mp_normal, io->mp_max_dist, io->mp_min_dist, cand, io->mp_desc, th_local, nn_ratio, io->match_local, nm, io->inview);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, nm, io->counts, 2);
  // ---- stage 4: trackLocalMap's optimizeCurrentPose on every feature with a map point (:272-299)
  k_chain_pose_inputs<<<gf, 256, 0, c->stream>>>(B, NF, NL, NP, io->feat_uv, io->feat_ur, io->feat_oct, io->match_last, io->last_pt, io->match_local, io->mp_pos, Xw, obs,
                                                oct);
  GL_HIP(hipGetLastError());
  rc = gl_optimize_current_pose(ctx, cam, prm, B, NF, io->pose_cw, Xw, obs, oct, io->outlier, ninl);
  if (rc != GL_OK) return rc;
  k_chain_counts<<<gb, 64, 0, c->stream>>>(B, ninl, io->counts, 3);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
