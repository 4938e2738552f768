// The projection / visibility loop in front of the matchers - Tracking::searchLocalPoints (tracking.cpp:233-256) and
// Localization::fuseObservations (localization.cpp:242-254): per map point Frame::project3 (frame.cpp:98-119; g2o SE3Quat::map,
// PinholeCamera::project3 + evaluateProjectionResult, pinhole_camera.cpp:46-66, 128-150) and MapPoint::checkScaleAndVisible
// (mappoint.cpp:257-303).  Output = the ProjStat array the matchers take (uvr, scale_pred, view_cos, dist) and the in-view flag, so
// that gl_search_by_projection / gl_fuse_search run on it without a trip through the host.  One thread per (frame, map point).
// Every float / double conversion of the reference is kept (compiled without contraction).  The scale prediction
// ceil(log(ratio) / log(1.2f)) is a FLOAT logarithm of the host's libm in the reference (std::log(float): mappoint.cpp has using
// namespace std), which is not correctly rounded (0.7 % of the arguments differ from the rounded double logarithm in the last bit), so
// no device logarithm can be trusted to give its level at the level boundaries.  The level is a monotone step function of ratio, so
// the HOST finds its seven steps once per scale factor - bisection over the float bit patterns with the host's own logf, and a check
// that the function is monotone around each step - and the device only compares ratio against them: the levels of the host's libm.
#include "gl_internal.hpp"

#include <cmath>
#include <cstring>
#include <mutex>

namespace {

struct ProjP {
  double fx, fy, cx, cy;
  float mbf;
  float step[7];  // step[L] = the largest ratio with level <= L
  int width, height, NP;
};

__device__ __forceinline__ void quat_rot(const double* q, const double* v, double* o) {  // Eigen Quaternion * Vector3
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  o[0] = v[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]);
  o[1] = v[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]);
  o[2] = v[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]);
}

__global__ __launch_bounds__(256) void k_project_map_points(ProjP P, int B, const double* __restrict__ pose_cw_all, const double* __restrict__ t_wc_all,
                                                            const double* __restrict__ pos_all, const double* __restrict__ normal_all,
                                                            const float* __restrict__ max_dist_all, const float* __restrict__ min_dist_all,
                                                            const uint8_t* __restrict__ cand_all, double* __restrict__ uvr_all,
                                                            int32_t* __restrict__ level_all, double* __restrict__ viewcos_all,
                                                            double* __restrict__ dist_all, uint8_t* __restrict__ inview_all) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long)B * P.NP) return;
  const int f = (int)(g / P.NP);
  uint8_t in_view = 0;
  int lvl = 0;
  double uvr[3] = {0.0, 0.0, 0.0}, vc = 0.0, dd = 0.0;
  if (cand_all[g]) {
    const double* pc = pose_cw_all + (size_t)f * 7;
    const double* pos = pos_all + g * 3;
    double ptc[3];
    quat_rot(pc, pos, ptc);  // getTcw().map(pt) = _r * xyz + _t
    ptc[0] += pc[4];
    ptc[1] += pc[5];
    ptc[2] += pc[6];
    bool ok = !(ptc[2] < 0.0);
    if (ok) {
      const double rz = 1.0 / ptc[2];
      const double kx = ptc[0] * rz, ky = ptc[1] * rz;
      const double u = P.fx * kx + P.cx, v = P.fy * ky + P.cy;
      const bool visible = u >= 0.0 && v >= 0.0 && u < (double)P.width && v < (double)P.height;
      ok = visible && ptc[2] > 0.0;  // KEYPOINT_VISIBLE (kMinimumDepth = 0)
      if (ok) {
        uvr[0] = u;
        uvr[1] = v;
        uvr[2] = u - (double)P.mbf / ptc[2];
      }
    }
    if (ok) {  // checkScaleAndVisible
      const float max_dist = 1.2f * max_dist_all[g], min_dist = 0.8f * min_dist_all[g];
      const double* tw = t_wc_all + (size_t)f * 3;
      const double vx = pos[0] - tw[0], vy = pos[1] - tw[1], vz = pos[2] - tw[2];
      const float dist = (float)sqrt(vx * vx + vy * vy + vz * vz);
      ok = !(dist < min_dist || dist > max_dist);
      if (ok) {
        const double* nrm = normal_all + g * 3;
        const float view_cos = (float)((vx * nrm[0] + vy * nrm[1] + vz * nrm[2]) / (double)dist);
        ok = !(view_cos < 0.5f);
        if (ok) {
          const float ratio = max_dist_all[g] / dist;
          int ls = 0;  // clamp(ceil(log(ratio) / scale_factor_log), 0, frame::num_levels - 1)
#pragma unroll
          for (int L = 0; L < 7; ++L) ls += ratio > P.step[L] ? 1 : 0;
          if (!(ratio <= 3.402823466e38f)) ls = 0;  // inf / NaN (dist == 0): the int conversion of the reference gives INT_MIN on x86 -> 0
          lvl = ls;
          vc = (double)view_cos;
          dd = (double)dist;
          in_view = 1;
        }
      }
    }
  }
  // (proj_stat.uvr is set for the points in view only, tracking.cpp:247-249: the others read 0)
  uvr_all[g * 3] = in_view ? uvr[0] : 0.0;
  uvr_all[g * 3 + 1] = in_view ? uvr[1] : 0.0;
  uvr_all[g * 3 + 2] = in_view ? uvr[2] : 0.0;
  level_all[g] = lvl;
  viewcos_all[g] = vc;
  dist_all[g] = dd;
  inview_all[g] = in_view;
}

// the unclamped level of MapPoint::checkScaleAndVisible with the host's libm
inline int host_level(float ratio, float sfl) { return (int)std::ceil(std::log(ratio) / sfl); }

// step[L] = the largest float with host_level <= L (L = 0 .. 6); false if host_level is not monotone around a step
bool level_steps(float sfl, float* step) {
  auto f2u = [](float x) { uint32_t u; memcpy(&u, &x, 4); return u; };
  auto u2f = [](uint32_t u) { float x; memcpy(&x, &u, 4); return x; };
  for (int L = 0; L < 7; ++L) {
    uint32_t lo = f2u(0.5f), hi = f2u(64.0f);  // level(0.5) <= 0 < 7 < level(64) for every scale factor in (1, 1.7]
    if (host_level(u2f(lo), sfl) > L || host_level(u2f(hi), sfl) <= L) return false;
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (host_level(u2f(mid), sfl) <= L) lo = mid;
      else hi = mid;
    }
    for (uint32_t d = 1; d <= 4096; ++d)
      if (host_level(u2f(lo - d), sfl) > L || host_level(u2f(hi + d), sfl) <= L) return false;
    step[L] = u2f(lo);
  }
  return true;
}

}  // namespace

extern "C" int gl_level_steps(float scale_factor, float* step7) {
  GL_REQUIRE(step7, "null argument");
  GL_REQUIRE(scale_factor > 1.0f && scale_factor <= 1.7f, "scale factor outside (1, 1.7]");
  static std::mutex mu;
  static float cached_sf = 0.0f, cached_step[7];
  std::lock_guard<std::mutex> g(mu);
  if (cached_sf != scale_factor) {
    const float sfl = std::log(scale_factor);  // config.cpp:57
    GL_REQUIRE(level_steps(sfl, cached_step), "the host's logf is not monotone around a level step");
    cached_sf = scale_factor;
  }
  memcpy(step7, cached_step, sizeof(cached_step));
  return GL_OK;
}

extern "C" int gl_project_map_points(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NP, const double* pose_cw_dev,
                                     const double* t_wc_dev, const double* pos_dev, const double* normal_dev, const float* max_dist_dev,
                                     const float* min_dist_dev, const uint8_t* cand_dev, double* uvr_dev, int32_t* level_dev,
                                     double* viewcos_dev, double* dist_dev, uint8_t* inview_dev) {
  GL_REQUIRE(ctx && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NP >= 1, "bad B / NP");
  GL_REQUIRE(cam->width > 0 && cam->height > 0, "camera without image size");
  GL_REQUIRE(pose_cw_dev && t_wc_dev && pos_dev && normal_dev && max_dist_dev && min_dist_dev && cand_dev && uvr_dev && level_dev &&
                 viewcos_dev && dist_dev && inview_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  ProjP P;
  // camera::fx ... are float config scalars (config.h:38-48); the PinholeCamera holds them as double intrinsics
  P.fx = (float)cam->fx;
  P.fy = (float)cam->fy;
  P.cx = (float)cam->cx;
  P.cy = (float)cam->cy;
  P.mbf = (float)cam->bf;                  // frame.cpp:23
  {
    const int rc = gl_level_steps(scale_factor, P.step);
    if (rc != GL_OK) return rc;
  }
  P.width = cam->width;
  P.height = cam->height;
  P.NP = NP;
  const long n = (long)B * NP;
  k_project_map_points<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(P, B, pose_cw_dev, t_wc_dev, pos_dev, normal_dev, max_dist_dev,
                                                                          min_dist_dev, cand_dev, uvr_dev, level_dev, viewcos_dev, dist_dev,
                                                                          inview_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// Tracking::searchLocalPoints, the device part in one call (tracking.cpp:233-270): the projection / visibility loop above, then
// ORBmatcher(0.8).searchByProjection(curr_frame_, local_mappoints_, mappoints_proj_stat, th) (orb_matcher.cpp:27-110) on its
// outputs, which stay in the context's scratch.  inview_dev (optional): is_in_view_ per map point, for the host's num_visible_++.
extern "C" int gl_search_local_points(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP, const double* feat_uv_dev,
                                      const float* feat_ur_dev, const int32_t* feat_oct_dev, const uint8_t* feat_desc_dev,
                                      const uint8_t* feat_taken_dev, const double* pose_cw_dev, const double* t_wc_dev, const double* mp_pos_dev,
                                      const double* mp_normal_dev, const float* mp_max_dist_dev, const float* mp_min_dist_dev,
                                      const uint8_t* mp_cand_dev, const uint8_t* mp_desc_dev, float th, float nn_ratio, int32_t* feat_match_dev,
                                      int32_t* nmatches_dev, uint8_t* inview_dev) {
  GL_REQUIRE(ctx && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NP >= 1 && NF >= 1, "bad B / NF / NP");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  const size_t n = (size_t)B * NP;
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  void* scratch = nullptr;
  const int rc0 = gl::ctx_scratch(c, up(n * 24) + 2 * up(n * 8) + up(n * 4) + up(n), &scratch);
  if (rc0 != GL_OK) return rc0;
  char* s = (char*)scratch;
  double* uvr = (double*)s;
  s += up(n * 24);
  double* viewcos = (double*)s;
  s += up(n * 8);
  double* dist = (double*)s;
  s += up(n * 8);
  int32_t* level = (int32_t*)s;
  s += up(n * 4);
  uint8_t* inview = inview_dev ? inview_dev : (uint8_t*)s;
  int rc = gl_project_map_points(ctx, cam, scale_factor, B, NP, pose_cw_dev, t_wc_dev, mp_pos_dev, mp_normal_dev, mp_max_dist_dev, mp_min_dist_dev,
                                 mp_cand_dev, uvr, level, viewcos, dist, inview);
  if (rc != GL_OK) return rc;
  return gl_search_by_projection(ctx, cam, scale_factor, B, NF, NP, feat_uv_dev, feat_ur_dev, feat_oct_dev, feat_desc_dev, feat_taken_dev, uvr, level,
                                 viewcos, inview, mp_desc_dev, th, nn_ratio, feat_match_dev, nmatches_dev);
}
