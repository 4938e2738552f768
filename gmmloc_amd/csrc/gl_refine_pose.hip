// Tracking::optimizeCurrentPose (tracking_opt.cpp:21-217): 6-DoF Levenberg-Marquardt over pose-only
// reprojection edges (g2o EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose, Huber), 4 gating rounds
// x optimize(10), one persistent group of NW waves per frame, entirely on-chip:
//   * thread t of the group owns the edges t, t + 64 NW, ...; one pass computes the residual, chi2, Huber
//     weight, the 2|3 x 6 Jacobian and accumulates the 21 unique entries of J^T W J, the 6 of b and the
//     robust chi2 in registers;
//   * a wave reduce-scatter (permlane swaps + DPP) + one LDS step give the 28 sums; the current and the trial
//     system live in LDS (224 B each), not in registers;
//   * every wave solves the 6x6 (LDL^T on the packed triangle, reciprocal pivots) and applies exp(delta) on
//     its own - pose as (R, t) in SGPRs, Rodrigues with a short series below |theta| = 0.01 - so nothing is
//     broadcast;
//   * the evaluation at the trial pose also builds the next system: an accepted LM step costs ONE pass.
// NW = 1 (no barrier at all, GL_POSE_WPS frames per SIMD) is the shape for large batches, NW = 4 / 8 the one
// for the frame-at-a-time caller: 0.34 ms instead of 0.90 ms for one frame of 1 000 edges.
// g2o control flow restated: OptimizationAlgorithmLevenberg::solve (lambda init 1e-5 max diag, rho test with
// +1e-3, x1/3..2/3 / x nu schedule, 10 trials), SparseOptimizer::optimize, levels via
// initializeOptimization(0), stale per-edge errors read by e->chi2() after the last trial (SURVEY.md
// Appendix A).
#include <algorithm>
#include <cstdlib>

#include "gl_device.hpp"
#include "gl_internal.hpp"

#pragma clang fp contract(fast)

using namespace gld;

namespace {

struct PoseKParams {
  double fx, fy, cx, cy, bf;
  double s2inv[8];
  double delta_mono, delta_stereo;
};

// un-robustified chi2 of one edge at pose (R,t) (e->computeError(); e->chi2())
GL_DEV double pose_edge_chi2(const PoseKParams& kp, const double* R, const double* t, const double* Xw,
                             const double* obs, int oc) {
  const double x = R[0] * Xw[0] + R[1] * Xw[1] + R[2] * Xw[2] + t[0];
  const double y = R[3] * Xw[0] + R[4] * Xw[1] + R[5] * Xw[2] + t[1];
  const double z = R[6] * Xw[0] + R[7] * Xw[1] + R[8] * Xw[2] + t[2];
  const double invz = 1.0 / z;
  const double pu = x * invz * kp.fx + kp.cx, pv = y * invz * kp.fy + kp.cy;
  const bool stereo = !(obs[2] < 0);
  const double e0 = obs[0] - pu, e1 = obs[1] - pv;
  const double e2 = stereo ? (obs[2] - (pu - kp.bf * invz)) : 0.0;
  const double s = kp.s2inv[oc];
  return e0 * (s * e0) + e1 * (s * e1) + e2 * (s * e2);
}

GL_DEV double rcp_nr(double a) {
  double x = __builtin_amdgcn_rcp(a);
  x = fma(fma(-a, x, 1.0), x, x);
  x = fma(fma(-a, x, 1.0), x, x);
  return x;
}
GL_DEV double uni(double v) {  // wave-uniform value -> SGPR pair
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
// the 28 totals of acc[] over the NW waves of the workgroup -> dst[0..27] in LDS.  Per wave a reduce-scatter
// (permlane swaps + DPP) leaves total s on lane wave_slot^-1(s); the owners park them in part[wave][], and lane
// s of every wave adds the NW partials in wave order (the same bits in every wave) - with one wave the barriers
// are no-ops that order the LDS write before the broadcast reads.
template <int NW>
GL_DEV void block_totals28(double* acc, double* part, double* dst, Coop& C) {
#pragma unroll
  for (int i = 28; i < 32; ++i) acc[i] = 0.0;
  const double r = wave_reduce_scatter32(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // dst / part may still be read by slower waves
  if (NW == 1) {
    if (wave_slot_owner(lane)) dst[wave_slot(lane)] = r;
  } else {
    if (wave_slot_owner(lane)) part[wave * 32 + wave_slot(lane)] = r;
    __syncthreads();
    if (wave == 0 && lane < 32) {
      double s = part[lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += part[w * 32 + lane];
      dst[lane] = s;
    }
  }
  __syncthreads();
  if (C.NB > 1) coop_totals<false>(C, dst);  // second level: the workgroups that share the frame
}
template <int NW>
GL_DEV double block_total1(double v, double* part, double* xs, Coop& C) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += shfl_xor_f64(v, o);
  if (NW == 1 && C.NB == 1) return uni(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = part[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) s += part[w];
  if (C.NB > 1) {
    __syncthreads();
    if (threadIdx.x < 32) xs[threadIdx.x] = threadIdx.x == 0 ? s : 0.0;
    __syncthreads();
    coop_totals<false>(C, xs);
    s = xs[0];
  }
  return uni(s);
}

struct PoseRt {
  double R[9], t[3];
};
GL_DEV PoseRt rt_uni(const PoseRt& P) {
  PoseRt U;
#pragma unroll
  for (int i = 0; i < 9; ++i) U.R[i] = uni(P.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) U.t[i] = uni(P.t[i]);
  return U;
}
// exp(dx) * P   (SE3Quat::exp, VertexSE3Expmap::oplusImpl) on the rotation matrix
GL_DEV PoseRt rt_update(const PoseRt& P, const double* u) {
  const double th2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
  double a, b, c;
  if (th2 < 1e-4) {
    a = fma(fma(fma(-1.0 / 5040, th2, 1.0 / 120), th2, -1.0 / 6), th2, 1.0);
    b = fma(fma(fma(-1.0 / 40320, th2, 1.0 / 720), th2, -1.0 / 24), th2, 0.5);
    c = fma(fma(fma(-1.0 / 362880, th2, 1.0 / 5040), th2, -1.0 / 120), th2, 1.0 / 6);
  } else {
    const double theta = sqrt(th2);
    double st, ct;
    sincos(theta, &st, &ct);
    const double it = 1.0 / theta;
    a = st * it;
    b = (1 - ct) * it * it;
    c = (theta - st) * it * it * it;
  }
  double Om[9], Om2[9], dR[9], V[9];
  skew(u, Om);
  mm3(Om, Om, Om2);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    dR[i] = I + a * Om[i] + b * Om2[i];
    V[i] = I + b * Om[i] + c * Om2[i];
  }
  PoseRt N;
  mm3(dR, P.R, N.R);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    N.t[i] = dR[i * 3] * P.t[0] + dR[i * 3 + 1] * P.t[1] + dR[i * 3 + 2] * P.t[2] + V[i * 3] * u[3] + V[i * 3 + 1] * u[4] +
             V[i * 3 + 2] * u[5];
  return rt_uni(N);
}

// 6x6 LDL^T on the packed upper triangle (row-major i <= j, as accumulated), (H + lambda I) x = b;
// fails on a non-positive / non-finite pivot like ldlt_solve<6>(..., require_positive = true)
#define GL_PU(i, j) ((i) * 6 - (i) * ((i)-1) / 2 + ((j) - (i)))
GL_DEV bool ldlt6_packed_pos(const double* H, const double* b, double lambda, double* x) {
  double a[21], iD[6], y[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) a[i] = H[i];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = a[GL_PU(j, j)] + lambda;
#pragma unroll
    for (int kk = 0; kk < j; ++kk) d -= a[GL_PU(kk, j)] * a[GL_PU(kk, j)] * a[GL_PU(kk, kk)];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    a[GL_PU(j, j)] = d;
    iD[j] = rcp_nr(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = a[GL_PU(j, i)];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) s -= a[GL_PU(kk, i)] * a[GL_PU(kk, j)] * a[GL_PU(kk, kk)];
      a[GL_PU(j, i)] = s * iD[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int kk = 0; kk < i; ++kk) s -= a[GL_PU(kk, i)] * y[kk];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= iD[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int kk = i + 1; kk < 6; ++kk) s -= a[GL_PU(i, kk)] * x[kk];
    x[i] = s;
  }
  return ok;
}

// one pass over the lane's edges at pose P: acc[0..20] H upper triangle, acc[21..26] b, acc[27] robust chi2
template <int NW>
GL_DEV void wave_pose_eval(const PoseKParams& kp, const double* __restrict__ s2tab, const PoseRt& P, bool robust, int e0, int es, int M,
                           const double* __restrict__ Xw, const double* __restrict__ obs,
                           const int32_t* __restrict__ octave, const uint8_t* __restrict__ level,
                           double* __restrict__ chi2_e, double* acc) {
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int e = e0; e < M; e += es) {
    const int oc = octave[e];
    if (oc < 0 || level[e] != 0) continue;
    const double X = Xw[(size_t)e * 3 + 0], Y = Xw[(size_t)e * 3 + 1], Z = Xw[(size_t)e * 3 + 2];
    const double ou = obs[(size_t)e * 3 + 0], ov = obs[(size_t)e * 3 + 1], our = obs[(size_t)e * 3 + 2];
    const bool stereo = !(our < 0);
    const double x = P.R[0] * X + P.R[1] * Y + P.R[2] * Z + P.t[0];
    const double y = P.R[3] * X + P.R[4] * Y + P.R[5] * Z + P.t[1];
    const double z = P.R[6] * X + P.R[7] * Y + P.R[8] * Z + P.t[2];
    const double invz = rcp_nr(z), invz2 = invz * invz;
    const double pu = x * invz * kp.fx + kp.cx;
    const double pv = y * invz * kp.fy + kp.cy;
    const double e0 = ou - pu, e1 = ov - pv;
    const double e2 = stereo ? (our - (pu - kp.bf * invz)) : 0.0;
    const double s = s2tab[oc];
    const double chi2 = e0 * (s * e0) + e1 * (s * e1) + e2 * (s * e2);
    chi2_e[e] = chi2;
    double rho0 = chi2, rho1 = 1.0;
    if (robust) huber(chi2, stereo ? kp.delta_stereo : kp.delta_mono, rho0, rho1);
    const double w = rho1 * s;
    double J0[6], J1[6], J2[6];
    J0[0] = x * y * invz2 * kp.fx;
    J0[1] = -(1 + (x * x * invz2)) * kp.fx;
    J0[2] = y * invz * kp.fx;
    J0[3] = -invz * kp.fx;
    J0[4] = 0;
    J0[5] = x * invz2 * kp.fx;
    J1[0] = (1 + y * y * invz2) * kp.fy;
    J1[1] = -x * y * invz2 * kp.fy;
    J1[2] = -x * invz * kp.fy;
    J1[3] = 0;
    J1[4] = -invz * kp.fy;
    J1[5] = y * invz2 * kp.fy;
    const double sb = stereo ? 1.0 : 0.0;
    J2[0] = sb * (J0[0] - kp.bf * y * invz2);
    J2[1] = sb * (J0[1] + kp.bf * x * invz2);
    J2[2] = sb * J0[2];
    J2[3] = sb * J0[3];
    J2[4] = 0;
    J2[5] = sb * (J0[5] - kp.bf * invz2);
    int q = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double w0 = w * J0[i], w1 = w * J1[i], w2 = w * J2[i];
#pragma unroll
      for (int j = i; j < 6; ++j) {
        acc[q] = fma(w0, J0[j], fma(w1, J1[j], fma(w2, J2[j], acc[q])));
        ++q;
      }
      acc[21 + i] = fma(-w0, e0, fma(-w1, e1, fma(-w2, e2, acc[21 + i])));
    }
    acc[27] += rho0;
  }
}

#ifndef GL_POSE_WPS
#define GL_POSE_WPS 3  // waves per SIMD the register budget is capped for (measured: 3 > 2 > 4, tools/pose_ab.py)
#endif
// NW waves per frame: 1 for batches (GL_POSE_WPS frames per SIMD), 4 / 8 when the frames are fewer than the
// SIMDs - the edges are dealt round the NW x 64 threads, the 28 sums meet in LDS, and every wave repeats the
// serial part (solve, pose update) on its own so that no broadcast is needed.
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? GL_POSE_WPS : (NW + 3) / 4) void k_optimize_current_pose(PoseKParams kp, int B, int M,
                                                                 double* __restrict__ pose_io,
                                                                 const double* __restrict__ Xw_all,
                                                                 const double* __restrict__ obs_all,
                                                                 const int32_t* __restrict__ oct_all,
                                                                 uint8_t* __restrict__ outlier_all,
                                                                 int32_t* __restrict__ ninlier,
                                                                 double* __restrict__ chi2_all, int NB,
                                                                 unsigned long long* parts) {
  __shared__ double s2tab[8];
  __shared__ double H[32], Hn[32];  // current / trial system {H upper (21), b (6), chi2}
  __shared__ double part[NW * 32];  // per-wave partial sums
  __shared__ double xs[32];         // scalar exchanges between the workgroups of a frame
  // NB > 1 (cooperative launch, few frames): workgroup pb of the NB that share frame f; the edges are dealt
  // round all NB x NW x 64 threads and every sum has a second level across the workgroups (gld::coop_totals)
  const int f = blockIdx.x / NB, lane = threadIdx.x;  // "lane" = thread of the workgroup's NW waves
  if (f >= B) return;
  Coop C{parts ? parts + (size_t)f * 2 * NB * 64 : nullptr, NB, (int)(blockIdx.x % NB), 0u};
  const int e0 = C.pb * 64 * NW + lane, es = 64 * NW * NB;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s2tab[j] = kp.s2inv[j];
  }
  __syncthreads();  // single wave: orders the table write before the reads
  const double* Xw = Xw_all + (size_t)f * M * 3;
  const double* obs = obs_all + (size_t)f * M * 3;
  const int32_t* octave = oct_all + (size_t)f * M;
  uint8_t* level = outlier_all + (size_t)f * M;  // is_outlier_ <=> level 1
  double* chi2_e = chi2_all + (size_t)f * M;

  // graph construction: count edges, clear outlier flags (tracking_opt.cpp:60-137)
  double cnt = 0.0;
  for (int e = e0; e < M; e += es) {
    if (octave[e] >= 0) {  // is_outlier_[i] is reset only where mappoints_[i] exists (:63-69); the flags of the
      level[e] = 0;        // other features stay the caller's
      cnt += 1.0;
    }
  }
  const int n_init = (int)block_total1<NW>(cnt, part, xs, C);
  if (n_init < 3) {  // :139-140
    if (lane == 0 && C.pb == 0) ninlier[f] = 0;
    return;
  }
  PoseRt P0;
  {
    const SE3 T0 = se3_load(pose_io + (size_t)f * 7);
    qtoR(T0.r, P0.R);
    P0.t[0] = T0.t[0];
    P0.t[1] = T0.t[1];
    P0.t[2] = T0.t[2];
    P0 = rt_uni(P0);
  }
  PoseRt P = P0;
  double acc[32];
  bool robust = true;
  int nbad = 0;
#pragma unroll 1
  for (int round = 0; round < 4; ++round) {
    P = P0;  // vertex_se3->setEstimate(curr_frame_->getTcw())  (:152)
    cnt = 0.0;
    for (int e = e0; e < M; e += es)
      if (octave[e] >= 0 && level[e] == 0) cnt += 1.0;
    const int nactive = (int)block_total1<NW>(cnt, part, xs, C);
    if (nactive > 0) {  // optimize(10); returns -1 untouched when nothing is active
      wave_pose_eval<NW>(kp, s2tab, P, robust, e0, es, M, Xw, obs, octave, level, chi2_e, acc);
      block_totals28<NW>(acc, part, H, C);
      double currentChi = uni(H[27]);
      bool sys_valid = true;
      double lambda = 0.0, ni = 2.0;
#pragma unroll 1
      for (int it = 0; it < 10; ++it) {
        if (!sys_valid) {  // computeActiveErrors + buildSystem at the (restored) estimate
          wave_pose_eval<NW>(kp, s2tab, P, robust, e0, es, M, Xw, obs, octave, level, chi2_e, acc);
          block_totals28<NW>(acc, part, H, C);
          currentChi = uni(H[27]);
          sys_valid = true;
        }
        if (it == 0) {  // computeLambdaInit: tau * max |diag|
          double md = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) md = fmax(fabs(H[GL_PU(i, i)]), md);
          lambda = uni(1e-5 * md);
          ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
          double dx[6];
          const bool ok2 = ldlt6_packed_pos(H, H + 21, lambda, dx);
          PoseRt Pn = P;
          double tempChi;
          if (ok2) {
            Pn = rt_update(P, dx);
            wave_pose_eval<NW>(kp, s2tab, Pn, robust, e0, es, M, Xw, obs, octave, level, chi2_e, acc);
            block_totals28<NW>(acc, part, Hn, C);
            tempChi = uni(Hn[27]);
          } else {
            tempChi = 1.7976931348623157e308;
          }
          double scale = 0.0;
#pragma unroll
          for (int j = 0; j < 6; ++j) scale += dx[j] * (lambda * dx[j] + H[21 + j]);
          scale += 1e-3;
          rho = uni((currentChi - tempChi) / scale);
          if (rho > 0 && isfinite(tempChi)) {
            const double u = 2 * rho - 1;
            double alpha = 1. - u * u * u;
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
            P = Pn;
            __syncthreads();
            if (lane < 27) H[lane] = Hn[lane];
            __syncthreads();
          } else {
            lambda *= ni;
            ni *= 2;
            // estimate restored (pop); H, b stay; per-edge errors stay those of the rejected trial
            // until the next computeActiveErrors
            if (!(rho < 0)) sys_valid = false;
          }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;  // Terminate
      }
    }
    // gating (:156-203): outliers are re-evaluated at the current estimate, inliers use the error of
    // the last computeActiveErrors; chi2 compared as float.
    cnt = 0.0;
    for (int e = e0; e < M; e += es) {
      const int oc = octave[e];
      if (oc < 0) continue;
      double c2;
      if (level[e] != 0)
        c2 = pose_edge_chi2(kp, P.R, P.t, Xw + (size_t)e * 3, obs + (size_t)e * 3, oc);
      else
        c2 = chi2_e[e];
      const bool stereo = !(obs[(size_t)e * 3 + 2] < 0);
      const float thr = stereo ? 7.815f : 5.991f;
      const bool bad = (float)c2 > thr;
      level[e] = bad ? 1 : 0;
      if (bad) cnt += 1.0;
    }
    nbad = (int)block_total1<NW>(cnt, part, xs, C);
    if (round == 2) robust = false;  // e->setRobustKernel(0) at it == 2
    if (n_init < 10) break;          // optimizer.edges().size() < 10
  }
  if (lane == 0 && C.pb == 0) {
    SE3 T;
    T.r = qfromR(P.R);
    T.t[0] = P.t[0];
    T.t[1] = P.t[1];
    T.t[2] = P.t[2];
    normalize_rotation(T);
    se3_store(T, pose_io + (size_t)f * 7);
    ninlier[f] = n_init - nbad;
  }
}

}  // namespace

extern "C" int gl_optimize_current_pose(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M,
                                        double* pose_dev, const double* Xw_dev, const double* obs_dev,
                                        const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev) {
  GL_REQUIRE(ctx && cam && prm, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && M >= 0, "bad B / M");
  GL_REQUIRE(pose_dev && ninlier_dev && (M == 0 || (Xw_dev && obs_dev && octave_dev && outlier_dev)), "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  PoseKParams kp;
  kp.fx = cam->fx;
  kp.fy = cam->fy;
  kp.cx = cam->cx;
  kp.cy = cam->cy;
  kp.bf = cam->bf;
  for (int i = 0; i < 8; ++i) kp.s2inv[i] = (double)prm->sigma2_inv[i];
  kp.delta_mono = (double)(float)sqrt(5.991);    // const float delta_mono = sqrt(5.991)   (:57)
  kp.delta_stereo = (double)(float)sqrt(7.815);  // const float delta_stereo = sqrt(7.815) (:58)
  void* scratch = nullptr;
  const size_t chi_bytes = (((size_t)B * (M > 0 ? M : 1) * sizeof(double) + 63) / 64) * 64;
  int rc = gl::ctx_scratch(c, chi_bytes + (size_t)B * 4096, &scratch);  // + the exchange words of the latency shape
  if (rc != GL_OK) return rc;
  {
    gl::TimerScope ts(c, GL_TIMER_REFINE_POSE);
    // waves per frame: 4 is never slower than 1 up to ~2 000 frames (0.34 vs 0.90 ms for one frame of 1 000
    // edges, 1.26 vs 1.80 ms for 1 024), 8 is the best between 32 and 256 frames; one wave per frame is the shape
    // for large batches (12 frames per CU, no barriers).  Never more waves than the frame has 64-edge slices.
    // GMMLOC_POSE_WAVES=1|4|8 forces a shape.
    int nw = B > 1536 ? 1 : (B > 32 && B <= 256) ? 8 : 4;
    while (nw > 1 && nw * 64 > M + 63) nw = nw == 8 ? 4 : 1;
    if (c->opt.pose_waves > 0) nw = (int)c->opt.pose_waves == 8 ? 8 : (int)c->opt.pose_waves == 4 ? 4 : 1;
    // very few frames: a frame's edges are dealt to NB <= 4 workgroups of 4 waves on as many CUs (one edge per
    // thread from 1 024 edges), sums exchanged between them (cooperative launch; GMMLOC_POSE_COOP=0 | 2..4)
    int nb = std::min(4, (M + 255) / 256);
    while (nb > 1 && B * nb > c->ncu) --nb;  // one frame 0.35 -> 0.28 ms, 64 frames 0.42 -> 0.35 ms (1 000 edges)
    bool coop = nb > 1;
    if (c->opt.pose_coop >= 0) {
      const int v = (int)c->opt.pose_coop;
      coop = v >= 2 && B * v <= 512;
      if (coop) nb = std::min(v, 4);
    }
    int one = 1;
    unsigned long long* none = nullptr;
    bool done = false;
    if (coop) {
      unsigned long long* parts = (unsigned long long*)((char*)scratch + chi_bytes);
      double* chi = (double*)scratch;
      GL_HIP(hipMemsetAsync(parts, 0, (size_t)B * 2 * nb * 64 * sizeof(unsigned long long), c->stream));
      void* args[] = {&kp, &B, &M, &pose_dev, &Xw_dev, &obs_dev, &octave_dev, &outlier_dev, &ninlier_dev, &chi, &nb, &parts};
      if (hipLaunchCooperativeKernel((const void*)k_optimize_current_pose<4>, dim3(B * nb), dim3(256), args, 0, c->stream) ==
          hipSuccess)
        done = true;
      else
        (void)hipGetLastError();  // not co-resident: the ordinary shapes below
    }
    if (done) {
    } else if (nw == 8)
      k_optimize_current_pose<8><<<B, 512, 0, c->stream>>>(kp, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev,
                                                          ninlier_dev, (double*)scratch, one, none);
    else if (nw == 4)
      k_optimize_current_pose<4><<<B, 256, 0, c->stream>>>(kp, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev,
                                                          ninlier_dev, (double*)scratch, one, none);
    else
      k_optimize_current_pose<1><<<B, 64, 0, c->stream>>>(kp, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev,
                                                         ninlier_dev, (double*)scratch, one, none);
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}
