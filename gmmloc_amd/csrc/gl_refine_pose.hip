// Tracking::optimizeCurrentPose (tracking_opt.cpp:21-217) as one persistent
// workgroup per frame: 6-DoF Levenberg-Marquardt over pose-only reprojection
// edges (g2o EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose, Huber),
// 4 gating rounds x optimize(10), entirely on-chip:
//   * every thread owns the edges e = tid, tid+T, ...; one pass computes the
//     residual, chi2, Huber weight, the 2|3 x 6 Jacobian and accumulates the 21
//     unique entries of J^T W J, the 6 of b and the robust chi2 in registers;
//   * one deterministic wave reduce-scatter + LDS step (gld::block_reduce) gives
//     all 28 sums to every thread, which then solve the 6x6 (LDL^T) and apply
//     exp(delta) redundantly -- no single-lane section, no global round trip;
//   * the evaluation at the trial pose also builds the next system, so an
//     accepted LM step costs ONE pass over the edges.
// g2o control flow restated: OptimizationAlgorithmLevenberg::solve (lambda init
// 1e-5 max diag, rho test with +1e-3, x1/3..2/3 / x nu schedule, 10 trials),
// SparseOptimizer::optimize, levels via initializeOptimization(0), stale per-edge
// errors read by e->chi2() after the last trial.
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

constexpr int T_POSE = 256;
constexpr int NW_POSE = T_POSE / 64;

// one pass over the frame's active edges at pose (R, t)
//  acc[0..20] H upper triangle (row-major), acc[21..26] b, acc[27] robust chi2
GL_DEV void pose_eval(const PoseKParams& kp, const double* R, const double* t, bool robust, int M,
                      const double* __restrict__ Xw, const double* __restrict__ obs,
                      const int32_t* __restrict__ octave, const uint8_t* __restrict__ level,
                      double* __restrict__ chi2_e, double* acc) {
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int e = threadIdx.x; e < M; e += T_POSE) {
    const int oc = octave[e];
    if (oc < 0 || level[e] != 0) continue;
    const double X = Xw[(size_t)e * 3 + 0], Y = Xw[(size_t)e * 3 + 1], Z = Xw[(size_t)e * 3 + 2];
    const double ou = obs[(size_t)e * 3 + 0], ov = obs[(size_t)e * 3 + 1], our = obs[(size_t)e * 3 + 2];
    const bool stereo = !(our < 0);
    const double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double invz = 1.0 / z, invz2 = invz * invz;
    const double pu = x * invz * kp.fx + kp.cx;
    const double pv = y * invz * kp.fy + kp.cy;
    const double e0 = ou - pu, e1 = ov - pv;
    const double e2 = stereo ? (our - (pu - kp.bf * invz)) : 0.0;
    const double s = kp.s2inv[oc];
    const double chi2 = e0 * (s * e0) + e1 * (s * e1) + e2 * (s * e2);
    chi2_e[e] = chi2;
    double rho0 = chi2, rho1 = 1.0;
    if (robust) huber(chi2, stereo ? kp.delta_stereo : kp.delta_mono, rho0, rho1);
    const double w = rho1 * s;
    // Jacobian rows (g2o types_six_dof_expmap.cpp, EdgeSE3ProjectXYZOnlyPose::linearizeOplus)
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
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) {
        acc[q] += w * (J0[i] * J0[j] + J1[i] * J1[j] + J2[i] * J2[j]);
        ++q;
      }
    // b += J^T * (-rho1 * Omega * err)
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] -= w * (J0[i] * e0 + J1[i] * e1 + J2[i] * e2);
    acc[27] += rho0;
  }
}

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

GL_DEV void unpack_sym6(const double* acc, double* H) {
  int q = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      H[i * 6 + j] = acc[q];
      H[j * 6 + i] = acc[q];
      ++q;
    }
}

__global__ __launch_bounds__(T_POSE) void k_optimize_current_pose(PoseKParams kp, int B, int M,
                                                                  double* __restrict__ pose_io,
                                                                  const double* __restrict__ Xw_all,
                                                                  const double* __restrict__ obs_all,
                                                                  const int32_t* __restrict__ oct_all,
                                                                  uint8_t* __restrict__ outlier_all,
                                                                  int32_t* __restrict__ ninlier,
                                                                  double* __restrict__ chi2_all) {
  __shared__ double red[NW_POSE * 32];
  const int f = blockIdx.x;
  if (f >= B) return;
  const double* Xw = Xw_all + (size_t)f * M * 3;
  const double* obs = obs_all + (size_t)f * M * 3;
  const int32_t* octave = oct_all + (size_t)f * M;
  uint8_t* level = outlier_all + (size_t)f * M;  // is_outlier_ <=> level 1
  double* chi2_e = chi2_all + (size_t)f * M;

  const SE3 T0 = se3_load(pose_io + (size_t)f * 7);
  double acc[32];

  // graph construction: count edges, clear outlier flags (tracking_opt.cpp:60-137)
  {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    for (int e = threadIdx.x; e < M; e += T_POSE) {
      level[e] = 0;
      if (octave[e] >= 0) acc[0] += 1.0;
    }
    block_reduce<1, NW_POSE>(acc, red);
  }
  const int n_init = (int)acc[0];
  if (n_init < 3) {  // :139-140
    if (threadIdx.x == 0) ninlier[f] = 0;
    return;
  }

  SE3 T = T0;
  bool robust = true;
  int nbad = 0;
  for (int round = 0; round < 4; ++round) {
    T = T0;  // vertex_se3->setEstimate(curr_frame_->getTcw())  (:152)
    // initializeOptimization(0): active = level-0 edges
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    for (int e = threadIdx.x; e < M; e += T_POSE)
      if (octave[e] >= 0 && level[e] == 0) acc[0] += 1.0;
    block_reduce<1, NW_POSE>(acc, red);
    const int nactive = (int)acc[0];

    if (nactive > 0) {  // optimize(10); returns -1 untouched when nothing is active
      double R[9], H[36], b[6];
      qtoR(T.r, R);
      pose_eval(kp, R, T.t, robust, M, Xw, obs, octave, level, chi2_e, acc);
      block_reduce<28, NW_POSE>(acc, red);
      unpack_sym6(acc, H);
#pragma unroll
      for (int i = 0; i < 6; ++i) b[i] = acc[21 + i];
      double currentChi = acc[27];
      bool sys_valid = true;
      double lambda = 0.0, ni = 2.0;
      for (int it = 0; it < 10; ++it) {
        if (!sys_valid) {  // computeActiveErrors + buildSystem at the (restored) estimate
          qtoR(T.r, R);
          pose_eval(kp, R, T.t, robust, M, Xw, obs, octave, level, chi2_e, acc);
          block_reduce<28, NW_POSE>(acc, red);
          unpack_sym6(acc, H);
#pragma unroll
          for (int i = 0; i < 6; ++i) b[i] = acc[21 + i];
          currentChi = acc[27];
          sys_valid = true;
        }
        if (it == 0) {  // computeLambdaInit: tau * max |diag|
          double md = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) md = fmax(fabs(H[i * 6 + i]), md);
          lambda = 1e-5 * md;
          ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
          double Hl[36], dx[6];
#pragma unroll
          for (int i = 0; i < 36; ++i) Hl[i] = H[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) Hl[i * 6 + i] += lambda;
          const bool ok2 = ldlt_solve<6>(Hl, b, dx, true);
          SE3 Tn = T;
          double tempChi;
          double Hn[36], bn[6];
          if (ok2) {
            Tn = se3_mul(se3_exp(dx), T);
            double Rn[9];
            qtoR(Tn.r, Rn);
            pose_eval(kp, Rn, Tn.t, robust, M, Xw, obs, octave, level, chi2_e, acc);
            block_reduce<28, NW_POSE>(acc, red);
            unpack_sym6(acc, Hn);
#pragma unroll
            for (int i = 0; i < 6; ++i) bn[i] = acc[21 + i];
            tempChi = acc[27];
          } else {
            tempChi = 1.7976931348623157e308;
          }
          double scale = 0.0;
#pragma unroll
          for (int j = 0; j < 6; ++j) scale += dx[j] * (lambda * dx[j] + b[j]);
          scale += 1e-3;
          rho = (currentChi - tempChi) / scale;
          if (rho > 0 && isfinite(tempChi)) {
            const double u = 2 * rho - 1;
            double alpha = 1. - u * u * u;
            alpha = fmin(alpha, 2. / 3.);
            const double sf = fmax(1. / 3., alpha);
            lambda *= sf;
            ni = 2;
            currentChi = tempChi;
            T = Tn;
#pragma unroll
            for (int i = 0; i < 36; ++i) H[i] = Hn[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = bn[i];
          } else {
            lambda *= ni;
            ni *= 2;
            // estimate restored (pop); H, b stay; per-edge errors stay those of the
            // rejected trial until the next computeActiveErrors
            if (!(rho < 0)) sys_valid = false;
          }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;  // Terminate
      }
    }

    // gating (:156-203): outliers are re-evaluated at the current estimate, inliers
    // use the error of the last computeActiveErrors; chi2 compared as float.
    double Rg[9];
    qtoR(T.r, Rg);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    for (int e = threadIdx.x; e < M; e += T_POSE) {
      const int oc = octave[e];
      if (oc < 0) continue;
      double c2;
      if (level[e] != 0)
        c2 = pose_edge_chi2(kp, Rg, T.t, Xw + (size_t)e * 3, obs + (size_t)e * 3, oc);
      else
        c2 = chi2_e[e];
      const bool stereo = !(obs[(size_t)e * 3 + 2] < 0);
      const float thr = stereo ? 7.815f : 5.991f;
      const bool bad = (float)c2 > thr;
      level[e] = bad ? 1 : 0;
      if (bad) acc[0] += 1.0;
    }
    block_reduce<1, NW_POSE>(acc, red);
    nbad = (int)acc[0];
    if (round == 2) robust = false;  // e->setRobustKernel(0) at it == 2
    if (n_init < 10) break;          // optimizer.edges().size() < 10
  }
  if (threadIdx.x == 0) {
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
  int rc = gl::ctx_scratch(c, (size_t)B * (M > 0 ? M : 1) * sizeof(double), &scratch);
  if (rc != GL_OK) return rc;
  {
    gl::TimerScope ts(c, GL_TIMER_REFINE_POSE);
    k_optimize_current_pose<<<B, T_POSE, 0, c->stream>>>(kp, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev,
                                                         ninlier_dev, (double*)scratch);
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}
