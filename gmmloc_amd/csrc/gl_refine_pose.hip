// Tracking::optimizeCurrentPose (tracking_opt.cpp:21-217): 6-DoF Levenberg-Marquardt over pose-only
// reprojection edges (g2o EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose, Huber), 4 gating rounds
// x optimize(10), one persistent group of NW waves per frame, entirely on-chip:
//   * a wave owns groups of <= 4 chunks of 64 consecutive edges; one pass computes the residual, chi2, Huber
//     weight, the 2|3 x 6 Jacobian and accumulates the 21 unique entries of J^T W J, the 6 of b and the
//     robust chi2 of a group in registers;
//   * a wave reduce-scatter (permlane swaps + DPP) + one LDS step give the 28 sums; the current and the trial
//     system live in LDS (224 B each), not in registers;
//   * every wave solves the 6x6 (LDL^T on the packed triangle, reciprocal pivots) and applies exp(delta) on
//     its own - pose as (R, t) in SGPRs, Rodrigues with a short series below |theta| = 0.01 - so nothing is
//     broadcast;
//   * the evaluation at the trial pose also builds the next system: an accepted LM step costs ONE pass.
// One wave per frame (no barrier at all, GL_POSE_WPS frames per SIMD) is the shape for large batches, a wave per
// group of <= 256 edges (up to 8) the one for the frame-at-a-time caller: 0.34 ms instead of 0.90 ms for one frame of
// 1 000 edges.  Every shape adds in the same canonical order (group_totals28): same bits whatever the batch.
// g2o control flow restated: OptimizationAlgorithmLevenberg::solve (lambda init 1e-5 max diag, rho test with
// +1e-3, x1/3..2/3 / x nu schedule, 10 trials), SparseOptimizer::optimize, levels via
// initializeOptimization(0), stale per-edge errors read by e->chi2() after the last trial (SURVEY.md
// Appendix A).
#include <algorithm>
#include <cstdlib>

#include "gl_device.hpp"
#include "gl_internal.hpp"
#include "gl_pose_compact.hpp"

#pragma clang fp contract(fast)

using namespace gld;

namespace {

// The residuals are kept in NORMALISED image coordinates (round 5; the structure refine's form, gl_ba_fast_impl.hpp): an observation
// enters as on = ((u - cx) / fx, (v - cy) / fy, (u_right - cx) / fx) - one fma per coordinate, the same expression in every launch
// shape -, so e_n = on - (x/z, y/z, (x - bn)/z) needs no intrinsics and chi2 = sx (e0^2 + e2^2) + sy e1^2 with sx = fx^2 / sigma^2,
// sy = fy^2 / sigma^2 per octave: the quadratic form s |obs - K pi(q)|^2 of EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose.
struct PoseKParams {
  double ifx, ify, ncx, ncy;  // 1 / fx, 1 / fy, -cx / fx, -cy / fy
  double bn;                  // bf / fx: the baseline in normalised image units
  double sx[8], sy[8];
  double delta_mono, delta_stereo;
};

GL_DEV double rcp_nr(double a) {
  double x = __builtin_amdgcn_rcp(a);
  x = fma(fma(-a, x, 1.0), x, x);
  x = fma(fma(-a, x, 1.0), x, x);
  return x;
}
GL_DEV double rsq_nr(double a) {
  double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}
// branch-free g2o::RobustKernelHuber (rho, rho'): no square root, no division, no divergent region in the edge's chain
GL_DEV void huber_bf(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  const double r = rsq_nr(fmax(e, 1e-300));
  const bool in = e <= dsqr;
  rho1 = in ? 1.0 : delta * r;
  rho0 = in ? e : (2.0 * delta * (e * r) - dsqr);
}
// camera-frame point q = R X + t, 1 / z, the normalised residual and the un-robustified chi2 of one edge (e->computeError();
// e->chi2()); on = the normalised observation, s2tab = {sx[8], sy[8]}
// (act = false: an edge that is absent or at level 1, evaluated all the same so that a thread's edges are ONE straight line of code
// the scheduler can interleave - 1 / z and the weights are zero for it, every term it adds to a sum is an exact +-0.0)
GL_DEV double pose_edge_res(const double* R, const double* t, double bn, const double* __restrict__ s2tab, int oc, bool stereo,
                            double X, double Y, double Z, double o0, double o1, double o2, double* q, double& iz, double* e,
                            double& sx, double& sy, bool act = true) {
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = fma(R[k * 3], X, fma(R[k * 3 + 1], Y, fma(R[k * 3 + 2], Z, t[k])));
  iz = rcp_nr(q[2]);
  iz = act ? iz : 0.0;
  e[0] = fma(-q[0], iz, o0);
  e[1] = fma(-q[1], iz, o1);
  e[2] = stereo ? fma(bn - q[0], iz, o2) : 0.0;  // u_right - (x - b) / z
  sx = act ? s2tab[oc] : 0.0;
  sy = act ? s2tab[8 + oc] : 0.0;
  return fma(sy * e[1], e[1], sx * fma(e[2], e[2], e[0] * e[0]));
}
GL_DEV double pose_edge_chi2(const PoseKParams& kp, const double* __restrict__ s2tab, const double* R, const double* t, const double* Xw,
                             const double* on, int oc, bool stereo) {
  double q[3], e[3], iz, sx, sy;
  return pose_edge_res(R, t, kp.bn, s2tab, oc, stereo, Xw[0], Xw[1], Xw[2], on[0], on[1], on[2], q, iz, e, sx, sy);
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
// ---- ONE summation order for every launch shape (the same canonical order as the structure refine,
// gl_ba_fast_impl.hpp): chunks of 64 consecutive edges, nch = ceil(M / 64); G = ceil(nch / 4) groups of
// S = ceil(nch / G) <= 4 consecutive chunks; lane j of group g adds the terms of its edges (g S + i) 64 + j in slot
// order; the 64 lane sums of a group meet in the butterfly of gld::wave_reduce_scatter32; blocks of two groups,
// B_k = g_2k + g_2k+1, are added in order.  A frame runs on NW waves (1 for large batches - no barrier at all -, up to
// 8 for few frames); wave w owns the groups w, w + NW, ...: whatever NW, every sum is built from the same terms in the
// same order, so the refined pose does not depend on how many frames ride in the call.
GL_DEV double add_nc(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
// group g's 28 lane sums (acc) -> its totals in red[g][0..31]
GL_DEV void group_totals28(double* acc, double* red, int g) {
#pragma unroll
  for (int i = 28; i < 32; ++i) acc[i] = 0.0;
  const double r = wave_reduce_scatter32(acc);
  const int lane = threadIdx.x & 63;
  if (wave_slot_owner(lane)) red[g * 32 + wave_slot(lane)] = r;
}
// red[0..G-1][32] -> dst[0..27]: blocks of two groups in order (wave 0; the caller brackets it with barriers)
GL_DEV void frame_totals28(const double* red, double* dst, int G) {
  const int lane = threadIdx.x;
  if (lane < 32) {
    double s = G > 1 ? add_nc(red[lane], red[32 + lane]) : red[lane];
    for (int b = 1; 2 * b < G; ++b) {
      const double blk = 2 * b + 1 < G ? add_nc(red[(2 * b) * 32 + lane], red[(2 * b + 1) * 32 + lane]) : red[(2 * b) * 32 + lane];
      s = add_nc(s, blk);
    }
    dst[lane] = s;
  }
}
// a count per thread -> the frame's total (integers: exact in any order)
GL_DEV double block_total1(double v, double* part) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += shfl_xor_f64(v, o);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return uni(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = part[0];
  for (int w = 1; w < nw; ++w) s += part[w];
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
  // dR = I + a [w]x + b [w]x^2,  V = I + b [w]x + c [w]x^2  with  [w]x^2 = w w^T - |w|^2 I  written out (the generic 3 x 3 products
  // multiply by the zeros of the skew matrix: the serial part of every pass)
  const double w0 = u[0], w1 = u[1], w2 = u[2];
  const double s00 = w0 * w0 - th2, s11 = w1 * w1 - th2, s22 = w2 * w2 - th2;
  const double s01 = w0 * w1, s02 = w0 * w2, s12 = w1 * w2;
  double dR[9], V[9];
  dR[0] = fma(b, s00, 1.0);
  dR[4] = fma(b, s11, 1.0);
  dR[8] = fma(b, s22, 1.0);
  dR[1] = fma(b, s01, -a * w2);
  dR[3] = fma(b, s01, a * w2);
  dR[2] = fma(b, s02, a * w1);
  dR[6] = fma(b, s02, -a * w1);
  dR[5] = fma(b, s12, -a * w0);
  dR[7] = fma(b, s12, a * w0);
  V[0] = fma(c, s00, 1.0);
  V[4] = fma(c, s11, 1.0);
  V[8] = fma(c, s22, 1.0);
  V[1] = fma(c, s01, -b * w2);
  V[3] = fma(c, s01, b * w2);
  V[2] = fma(c, s02, b * w1);
  V[6] = fma(c, s02, -b * w1);
  V[5] = fma(c, s12, -b * w0);
  V[7] = fma(c, s12, b * w0);
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

// The same system by 3 x 3 BLOCKS (round 6; see ldlt6_blocked in gl_ba_fast_impl.hpp): [[A, B], [B^T, C]] - A = L_a D_a L_a^T, Y = A^-1 [B | b_r],
// S = C - B^T Y_B, x_t = S^-1 (b_t - B^T y), x_r = y - Y_B x_t - the pivots (and the failure test) of the unpivoted LDL^T of the whole matrix,
// about 50 instructions deep where the column-by-column factorisation is a chain of ~190: every wave of the frame-at-a-time shapes runs
// this solve on a SIMD of its own, once per pass.
#ifndef GL_POSE_SOLVE_BLOCKED
#define GL_POSE_SOLVE_BLOCKED 1
#endif
GL_DEV bool pose_ldl3_factor(const double* D, double* f) {  // f = {l10, l20, l21, 1/d0, 1/d1, 1/d2}; false on a non-positive / non-finite pivot
  const double i0 = rcp_nr(D[0]);
  const double m2 = fma(D[0], D[3], -(D[1] * D[1]));  // d0 d1
  const double e2 = fma(D[0], D[4], -(D[1] * D[2]));  // d0 (D12 - l10 D02)
  const double im = rcp_nr(m2);
  const double l1 = D[1] * i0, l2 = D[2] * i0;
  const double i1 = D[0] * im, l3 = e2 * im;
  const double e = e2 * i0;
  const double d2 = fma(-l3, e, fma(-l2, D[2], D[5]));
  f[0] = l1;
  f[1] = l2;
  f[2] = l3;
  f[3] = i0;
  f[4] = i1;
  f[5] = rcp_nr(d2);
  const double d1 = m2 * i0;
  return D[0] > 0.0 && isfinite(D[0]) && d1 > 0.0 && isfinite(d1) && d2 > 0.0 && isfinite(d2);
}
GL_DEV void pose_ldl3_solve(const double* f, const double* b, double* x) {
  const double y1 = fma(-f[0], b[0], b[1]);
  const double y2 = fma(-f[2], y1, fma(-f[1], b[0], b[2]));
  x[2] = y2 * f[5];
  x[1] = fma(-f[2], x[2], y1 * f[4]);
  x[0] = fma(-f[1], x[2], fma(-f[0], x[1], b[0] * f[3]));
}
GL_DEV bool ldlt6_blocked_pos(const double* a, const double* b, double lambda, double* x) {
  const double A[6] = {a[GL_PU(0, 0)] + lambda, a[GL_PU(0, 1)], a[GL_PU(0, 2)], a[GL_PU(1, 1)] + lambda, a[GL_PU(1, 2)], a[GL_PU(2, 2)] + lambda};
  double fa[6];
  const bool ok_a = pose_ldl3_factor(A, fa);
  double Y[3][3], y[3];  // Y[j] = A^-1 (column j of B)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double col[3] = {a[GL_PU(0, 3 + j)], a[GL_PU(1, 3 + j)], a[GL_PU(2, 3 + j)]};
    pose_ldl3_solve(fa, col, Y[j]);
  }
  pose_ldl3_solve(fa, b, y);
  double S[6], s[3];
  {
    const int ri[6] = {0, 0, 0, 1, 1, 2}, ci[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int i = ri[e], j = ci[e];
      const double c = a[GL_PU(3 + i, 3 + j)] + (i == j ? lambda : 0.0);
      S[e] = fma(-a[GL_PU(2, 3 + i)], Y[j][2], fma(-a[GL_PU(1, 3 + i)], Y[j][1], fma(-a[GL_PU(0, 3 + i)], Y[j][0], c)));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = fma(-a[GL_PU(2, 3 + i)], y[2], fma(-a[GL_PU(1, 3 + i)], y[1], fma(-a[GL_PU(0, 3 + i)], y[0], b[3 + i])));
  }
  double fs[6];
  const bool ok_s = pose_ldl3_factor(S, fs);
  pose_ldl3_solve(fs, s, x + 3);
#pragma unroll
  for (int k = 0; k < 3; ++k) x[k] = fma(-Y[2][k], x[5], fma(-Y[1][k], x[4], fma(-Y[0][k], x[3], y[k])));
  return ok_a && ok_s;
}

// one edge: residual, chi2, Huber weight, and its terms of the pose system into acc[0..27].  With j_r the rows of the projection
// Jacobian in normalised coordinates - (iz, 0, c0), (0, iz, c1) and, stereo, (iz, 0, c2) - the edge's Jacobian is -j G,
// G = dq / dxi = [-[q]x | I] (VertexSE3Expmap: exp(xi) q), so  H += G^T A G,  b += G^T a  with the 3 x 3 block
// A = sum_r w_r j_r^T j_r (A(0,1) = 0: no row touches x and y) and a = sum_r w_r j_r e_r: 119 instructions where the three 6-vectors
// J_r and 27 x 3 multiply-adds were 232.  acc: H upper triangle row-major (21), b (6), robust chi2 (27).
GL_DEV void pose_edge_v(const PoseKParams& kp, const double* __restrict__ s2tab, const PoseRt& P, bool robust, int oc, bool stereo, double X,
                        double Y, double Z, double o0, double o1, double o2, double& chi2_out, double* acc, bool act = true) {
  double q[3], e[3], iz, sx, sy;
  const double chi2 = pose_edge_res(P.R, P.t, kp.bn, s2tab, oc, stereo, X, Y, Z, o0, o1, o2, q, iz, e, sx, sy, act);
  chi2_out = act ? chi2 : chi2_out;
  double rho0 = chi2, rho1 = 1.0;
  if (robust) huber_bf(chi2, stereo ? kp.delta_stereo : kp.delta_mono, rho0, rho1);
  const double wx = rho1 * sx, wy = rho1 * sy, t = stereo ? wx : 0.0;
  const double iz2 = iz * iz;
  const double c0 = -q[0] * iz2, c1 = -q[1] * iz2, c2 = fma(kp.bn, iz2, c0);
  const double wxc0 = wx * c0, wyc1 = wy * c1, tc2 = t * c2, wyiz = wy * iz;
  const double A0 = (wx + t) * iz2, A2 = iz * (wxc0 + tc2), A3 = wy * iz2, A4 = wyiz * c1;
  const double A5 = fma(tc2, c2, fma(wyc1, c1, wxc0 * c0));
  const double a0 = iz * fma(t, e[2], wx * e[0]), a1 = wyiz * e[1], a2 = fma(tc2, e[2], fma(wyc1, e[1], wxc0 * e[0]));
  // M = [q]x A (row r, column j) = (q x A[:, j])[r], A = {A0 0 A2; 0 A3 A4; A2 A4 A5}
  const double M0 = q[1] * A2, M1 = fma(q[1], A4, -q[2] * A3), M2 = fma(q[1], A5, -q[2] * A4);
  const double M3 = fma(q[2], A0, -q[0] * A2), M4 = -q[0] * A4, M5 = fma(q[2], A2, -q[0] * A5);
  const double M6 = -q[1] * A0, M7 = q[0] * A3, M8 = fma(q[0], A4, -q[1] * A2);
  // every term is rounded on its own before it enters its sum (add_nc): the same bits in every launch shape
  acc[0] = add_nc(acc[0], fma(q[1], M2, -q[2] * M1));
  acc[1] = add_nc(acc[1], fma(q[2], M0, -q[0] * M2));
  acc[2] = add_nc(acc[2], fma(q[0], M1, -q[1] * M0));
  acc[3] = add_nc(acc[3], M0);
  acc[4] = add_nc(acc[4], M1);
  acc[5] = add_nc(acc[5], M2);
  acc[6] = add_nc(acc[6], fma(q[2], M3, -q[0] * M5));
  acc[7] = add_nc(acc[7], fma(q[0], M4, -q[1] * M3));
  acc[8] = add_nc(acc[8], M3);
  acc[9] = add_nc(acc[9], M4);
  acc[10] = add_nc(acc[10], M5);
  acc[11] = add_nc(acc[11], fma(q[0], M7, -q[1] * M6));
  acc[12] = add_nc(acc[12], M6);
  acc[13] = add_nc(acc[13], M7);
  acc[14] = add_nc(acc[14], M8);
  acc[15] = add_nc(acc[15], A0);
  // (acc[16] = H(3,4): A(0,1) = 0 - nothing to add)
  acc[17] = add_nc(acc[17], A2);
  acc[18] = add_nc(acc[18], A3);
  acc[19] = add_nc(acc[19], A4);
  acc[20] = add_nc(acc[20], A5);
  acc[21] = add_nc(acc[21], fma(q[1], a2, -q[2] * a1));
  acc[22] = add_nc(acc[22], fma(q[2], a0, -q[0] * a2));
  acc[23] = add_nc(acc[23], fma(q[0], a1, -q[1] * a0));
  acc[24] = add_nc(acc[24], a0);
  acc[25] = add_nc(acc[25], a1);
  acc[26] = add_nc(acc[26], a2);
  acc[27] = add_nc(acc[27], rho0);
}

// the same from global memory (the shapes whose waves own more edges than registers hold)
GL_DEV void pose_edge(const PoseKParams& kp, const double* __restrict__ s2tab, const PoseRt& P, bool robust, int e,
                      const double* __restrict__ Xw, const double* __restrict__ obs, const int32_t* __restrict__ octave,
                      const uint8_t* __restrict__ level, double* __restrict__ chi2_e, double* acc) {
  const int oc = octave[e];
  if (oc < 0 || level[e] != 0) return;
  double c2;
  const double our = obs[(size_t)e * 3 + 2];
  pose_edge_v(kp, s2tab, P, robust, oc, !(our < 0), Xw[(size_t)e * 3 + 0], Xw[(size_t)e * 3 + 1], Xw[(size_t)e * 3 + 2],
              fma(obs[(size_t)e * 3 + 0], kp.ifx, kp.ncx), fma(obs[(size_t)e * 3 + 1], kp.ify, kp.ncy), fma(our, kp.ifx, kp.ncx), c2, acc);
  chi2_e[e] = c2;
}

// The frame-at-a-time shapes (a wave per group, G <= 8): a thread owns the <= 4 edges (wave S + i) 64 + lane of its
// group for the whole kernel and keeps them in REGISTERS - map point, observation, octave, level, stale chi2 - so a
// trial touches no global memory at all (two dependent L2 round trips per chunk before: most of a trial's 9 us).
// (Eight waves = two per SIMD have 256 registers each, not enough for four edges' coordinates next to the 28 sums:
// that instance keeps map point and observation in LDS - [slot][coordinate][thread], conflict-free - and the rest in
// registers.)
#ifdef GL_POSE_PROF  // diagnosis build (tools/pose_prof.py): thread 0's clock64() deltas per phase, returned in the frame's pose
#define POSE_PT(v) const long long v = clock64()
#define POSE_PADD(i, a, b) g_pf[i] += (b) - (a)
#define POSE_PARG , g_pf
#else
#define POSE_PT(v)
#define POSE_PADD(i, a, b)
#define POSE_PARG
#endif
struct EdgeRegs {
  double X[5][3], O[5][3], c2[5];
  int oc[5];   // octave | stereo << 4, < 0: no edge in the slot (no map point, or beyond the frame); O = the NORMALISED observation
  int lv[5];   // level = is_outlier_
};             // (slot 4: REGS == 3 only)
// REGS == 3, a frame of FIVE groups (1 025 .. 1 280 edges: the reference's 1 200 features) at the frame-at-a-time caller.  A wave per
// group puts two waves on one SIMD and one on the other three, and every pass waits for that SIMD: 13.2 k ticks per pass at 1 200 edges
// against 8.1 k at 1 000 (profiles/r5_pose_ab.txt).  Eight waves - the groups 0..3 on the waves 0..3, one chunk of group 4 on each of the
// waves 4..7 - were built first and bought 6 % (profiles/r5_pose_split.txt): two waves per SIMD repeat the serial solve beside each other
// and the older one waits 1.2 k ticks for the younger at every evaluation's first barrier.  So: FOUR waves, a SIMD each, wave w owns
// group w as ever and, as a FIFTH edge per thread, chunk w of group 4.  Group 4's level 1 - lane j adds its slots' terms in slot order
// - runs through an LDS transpose [value][slot][lane]; the four waves then share its 28 values in rounds of 8 and finish with
// wave_reduce_scatter8, which pairs the lanes like wave_reduce_scatter32: the same canonical order, the same bits (what the SPREAD shape
// of the structure refine does, gl_ba_fast_impl.hpp).
template <int REGS>
constexpr int pose_slots() { return REGS == 3 ? 5 : 4; }
// frame-local edge of slot i of this thread, or -1 (the caller tests e < M)
template <int REGS>
GL_DEV int pose_edge_index(int S, int i) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (REGS == 3 && i == 4) return w < S ? (4 * S + w) * 64 + l : -1;
  return i < S ? (w * S + i) * 64 + l : -1;
}
template <int MODE>
GL_DEV void edge_xo(const EdgeRegs& E, const double* xo, int i, double* X, double* O) {
  const int T = blockDim.x, t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    X[j] = (MODE == 1 || MODE == 3) ? E.X[i][j] : xo[(i * 6 + j) * T + t];
    O[j] = (MODE == 1 || MODE == 3) ? E.O[i][j] : xo[(i * 6 + 3 + j) * T + t];
  }
}
template <int MODE>
GL_DEV void pose_eval_regs(const PoseKParams& kp, const double* __restrict__ s2tab, const PoseRt& P, bool robust, int G, int S, EdgeRegs& E,
                           const double* xo, double* red, double* dst
#ifdef GL_POSE_PROF
                           , long long* g_pf
#endif
                           ) {
  const int wave = threadIdx.x >> 6;
  double acc[32];
  POSE_PT(tq0);
  __syncthreads();
  POSE_PT(tq1);
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  if (MODE == 3) {
    // five groups: the thread's four edges of group `wave` and, into sums of its own, chunk `wave` of group 4 - one basic block
    double acc5[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc5[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      double X[3], O[3];
      edge_xo<MODE>(E, xo, i, X, O);
      pose_edge_v(kp, s2tab, P, robust, E.oc[i] & 15, (E.oc[i] & 16) != 0, X[0], X[1], X[2], O[0], O[1], O[2], E.c2[i], i < 4 ? acc : acc5,
                  E.oc[i] >= 0 && E.lv[i] == 0);
    }
    double* tb = const_cast<double*>(xo);  // [value < 28][slot < 4][lane]: the terms of group 4 (0 + t = t)
#pragma unroll
    for (int v = 0; v < 28; ++v) tb[(v * 4 + wave) * 64 + (threadIdx.x & 63)] = acc5[v];
  } else if (S == 4) {
    // the frame's groups have four chunks (every frame of more than 768 edges): the thread's four edges as ONE basic block - no region
    // per edge -, so that the scheduler interleaves their dependent chains (a wave of these shapes has its SIMD to itself); an
    // absent / level-1 edge adds exact zeros (pose_edge_res): the same bits as the loop below
    // (round 6: a slot in which NO lane of the wave holds an edge is skipped - a wave-uniform test; its terms would all be exact zeros.
    // Problems whose edges were compacted and dealt over the waves - k_pose_compact below - leave most slots empty.)
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(E.oc[0] >= 0 && E.oc[1] >= 0 && E.oc[2] >= 0 && E.oc[3] >= 0) != 0ull))) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double X[3], O[3];
        edge_xo<MODE>(E, xo, i, X, O);
        pose_edge_v(kp, s2tab, P, robust, E.oc[i] & 15, (E.oc[i] & 16) != 0, X[0], X[1], X[2], O[0], O[1], O[2], E.c2[i], acc,
                    E.oc[i] >= 0 && E.lv[i] == 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!__builtin_amdgcn_readfirstlane((int)(__ballot(E.oc[i] >= 0) != 0ull))) continue;
        double X[3], O[3];
        edge_xo<MODE>(E, xo, i, X, O);
        pose_edge_v(kp, s2tab, P, robust, E.oc[i] & 15, (E.oc[i] & 16) != 0, X[0], X[1], X[2], O[0], O[1], O[2], E.c2[i], acc,
                    E.oc[i] >= 0 && E.lv[i] == 0);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (E.oc[i] >= 0 && E.lv[i] == 0) {
        double X[3], O[3];
        edge_xo<MODE>(E, xo, i, X, O);
        pose_edge_v(kp, s2tab, P, robust, E.oc[i] & 15, (E.oc[i] & 16) != 0, X[0], X[1], X[2], O[0], O[1], O[2], E.c2[i], acc);
      }
  }
  POSE_PT(tq2);
  group_totals28(acc, red, wave);
  if (MODE == 3) {
    __syncthreads();
    {  // group 4: level 1 in slot order, level 2 = the butterfly; wave k takes the values 8 k .. 8 k + 7
      const double* tb = xo;
      const int k = wave, l = threadIdx.x & 63;
      double y[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int val = k * 8 + kk;
        double x[4];  // (all four slots before the first add; an absent slot holds the zeros of its inactive edge)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = val < 28 ? tb[(val * 4 + j) * 64 + l] : 0.0;
        double s_ = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s_ = add_nc(s_, x[j]);
        y[kk] = s_;
      }
      const double t8 = wave_reduce_scatter8(y);
      const int vi = k * 8 + ((l >> 5) & 1) * 4 + ((l >> 4) & 1) * 2 + (l & 1);
      if ((l & 0xE) == 0) red[4 * 32 + vi] = t8;  // (values 28 .. 31: sums of zeros, like group_totals28 leaves them)
    }
  }
  POSE_PT(tq3);
  __syncthreads();
  if (wave == 0) frame_totals28(red, dst, G);
  __syncthreads();
  POSE_PT(tq4);
  POSE_PADD(0, tq0, tq1);  // entry barrier
  POSE_PADD(1, tq1, tq2);  // the thread's edges
  POSE_PADD(2, tq2, tq3);  // reduce-scatter of the group
  POSE_PADD(3, tq3, tq4);  // two barriers + the blocks
  POSE_PADD(6, 0, 1);      // passes
}

// one pass over the frame's edges at pose P -> dst[0..27] (LDS): H upper triangle (21), b (6), robust chi2.
// Wave w walks the groups w, w + NW, ...: lane sums of the group's S chunks in registers, tree, red[g]; then the blocks.
GL_DEV void pose_eval(const PoseKParams& kp, const double* __restrict__ s2tab, const PoseRt& P, bool robust, int G, int S, int M,
                      const double* __restrict__ Xw, const double* __restrict__ obs, const int32_t* __restrict__ octave,
                      const uint8_t* __restrict__ level, double* __restrict__ chi2_e, double* red, double* dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double acc[32];
  __syncthreads();  // red / dst may still be read by slower waves (no-op ordering for a single wave)
  for (int g = wave; g < G; g += nw) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    for (int i = 0; i < S; ++i) {
      const int e = (g * S + i) * 64 + lane;
      if (e >= M) break;
      pose_edge(kp, s2tab, P, robust, e, Xw, obs, octave, level, chi2_e, acc);
    }
    group_totals28(acc, red, g);
  }
  __syncthreads();
  if (wave == 0) frame_totals28(red, dst, G);
  __syncthreads();
}

#ifndef GL_POSE_WPS
#define GL_POSE_WPS 3  // waves per SIMD the register budget is capped for (measured: 3 > 2 > 4, tools/pose_ab.py)
#endif
// B problems of stride M as a launch sees them.  On-chip shapes (REGS != 0) can write an edge's final flag straight into ANOTHER problem:
// src_of[f M + slot] = the edge's place in a problem of stride M_dst whose flags are outlier_dst (a compacted problem answering for the
// caller's, gl_pose_compact.hpp).  skip: the frame's workgroup returns at once when (skip[f] != 0) == skip_if (the compacted and the
// full-stride problem of a frame: one of the two runs).
struct PoseProb {
  int M, G, S;
  double* pose_io;
  const double* Xw;
  const double* obs;
  const int32_t* oct;
  uint8_t* outlier;
  int32_t* ninlier;
  int nin_stride;
  double* chi2;
  const int32_t* src_of;
  uint8_t* outlier_dst;
  int M_dst;
  const int32_t* skip;
  int skip_if;
};

// NW = waves per frame the instance is compiled for (register budget): 1 for large batches (GL_POSE_WPS frames per
// SIMD, no barrier does anything), 4 / 8 when the frames are fewer than the SIMDs; launched with nw <= NW waves
// (never more than the frame has groups).  Every wave repeats the serial part (solve, pose update) on its own so that
// no broadcast is needed.
template <int NW, int REGS>  // REGS: 0 edges from global memory, 1 in registers, 2 coordinates in LDS, 3 five groups on four waves (registers)
__device__ __forceinline__ void pose_frame(const PoseKParams& kp, const int f, const PoseProb& pr) {
  __shared__ double s2tab[16];  // {sx[8], sy[8]}
  __shared__ double Hbuf[64];       // current / trial system {H upper (21), b (6), chi2}: two rows that swap roles when a trial is accepted
  double* H = Hbuf;                 // (no copy, no barrier pair: the next writer of the row that was current is wave 0 behind the next
  double* Hn = Hbuf + 32;           //  evaluation's two barriers, every reader of it is through by then)
  __shared__ double part[NW];       // per-wave counts
  extern __shared__ double red[];   // G x 32 group totals (then, REGS == 2: 4 slots x 6 coordinates x threads)
  const int M = pr.M, G = pr.G, S = pr.S;
  double* xo = red + G * 32;
  const int lane = threadIdx.x;  // "lane" = thread of the workgroup's waves
  if (pr.skip && (pr.skip[f] != 0) == (pr.skip_if != 0)) return;
  const int e0 = lane, es = blockDim.x;  // counting / gating loops: any order (integers, per-edge decisions)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s2tab[j] = kp.sx[j];
      s2tab[8 + j] = kp.sy[j];
    }
  }
  __syncthreads();  // single wave: orders the table write before the reads
  const double* Xw = pr.Xw + (size_t)f * M * 3;
  const double* obs = pr.obs + (size_t)f * M * 3;
  const int32_t* octave = pr.oct + (size_t)f * M;
  uint8_t* level = pr.outlier + (size_t)f * M;  // is_outlier_ <=> level 1
  double* chi2_e = pr.chi2 + (size_t)f * M;
  double* const pose_io = pr.pose_io;
  int32_t* const ninl_f = pr.ninlier + (size_t)f * pr.nin_stride;
  // an edge's final flag (on-chip shapes): into the problem's own array, or through src_of into the problem it was compacted from
  auto put_flag = [&](int idx, uint8_t v) {
    if (pr.src_of) pr.outlier_dst[(size_t)f * pr.M_dst + pr.src_of[(size_t)f * M + idx]] = v;
    else level[idx] = v;
  };

  // graph construction: count edges, clear outlier flags (tracking_opt.cpp:60-137)
  double cnt = 0.0;
  EdgeRegs E;
  if (REGS) {
#pragma unroll
    for (int i = 0; i < pose_slots<REGS>(); ++i) {
      const int e = pose_edge_index<REGS>(S, i);
      E.oc[i] = (e >= 0 && e < M) ? octave[e] : -1;
      E.lv[i] = 0;  // is_outlier_[i] is reset only where mappoints_[i] exists (:63-69): written back for those only
      E.c2[i] = 0.0;
      if (E.oc[i] >= 0) {
        if (!(obs[(size_t)e * 3 + 2] < 0)) E.oc[i] |= 16;  // stereo
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double on = fma(obs[(size_t)e * 3 + j], j == 1 ? kp.ify : kp.ifx, j == 1 ? kp.ncy : kp.ncx);
          if (REGS == 1 || REGS == 3) {
            E.X[i][j] = Xw[(size_t)e * 3 + j];
            E.O[i][j] = on;
          } else {
            xo[(i * 6 + j) * es + lane] = Xw[(size_t)e * 3 + j];
            xo[(i * 6 + 3 + j) * es + lane] = on;
          }
        }
        cnt += 1.0;
      } else if (REGS == 1 || REGS == 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) E.X[i][j] = E.O[i][j] = 0.0;
      } else {  // (an absent edge is evaluated with zero weights: its coordinates must be finite)
#pragma unroll
        for (int j = 0; j < 6; ++j) xo[(i * 6 + j) * es + lane] = 0.0;
      }
    }
  } else {
    for (int e = e0; e < M; e += es) {
      if (octave[e] >= 0) {  // is_outlier_[i] is reset only where mappoints_[i] exists (:63-69); the flags of the
        level[e] = 0;        // other features stay the caller's
        cnt += 1.0;
      }
    }
  }
  const int n_init = (int)block_total1(cnt, part);
  if (n_init < 3) {  // :139-140
    if (REGS) {  // the on-chip shapes keep the flags in registers: the reset of :63-69 happened before this return
#pragma unroll
      for (int i = 0; i < pose_slots<REGS>(); ++i) {
        if (E.oc[i] >= 0) put_flag(pose_edge_index<REGS>(S, i), 0);
      }
    }
    if (lane == 0) *ninl_f = 0;
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
  bool robust = true;
  int nbad = 0;
#ifdef GL_POSE_PROF
  long long g_pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long tk0 = clock64();
#endif
#pragma unroll 1
  for (int round = 0; round < 4; ++round) {
    P = P0;  // vertex_se3->setEstimate(curr_frame_->getTcw())  (:152)
    cnt = 0.0;
    __syncthreads();  // level[] of the previous round's gating is read by other threads below
    if (REGS) {
#pragma unroll
      for (int i = 0; i < pose_slots<REGS>(); ++i)
        if (E.oc[i] >= 0 && E.lv[i] == 0) cnt += 1.0;
    } else {
      for (int e = e0; e < M; e += es)
        if (octave[e] >= 0 && level[e] == 0) cnt += 1.0;
    }
    const int nactive = (int)block_total1(cnt, part);
    if (nactive > 0) {  // optimize(10); returns -1 untouched when nothing is active
      if (REGS) pose_eval_regs<REGS>(kp, s2tab, P, robust, G, S, E, xo, red, H POSE_PARG);
      else pose_eval(kp, s2tab, P, robust, G, S, M, Xw, obs, octave, level, chi2_e, red, H);
      double currentChi = uni(H[27]);
      bool sys_valid = true;
      double lambda = 0.0, ni = 2.0;
#pragma unroll 1
      for (int it = 0; it < 10; ++it) {
        if (!sys_valid) {  // computeActiveErrors + buildSystem at the (restored) estimate
          if (REGS) pose_eval_regs<REGS>(kp, s2tab, P, robust, G, S, E, xo, red, H POSE_PARG);
          else pose_eval(kp, s2tab, P, robust, G, S, M, Xw, obs, octave, level, chi2_e, red, H);
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
          POSE_PT(ts0);
          // (eight-wave instances: letting only the waves 0..3 solve and handing step + trial pose to the others through LDS was built and
          //  measured - the first barrier's wait halves, but the extra barrier and 28 - 81 spilled registers cost more: 14.0 k against
          //  13.3 k ticks per pass at 1 200 edges)
          const bool ok2 = GL_POSE_SOLVE_BLOCKED ? ldlt6_blocked_pos(H, H + 21, lambda, dx) : ldlt6_packed_pos(H, H + 21, lambda, dx);
          POSE_PT(ts1);
          POSE_PADD(4, ts0, ts1);  // 6 x 6 solve
          PoseRt Pn = P;
          if (ok2) Pn = rt_update(P, dx);
          POSE_PT(ts2);
          POSE_PADD(5, ts1, ts2);  // exp(dx) P
          double tempChi;
          if (ok2) {
            if (REGS) pose_eval_regs<REGS>(kp, s2tab, Pn, robust, G, S, E, xo, red, Hn POSE_PARG);
            else pose_eval(kp, s2tab, Pn, robust, G, S, M, Xw, obs, octave, level, chi2_e, red, Hn);
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
            double* const Hs = H;
            H = Hn;
            Hn = Hs;
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
    __syncthreads();  // chi2_e[] of the last evaluation was written by other threads (edge -> thread maps differ)
    if (REGS) {
#pragma unroll
      for (int i = 0; i < pose_slots<REGS>(); ++i) {
        if (E.oc[i] < 0) continue;
        double X[3], O[3];
        edge_xo<REGS ? REGS : 1>(E, xo, i, X, O);
        const bool stereo = (E.oc[i] & 16) != 0;
        const double c2 = E.lv[i] != 0 ? pose_edge_chi2(kp, s2tab, P.R, P.t, X, O, E.oc[i] & 15, stereo) : E.c2[i];
        const float thr = stereo ? 7.815f : 5.991f;
        const bool bad = (float)c2 > thr;
        E.lv[i] = bad ? 1 : 0;
        if (bad) cnt += 1.0;
      }
    }
    for (int e = e0; e < M && !REGS; e += es) {
      const int oc = octave[e];
      if (oc < 0) continue;
      double c2;
      const bool stereo = !(obs[(size_t)e * 3 + 2] < 0);
      if (level[e] != 0) {
        const double on[3] = {fma(obs[(size_t)e * 3 + 0], kp.ifx, kp.ncx), fma(obs[(size_t)e * 3 + 1], kp.ify, kp.ncy),
                              fma(obs[(size_t)e * 3 + 2], kp.ifx, kp.ncx)};
        c2 = pose_edge_chi2(kp, s2tab, P.R, P.t, Xw + (size_t)e * 3, on, oc, stereo);
      } else {
        c2 = chi2_e[e];
      }
      const float thr = stereo ? 7.815f : 5.991f;
      const bool bad = (float)c2 > thr;
      level[e] = bad ? 1 : 0;
      if (bad) cnt += 1.0;
    }
    nbad = (int)block_total1(cnt, part);
    if (round == 2) robust = false;  // e->setRobustKernel(0) at it == 2
    if (n_init < 10) break;          // optimizer.edges().size() < 10
  }
  if (REGS) {
#pragma unroll
    for (int i = 0; i < pose_slots<REGS>(); ++i)
      if (E.oc[i] >= 0) put_flag(pose_edge_index<REGS>(S, i), (uint8_t)E.lv[i]);
  }
  if (lane == 0) {
    SE3 T;
    T.r = qfromR(P.R);
    T.t[0] = P.t[0];
    T.t[1] = P.t[1];
    T.t[2] = P.t[2];
    normalize_rotation(T);
    se3_store(T, pose_io + (size_t)f * 7);
    *ninl_f = n_init - nbad;
#ifdef GL_POSE_PROF
    if (REGS) {  // {entry barrier, edges, reduce-scatter, barriers + blocks, solve, pose update, kernel total}; ninlier = passes
      g_pf[7] = clock64() - tk0;
      for (int i = 0; i < 6; ++i) pose_io[(size_t)f * 7 + i] = (double)g_pf[i];
      pose_io[(size_t)f * 7 + 6] = (double)g_pf[7];
      *ninl_f = (int)g_pf[6];
    }
#endif
  }
}

template <int NW, int REGS>
__global__ __launch_bounds__(64 * NW, NW == 1 ? GL_POSE_WPS : (NW == 4 && REGS == 2) ? 2 : (NW + 3) / 4) void k_optimize_current_pose(PoseKParams kp, int B, PoseProb pr) {
  if ((int)blockIdx.x < B) pose_frame<NW, REGS>(kp, (int)blockIdx.x, pr);
}
// ---- compacted problems (round 6; gl_pose_compact.hpp) ---------------------------------------------------------------------------------
struct PlainSrc {  // edge i of a caller's problem
  const double* Xw;
  const double* obs;
  const int32_t* oct;
  __device__ int octave(int i) const { return oct[i]; }
  __device__ void load(int i, double* X, double* O) const {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      X[j] = Xw[(size_t)i * 3 + j];
      O[j] = obs[(size_t)i * 3 + j];
    }
  }
};
__global__ __launch_bounds__(256) void k_pose_compact(int B, int M, const double* __restrict__ Xw, const double* __restrict__ obs,
                                                     const int32_t* __restrict__ oct, gl::PoseCompacted pc) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const PlainSrc src = {Xw + (size_t)b * M * 3, obs + (size_t)b * M * 3, oct + (size_t)b * M};
  gl::pose_compact_frame<256>(src, b, M, pc);
}
// the flags of compacted problems whose launch shape keeps them in memory (large batches), back to the caller's slots
__global__ __launch_bounds__(256) void k_pose_scatter(int B, int M, int MC, const int32_t* __restrict__ slot_of, const int32_t* __restrict__ ovf,
                                                     const uint8_t* __restrict__ outl_c, uint8_t* __restrict__ outlier) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)B * M) return;
  const size_t b = g / M;
  const int s_ = slot_of[g];
  if (!ovf[b] && s_ >= 0) outlier[g] = outl_c[b * MC + s_];  // (the flags of the slots without an edge stay the caller's)
}

PoseKParams pose_kparams(const gl_camera* cam, const gl_params* prm) {
  PoseKParams kp;
  kp.ifx = 1.0 / cam->fx;
  kp.ify = 1.0 / cam->fy;
  kp.ncx = -cam->cx / cam->fx;
  kp.ncy = -cam->cy / cam->fy;
  kp.bn = cam->bf / cam->fx;
  for (int i = 0; i < 8; ++i) {
    kp.sx[i] = (double)prm->sigma2_inv[i] * (cam->fx * cam->fx);
    kp.sy[i] = (double)prm->sigma2_inv[i] * (cam->fy * cam->fy);
  }
  kp.delta_mono = (double)(float)sqrt(5.991);    // const float delta_mono = sqrt(5.991)   (:57)
  kp.delta_stereo = (double)(float)sqrt(7.815);  // const float delta_stereo = sqrt(7.815) (:58)
  return kp;
}

// the launch shape of B problems of stride M (tools/pose_ab.py, tools/latency.py; the sums are built in the same order whatever the
// shape, so it never shows in the results; option pose_waves (1 | 4 | 8) forces the cap, pose_regs = 0 the global-memory variants):
//  * a wave per group of <= 256 edges with the frame's edges ON CHIP whenever that fills the CU's wave slots: frames of
//    <= 4 groups - in registers (424 VGPRs, one workgroup per CU) while every frame gets a CU of its own, coordinates in
//    LDS (two workgroups per CU) beyond; frames of 5 - 8 groups: coordinates in LDS, eight waves.  One frame of 1 000
//    edges 0.26 ms (0.40 from global memory, 1.2 ms on one wave); 4 096 full frames of 1 000 edges 3.2 ms (4.9 on one
//    wave each), of 2 000 edges 7.0 ms (11.4), of 300 edges 1.49 ms (1.75);
//  * one wave per frame, edges from global memory, 12 frames per CU in flight and no barrier, for the large batches whose
//    wave-per-group shape would leave wave slots empty: 3 groups (6 of 8 slots: 2.7 vs 2.5 ms ragged), 5 - 7 groups
//    (config 3, 2 149 frames of up to 1 200 edges: 0.78 M frames/s against 0.74 M);
//  * five groups at the frame-at-a-time caller (every frame has a CU): four waves, five edges per thread (REGS == 3).
struct PoseShape {
  int G, S, nw, NWC, REGS;
  size_t lds;
};
PoseShape pose_shape(const gl::Ctx* c, int B, int M) {
  PoseShape h;
  // the canonical summation order of a frame of stride M (see group_totals28): G groups of S <= 4 chunks of 64 edges
  const int nch = std::max(1, (M + 63) / 64);
  h.G = (nch + 3) / 4;
  h.S = (nch + h.G - 1) / h.G;
  const bool big = B > 1536;
  int nw = std::min(h.G, 8);
  if (big && !(h.G <= 4 && h.G != 3) && h.G != 8) nw = 1;
  if (c->opt.pose_waves > 0) nw = std::min(h.G, (int)c->opt.pose_waves >= 8 ? 8 : (int)c->opt.pose_waves >= 4 ? 4 : 1);
  h.lds = (size_t)h.G * 32 * sizeof(double);
  const bool on_chip = nw > 1 && nw == h.G && c->opt.pose_regs != 0;  // a wave per group: the frame's edges stay on chip
  const bool five = h.G == 5 && on_chip && B <= c->ncu && c->opt.pose_waves <= 0;
  if (five) {
    nw = 4;
    h.lds += (size_t)28 * 256 * sizeof(double);
    h.NWC = 4;
    h.REGS = 3;
  } else if (nw > 4) {
    h.NWC = 8;
    h.REGS = on_chip ? 2 : 0;
    if (on_chip) h.lds += (size_t)24 * 64 * nw * sizeof(double);
  } else if (nw > 1) {
    h.NWC = 4;
    h.REGS = on_chip ? (B > c->ncu ? 2 : 1) : 0;
    if (h.REGS == 2) h.lds += (size_t)24 * 64 * nw * sizeof(double);
  } else {
    h.NWC = 1;
    h.REGS = 0;
  }
  h.nw = nw;
  return h;
}
PoseProb pose_prob(const PoseShape& h, int M, double* pose, const double* Xw, const double* obs, const int32_t* oct, uint8_t* outlier, int32_t* ninlier,
                   int nin_stride, void* chi2) {
  PoseProb p;
  p.M = M;
  p.G = h.G;
  p.S = h.S;
  p.pose_io = pose;
  p.Xw = Xw;
  p.obs = obs;
  p.oct = oct;
  p.outlier = outlier;
  p.ninlier = ninlier;
  p.nin_stride = nin_stride;
  p.chi2 = (double*)chi2;
  p.src_of = nullptr;
  p.outlier_dst = nullptr;
  p.M_dst = 0;
  p.skip = nullptr;
  p.skip_if = 0;
  return p;
}
// the launch proper
int pose_launch(gl::Ctx* c, const PoseKParams& kp, int B, const PoseShape& h, const PoseProb& p) {
  gl::TimerScope ts(c, GL_TIMER_REFINE_POSE);
#define GL_POSE_LAUNCH(NWC, REGS)                                                                              \
  do {                                                                                                         \
    GL_HIP(gl::ensure_dynamic_lds(c, (const void*)k_optimize_current_pose<NWC, REGS>, h.lds));                 \
    k_optimize_current_pose<NWC, REGS><<<B, 64 * h.nw, h.lds, c->stream>>>(kp, B, p);                          \
  } while (0)
  if (h.NWC == 4 && h.REGS == 3) GL_POSE_LAUNCH(4, 3);
  else if (h.NWC == 8 && h.REGS == 2) GL_POSE_LAUNCH(8, 2);
  else if (h.NWC == 8) GL_POSE_LAUNCH(8, 0);
  else if (h.NWC == 4 && h.REGS == 2) GL_POSE_LAUNCH(4, 2);
  else if (h.NWC == 4 && h.REGS == 1) GL_POSE_LAUNCH(4, 1);
  else if (h.NWC == 4) GL_POSE_LAUNCH(4, 0);
  else GL_POSE_LAUNCH(1, 0);
#undef GL_POSE_LAUNCH
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // namespace

// gl_optimize_current_pose without the compaction
int gl::optimize_current_pose_plain(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M, double* pose_dev, const double* Xw_dev,
                                    const double* obs_dev, const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev, int nin_stride) {
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  void* scratch = nullptr;
  const size_t chi_bytes = (((size_t)B * (M > 0 ? M : 1) * sizeof(double) + 63) / 64) * 64;
  const int rc = gl::ctx_scratch(c, chi_bytes, &scratch);
  if (rc != GL_OK) return rc;
  const PoseShape h = pose_shape(c, B, M);
  return pose_launch(c, pose_kparams(cam, prm), B, h, pose_prob(h, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev, ninlier_dev, nin_stride, scratch));
}

// B problems of M slots whose compacted form `pc` exists (k_pose_compact, or the tracked-frame chain's gather): the compacted problem of
// every frame whose edges fitted, the full-stride one of the others; flags into outlier_dev (stride M), inlier counts into
// ninlier_dev[f * nin_stride].  Launches: the compacted problems', the full-stride problems' where M > MC (a frame's workgroup returns at
// once in one of the two) and - where the compacted problems' shape keeps its flags in memory - k_pose_scatter.
int gl::pose_compacted_launch(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M, double* pose_dev, const double* Xw_dev,
                              const double* obs_dev, const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev, int nin_stride,
                              const gl::PoseCompacted& pc) {
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  const int MC = pc.MC;
  const bool can_overflow = M > MC;
  void *chi_c = pc.chi_c, *chi_f = pc.chi_f;
  if (!chi_c || (can_overflow && !chi_f)) {
    auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
    void* scratch = nullptr;
    const int rc = gl::ctx_scratch(c, up((size_t)B * MC * 8) + (can_overflow ? up((size_t)B * M * 8) : 0), &scratch);
    if (rc != GL_OK) return rc;
    chi_c = scratch;
    chi_f = can_overflow ? (char*)scratch + up((size_t)B * MC * 8) : nullptr;
  }
  const PoseKParams kp = pose_kparams(cam, prm);
  const PoseShape hc = pose_shape(c, B, MC);
  PoseProb p = pose_prob(hc, MC, pose_dev, pc.Xw_c, pc.obs_c, pc.oct_c, pc.outl_c, ninlier_dev, nin_stride, chi_c);
  p.skip = pc.ovf;
  p.skip_if = 1;
  const bool direct = hc.REGS != 0;  // the on-chip shapes put the flags where they belong themselves
  if (direct) {
    p.src_of = pc.src_of;
    p.outlier_dst = outlier_dev;
    p.M_dst = M;
  }
  if (!can_overflow) {
    const int rc = pose_launch(c, kp, B, hc, p);
    if (rc != GL_OK) return rc;
  } else {
    const PoseShape hf = pose_shape(c, B, M);
    PoseProb pf = pose_prob(hf, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev, ninlier_dev, nin_stride, chi_f);
    pf.skip = pc.ovf;
    pf.skip_if = 0;
    // (both problems of a frame in ONE launch - workgroup f the compacted, B + f the full-stride one - was built and measured: the
    //  kernel that holds both bodies is 9 us slower per call than the launch it saves, 245 spilled SGPRs; profiles/r6_chain_fused_ab.txt)
    int rc = pose_launch(c, kp, B, hc, p);
    if (rc != GL_OK) return rc;
    rc = pose_launch(c, kp, B, hf, pf);
    if (rc != GL_OK) return rc;
  }
  if (!direct) {
    const size_t nm = (size_t)B * M;
    k_pose_scatter<<<(unsigned)((nm + 255) / 256), 256, 0, c->stream>>>(B, M, MC, pc.slot_of, pc.ovf, pc.outl_c, outlier_dev);
    GL_HIP(hipGetLastError());
  }
  return GL_OK;
}

extern "C" int gl_optimize_current_pose(gl_ctx_t* ctx, const gl_camera* cam, const gl_params* prm, int B, int M,
                                        double* pose_dev, const double* Xw_dev, const double* obs_dev,
                                        const int32_t* octave_dev, uint8_t* outlier_dev, int32_t* ninlier_dev) {
  GL_REQUIRE(ctx && cam && prm, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && M >= 0, "bad B / M");
  GL_REQUIRE(pose_dev && ninlier_dev && (M == 0 || (Xw_dev && obs_dev && octave_dev && outlier_dev)), "null buffer");
  GL_REQUIRE(M <= 64 * 4 * 512, "M above 131 072 edges per frame");
  gl::Ctx* c = gl::C(ctx);
  // option pose_compact: -1 (default) problems of more than 1 024 slots - the reference's frame: one slot per feature, 1 200 - are
  // compacted to 1 024 where their edges fit (0.31 -> 0.21 ms for one frame of 420 edges); 1: every problem of more than 256 slots
  // (pays when at most about half of the slots hold an edge); 0: never.  Decisions and tolerances as ever; the bits of a frame are a
  // function of the frame and this option, never of the batch.
  const int MC = gl::pose_compact_stride((int)c->opt.pose_compact, M, (int)c->opt.pose_compact_cap);
  if (MC == 0) return gl::optimize_current_pose_plain(ctx, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev, ninlier_dev, 1);
  GL_HIP(hipSetDevice(c->device));
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  const bool can_overflow = M > MC;
  const size_t chi_c_bytes = up((size_t)B * MC * 8), chi_f_bytes = can_overflow ? up((size_t)B * M * 8) : 0;
  void* scratch = nullptr;
  const int rc = gl::ctx_scratch(c, chi_c_bytes + chi_f_bytes + gl::pose_compacted_bytes(B, M, MC), &scratch);
  if (rc != GL_OK) return rc;
  gl::PoseCompacted pc;
  gl::pose_compacted_place((char*)scratch + chi_c_bytes + chi_f_bytes, B, M, MC, &pc);
  pc.chi_c = scratch;
  pc.chi_f = can_overflow ? (char*)scratch + chi_c_bytes : nullptr;
  k_pose_compact<<<B, 256, 0, c->stream>>>(B, M, Xw_dev, obs_dev, octave_dev, pc);
  GL_HIP(hipGetLastError());
  return gl::pose_compacted_launch(ctx, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, outlier_dev, ninlier_dev, 1, pc);
}
