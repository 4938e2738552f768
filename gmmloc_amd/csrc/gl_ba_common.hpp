// Shared device helpers of the bundle-adjustment kernels (gl_ba.hip, gl_ba_gen.hip).
#pragma once
#include "gl_device.hpp"
#include "gl_internal.hpp"

#pragma clang fp contract(fast)

namespace glba {
using namespace gld;

struct BaK {
  double fx, fy, cx, cy, bf;
  double s2inv[8];
  double delta_mono, delta_stereo;
  double ba_lambda2;   // (double) loc::ba_lambda2
  double str_thresh;   // (double)(tri_str_thresh * ba_lambda2) in float
  double gate_chi2;    // association gate of gl_track_frames (9.0), < 0 = keep all
  int first_as_prior;
};

constexpr int T_BA = 256;
constexpr int NW_BA = T_BA / 64;

struct PtLin {
  double q[3];
  double A[6];   // sym: 00 01 02 11 12 22
  double a[3];
  double Hc[6];
  double bc[3];
  double chi_r, rho0_r, chi_g;
  bool act_r, act_g;
};

GL_DEV void sym3_mul_vec(const double* S, const double* v, double* o) {
  o[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
  o[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
  o[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}
// inverse of a symmetric 3x3 by cofactors (Eigen compute_inverse<3>); result symmetric
GL_DEV void sym3_inv(const double* S, double* I) {
  const double c00 = S[3] * S[5] - S[4] * S[4];
  const double c01 = S[2] * S[4] - S[1] * S[5];
  const double c02 = S[1] * S[4] - S[2] * S[3];
  const double det = S[0] * c00 + S[1] * c01 + S[2] * c02;
  const double id = 1.0 / det;
  I[0] = c00 * id;
  I[1] = c01 * id;
  I[2] = c02 * id;
  I[3] = (S[0] * S[5] - S[2] * S[2]) * id;
  I[4] = (S[1] * S[2] - S[0] * S[4]) * id;
  I[5] = (S[0] * S[3] - S[1] * S[1]) * id;
}
// D = L Delta L^T of a symmetric positive definite 3x3 (unpivoted), f = {l10, l20, l21, 1/d0, 1/d1, 1/d2}, and the solve with
// the factors.  The single-pose refine kernels use these for the damped point blocks instead of the cofactor inverse: a point
// block may be arbitrarily badly scaled (a point that ran away along its plane: eigenvalues 400 / 7e-3 / lambda), the
// cofactor determinant is rounding noise there, the factorisation is backward stable (gl_ba_fast_impl.hpp: ldl3_factor_fast).
GL_DEV void ldl3_factor(const double* D, double* f) {
  const double i0 = 1.0 / D[0];
  const double l1 = D[1] * i0, l2 = D[2] * i0;
  const double d1 = fma(-l1, D[1], D[3]);
  const double e = fma(-l1, D[2], D[4]);
  const double i1 = 1.0 / d1;
  const double l3 = e * i1;
  const double d2 = fma(-l3, e, fma(-l2, D[2], D[5]));
  f[0] = l1;
  f[1] = l2;
  f[2] = l3;
  f[3] = i0;
  f[4] = i1;
  f[5] = 1.0 / d2;
}
GL_DEV void ldl3_solve(const double* f, const double* b, double* x) {
  const double y1 = fma(-f[0], b[0], b[1]);
  const double y2 = fma(-f[2], y1, fma(-f[1], b[0], b[2]));
  x[2] = y2 * f[5];
  x[1] = fma(-f[2], x[2], y1 * f[4]);
  x[0] = fma(-f[1], x[2], fma(-f[0], x[1], b[0] * f[3]));
}
// P = X * Y for symmetric X, Y (full 3x3 row-major result)
GL_DEV void sym3_mul(const double* X, const double* Y, double* P) {
  const double x[9] = {X[0], X[1], X[2], X[1], X[3], X[4], X[2], X[4], X[5]};
  const double y[9] = {Y[0], Y[1], Y[2], Y[1], Y[3], Y[4], Y[2], Y[4], Y[5]};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) P[i * 3 + j] = x[i * 3] * y[j] + x[i * 3 + 1] * y[3 + j] + x[i * 3 + 2] * y[6 + j];
}

// reprojection residual / chi2 at camera point q (un-robustified chi2 = s |e|^2)
GL_DEV double reproj_err(const BaK& k, const double* q, const double* ob, bool stereo, double s, double* e,
                         double& iz) {
  iz = 1.0 / q[2];
  const double pu = q[0] * iz * k.fx + k.cx, pv = q[1] * iz * k.fy + k.cy;
  e[0] = ob[0] - pu;
  e[1] = ob[1] - pv;
  e[2] = stereo ? (ob[2] - (pu - k.bf * iz)) : 0.0;
  return e[0] * (s * e[0]) + e[1] * (s * e[1]) + e[2] * (s * e[2]);
}

struct GmmRef {  // per-point association data (world frame)
  bool has, deg;
  double n[3];   // deg: plane normal (axis_.col(0))
  double L[9];   // non-deg: sqrt_info_ (lower)
  double mu[3];
};

GL_DEV void load_gmm(int a, const double* __restrict__ axis, const double* __restrict__ rec12,
                     const double* __restrict__ sqrt_info, const uint8_t* __restrict__ flags, GmmRef& g) {
  g.has = a >= 0;
  g.deg = false;
  if (!g.has) return;
  g.deg = flags[a] & 1;
#pragma unroll
  for (int i = 0; i < 3; ++i) g.mu[i] = rec12[(size_t)a * 12 + i];
  if (g.deg) {
#pragma unroll
    for (int i = 0; i < 3; ++i) g.n[i] = axis[(size_t)a * 9 + i * 3];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) g.L[i] = sqrt_info[(size_t)a * 9 + i];
  }
}

// un-robustified chi2 of the GMM edge at world point p
GL_DEV double gmm_chi2(const BaK& k, const GmmRef& g, const double* p) {
  const double d[3] = {p[0] - g.mu[0], p[1] - g.mu[1], p[2] - g.mu[2]};
  if (g.deg) {
    const double e = g.n[0] * d[0] + g.n[1] * d[1] + g.n[2] * d[2];
    return e * (k.ba_lambda2 * e);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double e = g.L[0 * 3 + i] * d[0] + g.L[1 * 3 + i] * d[1] + g.L[2 * 3 + i] * d[2];  // L^T d
    s += e * e;
  }
  return s;
}

// linearise one point at (R, t, p)
GL_DEV void lin_point(const BaK& k, const double* R, const double* t, const double* p, const double* ob, int oc,
                      const GmmRef& g, bool act_r, bool act_g, bool robust, PtLin& o) {
  o.act_r = act_r;
  o.act_g = act_g && g.has;
#pragma unroll
  for (int i = 0; i < 3; ++i) o.q[i] = R[i * 3] * p[0] + R[i * 3 + 1] * p[1] + R[i * 3 + 2] * p[2] + t[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    o.A[i] = 0.0;
    o.Hc[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.a[i] = 0.0;
    o.bc[i] = 0.0;
  }
  o.chi_r = o.rho0_r = o.chi_g = 0.0;
  if (o.act_r) {
    const bool stereo = !(ob[2] < 0);
    const double s = k.s2inv[oc];
    double e[3], iz;
    o.chi_r = reproj_err(k, o.q, ob, stereo, s, e, iz);
    double rho1 = 1.0;
    o.rho0_r = o.chi_r;
    if (robust) huber(o.chi_r, stereo ? k.delta_stereo : k.delta_mono, o.rho0_r, rho1);
    const double w = rho1 * s;
    const double iz2 = iz * iz;
    const double al = k.fx * iz, ga = k.fy * iz;
    const double b0 = -k.fx * o.q[0] * iz2, b1 = -k.fy * o.q[1] * iz2;
    const double b2 = b0 + k.bf * iz2;
    const double sb = stereo ? 1.0 : 0.0;
    o.A[0] = w * (al * al + sb * al * al);
    o.A[1] = 0.0;
    o.A[2] = w * (al * b0 + sb * al * b2);
    o.A[3] = w * ga * ga;
    o.A[4] = w * ga * b1;
    o.A[5] = w * (b0 * b0 + b1 * b1 + sb * b2 * b2);
    o.a[0] = w * al * (e[0] + sb * e[2]);
    o.a[1] = w * ga * e[1];
    o.a[2] = w * (b0 * e[0] + b1 * e[1] + sb * b2 * e[2]);
  }
  if (o.act_g) {
    const double d[3] = {p[0] - g.mu[0], p[1] - g.mu[1], p[2] - g.mu[2]};
    if (g.deg) {
      const double eg = g.n[0] * d[0] + g.n[1] * d[1] + g.n[2] * d[2];
      o.chi_g = eg * (k.ba_lambda2 * eg);
      double nc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) nc[i] = R[i * 3] * g.n[0] + R[i * 3 + 1] * g.n[1] + R[i * 3 + 2] * g.n[2];
      const double l = k.ba_lambda2;
      o.Hc[0] = l * nc[0] * nc[0];
      o.Hc[1] = l * nc[0] * nc[1];
      o.Hc[2] = l * nc[0] * nc[2];
      o.Hc[3] = l * nc[1] * nc[1];
      o.Hc[4] = l * nc[1] * nc[2];
      o.Hc[5] = l * nc[2] * nc[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) o.bc[i] = -l * eg * nc[i];
    } else {
      // e = L^T d, J = L^T: Hg = L L^T, bg = -L e
      double e[3], bg[3], Hg[9], RH[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) e[i] = g.L[0 * 3 + i] * d[0] + g.L[1 * 3 + i] * d[1] + g.L[2 * 3 + i] * d[2];
      o.chi_g = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) bg[i] = -(g.L[i * 3] * e[0] + g.L[i * 3 + 1] * e[1] + g.L[i * 3 + 2] * e[2]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Hg[i * 3 + j] = g.L[i * 3] * g.L[j * 3] + g.L[i * 3 + 1] * g.L[j * 3 + 1] + g.L[i * 3 + 2] * g.L[j * 3 + 2];
      mm3(R, Hg, RH);
      // Hc = RH * R^T (symmetric)
      o.Hc[0] = RH[0] * R[0] + RH[1] * R[1] + RH[2] * R[2];
      o.Hc[1] = RH[0] * R[3] + RH[1] * R[4] + RH[2] * R[5];
      o.Hc[2] = RH[0] * R[6] + RH[1] * R[7] + RH[2] * R[8];
      o.Hc[3] = RH[3] * R[3] + RH[4] * R[4] + RH[5] * R[5];
      o.Hc[4] = RH[3] * R[6] + RH[4] * R[7] + RH[5] * R[8];
      o.Hc[5] = RH[6] * R[6] + RH[7] * R[7] + RH[8] * R[8];
#pragma unroll
      for (int i = 0; i < 3; ++i) o.bc[i] = R[i * 3] * bg[0] + R[i * 3 + 1] * bg[1] + R[i * 3 + 2] * bg[2];
    }
  }
}

GL_DEV void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// EdgeSE3QuatPrior (factors.cpp:19-53): adds J^T Omega J / -J^T Omega e, returns chi2
GL_DEV double prior_terms(const SE3& inv_meas, const SE3& T, bool build, double* H /*6x6 +=*/, double* b /*+=*/) {
  const SE3 d = se3_mul(inv_meas, T);
  double e[6];
  se3_log(d, e);
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0));
  const double st = 1.0 / (0.01 * 0.01);
  const double om[6] = {sr, sr, sr, st, st, st};
  double chi = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) chi += e[i] * om[i] * e[i];
  if (build) {
    double Jr[36], Adj[36], J[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) Jr[i] = 0.0;
    double ps[9], ls[9];
    skew(e, ps);
    skew(e + 3, ls);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Jr[i * 6 + j] = 0.5 * ps[i * 3 + j];
        Jr[(i + 3) * 6 + j + 3] = 0.5 * ps[i * 3 + j];
        Jr[i * 6 + j + 3] = 0.5 * ls[i * 3 + j];
      }
#pragma unroll
    for (int i = 0; i < 6; ++i) Jr[i * 6 + i] += 1.0;
    const SE3 Ti = se3_inverse(T);
    double R[9], S[9], SR[9];
    qtoR(Ti.r, R);
    skew(Ti.t, S);
    mm3(S, R, SR);
#pragma unroll
    for (int i = 0; i < 36; ++i) Adj[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Adj[i * 6 + j] = R[i * 3 + j];
        Adj[(i + 3) * 6 + j + 3] = R[i * 3 + j];
        Adj[(i + 3) * 6 + j] = SR[i * 3 + j];
      }
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0.0;
        for (int l = 0; l < 6; ++l) s += Jr[i * 6 + l] * Adj[l * 6 + j];
        J[i * 6 + j] = s;
      }
    for (int i = 0; i < 6; ++i) {
      double s = 0.0;
      for (int r = 0; r < 6; ++r) s += J[r * 6 + i] * om[r] * e[r];
      b[i] -= s;
      for (int j = 0; j < 6; ++j) {
        double h = 0.0;
        for (int r = 0; r < 6; ++r) h += J[r * 6 + i] * om[r] * J[r * 6 + j];
        H[i * 6 + j] += h;
      }
    }
  }
  return chi;
}


GL_DEV double block_max(double v, double* lds) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmax(v, shfl_xor_f64(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = lds[0];
#pragma unroll
  for (int w = 1; w < NW_BA; ++w) m = fmax(m, lds[w]);
  return m;
}

struct GmmDev {
  const double* rec12;
  const double* axis;
  const double* sqrt_info;
  const double* hgw;
  const uint8_t* flags;
  const double* plane4;
};

inline BaK make_bak(const gl_camera* cam, const gl_params* prm, double gate) {
  BaK k;
  k.fx = cam->fx;
  k.fy = cam->fy;
  k.cx = cam->cx;
  k.cy = cam->cy;
  k.bf = cam->bf;
  for (int i = 0; i < 8; ++i) k.s2inv[i] = (double)prm->sigma2_inv[i];
  k.delta_mono = (double)(float)sqrt(5.991);    // thHuberMono   (localization_opt.cpp:629)
  k.delta_stereo = (double)(float)sqrt(7.815);  // thHuberStereo (:630)
  k.ba_lambda2 = (double)prm->ba_lambda2;
  k.str_thresh = (double)(prm->tri_str_thresh * prm->ba_lambda2);  // float product (:782)
  k.gate_chi2 = gate;
  k.first_as_prior = prm->ba_first_as_prior;
  return k;
}

}  // namespace glba
