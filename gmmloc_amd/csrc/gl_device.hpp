// Device-side fp64 algebra for the gfx950 kernels (wave64).  Written against
// the reference semantics cited at each function; NOT shared with oracle/.
// Files that need bit-reproducible results vs the oracle (gl_gmm.hip,
// gl_assoc.hip, gl_view.hip) are compiled with -ffp-contract=off and use
// explicit fma() only where the canonical evaluation order says so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_DEV __device__ __forceinline__

namespace gld {

// ---------------------------------------------------------------------------
// Canonical Mahalanobis evaluation: GaussianComponent::chi2 / MDist2
// (gaussian.cpp:65-70, gaussian.h:53-56)  (delta^T A) . delta, Eigen coefficient
// order, fused as GCC does under -O3 -march=native.  rec = mean[3], cov_inv[9].
// ---------------------------------------------------------------------------
GL_DEV double chi2_rec(const double* __restrict__ rec, double x, double y, double z) {
  const double d0 = x - rec[0], d1 = y - rec[1], d2 = z - rec[2];
  const double r0 = fma(d2, rec[3 + 6], fma(d1, rec[3 + 3], d0 * rec[3 + 0]));
  const double r1 = fma(d2, rec[3 + 7], fma(d1, rec[3 + 4], d0 * rec[3 + 1]));
  const double r2 = fma(d2, rec[3 + 8], fma(d1, rec[3 + 5], d0 * rec[3 + 2]));
  return fma(r2, d2, fma(r1, d1, r0 * d0));
}
// 2-D MDist2 (gaussian.h:123-126)
GL_DEV double mdist2_2d(const double* mean, const double* A, double u, double v) {
  const double d0 = u - mean[0], d1 = v - mean[1];
  const double r0 = fma(d1, A[2], d0 * A[0]);
  const double r1 = fma(d1, A[3], d0 * A[1]);
  return fma(r1, d1, r0 * d0);
}

// ---- 3x3 / 2x2 (row-major) -- Eigen fixed-size inverse / determinant --------
GL_DEV double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
         m[2] * (m[3] * m[7] - m[4] * m[6]);
}
GL_DEV double cof3(const double* m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
GL_DEV void inv3(const double* m, double* r) {
  const double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  const double invdet = 1.0 / det;
  r[0] = c0 * invdet;
  r[1] = c1 * invdet;
  r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet;
  r[4] = cof3(m, 1, 1) * invdet;
  r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet;
  r[7] = cof3(m, 1, 2) * invdet;
  r[8] = cof3(m, 2, 2) * invdet;
}
GL_DEV double det2(const double* m) { return m[0] * m[3] - m[2] * m[1]; }
GL_DEV void inv2(const double* m, double* r) {
  const double invdet = 1.0 / det2(m);
  r[0] = m[3] * invdet;
  r[2] = -m[2] * invdet;
  r[1] = -m[1] * invdet;
  r[3] = m[0] * invdet;
}

// symmetric eigen-decomposition, cyclic Jacobi (n = 2, 3); w ascending,
// V row-major with column c = eigenvector c.
template <int n>
GL_DEV void eig_sym(const double* Ain, double* w, double* V) {
  double A[n * n];
#pragma unroll
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = i + 1; j < n; ++j) A[i * n + j] = A[j * n + i];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off == 0.0) break;
#pragma unroll
    for (int p = 0; p < n; ++p)
#pragma unroll
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq != 0.0) {
          const double app = A[p * n + p], aqq = A[q * n + q];
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
          for (int k = 0; k < n; ++k) {
            const double akp = A[k * n + p], akq = A[k * n + q];
            A[k * n + p] = c * akp - s * akq;
            A[k * n + q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < n; ++k) {
            const double apk = A[p * n + k], aqk = A[q * n + k];
            A[p * n + k] = c * apk - s * aqk;
            A[q * n + k] = s * apk + c * aqk;
          }
          A[p * n + q] = 0.0;
          A[q * n + p] = 0.0;
#pragma unroll
          for (int k = 0; k < n; ++k) {
            const double vkp = V[k * n + p], vkq = V[k * n + q];
            V[k * n + p] = c * vkp - s * vkq;
            V[k * n + q] = s * vkp + c * vkq;
          }
        }
      }
  }
#pragma unroll
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
#pragma unroll
  for (int i = 0; i < n - 1; ++i) {
    int m = i;
#pragma unroll
    for (int j = i + 1; j < n; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      const double tw = w[i];
      w[i] = w[m];
      w[m] = tw;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        const double tv = V[k * n + i];
        V[k * n + i] = V[k * n + m];
        V[k * n + m] = tv;
      }
    }
  }
}

// lower Cholesky of SPD 3x3 (reads lower triangle) -- Eigen LLT::matrixL()
GL_DEV bool chol3_lower(const double* A, double* L) {
#pragma unroll
  for (int i = 0; i < 9; ++i) L[i] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double x = A[k * 3 + k];
#pragma unroll
    for (int j = 0; j < k; ++j) x -= L[k * 3 + j] * L[k * 3 + j];
    if (!(x > 0.0)) return false;
    x = sqrt(x);
    L[k * 3 + k] = x;
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      double s = A[i * 3 + k];
#pragma unroll
      for (int j = 0; j < k; ++j) s -= L[i * 3 + j] * L[k * 3 + j];
      L[i * 3 + k] = s / x;
    }
  }
  return true;
}

// ---- quaternion (x,y,z,w) / SE3 -- g2o::SE3Quat (se3quat.h) ----------------
struct Quat {
  double x, y, z, w;
};
struct SE3 {
  Quat r;
  double t[3];
};
GL_DEV Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
GL_DEV void qrot(const Quat& q, const double* v, double* out) {  // Eigen _transformVector
  double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux;
  uy += uy;
  uz += uz;
  out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
GL_DEV void qtoR(const Quat& q, double* R) {  // Eigen toRotationMatrix
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}
GL_DEV Quat qfromR(const double* m) {  // Eigen rotation matrix -> quaternion
  double c[4];
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    c[3] = 0.5 * t;
    t = 0.5 / t;
    c[0] = (m[7] - m[5]) * t;
    c[1] = (m[2] - m[6]) * t;
    c[2] = (m[3] - m[1]) * t;
  } else {
    // Eigen picks the largest diagonal element i and sets j = (i+1)%3, k = (j+1)%3.  Written as three
    // static-index cases: a run-time index into m[] would move the caller's whole rotation (and any
    // struct it lives in) to scratch memory, with a store on every update.
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > (i == 0 ? m[0] : m[4])) i = 2;
#define GL_QFROMR_CASE(I, J, K)                                            \
  {                                                                        \
    t = sqrt(m[I * 3 + I] - m[J * 3 + J] - m[K * 3 + K] + 1.0);            \
    c[I] = 0.5 * t;                                                        \
    t = 0.5 / t;                                                           \
    c[3] = (m[K * 3 + J] - m[J * 3 + K]) * t;                              \
    c[J] = (m[J * 3 + I] + m[I * 3 + J]) * t;                              \
    c[K] = (m[K * 3 + I] + m[I * 3 + K]) * t;                              \
  }
    if (i == 0) GL_QFROMR_CASE(0, 1, 2)
    else if (i == 1) GL_QFROMR_CASE(1, 2, 0)
    else GL_QFROMR_CASE(2, 0, 1)
#undef GL_QFROMR_CASE
  }
  return Quat{c[0], c[1], c[2], c[3]};
}
GL_DEV void normalize_rotation(SE3& T) {  // SE3Quat::normalizeRotation
  if (T.r.w < 0) {
    T.r.x = -T.r.x;
    T.r.y = -T.r.y;
    T.r.z = -T.r.z;
    T.r.w = -T.r.w;
  }
  const double n = sqrt(T.r.x * T.r.x + T.r.y * T.r.y + T.r.z * T.r.z + T.r.w * T.r.w);
  T.r.x /= n;
  T.r.y /= n;
  T.r.z /= n;
  T.r.w /= n;
}
GL_DEV SE3 se3_load(const double* p) {
  SE3 T;
  T.r = Quat{p[0], p[1], p[2], p[3]};
  T.t[0] = p[4];
  T.t[1] = p[5];
  T.t[2] = p[6];
  normalize_rotation(T);
  return T;
}
GL_DEV void se3_store(const SE3& T, double* p) {
  p[0] = T.r.x;
  p[1] = T.r.y;
  p[2] = T.r.z;
  p[3] = T.r.w;
  p[4] = T.t[0];
  p[5] = T.t[1];
  p[6] = T.t[2];
}
GL_DEV SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r = a;
  double rt[3];
  qrot(a.r, b.t, rt);
  r.t[0] += rt[0];
  r.t[1] += rt[1];
  r.t[2] += rt[2];
  r.r = qmul(a.r, b.r);
  normalize_rotation(r);
  return r;
}
GL_DEV SE3 se3_inverse(const SE3& a) {
  SE3 r;
  r.r = Quat{-a.r.x, -a.r.y, -a.r.z, a.r.w};
  const double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  qrot(r.r, nt, r.t);
  return r;
}
GL_DEV void skew(const double* v, double* S) {
  S[0] = 0;
  S[1] = -v[2];
  S[2] = v[1];
  S[3] = v[2];
  S[4] = 0;
  S[5] = -v[0];
  S[6] = -v[1];
  S[7] = v[0];
  S[8] = 0;
}
GL_DEV void mm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
// SE3Quat::exp(update), update = [omega, upsilon]
GL_DEV SE3 se3_exp(const double* u) {
  const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew(u, Om);
  mm3(Om, Om, Om2);
  double a, b, c;
  if (theta < 0.00001) {
    a = 1.0;
    b = 0.5;
    c = 1.0 / 6.0;
  } else {
    const double st = sin(theta), ct = cos(theta);
    a = st / theta;
    b = (1 - ct) / (theta * theta);
    c = (theta - st) / (theta * theta * theta);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * Om[i] + b * Om2[i];
    V[i] = I + b * Om[i] + c * Om2[i];
  }
  SE3 T;
  T.r = qfromR(R);
#pragma unroll
  for (int i = 0; i < 3; ++i) T.t[i] = V[i * 3 + 0] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
  normalize_rotation(T);
  return T;
}
// SE3Quat::log()
GL_DEV void se3_log(const SE3& T, double* res) {
  double R[9];
  qtoR(T.r, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double omega[3];
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double Om[9], Om2[9];
  double g;
  if (fabs(d) > 0.99999) {
#pragma unroll
    for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
    g = 1. / 12.;
  } else {
    const double theta = acos(d);
    const double f = theta / (2 * sqrt(1 - d * d));
#pragma unroll
    for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
    g = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  }
  skew(omega, Om);
  mm3(Om, Om, Om2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double vinv = ((i == j) ? 1.0 : 0.0) - 0.5 * Om[i * 3 + j] + g * Om2[i * 3 + j];
      s += vinv * T.t[j];
    }
    res[i] = omega[i];
    res[i + 3] = s;
  }
}

// g2o::RobustKernelHuber::robustify -> rho[0] (cost) and rho[1] (weight)
GL_DEV void huber(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho0 = e;
    rho1 = 1.0;
  } else {
    const double sqrte = sqrt(e);
    rho0 = 2 * sqrte * delta - dsqr;
    rho1 = delta / sqrte;
  }
}

// ---------------------------------------------------------------------------
// Symmetric solve by LDL^T without pivoting, N <= 6, all lanes redundantly.
// (g2o LinearSolverDense = Eigen::LDLT + isPositive(); LinearSolverEigen =
// SimplicialLDLT.)  H full row-major; returns false on a zero / (optionally)
// non-positive pivot.
// ---------------------------------------------------------------------------
template <int N>
GL_DEV bool ldlt_solve(const double* H, const double* b, double* x, bool require_positive) {
  double L[N * N], D[N], y[N];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = H[j * N + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k] * D[k];
    if (d == 0.0 || !isfinite(d)) ok = false;
    if (require_positive && !(d > 0.0)) ok = false;
    D[j] = d;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = H[i * N + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k] * D[k];
      L[i * N + j] = s / d;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] /= D[i];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
    x[i] = s;
  }
  return ok;
}

// ---------------------------------------------------------------------------
// Workgroup reduction of NV (<= 32) doubles per thread, deterministic order.
// Wave level: butterfly reduce-scatter -- at each of 5 stages a lane keeps half
// of its values and trades the other half with its xor-partner, so 32 values
// cost 16+8+4+2+1(+1) shuffles instead of 32*6.  Then one LDS slot per
// (wave, value), one __syncthreads, and every thread sums the per-wave partials
// in wave order (all threads end up with all NV totals).
// lds must hold NWAVES*32 doubles; a __syncthreads() is executed on entry so the
// buffer may be reused back-to-back.
// ---------------------------------------------------------------------------
GL_DEV double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

// Cross-lane moves for the reduce-scatter.  The in-row stages use DPP (no LDS traffic, no address
// VGPR): quad_perm xor-1 / xor-2, row_half_mirror (lane ^ 7), row_mirror (lane ^ 15); the two
// widest stages use the gfx950 v_permlane{32,16}_swap (see rs_swap_stage).
template <int CTRL>
GL_DEV double dpp_f64(double v) {
  union {
    double d;
    int i[2];
  } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xF, 0xF, false);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xF, 0xF, false);
  return b.d;
}
GL_DEV double swz16_f64(double v) {
  union {
    double d;
    int i[2];
  } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_ds_swizzle(a.i[0], 0x401F);  // bit-mask mode: and 0x1F, or 0, xor 0x10
  b.i[1] = __builtin_amdgcn_ds_swizzle(a.i[1], 0x401F);
  return b.d;
}
// one butterfly stage: keep H values, trade the other H with the partner (all indices static, so
// the value array stays in registers)
template <int H, int MODE>
GL_DEV void rs_stage(double* v, bool hi) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const double keep = hi ? v[i + H] : v[i];
    const double send = hi ? v[i] : v[i + H];
    double recv;
    if (MODE == 0) recv = dpp_f64<0xB1>(send);        // quad_perm [1,0,3,2]
    else if (MODE == 1) recv = dpp_f64<0x4E>(send);   // quad_perm [2,3,0,1]
    else if (MODE == 2) recv = dpp_f64<0x141>(send);  // row_half_mirror
    else if (MODE == 3) recv = dpp_f64<0x140>(send);  // row_mirror
    else recv = swz16_f64(send);
    v[i] = keep + recv;
  }
}
// gfx950 lane-swap stages: v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes
// 0..31 of the second, v_permlane16_swap the odd 16-lane rows of the first with the even rows of the
// second.  With a = v[i], b = v[i+H] the swap IS the select-and-exchange of a reduce-scatter stage:
// afterwards a + b holds value i in the low half / even rows and value i+H in the high half / odd
// rows -- 3 instructions per pair instead of 4 selects + 2 DPP moves + 1 add.
typedef unsigned gl_v2u __attribute__((ext_vector_type(2)));
template <int H, int WIDE>
GL_DEV void rs_swap_stage(double* v) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    union {
      double d;
      unsigned u[2];
    } a, b;
    a.d = v[i];
    b.d = v[i + H];
    gl_v2u lo, hi;
    if (WIDE == 32) {
      lo = __builtin_amdgcn_permlane32_swap(a.u[0], b.u[0], false, false);
      hi = __builtin_amdgcn_permlane32_swap(a.u[1], b.u[1], false, false);
    } else {
      lo = __builtin_amdgcn_permlane16_swap(a.u[0], b.u[0], false, false);
      hi = __builtin_amdgcn_permlane16_swap(a.u[1], b.u[1], false, false);
    }
    a.u[0] = lo.x;
    b.u[0] = lo.y;
    a.u[1] = hi.x;
    b.u[1] = hi.y;
    v[i] = a.d + b.d;
  }
}
// wave-only reduce-scatter: 32 values per lane in; out: the wave total of value `wave_slot(lane)`,
// present in the two lanes l and l ^ 15 (store from the lanes with wave_slot_owner(lane)).
// Stages: lane bit 5 (permlane32_swap), bit 4 (permlane16_swap), then DPP quad_perm xor-1 / xor-2 and
// row_half_mirror (lane ^ 7) inside a row, and a row_mirror (lane ^ 15) add to finish.  Mirror
// partners flip several lane bits at once, so those stages select on VIRTUAL bits that are equal in
// the partners of every later stage: v0 = l0^l2, v1 = l1^l2, v2 = l2^l3.
GL_DEV int wave_slot(int lane) {
  const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1;
  return (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | ((l0 ^ l2) << 2) | ((l1 ^ l2) << 1) | (l2 ^ l3);
}
GL_DEV bool wave_slot_owner(int lane) { return !(lane & 8); }
GL_DEV double wave_reduce_scatter32(double* v) {
  const int lane = threadIdx.x & 63;
  const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1;
  rs_swap_stage<16, 32>(v);
  rs_swap_stage<8, 16>(v);
  rs_stage<4, 0>(v, l0 ^ l2);
  rs_stage<2, 1>(v, l1 ^ l2);
  rs_stage<1, 2>(v, l2 ^ l3);
  return v[0] + dpp_f64<0x140>(v[0]);
}

// 16-value variant with the same lane pairings in the same order (32, 16, xor 1, xor 2, then lane ^ 7 and lane ^ 15 as full
// adds): every value meets the 64 lanes in the tree wave_reduce_scatter32 would use for it, so the totals are the same bits - for
// half the registers and instructions.  out: every lane holds the wave total of value wave_slot16(lane); store from the lanes
// with wave_slot16_owner(lane).
GL_DEV int wave_slot16(int lane) {
  const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1;
  return (((lane >> 5) & 1) << 3) | (((lane >> 4) & 1) << 2) | ((l0 ^ l2) << 1) | (l1 ^ l2);
}
GL_DEV bool wave_slot16_owner(int lane) { return !(lane & 0xC); }
GL_DEV double wave_reduce_scatter16(double* v) {
  const int lane = threadIdx.x & 63;
  const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1;
  rs_swap_stage<8, 32>(v);
  rs_swap_stage<4, 16>(v);
  rs_stage<2, 0>(v, l0 ^ l2);
  rs_stage<1, 1>(v, l1 ^ l2);
  double r = v[0];
  r = r + dpp_f64<0x141>(r);  // row_half_mirror: lane ^ 7
  r = r + dpp_f64<0x140>(r);  // row_mirror:      lane ^ 15
  return r;
}

// 8-value variant of wave_reduce_scatter32 with the same lane pairings in the same order (32, 16, 1, 2, 4, 8):
// in: v[0..7] per lane; out: every lane holds the wave total of value ((lane>>5)&1)*4 + ((lane>>4)&1)*2 + (l0^l2)
GL_DEV double wave_reduce_scatter8(double* v) {
  const int lane = threadIdx.x & 63;
  const int l0 = lane & 1, l2 = (lane >> 2) & 1;
  rs_swap_stage<4, 32>(v);
  rs_swap_stage<2, 16>(v);
  rs_stage<1, 0>(v, l0 ^ l2);
  double r = v[0];
  r = r + dpp_f64<0x4E>(r);   // quad_perm [2,3,0,1]: lane ^ 2
  r = r + dpp_f64<0x141>(r);  // row_half_mirror:     lane ^ 7
  r = r + dpp_f64<0x140>(r);  // row_mirror:          lane ^ 15
  return r;
}


template <int NV, int NWAVES>
GL_DEV void block_reduce(double* v /*[32] in, [NV] out*/, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = NV; i < 32; ++i) v[i] = 0.0;
  const double r = wave_reduce_scatter32(v);
  __syncthreads();
  if (wave_slot_owner(lane)) lds[wave * 32 + wave_slot(lane)] = r;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = lds[i];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) s += lds[w * 32 + i];
    v[i] = s;
  }
}

// ---- exchange of 32 partial sums between the NB co-resident workgroups that share one problem ---------------
// (latency shape of the structure refine: cooperative launch).  No barrier and no fence: a workgroup
// publishes its 32 sums as 64-bit words {32 bits of the value | sequence number of the reduction} with
// device-scope atomic stores, and reads everybody's words (its own included) with device-scope atomic loads until
// both halves of every value carry the current sequence number - one memory round trip when the others are
// already there, against three dependent ones (generation read, arrival, poll) + two fences for a counter
// barrier.  The partials are added in workgroup order, so every workgroup ends with the same bits.  Two buffers
// are used alternately: a workgroup can only be one reduction ahead of the slowest one (it needs that one's
// partial), so what it overwrites has been read by everybody.
// The workgroups of a problem must be co-resident for this to terminate.  A cooperative launch guarantees it but costs
// ~30 us per launch on this stack; the kernels are launched plainly and protect themselves instead: EVERY exchange has a
// time limit (option ba_rendezvous_us, default 200 us against the ~1 us a sibling's pass takes).  A workgroup whose
// siblings do not show up (the first exchange: they are queued behind another launch that is itself waiting for ITS
// siblings) or stop answering (a later one: preempted, or a word that never becomes visible) raises the problem's abort
// word and leaves; every poll watches that word.  The workgroups write their results to a STAGING area of the launch's
// scratch and count themselves done; the launcher always follows up with the one-workgroup kernel, which copies the staged
// result of a problem to the caller's buffers iff all its workgroups are done and recomputes the problem from the untouched
// inputs otherwise - same bits by construction, whenever the give-up happened.
struct Coop {
  unsigned long long* part;  // 2 buffers x NB (<= 8) workgroups x 32 values x 2 words, zero before the launch
  int NB, pb;
  unsigned seq;              // reductions so far (the same in every workgroup)
  int* ctl;                  // global {abort, done} of the problem, zero before the launch
  int* lds_fail;             // LDS word: a poll of this workgroup failed (read by everybody behind the next barrier)
  long long limit;           // time limit of an exchange in wall_clock64() ticks (100 MHz)
  int failed;                // this thread knows the problem is off
  int same_xcd;              // 1 once the workgroups have reported one and the same XCC id (and the host trusts the ids)
};
// poll helper of the exchange loops: true = give up (abort raised by a sibling, or this exchange timed out).  The clock is
// only read from the POLL_FREE-th unsuccessful poll on (a poll is a fabric round trip, ~1 us: an exchange that completes
// normally never pays for the clock); t0 is set by the first read.
// limit == 0 (tests): a workgroup gives up at its first unsuccessful look.  limit < 0 (tests, option ba_test_abort_seq):
// the LAST workgroup of every problem raises the abort word at its exchange number -limit, whatever it sees
// (coop_test_abort) - a give-up in the middle of the schedule - and the polls use the default limit.
constexpr int POLL_FREE = 12;
constexpr long long POLL_LIMIT_DEFAULT = 20000;  // 200 us
GL_DEV bool coop_give_up(const Coop& C, int abort_word, int& spins, long long& t0) {
  if (abort_word != 0) return true;
  if (C.limit != 0) {
    if (++spins < POLL_FREE) return false;
    const long long now = (long long)wall_clock64();
    if (spins == POLL_FREE) t0 = now;
    if (now - t0 <= (C.limit < 0 ? POLL_LIMIT_DEFAULT : C.limit)) return false;
  }
  __hip_atomic_store(C.ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
GL_DEV bool coop_test_abort(const Coop& C, unsigned seq) {
  if (C.limit < 0 && C.pb == C.NB - 1 && (long long)seq == -C.limit) {
    __hip_atomic_store(C.ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
// tot[0..31]: this workgroup's sums (LDS) -> the frame's sums (MAXIMUM: maxima instead).  The workgroups hold one GROUP
// of the canonical summation order each (gl_ba_fast_impl.hpp): blocks of two, B_k = g_2k + g_2k+1 (an absent partner
// adds 0.0, as in the one-workgroup kernel), then the blocks in order - every term rounded on its own.
template <bool MAXIMUM>
GL_DEV void coop_totals(Coop& C, double* tot) {
  const unsigned seq = ++C.seq;
  unsigned long long* buf = C.part + (size_t)(seq & 1u) * C.NB * 64;
  const int t = threadIdx.x;
  if (t < 32 && !C.failed) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(tot[t]);
    unsigned long long* mine = buf + ((size_t)C.pb * 32 + t) * 2;
    if (C.same_xcd) {  // the workgroups share an XCD (verified): plain stores stay in its L2, where the L1-bypassing polls find them
      __hip_atomic_store(mine, (bits << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(mine + 1, (bits & 0xffffffff00000000ull) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __hip_atomic_store(mine, (bits << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 1, (bits & 0xffffffff00000000ull) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // all NB partials are requested at once (one round trip when the others are already there)
    constexpr int NBMAX = 8;
    unsigned long long w0[NBMAX], w1[NBMAX];
    bool all, off = false;
    int spins = 0;
    long long t0 = 0;
    do {
      all = true;
      const int ab = __hip_atomic_load(C.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // with the same batch of requests
#pragma unroll
      for (int p = 0; p < NBMAX; ++p) {
        if (p < C.NB) {
          const unsigned long long* w = buf + ((size_t)p * 32 + t) * 2;
          w0[p] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1[p] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int p = 0; p < NBMAX; ++p)
        if (p < C.NB) all = all && (unsigned)w0[p] == seq && (unsigned)w1[p] == seq;
      if (!all) off = coop_give_up(C, ab, spins, t0);
    } while (!all && !off);
    if (off || coop_test_abort(C, seq)) *C.lds_fail = 1;
    double v[NBMAX];
#pragma unroll
    for (int p = 0; p < NBMAX; ++p)
      v[p] = p < C.NB ? __longlong_as_double((long long)((w1[p] & 0xffffffff00000000ull) | (w0[p] >> 32))) : 0.0;
    double s = v[0];
    if (MAXIMUM) {
#pragma unroll
      for (int p = 1; p < NBMAX; ++p)
        if (p < C.NB) s = fmax(s, v[p]);
    } else {
      s = v[0] + v[1];
#pragma unroll
      for (int b = 1; b < NBMAX / 2; ++b)
        if (2 * b < C.NB) s = s + (v[2 * b] + v[2 * b + 1]);
    }
    tot[t] = s;
  }
  __syncthreads();
  C.failed |= *C.lds_fail;
}

}  // namespace gld
