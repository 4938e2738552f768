// ============================================================================
// TEST INFRASTRUCTURE ONLY -- CPU oracle for the gmmloc hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build / link / call anything in oracle/.  The product (gmmloc_amd/) never
// does.
//
// Small fixed-size fp64 algebra restating the Eigen / g2o primitives the
// reference relies on.  Eigen and g2o are NOT vendored under /root/reference
// (eigen_catkin / g2o_catkin, unpinned), so these follow the
// published algorithms of Eigen 3.3 (Inverse_size3, determinant, Quaternion)
// and g2o (se3quat.h) -- PARITY UNPINNED against the real libraries.
// Compile with -ffp-contract=off: source order == evaluation order.
// ============================================================================
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace og {

// ---- 3x3 (row-major double[9]) ---------------------------------------------
// Eigen::Matrix3d::determinant(): bruteforce_det3_helper (Eigen/src/LU/Determinant.h)
// ---- switches for the oracle's DECLARED deviations from Eigen / g2o internals (tests/test_oracle_switches.py) ----------
// The reference's arithmetic lives in un-vendored Eigen / g2o, so four choices of this restatement cannot be
// pinned against the real code.  Each has an alternative here that is at least as close to the library it stands in for;
// the CPU test runs every golden problem and 200 soak problems under each alternative and requires the same decisions
// and poses within 1e-9 - it does not pin g2o, it shows the results do not hang on what could not be pinned.
//   ldlt   0: un-pivoted LDL^T (default)   1: diagonal pivoting, largest |d_ii| first (Eigen::LDLT, what
//          g2o::LinearSolverDense runs)     2: a static symmetric permutation (reversed elimination order: a stand-in for
//          the fill-reducing ordering of Eigen::SimplicialLDLT, what g2o::LinearSolverEigen runs)
//   eig    0: cyclic Jacobi (default)      1: Householder tridiagonalisation + implicit QL (the algorithm family of
//          Eigen::SelfAdjointEigenSolver::compute)
//   lambda_break  1: `if (!g2o_isfinite(_currentLambda)) break;` after a rejected Levenberg trial (newer g2o)
//   tri_order     order in which optimizeTriangulationVec walks its candidate set (the reference iterates an
//          unordered_set of pointers, localization_opt.cpp:144-158): 0 insertion, 1 reversed, 2 ascending / 3 descending index
struct Switches {
  int ldlt = 0, eig = 0, lambda_break = 0, tri_order = 0;
};
inline Switches& switches() {
  static Switches s;
  return s;
}

inline double det3(const double* m) {
  auto h = [&](int a, int b, int c) {
    return m[0 * 3 + a] * (m[1 * 3 + b] * m[2 * 3 + c] - m[1 * 3 + c] * m[2 * 3 + b]);
  };
  return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

// Eigen::Matrix3d::inverse(): cofactor form (Eigen/src/LU/InverseImpl.h,
// compute_inverse<.,.,3>).  result(i,j) = cofactor<j,i>(m) * invdet,
// det = (cof_col0 .* m.col(0)).sum().
inline double cof3(const double* m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
inline void inv3(const double* m, double* r) {
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

// 2x2 (row-major double[4]); Eigen compute_inverse<.,.,2>
inline double det2(const double* m) { return m[0] * m[3] - m[2] * m[1]; }
inline void inv2(const double* m, double* r) {
  const double invdet = 1.0 / det2(m);
  r[0] = m[3] * invdet;
  r[2] = -m[2] * invdet;
  r[1] = -m[1] * invdet;
  r[3] = m[0] * invdet;
}

// C(rxc) = A(rxk) * B(kxc), row-major, plain left-to-right accumulation
inline void matmul(const double* A, const double* B, double* C, int r, int k, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * c + j];
      C[i * c + j] = s;
    }
}
inline void transpose(const double* A, double* At, int r, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) At[j * r + i] = A[i * c + j];
}

// Symmetric eigen-decomposition (n = 2 or 3), cyclic Jacobi.  Eigen uses a
// tridiagonal-QL iteration (SelfAdjointEigenSolver::compute); the values agree
// to rounding and the vectors up to sign -- only thresholds on the values and
// n n^T / |n.v| of the vectors are consumed downstream.
// Output: w ascending, V column c = eigenvector c (V row-major n x n).
inline void eig_sym_jacobi(const double* Ain, int n, double* w, double* V) {
  double A[9];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  // symmetrise exactly like a SelfAdjointView reads the lower triangle
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A[i * n + j] = A[j * n + i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off == 0.0) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double app = A[p * n + p], aqq = A[q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {  // A <- A * J
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // A <- J^T * A
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        A[p * n + q] = 0.0;
        A[q * n + p] = 0.0;
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
  // sort ascending (selection sort on n <= 3), permuting columns of V
  for (int i = 0; i < n - 1; ++i) {
    int m = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      std::swap(w[i], w[m]);
      for (int k = 0; k < n; ++k) std::swap(V[k * n + i], V[k * n + m]);
    }
  }
}


// The same by Householder tridiagonalisation + implicit-shift QL iteration (EISPACK tred2 / tql2; the algorithm family
// of Eigen's SelfAdjointEigenSolver::compute: tridiagonalise, then shifted QR/QL sweeps).  Output as eig_sym_jacobi.
inline void eig_sym_ql(const double* Ain, int n, double* w, double* V) {
  double a[3][3], d[3], e[3];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) a[i][j] = (j <= i) ? Ain[i * n + j] : Ain[j * n + i];  // lower triangle, like a SelfAdjointView
  // tred2
  for (int i = n - 1; i > 0; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(a[i][k]);
      if (scale == 0.0) {
        e[i] = a[i][l];
      } else {
        for (int k = 0; k <= l; ++k) {
          a[i][k] /= scale;
          h += a[i][k] * a[i][k];
        }
        double f = a[i][l];
        const double g = (f >= 0.0 ? -std::sqrt(h) : std::sqrt(h));
        e[i] = scale * g;
        h -= f * g;
        a[i][l] = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          a[j][i] = a[i][j] / h;
          double gg = 0.0;
          for (int k = 0; k <= j; ++k) gg += a[j][k] * a[i][k];
          for (int k = j + 1; k <= l; ++k) gg += a[k][j] * a[i][k];
          e[j] = gg / h;
          f += e[j] * a[i][j];
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = a[i][j];
          const double gg = e[j] - hh * f;
          e[j] = gg;
          for (int k = 0; k <= j; ++k) a[j][k] -= (f * e[k] + gg * a[i][k]);
        }
      }
    } else {
      e[i] = a[i][l];
    }
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += a[i][k] * a[k][j];
        for (int k = 0; k <= l; ++k) a[k][j] -= g * a[k][i];
      }
    }
    d[i] = a[i][i];
    a[i][i] = 1.0;
    for (int j = 0; j <= l; ++j) a[j][i] = a[i][j] = 0.0;
  }
  // tql2
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (++iter > 60) break;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double sn = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = sn * e[i];
          const double b = c * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          sn = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * sn + 2.0 * c * b;
          d[i + 1] = g + (p = sn * r);
          g = c * r - b;
          for (int k = 0; k < n; ++k) {
            f = a[k][i + 1];
            a[k][i + 1] = sn * a[k][i] + c * f;
            a[k][i] = c * a[k][i] - sn * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  for (int i = 0; i < n; ++i) {
    w[i] = d[i];
    for (int k = 0; k < n; ++k) V[k * n + i] = a[k][i];
  }
  for (int i = 0; i < n - 1; ++i) {  // ascending
    int m = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      std::swap(w[i], w[m]);
      for (int k = 0; k < n; ++k) std::swap(V[k * n + i], V[k * n + m]);
    }
  }
}
inline void eig_sym(const double* Ain, int n, double* w, double* V) {
  if (switches().eig == 1) eig_sym_ql(Ain, n, w, V);
  else eig_sym_jacobi(Ain, n, w, V);
}

// Lower Cholesky factor of the SPD 3x3 A (reads the lower triangle), as
// Eigen::LLT<Matrix3d>::matrixL() (unblocked, column by column).
inline bool chol3_lower(const double* A, double* L) {
  for (int i = 0; i < 9; ++i) L[i] = 0.0;
  for (int k = 0; k < 3; ++k) {
    double x = A[k * 3 + k];
    for (int j = 0; j < k; ++j) x -= L[k * 3 + j] * L[k * 3 + j];
    if (!(x > 0.0)) return false;
    x = std::sqrt(x);
    L[k * 3 + k] = x;
    for (int i = k + 1; i < 3; ++i) {
      double s = A[i * 3 + k];
      for (int j = 0; j < k; ++j) s -= L[i * 3 + j] * L[k * 3 + j];
      L[i * 3 + k] = s / x;
    }
  }
  return true;
}

// Dense symmetric solve H x = b (n <= 192) by LDL^T without pivoting.
// g2o: LinearSolverDense = Eigen::LDLT + isPositive(); LinearSolverEigen =
// SimplicialLDLT.  `require_positive` mirrors the isPositive() check.
inline bool ldlt_solve_plain(const double* H, const double* b, double* x, int n, bool require_positive) {
  std::vector<double> Ls((size_t)n * n, 0.0), D(n, 0.0), y(n, 0.0);
  double* L = Ls.data();
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k] * D[k];
    if (d == 0.0 || !std::isfinite(d)) return false;
    if (require_positive && !(d > 0.0)) return false;
    D[j] = d;
    L[j * n + j] = 1.0;
    for (int i = j + 1; i < n; ++i) {
      double s = H[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k] * D[k];
      L[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < n; ++i) y[i] /= D[i];
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s;
  }
  return true;
}


// H x = b through P H P^T = L D L^T with a symmetric permutation: `dynamic` picks the largest remaining |diagonal| at
// every step (Eigen::LDLT), otherwise `perm0` is applied up front (any elimination order: SimplicialLDLT's ordering is one).
inline bool ldlt_solve_perm(const double* H, const double* b, double* x, int n, bool require_positive, bool dynamic, const int* perm0) {
  std::vector<double> A((size_t)n * n), bb(n), D(n), y(n), z(n);
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = perm0 ? perm0[i] : i;
  for (int i = 0; i < n; ++i) {
    bb[i] = b[perm[i]];
    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = H[(size_t)perm[i] * n + perm[j]];
  }
  std::vector<double> L((size_t)n * n, 0.0);
  for (int k = 0; k < n; ++k) {
    if (dynamic) {  // the trailing matrix A[k:, k:] is kept explicitly (right-looking), pivot = its largest |diagonal|
      int m = k;
      for (int i = k + 1; i < n; ++i)
        if (std::fabs(A[(size_t)i * n + i]) > std::fabs(A[(size_t)m * n + m])) m = i;
      if (m != k) {
        for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)m * n + j]);
        for (int i = 0; i < n; ++i) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + m]);
        for (int j = 0; j < k; ++j) std::swap(L[(size_t)k * n + j], L[(size_t)m * n + j]);
        std::swap(bb[k], bb[m]);
        std::swap(perm[k], perm[m]);
      }
    }
    const double d = A[(size_t)k * n + k];
    if (d == 0.0 || !std::isfinite(d)) return false;
    if (require_positive && !(d > 0.0)) return false;
    D[k] = d;
    L[(size_t)k * n + k] = 1.0;
    for (int i = k + 1; i < n; ++i) L[(size_t)i * n + k] = A[(size_t)i * n + k] / d;
    for (int i = k + 1; i < n; ++i)
      for (int j = k + 1; j <= i; ++j) {
        A[(size_t)i * n + j] -= L[(size_t)i * n + k] * d * L[(size_t)j * n + k];
        A[(size_t)j * n + i] = A[(size_t)i * n + j];
      }
  }
  for (int i = 0; i < n; ++i) {
    double s = bb[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < n; ++i) y[i] /= D[i];
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * z[k];
    z[i] = s;
  }
  for (int i = 0; i < n; ++i) x[perm[i]] = z[i];
  return true;
}
inline bool ldlt_solve(const double* H, const double* b, double* x, int n, bool require_positive) {
  const int mode = switches().ldlt;
  if (mode == 1) return ldlt_solve_perm(H, b, x, n, require_positive, true, nullptr);
  if (mode == 2) {
    std::vector<int> rev(n);
    for (int i = 0; i < n; ++i) rev[i] = n - 1 - i;
    return ldlt_solve_perm(H, b, x, n, require_positive, false, rev.data());
  }
  return ldlt_solve_plain(H, b, x, n, require_positive);
}

// ---- quaternion (Eigen coeff order x,y,z,w) --------------------------------
struct Quat {
  double x, y, z, w;
};
inline Quat qmul(const Quat& a, const Quat& b) {  // Eigen quat_product<..>::run
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
inline Quat qconj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }
inline Quat qinverse(const Quat& q) {  // QuaternionBase::inverse
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.0) return Quat{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
  return Quat{0, 0, 0, 0};
}
inline Quat qnormalized(const Quat& q) {
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}
inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
// QuaternionBase::_transformVector: v + w*(2 q x v) + q x (2 q x v)
inline void qrot(const Quat& q, const double* v, double* out) {
  const double qv[3] = {q.x, q.y, q.z};
  double uv[3], c2[3];
  cross3(qv, v, uv);
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  cross3(qv, uv, c2);
  out[0] = v[0] + q.w * uv[0] + c2[0];
  out[1] = v[1] + q.w * uv[1] + c2[1];
  out[2] = v[2] + q.w * uv[2] + c2[2];
}
inline void qtoR(const Quat& q, double* R) {  // QuaternionBase::toRotationMatrix
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
inline Quat qfromR(const double* m) {  // quaternionbase_assign_impl<Mat,3,3>
  Quat q;
  double c[4];
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    c[3] = 0.5 * t;
    t = 0.5 / t;
    c[0] = (m[2 * 3 + 1] - m[1 * 3 + 2]) * t;
    c[1] = (m[0 * 3 + 2] - m[2 * 3 + 0]) * t;
    c[2] = (m[1 * 3 + 0] - m[0 * 3 + 1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    c[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
  q.x = c[0];
  q.y = c[1];
  q.z = c[2];
  q.w = c[3];
  return q;
}

inline void skew(const double* v, double* S) {  // g2o se3_ops.hpp skew()
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

// ---- g2o::SE3Quat (g2o/types/slam3d/se3quat.h) ------------------------------
struct SE3 {
  Quat r{0, 0, 0, 1};
  double t[3]{0, 0, 0};
};
inline void se3_normalize_rotation(SE3& T) {
  if (T.r.w < 0) {
    T.r.x *= -1;
    T.r.y *= -1;
    T.r.z *= -1;
    T.r.w *= -1;
  }
  T.r = qnormalized(T.r);
}
inline SE3 se3_make(const Quat& q, const double* t) {
  SE3 T;
  T.r = q;
  T.t[0] = t[0];
  T.t[1] = t[1];
  T.t[2] = t[2];
  se3_normalize_rotation(T);
  return T;
}
inline void se3_map(const SE3& T, const double* x, double* out) {
  double r[3];
  qrot(T.r, x, r);
  out[0] = r[0] + T.t[0];
  out[1] = r[1] + T.t[1];
  out[2] = r[2] + T.t[2];
}
inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 res = a;
  double rt[3];
  qrot(a.r, b.t, rt);
  res.t[0] += rt[0];
  res.t[1] += rt[1];
  res.t[2] += rt[2];
  res.r = qmul(a.r, b.r);
  se3_normalize_rotation(res);
  return res;
}
inline SE3 se3_inverse(const SE3& a) {
  SE3 ret;
  ret.r = qconj(a.r);
  const double nt[3] = {a.t[0] * -1.0, a.t[1] * -1.0, a.t[2] * -1.0};
  qrot(ret.r, nt, ret.t);
  return ret;
}
inline SE3 se3_exp(const double* u) {  // u = [omega(3), upsilon(3)]
  const double* omega = u;
  const double* upsilon = u + 3;
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew(omega, Om);
  matmul(Om, Om, Om2, 3, 3, 3);
  double a, b, c;  // R = I + a Om + b Om2 ; V = I + b Om + c Om2
  if (theta < 0.00001) {
    a = 1.0;
    b = 0.5;
    c = 1.0 / 6.0;
  } else {
    a = std::sin(theta) / theta;
    b = (1 - std::cos(theta)) / (theta * theta);
    c = (theta - std::sin(theta)) / (std::pow(theta, 3));
  }
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * Om[i] + b * Om2[i];
    V[i] = I + b * Om[i] + c * Om2[i];
  }
  double t[3];
  matmul(V, upsilon, t, 3, 3, 1);
  return se3_make(qfromR(R), t);
}
inline void se3_log(const SE3& T, double* res) {
  double R[9];
  qtoR(T.r, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double omega[3], upsilon[3];
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double Vinv[9], Om[9], Om2[9];
  if (std::fabs(d) > 0.99999) {
    for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
    skew(omega, Om);
    matmul(Om, Om, Om2, 3, 3, 3);
    for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    const double theta = std::acos(d);
    const double f = theta / (2 * std::sqrt(1 - d * d));
    for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
    skew(omega, Om);
    matmul(Om, Om, Om2, 3, 3, 3);
    const double g = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + g * Om2[i];
  }
  matmul(Vinv, T.t, upsilon, 3, 3, 1);
  for (int i = 0; i < 3; ++i) {
    res[i] = omega[i];
    res[i + 3] = upsilon[i];
  }
}
inline void se3_adj(const SE3& T, double* A /*6x6 row-major*/) {
  double R[9], S[9], SR[9];
  qtoR(T.r, R);
  skew(T.t, S);
  matmul(S, R, SR, 3, 3, 3);
  for (int i = 0; i < 36; ++i) A[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[i * 6 + j] = R[i * 3 + j];
      A[(i + 3) * 6 + (j + 3)] = R[i * 3 + j];
      A[(i + 3) * 6 + j] = SR[i * 3 + j];
    }
}

}  // namespace og
