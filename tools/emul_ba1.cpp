// Host-side scalar emulation of the ARITHMETIC of k_ba1_fast (gmmloc_amd/csrc/gl_ba_fast_impl.hpp): the same per-point
// formulas (camera-frame blocks, normalised image coordinates, product-form Schur term, cached D^-1 point step, rcp / rsq
// Newton forms), summed in plain index order.  A debugging aid for numerical questions that do not depend on the summation
// order (tools/emul_ba1.py drives it against the oracle's Levenberg trace); NOT part of the product and not an oracle.
// Build: g++ -O2 -ffp-contract=off -shared -fPIC tools/emul_ba1.cpp -o build_tmp/libemul_ba1.so
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
int g_diag_trial = -1;

enum { F_EXISTS = 1, F_STEREO = 2, F_ASSOC = 4, F_DEG = 8, F_LEVR = 16, F_LEVG = 32 };

// variant switches (bit mask)
enum {
  V_PLAIN_SCHUR = 1,   // A - A D^-1 A and a - A u instead of the product forms
  V_SYM_C = 2,         // mean of the two triangles of the product
  V_STEP_CACHED = 4,   // pass B: eps = u - (A D^-1)^T gd from pass A's values (fp64) instead of the re-evaluated residual
  V_LDLT_REFINE = 8,   // one step of iterative refinement of the 6x6 solve
  V_DINV_LDL = 16,     // point block solve by LDL^T instead of the cofactor inverse
  V_WORLD_GMM = 32,
};

struct Pose {
  double R[9], t[3];
};

inline double rcp_nr(double a) {
  double x = 1.0 / a;
  x = std::fma(std::fma(-a, x, 1.0), x, x);
  return x;
}
inline double rsq_nr(double a) { return 1.0 / std::sqrt(a); }
inline void huber_bf(double e, double delta, double dsqr, double& rho0, double& rho1) {
  const double r = rsq_nr(std::fmax(e, 1e-300));
  const bool in = e <= dsqr;
  rho1 = in ? 1.0 : delta * r;
  rho0 = in ? e : (2.0 * delta * (e * r) - dsqr);
}
inline void mm3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
inline void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline void skew(const double* v, double* S) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
inline void sym3_mul_vec(const double* S, const double* v, double* o) {
  o[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
  o[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
  o[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}
void qtoR(const double* q, double* R) {  // (x y z w), Eigen toRotationMatrix
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void qfromR(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}

Pose pose_update(const Pose& P, const double* u) {
  const double th2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
  double a, b, c;
  if (th2 < 1e-4) {
    a = std::fma(std::fma(std::fma(-1.0 / 5040, th2, 1.0 / 120), th2, -1.0 / 6), th2, 1.0);
    b = std::fma(std::fma(std::fma(-1.0 / 40320, th2, 1.0 / 720), th2, -1.0 / 24), th2, 0.5);
    c = std::fma(std::fma(std::fma(-1.0 / 362880, th2, 1.0 / 5040), th2, -1.0 / 120), th2, 1.0 / 6);
  } else {
    const double theta = std::sqrt(th2);
    const double st = std::sin(theta), ct = std::cos(theta), it = 1.0 / theta;
    a = st * it;
    b = (1 - ct) * it * it;
    c = (theta - st) * it * it * it;
  }
  const double w0 = u[0], w1 = u[1], w2 = u[2];
  const double s00 = w0 * w0 - th2, s11 = w1 * w1 - th2, s22 = w2 * w2 - th2;
  const double s01 = w0 * w1, s02 = w0 * w2, s12 = w1 * w2;
  double dR[9], V[9];
  dR[0] = std::fma(b, s00, 1.0); dR[4] = std::fma(b, s11, 1.0); dR[8] = std::fma(b, s22, 1.0);
  dR[1] = std::fma(b, s01, -a * w2); dR[3] = std::fma(b, s01, a * w2);
  dR[2] = std::fma(b, s02, a * w1); dR[6] = std::fma(b, s02, -a * w1);
  dR[5] = std::fma(b, s12, -a * w0); dR[7] = std::fma(b, s12, a * w0);
  V[0] = std::fma(c, s00, 1.0); V[4] = std::fma(c, s11, 1.0); V[8] = std::fma(c, s22, 1.0);
  V[1] = std::fma(c, s01, -b * w2); V[3] = std::fma(c, s01, b * w2);
  V[2] = std::fma(c, s02, b * w1); V[6] = std::fma(c, s02, -b * w1);
  V[5] = std::fma(c, s12, -b * w0); V[7] = std::fma(c, s12, b * w0);
  Pose N;
  mm3(dR, P.R, N.R);
  for (int i = 0; i < 3; ++i)
    N.t[i] = dR[i * 3] * P.t[0] + dR[i * 3 + 1] * P.t[1] + dR[i * 3 + 2] * P.t[2] + V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
  return N;
}

#define GL_U(i, j) ((i) * 6 - (i) * ((i)-1) / 2 + ((j) - (i)))

void prior_error(const double* mi, const Pose& P, double* e) {
  double dR[9], dt[3];
  mm3(mi, P.R, dR);
  for (int i = 0; i < 3; ++i) dt[i] = std::fma(mi[i * 3], P.t[0], std::fma(mi[i * 3 + 1], P.t[1], std::fma(mi[i * 3 + 2], P.t[2], mi[9 + i])));
  const double d = 0.5 * (dR[0] + dR[4] + dR[8] - 1);
  const double v[3] = {dR[7] - dR[5], dR[2] - dR[6], dR[3] - dR[1]};
  double w[3], g;
  if (std::fabs(d) > 0.99999) {
    for (int i = 0; i < 3; ++i) w[i] = 0.5 * v[i];
    g = 1. / 12.;
  } else {
    const double theta = std::acos(d), sn = std::sqrt(1 - d * d);
    const double f = theta / (2 * sn);
    for (int i = 0; i < 3; ++i) w[i] = f * v[i];
    g = (1 - theta * (1 + d) / (2 * sn)) / (theta * theta);
  }
  double c1[3], c2[3];
  cross(w, dt, c1);
  cross(w, c1, c2);
  for (int i = 0; i < 3; ++i) {
    e[i] = w[i];
    e[3 + i] = std::fma(g, c2[i], std::fma(-0.5, c1[i], dt[i]));
  }
}
double prior_chi2(const double* e) {
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0)), st = 1.0 / (0.01 * 0.01);
  return sr * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + st * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
}
// rec: H (21 packed upper), b (6), chi2
void prior_record(const double* mi, const Pose& P, double* rec) {
  double e[6];
  prior_error(mi, P, e);
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0)), st = 1.0 / (0.01 * 0.01);
  double Ri[9], ti[3], S[9], SR[9], A1[9], Lh[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = P.R[j * 3 + i];
  for (int i = 0; i < 3; ++i) ti[i] = -(Ri[i * 3] * P.t[0] + Ri[i * 3 + 1] * P.t[1] + Ri[i * 3 + 2] * P.t[2]);
  skew(ti, S);
  mm3(S, Ri, SR);
  skew(e, A1);
  skew(e + 3, Lh);
  for (int i = 0; i < 9; ++i) {
    A1[i] = 0.5 * A1[i] + ((i % 4 == 0) ? 1.0 : 0.0);
    Lh[i] = 0.5 * Lh[i];
  }
  // Jr = [[A1, Lh], [0, A1]], Adj = [[Ri, 0], [SR, Ri]], J = Jr Adj
  double Jr[36] = {0}, Adj[36] = {0}, J[36];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Jr[i * 6 + j] = A1[i * 3 + j];
      Jr[i * 6 + 3 + j] = Lh[i * 3 + j];
      Jr[(3 + i) * 6 + 3 + j] = A1[i * 3 + j];
      Adj[i * 6 + j] = Ri[i * 3 + j];
      Adj[(3 + i) * 6 + j] = SR[i * 3 + j];
      Adj[(3 + i) * 6 + 3 + j] = Ri[i * 3 + j];
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0.0;
      for (int l = 0; l < 6; ++l) s = std::fma(Jr[i * 6 + l], Adj[l * 6 + j], s);
      J[i * 6 + j] = s;
    }
  for (int i = 0; i < 6; ++i) {
    double g = 0.0;
    for (int r = 0; r < 6; ++r) g = std::fma(J[r * 6 + i], (r < 3 ? sr : st) * e[r], g);
    rec[21 + i] = -g;
    for (int j = i; j < 6; ++j) {
      double h = 0.0;
      for (int r = 0; r < 6; ++r) h = std::fma(J[r * 6 + i] * (r < 3 ? sr : st), J[r * 6 + j], h);
      rec[GL_U(i, j)] = h;
    }
  }
  rec[27] = prior_chi2(e);
}

bool ldlt6_packed(double* a, const double* b, double lambda, double* x) {
  double iD[6];
  bool ok = true;
  for (int j = 0; j < 6; ++j) {
    double d = a[GL_U(j, j)] + lambda;
    for (int kk = 0; kk < j; ++kk) d -= a[GL_U(kk, j)] * a[GL_U(kk, j)] * a[GL_U(kk, kk)];
    if (d == 0.0 || !std::isfinite(d)) ok = false;
    a[GL_U(j, j)] = d;
    iD[j] = rcp_nr(d);
    for (int i = j + 1; i < 6; ++i) {
      double s = a[GL_U(j, i)];
      for (int kk = 0; kk < j; ++kk) s -= a[GL_U(kk, i)] * a[GL_U(kk, j)] * a[GL_U(kk, kk)];
      a[GL_U(j, i)] = s * iD[j];
    }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int kk = 0; kk < i; ++kk) s -= a[GL_U(kk, i)] * y[kk];
    y[i] = s;
  }
  for (int i = 0; i < 6; ++i) y[i] *= iD[i];
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int kk = i + 1; kk < 6; ++kk) s -= a[GL_U(i, kk)] * x[kk];
    x[i] = s;
  }
  return ok;
}

struct Frame {
  int L;
  const double* obn;     // L x 3 normalised observations
  const int32_t* fl0;    // L flag words (F_EXISTS | F_STEREO | F_ASSOC | F_DEG | octave << 8)
  const int32_t* assoc;  // L
  const double* plane4;  // K x 4
  const double* hgw;     // K x 6
  const double* mean;    // K x 3
  double sx[8], sy[8];
  double bn, lm, str_thresh;
  double dmono, dstereo;
  // fixed observer key-frames (gl_track_frames_anchored, F > 0): poses {R, t} x NF, per point and key-frame the normalised
  // observation and the octave (< 0: not observed)
  int NF = 0;
  const double* fRt = nullptr;   // NF x 12
  const double* fobn = nullptr;  // L x NF x 3
  const int32_t* foct = nullptr; // L x NF
};

struct Lin {
  double q[3], A[6], a[3], D[6], b[3];
  double rho0_r, rho1, chi_g;
};

double gmm_nondeg(const Frame& F, int a, const double* R, const double* p, double* Hc, double* bc) {
  const double* Hg = F.hgw + (size_t)a * 6;
  const double d[3] = {p[0] - F.mean[(size_t)a * 3], p[1] - F.mean[(size_t)a * 3 + 1], p[2] - F.mean[(size_t)a * 3 + 2]};
  double Hd[3];
  sym3_mul_vec(Hg, d, Hd);
  const double chi = d[0] * Hd[0] + d[1] * Hd[1] + d[2] * Hd[2];
  if (bc)
    for (int i = 0; i < 3; ++i) bc[i] = -(R[i * 3] * Hd[0] + R[i * 3 + 1] * Hd[1] + R[i * 3 + 2] * Hd[2]);
  if (Hc) {
    double RH[9];
    for (int i = 0; i < 3; ++i) {
      RH[i * 3 + 0] = R[i * 3] * Hg[0] + R[i * 3 + 1] * Hg[1] + R[i * 3 + 2] * Hg[2];
      RH[i * 3 + 1] = R[i * 3] * Hg[1] + R[i * 3 + 1] * Hg[3] + R[i * 3 + 2] * Hg[4];
      RH[i * 3 + 2] = R[i * 3] * Hg[2] + R[i * 3 + 1] * Hg[4] + R[i * 3 + 2] * Hg[5];
    }
    Hc[0] = RH[0] * R[0] + RH[1] * R[1] + RH[2] * R[2];
    Hc[1] = RH[0] * R[3] + RH[1] * R[4] + RH[2] * R[5];
    Hc[2] = RH[0] * R[6] + RH[1] * R[7] + RH[2] * R[8];
    Hc[3] = RH[3] * R[3] + RH[4] * R[4] + RH[5] * R[5];
    Hc[4] = RH[3] * R[6] + RH[4] * R[7] + RH[5] * R[8];
    Hc[5] = RH[6] * R[6] + RH[7] * R[7] + RH[8] * R[8];
  }
  return chi;
}

double reproj_n(const double* q, const double* ob, bool stereo, double bn, double sx, double sy, double* e, double& iz) {
  iz = rcp_nr(q[2]);
  e[0] = std::fma(-q[0], iz, ob[0]);
  e[1] = std::fma(-q[1], iz, ob[1]);
  e[2] = stereo ? std::fma(bn - q[0], iz, ob[2]) : 0.0;
  return std::fma(sy * e[1], e[1], sx * std::fma(e[2], e[2], e[0] * e[0]));
}

struct Pt {
  int fl;
  bool ar, ag;
  double sx, sy;
  const double* ob;
  const double* nd;
  int asc;
  int l;
  bool af;  // some fixed-observer edge of the point is active
};

// rows of a fixed key-frame's projection Jacobian, rotated into the CURRENT camera frame: j' = j (R_f R^T)
struct FixedEdge {
  bool stereo;
  double e[3], iz, chi, sx, sy;
  double j0[3], j1[3], j2[3];
};
void fixed_eval(const Frame& F, const double* Rf, const double* p, const double* ob, int oc, FixedEdge& E) {
  double q[3];
  for (int k = 0; k < 3; ++k) q[k] = std::fma(Rf[k * 3], p[0], std::fma(Rf[k * 3 + 1], p[1], std::fma(Rf[k * 3 + 2], p[2], Rf[9 + k])));
  E.stereo = !(ob[2] < -1e29);
  E.sx = F.sx[oc];
  E.sy = F.sy[oc];
  E.chi = reproj_n(q, ob, E.stereo, F.bn, E.sx, E.sy, E.e, E.iz);
  const double iz2 = E.iz * E.iz;
  const double c0 = -q[0] * iz2, c1 = -q[1] * iz2, c2 = std::fma(F.bn, iz2, c0);
  E.j0[0] = E.iz; E.j0[1] = 0.0; E.j0[2] = c0;
  E.j1[0] = 0.0; E.j1[1] = E.iz; E.j1[2] = c1;
  E.j2[0] = E.iz; E.j2[1] = 0.0; E.j2[2] = c2;
}
void rows_to_camera(const double* Rf, const Pose& P, FixedEdge& E) {
  // Rel = R_f R^T; j' = j Rel
  double Rel[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rel[i * 3 + j] = Rf[i * 3] * P.R[j * 3] + Rf[i * 3 + 1] * P.R[j * 3 + 1] + Rf[i * 3 + 2] * P.R[j * 3 + 2];
  double* rows[3] = {E.j0, E.j1, E.j2};
  for (int r = 0; r < 3; ++r) {
    const double a = rows[r][0], b = rows[r][1], c = rows[r][2];
    for (int j = 0; j < 3; ++j) rows[r][j] = a * Rel[j] + b * Rel[3 + j] + c * Rel[6 + j];
  }
}

double lin_fast(const Frame& F, const Pose& P, const Pt& c, const double* p, bool robust, Lin& o) {
  for (int k = 0; k < 3; ++k) o.q[k] = std::fma(P.R[k * 3], p[0], std::fma(P.R[k * 3 + 1], p[1], std::fma(P.R[k * 3 + 2], p[2], P.t[k])));
  for (int k = 0; k < 6; ++k) o.A[k] = 0.0;
  for (int k = 0; k < 3; ++k) o.a[k] = 0.0;
  o.rho0_r = 0.0;
  o.rho1 = 1.0;
  o.chi_g = 0.0;
  double chi_r = 0.0;
  if (c.ar) {
    const bool stereo = c.fl & F_STEREO;
    double e[3], iz;
    chi_r = reproj_n(o.q, c.ob, stereo, F.bn, c.sx, c.sy, e, iz);
    o.rho0_r = chi_r;
    if (robust) huber_bf(chi_r, stereo ? F.dstereo : F.dmono, stereo ? F.dstereo * F.dstereo : F.dmono * F.dmono, o.rho0_r, o.rho1);
    const double wx = o.rho1 * c.sx, wy = o.rho1 * c.sy;
    const double t = stereo ? wx : 0.0;
    const double iz2 = iz * iz;
    const double c0 = -o.q[0] * iz2, c1 = -o.q[1] * iz2, c2 = std::fma(F.bn, iz2, c0);
    const double wxc0 = wx * c0, wyc1 = wy * c1, tc2 = t * c2, wyiz = wy * iz;
    o.A[0] = (wx + t) * iz2;
    o.A[2] = iz * (wxc0 + tc2);
    o.A[3] = wy * iz2;
    o.A[4] = wyiz * c1;
    o.A[5] = std::fma(tc2, c2, std::fma(wyc1, c1, wxc0 * c0));
    o.a[0] = iz * std::fma(t, e[2], wx * e[0]);
    o.a[1] = wyiz * e[1];
    o.a[2] = std::fma(tc2, e[2], std::fma(wyc1, e[1], wxc0 * e[0]));
  }
  for (int k = 0; k < 6; ++k) o.D[k] = o.A[k];
  for (int k = 0; k < 3; ++k) o.b[k] = o.a[k];
  if (c.ag) {
    if (c.fl & F_DEG) {
      const double nx = c.nd[0], ny = c.nd[1], nz = c.nd[2];
      const double eg = std::fma(nz, p[2], std::fma(ny, p[1], nx * p[0])) - c.nd[3];
      const double lm = F.lm;
      o.chi_g = eg * (lm * eg);
      double nc[3], tn[3];
      for (int k = 0; k < 3; ++k) {
        nc[k] = std::fma(P.R[k * 3 + 2], nz, std::fma(P.R[k * 3 + 1], ny, P.R[k * 3] * nx));
        tn[k] = lm * nc[k];
      }
      o.D[0] = std::fma(tn[0], nc[0], o.D[0]);
      o.D[1] = std::fma(tn[0], nc[1], o.D[1]);
      o.D[2] = std::fma(tn[0], nc[2], o.D[2]);
      o.D[3] = std::fma(tn[1], nc[1], o.D[3]);
      o.D[4] = std::fma(tn[1], nc[2], o.D[4]);
      o.D[5] = std::fma(tn[2], nc[2], o.D[5]);
      for (int k = 0; k < 3; ++k) o.b[k] = std::fma(-eg, tn[k], o.b[k]);
    } else {
      double Hc[6], bc[3];
      o.chi_g = gmm_nondeg(F, c.asc, P.R, p, Hc, bc);
      for (int k = 0; k < 6; ++k) o.D[k] += Hc[k];
      for (int k = 0; k < 3; ++k) o.b[k] += bc[k];
    }
  }
  return chi_r;
}

double gmm_chi2_fast(const Frame& F, const double* nd, int fl, int asc, const double* p) {
  if (fl & F_DEG) {
    const double eg = (nd[0] * p[0] + nd[1] * p[1] + nd[2] * p[2]) - nd[3];
    return eg * (F.lm * eg);
  }
  return gmm_nondeg(F, asc, nullptr, p, nullptr, nullptr);
}

void sym3_inv_fast(const double* S, double* I) {
  const double c00 = S[3] * S[5] - S[4] * S[4];
  const double c01 = S[2] * S[4] - S[1] * S[5];
  const double c02 = S[1] * S[4] - S[2] * S[3];
  const double det = S[0] * c00 + S[1] * c01 + S[2] * c02;
  const double id = rcp_nr(det);
  I[0] = c00 * id;
  I[1] = c01 * id;
  I[2] = c02 * id;
  I[3] = (S[0] * S[5] - S[2] * S[2]) * id;
  I[4] = (S[1] * S[2] - S[0] * S[4]) * id;
  I[5] = (S[0] * S[3] - S[1] * S[1]) * id;
}


// D = L Delta L^T (unpivoted; D is symmetric positive definite): f = {l10, l20, l21, 1/d0, 1/d1, 1/d2}
void ldl3_factor(const double* D, double* f) {
  const double i0 = rcp_nr(D[0]);
  const double l1 = D[1] * i0, l2 = D[2] * i0;
  const double d1 = std::fma(-l1, D[1], D[3]);
  const double e = std::fma(-l1, D[2], D[4]);
  const double i1 = rcp_nr(d1);
  const double l3 = e * i1;
  const double d2 = std::fma(-l3, e, std::fma(-l2, D[2], D[5]));
  const double i2 = rcp_nr(d2);
  f[0] = l1; f[1] = l2; f[2] = l3; f[3] = i0; f[4] = i1; f[5] = i2;
}
// same factors with the first two pivots' reciprocals independent of each other (two serial reciprocals instead of three)
void ldl3_factor_par(const double* D, double* f) {
  const double i0 = rcp_nr(D[0]);
  const double m2 = std::fma(D[0], D[3], -(D[1] * D[1]));  // D0 d1
  const double e2 = std::fma(D[0], D[4], -(D[1] * D[2]));  // D0 e
  const double im = rcp_nr(m2);
  const double l1 = D[1] * i0, l2 = D[2] * i0;
  const double i1 = D[0] * im, l3 = e2 * im;
  const double e = e2 * i0;
  const double d2 = std::fma(-l3, e, std::fma(-l2, D[2], D[5]));
  const double i2 = rcp_nr(d2);
  f[0] = l1; f[1] = l2; f[2] = l3; f[3] = i0; f[4] = i1; f[5] = i2;
}
void ldl3_solve(const double* f, const double* b, double* x) {
  const double y0 = b[0];
  const double y1 = std::fma(-f[0], y0, b[1]);
  const double y2 = std::fma(-f[2], y1, std::fma(-f[1], y0, b[2]));
  const double z2 = y2 * f[5];
  const double z1 = std::fma(-f[2], z2, y1 * f[4]);
  const double z0 = std::fma(-f[1], z2, std::fma(-f[0], z1, y0 * f[3]));
  x[0] = z0; x[1] = z1; x[2] = z2;
}

void pose_terms(const double* q, const double* C, const double* c, bool with_rhs, double* acc) {
  const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
  double M[9];
  for (int j = 0; j < 3; ++j) {
    M[j] = std::fma(q[1], Cf[6 + j], -q[2] * Cf[3 + j]);
    M[3 + j] = std::fma(q[2], Cf[j], -q[0] * Cf[6 + j]);
    M[6 + j] = std::fma(q[0], Cf[3 + j], -q[1] * Cf[j]);
  }
  acc[0] += std::fma(q[1], M[2], -q[2] * M[1]);
  acc[1] += std::fma(q[2], M[0], -q[0] * M[2]);
  acc[2] += std::fma(q[0], M[1], -q[1] * M[0]);
  acc[3] += M[0];
  acc[4] += M[1];
  acc[5] += M[2];
  acc[6] += std::fma(q[2], M[3], -q[0] * M[5]);
  acc[7] += std::fma(q[0], M[4], -q[1] * M[3]);
  acc[8] += M[3];
  acc[9] += M[4];
  acc[10] += M[5];
  acc[11] += std::fma(q[0], M[7], -q[1] * M[6]);
  acc[12] += M[6];
  acc[13] += M[7];
  acc[14] += M[8];
  acc[15] += C[0];
  acc[16] += C[1];
  acc[17] += C[2];
  acc[18] += C[3];
  acc[19] += C[4];
  acc[20] += C[5];
  if (with_rhs) {
    acc[21] += std::fma(q[1], c[2], -q[2] * c[1]);
    acc[22] += std::fma(q[2], c[0], -q[0] * c[2]);
    acc[23] += std::fma(q[0], c[1], -q[1] * c[0]);
    acc[24] += c[0];
    acc[25] += c[1];
    acc[26] += c[2];
  }
}

void ad_product(const double* A, const double* Di, double* AD) {
  const double y[9] = {Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5]};
  for (int j = 0; j < 3; ++j) {
    AD[j] = std::fma(A[0], y[j], A[2] * y[6 + j]);
    AD[3 + j] = std::fma(A[3], y[3 + j], A[4] * y[6 + j]);
    AD[6 + j] = std::fma(A[2], y[j], std::fma(A[4], y[3 + j], A[5] * y[6 + j]));
  }
}

struct State {
  std::vector<double> sp, chir, un, bk, ucache, adcache;  // L x 3, L, L x 6, L x 3
  std::vector<int> fl;
  std::vector<double> chif;  // L x NF: stale chi2 of the fixed-observer edges (rho' between the two passes of a trial)
  std::vector<int> levf;     // L x NF: the edge is at level 1
};

bool fixed_active(const Frame& F, const State& S, int l, int f) {
  return (S.fl[l] & F_EXISTS) && F.foct[(size_t)l * F.NF + f] >= 0 && !S.levf[(size_t)l * F.NF + f];
}
bool any_fixed(const Frame& F, const State& S, int l) {
  for (int f = 0; f < F.NF; ++f)
    if (fixed_active(F, S, l, f)) return true;
  return false;
}
Pt make_pt(const Frame& F, const State& S, int l) {
  Pt c;
  c.fl = S.fl[l];
  c.ar = (c.fl & F_EXISTS) && !(c.fl & F_LEVR);
  c.ag = (c.fl & F_EXISTS) && (c.fl & F_ASSOC) && !(c.fl & F_LEVG);
  const int oc = (c.fl >> 8) & 7;
  c.sx = F.sx[oc];
  c.sy = F.sy[oc];
  c.ob = F.obn + (size_t)l * 3;
  const int a = F.assoc[l];
  c.nd = F.plane4 + (size_t)(a > 0 ? a : 0) * 4;
  c.asc = (c.fl & F_ASSOC) && !(c.fl & F_DEG) ? a : -1;
  c.l = l;
  c.af = any_fixed(F, S, l);
  return c;
}

// pass A / lambda init: the fixed-observer edges of point l add to its block and right-hand side (camera frame); returns
// their robustified chi2; leaves rho' of every edge in its stale-chi2 cell for pass B
double fixed_lin(const Frame& F, State& S, const Pose& P, int l, const double* p, bool robust, Lin& o, bool keep_rho) {
  double sum = 0.0;
  for (int f = 0; f < F.NF; ++f) {
    if (!fixed_active(F, S, l, f)) continue;
    const double* Rf = F.fRt + (size_t)f * 12;
    FixedEdge E;
    fixed_eval(F, Rf, p, F.fobn + ((size_t)l * F.NF + f) * 3, F.foct[(size_t)l * F.NF + f], E);
    double rho0 = E.chi, rho1 = 1.0;
    if (robust) huber_bf(E.chi, E.stereo ? F.dstereo : F.dmono, E.stereo ? F.dstereo * F.dstereo : F.dmono * F.dmono, rho0, rho1);
    sum += rho0;
    if (keep_rho) S.chif[(size_t)l * F.NF + f] = rho1;
    rows_to_camera(Rf, P, E);
    const double w[3] = {rho1 * E.sx, rho1 * E.sy, E.stereo ? rho1 * E.sx : 0.0};
    const double* rows[3] = {E.j0, E.j1, E.j2};
    for (int r = 0; r < 3; ++r) {
      const double* j = rows[r];
      const double wj[3] = {w[r] * j[0], w[r] * j[1], w[r] * j[2]};
      o.D[0] = std::fma(wj[0], j[0], o.D[0]);
      o.D[1] = std::fma(wj[0], j[1], o.D[1]);
      o.D[2] = std::fma(wj[0], j[2], o.D[2]);
      o.D[3] = std::fma(wj[1], j[1], o.D[3]);
      o.D[4] = std::fma(wj[1], j[2], o.D[4]);
      o.D[5] = std::fma(wj[2], j[2], o.D[5]);
      for (int k = 0; k < 3; ++k) o.b[k] = std::fma(E.e[r], wj[k], o.b[k]);
    }
  }
  return sum;
}
// pass B, step half: right-hand sides of the fixed edges at the linearisation point (they do not couple to the pose step)
void fixed_rhs(const Frame& F, const State& S, const Pose& P, int l, const double* p, double* rhs) {
  for (int f = 0; f < F.NF; ++f) {
    if (!fixed_active(F, S, l, f)) continue;
    const double* Rf = F.fRt + (size_t)f * 12;
    FixedEdge E;
    fixed_eval(F, Rf, p, F.fobn + ((size_t)l * F.NF + f) * 3, F.foct[(size_t)l * F.NF + f], E);
    const double rho1 = S.chif[(size_t)l * F.NF + f];
    rows_to_camera(Rf, P, E);
    const double w[3] = {rho1 * E.sx, rho1 * E.sy, E.stereo ? rho1 * E.sx : 0.0};
    const double* rows[3] = {E.j0, E.j1, E.j2};
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) rhs[k] = std::fma(E.e[r] * w[r], rows[r][k], rhs[k]);
  }
}
// pass B, evaluation half: chi2 of the fixed edges at the trial point; the stale chi2 is rewritten
double fixed_chi(const Frame& F, State& S, int l, const double* pn, bool robust) {
  double sum = 0.0;
  for (int f = 0; f < F.NF; ++f) {
    if (!fixed_active(F, S, l, f)) continue;
    FixedEdge E;
    fixed_eval(F, F.fRt + (size_t)f * 12, pn, F.fobn + ((size_t)l * F.NF + f) * 3, F.foct[(size_t)l * F.NF + f], E);
    S.chif[(size_t)l * F.NF + f] = E.chi;
    double rho0 = E.chi, rho1;
    if (robust) huber_bf(E.chi, E.stereo ? F.dstereo : F.dmono, E.stereo ? F.dstereo * F.dstereo : F.dmono * F.dmono, rho0, rho1);
    sum += rho0;
  }
  return sum;
}

int optimize_fast(const Frame& F, State& S, Pose& P, bool robust, int iters, int& trials, bool has_prior, const double* mi, int variant,
                  double* trace, int trace_cap) {
  const int L = F.L;
  int n_ar = 0, n_any = 0;
  for (int l = 0; l < L; ++l) {
    const Pt c = make_pt(F, S, l);
    n_ar += c.ar;
    n_any += c.ar || c.ag || c.af;
  }
  const bool pose_active = n_ar > 0 || has_prior;
  const bool prior_on = has_prior && pose_active;
  if (!pose_active && !n_any) return -1;
  double rec[2][32];
  int cur = 0;
  if (prior_on) prior_record(mi, P, rec[0]);
  double lambda = 0.0, ni = 2.0;
  int cj = 0;
  for (int it = 0; it < iters; ++it) {
    double rho = 0.0, currentChi = 0.0;
    int qmax = 0;
    if (it == 0) {
      double md = 0.0, acc[32] = {0};
      for (int l = 0; l < L; ++l) {
        const Pt c = make_pt(F, S, l);
        if (!(c.ar || c.ag || c.af)) continue;
        Lin o;
        lin_fast(F, P, c, &S.sp[(size_t)l * 3], robust, o);
        if (c.af) fixed_lin(F, S, P, l, &S.sp[(size_t)l * 3], robust, o, false);
        const double Hf[9] = {o.D[0], o.D[1], o.D[2], o.D[1], o.D[3], o.D[4], o.D[2], o.D[4], o.D[5]};
        for (int j = 0; j < 3; ++j) {
          double s = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) s += P.R[a * 3 + j] * Hf[a * 3 + b] * P.R[b * 3 + j];
          md = std::fmax(md, std::fabs(s));
        }
        if (c.ar) {
          const double zero[3] = {0, 0, 0};
          pose_terms(o.q, o.A, zero, false, acc);
        }
      }
      if (pose_active) {
        if (prior_on)
          for (int i = 0; i < 6; ++i) acc[GL_U(i, i)] += rec[cur][GL_U(i, i)];
        for (int i = 0; i < 6; ++i) md = std::fmax(md, std::fabs(acc[GL_U(i, i)]));
      }
      lambda = 1e-5 * md;
      ni = 2.0;
    }
    do {
      double acc[32] = {0};
      // pass A
      for (int l = 0; l < L; ++l) {
        const Pt c = make_pt(F, S, l);
        if (!(c.ar || c.ag || c.af)) continue;
        Lin o;
        const double* p = &S.sp[(size_t)l * 3];
        lin_fast(F, P, c, p, robust, o);
        double chi_f = 0.0;
        if (c.af) chi_f = fixed_lin(F, S, P, l, p, robust, o, true);
        acc[27] += (o.rho0_r + o.chi_g) + chi_f;
        double Dinv[6], u[3];
        const double D[6] = {o.D[0] + lambda, o.D[1], o.D[2], o.D[3] + lambda, o.D[4], o.D[5] + lambda};
        if (variant & V_DINV_LDL) {
          if (variant & 64) ldl3_factor_par(D, Dinv); else ldl3_factor(D, Dinv);
          ldl3_solve(Dinv, o.b, u);
        } else {
          sym3_inv_fast(D, Dinv);
          sym3_mul_vec(Dinv, o.b, u);
        }
        acc[28] += std::fma(u[0], o.b[0], std::fma(u[1], o.b[1], u[2] * o.b[2]));
        for (int j = 0; j < 6; ++j) S.un[(size_t)l * 6 + j] = Dinv[j];
        for (int j = 0; j < 3; ++j) S.ucache[(size_t)l * 3 + j] = u[j];
        for (int j = 0; j < 9; ++j) S.adcache[(size_t)l * 9 + j] = 0.0;
        if (c.ar) {
          S.chir[l] = o.rho1;
          double C[6], cc[3], AD[9];
          if (variant & V_DINV_LDL) {
            const double Af[9] = {o.A[0], o.A[1], o.A[2], o.A[1], o.A[3], o.A[4], o.A[2], o.A[4], o.A[5]};
            for (int r = 0; r < 3; ++r) ldl3_solve(Dinv, Af + r * 3, AD + r * 3);  // row r of A D^-1 = D^-1 (column r of A)
          } else {
            ad_product(o.A, Dinv, AD);
          }
          for (int j = 0; j < 9; ++j) S.adcache[(size_t)l * 9 + j] = AD[j];
          const int ri[6] = {0, 0, 0, 1, 1, 2}, ci[6] = {0, 1, 2, 1, 2, 2};
          if (variant & V_PLAIN_SCHUR) {
            // C = A - (A D^-1) A ; cc = a - (A D^-1) b
            const double Af[9] = {o.A[0], o.A[1], o.A[2], o.A[1], o.A[3], o.A[4], o.A[2], o.A[4], o.A[5]};
            for (int e = 0; e < 6; ++e) {
              const int r = ri[e], j = ci[e];
              C[e] = Af[r * 3 + j] - (AD[r * 3] * Af[j] + AD[r * 3 + 1] * Af[3 + j] + AD[r * 3 + 2] * Af[6 + j]);
            }
            for (int r = 0; r < 3; ++r) cc[r] = o.a[r] - (AD[r * 3] * o.b[0] + AD[r * 3 + 1] * o.b[1] + AD[r * 3 + 2] * o.b[2]);
          } else {
            const double M[9] = {(o.D[0] - o.A[0]) + lambda, o.D[1] - o.A[1], o.D[2] - o.A[2],
                                 o.D[1] - o.A[1], (o.D[3] - o.A[3]) + lambda, o.D[4] - o.A[4],
                                 o.D[2] - o.A[2], o.D[4] - o.A[4], (o.D[5] - o.A[5]) + lambda};
            for (int e = 0; e < 6; ++e) {
              const int r = ri[e], j = ci[e];
              C[e] = std::fma(M[r * 3], AD[j * 3], std::fma(M[r * 3 + 1], AD[j * 3 + 1], M[r * 3 + 2] * AD[j * 3 + 2]));
              if ((variant & V_SYM_C) && r != j) {
                const double lo = std::fma(M[j * 3], AD[r * 3], std::fma(M[j * 3 + 1], AD[r * 3 + 1], M[j * 3 + 2] * AD[r * 3 + 2]));
                C[e] = 0.5 * (C[e] + lo);
              }
            }
            for (int r = 0; r < 3; ++r) cc[r] = std::fma(M[r * 3], u[0], std::fma(M[r * 3 + 1], u[1], std::fma(M[r * 3 + 2], u[2], o.a[r] - o.b[r])));
          }

          if (g_diag_trial == trials) {
            typedef __float128 Q;
            Q Dq[9] = {(Q)o.D[0] + lambda, o.D[1], o.D[2], o.D[1], (Q)o.D[3] + lambda, o.D[4], o.D[2], o.D[4], (Q)o.D[5] + lambda};
            Q c00 = Dq[4] * Dq[8] - Dq[5] * Dq[5], c01 = Dq[2] * Dq[5] - Dq[1] * Dq[8], c02 = Dq[1] * Dq[5] - Dq[2] * Dq[4];
            Q det = Dq[0] * c00 + Dq[1] * c01 + Dq[2] * c02;
            Q Iq[9];
            Iq[0] = c00 / det; Iq[1] = c01 / det; Iq[2] = c02 / det;
            Iq[4] = (Dq[0] * Dq[8] - Dq[2] * Dq[2]) / det; Iq[5] = (Dq[1] * Dq[2] - Dq[0] * Dq[5]) / det; Iq[8] = (Dq[0] * Dq[4] - Dq[1] * Dq[1]) / det;
            Iq[3] = Iq[1]; Iq[6] = Iq[2]; Iq[7] = Iq[5];
            Q Aq[9] = {o.A[0], o.A[1], o.A[2], o.A[1], o.A[3], o.A[4], o.A[2], o.A[4], o.A[5]};
            Q Cq[9], ADq[9];
            for (int r = 0; r < 3; ++r) for (int j = 0; j < 3; ++j) { Q s = 0; for (int k = 0; k < 3; ++k) s += Aq[r*3+k] * Iq[k*3+j]; ADq[r*3+j] = s; }
            for (int r = 0; r < 3; ++r) for (int j = 0; j < 3; ++j) { Q s = Aq[r*3+j]; for (int k = 0; k < 3; ++k) s -= ADq[r*3+k] * Aq[k*3+j]; Cq[r*3+j] = s; }
            Q ccq[3];
            for (int r = 0; r < 3; ++r) { Q s = o.a[r]; for (int k = 0; k < 3; ++k) s -= ADq[r*3+k] * (Q)o.b[k]; ccq[r] = s; }
            const int fi[6] = {0, 1, 2, 4, 5, 8};
            double eC = 0, mC = 0, ec = 0, mc = 0;
            for (int e = 0; e < 6; ++e) { eC = std::fmax(eC, std::fabs((double)((Q)C[e] - Cq[fi[e]]))); mC = std::fmax(mC, std::fabs((double)Cq[fi[e]])); }
            for (int r = 0; r < 3; ++r) { ec = std::fmax(ec, std::fabs((double)((Q)cc[r] - ccq[r]))); mc = std::fmax(mc, std::fabs((double)ccq[r])); }
            std::printf("DIAG l=%d fl=%x ar=%d ag=%d |A|=%.3g detD=%.3g errC=%.3g |C|=%.3g errcc=%.3g |cc|=%.3g |q|=%.3g\n", l, c.fl, c.ar, c.ag, o.A[0] + o.A[3] + o.A[5], (double)det, eC, mC, ec, mc, std::sqrt(o.q[0]*o.q[0]+o.q[1]*o.q[1]+o.q[2]*o.q[2]));
          }
          pose_terms(o.q, C, cc, true, acc);
        }
      }
      double dxs[6] = {0, 0, 0, 0, 0, 0};
      bool ok = true;
      if (prior_on)
        for (int i = 0; i < 28; ++i) acc[i] += rec[cur][i];
      double g[6];
      for (int i = 0; i < 6; ++i) g[i] = acc[21 + i];
      const double sum_ub = acc[28];
      if (qmax == 0) currentChi = acc[27];
      if (pose_active) {
        double Hs[21];
        for (int i = 0; i < 21; ++i) Hs[i] = acc[i];
        ok = ldlt6_packed(acc, g, lambda, dxs);
        if (variant & V_LDLT_REFINE) {
          // residual r = g - (H + lambda I) x, one correction
          double r[6];
          for (int i = 0; i < 6; ++i) {
            long double s = g[i];
            for (int j = 0; j < 6; ++j) {
              const double h = i <= j ? Hs[GL_U(i, j)] : Hs[GL_U(j, i)];
              s -= (long double)(h + (i == j ? lambda : 0.0)) * dxs[j];
            }
            r[i] = (double)s;
          }
          // re-use factor in acc: solve L D L^T d = r
          double y[6], d[6];
          for (int i = 0; i < 6; ++i) {
            double s = r[i];
            for (int kk = 0; kk < i; ++kk) s -= acc[GL_U(kk, i)] * y[kk];
            y[i] = s;
          }
          for (int i = 0; i < 6; ++i) y[i] /= acc[GL_U(i, i)];
          for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int kk = i + 1; kk < 6; ++kk) s -= acc[GL_U(i, kk)] * d[kk];
            d[i] = s;
          }
          for (int i = 0; i < 6; ++i) dxs[i] += d[i];
        }
      }
      const bool ok2 = ok;
      Pose Pn = P;
      if (pose_active && ok2) Pn = pose_update(P, dxs);
      // pass B
      double sum_eps2 = 0.0, chi_t = 0.0;
      for (int l = 0; l < L; ++l) {
        const Pt c = make_pt(F, S, l);
        if (!(c.ar || c.ag || c.af)) continue;
        double* p = &S.sp[(size_t)l * 3];
        double q[3], gd[3], eps[3];
        for (int j = 0; j < 3; ++j) q[j] = std::fma(P.R[j * 3], p[0], std::fma(P.R[j * 3 + 1], p[1], std::fma(P.R[j * 3 + 2], p[2], P.t[j])));
        cross(dxs, q, gd);
        gd[0] += dxs[3];
        gd[1] += dxs[4];
        gd[2] += dxs[5];
        const bool stereo = c.fl & F_STEREO;
        if (variant & V_STEP_CACHED) {
          for (int j = 0; j < 3; ++j) {
            double e = S.ucache[(size_t)l * 3 + j];
            for (int a = 0; a < 3; ++a) e -= S.adcache[(size_t)l * 9 + a * 3 + j] * gd[a];
            eps[j] = e;
          }
        } else {
          double rhs[3] = {0.0, 0.0, 0.0};
          if (c.ar) {
            const double rho1 = S.chir[l];
            const double iz = rcp_nr(q[2]);
            const double iz2 = iz * iz;
            const double c0 = -q[0] * iz2, c1 = -q[1] * iz2, c2 = std::fma(F.bn, iz2, c0);
            const double wx = rho1 * c.sx, wy = rho1 * c.sy;
            const double f0 = wx * std::fma(-c0, gd[2], std::fma(-(q[0] + gd[0]), iz, c.ob[0]));
            const double f1 = wy * std::fma(-c1, gd[2], std::fma(-(q[1] + gd[1]), iz, c.ob[1]));
            const double f2 = stereo ? wx * std::fma(-c2, gd[2], std::fma(F.bn - (q[0] + gd[0]), iz, c.ob[2])) : 0.0;
            rhs[0] = iz * (f0 + f2);
            rhs[1] = iz * f1;
            rhs[2] = std::fma(c2, f2, std::fma(c1, f1, c0 * f0));
          }
          if (c.ag) {
            if (c.fl & F_DEG) {
              const double nx = c.nd[0], ny = c.nd[1], nz = c.nd[2];
              const double eg = std::fma(nz, p[2], std::fma(ny, p[1], nx * p[0])) - c.nd[3];
              const double m = -F.lm * eg;
              for (int j = 0; j < 3; ++j) rhs[j] = std::fma(m, std::fma(P.R[j * 3 + 2], nz, std::fma(P.R[j * 3 + 1], ny, P.R[j * 3] * nx)), rhs[j]);
            } else {
              double bc[3];
              gmm_nondeg(F, c.asc, P.R, p, nullptr, bc);
              for (int j = 0; j < 3; ++j) rhs[j] += bc[j];
            }
          }
          if (c.af) fixed_rhs(F, S, P, l, p, rhs);
          if (variant & V_DINV_LDL) ldl3_solve(&S.un[(size_t)l * 6], rhs, eps);
          else sym3_mul_vec(&S.un[(size_t)l * 6], rhs, eps);
        }
        sum_eps2 += eps[0] * eps[0] + eps[1] * eps[1] + eps[2] * eps[2];
        double pn[3];
        for (int j = 0; j < 3; ++j) pn[j] = p[j] + (P.R[j] * eps[0] + P.R[3 + j] * eps[1] + P.R[6 + j] * eps[2]);
        for (int j = 0; j < 3; ++j) {
          S.bk[(size_t)l * 3 + j] = p[j];
          p[j] = pn[j];
        }
        double chi = 0.0;
        if (c.ar) {
          double qn[3], e[3], iz;
          for (int j = 0; j < 3; ++j) qn[j] = std::fma(Pn.R[j * 3], pn[0], std::fma(Pn.R[j * 3 + 1], pn[1], std::fma(Pn.R[j * 3 + 2], pn[2], Pn.t[j])));
          const double c2 = reproj_n(qn, c.ob, stereo, F.bn, c.sx, c.sy, e, iz);
          S.chir[l] = c2;
          double r0 = c2, r1;
          if (robust) huber_bf(c2, stereo ? F.dstereo : F.dmono, stereo ? F.dstereo * F.dstereo : F.dmono * F.dmono, r0, r1);
          chi = r0;
        }
        if (c.ag) chi += gmm_chi2_fast(F, c.nd, c.fl, c.asc, pn);
        if (c.af) chi += fixed_chi(F, S, l, pn, robust);
        chi_t += chi;
      }
      if (prior_on) prior_record(mi, Pn, rec[cur ^ 1]);
      double scale = lambda * sum_eps2 + sum_ub;
      const double tempChi = ok2 ? (prior_on ? chi_t + rec[cur ^ 1][27] : chi_t) : 1.7976931348623157e308;
      if (pose_active)
        for (int i = 0; i < 6; ++i) scale += dxs[i] * (lambda * dxs[i] + g[i]);
      scale += 1e-3;
      rho = (currentChi - tempChi) / scale;
      if (trace && trials < trace_cap) {
        double* t = trace + (size_t)trials * 10;
        t[0] = currentChi;
        t[1] = tempChi;
        t[2] = lambda;
        t[3] = rho;
        for (int i = 0; i < 6; ++i) t[4 + i] = dxs[i];
      }
      if (rho > 0 && std::isfinite(tempChi)) {
        const double uu = 2 * rho - 1;
        double alpha = 1. - uu * uu * uu;
        alpha = std::fmin(alpha, 2. / 3.);
        lambda *= std::fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        P = Pn;
        cur ^= 1;
      } else {
        lambda *= ni;
        ni *= 2;
        for (int l = 0; l < L; ++l) {
          const Pt c = make_pt(F, S, l);
          if (c.ar || c.ag || c.af)
            for (int j = 0; j < 3; ++j) S.sp[(size_t)l * 3 + j] = S.bk[(size_t)l * 3 + j];
        }
      }
      qmax++;
      ++trials;
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (qmax == 10 || rho == 0) break;
  }
  return cj;
}

}  // namespace

// pose_io: 7 (qx qy qz qw tx ty tz); pts_io: L x 3; obn: L x 3 normalised; fl: L flag words; assoc: gated associations;
// trace: cap x 10 (currentChi, tempChi, lambda, rho, dx[6]); returns the number of Levenberg trials
extern "C" void emul_set_diag(int t) { g_diag_trial = t; }
namespace {
int g_NF = 0;
const double *g_fRt = nullptr, *g_fobn = nullptr;
const int32_t* g_foct = nullptr;
uint8_t* g_ferase = nullptr;
}  // namespace
// fixed observer key-frames of the NEXT emul_track call: poses {R (9), t (3)} x NF, normalised observations L x NF x 3 (third
// component < -1e29: mono), octaves L x NF (< 0: not observed), erase flags out L x NF
extern "C" void emul_set_fixed(int NF, const double* fRt, const double* fobn, const int32_t* foct, uint8_t* ferase) {
  g_NF = NF;
  g_fRt = fRt;
  g_fobn = fobn;
  g_foct = foct;
  g_ferase = ferase;
}
extern "C" int emul_track(int L, double* pose_io, double* pts_io, const double* obn, const int32_t* fl, const int32_t* assoc,
                          const double* plane4, const double* hgw, const double* mean, const double* sx, const double* sy, double bn,
                          double lm, double str_thresh, double dmono, double dstereo, int has_prior, int variant, double* trace,
                          int trace_cap, int32_t* fl_out) {
  Frame F;
  F.L = L;
  F.obn = obn;
  F.fl0 = fl;
  F.assoc = assoc;
  F.plane4 = plane4;
  F.hgw = hgw;
  F.mean = mean;
  for (int i = 0; i < 8; ++i) {
    F.sx[i] = sx[i];
    F.sy[i] = sy[i];
  }
  F.bn = bn;
  F.lm = lm;
  F.str_thresh = str_thresh;
  F.dmono = dmono;
  F.dstereo = dstereo;
  F.NF = g_NF;
  F.fRt = g_fRt;
  F.fobn = g_fobn;
  F.foct = g_foct;
  State S;
  S.chif.assign((size_t)L * (g_NF > 0 ? g_NF : 1), 0.0);
  S.levf.assign((size_t)L * (g_NF > 0 ? g_NF : 1), 0);
  S.sp.assign(pts_io, pts_io + (size_t)L * 3);
  S.chir.assign(L, 0.0);
  S.un.assign((size_t)L * 6, 0.0);
  S.bk.assign((size_t)L * 3, 0.0);
  S.ucache.assign((size_t)L * 3, 0.0);
  S.adcache.assign((size_t)L * 9, 0.0);
  S.fl.assign(fl, fl + L);
  Pose P;
  qtoR(pose_io, P.R);
  P.t[0] = pose_io[4];
  P.t[1] = pose_io[5];
  P.t[2] = pose_io[6];
  // inverse measurement {R^T, -R^T t} of the input pose
  double mi[12];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) mi[i * 3 + j] = P.R[j * 3 + i];
  for (int i = 0; i < 3; ++i) mi[9 + i] = -(mi[i * 3] * P.t[0] + mi[i * 3 + 1] * P.t[1] + mi[i * 3 + 2] * P.t[2]);
  int trials = 0;
  for (int phase = 0; phase < 3; ++phase) {
    optimize_fast(F, S, P, phase < 2, phase < 2 ? 5 : 40, trials, has_prior != 0, mi, variant, trace, trace_cap);
    if (phase == 2) break;
    for (int l = 0; l < L; ++l) {
      const int f = S.fl[l];
      const double* p = &S.sp[(size_t)l * 3];
      if (phase == 0) {
        if ((f & (F_ASSOC | F_DEG)) == (F_ASSOC | F_DEG)) {
          const double* nd = plane4 + (size_t)assoc[l] * 4;
          if (gmm_chi2_fast(F, nd, f, -1, p) > str_thresh) S.fl[l] |= F_LEVG;
        }
      } else {
        if (!(f & F_EXISTS)) continue;
        const double z = P.R[6] * p[0] + P.R[7] * p[1] + P.R[8] * p[2] + P.t[2];
        if (S.chir[l] > ((f & F_STEREO) ? 7.815 : 5.991) || !(z > 0.0)) S.fl[l] |= F_LEVR;
        for (int k = 0; k < F.NF; ++k) {  // the fixed observers' edges: same gate, depth in THEIR camera
          if (F.foct[(size_t)l * F.NF + k] < 0) continue;
          const double* Rf = F.fRt + (size_t)k * 12;
          const double zf = Rf[6] * p[0] + Rf[7] * p[1] + Rf[8] * p[2] + Rf[11];
          const bool st = !(F.fobn[((size_t)l * F.NF + k) * 3 + 2] < -1e29);
          if (S.chif[(size_t)l * F.NF + k] > (st ? 7.815 : 5.991) || !(zf > 0.0)) S.levf[(size_t)l * F.NF + k] = 1;
        }
      }
    }
  }
  if (g_ferase)
    for (int l = 0; l < L; ++l)
      for (int k = 0; k < F.NF; ++k) {
        uint8_t er = 0;
        if ((S.fl[l] & F_EXISTS) && F.foct[(size_t)l * F.NF + k] >= 0) {
          const double* p = &S.sp[(size_t)l * 3];
          const double* Rf = F.fRt + (size_t)k * 12;
          const double zf = Rf[6] * p[0] + Rf[7] * p[1] + Rf[8] * p[2] + Rf[11];
          const bool st = !(F.fobn[((size_t)l * F.NF + k) * 3 + 2] < -1e29);
          if (S.chif[(size_t)l * F.NF + k] > (st ? 7.815 : 5.991) || !(zf > 0.0)) er = 1;
        }
        g_ferase[(size_t)l * F.NF + k] = er;
      }
  double q[4];
  qfromR(P.R, q);
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double sgn = q[3] < 0 ? -1.0 : 1.0;
  for (int i = 0; i < 4; ++i) pose_io[i] = sgn * q[i] / n;
  for (int i = 0; i < 3; ++i) pose_io[4 + i] = P.t[i];
  std::memcpy(pts_io, S.sp.data(), sizeof(double) * (size_t)L * 3);
  if (fl_out)
    for (int l = 0; l < L; ++l) fl_out[l] = S.fl[l];
  return trials;
}
