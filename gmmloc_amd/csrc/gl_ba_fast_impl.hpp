// (included by gl_ba_fast.hip once per block shape: GL_BAF_NS / GL_BAF_TF / GL_BAF_MCAP)
// On-chip fast path of the single-pose structure-constrained refinement (same algorithm and
// control flow as k_ba1 in gl_ba.hip, which stays as the general / large-M path and as the
// A/B reference).  For M <= 2000 points per frame:
//   * 512 threads (8 waves, 2 per SIMD) per frame; the frame's mutable state lives in LDS as
//     SoA for the whole 5/5/40 schedule: current point (3 fp64), stale chi2 (1 fp64) and a
//     12-word slot with two lifetimes -- between the two passes of a Levenberg trial it caches
//     the per-point solve in fp32 {u = D^-1 b, A D^-1}, so pass B computes the point step
//     eps = u - (A D^-1)^T (omega x q + upsilon) without re-linearising (the step only needs
//     ~1e-7 relative accuracy, the trial state is then evaluated exactly in fp64; D^-1 itself
//     is NOT cached: its 1/lambda eigenvalue along an unconstrained ray would swamp fp32);
//     after pass B it holds the backup of the point while the trial point sits in place
//     (nothing to copy on acceptance).  80 B/point = 160 000 B + 2.6 KB of reduction scratch
//     of the CU's 160 KB.  Flag / level / octave bits of a thread's 4 points are a 64-bit
//     register word.  A trial touches global memory only for the read-only observations and
//     plane records (56 B/point per pass, coalesced); measured (cache-hot substitute) their
//     latency is fully hidden by the second wave of the SIMD;
//   * computeScale is evaluated as  lambda (sum|eps|^2 + |dx|^2) + sum u.b + dx.g  (the b-terms
//     of the point blocks collapse onto the reduced rhs g), so pass B needs neither b nor A;
//   * the pose is kept as (R, t) in SGPRs (v_readfirstlane after the solve: there is no scalar
//     fp64 ALU, uniform results otherwise occupy VGPRs) and updated by Rodrigues directly
//     (short series for |theta| < 0.01); reciprocals / inverse square roots use v_rcp_f64 /
//     v_rsq_f64 + two Newton steps instead of the IEEE division sequence; Huber is branch-free;
//   * fp64 needs 2 VGPRs per value, so the per-point working set (not the data) is what limits
//     occupancy: one point at a time per thread in rolled loops, no spills in the trial loop;
//   * two-level deterministic reduction: wave reduce-scatter (v_permlane32/16_swap + DPP) ->
//     LDS -> 32 lanes sum the 8 wave partials -> only the solving wave reads the totals back;
//     the 6x6 LDL^T + exp() run on wave 0 and are broadcast through LDS.
// The kernel is VALU-issue bound (every wave64 instruction holds its SIMD for 4 cycles; ~3100
// instructions per wave and trial at 2 waves/SIMD + ~450 on the serial solve).  Measured and
// dropped: software prefetch (spills), alternating s_setprio between the two waves of a SIMD
// (-1.4 %), trial pose through LDS, 256-thread blocks; see DESIGN.md section 8.
// Results agree with k_ba1 / the oracle to the north-star tolerance (tests/test_gpu_track.py).
namespace {
namespace GL_BAF_NS {

constexpr int TF = GL_BAF_TF;      // threads per frame
constexpr int NWF = TF / 64;
constexpr int PPTF = 4;            // point slots per thread (rolled loop)
constexpr int MCAP = GL_BAF_MCAP;  // LDS capacity in points: 80 B/point (+ 2.6 KB reduction / broadcast)

#ifdef GL_BA_PROF
__device__ unsigned long long g_prof[16];
#endif
#ifndef PROF_T
#ifdef GL_BA_PROF
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) if (blockIdx.x == 0 && threadIdx.x == 0) g_prof[slot] += (unsigned long long)((t1) - (t0))
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif
#endif

enum { F_EXISTS = 1, F_STEREO = 2, F_ASSOC = 4, F_DEG = 8, F_LEVR = 16, F_LEVG = 32 };

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
// branch-free g2o::RobustKernelHuber (rho, rho')
GL_DEV void huber_bf(double e, double delta, double dsqr, double& rho0, double& rho1) {
  const double r = rsq_nr(fmax(e, 1e-300));
  const bool in = e <= dsqr;
  rho1 = in ? 1.0 : delta * r;
  rho0 = in ? e : (2.0 * delta * (e * r) - dsqr);
}

// wave-uniform value -> SGPR pair (there is no scalar fp64 ALU on gfx950: uniform fp64 results sit
// in VGPRs unless moved explicitly; the pose / step are read by every per-point FMA)
GL_DEV double uni(double v) {
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

struct Pose {  // T_cw as rotation matrix + translation
  double R[9], t[3];
};

GL_DEV Pose pose_uni(const Pose& P) {
  Pose U;
#pragma unroll
  for (int i = 0; i < 9; ++i) U.R[i] = uni(P.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) U.t[i] = uni(P.t[i]);
  return U;
}
GL_DEV Pose pose_from_se3(const SE3& T) {
  Pose P;
  qtoR(T.r, P.R);
  P.t[0] = T.t[0];
  P.t[1] = T.t[1];
  P.t[2] = T.t[2];
  return pose_uni(P);
}
// exp(dx) * P  (g2o SE3Quat::exp, VertexSE3Expmap::oplusImpl) without the quaternion detour
GL_DEV Pose pose_update(const Pose& P, const double* u) {
  const double th2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
  double a, b, c;
  if (th2 < 1e-4) {
    // a = sin t/t, b = (1-cos t)/t^2, c = (t-sin t)/t^3 as series in t^2; for |t| < 0.01 (every LM step but
    // the first few) four terms reach 1e-19, and this also covers g2o's small-angle branch; avoids
    // sqrt, sincos and three divisions on the serial path between the two passes
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
  Pose N;
  mm3(dR, P.R, N.R);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    N.t[i] = dR[i * 3] * P.t[0] + dR[i * 3 + 1] * P.t[1] + dR[i * 3 + 2] * P.t[2] + V[i * 3] * u[3] + V[i * 3 + 1] * u[4] +
             V[i * 3 + 2] * u[5];
  return N;
}

struct Lds {      // per-frame state, SoA over MCAP points
  double* sp;     // 3 x MCAP  current point (world)
  double* chir;   // MCAP      stale chi2 of the reprojection edge (e->chi2())
  // 12 x MCAP 32-bit words, two lifetimes sharing one slot per point:
  //   pass A -> pass B : fp32 {u = D^-1 b (3), A D^-1 (3x3)} -- the per-point solve, so pass B does not
  //                      re-linearise: eps = u - (A D^-1)^T gd.  Both are O(1)-conditioned (unlike D^-1
  //                      itself, whose 1/lambda eigenvalue along an unconstrained ray would swamp fp32),
  //                      and the step only needs ~1e-7 relative accuracy: the trial state it produces is
  //                      evaluated exactly in fp64;
  //   pass B -> accept : backup of the point (3 x {lo, hi} words, exact fp64) while the trial point sits
  //                      in `sp`; restored only when the trial is rejected.
  int* un;
  double* stab;   // 8: 1/sigma^2 per pyramid octave
};
// per-point flag bits + octave (bits 8..10) live in REGISTERS: thread t owns points t + i*TF, i < 4,
// 16 bits each in one 64-bit word
typedef unsigned long long FlagW;
GL_DEV int fw_get(FlagW fw, int i) { return (int)((fw >> (16 * i)) & 0xffffull); }
GL_DEV void fw_or(FlagW& fw, int i, int bits) { fw |= (FlagW)bits << (16 * i); }

struct Lin {
  double q[3];
  double A[6];   // reprojection block w Jpi^T Jpi (camera frame, sym6; A[1] == 0)
  double a[3];   // its rhs
  double D[6];   // A + GMM block (no damping yet)
  double b[3];   // a + GMM rhs
  double rho0_r, chi_g;
};

GL_DEV double reproj_chi2(const BaK& k, const double* q, const double* ob, bool stereo, double s, double* e,
                          double& iz) {
  const double ou = ob[0], ov = ob[1], our = ob[2];
  iz = rcp_nr(q[2]);
  const double pu = q[0] * iz * k.fx + k.cx, pv = q[1] * iz * k.fy + k.cy;
  e[0] = ou - pu;
  e[1] = ov - pv;
  e[2] = stereo ? (our - (pu - k.bf * iz)) : 0.0;
  return s * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
}

// GMM edge of a NON-degenerate component (rare): EdgePt2Gaussian, e = L^T d, J = L^T, so
// J^T J = L L^T (precomputed per component, world frame), b = -L L^T d, chi2 = d^T L L^T d.
GL_DEV double gmm_nondeg(const GmmDev& gm, int a, const double* R, const double* p, double* Hc, double* bc) {
  double Hg[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) Hg[i] = gm.hgw[(size_t)a * 6 + i];
  const double d[3] = {p[0] - gm.rec12[(size_t)a * 12], p[1] - gm.rec12[(size_t)a * 12 + 1],
                       p[2] - gm.rec12[(size_t)a * 12 + 2]};
  double Hd[3];
  sym3_mul_vec(Hg, d, Hd);
  const double chi = d[0] * Hd[0] + d[1] * Hd[1] + d[2] * Hd[2];
  if (Hc) {
    // Hc = R Hg R^T, bc = -R Hg d
    double RH[9];
#pragma unroll
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
#pragma unroll
    for (int i = 0; i < 3; ++i) bc[i] = -(R[i * 3] * Hd[0] + R[i * 3 + 1] * Hd[1] + R[i * 3 + 2] * Hd[2]);
  }
  return chi;
}

// linearise point l at pose P and world point p; returns the un-robustified chi2_r
GL_DEV double lin_fast(const BaK& k, const GmmDev& gm, const Pose& P, const double* nd, int fl, int asc, double s,
                       const double* ob, const double* p, bool robust, Lin& o) {
  const bool act_r = !(fl & F_LEVR), act_g = (fl & F_ASSOC) && !(fl & F_LEVG);
#pragma unroll
  for (int c = 0; c < 3; ++c) o.q[c] = P.R[c * 3] * p[0] + P.R[c * 3 + 1] * p[1] + P.R[c * 3 + 2] * p[2] + P.t[c];
#pragma unroll
  for (int c = 0; c < 6; ++c) o.A[c] = 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) o.a[c] = 0.0;
  o.rho0_r = 0.0;
  o.chi_g = 0.0;
  double chi_r = 0.0;
  if (act_r) {
    const bool stereo = fl & F_STEREO;
    double e[3], iz;
    chi_r = reproj_chi2(k, o.q, ob, stereo, s, e, iz);
    double rho1 = 1.0;
    o.rho0_r = chi_r;
    if (robust) {
      const double dl = stereo ? k.delta_stereo : k.delta_mono;
      huber_bf(chi_r, dl, dl * dl, o.rho0_r, rho1);
    }
    // rows of Jpi: (al, 0, b0), (0, ga, b1) and, stereo only, (al, 0, b2); t = w on the stereo row, else 0
    const double w = rho1 * s;
    const double t = stereo ? w : 0.0;
    const double iz2 = iz * iz;
    const double al = k.fx * iz, ga = k.fy * iz;
    const double b0 = -k.fx * o.q[0] * iz2, b1 = -k.fy * o.q[1] * iz2;
    const double b2 = fma(k.bf, iz2, b0);
    const double wga = w * ga, tb2 = t * b2;
    o.A[0] = (w + t) * al * al;
    o.A[2] = al * fma(t, b2, w * b0);
    o.A[3] = wga * ga;
    o.A[4] = wga * b1;
    o.A[5] = fma(tb2, b2, w * fma(b1, b1, b0 * b0));
    o.a[0] = al * fma(t, e[2], w * e[0]);
    o.a[1] = wga * e[1];
    o.a[2] = fma(tb2, e[2], w * fma(b1, e[1], b0 * e[0]));
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) o.D[c] = o.A[c];
#pragma unroll
  for (int c = 0; c < 3; ++c) o.b[c] = o.a[c];
  if (act_g) {
    if (fl & F_DEG) {  // EdgePt2GaussianDeg x ba_lambda2: D += lm n' n'^T, b -= lm eg n'  (n' = R n)
      const double nx = nd[0], ny = nd[1], nz = nd[2];
      const double eg = fma(nz, p[2], fma(ny, p[1], nx * p[0])) - nd[3];
      const double lm = k.ba_lambda2;
      o.chi_g = eg * (lm * eg);
      double nc[3], tn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        nc[c] = fma(P.R[c * 3 + 2], nz, fma(P.R[c * 3 + 1], ny, P.R[c * 3] * nx));
        tn[c] = lm * nc[c];
      }
      o.D[0] = fma(tn[0], nc[0], o.D[0]);
      o.D[1] = fma(tn[0], nc[1], o.D[1]);
      o.D[2] = fma(tn[0], nc[2], o.D[2]);
      o.D[3] = fma(tn[1], nc[1], o.D[3]);
      o.D[4] = fma(tn[1], nc[2], o.D[4]);
      o.D[5] = fma(tn[2], nc[2], o.D[5]);
#pragma unroll
      for (int c = 0; c < 3; ++c) o.b[c] = fma(-eg, tn[c], o.b[c]);
    } else {
      double Hc[6], bc[3];
      o.chi_g = gmm_nondeg(gm, asc, P.R, p, Hc, bc);
#pragma unroll
      for (int c = 0; c < 6; ++c) o.D[c] += Hc[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) o.b[c] += bc[c];
    }
  }
  return chi_r;
}

GL_DEV double gmm_chi2_fast(const BaK& k, const GmmDev& gm, const double* nd, int fl, int asc, const double* p) {
  if (fl & F_DEG) {
    const double eg = (nd[0] * p[0] + nd[1] * p[1] + nd[2] * p[2]) - nd[3];
    return eg * (k.ba_lambda2 * eg);
  }
  return gmm_nondeg(gm, asc, nullptr, p, nullptr, nullptr);
}

GL_DEV void sym3_inv_fast(const double* S, double* I) {
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

GL_DEV void point_solve_fast(const Lin& o, double lambda, double* Dinv, double* u) {
  const double D[6] = {o.D[0] + lambda, o.D[1], o.D[2], o.D[3] + lambda, o.D[4], o.D[5] + lambda};
  sym3_inv_fast(D, Dinv);
  sym3_mul_vec(Dinv, o.b, u);
}

// acc[0..20] += upper(G^T C G), acc[21..26] += G^T c,  C symmetric (sym6)  (callers pass a zeroed v)
GL_DEV void accum_pose_sym(const double* q, const double* C, const double* c, double* acc) {
  // M = [q]x C (row r, column j) = (q x C[:, j])[r];  TL = [q]x C [q]x^T accumulated straight into acc
  // with two FMAs per entry (explicit: contraction alone would leave mul + fma + add)
  const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
  double M[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[j] = fma(q[1], Cf[6 + j], -q[2] * Cf[3 + j]);
    M[3 + j] = fma(q[2], Cf[j], -q[0] * Cf[6 + j]);
    M[6 + j] = fma(q[0], Cf[3 + j], -q[1] * Cf[j]);
  }
  acc[0] = fma(q[1], M[2], fma(-q[2], M[1], acc[0]));
  acc[1] = fma(q[2], M[0], fma(-q[0], M[2], acc[1]));
  acc[2] = fma(q[0], M[1], fma(-q[1], M[0], acc[2]));
  acc[3] += M[0];
  acc[4] += M[1];
  acc[5] += M[2];
  acc[6] = fma(q[2], M[3], fma(-q[0], M[5], acc[6]));
  acc[7] = fma(q[0], M[4], fma(-q[1], M[3], acc[7]));
  acc[8] += M[3];
  acc[9] += M[4];
  acc[10] += M[5];
  acc[11] = fma(q[0], M[7], fma(-q[1], M[6], acc[11]));
  acc[12] += M[6];
  acc[13] += M[7];
  acc[14] += M[8];
  acc[15] += C[0];
  acc[16] += C[1];
  acc[17] += C[2];
  acc[18] += C[3];
  acc[19] += C[4];
  acc[20] += C[5];
  acc[21] = fma(q[1], c[2], fma(-q[2], c[1], acc[21]));
  acc[22] = fma(q[2], c[0], fma(-q[0], c[2], acc[22]));
  acc[23] = fma(q[0], c[1], fma(-q[1], c[0], acc[23]));
  acc[24] += c[0];
  acc[25] += c[1];
  acc[26] += c[2];
}

// C = A - (A Dinv) A (symmetric sym6), AD = A Dinv (3x3)
GL_DEV void schur_C(const double* A, const double* AD, double* C) {
  const double Af[9] = {A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5]};
  const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const int i = ij[e][0], j = ij[e][1];
    C[e] = fma(-AD[i * 3], Af[j], fma(-AD[i * 3 + 1], Af[3 + j], fma(-AD[i * 3 + 2], Af[6 + j], A[e])));
  }
}

// ---- latency shape (GL_BAF_COOPERATIVE): the points of ONE frame are dealt to NB workgroups -------------------
// Every reduction of the optimiser then has a second level across the NB workgroups (gld::coop_totals: tagged
// words, no barrier); all workgroups hold the same totals, so the Levenberg control flow (accept / reject,
// lambda, termination) is identical in all of them and each repeats the 6x6 solve on its own.
#ifdef GL_BAF_COOPERATIVE
constexpr bool kCoop = true;
#else
constexpr bool kCoop = false;
#endif
// two-level deterministic workgroup reduction (all threads get the NV totals)
template <int NV>
GL_DEV void reduce2(double* v, double* red, double* tot, Coop& C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = NV; i < 32; ++i) v[i] = 0.0;
  const double r = wave_reduce_scatter32(v);
  // no barrier needed before writing `red`: its last readers (threads < 32) finished before the
  // closing barrier of the previous reduction, which every thread has passed
  if (wave_slot_owner(lane)) red[wave * 32 + wave_slot(lane)] = r;
  __syncthreads();
  if (threadIdx.x < 32) {
    double s = red[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NWF; ++w) s += red[w * 32 + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (kCoop && C.NB > 1) coop_totals<false>(C, tot);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = uni(tot[i]);
}
// same, but only wave 0 (the one that solves the reduced system) reads the totals back
template <int NV>
GL_DEV void reduce2_w0(double* v, double* red, double* tot, Coop& C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = NV; i < 32; ++i) v[i] = 0.0;
  const double r = wave_reduce_scatter32(v);
  if (wave_slot_owner(lane)) red[wave * 32 + wave_slot(lane)] = r;
  __syncthreads();
  if (threadIdx.x < 32) {
    double s = red[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NWF; ++w) s += red[w * 32 + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (kCoop && C.NB > 1) coop_totals<false>(C, tot);
  if (threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = uni(tot[i]);
  }
}
GL_DEV double reduce_max(double v, double* red, double* tot, Coop& C) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmax(v, shfl_xor_f64(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = red[0];
#pragma unroll
  for (int w = 1; w < NWF; ++w) m = fmax(m, red[w]);
  if (kCoop && C.NB > 1) {
    __syncthreads();  // tot may still be read from the previous reduction
    if (threadIdx.x < 32) tot[threadIdx.x] = m;
    __syncthreads();
    coop_totals<true>(C, tot);
    m = uni(tot[0]);
  }
  return m;
}

// 6x6 LDL^T in place on the packed upper triangle (21 values, row-major i <= j as produced by the
// reduction), reciprocal pivots; SimplicialLDLT semantics: fail on a zero pivot.
// Packed index of (i, j), i <= j.
#ifndef GL_U
#define GL_U(i, j) ((i) * 6 - (i) * ((i)-1) / 2 + ((j) - (i)))
#endif
GL_DEV bool ldlt6_packed(double* a, const double* b, double lambda, double* x) {
  // a(i,j), i<=j holds H(j,i) = H(i,j).  Column-oriented: l_ij (i > j) stored at a(j,i).
  double iD[6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = a[GL_U(j, j)] + lambda;
#pragma unroll
    for (int kk = 0; kk < j; ++kk) d -= a[GL_U(kk, j)] * a[GL_U(kk, j)] * a[GL_U(kk, kk)];
    if (d == 0.0 || !isfinite(d)) ok = false;
    a[GL_U(j, j)] = d;
    iD[j] = rcp_nr(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = a[GL_U(j, i)];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) s -= a[GL_U(kk, i)] * a[GL_U(kk, j)] * a[GL_U(kk, kk)];
      a[GL_U(j, i)] = s * iD[j];
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int kk = 0; kk < i; ++kk) s -= a[GL_U(kk, i)] * y[kk];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= iD[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int kk = i + 1; kk < 6; ++kk) s -= a[GL_U(i, kk)] * x[kk];
    x[i] = s;
  }
  return ok;
}

// per-point context loaded at the top of a loop iteration
struct PtCtx {
  int l, fl, asc;
  double s;
  double ob[3], nd[4], p[3];
  bool ar, ag;
};
// observations and plane records come from global memory (read-only, coalesced, L2-resident).
// (Software-prefetching slot i+1 was measured: it costs 14 VGPRs -> 6 spilled registers and
// ~1 GB of scratch writes per launch for no gain; the second wave of the SIMD hides the latency.)
GL_DEV bool load_pt(const Lds& D, FlagW fw, const double* __restrict__ gobs, const double* __restrict__ gnd,
                    const int32_t* __restrict__ gassoc, int L, int i, PtCtx& c) {
  c.l = threadIdx.x + i * TF;
  {  // issue the observation loads first (clamped index): they overlap the LDS reads / flag tests below
    const int lc = min(c.l, L - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) c.ob[j] = gobs[(size_t)lc * 3 + j];
  }
  if (c.l >= L) return false;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.nd[j] = gnd[(size_t)c.l * 4 + j];  // plane normal n and n . mean
  c.fl = fw_get(fw, i);
  if (!(c.fl & F_EXISTS)) return false;
  c.ar = !(c.fl & F_LEVR);
  c.ag = (c.fl & F_ASSOC) && !(c.fl & F_LEVG);
  if (!(c.ar || c.ag)) return false;
  c.s = D.stab[(c.fl >> 8) & 7];
  c.asc = (c.fl & F_ASSOC) && !(c.fl & F_DEG) ? gassoc[c.l] : -1;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.p[j] = D.sp[j * MCAP + c.l];
  return true;
}

// SparseOptimizer::optimize(iters), Levenberg
GL_DEV int optimize_fast(const BaK& k, const GmmDev& gm, const Lds& D, FlagW fw, Pose& P, int L, const double* __restrict__ gobs,
                         const int32_t* __restrict__ gassoc, const double* __restrict__ gnd, bool robust, int iters,
                         double* red, double* tot, int& trials, Coop& C) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
#pragma unroll 1
  for (int i = 0; i < PPTF; ++i) {
    const int l = threadIdx.x + i * TF;
    if (l >= L) break;
    const int fl = fw_get(fw, i);
    if (!(fl & F_EXISTS)) continue;
    const bool ar = !(fl & F_LEVR), ag = (fl & F_ASSOC) && !(fl & F_LEVG);
    if (ar) acc[0] += 1.0;
    if (ar || ag) acc[1] += 1.0;
  }
  reduce2<2>(acc, red, tot, C);
  const bool pose_active = acc[0] > 0.0;
  if (!pose_active && !(acc[1] > 0.0)) return -1;

  double lambda = 0.0, ni = 2.0;
  int cj = 0;
  for (int it = 0; it < iters; ++it) {
    double rho = 0.0, currentChi = 0.0;
    int qmax = 0;
    if (it == 0) {  // computeLambdaInit
      double md = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
#pragma unroll 1
      for (int i = 0; i < PPTF; ++i) {
        PtCtx c;
        if (!load_pt(D, fw, gobs, gnd, gassoc, L, i, c)) continue;
        Lin o;
        lin_fast(k, gm, P, c.nd, c.fl, c.asc, c.s, c.ob, c.p, robust, o);
        const double Hf[9] = {o.D[0], o.D[1], o.D[2], o.D[1], o.D[3], o.D[4], o.D[2], o.D[4], o.D[5]};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += P.R[a * 3 + j] * Hf[a * 3 + b] * P.R[b * 3 + j];
          md = fmax(md, fabs(s));
        }
        if (c.ar) {
          const double zero[3] = {0, 0, 0};
          accum_pose_sym(o.q, o.A, zero, acc);
        }
      }
      reduce2<21>(acc, red, tot, C);
      if (pose_active) {
#pragma unroll
        for (int i = 0; i < 6; ++i) md = fmax(md, fabs(acc[GL_U(i, i)]));
      }
      md = reduce_max(md, red, tot, C);
      lambda = uni(1e-5 * md);
      ni = 2.0;
    }
    do {
      PROF_T(tA0);
      // ---- pass A ---------------------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
#pragma unroll 1
      for (int i = 0; i < PPTF; ++i) {
        PtCtx c;
        if (!load_pt(D, fw, gobs, gnd, gassoc, L, i, c)) continue;
        Lin o;
        const double c2 = lin_fast(k, gm, P, c.nd, c.fl, c.asc, c.s, c.ob, c.p, robust, o);
        if (c.ar) D.chir[c.l] = c2;  // computeActiveErrors
        acc[27] += o.rho0_r + o.chi_g;
        double Dinv[6], u[3];
        point_solve_fast(o, lambda, Dinv, u);
        acc[28] = fma(u[0], o.b[0], fma(u[1], o.b[1], fma(u[2], o.b[2], acc[28])));
#pragma unroll
        for (int j = 0; j < 3; ++j) D.un[j * MCAP + c.l] = __float_as_int((float)u[j]);
        if (c.ar) {
          double C[6], cc[3], AD[9];
          sym3_mul(o.A, Dinv, AD);
#pragma unroll
          for (int j = 0; j < 9; ++j) D.un[(3 + j) * MCAP + c.l] = __float_as_int((float)AD[j]);
          schur_C(o.A, AD, C);
          cc[0] = fma(-o.A[0], u[0], fma(-o.A[1], u[1], fma(-o.A[2], u[2], o.a[0])));
          cc[1] = fma(-o.A[1], u[0], fma(-o.A[3], u[1], fma(-o.A[4], u[2], o.a[1])));
          cc[2] = fma(-o.A[2], u[0], fma(-o.A[4], u[1], fma(-o.A[5], u[2], o.a[2])));
          accum_pose_sym(o.q, C, cc, acc);
        } else {
#pragma unroll
          for (int j = 0; j < 9; ++j) D.un[(3 + j) * MCAP + c.l] = 0;
        }
      }
      PROF_T(tA1);
      reduce2_w0<29>(acc, red, tot, C);
      PROF_T(tA2);
      if (qmax == 0) currentChi = uni(tot[27]);
      // 6x6 solve + exp(dx) by wave 0 only; step, trial pose and status are broadcast through LDS
      double* bc = tot + 32;  // 20 doubles: dx[6] R[9] t[3] ok pad
      if (threadIdx.x < 64) {
        double dxs[6] = {0, 0, 0, 0, 0, 0};
        bool ok = true;
        if (pose_active) ok = ldlt6_packed(acc, acc + 21, lambda, dxs);
        Pose Pw = P;
        if (pose_active && ok) Pw = pose_update(P, dxs);
        if (threadIdx.x == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) bc[i] = dxs[i];
#pragma unroll
          for (int i = 0; i < 9; ++i) bc[6 + i] = Pw.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) bc[15 + i] = Pw.t[i];
          bc[18] = ok ? 1.0 : 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) bc[19 + i] = acc[21 + i];  // reduced rhs g and sum u.b, for computeScale
          bc[25] = acc[28];
        }
      }
      __syncthreads();
      double dx[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) dx[i] = uni(bc[i]);
      Pose Pn;
#pragma unroll
      for (int i = 0; i < 9; ++i) Pn.R[i] = uni(bc[6 + i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) Pn.t[i] = uni(bc[15 + i]);
      const bool ok2 = uni(bc[18]) != 0.0;
      PROF_T(tS);
      // ---- pass B ---------------------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
#pragma unroll 1
      for (int i = 0; i < PPTF; ++i) {
        PtCtx c;
        if (!load_pt(D, fw, gobs, gnd, gassoc, L, i, c)) continue;
        // eps = D^-1 (b - A gd) = u - (A D^-1)^T gd,  gd = omega x q + upsilon  (u, A D^-1: pass-A cache)
        double q[3], gd[3], eps[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) q[j] = P.R[j * 3] * c.p[0] + P.R[j * 3 + 1] * c.p[1] + P.R[j * 3 + 2] * c.p[2] + P.t[j];
        cross(dx, q, gd);
        gd[0] += dx[3];
        gd[1] += dx[4];
        gd[2] += dx[5];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double e = (double)__int_as_float(D.un[j * MCAP + c.l]);
#pragma unroll
          for (int a = 0; a < 3; ++a) e -= (double)__int_as_float(D.un[(3 + a * 3 + j) * MCAP + c.l]) * gd[a];
          eps[j] = e;
        }
        acc[0] += eps[0] * eps[0] + eps[1] * eps[1] + eps[2] * eps[2];
        double pn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) pn[j] = c.p[j] + (P.R[j] * eps[0] + P.R[3 + j] * eps[1] + P.R[6 + j] * eps[2]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // the trial point goes in place; the old point is backed up in the cache slot
          D.sp[j * MCAP + c.l] = pn[j];
          D.un[(2 * j) * MCAP + c.l] = __double2loint(c.p[j]);
          D.un[(2 * j + 1) * MCAP + c.l] = __double2hiint(c.p[j]);
        }
        if (c.ar) {
          double qn[3], e[3], iz;
#pragma unroll
          for (int j = 0; j < 3; ++j) qn[j] = Pn.R[j * 3] * pn[0] + Pn.R[j * 3 + 1] * pn[1] + Pn.R[j * 3 + 2] * pn[2] + Pn.t[j];
          const bool stereo = c.fl & F_STEREO;
          const double c2 = reproj_chi2(k, qn, c.ob, stereo, c.s, e, iz);
          D.chir[c.l] = c2;
          double r0 = c2, r1;
          if (robust) {
            const double dl = stereo ? k.delta_stereo : k.delta_mono;
            huber_bf(c2, dl, dl * dl, r0, r1);
          }
          acc[1] += r0;
        }
        if (c.ag) acc[1] += gmm_chi2_fast(k, gm, c.nd, c.fl, c.asc, pn);
      }
      PROF_T(tB1);
      reduce2<2>(acc, red, tot, C);
      PROF_T(tB2);
      // computeScale: sum_l eps.(lambda eps + b_l) + dx.(lambda dx + b_p).  With eps = u - D^-1 A gd the
      // b-terms collapse to  sum u.b + dx.g  (g = reduced rhs of pass A), so pass B needs no b at all.
      double scale = lambda * acc[0] + uni(bc[25]);
      const double tempChi = ok2 ? acc[1] : 1.7976931348623157e308;
      if (pose_active) {
#pragma unroll
        for (int i = 0; i < 6; ++i) scale += dx[i] * (lambda * dx[i] + uni(bc[19 + i]));
      }
      scale += 1e-3;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        const double uu = 2 * rho - 1;
        double alpha = 1. - uu * uu * uu;
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        P = Pn;
      } else {
        lambda *= ni;
        ni *= 2;
        PROF_ADD(7, 0, 1);  // rejected trials
#pragma unroll 1
        for (int i = 0; i < PPTF; ++i) {  // discardTop: restore the backed-up points
          const int l = threadIdx.x + i * TF;
          if (l >= L) break;
          const int fl = fw_get(fw, i);
          if (!(fl & F_EXISTS)) continue;
          const bool ar = !(fl & F_LEVR), ag = (fl & F_ASSOC) && !(fl & F_LEVG);
          if (!(ar || ag)) continue;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            D.sp[j * MCAP + l] = __hiloint2double(D.un[(2 * j + 1) * MCAP + l], D.un[(2 * j) * MCAP + l]);
        }
      }
      qmax++;
      ++trials;
      PROF_T(tE);
      PROF_ADD(0, tA0, tA1); PROF_ADD(1, tA1, tA2); PROF_ADD(2, tA2, tS); PROF_ADD(3, tS, tB1); PROF_ADD(4, tB1, tB2); PROF_ADD(5, tB2, tE); PROF_ADD(6, tA0, tA0 + 1);
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (qmax == 10 || rho == 0) break;
  }
  return cj;
}

__global__ __launch_bounds__(TF, 2) void k_ba1_fast(BaK k, GmmDev gm, int B, int L, double* __restrict__ pose_io,
                                                 double* __restrict__ pts_io, const double* __restrict__ obs_all,
                                                 const int32_t* __restrict__ oct_all, int32_t* __restrict__ assoc_all,
                                                 const double* __restrict__ d2_all, uint8_t* __restrict__ dropped_all,
                                                 uint8_t* __restrict__ erase_all, int32_t* __restrict__ iters_out,
                                                 double* __restrict__ pn_all, int32_t* __restrict__ trials_out
#ifdef GL_BAF_COOPERATIVE
                                                 ,
                                                 int NB, unsigned long long* parts
#endif
) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  Lds D;
  D.sp = smem;                          // 3 * MCAP
  D.chir = D.sp + 3 * MCAP;             // MCAP
  D.un = (int*)(D.chir + MCAP);         // 12 * MCAP words = 6 * MCAP doubles
  double* red = D.chir + 7 * MCAP;      // NWF * 32
  double* tot = red + NWF * 32;         // 32 (+ 32 broadcast slots)
  D.stab = tot + 64;                    // 8
  FlagW fw = 0;
  const int tid = threadIdx.x;
#ifdef GL_BAF_COOPERATIVE
  // workgroup pb of the NB that share frame f owns the points [l0, l0 + L) of the frame's Lf
  const int f = blockIdx.x / NB;
  if (f >= B) return;
  Coop C{parts + (size_t)f * 2 * NB * 64, NB, (int)(blockIdx.x % NB), 0u};
  const int Lf = L, Lp = (Lf + NB - 1) / NB;
  const int l0 = min(C.pb * Lp, Lf);
  L = min(Lp, Lf - l0);
  const size_t gbase = (size_t)f * Lf + l0;
#else
  const int f = blockIdx.x;
  if (f >= B) return;
  Coop C{nullptr, 1, 0, 0u};
  const size_t gbase = (size_t)f * L;
#endif
  double* gnd = pn_all + gbase * 4;  // per-point plane record {n, n.mu} (written once, then read-only)
  const double* gobs = obs_all + gbase * 3;
  int32_t* gassoc = assoc_all + gbase;
#pragma unroll 1
  for (int i = 0; i < PPTF; ++i) {
    const int l = tid + i * TF;
    if (l >= L) break;
    const size_t g = gbase + l;
    const int oc = oct_all[g];
    int a = assoc_all[g];
    // association gate chi2 <= 9 (checkMapAssociation, gmmloc_opt.cpp:230-232)
    if (d2_all && k.gate_chi2 >= 0 && !(d2_all[g] <= k.gate_chi2)) a = -1;
    if (oc < 0) a = -1;
    int fl = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) D.sp[j * MCAP + l] = pts_io[g * 3 + j];
    if (oc >= 0) {
      fl = F_EXISTS | ((oc & 7) << 8);
      if (!(gobs[(size_t)l * 3 + 2] < 0)) fl |= F_STEREO;
      if (a >= 0) {
        fl |= F_ASSOC;
        if (gm.flags[a] & 1) {
          fl |= F_DEG;
          const double nx = gm.axis[(size_t)a * 9], ny = gm.axis[(size_t)a * 9 + 3], nz = gm.axis[(size_t)a * 9 + 6];
          gnd[(size_t)l * 4] = nx;
          gnd[(size_t)l * 4 + 1] = ny;
          gnd[(size_t)l * 4 + 2] = nz;
          gnd[(size_t)l * 4 + 3] = nx * gm.rec12[(size_t)a * 12] + ny * gm.rec12[(size_t)a * 12 + 1] + nz * gm.rec12[(size_t)a * 12 + 2];
        }
      }
    }
    gassoc[l] = a;
    D.chir[l] = 0.0;
    fw_or(fw, i, fl);
  }
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) D.stab[j] = k.s2inv[j];  // static indices: a lane-indexed kernarg array would go through scratch
  }
  Pose P = pose_from_se3(se3_load(pose_io + (size_t)f * 7));
  __syncthreads();

  // schedule (:770-828): optimize(5) -> gate degenerate GMM edges -> optimize(5) -> gate reprojection
  // edges, robust kernels off -> optimize(40).  One rolled phase loop = one copy of the optimiser code.
  int it3 = 0, trials = 0;
#pragma unroll 1
  for (int phase = 0; phase < 3; ++phase) {
    it3 = optimize_fast(k, gm, D, fw, P, L, gobs, gassoc, gnd, phase < 2, phase < 2 ? 5 : 40, red, tot, trials, C);
    if (phase == 2) break;
#pragma unroll 1
    for (int i = 0; i < PPTF; ++i) {
      const int l = tid + i * TF;
      if (l >= L) break;
      const int fl = fw_get(fw, i);
      if (phase == 0) {  // fresh error of the degenerate GMM edges (:773-786)
        if ((fl & (F_ASSOC | F_DEG)) == (F_ASSOC | F_DEG)) {
          const double p[3] = {D.sp[l], D.sp[MCAP + l], D.sp[2 * MCAP + l]};
          const double nd[4] = {gnd[(size_t)l * 4], gnd[(size_t)l * 4 + 1], gnd[(size_t)l * 4 + 2], gnd[(size_t)l * 4 + 3]};
          if (gmm_chi2_fast(k, gm, nd, fl, -1, p) > k.str_thresh) fw_or(fw, i, F_LEVG);
        }
      } else {  // STALE chi2 of the reprojection edges, fresh depth test (:799-825)
        if (!(fl & F_EXISTS)) continue;
        const double z = P.R[6] * D.sp[l] + P.R[7] * D.sp[MCAP + l] + P.R[8] * D.sp[2 * MCAP + l] + P.t[2];
        if (D.chir[l] > ((fl & F_STEREO) ? 7.815 : 5.991) || !(z > 0.0)) fw_or(fw, i, F_LEVR);
      }
    }
    __syncthreads();
  }

#pragma unroll 1
  for (int i = 0; i < PPTF; ++i) {  // outputs (:837-879, :898-922)
    const int l = tid + i * TF;
    if (l >= L) break;
    const size_t g = gbase + l;
    const int fl = fw_get(fw, i);
    uint8_t dr = 0, er = 0;
    int a = gassoc[l];
    if (fl & F_EXISTS) {
      const double p[3] = {D.sp[l], D.sp[MCAP + l], D.sp[2 * MCAP + l]};
      if ((fl & (F_ASSOC | F_DEG)) == (F_ASSOC | F_DEG)) {
        const double nd[4] = {gnd[(size_t)l * 4], gnd[(size_t)l * 4 + 1], gnd[(size_t)l * 4 + 2], gnd[(size_t)l * 4 + 3]};
        if (gmm_chi2_fast(k, gm, nd, fl, -1, p) > k.str_thresh) dr = 1;
      }
      const double z = P.R[6] * p[0] + P.R[7] * p[1] + P.R[8] * p[2] + P.t[2];
      if (D.chir[l] > ((fl & F_STEREO) ? 7.815 : 5.991) || !(z > 0.0)) er = 1;
#pragma unroll
      for (int j = 0; j < 3; ++j) pts_io[g * 3 + j] = p[j];
    }
    if (dropped_all) dropped_all[g] = dr;
    if (erase_all) erase_all[g] = er;
    if (!dropped_all && dr) a = -1;
    gassoc[l] = a;
  }
  if (tid == 0 && C.pb == 0) {
    SE3 T;
    T.r = qfromR(P.R);
    T.t[0] = P.t[0];
    T.t[1] = P.t[1];
    T.t[2] = P.t[2];
    normalize_rotation(T);
    se3_store(T, pose_io + (size_t)f * 7);
    if (iters_out) iters_out[f] = it3;
    if (trials_out) trials_out[f] = trials;
#ifdef GL_BA_PROF
    if (f == 0)
      for (int i = 0; i < 7; ++i) pose_io[i] = (double)g_prof[i == 5 ? 7 : i];  // slot 5 reports the rejections  // debug build only: phase cycles instead of pose 0
#endif
  }
}

}  // namespace GL_BAF_NS
}  // namespace
