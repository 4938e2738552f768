// (included by gl_ba_fast.hip once per instance: GL_BAF_NS / GL_BAF_MCAP / GL_BAF_NW / GL_BAF_SPREAD / GL_BAF_STEP32 / GL_BAF_PRIOR / GL_BAF_FIXED)
// On-chip fast path of the single-pose structure-constrained refinement (same algorithm and control flow as
// k_ba1 in gl_ba.hip, which stays as the general / large-M path and as the A/B reference), M <= 2000 points.
//
// ONE summation order for every launch shape.  The result of a frame must not depend on how many frames ride
// in the call, so the order in which the per-point terms of every sum are added is a function of the frame's
// stride L alone (the "canonical order"):
//     chunks   c = l / 64 (64 consecutive points), nch = ceil(L / 64)
//     groups   G = ceil(nch / 4) groups; group g = the chunks g, g + G, g + 2 G, ... (S = ceil(nch / G) <= 4 of them):
//              neighbouring chunks sit in different groups, so a run of expensive points (see k_ba1_prep) spreads
//              over the waves instead of landing on one
//     level 1  lane j of group g adds the terms of its points (g + G i) 64 + j, i = 0 .. S-1, in that order
//     level 2  the 64 lane sums of a group meet in the butterfly tree of gld::wave_reduce_scatter32
//              (partners 32, 16, 1, 2, 4, 8 lanes apart)
//     level 3  blocks of two groups, B_k = g_2k + g_2k+1, then ((B_0 + B_1) + B_2) + B_3.
// Two kernels realise it, with the SAME per-point code (pt_lin / pt_pass_a / pt_pass_b below, every term is
// rounded on its own before it enters a sum: add_nc):
//   * DENSE  (batches): one workgroup of G waves per frame, wave g = group g, a thread owns the S points of its
//     lane and keeps level 1 in registers; 1 / 2 / 4 frames per CU by LDS class (MCAP 2000 / 1000 / 496);
//   * SPREAD (few frames, the frame-at-a-time caller): one point per thread, a workgroup of 256 threads = the
//     <= 4 slot waves of ONE group (one wave per SIMD), G co-resident workgroups per frame (cooperative launch
//     when > 1); level 1 runs through an LDS transpose (term [value][slot wave][lane]): the S waves of the group
//     share its values, add the S terms of each lane in slot order and finish with the same butterfly
//     (wave_reduce_scatter8); level 3 crosses the workgroups as tagged words (gld::Coop) in block order.
// tests/test_gpu_track.py::test_track_frames_bit_identical_across_shapes holds the two to equal bits.
//
// Per-frame state in LDS, SoA over the points, for the whole 5/5/40 schedule: current point (3 fp64), stale
// chi2 (1 fp64) and a 48-byte slot with two lifetimes: between the two passes of a Levenberg trial it holds
// the L Delta L^T factors of the damped point block D (6 fp64) of pass A, so that pass B computes the point step
//     eps = D^-1 (b - A (omega x q + upsilon)) = D^-1 (Jpi^T W (e - Jpi gd) + b_gmm)
// exactly in fp64 from the re-evaluated residual (no fp32 anywhere; GL_BAF_STEP32 instead caches the fp32
// pair {u = D^-1 b, A D^-1} of round 1: ~8 % faster, but its 1e-7 step error is amplified by badly conditioned
// frames - off by default, gl_ctx_set_option("ba_step32", 1)); after pass B the slot holds the backup of the
// point while the trial point sits in place (nothing to copy on acceptance).  80 B/point.
// computeScale is  lambda (sum|eps|^2 + |dx|^2) + sum u.b + dx.g ; the pose is (R, t) in SGPRs; rcp / rsq +
// two Newton steps; branch-free Huber; the 6x6 LDL^T + exp() run on wave 0 and are broadcast through LDS.
namespace {
namespace GL_BAF_NS {

constexpr int MCAP = GL_BAF_MCAP;  // LDS capacity in points
#if GL_BAF_SPREAD
constexpr bool kSpread = true;
#else
constexpr bool kSpread = false;
#endif
#if GL_BAF_STEP32
constexpr bool kStep32 = true;
#else
constexpr bool kStep32 = false;
#endif
#if GL_BAF_PRIOR
constexpr bool kPrior = true;   // instance with the gauge anchor of the pose (prior edge / fixed pose); the plain instances
#else                           // carry none of its code, so their register allocation is that of the unanchored refine
constexpr bool kPrior = false;
#endif
#if GL_BAF_FIXED
constexpr bool kFixed = true;   // instance with fixed observer key-frames (gl_track_frames_anchored, F = 1 .. 4): further reprojection
#else                           // edges of the frame's points with FIXED pose vertices (localization_opt.cpp:491-516, 706-760)
constexpr bool kFixed = false;
#endif
#ifndef GL_BAF_DYNB
#define GL_BAF_DYNB 0  // measured (profiles/r5_ab_dynb.txt): same bits, the waves end pass B together - and the refine takes 10 % LONGER
#endif
#ifndef GL_BAF_DYNB_PRIO
#define GL_BAF_DYNB_PRIO 1
#endif
// DENSE, exact step: the chunks of pass B are dealt DYNAMICALLY (optimize_fast); not with the fp32 cache, whose words fill the slot
constexpr bool kDynB = !kSpread && !kStep32 && GL_BAF_DYNB != 0;
#ifndef GL_BAF_ASYM
#define GL_BAF_ASYM 0
#endif
constexpr bool kAsym = GL_BAF_ASYM != 0 && GL_BAF_NW == 8;  // uneven deal of the chunks over the two waves of a SIMD (slot_off)
#ifndef GL_BAF_ALLSOLVE
#define GL_BAF_ALLSOLVE 0  // measured (profiles/r5_ab_allsolve.txt): same bits, refine 10.80 -> 11.09 ms per 4 096 frames - seven more waves issuing the ~350 dependent instructions cost more than the hand-over they save
#endif
constexpr bool kAllSolve = !kSpread && GL_BAF_ALLSOLVE != 0;  // DENSE: every wave solves the reduced system itself (optimize_fast)
constexpr int NWC = GL_BAF_NW;             // DENSE: GROUPS of the canonical order a frame of this LDS class has at most (= its waves, but see GPW)
#ifndef GL_BAF_GPW
#define GL_BAF_GPW 1
#endif
// GPW = 2 (the `x` instance of the largest class): a wave owns TWO groups of the canonical order - g and g + 4, four slots each, a
// reduce-scatter after each group's slots, so every sum is added exactly as by eight waves - and the 48-byte hand-over slot of a
// point (factors of pass A / backup of pass B) lives in GLOBAL memory, written and read back by the thread that owns the point.
// The frame then needs 4 waves and 32 B of LDS per point: TWO frames per CU like the 1 000-point class, one wave of each per SIMD
// (no older / younger wave of the same frame, one frame's solve behind the other's passes).
constexpr int GPW = GL_BAF_GPW;
constexpr bool kTwo = GPW == 2;
constexpr int NWV = NWC / GPW;             // waves of a frame at most
static_assert(GPW == 1 || (GPW == 2 && NWC == 8 && !kSpread && !kStep32), "two groups per wave: the plain batch instance of the 2 000-point class");
constexpr int NRED = kSpread ? 1 : NWC;    // group totals kept in LDS (a SPREAD workgroup is ONE group)
constexpr int TSP = 256;                   // SPREAD: threads of a workgroup = the <= 4 slot waves of its group

#ifdef GL_BA_TRACE  // debug build: (currentChi, tempChi, lambda, rho) of every Levenberg trial of frame 0 -> its points
__device__ double g_trace[10 * 128];
#define TRACE_TRIAL(t, a, b, c_, d, dxp)                            \
  if (blockIdx.x == 0 && threadIdx.x == 0 && (t) < 128) {            \
    g_trace[(t)*10] = (a);                                           \
    g_trace[(t)*10 + 1] = (b);                                       \
    g_trace[(t)*10 + 2] = (c_);                                      \
    g_trace[(t)*10 + 3] = (d);                                       \
    for (int i_ = 0; i_ < 6; ++i_) g_trace[(t)*10 + 4 + i_] = (dxp)[i_]; \
  }
#else
#define TRACE_TRIAL(t, a, b, c_, d, dxp)
#endif
#ifdef GL_BA_PROF
__device__ unsigned long long g_prof[16];
__device__ unsigned long long g_prof_k[16];  // a frame in the middle of the launch (block 1024): clock64 at the stations of the kernel; [12] lambda-init cycles
#define PROF_K(i) if (blockIdx.x == 1024 && threadIdx.x == 0) g_prof_k[i] = (unsigned long long)clock64()
#define PROF_KADD(i, t0, t1) if (blockIdx.x == 1024 && threadIdx.x == 0) g_prof_k[i] += (unsigned long long)((t1) - (t0))
__device__ unsigned long long g_prof_w[64 * 8 * 4];  // [trial < 64][wave][marker]: clock64 at pass-A end, after reduce A, pass-B start, pass-B end
#define PROF_W(trial, marker) \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (trial) < 64) g_prof_w[((trial)*8 + (threadIdx.x >> 6)) * 4 + (marker)] = (unsigned long long)clock64()
// [trial < 16][wave][pass A / B][slot 0 .. 3 start, pass end]: where inside a pass a wave's time goes (tools/prof_ba.py)
__device__ unsigned long long g_prof_s[16 * 8 * 2 * 5];
// [trial < 16][wave][pass][slot][0: the slot's global loads have arrived, 1: step half done (pass B), 2: slot done]
__device__ unsigned long long g_prof_q[10 * 8 * 2 * 4 * 3];
#define PROF_Q(trial, pass, slot, m) \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (trial) >= 6 && (trial) < 16 && (slot) < 4 && (pass) < 2) \
  g_prof_q[(((((trial) - 6) * 8 + (int)(threadIdx.x >> 6)) * 2 + (pass)) * 4 + (slot)) * 3 + (m)] = (unsigned long long)clock64()
#define PROF_S(trial, pass, slot) \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (trial) < 16 && (slot) < 5 && (pass) < 2) g_prof_s[(((trial)*8 + (threadIdx.x >> 6)) * 2 + (pass)) * 5 + (slot)] = (unsigned long long)clock64()
#else
#define PROF_W(trial, marker)
#define PROF_S(trial, pass, slot)
#define PROF_Q(trial, pass, slot, m)
#endif
#ifndef PROF_T
#ifdef GL_BA_PROF
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) if (blockIdx.x == 0 && threadIdx.x == 0) g_prof[slot] += (unsigned long long)((t1) - (t0))
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#define PROF_K(i)
#define PROF_KADD(i, t0, t1)
#endif
#endif

// F_AR / F_AG: the point's reprojection / GMM edge is ACTIVE (exists, level 0) - derived bits, refreshed whenever the
// levels change (fw_activity), so that a pass tests one mask instead of re-deriving them per point and trial
// (the values of the first four and the octave at bits 8..10 are what k_ba1_prep writes: gl_ba_fast.hip)
// (fixed-observer instances: bit 11 + k = the point's edge to fixed key-frame k is INACTIVE - not observed, or at level 1 -, F_AF =
// some fixed edge of the point is active; bits 8..10 hold the octave)
enum { F_EXISTS = 1, F_STEREO = 2, F_ASSOC = 4, F_DEG = 8, F_LEVR = 16, F_LEVG = 32, F_AR = 64, F_AG = 128, F_OFFF = 1 << 11, F_AF = 1 << 15 };
constexpr int kMaxFixed = 4;  // fixed observer key-frames the on-chip path takes (more: the general kernel, gl_ba_gen.hip)

// a + b that is never contracted with a multiplication feeding it: every term of a canonical sum is rounded on
// its own, whichever kernel adds it (this file is compiled with contraction allowed)
GL_DEV double add_nc(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

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

#ifndef GL_U
#define GL_U(i, j) ((i) * 6 - (i) * ((i)-1) / 2 + ((j) - (i)))  // packed upper triangle of a 6x6, i <= j
#endif

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
  // dR = I + a [w]x + b [w]x^2,  V = I + b [w]x + c [w]x^2  with  [w]x^2 = w w^T - |w|^2 I  written out (the generic
  // 3x3 products would multiply by the zeros of the skew matrix: ~50 instructions on the serial path of every trial)
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
  Pose N;
  mm3(dR, P.R, N.R);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    N.t[i] = dR[i * 3] * P.t[0] + dR[i * 3 + 1] * P.t[1] + dR[i * 3 + 2] * P.t[2] + V[i * 3] * u[3] + V[i * 3 + 1] * u[4] +
             V[i * 3 + 2] * u[5];
  return N;
}

struct Lds {      // per-frame state, SoA over MCAP points (index = local point index)
  double* sp;     // 3 x MCAP  current point (world)
  double* chir;   // MCAP      stale chi2 of the reprojection edge (e->chi2())
  // 6 x MCAP doubles, two lifetimes sharing one slot per point:
  //   pass A -> pass B : the factors of D (ldl3_factor_fast, 6 fp64)   [GL_BAF_STEP32: 12 fp32 words {u = D^-1 b (3), A D^-1 (3x3)}]
  //   pass B -> accept : backup of the point (3 fp64) while the trial point sits in `sp`; restored only
  //                      when the trial is rejected.
  double* un;
  // 24 uniform constants: [0..7] sx = fx^2 / sigma^2 and [8..15] sy = fy^2 / sigma^2 per pyramid octave (the residuals
  // are kept in NORMALISED image coordinates, see reproj_n), [16..19] Huber {delta, delta^2} mono / stereo
  double* stab;
  // fixed observer key-frames (kFixed): their poses {R (9), t (3)} x F in LDS; per point and key-frame, in the launch's scratch
  // (SoA by key-frame, the frame's permuted point order): normalised observation, octave | stereo << 4 (-1: none), and the
  // stale chi2 of the edge (its Huber weight rho' between the two passes of a trial, like `chir`)
  double* frt;
  const double* gfobn;
  const int32_t* gfoct;
  double* gchif;
  int F, Lf;
};
// per-point flag bits + octave (bits 8..10) live in REGISTERS: 16 bits per point slot of the thread
template <bool TWO>
struct FwHi {
  typedef unsigned type;
};
template <>
struct FwHi<true> {
  typedef unsigned long long type;
};
struct FlagW {  // slots 0..3 in `lo`, slot 4 (the fifth chunk of a big group, GL_BAF_ASYM) in `hi`; GPW = 2: the second group's slots 4..7 in `hi`
  unsigned long long lo;
  FwHi<kTwo>::type hi;
};
GL_DEV int fw_get(const FlagW& fw, int i) {
  if (kTwo) return (int)(((i < 4 ? fw.lo : (unsigned long long)fw.hi) >> (16 * (i & 3))) & 0xffffull);
  return i < 4 ? (int)((fw.lo >> (16 * i)) & 0xffffull) : (int)(fw.hi & 0xffffu);
}
GL_DEV void fw_or(FlagW& fw, int i, int bits) {
  if (i < 4) fw.lo |= (unsigned long long)bits << (16 * i);
  else if (kTwo) fw.hi |= (FwHi<kTwo>::type)((unsigned long long)bits << (16 * (i & 3)));
  else fw.hi |= (unsigned)bits;
}
GL_DEV void fw_activity(FlagW& fw, int i) {
  const int fl = fw_get(fw, i);
  int act = 0;
  if ((fl & F_EXISTS) && !(fl & F_LEVR)) act |= F_AR;
  if ((fl & F_EXISTS) && (fl & F_ASSOC) && !(fl & F_LEVG)) act |= F_AG;
  if (kFixed && (fl & F_EXISTS) && ((fl / F_OFFF) & 15) != 15) act |= F_AF;
  if (i < 4) fw.lo = (fw.lo & ~((unsigned long long)(F_AR | F_AG | F_AF) << (16 * i))) | ((unsigned long long)act << (16 * i));
  else if (kTwo) fw.hi = (FwHi<kTwo>::type)(((unsigned long long)fw.hi & ~((unsigned long long)(F_AR | F_AG | F_AF) << (16 * (i & 3)))) | ((unsigned long long)act << (16 * (i & 3))));
  else fw.hi = (fw.hi & ~(unsigned)(F_AR | F_AG | F_AF)) | (unsigned)act;
}

// the hand-over slot of a point (Lds::un).  GPW = 2: global memory, streamed (-DGL_BAF_UN_NT: non-temporal accesses)
GL_DEV void un_st(double* p, double v) {
#ifdef GL_BAF_UN_NT
  if (kTwo) {
    __builtin_nontemporal_store(v, p);
    return;
  }
#endif
  *p = v;
}
GL_DEV double un_ld(const double* p) {
#ifdef GL_BAF_UN_NT
  if (kTwo) return __builtin_nontemporal_load(p);
#endif
  return *p;
}
// the canonical order of a frame of stride L and this thread's place in it
struct Map {
  int S;      // point slots of this thread (DENSE: chunks of its group; SPREAD: 1)
  int base;   // frame-local index of the thread's first point; slot i is base + step i
  int lbase;  // LDS index of the first point (DENSE: = base; SPREAD: threadIdx.x)
  int step;   // 64 G: the chunks of a group are G apart
  int L;      // points of the frame
  bool asym;  // GL_BAF_ASYM, frames of 8 groups: the groups 0..3 take five chunks, the groups 4..7 three (slot_off)
  bool g2;    // GPW = 2: the wave's second group (wave + 4) exists in this frame (G > wave + 4): its slots 4..7 are points of the frame
};
// Offset of slot i from the thread's first point.  Frames of 8 groups (29 .. 32 chunks: the 2 000-point class), GL_BAF_ASYM: the two
// waves of a SIMD do not run at the same speed - the arbiter serves the older one first, the younger one fills the gaps, and once
// the older one is through its chunks the younger runs ALONE at half the issue rate (its dependent chains have nobody to hide
// behind): a third of every pass.  The chunks are therefore dealt unevenly: rounds 0..2 one chunk to each of the 8 groups (chunk
// 8 r + g), rounds 3 and 4 only to the groups 0..3 (chunk 24 + 4 (r - 3) + g): the older wave of every SIMD has five chunks,
// the younger three, and they end their passes together (profiles/r5_prof_ba_slots.txt).  Still a function of L alone.
GL_DEV int slot_off(const Map& mp, int i) {
  if (kTwo) return mp.step * (i & 3) + 64 * NWV * (i >> 2);  // slots 4..7: the chunks of group wave + 4
  return !mp.asym ? mp.step * i : (i < 3 ? 512 * i : 1536 + 256 * (i - 3));
}
// (GPW = 2) slot i belongs to a group this frame does not have: its chunk index would alias another group's
GL_DEV bool slot_absent(const Map& mp, int i) { return kTwo && i >= 4 && !mp.g2; }

struct Lin {
  double q[3];
  double A[6];   // reprojection block sum_r w_r j_r^T j_r (camera frame, sym6; A[1] == 0)
  double a[3];   // its rhs
  double D[6];   // A + GMM block (no damping yet)
  double b[3];   // a + GMM rhs
  double rho0_r, rho1, chi_g;
};

// uniform camera / weight constants of a launch (SGPRs)
struct Uni {
  double bn;    // bf / fx: baseline in normalised image units
  double lm;    // ba_lambda2
};

// Reprojection residual in NORMALISED image coordinates: the observations are stored once per launch as
// ((u - cx) / fx, (v - cy) / fy, (u_right - cx) / fx), so e_n = obs_n - (x/z, y/z, x/z - bn/z) needs no intrinsics and
// chi2 = sx (e0^2 + e2^2) + sy e1^2 with sx = fx^2 / sigma^2, sy = fy^2 / sigma^2 (per octave, LDS table) - the same
// quadratic form as s |obs - K pi(q)|^2 of EdgeStereoSE3ProjectXYZ / factors.cpp:66-168.
GL_DEV double reproj_n(const double* q, const double* ob, bool stereo, double bn, double sx, double sy, double* e, double& iz) {
  iz = rcp_nr(q[2]);
  e[0] = fma(-q[0], iz, ob[0]);
  e[1] = fma(-q[1], iz, ob[1]);
  e[2] = stereo ? fma(bn - q[0], iz, ob[2]) : 0.0;  // u_right - (x - b) / z
  return fma(sy * e[1], e[1], sx * fma(e[2], e[2], e[0] * e[0]));
}
// rows of the projection Jacobian in normalised coordinates: (iz, 0, c0), (0, iz, c1) and, stereo only, (iz, 0, c2)
struct Jpi {
  double c0, c1, c2;
};
GL_DEV Jpi jpi_n(const double* q, double iz, double bn) {
  Jpi J;
  const double iz2 = iz * iz;
  J.c0 = -q[0] * iz2;
  J.c1 = -q[1] * iz2;
  J.c2 = fma(bn, iz2, J.c0);
  return J;
}

// Hc = R Hg R^T (sym6): a point block from the world into the camera frame
GL_DEV void rot_sym(const double* R, const double* Hg, double* Hc) {
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
  if (bc) {
#pragma unroll
    for (int i = 0; i < 3; ++i) bc[i] = -(R[i * 3] * Hd[0] + R[i * 3 + 1] * Hd[1] + R[i * 3 + 2] * Hd[2]);
  }
  if (Hc) rot_sym(R, Hg, Hc);
  return chi;
}

// per-point context loaded at the top of a loop iteration
struct PtCtx {
  int l, ll, fl, asc;  // frame-local point index, LDS index, flags, non-degenerate component (or -1)
  double sx, sy;       // fx^2 / sigma^2, fy^2 / sigma^2 of the point's octave
  double ob[3], nd[4], p[3];
  bool ar, ag, af;
};

// Huber weight of the point's reprojection edge from its un-robustified chi2 (delta by edge type, LDS table)
GL_DEV void huber_pt(const Lds& D, bool stereo, double chi, double& rho0, double& rho1) {
  const double dl = D.stab[stereo ? 18 : 16], dsqr = D.stab[stereo ? 19 : 17];
  huber_bf(chi, dl, dsqr, rho0, rho1);
}

// linearise the point of context c at pose P; returns the un-robustified chi2_r
GL_DEV double lin_fast(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& P, const PtCtx& c, bool robust, Lin& o) {
  const double* p = c.p;
#pragma unroll
  for (int k = 0; k < 3; ++k) o.q[k] = fma(P.R[k * 3], p[0], fma(P.R[k * 3 + 1], p[1], fma(P.R[k * 3 + 2], p[2], P.t[k])));
#pragma unroll
  for (int k = 0; k < 6; ++k) o.A[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) o.a[k] = 0.0;
  o.rho0_r = 0.0;
  o.rho1 = 1.0;
  o.chi_g = 0.0;
  double chi_r = 0.0;
  if (c.ar) {
    const bool stereo = c.fl & F_STEREO;
    double e[3], iz;
    chi_r = reproj_n(o.q, c.ob, stereo, U.bn, c.sx, c.sy, e, iz);
    o.rho0_r = chi_r;
    if (robust) huber_pt(D, stereo, chi_r, o.rho0_r, o.rho1);
    // weights of the u / v rows and (t) of the stereo row
    const double wx = o.rho1 * c.sx, wy = o.rho1 * c.sy;
    const double t = stereo ? wx : 0.0;
    const Jpi J = jpi_n(o.q, iz, U.bn);
    const double iz2 = iz * iz;
    const double wxc0 = wx * J.c0, wyc1 = wy * J.c1, tc2 = t * J.c2, wyiz = wy * iz;
    o.A[0] = (wx + t) * iz2;
    o.A[2] = iz * (wxc0 + tc2);
    o.A[3] = wy * iz2;
    o.A[4] = wyiz * J.c1;
    o.A[5] = fma(tc2, J.c2, fma(wyc1, J.c1, wxc0 * J.c0));
    o.a[0] = iz * fma(t, e[2], wx * e[0]);
    o.a[1] = wyiz * e[1];
    o.a[2] = fma(tc2, e[2], fma(wyc1, e[1], wxc0 * e[0]));
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) o.D[k] = o.A[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) o.b[k] = o.a[k];
  if (c.ag) {
    if (c.fl & F_DEG) {  // EdgePt2GaussianDeg x ba_lambda2: D += lm n' n'^T, b -= lm eg n'  (n' = R n)
      const double nx = c.nd[0], ny = c.nd[1], nz = c.nd[2];
      const double eg = fma(nz, p[2], fma(ny, p[1], nx * p[0])) - c.nd[3];
      const double lm = U.lm;
      o.chi_g = eg * (lm * eg);
      double nc[3], tn[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        nc[k] = fma(P.R[k * 3 + 2], nz, fma(P.R[k * 3 + 1], ny, P.R[k * 3] * nx));
        tn[k] = lm * nc[k];
      }
      o.D[0] = fma(tn[0], nc[0], o.D[0]);
      o.D[1] = fma(tn[0], nc[1], o.D[1]);
      o.D[2] = fma(tn[0], nc[2], o.D[2]);
      o.D[3] = fma(tn[1], nc[1], o.D[3]);
      o.D[4] = fma(tn[1], nc[2], o.D[4]);
      o.D[5] = fma(tn[2], nc[2], o.D[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) o.b[k] = fma(-eg, tn[k], o.b[k]);
    } else {
      double Hc[6], bc[3];
      o.chi_g = gmm_nondeg(gm, c.asc, P.R, p, Hc, bc);
#pragma unroll
      for (int k = 0; k < 6; ++k) o.D[k] += Hc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) o.b[k] += bc[k];
    }
  }
  return chi_r;
}

GL_DEV double gmm_chi2_fast(double lm, const GmmDev& gm, const double* nd, int fl, int asc, const double* p) {
  if (fl & F_DEG) {
    const double eg = (nd[0] * p[0] + nd[1] * p[1] + nd[2] * p[2]) - nd[3];
    return eg * (lm * eg);
  }
  return gmm_nondeg(gm, asc, nullptr, p, nullptr, nullptr);
}

// ---- fixed observer key-frames (kFixed instances) --------------------------------------------------------------------
// The reference's structure BA holds the local map points with the OTHER key-frames that observe them, as fixed pose vertices
// (localization_opt.cpp:491-516; EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ with Huber, :706-760).  Such an edge touches the
// point's block and right-hand side only - never the reduced pose system directly.  It is evaluated in its key-frame's camera
// (q_f = R_f p + t_f, the normalised-coordinate residual of reproj_n), its Jacobian rows are taken to the WORLD frame
// (g = j R_f: R_f is constant for the whole kernel, in LDS), summed there over the point's key-frames and rotated into the
// current camera frame once per point: D += R (sum w g g^T) R^T, b += R (sum w e g).  tools/emul_fixed.py replays this on the
// host against the oracle's joint_optimization(P = 1, F fixed).
struct FxE {
  double e[3], chi, sx, sy;
  double g[9];  // Jacobian rows in the world frame
  bool stereo;
};
// edge (point l, key-frame k) at world point p; rows: also the Jacobian
GL_DEV bool fixed_eval(const Uni& U, const Lds& D, int k, int l, const double* p, bool rows, FxE& E) {
  const int oc = D.gfoct[(size_t)k * D.Lf + l];
  const double* Rf = D.frt + k * 12;
  double q[3], ob[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) ob[j] = D.gfobn[((size_t)k * D.Lf + l) * 3 + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) q[j] = fma(Rf[j * 3], p[0], fma(Rf[j * 3 + 1], p[1], fma(Rf[j * 3 + 2], p[2], Rf[9 + j])));
  E.stereo = (oc & 16) != 0;
  E.sx = D.stab[oc & 7];
  E.sy = D.stab[8 + (oc & 7)];
  double iz;
  E.chi = reproj_n(q, ob, E.stereo, U.bn, E.sx, E.sy, E.e, iz);
  if (rows) {
    const Jpi J = jpi_n(q, iz, U.bn);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      E.g[j] = fma(iz, Rf[j], J.c0 * Rf[6 + j]);
      E.g[3 + j] = fma(iz, Rf[3 + j], J.c1 * Rf[6 + j]);
      E.g[6 + j] = fma(iz, Rf[j], J.c2 * Rf[6 + j]);
    }
  }
  return true;
}
// sum over the point's active fixed edges of w e g (-> bw) and, with Hw, of w g g^T (world frame); rho' of every edge comes
// from (keep_rho < 0) / goes to (> 0) its stale-chi2 cell; returns the robustified chi2
GL_DEV double fixed_sum(const Uni& U, const Lds& D, const PtCtx& c, bool robust, double* Hw, double* bw, int keep_rho) {
  double sum = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) bw[j] = 0.0;
  if (Hw) {
#pragma unroll
    for (int j = 0; j < 6; ++j) Hw[j] = 0.0;
  }
#pragma unroll 1
  for (int k = 0; k < D.F; ++k) {
    if (c.fl & (F_OFFF << k)) continue;
    FxE E;
    fixed_eval(U, D, k, c.l, c.p, true, E);
    double rho0 = E.chi, rho1 = 1.0;
    double* cell = D.gchif + (size_t)k * D.Lf + c.l;
    if (keep_rho < 0) {
      rho1 = *cell;
    } else {
      if (robust) huber_pt(D, E.stereo, E.chi, rho0, rho1);
      if (keep_rho > 0) *cell = rho1;
    }
    sum += rho0;
    const double w[3] = {rho1 * E.sx, rho1 * E.sy, E.stereo ? rho1 * E.sx : 0.0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double* g = E.g + r * 3;
      const double wg[3] = {w[r] * g[0], w[r] * g[1], w[r] * g[2]};
      if (Hw) {
        Hw[0] = fma(wg[0], g[0], Hw[0]);
        Hw[1] = fma(wg[0], g[1], Hw[1]);
        Hw[2] = fma(wg[0], g[2], Hw[2]);
        Hw[3] = fma(wg[1], g[1], Hw[3]);
        Hw[4] = fma(wg[1], g[2], Hw[4]);
        Hw[5] = fma(wg[2], g[2], Hw[5]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) bw[j] = fma(E.e[r], wg[j], bw[j]);
    }
  }
  return sum;
}
// pass A / lambda init: the fixed edges' part of the point's block and right-hand side, in the camera frame (Hc sym6, bc);
// evaluated BEFORE the frame's own edge is linearised, so that only these nine values are live across lin_fast
GL_DEV double fixed_lin(const Uni& U, const Lds& D, const Pose& P, const PtCtx& c, bool robust, double* Hc, double* bc, bool keep_rho) {
  double Hw[6], bw[3];
  const double sum = fixed_sum(U, D, c, robust, Hw, bw, keep_rho ? 1 : 0);
  rot_sym(P.R, Hw, Hc);
#pragma unroll
  for (int j = 0; j < 3; ++j) bc[j] = fma(P.R[j * 3], bw[0], fma(P.R[j * 3 + 1], bw[1], P.R[j * 3 + 2] * bw[2]));
  return sum;
}
GL_DEV void fixed_add(const double* Hc, const double* bc, Lin& o) {
#pragma unroll
  for (int j = 0; j < 6; ++j) o.D[j] += Hc[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) o.b[j] += bc[j];
}
// pass B, evaluation half: robustified chi2 of the fixed edges at the trial point; the stale chi2 cells are rewritten
GL_DEV double fixed_chi(const Uni& U, const Lds& D, const PtCtx& c, const double* pn, bool robust) {
  double sum = 0.0;
#pragma unroll 1
  for (int k = 0; k < D.F; ++k) {
    if (c.fl & (F_OFFF << k)) continue;
    FxE E;
    fixed_eval(U, D, k, c.l, pn, false, E);
    D.gchif[(size_t)k * D.Lf + c.l] = E.chi;
    double r0 = E.chi, r1;
    if (robust) huber_pt(D, E.stereo, E.chi, r0, r1);
    sum += r0;
  }
  return sum;
}

// The damped point block D = A + (GMM block) + lambda I is factorised, D = L Delta L^T (unpivoted: D is symmetric positive
// definite), and every product with D^-1 - u = D^-1 b, the rows of A D^-1, the point step of pass B - is a pair of triangular
// solves with the factors f = {l10, l20, l21, 1/d0, 1/d1, 1/d2}.  Until round 4 the block was inverted by cofactors, like Eigen's
// fixed-size inverse() that g2o calls on its landmark blocks: fine for a well-conditioned block, but a point that has run away
// along its plane (soak frame map_v1 r63072: an outlier 5 km from the camera, eigenvalues of D = 400 / 7e-3 / lambda) has a
// determinant that is the rounding residue of three products of size 1e3 as soon as lambda < 1e-6, its "inverse" is garbage, and
// through G = [-[q]x | I] with |q| = 5e3 that garbage entered the reduced pose system amplified by |q|^2: pose steps of 1e-4 ..
// 1e-3 where the oracle's are 1e-6 - 3.6e-5 rad off an oracle that no perturbation moves by more than 4e-8.  (The oracle inverts
// by cofactors too; its subtraction form A - A D^-1 A turns the same garbage into a relative error of the point's small
// contribution.)  The factorisation is backward stable whatever the scaling of the block: the frame ends 3e-8 from the oracle,
// for the same instruction count (tools/emul_ba1.py replays the arithmetic on the host).
// The first two pivots' reciprocals do not depend on each other (1/d1 = d0 / (d0 d1), the leading 2 x 2 minor by one fma, which
// is as accurate as the recurrence d1 = D11 - l10 D10): two reciprocals in sequence instead of three.
GL_DEV void ldl3_factor_fast(const double* D, double* f) {
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
}
// back half of a solve: (L Delta L^T)^-1 b from y = L^-1 b
GL_DEV void ldl3_back(const double* f, double y0, double y1, double y2, double* x) {
  x[2] = y2 * f[5];
  x[1] = fma(-f[2], x[2], y1 * f[4]);
  x[0] = fma(-f[1], x[2], fma(-f[0], x[1], y0 * f[3]));
}
GL_DEV void ldl3_solve_fast(const double* f, const double* b, double* x) {
  const double y1 = fma(-f[0], b[0], b[1]);
  ldl3_back(f, b[0], y1, fma(-f[2], y1, fma(-f[1], b[0], b[2])), x);
}

GL_DEV void point_solve_fast(const Lin& o, double lambda, double* f, double* u) {
  const double D[6] = {o.D[0] + lambda, o.D[1], o.D[2], o.D[3] + lambda, o.D[4], o.D[5] + lambda};
  ldl3_factor_fast(D, f);
  ldl3_solve_fast(f, o.b, u);
}

// ---- gauge anchor of the frame's pose ----------------------------------------------------------------------------
// The reference never runs its structure BA without one (localization_opt.cpp:491-516, 556-581).  Per frame: the
// EdgeSE3QuatPrior of key-frame 0 (factors.cpp:19-53; measurement = the INPUT pose, sigma_rot 2 deg, sigma_t 1 cm) when
// loc::ba_first_as_prior, else the pose vertex is fixed (:578-580).  The inverse measurement {R (9), t (3)} is written by
// k_ba1_prep into the launch's scratch and fetched by the solving wave with scalar loads when it is needed (the LDS of the
// 1 000-point class has no 96 bytes to spare at two frames per CU, and 24 more live registers would spill in the passes).
// The edge's linearisation costs ~500 dependent double-precision instructions (log, the 6 x 6 Jacobian J = Jr Adj, J^T Omega J):
// on the solving wave, between the two passes of every trial, it made the anchored refine 60 % slower than the plain one.
// It is therefore evaluated at the TRIAL pose during pass B, by one designated wave (the last one; an idle slot wave on the
// latency shape) next to its points, into the second of two LDS records {H (21, packed upper), b (6), chi2}: the chi2 feeds
// this trial's accept / reject test, and an accepted trial makes the record the current one, which the next solve only has
// to add - a rejected trial keeps the old record.  The record of the input pose is made once, at the start of the kernel.
typedef const double __attribute__((address_space(4))) cdouble_k;
struct Anchor {
  bool has_prior, pose_fixed;
  const cdouble_k* mi;
  double* rec;  // LDS: 2 x 32 doubles
  int cur;      // which record belongs to the current pose
  double* work; // LDS work area of prior_record_wave (108 doubles, free during pass B and at the start of the kernel)
  bool rezero;  //   ... borrowed from the group totals: zero it again after use
  int wave;     // the wave that makes the records: wave 1 of a batch workgroup (on 8-wave frames the waves 0 - 3 finish pass B
                // ~3 k cycles before 4 - 7, which the issue priority favours less: the edge's ~400 instructions fit that slack);
                // the last slot wave - idle unless the group has all four chunks - on the latency shape
};
// _error = (_inverseMeasurement * Tj).log()   (SE3Quat::log: small-angle branch above |d| = 0.99999)
GL_DEV void prior_error(const cdouble_k* mi_k, const Pose& P, double* e) {
  double mi[12], dR[9], dt[3];
#pragma unroll
  for (int i = 0; i < 12; ++i) mi[i] = mi_k[i];
  mm3(mi, P.R, dR);
#pragma unroll
  for (int i = 0; i < 3; ++i) dt[i] = fma(mi[i * 3], P.t[0], fma(mi[i * 3 + 1], P.t[1], fma(mi[i * 3 + 2], P.t[2], mi[9 + i])));
  const double d = 0.5 * (dR[0] + dR[4] + dR[8] - 1);
  const double v[3] = {dR[7] - dR[5], dR[2] - dR[6], dR[3] - dR[1]};
  double w[3], g;
  if (fabs(d) > 0.99999) {
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = 0.5 * v[i];
    g = 1. / 12.;
  } else {
    const double theta = acos(d), sn = sqrt(1 - d * d);
    const double f = theta / (2 * sn);
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = f * v[i];
    g = (1 - theta * (1 + d) / (2 * sn)) / (theta * theta);  // tan(theta / 2) = sin / (1 + cos), theta in (0, pi)
  }
  // upsilon = V^-1 dt,  V^-1 = I - 1/2 [w]x + g [w]x^2  =  dt - 1/2 w x dt + g w x (w x dt)
  double c1[3], c2[3];
  cross(w, dt, c1);
  cross(w, c1, c2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    e[i] = w[i];
    e[3 + i] = fma(g, c2[i], fma(-0.5, c1[i], dt[i]));
  }
}
GL_DEV double prior_chi2(const double* e) {
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0)), st = 1.0 / (0.01 * 0.01);
  return sr * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + st * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
}
// linearizeOplus (factors.cpp:30-53): J = (I + 1/2 [[phi]x [upsilon]x; 0 [phi]x]) Adj(T^-1); H (packed upper, 21) += J^T Omega J,
// b -= J^T Omega e; returns chi2
GL_DEV double prior_lin(const cdouble_k* mi, const Pose& P, double* H, double* b) {
  double e[6];
  prior_error(mi, P, e);
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0)), st = 1.0 / (0.01 * 0.01);
  // T^-1 = (R^T, -R^T t);  Adj = [[Ri, 0], [[ti]x Ri, Ri]]
  double Ri[9], ti[3], S[9], SR[9], A1[9], Lh[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = P.R[j * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) ti[i] = -(Ri[i * 3] * P.t[0] + Ri[i * 3 + 1] * P.t[1] + Ri[i * 3 + 2] * P.t[2]);
  skew(ti, S);
  mm3(S, Ri, SR);
  skew(e, A1);
  skew(e + 3, Lh);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    A1[i] = 0.5 * A1[i] + ((i % 4 == 0) ? 1.0 : 0.0);
    Lh[i] = 0.5 * Lh[i];
  }
  double AR[9], ASR[9], LSR[9], LR[9];
  mm3(A1, Ri, AR);
  mm3(A1, SR, ASR);
  mm3(Lh, SR, LSR);
  mm3(Lh, Ri, LR);
  double J[36];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[i * 6 + j] = AR[i * 3 + j] + LSR[i * 3 + j];
      J[i * 6 + 3 + j] = LR[i * 3 + j];
      J[(3 + i) * 6 + j] = ASR[i * 3 + j];
      J[(3 + i) * 6 + 3 + j] = AR[i * 3 + j];
    }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double om_e[6] = {sr * e[0], sr * e[1], sr * e[2], st * e[3], st * e[4], st * e[5]};
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s = fma(J[r * 6 + i], om_e[r], s);
    if (b) b[i] -= s;
#pragma unroll
    for (int j = i; j < 6; ++j) {
      double h = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) h = fma(J[r * 6 + i] * (r < 3 ? sr : st), J[r * 6 + j], h);
      H[GL_U(i, j)] += h;
    }
  }
  return prior_chi2(e);
}

// The edge's record at pose P -> LDS {H (21, packed upper), b (6), chi2}, made by ONE wave and read behind a barrier.
// The 6 x 6 products run lane-parallel out of an LDS work area (108 doubles: Jr, Adj, J = Jr Adj; lane (i, j) owns one
// element): evaluated in registers (prior_lin: J[36] and ~120 live values) the edge took the register allocation of the
// PASS LOOPS from 36 to 181 spilled VGPRs - inlined - or forced everything that lives across a call into the callee-saved
// registers - as a function - and the whole anchored kernel ran at 0.4 - 0.6 of the plain one (profiles/history/r3_prior_cost.txt).
GL_DEV void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
GL_DEV void prior_record_wave(const cdouble_k* mi, const Pose& P, double* rec, double* work, bool rezero) {
  const int lane = threadIdx.x & 63;
  double e[6];
  prior_error(mi, P, e);
  // staging (static indices, one lane): Ri = R^T -> work[72..80], ti = -R^T t -> work[81..83], e -> work[84..89]
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int c = 0; c < 3; ++c) work[72 + a * 3 + c] = P.R[c * 3 + a];
      work[81 + a] = -(P.R[a] * P.t[0] + P.R[3 + a] * P.t[1] + P.R[6 + a] * P.t[2]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) work[84 + a] = e[a];
  }
  wave_lds_order();
  {  // lane (i, j) builds its element of  Jr = I + 1/2 [[phi]x [upsilon]x; 0 [phi]x]  (-> work[0..35])  and of
     // Adj(T^-1) = [[Ri, 0], [[ti]x Ri, Ri]]  (-> work[36..71]);  [v]x(a, b) = +-v[3 - a - b] off the diagonal
    const int l0 = min(lane, 35), bi = l0 / 6, bj = l0 - 6 * bi, a3 = bi % 3, b3 = bj % 3;
    const bool top = bi < 3, left = bj < 3, offd = a3 != b3;
    const int kx = offd ? 3 - a3 - b3 : 0;
    const double sgn = ((b3 - a3 + 3) % 3 == 1) ? -0.5 : 0.5;
    const double sk_phi = offd ? sgn * work[84 + kx] : 0.0, sk_ups = offd ? sgn * work[87 + kx] : 0.0;
    const double jr = (top == left) ? sk_phi + (offd ? 0.0 : 1.0) : (top ? sk_ups : 0.0);
    // ([ti]x Ri)(a, c) = ti[a+1] Ri[a+2][c] - ti[a+2] Ri[a+1][c]  (indices mod 3)
    const int a1 = (a3 + 1) % 3, a2 = (a3 + 2) % 3;
    const double sr_ac = work[81 + a1] * work[72 + a2 * 3 + b3] - work[81 + a2] * work[72 + a1 * 3 + b3];
    const double ri_ab = work[72 + a3 * 3 + b3];
    const double adj = (top == left) ? ri_ab : (top ? 0.0 : sr_ac);
    wave_lds_order();
    if (lane < 36) {
      work[l0] = jr;
      work[36 + l0] = adj;
    }
  }
  wave_lds_order();
  const int lc = min(lane, 35), i = lc / 6, j = lc - 6 * i;
  {  // J = Jr Adj
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 6; ++l) s = fma(work[i * 6 + l], work[36 + l * 6 + j], s);
    wave_lds_order();
    if (lane < 36) work[72 + lc] = s;
  }
  wave_lds_order();
  const double sr = 1.0 / ((2.0 * M_PI / 180.0) * (2.0 * M_PI / 180.0)), st = 1.0 / (0.01 * 0.01);
  {  // H = J^T Omega J (upper triangle, packed), b = -J^T Omega e
    double h = 0.0, g = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double w = r < 3 ? sr : st, jri = work[72 + r * 6 + i], jrj = work[72 + r * 6 + j];
      h = fma(jri * w, jrj, h);
      g = fma(jrj, w * e[r], g);  // (lanes i == 0: column j)
    }
    if (lane < 36 && i <= j) rec[i * 6 - i * (i - 1) / 2 + (j - i)] = h;
    if (lane < 6) rec[21 + lane] = -g;
  }
  if (lane == 0) rec[27] = prior_chi2(e);
  if (rezero) {  // the work area is borrowed from the group totals, whose rows of absent groups must read 0.0
    wave_lds_order();
    work[lane] = 0.0;
    if (lane < 44) work[64 + lane] = 0.0;
  }
}

// The terms of a point enter the sums through a sink: DENSE adds them to the thread's registers (level 1 of the
// canonical order), SPREAD just keeps them (they go to the LDS transpose).  Each index is written once per point.
#ifndef GL_BAF_FEWACC
#define GL_BAF_FEWACC 0
#endif
// (GL_BAF_FEWACC: TIMING EXPERIMENT ONLY, results are garbage - the 27 Schur terms of a point are folded into four accumulators, so that
// the passes keep their arithmetic but not the 29 live sums: what a kernel with 50 fewer registers would cost per pass, profiles/r6_w3_fewacc.txt)
struct SinkAcc {
  double* a;
  GL_DEV void put(int i, double v) const {
    if (GL_BAF_FEWACC && i < 27) a[i & 3] = add_nc(a[i & 3], v);
    else a[i] = add_nc(a[i], v);
  }
};
struct SinkSet {
  double* a;
  GL_DEV void put(int i, double v) const { a[i] = v; }
};

// pass B with the chunks dealt dynamically: a point's two terms wait in the free half of its slot (doubles 3 and 4: the factors have
// been read, the backup takes 0..2) for the thread that OWNS the point in the canonical order
struct SinkSlot {
  double* un;
  int ll;
  GL_DEV void put(int i, double v) const { un[(3 + i) * MCAP + ll] = v; }
};

// terms 0..20 <- upper(G^T C G), 21..26 <- G^T c,  G = [-[q]x | I], C symmetric (sym6)
template <class Sink>
GL_DEV void pose_terms(const double* q, const double* C, const double* c, bool with_rhs, const Sink& sk) {
  // M = [q]x C (row r, column j) = (q x C[:, j])[r];  TL = [q]x C [q]x^T
  const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
  double M[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[j] = fma(q[1], Cf[6 + j], -q[2] * Cf[3 + j]);
    M[3 + j] = fma(q[2], Cf[j], -q[0] * Cf[6 + j]);
    M[6 + j] = fma(q[0], Cf[3 + j], -q[1] * Cf[j]);
  }
  sk.put(0, fma(q[1], M[2], -q[2] * M[1]));
  sk.put(1, fma(q[2], M[0], -q[0] * M[2]));
  sk.put(2, fma(q[0], M[1], -q[1] * M[0]));
  sk.put(3, M[0]);
  sk.put(4, M[1]);
  sk.put(5, M[2]);
  sk.put(6, fma(q[2], M[3], -q[0] * M[5]));
  sk.put(7, fma(q[0], M[4], -q[1] * M[3]));
  sk.put(8, M[3]);
  sk.put(9, M[4]);
  sk.put(10, M[5]);
  sk.put(11, fma(q[0], M[7], -q[1] * M[6]));
  sk.put(12, M[6]);
  sk.put(13, M[7]);
  sk.put(14, M[8]);
  sk.put(15, C[0]);
  sk.put(16, C[1]);
  sk.put(17, C[2]);
  sk.put(18, C[3]);
  sk.put(19, C[4]);
  sk.put(20, C[5]);
  if (with_rhs) {
    sk.put(21, fma(q[1], c[2], -q[2] * c[1]));
    sk.put(22, fma(q[2], c[0], -q[0] * c[2]));
    sk.put(23, fma(q[0], c[1], -q[1] * c[0]));
    sk.put(24, c[0]);
    sk.put(25, c[1]);
    sk.put(26, c[2]);
  }
}

// AD = A D^-1 (3x3): row r = D^-1 (row r of A), three solves with the factors.  The reprojection block
// A = sum_r w_r j_r^T j_r = {A0, 0, A2, A3, A4, A5} (sym6) has A(0,1) = 0 by construction (no row of Jpi touches x and y): written
// out, the forward substitutions skip the multiplications by that zero (IEEE arithmetic may not drop them itself).
GL_DEV void ad_solve(const double* A, const double* f, double* AD) {
  {
    const double y1 = -f[0] * A[0];
    ldl3_back(f, A[0], y1, fma(-f[2], y1, fma(-f[1], A[0], A[2])), AD);
  }
  {  // y0 = 0
    const double y2 = fma(-f[2], A[3], A[4]);
    AD[5] = y2 * f[5];
    AD[4] = fma(-f[2], AD[5], A[3] * f[4]);
    AD[3] = fma(-f[1], AD[5], -f[0] * AD[4]);
  }
  {
    const double y1 = fma(-f[0], A[2], A[4]);
    ldl3_back(f, A[2], y1, fma(-f[2], y1, fma(-f[1], A[2], A[5])), AD + 6);
  }
}

// ---- reductions in the canonical order ---------------------------------------------------------------------
// (wave_reduce_scatter8 - the 8-value variant of gld::wave_reduce_scatter32 with the same lane pairings in the same order - lives in
// gl_device.hpp: the pose kernel's split shape uses it too)
struct Red {
  double* red;   // NRED x 32: per group (DENSE: wave) totals
  double* tot;   // 32 totals (+ 32 broadcast slots)
  double* red2;  // DENSE: NWC x 2 wave totals of pass B (zero where a wave is absent)
  double* tb;    // SPREAD: transpose buffer [value][TSP]
  int S;         // chunks per group
};

// totals of NV values over the frame -> tot[0..NV-1] (LDS), valid after the call for every thread.
// DENSE: v[] holds the lane sums (level 1 done in registers).  SPREAD: v[] holds this thread's terms.
template <int NV>
GL_DEV void reduce_to_tot(double* v, const Red& R, Coop& C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (!kSpread) {
#pragma unroll
    for (int i = NV; i < 32; ++i) v[i] = 0.0;
    const double r = wave_reduce_scatter32(v);
    // no barrier needed before writing `red`: its last readers (threads < 32) finished before the
    // closing barrier of the previous reduction, which every thread has passed
    if (wave_slot_owner(lane)) {
      R.red[wave * 32 + wave_slot(lane)] = r;
      // (GPW = 2: both groups of the wave in one row - only exact sums go this way: counts, XCC ids -, the second group's row reads 0.0)
      if (kTwo) R.red[(wave + NWV) * 32 + wave_slot(lane)] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 32) {  // absent groups hold zeros (never written after the initial clear); a class
      // with fewer groups than 8 just has no further (all-zero) blocks to add
      double s = NWC > 1 ? add_nc(R.red[threadIdx.x], R.red[32 + threadIdx.x]) : R.red[threadIdx.x];
#pragma unroll
      for (int b = 1; b < NWC / 2; ++b) s = add_nc(s, add_nc(R.red[(2 * b) * 32 + threadIdx.x], R.red[(2 * b + 1) * 32 + threadIdx.x]));
      R.tot[threadIdx.x] = s;
    }
    __syncthreads();
  } else {
    // level 1 through LDS: term [value][slot wave][lane]
#pragma unroll
    for (int i = 0; i < NV; ++i) R.tb[i * TSP + threadIdx.x] = v[i];
    __syncthreads();
    const int S = R.S, slot = wave;
    if (slot < S) {
      // the S waves of the group share its values in rounds of 8: wave `slot` takes rounds slot, slot + S, ...
      for (int r8 = slot; r8 * 8 < NV; r8 += S) {
        double y[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int val = r8 * 8 + kk;
          // all four slot values are fetched before the first add (absent slots add 0.0, which changes nothing: the sum
          // starts from +0.0): with a loop over the S slots every LDS read waited for the previous add - 32 serialised
          // LDS round trips, 3.9 k of a trial's 20 k cycles
          double x[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] = (val < NV && j < S) ? R.tb[val * TSP + j * 64 + lane] : 0.0;
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) s = add_nc(s, x[j]);
          y[kk] = s;
        }
        const double t8 = wave_reduce_scatter8(y);
        if ((lane & 0xE) == 0) R.red[r8 * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + (lane & 1)] = t8;
      }
    }
    __syncthreads();
    // levels 2 -> 3: this group's totals meet the other groups' (workgroups) in the canonical block order; a frame of one
    // group adds the absent partner's 0.0 like the one-workgroup kernel does
    if (threadIdx.x < 32) R.tot[threadIdx.x] = threadIdx.x < NV ? (C.NB > 1 ? R.red[threadIdx.x] : add_nc(R.red[threadIdx.x], 0.0)) : 0.0;
    __syncthreads();
    if (C.NB > 1) coop_totals<false>(C, R.tot);
  }
}
template <int NV>
GL_DEV void reduce2(double* v, const Red& R, Coop& C) {
  reduce_to_tot<NV>(v, R, C);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = uni(R.tot[i]);
}
// same, but only wave 0 (the one that solves the reduced system) reads the totals back
template <int NV>
GL_DEV void reduce2_w0(double* v, const Red& R, Coop& C) {
  reduce_to_tot<NV>(v, R, C);
  if (threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = uni(R.tot[i]);
  }
}
// ---- SPREAD shortcuts for the two reductions of every Levenberg trial (G > 1: the frame's groups sit in different
// workgroups).  Same canonical order; the group totals go from the wave that made them STRAIGHT into the tagged words of
// the exchange (gld::Coop) and wave 0 combines everybody's words - the tags order producer and consumer, so the path
// `red -> barrier -> tot -> barrier -> publish` (three of a reduction's four barriers) is gone.
// level 1 (LDS transpose, slot order) + level 2 (butterfly) of this group, published by the owning lanes
template <int NV>
GL_DEV void spread_publish(const double* v, const Red& R, const Coop& C, unsigned seq) {
  const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6, S = R.S;
#pragma unroll
  for (int i = 0; i < NV; ++i) R.tb[i * TSP + threadIdx.x] = v[i];
  __syncthreads();
  if (slot < S) {
    unsigned long long* buf = C.part + (size_t)(seq & 1u) * C.NB * 64 + (size_t)C.pb * 64;
    for (int r8 = slot; r8 * 8 < NV; r8 += S) {
      double y[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int val = r8 * 8 + kk;
        double x[4];  // all four slot values before the first add (see reduce_to_tot)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = (val < NV && j < S) ? R.tb[val * TSP + j * 64 + lane] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = add_nc(s, x[j]);
        y[kk] = s;
      }
      const double t8 = wave_reduce_scatter8(y);
      if ((lane & 0xE) == 0 && !C.failed) {
        const int vi = r8 * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + (lane & 1);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t8);
        if (C.same_xcd) {  // plain stores: the words stay in the XCD's L2, where the siblings' L1-bypassing loads find them
          __hip_atomic_store(buf + vi * 2, (bits << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(buf + vi * 2 + 1, (bits & 0xffffffff00000000ull) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {           // device scope: written through to memory, visible from every XCD
          __hip_atomic_store(buf + vi * 2, (bits << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(buf + vi * 2 + 1, (bits & 0xffffffff00000000ull) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}
// lane t (< 8 ceil(NV / 8)) of the calling wave: value t of every group -> the frame's total (blocks of two groups, in order)
GL_DEV double spread_collect(const Coop& C, unsigned seq, int t) {
  if (C.failed) return 0.0;
  const unsigned long long* buf = C.part + (size_t)(seq & 1u) * C.NB * 64;
  constexpr int NBMAX = 8;
  unsigned long long w0[NBMAX], w1[NBMAX];
  bool all, off = false;
  int spins = 0;
  long long t0 = 0;
  do {
    all = true;
#pragma unroll
    for (int p = 0; p < NBMAX; ++p) {
      if (p < C.NB) {
        const unsigned long long* w = buf + ((size_t)p * 32 + t) * 2;
        w0[p] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1[p] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the abort word travels with the same batch of requests (asked for afterwards it would add a round trip to every
    // poll that has to be repeated: one frame 0.440 -> 0.430 ms).  (Two polls in flight half a round trip apart were
    // measured too: 0.48 ms - a repeated poll is cheap, the wait is for the slowest sibling's pass, not for the fabric.)
    // Every exchange has the time limit of the rendezvous (gld::coop_give_up: the clock is read only after POLL_FREE
    // unsuccessful polls): a sibling that stops answering sends the frame to the follow-up kernel instead of hanging it.
    const int ab = __hip_atomic_load(C.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int p = 0; p < NBMAX; ++p)
      if (p < C.NB) all = all && (unsigned)w0[p] == seq && (unsigned)w1[p] == seq;
    if (!all) off = coop_give_up(C, ab, spins, t0);
  } while (!all && !off);
  if (off || coop_test_abort(C, seq)) *C.lds_fail = 1;
  double g[NBMAX];
#pragma unroll
  for (int p = 0; p < NBMAX; ++p) g[p] = p < C.NB ? __longlong_as_double((long long)((w1[p] & 0xffffffff00000000ull) | (w0[p] >> 32))) : 0.0;
  double s = add_nc(g[0], g[1]);
#pragma unroll
  for (int b = 1; b < NBMAX / 2; ++b)
    if (2 * b < C.NB) s = add_nc(s, add_nc(g[2 * b], g[2 * b + 1]));
  return s;
}
// pass A: 29 values, read by the solving wave only
GL_DEV void spread_reduce29_w0(double* v, const Red& R, Coop& C) {
  if (C.NB == 1) {
    reduce2_w0<29>(v, R, C);
    return;
  }
  const unsigned seq = ++C.seq;
  spread_publish<29>(v, R, C, seq);
  if (threadIdx.x < 64) {
    double s = 0.0;
    if (threadIdx.x < 32) s = spread_collect(C, seq, threadIdx.x);
    C.failed |= __builtin_amdgcn_readfirstlane(*(volatile int*)C.lds_fail);  // the other waves learn it behind pass B's barrier
    union {
      double d;
      int i[2];
    } u, w;
    u.d = s;
#pragma unroll
    for (int i = 0; i < 29; ++i) {
      w.i[0] = __builtin_amdgcn_readlane(u.i[0], i);
      w.i[1] = __builtin_amdgcn_readlane(u.i[1], i);
      v[i] = w.d;
    }
  }
}
// pass B: 2 values, needed by every thread
GL_DEV void spread_reduce2_all(double* v, const Red& R, Coop& C) {
  if (C.NB == 1) {
    reduce2<2>(v, R, C);
    return;
  }
  const unsigned seq = ++C.seq;
  spread_publish<2>(v, R, C, seq);
  if (threadIdx.x < 8) R.tot[threadIdx.x] = spread_collect(C, seq, threadIdx.x);
  __syncthreads();
  C.failed |= *C.lds_fail;
  v[0] = uni(R.tot[0]);
  v[1] = uni(R.tot[1]);
}

// ---- DENSE shortcuts for the two reductions of every Levenberg trial (same canonical order, fewer barriers) ----
// pass A (29 values, read by the solving wave only): group totals to red[], ONE barrier, then wave 0 adds the blocks
// itself and hands the totals round its lanes with v_readlane - no `tot` round trip, no second barrier.  The next
// writer of red[] is the next trial's pass A, two barriers later.
// (all_waves: every wave adds the blocks and takes the totals - GL_BAF_ALLSOLVE, the serial section without a hand-over)
// (GPW = 2) the totals of the group whose slots the wave has just finished -> row `row` of red[]; the accumulators start again at zero
GL_DEV void group_flush29(double* v, const Red& R, int row) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 29; i < 32; ++i) v[i] = 0.0;
  const double r = wave_reduce_scatter32(v);
  if (wave_slot_owner(lane)) R.red[row * 32 + wave_slot(lane)] = r;
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0.0;
}
GL_DEV void reduce29_w0_dense(double* v, const Red& R, bool all_waves = false) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = kTwo ? wave + NWV : wave;  // (GPW = 2: v[] holds the wave's SECOND group; the first went to its row at slot 4)
#pragma unroll
  for (int i = 29; i < 32; ++i) v[i] = 0.0;
  const double r = wave_reduce_scatter32(v);
  if (wave_slot_owner(lane)) R.red[row * 32 + wave_slot(lane)] = r;
  __syncthreads();
  if (wave == 0 || all_waves) {
    const int t = lane & 31;
    double s = NWC > 1 ? add_nc(R.red[t], R.red[32 + t]) : R.red[t];
#pragma unroll
    for (int b = 1; b < NWC / 2; ++b) s = add_nc(s, add_nc(R.red[(2 * b) * 32 + t], R.red[(2 * b + 1) * 32 + t]));
    union {
      double d;
      int i[2];
    } u, w;
    u.d = s;
#pragma unroll
    for (int i = 0; i < 29; ++i) {
      w.i[0] = __builtin_amdgcn_readlane(u.i[0], i);
      w.i[1] = __builtin_amdgcn_readlane(u.i[1], i);
      v[i] = w.d;
    }
  }
}
// one value summed over the wave in the canonical butterfly (partners 32, 16, 1, 2, 4, 8 lanes apart, every lane ends
// with the total): with a = b = x the lane swaps return own and partner value, whichever half the lane is in
GL_DEV double wave_allreduce_canon(double x) {
  union {
    double d;
    unsigned u[2];
  } a, b;
  a.d = x;
  gl_v2u lo = __builtin_amdgcn_permlane32_swap(a.u[0], a.u[0], false, false);
  gl_v2u hi = __builtin_amdgcn_permlane32_swap(a.u[1], a.u[1], false, false);
  a.u[0] = lo.x;
  b.u[0] = lo.y;
  a.u[1] = hi.x;
  b.u[1] = hi.y;
  x = a.d + b.d;
  a.d = x;
  lo = __builtin_amdgcn_permlane16_swap(a.u[0], a.u[0], false, false);
  hi = __builtin_amdgcn_permlane16_swap(a.u[1], a.u[1], false, false);
  a.u[0] = lo.x;
  b.u[0] = lo.y;
  a.u[1] = hi.x;
  b.u[1] = hi.y;
  x = a.d + b.d;
  x = x + dpp_f64<0xB1>(x);   // lane ^ 1
  x = x + dpp_f64<0x4E>(x);   // lane ^ 2
  x = x + dpp_f64<0x141>(x);  // lane ^ 7
  x = x + dpp_f64<0x140>(x);  // lane ^ 15
  return x;
}
// pass B (2 values, needed by every thread): wave totals to the small buffer red2, ONE barrier, every thread
// adds the blocks itself.  The buffer is rewritten one trial later, with the barriers of pass A in between.
GL_DEV void group_flush2(double* v, const Red& R, int row) {  // (GPW = 2: like group_flush29, for pass B's two sums)
  const double s0 = wave_allreduce_canon(v[0]), s1 = wave_allreduce_canon(v[1]);
  if ((threadIdx.x & 63) == 0) {
    R.red2[row * 2] = s0;
    R.red2[row * 2 + 1] = s1;
  }
  v[0] = v[1] = 0.0;
}
GL_DEV void reduce2_dense(double* v, const Red& R) {
  const int lane = threadIdx.x & 63, wave = kTwo ? (int)(threadIdx.x >> 6) + NWV : (int)(threadIdx.x >> 6);
  const double s0 = wave_allreduce_canon(v[0]), s1 = wave_allreduce_canon(v[1]);
  if (lane == 0) {
    R.red2[wave * 2] = s0;
    R.red2[wave * 2 + 1] = s1;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double s = NWC > 1 ? add_nc(R.red2[j], R.red2[2 + j]) : R.red2[j];
#pragma unroll
    for (int b = 1; b < NWC / 2; ++b) s = add_nc(s, add_nc(R.red2[(2 * b) * 2 + j], R.red2[(2 * b + 1) * 2 + j]));
    v[j] = uni(s);
  }
}

GL_DEV double reduce_max(double v, const Red& R, Coop& C) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmax(v, shfl_xor_f64(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) R.red[threadIdx.x >> 6] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  double m = R.red[0];
  for (int w = 1; w < nw; ++w) m = fmax(m, R.red[w]);
  if (kSpread && C.NB > 1) {
    __syncthreads();  // tot may still be read from the previous reduction
    if (threadIdx.x < 32) R.tot[threadIdx.x] = m;
    __syncthreads();
    coop_totals<true>(C, R.tot);
    m = uni(R.tot[0]);
  }
  __syncthreads();  // red[0..7] are rewritten by the next sum (group 0 owns them)
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

#ifndef GL_BAF_SOLVE_BLOCKED
#define GL_BAF_SOLVE_BLOCKED 1  // (0: the sequential LDL^T of rounds 1 - 5; profiles/r6_solve_blocked.txt)
#endif
// The same 6 x 6 system by BLOCKS (round 6): H + lambda I = [[A, B], [B^T, C]] with the rotation block A and the translation block C,
//     A = L_a D_a L_a^T,   Y = A^-1 [B | b_r],   S = C - B^T Y_B,   s = b_t - B^T y,   S = L_s D_s L_s^T,   x_t = S^-1 s,   x_r = y - Y_B x_t.
// The pivots are those of the unpivoted LDL^T of the whole matrix (D_a, then D_s): the same failure test.  What changes is the SHAPE of
// the work: the serial section of a trial - one wave, every other wave of the frame waiting - was a chain of ~190 dependent fp64
// instructions (six reciprocals one after the other, each behind the eliminations of the column before); here four right-hand sides
// go through the factors of A side by side and the two 3 x 3 factorisations have their first two reciprocals independent
// (ldl3_factor_fast): about 50 instructions deep for the same count.  Other rounding, same system: held to the oracle like everything else.
GL_DEV bool ldl3_factor_chk(const double* D, double* f) {  // ldl3_factor_fast + the pivot test of SimplicialLDLT (zero / non-finite pivot)
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
  return D[0] != 0.0 && isfinite(D[0]) && d1 != 0.0 && isfinite(d1) && d2 != 0.0 && isfinite(d2);
}
GL_DEV bool ldlt6_blocked(const double* a, const double* b, double lambda, double* x) {
  const double A[6] = {a[GL_U(0, 0)] + lambda, a[GL_U(0, 1)], a[GL_U(0, 2)], a[GL_U(1, 1)] + lambda, a[GL_U(1, 2)], a[GL_U(2, 2)] + lambda};
  double fa[6];
  const bool ok_a = ldl3_factor_chk(A, fa);
  double Y[3][3], y[3];  // Y[j] = A^-1 (column j of B)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double col[3] = {a[GL_U(0, 3 + j)], a[GL_U(1, 3 + j)], a[GL_U(2, 3 + j)]};
    ldl3_solve_fast(fa, col, Y[j]);
  }
  ldl3_solve_fast(fa, b, y);
  double S[6], s[3];
  {
    const int ri[6] = {0, 0, 0, 1, 1, 2}, ci[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int i = ri[e], j = ci[e];
      const double c = a[GL_U(3 + i, 3 + j)] + (i == j ? lambda : 0.0);
      S[e] = fma(-a[GL_U(2, 3 + i)], Y[j][2], fma(-a[GL_U(1, 3 + i)], Y[j][1], fma(-a[GL_U(0, 3 + i)], Y[j][0], c)));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = fma(-a[GL_U(2, 3 + i)], y[2], fma(-a[GL_U(1, 3 + i)], y[1], fma(-a[GL_U(0, 3 + i)], y[0], b[3 + i])));
  }
  double fs[6];
  const bool ok_s = ldl3_factor_chk(S, fs);
  ldl3_solve_fast(fs, s, x + 3);
#pragma unroll
  for (int k = 0; k < 3; ++k) x[k] = fma(-Y[2][k], x[5], fma(-Y[1][k], x[4], fma(-Y[0][k], x[3], y[k])));
  return ok_a && ok_s;
}

// backup of the current point in the slot (pass B) / restore on a rejected trial.  The fp32 cache of
// GL_BAF_STEP32 is read by its owner just before, but planes of OTHER points alias a double-indexed slot,
// so that variant keeps the {lo, hi} words in the owner's own 32-bit entries.
GL_DEV void backup_point(const Lds& D, int ll, const double* p) {
  if (kStep32) {
    int* un = (int*)D.un;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      un[(2 * j) * MCAP + ll] = __double2loint(p[j]);
      un[(2 * j + 1) * MCAP + ll] = __double2hiint(p[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) un_st(D.un + (unsigned)(j * MCAP + ll), p[j]);
  }
}
GL_DEV void restore_point(const Lds& D, int ll) {
  if (kStep32) {
    const int* un = (const int*)D.un;
#pragma unroll
    for (int j = 0; j < 3; ++j) D.sp[j * MCAP + ll] = __hiloint2double(un[(2 * j + 1) * MCAP + ll], un[(2 * j) * MCAP + ll]);
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) D.sp[j * MCAP + ll] = un_ld(D.un + (unsigned)(j * MCAP + ll));
  }
}

// The normalised observations and the plane records come from the launch's scratch in global memory (read-only after
// the set-up, coalesced, L2-resident).  A point takes part in a pass iff one of its edges is active: one mask test
// (slots beyond the frame never get a flag).
// (Software-prefetching slot i+1 was measured: it costs 14 VGPRs -> 6 spilled registers and
// ~1 GB of scratch writes per launch for no gain; the second wave of the SIMD hides the latency.)
GL_DEV bool load_pt(const Lds& D, const Map& mp, FlagW fw, const double* __restrict__ gobn, const double* __restrict__ gnd,
                    const int32_t* __restrict__ gassoc, int i, PtCtx& c) {
  c.fl = fw_get(fw, i);
  if (!(c.fl & (F_AR | F_AG | F_AF))) return false;
  c.l = mp.base + slot_off(mp, i);
  c.ll = mp.lbase + slot_off(mp, i);
#pragma unroll
  for (int j = 0; j < 3; ++j) c.ob[j] = gobn[(size_t)c.l * 3 + j];
  const int a = gassoc[c.l];
  const int ap = a > 0 ? a : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.nd[j] = gnd[(size_t)ap * 4 + j];  // plane normal n and n . mean: the map's table, by component
  c.ar = c.fl & F_AR;
  c.ag = c.fl & F_AG;
  c.af = kFixed && (c.fl & F_AF);
  const int oc = (c.fl >> 8) & 7;
  c.sx = D.stab[oc];
  c.sy = D.stab[8 + oc];
  c.asc = (c.fl & F_ASSOC) && !(c.fl & F_DEG) ? a : -1;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.p[j] = D.sp[j * MCAP + c.ll];
  return true;
}

// the same for point l with its flag word from memory (pass B, chunks dealt dynamically)
GL_DEV bool load_pt_at(const Lds& D, int l, int fl, const double* __restrict__ gobn, const double* __restrict__ gnd,
                       const int32_t* __restrict__ gassoc, PtCtx& c) {
  c.fl = fl;
  if (!(c.fl & (F_AR | F_AG | F_AF))) return false;
  c.l = l;
  c.ll = l;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.ob[j] = gobn[(size_t)c.l * 3 + j];
  const int a = gassoc[c.l];
  const int ap = a > 0 ? a : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.nd[j] = gnd[(size_t)ap * 4 + j];
  c.ar = c.fl & F_AR;
  c.ag = c.fl & F_AG;
  c.af = kFixed && (c.fl & F_AF);
  const int oc = (c.fl >> 8) & 7;
  c.sx = D.stab[oc];
  c.sy = D.stab[8 + oc];
  c.asc = (c.fl & F_ASSOC) && !(c.fl & F_DEG) ? a : -1;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.p[j] = D.sp[j * MCAP + c.ll];
  return true;
}

// SPREAD: a thread owns ONE point for the whole kernel, so what load_pt fetches from global memory - normalised
// observation, association, plane record: two dependent L2 round trips at the head of every pass, on a SIMD that has
// nothing else to run - is fetched once and kept in registers.
struct PtConst {
  double ob[3], nd[4];
  int a;
};
GL_DEV void load_const(const Map& mp, FlagW fw, const double* __restrict__ gobn, const double* __restrict__ gnd,
                       const int32_t* __restrict__ gassoc, PtConst& pc) {
#pragma unroll
  for (int j = 0; j < 3; ++j) pc.ob[j] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) pc.nd[j] = 0.0;
  pc.a = -1;
  if (!(fw_get(fw, 0) & F_EXISTS)) return;
  const int l = mp.base;
#pragma unroll
  for (int j = 0; j < 3; ++j) pc.ob[j] = gobn[(size_t)l * 3 + j];
  pc.a = gassoc[l];
  const int ap = pc.a > 0 ? pc.a : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) pc.nd[j] = gnd[(size_t)ap * 4 + j];
}
GL_DEV bool load_pt_const(const Lds& D, const Map& mp, FlagW fw, const PtConst& pc, PtCtx& c) {
  c.fl = fw_get(fw, 0);
  if (!(c.fl & (F_AR | F_AG))) return false;
  c.l = mp.base;
  c.ll = mp.lbase;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.ob[j] = pc.ob[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) c.nd[j] = pc.nd[j];
  c.ar = c.fl & F_AR;
  c.ag = c.fl & F_AG;
  c.af = false;  // (the fixed-observer instances are batch-shaped only)
  const int oc = (c.fl >> 8) & 7;
  c.sx = D.stab[oc];
  c.sy = D.stab[8 + oc];
  c.asc = (c.fl & F_ASSOC) && !(c.fl & F_DEG) ? pc.a : -1;
#pragma unroll
  for (int j = 0; j < 3; ++j) c.p[j] = D.sp[j * MCAP + c.ll];
  return true;
}

// ---- per-point bodies of the passes (shared by both kernels) -------------------------------------------------
// computeLambdaInit: pose-block terms 0..20 of the undamped reprojection Hessian, and the largest diagonal of the
// point block (world frame) into md
template <class Sink>
GL_DEV void pt_lambda_init(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& P, const PtCtx& c, bool robust, double& md, const Sink& sk) {
  Lin o;
  if (kFixed && c.af) {
    double Hx[6], bx[3];
    fixed_lin(U, D, P, c, robust, Hx, bx, false);
    lin_fast(U, gm, D, P, c, robust, o);
    fixed_add(Hx, bx, o);
  } else {
    lin_fast(U, gm, D, P, c, robust, o);
  }
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
    pose_terms(o.q, o.A, zero, false, sk);
  }
}
// the same with only what computeLambdaInit reads of the pose block: its DIAGONAL (terms 0, 6, 11, 15, 18, 20 of pose_terms, the same
// expressions - the same bits) as terms 0..5
template <class Sink>
GL_DEV void pt_lambda_init_diag(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& P, const PtCtx& c, bool robust, double& md, const Sink& sk) {
  Lin o;
  if (kFixed && c.af) {
    double Hx[6], bx[3];
    fixed_lin(U, D, P, c, robust, Hx, bx, false);
    lin_fast(U, gm, D, P, c, robust, o);
    fixed_add(Hx, bx, o);
  } else {
    lin_fast(U, gm, D, P, c, robust, o);
  }
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
    const double* q = o.q;
    const double* C = o.A;
    const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
    const double M1 = fma(q[1], Cf[7], -q[2] * Cf[4]), M2 = fma(q[1], Cf[8], -q[2] * Cf[5]);
    const double M3 = fma(q[2], Cf[0], -q[0] * Cf[6]), M5 = fma(q[2], Cf[2], -q[0] * Cf[8]);
    const double M6 = fma(q[0], Cf[3], -q[1] * Cf[0]), M7 = fma(q[0], Cf[4], -q[1] * Cf[1]);
    sk.put(0, fma(q[1], M2, -q[2] * M1));
    sk.put(1, fma(q[2], M3, -q[0] * M5));
    sk.put(2, fma(q[0], M7, -q[1] * M6));
    sk.put(3, C[0]);
    sk.put(4, C[3]);
    sk.put(5, C[5]);
  }
}
// pass A: linearise, point solve, Schur terms 0..26, robust chi2 (27), sum u.b (28); leaves the factors of D in the slot and the
// Huber weight rho' of the reprojection edge in the stale-chi2 cell (dead until pass B rewrites it)
template <class Sink>
GL_DEV void pt_pass_a(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& P, const PtCtx& c, bool robust, double lambda,
                      const Sink& sk) {
  Lin o;
  if (kFixed && c.af) {
    double Hx[6], bx[3];
    const double chi_f = fixed_lin(U, D, P, c, robust, Hx, bx, true);
    lin_fast(U, gm, D, P, c, robust, o);
    fixed_add(Hx, bx, o);
    sk.put(27, (o.rho0_r + o.chi_g) + chi_f);
  } else {
    lin_fast(U, gm, D, P, c, robust, o);
    sk.put(27, kFixed ? (o.rho0_r + o.chi_g) + 0.0 : o.rho0_r + o.chi_g);
  }
  double Df[6], u[3];
  point_solve_fast(o, lambda, Df, u);
  sk.put(28, fma(u[0], o.b[0], fma(u[1], o.b[1], u[2] * o.b[2])));
  if (kStep32) {
    int* un = (int*)D.un;
#pragma unroll
    for (int j = 0; j < 3; ++j) un[j * MCAP + c.ll] = __float_as_int((float)u[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 6; ++j) un_st(D.un + (unsigned)(j * MCAP + c.ll), Df[j]);
  }
  if (c.ar) {
    if (!kStep32) D.chir[c.ll] = o.rho1;
    double C[6], cc[3], AD[9];
    ad_solve(o.A, Df, AD);
    if (kStep32) {
      int* un = (int*)D.un;
#pragma unroll
      for (int j = 0; j < 9; ++j) un[(3 + j) * MCAP + c.ll] = __float_as_int((float)AD[j]);
    }
    // Schur contribution of the point without the subtraction A - A D^-1 A.  With M = D - A = lambda I + (GMM block) the
    // same matrix is M D^-1 A - a plain product - and the reduced right-hand side a - A u is M u - b_gmm.  For a point whose
    // only strong constraint is its reprojection (no GMM edge, small lambda) the result is of the size of lambda while A is
    // ~1e4: the subtracted form keeps ~5 of its 16 digits there, and on a frame whose reduced system is itself near
    // singular (gauge direction held by lambda alone, cond ~1e11) that noise decides accept / reject steps of the
    // Levenberg path (soak frame v1 r19656: 1.75e-5 m off an oracle that 400 perturbed runs do not move; with this form HIP
    // follows the oracle's path step for step, profiles/history/r2f_track_v1_r19656_trace_*.txt).
    {
      const double M[9] = {(o.D[0] - o.A[0]) + lambda, o.D[1] - o.A[1], o.D[2] - o.A[2],
                           o.D[1] - o.A[1], (o.D[3] - o.A[3]) + lambda, o.D[4] - o.A[4],
                           o.D[2] - o.A[2], o.D[4] - o.A[4], (o.D[5] - o.A[5]) + lambda};
      // C = M (A D^-1)^T:  C(r, j) = sum_k M(r, k) AD(j, k), upper triangle
      const int ri[6] = {0, 0, 0, 1, 1, 2}, ci[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int r = ri[e], j = ci[e];
        C[e] = fma(M[r * 3], AD[j * 3], fma(M[r * 3 + 1], AD[j * 3 + 1], M[r * 3 + 2] * AD[j * 3 + 2]));
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) cc[r] = fma(M[r * 3], u[0], fma(M[r * 3 + 1], u[1], fma(M[r * 3 + 2], u[2], o.a[r] - o.b[r])));
    }
    pose_terms(o.q, C, cc, true, sk);
  } else if (kStep32) {
    int* un = (int*)D.un;
#pragma unroll
    for (int j = 0; j < 9; ++j) un[(3 + j) * MCAP + c.ll] = 0;
  }
}
// pass B, first half: point step and trial point (in place, old point backed up in the slot) - needs the pose STEP dx
// only.  term 0 = |eps|^2
template <class Sink>
GL_DEV void pt_pass_b_step(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& P, const double* dx, const PtCtx& c, double* pn,
                           const Sink& sk) {
  // eps = D^-1 (b - A gd),  gd = omega x q + upsilon
  double q[3], gd[3], eps[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) q[j] = fma(P.R[j * 3], c.p[0], fma(P.R[j * 3 + 1], c.p[1], fma(P.R[j * 3 + 2], c.p[2], P.t[j])));
  cross(dx, q, gd);
  gd[0] += dx[3];
  gd[1] += dx[4];
  gd[2] += dx[5];
  const bool stereo = c.fl & F_STEREO;
  if (kStep32) {  // = u - (A D^-1)^T gd from the fp32 cache of pass A
    const int* un = (const int*)D.un;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double e = (double)__int_as_float(un[j * MCAP + c.ll]);
#pragma unroll
      for (int a = 0; a < 3; ++a) e -= (double)__int_as_float(un[(3 + a * 3 + j) * MCAP + c.ll]) * gd[a];
      eps[j] = e;
    }
  } else {  // exact: b - A gd = Jpi^T W (e - Jpi gd) + b_gmm from the re-evaluated residual, solved with the cached factors of D
    double rhs[3] = {0.0, 0.0, 0.0};
    if (kFixed && c.af) {  // the fixed edges do not couple to the pose step: their part of b, at the linearisation point
      double bw[3];
      fixed_sum(U, D, c, false, nullptr, bw, -1);
#pragma unroll
      for (int j = 0; j < 3; ++j) rhs[j] = fma(P.R[j * 3], bw[0], fma(P.R[j * 3 + 1], bw[1], P.R[j * 3 + 2] * bw[2]));
    }
    if (c.ar) {
      const double rho1 = D.chir[c.ll];  // the edge's Huber weight, left there by pass A
      const double iz = rcp_nr(q[2]);
      const Jpi J = jpi_n(q, iz, U.bn);
      const double wx = rho1 * c.sx, wy = rho1 * c.sy;
      // f_r = w_r (e_r - j_r . gd),  e_r - j_r . gd = obs_r - iz (q_r + gd_r) - c_r gd_z
      const double f0 = wx * fma(-J.c0, gd[2], fma(-(q[0] + gd[0]), iz, c.ob[0]));
      const double f1 = wy * fma(-J.c1, gd[2], fma(-(q[1] + gd[1]), iz, c.ob[1]));
      const double f2 = stereo ? wx * fma(-J.c2, gd[2], fma(U.bn - (q[0] + gd[0]), iz, c.ob[2])) : 0.0;
      const double r0 = iz * (f0 + f2), r1 = iz * f1, r2 = fma(J.c2, f2, fma(J.c1, f1, J.c0 * f0));
      rhs[0] = kFixed ? rhs[0] + r0 : r0;
      rhs[1] = kFixed ? rhs[1] + r1 : r1;
      rhs[2] = kFixed ? rhs[2] + r2 : r2;
    }
    if (c.ag) {
      if (c.fl & F_DEG) {
        const double nx = c.nd[0], ny = c.nd[1], nz = c.nd[2];
        const double eg = fma(nz, c.p[2], fma(ny, c.p[1], nx * c.p[0])) - c.nd[3];
        const double m = -U.lm * eg;
#pragma unroll
        for (int j = 0; j < 3; ++j) rhs[j] = fma(m, fma(P.R[j * 3 + 2], nz, fma(P.R[j * 3 + 1], ny, P.R[j * 3] * nx)), rhs[j]);
      } else {
        double bc[3];
        gmm_nondeg(gm, c.asc, P.R, c.p, nullptr, bc);
#pragma unroll
        for (int j = 0; j < 3; ++j) rhs[j] += bc[j];
      }
    }
    double Df[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) Df[j] = un_ld(D.un + (unsigned)(j * MCAP + c.ll));
    ldl3_solve_fast(Df, rhs, eps);
  }
  sk.put(0, eps[0] * eps[0] + eps[1] * eps[1] + eps[2] * eps[2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) pn[j] = c.p[j] + (P.R[j] * eps[0] + P.R[3 + j] * eps[1] + P.R[6 + j] * eps[2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) D.sp[j * MCAP + c.ll] = pn[j];  // the trial point goes in place,
  backup_point(D, c.ll, c.p);                                  // the old point into the slot
}
// pass B, second half: chi2 at the trial state (trial pose Pn, trial point pn).  term 1 = robust chi2 of the point's edges
template <class Sink>
GL_DEV void pt_pass_b_eval(const Uni& U, const GmmDev& gm, const Lds& D, const Pose& Pn, const PtCtx& c, const double* pn, bool robust,
                           const Sink& sk) {
  const bool stereo = c.fl & F_STEREO;
  double chi = 0.0;
  if (c.ar) {
    double qn[3], e[3], iz;
#pragma unroll
    for (int j = 0; j < 3; ++j) qn[j] = fma(Pn.R[j * 3], pn[0], fma(Pn.R[j * 3 + 1], pn[1], fma(Pn.R[j * 3 + 2], pn[2], Pn.t[j])));
    const double c2 = reproj_n(qn, c.ob, stereo, U.bn, c.sx, c.sy, e, iz);
    D.chir[c.ll] = c2;
    double r0 = c2, r1;
    if (robust) huber_pt(D, stereo, c2, r0, r1);
    chi = r0;
  }
  if (c.ag) chi += gmm_chi2_fast(U.lm, gm, c.nd, c.fl, c.asc, pn);
  if (kFixed && c.af) chi += fixed_chi(U, D, c, pn, robust);
  sk.put(1, chi);
}

// Issue priority inside a pass (DENSE, 8-wave frames).  The two waves of a SIMD (w and w + 4) run the same code; the
// arbiter favours the older one, which then finishes its 4 slots a third of the pass early and leaves the younger
// one alone on the SIMD at single-wave speed (measured: 11.6 k vs 18.3 k cycles).  Waves 4..7 sit at priority 1; waves
// 0..3 start a pass at 2 and drop to 0 for their last slot, so that the pair ends the pass together.
#if defined(GL_BAF_NO_PRIO)
#define GL_BAF_PRIO_PASS_BEGIN()
#define GL_BAF_PRIO_SLOT(i)
#else
#define GL_BAF_PRIO_PASS_BEGIN() \
  if (NWC == 8 && !kTwo && (threadIdx.x >> 6) < 4) __builtin_amdgcn_s_setprio(2)
#define GL_BAF_PRIO_SLOT(i) \
  if (NWC == 8 && !kTwo && (threadIdx.x >> 6) < 4 && (i) == (mp.asym ? GL_BAF_PRIO_DROP_ASYM : GL_BAF_PRIO_DROP)) __builtin_amdgcn_s_setprio(0)
#endif
#if defined(GL_BAF_NO_PRIO)
#define GL_BAF_PRIO_PASS_END()
#else
#define GL_BAF_PRIO_PASS_END() \
  if (NWC == 8 && !kTwo && (threadIdx.x >> 6) < 4) __builtin_amdgcn_s_setprio(1)  /* a dynamically dealt pass: every wave at the same priority */
#endif
#ifndef GL_BAF_PRIO_DROP_ASYM
#define GL_BAF_PRIO_DROP_ASYM 4
#endif
#ifndef GL_BAF_PRIO_DROP
#define GL_BAF_PRIO_DROP 3
#endif

#if defined(GL_BA_PROF) && defined(GL_BA_PROF_LOADS)  // (diagnosis only: the wait changes the schedule of the slot's head)
#define GL_BAF_PROF_LOADS(i)          \
  __builtin_amdgcn_s_waitcnt(0);      \
  PROF_Q(trials, prof_pass, i, 0)
#else
#define GL_BAF_PROF_LOADS(i)
#endif
// one pass over the thread's points: DENSE accumulates the terms in acc[] (level 1), SPREAD leaves the single
// point's terms there (zeros when the thread has no active point)
#define GL_BAF_PASS(BODY) GL_BAF_PASS2(BODY, )
// (MID: GPW = 2, what happens between the slots of the wave's first and second group - the first group's totals leave the accumulators)
#define GL_BAF_PASS2(BODY, MID)                                               \
  {                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) acc[i_] = 0.0;          \
    if (kSpread) {                                                            \
      const SinkSet sk{acc};                                                  \
      const int i = 0;                                                        \
      (void)i;                                                                \
      PtCtx c;                                                                \
      if (load_pt_const(D, mp, fw, pc, c)) { BODY; }                          \
    } else {                                                                  \
      const SinkAcc sk{acc};                                                  \
      GL_BAF_PRIO_PASS_BEGIN();                                               \
      _Pragma("unroll 1") for (int i = 0; i < mp.S; ++i) {                    \
        if (kTwo && i == 4) { MID; }                                          \
        GL_BAF_PRIO_SLOT(i);                                                  \
        PROF_S(trials, prof_pass, i);                                         \
        PtCtx c;                                                              \
        if (!load_pt(D, mp, fw, gobn, gnd, gassoc, i, c)) continue;               \
        GL_BAF_PROF_LOADS(i);                                                 \
        BODY;                                                                 \
        PROF_Q(trials, prof_pass, i, 2);                                      \
      }                                                                       \
    }                                                                         \
  }

// SparseOptimizer::optimize(iters), Levenberg
GL_DEV int optimize_fast(const Uni& U, const GmmDev& gm, const Lds& D, const Map& mp, FlagW fw, Pose& P,
                         const double* __restrict__ gobn, const int32_t* __restrict__ gassoc, const double* __restrict__ gnd,
                         const PtConst& pc, bool robust, int iters, const Red& R, int& trials, Coop& C, Anchor& An, const int32_t* gpfl) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  {
    const int ns = kSpread ? 1 : mp.S;
#pragma unroll 1
    for (int i = 0; i < ns; ++i) {
      const int fl = fw_get(fw, i);
      if (fl & F_AR) acc[0] += 1.0;
      if (fl & (F_AR | F_AG | F_AF)) acc[1] += 1.0;
    }
  }
  reduce2<2>(acc, R, C);
  // gl_ctx_set_edge_stats_buffer: the level-0 reprojection edges of this optimize() call (the levels only change between the calls)
  // and the trial count on entry wait in LDS for the end of the call - nothing of the statistic is live across the trial loops
  int* const est = (int*)(R.tot + 62);  // {edges x trials, edges x outer iterations, edges of this call, trials on entry}
  if (threadIdx.x == 0) {
    est[2] = (int)acc[0];
    est[3] = trials;
  }
  const bool pose_active = kPrior ? !An.pose_fixed && (acc[0] > 0.0 || An.has_prior) : acc[0] > 0.0;
  const bool prior_on = kPrior && An.has_prior && pose_active;
  if (!pose_active && !(acc[1] > 0.0)) return -1;

  double lambda = 0.0, ni = 2.0;
  int cj = 0;
  int prof_pass = 2;  // (GL_BA_PROF: which pass the slot markers belong to; 2 = the lambda initialisation, not recorded)
  (void)prof_pass;
  for (int it = 0; it < iters; ++it) {
    double rho = 0.0, currentChi = 0.0;
    int qmax = 0;
    if (it == 0) {  // computeLambdaInit
      PROF_T(tL0);
      double md = 0.0;
      if (!kSpread) {
        // DENSE: only the six diagonal sums, in the canonical order (level 2 = the butterfly of wave_allreduce_canon, level 3 = the blocks
        // of two groups), and the maximum of the point blocks beside them through the same LDS row: one barrier instead of five
        // (GPW = 2: the first group's six sums and the maximum so far go to the wave's first row between the two groups' slots - behind
        // the same barrier, which every wave reaches: the hook sits in front of the slot's activity test)
        GL_BAF_PASS2(pt_lambda_init_diag(U, gm, D, P, c, robust, md, sk), {
          double sd0[6];
          _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) sd0[i_] = wave_allreduce_canon(acc[i_]);
          double m0 = md;
          _Pragma("unroll") for (int o_ = 1; o_ < 64; o_ <<= 1) m0 = fmax(m0, shfl_xor_f64(m0, o_));
          __syncthreads();
          const int l0_ = threadIdx.x & 63;
          if (l0_ < 7) {
            double v0 = m0;
            _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) v0 = l0_ == i_ ? sd0[i_] : v0;
            R.red[(threadIdx.x >> 6) * 32 + l0_] = v0;
          }
          _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) acc[i_] = 0.0;
        });
        const int lane_ = threadIdx.x & 63, wave_ = kTwo ? (int)(threadIdx.x >> 6) + NWV : (int)(threadIdx.x >> 6);
        double sd[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) sd[i] = wave_allreduce_canon(acc[i]);
#pragma unroll
        for (int o_ = 1; o_ < 64; o_ <<= 1) md = fmax(md, shfl_xor_f64(md, o_));
        __syncthreads();  // (the rows of `red` may still be read: the previous trial's reduction)
        if (lane_ < 7) {
          double v = md;
#pragma unroll
          for (int i = 0; i < 6; ++i) v = lane_ == i ? sd[i] : v;
          R.red[wave_ * 32 + lane_] = v;
        }
        __syncthreads();
        double dg[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double s_ = NWC > 1 ? add_nc(R.red[i], R.red[32 + i]) : R.red[i];
#pragma unroll
          for (int b = 1; b < NWC / 2; ++b) s_ = add_nc(s_, add_nc(R.red[(2 * b) * 32 + i], R.red[(2 * b + 1) * 32 + i]));
          dg[i] = s_;
        }
        double mx = R.red[6];
#pragma unroll
        for (int w_ = 1; w_ < NWC; ++w_) mx = fmax(mx, R.red[w_ * 32 + 6]);
        md = mx;
        if (pose_active) {
#pragma unroll
          for (int i = 0; i < 6; ++i) md = fmax(md, fabs(prior_on ? dg[i] + An.rec[An.cur * 32 + GL_U(i, i)] : dg[i]));
        }
        __syncthreads();  // (the rows are rewritten by pass A's reduction)
      } else {
      GL_BAF_PASS(pt_lambda_init(U, gm, D, P, c, robust, md, sk));
      reduce2<21>(acc, R, C);
      if (pose_active) {
        if (prior_on) {
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[GL_U(i, i)] += An.rec[An.cur * 32 + GL_U(i, i)];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) md = fmax(md, fabs(acc[GL_U(i, i)]));
      }
      md = reduce_max(md, R, C);
      }
      lambda = uni(1e-5 * md);
      ni = 2.0;
      PROF_T(tL1);
      PROF_KADD(12, tL0, tL1);
    }
    do {
      PROF_T(tA0);
      // ---- pass A ---------------------------------------------------------------------------
      prof_pass = 0;
      GL_BAF_PASS2(pt_pass_a(U, gm, D, P, c, robust, lambda, sk), group_flush29(acc, R, (int)(threadIdx.x >> 6)));
      PROF_S(trials, 0, 4);
      PROF_T(tA1);
      PROF_W(trials, 0);
      if (kSpread) spread_reduce29_w0(acc, R, C);
      else reduce29_w0_dense(acc, R, kAllSolve);
      PROF_W(trials, 1);
      PROF_T(tA2);
      double* bc = R.tot + 32;  // 28 doubles: dx[6] R[9] t[3] ok g[6] sum u.b chi2, chi2 of the prior edge at the trial pose
      volatile int* pnflag = (volatile int*)(R.tot + 60);
      const int seq = trials + 1;
      double dx[6], gsc[7];  // the pose step; reduced rhs g and sum u.b, for computeScale
      bool ok2;
      Pose Pn = P;
      bool have_pn = false;
      if (kAllSolve) {
        // DENSE: EVERY wave solves the 6 x 6 system from the totals it has just added up itself, and builds the trial pose: the
        // serial section has no hand-over - no second barrier, no broadcast through LDS, no wait for the trial pose (the
        // ~350 dependent instructions run on all eight waves at once, two per SIMD, in each other's latencies)
        double dxs[6] = {0, 0, 0, 0, 0, 0};
        bool ok = true;
        if (prior_on) {  // the prior edge at the current pose: H_pp, b_p and chi2 from its record
          const double* rc = An.rec + An.cur * 32;
#pragma unroll
          for (int i = 0; i < 28; ++i) acc[i] += rc[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) gsc[i] = acc[21 + i];
        gsc[6] = acc[28];
        if (qmax == 0) currentChi = acc[27];
        if (pose_active) ok = GL_BAF_SOLVE_BLOCKED ? ldlt6_blocked(acc, acc + 21, lambda, dxs) : ldlt6_packed(acc, acc + 21, lambda, dxs);
#pragma unroll
        for (int i = 0; i < 6; ++i) dx[i] = dxs[i];
        ok2 = ok;
        if (pose_active && ok2) Pn = pose_uni(pose_update(P, dx));
        have_pn = true;
      } else {
      // 6x6 solve by wave 0; the step and the status are broadcast through LDS behind a barrier.  The trial pose
      // exp(dx) P is computed by wave 0 AFTER that barrier, while the other waves are already in pass B (its first use
      // is the evaluation of their first point, ~200 instructions in), and handed over through LDS + a sequence word.
      if (threadIdx.x < 64) {
        double dxs[6] = {0, 0, 0, 0, 0, 0};
        bool ok = true;
        if (prior_on) {  // the prior edge at the current pose: H_pp, b_p and chi2 from its record
          const double* rc = An.rec + An.cur * 32;
#pragma unroll
          for (int i = 0; i < 28; ++i) acc[i] += rc[i];
        }
        if (pose_active) ok = GL_BAF_SOLVE_BLOCKED ? ldlt6_blocked(acc, acc + 21, lambda, dxs) : ldlt6_packed(acc, acc + 21, lambda, dxs);
        if (threadIdx.x == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) bc[i] = dxs[i];
          bc[18] = ok ? 1.0 : 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) bc[19 + i] = acc[21 + i];  // reduced rhs g and sum u.b, for computeScale
          bc[25] = acc[28];
          bc[26] = acc[27];  // robust chi2 at the linearisation point
          if (kDynB) *(int*)(D.stab + 20) = 0;  // pass B's chunk queue
        }
      }
      __syncthreads();
      if (qmax == 0) currentChi = uni(bc[26]);
#pragma unroll
      for (int i = 0; i < 6; ++i) dx[i] = uni(bc[i]);
#pragma unroll
      for (int i = 0; i < 7; ++i) gsc[i] = uni(bc[19 + i]);
      ok2 = uni(bc[18]) != 0.0;
      if (threadIdx.x < 64) {
        Pose Pw = P;
        if (pose_active && ok2) Pw = pose_update(P, dx);
        if (threadIdx.x == 0) {
#pragma unroll
          for (int i = 0; i < 9; ++i) bc[6 + i] = Pw.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) bc[15 + i] = Pw.t[i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          *pnflag = seq;
        }
        Pn = pose_uni(Pw);
        have_pn = true;
      }
      }
      PROF_T(tS);
      PROF_W(trials, 2);
      // ---- pass B ---------------------------------------------------------------------------
      // the trial pose arrives while the step half of the first point is under way (wave 0 has it already)
#define GL_BAF_GET_PN()                                                                          \
  if (!have_pn) {                                                                                \
    while (*pnflag != seq) __builtin_amdgcn_s_sleep(1);                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 9; ++i_) Pn.R[i_] = uni(bc[6 + i_]);                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) Pn.t[i_] = uni(bc[15 + i_]);                \
    have_pn = true;                                                                              \
  }
      prof_pass = 1;
      if (kDynB) {
        // The chunks of pass B are dealt DYNAMICALLY: a wave that is through a chunk takes the next one from a counter in LDS.  The
        // two waves of a SIMD do not run at the same speed (the arbiter serves the older one first), and with four fixed chunks each
        // the older one is done a third of the pass before the younger, which then runs alone at half the issue rate
        // (profiles/r5_prof_ba_slots.txt).  Which wave evaluates a point does not show in the result: the point's two terms wait in
        // its slot, and the thread that OWNS the point in the canonical order adds them - in the same order as ever - behind a barrier.
        // The flag word of a point it does not own comes from the launch scratch (bits 16..31 of pfl, kept current by the owners).
        const int nch = (mp.L + 63) >> 6;
        volatile int* qc = (volatile int*)(D.stab + 20);
#if GL_BAF_DYNB_PRIO == 0
        GL_BAF_PRIO_PASS_END();
#else
        GL_BAF_PRIO_PASS_BEGIN();  // the older wave of a SIMD keeps the higher priority: it just takes more of the chunks
#endif
#pragma unroll 1
        for (;;) {
          int cq = 0;
          if ((threadIdx.x & 63) == 0) cq = atomicAdd((int*)qc, 1);
          cq = __builtin_amdgcn_readfirstlane(cq);
          if (cq >= nch) break;
          const int l = cq * 64 + (int)(threadIdx.x & 63);
          const int fl = l < mp.L ? (int)((unsigned)__hip_atomic_load(gpfl + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) : 0;
          PtCtx c;
          if (!load_pt_at(D, l, fl, gobn, gnd, gassoc, c)) continue;
          const SinkSlot sk{D.un, c.ll};
          double pn[3];
          pt_pass_b_step(U, gm, D, P, dx, c, pn, sk);
          GL_BAF_GET_PN();
          pt_pass_b_eval(U, gm, D, Pn, c, pn, robust, sk);
        }
        GL_BAF_GET_PN();
        __syncthreads();
        acc[0] = acc[1] = 0.0;
#pragma unroll 1
        for (int i = 0; i < mp.S; ++i) {  // level 1 of the canonical order: the owner adds its points' terms in slot order
          if (!(fw_get(fw, i) & (F_AR | F_AG | F_AF))) continue;
          const int ll = mp.lbase + slot_off(mp, i);
          acc[0] = add_nc(acc[0], D.un[3 * MCAP + ll]);
          acc[1] = add_nc(acc[1], D.un[4 * MCAP + ll]);
        }
      } else {
      GL_BAF_PASS2({
        double pn[3];
        pt_pass_b_step(U, gm, D, P, dx, c, pn, sk);
        PROF_Q(trials, 1, i, 1);
        GL_BAF_GET_PN();
        pt_pass_b_eval(U, gm, D, Pn, c, pn, robust, sk);
      }, group_flush2(acc, R, (int)(threadIdx.x >> 6)));
      GL_BAF_GET_PN();  // (a wave without an active point still takes the pose: P = Pn on acceptance)
      }
      PROF_S(trials, 1, 4);
#undef GL_BAF_GET_PN
      if (prior_on && (int)(threadIdx.x >> 6) == An.wave) prior_record_wave(An.mi, Pn, An.rec + (An.cur ^ 1) * 32, An.work, An.rezero);
      PROF_T(tB1);
      PROF_W(trials, 3);
      if (kSpread) spread_reduce2_all(acc, R, C);
      else reduce2_dense(acc, R);
      PROF_T(tB2);
      // computeScale: sum_l eps.(lambda eps + b_l) + dx.(lambda dx + b_p).  With eps = u - D^-1 A gd the
      // b-terms collapse to  sum u.b + dx.g  (g = reduced rhs of pass A), so pass B needs no b at all.
      double scale = lambda * acc[0] + gsc[6];
      const double tempChi = ok2 ? (prior_on ? acc[1] + uni(An.rec[(An.cur ^ 1) * 32 + 27]) : acc[1]) : 1.7976931348623157e308;
      if (pose_active) {
#pragma unroll
        for (int i = 0; i < 6; ++i) scale += dx[i] * (lambda * dx[i] + gsc[i]);
      }
      scale += 1e-3;
      rho = (currentChi - tempChi) / scale;
      TRACE_TRIAL(trials, currentChi, tempChi, lambda, rho, dx);
      if (rho > 0 && isfinite(tempChi)) {
        const double uu = 2 * rho - 1;
        double alpha = 1. - uu * uu * uu;
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        P = Pn;
        if (kPrior) An.cur ^= 1;  // the record made at the trial pose is the current one now
      } else {
        lambda *= ni;
        ni *= 2;
        PROF_ADD(7, 0, 1);  // rejected trials
        const int ns = kSpread ? 1 : mp.S;
#pragma unroll 1
        for (int i = 0; i < ns; ++i) {  // discardTop: restore the backed-up points
          if (fw_get(fw, i) & (F_AR | F_AG | F_AF)) restore_point(D, mp.lbase + slot_off(mp, i));
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
  if (threadIdx.x == 0) {
    est[0] += est[2] * (trials - est[3]);
    est[1] += est[2] * cj;
  }
  return cj;
}

// DENSE: <<<B, 64 G>>>, one workgroup per frame.  SPREAD: <<<B G, 256>>>, a workgroup per group of the frame
// (cooperative launch when G > 1).  G, S: the canonical order of stride L (launcher: canon_order()).  The frame's points
// are addressed through the permutation k_ba1_prep made (points associated with a NON-degenerate component last), so
// "point l" below is the l-th point of that order; pts_io / assoc_all are read and written through perm.
// (the body of the kernel: ONE frame - `fblock` is the block index of the plain launch, or the frame a persistent workgroup has drawn)
// a struct out of the kernel-argument segment (constant address space), word by word (scalar loads)
template <class T>
GL_DEV T ld_karg(const T __attribute__((address_space(4)))* p) {
  static_assert(sizeof(T) % 4 == 0, "");
  union U {
    T v;
    unsigned w[sizeof(T) / 4];
    GL_DEV U() {}
  } u;
  const unsigned __attribute__((address_space(4)))* q = (const unsigned __attribute__((address_space(4)))*)p;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) u.w[i] = q[i];
  return u.v;
}

typedef const BafKArgs __attribute__((address_space(4))) kargs_t;
GL_DEV void ba1_fast_frame(double* smem, const unsigned fblock, kargs_t* ka) {
  // the arguments the set-up and the passes use; the caller's OUTPUT arrays are read from the argument segment again where the
  // results are written (an opaque copy of the pointer: the compiler may not keep those ten pointers live across the passes)
  const BaK __attribute__((address_space(4)))& k = ka->k;
  const GmmDev gm = ld_karg(&ka->gm);
  const FixedV fxv = ld_karg(&ka->fxv);
  const int B = ka->B, L = ka->L, G = ka->G, S = ka->S, NB = ka->NB, xcc_trusted = ka->xcc_trusted, nb_prev = ka->nb_prev;
  double* const pose_io = ka->pose_io;
  double* const pts_io = ka->pts_io;
  int32_t* const assoc_all = ka->assoc_all;
  double* const pn_all = ka->pn_all;
  unsigned long long* const parts = ka->parts;
  int* const ctl = ka->ctl;
  const long long limit = ka->limit;
  const int32_t* const oct_all = ka->oct_all;
  const uint8_t* const prior_all = ka->prior_all;
  const double* const prior_mi = ka->prior_mi;
  double* const stage = ka->stage;
  int32_t* const counters = ka->counters;
  Lds D;
  D.sp = smem;                      // 3 * MCAP
  D.chir = D.sp + 3 * MCAP;         // MCAP
  // (GPW = 2: the hand-over slots of this workgroup's frame in global memory - one region per workgroup of the launch)
  D.un = kTwo ? ka->un_scratch + (size_t)blockIdx.x * 6 * MCAP : D.chir + MCAP;  // 6 * MCAP
  Red R;
  R.red = D.chir + (kTwo ? 1 : 7) * MCAP;  // NRED * 32
  R.tot = R.red + NRED * 32;        // 32 (+ 32 broadcast slots)
  D.stab = R.tot + 64;              // 24
  R.red2 = D.stab + 24;             // 16
  double* const prec = R.red2 + (NWC > 8 ? 2 * NWC : 16); // anchored instances: 2 x 32, the prior edge's records
  // ... and the work area of prior_record_wave: the group totals `red` where they are large enough (4 / 8 groups) - idle
  // during pass B and before the first reduction -, the transpose rows pass B does not use on the latency shape, 108 doubles
  // of its own in the small class (which has LDS to spare)
  constexpr int kPriorWorkOwn = (kPrior && !kSpread && NWC < 4) ? 108 : 0;
  D.frt = prec + (kPrior ? 64 : 0) + kPriorWorkOwn;  // fixed-observer instances: F x 12, the key-frames' poses {R, t}
  R.tb = D.frt + (kFixed ? kMaxFixed * 12 : 0);      // SPREAD: 29 x TSP
  double* const pwork = kSpread ? R.tb + 2 * TSP : (NWC >= 4 ? R.red : prec + 64);
  D.F = 0;
  D.Lf = L;
  D.gfobn = nullptr;
  D.gfoct = nullptr;
  D.gchif = nullptr;
  R.S = S;
  FlagW fw = {0ull, 0u};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  PROF_K(0);
#ifdef GL_BA_PROF
  if (blockIdx.x == 1024 && threadIdx.x == 0) g_prof_k[12] = 0;
  const unsigned long long prof_wg_t0 = wall_clock64();  // (100 MHz, the same counter on every CU)
#endif
  // SPREAD: the workgroups of a frame are given block indices that are equal modulo 8, which is what puts them on ONE XCD
  // on this hardware (observed, not promised: a matter of speed only) - block 8 k + x is group k % 8 of frame x + 8 (k / 8)
  const int f = kSpread ? (NB > 1 ? (int)(fblock & 7) + 8 * (int)(fblock >> 6) : (int)fblock) : (int)fblock;
  const int pb_ = kSpread && NB > 1 ? (int)((fblock >> 3) & 7) : 0;
  if (f >= B || pb_ >= NB) return;
  // latency-shape launches write to a staging area {points B x L x 3 | pose B x 7 (stride 8) | association B x L}
  double* const st_pts = stage;
  double* const st_pose = stage ? stage + (size_t)B * L * 3 : nullptr;
  int32_t* const st_assoc = stage ? (int32_t*)(st_pose + (size_t)B * 8) : nullptr;
  if (!kSpread && ctl) {  // follow-up of a latency-shape launch
    if (ctl[2 * f + 1] == nb_prev) {  // every workgroup of the frame finished there: its staged result becomes the answer
      const int32_t* oc = oct_all + (size_t)f * L;
      const int NT = blockDim.x;
      for (int l0 = threadIdx.x; l0 < L; l0 += 4 * NT) {  // four points per step, their loads requested together
        int o_[4], a_[4];
        double p_[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = min(l0 + q * NT, L - 1);
          const size_t g = (size_t)f * L + l;
          o_[q] = oc[l];
          a_[q] = st_assoc[g];
#pragma unroll
          for (int j = 0; j < 3; ++j) p_[q][j] = st_pts[g * 3 + j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = l0 + q * NT;
          if (l >= L) break;
          const size_t g = (size_t)f * L + l;
          if (o_[q] >= 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pts_io[g * 3 + j] = p_[q][j];
          }
          assoc_all[g] = a_[q];
        }
      }
      if (threadIdx.x < 7) pose_io[(size_t)f * 7 + threadIdx.x] = st_pose[(size_t)f * 8 + threadIdx.x];
      return;
    }
    if (counters && threadIdx.x == 0) atomicAdd(&counters[0], 1);  // GL_COUNTER_BA_REDONE: recomputed here, from the untouched inputs
  }
  Coop C{kSpread && NB > 1 ? parts + (size_t)f * 2 * NB * 64 : nullptr, kSpread ? NB : 1, pb_, 0u,
         ctl ? ctl + 2 * f : nullptr, (int*)(R.tot + 61), limit, 0, 0};
  Map mp;
  mp.L = L;
  if (kSpread) {  // workgroup pb = group pb; wave = slot; idle waves beyond S
    mp.S = 1;
    mp.asym = false;
    mp.g2 = false;
    mp.base = wave < S && C.pb < G ? (C.pb + G * wave) * 64 + lane : L;  // chunk g + G slot of group g
    mp.lbase = tid;
    mp.step = 0;
  } else {
    mp.asym = kAsym && G == 8;
    mp.S = kTwo ? 8 : mp.asym ? (wave < 4 ? 5 : 3) : S;  // (GPW = 2: four slots per group; a slot beyond the frame's S chunks per group lies behind L)
    mp.g2 = kTwo && wave + NWV < G;
    mp.base = wave * 64 + lane;  // group = wave: chunks wave, wave + G, ... (slot_off)
    mp.lbase = mp.base;
    mp.step = 64 * G;
  }
  const size_t gbase = (size_t)f * L;
  // launch scratch written by k_ba1_prep, read-only here: plane records, normalised observations, permutation, flags,
  // gated associations (all but perm in the permuted order)
  const PrepView pv = prep_view(pn_all, B, L);
  const double* gnd = gm.plane4;
  const double* gobn = pv.gobn + gbase * 3;
  const int32_t* gperm = pv.perm + gbase;
  const int32_t* gassoc = pv.assoc_p + gbase;
  const Uni U{uni(k.bf / k.fx), uni(k.ba_lambda2)};
  if (kFixed) {  // this frame's slices of the fixed-observer records (k_ba1_prep), its key-frames' poses into LDS
    D.F = fxv.F;
    D.gfobn = fxv.fobn + (size_t)f * fxv.F * L * 3;
    D.gfoct = fxv.foct + (size_t)f * fxv.F * L;
    D.gchif = fxv.chif + (size_t)f * fxv.F * L;
    if (tid < fxv.F * 12) D.frt[tid] = fxv.fRt[(size_t)f * fxv.F * 12 + tid];
  }
  for (int i = tid; i < NRED * 32; i += blockDim.x) R.red[i] = 0.0;
  if (tid < (NWC > 8 ? 2 * NWC : 16)) R.red2[tid] = 0.0;
  if (tid == 0) {
    *(int*)(R.tot + 60) = 0;  // sequence word of the trial-pose hand-over
    *(int*)(R.tot + 61) = 0;  // a poll of the exchange gave up (SPREAD)
    R.tot[62] = 0.0;          // edge statistics {edges x trials, edges x outer iterations} as two ints (optimize_fast)
  }
  {
    const int ns = kSpread ? 1 : mp.S;
#pragma unroll 1
    for (int i = 0; i < ns; ++i) {
      const int l = mp.base + slot_off(mp, i), ll = mp.lbase + slot_off(mp, i);
      if (slot_absent(mp, i)) continue;
      if (l >= L) {
        if (kTwo) continue;  // (the second group's slots follow)
        break;
      }
      const size_t g = gbase + gperm[l];
#pragma unroll
      for (int j = 0; j < 3; ++j) D.sp[j * MCAP + ll] = pts_io[g * 3 + j];
      D.chir[ll] = 0.0;
      const int pfl0 = pv.pfl[gbase + l] & 0xffff;  // (bits 16..31: the current flag word of an earlier launch on this scratch)
      fw_or(fw, i, pfl0);
      if (kFixed) {  // edges the point does not have (not observed by that key-frame / no such key-frame) count as inactive
#pragma unroll
        for (int kf = 0; kf < kMaxFixed; ++kf)
          if (kf >= D.F || D.gfoct[(size_t)kf * L + l] < 0) fw_or(fw, i, F_OFFF << kf);
      }
      fw_activity(fw, i);
      if (kDynB) pv.pfl[gbase + l] = pfl0 | (fw_get(fw, i) << 16);  // the current flag word, for whichever wave takes the point in pass B
    }
  }
  PtConst pc;
  if (kSpread) load_const(mp, fw, gobn, gnd, gassoc, pc);
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // static indices: a lane-indexed kernarg array would go through scratch
      D.stab[j] = k.s2inv[j] * (k.fx * k.fx);
      D.stab[8 + j] = k.s2inv[j] * (k.fy * k.fy);
    }
    D.stab[16] = k.delta_mono;
    D.stab[17] = k.delta_mono * k.delta_mono;
    D.stab[18] = k.delta_stereo;
    D.stab[19] = k.delta_stereo * k.delta_stereo;
  }
  Pose P = pose_from_se3(se3_load(pose_io + (size_t)f * 7));
  // vSE3->setFixed(idx_ == 0) or e->setMeasurement(kfi->getTcw()) (:556-581)
  const bool prior_flag = kPrior && prior_all ? prior_all[f] != 0 : false;
  Anchor An{prior_flag && k.first_as_prior != 0, prior_flag && k.first_as_prior == 0,
            (const cdouble_k*)(prior_mi + (size_t)f * 12), prec, 0, pwork, !kSpread && NWC >= 4,
            kSpread ? (int)(blockDim.x >> 6) - 1 : min(1, (int)(blockDim.x >> 6) - 1)};
#ifndef GL_BAF_NO_PRIO
  if (!kSpread && NWC == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  __syncthreads();
  if (kPrior && An.has_prior) {  // the record of the input pose (its work area may be the freshly zeroed group totals)
    if (wave == An.wave) prior_record_wave(An.mi, P, prec, pwork, An.rezero);
    __syncthreads();
  }
  if (kSpread && C.NB > 1 && xcc_trusted) {
    // do the frame's workgroups share an XCD?  Every contributing thread adds its XCC id and the id's square (through the
    // device-scope exchange: this is also the launch's rendezvous); they are all equal iff N sum(id^2) == sum(id)^2
    double x2[32];
    const double id = (double)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);
    x2[0] = id;
    x2[1] = id * id;
    reduce2<2>(x2, R, C);
    C.same_xcd = ((double)(C.NB * R.S * 64) * x2[1] == x2[0] * x2[0]) ? 1 : 0;
  }

  // schedule (:770-828): optimize(5) -> gate degenerate GMM edges -> optimize(5) -> gate reprojection
  // edges, robust kernels off -> optimize(40).  One rolled phase loop = one copy of the optimiser code.
  int it3 = 0, trials = 0, outer = 0;
  const int ns = kSpread ? 1 : mp.S;
  PROF_K(1);
#pragma unroll 1
  for (int phase = 0; phase < 3; ++phase) {
    it3 = optimize_fast(U, gm, D, mp, fw, P, gobn, gassoc, gnd, pc, phase < 2, phase < 2 ? 5 : 40, R, trials, C, An, pv.pfl + gbase);
    outer += it3 > 0 ? it3 : 0;
    PROF_K(2 + 2 * phase);
    if (phase == 2) break;
#pragma unroll 1
    for (int i = 0; i < ns; ++i) {
      const int l = mp.base + slot_off(mp, i), ll = mp.lbase + slot_off(mp, i);
      if (slot_absent(mp, i)) continue;
      if (l >= L) {
        if (kTwo) continue;  // (the second group's slots follow)
        break;
      }
      const int fl = fw_get(fw, i);
      if (phase == 0) {  // fresh error of the degenerate GMM edges (:773-786)
        if ((fl & (F_ASSOC | F_DEG)) == (F_ASSOC | F_DEG)) {
          const double p[3] = {D.sp[ll], D.sp[MCAP + ll], D.sp[2 * MCAP + ll]};
          const size_t ap = (size_t)gassoc[l];
          const double nd[4] = {gnd[ap * 4], gnd[ap * 4 + 1], gnd[ap * 4 + 2], gnd[ap * 4 + 3]};
          if (gmm_chi2_fast(U.lm, gm, nd, fl, -1, p) > k.str_thresh) fw_or(fw, i, F_LEVG);
        }
      } else {  // STALE chi2 of the reprojection edges, fresh depth test (:799-825)
        if (!(fl & F_EXISTS)) continue;
        const double z = P.R[6] * D.sp[ll] + P.R[7] * D.sp[MCAP + ll] + P.R[8] * D.sp[2 * MCAP + ll] + P.t[2];
        if (D.chir[ll] > ((fl & F_STEREO) ? 7.815 : 5.991) || !(z > 0.0)) fw_or(fw, i, F_LEVR);
        if (kFixed) {  // the fixed observers' edges: the same gate on their stale chi2, the depth in THEIR camera
#pragma unroll 1
          for (int kf = 0; kf < D.F; ++kf) {
            const int oc = D.gfoct[(size_t)kf * L + l];
            if (oc < 0) continue;
            const double* Rf = D.frt + kf * 12;
            const double zf = Rf[6] * D.sp[ll] + Rf[7] * D.sp[MCAP + ll] + Rf[8] * D.sp[2 * MCAP + ll] + Rf[11];
            if (D.gchif[(size_t)kf * L + l] > ((oc & 16) ? 7.815 : 5.991) || !(zf > 0.0)) fw_or(fw, i, F_OFFF << kf);
          }
        }
      }
      fw_activity(fw, i);
      if (kDynB) pv.pfl[gbase + l] = (pv.pfl[gbase + l] & 0xffff) | (fw_get(fw, i) << 16);
    }
    __syncthreads();
    PROF_K(3 + 2 * phase);
  }

  if (kSpread && C.NB > 1) {  // a frame whose workgroups lost each other writes nothing: the follow-up launch redoes it
    __syncthreads();
    C.failed |= *C.lds_fail;
    if (C.failed) return;
  }
  {  // ---- outputs ----
  const bool staged = kSpread && C.NB > 1;  // (results of a latency-shape launch go through the staging area)
  kargs_t* kb = ka;
  asm volatile("" : "+s"(kb));
  double* const pose_io = kb->pose_io;  // (these shadow the set-up's copies)
  double* const pts_io = kb->pts_io;
  int32_t* const assoc_all = kb->assoc_all;
  uint8_t* const dropped_all = kb->dropped_all;
  uint8_t* const erase_all = kb->erase_all;
  int32_t* const iters_out = kb->iters_out;
  int32_t* const trials_out = kb->trials_out;
  int32_t* const outer_out = kb->outer_out;
  int32_t* const edges_out = kb->edges_out;
  double* const stage_o = kb->stage;
  double* const st_pts = stage_o;
  double* const st_pose = stage_o ? stage_o + (size_t)B * L * 3 : nullptr;
  int32_t* const st_assoc = stage_o ? (int32_t*)(st_pose + (size_t)B * 8) : nullptr;
  double* const pts_out = staged ? st_pts : pts_io;
  int32_t* const assoc_out = staged ? st_assoc : assoc_all;
#pragma unroll 1
  for (int i = 0; i < ns; ++i) {  // outputs (:837-879, :898-922)
    const int l = mp.base + slot_off(mp, i), ll = mp.lbase + slot_off(mp, i);
    if (slot_absent(mp, i)) continue;
    if (l >= L) {
      if (kTwo) continue;
      break;
    }
    const size_t g = gbase + gperm[l];  // back to the caller's order
    const int fl = fw_get(fw, i);
    uint8_t dr = 0, er = 0;
    int a = gassoc[l];
    if (fl & F_EXISTS) {
      const double p[3] = {D.sp[ll], D.sp[MCAP + ll], D.sp[2 * MCAP + ll]};
      if ((fl & (F_ASSOC | F_DEG)) == (F_ASSOC | F_DEG)) {
        const double nd[4] = {gnd[(size_t)a * 4], gnd[(size_t)a * 4 + 1], gnd[(size_t)a * 4 + 2], gnd[(size_t)a * 4 + 3]};
        if (gmm_chi2_fast(U.lm, gm, nd, fl, -1, p) > k.str_thresh) dr = 1;
      }
      const double z = P.R[6] * p[0] + P.R[7] * p[1] + P.R[8] * p[2] + P.t[2];
      if (D.chir[ll] > ((fl & F_STEREO) ? 7.815 : 5.991) || !(z > 0.0)) er = 1;
#pragma unroll
      for (int j = 0; j < 3; ++j) pts_out[g * 3 + j] = p[j];
    }
    if (kFixed && fxv.ferase) {  // observations of the fixed key-frames the reference would erase (:855-879): stale chi2, fresh depth
#pragma unroll 1
      for (int kf = 0; kf < D.F; ++kf) {
        uint8_t fe = 0;
        const int oc = D.gfoct[(size_t)kf * L + l];
        if ((fl & F_EXISTS) && oc >= 0) {
          const double* Rf = D.frt + kf * 12;
          const double zf = Rf[6] * D.sp[ll] + Rf[7] * D.sp[MCAP + ll] + Rf[8] * D.sp[2 * MCAP + ll] + Rf[11];
          if (D.gchif[(size_t)kf * L + l] > ((oc & 16) ? 7.815 : 5.991) || !(zf > 0.0)) fe = 1;
        }
        fxv.ferase[g * D.F + kf] = fe;
      }
    }
    if (dropped_all) dropped_all[g] = dr;
    if (erase_all) erase_all[g] = er;
    if (!dropped_all && dr) a = -1;
    assoc_out[g] = a;
  }
  if (tid == 0 && C.pb == 0) {
    SE3 T;
    T.r = qfromR(P.R);
    T.t[0] = P.t[0];
    T.t[1] = P.t[1];
    T.t[2] = P.t[2];
    normalize_rotation(T);
    if (kPrior && An.pose_fixed) {  // a fixed vertex keeps the caller's bits
      if (staged) {
#pragma unroll
        for (int r = 0; r < 7; ++r) st_pose[(size_t)f * 8 + r] = pose_io[(size_t)f * 7 + r];
      }
    } else {
      se3_store(T, staged ? st_pose + (size_t)f * 8 : pose_io + (size_t)f * 7);
    }
    PROF_K(8);
#ifdef GL_BA_PROF
    if (f == 1024) g_prof_k[9] = (unsigned long long)trials;
#endif
    if (iters_out) iters_out[f] = it3;
    if (trials_out) trials_out[f] = trials;
    if (outer_out) outer_out[f] = outer;
    if (edges_out) {  // gl_ctx_set_edge_stats_buffer: level-0 reprojection edges x trials, x outer iterations
      const int* est = (const int*)(R.tot + 62);
      edges_out[2 * f] = est[0];
      edges_out[2 * f + 1] = est[1];
    }
#ifdef GL_BA_TRACE
    if (f == 0)
      for (int i = 0; i < 10 * 128 && i < L * 3; ++i) pts_io[i] = g_trace[i];
#endif
#ifdef GL_BA_PROF
    if (f == 0) {  // debug build only: phase cycles instead of pose 0, per-wave markers instead of the points of frame 0
      // (a latency-shape launch stages its results: the follow-up kernel copies them over the caller's buffers)
      double* const pr = staged ? st_pose : pose_io;
      for (int i = 0; i < 7; ++i) pr[i] = (double)g_prof[i == 5 ? 7 : i];  // slot 5 reports the rejections
    }
    if (f != 0 && f != 1024 && !kSpread) {  // every other frame: when and where its workgroup ran, in place of its pose (tools/prof_ba.py: CU timelines)
      pose_io[(size_t)f * 7] = (double)prof_wg_t0;
      pose_io[(size_t)f * 7 + 1] = (double)wall_clock64();
      pose_io[(size_t)f * 7 + 2] = (double)((__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) & 0xffffff) | ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) << 24));
      pose_io[(size_t)f * 7 + 3] = (double)trials;
    }
    if (f == 1024) {  // (the stations of THIS frame go out in place of its pose and first points)
      for (int i = 0; i < 7; ++i) pose_io[(size_t)f * 7 + i] = (double)g_prof_k[i];
      for (int i = 7; i < 13; ++i) pts_io[(size_t)f * L * 3 + (i - 7)] = (double)g_prof_k[i];
    }
    if (f == 0 && !staged) {
      for (int i = 0; i < 64 * 8 * 4 && i < L * 3; ++i) pts_io[i] = (double)g_prof_w[i];
      for (int i = 0; i < 16 * 8 * 2 * 5 && 2048 + i < L * 3; ++i) pts_io[2048 + i] = (double)g_prof_s[i];
      for (int i = 0; i < 10 * 8 * 2 * 4 * 3 && 2048 + 1280 + i < L * 3; ++i) pts_io[2048 + 1280 + i] = (double)g_prof_q[i];
    }
#endif
  }
  if (kSpread && C.ctl) {  // this workgroup's share of the frame is in place: count it done (the follow-up kernel wants all NB)
    __syncthreads();  // (the counter is read by the NEXT kernel on the stream: the kernel boundary publishes the staged results)
    if (tid == 0) __hip_atomic_fetch_add(C.ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }
}

// DENSE: <<<min(B, frames the device holds at once), 64 G>>> PERSISTENT workgroups that draw their frames from a counter (frame_ctr,
// zeroed by k_ba1_prep): a workgroup of the largest class owns a whole CU, and between two workgroups of a plain launch that CU sat
// idle for 16 us (median; 48 us mean: workgroup timelines of profiles/r5_prof_ba_stations.txt) - 6 % of the launch.  Which workgroup
// takes a frame shows nowhere in its result.  SPREAD (and frame_ctr == nullptr): the plain launch, one block per (frame, group).
#ifdef GL_BAF_THREADS
__global__ __launch_bounds__(GL_BAF_THREADS, GL_BAF_THREADS / 256) void k_ba1_fast(BafKArgs A) {
#elif GL_BAF_GPW == 2
__global__ __launch_bounds__(64 * NWV, 2) void k_ba1_fast(BafKArgs A) {
#else
__global__ __launch_bounds__(kSpread ? TSP : 512, 2) void k_ba1_fast(BafKArgs A) {
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_frame;
  const bool persistent = !kSpread && A.frame_ctr != nullptr;
  bool first = true;
  // The arguments are read from the kernel-argument segment AGAIN for every frame, through a pointer the compiler cannot see through:
  // hoisted out of the frame loop (they are loop-invariant loads) the ~40 of them stayed live across the whole body and pushed the
  // pass loops into 37 spilled VGPRs (+ 4 % on the refine); re-read, the body has the register allocation of the plain kernel.
  kargs_t* ka = (kargs_t*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll 1
  for (;;) {  // (ONE call site of the body: the plain launch makes a single trip with its block index)
    asm volatile("" : "+s"(ka));
    unsigned fr = blockIdx.x;
    if (persistent) {
      if (threadIdx.x == 0) s_frame = atomicAdd(ka->frame_ctr, 1);
      __syncthreads();
      fr = (unsigned)s_frame;
      if ((int)fr >= ka->B) break;
    } else if (!first) {
      break;
    }
    first = false;
    ba1_fast_frame(smem, fr, ka);
    if (persistent) __syncthreads();  // (the frame's last readers of the LDS state and of s_frame are through)
  }
}

}  // namespace GL_BAF_NS
}  // namespace
