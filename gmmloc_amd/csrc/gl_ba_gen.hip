// General local bundle adjustment: Localization::jointOptimization
// (localization_opt.cpp:456-925) with P free poses, F fixed poses, L marginalised points
// observed by any number of poses (mono / stereo, Huber), one GMM edge per point, optional
// pose prior.  One persistent workgroup per problem; no atomics, deterministic:
//   P1  one thread per point: linearise every active observation in its camera frame
//       (q, A = w Jpi^T Jpi, a = w Jpi^T e stored per observation), gather the world-frame
//       point block D = sum R^T A R + Hg + lambda I, D^-1 by cofactors, u = D^-1 b.
//   P2  one WAVE per reduced-camera block (j1, j2): lanes stride over pose j1's observation
//       list (pose-major CSR built once on the device), find the partner observation of
//       the same point in pose j2 through a precomputed match table, and accumulate
//       G1^T [A1 R1 D^-1 R2^T A2] G2 in registers; a wave reduce-scatter (gld::
//       wave_reduce_scatter32) then writes the 6x6 block -- the Schur complement
//       S = Hpp - sum W D^-1 W^T and g = bp - sum W D^-1 b without a single atomic.
//   solve  in-place right-looking LDL^T of the 6P x 6P system by the whole workgroup.
//   P3  back-substitution per point, trial state, new errors, rho test.
// Control flow and gating as in gl_ba.hip (g2o Levenberg, 5 / 5 / 40 schedule).
//
// A problem is worked on by NB co-resident workgroups (cooperative launch): the point / observation
// loops and the (j1, j2) blocks are strided over all of them, phases are separated by a
// per-problem barrier (atomic counter + device-scope fences), sums go through per-workgroup
// partials that every workgroup adds up in the same order, so all workgroups take identical
// branches and the result does not depend on NB.  NB = 1 (large batches) needs no global barrier.
#include <algorithm>
#include <cstdlib>

#include "gl_ba_common.hpp"

using namespace gld;
using namespace glba;

namespace {

#if defined(GL_BAGEN_PROF) || defined(GL_PIPE_PROF)  // phase cycles of workgroup 0 (tools/prof_bagen.py)
__device__ unsigned long long g_gprof[12];
#define GP_T(v) const long long v = clock64()
#define GP_ADD(slot, a, b) if (blockIdx.x == 0 && threadIdx.x == 0) g_gprof[slot] += (unsigned long long)((b) - (a))
#else
#define GP_T(v)
#define GP_ADD(slot, a, b)
#endif

struct GenP {
  int P, F, L, nobs;
  double* poses;
  const uint8_t* prior;
  double* pts;
  const int32_t* assoc;
  const int32_t* optr;
  const int32_t* opose;
  const double* ouvr;
  const int32_t* ooct;
  // scratch
  double* Rt;      // (P+F) x 12 current R, t
  double* RtN;     // P x 12 trial
  double* qN;      // P x 7 trial poses
  double* pinv;    // P x 7 prior inverse measurement
  double* pn;      // L x 3 trial points
  double* lin;     // nobs x 12: q3 A6 a3
  double* ptw;     // L x 12: Dinv6 u3 bl3
  double* pth;     // L x 6: the points' undamped blocks of the last full point pass (point_relambda; the persistent kernel's carve)
  double* chi_o;   // nobs
  double* S;       // n x n
  double* gv;      // n
  double* bp;      // n
  double* dxv;     // n
  double* pchi;    // P prior chi2 at the current state
  double* pchi2;   // P prior chi2 at the trial state
  double* prH;     // P x 36 + P x 6: prior information / rhs at the current state (per outer iteration)
  double* prb;
  int32_t* opoint; // nobs
  int32_t* pl_ptr; // P + 1
  int32_t* pl_obs; // nobs: pose-major list of the free poses' observations
  int32_t* pl_pos; // nobs: position of an observation in that list
  int32_t* pl_pt;  // nobs: point of the e-th entry of the list
  int32_t* plm;    // nobs x P: partner of the e-th entry in pose j2 (itself for its own pose); -1: none, or an edge at level 1
  uint8_t* lev_o;  // nobs
  uint8_t* lev_g;  // L
  uint8_t* pfree;  // P + F
  uint8_t* pact;   // P
  uint8_t* lact;   // L  point active this optimize()
  // multi-workgroup execution
  int ld;              // row stride of S (doubles)
  int NB, pb;          // workgroups per problem, index of this one
  int toggle;          // which partial-sum buffer the next reduction uses
  unsigned* bar;       // 64 arrival words of the problem's barrier
  unsigned epoch;      // barriers passed
  double* part;        // NB x 4 partial sums
  int* flagg;          // [0] solve status, [1] the stop word as workgroup 0 last saw it
  // setForceStopFlag (localization_opt.cpp:541-542): optional stop word, polled once per Levenberg trial by workgroup 0 and
  // acted on where g2o tests terminate(): before an outer iteration.  > 0: stop; < 0: a budget of -value outer iterations
  const int32_t* stop;
  int stop_seen, done_iters;
  int trials;  // Levenberg trials so far (linearise + reduce + solve + evaluate): the unit of the kernel's algorithmic work
};
GL_DEV bool stop_now(const GenP& G) { return G.stop_seen > 0 || (G.stop_seen < 0 && G.done_iters >= -G.stop_seen); }
GL_DEV int stop_word_load(const int32_t* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// k_ba_gen (the persistent kernel) keeps the ~40 addresses of a problem's state in a TABLE at the end of the problem's scratch area
// (written by the kernel's prologue) and takes them from there again after every problem-wide barrier, as scalar loads through a
// constant-address-space pointer the compiler cannot see through: computed once at the top they were live across all 85 000
// instructions of the kernel - 1 310 spilled SGPRs in 21 VGPRs of spill lanes, which in turn pushed 36 VGPRs to scratch.  An address
// is now live from the barrier in front of its phase to its last use in that phase: 0 spilled VGPRs, same bits.  (The mutable words -
// toggle, epoch, stop_seen, done_iters, trials -, the sizes and S - which may be a generic pointer into LDS: taken through memory it
// faulted (memory aperture violation on the first window whose system fits LDS) - stay in registers.  A first version kept the table in LDS: ds_read +
// v_readfirstlane per address, 2 % slower than the spills it removed.  Re-reading the camera / map constants BaK / GmmDev from the
// kernel-argument segment the same way changed nothing: what is left - 1 024 spilled SGPRs - are loop-carried scalars of the body, in
// v_writelane slots, not in memory.)
typedef const GenP __attribute__((address_space(4))) k_genp;
constexpr size_t GEN_TABLE_BYTES = 512;
static_assert(sizeof(GenP) <= GEN_TABLE_BYTES, "the address table of k_ba_gen lives in the last 512 bytes of a problem's scratch area");
GL_DEV void genp_fresh(GenP& G, k_genp* s) {
  asm volatile("" : "+s"(s));
#define GF(x) G.x = s->x
  GF(poses); GF(prior); GF(pts); GF(assoc); GF(optr); GF(opose); GF(ouvr); GF(ooct);
  GF(Rt); GF(RtN); GF(qN); GF(pinv); GF(pn); GF(lin); GF(ptw); GF(pth); GF(chi_o); GF(gv); GF(bp); GF(dxv); GF(pchi); GF(pchi2);
  GF(prH); GF(prb); GF(opoint); GF(pl_ptr); GF(pl_obs); GF(pl_pos); GF(pl_pt); GF(plm); GF(lev_o); GF(lev_g); GF(pfree); GF(pact); GF(lact);
  GF(bar); GF(part); GF(flagg); GF(stop);
#undef GF
}
#define GFRESH() genp_fresh(G, sG)

// barrier over the NB <= 64 workgroups of one problem (all co-resident: cooperative launch).  One flag word per
// workgroup (zeroed by the host): a workgroup announces its k-th arrival by storing k into its own word and the lanes
// of its first wave read everybody's word until all of them say k - plain stores to different words and one 256-byte
// read per poll, where NB read-modify-writes of ONE counter would queue up behind each other at the memory side.
GL_DEV void prob_sync(GenP& G) {
  GP_T(b0);
  __syncthreads();
  GP_T(b1);
  GP_ADD(7, b0, b1);  // waiting for the own workgroup
  if (G.NB > 1) {
    G.epoch += 1u;
    if (threadIdx.x < 64) {
      if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_store(&G.bar[G.pb], G.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int w = min((int)threadIdx.x, G.NB - 1);
      while (!__all((int)(__hip_atomic_load(&G.bar[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - G.epoch) >= 0))
        __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
    GP_T(b2);
    GP_ADD(10, b1, b2);  // (debug build: replaces the panel slot) the cross-workgroup part
  }
}
// problem-wide sum of NV (<= 4) per-thread values: workgroup reduction, partials to global memory,
// every workgroup adds the NB partials in index order (identical result everywhere)
template <int NV>
GL_DEV void prob_reduce(GenP& G, double* acc, double* red) {
  block_reduce<NV, NW_BA>(acc, red);
  if (G.NB == 1) return;
  // two partial buffers used alternately: the barrier of the NEXT reduction separates the readers of a
  // buffer from its next writers, so one barrier per reduction is enough
  double* part = G.part + (G.toggle ? 64 * 4 : 0);
  G.toggle ^= 1;
  if (threadIdx.x == 0)
    for (int i = 0; i < NV; ++i) part[G.pb * 4 + i] = acc[i];
  prob_sync(G);
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
    for (int b = 0; b < G.NB; ++b) s += part[b * 4 + i];
    acc[i] = s;
  }
}
GL_DEV double prob_max(GenP& G, double v, double* red) {
  v = block_max(v, red);
  if (G.NB == 1) return v;
  double* part = G.part + (G.toggle ? 64 * 4 : 0);
  G.toggle ^= 1;
  if (threadIdx.x == 0) part[G.pb * 4] = v;
  prob_sync(G);
  double m = part[0];
  for (int b = 1; b < G.NB; ++b) m = fmax(m, part[b * 4]);
  return m;
}
#define GSTART (G.pb * T_BA + (int)threadIdx.x)
#define GSTRIDE (G.NB * T_BA)

GL_DEV void load_Rt(const double* Rt, double* R, double* t) {
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Rt[i];
  t[0] = Rt[9];
  t[1] = Rt[10];
  t[2] = Rt[11];
}
// the same through scalar loads (wave-uniform address, data no kernel in flight writes): {R, t} land in SGPRs
typedef const double __attribute__((address_space(4))) cdouble_k;
GL_DEV void load_Rt_k(const cdouble_k* Rt, double* R, double* t) {
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Rt[i];
  t[0] = Rt[9];
  t[1] = Rt[10];
  t[2] = Rt[11];
}
GL_DEV void store_pose_Rt(const SE3& T, double* Rt) {
  double R[9];
  qtoR(T.r, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) Rt[i] = R[i];
  Rt[9] = T.t[0];
  Rt[10] = T.t[1];
  Rt[11] = T.t[2];
}

// reprojection linearisation in the camera frame: q, A (sym6), a; returns chi2, rho0
GL_DEV void lin_obs(const BaK& k, const double* R, const double* t, const double* p, const double* ob, int oc,
                    bool robust, double* q, double* A, double* a, double& chi2, double& rho0) {
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = R[i * 3] * p[0] + R[i * 3 + 1] * p[1] + R[i * 3 + 2] * p[2] + t[i];
  const bool stereo = !(ob[2] < 0);
  const double s = k.s2inv[oc];
  double e[3], iz;
  chi2 = reproj_err(k, q, ob, stereo, s, e, iz);
  double rho1 = 1.0;
  rho0 = chi2;
  if (robust) huber(chi2, stereo ? k.delta_stereo : k.delta_mono, rho0, rho1);
  const double w = rho1 * s;
  const double iz2 = iz * iz;
  const double al = k.fx * iz, ga = k.fy * iz;
  const double b0 = -k.fx * q[0] * iz2, b1 = -k.fy * q[1] * iz2;
  const double b2 = b0 + k.bf * iz2;
  const double sb = stereo ? 1.0 : 0.0;
  A[0] = w * (al * al + sb * al * al);
  A[1] = 0.0;
  A[2] = w * (al * b0 + sb * al * b2);
  A[3] = w * ga * ga;
  A[4] = w * ga * b1;
  A[5] = w * (b0 * b0 + b1 * b1 + sb * b2 * b2);
  a[0] = w * al * (e[0] + sb * e[2]);
  a[1] = w * ga * e[1];
  a[2] = w * (b0 * e[0] + b1 * e[1] + sb * b2 * e[2]);
}

GL_DEV void sym_to_full(const double* S, double* F) {
  F[0] = S[0];
  F[1] = S[1];
  F[2] = S[2];
  F[3] = S[1];
  F[4] = S[3];
  F[5] = S[4];
  F[6] = S[2];
  F[7] = S[4];
  F[8] = S[5];
}
GL_DEV void mm3t(const double* A, const double* B, double* C) {  // C = A * B^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
GL_DEV void mmt3(const double* A, const double* B, double* C) {  // C = A^T * B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// blk (6x6 row-major) = G1^T M G2 with G = [-[q]x | I]  ->  [[-Q1 M Q2, Q1 M], [-M Q2, M]]
GL_DEV void gmg(const double* q1, const double* M, const double* q2, double* blk) {
  double Q1M[9], MQ2[9], Q1MQ2[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // column j of Q1 M = q1 x M[:, j]
    const double col[3] = {M[j], M[3 + j], M[6 + j]};
    double r[3];
    cross(q1, col, r);
    Q1M[j] = r[0];
    Q1M[3 + j] = r[1];
    Q1M[6 + j] = r[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // row i of M Q2 = (M[i,:]) Q2 = -(q2 x M[i,:]) ... (v^T Q) = (Q^T v)^T = -(q x v)^T
    double r[3];
    cross(q2, &M[i * 3], r);
    MQ2[i * 3] = -r[0];
    MQ2[i * 3 + 1] = -r[1];
    MQ2[i * 3 + 2] = -r[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double r[3];
    cross(q2, &Q1M[i * 3], r);
    Q1MQ2[i * 3] = -r[0];
    Q1MQ2[i * 3 + 1] = -r[1];
    Q1MQ2[i * 3 + 2] = -r[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      blk[i * 6 + j] = -Q1MQ2[i * 3 + j];
      blk[i * 6 + 3 + j] = Q1M[i * 3 + j];
      blk[(i + 3) * 6 + j] = -MQ2[i * 3 + j];
      blk[(i + 3) * 6 + 3 + j] = M[i * 3 + j];
    }
}

// ---- P1 -------------------------------------------------------------------------------------
// returns (per-thread partial) robust chi2; mdiag = max landmark diagonal (world frame)
// sum over the LPP adjacent lanes that share one point (every lane gets it)
// Round-trip economy of the point passes.  A thread walks the observations of its point, and every record it
// needs (flag, pose index, key-point, then the pose) used to be fetched where the code first tests it - four to six
// dependent global-memory latencies per observation, on a workgroup with one wave per SIMD and nothing to switch
// to.  Now the record of an observation {flag, pose, octave, key-point} is fetched as one round BEFORE the flag is
// tested, one observation ahead of its use, and the pose-indexed data as a second round; `pin` keeps the compiler from
// sinking those loads back behind the tests.  The arithmetic is untouched.
GL_DEV void pin(double x) { asm volatile("" ::"v"(x)); }
GL_DEV void pin(int x) { asm volatile("" ::"v"(x)); }
struct ObsRec {
  int lv, j, oc;
  double ob[3];
};
GL_DEV void fetch_obs(const GenP& G, int o, int o_end, ObsRec& r) {
  const int oc = min(o, o_end - 1);  // (one past the end: re-reads the last record)
  r.lv = G.lev_o[oc];
  r.j = G.opose[oc];
  r.oc = G.ooct[oc];
#pragma unroll
  for (int i = 0; i < 3; ++i) r.ob[i] = G.ouvr[(size_t)oc * 3 + i];
}
GL_DEV void pin_obs(const ObsRec& r) {
  pin(r.lv);
  pin(r.j);
  pin(r.oc);
  pin(r.ob[0]);
  pin(r.ob[1]);
  pin(r.ob[2]);
}

template <int LPP>
GL_DEV double group_sum(double v) {
  if (LPP >= 2) v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
  if (LPP >= 4) v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
  if (LPP >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the eight (every lane of a quad holds its sum)
  return v;
}
// LPP adjacent lanes per point (1, 2 or 4: as many as the problem's workgroups have threads for): the
// observations of the point are dealt round them, the 3x3 point block and its rhs are summed over the group
// (KEEP: the undamped point block goes to pth - L x 6 - as well: what point_relambda needs)
template <int LPP, bool KEEP = false>
GL_DEV double pass_points(const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, double& mdiag, double* pth = nullptr) {
  double chi = 0.0;
  const int sub = (int)threadIdx.x & (LPP - 1);
  for (int l = GSTART / LPP; l < G.L; l += GSTRIDE / LPP) {
    if (!G.lact[l]) continue;  // the same for the LPP lanes of the group
    const double p[3] = {G.pts[(size_t)l * 3], G.pts[(size_t)l * 3 + 1], G.pts[(size_t)l * 3 + 2]};
    double H[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    const int o_end = G.optr[l + 1];
    ObsRec cur;
    if (G.optr[l] + sub < o_end) fetch_obs(G, G.optr[l] + sub, o_end, cur);
    for (int o = G.optr[l] + sub; o < o_end; o += LPP) {
      ObsRec nxt;
      fetch_obs(G, o + LPP, o_end, nxt);
      const ObsRec rec = cur;
      double R[9], t[3], q[3], A[6], a[3], c2, r0;
      load_Rt(G.Rt + (size_t)rec.j * 12, R, t);
      pin_obs(nxt);
      cur = nxt;
      if (rec.lv) continue;
      lin_obs(k, R, t, p, rec.ob, rec.oc, robust, q, A, a, c2, r0);
      G.chi_o[o] = c2;
      chi += r0;
      double* lo = G.lin + (size_t)o * 12;
#pragma unroll
      for (int i = 0; i < 3; ++i) lo[i] = q[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) lo[3 + i] = A[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) lo[9 + i] = a[i];
      double Af[9], AR[9], RAR[9];
      sym_to_full(A, Af);
      mm3(Af, R, AR);
      mmt3(R, AR, RAR);
      H[0] += RAR[0];
      H[1] += RAR[1];
      H[2] += RAR[2];
      H[3] += RAR[4];
      H[4] += RAR[5];
      H[5] += RAR[8];
#pragma unroll
      for (int i = 0; i < 3; ++i) bl[i] += R[i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2];
    }
    if (sub == 0 && G.assoc[l] >= 0 && !G.lev_g[l]) {
      GmmRef g;
      load_gmm(G.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
      const double d[3] = {p[0] - g.mu[0], p[1] - g.mu[1], p[2] - g.mu[2]};
      if (g.deg) {
        const double eg = g.n[0] * d[0] + g.n[1] * d[1] + g.n[2] * d[2];
        const double lm = k.ba_lambda2;
        chi += eg * (lm * eg);
        H[0] += lm * g.n[0] * g.n[0];
        H[1] += lm * g.n[0] * g.n[1];
        H[2] += lm * g.n[0] * g.n[2];
        H[3] += lm * g.n[1] * g.n[1];
        H[4] += lm * g.n[1] * g.n[2];
        H[5] += lm * g.n[2] * g.n[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) bl[i] -= lm * eg * g.n[i];
      } else {
        double e[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = g.L[i] * d[0] + g.L[3 + i] * d[1] + g.L[6 + i] * d[2];
        chi += e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        double Hg[9];
        mm3t(g.L, g.L, Hg);
        H[0] += Hg[0];
        H[1] += Hg[1];
        H[2] += Hg[2];
        H[3] += Hg[4];
        H[4] += Hg[5];
        H[5] += Hg[8];
#pragma unroll
        for (int i = 0; i < 3; ++i) bl[i] -= g.L[i * 3] * e[0] + g.L[i * 3 + 1] * e[1] + g.L[i * 3 + 2] * e[2];
      }
    }
    if (LPP > 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) H[i] = group_sum<LPP>(H[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) bl[i] = group_sum<LPP>(bl[i]);
    }
    mdiag = fmax(mdiag, fmax(fabs(H[0]), fmax(fabs(H[3]), fabs(H[5]))));
    double D[6] = {H[0] + lambda, H[1], H[2], H[3] + lambda, H[4], H[5] + lambda}, Dinv[6], u[3];
    sym3_inv(D, Dinv);
    sym3_mul_vec(Dinv, bl, u);
    if (KEEP && sub == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pth[(size_t)l * 6 + i] = H[i];
    }
    if (sub == 0) {
      double* pw = G.ptw + (size_t)l * 12;
#pragma unroll
      for (int i = 0; i < 6; ++i) pw[i] = Dinv[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        pw[6 + i] = u[i];
        pw[9 + i] = bl[i];
      }
    }
  }
  return chi;
}
GL_DEV double pass_points_lpp(int lpp, const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, double& mdiag) {
  if (lpp == 4) return pass_points<4, true>(k, gm, G, robust, lambda, mdiag, G.pth);
  if (lpp == 2) return pass_points<2, true>(k, gm, G, robust, lambda, mdiag, G.pth);
  return pass_points<1, true>(k, gm, G, robust, lambda, mdiag, G.pth);
}

// (the kernels of the pipelined shape: 4 lanes per point, 8 where a point has six or more observations on average - half
// the dependent loads per lane; instances of their own, so that the persistent kernel's code is what it was)
GL_DEV double pass_points_pipe(int lpp, const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, double& mdiag, double* pth) {
  if (lpp == 8) return pass_points<8, true>(k, gm, G, robust, lambda, mdiag, pth);
  return pass_points<4, true>(k, gm, G, robust, lambda, mdiag, pth);
}
// The point pass of a trial whose state is the one the previous point pass linearised at (a rejected trial's successor, the first
// trial after the lambda-init pass): only lambda has changed, and of everything the pass writes only {D^-1, u} = {(H + lambda I)^-1,
// D^-1 b} of the points depends on it - the observation records, b, chi2 and the largest diagonal entry are what they were.  The
// same expressions on the same values as the tail of pass_points: the same bits.  A thread per point.
GL_DEV void point_relambda(const GenP& G, double lambda, const double* pth) {
  for (int l = GSTART; l < G.L; l += GSTRIDE) {
    if (!G.lact[l]) continue;
    double* pw = G.ptw + (size_t)l * 12;
    double H[6], bl[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) H[i] = pth[(size_t)l * 6 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bl[i] = pw[9 + i];
    double D[6] = {H[0] + lambda, H[1], H[2], H[3] + lambda, H[4], H[5] + lambda}, Dinv[6], u[3];
    sym3_inv(D, Dinv);
    sym3_mul_vec(Dinv, bl, u);
#pragma unroll
    for (int i = 0; i < 6; ++i) pw[i] = Dinv[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pw[6 + i] = u[i];
  }
}

// ---- P3: back-substitution of the points, trial points, their chi2 (LPP lanes per point like P1) ------------
template <int LPP>
GL_DEV void pass_trial(const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, int P, double* acc) {
  const int sub = (int)threadIdx.x & (LPP - 1);
  for (int l = GSTART / LPP; l < G.L; l += GSTRIDE / LPP) {
    if (!G.lact[l]) continue;
    const double* pw = G.ptw + (size_t)l * 12;
    double rhs[3] = {0.0, 0.0, 0.0};
    if (sub == 0) {
      rhs[0] = pw[9];
      rhs[1] = pw[10];
      rhs[2] = pw[11];
    }
    const int o_beg = G.optr[l] + sub, o_end = G.optr[l + 1];
    for (int o = o_beg; o < o_end; o += LPP) {  // (two rounds of loads, then the tests: see fetch_obs)
      const int lv = G.lev_o[o], j = G.opose[o], jf = min(j, P - 1);
      double lo[9], dx[6], R[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) lo[i] = G.lin[(size_t)o * 12 + i];
      const int pa = G.pact[jf];
#pragma unroll
      for (int i = 0; i < 6; ++i) dx[i] = G.dxv[6 * jf + i];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = G.Rt[(size_t)j * 12 + i];
      pin(pa);
#pragma unroll
      for (int i = 0; i < 9; ++i) pin(lo[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) pin(dx[i]);
#pragma unroll
      for (int i = 0; i < 9; ++i) pin(R[i]);
      if (lv || j >= P || !pa) continue;
      double gd[3], Ag[3];
      cross(dx, lo, gd);
      gd[0] += dx[3];
      gd[1] += dx[4];
      gd[2] += dx[5];
      sym3_mul_vec(lo + 3, gd, Ag);
#pragma unroll
      for (int i = 0; i < 3; ++i) rhs[i] -= R[i] * Ag[0] + R[3 + i] * Ag[1] + R[6 + i] * Ag[2];
    }
    if (LPP > 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i) rhs[i] = group_sum<LPP>(rhs[i]);
    }
    double dl[3];
    sym3_mul_vec(pw, rhs, dl);
    const double pn[3] = {G.pts[(size_t)l * 3] + dl[0], G.pts[(size_t)l * 3 + 1] + dl[1], G.pts[(size_t)l * 3 + 2] + dl[2]};
    if (sub == 0) {
      acc[0] += dl[0] * (lambda * dl[0] + pw[9]) + dl[1] * (lambda * dl[1] + pw[10]) + dl[2] * (lambda * dl[2] + pw[11]);
#pragma unroll
      for (int i = 0; i < 3; ++i) G.pn[(size_t)l * 3 + i] = pn[i];
    }
    ObsRec cur;
    if (o_beg < o_end) fetch_obs(G, o_beg, o_end, cur);
    for (int o = o_beg; o < o_end; o += LPP) {
      ObsRec nxt;
      fetch_obs(G, o + LPP, o_end, nxt);
      const ObsRec rec = cur;
      const int j = rec.j;
      const double* Rt = (j < P) ? G.RtN + (size_t)j * 12 : G.Rt + (size_t)j * 12;
      double Rv[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) Rv[i] = Rt[i];
      pin_obs(nxt);
      cur = nxt;
#pragma unroll
      for (int i = 0; i < 12; ++i) pin(Rv[i]);
      if (rec.lv) continue;
      double q[3], e[3], iz;
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] = Rv[i * 3] * pn[0] + Rv[i * 3 + 1] * pn[1] + Rv[i * 3 + 2] * pn[2] + Rv[9 + i];
      const double* ob = rec.ob;
      const bool stereo = !(ob[2] < 0);
      const double c2 = reproj_err(k, q, ob, stereo, k.s2inv[rec.oc], e, iz);
      G.chi_o[o] = c2;
      double r0 = c2, r1;
      if (robust) huber(c2, stereo ? k.delta_stereo : k.delta_mono, r0, r1);
      acc[1] += r0;
    }
    if (sub == 0 && G.assoc[l] >= 0 && !G.lev_g[l]) {
      GmmRef g;
      load_gmm(G.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
      acc[1] += gmm_chi2(k, g, pn);
    }
  }
}
GL_DEV void pass_trial_lpp(int lpp, const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, int P, double* acc) {
  if (lpp == 4) return pass_trial<4>(k, gm, G, robust, lambda, P, acc);
  if (lpp == 2) return pass_trial<2>(k, gm, G, robust, lambda, P, acc);
  return pass_trial<1>(k, gm, G, robust, lambda, P, acc);
}

GL_DEV void pass_trial_pipe(int lpp, const BaK& k, const GmmDev& gm, const GenP& G, bool robust, double lambda, int P, double* acc) {
  if (lpp == 8) return pass_trial<8>(k, gm, G, robust, lambda, P, acc);
  return pass_trial<4>(k, gm, G, robust, lambda, P, acc);
}

// One entry of the Schur pass - the observation pair (o1 in pose j1, o2 in pose j2) of one point: its 6 x 6 block of the reduced camera
// system, - G1^T (A1 R1 D^-1 R2^T A2) G2 off the diagonal, G1^T A1 G1 minus that on it, and on the diagonal the six + six entries of
// g and b_p, added to the lane's 36 + 12 sums (v1[0..31], v2[0..15]: the block row-major, then g, then b_p).  l1 / l2: the two
// observations' records {q, A (sym6), a}, pw: the point's {D^-1 (sym6), u}.  Shared by the persistent kernel and kp_schur.
GL_DEV void schur_entry(const double* l1, const double* l2, const double* pw, const double* R1, const double* R2, bool schur, bool diag,
                        double* v1, double* v2) {
  double A1[9], A2[9], Dv[9];
  sym_to_full(l1 + 3, A1);
  sym_to_full(l2 + 3, A2);
  sym_to_full(pw, Dv);
  // The 6 x 6 block G1^T M G2 = [[-Q1 M Q2, Q1 M], [-M Q2, M]] (gmg) is formed and added QUADRANT BY QUADRANT - with the whole block
  // (and, on the diagonal, the whole of G1^T A1 G1 beside it) live at once the 48 accumulators did not fit the 256 registers and
  // were spilled around every entry (247 MB of scratch writes per 64-window launch).  Same expressions, same values.
  double M[9];
  if (schur) {  // M = A1 R1 D^-1 R2^T A2
    double X[9], Y[9], Z[9];
    mm3(A1, R1, X);
    mm3(X, Dv, Y);
    mm3t(Y, R2, Z);
    mm3(Z, A2, M);
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = 0.0;
  }
    double Q1M[9], Q1A[9];  // column j of Q1 M = q1 x M[:, j]; the same of A1 for the diagonal block's G1^T A1 G1
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double col[3] = {M[j], M[3 + j], M[6 + j]};
    double r[3];
    cross(l1, col, r);
    Q1M[j] = r[0];
    Q1M[3 + j] = r[1];
    Q1M[6 + j] = r[2];
  }
  if (!schur) {  // (gmg of the zero block gave zeros, not the -0.0 a cross product of zeros may)
#pragma unroll
    for (int i = 0; i < 9; ++i) Q1M[i] = 0.0;
  }
  if (diag) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double col[3] = {A1[j], A1[3 + j], A1[6 + j]};
      double r[3];
      cross(l1, col, r);
      Q1A[j] = r[0];
      Q1A[3 + j] = r[1];
      Q1A[6 + j] = r[2];
    }
  }
  // quadrant (R0, C0) of the block: bq[] its entries from M, hq[] the same of A1 (diagonal block); v += hq - bq  or  v += -bq
  auto fold = [&](const int R0, const int C0, const double* bq, const double* hq) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int idx = (R0 + i) * 6 + C0 + j;
        const double t = diag ? hq[i * 3 + j] - bq[i * 3 + j] : -bq[i * 3 + j];
        if (idx < 32) v1[idx] += t;
        else v2[idx - 32] += t;
      }
  };
  {  // bottom right: M;  top right: Q1 M
    fold(3, 3, M, A1);
    fold(0, 3, Q1M, Q1A);
  }
  {  // bottom left: -M Q2, row i = q2 x M[i, :] after the two sign changes of gmg;  top left: -Q1 M Q2 likewise from Q1 M
    double bl[9], hl[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double r[3];
      cross(l2, &M[i * 3], r);
      bl[i * 3] = r[0];
      bl[i * 3 + 1] = r[1];
      bl[i * 3 + 2] = r[2];
      if (diag) {
        cross(l1, &A1[i * 3], r);
        hl[i * 3] = r[0];
        hl[i * 3 + 1] = r[1];
        hl[i * 3 + 2] = r[2];
      }
    }
    if (!schur) {
#pragma unroll
      for (int i = 0; i < 9; ++i) bl[i] = 0.0;
    }
    fold(3, 0, bl, hl);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double r[3];
      cross(l2, &Q1M[i * 3], r);
      bl[i * 3] = r[0];
      bl[i * 3 + 1] = r[1];
      bl[i * 3 + 2] = r[2];
      if (diag) {
        cross(l1, &Q1A[i * 3], r);
        hl[i * 3] = r[0];
        hl[i * 3 + 1] = r[1];
        hl[i * 3 + 2] = r[2];
      }
    }
    if (!schur) {
#pragma unroll
      for (int i = 0; i < 9; ++i) bl[i] = 0.0;
    }
    fold(0, 0, bl, hl);
  }
  if (diag) {
    const double* aa = l1 + 9;  // bp = G^T a ; g = G^T (a - A1 R1 u)
    double c[3] = {aa[0], aa[1], aa[2]};
    if (schur) {
      const double* u = pw + 6;
      double Ru[3], ARu[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) Ru[i] = R1[i * 3] * u[0] + R1[i * 3 + 1] * u[1] + R1[i * 3 + 2] * u[2];
      sym3_mul_vec(l1 + 3, Ru, ARu);
      c[0] -= ARu[0];
      c[1] -= ARu[1];
      c[2] -= ARu[2];
    }
    double qc[3], qa[3];
    cross(l1, c, qc);
    cross(l1, aa, qa);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v2[4 + i] += qc[i];
      v2[7 + i] += c[i];
      v2[10 + i] += qa[i];
      v2[13 + i] += aa[i];
    }
  }
}

// ---- P2 -------------------------------------------------------------------------------------
// One wave per block (j1 <= j2) of the reduced camera system, its lanes over the observations of pose j1; when
// the problem's workgroups have at least 2 (4) waves per block, 2 (4) waves of a workgroup share a block (the
// observation list dealt round them, their 48 sums met in LDS).
GL_DEV void pass_blocks(const GenP& G, bool schur, double* p2part) {
  const int P = G.P, ld = G.ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = P * (P + 1) / 2;
  const int tw = G.NB * NW_BA;
  const int W = (tw >= 4 * nblk) ? 4 : (tw >= 2 * nblk) ? 2 : 1;
  const int grp = wave / W, sub = wave % W, gpw = NW_BA / W;
  const int rounds = (nblk + G.NB * gpw - 1) / (G.NB * gpw);
  for (int it = 0; it < rounds; ++it) {  // uniform trip count over the workgroup: barriers inside
    const int b = (it * G.NB + G.pb) * gpw + grp;
    bool act = b < nblk;
    // decode (j1 <= j2)
    int j1 = 0, rem = act ? b : 0;
    while (rem >= P - j1) {
      rem -= P - j1;
      ++j1;
    }
    const int j2 = j1 + rem;
    act = act && G.pact[j1] && G.pact[j2] && (schur || j1 == j2);
    double v1[32], v2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      v1[i] = 0.0;
      v2[i] = 0.0;
    }
    double R1[9], t1[3], R2[9], t2[3];
    load_Rt(G.Rt + (size_t)j1 * 12, R1, t1);
    load_Rt(G.Rt + (size_t)j2 * 12, R2, t2);
    // Software-pipelined variant (-DGL_BAGEN_PIPE, round 3: the indices and the three records of entry e + step are
    // requested while entry e is worked on) measured against this loop on one box (tools/ab_gen.sh, profiles/history/r3_bagen_ab.txt):
    // 20 + 8 key-frames 6.51 -> 6.41 ms, 256-problem batches 0.113 -> 0.107 ms per problem, but 8 + 4 key-frames
    // 2.89 -> 3.53 ms and 12 + 4 4.70 -> 5.57 ms - there 2 waves share a block, a lane has 8 entries, and the 33 live
    // prefetch registers take the kernel from 337 to 536 spilled VGPRs.  Not the default.
#ifndef GL_BAGEN_PIPE
    for (int e = G.pl_ptr[j1] + sub * 64 + lane; act && e < G.pl_ptr[j1 + 1]; e += 64 * W) {
      // (partner, observation and point come from position-indexed tables in one round of loads; the level flags
      // are folded into the partner table when they change)
      const int o2 = G.plm[(size_t)e * P + j2];
      const int o1 = G.pl_obs[e];
      const int l = G.pl_pt[e];
      if (o2 < 0) continue;
      const double* l1 = G.lin + (size_t)o1 * 12;
      const double* l2 = G.lin + (size_t)o2 * 12;
      const double* pw = G.ptw + (size_t)l * 12;
  #else
    // Software pipeline (round 3): the indices of entry e + step {partner, observation, point} and then its three records
    // are requested while entry e is being worked on - an entry cost two DEPENDENT L2 round trips (index -> record) in front
    // of ~500 instructions, 33 times in sequence per lane at 20 poses, on a workgroup with one wave per SIMD.
    const int e_end = act ? G.pl_ptr[j1 + 1] : 0, e_step = 64 * W;
    int e = G.pl_ptr[j1] + sub * 64 + lane;
    int n_o2 = -1, n_o1 = 0, n_l = 0;
    if (e < e_end) {
      n_o2 = G.plm[(size_t)e * P + j2];
      n_o1 = G.pl_obs[e];
      n_l = G.pl_pt[e];
    }
    double nl1[12], nl2[12], npw[9];
    {
      const int o2c = max(n_o2, 0);
#pragma unroll
      for (int i = 0; i < 12; ++i) nl1[i] = G.lin[(size_t)n_o1 * 12 + i];
#pragma unroll
      for (int i = 0; i < 12; ++i) nl2[i] = G.lin[(size_t)o2c * 12 + i];
#pragma unroll
      for (int i = 0; i < 9; ++i) npw[i] = G.ptw[(size_t)n_l * 12 + i];
    }
    for (; e < e_end; e += e_step) {
      const int o2 = n_o2;
      double l1[12], l2[12], pw[9];
#pragma unroll
      for (int i = 0; i < 12; ++i) l1[i] = nl1[i];
#pragma unroll
      for (int i = 0; i < 12; ++i) l2[i] = nl2[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) pw[i] = npw[i];
      {  // next entry: indices now, records as soon as they are there (both rounds overlap this entry's arithmetic)
        const int en = min(e + e_step, e_end - 1);  // (one past the end: re-reads a valid entry)
        n_o2 = (e + e_step < e_end) ? G.plm[(size_t)en * P + j2] : -1;
        n_o1 = G.pl_obs[en];
        n_l = G.pl_pt[en];
        const int o2c = max(n_o2, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) nl1[i] = G.lin[(size_t)n_o1 * 12 + i];
#pragma unroll
        for (int i = 0; i < 12; ++i) nl2[i] = G.lin[(size_t)o2c * 12 + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) npw[i] = G.ptw[(size_t)n_l * 12 + i];
      }
      if (o2 < 0) continue;
#endif
      schur_entry(l1, l2, pw, R1, R2, schur, j1 == j2, v1, v2);
    }
    double r1 = wave_reduce_scatter32(v1);
    double r2 = wave_reduce_scatter32(v2);
    if (W > 1) {
      if (wave_slot_owner(lane)) {
        p2part[wave * 64 + wave_slot(lane)] = r1;
        p2part[wave * 64 + 32 + wave_slot(lane)] = r2;
      }
      __syncthreads();
      if (sub == 0 && wave_slot_owner(lane)) {
        for (int w = 1; w < W; ++w) {
          r1 += p2part[(wave + w) * 64 + wave_slot(lane)];
          r2 += p2part[(wave + w) * 64 + 32 + wave_slot(lane)];
        }
      }
      __syncthreads();
    }
    if (act && sub == 0 && wave_slot_owner(lane)) {
      const int s = wave_slot(lane);
      {
        const int r = s / 6, c = s % 6;  // only the lower triangle of S is read: block (j2, j1) = block (j1, j2)^T
        G.S[j1 == j2 ? (size_t)(6 * j1 + r) * ld + 6 * j1 + c : (size_t)(6 * j2 + c) * ld + 6 * j1 + r] = r1;
      }
      if (s < 4) {
        const int r = (32 + s) / 6, c = (32 + s) % 6;
        G.S[j1 == j2 ? (size_t)(6 * j1 + r) * ld + 6 * j1 + c : (size_t)(6 * j2 + c) * ld + 6 * j1 + r] = r2;
      } else if (s < 10 && j1 == j2) {
        G.gv[6 * j1 + (s - 4)] = r2;
      } else if (s < 16 && j1 == j2) {
        G.bp[6 * j1 + (s - 10)] = r2;
      }
    }
  }
}

// ---- solve: LDL^T of the reduced camera system S x = g (lower triangle), whole workgroup; returns ok ------------
// Right-looking, blocked by the 6 x 6 pose blocks, the columns kept un-scaled (l_ik = S[i][k] / S[k][k] is applied
// on the fly).  Every element receives exactly the updates of the scalar algorithm - S[i][j] -= l_ik S[j][k] for
// ascending k - with the same operands, so the factor does not depend on the blocking or on who holds an element:
//   n <= 128 (up to 21 free poses): the trailing matrix lives in REGISTERS, element (i, j) with thread
//     (i mod 16, j mod 16) - up to 8 x 8 per thread; per pose block the six columns that become the panel go to the
//     LDS work matrix, one thread factorises the diagonal block, one thread per row finishes the panel, and the
//     trailing update reads 6 + 6 panel values per row / column tile of a thread and works on registers (an LDS-resident
//     trailing matrix costs 7 LDS reads and a write per 6 multiply-adds on a workgroup with one wave per SIMD).
//     The forward substitution rides along (see below); the backward one runs by pose blocks on the finished factor.
//   n > 128: scalar pivots on the matrix in memory, barrier-per-row substitutions.
// SP is `double*` (work matrix in global memory) or `lds_double*`: with the matrix known to be in LDS the accesses
// are ds_read / ds_write; through a generic pointer every one is a FLAT access (longer latency, a wait on both
// counters).  idg: 128 reciprocal pivots + 128 doubles for the right-hand side, in LDS.
typedef __attribute__((address_space(3))) double lds_double;

// prior information / lambda / unit diagonal of an inactive pose, for element (r, c <= r) of the assembled system
GL_DEV double diag_terms(const GenP& G, const BaK& k, double lambda, int r, int c, double v) {
  const int j = r / 6;
  if (c < 6 * j) return v;
  if (!G.pact[j]) return r == c ? 1.0 : v;
  if (G.prior[j] && k.first_as_prior) v += G.prH[(size_t)j * 36 + (r - 6 * j) * 6 + (c - 6 * j)];
  if (r == c) v += lambda;
  return v;
}

// diagonal block of a pose (one thread, registers), with the block's part of the forward substitution L y = g: row r
// receives y_r -= l_rk y_k in ascending k from the thread that has just formed l_rk
template <class SP>
GL_DEV void ldlt_diag_block(SP S, int ld, int base, double* idg, double* yv, int* s_flag) {
  double a[6][6], yb[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    yb[r] = yv[base + r];
#pragma unroll
    for (int c = 0; c <= r; ++c) a[r][c] = S[(size_t)(base + r) * ld + base + c];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double d = a[c][c];
    if (d == 0.0 || !isfinite(d)) *s_flag = 0;
    const double id = 1.0 / d;
    idg[base + c] = id;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double ci = a[r][c] * id;
#pragma unroll
      for (int j = c + 1; j <= r; ++j) a[r][j] -= ci * a[j][c];
      yb[r] = __builtin_fma(-ci, yb[c], yb[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    yv[base + r] = yb[r];
#pragma unroll
    for (int c = 0; c <= r; ++c) S[(size_t)(base + r) * ld + base + c] = a[r][c];
  }
}
// The same on a WAVE (round 3): lane r < 6 holds row r of the block; per pivot the pivot, the pivot row's y and the
// column's elements of the rows below travel as v_readlane broadcasts and every row updates its own elements - 6 x
// (a division + a handful of dependent operations) on the critical path instead of the ~350 sequential double-precision
// instructions of the one-thread version (2.8 k cycles per pose block, a third of the factorisation at 20 poses).
GL_DEV double readlane_f64(double v, int lane) {
  union {
    double d;
    int i[2];
  } u, w;
  u.d = v;
  w.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  w.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return w.d;
}
template <class SP>
GL_DEV void ldlt_diag_block_wave(SP S, int ld, int base, double* idg, double* yv, int* s_flag) {
  const int lane = threadIdx.x & 63, rr = min(lane, 5);
  double a[6], id[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) a[c] = c <= rr ? S[(size_t)(base + rr) * ld + base + c] : 0.0;
  double y = yv[base + rr];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double d = readlane_f64(a[c], c);
    bad = bad || d == 0.0 || !isfinite(d);
    id[c] = 1.0 / d;
    const double yc = readlane_f64(y, c);
    const double ci = a[c] * id[c];
#pragma unroll
    for (int j = c + 1; j < 6; ++j) {
      const double ajc = readlane_f64(a[c], j);  // a[j][c]
      if (rr >= j) a[j] -= ci * ajc;
    }
    if (rr > c) y = __builtin_fma(-ci, yc, y);
  }
  if (lane < 6) {
    yv[base + lane] = y;
    if (lane == 0) {  // (the reciprocal pivots are wave-uniform)
#pragma unroll
      for (int c = 0; c < 6; ++c) idg[base + c] = id[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
      if (c <= lane) S[(size_t)(base + lane) * ld + base + c] = a[c];
  }
  if (lane == 0 && bad) *s_flag = 0;
}
// panel: the block's six columns of row i (one thread per row), and the row's part of the forward substitution
template <class SP>
GL_DEV void ldlt_panel_row(SP S, int ld, int base, int i, const double* idg, double* yv) {
  double a[6], yi = yv[i];
#pragma unroll
  for (int c = 0; c < 6; ++c) a[c] = S[(size_t)i * ld + base + c];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double ci = a[c] * idg[base + c];
#pragma unroll
    for (int j = c + 1; j < 6; ++j) a[j] -= ci * S[(size_t)(base + j) * ld + base + c];
    yi = __builtin_fma(-ci, yv[base + c], yi);
  }
#pragma unroll
  for (int c = 1; c < 6; ++c) S[(size_t)i * ld + base + c] = a[c];
  yv[i] = yi;
}
// y holds L y = g.  z = D^-1 y by division, like the reference LDL^T (multiplying by the reciprocal shifts the last
// bit and the LM path at convergence); then L^T x = z by pose blocks from the last to the first: one thread finishes
// the six unknowns of a block, then every row above it takes their six contributions - x_r -= l_kr x_k in descending
// k, the order of a plain substitution, with l_kr = S[k][r] / S[r][r]; the reads S[k][r] of a step are contiguous
// over the rows.
template <class SP>
GL_DEV void ldlt_backward(SP S, int ld, int n, const double* idg, double* yv, double* g) {
  const int tid = threadIdx.x;
  if (tid < n) yv[tid] = yv[tid] / S[(size_t)tid * ld + tid];
  __syncthreads();
  for (int base = n - 6; base >= 0; base -= 6) {
    if (tid == 0) {
      double x[6], lk[6][6], ir[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        x[r] = yv[base + r];
        ir[r] = idg[base + r];
#pragma unroll
        for (int kq = r + 1; kq < 6; ++kq) lk[kq][r] = S[(size_t)(base + kq) * ld + base + r];
      }
#pragma unroll
      for (int kq = 5; kq >= 1; --kq)
#pragma unroll
        for (int r = 0; r < kq; ++r) x[r] = __builtin_fma(-(lk[kq][r] * ir[r]), x[kq], x[r]);
#pragma unroll
      for (int r = 0; r < 6; ++r) yv[base + r] = x[r];
    }
    __syncthreads();
    if (tid < base) {
      double y = yv[tid], sk[6], xk[6];
      const double ir = idg[tid];
#pragma unroll
      for (int kq = 0; kq < 6; ++kq) {
        sk[kq] = S[(size_t)(base + kq) * ld + tid];
        xk[kq] = yv[base + kq];
      }
#pragma unroll
      for (int kq = 5; kq >= 0; --kq) y = __builtin_fma(-(sk[kq] * ir), xk[kq], y);
      yv[tid] = y;
    }
    __syncthreads();
  }
  if (tid < n) g[tid] = yv[tid];
}

// Diagonal block AND panel of a pose in one phase (the solve kernel of the pipelined shape): the panel rows do, per pivot, exactly
// what the rows of the block below the pivot do - l_rc = a_rc / d_c from the broadcast pivot, a_rj -= l_rc a_jc with the
// broadcast column entries of the block's rows, y_r -= l_rc y_c - so they ride in the spare lanes: every wave keeps the six rows
// of the block in its lanes 0 .. 5 (redundantly) and 58 panel rows in the others.  One phase and one barrier less per pose, and
// the panel's own pass (0.6 - 0.7 k cycles) is gone; the operands and their order per element are those of ldlt_diag_block_wave
// + ldlt_panel_row.  The block's own rows and y leave wave 0 AFTER the barrier that follows (ad / yd): the other waves read them
// at their start.
template <class SP>
GL_DEV void ldlt_diag_panel_wave(SP S, int ld, int base, int n, double* idg, double* yv, int* s_flag, double* ad, double& yd, int wave) {
  const int lane = threadIdx.x & 63;
  const bool isd = lane < 6;
  const int prow = base + 6 + wave * 58 + (lane - 6);
  const int row = isd ? base + lane : prow, rr = min(row, n - 1);
  double a[6], id[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) a[c] = (!isd || c <= lane) ? S[(size_t)rr * ld + base + c] : 0.0;
  double y = yv[rr];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double d = readlane_f64(a[c], c);
    bad = bad || d == 0.0 || !isfinite(d);
#ifdef GL_PIPE_IEEE_DIV
    id[c] = 1.0 / d;
#else
    {  // reciprocal pivot by v_rcp_f64 + two Newton steps (<= 1 ulp from 1 / d): the correctly rounded division is 12 dependent
       // instructions on the one chain a pose step cannot shorten - pivot -> multiplier -> next pivot, six times per pose
      double x = __builtin_amdgcn_rcp(d);
      x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
      x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
      id[c] = x;
    }
#endif
    const double yc = readlane_f64(y, c);
    const double ci = a[c] * id[c];
#pragma unroll
    for (int j = c + 1; j < 6; ++j) {
      const double ajc = readlane_f64(a[c], j);  // a[j][c] of the block
      a[j] -= ci * ajc;  // (a lane of the block's own rows also touches its entries right of the diagonal: never read)
    }
    if (!isd || lane > c) y = __builtin_fma(-ci, yc, y);
  }
  if (!isd && prow < n) {
#pragma unroll
    for (int c = 1; c < 6; ++c) S[(size_t)prow * ld + base + c] = a[c];
    yv[prow] = y;
  }
  if (wave == 0 && lane == 0) {  // (the reciprocal pivots are wave-uniform)
#pragma unroll
    for (int c = 0; c < 6; ++c) idg[base + c] = id[c];
    if (bad) *s_flag = 0;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) ad[c] = a[c];
  yd = y;
}
// L^T x = z on ONE WAVE, no barrier: lane r holds rows r and r + 64; x_k travels as a v_readlane broadcast in descending k and
// every row above takes  y_r -= (S[k][r] / d_r) x_k  - the plain substitution, operand for operand what ldlt_backward does by
// pose blocks with two workgroup barriers per block (0.9 k cycles per block: a sixth to a third of the whole solve).
template <class SP>
GL_DEV void ldlt_backward_wave(SP S, int ld, int n, const double* idg, double* yv, double* g) {
  if (threadIdx.x < 64) {
    const int r = threadIdx.x, r0 = min(r, n - 1), r1 = min(r + 64, n - 1);
    double y0 = yv[r0] / S[(size_t)r0 * ld + r0];  // z = D^-1 y by division, like the reference LDL^T
    double y1 = yv[r1] / S[(size_t)r1 * ld + r1];
    const double ir0 = idg[r0], ir1 = idg[r1];
    int k = n - 1;
    for (; k >= 64; --k) {
      const double s0 = S[(size_t)k * ld + r0], s1 = S[(size_t)k * ld + r1];
      const double xk = readlane_f64(y1, k - 64);
      if (r + 64 < k) y1 = __builtin_fma(-(s1 * ir1), xk, y1);
      y0 = __builtin_fma(-(s0 * ir0), xk, y0);
    }
    for (; k >= 4; k -= 4) {  // (the four column entries requested before the dependent chain)
      double sk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sk[q] = S[(size_t)(k - q) * ld + r0] * ir0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double xk = readlane_f64(y0, k - q);
        if (r < k - q) y0 = __builtin_fma(-sk[q], xk, y0);
      }
    }
    for (; k >= 1; --k) {
      const double s0 = S[(size_t)k * ld + r0] * ir0;
      const double xk = readlane_f64(y0, k);
      if (r < k) y0 = __builtin_fma(-s0, xk, y0);
    }
    if (r < n) g[r] = y0;
    if (r + 64 < n) g[r + 64] = y1;
  }
}

// n <= 16 A <= 128, n a multiple of 6.  src (row stride lsrc): the assembled system, global memory or the work matrix
// itself; fuse: add the diagonal terms while loading.  (The solve kernel of the pipelined shape runs
// ldlt_solve_teams below: the same operands and order per element, so the same bits.)  The right-hand side is in yv = idg + 128 (put there by the caller).
template <int A, class SP>
GL_DEV bool ldlt_solve_tiles(const GenP& G, const BaK& k, double lambda, const double* src, int lsrc, bool fuse, SP S, int ld,
                             double* g, int n, int* s_flag, double* idg) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  static_assert(T_BA == 256, "thread (i mod 16, j mod 16) owns element (i, j)");
  double* yv = idg + 128;
  if (tid == 0) *s_flag = 1;
  double v[A][A];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = ti + 16 * a, j = tj + 16 * b;
      v[a][b] = (i < n && j <= i) ? src[(size_t)i * lsrc + j] : 0.0;
    }
  if (fuse) {
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        const int i = ti + 16 * a, j = tj + 16 * b;
        if (i < n && j <= i && j >= 6 * (i / 6)) v[a][b] = diag_terms(G, k, lambda, i, j, v[a][b]);
      }
  }
  for (int base = 0; base < n; base += 6) {
    GP_T(q0);
    const int m0 = base + 6;
    // the six columns of this pose leave the registers (elements of other threads' columns are not touched: the
    // trailing update of the previous block reads columns base - 6 ..., so no barrier is needed before these stores)
#pragma unroll
    for (int b = 0; b < A; ++b) {
      const int j = tj + 16 * b;
      if (j >= base && j < m0) {
#pragma unroll
        for (int a = b; a < A; ++a) {
          const int i = ti + 16 * a;
          if (i < n && j <= i) S[(size_t)i * ld + j] = v[a][b];
        }
      }
    }
    __syncthreads();
#ifdef GL_BAGEN_NO_WAVEDIAG
    if (tid == 0) ldlt_diag_block(S, ld, base, idg, yv, s_flag);
#else
    if (tid < 64) ldlt_diag_block_wave(S, ld, base, idg, yv, s_flag);
#endif
    __syncthreads();
    GP_T(q1);
    if (m0 + tid < n) ldlt_panel_row(S, ld, base, m0 + tid, idg, yv);
    __syncthreads();
    GP_T(q2);
    double idc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) idc[c] = idg[base + c];
    double ci[A][6];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      if (16 * a + 15 < m0) continue;  // (uniform) the whole row tile is factorised
      const int i = min(ti + 16 * a, n - 1);
#pragma unroll
      for (int c = 0; c < 6; ++c) ci[a][c] = S[(size_t)i * ld + base + c] * idc[c];
    }
#pragma unroll
    for (int b = 0; b < A; ++b) {
      if (16 * b + 15 < m0) continue;
      const int j = tj + 16 * b, jc = min(j, n - 1);
      double sj[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) sj[c] = S[(size_t)jc * ld + base + c];
#pragma unroll
      for (int a = b; a < A; ++a) {
        const int i = ti + 16 * a;
        if (j >= m0 && j <= i && i < n) {
          double x = v[a][b];
#pragma unroll
          for (int c = 0; c < 6; ++c) x -= ci[a][c] * sj[c];
          v[a][b] = x;
        }
      }
    }
    GP_T(q3);
    GP_ADD(9, q0, q1); GP_ADD(11, q2, q3);
  }
  ldlt_backward(S, ld, n, idg, yv, g);
  __syncthreads();
  return *s_flag != 0;
}
// The factorisation on TWO TEAMS of four waves (the solve kernel of the pipelined shape, 512 threads): team T = the 256 threads
// of ldlt_solve_tiles with the trailing matrix in their registers, team D = four waves that factorise diagonal block + panel
// (ldlt_diag_panel_wave).  One block AHEAD: as soon as T has updated the column tiles that hold the NEXT pose's six columns and
// put those columns into LDS, D factorises that pose while T applies the current panel to the rest of its tiles - the two
// phases that were a third each of a pose step run side by side (a T wave and a D wave per SIMD).  Per pose step:
//     T: TU1 (column tiles of the next pose) + column store | D: -          -> barrier
//     T: TU2 (the other tiles)                              | D: diag + panel of the next pose   -> barrier
// Operands and order per element are those of ldlt_solve_tiles: the same bits.
template <int A, class SP>
GL_DEV bool ldlt_solve_teams(const GenP& G, const BaK& k, double lambda, const double* src, int lsrc, SP S, int ld, double* g, int n,
                             int* s_flag, double* idg) {
  const int tid = threadIdx.x, tt = tid & 255, ti = tt >> 4, tj = tt & 15;
  const bool teamT = tid < 256;
  double* yv = idg + 128;
  if (tid == 0) *s_flag = 1;
  double v[A][A];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = ti + 16 * a, j = tj + 16 * b;
      v[a][b] = (teamT && i < n && j <= i) ? src[(size_t)i * lsrc + j] : 0.0;
    }
  if (teamT) {
    // diag_terms() without its branches and dependent loads: an element of the diagonal block of its row's pose sits in column
    // tile a or a - 1; flags and prior information per row tile come first
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const int i = ti + 16 * a, jp = min(i, n - 1) / 6, r6 = min(i, n - 1) - 6 * jp;
      const bool act = G.pact[jp] != 0, pri = act && G.prior[jp] && k.first_as_prior;
      const double* H = G.prH + (size_t)jp * 36 + r6 * 6;
#pragma unroll
      for (int b = (a > 0 ? a - 1 : 0); b <= a; ++b) {
        const int j = tj + 16 * b, c6 = j - 6 * jp;
        const bool in = i < n && j <= i && c6 >= 0;
        const double h = H[min(max(c6, 0), 5)];
        double x = v[a][b];
        x = pri ? x + h : x;
        if (i == j) x = act ? x + lambda : 1.0;
        v[a][b] = in ? x : v[a][b];
      }
    }
  }
  // the six columns [cb, cb + 6) leave the registers of their owners
  auto colstore = [&](int cb) {
#pragma unroll
    for (int b = 0; b < A; ++b) {
      if (b != cb / 16 && b != (cb + 5) / 16) continue;  // (uniform: six columns touch at most two column tiles)
      const int j = tj + 16 * b;
      if (j >= cb && j < cb + 6) {
#pragma unroll
        for (int a = b; a < A; ++a) {
          const int i = ti + 16 * a;
          if (i < n && j <= i) S[(size_t)i * ld + j] = v[a][b];
        }
      }
    }
  };
  double ad[6], yd = 0.0;  // D, wave 0, lanes 0 .. 5: the block's own rows, stored behind the barrier that follows (see ldlt_diag_panel_wave)
  auto diag_store = [&](int base) {
    if (tid >= 256 && tid < 256 + 6) {
      const int r = tid - 256;
      yv[base + r] = yd;
#pragma unroll
      for (int c = 0; c < 6; ++c)
        if (c <= r) S[(size_t)(base + r) * ld + base + c] = ad[c];
    }
  };
  if (teamT) colstore(0);
  __syncthreads();
  if (!teamT) ldlt_diag_panel_wave(S, ld, 0, n, idg, yv, s_flag, ad, yd, (tid >> 6) - 4);
  __syncthreads();
  for (int base = 0; base < n; base += 6) {
    const int m0 = base + 6;
    const int nb0 = m0 / 16, nb1 = (m0 + 5) / 16;  // column tiles of the next pose
    double ci[A][6];
    if (teamT) {
      double idc[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) idc[c] = idg[base + c];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        if (16 * a + 15 < m0) continue;  // (uniform) the whole row tile is factorised
        const int i = min(ti + 16 * a, n - 1);
#pragma unroll
        for (int c = 0; c < 6; ++c) ci[a][c] = S[(size_t)i * ld + base + c] * idc[c];
      }
    } else {
      diag_store(base);
    }
    auto update = [&](bool first) {
#pragma unroll
      for (int b = 0; b < A; ++b) {
        if (16 * b + 15 < m0) continue;
        if (((b == nb0) | (b == nb1)) != first) continue;  // (uniform)
        const int j = tj + 16 * b, jc = min(j, n - 1);
        double sj[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) sj[c] = S[(size_t)jc * ld + base + c];
#pragma unroll
        for (int a = b; a < A; ++a) {
          const int i = ti + 16 * a;
          if (j >= m0 && j <= i && i < n) {
            double x = v[a][b];
#pragma unroll
            for (int c = 0; c < 6; ++c) x -= ci[a][c] * sj[c];
            v[a][b] = x;
          }
        }
      }
    };
    if (m0 < n) {
      if (teamT) {
        update(true);
        colstore(m0);
      }
      __syncthreads();
      if (teamT) update(false);
      else ldlt_diag_panel_wave(S, ld, m0, n, idg, yv, s_flag, ad, yd, (tid >> 6) - 4);
      __syncthreads();
    }
  }
  __syncthreads();  // (the last pose's own rows come from team D)
  ldlt_backward_wave(S, ld, n, idg, yv, g);
  __syncthreads();
  return *s_flag != 0;
}
// n <= NMAX <= 48 (up to 8 free poses), round 3: the whole factorisation on ONE WAVE, lane r = row r held in registers
// (NMAX doubles), no barrier and no LDS inside: per pivot k the pivot and the column's elements of the rows below travel as
// v_readlane broadcasts (A[j][k] is register k of lane j - static indices in the unrolled code) and every lane updates its own
// row, S[r][j] -= l_rk S[j][k] in ascending k like the scalar loop; the forward substitution rides along, z = D^-1 y by
// division, L^T x = z as 48 wave sums.  ~4 300 instructions on one wave against the blocked algorithm's 8 x (3 barriers +
// a one-wave diagonal block + panel + register-tile update).  MEASURED (profiles/history/r3_bagen_shapes.txt), NOT ENABLED (-DGL_BAGEN_WAVE_LDLT): in
// the pipelined shape's solve kernel a window of 8 + 4 key-frames goes 2.96 -> 2.90 ms; inlined into the persistent kernel the
// 48-register rows take it from 272 to 809 spilled VGPRs and the same window from 2.87 to 3.65 ms.
// Rows n .. NMAX-1 are identity padding (they cost their share of the unrolled code: two instances, 24 and 48).
template <int NMAX>
GL_DEV bool ldlt_solve_wave(const GenP& G, const BaK& k, double lambda, const double* src, int lsrc, bool fuse, double* g, int n,
                            int* s_flag, const double* yv) {
  if (threadIdx.x < 64) {
    const int r = threadIdx.x & 63, rr = min(r, n - 1);
    double a[NMAX], y = r < n ? yv[rr] : 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      double v = (c == r) ? 1.0 : 0.0;  // identity padding beyond the system
      if (r < n && c <= r) {
        v = src[(size_t)r * lsrc + c];
        if (fuse && c >= 6 * (r / 6)) v = diag_terms(G, k, lambda, r, c, v);
      }
      a[c] = v;
    }
    bool bad = false;
#pragma unroll
    for (int kk = 0; kk < NMAX; ++kk) {
      const double d = readlane_f64(a[kk], kk);
      bad = bad || d == 0.0 || !isfinite(d);
      const double yk = readlane_f64(y, kk);
      const double l = a[kk] * (1.0 / d);  // l_rk of this lane's row (meaningful for r > k)
#pragma unroll
      for (int j = kk + 1; j < NMAX; ++j) a[j] = __builtin_fma(-l, readlane_f64(a[kk], j), a[j]);  // (entries right of the diagonal: never read)
      if (r > kk) {
        y = __builtin_fma(-l, yk, y);
        a[kk] = l;
      }
    }
    // z = D^-1 y (division, like the reference LDL^T); a[r] of lane r is d_r
    double dr = 1.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c == r) dr = a[c];
    double x = y / dr;
    // L^T x = z: x_k = z_k - sum_{r > k} l_rk x_r, k descending; lane r holds l_rk (register k) and x_r
#pragma unroll
    for (int kk = NMAX - 2; kk >= 0; --kk) {
      double t = r > kk ? a[kk] * x : 0.0;
      t += dpp_f64<0xB1>(t);   // lane ^ 1
      t += dpp_f64<0x4E>(t);   // lane ^ 2
      t += dpp_f64<0x141>(t);  // lane ^ 7  (row_half_mirror)
      t += dpp_f64<0x140>(t);  // lane ^ 15 (row_mirror): every lane of a row of 16 holds the row's sum
      const double tot = (readlane_f64(t, 0) + readlane_f64(t, 16)) + (readlane_f64(t, 32) + readlane_f64(t, 48));
      if (r == kk) x -= tot;
    }
    if (r < n) g[r] = x;
    if (r == 0) *s_flag = bad ? 0 : 1;
  }
  __syncthreads();
  return *s_flag != 0;
}

template <class SP>
GL_DEV bool ldlt_solve_small(const GenP& G, const BaK& k, double lambda, const double* src, int lsrc, bool fuse, SP S, int ld,
                             double* g, int n, int* s_flag, double* idg) {
#ifdef GL_BAGEN_WAVE_LDLT  // (measured, not the default: see ldlt_solve_wave)
  if (n <= 24) return ldlt_solve_wave<24>(G, k, lambda, src, lsrc, fuse, g, n, s_flag, idg + 128);
  if (n <= 48) return ldlt_solve_wave<48>(G, k, lambda, src, lsrc, fuse, g, n, s_flag, idg + 128);
#endif
  if (n <= 32) return ldlt_solve_tiles<2>(G, k, lambda, src, lsrc, fuse, S, ld, g, n, s_flag, idg);
  if (n <= 48) return ldlt_solve_tiles<3>(G, k, lambda, src, lsrc, fuse, S, ld, g, n, s_flag, idg);
  if (n <= 80) return ldlt_solve_tiles<5>(G, k, lambda, src, lsrc, fuse, S, ld, g, n, s_flag, idg);
  return ldlt_solve_tiles<8>(G, k, lambda, src, lsrc, fuse, S, ld, g, n, s_flag, idg);
}

// n > 128 (more than 21 free poses): scalar pivots, ONE barrier per pivot, on the matrix in memory; g in, x out
GL_DEV bool ldlt_solve_large(double* S, double* g, int n, int ld, int* s_flag) {
  const int tid = threadIdx.x;
  if (tid == 0) *s_flag = 1;
  __syncthreads();
  for (int kk = 0; kk < n; ++kk) {
    const double d = S[(size_t)kk * ld + kk];
    if (tid == 0 && (d == 0.0 || !isfinite(d))) *s_flag = 0;
    // trailing update with the un-scaled column: S[i][j] -= c_i c_j / d  (kk < j <= i)
    const double id = 1.0 / d;
    for (int i = kk + 1 + (tid >> 4); i < n && tid < T_BA; i += T_BA / 16) {  // (the solve kernel of the pipelined shape has 512 threads)
      const double ci = S[(size_t)i * ld + kk] * id;
      for (int j = kk + 1 + (tid & 15); j <= i; j += 16) S[(size_t)i * ld + j] -= ci * S[(size_t)j * ld + kk];
    }
    __syncthreads();
  }
  for (int kk = 0; kk < n; ++kk) {
    const double yk = g[kk] / S[(size_t)kk * ld + kk];
    __syncthreads();
    for (int i = kk + 1 + tid; i < n && tid < T_BA; i += T_BA) g[i] -= S[(size_t)i * ld + kk] * yk;
    __syncthreads();
  }
  for (int i = tid; i < n && tid < T_BA; i += T_BA) g[i] /= S[(size_t)i * ld + i];
  __syncthreads();
  for (int kk = n - 1; kk >= 0; --kk) {
    const double xk = g[kk];
    __syncthreads();
    for (int i = tid; i < kk && tid < T_BA; i += T_BA) g[i] -= S[(size_t)kk * ld + i] / S[(size_t)i * ld + i] * xk;
    __syncthreads();
  }
  return *s_flag != 0;
}

GL_DEV int gen_optimize(const BaK& k, const GmmDev& gm, GenP& G, k_genp* sG, bool robust, int iters, double* red, int* s_flag,
                        double* s_lds, double* p2part) {
  const int P = G.P, n = 6 * P, ld = G.ld, tid = threadIdx.x;
  // lanes per point in the point passes: as many (1, 2, 4) as the problem's threads allow in one round
  const int lpp = (G.NB * T_BA >= 4 * G.L) ? 4 : (G.NB * T_BA >= 2 * G.L) ? 2 : 1;
  double acc[32];
  // ---- initializeOptimization(0): active poses / points -----------------------------------
  for (int j = GSTART; j < P; j += GSTRIDE) G.pact[j] = (G.pfree[j] && G.prior[j] && k.first_as_prior) ? 1 : 0;
  prob_sync(G);
  GFRESH();
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int l = GSTART; l < G.L; l += GSTRIDE) {
    bool any = G.assoc[l] >= 0 && !G.lev_g[l];
    for (int o = G.optr[l]; o < G.optr[l + 1]; ++o)
      if (!G.lev_o[o]) {
        any = true;
        const int j = G.opose[o];
        if (G.pfree[j]) G.pact[j] = 1;  // benign same-value race
      }
    G.lact[l] = any ? 1 : 0;
    if (any) acc[0] += 1.0;
  }
  prob_sync(G);
  GFRESH();
  for (int j = GSTART; j < P; j += GSTRIDE)
    if (G.pact[j]) acc[1] += 1.0;
  prob_reduce<2>(G, acc, red);
  GFRESH();
  const bool any_point = acc[0] > 0.0, any_pose = acc[1] > 0.0;
  if (!any_point && !any_pose) return -1;
  double* S = G.S;  // where P2 / priors assemble the reduced system (global when NB > 1)

  double lambda = 0.0, ni = 2.0;
  int cj = 0;
  bool fresh = false;     // the records of the last point pass are those of the current state (point_relambda)
  double chi_keep = 0.0;  // this thread's part of the robust chi2 of that pass
  for (int it = 0; it < iters; ++it) {
    if (stop_now(G)) break;  // SparseOptimizer::optimize: `i < iterations && !terminate()`
    // Prior edges at the current state: that state changes once per outer iteration (an accepted trial ends the
    // inner loop), so their information / rhs / chi2 are formed here and not per trial, by the LAST threads of the
    // problem - which have no point of the first pass when the problem's threads outnumber its points - so that the
    // serial pose algebra overlaps the point pass; the barriers of that pass publish the result.
    {
      const int rev = G.NB * T_BA - 1 - (G.pb * T_BA + tid);
      if (rev < P) {
        const int j = rev;
        double H[36], b[6] = {0, 0, 0, 0, 0, 0}, chi = 0.0;
        for (int r = 0; r < 36; ++r) H[r] = 0.0;
        if (G.pact[j] && G.prior[j] && k.first_as_prior)
          chi = prior_terms(se3_load(G.pinv + (size_t)j * 7), se3_load(G.poses + (size_t)j * 7), true, H, b);
        for (int r = 0; r < 36; ++r) G.prH[(size_t)j * 36 + r] = H[r];
        for (int r = 0; r < 6; ++r) G.prb[(size_t)j * 6 + r] = b[r];
        G.pchi[j] = chi;
      }
    }
    if (it == 0) {  // computeLambdaInit
      double md = 0.0;
      chi_keep = pass_points_lpp(lpp, k, gm, G, robust, 0.0, md);
      fresh = true;
      for (int i = GSTART; i < n * ld; i += GSTRIDE) S[i] = 0.0;
      prob_sync(G);
      GFRESH();
      pass_blocks(G, false, p2part);
      prob_sync(G);
      GFRESH();
      for (int j = GSTART; j < P; j += GSTRIDE) {
        if (!G.pact[j]) continue;
        for (int r = 0; r < 6; ++r) md = fmax(md, fabs(S[(size_t)(6 * j + r) * ld + 6 * j + r] + G.prH[(size_t)j * 36 + r * 6 + r]));
      }
      md = prob_max(G, md, red);
      GFRESH();
      lambda = 1e-5 * md;
      ni = 2.0;
    }
    double rho = 0.0, currentChi = 0.0;
    int qmax = 0;
    do {
      // ---- P1 + P2 at the current state -------------------------------------------------------
      double md_unused = 0.0;
      GP_T(t0);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
      // (nothing has moved since the last point pass - a rejected trial's successor, the first trial after the lambda-init pass:
      // only the damped point blocks are rebuilt, like kp_lin; the thread's part of chi2 is the one it kept)
      if (fresh) point_relambda(G, lambda, G.pth);
      else chi_keep = pass_points_lpp(lpp, k, gm, G, robust, lambda, md_unused);
      acc[0] = chi_keep;
      for (int i = GSTART; i < n * ld; i += GSTRIDE) S[i] = 0.0;
      for (int i = GSTART; i < n; i += GSTRIDE) {
        G.gv[i] = 0.0;
        G.bp[i] = 0.0;
      }
      prob_reduce<1>(G, acc, red);  // (its barriers also publish P1's per-point results)
      GFRESH();
      double chiA = acc[0];
      if (G.NB == 1) __syncthreads();
      GP_T(t1);
      pass_blocks(G, true, p2part);
      prob_sync(G);
      GFRESH();
      GP_T(t2);
      // ---- workgroup 0: priors / inactive poses, solve, trial poses (P <= ~20 poses: not worth a barrier each)
      bool ok2 = true;
      GP_T(t3);
      if (G.pb == 0) {
        // n <= 128: the solve loads the assembled system into registers (from global memory when the problem has
        // several workgroups, adding the diagonal terms on the way; from this workgroup's LDS, in place, when it has
        // one) and factorises it in the LDS work matrix; larger systems are factorised in place in global memory
        const bool small = n <= 128 && s_lds != nullptr;
        if (G.NB == 1 || !small) {
          for (int j = tid; j < P; j += T_BA) {
            if (!G.pact[j]) {
              for (int r = 0; r < 6; ++r) S[(size_t)(6 * j + r) * ld + 6 * j + r] = 1.0;
              continue;
            }
            if (G.prior[j] && k.first_as_prior) {
              const double* H = G.prH + (size_t)j * 36;
              for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) S[(size_t)(6 * j + r) * ld + 6 * j + c] += H[r * 6 + c];
            }
            for (int r = 0; r < 6; ++r) S[(size_t)(6 * j + r) * ld + 6 * j + r] += lambda;
          }
        }
        // right-hand side (+ the prior's), into the solve's LDS vector when the system is small enough for it
        for (int i = tid; i < n; i += T_BA) {
          const int j = i / 6;
          double v = G.gv[i];
          if (G.pact[j] && G.prior[j] && k.first_as_prior) {
            const double b = G.prb[i];
            v += b;
            G.bp[i] += b;
          }
          G.dxv[i] = v;
          if (n <= 128) red[128 + i] = v;
        }
        __syncthreads();
        GP_T(u0);
        bool ok = true;
        if (any_pose)
          ok = small ? ldlt_solve_small(G, k, lambda, S, ld, G.NB > 1, (lds_double*)s_lds, ld, G.dxv, n, s_flag, red)
                     : ldlt_solve_large(S, G.dxv, n, ld, s_flag);
        if (tid == 0) {
          *G.flagg = ok ? 1 : 0;
          if (G.stop) G.flagg[1] = stop_word_load(G.stop);
        }
        __syncthreads();
        GP_T(u2);
        GP_ADD(6, t3, u0); GP_ADD(8, u0, u2);
        // trial poses
        for (int j = tid; j < P; j += T_BA) {
          const SE3 T = se3_load(G.poses + (size_t)j * 7);
          SE3 Tn = T;
          if (G.pact[j] && ok) {
            double dx[6];
            for (int r = 0; r < 6; ++r) dx[r] = G.dxv[6 * j + r];
            Tn = se3_mul(se3_exp(dx), T);
          } else {
            for (int r = 0; r < 6; ++r) G.dxv[6 * j + r] = 0.0;
          }
          se3_store(Tn, G.qN + (size_t)j * 7);
          store_pose_Rt(Tn, G.RtN + (size_t)j * 12);
          G.pchi2[j] = (G.pact[j] && G.prior[j] && k.first_as_prior)
                           ? prior_terms(se3_load(G.pinv + (size_t)j * 7), Tn, false, nullptr, nullptr)
                           : 0.0;
        }
      }
      prob_sync(G);
      GFRESH();
      ok2 = *G.flagg != 0;
      if (G.stop) G.stop_seen = G.flagg[1];
      for (int j = 0; j < P; ++j) chiA += G.pchi[j];
      if (qmax == 0) currentChi = chiA;
      GP_T(t4);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
      pass_trial_lpp(lpp, k, gm, G, robust, lambda, P, acc);
      prob_reduce<2>(G, acc, red);
      GFRESH();
      double scale = acc[0], tempChi = acc[1];
      for (int j = 0; j < P; ++j) tempChi += G.pchi2[j];
      for (int i = 0; i < n; ++i) scale += G.dxv[i] * (lambda * G.dxv[i] + G.bp[i]);
      if (!ok2) tempChi = 1.7976931348623157e308;
      scale += 1e-3;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        const double uu = 2 * rho - 1;
        double alpha = 1. - uu * uu * uu;
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        fresh = false;
        for (int j = GSTART; j < P; j += GSTRIDE) {
          if (!G.pact[j]) continue;
          for (int r = 0; r < 7; ++r) G.poses[(size_t)j * 7 + r] = G.qN[(size_t)j * 7 + r];
          for (int r = 0; r < 12; ++r) G.Rt[(size_t)j * 12 + r] = G.RtN[(size_t)j * 12 + r];
        }
        for (int l = GSTART; l < G.L; l += GSTRIDE) {
          if (!G.lact[l]) continue;
#pragma unroll
          for (int i = 0; i < 3; ++i) G.pts[(size_t)l * 3 + i] = G.pn[(size_t)l * 3 + i];
        }
      } else {
        lambda *= ni;
        ni *= 2;
        fresh = true;
      }
      prob_sync(G);
      GFRESH();
      qmax++;
      ++G.trials;
      GP_T(t5);
      GP_ADD(0, t0, t1); GP_ADD(1, t1, t2); GP_ADD(2, t2, t3); GP_ADD(3, t3, t4); GP_ADD(4, t4, t5); GP_ADD(5, 0, 1);
    } while (rho < 0 && qmax < 10 && !(G.stop_seen > 0));  // (g2o's retry loop tests terminate() too)
    ++cj;
    ++G.done_iters;
    if (qmax == 10 || rho == 0) break;
  }
  return cj;
}

// (TABLE: the same argument list and prologue, one launch in front of the real one: it WRITES the address table and returns, so that
// the table is constant - as the constant address space of the scalar loads in genp_fresh promises - for the whole dispatch that reads
// it: ADVICE r5.  Until round 6 every workgroup of the dispatch wrote the table itself and read it back behind a fence.)
template <bool TABLE>
__global__ __launch_bounds__(T_BA) void k_ba_gen_t(BaK k, GmmDev gm, int B, int NB, int P, int F, int L, int NOBS,
                                                 double* __restrict__ poses_all, const uint8_t* __restrict__ prior_all,
                                                 double* __restrict__ pts_all, const int32_t* __restrict__ assoc_all,
                                                 const int32_t* __restrict__ optr_all,
                                                 const int32_t* __restrict__ opose_all,
                                                 const double* __restrict__ ouvr_all,
                                                 const int32_t* __restrict__ ooct_all,
                                                 uint8_t* __restrict__ dropped_all, uint8_t* __restrict__ erase_all,
                                                 int32_t* __restrict__ iters_all, char* __restrict__ scratch,
                                                 size_t scratch_per_problem, int s_in_lds, const int32_t* __restrict__ stop_dev,
                                                 int32_t* __restrict__ trials_out) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];  // reduced camera system when it fits
  __shared__ double red[NW_BA * 32 + 128];  // reductions / reciprocal pivots (128) + the right-hand side of the solve (128)
  __shared__ double p2part[NW_BA * 64];  // pass_blocks: sums of the waves that share a block
  __shared__ int s_flag;
  const int f = blockIdx.x / NB, tid = threadIdx.x;
  if (f >= B) return;
  GenP G;
  G.NB = NB;
  G.pb = blockIdx.x % NB;
  G.P = P;
  G.F = F;
  G.L = L;
  G.poses = poses_all + (size_t)f * (P + F) * 7;
  G.prior = prior_all + (size_t)f * P;
  G.pts = pts_all + (size_t)f * L * 3;
  G.assoc = assoc_all + (size_t)f * L;
  G.optr = optr_all + (size_t)f * (L + 1);
  G.opose = opose_all + (size_t)f * NOBS;
  G.ouvr = ouvr_all + (size_t)f * NOBS * 3;
  G.ooct = ooct_all + (size_t)f * NOBS;
  G.nobs = G.optr[L];
  const int nobs = G.nobs, n = 6 * P;
  // scratch = | B x 512 B headers {64 barrier words, solve flag} (zeroed by the host) | per-problem areas |
  G.bar = (unsigned*)(scratch + (size_t)f * 512);
  G.flagg = (int*)(G.bar + 64);
  G.stop = stop_dev;
  G.stop_seen = 0;
  G.done_iters = 0;
  G.trials = 0;
  if (stop_dev && G.pb == 0 && tid == 0) G.flagg[1] = stop_word_load(stop_dev);  // (published by the set-up's first barrier)
  // carve the problem's area (doubles first, then ints, then bytes)
  char* s = scratch + (size_t)B * 512 + (size_t)f * scratch_per_problem;
  auto takeD = [&](size_t cnt) {
    double* p = (double*)s;
    s += cnt * 8;
    return p;
  };
  G.Rt = takeD((size_t)(P + F) * 12);
  G.RtN = takeD((size_t)P * 12);
  G.qN = takeD((size_t)P * 7);
  G.pinv = takeD((size_t)P * 7);
  G.pn = takeD((size_t)L * 3);
  G.lin = takeD((size_t)NOBS * 12);
  G.ptw = takeD((size_t)L * 12);
  G.chi_o = takeD((size_t)NOBS);
#ifndef GL_LD_PAD
#define GL_LD_PAD 1  // odd row stride: the column reads of the solve spread over the LDS banks
#endif
  G.ld = n + GL_LD_PAD;  // row stride of S
  G.S = takeD((size_t)n * G.ld);
  G.part = takeD((size_t)2 * 64 * 4);
  G.toggle = 0;
  G.epoch = 0u;
  // one workgroup: the in-place LDL^T does ~4n barriers, keep S next to the CU (6P <= 120)
  if (s_in_lds && NB == 1) G.S = dyn_lds;
  G.gv = takeD(n);
  G.bp = takeD(n);
  G.dxv = takeD(n);
  G.pchi = takeD(P);
  G.pchi2 = takeD(P);
  G.prH = takeD((size_t)P * 36);
  G.prb = takeD((size_t)P * 6);
  G.pth = takeD((size_t)L * 6);
  auto takeI = [&](size_t cnt) {
    int32_t* p = (int32_t*)s;
    s += ((cnt * 4 + 7) / 8) * 8;
    return p;
  };
  G.opoint = takeI(NOBS);
  G.pl_ptr = takeI(P + 1);
  G.pl_obs = takeI(NOBS);
  G.pl_pos = takeI(NOBS);
  G.pl_pt = takeI(NOBS);
  G.plm = takeI((size_t)NOBS * P);
  auto takeB = [&](size_t cnt) {
    uint8_t* p = (uint8_t*)s;
    s += ((cnt + 7) / 8) * 8;
    return p;
  };
  G.lev_o = takeB(NOBS);
  G.lev_g = takeB(L);
  G.pfree = takeB(P + F);
  G.pact = takeB(P);
  G.lact = takeB(L);
  // the address table (header comment of genp_fresh): written by the TABLE launch in front of this one, read-only here
  GenP* const tab = (GenP*)(scratch + (size_t)B * 512 + (size_t)(f + 1) * scratch_per_problem - GEN_TABLE_BYTES);
  k_genp* const sG = (k_genp*)tab;
  if (TABLE) {
    if (tid == 0 && G.pb == 0) *tab = G;
    return;
  }
  GFRESH();

  // ---- setup --------------------------------------------------------------------------------------
  for (int j = GSTART; j < P + F; j += GSTRIDE) {
    const SE3 T = se3_load(G.poses + (size_t)j * 7);
    store_pose_Rt(T, G.Rt + (size_t)j * 12);
    bool fr = j < P;
    if (fr && G.prior[j] && !k.first_as_prior) fr = false;  // vSE3->setFixed(idx_ == 0) (:578-580)
    G.pfree[j] = fr ? 1 : 0;
    if (j < P) se3_store(se3_inverse(T), G.pinv + (size_t)j * 7);  // e->setMeasurement(kf->getTcw())
  }
  for (int l = GSTART; l < L; l += GSTRIDE) {
    G.lev_g[l] = 0;
    for (int o = G.optr[l]; o < G.optr[l + 1]; ++o) {
      G.opoint[o] = l;
      G.lev_o[o] = 0;
      G.chi_o[o] = 0.0;
    }
  }
  for (size_t i = GSTART; i < (size_t)nobs * P; i += GSTRIDE) G.plm[i] = -1;
  prob_sync(G);
  GFRESH();
  if (stop_dev) {
    G.stop_seen = G.flagg[1];
    if (G.stop_seen > 0) {  // `if (pbStopFlag) if (*pbStopFlag) return;` (:765-767): nothing is written
      if (G.pb == 0 && tid == 0 && iters_all) iters_all[f] = 0;
      return;
    }
  }
  // pose-major CSR: wave-per-pose ordered compaction (two sweeps: count, fill)
  {
    const int lane = tid & 63, gw = G.pb * NW_BA + (tid >> 6), gnw = G.NB * NW_BA;
    for (int j = gw; j < P; j += gnw) {
      int cnt = 0;
      for (int o0 = 0; o0 < nobs; o0 += 64) {
        const int o = o0 + lane;
        const bool hit = o < nobs && G.opose[o] == j;
        cnt += __popcll(__ballot(hit));
      }
      if (lane == 0) G.pl_ptr[j + 1] = cnt;
    }
    prob_sync(G);
    GFRESH();
    if (G.pb == 0 && tid == 0) {
      G.pl_ptr[0] = 0;
      for (int j = 0; j < P; ++j) G.pl_ptr[j + 1] += G.pl_ptr[j];
    }
    prob_sync(G);
    GFRESH();
    for (int j = gw; j < P; j += gnw) {
      int base = G.pl_ptr[j];
      for (int o0 = 0; o0 < nobs; o0 += 64) {
        const int o = o0 + lane;
        const bool hit = o < nobs && G.opose[o] == j;
        const unsigned long long m = __ballot(hit);
        if (hit) {
          const int e = base + __popcll(m & ((1ull << lane) - 1ull));
          G.pl_obs[e] = o;
          G.pl_pos[o] = e;
          G.pl_pt[e] = G.opoint[o];
        }
        base += __popcll(m);
      }
    }
  }
  prob_sync(G);
  GFRESH();
  // partner table, indexed by list position: entry (e, j2) = the observation of the same point in free pose j2
  for (int l = GSTART; l < L; l += GSTRIDE)
    for (int o1 = G.optr[l]; o1 < G.optr[l + 1]; ++o1) {
      if (G.opose[o1] >= P) continue;
      const size_t row = (size_t)G.pl_pos[o1] * P;
      for (int o2 = G.optr[l]; o2 < G.optr[l + 1]; ++o2) {
        const int j2 = G.opose[o2];
        if (j2 < P) G.plm[row + j2] = o2;
      }
    }
  prob_sync(G);
  GFRESH();
  double* s_lds = s_in_lds ? dyn_lds : nullptr;

  // ---- schedule (:770-828) ---------------------------------------------------------------------------
  // (three inlined copies of the optimiser, each with its constants folded: a loop over the stages gives a third of
  // the code but other contraction decisions, i.e. other last bits than the validated build)
  gen_optimize(k, gm, G, sG, true, 5, red, &s_flag, s_lds, p2part);
  prob_sync(G);
  GFRESH();
  for (int l = GSTART; l < L; l += GSTRIDE) {
    GmmRef g;
    load_gmm(G.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
    if (g.has && g.deg && gmm_chi2(k, g, G.pts + (size_t)l * 3) > k.str_thresh) G.lev_g[l] = 1;
  }
  prob_sync(G);
  GFRESH();
  gen_optimize(k, gm, G, sG, true, 5, red, &s_flag, s_lds, p2part);
  prob_sync(G);
  GFRESH();
  int it3 = 0;
  if (!stop_now(G)) {  // bDoMore (:791-796)
    for (int l = GSTART; l < L; l += GSTRIDE)
      for (int o = G.optr[l]; o < G.optr[l + 1]; ++o) {
        const double* Rt = G.Rt + (size_t)G.opose[o] * 12;
        const double* p = G.pts + (size_t)l * 3;
        const double z = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2] + Rt[11];
        const bool stereo = !(G.ouvr[(size_t)o * 3 + 2] < 0);
        if (G.chi_o[o] > (stereo ? 7.815 : 5.991) || !(z > 0.0)) G.lev_o[o] = 1;
      }
    prob_sync(G);
    GFRESH();
    // the partner table forgets the edges that are now at level 1
    for (size_t i = GSTART; i < (size_t)G.pl_ptr[P] * P; i += GSTRIDE) {
      const int o2 = G.plm[i];
      if (o2 >= 0 && (G.lev_o[o2] || G.lev_o[G.pl_obs[i / P]])) G.plm[i] = -1;
    }
    prob_sync(G);
    GFRESH();
    it3 = gen_optimize(k, gm, G, sG, false, 40, red, &s_flag, s_lds, p2part);
    prob_sync(G);
    GFRESH();
  }
  // ---- outputs (:837-879) ---------------------------------------------------------------------------
  for (int l = GSTART; l < L; l += GSTRIDE) {
    GmmRef g;
    load_gmm(G.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
    const double* p = G.pts + (size_t)l * 3;
    dropped_all[(size_t)f * L + l] = (g.has && g.deg && gmm_chi2(k, g, p) > k.str_thresh) ? 1 : 0;
    for (int o = G.optr[l]; o < G.optr[l + 1]; ++o) {
      const double* Rt = G.Rt + (size_t)G.opose[o] * 12;
      const double z = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2] + Rt[11];
      const bool stereo = !(G.ouvr[(size_t)o * 3 + 2] < 0);
      erase_all[(size_t)f * NOBS + o] = (G.chi_o[o] > (stereo ? 7.815 : 5.991) || !(z > 0.0)) ? 1 : 0;
    }
  }
  if (G.pb == 0 && tid == 0 && iters_all) iters_all[f] = it3;
  if (G.pb == 0 && tid == 0 && trials_out) trials_out[f] = G.trials;
#ifdef GL_BAGEN_PROF
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 12; ++i) G.poses[i] = (double)g_gprof[i];  // debug build: phase cycles instead of poses 0-1
#endif
}

size_t gen_scratch_bytes(int P, int F, int L, int NOBS) {
  const size_t n = 6 * (size_t)P;
  size_t d = (size_t)(P + F) * 12 + (size_t)P * 12 + (size_t)P * 7 * 2 + (size_t)L * 3 + (size_t)NOBS * 12 +
             (size_t)L * 12 + NOBS + n * (n + 2) + 512 + 3 * n + 2 * P + 42 * (size_t)P + (size_t)L * 6;
  size_t i = (size_t)NOBS * 4 + (P + 1) + (size_t)NOBS * P + 32;
  size_t b = (size_t)NOBS + 2 * (size_t)L + (P + F) + P + 64;
  return d * 8 + i * 4 + b + 256 + 2 * GEN_TABLE_BYTES;  // (+ k_ba_gen's address table in the last 512 bytes of the 256-aligned area)
}

}  // namespace


// =====================================================================================================================
// The PIPELINED shape of the local BA (round 3): the same phases as k_ba_gen - P1 linearise, P2 Schur blocks, solve, P3 trial
// state, accept / reject - as SEPARATE kernels over the same scratch layout and the same per-element arithmetic (pass_points,
// pass_trial, gmg, the LDL^T of ldlt_solve_tiles element for element), driven by a per-problem state word in global memory:
// kp_lin (judges the previous trial, then P1) -> kp_schur (P2) -> kp_assemble -> kp_solve -> kp_trial (P3) per Levenberg trial.
// Why: the persistent kernel is one register allocation for every phase (512 registers, hundreds spilled, one wave per SIMD:
// every dependent load of P2 / P3 is exposed), its phases are separated by problem-wide barriers among 16 - 64 co-resident
// workgroups (10 - 14 % of a trial), and 15 - 63 of them idle while workgroup 0 factorises (a quarter to a third of a trial).
// Here every phase has its own allocation and its own grid - P2 runs one wave per 256 entries of a block's list, thousands of
// waves at 20 poses - kernel boundaries replace the barriers, and the factorisation is the only single-workgroup step.
// The Levenberg control flow is data dependent, so the host enqueues CYCLES of the five kernels ahead of the device
// (a cycle = one trial; a kernel whose problem has finished, or whose phase does not apply, exits at once) and looks at the
// device-side count of unfinished problems between chunks of cycles: the call returns with the work complete.
// Sums are deterministic: per-workgroup / per-chunk partials added in index order, independent of the batch.
struct alignas(16) PipeSt {  // per problem, in global memory
  // (the four words every kernel of a cycle tests first: one 16-byte load - pipe_ctl)
  int stage;     // 0, 1, 2: which optimize() of the 5 / 5 / 40 schedule; 3: finished; -1: none opened yet (the call's first cycle opens stage 0)
  int init;      // 1: the next cycle is the computeLambdaInit pass of the stage (no solve, no trial)
  int pend;      // 1: a trial state has been evaluated (solve + trial of this cycle); the next cycle's first kernel judges it
  int adv;       // 1: the stage is over (decided at the head of this cycle): this cycle's kernels run the gate and open the next stage (pipe_adv_*)
  int iters_max; // of this stage
  int it;        // outer iteration of the stage
  int qmax;      // trials of the outer iteration so far
  int cj;        // outer iterations completed in the stage
  int any_point, any_pose, ok;
  int trials, done_iters, stop_seen, it3;
#ifdef GL_PIPE_PROF
  long long dbg[16];
#endif
  int chunk_len;  // entries of a pose's list per wave of the Schur pass: ceil(longest list / nchunk), a multiple of 64
  double lambda, ni, currentChi, chiA, rho;
};
struct PipeA {  // kernel arguments (by value)
  BaK k;
  GmmDev gm;
  int B, P, F, L, NOBS;
  double* poses;
  const uint8_t* prior;
  double* pts;
  const int32_t* assoc;
  const int32_t* optr;
  const int32_t* opose;
  const double* ouvr;
  const int32_t* ooct;
  uint8_t* dropped;
  uint8_t* erase;
  int32_t* iters;
  int32_t* trials_out;
  const int32_t* stop;
  char* scratch;      // | B x 512 B headers | per-problem areas |  (the layout of k_ba_gen)
  size_t per;
  PipeSt* st;         // 2 x B: a cycle of parity `par` reads [par] (the state its predecessor left) in its first kernel, which
                      // writes the judged state to [par ^ 1]; the rest of the cycle works on [par ^ 1]
  int par;
  int cyc;            // index of the cycle (the call's last one is reported through unfinished[1]: the next call's look-ahead)
  double* partA;      // B x nba x 2   {robust chi2, max point diagonal} per workgroup of the point pass
  double* partD;      // B x nba x 2   {scale part, chi2 at the trial state}
  double* partS;      // B x nblk x nchunk x 48
  double* pth;        // B x L x 6: the points' undamped blocks of the last full point pass (point_relambda)
  int* unfinished;    // problems not at stage 3
  int nba, lpp, nblk, nchunk;
  PipeSt* stJ;        // B: the judged state and {accept, next} per window, when the verdict is a kernel of its own (ext_judge: kp_judge)
  int* verd;
  int ext_judge;
  int kper;           // chunks of a block per wave of the Schur pass (1; 2 in batches: the wave's set-up once for two chunks - the
                      // partial sums stay per chunk, so the bits do not depend on it)
  int fuse_asm;       // round 6, calls of a few windows: kp_solve adds the chunks of the blocks itself (no kp_assemble in the cycle)
};
struct PipeCtl {
  int stage, init, pend, adv;
};
// (fetched together: `st->stage >= 3 || st->adv` is two dependent round trips at the top of a 5 - 15 us kernel)
GL_DEV PipeCtl pipe_ctl(const PipeSt* st) {
  const int4 w = *(const int4*)st;
  return PipeCtl{w.x, w.y, w.z, w.w};
}
GL_DEV PipeSt* st_cur(const PipeA& a, int f) { return a.st + (size_t)(a.par ^ 1) * a.B + f; }
GL_DEV const PipeSt* st_prev(const PipeA& a, int f) { return a.st + (size_t)a.par * a.B + f; }
GL_DEV void genp_init(GenP& G, const PipeA& a, int f, int NB, int pb) {
  const int P = a.P, F = a.F, L = a.L, NOBS = a.NOBS, n = 6 * P;
  G.NB = NB;
  G.pb = pb;
  G.P = P;
  G.F = F;
  G.L = L;
  G.poses = a.poses + (size_t)f * (P + F) * 7;
  G.prior = a.prior + (size_t)f * P;
  G.pts = a.pts + (size_t)f * L * 3;
  G.assoc = a.assoc + (size_t)f * L;
  G.optr = a.optr + (size_t)f * (L + 1);
  G.opose = a.opose + (size_t)f * NOBS;
  G.ouvr = a.ouvr + (size_t)f * NOBS * 3;
  G.ooct = a.ooct + (size_t)f * NOBS;
  G.nobs = G.optr[L];
  G.bar = (unsigned*)(a.scratch + (size_t)f * 512);
  G.flagg = (int*)(G.bar + 64);
  G.stop = a.stop;
  G.stop_seen = 0;
  G.done_iters = 0;
  G.trials = 0;
  char* s = a.scratch + (size_t)a.B * 512 + (size_t)f * a.per;
  auto takeD = [&](size_t cnt) {
    double* p = (double*)s;
    s += cnt * 8;
    return p;
  };
  G.Rt = takeD((size_t)(P + F) * 12);
  G.RtN = takeD((size_t)P * 12);
  G.qN = takeD((size_t)P * 7);
  G.pinv = takeD((size_t)P * 7);
  G.pn = takeD((size_t)L * 3);
  G.lin = takeD((size_t)NOBS * 12);
  G.ptw = takeD((size_t)L * 12);
  G.chi_o = takeD((size_t)NOBS);
  G.ld = n + GL_LD_PAD;
  G.S = takeD((size_t)n * G.ld);
  G.part = takeD((size_t)2 * 64 * 4);
  G.toggle = 0;
  G.epoch = 0u;
  G.gv = takeD(n);
  G.bp = takeD(n);
  G.dxv = takeD(n);
  G.pchi = takeD(P);
  G.pchi2 = takeD(P);
  G.prH = takeD((size_t)P * 36);
  G.prb = takeD((size_t)P * 6);
  G.pth = takeD((size_t)L * 6);  // (the persistent kernel's carve; this shape keeps its own in PipeA)
  auto takeI = [&](size_t cnt) {
    int32_t* p = (int32_t*)s;
    s += ((cnt * 4 + 7) / 8) * 8;
    return p;
  };
  G.opoint = takeI(NOBS);
  G.pl_ptr = takeI(P + 1);
  G.pl_obs = takeI(NOBS);
  G.pl_pos = takeI(NOBS);
  G.pl_pt = takeI(NOBS);
  G.plm = takeI((size_t)NOBS * P);
  auto takeB = [&](size_t cnt) {
    uint8_t* p = (uint8_t*)s;
    s += ((cnt + 7) / 8) * 8;
    return p;
  };
  G.lev_o = takeB(NOBS);
  G.lev_g = takeB(L);
  G.pfree = takeB(P + F);
  G.pact = takeB(P);
  G.lact = takeB(L);
}

// prior edges at the current state (once per outer iteration: gen_optimize does the same)
GL_DEV void pipe_priors(const BaK& k, GenP& G, const double* poses) {
  const int NT = blockDim.x;
  for (int j = threadIdx.x; j < G.P; j += NT) {
    double H[36], b[6] = {0, 0, 0, 0, 0, 0}, chi = 0.0;
    for (int r = 0; r < 36; ++r) H[r] = 0.0;
    if (G.pact[j] && G.prior[j] && k.first_as_prior)
      chi = prior_terms(se3_load(G.pinv + (size_t)j * 7), se3_load(poses + (size_t)j * 7), true, H, b);
    for (int r = 0; r < 36; ++r) G.prH[(size_t)j * 36 + r] = H[r];
    for (int r = 0; r < 6; ++r) G.prb[(size_t)j * 6 + r] = b[r];
    G.pchi[j] = chi;
  }
}
GL_DEV bool pipe_stop_now(const PipeSt* st) { return st->stop_seen > 0 || (st->stop_seen < 0 && st->done_iters >= -st->stop_seen); }


// ---- the end of a stage (k_ba_gen's schedule code between its three optimize() calls, :773-879), spread over the kernels of ONE
// cycle so that nothing that walks the points or the observations sits on a single workgroup (one workgroup took 75 k cycles per
// stage change at 12 600 observations, 280 k at 58 600 - 4 - 6 % of the whole call):
//   (A) the head kernel, by point (each workgroup its own points, right after it has accepted them): the fresh error of the
//       degenerate GMM edges after stage 0 (:773-786), the point half of initializeOptimization(0) of the next stage - or, after
//       the last stage, the `dropped` output;
//   (B) the Schur pass's grid, by observation: the reprojection gate after stage 1 (bDoMore, :791-796), the observation half of
//       initializeOptimization(0) (same-value stores) - or the `erase` output;
//   (C) the solve kernel's workgroup: the counts, the loop variables of the next stage, its prior edges - or the scalar outputs.
// A stage that does not run (nothing active: optimize() returns -1; or the stop word) leaves `adv` set and the next cycle does
// the next gate.  ds = the stage that has ended (-1: none yet, the call opens), the same in the three kernels of the cycle.
GL_DEV int pipe_next_stage(const PipeSt& q) { return (q.stage == 1 && pipe_stop_now(&q)) ? 3 : q.stage + 1; }

GL_DEV void pipe_adv_points(const PipeA& a, const GenP& G, const PipeSt& q, int f, int pb) {
  const int ds = q.stage, next = pipe_next_stage(q), per = T_BA / a.lpp, tid = threadIdx.x;
  if (tid < per) {
    for (int l = pb * per + tid; l < G.L; l += a.nba * per) {
      GmmRef g;
      load_gmm(G.assoc[l], a.gm.axis, a.gm.rec12, a.gm.sqrt_info, a.gm.flags, g);
      const bool over = g.has && g.deg && gmm_chi2(a.k, g, G.pts + (size_t)l * 3) > a.k.str_thresh;
      int lg = G.lev_g[l];
      if (ds == 0 && over) {
        lg = 1;
        G.lev_g[l] = 1;
      }
      if (next >= 3) a.dropped[(size_t)f * G.L + l] = over ? 1 : 0;
      else G.lact[l] = (G.assoc[l] >= 0 && !lg) ? 1 : 0;
    }
  }
  if (pb == 0 && next < 3)
    for (int j = tid; j < G.P; j += T_BA) G.pact[j] = (G.pfree[j] && G.prior[j] && a.k.first_as_prior) ? 1 : 0;
}
// (B) `w` of `nw` waves of the problem; the reprojection gate of an observation at the current state: stale chi2 above the threshold
// of its edge type, or the point behind the camera (:799-825, :855-879)
GL_DEV void pipe_adv_obs(const PipeA& a, const GenP& G, const PipeSt& q, int f, int w, int nw) {
  const int ds = q.stage, next = pipe_next_stage(q), lane = threadIdx.x & 63, nobs = G.nobs;
  const bool gate = ds == 1 && !pipe_stop_now(&q);
  uint8_t* er = a.erase + (size_t)f * a.NOBS;
  for (int o = w * 64 + lane; o < nobs; o += nw * 64) {
    const int l = G.opoint[o], j = G.opose[o];
    const double chi = G.chi_o[o], ur = G.ouvr[(size_t)o * 3 + 2];
    int lv = G.lev_o[o];
    const int fr = G.pfree[j];
    const double* Rt = G.Rt + (size_t)j * 12;
    const double* p = G.pts + (size_t)l * 3;
    const double z = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2] + Rt[11];
    const bool bad = chi > (!(ur < 0) ? 7.815 : 5.991) || !(z > 0.0);
    if (gate && bad) {
      lv = 1;
      G.lev_o[o] = 1;  // (the Schur pass of this shape tests the level flags itself: no sweep of the partner table)
    }
    if (next >= 3) {
      er[o] = bad ? 1 : 0;
    } else if (!lv) {
      G.lact[l] = 1;
      if (fr) G.pact[j] = 1;
    }
  }
}
// (C)
GL_DEV void pipe_adv_open(const PipeA& a, GenP& G, PipeSt* st, int f, int* s_cnt) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const PipeSt q = *st;
  const int ds = q.stage, next = pipe_next_stage(q);
  if (next >= 3) {
    if (tid == 0) {
      const int it3 = (ds == 2 && q.it3 != -1) ? q.cj : q.it3;
      st->it3 = it3;
      st->stage = 3;
      st->adv = 0;
      st->pend = 0;
      if (a.iters) a.iters[f] = it3;
      if (a.trials_out) a.trials_out[f] = q.trials;
      atomicMax(a.unfinished + 1, a.cyc + 1);
      atomicSub(a.unfinished, 1);
    }
    return;
  }
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  int np = 0, na = 0;
  for (int l = tid; l < G.L; l += NT) np += G.lact[l] ? 1 : 0;
  for (int j = tid; j < G.P; j += NT) na += G.pact[j] ? 1 : 0;
  if (np) atomicAdd(&s_cnt[0], np);
  if (na) atomicAdd(&s_cnt[1], na);
  __syncthreads();
  const bool any = s_cnt[0] > 0 || s_cnt[1] > 0;
  const bool run = any && !pipe_stop_now(&q);
  if (tid == 0) {
    int it3 = (ds == 2 && q.it3 != -1) ? q.cj : q.it3;
    if (!run && next == 2 && !any) it3 = -1;
    st->it3 = it3;
    st->stage = next;
    st->any_point = s_cnt[0] > 0;
    st->any_pose = s_cnt[1] > 0;
    st->iters_max = next < 2 ? 5 : 40;
    st->it = 0;
    st->qmax = 0;
    st->cj = 0;
    st->init = 1;
    st->pend = 0;
    st->adv = run ? 0 : 1;
    st->rho = 0.0;
  }
  if (run) pipe_priors(a.k, G, G.poses);
}

// ---- set-up: k_ba_gen's, as parallel kernels (one workgroup builds the pose-major lists of 58 000 observations in 4 ms) ----
// (1) per-pose / per-point / per-observation initial state, the partner table cleared
// Workgroup -> (window f, part pb of its nper workgroups) of the per-window kernels, XCD-aware from 8 windows on: consecutive
// workgroup ids go round the 8 XCDs, so window f takes the ids congruent to f modulo 8 - what kp_lin writes for a window (its
// linearised observation records) is then in the L2 of the XCD on which kp_schur gathers it, and kp_trial finds the points there.
// A matter of speed only (the placement is observed, not promised).  Grid: pipe_grid(B, nper) workgroups.
GL_DEV bool pipe_wg(int B, int nper, int& f, int& pb) {
  const int b = (int)blockIdx.x;
  if (B >= 8) {
    const int sl = b >> 3;
    f = (b & 7) + 8 * (sl / nper);
    pb = sl % nper;
  } else {
    f = b / nper;
    pb = b % nper;
  }
  return f < B;
}
static inline int pipe_grid(int B, int nper) { return (B >= 8 ? 8 * ((B + 7) / 8) : B) * nper; }

__global__ __launch_bounds__(T_BA) void kp_setup_init(PipeA a) {
  const int f = blockIdx.x / a.nba, pb = blockIdx.x % a.nba, tid = threadIdx.x, P = a.P, F = a.F, L = a.L;
  GenP G;
  genp_init(G, a, f, a.nba, pb);
  PipeSt* st = st_cur(a, f);  // (the set-up kernels are launched with par = 1: buffer 0, which cycle 0 reads)
  if (pb == 0) {
    if (tid == 0) {
#ifdef GL_PIPE_PROF
      for (int i = 0; i < 16; ++i) st->dbg[i] = 0;
#endif
      const int sw = a.stop ? stop_word_load(a.stop) : 0;
      st->stage = sw > 0 ? 3 : -1;  // `if (pbStopFlag) if (*pbStopFlag) return;` (:765-767): nothing is written
      st->adv = 1;                  // cycle 0 opens stage 0 (initializeOptimization(0): pipe_adv_*)
      st->init = 0;
      st->pend = 0;
      st->cj = 0;
      st->qmax = 0;
      st->it = 0;
      st->iters_max = 0;
      st->any_point = 0;
      st->any_pose = 0;
      st->ok = 1;
      st->rho = 0.0;
      st->chiA = 0.0;
      st->trials = 0;
      st->done_iters = 0;
      st->it3 = 0;
      st->stop_seen = sw;
      st->lambda = 0.0;
      st->ni = 2.0;
      st->currentChi = 0.0;
      if (sw > 0) {
        if (a.iters) a.iters[f] = 0;
        atomicSub(a.unfinished, 1);
      }
    }
    for (int j = tid; j < P + F; j += T_BA) {
      const SE3 T = se3_load(G.poses + (size_t)j * 7);
      store_pose_Rt(T, G.Rt + (size_t)j * 12);
      bool fr = j < P;
      if (fr && G.prior[j] && !a.k.first_as_prior) fr = false;  // vSE3->setFixed(idx_ == 0) (:578-580)
      G.pfree[j] = fr ? 1 : 0;
      if (j < P) se3_store(se3_inverse(T), G.pinv + (size_t)j * 7);  // e->setMeasurement(kf->getTcw())
    }
  }
  for (int l = GSTART; l < L; l += GSTRIDE) {
    G.lev_g[l] = 0;
    for (int o = G.optr[l]; o < G.optr[l + 1]; ++o) {
      G.opoint[o] = l;
      G.lev_o[o] = 0;
      G.chi_o[o] = 0.0;
    }
  }
  for (size_t i = GSTART; i < (size_t)G.nobs * P; i += GSTRIDE) G.plm[i] = -1;
}
// (2) / (4) pose-major CSR of the free poses' observations, ascending within a pose: a stable counting sort - a wave per 512
// observations counts per pose (2), one workgroup scans the counts (3), the same waves scatter (4).  FILL = 0 / 1
constexpr int SORT_SPAN = 512;
template <int FILL>
__global__ __launch_bounds__(T_BA) void kp_setup_lists(PipeA a, int nws) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * NW_BA + (threadIdx.x >> 6);
  const int f = (int)(gw / nws), w = (int)(gw % nws);
  if (f >= a.B || st_cur(a, f)->stage >= 3) return;
  GenP G;
  genp_init(G, a, f, 1, 0);
  const int P = a.P, nobs = G.nobs;
  int* cnt = (int*)a.partS + ((size_t)f * nws + w) * P;  // (the Schur partials are idle during the set-up)
  const int o0 = w * SORT_SPAN, o1 = min(nobs, o0 + SORT_SPAN);
  // the wave's 512 observations (pose, point) fetched ONCE, eight per lane (re-reading the pose of every observation for
  // every pose made this pass 70 us at 58 000 observations); then a ballot per pose and round
  constexpr int NR = SORT_SPAN / 64;
  int ps[NR], pt[NR], start = 0;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int o = o0 + q * 64 + lane;
    ps[q] = o < o1 ? G.opose[o] : -1;
    pt[q] = (FILL && o < o1) ? G.opoint[o] : 0;
  }
  if (FILL && lane < P) start = cnt[lane];  // the list position of this wave's first observation of pose `lane` (from the scan)
  for (int jj = 0; jj < P; ++jj) {
    int run = FILL ? __shfl(start, jj & 63) : 0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const bool hit = ps[q] == jj;
      const unsigned long long m = __ballot(hit);
      if (FILL && hit) {
        const int o = o0 + q * 64 + lane;
        const int e = run + __popcll(m & ((1ull << lane) - 1ull));
        G.pl_obs[e] = o;
        G.pl_pos[o] = e;
        G.pl_pt[e] = pt[q];
      }
      run += __popcll(m);
    }
    if (!FILL && lane == 0) cnt[jj] = run;
  }
}
// (3) counts -> list positions (exclusive scan in (pose, wave) order), pl_ptr, the chunk length of the Schur pass
__global__ __launch_bounds__(T_BA) void kp_setup_scan(PipeA a, int nws) {
  const int f = blockIdx.x, tid = threadIdx.x, P = a.P;
  PipeSt* st = st_cur(a, f);
  if (st->stage >= 3) return;
  GenP G;
  genp_init(G, a, f, 1, 0);
  int* cnt = (int*)a.partS + (size_t)f * nws * P;
  __shared__ int tot[256];
  // a wave per pose: exclusive scan of its column of counts, 64 waves' counts per round (a thread per pose walking its column
  // was a chain of ~120 dependent read-modify-writes)
  for (int j = tid >> 6; j < P; j += NW_BA) {
    const int lane = tid & 63;
    int s = 0;
    for (int w0 = 0; w0 < nws; w0 += 64) {
      const int w = w0 + lane;
      const int c = w < nws ? cnt[(size_t)w * P + j] : 0;
      int x = c;  // inclusive scan over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
      }
      if (w < nws) cnt[(size_t)w * P + j] = s + x - c;
      s += __shfl(x, 63);
    }
    if (lane == 0) tot[j] = s;
  }
  __syncthreads();
  if (tid == 0) {
    int base = 0, longest = 1;
    G.pl_ptr[0] = 0;
    for (int j = 0; j < P; ++j) {
      longest = max(longest, tot[j]);
      const int t = tot[j];
      tot[j] = base;
      base += t;
      G.pl_ptr[j + 1] = base;
    }
    st->chunk_len = (((longest + a.nchunk - 1) / a.nchunk + 63) / 64) * 64;
  }
  __syncthreads();
  for (int i = tid; i < nws * P; i += T_BA) cnt[i] += tot[i % P];
}
// (5) partner table, by list position: entry (e, j2) = the observation of the same point in free pose j2
__global__ __launch_bounds__(T_BA) void kp_setup_partner(PipeA a) {
  const int f = blockIdx.x / a.nba, pb = blockIdx.x % a.nba, P = a.P;
  if (st_cur(a, f)->stage >= 3) return;
  GenP G;
  genp_init(G, a, f, a.nba, pb);
  // a thread per observation of a free pose (a thread per point walked (observations of the point)^2 dependent loads: 130 us
  // at 20 poses)
  for (int o1 = GSTART; o1 < G.nobs; o1 += GSTRIDE) {
    const int l = G.opoint[o1], j1 = G.opose[o1], e = G.pl_pos[o1];
    const int ob = G.optr[l], oe = G.optr[l + 1];
    if (j1 >= P) continue;
    const size_t row = (size_t)e * P;
    for (int o2 = ob; o2 < oe; ++o2) {
      const int j2 = G.opose[o2];
      if (j2 < P) G.plm[row + j2] = o2;
    }
  }
}
// ---- head of a cycle: the trial of the previous cycle is JUDGED (accept / reject, lambda, loop control: the arithmetic of
// SparseOptimizer / OptimizationAlgorithmLevenberg in k_ba_gen's order), then P1: point pass (linearise the observations, point
// blocks); nba workgroups per problem.  Every workgroup judges for itself - the sums it needs are a few hundred doubles, added in
// index order by one thread - and applies the verdict to what it is about to read: its own 64 points (trial -> current on
// acceptance) and an LDS copy of the poses (trial poses of the free key-frames on acceptance).  Workgroup 0 also writes the
// state word of this cycle, the accepted poses and, at the start of an outer iteration, the prior edges.  (A separate
// one-workgroup kernel did this at first: 13 - 23 us per cycle, most of it its own launch floor and the copy of every point.)
GL_DEV void pipe_judge(const PipeA& a, const GenP& G, const PipeSt& in, PipeSt& q, const double* s_d, int* accept, int* next) {
  const int P = a.P, n = 6 * P;
  double scale = 0.0, tempChi = 0.0;
  for (int w = 0; w < a.nba; ++w) {
    scale += s_d[2 * w];
    tempChi += s_d[2 * w + 1];
  }
  const double* dxs = s_d + 2 * a.nba;
  const double* bps = dxs + n;
  const double* pc2 = bps + n;
  q = in;
  const double lambda = q.lambda;
  if (q.qmax == 0) q.currentChi = q.chiA;
  for (int j = 0; j < P; ++j) tempChi += pc2[j];
  for (int i = 0; i < n; ++i) scale += dxs[i] * (lambda * dxs[i] + bps[i]);
  if (!q.ok) tempChi = 1.7976931348623157e308;
  scale += 1e-3;
  const double rho = (q.currentChi - tempChi) / scale;
  const bool acc = rho > 0 && isfinite(tempChi);
  if (acc) {
    const double uu = 2 * rho - 1;
    double alpha = 1. - uu * uu * uu;
    alpha = fmin(alpha, 2. / 3.);
    q.lambda = lambda * fmax(1. / 3., alpha);
    q.ni = 2;
    q.currentChi = tempChi;
  } else {
    q.lambda = lambda * q.ni;
    q.ni *= 2;
  }
  q.rho = rho;
  q.qmax += 1;
  q.trials += 1;
  int nx = 0;  // 0 retry, 1 next outer iteration, 2 stage over
  if (!(rho < 0 && q.qmax < 10 && !(q.stop_seen > 0))) {  // the outer iteration is over
    q.cj += 1;
    q.done_iters += 1;
    q.it += 1;
    nx = (q.qmax == 10 || rho == 0 || q.it >= q.iters_max || pipe_stop_now(&q)) ? 2 : 1;
    q.qmax = 0;
  }
  q.pend = 0;
  q.adv = nx == 2 ? 1 : 0;
  *accept = acc ? 1 : 0;
  *next = nx;
}
constexpr int PIPE_JUDGE_MAX = 2 * 256 + 2 * 132 + 24;  // partials of <= 256 workgroups, dx and b_p of <= 22 poses, their prior chi2
// the verdict on the trial this cycle has evaluated, once per window (ext_judge; see launch_ba_pipe): what the head of the next
// cycle's kp_lin would compute in each of its workgroups
__global__ __launch_bounds__(64) void kp_judge(PipeA a) {
  __shared__ double s_d[PIPE_JUDGE_MAX];
  const int f = blockIdx.x, tid = threadIdx.x, P = a.P, n = 6 * P;
  const PipeSt* st = st_cur(a, f);
  const PipeCtl ctl = pipe_ctl(st);
  if ((ctl.stage >= 3) | !ctl.pend) return;
  GenP G;
  genp_init(G, a, f, 1, 0);
  for (int i = tid; i < 2 * a.nba; i += 64) s_d[i] = a.partD[(size_t)f * a.nba * 2 + i];
  for (int i = tid; i < n; i += 64) {
    s_d[2 * a.nba + i] = G.dxv[i];
    s_d[2 * a.nba + n + i] = G.bp[i];
  }
  if (tid < P) s_d[2 * a.nba + 2 * n + tid] = G.pchi2[tid];
  __syncthreads();
  if (tid == 0) {
    PipeSt q;
    int acc, nx;
    pipe_judge(a, G, *st, q, s_d, &acc, &nx);
    a.stJ[f] = q;
    a.verd[2 * f] = acc;
    a.verd[2 * f + 1] = nx;
  }
}
__global__ __launch_bounds__(T_BA) void kp_lin(PipeA a) {
  __shared__ double red[NW_BA * 32 + 8];
  __shared__ double s_d[PIPE_JUDGE_MAX];
  __shared__ double s_Rt[32 * 12];
  __shared__ PipeSt s_q;
  __shared__ int s_accept, s_next;
  int f, pb;
  if (!pipe_wg(a.B, a.nba, f, pb)) return;
  const int tid = threadIdx.x, P = a.P, F = a.F, n = 6 * P;
  const PipeSt* sp = st_prev(a, f);
  PipeSt* sc = st_cur(a, f);
  const PipeCtl ctl = pipe_ctl(sp);
  if (ctl.stage >= 3) {  // finished: the state word travels on (both buffers must say so)
    if (pb == 0 && tid == 0) *sc = *sp;
    return;
  }
  GenP G;
  genp_init(G, a, f, a.nba, pb);
  const bool judge = ctl.pend != 0;
  const bool lds_ok = 2 * a.nba + 2 * n + P <= PIPE_JUDGE_MAX && P + F <= 32;  // (launch_ba_pipe only takes such windows)
  const bool own_judge = judge && !a.ext_judge;
  if (own_judge && lds_ok) {  // every term of the two sums requested at once
    for (int i = tid; i < 2 * a.nba; i += T_BA) s_d[i] = a.partD[(size_t)f * a.nba * 2 + i];
    for (int i = tid; i < n; i += T_BA) {
      s_d[2 * a.nba + i] = G.dxv[i];
      s_d[2 * a.nba + n + i] = G.bp[i];
    }
    if (tid < P) s_d[2 * a.nba + 2 * n + tid] = G.pchi2[tid];
  }
  // what the verdict will be applied to is requested in the same round trip as the terms it is made from: both versions of
  // the poses (two elements per thread) and the trial state of the thread's point
  PipeSt qin;
  int v_acc = 0, v_next = -1;
  if (tid == 0) {
    if (judge && a.ext_judge) {  // judged behind the trial (kp_judge)
      qin = a.stJ[f];
      v_acc = a.verd[2 * f];
      v_next = a.verd[2 * f + 1];
    } else {
      qin = *sp;
    }
  }
  double rt_cur[2] = {0.0, 0.0}, rt_new[2] = {0.0, 0.0};
  int rt_act[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * T_BA, j = i / 12;
    if (i < (P + F) * 12) {
      rt_cur[q] = G.Rt[i];
      if (j < P) {
        rt_new[q] = G.RtN[i];
        rt_act[q] = G.pact[j];
      }
    }
  }
  const int per = T_BA / a.lpp;  // points of a workgroup per round of the point pass (pass_points: l = GSTART / LPP)
  const int l0 = pb * per + tid;
  const bool has = tid < per && l0 < G.L;
  int la0 = 0;
  double pn0[3] = {0.0, 0.0, 0.0};
  if (has && judge) {
    la0 = G.lact[l0];
#pragma unroll
    for (int i = 0; i < 3; ++i) pn0[i] = G.pn[(size_t)l0 * 3 + i];
  }
  if (tid == 0 && !own_judge) {
    s_q = qin;
    s_accept = v_acc;
    s_next = v_next;
  }
  __syncthreads();
  if (tid == 0 && own_judge) {
    PipeSt q;
    int acc, nx;
    pipe_judge(a, G, qin, q, s_d, &acc, &nx);
    s_q = q;
    s_accept = acc;
    s_next = nx;
  }
  __syncthreads();
  const bool accept = s_accept != 0;
  const int next = s_next;
  // the poses this workgroup linearises at: LDS copy, trial poses of the active free key-frames on acceptance
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * T_BA;
    if (i < (P + F) * 12) s_Rt[i] = (accept && rt_act[q]) ? rt_new[q] : rt_cur[q];
  }
  // its points: trial -> current
  if (accept) {
    if (has && la0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) G.pts[(size_t)l0 * 3 + i] = pn0[i];
    }
    for (int l = l0 + a.nba * per; l < G.L; l += a.nba * per) {  // (further rounds: none with nba = ceil(L lpp / T_BA))
      if (tid >= per) break;
      if (!G.lact[l]) continue;
#pragma unroll
      for (int i = 0; i < 3; ++i) G.pts[(size_t)l * 3 + i] = G.pn[(size_t)l * 3 + i];
    }
  }
  if (pb == 0) {
    if (accept) {
      for (int j = tid; j < P; j += T_BA) {
        if (!G.pact[j]) continue;
        for (int r = 0; r < 7; ++r) G.poses[(size_t)j * 7 + r] = G.qN[(size_t)j * 7 + r];
        for (int r = 0; r < 12; ++r) G.Rt[(size_t)j * 12 + r] = G.RtN[(size_t)j * 12 + r];
      }
    }
    // the state the next outer iteration linearises the prior edges at (k_ba_gen: once per outer iteration)
    if (next == 1) pipe_priors(a.k, G, accept ? G.qN : G.poses);
    if (tid == 0) *sc = s_q;
  }
  __syncthreads();
  if (s_q.adv) {  // the stage is over (or none is open yet): this cycle's kernels do the gate and open the next one
    pipe_adv_points(a, G, s_q, f, pb);
    return;
  }
  double* pth = a.pth + (size_t)f * a.L * 6;
  // Nothing has moved since the last point pass (the trial was rejected, or this is the first trial after the stage's lambda-init
  // pass): only the damping is new - 20 of the 33 trials of the 8 + 4 test window
  // (not judged, not a lambda-init pass, stage open: the cycle after the lambda-init pass, whose solve kernel clears `init`)
  if (!s_q.init && !accept) {
    point_relambda(G, s_q.lambda, pth);
    return;
  }
  G.Rt = s_Rt;
  double acc[32], md = 0.0;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  acc[0] = pass_points_pipe(a.lpp, a.k, a.gm, G, s_q.stage < 2, s_q.init ? 0.0 : s_q.lambda, md, pth);
  block_reduce<1, NW_BA>(acc, red);
  md = block_max(md, red);
  if (threadIdx.x == 0) {
    double* pa = a.partA + ((size_t)f * a.nba + pb) * 2;
    pa[0] = acc[0];
    pa[1] = md;
  }
}

// ---- P2: one wave per (block (j1 <= j2) of the reduced camera system, chunk of pose j1's list: 1 / nchunk of the longest) ----------
__global__ __launch_bounds__(T_BA, 2) void kp_schur(PipeA a) {
  const int lane = threadIdx.x & 63;
  // Workgroup -> (window, four of its waves) through the XCD-aware map (pipe_wg): an XCD then works on one or two windows at a time and
  // the records its blocks share (every observation record is read by up to P blocks) stay in ITS 4 MB L2 instead of streaming
  // through all eight: 64 windows 0.191 -> 0.175 ms per window.  per_prob is a multiple of 4 (nchunk is).
  const int per_prob = a.nblk * a.nchunk / a.kper;  // waves of a window
  int f, wq;
  if (!pipe_wg(a.B, per_prob / NW_BA, f, wq)) return;
  // (the wave index as a SCALAR: block, chunk and the two poses' {R, t} - scalar loads - are then SGPR values, not 48 VGPRs)
  const int w = wq * NW_BA + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const PipeSt* st = st_cur(a, f);
  GenP G;
  genp_init(G, a, f, 1, 0);
  const int P = a.P, wpb = a.nchunk / a.kper, b = w / wpb, chunk0 = (w % wpb) * a.kper;
  int j1 = 0, rem = b;
  while (rem >= P - j1) {
    rem -= P - j1;
    ++j1;
  }
  const int j2 = j1 + rem;
  // A wave of this kernel lives for two entries per lane, so its life is the LENGTH OF ITS CHAIN OF DEPENDENT LOADS (the ISA of the
  // first version: eight scalar round trips one after the other before the loop, five to six vector ones per entry inside it - state
  // word, then the lists' bounds, then the partner, then the own observation, then the two level flags, then the point, then the
  // records piece by piece as the block algebra asked for them).  Here: everything that depends on the launch arguments alone in ONE
  // round trip - the state word, the chunk length, the list's bounds, the two poses (scalar loads; all of it written by earlier
  // kernels), the two activity flags - then per entry one round trip for its three indices (requested one entry ahead) and one for
  // the level flags, both observation records and the point's block together.
  typedef const int __attribute__((address_space(4))) cint_k;
  const cint_k* stw = (const cint_k*)st;
  const int c_stage = stw[0], c_init = stw[1], c_adv = stw[3];
  const int chunk_len = *(const cint_k*)&st->chunk_len;
  const int p_beg = ((const cint_k*)G.pl_ptr)[j1], p_end = ((const cint_k*)G.pl_ptr)[j1 + 1];
  double R1[9], t1[3], R2[9], t2[3];
  load_Rt_k((const cdouble_k*)(G.Rt + (size_t)j1 * 12), R1, t1);  // (written by an earlier kernel of the cycle, read-only here)
  load_Rt_k((const cdouble_k*)(G.Rt + (size_t)j2 * 12), R2, t2);
  const int pa1 = G.pact[j1], pa2 = G.pact[j2];
  struct Idx {
    int o1, o2, l;
  };
  // (requested before the state word is tested: a window that finished before its set-up - the stop word - has no lists, and
  // whatever its bounds say the addresses stay inside the window's arrays)
  auto fetch_idx = [&](int e, int e_end, Idx& x) {
    x.o1 = 0;
    x.o2 = -1;
    x.l = 0;
    if (e < e_end && e >= 0 && e < a.NOBS) {
      x.o2 = G.plm[(size_t)e * P + j2];
      x.o1 = G.pl_obs[e];
      x.l = G.pl_pt[e];
    }
  };
  Idx cur;
  {
    const int e0 = p_beg + chunk0 * chunk_len;
    fetch_idx(e0 + lane, min(p_end, e0 + chunk_len), cur);  // (the bounds are scalars of the same round trip: the first entry's indices follow it at once)
  }
  asm volatile("" ::"s"(c_stage), "s"(c_init), "s"(c_adv));
  pin(pa1);
  pin(pa2);
  if (c_stage >= 3) return;
  const bool schur = !c_init;
  if (c_adv) {
    pipe_adv_obs(a, G, *st, f, w, per_prob);
    return;
  }
  const bool act = pa1 && pa2 && (schur || j1 == j2);
  double* const pblk = a.partS + ((size_t)f * a.nblk + b) * a.nchunk * 48;
  for (int sc = 0; sc < a.kper; ++sc) {
    const int chunk = chunk0 + sc;
    const int e0 = p_beg + chunk * chunk_len, e1 = min(p_end, e0 + chunk_len);
    double v1[32], v2[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) v1[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v2[i] = 0.0;
    if (act) {
      for (int e = e0 + lane; e < e1; e += 64) {
        Idx nxt;
        fetch_idx(e + 64, e1, nxt);
        const int o1 = cur.o1, o2 = cur.o2, l = cur.l;
        int lv = 0;
        double r1[12], r2[12], pw[9];
#pragma unroll
        for (int i = 0; i < 12; ++i) r1[i] = r2[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) pw[i] = 0.0;
        if (o2 >= 0) {  // (the lanes without a partner - half of them - request nothing: at 256 windows the pass moves terabytes per second)
          lv = G.lev_o[o1] | G.lev_o[o2];  // an edge at level 1 (k_ba_gen folds these flags into the partner table)
#pragma unroll
          for (int i = 0; i < 12; ++i) r1[i] = G.lin[(size_t)o1 * 12 + i];
#pragma unroll
          for (int i = 0; i < 12; ++i) r2[i] = G.lin[(size_t)o2 * 12 + i];
#pragma unroll
          for (int i = 0; i < 9; ++i) pw[i] = G.ptw[(size_t)l * 12 + i];
        }
        pin(lv);
#pragma unroll
        for (int i = 0; i < 12; ++i) pin(r1[i]);
#pragma unroll
        for (int i = 0; i < 12; ++i) pin(r2[i]);
#pragma unroll
        for (int i = 0; i < 9; ++i) pin(pw[i]);
        cur = nxt;
        if (o2 < 0 || lv) continue;
        schur_entry(r1, r2, pw, R1, R2, schur, j1 == j2, v1, v2);
      }
    }
    if (sc + 1 < a.kper) {  // the next chunk's first indices travel during the reduction
      const int n0 = e0 + chunk_len;
      fetch_idx(n0 + lane, min(p_end, n0 + chunk_len), cur);
    }
    double* const ps = pblk + (size_t)chunk * 48;  // 48 sums of this chunk: [0..35] the block, [36..41] g, [42..47] b_p (diagonal blocks)
    if (!act || e0 >= e1) {  // nothing added (a chunk beyond the end of this pose's list): the sums of zeros
      if (lane < 48) ps[lane] = 0.0;
      continue;
    }
    const double s1 = wave_reduce_scatter32(v1), s2 = wave_reduce_scatter16(v2);  // (the 16 further sums: same tree, same bits)
    if (wave_slot_owner(lane)) ps[wave_slot(lane)] = s1;
    if (wave_slot16_owner(lane)) ps[32 + wave_slot16(lane)] = s2;
  }
}

// ---- P2b: the chunks of a block added in chunk order -> the assembled system (a thread per (block, sum); the solve kernel
// read nblk x nchunk x 48 partials itself at first - 63 us of its 155 at 20 poses; letting the last wave of a block do it
// behind device-scope fences made the Schur pass 3.5 x slower) ------------------------------------------------------------
// one sum of one block of window f: its chunks added in chunk order, the total to its place in the assembled system
GL_DEV void assemble_sum(const PipeA& a, const GenP& G, int f, int b, int sidx) {
  const int P = a.P, ld = 6 * P + GL_LD_PAD;
  int j1 = 0, rem = b;
  while (rem >= P - j1) {
    rem -= P - j1;
    ++j1;
  }
  const int j2 = j1 + rem;
  const double* ps = a.partS + ((size_t)f * a.nblk + b) * a.nchunk * 48 + sidx;
  double x[16];  // all chunks requested before the first add (the adds stay in chunk order); nchunk is 4 or 16
#pragma unroll
  for (int c = 0; c < 16; ++c) x[c] = c < a.nchunk ? ps[(size_t)c * 48] : 0.0;
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) v += x[c];
  if (sidx < 36) {  // only the lower triangle of S is read: block (j2, j1) = block (j1, j2)^T
    const int r = sidx / 6, c = sidx % 6;
    G.S[j1 == j2 ? (size_t)(6 * j1 + r) * ld + 6 * j1 + c : (size_t)(6 * j2 + c) * ld + 6 * j1 + r] = v;
  } else if (j1 == j2) {
    if (sidx < 42) G.gv[6 * j1 + (sidx - 36)] = v;
    else G.bp[6 * j1 + (sidx - 42)] = v;
  }
}
__global__ __launch_bounds__(T_BA) void kp_assemble(PipeA a) {
  const long g = (long)blockIdx.x * T_BA + threadIdx.x;
  const int per_prob = a.nblk * 48;
  const int f = (int)(g / per_prob), r0 = (int)(g % per_prob), b = r0 / 48, sidx = r0 % 48;
  if (f >= a.B) return;
  const PipeCtl ctl = pipe_ctl(st_cur(a, f));
  if ((ctl.stage >= 3) | ctl.adv) return;
  GenP G;
  genp_init(G, a, f, 1, 0);
  assemble_sum(a, G, f, b, sidx);
}

// ---- solve: (lambda init |) LDL^T of the assembled system, trial poses; at the end of a stage the gate / the next stage's
// opening instead; one workgroup per problem.  The factorisation loads its register tiles straight from the assembled system
// in global memory and adds the prior information / lambda / the unit diagonal of inactive poses on the way (diag_terms, as in
// the persistent kernel): a copy into LDS with those terms added there cost 10 - 60 k cycles of the kernel's 65 - 210 k.
constexpr int T_SOLVE = 512;  // two teams of four waves in the factorisation (ldlt_solve_teams)
__global__ __launch_bounds__(T_SOLVE) void kp_solve(PipeA a) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  __shared__ double red[NW_BA * 32 + 128 + 128];
  __shared__ double s_part[2 * 256 + 24];
  __shared__ double s_prH[22 * 36];  // prior information and the pose flags the factorisation adds / tests while it loads its tiles
  __shared__ uint8_t s_pact[32], s_prior[32];
  __shared__ int s_flag;
  __shared__ int s_cnt[2];
  const int f = blockIdx.x, tid = threadIdx.x, P = a.P, n = 6 * P;
  PipeSt* st = st_cur(a, f);
  const PipeCtl ctl = pipe_ctl(st);
  const double lambda = st->lambda;
  const int any_pose = st->any_pose;
  if (ctl.stage >= 3) return;
  GenP G;
  genp_init(G, a, f, 1, 0);
  if (ctl.adv) {  // the stage ended at the head of this cycle: gate, next stage (or the outputs)
    pipe_adv_open(a, G, st, f, s_cnt);
    return;
  }
  if (a.fuse_asm) {  // (a call of a few windows: the 5 us of a kernel of its own for 1 - 2 us of adds; the same sums in the same order)
    for (int r0 = tid; r0 < a.nblk * 48; r0 += T_SOLVE) assemble_sum(a, G, f, r0 / 48, r0 % 48);
    __syncthreads();
  }
#ifdef GL_PIPE_PROF
  const long long s0 = clock64();
#endif
  const int ld = G.ld;
  const bool small = n <= 128;
  const bool init = ctl.init != 0;
  // partial sums of the point pass, the prior chi2 and the reduced right-hand side: requested together, added in index order
  for (int i = tid; i < 2 * a.nba; i += T_SOLVE) s_part[i] = a.partA[(size_t)f * a.nba * 2 + i];
  if (tid < P) s_part[2 * a.nba + tid] = G.pchi[tid];
  const bool stage_lds = P <= 22;
  if (stage_lds) {
    for (int i = tid; i < P * 36; i += T_SOLVE) s_prH[i] = G.prH[i];
    if (tid < P) {
      s_pact[tid] = G.pact[tid];
      s_prior[tid] = G.prior[tid];
    }
  }
  double mine = 0.0;
  if (init) {
    for (int i = tid; i < n; i += T_SOLVE) {
      const int j = i / 6, r = i - 6 * j;
      if (G.pact[j]) mine = fabs(G.S[(size_t)i * ld + i] + G.prH[(size_t)j * 36 + r * 6 + r]);
    }
  } else {
    for (int i = tid; i < n; i += T_SOLVE) {
      const int j = i / 6;
      double v = G.gv[i];
      if (G.pact[j] && G.prior[j] && a.k.first_as_prior) {
        const double b = G.prb[i];
        v += b;
        G.bp[i] += b;
      }
      G.dxv[i] = v;
      if (small) red[128 + i] = v;
    }
  }
  __syncthreads();
  double chiA = 0.0, md = 0.0;
  for (int w = 0; w < a.nba; ++w) {
    chiA += s_part[2 * w];
    md = fmax(md, s_part[2 * w + 1]);
  }
#ifdef GL_PIPE_PROF
  const long long s1 = clock64();
#endif
  if (init) {  // computeLambdaInit: 1e-5 x the largest diagonal of H_pp (+ prior) and H_ll
    md = fmax(md, block_max(mine, red));
    if (tid == 0) {
      st->lambda = 1e-5 * md;
      st->ni = 2.0;
      st->init = 0;
      st->pend = 0;
    }
    return;
  }
  for (int j = 0; j < P; ++j) chiA += s_part[2 * a.nba + j];
#ifdef GL_PIPE_PROF
  const long long s2 = clock64();
#endif
  bool ok = true;
  if (any_pose) {
    if (small) {
      GenP GL = G;
      if (stage_lds) {
        GL.prH = s_prH;
        GL.pact = s_pact;
        GL.prior = s_prior;
      }
      lds_double* W = (lds_double*)dyn_lds;
      ok = n <= 32   ? ldlt_solve_teams<2>(GL, a.k, lambda, G.S, ld, W, ld, G.dxv, n, &s_flag, red)
           : n <= 48 ? ldlt_solve_teams<3>(GL, a.k, lambda, G.S, ld, W, ld, G.dxv, n, &s_flag, red)
           : n <= 80 ? ldlt_solve_teams<5>(GL, a.k, lambda, G.S, ld, W, ld, G.dxv, n, &s_flag, red)
                     : ldlt_solve_teams<8>(GL, a.k, lambda, G.S, ld, W, ld, G.dxv, n, &s_flag, red);
    } else {  // the scalar-pivot path factorises in global memory: the diagonal terms go in first
      for (int j = tid; j < P; j += T_SOLVE) {
        if (!G.pact[j]) {
          for (int r = 0; r < 6; ++r) G.S[(size_t)(6 * j + r) * ld + 6 * j + r] = 1.0;
          continue;
        }
        if (G.prior[j] && a.k.first_as_prior) {
          const double* H = G.prH + (size_t)j * 36;
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) G.S[(size_t)(6 * j + r) * ld + 6 * j + c] += H[r * 6 + c];
        }
        for (int r = 0; r < 6; ++r) G.S[(size_t)(6 * j + r) * ld + 6 * j + r] += lambda;
      }
      __syncthreads();
      ok = ldlt_solve_large(G.S, G.dxv, n, ld, &s_flag);
    }
  }
  __syncthreads();
#ifdef GL_PIPE_PROF
  const long long s3 = clock64();
#endif
  for (int j = tid; j < P; j += T_SOLVE) {  // trial poses
    const SE3 T = se3_load(G.poses + (size_t)j * 7);
    SE3 Tn = T;
    if (G.pact[j] && ok) {
      double dx[6];
      for (int r = 0; r < 6; ++r) dx[r] = G.dxv[6 * j + r];
      Tn = se3_mul(se3_exp(dx), T);
    } else {
      for (int r = 0; r < 6; ++r) G.dxv[6 * j + r] = 0.0;
    }
    se3_store(Tn, G.qN + (size_t)j * 7);
    store_pose_Rt(Tn, G.RtN + (size_t)j * 12);
    G.pchi2[j] = (G.pact[j] && G.prior[j] && a.k.first_as_prior)
                     ? prior_terms(se3_load(G.pinv + (size_t)j * 7), Tn, false, nullptr, nullptr)
                     : 0.0;
  }
  if (tid == 0) {
    st->ok = ok ? 1 : 0;
    st->chiA = chiA;
    st->pend = 1;
    if (a.stop) st->stop_seen = stop_word_load(a.stop);
  }
#ifdef GL_PIPE_PROF
  __syncthreads();
  if (tid == 0 && f == 0) {
    const long long s4 = clock64();
    st->dbg[8] += s1 - s0; st->dbg[9] += s2 - s1; st->dbg[10] += s3 - s2; st->dbg[11] += s4 - s3; st->dbg[12] += 1;
    printf("solve cycles (cumulative over the run): fetch %lld  right-hand side %lld  factorisation + substitutions %lld  trial poses %lld  calls %lld\n", st->dbg[8], st->dbg[9], st->dbg[10],
           st->dbg[11], st->dbg[12]);
  }
#endif
}

// ---- P3: back-substitution of the points, trial points, their chi2 -------------------------------------------------
__global__ __launch_bounds__(T_BA) void kp_trial(PipeA a) {
  __shared__ double red[NW_BA * 32 + 8];
  int f, pb;
  if (!pipe_wg(a.B, a.nba, f, pb)) return;
  const PipeSt* st = st_cur(a, f);
  const PipeCtl ctl = pipe_ctl(st);
  const double lambda = st->lambda;
  if ((ctl.stage >= 3) | !ctl.pend) return;
  GenP G;
  genp_init(G, a, f, a.nba, pb);
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  pass_trial_pipe(a.lpp, a.k, a.gm, G, ctl.stage < 2, lambda, a.P, acc);
  block_reduce<2, NW_BA>(acc, red);
  if (threadIdx.x == 0) {
    double* pd = a.partD + ((size_t)f * a.nba + pb) * 2;
    pd[0] = acc[0];
    pd[1] = acc[1];
  }
}

namespace gl {
size_t ba_gen_scratch_bytes(int B, int P, int F, int L, int NOBS) {
  const size_t per = ((gen_scratch_bytes(P, F, L, NOBS) + 255) / 256) * 256;
  return (size_t)B * 512 + per * B;
}

// the launch proper; scratch: ba_gen_scratch_bytes() bytes of the context's scratch block
// the persistent kernel proper and the launch that writes its address table (k_ba_gen_t<true>: same arguments, one wave per window)
static const auto k_ba_gen = &k_ba_gen_t<false>;
int launch_ba_gen(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int P, int F, int L, int NOBS,
                  double* poses_dev, const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                  const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev, const double* obs_uvr_dev,
                  const int32_t* obs_oct_dev, uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                  const int32_t* stop_dev, void* scratch, int force_nb = 0) {
  const size_t per = ((gen_scratch_bytes(P, F, L, NOBS) + 255) / 256) * 256;
  GmmDev gm{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags, g->plane4};
  const size_t n = 6 * (size_t)P;
  const size_t s_bytes = n * (n + GL_LD_PAD) * sizeof(double);
  int s_in_lds = s_bytes <= 136 * 1024 ? 1 : 0;  // 6 P <= 126: every system the register-tile solve takes
  const size_t lds = s_in_lds ? s_bytes : 0;
  if (s_in_lds)
    GL_HIP(gl::ensure_dynamic_lds(c, (const void*)k_ba_gen, s_bytes));
  // workgroups per problem: as many as stay co-resident (the per-problem barrier needs that; the
  // cooperative launch enforces it), at most 64; large batches run one workgroup per problem
  int NB = 1, bsub = B;
  {
    // the occupancy query is a driver call: the context (one device, one host thread) remembers it per LDS size
    auto key = std::make_pair((const void*)k_ba_gen, lds);
    auto hit = c->occupancy.find(key);
    if (hit == c->occupancy.end()) {
      int occ_val = 0;
      GL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_val, (const void*)k_ba_gen, T_BA, lds));
      hit = c->occupancy.emplace(key, occ_val).first;
    }
    const int occ = hit->second, ncu = c->ncu;
    const long cap = (long)occ * ncu;
    // measured optimum on single problems (tools/ba_nb.py), with up to 4 lanes per point in the
    // point passes and up to 4 waves per reduced-camera block: one workgroup up to ~600 observations (no
    // cross-workgroup barrier at all), 8 up to 2 000, 16 up to 6 000, 32 up to 16 000, 64 above; NOBS (the
    // stride) stands in for the observation count
    const int want = NOBS <= 600 ? 1 : NOBS <= 2000 ? 8 : NOBS <= 6000 ? 16 : NOBS <= 16000 ? 32 : 64;
    // The workgroup count of a problem decides the order of its partial sums, i.e. its bits: it is a function of the WINDOW
    // (its observation stride) alone, never of how many windows ride in the call - a batch too large to keep B x NB workgroups
    // co-resident is launched in sub-batches of cap / NB windows, one after the other on the stream (round 3 shrank NB instead:
    // a window's bits then depended on the batch size).
    NB = std::min(want, (int)std::min<long>(cap, 64));
    if (c->opt.bagen_nb > 0) NB = (int)std::min<long>(std::max(1, (int)c->opt.bagen_nb), std::min<long>(cap, 64));  // knob: that many workgroups per problem (tests)
    // bagen_mode 3 (opt-in, throughput of large batches): the whole batch in ONE launch, with as many workgroups per window as
    // stay co-resident - the round-3 rule; a window's bits then depend on the batch size
    if (c->opt.bagen_mode == 3) NB = (int)std::min<long>(NB, cap / B);
    if (force_nb > 0) NB = force_nb;  // (the packed route of gl_track_frames_anchored: one workgroup per frame, whatever the batch)
    if (NB < 2) NB = 1;
    bsub = NB > 1 ? (int)std::max<long>(1, cap / NB) : B;
  }
  {
    gl::TimerScope ts(c, GL_TIMER_BA);
    BaK kk = make_bak(cam, prm, -1.0);
    int32_t* stats_all = (c->stats && c->stats_n >= B) ? c->stats : nullptr;  // gl_ctx_set_stats_buffer: trials per problem
    size_t per_v = per;
    for (int b0 = 0, Bs = 0; b0 < B; b0 += Bs) {
      Bs = std::min(bsub, B - b0);
      // the sub-batch's slices of the caller's arrays (strides: gmmloc_hip.h); the scratch is re-used, launches are in stream order
      double* poses_s = poses_dev + (size_t)b0 * (P + F) * 7;
      const uint8_t* prior_s = prior_dev + (size_t)b0 * P;
      double* points_s = points_dev + (size_t)b0 * L * 3;
      const int32_t* assoc_s = assoc_dev + (size_t)b0 * L;
      const int32_t* optr_s = obs_ptr_dev + (size_t)b0 * (L + 1);
      const int32_t* opose_s = obs_pose_dev + (size_t)b0 * NOBS;
      const double* ouvr_s = obs_uvr_dev + (size_t)b0 * NOBS * 3;
      const int32_t* ooct_s = obs_oct_dev + (size_t)b0 * NOBS;
      uint8_t* dropped_s = assoc_dropped_dev + (size_t)b0 * L;
      uint8_t* erase_s = obs_erase_dev + (size_t)b0 * NOBS;
      int32_t* iters_s = iters_dev ? iters_dev + b0 : nullptr;
      int32_t* stats = stats_all ? stats_all + b0 : nullptr;
      char* scr = (char*)scratch;
      GL_HIP(hipMemsetAsync(scratch, 0, (size_t)Bs * 512, c->stream));
      bool launched = false;
      // A window's bits are a function of NB (the order of its partial sums): when the cooperative launch is refused - the
      // workgroups are not co-resident after all, another context holds CUs - the sub-batch is HALVED at the same NB until it
      // fits.  Only when a single window does not fit does the call fall back to one workgroup per problem (other bits, the same
      // arithmetic): GL_COUNTER_BA_COOP_FALLBACK counts those windows.
      auto write_table = [&](int bs) {  // (the table's addresses depend on the sub-batch size: again for every size that is tried)
        k_ba_gen_t<true><<<bs, 64, 0, c->stream>>>(kk, gm, bs, 1, P, F, L, NOBS, poses_s, prior_s, points_s, assoc_s, optr_s, opose_s, ouvr_s, ooct_s,
                                                   dropped_s, erase_s, iters_s, scr, per_v, s_in_lds, stop_dev, stats);
      };
      while (NB > 1 && !launched) {
        void* args[] = {&kk, &gm, &Bs, &NB, &P, &F, &L, &NOBS, &poses_s, &prior_s, &points_s, &assoc_s, &optr_s,
                        &opose_s, &ouvr_s, &ooct_s, &dropped_s, &erase_s, &iters_s, &scr, &per_v,
                        &s_in_lds, &stop_dev, &stats};
        write_table(Bs);
        hipError_t e = hipLaunchCooperativeKernel((const void*)k_ba_gen, dim3(Bs * NB), dim3(T_BA), args, lds, c->stream);
        launched = e == hipSuccess;
        if (!launched) {
          (void)hipGetLastError();
          if (Bs == 1) break;
          Bs = std::max(1, Bs / 2);
        }
      }
      if (!launched) {
        if (NB > 1) c->coop_fallbacks += Bs;
        write_table(Bs);
        k_ba_gen<<<Bs, T_BA, lds, c->stream>>>(kk, gm, Bs, 1, P, F, L, NOBS, poses_s, prior_s, points_s, assoc_s, optr_s,
                                               opose_s, ouvr_s, ooct_s, dropped_s, erase_s, iters_s,
                                               scr, per_v, s_in_lds, stop_dev, stats);
      }
    }
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// scratch of the pipelined shape: k_ba_gen's layout + the state words and the partial sums
static void pipe_shape(int P, int L, int NOBS, int* nba, int* lpp, int* nblk, int* nchunk) {
  *lpp = NOBS >= 6 * L ? 8 : 4;  // lanes per point in the point passes
  *nba = std::max(1, (L * *lpp + T_BA - 1) / T_BA);
  *nblk = P * (P + 1) / 2;
  // waves per block of the Schur pass (each takes 1 / nchunk of the longest pose list); at 20 poses 12 (2 520 waves of 4 rounds)
  // is 8 % faster than 16 (3 360 of 3: a second wave of workgroups behind the first)
  *nchunk = NOBS <= 1024 ? 4 : NOBS <= 40000 ? 16 : 12;
}
size_t ba_pipe_scratch_bytes(int B, int P, int F, int L, int NOBS) {
  int nba, lpp, nblk, nchunk;
  pipe_shape(P, L, NOBS, &nba, &lpp, &nblk, &nchunk);
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  return ba_gen_scratch_bytes(B, P, F, L, NOBS) + up((size_t)2 * B * sizeof(PipeSt)) + 2 * up((size_t)B * nba * 16) +
         up((size_t)B * nblk * nchunk * 48 * 8) + up((size_t)B * L * 6 * 8) + up((size_t)B * sizeof(PipeSt)) + up((size_t)B * 8) + up((size_t)B * nblk * 4) + 1024;
}

// One LANE of the pipelined local BA: a sub-batch of the call's windows with its own kernel arguments, scratch area and stream.
struct PipeLane {
  PipeA a;
  hipStream_t s = nullptr;
  int B = 0, schur_blocks = 0, par = 0, first = 0;
  size_t s_bytes = 0;
  int* hw = nullptr;
  bool done = false;
};
static int pipe_lane_setup(Ctx* c, PipeLane& ln, int stats_off, bool stats_ok, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int P, int F, int L, int NOBS,
                   double* poses_dev, const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                   const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev, const double* obs_uvr_dev,
                   const int32_t* obs_oct_dev, uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                   const int32_t* stop_dev, void* scratch) {
  PipeA& a = ln.a;
  a.k = make_bak(cam, prm, -1.0);
  a.gm = GmmDev{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags, g->plane4};
  a.B = B;
  a.P = P;
  a.F = F;
  a.L = L;
  a.NOBS = NOBS;
  a.poses = poses_dev;
  a.prior = prior_dev;
  a.pts = points_dev;
  a.assoc = assoc_dev;
  a.optr = obs_ptr_dev;
  a.opose = obs_pose_dev;
  a.ouvr = obs_uvr_dev;
  a.ooct = obs_oct_dev;
  a.dropped = assoc_dropped_dev;
  a.erase = obs_erase_dev;
  a.iters = iters_dev;
  a.trials_out = stats_ok ? c->stats + stats_off : nullptr;  // (checked once for the whole call: all lanes or none)
  a.stop = stop_dev;
  a.scratch = (char*)scratch;
  a.per = ((gen_scratch_bytes(P, F, L, NOBS) + 255) / 256) * 256;
  pipe_shape(P, L, NOBS, &a.nba, &a.lpp, &a.nblk, &a.nchunk);
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  char* s = (char*)scratch + ba_gen_scratch_bytes(B, P, F, L, NOBS);
  a.st = (PipeSt*)s;
  s += up((size_t)2 * B * sizeof(PipeSt));
  a.partA = (double*)s;
  s += up((size_t)B * a.nba * 16);
  a.partD = (double*)s;
  s += up((size_t)B * a.nba * 16);
  a.partS = (double*)s;
  s += up((size_t)B * a.nblk * a.nchunk * 48 * 8);
  a.pth = (double*)s;
  s += up((size_t)B * L * 6 * 8);
  a.stJ = (PipeSt*)s;
  s += up((size_t)B * sizeof(PipeSt));
  a.verd = (int*)s;
  s += up((size_t)B * 8);
  // The verdict on a trial as a kernel of its own behind kp_trial (one small workgroup per window) instead of in every one of the
  // window's 24 - 47 point-pass workgroups: in batches the judging head is 14 of kp_lin's 28 us (64 windows); alone it is a launch
  // more on a single window's critical path (break-even at 16 - 24 windows per lane: from 32 on; option pipe_judge = 0 / 1 overrides,
  // for A/B runs).  The same function on the same values: the same bits.  64 windows 0.122 -> 0.114 ms per window, 256 0.118 -> 0.113.
  a.ext_judge = c->opt.pipe_judge >= 0 ? (c->opt.pipe_judge != 0 ? 1 : 0) : (B >= 32 ? 1 : 0);
  a.unfinished = (int*)s;  // [0] problems not finished, [1] cycles the slowest of them needed
  const size_t n = 6 * (size_t)P;
  ln.s_bytes = n <= 128 ? n * (n + GL_LD_PAD) * sizeof(double) : 0;
  if (ln.s_bytes) GL_HIP(ensure_dynamic_lds(c, (const void*)kp_solve, ln.s_bytes));
  // (ln.hw: page-locked words of this lane - [0..1] the device's count of unfinished problems and the cycles of the slowest as copied
  // back between chunks of cycles, [2..3] their initial values)
  ln.hw[2] = B;
  ln.hw[3] = 0;
  GL_HIP(hipMemsetAsync(scratch, 0, (size_t)B * 512, ln.s));
  GL_HIP(hipMemcpyAsync(a.unfinished, ln.hw + 2, 2 * sizeof(int), hipMemcpyHostToDevice, ln.s));
  {
    const int nws = std::max(1, (NOBS + SORT_SPAN - 1) / SORT_SPAN);
    const int lblocks = (int)(((long)B * nws + NW_BA - 1) / NW_BA);
    a.par = 1;  // the set-up writes state buffer 0
    kp_setup_init<<<B * a.nba, T_BA, 0, ln.s>>>(a);
    kp_setup_lists<0><<<lblocks, T_BA, 0, ln.s>>>(a, nws);
    kp_setup_scan<<<B, T_BA, 0, ln.s>>>(a, nws);
    kp_setup_lists<1><<<lblocks, T_BA, 0, ln.s>>>(a, nws);
    kp_setup_partner<<<B * a.nba, T_BA, 0, ln.s>>>(a);
  }
  // Chunks of a block per wave of the Schur pass: a wave's set-up (its chain of scalar loads, the poses) is a third of its life at
  // one chunk - two entries per lane - so in batches a wave takes 2, 4 or 8 chunks one after the other, as many as leave the call
  // one full round of waves on the chip (256 CUs x 8).  The partial sums stay per chunk: the bits do not depend on it
  // (64 windows of 8 + 4 key-frames 0.164 -> 0.132 ms per window, 256 windows 0.166 -> 0.117; option schur_kper overrides, for A/B runs).
  auto kper_ok = [&](int k) { return k >= 1 && a.nchunk % k == 0 && (a.nblk * a.nchunk / k) % NW_BA == 0; };
  a.kper = 1;
  if (c->opt.schur_kper >= 1) {
    a.kper = (int)c->opt.schur_kper;
  } else {
    for (int k = 8; k > 1; k /= 2)
      if (kper_ok(k) && (long)B * a.nblk * a.nchunk / k >= 2048) {
        a.kper = k;
        break;
      }
  }
  if (!kper_ok(a.kper)) a.kper = 1;
  // the chunks of the blocks added by the solve kernel itself when the call is a few windows (option pipe_fuse_asm: -1 by the size of
  // the call, 0 never, 1 always): a single 8 + 4 window 66 -> 62 us per cycle; in batches the adds belong on the whole chip
  a.fuse_asm = c->opt.pipe_fuse_asm >= 0 ? (c->opt.pipe_fuse_asm != 0 ? 1 : 0) : ((long)B * a.nblk * a.nchunk <= 600 ? 1 : 0);
  ln.schur_blocks = pipe_grid(B, a.nblk * a.nchunk / a.kper / NW_BA);  // (XCD-aware workgroup map: pipe_wg)
  ln.B = B;
  return GL_OK;
}


// the kernels of `n` cycles of a lane, from cycle `first_cyc` of the call
static void pipe_lane_cycle(PipeLane& ln, int cyc) {
  PipeA& a = ln.a;
  a.par = ln.par;
  a.cyc = cyc;
  ln.par ^= 1;
  const int B = ln.B;
  kp_lin<<<pipe_grid(B, a.nba), T_BA, 0, ln.s>>>(a);
  kp_schur<<<ln.schur_blocks, T_BA, 0, ln.s>>>(a);
  if (!a.fuse_asm) kp_assemble<<<(int)(((long)B * a.nblk * 48 + T_BA - 1) / T_BA), T_BA, 0, ln.s>>>(a);
  kp_solve<<<B, T_SOLVE, ln.s_bytes, ln.s>>>(a);
  kp_trial<<<pipe_grid(B, a.nba), T_BA, 0, ln.s>>>(a);
  if (a.ext_judge) kp_judge<<<B, 64, 0, ln.s>>>(a);
}

// Lanes of a call.  Every kernel of a cycle is bound by the LENGTH of its workgroups' chains of dependent loads times the rounds of
// workgroups the chip needs for the batch, not by issue slots or bandwidth, and a cycle is a chain of five such kernels (the solve
// with one workgroup per window).  A call of 16 or more windows (up to 2.5 M observations together: beyond that the kernels keep the
// chip busy by themselves) is cut into two halves that run their cycles on two streams: the kernels of one half fill the gaps of the
// other's.  Windows do not interact, so the bits are those of any other split.  (Option pipe_lanes = 1 .. 4 overrides, for A/B runs.)
constexpr int PIPE_LANES_MAX = 4;
int pipe_lanes(const Ctx* c, int B, int NOBS) {
  if (c->opt.pipe_lanes >= 1) return std::max(1, std::min(std::min((int)c->opt.pipe_lanes, PIPE_LANES_MAX), B));
  return (B >= 16 && (long)B * NOBS <= 2500000) ? 2 : 1;
}
// windows [first, first + count) of lane k (every lane but the last a multiple of 8 windows: the XCD-aware workgroup map)
static void pipe_lane_windows(int B, int nl, int k, int* first, int* count) {
  const int per = nl == 1 ? B : (B >= 8 * nl ? ((B / nl + 7) / 8) * 8 : (B + nl - 1) / nl);
  *first = std::min(B, k * per);
  *count = std::max(0, std::min(B, (k + 1) * per) - *first);
  if (k == nl - 1) *count = B - *first;
}
static size_t pipe_lane_scratch_off(int B, int nl, int k, int P, int F, int L, int NOBS) {
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  size_t off = 0;
  for (int i = 0; i < k; ++i) {
    int f0, cnt;
    pipe_lane_windows(B, nl, i, &f0, &cnt);
    off += up(ba_pipe_scratch_bytes(std::max(cnt, 1), P, F, L, NOBS));
  }
  return off;
}
size_t ba_pipe_scratch_total(const Ctx* c, int B, int P, int F, int L, int NOBS) {
  const int nl = pipe_lanes(c, B, NOBS);
  return pipe_lane_scratch_off(B, nl, nl, P, F, L, NOBS);
}

static int launch_ba_pipe_lanes(Ctx* c, PipeLane* lane, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int P, int F, int L, int NOBS,
                               double* poses_dev, const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                               const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev, const double* obs_uvr_dev,
                               const int32_t* obs_oct_dev, uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                               const int32_t* stop_dev, void* scratch) {
  // page-locked words the devices' counts of unfinished problems are copied to between chunks of cycles (4 per lane)
  if (!c->host_word) GL_HIP(hipHostMalloc((void**)&c->host_word, 64, hipHostMallocDefault));
  const int nl = pipe_lanes(c, B, NOBS);
  if (nl > 1 && !c->ev_fork) GL_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  for (int k = 1; k < nl; ++k)
    if (!c->lane_stream[k - 1]) {
      GL_HIP(hipStreamCreateWithFlags(&c->lane_stream[k - 1], hipStreamNonBlocking));
      GL_HIP(hipEventCreateWithFlags(&c->ev_join[k - 1], hipEventDisableTiming));
    }
  TimerScope ts(c, GL_TIMER_BA);
  if (nl > 1) GL_HIP(hipEventRecord(c->ev_fork, c->stream));  // the further streams start behind what the caller has enqueued so far
  for (int k = 0; k < nl; ++k) {
    int w0, Bk;
    pipe_lane_windows(B, nl, k, &w0, &Bk);
    if (Bk == 0) {
      lane[k].done = true;
      continue;
    }
    lane[k].s = k == 0 ? c->stream : c->lane_stream[k - 1];
    if (k > 0) GL_HIP(hipStreamWaitEvent(lane[k].s, c->ev_fork, 0));
    lane[k].hw = c->host_word + 4 * k;
    char* sk = (char*)scratch + pipe_lane_scratch_off(B, nl, k, P, F, L, NOBS);
    const size_t w = (size_t)w0;
    const int rc = pipe_lane_setup(c, lane[k], w0, c->stats && c->stats_n >= B, g, cam, prm, Bk, P, F, L, NOBS, poses_dev + w * (P + F) * 7, prior_dev + w * P,
                                   points_dev + w * L * 3, assoc_dev + w * L, obs_ptr_dev + w * (L + 1), obs_pose_dev + w * NOBS,
                                   obs_uvr_dev + w * NOBS * 3, obs_oct_dev + w * NOBS, assoc_dropped_dev ? assoc_dropped_dev + w * L : nullptr,
                                   obs_erase_dev ? obs_erase_dev + w * NOBS : nullptr, iters_dev ? iters_dev + w : nullptr, stop_dev, sk);
    if (rc != GL_OK) return rc;
  }
  // a run needs 3 lambda-init cycles + its Levenberg trials (28 - 35 on the windows measured; every rejected trial adds
  // one) + 4 cycles that open / change the stage + the one that judges the last trial: enough cycles for the common case are enqueued before the
  // first look at the counter, fewer per look afterwards
  // (the look-ahead of a call is what the context's previous window needed, + 2: windows follow each other with similar trial
  // counts; a cycle beyond the end costs five empty launches, a cycle short a host round trip)
  int chunk = c->pipe_hint > 0 ? std::min(std::max(c->pipe_hint + 2, 12), 80) : 44, hint = 0;
  for (int total = 0;; total += chunk, chunk = 8) {
    for (int cyc = 0; cyc < chunk; ++cyc)
      for (int k = 0; k < nl; ++k)
        if (!lane[k].done) pipe_lane_cycle(lane[k], total + cyc);
    GL_HIP(hipGetLastError());
    for (int k = 0; k < nl; ++k)
      if (!lane[k].done) GL_HIP(hipMemcpyAsync(lane[k].hw, lane[k].a.unfinished, 2 * sizeof(int), hipMemcpyDeviceToHost, lane[k].s));
    bool all = true;
    for (int k = 0; k < nl; ++k) {
      if (lane[k].done) continue;
      GL_HIP(hipStreamSynchronize(lane[k].s));
      if (lane[k].hw[0] <= 0) {
        lane[k].done = true;
        hint = std::max(hint, lane[k].hw[1]);
      } else {
        all = false;
      }
    }
    if (all) break;
    if (total > 600) {  // 3 x (40 iterations x 10 trials) is the schedule's bound; this is a defect, not a slow problem
      set_error("launch_ba_pipe: problems did not finish in %d cycles", total + chunk);
      return GL_ERR_DEVICE;
    }
  }
  if (hint > 0) c->pipe_hint = hint;
  for (int k = 1; k < nl; ++k) {  // what follows on the caller's stream follows the other lanes too (they have finished: the host has seen their counters)
    if (!lane[k].s) continue;
    GL_HIP(hipEventRecord(c->ev_join[k - 1], lane[k].s));
    GL_HIP(hipStreamWaitEvent(c->stream, c->ev_join[k - 1], 0));
  }
  return GL_OK;
}

// The lanes run on further non-blocking streams and share the context's scratch block: whatever way the call ends, nothing of it may
// still be in flight when it returns - the next call on the context only orders itself behind c->stream and may overwrite or free
// that scratch.  Every error exit therefore waits for all the lanes' streams first (the regular exit has seen their counters).
int launch_ba_pipe(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int P, int F, int L, int NOBS,
                   double* poses_dev, const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                   const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev, const double* obs_uvr_dev,
                   const int32_t* obs_oct_dev, uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                   const int32_t* stop_dev, void* scratch) {
  PipeLane lane[PIPE_LANES_MAX];
  const int rc = launch_ba_pipe_lanes(c, lane, g, cam, prm, B, P, F, L, NOBS, poses_dev, prior_dev, points_dev, assoc_dev, obs_ptr_dev, obs_pose_dev,
                                      obs_uvr_dev, obs_oct_dev, assoc_dropped_dev, obs_erase_dev, iters_dev, stop_dev, scratch);
  if (rc != GL_OK) {
    for (int k = 0; k < PIPE_LANES_MAX; ++k)
      if (lane[k].s) (void)hipStreamSynchronize(lane[k].s);  // (keep the first error's message)
  }
  return rc;
}
}  // namespace gl

static int joint_optimization_impl(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm,
                                   int B, int P, int F, int L, int NOBS, double* poses_dev,
                                   const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                                   const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev,
                                   const double* obs_uvr_dev, const int32_t* obs_oct_dev,
                                   uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                                   const int32_t* stop_dev) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && P >= 1 && F >= 0 && L >= 1 && NOBS >= 1, "bad problem shape");
  GL_REQUIRE(poses_dev && prior_dev && points_dev && assoc_dev && obs_ptr_dev && obs_pose_dev && obs_uvr_dev &&
                 obs_oct_dev && assoc_dropped_dev && obs_erase_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  void* scratch = nullptr;
  // Shape (option bagen_mode: 0 by size, 1 the persistent kernel, 2 the pipelined shape).  Measured per Levenberg trial
  // (profiles/history/r3c_ba_modes.txt; the two shapes add their partial sums in different orders, so a window can take a different
  // NUMBER of trials in each - 22 against 33 on the 20 + 8 window - which says nothing about either): 8 + 4 key-frames / 12 600
  // observations 73 us pipelined against 103 persistent, 12 + 4 / 22 400 103 against 149, 20 + 8 / 58 600 155 against 299; equal
  // at 3 - 4 free poses (57 us), the persistent kernel ahead below (1 pose: 32 against 49) and in batches (64 windows of 8 + 4:
  // 0.18 against 0.23 ms per window).
  const bool pipe_fits = P <= 22 && P + F <= 32 && (size_t)L * (NOBS >= 6 * L ? 8 : 4) <= 65536;  // (the judging workgroups hold <= 256 partial sums and the poses in LDS)
  // (mode 0 chooses from the WINDOW alone - never from B: the two shapes add their partial sums in different orders, and a
  // window must not change its bits, or the call its blocking behaviour, with the number of windows that ride along)
  // (from 3 000 observations since the end of round 4 - 5 000 before: profiles/history/r4f_ba_modes.txt has the pipelined shape ahead from
  // 3 300 observations at every batch size, 9 % on one window and 30 % on eight, and per trial already at 2 100)
  const bool pipe = pipe_fits && (c->opt.bagen_mode == 2 || (c->opt.bagen_mode == 0 && NOBS >= 3000));
  int rc = gl::ctx_scratch(c, pipe ? gl::ba_pipe_scratch_total(c, B, P, F, L, NOBS) : gl::ba_gen_scratch_bytes(B, P, F, L, NOBS), &scratch);
  if (rc != GL_OK) return rc;
  if (pipe) {
    return gl::launch_ba_pipe(c, g, cam, prm, B, P, F, L, NOBS, poses_dev, prior_dev, points_dev, assoc_dev, obs_ptr_dev, obs_pose_dev,
                              obs_uvr_dev, obs_oct_dev, assoc_dropped_dev, obs_erase_dev, iters_dev, stop_dev, scratch);
  }
  return gl::launch_ba_gen(c, g, cam, prm, B, P, F, L, NOBS, poses_dev, prior_dev, points_dev, assoc_dev, obs_ptr_dev, obs_pose_dev,
                           obs_uvr_dev, obs_oct_dev, assoc_dropped_dev, obs_erase_dev, iters_dev, stop_dev, scratch);
}

extern "C" int gl_joint_optimization(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm,
                                     int B, int P, int F, int L, int NOBS, double* poses_dev,
                                     const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                                     const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev,
                                     const double* obs_uvr_dev, const int32_t* obs_oct_dev,
                                     uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev) {
  return joint_optimization_impl(ctx, gmm, cam, prm, B, P, F, L, NOBS, poses_dev, prior_dev, points_dev, assoc_dev, obs_ptr_dev,
                                 obs_pose_dev, obs_uvr_dev, obs_oct_dev, assoc_dropped_dev, obs_erase_dev, iters_dev, nullptr);
}

extern "C" int gl_joint_optimization_stoppable(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm,
                                               int B, int P, int F, int L, int NOBS, double* poses_dev,
                                               const uint8_t* prior_dev, double* points_dev, const int32_t* assoc_dev,
                                               const int32_t* obs_ptr_dev, const int32_t* obs_pose_dev,
                                               const double* obs_uvr_dev, const int32_t* obs_oct_dev,
                                               uint8_t* assoc_dropped_dev, uint8_t* obs_erase_dev, int32_t* iters_dev,
                                               const int32_t* stop_flag) {
  return joint_optimization_impl(ctx, gmm, cam, prm, B, P, F, L, NOBS, poses_dev, prior_dev, points_dev, assoc_dev, obs_ptr_dev,
                                 obs_pose_dev, obs_uvr_dev, obs_oct_dev, assoc_dropped_dev, obs_erase_dev, iters_dev, stop_flag);
}

// ---- gl_track_frames_anchored with F > 0 fixed observer key-frames --------------------------------------------------
// The per-frame layout (one observation per point in the frame itself + up to F in fixed key-frames, dense B x M x F
// arrays) is packed into the flat problems of gl_joint_optimization - P = 1 free pose, F fixed ones, observations in CSR
// order by point: the frame's own first, then the fixed key-frames' in index order - and solved by the general kernel;
// localization_opt.cpp:491-516 (fixed observers), :556-581 (prior / fixed first key-frame), :706-760 (edges).
namespace {
constexpr int PK_T = 256;
__global__ __launch_bounds__(PK_T) void k_track_pack(int B, int M, int F, double gate, const double* __restrict__ pose, const uint8_t* __restrict__ prior,
                                                     const double* __restrict__ fpose, const double* __restrict__ obs, const int32_t* __restrict__ oct,
                                                     const double* __restrict__ fobs, const int32_t* __restrict__ foct, int32_t* __restrict__ assoc,
                                                     const double* __restrict__ d2, double* __restrict__ poses, uint8_t* __restrict__ prior_o,
                                                     int32_t* __restrict__ optr, int32_t* __restrict__ opose, double* __restrict__ ouvr,
                                                     int32_t* __restrict__ ooct) {
  __shared__ int wsum[PK_T / 64];
  __shared__ int s_base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (f >= B) return;
  const int NOBS = M * (1 + F);
  if (tid < 7 * (1 + F)) {
    const int j = tid / 7, r = tid % 7;
    poses[((size_t)f * (1 + F) + j) * 7 + r] = j == 0 ? pose[(size_t)f * 7 + r] : fpose[((size_t)f * F + (j - 1)) * 7 + r];
  }
  if (tid == 0) {
    prior_o[f] = prior ? prior[f] : 0;
    s_base = 0;
  }
  __syncthreads();
  for (int l0 = 0; l0 < M; l0 += PK_T) {
    const int l = l0 + tid;
    int cnt = 0, oc = -1;
    if (l < M) {
      const size_t g = (size_t)f * M + l;
      oc = oct[g];
      int a = assoc[g];
      if (d2 && gate >= 0 && !(d2[g] <= gate)) a = -1;  // checkMapAssociation's gate (gmmloc_opt.cpp:230-232)
      if (oc < 0) a = -1;
      assoc[g] = a;
      if (oc >= 0) {
        cnt = 1;
        for (int j = 0; j < F; ++j) cnt += foct[g * F + j] >= 0 ? 1 : 0;
      }
    }
    // exclusive scan over the 256 points of the round
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const int at = before + inc - cnt;
    if (l < M) {
      const size_t g = (size_t)f * M + l;
      optr[(size_t)f * (M + 1) + l] = at;
      if (oc >= 0) {
        size_t o = (size_t)f * NOBS + at;
        opose[o] = 0;
        ooct[o] = oc;
        for (int r = 0; r < 3; ++r) ouvr[o * 3 + r] = obs[g * 3 + r];
        for (int j = 0; j < F; ++j) {
          const int fo = foct[g * F + j];
          if (fo < 0) continue;
          ++o;
          opose[o] = 1 + j;
          ooct[o] = fo;
          for (int r = 0; r < 3; ++r) ouvr[o * 3 + r] = fobs[(g * F + j) * 3 + r];
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int t = s_base;
      for (int w = 0; w < PK_T / 64; ++w) t += wsum[w];
      s_base = t;
    }
    __syncthreads();
  }
  if (tid == 0) optr[(size_t)f * (M + 1) + M] = s_base;
}

__global__ __launch_bounds__(PK_T) void k_track_unpack(int B, int M, int F, const double* __restrict__ poses, const int32_t* __restrict__ optr,
                                                       const int32_t* __restrict__ opose, const uint8_t* __restrict__ dropped,
                                                       const uint8_t* __restrict__ erase, double* __restrict__ pose, int32_t* __restrict__ assoc,
                                                       uint8_t* __restrict__ ferase) {
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= B) return;
  const int NOBS = M * (1 + F);
  if (tid < 7) pose[(size_t)f * 7 + tid] = poses[(size_t)f * (1 + F) * 7 + tid];
  for (int l = tid; l < M; l += PK_T) {
    const size_t g = (size_t)f * M + l;
    if (dropped[g]) assoc[g] = -1;  // the final association, as gl_track_frames reports it
    if (ferase) {
      for (int j = 0; j < F; ++j) ferase[g * F + j] = 0;
      for (int o = optr[(size_t)f * (M + 1) + l]; o < optr[(size_t)f * (M + 1) + l + 1]; ++o) {
        const int j = opose[(size_t)f * NOBS + o];
        if (j >= 1) ferase[g * F + (j - 1)] = erase[(size_t)f * NOBS + o];
      }
    }
  }
}
}  // namespace

namespace gl {
int track_frames_fixed(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B, int M,
                       double* pose_dev, double* Xw_dev, const double* obs_dev, const int32_t* octave_dev, int32_t* assoc_dev,
                       double* d2_dev, const gl_track_anchor* an) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (B == 0 || M == 0) return GL_OK;
  GL_REQUIRE(B > 0 && M > 0, "bad B / M");
  GL_REQUIRE(pose_dev && Xw_dev && obs_dev && octave_dev && assoc_dev, "null buffer");
  const int F = an->F;
  GL_REQUIRE(an->fixed_pose_dev && an->fixed_obs_dev && an->fixed_oct_dev, "null buffer of the fixed observer key-frames");
  Ctx* c = C(ctx);
  Gmm* g = G(gmm);
  GL_HIP(hipSetDevice(c->device));
  const size_t n = (size_t)B * M, NOBS = (size_t)M * (1 + F);
  const bool use_grid = g->grid.enabled && c->opt.assoc_grid != 0;
  const size_t assoc_bytes = use_grid ? assoc_index_scratch_bytes(g->K, (int)n, d2_dev != nullptr) : assoc_scratch_bytes(g->K, (int)n);
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  // | association scratch | d2 | poses | prior | optr | opose | ouvr | ooct | dropped | erase | general kernel |
  size_t off = up(assoc_bytes);
  const size_t o_d2 = off;      off += up(n * 8);
  const size_t o_poses = off;   off += up((size_t)B * (1 + F) * 56);
  const size_t o_prior = off;   off += up((size_t)B);
  const size_t o_optr = off;    off += up((size_t)B * (M + 1) * 4);
  const size_t o_opose = off;   off += up((size_t)B * NOBS * 4);
  const size_t o_ouvr = off;    off += up((size_t)B * NOBS * 24);
  const size_t o_ooct = off;    off += up((size_t)B * NOBS * 4);
  const size_t o_drop = off;    off += up(n);
  const size_t o_erase = off;   off += up((size_t)B * NOBS);
  const size_t o_gen = off;     off += ba_gen_scratch_bytes(B, 1, F, M, (int)NOBS);
  void* scratch = nullptr;
  int rc = ctx_scratch(c, off + 256, &scratch);
  if (rc != GL_OK) return rc;
  char* s = (char*)scratch;
  double* d2 = d2_dev ? d2_dev : (double*)(s + o_d2);
  if (use_grid)
    rc = launch_assoc_index(c, g, Xw_dev, (int)n, assoc_dev, d2, d2_dev != nullptr, scratch);
  else
    rc = launch_assoc_brute(c, g, Xw_dev, (int)n, assoc_dev, d2);
  if (rc != GL_OK) return rc;
  {
    TimerScope ts(c, GL_TIMER_BA_PREP);
    k_track_pack<<<B, PK_T, 0, c->stream>>>(B, M, F, 9.0, pose_dev, an->prior_dev, an->fixed_pose_dev, obs_dev, octave_dev, an->fixed_obs_dev,
                                            an->fixed_oct_dev, assoc_dev, d2, (double*)(s + o_poses), (uint8_t*)(s + o_prior),
                                            (int32_t*)(s + o_optr), (int32_t*)(s + o_opose), (double*)(s + o_ouvr), (int32_t*)(s + o_ooct));
  }
  GL_HIP(hipGetLastError());
  rc = launch_ba_gen(c, g, cam, prm, B, 1, F, M, (int)NOBS, (double*)(s + o_poses), (const uint8_t*)(s + o_prior), Xw_dev, assoc_dev,
                     (const int32_t*)(s + o_optr), (const int32_t*)(s + o_opose), (const double*)(s + o_ouvr), (const int32_t*)(s + o_ooct),
                     (uint8_t*)(s + o_drop), (uint8_t*)(s + o_erase), nullptr, nullptr, s + o_gen,
                     1);  // one workgroup per frame: per-frame batches are large, and a frame's bits must not depend on the batch
  if (rc != GL_OK) return rc;
  k_track_unpack<<<B, PK_T, 0, c->stream>>>(B, M, F, (const double*)(s + o_poses), (const int32_t*)(s + o_optr), (const int32_t*)(s + o_opose),
                                            (const uint8_t*)(s + o_drop), (const uint8_t*)(s + o_erase), pose_dev, assoc_dev, an->fixed_erase_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
}  // namespace gl
