// Single-pose structure-constrained refinement, on-chip fast path: the instances of gl_ba_fast_impl.hpp
// (DENSE by LDS class, SPREAD; exact fp64 point step, and the fp32-cached step as an option) + the launcher.
#include <algorithm>
#include <cstdlib>

#include "gl_ba_common.hpp"

using namespace gld;
using namespace glba;

// ---- launch scratch and its producer ------------------------------------------------------------------------------
// Per point of the launch, in the PERMUTED order of its frame: the observation in normalised image coordinates (24 B),
// the original index (perm), the flag word, the gated association: 36 B.  (The plane record {n, n.mu} of a point's
// component is NOT copied per point: the passes fetch it from the map's K x 4 table by the association - 128 KB shared
// by every frame of an XCD instead of 64 KB per frame, which is what lets the per-pass streams of the 32 frames of an
// XCD stay in its 4 MB L2: fabric-side reads 12.0 -> 0.6 GB per launch.)
namespace {
struct PrepView {
  double* gobn;
  int32_t* perm;
  int32_t* pfl;
  int32_t* assoc_p;
};
// Fixed observer key-frames of an anchored launch (kFixed instances of the refine), written by k_ba1_prep into the launch's
// scratch: the key-frames' poses as {R, t}; per point and key-frame - SoA by key-frame, the frame's permuted point order -
// the normalised observation, octave | stereo << 4 (-1: not observed) and the edge's stale chi2; ferase: the caller's output.
struct FixedV {
  int F;
  const double* fRt;   // B x F x 12
  const double* fobn;  // B x F x L x 3
  const int32_t* foct; // B x F x L
  double* chif;        // B x F x L
  uint8_t* ferase;     // B x L x F (caller's order) or null
};
// the ONE argument of k_ba1_fast (the kernel re-reads its fields from the kernel-argument segment for every frame it draws)
struct BafKArgs {
  glba::BaK k;
  glba::GmmDev gm;
  int B, L, G, S;
  double* pose_io;
  double* pts_io;
  int32_t* assoc_all;
  uint8_t* dropped_all;
  uint8_t* erase_all;
  int32_t* iters_out;
  double* pn_all;
  int32_t* trials_out;
  int NB;
  unsigned long long* parts;
  int* ctl;
  long long limit;
  int xcc_trusted;
  const int32_t* oct_all;
  const uint8_t* prior_all;
  const double* prior_mi;
  double* stage;
  int nb_prev;
  int32_t* counters;
  int32_t* outer_out;
  FixedV fxv;
  int32_t* edges_out;
  int* frame_ctr;
  double* un_scratch;  // the two-groups-per-wave instance: 6 x 2000 doubles per workgroup of the launch (the points' hand-over slots)
};
__host__ __device__ inline PrepView prep_view(double* scratch, int B, int L) {
  PrepView v;
  const size_t n = (size_t)B * L;
  v.gobn = scratch;
  v.perm = (int32_t*)(scratch + n * 3);
  v.pfl = v.perm + n;
  v.assoc_p = v.pfl + n;
  return v;
}

// Set-up of a refine launch, one workgroup per frame: the association gate chi2 <= 9 (checkMapAssociation,
// gmmloc_opt.cpp:230-232), the flag word of every point, its normalised observation - and the ORDER the
// refine walks the frame in: a stable partition that puts the points associated with a NON-degenerate component (the
// volumetric ~5 % of a map, whose EdgePt2Gaussian needs the full 3x3 block R L L^T R^T instead of a rank-1 plane term)
// behind all the others.  A wave of 64 consecutive points then takes the expensive branch only in the last chunk or two of
// a frame instead of in 96 % of its slots (one such lane was enough to make the wave issue ~100 extra instructions per
// point and pass).  The order is a function of the frame's data alone, so the canonical summation order built on it stays
// independent of the launch shape and of the batch.
constexpr int PREP_T = 256, PREP_C = 8;  // rounds of 256 consecutive points: frames up to 2 048 points
#ifndef GL_PREP_EMPTY_LAST
#define GL_PREP_EMPTY_LAST 1  // (0: the order of rounds 3 - 5, for A/B runs: profiles/r6_track_sparse.txt)
#endif
__global__ __launch_bounds__(PREP_T) void k_ba1_prep(BaK k, GmmDev gm, int B, int L, const double* __restrict__ obs_all,
                                                    const int32_t* __restrict__ oct_all, int32_t* __restrict__ assoc_all,
                                                    const double* __restrict__ d2_all, double* __restrict__ scratch,
                                                    unsigned long long* __restrict__ xwords, int* __restrict__ xctl, int nxw,
                                                    const double* __restrict__ pose_all, const uint8_t* __restrict__ prior_all,
                                                    double* __restrict__ prior_mi, int F, const double* __restrict__ fpose_all,
                                                    const double* __restrict__ fobs_all, const int32_t* __restrict__ foct_all, double* __restrict__ fRt,
                                                    double* __restrict__ fobn, int32_t* __restrict__ foct, double* __restrict__ chif, int* __restrict__ frame_ctr) {
  __shared__ int cnt[PREP_C][PREP_T / 64];   // non-degenerate-component points per (round, wave)
  __shared__ int cnt0[PREP_C][PREP_T / 64];  // slots WITHOUT a map point per (round, wave) (round 6: they go last, see below)
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (f >= B) return;
  if (frame_ctr && f == 0 && tid == 0) *frame_ctr = 0;  // the queue the refine's persistent workgroups draw their frames from
  if (xwords) {  // latency shape next: this frame's exchange words and {abort, done} start at zero (no separate memset)
    for (int i = tid; i < nxw; i += PREP_T) xwords[(size_t)f * nxw + i] = 0ull;
    if (tid < 2) xctl[2 * f + tid] = 0;
  }
  if (prior_all && prior_all[f] && tid == 0) {  // EdgeSE3QuatPrior::_inverseMeasurement of the frame's INPUT pose, as {R, t}
    const SE3 Ti = se3_inverse(se3_load(pose_all + (size_t)f * 7));
    double Ri[9];
    qtoR(Ti.r, Ri);
#pragma unroll
    for (int i = 0; i < 9; ++i) prior_mi[(size_t)f * 12 + i] = Ri[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) prior_mi[(size_t)f * 12 + 9 + i] = Ti.t[i];
  }
  if (F > 0 && tid < F) {  // the fixed key-frames' poses T_cw as {R, t}
    const SE3 T = se3_load(fpose_all + ((size_t)f * F + tid) * 7);
    double Rm[9];
    qtoR(T.r, Rm);
    double* o = fRt + ((size_t)f * F + tid) * 12;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = Rm[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[9 + i] = T.t[i];
  }
  const size_t gbase = (size_t)f * L;
  const PrepView pv = prep_view(scratch, B, L);
  const int rounds = (L + PREP_T - 1) / PREP_T;
  // round j, thread t: point j 256 + t (coalesced); index order = (round, wave, lane)
  int a_[PREP_C], fl_[PREP_C];
#pragma unroll
  for (int j = 0; j < PREP_C; ++j) {
    const int l = j * PREP_T + tid;
    a_[j] = -1;
    fl_[j] = 0;
    if (j < rounds && l < L) {
      const size_t g = gbase + l;
      const int oc = oct_all[g];
      int a = assoc_all[g];
      if (d2_all && k.gate_chi2 >= 0 && !(d2_all[g] <= k.gate_chi2)) a = -1;
      if (oc < 0) a = -1;
      assoc_all[g] = a;  // the gated association, in the caller's order (the refine writes the final one)
      int fl = 0;
      if (oc >= 0) {
        fl = 1 | ((oc & 7) << 8);                              // F_EXISTS, octave
        if (!(obs_all[g * 3 + 2] < 0)) fl |= 2;                // F_STEREO
        if (a >= 0) fl |= 4 | ((gm.flags[a] & 1) ? 8 : 0);     // F_ASSOC, F_DEG
      }
      a_[j] = a;
      fl_[j] = fl;
    }
    const unsigned long long bal = __ballot((fl_[j] & 12) == 4);  // associated and not degenerate
    const unsigned long long bal0 = __ballot(GL_PREP_EMPTY_LAST && fl_[j] == 0 && j < rounds && j * PREP_T + tid < L);  // no map point in the slot
    if (lane == 0) {
      cnt[j][wave] = __popcll(bal);
      cnt0[j][wave] = __popcll(bal0);
    }
  }
  __syncthreads();
  int total = 0, total0 = 0;
#pragma unroll
  for (int j = 0; j < PREP_C; ++j)
#pragma unroll
    for (int w = 0; w < PREP_T / 64; ++w) {
      total += cnt[j][w];
      total0 += cnt0[j][w];
    }
  // Round 6: a stable partition in THREE classes - the points of planar components (and the unassociated ones), the points of
  // non-degenerate components, the slots without a map point.  The reference's frame has one slot per FEATURE (1 200) and a few hundred
  // map points: with the empty slots LAST the chunks behind the frame's points hold nothing, every wave skips them with one test
  // (load_pt: no active edge in any lane), and since a group's chunks are interleaved (g, g + G, ...) the points still spread evenly
  // over the waves - the refine costs what its POINTS cost, not what its slots cost.
  const int n_others = L - total - total0, n_exist = L - total0;
  const double ifx = 1.0 / k.fx, ify = 1.0 / k.fy;
  int base = 0, base0 = 0;  // non-degenerate-component points / empty slots before this (round, wave)
#pragma unroll
  for (int j = 0; j < PREP_C; ++j) {
#pragma unroll
    for (int w = 0; w < PREP_T / 64; ++w)
      if (w < wave) {
        base += cnt[j][w];
        base0 += cnt0[j][w];
      }
    const int l = j * PREP_T + tid;
    const bool isnd = (fl_[j] & 12) == 4;
    const bool isempty = GL_PREP_EMPTY_LAST && fl_[j] == 0 && j < rounds && l < L;
    const unsigned long long bal = __ballot(isnd), bal0 = __ballot(isempty);
    const int before = base + __popcll(bal & ((1ull << lane) - 1ull));
    const int before0 = base0 + __popcll(bal0 & ((1ull << lane) - 1ull));
    if (j < rounds && l < L) {
      const int lp = isempty ? n_exist + before0 : isnd ? n_others + before : l - before - before0;  // stable in all three classes
      const size_t g = gbase + l, gp = gbase + lp;
      pv.perm[gp] = l;
      pv.pfl[gp] = fl_[j];
      pv.assoc_p[gp] = a_[j];
      pv.gobn[gp * 3] = (obs_all[g * 3] - k.cx) * ifx;
      pv.gobn[gp * 3 + 1] = (obs_all[g * 3 + 1] - k.cy) * ify;
      pv.gobn[gp * 3 + 2] = (obs_all[g * 3 + 2] - k.cx) * ifx;
      for (int kf = 0; kf < F; ++kf) {  // the point's observations by the fixed key-frames, by key-frame, in the permuted order
        const size_t src = g * F + kf, dst = ((size_t)f * F + kf) * L + lp;
        const int fo = fl_[j] ? foct_all[src] : -1;
        const double u = fobs_all[src * 3], v = fobs_all[src * 3 + 1], ur = fobs_all[src * 3 + 2];
        fobn[dst * 3] = (u - k.cx) * ifx;
        fobn[dst * 3 + 1] = (v - k.cy) * ify;
        fobn[dst * 3 + 2] = (ur - k.cx) * ifx;
        foct[dst] = fo < 0 ? -1 : ((fo & 7) | (!(ur < 0) ? 16 : 0));
        chif[dst] = 0.0;
      }
    }
#pragma unroll
    for (int w = 0; w < PREP_T / 64; ++w)
      if (w >= wave) {  // the rest of this round: the bases now count everything before round j + 1
        base += cnt[j][w];
        base0 += cnt0[j][w];
      }
  }
}
}  // namespace

// (namespace, LDS capacity in points, waves at most, SPREAD, fp32-cached step, gauge anchor of the pose)
#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd496  // DENSE, exact step: 4 frames per CU
#define GL_BAF_MCAP 496
#define GL_BAF_NW 2
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd1000  // 2 frames per CU
#define GL_BAF_MCAP 1000
#define GL_BAF_NW 4
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#define GL_BAF_NS bafd2000  // 1 frame per CU
#if GL_BAF_W3
#define GL_BAF_MCAP 1984
#define GL_BAF_NW 12
#define GL_BAF_THREADS 768
#else
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#endif
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_THREADS
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED

#define GL_BAF_NS bafs  // SPREAD (latency shape), exact step
#define GL_BAF_MCAP 256
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 1
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED

// fp32-cached point step (option ba_step32): the SPREAD kernel and the largest DENSE class
#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafs32  // 
#define GL_BAF_MCAP 256
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 1
#define GL_BAF_STEP32 1
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd2000s32  // 
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 1
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

// anchored instances (gl_track_frames_anchored: prior edge on the frame's pose, or fixed pose), exact step
#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd2000x  // 2 frames per CU: two groups of the canonical order per wave, the hand-over slots in global memory (GL_BAF_GPW)
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 0
#define GL_BAF_FIXED 0
#undef GL_BAF_GPW
#define GL_BAF_GPW 2
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_GPW
#define GL_BAF_GPW 1
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd496p  // 
#define GL_BAF_MCAP 496
#define GL_BAF_NW 2
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd1000p  // (992 points: two frames per CU with the 512 bytes of the prior edge's records)
#define GL_BAF_MCAP 992
#define GL_BAF_NW 4
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd2000p  // 
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafsp  // 
#define GL_BAF_MCAP 256
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 1
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

// anchored instances WITH fixed observer key-frames (F = 1 .. 4; the prior edge / fixed pose stays a per-frame flag), batch shape
#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd496f
#define GL_BAF_MCAP 496
#define GL_BAF_NW 2
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 1
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd1000f  // (984 points: two frames per CU with the prior edge's records and the key-frames' poses)
#define GL_BAF_MCAP 984
#define GL_BAF_NW 4
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 1
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK
#define GL_BAF_NS bafd2000f
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#define GL_BAF_PRIOR 1
#define GL_BAF_FIXED 1
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32
#undef GL_BAF_PRIOR
#undef GL_BAF_FIXED
#endif

#ifndef GL_BAF_QUICK  // (tools/baf_quick.sh: the 2 000-point batch instance and the latency shape alone, for register / ISA checks)
namespace {
// HW_REG_XCC_ID (hwreg 20, 4 bits): the XCD the wave runs on
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }
__global__ void k_xcc_probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
}  // namespace

namespace gl {

// The same-XCD form of the latency shape's exchange (plain stores that stay in the XCD's L2, read by L1-bypassing loads)
// is only valid if the workgroups of a frame really share an XCD.  The kernel checks that at run time with the XCC ids
// its workgroups report; this probe decides whether those reports can be trusted at all: 64 blocks must report ids in
// 0..7, block b the same as block b % 8, and the eight residues eight different ones (the known placement of this
// hardware).  Anything else - another partition mode, another device - leaves the device-scope form in place.
bool probe_xcc_ids(Ctx* c) {
  int* d = nullptr;
  if (hipMalloc((void**)&d, 64 * sizeof(int)) != hipSuccess) return false;
  int h[64];
  bool ok = hipMemsetAsync(d, 0xff, 64 * sizeof(int), c->stream) == hipSuccess;
  if (ok) {
    k_xcc_probe<<<64, 64, 0, c->stream>>>(d);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess;
  }
  (void)hipFree(d);
  if (!ok) return false;
  unsigned seen = 0;
  for (int b = 0; b < 64; ++b) {
    if (h[b] < 0 || h[b] > 7 || h[b] != h[b & 7]) return false;
    seen |= 1u << h[b];
  }
  return seen == 0xffu;
}

bool ba1_fast_supported(int L) { return L <= 2000; }

// the canonical summation order of a frame of stride L (gl_ba_fast_impl.hpp): G groups of S chunks of 64 points
#ifndef GL_BAF_W3
#define GL_BAF_W3 0  // EXPERIMENT (profiles/r5_w3_*.txt): the largest class at THREE waves per SIMD - groups of 3 chunks, 12 waves, 168 registers
#endif
static void canon_order(int L, int* G, int* S) {
  const int nch = (L + 63) / 64;
  *G = (GL_BAF_W3 && L > 1000) ? (nch + 2) / 3 : (nch + 3) / 4;
  *S = (nch + *G - 1) / *G;
}

typedef void (*BafKernel)(BafKArgs);

struct BafArgs {
  BaK k;
  GmmDev gm;
  int B, L, G, S;
  double *pose, *pts;
  const double* obs;
  const int32_t* oct;
  int32_t* assoc;
  const double* d2;
  uint8_t *dropped, *erase;
  int32_t* iters;
  double* pn;
  int32_t* stats;
  int32_t* stats_iters = nullptr;
  int32_t* stats_edges = nullptr;
  int* fctr = nullptr;  // frame queue of the persistent DENSE workgroups (in the launch scratch, zeroed by k_ba1_prep), or null: plain launch
  int NB;
  unsigned long long* parts;
  int* ctl = nullptr;  // per frame {abort, done} of a latency-shape launch (the follow-up DENSE launch skips the done ones)
  const uint8_t* prior = nullptr;  // per frame: gauge anchor of the pose (prior edge / fixed), or null
  const double* prior_mi = nullptr;  // per frame: inverse measurement of the prior edge {R, t} (written by k_ba1_prep)
  double* stage = nullptr;           // latency shape: staging area of the results {points | pose | association}
  int nb_prev = 0;                   // follow-up launch: workgroups per frame of the latency-shape launch before it
  int32_t* counters = nullptr;       // the context's device counters (gl_ctx_counter_read)
  FixedV fx = {0, nullptr, nullptr, nullptr, nullptr, nullptr};  // fixed observer key-frames (F > 0: the kFixed instances)
};

// one workgroup of G waves per frame; LDS class by stride (4 / 2 / 1 frames per CU)
static int launch_dense(Ctx* c, BafArgs& a) {
  const bool fixed = a.fx.F > 0;
  const bool anch = a.prior || fixed;                  // (the fixed-observer instances carry the anchored code: the prior is a per-frame flag)
  const bool s32 = c->opt.ba_step32 != 0 && !anch;     // (the anchored instances exist with the exact step only)
  // (the anchored middle classes give 8 / 16 points for the prior edge's LDS records and the key-frames' poses)
  const int mid = fixed ? 984 : a.prior ? 992 : 1000;
  const int cap = s32 ? 2000 : (a.L <= 496 ? 496 : a.L <= mid ? mid : 2000);
  const BafKernel kern0 = s32 ? bafd2000s32::k_ba1_fast
                         : fixed ? (cap == 496 ? bafd496f::k_ba1_fast : cap == 984 ? bafd1000f::k_ba1_fast : bafd2000f::k_ba1_fast)
                         : a.prior ? (cap == 496 ? bafd496p::k_ba1_fast : cap == 992 ? bafd1000p::k_ba1_fast : bafd2000p::k_ba1_fast)
                                   : (cap == 496 ? bafd496::k_ba1_fast : cap == 1000 ? bafd1000::k_ba1_fast : bafd2000::k_ba1_fast);
  // the largest class, plain refine: two frames per CU (bafd2000x: two groups per wave, hand-over slots in global memory; same bits)
  const bool two = !GL_BAF_W3 && c->opt.ba_two_frames != 0 && kern0 == bafd2000::k_ba1_fast && !a.ctl;
  const BafKernel kern = two ? bafd2000x::k_ba1_fast : kern0;
  const int threads = two ? 64 * std::min(a.G, 4) : 64 * a.G;
  const bool w3 = GL_BAF_W3 && kern == bafd2000::k_ba1_fast;  // (experiment: 12 groups' totals, MCAP 1984)
  if (w3 && a.L > 1984) return GL_ERR_ARG;
  const size_t lds = w3 ? (size_t)(10 * 1984 + 12 * 32 + 64 + 24 + 24) * sizeof(double)
                     : two ? (size_t)(4 * cap + 8 * 32 + 64 + 40) * sizeof(double)
                        : (size_t)(10 * cap + (cap == 496 ? 2 : cap <= 1000 ? 4 : 8) * 32 + 64 + 40 + (anch ? 64 + (cap == 496 ? 108 : 0) : 0) + (fixed ? 48 : 0)) * sizeof(double);
  GL_HIP(ensure_dynamic_lds(c, (const void*)kern, lds));
  a.NB = 1;
  a.parts = nullptr;
  // persistent workgroups (option ba_persist, default on): as many as the device holds at once, drawing frames from a.fctr
  int grid = a.B;
  int* fctr = nullptr;
  if (c->opt.ba_persist != 0 && a.fctr) {
    const auto key = std::make_pair((const void*)kern, lds + (size_t)threads);
    auto hit = c->occupancy.find(key);
    if (hit == c->occupancy.end()) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, threads, lds) != hipSuccess) occ = 0;
      hit = c->occupancy.emplace(key, occ).first;
    }
    if (hit->second > 0 && (long)hit->second * c->ncu < (long)a.B) {
      grid = hit->second * c->ncu;
      fctr = a.fctr;
    }
  }
  void* un = nullptr;  // (bafd2000x: 96 KB of hand-over slots per workgroup of the launch, in the context's second scratch block)
  if (two) {
    const int rc = gl::ctx_scratch_b(c, (size_t)grid * 6 * 2000 * sizeof(double), &un);
    if (rc != GL_OK) return rc;
  }
  const BafKArgs ka{a.k, a.gm, a.B, a.L, a.G, a.S, a.pose, a.pts, a.assoc, a.dropped, a.erase, a.iters, a.pn, a.stats, a.NB, a.parts, a.ctl, 0ll, 0, a.oct, a.prior,
                    a.prior_mi, a.stage, a.nb_prev, a.counters, a.stats_iters, a.fx, a.stats_edges, fctr, (double*)un};
  kern<<<grid, threads, lds, c->stream>>>(ka);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// Few frames (the frame-at-a-time caller): one point per thread, a workgroup of 256 threads = the <= 4 slot waves of ONE
// group, G workgroups per frame on as many CUs (one wave per SIMD: nothing to share the issue slots with); with G > 1 the
// reductions of a Levenberg trial cross the workgroups through tagged words in global memory.  That needs the workgroups
// of a frame co-resident.  A cooperative launch guarantees it and costs 31 us per call here (0.463 vs 0.432 ms for one
// frame); the kernel is launched plainly instead, with the rendezvous protocol of gld::Coop: a frame whose workgroups do not
// all show up within the time limit (another launch holds the CUs and waits for its own) gives up without writing
// anything, and the one-workgroup kernel that follows redoes exactly those frames - same bits, so the caller never sees
// which kernel answered.  Returns 1 when the shape does not fit the device at all (the caller goes DENSE).
static int launch_spread(Ctx* c, BafArgs& a, void* scratch) {
  const BafKernel kern = a.prior ? bafsp::k_ba1_fast : c->opt.ba_step32 != 0 ? bafs32::k_ba1_fast : bafs::k_ba1_fast;
  const size_t lds = (size_t)(10 * 256 + 1 * 32 + 64 + 40 + (a.prior ? 64 : 0) + 29 * 256) * sizeof(double);
  GL_HIP(ensure_dynamic_lds(c, (const void*)kern, lds));
  a.NB = a.G;
  {  // all the workgroups of the launch must fit the device at once (the occupancy answer is cached per context)
    const auto key = std::make_pair((const void*)kern, lds);
    auto hit = c->occupancy.find(key);
    if (hit == c->occupancy.end()) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, lds) != hipSuccess) occ = 0;
      hit = c->occupancy.emplace(key, occ).first;
    }
    // ... per XCD: the kernel keeps the workgroups of a frame on ONE XCD (frame f -> XCD f % 8), so what has to fit is the
    // ceil(B / 8) frames of an XCD into that XCD's share of the slots (a frame whose siblings queue behind the resident
    // ones would sit in the rendezvous until its time limit and be redone by the follow-up kernel: correct, but slow)
    if ((long)((a.B + 7) / 8) * a.NB > (long)hit->second * (c->ncu / 8)) return 1;
  }
  // the exchange words of the frames sit behind the per-point records, {abort, done} per frame behind them
  a.parts = (unsigned long long*)((char*)scratch + (((size_t)a.B * a.L * 36 + 63) / 64) * 64);
  a.ctl = (int*)(a.parts + (size_t)a.B * 2 * a.NB * 64);  // (both zeroed by k_ba1_prep)
  long long limit = (long long)(c->opt.ba_rendezvous_us * 100.0);  // wall_clock64() ticks at 100 MHz
  if (c->opt.ba_test_abort_seq > 0) limit = -(long long)c->opt.ba_test_abort_seq;  // tests: a give-up in the middle of the schedule
  // (NB > 1: 64 block indices per 8 frames, the kernel's map from block to (frame, group) keeps a frame on one XCD)
  const int grid = a.NB > 1 ? 64 * ((a.B + 7) / 8) : a.B;
  const BafKArgs ka{a.k, a.gm, a.B, a.L, a.G, a.S, a.pose, a.pts, a.assoc, a.dropped, a.erase, a.iters, a.pn, a.stats, a.NB, a.parts, a.ctl, limit,
                    (c->xcc_ids_trusted && c->opt.ba_same_xcd != 0) ? 1 : 0, a.oct, a.prior, a.prior_mi, a.stage, 0, a.counters, a.stats_iters, a.fx, a.stats_edges, nullptr, nullptr};
  kern<<<grid, 256, lds, c->stream>>>(ka);
  GL_HIP(hipGetLastError());
  return a.NB > 1 ? 2 : GL_OK;  // 2: follow up with DENSE for the frames that did not complete
}

// Shape: SPREAD when every workgroup of the batch gets a CU of its own (B NB <= CUs: the frame-at-a-time
// caller, small batches), DENSE otherwise.  Both add in the same order: the choice never shows in the results
// (option ba_shape forces one: 0 DENSE, 1 SPREAD).
int launch_ba1_fast(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, double* pose,
                    double* pts, const double* obs, const int32_t* oct, int32_t* assoc, const double* d2, double gate,
                    uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch, const uint8_t* prior, const TrackFixed* fixed) {
  BafArgs a;
  a.prior = prior;
  const int F = fixed ? fixed->F : 0;
  double *fRt = nullptr, *fobn = nullptr, *chif = nullptr;
  int32_t* foct = nullptr;
  if (F > 0) {  // the fixed-observer records sit behind everything else of the launch's scratch (ba1_scratch_bytes)
    char* fs = (char*)scratch + ((ba1_scratch_bytes(B, L, 0) + 63) / 64) * 64;
    fRt = (double*)fs;
    fobn = fRt + (size_t)B * F * 12;
    chif = fobn + (size_t)B * F * L * 3;
    foct = (int32_t*)(chif + (size_t)B * F * L);
    a.fx = FixedV{F, fRt, fobn, foct, chif, fixed->erase};
  }
  a.k = make_bak(cam, prm, gate);
  a.gm = GmmDev{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags, g->plane4};
  a.B = B;
  a.L = L;
  canon_order(L, &a.G, &a.S);
  a.pose = pose;
  a.pts = pts;
  a.obs = obs;
  a.oct = oct;
  a.assoc = assoc;
  a.d2 = d2;
  a.dropped = dropped;
  a.erase = erase;
  a.iters = iters;
  a.pn = (double*)scratch;
  a.stats = (c->stats && c->stats_n >= B) ? c->stats : nullptr;
  a.stats_iters = a.stats ? c->stats_iters : nullptr;
  a.stats_edges = (c->stats_edges && c->stats_edges_n >= B) ? c->stats_edges : nullptr;
  a.counters = c->counters;
  bool spread = (long)B * a.G <= 2 * c->ncu;  // two workgroups of 256 threads fit a CU (LDS 2 x 80 KB, 2 waves per SIMD; checked in launch_spread)
  if (c->opt.ba_shape == 0) spread = false;
  if (c->opt.ba_shape == 1) spread = true;  // forced (tests); a shape that does not fit the device still goes DENSE
  if (F > 0) spread = false;                // (fixed observers: batch-shaped instances only)
  {  // set-up: gate, flags, normalised observations, the order the refine walks each frame in - and, before a latency-shape
     // launch, the zeros its exchange words start from
    TimerScope ts(c, GL_TIMER_BA_PREP);
    unsigned long long* xw = nullptr;
    if (spread) xw = (unsigned long long*)((char*)scratch + (((size_t)B * L * 36 + 63) / 64) * 64);
    const int nxw = 2 * a.G * 64;
    // inverse measurements of the prior edges: behind the records, the exchange words and {abort, done} (ba1_scratch_bytes)
    a.prior_mi = (double*)((char*)scratch + (((size_t)B * L * 36 + 63) / 64) * 64 + (size_t)B * (8192 + 8) + 64);
    a.fctr = (int*)((char*)scratch + (((size_t)B * L * 36 + 63) / 64) * 64 + (size_t)B * (8192 + 8));  // (first word of the 64-byte pad in front of them)
    if (spread) a.stage = (double*)(a.prior_mi + (size_t)B * 12);  // {points B x L x 3 | pose B x 8 | association B x L}
    k_ba1_prep<<<B, PREP_T, 0, c->stream>>>(a.k, a.gm, B, L, obs, oct, assoc, d2, (double*)scratch, xw, xw ? (int*)(xw + (size_t)B * nxw) : nullptr, nxw,
                                            pose, prior, (double*)a.prior_mi, F, F ? fixed->pose : nullptr, F ? fixed->obs : nullptr,
                                            F ? fixed->oct : nullptr, fRt, fobn, foct, chif, a.fctr);
  }
  GL_HIP(hipGetLastError());
  TimerScope ts(c, GL_TIMER_BA);  // the refine kernel proper
  if (spread) {
    const int rc = launch_spread(c, a, scratch);
    if (rc <= 0) return rc;
    if (rc == 1) a.ctl = nullptr;  // shape does not fit: every frame goes DENSE
    else a.nb_prev = a.NB;         // follow-up: staged results of the complete frames -> the caller's buffers, the others redone
  }
  return launch_dense(c, a);
}

}  // namespace gl
#endif  // GL_BAF_QUICK
