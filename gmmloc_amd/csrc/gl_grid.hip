// Exact cell index for the Mahalanobis argmin (GL_ASSOC_BRUTE).
//
// The north-star association is  argmin_k chi2_k(p)  over ALL K components
// (GaussianComponent::chi2, gaussian.cpp:65-70), and what the callers keep is gated at
// chi2 <= 9 (gmmloc_opt.cpp:230-232).  A component can only reach chi2_k(p) <= T if p lies in
// its T-ellipsoid, so a uniform grid over the map in which every cell lists the components whose
// T-ellipsoid can touch the cell answers the query exactly for every point whose minimum is
// <= T: all candidates with chi2 <= T are in the point's cell list, the minimum over the list is
// the global minimum, and ties resolve to the lowest index because the lists are ascending.
// Points whose list minimum is > T (outliers, ~15 % of a frame) are either reported as
// "no association" (the fused tracking path, where chi2 > 9 is dropped anyway) or re-done by an
// all-pairs sweep restricted to them (k_assoc_brute with a point list), so GL_ASSOC_BRUTE keeps returning exactly what the all-pairs
// sweep (GL_ASSOC_EXHAUSTIVE, k_assoc_brute) returns -- with ~10-20 instead of K evaluations
// per point on the EuRoC maps and on the synthetic configs[1] map.
//
// Conservative registration (at gl_gmm_create; the O(K) set-up on the host, the enumeration on the device), component k in cell C iff
//   (1) C overlaps the axis-aligned box  mu_k +- sqrt(T cov_ii)  (exact AABB of the ellipsoid), and
//   (2) the cell centre c passes  (c-mu)^T ((1+1/b) T cov + (1+b) rho^2 I)^-1 (c-mu) <= 1  for
//       b in {1/2, 1, 2}, rho = half diagonal of the cell: the outer ellipsoidal bound of
//       E(T cov) (+) Ball(rho), which contains the centre of every cell the ellipsoid touches.
// Rounding is covered by margins: T is inflated by 4e-6 for registration and by 1e-6 for the
// "resolved" test, cells by 1e-9; components that are not SPD / finite, have a condition number
// above 1e8 (where the computed chi2 may differ from the exact form by more than the margin) or
// would cover more than 2^22 cells go to a short global list every point evaluates.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <hipcub/hipcub.hpp>

#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {

typedef const double __attribute__((address_space(4))) cdouble;

struct GridDev {
  double lo[3];
  double inv_h;
  int dim[3];
  int nglob;
  double t_resolve;
  const int32_t* ptr;
  const int32_t* idx;
  const int32_t* glob;
  const int4* cell4;
  const unsigned long long* cell8;  // round 6: the packed cell in 8 bytes (K < 2^20), or null
};

// A packed cell: the count and, for up to three candidates, the component indices themselves (ascending, like the CSR list); a longer
// list keeps its place in the CSR array.  16-byte form {n, a, b, c} / {n, e0, -, -}; 8-byte form (round 6: half the table - 87 MB on the
// bench map, inside the 256 MB of Infinity Cache - for one 128-byte line per point either way):
//     bit 63 = 0: n in bits 0..1, the three indices in 20 bits each from bit 2;    bit 63 = 1: e0 in bits 0..31, n in bits 32..61
GL_DEV int4 cell_unpack8(unsigned long long w) {
  if (w >> 63) return make_int4((int)((w >> 32) & 0x3fffffffull), (int)(w & 0xffffffffull), 0, 0);
  return make_int4((int)(w & 3ull), (int)((w >> 2) & 0xfffffull), (int)((w >> 22) & 0xfffffull), (int)((w >> 42) & 0xfffffull));
}
GL_DEV bool grid_packed(const GridDev& G) { return G.cell4 != nullptr || G.cell8 != nullptr; }
GL_DEV int4 cell_load(const GridDev& G, int c) { return G.cell8 ? cell_unpack8(G.cell8[c]) : G.cell4[c]; }

// lexicographic (chi2, index) minimum: the all-pairs sweep keeps the FIRST index of the minimum
GL_DEV void upd_min(double d, int k, double& best, int& bi) {
  const bool take = (d < best) || (d == best && k < bi);
  best = take ? d : best;
  bi = take ? k : bi;
}

__global__ __launch_bounds__(256) void k_assoc_cells(const double* __restrict__ rec12, GridDev G,
                                                     const double* __restrict__ pts, int N,
                                                     int32_t* __restrict__ out_idx, double* __restrict__ out_d2,
                                                     int32_t* __restrict__ rest_list, int32_t* __restrict__ rest_count) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double x = pts[(size_t)n * 3], y = pts[(size_t)n * 3 + 1], z = pts[(size_t)n * 3 + 2];
  double best = __builtin_inf();
  int bi = 0x7fffffff;
  cdouble* rc = (cdouble*)rec12;
  for (int j = 0; j < G.nglob; ++j) {  // wave-uniform: records through scalar loads
    const int k = G.glob[j];
    cdouble* r = rc + (size_t)k * 12;
    const double d0 = x - r[0], d1 = y - r[1], d2 = z - r[2];
    const double r0 = fma(d2, r[9], fma(d1, r[6], d0 * r[3]));
    const double r1 = fma(d2, r[10], fma(d1, r[7], d0 * r[4]));
    const double r2 = fma(d2, r[11], fma(d1, r[8], d0 * r[5]));
    upd_min(fma(r2, d2, fma(r1, d1, r0 * d0)), k, best, bi);
  }
  const double fx = (x - G.lo[0]) * G.inv_h, fy = (y - G.lo[1]) * G.inv_h, fz = (z - G.lo[2]) * G.inv_h;
  const bool inside = fx >= 0.0 && fx < (double)G.dim[0] && fy >= 0.0 && fy < (double)G.dim[1] && fz >= 0.0 &&
                      fz < (double)G.dim[2];  // false for NaN
  if (inside) {
    const int c = ((int)fz * G.dim[1] + (int)fy) * G.dim[0] + (int)fx;
    if (grid_packed(G)) {  // packed cell: the list of up to three candidates arrives with the count (ascending, like the CSR list)
      const int4 q = cell_load(G, c);
      if (q.x <= 3) {
        if (q.x > 0) upd_min(chi2_rec(rec12 + (size_t)q.y * 12, x, y, z), q.y, best, bi);
        if (q.x > 1) upd_min(chi2_rec(rec12 + (size_t)q.z * 12, x, y, z), q.z, best, bi);
        if (q.x > 2) upd_min(chi2_rec(rec12 + (size_t)q.w * 12, x, y, z), q.w, best, bi);
      } else {
        for (int e = q.y; e < q.y + q.x; ++e) {
          const int k = G.idx[e];
          upd_min(chi2_rec(rec12 + (size_t)k * 12, x, y, z), k, best, bi);
        }
      }
    } else {
      const int e0 = G.ptr[c], e1 = G.ptr[c + 1];
      for (int e = e0; e < e1; ++e) {
        const int k = G.idx[e];
        upd_min(chi2_rec(rec12 + (size_t)k * 12, x, y, z), k, best, bi);
      }
    }
  }
  if (best <= G.t_resolve) {
    out_idx[n] = bi;
    if (out_d2) out_d2[n] = best;
  } else {
    out_idx[n] = -1;
    if (out_d2) out_d2[n] = __builtin_inf();
    if (rest_list) rest_list[atomicAdd(rest_count, 1)] = n;
  }
}

// The same association with a WAVE-COOPERATIVE record gather.  In k_assoc_cells every lane fetches its own candidates' 96-byte
// records: 6 loads of 16 bytes per candidate with 64 different lines per instruction.  Here the wave lists its candidates
// (<= 3 per lane from the packed cell, compacted with a prefix sum over the lanes), six lanes fetch one record as six
// neighbouring 16-byte pieces and the records reach their points through LDS, 60 records per round; lanes whose cell holds
// more than three candidates walked their list alone until round 5 (LONG = false; see below).  Same arithmetic per pair, same (chi2, index) minimum.
// Measured on the bench points: 0.421 -> 0.392 ms per 8.19 M points - the address cycles of the gather drop to a third, the
// ~50 M line requests per launch to the XCDs' L2 (two lines per record, the 393 KB of records do not live in a 32 KB L1) do not.
// STRIDE = 16 (round 5): the gather reads the copy of the records that holds one record per 128-byte line (CellIndex::rec16) - one
// line request per record instead of 1.75 on average; the long lists and the global list stay on rec12 (the same values).
#ifndef GL_CG_REC
#define GL_CG_REC 60
#endif
#ifndef GL_CG_PIPE
#define GL_CG_PIPE 1
#endif
#ifndef GL_CG_IDXB
#define GL_CG_IDXB 1
#endif
#ifndef GL_CG_BAL
#define GL_CG_BAL 1
#endif
#ifndef GL_CG_IDS
#define GL_CG_IDS 304
#endif
constexpr int CG_REC = GL_CG_REC;  // records per round: 10 per load instruction (6 lanes each, 4 lanes idle), 6 instructions
constexpr int CG_LONE_LANES = 32;  // a wave gathers its long lists cooperatively while at most half of its lanes have one
constexpr int CG_TAG = 26, CG_KMASK = (1 << CG_TAG) - 1;  // (component indices below 2^26: launch_assoc_index)
constexpr int CG_IDS = GL_CG_IDS;  // candidates of a wave per chunk of its table (the bench points: 201 per wave on average; 192 - a second chunk for most waves - costs 15 %)
template <int STRIDE, bool LONG, bool BAL>
__global__ __launch_bounds__(256) void k_assoc_cells_coop(const double* __restrict__ rec12, const double* __restrict__ recg, GridDev G,
                                                          const double* __restrict__ pts, int N,
                                                          int32_t* __restrict__ out_idx, double* __restrict__ out_d2,
                                                          int32_t* __restrict__ rest_list, int32_t* __restrict__ rest_count) {
  __shared__ __attribute__((aligned(16))) double s_rec[4][CG_REC * 12];
  __shared__ int s_id[4][CG_IDS];
  __shared__ double s_d[BAL ? 4 : 1][BAL ? CG_IDS : 1];  // BAL: chi2 of the wave's candidates, by position in its table
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  const double x = pts[(size_t)nc * 3], y = pts[(size_t)nc * 3 + 1], z = pts[(size_t)nc * 3 + 2];
  double best = __builtin_inf();
  int bi = 0x7fffffff;
  cdouble* rc = (cdouble*)rec12;
  for (int j = 0; j < G.nglob; ++j) {  // wave-uniform: records through scalar loads
    const int k = G.glob[j];
    cdouble* r = rc + (size_t)k * 12;
    const double d0 = x - r[0], d1 = y - r[1], d2 = z - r[2];
    const double r0 = fma(d2, r[9], fma(d1, r[6], d0 * r[3]));
    const double r1 = fma(d2, r[10], fma(d1, r[7], d0 * r[4]));
    const double r2 = fma(d2, r[11], fma(d1, r[8], d0 * r[5]));
    upd_min(fma(r2, d2, fma(r1, d1, r0 * d0)), k, best, bi);
  }
  const double fx = (x - G.lo[0]) * G.inv_h, fy = (y - G.lo[1]) * G.inv_h, fz = (z - G.lo[2]) * G.inv_h;
  const bool inside = live && fx >= 0.0 && fx < (double)G.dim[0] && fy >= 0.0 && fy < (double)G.dim[1] && fz >= 0.0 &&
                      fz < (double)G.dim[2];  // false for NaN
  int4 q = make_int4(0, 0, 0, 0);
  if (inside) q = cell_load(G, ((int)fz * G.dim[1] + (int)fy) * G.dim[0] + (int)fx);
  // EVERY list goes through the cooperative gather (round 5, LONG = true): a cell with more than three candidates keeps its list in the
  // CSR array, and until round 5 such a lane walked it alone behind the gather - one dependent index load and six 16-byte record loads
  // per candidate while the rest of the wave waited (a third of the bench points; the time of the kernel followed THEM:
  // profiles/r5_assoc_cell_sweep.txt, the 3 cm row).  Now the lane copies its list's indices into the wave's candidate table like
  // the short lists' three, in chunks of CG_IDS candidates.
  // ... unless the wave's lists are long throughout (the stress map: 74 candidates per point): then every lane is busy walking its own
  // list and the gather's rounds of 60 are the slower way (measured 1.05 against 0.38 ms per 50 000 points).  Decided per wave by
  // the number of lanes with a long list.
  const bool lone = !LONG || __popcll(__ballot(q.x > 3)) > CG_LONE_LANES;
  const int cnt = lone ? (q.x <= 3 ? q.x : 0) : q.x;
  // exclusive prefix sum of the counts over the wave
  int pos = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(pos, o);
    if (lane >= o) pos += t;
  }
  const int total = __shfl(pos, 63);
  pos -= cnt;
  const int sub = lane / 6, part = lane - 6 * sub;  // lanes 60 .. 63 idle in the gather
  const int otag = BAL ? lane << CG_TAG : 0;        // BAL: a candidate carries its point's lane in the bits above the component index
  for (int c0 = 0; c0 < total; c0 += CG_IDS) {
    const int jlo = max(0, c0 - pos), jhi = min(cnt, c0 + CG_IDS - pos);  // this lane's candidates of the chunk
    if (q.x <= 3) {
      if (0 >= jlo && 0 < jhi) s_id[wave][pos - c0] = q.y | otag;
      if (1 >= jlo && 1 < jhi) s_id[wave][pos + 1 - c0] = q.z | otag;
      if (2 >= jlo && 2 < jhi) s_id[wave][pos + 2 - c0] = q.w | otag;
    } else if (!lone) {
#if GL_CG_IDXB > 1
      for (int j0 = jlo; j0 < jhi; j0 += GL_CG_IDXB) {  // GL_CG_IDXB index loads in flight together (one after the other they were a list's length of round trips)
        int v[GL_CG_IDXB];
#pragma unroll
        for (int u = 0; u < GL_CG_IDXB; ++u) v[u] = j0 + u < jhi ? G.idx[q.y + j0 + u] : 0;
#pragma unroll
        for (int u = 0; u < GL_CG_IDXB; ++u)
          if (j0 + u < jhi) s_id[wave][pos + j0 + u - c0] = v[u] | otag;
      }
#else
      for (int j = jlo; j < jhi; ++j) s_id[wave][pos + j - c0] = G.idx[q.y + j] | otag;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nch = min(CG_IDS, total - c0);
    // the loads of round r + 1 are in flight while round r is evaluated out of LDS (a round was: request, wait a whole L2 round trip,
    // write to LDS, evaluate - one after the other; the time of the kernel was linear in the rounds per wave)
    double2 piece[CG_REC / 10];
    auto request = [&](int r0) {
      const int nr = min(CG_REC, nch - r0);
#pragma unroll
      for (int t = 0; t < CG_REC / 10; ++t) {
        const int rr = t * 10 + sub;
        const bool ld = lane < 60 && rr < nr;
        const int id = s_id[wave][ld ? r0 + rr : 0] & CG_KMASK;
        piece[t] = ld ? *(const double2*)(recg + (size_t)id * STRIDE + part * 2) : make_double2(0.0, 0.0);
      }
    };
    if (GL_CG_PIPE) request(0);
    for (int r0 = 0; r0 < nch; r0 += CG_REC) {
      const int nr = min(CG_REC, nch - r0);
      if (!GL_CG_PIPE) request(r0);
#pragma unroll
      for (int t = 0; t < CG_REC / 10; ++t) {
        const int rr = t * 10 + sub;
        if (lane < 60 && rr < nr) *(double2*)(&s_rec[wave][rr * 12 + part * 2]) = piece[t];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (GL_CG_PIPE && r0 + CG_REC < nch) request(r0 + CG_REC);
      if (BAL) {
        // BALANCED evaluation (round 5): candidate r0 + l of the table is evaluated by LANE l against its point, whose coordinates come
        // from the owner's registers (ds_bpermute) - one evaluation per lane and round.  Evaluated by their owners, a round cost as many
        // passes through the ~40 instructions of a pair as its longest list is long, with most lanes masked out: 1 012 vector
        // instructions per wave for 201 pairs, the SIMDs' vector ALUs 73 % busy (profiles/r5n_traffic.json).
        const bool ev = lane < nr;
        const int w = s_id[wave][ev ? r0 + lane : 0];
        const int own = (int)((unsigned)w >> CG_TAG);
        const double ox = __shfl(x, own), oy = __shfl(y, own), oz = __shfl(z, own);
        double rec[12];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          const double2 v = *(const double2*)(&s_rec[wave][(ev ? lane : 0) * 12 + e * 2]);
          rec[2 * e] = v.x;
          rec[2 * e + 1] = v.y;
        }
        const double d = chi2_rec(rec, ox, oy, oz);
        if (ev) s_d[wave][r0 + lane] = d;
      } else {
        const int ja = max(jlo, c0 + r0 - pos), jb = min(jhi, c0 + r0 + nr - pos);
        for (int j = ja; j < jb; ++j) {
          const int r = pos + j - c0 - r0;
          const int k = s_id[wave][r0 + r];
          double rec[12];
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            const double2 v = *(const double2*)(&s_rec[wave][r * 12 + e * 2]);
            rec[2 * e] = v.x;
            rec[2 * e + 1] = v.y;
          }
          upd_min(chi2_rec(rec, x, y, z), k, best, bi);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (BAL) {  // the owner takes the lexicographic minimum over its candidates' (chi2, index): a read and a compare per candidate
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int j = jlo; j < jhi; ++j) {
        const int i = pos + j - c0;
        upd_min(s_d[wave][i], s_id[wave][i] & CG_KMASK, best, bi);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lone && inside && q.x > 3) {  // long list: by the lane itself
    for (int e = q.y; e < q.y + q.x; ++e) {
      const int k = G.idx[e];
      upd_min(chi2_rec(rec12 + (size_t)k * 12, x, y, z), k, best, bi);
    }
  }
  if (!live) return;
  if (best <= G.t_resolve) {
    out_idx[n] = bi;
    if (out_d2) out_d2[n] = best;
  } else {
    out_idx[n] = -1;
    if (out_d2) out_d2[n] = __builtin_inf();
    if (rest_list) rest_list[atomicAdd(rest_count, 1)] = n;
  }
}

__global__ __launch_bounds__(256) void k_index_work(GridDev G, const double* __restrict__ pts, int N,
                                                    unsigned long long* __restrict__ total) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  unsigned long long w = 0;
  if (n < N) {
    w = (unsigned long long)G.nglob;
    const double fx = (pts[(size_t)n * 3] - G.lo[0]) * G.inv_h, fy = (pts[(size_t)n * 3 + 1] - G.lo[1]) * G.inv_h,
                 fz = (pts[(size_t)n * 3 + 2] - G.lo[2]) * G.inv_h;
    if (fx >= 0.0 && fx < (double)G.dim[0] && fy >= 0.0 && fy < (double)G.dim[1] && fz >= 0.0 && fz < (double)G.dim[2]) {
      const int c = ((int)fz * G.dim[1] + (int)fy) * G.dim[0] + (int)fx;
      w += (unsigned long long)(G.ptr[c + 1] - G.ptr[c]);
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) w += __shfl_xor(w, o, 64);
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(total, w);
}

// ---- index build on the device -------------------------------------------------------------------
// The O(K) part (eigenvalues, boxes, grid level, the three bounding forms per component) stays on the host; the
// enumeration - every cell of every component's box against the three forms, ~10^8 tests for 65 536 Gaussians - and the
// ordering of the entries run here: one workgroup per component appends the keys (cell << 32 | component) of the cells
// that pass (wave-aggregated atomics), a radix sort orders them by cell then component (= the ascending lists the
// tie rule of the sweep needs, deterministically), and the CSR pointers are read off the sorted keys.
struct CompReg {
  double mu[3];
  double Minv[18];  // the three forms, sym6 each
  int i0[3], n[3];  // first cell and cell count of the box per axis
  int k, pad;
};
struct GridGeom {
  double lo[3], h;
  int dim[3];
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_index_register(const CompReg* __restrict__ regs, GridGeom gg, unsigned long long* __restrict__ total,
                                                        unsigned long long* __restrict__ keys) {
  __shared__ CompReg r;
  if (threadIdx.x == 0) r = regs[blockIdx.x];
  __syncthreads();
  const int n0 = r.n[0], n01 = r.n[0] * r.n[1], ncell = n01 * r.n[2];
  const int lane = threadIdx.x & 63;
  unsigned long long mine = 0;
  for (int base = 0; base < ncell; base += 256) {
    const int t = base + (int)threadIdx.x;
    bool in = t < ncell;
    unsigned long long key = 0;
    if (in) {
      const int iz = t / n01, rem = t - iz * n01, iy = rem / n0, ix = rem - iy * n0;
      const int cx = r.i0[0] + ix, cy = r.i0[1] + iy, cz = r.i0[2] + iz;
      const double d[3] = {gg.lo[0] + (cx + 0.5) * gg.h - r.mu[0], gg.lo[1] + (cy + 0.5) * gg.h - r.mu[1], gg.lo[2] + (cz + 0.5) * gg.h - r.mu[2]};
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const double* I = r.Minv + b * 6;
        const double q = d[0] * (I[0] * d[0] + I[1] * d[1] + I[2] * d[2]) + d[1] * (I[1] * d[0] + I[3] * d[1] + I[4] * d[2]) +
                         d[2] * (I[2] * d[0] + I[4] * d[1] + I[5] * d[2]);
        in = in && (q <= 1.0 + 1e-9);
      }
      key = ((unsigned long long)(unsigned)((cz * gg.dim[1] + cy) * gg.dim[0] + cx) << 32) | (unsigned)r.k;
    }
    const unsigned long long bal = __ballot(in);
    if (!bal) continue;
    if (!FILL) {
      if (lane == 0) mine += (unsigned long long)__popcll(bal);
    } else {
      unsigned long long pos = 0;
      if (lane == 0) pos = atomicAdd(total, (unsigned long long)__popcll(bal));
      pos = __shfl(pos, 0, 64);
      if (in) keys[pos + __popcll(bal & ((1ull << lane) - 1ull))] = key;
    }
  }
  if (!FILL && lane == 0 && mine) atomicAdd(total, mine);
}
// idx[i] = component of the i-th sorted key;  ptr[c] = first key of a cell >= c
__global__ void k_index_csr(const unsigned long long* __restrict__ keys, size_t nnz, size_t ncell, int32_t* __restrict__ ptr, int32_t* __restrict__ idx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) idx[i] = (int32_t)(unsigned)(keys[i] & 0xffffffffull);
  if (i <= ncell) {
    size_t lo = 0, hi = nnz;
    while (lo < hi) {
      const size_t mid = (lo + hi) >> 1;
      if ((keys[mid] >> 32) < i) lo = mid + 1;
      else hi = mid;
    }
    ptr[i] = (int32_t)lo;
  }
}
// packed cells from the CSR: {count, i0, i1, i2} for lists of up to three, {count, offset, 0, 0} above
__global__ void k_rec_pad(const double* __restrict__ rec12, int K, double* __restrict__ rec16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * 16) return;
  const int k = i >> 4, e = i & 15;
  rec16[i] = e < 12 ? rec12[(size_t)k * 12 + e] : 0.0;
}
__global__ void k_index_pack(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, size_t ncell, int4* __restrict__ cell4) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int e0 = ptr[c], n = ptr[c + 1] - e0;
  int4 q = make_int4(n, 0, 0, 0);
  if (n > 3) {
    q.y = e0;
  } else {
    if (n > 0) q.y = idx[e0];
    if (n > 1) q.z = idx[e0 + 1];
    if (n > 2) q.w = idx[e0 + 2];
  }
  cell4[c] = q;
}
__global__ void k_index_pack8(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, size_t ncell, unsigned long long* __restrict__ cell8) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int e0 = ptr[c], n = ptr[c + 1] - e0;
  unsigned long long w;
  if (n > 3) {
    w = (1ull << 63) | ((unsigned long long)(unsigned)n << 32) | (unsigned long long)(unsigned)e0;
  } else {
    w = (unsigned long long)n;
    if (n > 0) w |= (unsigned long long)(unsigned)idx[e0] << 2;
    if (n > 1) w |= (unsigned long long)(unsigned)idx[e0 + 1] << 22;
    if (n > 2) w |= (unsigned long long)(unsigned)idx[e0 + 2] << 42;
  }
  cell8[c] = w;
}
// sum over the registered components of the list length at their own mean (does the index prune?)
__global__ void k_index_mean_lists(const CompReg* __restrict__ regs, int nreg, GridGeom gg, const int32_t* __restrict__ ptr,
                                   unsigned long long* __restrict__ sum) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long w = 0;
  if (j < nreg) {
    int ci[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) ci[a] = min(gg.dim[a] - 1, max(0, (int)floor((regs[j].mu[a] - gg.lo[a]) / gg.h)));
    const size_t cell = ((size_t)ci[2] * gg.dim[1] + ci[1]) * gg.dim[0] + ci[0];
    w = (unsigned long long)(ptr[cell + 1] - ptr[cell]);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) w += __shfl_xor(w, o, 64);
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(sum, w);
}

// ---- host build ------------------------------------------------------------------------------
// eigenvalues of a symmetric 3x3 (cyclic Jacobi), ascending
void eig3_sym(const double* c, double* w) {
  double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[4], c[5]}, {c[2], c[5], c[8]}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = cs * akp - sn * akq;
          a[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = cs * apk - sn * aqk;
          a[q][k] = sn * apk + cs * aqk;
        }
      }
  }
  w[0] = a[0][0];
  w[1] = a[1][1];
  w[2] = a[2][2];
  std::sort(w, w + 3);
}

// inverse of a symmetric positive definite 3x3 given as sym6 {00 01 02 11 12 22}
void sym3_inv_host(const double* S, double* I) {
  const double c00 = S[3] * S[5] - S[4] * S[4], c01 = S[2] * S[4] - S[1] * S[5], c02 = S[1] * S[4] - S[2] * S[3];
  const double id = 1.0 / (S[0] * c00 + S[1] * c01 + S[2] * c02);
  I[0] = c00 * id;
  I[1] = c01 * id;
  I[2] = c02 * id;
  I[3] = (S[0] * S[5] - S[2] * S[2]) * id;
  I[4] = (S[1] * S[2] - S[0] * S[4]) * id;
  I[5] = (S[0] * S[3] - S[1] * S[1]) * id;
}

}  // namespace

namespace gl {

constexpr double kT0 = 9.0;  // the association gate (gmmloc_opt.cpp:230-232)

void free_cell_index(Gmm* g) {
  if (g->grid.ptr) (void)hipFree(g->grid.ptr);
  if (g->grid.idx) (void)hipFree(g->grid.idx);
  if (g->grid.glob) (void)hipFree(g->grid.glob);
  if (g->grid.cell4) (void)hipFree(g->grid.cell4);
  if (g->grid.cell8) (void)hipFree(g->grid.cell8);
  if (g->grid.rec16) (void)hipFree(g->grid.rec16);
  g->grid = CellIndex();
}

int build_cell_index(Ctx* c, Gmm* g) {
  g->grid = CellIndex();
  // (options of the creating context, read from GMMLOC_<NAME> once at gl_ctx_create like every other knob: assoc_grid,
  // assoc_cell, assoc_globcells, assoc_pack_mb - no environment scan here)
  if (c->opt.assoc_grid == 0) return GL_OK;  // knob: serve GL_ASSOC_BRUTE with the all-pairs sweep
  const int K = g->K;
  const double T = kT0;
  const double t_reg = T * (1.0 + 4e-6);
  std::vector<double> ext((size_t)K * 3), semi((size_t)K * 3);
  std::vector<uint8_t> ok(K, 0);
  std::vector<int32_t> glob;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < K; ++k) {
    const double* cv = &g->h_cov[(size_t)k * 9];
    const double* mu = &g->h_mean[(size_t)k * 3];
    bool fin = true;
    for (int i = 0; i < 9; ++i) fin = fin && std::isfinite(cv[i]);
    for (int i = 0; i < 3; ++i) fin = fin && std::isfinite(mu[i]);
    if (!fin) continue;  // chi2 is NaN for every point: never the argmin of the sweep either
    double w[3];
    eig3_sym(cv, w);
    const double asym = std::fabs(cv[1] - cv[3]) + std::fabs(cv[2] - cv[6]) + std::fabs(cv[5] - cv[7]);
    if (!(w[0] > 0.0) || !(w[2] / w[0] <= 1e8) || asym > 1e-12 * w[2]) {
      glob.push_back(k);  // no usable bound: evaluated for every point
      continue;
    }
    ok[k] = 1;
    for (int a = 0; a < 3; ++a) semi[(size_t)k * 3 + a] = std::sqrt(t_reg * w[a]);  // semi-axes of the T-ellipsoid
    for (int a = 0; a < 3; ++a) {
      ext[(size_t)k * 3 + a] = std::sqrt(t_reg * cv[a * 4]) * (1.0 + 1e-9);
      lo[a] = std::min(lo[a], mu[a] - ext[(size_t)k * 3 + a]);
      hi[a] = std::max(hi[a], mu[a] + ext[(size_t)k * 3 + a]);
    }
  }
  if ((int)glob.size() > K / 4 || lo[0] > hi[0]) return GL_OK;  // index would not pay: stay exhaustive
  double span[3], smax = 0.0;
  for (int a = 0; a < 3; ++a) {
    const double m = 1e-6 * (hi[a] - lo[a]) + 1e-9;
    lo[a] -= m;
    hi[a] += m;
    span[a] = hi[a] - lo[a];
    smax = std::max(smax, span[a]);
  }
  // finest power-of-two subdivision of the longest side within the cell / entry budgets
  const double max_cells = 16.0e6, max_ins = 12.0e6;
  double h = smax;
  for (int lvl = 1; lvl <= 10; ++lvl) {
    const double hc = smax / (double)(1 << lvl);
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) cells *= std::ceil(span[a] / hc);
    if (cells > max_cells) break;
    // entries a component will register ~ volume of its T-ellipsoid grown by the cell's half diagonal, in cells
    // (the axis-aligned box of an oblique thin plane over-counts them 10-50 x and kept the grid needlessly coarse:
    // 11 candidates per point at 0.25 m against 5.4 at 0.125 m on the bench map, 0.98 -> 0.62 ms per 8.19 M points)
    const double rho_c = hc * std::sqrt(3.0) * 0.5;
    double ins = 0.0;
    for (int k = 0; k < K && ins <= max_ins; ++k) {
      if (!ok[k]) continue;
      double vol = 4.0 / 3.0 * M_PI;
      for (int a = 0; a < 3; ++a) vol *= semi[(size_t)k * 3 + a] + rho_c;
      double box = 1.0;
      for (int a = 0; a < 3; ++a) box *= std::floor(2.0 * ext[(size_t)k * 3 + a] / hc) + 2.0;
      ins += std::min(std::min(vol / (hc * hc * hc) + 1.0, box), 4194304.0);
    }
    if (ins > max_ins) break;
    h = hc;
  }
  if (c->opt.assoc_cell > 0) h = c->opt.assoc_cell;  // tuning knob (metres)
  int dim[3];
  for (int a = 0; a < 3; ++a) dim[a] = std::max(1, (int)std::ceil(span[a] / h));
  const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
  const double eps_idx = 1e-9;
  const double rho = h * std::sqrt(3.0) * 0.5 * (1.0 + 1e-9) + 1e-9 * smax;
  double glob_cells = 4194304.0;  // boxes above 2^22 cells: evaluated for every point instead
  if (c->opt.assoc_globcells > 0) glob_cells = c->opt.assoc_globcells;  // tuning knob
  std::vector<CompReg> regs;
  regs.reserve(K);
  const double betas[3] = {0.5, 1.0, 2.0};
  for (int k = 0; k < K; ++k) {
    if (!ok[k]) continue;
    const double* cv = &g->h_cov[(size_t)k * 9];
    const double* mu = &g->h_mean[(size_t)k * 3];
    CompReg r;
    double ncells_k = 1.0;
    for (int a = 0; a < 3; ++a) {
      int i0 = (int)std::floor((mu[a] - ext[(size_t)k * 3 + a] - lo[a]) / h - eps_idx);
      int i1 = (int)std::floor((mu[a] + ext[(size_t)k * 3 + a] - lo[a]) / h + eps_idx);
      i0 = std::max(i0, 0);
      i1 = std::min(i1, dim[a] - 1);
      r.i0[a] = i0;
      r.n[a] = i1 - i0 + 1;
      r.mu[a] = mu[a];
      ncells_k *= (double)r.n[a];
    }
    if (ncells_k > glob_cells) {
      glob.push_back(k);
      continue;
    }
    for (int b = 0; b < 3; ++b) {
      const double s1 = (1.0 + 1.0 / betas[b]) * t_reg, s2 = (1.0 + betas[b]) * rho * rho;
      const double M[6] = {s1 * cv[0] + s2, s1 * cv[1], s1 * cv[2], s1 * cv[4] + s2, s1 * cv[5], s1 * cv[8] + s2};
      sym3_inv_host(M, r.Minv + b * 6);
    }
    r.k = k;
    r.pad = 0;
    regs.push_back(r);
  }
  if ((int)glob.size() > K / 4) return GL_OK;
  std::sort(glob.begin(), glob.end());
  // ---- enumeration, ordering and CSR on the device
  GL_HIP(hipSetDevice(g->device));
  GridGeom gg;
  for (int a = 0; a < 3; ++a) {
    gg.lo[a] = lo[a];
    gg.dim[a] = dim[a];
  }
  gg.h = h;
  const int nreg = (int)regs.size();
  CompReg* d_regs = nullptr;
  unsigned long long *d_cnt = nullptr, *d_keys = nullptr, *d_sorted = nullptr;
  void* d_tmp = nullptr;
  int32_t *d_ptr = nullptr, *d_idx = nullptr;
  auto cleanup = [&]() {
    for (void* p_ : {(void*)d_regs, (void*)d_cnt, (void*)d_keys, (void*)d_sorted, d_tmp})
      if (p_) (void)hipFree(p_);
  };
  auto fail = [&](hipError_t e) {
    cleanup();
    if (d_ptr) (void)hipFree(d_ptr);
    if (d_idx) (void)hipFree(d_idx);
    set_error("cell index build: %s", hipGetErrorString(e));
    return GL_ERR_DEVICE;
  };
#define GL_TRY(x)                            \
  do {                                       \
    const hipError_t e_ = (x);               \
    if (e_ != hipSuccess) return fail(e_);   \
  } while (0)
  hipStream_t st = c->stream;
  unsigned long long nnz = 0, mean_sum = 0;
  GL_TRY(hipMalloc((void**)&d_cnt, 16));
  GL_TRY(hipMemsetAsync(d_cnt, 0, 16, st));
  GL_TRY(hipMalloc((void**)&d_ptr, (ncell + 1) * 4));
  if (nreg) {
    GL_TRY(hipMalloc((void**)&d_regs, sizeof(CompReg) * nreg));
    GL_TRY(hipMemcpyAsync(d_regs, regs.data(), sizeof(CompReg) * nreg, hipMemcpyHostToDevice, st));
    k_index_register<false><<<nreg, 256, 0, st>>>(d_regs, gg, d_cnt, nullptr);
    GL_TRY(hipGetLastError());
    GL_TRY(hipMemcpyAsync(&nnz, d_cnt, 8, hipMemcpyDeviceToHost, st));
    GL_TRY(hipStreamSynchronize(st));
  }
  GL_TRY(hipMalloc((void**)&d_idx, std::max<size_t>(nnz, 1) * 4));
  if (nnz) {
    GL_TRY(hipMalloc((void**)&d_keys, nnz * 8));
    GL_TRY(hipMalloc((void**)&d_sorted, nnz * 8));
    GL_TRY(hipMemsetAsync(d_cnt, 0, 8, st));
    k_index_register<true><<<nreg, 256, 0, st>>>(d_regs, gg, d_cnt, d_keys);
    GL_TRY(hipGetLastError());
    int cell_bits = 1;
    while (((size_t)1 << cell_bits) < ncell) ++cell_bits;
    size_t tmp_bytes = 0;
    GL_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_sorted, (int)nnz, 0, 32 + cell_bits, st));
    GL_TRY(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 1)));
    GL_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_keys, d_sorted, (int)nnz, 0, 32 + cell_bits, st));
  }
  {
    const size_t nthr = std::max<size_t>(nnz, ncell + 1);
    k_index_csr<<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(d_sorted, (size_t)nnz, ncell, d_ptr, d_idx);
    GL_TRY(hipGetLastError());
  }
  if (nreg) {  // the index must actually prune: the mean list length seen at the component means (where the map points
               // are) has to stay far below K, otherwise the plain sweep with scalar-operand records is faster
    k_index_mean_lists<<<(nreg + 255) / 256, 256, 0, st>>>(d_regs, nreg, gg, d_ptr, d_cnt + 1);
    GL_TRY(hipGetLastError());
    GL_TRY(hipMemcpyAsync(&mean_sum, d_cnt + 1, 8, hipMemcpyDeviceToHost, st));
  }
  GL_TRY(hipStreamSynchronize(st));
  cleanup();
  // per evaluated pair the gather kernel is ~20x slower than the sweep (92 G vs 1.8 T pairs/s measured)
  if (nreg > 0 && (double)mean_sum / nreg + (double)glob.size() > K / 24.0) {
    (void)hipFree(d_ptr);
    (void)hipFree(d_idx);
    return GL_OK;
  }
  CellIndex& G = g->grid;
  G.ptr = d_ptr;
  G.idx = d_idx;
  GL_HIP(hipMalloc((void**)&G.glob, std::max<size_t>(glob.size(), 1) * 4));
  if (!glob.empty()) GL_HIP(hipMemcpy(G.glob, glob.data(), glob.size() * 4, hipMemcpyHostToDevice));
#undef GL_TRY
  for (int a = 0; a < 3; ++a) {
    G.lo[a] = lo[a];
    G.dim[a] = dim[a];
  }
  G.h = h;
  G.inv_h = 1.0 / h;
  G.t_resolve = T * (1.0 + 1e-6);
  G.nglob = (int)glob.size();
  G.nnz = (size_t)nnz;
  G.ncell = ncell;
  G.cell4 = nullptr;
  G.cell8 = nullptr;
  // the packed cell table within the option's memory budget (assoc_pack_mb, default 512, 0 = never): 8 bytes per cell where the
  // component indices fit 20 bits and the option assoc_cell8 says so (default; the bench map: 11 M cells = 87 MB), else 16 bytes per
  // cell (174 MB); gl_gmm_index_bytes reports its size
  const bool want8 = c->opt.assoc_cell8 != 0 && g->K < (1 << 20) && nnz < (1ll << 31);
  const double cell_bytes = want8 ? 8.0 : 16.0;
  if ((double)ncell * cell_bytes <= c->opt.assoc_pack_mb * 1048576.0) {
    void* tab = nullptr;
    if (hipMalloc(&tab, ncell * (size_t)cell_bytes) == hipSuccess) {
      if (want8) k_index_pack8<<<(unsigned)((ncell + 255) / 256), 256, 0, c->stream>>>(d_ptr, d_idx, ncell, (unsigned long long*)tab);
      else k_index_pack<<<(unsigned)((ncell + 255) / 256), 256, 0, c->stream>>>(d_ptr, d_idx, ncell, (int4*)tab);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(tab);
        tab = nullptr;
      }
    } else {
      (void)hipGetLastError();
      tab = nullptr;
    }
    (want8 ? G.cell8 : G.cell4) = tab;
  }
  G.rec16 = nullptr;
  if (G.cell4 || G.cell8) {  // one record per 128-byte line for the cooperative gather (K x 128 B: 512 KB for 4 096 Gaussians)
    if (hipMalloc(&G.rec16, (size_t)g->K * 128) == hipSuccess) {
      k_rec_pad<<<(g->K * 16 + 255) / 256, 256, 0, c->stream>>>(g->rec12, g->K, G.rec16);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(G.rec16);
        G.rec16 = nullptr;
      }
    } else {
      (void)hipGetLastError();
      G.rec16 = nullptr;
    }
  }
  G.enabled = true;
  return GL_OK;
}

// | count (64 B) | list N x 4 | (resolve_all: partial minima of the sweep) |
static size_t index_list_bytes(int N) { return (((size_t)N * 4 + 64) + 63) / 64 * 64; }
size_t assoc_index_scratch_bytes(int K, int N, bool resolve_all) {
  return index_list_bytes(N) + (resolve_all ? assoc_scratch_bytes(K, N, true) : 0);
}

static GridDev grid_dev(const CellIndex& I) {
  GridDev G;
  for (int a = 0; a < 3; ++a) {
    G.lo[a] = I.lo[a];
    G.dim[a] = I.dim[a];
  }
  G.inv_h = I.inv_h;
  G.nglob = I.nglob;
  G.t_resolve = I.t_resolve;
  G.ptr = I.ptr;
  G.idx = I.idx;
  G.glob = I.glob;
  G.cell4 = (const int4*)I.cell4;
  G.cell8 = (const unsigned long long*)I.cell8;
  return G;
}

// idx / d2 (d2 may be NULL) for N points.  With resolve_all the unresolved points are swept
// exhaustively (exact GL_ASSOC_BRUTE result); without, they are reported as -1 / +inf (callers that
// drop chi2 > 9 anyway).  scratch: assoc_index_scratch_bytes(K, N, resolve_all).
int launch_assoc_index(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2, bool resolve_all,
                       void* scratch) {
  const GridDev G = grid_dev(g->grid);
  int32_t* count = (int32_t*)scratch;
  int32_t* list = count + 16;
  TimerScope ts(c, GL_TIMER_ASSOC);
  if (resolve_all) GL_HIP(hipMemsetAsync(count, 0, 4, c->stream));
  if ((G.cell4 || G.cell8) && c->opt.assoc_coop != 0 && N >= 4096) {
    const bool pad = g->grid.rec16 && c->opt.assoc_rec_pad != 0;
    const double* recg = pad ? g->grid.rec16 : g->rec12;
    int32_t* rl = resolve_all ? list : nullptr;
    const dim3 gr((N + 255) / 256);
    if (c->opt.assoc_coop_long != 0 && GL_CG_BAL && g->K < (1 << 26) && c->opt.assoc_coop_bal != 0) {
      if (pad) k_assoc_cells_coop<16, true, true><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
      else k_assoc_cells_coop<12, true, true><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
    } else if (c->opt.assoc_coop_long != 0) {
      if (pad) k_assoc_cells_coop<16, true, false><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
      else k_assoc_cells_coop<12, true, false><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
    } else {
      if (pad) k_assoc_cells_coop<16, false, false><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
      else k_assoc_cells_coop<12, false, false><<<gr, 256, 0, c->stream>>>(g->rec12, recg, G, pts, N, idx, d2, rl, count);
    }
  } else
    k_assoc_cells<<<(N + 255) / 256, 256, 0, c->stream>>>(g->rec12, G, pts, N, idx, d2, resolve_all ? list : nullptr, count);
  GL_HIP(hipGetLastError());
  if (resolve_all)  // the unresolved points go through the all-pairs sweep (grid sized for N, empty tiles exit)
    return launch_assoc_sweep(c, g, pts, N, idx, d2, list, count, (char*)scratch + index_list_bytes(N));
  return GL_OK;
}

}  // namespace gl

extern "C" {

int gl_gmm_index_bytes(const gl_gmm_t* gmm, double bytes[3]) {
  GL_REQUIRE(gmm && bytes, "null argument");
  const gl::CellIndex& I = gl::G(gmm)->grid;
  bytes[0] = I.enabled ? (double)(I.ncell + 1) * 4.0 : 0.0;  // cell pointers
  bytes[1] = I.enabled ? (double)I.nnz * 4.0 + (double)I.nglob * 4.0 : 0.0;  // candidate lists
  bytes[2] = I.cell8 ? (double)I.ncell * 8.0 : I.cell4 ? (double)I.ncell * 16.0 : 0.0;  // packed cell table (options assoc_pack_mb, assoc_cell8)
  return GL_OK;
}

int gl_gmm_index_info(const gl_gmm_t* gmm, double info[8]) {
  GL_REQUIRE(gmm && info, "null argument");
  const gl::CellIndex& I = gl::G(gmm)->grid;
  info[0] = I.enabled ? 1.0 : 0.0;
  info[1] = I.h;
  info[2] = I.dim[0];
  info[3] = I.dim[1];
  info[4] = I.dim[2];
  info[5] = (double)I.nnz;
  info[6] = I.nglob;
  info[7] = I.t_resolve;
  return GL_OK;
}

int gl_assoc_index_work(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int64_t* pairs_dev) {
  GL_REQUIRE(ctx && gmm && pairs_dev, "null argument");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  GL_HIP(hipMemsetAsync(pairs_dev, 0, 8, c->stream));
  if (N == 0 || !g->grid.enabled) return GL_OK;
  GL_REQUIRE(N > 0 && pts_dev, "bad N / pts");
  k_index_work<<<(N + 255) / 256, 256, 0, c->stream>>>(gl::grid_dev(g->grid), pts_dev, N, (unsigned long long*)pairs_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // extern "C"
