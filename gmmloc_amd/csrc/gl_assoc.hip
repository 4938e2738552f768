// Association kernels (gfx950, wave64).
//
//  k_assoc_brute : for every point, argmin_k of GaussianComponent::chi2
//                  (gaussian.cpp:65-70) over a K-range.  Every lane owns PPT points in
//                  registers; the {mean, cov_inv} record (96 B) of the current Gaussian is
//                  wave-uniform and comes in through scalar loads as SGPR operands.
//                  The kernel is VALU-issue bound (every wave64 instruction occupies its SIMD
//                  for 4 cycles): 15 v_{mul,fma,add}_f64 + 1 v_min_f64 per (point, Gaussian)
//                  pair, exact argmin recovered per 8-Gaussian chunk (see the kernel).
//                  History: LDS-staged tiles + compare/select were 20.9 slots/pair; a wave-vote
//                  guard around the selects was slower still (the branch serialises issue).
//  grid = (point tiles, K splits); K is split so that even one frame (2k
//  points) produces >= ~2k waves for the 1024 SIMDs; partial minima are
//  merged by k_assoc_merge in ascending-k order, so ties keep the lowest index
//  exactly like the sequential CPU loop.
//
//  k_knn3d       : exact k-NN (k <= 8) on the means, ascending squared L2,
//                  ties by lower index (GMM::queryPoint's knnSearch,
//                  gaussian_mixture.cpp:553-558); a thread per query, or a wave per
//                  query (k_knn3d_wave) when the queries are few.
//
// Compiled with -ffp-contract=off; the only fused ops are the explicit fma()
// of the canonical chi2 (gl_device.hpp) -> bit-identical to the fp64 CPU order.
#include <cstdlib>

#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {


// The record of Gaussian k is the same for every lane, so it is fetched with SCALAR loads
// (constant address space -> s_load_dwordx8/x16 through the scalar cache) and used as the SGPR
// operand of the VALU instructions: no LDS staging, no barriers, no VGPRs for the record.
typedef const double __attribute__((address_space(4))) cdouble;


// Per (point, Gaussian) pair the loop issues the 15 canonical v_{add,mul,fma}_f64 + ONE v_min_f64.
// The argmin index is recovered exactly afterwards: per chunk of 8 Gaussians only {chunk-min < best}
// is tracked (3 instructions per chunk), and the winning chunk is re-evaluated at the end with the
// same operation sequence (bit-identical values) to find the first k whose chi2 equals the minimum
// -- the same lowest-index tie rule as the sequential compare/select loop it replaces
// (GaussianComponent::chi2 sweep, gaussian.cpp:65-70), at 16.4 instead of 20.9 issue slots per pair.
template <int PPT, int kChunk>
__global__ __launch_bounds__(256) void k_assoc_brute(const double* __restrict__ rec12, int K, int kchunk,
                                                     const double* __restrict__ pts, int Nstride,
                                                     double* __restrict__ out_d2, int32_t* __restrict__ out_idx,
                                                     const int32_t* __restrict__ list,
                                                     const int32_t* __restrict__ count_dev) {
  // optional indirection: sweep only the points list[0 .. *count_dev) (the ones the cell index left
  // unresolved); partial results are stored compactly (entry i belongs to point list[i])
  const int N = count_dev ? *count_dev : Nstride;
  if (blockIdx.x * 256 * PPT >= N) return;
  const int tid = threadIdx.x;
  const int k_begin = blockIdx.y * kchunk;
  const int k_end = min(K, k_begin + kchunk);
  const int p0 = (blockIdx.x * 256 + tid) * PPT;
  cdouble* rc = (cdouble*)rec12;

  double px[PPT], py[PPT], pz[PPT], best[PPT];
  int bc[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    int n = min(p0 + p, N - 1);
    if (list) n = list[n];
    px[p] = pts[(size_t)n * 3 + 0];
    py[p] = pts[(size_t)n * 3 + 1];
    pz[p] = pts[(size_t)n * 3 + 2];
    best[p] = __builtin_inf();
    bc[p] = -1;
  }

  auto pair_chi2 = [&](cdouble* rec, int p) {
    const double d0 = px[p] - rec[0], d1 = py[p] - rec[1], d2 = pz[p] - rec[2];
    const double r0 = fma(d2, rec[9], fma(d1, rec[6], d0 * rec[3]));
    const double r1 = fma(d2, rec[10], fma(d1, rec[7], d0 * rec[4]));
    const double r2 = fma(d2, rec[11], fma(d1, rec[8], d0 * rec[5]));
    return fma(r2, d2, fma(r1, d1, r0 * d0));
  };
  for (int k0 = k_begin; k0 < k_end; k0 += kChunk) {
    double cmin[PPT];
    cdouble* rec = rc + (size_t)k0 * 12;
    if (k0 + kChunk <= k_end) {  // whole chunk: straight-line code, the scheduler hoists the scalar loads
#pragma unroll
      for (int p = 0; p < PPT; ++p) cmin[p] = pair_chi2(rec, p);
#pragma unroll
      for (int g = 1; g < kChunk; ++g)
#pragma unroll
        for (int p = 0; p < PPT; ++p) cmin[p] = __builtin_fmin(cmin[p], pair_chi2(rec + g * 12, p));  // NaN-ignoring, like `d < best`
    } else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) cmin[p] = pair_chi2(rec, p);
      for (int g = 1; g < k_end - k0; ++g)
#pragma unroll
        for (int p = 0; p < PPT; ++p) cmin[p] = __builtin_fmin(cmin[p], pair_chi2(rec + g * 12, p));
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const bool lt = cmin[p] < best[p];
      best[p] = __builtin_fmin(best[p], cmin[p]);
      bc[p] = lt ? k0 : bc[p];
    }
  }
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int n = p0 + p;
    if (n >= N) continue;
    int bi = -1;
    double bd = __builtin_inf();
    if (bc[p] >= 0) {
      for (int g = kChunk - 1; g >= 0; --g) {  // descending: the lowest matching index is kept
        const int kk = bc[p] + g;
        if (kk < k_end) {
          const double d = chi2_rec(rec12 + (size_t)kk * 12, px[p], py[p], pz[p]);
          if (d == best[p]) {
            bi = kk;
            bd = d;
          }
        }
      }
    }
    out_d2[(size_t)blockIdx.y * Nstride + n] = bd;
    out_idx[(size_t)blockIdx.y * Nstride + n] = bi;
  }
}

// merge the per-K-split partial minima in ascending split (= ascending k) order
__global__ void k_assoc_merge(const double* __restrict__ part_d2, const int32_t* __restrict__ part_idx, int nsplit,
                              int Nstride, int32_t* __restrict__ idx, double* __restrict__ d2,
                              const int32_t* __restrict__ list, const int32_t* __restrict__ count_dev) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = count_dev ? *count_dev : Nstride;
  if (n >= N) return;
  double best = __builtin_inf();
  int bi = -1;
  for (int s = 0; s < nsplit; ++s) {
    const double d = part_d2[(size_t)s * Nstride + n];
    if (d < best) {
      best = d;
      bi = part_idx[(size_t)s * Nstride + n];
    }
  }
  const int o = list ? list[n] : n;
  idx[o] = bi;
  if (d2) d2[o] = best;
}

// exact k-NN on the means: one thread per query, means staged in LDS tiles.
template <int KNN>
__global__ __launch_bounds__(256) void k_knn3d(const double* __restrict__ mean, int K, const double* __restrict__ pts,
                                               int N, int32_t* __restrict__ out_idx, double* __restrict__ out_dist) {
  constexpr int TG = 512;
  __shared__ double tile[TG * 3];
  const int tid = threadIdx.x;
  const int n = blockIdx.x * 256 + tid;
  const int nn = min(n, N - 1);
  const double qx = pts[(size_t)nn * 3 + 0], qy = pts[(size_t)nn * 3 + 1], qz = pts[(size_t)nn * 3 + 2];
  double dist[KNN];
  int idx[KNN];
#pragma unroll
  for (int i = 0; i < KNN; ++i) {
    dist[i] = __builtin_inf();
    idx[i] = -1;
  }
  for (int k0 = 0; k0 < K; k0 += TG) {
    const int ng = min(TG, K - k0);
    __syncthreads();
    for (int i = tid; i < ng * 3; i += 256) tile[i] = mean[(size_t)k0 * 3 + i];
    __syncthreads();
    for (int g = 0; g < ng; ++g) {
      const double d0 = qx - tile[g * 3 + 0], d1 = qy - tile[g * 3 + 1], d2 = qz - tile[g * 3 + 2];
      const double d = (d0 * d0 + d1 * d1) + d2 * d2;  // kdtree_distance, gaussian_mixture.h:33-39
      if (d < dist[KNN - 1]) {
        // insert after every entry with dist <= d (KNNResultSet::addPoint order)
        double cd = d;
        int ci = k0 + g;
#pragma unroll
        for (int i = 0; i < KNN; ++i) {
          const bool sw = cd < dist[i];
          const double td = sw ? dist[i] : cd;
          const int ti = sw ? idx[i] : ci;
          dist[i] = sw ? cd : dist[i];
          idx[i] = sw ? ci : idx[i];
          cd = td;
          ci = ti;
        }
      }
    }
  }
  if (n < N) {
#pragma unroll
    for (int i = 0; i < KNN; ++i) {
      out_idx[(size_t)n * KNN + i] = idx[i];
      if (out_dist) out_dist[(size_t)n * KNN + i] = dist[i];
    }
  }
}

// Few queries (the reference asks for one point at a time): a WAVE per query.  Lane l keeps the KNN best of the
// components l, l + 64, ... (visited in ascending index, inserted after the equal ones), then KNN rounds of a
// lexicographic (distance, index) wave argmin pop the winners - the same ascending-distance, lower-index-first
// order.  One thread scanning 3 299 means took 0.29 ms.
template <int KNN>
__global__ __launch_bounds__(256) void k_knn3d_wave(const double* __restrict__ mean, int K, const double* __restrict__ pts,
                                                    int N, int32_t* __restrict__ out_idx, double* __restrict__ out_dist) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;  // the whole wave
  const double qx = pts[(size_t)n * 3 + 0], qy = pts[(size_t)n * 3 + 1], qz = pts[(size_t)n * 3 + 2];
  double dist[KNN];
  int idx[KNN];
#pragma unroll
  for (int i = 0; i < KNN; ++i) {
    dist[i] = __builtin_inf();
    idx[i] = 0x7fffffff;
  }
  for (int g = lane; g < K; g += 64) {
    const double d0 = qx - mean[(size_t)g * 3 + 0], d1 = qy - mean[(size_t)g * 3 + 1], d2 = qz - mean[(size_t)g * 3 + 2];
    const double d = (d0 * d0 + d1 * d1) + d2 * d2;  // kdtree_distance, gaussian_mixture.h:33-39
    if (d < dist[KNN - 1]) {
      bool c[KNN];
#pragma unroll
      for (int i = 0; i < KNN; ++i) c[i] = d < dist[i];  // monotone: dist is ascending
#pragma unroll
      for (int i = KNN - 1; i >= 0; --i) {
        const bool prev = i > 0 && c[i > 0 ? i - 1 : 0];
        const double nd = prev ? dist[i > 0 ? i - 1 : 0] : d;
        const int ni = prev ? idx[i > 0 ? i - 1 : 0] : g;
        dist[i] = c[i] ? nd : dist[i];
        idx[i] = c[i] ? ni : idx[i];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KNN; ++r) {
    double bd = dist[0];
    int bi = idx[0];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double od = shfl_xor_f64(bd, o);
      const int oi = __shfl_xor(bi, o, 64);
      const bool t = od < bd || (od == bd && oi < bi);
      bd = t ? od : bd;
      bi = t ? oi : bi;
    }
    if (lane == 0) {
      out_idx[(size_t)n * KNN + r] = bi == 0x7fffffff ? -1 : bi;
      if (out_dist) out_dist[(size_t)n * KNN + r] = bd;
    }
    const bool pop = idx[0] == bi && bi != 0x7fffffff;  // the lane that owns the winner drops it
#pragma unroll
    for (int i = 0; i < KNN - 1; ++i) {
      dist[i] = pop ? dist[i + 1] : dist[i];
      idx[i] = pop ? idx[i + 1] : idx[i];
    }
    if (pop) {
      dist[KNN - 1] = __builtin_inf();
      idx[KNN - 1] = 0x7fffffff;
    }
  }
}

// queryPoint: nearest mean (ret_index[0]) + its chi2 (gaussian_mixture.cpp:545-576)
__global__ void k_nearest_chi2(const int32_t* __restrict__ knn_idx, int knn, const double* __restrict__ rec12,
                               const double* __restrict__ pts, int N, int32_t* __restrict__ idx,
                               double* __restrict__ d2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int k = knn_idx[(size_t)n * knn];
  idx[n] = k;
  if (d2) d2[n] = k >= 0 ? chi2_rec(rec12 + (size_t)k * 12, pts[(size_t)n * 3], pts[(size_t)n * 3 + 1], pts[(size_t)n * 3 + 2]) : __builtin_inf();
}

}  // namespace

namespace gl {

// Launch shape (tools/tune_assoc.py sweep on MI355X, profiles/history/r1_assoc_tuning.txt):
//  * points per thread: 4 amortises the broadcast LDS reads best once there are enough points;
//  * K splits: aim for ~4096 workgroups so the tail is short; a single 2 000-point frame still
//    gets 8 x 64 workgroups.
// launch-shape tuning knobs of tools/tune_assoc.py (GMMLOC_ASSOC_CHUNK / _PPT / _NSPLIT): process-wide, read from
// the environment ONCE (first use), never on the call path
struct AssocTune {
  int chunk = 8, ppt = 0, nsplit = 0;
};
static const AssocTune& assoc_tune() {
  static const AssocTune t = [] {
    AssocTune a;
    if (const char* e = getenv("GMMLOC_ASSOC_CHUNK")) a.chunk = atoi(e) == 16 ? 16 : 8;
    if (const char* e = getenv("GMMLOC_ASSOC_PPT")) a.ppt = atoi(e);
    if (const char* e = getenv("GMMLOC_ASSOC_NSPLIT")) a.nsplit = atoi(e);
    return a;
  }();
  return t;
}
static int assoc_minchunk() { return assoc_tune().chunk; }
// `listed`: the sweep of a point LIST whose length only the device knows (the points the cell index left unresolved,
// typically 5 - 15 % of N).  The grid must cover N, but the shape is chosen for N / 16 points so that the few point
// tiles that have work still make ~1 000 workgroups (config 5: 2 500 of 50 000 points x 65 536 Gaussians were 252
// workgroups of 1 024 points, one per CU: 0.40 ms for a tenth of a millisecond of arithmetic).
static void assoc_shape(int K, int N, bool listed, int* ppt_o, int* ptiles_o, int* nsplit_o, int* kchunk_o) {
  const int kChunk = assoc_minchunk();
  const int Ne = listed ? std::max(N / 16, 256) : N;
  int ppt = 1;
  if (Ne >= 8192) ppt = 2;
  if (Ne >= 16384) ppt = 4;
  if (assoc_tune().ppt > 0) ppt = assoc_tune().ppt;  // tuning knob (1, 2 or 4)
  const int ptiles = (N + 256 * ppt - 1) / (256 * ppt);
  const int etiles = (Ne + 256 * ppt - 1) / (256 * ppt);  // tiles expected to have work
  const int target_blocks = (Ne >= 8192) ? 4096 : (listed ? 1024 : 512);
  int nsplit = (target_blocks + etiles - 1) / etiles;
  const int max_split = (K + 63) / 64;  // >= 64 Gaussians per split
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  if (assoc_tune().nsplit > 0) nsplit = assoc_tune().nsplit;  // tuning knob
  int kchunk = (K + nsplit - 1) / nsplit;
  kchunk = (kchunk + kChunk - 1) / kChunk * kChunk;  // whole min-chunks per split
  nsplit = (K + kchunk - 1) / kchunk;
  *ppt_o = ppt;
  *ptiles_o = ptiles;
  *nsplit_o = nsplit;
  *kchunk_o = kchunk;
}

size_t assoc_scratch_bytes(int K, int N, bool listed) {
  int ppt, ptiles, nsplit, kchunk;
  assoc_shape(K, N, listed, &ppt, &ptiles, &nsplit, &kchunk);
  return (size_t)nsplit * N * 12 + 64;
}

// shared with gl_ba.hip: brute association of N points; outputs on device
int launch_assoc_brute(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2) {
  return launch_assoc_sweep(c, g, pts, N, idx, d2, nullptr, nullptr, nullptr);
}

// all-pairs sweep of N points, or (list / count_dev given) of the listed subset; `scratch` (may be
// NULL: taken from the context) must hold assoc_scratch_bytes(K, N, list != NULL)
int launch_assoc_sweep(Ctx* c, const Gmm* g, const double* pts, int N, int32_t* idx, double* d2, const int32_t* list,
                       const int32_t* count_dev, void* scratch) {
  const int K = g->K;
  int ppt, ptiles, nsplit, kchunk;
  assoc_shape(K, N, list != nullptr, &ppt, &ptiles, &nsplit, &kchunk);
  double* part_d2;
  int32_t* part_idx;
  if (nsplit > 1 || !d2 || list) {
    const size_t bytes = (size_t)nsplit * N * 12 + 64;
    if (!scratch) {
      int rc = ctx_scratch(c, bytes, &scratch);
      if (rc != GL_OK) return rc;
    }
    part_d2 = (double*)scratch;
    part_idx = (int32_t*)((char*)scratch + (size_t)nsplit * N * 8);
  } else {
    part_d2 = d2;
    part_idx = idx;
  }
  {
    TimerScope ts(c, GL_TIMER_ASSOC);
    const dim3 grid(ptiles, nsplit);
    const int ch = assoc_minchunk();
#define GL_ASSOC_LAUNCH(P, C) \
  k_assoc_brute<P, C><<<grid, 256, 0, c->stream>>>(g->rec12, K, kchunk, pts, N, part_d2, part_idx, list, count_dev)
    if (ch == 16) {
      if (ppt == 1) GL_ASSOC_LAUNCH(1, 16);
      else if (ppt == 2) GL_ASSOC_LAUNCH(2, 16);
      else if (ppt == 8) GL_ASSOC_LAUNCH(8, 16);
      else GL_ASSOC_LAUNCH(4, 16);
    } else {
      if (ppt == 1) GL_ASSOC_LAUNCH(1, 8);
      else if (ppt == 2) GL_ASSOC_LAUNCH(2, 8);
      else if (ppt == 8) GL_ASSOC_LAUNCH(8, 8);
      else GL_ASSOC_LAUNCH(4, 8);
    }
#undef GL_ASSOC_LAUNCH
  }
  GL_HIP(hipGetLastError());
  if (part_idx != idx) {
    k_assoc_merge<<<(N + 255) / 256, 256, 0, c->stream>>>(part_d2, part_idx, nsplit, N, idx, d2, list, count_dev);
    GL_HIP(hipGetLastError());
  }
  return GL_OK;
}

}  // namespace gl

extern "C" {

int gl_knn3d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int k, int32_t* idx_dev,
             double* dist_dev) {
  GL_REQUIRE(ctx && gmm, "null argument");
  GL_REQUIRE(k >= 1 && k <= 8, "k must be in [1, 8]");
  if (N == 0) return GL_OK;
  GL_REQUIRE(idx_dev, "null argument");
  GL_REQUIRE(N > 0 && pts_dev, "bad N / pts");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  const int grid = (N + 255) / 256;
  const bool few = N <= 16 * c->ncu;  // a wave per query while the waves do not fill the chip
#define GL_KNN_CASE(KK)                                                                                        \
  case KK:                                                                                                     \
    if (few)                                                                                                   \
      k_knn3d_wave<KK><<<(N + 3) / 4, 256, 0, c->stream>>>(g->mean, g->K, pts_dev, N, idx_dev, dist_dev);        \
    else                                                                                                       \
      k_knn3d<KK><<<grid, 256, 0, c->stream>>>(g->mean, g->K, pts_dev, N, idx_dev, dist_dev);                  \
    break;
  switch (k) {
    GL_KNN_CASE(1)
    GL_KNN_CASE(2)
    GL_KNN_CASE(3)
    GL_KNN_CASE(4)
    GL_KNN_CASE(5)
    GL_KNN_CASE(6)
    GL_KNN_CASE(7)
    GL_KNN_CASE(8)
  }
#undef GL_KNN_CASE
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_associate3d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const double* pts_dev, int N, int mode, int32_t* idx_dev,
                   double* d2_dev) {
  GL_REQUIRE(ctx && gmm, "null argument");
  GL_REQUIRE(mode == GL_ASSOC_BRUTE || mode == GL_ASSOC_KNN5_EUCLID || mode == GL_ASSOC_EXHAUSTIVE, "unknown mode");
  if (N == 0) return GL_OK;
  GL_REQUIRE(idx_dev, "null argument");
  GL_REQUIRE(N > 0 && pts_dev, "bad N / pts");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  // small problems are launch-bound: the sweep's two launches beat index + list + sweep of the rest
  double min_pairs = 6.7e7;
  if (c->opt.assoc_index_min >= 0) min_pairs = c->opt.assoc_index_min;  // option (tests force the index with 0)
  const bool small = (double)N * g->K < min_pairs;
  if (mode == GL_ASSOC_EXHAUSTIVE || (mode == GL_ASSOC_BRUTE && (!g->grid.enabled || c->opt.assoc_grid == 0 || small)))
    return gl::launch_assoc_brute(c, g, pts_dev, N, idx_dev, d2_dev);
  void* scratch = nullptr;
  if (mode == GL_ASSOC_BRUTE) {  // same result through the exact cell index (gl_grid.hip)
    int rc = gl::ctx_scratch(c, gl::assoc_index_scratch_bytes(g->K, N, true), &scratch);
    if (rc != GL_OK) return rc;
    return gl::launch_assoc_index(c, g, pts_dev, N, idx_dev, d2_dev, true, scratch);
  }
  int rc = gl::ctx_scratch(c, (size_t)N * 5 * 4, &scratch);
  if (rc != GL_OK) return rc;
  if (N <= 16 * c->ncu)
    k_knn3d_wave<5><<<(N + 3) / 4, 256, 0, c->stream>>>(g->mean, g->K, pts_dev, N, (int32_t*)scratch, nullptr);
  else
    k_knn3d<5><<<(N + 255) / 256, 256, 0, c->stream>>>(g->mean, g->K, pts_dev, N, (int32_t*)scratch, nullptr);
  GL_HIP(hipGetLastError());
  k_nearest_chi2<<<(N + 255) / 256, 256, 0, c->stream>>>((int32_t*)scratch, 5, g->rec12, pts_dev, N, idx_dev, d2_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // extern "C"
