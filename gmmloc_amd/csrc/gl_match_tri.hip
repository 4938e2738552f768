// ORBmatcher::searchForTriangulation (orb_matcher.cpp:141-293) with checkEpipolarDist (:119-139) and computeThreeMaxima
// (:544-578) for B key-frame pairs: the producer of the matches Localization::createMapPoints triangulates
// (localization_opt.cpp:266; gl_create_map_points takes them from here).  Integer / byte work: per vocabulary node the two
// key-frames share, 256-bit Hamming distances between its features of key-frame 1 and of key-frame 2, the epipole test of
// mono pairs, the epipolar chi2, then the rotation histogram.
//
// The reference loop is ORDER DEPENDENT like searchByProjection's (gl_match.hip): a feature of key-frame 2 taken by an
// earlier feature of key-frame 1 - earlier node, or earlier in the node's list - is skipped by every later one, which falls
// back to its next-best partner.  Same fixed point: queries are numbered by their position in key-frame 1's feature-vector
// list (= the order the reference visits them in: std::map iterates node ids ascending, the lists are in push order); in every
// round each query picks its best partner among those not owned by a LOWER query in the previous round, owner[partner] = min
// query that picked it; the choices of the queries 0 .. r-1 are final after round r, and the iteration stops when the owner
// table repeats.  "Best" is the reference's scan: minimum descriptor distance among the partners that pass the geometric
// tests, the LAST one of equal distance (`dist > bestDist` only skips worse candidates) - the geometric tests do not depend on
// the state of the scan, so this is what the sequential scan returns.
// The DBoW2 feature vectors arrive as CSR (node ids ascending, node_ptr, node_idx in list order); the fundamental matrix and
// the epipole are the host's (Eigen: K^-T E K^-1 and quaternion products, MathUtils::computeFundamentalMatrix).
// Every float / double conversion of the reference is kept (the file is compiled without contraction).
#include <climits>

#include "gl_internal.hpp"

namespace {

constexpr int T_T = 512;

struct TriP {
  int N1, N2, NN1, NN2, only_stereo, check_orientation;
  float sf[8], sigma2[8];
};

__device__ __forceinline__ int hamming256(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) d += __popc(a[w] ^ b[w]);
  return d;
}

// largest i with ptr[i] <= a  (the node of list entry a)
__device__ __forceinline__ int node_of(const int32_t* __restrict__ ptr, int nn, int a) {
  int lo = 0, hi = nn;  // ptr[lo] <= a < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= a) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(T_T) void k_search_for_triangulation(
    TriP P, int B, const double* __restrict__ uv1_all, const float* __restrict__ ur1_all, const int32_t* __restrict__ oct1_all,
    const float* __restrict__ angle1_all, const uint8_t* __restrict__ desc1_all, const uint8_t* __restrict__ mp1_all,
    const int32_t* __restrict__ nn1_all, const int32_t* __restrict__ nid1_all, const int32_t* __restrict__ nptr1_all,
    const int32_t* __restrict__ nidx1_all, const double* __restrict__ uv2_all, const float* __restrict__ ur2_all,
    const int32_t* __restrict__ oct2_all, const float* __restrict__ angle2_all, const uint8_t* __restrict__ desc2_all,
    const uint8_t* __restrict__ mp2_all, const int32_t* __restrict__ nn2_all, const int32_t* __restrict__ nid2_all,
    const int32_t* __restrict__ nptr2_all, const int32_t* __restrict__ nidx2_all, const double* __restrict__ fmat_all,
    const float* __restrict__ epi_all, int32_t* __restrict__ match_all, int32_t* __restrict__ nmatches_all, int32_t* __restrict__ counters, uint4* __restrict__ cache_all) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int N1 = P.N1, N2 = P.N2;
  int32_t* owner = lds;             // N2: lowest query that picked the feature in the previous round (-1: not available at all)
  int32_t* owner_n = owner + N2;    // N2: being rebuilt
  int32_t* choice = owner_n + N2;   // N1 (by query)
  int32_t* q_idx1 = choice + N1;    // N1: the query's feature of key-frame 1, or -1 (not a query: map point, mono under only-stereo, node not shared)
  int32_t* q_lo = q_idx1 + N1;      // N1: its partners = node_idx2[q_lo .. q_hi)
  int32_t* q_hi = q_lo + N1;
  // the three tables the queries are set up from (two binary searches per query: sixteen dependent GLOBAL loads each before round 5)
  int32_t* s_nptr1 = q_hi + N1;            // NN1 + 1
  int32_t* s_nid2 = s_nptr1 + P.NN1 + 1;   // NN2
  int32_t* s_nptr2 = s_nid2 + P.NN2;       // NN2 + 1
  __shared__ int s_changed, s_hist[32], s_keep[4], s_cnt[T_T / 64];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= B) return;
  const double* uv1 = uv1_all + (size_t)f * N1 * 2;
  const float* ur1 = ur1_all + (size_t)f * N1;
  const int32_t* oct1 = oct1_all + (size_t)f * N1;
  const uint32_t* desc1 = (const uint32_t*)(desc1_all + (size_t)f * N1 * 32);
  const uint8_t* mp1 = mp1_all + (size_t)f * N1;
  const double* uv2 = uv2_all + (size_t)f * N2 * 2;
  const float* ur2 = ur2_all + (size_t)f * N2;
  const int32_t* oct2 = oct2_all + (size_t)f * N2;
  const uint32_t* desc2 = (const uint32_t*)(desc2_all + (size_t)f * N2 * 32);
  const uint8_t* mp2 = mp2_all + (size_t)f * N2;
  const int nn1 = min(nn1_all[f], P.NN1), nn2 = min(nn2_all[f], P.NN2);
  const int32_t* nid1 = nid1_all + (size_t)f * P.NN1;
  const int32_t* nptr1 = nptr1_all + (size_t)f * (P.NN1 + 1);
  const int32_t* nidx1 = nidx1_all + (size_t)f * N1;
  const int32_t* nid2 = nid2_all + (size_t)f * P.NN2;
  const int32_t* nptr2 = nptr2_all + (size_t)f * (P.NN2 + 1);
  const int32_t* nidx2 = nidx2_all + (size_t)f * N2;
  const double* F = fmat_all + (size_t)f * 9;
  const float ex = epi_all[2 * f], ey = epi_all[2 * f + 1];
  const int nq = nn1 > 0 ? min(nptr1[nn1], N1) : 0;  // list entries of key-frame 1 = queries, in the reference's visiting order

  // ---- queries: list entry a of key-frame 1 -> its feature and the partner list of the same node in key-frame 2 ----
  for (int i = tid; i <= nn1; i += T_T) s_nptr1[i] = nptr1[i];
  for (int i = tid; i <= nn2; i += T_T) {
    s_nptr2[i] = nptr2[i];
    if (i < nn2) s_nid2[i] = nid2[i];
  }
  __syncthreads();
  for (int a = tid; a < N1; a += T_T) {
    int idx1 = -1, lo = 0, hi = 0;
    if (a < nq) {
      const int n1 = node_of(s_nptr1, nn1, a);
      const int id = nid1[n1];
      int l = 0, h = nn2;  // lower_bound of id in nid2
      while (l < h) {
        const int mid = (l + h) >> 1;
        if (s_nid2[mid] < id) l = mid + 1;
        else h = mid;
      }
      if (l < nn2 && s_nid2[l] == id) {
        const int i1 = nidx1[a];
        if (i1 >= 0 && i1 < N1 && oct1[i1] >= 0 && !mp1[i1] && !(P.only_stereo && !(ur1[i1] >= 0))) {
          idx1 = i1;
          lo = s_nptr2[l];
          hi = min(s_nptr2[l + 1], N2);
        }
      }
    }
    q_idx1[a] = idx1;
    q_lo[a] = lo;
    q_hi[a] = hi;
    choice[a] = -1;
  }
  for (int i = tid; i < N2; i += T_T) owner[i] = (oct2[i] >= 0 && !mp2[i] && !(P.only_stereo && !(ur2[i] >= 0))) ? INT_MAX : -1;
  __syncthreads();

  // ---- rounds of the fixed point ------------------------------------------------------------------------------------
  // Round 5: what a query can ever take is fixed by the tests that do not depend on the owners - the partners of its vocabulary node
  // with a distance <= TH_LOW that pass the epipole / epipolar tests; the owners only REMOVE partners.  The reference keeps, of
  // those, the smallest distance and among equal distances the LAST in list order: the smallest key
  //     dist << 26 | (1023 - position in the partner list) << 16 | feature of key-frame 2.
  // Round 1 evaluates every partner once (descriptors requested four at a time) and leaves the query's three smallest keys and
  // their number in a 16-byte record; a later round takes the first key whose feature no lower query owns, and walks again only if
  // all three are gone and there were more.
  constexpr uint32_t EMPTY = 0xffffffffu;
  uint4* cache = cache_all + (size_t)f * N1;
  auto walk = [&](int m, int idx1, uint32_t& k0, uint32_t& k1, uint32_t& k2, uint32_t& npass) {
    k0 = k1 = k2 = EMPTY;
    npass = 0;
    const bool bStereo1 = ur1[idx1] >= 0;
    const double u1 = uv1[2 * idx1], v1 = uv1[2 * idx1 + 1];
    // checkEpipolarDist: the line of kp1 in key-frame 2
    const double ea = u1 * F[0] + v1 * F[3] + F[6];
    const double eb = u1 * F[1] + v1 * F[4] + F[7];
    const double ec = u1 * F[2] + v1 * F[5] + F[8];
    const float den = (float)(ea * ea + eb * eb);
    uint32_t d1[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) d1[w] = desc1[(size_t)idx1 * 8 + w];
    const int lo = q_lo[m], hi = q_hi[m];
    for (int b0 = lo; b0 < hi; b0 += 4) {
      int id[4];
      uint4 da[4], db[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int idx2 = b0 + j < hi ? nidx2[b0 + j] : -1;
        if (idx2 >= N2) idx2 = -1;
        if (idx2 >= 0 && owner[idx2] < m) idx2 = -1;  // not available at all (-1), or taken by an earlier feature of key-frame 1
        id[j] = idx2;
        da[j] = db[j] = make_uint4(0, 0, 0, 0);
        if (idx2 >= 0) {
          const uint4* src = (const uint4*)(desc2 + (size_t)idx2 * 8);
          da[j] = src[0];
          db[j] = src[1];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx2 = id[j];
        if (idx2 < 0) continue;
        const int dist = __popc(d1[0] ^ da[j].x) + __popc(d1[1] ^ da[j].y) + __popc(d1[2] ^ da[j].z) + __popc(d1[3] ^ da[j].w) +
                         __popc(d1[4] ^ db[j].x) + __popc(d1[5] ^ db[j].y) + __popc(d1[6] ^ db[j].z) + __popc(d1[7] ^ db[j].w);
        if (dist > 50) continue;  // TH_LOW
        const double u2 = uv2[2 * idx2], v2 = uv2[2 * idx2 + 1];
        const int oc2 = oct2[idx2] & 7;
        if (!bStereo1 && !(ur2[idx2] >= 0)) {
          const float distex = (float)((double)ex - u2);
          const float distey = (float)((double)ey - v2);
          if (distex * distex + distey * distey < 100.0f * P.sf[oc2]) continue;
        }
        const float num = (float)(ea * u2 + eb * v2 + ec);
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (!((double)dsqr < 3.84 * (double)P.sigma2[oc2])) continue;
        const int ord = b0 + j - lo;
        if (ord > 1023) {  // (a node with more than 1 024 partners: this query walks in every round)
          npass = EMPTY;
          continue;
        }
        uint32_t kx = ((uint32_t)dist << 26) | ((uint32_t)(1023 - ord) << 16) | (uint32_t)idx2;
        if (npass != EMPTY) ++npass;
        if (kx < k2) {
          k2 = kx;
          if (k2 < k1) {
            const uint32_t t = k1;
            k1 = k2;
            k2 = t;
          }
          if (k1 < k0) {
            const uint32_t t = k0;
            k0 = k1;
            k1 = t;
          }
        }
      }
    }
  };
  // (a node of more than 1 024 partners cannot be keyed: such a query is evaluated the sequential way, in every round)
  auto walk_seq = [&](int m, int idx1) -> int {
    const bool bStereo1 = ur1[idx1] >= 0;
    const double u1 = uv1[2 * idx1], v1 = uv1[2 * idx1 + 1];
    const double ea = u1 * F[0] + v1 * F[3] + F[6];
    const double eb = u1 * F[1] + v1 * F[4] + F[7];
    const double ec = u1 * F[2] + v1 * F[5] + F[8];
    const float den = (float)(ea * ea + eb * eb);
    uint32_t d1[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) d1[w] = desc1[(size_t)idx1 * 8 + w];
    int bestDist = 50, bestIdx2 = -1;  // TH_LOW
    for (int b = q_lo[m]; b < q_hi[m]; ++b) {
      const int idx2 = nidx2[b];
      if (idx2 < 0 || idx2 >= N2) continue;
      if (owner[idx2] < m) continue;
      const int dist = hamming256(d1, desc2 + (size_t)idx2 * 8);
      if (dist > 50 || dist > bestDist) continue;
      const double u2 = uv2[2 * idx2], v2 = uv2[2 * idx2 + 1];
      const int oc2 = oct2[idx2] & 7;
      if (!bStereo1 && !(ur2[idx2] >= 0)) {
        const float distex = (float)((double)ex - u2);
        const float distey = (float)((double)ey - v2);
        if (distex * distex + distey * distey < 100.0f * P.sf[oc2]) continue;
      }
      const float num = (float)(ea * u2 + eb * v2 + ec);
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)P.sigma2[oc2])) continue;
      bestIdx2 = idx2;
      bestDist = dist;
    }
    return bestIdx2;
  };
  int rounds = 0;
  for (;;) {
    for (int i = tid; i < N2; i += T_T) owner_n[i] = owner[i] < 0 ? -1 : INT_MAX;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int m = tid; m < nq; m += T_T) {
      const int idx1 = q_idx1[m];
      int bestIdx2 = -1;
      if (idx1 >= 0) {
        uint4 rec = make_uint4(EMPTY, EMPTY, EMPTY, 0);
        bool need_walk = rounds == 0;
        if (rounds > 0) {
          rec = cache[m];
          if (rec.w == EMPTY) {
            need_walk = true;
          } else {
            const uint32_t ks[3] = {rec.x, rec.y, rec.z};
            bool found = false;
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (!found && ks[j] != EMPTY && owner[ks[j] & 0xffffu] >= m) {
                bestIdx2 = (int)(ks[j] & 0xffffu);
                found = true;
              }
            if (!found && rec.w > 3u) need_walk = true;  // all three gone and there were more
          }
        }
        if (need_walk) {
          uint32_t k0, k1, k2, npass;
          walk(m, idx1, k0, k1, k2, npass);
          if (npass == EMPTY) {
            bestIdx2 = walk_seq(m, idx1);
            if (rounds == 0) cache[m] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
          } else {
            bestIdx2 = k0 == EMPTY ? -1 : (int)(k0 & 0xffffu);
            if (rounds == 0) cache[m] = make_uint4(k0, k1, k2, npass);
          }
        }
      }
      choice[m] = bestIdx2;
      if (bestIdx2 >= 0) atomicMin(&owner_n[bestIdx2], m);
    }
    __syncthreads();
    int ch = 0;
    for (int i = tid; i < N2; i += T_T) {
      const int o = owner_n[i];
      if (o != owner[i]) ch = 1;
      owner[i] = o;
    }
    if (ch) s_changed = 1;
    __syncthreads();
    ++rounds;
    if (!s_changed || rounds > nq + 1) break;
    __syncthreads();
  }

  // ---- rotation consistency (:235-246, :264-281, computeThreeMaxima :544-578) ----------------------------------------
  // (a query keeps its choice iff it owns it: in the fixed point every choice is owned by its query)
  if (P.check_orientation) {
    const float* angle1 = angle1_all + (size_t)f * N1;
    const float* angle2 = angle2_all + (size_t)f * N2;
    const float factor = 30 / 360.0f;
    auto bin_of = [&](int m) -> int {
      float rot = angle1[q_idx1[m]] - angle2[choice[m]];
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == 30) bin = 0;
      return bin;
    };
    if (tid < 32) s_hist[tid] = 0;
    __syncthreads();
    for (int m = tid; m < nq; m += T_T)
      if (choice[m] >= 0) {
        const int b = bin_of(m);
        if (b >= 0 && b < 30) atomicAdd(&s_hist[b], 1);
      }
    __syncthreads();
    if (tid == 0) {
      int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
      for (int i = 0; i < 30; i++) {
        const int sz = s_hist[i];
        if (sz > max1) {
          max3 = max2;
          max2 = max1;
          max1 = sz;
          ind3 = ind2;
          ind2 = ind1;
          ind1 = i;
        } else if (sz > max2) {
          max3 = max2;
          max2 = sz;
          ind3 = ind2;
          ind2 = i;
        } else if (sz > max3) {
          max3 = sz;
          ind3 = i;
        }
      }
      if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
      } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
      }
      s_keep[0] = ind1;
      s_keep[1] = ind2;
      s_keep[2] = ind3;
    }
    __syncthreads();
    for (int m = tid; m < nq; m += T_T)
      if (choice[m] >= 0) {
        const int b = bin_of(m);
        if (b >= 0 && b < 30 && b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) choice[m] = -1;
      }
    __syncthreads();
  }

  // ---- outputs: matches12 by feature of key-frame 1 ----------------------------------------------------------------
  int32_t* match = match_all + (size_t)f * N1;
  for (int i = tid; i < N1; i += T_T) match[i] = -1;
  __syncthreads();
  int cnt = 0;
  for (int m = tid; m < nq; m += T_T)
    if (q_idx1[m] >= 0 && choice[m] >= 0) {
      match[q_idx1[m]] = choice[m];
      ++cnt;
    }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < T_T / 64; ++w) tot += s_cnt[w];
    nmatches_all[f] = tot;
    if (counters) {  // GL_COUNTER_MATCH_ROUNDS / _UNITS
      atomicAdd(&counters[1], rounds);
      atomicAdd(&counters[2], 1);
    }
  }
}

// ---- matches -> the per-match arrays of gl_create_map_points, on the device ----------------------------------------------
// Localization::createMapPoints walks matched_pairs (ascending feature index of key-frame 1) and reads, per match, the two
// key-frames' poses, key-points (u, v, u_right), depths, octaves and candidate components (kf->comps_[idx], localization_opt.cpp:
// 286-420).  k_tri_offsets: exclusive scan of the pairs' match counts; k_tri_gather: one workgroup per pair compacts its matches
// in index order behind the pair's offset.
__global__ __launch_bounds__(256) void k_tri_offsets(int B, const int32_t* __restrict__ nmatches, int32_t* __restrict__ pair_off) {
  __shared__ int s_part[256];
  const int tid = threadIdx.x, per = (B + 255) / 256, b0 = tid * per, b1 = min(B, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += max(nmatches[b], 0);
  s_part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = s_part[t];
      s_part[t] = run;
      run += v;
    }
    pair_off[B] = run;
  }
  __syncthreads();
  int run = s_part[tid];
  for (int b = b0; b < b1; ++b) {
    pair_off[b] = run;
    run += max(nmatches[b], 0);
  }
}

struct TriKf {  // one key-frame side of the gather (strides N, k candidates)
  const double* pose;    // B x 7
  const double* uv;      // B x N x 2
  const float* ur;       // B x N
  const float* depth;    // B x N
  const int32_t* oct;    // B x N
  const int32_t* cand;   // B x N x k
  const int32_t* ncand;  // B x N
};
struct TriOut {
  double* pose;
  double* uvr;
  float* depth;
  int32_t* oct;
  int32_t* cand;
  int32_t* ncand;
};

__global__ __launch_bounds__(256) void k_tri_gather(int B, int N1, int N2, int k, int cap, const int32_t* __restrict__ match_all,
                                                    const int32_t* __restrict__ pair_off, TriKf a, TriKf b, TriOut oa, TriOut ob,
                                                    int32_t* __restrict__ pair_of, int32_t* __restrict__ idx1_of, int32_t* __restrict__ idx2_of) {
  __shared__ int s_w[4], s_base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (f >= B) return;
  const int32_t* match = match_all + (size_t)f * N1;
  if (tid == 0) s_base = pair_off[f];
  __syncthreads();
  for (int i0 = 0; i0 < N1; i0 += 256) {
    const int i = i0 + tid;
    const int j = i < N1 ? match[i] : -1;
    const bool on = j >= 0 && j < N2;
    const unsigned long long bal = __ballot(on);
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += s_w[w];
    const int at = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (on && at < cap) {
      const size_t g1 = (size_t)f * N1 + i, g2 = (size_t)f * N2 + j;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        oa.pose[(size_t)at * 7 + r] = a.pose[(size_t)f * 7 + r];
        ob.pose[(size_t)at * 7 + r] = b.pose[(size_t)f * 7 + r];
      }
      oa.uvr[(size_t)at * 3] = a.uv[g1 * 2];
      oa.uvr[(size_t)at * 3 + 1] = a.uv[g1 * 2 + 1];
      oa.uvr[(size_t)at * 3 + 2] = (double)a.ur[g1];
      ob.uvr[(size_t)at * 3] = b.uv[g2 * 2];
      ob.uvr[(size_t)at * 3 + 1] = b.uv[g2 * 2 + 1];
      ob.uvr[(size_t)at * 3 + 2] = (double)b.ur[g2];
      oa.depth[at] = a.depth[g1];
      ob.depth[at] = b.depth[g2];
      oa.oct[at] = a.oct[g1];
      ob.oct[at] = b.oct[g2];
      oa.ncand[at] = a.ncand[g1];
      ob.ncand[at] = b.ncand[g2];
      for (int c = 0; c < k; ++c) {
        oa.cand[(size_t)at * k + c] = a.cand[g1 * k + c];
        ob.cand[(size_t)at * k + c] = b.cand[g2 * k + c];
      }
      pair_of[at] = f;
      idx1_of[at] = i;
      idx2_of[at] = j;
    }
    __syncthreads();
    if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
}

}  // namespace

extern "C" int gl_gather_triangulation_matches(gl_ctx_t* ctx, int B, int N1, int N2, int k, int cap, const int32_t* match12_dev,
                                               const int32_t* nmatches_dev, const double* pose1_dev, const double* uv1_dev,
                                               const float* ur1_dev, const float* depth1_dev, const int32_t* oct1_dev, const int32_t* cand1_dev,
                                               const int32_t* ncand1_dev, const double* pose2_dev, const double* uv2_dev, const float* ur2_dev,
                                               const float* depth2_dev, const int32_t* oct2_dev, const int32_t* cand2_dev,
                                               const int32_t* ncand2_dev, int32_t* pair_off_dev, double* m_pose1_dev, double* m_uvr1_dev,
                                               float* m_depth1_dev, int32_t* m_oct1_dev, int32_t* m_cand1_dev, int32_t* m_n1_dev,
                                               double* m_pose2_dev, double* m_uvr2_dev, float* m_depth2_dev, int32_t* m_oct2_dev,
                                               int32_t* m_cand2_dev, int32_t* m_n2_dev, int32_t* m_pair_dev, int32_t* m_idx1_dev,
                                               int32_t* m_idx2_dev) {
  GL_REQUIRE(ctx, "null context");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && N1 >= 1 && N2 >= 1 && k >= 1 && cap >= 1, "bad B / N1 / N2 / k / cap");
  GL_REQUIRE(match12_dev && nmatches_dev && pose1_dev && uv1_dev && ur1_dev && depth1_dev && oct1_dev && cand1_dev && ncand1_dev && pose2_dev &&
                 uv2_dev && ur2_dev && depth2_dev && oct2_dev && cand2_dev && ncand2_dev && pair_off_dev && m_pose1_dev && m_uvr1_dev &&
                 m_depth1_dev && m_oct1_dev && m_cand1_dev && m_n1_dev && m_pose2_dev && m_uvr2_dev && m_depth2_dev && m_oct2_dev &&
                 m_cand2_dev && m_n2_dev && m_pair_dev && m_idx1_dev && m_idx2_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  k_tri_offsets<<<1, 256, 0, c->stream>>>(B, nmatches_dev, pair_off_dev);
  k_tri_gather<<<B, 256, 0, c->stream>>>(B, N1, N2, k, cap, match12_dev, pair_off_dev,
                                         TriKf{pose1_dev, uv1_dev, ur1_dev, depth1_dev, oct1_dev, cand1_dev, ncand1_dev},
                                         TriKf{pose2_dev, uv2_dev, ur2_dev, depth2_dev, oct2_dev, cand2_dev, ncand2_dev},
                                         TriOut{m_pose1_dev, m_uvr1_dev, m_depth1_dev, m_oct1_dev, m_cand1_dev, m_n1_dev},
                                         TriOut{m_pose2_dev, m_uvr2_dev, m_depth2_dev, m_oct2_dev, m_cand2_dev, m_n2_dev}, m_pair_dev, m_idx1_dev,
                                         m_idx2_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

extern "C" int gl_search_for_triangulation(gl_ctx_t* ctx, float scale_factor, int B, int N1, int N2, int NN1, int NN2,
                                           const double* uv1_dev, const float* ur1_dev, const int32_t* oct1_dev, const float* angle1_dev,
                                           const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev,
                                           const int32_t* node_id1_dev, const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev,
                                           const double* uv2_dev, const float* ur2_dev, const int32_t* oct2_dev, const float* angle2_dev,
                                           const uint8_t* desc2_dev, const uint8_t* has_mp2_dev, const int32_t* nnode2_dev,
                                           const int32_t* node_id2_dev, const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev,
                                           const double* fmat_dev, const float* epipole_dev, int only_stereo, int check_orientation,
                                           int32_t* match12_dev, int32_t* nmatches_dev) {
  GL_REQUIRE(ctx, "null context");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && N1 >= 1 && N2 >= 1 && NN1 >= 1 && NN2 >= 1, "bad B / N1 / N2 / NN1 / NN2");
  GL_REQUIRE(N1 <= 4096 && N2 <= 4096, "N1 / N2 above the on-chip capacity (4096 features per key-frame)");
  GL_REQUIRE(uv1_dev && ur1_dev && oct1_dev && angle1_dev && desc1_dev && has_mp1_dev && nnode1_dev && node_id1_dev && node_ptr1_dev &&
                 node_idx1_dev && uv2_dev && ur2_dev && oct2_dev && angle2_dev && desc2_dev && has_mp2_dev && nnode2_dev && node_id2_dev &&
                 node_ptr2_dev && node_idx2_dev && fmat_dev && epipole_dev && match12_dev && nmatches_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  TriP P;
  P.N1 = N1;
  P.N2 = N2;
  P.NN1 = NN1;
  P.NN2 = NN2;
  P.only_stereo = only_stereo;
  P.check_orientation = check_orientation;
  P.sf[0] = 1.0f;  // init_config.hpp:63-79
  P.sigma2[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    P.sf[i] = P.sf[i - 1] * scale_factor;
    P.sigma2[i] = P.sf[i] * P.sf[i];
  }
  const size_t lds = ((size_t)2 * N2 + 4 * (size_t)N1 + (size_t)NN1 + 2 * (size_t)NN2 + 2) * sizeof(int32_t);
  GL_REQUIRE_LDS(c, lds);
  GL_HIP(gl::ensure_dynamic_lds(c, (const void*)k_search_for_triangulation, lds));
  void* cache = nullptr;  // 16 bytes per query: its three best partners of round 1 (the kernel's round loop)
  {
    const int rc = gl::ctx_scratch_b(c, (size_t)B * N1 * sizeof(uint4), &cache);
    if (rc != GL_OK) return rc;
  }
  k_search_for_triangulation<<<B, T_T, lds, c->stream>>>(P, B, uv1_dev, ur1_dev, oct1_dev, angle1_dev, desc1_dev, has_mp1_dev, nnode1_dev,
                                                        node_id1_dev, node_ptr1_dev, node_idx1_dev, uv2_dev, ur2_dev, oct2_dev, angle2_dev,
                                                        desc2_dev, has_mp2_dev, nnode2_dev, node_id2_dev, node_ptr2_dev, node_idx2_dev, fmat_dev,
                                                        epipole_dev, match12_dev, nmatches_dev, c->counters, (uint4*)cache);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
