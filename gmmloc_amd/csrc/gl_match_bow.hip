// ORBmatcher::searchByBoW (orb_matcher.cpp:295-408) with computeThreeMaxima (:544-578) for B key-frame / frame pairs: the matcher
// of Tracking::trackReferenceKeyFrame (tracking.cpp:303) - the map points of the reference key-frame are handed to the features
// of the current frame that share their vocabulary node and pass the distance / ratio tests.  Integer / byte work: per shared
// node, 256-bit Hamming distances between the key-frame's features that hold a valid map point and the frame's features.
//
// The reference loop is ORDER DEPENDENT like the other matchers' (gl_match.hip, gl_match_tri.hip): a frame feature taken by an
// earlier key-frame feature - earlier in the node's list; nodes do not interact, a feature sits in one node - is skipped by
// every later one, which then sees other best / second-best distances.  Same fixed point: queries are numbered by their position
// in the key-frame's feature-vector list (the reference's visiting order); in every round each query scans its node's frame
// features that are not owned by a LOWER query in the previous round - best = FIRST of the minimum distance (`dist < bestDist1`),
// second best = the minimum of the rest (a later equal distance becomes second best, and the ratio test then fails) -, decides
// (bestDist1 <= TH_LOW and bestDist1 < nn_ratio * bestDist2, in float), and an accepting query claims its feature:
// owner[feature] = the lowest query that claimed it.  The decisions of the queries 0 .. r-1 of a node are final after round r;
// the iteration stops when the owner table repeats.  Then the rotation histogram.
#include <climits>

#include "gl_internal.hpp"

namespace {

constexpr int T_B = 512;

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

__global__ __launch_bounds__(T_B) void k_search_by_bow(int B, int N1, int N2, int NN1, int NN2, float nn_ratio, int check_orientation,
                                                       const float* __restrict__ angle1_all, const uint8_t* __restrict__ desc1_all,
                                                       const uint8_t* __restrict__ mp1_all, const int32_t* __restrict__ nn1_all,
                                                       const int32_t* __restrict__ nid1_all, const int32_t* __restrict__ nptr1_all,
                                                       const int32_t* __restrict__ nidx1_all, const float* __restrict__ angle2_all,
                                                       const uint8_t* __restrict__ desc2_all, const int32_t* __restrict__ nn2_all,
                                                       const int32_t* __restrict__ nid2_all, const int32_t* __restrict__ nptr2_all,
                                                       const int32_t* __restrict__ nidx2_all, int32_t* __restrict__ match_all,
                                                       int32_t* __restrict__ nmatches_all, int32_t* __restrict__ counters, uint4* __restrict__ cache_all,
                                                       const int32_t* __restrict__ run_flag) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  int32_t* owner = lds;            // N2: lowest query that claimed the frame feature in the previous round (INT_MAX: nobody)
  int32_t* owner_n = owner + N2;   // N2: being rebuilt
  int32_t* choice = owner_n + N2;  // N1 (by query): the frame feature an ACCEPTING query takes, else -1
  int32_t* q_idx1 = choice + N1;   // N1: the query's key-frame feature, or -1 (no valid map point, node not shared)
  int32_t* q_lo = q_idx1 + N1;     // N1: its candidates = node_idx2[q_lo .. q_hi)
  int32_t* q_hi = q_lo + N1;
  // the three tables the queries are set up from (two binary searches per query: sixteen dependent GLOBAL loads each before round 5)
  int32_t* s_nptr1 = q_hi + N1;            // NN1 + 1
  int32_t* s_nid2 = s_nptr1 + NN1 + 1;   // NN2
  int32_t* s_nptr2 = s_nid2 + NN2;       // NN2 + 1
  __shared__ int s_changed, s_hist[32], s_keep[4], s_cnt[T_B / 64];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= B) return;
  if (run_flag && run_flag[f] == 0) return;  // (gl_track_frame_chain's trackKeyFrame fallback: only the frames that need it)
  const uint32_t* desc1 = (const uint32_t*)(desc1_all + (size_t)f * N1 * 32);
  const uint8_t* mp1 = mp1_all + (size_t)f * N1;
  const uint32_t* desc2 = (const uint32_t*)(desc2_all + (size_t)f * N2 * 32);
  const int nn1 = min(nn1_all[f], NN1), nn2 = min(nn2_all[f], NN2);
  const int32_t* nid1 = nid1_all + (size_t)f * NN1;
  const int32_t* nptr1 = nptr1_all + (size_t)f * (NN1 + 1);
  const int32_t* nidx1 = nidx1_all + (size_t)f * N1;
  const int32_t* nid2 = nid2_all + (size_t)f * NN2;
  const int32_t* nptr2 = nptr2_all + (size_t)f * (NN2 + 1);
  const int32_t* nidx2 = nidx2_all + (size_t)f * N2;
  const int nq = nn1 > 0 ? min(nptr1[nn1], N1) : 0;  // list entries of the key-frame = queries, in the reference's visiting order

  for (int i = tid; i <= nn1; i += T_B) s_nptr1[i] = nptr1[i];
  for (int i = tid; i <= nn2; i += T_B) {
    s_nptr2[i] = nptr2[i];
    if (i < nn2) s_nid2[i] = nid2[i];
  }
  __syncthreads();
  for (int a = tid; a < N1; a += T_B) {
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
        if (i1 >= 0 && i1 < N1 && mp1[i1]) {
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
  for (int i = tid; i < N2; i += T_B) owner[i] = INT_MAX;
  __syncthreads();

  // Round 5 (as in gl_match.hip / gl_match_tri.hip): a query's partners and their distances do not depend on the owners, which only
  // REMOVE partners.  best = first of the smallest distance, second best = the smallest of the rest: the two smallest keys
  //     dist << 22 | position in the partner list << 12 | feature of the frame.
  // Round 1 evaluates every partner once (descriptors requested four at a time) and leaves the three smallest keys and their number
  // in a 16-byte record; a later round decides from the keys whose features no lower query owns whenever two are left or the
  // record held every partner, and walks again otherwise.
  constexpr uint32_t EMPTY = 0xffffffffu;
  uint4* cache = cache_all + (size_t)f * N1;
  auto decide = [&](uint32_t a, uint32_t b) -> int {
    if (a == EMPTY) return -1;
    const int bestDist1 = (int)(a >> 22), bestDist2 = b == EMPTY ? 256 : (int)(b >> 22);
    return (bestDist1 <= 50 && (float)bestDist1 < nn_ratio * (float)bestDist2) ? (int)(a & 0xfffu) : -1;  // TH_LOW, nn_ratio_
  };
  auto walk = [&](int m, int idx1, uint32_t& k0, uint32_t& k1, uint32_t& k2, uint32_t& npass) {
    k0 = k1 = k2 = EMPTY;
    npass = 0;
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
        if (idx2 >= 0 && owner[idx2] < m) idx2 = -1;  // matches[realIdxF] is set: taken by an earlier key-frame feature
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
        if (dist >= 256) continue;  // (never below the initial best / second best)
        const int ord = b0 + j - lo;
        if (ord > 1023) {  // (a node of more than 1 024 partners cannot be keyed: the sequential evaluation, in every round)
          npass = EMPTY;
          continue;
        }
        const uint32_t kx = ((uint32_t)dist << 22) | ((uint32_t)ord << 12) | (uint32_t)idx2;
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
  auto walk_seq = [&](int m, int idx1) -> int {
    uint32_t d1[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) d1[w] = desc1[(size_t)idx1 * 8 + w];
    int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
    for (int b = q_lo[m]; b < q_hi[m]; ++b) {
      const int idx2 = nidx2[b];
      if (idx2 < 0 || idx2 >= N2) continue;
      if (owner[idx2] < m) continue;
      const int dist = hamming256(d1, desc2 + (size_t)idx2 * 8);
      if (dist < bestDist1) {
        bestDist2 = bestDist1;
        bestDist1 = dist;
        bestIdxF = idx2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    return (bestDist1 <= 50 && (float)bestDist1 < nn_ratio * (float)bestDist2) ? bestIdxF : -1;
  };
  int rounds = 0;
  for (;;) {
    for (int i = tid; i < N2; i += T_B) owner_n[i] = INT_MAX;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int m = tid; m < nq; m += T_B) {
      const int idx1 = q_idx1[m];
      int take = -1;
      if (idx1 >= 0) {
        bool need_walk = rounds == 0;
        if (rounds > 0) {
          const uint4 rec = cache[m];
          if (rec.w == EMPTY) {
            need_walk = true;
          } else {
            uint32_t a = EMPTY, b2 = EMPTY;
            int nav = 0;
            const uint32_t ks[3] = {rec.x, rec.y, rec.z};
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (ks[j] != EMPTY && owner[ks[j] & 0xfffu] >= m) {
                if (nav == 0) a = ks[j];
                else if (nav == 1) b2 = ks[j];
                ++nav;
              }
            if (nav >= 2 || rec.w <= 3u) take = decide(a, b2);
            else need_walk = true;
          }
        }
        if (need_walk) {
          uint32_t k0, k1, k2, npass;
          walk(m, idx1, k0, k1, k2, npass);
          if (npass == EMPTY) {
            take = walk_seq(m, idx1);
            if (rounds == 0) cache[m] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
          } else {
            take = decide(k0, k1);
            if (rounds == 0) cache[m] = make_uint4(k0, k1, k2, npass);
          }
        }
      }
      choice[m] = take;
      if (take >= 0) atomicMin(&owner_n[take], m);
    }
    __syncthreads();
    int ch = 0;
    for (int i = tid; i < N2; i += T_B) {
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

  // ---- rotation consistency (:362-374, :391-405, computeThreeMaxima :544-578) ------------------------------------------
  // (in the fixed point every accepting query owns its choice)
  if (check_orientation) {
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
    for (int m = tid; m < nq; m += T_B)
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
    for (int m = tid; m < nq; m += T_B)
      if (choice[m] >= 0) {
        const int b = bin_of(m);
        if (b >= 0 && b < 30 && b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) choice[m] = -1;
      }
    __syncthreads();
  }

  // ---- outputs: matches by FRAME feature (the key-frame feature whose map point it gets) ------------------------------------
  int32_t* match = match_all + (size_t)f * N2;
  for (int i = tid; i < N2; i += T_B) match[i] = -1;
  __syncthreads();
  int cnt = 0;
  for (int m = tid; m < nq; m += T_B)
    if (q_idx1[m] >= 0 && choice[m] >= 0) {
      match[choice[m]] = q_idx1[m];
      ++cnt;
    }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < T_B / 64; ++w) tot += s_cnt[w];
    nmatches_all[f] = tot;
    if (counters) {  // GL_COUNTER_MATCH_ROUNDS / _UNITS
      atomicAdd(&counters[1], rounds);
      atomicAdd(&counters[2], 1);
    }
  }
}

}  // namespace

extern "C" int gl_search_by_bow(gl_ctx_t* ctx, float nn_ratio, int check_orientation, int B, int N1, int N2, int NN1, int NN2,
                                const float* angle1_dev, const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev,
                                const int32_t* node_id1_dev, const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev,
                                const float* angle2_dev, const uint8_t* desc2_dev, const int32_t* nnode2_dev, const int32_t* node_id2_dev,
                                const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev, int32_t* match21_dev, int32_t* nmatches_dev) {
  return gl::launch_bow_gated(ctx, nn_ratio, check_orientation, B, N1, N2, NN1, NN2, angle1_dev, desc1_dev, has_mp1_dev, nnode1_dev, node_id1_dev,
                              node_ptr1_dev, node_idx1_dev, angle2_dev, desc2_dev, nnode2_dev, node_id2_dev, node_ptr2_dev, node_idx2_dev, match21_dev,
                              nmatches_dev, nullptr);
}

// the same with a per-frame switch (run_flag B int32, 0: the frame's workgroup returns at once and writes nothing; null: every frame)
int gl::launch_bow_gated(gl_ctx_t* ctx, float nn_ratio, int check_orientation, int B, int N1, int N2, int NN1, int NN2, const float* angle1_dev,
                         const uint8_t* desc1_dev, const uint8_t* has_mp1_dev, const int32_t* nnode1_dev, const int32_t* node_id1_dev,
                         const int32_t* node_ptr1_dev, const int32_t* node_idx1_dev, const float* angle2_dev, const uint8_t* desc2_dev,
                         const int32_t* nnode2_dev, const int32_t* node_id2_dev, const int32_t* node_ptr2_dev, const int32_t* node_idx2_dev,
                         int32_t* match21_dev, int32_t* nmatches_dev, const int32_t* run_flag) {
  GL_REQUIRE(ctx, "null context");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && N1 >= 1 && N2 >= 1 && NN1 >= 1 && NN2 >= 1, "bad B / N1 / N2 / NN1 / NN2");
  GL_REQUIRE(N1 <= 4096 && N2 <= 4096, "N1 / N2 above the on-chip capacity (4096 features per frame)");
  GL_REQUIRE(angle1_dev && desc1_dev && has_mp1_dev && nnode1_dev && node_id1_dev && node_ptr1_dev && node_idx1_dev && angle2_dev &&
                 desc2_dev && nnode2_dev && node_id2_dev && node_ptr2_dev && node_idx2_dev && match21_dev && nmatches_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  const size_t lds = ((size_t)2 * N2 + 4 * (size_t)N1 + (size_t)NN1 + 2 * (size_t)NN2 + 2) * sizeof(int32_t);
  GL_REQUIRE_LDS(c, lds);
  GL_HIP(gl::ensure_dynamic_lds(c, (const void*)k_search_by_bow, lds));
  void* cache = nullptr;  // 16 bytes per query: its three best partners of round 1
  {
    const int rc = gl::ctx_scratch_b(c, (size_t)B * N1 * sizeof(uint4), &cache);
    if (rc != GL_OK) return rc;
  }
  k_search_by_bow<<<B, T_B, lds, c->stream>>>(B, N1, N2, NN1, NN2, nn_ratio, check_orientation, angle1_dev, desc1_dev, has_mp1_dev, nnode1_dev,
                                              node_id1_dev, node_ptr1_dev, node_idx1_dev, angle2_dev, desc2_dev, nnode2_dev, node_id2_dev,
                                              node_ptr2_dev, node_idx2_dev, match21_dev, nmatches_dev, c->counters, (uint4*)cache, run_flag);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
