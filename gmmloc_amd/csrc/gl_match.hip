// ORBmatcher::searchByProjection(Frame&, mappts, stats, th) (orb_matcher.cpp:27-110) for B frames:
// the step that produces the 3-D / 2-D correspondences Tracking::optimizeCurrentPose consumes
// (SURVEY.md 8f rank 2).  Integer / byte work: a 64 x 48 bucket grid of the frame's features
// (Frame::assignFeaturesToGrid, frame.cpp:54-79), a window walk per projected map point
// (Frame::getFeaturesInArea, frame.cpp:121-177), 256-bit Hamming distances
// (ORBmatcher::DescriptorDistance, orb_matcher.cpp:580-596), best / second-best with the
// level-aware ratio test.
//
// The reference loop is ORDER DEPENDENT: a feature taken by map point m is skipped by every later
// map point, which then falls back to its next-best candidate.  The kernel reproduces that exactly
// with a fixed-point iteration: in every round each map point m picks its best feature among those
// not owned by a map point < m (owners of the previous round), then owner[f] = min{m : choice(m) = f}
// is rebuilt with LDS atomics; by induction the choices of map points 0..r-1 are final after round
// r, and the iteration stops when the owner table repeats (2-5 rounds on realistic frames).
//
// Round 5: a round no longer repeats the window walk.  What a query can ever choose from is fixed by the filters that do not
// depend on the owners (level, window, u_right, taken on entry); the owners only REMOVE candidates.  So round 1 walks the window
// once - the CSR entries of all covered columns as ONE flattened loop, the surviving candidates collected four at a time and their
// descriptors requested together - and leaves each query's three best candidates as sorted keys
//     dist << 24 | position among its candidates << 16 | octave << 12 | feature
// (64-bit keys with wider fields while walking; a query whose candidates do not fit the packed form just walks again)
// (the reference's best / second-best bookkeeping is "first of the smallest distance, then first of the smallest of the rest" in
// visiting order = the two smallest keys) plus the number of candidates in a 16-byte record; every later round reads the record,
// drops the keys whose feature a lower query owns, and decides - exactly - whenever two keys survive or the record held all the
// candidates; only a query that lost two of its three best AND had more walks again.
//
// One workgroup per frame (512 threads in batches; 1 024 and the descriptors in LDS as well when there are no more
// frames than CUs); grid CSR, owner tables and choices live in LDS.  Every float / double
// conversion of the reference (float window, float grid scale, double feature coordinates) is kept
// so that the candidate sets, their visiting order (cell column, cell row, feature index) and hence
// the ties are identical.
#include <climits>

#include "gl_internal.hpp"

namespace {

constexpr int GC = 64, GR = 48, NCELL = GC * GR;  // frame::grid_cols / grid_rows (config.h:57)
constexpr int T_M = 512;

struct MatchP {
  int NF, NP;
  float col_inv, row_inv, th, nn_ratio;
  float sf[8];
  // frame-to-frame mode (orb_matcher.cpp:410-542)
  double fx, fy, cx, cy;
  float mbf, mb;
  int width, height, mono, check_orientation;
};

// 32-bit words of the kernel's LDS in front of the per-entry records: cell_ptr, cursor, cell_idx, owner, owner_n, qorder (u16)
// (round 6: the per-query `choice` table - written every round, read by nobody - is gone: 4 bytes per map point that kept a frame with
//  3 000 local map points from sharing its CU with a second one - 85.8 KB -> 73.8 KB)
__host__ __device__ inline int lds_ints(int NF, int NP) { return (2 * NCELL + 1 + 3 * NF + (NP + 1) / 2 + 3) & ~3; }

// what one query point (a projected map point) asks of the feature grid
struct Query {
  bool valid;
  float x, y, rr;          // window centre / half size (float, as the reference passes them)
  int minLevel, maxLevel;  // getFeaturesInArea level filter
  bool ratio_test;         // best / second-best test of the local-map overload
  double ur_d;             // predicted u_right: compared in double (overload 1) ...
  float ur_f;              // ... or in float (overload 2)
  bool ur_float;
};

// Eigen Quaternion * Vector3:  uv = 2 q.vec x v;  v + w uv + q.vec x uv   (g2o SE3Quat::map)
__device__ __forceinline__ void quat_rot(const double* q, const double* v, double* o) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  o[0] = v[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]);
  o[1] = v[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]);
  o[2] = v[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]);
}

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) d += __popc(a[w] ^ b[w]);
  return d;
}

// MODE 0: searchByProjection(Frame&, mappts, stats, th): query points = projected local map points
//         (mp_uvr = ProjStat::uvr, mp_level = scale_pred, mp_viewcos).
// MODE 1: searchByProjection(CurrentFrame, LastFrame, th, bMono): query points = the last frame's map
//         points (mp_uvr = world position, mp_level = octave of the last frame's feature), projected here
//         with the current pose; rotation-consistency histogram at the end (feat_angle / mp_angle).
template <int MODE, bool DL>
__global__ __launch_bounds__(DL ? 1024 : T_M, 4) void k_search_by_projection(  // (four waves per SIMD: two frames per CU in batches)
    MatchP P, int B, const double* __restrict__ feat_uv_all, const float* __restrict__ feat_ur_all,
    const int32_t* __restrict__ feat_oct_all, const uint8_t* __restrict__ feat_desc_all,
    const uint8_t* __restrict__ feat_taken_all, const double* __restrict__ mp_uvr_all,
    const int32_t* __restrict__ mp_level_all, const double* __restrict__ mp_viewcos_all,
    const uint8_t* __restrict__ mp_valid_all, const uint8_t* __restrict__ mp_desc_all,
    int32_t* __restrict__ feat_match_all, int32_t* nmatches_all,  // (not restrict: the chain's wider second search gates on the counts it rewrites)
    const double* __restrict__ pose_cw_all, const double* __restrict__ pose_lw_all,
    const float* __restrict__ feat_angle_all, const float* __restrict__ mp_angle_all, int32_t* __restrict__ counters,
    uint4* __restrict__ cache_all, const int32_t* gate_nm, int gate_min) {
  constexpr int TM = DL ? 1024 : T_M;  // threads per frame: the latency shape doubles them
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  int32_t* cell_ptr = lds;                   // NCELL + 1
  int32_t* cursor = cell_ptr + NCELL + 1;    // NCELL (grid build only)
  int32_t* cell_idx = cursor + NCELL;        // NF
  int32_t* owner = cell_idx + P.NF;          // NF   owner of the previous round (-1: taken on entry)
  int32_t* owner_n = owner + P.NF;           // NF   being rebuilt
  // per CSR entry, so that the window walk touches LDS only: {u, v} double, {u_right bits, octave}
  uint16_t* qorder = (uint16_t*)(owner_n + P.NF);  // NP: the queries sorted by window class (round 5: equal work per lane)
  double2* rec_uv = (double2*)(lds + lds_ints(P.NF, P.NP));  // 16-byte aligned
  int2* rec_ro = (int2*)(rec_uv + P.NF);
  // DL (few frames: one workgroup per CU anyway): the 256-bit descriptors too, in CSR order - a candidate that
  // survives the window test otherwise costs a dependent global load of 32 bytes
  uint4* rec_desc = (uint4*)(rec_ro + ((P.NF + 1) & ~1));  // 2 x uint4 per entry, 16-byte aligned
  __shared__ int s_changed, s_scan[32], s_hist[32], s_keep[4];
  __shared__ int s_fast, s_cls[17], s_nrw;
  __shared__ double s_pose[8];
  __shared__ int s_dir;
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= B) return;
  if (gate_nm && gate_nm[f] >= gate_min) return;  // (gl_track_frame_chain's wider second search: only the frames that need it)
#ifdef GL_MATCH_PROF
  const long long tp0 = clock64();
  long long tp_round = 0;
#endif
  const int NF = P.NF, NP = P.NP;
  const double* feat_uv = feat_uv_all + (size_t)f * NF * 2;
  const float* feat_ur = feat_ur_all + (size_t)f * NF;
  const int32_t* feat_oct = feat_oct_all + (size_t)f * NF;
  const uint32_t* feat_desc = (const uint32_t*)(feat_desc_all + (size_t)f * NF * 32);
  const uint8_t* feat_taken = feat_taken_all + (size_t)f * NF;
  const double* mp_uvr = mp_uvr_all + (size_t)f * NP * 3;
  const int32_t* mp_level = mp_level_all + (size_t)f * NP;
  const double* mp_viewcos = MODE == 0 ? mp_viewcos_all + (size_t)f * NP : nullptr;
  const uint8_t* mp_valid = mp_valid_all + (size_t)f * NP;
  const uint32_t* mp_desc = (const uint32_t*)(mp_desc_all + (size_t)f * NP * 32);

  // ---- assignFeaturesToGrid: CSR by cell (ix * GR + iy), ascending feature index inside a cell ----
  for (int c = tid; c <= NCELL; c += TM) cell_ptr[c] = 0;
  if (tid == 0) s_fast = 1;
  if (tid < 17) s_cls[tid] = 0;
  __syncthreads();
  auto cell_of = [&](int i) -> int {
    if (feat_oct[i] < 0) return -1;  // padding slot
    const double px = round((feat_uv[2 * i] - 0.0f) * P.col_inv), py = round((feat_uv[2 * i + 1] - 0.0f) * P.row_inv);
    if (!(px >= 0 && px < GC && py >= 0 && py < GR)) return -1;  // also rejects NaN
    return (int)px * GR + (int)py;
  };
  {
    // the fast walk (32-bit window test, one 16-byte record per entry) is exact iff every feature coordinate is a float value - the
    // reference's (float)(double u - (double)x) is then the correctly rounded float difference, which is what u - x in float is
    // (a double has more than 2 x 24 + 2 bits) - and the octaves fit the level mask
    bool fok = true;
    for (int i = tid; i < NF; i += TM) {
      const int c = cell_of(i);
      if (c >= 0) {
        atomicAdd(&cell_ptr[c + 1], 1);
        const double u = feat_uv[2 * i], v = feat_uv[2 * i + 1];
        fok = fok && (double)(float)u == u && (double)(float)v == v && feat_oct[i] <= 15;
      }
    }
    if (!fok) s_fast = 0;
  }
  __syncthreads();
  {  // exclusive scan of NCELL counts: each thread scans a contiguous chunk, then the chunk sums
    constexpr int CH = (NCELL + TM - 1) / TM;
    const int c0 = tid * CH, c1 = min(NCELL, c0 + CH);
    int s = 0;
    for (int c = c0; c < c1; ++c) s += cell_ptr[c + 1];
    // exclusive scan of the TM chunk sums: shuffle scan inside each wave, then the wave totals
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if ((tid & 63) >= o) inc += up;
    }
    if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
    __syncthreads();
    int run = inc - s;
    for (int w = 0; w < (tid >> 6); ++w) run += s_scan[w];
    __syncthreads();  // s_scan is reused below
    for (int c = c0; c < c1; ++c) {
      const int v = cell_ptr[c + 1];
      cell_ptr[c + 1] = run + v;  // inclusive end; cell_ptr[c] (= end of c-1) is its start
      run += v;
    }
    __syncthreads();
  }
  // fill through a per-cell cursor, then put every (short) cell list in ascending feature order: that
  // is the push_back order of the reference and decides ties between equal Hamming distances
  for (int c = tid; c < NCELL; c += TM) cursor[c] = 0;
  __syncthreads();
  for (int i = tid; i < NF; i += TM) {
    const int c = cell_of(i);
    if (c >= 0) cell_idx[cell_ptr[c] + atomicAdd(&cursor[c], 1)] = i;
  }
  __syncthreads();
  for (int c = tid; c < NCELL; c += TM) {
    const int e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
    for (int e = e0 + 1; e < e1; ++e) {  // insertion sort
      const int v = cell_idx[e];
      int k = e - 1;
      while (k >= e0 && cell_idx[k] > v) {
        cell_idx[k + 1] = cell_idx[k];
        --k;
      }
      cell_idx[k + 1] = v;
    }
  }
  __syncthreads();
  for (int e = tid; e < cell_ptr[NCELL]; e += TM) {
    const int i = cell_idx[e];
    if (s_fast) ((float4*)rec_uv)[e] = make_float4((float)feat_uv[2 * i], (float)feat_uv[2 * i + 1], feat_ur[i], __int_as_float((feat_oct[i] & 0xff) | (i << 8)));
    else rec_uv[e] = make_double2(feat_uv[2 * i], feat_uv[2 * i + 1]);
    rec_ro[e] = make_int2(__float_as_int(feat_ur[i]), (feat_oct[i] & 0xff) | (i << 8));  // {u_right, octave | feature << 8}
    if (DL) {
      const uint4* src = (const uint4*)(feat_desc + (size_t)i * 8);
      rec_desc[2 * e] = src[0];
      rec_desc[2 * e + 1] = src[1];
    }
  }
  __syncthreads();

  // ---- owners on entry -------------------------------------------------------------------------------
  for (int i = tid; i < NF; i += TM) owner[i] = feat_taken[i] ? -1 : INT_MAX;
  __syncthreads();

  if (MODE == 1 && tid == 0) {  // current pose, direction of motion (orb_matcher.cpp:421-428)
    const double* pc = pose_cw_all + (size_t)f * 7;
    const double* pl = pose_lw_all + (size_t)f * 7;
    const double qi[4] = {-pc[0], -pc[1], -pc[2], pc[3]};
    const double mt[3] = {pc[4] * -1., pc[5] * -1., pc[6] * -1.};
    double twc[3], tlc[3];
    quat_rot(qi, mt, twc);  // Twc = Tcw.inverse()
    quat_rot(pl, twc, tlc);
    const double tz = tlc[2] + pl[6];
    s_dir = (tz > P.mb && !P.mono) ? 1 : ((-tz > P.mb && !P.mono) ? 2 : 0);
    for (int i = 0; i < 7; ++i) s_pose[i] = pc[i];
  }
  __syncthreads();

  const bool bFactor = P.th != 1.0;
  auto make_query = [&](int m) -> Query {
    Query q;
    q.valid = mp_valid[m] != 0;
    q.ratio_test = MODE == 0;
    q.ur_float = MODE == 1;
    q.ur_d = 0.0;
    q.ur_f = 0.f;
    q.x = q.y = q.rr = 0.f;
    q.minLevel = q.maxLevel = -1;
    if (!q.valid) return q;
    if (MODE == 0) {
      const int lvl = mp_level[m];
      float r = ((float)mp_viewcos[m] > 0.998) ? 2.5f : 4.0f;
      if (bFactor) r *= P.th;
      q.rr = r * P.sf[lvl & 7];
      q.x = (float)mp_uvr[3 * m];
      q.y = (float)mp_uvr[3 * m + 1];
      q.ur_d = mp_uvr[3 * m + 2];
      q.minLevel = lvl - 1;
      q.maxLevel = lvl;
    } else {
      double ptc[3];
      quat_rot(s_pose, mp_uvr + 3 * m, ptc);
      ptc[0] += s_pose[4];
      ptc[1] += s_pose[5];
      ptc[2] += s_pose[6];
      const float xc = (float)ptc[0], yc = (float)ptc[1], invzc = (float)(1.0 / ptc[2]);
      if (invzc < 0) {
        q.valid = false;
        return q;
      }
      const float u = (float)(P.fx * xc * invzc + P.cx), v = (float)(P.fy * yc * invzc + P.cy);
      if (u < 0 || u > P.width || v < 0 || v > P.height) {
        q.valid = false;
        return q;
      }
      const int oct = mp_level[m];
      q.rr = P.th * P.sf[oct & 7];
      q.x = u;
      q.y = v;
      q.ur_f = u - P.mbf * invzc;
      if (s_dir == 1) {
        q.minLevel = oct;
        q.maxLevel = -1;
      } else if (s_dir == 2) {
        q.minLevel = 0;
        q.maxLevel = oct;
      } else {
        q.minLevel = oct - 1;
        q.maxLevel = oct + 1;
      }
    }
    return q;
  };

#ifdef GL_MATCH_PROF
  const long long tp1 = clock64();
#endif
  // queries by window class (half size of the window = a factor x the scale of the level: 16 classes), so that the lanes of a wave
  // walk windows of like size: counting sort, stable within a class up to the order of the atomics (the order only decides
  // which thread takes which query - never a result)
  {
    auto cls_of = [&](int m) -> int {
      if (!mp_valid[m]) return 16;
      const int lvl = mp_level[m] & 7;
      return MODE == 0 ? (lvl | (((float)mp_viewcos[m] > 0.998) ? 0 : 8)) : lvl;
    };
    for (int m = tid; m < NP; m += TM) atomicAdd(&s_cls[cls_of(m)], 1);
    __syncthreads();
    if (tid == 0) {  // exclusive scan, the LARGEST windows first (they set the pace of a wave), the invalid ones last
      const int ord[17] = {15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 16};
      int run = 0;
      for (int k = 0; k < 17; ++k) {
        const int n = s_cls[ord[k]];
        s_cls[ord[k]] = run;
        run += n;
      }
    }
    __syncthreads();
    for (int m = tid; m < NP; m += TM) qorder[atomicAdd(&s_cls[cls_of(m)], 1)] = (uint16_t)m;
    __syncthreads();
  }
#ifdef GL_MATCH_PROF
  const long long tp1b = clock64();
  long long tp_r0 = 0;
#endif
  constexpr uint32_t EMPTY = 0xffffffffu;
  uint16_t* lst = (uint16_t*)cursor;  // 4 x TM entry indices: a thread's collected candidates (the cursors are dead after the grid build)
  uint4* cache = cache_all + (size_t)f * NP;

  // the window walk of query m (all lanes of a wave walk together: the loop and the flushes are wave-uniform): the three smallest
  // keys among its candidates NOT owned by a lower query (k0 <= k1 <= k2), their number, and `bad` when a key cannot hold them
  // key while walking: dist << 48 | position << 20 | octave << 12 | feature
  constexpr unsigned long long EMPTY64 = ~0ull;
#ifdef GL_MATCH_PROF
  long long pw_setup = 0, pw_loop = 0, pw_flush = 0, pr_a = 0, pr_b = 0, pr_c = 0;
  int pw_iter = 0, pw_nflush = 0, pw_walks = 0;
#define PW_T(v) const long long v = clock64()
#else
#define PW_T(v)
#endif
  auto walk = [&](bool act, int m, const Query& q, unsigned long long& k0, unsigned long long& k1, unsigned long long& k2, uint32_t& npass, bool& bad) {
    k0 = k1 = k2 = EMPTY64;
    npass = 0;
    bad = false;
    uint32_t seq = 0;
    PW_T(pw0);
    const float rr = q.rr, x = q.x, y = q.y;
    int x0 = 1, x1 = 0, y0 = 0, y1 = 0;
    if (act) {
      x0 = max(0, (int)floorf((x - 0.0f - rr) * P.col_inv));
      x1 = min(GC - 1, (int)ceilf((x - 0.0f + rr) * P.col_inv));
      y0 = max(0, (int)floorf((y - 0.0f - rr) * P.row_inv));
      y1 = min(GR - 1, (int)ceilf((y - 0.0f + rr) * P.row_inv));
      if (!(x0 < GC && x1 >= 0 && y0 < GR && y1 >= 0) || y0 > y1) x1 = x0 - 1;  // nothing to visit
    }
    const int minLevel = q.minLevel, maxLevel = q.maxLevel;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    uint32_t dm[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) dm[w] = act ? mp_desc[(size_t)m * 8 + w] : 0u;
    int cnt = 0;
    auto flush = [&]() {  // Hamming distances of the <= 4 collected candidates, their descriptors requested together
      PW_T(pf0);
      uint4 da[4], db[4];
      int roy[4], own[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        roy[j] = 0;
        own[j] = INT_MAX;
        da[j] = db[j] = make_uint4(0, 0, 0, 0);
        if (j < cnt) {
          const int ej = lst[j * TM + tid];
          roy[j] = rec_ro[ej].y;
          own[j] = owner[roy[j] >> 8];
          if (DL) {
            da[j] = rec_desc[2 * ej];
            db[j] = rec_desc[2 * ej + 1];
          } else {
            const uint4* src = (const uint4*)(feat_desc + (size_t)(roy[j] >> 8) * 8);
            da[j] = src[0];
            db[j] = src[1];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < cnt && own[j] >= m) {  // not taken on entry (-1), not owned by an earlier map point
          const int dist = __popc(dm[0] ^ da[j].x) + __popc(dm[1] ^ da[j].y) + __popc(dm[2] ^ da[j].z) + __popc(dm[3] ^ da[j].w) +
                           __popc(dm[4] ^ db[j].x) + __popc(dm[5] ^ db[j].y) + __popc(dm[6] ^ db[j].z) + __popc(dm[7] ^ db[j].w);
          const int oc = roy[j] & 0xff, idx = roy[j] >> 8;
          if (seq > 255u || oc > 15) bad = true;  // does not fit the packed record: this query walks in every round
          if (dist < 256) {  // (a distance of 256 never beats the initial best / second best of the reference's loop)
            const unsigned long long kx = ((unsigned long long)dist << 48) | ((unsigned long long)seq << 20) | ((unsigned long long)oc << 12) | (unsigned long long)idx;
            ++npass;
            if (kx < k2) {
              k2 = kx;
              if (k2 < k1) {
                const unsigned long long t = k1;
                k1 = k2;
                k2 = t;
              }
              if (k1 < k0) {
                const unsigned long long t = k0;
                k0 = k1;
                k1 = t;
              }
            }
          }
          ++seq;
        }
      }
      cnt = 0;
#ifdef GL_MATCH_PROF
      pw_flush += clock64() - pf0;
      ++pw_nflush;
#endif
    };
    int ix = x0 - 1, e = 0, e1 = 0;
#ifdef GL_MATCH_PROF
    dm[0] += __builtin_amdgcn_readfirstlane(dm[0]) & 0;  // (the descriptor's load has landed before the clock is read)
    const long long pw1 = clock64();
    pw_setup += pw1 - pw0;
    ++pw_walks;
#endif
    bool more = act && x0 <= x1;
    if (s_fast) {
      // level filter as a bit mask over the octaves 0..15 (getFeaturesInArea: minLevel / maxLevel, frame.cpp:121-177)
      uint32_t lm = 0xffffu;
      if (bCheckLevels) {
        lm = 0;
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (!(o < minLevel) && !(maxLevel >= 0 && o > maxLevel)) lm |= 1u << o;
      }
      const float4* rec16 = (const float4*)rec_uv;
      // Round 6: the loop is a chain of LDS round trips, not of instructions (profiles/r6_match_stations.txt: ~0.4 us per iteration with
      // the CU to itself) - a column's range, then an entry, then its feature's owner, one after the other.  Now an iteration waits ONCE:
      // the NEXT column's range is requested a column ahead, two entries are read per iteration, and the owner test moved into the
      // flush, where the (<= 4) owners are requested together with the descriptors.
      int ne = 0, ne1 = 0;  // the range of column ix + 1
      if (more) {
        ne = cell_ptr[x0 * GR + y0];
        ne1 = cell_ptr[x0 * GR + y1 + 1];
      }
      auto entry = [&](const float4 r, int ee) {
        const int pk = __float_as_int(r.w);
        bool ok = ((lm >> (pk & 0xff)) & 1u) != 0 && fabsf(r.x - x) < rr && fabsf(r.y - y) < rr;
        if (ok && r.z > 0) {
          const float er = q.ur_float ? fabsf(q.ur_f - r.z) : (float)fabs(q.ur_d - (double)r.z);
          if (er > rr) ok = false;
        }
        if (ok) {
          lst[cnt * TM + tid] = (uint16_t)ee;
          ++cnt;
        }
      };
      while (__any(more)) {
        if (more && e >= e1) {  // next column: cells (ix, y0..y1) are contiguous in the CSR
          ++ix;
          if (ix > x1) {
            more = false;
          } else {
            e = ne;
            e1 = ne1;
            if (ix < x1) {
              ne = cell_ptr[(ix + 1) * GR + y0];
              ne1 = cell_ptr[(ix + 1) * GR + y1 + 1];
            }
          }
        }
        const bool h0 = more && e < e1, h1 = more && e + 1 < e1;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        if (h0) r0 = rec16[e];
        if (h1) r1 = rec16[e + 1];
        if (h0) entry(r0, e);
        if (__any(cnt == 4)) flush();
        if (h1) entry(r1, e + 1);
        if (__any(cnt == 4)) flush();
        e += h1 ? 2 : (h0 ? 1 : 0);
#ifdef GL_MATCH_PROF
        ++pw_iter;
#endif
      }
    } else {
      while (__any(more)) {
        bool have = false;
        int ecur = 0;
        if (more) {
          if (e >= e1) {  // next column: cells (ix, y0..y1) are contiguous in the CSR
            ++ix;
            if (ix > x1) {
              more = false;
            } else {
              e = cell_ptr[ix * GR + y0];
              e1 = cell_ptr[ix * GR + y1 + 1];
            }
          }
          if (more && e < e1) {
            ecur = e++;
            have = true;
          }
        }
        if (have) {
          const int2 ro = rec_ro[ecur];
          const double2 fuv = rec_uv[ecur];
          const int oc = ro.y & 0xff;
          bool ok = true;
          if (bCheckLevels) {
            if (oc < minLevel) ok = false;
            if (maxLevel >= 0 && oc > maxLevel) ok = false;
          }
          const float distx = (float)(fuv.x - (double)x), disty = (float)(fuv.y - (double)y);
          if (!(fabsf(distx) < rr && fabsf(disty) < rr)) ok = false;
          if (ok && owner[ro.y >> 8] < m) ok = false;  // taken on entry (-1) or by an earlier map point
          if (ok) {
            const float ur = __int_as_float(ro.x);
            if (ur > 0) {
              const float er = q.ur_float ? fabsf(q.ur_f - ur) : (float)fabs(q.ur_d - (double)ur);
              if (er > rr) ok = false;
            }
          }
          if (ok) {
            lst[cnt * TM + tid] = (uint16_t)ecur;
            ++cnt;
          }
        }
        if (__any(cnt == 4)) flush();
      }
    }
    if (__any(cnt > 0)) flush();
#ifdef GL_MATCH_PROF
    pw_loop += clock64() - pw1;
#endif
  };
  // the reference's verdict on (best, second best) = the two smallest keys
  auto pack = [&](unsigned long long k) -> uint32_t {
    return k == EMPTY64 ? EMPTY : (uint32_t)((k >> 48) << 24) | (uint32_t)(((k >> 20) & 255u) << 16) | (uint32_t)(k & 0xffffu);
  };
  auto unpack = [&](uint32_t k) -> unsigned long long {
    return k == EMPTY ? EMPTY64 : ((unsigned long long)(k >> 24) << 48) | ((unsigned long long)((k >> 16) & 255u) << 20) | (unsigned long long)(k & 0xffffu);
  };
  auto decide = [&](unsigned long long a, unsigned long long b, bool ratio_test) -> int {
    if (a == EMPTY64) return -1;
    const int bestDist = (int)(a >> 48), bestLevel = (int)((a >> 12) & 255u), bestIdx = (int)(a & 0xfffu);
    const int bestDist2 = b == EMPTY64 ? 256 : (int)(b >> 48), bestLevel2 = b == EMPTY64 ? -1 : (int)((b >> 12) & 255u);
    if (!(bestDist <= 100)) return -1;  // TH_HIGH
    if (ratio_test && bestLevel == bestLevel2 && (float)bestDist > P.nn_ratio * (float)bestDist2) return -1;
    return bestIdx;
  };

  // A listed query walked by a WHOLE wave (round 6): lane c takes the range of the window's column c, a scan numbers the window's entries
  // t = 0 .. T - 1 in visiting order (columns ascending, a column's cells are contiguous in the CSR: ascending entry index), the lanes
  // take the entries t = lane, lane + 64, ... - level / window / u_right tests, owner, Hamming distance - and the two smallest keys
  // {distance, entry index = visiting order} meet in a butterfly: four LDS round trips where the lane-per-query walk makes ~20 dependent
  // iterations (profiles/r6_match_walk_ab.txt).  m is wave-uniform; returns the feature the reference's loop would choose, or -1.
  auto wave_min64 = [&](unsigned long long v) -> unsigned long long {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long w = __shfl_xor(v, o, 64);
      v = w < v ? w : v;
    }
    return v;
  };
  auto wave_walk = [&](int m) -> int {
    const int lane = tid & 63;
    const Query q = make_query(m);
    if (!q.valid) return -1;
    const float rr = q.rr, x = q.x, y = q.y;
    const int x0 = max(0, (int)floorf((x - 0.0f - rr) * P.col_inv)), x1 = min(GC - 1, (int)ceilf((x - 0.0f + rr) * P.col_inv));
    const int y0 = max(0, (int)floorf((y - 0.0f - rr) * P.row_inv)), y1 = min(GR - 1, (int)ceilf((y - 0.0f + rr) * P.row_inv));
    if (!(x0 < GC && x1 >= 0 && y0 < GR && y1 >= 0) || y0 > y1 || x0 > x1) return -1;
    const int ncols = x1 - x0 + 1;
    int cs = 0, cn = 0;
    if (lane < ncols) {
      cs = cell_ptr[(x0 + lane) * GR + y0];
      cn = cell_ptr[(x0 + lane) * GR + y1 + 1] - cs;
    }
    uint32_t dm[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) dm[w] = mp_desc[(size_t)m * 8 + w];
    int inc = cn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    const int cp = inc - cn, T = __shfl(inc, 63, 64);
    const int minLevel = q.minLevel, maxLevel = q.maxLevel;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    unsigned long long b0 = EMPTY64, b1 = EMPTY64;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      int e = -1;
      for (int c = 0; c < ncols; ++c) {  // (c is uniform: three broadcasts per column)
        const int pc = __shfl(cp, c, 64), nc = __shfl(cn, c, 64), sc = __shfl(cs, c, 64);
        if (t >= pc && t < pc + nc) e = sc + (t - pc);
      }
      if (e >= 0) {
        const int2 ro = rec_ro[e];
        const int oc = ro.y & 0xff, idx = ro.y >> 8;
        bool ok = true;
        if (bCheckLevels) {
          if (oc < minLevel) ok = false;
          if (maxLevel >= 0 && oc > maxLevel) ok = false;
        }
        if (s_fast) {
          const float4 r = ((const float4*)rec_uv)[e];
          if (!(fabsf(r.x - x) < rr && fabsf(r.y - y) < rr)) ok = false;
        } else {
          const double2 fuv = rec_uv[e];
          const float distx = (float)(fuv.x - (double)x), disty = (float)(fuv.y - (double)y);
          if (!(fabsf(distx) < rr && fabsf(disty) < rr)) ok = false;
        }
        if (ok && owner[idx] < m) ok = false;  // taken on entry (-1) or by an earlier map point
        if (ok) {
          const float ur = __int_as_float(ro.x);
          if (ur > 0) {
            const float er = q.ur_float ? fabsf(q.ur_f - ur) : (float)fabs(q.ur_d - (double)ur);
            if (er > rr) ok = false;
          }
        }
        if (ok) {
          uint4 da, db;
          if (DL) {
            da = rec_desc[2 * e];
            db = rec_desc[2 * e + 1];
          } else {
            const uint4* src = (const uint4*)(feat_desc + (size_t)idx * 8);
            da = src[0];
            db = src[1];
          }
          const int dist = __popc(dm[0] ^ da.x) + __popc(dm[1] ^ da.y) + __popc(dm[2] ^ da.z) + __popc(dm[3] ^ da.w) + __popc(dm[4] ^ db.x) +
                           __popc(dm[5] ^ db.y) + __popc(dm[6] ^ db.z) + __popc(dm[7] ^ db.w);
          if (dist < 256) {
            const unsigned long long kx = ((unsigned long long)dist << 48) | ((unsigned long long)e << 20) | ((unsigned long long)oc << 12) | (unsigned long long)idx;
            if (kx < b0) {
              b1 = b0;
              b0 = kx;
            } else if (kx < b1) {
              b1 = kx;
            }
          }
        }
      }
    }
    const unsigned long long m0 = wave_min64(b0);
    const unsigned long long m1 = wave_min64(b0 == m0 ? b1 : b0);  // (keys are distinct: the entry index is part of them)
    return decide(m0, m1, MODE == 0);
  };

  int rounds = 0;
#ifdef GL_MATCH_PROF
  int rewalks = 0;
#endif
  const int nq_rounds = ((NP + TM - 1) / TM) * TM;  // every thread makes the same number of trips (the walk is wave-uniform)
  // Round 6: the queries that must walk again in a later round (a few per cent) are LISTED and walked densely - a wave per 64 of them -
  // instead of where they sit: one such query in a wave made the whole wave walk, and round 2 cost half of round 1
  // (profiles/r6_match_stations.txt).  The list lives behind the walk's candidate lists in the grid build's dead cursors; a query that
  // finds it full walks in place, as before.
  uint16_t* rw_list = lst + 4 * TM;
  constexpr int RW_CAP = 2 * NCELL - 4 * TM;
  static_assert(RW_CAP >= 1024, "the cursors hold the walk's lists and the list of queries that walk again");
  while (true) {
    for (int i = tid; i < NF; i += TM) owner_n[i] = feat_taken[i] ? -1 : INT_MAX;
    if (tid == 0) {
      s_changed = 0;
      s_nrw = 0;
    }
    __syncthreads();
    PW_T(pr0);
    for (int sq = tid; sq < nq_rounds; sq += TM) {
      const bool in = sq < NP;
      const int m = in ? (int)qorder[sq] : NP;  // (the same thread has the query in every round: it reads its own record)
      int bestIdx = -1;
      bool need_walk = in && rounds == 0, listed = false;
      uint4 rec = make_uint4(EMPTY, EMPTY, EMPTY, 0);
      if (in && rounds > 0) {  // the record of round 1 minus what lower queries own now
        rec = cache[m];
        uint32_t a = EMPTY, b = EMPTY;
        int nav = 0;
        const uint32_t ks[3] = {rec.x, rec.y, rec.z};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (ks[j] != EMPTY && owner[ks[j] & 0xfffu] >= m) {
            if (nav == 0) a = ks[j];
            else if (nav == 1) b = ks[j];
            ++nav;
          }
        }
        if (rec.w != EMPTY && (nav >= 2 || rec.w <= 3u)) {
          bestIdx = decide(unpack(a), unpack(b), MODE == 0);
        } else {
          const int pos = atomicAdd(&s_nrw, 1);
          if (pos < RW_CAP) {
            rw_list[pos] = (uint16_t)m;
            listed = true;
          } else {
            need_walk = true;
          }
        }
      }
      if (__any(need_walk)) {
        Query q;
        q.valid = false;
        q.x = q.y = q.rr = 0.f;
        q.minLevel = q.maxLevel = -1;
        q.ratio_test = MODE == 0;
        q.ur_float = MODE == 1;
        q.ur_d = 0.0;
        q.ur_f = 0.f;
        if (need_walk) q = make_query(m);
        unsigned long long k0, k1, k2;
        uint32_t npass;
        bool bad;
        walk(need_walk && q.valid, m, q, k0, k1, k2, npass, bad);
        if (need_walk) {
          bestIdx = decide(k0, k1, MODE == 0);
          if (rounds == 0) cache[m] = make_uint4(pack(k0), pack(k1), pack(k2), bad ? EMPTY : npass);
#ifdef GL_MATCH_PROF
          else ++rewalks;
#endif
        }
      }
      if (in && !listed) {
        if (bestIdx >= 0) atomicMin(&owner_n[bestIdx], m);
      }
    }
    PW_T(pr1);
    if (rounds > 0) {  // the listed queries: a WAVE each (the owners they read are the previous round's: any order)
      __syncthreads();
      constexpr int NWV = TM / 64;
      const int nrw = min(s_nrw, RW_CAP);
      for (int k = tid >> 6; k < nrw; k += NWV) {
        const int m = (int)rw_list[k];
        const int bestIdx = wave_walk(m);
        if ((tid & 63) == 0) {
          if (bestIdx >= 0) atomicMin(&owner_n[bestIdx], m);
        }
#ifdef GL_MATCH_PROF
        ++rewalks;
#endif
      }
    }
    __syncthreads();
    PW_T(pr2);
    int ch = 0;
    for (int i = tid; i < NF; i += TM) {
      const int o = owner_n[i];
      if (o != owner[i]) ch = 1;
      owner[i] = o;
    }
    if (ch) s_changed = 1;
    __syncthreads();
    ++rounds;
#ifdef GL_MATCH_PROF
    if (rounds == 1) tp_r0 = clock64();
    else {
      pr_a += pr1 - pr0;
      pr_b += pr2 - pr1;
      pr_c += clock64() - pr2;
    }
#endif
    if (!s_changed || rounds > NP + 1) break;
    __syncthreads();
  }

  // ---- rotation consistency (orb_matcher.cpp:498-539, computeThreeMaxima :544-578) ---------------------
  if (MODE == 1 && P.check_orientation) {
    const float* feat_angle = feat_angle_all + (size_t)f * NF;
    const float* mp_angle = mp_angle_all + (size_t)f * NP;
    const float factor = 30 / 360.0f;
    auto bin_of = [&](int i) -> int {  // i: matched feature, owner[i]: its last-frame feature
      float rot = mp_angle[owner[i]] - feat_angle[i];
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == 30) bin = 0;
      return bin;
    };
    if (tid < 32) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < NF; i += TM) {
      const int o = owner[i];
      if (o >= 0 && o != INT_MAX) {
        const int b = bin_of(i);
        if (b >= 0 && b < 30) atomicAdd(&s_hist[b], 1);
      }
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
    for (int i = tid; i < NF; i += TM) {
      const int o = owner[i];
      if (o >= 0 && o != INT_MAX) {
        const int b = bin_of(i);
        if (b >= 0 && b < 30 && b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) owner[i] = INT_MAX;
      }
    }
    __syncthreads();
  }

#ifdef GL_MATCH_PROF
  const long long tp2 = clock64();
#endif
  // ---- outputs ------------------------------------------------------------------------------------------
  int32_t* feat_match = feat_match_all + (size_t)f * NF;
  int cnt = 0;
  for (int i = tid; i < NF; i += TM) {
    const int o = owner[i];
    const bool matched = o >= 0 && o != INT_MAX;
    feat_match[i] = matched ? o : -1;
    cnt += matched ? 1 : 0;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) s_scan[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < TM / 64; ++w) tot += s_scan[w];
    nmatches_all[f] = tot;
    if (counters) {  // GL_COUNTER_MATCH_ROUNDS / _UNITS: rounds of the owner fixed point, frames
      atomicAdd(&counters[1], rounds);
      atomicAdd(&counters[2], 1);
    }
#ifdef GL_MATCH_PROF
    feat_match[0] = (int)((tp1 - tp0) >> 4);
    feat_match[1] = (int)((tp2 - tp1) >> 4);
    feat_match[2] = (int)((clock64() - tp2) >> 4);
    feat_match[3] = rounds;
    feat_match[4] = (int)((tp1b - tp1) >> 4);   // the queries sorted by window class
    feat_match[5] = (int)((tp_r0 - tp1b) >> 4);  // round 1 (every query walks its window)
    feat_match[6] = rewalks;                     // thread 0's walks in later rounds
    feat_match[7] = (int)(pw_setup >> 4);        // thread 0's walks: query set-up (its descriptor's load), ...
    feat_match[8] = (int)(pw_loop >> 4);         // ... the window loop with its flushes, ...
    feat_match[9] = (int)(pw_flush >> 4);        // ... of which flushes
    feat_match[10] = pw_iter;                    // iterations of the window loop, flushes, walks
    feat_match[11] = pw_nflush;
    feat_match[12] = pw_walks;
    feat_match[13] = s_nrw;
    feat_match[14] = (int)(pr_a >> 4);           // later rounds: the queries from their records, the listed walks + barrier, the owners compared
    feat_match[15] = (int)(pr_b >> 4);
    feat_match[16] = (int)(pr_c >> 4);                      // queries listed to walk again in the last round
#endif
  }
}

}  // namespace

namespace {

int launch_match(int mode, gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP,
                 const double* feat_uv, const float* feat_ur, const int32_t* feat_oct, const uint8_t* feat_desc,
                 const uint8_t* feat_taken, const double* mp_uvr, const int32_t* mp_level, const double* mp_viewcos,
                 const uint8_t* mp_valid, const uint8_t* mp_desc, float th, float nn_ratio, int32_t* feat_match,
                 int32_t* nmatches, const double* pose_cw, const double* pose_lw, const float* feat_angle,
                 const float* mp_angle, int mono, int check_orientation, const int32_t* gate_nm = nullptr, int gate_min = 0) {
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  MatchP P;
  P.NF = NF;
  P.NP = NP;
  P.col_inv = static_cast<float>(GC) / cam->width;  // init_config.hpp:50-54
  P.row_inv = static_cast<float>(GR) / cam->height;
  P.th = th;
  P.nn_ratio = nn_ratio;
  P.sf[0] = 1.0f;  // init_config.hpp:63-79
  for (int i = 1; i < 8; ++i) P.sf[i] = P.sf[i - 1] * scale_factor;
  // camera::fx ... are float config scalars (config.h:38-48); PinholeCamera::fx() returns them as double
  const float cfx = (float)cam->fx, cfy = (float)cam->fy, ccx = (float)cam->cx, ccy = (float)cam->cy, cbf = (float)cam->bf;
  P.fx = cfx;
  P.fy = cfy;
  P.cx = ccx;
  P.cy = ccy;
  P.mbf = cbf;        // frame.cpp:23
  P.mb = cbf / cfx;   // frame.cpp:24
  P.width = cam->width;
  P.height = cam->height;
  P.mono = mono;
  P.check_orientation = check_orientation;
  size_t lds = (size_t)lds_ints(NF, NP) * sizeof(int32_t) + (size_t)NF * 24;
  // latency shape (no more frames than CUs): descriptors in LDS as well, if they fit
  const size_t lds_dl = lds + 8 + (size_t)NF * 32;
  bool dl = B <= c->ncu && lds_dl + 8 * 1024 <= (size_t)c->lds_max;
  if (c->opt.match_desc_lds >= 0) dl = c->opt.match_desc_lds != 0 && lds_dl + 8 * 1024 <= (size_t)c->lds_max;
  if (dl) lds = lds_dl;
  auto kern = mode == 0 ? (dl ? k_search_by_projection<0, true> : k_search_by_projection<0, false>)
                        : (dl ? k_search_by_projection<1, true> : k_search_by_projection<1, false>);
  GL_REQUIRE_LDS(c, lds);
  GL_HIP(gl::ensure_dynamic_lds(c, (const void*)kern, lds));
  void* cache = nullptr;  // 16 bytes per query: its three best candidates of round 1 (the kernel's header)
  {
    const int rc = gl::ctx_scratch_b(c, (size_t)B * NP * sizeof(uint4), &cache);
    if (rc != GL_OK) return rc;
  }
  kern<<<B, dl ? 1024 : T_M, lds, c->stream>>>(P, B, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, mp_uvr, mp_level, mp_viewcos,
                                   mp_valid, mp_desc, feat_match, nmatches, pose_cw, pose_lw, feat_angle, mp_angle, c->counters, (uint4*)cache, gate_nm, gate_min);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // namespace

extern "C" int gl_search_by_projection(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP,
                                       const double* feat_uv_dev, const float* feat_ur_dev, const int32_t* feat_oct_dev,
                                       const uint8_t* feat_desc_dev, const uint8_t* feat_taken_dev,
                                       const double* mp_uvr_dev, const int32_t* mp_level_dev,
                                       const double* mp_viewcos_dev, const uint8_t* mp_valid_dev,
                                       const uint8_t* mp_desc_dev, float th, float nn_ratio, int32_t* feat_match_dev,
                                       int32_t* nmatches_dev) {
  GL_REQUIRE(ctx && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NF >= 1 && NP >= 1, "bad B / NF / NP");
  GL_REQUIRE(NF <= 3072 && NP <= 4096, "NF / NP above the on-chip capacity (3072 features, 4096 map points)");
  GL_REQUIRE(cam->width > 0 && cam->height > 0, "camera width / height not set");
  GL_REQUIRE(feat_uv_dev && feat_ur_dev && feat_oct_dev && feat_desc_dev && feat_taken_dev && mp_uvr_dev &&
                 mp_level_dev && mp_viewcos_dev && mp_valid_dev && mp_desc_dev && feat_match_dev && nmatches_dev,
             "null buffer");
  return launch_match(0, ctx, cam, scale_factor, B, NF, NP, feat_uv_dev, feat_ur_dev, feat_oct_dev, feat_desc_dev,
                      feat_taken_dev, mp_uvr_dev, mp_level_dev, mp_viewcos_dev, mp_valid_dev, mp_desc_dev, th, nn_ratio,
                      feat_match_dev, nmatches_dev, nullptr, nullptr, nullptr, nullptr, 0, 0);
}

int gl::launch_match_frame_gated(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NL, const double* pose_cw,
                                 const double* pose_lw, const double* feat_uv, const float* feat_ur, const int32_t* feat_oct, const float* feat_angle,
                                 const uint8_t* feat_desc, const uint8_t* feat_taken, const double* last_pt, const uint8_t* last_valid,
                                 const int32_t* last_oct, const float* last_angle, const uint8_t* last_desc, float th, int mono, int check_orientation,
                                 int32_t* feat_match, int32_t* nmatches, const int32_t* gate_nm, int gate_min) {
  return launch_match(1, ctx, cam, scale_factor, B, NF, NL, feat_uv, feat_ur, feat_oct, feat_desc, feat_taken, last_pt, last_oct, nullptr, last_valid,
                      last_desc, th, 0.f, feat_match, nmatches, pose_cw, pose_lw, feat_angle, last_angle, mono, check_orientation, gate_nm, gate_min);
}

extern "C" int gl_search_by_projection_frame(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF,
                                             int NL, const double* pose_cw_dev, const double* pose_lw_dev,
                                             const double* feat_uv_dev, const float* feat_ur_dev,
                                             const int32_t* feat_oct_dev, const float* feat_angle_dev,
                                             const uint8_t* feat_desc_dev, const uint8_t* feat_taken_dev,
                                             const double* last_pt_dev, const uint8_t* last_valid_dev,
                                             const int32_t* last_oct_dev, const float* last_angle_dev,
                                             const uint8_t* last_desc_dev, float th, int mono, int check_orientation,
                                             int32_t* feat_match_dev, int32_t* nmatches_dev) {
  GL_REQUIRE(ctx && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NF >= 1 && NL >= 1, "bad B / NF / NL");
  GL_REQUIRE(NF <= 3072 && NL <= 4096, "NF / NL above the on-chip capacity (3072 features, 4096 map points)");
  GL_REQUIRE(cam->width > 0 && cam->height > 0, "camera width / height not set");
  GL_REQUIRE(pose_cw_dev && pose_lw_dev && feat_uv_dev && feat_ur_dev && feat_oct_dev && feat_angle_dev &&
                 feat_desc_dev && feat_taken_dev && last_pt_dev && last_valid_dev && last_oct_dev && last_angle_dev &&
                 last_desc_dev && feat_match_dev && nmatches_dev,
             "null buffer");
  return launch_match(1, ctx, cam, scale_factor, B, NF, NL, feat_uv_dev, feat_ur_dev, feat_oct_dev, feat_desc_dev,
                      feat_taken_dev, last_pt_dev, last_oct_dev, nullptr, last_valid_dev, last_desc_dev, th, 0.f,
                      feat_match_dev, nmatches_dev, pose_cw_dev, pose_lw_dev, feat_angle_dev, last_angle_dev, mono,
                      check_orientation);
}
