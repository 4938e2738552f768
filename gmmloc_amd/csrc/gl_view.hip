// GMMLoc::associateMapElements (gmmloc_opt.cpp:115-135) for B key-frames:
//   GMM::renderView            (gaussian_mixture.cpp:271-371)   A3 + A4
//   GMM::searchCorrespondence  (gaussian_mixture.cpp:484-534)   A5
// One workgroup per key-frame (256 threads in batches, 1 024 when the views are fewer than the CUs).
//   phase 1  all K components in parallel: view-cosine cull (degenerate planes), projection
//            J R Sigma R^T J^T (GMMUtility::projectGaussian, gmm_utils.cpp:121-146 with
//            PinholeCamera::project3, pinhole_camera.cpp:68-150), 2-D eigenvalue cull; ordered
//            compaction (ballot / popcount prefix) keeps the reference's k = 0..K-1 order.
//   phase 2  the order-dependent occlusion merge (:328-355), in rounds of MG candidates: the
//            Bhattacharyya distances are evaluated in parallel (exact screen, then the reference's
//            expression for the surviving pairs), the (distance, index) argmin prefers the lower
//            index on ties exactly like the sequential `dist < min_dist` scan, and one wave replays
//            the replace-in-place / append decisions in order (details at the phase).
//   phase 3  stable rank sort by depth, descending (:362-364).
//   phase 4  per feature exact 5-NN on the 2-D means (nanoflann's result order: ascending,
//            ties by lower index) + MDist2 < 9 gate, in kNN order.
// Compiled with -ffp-contract=off: cull / merge decisions follow the fp64 CPU order.
#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {

#ifdef GL_VIEW_PROF  // debug build (tools/prof_view.py): phase cycles are returned in place of the view list
#define VPROF(x) const long long x = clock64()
#define GL_VIEW_PROF_ARG (view_ids_out + (size_t)f * view_cap)
#else
#define VPROF(x)
#define GL_VIEW_PROF_ARG nullptr
#endif

constexpr int SLOT_LDS = 640;  // accepted 2-D components kept in LDS (40 KB); V is 100-300 on the EuRoC maps
constexpr int REC = 8;  // m0 m1 c00 c01 c10 c11 det depth
constexpr int MG = 16;  // candidates per merge round (one DPP row)

struct ViewK {
  double fx, fy, cx, cy;
  int width, height;
};

// GMMUtility::BHCoefficient<GaussianComponent2d> (gmm_utils.h:30-52)
GL_DEV double bh2(const double* g0, const double* g1) {
  double cov[4], inv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cov[i] = (g0[2 + i] + g1[2 + i]) / 2.0;
  const double d0 = g1[0] - g0[0], d1 = g1[1] - g0[1];
  inv2(cov, inv);
  const double r0 = d0 * inv[0] + d1 * inv[2];
  const double r1 = d0 * inv[1] + d1 * inv[3];
  double q = r0 * d0 + r1 * d1;
  q /= 8.0;
  const double l = log(det2(cov) / sqrt(g0[6] * g1[6])) / 2.0;
  return q + l;
}

// true => BHCoefficient(g0, g1) is NaN, +inf or above 1.0: the pair can neither pass the 0.8 threshold nor be an
// argmin that matters.  With cov = (cov0 + cov1) / 2 positive definite and not closer to singular than
// 1 - rho^2 = 1e-10, the Mahalanobis part d^T cov^-1 d / 8 is above 2 when d^T adj(cov) d > 16 det (no division;
// the reference's own rounding of that term is far inside the margin to the 1.8 that would be needed), and the
// log part is above -1 when det^2 >= e^-4 det0 det1 (it is never negative for positive definite inputs).  The
// range guards keep the products finite; a non-positive det0 det1 makes the reference's value NaN or +inf.
GL_DEV bool bh2_far(const double* g0, const double* g1) {
  const double c00 = (g0[2] + g1[2]) / 2.0, c01 = (g0[3] + g1[3]) / 2.0, c10 = (g0[4] + g1[4]) / 2.0,
               c11 = (g0[5] + g1[5]) / 2.0;
  const double d0 = g1[0] - g0[0], d1 = g1[1] - g0[1];
  const double det = c00 * c11 - c10 * c01, P = g0[6] * g1[6];
  const double qa = (d0 * c11 - d1 * c10) * d0 + (d1 * c00 - d0 * c01) * d1;
  const double as = c01 - c10;  // rounding-level in a projected covariance; anything else is not screened
  return qa > 16.0 * det && c00 > 0.0 && det >= 1e-10 * (c00 * c11) && as * as <= 1e-12 * det && det > 0.0 &&
         det < 1e150 && qa < 1e300 && P < 1e300 && det * det >= 0.0184 * P;
}

GL_DEV double lane_bcast(double v, int src) {  // src wave-uniform
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// one butterfly stage of the lexicographic (distance, slot) argmin; dp rides along
template <int CTRL>
GL_DEV void lexmin_stage(double& d, int& i, double& dp) {
  const double od = dpp_f64<CTRL>(d), odp = dpp_f64<CTRL>(dp);
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false);
  const bool t = od < d || (od == d && oi < i);
  d = t ? od : d;
  dp = t ? odp : dp;
  i = t ? oi : i;
}

// dup |= (key == key of the lane k places round the 16-lane row), k = 1..K
template <int K>
GL_DEV void dup_stages(int key, bool& dup) {
  if constexpr (K > 0) {
    dup = dup || key == __builtin_amdgcn_update_dpp(0, key, 0x120 + K, 0xF, 0xF, false);  // row_ror:K
    dup_stages<K - 1>(key, dup);
  }
}

// order-preserving 64-bit key of a double that is not NaN (-0 folded onto +0)
GL_DEV unsigned long long dkey(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d + 0.0);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// GMM::searchCorrespondence (gaussian_mixture.cpp:484-534) for the N features of one view: exact knn-NN on the
// 2-D means of the V rendered components (nanoflann's result order: ascending distance, ties by lower index)
// + the MDist2 < 9 gate, in kNN order.  One feature per thread and pass; the means are staged in LDS in chunks
// of MCH (one chunk, staged once, for any realistic view).  Entries are taken 32 at a time (the first 8 alone):
// first only the distances and one bit per entry that beats the current knn-th neighbour, then the insertion
// network for the set bits, in index order and re-checked against the bound as it tightens.  A lane inserts
// ~25 times over a few hundred entries, but some lane of the wave does at almost every entry: testing and
// inserting in one loop makes every entry pay for the network.  KNN = 0: knn at run time.
template <int KNN, int T_VIEW>
GL_DEV void search_correspondence(int knn_rt, double2* s_mean, const double* __restrict__ sorted,
                                  const int32_t* __restrict__ sorted_id, const double* __restrict__ uv, int nf, int N,
                                  int V, int32_t* __restrict__ cand_out, int32_t* __restrict__ ncand_out,
                                  int32_t* prof = nullptr) {
#ifdef GL_VIEW_PROF
  long long kp_p1 = 0, kp_p2 = 0, kp_tail = 0;
  int kp_it = 0, kp_net = 0;
#endif
  const int knn = KNN ? KNN : knn_rt;
  constexpr int KC = KNN ? KNN : 8;
  constexpr int MCH = SLOT_LDS * REC / 2;
  const int tid = threadIdx.x;
  bool staged = false;
  for (int n0 = 0; n0 < N; n0 += T_VIEW) {  // uniform trip count: the staging barriers sit inside
    const int n = n0 + tid;
    const bool live = n < nf && V > 0;
    double dist[KC], worst = __builtin_inf(), fu = 0.0, fv = 0.0;
    int idx[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      dist[i] = __builtin_inf();
      idx[i] = -1;
    }
    if (live) {
      fu = uv[(size_t)n * 2];
      fv = uv[(size_t)n * 2 + 1];
    }
    for (int j0 = 0; j0 < V; j0 += MCH) {
      const int jn = min(MCH, V - j0);
      if (!staged) {
        __syncthreads();
        for (int j = tid; j < jn; j += T_VIEW)
          s_mean[j] = make_double2(sorted[(size_t)(j0 + j) * REC], sorted[(size_t)(j0 + j) * REC + 1]);
        __syncthreads();
        staged = V <= MCH;
      }
      if (!live) continue;
      for (int jb = 0; jb < jn; jb += (jb == 0 ? 8 : 32)) {
        const int je = min(jb == 0 ? 8 : 32, jn - jb);
        unsigned mask = 0;
        VPROF(t_k0);
#pragma unroll
        for (int t8 = 0; t8 < 32; t8 += 8) {  // 8 loads in flight; entries past je (inside the LDS block) are masked
          if (t8 < je) {
            double2 mj[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) mj[t] = s_mean[jb + t8 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const double d0 = fu - mj[t].x, d1 = fv - mj[t].y;
              const double cd = d0 * d0 + d1 * d1;  // kdtree_distance (gaussian_mixture.h:71-76)
              mask |= (cd < worst) ? (1u << (t8 + t)) : 0u;
            }
          }
        }
        if (je < 32) mask &= (1u << je) - 1u;
        VPROF(t_k1);
        while (mask) {
#ifdef GL_VIEW_PROF
          kp_it++;
#endif
          const int t = __ffs(mask) - 1;
          mask &= mask - 1;
          const double2 mj = s_mean[jb + t];
          const double d0 = fu - mj.x, d1 = fv - mj.y;
          const double cd = d0 * d0 + d1 * d1;
          if (!(cd < worst)) continue;  // the bound has tightened since the bit was set
#ifdef GL_VIEW_PROF
          kp_net++;
#endif
          // sorted insertion, after the equal ones (the sequential `cd < dist[i]` swap chain, gaussian_mixture.cpp /
          // nanoflann KNNResultSet), as independent selects: dist is ascending, so c[i] = cd < dist[i] is monotone
          const int ci = j0 + jb + t;
          bool c[KC];
#pragma unroll
          for (int i = 0; i < KC; ++i) c[i] = cd < dist[i];
#pragma unroll
          for (int i = KC - 1; i >= 0; --i) {
            if (i < knn) {
              const bool prev = i > 0 && c[i > 0 ? i - 1 : 0];
              const double nd = prev ? dist[i > 0 ? i - 1 : 0] : cd;
              const int ni = prev ? idx[i > 0 ? i - 1 : 0] : ci;
              dist[i] = c[i] ? nd : dist[i];
              idx[i] = c[i] ? ni : idx[i];
            }
          }
          if (KNN) {
            worst = dist[KC - 1];
          } else {
#pragma unroll
            for (int i = 0; i < KC; ++i) worst = (i == knn - 1) ? dist[i] : worst;
          }
        }
#ifdef GL_VIEW_PROF
        kp_p1 += t_k1 - t_k0;
        kp_p2 += clock64() - t_k1;
#endif
      }
    }
    VPROF(t_k2);
    if (n >= N) continue;
    int32_t* co = cand_out + (size_t)n * knn;
    int m = 0;
    if (live) {
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        if (i < knn && idx[i] >= 0) {
          const double* sr = &sorted[(size_t)idx[i] * REC];
          if (mdist2_2d(sr, sr + 2, fu, fv) < 9.0) co[m++] = sorted_id[idx[i]];  // check_mdist2 (:521-527)
        }
      }
    }
    ncand_out[n] = m;
    for (; m < knn; ++m) co[m] = -1;
#ifdef GL_VIEW_PROF
    kp_tail += clock64() - t_k2;
#endif
  }
#ifdef GL_VIEW_PROF
  // wave 0: iterations of the divergent insertion loop = max over the lanes
  int wit = kp_it, wnet = kp_net;
  for (int o = 1; o < 64; o <<= 1) {
    wit = max(wit, __shfl_xor(wit, o, 64));
    wnet = max(wnet, __shfl_xor(wnet, o, 64));
  }
  if (tid == 0 && prof) {
    prof[11] = (int)(kp_p1 >> 4);
    prof[12] = (int)(kp_p2 >> 4);
    prof[13] = (int)(kp_tail >> 4);
    prof[14] = kp_it;
    prof[15] = wit;
    prof[16] = wnet;
  }
#endif
}

// T_VIEW threads per view: 256 for batches (3 views per CU), 1024 when the views are fewer than the CUs - every
// phase is a chain of dependent fp64 instructions and LDS reads, and one wave per SIMD hides none of it.
template <int T_VIEW>
__global__ __launch_bounds__(T_VIEW) void k_search2d(ViewK vk, int B, int K, int slot_lds, const double* __restrict__ rec12,
                                                     const double* __restrict__ cov3, const double* __restrict__ axis,
                                                     const uint8_t* __restrict__ flags,
                                                     const double* __restrict__ pose_all, int N,
                                                     const double* __restrict__ uv_all,
                                                     const int32_t* __restrict__ nfeat_all, int knn,
                                                     int32_t* __restrict__ cand_out, int32_t* __restrict__ ncand_out,
                                                     int view_cap, int32_t* __restrict__ view_ids_out,
                                                     int32_t* __restrict__ nview_out, double* __restrict__ scratch) {
  constexpr int NW_VIEW = T_VIEW / 64;
  constexpr int NEAR_CAP = 2 * T_VIEW;  // pairs kept for the exact Bhattacharyya distance per round
  __shared__ int s_wcount[NW_VIEW];
  const int f = blockIdx.x;
  if (f >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // per-view scratch: candidates, slots, sorted (each K records) + ids
  double* cand = scratch + (size_t)f * ((size_t)K * (3 * REC + 2));
  double* slot = cand + (size_t)K * REC;
  double* sorted = slot + (size_t)K * REC;  // m0 m1 inv00 inv01 inv10 inv11 (6 of REC used)
  int32_t* cand_id = reinterpret_cast<int32_t*>(sorted + (size_t)K * REC);
  int32_t* slot_id = cand_id + K;
  int32_t* sorted_id = slot_id + K;  // (K*2 doubles hold 4K int32: ids use 3K)

#ifdef GL_VIEW_PROF
  const long long tp0 = clock64();
#endif
  const SE3 T = se3_load(pose_all + (size_t)f * 7);
  double R[9];
  qtoR(T.r, R);
  // t_w_c = -(rot_c_w.inverse() * t_c_w)   (:289)
  double twc[3];
  {
    const double n2 = T.r.x * T.r.x + T.r.y * T.r.y + T.r.z * T.r.z + T.r.w * T.r.w;
    const Quat qi{-T.r.x / n2, -T.r.y / n2, -T.r.z / n2, T.r.w / n2};
    double rt[3];
    qrot(qi, T.t, rt);
    twc[0] = -rt[0];
    twc[1] = -rt[1];
    twc[2] = -rt[2];
  }
  const double view_cos_thresh = cos(78.0 * M_PI / 180.0);

  // ---- phase 1: project + cull, ordered compaction -----------------------------------------
  int C = 0;
  for (int k0 = 0; k0 < K; k0 += T_VIEW) {
    const int k = k0 + tid;
    bool keep = false;
    double rec[REC];
    if (k < K) {
      const double mu[3] = {rec12[(size_t)k * 12], rec12[(size_t)k * 12 + 1], rec12[(size_t)k * 12 + 2]};
      bool pass = true;
      if (flags[k] & 1) {  // STEP.0 check view cos (:283-302)
        double po[3] = {mu[0] - twc[0], mu[1] - twc[1], mu[2] - twc[2]};
        const double nn = sqrt(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]);
        po[0] /= nn;
        po[1] /= nn;
        po[2] /= nn;
        const double vc = fabs(po[0] * axis[(size_t)k * 9] + po[1] * axis[(size_t)k * 9 + 3] + po[2] * axis[(size_t)k * 9 + 6]);
        if (vc < view_cos_thresh) pass = false;
      }
      if (pass) {
        double r[3];
        qrot(T.r, mu, r);
        const double x = r[0] + T.t[0], y = r[1] + T.t[1], z = r[2] + T.t[2];
        const double rz = 1.0 / z;
        double kx = x * rz, ky = y * rz;
        const double rz2 = rz * rz;
        const double J[6] = {vk.fx * rz, 0.0, -vk.fx * x * rz2, 0.0, vk.fy * rz, -vk.fy * y * rz2};
        kx = vk.fx * kx + vk.cx;
        ky = vk.fy * ky + vk.cy;
        const bool visible = kx >= 0.0 && ky >= 0.0 && kx < (double)vk.width && ky < (double)vk.height;
        if (visible && z > 0.0) {
          // jacob_proj * rot * cov3d * rot^T * jacob_proj^T, evaluated left to right
          double JR[6], JRS[6], M[6], c2[4];
          const double* S = cov3 + (size_t)k * 9;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              double s = 0.0;
#pragma unroll
              for (int l = 0; l < 3; ++l) s += J[i * 3 + l] * R[l * 3 + j];
              JR[i * 3 + j] = s;
            }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              double s = 0.0;
#pragma unroll
              for (int l = 0; l < 3; ++l) s += JR[i * 3 + l] * S[l * 3 + j];
              JRS[i * 3 + j] = s;
            }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              double s = 0.0;
#pragma unroll
              for (int l = 0; l < 3; ++l) s += JRS[i * 3 + l] * R[j * 3 + l];
              M[i * 3 + j] = s;
            }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              double s = 0.0;
#pragma unroll
              for (int l = 0; l < 3; ++l) s += M[i * 3 + l] * J[j * 3 + l];
              c2[i * 2 + j] = s;
            }
          double w[2], V[4];
          eig_sym<2>(c2, w, V);
          if (!(w[0] < 4.0 && w[1] < 4.0)) {  // check_cov_2d (:311-317)
            keep = true;
            rec[0] = kx;
            rec[1] = ky;
            rec[2] = c2[0];
            rec[3] = c2[1];
            rec[4] = c2[2];
            rec[5] = c2[3];
            rec[6] = det2(c2);
            double rr[3];
            qrot(T.r, mu, rr);
            rec[7] = rr[2] + T.t[2];  // proj_d_ (:322-325)
          }
        }
      }
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wcount[wave] = __popcll(m);
    __syncthreads();
    int base = C;
    for (int w = 0; w < wave; ++w) base += s_wcount[w];
    int tot = 0;
    for (int w = 0; w < NW_VIEW; ++w) tot += s_wcount[w];
    if (keep) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
      for (int i = 0; i < REC; ++i) cand[(size_t)pos * REC + i] = rec[i];
      cand_id[pos] = k;
    }
    C += tot;
    __syncthreads();
  }

#ifdef GL_VIEW_PROF
  const long long tp1 = clock64();
#endif
  // ---- phase 2: sequential occlusion merge (:328-355), MG candidates per round ----------------
  // The reference visits the candidates one by one: nearest accepted component by Bhattacharyya distance,
  // then replace-if-nearer / discard (distance < 0.8) or append.  Only the decisions are sequential; the
  // distances are not.  A round takes the next MG candidates and
  //   1. screens every (accepted slot, candidate) pair with bh2_far - a pair whose distance provably is
  //      >= 0.8 or NaN can influence neither the threshold test nor the argmin below it - and compacts the
  //      others into a list;
  //   2. evaluates BHCoefficient exactly for the listed pairs and for the pairs inside the round, and takes
  //      the (distance, slot) argmin per candidate with LDS atomics (ordered key, then lowest slot);
  //   3. replays the MG decisions in order on wave 0: a slot overwritten or appended earlier in the round
  //      is seen through the in-round distances; if the slot a candidate's argmin points at was overwritten
  //      the round ends before that candidate and the next round starts there.
  // The accepted list lives in LDS (beyond slot_lds entries: in the global scratch).
  extern __shared__ __attribute__((aligned(16))) double lds_slot[];  // SLOT_LDS x REC
  __shared__ double s_cr[MG * REC];                                  // the round's candidate records
  __shared__ int s_cid[MG];
  __shared__ double s_D[MG * MG];           // [g][h], h < g: BHCoefficient(record h as a slot, candidate g)
  __shared__ unsigned long long s_key[MG];  // min ordered key of the exact distances to the old slots
  __shared__ double s_best[MG];
  __shared__ int s_bj[MG];
  __shared__ int s_near[NEAR_CAP];
  __shared__ int s_ncnt;
  __shared__ unsigned s_inter;  // bit g: some earlier candidate of the round is below the threshold of candidate g
  __shared__ int s_state[4];  // nslots, next candidate, spill request
  __shared__ int s_slot_id[SLOT_LDS];
  // the accepted list: LDS, or (beyond slot_lds entries) the global scratch - never through one generic
  // pointer, whose loads would be flat instructions with global-memory latency
  bool in_lds = true;
  auto slot_ld = [&](int j, int i) -> double { return in_lds ? lds_slot[j * REC + i] : slot[(size_t)j * REC + i]; };
  auto slot_rec = [&](int j, double* r) {
    if (in_lds) {
#pragma unroll
      for (int i = 0; i < REC; ++i) r[i] = lds_slot[j * REC + i];
    } else {
#pragma unroll
      for (int i = 0; i < REC; ++i) r[i] = slot[(size_t)j * REC + i];
    }
  };
  int nslots = 0;
  int c0 = 0, pf_c0 = -1, pf_id = 0;
  double pf_rec = 0.0;
#ifdef GL_VIEW_PROF
  long long pr_screen = 0, pr_exact = 0, pr_resolve = 0, pr_q0 = 0, pr_q1 = 0, pr_q2 = 0;
  int pr_rounds = 0, pr_near = 0;
#endif
  while (c0 < C) {
    VPROF(t_r0);
    const int Gb = min(MG, C - c0);
    const int n0 = nslots;
    // the records of the round were fetched while the previous round was resolved, unless that one ended early
    if (pf_c0 != c0) {
      if (tid < Gb * REC) pf_rec = cand[(size_t)c0 * REC + tid];
      if (tid < Gb) pf_id = cand_id[c0 + tid];
    }
    if (tid < Gb * REC) s_cr[tid] = pf_rec;
    if (tid < Gb) s_cid[tid] = pf_id;
    pf_c0 = c0 + MG;
    if (tid < min(MG, C - pf_c0) * REC) pf_rec = cand[(size_t)pf_c0 * REC + tid];
    if (tid < min(MG, C - pf_c0)) pf_id = cand_id[pf_c0 + tid];
    if (tid < MG) {
      s_key[tid] = ~0ull;
      s_bj[tid] = 0x7fffffff;
      s_best[tid] = 1.7976931348623157e308;
    }
    if (tid == 0) {
      s_ncnt = 0;
      s_inter = 0;
    }
    __syncthreads();

    // 1. screen: pair p = (slot p / MG, candidate p % MG)
    const int npair = n0 * MG;
    double crg[REC];  // this thread's candidate of the round: tid % MG in every pass (T_VIEW % MG == 0)
#pragma unroll
    for (int i = 0; i < REC; ++i) crg[i] = s_cr[(tid & (MG - 1)) * REC + i];
    for (int p0 = 0; p0 < npair; p0 += 2 * T_VIEW) {  // two independent pairs in flight per thread
      bool near[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = p0 + u * T_VIEW + tid;
        near[u] = false;
        if (p < npair && (p & (MG - 1)) < Gb) {
          double sr[REC];
          slot_rec(p / MG, sr);
          near[u] = !bh2_far(sr, crg);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned long long m = __ballot(near[u]);
        if (m) {  // wave-uniform
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_ncnt, __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          if (near[u] && pos < NEAR_CAP) s_near[pos] = p0 + u * T_VIEW + tid;
        }
      }
    }

    __syncthreads();
    VPROF(t_r1);
    const int ncnt = s_ncnt;
    const bool listed = ncnt <= NEAR_CAP;  // else (degenerate view): every pair is evaluated exactly, twice
    // 2. exact distances
    double dd[NEAR_CAP / T_VIEW];
    int pp[NEAR_CAP / T_VIEW];
    if (listed) {
#pragma unroll
      for (int i = 0; i < NEAR_CAP / T_VIEW; ++i) {
        const int t = tid + i * T_VIEW;
        pp[i] = -1;
        dd[i] = 0.0;
        if (t < ncnt) {
          const int p = s_near[t];
          double sr[REC];
          slot_rec(p / MG, sr);
          const double d = bh2(sr, &s_cr[(p & (MG - 1)) * REC]);
          if (d < 1.7976931348623157e308) {
            pp[i] = p;
            dd[i] = d;
            atomicMin(&s_key[p & (MG - 1)], dkey(d));
          }
        }
      }
    } else {
      for (int p = tid; p < npair; p += T_VIEW) {
        if ((p & (MG - 1)) >= Gb) continue;
        double sr[REC];
        slot_rec(p / MG, sr);
        const double d = bh2(sr, &s_cr[(p & (MG - 1)) * REC]);
        if (d < 1.7976931348623157e308) atomicMin(&s_key[p & (MG - 1)], dkey(d));
      }
    }
    {
      const int q = T_VIEW - 1 - tid, g = q / MG, h = q % MG;
      if (q < MG * MG && h < g && g < Gb) {
        const double d = bh2(&s_cr[h * REC], &s_cr[g * REC]);
        s_D[g * MG + h] = d;
        if (d < 0.8) atomicOr(&s_inter, 1u << g);
      }
    }
    __syncthreads();
    if (listed) {
#pragma unroll
      for (int i = 0; i < NEAR_CAP / T_VIEW; ++i) {
        if (pp[i] >= 0 && dkey(dd[i]) == s_key[pp[i] & (MG - 1)]) {
          atomicMin(&s_bj[pp[i] & (MG - 1)], pp[i] / MG);
          s_best[pp[i] & (MG - 1)] = dd[i];
        }
      }
    } else {
      for (int p = tid; p < npair; p += T_VIEW) {
        if ((p & (MG - 1)) >= Gb) continue;
        double sr[REC];
        slot_rec(p / MG, sr);
        const double d = bh2(sr, &s_cr[(p & (MG - 1)) * REC]);
        if (d < 1.7976931348623157e308 && dkey(d) == s_key[p & (MG - 1)]) {
          atomicMin(&s_bj[p & (MG - 1)], p / MG);
          s_best[p & (MG - 1)] = d;
        }
      }
    }
    __syncthreads();
    VPROF(t_r2);
    // 3. the decisions, in order, on wave 0.  Lane h (mod MG) holds candidate h as a member of the round:
    //    whether it still owns a slot (live), which one (tgt), its depth, and its distances to the later
    //    candidates.  Most rounds need no order at all: when no two candidates of the round are below the
    //    threshold of each other and no two share their old argmin, every decision depends on the old slots
    //    alone and the lanes decide in parallel (the appended ones numbered by a prefix count).  Otherwise
    //    step g takes the (distance, slot) argmin over the live members with DPP stages, merges it with the
    //    argmin over the old slots, decides, and updates the members.  A slot overwritten earlier in the
    //    round is seen through its live member; if it was the old argmin of candidate g the round ends
    //    before g (the next round starts there).
    if (wave == 0) {
      const int h = lane & (MG - 1);
      const unsigned gb_mask = (1u << Gb) - 1u;  // Gb <= 16
      double my_best = s_best[h];
      int my_bj = s_bj[h];
      const double my_dpc = s_cr[h * REC + 7];
      if (!(my_best < 0.8)) {  // no old slot below the threshold: which one was the argmin does not matter
        my_best = 1.7976931348623157e308;
        my_bj = 0x7fffffff;
      }
      const double my_dpo = my_bj != 0x7fffffff ? slot_ld(my_bj, 7) : 0.0;
      // per candidate, as bit masks: an old slot below the threshold; nearer than that slot; some earlier
      // candidate of the round below the threshold (only then the members' distances can matter)
      const unsigned near_old = (unsigned)__ballot(my_best < 0.8) & gb_mask;
      const unsigned rep_old = (unsigned)__ballot(my_dpc < my_dpo) & gb_mask;
      const unsigned inter = s_inter & gb_mask;
      const int appends = __popc(~near_old & gb_mask);
      bool fast = inter == 0 && !(in_lds && n0 + appends >= slot_lds);
      if (fast && __popc(near_old) >= 2) {  // two candidates with the same old argmin interact through it
        const int key = ((near_old >> h) & 1u) ? my_bj : -1 - h;
        bool dup = false;
        dup_stages<MG - 1>(key, dup);
        fast = !__ballot(dup);
      }
      bool live = false;
      int tgt = -1;
      int ns = n0, done = Gb, spill = 0;
#ifdef GL_VIEW_PROF
      pr_q0 += fast ? 1 : 0;
#endif
      if (fast) {
        const bool nr = (near_old >> h) & 1u;
        const int dst = nr ? (((rep_old >> h) & 1u) ? my_bj : -2) : n0 + __popc(~near_old & gb_mask & ((1u << h) - 1u));
        live = h < Gb && dst >= 0;
        tgt = dst;
        ns = n0 + appends;
      } else {
        bool stop = false;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
          if (g < Gb && !stop) {  // wave-uniform
            const int bjs_g = __builtin_amdgcn_readlane(my_bj, g);
            if (((near_old >> g) & 1u) && __ballot(live && tgt == bjs_g)) {
              stop = true;  // the old argmin slot no longer holds what the distance was taken to
              done = g;
            } else {
              int dst;
              if (!((inter >> g) & 1u)) {
                dst = ((near_old >> g) & 1u) ? (((rep_old >> g) & 1u) ? bjs_g : -2) : ns;
              } else {
                const double bst_g = lane_bcast(my_best, g), dpc_g = lane_bcast(my_dpc, g), dpo_g = lane_bcast(my_dpo, g);
                const double Dg = s_D[g * MG + h];                     // meaningful for h < g only, where live can be set
                const bool valid = live && Dg < 1.7976931348623157e308;  // NaN / inf never win
                double d = valid ? Dg : 1.7976931348623157e308, dp = my_dpc;
                int i = valid ? tgt : 0x7fffffff;
                lexmin_stage<0xB1>(d, i, dp);   // quad_perm [1,0,3,2]
                lexmin_stage<0x4E>(d, i, dp);   // quad_perm [2,3,0,1]
                lexmin_stage<0x141>(d, i, dp);  // row_half_mirror
                lexmin_stage<0x140>(d, i, dp);  // row_mirror
                const bool take = d < bst_g || (d == bst_g && i < bjs_g);
                const double best = take ? d : bst_g, dep = take ? dp : dpo_g;
                const int bj = take ? i : bjs_g;
                dst = ns;                                         // append
                if (best < 0.8) dst = (dpc_g < dep) ? bj : -2;  // replace the farther component / discard
                dst = __builtin_amdgcn_readfirstlane(dst);
              }
              if (dst >= 0) {
                live = live && tgt != dst;
                if (h == g) {
                  live = true;
                  tgt = dst;
                }
                if (dst == ns) {
                  ++ns;
                  if (ns == slot_lds && in_lds) {  // rare: the list continues in the global scratch
                    stop = true;
                    done = g + 1;
                    spill = 1;
                  }
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int e0 = 0; e0 < MG * REC; e0 += 64) {  // element = (member, word): the surviving members write their records
        const int e = e0 + lane, m = e >> 3;
        const int m_tgt = __shfl(tgt, m, 64);
        const int m_live = __shfl((int)live, m, 64);
        if (m_live) {
          if (in_lds)
            lds_slot[m_tgt * REC + (e & 7)] = s_cr[e];
          else
            slot[(size_t)m_tgt * REC + (e & 7)] = s_cr[e];
        }
      }
      if (lane < MG && live) {
        if (in_lds)
          s_slot_id[tgt] = s_cid[lane];
        else
          slot_id[tgt] = s_cid[lane];
      }
      if (lane == 0) {
        s_state[0] = ns;
        s_state[1] = c0 + done;
        s_state[2] = spill;
      }
    }
    if (!in_lds) __threadfence_block();
    __syncthreads();
    nslots = s_state[0];
    c0 = s_state[1];
    if (s_state[2]) {
      for (int i = tid; i < nslots * REC; i += T_VIEW) slot[i] = lds_slot[i];
      for (int i = tid; i < nslots; i += T_VIEW) slot_id[i] = s_slot_id[i];
      in_lds = false;
      __threadfence_block();
    }
    __syncthreads();  // s_state / s_cr are rewritten by the next round
#ifdef GL_VIEW_PROF
    const long long t_r3 = clock64();
    pr_screen += t_r1 - t_r0;
    pr_exact += t_r2 - t_r1;
    pr_resolve += t_r3 - t_r2;
    pr_rounds++;
    pr_near += ncnt;
#endif
  }
  const int V = nslots;
#ifdef GL_VIEW_PROF
  const long long tp2 = clock64();
#endif

  // ---- phase 3: stable sort by depth descending --------------------------------------------
  for (int j = tid; j < V; j += T_VIEW) {
    double rj[REC];
    slot_rec(j, rj);
    const double dj = rj[7];
    int rank = 0;
    if (in_lds) {
      for (int i = 0; i < V; ++i) {
        const double di = lds_slot[i * REC + 7];
        rank += (di > dj || (di == dj && i < j)) ? 1 : 0;
      }
    } else {
      for (int i = 0; i < V; ++i) {
        const double di = slot[(size_t)i * REC + 7];
        rank += (di > dj || (di == dj && i < j)) ? 1 : 0;
      }
    }
    double inv[4];
    inv2(&rj[2], inv);
    sorted[(size_t)rank * REC + 0] = rj[0];
    sorted[(size_t)rank * REC + 1] = rj[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) sorted[(size_t)rank * REC + 2 + i] = inv[i];
    sorted_id[rank] = in_lds ? s_slot_id[j] : slot_id[j];
  }
  __threadfence_block();
  __syncthreads();
  if (view_ids_out) {
    for (int j = tid; j < view_cap; j += T_VIEW) view_ids_out[(size_t)f * view_cap + j] = j < V ? sorted_id[j] : -1;
  }
  if (nview_out && tid == 0) nview_out[f] = V;

#ifdef GL_VIEW_PROF
  const long long tp3 = clock64();
#endif
  // ---- phase 4: searchCorrespondence ----------------------------------------------------------
  const int nf = nfeat_all ? min(nfeat_all[f], N) : N;
  double2* s_mean = reinterpret_cast<double2*>(lds_slot);  // the slot list is dead by now
  if (knn == 5)
    search_correspondence<5, T_VIEW>(5, s_mean, sorted, sorted_id, uv_all + (size_t)f * N * 2, nf, N, V, cand_out + (size_t)f * N * 5,
                             ncand_out + (size_t)f * N, GL_VIEW_PROF_ARG);
  else
    search_correspondence<0, T_VIEW>(knn, s_mean, sorted, sorted_id, uv_all + (size_t)f * N * 2, nf, N, V,
                             cand_out + (size_t)f * N * knn, ncand_out + (size_t)f * N);
#ifdef GL_VIEW_PROF
  __syncthreads();
  if (tid == 0 && nview_out) {  // debug build: cycles of phases 1..4, C and V instead of the view list
    const long long tp4 = clock64();
    view_ids_out[(size_t)f * view_cap + 0] = (int)((tp1 - tp0) >> 4);
    view_ids_out[(size_t)f * view_cap + 1] = (int)((tp2 - tp1) >> 4);
    view_ids_out[(size_t)f * view_cap + 2] = (int)((tp3 - tp2) >> 4);
    view_ids_out[(size_t)f * view_cap + 3] = (int)((tp4 - tp3) >> 4);
    view_ids_out[(size_t)f * view_cap + 4] = C;
    view_ids_out[(size_t)f * view_cap + 5] = V;
    view_ids_out[(size_t)f * view_cap + 6] = (int)(pr_screen >> 4);
    view_ids_out[(size_t)f * view_cap + 7] = (int)(pr_exact >> 4);
    view_ids_out[(size_t)f * view_cap + 8] = (int)(pr_resolve >> 4);
    view_ids_out[(size_t)f * view_cap + 9] = pr_rounds;
    view_ids_out[(size_t)f * view_cap + 10] = pr_near;
    view_ids_out[(size_t)f * view_cap + 17] = (int)pr_q0;
  }
#endif
}

}  // namespace

extern "C" int gl_search2d(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, int B, const double* pose_dev,
                           int N, const double* uv_dev, const int32_t* nfeat_dev, int k, int32_t* cand_dev,
                           int32_t* ncand_dev, int view_cap, int32_t* view_ids_dev, int32_t* nview_dev) {
  GL_REQUIRE(ctx && gmm && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && N >= 0 && k >= 1 && k <= 8, "bad B / N / k");
  GL_REQUIRE(pose_dev && (N == 0 || (uv_dev && cand_dev && ncand_dev)), "null buffer");
  GL_REQUIRE(!view_ids_dev || view_cap > 0, "view_cap must be positive with view_ids_dev");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  void* scratch = nullptr;
  const size_t per_view = (size_t)g->K * (3 * REC + 2) * sizeof(double);
  int rc = gl::ctx_scratch(c, per_view * B + 64, &scratch);
  if (rc != GL_OK) return rc;
  ViewK vk{cam->fx, cam->fy, cam->cx, cam->cy, cam->width, cam->height};
  int slot_lds = SLOT_LDS;
  if (c->opt.view_slot_lds > 0) slot_lds = std::max(1, std::min(SLOT_LDS, (int)c->opt.view_slot_lds));  // tests: the spill path
  // latency shape up to one view per CU, throughput shape above (option view_threads forces one)
  int threads = B <= c->ncu ? 1024 : 256;
  if (c->opt.view_threads > 0) threads = (int)c->opt.view_threads == 1024 ? 1024 : 256;
  const size_t lds = (size_t)SLOT_LDS * REC * sizeof(double);
  if (threads == 1024)
    k_search2d<1024><<<B, 1024, lds, c->stream>>>(vk, B, g->K, slot_lds, g->rec12, g->cov, g->axis, g->flags, pose_dev, N,
                                                  uv_dev, nfeat_dev, k, cand_dev, ncand_dev, view_cap, view_ids_dev,
                                                  nview_dev, (double*)scratch);
  else
    k_search2d<256><<<B, 256, lds, c->stream>>>(vk, B, g->K, slot_lds, g->rec12, g->cov, g->axis, g->flags, pose_dev, N,
                                                uv_dev, nfeat_dev, k, cand_dev, ncand_dev, view_cap, view_ids_dev, nview_dev,
                                                (double*)scratch);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
