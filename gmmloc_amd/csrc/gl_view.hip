// GMMLoc::associateMapElements (gmmloc_opt.cpp:115-135) for B key-frames:
//   GMM::renderView            (gaussian_mixture.cpp:271-371)   A3 + A4
//   GMM::searchCorrespondence  (gaussian_mixture.cpp:484-534)   A5
// One workgroup per key-frame.
//   phase 1  all K components in parallel: view-cosine cull (degenerate planes), projection
//            J R Sigma R^T J^T (GMMUtility::projectGaussian, gmm_utils.cpp:121-146 with
//            PinholeCamera::project3, pinhole_camera.cpp:68-150), 2-D eigenvalue cull; ordered
//            compaction (ballot / popcount prefix) keeps the reference's k = 0..K-1 order.
//   phase 2  the order-dependent occlusion merge: candidates are visited in order; the
//            argmin Bhattacharyya distance over the list accepted SO FAR is evaluated by
//            the whole workgroup (one slot per thread, strided) and reduced with a
//            (distance, index) argmin that prefers the lower index on ties, exactly like
//            the sequential `dist < min_dist` scan; replace-in-place / append as :341-355.
//   phase 3  stable rank sort by depth, descending (:362-364).
//   phase 4  per feature exact 5-NN on the 2-D means (nanoflann's result order: ascending,
//            ties by lower index) + MDist2 < 9 gate, in kNN order.
// Compiled with -ffp-contract=off: cull / merge decisions follow the fp64 CPU order.
#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {

constexpr int T_VIEW = 256;
constexpr int NW_VIEW = T_VIEW / 64;
constexpr int SLOT_LDS = 640;  // accepted 2-D components kept in LDS (40 KB); V is 100-300 on the EuRoC maps
constexpr int REC = 8;  // m0 m1 c00 c01 c10 c11 det depth

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

__global__ __launch_bounds__(T_VIEW) void k_search2d(ViewK vk, int B, int K, const double* __restrict__ rec12,
                                                     const double* __restrict__ cov3, const double* __restrict__ axis,
                                                     const uint8_t* __restrict__ flags,
                                                     const double* __restrict__ pose_all, int N,
                                                     const double* __restrict__ uv_all,
                                                     const int32_t* __restrict__ nfeat_all, int knn,
                                                     int32_t* __restrict__ cand_out, int32_t* __restrict__ ncand_out,
                                                     int view_cap, int32_t* __restrict__ view_ids_out,
                                                     int32_t* __restrict__ nview_out, double* __restrict__ scratch) {
  __shared__ int s_wcount[NW_VIEW];
  __shared__ double s_rd[NW_VIEW];
  __shared__ int s_ri[NW_VIEW];
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
  // ---- phase 2: sequential occlusion merge (:328-355) -------------------------------------
  // The accepted list lives in LDS (spilled to the global scratch only beyond SLOT_LDS entries), the
  // candidate record is fetched one iteration ahead by every thread (same address: one transaction), and
  // a candidate costs two barriers: argmin exchange, and publication of the slot it writes.
  extern __shared__ __attribute__((aligned(16))) double lds_slot[];  // SLOT_LDS x REC
  double* slotp = lds_slot;
  int nslots = 0;
  double cn[REC];
  int idn = 0;
  if (C > 0) {
#pragma unroll
    for (int i = 0; i < REC; ++i) cn[i] = cand[i];
    idn = cand_id[0];
  }
  for (int c = 0; c < C; ++c) {
    double cr[REC];
#pragma unroll
    for (int i = 0; i < REC; ++i) cr[i] = cn[i];
    const int idc = idn;
    if (c + 1 < C) {
#pragma unroll
      for (int i = 0; i < REC; ++i) cn[i] = cand[(size_t)(c + 1) * REC + i];
      idn = cand_id[c + 1];
    }
    int action_slot;  // -2 discard, -1 append, >= 0 replace that slot
    if (nslots == 0) {
      action_slot = -1;
    } else {
      double best = 1.7976931348623157e308;
      int bj = 0x7fffffff;
      for (int j = tid; j < nslots; j += T_VIEW) {
        double sr[REC];
#pragma unroll
        for (int i = 0; i < REC; ++i) sr[i] = slotp[(size_t)j * REC + i];
        const double d = bh2(sr, cr);
        if (d < best) {
          best = d;
          bj = j;
        }
      }
      // (dist, index) argmin, lower index wins ties
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double od = shfl_xor_f64(best, o);
        const int oj = __shfl_xor(bj, o, 64);
        if (od < best || (od == best && oj < bj)) {
          best = od;
          bj = oj;
        }
      }
      if (lane == 0) {
        s_rd[wave] = best;
        s_ri[wave] = bj;
      }
      __syncthreads();  // also: every read of the slot list is done before it changes
      best = s_rd[0];
      bj = s_ri[0];
#pragma unroll
      for (int w = 1; w < NW_VIEW; ++w)
        if (s_rd[w] < best || (s_rd[w] == best && s_ri[w] < bj)) {
          best = s_rd[w];
          bj = s_ri[w];
        }
      if (bj == 0x7fffffff) bj = 0;  // every distance NaN: min_idx stays at its initial value
      if (best < 0.8) {
        action_slot = (cr[7] < slotp[(size_t)bj * REC + 7]) ? bj : -2;
      } else {
        action_slot = -1;
      }
      if (action_slot == -2) __syncthreads();  // s_rd / s_ri are rewritten by the next candidate
    }
    if (action_slot != -2) {
      if (nslots > 0) __syncthreads();  // the depth read above precedes the overwrite
      const int dst = action_slot == -1 ? nslots : action_slot;
      if (tid < REC) slotp[(size_t)dst * REC + tid] = cr[tid];
      if (tid == REC) slot_id[dst] = idc;
      if (action_slot == -1) {
        ++nslots;
        if (nslots == SLOT_LDS && slotp == lds_slot) {  // rare: continue in the global scratch
          __syncthreads();
          for (int i = tid; i < SLOT_LDS * REC; i += T_VIEW) slot[i] = lds_slot[i];
          slotp = slot;
          __threadfence_block();
        }
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  slot = slotp;
  const int V = nslots;
#ifdef GL_VIEW_PROF
  const long long tp2 = clock64();
#endif

  // ---- phase 3: stable sort by depth descending --------------------------------------------
  for (int j = tid; j < V; j += T_VIEW) {
    const double dj = slot[(size_t)j * REC + 7];
    int rank = 0;
    for (int i = 0; i < V; ++i) {
      const double di = slot[(size_t)i * REC + 7];
      rank += (di > dj || (di == dj && i < j)) ? 1 : 0;
    }
    double inv[4];
    inv2(&slot[(size_t)j * REC + 2], inv);
    sorted[(size_t)rank * REC + 0] = slot[(size_t)j * REC + 0];
    sorted[(size_t)rank * REC + 1] = slot[(size_t)j * REC + 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) sorted[(size_t)rank * REC + 2 + i] = inv[i];
    sorted_id[rank] = slot_id[j];
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
  // The kNN loop reads the mean of every rendered component for every feature: the means are staged in LDS
  // (the slot list is dead by now) in chunks of MCH - one chunk, staged once, for any realistic view - and an
  // entry that is not nearer than the current knn-th neighbour skips the insertion network.
  double2* s_mean = reinterpret_cast<double2*>(lds_slot);
  constexpr int MCH = SLOT_LDS * REC / 2;
  bool staged = false;
  for (int n0 = 0; n0 < N; n0 += T_VIEW) {  // uniform trip count: the staging barriers sit inside
    const int n = n0 + tid;
    const bool live = n < nf && V > 0;
    double dist[8], worst = __builtin_inf(), fu = 0.0, fv = 0.0;
    int idx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dist[i] = __builtin_inf();
      idx[i] = -1;
    }
    if (live) {
      fu = uv_all[((size_t)f * N + n) * 2];
      fv = uv_all[((size_t)f * N + n) * 2 + 1];
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
      for (int j = 0; j < jn; ++j) {
        const double2 mj = s_mean[j];
        const double d0 = fu - mj.x, d1 = fv - mj.y;
        double cd = d0 * d0 + d1 * d1;  // kdtree_distance (gaussian_mixture.h:71-76)
        if (!(cd < worst)) continue;
        int ci = j0 + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < knn) {
            const bool sw = cd < dist[i];
            const double td = sw ? dist[i] : cd;
            const int ti = sw ? idx[i] : ci;
            dist[i] = sw ? cd : dist[i];
            idx[i] = sw ? ci : idx[i];
            cd = td;
            ci = ti;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) worst = (i == knn - 1) ? dist[i] : worst;
      }
    }
    if (n >= N) continue;
    int32_t* co = cand_out + ((size_t)f * N + n) * knn;
    int m = 0;
    if (live) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < knn && idx[i] >= 0) {
          const double* sr = &sorted[(size_t)idx[i] * REC];
          if (mdist2_2d(sr, sr + 2, fu, fv) < 9.0) co[m++] = sorted_id[idx[i]];  // check_mdist2 (:521-527)
        }
      }
    }
    ncand_out[(size_t)f * N + n] = m;
    for (; m < knn; ++m) co[m] = -1;
  }
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
  k_search2d<<<B, T_VIEW, (size_t)SLOT_LDS * REC * sizeof(double), c->stream>>>(vk, B, g->K, g->rec12, g->cov, g->axis, g->flags, pose_dev, N, uv_dev,
                                          nfeat_dev, k, cand_dev, ncand_dev, view_cap, view_ids_dev, nview_dev,
                                          (double*)scratch);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
