// Localization::fuseObservations (localization.cpp:226-318), the matching half, for B key-frames: every map point of the
// neighbourhood that the key-frame does not observe yet is projected into it (host: project3 + checkScaleAndVisible, the same
// ProjStat the host builds for searchByProjection) and looks for its most similar feature inside a window - Frame::getFeaturesInArea
// (frame.cpp:121-177) on the 64 x 48 bucket grid of Frame::assignFeaturesToGrid (frame.cpp:54-79), the pyramid level within one of
// the predicted one, the chi2 of the pixel (+ disparity) error against 5.99 / 7.8, 256-bit Hamming distance, TH_LOW.  The map points
// do NOT interact here (nothing is marked as taken inside the loop: what the reference does with a match - add the observation, or
// replace one of the two map points by the other - is graph work of the host, in list order), so this is one independent window
// walk per map point; the visiting order of getFeaturesInArea (cell column, cell row, feature index) decides ties (`dist <
// best_dist`: the first of equal distances) and is kept.  Every float / double conversion of the reference is kept (the file is
// compiled without contraction).
#include <climits>

#include "gl_internal.hpp"

namespace {

constexpr int GC = 64, GR = 48, NCELL = GC * GR;  // frame::grid_cols / grid_rows (config.h:57)
constexpr int T_F = 512;

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) d += __popc(a[w] ^ b[w]);
  return d;
}

struct FuseP {
  int NF, NP;
  float col_inv, row_inv, th;
  float sf[8], sigma2_inv[8];
};

// REC: the frame's features also as 16-byte records in LDS, in CSR order (the launcher sets it when they fit next to the grid and the
// queries' order), and the walk below waits for the LDS instead of for three dependent global loads per visited entry.
template <bool REC>
__global__ __launch_bounds__(T_F) void k_fuse_search(FuseP P, int B, const double* __restrict__ feat_uv_all, const float* __restrict__ feat_ur_all,
                                                     const int32_t* __restrict__ feat_oct_all, const uint8_t* __restrict__ feat_desc_all,
                                                     const double* __restrict__ mp_uvr_all, const int32_t* __restrict__ mp_level_all,
                                                     const uint8_t* __restrict__ mp_valid_all, const uint8_t* __restrict__ mp_desc_all,
                                                     int32_t* __restrict__ best_idx_all, int32_t* __restrict__ best_dist_all) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  int32_t* cell_ptr = lds;                 // NCELL + 1
  int32_t* cursor = cell_ptr + NCELL + 1;  // NCELL (grid build only)
  int32_t* cell_idx = cursor + NCELL;      // NF
  float4* rec16 = (float4*)(lds + ((2 * NCELL + 1 + P.NF + 3) & ~3));  // REC: {u, v, u_right, octave | feature << 8} per CSR entry
  uint16_t* qorder = (uint16_t*)(rec16 + P.NF);                         // REC: the queries sorted by window class
  __shared__ int s_scan[T_F / 64], s_fast, s_cls[9];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= B) return;
  const int NF = P.NF, NP = P.NP;
  const double* feat_uv = feat_uv_all + (size_t)f * NF * 2;
  const float* feat_ur = feat_ur_all + (size_t)f * NF;
  const int32_t* feat_oct = feat_oct_all + (size_t)f * NF;
  const uint32_t* feat_desc = (const uint32_t*)(feat_desc_all + (size_t)f * NF * 32);
  const double* mp_uvr = mp_uvr_all + (size_t)f * NP * 3;
  const int32_t* mp_level = mp_level_all + (size_t)f * NP;
  const uint8_t* mp_valid = mp_valid_all + (size_t)f * NP;
  const uint32_t* mp_desc = (const uint32_t*)(mp_desc_all + (size_t)f * NP * 32);

  // ---- assignFeaturesToGrid: CSR by cell (ix * GR + iy), ascending feature index inside a cell (as in gl_match.hip) ----
  for (int c = tid; c <= NCELL; c += T_F) cell_ptr[c] = 0;
  if (tid == 0) s_fast = 1;
  if (tid < 9) s_cls[tid] = 0;
  __syncthreads();
  auto cell_of = [&](int i) -> int {
    if (feat_oct[i] < 0) return -1;  // padding slot
    const double px = round((feat_uv[2 * i] - 0.0f) * P.col_inv), py = round((feat_uv[2 * i + 1] - 0.0f) * P.row_inv);
    if (!(px >= 0 && px < GC && py >= 0 && py < GR)) return -1;  // also rejects NaN
    return (int)px * GR + (int)py;
  };
  {
    // the record walk is exact iff every feature coordinate is a float value (the reference's (float)(double u - (double)x) is then the
    // correctly rounded float difference, and (double)(float)u is u in its double expressions) and the octaves fit the record
    bool fok = true;
    for (int i = tid; i < NF; i += T_F) {
      const int c = cell_of(i);
      if (c >= 0) {
        atomicAdd(&cell_ptr[c + 1], 1);
        const double u = feat_uv[2 * i], v = feat_uv[2 * i + 1];
        fok = fok && (double)(float)u == u && (double)(float)v == v && feat_oct[i] <= 255;
      }
    }
    if (!fok) s_fast = 0;
  }
  __syncthreads();
  {
    constexpr int CH = (NCELL + T_F - 1) / T_F;
    const int c0 = tid * CH, c1 = min(NCELL, c0 + CH);
    int s = 0;
    for (int c = c0; c < c1; ++c) s += cell_ptr[c + 1];
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
    for (int c = c0; c < c1; ++c) {
      const int v = cell_ptr[c + 1];
      cell_ptr[c + 1] = run + v;
      run += v;
    }
    __syncthreads();
  }
  for (int c = tid; c < NCELL; c += T_F) cursor[c] = 0;
  __syncthreads();
  for (int i = tid; i < NF; i += T_F) {
    const int c = cell_of(i);
    if (c >= 0) cell_idx[cell_ptr[c] + atomicAdd(&cursor[c], 1)] = i;
  }
  __syncthreads();
  for (int c = tid; c < NCELL; c += T_F) {
    const int e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
    for (int e = e0 + 1; e < e1; ++e) {  // insertion sort: the push_back order of the reference
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

  if (REC && s_fast) {
    // ---- round 6: the walk of gl_match.hip's first round (no owners here: the map points do not interact) -------------------------
    for (int e = tid; e < cell_ptr[NCELL]; e += T_F) {
      const int i = cell_idx[e];
      rec16[e] = make_float4((float)feat_uv[2 * i], (float)feat_uv[2 * i + 1], feat_ur[i], __int_as_float((feat_oct[i] & 0xff) | (i << 8)));
    }
    // queries by window class (the half size of the window is a factor x the scale of the predicted level), the LARGEST first, so that
    // the lanes of a wave walk windows of like size (the order only decides which thread takes which query)
    auto cls_of = [&](int m) -> int { return mp_valid[m] ? (mp_level[m] & 7) : 8; };
    for (int m = tid; m < NP; m += T_F) atomicAdd(&s_cls[cls_of(m)], 1);
    __syncthreads();
    if (tid == 0) {
      const int ord[9] = {7, 6, 5, 4, 3, 2, 1, 0, 8};
      int run = 0;
      for (int k = 0; k < 9; ++k) {
        const int n = s_cls[ord[k]];
        s_cls[ord[k]] = run;
        run += n;
      }
    }
    __syncthreads();
    for (int m = tid; m < NP; m += T_F) qorder[atomicAdd(&s_cls[cls_of(m)], 1)] = (uint16_t)m;
    __syncthreads();
    uint16_t* lst = (uint16_t*)cursor;  // 4 x T_F entry indices: a thread's collected candidates (the cursors are dead after the grid build)
    const int nq_rounds = ((NP + T_F - 1) / T_F) * T_F;
    for (int sq = tid; sq < nq_rounds; sq += T_F) {
      const bool in = sq < NP;
      const int m = in ? (int)qorder[sq] : 0;
      const bool act = in && mp_valid[m] != 0;
      int best_dist = 256, best_idx = -1;
      int lvl_pred = 0;
      double ux = 0.0, uy = 0.0, ur = 0.0;
      float x = 0.f, y = 0.f, rr = 0.f;
      int x0 = 1, x1 = 0, y0 = 0, y1 = 0;
      uint32_t dm[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) dm[w] = act ? mp_desc[(size_t)m * 8 + w] : 0u;
      if (act) {
        lvl_pred = mp_level[m];
        ux = mp_uvr[3 * m];
        uy = mp_uvr[3 * m + 1];
        ur = mp_uvr[3 * m + 2];
        rr = P.th * P.sf[lvl_pred & 7];
        x = (float)ux;
        y = (float)uy;
        x0 = max(0, (int)floorf((x - 0.0f - rr) * P.col_inv));
        x1 = min(GC - 1, (int)ceilf((x - 0.0f + rr) * P.col_inv));
        y0 = max(0, (int)floorf((y - 0.0f - rr) * P.row_inv));
        y1 = min(GR - 1, (int)ceilf((y - 0.0f + rr) * P.row_inv));
        if (!(x0 < GC && x1 >= 0 && y0 < GR && y1 >= 0) || y0 > y1) x1 = x0 - 1;  // nothing to visit
      }
      int cnt = 0;
      auto flush = [&]() {  // Hamming distances of the <= 4 collected candidates (in visiting order), their descriptors requested together
        uint4 da[4], db[4];
        int fi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fi[j] = 0;
          da[j] = db[j] = make_uint4(0, 0, 0, 0);
          if (j < cnt) {
            fi[j] = __float_as_int(rec16[lst[j * T_F + tid]].w) >> 8;
            const uint4* src = (const uint4*)(feat_desc + (size_t)fi[j] * 8);
            da[j] = src[0];
            db[j] = src[1];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < cnt) {
            const int dist = __popc(dm[0] ^ da[j].x) + __popc(dm[1] ^ da[j].y) + __popc(dm[2] ^ da[j].z) + __popc(dm[3] ^ da[j].w) +
                             __popc(dm[4] ^ db[j].x) + __popc(dm[5] ^ db[j].y) + __popc(dm[6] ^ db[j].z) + __popc(dm[7] ^ db[j].w);
            if (dist < best_dist) {
              best_dist = dist;
              best_idx = fi[j];
            }
          }
        }
        cnt = 0;
      };
      auto entry = [&](const float4 r, int ee) {
        const int pk = __float_as_int(r.w), kpLevel = pk & 0xff;
        bool ok = fabsf(r.x - x) < rr && fabsf(r.y - y) < rr && !(kpLevel < lvl_pred - 1 || kpLevel > lvl_pred);
        if (ok) {
          const double dx = (double)r.x - ux, dy = (double)r.y - uy;
          double err;  // Feature::error(Vector3d): squared norm of (uv - obs) or of (uvr - obs)
          if (r.z < 0.0f) {
            err = dx * dx + dy * dy;
          } else {
            const double dz = (double)r.z - ur;
            err = dx * dx + dy * dy + dz * dz;
          }
          err *= P.sigma2_inv[kpLevel & 7];
          const double thresh = r.z >= 0 ? 7.8 : 5.99;
          if (err > thresh) ok = false;
        }
        if (ok) {
          lst[cnt * T_F + tid] = (uint16_t)ee;
          ++cnt;
        }
      };
      bool more = act && x0 <= x1;
      int ix = x0 - 1, e = 0, e1 = 0, ne = 0, ne1 = 0;  // (ne, ne1): the range of column ix + 1, requested a column ahead
      if (more) {
        ne = cell_ptr[x0 * GR + y0];
        ne1 = cell_ptr[x0 * GR + y1 + 1];
      }
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
      }
      if (__any(cnt > 0)) flush();
      if (in) {
        best_idx_all[(size_t)f * NP + m] = best_dist <= 50 ? best_idx : -1;  // TH_LOW
        best_dist_all[(size_t)f * NP + m] = best_dist;
      }
    }
    return;
  }
  // ---- one window walk per map point (:250-291) ------------------------------------------------------------------------
  for (int m = tid; m < NP; m += T_F) {
    int best_dist = 256, best_idx = -1;
    if (mp_valid[m]) {
      const int lvl_pred = mp_level[m];
      const double ux = mp_uvr[3 * m], uy = mp_uvr[3 * m + 1], ur = mp_uvr[3 * m + 2];
      const float radius = P.th * P.sf[lvl_pred & 7];
      const float x = (float)ux, y = (float)uy, rr = radius;  // getFeaturesInArea takes const float&
      const int x0 = max(0, (int)floorf((x - 0.0f - rr) * P.col_inv));
      const int x1 = min(GC - 1, (int)ceilf((x - 0.0f + rr) * P.col_inv));
      const int y0 = max(0, (int)floorf((y - 0.0f - rr) * P.row_inv));
      const int y1 = min(GR - 1, (int)ceilf((y - 0.0f + rr) * P.row_inv));
      if (x0 < GC && x1 >= 0 && y0 < GR && y1 >= 0) {
        uint32_t dm[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) dm[w] = mp_desc[(size_t)m * 8 + w];
        for (int ix = x0; ix <= x1; ++ix) {
          const int e0 = cell_ptr[ix * GR + y0], e1 = cell_ptr[ix * GR + y1 + 1];  // cells (ix, y0..y1) are contiguous
          for (int e = e0; e < e1; ++e) {
            const int idx = cell_idx[e];
            const double fu = feat_uv[2 * idx], fv = feat_uv[2 * idx + 1];
            const float distx = (float)(fu - (double)x), disty = (float)(fv - (double)y);
            if (!(fabsf(distx) < rr && fabsf(disty) < rr)) continue;
            const int kpLevel = feat_oct[idx];
            if (kpLevel < lvl_pred - 1 || kpLevel > lvl_pred) continue;
            const float kur = feat_ur[idx];
            const double dx = fu - ux, dy = fv - uy;
            double err;  // Feature::error(Vector3d): squared norm of (uv - obs) or of (uvr - obs)
            if (kur < 0.0f) {
              err = dx * dx + dy * dy;
            } else {
              const double dz = (double)kur - ur;
              err = dx * dx + dy * dy + dz * dz;
            }
            err *= P.sigma2_inv[kpLevel & 7];
            const double thresh = kur >= 0 ? 7.8 : 5.99;
            if (err > thresh) continue;
            const int dist = hamming256(dm, feat_desc + (size_t)idx * 8);
            if (dist < best_dist) {
              best_dist = dist;
              best_idx = idx;
            }
          }
        }
      }
    }
    best_idx_all[(size_t)f * NP + m] = best_dist <= 50 ? best_idx : -1;  // TH_LOW
    best_dist_all[(size_t)f * NP + m] = best_dist;
  }
}

}  // namespace

extern "C" int gl_fuse_search(gl_ctx_t* ctx, const gl_camera* cam, float scale_factor, int B, int NF, int NP, const double* feat_uv_dev,
                              const float* feat_ur_dev, const int32_t* feat_oct_dev, const uint8_t* feat_desc_dev, const double* mp_uvr_dev,
                              const int32_t* mp_level_dev, const uint8_t* mp_valid_dev, const uint8_t* mp_desc_dev, float th,
                              int32_t* best_idx_dev, int32_t* best_dist_dev) {
  GL_REQUIRE(ctx && cam, "null argument");
  if (B == 0) return GL_OK;
  GL_REQUIRE(B > 0 && NF >= 1 && NP >= 1, "bad B / NF / NP");
  GL_REQUIRE(NF <= 16384, "NF above the on-chip capacity (16384 features per key-frame)");
  GL_REQUIRE(cam->width > 0 && cam->height > 0, "camera without image size");
  GL_REQUIRE(feat_uv_dev && feat_ur_dev && feat_oct_dev && feat_desc_dev && mp_uvr_dev && mp_level_dev && mp_valid_dev && mp_desc_dev &&
                 best_idx_dev && best_dist_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  FuseP P;
  P.NF = NF;
  P.NP = NP;
  P.col_inv = static_cast<float>(GC) / cam->width;  // frame.cpp:33-34
  P.row_inv = static_cast<float>(GR) / cam->height;
  P.th = th;
  P.sf[0] = 1.0f;  // init_config.hpp:63-79
  P.sigma2_inv[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    P.sf[i] = P.sf[i - 1] * scale_factor;
    const float s2 = P.sf[i] * P.sf[i];
    P.sigma2_inv[i] = 1.0f / s2;
  }
  size_t lds = ((size_t)2 * NCELL + 1 + NF) * sizeof(int32_t);
  // the record walk: 16 bytes per feature and 2 per map point more; two key-frames per CU must still fit (NP <= 65 535: 16-bit query order)
  const size_t lds_rec = (((size_t)2 * NCELL + 1 + NF + 3) & ~(size_t)3) * sizeof(int32_t) + (size_t)NF * 16 + (size_t)NP * 2;
  const bool rec = NP <= 65535 && lds_rec <= 80 * 1024 && c->opt.fuse_records != 0;
  if (rec) lds = lds_rec;
  GL_REQUIRE_LDS(c, lds);
  auto kern = rec ? k_fuse_search<true> : k_fuse_search<false>;
  GL_HIP(gl::ensure_dynamic_lds(c, (const void*)kern, lds));
  kern<<<B, T_F, lds, c->stream>>>(P, B, feat_uv_dev, feat_ur_dev, feat_oct_dev, feat_desc_dev, mp_uvr_dev, mp_level_dev, mp_valid_dev,
                                            mp_desc_dev, best_idx_dev, best_dist_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
