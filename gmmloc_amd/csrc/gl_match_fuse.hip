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

__global__ __launch_bounds__(T_F) void k_fuse_search(FuseP P, int B, const double* __restrict__ feat_uv_all, const float* __restrict__ feat_ur_all,
                                                     const int32_t* __restrict__ feat_oct_all, const uint8_t* __restrict__ feat_desc_all,
                                                     const double* __restrict__ mp_uvr_all, const int32_t* __restrict__ mp_level_all,
                                                     const uint8_t* __restrict__ mp_valid_all, const uint8_t* __restrict__ mp_desc_all,
                                                     int32_t* __restrict__ best_idx_all, int32_t* __restrict__ best_dist_all) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  int32_t* cell_ptr = lds;                 // NCELL + 1
  int32_t* cursor = cell_ptr + NCELL + 1;  // NCELL (grid build only)
  int32_t* cell_idx = cursor + NCELL;      // NF
  __shared__ int s_scan[T_F / 64];
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
  __syncthreads();
  auto cell_of = [&](int i) -> int {
    if (feat_oct[i] < 0) return -1;  // padding slot
    const double px = round((feat_uv[2 * i] - 0.0f) * P.col_inv), py = round((feat_uv[2 * i + 1] - 0.0f) * P.row_inv);
    if (!(px >= 0 && px < GC && py >= 0 && py < GR)) return -1;  // also rejects NaN
    return (int)px * GR + (int)py;
  };
  for (int i = tid; i < NF; i += T_F) {
    const int c = cell_of(i);
    if (c >= 0) atomicAdd(&cell_ptr[c + 1], 1);
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
  const size_t lds = ((size_t)2 * NCELL + 1 + NF) * sizeof(int32_t);
  GL_REQUIRE_LDS(c, lds);
  GL_HIP(gl::ensure_dynamic_lds(c, (const void*)k_fuse_search, lds));
  k_fuse_search<<<B, T_F, lds, c->stream>>>(P, B, feat_uv_dev, feat_ur_dev, feat_oct_dev, feat_desc_dev, mp_uvr_dev, mp_level_dev, mp_valid_dev,
                                            mp_desc_dev, best_idx_dev, best_dist_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
