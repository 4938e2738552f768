// GMM map construction on the GPU.
//   k_build_components : GaussianComponent ctor + decompose
//                        (gaussian.h:30-39, gaussian.cpp:36-63)           -- A0
//   k_nbs_count/_fill  : GMM::GMM neighbour graph, Bhattacharyya distance
//                        (gaussian_mixture.cpp:61-78, gmm_utils.h:30-52)  -- A2
// Compiled with -ffp-contract=off: cov_inv must be bit-identical to the fp64
// CPU evaluation order (it feeds the bit-exact association indices).
#include <algorithm>
#include <cstring>

#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {

__global__ void k_build_components(int K, const double* __restrict__ mean_in, const double* __restrict__ cov_in,
                                   double* __restrict__ rec12, double* __restrict__ det, double* __restrict__ scale,
                                   double* __restrict__ axis, double* __restrict__ sqrt_info,
                                   double* __restrict__ hgw, double* __restrict__ plane4, uint8_t* __restrict__ flags) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double cov[9], inv[9], w[3], V[9], L[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) cov[i] = cov_in[(size_t)k * 9 + i];
  inv3(cov, inv);
  double* rec = rec12 + (size_t)k * 12;
  rec[0] = mean_in[(size_t)k * 3 + 0];
  rec[1] = mean_in[(size_t)k * 3 + 1];
  rec[2] = mean_in[(size_t)k * 3 + 2];
#pragma unroll
  for (int i = 0; i < 9; ++i) rec[3 + i] = inv[i];
  det[k] = det3(cov);
  eig_sym<3>(cov, w, V);
  const bool ok = chol3_lower(inv, L);
#pragma unroll
  for (int i = 0; i < 3; ++i) scale[(size_t)k * 3 + i] = w[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    axis[(size_t)k * 9 + i] = V[i];
    sqrt_info[(size_t)k * 9 + i] = ok ? L[i] : __builtin_nan("");
  }
  {  // J^T J of EdgePt2Gaussian (J = sqrt_info^T): L L^T, symmetric 00 01 02 11 12 22
    const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int i = ii[e], j = jj[e];
      hgw[(size_t)k * 6 + e] = ok ? (L[i * 3] * L[j * 3] + L[i * 3 + 1] * L[j * 3 + 1] + L[i * 3 + 2] * L[j * 3 + 2]) : __builtin_nan("");
    }
  }
  {  // EdgePt2GaussianDeg's plane (factors.cpp:55-64): normal = axis_.col(0), offset n . mean
    const double nx = V[0], ny = V[3], nz = V[6];
    double* pl = plane4 + (size_t)k * 4;
    pl[0] = nx;
    pl[1] = ny;
    pl[2] = nz;
    pl[3] = nx * rec[0] + ny * rec[1] + nz * rec[2];
  }
  const bool deg = w[0] < 1e-4;                     // gaussian.cpp:44
  const bool salient = (w[1] > 0.2 && w[2] > 0.2);  // gaussian.cpp:51-54
  flags[k] = (uint8_t)((deg ? 1 : 0) | (salient ? 2 : 0));
}

// GMMUtility::BHCoefficient<GaussianComponent> (gmm_utils.h:30-52)
GL_DEV double bh3(const double* c0, const double* m0, double det0, const double* c1, const double* m1, double det1) {
  double cov[9], inv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) cov[i] = (c0[i] + c1[i]) / 2.0;
  const double d[3] = {m1[0] - m0[0], m1[1] - m0[1], m1[2] - m0[2]};
  inv3(cov, inv);
  double r[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) r[j] = (d[0] * inv[0 * 3 + j] + d[1] * inv[1 * 3 + j]) + d[2] * inv[2 * 3 + j];
  double d0 = (r[0] * d[0] + r[1] * d[1]) + r[2] * d[2];
  d0 /= 8.0;
  const double d1 = log(det3(cov) / sqrt(det0 * det1)) / 2.0;
  return d0 + d1;
}

// One wave per row i; lanes sweep j in chunks of 64.  FILL = false: count only.
// Rows keep the reference's push order (ascending j).
template <bool FILL>
__global__ __launch_bounds__(256) void k_nbs(int K, double thresh, const double* __restrict__ rec12,
                                             const double* __restrict__ cov, const double* __restrict__ det,
                                             int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr,
                                             int32_t* __restrict__ nbs_idx, double* __restrict__ nbs_dist) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= K) return;
  double ci[9], mi[3];
#pragma unroll
  for (int a = 0; a < 9; ++a) ci[a] = cov[(size_t)i * 9 + a];
#pragma unroll
  for (int a = 0; a < 3; ++a) mi[a] = rec12[(size_t)i * 12 + a];
  const double deti = det[i];
  int base = FILL ? row_ptr[i] : 0;
  int cnt = 0;
  for (int j0 = 0; j0 < K; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    double dist = 0.0;
    if (j < K && j != i) {
      double cj[9], mj[3];
#pragma unroll
      for (int a = 0; a < 9; ++a) cj[a] = cov[(size_t)j * 9 + a];
#pragma unroll
      for (int a = 0; a < 3; ++a) mj[a] = rec12[(size_t)j * 12 + a];
      dist = bh3(ci, mi, deti, cj, mj, det[j]);
      hit = dist < thresh;
    }
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      const int pos = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
      nbs_idx[pos] = j;
      nbs_dist[pos] = dist;
    }
    cnt += __popcll(m);
  }
  if (!FILL && lane == 0) counts[i] = cnt;
}

}  // namespace

namespace gl {

int launch_build_components(Ctx* c, Gmm* g) {
  double *d_mean = nullptr, *d_cov = nullptr;
  const int K = g->K;
  GL_HIP(hipMalloc(&d_mean, sizeof(double) * 3 * K));
  GL_HIP(hipMemcpyAsync(d_mean, g->h_mean.data(), sizeof(double) * 3 * K, hipMemcpyHostToDevice, c->stream));
  GL_HIP(hipMemcpyAsync(g->cov, g->h_cov.data(), sizeof(double) * 9 * K, hipMemcpyHostToDevice, c->stream));
  GL_HIP(hipMemcpyAsync(g->mean, g->h_mean.data(), sizeof(double) * 3 * K, hipMemcpyHostToDevice, c->stream));
  (void)d_cov;
  k_build_components<<<(K + 127) / 128, 128, 0, c->stream>>>(K, d_mean, g->cov, g->rec12, g->det, g->scale, g->axis,
                                                             g->sqrt_info, g->hgw, g->plane4, g->flags);
  GL_HIP(hipGetLastError());
  GL_HIP(hipStreamSynchronize(c->stream));
  GL_HIP(hipFree(d_mean));
  return GL_OK;
}

int launch_build_neighbours(Ctx* c, Gmm* g) {
  const int K = g->K;
  int32_t* d_counts = nullptr;
  GL_HIP(hipMalloc(&d_counts, sizeof(int32_t) * (K + 1)));
  const int wpb = 4;
  const int grid = (K + wpb - 1) / wpb;
  k_nbs<false><<<grid, wpb * 64, 0, c->stream>>>(K, g->prm.neighbor_dist_thresh, g->rec12, g->cov, g->det, d_counts,
                                                 nullptr, nullptr, nullptr);
  GL_HIP(hipGetLastError());
  std::vector<int32_t> counts(K + 1, 0);
  GL_HIP(hipMemcpyAsync(counts.data(), d_counts, sizeof(int32_t) * K, hipMemcpyDeviceToHost, c->stream));
  GL_HIP(hipStreamSynchronize(c->stream));
  std::vector<int32_t> ptr(K + 1, 0);
  for (int i = 0; i < K; ++i) ptr[i + 1] = ptr[i] + counts[i];
  g->nnz = ptr[K];
  GL_HIP(hipMalloc(&g->nbs_ptr, sizeof(int32_t) * (K + 1)));
  GL_HIP(hipMalloc(&g->nbs_idx, sizeof(int32_t) * std::max(1, g->nnz)));
  GL_HIP(hipMalloc(&g->nbs_dist, sizeof(double) * std::max(1, g->nnz)));
  GL_HIP(hipMemcpyAsync(g->nbs_ptr, ptr.data(), sizeof(int32_t) * (K + 1), hipMemcpyHostToDevice, c->stream));
  k_nbs<true><<<grid, wpb * 64, 0, c->stream>>>(K, g->prm.neighbor_dist_thresh, g->rec12, g->cov, g->det, nullptr,
                                                g->nbs_ptr, g->nbs_idx, g->nbs_dist);
  GL_HIP(hipGetLastError());
  GL_HIP(hipStreamSynchronize(c->stream));
  GL_HIP(hipFree(d_counts));
  return GL_OK;
}

}  // namespace gl

extern "C" {

int gl_gmm_create(gl_ctx_t* ctx, const double* mean, const double* cov, int K, const gl_params* prm,
                  gl_gmm_t** out) {
  GL_REQUIRE(ctx && mean && cov && out, "null argument");
  GL_REQUIRE(K > 0, "K must be positive");  // gmm_utils.cpp:32-35 (empty file -> false)
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  gl::Gmm* g = new gl::Gmm();
  g->device = c->device;
  g->K = K;
  if (prm)
    g->prm = *prm;
  else
    gl_default_params(&g->prm);
  g->h_mean.assign(mean, mean + (size_t)3 * K);
  g->h_cov.assign(cov, cov + (size_t)9 * K);
  int rc = GL_OK;
  auto alloc = [&](void** p, size_t bytes) {
    if (rc == GL_OK && hipMalloc(p, bytes) != hipSuccess) {
      gl::set_error("gl_gmm_create: hipMalloc(%zu) failed", bytes);
      rc = GL_ERR_NOMEM;
    }
  };
  alloc((void**)&g->rec12, sizeof(double) * 12 * K);
  alloc((void**)&g->mean, sizeof(double) * 3 * K);
  alloc((void**)&g->cov, sizeof(double) * 9 * K);
  alloc((void**)&g->det, sizeof(double) * K);
  alloc((void**)&g->scale, sizeof(double) * 3 * K);
  alloc((void**)&g->axis, sizeof(double) * 9 * K);
  alloc((void**)&g->sqrt_info, sizeof(double) * 9 * K);
  alloc((void**)&g->hgw, sizeof(double) * 6 * K);
  alloc((void**)&g->plane4, sizeof(double) * 4 * K);
  alloc((void**)&g->flags, K);
  if (rc == GL_OK) rc = gl::launch_build_components(c, g);
  if (rc == GL_OK) rc = gl::launch_build_neighbours(c, g);
  if (rc == GL_OK) rc = gl::build_cell_index(c, g);
  if (rc != GL_OK) {
    gl_gmm_destroy((gl_gmm_t*)g);
    return rc;
  }
  *out = (gl_gmm_t*)g;
  return GL_OK;
}

int gl_gmm_load_file(gl_ctx_t* ctx, const char* path, const gl_params* prm, gl_gmm_t** out) {
  GL_REQUIRE(ctx && path && out, "null argument");
  std::vector<double> mean, cov;
  const int rc = gl::read_gmm_file(path, mean, cov);
  if (rc != GL_OK) return rc;
  return gl_gmm_create(ctx, mean.data(), cov.data(), (int)(mean.size() / 3), prm, out);
}

int gl_gmm_save_file(const gl_gmm_t* gmm, const char* path) {
  GL_REQUIRE(gmm && path, "null argument");
  gl::Gmm* g = gl::G(gmm);
  std::vector<uint8_t> flags(g->K);
  GL_HIP(hipSetDevice(g->device));
  GL_HIP(hipMemcpy(flags.data(), g->flags, g->K, hipMemcpyDeviceToHost));
  return gl::write_gmm_file(path, g->h_mean.data(), g->h_cov.data(), flags.data(), g->K);
}

int gl_gmm_destroy(gl_gmm_t* gmm) {
  if (!gmm) return GL_OK;
  gl::Gmm* g = gl::G(gmm);
  (void)hipSetDevice(g->device);
  void* ptrs[] = {g->rec12, g->mean, g->cov, g->det, g->scale, g->axis, g->sqrt_info, g->hgw, g->plane4, g->flags, g->nbs_ptr, g->nbs_idx, g->nbs_dist};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  gl::free_cell_index(g);
  delete g;
  return GL_OK;
}

int gl_gmm_count(const gl_gmm_t* gmm) { return gmm ? gl::G(gmm)->K : GL_ERR_ARG; }
int gl_gmm_nbs_count(const gl_gmm_t* gmm) { return gmm ? gl::G(gmm)->nnz : GL_ERR_ARG; }

int gl_gmm_get(const gl_gmm_t* gmm, int field, void* host_out, size_t bytes) {
  GL_REQUIRE(gmm && host_out, "null argument");
  gl::Gmm* g = gl::G(gmm);
  const size_t K = (size_t)g->K;
  const void* src = nullptr;
  size_t need = 0;
  bool host_src = false;
  switch (field) {
    case GL_F_MEAN: src = g->h_mean.data(); need = K * 24; host_src = true; break;
    case GL_F_COV: src = g->h_cov.data(); need = K * 72; host_src = true; break;
    case GL_F_DET: src = g->det; need = K * 8; break;
    case GL_F_SCALE: src = g->scale; need = K * 24; break;
    case GL_F_AXIS: src = g->axis; need = K * 72; break;
    case GL_F_SQRT_INFO: src = g->sqrt_info; need = K * 72; break;
    case GL_F_FLAGS: src = g->flags; need = K; break;
    case GL_F_NBS_PTR: src = g->nbs_ptr; need = (K + 1) * 4; break;
    case GL_F_NBS_IDX: src = g->nbs_idx; need = (size_t)g->nnz * 4; break;
    case GL_F_NBS_DIST: src = g->nbs_dist; need = (size_t)g->nnz * 8; break;
    case GL_F_COV_INV: need = K * 72; break;
    default: gl::set_error("gl_gmm_get: unknown field %d", field); return GL_ERR_ARG;
  }
  GL_REQUIRE(bytes >= need, "output buffer too small");
  if (host_src) {
    memcpy(host_out, src, need);
    return GL_OK;
  }
  GL_HIP(hipSetDevice(g->device));
  if (field == GL_F_COV_INV) {  // strided out of rec12
    std::vector<double> rec(K * 12);
    GL_HIP(hipMemcpy(rec.data(), g->rec12, K * 96, hipMemcpyDeviceToHost));
    double* o = (double*)host_out;
    for (size_t k = 0; k < K; ++k) memcpy(o + 9 * k, rec.data() + 12 * k + 3, 72);
    return GL_OK;
  }
  if (need) GL_HIP(hipMemcpy(host_out, src, need, hipMemcpyDeviceToHost));
  return GL_OK;
}

}  // extern "C"
