// Single-pose structure-constrained refinement, on-chip fast path: three instantiations of
// gl_ba_fast_impl.hpp (block shape = threads per frame / LDS capacity in points) + the launcher.
#include <algorithm>
#include <cstdlib>

#include "gl_ba_common.hpp"

using namespace gld;
using namespace glba;

#define GL_BAF_NS baf512
#define GL_BAF_TF 512
#define GL_BAF_MCAP 2000
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_TF
#undef GL_BAF_MCAP

#define GL_BAF_NS baf256
#define GL_BAF_TF 256
#define GL_BAF_MCAP 1000
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_TF
#undef GL_BAF_MCAP

#define GL_BAF_NS baf128
#define GL_BAF_TF 128
#define GL_BAF_MCAP 500
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_TF
#undef GL_BAF_MCAP

// latency shape: the 512-thread kernel with the points of one frame dealt to NB <= 4 workgroups
#define GL_BAF_NS baf512c
#define GL_BAF_TF 512
#define GL_BAF_MCAP 2000
#define GL_BAF_COOPERATIVE
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_TF
#undef GL_BAF_MCAP
#undef GL_BAF_COOPERATIVE

namespace gl {

// Block shapes: a frame of up to 500 / 1000 / 2000 points runs on 128 / 256 / 512 threads with
// 40 / 80 / 160 KB of LDS, i.e. 4 / 2 / 1 frames per CU (always 2 waves per SIMD).  The small shapes
// matter for real tracking frames (150-500 map points): several frames share a CU, so the serial
// 6x6 solve and the barriers of one frame overlap with the passes of the others.
bool ba1_fast_supported(int L) { return L <= 2000; }

template <typename KernelT>
static int launch_shape(KernelT kern, int TF_, int MCAP_, Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm,
                        int B, int L, double* pose, double* pts, const double* obs, const int32_t* oct, int32_t* assoc,
                        const double* d2, double gate, uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch) {
  const size_t lds = (size_t)(10 * MCAP_ + (TF_ / 64) * 32 + 64 + 8) * sizeof(double);
  static size_t lds_set[3] = {0, 0, 0};  // per block shape
  GL_HIP(ensure_dynamic_lds((const void*)kern, lds, &lds_set[TF_ == 128 ? 0 : TF_ == 256 ? 1 : 2]));
  GmmDev gm{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags};
  {
    TimerScope ts(c, GL_TIMER_BA);
    kern<<<B, TF_, lds, c->stream>>>(make_bak(cam, prm, gate), gm, B, L, pose, pts, obs, oct, assoc, d2, dropped, erase,
                                     iters, (double*)scratch, (c->stats && c->stats_n >= B) ? c->stats : nullptr);
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// Few frames (the frame-at-a-time caller): one frame's points are dealt to NB = ceil(L / 512) workgroups of 512
// threads - one point per thread - on NB CUs; the two reductions of a Levenberg trial then cross the
// workgroups through tagged words in global memory (cooperative launch keeps them co-resident).
// One frame of 2 000 points: 0.68 -> 0.47 ms; 64 frames: 1.11 -> 0.79 ms.
static int launch_coop(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, int NB, double* pose,
                       double* pts, const double* obs, const int32_t* oct, int32_t* assoc, const double* d2, double gate,
                       uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch) {
  const size_t lds = (size_t)(10 * 2000 + (512 / 64) * 32 + 64 + 8) * sizeof(double);
  static size_t lds_set = 0;
  GL_HIP(ensure_dynamic_lds((const void*)baf512c::k_ba1_fast, lds, &lds_set));
  GmmDev gm{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags};
  BaK kk = make_bak(cam, prm, gate);
  // the exchange words of the frames sit behind the plane records
  unsigned long long* parts = (unsigned long long*)((char*)scratch + (((size_t)B * L * 32 + 63) / 64) * 64);
  double* pn = (double*)scratch;
  int32_t* stats = (c->stats && c->stats_n >= B) ? c->stats : nullptr;
  {
    TimerScope ts(c, GL_TIMER_BA);
    GL_HIP(hipMemsetAsync(parts, 0, (size_t)B * 2 * NB * 64 * sizeof(unsigned long long), c->stream));
    void* args[] = {&kk, &gm, &B, &L, &pose, &pts, &obs, &oct, &assoc, &d2, &dropped, &erase, &iters, &pn, &stats, &NB, &parts};
    if (hipLaunchCooperativeKernel((const void*)baf512c::k_ba1_fast, dim3(B * NB), dim3(512), args, lds, c->stream) !=
        hipSuccess) {
      (void)hipGetLastError();  // not co-resident on this device: the caller falls back to one workgroup per frame
      return 1;
    }
  }
  return GL_OK;
}

int launch_ba1_fast(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, double* pose,
                    double* pts, const double* obs, const int32_t* oct, int32_t* assoc, const double* d2, double gate,
                    uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch) {
  {
    int NB = (L + 511) / 512;  // <= 4
    bool coop = NB > 1 && B * NB <= c->ncu;  // one 160-KB workgroup per CU (measured: 64 frames 0.84 vs 1.15 ms)
    if (const char* e = getenv("GMMLOC_BA_COOP")) {  // 0 = never; n >= 2 = that many workgroups per frame (tests)
      const int v = atoi(e);
      coop = v >= 2 && B * v <= c->ncu;
      if (coop) NB = std::min(v, 4);
    }
    if (coop) {
      const int rc = launch_coop(c, g, cam, prm, B, L, NB, pose, pts, obs, oct, assoc, d2, gate, dropped, erase, iters, scratch);
      if (rc <= 0) return rc;
    }
  }
  // small shapes put several frames on a CU; with no more frames than CUs a frame takes the 512 threads instead
  // (300 points, one frame: 0.51 -> 0.30 ms; 64 frames: 0.80 -> 0.66 ms)
  int shape = B <= c->ncu ? 512 : (L <= 500 ? 128 : (L <= 1000 ? 256 : 512));
  if (const char* e = getenv("GMMLOC_BA_THREADS")) {  // tuning knob: force a block shape that fits
    const int t = atoi(e);
    if ((t == 128 && L <= 500) || (t == 256 && L <= 1000) || t == 512) shape = t;
  }
  if (shape == 128)
    return launch_shape(baf128::k_ba1_fast, 128, 500, c, g, cam, prm, B, L, pose, pts, obs, oct, assoc, d2, gate, dropped,
                        erase, iters, scratch);
  if (shape == 256)
    return launch_shape(baf256::k_ba1_fast, 256, 1000, c, g, cam, prm, B, L, pose, pts, obs, oct, assoc, d2, gate, dropped,
                        erase, iters, scratch);
  return launch_shape(baf512::k_ba1_fast, 512, 2000, c, g, cam, prm, B, L, pose, pts, obs, oct, assoc, d2, gate, dropped,
                      erase, iters, scratch);
}

}  // namespace gl
