// Single-pose structure-constrained refinement, on-chip fast path: the instances of gl_ba_fast_impl.hpp
// (DENSE by LDS class, SPREAD; exact fp64 point step, and the fp32-cached step as an option) + the launcher.
#include <algorithm>
#include <cstdlib>

#include "gl_ba_common.hpp"

using namespace gld;
using namespace glba;

// (namespace, LDS capacity in points, waves at most, SPREAD, fp32-cached step)
#define GL_BAF_NS bafd496   // DENSE, exact step: 4 frames per CU
#define GL_BAF_MCAP 496
#define GL_BAF_NW 2
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

#define GL_BAF_NS bafd1000  // 2 frames per CU
#define GL_BAF_MCAP 1000
#define GL_BAF_NW 4
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

#define GL_BAF_NS bafd2000  // 1 frame per CU
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

#define GL_BAF_NS bafs      // SPREAD (latency shape), exact step
#define GL_BAF_MCAP 512
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 1
#define GL_BAF_STEP32 0
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

// fp32-cached point step (option ba_step32): the SPREAD kernel and the largest DENSE class
#define GL_BAF_NS bafs32
#define GL_BAF_MCAP 512
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 1
#define GL_BAF_STEP32 1
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

#define GL_BAF_NS bafd2000s32
#define GL_BAF_MCAP 2000
#define GL_BAF_NW 8
#define GL_BAF_SPREAD 0
#define GL_BAF_STEP32 1
#include "gl_ba_fast_impl.hpp"
#undef GL_BAF_NS
#undef GL_BAF_MCAP
#undef GL_BAF_NW
#undef GL_BAF_SPREAD
#undef GL_BAF_STEP32

namespace gl {

bool ba1_fast_supported(int L) { return L <= 2000; }

// the canonical summation order of a frame of stride L (gl_ba_fast_impl.hpp): G groups of S chunks of 64 points
static void canon_order(int L, int* G, int* S) {
  const int nch = (L + 63) / 64;
  *G = (nch + 3) / 4;
  *S = (nch + *G - 1) / *G;
}

typedef void (*BafKernel)(BaK, GmmDev, int, int, int, int, double*, double*, const double*, const int32_t*, int32_t*, const double*,
                          uint8_t*, uint8_t*, int32_t*, double*, int32_t*, int, unsigned long long*);

struct BafArgs {
  BaK k;
  GmmDev gm;
  int B, L, G, S;
  double *pose, *pts;
  const double* obs;
  const int32_t* oct;
  int32_t* assoc;
  const double* d2;
  uint8_t *dropped, *erase;
  int32_t* iters;
  double* pn;
  int32_t* stats;
  int NB;
  unsigned long long* parts;
};

// one workgroup of G waves per frame; LDS class by stride (4 / 2 / 1 frames per CU)
static int launch_dense(Ctx* c, BafArgs& a) {
  const bool s32 = c->opt.ba_step32 != 0;
  const int cap = s32 ? 2000 : (a.L <= 496 ? 496 : a.L <= 1000 ? 1000 : 2000);
  const BafKernel kern = s32 ? bafd2000s32::k_ba1_fast : cap == 496 ? bafd496::k_ba1_fast : cap == 1000 ? bafd1000::k_ba1_fast : bafd2000::k_ba1_fast;
  const size_t lds = (size_t)(10 * cap + (cap == 496 ? 2 : cap == 1000 ? 4 : 8) * 32 + 64 + 40) * sizeof(double);
  GL_HIP(ensure_dynamic_lds(c, (const void*)kern, lds));
  a.NB = 1;
  a.parts = nullptr;
  {
    TimerScope ts(c, GL_TIMER_BA);
    kern<<<a.B, 64 * a.G, lds, c->stream>>>(a.k, a.gm, a.B, a.L, a.G, a.S, a.pose, a.pts, a.obs, a.oct, a.assoc, a.d2, a.dropped, a.erase,
                                            a.iters, a.pn, a.stats, a.NB, a.parts);
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// Few frames (the frame-at-a-time caller): one point per thread, 512 threads = one block of two groups per workgroup,
// NB = ceil(G / 2) workgroups per frame on as many CUs; with NB > 1 the reductions of a Levenberg trial cross the
// workgroups through tagged words in global memory (cooperative launch keeps them co-resident).
// Returns 1 when the cooperative launch is refused (the caller falls back to DENSE).
static int launch_spread(Ctx* c, BafArgs& a, void* scratch) {
  const BafKernel kern = c->opt.ba_step32 != 0 ? bafs32::k_ba1_fast : bafs::k_ba1_fast;
  const size_t lds = (size_t)(10 * 512 + 2 * 32 + 64 + 40 + 29 * 512) * sizeof(double);
  GL_HIP(ensure_dynamic_lds(c, (const void*)kern, lds));
  a.NB = (a.G + 1) / 2;
  // the exchange words of the frames sit behind the plane records
  a.parts = (unsigned long long*)((char*)scratch + (((size_t)a.B * a.L * 56 + 63) / 64) * 64);
  TimerScope ts(c, GL_TIMER_BA);
  if (a.NB == 1) {
    kern<<<a.B, 512, lds, c->stream>>>(a.k, a.gm, a.B, a.L, a.G, a.S, a.pose, a.pts, a.obs, a.oct, a.assoc, a.d2, a.dropped, a.erase, a.iters,
                                       a.pn, a.stats, a.NB, a.parts);
    GL_HIP(hipGetLastError());
    return GL_OK;
  }
  GL_HIP(hipMemsetAsync(a.parts, 0, (size_t)a.B * 2 * a.NB * 64 * sizeof(unsigned long long), c->stream));
  void* args[] = {&a.k, &a.gm, &a.B, &a.L, &a.G, &a.S, &a.pose, &a.pts, &a.obs, &a.oct, &a.assoc, &a.d2, &a.dropped, &a.erase, &a.iters, &a.pn,
                  &a.stats, &a.NB, &a.parts};
  if (hipLaunchCooperativeKernel((const void*)kern, dim3(a.B * a.NB), dim3(512), args, lds, c->stream) != hipSuccess) {
    (void)hipGetLastError();  // not co-resident on this device
    return 1;
  }
  return GL_OK;
}

// Shape: SPREAD when every workgroup of the batch gets a CU of its own (B NB <= CUs: the frame-at-a-time
// caller, small batches), DENSE otherwise.  Both add in the same order: the choice never shows in the results
// (option ba_shape forces one: 0 DENSE, 1 SPREAD).
int launch_ba1_fast(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, double* pose,
                    double* pts, const double* obs, const int32_t* oct, int32_t* assoc, const double* d2, double gate,
                    uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch) {
  BafArgs a;
  a.k = make_bak(cam, prm, gate);
  a.gm = GmmDev{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags};
  a.B = B;
  a.L = L;
  canon_order(L, &a.G, &a.S);
  a.pose = pose;
  a.pts = pts;
  a.obs = obs;
  a.oct = oct;
  a.assoc = assoc;
  a.d2 = d2;
  a.dropped = dropped;
  a.erase = erase;
  a.iters = iters;
  a.pn = (double*)scratch;
  a.stats = (c->stats && c->stats_n >= B) ? c->stats : nullptr;
  const int NB = (a.G + 1) / 2;
  bool spread = (long)B * NB <= c->ncu;
  if (c->opt.ba_shape == 0) spread = false;
  if (c->opt.ba_shape == 1) spread = true;  // forced (tests); a refused cooperative launch still falls back
  if (spread) {
    const int rc = launch_spread(c, a, scratch);
    if (rc <= 0) return rc;
  }
  return launch_dense(c, a);
}

}  // namespace gl
