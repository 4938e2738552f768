// Standalone check of gld::wave_reduce_scatter32 / block_reduce on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igmmloc_amd/csrc tools/test_reduce.hip -o /tmp/test_reduce && /tmp/test_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gl_device.hpp"
using namespace gld;
__global__ void k(const double* in, double* out, int* slot) {
  double v[32];
  for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x * 32 + i];
  const double r = wave_reduce_scatter32(v);
  out[threadIdx.x] = r;
  slot[threadIdx.x] = wave_slot_owner(threadIdx.x & 63) ? wave_slot(threadIdx.x & 63) : -1;
}
int main() {
  const int T = 128;
  double* in; double* out; int* slot;
  hipMallocManaged(&in, T * 32 * 8); hipMallocManaged(&out, T * 8); hipMallocManaged(&slot, T * 4);
  for (int t = 0; t < T; ++t) for (int i = 0; i < 32; ++i) in[t * 32 + i] = (double)((t * 131 + i * 17) % 1009) + 0.25 * i;
  k<<<1, T>>>(in, out, slot);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  int bad = 0;
  for (int w = 0; w < T / 64; ++w) {
    int seen[32] = {0};
    for (int l = 0; l < 64; ++l) {
      const int s = slot[w * 64 + l];
      if (s < 0) continue;
      seen[s]++;
      double ref = 0; for (int t = 0; t < 64; ++t) ref += in[(w * 64 + t) * 32 + s];
      if (fabs(ref - out[w * 64 + l]) > 1e-9 * fabs(ref)) { if (bad < 5) printf("wave %d lane %d slot %d: %f vs %f\n", w, l, s, out[w*64+l], ref); ++bad; }
    }
    for (int i = 0; i < 32; ++i) if (seen[i] != 1) { printf("slot %d seen %d times\n", i, seen[i]); ++bad; }
  }
  printf(bad ? "FAILED (%d)\n" : "reduce-scatter OK\n", bad);
  return bad != 0;
}
