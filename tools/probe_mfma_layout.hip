// Lane layout of v_mfma_f64_4x4x4_4b_f64, probed: one 1.0 in A (lane la) and one in B (lane lb) -> the lane(s) of D that are non-zero.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_layout.hip -o build_tmp/probe_mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(unsigned long long* hit) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) hit[la * 64 + lb] = m;
    }
}
int main() {
  unsigned long long *d, h[4096];
  (void)hipMalloc(&d, sizeof(h));
  k_probe<<<1, 64>>>(d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb]) {
        printf(" B%d->D", lb);
        for (int l = 0; l < 64; ++l)
          if (h[la * 64 + lb] >> l & 1) printf("%d,", l);
      }
    printf("\n");
  }
  return 0;
}
