// Cost of the per-problem barrier of k_ba_gen (atomic counter + device-scope fences) for NB co-resident
// workgroups, with the workgroups spread over the XCDs (consecutive ids) or packed on one XCD (ids = 8 k).
//   hipcc --offload-arch=gfx950 -O3 tools/bench_barrier.hip -o build_tmp/bench_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ void prob_sync(unsigned* bar, int NB) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned gen = atomicAdd(&bar[1], 0u);
    if (atomicAdd(&bar[0], 1u) == (unsigned)NB - 1u) {
      bar[0] = 0u;
      __threadfence();
      atomicAdd(&bar[1], 1u);
    } else {
      while (atomicAdd(&bar[1], 0u) == gen) __builtin_amdgcn_s_sleep(2);
    }
    __threadfence();
  }
  __syncthreads();
}
__global__ void k(unsigned* bar, int NB, int iters, int stride, double* junk, int dirty) {
  if (blockIdx.x % stride) return;
  const int pb = blockIdx.x / stride;
  for (int i = 0; i < iters; ++i) {
    if (dirty) for (int j = threadIdx.x; j < dirty; j += blockDim.x) junk[(size_t)pb * dirty + j] = i + j;  // dirty lines to write back
    prob_sync(bar, NB);
  }
}
int main() {
  unsigned* bar; double* junk;
  hipMalloc(&bar, 64); hipMalloc(&junk, 64 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dirty : {0, 4096}) for (int stride : {1, 8}) for (int NB : {2, 8, 32}) {
    hipMemset(bar, 0, 64);
    int iters = 2000;
    void* args[] = {&bar, &NB, &iters, &stride, &junk, &dirty};
    hipLaunchCooperativeKernel((const void*)k, dim3(NB * stride), dim3(256), args, 0, 0);  // warm
    hipDeviceSynchronize();
    hipMemset(bar, 0, 64);
    hipEventRecord(e0);
    hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(NB * stride), dim3(256), args, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dirty %5d doubles/wg  stride %d (%s)  NB %2d : %.2f us per barrier  (%s)\n", dirty, stride, stride == 8 ? "one XCD" : "spread", NB, 1e3 * ms / iters, hipGetErrorString(e));
  }
  return 0;
}
