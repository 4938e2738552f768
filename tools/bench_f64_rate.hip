// Issue rate of the fp64 vector instructions of gfx950, per wave: v_fma_f64 / v_mul_f64 / v_add_f64 in NCH independent chains,
// one to four waves per SIMD (a workgroup of 256 / 512 / 768 / 1024 threads on every CU).  Prints clock64() cycles per instruction
// and wave.   hipcc --offload-arch=gfx950 -O3 tools/bench_f64_rate.hip -o build_tmp/bench_f64_rate && build_tmp/bench_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int NCH>
__global__ void k_rate(double* out, long long* cyc, int iters, double seed) {
  double a[NCH];
  const double m = 1.0000001 + seed * 1e-9, c = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NCH; ++i) a[i] = 1.0 + i * 0.001 + c;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int NCH>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 4000;  // (long launches: the launch overhead is < 1 % of them)
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  for (int w = 0; w < 3; ++w) k_rate<OP, NCH><<<blocks, threads>>>(out, cyc, iters, 0.0);
  hipDeviceSynchronize();
  // the same launch under HIP events: instructions per second and SIMD against the 0.6 G/s (2.4 GHz / 4 cycles per wave64 fp64
  // instruction) the 78.6 TFLOP/s are computed from - clock64() ticks are not core cycles
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 20;
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) k_rate<OP, NCH><<<blocks, threads>>>(out, cyc, iters, 0.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ginst = (double)iters * 8 * NCH * (threads / 256) / (ms * 1e-3 / reps) * 1e-9;
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : h) mean += (double)v;
  mean /= h.size();
  printf("%-28s %2d chains, %d waves per SIMD: %6.2f ticks per instruction and wave (%.2f per SIMD); %.3f G instructions/s per SIMD by HIP events = %.2f of 0.6\n",
         name, NCH, threads / 256, mean / ((double)iters * 8 * NCH), mean / ((double)iters * 8 * NCH) / (threads / 256), ginst, ginst / 0.6);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 768, 1024}) {
    run<0, 8>("v_fma_f64", threads);
    run<1, 8>("v_mul_f64", threads);
    run<2, 8>("v_add_f64", threads);
    run<0, 1>("v_fma_f64 (dependent)", threads);
    run<1, 1>("v_mul_f64 (dependent)", threads);
    run<2, 1>("v_add_f64 (dependent)", threads);
    run<0, 2>("v_fma_f64", threads);
    run<0, 4>("v_fma_f64", threads);
    run<0, 16>("v_fma_f64", threads);
  }
  return 0;
}
