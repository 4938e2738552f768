// Can the idle fp64 matrix pipe take the per-trial cross-lane sums of k_ba1_fast?  ("What's weak" #3.)
// One workgroup of 8 waves per CU (2 per SIMD, as the refine kernel runs), every lane holds 29 partial sums; per iteration a
// filler of dependent-chain fp64 FMAs stands in for a pass (4 point slots), then the 29 values are summed over the wave by
//   mode 0  the kernel's butterfly (gld::wave_reduce_scatter32: permlane32/16 swaps + DPP, 32 values -> one total per lane)
//   mode 1  v_mfma_f64_16x16x4_f64 with B = 1: D[r][c] = sum_k A[r][k] adds the 4 lanes {r, r+16, r+32, r+48} of one register;
//           3 VALU adds fold the 4 result registers of a lane, a second MFMA adds the 4 rows: 2 MFMAs + 3 adds per VALUE
//   mode 2  hybrid: one MFMA + 3 adds per value, then the butterfly's two lane-swap stages across the rows
// and the cycles of the reduction alone (s_memtime around it, mean over waves and iterations) and of the whole iteration are
// reported.  K of the instruction is 4, i.e. one MFMA adds FOUR lanes of ONE register: 64 lanes x 29 values need >= 29 x 2
// of them, each 8 passes on the SIMD's matrix pipe, in sequence at the end of a pass, where every wave of the workgroup arrives
// together and wave 0 waits for the totals.
//   hipcc --offload-arch=gfx950 -O3 -I gmmloc_amd/csrc tools/bench_mfma_reduce.hip -o build_tmp/bench_mfma_reduce
#include <hip/hip_runtime.h>

#include <cstdio>

#include "gl_device.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma_rowsum(double x) {  // every lane: sum of x over the lanes (l % 16) + 16 k, k = 0..3, folded over the lane's 4 result rows
  const v4d z = {0.0, 0.0, 0.0, 0.0};
  const v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0, z, 0, 0, 0);
  return (d[0] + d[1]) + (d[2] + d[3]);
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(double* out, long long* cyc, int iters, int filler) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double red[8 * 32];
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = i < 29 ? 1e-3 * (double)((threadIdx.x * 31 + i * 7) & 255) : 0.0;
  double f0 = 1.0 + 1e-9 * lane, f1 = 0.5, f2 = 0.25, f3 = 0.125;
  long long t_red = 0, t_all = 0;
  double sink = 0.0;
  const long long tb = clock64();
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < filler; ++j) {  // 4 independent fp64 FMA chains
      f0 = fma(f0, 0.999999, 1e-7);
      f1 = fma(f1, 0.999998, 2e-7);
      f2 = fma(f2, 0.999997, 3e-7);
      f3 = fma(f3, 0.999996, 4e-7);
    }
    double w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = i < 29 ? v[i] + f0 * (double)(i + 1) : 0.0;
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = clock64();
    if (MODE == 0) {
      const double r = gld::wave_reduce_scatter32(w);
      if (gld::wave_slot_owner(lane)) red[wave * 32 + gld::wave_slot(lane)] = r;
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 29; ++i) {
        const double t = mfma_rowsum(w[i]);       // 16 row sums, folded: t_m, m = lane / 16
        const v4d z = {0.0, 0.0, 0.0, 0.0};
        const v4d e = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, z, 0, 0, 0);  // sum over the 4 rows -> every lane
        w[i] = e[0];
      }
      if (lane < 29) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < 29; ++i) r = lane == i ? w[i] : r;
        red[wave * 32 + lane] = r;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 29; ++i) w[i] = mfma_rowsum(w[i]);
      gld::rs_swap_stage<16, 32>(w);
      gld::rs_swap_stage<8, 16>(w);
      // lane (row m): w[0..7] = totals of the values 16 (m >> 1) + 8 (m & 1) + 0..7
      if ((lane & 15) < 8) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) r = (lane & 7) == i ? w[i] : r;
        red[wave * 32 + ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + (lane & 7)] = r;
      }
    }
    const long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x < 32) {
      double s = 0.0;
      for (int ww = 0; ww < 8; ++ww) s += red[ww * 32 + threadIdx.x];
      sink += s;
    }
    __syncthreads();
    t_red += t1 - t0;
  }
  t_all = clock64() - tb;
  out[blockIdx.x * 512 + threadIdx.x] = sink + f0 + f1 + f2 + f3;
  if (lane == 0) {
    cyc[(blockIdx.x * 8 + wave) * 2] = t_red;
    cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t_all;
  }
}

int main() {
  const int NB = 256, iters = 200;
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * NB * 512);
  hipMalloc(&cyc, sizeof(long long) * NB * 16);
  long long h[NB * 16];
  double ho[32];
  const char* names[3] = {"butterfly (permlane swaps + DPP)", "2 x v_mfma_f64_16x16x4 per value", "1 x MFMA per value + 2 swap stages"};
  for (int filler : {0, 325}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k<0><<<NB, 512>>>(out, cyc, iters, filler);
        if (mode == 1) k<1><<<NB, 512>>>(out, cyc, iters, filler);
        if (mode == 2) k<2><<<NB, 512>>>(out, cyc, iters, filler);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
      double r = 0, a = 0;
      for (int i = 0; i < NB * 8; ++i) {
        r += h[i * 2];
        a += h[i * 2 + 1];
      }
      printf("filler %4d FMA-quads per iteration  %-36s : reduction %7.0f cycles per wave, whole iteration %8.0f cycles  (check %.6f)\n", filler,
             names[mode], r / (NB * 8.0 * iters), a / (NB * 8.0 * iters), ho[0]);
    }
  }
  return 0;
}
