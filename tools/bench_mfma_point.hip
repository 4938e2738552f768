// Can the fp64 MATRIX pipe take the per-point 3 x 3 products / the pose-block contraction of k_ba1_fast?  ("Next" #5:
// the contraction form, not the lane-adder form of tools/bench_mfma_reduce.hip.)
//
// The refine owns one point per LANE: its 3 x 3 blocks (A D^-1, M, C = M (A D^-1)^T ...) sit in the lane's registers, and a product
// of two of them is 27 (18 for a symmetric result) v_fma_f64 of that lane.  The only fp64 MFMAs of gfx950 are
//   v_mfma_f64_16x16x4_f64     D(16 x 16) += A(16 x 4) B(4 x 16), one value of A and of B per lane, 4 of D      (1 024 MACs, 8 passes)
//   v_mfma_f64_4x4x4_4b_f64    four independent blocks: D_b(4 x 4) += A_b(4 x 4) B_b(4 x 4), b = lane / 16; lane (b, i, k) holds
//                              A_b[i][k] and B_b[k][i], and ONE value of D_b                                     (256 MACs, 4 passes)
// so a per-point product needs its operands spread over 16 LANES (one element each), i.e. a change of layout in front of and
// behind every product, and a 3 x 3 product fills 27 of a block's 64 MACs.  This file times, per 64 points (one wave slot):
//   mode 0  VALU      NPROD 3 x 3 products as 27 v_fma_f64 per lane and product (the kernel's form)
//   mode 1  MFMA+LDS  the same products through v_mfma_f64_4x4x4_4b: every lane writes its two 3 x 3 operands to LDS (18 ds_write_b64),
//                     16 MFMAs (4 points each) read theirs in the 16-lanes-per-point layout (2 ds_read_b64 each), the results go
//                     back through LDS into the owner's registers - EQUAL results (check column)
//   mode 2  MFMA pipe alone: the 16 MFMAs per product on operands already in the matrix layout (no layout change; results
//                     are NOT the products of the lane's blocks) - the floor of any matrix-pipe form
//   mode 3  pose-block contraction  H(6 x 7) += sum over 64 points and 3 rows of g_r^T (C g)_r  as 48 v_mfma_f64_4x4x4_4b (the 2 x 2
//                     tiles of the 6 x 7 block = the four blocks; K = 4 of the 192 (point, row) pairs per instruction), pipe alone
//   mode 4  ... the same sums on the VALU: pose_terms' 36 FMAs + 27 adds per lane, then nothing (the butterfly is per PASS, not per slot)
// each beside a FILLER of independent fp64 FMA chains on the same wave (what the rest of a pass is), so that a matrix pipe that
// really runs beside the VALU shows as "whole iteration < filler + product".  2 waves per SIMD as in the refine.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_mfma_point.hip -o build_tmp/bench_mfma_point
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define NPROD 3  // 3 x 3 products per point and slot that are candidates (A D^-1 rows are solves; M (A D^-1)^T, R H R^T twice)

__device__ __forceinline__ void mm3_valu(const double* X, const double* Y, double* Z) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Z[i * 3 + j] = fma(X[i * 3], Y[j], fma(X[i * 3 + 1], Y[3 + j], X[i * 3 + 2] * Y[6 + j]));
}

// Lane layout of v_mfma_f64_4x4x4_4b_f64, PROBED on an MI355X (tools/probe_mfma_layout.hip, profiles/r5_mfma_layout_probe.txt): the four
// blocks are NOT four groups of 16 consecutive lanes.  With k = lane / 16, b = (lane / 4) % 4, e = lane % 4:
//     A_b[i = e][k] and B_b[k][j = e] sit in lane 16 k + 4 b + e;   D_b[i][j] comes out in lane 16 i + 4 b + j.
// A point's 4 x 4 operand is therefore spread over four lanes in each of the wave's four 16-lane rows.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(double* out, long long* cyc, int iters, int filler) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per wave: operands 64 points x 2 x 16 doubles (padded 4 x 4); the results overwrite the first operand (every address is read
  // and written by the same lane)
  extern __shared__ double lds[];
  double* wX = lds + (size_t)wave * (64 * 16 * 2);
  double* wY = wX + 64 * 16;
  double* wZ = wX;
  double X[9], Y[9], Z[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    X[i] = 1.0 + 1e-3 * (double)((threadIdx.x * 7 + i * 13) & 63);
    Y[i] = 0.5 + 1e-3 * (double)((threadIdx.x * 11 + i * 5) & 63);
    Z[i] = 0.0;
  }
  if (MODE == 1 || MODE == 2 || MODE == 3) {  // zero the padding once (row / column 3 of the 4 x 4 blocks)
    for (int i = lane; i < 64 * 16 * 2; i += 64) wX[i] = 0.0;
  }
  double f0 = 1.0 + 1e-9 * lane, f1 = 0.5, f2 = 0.25, f3 = 0.125;
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  double hacc = 0.0;
  __syncthreads();
  const long long tb = clock64();
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < filler; ++j) {  // 4 independent fp64 FMA chains: the rest of a slot
      f0 = fma(f0, 0.999999, 1e-7);
      f1 = fma(f1, 0.999998, 2e-7);
      f2 = fma(f2, 0.999997, 3e-7);
      f3 = fma(f3, 0.999996, 4e-7);
    }
    X[0] += 1e-12 * f0;  // (the products depend on the iteration)
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        double T[9];
        mm3_valu(X, Y, T);
#pragma unroll
        for (int i = 0; i < 9; ++i) Z[i] += T[i];
        Y[p] += 1e-9;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        // owner layout -> LDS: point = lane, element (i, k) at [point][i * 4 + k] (A) and (k, j) at [point][j * 4 + k] (B, transposed:
        // the MFMA's lane (b, i, k) wants B_b[k][i])
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            wX[lane * 16 + i * 4 + kk] = X[i * 3 + kk];   // A[i][k] at [point][i * 4 + k]
            wY[lane * 16 + kk * 4 + i] = Y[kk * 3 + i];   // B[k][j] at [point][k * 4 + j]  (B = Y: B[k][j] = Y[k][j], here j = i)
          }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // 16 MFMAs, 4 points each: lane (b = lane / 16, e = lane % 16) reads element e of point 4 g + b
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int pt = 4 * g + ((lane >> 2) & 3), e = lane & 3, kq = lane >> 4;
          const double a = wX[pt * 16 + e * 4 + kq], b = wY[pt * 16 + kq * 4 + e];   // A[i = e][k], B[k][j = e]
          const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);   // lane holds D[i = kq][j = e] of its point
          wZ[pt * 16 + kq * 4 + e] = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) Z[i * 3 + j] += wZ[lane * 16 + i * 4 + j];
        Y[p] += 1e-9;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (the results overwrote the first operand's image in the same [i * 4 + j] layout: row / column 3 of a product of operands
        // padded with zeros are zero again, nothing to clear)
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        double d[4] = {Z[p], 0.0, 0.0, 0.0};  // four independent accumulation chains
#pragma unroll
        for (int g = 0; g < 16; ++g) d[g & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(X[g % 9], Y[(g + p) % 9], d[g & 3], 0, 0, 0);
        Z[p] = (d[0] + d[1]) + (d[2] + d[3]);
      }
    } else if (MODE == 3) {
      double d[4] = {hacc, 0.0, 0.0, 0.0};  // four partial accumulators (a single one would chain 48 dependent MFMAs)
#pragma unroll
      for (int g = 0; g < 48; ++g) d[g & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(X[g % 9], Y[(g + 3) % 9], d[g & 3], 0, 0, 0);
      hacc = (d[0] + d[1]) + (d[2] + d[3]);
    } else {
      // pose_terms of gl_ba_fast_impl.hpp: M = [q]x C (18 ops), 6 + 3 cross terms (18 ops), 27 separately rounded adds
      const double* q = X;
      const double* Cm = Y;
      const double Cf[9] = {Cm[0], Cm[1], Cm[2], Cm[1], Cm[3], Cm[4], Cm[2], Cm[4], Cm[5]};
      double M[9];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        M[j] = fma(q[1], Cf[6 + j], -q[2] * Cf[3 + j]);
        M[3 + j] = fma(q[2], Cf[j], -q[0] * Cf[6 + j]);
        M[6 + j] = fma(q[0], Cf[3 + j], -q[1] * Cf[j]);
      }
      double t[27];
      t[0] = fma(q[1], M[2], -q[2] * M[1]);
      t[1] = fma(q[2], M[0], -q[0] * M[2]);
      t[2] = fma(q[0], M[1], -q[1] * M[0]);
      t[3] = M[0]; t[4] = M[1]; t[5] = M[2];
      t[6] = fma(q[2], M[3], -q[0] * M[5]);
      t[7] = fma(q[0], M[4], -q[1] * M[3]);
      t[8] = M[3]; t[9] = M[4]; t[10] = M[5];
      t[11] = fma(q[0], M[7], -q[1] * M[6]);
      t[12] = M[6]; t[13] = M[7]; t[14] = M[8];
      t[15] = Cm[0]; t[16] = Cm[1]; t[17] = Cm[2]; t[18] = Cm[3]; t[19] = Cm[4]; t[20] = Cm[5];
      t[21] = fma(q[1], Cm[8], -q[2] * Cm[7]);
      t[22] = fma(q[2], Cm[6], -q[0] * Cm[8]);
      t[23] = fma(q[0], Cm[7], -q[1] * Cm[6]);
      t[24] = Cm[6]; t[25] = Cm[7]; t[26] = Cm[8];
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] += t[i];
    }
  }
  const long long te = clock64();
  double s = f0 + f1 + f2 + f3 + hacc;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += Z[i];
#pragma unroll
  for (int i = 0; i < 27; ++i) s += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = te - tb;
}

template <int MODE>
static void run(const char* name, int iters, int filler, double* out, long long* cyc, double* hout, long long* hcyc) {
  const int NB = 256;
  const size_t lds = 8 * 64 * 16 * 2 * sizeof(double);  // 128 KB: one workgroup per CU
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<MODE><<<NB, 512, lds>>>(out, cyc, 10, filler);  // warm-up
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", name);
    return;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<NB, 512, lds>>>(out, cyc, iters, filler);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hcyc, cyc, NB * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  hipMemcpy(hout, out, sizeof(double) * 64, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < NB * 8; ++i) mean += (double)hcyc[i];
  mean /= NB * 8;
  double chk = 0;
  for (int i = 0; i < 64; ++i) chk += hout[i];
  printf("filler %4d FMA-quads  %-44s : whole iteration %9.1f clock64 ticks per wave (2 waves per SIMD), %8.3f us per iteration by HIP events  (check %.9g)\n",
         filler, name, mean / iters, 1e3 * ms / iters, chk);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  double *out, *hout = (double*)malloc(64 * sizeof(double));
  long long *cyc, *hcyc = (long long*)malloc(256 * 8 * sizeof(long long));
  hipMalloc(&out, sizeof(double) * 256 * 512);
  hipMalloc(&cyc, sizeof(long long) * 256 * 8);
  printf("# tools/bench_mfma_point.hip: per 64 points (one wave slot), %d 3 x 3 products per point; 256 workgroups x 8 waves, %d iterations;\n", NPROD, iters);
  printf("# 'filler' = independent fp64 FMA chains of the same wave per iteration (a slot of pass A is ~400 instructions = 100 quads).\n");
  printf("# Columns are WHOLE-ITERATION times; a form that overlaps with the VALU shows as less than filler + its own time at filler 0.\n");
  for (int filler : {0, 100}) {
    run<0>("VALU: 27 v_fma_f64 per product and lane", iters, filler, out, cyc, hout, hcyc);
    run<1>("v_mfma_f64_4x4x4_4b + LDS layout change (= results)", iters, filler, out, cyc, hout, hcyc);
    run<2>("v_mfma_f64_4x4x4_4b pipe alone (16 per product)", iters, filler, out, cyc, hout, hcyc);
    run<4>("pose block, VALU: pose_terms + 27 adds", iters, filler, out, cyc, hout, hcyc);
    run<3>("pose block, 48 x v_mfma_f64_4x4x4_4b pipe alone", iters, filler, out, cyc, hout, hcyc);
  }
  return 0;
}
