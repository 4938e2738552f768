// Compacted problems of gl_optimize_current_pose (round 6), shared by gl_refine_pose.hip (the public entry point compacts a caller's
// problem) and gl_chain.hip (the tracked-frame chain compacts while it gathers a frame's associations into its pose problem).
//
// k_optimize_current_pose costs what its SLOTS cost, and the reference's frame has one slot per FEATURE (1 200) of which a few hundred
// hold a map point: five groups of the summation order where <= 1 024 slots are four, every wave evaluating four mostly empty slots per
// pass.  pose_compact_frame (one workgroup per frame) moves the edges, in slot order, to the front of a problem of stride
// MC <= 1 024 and DEALS the list's chunks over the groups (chunk c -> group c % G, its c / G-th chunk: the waves of the frame-at-a-time
// shapes get equal shares and skip the slots nobody uses).  A frame with more than MC edges keeps its full-stride problem (ovf[b] = 1:
// the compacted problem's workgroup returns at once, the full-stride one's runs).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace gl {

// what the compaction leaves for B problems of M slots (device pointers)
struct PoseCompacted {
  int MC;             // stride of the compacted problems
  double* Xw_c;       // B x MC x 3
  double* obs_c;      // B x MC x 3
  int32_t* oct_c;     // B x MC   (-1: no edge in the slot)
  uint8_t* outl_c;    // B x MC   (the flags of the shapes that keep them in memory)
  int32_t* slot_of;   // B x M    slot of the compacted problem that holds edge i, or -1
  int32_t* src_of;    // B x MC   the edge a slot holds (meaningless where oct_c < 0)
  int32_t* ovf;       // B        1: more than MC edges
  void* chi_c;        // B x MC doubles (shapes with the edges in memory), or null: taken from the context's first scratch block
  void* chi_f;        // B x M doubles, or null
};

// SRC: int octave(int i) const (< 0: no edge);  void load(int i, double* X, double* O) const   - edge i of THIS frame
// NT = threads of the workgroup (a multiple of 64)
template <int NT, class SRC>
__device__ __forceinline__ bool pose_compact_frame(const SRC& src, int b, int M, const PoseCompacted& pc) {  // -> the frame has more than MC edges
  constexpr int NWV = NT / 64;
  __shared__ int s_w[NWV];
  const int MC = pc.MC, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cnt = 0;
  for (int i = tid; i < M; i += NT) cnt += src.octave(i) >= 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  __syncthreads();  // (s_w may still be read by the caller's previous use of this function)
  if (lane == 0) s_w[wave] = cnt;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) total += s_w[w];
  const bool over = total > MC;
  if (tid == 0) pc.ovf[b] = over ? 1 : 0;
  const int nch = (MC + 63) / 64, DG = (nch + 3) / 4, DS = (nch + DG - 1) / DG;
  const bool deal = DG * DS == nch && (MC & 63) == 0;
  for (int s_ = tid; s_ < MC; s_ += NT) {
    pc.oct_c[(size_t)b * MC + s_] = -1;
    pc.outl_c[(size_t)b * MC + s_] = 0;
  }
  __syncthreads();
  int base = 0;
  for (int i0 = 0; i0 < M; i0 += NT) {
    const int i = i0 + tid;
    const int oc = i < M ? src.octave(i) : -1;
    const bool act = oc >= 0;
    const unsigned long long bal = __ballot(act);
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int pre = base + __popcll(bal & ((1ull << lane) - 1ull)), round_total = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      pre += w < wave ? s_w[w] : 0;
      round_total += s_w[w];
    }
    if (i < M) {
      const size_t g = (size_t)b * M + i;
      const bool comp = act && !over;
      const int cch = pre >> 6, slot = deal ? (((cch % DG) * DS + cch / DG) << 6) + (pre & 63) : pre;
      pc.slot_of[g] = comp ? slot : -1;
      if (comp) {
        const size_t sc = (size_t)b * MC + slot;
        double X[3], O[3];
        src.load(i, X, O);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          pc.Xw_c[sc * 3 + j] = X[j];
          pc.obs_c[sc * 3 + j] = O[j];
        }
        pc.oct_c[sc] = oc;
        pc.src_of[sc] = i;
      }
    }
    base += round_total;
    __syncthreads();
  }
  return over;
}

// bytes of the buffers of a PoseCompacted (without chi_c / chi_f) and their placement in a block
inline size_t pose_compacted_bytes(int B, int M, int MC) {
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  const size_t nc = (size_t)B * MC, nm = (size_t)B * M;
  return 2 * up(nc * 24) + 2 * up(nc * 4) + up(nm * 4) + up((size_t)B * 4) + up(nc);
}
inline char* pose_compacted_place(char* s, int B, int M, int MC, PoseCompacted* pc) {
  auto up = [](size_t v) { return ((v + 255) / 256) * 256; };
  const size_t nc = (size_t)B * MC, nm = (size_t)B * M;
  auto take = [&](size_t bytes) {
    char* p = s;
    s += up(bytes);
    return (void*)p;
  };
  pc->MC = MC;
  pc->Xw_c = (double*)take(nc * 24);
  pc->obs_c = (double*)take(nc * 24);
  pc->oct_c = (int32_t*)take(nc * 4);
  pc->src_of = (int32_t*)take(nc * 4);
  pc->slot_of = (int32_t*)take(nm * 4);
  pc->ovf = (int32_t*)take((size_t)B * 4);
  pc->outl_c = (uint8_t*)take(nc);
  pc->chi_c = pc->chi_f = nullptr;
  return s;
}
// the stride the compaction uses for problems of M slots, or 0: not compacted (option pose_compact: -1 problems of more than 1 024 slots,
// 1 of more than 256, 0 never; cap: option pose_compact_cap, 1 024 - the tests lower it to drive frames into the full-stride problem)
inline int pose_compact_stride(int mode, int M, int cap) {
  if (mode == 0 || M <= (mode > 0 ? 256 : 1024)) return 0;
  cap = cap < 256 ? 256 : cap > 1024 ? 1024 : 256 * (cap / 256);
  const int r = 256 * ((M + 255) / 256);
  return r < cap ? r : cap;
}

}  // namespace gl
