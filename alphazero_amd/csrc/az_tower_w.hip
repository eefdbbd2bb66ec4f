// az_tower_w.hip -- the 64-channel residual conv tower of the GomokuModel family on 15x15 boards in FLOAT32 by the
// Winograd transform F(2x2, 3x3), for MI355X (gfx950).
//
// Same network and same contract as az_tower_f32.hip (examples/gomoku/model.py:41-90: conv3x3(9->64)+BN+ReLU, residual
// blocks of two conv3x3(64->64)+BN, the 1x1 head convs+BN+ReLU; the heads' Linear layers follow in k_heads_f32), the same
// arithmetic TYPE as the reference (nn_models/base.py:17-19: plain f32, no autocast) -- every operand, product and sum is
// f32 -- but a different ALGORITHM for the 3x3 convolutions: the one cuDNN / MIOpen pick for f32 3x3 convolutions on a GPU,
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      (Lavin & Gray; 2x2 outputs from a 4x4 input patch, 16 products
//                                                      per output pair of pairs instead of 36: 2.25x fewer multiplies)
// The 15x15 board is covered by 8 x 8 tiles of 2x2 outputs (the 16th row / column is dropped).  For each of the 16 transform
// points the layer is a GEMM  M_p[co 64][tile 64] = U_p[co][ci 64] V_p[ci][tile]  on v_mfma_f32_16x16x4_f32: 4,096 MFMAs per
// 64->64 layer and board against the direct convolution's 8,640 (az_tower_f32.hip).  The MFMA pipe still bounds the kernel
// (157.3 TFLOP/s dense f32); the transforms are a few dozen VALU instructions per 64 MFMAs.
//
// One workgroup of 8 waves owns one board at a time for the whole tower (9x9: two boards); activations live in LDS, the board
// inside a frame of zeros (act[row -1 .. 16][col -1 .. 16][64 ch] f32, 264 B per cell, 4,800 B per row: with these two pitches the
// 16 tiles x 2 channel pairs of a half-wave's ds_read_b64 fall on 32 different bank pairs), layers run in place.  Wave w = (tile
// group tg = w & 3: tile rows 2 tg, 2 tg + 1 x 8 tile columns = the N = 16 of the MFMA; channel half ch = w >> 2: 2 tiles of 16
// output channels) and holds ALL 16 transform points of its 16 tiles x 32 channels in 128 accumulator registers, so that
//   * a lane builds its B operands itself: it reads the 4x4 patch of its tile for 2 input channels (16 ds_read_b64 at the tile's
//     base + instruction immediates: the frame makes every element a plain read), forms V = B^T d B in registers (32 two-channel
//     adds) and feeds 64 MFMAs (16 points x 2 channel tiles x 2 k-steps) with it -- the transformed input never exists in memory;
//   * the inverse transform A^T M A is in-lane too: the 16 M_p of an (output channel, tile) pair are 16 registers of one lane.
// Transformed weights U (host: G g G^T in f64, rounded to f32; 256 KiB per layer) stream L2 -> LDS by LDS-DMA in k-steps of 4
// input channels (16 KiB) through four buffers, three k-steps ahead of their use and straight across layer and board boundaries
// (one LDS-only barrier per k-step; conv_w), laid out so that a lane's A operands of four points are one conflict-free
// ds_read_b128.  Bias and the block's residual enter through the accumulators: A^T M A maps M[0][0], -M[0][3], -M[3][0], M[3][3]
// to the four outputs of the tile one to one, so the lane that overwrites a block's input in the first conv's epilogue reads it
// there and starts the second conv's accumulators at bias + x (as az_tower_f32.hip); the other twelve accumulators of a tile start
// from the first MFMA's zero operand.  What bounds the kernel (measured): f32 MFMAs and VALU instructions do not
// co-execute, so the transforms' adds are MFMA time -- MFMA pipe busy 0.76, other VALU 0.13.  Every LDS-DMA and every LDS read of
// the conv loop is issued from inline assembly with hand-counted vmcnt / lgkmcnt waits (see glds16, lds_r2_imm, kstep_mfma).
// Boards are handed out by the library's ticket counter (az_tower.hip); the input is the engine's packed leaf batch (8
// bitboards + player) with its needs-the-network flags, or dense f32 planes.  Output: the f32 feature rows k_heads_f32 reads.
// C ABI at the bottom; declared in include/az_engine.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <utility>

#include "../../include/az_engine.h"

int tower_device(hipStream_t stream, int *dev);             // az_tower.hip
unsigned int *tower_tickets(int dev, hipStream_t stream);   // az_tower.hip
void tower_set_error(const std::string &msg);               // az_tower.hip

namespace aztw {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int C = 64;
constexpr int CSB = 264;                      // bytes per cell (64 f32 + 8)
constexpr int STAGE_BYTES = 32768;            // U of 8 input channels (two k-steps): [j 2][kq 4][ct 4][pg 4][co 16][4 points] f32
constexpr int KSTEP_BYTES = STAGE_BYTES / 2;  // one k-step (j): 4 input channels, one per 16-lane group kq
constexpr int MAX_LAYERS = 36;                // bias rows kept in LDS (what the CU's 160 KiB leave beside the framed 15x15 board)
constexpr int NTH = 512;
constexpr int OFF_ACT = 2 * STAGE_BYTES;

// Board geometry.  15x15: 8 x 8 tiles = four groups of 16, one board per workgroup.  9x9: 5 x 5 tiles = two groups (16 + 9 tiles,
// seven idle lanes), TWO boards per workgroup (wave w: board (w & 3) / 2); a ticket then stands for a pair of boards.
template <int N>
struct Geo {
  static constexpr int NV = N * N, TR = (N + 1) / 2, NT = TR * TR, GPB = (NT + 15) / 16, BPW = 4 / GPB;
  static_assert(GPB * BPW == 4, "the four tile groups of a workgroup");
  // The board sits in a frame of zeros: rows / columns -1 .. N + 1 (the tiles cover N + 1, their 4x4 patches reach one further on
  // each side).  No activation is ever written there, so every patch element is a plain read at the lane's tile base + an
  // instruction immediate: no per-element address registers, no tests.  Bytes per row: 18 cells + 48 / 12 cells + 72 -- with
  // CSB = 33 and RSB = 600 / 405 units of 8 B, tile (ty, tx) sits at 2 (ty RSB + tx CSB): 8 ty + tx / 5 ty + tx modulo 16 is
  // different for the 16 tiles of a group (the bank pairs of a half-wave's ds_read_b64).
  static constexpr int HR = N + 3;
  static constexpr int RSB = N == 15 ? 4800 : 3240;
  static_assert(RSB >= HR * CSB, "a row holds its frame");
  static constexpr int BOARD_BYTES = HR * RSB;
  static constexpr int DUMP_OFF = BPW * BOARD_BYTES;         // where the outputs of cells that do not exist go
  static constexpr int ACT_BYTES = (DUMP_OFF + CSB + 15) / 16 * 16;
  static __host__ __device__ constexpr int cell_off(int r, int c) { return (r + 1) * RSB + (c + 1) * CSB; }
  static constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = 8 * SWB;  // packed input: 8 states x 2 colours x WC words
  static constexpr int RGS = RGW + 1;                        // words per board in an input buffer: the ring words, the player word
  static constexpr int OFF_BL = OFF_ACT + ACT_BYTES;
  static constexpr int OFF_HW = OFF_BL + (1 + MAX_LAYERS) * C * 4;
  // [board][2 channel halves][3 heads][NV] partial head sums: in the fourth weight buffer, which is idle between a board's last
  // k-step and the next board's first (the stream's next three k-steps go to buffers 0 .. 2)
  static constexpr int OFF_HP = 3 * KSTEP_BYTES;
  static_assert(BPW * 2 * 3 * NV * 4 <= KSTEP_BYTES, "the head sums fit a weight buffer");
  static constexpr int OFF_RG = OFF_HW + 208 * 4;            // (3 x 64 head filters + 3 biases (+ pad) sit at OFF_HW)
  static constexpr int OFF_SL = OFF_RG + (2 * BPW * RGS * 4 + 15) / 16 * 16;  // two input buffers: the running ticket's, the one fetched ahead
  static constexpr int LDS_BYTES = OFF_SL + 16 * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert(OFF_BL % 16 == 0 && OFF_HW % 16 == 0 && OFF_HP % 16 == 0 && OFF_RG % 16 == 0, "alignment");
  static_assert(BPW * 2 * NV <= NTH && BPW * NV <= NTH && BPW * (RGW + 1) <= NTH, "one pass of the workgroup per phase");
};

#ifndef AZ_W_ABLATE
#define AZ_W_ABLATE 0  // diagnostic builds (wrong results, timing only): 1 no patch reads / input transform, 2 no stage barriers,
#endif                 // 4 no U reads, 8 no MFMAs, 16 no stage DMA

#ifndef AZ_W_STAMP
#define AZ_W_STAMP 0  // diagnostic build: s_memtime stamps of workgroup 7's sixth board, per wave (az_debug_w_stamps)
#endif
#if AZ_W_STAMP
static __device__ unsigned long long g_w_stamps[8 * 32];
#define WSTAMP(k) do { if (stamp_on && lane == 0) g_w_stamps[wave * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif

#define LDS_PTR(T, a) ((__attribute__((address_space(3))) T *)(uintptr_t)(a))
__device__ __forceinline__ f32x2 lds_r2(uint32_t a) { return *LDS_PTR(const f32x2, a); }
__device__ __forceinline__ f32x4 lds_r4(uint32_t a) { return *LDS_PTR(const f32x4, a); }
__device__ __forceinline__ void lds_w2(uint32_t a, f32x2 v) { *LDS_PTR(f32x2, a) = v; }

// async global -> LDS copy of 16 B per lane (LDS-DMA): the LDS destination is the wave-uniform `lds_base` + lane * 16.  Issued from
// inline assembly ON PURPOSE: the compiler makes every LDS read that might alias an LDS-DMA it knows of wait for vmcnt(0), which
// would serialise the weight stream that runs three k-steps ahead of its use; the reads of a buffer are ordered behind its DMA by
// hand (conv_w: vmcnt(2) + barrier).  A copy the compiler does not see can only make ITS OWN vmcnt waits stricter, never weaker.
// (address = wave-uniform 64-bit base in SGPRs + the lane's 32-bit byte offset: no VALU, no 64-bit VGPR pointers.)
__device__ __forceinline__ void glds16(const char *g_base, uint32_t lane_off, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_base), "v"(lane_off), "s"(g_base) : "memory");
}
// ... of 4 B per lane (the packed input words and the flag bytes fetched a ticket ahead)
__device__ __forceinline__ void glds4(const void *g_base, uint32_t lane_off, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(lds_base), "v"(lane_off), "s"(g_base) : "memory");
}
// workgroup barrier that orders LDS accesses only: __syncthreads() also drains every outstanding LDS-DMA (vmcnt(0)), which
// would serialise the weight stream that runs three k-steps ahead; what has to have landed is waited for explicitly (vmcnt(2))
// (a workgroup fence, even one restricted to the local address space, makes the compiler wait for the LDS-DMA too: the wave's own
// LDS accesses are drained by hand, and the "memory" clobber keeps the compiler from moving accesses across the barrier)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// one k-step of U (16 KiB, contiguous in global memory and in LDS alike): 1 KiB per wave-instruction, 2 per wave
__device__ __forceinline__ void kstep_dma(const char *src, uint32_t dst, int wave, int lane) {
#pragma unroll
  for (int p = 0; p < 2; ++p) glds16(src + (p * 8 + wave) * 1024, (uint32_t)lane * 16u, dst + (p * 8 + wave) * 1024);
}

// the lane's 4x4 patch of its two input channels of stage `so / 32` (16 ds_read_b64 at the tile's base `pb`; element and stage
// offsets are compile-time constants: they travel in the instruction)
// Issued from inline assembly so that the offsets DO travel in the instructions: left to itself the compiler keeps the sixteen
// base + element sums in registers (common subexpressions of all stages), which the kernel does not have.  The compiler does not
// count these reads in lgkmcnt: patch_wait() drains them before the transform (requests return in order, so a read the compiler
// does not know of can only make its own counted waits stricter).
template <int OFF>
__device__ __forceinline__ f32x2 lds_r2_imm(uint32_t base) {
  f32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(OFF));
  return r;
}
template <int RSB, int SO, int... I>
__device__ __forceinline__ void patch_read_seq(uint32_t pb, f32x2 (&d)[16], std::integer_sequence<int, I...>) {
  ((d[I] = (AZ_W_ABLATE & 1) ? (f32x2){__uint_as_float(pb + I), 1.0f} : lds_r2_imm<(I >> 2) * RSB + (I & 3) * CSB + SO>(pb)), ...);
}
template <int RSB, int SO>
__device__ __forceinline__ void patch_read(uint32_t pb, f32x2 (&d)[16]) {
  patch_read_seq<RSB, SO>(pb, d, std::make_integer_sequence<int, 16>{});
}
// ... two elements at a time (2 g, 2 g + 1): spread over the eight MFMA groups of a k-step, so that no U read queues behind
// sixteen patch reads (LDS requests return in order)
template <int RSB, int SO, int G2>
__device__ __forceinline__ void patch_read_pair(uint32_t pb, f32x2 (&d)[16]) {
  if (AZ_W_ABLATE & 1) { d[2 * G2] = (f32x2){__uint_as_float(pb), 1.0f}; d[2 * G2 + 1] = d[2 * G2]; return; }
  d[2 * G2] = lds_r2_imm<((2 * G2) >> 2) * RSB + ((2 * G2) & 3) * CSB + SO>(pb);
  d[2 * G2 + 1] = lds_r2_imm<((2 * G2 + 1) >> 2) * RSB + ((2 * G2 + 1) & 3) * CSB + SO>(pb);
}
__device__ __forceinline__ void patch_wait(f32x2 (&d)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
  asm volatile("" : "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
}
// a loop whose index is a compile-time constant in the body
template <int CNT, class F, int... I>
__device__ __forceinline__ void static_for_seq(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int CNT, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_seq<CNT>(f, std::make_integer_sequence<int, CNT>{}); }
// V = B^T d B (32 two-channel adds), point 4 i + q
__device__ __forceinline__ void patch_transform(const f32x2 (&d)[16], f32x2 (&v)[16]) {
  f32x2 t[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // B^T d
    t[c] = d[c] - d[8 + c];
    t[4 + c] = d[4 + c] + d[8 + c];
    t[8 + c] = d[8 + c] - d[4 + c];
    t[12 + c] = d[4 + c] - d[12 + c];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (B^T d) B
    v[4 * i] = t[4 * i] - t[4 * i + 2];
    v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
    v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
    v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
  }
}
// the 32 MFMAs of a k-step (channel J of the lane's pair): 16 points x 2 channel tiles.  U comes from the k-step's buffer at
// `ab` in 8 ds_read_b128, each requested one group of four MFMAs ahead: group 0 arrives in `a` (requested during the k-step
// before), and the last group's slot requests group 0 of the NEXT k-step from `abn` (its buffer has landed and is visible:
// conv_w) -- no U read is waited for right after a barrier.  FIRST: the layer's first k-step -- only the four points that
// carry bias / residual (0, 3, 12, 15) have live accumulators, the others start from the instruction's zero operand.
// U reads as inline assembly too (offset in the instruction), so that EVERY LDS read of the conv loop is outside the compiler's
// lgkmcnt bookkeeping and the waits can be counted by hand: requests return in order, and mixing counted with uncounted reads makes
// the compiler's own waits drain the queue.
template <int OFF>
__device__ __forceinline__ f32x4 lds_r4_imm(uint32_t base) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(OFF));
  return r;
}
template <int CNT>
__device__ __forceinline__ void lgkm_wait(f32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
// LDS requests group g of a k-step issues, in this order: U of group g + 2, U of the next k-step's group 0 / 1 (g = 2 / 3), two
// patch elements (PATCH: a k-step under which the next stage's patch is fetched)
template <bool LAST, bool PATCH>
__host__ __device__ constexpr int kstep_issued(int g) { return g < 0 ? 0 : (g + 2 < 8 ? 1 : 0) + (!LAST && (g == 2 || g == 3) ? 1 : 0) + (PATCH ? 2 : 0); }

// the 32 MFMAs of a k-step (channel J of the lane's pair): 16 points x 2 channel tiles.  U comes from the k-step's buffer at `ab`
// in 8 ds_read_b128, each requested TWO groups of four MFMAs ahead: groups 0 and 1 arrive in `a`, `a1` (requested during the k-step
// before: the next k-step's buffer has landed and is visible, conv_w); PATCH: the 16 patch reads of the next stage ride along, two
// per group.  Before group g's MFMAs: lgkmcnt(everything issued after U(g)).  FIRST: the layer's first k-step -- only the four
// points that carry bias / residual (0, 3, 12, 15) have live accumulators, the others start from the instruction's zero operand.
template <bool FIRST, bool LAST, bool PATCH, int J, int RSB, int SO>
__device__ __forceinline__ void kstep_mfma(uint32_t ab, uint32_t abn, uint32_t pb, f32x4 &a, f32x4 &a1, const f32x2 (&v)[16], f32x2 (&d)[16],
                                           f32x4 (&acc)[16][2]) {
  f32x4 a0n = a, a1n = a1;
  static_for<8>([&](auto gc) {  // group g: point row i = g >> 1, channel tile ct = g & 1
    constexpr int g = decltype(gc)::value, i = g >> 1, ct = g & 1;
    f32x4 an = a;
    if constexpr (g + 2 < 8) an = lds_r4_imm<((g + 2) & 1) * 1024 + ((g + 2) >> 1) * 256>(ab);
    if constexpr (g == 2 && !LAST) a0n = lds_r4_imm<0>(abn);
    if constexpr (g == 3 && !LAST) a1n = lds_r4_imm<1024>(abn);
    if constexpr (PATCH) patch_read_pair<RSB, SO, g>(pb, d);
    // (groups 0, 1: their U was requested a k-step ago and drained by the barrier's lgkmcnt(0); the wait -- everything this k-step
    // has issued so far may stay in flight -- only ties the MFMAs behind that barrier for the compiler)
    if constexpr (g >= 2) lgkm_wait<kstep_issued<LAST, PATCH>(g - 2) - 1 + kstep_issued<LAST, PATCH>(g - 1) + kstep_issued<LAST, PATCH>(g)>(a);
    else lgkm_wait<kstep_issued<LAST, PATCH>(g - 1) + kstep_issued<LAST, PATCH>(g)>(a);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = 4 * i + q;
      const bool fresh = FIRST && !(p == 0 || p == 3 || p == 12 || p == 15);
      if (AZ_W_ABLATE & 8) acc[p][ct][q] = (fresh ? 0.0f : acc[p][ct][q]) + a[q] * v[p][J];
      else acc[p][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], v[p][J], fresh ? (f32x4){0.0f, 0.0f, 0.0f, 0.0f} : acc[p][ct], 0, 0, 0);
    }
    a = a1;
    if constexpr (g + 2 < 8) a1 = an;
    else a1 = g == 6 ? a0n : a1n;
  });
}

// One conv layer in the transform domain: acc[point][ct] = (preloaded bias / residual) + U V over T = 2 NS k-steps of 4 input
// channels (per lane: one), fully unrolled -- every LDS offset is an instruction immediate.  U streams through FOUR 16 KiB
// buffers, k-step t in buffer t & 3 (T is a multiple of 4: every layer starts in buffer 0), three k-steps ahead of its use and
// straight across layer and board boundaries (after this layer's last k-step the stream continues with `wnext`: the next
// layer, or the next board's stem).  Top of k-step t: wait until the DMA of k-step t + 1 has landed (the one of t + 2 may be in
// flight: vmcnt(2)), barrier -- every wave has finished k-step t - 1, whose buffer is the one k-step t + 3 goes to, and k-step
// t + 1's buffer is now complete and visible to all, so its first U read may be issued during k-step t -- then the 32 MFMAs on
// the V formed one stage (two k-steps) ago; the next stage's patch is requested between the stage's two k-steps and
// transformed after the second.  f32 MFMAs and ordinary VALU instructions share the SIMD's multipliers (nothing co-executes:
// az_tower_f32.hip), so every VALU instruction in this loop is MFMA time: immediates, no address arithmetic.
template <int NS, int RSB>
__device__ __forceinline__ void conv_w(uint32_t lds0, uint32_t pb, uint32_t abase, const char *wg, const char *wnext,
                                       int wave, int lane, f32x4 (&acc)[16][2]) {
  constexpr int T = 2 * NS;
  static_assert(T % 4 == 0, "a layer ends in the buffer it started in");
  f32x2 d[16], v[16];
  f32x4 a, a1;
  static_for<T>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (t == 0 || !(AZ_W_ABLATE & 2)) lds_barrier();
    if (!(AZ_W_ABLATE & 16)) kstep_dma(t + 3 < T ? wg + (size_t)(t + 3) * KSTEP_BYTES : wnext + (size_t)(t + 3 - T) * KSTEP_BYTES,
                                       lds0 + (uint32_t)(((t + 3) & 3) * KSTEP_BYTES), wave, lane);
    const uint32_t ab = abase + (uint32_t)((t & 3) * KSTEP_BYTES), abn = abase + (uint32_t)(((t + 1) & 3) * KSTEP_BYTES);
    if constexpr (t == 0) {
      a = lds_r4_imm<0>(ab);
      a1 = lds_r4_imm<1024>(ab);
      patch_read<RSB, 0>(pb, d);
      patch_wait(d);  // (lgkmcnt(0): the two U reads as well)
      lgkm_wait<0>(a);
      lgkm_wait<0>(a1);
      patch_transform(d, v);
    }
    // (odd k-steps: the next stage's patch in flight under the MFMAs, two elements per MFMA group)
    kstep_mfma<t == 0, t + 1 == T, ((t & 1) && t + 1 < T), (t & 1), RSB, (t / 2 + 1) * 32>(ab, abn, pb, a, a1, v, d, acc);
    if constexpr ((t & 1) && t + 1 < T) {
      patch_wait(d);
      patch_transform(d, v);
    }
  });
}

template <int N>
__global__ __launch_bounds__(NTH, 2) void k_tower_f32w(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                        const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                        const char *__restrict__ u0, const char *__restrict__ ul,
                                                        const float *__restrict__ bias, int L, const float *__restrict__ headw,
                                                        const float *__restrict__ headb, float *__restrict__ feat, int FS,
                                                        unsigned int *tickets) {
  using Q = Geo<N>;
  constexpr int NV = Q::NV, TR = Q::TR, NT = Q::NT, GPB = Q::GPB, BPW = Q::BPW, RSB = Q::RSB, BOARD_BYTES = Q::BOARD_BYTES;
  constexpr int DUMP_OFF = Q::DUMP_OFF, ACT_BYTES = Q::ACT_BYTES, WC = Q::WC, SWB = Q::SWB, RGW = Q::RGW, RGS = Q::RGS;
  extern __shared__ __attribute__((aligned(1024))) char smem_w[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_w;
  const uint32_t act0 = lds0 + OFF_ACT;
  char *act = smem_w + OFF_ACT;
  float *bl = (float *)(smem_w + Q::OFF_BL);
  float *hw = (float *)(smem_w + Q::OFF_HW);
  float *hp = (float *)(smem_w + Q::OFF_HP);
  uint32_t *rgb = (uint32_t *)(smem_w + Q::OFF_RG);
  // {0: this ticket (a group of BPW boards), 1: the next ticket, 2: which of its boards are evaluated (bit b; -1: not known yet),
  //  3: the ticket whose packed input sits in the input buffer the running ticket does not use (-1: none), 4 + b: board b of this
  //  ticket is evaluated, 6: landing word of the flag bytes fetched ahead, 7: the ticket to fetch ahead (-1: none), 8: its boards}
  int *slots = (int *)(smem_w + Q::OFF_SL);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4, chalf = wave >> 2;
  const int wb = (wave & 3) / GPB, tg = (wave & 3) % GPB;  // the wave's board of the workgroup and tile group of that board
  const int tile = 16 * tg + n, ty = tile / TR, tx = tile - ty * TR;  // (tile >= NT: an idle lane -- reads zeros, writes to the dump cell)

  for (int i = tid; i < ACT_BYTES / 16; i += NTH) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // the frames stay zero for good
  for (int i = tid; i < (1 + L) * C; i += NTH) bl[i] = bias[i];
  for (int i = tid; i < 3 * C + 3; i += NTH) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];

  // the lane's 4x4 input patch starts at cell (2 ty - 1, 2 tx - 1) = frame cell (2 ty, 2 tx); its channel pair 2 kq (+ 8 per stage).
  // (An idle lane of a 9x9 group reads tile 0's patch: finite values whose results go to the dump cell.)
  const uint32_t pb = act0 + (uint32_t)(wb * BOARD_BYTES + (tile < NT ? 2 * ty * RSB + 2 * tx * CSB : 0) + kq * 8);
  // Everything the epilogues need per lane (its output cells, its channels) is re-derived from a laundered copy of the lane index
  // inside each epilogue: hoisted out of the board loop these values would sit in registers through the conv loops, which run
  // at the register budget (a spill there is a scratch load, and a scratch load is a vmcnt(0) in the weight stream)
  auto fresh = [](int x) { asm volatile("" : "+v"(x)); return x; };
  const uint32_t abase = lds0 + (uint32_t)(kq * 4096 + chalf * 2048 + n * 16);  // U of a k-step: [kq 4][ct 4][i 4][co 16][q 4]

  for (int t = 0; t < 3; ++t) kstep_dma(u0 + t * KSTEP_BYTES, lds0 + (uint32_t)(t * KSTEP_BYTES), wave, lane);  // the stem's first three k-steps
  if (tid == 0) { slots[1] = (int)blockIdx.x; slots[2] = -1; slots[3] = -1; slots[6] = 0; slots[7] = -1; slots[8] = 0; }
  __syncthreads();
  int rsel = 0;  // (uniform) which input buffer the running ticket expands

  f32x4 acc[16][2];
  // A^T M A sends M[0][0], -M[0][3], -M[3][0], M[3][3] to outputs 00, 01, 10, 11: the accumulators of these four points carry
  // bias (+ residual) into the layer; the other twelve start from the first MFMA's zero operand (stage_mfma<true>)
  auto preload = [&](const f32x4 (&pre)[4][2]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      acc[0][ct] = pre[0][ct];
      acc[3][ct] = -pre[1][ct];
      acc[12][ct] = -pre[2][ct];
      acc[15][ct] = pre[3][ct];
    }
  };
  // The epilogue of a layer.  Part 1, BEFORE the barrier that ends the layer's reads (under the wait for the slowest wave): the
  // next layer's bias and -- when the activations this lane is about to overwrite are the residual of the block's second conv --
  // those activations (only this lane ever writes them).  Part 2, after the barrier: inverse transform, ReLU, store in place; the
  // next layer's accumulators start at bias (+ residual).
  f32x4 epre[4][2];
  uint32_t oc[4];  // the lane's 2x2 outputs (first of its channels; ct adds 64 B)
  auto epilogue_fetch = [&](bool with_res, const float *bnext) {
    const int ln = fresh(lane), tile_ = 16 * tg + (ln & 15), kq_ = ln >> 4, ty_ = tile_ / TR, tx_ = tile_ - ty_ * TR;
    const int bo = 32 * chalf + 4 * kq_;  // first of the lane's channels (ct adds 16)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 2 * ty_ + (i >> 1), c = 2 * tx_ + (i & 1);
      oc[i] = act0 + (uint32_t)((tile_ < NT && r < N && c < N ? wb * BOARD_BYTES + Q::cell_off(r, c) : DUMP_OFF) + bo * 4);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x4 bv = bnext != nullptr ? *(const f32x4 *)(bnext + bo + 16 * ct) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        epre[o][ct] = bv;
        if (with_res) {
          const uint32_t a = oc[o] + (uint32_t)(ct * 64);
          const f32x2 x0 = lds_r2(a), x1 = lds_r2(a + 8);
          epre[o][ct] = bv + (f32x4){x0[0], x0[1], x1[0], x1[1]};
        }
      }
    }
  };
  auto epilogue = [&]() {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 s0[4], s1[4], y[4];  // y: [output 2 a + b] (4 channels r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s0[q] = (acc[q][ct] + acc[4 + q][ct]) + acc[8 + q][ct];
        s1[q] = (acc[4 + q][ct] - acc[8 + q][ct]) - acc[12 + q][ct];
      }
      y[0] = (s0[0] + s0[1]) + s0[2];
      y[1] = (s0[1] - s0[2]) - s0[3];
      y[2] = (s1[0] + s1[1]) + s1[2];
      y[3] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const uint32_t a = oc[o] + (uint32_t)(ct * 64);
        lds_w2(a, (f32x2){fmaxf(y[o][0], 0.0f), fmaxf(y[o][1], 0.0f)});
        lds_w2(a + 8, (f32x2){fmaxf(y[o][2], 0.0f), fmaxf(y[o][3], 0.0f)});
      }
    }
    preload(epre);
  };

#if AZ_W_STAMP
  int it = 0;
#endif
  for (;;) {
#if AZ_W_STAMP
    const bool stamp_on = blockIdx.x == 7 && it == 6;
    ++it;
#endif
    WSTAMP(0);
    if (tid == 0) {  // this ticket: the one drawn ahead, or -- when none of its leaves needs the network (flags off) -- the next that does
      int tk = slots[1], fl = slots[2];
      const int NG = (G + BPW - 1) / BPW;
      while (tk < NG && flags != nullptr) {
        if (fl < 0) {
          fl = 0;
#pragma unroll
          for (int b = 0; b < BPW; ++b) fl |= tk * BPW + b < G && flags[tk * BPW + b] != 0 ? 1 << b : 0;
        }
        if (fl != 0) break;
        tk = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
        fl = -1;
      }
      slots[0] = tk;
#pragma unroll
      for (int b = 0; b < BPW; ++b) slots[4 + b] = tk < NG && tk * BPW + b < G && (flags == nullptr || ((fl >> b) & 1) != 0) ? 1 : 0;
    }
    __syncthreads();
    const int gt = slots[0];
    if (gt >= (G + BPW - 1) / BPW) break;
    const int g = gt * BPW;  // the ticket's first board
    WSTAMP(1);
    unsigned int drawn = 0;
    if (tid == 0) drawn = atomicAdd(&tickets[0], 1u);  // the next ticket (lands while the input is expanded)
    if (x == nullptr) {
      if (__builtin_amdgcn_readfirstlane((int)(slots[3] == gt)) != 0) {
        rsel ^= 1;  // fetched ahead, under the ticket before (below): it sits in the other buffer
      } else {
        if (tid < BPW * RGS) {
          const int b = BPW == 1 ? 0 : tid / RGS, i = tid - b * RGS;
          if (slots[4 + b] != 0) rgb[rsel * BPW * RGS + tid] = i < RGW ? ring[(size_t)(g + b) * RGW + i] : (uint32_t)leafrec[g + b].w;
        }
        __syncthreads();
      }
    }
    const uint32_t *rgc = rgb + rsel * BPW * RGS;
    if (tid < BPW * NV) {
      const int b = BPW == 1 ? 0 : tid / NV, cell = tid - b * NV, row = cell / N, col = cell - row * N;
      if (slots[4 + b] != 0) {  // (a board that is not evaluated keeps whatever its region holds: finite values, results discarded)
        const uint32_t *rg = rgc + b * RGS;
        float c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = 0.0f;
        if (x != nullptr) {
          const float *xr = x + (size_t)(g + b) * 9 * NV + cell;
#pragma unroll
          for (int cc = 0; cc < 9; ++cc) c[cc] = xr[cc * NV];
        } else {
          const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            c[k] = (float)(int)((rg[k * SWB + wi] >> sh) & 1u) - (float)(int)((rg[k * SWB + WC + wi] >> sh) & 1u);
          c[8] = (int)rg[RGW] > 0 ? 1.0f : -1.0f;
        }
        const uint32_t a = act0 + (uint32_t)(b * BOARD_BYTES + Q::cell_off(row, col));
#pragma unroll
        for (int k = 0; k < 8; ++k) lds_w2(a + 8 * k, (f32x2){c[2 * k], c[2 * k + 1]});
      }
    }
    // The NEXT ticket is looked up while this one is computed, in steps that are never waited for: its number is known here; the
    // flag bytes of its boards (one aligned word) travel into slots[6] by LDS-DMA under the stem (the conv loop's own vmcnt waits
    // cover them: requests return in order); after the stem the ticket to fetch is published; after the first block's first conv
    // its boards' packed input is sent into the input buffer this ticket does not use, again by LDS-DMA.  (The copies are issued
    // from inline assembly like the weight stream's: one more in front of a k-step's two only makes that k-step's vmcnt(2) stricter.)
    if (tid == 0) {
      const int nx = (int)(gridDim.x + drawn);
      slots[1] = nx;
      slots[2] = -1;
      slots[7] = -1;
      if (nx < (G + BPW - 1) / BPW && flags != nullptr && x == nullptr && L > 0)
        glds4(flags, (uint32_t)(nx * BPW) & ~3u, lds0 + (uint32_t)(Q::OFF_SL + 6 * 4));
    }
    {  // stem: accumulators start at its bias
      f32x4 pre[4][2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 bv = *(const f32x4 *)(bl + 32 * chalf + 4 * (fresh(lane) >> 4) + 16 * ct);
#pragma unroll
        for (int o = 0; o < 4; ++o) pre[o][ct] = bv;
      }
      preload(pre);
    }
    // (the first stage's barrier orders the input planes before the patch reads)
    WSTAMP(3);
    conv_w<2, RSB>(lds0, pb, abase, u0, L > 0 ? ul : u0, wave, lane, acc);
    WSTAMP(4);
    epilogue_fetch(false, L > 0 ? bl + C : nullptr);
    __syncthreads();  // every wave has read the layer's input: the outputs may overwrite it
    WSTAMP(5);
    epilogue();
    WSTAMP(6);
    if (tid == 0 && L > 0 && x == nullptr) {  // the ticket to fetch ahead: the next one when any of its leaves needs the network (else it is resolved at the top)
      const int nx = slots[1];
      int mask = 0;
      if (nx < (G + BPW - 1) / BPW) {
        const int w = *(volatile int *)(slots + 6), sh = 8 * ((nx * BPW) & 3);
#pragma unroll
        for (int b = 0; b < BPW; ++b)
          if (nx * BPW + b < G && (flags == nullptr || ((w >> (sh + 8 * b)) & 0xFF) != 0)) mask |= 1 << b;
        if (flags != nullptr) slots[2] = mask;
      }
      slots[8] = mask;
      slots[7] = mask != 0 ? nx : -1;
    }
    for (int l = 0; l < L; l += 2) {
      const char *w1 = ul + (size_t)l * 8 * STAGE_BYTES, *w2 = w1 + (size_t)8 * STAGE_BYTES;
      conv_w<8, RSB>(lds0, pb, abase, w1, w2, wave, lane, acc);
      WSTAMP(7 + 3 * l);
      if (l == 0 && x == nullptr) {  // (slots[7], slots[8] were written before this conv's barriers)
        const int pf = __builtin_amdgcn_readfirstlane(slots[7]);
        if (pf >= 0 && tid < BPW * RGS) {  // word `tid` of the other input buffer: board b's ring word i / its player word
          const int b = BPW == 1 ? 0 : tid / RGS, i = tid - b * RGS;
          const uint32_t dst = lds0 + (uint32_t)(Q::OFF_RG + ((rsel ^ 1) * BPW * RGS + wave * 64) * 4);
          if (((slots[8] >> b) & 1) != 0) {
            if (i < RGW) glds4(ring, (uint32_t)((pf * BPW + b) * RGW + i) * 4u, dst);
            else glds4(leafrec, (uint32_t)(pf * BPW + b) * 16u + 12u, dst);
          }
          if (tid == 0) slots[3] = pf;
        }
      }
      epilogue_fetch(true, bl + (2 + l) * C);
      __syncthreads();
      WSTAMP(8 + 3 * l);
      epilogue();
      WSTAMP(9 + 3 * l);
      conv_w<8, RSB>(lds0, pb, abase, w2, l + 2 < L ? w2 + (size_t)8 * STAGE_BYTES : u0, wave, lane, acc);
      WSTAMP(10 + 3 * l);
      epilogue_fetch(false, l + 2 < L ? bl + (3 + l) * C : nullptr);
      __syncthreads();
      WSTAMP(11 + 3 * l);
      epilogue();
      WSTAMP(12 + 3 * l);
    }
    __syncthreads();
    WSTAMP(28);
    // ---- heads: the three 1x1 convs + ReLU from the finished activations: a thread takes a cell and half of its channels
    if (tid < BPW * 2 * NV) {
      const int b = BPW == 1 ? 0 : tid / (2 * NV), e = tid - b * 2 * NV;
      const int half = e >= NV ? 1 : 0, cell = e - half * NV, row = cell / N, col = cell - row * N;
      const uint32_t a = act0 + (uint32_t)(b * BOARD_BYTES + Q::cell_off(row, col) + half * 128);
      const float *w = hw + half * 32;
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const f32x2 v = lds_r2(a + 8 * k);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s0 = fmaf(v[j], w[2 * k + j], s0);
          s1 = fmaf(v[j], w[C + 2 * k + j], s1);
          s2 = fmaf(v[j], w[2 * C + 2 * k + j], s2);
        }
      }
      float *hb = hp + b * 6 * NV;
      hb[(half * 3 + 0) * NV + cell] = s0;
      hb[(half * 3 + 1) * NV + cell] = s1;
      hb[(half * 3 + 2) * NV + cell] = s2;
    }
    WSTAMP(29);
    __syncthreads();
    WSTAMP(30);
    // feature rows [policy NV | pad to 16 | value0 NV | value1 NV | pad]: whole-wave consecutive 4-byte stores
    constexpr int KP = (NV + 15) / 16 * 16;
    for (int e0 = tid; e0 < BPW * 3 * NV; e0 += NTH) {
      const int b = BPW == 1 ? 0 : e0 / (3 * NV), e = e0 - b * 3 * NV;
      const int k = e / NV, cell = e - k * NV;
      if (slots[4 + b] == 0) continue;
      const float *hb = hp + b * 6 * NV;
      const float v = hb[k * NV + cell] + hb[(3 + k) * NV + cell];
      feat[(size_t)(g + b) * FS + (k == 0 ? 0 : KP + (k - 1) * NV) + cell] = fmaxf(v + hw[3 * C + k], 0.0f);
    }
    // (the next board's first barriers order these reads of hp / act before anything overwrites them)
    WSTAMP(31);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage prefetched last must have landed before the LDS is given back
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

}  // namespace aztw

#if AZ_W_STAMP
extern "C" int az_debug_w_stamps(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(aztw::g_w_stamps), sizeof(unsigned long long) * 8 * 32) == hipSuccess ? 0 : -3;
}
#endif

static std::mutex g_w_mutex;
static unsigned int g_w_attr_done[64];

extern "C" int az_tower_forward_f32w(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                                     const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *u0_dev,
                                     const void *ul_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev,
                                     float *feat_f32_dev, int32_t feat_stride, void *stream) {
  using namespace aztw;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !u0_dev || !bias_dev || !headw_dev || !headb_dev || !feat_f32_dev ||
      (n_conv_layers > 0 && !ul_dev)) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1) || n_conv_layers > MAX_LAYERS) {
    tower_set_error("Winograd f32 tower: n_boards >= 1, an even number of conv layers, at most 36");
    return AZ_EINVAL;
  }
  if ((board_size != 15 && board_size != 9) || channels != 64) { tower_set_error("Winograd f32 tower: 9x9 or 15x15 boards, 64 channels"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { tower_set_error("feat_stride below az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
  // what the kernel's addressing assumes: 32-bit byte offsets into the packed batch (ring records of up to 516 bytes), the weight
  // stream and the leaf records moved 16 bytes at a time, and the look-ahead's flag read = the ALIGNED word that holds the next
  // ticket's flag bytes (an aligned word never leaves the page its valid byte lies in; an unaligned one could)
  if (n_boards > (1 << 22)) { tower_set_error("Winograd f32 tower: at most 4,194,304 boards per launch"); return AZ_EINVAL; }
  auto misaligned = [](const void *p, uintptr_t a) { return p != nullptr && ((uintptr_t)p & (a - 1)) != 0; };
  if (misaligned(x_f32_dev, 4) || misaligned(packed_ring_dev, 4) || misaligned(packed_leaf_dev, 16) || misaligned(flags_dev, 4) ||
      misaligned(u0_dev, 16) || misaligned(ul_dev, 16) || misaligned(bias_dev, 4) || misaligned(headw_dev, 4) || misaligned(headb_dev, 4) ||
      misaligned(feat_f32_dev, 4)) {
    tower_set_error("Winograd f32 tower: misaligned pointer (leaf records and transformed weights: 16 bytes; everything else, the flag bytes included: 4)");
    return AZ_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cur = 0;
  if (tower_device(s, &dev) != 0 || dev < 0 || dev >= 64) { tower_set_error("cannot tell the stream's device"); return AZ_EHIP; }
  if (hipGetDevice(&cur) != hipSuccess) { tower_set_error("hipGetDevice failed"); return AZ_EHIP; }
  struct DeviceGuard {
    int back;
    DeviceGuard(int want, int cur_) : back(want != cur_ ? cur_ : -1) { if (back >= 0) (void)hipSetDevice(want); }
    ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
  } guard(dev, cur);
  unsigned int *tickets = tower_tickets(dev, s);
  if (tickets == nullptr) return AZ_EHIP;
  const bool use_flags = !x_f32_dev && flags_dev;
  auto launch = [&](auto kern, int lds, int bpw, unsigned int variant) -> int {
    {
      std::lock_guard<std::mutex> lock(g_w_mutex);
      if (!(g_w_attr_done[dev] & variant)) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
          tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
          return AZ_EHIP;
        }
        g_w_attr_done[dev] |= variant;
      }
    }
    const int groups = (n_boards + bpw - 1) / bpw, grid = groups < 256 ? groups : 256;  // one workgroup per CU (its LDS: 125 - 141 KiB)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds, s, x_f32_dev, (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)u0_dev, (const char *)ul_dev, bias_dev,
                       n_conv_layers, headw_dev, headb_dev, feat_f32_dev, (int)feat_stride, tickets);
    return AZ_OK;
  };
  const int rc = board_size == 15 ? launch(k_tower_f32w<15>, Geo<15>::LDS_BYTES, Geo<15>::BPW, 1u)
                                  : launch(k_tower_f32w<9>, Geo<9>::LDS_BYTES, Geo<9>::BPW, 2u);
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_f32w launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}
