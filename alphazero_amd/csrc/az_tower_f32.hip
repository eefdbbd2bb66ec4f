// az_tower_f32.hip -- the residual conv tower and the heads' Linear layers in FLOAT32, for MI355X (gfx950).
//
// The reference evaluates its network in f32 (nn_models/base.py:17-19, core/MCTS_alphazero.py:161-162: no autocast), so
// this is the evaluator at the reference's own precision; az_tower.hip is the bf16 throughput version of the same tower.
// Same network family (examples/gomoku/model.py:41-90: conv3x3(9->64)+BN+ReLU, residual blocks of two conv3x3(64->64)+BN,
// the 1x1 head convs+BN+ReLU, then the heads' Linear layers), same structure of the kernel: one workgroup (4 waves) owns
// one board (the 32-cell-tile kernel: three 9x9 boards) for the whole tower, activations resident in LDS, every conv an implicit GEMM
//   D[co][cell] = sum_{tap, ci} W[co][tap][ci] * X[ci][cell + shift(tap)]
// on f32 MFMAs (f32 operands, f32 accumulate: bit for bit a k-ordered fmaf chain; dense peak 157.3 TFLOP/s = 1/16 of the
// bf16 rate, so the MFMA pipe -- not LDS, not the weight stream -- bounds these kernels, and ordinary VALU instructions do
// not execute under an f32 MFMA: every instruction that is not an MFMA is launch time).
//
// Two kernels:
//  * k_tower_f32_t16<CW, N> (15x15 boards with 64- or 128-channel trunks -- the benchmark's kernel -- and 9x9 boards with 64):
//    v_mfma_f32_16x16x4_f32 on tiles of 16 cells (225 = 14 tiles + one cell, 81 = 5 tiles + one cell: the odd cell on the VALU),
//    padded linear activation rows, weights from L2 into four rotating register sets three k-steps ahead, one barrier per
//    layer.  Described at its definition below.
//  * k_tower_f32<N, BPW> (towers deeper than T16_MAX_LAYERS conv layers, and the diagnostic AZ_TOWER_F32_TILE32=1; 9x9 boards
//    three per workgroup): v_mfma_f32_32x32x2_f32 on 32-cell tiles.  LDS: act[cell][64] f32, 256 B per cell, the sixteen 16-byte channel chunks of a
//    cell XOR-swizzled with (cell & 15) (conflict-free ds_read_b128; taps that fall off the board read a zero row).  One
//    ds_read_b128 feeds FOUR MFMAs: the instruction wants A[co][k = lane>>5] / B[k = lane>>5][cell], one float each; a lane
//    of half h reads channels 8m + 4h .. + 3 and MFMA t of k-step m takes element t, i.e. the k pair {8m + t, 8m + 4 + t} --
//    any pairing works as long as weights and activations agree.  Weights [tap][k-step][half][co 64][4 ci] f32 streamed 4
//    k-steps (8 KiB) at a time by LDS-DMA, double buffered, under the MFMAs; 2 cell tiles x 2 channel tiles of 32x32 per
//    wave; in-place layers; the block's residual in registers.
// Boards are handed out by the library's ticket counter (az_tower.hip); the input is the engine's packed leaf batch
// (8 bitboards + player) or dense f32 planes.  C ABI at the bottom; declared in include/az_engine.h.
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

// (as in az_tower.hip) a launch belongs to the stream's GPU, whatever device the calling thread has current
namespace {
struct StreamDeviceGuard {
  int dev = -1, back = -1;
  bool ok = false;
  explicit StreamDeviceGuard(hipStream_t s) {
    int cur = 0;
    if (tower_device(s, &dev) != 0 || dev < 0 || dev >= 64 || hipGetDevice(&cur) != hipSuccess) return;
    if (dev != cur) { if (hipSetDevice(dev) != hipSuccess) return; back = cur; }
    ok = true;
  }
  ~StreamDeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
};
}  // namespace

namespace aztf32 {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int C = 64;               // trunk width
constexpr int ROW = 4 * C;          // bytes per cell
constexpr int KSB = 2 * C * 16;     // bytes of one k-step of weights: [half 2][co 64][4 ci] f32 = 2 KiB
constexpr int KPS = 4;              // k-steps per stage of a 64 -> 64 layer (8 KiB; 18 stages per layer)
constexpr int KPS0 = 2;             // ... of the stem (9 channels padded to 16 = 2 k-steps per tap: one tap per stage)
constexpr int STAGE_BYTES = KPS * KSB;
constexpr int NPT = 2;              // 32-cell tiles per wave

template <int N, int BPW>
struct GeoF {
  static constexpr int NV = N * N, NVT = BPW * NV;
  static constexpr int ZERO_OFF = NVT * ROW;
  static constexpr int ACT_BYTES = NVT * ROW + ROW;
  static constexpr int HW_BYTES = ((3 * C + 3) * 4 + 15) / 16 * 16;
  static constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = BPW * 8 * SWB;  // packed input words per group
  static constexpr int RG_BYTES = ((RGW + 2 * BPW + 1) * 4 + 15) / 16 * 16;     // ring words, BPW players, BPW slots, the ticket
  static_assert(NVT <= 4 * NPT * 32, "4 waves x 2 tiles of 32 cells");
  static constexpr size_t lds_bytes() { return (size_t)ACT_BYTES + 2 * STAGE_BYTES + HW_BYTES + RG_BYTES; }
};

__device__ __forceinline__ int act_addr(int p, int chunk) { return p * ROW + ((chunk ^ (p & 15)) << 4); }

// Where input channel c (0..8) of the stem sits among the 16 padded channel positions, in the activations and in w0 alike:
// positions 4 q + t with t < 3 only (0,1,2, 4,5,6, 8,9,10), so that element 3 of every 16-byte chunk is zero and the 16-cell-tile
// kernel's stem issues three MFMAs per fragment instead of four (a quarter of the stem's MFMA time, 1.5 % of a two-block tower;
// the dropped terms are exact zeros; the placement only fixes the ORDER in which the nine channels of a tap are summed, the
// same in both kernels of this file).
__host__ __device__ constexpr int stem_pos(int c) { return c + c / 3; }

// async global -> LDS copy of 16 B per lane: the LDS destination is the wave-uniform `lds_base` + lane * 16
__device__ __forceinline__ void glds16(const char *g_lane, char *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                   (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)lds_base, 16, 0, 0);
}

// `bytes` (a multiple of 4 KiB) of weights, contiguous in global memory and in the LDS stage alike: 1 KiB per
// wave-instruction, wave w moves pieces w, w + 4, ...
template <int BYTES>
__device__ __forceinline__ void stage_w(const char *src, char *dst, int wave, int lane) {
  static_assert(BYTES % 4096 == 0, "whole rounds of the four waves");
#pragma unroll
  for (int p = 0; p < BYTES / 4096; ++p) glds16(src + (p * 4 + wave) * 1024 + lane * 16, dst + (p * 4 + wave) * 1024);
}

// One conv layer: acc[pt][ct] += W * X.  KS = k-steps (8 input channels each) per tap, KPSL = k-steps per stage.
// On entry stage 0 of THIS layer is already in buffer `buf` (landed and barriered); during the last stage the first stage
// of the NEXT layer (wnext, NEXT_BYTES; nullptr = none) is prefetched.  Every stage ends with a barrier, so after the
// last one every wave has finished reading act and the epilogue may overwrite it.  The k-steps of a stage are software
// pipelined through two register sets (the four fragment reads of step i + 1 are issued before the 16 MFMAs of step i).
template <int KS, int KPSL, int NEXT_BYTES, int N, int ZERO_OFF>
__device__ __forceinline__ void conv_layer(const char *act, char *wbuf, int &buf, const char *wg, const char *wnext, int wave,
                                           int lane, const int (&pp)[NPT], const uint32_t (&vm)[NPT], f32x16 (&acc)[NPT][2]) {
  static_assert(KS % KPSL == 0, "a stage never straddles two taps");
  constexpr int SPT = KS / KPSL, NSTAGE = 9 * SPT, SB = KPSL * KSB;
  const int h = lane >> 5, r = lane & 31;
  const uint32_t act32 = (uint32_t)(uintptr_t)act;
#pragma unroll 1
  for (int sg = 0; sg < NSTAGE; ++sg) {
    const int tap = sg / SPT, ks0 = (sg % SPT) * KPSL;
    const int d = (tap / 3 - 1) * N + (tap % 3 - 1);
    uint32_t bb[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      const uint32_t q = (uint32_t)(pp[pt] + d);
      const uint32_t in = (q << 8) | (((q & 15u) ^ (uint32_t)h) << 4);
      bb[pt] = ((vm[pt] >> tap) & 1u) ? in : (uint32_t)(ZERO_OFF | (h << 4));
    }
    const uint32_t wa32 = (uint32_t)(uintptr_t)(wbuf + ((buf + sg) & 1) * STAGE_BYTES + h * (C * 16) + r * 16);
    f32x4 fa[2][2], fb[2][NPT];
    auto load = [&](int i, int s) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[s][0]) : "v"(wa32), "i"(i * KSB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[s][1]) : "v"(wa32), "i"(i * KSB + 512));
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[s][pt]) : "v"(act32 + (bb[pt] ^ (uint32_t)((ks0 + i) << 5))));
    };
    load(0, 0);
#pragma unroll
    for (int i = 0; i < KPSL; ++i) {
      if (i + 1 < KPSL) {
        load(i + 1, (i + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // step i's four reads have returned
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
          acc[pt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i & 1][0][t], fb[i & 1][pt][t], acc[pt][0], 0, 0, 0);
          acc[pt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i & 1][1][t], fb[i & 1][pt][t], acc[pt][1], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 0) {  // the DMA of the next stage's weights: issued under this stage's MFMAs
        char *wn = wbuf + ((buf + sg + 1) & 1) * STAGE_BYTES;
        if (sg < NSTAGE - 1) stage_w<SB>(wg + (size_t)(sg + 1) * SB, wn, wave, lane);
        else if (wnext != nullptr) stage_w<NEXT_BYTES>(wnext, wn, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next stage has landed
    __syncthreads();
  }
  buf = (buf + NSTAGE) & 1;
}

// acc[pt][ct][4q + j] = channel ct*32 + 8q + 4h + j of cell pp[pt]  (C/D map of the 32x32 MFMAs: row = (reg & 3) + 8 (reg >> 2) + 4 h)
template <bool RES>
__device__ __forceinline__ void init_acc(const float *__restrict__ bias, int lane, f32x16 (&acc)[NPT][2], const f32x4 (&res)[NPT][2][4]) {
  const int h = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4 *)(bias + ct * 32 + q * 8 + h * 4);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[pt][ct][4 * q + j] = RES ? b[j] + res[pt][ct][q][j] : b[j];
    }
}

__device__ __forceinline__ void epilogue(char *act, int lane, const int (&pp)[NPT], const bool (&ok)[NPT], const f32x16 (&acc)[NPT][2]) {
  const int h = lane >> 5;
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[pt][ct][4 * q + j], 0.0f);
        if (ok[pt]) *(f32x4 *)(act + act_addr(pp[pt], ct * 8 + 2 * q + h)) = v;
      }
}

// x: (G, 9, N, N) f32, or nullptr to read the engine's packed leaf batch (ring u32 (G,8,SWB) + leaf records).
// w0: [9 taps][2][2][64][4] f32 (9 input channels padded to 16).  wl: [L][9][8][2][64][4] f32.  bias: [1+L][64] f32.
// headw: [3][64] f32 (policy, value0, value1), headb: [3].  feat: (G, FS) f32 = the heads' Linear inputs,
// row layout [policy NV | pad to 16 | value0 NV | value1 NV | pad to 16] (pads are never written).
template <int N, int BPW>
__global__ __launch_bounds__(256, 2) void k_tower_f32(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                      const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                      const char *__restrict__ w0, const char *__restrict__ wl,
                                                      const float *__restrict__ bias, int L, const float *__restrict__ headw,
                                                      const float *__restrict__ headb, float *__restrict__ feat, int FS,
                                                      unsigned int *tickets, int /*BL: k_tower_f32_t16<.., true> only*/, float * /*stash*/) {
  using Q = GeoF<N, BPW>;
  constexpr int NV = Q::NV, NVT = Q::NVT, WC = Q::WC, SWB = Q::SWB, RGW = Q::RGW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *act = smem;
  char *wbuf = smem + Q::ACT_BYTES;
  float *hw = (float *)(wbuf + 2 * STAGE_BYTES);
  uint32_t *rg = (uint32_t *)((char *)hw + Q::HW_BYTES);  // RGW ring words, BPW players, BPW slots, the ticket
  int *slots = (int *)rg + RGW + BPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < Q::ACT_BYTES / 16; i += 256) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // incl. the zero row
  for (int i = tid; i < 3 * C + 3; i += 256) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];

  int pp[NPT];
  bool ok[NPT];
  uint32_t vm[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int m = (NPT * wave + pt) * 32 + (lane & 31);
    ok[pt] = m < NVT;
    pp[pt] = ok[pt] ? m : 0;
    const int cell = pp[pt] % NV, row = cell / N, col = cell % N;
    uint32_t mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int rr = row + t / 3 - 1, cc = col + t % 3 - 1;
      if (rr >= 0 && rr < N && cc >= 0 && cc < N) mask |= 1u << t;
    }
    vm[pt] = ok[pt] ? mask : 0u;
  }
  const int ngroups = (G + BPW - 1) / BPW;
  int buf = 0;
  stage_w<KPS0 * KSB>(w0, wbuf, wave, lane);  // the stem's first stage (kept in flight across the first ticket draw)
  int ticket = (int)blockIdx.x;                // the first gridDim.x groups go by block index, the counter hands out the rest

  for (;;) {
    // ---- this workgroup's next group of BPW consecutive slots; in flag mode only the flagged ones are evaluated
    __syncthreads();  // (everybody is done with rg / slots of the previous group)
    if (tid == 0) {
      for (;;) {
        bool any = false;
        if (ticket < ngroups) {
          for (int b = 0; b < BPW; ++b) {
            const int g = ticket * BPW + b;
            const bool on = g < G && (flags == nullptr || flags[g] != 0);
            slots[b] = on ? g : -1;
            any |= on;
          }
        }
        if (any || ticket >= ngroups) break;
        ticket = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // a group of terminal leaves: one more draw
      }
      slots[BPW] = ticket;
    }
    __syncthreads();
    if (slots[BPW] >= ngroups) break;
    if (tid == 0) ticket = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // the next group's ticket: arrives under this group's tower
    if (x == nullptr) {
      for (int i = tid; i < RGW + BPW; i += 256) {
        const int b = i < RGW ? i / (8 * SWB) : i - RGW, g = slots[b];
        uint32_t v = 0;
        if (g >= 0) v = i < RGW ? ring[(size_t)g * 8 * SWB + i % (8 * SWB)] : (uint32_t)leafrec[g].w;
        rg[i] = v;
      }
      __syncthreads();
    }
    // ---- input -> act[cell][0..15] (channels 9..15 zero)
    for (int m = tid; m < NVT; m += 256) {
      const int b = m / NV, cell = m % NV, g = slots[b];
      float c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = 0.0f;
      if (g >= 0) {
        if (x != nullptr) {
          const float *xr = x + (size_t)g * 9 * NV + cell;
#pragma unroll
          for (int ch = 0; ch < 9; ++ch) c[stem_pos(ch)] = xr[ch * NV];
        } else {
          const uint32_t *rb = rg + b * 8 * SWB;
          const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            c[stem_pos(k)] = (float)(int)((rb[k * SWB + wi] >> sh) & 1u) - (float)(int)((rb[k * SWB + WC + wi] >> sh) & 1u);
          c[stem_pos(8)] = (int)rg[RGW + b] > 0 ? 1.0f : -1.0f;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *(f32x4 *)(act + act_addr(m, k)) = (f32x4){c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first group: the stem's first weight stage)
    __syncthreads();

    f32x16 acc[NPT][2];
    f32x4 res[NPT][2][4];
    // ---- stem
    init_acc<false>(bias, lane, acc, res);
    if (L > 0) conv_layer<2, KPS0, KPS * KSB, N, Q::ZERO_OFF>(act, wbuf, buf, w0, wl, wave, lane, pp, vm, acc);
    else conv_layer<2, KPS0, KPS0 * KSB, N, Q::ZERO_OFF>(act, wbuf, buf, w0, w0, wave, lane, pp, vm, acc);
    if (L > 0) {
      epilogue(act, lane, pp, ok, acc);
      __syncthreads();
    }
    // ---- residual blocks
    for (int l = 0; l < L; l += 2) {
      {
        const int h = lane >> 5;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) res[pt][ct][q] = *(const f32x4 *)(act + act_addr(pp[pt], ct * 8 + 2 * q + h));
      }
      init_acc<false>(bias + (1 + l) * C, lane, acc, res);
      conv_layer<8, KPS, KPS * KSB, N, Q::ZERO_OFF>(act, wbuf, buf, wl + (size_t)l * 9 * 8 * KSB, wl + (size_t)(l + 1) * 9 * 8 * KSB, wave, lane,
                                                    pp, vm, acc);
      epilogue(act, lane, pp, ok, acc);
      __syncthreads();
      init_acc<true>(bias + (2 + l) * C, lane, acc, res);
      if (l + 2 < L) {
        conv_layer<8, KPS, KPS * KSB, N, Q::ZERO_OFF>(act, wbuf, buf, wl + (size_t)(l + 1) * 9 * 8 * KSB, wl + (size_t)(l + 2) * 9 * 8 * KSB, wave,
                                                      lane, pp, vm, acc);
        epilogue(act, lane, pp, ok, acc);
        __syncthreads();
      } else {  // the tower's last layer: its output stays in the accumulators for the heads; prefetch the next board's stem
        conv_layer<8, KPS, KPS0 * KSB, N, Q::ZERO_OFF>(act, wbuf, buf, wl + (size_t)(l + 1) * 9 * 8 * KSB, w0, wave, lane, pp, vm, acc);
      }
    }
    // ---- heads: the three 1x1 convs (+ folded BN) + ReLU from the last layer's accumulators (a lane holds 32 of its cell's 64
    // channels, the other 32 sit in lane ^ 32), then the f32 rows of the heads' Linear inputs
    {
      const int h = lane >> 5;
      float hs[NPT][3];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) hs[pt][0] = hs[pt][1] = hs[pt][2] = 0.0f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 wv[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) wv[k] = *(const f32x4 *)(hw + k * C + ct * 32 + q * 8 + h * 4);
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = fmaxf(acc[pt][ct][4 * q + j], 0.0f);
#pragma unroll
              for (int k = 0; k < 3; ++k) hs[pt][k] = fmaf(a, wv[k][j], hs[pt][k]);
            }
        }
      constexpr int KP = (NV + 15) / 16 * 16;
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        float t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // lower half's sum + upper half's, in that order on both halves
          const float other = __shfl_xor(hs[pt][k], 32, 64);
          t[k] = h == 0 ? hs[pt][k] + other : other + hs[pt][k];
        }
        if ((pt & 1) != h || !ok[pt]) continue;  // even tiles are written by the lower half-wave, odd ones by the upper
        const int cell = pp[pt] % NV, g = slots[pp[pt] / NV];
        if (g < 0) continue;
        float *fr = feat + (size_t)g * FS;
        fr[cell] = fmaxf(t[0] + hw[3 * C + 0], 0.0f);
        fr[KP + cell] = fmaxf(t[1] + hw[3 * C + 1], 0.0f);
        fr[KP + NV + cell] = fmaxf(t[2] + hw[3 * C + 2], 0.0f);
      }
    }
  }
  // past the end: report in; the last workgroup to do so re-arms the counters for the next launch
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 15x15 boards: the same tower on v_mfma_f32_16x16x4_f32 with 16-cell tiles.  225 cells = 14 tiles of 16 + ONE cell, where
// eight 32-cell tiles spend 12 % of the MFMA time on 31 padding cells.  Wave w owns cell tiles 7 (w & 1) .. + 6 and the 32
// output channels 32 (w >> 1) .. + 31: 7 x 2 accumulator tiles of 16 x 16 (56 registers), 14 per wave, all waves equal.
// Cell 224 (the corner, four taps on the board) is computed on the VALU: wave w takes its output channels 16 w .. + 15 in the
// MFMA's own lane layout (lane = channel x k quarter) -- exactly the rows its a0 weight fragment holds (WFrag) -- with one
// more ds_read_b128 and four fmaf per k-step.
// 9x9 boards (T16Geo below): 81 cells = 5 tiles + cell 80; every wave takes all five cell tiles and ONE channel tile (wave w =
// channels 16 w .. + 15, its a0 fragment; no a1): 20 accumulator registers, 128 VGPRs, four workgroups per CU (measured:
// 0.80 ms per 4,096 boards with 4, 0.84 with 3, 0.82 with 2 workgroups per CU; the 32-cell-tile kernel: 0.91).
// The instruction takes A[co][k = lane >> 4] / B[k = lane >> 4][cell = lane & 15]; a lane of k-quarter kq reads the 16
// bytes of channels 16 m + 4 kq .. + 3 once and feeds four MFMAs (MFMA t: the k set {16 m + 4 kq' + t}).  Weights packed
// [tap][16-byte chunk of ci][co][4] f32, read straight from global memory (L2).
// Diagnostic build only (-DAZ_F32_STAMP=1): s_memtime stamps of one steady-state board of workgroup 7, per wave, into a
// buffer of their own (read back with az_debug_f32_stamps); the product build executes no stamp.
#ifndef AZ_F32_STAMP
#define AZ_F32_STAMP 0
#endif
#if AZ_F32_STAMP
static __device__ unsigned long long g_f32_stamps[4 * 32];
static __device__ unsigned long long g_f32_wg[1024 * 4];  // per workgroup: realtime start, end, boards, HW_ID
#define FSTAMP(k) do { if (stamp_on && lane == 0) g_f32_stamps[wave * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif
constexpr int T16_MAX_LAYERS = 40;  // bias rows the 16-cell-tile kernel keeps in LDS (deeper towers run on k_tower_f32)

// One layer.  The weight fragments are read straight from global memory (L2-resident: 0.6 MB for the whole 64-channel tower)
// into registers, three k-steps ahead of their MFMAs (four register sets in rotation), not through an LDS stage shared by
// the workgroup: a staged layer needs a workgroup barrier per stage (18 per layer) and LDS-DMA issues between the MFMAs;
// this way there is ONE barrier per layer (the in-place epilogue must wait for every wave's reads) and nothing but fragment
// reads between MFMAs.  The activation fragments run half a k-step ahead through two register groups that are re-used as
// soon as their MFMAs have been issued: group A = cell tiles 0..3 + the operand of cell 224, group B = cell tiles 4..6; with
// every k-step's reads exposed the MFMA pipe idled 20 % of the launch (the partner wave of the SIMD does not reliably cover
// them).
// A lane's weights of one k-step, for both 16-channel tiles of its wave's 32 channels.  ODD waves hold the tiles SWAPPED: a0 =
// channels 32 ch + 16 .. + 31, a1 = 32 ch .. + 15 (accumulator tile ct of wave w is channel tile 2 ch + (ct ^ (w & 1))
// everywhere in the kernel), so that a0's row is always channel 16 w + r -- the row cell 224's VALU sum needs: no third load.
struct WFrag { f32x4 a0, a1; };

// Weight requests: uniform base + the lane's 32-bit offset, ordinary loads whose waits the compiler counts.  (Tried: the same
// requests as inline asm in the scalar-base form with hand-counted `s_waitcnt vmcnt(6)` -- 51 instead of 59 VALU instructions
// per tap and no conservative wait at the loop header, but the register allocator copies three of the four a0 fragments at
// the end of every tap (loop-carried values), and a copy of a register whose load is still in flight reads stale data: the
// compiler must know about these loads.)
// Geometry of the 16-cell-tile kernel.  N*N = 16 k + 1 for both boards: 15x15 = 14 tiles + cell 224, 9x9 = 5 tiles + cell 80.
//   15x15: the cell tiles are split over two groups of waves (7 each), a wave owns 2 channel tiles (32 channels);
//   9x9:   every wave sees all 5 cell tiles and owns ONE channel tile (16 channels, wave w = channels 16 w .. + 15): 20
//          accumulator registers, so twice as many workgroups per CU cover each other's barriers and between-layer phases.
template <int CW, int N>
struct T16Geo {
  static constexpr int NV = N * N, LCELL = NV - 1, NSPLIT = N == 15 ? 2 : 1, NT = NV / 16 / NSPLIT, NCT = N == 15 ? 2 : 1;
  static constexpr int NA = (NT + 1) / 2, NB = NT - NA;             // cell tiles of activation register groups A and B
  static constexpr int NCG = CW / (16 * NCT), NW = NCG * NSPLIT;  // channel groups; waves
  static constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = 8 * SWB, KP = (NV + 15) / 16 * 16;
  static constexpr int WG_PER_CU = CW == 128 ? 1 : (N == 15 ? 2 : 4);
  static_assert(NV % 16 == 1 && (NV / 16) % NSPLIT == 0 && 9 * NT <= 64, "whole tiles + one cell; the tap mask fits 64 bits");
};

template <int CW, int NCT>
__device__ __forceinline__ void wload(WFrag &w, const char *wg, uint32_t woff, int cto, int kk) {
  constexpr int KSBW = 4 * CW * 16;  // bytes of one k-step of 16 input channels: [kq 4][co CW][4] f32
#ifdef AZ_F32_WSAME  // diagnostic build: every k-step reads k-step 0's weights (always cache hits; wrong results, timing only)
  kk = 0;
#endif
  // woff = the lane's offset in the wave pair's FIRST tile; which tile a0 / a1 take is uniform per wave (cto) and goes into
  // the scalar base: one 32-bit lane offset serves both loads
  const char *base = wg + (size_t)kk * KSBW;
  w.a0 = *(const f32x4 *)(base + cto * 256 + woff);
  if constexpr (NCT == 2) w.a1 = *(const f32x4 *)(base + (256 - cto * 256) + woff);
}

// an opaque copy: what is derived from it is computed where it is used, not hoisted out of the board loop and spilled
__device__ __forceinline__ int fresh(int v) { asm volatile("" : "+v"(v)); return v; }

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int CNT, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, CNT>{}); }
template <int OFF>
__device__ __forceinline__ void dsr128(f32x4 &d, uint32_t a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF)); }

// On entry wf[0..2] hold (or are about to receive) k-steps 0..2 of this layer; the weights of wnext's first k-steps are
// requested during its last three (see step below).  KS16 = k-steps per tap.  No barrier inside: the caller synchronises.
// The activation rows are PADDED, not swizzled (T16_ROW<CW> = 4 CW + 16 bytes per cell): a fragment address is then linear in
// cell, k-step and k-quarter, so the k-step is the instruction's immediate offset and a tap costs one base, an add and a
// select per cell tile -- no per-read VALU instruction at all (VALU does not overlap with f32 MFMAs: every one of them
// is launch time).  The price is one 2-way bank conflict per 16-lane group of a ds_read_b128 (cells c .. c+15 of k-quarter
// pairs land on 16-byte bank units (cell + kq + 4m) mod 16: the two quarters of a group overlap in one unit), on an LDS
// array that is otherwise 11 % busy.  The loop runs over taps; the k-steps of a tap are unrolled.
template <int CW> constexpr int T16_ROW = 4 * CW + 16;

template <int KS16, bool LEFT, int CW, int N>
__device__ __forceinline__ void conv_layer16g(const char *act, const char *wg, const char *wnext, int wave, int lane, const int pp0,
                                              const unsigned long long vm, f32x4 (&acc)[T16Geo<CW, N>::NT][T16Geo<CW, N>::NCT], float &lc,
                                              WFrag (&wf)[4]) {
  using Q = T16Geo<CW, N>;
  constexpr int ROWP = T16_ROW<CW>, LCELL = Q::LCELL, NT = Q::NT, NCT = Q::NCT, NA = Q::NA, NB = Q::NB;
  const int ln = fresh(lane);  // (per-lane addresses are formed here, per layer: hoisted 64-bit lane pointers were spilled across the loops)
  const int kq = ln >> 4, r = ln & 15, cg = Q::NSPLIT == 2 ? wave >> 1 : wave;
  const uint32_t act32 = (uint32_t)(uintptr_t)act;
  // first row of the wave's channel group + r; a0's row is channel 16 wave + r (tile cto of a pair of tiles), a1's the other tile
  const uint32_t woff = kq * (CW * 16) + (cg * (16 * NCT) + r) * 16;
  const int cto = NCT == 2 ? wave & 1 : 0;
  uint32_t bb[NT], lx32 = 0, tbase = 0, tmask = 0;
  const uint32_t zaddr = act32 + (uint32_t)(Q::NV * ROWP + (kq << 4));  // the zero row
  auto set_tap_A = [&](int tap) {  // (bit NT tap + pt of vm: that tap of cell pp0 + 16 pt is on the board)
    const int d = (tap / 3 - 1) * N + (tap % 3 - 1);
    tbase = act32 + (uint32_t)((pp0 + d) * ROWP + (kq << 4));  // ("negative" rows only occur masked, or with 16 pt rows added)
    tmask = (uint32_t)(vm >> (NT * tap));
#pragma unroll
    for (int pt = 0; pt < NA; ++pt) bb[pt] = (tmask & (1u << pt)) ? tbase + (uint32_t)(16 * pt * ROWP) : zaddr;
    // cell 224: taps with dr <= 0 and dc <= 0 are on the board for that corner cell; the others read the zero row
    lx32 = (tap / 3 <= 1 && tap % 3 <= 1) ? act32 + (uint32_t)((LCELL + d) * ROWP + (kq << 4)) : zaddr;
  };
  auto set_tap_B = [&]() {  // (after set_tap_A of the same tap)
#pragma unroll
    for (int pt = NA; pt < NT; ++pt) bb[pt] = (tmask & (1u << pt)) ? tbase + (uint32_t)(16 * pt * ROWP) : zaddr;
  };
  f32x4 fbA[NA], fbB[NB], lx;
  auto loadA = [&](auto mc) {  // NA + 1 LDS reads: cell tiles 0..NA-1 and the operand of the last cell
    constexpr int OFF = decltype(mc)::value * 64;
#pragma unroll
    for (int pt = 0; pt < NA; ++pt) dsr128<OFF>(fbA[pt], bb[pt]);
    dsr128<OFF>(lx, lx32);
  };
  auto loadB = [&](auto mc) {  // NB LDS reads: cell tiles NA..NT-1
    constexpr int OFF = decltype(mc)::value * 64;
#pragma unroll
    for (int pt = 0; pt < NB; ++pt) dsr128<OFF>(fbB[pt], bb[NA + pt]);
  };
  // One k-step.  M = k-step of the tap, S = the weight register set that holds its weights.  FOUR sets in rotation: at k-step k
  // the weights of k-step k + 3 are requested into set (S + 3) % 4, so a request has three k-steps (about 10 k cycles with the
  // partner wave on the pipe) to come back from L2 -- with one k-step of cover 5 % of the launch was exposed weight latency
  // (measured with a build that reads the same weights every k-step, -DAZ_F32_WSAME).  A layer of 9 KS16 = 36 or 72 k-steps
  // leaves the rotation where it found it and requests the next layer's k-steps 0..2 during its last three.  The stem has 9:
  // its last three k-steps request the next layer's k-steps 1..3 (into sets 1..3) and k-step 0 follows after its last MFMAs,
  // under the epilogue and the barriers between the layers.
  constexpr int KSTEPS = 9 * KS16, NEXT_SHIFT = KSTEPS % 4;
  static_assert(NEXT_SHIFT == 0 || NEXT_SHIFT == 1, "36, 72 or 9 k-steps");
  auto step = [&](int tap, auto mc, auto sc) {
    constexpr int M = decltype(mc)::value, S = decltype(sc)::value, V = (S + 3) % 4;
    constexpr bool TAP_END = M == KS16 - 1;
    constexpr int NTT = KS16 == 1 ? 3 : 4;  // the stem's fourth element is a zero channel (stem_pos): no MFMA for it
    const bool last = TAP_END && tap == 8;
    {
      const int kn = tap * KS16 + M + 3;
      const bool own = kn < KSTEPS;
      wload<CW, NCT>(wf[V], own ? wg : wnext, woff, cto, own ? kn : kn - KSTEPS + NEXT_SHIFT);
    }
    loadB(mc);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NB) : "memory");  // group A of this k-step has returned
    __builtin_amdgcn_sched_barrier(0);
    if (LEFT) {  // cell 224, output channel 16 wave + r: that row of the weights IS this lane's a0 (see WFrag)
#pragma unroll
      for (int t = 0; t < NTT; ++t) lc = fmaf(wf[S].a0[t], lx[t], lc);
      // (pins the sum HERE: left alone, the optimiser sinks the four k-steps' fmafs of a tap to the end of the loop body, which keeps
      // every weight set alive across the requests that refill it -- copies, and a wait for all of them, at the end of each tap)
      asm volatile("" : "+v"(lc));
    }
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int pt = 0; pt < NA; ++pt) {
        acc[pt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[S].a0[t], fbA[pt][t], acc[pt][0], 0, 0, 0);
        if constexpr (NCT == 2) acc[pt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[S].a1[t], fbA[pt][t], acc[pt][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP_END) {
      if (!last) {
        set_tap_A(tap + 1);
        loadA(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA + 1) : "memory");  // group B of this k-step has returned
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      loadA(std::integral_constant<int, M + 1>{});
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA + 1) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP_END) {
      if (!last) set_tap_B();
    }
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int pt = 0; pt < NB; ++pt) {
        acc[NA + pt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[S].a0[t], fbB[pt][t], acc[NA + pt][0], 0, 0, 0);
        if constexpr (NCT == 2) acc[NA + pt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[S].a1[t], fbB[pt][t], acc[NA + pt][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  set_tap_A(0);
  set_tap_B();
  loadA(std::integral_constant<int, 0>{});
  using I0 = std::integral_constant<int, 0>;
  if constexpr (KS16 == 1) {  // the stem: four taps (= four k-steps) per iteration, then tap 8
#pragma unroll 1
    for (int t4 = 0; t4 < 8; t4 += 4) static_for<4>([&](auto jc) { step(t4 + decltype(jc)::value, I0{}, jc); });
    step(8, I0{}, I0{});
    wload<CW, NCT>(wf[0], wnext, woff, cto, 0);
  } else {
    static_assert(KS16 % 4 == 0, "a tap leaves the four weight sets where it found them");
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap)
      static_for<KS16>([&](auto mc) { step(tap, mc, std::integral_constant<int, decltype(mc)::value % 4>{}); });
  }
}

// QK = true: the same kernel for the reference's QuickModel family (nn_models/quick_model.py:67-93, resnet_blocks.py:28-40):
// residual blocks of BL x (conv3x3 + BN + ReLU) whose sum x + block(x) is taken AFTER the last ReLU and is not activated again,
// and 3x3 head convs (2 policy channels, 1 value channel).  A block's input has to outlive BL in-place layers: the lane that
// overwrites an activation in the block's first epilogue sends the old value to `stash` (global memory, one private region per
// workgroup, each wave-level store 1 KiB contiguous; the same lane reads it back in the block's last epilogue: 0.9 MB per board
// of a 4 x 5-layer tower at 128 channels, L2 / Infinity Cache traffic) -- 56 more live registers per lane do not fit, a second
// activation buffer does not fit the LDS at 128 channels.  The head convs need every cell's neighbours, so the last layer is
// written to LDS like any other and the three head channels are summed on the VALU (0.06 % of the tower's flops) with the
// filters [3][9 taps][CW] read through the scalar cache (headw; wave-uniform addresses).
template <int CW, int N, bool QK = false>
__global__ __launch_bounds__((64 * T16Geo<CW, N>::NW), (T16Geo<CW, N>::WG_PER_CU == 1 ? 2 : T16Geo<CW, N>::WG_PER_CU)) void k_tower_f32_t16(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                          const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                          const char *__restrict__ w0, const char *__restrict__ wl,
                                                          const float *__restrict__ bias, int L, const float *__restrict__ headw,
                                                          const float *__restrict__ headb, float *__restrict__ feat, int FS,
                                                          unsigned int *tickets, int BL, float *__restrict__ stash) {
  using Q = T16Geo<CW, N>;
  constexpr int NV = Q::NV, WC = Q::WC, SWB = Q::SWB, RGW = Q::RGW, LCELL = Q::LCELL, KP = Q::KP, NT = Q::NT, NCT = Q::NCT;
  constexpr int C = CW, ROW = T16_ROW<CW>, NW = Q::NW, NTH = 64 * NW, NCQ = Q::NCG;  // padded row bytes; waves; channel groups
  constexpr int ACT_BYTES = NV * ROW + ROW, HW_BYTES = ((3 * C + 3) * 4 + 15) / 16 * 16, RGS = RGW + 4, RG_BYTES = (2 * RGS + 8) * 4;
  constexpr int KSB16 = 4 * C * 16;
  extern __shared__ __attribute__((aligned(1024))) char smem_t16[];  // (conv_layer16g XORs chunk bits into finished addresses)
  char *act = smem_t16;
  float *hw = (float *)(smem_t16 + ACT_BYTES);
  uint32_t *rgb = (uint32_t *)((char *)hw + HW_BYTES);  // two buffers of RGW ring words + the player word (RGS words apart)
  // {this board's slot, its ticket (G = past the end), the slot looked up ahead (-1: past the end, -2: not looked up), the next
  // ticket, that ticket's flag byte}: the look-ahead state lives HERE, not in registers of thread 0 that would have to survive
  // the conv loops (the kernel runs at its VGPR budget: they were spilled, a scratch store and a forced wait each per board)
  int *slots = (int *)rgb + 2 * RGS;
  float *hp = (float *)((char *)rgb + RG_BYTES);        // [NCQ channel groups][3 heads][NV cells]: partial head sums
  float *lh = hp + NCQ * 3 * NV;                        // [NW waves][3]: head sums of the last cell
  float *bl = hp + (NCQ * 3 * NV + 3 * NW + 3) / 4 * 4;  // [(1 + L)][C]: every layer's bias (L <= T16_MAX_LAYERS), 16-byte aligned
  auto act_addr = [](int p, int chunk) { return p * T16_ROW<CW> + (chunk << 4); };  // padded rows, no swizzle (conv_layer16g)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ch: the wave's channel group (NCT tiles of 16), chalf: its group of cell tiles; cto: accumulator tile ct holds channel tile NCT ch + (ct ^ cto)
  const int kq = lane >> 4, ch = Q::NSPLIT == 2 ? wave >> 1 : wave, chalf = Q::NSPLIT == 2 ? wave & 1 : 0, cto = NCT == 2 ? wave & 1 : 0;

  for (int i = tid; i < ACT_BYTES / 16; i += NTH) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // incl. the zero row
  for (int i = tid; i < 3 * C + 3; i += NTH) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];
  for (int i = tid; i < (1 + L) * C; i += NTH) bl[i] = bias[i];

  const int pp0 = NT * chalf * 16 + (lane & 15);  // this lane's cells: pp0 + 16 pt
  unsigned long long vm = 0;                       // bit NT tap + pt: that tap of cell pp0 + 16 pt is on the board
#pragma unroll
  for (int pt = 0; pt < NT; ++pt) {
    const int cell = pp0 + 16 * pt, row = cell / N, col = cell % N;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int rr = row + t / 3 - 1, cc = col + t % 3 - 1;
      if (rr >= 0 && rr < N && cc >= 0 && cc < N) vm |= 1ull << (NT * t + pt);
    }
  }
  WFrag wf[4];  // weight fragments of the running k-step and the three after it (global -> registers, three k-steps ahead)
  {
    const uint32_t woff0 = (uint32_t)(kq * (C * 16) + (ch * (16 * NCT) + (lane & 15)) * 16);
    wload<CW, NCT>(wf[0], w0, woff0, cto, 0);  // the stem's first three k-steps
    wload<CW, NCT>(wf[1], w0, woff0, cto, 1);
    wload<CW, NCT>(wf[2], w0, woff0, cto, 2);
  }
  // Boards are handed out by the ticket counter.  The NEXT board is looked up while the current one is computed, in steps
  // that are never waited for and hold no register across a conv loop: thread 0 draws the next ticket (a returning atomic)
  // after the stem's conv loop and stores it two barriers later, then sends the ticket's flag byte into LDS by LDS-DMA; after
  // the first block's first conv it publishes the slot (or -2: flagged off, look again at the top); after the next barrier
  // RGW + 1 lanes send the slot's packed input by LDS-DMA into the input buffer the running board does not use.
  if (tid == 0) { slots[2] = -2; slots[3] = (int)blockIdx.x; slots[4] = 0; }
  int rsel = 0;  // (uniform) which of the two input buffers the running board expands

#if AZ_F32_STAMP
  int it = 0;
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (;;) {
#if AZ_F32_STAMP
    const bool stamp_on = blockIdx.x == 7 && it == 6;
    ++it;
#endif
    FSTAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead's LDS-DMA (next input words) has landed
    __syncthreads();
    if (tid == 0) {
      int ps = slots[2], tk = slots[3];
      if (ps == -2) {  // not looked up ahead (first board, no residual block, or the ticket drawn ahead was a terminal leaf)
        for (;;) {
          const bool on = tk < G && (flags == nullptr || flags[tk] != 0);
          if (on || tk >= G) break;
          tk = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // a terminal leaf: one more draw
        }
        ps = tk < G ? tk : -1;
      }
      slots[0] = ps;
      slots[1] = ps < 0 ? G : ps;
    }
    __syncthreads();
    if (slots[1] >= G) break;
    FSTAMP(1);
    const int g = slots[0];
    if (x == nullptr) {
      if (__builtin_amdgcn_readfirstlane((int)(slots[2] == g)) != 0) {
        rsel ^= 1;  // the look-ahead fetched this board's input into the other buffer
      } else {
        uint32_t *rgw = rgb + rsel * RGS;
        for (int i = tid; i < RGW + 1; i += NTH) rgw[i] = i < RGW ? ring[(size_t)g * RGW + i] : (uint32_t)leafrec[g].w;
      }
      __syncthreads();
    }
    const uint32_t *rg = rgb + rsel * RGS;
    for (int cell = tid; cell < NV; cell += NTH) {
      float c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = 0.0f;
      if (x != nullptr) {
        const float *xr = x + (size_t)g * 9 * NV + cell;
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) c[stem_pos(cc)] = xr[cc * NV];
      } else {
        const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          c[stem_pos(k)] = (float)(int)((rg[k * SWB + wi] >> sh) & 1u) - (float)(int)((rg[k * SWB + WC + wi] >> sh) & 1u);
        c[stem_pos(8)] = (int)rg[RGW] > 0 ? 1.0f : -1.0f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *(f32x4 *)(act + act_addr(cell, k)) = (f32x4){c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]};
    }
    __syncthreads();
    FSTAMP(2);

    f32x4 acc[NT][NCT];
    float lc = 0.0f, lres = 0.0f;  // the last cell (224 / 80): output channel lco of this wave's 16, partial over this lane's k quarter
    // Everything the phases between the conv layers need per lane (cell, k quarter, channel of cell 224, shuffle partners) is
    // re-derived from a laundered copy of the lane index inside each phase: left to itself the compiler hoists ~40 such
    // values out of the board loop, spills them, and reloads them right after a conv loop -- behind the weight requests in flight.
    auto corner_ptr = [&](int ln) { const int c = wave * 16 + (ln & 15); return (float *)(act + act_addr(LCELL, c >> 2) + (c & 3) * 4); };
    // t + (t of lane ^ 16), t + (t of lane ^ 32): gfx950's row / half swaps instead of an LDS round trip (the two results of a swap
    // are {own, partner} in one order or the other; the sum is the same bits either way); sum over a 16-lane row by rotations
    // (after each step the values are symmetric under the bits already folded, so a rotation fetches what lane ^ off holds)
    auto add16 = [](float t) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };
    auto add32 = [](float t) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };
    auto rowsum = [](float t) {
      t = t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
      t = t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
      t = t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xF, 0xF, false));  // row_ror:2
      t = t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xF, 0xF, false));  // row_ror:1
      return t;
    };
    auto init = [&](const float *b) {
      const int kq = fresh(lane) >> 4;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const f32x4 bv = *(const f32x4 *)(b + (ch * NCT + (ct ^ cto)) * 16 + kq * 4);
#pragma unroll
        for (int pt = 0; pt < NT; ++pt) acc[pt][ct] = bv;
      }
      lc = 0.0f;
    };
    auto lfinal = [&](const float *b, bool with_res) {  // the four k quarters of cell 224, then bias (+ residual)
      const int ln = fresh(lane), lco = wave * 16 + (ln & 15);
      lc = add32(add16(lc));
      lc = (with_res ? b[lco] + lres : b[lco]) + lc;
    };
    auto epilogue = [&]() {
      const int ln = fresh(lane), kq = ln >> 4, pp0 = NT * chalf * 16 + (ln & 15);
#pragma unroll
      for (int pt = 0; pt < NT; ++pt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[pt][ct][j], 0.0f);
          *(f32x4 *)(act + act_addr((pp0 + 16 * pt), (ch * NCT + (ct ^ cto)) * 4 + kq)) = v;
        }
      if (kq == 0) *corner_ptr(ln) = fmaxf(lc, 0.0f);
    };
    // The epilogue of a block's FIRST conv, fused with the accumulator init of its second: a lane overwrites exactly the
    // activations it will need again as the block's residual, so it reads them here, just before it stores over them, and
    // starts the next accumulators at bias + residual -- the residual never sits in registers across a conv layer (56 VGPRs
    // that used to be live through the first conv; with them the kernel spilled and every reload waited behind the weight
    // requests in flight).  Same arithmetic as before: acc = (bias + x) first, then the MFMA chain.
    auto epilogue_res = [&](const float *bnext) {
      const int ln = fresh(lane), kq = ln >> 4, pp0 = NT * chalf * 16 + (ln & 15);
      float *const lptr = corner_ptr(ln);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const f32x4 bv = *(const f32x4 *)(bnext + (ch * NCT + (ct ^ cto)) * 16 + kq * 4);
#pragma unroll
        for (int pt = 0; pt < NT; ++pt) {
          f32x4 *const p = (f32x4 *)(act + act_addr((pp0 + 16 * pt), (ch * NCT + (ct ^ cto)) * 4 + kq));
          const f32x4 xr = *p;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[pt][ct][j], 0.0f);
          *p = v;
          acc[pt][ct] = bv + xr;
        }
      }
      lres = *lptr;  // (every lane of the four k quarters reads it before the quarter-0 lane stores)
      if (kq == 0) *lptr = fmaxf(lc, 0.0f);
      lc = 0.0f;
    };
    // ---- stem
    init(bl);
    conv_layer16g<1, true, CW, N>(act, w0, L > 0 ? wl : w0, wave, lane, pp0, vm, acc, lc, wf);  // (nine k-steps: the prefetch lands in wf[1])
    // the next ticket: drawn here (a returning atomic), stored after the two barriers below -- in a register across no conv loop
    unsigned int drawn = 0;
    if (tid == 0) drawn = atomicAdd(&tickets[0], 1u);
    lfinal(bl, false);
    FSTAMP(3);
    if (L > 0) {
      __syncthreads();  // every wave has read the layer's input: the outputs may overwrite it
      epilogue();
      __syncthreads();
    }
    FSTAMP(4);
    if (tid == 0) {
      const int tk = (int)(gridDim.x + drawn);
      slots[2] = -2;  // (every thread has compared it with this board's slot, three barriers ago)
      slots[3] = tk;
      slots[4] = 0;
      // the ticket's flag byte travels by LDS-DMA into slots[4]: read back after the first block's first conv, whose own
      // wait-counts (requests return in order) have by then covered it
      if (tk < G && flags != nullptr)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(flags + tk),
                                         (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(slots + 4), 1, 0, 0);
    }
    if constexpr (QK) {
      // ---- QuickModel blocks: every layer conv + bias + ReLU in place; first layer of a block stashes its input, last adds it back
      constexpr int STW = NT * NCT * 256 + 64;  // floats per wave in the stash: NT x NCT fragments of 64 lanes x 4, the corner cell's 16
      float *const st = stash + ((size_t)blockIdx.x * NW + wave) * STW;
      auto epilogue_q = [&](bool first, bool lastl) {
        const int ln = fresh(lane), kq = ln >> 4, pp0 = NT * chalf * 16 + (ln & 15);
        float *const lptr = corner_ptr(ln);
        float *const sl = st + ln * 4;
        if (lastl && !first) {  // the block's last layer: relu(conv) + the stashed block input (all requests first, then the sums)
          f32x4 xr[NT][NCT];
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) xr[pt][ct] = *(const f32x4 *)(sl + (pt * NCT + ct) * 256);
          const float lr = kq == 0 ? st[NT * NCT * 256 + ln] : 0.0f;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[pt][ct][j], 0.0f) + xr[pt][ct][j];
              *(f32x4 *)(act + act_addr((pp0 + 16 * pt), (ch * NCT + (ct ^ cto)) * 4 + kq)) = v;
            }
          if (kq == 0) *lptr = fmaxf(lc, 0.0f) + lr;
        } else {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
              f32x4 *const p = (f32x4 *)(act + act_addr((pp0 + 16 * pt), (ch * NCT + (ct ^ cto)) * 4 + kq));
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[pt][ct][j], 0.0f);
              if (first) {  // (this lane overwrites exactly the activations it stashes: no other lane needs them again)
                const f32x4 xo = *p;
                if (lastl) v = v + xo;  // a one-layer block
                else *(f32x4 *)(sl + (pt * NCT + ct) * 256) = xo;
              }
              *p = v;
            }
          if (kq == 0) {
            float v = fmaxf(lc, 0.0f);
            if (first) {
              const float xo = *lptr;
              if (lastl) v = v + xo;
              else st[NT * NCT * 256 + ln] = xo;
            }
            *lptr = v;
          }
        }
      };
      constexpr size_t LSTRIDE = (size_t)9 * (CW / 16) * KSB16;
      int j = 0;  // layer of its block
      for (int l = 0; l < L; ++l) {
        init(bl + (1 + l) * C);
        conv_layer16g<CW / 16, true, CW, N>(act, wl + (size_t)l * LSTRIDE, l + 1 < L ? wl + (size_t)(l + 1) * LSTRIDE : w0, wave, lane, pp0, vm, acc, lc, wf);
        lfinal(bl + (1 + l) * C, false);
        if (l == 0 && tid == 0) {  // publish the next board's slot
          const int tk = slots[3];
          slots[2] = tk >= G ? -1 : ((flags == nullptr || (slots[4] & 0xFF) != 0) ? tk : -2);
        }
        __syncthreads();  // every wave has read the layer's input
        epilogue_q(j == 0, j == BL - 1);
        constexpr int RWV = (RGW + 63) / 64;
        if (l == 0 && x == nullptr && tid <= 64 * RWV) {  // the next board's packed input by LDS-DMA (as in the other branch)
          const int gn = __builtin_amdgcn_readfirstlane(slots[2]);
          if (gn >= 0) {
            uint32_t *dst = rgb + (rsel ^ 1) * RGS;
            if (wave < RWV) {
              if (tid < RGW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ring + (size_t)gn * RGW + tid),
                                                 (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(dst + wave * 64), 4, 0, 0);
            } else if (tid == 64 * RWV)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)&leafrec[gn].w,
                                               (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(dst + RGW), 4, 0, 0);
          }
        }
        __syncthreads();
        j = j + 1 == BL ? 0 : j + 1;
      }
      // ---- heads: the three 3x3 head convs (+ folded BN) on the VALU from the finished activations.  A thread takes one cell and
      // CPS of its channels (splits are wave-uniform: the filters come through the scalar cache); partial sums per split in LDS
      {
        constexpr int CELLS_T = N == 9 ? 128 : 256, SPL = NTH / CELLS_T, CPS = CW / SPL, KPQ = (2 * NV + 15) / 16 * 16;
        static_assert(SPL >= 1 && SPL <= NCQ && CELLS_T >= NV, "a thread per cell and split; hp holds SPL x 3 x NV sums");
        const int cell = fresh(tid) & (CELLS_T - 1), split = wave / (CELLS_T / 64);
        if (cell < NV) {
          const int row = cell / N, col = cell - row * N;
          const float *wq = headw + split * CPS;
          float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll 1
          for (int tap = 0; tap < 9; ++tap) {
            const int dr = tap / 3 - 1, dc = tap - (tap / 3) * 3 - 1, rr = row + dr, cc = col + dc;
            const bool on = rr >= 0 && rr < N && cc >= 0 && cc < N;
            const char *ap = act + (on ? cell + dr * N + dc : NV) * ROW + split * (CPS * 4);
            const float *wt = wq + tap * CW;
#pragma unroll
            for (int c4 = 0; c4 < CPS / 4; ++c4) {
              const f32x4 a = *(const f32x4 *)(ap + c4 * 16);
              const f32x4 u0 = *(const f32x4 *)(wt + c4 * 4), u1 = *(const f32x4 *)(wt + 9 * CW + c4 * 4), u2 = *(const f32x4 *)(wt + 18 * CW + c4 * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s0 = fmaf(a[e], u0[e], s0);
                s1 = fmaf(a[e], u1[e], s1);
                s2 = fmaf(a[e], u2[e], s2);
              }
            }
          }
          hp[(split * 3 + 0) * NV + cell] = s0;
          hp[(split * 3 + 1) * NV + cell] = s1;
          hp[(split * 3 + 2) * NV + cell] = s2;
        }
        __syncthreads();
        // feature rows [policy0 NV | policy1 NV | pad to 16 | value NV | pad to 16] (Flatten of (2, N, N) and of (1, N, N))
        float *fr = feat + (size_t)g * FS;
        for (int e = tid; e < 3 * NV; e += NTH) {
          const int k = e / NV, cell2 = e - k * NV;
          float v = 0.0f;
#pragma unroll
          for (int c = 0; c < SPL; ++c) v += hp[(3 * c + k) * NV + cell2];
          fr[(k < 2 ? k * NV : KPQ) + cell2] = fmaxf(v + hw[3 * C + k], 0.0f);
        }
      }
    } else {
    // ---- residual blocks
    for (int l = 0; l < L; l += 2) {
      init(bl + (1 + l) * C);
      FSTAMP(5 + 4 * l);
      conv_layer16g<CW / 16, true, CW, N>(act, wl + (size_t)l * 9 * (CW / 16) * KSB16, wl + (size_t)(l + 1) * 9 * (CW / 16) * KSB16, wave, lane, pp0, vm, acc, lc, wf);
      lfinal(bl + (1 + l) * C, false);
      FSTAMP(6 + 4 * l);
      if (l == 0 && tid == 0) {  // publish the next board's slot
        const int tk = slots[3];
        slots[2] = tk >= G ? -1 : ((flags == nullptr || (slots[4] & 0xFF) != 0) ? tk : -2);
      }
      __syncthreads();
      FSTAMP(7 + 4 * l);
      epilogue_res(bl + (2 + l) * C);
      constexpr int RWV = (RGW + 63) / 64;  // waves that carry ring words
      if (l == 0 && x == nullptr && tid <= 64 * RWV) {  // (slots[2] was written before the barrier above)
        // the next board's packed input, by LDS-DMA into the buffer the running board does not use (the first RWV waves: the
        // RGW ring words, lane 0 of the wave after them: the player word); waited for at the top of the next board
        const int gn = __builtin_amdgcn_readfirstlane(slots[2]);
        if (gn >= 0) {
          uint32_t *dst = rgb + (rsel ^ 1) * RGS;
          if (wave < RWV) {
            if (tid < RGW)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ring + (size_t)gn * RGW + tid),
                                               (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(dst + wave * 64), 4, 0, 0);
          } else if (tid == 64 * RWV)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)&leafrec[gn].w,
                                             (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)(dst + RGW), 4, 0, 0);
        }
      }
      __syncthreads();
      FSTAMP(8 + 4 * l);
      FSTAMP(9 + 4 * l);
      if (l + 2 < L) {
        conv_layer16g<CW / 16, true, CW, N>(act, wl + (size_t)(l + 1) * 9 * (CW / 16) * KSB16, wl + (size_t)(l + 2) * 9 * (CW / 16) * KSB16, wave, lane, pp0, vm, acc, lc,
                                   wf);
        lfinal(bl + (2 + l) * C, true);
        FSTAMP(10 + 4 * l);
        __syncthreads();
        FSTAMP(11 + 4 * l);
        epilogue();
        __syncthreads();
        FSTAMP(12 + 4 * l);
      } else {  // the tower's last layer: its output stays in the accumulators for the heads; prefetch the next board's stem
        conv_layer16g<CW / 16, true, CW, N>(act, wl + (size_t)(l + 1) * 9 * (CW / 16) * KSB16, w0, wave, lane, pp0, vm, acc, lc, wf);
        lfinal(bl + (2 + l) * C, true);
        FSTAMP(10 + 4 * l);
      }
    }
    // ---- heads: 1x1 convs + ReLU from the accumulators.  A lane holds 8 of its cell's channels (2 channel tiles x 4), the
    // four k-quarter groups of a wave hold the 32 channels of its half, the other half sits in wave ^ 2.
    {
      const int ln = fresh(lane), kq = ln >> 4, pp0 = NT * chalf * 16 + (ln & 15), lco = wave * 16 + (ln & 15);
      float hs[NT][3];
#pragma unroll
      for (int pt = 0; pt < NT; ++pt) hs[pt][0] = hs[pt][1] = hs[pt][2] = 0.0f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x4 wv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wv[k] = *(const f32x4 *)(hw + k * C + (ch * NCT + (ct ^ cto)) * 16 + kq * 4);
#pragma unroll
        for (int pt = 0; pt < NT; ++pt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = fmaxf(acc[pt][ct][j], 0.0f);
#pragma unroll
            for (int k = 0; k < 3; ++k) hs[pt][k] = fmaf(a, wv[k][j], hs[pt][k]);
          }
      }
      FSTAMP(26);
#pragma unroll
      for (int pt = 0; pt < NT; ++pt)
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // sum over the four k-quarter groups (same order on every lane of a cell)
          float t = hs[pt][k];
          t = add32(add16(t));
          hs[pt][k] = t;
        }
      FSTAMP(27);
      {  // cell 224: this wave's 16 channels (lanes of k quarter 0), butterfly over those lanes, partial per wave to LDS
        const float a = kq == 0 ? fmaxf(lc, 0.0f) : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float t = a * hw[k * C + lco];
          t = rowsum(t);
          if (ln == 0) lh[wave * 3 + k] = t;
        }
      }
#pragma unroll
      for (int pt = 0; pt < NT; ++pt) {
        if ((pt & 3) != kq) continue;  // the four k-quarter groups share the LDS stores (every lane holds the sum)
#pragma unroll
        for (int k = 0; k < 3; ++k) hp[(ch * 3 + k) * NV + pp0 + 16 * pt] = hs[pt][k];
      }
      FSTAMP(28);
      __syncthreads();
      FSTAMP(29);
      // the feature rows leave as whole-wave, consecutive 4-byte stores (a handful per wave: partial-wave scattered stores
      // cost several hundred cycles of issue each)
      float *fr = feat + (size_t)g * FS;
      for (int e = tid; e < 3 * NV; e += NTH) {
        const int k = e / NV, cell = e - k * NV;
        float v = 0.0f;
        if (cell == LCELL) {
#pragma unroll
          for (int w = 0; w < NW; ++w) v += lh[3 * w + k];
        } else {
#pragma unroll
          for (int c = 0; c < NCQ; ++c) v += hp[(3 * c + k) * NV + cell];
        }
        fr[(k == 0 ? 0 : KP + (k - 1) * NV) + cell] = fmaxf(v + hw[3 * C + k], 0.0f);
      }
    }
    }  // !QK
    FSTAMP(30);
  }
#if AZ_F32_STAMP
  if (tid == 0) {
    unsigned long long *o = g_f32_wg + 4 * blockIdx.x;
    o[0] = wg_t0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (unsigned long long)(it - 1);
    o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
  }
#endif
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The heads' Linear layers in f32 (examples/gomoku/model.py:69-90): logits = feat[:, :NV] @ Wp^T + bp;
// value = tanh(relu(feat[:, KP:KP+2NV] @ W1^T + b1) @ w2 + b2), as f32 MFMA GEMMs read straight from L2.  One workgroup = 32
// boards x one job (blockIdx.y: four policy output tiles, one per wave | the value head).  A = features (lane: board
// l & 31, k half l >> 5: one 16-byte load = the lane's operands of four MFMAs), B = weights packed
// [n-tile][k-step][half][n 32][4 k] f32, zero padded.
template <int NV>
__global__ __launch_bounds__(256, 2) void k_heads_f32(const float *__restrict__ feat, int FS, int G, const char *__restrict__ wp,
                                                      const float *__restrict__ bp, const char *__restrict__ w1,
                                                      const float *__restrict__ b1, const float *__restrict__ w2, float b2,
                                                      float *__restrict__ logits, float *__restrict__ values) {
  constexpr int KP = (NV + 15) / 16 * 16, KSP = KP / 8, NTP = (NV + 31) / 32;
  constexpr int KV = (2 * NV + 15) / 16 * 16, KSV = KV / 8, KH = KSV / 2;
  __shared__ float part[4][32][33];
  __shared__ float vsum[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g0 = blockIdx.x * 32;
  const float *fa = feat + (size_t)min(g0 + r, G - 1) * FS + 4 * h;  // rows past the batch: a copy of the last row, not stored
  constexpr int PB = (NTP + 3) / 4;
  const int job = blockIdx.y;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  if (job < PB) {
    const int nt = job * 4 + wave;
    if (nt >= NTP) return;
    const char *wb = wp + ((size_t)nt * KSP * 2 + h) * 512 + r * 16;
#pragma unroll 6
    for (int ks = 0; ks < KSP; ++ks) {
      const f32x4 b = *(const f32x4 *)(wb + (size_t)ks * 1024);
      const f32x4 a = *(const f32x4 *)(fa + ks * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    const int n = nt * 32 + r;
    if (n < NV) {
      const float bias = bp[n];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        if (g0 + row < G) logits[(size_t)(g0 + row) * NV + n] = acc[i] + bias;
      }
    }
    return;
  }
  {  // value: hidden tile (wave & 1), K half (wave >> 1)
    const int nt = wave & 1, kh = wave >> 1;
    const char *wb = w1 + ((size_t)nt * KSV * 2 + h) * 512 + r * 16;
#pragma unroll 6
    for (int k = 0; k < KH; ++k) {
      const int ks = kh * KH + k;
      const f32x4 b = *(const f32x4 *)(wb + (size_t)ks * 1024);
      const f32x4 a = *(const f32x4 *)(fa + KP + ks * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i];
    __syncthreads();
    if (wave < 2) {
      const int n = wave * 32 + r;
      const float bn = b1[n], wn = w2[n];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        float t = fmaxf(part[wave][row][r] + part[wave + 2][row][r] + bn, 0.0f) * wn;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if (r == 0) vsum[wave][row] = t;
      }
    }
    __syncthreads();
    if (tid < 32 && g0 + tid < G) values[g0 + tid] = tanhf(vsum[0][tid] + vsum[1][tid] + b2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The QuickModel heads' Linear layers in f32 (nn_models/quick_model.py:138-154, 185-199 with config/quick_model_config.py's
// defaults: two Linear layers per head, 64 hidden features):
//   logits = relu(feat[:, :2NV] @ W1p^T + b1p) @ W2p^T + b2p;   value = tanh(relu(feat[:, KPQ:KPQ+NV] @ W1v^T + b1v) @ w2v + b2v)
// One workgroup = 32 boards.  Phase 1: wave w = hidden tile (w & 1) of head (w >> 1) as an f32 MFMA GEMM (A = feature rows,
// B = weights packed [n-tile][k-step of 8][half][n 32][4 k], zero padded), bias + ReLU into LDS.  Phase 2: the policy's second
// layer as MFMA GEMMs from LDS (output tiles wave, wave + 4, ...); the value's second layer as a k-ordered fmaf chain per board.
template <int NV>
__global__ __launch_bounds__(256, 2) void k_quick_heads_f32(const float *__restrict__ feat, int FS, int G, const char *__restrict__ w1p,
                                                            const float *__restrict__ b1p, const char *__restrict__ w2p,
                                                            const float *__restrict__ b2p, const char *__restrict__ w1v,
                                                            const float *__restrict__ b1v, const float *__restrict__ w2v, float b2v,
                                                            float *__restrict__ logits, float *__restrict__ values) {
  constexpr int KPQ = (2 * NV + 15) / 16 * 16, KVQ = (NV + 15) / 16 * 16, NTP = (NV + 31) / 32, HS = 68;
  __shared__ __attribute__((aligned(16))) float hid[2][32][HS];  // [head][board][hidden unit], rows padded to 68 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g0 = blockIdx.x * 32;
  const float *fa = feat + (size_t)min(g0 + r, G - 1) * FS + 4 * h;  // rows past the batch: a copy of the last row, not stored
  {
    const int head = wave >> 1, nt = wave & 1;
    const int ksn = head == 0 ? KPQ / 8 : KVQ / 8;
    const char *wb = (head == 0 ? w1p : w1v) + ((size_t)nt * ksn * 2 + h) * 512 + r * 16;
    const float *fh = fa + (head == 0 ? 0 : KPQ);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll 6
    for (int ks = 0; ks < ksn; ++ks) {
      const f32x4 b = *(const f32x4 *)(wb + (size_t)ks * 1024);
      const f32x4 a = *(const f32x4 *)(fh + ks * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    const int n = nt * 32 + r;
    const float bn = (head == 0 ? b1p : b1v)[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) hid[head][(i & 3) + 8 * (i >> 2) + 4 * h][n] = fmaxf(acc[i] + bn, 0.0f);
  }
  __syncthreads();
  for (int nt = wave; nt < NTP; nt += 4) {
    const char *wb = w2p + ((size_t)nt * 8 * 2 + h) * 512 + r * 16;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const f32x4 b = *(const f32x4 *)(wb + (size_t)ks * 1024);
      const f32x4 a = *(const f32x4 *)(&hid[0][r][ks * 8 + 4 * h]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    const int n = nt * 32 + r;
    if (n < NV) {
      const float bias = b2p[n];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        if (g0 + row < G) logits[(size_t)(g0 + row) * NV + n] = acc[i] + bias;
      }
    }
  }
  if (tid < 32 && g0 + tid < G) {
    float t = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) t = fmaf(hid[1][tid][k], w2v[k], t);
    values[g0 + tid] = tanhf(t + b2v);
  }
}

}  // namespace aztf32

static std::mutex g_f32_mutex;
static unsigned int g_f32_attr_done[64];

#if AZ_F32_STAMP
extern "C" int az_debug_f32_stamps(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(aztf32::g_f32_stamps), sizeof(unsigned long long) * 4 * 32) == hipSuccess ? 0 : -3;
}
extern "C" int az_debug_f32_wg(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(aztf32::g_f32_wg), sizeof(unsigned long long) * 1024 * 4) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

int az_heads_forward_f32(int32_t board_size, int32_t n_boards, const float *feat_dev, int32_t feat_stride, const void *wp_dev,
                         const float *bp_dev, const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2,
                         float *logits_dev, float *values_dev, void *stream) {
  using namespace aztf32;
  if (!feat_dev || !wp_dev || !bp_dev || !w1_dev || !b1_dev || !w2_dev || !logits_dev || !values_dev) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || (board_size != 9 && board_size != 15)) { tower_set_error("n_boards >= 1, board_size 9 or 15"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size) || (feat_stride & 3)) { tower_set_error("feat_stride: a multiple of 4, at least az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
  const int ntp = (board_size * board_size + 31) / 32;
  const dim3 grid((n_boards + 31) / 32, (ntp + 3) / 4 + 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  StreamDeviceGuard sguard(s);
  if (!sguard.ok) { tower_set_error("cannot select the stream's device"); return AZ_EHIP; }
  if (board_size == 15)
    hipLaunchKernelGGL((k_heads_f32<225>), grid, block, 0, s, feat_dev, (int)feat_stride, (int)n_boards, (const char *)wp_dev, bp_dev,
                       (const char *)w1_dev, b1_dev, w2_dev, b2, logits_dev, values_dev);
  else
    hipLaunchKernelGGL((k_heads_f32<81>), grid, block, 0, s, feat_dev, (int)feat_stride, (int)n_boards, (const char *)wp_dev, bp_dev,
                       (const char *)w1_dev, b1_dev, w2_dev, b2, logits_dev, values_dev);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_heads_f32 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

int az_tower_forward_f32(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                         const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *w0_dev,
                         const void *wl_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev, float *feat_f32_dev,
                         int32_t feat_stride, void *stream) {
  using namespace aztf32;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !bias_dev || !headw_dev || !headb_dev || !feat_f32_dev ||
      (n_conv_layers > 0 && !wl_dev)) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1)) { tower_set_error("n_boards >= 1, even n_conv_layers"); return AZ_EINVAL; }
  if ((board_size != 9 && board_size != 15) || (channels != 64 && channels != 128)) {
    tower_set_error("f32 tower: board_size 9 or 15, 64 or 128 channels");
    return AZ_EINVAL;
  }
  if (channels == 128 && n_conv_layers > T16_MAX_LAYERS) { tower_set_error("f32 tower, 128 channels: at most 40 conv layers"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { tower_set_error("feat_stride below az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
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
  auto launch = [&](auto kern, size_t lds, int bpw, int variant, int wg_cu = 2, int threads = 256) -> int {
    if (lds * wg_cu > 160 * 1024) { tower_set_error("f32 tower geometry exceeds the CU's LDS"); return AZ_EINVAL; }
    {
      std::lock_guard<std::mutex> lock(g_f32_mutex);
      if (!(g_f32_attr_done[dev] & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
          return AZ_EHIP;
        }
        g_f32_attr_done[dev] |= 1u << variant;
      }
    }
    static const int wg_per_cu = getenv("AZ_TOWER_F32_WG1") ? 1 : 64;  // diagnostic: one workgroup per CU
    const int groups = (n_boards + bpw - 1) / bpw, cap = 256 * (wg_per_cu < wg_cu ? wg_per_cu : wg_cu);  // workgroups that fit the chip
    const int grid = groups < cap ? groups : cap;
    const bool use_flags = !x_f32_dev && flags_dev;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, x_f32_dev, (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)w0_dev, (const char *)wl_dev, bias_dev,
                       n_conv_layers, headw_dev, headb_dev, feat_f32_dev, (int)feat_stride, tickets, 2, (float *)nullptr);
    return AZ_OK;
  };
  int rc;
  auto lds_t16 = [](int cw, int n) -> size_t {  // act + zero row, head filters, packed input, head partial sums, bias table
    const int nv = n * n, rgw = 16 * ((nv + 31) / 32), ncg = n == 15 ? cw / 32 : cw / 16, nw = n == 15 ? cw / 16 : cw / 16;
    return (size_t)((nv + 1) * (4 * cw + 16)) + ((3 * cw + 3) * 4 + 15) / 16 * 16 + (2 * (rgw + 4) + 8) * 4 + (ncg * 3 * nv + 3 * nw + 3) / 4 * 16 +
           (size_t)(1 + T16_MAX_LAYERS) * cw * 4;
  };
  static const bool use32 = getenv("AZ_TOWER_F32_TILE32") != nullptr;  // diagnostic: the 32-cell-tile kernel on 15x15 too
  if (channels == 128 && board_size == 15) rc = launch(k_tower_f32_t16<128, 15>, lds_t16(128, 15), 1, 3, 1, 512);
  else if (channels == 128) rc = launch(k_tower_f32_t16<128, 9>, lds_t16(128, 9), 1, 5, 1, 512);  // (8 waves x one 16-channel tile, one workgroup per CU)
  else if (board_size == 15 && !use32 && n_conv_layers <= T16_MAX_LAYERS) rc = launch(k_tower_f32_t16<64, 15>, lds_t16(64, 15), 1, 2);
  else if (board_size == 15) rc = launch(k_tower_f32<15, 1>, GeoF<15, 1>::lds_bytes(), 1, 0);
  else if (!use32 && n_conv_layers <= T16_MAX_LAYERS) rc = launch(k_tower_f32_t16<64, 9>, lds_t16(64, 9), 1, 4, T16Geo<64, 9>::WG_PER_CU);
  else rc = launch(k_tower_f32<9, 3>, GeoF<9, 3>::lds_bytes(), 3, 1);
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_f32 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

int64_t az_quick_stash_bytes(int32_t board_size, int32_t channels) {
  using namespace aztf32;
  if (board_size == 15 && channels == 128) return (int64_t)256 * 1 * T16Geo<128, 15>::NW * (T16Geo<128, 15>::NT * T16Geo<128, 15>::NCT * 256 + 64) * 4;
  if (board_size == 9 && channels == 128) return (int64_t)256 * 1 * T16Geo<128, 9>::NW * (T16Geo<128, 9>::NT * T16Geo<128, 9>::NCT * 256 + 64) * 4;
  if (board_size == 15 && channels == 64) return (int64_t)256 * 2 * T16Geo<64, 15>::NW * (T16Geo<64, 15>::NT * T16Geo<64, 15>::NCT * 256 + 64) * 4;
  if (board_size == 9 && channels == 64) return (int64_t)256 * T16Geo<64, 9>::WG_PER_CU * T16Geo<64, 9>::NW * (T16Geo<64, 9>::NT * T16Geo<64, 9>::NCT * 256 + 64) * 4;
  return -1;
}

int az_quick_tower_forward_f32(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_blocks, int32_t block_layers,
                               const float *x_f32_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                               const void *w0_dev, const void *wl_dev, const float *bias_dev, const float *headw_dev,
                               const float *headb_dev, float *feat_f32_dev, int32_t feat_stride, void *stash_dev, int64_t stash_bytes,
                               void *stream) {
  using namespace aztf32;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !wl_dev || !bias_dev || !headw_dev || !headb_dev || !feat_f32_dev ||
      !stash_dev) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_blocks < 1 || block_layers < 1 || (int64_t)n_blocks * block_layers > T16_MAX_LAYERS) {
    tower_set_error("quick tower: n_boards >= 1, n_blocks >= 1, block_layers >= 1, at most 40 conv layers after the transition conv");
    return AZ_EINVAL;
  }
  const int64_t need = az_quick_stash_bytes(board_size, channels);
  if (need < 0) { tower_set_error("quick tower: board_size 9 or 15, 64 or 128 channels"); return AZ_EINVAL; }
  if (stash_bytes < need) { tower_set_error("quick tower: stash_bytes below az_quick_stash_bytes(board_size, channels)"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { tower_set_error("feat_stride below az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
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
  const int L = n_blocks * block_layers;
  auto lds_t16 = [](int cw, int n) -> size_t {  // as in az_tower_forward_f32
    const int nv = n * n, rgw = 16 * ((nv + 31) / 32), ncg = n == 15 ? cw / 32 : cw / 16, nw = cw / 16;
    return (size_t)((nv + 1) * (4 * cw + 16)) + ((3 * cw + 3) * 4 + 15) / 16 * 16 + (2 * (rgw + 4) + 8) * 4 + (ncg * 3 * nv + 3 * nw + 3) / 4 * 16 +
           (size_t)(1 + T16_MAX_LAYERS) * cw * 4;
  };
  auto launch = [&](auto kern, size_t lds, int variant, int wg_cu, int threads) -> int {
    if (lds * wg_cu > 160 * 1024) { tower_set_error("quick tower geometry exceeds the CU's LDS"); return AZ_EINVAL; }
    {
      std::lock_guard<std::mutex> lock(g_f32_mutex);
      if (!(g_f32_attr_done[dev] & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
          return AZ_EHIP;
        }
        g_f32_attr_done[dev] |= 1u << variant;
      }
    }
    const int cap = 256 * wg_cu, grid = n_boards < cap ? n_boards : cap;  // (the stash holds one region per workgroup of a full grid)
    const bool use_flags = !x_f32_dev && flags_dev;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, x_f32_dev, (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)w0_dev, (const char *)wl_dev, bias_dev, L,
                       headw_dev, headb_dev, feat_f32_dev, (int)feat_stride, tickets, (int)block_layers, (float *)stash_dev);
    return AZ_OK;
  };
  int rc;
  if (channels == 128 && board_size == 15) rc = launch(k_tower_f32_t16<128, 15, true>, lds_t16(128, 15), 8, 1, 512);
  else if (channels == 128) rc = launch(k_tower_f32_t16<128, 9, true>, lds_t16(128, 9), 11, 1, 512);
  else if (board_size == 15) rc = launch(k_tower_f32_t16<64, 15, true>, lds_t16(64, 15), 9, 2, 256);
  else rc = launch(k_tower_f32_t16<64, 9, true>, lds_t16(64, 9), 10, T16Geo<64, 9>::WG_PER_CU, 256);
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_f32_t16 (quick) launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

int az_quick_heads_forward_f32(int32_t board_size, int32_t n_boards, const float *feat_dev, int32_t feat_stride, const void *w1p_dev,
                               const float *b1p_dev, const void *w2p_dev, const float *b2p_dev, const void *w1v_dev,
                               const float *b1v_dev, const float *w2v_dev, float b2v, float *logits_dev, float *values_dev, void *stream) {
  using namespace aztf32;
  if (!feat_dev || !w1p_dev || !b1p_dev || !w2p_dev || !b2p_dev || !w1v_dev || !b1v_dev || !w2v_dev || !logits_dev || !values_dev) {
    tower_set_error("null argument");
    return AZ_EINVAL;
  }
  if (n_boards < 1 || (board_size != 9 && board_size != 15)) { tower_set_error("n_boards >= 1, board_size 9 or 15"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size) || (feat_stride & 3)) { tower_set_error("feat_stride: a multiple of 4, at least az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
  const dim3 grid((n_boards + 31) / 32), block(256);
  hipStream_t s = (hipStream_t)stream;
  StreamDeviceGuard sguard(s);
  if (!sguard.ok) { tower_set_error("cannot select the stream's device"); return AZ_EHIP; }
  if (board_size == 15)
    hipLaunchKernelGGL((k_quick_heads_f32<225>), grid, block, 0, s, feat_dev, (int)feat_stride, (int)n_boards, (const char *)w1p_dev, b1p_dev,
                       (const char *)w2p_dev, b2p_dev, (const char *)w1v_dev, b1v_dev, w2v_dev, b2v, logits_dev, values_dev);
  else
    hipLaunchKernelGGL((k_quick_heads_f32<81>), grid, block, 0, s, feat_dev, (int)feat_stride, (int)n_boards, (const char *)w1p_dev, b1p_dev,
                       (const char *)w2p_dev, b2p_dev, (const char *)w1v_dev, b1v_dev, w2v_dev, b2v, logits_dev, values_dev);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_quick_heads_f32 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

}  // extern "C"
