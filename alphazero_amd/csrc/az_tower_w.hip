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
// One workgroup of 8 waves owns one board at a time for the whole tower; activations live in LDS (act[row][col][64 ch] f32,
// 264 B per cell, 4,032 B per row: with these two pitches the 16 tiles x 2 channel pairs of a half-wave's ds_read_b64 fall on
// 32 different bank pairs), layers run in place.  Wave w = (tile group tg = w & 3: tile rows 2 tg, 2 tg + 1 x 8 tile columns
// = the N = 16 of the MFMA; channel half ch = w >> 2: 2 tiles of 16 output channels) and holds ALL 16 transform points of its
// 16 tiles x 32 channels in 128 accumulator registers, so that
//   * a lane builds its B operands itself: it reads the 4x4 patch of its tile for 2 input channels (16 ds_read_b64, addresses
//     fixed per lane: cells off the board point at a zero cell), forms V = B^T d B in registers (32 packed adds) and feeds 64
//     MFMAs (16 points x 2 channel tiles x 2 k-steps) with it -- the transformed input never exists in memory;
//   * the inverse transform A^T M A is in-lane too: the 16 M_p of an (output channel, tile) pair are 16 registers of one lane.
// Transformed weights U (host: G g G^T in f64, rounded to f32; 256 KiB per layer) stream L2 -> LDS by LDS-DMA in stages of 8
// input channels (32 KiB, double buffered, one barrier per stage), laid out so that a lane's A operands of four points are one
// conflict-free ds_read_b128.  Bias and the block's residual enter through the accumulators: A^T M A maps
// M[0][0], -M[0][3], -M[3][0], M[3][3] to the four outputs of the tile one to one, so the lane that overwrites a block's
// input in the first conv's epilogue reads it there and starts the second conv's accumulators at bias + x (as az_tower_f32.hip).
// Boards are handed out by the library's ticket counter (az_tower.hip); the input is the engine's packed leaf batch (8
// bitboards + player) with its needs-the-network flags, or dense f32 planes.  Output: the f32 feature rows k_heads_f32 reads.
// C ABI at the bottom; declared in include/az_engine.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <string>

#include "../../include/az_engine.h"

int tower_device(hipStream_t stream, int *dev);             // az_tower.hip
unsigned int *tower_tickets(int dev, hipStream_t stream);   // az_tower.hip
void tower_set_error(const std::string &msg);               // az_tower.hip

namespace aztw {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int N = 15, NV = N * N, C = 64;
constexpr int CSB = 264;                      // bytes per cell (64 f32 + 8)
constexpr int RSB = 4032;                     // bytes per board row (15 cells + 72)
constexpr int ZERO_OFF = N * RSB;             // a cell of zeros: every patch element off the board
constexpr int DUMP_OFF = ZERO_OFF + CSB;      // where the outputs of the 16th row / column go
constexpr int ACT_BYTES = (DUMP_OFF + CSB + 15) / 16 * 16;
constexpr int STAGE_BYTES = 32768;            // U of 8 input channels: [j 2][kq 4][ct 4][pg 4][co 16][4 points] f32
constexpr int MAX_LAYERS = 40;                // bias rows kept in LDS
constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = 8 * SWB;  // packed input: 8 states x 2 colours x 8 words
constexpr int NTH = 512;

constexpr int OFF_ACT = 2 * STAGE_BYTES;
constexpr int OFF_BL = OFF_ACT + ACT_BYTES;
constexpr int OFF_HW = OFF_BL + (1 + MAX_LAYERS) * C * 4;
constexpr int OFF_HP = OFF_HW + 208 * 4;                       // 3 x 64 head filters + 3 biases (+ pad)
constexpr int OFF_RG = OFF_HP + ((2 * 3 * NV + 3) / 4 * 4) * 4;  // [2 channel halves][3 heads][NV] partial head sums
constexpr int OFF_SL = OFF_RG + (RGW + 4) * 4;
constexpr int LDS_BYTES = OFF_SL + 8 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(OFF_BL % 16 == 0 && OFF_HW % 16 == 0 && OFF_HP % 16 == 0 && OFF_RG % 16 == 0, "alignment");

#define LDS_PTR(T, a) ((__attribute__((address_space(3))) T *)(uintptr_t)(a))
__device__ __forceinline__ f32x2 lds_r2(uint32_t a) { return *LDS_PTR(const f32x2, a); }
__device__ __forceinline__ f32x4 lds_r4(uint32_t a) { return *LDS_PTR(const f32x4, a); }
__device__ __forceinline__ void lds_w2(uint32_t a, f32x2 v) { *LDS_PTR(f32x2, a) = v; }

// async global -> LDS copy of 16 B per lane: the LDS destination is the wave-uniform `lds_base` + lane * 16
__device__ __forceinline__ void glds16(const char *g_lane, uint32_t lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                   (__attribute__((address_space(3))) void *)(uintptr_t)lds_base, 16, 0, 0);
}
// one stage of U (32 KiB, contiguous in global memory and in LDS alike): 1 KiB per wave-instruction, 4 per wave
__device__ __forceinline__ void stage_dma(const char *src, uint32_t dst, int wave, int lane) {
#pragma unroll
  for (int p = 0; p < 4; ++p) glds16(src + (p * 8 + wave) * 1024 + lane * 16, dst + (p * 8 + wave) * 1024);
}

// One conv layer in the transform domain: acc[point][ct] += U V over `ns` stages of 8 input channels.  On entry the layer's
// stage 0 is on its way into buffer `buf`; the last stage prefetches `wnext` (stage 0 of whatever runs next).  A stage:
// wait for its weights, barrier (every wave has also finished the stage before: the other buffer is free), start the next
// stage's DMA, then per wave 16 patch reads -> V (registers) -> 2 k-steps x (8 ds_read_b128 of U, 32 MFMAs).
__device__ __forceinline__ void conv_w(uint32_t lds0, const uint32_t (&pa)[16], uint32_t abase, const char *wg, const char *wnext, int ns,
                                       int wave, int lane, int &buf, f32x4 (&acc)[16][2]) {
#pragma unroll 1
  for (int s = 0; s < ns; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage_dma(s + 1 < ns ? wg + (size_t)(s + 1) * STAGE_BYTES : wnext, lds0 + (uint32_t)((buf ^ 1) * STAGE_BYTES), wave, lane);
    const uint32_t so = (uint32_t)s * 32u;
    f32x2 d[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i >> 2][i & 3] = lds_r2(pa[i] + so);
    f32x2 t[4][4], v[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // B^T d
      t[0][c] = d[0][c] - d[2][c];
      t[1][c] = d[1][c] + d[2][c];
      t[2][c] = d[2][c] - d[1][c];
      t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (B^T d) B
      v[i][0] = t[i][0] - t[i][2];
      v[i][1] = t[i][1] + t[i][2];
      v[i][2] = t[i][2] - t[i][1];
      v[i][3] = t[i][1] - t[i][3];
    }
    const uint32_t ab = abase + (uint32_t)(buf * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 a = lds_r4(ab + (uint32_t)(j * 16384 + ct * 1024 + i * 256));
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * i + q][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], v[i][q][j], acc[4 * i + q][ct], 0, 0, 0);
        }
    buf ^= 1;
  }
}

__global__ __launch_bounds__(NTH, 2) void k_tower_f32w(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                        const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                        const char *__restrict__ u0, const char *__restrict__ ul,
                                                        const float *__restrict__ bias, int L, const float *__restrict__ headw,
                                                        const float *__restrict__ headb, float *__restrict__ feat, int FS,
                                                        unsigned int *tickets) {
  extern __shared__ __attribute__((aligned(1024))) char smem_w[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_w;
  const uint32_t act0 = lds0 + OFF_ACT;
  char *act = smem_w + OFF_ACT;
  float *bl = (float *)(smem_w + OFF_BL);
  float *hw = (float *)(smem_w + OFF_HW);
  float *hp = (float *)(smem_w + OFF_HP);
  uint32_t *rg = (uint32_t *)(smem_w + OFF_RG);
  int *slots = (int *)(smem_w + OFF_SL);  // {this board, the next ticket, its flag byte (-1: not read)}
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4, tg = wave & 3, chalf = wave >> 2;
  const int ty = 2 * tg + (n >> 3), tx = n & 7;

  for (int i = tid; i < ACT_BYTES / 16; i += NTH) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // incl. the zero and dump cells
  for (int i = tid; i < (1 + L) * C; i += NTH) bl[i] = bias[i];
  for (int i = tid; i < 3 * C + 3; i += NTH) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];

  uint32_t pa[16];  // the lane's 4x4 input patch: rows 2 ty - 1 .. + 3, columns 2 tx - 1 .. + 3, its channel pair 2 kq (+ 8 s)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 2 * ty - 1 + (i >> 2), c = 2 * tx - 1 + (i & 3);
    pa[i] = act0 + (uint32_t)((r >= 0 && r < N && c >= 0 && c < N ? r * RSB + c * CSB : ZERO_OFF) + kq * 8);
  }
  uint32_t oc[4];  // the lane's 2x2 outputs: channels 32 chalf + 16 ct + 4 kq .. + 3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 2 * ty + (i >> 1), c = 2 * tx + (i & 1);
    oc[i] = act0 + (uint32_t)((r < N && c < N ? r * RSB + c * CSB : DUMP_OFF) + (32 * chalf + 4 * kq) * 4);
  }
  const uint32_t abase = lds0 + (uint32_t)(kq * 4096 + chalf * 2048 + n * 16);
  const int bo = 32 * chalf + 4 * kq;  // first of the lane's channels (ct adds 16)

  int buf = 0;
  stage_dma(u0, lds0, wave, lane);
  if (tid == 0) { slots[1] = (int)blockIdx.x; slots[2] = -1; }
  __syncthreads();

  f32x4 acc[16][2];
  auto preload = [&](const f32x4 (&pre)[4][2]) {  // A^T M A sends M[0][0], -M[0][3], -M[3][0], M[3][3] to outputs 00, 01, 10, 11
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[p][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      acc[0][ct] = pre[0][ct];
      acc[3][ct] = -pre[1][ct];
      acc[12][ct] = -pre[2][ct];
      acc[15][ct] = pre[3][ct];
    }
  };
  // inverse transform, ReLU, store in place; the next layer's accumulators start at its bias (+ the activations this lane
  // overwrites, when they are the residual of the block's second conv)
  auto epilogue = [&](bool with_res, const float *bnext) {
    float y[4][2][4];  // [output 2 a + b][ct][r]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s0[q] = (acc[q][ct][r] + acc[4 + q][ct][r]) + acc[8 + q][ct][r];
          s1[q] = (acc[4 + q][ct][r] - acc[8 + q][ct][r]) - acc[12 + q][ct][r];
        }
        y[0][ct][r] = (s0[0] + s0[1]) + s0[2];
        y[1][ct][r] = (s0[1] - s0[2]) - s0[3];
        y[2][ct][r] = (s1[0] + s1[1]) + s1[2];
        y[3][ct][r] = (s1[1] - s1[2]) - s1[3];
      }
    f32x4 pre[4][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x4 bv = bnext != nullptr ? *(const f32x4 *)(bnext + bo + 16 * ct) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const uint32_t a = oc[o] + (uint32_t)(ct * 64);
        f32x4 base = bv;
        if (with_res) {
          const f32x2 x0 = lds_r2(a), x1 = lds_r2(a + 8);
          base = bv + (f32x4){x0[0], x0[1], x1[0], x1[1]};
        }
        lds_w2(a, (f32x2){fmaxf(y[o][ct][0], 0.0f), fmaxf(y[o][ct][1], 0.0f)});
        lds_w2(a + 8, (f32x2){fmaxf(y[o][ct][2], 0.0f), fmaxf(y[o][ct][3], 0.0f)});
        pre[o][ct] = base;
      }
    }
    preload(pre);
  };

  for (;;) {
    if (tid == 0) {  // this board: the ticket drawn ahead, or -- when its leaf is terminal (flag off) -- the next that is not
      int tk = slots[1], fl = slots[2];
      while (tk < G && flags != nullptr) {
        if (fl < 0) fl = (int)flags[tk];
        if (fl != 0) break;
        tk = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
        fl = -1;
      }
      slots[0] = tk;
    }
    __syncthreads();
    const int g = slots[0];
    if (g >= G) break;
    unsigned int drawn = 0;
    if (tid == 0) drawn = atomicAdd(&tickets[0], 1u);  // the next ticket (lands while the input is fetched)
    if (x == nullptr) {
      for (int i = tid; i < RGW + 1; i += NTH) rg[i] = i < RGW ? ring[(size_t)g * RGW + i] : (uint32_t)leafrec[g].w;
      __syncthreads();
    }
    if (tid < NV) {
      const int cell = tid, row = cell / N, col = cell - row * N;
      float c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = 0.0f;
      if (x != nullptr) {
        const float *xr = x + (size_t)g * 9 * NV + cell;
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) c[cc] = xr[cc * NV];
      } else {
        const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          c[k] = (float)(int)((rg[k * SWB + wi] >> sh) & 1u) - (float)(int)((rg[k * SWB + WC + wi] >> sh) & 1u);
        c[8] = (int)rg[RGW] > 0 ? 1.0f : -1.0f;
      }
      const uint32_t a = act0 + (uint32_t)(row * RSB + col * CSB);
#pragma unroll
      for (int k = 0; k < 8; ++k) lds_w2(a + 8 * k, (f32x2){c[2 * k], c[2 * k + 1]});
    }
    if (tid == 0) {
      slots[1] = (int)(gridDim.x + drawn);
      slots[2] = -1;
    }
    {  // stem: accumulators start at its bias
      f32x4 pre[4][2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 bv = *(const f32x4 *)(bl + bo + 16 * ct);
#pragma unroll
        for (int o = 0; o < 4; ++o) pre[o][ct] = bv;
      }
      preload(pre);
    }
    // (the first stage's barrier orders the input planes before the patch reads)
    conv_w(lds0, pa, abase, u0, L > 0 ? ul : u0, 2, wave, lane, buf, acc);
    __syncthreads();  // every wave has read the layer's input: the outputs may overwrite it
    epilogue(false, L > 0 ? bl + C : nullptr);
    for (int l = 0; l < L; l += 2) {
      const char *w1 = ul + (size_t)l * 8 * STAGE_BYTES, *w2 = w1 + (size_t)8 * STAGE_BYTES;
      conv_w(lds0, pa, abase, w1, w2, 8, wave, lane, buf, acc);
      __syncthreads();
      epilogue(true, bl + (2 + l) * C);
      conv_w(lds0, pa, abase, w2, l + 2 < L ? w2 + (size_t)8 * STAGE_BYTES : u0, 8, wave, lane, buf, acc);
      __syncthreads();
      epilogue(false, l + 2 < L ? bl + (3 + l) * C : nullptr);
    }
    __syncthreads();
    // ---- heads: the three 1x1 convs + ReLU from the finished activations: a thread takes a cell and half of its channels
    for (int e = tid; e < 2 * NV; e += NTH) {
      const int half = e >= NV ? 1 : 0, cell = e - half * NV, row = cell / N, col = cell - row * N;
      const uint32_t a = act0 + (uint32_t)(row * RSB + col * CSB + half * 128);
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
      hp[(half * 3 + 0) * NV + cell] = s0;
      hp[(half * 3 + 1) * NV + cell] = s1;
      hp[(half * 3 + 2) * NV + cell] = s2;
    }
    __syncthreads();
    // feature rows [policy NV | pad to 16 | value0 NV | value1 NV | pad]: whole-wave consecutive 4-byte stores
    constexpr int KP = (NV + 15) / 16 * 16;
    float *fr = feat + (size_t)g * FS;
    for (int e = tid; e < 3 * NV; e += NTH) {
      const int k = e / NV, cell = e - k * NV;
      const float v = hp[k * NV + cell] + hp[(3 + k) * NV + cell];
      fr[(k == 0 ? 0 : KP + (k - 1) * NV) + cell] = fmaxf(v + hw[3 * C + k], 0.0f);
    }
    // (the next board's first barriers order these reads of hp / act before anything overwrites them)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage prefetched last must have landed before the LDS is given back
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

}  // namespace aztw

static std::mutex g_w_mutex;
static bool g_w_attr_done[64];

extern "C" int az_tower_forward_f32w(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                                     const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *u0_dev,
                                     const void *ul_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev,
                                     float *feat_f32_dev, int32_t feat_stride, void *stream) {
  using namespace aztw;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !u0_dev || !bias_dev || !headw_dev || !headb_dev || !feat_f32_dev ||
      (n_conv_layers > 0 && !ul_dev)) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1) || n_conv_layers > MAX_LAYERS) {
    tower_set_error("Winograd f32 tower: n_boards >= 1, an even number of conv layers, at most 40");
    return AZ_EINVAL;
  }
  if (board_size != 15 || channels != 64) { tower_set_error("Winograd f32 tower: 15x15 boards, 64 channels"); return AZ_EINVAL; }
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
  {
    std::lock_guard<std::mutex> lock(g_w_mutex);
    if (!g_w_attr_done[dev]) {
      if (hipFuncSetAttribute((const void *)k_tower_f32w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) {
        tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return AZ_EHIP;
      }
      g_w_attr_done[dev] = true;
    }
  }
  const int grid = n_boards < 256 ? n_boards : 256;  // one workgroup per CU (144 KiB of LDS each)
  const bool use_flags = !x_f32_dev && flags_dev;
  hipLaunchKernelGGL(k_tower_f32w, dim3(grid), dim3(NTH), LDS_BYTES, s, x_f32_dev, (const uint32_t *)packed_ring_dev,
                     (const int4 *)packed_leaf_dev, (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)u0_dev,
                     (const char *)ul_dev, bias_dev, n_conv_layers, headw_dev, headb_dev, feat_f32_dev, (int)feat_stride, tickets);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_f32w launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}
