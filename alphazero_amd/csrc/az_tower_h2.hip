// az_tower_h2.hip -- "f16x2": the GomokuModel-family conv tower at f32-level accuracy on the F16 MFMA pipe (gfx950).
//
// EXPLORATORY / SECONDARY (not the reference's arithmetic; the headline stays az_tower_f32.hip).  The f32 tower sits at
// 86-87 % of a 157 TFLOP/s roof: the f32 MFMA rate is 1/16 of the 16-bit rate.  Here every f32 operand is split into two f16
// halves, x = x1 + x2 with x1 = f16(x), x2 = f16(x - x1) (22 significant bits; f16 x f16 products are exact in f32), and a
// product w x is evaluated as  w1 x1 + w1 x2 + w2 x1  (the dropped w2 x2 is 2^-22 of the product: the size of the
// representation error itself), three v_mfma_f32_16x16x32_f16 with f32 accumulation in place of eight
// v_mfma_f32_16x16x4_f32: 48 instead of 256 MFMA cycles per 16x16x32 block.
//   * range: f16 overflows at 65,504 and loses precision below 6e-5, so activations are stored x SA = 16 and every
//     layer's weights x a power of two SW_l chosen by the host (max |w| SW_l <= 2^14); both scalings are exact and are
//     undone on the f32 accumulators (y SA = acc / SW_l).  An activation above 4,094 cannot be represented: the kernel
//     then raises a flag (az_tower_h2_overflow) and the caller must fall back to the f32 tower.
//   * structure: one workgroup (4 waves) owns one board for the whole tower, two workgroups per CU.  Activations live in
//     LDS as act[cell][half][64 ch] f16 -- 256 B per cell, the f32 tower's footprint -- in padded rows (272 B: the 16 cells
//     of a fragment read hit 64 distinct banks), + a zero row for taps that leave the board.  A conv is the implicit GEMM
//     D[co][cell] = sum W[co][tap, ci] X[tap, ci][cell]: wave w owns cell tiles 8 (w & 1) .. + 7 (16 cells each; 225 = 14
//     tiles + one cell, the last tile is padding but for that cell) x the two 16-channel tiles 2 (w >> 1), + 1: 16
//     accumulator tiles; per k-step of 32 channels and tile: two ds_read_b128 (both halves) feed six MFMAs.  Weights
//     [tap][k-step][co tile][half][lane][8] f16 come straight from L2 one k-step ahead.  Layers run in place: barrier,
//     epilogue (scale, ReLU, split, store; a block's first epilogue reads the activations it overwrites -- the block's
//     residual -- and starts the second conv's accumulators at bias + residual), barrier.
//   * the 1x1 head convs + BN + ReLU are summed on the VALU from the last activations; feature rows as az_tower_f32.hip
//     writes them, so az_heads_forward_f32 finishes the evaluation.
// Boards are handed out by the library's ticket counter; input: the engine's packed leaf batch (+ flags) or dense f32 planes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "../../include/az_engine.h"

int tower_device(hipStream_t stream, int *dev);             // az_tower.hip
unsigned int *tower_tickets(int dev, hipStream_t stream);   // az_tower.hip
void tower_set_error(const std::string &msg);               // az_tower.hip

namespace azh2 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int C = 64;                 // trunk width
constexpr int ROW = 2 * C * 2 + 16;   // bytes of one cell's row: [half 0: 64 f16][half 1: 64 f16] + 16 B of padding
constexpr float SA = 16.0f;           // activations are stored x SA

template <int N>
struct Geo {
  static constexpr int NV = N * N, NTILE = (NV + 15) / 16, TPW = (NTILE + 1) / 2;
  static constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = 8 * SWB, KP = (NV + 15) / 16 * 16;
  static constexpr int ACT_BYTES = (NV + 1) * ROW;  // + the zero row (cell index NV)
  static constexpr int HW_FLOATS = 3 * C + 4;
  static constexpr size_t lds_bytes() { return (size_t)ACT_BYTES + (RGW + 4) * 4 + HW_FLOATS * 4; }
};

__device__ __forceinline__ void split(float v, _Float16 &a, _Float16 &b) {
  a = (_Float16)v;
  b = (_Float16)(v - (float)a);
}

// One conv layer's k loop: KS k-steps of 32 input channels per tap (1 for the stem, 2 for a 64 -> 64 layer).
// acc[pt][ct] += W x over the 9 taps; `w` points at this layer's packed weights, `wnext` at the next layer's (its first
// fragments are requested under this layer's last k-step), `a` holds the fragments of the first k-step on entry and of
// the next layer's first k-step on exit.
template <int N, int KS, bool X2 = true>
__device__ __forceinline__ void conv_layer(const char *act, const char *__restrict__ w, const char *__restrict__ wnext, const int cg,
                                           const int lane, const int ntiles, const int (&cellb)[Geo<N>::TPW], const unsigned long long (&vm)[2],
                                           f4 (&acc)[Geo<N>::TPW][2], h8 (&a)[2][2]) {
  using Q = Geo<N>;
  constexpr int TPW = Q::TPW, STEPS = 9 * KS;
  const int kgoff = (lane >> 4) * 16;
  // fragment of (tap, k-step) for co tile 2 cg + ct, half s: 1 KiB per fragment, [tap][ks][ct 4][s 2][lane][8]
  auto wptr = [&](const char *base, int step) { return base + (size_t)(step * 4 + 2 * cg) * 2048 + lane * 16; };
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    const int shift = ((t / 3 - 1) * N + (t % 3 - 1)) * ROW;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int step = t * KS + ks;
      h8 an[2][2];
      {
        const char *p = step + 1 < STEPS ? wptr(w, step + 1) : wptr(wnext, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int s = 0; s < 2; ++s) an[ct][s] = *(const h8 *)(p + (ct * 2 + s) * 1024);
      }
#pragma unroll
      for (int pt = 0; pt < TPW; ++pt) {
        if (pt < ntiles) {  // (wave-uniform: the odd half of a 15-tile board has 7 tiles)
          const bool on = (vm[pt >> 2] >> (9 * (pt & 3) + t)) & 1ull;
          const char *row = act + (on ? cellb[pt] + shift : Q::NV * ROW) + ks * 64 + kgoff;
          const h8 b0 = *(const h8 *)row, b1 = *(const h8 *)(row + 128);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ct][1], b0, acc[pt][ct], 0, 0, 0);
            if (X2) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ct][0], b1, acc[pt][ct], 0, 0, 0);  // (x2 == 0: adds nothing)
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ct][0], b0, acc[pt][ct], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s) a[ct][s] = an[ct][s];
    }
  }
}

template <int N>
__global__ __launch_bounds__(256, 2) void k_tower_h2(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                     const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                     const char *__restrict__ w0, const char *__restrict__ wl,
                                                     const float *__restrict__ bias_s, const float *__restrict__ sw, int L,
                                                     const float *__restrict__ headw, const float *__restrict__ headb,
                                                     float *__restrict__ feat, int FS, unsigned int *tickets, int *overflow) {
  using Q = Geo<N>;
  constexpr int NV = Q::NV, TPW = Q::TPW, WC = Q::WC, SWB = Q::SWB, RGW = Q::RGW, KP = Q::KP;
  constexpr int WL_BYTES = 9 * 2 * 4 * 2 * 1024;  // one 64 -> 64 layer's packed weights
  extern __shared__ __attribute__((aligned(16))) char smem_h2[];
  char *act = smem_h2;
  uint32_t *rg = (uint32_t *)(smem_h2 + Q::ACT_BYTES);  // the board's packed input: RGW ring words + the player word
  int *slot = (int *)rg + RGW + 1;                       // this round's board (uniform hand-over from thread 0)
  float *hw = (float *)(rg + RGW + 4);                   // head filters [3][64] + head biases [3]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave >> 1;                                           // channel group: co tiles 2 cg, 2 cg + 1
  const int chalf = (wave ^ (int)(blockIdx.x >> 8)) & 1;              // which half of the cell tiles (flipped between the two workgroups of a CU)
  const int tile0 = chalf * TPW, ntiles = min(TPW, Q::NTILE - tile0);
  const int kg = lane >> 4;

  for (int i = tid; i < Q::ACT_BYTES / 16; i += 256) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // incl. the zero row
  for (int i = tid; i < 3 * C + 3; i += 256) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];

  int cellb[TPW];             // byte offset of this lane's cell of tile pt (cells past the board: the zero row)
  unsigned long long vm[2] = {0ull, 0ull};  // vm[pt / 4] bit 9 (pt % 4) + tap: that tap of the cell is on the board
#pragma unroll
  for (int pt = 0; pt < TPW; ++pt) {
    const int cell = (tile0 + pt) * 16 + (lane & 15), row = cell / N, col = cell % N;
    cellb[pt] = min(cell, NV) * ROW;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int rr = row + t / 3 - 1, cc = col + t % 3 - 1;
      if (cell < NV && rr >= 0 && rr < N && cc >= 0 && cc < N) vm[pt >> 2] |= 1ull << (9 * (pt & 3) + t);
    }
  }
  static_assert(Geo<N>::TPW <= 8, "tap masks");

  h8 a[2][2];  // weight fragments of the k-step about to run (global -> registers, one k-step ahead)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[ct][s] = *(const h8 *)(w0 + (size_t)(2 * cg) * 2048 + lane * 16 + (ct * 2 + s) * 1024);

  int ticket = (int)blockIdx.x;
  bool ovf = false;
  for (;;) {
    __syncthreads();  // the previous board's head phase is done with `act`, `rg`, `slot`
    if (tid == 0) {
      while (ticket < G && flags != nullptr && flags[ticket] == 0) ticket = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // terminal leaf
      slot[0] = ticket;
    }
    __syncthreads();
    const int g = slot[0];
    if (g >= G) break;
    if (tid == 0) ticket = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // the next board's ticket
    // ---- input: 9 planes -> channels 0..8 of a 32-channel k-step, x SA, both halves
    if (x == nullptr) {
      for (int i = tid; i < RGW + 1; i += 256) rg[i] = i < RGW ? ring[(size_t)g * RGW + i] : (uint32_t)leafrec[g].w;
      __syncthreads();
    }
    for (int cell = tid; cell < NV; cell += 256) {
      _Float16 c0[32], c1[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) { c0[k] = (_Float16)0.0f; c1[k] = (_Float16)0.0f; }
      if (x != nullptr) {
        const float *xr = x + (size_t)g * 9 * NV + cell;
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) {
          const float v = xr[cc * NV] * SA;
          ovf |= !(fabsf(v) <= 65504.0f);
          split(v, c0[cc], c1[cc]);
        }
      } else {
        const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          c0[k] = (_Float16)(SA * ((float)(int)((rg[k * SWB + wi] >> sh) & 1u) - (float)(int)((rg[k * SWB + WC + wi] >> sh) & 1u)));
        c0[8] = (_Float16)((int)rg[RGW] > 0 ? SA : -SA);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *(h8 *)(act + cell * ROW + q * 16) = (h8){c0[8 * q], c0[8 * q + 1], c0[8 * q + 2], c0[8 * q + 3], c0[8 * q + 4], c0[8 * q + 5], c0[8 * q + 6], c0[8 * q + 7]};
        *(h8 *)(act + cell * ROW + 128 + q * 16) = (h8){c1[8 * q], c1[8 * q + 1], c1[8 * q + 2], c1[8 * q + 3], c1[8 * q + 4], c1[8 * q + 5], c1[8 * q + 6], c1[8 * q + 7]};
      }
    }
    __syncthreads();

    f4 acc[TPW][2];
    // accumulators start at the layer's (pre-scaled) bias: this lane's rows are channels 16 (2 cg + ct) + 4 kg .. + 3
    auto init = [&](int layer) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f4 bv = *(const f4 *)(bias_s + layer * C + (2 * cg + ct) * 16 + kg * 4);
#pragma unroll
        for (int pt = 0; pt < TPW; ++pt) acc[pt][ct] = bv;
      }
    };
    // y SA = relu(acc / SW_layer) -> two f16 halves, stored in place.  with_res: the activations being overwritten are the
    // block's residual: read them first and start the NEXT layer's accumulators at (bias + residual) x SW_next x SA.
    auto epilogue = [&](int layer, bool with_res) {
      const float inv = 1.0f / sw[layer];
      const float swn = with_res ? sw[layer + 1] : 0.0f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f4 bn = {0.0f, 0.0f, 0.0f, 0.0f};
        if (with_res) bn = *(const f4 *)(bias_s + (layer + 1) * C + (2 * cg + ct) * 16 + kg * 4);
#pragma unroll
        for (int pt = 0; pt < TPW; ++pt) {
          const int cell = (tile0 + pt) * 16 + (lane & 15);
          if (pt < ntiles && cell < NV) {
            char *p = act + cell * ROW + ((2 * cg + ct) * 16 + kg * 4) * 2;
            h4 o0, o1;
            f4 r;
            if (with_res) {
              const h4 x0 = *(const h4 *)p, x1 = *(const h4 *)(p + 128);
#pragma unroll
              for (int j = 0; j < 4; ++j) r[j] = (float)x0[j] + (float)x1[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float y = fmaxf(acc[pt][ct][j] * inv, 0.0f);
              ovf |= !(y <= 65504.0f);
              _Float16 ya, yb;
              split(y, ya, yb);
              o0[j] = ya;
              o1[j] = yb;
            }
            *(h4 *)p = o0;
            *(h4 *)(p + 128) = o1;
            if (with_res) acc[pt][ct] = bn + r * swn;
          } else if (with_res) {
            acc[pt][ct] = bn;
          }
        }
      }
    };

    // ---- stem
    init(0);
    // (the packed input is +-16 / 0: exact in its first f16 half, so the stem needs two products, not three)
    if (x == nullptr) conv_layer<N, 1, false>(act, w0, L > 0 ? wl : w0, cg, lane, ntiles, cellb, vm, acc, a);
    else conv_layer<N, 1, true>(act, w0, L > 0 ? wl : w0, cg, lane, ntiles, cellb, vm, acc, a);
    __syncthreads();  // every wave has read the layer's input: the outputs may overwrite it
    epilogue(0, false);
    __syncthreads();
    // ---- residual blocks: conv, ReLU, conv + x, ReLU
    for (int l = 0; l < L; l += 2) {
      init(1 + l);
      conv_layer<N, 2>(act, wl + (size_t)l * WL_BYTES, wl + (size_t)(l + 1) * WL_BYTES, cg, lane, ntiles, cellb, vm, acc, a);
      __syncthreads();
      epilogue(1 + l, true);  // (also starts acc of the block's second conv at bias + residual)
      __syncthreads();
      conv_layer<N, 2>(act, wl + (size_t)(l + 1) * WL_BYTES, l + 2 < L ? wl + (size_t)(l + 2) * WL_BYTES : w0, cg, lane, ntiles, cellb,
                       vm, acc, a);
      __syncthreads();
      epilogue(2 + l, false);
      __syncthreads();
    }
    // ---- the three 1x1 head convs (+ folded BN, ReLU): one thread per cell, f32 on the VALU
    for (int cell = tid; cell < NV; cell += 256) {
      float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
      const char *p = act + cell * ROW;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const h8 x0 = *(const h8 *)(p + q * 16), x1 = *(const h8 *)(p + 128 + q * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = ((float)x0[j] + (float)x1[j]) * (1.0f / SA);
          const int c = q * 8 + j;
          t0 = fmaf(v, hw[c], t0);
          t1 = fmaf(v, hw[C + c], t1);
          t2 = fmaf(v, hw[2 * C + c], t2);
        }
      }
      float *fr = feat + (size_t)g * FS;
      fr[cell] = fmaxf(t0 + hw[3 * C + 0], 0.0f);
      fr[KP + cell] = fmaxf(t1 + hw[3 * C + 1], 0.0f);
      fr[KP + NV + cell] = fmaxf(t2 + hw[3 * C + 2], 0.0f);
    }
  }
  if (ovf && overflow != nullptr) atomicOr(overflow, 1);
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

static std::mutex g_mutex;
static unsigned int g_attr_done[64];

}  // namespace azh2

extern "C" {

int az_tower_forward_h2(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                        const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *w0_dev,
                        const void *wl_dev, const float *bias_scaled_dev, const float *sw_dev, const float *headw_dev,
                        const float *headb_dev, float *feat_f32_dev, int32_t feat_stride, int32_t *overflow_dev, void *stream) {
  using namespace azh2;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !bias_scaled_dev || !sw_dev || !headw_dev || !headb_dev ||
      !feat_f32_dev || (n_conv_layers > 0 && !wl_dev)) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1)) { tower_set_error("n_boards >= 1, even n_conv_layers"); return AZ_EINVAL; }
  if ((board_size != 9 && board_size != 15) || channels != 64) { tower_set_error("f16x2 tower: board_size 9 or 15, 64 channels"); return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { tower_set_error("feat_stride below az_heads_feat_stride(board_size)"); return AZ_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cur = 0;
  if (tower_device(s, &dev) != 0 || dev < 0 || dev >= 64) { tower_set_error("cannot tell the stream's device"); return AZ_EHIP; }
  if (hipGetDevice(&cur) != hipSuccess) { tower_set_error("hipGetDevice failed"); return AZ_EHIP; }
  struct Guard {
    int back;
    Guard(int want, int cur_) : back(want != cur_ ? cur_ : -1) { if (back >= 0) (void)hipSetDevice(want); }
    ~Guard() { if (back >= 0) (void)hipSetDevice(back); }
  } guard(dev, cur);
  unsigned int *tickets = tower_tickets(dev, s);
  if (tickets == nullptr) return AZ_EHIP;
  const bool use_flags = !x_f32_dev && flags_dev;
  auto launch = [&](auto kern, size_t lds, int variant) -> int {
    {
      std::lock_guard<std::mutex> lock(g_mutex);
      if (!(g_attr_done[dev] & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
          return AZ_EHIP;
        }
        g_attr_done[dev] |= 1u << variant;
      }
    }
    const int grid = n_boards < 512 ? n_boards : 512;  // two workgroups per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, x_f32_dev, (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), (int)n_boards, (const char *)w0_dev, (const char *)wl_dev,
                       bias_scaled_dev, sw_dev, (int)n_conv_layers, headw_dev, headb_dev, feat_f32_dev, (int)feat_stride, tickets,
                       (int *)overflow_dev);
    return AZ_OK;
  };
  const int rc = board_size == 15 ? launch(k_tower_h2<15>, Geo<15>::lds_bytes(), 0) : launch(k_tower_h2<9>, Geo<9>::lds_bytes(), 1);
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_h2 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

}  // extern "C"
