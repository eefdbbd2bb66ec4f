// az_smallnet.hip -- the reference's two small example networks as ONE float32 kernel each, for MI355X (gfx950).
//
//   examples/tictactoe/model.py      (B,9,3,3):  conv3x3(9->64)+ReLU, conv3x3(64->64)+ReLU, Flatten;
//                                     policy Linear(576->128)+ReLU, Linear(128->9); value Linear(576->64)+ReLU, Linear(64->1), tanh
//   examples/dots_and_boxes/model.py (B,25,4,4): conv3x3(25->64)+ReLU, conv3x3(64->64)+ReLU;
//                                     policy conv1x1(64->2)+ReLU, Flatten, Linear(32->24);
//                                     value conv1x1(64->1)+ReLU, Flatten, Linear(16->64)+ReLU, Linear(64->1), tanh
// No counterpart in the reference beyond the model contract (nn_models/base.py:17-19).  Through PyTorch an evaluation of
// these nets is ~14 launches for a few MFLOP per board; BASELINE configs 1-2 (1,000 / 3,000 concurrent games) are bound
// by exactly that.  Here a workgroup (4 waves) takes a tile of 64 board cells -- 7 tic-tac-toe boards or 4 dots-and-boxes
// boards -- through the whole network with the activations in LDS:
//   both convs are implicit GEMMs D[co][cell] = sum_{tap,ci} W[co][tap][ci] X[ci][cell + shift] on v_mfma_f32_16x16x4_f32
//   (f32 operands, f32 accumulate: the reference's arithmetic; a lane of k-quarter kq reads the 16 bytes of channels
//   16 m + 4 kq .. + 3 once and feeds four MFMAs); wave w owns output channels 16 w .. + 15 for all four 16-cell tiles;
//   a tap that leaves the board (or the tile's own board) reads a zero row; weights come straight from global memory
//   (L2), [tap][16-byte chunk of ci][co 64][4 ci] f32, three k-steps ahead (four registers in rotation);
//   tic-tac-toe's Linear(576 -> 128 | 64) is one more MFMA GEMM over the flattened conv output (boards as columns, the
//   192 hidden units as rows, k = (cell, channel) with the weights permuted to that order by the host); everything
//   smaller (1x1 head convs, the last Linear layers, tanh) is VALU work on LDS operands.
// Input: dense f32 planes (G, C_in, H, W) or the engine's packed leaf batch (8 bitboard states + player per slot) with the
// needs-the-network flags; outputs logits (G, A) and values (G) directly -- one launch per evaluation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>

#include "../../include/az_engine.h"

void tower_set_error(const std::string &msg);  // az_tower.hip (az_tower_last_error)
int tower_device(hipStream_t stream, int *dev);  // az_tower.hip: the device that owns the stream (current device for the NULL stream)

namespace azsmall {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int C = 64, ROW = 4 * C, ZROW = 64;  // trunk width; bytes per cell; index of the all-zero row

template <int GAME> struct Net;
template <> struct Net<AZ_GAME_TICTACTOE> {
  static constexpr int H = 3, W = 3, HW = 9, CIN = 9, CINP = 16, TB = 7, A = 9, SW = 1, NHID = 192;
};
template <> struct Net<AZ_GAME_DAB> {
  static constexpr int H = 4, W = 4, HW = 16, CIN = 25, CINP = 32, TB = 4, A = 24, SW = 2, NHID = 64;
};

__device__ __forceinline__ int act_addr(int cell, int chunk) { return cell * ROW + ((chunk ^ (cell & 15)) << 4); }

// plane value of packed state words (az_games.hip.h: TicTacToe::plane / Dab::plane)
template <int GAME>
__device__ __forceinline__ float plane(const uint32_t *s, int ch, int pos) {
  if (GAME == AZ_GAME_TICTACTOE) return (float)(int)((s[0] >> pos) & 1u) - (float)(int)((s[0] >> (9 + pos)) & 1u);
  const int r = pos >> 2, c = pos & 3;
  if (ch == 0) return (c < 3) ? (float)(int)((s[0] >> (r * 3 + c)) & 1u) : 0.0f;
  if (ch == 1) return (r < 3) ? (float)(int)((s[0] >> (12 + r * 4 + c)) & 1u) : 0.0f;
  if (r < 3 && c < 3) return (float)(int)((s[1] >> (r * 3 + c)) & 1u) - (float)(int)((s[1] >> (9 + r * 3 + c)) & 1u);
  return 0.0f;
}

// One 3x3 conv + bias + ReLU over the tile: src rows of SRC_ROW bytes (SWZ: 256-byte rows, chunk-swizzled), KS k-steps of 16
// input channels per tap; wave w computes output channels 16 w .. + 15 of all 64 cells and writes them to dst.
template <int KS, int SRC_ROW, bool SWZ, int W>
__device__ __forceinline__ void conv3x3(const char *src, const char *__restrict__ wg, const float *__restrict__ bias, char *dst,
                                        int wave, int lane, unsigned long long vm) {
  const int kq = lane >> 4, r = lane & 15;
  const char *wlane = wg + kq * (C * 16) + (wave * 16 + r) * 16;
  f32x4 acc[4];
  {
    const f32x4 bv = *(const f32x4 *)(bias + wave * 16 + kq * 4);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = bv;
  }
  // weights: four registers in rotation, requested three k-steps ahead (everything unrolled: the compiler counts the waits)
  constexpr int NK = 9 * KS;
  f32x4 wq[4];
#pragma unroll
  for (int i = 0; i < 3 && i < NK; ++i) wq[i] = *(const f32x4 *)(wlane + (size_t)i * (4 * C * 16));
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
    int q[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) q[ct] = ((vm >> (9 * ct + tap)) & 1ull) ? ct * 16 + r + d : ZROW;
#pragma unroll
    for (int m = 0; m < KS; ++m) {
      const int kk = tap * KS + m;
      if (kk + 3 < NK) wq[(kk + 3) & 3] = *(const f32x4 *)(wlane + (size_t)(kk + 3) * (4 * C * 16));
      const f32x4 wa = wq[kk & 3];
      f32x4 xb[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        xb[ct] = *(const f32x4 *)(src + (SWZ ? act_addr(q[ct], 4 * m + kq) : q[ct] * SRC_ROW + (4 * m + kq) * 16));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], xb[ct][t], acc[ct], 0, 0, 0);
    }
  }
  // acc[ct][j] = channel 16 w + 4 kq + j of cell 16 ct + r
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[ct][j], 0.0f);
    *(f32x4 *)(dst + act_addr(ct * 16 + r, wave * 4 + kq)) = v;
  }
}

// x: (G, CIN, H, W) f32 or nullptr (packed: ring u32 (G, 8, SW), leafrec int4 (G): .w = player); flags: u8 (G) or nullptr.
// w1: [9][CINP/4][64][4] f32, b1 [64]; w2: [9][16][64][4] f32, b2 [64]; head: the game's blob (see az_smallnet_forward_f32).
template <int GAME>
__global__ __launch_bounds__(256) void k_smallnet_f32(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                      const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                      const char *__restrict__ w1, const float *__restrict__ b1,
                                                      const char *__restrict__ w2, const float *__restrict__ b2,
                                                      const float *__restrict__ head, float *__restrict__ logits,
                                                      float *__restrict__ values) {
  using Q = Net<GAME>;
  constexpr int HW = Q::HW, TB = Q::TB, CINP = Q::CINP, A = Q::A, SW = Q::SW;
  __shared__ __attribute__((aligned(16))) char act0[65 * CINP * 4];
  __shared__ __attribute__((aligned(16))) char act1[65 * ROW];
  __shared__ __attribute__((aligned(16))) char act2[65 * ROW];
  __shared__ __attribute__((aligned(16))) float hid[GAME == AZ_GAME_TICTACTOE ? 16 * 192 : 64 * 4 + 4 * 64];
  __shared__ int slots[TB];
  __shared__ int any_on;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, r = lane & 15;
  const int g0 = blockIdx.x * TB;
  if (tid == 0) any_on = 0;
  __syncthreads();
  if (tid < TB) {
    const int g = g0 + tid;
    const bool on = g < G && (flags == nullptr || flags[g] != 0);
    slots[tid] = on ? g : -1;
    if (on) any_on = 1;
  }
  for (int i = tid; i < 65 * CINP; i += 256) ((float *)act0)[i] = 0.0f;  // padding channels, unused cells, the zero row
  for (int i = tid; i < 64; i += 256) {
    ((float *)act1)[ZROW * C + i] = 0.0f;
    ((float *)act2)[ZROW * C + i] = 0.0f;
  }
  __syncthreads();
  if (!any_on) return;
  // ---- input planes -> act0[cell][ci]
  for (int e = tid; e < TB * HW * Q::CIN; e += 256) {
    const int cell = e % (TB * HW), ci = e / (TB * HW), b = cell / HW, pos = cell % HW, g = slots[b];
    float v = 0.0f;
    if (g >= 0) {
      if (x != nullptr) v = x[((size_t)g * Q::CIN + ci) * HW + pos];
      else if (ci == Q::CIN - 1) v = (float)leafrec[g].w;
      else {
        constexpr int CS = (Q::CIN - 1) / 8;
        v = plane<GAME>(ring + ((size_t)g * 8 + ci / CS) * SW, ci % CS, pos);
      }
    }
    ((float *)act0)[cell * CINP + ci] = v;
  }
  // tap validity of this lane's four cells (cell = 16 ct + r): inside its own board, and the cell itself in use
  unsigned long long vm = 0;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int cell = ct * 16 + r, pos = cell % HW, row = pos / Q::W, col = pos % Q::W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int rr = row + t / 3 - 1, cc = col + t % 3 - 1;
      if (cell < TB * HW && rr >= 0 && rr < Q::H && cc >= 0 && cc < Q::W) vm |= 1ull << (9 * ct + t);
    }
  }
  __syncthreads();
  conv3x3<CINP / 16, CINP * 4, false, Q::W>(act0, w1, b1, act1, wave, lane, vm);
  __syncthreads();
  conv3x3<4, ROW, true, Q::W>(act1, w2, b2, act2, wave, lane, vm);
  __syncthreads();

  if (GAME == AZ_GAME_TICTACTOE) {
    // head blob: W1' [36 k-steps][4 kq][192][4] (rows 0..127 policy hidden, 128..191 value hidden; k = pos * 64 + c),
    // b1' [192], Wp2 [9][128], bp2 [9], wv2 [64], bv2 [1]
    const char *W1 = (const char *)head;
    const float *b1h = head + 36 * 4 * 192 * 4, *Wp2 = b1h + 192, *bp2 = Wp2 + 9 * 128, *wv2 = bp2 + 9, *bv2 = wv2 + 64;
    f32x4 acc[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[nt] = *(const f32x4 *)(b1h + wave * 48 + nt * 16 + kq * 4);
    const int rowbase = r < TB ? r * HW : -1;  // column b = r of the GEMM: board r of the tile
    const char *wlane = W1 + (size_t)kq * (192 * 16) + (wave * 48 + r) * 16;
    // 36 k-steps, unrolled; the weight fragments (three per k-step) run two k-steps ahead through three register sets
    constexpr int NKH = HW * 4;
    f32x4 wq[3][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) wq[i][nt] = *(const f32x4 *)(wlane + (size_t)i * (4 * 192 * 16) + nt * 256);
#pragma unroll
    for (int ks = 0; ks < NKH; ++ks) {
      const int pos = ks >> 2, m = ks & 3;
      const int row = rowbase >= 0 ? rowbase + pos : ZROW;
      if (ks + 2 < NKH) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) wq[(ks + 2) % 3][nt] = *(const f32x4 *)(wlane + (size_t)(ks + 2) * (4 * 192 * 16) + nt * 256);
      }
      const f32x4 xb = *(const f32x4 *)(act2 + act_addr(row, 4 * m + kq));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[ks % 3][nt][t], xb[t], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {  // acc[nt][j] = hidden unit 48 w + 16 nt + 4 kq + j of board r
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[nt][j], 0.0f);
      *(f32x4 *)(hid + r * 192 + wave * 48 + nt * 16 + kq * 4) = v;
    }
    __syncthreads();
    if (tid < TB * A) {
      const int b = tid / A, a = tid % A, g = slots[b];
      if (g >= 0) {
        float s = bp2[a];
        for (int j = 0; j < 128; ++j) s = fmaf(hid[b * 192 + j], Wp2[a * 128 + j], s);
        logits[(size_t)g * A + a] = s;
      }
    } else if (tid >= 128 && tid < 128 + TB) {
      const int b = tid - 128, g = slots[b];
      if (g >= 0) {
        float s = bv2[0];
        for (int j = 0; j < 64; ++j) s = fmaf(hid[b * 192 + 128 + j], wv2[j], s);
        values[g] = tanhf(s);
      }
    }
  } else {
    // head blob: hw [3][64] (policy conv 2 rows, value conv 1 row), hb [3], Wp [24][32], bp [24], W1 [64][16], b1 [64],
    // w2 [64], b2 [1]
    const float *hwt = head, *hb = hwt + 3 * 64, *Wp = hb + 3, *bp = Wp + 24 * 32, *W1 = bp + 24, *b1h = W1 + 64 * 16,
                *w2h = b1h + 64, *b2h = w2h + 64;
    float *f = hid, *vh = hid + 64 * 4;  // f[cell][4]: the three 1x1-conv outputs; vh[board][64]: the value head's hidden layer
    if (tid < 192) {
      const int cell = tid & 63, k = tid >> 6;
      float s = hb[k];
      for (int c4 = 0; c4 < 16; ++c4) {
        const f32x4 a = *(const f32x4 *)(act2 + act_addr(cell, c4));
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(a[j], hwt[k * 64 + c4 * 4 + j], s);
      }
      f[cell * 4 + k] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    {
      const int b = tid >> 6, j = tid & 63;  // value hidden: Linear(16 -> 64) + ReLU on the value conv's plane
      float s = b1h[j];
      for (int pos = 0; pos < 16; ++pos) s = fmaf(f[(b * 16 + pos) * 4 + 2], W1[j * 16 + pos], s);
      vh[b * 64 + j] = fmaxf(s, 0.0f);
    }
    if (tid < TB * A) {  // policy: Linear(32 -> 24) on Flatten((2, 4, 4)): input index c * 16 + pos
      const int b = tid / A, a = tid % A, g = slots[b];
      if (g >= 0) {
        float s = bp[a];
        for (int i = 0; i < 32; ++i) s = fmaf(f[(b * 16 + (i & 15)) * 4 + (i >> 4)], Wp[a * 32 + i], s);
        logits[(size_t)g * A + a] = s;
      }
    }
    __syncthreads();
    if (tid < TB) {
      const int g = slots[tid];
      if (g >= 0) {
        float s = b2h[0];
        for (int j = 0; j < 64; ++j) s = fmaf(vh[tid * 64 + j], w2h[j], s);
        values[g] = tanhf(s);
      }
    }
  }
}

}  // namespace azsmall

extern "C" {

int32_t az_smallnet_head_floats(int32_t game) {
  if (game == AZ_GAME_TICTACTOE) return 36 * 4 * 192 * 4 + 192 + 9 * 128 + 9 + 64 + 1;
  if (game == AZ_GAME_DAB) return 3 * 64 + 3 + 24 * 32 + 24 + 64 * 16 + 64 + 64 + 1;
  return 0;
}

int az_smallnet_forward_f32(int32_t game, int32_t n_boards, const float *x_f32_dev, const void *packed_ring_dev,
                            const void *packed_leaf_dev, const void *flags_dev, const void *w1_dev, const float *b1_dev,
                            const void *w2_dev, const float *b2_dev, const float *head_dev, float *logits_dev, float *values_dev,
                            void *stream) {
  using namespace azsmall;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !head_dev || !logits_dev ||
      !values_dev) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || (game != AZ_GAME_TICTACTOE && game != AZ_GAME_DAB)) { tower_set_error("n_boards >= 1; game tic-tac-toe or dots-and-boxes"); return AZ_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cur = 0;
  if (tower_device(s, &dev) != 0) { tower_set_error("cannot tell the stream's device"); return AZ_EHIP; }
  if (hipGetDevice(&cur) != hipSuccess) { tower_set_error("hipGetDevice failed"); return AZ_EHIP; }
  struct DeviceGuard {  // launch on the device that owns the stream, whatever the caller's current device is
    int back;
    DeviceGuard(int want, int cur_) : back(want != cur_ ? cur_ : -1) { if (back >= 0) (void)hipSetDevice(want); }
    ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
  } guard(dev, cur);
  const uint8_t *fl = x_f32_dev ? nullptr : (const uint8_t *)flags_dev;
  if (game == AZ_GAME_TICTACTOE) {
    const int tb = Net<AZ_GAME_TICTACTOE>::TB;
    hipLaunchKernelGGL((k_smallnet_f32<AZ_GAME_TICTACTOE>), dim3((n_boards + tb - 1) / tb), dim3(256), 0, s, x_f32_dev,
                       (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev, fl, (int)n_boards, (const char *)w1_dev, b1_dev,
                       (const char *)w2_dev, b2_dev, head_dev, logits_dev, values_dev);
  } else {
    const int tb = Net<AZ_GAME_DAB>::TB;
    hipLaunchKernelGGL((k_smallnet_f32<AZ_GAME_DAB>), dim3((n_boards + tb - 1) / tb), dim3(256), 0, s, x_f32_dev,
                       (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev, fl, (int)n_boards, (const char *)w1_dev, b1_dev,
                       (const char *)w2_dev, b2_dev, head_dev, logits_dev, values_dev);
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_smallnet_f32 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

}  // extern "C"
