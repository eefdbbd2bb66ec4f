// az_tower_h1.hip -- the 15x15 / 64-channel GomokuModel-family conv tower on the 16-bit MFMA pipe with ROW TILES (gfx950).
//
// The throughput-mode tower of az_tower.hip (bf16, 32-cell tiles) spends 12 % of its MFMAs on the 31 padding cells of its eighth
// tile and 53 VALU instructions per 48 MFMAs on tap masks and swizzled fragment addresses.  This kernel removes both by
// making a cell tile ONE BOARD ROW: 15 cells + one lane on a zero column.
//   * LDS image: act[row -1 .. 15][col -1 .. 14][64 ch], rows of 16 cells -- the zero column at col -1 of row r is also col 15
//     of row r - 1, rows -1 and 15 are zero -- 160 bytes per cell (128 of data + 32 of padding).  A tap moves all 16 lanes of
//     a tile by the same number of cells and never across a gap, so (i) there are no tap masks: the halo is real zeros, (ii)
//     every B-fragment address is ONE per-lane base + an instruction immediate (tile, tap, k-step), zero VALU in the k loop,
//     and (iii) with the 160-byte pitch the 16 lanes of each ds_read_b128 group always fall on 16 different bank quads.
//   * 15 tiles of 16 x 4 channel tiles of 16 per board (6.25 % padding columns): wave w owns cell tiles 8 (w & 1) .. + 7 (the
//     odd half has 7; the halves are swapped between the two workgroups of a CU so every SIMD carries 15) x channel tiles
//     2 (w >> 1), + 1, whose rows the host orders so that a lane's 2 x 4 results are eight CONSECUTIVE channels (one 16-byte
//     store per tile and layer).  Per k-step of 32 channels: 8 ds_read_b128 feed 16 v_mfma_f32_16x16x32; the weights
//     [step][co tile 4][lane][8] come straight from L2 into registers two k-steps ahead (no LDS traffic for A at all).
//   * the stem is THREE steps, one per kernel row: the input expansion writes a cell's k-step as [left neighbour's 9 planes |
//     its own | right neighbour's] (k = 9 tx + plane, 27 of 32 used), so a row's three taps are one MFMA step.
//   * everything outside the k loop that can be an MFMA is one (VALU instructions issue at a fraction of their rate beside the
//     partner workgroup's MFMA stream): the block's residual add (a 0 / 1 selector A fragment routes the eight channels a
//     lane reads back -- that tile's B fragment of k-step w >> 1 -- into its accumulator rows, C = the next layer's bias) and the
//     three 1x1 head convs (the filters as rows 0 .. 2 of an A fragment, two MFMAs per tile).
//   * boards come from the library's ticket counter ONE BOARD AHEAD: ticket, flag byte and packed input of the next board are
//     fetched under the current board's layers; biases and scales sit in LDS.
//   * element type: bf16 (no scaling; bf16 feature rows for az_heads_forward), or f16 with exact power-of-two scalings
//     (activations x 16; a layer's weights x a power of two only when they are all tiny; undone on the f32 accumulators; an
//     activation above 4,094 raises the flag; f32 feature rows for az_heads_forward_f32): 8x the precision of bf16 at the same
//     MFMA rate.
// Layers run in place (barrier, epilogue, barrier).  Input: the engine's packed leaf batch (+ flags) or dense f32 planes.
// No counterpart in the reference beyond the model contract (nn_models/base.py:17-19; network: examples/gomoku/model.py:41-90).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <type_traits>

#include "../../include/az_engine.h"

int tower_device(hipStream_t stream, int *dev);             // az_tower.hip
unsigned int *tower_tickets(int dev, hipStream_t stream);   // az_tower.hip
void tower_set_error(const std::string &msg);               // az_tower.hip

namespace azh1 {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char *gptr;

#ifndef AZ_H1_ABLATE
#define AZ_H1_ABLATE 0  // timing diagnostics only (wrong results): 1 = no weight loads, 2 = no fragment reads, 3 = neither
#endif
// Diagnostic build only (-DAZ_H1_STAMP=1): wave-level s_memtime stamps of every workgroup's 10th board, read back with
// az_h1_stamps().  The product build executes no stamp.
#ifndef AZ_H1_STAMP
#define AZ_H1_STAMP 0
#endif
#if AZ_H1_STAMP
__device__ unsigned long long g_h1_stamp[512 * 4 * 32];
#define STAMP(k) do { if (nboard == 10) tsv[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#ifndef AZ_H1_PRIO
#define AZ_H1_PRIO 0
#endif

constexpr int C = 64, N = 15, NV = N * N;
constexpr int PITCH = 160;                 // bytes per cell: 64 x 16 bit + 32 of padding (conflict-free fragment reads)
constexpr int NCELL = 17 * 16 + 2;         // idx(r, c) = (r + 1) * 16 + c + 1, r = -1 .. 15, c = -1 .. 14 (+ the two a corner tap of the pad lane touches)
constexpr int ACT_BYTES = NCELL * PITCH;
constexpr int NTILE = 15, TPW = 8;
constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = 8 * SWB, KP = (NV + 15) / 16 * 16;
constexpr int DUMP_BYTES = TPW * 16 * PITCH;  // where the pad lanes' epilogue stores go (same immediates as the real ones)
constexpr int RG_BYTES = ((RGW + 1) * 4 + 15) / 16 * 16;  // one board's packed input: RGW ring words + the player word
// act | dump | two packed inputs (this board's, the next one's) | hand-over words | head filter fragments | residual selector fragments | per layer: 64 biases + 1 / SW + SW
static size_t lds_bytes(int n_conv_layers) { return (size_t)ACT_BYTES + DUMP_BYTES + 2 * RG_BYTES + 16 + 4096 + (size_t)(n_conv_layers + 1) * (C + 4) * 4; }
constexpr int MAX_LAYERS = 64;  // (the bias table grows with depth: two workgroups fit a CU's 160 KiB up to 44 conv layers, one beyond)

typedef float f2 __attribute__((ext_vector_type(2)));

struct F16 {
  typedef _Float16 T;
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static constexpr float SA = 16.0f, MAXV = 65504.0f;
  static constexpr bool SCALED = true;
  static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  // two f32 -> packed f16 (round to nearest even), ReLU on the packed pair
  static __device__ __forceinline__ v2 cvt_relu(f2 y) {
    const v2 z = {(_Float16)0.0f, (_Float16)0.0f};
    return __builtin_elementwise_max(__builtin_convertvector(y, v2), z);
  }
  static __device__ __forceinline__ v2 vmax(v2 a, v2 b) { return __builtin_elementwise_max(a, b); }
  static __device__ __forceinline__ bool over(v2 m) { return !((float)m[0] <= MAXV) || !((float)m[1] <= MAXV); }
};
struct BF16 {
  typedef __bf16 T;
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static constexpr float SA = 1.0f, MAXV = 3.0e38f;
  static constexpr bool SCALED = false;
  static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  // two f32 -> packed bf16 (one v_cvt_pk_bf16_f32); ReLU as a signed 16-bit max with 0 (a negative bf16 is a negative int16)
  static __device__ __forceinline__ v2 cvt_relu(f2 y) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 z = {0, 0};
    return __builtin_bit_cast(v2, __builtin_elementwise_max(__builtin_bit_cast(s2, __builtin_convertvector(y, v2)), z));
  }
  static __device__ __forceinline__ v2 vmax(v2 a, v2) { return a; }
  static __device__ __forceinline__ bool over(v2) { return false; }
};

__device__ __forceinline__ int cell_idx(int r, int c) { return (r + 1) * 16 + c + 1; }

// One conv layer's k loop, fully unrolled: KS k-steps of 32 input channels per tap (1 for the stem, 2 for a 64 -> 64 layer),
// NT = this wave's cell tiles (8, or 7 for the odd half).
// bl = this lane's fragment base (cell (first row of the wave's tiles - 1, lane % 16 - 1), channel chunk lane / 16): the
// fragment of (tile pt, tap (ty, tx), k-step ks) is at bl + ((pt + ty) * 16 + tx) * PITCH + ks * 64, an immediate.
// a[d][ct]: the weight fragments of the next two k-steps (on exit: of the next layer's first two); their addresses are a
// wave-uniform pointer (kept in SGPRs: the empty asm stops the compiler from turning 18 step offsets into 18 loop-invariant
// 64-bit VGPR pairs) + lane * 16.  One fragment request per two MFMAs (a step's eight requests in one burst before its 16
// MFMAs: 5 % slower), into the registers the two MFMAs just read (a second register set cost spills elsewhere).
template <typename E, int KS, int NT, bool STEM>
__device__ __forceinline__ void conv_layer(const char *bl, const char *__restrict__ w, const char *__restrict__ wnext, const int cg,
                                           const unsigned lane16, f4 (&acc)[TPW][2], typename E::v8 (&a)[2][2]) {
  typedef typename E::v8 v8;
  constexpr int STEPS = STEM ? 3 : 9 * KS;  // (the stem: one step per kernel ROW, its three taps x 9 planes packed into one k-step)
  const char *wu = w + (size_t)(2 * cg) * 1024, *wnu = wnext + (size_t)(2 * cg) * 1024;
  v8 b[NT];  // ONE set of fragments: the request for tile pt of step i + 1 goes out right after the MFMAs of tile pt of step i
  auto frag = [&](int step, int pt) -> const v8 * {
    if (STEM) return (const v8 *)(bl + (step * 16 + 1) * PITCH + pt * 16 * PITCH);
    const int t = step / KS, ks = step % KS;
    return (const v8 *)(bl + ((t / 3) * 16 + (t % 3)) * PITCH + ks * 64 + pt * 16 * PITCH);
  };
#pragma unroll
  for (int pt = 0; pt < NT; ++pt) b[pt] = *frag(0, pt);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    gptr ps = (gptr)(step + 2 < STEPS ? wu + (step + 2) * 4096 : wnu + (step + 2 - STEPS) * 4096);
    asm volatile("" : "+s"(ps));
    v8 an0, an1;
#pragma unroll
    for (int pt = 0; pt < NT; ++pt) {
      if (pt == 1) {
#if AZ_H1_ABLATE & 1
        an0 = a[0][0];
        an1 = a[0][1];
#else
        an0 = *(const __attribute__((address_space(1))) v8 *)(ps + lane16);
        an1 = *(const __attribute__((address_space(1))) v8 *)(ps + lane16 + 1024);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = E::mfma(a[0][ct], b[pt], acc[pt][ct]);
      __builtin_amdgcn_sched_barrier(0);
#if !(AZ_H1_ABLATE & 2)
      if (step + 1 < STEPS) b[pt] = *frag(step + 1, pt);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    a[0][0] = a[1][0];
    a[0][1] = a[1][1];
    a[1][0] = an0;
    a[1][1] = an1;
  }
}

template <typename E, int KS, bool STEM = false>
__device__ __forceinline__ void conv_layer(const char *bl, const char *__restrict__ w, const char *__restrict__ wnext, const int cg,
                                           const unsigned lane16, const int ntiles, f4 (&acc)[TPW][2], typename E::v8 (&a)[2][2]) {
  if (ntiles == TPW) conv_layer<E, KS, TPW, STEM>(bl, w, wnext, cg, lane16, acc, a);  // (wave-uniform)
  else conv_layer<E, KS, TPW - 1, STEM>(bl, w, wnext, cg, lane16, acc, a);
}

template <typename E>
__global__ __launch_bounds__(256, 2) void k_tower_h1(const float *__restrict__ x, const uint32_t *__restrict__ ring,
                                                     const int4 *__restrict__ leafrec, const uint8_t *__restrict__ flags, int G,
                                                     const char *__restrict__ w0, const char *__restrict__ wl,
                                                     const float *__restrict__ bias_s, const float *__restrict__ sw, int L,
                                                     const float *__restrict__ headw, const float *__restrict__ headb,
                                                     void *__restrict__ feat, int FS, int feat16, unsigned int *tickets, int *overflow) {
  typedef typename E::T T;
  typedef typename E::v8 v8;
  typedef typename E::v2 v2;
  constexpr float SA = E::SA;
  constexpr int WL_BYTES = 9 * 2 * 4 * 1024;  // one 64 -> 64 layer's packed weights
  extern __shared__ __attribute__((aligned(16))) char smem_h1[];
  char *act = smem_h1;
  uint32_t *rgb = (uint32_t *)(smem_h1 + ACT_BYTES + DUMP_BYTES);  // two packed inputs: this board's and the next one's
  int *slot = (int *)(smem_h1 + ACT_BYTES + DUMP_BYTES + 2 * RG_BYTES);  // the next board (uniform hand-over from thread 0)
  char *hfr = (char *)(slot + 4);                      // the head filters as two A fragments: [k-step 2][lane 64][8]
  char *pfr = hfr + 2048;                              // the residual selectors as two A fragments: [ct 2][lane 64][8]
  float *tab = (float *)(pfr + 2048);                  // per layer: [64 biases x SW x SA][1 / SW][SW][2 unused]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave >> 1;                                // channel group: co tiles 2 cg, 2 cg + 1
  const int chalf = (wave ^ (int)(blockIdx.x >> 8)) & 1;   // which half of the rows (flipped between the two workgroups of a CU)
  const int row0 = chalf * TPW, ntiles = min(TPW, NTILE - row0);
  const int kg = lane >> 4, j = lane & 15;
  const unsigned lane16 = (unsigned)lane * 16u;

#if AZ_H1_PRIO == 1  // diagnostic: the second workgroup of a CU (by dispatch order) runs at a higher wave priority
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(3);
#elif AZ_H1_PRIO == 2  // diagnostic: ... starts half a layer late
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(100);
#endif
  for (int i = tid; i < (ACT_BYTES + DUMP_BYTES) / 16; i += 256) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // the halo stays zero for good
  // biases and scales of every layer: read from LDS inside the board loop (an L2 round trip per layer would stand exposed)
  for (int i = tid; i < (L + 1) * (C + 4); i += 256) {
    const int l = i / (C + 4), c = i % (C + 4);
    tab[i] = c < C ? bias_s[l * C + c] : c == C ? 1.0f / sw[l] : c == C + 1 ? sw[l] : 0.0f;
  }
  // the three 1x1 head convs as one more MFMA: A = the head filters in rows 0 .. 2 of a 16-row tile (two k-steps), kept in
  // LDS (in registers for the whole kernel they were spilled, and every tile of the head phase waited for scratch)
  if (wave < 2) {
    v8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (T)(j < 3 ? headw[j * C + 32 * wave + 8 * kg + i] : 0.0f);
    *(v8 *)(hfr + wave * 1024 + lane * 16) = f;
  }
  // the block's residual enters the second conv's accumulators through the MFMA as well: the eight channels a lane reads
  // back for a tile are that tile's B fragment of k-step cg, and row m of selector ct picks k = 8 (m / 4) + 4 ct + m % 4 --
  // the channel the lane's accumulator row stands for.  (As VALU work -- 16 conversions and 16 adds per tile, issued beside
  // the partner workgroup's MFMAs at a quarter of their normal rate -- this was 3,000 cycles per block.)
  if (wave >= 2) {
    const int ct = wave - 2;
    v8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (T)((kg == (j >> 2) && i == 4 * ct + (j & 3)) ? 1.0f : 0.0f);
    *(v8 *)(pfr + ct * 1024 + lane * 16) = f;
  }
  const float hb0 = headb[0], hb1 = headb[1], hb2 = headb[2];
  v2 mx = {(T)0.0f, (T)0.0f}, mx7 = mx;  // running maxima of everything stored (f16: an overflow shows as +inf); mx7: tile slot 7

  const char *bl = act + cell_idx(row0 - 1, j - 1) * PITCH + kg * 16;  // tap (0, 0) of tile 0, k-step 0
  // this lane's 8 output channels (32 cg + 8 kg .. + 7) of tile 0; the pad lane stores into a dump region instead (no exec masks)
  char *ol = j < N ? act + cell_idx(row0, j) * PITCH + (32 * cg + 8 * kg) * 2 : act + ACT_BYTES + kg * 16;
  char *ol7 = ntiles == TPW ? ol + (TPW - 1) * 16 * PITCH : act + ACT_BYTES + (TPW - 1) * 16 * PITCH + lane * 16;  // slot 7 of a 7-tile wave: the dump's last rows

  v8 a[2][2];  // weight fragments of the next two k-steps (global -> registers)
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) a[d][ct] = *(const v8 *)(w0 + (size_t)(d * 4 + 2 * cg + ct) * 1024 + lane * 16);

  // Boards come from the ticket counter, ONE BOARD AHEAD: while board g runs, thread 0 settles which board is next (skipping
  // terminal leaves), every thread fetches a word of that board's packed input into a register and parks it in the other
  // LDS buffer -- no board starts on an exposed HBM round trip.  Only the first board of a workgroup pays them.
  auto fetch_word = [&](int gb) -> uint32_t {  // thread i < RGW + 1: word i of board gb's packed input
    if (x != nullptr || gb >= G || tid > RGW) return 0u;
    return tid < RGW ? ring[(size_t)gb * RGW + tid] : (uint32_t)leafrec[gb].w;
  };
  int nxt = 0;  // (thread 0) the ticket after the board being settled
  if (tid == 0) {
    int t = (int)blockIdx.x;
    while (t < G && flags != nullptr && flags[t] == 0) t = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));  // terminal leaf
    slot[0] = t;
    nxt = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
  }
  __syncthreads();
  int g = slot[0], buf = 0;
  if (tid <= RGW) rgb[tid] = fetch_word(g);
  bool ovf = false;
#if AZ_H1_STAMP
  unsigned long long tsv[32];
  for (int k = 0; k < 32; ++k) tsv[k] = 0;
  int nboard = 0;
#endif
  while (g < G) {
#if AZ_H1_STAMP
    ++nboard;
#endif
    __syncthreads();  // the previous board's head phase is done with `act`; this board's packed input is in rgb[buf]
    const uint32_t *rg = rgb + buf * (RG_BYTES / 4);
    STAMP(0);
    uint8_t nflag = 1;
    if (tid == 0 && nxt < G && flags != nullptr) nflag = flags[nxt];  // (consumed after the stem)
    // ---- input, x SA: the stem's k-step of a cell = the 9 planes of its left neighbour, its own, its right neighbour's
    // (k = 9 tx + plane, 27 of 32 used; a neighbour off the board is zeros) -- one MFMA step per kernel row instead of one per tap
    if (tid < NV) {
      const int cell = tid, col = cell % N;
      char *p = act + cell_idx(cell / N, col) * PITCH;
      const bool on0 = col > 0, on2 = col < N - 1;
      auto put = [&](const uint32_t (&hv)[27]) {  // 27 16-bit patterns -> the cell's first 64 bytes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t d[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = 8 * q + 2 * i;
            d[i] = (k < 27 ? hv[k < 27 ? k : 0] : 0u) | (k + 1 < 27 ? hv[k + 1 < 27 ? k + 1 : 0] << 16 : 0u);
          }
          *(uint4 *)(p + 16 * q) = make_uint4(d[0], d[1], d[2], d[3]);
        }
      };
      if (x != nullptr) {  // dense planes (tests, direct callers): plain loads
        uint32_t hv[27];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const bool on = tx == 0 ? on0 : tx == 2 ? on2 : true;
          const float *xr = x + (size_t)g * 9 * NV + (on ? cell + tx - 1 : cell);
#pragma unroll
          for (int pl = 0; pl < 9; ++pl) {
            const float v = on ? xr[pl * NV] * SA : 0.0f;
            ovf |= !(fabsf(v) <= E::MAXV);
            hv[9 * tx + pl] = (uint32_t)__builtin_bit_cast(uint16_t, (T)v);
          }
        }
        put(hv);
      } else {  // the packed leaf batch: per plane, the three neighbours' bits come out of one 64-bit shift
        uint32_t hv[27];
        const uint32_t POS = (uint32_t)__builtin_bit_cast(uint16_t, (T)SA), NEG = POS | 0x8000u;
        const int lo = cell > 0 ? cell - 1 : 0, sh = lo & 31, wi0 = lo >> 5;
        const uint32_t mask3 = (on0 ? 1u : 0u) | 2u | (on2 ? 4u : 0u);  // bit tx: that neighbour is on the board
        const uint32_t pv = (int)rg[RGW] > 0 ? POS : NEG;
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
          // (word wi0 + 1 may be the first word of what follows this bitboard: its bits land on neighbours off the board)
          const unsigned long long wa = (unsigned long long)rg[pl * SWB + wi0] | ((unsigned long long)rg[pl * SWB + wi0 + 1] << 32);
          const unsigned long long wb = (unsigned long long)rg[pl * SWB + WC + wi0] | ((unsigned long long)rg[pl * SWB + WC + wi0 + 1] << 32);
          uint32_t ta = (uint32_t)(wa >> sh), tb = (uint32_t)(wb >> sh);
          if (cell == 0) { ta <<= 1; tb <<= 1; }
          ta &= mask3;
          tb &= mask3;
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) hv[9 * tx + pl] = ((ta >> tx) & 1u) * POS + ((tb >> tx) & 1u) * NEG;  // (a cell is not both)
        }
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) hv[9 * tx + 8] = ((mask3 >> tx) & 1u) * pv;
        put(hv);
      }
    }
    __syncthreads();
    STAMP(1);

    f4 acc[TPW][2];
    // accumulators start at the layer's (pre-scaled) bias: this lane's rows of channel tile ct are channels
    // 32 cg + 8 kg + 4 ct .. + 3 (the host packs the weight rows in that order), so its eight outputs are one 16-byte chunk
    auto init = [&](int layer) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f4 bv = *(const f4 *)(tab + layer * (C + 4) + 32 * cg + 8 * kg + 4 * ct);
#pragma unroll
        for (int pt = 0; pt < TPW; ++pt) acc[pt][ct] = bv;
      }
    };
    // y SA = relu(acc / SW_layer), stored in place: packed scale, packed convert, packed ReLU, one 8-byte store per four
    // channels.  with_res: the activations being overwritten are the block's residual: read them first and start the NEXT
    // layer's accumulators at (bias + residual) x SW_next x SA.  (The pad lane reads and writes its dump region.)
    auto epilogue = [&](int layer, bool with_res) {
      // (wave-uniform, read as scalars: the layers whose scales are all 1 -- every layer of a bf16 tower, and of an f16 tower
      // unless its weights are tiny -- take the branch without the multiplies)
      const float inv = E::SCALED ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tab[layer * (C + 4) + C]))) : 1.0f;
      const float swn = (E::SCALED && with_res)
                            ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tab[(layer + 1) * (C + 4) + C + 1])))
                            : 1.0f;
      // Branch-free over all eight tile slots: a wave of the odd half has no tile 7 -- its slot 7 holds whatever, reads and
      // writes the dump region (ol7) and stays out of the overflow maximum.  Every residual read stands before the first
      // store (the compiler cannot tell that they do not alias).
      v8 xr[TPW];
      if (with_res) {
#pragma unroll
        for (int pt = 0; pt < TPW; ++pt) xr[pt] = *(const v8 *)(pt < TPW - 1 ? ol + pt * 16 * PITCH : ol7);
      }
      __builtin_amdgcn_sched_barrier(0);
      auto body = [&](auto unit_c) {
        constexpr bool UNIT = decltype(unit_c)::value;
        const f2 inv2 = {inv, inv};
#pragma unroll
        for (int pt = 0; pt < TPW; ++pt) {
          v2 o[4];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            f2 y01 = {acc[pt][ct][0], acc[pt][ct][1]}, y23 = {acc[pt][ct][2], acc[pt][ct][3]};
            if (!UNIT) { y01 *= inv2; y23 *= inv2; }
            o[2 * ct] = E::cvt_relu(y01);
            o[2 * ct + 1] = E::cvt_relu(y23);
            if (pt < TPW - 1) mx = E::vmax(E::vmax(mx, o[2 * ct]), o[2 * ct + 1]);
            else mx7 = E::vmax(E::vmax(mx7, o[2 * ct]), o[2 * ct + 1]);
          }
          *(v8 *)(pt < TPW - 1 ? ol + pt * 16 * PITCH : ol7) = (v8){o[0][0], o[0][1], o[1][0], o[1][1], o[2][0], o[2][1], o[3][0], o[3][1]};
          __builtin_amdgcn_sched_barrier(0);  // tile by tile: scaled copies of all 64 accumulators at once would spill
        }
        if (with_res) {  // acc of the block's second conv = bias + residual x SW_next: one selector MFMA per tile and channel tile
          const f4 bn0 = *(const f4 *)(tab + (layer + 1) * (C + 4) + 32 * cg + 8 * kg);
          const f4 bn1 = *(const f4 *)(tab + (layer + 1) * (C + 4) + 32 * cg + 8 * kg + 4);
          v8 p0 = *(const v8 *)(pfr + lane * 16), p1 = *(const v8 *)(pfr + 1024 + lane * 16);
          if (!UNIT) {
            const T s = (T)swn;  // (a power of two <= 2^15: the host caps the f16 scales there)
            p0 *= s;
            p1 *= s;
          }
#pragma unroll
          for (int pt = 0; pt < TPW; ++pt) {
            acc[pt][0] = E::mfma(p0, xr[pt], bn0);
            acc[pt][1] = E::mfma(p1, xr[pt], bn1);
          }
          // finished HERE (an empty asm the optimiser must assume reads them): sunk into the next conv layer's first step they
          // doubled its accumulator registers and spilled
#pragma unroll
          for (int pt = 0; pt < TPW; ++pt) asm volatile("" : "+v"(acc[pt][0]), "+v"(acc[pt][1]));
        }
      };
      if (!E::SCALED || (inv == 1.0f && swn == 1.0f)) body(std::true_type{});
      else body(std::false_type{});
    };

    // ---- stem
    init(0);
    conv_layer<E, 1, true>(bl, w0, L > 0 ? wl : w0, cg, lane16, ntiles, acc, a);
    STAMP(2);
    __syncthreads();  // every wave has read the layer's input: the outputs may overwrite it
    STAMP(3);
    epilogue(0, false);
    if (tid == 0) {  // settle the next board (its flag byte arrived under the stem) and draw the ticket after it
      while (nxt < G && nflag == 0) {  // terminal leaf: rare, and only then a round trip stands exposed
        nxt = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
        nflag = nxt < G ? flags[nxt] : 1;
      }
      slot[0] = nxt;
      nxt = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    const int gn = slot[0];
    const uint32_t pre = fetch_word(gn);  // lands under the next conv layer (or the heads); parked in LDS before the loop turns
    // ---- residual blocks: conv, ReLU, conv + x, ReLU
    for (int l = 0; l < L; l += 2) {
      init(1 + l);
      conv_layer<E, 2>(bl, wl + (size_t)l * WL_BYTES, wl + (size_t)(l + 1) * WL_BYTES, cg, lane16, ntiles, acc, a);
      STAMP(6 + 4 * l);
      __syncthreads();
      STAMP(7 + 4 * l);
      epilogue(1 + l, true);  // (also starts acc of the block's second conv at bias + residual)
      STAMP(8 + 4 * l);
      __syncthreads();
      STAMP(9 + 4 * l);
      conv_layer<E, 2>(bl, wl + (size_t)(l + 1) * WL_BYTES, l + 2 < L ? wl + (size_t)(l + 2) * WL_BYTES : w0, cg, lane16, ntiles, acc, a);
      STAMP(10 + 4 * l);
      __syncthreads();
      STAMP(11 + 4 * l);
      epilogue(2 + l, false);
      STAMP(12 + 4 * l);
      __syncthreads();
      STAMP(13 + 4 * l);
    }
    // ---- the three 1x1 head convs (+ folded BN, ReLU): two MFMAs per tile, four tiles per wave (channel group cg takes
    // tiles 4 cg .. 4 cg + 3 of its rows); rows 0 .. 2 of the result sit in lanes 0 .. 15 (one cell each)
    {
      const v8 ah0 = *(const v8 *)(hfr + lane * 16), ah1 = *(const v8 *)(hfr + 1024 + lane * 16);
      const char *q = bl + (16 + 1) * PITCH + cg * 4 * 16 * PITCH;  // the centre tap of tile 4 cg
      const size_t r0 = (size_t)g * FS + (row0 + 4 * cg) * N + j;
      float *fr = (float *)feat + r0;
      uint16_t *fr16 = (uint16_t *)feat + r0;  // (feat16: bf16 rows for az_heads_forward)
      auto put = [&](int i, const f4 &h) {
        if (kg == 0 && j < N) {
          const float y0 = fmaxf(h[0] * (1.0f / SA) + hb0, 0.0f), y1 = fmaxf(h[1] * (1.0f / SA) + hb1, 0.0f);
          const float y2 = fmaxf(h[2] * (1.0f / SA) + hb2, 0.0f);
          if (feat16) {
            fr16[i * N] = __builtin_bit_cast(uint16_t, (__bf16)y0);
            fr16[KP + i * N] = __builtin_bit_cast(uint16_t, (__bf16)y1);
            fr16[KP + NV + i * N] = __builtin_bit_cast(uint16_t, (__bf16)y2);
          } else {
            fr[i * N] = y0;
            fr[KP + i * N] = y1;
            fr[KP + NV + i * N] = y2;
          }
        }
      };
      v8 bq[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        bq[i][0] = *(const v8 *)(q + i * 16 * PITCH);
        bq[i][1] = *(const v8 *)(q + i * 16 * PITCH + 64);
      }
      f4 h[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        h[i] = E::mfma(ah0, bq[i][0], (f4){0.0f, 0.0f, 0.0f, 0.0f});
        h[i] = E::mfma(ah1, bq[i][1], h[i]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) put(i, h[i]);
      if (cg == 0 || ntiles == TPW) {  // (wave-uniform: the odd half has no tile 7)
        f4 h3 = E::mfma(ah0, *(const v8 *)(q + 3 * 16 * PITCH), (f4){0.0f, 0.0f, 0.0f, 0.0f});
        h3 = E::mfma(ah1, *(const v8 *)(q + 3 * 16 * PITCH + 64), h3);
        put(3, h3);
      }
    }
    STAMP(30);
    if (tid <= RGW) rgb[(buf ^ 1) * (RG_BYTES / 4) + tid] = pre;
    g = gn;
    buf ^= 1;
    STAMP(31);
  }
#if AZ_H1_STAMP
  if (lane == 0)
    for (int k = 0; k < 32; ++k) g_h1_stamp[((size_t)blockIdx.x * 4 + wave) * 32 + k] = tsv[k];
#endif
  ovf |= j < N && (E::over(mx) || (ntiles == TPW && E::over(mx7)));  // (the pad lane's values mean nothing)
  if (ovf && overflow != nullptr) atomicOr(overflow, 1);
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
}

static std::mutex g_mutex;
static unsigned int g_attr_done[64];

}  // namespace azh1

#if AZ_H1_STAMP
extern "C" int az_h1_stamps(unsigned long long *out_host) {  // diagnostic build: [512 workgroups][4 waves][32] s_memtime stamps
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(azh1::g_h1_stamp), sizeof(unsigned long long) * 512 * 4 * 32) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

int az_tower_forward_h1(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, int32_t elem,
                        const float *x_f32_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                        const void *w0_dev, const void *wl_dev, const float *bias_scaled_dev, const float *sw_dev,
                        const float *headw_dev, const float *headb_dev, void *feat_dev, int32_t feat_bf16, int32_t feat_stride,
                        int32_t *overflow_dev, void *stream) {
  using namespace azh1;
  if ((!x_f32_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !bias_scaled_dev || !sw_dev || !headw_dev || !headb_dev ||
      !feat_dev || (n_conv_layers > 0 && !wl_dev)) { tower_set_error("null argument"); return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1) || n_conv_layers > MAX_LAYERS) {
    tower_set_error("n_boards >= 1, even n_conv_layers <= 64");
    return AZ_EINVAL;
  }
  if (board_size != 15 || channels != 64) { tower_set_error("row-tile tower: board_size 15, 64 channels"); return AZ_EINVAL; }
  if (elem != AZ_H1_F16 && elem != AZ_H1_BF16) { tower_set_error("elem: AZ_H1_F16 or AZ_H1_BF16"); return AZ_EINVAL; }
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
  auto launch = [&](auto kern, int variant) -> int {
    {
      std::lock_guard<std::mutex> lock(g_mutex);
      if (!(g_attr_done[dev] & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(MAX_LAYERS)) != hipSuccess) {
          tower_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
          return AZ_EHIP;
        }
        g_attr_done[dev] |= 1u << variant;
      }
    }
    const int grid = n_boards < 512 ? n_boards : 512;  // two workgroups per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes(n_conv_layers), s, x_f32_dev, (const uint32_t *)packed_ring_dev, (const int4 *)packed_leaf_dev,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), (int)n_boards, (const char *)w0_dev, (const char *)wl_dev,
                       bias_scaled_dev, sw_dev, (int)n_conv_layers, headw_dev, headb_dev, feat_dev, (int)feat_stride, feat_bf16 ? 1 : 0, tickets,
                       (int *)overflow_dev);
    return AZ_OK;
  };
  const int rc = elem == AZ_H1_F16 ? launch(k_tower_h1<F16>, 0) : launch(k_tower_h1<BF16>, 1);
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { tower_set_error(std::string("k_tower_h1 launch: ") + hipGetErrorString(err)); return AZ_EHIP; }
  return AZ_OK;
}

}  // extern "C"
