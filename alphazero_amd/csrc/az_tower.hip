// az_tower.hip -- fused residual conv tower (policy/value trunk + head 1x1 convs) and the heads' Linear layers, for MI355X.
//
// No counterpart in the reference beyond the model contract (nn_models/base.py:17-19): it evaluates
// the GomokuModel family of examples/gomoku/model.py:41-90 -- conv3x3(9->64)+BN+ReLU, `blocks` residual
// blocks of two conv3x3(64->64)+BN, then the 1x1 head convs+BN+ReLU -- in ONE kernel (k_tower), with the board's
// activations resident in LDS for the whole tower and every conv an implicit GEMM on bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, f32 accumulate), and the heads' Linear layers in a second small kernel (k_heads).
// BatchNorm (eval mode) is folded into the conv weights and a per-channel bias by the host (alphazero_amd/fused.py).
// The same kernel has a QuickModel instantiation (k_tower<.., QK = true>; nn_models/quick_model.py, resnet_blocks.py:28-40:
// blocks of n x (conv3x3, BN, ReLU) added to their input after the last ReLU, 3x3 head convs) with k_quick_heads for that
// family's two-layer heads.  C ABI at the bottom; declared in include/az_engine.h.
//
// k_tower (one workgroup = 4 waves = BPW boards at a time; persistent, two workgroups per CU):
//   LDS act[cell][64 ch] bf16, cells packed without halo (cell = board*N*N + row*N + col), 128 B per
//   cell, the eight 16-byte channel chunks of a cell XOR-swizzled with (cell>>1)&7: the 32 lanes of a
//   position tile hold 32 CONSECUTIVE cells, a tap shifts all of them by the same amount, so the 16
//   lanes of every ds_read_b128 group hit 16 different bank quads (conflict-free).  Taps that fall off
//   the board read a 128-byte zero row instead (per-lane 9-bit tap-validity mask).
//   GEMM orientation: D[co][pos] = sum_k W[co][k] * X[k][pos], k = (tap, ci).  A = weights (rows = out
//   channels), B = activations (cols = positions): the accumulator then holds, per lane, ONE position
//   and 4-channel runs, which pack into 8-byte LDS stores in the epilogue.
//   8 position tiles of 32 (2 per wave) x 2 channel tiles of 32: 4 accumulators (64 VGPRs) per wave.
//   Weights: one kernel row (3 taps, 24 KiB) per stage, global -> LDS by LDS-DMA, double buffered, shared by
//   the 4 waves, issued under the running stage's MFMAs; fragment reads software pipelined by hand (inline-asm
//   ds_read_b128 + counted lgkmcnt waits).  The layer is computed in place (outputs stay in accumulators until
//   every wave has finished reading the input, then overwrite it); the residual is kept in registers; the
//   accumulators start at bias (+ residual), the epilogue is ReLU + pack + store.  The stem (18 KiB of weights) is one
//   stage without inner barriers; the heads' 1x1 convs are computed from the last layer's f32 accumulators on the VALU.
//   Boards are handed out by a ticket counter (see the kernel); biases and the next board's packed input
//   arrive by LDS-DMA ahead of their use.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "../../include/az_engine.h"

namespace aztower {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;


__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// An empty asm the compiler must assume rewrites v: address arithmetic derived from the result cannot be hoisted out
// of the persistent loop and parked in registers across the conv layers (that cost ~100 VGPRs and serialised the
// fragment reads of the inner loop).
__device__ __forceinline__ int fresh(int v) { asm volatile("" : "+v"(v)); return v; }

// byte offset of 16-byte channel chunk `chunk` of cell p: rows of 2C bytes, chunks XOR-swizzled so that the 16 lanes of a
// ds_read_b128 group (16 consecutive cells, same logical chunk) cover all 64 banks
template <int C>
__device__ __forceinline__ int act_addr(int p, int chunk) {
  if (C == 64) return p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4);
  return p * 256 + ((chunk ^ (p & 15)) << 4);
}

// N = board, BPW = boards per workgroup, NPT = 32-cell tiles per wave, ST = taps staged per barrier, C = trunk width
// (64 or 128).  Four waves share the cells; with C = 128 a second set of four waves (w + 4) computes the upper 64
// output channels of the same cells (8 waves, one workgroup per CU).
template <int N, int BPW, int NPT, int ST_, int C_>
struct Geo {
  static constexpr int C = C_, CS = C / 64, NW = 4 * CS;  // waves per workgroup
  static constexpr int KS = C / 16;                        // 16-channel k-steps per tap
  static constexpr int KB = C * 16;                        // bytes of one (k-step, k-half) weight block: C rows of 8 bf16
  static constexpr int TAP_BYTES = KS * 2 * KB;            // one tap of a C->C layer: [ks][half 2][co C][8] bf16
  static constexpr int ROW = 2 * C;                        // bytes per cell
  static constexpr int NV = N * N, NVT = BPW * NV, ST = ST_;
  static constexpr int ZERO_OFF = NVT * ROW;               // one all-zero row: where taps off the board read
  static constexpr int ACT_BYTES = NVT * ROW + ROW;
  static constexpr int STAGE_BYTES = ST * TAP_BYTES;
  static constexpr int HW_BYTES = ((3 * C + 3) * 4 + 15) / 16 * 16;  // head filters + biases in LDS, f32
  static constexpr int HP_BYTES = CS > 1 ? 4 * NPT * 32 * 3 * 4 : 0; // partial head sums of the upper-channel waves
  static constexpr int WC = (NV + 31) / 32, SWB = 2 * WC, RGW = BPW * 8 * SWB;  // packed input words per group
  static constexpr int RG_BYTES = ((RGW * 4 + 8 * BPW + 4 + 15) / 16) * 16;     // ring words, BPW players, BPW slots, the group's ticket
  static_assert(C == 64 || C == 128, "trunk width 64 or 128");
  static_assert(NVT <= 4 * NPT * 32, "4 cell-waves x NPT tiles of 32 cells");
  static_assert(NPT == 2 || NPT == 4, "conv_layer counts its fragment reads for 2 or 4 cell tiles per wave");
  static_assert(RGW * 4 <= 64 * 16, "one 16-byte LDS-DMA per lane moves the packed input");
  static size_t lds_bytes() {  // any depth: the bias table is a three-layer ring
    return (size_t)ACT_BYTES + 2 * STAGE_BYTES + HW_BYTES + 2 * RG_BYTES + 3 * C * 4 + HP_BYTES;
  }
};

// async global -> LDS copy of 16 B per lane: the LDS destination is the wave-uniform `lds_base` + lane*16
__device__ __forceinline__ void glds16(const char *g_lane, char *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                   (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)lds_base, 16, 0, 0);
}

// Diagnostic build only (-DAZ_TOWER_STAMP=1, wrong outputs): wave-level s_memtime stamps of one steady-state group of
// block 7, written over the start of vfeat as u64[4 waves][32].  The product build executes no stamp.
#ifndef AZ_TOWER_STAMP
#define AZ_TOWER_STAMP 0
#endif
#if AZ_TOWER_STAMP
#define STAMP(k) do { if (stamp_on) tsv[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#ifndef AZ_TOWER_ABLATE
#define AZ_TOWER_ABLATE 0  // 0 = product; 1 .. 4 = timing diagnostics only (wrong results)
#endif
// 15x15 geometry: boards per workgroup / cell tiles per wave / taps per stage.  (1,2,3): 2x2 accumulator tiles,
// one kernel row of weights per barrier.  (2,4,1): 4x2 accumulator tiles (A fragments reused over 4 cell tiles:
// 25 % fewer LDS fragment reads), one tap per barrier so that two workgroups still fit a CU's LDS.
#ifndef AZ_TOWER15_BPW
#define AZ_TOWER15_BPW 1
#define AZ_TOWER15_NPT 2
#define AZ_TOWER15_ST 3
#endif
#ifndef AZ_TOWER9_ST
#define AZ_TOWER9_ST 1  // 9x9 (3 boards per workgroup): per-tap staging keeps two workgroups within a CU's 160 KiB of LDS
#endif
#ifndef AZ_TOWER_EARLY_BARRIER
#define AZ_TOWER_EARLY_BARRIER 1
#endif
#ifndef AZ_TOWER_WAVES_PER_SIMD
#define AZ_TOWER_WAVES_PER_SIMD 2
#endif

// async global -> LDS copy of 16 B per lane with an immediate byte offset applied to BOTH addresses
template <int OFF>
__device__ __forceinline__ void glds16o(const char *g_lane, char *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                   (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)lds_base, 16, OFF, 0);
}

// stage ST taps of weights (NCHUNK = ST*KS*2*(C/64) KiB, [tap][ks][half][co][8] bf16, contiguous in global memory and in the
// LDS buffer alike): 1 KiB per wave-instruction, each wave a run of consecutive pieces, four pieces per address
// pair through the instruction's immediate offset -- a handful of instructions per stage, issued under the MFMAs
template <int KS, int ST, int NW, int CS>
__device__ __forceinline__ void stage_w(const char *wg_src, char *dst, int wave, int lane) {
  constexpr int NCHUNK = ST * KS * 2 * CS, PW = (NCHUNK + NW - 1) / NW;
#if AZ_TOWER_ABLATE == 3  // diagnostic build: no weight streaming at all
  return;
#endif
  const int first = wave * PW;  // wave-uniform
  const char *g = wg_src + first * 1024 + lane * 16;
  char *l = dst + first * 1024;
#pragma unroll
  for (int p = 0; p < PW; p += 4) {
    const char *gp = g + p * 1024;
    char *lp = l + p * 1024;
    if (first + p + 0 < NCHUNK) glds16o<0>(gp, lp);
    if (p + 1 < PW && first + p + 1 < NCHUNK) glds16o<1024>(gp, lp);
    if (p + 2 < PW && first + p + 2 < NCHUNK) glds16o<2048>(gp, lp);
    if (p + 3 < PW && first + p + 3 < NCHUNK) glds16o<3072>(gp, lp);
  }
}

// one conv layer, in registers: acc[pt][ct] += W * X.  KS = 16-channel k-steps per tap (1 or 4).
// pp[pt] = this lane's cell, vm[pt] = bit t set when tap t of that cell is on the board.
// Weights are staged ST taps at a time, double buffered (9/ST barriers per layer).  On entry stage 0 of THIS
// layer is already in buffer `buf` (landed and barriered); during the last stage the first stage of the NEXT
// layer (wnext, KSN k-steps per tap; nullptr = none) is prefetched, so no layer starts with an exposed DMA.
// The k-steps of a stage are software pipelined through two register sets: the four fragment reads of step
// i+1 are issued before the four MFMAs of step i; the next stage's DMA is issued after the stage's first MFMAs
// (under them, not before them), and the activation fragments of the NEXT stage's first k-step are requested
// before the stage barrier (the layer's input does not change while the layer runs), so a wave comes out of the
// barrier with half of its operands already on the way.  A lane's B address for a tap is ONE base (the 128-byte
// row of the shifted cell with the swizzle of its first chunk folded in, or the 128-byte zero row when the tap is
// off the board); the k-step then only XORs (ks << 5) into it.
// Fragment reads are inline asm with hand-counted waits: hipcc's own ds_read + MFMA schedule here reads each
// fragment just before its MFMA and waits lgkmcnt(0) every time (one exposed LDS round trip per two MFMAs).
// BS = bytes between the two weight buffers; STN = taps per stage of the NEXT layer (the stem, 18 KiB of weights in all,
// runs as one stage without any intermediate barrier when a buffer holds it).
template <int KS, int KSN, int N, int ZERO_OFF, int NPT, int ST, int NW, int STN, int BS, int C>
__device__ __forceinline__ void conv_layer(const char *act, char *wbuf, int &buf, const char *wg, const char *wnext,
                                           int tid, int lane, const int (&pp)[NPT], const uint32_t (&vm)[NPT],
                                           f32x16 (&acc)[NPT][2], unsigned long long *twait = nullptr) {
  constexpr int KB = C * 16, TB = KS * 2 * KB;  // bytes per (k-step, k-half) block, per tap
  constexpr int NSTAGE = 9 / ST, STAGE_BYTES = BS, NK = ST * KS;
  static_assert(ST * TB <= BS && STN * KSN * 2 * KB <= BS, "a stage fits its buffer");
  constexpr bool XPF = (NK % 2 == 0) && NSTAGE > 1;  // cross-barrier prefetch needs the last k-step in register set 1
  // 1: the 128-channel geometry only (8 waves in lockstep, 9 stages per layer: -2.4 %; at 64 channels it measured +-0 on
  // 15x15, three spilled registers, and +0.9 % on 9x9: there the partner workgroup already fills the gap); 2: all
  constexpr bool EARLY = AZ_TOWER_EARLY_BARRIER == 2 || (AZ_TOWER_EARLY_BARRIER == 1 && C == 128);
  const int h = lane >> 5, r = lane & 31, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cot0 = C > 64 ? (wave >> 2) * 2 : 0;  // this wave's first 32-channel output tile
  const uint32_t act32 = (uint32_t)(uintptr_t)act;
  // Finished LDS addresses (base included: the activation buffer starts the 1024-byte aligned dynamic LDS, so the k-step's
  // chunk bits can be XOR-ed into them).  A lane's NPT cells are 32 apart, so they share the swizzle term of a tap: one base
  // per tap, then an add and a select per cell tile (a lane whose tile is padding has vm == 0 and reads the zero row).
  uint32_t bb[ST][NPT];
  const uint32_t zaddr = act32 + (uint32_t)(ZERO_OFF | (h << 4));
  auto bases = [&](int sg) {
#pragma unroll
    for (int j = 0; j < ST; ++j) {
      const int tap = sg * ST + j;
      const int d = (tap / 3 - 1) * N + (tap % 3 - 1);
      const uint32_t q0 = (uint32_t)(pp[0] + d);  // ("negative" for the first cells: only used with 32 pt rows added, or masked)
      const uint32_t base = act32 + (C == 64 ? (q0 << 7) | ((((q0 >> 1) & 7u) ^ (uint32_t)h) << 4) : (q0 << 8) | (((q0 & 15u) ^ (uint32_t)h) << 4));
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
        bb[j][pt] = ((vm[pt] >> tap) & 1u) ? base + (uint32_t)(pt * 32 * (C == 64 ? 128 : 256)) : zaddr;
    }
  };
  bf16x8 fa[2][2], fb[2][NPT];
  auto load_b = [&](int i, int s) {
    const int j = i / KS, ks = i % KS;
#if AZ_TOWER_ABLATE == 2 || AZ_TOWER_ABLATE == 4
    if (i < 2) for (int pt = 0; pt < NPT; ++pt) fb[s][pt] = (bf16x8)(__bf16)(AZ_TOWER_ABLATE == 2 ? (float)bb[j][pt] : 0.0f);
    return;
#endif
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[s][pt]) : "v"(bb[j][pt] ^ (uint32_t)(ks << 5)));
  };
  bases(0);
  if (XPF) load_b(0, 0);
#pragma unroll 1
  for (int sg = 0; sg < NSTAGE; ++sg) {
    const uint32_t wa32 = (uint32_t)(uintptr_t)(wbuf + ((buf + sg) & 1) * STAGE_BYTES + h * KB + (cot0 * 32 + r) * 16);
    const uint32_t wa32n = (uint32_t)(uintptr_t)(wbuf + ((buf + sg + 1) & 1) * STAGE_BYTES + h * KB + (cot0 * 32 + r) * 16);
    auto load_a_at = [&](uint32_t base, int i, int s) {
      const int j = i / KS, ks = i % KS;
#if AZ_TOWER_ABLATE == 2 || AZ_TOWER_ABLATE == 4
      if (i < 2) fa[s][0] = fa[s][1] = (bf16x8)(__bf16)(AZ_TOWER_ABLATE == 2 ? 1.0f : 0.0f);
      return;
#endif
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[s][0]) : "v"(base), "i"(j * TB + ks * 2 * KB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[s][1]) : "v"(base), "i"(j * TB + ks * 2 * KB + 512));
    };
    auto load_a = [&](int i, int s) { load_a_at(wa32, i, s); };
    if (!XPF) { if (sg > 0) bases(sg); load_b(0, 0); }
    if (!(XPF && EARLY) || sg == 0) load_a(0, 0);  // (EARLY: a later stage's first fragments were requested before the previous stage's last MFMAs)
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      if (i + 1 < NK) {
        load_a(i + 1, (i + 1) & 1);
        load_b(i + 1, (i + 1) & 1);
        if (NPT == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // step i's reads have returned
        else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      } else if (XPF && EARLY) {
        // The stage barrier comes BEFORE the last k-step's MFMAs: this wave's reads of the stage have all returned and
        // its share of the next stage's weights has landed, so after the barrier the next stage's first fragments (A and
        // B) are requested and travel under those MFMAs -- no stage starts with an exposed LDS round trip.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (sg + 1 < NSTAGE) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          __builtin_amdgcn_sched_barrier(0);
          bases(sg + 1);
          load_b(0, 0);
          load_a_at(wa32n, 0, 0);
        } else {
          __syncthreads();  // the layer's last stage: every wave has finished reading act -> the epilogues may overwrite it
        }
      } else if (XPF && sg + 1 < NSTAGE) {
        bases(sg + 1);  // the next stage's first activation fragments: in flight across the stage barrier
        load_b(0, 0);
        if (NPT == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
#if AZ_TOWER_ABLATE == 1  // diagnostic build: LDS fragment reads kept alive, no MFMAs
        asm volatile("" ::"v"(fa[i & 1][0]), "v"(fa[i & 1][1]), "v"(fb[i & 1][pt]));
#else
        acc[pt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][0], fb[i & 1][pt], acc[pt][0], 0, 0, 0);
        acc[pt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][1], fb[i & 1][pt], acc[pt][1], 0, 0, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 0) {  // the DMA of the next stage's weights: issued under this stage's MFMAs
        char *wn = wbuf + ((buf + sg + 1) & 1) * STAGE_BYTES;
        if (sg < NSTAGE - 1) stage_w<KS, ST, NW, C / 64>(wg + (sg + 1) * ST * TB, wn, wave, lane);
        else if (wnext != nullptr) stage_w<KSN, STN, NW, C / 64>(wnext, wn, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if AZ_TOWER_STAMP
    const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
    if (XPF && EARLY) continue;  // (its barrier stands before the stage's last MFMAs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next stage has landed
    if (XPF && sg + 1 < NSTAGE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // compiler-level ordering only matters for the DMA'd bytes:
      __builtin_amdgcn_s_barrier();                            // a raw barrier, so the prefetched reads stay in flight
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      __syncthreads();  // after the last stage: every wave has finished reading act -> the epilogue may overwrite it
    }
#if AZ_TOWER_STAMP
    if (twait) *twait += __builtin_amdgcn_s_memtime() - tw0;
#endif
  }
  buf = (buf + NSTAGE) & 1;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

// two f32 -> packed bf16 (one v_cvt_pk_bf16_f32), then ReLU on the packed pair as a signed 16-bit max with 0
// (a negative bf16 has its sign bit set, i.e. is a negative int16): one v_pk_max_i16 for two values
__device__ __forceinline__ uint32_t pack_relu(float a, float b) {
  const bf16x2 t = __builtin_convertvector((f32x2){a, b}, bf16x2);
  const i16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, t), z));
}

// accumulators start at the layer's bias (+ the block's residual): the epilogue is then only ReLU + pack + store.
// The 8 bias reads go out back to back (one LDS round trip per layer, not one per 4 values).
template <bool RES, int NPT>
__device__ __forceinline__ void init_acc(const float *bias, int lane, f32x16 (&acc)[NPT][2], const uint2 (&res)[NPT][2][4]) {
  const int h = fresh(lane) >> 5;
  float4 bv[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[ct][q] = *(const float4 *)(bias + ct * 32 + q * 8 + h * 4);
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = bv[ct][q].x, v1 = bv[ct][q].y, v2 = bv[ct][q].z, v3 = bv[ct][q].w;
        if (RES) {
          const uint2 rr = res[pt][ct][q];
          v0 += bf_lo(rr.x); v1 += bf_hi(rr.x); v2 += bf_lo(rr.y); v3 += bf_hi(rr.y);
        }
        acc[pt][ct][4 * q + 0] = v0; acc[pt][ct][4 * q + 1] = v1; acc[pt][ct][4 * q + 2] = v2; acc[pt][ct][4 * q + 3] = v3;
      }
}

// ReLU, bf16, written back in place.  (Keeping the packed values in registers as the next block's residual instead of
// reading them back was tried: 32 more live VGPRs across the epilogue, spills, slower.)
template <int NPT, int C>
__device__ __forceinline__ void epilogue(char *act, int lane, int cot0, const int (&pp)[NPT], const bool (&ok)[NPT],
                                         const f32x16 (&acc)[NPT][2]) {
  const int h = fresh(lane) >> 5;
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int pe = fresh(pp[pt]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 v = make_uint2(pack_relu(acc[pt][ct][4 * q + 0], acc[pt][ct][4 * q + 1]), pack_relu(acc[pt][ct][4 * q + 2], acc[pt][ct][4 * q + 3]));
        if (ok[pt]) *(uint2 *)(act + act_addr<C>(pe, (cot0 + ct) * 4 + q) + 8 * h) = v;
      }
  }
}

// heads' 1x1 convs from the accumulators of the last trunk layer (bias and residual already in them, ReLU applied here in f32).
// hw: [3][C] filters + [3] biases, f32, in LDS.  acc[pt][ct][4q + j] = channel (cot0+ct)*32 + q*8 + (lane>>5)*4 + j of cell
// pp[pt].  C = 128: waves 4..7 hold the upper 64 channels of the cells of waves 0..3 and hand their partial sums over
// through `hp` (one barrier); waves 0..3 write the rows.
template <int NPT, int NV, int C>
__device__ __forceinline__ void heads_from_acc(const float *hw, float *hp, int lane_, int wave, const int (&pp)[NPT],
                                               const bool (&ok)[NPT], const int *slots, const f32x16 (&acc)[NPT][2],
                                               uint16_t *feat, int FS) {
  const int lane = fresh(lane_), h = lane >> 5, r = lane & 31;
  const int cot0 = C > 64 ? (wave >> 2) * 2 : 0, cw = wave & 3;
  f32x2 hs[NPT][3];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) hs[pt][0] = hs[pt][1] = hs[pt][2] = (f32x2){0.0f, 0.0f};
  // every LDS operand of the head is requested up front (one round trip: under the partner workgroup's fragment traffic a
  // dependent LDS access costs several hundred cycles); packed f32 FMAs; ReLU as a signed integer max (a negative
  // float is a negative int)
  float4 wv[2][4][3];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 3; ++k) wv[ct][q][k] = *(const float4 *)(hw + k * C + (cot0 + ct) * 32 + q * 8 + h * 4);
  const float hb0 = hw[3 * C + 0], hb1 = hw[3 * C + 1], hb2 = hw[3 * C + 2];
  int gs[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) gs[pt] = slots[pp[pt] / NV];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        f32x2 va, vb;
        va[0] = __int_as_float(max(__float_as_int(acc[pt][ct][4 * q + 0]), 0));
        va[1] = __int_as_float(max(__float_as_int(acc[pt][ct][4 * q + 1]), 0));
        vb[0] = __int_as_float(max(__float_as_int(acc[pt][ct][4 * q + 2]), 0));
        vb[1] = __int_as_float(max(__float_as_int(acc[pt][ct][4 * q + 3]), 0));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          hs[pt][k] = __builtin_elementwise_fma(va, (f32x2){wv[ct][q][k].x, wv[ct][q][k].y}, hs[pt][k]);
          hs[pt][k] = __builtin_elementwise_fma(vb, (f32x2){wv[ct][q][k].z, wv[ct][q][k].w}, hs[pt][k]);
        }
      }
  float t[NPT][3];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // own half + the other half-wave's: v_permlane32_swap (no LDS round trip)
      const float own = hs[pt][k][0] + hs[pt][k][1];
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(own), __float_as_uint(own), false, false);
      t[pt][k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
  if (C > 64) {
    if (wave >= 4 && h == 0) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int k = 0; k < 3; ++k) hp[((cw * NPT + pt) * 32 + r) * 3 + k] = t[pt][k];
    }
    __syncthreads();
    if (wave >= 4) return;
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int k = 0; k < 3; ++k) t[pt][k] += hp[((cw * NPT + pt) * 32 + r) * 3 + k];
  }
  constexpr int KP = (NV + 15) / 16 * 16;
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    if ((pt & 1) != h || !ok[pt]) continue;  // even tiles are written by the lower half-wave, odd ones by the upper
    const int cell = fresh(pp[pt]) % NV, g = gs[pt];
    if (g < 0) continue;
    // the heads' Linear inputs, bf16: [policy NV | pad to 16 | value0 NV | value1 NV | pad] (az_heads_forward reads it)
    uint16_t *fr = feat + (size_t)g * FS;
    fr[cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(t[pt][0] + hb0, 0.0f));
    fr[KP + cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(t[pt][1] + hb1, 0.0f));
    fr[KP + NV + cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(t[pt][2] + hb2, 0.0f));
  }
}

// QuickModel family (nn_models/resnet_blocks.py:28-40): the last layer of a block writes relu(conv) + x, x = the block's input
// (kept in registers as packed bf16 since the block's first layer, like the GomokuModel block's residual), and is not
// activated again.  The sum is taken in f32 and rounded to bf16 once.
template <int NPT, int C>
__device__ __forceinline__ void epilogue_add(char *act, int lane, int cot0, const int (&pp)[NPT], const bool (&ok)[NPT],
                                             const f32x16 (&acc)[NPT][2], const uint2 (&res)[NPT][2][4]) {
  const int h = fresh(lane) >> 5;
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int pe = fresh(pp[pt]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 rr = res[pt][ct][q];
        const f32x2 a = {fmaxf(acc[pt][ct][4 * q + 0], 0.0f) + bf_lo(rr.x), fmaxf(acc[pt][ct][4 * q + 1], 0.0f) + bf_hi(rr.x)};
        const f32x2 b = {fmaxf(acc[pt][ct][4 * q + 2], 0.0f) + bf_lo(rr.y), fmaxf(acc[pt][ct][4 * q + 3], 0.0f) + bf_hi(rr.y)};
        const uint2 v = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)),
                                   __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2)));
        if (ok[pt]) *(uint2 *)(act + act_addr<C>(pe, (cot0 + ct) * 4 + q) + 8 * h) = v;
      }
  }
}

// QuickModel heads: the three 3x3 head convs (2 policy channels, 1 value channel; BN folded, ReLU) from the finished bf16
// activations in LDS, on the VALU in f32: thread m < NVT takes cell m and 64 of its channels (C = 128: waves 4..7 take the upper
// 64 and hand their partial sums over through `hp`); the filters hq [3][9 taps][C] f32 come through the scalar cache (the
// channel split is wave-uniform).  Ends with a workgroup barrier: every read of `act` is done before the next board's input
// is expanded over it.  Feature rows, bf16: [policy0 NV | policy1 NV | pad to 16 | value NV | pad to 16].
template <int N, int NVT, int NW, int ZERO_OFF, int C>
__device__ __forceinline__ void heads_quick(const char *act, const float *__restrict__ hq, const float *hw, float *hp, int tid_, int wave,
                                            const int *slots, uint16_t *feat, int FS) {
  constexpr int NV = N * N, KPQ = (2 * NV + 15) / 16 * 16;
  static_assert(NVT <= 256 && (NW == 4 || NW == 8), "one thread per cell and 64-channel split");
  const int m = fresh(tid_) & 255, split = wave >> 2;  // (C = 64: split is 0)
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  const bool live = m < NVT;
  const int mc = live ? m : 0, cell = mc % NV, row = cell / N, col = cell - row * N;
  const float *wq = hq + split * 64;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int dr = tap / 3 - 1, dc = tap - (tap / 3) * 3 - 1, rr = row + dr, cc = col + dc;
    const bool on = live && rr >= 0 && rr < N && cc >= 0 && cc < N;
    const int q = mc + dr * N + dc;
    const float *wt = wq + tap * C;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {  // 16-byte chunk = 8 channels
      const uint4 a = *(const uint4 *)(act + (on ? act_addr<C>(q, split * 8 + ch) : ZERO_OFF + (ch << 4)));
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf_lo(aw[e]), hi = bf_hi(aw[e]);
        const int c0 = ch * 8 + 2 * e;
        s0 = fmaf(lo, wt[c0], s0); s0 = fmaf(hi, wt[c0 + 1], s0);
        s1 = fmaf(lo, wt[9 * C + c0], s1); s1 = fmaf(hi, wt[9 * C + c0 + 1], s1);
        s2 = fmaf(lo, wt[18 * C + c0], s2); s2 = fmaf(hi, wt[18 * C + c0 + 1], s2);
      }
    }
  }
  if (NW == 8 && wave >= 4 && live) { hp[m] = s0; hp[256 + m] = s1; hp[512 + m] = s2; }
  __syncthreads();
  if (wave >= 4 || !live) return;
  if (NW == 8) { s0 += hp[m]; s1 += hp[256 + m]; s2 += hp[512 + m]; }
  const int g = slots[m / NV];
  if (g < 0) return;
  uint16_t *fr = feat + (size_t)g * FS;
  fr[cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(s0 + hw[3 * C + 0], 0.0f));
  fr[NV + cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(s1 + hw[3 * C + 1], 0.0f));
  fr[KPQ + cell] = __builtin_bit_cast(uint16_t, (__bf16)fmaxf(s2 + hw[3 * C + 2], 0.0f));
}

// async global -> LDS copy of 4 B per lane
__device__ __forceinline__ void glds4(const void *g_lane, void *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                   (__attribute__((address_space(3))) void *)(uint32_t)(uintptr_t)lds_base, 4, 0, 0);
}

// x: (G, 9, N, N) bf16, or nullptr to read the engine's packed leaf batch (ring u32 (G,8,SWB) + leaf records).
// w0: [9][1][2][C][8] bf16 (first conv, 9 input channels padded to 16).  wl: [L][9][C/16][2][C][8] bf16.
// bias: [1+L][C] f32.  headw: [3][C] f32 (policy, value0, value1), headb: [3].
// feat: (G, FS) bf16 = the inputs of the heads' Linear layers (layout: see the heads' epilogue and k_heads).
// QK = true: the reference's QuickModel family (nn_models/quick_model.py:67-93): L = n_blocks x BL layers of conv3x3 + BN + ReLU,
// a block's input added back after its last ReLU (epilogue_add), 3x3 head convs (heads_quick); headw = [3][9][C] f32.
template <int N, int BPW, int NPT, int ST, int C, bool QK = false>
__global__ __launch_bounds__(C * 4, (C == 64 ? AZ_TOWER_WAVES_PER_SIMD : 1)) void k_tower(const uint16_t *__restrict__ x, const uint32_t *__restrict__ ring,
                                               const int4 *__restrict__ leafrec, const int32_t *__restrict__ list,
                                               const int32_t *__restrict__ count, const uint8_t *__restrict__ flags, int G,
                                               const char *__restrict__ w0,
                                               const char *__restrict__ wl, const float *__restrict__ bias, int L,
                                               const float *__restrict__ headw, const float *__restrict__ headb,
                                               uint16_t *__restrict__ feat, int FS, unsigned int *tickets, int BL) {
  using Q = Geo<N, BPW, NPT, ST, C>;
  constexpr int NW = Q::NW, KS = Q::KS, TAP_BYTES = Q::TAP_BYTES;
  constexpr int WC = Q::WC, SWB = Q::SWB, RGW = Q::RGW, RG_STRIDE = Q::RG_BYTES, STAGE_BYTES = Q::STAGE_BYTES, NTH = NW * 64;
  // the stem's taps per stage: all nine (18 KiB at C = 64, no barrier inside the layer) when a weight buffer holds them
#ifdef AZ_TOWER_STEM_ST
  constexpr int ST0 = AZ_TOWER_STEM_ST;
#else
  constexpr int ST0 = 9 * 2 * Q::KB <= STAGE_BYTES ? 9 : (3 * 2 * Q::KB <= STAGE_BYTES ? 3 : 1);
#endif
  extern __shared__ __attribute__((aligned(1024))) char smem_bf16[];  // (conv_layer XORs chunk bits into finished addresses)
  char *const smem = smem_bf16;
  char *act = smem;
  char *wbuf = smem + Q::ACT_BYTES;
  float *hw = (float *)(wbuf + 2 * STAGE_BYTES);  // 3*C + 3 floats
  char *rgb = wbuf + 2 * STAGE_BYTES + Q::HW_BYTES;  // 2 x {RGW ring words, BPW player words}
  float *bl = (float *)(rgb + 2 * RG_STRIDE);     // 3 x C biases: the running layer's, the next one's (DMA'd under this layer's conv), the last one's
  float *hp = bl + 3 * C;                         // (C = 128) partial head sums of waves 4..7
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3, cot0 = C > 64 ? (wave >> 2) * 2 : 0;  // cell-wave index; first 32-channel output tile

  for (int i = tid; i < Q::ACT_BYTES / 16; i += NTH) ((uint4 *)act)[i] = make_uint4(0, 0, 0, 0);  // incl. the zero slot
  for (int i = tid; i < 3 * C + 3; i += NTH) hw[i] = i < 3 * C ? headw[i] : headb[i - 3 * C];
  for (int i = tid; i < C; i += NTH) bl[i] = bias[i];  // the stem's; every later layer's row arrives by LDS-DMA, one layer ahead

  // this lane's NPT cells (B-operand column / accumulator column): tile = NPT*cw + pt
  int pp[NPT];
  bool ok[NPT];
  uint32_t vm[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int m = (NPT * cw + pt) * 32 + (lane & 31);
    ok[pt] = m < Q::NVT;
    pp[pt] = ok[pt] ? m : 0;
    const int cell = pp[pt] % Q::NV, row = cell / N, col = cell % N;
    uint32_t mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int rr = row + t / 3 - 1, cc = col + t % 3 - 1;
      if (rr >= 0 && rr < N && cc >= 0 && cc < N) mask |= 1u << t;
    }
    vm[pt] = ok[pt] ? mask : 0u;
  }

  // boards to evaluate: the engine's compact list (packed mode) or all G rows
  const int nb_total = (list != nullptr && count != nullptr) ? min(*count, G) : G;
  const int ngroups = (nb_total + BPW - 1) / BPW;
  constexpr int BOARD_BYTES = 8 * SWB * 4;
  // Groups of BPW boards are handed out by a ticket counter, not by block index: the two workgroups that share a CU
  // do not run at the same speed (the MFMA pipe favours the older one's waves: measured 818 vs 1,008 us for 32 boards
  // each), so a fixed split leaves every CU half empty for the last fifth of the launch.  tickets[0] = next group - gridDim.x (the first gridDim.x groups go by block index),
  // tickets[1] = workgroups that have drawn a ticket past the end; the last of them zeroes both for the next launch.
  // Wave 0 runs the protocol one group ahead, in three steps spread over the current group's tower so that none of
  // them is waited for: draw (atomic), look the slots up in the list, start the DMA of the packed input.
  int tk_draw = 0, tk_next = 0, sl_next = -1;  // wave 0: the next group's ticket; lane b < BPW: its b-th slot
  auto draw = [&]() {  // step 1: lane 0's returning atomic; nothing reads the result before step 2
    if (wave == 0 && lane == 0) tk_draw = (int)(gridDim.x + atomicAdd(&tickets[0], 1u));
  };
  auto lookup = [&]() {  // step 2 (uses the ticket)
    if (wave != 0) return;
    tk_next = __builtin_amdgcn_readfirstlane(tk_draw);
    for (;;) {
      const int ci = tk_next * BPW + lane;
      sl_next = (tk_next < ngroups && lane < BPW && ci < nb_total) ? (list != nullptr ? list[ci] : ci) : -1;
      if (flags == nullptr || tk_next >= ngroups) break;
      // flag mode: a group is BPW consecutive slots, of which only the flagged ones are evaluated; a group without any
      // (7 % of the single-board groups: terminal leaves) costs one more draw, waited for on the spot
      if (sl_next >= 0 && flags[sl_next] == 0) sl_next = -1;
      if (__ballot(sl_next >= 0) != 0ull) break;
      tk_next = __builtin_amdgcn_readfirstlane(lane == 0 ? (int)(gridDim.x + atomicAdd(&tickets[0], 1u)) : 0);
    }
  };
  auto fetch = [&](int which) {  // step 3 (uses the slots): slots + ticket by ds_write, ring words + players by LDS-DMA
    if (wave != 0) return;
    char *dst = rgb + which * RG_STRIDE;
    int dma_slot = -1;
    const int db = (lane * 16) / BOARD_BYTES;  // the board this lane's 16-byte DMA piece belongs to
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      const int sb = __shfl(sl_next, b, 64);
      if (db == b) dma_slot = sb;
    }
    if (lane < BPW) ((int *)dst)[RGW + BPW + lane] = sl_next;
    if (lane == 0) ((int *)dst)[RGW + 2 * BPW] = tk_next;
    if (x != nullptr) return;
    if (lane * 16 < BPW * BOARD_BYTES && dma_slot >= 0)
      glds16((const char *)ring + (size_t)dma_slot * BOARD_BYTES + (lane * 16) % BOARD_BYTES, dst);
    if (lane < BPW && sl_next >= 0) glds4(&leafrec[sl_next].w, dst + RGW * 4);
  };
  int buf = 0, inb = 0, bslot = 0;
  [[maybe_unused]] int it = 0;  // boards done (read by the stamp build only)
  // bias row of `layer` -> the ring slot after the running one (wave 0; lands before the running conv's first stage barrier)
  auto bias_ahead = [&](int layer) {
    if (wave < C / 64) glds4(bias + layer * C + wave * 64 + lane, (char *)(bl + (bslot == 2 ? 0 : bslot + 1) * C + wave * 64));
  };
  auto bias_step = [&]() { bslot = bslot == 2 ? 0 : bslot + 1; };
  // prologue: the stem's first weight stage and the first group's input
  stage_w<1, ST0, NW, C / 64>(w0, wbuf, wave, lane);
  tk_draw = (int)blockIdx.x;  // the first group needs no atomic: tickets 0 .. gridDim.x-1 are the block indices, the counter hands out the rest
  lookup();
  fetch(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#if AZ_TOWER_STAMP
  unsigned long long tsv[32];
  for (int k = 0; k < 32; ++k) tsv[k] = 0;
  tsv[20] = __builtin_amdgcn_s_memtime();
  tsv[21] = __builtin_amdgcn_s_memrealtime();
#endif
  for (;; ++it) {
    const uint32_t *rg = (const uint32_t *)(rgb + inb * RG_STRIDE);
    const int *slots = (const int *)rg + RGW + BPW;
    if (slots[BPW] >= ngroups) break;  // this workgroup's ticket is past the end (the same word for every thread)
#if AZ_TOWER_STAMP
    const bool stamp_on = blockIdx.x == 7 && it == 5;
    unsigned long long tw = 0;
#endif
    STAMP(0);
#if AZ_TOWER_STAMP
    if (stamp_on) tsv[30] = __builtin_amdgcn_s_memrealtime();
#endif
    draw();  // the next group's ticket: on its way while this group's input is expanded and the stem runs
    // ---- input -> act[cell][0..15] (channels 9..15 zero): the NCHW bf16 planes, or the packed leaf batch
    // (8 bitboard states per board + the player to move), expanded here
    for (int m = fresh(tid); m < Q::NVT; m += NTH) {
      const int b = m / Q::NV, cell = m % Q::NV, g = slots[b];
      uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (g >= 0) {
        if (x != nullptr) {
          const uint16_t *xr = x + (size_t)g * 9 * Q::NV + cell;
#pragma unroll
          for (int ch = 0; ch < 9; ++ch) c[ch >> 1] |= (uint32_t)xr[ch * Q::NV] << ((ch & 1) * 16);
        } else {
          const uint32_t *rb = rg + b * 8 * SWB;
          const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
          for (int k = 0; k < 8; ++k) {  // +1 -> 0x3F80, -1 -> 0xBF80 (bf16)
            const uint32_t mine = (rb[k * SWB + wi] >> sh) & 1u, theirs = (rb[k * SWB + WC + wi] >> sh) & 1u;
            const uint32_t v = mine ? 0x3F80u : (theirs ? 0xBF80u : 0u);
            c[k >> 1] |= v << ((k & 1) * 16);
          }
          c[4] = (int)rg[RGW + b] > 0 ? 0x3F80u : 0xBF80u;
        }
      }
      *(uint4 *)(act + act_addr<C>(m, 0)) = make_uint4(c[0], c[1], c[2], c[3]);
      *(uint4 *)(act + act_addr<C>(m, 1)) = make_uint4(c[4], c[5], c[6], c[7]);
    }
    inb ^= 1;
    __syncthreads();
    STAMP(1);

    f32x16 acc[NPT][2];
    uint2 res[NPT][2][4];
    // ---- stem
    init_acc<false, NPT>(bl + bslot * C + cot0 * 32, lane, acc, res);
    if (L > 0) bias_ahead(1);
    else bias_ahead(0);  // (the next group's stem; harmless when there is none)
    if (L > 0) conv_layer<1, KS, N, Q::ZERO_OFF, NPT, ST0, NW, ST, STAGE_BYTES, C>(act, wbuf, buf, w0, wl, tid, lane, pp, vm, acc);
    else conv_layer<1, 1, N, Q::ZERO_OFF, NPT, ST0, NW, ST0, STAGE_BYTES, C>(act, wbuf, buf, w0, w0, tid, lane, pp, vm, acc);
    STAMP(2);
    bias_step();
    if (L > 0) {
      epilogue<NPT, C>(act, lane, cot0, pp, ok, acc);
      __syncthreads();
    }
    STAMP(3);
    lookup();  // the ticket drawn at the top of the loop has long arrived
    if (L == 0) {  // no conv layer (and so no stage barrier) follows: the next group's ticket, slots and input must have
      fetch(inb);  // landed before any wave reads them at the loop top
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if constexpr (QK) {
      // ---- QuickModel blocks (L >= 1): every layer conv + bias + ReLU in place; the block's input stays in `res` until its last layer
      int j = 0;
      for (int l = 0; l < L; ++l) {
        if (j == 0) {
          const int h = fresh(lane) >> 5;
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
            const int pe = fresh(pp[pt]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int q = 0; q < 4; ++q) res[pt][ct][q] = *(const uint2 *)(act + act_addr<C>(pe, (cot0 + ct) * 4 + q) + 8 * h);
          }
        }
        init_acc<false, NPT>(bl + bslot * C + cot0 * 32, lane, acc, res);
        bias_ahead(l + 1 < L ? 2 + l : 0);
        if (l + 1 < L)
          conv_layer<KS, KS, N, Q::ZERO_OFF, NPT, ST, NW, ST, STAGE_BYTES, C>(act, wbuf, buf, wl + (size_t)l * 9 * TAP_BYTES, wl + (size_t)(l + 1) * 9 * TAP_BYTES,
                                           tid, lane, pp, vm, acc);
        else  // the board's last layer: prefetch the next board's stem weights
          conv_layer<KS, 1, N, Q::ZERO_OFF, NPT, ST, NW, ST0, STAGE_BYTES, C>(act, wbuf, buf, wl + (size_t)l * 9 * TAP_BYTES, w0, tid, lane, pp, vm, acc);
        bias_step();
        if (j == BL - 1) epilogue_add<NPT, C>(act, lane, cot0, pp, ok, acc, res);
        else epilogue<NPT, C>(act, lane, cot0, pp, ok, acc);
        __syncthreads();
        if (l == 0) fetch(inb);  // the next group's packed input: lands under the rest of this group's tower
        j = j + 1 == BL ? 0 : j + 1;
      }
      heads_quick<N, Q::NVT, NW, Q::ZERO_OFF, C>(act, headw, hw, hp, tid, wave, slots, feat, FS);
    } else {
    // ---- residual blocks
    for (int l = 0; l < L; l += 2) {
      {  // res = this block's input
        const int h = fresh(lane) >> 5;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
          const int pe = fresh(pp[pt]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) res[pt][ct][q] = *(const uint2 *)(act + act_addr<C>(pe, (cot0 + ct) * 4 + q) + 8 * h);
        }
      }
#if AZ_TOWER_STAMP
      const int sk = 4 + (l & 2) * 4;
      unsigned long long *twp = &tw;
#else
      unsigned long long *twp = nullptr;
#endif
      init_acc<false, NPT>(bl + bslot * C + cot0 * 32, lane, acc, res);
      bias_ahead(2 + l);
      STAMP(sk + 0);
      conv_layer<KS, KS, N, Q::ZERO_OFF, NPT, ST, NW, ST, STAGE_BYTES, C>(act, wbuf, buf, wl + (size_t)l * 9 * TAP_BYTES, wl + (size_t)(l + 1) * 9 * TAP_BYTES, tid,
                                       lane, pp, vm, acc, twp);
      STAMP(sk + 1);
      bias_step();
      epilogue<NPT, C>(act, lane, cot0, pp, ok, acc);
      __syncthreads();
      STAMP(sk + 2);
      if (l == 0) fetch(inb);  // the next group's packed input: lands under the rest of this group's tower
      init_acc<true, NPT>(bl + bslot * C + cot0 * 32, lane, acc, res);
      if (l + 2 < L) bias_ahead(3 + l);
      else bias_ahead(0);
      STAMP(sk + 3);
      if (l + 2 < L)
        conv_layer<KS, KS, N, Q::ZERO_OFF, NPT, ST, NW, ST, STAGE_BYTES, C>(act, wbuf, buf, wl + (size_t)(l + 1) * 9 * TAP_BYTES, wl + (size_t)(l + 2) * 9 * TAP_BYTES,
                                         tid, lane, pp, vm, acc, twp);
      else  // last layer of this board: prefetch the next board's stem weights
        conv_layer<KS, 1, N, Q::ZERO_OFF, NPT, ST, NW, ST0, STAGE_BYTES, C>(act, wbuf, buf, wl + (size_t)(l + 1) * 9 * TAP_BYTES, w0, tid, lane,
                                         pp, vm, acc, twp);
      STAMP(sk + 4);
      bias_step();
      if (l + 2 < L) {  // (the last layer's output stays in the accumulators for the heads)
        epilogue<NPT, C>(act, lane, cot0, pp, ok, acc);
        __syncthreads();
      }
      STAMP(sk + 5);
#if AZ_TOWER_STAMP
      if (stamp_on) tsv[sk + 6] = tw;
#endif
    }
    // ---- heads: the three 1x1 convs (+folded BN) + ReLU, straight from the last layer's f32 accumulators (a lane holds
    // 32 of its cell's 64 channels; the other half sits in lane ^ 32): 3 x 32 FMAs per cell on the VALU against the f32
    // filter table in LDS (broadcast reads), one cross-half exchange, then the bf16 rows of the heads' Linear inputs
    heads_from_acc<NPT, Q::NV, C>(hw, hp, lane, wave, pp, ok, slots, acc, feat, FS);
    }  // !QK
    STAMP(28);
    STAMP(29);  // (no barrier: the last conv ended with one, and nothing has read act or the weight buffers since)
#if AZ_TOWER_STAMP
    if (stamp_on) tsv[31] = __builtin_amdgcn_s_memrealtime();
#endif
  }
  // past the end: report in; the last workgroup to do so re-arms the counters for the next launch
  if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) {
    tickets[0] = 0u;
    tickets[1] = 0u;
    __threadfence();
  }
#if AZ_TOWER_STAMP
  if (tid == 0) {  // every workgroup: start / end on the global 100 MHz counter, behind the per-wave rows
    unsigned long long *o = (unsigned long long *)feat + 4 * 32 + 4 * blockIdx.x;
    o[0] = tsv[21]; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = it; o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4) << 8);  // XCC_ID | HW_ID << 8
  }
  if (blockIdx.x == 7 && lane == 0) {  // whole-kernel stamps of one workgroup
    tsv[22] = __builtin_amdgcn_s_memtime();
    tsv[23] = __builtin_amdgcn_s_memrealtime();
    tsv[24] = it;
    for (int k = 0; k < 32; ++k) ((unsigned long long *)feat)[wave * 32 + k] = tsv[k];
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// The heads' Linear layers (examples/gomoku/model.py:69-90: policy Linear(NV -> NV); value Linear(2 NV -> 64), ReLU,
// Linear(64 -> 1), tanh) for a whole batch, as two small bf16 MFMA GEMMs read straight from L2 (no LDS staging: the
// feature rows and the fragment-packed weights are read once per wave, the kernel is one round of 4-wave workgroups).
// One workgroup = 32 boards.  D[board][n] = sum_k feat[board][k] * W[n][k]: A = features (lane: board l&31, k-half l>>5),
// B = weights (lane: output n = l&31), so a register of the result is 32 consecutive outputs of one board (coalesced).
//   policy: wave w of workgroup (x, y) computes output tile 4y + w over K = KP;
//   value:  wave w computes hidden tile w&1 over K-half w>>1 of KV; halves are summed through LDS, then
//           v = tanh(sum_n relu(hid[n] + b1[n]) * w2[n] + b2) by a lane butterfly.
// feat (G, FS) bf16: [0, KP) policy features (NV real), [KP, KP + KV) value features (2 NV real); pads are zero.
// wp: [NTP][KSP][2][32][8] bf16, w1: [2][KSV][2][32][8] bf16 ([n-tile][k-step][k-half][n][8 k]), zero padded.
template <int NV>
__global__ __launch_bounds__(256, 2) void k_heads(const uint16_t *__restrict__ feat, int FS, int G, const char *__restrict__ wp,
                                                  const float *__restrict__ bp, const char *__restrict__ w1,
                                                  const float *__restrict__ b1, const float *__restrict__ w2, float b2,
                                                  float *__restrict__ logits, float *__restrict__ values) {
  constexpr int KP = (NV + 15) / 16 * 16, KSP = KP / 16, NTP = (NV + 31) / 32;
  constexpr int KV = (2 * NV + 15) / 16 * 16, KSV = KV / 16, KH = (KSV + 1) / 2;
  constexpr int MT = 1;  // 32-board tiles per workgroup (2 measured slower, 31.6 vs 23.0 us: half as many workgroups to hide the L2 latency)
  __shared__ float part[4][MT * 32][33];
  __shared__ float vsum[2][MT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g0 = blockIdx.x * (MT * 32);
  const uint16_t *fa[MT];  // rows past the batch are computed on a copy of the last row and not stored
#pragma unroll
  for (int m = 0; m < MT; ++m) fa[m] = feat + (size_t)min(g0 + m * 32 + r, G - 1) * FS + 8 * h;
  // blockIdx.y = the job: policy output tiles 4y .. 4y+3 (one per wave), or (the last y) the value head -- three times
  // the workgroups of a one-job-after-the-other kernel, each a third as long: the kernel is bound by L2 latency
  constexpr int PB = (NTP + 3) / 4;
  const int job = blockIdx.y;
  // ---- policy
  if (job < PB) {
    const int nt = job * 4 + wave;
    if (nt >= NTP) return;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
    const char *wb = wp + ((size_t)nt * KSP * 2 + h) * 512 + r * 16;
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) {
      const bf16x8 b = *(const bf16x8 *)(wb + (size_t)ks * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 a = *(const bf16x8 *)(fa[m] + ks * 16);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
      }
    }
    const int n = nt * 32 + r;
    if (n < NV) {
      const float bias = bp[n];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (g0 + row < G) logits[(size_t)(g0 + row) * NV + n] = acc[m][i] + bias;
        }
    }
    return;
  }
  // ---- value: hidden tile (wave & 1), K half (wave >> 1)
  {
    const int nt = wave & 1, kh = wave >> 1;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
    const char *wb = w1 + ((size_t)nt * KSV * 2 + h) * 512 + r * 16;
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int ks = kh * KH + k;
      if (ks < KSV) {
        const bf16x8 b = *(const bf16x8 *)(wb + (size_t)ks * 1024);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8 a = *(const bf16x8 *)(fa[m] + KP + ks * 16);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) part[wave][m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[m][i];
    __syncthreads();
    if (wave < 2) {
      const int n = wave * 32 + r;
      const float bn = b1[n], wn = w2[n];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          float t = fmaxf(part[wave][row][r] + part[wave + 2][row][r] + bn, 0.0f) * wn;
#pragma unroll
          for (int off = 16; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
          if (r == 0) vsum[wave][row] = t;
        }
    }
    __syncthreads();
    if (tid < MT * 32 && g0 + tid < G) values[g0 + tid] = tanhf(vsum[0][tid] + vsum[1][tid] + b2);
  }
}

// The QuickModel heads' Linear layers (nn_models/quick_model.py:138-154, 185-199 with config/quick_model_config.py's defaults)
// on the bf16 feature rows [policy0 NV | policy1 NV | pad | value NV | pad]:
//   logits = relu(feat[:, :2NV] @ W1p^T + b1p) @ W2p^T + b2p;   value = tanh(relu(feat[:, KPQ:KPQ+NV] @ W1v^T + b1v) @ w2v + b2v)
// One workgroup = 32 boards.  Phase 1: wave w = hidden tile (w & 1) of head (w >> 1), a bf16 MFMA GEMM read straight from L2,
// bias + ReLU into LDS (the policy's hidden units as bf16, the next GEMM's operand; the value's as f32).  Phase 2: the policy's
// second layer as bf16 MFMA GEMMs from LDS (output tiles wave, wave + 4, ...), the value's as a k-ordered fmaf chain per board.
// w1p [2][KPQ/16][2][32][8], w2p [ceil(NV/32)][4][2][32][8], w1v [2][KVQ/16][2][32][8] bf16, zero padded.
template <int NV>
__global__ __launch_bounds__(256, 2) void k_quick_heads(const uint16_t *__restrict__ feat, int FS, int G, const char *__restrict__ w1p,
                                                        const float *__restrict__ b1p, const char *__restrict__ w2p,
                                                        const float *__restrict__ b2p, const char *__restrict__ w1v,
                                                        const float *__restrict__ b1v, const float *__restrict__ w2v, float b2v,
                                                        float *__restrict__ logits, float *__restrict__ values) {
  constexpr int KPQ = (2 * NV + 15) / 16 * 16, KVQ = (NV + 15) / 16 * 16, NTP = (NV + 31) / 32;
  __shared__ __attribute__((aligned(16))) uint16_t hidp[32][72];  // rows of 144 bytes: 16-byte aligned, conflict-free b128 reads
  __shared__ float hidv[32][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g0 = blockIdx.x * 32;
  const uint16_t *fa = feat + (size_t)min(g0 + r, G - 1) * FS + 8 * h;  // rows past the batch: a copy of the last row, not stored
  {
    const int head = wave >> 1, nt = wave & 1;
    const int ksn = head == 0 ? KPQ / 16 : KVQ / 16;
    const char *wb = (head == 0 ? w1p : w1v) + ((size_t)nt * ksn * 2 + h) * 512 + r * 16;
    const uint16_t *fh = fa + (head == 0 ? 0 : KPQ);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int ks = 0; ks < ksn; ++ks) {
      const bf16x8 b = *(const bf16x8 *)(wb + (size_t)ks * 1024);
      const bf16x8 a = *(const bf16x8 *)(fh + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    const int n = nt * 32 + r;
    const float bn = (head == 0 ? b1p : b1v)[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float v = fmaxf(acc[i] + bn, 0.0f);
      if (head == 0) hidp[row][n] = __builtin_bit_cast(uint16_t, (__bf16)v);
      else hidv[row][n] = v;
    }
  }
  __syncthreads();
  for (int nt = wave; nt < NTP; nt += 4) {
    const char *wb = w2p + ((size_t)nt * 4 * 2 + h) * 512 + r * 16;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 b = *(const bf16x8 *)(wb + (size_t)ks * 1024);
      const bf16x8 a = *(const bf16x8 *)(&hidp[r][ks * 16 + 8 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
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
    for (int k = 0; k < 64; ++k) t = fmaf(hidv[tid][k], w2v[k], t);
    values[g0 + tid] = tanhf(t + b2v);
  }
}

}  // namespace aztower

static thread_local std::string g_tower_err;  // per host thread: engines on different threads do not share (or race on) a message
void tower_set_error(const std::string &msg) { g_tower_err = msg; }  // (az_tower_f32.hip reports through the same az_tower_last_error)

// Ticket counters of the tower launches: {next group, workgroups done} pairs, zero at load and re-armed by each launch's
// last workgroup, so a pair is free again as soon as its launch has ended.  Two launches may share a pair only if they
// cannot overlap, hence:
//   - eager launches: one pair per (GPU, stream) -- launches on one stream run in order; pairs 0 .. EAGER-1, looked up in
//     a small table (least recently used entry recycled when more than EAGER streams have launched towers on the GPU);
//   - launches recorded into a HIP graph (stream capture): ONE pair per capture (keyed on the capture id of
//     hipStreamGetCaptureInfo; the launches inside one capture are serialised by the graph's edges, and a search graph
//     holds hundreds of them), from EAGER .. PAIRS-1, round robin over captures, never handed to an eager stream -- a graph
//     replays with the pair baked in, on whatever stream replays it.  (More than PAIRS - EAGER graphs with tower launches
//     alive at once on one GPU share pairs: replay those on one stream.  INTEGRATION.md states the rule.)
// A __device__ array exists once per GPU; its address is looked up per device, for the device that owns the stream.
#define AZ_TOWER_TICKET_PAIRS 64
#define AZ_TOWER_EAGER_PAIRS 32
static __device__ unsigned int g_tower_tickets[2 * AZ_TOWER_TICKET_PAIRS];
static std::mutex g_ticket_mutex;  // engines on different host threads may launch towers on the same GPU
struct TicketDev {
  unsigned int *base;
  hipStream_t stream[AZ_TOWER_EAGER_PAIRS];
  unsigned long long used[AZ_TOWER_EAGER_PAIRS];
  int n_streams;
  unsigned long long clock;
  unsigned int cap_seq;
  unsigned long long cap_id[AZ_TOWER_TICKET_PAIRS - AZ_TOWER_EAGER_PAIRS];  // capture id that owns each graph pair (0 = none yet)
  unsigned int attr_done;  // bit k: hipFuncSetAttribute issued for kernel variant k on this GPU
};
static TicketDev g_td[64];

// the device a launch on `stream` runs on (the current device for the NULL stream)
int tower_device(hipStream_t stream, int *dev) {  // (also used by az_tower_f32.hip)
  hipDevice_t d = 0;
  if (stream != nullptr && hipStreamGetDevice(stream, &d) == hipSuccess) { *dev = (int)d; return 0; }
  (void)hipGetLastError();
  return hipGetDevice(dev) == hipSuccess ? 0 : -1;
}

// Launches belong to the stream's GPU, whatever device the calling thread has current: every entry point of the network
// kernels (towers AND heads) holds one of these for its whole body and puts the caller's device back on the way out.
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

// -> the ticket pair of this launch, or nullptr (message in g_tower_err).  `dev` is current.
unsigned int *tower_tickets(int dev, hipStream_t stream) {  // (also used by az_tower_f32.hip)
  std::lock_guard<std::mutex> lock(g_ticket_mutex);
  TicketDev &t = g_td[dev];
  if (t.base == nullptr) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tower_tickets)) != hipSuccess || p == nullptr) {
      g_tower_err = "hipGetSymbolAddress(g_tower_tickets) failed";
      return nullptr;
    }
    t.base = (unsigned int *)p;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  if (stream != nullptr && hipStreamGetCaptureInfo(stream, &cs, &cid) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
  if (cs == hipStreamCaptureStatusActive) {
    constexpr unsigned int NG = AZ_TOWER_TICKET_PAIRS - AZ_TOWER_EAGER_PAIRS;
    const unsigned long long key = cid + 1;  // (0 marks an unused entry)
    for (unsigned int i = 0; i < NG; ++i)
      if (t.cap_id[i] == key) return t.base + 2 * (AZ_TOWER_EAGER_PAIRS + i);
    const unsigned int slot = t.cap_seq++ % NG;  // a new capture: the next pair, round robin
    t.cap_id[slot] = key;
    return t.base + 2 * (AZ_TOWER_EAGER_PAIRS + slot);
  }
  ++t.clock;
  int slot = -1;
  for (int i = 0; i < t.n_streams; ++i)
    if (t.stream[i] == stream) { slot = i; break; }
  if (slot < 0) {
    if (t.n_streams < AZ_TOWER_EAGER_PAIRS) slot = t.n_streams++;
    else {  // recycle the least recently used entry (a stream that has long been idle or destroyed)
      slot = 0;
      for (int i = 1; i < AZ_TOWER_EAGER_PAIRS; ++i)
        if (t.used[i] < t.used[slot]) slot = i;
    }
    t.stream[slot] = stream;
  }
  t.used[slot] = t.clock;
  return t.base + 2 * slot;
}

// Scratch of az_tower_heads_forward_static (the entry point for callers that cannot allocate): the feature rows that
// travel from k_tower to k_heads.  One copy per GPU, held until the process ends; `scratch` is only a host-side
// stand-in for the symbol -- the device address is looked up once per GPU (hipGetSymbolAddress, with that GPU current)
// and everything below uses that pointer, never the name.
static __device__ __align__(256) unsigned char scratch[AZ_STATIC_SCRATCH_BYTES];
static unsigned char *g_scratch_ptr[64];

// -> the scratch of GPU `dev` (which must be current), or nullptr
static unsigned char *static_scratch(int dev) {
  std::lock_guard<std::mutex> lock(g_ticket_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      g_tower_err = "hipGetSymbolAddress(scratch) failed";
      return nullptr;
    }
    g_scratch_ptr[dev] = (unsigned char *)p;
  }
  return g_scratch_ptr[dev];
}

// Called by az_create (the library's set-up call for a GPU) with that GPU current: looks the per-GPU addresses up ahead
// of the first launch, so that an entry point first called under stream capture has nothing left to resolve.
int aztower_init_device(int dev) {
  if (dev < 0 || dev >= 64) return AZ_EINVAL;
  if (static_scratch(dev) == nullptr) return AZ_EHIP;
  return tower_tickets(dev, nullptr) != nullptr ? AZ_OK : AZ_EHIP;
}

extern "C" {

const char *az_tower_last_error(void) { return g_tower_err.c_str(); }

int32_t az_heads_feat_stride(int32_t board_size) {
  const int nv = board_size * board_size;
  return (nv + 15) / 16 * 16 + (2 * nv + 15) / 16 * 16;
}

int az_heads_forward(int32_t board_size, int32_t n_boards, const void *feat_bf16_dev, int32_t feat_stride, const void *wp_dev,
                     const float *bp_dev, const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2,
                     float *logits_dev, float *values_dev, void *stream) {
  using namespace aztower;
  if (!feat_bf16_dev || !wp_dev || !bp_dev || !w1_dev || !b1_dev || !w2_dev || !logits_dev || !values_dev) { g_tower_err = "null argument"; return AZ_EINVAL; }
  if (n_boards < 1 || (board_size != 9 && board_size != 15)) { g_tower_err = "n_boards >= 1, board_size 9 or 15"; return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size) || (feat_stride & 7)) { g_tower_err = "feat_stride: a multiple of 8, at least az_heads_feat_stride(board_size)"; return AZ_EINVAL; }
  const int ntp = (board_size * board_size + 31) / 32;
  const dim3 grid((n_boards + 31) / 32, (ntp + 3) / 4 + 1), block(256);  // k_heads: 32 boards x one job (4 policy tiles | value) per workgroup
  hipStream_t s = (hipStream_t)stream;
  StreamDeviceGuard sguard(s);
  if (!sguard.ok) { g_tower_err = "cannot select the stream's device"; return AZ_EHIP; }
  if (board_size == 15)
    hipLaunchKernelGGL((k_heads<225>), grid, block, 0, s, (const uint16_t *)feat_bf16_dev, (int)feat_stride, (int)n_boards, (const char *)wp_dev,
                       bp_dev, (const char *)w1_dev, b1_dev, w2_dev, b2, logits_dev, values_dev);
  else
    hipLaunchKernelGGL((k_heads<81>), grid, block, 0, s, (const uint16_t *)feat_bf16_dev, (int)feat_stride, (int)n_boards, (const char *)wp_dev,
                       bp_dev, (const char *)w1_dev, b1_dev, w2_dev, b2, logits_dev, values_dev);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { g_tower_err = std::string("k_heads launch: ") + hipGetErrorString(err); return AZ_EHIP; }
  return AZ_OK;
}

int az_tower_forward(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const void *x_bf16_dev,
                     const void *packed_ring_dev, const void *packed_leaf_dev, const void *list_dev,
                     const void *count_dev, const void *flags_dev, const void *w0_dev, const void *wl_dev, const float *bias_dev,
                     const float *headw_dev, const float *headb_dev, void *feat_bf16_dev, int32_t feat_stride, void *stream) {
  using namespace aztower;
  if ((!x_bf16_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !bias_dev || !headw_dev || !headb_dev || !feat_bf16_dev ||
      (n_conv_layers > 0 && !wl_dev)) { g_tower_err = "null argument"; return AZ_EINVAL; }
  if (n_boards < 1 || n_conv_layers < 0 || (n_conv_layers & 1)) { g_tower_err = "n_boards >= 1, even n_conv_layers"; return AZ_EINVAL; }
  if ((board_size != 9 && board_size != 15) || (channels != 64 && channels != 128)) { g_tower_err = "board_size must be 9 or 15, channels 64 or 128"; return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { g_tower_err = "feat_stride below az_heads_feat_stride(board_size)"; return AZ_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  hipError_t err;
  int dev = 0, cur = 0;
  if (tower_device(s, &dev) != 0 || dev < 0 || dev >= 64) { g_tower_err = "cannot tell the stream's device"; return AZ_EHIP; }
  if (hipGetDevice(&cur) != hipSuccess) { g_tower_err = "hipGetDevice failed"; return AZ_EHIP; }
  // the counters and the launch belong to the stream's GPU, whatever device the calling thread has current
  struct DeviceGuard {
    int back;
    DeviceGuard(int want, int cur_) : back(want != cur_ ? cur_ : -1) { if (back >= 0) (void)hipSetDevice(want); }
    ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
  } guard(dev, cur);
  unsigned int *tickets = tower_tickets(dev, s);
  if (tickets == nullptr) return AZ_EHIP;
  int variant = 0;
  auto launch = [&](auto kern, size_t lds, int bpw, int nw, int wg_per_cu) -> int {
    if (lds * wg_per_cu > 160 * 1024) { g_tower_err = "tower geometry exceeds the CU's LDS"; return AZ_EINVAL; }
    {  // once per kernel variant and GPU
      std::lock_guard<std::mutex> lock(g_ticket_mutex);
      if (!(g_td[dev].attr_done & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          g_tower_err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
          return AZ_EHIP;
        }
        g_td[dev].attr_done |= 1u << variant;
      }
    }
    const int groups = (n_boards + bpw - 1) / bpw, cap = 256 * wg_per_cu;  // workgroups that fit the chip
    const int grid = groups < cap ? groups : cap;
    const bool use_list = !x_bf16_dev && list_dev && count_dev;
    const bool use_flags = !x_bf16_dev && !use_list && flags_dev;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, s, (const uint16_t *)x_bf16_dev, (const uint32_t *)packed_ring_dev,
                       (const int4 *)packed_leaf_dev, (const int32_t *)(use_list ? list_dev : nullptr),
                       (const int32_t *)(use_list ? count_dev : nullptr), (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)w0_dev, (const char *)wl_dev, bias_dev,
                       n_conv_layers, headw_dev, headb_dev, (uint16_t *)feat_bf16_dev, (int)feat_stride, tickets, 2);
    return AZ_OK;
  };
  // 64 channels: 4 waves, two workgroups per CU.  128 channels: 8 waves (two sets of four, 64 output channels each), one
  // workgroup per CU, one tap (32 KiB of weights) per stage.
  int rc;
  if (channels == 64) {
    using G15 = Geo<15, AZ_TOWER15_BPW, AZ_TOWER15_NPT, AZ_TOWER15_ST, 64>;
    using G9 = Geo<9, 3, 2, AZ_TOWER9_ST, 64>;
    variant = board_size == 15 ? 0 : 1;
    if (board_size == 15) rc = launch(k_tower<15, AZ_TOWER15_BPW, AZ_TOWER15_NPT, AZ_TOWER15_ST, 64>, G15::lds_bytes(), AZ_TOWER15_BPW, G15::NW, AZ_TOWER_WAVES_PER_SIMD);
    else rc = launch(k_tower<9, 3, 2, AZ_TOWER9_ST, 64>, G9::lds_bytes(), 3, G9::NW, AZ_TOWER_WAVES_PER_SIMD);
  } else {
    using G15 = Geo<15, 1, 2, 1, 128>;
    using G9 = Geo<9, 3, 2, 1, 128>;
    variant = board_size == 15 ? 2 : 3;
    if (board_size == 15) rc = launch(k_tower<15, 1, 2, 1, 128>, G15::lds_bytes(), 1, G15::NW, 1);
    else rc = launch(k_tower<9, 3, 2, 1, 128>, G9::lds_bytes(), 3, G9::NW, 1);
  }
  if (rc != AZ_OK) return rc;
  err = hipGetLastError();
  if (err != hipSuccess) { g_tower_err = std::string("k_tower launch: ") + hipGetErrorString(err); return AZ_EHIP; }
  return AZ_OK;
}

int32_t az_static_max_boards(int32_t board_size) {
  if (board_size != 9 && board_size != 15) return 0;
  return (int32_t)(AZ_STATIC_SCRATCH_BYTES / ((size_t)az_heads_feat_stride(board_size) * 2));
}

int az_tower_heads_forward_static(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers,
                                  const void *x_bf16_dev, const void *packed_ring_dev, const void *packed_leaf_dev,
                                  const void *flags_dev, const void *w0_dev, const void *wl_dev, const float *bias_dev,
                                  const float *headw_dev, const float *headb_dev, const void *wp_dev, const float *bp_dev,
                                  const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2, float *logits_dev,
                                  float *values_dev, void *stream) {
  if (board_size != 9 && board_size != 15) { g_tower_err = "board_size must be 9 or 15"; return AZ_EINVAL; }
  if (n_boards < 1) { g_tower_err = "n_boards >= 1"; return AZ_EINVAL; }
  const int32_t max_boards = az_static_max_boards(board_size);
  if (n_boards > max_boards) {
    g_tower_err = "az_tower_heads_forward_static: n_boards " + std::to_string(n_boards) + " exceeds the static scratch (" +
                  std::to_string(AZ_STATIC_SCRATCH_BYTES) + " bytes = " + std::to_string(max_boards) + " boards of this size); "
                  "split the batch or use az_tower_forward + az_heads_forward with a caller-owned buffer";
    return AZ_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  StreamDeviceGuard sguard(s);  // the stream's GPU stays current across the memset, the tower and the heads
  if (!sguard.ok) { g_tower_err = "cannot select the stream's device"; return AZ_EHIP; }
  unsigned char *feat = static_scratch(sguard.dev);  // this GPU's copy of the array (looked up on the first call, or by az_create before)
  if (feat == nullptr) return AZ_EHIP;
  const int32_t stride = az_heads_feat_stride(board_size);
  // the array keeps the last call's contents (another board size has another row layout): the pad columns the kernels
  // never write, and the rows of slots the tower skips, must read as zero
  if (hipMemsetAsync(feat, 0, (size_t)n_boards * stride * 2, s) != hipSuccess) {
    g_tower_err = "hipMemsetAsync(scratch) failed";
    (void)hipGetLastError();
    return AZ_EHIP;
  }
  int rc = az_tower_forward(board_size, channels, n_boards, n_conv_layers, x_bf16_dev, packed_ring_dev, packed_leaf_dev, nullptr,
                            nullptr, flags_dev, w0_dev, wl_dev, bias_dev, headw_dev, headb_dev, feat, stride, stream);
  if (rc != AZ_OK) return rc;
  return az_heads_forward(board_size, n_boards, feat, stride, wp_dev, bp_dev, w1_dev, b1_dev, w2_dev, b2, logits_dev, values_dev,
                          stream);
}

int az_quick_tower_forward(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_blocks, int32_t block_layers,
                           const void *x_bf16_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                           const void *w0_dev, const void *wl_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev,
                           void *feat_bf16_dev, int32_t feat_stride, void *stream) {
  using namespace aztower;
  if ((!x_bf16_dev && (!packed_ring_dev || !packed_leaf_dev)) || !w0_dev || !wl_dev || !bias_dev || !headw_dev || !headb_dev || !feat_bf16_dev) {
    g_tower_err = "null argument";
    return AZ_EINVAL;
  }
  if (n_boards < 1 || n_blocks < 1 || block_layers < 1 || (int64_t)n_blocks * block_layers > 1024) {
    g_tower_err = "quick tower: n_boards >= 1, n_blocks >= 1, block_layers >= 1";
    return AZ_EINVAL;
  }
  if ((board_size != 9 && board_size != 15) || (channels != 64 && channels != 128)) { g_tower_err = "board_size must be 9 or 15, channels 64 or 128"; return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size)) { g_tower_err = "feat_stride below az_heads_feat_stride(board_size)"; return AZ_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cur = 0;
  if (tower_device(s, &dev) != 0 || dev < 0 || dev >= 64) { g_tower_err = "cannot tell the stream's device"; return AZ_EHIP; }
  if (hipGetDevice(&cur) != hipSuccess) { g_tower_err = "hipGetDevice failed"; return AZ_EHIP; }
  struct DeviceGuard {
    int back;
    DeviceGuard(int want, int cur_) : back(want != cur_ ? cur_ : -1) { if (back >= 0) (void)hipSetDevice(want); }
    ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
  } guard(dev, cur);
  unsigned int *tickets = tower_tickets(dev, s);
  if (tickets == nullptr) return AZ_EHIP;
  const int L = n_blocks * block_layers;
  auto launch = [&](auto kern, size_t lds, int bpw, int nw, int wg_per_cu, int variant) -> int {
    if (lds * wg_per_cu > 160 * 1024) { g_tower_err = "tower geometry exceeds the CU's LDS"; return AZ_EINVAL; }
    {
      std::lock_guard<std::mutex> lock(g_ticket_mutex);
      if (!(g_td[dev].attr_done & (1u << variant))) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          g_tower_err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
          return AZ_EHIP;
        }
        g_td[dev].attr_done |= 1u << variant;
      }
    }
    const int groups = (n_boards + bpw - 1) / bpw, cap = 256 * wg_per_cu;
    const int grid = groups < cap ? groups : cap;
    const bool use_flags = !x_bf16_dev && flags_dev;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, s, (const uint16_t *)x_bf16_dev, (const uint32_t *)packed_ring_dev,
                       (const int4 *)packed_leaf_dev, (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (const uint8_t *)(use_flags ? flags_dev : nullptr), n_boards, (const char *)w0_dev, (const char *)wl_dev, bias_dev, L,
                       headw_dev, headb_dev, (uint16_t *)feat_bf16_dev, (int)feat_stride, tickets, (int)block_layers);
    return AZ_OK;
  };
  int rc;
  if (channels == 64) {
    using G15 = Geo<15, AZ_TOWER15_BPW, AZ_TOWER15_NPT, AZ_TOWER15_ST, 64>;
    using G9 = Geo<9, 3, 2, AZ_TOWER9_ST, 64>;
    if (board_size == 15) rc = launch(k_tower<15, AZ_TOWER15_BPW, AZ_TOWER15_NPT, AZ_TOWER15_ST, 64, true>, G15::lds_bytes(), AZ_TOWER15_BPW, G15::NW, AZ_TOWER_WAVES_PER_SIMD, 4);
    else rc = launch(k_tower<9, 3, 2, AZ_TOWER9_ST, 64, true>, G9::lds_bytes(), 3, G9::NW, AZ_TOWER_WAVES_PER_SIMD, 5);
  } else {
    using G15 = Geo<15, 1, 2, 1, 128>;
    using G9 = Geo<9, 3, 2, 1, 128>;
    if (board_size == 15) rc = launch(k_tower<15, 1, 2, 1, 128, true>, G15::lds_bytes(), 1, G15::NW, 1, 6);
    else rc = launch(k_tower<9, 3, 2, 1, 128, true>, G9::lds_bytes(), 3, G9::NW, 1, 7);
  }
  if (rc != AZ_OK) return rc;
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { g_tower_err = std::string("k_tower (quick) launch: ") + hipGetErrorString(err); return AZ_EHIP; }
  return AZ_OK;
}

int az_quick_heads_forward(int32_t board_size, int32_t n_boards, const void *feat_bf16_dev, int32_t feat_stride, const void *w1p_dev,
                           const float *b1p_dev, const void *w2p_dev, const float *b2p_dev, const void *w1v_dev, const float *b1v_dev,
                           const float *w2v_dev, float b2v, float *logits_dev, float *values_dev, void *stream) {
  using namespace aztower;
  if (!feat_bf16_dev || !w1p_dev || !b1p_dev || !w2p_dev || !b2p_dev || !w1v_dev || !b1v_dev || !w2v_dev || !logits_dev || !values_dev) {
    g_tower_err = "null argument";
    return AZ_EINVAL;
  }
  if (n_boards < 1 || (board_size != 9 && board_size != 15)) { g_tower_err = "n_boards >= 1, board_size 9 or 15"; return AZ_EINVAL; }
  if (feat_stride < az_heads_feat_stride(board_size) || (feat_stride & 7)) { g_tower_err = "feat_stride: a multiple of 8, at least az_heads_feat_stride(board_size)"; return AZ_EINVAL; }
  const dim3 grid((n_boards + 31) / 32), block(256);
  hipStream_t s = (hipStream_t)stream;
  StreamDeviceGuard sguard(s);
  if (!sguard.ok) { g_tower_err = "cannot select the stream's device"; return AZ_EHIP; }
  if (board_size == 15)
    hipLaunchKernelGGL((k_quick_heads<225>), grid, block, 0, s, (const uint16_t *)feat_bf16_dev, (int)feat_stride, (int)n_boards, (const char *)w1p_dev,
                       b1p_dev, (const char *)w2p_dev, b2p_dev, (const char *)w1v_dev, b1v_dev, w2v_dev, b2v, logits_dev, values_dev);
  else
    hipLaunchKernelGGL((k_quick_heads<81>), grid, block, 0, s, (const uint16_t *)feat_bf16_dev, (int)feat_stride, (int)n_boards, (const char *)w1p_dev,
                       b1p_dev, (const char *)w2p_dev, b2p_dev, (const char *)w1v_dev, b1v_dev, w2v_dev, b2v, logits_dev, values_dev);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) { g_tower_err = std::string("k_quick_heads launch: ") + hipGetErrorString(err); return AZ_EHIP; }
  return AZ_OK;
}

}  // extern "C"
