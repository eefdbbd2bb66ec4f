// az_engine.hip -- many-tree MCTS self-play engine for MI355X (gfx950, wave64).  C ABI: include/az_engine.h
//
// Data layout in HBM (all owned by the engine, allocated once in az_create):
//   per tree g (one per game slot), a bump-allocated pool of CHILD RECORDS ("edges"), struct of arrays,
//   tree-major so that the children of one node are one contiguous, coalesced run.  A record has a LIGHT part, written
//   at expansion and read for every child a descent scans, and a HEAVY part that exists only once the child has been
//   TOUCHED (chosen by a descent at least once):
//     edgeP    f32  [G][E]   prior of the child (the parent's prior[0, action])                       light, 4 B
//     edgeAct  u8   [G][E]   action of the child (every game here has A <= 256)                       light, 1 B
//     edgeNM   2x32 [G][E]   {child.visits i32, meta u32}                                             heavy, 8 B
//                            meta = touched children of the child's OWN block (bits 0-8) |
//                            child.player == parent.player (bit 9) | 1 + first edge of the child's block (bits 10-31)
//     edgeQ    f64  [G][E]   child.value (running mean, the child's own point of view)                heavy, 8 B
//   An untouched child has visits 0 and value 0.0 by definition, so its PUCT score needs its prior only: a descent
//   reads 5 B per untouched child and 21 B per touched one.  To know WHICH records are touched without a dependent
//   load, a block keeps its touched records first -- positions [0, t) in order of first touch, [t, count) untouched --
//   and t travels with the pointer to the block (the parent record's meta; rootV for the root's block): the descent
//   that first chooses record j >= t swaps it with record t, initialises its heavy part {0 visits, 0.0} and bumps t.
//   Records are therefore NOT in action order: ties of the PUCT score are broken on the action itself (strict '>' in
//   the reference's loop over ascending actions = lowest action wins).
//   A node's block holds one record per legal action, and every game here has exactly
//   PMAX - (moves played) legal actions in a non-terminal position, so a block needs no length field and
//   there is no node table: the root's block starts at edge 0.
//   edgeCnt i32 [G] (edges used; 0 = root not expanded);  rootN i32 [G], rootQ f64 [G]
//   per simulation:  path i32 [G][D] (edge followed at each level), leaf int4 [G]
//   {depth, kind, value bits, player}, leafRing u32 [G][8][SW] = the leaf's 8-state history as bitboards,
//   oldest first (entry 7 = the leaf position itself): the packed form of the NN input
//   per game: the whole trajectory as bitboards, traj u32 [G][PMAX+1][SW] (entry t = position after
//   t plies; the 8-deep history window of the reference is just entries t-7..t), trajPlayer i8,
//   trajVisits u16 [G][PMAX][A] (pi is kept as integer visit counts and divided on emission).
//
// Children are never materialised as boards (the reference clones A boards per expansion,
// core/MCTS_alphazero.py:51-56): a descent replays its moves on a bitboard held in LDS.
//
// Execution model: ONE WAVEFRONT PER TREE (blockDim = 64, grid = slots).  Lanes stride over the
// children of a node; argmax / sums are DPP/shuffle butterflies; the board and the 8-state history
// ring live in LDS.  Every kernel is a single pass over independent trees: no inter-workgroup
// communication, no atomics except the error flag.
//
// Floating point follows the reference's dtype chain exactly (see k_select_gather) and is compiled
// with -ffp-contract=off and correctly rounded f32 divide/sqrt; expf/logf are fmaf polynomials so
// that the CPU oracle used by the tests reproduces them bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/az_engine.h"
#include "az_games.hip.h"

namespace az {

enum : int { KIND_EVAL = 0, KIND_TERMINAL = 1 };
constexpr int WAVE_KMAX = 16;  // most leaves per tree per step
// Trees per workgroup of the per-simulation kernels (one wavefront each).  A CU takes at most 16 workgroups at a time,
// so one-wave workgroups leave it at 16 resident waves whatever the registers allow (measured: 3.7 waves per SIMD, the
// launch 40 % longer than its waves' lifetimes add up to); with 4 trees per workgroup residency is register-bound.
// The waves of a workgroup share nothing and never meet at a barrier (wave_sync).
#ifndef AZ_TREE_WPB
#define AZ_TREE_WPB 4
#endif
constexpr int TREE_WPB = AZ_TREE_WPB;
#ifdef AZ_TREE_OCC  // ask for at least this many waves per SIMD (register budget 512 / AZ_TREE_OCC)
#define AZ_TREE_BOUNDS __launch_bounds__(64 * TREE_WPB, AZ_TREE_OCC)
#else
#define AZ_TREE_BOUNDS __launch_bounds__(64 * TREE_WPB)
#endif
enum : int { ERR_EDGE_CAP = 1, ERR_NODE_CAP = 2, ERR_DEPTH_CAP = 4, ERR_ILLEGAL = 8, ERR_NO_ROOT = 16, ERR_EMIT_SETPOS = 32 };

struct View {  // kernel argument: raw device pointers + sizes
  int G, E, D;
  int K;        // leaves per tree per step (1 = the reference's sequential search); per-simulation arrays hold G*K rows
  int win;      // gomoku: stones in a row that win (WIN_LENGTH)
  int hlen;     // HISTORY_LEN (config/basic.py:10): states in the dense NN input / the emitted samples, 1 .. 8
  float c_puct, epsilon;
  double vl;    // VIRTUAL_LOSS (config/basic.py:25), used by the wave kernels only
  int32_t *waveCnt;              // [G] leaves queued by the last k_select_wave
  unsigned long long *remain;    // [2] {sum, max} of simulations left (az_search_remaining)
  float *edgeP;       // light: prior
  uint8_t *edgeAct;   // light: action
  int2 *edgeNM;       // heavy: {visits, meta}
  double *edgeQ;      // heavy: value
  int32_t *edgeCnt;
  int32_t *rootV;     // [G] touched records of the root's block
  int32_t *rootN;
  double *rootQ;
  int32_t *path;
  int4 *leaf;
  uint32_t *leafRing;
  int32_t *evalCount;  // [1] leaves of this simulation step that need the network ...
  int32_t *evalList;   // [G] ... and their slots, in slot order
  uint8_t *evalFlag;   // [G padded to 16 K] 1 where the slot's leaf needs the network
  uint32_t *traj;
  int8_t *trajPlayer;
  uint16_t *trajVisits;
  int32_t *ply;
  int32_t *startPly;   // [G] 0 for a game started by az_reset_games; the root's ply for a slot loaded by az_set_position
  int32_t *player;
  uint8_t *active;
  float *result;
  unsigned long long *stat;  // [G][NSTAT]
  int32_t *err;
};

constexpr uint32_t META_V = 0x1FFu, META_SAME = 0x200u;  // META_V: touched records of the child's own block
constexpr int NSTAT = 8;  // per-slot counters: sims, depth, children scanned, evals, terminals, expanded, touched scanned, -
constexpr int META_KID_SHIFT = 10;
constexpr int MAX_EDGE_CAP = (1 << 22) - 2;

// ------------------------------------------------------------------------- math helpers -------
__device__ __forceinline__ float az_expf(float x) {  // x <= 0; same fmaf chain as oracle/az_oracle.c
  if (x < -87.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(-n, 0.693145751953125f, x);
  r = fmaf(-n, 1.428606765330187045e-06f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  p = fmaf(p * r, r, r);
  p = p + 1.0f;
  return p * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

__device__ __forceinline__ float az_logf(float x) {  // x > 0, normal
  uint32_t u = __float_as_uint(x);
  int e = (int)(u >> 23) - 127;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
  const float s = (m - 1.0f) / (m + 1.0f);
  const float z = s * s;
  float p = 0.0909090936183929f;
  p = fmaf(p, z, 0.1111111119389534f);
  p = fmaf(p, z, 0.1428571492433548f);
  p = fmaf(p, z, 0.2f);
  p = fmaf(p, z, 0.3333333432674408f);
  p = fmaf(p * z, s, s);
  const float fe = (float)e;
  return fmaf(fe, 0.693145751953125f, fmaf(fe, 1.428606765330187045e-06f, 2.0f * p));
}

__device__ __forceinline__ float az_powf(float x, float y) {  // torch.pow(pi, 1/tao), pi in [0,1]
  if (y == 1.0f) return x;
  if (x <= 0.0f) return 0.0f;
  float t = y * az_logf(x);
  if (t > 0.0f) t = 0.0f;
  return az_expf(t);
}

__device__ __forceinline__ float wave_sum(float v) {  // xor butterfly, fixed order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Counters of a slot: lane k < 7 owns counter k.  A kernel loads its lane's counter with its first batch of loads, the
// bodies add to the register, and the kernel stores it on the way out: no dependent load at the end of a wave's life (a
// load / add / store per counter by lane 0 was one), no atomics (16,384 x 7 of them per launch cost more than they saved).
struct StatAcc {
  unsigned long long v = 0;
  unsigned long long *p = nullptr;
  __device__ __forceinline__ void load(unsigned long long *st, int lane) { if (lane < 7) { p = st + lane; v = *p; } }
  __device__ __forceinline__ void add(int lane, unsigned long long x0, unsigned long long x1, unsigned long long x2, unsigned long long x3,
                                      unsigned long long x4, unsigned long long x5, unsigned long long x6) {
    v += lane == 0 ? x0 : lane == 1 ? x1 : lane == 2 ? x2 : lane == 3 ? x3 : lane == 4 ? x4 : lane == 5 ? x5 : x6;
  }
  __device__ __forceinline__ void store() { if (p) *p = v; }
};

template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
struct bf16_t { uint16_t u; };
struct f16_t { uint16_t u; };
// the batch only ever holds -1, 0, +1: exact in both 16-bit formats
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return bf16_t{(uint16_t)(__float_as_uint(v) >> 16)}; }
template <> __device__ __forceinline__ f16_t cvt_out<f16_t>(float v) {
  return f16_t{(uint16_t)(v == 0.0f ? 0u : (v > 0.0f ? 0x3C00u : 0xBC00u))};
}
template <typename T> __device__ __forceinline__ float cvt_in(T v);
template <> __device__ __forceinline__ float cvt_in<float>(float v) { return v; }
template <> __device__ __forceinline__ float cvt_in<bf16_t>(bf16_t v) { return __uint_as_float((uint32_t)v.u << 16); }
template <> __device__ __forceinline__ float cvt_in<f16_t>(f16_t v) {
  return v.u == 0 ? 0.0f : ((v.u & 0x8000u) ? -1.0f : 1.0f);
}

// NN input of the position whose 8-state history ring is `hist` (oldest at index head):
// core/MCTS_alphazero.py:202-209 + utils/history_manager.py:41-47 + games/base.py:116-121.  hlen = the reference's HISTORY_LEN
// (config/basic.py:10; 8 as shipped): the newest hlen states of the ring, oldest first, then the player plane.
template <class GT, typename OutT>
__device__ __forceinline__ void write_planes(const uint32_t (*hist)[GT::SW], int head, int player, OutT *out, int lane, int hlen) {
  const int cin = hlen * GT::CS + 1, NIN = cin * GT::HW;
  for (int idx = lane; idx < NIN; idx += 64) {
    const int c = idx / GT::HW, cell = idx % GT::HW;
    float v;
    if (c == cin - 1) v = (float)player;
    else v = GT::plane(hist[(head + 8 - hlen + c / GT::CS) & 7], c % GT::CS, cell);
    out[idx] = cvt_out<OutT>(v);
  }
}

// load the ring of the position at ply `ply` of game g: entries ply-7..ply, zeros before the start
template <class GT>
__device__ __forceinline__ void load_ring(const uint32_t *tb, int ply, uint32_t (*hist)[GT::SW], int lane) {
  for (int i = lane; i < 8 * GT::SW; i += 64) {
    const int k = i / GT::SW, w = i % GT::SW, t = ply - 7 + k;
    hist[k][w] = t >= 0 ? tb[(size_t)t * GT::SW + w] : 0u;
  }
}

// ------------------------------------------------------------------------- select + gather ----
// One simulation's descent.  MCTS.search :145-160; MCTSNode.select :59-78 with the dtype chain
//   puct = f32(q64) + ((f32(c_puct) * P) * f32(sqrt64(N_parent))) / f32(n + 1),   strict '>' (lowest
// action wins ties), q64 = +child.value if the child has the parent's player else -child.value.
// One dependent memory round trip per level: the three child streams are read together, and the winner's
// meta word and visit count travel through the argmax butterfly, so nothing is re-read.
// What a descent needs before its first tree level: the slot's words and this lane's share of the history ring + the
// current board.  k_expand_select fills it while the expansion of the previous simulation runs (the words are the
// ones the expansion reads or writes anyway, the ring only depends on the ply), so the descent starts without the
// two memory round trips a stand-alone k_select_gather spends here.
// One level of a descent: the block's records as this lane holds them (records lane, lane + 64, ...).
template <class GT>
struct Lvl {
  static constexpr int NI = (GT::A + 63) / 64;
  float p[NI];
  uint32_t act[NI];
  int2 nm[NI];
  double q[NI];
};
template <class GT>
struct SelPre {
  static constexpr int RW = (8 * GT::SW + 63) / 64;  // ring words per lane
  bool active;
  int ply, player, used, rootN, rootV;
  uint32_t rw[RW], cw;
};
template <class GT>
__device__ __forceinline__ void sel_load_ring(const View &v, const int g, const int lane, SelPre<GT> &s) {
  const uint32_t *tb = v.traj + (size_t)g * (GT::PMAX + 1) * GT::SW;
#pragma unroll
  for (int k = 0; k < SelPre<GT>::RW; ++k) {
    const int i = lane + 64 * k, hk = i / GT::SW, w = i % GT::SW, t = s.ply - 7 + hk;
    s.rw[k] = (i < 8 * GT::SW && t >= 0) ? tb[(size_t)t * GT::SW + w] : 0u;
  }
  s.cw = lane < GT::SW ? tb[(size_t)s.ply * GT::SW + lane] : 0u;
}

// A node's block is read with ALL its loads in flight at once (indices past the block are clamped to its last record and
// ignored), so a tree level costs one memory round trip: the light part of every record, the heavy part of the first
// `vt` (touched) ones only.
template <class GT>
__device__ __forceinline__ void lvl_fetch(const View &v, const size_t eb, const int base, const int cnt, const int vt, const int lane,
                                          Lvl<GT> &L) {
  // (wave-uniform block pointers + a 32-bit lane index: scalar base / vector offset addressing, no 64-bit lane arithmetic)
  const float *P = v.edgeP + eb + base;
  const uint8_t *Ac = v.edgeAct + eb + base;
  const int2 *NM = v.edgeNM + eb + base;
  const double *Q = v.edgeQ + eb + base;
#pragma unroll
  for (int k = 0; k < Lvl<GT>::NI; ++k) {
    const int i = lane + 64 * k;
    const uint32_t ic = (uint32_t)min(i, cnt - 1);
    L.p[k] = P[ic];
    L.act[k] = Ac[ic];
    if (i < vt) { L.nm[k] = NM[ic]; L.q[k] = Q[ic]; }
    else { L.nm[k] = make_int2(0, 0); L.q[k] = 0.0; }
  }
}
struct Pick { int pos, act, n; uint32_t meta; };
// MCTSNode.select :59-78 over one block.  `queued(pos)`: descents of the running wave step already queued through the
// record (virtual loss, wave mode only).  Ties: the reference keeps the first maximum of its loop over ascending
// actions, i.e. the lowest action; records are not in action order here, so the action is part of the comparison key.
// The argmax runs on ONE 64-bit key per candidate -- {order-preserving image of the f32 score | ~(action, position)} --
// so a butterfly round is two lane exchanges and one compare; the winner's visits and meta word are read from the lane
// that holds them afterwards.  (Scores are never -0.0 -- a sum that cancels rounds to +0.0 -- and "+ 0.0f" makes sure;
// a NaN score, which `>` never selects in the reference, maps to key 0 = no candidate.)
template <class GT, bool WAVE, class QF>
__device__ __forceinline__ Pick lvl_pick(const View &v, const Lvl<GT> &L, const int cnt, const int vt, const int nParent, const int lane,
                                         QF queued) {
  // f32(sqrt64(N)) of the reference == the correctly rounded sqrtf(N) (N < 2^24; 53 >= 2 * 24 + 2 bits make the double
  // rounding harmless), which is what this file is compiled to
  const float sqrtN = sqrtf((float)nParent);
  unsigned long long best = 0ull;
#pragma unroll
  for (int k = 0; k < Lvl<GT>::NI; ++k) {
    const int i = lane + 64 * k;
    float sc;
    if (64 * k >= vt) {
      // every record this slot holds is untouched (wave-uniform test): visits 0, value +-0.0, no queued descent, so the
      // score is (c * P) * sqrt(N) / 1 + (+-0) = (c * P) * sqrt(N) exactly -- no divide, no f64
      sc = (v.c_puct * L.p[k]) * sqrtN;
    } else {
      const int n = L.nm[k].x;
      const double q0 = n > 0 ? L.q[k] : 0.0;  // untouched / unvisited: child.value is still its initial 0.0
      double qs = ((uint32_t)L.nm[k].y & META_SAME) ? q0 : -q0;
      if (WAVE) qs = qs - (v.vl * (double)queued(i)) / (double)(n + 1);  // core/MCTS_alphazero.py:70
      const float u = ((v.c_puct * L.p[k]) * sqrtN) / (float)(n + 1);
      sc = (float)qs + u;
    }
    sc = sc + 0.0f;
    const uint32_t b = __float_as_uint(sc);
    const uint32_t mono = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);  // unsigned order == float order
    const uint32_t key = ~((L.act[k] << 16) | (uint32_t)i);                    // larger == lower action
    const unsigned long long cand = (i < cnt && sc == sc) ? (((unsigned long long)mono << 32) | key) : 0ull;
    best = cand > best ? cand : best;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), off, 64), lo = (uint32_t)__shfl_xor((int)(uint32_t)best, off, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    best = o > best ? o : best;
  }
  const uint32_t hiw = (uint32_t)uni((int)(uint32_t)(best >> 32)), key = ~(uint32_t)uni((int)(uint32_t)best);
  Pick w;
  if (hiw == 0u) { w.pos = 0x7fffffff; w.act = 0; w.n = 0; w.meta = 0; return w; }  // no candidate
  w.pos = (int)(key & 0xFFFFu);
  w.act = (int)(key >> 16);
  int n = L.nm[0].x, m = L.nm[0].y;  // the winner's heavy words, from the lane and slot that hold them (zeros if untouched)
#pragma unroll
  for (int k = 1; k < Lvl<GT>::NI; ++k) { n = (w.pos >> 6) == k ? L.nm[k].x : n; m = (w.pos >> 6) == k ? L.nm[k].y : m; }
  w.n = __builtin_amdgcn_readlane(n, w.pos & 63);
  w.meta = (uint32_t)__builtin_amdgcn_readlane(m, w.pos & 63);
  return w;
}
template <class GT>
__device__ __forceinline__ float lvl_take_f(const float (&x)[Lvl<GT>::NI], int slot, int srcLane) {
  float r = x[0];
#pragma unroll
  for (int k = 1; k < Lvl<GT>::NI; ++k) r = slot == k ? x[k] : r;
  return __shfl(r, srcLane, 64);
}
template <class GT>
__device__ __forceinline__ uint32_t lvl_take_u(const uint32_t (&x)[Lvl<GT>::NI], int slot, int srcLane) {
  uint32_t r = x[0];
#pragma unroll
  for (int k = 1; k < Lvl<GT>::NI; ++k) r = slot == k ? x[k] : r;
  return (uint32_t)__shfl((int)r, srcLane, 64);
}
// First touch of record `pos` (>= vt) of the block at `base`: it changes places with record vt, gets its heavy part
// {0 visits, same-player flag, no block yet, value 0.0}, and the block's touched count -- kept beside the pointer to the
// block: the parent record's meta, or rootV -- goes up by one.  L still holds the block as fetched.
template <class GT>
__device__ __forceinline__ void lvl_touch(const View &v, const size_t eb, const int base, const int vt, const Pick &w, const bool same,
                                          const Lvl<GT> &L, const int lane, const int parentEdge, const uint32_t parentMeta, const int g) {
  const float pv = lvl_take_f<GT>(L.p, vt >> 6, vt & 63), pw = lvl_take_f<GT>(L.p, w.pos >> 6, w.pos & 63);
  const uint32_t av = lvl_take_u<GT>(L.act, vt >> 6, vt & 63);
  if (lane == 0) {
    const size_t et = eb + base + vt, ep = eb + base + w.pos;
    if (w.pos != vt) {
      v.edgeP[ep] = pv; v.edgeAct[ep] = (uint8_t)av;
      v.edgeP[et] = pw; v.edgeAct[et] = (uint8_t)w.act;
    }
    v.edgeNM[et] = make_int2(0, same ? (int)META_SAME : 0);
    v.edgeQ[et] = 0.0;
    if (parentEdge >= 0) v.edgeNM[eb + parentEdge].y = (int)(parentMeta + 1u);
    else v.rootV[g] = vt + 1;
  }
}

template <class GT, typename OutT>
__device__ __forceinline__ void select_gather_body(const View &v, OutT *nn_input, const int g, const int lane, const SelPre<GT> &s,
                                                   const int wid, StatAcc &sa) {
  __shared__ uint32_t histAll[TREE_WPB][8][GT::SW];
  __shared__ uint32_t curAll[TREE_WPB][GT::SW];
  uint32_t(*hist)[GT::SW] = histAll[wid];
  uint32_t *cur = curAll[wid];
  static_assert(GT::A <= 256, "edgeAct holds the action in one byte");
  const bool slotActive = s.active;
  const int ply = s.ply;
  int pl = s.player;
  const int used = s.used;
  int nParent = s.rootN;
  if (!slotActive) { if (lane == 0) v.evalFlag[g] = 0; return; }
  const size_t eb = (size_t)g * v.E;

  // the root's block is requested before the history ring is staged, so that it travels together with the ring and the board
  Lvl<GT> L;
  constexpr int RW = SelPre<GT>::RW;
  int vt = s.rootV;
  if (used > 0) lvl_fetch<GT>(v, eb, 0, GT::PMAX - ply, vt, lane, L);
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int i = lane + 64 * k;
    if (i < 8 * GT::SW) hist[i / GT::SW][i % GT::SW] = s.rw[k];
  }
  if (lane < GT::SW) cur[lane] = s.cw;
  wave_sync();

  int head = 0, depth = 0, kind = KIND_EVAL, base = 0, parentEdge = -1;
  uint32_t parentMeta = 0;
  float tval = 0.0f;
  unsigned long long scanned = 0, scannedTouched = 0;
  if (used > 0) {
    for (;;) {
      if (depth >= v.D) { if (lane == 0) atomicOr(v.err, ERR_DEPTH_CAP); return; }
      const int cnt = GT::PMAX - (ply + depth);  // legal actions of a non-terminal position
      const Pick w = lvl_pick<GT, false>(v, L, cnt, vt, nParent, lane, [](int) { return 0; });
      scanned += (unsigned)cnt;
      scannedTouched += (unsigned)vt;
      if (w.pos >= cnt) { if (lane == 0) atomicOr(v.err, ERR_NO_ROOT); return; }
      const int kid = (int)(w.meta >> META_KID_SHIFT);
      const bool touched = w.pos < vt;
      // the child's block: on its way while the move is replayed on the board.  (An untouched record has no block yet, so
      // the registers the touch below reads -- the block as fetched -- are never the ones this fetch overwrites.)
      if (kid) lvl_fetch<GT>(v, eb, kid - 1, cnt - 1, (int)(w.meta & META_V), lane, L);
      const int edge = base + (touched ? w.pos : vt);
      if (lane == 0) v.path[(size_t)g * v.D + depth] = edge;
      ++depth;
      const int mover = pl;
      pl = GT::apply(cur, w.act, mover, lane);
      if (lane < GT::SW) hist[head][lane] = cur[lane];
      head = (head + 1) & 7;
      wave_sync();
      if (!touched) lvl_touch<GT>(v, eb, base, vt, w, pl == mover, L, lane, parentEdge, parentMeta, g);
      if (kid) {
        parentEdge = edge; parentMeta = w.meta;
        base = kid - 1; vt = (int)(w.meta & META_V); nParent = w.n;
        continue;
      }
      const int term = uni(GT::terminal(cur, w.act, mover, ply + depth, lane, v.win));
      if (term != T_NONE) { kind = KIND_TERMINAL; tval = terminal_value(term, pl); }
      break;
    }
  }
  if (lane == 0) {
    v.leaf[g] = make_int4(depth, kind, (int)__float_as_uint(tval), pl);
    v.evalFlag[g] = kind == KIND_EVAL;
  }
  sa.add(lane, 1, (unsigned)depth, scanned, kind == KIND_EVAL, kind == KIND_TERMINAL, 0, scannedTouched);
  if (kind == KIND_EVAL) {
    uint32_t *lr = v.leafRing + (size_t)g * 8 * GT::SW;
    for (int i = lane; i < 8 * GT::SW; i += 64) lr[i] = hist[(head + i / GT::SW) & 7][i % GT::SW];
    if (nn_input) write_planes<GT, OutT>(hist, head, pl, nn_input + (size_t)g * ((v.hlen * GT::CS + 1) * GT::HW), lane, v.hlen);
  }
}

template <class GT, typename OutT>
__global__ AZ_TREE_BOUNDS void k_select_gather(View v, OutT *nn_input) {
  const int wid = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, g = blockIdx.x * TREE_WPB + wid;  // (wid in an SGPR: everything indexed by g stays scalar)
  if (g >= v.G) return;  // (no workgroup barrier anywhere below: a wave may leave early)
  SelPre<GT> s;
  StatAcc sa;
  sa.load(v.stat + (size_t)g * NSTAT, lane);
  s.active = v.active[g] != 0;  // one batch
  s.ply = v.ply[g];
  s.player = v.player[g];
  s.used = v.edgeCnt[g];
  s.rootN = v.rootN[g];
  s.rootV = v.rootV[g];
  sel_load_ring<GT>(v, g, lane, s);  // (all valid memory for an inactive slot too)
  select_gather_body<GT, OutT>(v, nn_input, g, lane, s, wid, sa);
  sa.store();
}

// Compact list of the slots whose leaf needs the network, in slot order: one workgroup; each thread owns a
// contiguous run of `per` slots (a multiple of 16) whose byte flags it reads 16 at a time, then a block-wide scan
// of the per-thread counts in LDS.  (A single atomic counter bumped by 16,384 waves costs ~190 us: one word takes
// ~88 returning atomics per us.)
__global__ __launch_bounds__(1024) void k_compact_leaves(View v, int per) {
  __shared__ int part[1024];
  const int t = threadIdx.x, lo = t * per;
  const uint4 *fl = (const uint4 *)(v.evalFlag + lo);
  int n = 0;
  for (int c = 0; c < per / 16; ++c) {
    const uint4 f = fl[c];
    n += __popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w);  // flags are 0 / 1 bytes
  }
  part[t] = n;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const int add = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int pos = part[t] - n;
  if (n > 0)
    for (int c = 0; c < per / 16; ++c) {
      const uint4 f = fl[c];
      const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if ((w[k >> 2] >> (8 * (k & 3))) & 1u) v.evalList[pos++] = lo + c * 16 + k;
    }
  if (t == 1023) v.evalCount[0] = part[1023];
}

// ------------------------------------------------------------------------- expand + backup ----
// MCTS._get_prior :180-192, MCTSNode.expand :43-57, MCTSNode.update :80-88, MCTS._bacaward :194-200.
// PRE: also prepare the NEXT descent of the same slot (k_expand_select): its slot words ride with this body's first batch
// (edge count and root visits are handed over as this body leaves them), its ring words with the second.
// Wave mode (k_expand_wave): the leaves of one tree are expanded one after the other by the same wavefront; `vs` is the
// leaf's row in the per-simulation arrays (g * K + j), and the tree's edge count / root statistics travel in registers.
struct WaveCarry { int used, rootN; double rootQ; };

template <class GT, bool PRE>
__device__ __forceinline__ void expand_backup_body(const View &v, const float *logits, const float *values, const float *noise,
                                                   const int g, const int lane, SelPre<GT> *sp, const int vs, WaveCarry *wc,
                                                   const int wid, StatAcc &sa) {
  const bool slotActive = v.active[g] != 0;  // tested after the batch below is on its way (all of it is valid memory)
  __shared__ uint32_t stAll[TREE_WPB][GT::SW];
  uint32_t *st = stAll[wid];
  constexpr int A = GT::A, NI = (A + 63) / 64;
  const size_t eb = (size_t)g * v.E;
  const int32_t *path = v.path + (size_t)vs * v.D;
  // Everything that depends only on the slot is requested up front, in one batch (one memory round trip): the leaf
  // record, the leaf board, the network's row, the root statistics and the first 64 path entries (entries past the
  // leaf's depth are stale and ignored).  The second batch is the path's child records; then come the stores.
  const int4 lf = v.leaf[vs];
  const uint32_t bw = lane < GT::SW ? v.leafRing[((size_t)vs * 8 + 7) * GT::SW + lane] : 0u;
  float p[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int a = lane + 64 * i;
    p[i] = (logits + (size_t)vs * A)[(uint32_t)min(a, A - 1)];  // (uniform row pointer + 32-bit lane offset, here and below)
  }
  const float nnv = values[vs];
  const int used = wc ? wc->used : v.edgeCnt[g];
  const int rootN = wc ? wc->rootN : v.rootN[g];
  const double rootQ = wc ? wc->rootQ : v.rootQ[g];
  const int pth = lane < v.D ? path[lane] : 0;
  if (PRE) {
    sp->active = slotActive;
    sp->ply = v.ply[g];
    sp->player = v.player[g];
    sp->used = used;    // (what the descent finds if this body leaves early on an error)
    sp->rootN = rootN;
    sp->rootV = v.rootV[g];  // (0 right after the root's own expansion: k_begin_search cleared it)
  }
  if (!slotActive) return;
  const int depth = uni(lf.x), kind = uni(lf.y);
  // (Tried: requesting the root's block here too, one round trip ahead of the descent that needs it, and patching the one
  // record the backup rewrites -- 35.6 -> 49.6 us per launch, with or without the spills it caused: not kept.)
  if (PRE) sel_load_ring<GT>(v, g, lane, *sp);  // on its way with the path records below
  // this lane's path edge (lane j owns path edge depth-1-j), fetched before the expansion's arithmetic
  const int pj = depth <= 64 ? __shfl(pth, max(depth - 1 - lane, 0), 64) : path[depth - 1 - lane];
  const bool onPath = lane < depth;
  const uint32_t pe = (uint32_t)(onPath ? pj : 0);  // index into this tree's record arrays
  int2 *const tNM = v.edgeNM + eb;
  double *const tQ = v.edgeQ + eb;
  int2 pnm = make_int2(0, 0);  // {visits, meta}: a path record has been touched, so its heavy part exists
  double pq = 0.0;
  if (onPath) { pnm = tNM[pe]; pq = tQ[pe]; }
  const uint32_t pmeta = (uint32_t)pnm.y;
  if (pnm.x == 0) pq = 0.0;  // first visit (the touch wrote 0.0 anyway)

  double val;
  if (kind == KIND_EVAL) {
    if (lane < GT::SW) st[lane] = bw;
    wave_sync();
    bool lg[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      if (a >= A) p[i] = -INFINITY;
      mx = fmaxf(mx, p[i]);
    }
    mx = wave_max(mx);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      p[i] = a < A ? az_expf(p[i] - mx) : 0.0f;
      acc = acc + p[i];
    }
    const float inv = 1.0f / wave_sum(acc);
    const bool mix = (noise != nullptr) && depth == 0;  // node is root_node and use_Dirichlet
    const float keep = (float)(1.0 - (double)v.epsilon);
    float nz[NI];
    if (mix) {  // wave-uniform: the Dirichlet row, all of it in one batch
#pragma unroll
      for (int i = 0; i < NI; ++i) nz[i] = (noise + (size_t)g * A)[(uint32_t)min(lane + 64 * i, A - 1)];
    }
    acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      float x = p[i] * inv;
      if (mix && a < A) x = keep * x + v.epsilon * nz[i];
      lg[i] = a < A && GT::legal(st, a);
      x = x * (lg[i] ? 1.0f : 0.0f);
      p[i] = x;
      acc = acc + x;
    }
    const float tot = wave_sum(acc);
    int nLegal = 0;
    unsigned long long bal[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { bal[i] = __ballot(lg[i]); nLegal += __popcll(bal[i]); }
    if (used + nLegal > v.E) { if (lane == 0) atomicOr(v.err, ERR_EDGE_CAP); return; }
    int off = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      if (lg[i]) {
        const uint32_t pos = (uint32_t)(off + __popcll(bal[i] & ((1ull << lane) - 1ull)));
        (v.edgeP + eb + used)[pos] = p[i] / tot;  // the light part only: 5 B per child; the heavy part waits for the child's first touch
        (v.edgeAct + eb + used)[pos] = (uint8_t)a;
      }
      off += __popcll(bal[i]);
    }
    if (PRE) sp->used = used + nLegal;
    if (wc) wc->used = used + nLegal;
    if (lane == 0) {
      v.edgeCnt[g] = used + nLegal;
      // the leaf's own edge is path[depth-1] = lane 0's path edge: link it to its new block
      if (depth > 0) tNM[pe].y = (int)(pmeta | ((uint32_t)(used + 1) << META_KID_SHIFT));
    }
    sa.add(lane, 0, 0, 0, 0, 0, (unsigned)nLegal, 0);
    val = (double)nnv;  // value.item()
  } else {
    val = (double)__uint_as_float((uint32_t)lf.z);
  }
  // backup: lane j owns path edge depth-1-j; the value it receives has been negated once for every
  // edge BELOW it whose child is the other player's node.
  {
    const bool flip = onPath && (pmeta & META_SAME) == 0u;
    const unsigned long long fm = __ballot(flip);
    const int below = __popcll(fm & ((1ull << lane) - 1ull));
    const double x = (below & 1) ? -val : val;
    const int n = pnm.x + 1;
    const double qNew = pq + (x - pq) / (double)n;
    if (onPath) {
      tNM[pe].x = n;
      tQ[pe] = qNew;
    }
    if (__popcll(fm) & 1) val = -val;
  }
  for (int base = 64; base < depth; base += 64) {  // paths deeper than one wave (never at these board sizes' usual depths)
    const int j = base + lane;
    bool flip = false;
    size_t e = 0;
    if (j < depth) {
      e = eb + path[depth - 1 - j];
      flip = ((uint32_t)v.edgeNM[e].y & META_SAME) == 0u;
    }
    const unsigned long long fm = __ballot(flip);
    if (j < depth) {
      const int below = __popcll(fm & ((1ull << lane) - 1ull));
      const double x = (below & 1) ? -val : val;
      int *np = &v.edgeNM[e].x;
      const int n = *np + 1;
      const double q = n > 1 ? v.edgeQ[e] : 0.0;
      *np = n;
      v.edgeQ[e] = q + (x - q) / (double)n;
    }
    if (__popcll(fm) & 1) val = -val;
  }
  if (PRE) sp->rootN = rootN + 1;
  const double newRootQ = rootQ + (val - rootQ) / (double)(rootN + 1);
  if (wc) { wc->rootN = rootN + 1; wc->rootQ = newRootQ; }
  if (lane == 0) {
    v.rootN[g] = rootN + 1;
    v.rootQ[g] = newRootQ;
  }
}

template <class GT>
__global__ AZ_TREE_BOUNDS void k_expand_backup(View v, const float *logits, const float *values,
                                                                 const float *noise) {
  const int wid = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, g = blockIdx.x * TREE_WPB + wid;  // (wid in an SGPR: everything indexed by g stays scalar)
  if (g >= v.G) return;
  StatAcc sa;
  sa.load(v.stat + (size_t)g * NSTAT, lane);
  expand_backup_body<GT, false>(v, logits, values, noise, g, lane, nullptr, g, nullptr, wid, sa);
  sa.store();
}

// ------------------------------------------------------------------------- wave search --------
// Virtual loss and in-tree leaf batching.  The reference carries the hooks -- MCTSNode.pending / .virtual_loss
// (core/MCTS_alphazero.py:32-33), `q - child.virtual_loss / (child.visits + 1)` in select (:70), USE_VIRTUAL_LOSS /
// VIRTUAL_LOSS / INFER_BATCH_SIZE (config/basic.py:19-30) -- and no code that sets them; the rule implemented here (and
// restated in oracle/az_oracle.c, az_oracle_search_wave) is: a step makes up to min(K, simulations left) descents of a
// tree, one after the other, a child's virtual loss being VIRTUAL_LOSS x (descents already queued in this step that pass
// through it); tree statistics do not change inside a step; a descent that ends on a leaf already queued closes the
// step and is dropped; the queued leaves are evaluated as rows g*K .. g*K+count-1 of ONE network batch and then expanded
// and backed up in queue order.  K = 1 is the sequential search, bit for bit.
template <class GT, typename OutT>
__global__ __launch_bounds__(64) void k_select_wave(View v, OutT *nn_input, int target) {
  constexpr int DL = GT::PMAX + 1, RW = SelPre<GT>::RW;
  __shared__ uint32_t hist[8][GT::SW];
  __shared__ uint32_t cur[GT::SW];
  __shared__ int32_t wp[WAVE_KMAX][DL];  // the queued descents' paths: edge followed at each level
  __shared__ int32_t wlen[WAVE_KMAX];
  const int g = blockIdx.x, lane = threadIdx.x, K = v.K;
  SelPre<GT> s;
  StatAcc sa;
  sa.load(v.stat + (size_t)g * NSTAT, lane);
  s.active = v.active[g] != 0;
  s.ply = v.ply[g];
  s.player = v.player[g];
  s.used = v.edgeCnt[g];
  s.rootN = v.rootN[g];
  s.rootV = v.rootV[g];
  sel_load_ring<GT>(v, g, lane, s);
  if (lane < K) v.evalFlag[(size_t)g * K + lane] = 0;
  const int budget = s.active ? min(K, target - s.rootN) : 0;
  if (budget <= 0) { if (lane == 0) v.waveCnt[g] = 0; return; }
  const int ply = s.ply, used = s.used;
  const size_t eb = (size_t)g * v.E;
  Lvl<GT> L;
  int nq = 0, rootV = s.rootV;  // (the root's touched count moves as this step's descents touch new children)
  unsigned long long scanned = 0, scannedTouched = 0, sumDepth = 0;
  int nEval = 0, nTerm = 0;
  for (int j = 0; j < budget; ++j) {
    wave_sync();  // the previous descent is done with hist / cur, and its record swaps are ordered before this descent's loads
    int vt = rootV;
    if (used > 0) lvl_fetch<GT>(v, eb, 0, GT::PMAX - ply, vt, lane, L);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int i = lane + 64 * k;
      if (i < 8 * GT::SW) hist[i / GT::SW][i % GT::SW] = s.rw[k];
    }
    if (lane < GT::SW) cur[lane] = s.cw;
    wave_sync();
    uint32_t share = (1u << nq) - 1u;  // queued descents that followed the same edges so far
    int head = 0, depth = 0, kind = KIND_EVAL, base = 0, pl = s.player, nParent = s.rootN, parentEdge = -1;
    uint32_t parentMeta = 0;
    float tval = 0.0f;
    if (used > 0) {
      for (;;) {
        if (depth >= v.D || depth >= DL) { if (lane == 0) atomicOr(v.err, ERR_DEPTH_CAP); return; }
        const int cnt = GT::PMAX - (ply + depth);
        const Pick w = lvl_pick<GT, true>(v, L, cnt, vt, nParent, lane, [&](int i) {
          int c = 0;  // queued descents through this child (queued paths only hold touched records, whose positions are final)
          for (uint32_t m = share; m; m &= m - 1u) {
            const int p = __ffs((int)m) - 1;
            c += (wlen[p] > depth && wp[p][depth] == base + i) ? 1 : 0;
          }
          return c;
        });
        scanned += (unsigned)cnt;
        scannedTouched += (unsigned)vt;
        if (w.pos >= cnt) { if (lane == 0) atomicOr(v.err, ERR_NO_ROOT); return; }
        const int kid = (int)(w.meta >> META_KID_SHIFT);
        const bool touched = w.pos < vt;
        if (kid) lvl_fetch<GT>(v, eb, kid - 1, cnt - 1, (int)(w.meta & META_V), lane, L);  // (never together with the touch below)
        const int edge = base + (touched ? w.pos : vt);
        uint32_t ns = 0;
        for (uint32_t m = share; m; m &= m - 1u) {
          const int p = __ffs((int)m) - 1;
          if (wlen[p] > depth && wp[p][depth] == edge) ns |= 1u << p;
        }
        share = ns;
        if (lane == 0) wp[nq][depth] = edge;
        ++depth;
        const int mover = pl;
        pl = GT::apply(cur, w.act, mover, lane);
        if (lane < GT::SW) hist[head][lane] = cur[lane];
        head = (head + 1) & 7;
        wave_sync();
        if (!touched) {
          lvl_touch<GT>(v, eb, base, vt, w, pl == mover, L, lane, parentEdge, parentMeta, g);
          if (parentEdge < 0) rootV = vt + 1;
        }
        if (kid) {
          parentEdge = edge; parentMeta = w.meta;
          base = kid - 1; vt = (int)(w.meta & META_V); nParent = w.n;
          continue;
        }
        const int term = uni(GT::terminal(cur, w.act, mover, ply + depth, lane, v.win));
        if (term != T_NONE) { kind = KIND_TERMINAL; tval = terminal_value(term, pl); }
        break;
      }
    }
    bool dup = false;  // the leaf reached is already queued <=> a queued descent shares the whole path and ends here
    for (uint32_t m = share; m; m &= m - 1u) dup |= wlen[__ffs((int)m) - 1] == depth;
    if (dup) break;
    const size_t vs = (size_t)g * K + nq;
    if (lane == 0) {
      wlen[nq] = depth;
      v.leaf[vs] = make_int4(depth, kind, (int)__float_as_uint(tval), pl);
      v.evalFlag[vs] = kind == KIND_EVAL;
    }
    for (int l = lane; l < depth; l += 64) v.path[vs * v.D + l] = wp[nq][l];
    if (kind == KIND_EVAL) {
      uint32_t *lr = v.leafRing + vs * 8 * GT::SW;
      for (int i = lane; i < 8 * GT::SW; i += 64) lr[i] = hist[(head + i / GT::SW) & 7][i % GT::SW];
      if (nn_input) write_planes<GT, OutT>(hist, head, pl, nn_input + vs * ((v.hlen * GT::CS + 1) * GT::HW), lane, v.hlen);
      ++nEval;
    } else {
      ++nTerm;
    }
    sumDepth += (unsigned)depth;
    ++nq;
  }
  if (lane == 0) {
    v.waveCnt[g] = nq;
  }
  sa.add(lane, (unsigned)nq, sumDepth, scanned, (unsigned)nEval, (unsigned)nTerm, 0, scannedTouched);
  sa.store();
}

template <class GT>
__global__ __launch_bounds__(64) void k_expand_wave(View v, const float *logits, const float *values, const float *noise) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int cnt = v.waveCnt[g];
  WaveCarry wc{v.edgeCnt[g], v.rootN[g], v.rootQ[g]};
  if (!v.active[g] || cnt <= 0) return;
  StatAcc sa;
  sa.load(v.stat + (size_t)g * NSTAT, lane);
  for (int j = 0; j < cnt; ++j) {
    expand_backup_body<GT, false>(v, logits, values, noise, g, lane, nullptr, g * v.K + j, &wc, 0, sa);
    wave_sync();  // the next leaf's path may share edges this one has just updated; `st` is reused
  }
  if (lane == 0) v.waveCnt[g] = 0;
  sa.store();
}

// simulations still to run for a search of `target` simulations: {sum over active trees, max} (az_search_remaining)
__global__ __launch_bounds__(256) void k_search_remaining(View v, int target) {
  __shared__ unsigned long long ssum[256];
  __shared__ int smax[256];
  const int t = threadIdx.x;
  unsigned long long sum = 0;
  int mx = 0;
  for (int g = blockIdx.x * 256 + t; g < v.G; g += gridDim.x * 256) {
    const int r = v.active[g] ? max(target - v.rootN[g], 0) : 0;
    sum += (unsigned)r;
    mx = max(mx, r);
  }
  ssum[t] = sum; smax[t] = mx;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) { ssum[t] += ssum[t + off]; smax[t] = max(smax[t], smax[t + off]); }
    __syncthreads();
  }
  if (t == 0) { atomicAdd(&v.remain[0], ssum[0]); atomicMax(&v.remain[1], (unsigned long long)smax[0]); }
}

// The end of simulation i and the beginning of simulation i+1 of the same tree in one launch (the search loop
// alternates network <-> tree, so this halves the tree launches and their dispatch gaps).  The descent re-reads
// records the backup has just written -- by the same wavefront, through the same L1: a workgroup-scope fence (what
// __syncthreads() carries) orders them.  (An agent-scope __threadfence() here writes back the L2 once per wave: 590 us.)
#ifdef AZ_TREE_STAMP  // diagnostic build: when every wave of k_expand_select started and ended (scripts/dev/tree_stamp.py)
__device__ unsigned long long g_tree_stamp[65536][3];  // {start, end} in 100 MHz ticks, {XCC | SE | CU ids}
#endif

template <class GT, typename OutT>
__global__ AZ_TREE_BOUNDS void k_expand_select(View v, const float *logits, const float *values, const float *noise,
                                                                 OutT *nn_input) {
  const int wid = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, g = blockIdx.x * TREE_WPB + wid;  // (wid in an SGPR: everything indexed by g stays scalar)
  if (g >= v.G) return;
#ifdef AZ_TREE_STAMP
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
  SelPre<GT> s;
  StatAcc sa;
  sa.load(v.stat + (size_t)g * NSTAT, lane);
  expand_backup_body<GT, true>(v, logits, values, noise, g, lane, &s, g, nullptr, wid, sa);
  wave_sync();
  select_gather_body<GT, OutT>(v, nn_input, g, lane, s, wid, sa);
  sa.store();
#ifdef AZ_TREE_STAMP
  if (lane == 0 && g < 65536) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_tree_stamp[g][0] = t_start;
    g_tree_stamp[g][1] = __builtin_amdgcn_s_memrealtime();
    g_tree_stamp[g][2] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

#ifdef AZ_TREE_STAMP
extern "C" int az_tree_stamps(unsigned long long *out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_tree_stamp), sizeof(unsigned long long) * 3 * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------- root policy --------
template <class GT>
__global__ __launch_bounds__(64) void k_root_policy(View v, float *pi, int32_t *visits, double *childQ, float *rootPrior) {
  const int g = blockIdx.x, lane = threadIdx.x;
  constexpr int A = GT::A;
  __shared__ int edgeOf[A];  // action -> root edge + 1, 0 where the action is illegal
  for (int a = lane; a < A; a += 64) edgeOf[a] = 0;
  __syncthreads();
  const bool have = v.active[g] && v.edgeCnt[g] > 0;
  const size_t eb = (size_t)g * v.E;
  float tot = 1.0f;
  const int vt = have ? v.rootV[g] : 0;  // records [0, vt) of the root's block have been touched: only they hold visits
  if (have) {
    const int cnt = GT::PMAX - v.ply[g];
    int s = 0;
    for (int i = lane; i < cnt; i += 64) {
      if (i < vt) s += v.edgeNM[eb + i].x;
      edgeOf[v.edgeAct[eb + i]] = i + 1;
    }
    tot = (float)wave_sum_i(s);
  }
  __syncthreads();
  for (int a = lane; a < A; a += 64) {
    const int k = edgeOf[a];
    const size_t e = eb + (k > 0 ? k - 1 : 0);
    const size_t o = (size_t)g * A + a;
    const int n = (k && k - 1 < vt) ? v.edgeNM[e].x : 0;
    if (pi) pi[o] = k ? (float)n / tot : 0.0f;
    if (visits) visits[o] = n;
    if (childQ) childQ[o] = n > 0 ? v.edgeQ[e] : 0.0;
    if (rootPrior) rootPrior[o] = k ? v.edgeP[e] : 0.0f;
  }
}

template <class GT>
__global__ __launch_bounds__(64) void k_root_nn_state(View v, float *out) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (!v.active[g]) return;
  __shared__ uint32_t hist[8][GT::SW];
  const int ply = v.ply[g];
  load_ring<GT>(v.traj + (size_t)g * (GT::PMAX + 1) * GT::SW, ply, hist, lane);
  __syncthreads();
  write_planes<GT, float>(hist, 0, v.player[g], out + (size_t)g * ((v.hlen * GT::CS + 1) * GT::HW), lane, v.hlen);
}

// ------------------------------------------------------------------------- play a move --------
// training/self_play.py:73-83, games/base.py:22-33, and the result flip of :86-89.
template <class GT>
__global__ __launch_bounds__(64) void k_play_moves(View v, const float *invTao, const float *expo, const int32_t *forced,
                                                   int32_t *actions, uint8_t *done) {
  const int g = blockIdx.x, lane = threadIdx.x;
  constexpr int A = GT::A;
  if (!v.active[g]) {
    if (lane == 0) { if (actions) actions[g] = -1; if (done) done[g] = 0; }
    return;
  }
  __shared__ uint32_t cur[GT::SW];
  __shared__ uint16_t vis[A];
  for (int a = lane; a < A; a += 64) vis[a] = 0;
  const int ply = v.ply[g], player = v.player[g];
  uint32_t *tb = v.traj + (size_t)g * (GT::PMAX + 1) * GT::SW;
  if (lane < GT::SW) cur[lane] = tb[(size_t)ply * GT::SW + lane];
  __syncthreads();
  if (v.edgeCnt[g] == 0 || ply >= GT::PMAX) { if (lane == 0) atomicOr(v.err, ERR_NO_ROOT); return; }
  const int cnt = GT::PMAX - ply;
  const size_t eb = (size_t)g * v.E;
  const int vt = v.rootV[g];  // untouched root children have no visits: pi = 0, never sampled (w / expo = 0 loses to any visited child)
  int s = 0;
  for (int i = lane; i < vt; i += 64) s += v.edgeNM[eb + i].x;
  const float tot = (float)wave_sum_i(s);
  const float it = invTao[g];
  float best = -INFINITY;
  int bestA = 0x7fffffff;
  for (int i = lane; i < cnt; i += 64) {
    const int a = (int)v.edgeAct[eb + i], n = i < vt ? v.edgeNM[eb + i].x : 0;
    vis[a] = (uint16_t)n;
    const float pi = (float)n / tot;
    const float q = az_powf(pi, it) / expo[(size_t)g * A + a];
    if (q > best || (q == best && a < bestA)) { best = q; bestA = a; }  // (records are not in action order)
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(bestA, off, 64);
    if (ob > best || (ob == best && oa < bestA)) { best = ob; bestA = oa; }
  }
  int a = uni(bestA);
  if (forced && forced[g] >= 0) a = uni(forced[g]);
  __syncthreads();
  if (a < 0 || a >= A || !GT::legal(cur, a)) { if (lane == 0) atomicOr(v.err, ERR_ILLEGAL); return; }
  uint16_t *tv = v.trajVisits + ((size_t)g * GT::PMAX + ply) * A;
  for (int k = lane; k < A; k += 64) tv[k] = vis[k];
  const int np = GT::apply(cur, a, player, lane);
  if (lane < GT::SW) tb[(size_t)(ply + 1) * GT::SW + lane] = cur[lane];
  const int term = uni(GT::terminal(cur, a, player, ply + 1, lane, v.win));
  if (lane == 0) {
    v.trajPlayer[(size_t)g * (GT::PMAX + 1) + ply] = (int8_t)player;
    v.ply[g] = ply + 1;
    v.player[g] = np;
    if (actions) actions[g] = a;
    if (done) done[g] = term != T_NONE;
    if (term != T_NONE) {
      v.active[g] = 0;
      float r = terminal_value(term, np);  // game.evaluation(): the player to move's view
      v.result[g] = (np == player) ? r : -r;  // ... flipped to the last mover's (self_play.py:89)
    }
  }
}

// ------------------------------------------------------------------------- emit samples -------
// training/self_play.py:110-139: sample j of a finished game is ply t = T-1-j; its z is the result if
// that ply's mover is the last mover, else its negation (the cumulative flips of :131 telescope).
template <class GT>
__global__ __launch_bounds__(64) void k_emit_samples(View v, const int32_t *slots, const int32_t *offsets, float *states,
                                                     float *pis, float *zs) {
  const int g = slots[blockIdx.x], j = blockIdx.y, lane = threadIdx.x;
  constexpr int A = GT::A;
  if (g < 0 || g >= v.G) return;
  if (v.startPly[g] != 0) {  // a slot loaded by az_set_position has no trajectory before its root: nothing to emit
    if (j == 0 && lane == 0) atomicOr(v.err, ERR_EMIT_SETPOS);
    return;
  }
  const int T = v.ply[g];
  if (j >= T) return;
  const int t = T - 1 - j;
  const size_t row = (size_t)offsets[blockIdx.x] + j;
  __shared__ uint32_t hist[8][GT::SW];
  load_ring<GT>(v.traj + (size_t)g * (GT::PMAX + 1) * GT::SW, t, hist, lane);
  __syncthreads();
  const int8_t *tp = v.trajPlayer + (size_t)g * (GT::PMAX + 1);
  const int pl = tp[t];
  write_planes<GT, float>(hist, 0, pl, states + row * ((v.hlen * GT::CS + 1) * GT::HW), lane, v.hlen);
  const uint16_t *tv = v.trajVisits + ((size_t)g * GT::PMAX + t) * A;
  int s = 0;
  for (int a = lane; a < A; a += 64) s += tv[a];
  const float tot = (float)wave_sum_i(s);
  for (int a = lane; a < A; a += 64) pis[row * A + a] = (float)tv[a] / tot;
  if (lane == 0) {
    const float r = v.result[g];
    zs[row] = (pl == tp[T - 1]) ? r : -r;
  }
}

// ------------------------------------------------------------------------- augmentation -------
// The 8-fold symmetry expansion the reference's pools apply on put:
// utils/data_enhancer.py:13-70 for the stone games (for k in 0..3: rot90(k), then rot90(k) + flip of the
// last dim) and examples/dots_and_boxes/game.py:158-320 for dots-and-boxes (edge / box permutations, where
// "flip" is a transpose).  Gather form: output variant t = 2k + flip of sample i goes to row
// (write_idx + 8 i + t) mod capacity, i.e. straight into ring-buffer layout (utils/share_ring_buffer.py:39-46).
struct AugTables {
  uint8_t invEdge[8][60];  // dots-and-boxes (square boards up to 5 x 5 boxes): source action of destination action, per variant
  uint8_t invBox[8][25];
};

template <class GT>
__global__ __launch_bounds__(64) void k_augment(AugTables tb, const float *in_s, const float *in_p, const float *in_z,
                                                float *out_s, float *out_p, float *out_z, long long cap, long long wr, int hlen) {
  const int i = blockIdx.x, t = blockIdx.y, lane = threadIdx.x, k = t >> 1;
  const bool flip = t & 1;
  constexpr int A = GT::A, HW = GT::HW, N = GT::H;
  const int CIN = hlen * GT::CS + 1, NIN = CIN * HW;
  long long row = wr + 8ll * i + t;
  if (cap > 0) row %= cap;
  const float *is = in_s + (size_t)i * NIN, *ip = in_p + (size_t)i * A;
  float *os = out_s + (size_t)row * NIN, *op = out_p + (size_t)row * A;
  if constexpr (GT::CS == 1) {
    auto src = [&](int cell) {
      const int oi = cell / N, oj = cell % N, jj = flip ? N - 1 - oj : oj;
      const int si = k == 0 ? oi : (k == 1 ? jj : (k == 2 ? N - 1 - oi : N - 1 - jj));
      const int sj = k == 0 ? jj : (k == 1 ? N - 1 - oi : (k == 2 ? N - 1 - jj : oi));
      return si * N + sj;
    };
    for (int idx = lane; idx < NIN; idx += 64) os[idx] = is[(idx / HW) * HW + src(idx % HW)];
    for (int a = lane; a < A; a += 64) op[a] = ip[src(a)];
  } else {  // a square board of BC x BC boxes, WD x WD dots (az_augment_samples refuses the others, as the reference's flip does)
    constexpr int BC = GT::BC, WD = GT::WD, NH = GT::NH;
    for (int idx = lane; idx < NIN; idx += 64) {
      const int c = idx / HW, cell = idx % HW, r = cell / WD, col = cell % WD;
      float v = 0.0f;
      if (c == CIN - 1) v = is[idx];
      else {
        const int blk = c / 3, ch = c % 3;
        if (ch == 2) {
          if (r < BC && col < BC) { const int b = tb.invBox[t][r * BC + col]; v = is[(blk * 3 + 2) * HW + (b / BC) * WD + b % BC]; }
        } else if (ch == 0 ? col < BC : r < BC) {
          const int a = tb.invEdge[t][ch == 0 ? r * BC + col : NH + r * WD + col];
          v = a < NH ? is[(blk * 3) * HW + (a / BC) * WD + a % BC] : is[(blk * 3 + 1) * HW + ((a - NH) / WD) * WD + (a - NH) % WD];
        }
      }
      os[idx] = v;
    }
    for (int a = lane; a < A; a += 64) op[a] = ip[tb.invEdge[t][a]];
  }
  if (lane == 0) out_z[row] = in_z[i];
}

// ------------------------------------------------------------------------- per-game RNG -------
// The reference seeds one torch generator per WORKER (training/alphazero_trainer.py:175-177) and draws, per move, a
// Dirichlet(alpha) row (core/MCTS_alphazero.py:187) and the multinomial sample of training/self_play.py:73-74 (on ATen's
// single-sample path: argmax(w / Exponential(1))).  Thousands of concurrent games cannot share a stream, and a stream per
// slot would make a game's noise depend on where it happens to run.  Here every row is a pure function of
// (seed, GLOBAL game id, ply, purpose): Philox4x32-10 with key = seed, counter = {element | attempt << 16,
// ply | purpose << 16, game id lo, game id hi}.  A game therefore plays the same moves whatever slot, slot count, rank or
// world size it lands on ("per-game RNG = Philox(seed, subsequence = game id)").
//   Exponential(1):  -log(u),  u = ((w0 >> 9) + 0.5) * 2^-23 in (0, 1)
//   Gamma(alpha <= 1, 1) by Ahrens-Dieter GS (uniforms, exp, log only; no normals): b = (e + alpha) / e; repeat
//     P = b U;  P <= 1: X = P^(1/alpha), accept if W <= exp(-X);  else X = -log((b - P) / alpha), accept if W <= X^(alpha-1)
//   with a fresh Philox block per attempt; Dirichlet = Gamma row / its sum (the engine's wave-order sum).
// exp / log are the fmaf polynomials above, so oracle/az_oracle.c (az_oracle_draw_rng) reproduces every row bit for bit.
// Not torch's Dirichlet stream: the tables stay DATA at the ABI (az_expand_backup / az_play_moves take any rows), and a
// seeded replay of the reference uses rows drawn by torch (SelfPlayRunner(reference_rng=True)).
struct Philox { uint32_t c[4]; };
__host__ __device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float rng_u01(uint32_t w) { return ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f; }  // (0, 1), exact
__device__ __forceinline__ float rng_pow(float x, float y) {  // x > 0, result <= 1 wanted: exp(min(y log x, 0))
  float t = y * az_logf(x);
  if (t > 0.0f) t = 0.0f;
  return az_expf(t);
}
enum : uint32_t { RNG_DIRICHLET = 0u, RNG_EXPO = 1u };
constexpr int RNG_MAX_ATTEMPTS = 64;

__device__ __forceinline__ float rng_gamma(float alpha, float inv_alpha, float b, uint32_t a, uint32_t c1, uint32_t g0, uint32_t g1,
                                           uint32_t k0, uint32_t k1) {
  for (int t = 0; t < RNG_MAX_ATTEMPTS; ++t) {
    const Philox r = philox4x32_10(a | ((uint32_t)t << 16), c1, g0, g1, k0, k1);
    const float U = rng_u01(r.c[0]), W = rng_u01(r.c[1]);
    const float P = b * U;
    if (P <= 1.0f) {
      const float X = rng_pow(P, inv_alpha);
      if (W <= az_expf(-X)) return X;
    } else {
      const float rest = (b - P) / alpha;
      if (!(rest > 0.0f)) continue;
      const float X = -az_logf(rest);
      if (X > 0.0f && W <= rng_pow(X, alpha - 1.0f)) return X;
    }
  }
  return 1.0e-30f;  // (probability ~ 1e-30 per element: never seen; keeps the row finite)
}

template <class GT>
__global__ __launch_bounds__(64) void k_draw_rng(View v, unsigned long long seed, const long long *gameIds, float alpha, float *noise,
                                                 float *expo) {
  const int g = blockIdx.x, lane = threadIdx.x;
  constexpr int A = GT::A, NI = (A + 63) / 64;
  if (!v.active[g]) return;
  const unsigned long long gid = (unsigned long long)gameIds[g];
  const uint32_t g0 = (uint32_t)gid, g1 = (uint32_t)(gid >> 32), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t ply = (uint32_t)v.ply[g];
  if (expo) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      if (a < A) expo[(size_t)g * A + a] = -az_logf(rng_u01(philox4x32_10((uint32_t)a, ply | (RNG_EXPO << 16), g0, g1, k0, k1).c[0]));
    }
  }
  if (noise) {
    const float inv_alpha = 1.0f / alpha, b = (2.7182817459106445f + alpha) / 2.7182817459106445f;
    float x[NI], acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      x[i] = a < A ? rng_gamma(alpha, inv_alpha, b, (uint32_t)a, ply | (RNG_DIRICHLET << 16), g0, g1, k0, k1) : 0.0f;
      acc = acc + x[i];
    }
    const float tot = wave_sum(acc);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int a = lane + 64 * i;
      if (a < A) noise[(size_t)g * A + a] = x[i] / tot;
    }
  }
}

// ------------------------------------------------------------------------- small kernels ------
template <class GT>
__global__ void k_reset_games(View v, const uint8_t *mask) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= v.G || (mask && !mask[g])) return;
  uint32_t *tb = v.traj + (size_t)g * (GT::PMAX + 1) * GT::SW;
  for (int w = 0; w < GT::SW; ++w) tb[w] = 0u;
  v.ply[g] = 0;
  v.startPly[g] = 0;
  v.player[g] = 1;
  v.active[g] = 1;
  v.result[g] = 0.0f;
  v.edgeCnt[g] = 0;
  v.rootV[g] = 0;
  v.rootN[g] = 0;
  v.rootQ[g] = 0.0;
  v.waveCnt[g] = 0;
}

__global__ void k_begin_search(View v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= v.G) return;
  v.edgeCnt[g] = 0;
  v.rootV[g] = 0;
  v.rootN[g] = 0;
  v.rootQ[g] = 0.0;
  v.waveCnt[g] = 0;
}

__global__ void k_game_info(View v, int32_t *plies, int32_t *player, uint8_t *active, float *result) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= v.G) return;
  if (plies) plies[g] = v.ply[g];
  if (player) player[g] = v.player[g];
  if (active) active[g] = v.active[g];
  if (result) result[g] = v.result[g];
}

// Synthetic evaluator, identical to az_oracle_eval_hash in oracle/az_oracle.c.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <typename InT>
__global__ __launch_bounds__(64) void k_eval_hash(int nin, int A, const InT *nn_input, float scale, float *logits, float *values) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const InT *x = nn_input + (size_t)g * nin;
  uint32_t h = 0;
  for (int i = lane; i < nin; i += 64) {
    const uint32_t code = (uint32_t)((int)cvt_in<InT>(x[i]) + 1);
    h += mix32((uint32_t)i * 2654435761u + code * 40503u + 12345u);
  }
  h = (uint32_t)wave_sum_i((int)h);
  for (int a = lane; a < A; a += 64) {
    const uint32_t r = mix32(h + (uint32_t)a * 0x9E3779B9u + 1u);
    logits[(size_t)g * A + a] = scale * ((float)((r >> 8) & 0xFFFFu) * (1.0f / 65536.0f) - 0.5f);
  }
  if (lane == 0) {
    const uint32_t r = mix32(h ^ 0x00ABCDEFu);
    values[g] = (float)((r >> 8) & 0xFFFFu) * (1.0f / 32768.0f) - 1.0f;
  }
}

// The same evaluator fed from the PACKED leaf batch (8 bitboard states + the player of the leaf record): what a caller
// that passes nn_input == NULL to az_select_gather / az_expand_select has to read -- same numbers as k_eval_hash on the
// dense planes of the same leaves (the hash is a commutative sum over the input elements).
template <class GT>
__global__ __launch_bounds__(64) void k_eval_hash_packed(View v, float scale, float *logits, float *values) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (!v.evalFlag[g]) return;  // terminal leaf / inactive slot: the row keeps its old contents, as with the dense batch
  constexpr int A = GT::A;
  const int cin = v.hlen * GT::CS + 1, NIN = cin * GT::HW, skip = 8 - v.hlen;
  __shared__ uint32_t ring[8][GT::SW];
  for (int i = lane; i < 8 * GT::SW; i += 64) ring[i / GT::SW][i % GT::SW] = v.leafRing[(size_t)g * 8 * GT::SW + i];
  const int player = v.leaf[g].w;
  __syncthreads();
  uint32_t h = 0;
  for (int i = lane; i < NIN; i += 64) {
    const int c = i / GT::HW, cell = i % GT::HW;
    const float x = c == cin - 1 ? (float)player : GT::plane(ring[skip + c / GT::CS], c % GT::CS, cell);
    const uint32_t code = (uint32_t)((int)x + 1);
    h += mix32((uint32_t)i * 2654435761u + code * 40503u + 12345u);
  }
  h = (uint32_t)wave_sum_i((int)h);
  for (int a = lane; a < A; a += 64) {
    const uint32_t r = mix32(h + (uint32_t)a * 0x9E3779B9u + 1u);
    logits[(size_t)g * A + a] = scale * ((float)((r >> 8) & 0xFFFFu) * (1.0f / 65536.0f) - 0.5f);
  }
  if (lane == 0) {
    const uint32_t r = mix32(h ^ 0x00ABCDEFu);
    values[g] = (float)((r >> 8) & 0xFFFFu) * (1.0f / 32768.0f) - 1.0f;
  }
}

}  // namespace az

// =========================================================================== host side =========
using namespace az;

struct az_engine {
  az_config cfg;
  az_dims dims;
  View v;
  std::vector<void *> allocs;
  std::string err;
  int nin;
};

static std::string g_create_err;
int aztower_init_device(int dev);  // az_tower.hip: per-GPU addresses of the library's static device arrays

// Every entry point runs with the engine's GPU current, whatever device the calling thread had selected (a process with
// several engines, or a caller that never called hipSetDevice), and puts the caller's device back on the way out.
struct DeviceGuard {
  int back = -1;
  explicit DeviceGuard(int want) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) back = cur;
  }
  ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
};

#define AZ_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      e->err = std::string(#call) + ": " + hipGetErrorString(_e);                           \
      return AZ_EHIP;                                                                       \
    }                                                                                       \
  } while (0)

// dispatch on the game: body sees `GT`
#define AZ_DISPATCH(e, BODY)                                                                \
  do {                                                                                      \
    switch ((e)->cfg.game) {                                                                \
      case AZ_GAME_TICTACTOE: { using GT = az::TicTacToe; BODY; } break;                    \
      case AZ_GAME_DAB:  /* examples/dots_and_boxes/game.py:24-25 BOX_ROWS x BOX_COLS, 2 .. 5 each */ \
        switch ((e)->cfg.board_size * 8 + (e)->cfg.board_cols) {                            \
          AZ_DAB_CASE(2, 2, BODY) AZ_DAB_CASE(2, 3, BODY) AZ_DAB_CASE(2, 4, BODY) AZ_DAB_CASE(2, 5, BODY) \
          AZ_DAB_CASE(3, 2, BODY) AZ_DAB_CASE(3, 4, BODY) AZ_DAB_CASE(3, 5, BODY)           \
          AZ_DAB_CASE(4, 2, BODY) AZ_DAB_CASE(4, 3, BODY) AZ_DAB_CASE(4, 4, BODY) AZ_DAB_CASE(4, 5, BODY) \
          AZ_DAB_CASE(5, 2, BODY) AZ_DAB_CASE(5, 3, BODY) AZ_DAB_CASE(5, 4, BODY) AZ_DAB_CASE(5, 5, BODY) \
          default: { using GT = az::Dab<3, 3>; BODY; } break;                               \
        }                                                                                   \
        break;                                                                              \
      default:  /* gomoku N x N: examples/gomoku/game.py's BOARD_SIZE, any value 3 .. 15 */       \
        switch ((e)->cfg.board_size) {                                                      \
          AZ_GOMOKU_CASE(3, BODY) AZ_GOMOKU_CASE(4, BODY) AZ_GOMOKU_CASE(5, BODY)           \
          AZ_GOMOKU_CASE(6, BODY) AZ_GOMOKU_CASE(7, BODY) AZ_GOMOKU_CASE(8, BODY)           \
          AZ_GOMOKU_CASE(9, BODY) AZ_GOMOKU_CASE(10, BODY) AZ_GOMOKU_CASE(11, BODY)         \
          AZ_GOMOKU_CASE(12, BODY) AZ_GOMOKU_CASE(13, BODY) AZ_GOMOKU_CASE(14, BODY)        \
          default: { using GT = az::Gomoku<15>; BODY; } break;                              \
        }                                                                                   \
        break;                                                                              \
    }                                                                                       \
  } while (0)
// (variadic: by the time BODY arrives here its own macros are expanded and their commas stand at the top level)
#define AZ_GOMOKU_CASE(NN, ...) case NN: { using GT = az::Gomoku<NN>; __VA_ARGS__; } break;
#define AZ_DAB_CASE(RR, CC, ...) case RR * 8 + CC: { using GT = az::Dab<RR, CC>; __VA_ARGS__; } break;

template <class GT>
static void fill_dims(az_dims &d) {
  d.n_action = GT::A; d.state_channels = GT::CS; d.height = GT::H; d.width = GT::WD;
  d.nn_channels = GT::CIN; d.max_plies = GT::PMAX; d.state_words = GT::SW;  // (nn_channels: az_create rescales it by history_len)
}

static int launch_check(az_engine *e, const char *what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    e->err = std::string(what) + ": " + hipGetErrorString(err);
    return AZ_EHIP;
  }
  return AZ_OK;
}

extern "C" {

const char *az_last_error(const az_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int az_create(const az_config *cfg, az_engine **out) {
  if (!cfg || !out) { g_create_err = "null argument"; return AZ_EINVAL; }
  *out = nullptr;
  if (cfg->game < AZ_GAME_TICTACTOE || cfg->game > AZ_GAME_GOMOKU) { g_create_err = "unknown game"; return AZ_EINVAL; }
  const int win = cfg->win_length == 0 ? 5 : cfg->win_length;  // examples/gomoku/game.py:22
  const int hlen = cfg->history_len == 0 ? 8 : cfg->history_len;  // config/basic.py:10 _HISTORY_LEN
  if (hlen < 1 || hlen > 8) { g_create_err = "history_len (HISTORY_LEN, config/basic.py:10) 1 .. 8; 0 = the reference's 8"; return AZ_EINVAL; }
  // dots-and-boxes: board_size = BOX_ROWS (0 = the reference's 3), board_cols = BOX_COLS (0 = the same as the rows)
  const int dab_rows = cfg->board_size == 0 ? 3 : cfg->board_size, dab_cols = cfg->board_cols == 0 ? dab_rows : cfg->board_cols;
  if (cfg->game == AZ_GAME_DAB && (dab_rows < 2 || dab_rows > 5 || dab_cols < 2 || dab_cols > 5)) {
    g_create_err = "dots-and-boxes: board_size (BOX_ROWS; 0 = 3) and board_cols (BOX_COLS; 0 = the rows) 2 .. 5"; return AZ_EINVAL;
  }
  if (cfg->game == AZ_GAME_GOMOKU && (cfg->board_size < 3 || cfg->board_size > 15 || win < 3 || win > 6 || win > cfg->board_size)) {
    g_create_err = "gomoku: board_size 3 .. 15, win_length 3 .. 6 (0 = 5) and <= board_size"; return AZ_EINVAL;
  }
  if (cfg->n_slots < 1 || cfg->max_sims < 1 || cfg->max_sims > 65535) {
    g_create_err = "n_slots >= 1 and 1 <= max_sims <= 65535 required"; return AZ_EINVAL;
  }
  if (cfg->nn_dtype < AZ_NN_F32 || cfg->nn_dtype > AZ_NN_F16) { g_create_err = "bad nn_dtype"; return AZ_EINVAL; }
  if (cfg->leaves_per_step < 0 || cfg->leaves_per_step > WAVE_KMAX) { g_create_err = "leaves_per_step must be 0 .. 16"; return AZ_EINVAL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device < 0 || cfg->device >= ndev) {
    g_create_err = "no usable HIP device (this library has no CPU path)"; return AZ_EHIP;
  }
  az_engine *e = new az_engine();
  e->cfg = *cfg;
  e->cfg.win_length = win;
  e->cfg.history_len = hlen;
  if (cfg->game == AZ_GAME_DAB) { e->cfg.board_size = dab_rows; e->cfg.board_cols = dab_cols; }
  memset(&e->dims, 0, sizeof(e->dims));
  AZ_DISPATCH(e, fill_dims<GT>(e->dims));
  az_dims &d = e->dims;
  d.nn_channels = hlen * d.state_channels + 1;  // HISTORY_LEN * C + 1: core/MCTS_alphazero.py:202-209
  const int A = d.n_action;
  d.n_slots = cfg->n_slots;
  d.node_cap = cfg->max_sims + 1;
  const long long worst = (long long)cfg->max_sims * A;
  const long long want = cfg->edge_cap > 0 ? (long long)cfg->edge_cap : worst;
  if (want > MAX_EDGE_CAP) {
    g_create_err = "edge_cap (max_sims * A by default) exceeds the 22-bit child-block pointer: at most " + std::to_string(MAX_EDGE_CAP);
    delete e;
    return AZ_EINVAL;
  }
  d.edge_cap = (int)want;
  if (d.edge_cap < A) d.edge_cap = A;
  d.depth_cap = (cfg->max_sims < d.max_plies ? cfg->max_sims : d.max_plies) + 1;
  e->nin = d.nn_channels * d.height * d.width;
  if ((hipSetDevice(cfg->device)) != hipSuccess) { g_create_err = "hipSetDevice failed"; delete e; return AZ_EHIP; }
  if (aztower_init_device(cfg->device) != AZ_OK) { g_create_err = "looking up the library's static device arrays failed"; delete e; return AZ_EHIP; }

  View &v = e->v;
  memset(&v, 0, sizeof(v));
  v.G = d.n_slots; v.E = d.edge_cap; v.D = d.depth_cap;
  v.K = cfg->leaves_per_step > 1 ? cfg->leaves_per_step : 1;
  v.win = win;
  v.hlen = hlen;
  v.vl = 0.3;  // VIRTUAL_LOSS, config/basic.py:25 (az_set_virtual_loss)
  d.leaves_per_step = v.K;
  v.c_puct = cfg->c_puct; v.epsilon = cfg->epsilon;
  const size_t G = (size_t)d.n_slots, E = (size_t)d.edge_cap, GK = G * (size_t)v.K;  // GK: rows of the per-simulation arrays
  int64_t total = 0;
  bool ok = true;
  auto alloc = [&](size_t bytes) -> void * {
    void *p = nullptr;
    if (!ok) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { ok = false; return nullptr; }
    e->allocs.push_back(p);
    total += (int64_t)bytes;
    return p;
  };
  v.edgeQ = (double *)alloc(G * E * sizeof(double));
  v.edgeNM = (int2 *)alloc(G * E * sizeof(int2));
  v.edgeP = (float *)alloc(G * E * sizeof(float));
  v.edgeAct = (uint8_t *)alloc(G * E);
  v.edgeCnt = (int32_t *)alloc(G * sizeof(int32_t));
  v.rootV = (int32_t *)alloc(G * sizeof(int32_t));
  v.rootN = (int32_t *)alloc(G * sizeof(int32_t));
  v.rootQ = (double *)alloc(G * sizeof(double));
  v.path = (int32_t *)alloc(GK * (size_t)d.depth_cap * sizeof(int32_t));
  v.leaf = (int4 *)alloc(GK * sizeof(int4));
  v.leafRing = (uint32_t *)alloc(GK * 8 * (size_t)d.state_words * sizeof(uint32_t));
  v.evalCount = (int32_t *)alloc(64);
  v.evalList = (int32_t *)alloc(GK * sizeof(int32_t));
  v.waveCnt = (int32_t *)alloc(G * sizeof(int32_t));
  v.remain = (unsigned long long *)alloc(2 * sizeof(unsigned long long));
  const size_t flag_bytes = (size_t)((GK + 16383) / 16384) * 16384;  // 1024 threads x a multiple of 16 flags
  v.evalFlag = (uint8_t *)alloc(flag_bytes);
  v.traj = (uint32_t *)alloc(G * (size_t)(d.max_plies + 1) * d.state_words * sizeof(uint32_t));
  v.trajPlayer = (int8_t *)alloc(G * (size_t)(d.max_plies + 1));
  v.trajVisits = (uint16_t *)alloc(G * (size_t)d.max_plies * A * sizeof(uint16_t));
  v.ply = (int32_t *)alloc(G * sizeof(int32_t));
  v.startPly = (int32_t *)alloc(G * sizeof(int32_t));
  v.player = (int32_t *)alloc(G * sizeof(int32_t));
  v.active = (uint8_t *)alloc(G);
  v.result = (float *)alloc(G * sizeof(float));
  v.stat = (unsigned long long *)alloc(G * NSTAT * sizeof(unsigned long long));
  v.err = (int32_t *)alloc(sizeof(int32_t));
  if (!ok) {
    g_create_err = "hipMalloc failed (engine needs " + std::to_string(total >> 20) + "+ MiB)";
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
    return AZ_ENOMEM;
  }
  d.device_bytes = total;
  // small state starts defined; the big pools need no clearing (bump allocated, written before read)
  if (hipMemset(v.stat, 0, G * NSTAT * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(v.err, 0, sizeof(int32_t)) != hipSuccess || hipMemset(v.evalCount, 0, 64) != hipSuccess || hipMemset(v.evalFlag, 0, flag_bytes) != hipSuccess || hipMemset(v.active, 0, G) != hipSuccess ||
      hipMemset(v.ply, 0, G * sizeof(int32_t)) != hipSuccess || hipMemset(v.leaf, 0, GK * sizeof(int4)) != hipSuccess || hipMemset(v.waveCnt, 0, G * sizeof(int32_t)) != hipSuccess ||
      hipMemset(v.trajPlayer, 0, G * (size_t)(d.max_plies + 1)) != hipSuccess) {
    g_create_err = "hipMemset failed";
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
    return AZ_EHIP;
  }
  *out = e;
  int rc = az_reset_games(e, nullptr, nullptr);
  if (rc == AZ_OK && hipDeviceSynchronize() != hipSuccess) rc = AZ_EHIP;
  if (rc != AZ_OK) { g_create_err = e->err; az_destroy(e); *out = nullptr; }
  return rc;
}

int az_destroy(az_engine *e) {
  if (!e) return AZ_EINVAL;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  for (void *p : e->allocs) (void)hipFree(p);
  delete e;
  return AZ_OK;
}

int az_get_dims(const az_engine *e, az_dims *out) {
  if (!e || !out) return AZ_EINVAL;
  *out = e->dims;
  return AZ_OK;
}

int az_check(az_engine *e, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  int32_t flag = 0;
  AZ_HIP(hipMemcpyAsync(&flag, e->v.err, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream));
  AZ_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (flag) {
    e->err = "device error flags:";
    if (flag & ERR_EDGE_CAP) e->err += " edge pool exhausted (raise edge_cap/max_sims)";
    if (flag & ERR_DEPTH_CAP) e->err += " path deeper than depth_cap";
    if (flag & ERR_ILLEGAL) e->err += " illegal move";
    if (flag & ERR_NO_ROOT) e->err += " root not expanded / no child to select";
    if (flag & ERR_EMIT_SETPOS) e->err += " az_emit_samples on a slot loaded by az_set_position (no trajectory before its root)";
    return AZ_EDEVICE;
  }
  return AZ_OK;
}

int az_reset_games(az_engine *e, const uint8_t *mask_dev, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  const int G = e->v.G;
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_reset_games<GT>), dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->v, mask_dev));
  return launch_check(e, "k_reset_games");
}

int az_set_position(az_engine *e, int32_t slot, const float *history_host, int32_t root_player, void *stream) {
  if (!e || !history_host) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  if (slot < 0 || slot >= e->v.G) { e->err = "slot out of range"; return AZ_EINVAL; }
  if (root_player != 1 && root_player != -1) { e->err = "root_player must be +1 or -1"; return AZ_EINVAL; }
  const az_dims &d = e->dims;
  const int numel = d.state_channels * d.height * d.width, SW = d.state_words;
  std::vector<uint32_t> words((size_t)8 * SW, 0u);
  int nm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int first = 8 - e->v.hlen;  // history_host holds the newest HISTORY_LEN states (the reference's get_history()), oldest first
  for (int k = first; k < 8; ++k) {
    bool ok = false;
    AZ_DISPATCH(e, ok = GT::encode(history_host + (size_t)(k - first) * numel, words.data() + (size_t)k * SW, &nm[k]));
    if (!ok) { e->err = "history holds a value outside {-1,0,1} or outside the board"; return AZ_EINVAL; }
  }
  const int ply = nm[7];
  if (ply >= d.max_plies) { e->err = "position has no move left"; return AZ_EINVAL; }
  {  // a root that is already won is terminal: the reference evaluates it by rule (core/MCTS_alphazero.py:157-158), there is no search
    bool won = false;
    AZ_DISPATCH(e, won = GT::host_won(words.data() + (size_t)7 * SW, e->cfg.win_length));
    if (won) { e->err = "position is already terminal (a side has won)"; return AZ_EINVAL; }
  }
  // entries older than the start of the game must be the initial (empty) state
  for (int k = first; k < 8; ++k) {
    const int t = ply - 7 + k;
    if (t < 0 && nm[k] != 0) { e->err = "history is longer than the game (non-empty state before ply 0)"; return AZ_EINVAL; }
  }
  hipStream_t s = (hipStream_t)stream;
  uint32_t *tb = e->v.traj + (size_t)slot * (d.max_plies + 1) * SW;
  for (int k = 0; k < 8; ++k) {  // (states older than the history the caller holds go in as empty boards: no output reads them)
    const int t = ply - 7 + k;
    if (t < 0) continue;
    AZ_HIP(hipMemcpyAsync(tb + (size_t)t * SW, words.data() + (size_t)k * SW, SW * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  }
  const int32_t one_ply = ply, one_player = root_player;
  const uint8_t act = 1;
  const int32_t zero = 0;
  const double zero_q = 0.0;
  AZ_HIP(hipMemcpyAsync(e->v.ply + slot, &one_ply, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.player + slot, &one_player, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.active + slot, &act, 1, hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.edgeCnt + slot, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.rootV + slot, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.rootN + slot, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.rootQ + slot, &zero_q, sizeof(double), hipMemcpyHostToDevice, s));
  AZ_HIP(hipMemcpyAsync(e->v.startPly + slot, &one_ply, sizeof(int32_t), hipMemcpyHostToDevice, s));
  AZ_HIP(hipStreamSynchronize(s));  // host staging buffers die here
  return AZ_OK;
}

int az_begin_search(az_engine *e, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  const int G = e->v.G;
  hipLaunchKernelGGL(k_begin_search, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->v);
  return launch_check(e, "k_begin_search");
}

int az_leaf_buffers(const az_engine *e, const void **ring_dev, const void **leaf_dev, const void **list_dev,
                    const void **count_dev, const void **flags_dev) {
  if (!e) return AZ_EINVAL;
  if (flags_dev) *flags_dev = e->v.evalFlag;
  if (ring_dev) *ring_dev = e->v.leafRing;
  if (leaf_dev) *leaf_dev = e->v.leaf;
  if (list_dev) *list_dev = e->v.evalList;
  if (count_dev) *count_dev = e->v.evalCount;
  return AZ_OK;
}

int az_select_gather(az_engine *e, void *nn_input_dev, void *stream) {
  if (!e) return AZ_EINVAL;
  if (e->v.K > 1) { e->err = "engine created with leaves_per_step > 1: use az_select_wave / az_expand_wave"; return AZ_EINVAL; }
  DeviceGuard guard(e->cfg.device);
  const dim3 grid((e->v.G + TREE_WPB - 1) / TREE_WPB), block(64 * TREE_WPB);
  hipStream_t s = (hipStream_t)stream;
  switch (e->cfg.nn_dtype) {
    case AZ_NN_F32: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_gather<GT, float>), grid, block, 0, s, e->v, (float *)nn_input_dev)); break;
    case AZ_NN_BF16: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_gather<GT, bf16_t>), grid, block, 0, s, e->v, (bf16_t *)nn_input_dev)); break;
    default: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_gather<GT, f16_t>), grid, block, 0, s, e->v, (f16_t *)nn_input_dev)); break;
  }
  return launch_check(e, "k_select_gather");
}

int az_compact_leaves(az_engine *e, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  hipLaunchKernelGGL(k_compact_leaves, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->v, ((e->v.G * e->v.K + 16383) / 16384) * 16);
  return launch_check(e, "k_compact_leaves");
}

int az_expand_backup(az_engine *e, const float *logits_dev, const float *values_dev, const float *noise_dev, void *stream) {
  if (!e || !logits_dev || !values_dev) return AZ_EINVAL;
  if (e->v.K > 1) { e->err = "engine created with leaves_per_step > 1: use az_select_wave / az_expand_wave"; return AZ_EINVAL; }
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_expand_backup<GT>), dim3((e->v.G + TREE_WPB - 1) / TREE_WPB), dim3(64 * TREE_WPB), 0, (hipStream_t)stream, e->v, logits_dev, values_dev, noise_dev));
  return launch_check(e, "k_expand_backup");
}

int az_expand_select(az_engine *e, const float *logits_dev, const float *values_dev, const float *noise_dev, void *nn_input_dev,
                     void *stream) {
  if (!e || !logits_dev || !values_dev) return AZ_EINVAL;
  if (e->v.K > 1) { e->err = "engine created with leaves_per_step > 1: use az_select_wave / az_expand_wave"; return AZ_EINVAL; }
  DeviceGuard guard(e->cfg.device);
  const dim3 grid((e->v.G + TREE_WPB - 1) / TREE_WPB), block(64 * TREE_WPB);
  hipStream_t s = (hipStream_t)stream;
  switch (e->cfg.nn_dtype) {
    case AZ_NN_F32: AZ_DISPATCH(e, hipLaunchKernelGGL((k_expand_select<GT, float>), grid, block, 0, s, e->v, logits_dev, values_dev, noise_dev, (float *)nn_input_dev)); break;
    case AZ_NN_BF16: AZ_DISPATCH(e, hipLaunchKernelGGL((k_expand_select<GT, bf16_t>), grid, block, 0, s, e->v, logits_dev, values_dev, noise_dev, (bf16_t *)nn_input_dev)); break;
    default: AZ_DISPATCH(e, hipLaunchKernelGGL((k_expand_select<GT, f16_t>), grid, block, 0, s, e->v, logits_dev, values_dev, noise_dev, (f16_t *)nn_input_dev)); break;
  }
  return launch_check(e, "k_expand_select");
}

int az_select_wave(az_engine *e, void *nn_input_dev, int32_t sims_target, void *stream) {
  if (!e || sims_target < 1) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  if (sims_target > e->cfg.max_sims) { e->err = "sims_target exceeds max_sims"; return AZ_EINVAL; }
  const dim3 grid(e->v.G), block(64);
  hipStream_t s = (hipStream_t)stream;
  switch (e->cfg.nn_dtype) {
    case AZ_NN_F32: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_wave<GT, float>), grid, block, 0, s, e->v, (float *)nn_input_dev, (int)sims_target)); break;
    case AZ_NN_BF16: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_wave<GT, bf16_t>), grid, block, 0, s, e->v, (bf16_t *)nn_input_dev, (int)sims_target)); break;
    default: AZ_DISPATCH(e, hipLaunchKernelGGL((k_select_wave<GT, f16_t>), grid, block, 0, s, e->v, (f16_t *)nn_input_dev, (int)sims_target)); break;
  }
  return launch_check(e, "k_select_wave");
}

int az_expand_wave(az_engine *e, const float *logits_dev, const float *values_dev, const float *noise_dev, void *stream) {
  if (!e || !logits_dev || !values_dev) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_expand_wave<GT>), dim3(e->v.G), dim3(64), 0, (hipStream_t)stream, e->v, logits_dev, values_dev, noise_dev));
  return launch_check(e, "k_expand_wave");
}

int az_search_remaining(az_engine *e, int32_t sims_target, int64_t *sum_out, int32_t *max_out, void *stream) {
  if (!e || sims_target < 1) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  hipStream_t s = (hipStream_t)stream;
  unsigned long long h[2] = {0, 0};
  AZ_HIP(hipMemsetAsync(e->v.remain, 0, sizeof(h), s));
  const int blocks = (e->v.G + 255) / 256 < 64 ? (e->v.G + 255) / 256 : 64;
  hipLaunchKernelGGL(k_search_remaining, dim3(blocks), dim3(256), 0, s, e->v, (int)sims_target);
  AZ_HIP(hipMemcpyAsync(h, e->v.remain, sizeof(h), hipMemcpyDeviceToHost, s));
  AZ_HIP(hipStreamSynchronize(s));
  if (sum_out) *sum_out = (int64_t)h[0];
  if (max_out) *max_out = (int32_t)h[1];
  return AZ_OK;
}

int az_set_virtual_loss(az_engine *e, double virtual_loss) {
  if (!e) return AZ_EINVAL;
  if (!(virtual_loss > 0.0)) { e->err = "VIRTUAL_LOSS must be > 0 (config/basic.py:46-47)"; return AZ_EINVAL; }
  e->v.vl = virtual_loss;
  return AZ_OK;
}

int az_root_policy(az_engine *e, float *pi_dev, int32_t *visits_dev, double *child_q_dev, float *root_prior_dev, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_root_policy<GT>), dim3(e->v.G), dim3(64), 0, (hipStream_t)stream, e->v, pi_dev, visits_dev, child_q_dev, root_prior_dev));
  return launch_check(e, "k_root_policy");
}

int az_root_nn_state(az_engine *e, float *nn_state_dev, void *stream) {
  if (!e || !nn_state_dev) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_root_nn_state<GT>), dim3(e->v.G), dim3(64), 0, (hipStream_t)stream, e->v, nn_state_dev));
  return launch_check(e, "k_root_nn_state");
}

int az_play_moves(az_engine *e, const float *inv_tao_dev, const float *expo_dev, const int32_t *forced_actions_dev,
                  int32_t *actions_dev, uint8_t *done_dev, void *stream) {
  if (!e || !inv_tao_dev || !expo_dev) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_play_moves<GT>), dim3(e->v.G), dim3(64), 0, (hipStream_t)stream, e->v, inv_tao_dev, expo_dev,
                                    forced_actions_dev, actions_dev, done_dev));
  return launch_check(e, "k_play_moves");
}

int az_game_info(az_engine *e, int32_t *plies_dev, int32_t *player_dev, uint8_t *active_dev, float *result_dev, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  const int G = e->v.G;
  hipLaunchKernelGGL(k_game_info, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->v, plies_dev, player_dev, active_dev, result_dev);
  return launch_check(e, "k_game_info");
}

int az_emit_samples(az_engine *e, const int32_t *slots_dev, const int32_t *offsets_dev, int32_t n_games, float *states_dev,
                    float *pis_dev, float *z_dev, void *stream) {
  if (!e || !slots_dev || !offsets_dev || !states_dev || !pis_dev || !z_dev || n_games < 0) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  if (n_games == 0) return AZ_OK;
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_emit_samples<GT>), dim3(n_games, e->dims.max_plies), dim3(64), 0, (hipStream_t)stream, e->v,
                                    slots_dev, offsets_dev, states_dev, pis_dev, z_dev));
  return launch_check(e, "k_emit_samples");
}

// Device -> pinned host copy by a handful of workgroups.  A D2H hipMemcpyAsync runs as a copy kernel on this stack
// (__amd_rocclr_copyBuffer) that fills the chip: measured here, the search kernels it overlaps with ran 2-30x slower
// for the 5-10 ms the samples of a move take.  The copy is PCIe bound, so 16 workgroups keep the link busy and leave
// 240 CUs to the search.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__global__ __launch_bounds__(256) void k_copy_to_host(const u32x4_t *__restrict__ src, u32x4_t *__restrict__ dst, size_t n16,
                                                      const uint8_t *__restrict__ src_tail, uint8_t *__restrict__ dst_tail, int tail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

int az_copy_to_host(void *dst_pinned_host, const void *src_dev, uint64_t bytes, int32_t n_workgroups, void *stream) {
  if (!dst_pinned_host || !src_dev || n_workgroups < 1 || n_workgroups > 1024) return AZ_EINVAL;
  if (((uintptr_t)dst_pinned_host | (uintptr_t)src_dev) & 15u) return AZ_EINVAL;
  if (bytes == 0) return AZ_OK;
  void *dst_dev = nullptr;  // the device's address of the pinned allocation (the same value under unified addressing)
  if (hipHostGetDevicePointer(&dst_dev, dst_pinned_host, 0) != hipSuccess || dst_dev == nullptr) { (void)hipGetLastError(); return AZ_EINVAL; }
  const size_t n16 = bytes / 16;
  const int tail = (int)(bytes % 16);
  hipLaunchKernelGGL(k_copy_to_host, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4_t *)src_dev, (u32x4_t *)dst_dev, n16,
                     (const uint8_t *)src_dev + n16 * 16, (uint8_t *)dst_dev + n16 * 16, tail);
  return hipGetLastError() == hipSuccess ? AZ_OK : AZ_EHIP;
}

static void dab_rot(int &r, int &c, int k, int n) {  // examples/dots_and_boxes/game.py:305-313 on an n x n grid
  for (int q = 0; q < (k & 3); ++q) { const int nr = n - 1 - c, nc = r; r = nr; c = nc; }
}
static AugTables make_aug_tables(int B) {  // B x B boxes, D x D dots
  AugTables tb;
  memset(&tb, 0, sizeof(tb));
  const int D = B + 1, NH = D * B;
  for (int t = 0; t < 8; ++t) {
    const int k = t >> 1;
    const bool flip = t & 1;
    for (int a = 0; a < 2 * NH; ++a) {  // _transform_edge :261-292
      const bool hz = a < NH;
      int r1 = hz ? a / B : (a - NH) / D, c1 = hz ? a % B : (a - NH) % D;
      int r2 = hz ? r1 : r1 + 1, c2 = hz ? c1 + 1 : c1;
      dab_rot(r1, c1, k, D);
      dab_rot(r2, c2, k, D);
      if (flip) { int x = r1; r1 = c1; c1 = x; x = r2; r2 = c2; c2 = x; }
      const int na = (r1 == r2) ? r1 * B + (c1 < c2 ? c1 : c2) : NH + (r1 < r2 ? r1 : r2) * D + c1;
      tb.invEdge[t][na] = (uint8_t)a;
    }
    for (int b = 0; b < B * B; ++b) {  // _transform_box :294-303
      int r = b / B, c = b % B;
      dab_rot(r, c, k, B);
      if (flip) { const int x = r; r = c; c = x; }
      tb.invBox[t][r * B + c] = (uint8_t)b;
    }
  }
  return tb;
}

int az_augment_samples(az_engine *e, int32_t n, const float *states_dev, const float *pis_dev, const float *z_dev,
                       float *out_states_dev, float *out_pis_dev, float *out_z_dev, int64_t ring_capacity,
                       int64_t ring_write_idx, void *stream) {
  if (!e || !states_dev || !pis_dev || !z_dev || !out_states_dev || !out_pis_dev || !out_z_dev || n < 0) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  if (ring_capacity < 0 || ring_write_idx < 0 || (ring_capacity > 0 && ring_capacity < 8)) { e->err = "bad ring arguments"; return AZ_EINVAL; }
  if (n == 0) return AZ_OK;
  if (e->cfg.game == AZ_GAME_DAB && e->cfg.board_size != e->cfg.board_cols) {  // examples/dots_and_boxes/game.py:318-319 raises too
    e->err = "az_augment_samples: the dots-and-boxes symmetries need a square board (BOX_ROWS == BOX_COLS)"; return AZ_EINVAL;
  }
  static const AugTables tbs[4] = {make_aug_tables(2), make_aug_tables(3), make_aug_tables(4), make_aug_tables(5)};
  const AugTables &tb = tbs[e->cfg.game == AZ_GAME_DAB ? e->cfg.board_size - 2 : 1];
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_augment<GT>), dim3(n, 8), dim3(64), 0, (hipStream_t)stream, tb, states_dev, pis_dev, z_dev,
                                    out_states_dev, out_pis_dev, out_z_dev, (long long)ring_capacity, (long long)ring_write_idx, e->v.hlen));
  return launch_check(e, "k_augment");
}

int az_draw_rng(az_engine *e, uint64_t seed, const int64_t *game_ids_dev, float alpha, float *noise_dev, float *expo_dev, void *stream) {
  if (!e || !game_ids_dev || (!noise_dev && !expo_dev)) return AZ_EINVAL;
  if (noise_dev && !(alpha > 0.0f && alpha <= 1.0f)) { e->err = "az_draw_rng: the Dirichlet sampler takes 0 < alpha <= 1 (ALPHA = 0.3, config/basic.py:13)"; return AZ_EINVAL; }
  DeviceGuard guard(e->cfg.device);
  AZ_DISPATCH(e, hipLaunchKernelGGL((k_draw_rng<GT>), dim3(e->v.G), dim3(64), 0, (hipStream_t)stream, e->v, (unsigned long long)seed,
                                    (const long long *)game_ids_dev, alpha, noise_dev, expo_dev));
  return launch_check(e, "k_draw_rng");
}

int az_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  if (!counter || !key || !out) return AZ_EINVAL;
  const az::Philox r = az::philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = r.c[i];
  return AZ_OK;
}

int az_read_stats(az_engine *e, az_stats *out, void *stream) {
  if (!e || !out) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  const size_t n = (size_t)e->v.G * NSTAT;
  std::vector<unsigned long long> h(n);
  AZ_HIP(hipMemcpyAsync(h.data(), e->v.stat, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
  AZ_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long acc[NSTAT] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t g = 0; g < (size_t)e->v.G; ++g)
    for (int k = 0; k < NSTAT; ++k) acc[k] += h[g * NSTAT + k];
  out->sims = acc[0]; out->sum_depth = acc[1]; out->sum_children = acc[2];
  out->evals = acc[3]; out->terminals = acc[4]; out->expanded_edges = acc[5]; out->sum_touched = acc[6];
  return AZ_OK;
}

int az_reset_stats(az_engine *e, void *stream) {
  if (!e) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  AZ_HIP(hipMemsetAsync(e->v.stat, 0, (size_t)e->v.G * NSTAT * sizeof(unsigned long long), (hipStream_t)stream));
  return AZ_OK;
}

int az_eval_hash(az_engine *e, const void *nn_input_dev, float scale, float *logits_dev, float *values_dev, void *stream) {
  if (!e || !logits_dev || !values_dev) return AZ_EINVAL;
  DeviceGuard guard(e->cfg.device);
  const dim3 grid(e->v.G * e->v.K), block(64);  // one row per (slot, leaf of the step)
  hipStream_t s = (hipStream_t)stream;
  const int A = e->dims.n_action;
  if (!nn_input_dev) {  // read the packed leaf batch of the last descent
    AZ_DISPATCH(e, hipLaunchKernelGGL((k_eval_hash_packed<GT>), grid, block, 0, s, e->v, scale, logits_dev, values_dev));
    return launch_check(e, "k_eval_hash_packed");
  }
  switch (e->cfg.nn_dtype) {
    case AZ_NN_F32: hipLaunchKernelGGL((k_eval_hash<float>), grid, block, 0, s, e->nin, A, (const float *)nn_input_dev, scale, logits_dev, values_dev); break;
    case AZ_NN_BF16: hipLaunchKernelGGL((k_eval_hash<bf16_t>), grid, block, 0, s, e->nin, A, (const bf16_t *)nn_input_dev, scale, logits_dev, values_dev); break;
    default: hipLaunchKernelGGL((k_eval_hash<f16_t>), grid, block, 0, s, e->nin, A, (const f16_t *)nn_input_dev, scale, logits_dev, values_dev); break;
  }
  return launch_check(e, "k_eval_hash");
}

}  // extern "C"
