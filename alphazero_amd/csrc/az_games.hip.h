// az_games.hip.h -- bitboard rules of the three games, for one wavefront per position.
//
// A state is SW u32 words kept in LDS, private to ONE wavefront.  Every function is called by all 64 lanes of the wave
// with wave-uniform arguments unless it says "per lane".  Synchronisation is wave-level only (wave_sync: no s_barrier),
// so several waves of one workgroup can each run a position of their own, at their own pace.
//
// Reference semantics (files under the reference tree):
//   tictactoe  examples/tictactoe/game.py:20-77
//   dab        examples/dots_and_boxes/game.py:36-156   (BOX_ROWS x BOX_COLS boxes, 2 .. 5 each; 3 x 3 as shipped)
//   gomoku     examples/gomoku/game.py:24-104
//   terminal value  games/base.py:104-114
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace az {

// terminal() results
enum : int { T_NONE = 0, T_WIN_P1 = 1, T_WIN_P2 = 2, T_NO_WINNER = 3 };

// Order this wave's own LDS / global accesses (lane 0 writes, all lanes read): a workgroup-scope fence for the memory
// model and a wave barrier for the compiler -- NOT an s_barrier.  One wavefront executes in lock step and its LDS and
// vector-memory instructions are issued in program order, so nothing else is needed between lanes of one wave; and
// because no workgroup barrier is involved, the waves of a workgroup may take different numbers of steps.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t bit_at(const uint32_t *w, int i) { return (w[i >> 5] >> (i & 31)) & 1u; }

// ------------------------------------------------------------------------------------------
// TicTacToe: word0 bits 0-8 = PLAYER1 stones (cell = row*3+col), bits 9-17 = PLAYER2 stones.
struct TicTacToe {
  static constexpr int A = 9, SW = 1, CS = 1, H = 3, WD = 3, HW = 9, CIN = 9, PMAX = 9;

  __device__ static bool legal(const uint32_t *s, int a) {  // per lane
    return (((s[0] | (s[0] >> 9)) >> a) & 1u) == 0u;
  }
  __device__ static bool same_player(const uint32_t *, int) { return false; }  // per lane
  __device__ static int apply(uint32_t *s, int a, int player, int lane) {
    uint32_t w = s[0] | (1u << (a + (player == 1 ? 0 : 9)));
    wave_sync();
    if (lane == 0) s[0] = w;
    wave_sync();
    return -player;
  }
  // after `mover` played `a`; nmoves = stones on the board
  __device__ static int terminal(const uint32_t *s, int, int mover, int nmoves, int, int /*win*/) {
    const uint32_t b = (mover == 1 ? s[0] : (s[0] >> 9)) & 0x1FFu;
    const uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    bool win = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) win |= (b & L[i]) == L[i];
    if (win) return mover == 1 ? T_WIN_P1 : T_WIN_P2;
    return nmoves == 9 ? T_NO_WINNER : T_NONE;
  }
  __device__ static float plane(const uint32_t *s, int, int cell) {  // per lane
    return (float)(int)((s[0] >> cell) & 1u) - (float)(int)((s[0] >> (9 + cell)) & 1u);
  }
  // host: f32 state (3,3) -> words; returns false on a value outside {-1,0,1}
  __host__ static bool encode(const float *st, uint32_t *w, int *nmoves) {
    w[0] = 0; *nmoves = 0;
    for (int i = 0; i < 9; ++i) {
      if (st[i] == 1.0f) { w[0] |= 1u << i; ++*nmoves; }
      else if (st[i] == -1.0f) { w[0] |= 1u << (9 + i); ++*nmoves; }
      else if (st[i] != 0.0f) return false;
    }
    return true;
  }
  // host: has either side already completed a line?  (az_set_position refuses such roots: the reference evaluates them by rule)
  __host__ static bool host_won(const uint32_t *w, int /*win*/ = 0) {
    const uint32_t L[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x111u, 0x054u};
    for (int i = 0; i < 8; ++i)
      if ((w[0] & L[i]) == L[i] || ((w[0] >> 9) & L[i]) == L[i]) return true;
    return false;
  }
};

// ------------------------------------------------------------------------------------------
// Dots and boxes with R x C boxes (examples/dots_and_boxes/game.py:24-25 BOX_ROWS, BOX_COLS; 3 x 3 in the reference as
// shipped, any 2 .. 5 here): words [0,EW) bit a = edge of action a (a < NH horizontal r=a/C, c=a%C; else vertical
// idx=a-NH, r=idx/(C+1), c=idx%(C+1)); then the owners: bits [0,B) boxes of PLAYER1 (box = r*C+c), bits [OB,OB+B) PLAYER2
// (OB = B in one word while 2B <= 32, else a word each).  3 x 3: word0 = 24 edges, word1 = 9 + 9 owner bits.
template <int R, int C>
struct Dab {
  static constexpr int BR = R, BC = C, NH = (R + 1) * C, NV = R * (C + 1), A = NH + NV, B = R * C;
  static constexpr int EW = (A + 31) / 32, OW = (2 * B <= 32) ? 1 : 2, OB = (2 * B <= 32) ? B : 32, SW = EW + OW;
  static constexpr int CS = 3, H = R + 1, WD = C + 1, HW = H * WD, CIN = 25, PMAX = A;
  static constexpr uint64_t ALL = (A == 64) ? ~0ull : ((1ull << A) - 1ull);
  static constexpr uint32_t BM = (1u << B) - 1u;

  __host__ __device__ static uint64_t edges(const uint32_t *s) { return EW == 1 ? (uint64_t)s[0] : ((uint64_t)s[0] | ((uint64_t)s[EW - 1] << 32)); }
  __host__ __device__ static uint32_t own1(const uint32_t *s) { return s[EW] & BM; }
  __host__ __device__ static uint32_t own2(const uint32_t *s) { return (OW == 1 ? (s[EW] >> OB) : s[EW + OW - 1]) & BM; }
  __host__ __device__ static uint64_t box_mask(int b) {
    const int br = b / C, bc = b % C;
    return (1ull << (br * C + bc)) | (1ull << ((br + 1) * C + bc)) | (1ull << (NH + br * (C + 1) + bc)) |
           (1ull << (NH + br * (C + 1) + bc + 1));
  }
  // the (up to) two boxes an edge borders; -1 where there is none  (game.py:126-156)
  __host__ __device__ static void adjacent(int a, int &b1, int &b2) {
    if (a < NH) {
      const int r = a / C, c = a % C;
      b1 = r > 0 ? (r - 1) * C + c : -1;
      b2 = r < R ? r * C + c : -1;
    } else {
      const int idx = a - NH, r = idx / (C + 1), c = idx % (C + 1);
      b1 = c > 0 ? r * C + c - 1 : -1;
      b2 = c < C ? r * C + c : -1;
    }
  }
  // boxes that playing `a` completes (bit b), given edges e and the owned boxes of both sides
  __host__ __device__ static uint32_t completed(uint64_t e, uint32_t owned, int a) {
    int b1, b2;
    adjacent(a, b1, b2);
    const uint64_t e2 = e | (1ull << a);
    uint32_t done = 0;
    if (b1 >= 0 && !((owned >> b1) & 1u) && (e2 & box_mask(b1)) == box_mask(b1)) done |= 1u << b1;
    if (b2 >= 0 && !((owned >> b2) & 1u) && (e2 & box_mask(b2)) == box_mask(b2)) done |= 1u << b2;
    return done;
  }
  __device__ static bool legal(const uint32_t *s, int a) { return bit_at(s, a) == 0u; }
  __device__ static bool same_player(const uint32_t *s, int a) { return completed(edges(s), own1(s) | own2(s), a) != 0u; }
  __device__ static int apply(uint32_t *s, int a, int player, int lane) {
    const uint32_t done = completed(edges(s), own1(s) | own2(s), a);
    const int ei = a >> 5, oi = EW + ((OW == 2 && player != 1) ? 1 : 0);
    const uint32_t ew = s[ei] | (1u << (a & 31));
    const uint32_t ow = s[oi] | (done << ((OW == 1 && player != 1) ? OB : 0));
    wave_sync();
    if (lane == 0) {
      s[ei] = ew;
      s[oi] = ow;
    }
    wave_sync();
    return done ? player : -player;
  }
  __device__ static int terminal(const uint32_t *s, int, int, int, int, int /*win*/) {
    if ((edges(s) & ALL) != ALL) return T_NONE;
    const int p1 = __popc(own1(s)), p2 = __popc(own2(s));
    return p1 > p2 ? T_WIN_P1 : (p2 > p1 ? T_WIN_P2 : T_NO_WINNER);
  }
  __device__ static float plane(const uint32_t *s, int ch, int cell) {
    const int r = cell / WD, c = cell % WD;
    if (ch == 0) return (c < C) ? (float)(int)bit_at(s, r * C + c) : 0.0f;
    if (ch == 1) return (r < R) ? (float)(int)bit_at(s, NH + r * (C + 1) + c) : 0.0f;
    if (r < R && c < C) return (float)(int)((own1(s) >> (r * C + c)) & 1u) - (float)(int)((own2(s) >> (r * C + c)) & 1u);
    return 0.0f;
  }
  __host__ static bool encode(const float *st, uint32_t *w, int *nmoves) {
    for (int i = 0; i < SW; ++i) w[i] = 0;
    *nmoves = 0;
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < WD; ++c) {
        const float h = st[r * WD + c], v = st[HW + r * WD + c], o = st[2 * HW + r * WD + c];
        if (c < C) { if (h == 1.0f) { const int a = r * C + c; w[a >> 5] |= 1u << (a & 31); ++*nmoves; } else if (h != 0.0f) return false; }
        else if (h != 0.0f) return false;
        if (r < R) { if (v == 1.0f) { const int a = NH + r * (C + 1) + c; w[a >> 5] |= 1u << (a & 31); ++*nmoves; } else if (v != 0.0f) return false; }
        else if (v != 0.0f) return false;
        if (r < R && c < C) {
          if (o == 1.0f) w[EW] |= 1u << (r * C + c);
          else if (o == -1.0f) { if (OW == 1) w[EW] |= 1u << (OB + r * C + c); else w[EW + 1] |= 1u << (r * C + c); }
          else if (o != 0.0f) return false;
        } else if (o != 0.0f) return false;
      }
    return true;
  }
  __host__ static bool host_won(const uint32_t *, int /*win*/ = 0) { return false; }  // terminal only when every edge is drawn (ply == PMAX)
};

// ------------------------------------------------------------------------------------------
// Gomoku NxN: words [0,WC) bit a = PLAYER1 stone at cell a = row*N+col; words [WC,2WC) PLAYER2.
template <int N>
struct Gomoku {
  static constexpr int A = N * N, WC = (A + 31) / 32, SW = 2 * WC, CS = 1, H = N, WD = N, HW = A,
                       CIN = 9, PMAX = A;

  __device__ static bool legal(const uint32_t *s, int a) { return (bit_at(s, a) | bit_at(s + WC, a)) == 0u; }
  __device__ static bool same_player(const uint32_t *, int) { return false; }
  __device__ static int apply(uint32_t *s, int a, int player, int lane) {
    const int wi = (player == 1 ? 0 : WC) + (a >> 5);
    const uint32_t w = s[wi] | (1u << (a & 31));
    wave_sync();
    if (lane == 0) s[wi] = w;
    wave_sync();
    return -player;
  }
  // Only the mover can have made a line (the position before was not terminal), so the row-major
  // scan of the reference (game.py:62-89) reduces to the four lines through the new stone.
  // K = WIN_LENGTH (game.py:22; 3 .. 6, five by default).  Lane l < 4 (2K - 1) looks at direction l / (2K - 1), offset
  // l % (2K - 1) - (K - 1); any K consecutive set bits in a (2K - 1)-window win (longer rows count, as in the reference).
  template <int K>
  __device__ static int terminal_k(const uint32_t *s, int a, int mover, int nmoves, int lane) {
    constexpr int span = 2 * K - 1;
    const uint32_t *mine = s + (mover == 1 ? 0 : WC);
    const int r = a / N, c = a % N;
    bool bit = false;
    if (lane < 4 * span) {
      const int d = lane / span, k = lane % span - (K - 1);
      const int dr = (d == 0) ? 0 : 1, dc = (d == 0) ? 1 : (d == 1 ? 0 : (d == 2 ? 1 : -1));
      const int rr = r + k * dr, cc = c + k * dc;
      if (rr >= 0 && rr < N && cc >= 0 && cc < N) bit = bit_at(mine, rr * N + cc) != 0u;
    }
    const unsigned long long m = __ballot(bit);
    bool win = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t w = (uint32_t)(m >> (span * d)) & ((1u << span) - 1u);
      uint32_t run = w;
#pragma unroll
      for (int i = 1; i < K; ++i) run &= w >> i;  // bit j of run: stones j .. j + K - 1 of the window are all set
      win |= (run != 0u);
    }
    if (win) return mover == 1 ? T_WIN_P1 : T_WIN_P2;
    return nmoves == A ? T_NO_WINNER : T_NONE;
  }
  // (K is wave-uniform; each value gets the code of a compile-time constant -- as a run-time loop the check cost the tree
  // kernel 3 of its 36 us)
  __device__ static int terminal(const uint32_t *s, int a, int mover, int nmoves, int lane, int K) {
    switch (K) {
      case 3: return terminal_k<3>(s, a, mover, nmoves, lane);
      case 4: return terminal_k<4>(s, a, mover, nmoves, lane);
      case 6: return terminal_k<6>(s, a, mover, nmoves, lane);
      default: return terminal_k<5>(s, a, mover, nmoves, lane);
    }
  }
  __device__ static float plane(const uint32_t *s, int, int cell) {
    return (float)(int)bit_at(s, cell) - (float)(int)bit_at(s + WC, cell);
  }
  __host__ static bool encode(const float *st, uint32_t *w, int *nmoves) {
    for (int i = 0; i < SW; ++i) w[i] = 0;
    *nmoves = 0;
    for (int i = 0; i < A; ++i) {
      if (st[i] == 1.0f) { w[i >> 5] |= 1u << (i & 31); ++*nmoves; }
      else if (st[i] == -1.0f) { w[WC + (i >> 5)] |= 1u << (i & 31); ++*nmoves; }
      else if (st[i] != 0.0f) return false;
    }
    return true;
  }
  __host__ static bool host_won(const uint32_t *w, int K = 5) {  // K (or more) in a row of either colour: game.py:62-89
    const int DR[4] = {0, 1, 1, 1}, DC[4] = {1, 0, 1, -1};
    for (int side = 0; side < 2; ++side) {
      const uint32_t *b = w + side * WC;
      for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c)
          for (int d = 0; d < 4; ++d) {
            int k = 0;
            for (; k < K; ++k) {
              const int rr = r + k * DR[d], cc = c + k * DC[d];
              if (rr < 0 || rr >= N || cc < 0 || cc >= N) break;
              const int i = rr * N + cc;
              if (!((b[i >> 5] >> (i & 31)) & 1u)) break;
            }
            if (k == K) return true;
          }
    }
    return false;
  }
};

// games/base.py:104-114 from the point of view of `player` (the player to move at that node)
__device__ __forceinline__ float terminal_value(int term, int player) {
  if (term == T_WIN_P1) return player == 1 ? 1.0f : -1.0f;
  if (term == T_WIN_P2) return player == -1 ? 1.0f : -1.0f;
  return 0.0f;
}

}  // namespace az
