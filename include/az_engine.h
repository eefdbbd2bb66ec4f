/*
 * az_engine.h -- C ABI of the MI355X (gfx950) many-tree self-play engine.
 *
 * The reference (BlueRose2020/alphazero) is pure Python and has no FFI of its own; the boundary it
 * does have is three duck-typed Python contracts.  This header is the native
 * layer underneath them: each entry point replaces the interpreted code cited next to it, for
 * thousands of games at once.  Host code (alphazero_amd/engine.py) binds it with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success and a negative AZ_E* code on failure; no exceptions cross
 *     the ABI; az_last_error() returns a message for the last failure on that engine;
 *   - pointers named *_dev are device pointers owned by the CALLER (e.g. torch tensors) and only
 *     borrowed for the call; tree and game arrays are owned by the engine;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous
 *     on that stream unless stated otherwise; one host thread per engine;
 *   - one engine lives on one GPU (az_config.device); games never move between engines.
 *   - there is no CPU fallback: az_create fails if no HIP device is usable.
 */
#ifndef AZ_ENGINE_H
#define AZ_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GAME_TICTACTOE 0 /* examples/tictactoe/game.py   S=(3,3)    A=9   */
#define AZ_GAME_DAB 1       /* examples/dots_and_boxes/game.py S=(3,4,4) A=24 (3x3 boxes) */
#define AZ_GAME_GOMOKU 2    /* examples/gomoku/game.py      S=(N,N)    A=N*N, N in {9,15} */

#define AZ_NN_F32 0
#define AZ_NN_BF16 1
#define AZ_NN_F16 2

#define AZ_OK 0
#define AZ_EINVAL (-1)   /* bad argument / unsupported configuration */
#define AZ_ENOMEM (-2)   /* device allocation failed */
#define AZ_EHIP (-3)     /* a HIP call failed (message in az_last_error) */
#define AZ_EDEVICE (-4)  /* a kernel flagged an internal error (capacity, illegal move) */

typedef struct az_engine az_engine;

typedef struct az_config {
  int32_t game;        /* AZ_GAME_* */
  int32_t board_size;  /* gomoku N = examples/gomoku/game.py:21 BOARD_SIZE, 3 .. 15; dots-and-boxes: BOX_ROWS
                          (examples/dots_and_boxes/game.py:24), 2 .. 5, 0 = the reference's 3; ignored for tic-tac-toe */
  int32_t n_slots;     /* concurrent games (one tree each) */
  int32_t max_sims;    /* simulations per move the trees are sized for (TRAIN_NUM_SIMULATION) */
  float c_puct;        /* config/basic.py:12 */
  float epsilon;       /* config/basic.py:14, Dirichlet mix weight */
  int32_t nn_dtype;    /* AZ_NN_*: element type of the NN-input batch the engine writes */
  int32_t device;      /* HIP device ordinal */
  int32_t edge_cap;    /* child records per tree; 0 = worst case max_sims * A */
  int32_t leaves_per_step; /* 0 or 1: the reference's sequential search (one leaf per tree per step); 2 .. 16: virtual loss /
                              in-tree leaf batching (INFER_BATCH_SIZE, config/basic.py:28) through az_select_wave /
                              az_expand_wave -- the per-simulation batch then has n_slots * leaves_per_step rows */
  int32_t win_length;  /* gomoku: stones in a row that win = examples/gomoku/game.py:22 WIN_LENGTH, 3 .. 6 and <= board_size;
                          0 = the reference's 5; ignored otherwise */
  int32_t board_cols;  /* dots-and-boxes: BOX_COLS (examples/dots_and_boxes/game.py:25), 2 .. 5, 0 = the same as board_size;
                          ignored otherwise */
  int32_t history_len; /* states in the network input = config/basic.py:10 _HISTORY_LEN, 1 .. 8; 0 = the reference's 8.  The
                          dense batch, az_root_nn_state and the emitted samples then have history_len * C + 1 planes and
                          az_set_position reads history_len states; the packed leaf batch always carries 8 states (the
                          hand-written networks read it and exist for 8 only). */
} az_config;

typedef struct az_dims {
  int32_t n_action;       /* A */
  int32_t state_channels; /* channels of one state (1 for 2-D states) */
  int32_t height, width;
  int32_t nn_channels;    /* C_in = history_len * state_channels + 1 (USE_HISTORY; HISTORY_LEN = 8 as shipped) */
  int32_t max_plies;
  int32_t state_words;    /* u32 words of one bitboard state */
  int32_t edge_cap, node_cap, depth_cap;
  int32_t n_slots;
  int32_t leaves_per_step; /* K >= 1: rows per slot in the NN batch, logits, values and the packed leaf batch */
  int64_t device_bytes;   /* HBM held by the engine */
} az_dims;

/* counters summed over all slots since az_create (or the last az_reset_stats) */
typedef struct az_stats {
  uint64_t sims;          /* simulations run */
  uint64_t sum_depth;     /* sum of selection depths (edges followed) */
  uint64_t sum_children;  /* child records scanned by select (light part: prior 4 B + action 1 B each) */
  uint64_t evals;         /* leaves that needed the network */
  uint64_t terminals;     /* leaves that were terminal positions */
  uint64_t expanded_edges;/* child records written by expand */
  uint64_t sum_touched;   /* of sum_children, the records already touched, whose heavy part (visits, meta, value: 16 B) select also reads */
} az_stats;

/* Lifetime.  az_create sizes and allocates every array once; nothing else allocates. */
int az_create(const az_config *cfg, az_engine **out);
int az_destroy(az_engine *e);
int az_get_dims(const az_engine *e, az_dims *out);
const char *az_last_error(const az_engine *e); /* e == NULL: last az_create failure */
int az_check(az_engine *e, void *stream);      /* synchronises; AZ_EDEVICE if a kernel flagged an error */

/* games/base.py:15-20  BaseGame.reset(): initial state, PLAYER1 to move, history = 8x initial.
 * mask_dev: n_slots bytes, non-zero = reset that slot; NULL = all slots. */
int az_reset_games(az_engine *e, const uint8_t *mask_dev, void *stream);

/* Load an arbitrary position into one slot (the MCTS.search(root_state, root_player,
 * history_state) arguments, core/MCTS_alphazero.py:128-137).  history_host: history_len (8) states of
 * state_channels*height*width f32 each, oldest first, newest == root state (utils/history_manager.py:41-47).
 * Host pointer; synchronous.  A root that is already decided (a side has won, or no move is left) is refused with
 * AZ_EINVAL: the reference evaluates such a node by rule (:157-158), there is nothing to search.  The slot gets a
 * fresh root (no tree, zero root statistics); it may be searched and played on, but it has no trajectory before the
 * root, so az_emit_samples refuses it (device error flag, AZ_EDEVICE from az_check) unless the root is the initial
 * position. */
int az_set_position(az_engine *e, int32_t slot, const float *history_host, int32_t root_player, void *stream);

/* core/MCTS_alphazero.py:142  a fresh, unexpanded root for every active slot (no tree reuse). */
int az_begin_search(az_engine *e, void *stream);

/* One simulation's descent in every active tree: core/MCTS_alphazero.py:145-160 (history load, the
 * select loop :149-154 = MCTSNode.select :59-78 + game next_state/is_terminal, terminal evaluation
 * :157-158) and _node2nn_state :202-209.  Writes row g of nn_input_dev, shape
 * (n_slots, nn_channels, height, width) of az_config.nn_dtype, for every slot whose leaf needs the
 * network; other rows are left untouched.  The same batch is always kept in packed form (8 bitboard
 * states per leaf, see az_leaf_buffers); nn_input_dev may be NULL when only that form is consumed. */
int az_select_gather(az_engine *e, void *nn_input_dev, void *stream);

/* The packed leaf batch of the last az_select_gather, engine-owned, read-only for the caller:
 * ring_dev -> u32 (n_slots, 8, state_words): each leaf's 8-state history as bitboards, oldest first;
 * leaf_dev -> int32 x4 per slot {depth, kind (0 = needs the network, 1 = terminal), value bits, player};
 * list_dev -> i32 (n_slots): the slots whose leaf needs the network, ascending, and count_dev -> i32: how many
 * (terminal leaves, which the reference re-evaluates by rule on every revisit, are absent); both are filled by
 * az_compact_leaves, to be called after az_select_gather when they are wanted;
 * flags_dev -> u8 (n_slots): 1 where the slot's leaf needs the network (always filled; what az_tower_forward scans
 * when it is given no list).
 * Consumed by az_tower_forward; ~8x fewer bytes than the dense planes.  Any out pointer may be NULL. */
int az_leaf_buffers(const az_engine *e, const void **ring_dev, const void **leaf_dev, const void **list_dev,
                    const void **count_dev, const void **flags_dev);
int az_compact_leaves(az_engine *e, void *stream);

/* The rest of that simulation: _get_prior :180-192 (softmax, Dirichlet mix at the root, legal mask,
 * renormalise), MCTSNode.expand :43-57, value.item() :170 and _bacaward :194-200.
 * logits_dev (n_slots, A) f32, values_dev (n_slots) f32: the model's outputs for the batch written by
 * az_select_gather.  noise_dev (n_slots, A) f32: one Dirichlet(alpha) draw per slot, applied when the
 * leaf is the root; NULL = use_Dirichlet False. */
int az_expand_backup(az_engine *e, const float *logits_dev, const float *values_dev,
                     const float *noise_dev, void *stream);

/* az_expand_backup for the running simulation followed by az_select_gather for the next one, of every tree, in ONE
 * launch: the search loop (MCTS.search :144-172) alternates network and tree work, so all but the first descent and the
 * last backup of a move can go out this way (half the tree launches).  Same arguments and results as the two calls. */
int az_expand_select(az_engine *e, const float *logits_dev, const float *values_dev, const float *noise_dev,
                     void *nn_input_dev, void *stream);

/* Virtual loss / in-tree leaf batching, for engines created with leaves_per_step K > 1.  The reference carries the
 * hooks -- MCTSNode.pending / .virtual_loss (core/MCTS_alphazero.py:32-33), `q - child.virtual_loss / (child.visits + 1)`
 * in select (:70), USE_VIRTUAL_LOSS / VIRTUAL_LOSS = 0.3 / INFER_BATCH_SIZE (config/basic.py:19-30) -- but no code that
 * sets them, so the rule is defined here (and restated in oracle/az_oracle.c: az_oracle_search_wave):
 *   az_select_wave: every active tree makes up to min(K, sims_target - simulations done) descents, one after the other;
 *   a child's virtual loss is VIRTUAL_LOSS x (descents already queued in this call that pass through it); tree
 *   statistics do not change inside the call; a descent that ends on a leaf already queued closes the tree's wave and is
 *   dropped.  Leaf j of slot g is row g*K + j of nn_input_dev ((n_slots*K, nn_channels, height, width), may be NULL) and
 *   of the packed leaf batch / flags (az_leaf_buffers; rows of unused j have flag 0).
 *   az_expand_wave: logits_dev (n_slots*K, A), values_dev (n_slots*K): every tree's queued leaves are expanded (or, if
 *   terminal, valued by rule) and backed up in queue order; noise_dev (n_slots, A) or NULL as for az_expand_backup.
 *   az_search_remaining: synchronises; *sum_out = simulations still to run over all active trees for a search of
 *   sims_target simulations, *max_out = the most any one tree has left (either may be NULL).  Loop select_wave ->
 *   network -> expand_wave until it reports 0.
 * With K == 1 the pair is az_select_gather / az_expand_backup bit for bit.  The sequential entry points are refused on a
 * K > 1 engine. */
int az_select_wave(az_engine *e, void *nn_input_dev, int32_t sims_target, void *stream);
int az_expand_wave(az_engine *e, const float *logits_dev, const float *values_dev, const float *noise_dev, void *stream);
int az_search_remaining(az_engine *e, int32_t sims_target, int64_t *sum_out, int32_t *max_out, void *stream);
int az_set_virtual_loss(az_engine *e, double virtual_loss); /* > 0 (config/basic.py:46-47); default 0.3 */

/* core/MCTS_alphazero.py:174-178: pi = visits / visits.sum() as (n_slots, A) f32.  Optional extras
 * (NULL to skip): root child visits i32 (n_slots, A), child values f64 (n_slots, A) from the child's
 * own view, the root prior f32 (n_slots, A). */
int az_root_policy(az_engine *e, float *pi_dev, int32_t *visits_dev, double *child_q_dev,
                   float *root_prior_dev, void *stream);

/* training/self_play.py:38-56: the sample's nn_state = cat(history, player plane) of every active
 * slot's current position, f32 (n_slots, nn_channels, height, width). */
int az_root_nn_state(az_engine *e, float *nn_state_dev, void *stream);

/* training/self_play.py:73-83 + games/base.py:22-33: w = pi ** (1/tao); action = multinomial(w, 1),
 * taken as argmax(w / expo) over Exponential(1) draws (ATen's single-sample path); game.step(action);
 * the position, pi (as visit counts) and player are appended to the slot's on-device trajectory.
 * inv_tao_dev (n_slots) f32 = 1/tao per slot; expo_dev (n_slots, A) f32; forced_actions_dev
 * (n_slots) i32 or NULL: when given and >= 0 that action is played instead of a sampled one.
 * actions_dev (n_slots) i32 out (-1 for inactive slots); done_dev (n_slots) u8 out: 1 where the game
 * just ended (the slot is then inactive until az_reset_games). */
int az_play_moves(az_engine *e, const float *inv_tao_dev, const float *expo_dev,
                  const int32_t *forced_actions_dev, int32_t *actions_dev, uint8_t *done_dev, void *stream);

/* Per-game, counter-based random rows ("per-game RNG = Philox(seed, subsequence = game id)").  The
 * reference seeds one torch generator per worker (training/alphazero_trainer.py:175-177) and draws per move a
 * Dirichlet(ALPHA) row (core/MCTS_alphazero.py:187) and the multinomial sample of training/self_play.py:73-74
 * (= argmax(w / Exponential(1))).  Here, for every ACTIVE slot g, the rows are a pure function of
 * (seed, game_ids_dev[g], the slot's current ply, purpose) -- Philox4x32-10, key = seed, counter = {element | attempt << 16,
 * ply | purpose << 16, game id} -- so a game plays the same moves on any slot, slot count, rank or world size.
 *   noise_dev (n_slots, A) f32 or NULL: Dirichlet(alpha) rows (0 < alpha <= 1: Gamma by Ahrens-Dieter GS, normalised) for
 *   az_expand_backup / az_expand_select; expo_dev (n_slots, A) f32 or NULL: Exponential(1) rows for az_play_moves.
 *   game_ids_dev (n_slots) i64: the global game id of each slot (rank + world * k for the k-th game of a rank).
 * Rows of inactive slots are left untouched.  NOT torch's random stream (bit-equality with torch's Dirichlet is not
 * claimed: the tables remain data at this ABI, and a seeded replay of the reference feeds rows drawn by torch);
 * restated bit for bit in oracle/az_oracle.c (az_oracle_draw_rng).  az_philox4x32_10: the block function on the host
 * (Random123 known-answer vectors are checked against it). */
int az_draw_rng(az_engine *e, uint64_t seed, const int64_t *game_ids_dev, float alpha, float *noise_dev, float *expo_dev,
                void *stream);
int az_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

/* Per-slot game info, device-to-device copies into caller buffers (any may be NULL):
 * plies i32, player to move i32, active u8, result f32 (outcome from the last mover's view,
 * training/self_play.py:86-89; valid once done). */
int az_game_info(az_engine *e, int32_t *plies_dev, int32_t *player_dev, uint8_t *active_dev,
                 float *result_dev, void *stream);

/* training/self_play.py:110-139 _generate_experience_date for n_games finished slots: for game i
 * (slot slots_dev[i], T plies) writes T samples at rows offsets_dev[i] .. offsets_dev[i]+T-1, LAST
 * position first (the reference's put order): nn_state f32 (rows, nn_channels*height*width), pi f32
 * (rows, A), z f32 (rows). */
int az_emit_samples(az_engine *e, const int32_t *slots_dev, const int32_t *offsets_dev, int32_t n_games,
                    float *states_dev, float *pis_dev, float *z_dev, void *stream);

/* Device -> pinned host copy of `bytes` bytes by n_workgroups (1..1024; 16 is enough for the PCIe link) workgroups on
 * `stream`: how finished games' samples reach the host pool (training/self_play.py:124-126 hands the pool
 * `.detach().clone()`d CPU tensors) without a chip-wide copy kernel slowing the search it overlaps with.
 * dst_pinned_host: hipHostMalloc'ed / torch pin_memory() memory; both pointers 16-byte aligned.  The data is
 * visible to the host once the stream has reached this point (event or stream synchronize). */
int az_copy_to_host(void *dst_pinned_host, const void *src_dev, uint64_t bytes, int32_t n_workgroups, void *stream);

/* the 8-fold symmetry expansion the reference's pools apply on every put
 * (utils/experience_pool.py:27-31 -> game_cls.get_enhanced_data: utils/data_enhancer.py:13-70;
 * examples/dots_and_boxes/game.py:158-320), on the device: sample i (rows of states_dev (n, C_in*H*W),
 * pis_dev (n, A), z_dev (n)) yields 8 samples, in the reference's order (rot90 k, then its flip, k = 0..3),
 * written to rows (ring_write_idx + 8 i + t) mod ring_capacity of the out buffers, i.e. in the layout of
 * utils/share_ring_buffer.py:39-46; ring_capacity 0 = no wrap (rows 8 i + t). */
int az_augment_samples(az_engine *e, int32_t n, const float *states_dev, const float *pis_dev, const float *z_dev,
                       float *out_states_dev, float *out_pis_dev, float *out_z_dev, int64_t ring_capacity,
                       int64_t ring_write_idx, void *stream);

/* Counters for the roofline (algorithmic bytes per simulation need the measured mean depth and mean
 * children per visited node).  Synchronises. */
int az_read_stats(az_engine *e, az_stats *out, void *stream);
int az_reset_stats(az_engine *e, void *stream);

/* Row-tile tower (csrc/az_tower_h1.hip): the 15x15, 64-channel GomokuModel-family trunk (the network az_tower_forward
 * evaluates; model contract nn_models/base.py:17-19, examples/gomoku/model.py:41-90) on the 16-bit MFMA pipe with one board
 * row per cell tile -- no tap masks, no padding tile; 15 % faster than az_tower_forward on the same arithmetic.
 * elem: AZ_H1_BF16 (bf16 operands, f32 accumulation, bf16 activations: the arithmetic of az_tower_forward) or AZ_H1_F16 (f16
 * operands with exact power-of-two scalings: 8x the operand precision at the same rate).
 * Inputs as az_tower_forward_f32 (dense f32 planes, or the packed leaf batch + flags); feature rows (n_boards, feat_stride):
 * f32 for az_heads_forward_f32, or -- feat_bf16 != 0 -- bf16 for az_heads_forward.
 * Weights, 16-bit values of w * sw[layer] (alphazero_amd.fused packs them: _pack_stem_h1, _pack3x3_h1):
 *   wl_dev [layer][tap 9][k-step 2][co tile 4][lane 64][8]: lane = 16 kg + m holds input channels 32 ks + 8 kg .. + 7 of the
 *     output channel in row m of co tile t, which is channel 32 (t / 2) + 8 (m / 4) + 4 (t % 2) + m % 4;
 *   w0_dev [kernel row 3][co tile 4][lane 64][8]: one k-step per kernel row, k = 9 tx + plane (tx = the tap's column, 27 of
 *     32 used), rows ordered as above.
 * bias_scaled_dev (1 + n_conv_layers, 64) f32 = bias * sw[layer] * SA, SA = 16 (f16) or 1 (bf16), in NATURAL channel order;
 * sw_dev (1 + n_conv_layers) f32 powers of two <= 2^15 (all 1 for bf16; 1 for an f16 layer unless its weights are tiny);
 * headw_dev (3, 64), headb_dev (3) f32 as az_tower_forward_f32.  n_conv_layers even, <= 64.
 * overflow_dev (1) i32, may be NULL: set to 1 when an f16 activation exceeds the format's range (|x| > 4,094) -- results are
 * then invalid and the caller must use another tower. */
#define AZ_H1_F16 0
#define AZ_H1_BF16 1
int az_tower_forward_h1(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, int32_t elem,
                        const float *x_f32_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                        const void *w0_dev, const void *wl_dev, const float *bias_scaled_dev, const float *sw_dev,
                        const float *headw_dev, const float *headb_dev, void *feat_dev, int32_t feat_bf16, int32_t feat_stride,
                        int32_t *overflow_dev, void *stream);

/* Winograd f32 tower (csrc/az_tower_w.hip): the 64-channel GomokuModel-family trunk on 15x15 or 9x9 boards (the network az_tower_forward_f32
 * evaluates; model contract nn_models/base.py:17-19, examples/gomoku/model.py:41-90) in plain f32 arithmetic -- f32 operands,
 * f32 products and sums on v_mfma_f32_16x16x4_f32 -- with the 3x3 convolutions computed by the Winograd transform F(2x2, 3x3)
 * (the algorithm cuDNN / MIOpen choose for f32 3x3 convolutions): 2.1x fewer MFMAs than the direct convolution of
 * az_tower_forward_f32, same accuracy class, not bit-equal to it (another summation order).
 * Same inputs (dense f32 planes, or the packed leaf batch + flags) and f32 feature rows as az_tower_forward_f32; finish with
 * az_heads_forward_f32.  Weights = G g G^T of the BN-folded filters (alphazero_amd.fused._pack_winograd_f32):
 *   u0_dev [stage 2][j 2][kq 4][co tile 4][i 4][co 16][q 4] f32 (input channel 8 stage + 2 kq + j, 9 of 16 used; transform
 *   point (i, q); output channel 16 tile + co), ul_dev [n_conv_layers][stage 8][...] likewise;
 *   bias_dev [1 + n_conv_layers][64], headw_dev (3, 64), headb_dev (3) f32 as az_tower_forward_f32.
 * board_size 15 or 9 (two boards per workgroup), channels 64, n_conv_layers even, <= 36; 1 <= n_boards <= 4,194,304 (32-bit byte
 * offsets into the packed batch); packed_leaf_dev, u0_dev, ul_dev 16-byte aligned, every other pointer -- flags_dev included: the
 * look-ahead reads the aligned word that holds a ticket's flag bytes -- 4-byte aligned (AZ_EINVAL otherwise). */
int az_tower_forward_f32w(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                          const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *u0_dev,
                          const void *ul_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev,
                          float *feat_f32_dev, int32_t feat_stride, void *stream);

/* EXPLORATORY, SECONDARY (csrc/az_tower_h2.hip): the 64-channel GomokuModel-family tower (the network az_tower_forward_f32
 * evaluates; model contract nn_models/base.py:17-19) at f32-LEVEL ACCURACY on the f16 MFMA pipe -- every f32 operand split
 * into two f16 halves (22 significant bits), w x = w1 x1 + w1 x2 + w2 x1 on v_mfma_f32_16x16x32_f16 with f32 accumulation.
 * NOT the reference's arithmetic and not bit-equal to the f32 tower; same inputs / feature rows as az_tower_forward_f32
 * (finish with az_heads_forward_f32).  w0_dev / wl_dev: [tap 9][k-step][co tile 4][half 2][lane 64][8] f16 of w * sw[layer];
 * bias_scaled_dev (1 + n_conv_layers, 64) f32 = bias * sw[layer] * 16; sw_dev (1 + n_conv_layers) f32 powers of two
 * (alphazero_amd.fused packs them).  overflow_dev (1) i32, may be NULL: set to 1 when an activation exceeds f16's range
 * (|x| > 4,094) -- results are then invalid and the caller must use the f32 tower. */
int az_tower_forward_h2(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                        const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *w0_dev,
                        const void *wl_dev, const float *bias_scaled_dev, const float *sw_dev, const float *headw_dev,
                        const float *headb_dev, float *feat_f32_dev, int32_t feat_stride, int32_t *overflow_dev, void *stream);

/* No counterpart in the reference: a deterministic synthetic evaluator ("hash network") with exact
 * f32 outputs, so tree parity can be tested bit for bit without a neural network and the tree
 * kernels can be benchmarked alone.  Reads the batch written by az_select_gather; nn_input_dev == NULL: reads the packed
 * leaf batch instead (same outputs; rows of slots whose leaf needs no evaluation are left untouched). */
int az_eval_hash(az_engine *e, const void *nn_input_dev, float scale, float *logits_dev,
                 float *values_dev, void *stream);

/* No counterpart in the reference beyond the model contract (nn_models/base.py:17-19): a fused bf16 MFMA
 * evaluator for the residual conv tower of the GomokuModel family (examples/gomoku/model.py:41-90):
 * stem conv3x3(9->64), n_conv_layers/2 residual blocks of two conv3x3(64->64), the 1x1 head convs, all
 * with eval-mode BatchNorm folded in by the host and ReLU fused, activations resident in LDS.
 *   x_bf16_dev (n_boards, 9, N, N) bf16 = the batch az_select_gather writes with AZ_NN_BF16, or NULL to read
 *   the packed leaf batch instead (packed_ring_dev / packed_leaf_dev from az_leaf_buffers of a gomoku engine
 *   of the same board size; both may be NULL when x_bf16_dev is given); with list_dev / count_dev (same call,
 *   or NULL) only the listed slots are evaluated and only their rows of feat are written; with flags_dev (same
 *   call; used when no list is given) only the slots whose flag byte is non-zero -- no compaction launch needed;
 *   channels C in {64, 128} = the trunk width;
 *   w0_dev [9][1][2][C][8] bf16, wl_dev [n_conv_layers][9][C/16][2][C][8] bf16 ([tap][k-step][k-half][co][8 ci]);
 *   bias_dev [1+n_conv_layers][C] f32; headw_dev [3][C] f32, headb_dev [3] f32 (policy, value0, value1);
 *   feat_bf16_dev (n_boards, feat_stride) bf16 out: the inputs of the heads' Linear layers, row layout
 *   [policy N*N | zero pad to a multiple of 16 | value0 N*N | value1 N*N | zero pad to a multiple of 16];
 *   feat_stride >= az_heads_feat_stride(N) elements; the pads are never written: hand in a zeroed buffer.
 *   board_size N in {9, 15}.  Groups of boards are handed to workgroups through a ticket counter that lives in the
 *   library, on the GPU that owns `stream` (the calling thread's current device is irrelevant): one counter pair per
 *   stream for eager launches (32 streams per GPU before the least recently used entry is recycled), a pair of its
 *   own for every launch recorded into a HIP graph (32 per GPU before they repeat).  Errors: az_tower_last_error(),
 *   per host thread. */
int az_tower_forward(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const void *x_bf16_dev,
                     const void *packed_ring_dev, const void *packed_leaf_dev, const void *list_dev,
                     const void *count_dev, const void *flags_dev, const void *w0_dev, const void *wl_dev, const float *bias_dev,
                     const float *headw_dev, const float *headb_dev, void *feat_bf16_dev, int32_t feat_stride, void *stream);

/* The heads' Linear layers of the same network family (examples/gomoku/model.py:69-72, 84-90) for the whole batch:
 *   logits = feat[:, :N*N] @ Wp^T + bp;   value = tanh(relu(feat[:, KP:KP+2N*N] @ W1^T + b1) @ w2 + b2)
 * as bf16 MFMA GEMMs with f32 accumulation.  feat as written by az_tower_forward;
 *   wp_dev [ceil(N*N/32)][KP/16][2][32][8] bf16 and w1_dev [2][KV/16][2][32][8] bf16: [output tile][k-step][k-half][output][8 k],
 *   zero padded (KP, KV = N*N and 2*N*N rounded up to 16); bp_dev (N*N) f32, b1_dev (64) f32, w2_dev (64) f32, b2;
 *   logits_dev (n_boards, N*N) f32 and values_dev (n_boards) f32 out (every row is computed). */
int32_t az_heads_feat_stride(int32_t board_size);
int az_heads_forward(int32_t board_size, int32_t n_boards, const void *feat_bf16_dev, int32_t feat_stride, const void *wp_dev,
                     const float *bp_dev, const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2,
                     float *logits_dev, float *values_dev, void *stream);
const char *az_tower_last_error(void);

/* The same two calls at the REFERENCE'S precision: the reference runs its network in float32 (nn_models/base.py:17-19
 * under core/MCTS_alphazero.py:161-162, no autocast).  f32 operands and f32 accumulation on the f32 MFMAs (each output
 * is a k-ordered fmaf chain; dense peak 157.3 TFLOP/s).  64- or 128-channel trunks on board 9 or 15 (128 channels: at most 40
 * conv layers).
 *   x_f32_dev (n_boards, 9, N, N) f32 or NULL to read the packed leaf batch (with flags_dev, only the flagged slots);
 *   w0_dev [9 taps][2][2][C][4] f32, wl_dev [n_conv_layers][9][C/8][2][C][4] f32 ([tap][k-step of 8 ci][k-half][co][4 ci]);
 *   in w0 the 9 input channels sit at positions c + c / 3 of the 16 (0,1,2, 4,5,6, 8,9,10), the other positions are zero;
 *   bias_dev / headw_dev / headb_dev as for az_tower_forward; feat_f32_dev (n_boards, feat_stride) f32 out, same row
 *   layout and stride rule as az_tower_forward's (zeroed buffer: the pads are never written);
 *   az_heads_forward_f32: wp_dev [ceil(N*N/32)][KP/8][2][32][4] f32, w1_dev [2][KV/8][2][32][4] f32. */
int az_tower_forward_f32(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers, const float *x_f32_dev,
                         const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev, const void *w0_dev,
                         const void *wl_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev, float *feat_f32_dev,
                         int32_t feat_stride, void *stream);
int az_heads_forward_f32(int32_t board_size, int32_t n_boards, const float *feat_dev, int32_t feat_stride, const void *wp_dev,
                         const float *bp_dev, const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2,
                         float *logits_dev, float *values_dev, void *stream);

/* The reference's QuickModel family at the reference's precision (nn_models/quick_model.py:67-93 with the layout of
 * config/quick_model_config.py's defaults -- BatchNorm on, one transition conv, 3x3 head convs of 2 / 1 channels, two Linear
 * layers of 64 features per head -- at any depth; residual block nn_models/resnet_blocks.py:28-40: x + (conv3x3, BN, ReLU) x
 * block_layers, the sum taken after the last ReLU and not activated again; tanh on the value, quick_model.py:102-104).  The same
 * f32 MFMA tower kernel as az_tower_forward_f32 in its QuickModel instantiation, and the heads' Linear layers.
 *   n_blocks x block_layers conv layers after the transition conv (at most 40); x / packed_* / flags / w0 / wl / bias as for
 *   az_tower_forward_f32 (wl_dev [n_blocks * block_layers][9][C/8][2][C][4] f32, bias_dev [1 + n_blocks * block_layers][C]);
 *   headw_dev [3][9 taps][C] f32 = the 3x3 head convs (policy channel 0, policy channel 1, value) with BN folded in, headb_dev [3];
 *   feat_f32_dev (n_boards, feat_stride) f32 out, row layout [policy0 N*N | policy1 N*N | zero pad to a multiple of 16 |
 *   value N*N | zero pad to a multiple of 16] = az_heads_feat_stride(N) floats (zeroed buffer: the pads are never written);
 *   stash_dev: az_quick_stash_bytes(N, C) bytes of device memory the kernel keeps the running blocks' inputs in (a private
 *   region per workgroup; contents meaningless between calls; calls that share one stash must not overlap in time).
 * az_quick_heads_forward_f32:  logits = relu(feat[:, :2N*N] @ W1p^T + b1p) @ W2p^T + b2p;
 *                              value = tanh(relu(feat[:, KPQ:KPQ+N*N] @ W1v^T + b1v) @ w2v + b2v),  KPQ = 2N*N rounded up to 16;
 *   w1p_dev [2][KPQ/8][2][32][4], w2p_dev [ceil(N*N/32)][8][2][32][4], w1v_dev [2][KVQ/8][2][32][4] f32 ([output tile][k-step of
 *   8][k-half][output][4 k], zero padded; KVQ = N*N rounded up to 16); b1p (64), b2p (N*N), b1v (64), w2v (64) f32, b2v.
 * Boards 9 and 15, 64 or 128 channels.  Errors through az_tower_last_error(). */
int64_t az_quick_stash_bytes(int32_t board_size, int32_t channels);
int az_quick_tower_forward_f32(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_blocks, int32_t block_layers,
                               const float *x_f32_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                               const void *w0_dev, const void *wl_dev, const float *bias_dev, const float *headw_dev,
                               const float *headb_dev, float *feat_f32_dev, int32_t feat_stride, void *stash_dev, int64_t stash_bytes,
                               void *stream);
int az_quick_heads_forward_f32(int32_t board_size, int32_t n_boards, const float *feat_dev, int32_t feat_stride, const void *w1p_dev,
                               const float *b1p_dev, const void *w2p_dev, const float *b2p_dev, const void *w1v_dev,
                               const float *b1v_dev, const float *w2v_dev, float b2v, float *logits_dev, float *values_dev, void *stream);

/* The QuickModel family in bf16 throughput mode (bf16 operands and activations, f32 accumulation; NOT the reference's
 * precision): the bf16 tower kernel of az_tower_forward in its QuickModel instantiation and the heads' Linear layers.  Arguments
 * as for az_quick_tower_forward_f32 / az_quick_heads_forward_f32 with bf16 tensors packed as for az_tower_forward /
 * az_heads_forward: x_bf16_dev (n_boards, 9, N, N) bf16 or NULL (packed leaf batch; flags_dev or NULL; no list mode),
 * w0_dev [9][1][2][C][8], wl_dev [n_blocks * block_layers][9][C/16][2][C][8] bf16, bias_dev [1 + n_blocks * block_layers][C] f32,
 * headw_dev [3][9 taps][C] f32, headb_dev [3] f32; feat_bf16_dev (n_boards, feat_stride) bf16, row layout as the f32 call's;
 * the block's input stays in registers (no stash), any depth; channels 64 or 128, board 9 or 15.
 * az_quick_heads_forward: w1p_dev [2][KPQ/16][2][32][8], w2p_dev [ceil(N*N/32)][4][2][32][8], w1v_dev [2][KVQ/16][2][32][8] bf16;
 * biases and w2v f32. */
int az_quick_tower_forward(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_blocks, int32_t block_layers,
                           const void *x_bf16_dev, const void *packed_ring_dev, const void *packed_leaf_dev, const void *flags_dev,
                           const void *w0_dev, const void *wl_dev, const float *bias_dev, const float *headw_dev, const float *headb_dev,
                           void *feat_bf16_dev, int32_t feat_stride, void *stream);
int az_quick_heads_forward(int32_t board_size, int32_t n_boards, const void *feat_bf16_dev, int32_t feat_stride, const void *w1p_dev,
                           const float *b1p_dev, const void *w2p_dev, const float *b2p_dev, const void *w1v_dev, const float *b1v_dev,
                           const float *w2v_dev, float b2v, float *logits_dev, float *values_dev, void *stream);

/* The reference's two small example networks, one launch per evaluation, float32 on v_mfma_f32_16x16x4_f32 (the
 * reference's precision): examples/tictactoe/model.py (game AZ_GAME_TICTACTOE, input (B,9,3,3), 9 logits) and
 * examples/dots_and_boxes/model.py (game AZ_GAME_DAB, input (B,25,4,4), 24 logits); contract nn_models/base.py:17-19
 * ((policy logits, value in [-1,1])).  Input: x_f32_dev dense planes, or NULL = the engine's packed leaf batch
 * (packed_ring_dev / packed_leaf_dev from az_get_buffers, flags_dev = the needs-the-network bytes or NULL = all);
 * slots whose flag is 0 keep their logits / values rows.  Weights (device, packed by the host mirror
 * alphazero_amd/fused.py:SmallNetEvaluator): w1_dev [9 taps][C_in_padded/4][64][4] f32 (C_in padded to 16 / 32 with zeros),
 * w2_dev [9][16][64][4] f32, b1_dev / b2_dev [64]; head_dev = az_smallnet_head_floats(game) floats:
 *   tic-tac-toe:    W1' [36][4][192][4] (rows 0..127 the policy head's hidden layer, 128..191 the value head's; k = cell * 64 +
 *                   channel), b1' [192], policy Linear(128->9) weight [9][128] and bias [9], value Linear(64->1) weight [64], bias [1];
 *   dots-and-boxes: 1x1 head convs [3][64] (2 policy rows, 1 value row) and biases [3], policy Linear(32->24) [24][32] + [24],
 *                   value Linear(16->64) [64][16] + [64], Linear(64->1) [64] + [1].
 * Errors through az_tower_last_error(). */
int32_t az_smallnet_head_floats(int32_t game);
int az_smallnet_forward_f32(int32_t game, int32_t n_boards, const float *x_f32_dev, const void *packed_ring_dev,
                            const void *packed_leaf_dev, const void *flags_dev, const void *w1_dev, const float *b1_dev,
                            const void *w2_dev, const float *b2_dev, const float *head_dev, float *logits_dev, float *values_dev,
                            void *stream);

/* No counterpart in the reference.  For callers that CANNOT allocate device memory around the evaluation -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc: az_tower_forward followed by
 * az_heads_forward in one call, with the feature rows that travel between the two kernels kept in a static array
 * inside the library (AZ_STATIC_SCRATCH_BYTES per GPU, held until the process ends) instead of a caller-owned
 * feat buffer.  Arguments as for those two calls (flags_dev may be NULL; there is no list mode); the caller still
 * passes inputs, weights and the logits / values outputs.  Calls no hipMalloc / hipFree and no allocating library
 * routine.  n_boards above az_static_max_boards(board_size) (4,468 boards of 15x15, 11,565 of 9x9) is refused with
 * AZ_EINVAL and a message in az_tower_last_error().  Every call on a GPU shares that GPU's array: calls that overlap in
 * time on one GPU (two streams, two threads) race -- serialise them. */
#define AZ_STATIC_SCRATCH_BYTES (6u << 20)
int32_t az_static_max_boards(int32_t board_size);
int az_tower_heads_forward_static(int32_t board_size, int32_t channels, int32_t n_boards, int32_t n_conv_layers,
                                  const void *x_bf16_dev, const void *packed_ring_dev, const void *packed_leaf_dev,
                                  const void *flags_dev, const void *w0_dev, const void *wl_dev, const float *bias_dev,
                                  const float *headw_dev, const float *headb_dev, const void *wp_dev, const float *bp_dev,
                                  const void *w1_dev, const float *b1_dev, const float *w2_dev, float b2, float *logits_dev,
                                  float *values_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AZ_ENGINE_H */
