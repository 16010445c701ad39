/*
 * azhip.h -- C ABI of libazhip.so, the MI355X (gfx950) self-play engine.
 *
 * The reference (mtb2718/alphazero) is pure Python and has no FFI layer; its
 * boundary is the duck-typed Python surface Game / MCTreeNode / run_mcts /
 * SelfPlayWorker.  This library is what that surface binds to instead of the
 * reference's per-object Python: each entry point names the reference code
 * (file:line, relative to the reference checkout) whose work it does for G
 * games at once.
 *
 * Conventions
 *   - plain C types only; every pointer is either a HOST pointer or a DEVICE
 *     pointer as marked;  no torch / C++ types cross the boundary.
 *   - every call that launches work takes a hipStream_t (passed as void*) and
 *     is stream-ordered and asynchronous unless marked "synchronous".
 *   - return value: 0 = ok, negative = error; az_last_error() gives the text.
 *   - one engine per device, driven by one host thread (like the reference,
 *     not thread-safe).  The caller owns all I/O buffers, the engine owns the
 *     node pools and per-game state.
 *   - "slot" = index of an edge in a node's COMPACTED legal-move list, exactly
 *     the reference's action_index (alphazero/mcts.py:124-126).
 */
#ifndef AZHIP_H
#define AZHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GAME_TTT 0   /* alphazero/games/tictactoe.py:9-62   : 9 actions, 3x3 */
#define AZ_GAME_C4  1   /* alphazero/games/connectfour.py:75-120: 7 actions, 6x7 */
#define AZ_GAME_GENERIC 2 /* any alphazero/game.py:5-65 subclass with <= 15 actions: the tree lives on the
                             device, the rules stay with the caller (az_generic_* below)                 */

#define AZ_MAX_ACTIONS 9
#define AZ_MAX_MOVES   42
#define AZ_NO_NODE     0xFFFFFFFFu

/* leaf_info word written by az_select / az_begin_move, one per game:
 *   bits 0..7   kind      bits 8..15  number of moves played at the leaf
 *   bits 16..24 legal-action mask of the leaf position                      */
#define AZ_LEAF_NONE     0  /* nothing to evaluate for this game             */
#define AZ_LEAF_EVAL     1  /* un-expanded non-terminal leaf: needs (p, v)   */
#define AZ_LEAF_TERMINAL 2  /* terminal leaf: valued by the rules            */
#define AZ_LEAF_ROOT     3  /* un-expanded search root: needs p (v unused)   */
#define AZ_LEAF_CACHED   4  /* un-expanded leaf whose (p, v) az_sim_step found
                               in the evaluation cache and wrote to rows g of
                               p and v itself: the evaluator leaves it alone  */

#define AZ_WINNER_NONE (-2)
#define AZ_WINNER_DRAW (-1)

typedef struct az_engine az_engine;

typedef struct {
    int32_t  game;               /* AZ_GAME_*                                             */
    int32_t  num_games;          /* G concurrent games on this device                     */
    int32_t  node_capacity;      /* nodes per game pool; 0 = S*max_moves+max_moves+1      */
    int32_t  num_simulations;    /* S, config selfplay.num_expansions_per_sim             */
    int32_t  exploration_depth;  /* config selfplay.exploration_depth (selfplay.py:26)    */
    int32_t  reuse_search_tree;  /* config selfplay.reuse_search_tree (selfplay.py:34-38) */
    int32_t  device;             /* HIP device ordinal                                    */
    int32_t  model_version;      /* stamped into finished-game records                    */
    double   alpha, epsilon;     /* Dirichlet noise, mcts.py:110 defaults 0.3 / 0.25      */
    double   c_base, c_init;     /* mcts.py:52 defaults 19652 / 1.25                      */
    uint64_t seed;               /* device RNG: stream of game uid = f(seed, uid)         */
    uint64_t first_game_uid;     /* uid of game slot 0; slot g starts at first+g, and a
                                    restarted slot continues at uid + uid_stride          */
    uint64_t uid_stride;         /* 0 = num_games (set to world_size*G when sharded)      */
} az_engine_config;

/* one finished game, replaces the row written by
 * SelfPlayDatabase.write_game (alphazero/replay_buffer.py:28-47)            */
typedef struct {
    uint64_t uid;
    int32_t  num_moves;
    int32_t  winner;                                   /* -1 draw, 0, 1      */
    int32_t  model_version;                            /* current when the game ended   */
    int32_t  first_model_version;                      /* current when the game started */
    uint8_t  history[AZ_MAX_MOVES + 6];                /* actions            */
    uint8_t  num_edges[AZ_MAX_MOVES + 6];              /* legal moves / move */
    uint32_t visits[AZ_MAX_MOVES][AZ_MAX_ACTIONS];     /* compacted root N   */
} az_game_record;

/* throughput / roofline counters summed over games (host struct) */
typedef struct {
    uint64_t sims;          /* simulations completed                          */
    uint64_t select_steps;  /* edges descended (sum of select depths)         */
    uint64_t leaf_evals;    /* non-terminal leaves expanded                   */
    uint64_t root_evals;    /* roots expanded                                 */
    uint64_t moves;         /* positions produced                             */
    uint64_t games;         /* games finished                                 */
    uint64_t errors;        /* games that hit node-pool exhaustion            */
} az_stats;

const char* az_last_error(void);
int  az_version(void);

/* ---- engine lifetime (synchronous) ------------------------------------ */
int  az_engine_create(const az_engine_config* cfg, az_engine** out);
int  az_engine_destroy(az_engine* e);
/* selfplay.py:13-14: every slot gets an empty game and an un-expanded root  */
int  az_reset_games(az_engine* e, void* stream);
/* Every game in progress is given up (no record) and its slot starts the game that would have come next, with an
 * un-expanded root, under the model version current now.  For callers that want the reference's rule that a game is
 * played from start to end with one set of weights (selfplay.py:16-22,41): call it when new weights arrive.        */
int  az_abandon_games(az_engine* e, void* stream);
/* PUCT table, HOST pointer, n rows of {Cs, sqrt(Ns)} indexed by Ns = sum N
 * (mcts.py:55-57 evaluated by the host's numpy, so the transcendental is the
 * reference's own).  Synchronous.  az_engine_create installs a table from the
 * C library's log that covers every visit sum the engine's searches can reach
 * (num_simulations * max moves with tree reuse); the device never computes a
 * logarithm itself: a visit sum beyond the table's last row uses that row and
 * raises error bit 3 (az_engine_error_flags).                                */
int  az_set_puct_lut(az_engine* e, const double* lut_host, int32_t n);
int  az_set_model_version(az_engine* e, int32_t version);

/* ---- one move of self-play, lock-step over all G games ------------------
 * selfplay.py:24-38 / mcts.py:110-158, per move:
 *   az_begin_move   -> leaves for un-expanded roots           (mcts.py:114-115)
 *   az_apply_root   <- p for those; noise on every root       (mcts.py:116-117)
 *   S x { az_select -> leaves ; az_expand_backup <- p, v }    (mcts.py:119-158)
 *   az_choose_and_advance                                     (selfplay.py:26-38)
 * DEVICE buffers: leaf_bb uint64[G][2] = {stones of side to move, stones of
 * previous mover} as bitboards (C4: bit col*7+row; TTT: bit square);
 * leaf_info int32[G]; p float[G][A] = ACTION-indexed soft-maxed priors
 * (entries of illegal actions ignored); v float[G].                          */
int  az_begin_move(az_engine* e, uint64_t* leaf_bb, int32_t* leaf_info, void* stream);
/* eta: DEVICE double[G][A] slot-indexed Dirichlet draws, or NULL = device RNG */
int  az_apply_root(az_engine* e, const float* p, const double* eta, void* stream);
int  az_select(az_engine* e, uint64_t* leaf_bb, int32_t* leaf_info, void* stream);
int  az_expand_backup(az_engine* e, const float* p, const float* v, void* stream);
/* fused az_expand_backup (if p != NULL) followed by az_select.  With the
 * evaluation cache enabled rows of p and v are also WRITTEN (AZ_LEAF_CACHED) */
int  az_sim_step(az_engine* e, float* p, float* v,
                 uint64_t* leaf_bb, int32_t* leaf_info, void* stream);
/* Evaluation cache (no counterpart in the reference, which evaluates every
 * leaf: mcts.py:139-142).  A direct-mapped table of 2^log2_entries 64-byte
 * entries in device memory, keyed by the leaf position and the model version:
 * az_expand_backup / az_sim_step record the (p, v) they are given for an
 * AZ_LEAF_EVAL leaf, and az_sim_step answers a later leaf with the same
 * position and version from the table (AZ_LEAF_CACHED).  Only for evaluators
 * that are pure functions of the position whose rows do not depend on the
 * rest of the batch; az_set_model_version invalidates.  log2_entries 0 turns
 * it off and frees the table.  Synchronous.                                  */
int  az_eval_cache_enable(az_engine* e, int32_t log2_entries);
/* leaves answered from the cache since it was enabled.  Synchronous.         */
int  az_eval_cache_hits(az_engine* e, uint64_t* hits, void* stream);
/* num_sims simulations per game inside ONE launch with the UniformModel
 * evaluated in place (models.py:90-95: p = 1/n_legal, v = 0)                */
int  az_run_sims_uniform(az_engine* e, int32_t num_sims, void* stream);
/* roots needing expansion get the uniform prior in place, then noise (eta as
 * in az_apply_root)                                                          */
int  az_begin_move_uniform(az_engine* e, const double* eta, void* stream);
/* u: DEVICE double[G] uniforms for sample_action, or NULL = device RNG.
 * Optional DEVICE outputs (may be NULL): action int32[G], num_edges int32[G],
 * visits uint32[G][A] (compacted, zero padded), winner int32[G]
 * (AZ_WINNER_NONE while the game goes on).  Finished games are appended to
 * the engine's record ring and their slot restarts with a new uid.          */
int  az_choose_and_advance(az_engine* e, const double* u, int32_t* action, int32_t* num_edges,
                           uint32_t* visits, int32_t* winner, void* stream);

/* ---- finished games (synchronous on `stream`) --------------------------
 * copies up to max_records finished games to HOST memory and empties the ring;
 * replaces ReplayBufferDataset.add_game (replay_buffer.py:165-167)          */
int  az_drain_finished(az_engine* e, az_game_record* out_host, int32_t max_records,
                       int32_t* num_out, void* stream);
/* The same hand-over without leaving the device and without a host synchronisation: up to max_records
 * finished games are copied to out_dev (DEVICE az_game_record[max_records]), head_dev (DEVICE int32[2])
 * receives {records copied, error bits as az_engine_error_flags}, the ring is emptied; everything is
 * ordered on `stream`.  For records that travel on to another rank (gather over RCCL) or into a
 * device-side replay window (az_examples_from_records).  Replaces replay_buffer.py:165-167.           */
int  az_take_finished_device(az_engine* e, az_game_record* out_dev, int32_t max_records, int32_t* head_dev, void* stream);
int  az_get_stats(az_engine* e, az_stats* out_host, void* stream);
/* Error bits of any game, as seen by the az_drain_finished calls since the last az_reset_games: bit 0 = a
 * game's node pool ran out (its later simulations are lost), bit 1 = the record ring overflowed (finished
 * games were dropped: drain more often), bit 4 = a wave of az_c4_move_fused found nothing to do for far
 * longer than any healthy launch lets it wait and the launch was abandoned (a fault in the library or
 * the device; that move's searches are incomplete), bit 3 = a node's visit sum went beyond the PUCT table
 * (az_set_puct_lut: install a longer one; scores past the end used the last row), bit 5 = a search path of
 * a user-defined game was deeper than 48 levels (az_generic_select failed and left the tree alone).  Also
 * brought up to date by az_get_tree.  Host-side read, no device work; -1 for a null engine.  The reference
 * cannot fail this way (it allocates per node and per game).                                            */
int  az_engine_error_flags(const az_engine* e);

/* Training examples of finished games, built on the device: what ReplayBufferDataset.__getitem__
 * (alphazero/replay_buffer.py:113-147) returns for every state of n records, in game-major order.
 * records = DEVICE az_game_record[n]; state_offset = DEVICE int32[n+1], exclusive prefix sum of
 * num_moves (total = state_offset[n]).  Outputs (DEVICE): x float32[total][2][H][W] = render() of
 * the position before move m, p float32[total][A] = visits / sum with the visits cast to int16 as
 * the reference stores them (replay_buffer.py:30-35), z float32[total] = outcome for the player to
 * move (game.py:42-47), valid uint8[total][A], version int32[total] (may be NULL).            */
int  az_examples_from_records(int32_t game, const az_game_record* records, int32_t n, const int32_t* state_offset,
                              int32_t total, float* x, float* p, float* z, uint8_t* valid, int32_t* version,
                              void* stream);

/* ---- tree transfer for the single-game MCTreeNode / run_mcts mirror ------
 * node image (HOST): uint32[num_nodes][A+1][4]; edge i = {N, W bits, P bits,
 * child index or AZ_NO_NODE}; last row = {num_edges, sum N, 0, 0}.
 * az_set_position loads a root position (history replayed on the host side
 * into bitboards by the caller) and a tree for game slot g; root_node may be
 * AZ_NO_NODE for an un-expanded root.  Synchronous.                          */
int  az_set_position(az_engine* e, int32_t g, uint64_t bb_next, uint64_t bb_prev, int32_t num_moves,
                     const uint32_t* nodes_host, int32_t num_nodes, uint32_t root_node);
int  az_get_tree(az_engine* e, int32_t g, uint32_t* nodes_host, int32_t max_nodes,
                 int32_t* num_nodes, uint32_t* root_node);
int  az_node_words(const az_engine* e);   /* uint32 words per node = 4*(A+1) */

/* ---- user-defined games (engine created with AZ_GAME_GENERIC, node_capacity given) ---------------------
 * The search tree, PUCT select, expansion and backup run on the device with the numerics of the built-in
 * games; the Game object (alphazero/game.py:5-65: valid_actions, take_action, terminal, terminal_value,
 * render) is only ever touched by the caller.  Per simulation (alphazero/mcts.py:119-158):
 *   az_generic_select         -> depth and the action INDICES chosen from the root (mcts.py:123-127); the caller
 *                                replays them on a clone of its game (a terminal state is never expanded, so
 *                                the walk ends on it or on an un-expanded edge, as the reference's does)
 *   az_generic_expand_backup  <- n_edges = number of legal moves at the leaf and their priors (0 for a terminal
 *                                leaf: nothing is expanded), v_edge = the value for the leaf's edge: the
 *                                terminal value (mcts.py:134) or the evaluator's value negated (mcts.py:144)
 * az_generic_root expands the root when it has no node yet (n_edges, priors) and, when eta is given, blends
 * the Dirichlet noise (mcts.py:114-117).  Host pointers; az_generic_select is synchronous.  Trees go in and
 * out with az_set_position / az_get_tree (A = 15: 64 words per node).  One game per call.               */
int  az_generic_root(az_engine* e, int32_t g, int32_t n_edges, const float* p_host, const double* eta_host, void* stream);
int  az_generic_select(az_engine* e, int32_t g, int32_t* depth_host, uint8_t* slots_host, void* stream);
int  az_generic_expand_backup(az_engine* e, int32_t g, int32_t n_edges, const float* p_host, float v_edge, void* stream);

/* ---- stateless bulk game kernels (K1-K4), DEVICE buffers -----------------
 * hist uint8[n][stride], len int32[n]  ->  bitboards + rule queries
 * (connectfour.py:80-120, tictactoe.py:26-62, game.py:23-47)               */
int  az_positions_from_history(int32_t game, const uint8_t* hist, const int32_t* len, int32_t stride,
                               int32_t n, uint64_t* bb, int32_t* num_moves, void* stream);
int  az_game_query(int32_t game, const uint64_t* bb, const int32_t* num_moves, int32_t n,
                   int32_t* legal_mask, int32_t* terminal, int32_t* winner, void* stream);
int  az_game_step(int32_t game, uint64_t* bb, int32_t* num_moves, const int32_t* action, int32_t n,
                  int32_t* status, void* stream);
/* x float[n][2][H][W], plane 0 = side to move (connectfour.py:112-120);
 * valid uint8[n][A] (mcts.py:103-104); either may be NULL                    */
int  az_render(int32_t game, const uint64_t* bb, int32_t n, float* x, uint8_t* valid, void* stream);
/* mcts.py:106: softmax over the legal logits only; p float[n][A], zeros at
 * illegal actions.  legal from leaf_info (bits 16..24).                      */
int  az_masked_softmax(int32_t game, const float* logits, const int32_t* leaf_info, int32_t n,
                       float* p, void* stream);
/* mcts.py:52-59 for n nodes given as arrays (N uint32, W, P float, [n][A],
 * num_edges int32[n]) -> score double[n][A], argmax int32[n]; uses the
 * engine's PUCT table                                                        */
int  az_ucb_score(az_engine* e, const uint32_t* N, const float* W, const float* P,
                  const int32_t* num_edges, int32_t n, double* score, int32_t* argmax, void* stream);

/* A training batch from a replay window kept on the device as a ring of `capacity` states (tensors wx
 * [capacity][x_floats], wp [capacity][num_actions], wz [capacity], wvalid uint8 [capacity][num_actions]; count_dev =
 * DEVICE int64 {states ever added}): `batch` states drawn uniformly from the last min(count, capacity), without
 * replacement while batch fits, with replacement otherwise -- ReplayBufferSampler, alphazero/replay_buffer.py:175-195
 * -- and gathered into x / p / z / valid.  The draw is a function of (seed, draw_dev[0]); the launch increments
 * draw_dev[0] (DEVICE int64), so a recorded launch draws a new batch at every replay.  capacity <= 262144,
 * batch <= 2048.  One launch, no host synchronisation.                                                           */
int  az_sample_window(const float* wx, const float* wp, const float* wz, const uint8_t* wvalid, int32_t x_floats,
                      int32_t num_actions, const int64_t* count_dev, int32_t capacity, uint64_t seed, int64_t* draw_dev,
                      int32_t batch, float* x, float* p, float* z, uint8_t* valid, void* stream);

/* ---- training step of the Connect Four network, hand-written forward and backward ---------------------------
 * Replaces, for AlphaZeroC4 with 32 trunk channels in training mode, the forward, loss and backward of
 * alphazero/training.py:57-71 (model(x, p_valid), AlphaZeroLoss, total.backward(): alphazero/models.py:6-87,
 * games/connectfour.py:47-72, loss.py:29-63) -- fp32, batch-norm on the statistics of the batch, running statistics
 * updated as torch does (momentum bn_momentum, unbiased variance, once per application of the shared block).  All
 * pointers are DEVICE pointers to tensors in torch's own layouts (conv weights [co][ci][ky][kx], linear weights
 * [out][in]); every *_grad tensor is overwritten with d(policy loss + value loss) / d(parameter).                */
typedef struct az_bn {
    const float* weight; const float* bias;              /* [channels] */
    float* running_mean; float* running_var;             /* [channels], updated */
    int64_t* num_batches_tracked;                        /* [1], incremented once per application */
    float* weight_grad; float* bias_grad;
} az_bn;
typedef struct az_c4_train_net {
    int32_t num_blocks, reserved;
    float bn_momentum, bn_eps;                           /* torch defaults: 0.1, 1e-5 */
    const float* stem_w; float* stem_w_grad;             /* [32][2][3][3] */
    az_bn stem_bn;
    const float* c1_w; float* c1_w_grad; az_bn bn1;      /* the shared residual block: [32][32][3][3] each */
    const float* c2_w; float* c2_w_grad; az_bn bn2;
    const float* pol_conv_w; float* pol_conv_w_grad;     /* [2][32][1][1] */
    const float* pol_conv_b; float* pol_conv_b_grad;     /* [2] */
    az_bn pol_bn;
    const float* pol_fc_w; float* pol_fc_w_grad;         /* [42][84] */
    const float* pol_fc_b; float* pol_fc_b_grad;         /* [42] */
    const float* val_conv_w; float* val_conv_w_grad;     /* [1][32][1][1] */
    az_bn val_bn;
    const float* val_fc1_w; float* val_fc1_w_grad;       /* [256][42] */
    const float* val_fc1_b; float* val_fc1_b_grad;       /* [256] */
    const float* val_fc2_w; float* val_fc2_w_grad;       /* [1][256] */
    const float* val_fc2_b; float* val_fc2_b_grad;       /* [1] */
} az_c4_train_net;
/* bytes of DEVICE workspace az_c4_train_fwd_bwd needs for this batch size and depth (contents need not be kept
 * between calls and need no initialisation)                                                                   */
long az_c4_train_workspace_bytes(int32_t batch, int32_t num_blocks);
/* x float [batch][2][6][7] (ConnectFour.render planes), valid uint8 [batch][7], target_p float [batch][7], z float
 * [batch]; out3 DEVICE float[3] = {policy loss, value loss, max |sum(target) - 1|} as az_loss_fwd_bwd;
 * logits7_out [batch][7] / vhat_out [batch] receive the network's outputs when not NULL.  No allocation, no host
 * synchronisation (it can be recorded in a HIP graph).  batch >= 2.
 * Two launches on `stream` when the batch fits the device with one board per compute unit
 * (az_c4_train_one_launch_max_batch): the weights' re-ordering and ONE launch for forward, loss, backward and the
 * batch sums, whose workgroups exchange the batch statistics among themselves; it needs all `batch` workgroups
 * resident, waits about a second for a workgroup that does not arrive and then ends with out3 = {NaN, NaN, +inf}
 * and no gradient to be used.  Larger batches, or AZ_TRAIN_CHAIN=1 in the environment: the chain below.  The first
 * call on a device sets a kernel attribute: make it outside any stream capture.                                   */
int  az_c4_train_fwd_bwd(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p,
                         const float* z, int32_t batch, void* workspace, float* out3, float* logits7_out,
                         float* vhat_out, void* stream);
/* the same pass as a chain of 4 * num_blocks + 11 launches, one step each, the launch boundaries as the only
 * synchronisation: any batch size, the same bits as the one-launch form                                          */
int  az_c4_train_fwd_bwd_chain(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p,
                               const float* z, int32_t batch, void* workspace, float* out3, float* logits7_out,
                               float* vhat_out, void* stream);
/* largest batch the one-launch form takes on the current device for this depth (0: none)                        */
int  az_c4_train_one_launch_max_batch(int32_t num_blocks, int32_t* max_batch);

/* ---- fused Connect Four network forward (K8/K12) ---------------------------
 * Whole AlphaZeroC4 forward (alphazero/models.py:70-87, connectfour.py:54-72) for n
 * leaves given as bitboards, BN folded, trunk channels <= 32, bf16 MFMA trunk with
 * fp32 accumulation; writes p float[n][7] (soft-max over legal columns, zeros at full
 * columns, mcts.py:106) and v float[n].  blob = DEVICE buffer of az_c4net_blob_bytes()
 * bytes packed by alphazero_amd/net.py (layout in csrc/az_net.hip).               */
int  az_c4net_blob_bytes(void);
/* The module's parameters -> the packed weights of az_c4net_bf16_forward / az_c4_move_fused, in one launch: eval-mode
 * batch-norm (running statistics, bn_eps) folded into the convs in double, rounded to float and then to bf16 -- the
 * bytes alphazero_amd.net._pack_c4_blob_torch produces with tensor operations.  `net` as for az_c4_train_fwd_bwd (the
 * *_grad and num_batches_tracked members are not read); channels = the trunk's width, 1..32; blob DEVICE
 * [az_c4net_blob_bytes()].  This is what reaches the self-play kernels after every published model
 * (alphazero/selfplay.py:16-22 pulls the latest weights before each game).                                         */
int  az_c4net_pack_bf16(const az_c4_train_net* net, int32_t channels, double bn_eps, uint8_t* blob, void* stream);
int  az_c4net_bf16_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                           const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream);

/* Same network entirely in fp32 (weights, activations, accumulation; trunk on
 * v_mfma_f32_32x32x2_f32): the parity path, agrees with the fp32 torch statement to ~1e-6.
 * blob = DEVICE float buffer of az_c4net_f32_blob_floats() floats (alphazero_amd/net.py
 * pack_c4_blob_f32).  Evaluates every one of the n leaves.                                */
int  az_c4net_f32_blob_floats(void);
int  az_c4net_f32_forward(const float* blob, int32_t num_blocks, const uint64_t* leaf_bb, int32_t n,
                          float* p, float* v, void* stream);

/* Same network at fp32 grade on the fp16 matrix cores: every operand of the trunk is a pair of fp16 numbers
 * (leading and following 11 bits), every product three v_mfma_f32_16x16x32_f16 with fp32 accumulation, the fully
 * connected heads in fp32 (csrc/az_net_hx_dev.h).  Outputs within 1e-5 relative of the fp32 torch statement, like
 * az_c4net_f32_forward, at several times its rate; a leaf's (p, v) do not depend on the batch it is in.  blob =
 * DEVICE buffer of az_c4net_hx_blob_bytes() bytes (alphazero_amd/net.py pack_c4_blob_hx); leaf_info as for
 * az_c4net_bf16_forward (NULL = every leaf).  Activations saturate at 65504.  Replaces the same reference lines. */
int  az_c4net_hx_blob_bytes(void);
/* az_c4net_pack_bf16 for this layout (the bytes of alphazero_amd.net._pack_c4_blob_hx_torch).  A folded weight beyond
 * the fp16 range (|w| >= 6e4) sets range_error_dev[0] (DEVICE int32, never cleared here) to 1.                    */
int  az_c4net_pack_hx(const az_c4_train_net* net, int32_t channels, double bn_eps, uint8_t* blob,
                      int32_t* range_error_dev, void* stream);
int  az_c4net_hx_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                         const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream);

/* 128-channel network (e.g. 10x128) on bf16 MFMA: stem, residual tower with LDS-resident
 * activations, heads -- three launches, plus one that lists the leaves to evaluate when leaf_info is
 * given (kinds EVAL / ROOT, as az_c4net_bf16_forward; NULL = every leaf).  workspace = DEVICE buffer
 * of az_c4net128_workspace_bytes(n) bytes (contents need not be preserved between calls), blob =
 * az_c4net128_blob_bytes() bytes packed by alphazero_amd/net.py pack_c4_blob_128.  Rows of p / v
 * whose leaf is not evaluated are left untouched.  n <= 32768 when leaf_info is given.
 * Replaces the same reference lines as az_c4net_bf16_forward.                                   */
int  az_c4net128_blob_bytes(void);
long az_c4net128_workspace_bytes(int32_t n);
int  az_c4net128_bf16_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                              const int32_t* leaf_info, int32_t n, void* workspace, float* p, float* v,
                              void* stream);

/* All S simulations of a move in ONE launch, Connect Four with the <= 32-channel network: replaces
 * the S x { network, az_sim_step } loop above (selfplay.py:24-38 / mcts.py:119-158 between
 * az_apply_root and az_choose_and_advance).  A workgroup owns its games for the whole move, and
 * every game goes through select, network, expand / backup at its own pace (the games of a workgroup
 * are not in lock step; which games share a workgroup is decided per call from their cost in the
 * previous call).  blob as for az_c4net_bf16_forward; leaf_bb / leaf_info / p / v are the engine-sized
 * DEVICE buffers of the stepwise calls (used as scratch between search steps and network tasks).  Up
 * to az_c4_move_fused_max_games() games per engine (16384 on an MI355X); the evaluation cache is used
 * when enabled.  Results are bit-identical to the stepwise loop with az_c4net_bf16_forward.          */
int  az_c4_move_fused(az_engine* e, const uint8_t* blob, int32_t num_blocks, uint64_t* leaf_bb,
                      int32_t* leaf_info, float* p, float* v, void* stream);
/* az_c4_move_fused with the network at fp32 grade (the arithmetic of az_c4net_hx_forward; blob =
 * az_c4net_hx_blob_bytes() bytes, pack_c4_blob_hx): replaces the same reference lines at the reference's own
 * network precision (alphazero/mcts.py:94-107 evaluates in fp32).  Every leaf is evaluated alone by one wave,
 * trunk and heads; same limits, same staging buffers, same use of the evaluation cache.  Results are bit-identical
 * to the stepwise loop with az_c4net_hx_forward.                                                            */
int  az_c4_move_fused_hx(az_engine* e, const uint8_t* blob, int32_t num_blocks, uint64_t* leaf_bb,
                         int32_t* leaf_info, float* p, float* v, void* stream);
/* largest engine (games) az_c4_move_fused accepts on the engine's device: 64 games per workgroup, one
 * workgroup per compute unit (or per az_c4_move_fused_limit_workgroups)                                 */
int  az_c4_move_fused_max_games(const az_engine* e);
/* No counterpart in the reference (which has one game per process, alphazero/train.py:131-134).  By default
 * az_c4_move_fused spreads the engine's games over one workgroup per compute unit; this call caps the number
 * of workgroups (0 = back to the default), e.g. to leave compute units to a training stream, or to put
 * many games into each workgroup of a small engine.  How many games share a workgroup decides how its
 * leaves are batched (alone, in pairs, heads per board or per batch) but never what a game's tree becomes. */
int  az_c4_move_fused_limit_workgroups(az_engine* e, int32_t max_workgroups);
/* What the workgroups of az_c4_move_fused did since the last call (or az_reset_games), HOST uint64[4]:
 * network tasks with two boards, with one board, head batches, boards in head batches.  Synchronous;
 * resets the counters.  No counterpart in the reference (alphazero/mcts.py:94-107 evaluates one leaf).   */
int  az_c4_move_fused_counters(az_engine* e, uint64_t* out4_host, void* stream);

/* az_c4_move_fused without its four staging arguments and without any allocation.  No counterpart in the
 * reference; for callers that cannot allocate device memory or pass a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The staging arrays (leaf positions, leaf words, priors,
 * values that travel between search steps and network tasks) live in a fixed area of device memory that is part
 * of the library image, 832 KiB per GPU, sized for engines of up to 16384 games: a larger engine is refused
 * with an error.  The call issues its two kernel launches on `stream` and nothing else (no hipMalloc / hipFree,
 * no memset: every staging word is written by the kernel before it is read, so stale contents of the previous
 * call are harmless).  All engines on one GPU share that GPU's area: two calls that overlap in time on one GPU
 * race; calls ordered on one stream, or on different GPUs, do not.  Trees, visit counts and records are
 * bit-identical to az_c4_move_fused.                                                                     */
int  az_c4_move_fused_static(az_engine* e, const uint8_t* blob, int32_t num_blocks, void* stream);

/* Tic-tac-toe network (alphazero/models.py:70-87 with shape_in [2,3,3], shape_out [9], as in
 * alphazero/config/tictactoe.yaml:4-11) entirely in fp32, one launch: stem, shared residual block x
 * num_blocks, policy head with soft-max over the empty squares (mcts.py:101-106), value head.
 * channels = 4, 8 or 16 trunk channels.  blob = DEVICE float buffer of
 * az_tttnet_f32_blob_floats(channels) floats (alphazero_amd/net.py pack_ttt_blob_f32; -1 for an
 * unsupported width).  leaf_bb as written by the engine (bit = square); leaf_info may be NULL (every
 * row is evaluated), else only rows of kind EVAL / ROOT are evaluated and written.
 * p float[n][9], v float[n].                                                                      */
int  az_tttnet_f32_blob_floats(int32_t channels);
int  az_tttnet_f32_forward(const float* blob, int32_t channels, int32_t num_blocks, const uint64_t* leaf_bb,
                           const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream);

/* ---- training step, the hand-written parts (alphazero/training.py:42-75) ---------------------------------
 * az_loss_fwd_bwd: AlphaZeroLoss (alphazero/loss.py:29-63) and its gradients in one launch.  DEVICE buffers:
 * logits float[B][A] (the network's raw policy output), v_hat float[B], target float[B][A] (search
 * distribution), z float[B], valid uint8[B][A]  ->  out3 float[3] = {policy loss = batch mean of the masked
 * cross-entropy, value loss = batch mean of (z - v_hat)^2, max |sum(target) - 1| (the reference asserts it is
 * below 1e-6, loss.py:58-59)}, dlogits float[B][A] and dv float[B] = gradients of (policy loss) and (value
 * loss) with respect to logits and v_hat.  A = 7 or 9.
 * az_sgd_momentum: torch.optim.SGD's update as the reference configures it (training.py:27-29: momentum,
 * weight decay, no dampening) over one flat fp32 buffer of n elements: d = grad + wd*p; buf = first_step ? d :
 * momentum*buf + d; p -= lr*buf.                                                                          */
int  az_loss_fwd_bwd(int32_t num_actions, const float* logits, const float* v_hat, const float* target, const float* z,
                     const uint8_t* valid, int32_t batch, float* out3, float* dlogits, float* dv, void* stream);
int  az_sgd_momentum(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                     float weight_decay, int32_t first_step, void* stream);
/* az_sgd_momentum with the learning rate read from DEVICE memory (lr_dev float[1]) when the kernel runs, so that
 * the launch can be recorded once in a HIP graph and replayed while a schedule (training.py:30-31, MultiStepLR)
 * changes the rate; momentum_buf must start as zeros (then the first step's buf = d, as torch.optim.SGD's).   */
int  az_sgd_momentum_lr_dev(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                            float momentum, float weight_decay, void* stream);

/* az_sgd_momentum_lr_dev behind a check that is evaluated on the device: the update (parameters and momentum) is
 * applied only when guard_dev[0] < guard_limit; otherwise nothing is written and withheld_dev[0] (DEVICE int32) is
 * incremented.  For recorded steps, where the host cannot look at the batch between the loss and the update: with
 * guard_dev = out3 + 2 of az_loss_fwd_bwd and guard_limit = 1e-6 a batch that fails the reference's assertion
 * "Target policy must be a valid distribution" (alphazero/loss.py:58-59) changes no weight, as in the reference,
 * where the assertion fires before optimizer.step() (training.py:59-71).                                       */
int  az_sgd_momentum_guarded(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                             float momentum, float weight_decay, const float* guard_dev, float guard_limit,
                             int32_t* withheld_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
