/*
 * takgpu.h — C ABI of libtakgpu.so, the MI355X (gfx950) batched Tak self-play engine.
 *
 * This is the drop-in boundary for the ONE hot path of ViliamVadocz/tak: batched AlphaZero
 * self-play (board move-gen / step / terminal / encode, policy-value resnet forward, MCTS).
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repo).  Plain pointers and sizes only; no torch / tch types cross this line.
 *
 * Conventions
 *   - every function returns TG_OK (0) or a negative TgStatus; tg_last_error() gives the
 *     thread-local message of the last failure.  Nothing aborts, nothing throws.
 *   - "host" entry points take host pointers, do H2D, run the HIP kernels, D2H, and
 *     synchronise before returning.  "_dev" entry points take device pointers that must stay
 *     valid until the engine's stream is synchronised (tg_sync) and do not synchronise.
 *   - the caller owns every buffer it passes; the engine never frees caller memory.
 *   - one engine handle per GPU / per process; calls on one handle must be serialised by the
 *     caller (same contract as `&NET` in reference train/src/self_play.rs:96).
 *   - there is NO CPU fallback: if no HIP device is present every compute entry point fails
 *     with TG_ERR_NO_DEVICE.
 */
#ifndef TAKGPU_H
#define TAKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 5

/* The library is built with -fvisibility=hidden: the entry points below are ALL it exports. */
#if defined(__GNUC__) || defined(__clang__)
#define TG_API __attribute__((visibility("default")))
#else
#define TG_API
#endif

/* ---------------------------------------------------------------------------------------
 * Status codes.  TG_PLAY_* mirror reference tak/src/error.rs:4-15 (PlayError) and
 * :55-58 / :72-76 (StackError / TakeError) one to one.
 * ------------------------------------------------------------------------------------- */
typedef enum TgStatus {
    TG_OK = 0,
    TG_ERR_INVALID_ARG = -1,
    TG_ERR_NO_DEVICE = -2,
    TG_ERR_HIP = -3,
    TG_ERR_WEIGHTS = -4,       /* missing / wrongly sized tensor at tg_net_finalize            */
    TG_ERR_ARENA_OVERFLOW = -5,/* the MCTS node pool is exhausted (TgSearchConfig.arena_nodes)     */
    TG_ERR_NAN = -6,           /* reference panics "tried comparing nan" (search/mcts.rs:110)   */
    TG_ERR_STATE = -7,         /* call sequence violated (e.g. search before weights)          */
    TG_ERR_ILLEGAL_MOVE = -8,  /* a move in a batch failed; per-item code in the status array  */
    TG_ERR_LIMIT = -9          /* one of the fixed capacities below (TG_LIMIT_*) was exceeded      */
} TgStatus;

/* Fixed capacities of the search / self-play engine (the reference's heap structures have none).
 * Self-play (tg_selfplay_*): a game that exceeds one is RETIRED — its staged examples are discarded, its slot restarts a
 * fresh game (next generation, as after a finished game), TgSelfPlayStats.aborted_games counts it — and every other game
 * keeps running: one runaway game does not stop the other 4095.
 * Caller-driven search (tg_search_run: Player, pit): exceeding one sets a sticky error on the engine (TG_ERR_LIMIT from
 * the next polling call) until tg_search_reset / tg_search_create / tg_selfplay_create. */
#define TG_LIMIT_DEPTH 256          /* longest selection path (plies below the root)                     */
#define TG_LIMIT_VISITS (1 << 22)   /* visits + virtual visits of one node (exploration-rate table)      */
#define TG_LIMIT_GAME_PLIES 512     /* examples staged per self-play game = plies of one game            */

typedef enum TgPlayError { /* per-item result of tg_play, 0 = Ok(()) */
    TG_PLAY_OK = 0,
    TG_PLAY_OUT_OF_BOUNDS = 1,
    TG_PLAY_ALREADY_OCCUPIED = 2,
    TG_PLAY_NO_CAPSTONE = 3,
    TG_PLAY_NO_STONES = 4,
    TG_PLAY_OPENING_NON_FLAT = 5,
    TG_PLAY_EMPTY_SQUARE = 6,
    TG_PLAY_STACK_NOT_OWNED = 7,
    TG_PLAY_STACK_WALL = 8,      /* StackError::Wall  */
    TG_PLAY_STACK_CAP = 9,       /* StackError::Cap   */
    TG_PLAY_TAKE_ZERO = 10,      /* TakeError::Zero   */
    TG_PLAY_TAKE_CARRY_LIMIT = 11,
    TG_PLAY_TAKE_STACK_SIZE = 12,
    TG_PLAY_SPREAD_OUT_OF_BOUNDS = 13
} TgPlayError;

/* reference tak/src/game_result.rs:3-8 (GameResult) flattened to one byte */
typedef enum TgResult {
    TG_ONGOING = 0,
    TG_WHITE_ROAD = 1,
    TG_WHITE_FLAT = 2,
    TG_BLACK_ROAD = 3,
    TG_BLACK_FLAT = 4,
    TG_DRAW = 5,            /* Draw { reversible_plies: false } */
    TG_DRAW_REVERSIBLE = 6  /* Draw { reversible_plies: true }  */
} TgResult;

/* ---------------------------------------------------------------------------------------
 * Packed game state (replaces the non-POD `Game<N>` of reference tak/src/game.rs:24-35,
 * `Board<N>` board.rs:7-10 and `Tile` tile.rs:6-10).
 *
 * Square index sq = row * N + col (row = y, col = x, same as board.rs:24-27 data[y][x]).
 * stack[sq]: colours bottom→top, bit i = colour of the i-th stone from the bottom,
 *            0 = white, 1 = black; bits ≥ height are zero.  Max height 2·(stones+caps) ≤ 62.
 * meta[sq]:  bits 0..5 = height (number of stones), bits 6..7 = piece type of the TOP stone
 *            (0 flat, 1 wall, 2 cap); 0 when the square is empty (Tile::default()).
 * Two layouts: N ≤ 5 → TgState5 (256 B, 25 slots; for N < 5 only the first N·N are used),
 *              N = 6 → TgState6 (384 B, 36 slots).  tg_state_bytes(N) tells which.
 * Within a state the fields are arrays over squares (SoA): a 64-lane wavefront loads one
 * game with lane = square, fully coalesced.
 * ------------------------------------------------------------------------------------- */
typedef struct TgHeader {
    uint8_t n;                 /* board size 3..6                                      */
    uint8_t to_move;           /* 0 white, 1 black (Game::to_move)                     */
    uint16_t ply;
    uint8_t white_stones, white_caps, black_stones, black_caps;
    int8_t half_komi;
    uint8_t reversible_plies;
    uint8_t reserved[6];
} TgHeader; /* 16 bytes */

typedef struct TgState5 {
    uint64_t stack[25];
    uint8_t meta[25];
    uint8_t pad[15];
    TgHeader h;
} TgState5; /* 256 bytes */

typedef struct TgState6 {
    uint64_t stack[36];
    uint8_t meta[36];
    uint8_t pad[44];
    TgHeader h;
} TgState6; /* 384 bytes */

#define TG_STATE5_BYTES 256
#define TG_STATE6_BYTES 384
#define TG_META(height, top) ((uint8_t)((height) | ((top) << 6)))
#define TG_META_HEIGHT(m) ((m) & 63)
#define TG_META_TOP(m) ((m) >> 6)

/* size in bytes of one packed state for board size n (256 for n ≤ 5, 384 for n = 6) */
TG_API size_t tg_state_bytes(int n);

/* ---------------------------------------------------------------------------------------
 * Move code (replaces takparse 0.5.5 `Move{square, kind}`; reference call sites
 * tak/src/game.rs:121-125, move_gen.rs:19-75):
 *   bits  0..5   square index row*N+col
 *   bits  6..7   place: piece (0 flat, 1 wall, 2 cap);  spread: direction
 *                (0 Up=row+1 '+', 1 Down=row-1 '-', 2 Left=col-1 '<', 3 Right=col+1 '>')
 *   bits  8..15  0 for a placement; for a spread the 8-bit drop pattern, MSB first, one bit
 *                per carried stone in drop order, 1 = this stone is the last one dropped on
 *                its square (so popcount = squares covered, 8 - ctz = stones picked up).
 *                drops [2,1] → 0b0110_0000.  (takparse `Pattern::mask()` layout as used by
 *                reference alpha-tak/src/search/move_map.rs:35; see DESIGN.md "unpinned".)
 * ------------------------------------------------------------------------------------- */
typedef uint16_t TgMove;
#define TG_MAX_MOVES 512 /* capacity of one move list in tg_movegen */

/* ---------------------------------------------------------------------------------------
 * Engine
 * ------------------------------------------------------------------------------------- */
typedef struct TgEngine TgEngine;

typedef enum TgPolicyHead {
    TG_HEAD_FC5 = 0,  /* Net5: Linear(F*25 → 1575), legacy move LUT (net5.rs:56-61, move_map.rs:21-24) */
    TG_HEAD_CONV = 1  /* Net6: conv3x3(F → 3+4(2^N-2)), index ch*N²+row*N+col (net6.rs:56, move_map.rs:26-46) */
} TgPolicyHead;

typedef enum TgEvaluator {
    TG_EVAL_RESNET = 0, /* the policy/value resnet (needs weights)                           */
    TG_EVAL_DUMMY = 1,  /* DummyNet of alpha-tak/src/search/tests.rs:29-34: policy 1.0, eval 0 */
    TG_EVAL_HASH = 2    /* test evaluator: deterministic pseudo-random policy/eval from the state */
} TgEvaluator;

typedef struct TgConfig {
    int32_t abi_version;   /* TG_ABI_VERSION */
    int32_t device;        /* HIP device ordinal */
    int32_t board_size;    /* 3..6 */
    int32_t res_blocks;    /* RES_BLOCKS (net5.rs:16 / net6.rs:16), runtime here */
    int32_t filters;       /* FILTERS (net5.rs:17), multiple of 32 */
    int32_t policy_head;   /* TgPolicyHead */
    int32_t evaluator;     /* TgEvaluator */
    int32_t max_batch;     /* largest n passed to tg_policy_eval / number of concurrent games */
} TgConfig;

TG_API int tg_engine_create(const TgConfig* cfg, TgEngine** out);
TG_API void tg_engine_destroy(TgEngine* e);
TG_API const char* tg_last_error(void);
TG_API int tg_sync(TgEngine* e);
/* the hipStream_t the engine launches on (as void*), for callers timing with HIP events */
TG_API void* tg_stream(TgEngine* e);

/* Which card the engine runs on, as the HIP runtime names it — so that a multi-process launch (one rank per GPU, reference
 * train/src/self_play.rs:98,102-104: one shard per process) can show that its N ranks sat on N DISTINCT devices: two ranks that
 * were handed the same card report the same pci_bus_id. */
typedef struct TgDeviceInfo {
    int32_t hip_device;     /* TgConfig.device: the ordinal inside this process (after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) */
    int32_t cu_count;       /* hipDeviceProp_t.multiProcessorCount: 256 on an MI355X                              */
    int32_t clock_khz;      /* hipDeviceProp_t.clockRate                                                           */
    int32_t reserved;
    uint64_t total_mem;     /* bytes of device memory                                                              */
    char pci_bus_id[32];    /* hipDeviceGetPCIBusId: "domain:bus:device.function"                                   */
    char name[128];         /* hipDeviceProp_t.name                                                                 */
    char arch[64];          /* hipDeviceProp_t.gcnArchName, e.g. "gfx950:sramecc+:xnack-"                          */
} TgDeviceInfo;
TG_API int tg_device_info(TgEngine* e, TgDeviceInfo* out);

/* sizes — reference alpha-tak/src/repr/game.rs:12-15 (input_channels) and repr/moves.rs:6-31
 * (possible_moves / output_size).  policy_size depends on the head. */
TG_API int tg_input_channels(int n);
TG_API int tg_policy_size(int n, int policy_head);

/* ---------------------------------------------------------------------------------------
 * Board operators on batches (host buffers).  `states` is n packed states of
 * tg_state_bytes(board_size) each.
 * ------------------------------------------------------------------------------------- */

/* Game::possible_moves (tak/src/move_gen.rs:7-30), same order.  moves: n*TG_MAX_MOVES codes,
 * counts: n.  A position with more than TG_MAX_MOVES moves yields TG_ERR_INVALID_ARG. */
TG_API int tg_movegen(TgEngine* e, int n, const void* states, TgMove* moves, int32_t* counts);

/* Game::play (tak/src/game.rs:121-130) in place; status[i] = TgPlayError.  Returns
 * TG_ERR_ILLEGAL_MOVE if any item failed (failed items are left unchanged — the behaviour of
 * Game::safe_play, game.rs:136-145). */
TG_API int tg_play(TgEngine* e, int n, void* states, const TgMove* moves, uint8_t* status);

/* Game::result (tak/src/game.rs:220-267); results[i] = TgResult */
TG_API int tg_result(TgEngine* e, int n, const void* states, uint8_t* results);

/* game_repr (alpha-tak/src/repr/game.rs:19-51): planes n × C_in × N × N f32, NCHW like the
 * reference tensor (index c*N² + row*N + col). */
TG_API int tg_encode(TgEngine* e, int n, const void* states, float* planes);

/* move_index (alpha-tak/src/search/move_map.rs:19-48) for k moves; -1 where the reference
 * would panic ("could not map turn to index"). */
TG_API int tg_move_index(TgEngine* e, int k, const TgMove* moves, int32_t* index);

/* Example::to_tensors (alpha-tak/src/example.rs:62-78) over Symmetry (tak/src/symm.rs:7-97): the 8 dihedral
 * images of n examples.  In: n states, per example n_moves[i] (move, visits) pairs in rows of TG_MAX_MOVES.
 * Out: 8n packed states (image order of symm.rs:11-20; feed them to tg_encode for the input planes) and 8n
 * policy targets of P floats: visits/total at move_index of the transformed move, 0 elsewhere. */
TG_API int tg_augment_examples(TgEngine* e, int n, const void* states, const int32_t* n_moves, const TgMove* moves,
                        const uint32_t* visits, void* out_states, float* pi);

/* perft of tak/tests/perft.rs:3-18 evaluated on the GPU: one count per input state.
 * depth ≥ 0.  Expands level by level on the device (movegen+play+result kernels). */
TG_API int tg_perft(TgEngine* e, int n, const void* states, int depth, uint64_t* counts);

/* ---------------------------------------------------------------------------------------
 * Network (replaces Network<N>, alpha-tak/src/model/network.rs:26-35, without tch types)
 * ------------------------------------------------------------------------------------- */

/* Provide one tensor in tch/libtorch layout (conv OIHW, linear [out,in], BN vectors).
 * Names: "conv0.weight" "conv0.bias" "bn0.weight" "bn0.bias" "bn0.running_mean"
 * "bn0.running_var"; "res{i}.conv1.weight" … "res{i}.bn2.running_var" (i = 0..R-1);
 * "policy.weight" "policy.bias" (FC5: [1575, F*25]; CONV: [ch, F, 3, 3]); "value.weight"
 * "value.bias" ([1, F*N*N]).  Creation order of net5.rs:29-62 / net6.rs:29-57. */
TG_API int tg_net_set_tensor(TgEngine* e, const char* name, const float* data, size_t count);
/* Network::default() (net5.rs:29-73 / net6.rs:29-69): fill every tensor with tch's default initialisers (conv: weight
 * U(±1/sqrt(fan_in)), bias 0; BatchNorm: weight U(0,1), bias 0, mean 0, var 1; linear: weight and bias U(±1/sqrt(in))),
 * drawn from Philox(seed).  The reference draws from libtorch's global generator: same distribution, other values. */
TG_API int tg_net_init_random(TgEngine* e, uint64_t seed);
/* read back a tensor (tch layout) as last set / initialised / committed by the trainer — what Network::save writes */
TG_API int tg_net_get_tensor(TgEngine* e, const char* name, float* out, size_t count);
/* Fold BN (eval mode, eps 1e-5) into the convs, re-layout for the MFMA kernels, upload. */
TG_API int tg_net_finalize(TgEngine* e);
/* Arithmetic of the residual tower (call before tg_net_finalize).
 * TG_PRECISION_F32 (default): f32 operands on v_mfma_f32_16x16x4_f32 — exact f32 fmaf chains.
 * TG_PRECISION_BF16X3: every f32 operand is carried as hi + lo bf16 halves and a product is the sum of three bf16
 *   MFMAs (hi·hi + lo·hi + hi·lo) accumulated in f32 — 16 mantissa bits per operand, 3/16 of the f32 MFMA time.
 *   Measured deviation from the f32 forward ≤ 1e-5 relative on the policy and ≤ 3e-6 on the eval, inside the 1e-4 the
 *   reference comparison allows; supported for 5×5 (64 / 128 filters) and 6×6 (128 filters).  Heads stay f32. */
typedef enum TgPrecision { TG_PRECISION_F32 = 0, TG_PRECISION_BF16X3 = 1 } TgPrecision;
TG_API int tg_net_set_precision(TgEngine* e, int precision);

/* Network::policy_eval (net5.rs:120-130 / net6.rs:124-138): n states → policy n×P
 * (full softmax, not masked) and eval n (tanh).  n = 0 is allowed (returns TG_OK). */
TG_API int tg_policy_eval(TgEngine* e, int n, const void* states, float* policy, float* eval);

/* Network::forward_mcts (net5.rs:106-111) on already-encoded planes (n × C_in × N × N, NCHW,
 * host).  Same outputs as tg_policy_eval. */
TG_API int tg_forward_mcts(TgEngine* e, int n, const float* planes, float* policy, float* eval);

/* Device-resident variants used by bench.py: d_states / d_policy / d_eval are device
 * pointers; no synchronisation. */
TG_API int tg_policy_eval_dev(TgEngine* e, int n, const void* d_states, float* d_policy, float* d_eval);

/* ---------------------------------------------------------------------------------------
 * Search (replaces Node + Node::{virtual_rollout, devirtualize_path, select, apply_dirichlet,
 * pick_move, play}, alpha-tak/src/search/{node,mcts,noise,play}.rs) for `games` independent
 * trees stepped in lock-step, `batch` leaves per game per iteration (1 = the loop body of
 * train/src/self_play.rs:181-210; 16 = Player::rollout, player.rs:77-110).
 * ------------------------------------------------------------------------------------- */
typedef struct TgSearchConfig {
    int32_t games;            /* concurrent games (≤ cfg.max_batch)                          */
    int32_t arena_nodes;      /* AVERAGE node budget per game: all trees live in ONE pool of games × arena_nodes nodes
                                 (24 B each) handed out in chunks, so a single tree may be far larger than this while
                                 others are small.  A rollout adds one node per legal move of the expanded leaf (≈ 45 on
                                 5×5, ≈ 80 on 6×6); the subtree under the move played is kept (tree reuse) and the rest
                                 returns to the pool.  The reference's own workload — 32 games × 10 000 rollouts on 6×6 —
                                 peaks at ≈ 2^20 per game (31 M nodes, 0.75 GB in all: profiles/r02_soak_reference_workload.log); 4096 5×5 games at 400 rollouts ≈ 2^16 – 2^17.
                                 Headroom for re-rooting: on a move the kept subtree is copied into fresh chunks BEFORE the old
                                 tree's chunks are reusable (they are published at the next launch), so the pool must hold,
                                 for every game that moves in the same call, its old tree plus its kept subtree (≤ the old
                                 tree) — there is no per-game guarantee that a kept subtree fits, only the pool total.
                                 Exhausting the pool → TG_ERR_ARENA_OVERFLOW (sticky until reset).
                                 0 = auto: half of the free device memory, at most 2^22 nodes per game and 2^32 in all */
    float exploration_base;   /* EXPLORATION_BASE 500 (mcts.rs:7) */
    float exploration_init;   /* EXPLORATION_INIT 4   (mcts.rs:8) */
    uint64_t seed;            /* counter-based RNG key (noise, move sampling, openings)      */
    uint32_t slot_base;       /* global index of this engine's first game: RNG streams are keyed by
                                 (seed, slot_base + g, …) so shards on different GPUs play different,
                                 reproducible games (rank r of a sharded run passes r * games)        */
    uint32_t batch;           /* virtual rollouts per tree and iteration before the one network call — the batching of
                                 `Player` (player.rs:77-93; pit.rs BATCH_SIZE 16).  0 = 1 (self_play_parallel: one leaf per game).
                                 games × batch ≤ TgConfig.max_batch */
    int32_t visit_limit;      /* entries of the exploration-rate table = largest visits + virtual visits of one node;
                                 0 = TG_LIMIT_VISITS (16 MB).  Smaller values serve tests of the limit handling */
    int32_t reserved;
} TgSearchConfig;

TG_API int tg_search_create(TgEngine* e, const TgSearchConfig* cfg);
/* (re)start every tree as Node::default() with the given root states (host, games packed states).  The states are checked
 * on the host (board size, heights, colour bits, reserves): one no game can reach → TG_ERR_INVALID_ARG naming it. */
TG_API int tg_search_reset(TgEngine* e, const void* states);
/* run `iters` lock-step iterations (virtual_rollout → policy_eval → devirtualize_path).
 * active: optional host mask (games bytes, 0 = skip this game), NULL = all. */
TG_API int tg_search_run(TgEngine* e, int iters, const uint8_t* active);
/* Node::apply_dirichlet (noise.rs:6-16) on every active root with engine RNG
 * (stream = (seed, game, ply)). */
TG_API int tg_search_apply_dirichlet(TgEngine* e, float alpha, float ratio, const uint8_t* active);
/* Node::apply_dirichlet with caller-supplied noise (games × TG_MAX_MOVES f32, row g holds one
 * sample per child of root g) — lets a test feed the same samples to the oracle. */
TG_API int tg_search_apply_noise(TgEngine* e, const float* noise, float ratio, const uint8_t* active);
/* Node::improved_policy (play.rs:13-21) + root stats: per game the root's children in
 * possible_moves order.  moves/visits/prior/q: games × TG_MAX_MOVES; counts: games;
 * root_visits / root_q: games (any pointer may be NULL). */
TG_API int tg_search_root(TgEngine* e, TgMove* moves, uint32_t* visits, float* prior, float* q,
                   int32_t* counts, uint32_t* root_visits, float* root_q);
/* Node::play (play.rs:26-43) + Game::play: advance each active game by moves[g] with tree reuse: the chosen child's
 * subtree is copied breadth-first into fresh chunks of the shared node pool and the game's old chunks return to the pool.
 * The returned chunks become available to allocators only at the next launch, so WHILE a move is played the pool holds the
 * old trees and the kept subtrees at once (see arena_nodes: headroom). */
TG_API int tg_search_play(TgEngine* e, const TgMove* moves, const uint8_t* active);
/* current root states (games packed states) */
TG_API int tg_search_states(TgEngine* e, void* states);
/* canonical serialisation of game g's whole tree, depth-first in child order, one record per
 * initialised node: {move u16, n_children u16, visits u32, virtual u32, result u32,
 * prior f32 bits, q f32 bits}.  Returns record count through *n_records (cap = capacity). */
typedef struct TgNodeRecord {
    uint16_t move;
    uint16_t n_children;
    uint32_t visits;
    uint32_t virtual_visits;
    uint32_t result;
    uint32_t prior_bits;
    uint32_t q_bits;
} TgNodeRecord;
TG_API int tg_search_dump(TgEngine* e, int game, TgNodeRecord* records, size_t capacity, size_t* n_records);
/* counters since tg_search_create: expansions = completed rollouts (terminal ones included,
 * as in the reference's ROLLOUTS loop), evals = leaves sent to the network */
TG_API int tg_search_counters(TgEngine* e, uint64_t* expansions, uint64_t* evals);
/* occupancy of the node pool (for sizing TgSearchConfig.arena_nodes): nodes the pool holds, nodes in chunks currently owned
 * by trees, and the largest number of nodes ever owned at once since tg_search_create / tg_search_reset.  Synchronises. */
TG_API int tg_search_pool(TgEngine* e, uint64_t* nodes_total, uint64_t* nodes_in_use, uint64_t* nodes_peak);

/* ---------------------------------------------------------------------------------------
 * Self-play driver (replaces self_play_parallel, train/src/self_play.rs:96-262).
 * All compile-time constants of the reference (self_play.rs:10-19,94) are runtime here.
 * ------------------------------------------------------------------------------------- */
typedef struct TgSelfPlayConfig {
    int32_t rollouts;        /* ROLLOUTS per move (reference 10_000; BASELINE 400 / 100)     */
    int32_t noise_plies;     /* NOISE_PLIES 80   */
    int32_t exploit_plies;   /* EXPLOIT_PLIES 40 */
    float noise_alpha;       /* NOISE_ALPHA 0.2  */
    float noise_ratio;       /* NOISE_RATIO 0.3  */
    int32_t komi;            /* Game::with_komi(2) */
    int32_t total_games;     /* SELF_PLAY_GAMES: finished games are replaced until
                                completed + games ≥ total_games (self_play.rs:151,237); 0 = endless */
    int32_t max_examples;    /* capacity of the device example ring drained by tg_selfplay_drain */
    int32_t max_game_plies;  /* a game is retired (see TG_LIMIT_*) when it would stage more examples than this;
                                0 = TG_LIMIT_GAME_PLIES, the size of the per-game staging area (the largest value allowed) */
    int32_t reserved;
} TgSelfPlayConfig;

/* One training example, fixed-size record (reference alpha-tak/src/example.rs:29-33):
 * the position before the move, the visit count of every legal move in possible_moves
 * order, and the final result from the mover's perspective (self_play.rs:245-251). */
typedef struct TgExampleHeader {
    int32_t game_id;      /* global id of the game this example came from */
    int32_t n_moves;
    float result;         /* +1 / -1 / 0 from the perspective of the side to move */
    int32_t reserved;
} TgExampleHeader;

TG_API int tg_selfplay_create(TgEngine* e, const TgSearchConfig* scfg, const TgSelfPlayConfig* cfg);
/* run `plies` lock-step plies of self_play_parallel's outer loop (opening → instant-win scan
 * → noise → rollouts → pick/play/recycle).  Asynchronous; tg_sync() to wait. */
TG_API int tg_selfplay_step(TgEngine* e, int plies);
/* statistics: finished games, emitted examples, expansions, network evals */
typedef struct TgSelfPlayStats {
    uint64_t games_finished;
    uint64_t examples;
    uint64_t expansions;
    uint64_t evals;
    uint64_t plies;
    uint64_t white_wins, black_wins, draws;
    uint64_t instant_wins;
    uint64_t dropped_examples; /* finished examples overwritten in the output ring before tg_selfplay_drain fetched them
                                  (max_examples too small for the drain interval); 0 in a loss-free run */
    uint64_t aborted_games;    /* games retired because they exceeded a TG_LIMIT_* capacity (not in games_finished, no examples) */
    uint64_t alive_games;      /* slots still playing: 0 once every slot has retired (completed + games ≥ total_games) —
                                  the end of self_play_parallel's `while` loop (self_play.rs:107) */
} TgSelfPlayStats;
TG_API int tg_selfplay_stats(TgEngine* e, TgSelfPlayStats* out);
/* copy out up to `cap` finished examples (headers + states + moves + visits) and remove them
 * from the ring.  states: cap packed states; moves/visits: cap × TG_MAX_MOVES.  Examples the ring has already
 * overwritten (more than max_examples finished since the last drain) are skipped and counted in
 * TgSelfPlayStats.dropped_examples: size max_examples ≥ games × expected plies per drain interval. */
TG_API int tg_selfplay_drain(TgEngine* e, int cap, TgExampleHeader* headers, void* states, TgMove* moves,
                      uint32_t* visits, int32_t* n_out);

/* ---------------------------------------------------------------------------------------
 * Training step (replaces Network::train / train_inner, alpha-tak/src/model/network.rs:37-97, and the
 * forward_training of net5.rs:113-118 / net6.rs:111-122).  SURVEY.md §8(f) N2 / config C5.
 * The master parameters live on the device in tch layout; every chunk is augmented with the 8 symmetries
 * (Example::to_tensors), encoded, run forward with BatchNorm in training mode (batch statistics, running
 * statistics updated with momentum), and back-propagated with hand-written HIP kernels; gradients accumulate
 * over `chunks_in_step` chunks, then one Adam step (L2 weight decay, as tch's nn::Adam{wd}).
 * Data parallel: after tg_train_comm_init every optimiser step all-reduces (RCCL, sum, f32) the flat gradient
 * buffer and divides by the world size — the only collective of the whole build.
 * ------------------------------------------------------------------------------------- */
typedef struct TgTrainConfig {
    float learning_rate;    /* LEARNING_RATE 1e-4 (network.rs:14) */
    float weight_decay;     /* WEIGHT_DECAY 1e-4 (network.rs:15)  */
    float beta1, beta2, eps;/* tch Adam defaults 0.9, 0.999, 1e-8 */
    float bn_momentum;      /* tch BatchNormConfig default 0.1 */
    float bn_eps;           /* 1e-5 */
    int32_t chunk_size;     /* CHUNK_SIZE 500 examples; ×8 symmetries = positions per forward/backward */
    int32_t chunks_in_step; /* CHUNKS_IN_STEP 20 */
    int32_t reserved;
} TgTrainConfig;

/* Adam{wd}.build(vs, lr) (network.rs:40-45): creates the trainer from the tensors given to tg_net_set_tensor
 * (all of them, as for tg_net_finalize) with zero optimiser state and zero gradients. */
TG_API int tg_train_create(TgEngine* e, const TgTrainConfig* cfg);
/* train_inner (network.rs:58-97) on one chunk of n ≤ chunk_size examples (layout of tg_selfplay_drain:
 * states, n_moves, rows of TG_MAX_MOVES moves / visits; results = Example::result).  Returns the two losses
 * the reference prints (loss_p, loss_z); *stepped = 1 when this chunk completed an optimiser step.  Examples are validated on
 * the host (a reachable state, 1 ≤ n_moves ≤ TG_MAX_MOVES, at least one visit) → TG_ERR_INVALID_ARG. */
TG_API int tg_train_chunk(TgEngine* e, int n, const void* states, const int32_t* n_moves, const TgMove* moves, const uint32_t* visits,
                   const float* results, float* loss_p, float* loss_z, int32_t* stepped);
/* Network::train (network.rs:37-56): fresh Adam state and zeroed gradients, shuffle (Philox keyed by seed; the reference
 * uses thread_rng), chunks_exact(chunk_size) → tg_train_chunk each.  mean losses over the chunks are returned.  Every
 * example is validated (state, move count, visit sum — the checks of tg_train_chunk) before the first chunk, and with a
 * communicator or a reduction hook attached the ranks exchange their verdicts through it (one 16-byte all-reduce): if any
 * rank refuses its examples EVERY rank returns TG_ERR_INVALID_ARG and none trains, so an argument error cannot strand the
 * others in a collective.  All ranks must therefore call tg_train together.  tg_train_chunk validates only its own chunk — a
 * data-parallel caller of tg_train_chunk must agree on errors across ranks itself before the chunk that completes an
 * optimiser step.
 * Execution: the weight gradients of a chunk run on a stream of their own beside the data-gradient chain
 * (TG_TRAIN_ONE_STREAM=1: on the chain's stream); while chunk k runs, tg_train gathers and uploads chunk k + 1 on a copy stream and
 * enqueues it behind chunk k.  The chunks still execute one after the other: the result is that of tg_train_chunk per chunk, bit for bit. */
TG_API int tg_train(TgEngine* e, int n, const void* states, const int32_t* n_moves, const TgMove* moves, const uint32_t* visits,
             const float* results, uint64_t seed, float* mean_loss_p, float* mean_loss_z, int32_t* steps);
/* opt.step(); opt.zero_grad() now (network.rs:92-96), whatever the chunk counter says */
TG_API int tg_train_step(TgEngine* e);
/* the permutation tg_train(seed) visits n examples in (refs.shuffle, network.rs:49-50 — Philox here, thread_rng there): chunk k of
 * tg_train = examples order[k·chunk_size … (k+1)·chunk_size) handed to tg_train_chunk in that order.  Host only, no engine. */
TG_API int tg_train_order(uint64_t seed, int n, int32_t* order);
/* forward_training (net5.rs:113-118): n ≤ 8·chunk_size states → log_softmax policy (n × P) and eval (n),
 * BatchNorm on the statistics of this batch (running statistics are updated, as in libtorch). */
TG_API int tg_train_forward(TgEngine* e, int n, const void* states, float* logp, float* eval);
/* current value of a parameter / BN buffer (names of tg_net_set_tensor) and of its accumulated gradient */
TG_API int tg_train_get_tensor(TgEngine* e, const char* name, float* out, size_t count);
TG_API int tg_train_get_grad(TgEngine* e, const char* name, float* out, size_t count);
/* Intermediate tensors of the training step, for parity tests and error budgets (no counterpart in the reference; libtorch users
 * would register hooks).  tg_train_debug_read: what = "planes" (NHWC input [rows][cin_pad]); "z" / "y" (conv output / activation
 * of conv layer `layer` = 0 conv0, 1 + 2i res{i}.conv1, 2 + 2i res{i}.conv2; [rows][filters]); "mean" / "invstd" (the batch
 * statistics that layer's BatchNorm normalised with; [filters]) — all of the last forward pass; "dy" / "dz" / "dx" (gradient
 * w.r.t. y before the ReLU mask, w.r.t. z, and the data gradient handed to the layer below — for conv1 of a block with the skip
 * path's gradient added; [rows][filters]) of the layer armed with tg_train_debug_capture BEFORE the chunk (layer < 0 disarms;
 * three device copies per chunk while armed).  count = floats to read (≤ the tensor). */
TG_API int tg_train_debug_capture(TgEngine* e, int layer);
TG_API int tg_train_debug_read(TgEngine* e, const char* what, int layer, float* out, size_t count);
/* make the trained parameters the ones tg_policy_eval / search / self-play use (tg_net_set_tensor of every
 * tensor + tg_net_finalize).  With a communicator the BN running statistics are averaged over the ranks first. */
TG_API int tg_train_commit(TgEngine* e);
/* RCCL communicator for the gradient all-reduce: rank 0 calls tg_comm_unique_id (128 bytes) and hands the id
 * to every rank (any host transport); then every rank calls tg_train_comm_init. */
TG_API int tg_comm_unique_id(void* id128);
TG_API int tg_train_comm_init(TgEngine* e, int rank, int world_size, const void* id128);
/* The same reduction through a caller-supplied function instead of RCCL — ranks that share one GPU (RCCL refuses duplicate
 * devices), a host transport (gloo, MPI, a socket from Rust), or a test.  The optimiser step calls
 * fn(ctx, d_buf, count, stream) with the flat gradient buffer (device memory, `count` floats); on return — or, if fn only
 * enqueues work, in the order of `stream` (the engine's hipStream_t, tg_stream) — d_buf must hold the SUM over all world_size ranks.
 * The step then applies Adam to d_buf / world_size, so every rank that saw the same sum ends with bit-identical parameters.
 * tg_train_commit reduces the BatchNorm running statistics through the same function.  fn returns 0 or an error code
 * (→ TG_ERR_STATE).  fn = NULL removes the hook.  Mutually exclusive with tg_train_comm_init. */
typedef int (*TgAllReduceFn)(void* ctx, float* d_buf, size_t count, void* hip_stream);
TG_API int tg_train_set_allreduce(TgEngine* e, TgAllReduceFn fn, void* ctx, int world_size);
/* device address and length (floats) of the flat gradient buffer the reduction operates on: parameters in the creation
 * order of tg_net_set_tensor's names, BatchNorm buffers excluded */
TG_API int tg_train_grad_buffer(TgEngine* e, float** d_grads, size_t* count);
/* Measurement: every gradient all-reduce an optimiser step issues (RCCL or the hook) is bracketed
 * with HIP events on the engine stream; this call synchronises the stream and returns their total duration in
 * milliseconds and their number since tg_train_create.  Zero reductions on a single-rank trainer. */
TG_API int tg_train_comm_stats(TgEngine* e, double* ms_total, int64_t* reductions);

/* What is attached to the optimiser step's reduction, as the library itself sees it — so that a launch log can answer "did
 * RCCL see N ranks, and which RCCL": ncclCommCount / ncclCommUserRank of the communicator, ncclGetVersion, and the file
 * ncclAllReduce was bound from (dladdr).  One copy of RCCL per process: a librccl.so.1 the process has already mapped
 * (PyTorch's, when the host uses torch.distributed) is bound in preference to loading another (lib_was_mapped = 1). */
typedef struct TgCommInfo {
    int32_t attached;       /* 0 nothing, 1 RCCL communicator (tg_train_comm_init), 2 caller's function (tg_train_set_allreduce) */
    int32_t world_size;     /* as given to tg_train_comm_init / tg_train_set_allreduce; 1 if nothing is attached */
    int32_t rank;
    int32_t nccl_count;     /* ncclCommCount(comm), -1 without a communicator */
    int32_t nccl_rank;      /* ncclCommUserRank(comm), -1 without a communicator */
    int32_t nccl_version;   /* ncclGetVersion, -1 if librccl was never loaded */
    int32_t lib_was_mapped; /* 1: the bound librccl.so.1 was already in the process when libtakgpu first needed it */
    int32_t reserved;
    char lib_path[256];     /* the shared object ncclAllReduce resolves into; "" if librccl was never loaded */
} TgCommInfo;
TG_API int tg_train_comm_info(TgEngine* e, TgCommInfo* out);
/* Preflight of the reduction the optimiser step will use (RCCL communicator or the caller's function), before the first chunk:
 * every rank contributes 1.0f, ONE float is summed over the ranks on the engine stream, and the call returns when the result is
 * back on the host — TG_ERR_STATE if it is not world_size.  *ms = wall-clock milliseconds of that round trip (the first
 * collective on a communicator also pays RCCL's connection set-up: ring / tree construction over xGMI).  A launch that
 * cannot form a ring stops HERE, with nothing of a training step enqueued, and reads differently from a step that hangs.
 * Collective: every rank must call it.  Single-rank trainer (nothing attached): *ms = 0, returns TG_OK. */
TG_API int tg_train_comm_preflight(TgEngine* e, double* ms);

/* ---------------------------------------------------------------------------------------
 * Pit (replaces `pit`, train/src/pit.rs:15-96): the new network against the old one,
 * `pairs` openings × both colours, all 2·pairs games concurrently — one engine handle per weight set, each
 * holding one tree per game (the reference gives each side its own `Player`).  The side to move runs `rollouts`
 * iterations of `batch` virtual rollouts + one evaluation (reference: ROLLOUTS 50 × Player::rollout with BATCH_SIZE 16),
 * the waiting side `idle_rollouts` iterations (the batch a `Player` keeps in flight, player.rs:65-66), moves are
 * pick_move(exploit = true).  Openings: a1, a random far corner, `random_plies` random Flat/Cap placements
 * (pit.rs:33-63), drawn from Philox(seed).  All games run at once, so the early exit of pit.rs:20-23 cannot save any
 * work; its effect on the counts is reproduced in the ref_* fields (the tallies of the openings the reference's loop would
 * have played before breaking, in its order).  Not reproduced: the exact interleaving of the waiting player's stale batch
 * with the moves.  The caller applies the gate
 * (main.rs:102: win_rate > 0.55).  Both engines' search / self-play state is replaced (tg_search_create is called on
 * each); 2·pairs·batch ≤ max_batch of both engines.
 * ------------------------------------------------------------------------------------- */
typedef struct TgPitConfig {
    int32_t pairs;          /* PIT_GAMES 128 */
    int32_t rollouts;       /* ROLLOUTS 50 iterations per move */
    int32_t idle_rollouts;  /* 1 */
    int32_t random_plies;   /* RANDOM_PLIES 2 */
    int32_t komi;           /* Game::with_komi(2) */
    int32_t max_plies;      /* safety cap on the game length, 0 = none */
    int32_t arena_nodes;    /* per-game tree arena of both engines, 0 = 16384 */
    int32_t batch;          /* BATCH_SIZE 16 virtual rollouts per iteration; 0 = 1 */
    uint64_t seed;
} TgPitConfig;
typedef struct TgPitResult {
    uint32_t wins, losses, draws; /* from the new network's point of view (PitResult, pit.rs:98-103) */
    uint32_t unfinished;          /* games cut by max_plies */
    uint32_t plies;               /* lock-step plies played */
    uint32_t reserved;
    double win_rate;              /* wins / (wins + losses), pit.rs:105-110 */
    /* The same with pit.rs:20-23 applied: openings are counted in order (White game, then Black game of each) until
     * wins > pairs + pairs/10 or losses > pairs - pairs/10 holds before an opening — what `pit` returns. */
    uint32_t ref_wins, ref_losses, ref_draws;
    uint32_t ref_pairs;           /* openings counted (= pairs when the loop never breaks) */
    double ref_win_rate;
} TgPitResult;
TG_API int tg_pit(TgEngine* e_new, TgEngine* e_old, const TgPitConfig* cfg, TgPitResult* out);

/* ---------------------------------------------------------------------------------------
 * Text formats at the edge of the path (host only).  PTN moves / TPS positions follow takparse 0.5.5's
 * Display + FromStr as used by tak/src/game.rs:79 and tak/src/tps.rs:7-96; the example line follows
 * alpha-tak/src/example.rs:81-133 ("{tps};{w_stones};{w_caps};{b_stones};{b_caps};{half_komi};{result};
 * {move:visits,…}"), so drained examples can be written as the reference's `_examples/{time}.data` files.
 * format functions return the text length (≥ 0) or a negative TgStatus.
 * ------------------------------------------------------------------------------------- */
TG_API int tg_format_move(int n, TgMove mv, char* buf, size_t cap);
TG_API int tg_parse_move(int n, const char* text, TgMove* out);
TG_API int tg_format_tps(int n, const void* state, char* buf, size_t cap);
TG_API int tg_parse_tps(int n, const char* text, void* state); /* reserves derived from the board, half_komi 0 */
TG_API int tg_format_example(int n, const void* state, int n_moves, const TgMove* moves, const uint32_t* visits, float result,
                      char* buf, size_t cap);
TG_API int tg_parse_example(int n, const char* line, void* state, int cap_moves, TgMove* moves, uint32_t* visits,
                     int32_t* n_moves, float* result);

/* ---------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart: the reference has no profiling).
 * While enabled, every `sample_every`-th network forward brackets each residual-tower 3×3 conv
 * launch with HIP events on the engine stream; tg_profile_read synchronises and returns the totals.
 * ------------------------------------------------------------------------------------- */
typedef struct TgProfile {
    uint64_t conv_launches;   /* residual-tower conv launches timed (F→F 3×3, the dominant kernel) */
    double conv_ms;           /* their summed duration                                           */
    uint64_t forwards;        /* network forwards timed end to end                               */
    double forward_ms;
    int64_t conv_rows;        /* M = positions × N² of the timed launches (all equal)            */
    int64_t conv_flops;       /* algorithmic FLOPs of one timed launch: 2·M·9·F·F (per-layer path) or, for the fused
                                 tower, 2·M·9·(C_in·F + 2R·F·F) — every input plane counted as data                */
    int64_t conv_flops_executed; /* FLOPs of the MFMAs the timed launch issues: less than conv_flops when the fused tower
                                 takes the per-position constant input planes (reserves, colour, fcd) as a bias and runs
                                 layer 0 over the board planes only — price the MFMA pipe against THIS figure       */
} TgProfile;
TG_API int tg_profile_enable(TgEngine* e, int sample_every); /* 0 disables */
/* Board-path micro-benchmark: uploads n states + one legal move each, then runs `reps`
 * fused passes (Game::play → Game::result → possible_moves count → game_repr planes, all on the device,
 * inputs resident in HBM) bracketed by HIP events.  Returns the average pass time; out_* (host, optional)
 * receive the outputs of the last pass: stepped states, results, move counts. */
TG_API int tg_board_pass_bench(TgEngine* e, int n, const void* states, const TgMove* moves, int reps, double* avg_ms,
                        void* out_states, uint8_t* out_results, int32_t* out_counts);
TG_API int tg_profile_read(TgEngine* e, TgProfile* out);     /* synchronises; resets the totals */

/* A/B switches.  The library reads a fixed set of TG_* environment variables (every one is read through ONE
 * function: a switch is on when the variable is set to anything but "" or "0"; the two numeric ones — TG_TOWER_VARIANT,
 * TG_WGRAD_PW — take their value, 0 = off).  tg_debug_switches writes the switches that are ON in this process's
 * environment as "NAME=value NAME=value …" into buf (always NUL-terminated, truncated to cap) and returns their number:
 * 0 on a measured run — bench.py prints the list as config.switches_set.  Needs no engine and no GPU. */
TG_API int tg_debug_switches(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TAKGPU_H */
