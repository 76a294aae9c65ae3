// search.cuh — device-resident MCTS data layout shared by search_kernels.hip and search.hip.
//
// Replaces the heap tree of reference alpha-tak/src/search/node.rs:3-39 (Node { policy,
// expected_reward, result, visits, virtual_visits, children: Box<[(Move, Node)]> }).
// A node is two records so that PUCT selection streams only the hot one:
//   NodeHot  (16 B)  prior, q (= expected_reward), visits, virtual visits   — read per child per level
//   NodeCold ( 8 B)  first-child index, move code, n_children | result<<12   — read for the chosen child
// The children of a node are contiguous (same order as Game::possible_moves), so a wave scanning them
// issues fully coalesced 16-byte-per-lane loads.
//
// Memory: ONE node pool shared by all games, cut into chunks of 2^chunk_shift nodes.  A game bump-allocates the
// children blocks of its expansions inside its open chunk and takes a new chunk from the pool's free ring when a
// block does not fit (one atomic per ≈ 2^chunk_shift / branching expansions instead of one per expansion, and none
// on a per-game counter).  So a tree is as large as the pool allows — a 10 000-rollout search of 32 games and a
// 400-rollout search of 4096 games run on the same allocation, sized by the SUM of the trees, not by games × the
// largest tree.  On a move (tree reuse, search/play.rs:26-43) the kept subtree is copied breadth-first into fresh
// chunks — the copy is its own work queue (Cheney), so no depth or width limit — and the old chunks go back to
// the ring (published to allocators at the next kernel boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

struct NodeHot {
    float prior, q;
    uint32_t visits, virt;
};
struct NodeCold {
    uint32_t child;   // pool index of the first child (0 = no children; chunk 0 is never handed out)
    uint16_t mv;      // move that leads here (TgMove)
    uint16_t nres;    // n_children (12 bits) | TgResult << 12
};
static_assert(sizeof(NodeHot) == 16 && sizeof(NodeCold) == 8, "node records");

constexpr int MAX_DEPTH = 256;       // longest selection path kept per game
constexpr int EX_MOVES = 512;        // = TG_MAX_MOVES

enum : uint32_t {
    ERRF_ARENA = 1u,     // node pool exhausted
    ERRF_NAN = 2u,       // NaN upper confidence bound (reference panics, mcts.rs:110)
    ERRF_DEPTH = 4u,     // selection path longer than MAX_DEPTH
    ERRF_CTAB = 8u,      // visit count beyond the exploration-rate table
    ERRF_MOVE = 16u,     // move not among the root's children / unmapped policy index
    ERRF_EXAMPLES = 32u, // a game produced more examples than its staging area holds
    ERRF_MOVES = 64u,    // a position with more than TG_MAX_MOVES legal moves
    ERRF_PICK = 128u     // WeightedIndex over all-zero visits (reference panics, play.rs:62)
};

// one training example as staged / emitted on the device
struct ExampleRec {
    int32_t slot;        // global slot id
    int32_t generation;  // index of the game played in that slot
    int32_t n_moves;
    float result;
};

struct SearchDev {
    // trees: one pool of n_chunks << chunk_shift nodes
    NodeHot* hot;        // [pool]
    NodeCold* cold;      // [pool]
    uint32_t* root;      // [G] pool index of the game's root
    uint32_t* alloc;     // [G][2] next free node and end of the game's open chunk
    uint32_t* chunk_head;// [G] the game's most recently taken chunk; its chunks are chained through chunk_link
    uint32_t* chunk_link;// [n_chunks] the chunk the same game took before this one (0 = none)
    uint32_t* chunk_fwd; // [n_chunks] re-root copy only: the chunk taken after this one …
    uint32_t* chunk_used;// [n_chunks] … and how many of this chunk's nodes were filled when it was closed
    uint32_t* free_ring; // [n_chunks] ids of free chunks
    unsigned long long* pool_ctl;  // [0] chunks taken, [1] chunks returned, [2] returned-and-published (visible to takers), [3] peak owned
    uint32_t n_chunks;
    int chunk_shift;
    // games
    uint8_t* root_state; // [G][state bytes]
    uint8_t* alive;      // [G]
    uint8_t* abort;      // [G] self-play: ERRF_* bits of the capacity this game has exceeded (retired at the end of the ply)
    uint32_t* generation;// [G]
    // per-iteration leaf hand-off
    int32_t* path_len;   // [G·batch]  (every per-leaf array below has G·batch entries, slot = g·batch + pass)
    uint32_t* path;      // [G·batch][MAX_DEPTH]
    uint8_t* leaf_kind;  // [G·batch] 0 skipped, 1 needs evaluation, 2 terminal (already backed up)
    uint32_t* leaf_rec;  // [G·batch][2] children block and child count of the expanded leaf (written by the select: the backup
                         // does not walk path → leaf → children)
    uint16_t* child_pidx;// [G·batch][EX_MOVES] policy index of every child of the expanded leaf (move_index at expansion time,
                         // 0xFFFF = unmapped): the backup gathers logits by it instead of re-deriving it from the moves
    uint64_t* leaf_hash; // [G] (TG_EVAL_HASH)
    float* planes;       // [G][nsq][cin_pad] NHWC network input (null when the tower encodes from leaf_state)
    uint8_t* leaf_state; // [G][state bytes] packed leaf positions
    float* policy;       // [G][P]
    const float* logits; // FC head: [G·batch][logit_ld] policy logits with the value pre-activation in column P; when set the
                         // backup takes softmax and tanh itself (softmax.cuh) and `policy` / `eval` are not produced
    int logit_ld;
    const float* fc_stats; // with `logits`: per leaf and column block {max, Σexp(x − max)} emitted by the policy FC (softmax.cuh);
    int fc_blocks;         // null → the backup reduces the whole logits row itself
    int fc_stride;         // pairs per leaf in fc_stats (the exact-f32 FC appends {value pre-activation, 0} behind the blocks)
    const float* child_logit; // [G·batch][EX_MOVES] logit of every child of the expanded leaf, written by the policy FC's epilogue from
                           // child_pidx (exact-f32 FC at full batches, round 4): then the FC writes no logits rows, the backup reads
                           // this and the statistics record (value pre-activation = pair fc_blocks) and nothing of `logits`
    float* eval;         // [G]
    // constants
    const float* ctab;   // exploration_rate(n) for integer n (host logf, mcts.rs:10-12)
    const int16_t* lut5;
    uint32_t* err;       // error flag word
    unsigned long long* counters;  // [G][2] per game: expansions, evals (summed on read; a single shared word would
                                   // serialise 2·G same-address atomics per iteration, ≈ 11 ns each)
    int G, n, cin_pad, P, ctab_size, legacy5, evaluator;
    int batch, pass;     // virtual rollouts per tree and iteration; the one this launch performs (leaf slot = g·batch + pass)
    int retire;          // 1 = a game past a capacity is retired on its own (self-play); 0 = sticky engine error (caller-driven search)
    uint32_t slot_base;
    uint64_t seed;
};

struct SelfPlayDev {
    // staging of the current game's examples per slot
    ExampleRec* st_hdr;   // [G][ex_per_game]
    uint8_t* st_state;    // [G][ex_per_game][bytes]
    uint16_t* st_moves;   // [G][ex_per_game][EX_MOVES]
    uint32_t* st_visits;  // [G][ex_per_game][EX_MOVES]
    int32_t* st_count;    // [G]
    // output ring
    ExampleRec* out_hdr;  // [max_examples]
    uint8_t* out_state;
    uint16_t* out_moves;
    uint32_t* out_visits;
    // phase hand-off
    uint8_t* fin;         // [G] TgResult of a game that just ended (0 = still running)
    uint8_t* recycle;     // [G]
    uint32_t* out_off;    // [G] first ring slot for this game's examples
    int32_t* chosen;      // [G] child ordinal picked this ply (-1 none)
    uint8_t* mask;        // [G] scratch mask (noise phase)
    unsigned long long* stats;  // games_finished, examples, plies, white, black, draws, instant_wins
    int ex_per_game, max_examples;
    int max_game_plies;   // ≤ ex_per_game: a game that would stage more examples is retired
    int rollouts, noise_plies, exploit_plies, komi, total_games;
    float noise_alpha, noise_ratio;
};

enum { ST_FINISHED = 0, ST_EXAMPLES = 1, ST_PLIES = 2, ST_WHITE = 3, ST_BLACK = 4, ST_DRAWS = 5, ST_INSTANT = 6, ST_ABORTED = 7, ST_COUNT = 8 };
constexpr uint8_t FIN_ABORTED = 0x80;  // SelfPlayDev.fin: not a TgResult — the game exceeded a capacity and is retired without examples

}  // namespace tg
