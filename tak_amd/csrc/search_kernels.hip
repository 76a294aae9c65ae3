// search_kernels.hip — GPU-resident MCTS (one wavefront per game) and the self-play driver kernels.
//
// Replaces reference alpha-tak/src/search/mcts.rs (virtual_rollout :26-65, select :94-118,
// devirtualize_path :67-91, update_concrete :120-124), search/noise.rs, search/play.rs and the
// per-ply phases of train/src/self_play.rs:108-259.  The float arithmetic of PUCT and of the value
// backup is written in the reference's operation order and compiled with -ffp-contract=off, the
// exploration rate comes from a host-computed table over the (integer) visit count, and both argmaxes
// keep the LAST maximum like Iterator::max_by / max_by_key, so that search traces are reproducible bit
// for bit against a scalar CPU statement of the same algorithm.
#include <stdlib.h>

#include "board.cuh"
#include "kernels.h"
#include "rng.cuh"
#include "search.cuh"
#include "softmax.cuh"

namespace tg {

#ifndef TG_WPB
#define TG_WPB 4
#endif
constexpr int WPB = TG_WPB;  // waves (games) per block (4: 256 threads; other values: scripts/probes/tree_wpb_probe.sh)

__device__ inline int game_of_wave() { return (int)(blockIdx.x * WPB + (threadIdx.x >> 6)); }
__device__ inline void flag(const SearchDev& S, uint32_t bit) { atomicOr(S.err, bit); }
// A game ran into one of the fixed capacities (TG_LIMIT_*, takgpu.h).  Self-play retires that game alone — the bit is kept
// per game, its wave stops touching the tree, and the end of the ply discards its examples and restarts the slot;
// a caller-driven search has nobody to restart the game, so the engine-wide sticky error stays.  Called by the whole wave.
__device__ inline void limit_hit(const SearchDev& S, int g, uint32_t bit) {
    if (S.retire) { if (lane_id() == 0) S.abort[g] = (uint8_t)(S.abort[g] | bit); }
    else flag(S, bit);
}
__device__ inline void wave_sync_mem() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Diagnostic build only (-DTG_TREE_STAMPS, scripts/probes/tree_stamps.py): s_memtime stamps of the waves of the first 64 games
// at the phase boundaries of k_backup_select; tg_debug_tree_stamps copies them out.  The product build compiles none of it.
#ifdef TG_TREE_STAMPS
__device__ unsigned long long g_tree_stamps[64][32];
#define TG_TSTAMP(g, slot)                                                                                    \
    do {                                                                                                      \
        if ((g) < 64 && (slot) < 32 && (threadIdx.x & 63) == 0) {                                             \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
            g_tree_stamps[(g)][(slot)] = __builtin_amdgcn_s_memtime();                                        \
        }                                                                                                     \
    } while (0)
#else
#define TG_TSTAMP(g, slot) do { } while (0)
#endif

// ---- node pool (search.cuh): chunks of 2^chunk_shift nodes handed out from a ring of free chunk ids ----
// Take one chunk for the calling wave (wave-uniform result, 0 = pool exhausted).  Only chunks whose return was
// published before this kernel started are handed out, so a taker never reads a ring slot that a concurrent
// pool_give of the same launch has reserved but not yet written.
__device__ inline uint32_t pool_take(const SearchDev& S) {
    uint32_t c = 0;
    if (lane_id() == 0) {
        const unsigned long long h = atomicAdd(&S.pool_ctl[0], 1ull);
        if (h < S.pool_ctl[2]) c = S.free_ring[h % S.n_chunks];
    }
    return uni((uint32_t)__shfl((int)c, 0));
}
// Return every chunk of a game's chain (lane 0 walks it; a chain is a handful to a few hundred chunks, once per move)
__device__ inline void pool_give_chain(const SearchDev& S, uint32_t head) {
    if (lane_id() != 0) return;
    for (uint32_t c = head; c != 0;) {
        const uint32_t next = S.chunk_link[c];
        const unsigned long long t = atomicAdd(&S.pool_ctl[1], 1ull);
        S.free_ring[t % S.n_chunks] = c;
        c = next;
    }
}
// after a kernel that returned chunks: make them available to the following launches
// (also the high-water mark of chunks owned by trees, sampled here — after every re-root, when the trees are smallest — and
// therefore taken BEFORE the returns of this launch are counted: the occupancy just before the move was played)
__global__ void k_pool_publish(SearchDev S) {
    const unsigned long long owned = S.pool_ctl[0] - (S.pool_ctl[2] - (unsigned long long)(S.n_chunks - 1));
    if (owned > S.pool_ctl[3]) S.pool_ctl[3] = owned;
    S.pool_ctl[2] = S.pool_ctl[1];
}

__device__ inline uint64_t ws_hash(const WState& s, const Geom& g) {
    // same function of the packed bytes as the CPU statement: stack words, meta bytes, 10 header bytes
    uint64_t h = 0x243F6A8885A308D3ull;
    for (int i = 0; i < g.nsq; i++) h = mix64(h ^ shfl64(s.stack, i)) + (uint64_t)i;
    for (int i = 0; i < g.nsq; i++) {
        uint32_t hh = (uint32_t)__shfl((int)s.height, i), tp = (uint32_t)__shfl((int)s.top, i);
        uint64_t meta = hh | ((hh ? tp : 0u) << 6);
        h = mix64(h ^ (meta << 8) ^ (uint64_t)(i + 1));
    }
    uint64_t a = (uint64_t)g.n | ((uint64_t)s.to_move << 8) | ((uint64_t)(s.ply & 0xffff) << 16) | ((uint64_t)s.ws << 32) |
                 ((uint64_t)s.wc << 40) | ((uint64_t)s.bs << 48) | ((uint64_t)s.bc << 56);
    uint64_t b = ((uint64_t)s.half_komi & 0xff) | ((uint64_t)(s.rev & 0xff) << 8);
    h = mix64(h ^ a);
    h = mix64(h ^ b);
    return h;
}

// update_concrete, mcts.rs:120-124
__device__ inline void update_concrete(NodeHot& h, float reward) {
    float cumulative = h.q * (float)h.visits;
    h.visits += 1;
    h.q = (cumulative + reward) / (float)h.visits;
}

// ------------------------------------------------------------------------------------------------
// virtual_rollout (+ select): descend, expand the first uninitialised node, mark the path with a
// virtual visit (or back a concrete result up when the rollout ends on a terminal node), and leave
// the leaf encoded in the network input batch.
// ------------------------------------------------------------------------------------------------
// NB: board size as a compile-time constant (5, 6; 0 = read S.n).  With n known the geometry masks fold to immediates and the
// many divisions by n / n² of move generation, play and the encoder become multiply-shifts instead of ≈ 20-instruction
// software divisions — the tree kernels are bound by vector issue, so this is time.
// What the fused kernel loads of the root BEFORE the backup runs (none of it is written by the backup), so that the select does
// not start with a chain of dependent loads: the root's index, packed state and cold record; `hot_known`: the backup also handed
// over the root's (visits, virtual) as it left them.
struct RootPre {
    bool on = false, hot_known = false;
    uint32_t root = 0, vis = 0, vv = 0;
    NodeCold cold;
    WRaw raw;  // the packed root position as requested (ws_load_raw); unpacked by the select
    uint32_t alive_v = 1, abort_v = 0;  // S.alive[g], S.abort[g] (neither changes between the request and the select)
};

template <int NB>
__device__ __forceinline__ void select_pass(const SearchDev& S, const uint8_t* __restrict__ active, const int g, const int pass,
                                            uint32_t* path, uint16_t* mvl, const RootPre& pre = RootPre()) {
    const int lane = lane_id();
    // leaf slot of this pass: `batch` virtual rollouts per tree and iteration (Player's batching, player.rs:77-93), pass p
    // writing slot g·batch + p
    const size_t slot = (size_t)g * (size_t)S.batch + (size_t)pass;
    {
        // (both flags and the game's allocation cursor are requested together; read one after the other behind `||` they were
        // two round trips in a row in front of the descent)
        const uint32_t alive_v = pre.on ? pre.alive_v : (uint32_t)S.alive[g];
        const uint32_t abort_v = pre.on ? pre.abort_v : (uint32_t)S.abort[g];
        if (!uni(alive_v) || (active && !active[g]) || (S.retire && uni(abort_v))) {
            if (lane == 0) S.leaf_kind[slot] = 0;
            return;
        }
    }
    // the open chunk of the game's node allocation, needed only if this descent expands a leaf: requested now, it is there by then
    const uint32_t alloc_a = S.alloc[2 * g], alloc_e = S.alloc[2 * g + 1];
    const Geom geo = make_geom(NB ? NB : S.n);
    WState s;
    if (pre.on) ws_unpack(s, pre.raw, geo);
    else ws_load(s, S.root_state + (size_t)g * geo.bytes, geo);
    NodeHot* hot = S.hot;
    NodeCold* cold = S.cold;
    const uint32_t root = pre.on ? pre.root : uni(S.root[g]);
    const uint32_t root_color = s.to_move;
    uint32_t node = root;
    int depth = 0;
    uint32_t res = TG_ONGOING;
    bool terminal = false;

    // The (visits, virtual, child, n|result) of the node being visited travel in registers: they are read
    // once for the root and afterwards come out of the children scan of the level above, so each level
    // costs ONE dependent memory round trip (the coalesced hot + cold records of all children).
    uint32_t vis, vv, nres, cbase;
    {
        NodeCold nc = pre.on ? pre.cold : cold[root];
        nres = uni((uint32_t)nc.nres); cbase = uni(nc.child);
        if (pre.on && pre.hot_known) { vis = pre.vis; vv = pre.vv; }
        else { NodeHot nh = hot[root]; vis = uni(nh.visits); vv = uni(nh.virt); }
    }
    TG_TSTAMP(g, 4);  // root state + root record loaded
    // The records of the first 64 children of the node about to be visited are requested as soon as its children block is
    // known — for the root here, for every later level right before the move is played on the wave's position — so that the
    // round trip of the children scan passes under ws_play instead of after it.
    NodeHot pf_h, pf_h2;   // children lane and lane + 64 (the opening's 70-odd placements do not fit one round of the scan)
    NodeCold pf_c, pf_c2;
    pf_h.prior = 0.0f; pf_h.q = 0.0f; pf_h.visits = 0; pf_h.virt = 0;
    pf_c.child = 0; pf_c.mv = 0; pf_c.nres = 0;
    pf_h2 = pf_h; pf_c2 = pf_c;
    float c_pf = 0.0f;  // exploration_rate(visits + virtual) of the node about to be visited (mcts.rs:10-12), from the table
    auto prefetch_children = [&]() {
        // (a vector load of one address: it returns with the children's records instead of on the scalar path in front of them)
        const uint32_t tn = vis + vv;
        c_pf = S.ctab[tn < (uint32_t)S.ctab_size ? tn : (uint32_t)S.ctab_size - 1u];
        if ((vis | vv) != 0u && (nres >> 12) == TG_ONGOING && (uint32_t)lane < (nres & 0xfffu)) {
            pf_h = hot[cbase + (uint32_t)lane];
            pf_c = cold[cbase + (uint32_t)lane];
            if ((uint32_t)lane + 64u < (nres & 0xfffu)) {
                pf_h2 = hot[cbase + (uint32_t)lane + 64u];
                pf_c2 = cold[cbase + (uint32_t)lane + 64u];
            }
        }
    };
    prefetch_children();
    for (;;) {
        if (vis == 0 && vv == 0) {
            // uninitialised node: initialise it and stop (mcts.rs:41-53)
            TG_TSTAMP(g, 24);  // descent done
            res = ws_result(s, geo);
            TG_TSTAMP(g, 25);
            uint32_t count = 0, cb = 0;
            if (res == TG_ONGOING) {
                // the legal moves are staged in LDS so that the children block can be placed once its size is known
                count = (uint32_t)ws_movegen(s, geo, EX_MOVES, [&](int idx, uint32_t code) { mvl[idx] = (uint16_t)code; });
                TG_TSTAMP(g, 26);
                if (count > (uint32_t)EX_MOVES) {
                    limit_hit(S, g, ERRF_MOVES);
                    if (lane == 0) S.leaf_kind[slot] = 0;
                    return;
                }
                uint32_t a = uni(alloc_a), end = uni(alloc_e);
                if (a + count > end) {  // the block does not fit into the game's open chunk: take the next one
                    const uint32_t c = pool_take(S);
                    if (c == 0) {
                        flag(S, ERRF_ARENA);
                        if (lane == 0) S.leaf_kind[slot] = 0;
                        return;
                    }
                    a = c << S.chunk_shift;
                    end = a + (1u << S.chunk_shift);
                    if (lane == 0) {
                        S.chunk_link[c] = S.chunk_head[g];
                        S.chunk_head[g] = c;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float temp_policy = 1.0f / (float)count;
                bool bad = false;
                for (uint32_t i = lane; i < count; i += 64) {
                    NodeCold cc;
                    cc.child = 0; cc.mv = mvl[i]; cc.nres = 0;
                    cold[a + i] = cc;
                    NodeHot c;
                    c.prior = temp_policy; c.q = 0.0f; c.visits = 0; c.virt = 0;
                    hot[a + i] = c;
                    // the child's policy index now, while its move is at hand (move_index, move_map.rs:19-48): the backup of this
                    // leaf gathers the network's output by it
                    const int idx = move_index_dev((uint32_t)mvl[i], NB ? NB : S.n, S.legacy5 != 0, S.lut5);
                    const bool ok = idx >= 0 && idx < S.P;
                    bad |= !ok;
                    S.child_pidx[slot * EX_MOVES + i] = ok ? (uint16_t)idx : (uint16_t)0xFFFF;
                }
                if (__ballot(bad)) flag(S, ERRF_MOVE);  // "could not map turn to index" (move_map.rs:24)
                cb = a;
                if (lane == 0) { S.alloc[2 * g] = a + count; S.alloc[2 * g + 1] = end; }
            }
            if (lane == 0) {
                cold[node].child = cb;
                cold[node].nres = (uint16_t)(count | (res << 12));
                S.leaf_rec[2 * slot] = cb;
                S.leaf_rec[2 * slot + 1] = count;
            }
            terminal = res != TG_ONGOING;
            TG_TSTAMP(g, 27);  // children created
            break;
        }
        res = nres >> 12;
        if (res != TG_ONGOING) { terminal = true; break; }  // known terminal node: same result again (mcts.rs:35-38)
        // ---- select, mcts.rs:94-118 ----
        const uint32_t nchild = nres & 0xfffu;
        const uint32_t nsum = vis + vv;
        const float visit_count = (float)nsum;
        uint32_t ti = nsum;
        if ((int)ti >= S.ctab_size) {
            limit_hit(S, g, ERRF_CTAB);
            if (S.retire) {  // nothing of this rollout has touched the tree yet (virtual visits are marked in the unwind)
                if (lane == 0) S.leaf_kind[slot] = 0;
                return;
            }
            ti = (uint32_t)S.ctab_size - 1;
        }
        const float c_rate = c_pf;  // = S.ctab[ti], requested with the children
        const float root_n = sqrtf(visit_count);
        float best = -INFINITY;
        int best_i = -1;
        bool nan = false;
        NodeHot bh;      // records of this lane's best child
        NodeCold bc;
        bh.prior = 0.0f; bh.q = 0.0f; bh.visits = 0; bh.virt = 0;
        bc.child = 0; bc.mv = 0; bc.nres = 0;
        for (uint32_t i = lane; i < nchild; i += 64) {
            NodeHot ch;
            NodeCold cc;
            if (i == (uint32_t)lane) { ch = pf_h; cc = pf_c; }  // the first 128 children were requested a level ago
            else if (i == (uint32_t)lane + 64u) { ch = pf_h2; cc = pf_c2; }
            else { ch = hot[cbase + i]; cc = cold[cbase + i]; }
            float cn = (float)(ch.visits + ch.virt);
            float qv = (ch.visits | ch.virt) ? (ch.q * (float)ch.visits - (float)ch.virt) / cn : 0.0f;
            float u = qv + c_rate * ch.prior * (root_n / (1.0f + cn));
            if (u != u) nan = true;
            if (u >= best) { best = u; best_i = (int)i; bh = ch; bc = cc; }
        }
        // the wave's best (value, index): the largest pair under (value, then index) — `max_by` keeps the LAST maximum
        // (mcts.rs:107-117).  Six DPP steps (row_shr 1, 2, 4, 8, row_bcast15, row_bcast31: an inclusive max-scan whose lane 63
        // holds the total) instead of six butterfly rounds of two ds_bpermute each — 12 trips through the LDS crossbar, one after
        // the other, at every level of every descent.  The order of a total order's maximum does not matter: same winner.
        float wb = best;
        int wi = best_i;
#define TG_ARGMAX_STEP(CTRL, ROWS)                                                                                               \
        {                                                                                                                        \
            const float ob = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(wb), CTRL, ROWS, 0xf, false)); \
            const int oi = __builtin_amdgcn_update_dpp(-1, wi, CTRL, ROWS, 0xf, false);                                        \
            if (ob > wb || (ob == wb && oi > wi)) { wb = ob; wi = oi; }                                                          \
        }
        TG_ARGMAX_STEP(0x111, 0xf) TG_ARGMAX_STEP(0x112, 0xf) TG_ARGMAX_STEP(0x114, 0xf) TG_ARGMAX_STEP(0x118, 0xf)
        TG_ARGMAX_STEP(0x142, 0xa) TG_ARGMAX_STEP(0x143, 0xc)
#undef TG_ARGMAX_STEP
        wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wb), 63));
        wi = __builtin_amdgcn_readlane(wi, 63);
        if (__ballot(nan)) flag(S, ERRF_NAN);
        if (wi < 0) {  // cannot happen for a consistent tree; never index out of the arena
            flag(S, ERRF_NAN);
            if (lane == 0) S.leaf_kind[slot] = 0;
            return;
        }
        // the winning lane (the one whose own best is the wave's best) hands its child's records down
        const int src = __builtin_ctzll(__ballot(best_i == wi));
        const uint32_t chosen = cbase + (uint32_t)wi;
        const uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bc.mv, src);  // (v_readlane: no LDS round trip)
        vis = (uint32_t)__builtin_amdgcn_readlane((int)bh.visits, src);
        vv = (uint32_t)__builtin_amdgcn_readlane((int)bh.virt, src);
        nres = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bc.nres, src);
        cbase = (uint32_t)__builtin_amdgcn_readlane((int)bc.child, src);
        TG_TSTAMP(g, 5 + 2 * (depth < 9 ? depth : 9));  // children scanned, best child known
        prefetch_children();  // of the chosen child, under the play of its move
        ws_play(s, mv, geo);
        TG_TSTAMP(g, 6 + 2 * (depth < 9 ? depth : 9));  // move played
        if (depth >= MAX_DEPTH) {
            limit_hit(S, g, ERRF_DEPTH);
            if (lane == 0) S.leaf_kind[slot] = 0;
            return;
        }
        if (lane == 0) path[depth] = chosen;
        depth++;
        node = chosen;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- unwind (mcts.rs:55-62): lane d handles the node at depth d ----
    const bool winner = res >= TG_WHITE_ROAD && res <= TG_BLACK_FLAT;
    const uint32_t wcolor = (res == TG_WHITE_ROAD || res == TG_WHITE_FLAT) ? 0u : 1u;
    for (int d = lane; d <= depth; d += 64) {
        uint32_t nd = d == 0 ? root : path[d - 1];
        if (!terminal && S.batch == 1) {
            // virtual_visits += 1 as a returnless atomic: nothing to wait for (the record's load and store were a round trip
            // in front of the leaf's stores); the tree belongs to this wave alone.  Only with one rollout per launch: the atomic
            // is performed in L2, and a second pass of the same wave would read the record through its L1
            __hip_atomic_fetch_add(&hot[nd].virt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            NodeHot h = hot[nd];
            if (terminal) {
                uint32_t curr = root_color ^ (uint32_t)(d & 1);
                float reward = winner ? (wcolor == curr ? -1.0f : 1.0f) : 0.0f;
                update_concrete(h, reward);
            } else {
                h.virt += 1;
            }
            hot[nd] = h;
        }
    }
    TG_TSTAMP(g, 28);  // virtual visits marked
    uint32_t* gpath = S.path + slot * MAX_DEPTH;
    for (int d = lane; d < depth; d += 64) gpath[d] = path[d];
    if (!terminal) {
        if (S.evaluator == TG_EVAL_RESNET) {
            if (S.planes) ws_encode<true>(s, geo, S.planes + slot * geo.nsq * S.cin_pad, S.cin_pad);
            else ws_store(s, S.leaf_state + slot * geo.bytes, geo);  // game_repr happens inside the fused tower
        }
        else if (S.evaluator == TG_EVAL_HASH) {
            uint64_t h = ws_hash(s, geo);
            if (lane == 0) S.leaf_hash[slot] = h;
        }
    }
    TG_TSTAMP(g, 29);  // leaf state stored
    if (lane == 0) {
        S.path_len[slot] = depth;
        S.leaf_kind[slot] = terminal ? 2 : 1;
        __hip_atomic_fetch_add(&S.counters[2 * (size_t)g], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (returnless: no round trip)
        if (!terminal) __hip_atomic_fetch_add(&S.counters[2 * (size_t)g + 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One wave per game.  S.pass ≥ 0: that one virtual rollout; S.pass < 0: all `batch` virtual rollouts of the iteration one after
// the other — a game's tree is only ever touched by its own wave, so the passes need no kernel boundary between them, only
// the wave's own stores ordered before its later loads (s_waitcnt: a wave's accesses go through one in-order, write-through
// L1 path; an agent-scope fence would write back the whole L2 per wave and cost more than the launches it replaces).
template <int NB>
__global__ __launch_bounds__(WPB * 64) void k_select(SearchDev S, const uint8_t* __restrict__ active) {
    __shared__ uint32_t path_lds[WPB][MAX_DEPTH];
    __shared__ uint16_t mv_lds[WPB][EX_MOVES];
    const int g = game_of_wave();
    if (g >= S.G) return;
    uint32_t* path = path_lds[threadIdx.x >> 6];
    const int p0 = S.pass < 0 ? 0 : S.pass, p1 = S.pass < 0 ? S.batch : S.pass + 1;
    for (int p = p0; p < p1; p++) {
        select_pass<NB>(S, active, g, p, path, mv_lds[threadIdx.x >> 6]);
        if (p + 1 < p1) wave_sync_mem();
    }
}

// ------------------------------------------------------------------------------------------------
// devirtualize_path, mcts.rs:67-91: real priors for the leaf's children, value backed up with
// alternating sign, virtual visits removed.
// ------------------------------------------------------------------------------------------------
// Every load that does not depend on another load is issued first (leaf kind, path length, the leaf's children block and their
// policy indices — both recorded by the select that expanded it —, the FC's softmax statistics, the value logit, this lane's
// path entry); the second round is the gather of the children's logits and the path nodes' records; then the stores.  Round 2
// walked path → leaf → children's moves → index table → logits, five dependent round trips (19 k of the wave's 56 k cycles).
// Returns through `pre` the root's (visits, virtual) as this backup leaves them, for the select that follows in the same launch.
template <int NB>
__device__ __forceinline__ void backup_pass(const SearchDev& S, const int g, const int pass, const uint32_t root_v, RootPre* pre = nullptr) {
    const size_t slot = (size_t)g * (size_t)S.batch + (size_t)pass;
    const int lane = lane_id();
    NodeHot* hot = S.hot;
    const uint32_t* path = S.path + slot * MAX_DEPTH;
    const float* lrow = S.logits ? S.logits + slot * (size_t)S.logit_ld : nullptr;
    const float* pol = S.policy + slot * S.P;
    const uint16_t* pidx = S.child_pidx + slot * EX_MOVES;
    // round 1 — every load whose address does not come out of another load, requested back to back and unconditionally (all
    // addresses are valid whatever the slot holds; the wave-uniform values are made scalar only after the last request): the
    // leaf's kind, path length and children block, this lane's child index and path entry, the statistics, the value logit
    const uint32_t kind_v = (uint32_t)S.leaf_kind[slot];
    const uint32_t len_v = (uint32_t)S.path_len[slot];
    const uint32_t cb_v = S.leaf_rec[2 * slot], n_v = S.leaf_rec[2 * slot + 1];
    const uint32_t raw_idx = (uint32_t)pidx[lane], raw_idx2 = (uint32_t)pidx[lane + 64];  // (EX_MOVES ≥ 128 entries per slot)
    const uint32_t raw_nd = path[lane ? lane - 1 : 0];
    // round 4: the policy FC's epilogue has already picked the children's logits out of its accumulators (child_logit, in child
    // order): two coalesced loads in THIS round replace the dependent gather logits[pidx[child]] of round 2, and the value
    // pre-activation comes with the statistics record
    const float* clog = S.child_logit ? S.child_logit + slot * EX_MOVES : nullptr;
    const float clog_v = clog ? clog[lane] : 0.0f, clog_v2 = clog ? clog[lane + 64] : 0.0f;
    const float vlogit_v = (S.evaluator == TG_EVAL_RESNET && lrow && !clog) ? lrow[S.P] : 0.0f;
    // root_v = S.root[g] as the caller requested it, not yet waited for: it was the first request, so making it scalar here
    // waits for that one load alone, and the select's cold record of the root joins the requests above
    const uint32_t root = uni(root_v);
    if (pre) { pre->root = root; pre->cold = S.cold[root]; }
    const bool live = uni(kind_v) == 1u;
    const int L = (int)uni(len_v);
    const uint32_t cb = uni(cb_v), nchild = uni(n_v);
    float e;
    uint64_t hsh = 0;
    float lmx = 0.0f, linv = 0.0f;
    if (S.evaluator == TG_EVAL_RESNET && lrow) {
        float vlogit = vlogit_v;
        if (S.fc_stats)  // one 8-byte load per lane instead of the whole row
            fc_combine_stats(S.fc_stats + (size_t)uni((uint32_t)slot) * (size_t)S.fc_stride * 2, S.fc_blocks, lmx, linv, clog ? &vlogit : nullptr);
        else if (live) softmax_stats_wave(lrow, S.P, lmx, linv);
        e = tanhf(vlogit);
    } else if (S.evaluator == TG_EVAL_RESNET) e = S.eval[slot];
    else if (S.evaluator == TG_EVAL_HASH) { hsh = S.leaf_hash[slot]; e = hash_eval(hsh); }
    else e = 0.0f;
    if (!live) return;  // (terminal leaf: backed up concretely by the select; skipped game)
    const uint32_t my_idx = (uint32_t)lane < nchild ? raw_idx : 0xFFFFu;
    const uint32_t my_nd = lane == 0 ? root : raw_nd;
    TG_TSTAMP(g, 1);  // independent loads done
    // round 2: priors of the leaf's children (devirtualize_path, mcts.rs:80-84)
    bool bad = false;
    for (uint32_t i = lane; i < nchild; i += 64) {
        const uint32_t idx = i == (uint32_t)lane ? my_idx : i == (uint32_t)lane + 64u ? raw_idx2 : (uint32_t)pidx[i];
        float p;
        if (idx == 0xFFFFu) { bad = true; p = 0.0f; }
        else if (S.evaluator == TG_EVAL_RESNET) {
            if (clog) p = stat_exp((i == (uint32_t)lane ? clog_v : i == (uint32_t)lane + 64u ? clog_v2 : clog[i]) - lmx) * linv;
            else p = !lrow ? pol[idx] : S.fc_stats ? stat_exp(lrow[idx] - lmx) * linv : expf(lrow[idx] - lmx) * linv;
        }
        else if (S.evaluator == TG_EVAL_HASH) p = hash_policy(hsh, idx);
        else p = 1.0f;
        hot[cb + i].prior = p;
    }
    if (__ballot(bad)) flag(S, ERRF_MOVE);
    TG_TSTAMP(g, 2);  // priors written
    uint32_t rv = 0, rvv = 0;
    for (int d = lane; d <= L; d += 64) {
        uint32_t nd = d == lane ? my_nd : path[d - 1];
        NodeHot h = hot[nd];
        h.virt -= 1;
        float ev = ((L - d) & 1) ? e : -e;  // the leaf sees -eval, its parent +eval, …
        update_concrete(h, ev);
        // the prior of this node may just have been rewritten above only if it is a child of the leaf,
        // which it is not (it lies on the path), so writing the whole record back is safe
        hot[nd].q = h.q;
        hot[nd].visits = h.visits;
        hot[nd].virt = h.virt;
        if (d == 0) { rv = h.visits; rvv = h.virt; }
    }
    if (pre) {  // lane 0 handled the root (d = 0)
        pre->vis = uni(rv);
        pre->vv = uni(rvv);
        pre->hot_known = true;
    }
}

// S.pass as in k_select: one de-virtualisation, or all of the iteration's in rollout order
template <int NB>
__global__ __launch_bounds__(WPB * 64) void k_backup(SearchDev S) {
    const int g = game_of_wave();
    if (g >= S.G) return;
    const int p0 = S.pass < 0 ? 0 : S.pass, p1 = S.pass < 0 ? S.batch : S.pass + 1;
    for (int p = p0; p < p1; p++) {
        backup_pass<NB>(S, g, p, S.root[g]);
        if (p + 1 < p1) wave_sync_mem();
    }
}

// De-virtualise iteration i and select the leaf of iteration i + 1 in one launch (one leaf per game): the two touch the same
// few nodes of the same tree from the same wave, so the second finds them in cache and one kernel boundary per iteration
// disappears.  Same device functions as the separate kernels → same trees.
template <int NB>
__global__ __launch_bounds__(WPB * 64) void k_backup_select(SearchDev S) {
    __shared__ uint32_t path_lds[WPB][MAX_DEPTH];
    __shared__ uint16_t mv_lds[WPB][EX_MOVES];
    const int g = game_of_wave();
    if (g >= S.G) return;
    TG_TSTAMP(g, 0);
    // what the select needs of the root and the backup does not write is requested before the backup: the root's index, its
    // cold record and the packed root position arrive under the backup's own round trips
    // (requested, none of them waited for: the first wait is inside the backup, behind its own independent requests)
    RootPre pre;
    const uint32_t root_v = S.root[g];
    const Geom geo = make_geom(NB ? NB : S.n);
    pre.raw = ws_load_raw(S.root_state + (size_t)g * geo.bytes, geo);
    pre.alive_v = (uint32_t)S.alive[g];
    pre.abort_v = (uint32_t)S.abort[g];
    pre.on = true;
    backup_pass<NB>(S, g, 0, root_v, &pre);
    wave_sync_mem();
    TG_TSTAMP(g, 3);  // path updated
    select_pass<NB>(S, nullptr, g, 0, path_lds[threadIdx.x >> 6], mv_lds[threadIdx.x >> 6], pre);
    TG_TSTAMP(g, 31);
}

// ------------------------------------------------------------------------------------------------
// apply_dirichlet, noise.rs:6-16
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dirichlet(SearchDev S, const uint8_t* __restrict__ active, float alpha, float ratio) {
    __shared__ double gam[EX_MOVES];
    __shared__ double sum_s;
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    if (!S.alive[g] || (active && !active[g])) return;
    const Geom geo = make_geom(S.n);
    const uint32_t root = S.root[g];
    NodeHot* hot = S.hot;
    NodeCold rc = S.cold[root];
    const uint32_t nchild = (uint32_t)rc.nres & 0xfffu, cb = rc.child;
    if (nchild == 0 || hot[root].visits == 0) return;  // reference asserts visits > 0
    const uint32_t* hdr = (const uint32_t*)(S.root_state + (size_t)g * geo.bytes + geo.bytes - 16);
    const uint32_t ply = hdr[0] >> 16;
    const uint32_t gen = S.generation[g];
    for (uint32_t i = lane; i < nchild && i < EX_MOVES; i += 64)
        gam[i] = gamma_sample((double)alpha, S.seed, S.slot_base + (uint32_t)g, gen, ply, i);
    __syncthreads();
    if (lane == 0) {
        double sum = 0.0;
        for (uint32_t i = 0; i < nchild; i++) sum += gam[i];  // index order: matches the sequential statement
        sum_s = sum;
    }
    __syncthreads();
    const double sum = sum_s;
    for (uint32_t i = lane; i < nchild; i += 64) {
        float noise = sum > 0.0 ? (float)(gam[i] / sum) : (float)(1.0 / (double)nchild);
        float p = hot[cb + i].prior;
        hot[cb + i].prior = noise * ratio + p * (1.0f - ratio);
    }
}

__global__ __launch_bounds__(64) void k_apply_noise(SearchDev S, const uint8_t* __restrict__ active, const float* __restrict__ noise, float ratio) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    if (!S.alive[g] || (active && !active[g])) return;
    NodeHot* hot = S.hot;
    NodeCold rc = S.cold[S.root[g]];
    const uint32_t nchild = (uint32_t)rc.nres & 0xfffu, cb = rc.child;
    for (uint32_t i = lane; i < nchild && i < EX_MOVES; i += 64) {
        float p = hot[cb + i].prior;
        hot[cb + i].prior = noise[(size_t)g * EX_MOVES + i] * ratio + p * (1.0f - ratio);
    }
}

// ------------------------------------------------------------------------------------------------
// tree reuse: Node::play, play.rs:26-43.  op[g]: -2 = reset the tree (Node::default()), -1 = nothing,
// ≥ 0 = make that child the root.  The kept subtree is copied breadth-first into fresh chunks and the game's
// old chunks go back to the pool.  The copy is its own work queue (Cheney): a node is first copied with the OLD
// index of its children block; a scan pointer follows the allocation pointer through the new chunks and, for
// every copied node that has children, copies that block behind the allocation pointer and patches the index.
// No queue in LDS, no limit on the width or depth of the tree.  One wave per game.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_reroot(SearchDev S, const int32_t* __restrict__ op) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int o = op[g];
    if (o == -1) return;
    const uint32_t CH = 1u << S.chunk_shift;
    const uint32_t old_root = S.root[g];
    const uint32_t old_head = S.chunk_head[g];
    NodeHot* hot = S.hot;
    NodeCold* cold = S.cold;
    if (o == -2) {
        pool_give_chain(S, old_head);
        const uint32_t c = pool_take(S);
        if (c == 0) { flag(S, ERRF_ARENA); return; }
        const uint32_t a = c << S.chunk_shift;
        if (lane == 0) {
            NodeHot h; h.prior = 0.0f; h.q = 0.0f; h.visits = 0; h.virt = 0;
            NodeCold cc; cc.child = 0; cc.mv = 0; cc.nres = 0;
            hot[a] = h;
            cold[a] = cc;
            S.root[g] = a;
            S.alloc[2 * g] = a + 1;
            S.alloc[2 * g + 1] = a + CH;
            S.chunk_link[c] = 0;
            S.chunk_head[g] = c;
        }
        return;
    }
    const NodeCold rc = cold[old_root];
    const uint32_t rn = (uint32_t)rc.nres & 0xfffu;
    if ((uint32_t)o >= rn) { flag(S, ERRF_MOVE); return; }
    const uint32_t oc = rc.child + (uint32_t)o;
    uint32_t cur = pool_take(S);  // the chunk being filled
    if (cur == 0) { flag(S, ERRF_ARENA); return; }
    const uint32_t first = cur;
    uint32_t aoff = 1;            // nodes used in `cur`
    if (lane == 0) {
        hot[first << S.chunk_shift] = hot[oc];
        cold[first << S.chunk_shift] = cold[oc];  // .child still names the old block: patched when the scan reaches it
        S.chunk_link[first] = 0;
    }
    wave_sync_mem();
    uint32_t scan_c = first, scan_off = 0;
    for (;;) {
        const uint32_t limit = scan_c == cur ? aoff : uni(S.chunk_used[scan_c]);
        if (scan_off >= limit) {
            if (scan_c == cur) break;
            scan_c = uni(S.chunk_fwd[scan_c]);
            scan_off = 0;
            continue;
        }
        const uint32_t m = min(64u, limit - scan_off);
        const uint32_t idx = (scan_c << S.chunk_shift) + scan_off + (uint32_t)lane;
        const bool on = (uint32_t)lane < m;
        NodeCold c;
        c.child = 0; c.mv = 0; c.nres = 0;
        if (on) c = cold[idx];
        const uint32_t nch = on ? ((uint32_t)c.nres & 0xfffu) : 0u;
        uint32_t my_new = 0;
        bool failed = false;
        for (uint64_t bb = __ballot(nch > 0); bb; bb &= bb - 1) {
            const int l = __builtin_ctzll(bb);
            const uint32_t cnt = uni((uint32_t)__shfl((int)nch, l));
            const uint32_t src = uni((uint32_t)__shfl((int)c.child, l));
            if (aoff + cnt > CH) {  // close the chunk, open the next
                const uint32_t c2 = pool_take(S);
                if (c2 == 0) { failed = true; break; }
                if (lane == 0) { S.chunk_used[cur] = aoff; S.chunk_fwd[cur] = c2; S.chunk_link[c2] = cur; }
                cur = c2;
                aoff = 0;
            }
            const uint32_t dst = (cur << S.chunk_shift) + aoff;
            for (uint32_t k = lane; k < cnt; k += 64) {
                hot[dst + k] = hot[src + k];
                cold[dst + k] = cold[src + k];
            }
            if (lane == l) my_new = dst;
            aoff += cnt;
        }
        if (failed) { flag(S, ERRF_ARENA); return; }  // the error is sticky: the engine refuses further work until reset
        if (nch > 0) cold[idx].child = my_new;
        scan_off += m;
        wave_sync_mem();  // the scan reads back what this wave has just written
    }
    if (lane == 0) {
        S.root[g] = first << S.chunk_shift;
        S.alloc[2 * g] = (cur << S.chunk_shift) + aoff;
        S.alloc[2 * g + 1] = (cur + 1) << S.chunk_shift;
        S.chunk_head[g] = cur;
    }
    pool_give_chain(S, old_head);
}

// root children → host-visible arrays (Node::improved_policy, play.rs:13-21, plus priors / q)
__global__ __launch_bounds__(64) void k_root_stats(SearchDev S, uint16_t* moves, uint32_t* visits, float* prior, float* q,
                                                   int32_t* counts, uint32_t* root_visits, float* root_q) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t root = S.root[g];
    NodeCold rc = S.cold[root];
    NodeHot rh = S.hot[root];
    const uint32_t nchild = (uint32_t)rc.nres & 0xfffu, cb = rc.child;
    if (lane == 0) { counts[g] = (int32_t)nchild; root_visits[g] = rh.visits; root_q[g] = rh.q; }
    for (uint32_t i = lane; i < nchild && i < EX_MOVES; i += 64) {
        NodeHot h = S.hot[cb + i];
        size_t o = (size_t)g * EX_MOVES + i;
        moves[o] = S.cold[cb + i].mv;
        visits[o] = h.visits;
        prior[o] = h.prior;
        q[o] = h.q;
    }
}

// tg_search_play: find the child for a caller-chosen move, play it on the root state
__global__ __launch_bounds__(WPB * 64) void k_play_move(SearchDev S, const uint16_t* __restrict__ moves, const uint8_t* __restrict__ active,
                                                   int32_t* __restrict__ op) {
    const int g = game_of_wave();
    if (g >= S.G) return;
    const int lane = lane_id();
    if (!S.alive[g] || (active && !active[g])) { if (lane == 0) op[g] = -1; return; }
    const Geom geo = make_geom(S.n);
    NodeCold rc = S.cold[S.root[g]];
    const uint32_t nchild = uni((uint32_t)rc.nres) & 0xfffu, cb = uni(rc.child);
    const uint32_t mv = uni((uint32_t)moves[g]);
    int found = -1;
    for (uint32_t i0 = 0; i0 < nchild && found < 0; i0 += 64) {
        uint32_t i = i0 + lane;
        bool hit = i < nchild && S.cold[cb + i].mv == mv;
        uint64_t bb = __ballot(hit);
        if (bb) found = (int)i0 + __builtin_ctzll(bb);
    }
    if (found < 0) {  // "tried to play an invalid move" / "node must be initialized" (play.rs:10,35)
        flag(S, ERRF_MOVE);
        if (lane == 0) op[g] = -1;
        return;
    }
    WState s;
    uint8_t* st = S.root_state + (size_t)g * geo.bytes;
    ws_load(s, st, geo);
    ws_play(s, mv, geo);
    ws_store(s, st, geo);
    if (lane == 0) op[g] = found;
}

// ------------------------------------------------------------------------------------------------
// self_play_parallel phases (train/src/self_play.rs:108-259)
// ------------------------------------------------------------------------------------------------

// (a) opening, :110-116.  "a1", then one of the two far corners (reference hard-codes the 6×6 names
// a6 / f6; generalised to (0, N-1) / (N-1, N-1)).
__global__ __launch_bounds__(WPB * 64) void k_sp_opening(SearchDev S) {
    const int g = game_of_wave();
    if (g >= S.G) return;
    if (!S.alive[g]) return;
    const Geom geo = make_geom(S.n);
    WState s;
    uint8_t* st = S.root_state + (size_t)g * geo.bytes;
    ws_load(s, st, geo);
    if (s.ply != 0) return;
    ws_play(s, 0u /* a1 flat */, geo);
    U4 r = rng_draw(S.seed, S.slot_base + (uint32_t)g, S.generation[g], 0, RNG_OPENING, 0, 0);
    uint32_t col = (r.v[0] & 1u) ? 0u : (uint32_t)(geo.n - 1);
    ws_play(s, (uint32_t)((geo.n - 1) * geo.n) + col, geo);
    ws_store(s, st, geo);
}

__device__ inline void stage_example(const SearchDev& S, const SelfPlayDev& P, int g, const WState& s, const Geom& geo,
                                     uint32_t nmoves, int& slot_out) {
    // reserves the next staging slot of game g and writes header + state; returns the slot (or -1)
    int k = P.st_count[g];
    if (k >= P.max_game_plies) { limit_hit(S, g, ERRF_EXAMPLES); slot_out = -1; return; }
    size_t e = (size_t)g * P.ex_per_game + k;
    ws_store(s, P.st_state + e * geo.bytes, geo);
    if (lane_id() == 0) {
        ExampleRec h;
        h.slot = (int32_t)(S.slot_base + (uint32_t)g);
        h.generation = (int32_t)S.generation[g];
        h.n_moves = (int32_t)nmoves;
        h.result = 0.0f;
        P.st_hdr[e] = h;
        P.st_count[g] = k + 1;
    }
    slot_out = k;
}

// (b) instant-win scan, :119-171
__global__ __launch_bounds__(WPB * 64) void k_sp_instant_win(SearchDev S, SelfPlayDev P) {
    __shared__ uint16_t mv_lds[WPB][EX_MOVES];
    __shared__ uint8_t win_lds[WPB][EX_MOVES];
    const int g = game_of_wave();
    if (g >= S.G) return;
    const int lane = lane_id();
    if (lane == 0) P.fin[g] = 0;
    if (!S.alive[g]) return;
    const Geom geo = make_geom(S.n);
    WState s;
    ws_load(s, S.root_state + (size_t)g * geo.bytes, geo);
    uint16_t* mv = mv_lds[threadIdx.x >> 6];
    uint8_t* wl = win_lds[threadIdx.x >> 6];
    int count = ws_movegen(s, geo, EX_MOVES, [&](int idx, uint32_t code) { mv[idx] = (uint16_t)code; });
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (count > EX_MOVES) {  // (no such position exists on boards up to 6×6; kept as a bound on the LDS staging)
        limit_hit(S, g, ERRF_MOVES);
        if (lane == 0) P.fin[g] = FIN_ABORTED;
        return;
    }
    bool win = false;
    for (int k = 0; k < count; k++) {
        WState t = s;
        ws_play(t, (uint32_t)mv[k], geo);
        uint32_t r = ws_result(t, geo);
        bool w = (r >= TG_WHITE_ROAD && r <= TG_BLACK_FLAT) && (((r == TG_WHITE_ROAD || r == TG_WHITE_FLAT) ? 0u : 1u) == s.to_move);
        if (lane == 0) wl[k] = w ? 1 : 0;
        win |= w;
    }
    if (!win) return;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int slot;
    stage_example(S, P, g, s, geo, (uint32_t)count, slot);
    if (slot < 0) {  // the game is past max_game_plies: retired without examples
        if (lane == 0) P.fin[g] = FIN_ABORTED;
        return;
    }
    {
        size_t e = (size_t)g * P.ex_per_game + slot;
        for (int k = lane; k < count; k += 64) {
            P.st_moves[e * EX_MOVES + k] = mv[k];
            P.st_visits[e * EX_MOVES + k] = wl[k] ? 1000u : 1u;  // fake visits, :129-134
        }
    }
    if (lane == 0) {
        P.fin[g] = (uint8_t)(s.to_move == 0 ? TG_WHITE_FLAT : TG_BLACK_FLAT);  // Winner { color: to_move, road: false }, :147
        atomicAdd(&P.stats[ST_INSTANT], 1ull);
    }
}

// ordered bookkeeping of the games that ended in this phase (slot order = the reference's loop order):
// completed counter, recycle-or-retire decision (:151,237), contiguous output ranges for their examples.
__global__ __launch_bounds__(1024) void k_sp_finish_scan(SearchDev S, SelfPlayDev P) {
    __shared__ uint32_t s_fin[1024], s_ex[1024];
    const int tid = threadIdx.x;
    const int per = (S.G + 1023) / 1024;
    const int g0 = tid * per, g1 = min(g0 + per, S.G);
    uint32_t nf = 0, ne = 0;
    // (a retired game — FIN_ABORTED — is not a completed game and emits nothing; its slot always restarts)
    for (int g = g0; g < g1; g++)
        if (P.fin[g] && P.fin[g] != FIN_ABORTED) { nf++; ne += (uint32_t)P.st_count[g]; }
    s_fin[tid] = nf;
    s_ex[tid] = ne;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis–Steele inclusive scan
        uint32_t a = tid >= d ? s_fin[tid - d] : 0u, b = tid >= d ? s_ex[tid - d] : 0u;
        __syncthreads();
        s_fin[tid] += a;
        s_ex[tid] += b;
        __syncthreads();
    }
    const unsigned long long done0 = P.stats[ST_FINISHED], ex0 = P.stats[ST_EXAMPLES];
    uint32_t rf = s_fin[tid] - nf, re = s_ex[tid] - ne;
    for (int g = g0; g < g1; g++) {
        if (!P.fin[g]) continue;
        if (P.fin[g] == FIN_ABORTED) { P.recycle[g] = 1; continue; }
        unsigned long long completed = done0 + rf + 1;
        P.recycle[g] = (P.total_games == 0 || completed + (unsigned long long)S.G < (unsigned long long)P.total_games) ? 1 : 0;
        P.out_off[g] = (uint32_t)((ex0 + re) % (unsigned long long)P.max_examples);
        rf++;
        re += (uint32_t)P.st_count[g];
    }
    __syncthreads();
    if (tid == 1023) {
        P.stats[ST_FINISHED] = done0 + s_fin[1023];
        P.stats[ST_EXAMPLES] = ex0 + s_ex[1023];
    }
}

// complete the finished games' examples (:158-169, :245-256), reset tree and game (or retire the slot)
__global__ __launch_bounds__(WPB * 64) void k_sp_finish_apply(SearchDev S, SelfPlayDev P, int32_t* __restrict__ op) {
    const int g = game_of_wave();
    if (g >= S.G) return;
    const int lane = lane_id();
    const uint32_t r = P.fin[g];
    if (!r) return;
    const Geom geo = make_geom(S.n);
    const bool aborted = r == FIN_ABORTED;
    const float white_result = (r == TG_WHITE_ROAD || r == TG_WHITE_FLAT) ? 1.0f : (r == TG_BLACK_ROAD || r == TG_BLACK_FLAT) ? -1.0f : 0.0f;
    const int cnt = aborted ? 0 : P.st_count[g];
    const uint32_t off = P.out_off[g];
    for (int k = 0; k < cnt; k++) {
        size_t e = (size_t)g * P.ex_per_game + k;
        size_t o = (size_t)((off + (uint32_t)k) % (uint32_t)P.max_examples);
        ExampleRec h = P.st_hdr[e];
        const uint8_t* st = P.st_state + e * geo.bytes;
        uint32_t to_move = st[geo.bytes - 16 + 1];
        h.result = to_move == 0 ? white_result : -white_result;
        if (lane == 0) P.out_hdr[o] = h;
        for (int b = lane; b < geo.bytes / 4; b += 64) ((uint32_t*)(P.out_state + o * geo.bytes))[b] = ((const uint32_t*)st)[b];
        for (int m = lane; m < h.n_moves; m += 64) {
            P.out_moves[o * EX_MOVES + m] = P.st_moves[e * EX_MOVES + m];
            P.out_visits[o * EX_MOVES + m] = P.st_visits[e * EX_MOVES + m];
        }
    }
    WState s;
    ws_start(s, geo, P.komi * 2);
    ws_store(s, S.root_state + (size_t)g * geo.bytes, geo);
    if (lane == 0) {
        P.st_count[g] = 0;
        S.generation[g] += 1;
        if (!P.recycle[g]) S.alive[g] = 0;
        op[g] = -2;  // *node = Node::default()
        S.abort[g] = 0;
        atomicAdd(&P.stats[aborted ? ST_ABORTED : white_result > 0 ? ST_WHITE : white_result < 0 ? ST_BLACK : ST_DRAWS], 1ull);
    }
}

__global__ void k_sp_noise_mask(SearchDev S, SelfPlayDev P) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.G) return;
    const Geom geo = make_geom(S.n);
    const uint32_t* hdr = (const uint32_t*)(S.root_state + (size_t)g * geo.bytes + geo.bytes - 16);
    P.mask[g] = (S.alive[g] && (int)(hdr[0] >> 16) < P.noise_plies) ? 1 : 0;
}

// (e) pick_move (play.rs:49-67), example (:222-225), play (:227-228), result (:230)
__global__ __launch_bounds__(WPB * 64) void k_sp_pick(SearchDev S, SelfPlayDev P, int32_t* __restrict__ op) {
    const int g = game_of_wave();
    if (g >= S.G) return;
    const int lane = lane_id();
    if (lane == 0) { P.fin[g] = 0; op[g] = -1; }
    if (!S.alive[g]) return;
    if (S.abort[g]) {  // the search of this ply ran into a capacity (depth, visits): retire the game
        if (lane == 0) P.fin[g] = FIN_ABORTED;
        return;
    }
    const Geom geo = make_geom(S.n);
    const NodeHot* hot = S.hot;
    const NodeCold* cold = S.cold;
    NodeCold rc = cold[S.root[g]];
    const uint32_t nchild = uni((uint32_t)rc.nres) & 0xfffu, cb = uni(rc.child);
    WState s;
    uint8_t* st = S.root_state + (size_t)g * geo.bytes;
    ws_load(s, st, geo);
    if (nchild == 0) { flag(S, ERRF_PICK); return; }
    int pick = -1;
    if ((int)s.ply >= P.exploit_plies) {
        // max_by_key: most visits, last on ties
        uint32_t bv = 0;
        int bi = -1;
        for (uint32_t i = lane; i < nchild; i += 64) {
            uint32_t v = hot[cb + i].visits;
            if (bi < 0 || v >= bv) { bv = v; bi = (int)i; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            uint32_t ov = (uint32_t)__shfl_xor((int)bv, d);
            int oi = __shfl_xor(bi, d);
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi))) { bv = ov; bi = oi; }
        }
        pick = bi;
    } else {
        // WeightedIndex over visits with one RNG_PICK draw
        unsigned long long total = 0;
        for (uint32_t i = lane; i < nchild; i += 64) total += hot[cb + i].visits;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(total >> 32), d) << 32) |
                                   (uint32_t)__shfl_xor((int)(uint32_t)total, d);
            total += o;
        }
        if (total == 0) { flag(S, ERRF_PICK); return; }
        U4 r = rng_draw(S.seed, S.slot_base + (uint32_t)g, S.generation[g], s.ply, RNG_PICK, 0, 0);
        unsigned long long x = ((unsigned long long)r.v[0] << 32) | r.v[1];
        unsigned long long target = __umul64hi(x, total);
        unsigned long long carry = 0;
        for (uint32_t i0 = 0; i0 < nchild && pick < 0; i0 += 64) {
            uint32_t i = i0 + lane;
            int v = i < nchild ? (int)hot[cb + i].visits : 0;
            int incl = wave_inclusive_scan(v);
            uint64_t bb = __ballot(i < nchild && carry + (unsigned long long)incl > target);
            if (bb) pick = (int)i0 + __builtin_ctzll(bb);
            carry += (unsigned long long)__shfl(incl, 63);
        }
        if (pick < 0) pick = (int)nchild - 1;
    }
    // example = (game before the move, visit counts of every child)
    int slot;
    stage_example(S, P, g, s, geo, nchild, slot);
    if (slot < 0) {  // the game is past max_game_plies
        if (lane == 0) P.fin[g] = FIN_ABORTED;
        return;
    }
    {
        size_t e = (size_t)g * P.ex_per_game + slot;
        for (uint32_t i = lane; i < nchild && i < EX_MOVES; i += 64) {
            P.st_moves[e * EX_MOVES + i] = cold[cb + i].mv;
            P.st_visits[e * EX_MOVES + i] = hot[cb + i].visits;
        }
    }
    const uint32_t mv = uni((uint32_t)cold[cb + (uint32_t)pick].mv);
    ws_play(s, mv, geo);
    ws_store(s, st, geo);
    const uint32_t res = ws_result(s, geo);
    if (lane == 0) {
        P.fin[g] = (uint8_t)res;
        op[g] = pick;
        P.chosen[g] = pick;
    }
}

__global__ void k_sp_count_ply(SelfPlayDev P) { P.stats[ST_PLIES] += 1; }

// ---- launchers --------------------------------------------------------------------------------
static inline dim3 wgrid(int G) { return dim3((G + WPB - 1) / WPB); }

// the tree kernels are compiled for the board sizes of the BASELINE configs (n as a constant) and once for any size
static const bool g_runtime_n = env_on("TG_RUNTIME_N");  // A/B: the generic instantiation for every size (same results)
#define TG_BY_BOARD(KERNEL, ...)                                                                          \
    do {                                                                                                  \
        if (S.n == 5 && !g_runtime_n) hipLaunchKernelGGL(KERNEL<5>, wgrid(S.G), dim3(WPB * 64), 0, st, __VA_ARGS__);      \
        else if (S.n == 6 && !g_runtime_n) hipLaunchKernelGGL(KERNEL<6>, wgrid(S.G), dim3(WPB * 64), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<0>, wgrid(S.G), dim3(WPB * 64), 0, st, __VA_ARGS__);                    \
    } while (0)
void launch_select(hipStream_t st, const SearchDev& S, const uint8_t* active) { TG_BY_BOARD(k_select, S, active); }
void launch_backup(hipStream_t st, const SearchDev& S) { TG_BY_BOARD(k_backup, S); }
void launch_backup_select(hipStream_t st, const SearchDev& S) { TG_BY_BOARD(k_backup_select, S); }
void launch_dirichlet(hipStream_t st, const SearchDev& S, const uint8_t* active, float alpha, float ratio) {
    hipLaunchKernelGGL(k_dirichlet, dim3(S.G), dim3(64), 0, st, S, active, alpha, ratio);
}
void launch_apply_noise(hipStream_t st, const SearchDev& S, const uint8_t* active, const float* noise, float ratio) {
    hipLaunchKernelGGL(k_apply_noise, dim3(S.G), dim3(64), 0, st, S, active, noise, ratio);
}
void launch_reroot(hipStream_t st, const SearchDev& S, const int32_t* op) {
    hipLaunchKernelGGL(k_reroot, dim3(S.G), dim3(64), 0, st, S, op);
    hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(1), 0, st, S);
}
void launch_root_stats(hipStream_t st, const SearchDev& S, uint16_t* moves, uint32_t* visits, float* prior, float* q, int32_t* counts,
                       uint32_t* root_visits, float* root_q) {
    hipLaunchKernelGGL(k_root_stats, dim3(S.G), dim3(64), 0, st, S, moves, visits, prior, q, counts, root_visits, root_q);
}
void launch_play_move(hipStream_t st, const SearchDev& S, const uint16_t* moves, const uint8_t* active, int32_t* op) {
    hipLaunchKernelGGL(k_play_move, wgrid(S.G), dim3(WPB * 64), 0, st, S, moves, active, op);
}
void launch_sp_opening(hipStream_t st, const SearchDev& S) { hipLaunchKernelGGL(k_sp_opening, wgrid(S.G), dim3(WPB * 64), 0, st, S); }
void launch_sp_instant_win(hipStream_t st, const SearchDev& S, const SelfPlayDev& P) {
    hipLaunchKernelGGL(k_sp_instant_win, wgrid(S.G), dim3(WPB * 64), 0, st, S, P);
}
void launch_sp_finish(hipStream_t st, const SearchDev& S, const SelfPlayDev& P, int32_t* op) {
    hipLaunchKernelGGL(k_sp_finish_scan, dim3(1), dim3(1024), 0, st, S, P);
    hipLaunchKernelGGL(k_sp_finish_apply, wgrid(S.G), dim3(WPB * 64), 0, st, S, P, op);
}
void launch_sp_noise_mask(hipStream_t st, const SearchDev& S, const SelfPlayDev& P) {
    hipLaunchKernelGGL(k_sp_noise_mask, dim3((S.G + 255) / 256), dim3(256), 0, st, S, P);
}
void launch_sp_pick(hipStream_t st, const SearchDev& S, const SelfPlayDev& P, int32_t* op) {
    hipLaunchKernelGGL(k_sp_pick, wgrid(S.G), dim3(WPB * 64), 0, st, S, P, op);
}
void launch_sp_count_ply(hipStream_t st, const SelfPlayDev& P) { hipLaunchKernelGGL(k_sp_count_ply, dim3(1), dim3(1), 0, st, P); }

#ifdef TG_TREE_STAMPS
extern "C" int tg_debug_tree_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_stamps), sizeof(g_tree_stamps));
}
#endif

}  // namespace tg
