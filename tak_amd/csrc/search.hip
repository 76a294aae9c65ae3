// search.hip — host side of the search / self-play entry points of include/takgpu.h: device
// allocation, the per-iteration schedule (select → network → backup, no host synchronisation inside a
// call), the per-ply schedule of self_play_parallel, result read-back and the tree dump.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "engine.h"
#include "kernels.h"
#include "search.cuh"

namespace tg {

struct Search {
    TgSearchConfig cfg;
    SearchDev d;
    DevBuf hot, cold, root, alloc, chunk_head, chunk_link, chunk_fwd, chunk_used, free_ring, pool_ctl, root_state, alive, generation, path_len, path, leaf_kind, leaf_rec, child_pidx, child_logit, leaf_hash, planes, leaf_state, policy, eval,
        ctab, err, counters, op, active, noise, abort;
    DevBuf r_moves, r_visits, r_prior, r_q, r_counts, r_rv, r_rq, s_moves;
    // self-play
    bool selfplay = false;
    TgSelfPlayConfig spcfg;
    SelfPlayDev p;
    DevBuf st_hdr, st_state, st_moves, st_visits, st_count, out_hdr, out_state, out_moves, out_visits, fin, recycle, out_off, chosen,
        mask, stats;
    unsigned long long drained = 0, dropped = 0;
};

void search_destroy(Search* s) { delete s; }

static int describe_errors(uint32_t bits) {
    std::string msg;
    int code = TG_ERR_STATE;
    auto add = [&](uint32_t b, const char* m, int c) {
        if (bits & b) { if (!msg.empty()) msg += "; "; msg += m; code = c; }
    };
    add(ERRF_DEPTH, "selection path deeper than 256 plies (TG_LIMIT_DEPTH)", TG_ERR_LIMIT);
    add(ERRF_CTAB, "a node was visited more often than the exploration-rate table is long (TG_LIMIT_VISITS / TgSearchConfig.visit_limit)", TG_ERR_LIMIT);
    add(ERRF_EXAMPLES, "a game lasted more than 512 plies (TG_LIMIT_GAME_PLIES)", TG_ERR_LIMIT);
    add(ERRF_MOVES, "a position has more than TG_MAX_MOVES legal moves", TG_ERR_LIMIT);
    add(ERRF_PICK, "pick_move on a root without visits", TG_ERR_STATE);
    add(ERRF_MOVE, "move is not a child of the root / has no policy index", TG_ERR_ILLEGAL_MOVE);
    add(ERRF_NAN, "NaN upper confidence bound (reference: \"tried comparing nan\")", TG_ERR_NAN);
    add(ERRF_ARENA, "MCTS node pool exhausted (raise TgSearchConfig.arena_nodes)", TG_ERR_ARENA_OVERFLOW);
    return fail(code, msg);
}

int search_poll_errors(TgEngine* e) {
    if (int rc = net_poll_errors(e)) return rc;
    if (!e || !e->search) return TG_OK;
    uint32_t bits = 0;
    TG_HIP(hipMemcpy(&bits, e->search->err.p, 4, hipMemcpyDeviceToHost));
    if (!bits) return TG_OK;
    return describe_errors(bits);
}

static int need_search(TgEngine* e) {
    if (!e) return fail(TG_ERR_INVALID_ARG, "null engine");
    if (!e->search) return fail(TG_ERR_STATE, "tg_search_create / tg_selfplay_create has not been called");
    TG_HIP(hipSetDevice(e->cfg.device));
    return TG_OK;
}

static int upload_mask(TgEngine* e, const uint8_t* active, const uint8_t** d_out) {
    *d_out = nullptr;
    if (!active) return TG_OK;
    Search* s = e->search;
    TG_HIP(hipMemcpyAsync(s->active.p, active, (size_t)s->d.G, hipMemcpyHostToDevice, e->stream));
    *d_out = s->active.as<uint8_t>();
    return TG_OK;
}

// Every chunk but chunk 0 free, no game owning one: the state after create and after tg_search_reset (which drops every
// tree, and with them a pool that an exhaustion has left inconsistent).  search_reset_trees then gives each game its root chunk.
static int pool_init(TgEngine* e, Search* s) {
    const size_t G = (size_t)s->cfg.games, n_chunks = s->d.n_chunks;
    const int chunk_shift = s->d.chunk_shift;
    std::vector<uint32_t> ring(n_chunks, 0u);
    for (size_t i = 0; i + 1 < n_chunks; i++) ring[i] = (uint32_t)(i + 1);
    const unsigned long long ctl[4] = {0ull, (unsigned long long)(n_chunks - 1), (unsigned long long)(n_chunks - 1), 0ull};
    // everything on the engine's (non-blocking) stream: a legacy-stream hipMemset would not be ordered against the
    // kernels that follow on it
    hipStream_t st = e->stream;
    TG_HIP(hipMemcpyAsync(s->free_ring.p, ring.data(), n_chunks * 4, hipMemcpyHostToDevice, st));
    TG_HIP(hipMemcpyAsync(s->pool_ctl.p, ctl, sizeof ctl, hipMemcpyHostToDevice, st));
    TG_HIP(hipMemsetAsync(s->chunk_head.p, 0, G * 4, st));
    TG_HIP(hipMemsetAsync(s->chunk_link.p, 0, n_chunks * 4, st));
    TG_HIP(hipMemsetAsync(s->root.p, 0, G * 4, st));
    TG_HIP(hipMemsetAsync(s->alloc.p, 0, G * 8, st));
    // chunk 0 is where a not-yet-reset game's root index (0) points: an empty node without children
    TG_HIP(hipMemsetAsync(s->hot.p, 0, sizeof(NodeHot) << chunk_shift, st));
    TG_HIP(hipMemsetAsync(s->cold.p, 0, sizeof(NodeCold) << chunk_shift, st));
    TG_HIP(hipStreamSynchronize(st));  // `ring` / `ctl` are host temporaries
    return TG_OK;
}

static int search_alloc(TgEngine* e, const TgSearchConfig* cfg) {
    if (!e) return fail(TG_ERR_INVALID_ARG, "null engine");
    if (!cfg || cfg->games <= 0 || cfg->games > e->cfg.max_batch) return fail(TG_ERR_INVALID_ARG, "games must be in 1..max_batch");
    const size_t B = cfg->batch ? cfg->batch : 1;  // virtual rollouts per tree and iteration
    if (B > 4096 || (e->cfg.evaluator == TG_EVAL_RESNET && (size_t)cfg->games * B > (size_t)e->cfg.max_batch))
        return fail(TG_ERR_INVALID_ARG, "games x batch leaves per iteration exceed max_batch");
    if (cfg->arena_nodes != 0 && cfg->arena_nodes < 1024) return fail(TG_ERR_INVALID_ARG, "arena_nodes must be 0 (auto) or at least 1024");
    if (cfg->visit_limit < 0 || (cfg->visit_limit != 0 && cfg->visit_limit < 16) || cfg->visit_limit > TG_LIMIT_VISITS)
        return fail(TG_ERR_INVALID_ARG, "visit_limit must be 0 (= TG_LIMIT_VISITS) or in 16..TG_LIMIT_VISITS");
    if (e->cfg.evaluator == TG_EVAL_RESNET && !net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    TG_HIP(hipSetDevice(e->cfg.device));
    if (e->search) {
        TG_HIP(hipStreamSynchronize(e->stream));
        net_set_gather(e, nullptr);  // (its buffers go with the search)
        search_destroy(e->search);
        e->search = nullptr;
    }
    std::unique_ptr<Search> sp(new Search());
    Search* s = sp.get();
    s->cfg = *cfg;
    // One node pool for all games (search.cuh).  arena_nodes is the AVERAGE budget per game: pool = games × arena_nodes,
    // and a single tree may grow far beyond it while others are small.  0 = auto: half of the free device memory,
    // at most 2^22 nodes per game (the reference's 10 000 rollouts on 6×6 peak at ≈ 2^20 per game) and at most what a
    // 32-bit node index addresses.
    const size_t node_bytes = sizeof(NodeHot) + sizeof(NodeCold);
    size_t pool_nodes;
    if (s->cfg.arena_nodes == 0) {
        size_t free_b = 0, total_b = 0;
        TG_HIP(hipMemGetInfo(&free_b, &total_b));
        // (a 32-bit node index addresses 2^32 nodes; the slack chunks added below count against that)
        const size_t index_limit = ((size_t)1 << 32) - (((size_t)3 * cfg->games + 2) << 11);
        pool_nodes = std::min<size_t>({free_b / 2 / node_bytes, (size_t)cfg->games << 22, index_limit});
        pool_nodes = std::max<size_t>(pool_nodes, (size_t)cfg->games << 12);
        s->cfg.arena_nodes = (int32_t)std::min<size_t>(pool_nodes / (size_t)cfg->games, (size_t)1 << 30);
    } else {
        pool_nodes = (size_t)cfg->games * (size_t)cfg->arena_nodes;
        if (pool_nodes > ((size_t)1 << 32) - (((size_t)3 * cfg->games + 2) << 11))
            return fail(TG_ERR_INVALID_ARG, "games x arena_nodes exceeds the 2^32 nodes a pool can hold");
    }
    cfg = &s->cfg;
    const size_t G = (size_t)cfg->games;
    const int chunk_shift = pool_nodes / G >= 16384 ? 11 : 10;
    // + 3 chunks per game: the open chunk's unused tail, the first chunk of a re-rooted copy while the old tree still
    // holds its own, and chunk 0, which is never handed out (index 0 = "no children")
    const size_t n_chunks = (pool_nodes >> chunk_shift) + 3 * G + 1;
    if ((n_chunks << chunk_shift) > ((size_t)1 << 32)) return fail(TG_ERR_INVALID_ARG, "node pool exceeds 2^32 nodes");
    const int cin_pad = e->cin_pad;
    TG_HIP(s->hot.ensure((n_chunks << chunk_shift) * sizeof(NodeHot)));
    TG_HIP(s->cold.ensure((n_chunks << chunk_shift) * sizeof(NodeCold)));
    TG_HIP(s->root.ensure(G * 4));
    TG_HIP(s->alloc.ensure(G * 8));
    TG_HIP(s->chunk_head.ensure(G * 4));
    TG_HIP(s->chunk_link.ensure(n_chunks * 4));
    TG_HIP(s->chunk_fwd.ensure(n_chunks * 4));
    TG_HIP(s->chunk_used.ensure(n_chunks * 4));
    TG_HIP(s->free_ring.ensure(n_chunks * 4));
    TG_HIP(s->pool_ctl.ensure(4 * 8));
    TG_HIP(s->root_state.ensure(G * e->g.bytes));
    TG_HIP(s->alive.ensure(G));
    TG_HIP(s->abort.ensure(G));
    TG_HIP(s->generation.ensure(G * 4));
    TG_HIP(s->path_len.ensure(G * B * 4));
    TG_HIP(s->path.ensure(G * B * MAX_DEPTH * 4));
    TG_HIP(s->leaf_kind.ensure(G * B));
    TG_HIP(s->leaf_rec.ensure(G * B * 8));
    TG_HIP(s->child_pidx.ensure(G * B * EX_MOVES * 2));
    TG_HIP(s->child_logit.ensure(G * B * EX_MOVES * 4));
    TG_HIP(hipMemsetAsync(s->leaf_rec.p, 0, G * B * 8, e->stream));
    TG_HIP(s->leaf_hash.ensure(G * B * 8));
    TG_HIP(s->op.ensure(G * 4));
    TG_HIP(s->active.ensure(G));
    TG_HIP(s->noise.ensure(G * EX_MOVES * 4));
    TG_HIP(s->err.ensure(4));
    TG_HIP(s->counters.ensure(G * 16));
    TG_HIP(s->r_moves.ensure(G * EX_MOVES * 2));
    TG_HIP(s->r_visits.ensure(G * EX_MOVES * 4));
    TG_HIP(s->r_prior.ensure(G * EX_MOVES * 4));
    TG_HIP(s->r_q.ensure(G * EX_MOVES * 4));
    TG_HIP(s->r_counts.ensure(G * 4));
    TG_HIP(s->r_rv.ensure(G * 4));
    TG_HIP(s->r_rq.ensure(G * 4));
    TG_HIP(s->s_moves.ensure(G * 2));
    if (e->cfg.evaluator == TG_EVAL_RESNET) {
        if (net_takes_states(e)) {
            TG_HIP(s->leaf_state.ensure(G * B * e->g.bytes));
            TG_HIP(hipMemsetAsync(s->leaf_state.p, 0, s->leaf_state.bytes, e->stream));
        } else
        TG_HIP(s->planes.ensure(G * B * e->g.nsq * cin_pad * 4));
        TG_HIP(s->policy.ensure(G * B * (size_t)e->policy_size * 4));
        TG_HIP(s->eval.ensure(G * B * 4));
        if (s->planes.p) TG_HIP(hipMemsetAsync(s->planes.p, 0, s->planes.bytes, e->stream));
    }
    // exploration_rate(n) = ln((1 + n + base) / base) + init for integer visit counts (mcts.rs:10-12),
    // evaluated once on the host in f32 so that every GPU and the CPU agree on the last bit
    const int ctab_size = cfg->visit_limit ? cfg->visit_limit : TG_LIMIT_VISITS;
    std::vector<float> ctab(ctab_size);
    for (int i = 0; i < ctab_size; i++) {
        float nf = (float)i;
        ctab[i] = logf((1.0f + nf + cfg->exploration_base) / cfg->exploration_base) + cfg->exploration_init;
    }
    TG_HIP(s->ctab.ensure((size_t)ctab_size * 4));
    TG_HIP(hipMemcpy(s->ctab.p, ctab.data(), (size_t)ctab_size * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemsetAsync(s->err.p, 0, 4, e->stream));
    TG_HIP(hipMemsetAsync(s->counters.p, 0, G * 16, e->stream));
    TG_HIP(hipMemsetAsync(s->generation.p, 0, G * 4, e->stream));
    TG_HIP(hipMemsetAsync(s->alive.p, 0, G, e->stream));
    TG_HIP(hipMemsetAsync(s->abort.p, 0, G, e->stream));
    SearchDev& d = s->d;
    d.hot = s->hot.as<NodeHot>(); d.cold = s->cold.as<NodeCold>(); d.root = s->root.as<uint32_t>(); d.alloc = s->alloc.as<uint32_t>();
    d.chunk_head = s->chunk_head.as<uint32_t>(); d.chunk_link = s->chunk_link.as<uint32_t>(); d.chunk_fwd = s->chunk_fwd.as<uint32_t>();
    d.chunk_used = s->chunk_used.as<uint32_t>(); d.free_ring = s->free_ring.as<uint32_t>();
    d.pool_ctl = s->pool_ctl.as<unsigned long long>(); d.n_chunks = (uint32_t)n_chunks; d.chunk_shift = chunk_shift;
    d.root_state = s->root_state.as<uint8_t>(); d.alive = s->alive.as<uint8_t>(); d.abort = s->abort.as<uint8_t>(); d.generation = s->generation.as<uint32_t>();
    d.path_len = s->path_len.as<int32_t>(); d.path = s->path.as<uint32_t>(); d.leaf_kind = s->leaf_kind.as<uint8_t>();
    d.leaf_rec = s->leaf_rec.as<uint32_t>(); d.child_pidx = s->child_pidx.as<uint16_t>();
    d.leaf_hash = s->leaf_hash.as<uint64_t>(); d.planes = s->planes.as<float>(); d.leaf_state = s->leaf_state.as<uint8_t>();
    d.policy = s->policy.as<float>();
    d.eval = s->eval.as<float>(); d.ctab = s->ctab.as<float>(); d.lut5 = e->lut5.as<int16_t>(); d.err = s->err.as<uint32_t>();
    d.counters = s->counters.as<unsigned long long>();
    {
        int prc = pool_init(e, s);
        if (prc) return prc;
    }
    d.G = cfg->games; d.n = e->g.n; d.cin_pad = cin_pad; d.P = e->policy_size; d.ctab_size = ctab_size;
    d.legacy5 = e->legacy5 ? 1 : 0; d.evaluator = e->cfg.evaluator; d.slot_base = cfg->slot_base; d.seed = cfg->seed;
    d.batch = (int)B; d.pass = 0;
    d.retire = 0;  // tg_selfplay_create turns it on: only the self-play driver can restart a game on its own
    d.logits = nullptr; d.logit_ld = 0; d.fc_stats = nullptr; d.fc_blocks = 0; d.fc_stride = 0; d.child_logit = nullptr;  // refreshed before every iteration (bind_logits)
    e->search = sp.release();
    return TG_OK;
}

// every tree = Node::default(), every game alive with the given root state
static int search_reset_trees(TgEngine* e) {
    Search* s = e->search;
    const size_t G = (size_t)s->d.G;
    std::vector<int32_t> op(G, -2);
    TG_HIP(hipMemcpyAsync(s->op.p, op.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    launch_reroot(e->stream, s->d, s->op.as<int32_t>());
    TG_HIP(hipGetLastError());
    TG_HIP(hipStreamSynchronize(e->stream));
    return TG_OK;
}

// FC-head networks: the backup reads the network's logits buffer directly (softmax statistics and tanh in the tree kernel,
// softmax.cuh).  Bound per call: the weights may have been re-finalised (tg_train_commit, another precision) since the
// search was created.
static void bind_logits(TgEngine* e) {
    Search* s = e->search;
    s->d.logits = nullptr;
    s->d.logit_ld = 0;
    s->d.fc_stats = nullptr;
    s->d.fc_blocks = 0;
    s->d.fc_stride = 0;
    s->d.child_logit = nullptr;
    net_set_gather(e, nullptr);
    static const bool off = env_on("TG_DUAL_STREAM") || env_on("TG_NO_FUSED_SOFTMAX");
    if (off || e->cfg.evaluator != TG_EVAL_RESNET) return;
    int ld = 0;
    const float* lg = net_fc_logits(e, &ld);
    if (lg) {
        s->d.logits = lg; s->d.logit_ld = ld;
        int blocks = 0, stride = 0;
        const float* fs = net_fc_stats(e, &blocks, &stride);
        if (fs) { s->d.fc_stats = fs; s->d.fc_blocks = blocks; s->d.fc_stride = stride; }
        // exact-f32 FC at a batch its ring kernel serves: the FC's epilogue hands the backup the children's logits directly
        if (fs && !s->d.planes && net_gather_ok(e, s->d.G * s->d.batch)) {
            const FcGatherArgs g{s->d.child_pidx, s->d.leaf_rec, s->child_logit.as<float>(), EX_MOVES};
            net_set_gather(e, &g);
            s->d.child_logit = s->child_logit.as<float>();
        }
    }
}

// The gather target is armed for the iterations of ONE call and disarmed when it returns: a later logits-only forward by anyone
// else must not find the search's buffers behind a stale pointer.
struct GatherScope {
    TgEngine* e;
    ~GatherScope() { net_set_gather(e, nullptr); }
};

// one lock-step iteration: the body of train/src/self_play.rs:181-210
static int search_iterate(TgEngine* e, const uint8_t* d_active) {
    Search* s = e->search;
    bind_logits(e);
    GatherScope scope{e};
    // `batch` virtual rollouts per tree (Player's batching model, alpha-tak/src/player.rs:77-93) run back to back inside the
    // select kernel (a game's tree belongs to one wave), then ONE network batch of games × batch leaves, then the
    // de-virtualisations in the same order inside the backup kernel
    SearchDev d = s->d;
    d.pass = s->d.batch > 1 ? -1 : 0;  // -1: the kernel runs the `batch` passes of a game back to back in that game's wave
    launch_select(e->stream, d, d_active);
    TG_HIP(hipGetLastError());
    if (e->cfg.evaluator == TG_EVAL_RESNET) {
        const int leaves = s->d.G * s->d.batch;
        float* pol = s->d.logits ? nullptr : s->d.policy;  // logits mode: the backup takes softmax / tanh itself
        int rc = s->d.planes ? net_forward_dev(e, leaves, s->d.planes, pol, s->d.eval)
                             : net_forward_states_dev(e, leaves, s->d.leaf_state, pol, s->d.eval);
        if (rc) return rc;
    }
    launch_backup(e->stream, d);
    TG_HIP(hipGetLastError());
    return TG_OK;
}

// Rollouts on two streams (opt-in, TG_DUAL_STREAM=1): the games are split into two halves that run the select → network → backup chain
// independently (they share nothing but read-only weights), so the latency-bound tree kernels of one half overlap the
// MFMA kernels of the other.  Per-game results do not depend on the batch a position is evaluated in, so trees are
// identical to the single-stream schedule.  An iteration the profiler samples runs alone on the engine stream.
static SearchDev half_view(const SearchDev& d, int g0, int count, size_t state_bytes) {
    SearchDev v = d;  // the node pool is shared: only the per-game arrays are offset
    v.root += g0; v.alloc += 2 * (size_t)g0; v.chunk_head += g0;
    v.root_state += (size_t)g0 * state_bytes; v.alive += g0; v.generation += g0;
    v.path_len += g0; v.path += (size_t)g0 * MAX_DEPTH; v.leaf_kind += g0; v.leaf_hash += g0;
    v.leaf_rec += 2 * (size_t)g0; v.child_pidx += (size_t)g0 * EX_MOVES; v.abort += g0;  // (dual stream runs with batch 1)
    v.leaf_state += (size_t)g0 * state_bytes; v.policy += (size_t)g0 * d.P; v.eval += g0;
    v.counters += (size_t)g0 * 2;
    v.slot_base += (uint32_t)g0;
    v.G = count;
    return v;
}

static bool dual_stream_ok(TgEngine* e) {
    // measured on MI355X at C2: no gain (exact f32 4.06 M vs 4.11 M expansions/s single-stream, bf16x3 11.8 M vs 12.4 M) —
    // the half-batch MFMA kernels fill the chip less well than they overlap; kept opt-in
    static const bool on = env_on("TG_DUAL_STREAM");
    Search* s = e->search;
    return on && s->d.batch == 1 && e->cfg.evaluator == TG_EVAL_RESNET && !s->d.planes && net_takes_states(e) && s->d.G >= 512;
}

static int search_iterate_many(TgEngine* e, int iters) {
    Search* s = e->search;
    if (iters <= 0) return TG_OK;
    bind_logits(e);
    GatherScope scope{e};
    if (!dual_stream_ok(e)) {
        if (s->d.batch == 1 && iters > 1 && !env_on("TG_NO_FUSED_BACKUP_SELECT")) {
            // select(0) | net | backup(0)+select(1) | net | … | backup(iters-1): one tree kernel per iteration
            SearchDev d = s->d;
            d.pass = 0;
            launch_select(e->stream, d, nullptr);
            for (int i = 0; i < iters; i++) {
                if (e->cfg.evaluator == TG_EVAL_RESNET) {
                    float* pol = s->d.logits ? nullptr : s->d.policy;
                    int rc = s->d.planes ? net_forward_dev(e, s->d.G, s->d.planes, pol, s->d.eval)
                                         : net_forward_states_dev(e, s->d.G, s->d.leaf_state, pol, s->d.eval);
                    if (rc) return rc;
                }
                if (i + 1 < iters) launch_backup_select(e->stream, d);
                else launch_backup(e->stream, d);
            }
            TG_HIP(hipGetLastError());
            return TG_OK;
        }
        for (int i = 0; i < iters; i++) {
            int rc = search_iterate(e, nullptr);
            if (rc) return rc;
        }
        return TG_OK;
    }
    for (int h = 0; h < 2; h++)
        if (!e->half_stream[h]) TG_HIP(hipStreamCreateWithFlags(&e->half_stream[h], hipStreamNonBlocking));
    for (int k = 0; k < 3; k++)
        if (!e->half_event[k]) TG_HIP(hipEventCreateWithFlags(&e->half_event[k], hipEventDisableTiming));
    const int G = s->d.G;
    const int g0 = (G / 2 + 15) / 16 * 16;  // whole workgroups of the tower in both halves
    const SearchDev view[2] = {half_view(s->d, 0, g0, e->g.bytes), half_view(s->d, g0, G - g0, e->g.bytes)};
    const int first[2] = {0, g0};
    bool forked = false;
    auto fork = [&]() -> int {
        TG_HIP(hipEventRecord(e->half_event[2], e->stream));
        for (int h = 0; h < 2; h++) TG_HIP(hipStreamWaitEvent(e->half_stream[h], e->half_event[2], 0));
        forked = true;
        return TG_OK;
    };
    auto join = [&]() -> int {
        for (int h = 0; h < 2; h++) {
            TG_HIP(hipEventRecord(e->half_event[h], e->half_stream[h]));
            TG_HIP(hipStreamWaitEvent(e->stream, e->half_event[h], 0));
        }
        forked = false;
        return TG_OK;
    };
    for (int i = 0; i < iters; i++) {
        if (net_profile_due(e)) {  // timed alone, whole batch, on the engine stream
            if (forked) { int rc = join(); if (rc) return rc; }
            int rc = search_iterate(e, nullptr);
            if (rc) return rc;
            continue;
        }
        if (!forked) { int rc = fork(); if (rc) return rc; }
        net_profile_skip(e);
        for (int h = 0; h < 2; h++) {
            hipStream_t st = e->half_stream[h];
            launch_select(st, view[h], nullptr);
            int rc = net_forward_states_at(e, view[h].G, view[h].leaf_state, view[h].policy, view[h].eval, st, first[h]);
            if (rc) return rc;
            launch_backup(st, view[h]);
        }
        TG_HIP(hipGetLastError());
    }
    if (forked) return join();
    return TG_OK;
}

static int read_counters(TgEngine* e, unsigned long long* expansions, unsigned long long* evals) {
    Search* s = e->search;
    std::vector<unsigned long long> h((size_t)s->d.G * 2);
    TG_HIP(hipMemcpy(h.data(), s->counters.p, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long a = 0, b = 0;
    for (int g = 0; g < s->d.G; g++) { a += h[2 * (size_t)g]; b += h[2 * (size_t)g + 1]; }
    *expansions = a;
    *evals = b;
    return TG_OK;
}

static int sync_and_check(TgEngine* e) {
    TG_HIP(hipStreamSynchronize(e->stream));
    return search_poll_errors(e);
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_search_create(TgEngine* e, const TgSearchConfig* cfg) {
    int rc = search_alloc(e, cfg);
    if (rc) return rc;
    return TG_OK;
}

int tg_search_reset(TgEngine* e, const void* states) {
    int rc = need_search(e);
    if (rc) return rc;
    if (!states) return fail(TG_ERR_INVALID_ARG, "tg_search_reset: null states");
    Search* s = e->search;
    rc = validate_states(e, s->d.G, (const uint8_t*)states, "tg_search_reset");
    if (rc) return rc;
    const size_t G = (size_t)s->d.G;
    rc = pool_init(e, s);
    if (rc) return rc;
    TG_HIP(hipMemcpyAsync(s->root_state.p, states, G * e->g.bytes, hipMemcpyHostToDevice, e->stream));
    TG_HIP(hipMemsetAsync(s->alive.p, 1, G, e->stream));
    TG_HIP(hipMemsetAsync(s->err.p, 0, 4, e->stream));
    return search_reset_trees(e);
}

int tg_search_run(TgEngine* e, int iters, const uint8_t* active) {
    int rc = need_search(e);
    if (rc) return rc;
    if (iters < 0) return fail(TG_ERR_INVALID_ARG, "tg_search_run: negative iters");
    const uint8_t* d_active;
    rc = upload_mask(e, active, &d_active);
    if (rc) return rc;
    if (!d_active) rc = search_iterate_many(e, iters);
    else
        for (int i = 0; i < iters && !rc; i++) rc = search_iterate(e, d_active);
    if (rc) return rc;
    if (active) TG_HIP(hipStreamSynchronize(e->stream));  // the staged mask must outlive the launches
    return TG_OK;
}

int tg_search_apply_dirichlet(TgEngine* e, float alpha, float ratio, const uint8_t* active) {
    int rc = need_search(e);
    if (rc) return rc;
    const uint8_t* d_active;
    rc = upload_mask(e, active, &d_active);
    if (rc) return rc;
    launch_dirichlet(e->stream, e->search->d, d_active, alpha, ratio);
    TG_HIP(hipGetLastError());
    return sync_and_check(e);
}

int tg_search_apply_noise(TgEngine* e, const float* noise, float ratio, const uint8_t* active) {
    int rc = need_search(e);
    if (rc) return rc;
    if (!noise) return fail(TG_ERR_INVALID_ARG, "tg_search_apply_noise: null noise");
    Search* s = e->search;
    const uint8_t* d_active;
    rc = upload_mask(e, active, &d_active);
    if (rc) return rc;
    TG_HIP(hipMemcpyAsync(s->noise.p, noise, (size_t)s->d.G * EX_MOVES * 4, hipMemcpyHostToDevice, e->stream));
    launch_apply_noise(e->stream, s->d, d_active, s->noise.as<float>(), ratio);
    TG_HIP(hipGetLastError());
    return sync_and_check(e);
}

int tg_search_root(TgEngine* e, TgMove* moves, uint32_t* visits, float* prior, float* q, int32_t* counts, uint32_t* root_visits,
                   float* root_q) {
    int rc = need_search(e);
    if (rc) return rc;
    Search* s = e->search;
    const size_t G = (size_t)s->d.G;
    TG_HIP(hipMemsetAsync(s->r_moves.p, 0, G * EX_MOVES * 2, e->stream));
    TG_HIP(hipMemsetAsync(s->r_visits.p, 0, G * EX_MOVES * 4, e->stream));
    TG_HIP(hipMemsetAsync(s->r_prior.p, 0, G * EX_MOVES * 4, e->stream));
    TG_HIP(hipMemsetAsync(s->r_q.p, 0, G * EX_MOVES * 4, e->stream));
    launch_root_stats(e->stream, s->d, s->r_moves.as<uint16_t>(), s->r_visits.as<uint32_t>(), s->r_prior.as<float>(), s->r_q.as<float>(),
                      s->r_counts.as<int32_t>(), s->r_rv.as<uint32_t>(), s->r_rq.as<float>());
    TG_HIP(hipGetLastError());
    rc = sync_and_check(e);
    if (rc) return rc;
    if (moves) TG_HIP(hipMemcpy(moves, s->r_moves.p, G * EX_MOVES * 2, hipMemcpyDeviceToHost));
    if (visits) TG_HIP(hipMemcpy(visits, s->r_visits.p, G * EX_MOVES * 4, hipMemcpyDeviceToHost));
    if (prior) TG_HIP(hipMemcpy(prior, s->r_prior.p, G * EX_MOVES * 4, hipMemcpyDeviceToHost));
    if (q) TG_HIP(hipMemcpy(q, s->r_q.p, G * EX_MOVES * 4, hipMemcpyDeviceToHost));
    if (counts) TG_HIP(hipMemcpy(counts, s->r_counts.p, G * 4, hipMemcpyDeviceToHost));
    if (root_visits) TG_HIP(hipMemcpy(root_visits, s->r_rv.p, G * 4, hipMemcpyDeviceToHost));
    if (root_q) TG_HIP(hipMemcpy(root_q, s->r_rq.p, G * 4, hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_search_play(TgEngine* e, const TgMove* moves, const uint8_t* active) {
    int rc = need_search(e);
    if (rc) return rc;
    if (!moves) return fail(TG_ERR_INVALID_ARG, "tg_search_play: null moves");
    Search* s = e->search;
    const uint8_t* d_active;
    rc = upload_mask(e, active, &d_active);
    if (rc) return rc;
    TG_HIP(hipMemcpyAsync(s->s_moves.p, moves, (size_t)s->d.G * 2, hipMemcpyHostToDevice, e->stream));
    launch_play_move(e->stream, s->d, s->s_moves.as<uint16_t>(), d_active, s->op.as<int32_t>());
    launch_reroot(e->stream, s->d, s->op.as<int32_t>());
    TG_HIP(hipGetLastError());
    return sync_and_check(e);
}

int tg_search_states(TgEngine* e, void* states) {
    int rc = need_search(e);
    if (rc) return rc;
    if (!states) return fail(TG_ERR_INVALID_ARG, "tg_search_states: null states");
    rc = sync_and_check(e);
    if (rc) return rc;
    TG_HIP(hipMemcpy(states, e->search->root_state.p, (size_t)e->search->d.G * e->g.bytes, hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_search_dump(TgEngine* e, int game, TgNodeRecord* records, size_t capacity, size_t* n_records) {
    int rc = need_search(e);
    if (rc) return rc;
    Search* s = e->search;
    if (game < 0 || game >= s->d.G || !n_records) return fail(TG_ERR_INVALID_ARG, "tg_search_dump: bad arguments");
    rc = sync_and_check(e);
    if (rc) return rc;
    // gather the game's chunks (its chain through chunk_link) and index the tree by pool index
    uint32_t root = 0, head = 0;
    TG_HIP(hipMemcpy(&root, s->root.as<uint32_t>() + game, 4, hipMemcpyDeviceToHost));
    TG_HIP(hipMemcpy(&head, s->chunk_head.as<uint32_t>() + game, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> link(s->d.n_chunks);
    TG_HIP(hipMemcpy(link.data(), s->chunk_link.p, (size_t)s->d.n_chunks * 4, hipMemcpyDeviceToHost));
    const int sh = s->d.chunk_shift;
    const size_t CH = (size_t)1 << sh;
    std::map<uint32_t, size_t> where;  // chunk id → position in the host copies
    std::vector<NodeHot> hot_h;
    std::vector<NodeCold> cold_h;
    for (uint32_t c = head; c != 0; c = link[c]) {
        if (c >= s->d.n_chunks || where.count(c)) return fail(TG_ERR_STATE, "tg_search_dump: corrupt chunk chain");
        where[c] = hot_h.size();
        hot_h.resize(hot_h.size() + CH);
        cold_h.resize(cold_h.size() + CH);
        TG_HIP(hipMemcpy(hot_h.data() + where[c], s->hot.as<NodeHot>() + ((size_t)c << sh), CH * sizeof(NodeHot), hipMemcpyDeviceToHost));
        TG_HIP(hipMemcpy(cold_h.data() + where[c], s->cold.as<NodeCold>() + ((size_t)c << sh), CH * sizeof(NodeCold), hipMemcpyDeviceToHost));
    }
    bool corrupt = false;
    auto at = [&](uint32_t nd) -> size_t {
        auto it = where.find(nd >> sh);
        if (it == where.end()) { corrupt = true; return 0; }
        return it->second + (nd & (CH - 1));
    };
    if (where.empty()) return fail(TG_ERR_STATE, "tg_search_dump: the game has no tree");
    // depth-first in child order; uninitialised children become leaf records with n_children = 0xFFFF
    std::vector<TgNodeRecord> out;
    struct Frame { uint32_t node; uint32_t next; };
    std::vector<Frame> stack;
    auto emit = [&](uint32_t nd, bool is_root) {
        const NodeHot& h = hot_h[at(nd)];
        const NodeCold& c = cold_h[at(nd)];
        TgNodeRecord r;
        r.move = is_root ? 0 : c.mv;
        r.n_children = (uint16_t)(c.nres & 0xfff);
        r.visits = h.visits;
        r.virtual_visits = h.virt;
        r.result = c.nres >> 12;
        std::memcpy(&r.prior_bits, &h.prior, 4);
        std::memcpy(&r.q_bits, &h.q, 4);
        out.push_back(r);
    };
    emit(root, true);
    stack.push_back({root, 0u});
    while (!stack.empty() && !corrupt) {
        Frame& f = stack.back();
        const NodeCold& fc = cold_h[at(f.node)];
        uint32_t nch = fc.nres & 0xfff;
        if (f.next >= nch) { stack.pop_back(); continue; }
        uint32_t c = fc.child + f.next;
        f.next++;
        const NodeHot& ch = hot_h[at(c)];
        if (corrupt) break;
        if (ch.visits != 0 || ch.virt != 0) {
            emit(c, false);
            stack.push_back({c, 0u});
        } else {
            TgNodeRecord r;
            r.move = cold_h[at(c)].mv;
            r.n_children = 0xFFFF;
            r.visits = 0; r.virtual_visits = 0; r.result = 0;
            std::memcpy(&r.prior_bits, &ch.prior, 4);
            std::memcpy(&r.q_bits, &ch.q, 4);
            out.push_back(r);
        }
    }
    if (corrupt) return fail(TG_ERR_STATE, "tg_search_dump: corrupt tree (a child index points outside the game's chunks)");
    *n_records = out.size();
    if (out.size() > capacity || (!records && !out.empty())) return fail(TG_ERR_INVALID_ARG, "tg_search_dump: capacity too small");
    std::memcpy(records, out.data(), out.size() * sizeof(TgNodeRecord));
    return TG_OK;
}

int tg_search_counters(TgEngine* e, uint64_t* expansions, uint64_t* evals) {
    int rc = need_search(e);
    if (rc) return rc;
    rc = sync_and_check(e);
    if (rc) return rc;
    unsigned long long c[2];
    rc = read_counters(e, &c[0], &c[1]);
    if (rc) return rc;
    if (expansions) *expansions = c[0];
    if (evals) *evals = c[1];
    return TG_OK;
}

int tg_search_pool(TgEngine* e, uint64_t* nodes_total, uint64_t* nodes_in_use, uint64_t* nodes_peak) {
    int rc = need_search(e);
    if (rc) return rc;
    rc = sync_and_check(e);
    if (rc) return rc;
    Search* s = e->search;
    unsigned long long ctl[4];
    TG_HIP(hipMemcpy(ctl, s->pool_ctl.p, sizeof ctl, hipMemcpyDeviceToHost));
    // chunks handed out so far minus chunks returned beyond the initial fill of the ring = chunks owned by trees
    const unsigned long long initial = (unsigned long long)s->d.n_chunks - 1;
    const unsigned long long owned = ctl[0] - (ctl[1] - initial);
    if (nodes_total) *nodes_total = (uint64_t)initial << s->d.chunk_shift;
    if (nodes_in_use) *nodes_in_use = (uint64_t)owned << s->d.chunk_shift;
    if (nodes_peak) *nodes_peak = (uint64_t)ctl[3] << s->d.chunk_shift;
    return TG_OK;
}

// ---- self-play -----------------------------------------------------------------------------------

int tg_selfplay_create(TgEngine* e, const TgSearchConfig* scfg, const TgSelfPlayConfig* cfg) {
    if (!cfg || !scfg) return fail(TG_ERR_INVALID_ARG, "null self-play config");
    if (cfg->rollouts < 1 || cfg->max_examples < 1) return fail(TG_ERR_INVALID_ARG, "rollouts and max_examples must be positive");
    if (cfg->max_game_plies < 0 || cfg->max_game_plies > TG_LIMIT_GAME_PLIES)
        return fail(TG_ERR_INVALID_ARG, "max_game_plies must be 0 (= TG_LIMIT_GAME_PLIES) or in 1..TG_LIMIT_GAME_PLIES");
    TgSearchConfig sc1 = *scfg;
    sc1.batch = 1;  // self_play_parallel gathers ONE leaf per game and iteration (self_play.rs:181-210)
    int rc = search_alloc(e, &sc1);
    if (rc) return rc;
    Search* s = e->search;
    s->selfplay = true;
    s->spcfg = *cfg;
    s->d.retire = 1;
    const size_t G = (size_t)s->d.G, sb = (size_t)e->g.bytes;
    const int epg = TG_LIMIT_GAME_PLIES;
    const size_t ME = (size_t)cfg->max_examples;
    TG_HIP(s->st_hdr.ensure(G * epg * sizeof(ExampleRec)));
    TG_HIP(s->st_state.ensure(G * epg * sb));
    TG_HIP(s->st_moves.ensure(G * epg * EX_MOVES * 2));
    TG_HIP(s->st_visits.ensure(G * epg * EX_MOVES * 4));
    TG_HIP(s->st_count.ensure(G * 4));
    TG_HIP(s->out_hdr.ensure(ME * sizeof(ExampleRec)));
    TG_HIP(s->out_state.ensure(ME * sb));
    TG_HIP(s->out_moves.ensure(ME * EX_MOVES * 2));
    TG_HIP(s->out_visits.ensure(ME * EX_MOVES * 4));
    TG_HIP(s->fin.ensure(G));
    TG_HIP(s->recycle.ensure(G));
    TG_HIP(s->out_off.ensure(G * 4));
    TG_HIP(s->chosen.ensure(G * 4));
    TG_HIP(s->mask.ensure(G));
    TG_HIP(s->stats.ensure(ST_COUNT * 8));
    TG_HIP(hipMemsetAsync(s->st_count.p, 0, G * 4, e->stream));
    TG_HIP(hipMemsetAsync(s->stats.p, 0, ST_COUNT * 8, e->stream));
    TG_HIP(hipMemsetAsync(s->fin.p, 0, G, e->stream));
    SelfPlayDev& p = s->p;
    p.st_hdr = s->st_hdr.as<ExampleRec>(); p.st_state = s->st_state.as<uint8_t>(); p.st_moves = s->st_moves.as<uint16_t>();
    p.st_visits = s->st_visits.as<uint32_t>(); p.st_count = s->st_count.as<int32_t>();
    p.out_hdr = s->out_hdr.as<ExampleRec>(); p.out_state = s->out_state.as<uint8_t>(); p.out_moves = s->out_moves.as<uint16_t>();
    p.out_visits = s->out_visits.as<uint32_t>();
    p.fin = s->fin.as<uint8_t>(); p.recycle = s->recycle.as<uint8_t>(); p.out_off = s->out_off.as<uint32_t>();
    p.chosen = s->chosen.as<int32_t>(); p.mask = s->mask.as<uint8_t>(); p.stats = s->stats.as<unsigned long long>();
    p.ex_per_game = epg; p.max_examples = cfg->max_examples;
    p.max_game_plies = cfg->max_game_plies ? cfg->max_game_plies : epg;
    p.rollouts = cfg->rollouts; p.noise_plies = cfg->noise_plies; p.exploit_plies = cfg->exploit_plies; p.komi = cfg->komi;
    p.total_games = cfg->total_games; p.noise_alpha = cfg->noise_alpha; p.noise_ratio = cfg->noise_ratio;
    // games[i] = Game::with_komi(komi), nodes[i] = Node::default()  (self_play.rs:102-103)
    std::vector<uint8_t> start(sb, 0);
    {
        int stones, caps;
        starting_stones(e->g.n, stones, caps);
        TgHeader* h = (TgHeader*)(start.data() + sb - sizeof(TgHeader));
        h->n = (uint8_t)e->g.n; h->to_move = 0; h->ply = 0;
        h->white_stones = h->black_stones = (uint8_t)stones;
        h->white_caps = h->black_caps = (uint8_t)caps;
        h->half_komi = (int8_t)(cfg->komi * 2); h->reversible_plies = 0;
    }
    std::vector<uint8_t> all(G * sb);
    for (size_t g = 0; g < G; g++) std::memcpy(&all[g * sb], start.data(), sb);
    TG_HIP(hipMemcpy(s->root_state.p, all.data(), all.size(), hipMemcpyHostToDevice));
    TG_HIP(hipMemsetAsync(s->alive.p, 1, G, e->stream));
    return search_reset_trees(e);
}

int tg_selfplay_step(TgEngine* e, int plies) {
    int rc = need_search(e);
    if (rc) return rc;
    Search* s = e->search;
    if (!s->selfplay) return fail(TG_ERR_STATE, "tg_selfplay_create has not been called");
    hipStream_t st = e->stream;
    const size_t G = (size_t)s->d.G;
    int32_t* op = s->op.as<int32_t>();
    for (int ply = 0; ply < plies; ply++) {
        launch_sp_opening(st, s->d);                                   // (a) :110-116
        launch_sp_instant_win(st, s->d, s->p);                         // (b) :119-171
        TG_HIP(hipMemsetAsync(op, 0xFF, G * 4, st));
        launch_sp_finish(st, s->d, s->p, op);
        launch_reroot(st, s->d, op);
        launch_sp_noise_mask(st, s->d, s->p);                          // (c) :174-180
        rc = search_iterate(e, s->p.mask);
        if (rc) return rc;
        launch_dirichlet(st, s->d, s->p.mask, s->p.noise_alpha, s->p.noise_ratio);
        rc = search_iterate_many(e, s->p.rollouts);                    // (d) :181-210
        if (rc) return rc;
        launch_sp_pick(st, s->d, s->p, op);                            // (e) :212-258
        launch_sp_finish(st, s->d, s->p, op);
        launch_reroot(st, s->d, op);
        launch_sp_count_ply(st, s->p);
        TG_HIP(hipGetLastError());
    }
    return TG_OK;
}

int tg_selfplay_stats(TgEngine* e, TgSelfPlayStats* out) {
    int rc = need_search(e);
    if (rc) return rc;
    Search* s = e->search;
    if (!s->selfplay || !out) return fail(TG_ERR_STATE, "tg_selfplay_create has not been called");
    rc = sync_and_check(e);
    if (rc) return rc;
    unsigned long long st[ST_COUNT], c[2];
    TG_HIP(hipMemcpy(st, s->stats.p, sizeof st, hipMemcpyDeviceToHost));
    rc = read_counters(e, &c[0], &c[1]);
    if (rc) return rc;
    out->games_finished = st[ST_FINISHED]; out->examples = st[ST_EXAMPLES]; out->plies = st[ST_PLIES];
    out->white_wins = st[ST_WHITE]; out->black_wins = st[ST_BLACK]; out->draws = st[ST_DRAWS]; out->instant_wins = st[ST_INSTANT];
    out->expansions = c[0]; out->evals = c[1];
    out->aborted_games = st[ST_ABORTED];
    {
        std::vector<uint8_t> alive((size_t)s->d.G);
        TG_HIP(hipMemcpy(alive.data(), s->alive.p, alive.size(), hipMemcpyDeviceToHost));
        out->alive_games = 0;
        for (uint8_t a : alive) out->alive_games += a ? 1u : 0u;
    }
    {   // examples the ring has overwritten since the last drain count as dropped as soon as they are observable
        const unsigned long long ME = (unsigned long long)s->p.max_examples;
        unsigned long long lost = st[ST_EXAMPLES] - s->drained > ME ? st[ST_EXAMPLES] - s->drained - ME : 0ull;
        out->dropped_examples = s->dropped + lost;
    }
    return TG_OK;
}

int tg_selfplay_drain(TgEngine* e, int cap, TgExampleHeader* headers, void* states, TgMove* moves, uint32_t* visits, int32_t* n_out) {
    int rc = need_search(e);
    if (rc) return rc;
    Search* s = e->search;
    if (!s->selfplay) return fail(TG_ERR_STATE, "tg_selfplay_create has not been called");
    if (cap < 0 || !n_out || (cap > 0 && (!headers || !states || !moves || !visits))) return fail(TG_ERR_INVALID_ARG, "tg_selfplay_drain: bad arguments");
    rc = sync_and_check(e);
    if (rc) return rc;
    unsigned long long total = 0;
    TG_HIP(hipMemcpy(&total, s->stats.as<unsigned long long>() + ST_EXAMPLES, 8, hipMemcpyDeviceToHost));
    const unsigned long long ME = (unsigned long long)s->p.max_examples;
    if (total - s->drained > ME) {  // older ones were overwritten in the ring: skipped, and counted
        s->dropped += total - ME - s->drained;
        s->drained = total - ME;
    }
    const size_t sb = (size_t)e->g.bytes;
    const unsigned long long avail = total - s->drained;
    const int k = (int)std::min<unsigned long long>((unsigned long long)cap, avail);
    // the k examples are consecutive ring entries: at most two contiguous runs per array (wrap-around), one copy each
    std::vector<ExampleRec> hdr((size_t)k);
    for (int done = 0; done < k;) {
        const size_t o = (size_t)((s->drained + done) % ME);
        const int run = (int)std::min<size_t>((size_t)(k - done), (size_t)ME - o);
        TG_HIP(hipMemcpy(hdr.data() + done, s->out_hdr.as<ExampleRec>() + o, (size_t)run * sizeof(ExampleRec), hipMemcpyDeviceToHost));
        TG_HIP(hipMemcpy((uint8_t*)states + (size_t)done * sb, s->out_state.as<uint8_t>() + o * sb, (size_t)run * sb, hipMemcpyDeviceToHost));
        TG_HIP(hipMemcpy(moves + (size_t)done * EX_MOVES, s->out_moves.as<uint16_t>() + o * EX_MOVES, (size_t)run * EX_MOVES * 2, hipMemcpyDeviceToHost));
        TG_HIP(hipMemcpy(visits + (size_t)done * EX_MOVES, s->out_visits.as<uint32_t>() + o * EX_MOVES, (size_t)run * EX_MOVES * 4, hipMemcpyDeviceToHost));
        done += run;
    }
    for (int i = 0; i < k; i++) {
        headers[i].game_id = hdr[i].slot | (hdr[i].generation << 20);
        headers[i].n_moves = hdr[i].n_moves;
        headers[i].result = hdr[i].result;
        headers[i].reserved = 0;
        // entries past n_moves are whatever an earlier example left in the ring slot: clear them for the caller
        const size_t nm = (size_t)std::min(std::max(hdr[i].n_moves, 0), (int32_t)EX_MOVES);
        std::memset(moves + (size_t)i * EX_MOVES + nm, 0, (EX_MOVES - nm) * 2);
        std::memset(visits + (size_t)i * EX_MOVES + nm, 0, (EX_MOVES - nm) * 4);
    }
    s->drained += (unsigned long long)k;
    *n_out = k;
    return TG_OK;
}

}  // extern "C"
