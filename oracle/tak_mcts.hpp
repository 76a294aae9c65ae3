// ORACLE — TEST INFRASTRUCTURE ONLY (see tak_rules.hpp).
// CPU restatement of the reference's MCTS (alpha-tak/src/search/{node,mcts,play,noise}.rs) and of
// the batched self-play driver (train/src/self_play.rs:96-262): heap-allocated tree of nodes,
// scalar PUCT with virtual loss, one leaf per game per iteration.
//
// The reference draws its randomness from rand::thread_rng (not reproducible).  Here — and in the
// product — every random decision comes from a counter-based generator (Philox4x32-10) keyed by
// (seed; slot, generation, ply, purpose, index), with Dirichlet noise built from f64 arithmetic
// restricted to + - * / so that CPU and GPU agree bit for bit.  This file is an independent
// implementation of that spec; it shares no code with tak_amd/.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "tak_repr.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------
// deterministic random numbers (spec: DESIGN.md §RNG)
// ---------------------------------------------------------------------------------------
struct Philox {
    static void round(uint32_t c[4], const uint32_t k[2]) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static void gen(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t c[4] = {c0, c1, c2, c3};
        for (int r = 0; r < 10; r++) {
            round(c, k);
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; i++) out[i] = c[i];
    }
};
enum : uint32_t { RNG_OPENING = 1, RNG_GAMMA = 2, RNG_PICK = 3 };
inline void rng_draw(uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply, uint32_t purpose,
                     uint32_t index, uint32_t attempt, uint32_t out[4]) {
    Philox::gen(seed, slot, generation, ply | (purpose << 16), index | (attempt << 16), out);
}

inline double bits_to_double(uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; }
inline uint64_t double_to_bits(double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; }

// sqrt / log / exp from + - * / only (deterministic across CPU and GPU; not correctly rounded)
inline double dsqrt(double a) {
    double x = bits_to_double((double_to_bits(a) >> 1) + 0x1FF8000000000000ull);
    for (int i = 0; i < 6; i++) x = 0.5 * (x + a / x);
    return x;
}
inline double dlog(double x) {
    uint64_t b = double_to_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = bits_to_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double t = (m - 1.0) / (m + 1.0);
    double t2 = t * t;
    double p = 1.0 / 21.0;
    p = p * t2 + 1.0 / 19.0;
    p = p * t2 + 1.0 / 17.0;
    p = p * t2 + 1.0 / 15.0;
    p = p * t2 + 1.0 / 13.0;
    p = p * t2 + 1.0 / 11.0;
    p = p * t2 + 1.0 / 9.0;
    p = p * t2 + 1.0 / 7.0;
    p = p * t2 + 1.0 / 5.0;
    p = p * t2 + 1.0 / 3.0;
    p = p * t2 + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * t * p;
}
inline double dexp(double y) {
    if (y < -700.0) return 0.0;
    if (y > 700.0) y = 700.0;
    double t = y * 1.4426950408889634;
    long long k = (long long)(t < 0 ? t - 0.5 : t + 0.5);
    double r = (y - (double)k * 0.6931471803691238) - (double)k * 1.9082149292705877e-10;
    double p = 1.0;
    for (int i = 18; i >= 1; i--) p = 1.0 + p * (r / (double)i);
    return bits_to_double(double_to_bits(p) + ((uint64_t)k << 52));
}
inline double u32_to_unit(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// Gamma(alpha, 1) by Marsaglia–Tsang (with the U^(1/alpha) boost for alpha < 1); all draws for
// child `index` come from rng_draw(..., RNG_GAMMA, index, attempt).
inline double gamma_sample(double alpha, uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply, uint32_t index) {
    double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    double d = a - 1.0 / 3.0;
    double c = 1.0 / dsqrt(9.0 * d);
    for (uint32_t attempt = 0; attempt < 65535; attempt++) {
        uint32_t r[4];
        rng_draw(seed, slot, generation, ply, RNG_GAMMA, index, attempt, r);
        double v1 = 2.0 * u32_to_unit(r[0]) - 1.0;
        double v2 = 2.0 * u32_to_unit(r[1]) - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s >= 1.0 || s == 0.0) continue;
        double x = v1 * dsqrt(-2.0 * dlog(s) / s);
        double w = 1.0 + c * x;
        if (w <= 0.0) continue;
        double v = w * w * w;
        double u = u32_to_unit(r[2]);
        if (dlog(u) < 0.5 * x * x + d - d * v + d * dlog(v)) {
            double g = d * v;
            if (alpha < 1.0) g = g * dexp(dlog(u32_to_unit(r[3])) / alpha);
            return g;
        }
    }
    return d;
}

// ---------------------------------------------------------------------------------------
// evaluators (stand-ins for Network::policy_eval, model/network.rs:34)
// ---------------------------------------------------------------------------------------
// callback: n packed states in, policy n×P and eval n out
typedef void (*EvalFn)(void* ctx, int n, const uint8_t* states, float* policy, float* eval);

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
// hash of the meaningful bytes of a packed state (stack words, meta bytes, first 10 header bytes)
inline uint64_t state_hash(const uint8_t* st, int n) {
    int slots = n <= 5 ? 25 : 36;
    size_t bytes = state_bytes(n);
    uint64_t h = 0x243F6A8885A308D3ull;
    const uint64_t* stack = (const uint64_t*)st;
    for (int i = 0; i < n * n; i++) h = mix64(h ^ stack[i]) + (uint64_t)i;
    const uint8_t* meta = st + 8 * slots;
    for (int i = 0; i < n * n; i++) h = mix64(h ^ ((uint64_t)meta[i] << 8) ^ (uint64_t)(i + 1));
    const uint8_t* hd = st + bytes - sizeof(TgHeader);
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 8; i++) a |= (uint64_t)hd[i] << (8 * i);
    for (int i = 8; i < 10; i++) b |= (uint64_t)hd[i] << (8 * (i - 8));
    h = mix64(h ^ a);
    h = mix64(h ^ b);
    return h;
}
inline float hash_policy(uint64_t h, uint32_t index) {  // in (0, 1], exact in f32
    uint64_t v = mix64(h ^ (0x9E3779B97F4A7C15ull * (uint64_t)(index + 1)));
    return (float)((uint32_t)(v >> 40) + 1u) * (1.0f / 16777216.0f);
}
inline float hash_eval(uint64_t h) {  // in [-1, 1), exact in f32
    uint64_t v = mix64(h ^ 0xD6E8FEB86659FD93ull);
    return (float)(uint32_t)(v >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

enum { EVAL_CALLBACK = 0, EVAL_DUMMY = 1, EVAL_HASH = 2 };

struct Evaluator {
    int kind = EVAL_DUMMY;
    EvalFn fn = nullptr;
    void* ctx = nullptr;
    int n = 5;
    int policy_size = 1575;
    // Network::policy_eval for a batch of games
    void run(const std::vector<Game>& games, std::vector<std::vector<float>>& policy, std::vector<float>& eval) {
        size_t b = games.size();
        policy.assign(b, std::vector<float>());
        eval.assign(b, 0.0f);
        if (b == 0) return;  // net5.rs:121-123
        size_t sb = state_bytes(n);
        if (kind == EVAL_DUMMY) {  // search/tests.rs:29-34
            for (size_t i = 0; i < b; i++) policy[i].assign(policy_size, 1.0f);
            return;
        }
        std::vector<uint8_t> st(b * sb);
        for (size_t i = 0; i < b; i++) pack(games[i], &st[i * sb]);
        if (kind == EVAL_HASH) {
            for (size_t i = 0; i < b; i++) {
                uint64_t h = state_hash(&st[i * sb], n);
                policy[i].resize(policy_size);
                for (int j = 0; j < policy_size; j++) policy[i][j] = hash_policy(h, (uint32_t)j);
                eval[i] = hash_eval(h);
            }
            return;
        }
        std::vector<float> pol(b * (size_t)policy_size);
        fn(ctx, (int)b, st.data(), pol.data(), eval.data());
        for (size_t i = 0; i < b; i++) policy[i].assign(pol.begin() + i * policy_size, pol.begin() + (i + 1) * policy_size);
    }
};

// ---------------------------------------------------------------------------------------
// Node, search/node.rs:3-39
// ---------------------------------------------------------------------------------------
struct Node {
    float policy = 0.0f;
    float expected_reward = 0.0f;
    uint8_t result = TG_ONGOING;
    uint32_t visits = 0;
    uint32_t virtual_visits = 0;
    std::vector<Move> moves;     // children: Box<[(Move, Node)]>
    std::vector<Node> children;

    bool is_initialized() const { return visits != 0 || virtual_visits != 0; }           // node.rs:24-26
    float visit_count() const { return (float)(visits + virtual_visits); }               // node.rs:29-31
    float expected_reward_with_losses() const {                                          // node.rs:34-39
        if (!is_initialized()) return 0.0f;
        return (expected_reward * (float)visits - (float)virtual_visits) / visit_count();
    }
};

struct SearchParams {
    float exploration_base = 500.0f;  // mcts.rs:7
    float exploration_init = 4.0f;    // mcts.rs:8
};

struct SearchError {
    bool nan = false;
    void merge(const SearchError& o) { nan = nan || o.nan; }
};

inline float exploration_rate(float n, const SearchParams& p) {                          // mcts.rs:10-12
    return logf((1.0f + n + p.exploration_base) / p.exploration_base) + p.exploration_init;
}

inline bool is_winner(uint8_t r) { return r >= TG_WHITE_ROAD && r <= TG_BLACK_FLAT; }
inline uint8_t winner_color(uint8_t r) { return (r == TG_WHITE_ROAD || r == TG_WHITE_FLAT) ? WHITE : BLACK; }

inline void update_concrete(Node& nd, float reward) {                                    // mcts.rs:120-124
    float cumulative = nd.expected_reward * (float)nd.visits;
    nd.visits += 1;
    nd.expected_reward = (cumulative + reward) / (float)nd.visits;
}

uint8_t virtual_rollout(Node& nd, Game& game, std::vector<int>& path, const SearchParams& sp, SearchError& err);

// select, mcts.rs:94-118: argmax of the upper confidence bound, LAST maximum wins (max_by)
inline uint8_t select(Node& nd, Game& game, std::vector<int>& path, const SearchParams& sp, SearchError& err) {
    float visit_count = nd.visit_count();
    float best = 0.0f;
    int best_i = -1;
    for (size_t i = 0; i < nd.children.size(); i++) {
        const Node& child = nd.children[i];
        float ucb = child.expected_reward_with_losses() +
                    exploration_rate(visit_count, sp) * child.policy * (sqrtf(visit_count) / (1.0f + child.visit_count()));
        if (ucb != ucb) { err.nan = true; }
        if (best_i < 0 || ucb >= best) { best = ucb; best_i = (int)i; }
    }
    game.play(nd.moves[best_i]);
    path.push_back(best_i);
    return virtual_rollout(nd.children[best_i], game, path, sp, err);
}

// virtual_rollout, mcts.rs:26-65
inline uint8_t virtual_rollout(Node& nd, Game& game, std::vector<int>& path, const SearchParams& sp, SearchError& err) {
    uint8_t curr_color = game.to_move;
    uint8_t result;
    if (nd.is_initialized()) {
        result = nd.result == TG_ONGOING ? select(nd, game, path, sp, err) : nd.result;
    } else {
        nd.result = game.result();
        if (nd.result == TG_ONGOING) {
            game.possible_moves(nd.moves);
            float temp_policy = 1.0f / (float)nd.moves.size();
            nd.children.assign(nd.moves.size(), Node());
            for (auto& c : nd.children) c.policy = temp_policy;
        }
        result = nd.result;
    }
    if (is_winner(result)) update_concrete(nd, winner_color(result) == curr_color ? -1.0f : 1.0f);
    else if (result == TG_DRAW || result == TG_DRAW_REVERSIBLE) update_concrete(nd, 0.0f);
    else nd.virtual_visits += 1;
    return result;
}

// devirtualize_path, mcts.rs:67-91
inline float devirtualize_path(Node& nd, const std::vector<int>& path, size_t pos, const std::vector<float>& policy,
                               float net_eval, int n) {
    nd.virtual_visits -= 1;
    float eval;
    if (pos < path.size()) {
        eval = devirtualize_path(nd.children[path[pos]], path, pos + 1, policy, net_eval, n);
    } else {
        for (size_t i = 0; i < nd.children.size(); i++) nd.children[i].policy = policy[move_index(nd.moves[i], n, (int)policy.size())];
        eval = net_eval;
    }
    eval = -eval;
    update_concrete(nd, eval);
    return eval;
}

// apply_dirichlet, noise.rs:6-16, with the samples supplied by the caller
inline void apply_noise(Node& nd, const float* noise, float ratio) {
    for (size_t i = 0; i < nd.children.size(); i++) nd.children[i].policy = noise[i] * ratio + nd.children[i].policy * (1.0f - ratio);
}
inline void dirichlet_samples(size_t k, double alpha, uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply,
                              std::vector<float>& out) {
    std::vector<double> g(k);
    double sum = 0.0;
    for (size_t i = 0; i < k; i++) { g[i] = gamma_sample(alpha, seed, slot, generation, ply, (uint32_t)i); sum += g[i]; }
    out.resize(k);
    for (size_t i = 0; i < k; i++) out[i] = sum > 0.0 ? (float)(g[i] / sum) : (float)(1.0 / (double)k);
}

// pick_move, play.rs:49-67.  exploitation → most visits, LAST on ties (max_by_key);
// otherwise sample ∝ visits with one RNG_PICK draw.
inline int pick_move(const Node& nd, bool exploitation, uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply) {
    if (exploitation) {
        int best = -1; uint32_t bv = 0;
        for (size_t i = 0; i < nd.children.size(); i++) if (best < 0 || nd.children[i].visits >= bv) { bv = nd.children[i].visits; best = (int)i; }
        return best;
    }
    uint64_t total = 0;
    for (auto& c : nd.children) total += c.visits;
    if (total == 0) return -1;  // WeightedIndex::new fails → the reference panics
    uint32_t r[4];
    rng_draw(seed, slot, generation, ply, RNG_PICK, 0, 0, r);
    uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    uint64_t target = (uint64_t)(((unsigned __int128)x * total) >> 64);
    uint64_t cum = 0;
    for (size_t i = 0; i < nd.children.size(); i++) { cum += nd.children[i].visits; if (cum > target) return (int)i; }
    return (int)nd.children.size() - 1;
}

// Node::play, play.rs:26-43 (tree reuse): the chosen child becomes the root
inline void node_play(Node& nd, int index) {
    Node child = std::move(nd.children[index]);
    nd = std::move(child);
}

// canonical depth-first serialisation of all initialised nodes (see TgNodeRecord)
inline void dump_tree(const Node& nd, TgMove move, int n, std::vector<TgNodeRecord>& out) {
    TgNodeRecord r;
    r.move = move;
    r.n_children = (uint16_t)nd.children.size();
    r.visits = nd.visits;
    r.virtual_visits = nd.virtual_visits;
    r.result = nd.result;
    std::memcpy(&r.prior_bits, &nd.policy, 4);
    std::memcpy(&r.q_bits, &nd.expected_reward, 4);
    out.push_back(r);
    for (size_t i = 0; i < nd.children.size(); i++) {
        const Node& c = nd.children[i];
        if (c.is_initialized()) dump_tree(c, encode_move(nd.moves[i], n), n, out);
        else {
            // uninitialised children carry only a prior: one leaf record, n_children = 0xFFFF
            TgNodeRecord l;
            l.move = encode_move(nd.moves[i], n);
            l.n_children = 0xFFFF;
            l.visits = 0; l.virtual_visits = 0; l.result = 0;
            std::memcpy(&l.prior_bits, &c.policy, 4);
            std::memcpy(&l.q_bits, &c.expected_reward, 4);
            out.push_back(l);
        }
    }
}

// ---------------------------------------------------------------------------------------
// lock-step search over `games` trees: the body of train/src/self_play.rs:181-210, with nodes
// indexed by game id (the reference's filter_map/zip misalignment at :183-189 is NOT reproduced).
// ---------------------------------------------------------------------------------------
struct Example {
    Game game;
    std::vector<Move> moves;
    std::vector<uint32_t> visits;
    float result = 0.0f;
    int slot = 0;
    int generation = 0;
};

struct Search {
    int n = 5;
    SearchParams sp;
    Evaluator ev;
    uint64_t seed = 0;
    std::vector<Node> nodes;
    std::vector<Game> games;
    std::vector<uint8_t> alive;
    std::vector<uint32_t> generation;
    uint64_t expansions = 0, evals = 0;
    SearchError err;

    void reset(const std::vector<Game>& roots) {
        games = roots;
        nodes.assign(roots.size(), Node());
        alive.assign(roots.size(), 1);
        generation.assign(roots.size(), 0);
    }
    int threads = 1;  // > 1: the per-game phases run under OpenMP (games are independent; results identical to 1 thread)
    int batch = 1;    // virtual rollouts per tree and iteration before the one evaluation: Player's batching (player.rs:77-93)

    // one iteration for every game whose mask byte is non-zero
    void iterate(const uint8_t* active) {
        const int G = (int)games.size(), B = batch < 1 ? 1 : batch;
        std::vector<uint8_t> res((size_t)G * B, 0xff);
        std::vector<Game> leaf((size_t)G * B);
        std::vector<std::vector<int>> path((size_t)G * B);
        std::vector<SearchError> errs(G);
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
        for (int i = 0; i < G; i++) {
            if (!alive[i] || (active && !active[i])) continue;
            for (int b = 0; b < B; b++) {  // (0..batch).filter_map(virtual_rollout), player.rs:79-90
                const size_t k = (size_t)i * B + b;
                leaf[k] = games[i];  // game.clone()
                res[k] = virtual_rollout(nodes[i], leaf[k], path[k], sp, errs[i]);
            }
        }
        std::vector<size_t> idx;
        std::vector<Game> for_eval;
        for (size_t k = 0; k < res.size(); k++) {
            if (res[k] == 0xff) continue;
            expansions++;
            if (res[k] == TG_ONGOING) { idx.push_back(k); for_eval.push_back(leaf[k]); }
        }
        for (int i = 0; i < G; i++) err.merge(errs[i]);
        std::vector<std::vector<float>> policy;
        std::vector<float> eval;
        ev.run(for_eval, policy, eval);
        evals += for_eval.size();
        // de-virtualise in rollout order; the paths of one tree are handled by one thread, in order
        std::vector<std::pair<size_t, size_t>> span(G, {0, 0});  // [first, last) into idx per game
        {
            size_t k = 0;
            for (int i = 0; i < G; i++) {
                span[i].first = k;
                while (k < idx.size() && idx[k] / (size_t)B == (size_t)i) k++;
                span[i].second = k;
            }
        }
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
        for (int i = 0; i < G; i++)
            for (size_t k = span[i].first; k < span[i].second; k++) devirtualize_path(nodes[i], path[idx[k]], 0, policy[k], eval[k], n);
    }
};

// ---------------------------------------------------------------------------------------
// self_play_parallel, train/src/self_play.rs:96-262, all constants runtime
// ---------------------------------------------------------------------------------------
struct SelfPlayParams {
    int rollouts = 400;        // ROLLOUTS
    int noise_plies = 80;      // NOISE_PLIES
    int exploit_plies = 40;    // EXPLOIT_PLIES
    float noise_alpha = 0.2f;  // NOISE_ALPHA
    float noise_ratio = 0.3f;  // NOISE_RATIO
    int komi = 2;              // Game::with_komi(2)
    int total_games = 0;       // SELF_PLAY_GAMES (0 = endless)
    uint32_t slot_base = 0;    // global index of this shard's first slot (multi-GPU sharding)
};

struct SelfPlay {
    Search s;
    SelfPlayParams p;
    std::vector<std::vector<Example>> incomplete;
    std::vector<Example> examples;
    uint64_t completed = 0, plies = 0, white_wins = 0, black_wins = 0, draws = 0, instant_wins = 0;

    void init(int n, int games) {
        s.n = n;
        std::vector<Game> roots(games, Game::start(n, p.komi * 2));
        s.reset(roots);
        incomplete.assign(games, {});
    }
    static float result_to_number(uint8_t r) {  // self_play.rs:264-275
        if (r == TG_WHITE_ROAD || r == TG_WHITE_FLAT) return 1.0f;
        if (r == TG_BLACK_ROAD || r == TG_BLACK_FLAT) return -1.0f;
        return 0.0f;
    }
    void finish_game(size_t i, uint8_t result) {
        completed += 1;
        if (is_winner(result)) { if (winner_color(result) == WHITE) white_wins++; else black_wins++; } else draws++;
        float white_result = result_to_number(result);
        s.nodes[i] = Node();
        // recycle while completed + WORKERS < SELF_PLAY_GAMES (self_play.rs:151,237)
        if (p.total_games == 0 || completed + s.games.size() < (uint64_t)p.total_games) {
            s.games[i] = Game::start(s.n, p.komi * 2);
        } else {
            s.alive[i] = 0;
        }
        for (auto& ex : incomplete[i]) {
            ex.result = ex.game.to_move == WHITE ? white_result : -white_result;
            examples.push_back(ex);
        }
        incomplete[i].clear();
        s.generation[i] += 1;
    }
    bool any_alive() const { for (auto a : s.alive) if (a) return true; return false; }

    // one pass of the outer `while` loop body (self_play.rs:108-259)
    void ply_step() {
        const int n = s.n;
        size_t G = s.games.size();
        // (a) opening, :110-116.  Far corners of the reference's 6×6 "a6"/"f6" generalised to N.
        for (size_t i = 0; i < G; i++) {
            if (!s.alive[i]) continue;
            Game& g = s.games[i];
            if (g.ply == 0) {
                Move m; m.col = 0; m.row = 0; m.piece = FLAT;
                g.play(m);
                uint32_t r[4];
                rng_draw(s.seed, p.slot_base + (uint32_t)i, s.generation[i], 0, RNG_OPENING, 0, 0, r);
                Move m2; m2.piece = FLAT; m2.row = (uint8_t)(n - 1); m2.col = (r[0] & 1) ? 0 : (uint8_t)(n - 1);
                g.play(m2);
            }
        }
        // (b) instant-win scan, :119-171
        for (size_t i = 0; i < G; i++) {
            if (!s.alive[i]) continue;
            Game& g = s.games[i];
            std::vector<Move> moves;
            g.possible_moves(moves);
            bool win = false;
            Example ex;
            ex.game = g; ex.moves = moves; ex.slot = (int)(p.slot_base + i); ex.generation = (int)s.generation[i];
            for (auto& m : moves) {
                Game c = g;
                c.play(m);
                uint8_t r = c.result();
                if (is_winner(r) && winner_color(r) == g.to_move) { win = true; ex.visits.push_back(1000); }
                else ex.visits.push_back(1);
            }
            if (win) {
                incomplete[i].push_back(ex);
                instant_wins++;
                finish_game(i, g.to_move == WHITE ? TG_WHITE_FLAT : TG_BLACK_FLAT);  // Winner{color: to_move, road: false}
            }
        }
        // (c) root evaluation + Dirichlet noise, :174-180
        {
            std::vector<uint8_t> mask(G, 0);
            bool any = false;
            for (size_t i = 0; i < G; i++) if (s.alive[i] && s.games[i].ply < p.noise_plies) { mask[i] = 1; any = true; }
            if (any) {
                s.iterate(mask.data());
                for (size_t i = 0; i < G; i++) if (mask[i]) {
                    std::vector<float> noise;
                    dirichlet_samples(s.nodes[i].children.size(), (double)p.noise_alpha, s.seed, p.slot_base + (uint32_t)i,
                                      s.generation[i], s.games[i].ply, noise);
                    apply_noise(s.nodes[i], noise.data(), p.noise_ratio);
                }
            }
        }
        // (d) rollouts, :181-210
        for (int r = 0; r < p.rollouts; r++) s.iterate(nullptr);
        // (e) pick, record, play, recycle, :212-258
        for (size_t i = 0; i < G; i++) {
            if (!s.alive[i]) continue;
            Game& g = s.games[i];
            Node& nd = s.nodes[i];
            int pick = pick_move(nd, g.ply >= p.exploit_plies, s.seed, p.slot_base + (uint32_t)i, s.generation[i], g.ply);
            Example ex;
            ex.game = g; ex.moves = nd.moves; ex.slot = (int)(p.slot_base + i); ex.generation = (int)s.generation[i];
            for (auto& c : nd.children) ex.visits.push_back(c.visits);  // improved_policy, play.rs:13-21
            incomplete[i].push_back(ex);
            Move mv = nd.moves[pick];
            node_play(nd, pick);
            g.play(mv);
            uint8_t res = g.result();
            if (res != TG_ONGOING) finish_game(i, res);
        }
        plies++;
    }
};

}  // namespace orc
