// ORACLE — TEST INFRASTRUCTURE ONLY (see tak_rules.hpp).  C entry points over the CPU
// restatement, loaded by oracle/oracle.py through ctypes.  Built by oracle/Makefile into
// oracle/_build/liboracle.so.  The product (tak_amd/, libtakgpu.so) never links this.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "tak_mcts.hpp"

using namespace orc;

extern "C" {

size_t orc_state_bytes(int n) { return state_bytes(n); }
int orc_input_channels(int n) { return input_channels(n); }
int orc_policy_size(int n, int head) { return head == TG_HEAD_FC5 ? possible_moves_count(n) : output_size(n); }
int orc_possible_moves_count(int n) { return possible_moves_count(n); }
int orc_output_size(int n) { return output_size(n); }

void orc_new_game(int n, int half_komi, uint8_t* out) { pack(Game::start(n, half_komi), out); }

// Game::from_ptn_moves (game.rs:76-82): whitespace-separated PTN; returns 0 or the TgPlayError
// of the first failing move (100 + index of a move that does not parse).
int orc_from_ptn_moves(int n, int half_komi, const char* ptn, uint8_t* out) {
    Game g = Game::start(n, half_komi);
    std::istringstream ss(ptn);
    std::string tok;
    int i = 0;
    while (ss >> tok) {
        Move m;
        if (!parse_ptn(tok, m)) return 100 + i;
        int err = g.play(m);
        if (err) return err;
        i++;
    }
    pack(g, out);
    return 0;
}

int orc_parse_move(int n, const char* ptn) {
    Move m;
    if (!parse_ptn(ptn, m)) return -1;
    if (m.col >= n || m.row >= n) return -1;
    return encode_move(m, n);
}

int orc_format_move(int n, int code, char* buf, int cap) {
    std::string s = format_ptn(decode_move((TgMove)code, n));
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int orc_to_tps(int n, const uint8_t* state, char* buf, int cap) {
    std::string s = to_tps(unpack(state, n));
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// Game::play with Game::safe_play's keep-on-error behaviour (game.rs:136-145)
void orc_play(int n, int k, uint8_t* states, const TgMove* moves, uint8_t* status) {
    size_t sb = state_bytes(n);
#pragma omp parallel for schedule(static) if (k >= 4096)
    for (int i = 0; i < k; i++) {
        Game g = unpack(states + i * sb, n);
        Move m = decode_move(moves[i], n);
        int err;
        if ((moves[i] & 63) >= n * n) err = TG_PLAY_OUT_OF_BOUNDS;
        else if (!m.spread && m.piece > CAP) err = TG_PLAY_OUT_OF_BOUNDS;
        else err = g.play(m);
        status[i] = (uint8_t)err;
        if (!err) pack(g, states + i * sb);
    }
}

void orc_movegen(int n, int k, const uint8_t* states, TgMove* moves, int32_t* counts) {
    size_t sb = state_bytes(n);
#pragma omp parallel for schedule(static) if (k >= 4096)
    for (int i = 0; i < k; i++) {
        std::vector<Move> mv;
        Game g = unpack(states + i * sb, n);
        g.possible_moves(mv);
        counts[i] = (int32_t)mv.size();
        for (size_t j = 0; j < mv.size() && j < TG_MAX_MOVES; j++) moves[(size_t)i * TG_MAX_MOVES + j] = encode_move(mv[j], n);
    }
}

void orc_result(int n, int k, const uint8_t* states, uint8_t* results) {
    size_t sb = state_bytes(n);
#pragma omp parallel for schedule(static) if (k >= 4096)
    for (int i = 0; i < k; i++) results[i] = unpack(states + i * sb, n).result();
}

void orc_encode(int n, int k, const uint8_t* states, float* planes) {
    size_t sb = state_bytes(n);
    size_t per = (size_t)input_channels(n) * n * n;
#pragma omp parallel for schedule(static) if (k >= 4096)
    for (int i = 0; i < k; i++) game_repr(unpack(states + i * sb, n), planes + i * per);
}

void orc_move_index(int n, int k, const TgMove* moves, int32_t* index) {
    for (int i = 0; i < k; i++) index[i] = move_index(decode_move(moves[i], n), n);
}

uint64_t orc_perft(int n, const uint8_t* state, int depth) { return perft(unpack(state, n), depth); }

// newline-joined PTN of the regenerated legacy 5×5 move table (move_map.rs:51-201)
int orc_legacy5_table(char* buf, int cap) {
    std::string s;
    const auto& t = legacy5_table();
    for (size_t i = 0; i < t.size(); i++) { if (i) s += '\n'; s += format_ptn(t[i]); }
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)t.size();
}

// Positions sampled from pseudo-random play (uniform over legal moves), one position per
// output slot, at a pseudo-random ply < max_plies of its own game (restarting when a game
// ends).  Used to build test / bench inputs.
void orc_random_positions(int n, int count, uint64_t seed, int max_plies, int half_komi, uint8_t* out) {
    size_t sb = state_bytes(n);
#pragma omp parallel for schedule(dynamic, 256) if (count >= 4096)
    for (int i = 0; i < count; i++) {
        std::vector<Move> mv;
        uint64_t s = mix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(i + 1)));
        int target = (int)(mix64(s) % (uint64_t)(max_plies + 1));
        Game g = Game::start(n, half_komi);
        for (int p = 0; p < target; p++) {
            if (g.result() != TG_ONGOING) break;
            g.possible_moves(mv);
            s = mix64(s + 0x632BE59BD9B4E019ull);
            Game c = g;
            c.play(mv[s % mv.size()]);
            if (c.result() != TG_ONGOING && (s >> 60) != 0) break;  // mostly keep positions non-terminal
            g = c;
        }
        pack(g, out + i * sb);
    }
}

// Whole pseudo-random games steered towards particular endings (test-input generator): every game is played
// with the rules above until it ends or max_plies is reached; `style` weights the move classes so that the
// rarer branches of Game::result (game.rs:220-267) are reached by PLAY — board-full and reserves-exhausted
// flat counts with every komi parity, the 50-reversible-plies draw, tall stacks:
//   0 uniform over the legal moves              1 flat placements whenever one exists (board fills up)
//   2 mostly walls, few spreads (board full, few flats)   3 stacking game: spreads 60 %, placements mostly walls
//   4 as 3 for a pseudo-random number of plies, then spreads only (reversible_plies runs up to 50)
// avoid_roads: a candidate that completes a road is re-drawn (up to 8 times), so games last until another ending.
// half_komi = 127: a pseudo-random half-komi in [-5, 6] per game.  Outputs per game: the final state, the
// state before the last move, that move, the final result, and the highest stack seen at the end.
void orc_playouts(int n, int count, uint64_t seed, int half_komi, int style, int avoid_roads, int max_plies, uint8_t* out_final,
                  uint8_t* out_prev, TgMove* out_move, uint8_t* out_result, uint8_t* out_max_height) {
    size_t sb = state_bytes(n);
#pragma omp parallel for schedule(dynamic, 64) if (count >= 256)
    for (int i = 0; i < count; i++) {
        std::vector<Move> mv, pool;
        uint64_t s = mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(i + 1)));
        auto next = [&]() { s = mix64(s + 0x632BE59BD9B4E019ull); return s; };
        int hk = half_komi == 127 ? (int)(next() % 12) - 5 : half_komi;
        Game g = Game::start(n, hk), prev = g;
        Move last;
        const int switch_ply = 8 + (int)(next() % 40);
        uint8_t res = TG_ONGOING;
        for (int p = 0; p < max_plies; p++) {
            res = g.result();
            if (res != TG_ONGOING) break;
            g.possible_moves(mv);
            Game c;
            Move pick;
            for (int attempt = 0; attempt < 8; attempt++) {
                // weight by move class: collect the preferred class, fall back to all moves
                int want = -1;  // 0 flat, 1 wall, 2 cap, 3 spread, -1 any
                uint64_t r = next() % 100;
                if (style == 1) want = 0;
                else if (style == 2) want = r < 65 ? 1 : r < 95 ? 0 : 3;
                else if (style == 3 || (style == 4 && p < switch_ply)) want = r < 60 ? 3 : r < 85 ? 1 : r < 97 ? 0 : 2;
                else if (style == 4) want = 3;
                pool.clear();
                if (want >= 0)
                    for (const Move& m : mv)
                        if ((m.spread ? 3 : (int)m.piece) == want) pool.push_back(m);
                const std::vector<Move>& from = pool.empty() ? mv : pool;
                pick = from[next() % from.size()];
                c = g;
                c.play(pick);
                if (!avoid_roads) break;
                uint8_t r2 = c.result();
                if (r2 != TG_WHITE_ROAD && r2 != TG_BLACK_ROAD) break;
            }
            prev = g;
            last = pick;
            g = c;
        }
        res = g.result();
        pack(g, out_final + (size_t)i * sb);
        pack(prev, out_prev + (size_t)i * sb);
        out_move[i] = encode_move(last, n);
        out_result[i] = res;
        int mh = 0;
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) mh = g.board[y][x].len > mh ? g.board[y][x].len : mh;
        out_max_height[i] = (uint8_t)mh;
    }
}

// seeded whole games as in tak/tests/symm.rs:3-27 / tps.rs:26-55: always play moves[seed % count]
// until the game ends; returns the number of plies, final state in out, result in *result.
int orc_seeded_game(int n, uint64_t seed, uint8_t* out, uint8_t* result) {
    Game g = Game::start(n, 0);
    std::vector<Move> mv;
    while (g.result() == TG_ONGOING) {
        g.possible_moves(mv);
        int err = g.play(mv[seed % mv.size()]);
        if (err) return -err;
    }
    pack(g, out);
    *result = g.result();
    return g.ply;
}

// ------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------
struct SearchHandle { Search s; };

void* orc_search_new(int n, int eval_kind, EvalFn fn, void* ctx, int policy_size, float base, float init, uint64_t seed) {
    SearchHandle* h = new SearchHandle();
    h->s.n = n;
    h->s.ev.kind = eval_kind; h->s.ev.fn = fn; h->s.ev.ctx = ctx; h->s.ev.n = n; h->s.ev.policy_size = policy_size;
    h->s.sp.exploration_base = base; h->s.sp.exploration_init = init;
    h->s.seed = seed;
    return h;
}
void orc_search_free(void* h) { delete (SearchHandle*)h; }

void orc_search_reset(void* hh, int games, const uint8_t* states) {
    Search& s = ((SearchHandle*)hh)->s;
    size_t sb = state_bytes(s.n);
    std::vector<Game> roots;
    for (int i = 0; i < games; i++) roots.push_back(unpack(states + i * sb, s.n));
    s.reset(roots);
}

int orc_search_run(void* hh, int iters, const uint8_t* active) {
    Search& s = ((SearchHandle*)hh)->s;
    for (int i = 0; i < iters; i++) s.iterate(active);
    return s.err.nan ? TG_ERR_NAN : 0;
}

void orc_search_apply_noise(void* hh, const float* noise, float ratio, const uint8_t* active) {
    Search& s = ((SearchHandle*)hh)->s;
    for (size_t g = 0; g < s.nodes.size(); g++) if (!active || active[g]) apply_noise(s.nodes[g], noise + g * TG_MAX_MOVES, ratio);
}

void orc_search_apply_dirichlet(void* hh, float alpha, float ratio, const uint8_t* active) {
    Search& s = ((SearchHandle*)hh)->s;
    for (size_t g = 0; g < s.nodes.size(); g++) if (!active || active[g]) {
        std::vector<float> noise;
        dirichlet_samples(s.nodes[g].children.size(), (double)alpha, s.seed, (uint32_t)g, s.generation[g], s.games[g].ply, noise);
        apply_noise(s.nodes[g], noise.data(), ratio);
    }
}

void orc_search_root(void* hh, TgMove* moves, uint32_t* visits, float* prior, float* q, int32_t* counts,
                     uint32_t* root_visits, float* root_q) {
    Search& s = ((SearchHandle*)hh)->s;
    for (size_t g = 0; g < s.nodes.size(); g++) {
        const Node& nd = s.nodes[g];
        if (counts) counts[g] = (int32_t)nd.children.size();
        if (root_visits) root_visits[g] = nd.visits;
        if (root_q) root_q[g] = nd.expected_reward;
        for (size_t i = 0; i < nd.children.size() && i < TG_MAX_MOVES; i++) {
            size_t o = g * TG_MAX_MOVES + i;
            if (moves) moves[o] = encode_move(nd.moves[i], s.n);
            if (visits) visits[o] = nd.children[i].visits;
            if (prior) prior[o] = nd.children[i].policy;
            if (q) q[o] = nd.children[i].expected_reward;
        }
    }
}

int orc_search_play(void* hh, const TgMove* moves, const uint8_t* active) {
    Search& s = ((SearchHandle*)hh)->s;
    for (size_t g = 0; g < s.nodes.size(); g++) {
        if (active && !active[g]) continue;
        Node& nd = s.nodes[g];
        Move m = decode_move(moves[g], s.n);
        int idx = -1;
        for (size_t i = 0; i < nd.moves.size(); i++) if (nd.moves[i] == m) { idx = (int)i; break; }
        if (idx < 0) return TG_ERR_ILLEGAL_MOVE;  // "tried to play an invalid move", play.rs:35
        node_play(nd, idx);
        s.games[g].play(m);
    }
    return 0;
}

void orc_search_states(void* hh, uint8_t* states) {
    Search& s = ((SearchHandle*)hh)->s;
    size_t sb = state_bytes(s.n);
    for (size_t g = 0; g < s.games.size(); g++) pack(s.games[g], states + g * sb);
}

int orc_search_dump(void* hh, int game, TgNodeRecord* records, size_t capacity, size_t* n_records) {
    Search& s = ((SearchHandle*)hh)->s;
    std::vector<TgNodeRecord> out;
    dump_tree(s.nodes[game], 0, s.n, out);
    *n_records = out.size();
    if (out.size() > capacity) return TG_ERR_INVALID_ARG;
    std::memcpy(records, out.data(), out.size() * sizeof(TgNodeRecord));
    return 0;
}

void orc_search_counters(void* hh, uint64_t* expansions, uint64_t* evals) {
    Search& s = ((SearchHandle*)hh)->s;
    *expansions = s.expansions; *evals = s.evals;
}

// ------------------------------------------------------------------------------------------
// self-play
// ------------------------------------------------------------------------------------------
struct SelfPlayHandle { SelfPlay sp; };

void* orc_selfplay_new(int n, int games, int eval_kind, EvalFn fn, void* ctx, int policy_size, float base, float init,
                       uint64_t seed, const TgSelfPlayConfig* cfg, uint32_t slot_base) {
    SelfPlayHandle* h = new SelfPlayHandle();
    SelfPlay& sp = h->sp;
    sp.p.rollouts = cfg->rollouts; sp.p.noise_plies = cfg->noise_plies; sp.p.exploit_plies = cfg->exploit_plies;
    sp.p.noise_alpha = cfg->noise_alpha; sp.p.noise_ratio = cfg->noise_ratio; sp.p.komi = cfg->komi;
    sp.p.total_games = cfg->total_games; sp.p.slot_base = slot_base;
    sp.s.ev.kind = eval_kind; sp.s.ev.fn = fn; sp.s.ev.ctx = ctx; sp.s.ev.n = n; sp.s.ev.policy_size = policy_size;
    sp.s.sp.exploration_base = base; sp.s.sp.exploration_init = init;
    sp.s.seed = seed;
    sp.init(n, games);
    return h;
}
void orc_selfplay_free(void* h) { delete (SelfPlayHandle*)h; }
// OpenMP threads for the per-game phases of the lock-step loop (1 = serial; results do not depend on it)
void orc_selfplay_threads(void* hh, int threads) { ((SelfPlayHandle*)hh)->sp.s.threads = threads < 1 ? 1 : threads; }
void orc_search_batch(void* hh, int batch) { ((SearchHandle*)hh)->s.batch = batch < 1 ? 1 : batch; }
void orc_search_threads(void* hh, int threads) { ((SearchHandle*)hh)->s.threads = threads < 1 ? 1 : threads; }

int orc_selfplay_step(void* hh, int plies) {
    SelfPlay& sp = ((SelfPlayHandle*)hh)->sp;
    for (int i = 0; i < plies && sp.any_alive(); i++) sp.ply_step();
    return sp.s.err.nan ? TG_ERR_NAN : 0;
}

void orc_selfplay_stats(void* hh, TgSelfPlayStats* out) {
    SelfPlay& sp = ((SelfPlayHandle*)hh)->sp;
    out->games_finished = sp.completed; out->examples = sp.examples.size();
    out->expansions = sp.s.expansions; out->evals = sp.s.evals; out->plies = sp.plies;
    out->white_wins = sp.white_wins; out->black_wins = sp.black_wins; out->draws = sp.draws;
    out->instant_wins = sp.instant_wins;
    out->dropped_examples = 0;
    out->aborted_games = 0;  // the reference's heap structures have no capacities to exceed
    out->alive_games = 0;
    for (auto a : sp.s.alive) out->alive_games += a ? 1u : 0u;
}

// examples in emission order; removes what it returns.  header.game_id = slot | generation << 20
int orc_selfplay_drain(void* hh, int cap, TgExampleHeader* headers, uint8_t* states, TgMove* moves, uint32_t* visits, int32_t* n_out) {
    SelfPlay& sp = ((SelfPlayHandle*)hh)->sp;
    size_t sb = state_bytes(sp.s.n);
    int k = 0;
    for (; k < cap && k < (int)sp.examples.size(); k++) {
        const Example& ex = sp.examples[k];
        headers[k].game_id = ex.slot | (ex.generation << 20);
        headers[k].n_moves = (int32_t)ex.moves.size();
        headers[k].result = ex.result;
        headers[k].reserved = 0;
        pack(ex.game, states + k * sb);
        for (size_t j = 0; j < ex.moves.size() && j < TG_MAX_MOVES; j++) {
            moves[(size_t)k * TG_MAX_MOVES + j] = encode_move(ex.moves[j], sp.s.n);
            visits[(size_t)k * TG_MAX_MOVES + j] = ex.visits[j];
        }
    }
    sp.examples.erase(sp.examples.begin(), sp.examples.begin() + k);
    *n_out = k;
    return 0;
}

void orc_selfplay_states(void* hh, uint8_t* states, uint8_t* alive) {
    SelfPlay& sp = ((SelfPlayHandle*)hh)->sp;
    size_t sb = state_bytes(sp.s.n);
    for (size_t g = 0; g < sp.s.games.size(); g++) { pack(sp.s.games[g], states + g * sb); alive[g] = sp.s.alive[g]; }
}

// Example::to_tensors: k examples → 8k symmetric states + 8k policy targets (policy_size each)
void orc_augment(int n, int k, int policy_size, const uint8_t* states, const int32_t* n_moves, const TgMove* moves,
                 const uint32_t* visits, uint8_t* out_states, float* pi) {
    size_t sb = state_bytes(n);
    for (int e = 0; e < k; e++) {
        Game g = unpack(states + e * sb, n);
        std::vector<Move> mv;
        std::vector<uint32_t> vs;
        for (int j = 0; j < n_moves[e]; j++) {
            mv.push_back(decode_move(moves[(size_t)e * TG_MAX_MOVES + j], n));
            vs.push_back(visits[(size_t)e * TG_MAX_MOVES + j]);
        }
        Game og[8];
        std::vector<float> opi[8];
        example_symmetries(g, mv, vs, policy_size, og, opi);
        for (int i = 0; i < 8; i++) {
            pack(og[i], out_states + ((size_t)e * 8 + i) * sb);
            std::memcpy(pi + ((size_t)e * 8 + i) * policy_size, opi[i].data(), (size_t)policy_size * 4);
        }
    }
}

// rng / math spec probes (so tests can compare the product's device implementation)
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) { Philox::gen(seed, c0, c1, c2, c3, out); }
void orc_dirichlet(int k, double alpha, uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply, float* out) {
    std::vector<float> v;
    dirichlet_samples((size_t)k, alpha, seed, slot, generation, ply, v);
    std::memcpy(out, v.data(), (size_t)k * 4);
}
uint64_t orc_state_hash(int n, const uint8_t* state) { return state_hash(state, n); }
float orc_hash_policy(uint64_t h, uint32_t i) { return hash_policy(h, i); }
float orc_hash_eval(uint64_t h) { return hash_eval(h); }

}  // extern "C"
