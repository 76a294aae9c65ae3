// pit.hip — batched two-network evaluation: `pit` of reference train/src/pit.rs:15-96 (and the gate of
// train/src/main.rs:98-106) on top of the C ABI.  Host code only: it drives two engine handles (one per
// weight set) through tg_search_*; every game of the match is played concurrently — 2·pairs lock-step games,
// each with one tree per network — instead of the reference's one game at a time through `Player`.  `Player`'s batching
// (`batch` virtual rollouts per tree, then one evaluation) is TgSearchConfig.batch.
#include <cstring>
#include <vector>

#include "engine.h"
#include "rng.cuh"

using namespace tg;

namespace {

enum : uint32_t { RNG_PIT_CORNER = 8, RNG_PIT_RANDOM = 9 };

void start_state(int n, int komi, uint8_t* st, size_t bytes) {  // Game::with_komi (tak/src/game.rs:58-63)
    std::memset(st, 0, bytes);
    TgHeader* h = (TgHeader*)(st + bytes - sizeof(TgHeader));
    int stones, caps;
    starting_stones(n, stones, caps);
    h->n = (uint8_t)n;
    h->white_stones = h->black_stones = (uint8_t)stones;
    h->white_caps = h->black_caps = (uint8_t)caps;
    h->half_komi = (int8_t)(2 * komi);
}

}  // namespace

extern "C" int tg_pit(TgEngine* e_new, TgEngine* e_old, const TgPitConfig* cfg, TgPitResult* out) {
    if (!e_new || !e_old || !cfg || !out) return fail(TG_ERR_INVALID_ARG, "tg_pit: null argument");
    if (e_new == e_old) return fail(TG_ERR_INVALID_ARG, "tg_pit: the two networks need two engine handles");
    if (e_new->cfg.board_size != e_old->cfg.board_size || e_new->cfg.policy_head != e_old->cfg.policy_head)
        return fail(TG_ERR_INVALID_ARG, "tg_pit: engines differ in board size / policy head");
    if (cfg->pairs <= 0 || cfg->rollouts <= 0 || cfg->random_plies < 0) return fail(TG_ERR_INVALID_ARG, "tg_pit: bad configuration");
    const int n = e_new->cfg.board_size, G = 2 * cfg->pairs;
    const int batch = cfg->batch > 0 ? cfg->batch : 1;
    if ((long long)G * batch > e_new->cfg.max_batch || (long long)G * batch > e_old->cfg.max_batch)
        return fail(TG_ERR_INVALID_ARG, "tg_pit: 2·pairs·batch exceeds max_batch");
    const size_t sb = tg_state_bytes(n);
    const int idle_rollouts = cfg->idle_rollouts > 0 ? cfg->idle_rollouts : 1;  // the waiting tree needs an expanded root (play.rs:35)
    std::memset(out, 0, sizeof(*out));

    // ---- openings (pit.rs:33-63): a1, a random far corner, then random_plies random Flat/Cap placements ----
    std::vector<uint8_t> op((size_t)cfg->pairs * sb);
    std::vector<TgMove> mv(cfg->pairs), list((size_t)cfg->pairs * TG_MAX_MOVES);
    std::vector<int32_t> counts(cfg->pairs);
    std::vector<uint8_t> status(cfg->pairs);
    for (int p = 0; p < cfg->pairs; p++) start_state(n, cfg->komi, op.data() + (size_t)p * sb, sb);
    int rc;
    for (int ply = 0; ply < 2 + cfg->random_plies; ply++) {
        if (ply == 0) for (int p = 0; p < cfg->pairs; p++) mv[p] = 0;  // "a1"
        else if (ply == 1)
            for (int p = 0; p < cfg->pairs; p++) {
                U4 r = rng_draw(cfg->seed, (uint32_t)p, 0, 1, RNG_PIT_CORNER, 0, 0);
                mv[p] = (TgMove)((r.v[0] & 1u) ? n * n - 1 : (n - 1) * n);  // the reference's "a6" / "f6", generalised
            }
        else {
            rc = tg_movegen(e_new, cfg->pairs, op.data(), list.data(), counts.data());
            if (rc) return rc;
            for (int p = 0; p < cfg->pairs; p++) {
                std::vector<TgMove> ok;
                for (int i = 0; i < counts[p]; i++) {
                    TgMove m = list[(size_t)p * TG_MAX_MOVES + i];
                    if ((m >> 8) == 0 && ((m >> 6) & 3) != 1) ok.push_back(m);  // MoveKind::Place(Flat | Cap), pit.rs:52
                }
                if (ok.empty()) return fail(TG_ERR_STATE, "tg_pit: no placement available in the opening");
                U4 r = rng_draw(cfg->seed, (uint32_t)p, 0, (uint32_t)ply, RNG_PIT_RANDOM, 0, 0);
                uint64_t x = ((uint64_t)r.v[0] << 32) | r.v[1];
                mv[p] = ok[(size_t)(((unsigned __int128)x * ok.size()) >> 64)];
            }
        }
        rc = tg_play(e_new, cfg->pairs, op.data(), mv.data(), status.data());
        if (rc) return rc;
    }
    // game g = 2·pair + colour: the new network plays White in the even games, Black in the odd ones (pit.rs:27)
    std::vector<uint8_t> states((size_t)G * sb);
    for (int g = 0; g < G; g++) std::memcpy(states.data() + (size_t)g * sb, op.data() + (size_t)(g / 2) * sb, sb);

    TgSearchConfig sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.games = G;
    sc.arena_nodes = cfg->arena_nodes > 0 ? cfg->arena_nodes : 1 << 14;
    sc.exploration_base = 500.0f;
    sc.exploration_init = 4.0f;
    sc.seed = cfg->seed;
    sc.batch = (uint32_t)batch;
    TgEngine* eng[2] = {e_new, e_old};
    for (TgEngine* e : eng) {
        rc = tg_search_create(e, &sc);
        if (rc) return rc;
        rc = tg_search_reset(e, states.data());
        if (rc) return rc;
    }
    std::vector<uint8_t> results(G), alive(G, 1), final_res(G, TG_ONGOING), act[2], idle[2];
    for (int k = 0; k < 2; k++) { act[k].resize(G); idle[k].resize(G); }
    std::vector<TgMove> moves[2], chosen(G);
    std::vector<uint32_t> visits[2];
    std::vector<int32_t> cnt[2];
    for (int k = 0; k < 2; k++) { moves[k].resize((size_t)G * TG_MAX_MOVES); visits[k].resize((size_t)G * TG_MAX_MOVES); cnt[k].resize(G); }
    for (int ply = 0;; ply++) {
        rc = tg_result(e_new, G, states.data(), results.data());
        if (rc) return rc;
        int live = 0;
        for (int g = 0; g < G; g++) {
            if (alive[g] && results[g] != TG_ONGOING) {  // PitResult::update, pit.rs:113-126
                alive[g] = 0;
                final_res[g] = results[g];
                const bool new_is_white = (g & 1) == 0;
                if (results[g] == TG_DRAW || results[g] == TG_DRAW_REVERSIBLE) out->draws++;
                else {
                    const bool white_won = results[g] == TG_WHITE_ROAD || results[g] == TG_WHITE_FLAT;
                    if (white_won == new_is_white) out->wins++; else out->losses++;
                }
            }
            live += alive[g];
        }
        if (!live) break;
        if (cfg->max_plies > 0 && ply >= cfg->max_plies) { out->unfinished = (uint32_t)live; break; }
        for (int g = 0; g < G; g++) {
            const TgHeader* h = (const TgHeader*)(states.data() + (size_t)g * sb + sb - sizeof(TgHeader));
            const bool new_to_move = alive[g] && ((h->to_move == 0) == ((g & 1) == 0));
            act[0][g] = new_to_move; idle[0][g] = alive[g] && !new_to_move;
            act[1][g] = idle[0][g];  idle[1][g] = new_to_move;
        }
        // the side to move runs `rollouts` batches of `batch` virtual rollouts (pit.rs:78-80: ROLLOUTS × Player::rollout); the
        // waiting side gets the batch `Player` keeps in flight (player.rs:65-66,140), which also expands its root
        for (int k = 0; k < 2; k++) {
            rc = tg_search_run(eng[k], cfg->rollouts, act[k].data());
            if (rc) return rc;
            rc = tg_search_run(eng[k], idle_rollouts, idle[k].data());
            if (rc) return rc;
            rc = tg_search_root(eng[k], moves[k].data(), visits[k].data(), nullptr, nullptr, cnt[k].data(), nullptr, nullptr);
            if (rc) return rc;
        }
        for (int g = 0; g < G; g++) {  // pick_move(exploit = true): most visited, the LAST one on ties (play.rs:54-57)
            chosen[g] = 0;
            if (!alive[g]) continue;
            const int k = act[0][g] ? 0 : 1;
            const uint32_t* v = visits[k].data() + (size_t)g * TG_MAX_MOVES;
            int best = -1;
            for (int i = 0; i < cnt[k][g]; i++) if (best < 0 || v[i] >= v[best]) best = i;
            if (best < 0) return fail(TG_ERR_STATE, "tg_pit: root without children");
            chosen[g] = moves[k][(size_t)g * TG_MAX_MOVES + best];
        }
        for (int k = 0; k < 2; k++) {
            rc = tg_search_play(eng[k], chosen.data(), alive.data());
            if (rc) return rc;
        }
        rc = tg_search_states(e_new, states.data());
        if (rc) return rc;
        out->plies++;
    }
    out->win_rate = (out->wins + out->losses) ? (double)out->wins / (double)(out->wins + out->losses) : 0.0;  // pit.rs:105-110
    // the reference plays the openings one after the other and stops once the verdict cannot change (pit.rs:20-23)
    const uint32_t P = (uint32_t)cfg->pairs;
    for (uint32_t i = 0; i < P; i++) {
        if (out->ref_wins > P + P / 10 || out->ref_losses > P - P / 10) break;
        out->ref_pairs++;
        for (int c = 0; c < 2; c++) {  // Color::White, then Color::Black (pit.rs:27)
            const uint8_t r = final_res[2 * i + c];
            if (r == TG_ONGOING) continue;  // cut by max_plies
            if (r == TG_DRAW || r == TG_DRAW_REVERSIBLE) out->ref_draws++;
            else if ((r == TG_WHITE_ROAD || r == TG_WHITE_FLAT) == (c == 0)) out->ref_wins++;
            else out->ref_losses++;
        }
    }
    out->ref_win_rate = (out->ref_wins + out->ref_losses) ? (double)out->ref_wins / (double)(out->ref_wins + out->ref_losses) : 0.0;
    return TG_OK;
}
