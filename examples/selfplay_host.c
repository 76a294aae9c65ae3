/* A host written against include/takgpu.h alone — plain C99, no Python, no torch — doing what the reference's
 * `train` binary does around self_play_parallel (train/src/main.rs:60-84): build a randomly initialised network
 * (Network::default), check the rules engine on the perft known answers of tak/tests/perft.rs, play batched
 * self-play games, drain the examples and print the first one in the reference's text format (example.rs:81-99).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/selfplay_host.c -Ltak_amd -ltakgpu -Wl,-rpath,$PWD/tak_amd -o selfplay_host
 *   ./selfplay_host [games] [rollouts] [plies]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "takgpu.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ < 0) {                                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, tg_last_error());      \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv) {
    const int games = argc > 1 ? atoi(argv[1]) : 256;
    const int rollouts = argc > 2 ? atoi(argv[2]) : 32;
    const int plies = argc > 3 ? atoi(argv[3]) : 12;

    TgConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = TG_ABI_VERSION;
    cfg.device = 0;
    cfg.board_size = 5;
    cfg.res_blocks = 6;
    cfg.filters = 64;
    cfg.policy_head = TG_HEAD_FC5;
    cfg.evaluator = TG_EVAL_RESNET;
    cfg.max_batch = games;
    TgEngine* e = NULL;
    CHECK(tg_engine_create(&cfg, &e));
    CHECK(tg_net_init_random(e, 0));
    CHECK(tg_net_finalize(e));

    /* Game::with_komi(2) on an empty 5x5 board (game.rs:10-20,58-63) */
    TgState5 start;
    memset(&start, 0, sizeof start);
    start.h.n = 5;
    start.h.white_stones = start.h.black_stones = 21;
    start.h.white_caps = start.h.black_caps = 1;
    start.h.half_komi = 4;

    uint64_t perft = 0;
    CHECK(tg_perft(e, 1, &start, 3, &perft)); /* tak/tests/perft.rs: 5x5 depth 3 = 43 320 */
    printf("perft(5x5, depth 3) = %llu\n", (unsigned long long)perft);
    if (perft != 43320ull) {
        fprintf(stderr, "perft mismatch\n");
        return 1;
    }

    float* policy = (float*)malloc(sizeof(float) * (size_t)tg_policy_size(5, TG_HEAD_FC5));
    float eval = 0.0f, sum = 0.0f;
    CHECK(tg_policy_eval(e, 1, &start, policy, &eval)); /* Network::policy_eval */
    for (int i = 0; i < tg_policy_size(5, TG_HEAD_FC5); i++) sum += policy[i];
    printf("policy_eval(start): sum(policy) = %.6f, eval = %+.6f\n", sum, eval);
    if (sum < 0.999f || sum > 1.001f || eval < -1.0f || eval > 1.0f) {
        fprintf(stderr, "network output out of range\n");
        return 1;
    }

    TgSearchConfig sc;
    memset(&sc, 0, sizeof sc);
    sc.games = games;
    sc.arena_nodes = 1 << 15;
    sc.exploration_base = 500.0f;
    sc.exploration_init = 4.0f;
    sc.seed = 1;
    TgSelfPlayConfig sp;
    memset(&sp, 0, sizeof sp);
    sp.rollouts = rollouts;
    sp.noise_plies = 80;
    sp.exploit_plies = 40;
    sp.noise_alpha = 0.2f;
    sp.noise_ratio = 0.3f;
    sp.komi = 2;
    sp.total_games = 0;
    sp.max_examples = games * (plies + 2);
    CHECK(tg_selfplay_create(e, &sc, &sp));
    CHECK(tg_selfplay_step(e, plies));
    CHECK(tg_sync(e));
    TgSelfPlayStats st;
    CHECK(tg_selfplay_stats(e, &st));
    printf("self-play: %d games x %d plies, %llu expansions, %llu network evals, %llu games finished, %llu examples\n", games, plies,
           (unsigned long long)st.expansions, (unsigned long long)st.evals, (unsigned long long)st.games_finished,
           (unsigned long long)st.examples);
    if (st.plies != (uint64_t)plies || st.expansions == 0) {
        fprintf(stderr, "self-play did not advance\n");
        return 1;
    }

    const int cap = 64;
    TgExampleHeader* hdr = (TgExampleHeader*)malloc(sizeof(TgExampleHeader) * cap);
    TgState5* states = (TgState5*)malloc(sizeof(TgState5) * cap);
    TgMove* moves = (TgMove*)malloc(sizeof(TgMove) * cap * TG_MAX_MOVES);
    uint32_t* visits = (uint32_t*)malloc(sizeof(uint32_t) * cap * TG_MAX_MOVES);
    int32_t n_out = 0;
    CHECK(tg_selfplay_drain(e, cap, hdr, states, moves, visits, &n_out));
    printf("drained %d examples of finished games\n", n_out);
    if (n_out > 0) {
        char line[16384];
        CHECK(tg_format_example(5, &states[0], hdr[0].n_moves, moves, visits, hdr[0].result, line, sizeof line));
        printf("example[0] (game %d): %.120s...\n", hdr[0].game_id, line);
    }
    free(hdr); free(states); free(moves); free(visits); free(policy);
    tg_engine_destroy(e);
    printf("OK\n");
    return 0;
}
