// search.hip — placeholder, replaced by the real MCTS kernels
#include "engine.h"
namespace tg { struct Search {}; void search_destroy(Search* s) { delete s; } }
using namespace tg;
#define NOTYET return fail(TG_ERR_STATE, "search not built yet")
extern "C" {
int tg_search_create(TgEngine*, const TgSearchConfig*) { NOTYET; }
int tg_search_reset(TgEngine*, const void*) { NOTYET; }
int tg_search_run(TgEngine*, int, const uint8_t*) { NOTYET; }
int tg_search_apply_dirichlet(TgEngine*, float, float, const uint8_t*) { NOTYET; }
int tg_search_apply_noise(TgEngine*, const float*, float, const uint8_t*) { NOTYET; }
int tg_search_root(TgEngine*, TgMove*, uint32_t*, float*, float*, int32_t*, uint32_t*, float*) { NOTYET; }
int tg_search_play(TgEngine*, const TgMove*, const uint8_t*) { NOTYET; }
int tg_search_states(TgEngine*, void*) { NOTYET; }
int tg_search_dump(TgEngine*, int, TgNodeRecord*, size_t, size_t*) { NOTYET; }
int tg_search_counters(TgEngine*, uint64_t*, uint64_t*) { NOTYET; }
int tg_selfplay_create(TgEngine*, const TgSearchConfig*, const TgSelfPlayConfig*) { NOTYET; }
int tg_selfplay_step(TgEngine*, int) { NOTYET; }
int tg_selfplay_stats(TgEngine*, TgSelfPlayStats*) { NOTYET; }
int tg_selfplay_drain(TgEngine*, int, TgExampleHeader*, void*, TgMove*, uint32_t*, int32_t*) { NOTYET; }
}
