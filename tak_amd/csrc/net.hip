// net.hip — placeholder, replaced by the real network kernels
#include "engine.h"
namespace tg {
struct Net {};
int net_create(TgEngine*) { return TG_OK; }
void net_destroy(Net* n) { delete n; }
int net_set_tensor(TgEngine*, const char*, const float*, size_t) { return fail(TG_ERR_STATE, "network not built yet"); }
int net_finalize(TgEngine*) { return fail(TG_ERR_STATE, "network not built yet"); }
bool net_ready(const TgEngine*) { return false; }
int net_forward_dev(TgEngine*, int, const float*, float*, float*) { return fail(TG_ERR_STATE, "network not built yet"); }
}
using namespace tg;
extern "C" {
int tg_net_set_tensor(TgEngine* e, const char* name, const float* data, size_t count) { return net_set_tensor(e, name, data, count); }
int tg_net_finalize(TgEngine* e) { return net_finalize(e); }
int tg_policy_eval(TgEngine*, int, const void*, float*, float*) { return fail(TG_ERR_STATE, "network not built yet"); }
int tg_forward_mcts(TgEngine*, int, const float*, float*, float*) { return fail(TG_ERR_STATE, "network not built yet"); }
int tg_policy_eval_dev(TgEngine*, int, const void*, float*, float*) { return fail(TG_ERR_STATE, "network not built yet"); }
}
