// train.hip — host side of the training step: Network::train / train_inner (reference
// alpha-tak/src/model/network.rs:37-97) and forward_training (net5.rs:113-118, net6.rs:111-122) without tch.
//
// Master parameters, gradients and Adam moments are flat f32 device buffers in tch layout (conv OIHW,
// linear [out,in], BN vectors) in the creation order of net5.rs:29-62, so Adam is one elementwise kernel and
// the data-parallel gradient exchange is ONE RCCL all-reduce of the flat gradient buffer per optimiser step.
// Before a forward the parameters are re-packed (cheap, ≤ 8 M floats) into the MFMA fragment layouts of the
// forward and data-gradient convolutions.  Schedule per chunk:
//   augment ×8 → encode NHWC → [conv → BN(batch stats) → ReLU(+skip)]×(1+2R) → heads → losses
//   → head gradients → per layer in reverse: BN backward → weight gradient (TN implicit GEMM) → data gradient
// on three streams: the chain on the engine stream, everything that only produces gradients (weight gradients, bias finalisations)
// on a second one beside it, the next chunk's examples on a copy stream (tg_train).
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

#include "engine.h"
#include "kernels.h"
#include "rng.cuh"

namespace tg {

struct Id128 { char bytes[128]; };  // ncclUniqueId

namespace {

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct ParamInfo {
    std::string name;
    size_t off = 0, count = 0;
    bool buffer = false;  // BN running statistics (not trained)
};

struct TrainConv {
    int I = 0, O = 0;        // channels
    int in_stride = 0;       // row stride of its input activation
    int OP = 0;              // padded output channels of the forward fragments
    int out_stride = 0;      // row stride of its output (z)
    int bn = -1;             // BatchNorm index, -1 for the policy conv head
    size_t w = 0, b = 0, gamma = 0, beta = 0;  // offsets into params / grads
    size_t rmean = 0, rvar = 0;                // offsets into the BN buffer
    DevBuf wf, wb, bias_pad;                   // forward fragments, data-gradient fragments, padded bias
};

// Everything the chunk in flight owns: its inputs and targets, z / y of every layer, the heads' buffers, the backward pass's
// scratch and workspaces, its two streams.
struct Chunk {
    hipStream_t st = nullptr, wg = nullptr;  // the chain's stream (the engine stream) and the weight gradients' stream
    // chunk inputs / targets.  The caller's examples arrive in one of TWO sets of buffers (round 5): tg_train gathers and uploads chunk
    // k + 1 on a copy stream of its own while the GPU works on chunk k (the host's gather + five copies were 0.2 ms per chunk with the
    // GPU idle behind them: 18.28 against 18.07 ms per chunk through tg_train_chunk, whose caller does that work before the clock starts)
    struct Examples {
        DevBuf states, nmoves, moves, visits, zt;
        std::vector<uint8_t> h_states;   // host staging: alive until the set is uploaded again
        std::vector<int32_t> h_nm;
        std::vector<TgMove> h_moves;
        std::vector<uint32_t> h_visits;
        std::vector<float> h_z8;
        hipEvent_t uploaded = nullptr;   // recorded on the copy stream behind the set's five copies
        // the chunk that runs on this set: issued, not yet collected.  `done` stands behind its loss sums (and its optimiser step, if one
        // fell due) on the chain's stream: chunk_collect waits for IT, not for the stream — tg_train has the next chunk enqueued by then
        hipEvent_t done = nullptr;
        bool in_flight = false;
        int B_flight = 0, did_step = 0;
    } ex[2];
    hipStream_t up = nullptr;            // the copy stream
    DevBuf states_aug, pi, planes;
    std::vector<DevBuf> z, y;  // per conv layer: conv output, activation after BN / ReLU (/ skip)
    // heads
    DevBuf logits, dlogits, logp, eval, dpre, loss_p_rows, loss_z_rows, loss_sums;
    // backward: dz alternates between two buffers (the weight gradient of layer l reads one while the chain fills the other)
    DevBuf d_a, d_b, dz, dz2, gskip, stats, mean_g, mean_gx;
    DevBuf part_d, part_w;  // workspaces: the chain's double partials, the weight gradients' split-K partials
    DevBuf part_h;          // two halves: the heads' bias / value gradients' partials (weight gradients' stream)
    DevBuf part_b[2];       // dz's column sums (the conv bias gradient's partials), alternating like dz
    // ev_dz[k]: dz buffer k holds this layer's dz (chain → weight gradients); ev_head: the forward pass is complete;
    // ev_wgl[l]: conv l's weight gradient has read its dz buffer (index convs.size() = the heads)
    hipEvent_t ev_dz[2] = {nullptr, nullptr}, ev_head = nullptr;
    std::vector<hipEvent_t> ev_wgl;
    // tg_train_debug_capture: the backward pass of the next chunks keeps layer cap_layer's dy (gradient w.r.t. its activation,
    // before the ReLU mask), dz and the data gradient it hands down
    int cap_layer = -1;
    DevBuf cap_dy, cap_dz, cap_dx;
    ~Chunk() {
        for (hipEvent_t ev : ev_dz) if (ev) (void)hipEventDestroy(ev);
        if (ev_head) (void)hipEventDestroy(ev_head);
        for (hipEvent_t ev : ev_wgl) if (ev) (void)hipEventDestroy(ev);
        if (wg) (void)hipStreamDestroy(wg);
        for (Examples& x : ex) { if (x.uploaded) (void)hipEventDestroy(x.uploaded); if (x.done) (void)hipEventDestroy(x.done); }
        if (up) (void)hipStreamDestroy(up);
    }
};

// ---- RCCL through dlopen: self-play users never load it -----------------------------------------
// ONE copy of RCCL per process, on purpose: if the process has already mapped a librccl.so.1 — PyTorch's torch/lib/librccl.so
// carries that soname, and torch.distributed's NCCL backend is how bench.py's ranks meet — that copy is bound
// (dlopen RTLD_NOLOAD probe), so proxy threads, IPC handles and HSA signal pools exist once; only a process without one (the Rust
// host, a C host) loads librccl.so.1 from the loader's search path (/opt/rocm/lib).  tg_train_comm_info reports which file
// ncclAllReduce came from (dladdr) and what ncclGetVersion / ncclCommCount say, so a launch log answers "did RCCL see N ranks,
// and which RCCL".
struct Rccl {
    void* lib = nullptr;
    bool was_mapped = false;  // the RTLD_NOLOAD probe found a copy already in the process
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /*ncclUniqueId by value*/, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return TG_OK;
    bool mapped = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) { mapped = false; h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL); }
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(TG_ERR_STATE, std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(h, "ncclGetVersion");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(TG_ERR_STATE, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce");
    g_rccl.lib = h;
    g_rccl.was_mapped = mapped;
    return TG_OK;
}
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h
}  // namespace

struct Trainer {
    TgTrainConfig cfg{};
    int Bmax = 0;  // positions per chunk
    std::vector<ParamInfo> infos;
    std::map<std::string, int> index;
    size_t n_params = 0, n_buffers = 0;
    DevBuf params, grads, adam_m, adam_v, bnbuf;
    std::vector<TrainConv> convs;  // conv0, res0.conv1, res0.conv2, …
    bool conv_head = false;
    TrainConv pol;                 // conv policy head
    // FC policy head
    size_t fc_w = 0, fc_b = 0;
    int NP = 0, Pp = 0, KP = 0;
    DevBuf fc_wf, fc_wb, fc_bias;
    // value head
    size_t val_w = 0, val_b = 0;
    DevBuf wv;
    Chunk chunk;
    DevBuf zero_bias;
    bool packed = false;
    uint64_t adam_t = 0;
    int chunk_num = 0;
    // communicator (RCCL) or caller-supplied reduction (tg_train_set_allreduce)
    void* comm = nullptr;
    int world = 1, rank = 0;
    TgAllReduceFn hook = nullptr;
    void* hook_ctx = nullptr;
    // gradient all-reduces timed with HIP events on the engine stream (tg_train_comm_stats): TWO event pairs used in turn.  tg_train
    // enqueues chunk k + 1 — with its optimiser step, if one falls due — before it collects chunk k, so with chunks_in_step = 1 the
    // previous step's pair is still running when the next step is issued; the pair before THAT belongs to a chunk that has been
    // collected (its `done` event stands behind the step), so folding it never blocks the host and no step is dropped from the count.
    hipEvent_t ar_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool ar_pending[2] = {false, false};
    int ar_next = 0;
    double ar_ms = 0.0;
    int64_t ar_count = 0;
    void ar_fold(int pair) {
        if (!ar_pending[pair]) return;
        float ms = 0.0f;
        if (hipEventSynchronize(ar_ev[pair][1]) == hipSuccess && hipEventElapsedTime(&ms, ar_ev[pair][0], ar_ev[pair][1]) == hipSuccess) {
            ar_ms += ms;
            ar_count++;
        }
        ar_pending[pair] = false;
    }
    DevBuf err_flag;  // one float: the ranks agree on an argument error before the first chunk of tg_train
    ~Trainer() {
        for (auto& pair : ar_ev) for (hipEvent_t ev : pair) if (ev) (void)hipEventDestroy(ev);
        if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
    }
};

void trainer_destroy(Trainer* t) { delete t; }

namespace {

size_t add_info(Trainer* t, const std::string& name, size_t count, bool buffer) {
    ParamInfo pi;
    pi.name = name;
    pi.count = count;
    pi.buffer = buffer;
    size_t& cursor = buffer ? t->n_buffers : t->n_params;
    pi.off = cursor;
    cursor += (count + 3) / 4 * 4;  // 16-byte aligned slots; the padding stays zero
    t->index[name] = (int)t->infos.size();
    t->infos.push_back(pi);
    return pi.off;
}

void add_conv(Trainer* t, TrainConv& c, const std::string& conv, const std::string& bn, int O, int I) {
    c.O = O; c.I = I;
    c.w = add_info(t, conv + ".weight", (size_t)O * I * 9, false);
    c.b = add_info(t, conv + ".bias", O, false);
    (void)bn;
}
void add_bn(Trainer* t, TrainConv& c, const std::string& bn, int F, int bn_index) {
    c.bn = bn_index;
    c.gamma = add_info(t, bn + ".weight", F, false);
    c.beta = add_info(t, bn + ".bias", F, false);
    c.rmean = add_info(t, bn + ".running_mean", F, true);
    c.rvar = add_info(t, bn + ".running_var", F, true);
}

int pack_params(TgEngine* e, hipStream_t st) {
    Trainer* t = e->trainer;
    if (t->packed) return TG_OK;
    const float* P = t->params.as<float>();
    const int F = e->cfg.filters, nsq = e->g.nsq;
    auto pack_conv = [&](TrainConv& c, bool need_bwd) -> hipError_t {
        hipError_t err = launch_pack_conv_fwd(st, P + c.w, c.O, c.I, c.in_stride, c.OP, c.wf.as<float>());
        if (err != hipSuccess) return err;
        err = launch_pad_copy(st, P + c.b, c.O, c.OP, c.bias_pad.as<float>());
        if (err != hipSuccess) return err;
        if (need_bwd) err = launch_pack_conv_bwd(st, P + c.w, c.O, c.I, c.out_stride, round_up(c.I, 64), c.wb.as<float>());
        return err;
    };
    for (size_t l = 0; l < t->convs.size(); l++) TG_HIP(pack_conv(t->convs[l], l > 0));
    if (t->conv_head) TG_HIP(pack_conv(t->pol, true));
    else {
        TG_HIP(launch_pack_fc_fwd(st, P + t->fc_w, e->policy_size, F, nsq, t->NP, t->fc_wf.as<float>()));
        TG_HIP(launch_pack_fc_bwd(st, P + t->fc_w, e->policy_size, F, nsq, t->Pp, t->KP, t->fc_wb.as<float>()));
        TG_HIP(launch_pad_copy(st, P + t->fc_b, e->policy_size, t->NP, t->fc_bias.as<float>()));
    }
    TG_HIP(launch_pack_value(st, P + t->val_w, F, nsq, t->wv.as<float>()));
    t->packed = true;
    return TG_OK;
}

// forward in training mode from the NHWC planes of B positions; fills z/y of every layer, logits, eval
int forward_train(TgEngine* e, Chunk& w, int B, bool with_targets, float* d_logp, int slot = 0) {
    Trainer* t = e->trainer;
    hipStream_t st = w.st;
    const int F = e->cfg.filters, nsq = e->g.nsq, N = e->g.n, M = B * nsq;
    float* P = t->params.as<float>();
    float* BN = t->bnbuf.as<float>();
    float* stats = w.stats.as<float>();
    int rc = pack_params(e, st);
    if (rc) return rc;
    const size_t L = t->convs.size();
    for (size_t l = 0; l < L; l++) {
        TrainConv& c = t->convs[l];
        float* z = w.z[l].as<float>();
        const float* in = l == 0 ? w.planes.as<float>() : w.y[l - 1].as<float>();
        // the halo kernel (F → F layers at full chunks) hands out BatchNorm's column sums with the convolution; elsewhere two
        // reduction passes over z follow
        int stat_blocks = 0;
        static const bool conv_stats = !env_on("TG_NO_CONV_STATS");
        TG_HIP(launch_conv3x3(st, in, c.wf.as<float>(), c.bias_pad.as<float>(), nullptr, z, M, N, c.in_stride, c.OP, F, F, false,
                              (c.OP == F && conv_stats) ? w.part_d.as<double>() : nullptr, &stat_blocks, nullptr));
        float* mean = stats + (size_t)c.bn * 2 * F;
        float* invstd = mean + F;
        if (stat_blocks > 0)
            TG_HIP(launch_bn_stats_from_partials(st, w.part_d.as<double>(), stat_blocks, M, F, t->cfg.bn_eps, t->cfg.bn_momentum, mean, invstd,
                                                 BN + c.rmean, BN + c.rvar));
        else
            TG_HIP(launch_bn_stats(st, z, M, F, t->cfg.bn_eps, t->cfg.bn_momentum, w.part_d.as<double>(), mean, invstd, BN + c.rmean,
                                   BN + c.rvar));
        // conv2 of block i (l = 2, 4, …) adds the block input: y of layer l-2
        const float* skip = (l >= 2 && (l % 2) == 0) ? w.y[l - 2].as<float>() : nullptr;
        TG_HIP(launch_bn_fwd_apply(st, z, mean, invstd, P + c.gamma, P + c.beta, skip, w.y[l].as<float>(), M, F));
    }
    const float* s = w.y.back().as<float>();
    const float inv_b = 1.0f / (float)B;
    const float* pi = with_targets ? w.pi.as<float>() : nullptr;
    if (t->conv_head) {
        TrainConv& c = t->pol;
        TG_HIP(launch_conv3x3(st, s, c.wf.as<float>(), c.bias_pad.as<float>(), nullptr, w.logits.as<float>(), M, N, F, c.OP, c.OP, c.O, false));
        TG_HIP(launch_policy_loss(st, w.logits.as<float>(), nsq * c.OP, true, nsq, c.OP, e->policy_size, B, pi, inv_b,
                                  w.dlogits.as<float>(), d_logp, w.loss_p_rows.as<float>()));
    } else {
        TG_HIP(launch_gemm(st, s, nsq * F, t->fc_wf.as<float>(), t->fc_bias.as<float>(), w.logits.as<float>(), B, nsq * F, t->NP, t->NP,
                           e->policy_size));
        TG_HIP(launch_policy_loss(st, w.logits.as<float>(), t->NP, false, nsq, 0, e->policy_size, B, pi, inv_b, w.dlogits.as<float>(),
                                  d_logp, w.loss_p_rows.as<float>()));
    }
    TG_HIP(launch_value_train(st, s, t->wv.as<float>(), P + t->val_b, B, nsq * F, with_targets ? w.ex[slot].zt.as<float>() : nullptr, inv_b,
                              w.eval.as<float>(), w.dpre.as<float>(), w.loss_z_rows.as<float>()));
    return TG_OK;
}

// (behind forward_train)
int backward_train(TgEngine* e, Chunk& w, int B) {
    Trainer* t = e->trainer;
    hipStream_t st = w.st;
    const int F = e->cfg.filters, nsq = e->g.nsq, N = e->g.n, M = B * nsq;
    const int L = (int)t->convs.size();
    float* P = t->params.as<float>();
    float* G = t->grads.as<float>();
    float* stats = w.stats.as<float>();
    double* part_d = w.part_d.as<double>();
    float* part_w = w.part_w.as<float>();
    const float* s = w.y.back().as<float>();
    float* dcur = w.d_a.as<float>();
    float* dtmp = w.d_b.as<float>();
    float* dzb[2] = {w.dz.as<float>(), w.dz2.as<float>()};
    float* gskip = w.gskip.as<float>();
    const float* zero_bias = t->zero_bias.as<float>();
    // Two streams (round 4).  A layer's weight gradient (dz ⊗ x) and its data gradient (dz ∗ wᵀ, then the BatchNorm backward of the
    // layer below) only share their INPUT, so the weight gradients — the policy head's first — run on `wg` while the chain
    // heads → BatchNorm backward → data gradient → … stays on the engine stream, and with them everything else that only produces
    // gradients (conv bias finalisation, the heads' bias and value gradients): 19.3 – 19.6 → 18.1 – 18.5 ms per chunk of the C5 network.
    // What is gained is every launch's ramp and tail and the split-K reductions beside the other stream's kernel; the two MFMA kernels
    // of a layer share the machine and end together, so the chain's HBM-bound BatchNorm passes still run between them, not under them
    // (profiles/r04_g_train_overlap.txt).  Same kernels on the same operands, every gradient tensor still written by one launch → the
    // same bits as the single-stream order (TG_TRAIN_ONE_STREAM=1; tests/test_gpu_train.py).  dz alternates between two buffers;
    // the workspaces are per stream (part_w, part_h, part_b: weight gradients' stream, part_d: the chain).
    static const bool one_stream = env_on("TG_TRAIN_ONE_STREAM");
    hipStream_t wg = one_stream ? st : w.wg;
    const bool two = wg != st;
    if (two) {  // everything the forward pass left on the chain's stream precedes the first weight gradient
        TG_HIP(hipEventRecord(w.ev_head, st));
        TG_HIP(hipStreamWaitEvent(wg, w.ev_head, 0));
    }
    // ---- heads: dS = d(policy) + d(value) ----
    // (the heads' own gradients — policy weights and bias, value weights and bias — all on the weight gradients' stream, with workspaces
    // of their own: the chain starts with the data gradient the tower waits for)
    double* part_h0 = w.part_h.as<double>();
    double* part_h1 = (double*)((char*)w.part_h.p + w.part_h.bytes / 2);
    if (t->conv_head) {
        TrainConv& c = t->pol;
        const float* dl = w.dlogits.as<float>();
        TG_HIP(launch_conv3x3(st, dl, c.wb.as<float>(), zero_bias, nullptr, dcur, M, N, c.OP, round_up(F, 64), F, F, false));
        TG_HIP(launch_wgrad_conv(wg, s, F, F, dl, c.OP, c.O, B, N, part_w, G + c.w));
        TG_HIP(launch_colsum_acc(wg, dl, M, c.OP, c.O, part_h0, G + c.b));
    } else {
        const float* dl = w.dlogits.as<float>();
        TG_HIP(launch_gemm(st, dl, t->NP, t->fc_wb.as<float>(), zero_bias, dcur, B, t->Pp, t->KP, nsq * F, nsq * F));
        TG_HIP(launch_wgrad_fc(wg, s, nsq * F, dl, t->NP, e->policy_size, B, F, nsq, part_w, G + t->fc_w));
        TG_HIP(launch_colsum_acc(wg, dl, B, t->NP, e->policy_size, part_h0, G + t->fc_b));
    }
    TG_HIP(launch_value_bwd(st, s, w.dpre.as<float>(), t->wv.as<float>(), B, F, nsq, dcur, part_h1, G + t->val_w, G + t->val_b, wg));
    TG_HIP(hipEventRecord(w.ev_wgl[L], wg));
    // ---- tower, last layer first ----
    // The data-gradient convolution of layer l produces dy of layer l − 1 — its epilogue also takes that layer's
    // BatchNorm-backward sums Σg, Σg·x̂ while dy is in registers (halo kernel, full chunks), and the pass over dy, y and z that took
    // them (k_col_reduce, 27 µs per layer) is skipped; TG_NO_BWD_SUMS_FUSION restores it (other summation order: other low bits)
    static const bool fuse_sums = !env_on("TG_NO_BWD_SUMS_FUSION");
    const size_t act_bytes = (size_t)M * F * 4;
    int sums_in_part = 0;
    for (int l = L - 1; l >= 0; l--) {
        TrainConv& c = t->convs[l];
        float* mean = stats + (size_t)c.bn * 2 * F;
        float* invstd = mean + F;
        const int k = two ? (l & 1) : 0;
        float* dz = dzb[k];
        const bool block_end = l >= 2 && (l % 2) == 0;  // conv2: its masked gradient also flows into the skip
        const bool block_begin = (l % 2) == 1;          // conv1 closes the block: its data gradient joins the gradient that went through the skip
        const bool cap = l == w.cap_layer;
        if (two && l + 2 < L) TG_HIP(hipStreamWaitEvent(st, w.ev_wgl[l + 2], 0));  // layer l + 2's weight gradient has read this dz buffer
        const float* x = l == 0 ? w.planes.as<float>() : w.y[l - 1].as<float>();
        if (cap) TG_HIP(hipMemcpyAsync(w.cap_dy.p, dcur, act_bytes, hipMemcpyDeviceToDevice, st));
        // (the conv bias gradient — dz's column sums — is finalised on the weight gradients' stream: one launch less in the chain)
        int colsum_rows = 0;
        TG_HIP(launch_bn_bwd(st, dcur, w.y[l].as<float>(), w.z[l].as<float>(), mean, invstd, P + c.gamma, M, F, part_d, w.mean_g.as<double>(),
                             w.mean_gx.as<double>(), G + c.gamma, G + c.beta, dz, block_end ? gskip : nullptr, G + c.b, sums_in_part,
                             w.part_b[k].as<double>(), &colsum_rows));
        sums_in_part = 0;
        if (cap) TG_HIP(hipMemcpyAsync(w.cap_dz.p, dz, act_bytes, hipMemcpyDeviceToDevice, st));
        if (two) {
            TG_HIP(hipEventRecord(w.ev_dz[k], st));
            TG_HIP(hipStreamWaitEvent(wg, w.ev_dz[k], 0));
        }
        TG_HIP(launch_colsum_finalize(wg, w.part_b[k].as<double>(), colsum_rows, F, F, G + c.b));
        TG_HIP(launch_wgrad_conv(wg, x, c.in_stride, c.I, dz, F, c.O, B, N, part_w, G + c.w));
        TG_HIP(hipEventRecord(w.ev_wgl[l], wg));
        if (l == 0) break;
        float* dst = block_end ? dtmp : dcur;
        const TrainConv& below = t->convs[l - 1];
        const float* mean_b = stats + (size_t)below.bn * 2 * F;
        const ConvBnBwdIn bnb{w.y[l - 1].as<float>(), w.z[l - 1].as<float>(), mean_b, mean_b + F};
        // (in place where dst = dcur: a workgroup reads the rows of its own positions and writes them after its last read)
        TG_HIP(launch_conv3x3(st, dz, c.wb.as<float>(), zero_bias, block_begin ? gskip : nullptr, dst, M, N, F,
                              round_up(c.I, 64), F, F, false, fuse_sums ? part_d : nullptr, fuse_sums ? &sums_in_part : nullptr,
                              fuse_sums ? &bnb : nullptr));
        if (cap) TG_HIP(hipMemcpyAsync(w.cap_dx.p, dst, act_bytes, hipMemcpyDeviceToDevice, st));
        if (block_end) std::swap(dcur, dtmp);
    }
    // the chain's stream continues (loss sums, optimiser step, the next chunk) behind the last weight gradient
    if (two) TG_HIP(hipStreamWaitEvent(st, w.ev_wgl[0], 0));
    return TG_OK;
}

// Sum a device buffer over the ranks, in place, ordered on `st`: the caller's hook if one is set, else RCCL.
// Returns false through *reduced when the trainer is single-rank (nothing to do).
static int all_reduce_sum(TgEngine* e, hipStream_t st, float* d_buf, size_t count, const char* what, bool* reduced) {
    Trainer* t = e->trainer;
    *reduced = false;
    if (t->hook) {
        int rc = t->hook(t->hook_ctx, d_buf, count, (void*)st);
        if (rc) return fail(TG_ERR_STATE, std::string("all-reduce hook failed (") + what + "), code " + std::to_string(rc));
        *reduced = true;
    } else if (t->comm) {
        int rc = g_rccl.AllReduce(d_buf, d_buf, count, NCCL_FLOAT32, NCCL_SUM, t->comm, st);
        if (rc) return fail(TG_ERR_HIP, std::string("ncclAllReduce (") + what + "): " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
        *reduced = true;
    }
    return TG_OK;
}

// on `st`, behind every chunk of the step (the chain's stream has waited for the chunk's weight gradients)
int optimizer_step(TgEngine* e, hipStream_t st) {
    Trainer* t = e->trainer;
    float gscale = 1.0f;
    {
        bool reduced;
        const bool timed = t->hook || t->comm;
        const int pair = t->ar_next;
        if (timed) {
            t->ar_fold(pair);  // the step before the previous one: its chunk has been collected, the events are complete
            if (!t->ar_ev[pair][0]) {
                TG_HIP(hipEventCreate(&t->ar_ev[pair][0]));
                TG_HIP(hipEventCreate(&t->ar_ev[pair][1]));
            }
            TG_HIP(hipEventRecord(t->ar_ev[pair][0], st));
        }
        int rc = all_reduce_sum(e, st, t->grads.as<float>(), t->n_params, "gradients", &reduced);
        if (timed) {
            t->ar_pending[pair] = hipEventRecord(t->ar_ev[pair][1], st) == hipSuccess;
            t->ar_next = pair ^ 1;
        }
        if (rc) return rc;
        if (reduced) gscale = 1.0f / (float)t->world;
    }
    t->adam_t++;
    const double b1 = t->cfg.beta1, b2 = t->cfg.beta2;
    float bc1 = (float)(1.0 - std::pow(b1, (double)t->adam_t));
    float bc2s = (float)std::sqrt(1.0 - std::pow(b2, (double)t->adam_t));
    TG_HIP(launch_adam(st, t->params.as<float>(), t->grads.as<float>(), t->adam_m.as<float>(), t->adam_v.as<float>(), t->n_params,
                       t->cfg.learning_rate, t->cfg.beta1, t->cfg.beta2, t->cfg.eps, t->cfg.weight_decay, bc1, bc2s, gscale));
    TG_HIP(hipMemsetAsync(t->grads.p, 0, t->n_params * 4, st));
    t->packed = false;
    return pack_params(e, st);
}

int need_trainer(TgEngine* e) {
    if (!e) return fail(TG_ERR_INVALID_ARG, "null engine");
    if (!e->trainer) return fail(TG_ERR_STATE, "no trainer (tg_train_create)");
    hipError_t err = hipSetDevice(e->cfg.device);
    if (err != hipSuccess) return fail(TG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(err));
    return TG_OK;
}

// One chunk whose upload into example set `slot` has been issued (n examples, zt the 8n value targets): everything up
// to and including an optimiser step that falls due is ISSUED on the two streams; chunk_collect waits for it and reads the losses.
int chunk_issue(TgEngine* e, Chunk& w, int n, int slot) {
    Trainer* t = e->trainer;
    hipStream_t st = w.st;
    const int B = n * 8;
    Chunk::Examples& x = w.ex[slot];
    TG_HIP(hipStreamWaitEvent(st, x.uploaded, 0));
    TG_HIP(hipMemsetAsync(w.pi.p, 0, (size_t)B * e->policy_size * 4, st));
    launch_augment(st, x.states.as<uint8_t>(), x.nmoves.as<int32_t>(), x.moves.as<uint16_t>(), x.visits.as<uint32_t>(), n,
                   e->g.n, e->policy_size, e->legacy5, e->lut5.as<int16_t>(), w.states_aug.as<uint8_t>(), w.pi.as<float>());
    TG_HIP(hipGetLastError());
    launch_encode_nhwc(st, w.states_aug.as<uint8_t>(), B, e->g.n, w.planes.as<float>(), e->cin_pad);
    TG_HIP(hipGetLastError());
    int rc = forward_train(e, w, B, true, nullptr, slot);
    if (rc) return rc;
    rc = backward_train(e, w, B);
    if (rc) return rc;
    TG_HIP(launch_sum_rows(st, w.loss_p_rows.as<float>(), B, w.loss_sums.as<double>() + 2 * slot));
    TG_HIP(launch_sum_rows(st, w.loss_z_rows.as<float>(), B, w.loss_sums.as<double>() + 2 * slot + 1));
    x.did_step = 0;
    x.B_flight = B;
    x.in_flight = true;
    t->chunk_num++;
    if (t->chunk_num % t->cfg.chunks_in_step == 0) {  // network.rs:92
        rc = optimizer_step(e, st);
        if (rc) return rc;
        x.did_step = 1;
    }
    TG_HIP(hipEventRecord(x.done, st));
    return TG_OK;
}
int chunk_collect(TgEngine* e, Chunk& w, int slot, float* loss_p, float* loss_z, int32_t* stepped) {
    (void)e;
    Chunk::Examples& x = w.ex[slot];
    if (!x.in_flight) return fail(TG_ERR_STATE, "internal: no chunk in flight on this example set");
    x.in_flight = false;
    double sums[2];
    // behind the chunk's own `done` event, on the copy stream: the chain's stream may already hold the next chunk
    TG_HIP(hipStreamWaitEvent(w.up, x.done, 0));
    TG_HIP(hipMemcpyAsync(sums, w.loss_sums.as<double>() + 2 * slot, 16, hipMemcpyDeviceToHost, w.up));
    TG_HIP(hipStreamSynchronize(w.up));
    if (loss_p) *loss_p = (float)(sums[0] / x.B_flight);
    if (loss_z) *loss_z = (float)(sums[1] / x.B_flight);
    if (stepped) *stepped = x.did_step;
    return TG_OK;
}
// after an error in the middle of a chunk: nothing of this trainer is left running
void chunk_drain(Trainer* t) {
    Chunk& w = t->chunk;
    if (w.up) (void)hipStreamSynchronize(w.up);
    if (w.wg) (void)hipStreamSynchronize(w.wg);
    if (w.st) (void)hipStreamSynchronize(w.st);
    for (Chunk::Examples& x : w.ex) x.in_flight = false;
}

// The complete host-side check of ONE example (index s of the caller's arrays), used by tg_train_chunk's upload and by
// tg_train's pass over all examples: a reachable state, 1 ≤ n_moves ≤ TG_MAX_MOVES, at least one visit.
int validate_example(const TgEngine* e, int s, const uint8_t* states, const int32_t* n_moves, const uint32_t* visits) {
    int vrc = validate_states(e, 1, states + (size_t)s * e->g.bytes, "training example");
    if (vrc) return vrc;
    if (n_moves[s] <= 0 || n_moves[s] > TG_MAX_MOVES)
        return fail(TG_ERR_INVALID_ARG, "training example " + std::to_string(s) + ": n_moves out of range");
    uint64_t total = 0;
    for (int k = 0; k < n_moves[s]; k++) total += visits[(size_t)s * TG_MAX_MOVES + k];
    if (total == 0) return fail(TG_ERR_INVALID_ARG, "training example " + std::to_string(s) + " without visits (the policy target would be 0/0)");
    return TG_OK;
}

// Gathers n examples (in `order`, or as they stand) into example set `slot`'s host staging and issues their five copies on the copy
// stream; chunk_issue makes the chunk wait for them.  The set must not be in use: its previous chunk has been collected.
// validated = the caller (tg_train) has already checked every example
int upload_chunk(TgEngine* e, Chunk& w, int slot, int n, const uint8_t* states, const int32_t* n_moves, const TgMove* moves,
                 const uint32_t* visits, const float* results, const int* order, bool validated = false) {
    Chunk::Examples& x = w.ex[slot];
    const size_t sb = e->g.bytes;
    x.h_z8.resize((size_t)n * 8);
    x.h_states.resize((size_t)n * sb);
    x.h_nm.resize(n);
    x.h_moves.resize((size_t)n * TG_MAX_MOVES);
    x.h_visits.resize((size_t)n * TG_MAX_MOVES);
    for (int i = 0; i < n; i++) {
        const int s = order ? order[i] : i;
        if (!validated) {
            int vrc = validate_example(e, s, states, n_moves, visits);
            if (vrc) return vrc;
        }
        for (int k = 0; k < 8; k++) x.h_z8[(size_t)i * 8 + k] = results[s];
        std::memcpy(x.h_states.data() + (size_t)i * sb, states + (size_t)s * sb, sb);
        x.h_nm[i] = n_moves[s];
        std::memcpy(x.h_moves.data() + (size_t)i * TG_MAX_MOVES, moves + (size_t)s * TG_MAX_MOVES, (size_t)TG_MAX_MOVES * sizeof(TgMove));
        std::memcpy(x.h_visits.data() + (size_t)i * TG_MAX_MOVES, visits + (size_t)s * TG_MAX_MOVES, (size_t)TG_MAX_MOVES * 4);
    }
    hipStream_t up = w.up;
    TG_HIP(hipMemcpyAsync(x.states.p, x.h_states.data(), (size_t)n * sb, hipMemcpyHostToDevice, up));
    TG_HIP(hipMemcpyAsync(x.nmoves.p, x.h_nm.data(), (size_t)n * 4, hipMemcpyHostToDevice, up));
    TG_HIP(hipMemcpyAsync(x.moves.p, x.h_moves.data(), (size_t)n * TG_MAX_MOVES * 2, hipMemcpyHostToDevice, up));
    TG_HIP(hipMemcpyAsync(x.visits.p, x.h_visits.data(), (size_t)n * TG_MAX_MOVES * 4, hipMemcpyHostToDevice, up));
    TG_HIP(hipMemcpyAsync(x.zt.p, x.h_z8.data(), x.h_z8.size() * 4, hipMemcpyHostToDevice, up));
    TG_HIP(hipEventRecord(x.uploaded, up));
    return TG_OK;
}

// refs.shuffle (network.rs:49-50): Fisher–Yates driven by Philox(seed; i)
void shuffle_order(uint64_t seed, int n, int* order) {
    std::iota(order, order + n, 0);
    for (int i = n - 1; i > 0; i--) {
        U4 r = philox4x32_10(seed, (uint32_t)i, 0x7261696eu, 0, 0);
        uint64_t x = ((uint64_t)r.v[0] << 32) | r.v[1];
        int j = (int)(((unsigned __int128)x * (unsigned __int128)(i + 1)) >> 64);
        std::swap(order[i], order[j]);
    }
}

}  // namespace
}  // namespace tg

using namespace tg;

extern "C" {

int tg_train_create(TgEngine* e, const TgTrainConfig* cfg) {
    if (!e || !cfg) return fail(TG_ERR_INVALID_ARG, "tg_train_create: null argument");
    const auto* tensors = net_tensors(e);
    if (!tensors) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    if (cfg->chunk_size <= 0 || cfg->chunks_in_step <= 0) return fail(TG_ERR_INVALID_ARG, "chunk_size and chunks_in_step must be positive");
    const int F = e->cfg.filters, R = e->cfg.res_blocks, nsq = e->g.nsq, N = e->g.n, P = e->policy_size;
    if (256 % (F / 4) != 0) return fail(TG_ERR_INVALID_ARG, "training needs filters in {32, 64, 128, 256}");
    TG_HIP(hipSetDevice(e->cfg.device));
    std::unique_ptr<Trainer> t(new Trainer());
    t->cfg = *cfg;
    t->Bmax = cfg->chunk_size * 8;
    t->conv_head = e->cfg.policy_head == TG_HEAD_CONV;
    // ---- parameter registry, creation order of net5.rs:29-62 / net6.rs:29-57 ----
    t->convs.resize(1 + 2 * R);
    int bn = 0;
    add_conv(t.get(), t->convs[0], "conv0", "bn0", F, e->cin);
    add_bn(t.get(), t->convs[0], "bn0", F, bn++);
    for (int i = 0; i < R; i++) {
        std::string p = "res" + std::to_string(i);
        add_conv(t.get(), t->convs[1 + 2 * i], p + ".conv1", "", F, F);
        add_conv(t.get(), t->convs[2 + 2 * i], p + ".conv2", "", F, F);
        add_bn(t.get(), t->convs[1 + 2 * i], p + ".bn1", F, bn++);
        add_bn(t.get(), t->convs[2 + 2 * i], p + ".bn2", F, bn++);
    }
    if (t->conv_head) add_conv(t.get(), t->pol, "policy", "", P / nsq, F);
    else {
        t->fc_w = add_info(t.get(), "policy.weight", (size_t)P * F * nsq, false);
        t->fc_b = add_info(t.get(), "policy.bias", P, false);
    }
    t->val_w = add_info(t.get(), "value.weight", (size_t)F * nsq, false);
    t->val_b = add_info(t.get(), "value.bias", 1, false);
    // ---- upload the tensors ----
    std::vector<float> hp(t->n_params, 0.0f), hb(t->n_buffers, 0.0f);
    for (const ParamInfo& pi : t->infos) {
        auto it = tensors->find(pi.name);
        if (it == tensors->end()) return fail(TG_ERR_WEIGHTS, "missing tensor " + pi.name);
        if (it->second.size() != pi.count)
            return fail(TG_ERR_WEIGHTS, "tensor " + pi.name + " has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(pi.count));
        std::memcpy((pi.buffer ? hb.data() : hp.data()) + pi.off, it->second.data(), pi.count * 4);
    }
    TG_HIP(t->params.ensure(t->n_params * 4));
    TG_HIP(t->grads.ensure(t->n_params * 4));
    TG_HIP(t->adam_m.ensure(t->n_params * 4));
    TG_HIP(t->adam_v.ensure(t->n_params * 4));
    TG_HIP(t->bnbuf.ensure(t->n_buffers * 4));
    TG_HIP(hipMemcpy(t->params.p, hp.data(), t->n_params * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(t->bnbuf.p, hb.data(), t->n_buffers * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(t->grads.p, 0, t->n_params * 4));
    TG_HIP(hipMemset(t->adam_m.p, 0, t->n_params * 4));
    TG_HIP(hipMemset(t->adam_v.p, 0, t->n_params * 4));
    // ---- per-layer buffers ----
    const size_t B = (size_t)t->Bmax, M = B * nsq;
    size_t part_w_floats = 0;
    int max_op = round_up(F, 64);
    for (size_t l = 0; l < t->convs.size(); l++) {
        TrainConv& c = t->convs[l];
        c.in_stride = l == 0 ? e->cin_pad : F;
        c.OP = round_up(F, 64);
        c.out_stride = F;
        TG_HIP(c.wf.ensure((size_t)9 * c.in_stride * c.OP * 4));
        TG_HIP(c.bias_pad.ensure((size_t)c.OP * 4));
        if (l > 0) TG_HIP(c.wb.ensure((size_t)9 * c.out_stride * round_up(c.I, 64) * 4));
        part_w_floats = std::max(part_w_floats, wgrad_conv_workspace((int)B, N, c.I, c.O));
    }
    size_t logit_row;
    if (t->conv_head) {
        TrainConv& c = t->pol;
        c.in_stride = F;
        c.OP = round_up(c.O, 64);
        c.out_stride = c.OP;
        max_op = std::max(max_op, c.OP);
        TG_HIP(c.wf.ensure((size_t)9 * F * c.OP * 4));
        TG_HIP(c.bias_pad.ensure((size_t)c.OP * 4));
        TG_HIP(c.wb.ensure((size_t)9 * c.OP * round_up(F, 64) * 4));
        part_w_floats = std::max(part_w_floats, wgrad_conv_workspace((int)B, N, F, c.O));
        logit_row = (size_t)nsq * c.OP;
    } else {
        const int K = F * nsq;
        t->NP = (K % 64 == 0) ? round_up(P, 208) : round_up(P, 64);
        t->Pp = round_up(P, 32);
        t->KP = round_up(K, 64);
        if (t->Pp > t->NP) return fail(TG_ERR_INVALID_ARG, "internal: FC padding");
        max_op = std::max(max_op, std::max(t->NP, t->KP));
        TG_HIP(t->fc_wf.ensure((size_t)K * t->NP * 4));
        TG_HIP(t->fc_wb.ensure((size_t)t->Pp * t->KP * 4));
        TG_HIP(t->fc_bias.ensure((size_t)t->NP * 4));
        part_w_floats = std::max(part_w_floats, wgrad_fc_workspace((int)B, K, P));
        logit_row = (size_t)t->NP;
    }
    TG_HIP(t->wv.ensure((size_t)F * nsq * 4));
    TG_HIP(t->zero_bias.ensure((size_t)max_op * 4));
    TG_HIP(hipMemset(t->zero_bias.p, 0, (size_t)max_op * 4));
    // ---- what the chunk in flight owns ----
    size_t part_d_bytes;
    {   // double partials: column reductions over up to max(F, logit columns) channels, value weight gradient
        int rpb;
        size_t a = (size_t)col_reduce_blocks((int)M, F, &rpb) * 2 * F;
        if (t->conv_head) a = std::max(a, (size_t)col_reduce_blocks((int)M, t->pol.OP, &rpb) * 2 * t->pol.OP);
        else a = std::max(a, (size_t)32 * 2 * t->NP);
        a = std::max(a, (size_t)(2 * ((size_t)B + 16)) * 2 * F);  // the conv kernels' partial rows: ≤ 2 per position
        size_t b = (size_t)32 * ((size_t)F * nsq + 1);
        part_d_bytes = std::max(a, b) * 8;
    }
    const size_t L = t->convs.size();
    {
        Chunk& w = t->chunk;
        w.st = e->stream;
        // (stream priorities — the weight gradients below the chain — were measured: 1 – 12 % slower)
        TG_HIP(hipStreamCreateWithFlags(&w.wg, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) TG_HIP(hipEventCreateWithFlags(&w.ev_dz[k], hipEventDisableTiming));
        TG_HIP(hipEventCreateWithFlags(&w.ev_head, hipEventDisableTiming));
        w.ev_wgl.assign(L + 1, nullptr);
        for (hipEvent_t& ev : w.ev_wgl) TG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        w.z.resize(L);
        w.y.resize(L);
        for (size_t l = 0; l < L; l++) {
            TG_HIP(w.z[l].ensure(M * F * 4));
            TG_HIP(w.y[l].ensure(M * F * 4));
        }
        TG_HIP(hipStreamCreateWithFlags(&w.up, hipStreamNonBlocking));
        for (Chunk::Examples& x : w.ex) {
            TG_HIP(x.states.ensure((size_t)cfg->chunk_size * e->g.bytes));
            TG_HIP(x.nmoves.ensure((size_t)cfg->chunk_size * 4));
            TG_HIP(x.moves.ensure((size_t)cfg->chunk_size * TG_MAX_MOVES * 2));
            TG_HIP(x.visits.ensure((size_t)cfg->chunk_size * TG_MAX_MOVES * 4));
            TG_HIP(hipMemset(x.moves.p, 0, (size_t)cfg->chunk_size * TG_MAX_MOVES * 2));
            TG_HIP(hipMemset(x.visits.p, 0, (size_t)cfg->chunk_size * TG_MAX_MOVES * 4));
            TG_HIP(x.zt.ensure(B * 4));
            TG_HIP(hipEventCreateWithFlags(&x.uploaded, hipEventDisableTiming));
            TG_HIP(hipEventCreateWithFlags(&x.done, hipEventDisableTiming));
        }
        TG_HIP(w.states_aug.ensure(B * e->g.bytes));
        TG_HIP(w.pi.ensure(B * P * 4));
        TG_HIP(w.planes.ensure(M * e->cin_pad * 4));
        TG_HIP(w.logits.ensure(B * logit_row * 4));
        TG_HIP(w.dlogits.ensure(B * logit_row * 4));
        TG_HIP(hipMemset(w.dlogits.p, 0, B * logit_row * 4));  // padding columns stay zero
        TG_HIP(w.logp.ensure(B * P * 4));
        TG_HIP(w.eval.ensure(B * 4));
        TG_HIP(w.dpre.ensure(B * 4));
        TG_HIP(w.loss_p_rows.ensure(B * 4));
        TG_HIP(w.loss_z_rows.ensure(B * 4));
        TG_HIP(w.loss_sums.ensure(32));
        TG_HIP(w.d_a.ensure(M * F * 4));
        TG_HIP(w.d_b.ensure(M * F * 4));
        TG_HIP(w.dz.ensure(M * F * 4));
        TG_HIP(w.dz2.ensure(M * F * 4));
        TG_HIP(w.gskip.ensure(M * F * 4));
        TG_HIP(w.stats.ensure((size_t)bn * 2 * F * 4));
        TG_HIP(w.mean_g.ensure((size_t)F * 8));
        TG_HIP(w.mean_gx.ensure((size_t)F * 8));
        TG_HIP(w.part_d.ensure(part_d_bytes));
        TG_HIP(w.part_w.ensure(part_w_floats * 4));
        for (int k = 0; k < 2; k++) TG_HIP(w.part_b[k].ensure(part_d_bytes));
        TG_HIP(w.part_h.ensure(2 * ((part_d_bytes + 255) / 256 * 256)));
    }
    delete e->trainer;
    e->trainer = t.release();
    return TG_OK;
}

int tg_train_chunk(TgEngine* e, int n, const void* states, const int32_t* n_moves, const TgMove* moves, const uint32_t* visits,
                   const float* results, float* loss_p, float* loss_z, int32_t* stepped) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (n <= 0 || n > e->trainer->cfg.chunk_size || !states || !n_moves || !moves || !visits || !results)
        return fail(TG_ERR_INVALID_ARG, "tg_train_chunk: bad arguments (1 ≤ n ≤ chunk_size)");
    Chunk& w = e->trainer->chunk;
    rc = upload_chunk(e, w, 0, n, (const uint8_t*)states, n_moves, moves, visits, results, nullptr);
    if (rc) return rc;
    rc = chunk_issue(e, w, n, 0);
    if (rc) {
        chunk_drain(e->trainer);
        return rc;
    }
    rc = chunk_collect(e, w, 0, loss_p, loss_z, stepped);
    if (rc) {  // the chunk (and an optimiser step that fell due) may still be running on both streams: nothing is left behind
        chunk_drain(e->trainer);
        return rc;
    }
    TG_HIP(hipStreamSynchronize(w.st));  // (tg_train_chunk returns with the engine stream idle, as it always did)
    return TG_OK;
}

int tg_train(TgEngine* e, int n, const void* states, const int32_t* n_moves, const TgMove* moves, const uint32_t* visits,
             const float* results, uint64_t seed, float* mean_loss_p, float* mean_loss_z, int32_t* steps) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!states || !n_moves || !moves || !visits || !results))) return fail(TG_ERR_INVALID_ARG, "tg_train: bad arguments");
    Trainer* t = e->trainer;
    // Every example is checked — completely, once — before the first chunk.  With several ranks the verdicts are then summed
    // through the reduction (RCCL or the hook): a rank whose examples are fine returns TG_ERR_INVALID_ARG together with the
    // rank that found the bad one, instead of waiting for it in the first optimiser step's all-reduce.
    int bad = TG_OK;
    for (int i = 0; i < n && bad == TG_OK; i++) bad = validate_example(e, i, (const uint8_t*)states, n_moves, visits);
    if (t->hook || t->comm) {
        const std::string local_msg = bad ? tg_last_error() : "";
        TG_HIP(t->err_flag.ensure(16));
        const float mine[4] = {bad ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f};
        TG_HIP(hipMemcpyAsync(t->err_flag.p, mine, 16, hipMemcpyHostToDevice, e->stream));
        bool reduced;
        rc = all_reduce_sum(e, e->stream, t->err_flag.as<float>(), 4, "argument check", &reduced);
        if (rc) return rc;
        float all[4];
        TG_HIP(hipMemcpyAsync(all, t->err_flag.p, 16, hipMemcpyDeviceToHost, e->stream));
        TG_HIP(hipStreamSynchronize(e->stream));
        if (bad) return fail(TG_ERR_INVALID_ARG, local_msg);
        if (all[0] != 0.0f) return fail(TG_ERR_INVALID_ARG, "tg_train: " + std::to_string((int)all[0]) + " other rank(s) refused their examples; no rank trains");
    } else if (bad) return bad;
    // a fresh optimiser per call (network.rs:40-45) on fresh gradients: the reference trains a fresh copy of the network
    // each round (train/src/main.rs `copy`: save + load into a new VarStore), so the gradients an incomplete last step
    // left behind (n / chunk_size not a multiple of chunks_in_step) never reach the next call's first step
    TG_HIP(hipMemsetAsync(t->grads.p, 0, t->n_params * 4, e->stream));
    TG_HIP(hipMemsetAsync(t->adam_m.p, 0, t->n_params * 4, e->stream));
    TG_HIP(hipMemsetAsync(t->adam_v.p, 0, t->n_params * 4, e->stream));
    t->adam_t = 0;
    t->chunk_num = 0;
    std::vector<int> order(n);
    shuffle_order(seed, n, order.data());
    const int cs = t->cfg.chunk_size;
    double sp = 0.0, sz = 0.0;
    int chunks = 0, nsteps = 0;
    Chunk& w = t->chunk;
    // chunks_exact: the remainder is dropped.  While the GPU works on chunk k the host gathers and uploads chunk k + 1 (copy stream, the other
    // example set) and ENQUEUES it behind chunk k, then waits for chunk k's losses only (its `done` event): the chunks run one after
    // the other on the same streams as before, with no host round trip between them.
    const int total = n / cs;
    if (total > 0) {
        rc = upload_chunk(e, w, 0, cs, (const uint8_t*)states, n_moves, moves, visits, results, order.data(), true);
        if (rc == TG_OK) rc = chunk_issue(e, w, cs, 0);
    }
    for (int k = 0; k < total && rc == TG_OK; k++) {
        if (k + 1 < total) {
            rc = upload_chunk(e, w, (k + 1) & 1, cs, (const uint8_t*)states, n_moves, moves, visits, results, order.data() + (size_t)(k + 1) * cs, true);
            if (rc == TG_OK) rc = chunk_issue(e, w, cs, (k + 1) & 1);
        }
        float lp = 0.0f, lz = 0.0f;
        int32_t did = 0;
        if (rc == TG_OK) rc = chunk_collect(e, w, k & 1, &lp, &lz, &did);
        if (rc == TG_OK) { sp += lp; sz += lz; chunks++; nsteps += did; }
    }
    if (rc) {
        chunk_drain(t);
        return rc;
    }
    TG_HIP(hipStreamSynchronize(w.st));
    if (mean_loss_p) *mean_loss_p = chunks ? (float)(sp / chunks) : 0.0f;
    if (mean_loss_z) *mean_loss_z = chunks ? (float)(sz / chunks) : 0.0f;
    if (steps) *steps = nsteps;
    return TG_OK;
}

int tg_train_order(uint64_t seed, int n, int32_t* order) {
    if (n < 0 || (n > 0 && !order)) return fail(TG_ERR_INVALID_ARG, "tg_train_order: bad arguments");
    shuffle_order(seed, n, order);
    return TG_OK;
}

int tg_train_step(TgEngine* e) {
    int rc = need_trainer(e);
    if (rc) return rc;
    rc = optimizer_step(e, e->stream);
    if (rc) return rc;
    TG_HIP(hipStreamSynchronize(e->stream));
    return TG_OK;
}

int tg_train_forward(TgEngine* e, int n, const void* states, float* logp, float* eval) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    if (n <= 0 || n > t->Bmax || !states || !logp || !eval) return fail(TG_ERR_INVALID_ARG, "tg_train_forward: bad arguments (1 ≤ n ≤ 8·chunk_size)");
    Chunk& w = t->chunk;
    hipStream_t st = w.st;
    TG_HIP(hipMemcpyAsync(w.states_aug.p, states, (size_t)n * e->g.bytes, hipMemcpyHostToDevice, st));
    launch_encode_nhwc(st, w.states_aug.as<uint8_t>(), n, e->g.n, w.planes.as<float>(), e->cin_pad);
    TG_HIP(hipGetLastError());
    rc = forward_train(e, w, n, false, w.logp.as<float>());
    if (rc) return rc;
    TG_HIP(hipMemcpyAsync(logp, w.logp.p, (size_t)n * e->policy_size * 4, hipMemcpyDeviceToHost, st));
    TG_HIP(hipMemcpyAsync(eval, w.eval.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    TG_HIP(hipStreamSynchronize(st));
    return TG_OK;
}

static int get_common(TgEngine* e, const char* name, float* out, size_t count, bool grad) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    if (!name || !out) return fail(TG_ERR_INVALID_ARG, "null argument");
    auto it = t->index.find(name);
    if (it == t->index.end()) return fail(TG_ERR_INVALID_ARG, std::string("unknown tensor ") + name);
    const ParamInfo& pi = t->infos[it->second];
    if (pi.count != count) return fail(TG_ERR_INVALID_ARG, std::string("tensor ") + name + " has " + std::to_string(pi.count) + " elements");
    if (grad && pi.buffer) return fail(TG_ERR_INVALID_ARG, std::string(name) + " is a buffer and has no gradient");
    const float* src = (pi.buffer ? t->bnbuf.as<float>() : grad ? t->grads.as<float>() : t->params.as<float>()) + pi.off;
    TG_HIP(hipStreamSynchronize(e->stream));
    TG_HIP(hipMemcpy(out, src, count * 4, hipMemcpyDeviceToHost));
    return TG_OK;
}
int tg_train_get_tensor(TgEngine* e, const char* name, float* out, size_t count) { return get_common(e, name, out, count, false); }
int tg_train_get_grad(TgEngine* e, const char* name, float* out, size_t count) { return get_common(e, name, out, count, true); }

// Test / diagnostic access to the training step's intermediate tensors (tests/test_gpu_c5_realsize.py, scripts/train_error_budget.py):
// what = "planes" (NHWC input, [rows][cin_pad]), "z" / "y" (conv output, activation of layer `layer`, [rows][F]), "mean" / "invstd" (the
// batch statistics BatchNorm `layer` normalised with, [F]) of the last forward pass; "dy" / "dz" / "dx" of the layer armed by
// tg_train_debug_capture (gradient w.r.t. y before the ReLU mask, w.r.t. z, and the data gradient handed to the layer below — for
// conv1 of a block with the skip path's gradient added), [rows][F].  count = the number of floats expected.
int tg_train_debug_capture(TgEngine* e, int layer) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    Chunk& w = t->chunk;
    if (layer >= (int)t->convs.size()) return fail(TG_ERR_INVALID_ARG, "tg_train_debug_capture: no such layer");
    w.cap_layer = layer < 0 ? -1 : layer;
    if (layer >= 0) {
        const size_t bytes = (size_t)t->Bmax * e->g.nsq * e->cfg.filters * 4;
        TG_HIP(w.cap_dy.ensure(bytes));
        TG_HIP(w.cap_dz.ensure(bytes));
        TG_HIP(w.cap_dx.ensure(bytes));
    }
    return TG_OK;
}
int tg_train_debug_read(TgEngine* e, const char* what, int layer, float* out, size_t count) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (!what || !out) return fail(TG_ERR_INVALID_ARG, "null argument");
    Trainer* t = e->trainer;
    Chunk& w = t->chunk;
    const int L = (int)t->convs.size(), F = e->cfg.filters;
    const std::string k = what;
    const void* src = nullptr;
    size_t have = 0;
    if (k == "planes") { src = w.planes.p; have = w.planes.bytes / 4; }
    else if (layer < 0 || layer >= L) return fail(TG_ERR_INVALID_ARG, "tg_train_debug_read: no such layer");
    else if (k == "z") { src = w.z[layer].p; have = w.z[layer].bytes / 4; }
    else if (k == "y") { src = w.y[layer].p; have = w.y[layer].bytes / 4; }
    else if (k == "mean") { src = w.stats.as<float>() + (size_t)t->convs[layer].bn * 2 * F; have = F; }
    else if (k == "invstd") { src = w.stats.as<float>() + (size_t)t->convs[layer].bn * 2 * F + F; have = F; }
    else if (layer != w.cap_layer) return fail(TG_ERR_STATE, "tg_train_debug_read: layer not armed (tg_train_debug_capture)");
    else if (k == "dy") { src = w.cap_dy.p; have = w.cap_dy.bytes / 4; }
    else if (k == "dz") { src = w.cap_dz.p; have = w.cap_dz.bytes / 4; }
    else if (k == "dx") { src = w.cap_dx.p; have = w.cap_dx.bytes / 4; }
    else return fail(TG_ERR_INVALID_ARG, std::string("tg_train_debug_read: unknown tensor ") + what);
    if (count > have) return fail(TG_ERR_INVALID_ARG, "tg_train_debug_read: count exceeds the buffer");
    TG_HIP(hipStreamSynchronize(e->stream));
    TG_HIP(hipMemcpy(out, src, count * 4, hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_train_commit(TgEngine* e) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    bool averaged = false;
    if (t->n_buffers) {  // data parallel: every rank normalised with its own batches → average the running statistics
        rc = all_reduce_sum(e, e->stream, t->bnbuf.as<float>(), t->n_buffers, "BatchNorm running statistics", &averaged);
        if (rc) return rc;
    }
    TG_HIP(hipStreamSynchronize(e->stream));
    std::vector<float> hp(t->n_params), hb(t->n_buffers);
    TG_HIP(hipMemcpy(hp.data(), t->params.p, t->n_params * 4, hipMemcpyDeviceToHost));
    TG_HIP(hipMemcpy(hb.data(), t->bnbuf.p, t->n_buffers * 4, hipMemcpyDeviceToHost));
    if (averaged) {
        for (float& v : hb) v /= (float)t->world;
        TG_HIP(hipMemcpy(t->bnbuf.p, hb.data(), t->n_buffers * 4, hipMemcpyHostToDevice));
    }
    for (const ParamInfo& pi : t->infos) {
        rc = net_set_tensor(e, pi.name.c_str(), (pi.buffer ? hb.data() : hp.data()) + pi.off, pi.count);
        if (rc) return rc;
    }
    return net_finalize(e);
}

int tg_train_set_allreduce(TgEngine* e, TgAllReduceFn fn, void* ctx, int world_size) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (world_size < 1) return fail(TG_ERR_INVALID_ARG, "tg_train_set_allreduce: world_size must be ≥ 1");
    Trainer* t = e->trainer;
    if (t->comm && fn) return fail(TG_ERR_STATE, "tg_train_set_allreduce: an RCCL communicator is already attached (tg_train_comm_init)");
    t->hook = fn;
    t->hook_ctx = ctx;
    if (fn) t->world = world_size;
    else if (!t->comm) t->world = 1;
    return TG_OK;
}

int tg_train_grad_buffer(TgEngine* e, float** d_grads, size_t* count) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (!d_grads || !count) return fail(TG_ERR_INVALID_ARG, "null argument");
    *d_grads = e->trainer->grads.as<float>();
    *count = e->trainer->n_params;
    return TG_OK;
}

int tg_train_comm_stats(TgEngine* e, double* ms_total, int64_t* reductions) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    TG_HIP(hipStreamSynchronize(e->stream));
    t->ar_fold(t->ar_next);  // the older pair first
    t->ar_fold(t->ar_next ^ 1);
    if (ms_total) *ms_total = t->ar_ms;
    if (reductions) *reductions = t->ar_count;
    return TG_OK;
}

int tg_train_comm_info(TgEngine* e, TgCommInfo* out) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (!out) return fail(TG_ERR_INVALID_ARG, "null argument");
    Trainer* t = e->trainer;
    std::memset(out, 0, sizeof(*out));
    out->attached = t->comm ? 1 : t->hook ? 2 : 0;
    out->world_size = t->world;
    out->rank = t->rank;
    out->nccl_count = out->nccl_rank = out->nccl_version = -1;
    if (g_rccl.lib) {
        out->lib_was_mapped = g_rccl.was_mapped ? 1 : 0;
        if (g_rccl.GetVersion) { int v = -1; if (g_rccl.GetVersion(&v) == 0) out->nccl_version = v; }
        Dl_info di;
        if (dladdr((void*)g_rccl.AllReduce, &di) && di.dli_fname) std::snprintf(out->lib_path, sizeof(out->lib_path), "%s", di.dli_fname);
    }
    if (t->comm) {
        int v = -1;
        if (g_rccl.CommCount && g_rccl.CommCount(t->comm, &v) == 0) out->nccl_count = v;
        v = -1;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(t->comm, &v) == 0) out->nccl_rank = v;
    }
    return TG_OK;
}

int tg_train_comm_preflight(TgEngine* e, double* ms) {
    int rc = need_trainer(e);
    if (rc) return rc;
    Trainer* t = e->trainer;
    if (ms) *ms = 0.0;
    if (!t->hook && !t->comm) return TG_OK;
    TG_HIP(t->err_flag.ensure(16));
    const float one[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    TG_HIP(hipStreamSynchronize(e->stream));
    const auto t0 = std::chrono::steady_clock::now();
    TG_HIP(hipMemcpyAsync(t->err_flag.p, one, 16, hipMemcpyHostToDevice, e->stream));
    bool reduced;
    rc = all_reduce_sum(e, e->stream, t->err_flag.as<float>(), 1, "preflight", &reduced);
    if (rc) return rc;
    TG_HIP(hipMemcpyAsync(sum, t->err_flag.p, 16, hipMemcpyDeviceToHost, e->stream));
    TG_HIP(hipStreamSynchronize(e->stream));
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (sum[0] != (float)t->world)
        return fail(TG_ERR_STATE, "tg_train_comm_preflight: the sum of one 1.0f per rank is " + std::to_string(sum[0]) + ", expected " +
                                      std::to_string(t->world) + " — the reduction does not span the ranks it was set up for");
    return TG_OK;
}

int tg_comm_unique_id(void* id128) {
    if (!id128) return fail(TG_ERR_INVALID_ARG, "null id");
    int rc = rccl_load();
    if (rc) return rc;
    int r = g_rccl.GetUniqueId(id128);
    if (r) return fail(TG_ERR_HIP, "ncclGetUniqueId failed");
    return TG_OK;
}

int tg_train_comm_init(TgEngine* e, int rank, int world_size, const void* id128) {
    int rc = need_trainer(e);
    if (rc) return rc;
    if (!id128 || world_size < 1 || rank < 0 || rank >= world_size) return fail(TG_ERR_INVALID_ARG, "tg_train_comm_init: bad arguments");
    rc = rccl_load();
    if (rc) return rc;
    Trainer* t = e->trainer;
    if (t->hook) return fail(TG_ERR_STATE, "tg_train_comm_init: a reduction hook is set (tg_train_set_allreduce)");
    if (t->comm) { g_rccl.CommDestroy(t->comm); t->comm = nullptr; }
    Id128 id;
    std::memcpy(id.bytes, id128, 128);
    int r = g_rccl.CommInitRank(&t->comm, world_size, id, rank);
    if (r) { t->comm = nullptr; return fail(TG_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); }
    t->world = world_size;
    t->rank = rank;
    return TG_OK;
}

}  // extern "C"
