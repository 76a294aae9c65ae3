// net.hip — host side of the network: tensor intake in tch/libtorch layout, BatchNorm folding,
// re-layout for the MFMA kernels, and the forward schedule (one kernel per conv layer, heads,
// softmax) on the engine stream.  Replaces Network<N> of reference alpha-tak/src/model/network.rs:26-35
// and the concrete Net5 / Net6 (model/net5.rs, net6.rs, res_block.rs) without any tch type.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "engine.h"
#include "kernels.h"
#include "softmax.cuh"
#include "rng.cuh"

namespace tg {

struct ConvLayer {
    DevBuf w, b;      // Wp [K/16][CoutP][16], bias [CoutP]
    int cin_pad = 0;  // channels per input row (multiple of 16)
    int cout = 0, cout_pad = 0;
};

struct Net {
    std::map<std::string, std::vector<float>> tensors;  // as given (tch layouts)
    bool ready = false;
    int F = 0, R = 0, cin = 0, cin_pad = 0;
    ConvLayer conv0;
    std::vector<ConvLayer> res1, res2;
    ConvLayer policy_conv;            // TG_HEAD_CONV
    DevBuf policy_w, policy_b;        // TG_HEAD_FC5: Wp [K/16][NP][16]
    DevBuf policy_w_lin;              // the same, every (chunk, tile) block of 16 columns in the MFMA fragment's lane order (k_fc_ring's LDS-DMA source)
    int policy_np = 0;                // padded FC outputs
    bool value_in_fc = false;         // the value head rides in padding column P of the policy FC (logit P = value pre-activation)
    DevBuf value_w;                   // [nsq*F] in NHWC order
    float value_b = 0.0f;
    // activations (max_batch positions)
    DevBuf x, y, logits, planes_nhwc, planes_nchw;
    DevBuf s3_fc_ring;                // the same weights in k_fc_s3_ring's layout [chunk of 32][tile 0 … 98][hi|lo][lane] (full batches)
    bool s3_fc_ring_on = false;
    DevBuf fc_stats;                  // [max_batch][fc_stat_blocks][2]: block-wise softmax statistics of the policy FC (softmax.cuh)
    bool fc_stats_on = false;
    int fc_stat_blocks = 0;           // FC_STAT_BLOCKS (11) on the exact path, s3_np / 112 on the split-bf16 FC
    int fc_stat_stride = 0;           // pairs per row of fc_stats: the exact path appends {value pre-activation, 0}
    FcGatherArgs gather{};            // set by the search (net_set_gather): logits-only forwards of the exact FC write the
    bool gather_on = false;           // children's logits of every leaf instead of the logits rows
    bool fused = false;  // whole tower in one launch (k_tower)
    TowerParams tower;
    ConvLayer conv0_tower;             // conv0 with the last input chunk permuted (layer 0 of the fused towers)
    ConvLayer conv0_board;             // conv0 over the board planes only (TowerParams.cb)
    DevBuf cplane_sums;                // S[constant plane][border class][F] (TowerParams.cb)
    int64_t conv_flops_exec = 0;       // MFMA FLOPs the timed launch actually issues (≤ conv_flops when constant planes are a bias)
    DevBuf halo_map;                   // tile slot → square table of the halo tower (k_tower_halo)
    DevBuf split_ctl;                  // k_tower_split: a counter line per position, then the error word a bounded wait raises (TOWER_SPLIT_CTL_WORDS)
    DevBuf split_buf;                  // k_tower_split: two exchange buffers of TOWER_SPLIT_MAX_BATCH positions × n² × F floats
    DevBuf s3_halo_map;                // the same for the split tower's workgroup (k_tower_s3_halo)
    size_t logit_row = 0;              // floats per position in `logits`
    int precision = TG_PRECISION_F32;  // tg_net_set_precision
    bool s3 = false;                   // split-bf16 tower in use
    TowerS3Params tower_s3;
    std::vector<DevBuf> s3_w;
    DevBuf s3_w0_board;    // conv0 over the board planes only, split fragments with one 32-channel chunk (TowerS3Params.cb)
    DevBuf s3_head;        // conv policy head weights, split (computed inside k_tower_s3)
    bool s3_head_on = false;
    DevBuf s3_fc, s3_fc_b; // policy FC weights (split) and bias padded to s3_np
    int s3_np = 0;         // padded outputs of the split FC
    bool s3_fc_on = false; // FC head on the split path (tower writes split activations)
    // measurement hooks (tg_profile_*)
    int prof_every = 0;
    uint64_t prof_counter = 0;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::vector<hipEvent_t>> ev_chains;  // per sampled forward: [start, after conv0, after each tower conv…, end]
    double conv_ms = 0.0, fwd_ms = 0.0;
    uint64_t conv_n = 0, fwd_n = 0;
    int64_t conv_rows = 0, conv_flops = 0;
    ~Net() {
        for (auto& c : ev_chains) for (auto ev : c) (void)hipEventDestroy(ev);
        for (auto ev : ev_pool) (void)hipEventDestroy(ev);
    }
};

static hipEvent_t prof_event(Net* n, hipStream_t st) {
    hipEvent_t ev = nullptr;
    if (!n->ev_pool.empty()) { ev = n->ev_pool.back(); n->ev_pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    (void)hipEventRecord(ev, st);
    return ev;
}

static void prof_collect(Net* n) {  // the stream must be idle
    for (auto& c : n->ev_chains) {
        float ms = 0.0f;
        if (c.size() >= 2 && hipEventElapsedTime(&ms, c.front(), c.back()) == hipSuccess) { n->fwd_ms += ms; n->fwd_n++; }
        for (size_t i = 1; i + 2 < c.size(); i++)  // intervals [after conv0 … after last tower conv]
            if (hipEventElapsedTime(&ms, c[i], c[i + 1]) == hipSuccess) { n->conv_ms += ms; n->conv_n++; }
        for (auto ev : c) n->ev_pool.push_back(ev);
    }
    n->ev_chains.clear();
}

void net_destroy(Net* n) { delete n; }

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

int net_create(TgEngine* e) {
    Net* n = new Net();
    e->net = n;
    n->F = e->cfg.filters;
    n->R = e->cfg.res_blocks;
    n->cin = e->cin;
    n->cin_pad = e->cin_pad;
    return TG_OK;
}

int net_set_tensor(TgEngine* e, const char* name, const float* data, size_t count) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    if (!name || (!data && count)) return fail(TG_ERR_INVALID_ARG, "tg_net_set_tensor: null argument");
    e->net->tensors[name] = std::vector<float>(data, data + count);
    e->net->ready = false;
    return TG_OK;
}

bool net_ready(const TgEngine* e) { return e && e->net && e->net->ready; }
const std::map<std::string, std::vector<float>>* net_tensors(const TgEngine* e) { return e && e->net ? &e->net->tensors : nullptr; }

namespace {

struct Folded {
    std::vector<float> w, b;  // OIHW weights scaled per output channel, bias
};

const std::vector<float>* find(Net* n, const std::string& name, size_t want, std::string& err) {
    auto it = n->tensors.find(name);
    if (it == n->tensors.end()) { err = "missing tensor " + name; return nullptr; }
    if (it->second.size() != want) {
        err = "tensor " + name + " has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(want);
        return nullptr;
    }
    return &it->second;
}

// conv (O,I,3,3)+bias followed by BatchNorm in eval mode (running stats, eps = 1e-5, tch default):
// y = (conv(x) + b - mean) * gamma / sqrt(var + eps) + beta  →  w' = w*s, b' = (b - mean)*s + beta
bool fold_conv_bn(Net* n, const std::string& conv, const std::string& bn, int O, int I, Folded& out, std::string& err) {
    auto w = find(n, conv + ".weight", (size_t)O * I * 9, err);
    auto b = w ? find(n, conv + ".bias", O, err) : nullptr;
    if (!w || !b) return false;
    out.w = *w;
    out.b = *b;
    if (bn.empty()) return true;
    auto g = find(n, bn + ".weight", O, err);
    auto be = g ? find(n, bn + ".bias", O, err) : nullptr;
    auto mu = be ? find(n, bn + ".running_mean", O, err) : nullptr;
    auto var = mu ? find(n, bn + ".running_var", O, err) : nullptr;
    if (!var) return false;
    for (int o = 0; o < O; o++) {
        float s = (*g)[o] / std::sqrt((*var)[o] + 1e-5f);
        for (int k = 0; k < I * 9; k++) out.w[(size_t)o * I * 9 + k] *= s;
        out.b[o] = ((*b)[o] - (*mu)[o]) * s + (*be)[o];
    }
    return true;
}

// OIHW → Wp[(tap*Ipad + c)/16][CoutP][(tap*Ipad + c)%16], tap = ky*3 + kx, zero padded
// last_t = 2 / 3: the real channels of the last 16-channel chunk go to positions 4·(i / last_t) + i % last_t
// (conv_mainloop.cuh conv_last_chunk_perm: layer 0 of the fused towers); 4 = channel order
hipError_t upload_conv(const Folded& f, int O, int I, int Ipad, ConvLayer& L, int last_t = 4) {
    int OP = round_up(O, 64);
    size_t K = (size_t)9 * Ipad;
    std::vector<float> wp(K * OP, 0.0f), bp(OP, 0.0f);
    for (int o = 0; o < O; o++) {
        bp[o] = f.b[o];
        for (int c = 0; c < I; c++)
            for (int tap = 0; tap < 9; tap++) {
                int pos = c & 15;
                if (last_t < 4 && c >= Ipad - 16) { const int i = c - (Ipad - 16); pos = 4 * (i / last_t) + i % last_t; }
                size_t k = (size_t)tap * Ipad + (c & ~15) + pos;
                wp[((k >> 4) * OP + o) * 16 + (k & 15)] = f.w[((size_t)o * I + c) * 9 + tap];
            }
    }
    L.cin_pad = Ipad; L.cout = O; L.cout_pad = OP;
    hipError_t e = L.w.ensure(wp.size() * 4);
    if (e != hipSuccess) return e;
    e = L.b.ensure(bp.size() * 4);
    if (e != hipSuccess) return e;
    e = hipMemcpy(L.w.p, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    return hipMemcpy(L.b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice);
}

uint16_t f32_to_bf16(float x) {  // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
}

// OIHW (BN folded) → [chunk = tap·KC + kc][cout tile][hi|lo][q][cout][8 bf16], channel = 32·kc + 8q + j, zero padded
hipError_t upload_conv_s3(const Folded& f, int O, int I, int KC, DevBuf& buf, int OP = 0) {
    if (!OP) OP = O;  // output channels of the fragment layout (multiple of 16), O of them real
    std::vector<uint16_t> w((size_t)9 * KC * OP * 64, 0);
    for (int tap = 0; tap < 9; tap++)
        for (int kc = 0; kc < KC; kc++)
            for (int o = 0; o < O; o++)
                for (int q = 0; q < 4; q++)
                    for (int j = 0; j < 8; j++) {
                        int c = 32 * kc + 8 * q + j;
                        if (c >= I) continue;
                        float v = f.w[((size_t)o * I + c) * 9 + tap];
                        uint16_t hi = f32_to_bf16(v);
                        uint16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
                        // [chunk][tile of 16 couts][hi|lo][q][cout in tile][8 bf16].  A wave computes two tiles = 32 output channels;
                        // the MFMA leaves rows 4q' … 4q' + 3 of a tile in lane group q', so channel 8q' + 4t + i of the 32 goes to
                        // row 4q' + i of tile t: lane group q' then holds channels 8q' … 8q' + 7 — one 16-byte slot of the split image
                        const int o32 = o & 31, tile = (o >> 5) * 2 + ((o32 >> 2) & 1), row = (o32 >> 3) * 4 + (o32 & 3);
                        size_t slot = (((((size_t)tap * KC + kc) * (OP / 16) + tile) * 2) * 4 + q) * 16 + row;
                        w[slot * 8 + j] = hi;
                        w[(slot + 64) * 8 + j] = lo;
                    }
    hipError_t e = buf.ensure(w.size() * 2);
    if (e != hipSuccess) return e;
    return hipMemcpy(buf.p, w.data(), w.size() * 2, hipMemcpyHostToDevice);
}

}  // namespace

int net_finalize(TgEngine* e) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    TG_HIP(hipSetDevice(e->cfg.device));
    Net* n = e->net;
    const int F = n->F, R = n->R, nsq = e->g.nsq;
    std::string err;
    Folded f;
    if (!fold_conv_bn(n, "conv0", "bn0", F, n->cin, f, err)) return fail(TG_ERR_WEIGHTS, err);
    TG_HIP(upload_conv(f, F, n->cin, n->cin_pad, n->conv0));
    n->res1.clear(); n->res2.clear();
    n->res1.resize(R); n->res2.resize(R);
    for (int i = 0; i < R; i++) {
        std::string p = "res" + std::to_string(i);
        if (!fold_conv_bn(n, p + ".conv1", p + ".bn1", F, F, f, err)) return fail(TG_ERR_WEIGHTS, err);
        TG_HIP(upload_conv(f, F, F, F, n->res1[i]));
        if (!fold_conv_bn(n, p + ".conv2", p + ".bn2", F, F, f, err)) return fail(TG_ERR_WEIGHTS, err);
        TG_HIP(upload_conv(f, F, F, F, n->res2[i]));
    }
    const int P = e->policy_size;
    if (e->cfg.policy_head == TG_HEAD_CONV) {
        int ch = P / nsq;
        if (!fold_conv_bn(n, "policy", "", ch, F, f, err)) return fail(TG_ERR_WEIGHTS, err);
        TG_HIP(upload_conv(f, ch, F, F, n->policy_conv));
    } else {
        // Linear [P, F*nsq] over the NCHW flattening c*nsq + sq (net5.rs:86 view) → K order sq*F + c
        size_t K = (size_t)F * nsq;
        auto w = find(n, "policy.weight", (size_t)P * K, err);
        auto b = w ? find(n, "policy.bias", P, err) : nullptr;
        if (!b) return fail(TG_ERR_WEIGHTS, err);
        // padded to 1664 columns (the FC kernels use the first 99 tiles = 1584 of them, softmax.cuh; the generic k_gemm and
        // k_fc_small want multiples of 64 / 32); shapes with K % 64 != 0 go through k_gemm
        int NP = (K % 64 == 0) ? round_up(P, 208) : round_up(P, 64);
        std::vector<float> wp(K * NP, 0.0f), bp(NP, 0.0f);
        for (int o = 0; o < P; o++) {
            bp[o] = (*b)[o];
            for (int c = 0; c < F; c++)
                for (int sq = 0; sq < nsq; sq++) {
                    size_t k = (size_t)sq * F + c;
                    wp[((k >> 4) * NP + o) * 16 + (k & 15)] = (*w)[(size_t)o * K + (size_t)c * nsq + sq];
                }
        }
        // the FC is padded anyway: column P carries the value head (Linear(F·nsq → 1), net5.rs:62), so the value costs no
        // kernel of its own — the softmax kernel applies the tanh
        n->value_in_fc = false;
        {
            auto wv = find(n, "value.weight", K, err);
            auto bv = wv ? find(n, "value.bias", 1, err) : nullptr;
            if (!bv) return fail(TG_ERR_WEIGHTS, err);
            if (NP > P && !env_on("TG_SEPARATE_VALUE_HEAD")) {
                bp[P] = (*bv)[0];
                for (int c = 0; c < F; c++)
                    for (int sq = 0; sq < nsq; sq++) {
                        size_t k = (size_t)sq * F + c;
                        wp[((k >> 4) * NP + P) * 16 + (k & 15)] = (*wv)[(size_t)c * nsq + sq];
                    }
                n->value_in_fc = true;
            }
        }
        n->policy_np = NP;
        TG_HIP(n->policy_w.ensure(wp.size() * 4));
        TG_HIP(n->policy_b.ensure(bp.size() * 4));
        TG_HIP(hipMemcpy(n->policy_w.p, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
        {   // block (chunk, tile): slot q·16 + r holds the 4 floats (k = 16·chunk + 4q … 4q + 3) of column 16·tile + r
            std::vector<float> wl(wp.size());
            const size_t tiles = (size_t)NP / 16;
            for (size_t c = 0; c < K / 16; c++)
                for (size_t t = 0; t < tiles; t++)
                    for (size_t r = 0; r < 16; r++)
                        for (size_t qq = 0; qq < 4; qq++)
                            for (size_t u = 0; u < 4; u++)
                                wl[((c * tiles + t) * 64 + qq * 16 + r) * 4 + u] = wp[((c * NP + t * 16 + r) * 4 + qq) * 4 + u];
            TG_HIP(n->policy_w_lin.ensure(wl.size() * 4));
            TG_HIP(hipMemcpy(n->policy_w_lin.p, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
        }
        TG_HIP(hipMemcpy(n->policy_b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    }
    {
        size_t K = (size_t)F * nsq;
        auto w = find(n, "value.weight", K, err);
        auto b = w ? find(n, "value.bias", 1, err) : nullptr;
        if (!b) return fail(TG_ERR_WEIGHTS, err);
        std::vector<float> wv(K);
        for (int c = 0; c < F; c++)
            for (int sq = 0; sq < nsq; sq++) wv[(size_t)sq * F + c] = (*w)[(size_t)c * nsq + sq];
        n->value_b = (*b)[0];
        TG_HIP(n->value_w.ensure(K * 4));
        TG_HIP(hipMemcpy(n->value_w.p, wv.data(), K * 4, hipMemcpyHostToDevice));
    }
    n->fused = tower_supported(e->g.n, F, n->cin_pad) && 1 + 2 * R <= 48 && !env_on("TG_NO_FUSED_TOWER");
    if (n->fused) {
        TowerParams& T = n->tower;
        T.nlayers = 1 + 2 * R; T.cin_pad = n->cin_pad; T.F = F;
        // layer 0: 72 of 80 (5×5) / 92 of 96 (6×6) input channels are real; with the last chunk permuted 2 of 20 / 1 of 24
        // MFMAs per tile and tap multiply padding only and are skipped — the towers get their own copy of the weights
        const int tail = n->cin - (n->cin_pad - 16);
        T.cin_last_t = (tail > 0 && tail <= 12 && !env_on("TG_NO_CIN_PERM")) ? (tail + 3) / 4 : 4;
        if (T.cin_last_t < 2) T.cin_last_t = 4;
        {
            Folded f0;
            if (!fold_conv_bn(n, "conv0", "bn0", F, n->cin, f0, err)) return fail(TG_ERR_WEIGHTS, err);
            TG_HIP(upload_conv(f0, F, n->cin, n->cin_pad, n->conv0_tower, T.cin_last_t));
        }
        T.w[0] = n->conv0_tower.w.as<float>(); T.b[0] = n->conv0.b.as<float>();
        // Constant planes as a bias (kernels.h TowerParams.cb; states entry): conv0 over the board planes, and per constant
        // plane and border class the sum of its folded weights over the taps that stay on the board (tap order, f32)
        T.cb = 0; T.cb_cin_pad = 32; T.cb_last_t = 3; T.w0_board = nullptr; T.cplane_sums = nullptr;
        {
            const int N = e->g.n, bc = board_channels(N), nconst = n->cin - bc;
            if (!env_on("TG_NO_CONST_BIAS") && bc > 16 && bc <= 28 && nconst > 0) {
                Folded f0, fb;
                if (!fold_conv_bn(n, "conv0", "bn0", F, n->cin, f0, err)) return fail(TG_ERR_WEIGHTS, err);
                fb.b = f0.b;
                fb.w.resize((size_t)F * bc * 9);
                for (int o = 0; o < F; o++)
                    for (int c = 0; c < bc; c++)
                        for (int tap = 0; tap < 9; tap++) fb.w[((size_t)o * bc + c) * 9 + tap] = f0.w[((size_t)o * n->cin + c) * 9 + tap];
                TG_HIP(upload_conv(fb, F, bc, T.cb_cin_pad, n->conv0_board, T.cb_last_t));
                std::vector<float> S((size_t)nconst * 9 * F, 0.0f);
                for (int pl = 0; pl < nconst; pl++)
                    for (int cls = 0; cls < 9; cls++) {
                        const int cy = cls / 3, cx = cls % 3;  // 0: first row / column, 1: interior, 2: last
                        for (int o = 0; o < F; o++) {
                            float sum = 0.0f;
                            for (int tap = 0; tap < 9; tap++) {
                                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                                if ((cy == 0 && dy < 0) || (cy == 2 && dy > 0) || (cx == 0 && dx < 0) || (cx == 2 && dx > 0)) continue;
                                sum += f0.w[((size_t)o * n->cin + bc + pl) * 9 + tap];
                            }
                            S[((size_t)pl * 9 + cls) * F + o] = sum;
                        }
                    }
                TG_HIP(n->cplane_sums.ensure(S.size() * 4));
                TG_HIP(hipMemcpy(n->cplane_sums.p, S.data(), S.size() * 4, hipMemcpyHostToDevice));
                T.cb = 1; T.w0_board = n->conv0_board.w.as<float>(); T.cplane_sums = n->cplane_sums.as<float>();
            }
        }
        for (int i = 0; i < R; i++) {
            T.w[1 + 2 * i] = n->res1[i].w.as<float>(); T.b[1 + 2 * i] = n->res1[i].b.as<float>();
            T.w[2 + 2 * i] = n->res2[i].w.as<float>(); T.b[2 + 2 * i] = n->res2[i].b.as<float>();
        }
        T.slotmap = nullptr; T.halo_pw = 0; T.halo_ps = 0;
        T.split_flags = nullptr; T.split_err = nullptr;
        if (T.cb && (F == 128 || (F == 64 && e->g.n == 5))) {  // small batches split a position over several workgroups (k_tower_split)
            TG_HIP(n->split_ctl.ensure((size_t)TOWER_SPLIT_CTL_WORDS * 4));
            TG_HIP(hipMemset(n->split_ctl.p, 0, (size_t)TOWER_SPLIT_CTL_WORDS * 4));
            TG_HIP(n->split_buf.ensure((size_t)2 * TOWER_SPLIT_MAX_BATCH * nsq * F * 4));
            T.split_flags = n->split_ctl.as<unsigned>();
            T.split_err = n->split_ctl.as<int>() + TOWER_SPLIT_CTL_WORDS - 32;
        }
        // FC-head networks whose value head rides in the FC's padding column: nothing but the policy FC reads the tower's
        // output, so it is written in the FC's fragment order
        T.frag_out = (e->cfg.policy_head == TG_HEAD_FC5 && n->value_in_fc && fc_frag_supported(nsq * F, n->policy_np) &&
                      !env_on("TG_NO_FRAG_OUT")) ? 1 : 0;
        int pw, ps;
        if (tower_halo_geometry(e->g.n, F, &pw, &ps)) {
            std::vector<uint32_t> map((size_t)((pw * nsq + 15) / 16) * 16);
            tower_halo_slotmap(e->g.n, pw, ps, map.data());
            TG_HIP(n->halo_map.ensure(map.size() * 4));
            TG_HIP(hipMemcpy(n->halo_map.p, map.data(), map.size() * 4, hipMemcpyHostToDevice));
            T.slotmap = n->halo_map.as<uint32_t>(); T.halo_pw = pw; T.halo_ps = ps;
        }
    }
    n->s3 = false;
    if (n->precision == TG_PRECISION_BF16X3) {
        if (!tower_s3_supported(e->g.n, F) || 1 + 2 * R > 48 || n->cin > 96)
            return fail(TG_ERR_INVALID_ARG, "TG_PRECISION_BF16X3 supports 5x5 with 64 or 128 filters and 6x6 with 128 filters");
        TowerS3Params& T = n->tower_s3;
        T.nlayers = 1 + 2 * R; T.cin_pad = n->cin_pad; T.F = F;
        n->s3_w.clear();
        n->s3_w.resize(T.nlayers);
        Folded g;
        if (!fold_conv_bn(n, "conv0", "bn0", F, n->cin, g, err)) return fail(TG_ERR_WEIGHTS, err);
        TG_HIP(upload_conv_s3(g, F, n->cin, 3, n->s3_w[0]));
        T.w[0] = n->s3_w[0].p; T.b[0] = n->conv0.b.as<float>();
        for (int i = 0; i < R; i++) {
            std::string p = "res" + std::to_string(i);
            if (!fold_conv_bn(n, p + ".conv1", p + ".bn1", F, F, g, err)) return fail(TG_ERR_WEIGHTS, err);
            TG_HIP(upload_conv_s3(g, F, F, F / 32, n->s3_w[1 + 2 * i]));
            if (!fold_conv_bn(n, p + ".conv2", p + ".bn2", F, F, g, err)) return fail(TG_ERR_WEIGHTS, err);
            TG_HIP(upload_conv_s3(g, F, F, F / 32, n->s3_w[2 + 2 * i]));
            T.w[1 + 2 * i] = n->s3_w[1 + 2 * i].p; T.b[1 + 2 * i] = n->res1[i].b.as<float>();
            T.w[2 + 2 * i] = n->s3_w[2 + 2 * i].p; T.b[2 + 2 * i] = n->res2[i].b.as<float>();
        }
        // constant planes as a bias on the split towers too: the f32 table S of the exact path, conv0 restricted to the board planes
        T.cb = 0; T.w0_board = nullptr; T.cplane_sums = nullptr;
        if (n->fused && n->tower.cb) {
            const int bc = board_channels(e->g.n);
            Folded fb, g0;
            if (!fold_conv_bn(n, "conv0", "bn0", F, n->cin, g0, err)) return fail(TG_ERR_WEIGHTS, err);
            fb.b = g0.b;
            fb.w.resize((size_t)F * bc * 9);
            for (int o = 0; o < F; o++)
                for (int c = 0; c < bc; c++)
                    for (int tap = 0; tap < 9; tap++) fb.w[((size_t)o * bc + c) * 9 + tap] = g0.w[((size_t)o * n->cin + c) * 9 + tap];
            TG_HIP(upload_conv_s3(fb, F, bc, 1, n->s3_w0_board));
            T.cb = 1; T.w0_board = n->s3_w0_board.p; T.cplane_sums = n->cplane_sums.as<float>();
        }
        T.slotmap = nullptr; T.halo_ps = 0;
        int pw3, ps3;
        if (tower_s3_halo_geometry(e->g.n, F, &pw3, &ps3)) {
            std::vector<uint32_t> map((size_t)((pw3 * nsq + 15) / 16) * 16);
            tower_halo_slotmap(e->g.n, pw3, ps3, map.data());
            TG_HIP(n->s3_halo_map.ensure(map.size() * 4));
            TG_HIP(hipMemcpy(n->s3_halo_map.p, map.data(), map.size() * 4, hipMemcpyHostToDevice));
            T.slotmap = n->s3_halo_map.as<uint32_t>(); T.halo_ps = ps3;
        }
        n->s3 = true;
        n->s3_fc_on = false;
        T.head_w = nullptr; T.head_b = nullptr; T.head_out = nullptr; T.head_cout = 0;
        n->s3_head_on = false;
        if (e->cfg.policy_head == TG_HEAD_CONV && n->policy_conv.cout_pad % 32 == 0 && !env_on("TG_S3_NO_HEAD")) {
            const int ch = P / nsq;
            if (!fold_conv_bn(n, "policy", "", ch, F, g, err)) return fail(TG_ERR_WEIGHTS, err);
            TG_HIP(upload_conv_s3(g, ch, F, F / 32, n->s3_head, n->policy_conv.cout_pad));
            T.head_w = n->s3_head.p; T.head_b = n->policy_conv.b.as<float>(); T.head_cout = n->policy_conv.cout_pad;
            n->s3_head_on = true;
        }
        const int s3np = round_up(P, 112);  // column blocks of 112 outputs (k_fc_s3b); TG_S3_FC_WIDE=1 keeps the 208-wide kernel
        n->s3_np = env_on("TG_S3_FC_WIDE") ? n->policy_np : s3np;
        if (e->cfg.policy_head == TG_HEAD_FC5 && fc_s3_supported(F * nsq, n->s3_np)) {
            // Linear [P, F·nsq] → split bf16 fragments, k = sq·F + c (the order of the activations)
            const size_t K = (size_t)F * nsq;
            const int NP = n->s3_np, CB = fc_s3_cols(NP);
            auto w = find(n, "policy.weight", (size_t)P * K, err);
            auto bsrc = w ? find(n, "policy.bias", P, err) : nullptr;
            if (!bsrc) return fail(TG_ERR_WEIGHTS, err);
            std::vector<uint16_t> ws((size_t)(K / 32) * NP * 64, 0);
            for (int o = 0; o < P; o++)
                for (size_t k = 0; k < K; k++) {
                    int sq = (int)(k / F), c = (int)(k % F);
                    float v = (*w)[(size_t)o * K + (size_t)c * nsq + sq];
                    uint16_t hi = f32_to_bf16(v), lo = f32_to_bf16(v - bf16_to_f32(hi));
                    // [chunk][column block][q][hi|lo][column][8 bf16]: the LDS plane layout of k_fc_s3 / k_fc_s3b
                    const size_t cb = (size_t)o / CB, col = (size_t)o % CB, q = (k & 31) >> 3;
                    size_t slot = ((((k >> 5) * (size_t)(NP / CB) + cb) * 4 + q) * 2) * CB + col;
                    ws[slot * 8 + (k & 7)] = hi;
                    ws[(slot + CB) * 8 + (k & 7)] = lo;
                }
            if (n->value_in_fc && NP > P) {  // column P = the value head, split like every other column
                auto wv = find(n, "value.weight", K, err);
                if (!wv) return fail(TG_ERR_WEIGHTS, err);
                for (size_t k = 0; k < K; k++) {
                    int sq = (int)(k / F), c = (int)(k % F);
                    float v = (*wv)[(size_t)c * nsq + sq];
                    uint16_t hi = f32_to_bf16(v), lo = f32_to_bf16(v - bf16_to_f32(hi));
                    const size_t cb = (size_t)P / CB, col = (size_t)P % CB, q = (k & 31) >> 3;
                    size_t slot = ((((k >> 5) * (size_t)(NP / CB) + cb) * 4 + q) * 2) * CB + col;
                    ws[slot * 8 + (k & 7)] = hi;
                    ws[(slot + CB) * 8 + (k & 7)] = lo;
                }
            }
            TG_HIP(n->s3_fc.ensure(ws.size() * 2));
            TG_HIP(hipMemcpy(n->s3_fc.p, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
            // the ring kernel's copy: [chunk][tile][hi|lo][lane = q·16 + column within the tile][8 bf16]: a (chunk, tile, half) block
            // is one KB in the reader's lane order = one LDS-DMA instruction
            n->s3_fc_ring_on = false;
            if (P + 1 <= FC_TILES * 16 && K % 64 == 0) {
                std::vector<uint16_t> wr((size_t)(K / 32) * FC_TILES * 2 * 64 * 8, 0);
                auto put = [&](size_t o, size_t k, float v) {
                    uint16_t hi = f32_to_bf16(v), lo = f32_to_bf16(v - bf16_to_f32(hi));
                    const size_t lane = ((k & 31) >> 3) * 16 + (o & 15);
                    const size_t slot = (((k >> 5) * FC_TILES + (o >> 4)) * 2) * 64 + lane;
                    wr[slot * 8 + (k & 7)] = hi;
                    wr[(slot + 64) * 8 + (k & 7)] = lo;
                };
                for (int o = 0; o < P; o++)
                    for (size_t k = 0; k < K; k++) put((size_t)o, k, (*w)[(size_t)o * K + (size_t)(k % F) * nsq + k / F]);
                if (n->value_in_fc && NP > P) {
                    auto wv = find(n, "value.weight", K, err);
                    if (!wv) return fail(TG_ERR_WEIGHTS, err);
                    for (size_t k = 0; k < K; k++) put((size_t)P, k, (*wv)[(size_t)(k % F) * nsq + k / F]);
                }
                TG_HIP(n->s3_fc_ring.ensure(wr.size() * 2));
                TG_HIP(hipMemcpy(n->s3_fc_ring.p, wr.data(), wr.size() * 2, hipMemcpyHostToDevice));
                n->s3_fc_ring_on = true;
            }
            std::vector<float> bp(NP, 0.0f);
            std::copy(bsrc->begin(), bsrc->end(), bp.begin());
            if (n->value_in_fc && NP > P) bp[P] = n->value_b;
            else n->value_in_fc = false;
            TG_HIP(n->s3_fc_b.ensure(bp.size() * 4));
            TG_HIP(hipMemcpy(n->s3_fc_b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
            n->s3_fc_on = true;
        }
    }
    size_t mb = (size_t)e->cfg.max_batch;
    TG_HIP(n->x.ensure(((mb + 15) / 16 * 16) * nsq * F * 4));  // (whole tiles of 16 positions: fragment-major FC input)
    TG_HIP(n->y.ensure(mb * nsq * F * 4));
    size_t logit_row = e->cfg.policy_head == TG_HEAD_CONV ? (size_t)nsq * n->policy_conv.cout_pad : (size_t)std::max(n->policy_np, n->s3_np);
    n->logit_row = logit_row;
    TG_HIP(n->logits.ensure(mb * logit_row * 4));
    TG_HIP(n->planes_nhwc.ensure(mb * nsq * n->cin_pad * 4));
    // f32 FC head with the value column: the FC emits the softmax statistics per column block, nobody re-reads whole rows
    const bool s3fc = n->s3 && n->s3_fc_on;
    // (one statistics geometry — softmax.cuh — on both precisions since round 4: the split-bf16 FC's ring kernel emits it from its
    // epilogue, k_fc_stats computes it behind k_fc_s3b for ≤ 512 rows)
    n->fc_stats_on = e->cfg.policy_head == TG_HEAD_FC5 && n->value_in_fc && !env_on("TG_NO_FC_STATS") && e->policy_size + 1 <= FC_TILES * 16 &&
                     (s3fc ? n->s3_np >= FC_TILES * 16 : fc_stats_supported(nsq * F, n->policy_np, n->policy_np));
    n->fc_stat_blocks = n->fc_stats_on ? FC_STAT_BLOCKS : 0;
    n->fc_stat_stride = n->fc_stats_on ? FC_STAT_STRIDE : 0;
    if (n->fc_stats_on) TG_HIP(n->fc_stats.ensure(mb * (size_t)n->fc_stat_stride * 2 * 4));
    n->ready = true;
    return TG_OK;
}

static int net_forward_impl(TgEngine* e, int nb, const float* d_planes, const uint8_t* d_states, float* d_policy, float* d_eval,
                            hipStream_t st = nullptr, int pos0 = 0);

// planes NHWC [nb][nsq][cin_pad] (device) → policy [nb][P] (softmax, reference order), eval [nb]
int net_forward_dev(TgEngine* e, int nb, const float* d_planes, float* d_policy, float* d_eval) {
    return net_forward_impl(e, nb, d_planes, nullptr, d_policy, d_eval);
}

bool net_takes_states(const TgEngine* e) { return net_ready(e) && (e->net->fused || e->net->s3); }

// FC head with the value column: the logits buffer ([max_batch][*ld], logit P = value pre-activation) the search reads when it
// passes d_policy = nullptr to the forward (no softmax kernel, no probabilities in HBM); nullptr for every other topology
const float* net_fc_logits(const TgEngine* e, int* ld) {
    if (!net_ready(e)) return nullptr;
    const Net* n = e->net;
    if (e->cfg.policy_head != TG_HEAD_FC5 || !n->value_in_fc || e->policy_size > 2048) return nullptr;
    if (n->s3 && !n->s3_fc_on) return nullptr;  // split tower with the f32 FC: keep the plain path
    *ld = n->s3 && n->s3_fc_on ? n->s3_np : n->policy_np;  // the row stride the FC kernel in use writes
    return n->logits.as<float>();
}

// the block statistics that go with net_fc_logits' buffer ([max_batch][*blocks][2]), or nullptr: the consumer then takes max and
// Σexp over the whole row itself (softmax_stats_wave)
const float* net_fc_stats(const TgEngine* e, int* blocks, int* stride) {
    if (!net_ready(e) || !e->net->fc_stats_on) return nullptr;
    if (e->net->s3 && !e->net->s3_fc_on) return nullptr;  // split tower with the f32 FC: net_fc_logits declines too
    *blocks = e->net->fc_stat_blocks;
    *stride = e->net->fc_stat_stride;
    return e->net->fc_stats.as<float>();
}

// Search iterations on the exact-f32 FC head need, of the FC's output, only the logits of every leaf's children: with a gather
// target set (and a batch the ring kernel serves), a logits-only forward (d_policy = nullptr) of `leaves` rows writes
// child_logit[row][child] and the statistics record (with the value pre-activation) — no logits rows.  net_gather_ok tells
// the search whether the next such forward will do so (then its backup reads child_logit, else the logits buffer).
bool net_gather_ok(const TgEngine* e, int leaves) {
    if (!net_ready(e)) return false;
    const Net* n = e->net;
    static const bool off = env_on("TG_NO_FC_GATHER");  // A/B: logits rows + the backup's own gather (same bits)
    if (off || e->cfg.policy_head != TG_HEAD_FC5 || !n->fc_stats_on || !n->value_in_fc) return false;
    if (n->s3) return n->s3_fc_on && n->s3_fc_ring_on && !env_on("TG_S3_NO_FC_RING") && fc_s3_ring_supported(leaves, e->g.nsq * n->F, e->policy_size + 1);
    return fc_gather_supported(leaves, e->g.nsq * n->F, n->policy_np);
}
void net_set_gather(TgEngine* e, const FcGatherArgs* g) {
    if (!e || !e->net) return;
    e->net->gather_on = g != nullptr;
    if (g) e->net->gather = *g;
}

// half batch on its own stream (only when the tower encodes from states); see search.hip
int net_forward_states_at(TgEngine* e, int nb, const uint8_t* d_states, float* d_policy, float* d_eval, hipStream_t st, int pos0) {
    if (!net_takes_states(e)) return fail(TG_ERR_STATE, "net_forward_states_at needs the fused tower");
    return net_forward_impl(e, nb, nullptr, d_states, d_policy, d_eval, st, pos0);
}
// true when the next forward will be sampled by the profiler (it must then run alone on the GPU to be timed)
bool net_profile_due(const TgEngine* e) {
    const Net* n = e->net;
    return n && n->prof_every > 0 && (n->prof_counter % (uint64_t)n->prof_every) == 0;
}
// a forward that was not sampled still advances the sampling counter
void net_profile_skip(TgEngine* e) { if (e->net && e->net->prof_every > 0) e->net->prof_counter++; }

int net_forward_states_dev(TgEngine* e, int nb, const uint8_t* d_states, float* d_policy, float* d_eval) {
    if (!net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    if (nb <= 0) return TG_OK;
    if (e->net->fused || e->net->s3) return net_forward_impl(e, nb, nullptr, d_states, d_policy, d_eval);
    launch_encode_nhwc(e->stream, d_states, nb, e->g.n, e->net->planes_nhwc.as<float>(), e->net->cin_pad);
    TG_HIP(hipGetLastError());
    return net_forward_impl(e, nb, e->net->planes_nhwc.as<float>(), nullptr, d_policy, d_eval);
}

// st / pos0: the stream to launch on (default: the engine stream) and the first position slot of the activation buffers
// to use — two half batches on two streams work on disjoint slices of the same buffers (search.hip, dual-stream rollouts)
static int net_forward_impl(TgEngine* e, int nb, const float* d_planes, const uint8_t* d_states, float* d_policy, float* d_eval,
                            hipStream_t st, int pos0) {
    if (!net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    if (nb <= 0) return TG_OK;
    if (pos0 < 0 || pos0 + nb > e->cfg.max_batch) return fail(TG_ERR_INVALID_ARG, "batch larger than max_batch");
    Net* n = e->net;
    if (!st) st = e->stream;
    const int F = n->F, nsq = e->g.nsq, N = e->g.n;
    const int M = nb * nsq;
    float* x = n->x.as<float>() + (size_t)pos0 * nsq * F;
    float* y = n->y.as<float>() + (size_t)pos0 * nsq * F;
    std::vector<hipEvent_t>* chain = nullptr;
    if (st == e->stream && n->prof_every > 0 && (n->prof_counter++ % (uint64_t)n->prof_every) == 0) {
        if (n->ev_chains.size() >= 256) {  // bound the number of pending events
            TG_HIP(hipStreamSynchronize(st));
            prof_collect(n);
        }
        n->ev_chains.emplace_back();
        chain = &n->ev_chains.back();
        n->conv_rows = M;
        // algorithmic FLOPs of one timed launch: one F→F conv (per-layer path) or the whole tower (fused)
        n->conv_flops = (n->fused || n->s3) ? 2ll * M * 9 * ((long long)n->cin * F + 2ll * n->R * F * F) : 2ll * M * 9 * F * F;
        // executed: with the constant planes as a bias layer 0 multiplies 16 + 4·cb_last_t input channels per tap, not cin
        n->conv_flops_exec = (n->fused && !n->s3 && d_states && n->tower.cb)
                                 ? 2ll * M * 9 * ((long long)(16 + 4 * n->tower.cb_last_t) * F + 2ll * n->R * F * F) : n->conv_flops;
        chain->push_back(prof_event(n, st));
    }
    if (n->s3) {
        if (chain) chain->push_back(prof_event(n, st));
        const bool split_out = n->s3_fc_on || n->s3_head_on;
        TowerS3Params T3 = n->tower_s3;
        if (n->s3_head_on) T3.head_out = n->logits.as<float>() + (size_t)pos0 * n->logit_row;
        if (d_states) TG_HIP(launch_tower_s3_states(st, d_states, T3, x, nb, N, split_out));
        else TG_HIP(launch_tower_s3(st, d_planes, T3, x, nb, N, split_out));
        if (chain) chain->push_back(prof_event(n, st));
    } else if (n->fused) {
        if (chain) chain->push_back(prof_event(n, st));
        // (small batches of wide networks run split by channel tile through the exchange buffers; they and the position counters
        // belong to the engine stream's launches)
        if (d_states) TG_HIP(launch_tower_states(st, d_states, n->tower, x, nb, N, st == e->stream ? n->split_buf.as<float>() : nullptr));
        else TG_HIP(launch_tower(st, d_planes, n->tower, x, nb, N));
        if (chain) chain->push_back(prof_event(n, st));
    } else {
        TG_HIP(launch_conv3x3(st, d_planes, n->conv0.w.as<float>(), n->conv0.b.as<float>(), nullptr, x, M, N, n->cin_pad,
                              n->conv0.cout_pad, F, F, true));
        if (chain) chain->push_back(prof_event(n, st));
        for (int i = 0; i < n->R; i++) {
            TG_HIP(launch_conv3x3(st, x, n->res1[i].w.as<float>(), n->res1[i].b.as<float>(), nullptr, y, M, N, F,
                                  n->res1[i].cout_pad, F, F, true));
            if (chain) chain->push_back(prof_event(n, st));
            TG_HIP(launch_conv3x3(st, y, n->res2[i].w.as<float>(), n->res2[i].b.as<float>(), x, x, M, N, F,
                                  n->res2[i].cout_pad, F, F, true));
            if (chain) chain->push_back(prof_event(n, st));
        }
    }
    float* logits = n->logits.as<float>() + (size_t)pos0 * n->logit_row;
    bool value_done = false;
    if (e->cfg.policy_head == TG_HEAD_CONV) {
        const ConvLayer& L = n->policy_conv;
        if (!(n->s3 && n->s3_head_on))
            TG_HIP(launch_conv3x3(st, x, L.w.as<float>(), L.b.as<float>(), nullptr, logits, M, N, F, L.cout_pad, L.cout_pad, L.cout, false));
        // (exact-f32 path: the softmax's launch computes the value head too — one launch less per forward)
        const ValueHeadArgs vh{x, n->value_w.as<float>(), n->value_b, nsq * F, d_eval};
        const bool plain_value = !(n->s3 && (n->s3_fc_on || n->s3_head_on));
        TG_HIP(launch_softmax(st, logits, nsq * L.cout_pad, true, nsq, L.cout_pad, e->policy_size, nb, d_policy, nullptr, plain_value ? &vh : nullptr, &value_done));
    } else if (n->s3 && n->s3_fc_on) {
        float* stats = n->fc_stats_on ? n->fc_stats.as<float>() + (size_t)pos0 * n->fc_stat_stride * 2 : nullptr;
        const bool gather = !d_policy && pos0 == 0 && n->gather_on && net_gather_ok(e, nb);
        TG_HIP(launch_fc_s3(st, x, n->s3_fc.p, n->s3_fc_ring_on ? n->s3_fc_ring.p : nullptr, n->s3_fc_b.as<float>(), logits, nb, nsq * F, n->s3_np, n->s3_np,
                            e->policy_size + (n->value_in_fc ? 1 : 0), stats, e->policy_size, gather ? &n->gather : nullptr));
        if (d_policy && stats) TG_HIP(launch_softmax_stats(st, logits, n->s3_np, stats, n->fc_stat_blocks, n->fc_stat_stride, e->policy_size, nb, d_policy, d_eval));
        else if (d_policy) TG_HIP(launch_softmax(st, logits, n->s3_np, false, nsq, 0, e->policy_size, nb, d_policy, n->value_in_fc ? d_eval : nullptr));
    } else {
        float* stats = n->fc_stats_on ? n->fc_stats.as<float>() + (size_t)pos0 * n->fc_stat_stride * 2 : nullptr;
        const bool gather = !d_policy && pos0 == 0 && n->gather_on && net_gather_ok(e, nb);
        static const bool lin_src = !env_on("TG_FC_PERMUTED_SRC");  // A/B: the ring's LDS-DMA reads the [chunk][column][q] layout (same bits)
        TG_HIP(launch_gemm(st, x, nsq * F, n->policy_w.as<float>(), n->policy_b.as<float>(), logits, nb, nsq * F, n->policy_np,
                           n->policy_np, e->policy_size + (n->value_in_fc ? 1 : 0), !n->s3 && n->fused && n->tower.frag_out,
                           stats, e->policy_size, gather ? &n->gather : nullptr, lin_src ? n->policy_w_lin.as<float>() : nullptr));
        if (d_policy && stats) TG_HIP(launch_softmax_stats(st, logits, n->policy_np, stats, n->fc_stat_blocks, n->fc_stat_stride, e->policy_size, nb, d_policy, d_eval));
        else if (d_policy) TG_HIP(launch_softmax(st, logits, n->policy_np, false, nsq, 0, e->policy_size, nb, d_policy, n->value_in_fc ? d_eval : nullptr));
    }
    if (!d_policy && !(e->cfg.policy_head == TG_HEAD_FC5 && n->value_in_fc)) return fail(TG_ERR_STATE, "logits-only forward needs the FC head with the value column");
    if (e->cfg.policy_head == TG_HEAD_FC5 && n->value_in_fc) {
        // eval = tanh(logit P), written by the softmax kernel (or taken by the tree backup straight from the logits)
    } else if (n->s3 && (n->s3_fc_on || n->s3_head_on)) TG_HIP(launch_value_head_s3(st, x, n->value_w.as<float>(), n->value_b, nb, nsq * F, d_eval));
    else if (!value_done) TG_HIP(launch_value_head(st, x, n->value_w.as<float>(), n->value_b, nb, nsq * F, d_eval));
    if (chain) chain->push_back(prof_event(n, st));
    return TG_OK;
}

// after a stream sync: has a bounded wait of k_tower_split given up?  (never seen; it would mean a sibling workgroup of a position
// was not running — the counters are reset so that the next launch starts clean)
int net_poll_errors(TgEngine* e) {
    if (!e || !e->net || !e->net->split_ctl.p) return TG_OK;
    int err = 0;
    TG_HIP(hipMemcpy(&err, e->net->split_ctl.as<int>() + TOWER_SPLIT_CTL_WORDS - 32, 4, hipMemcpyDeviceToHost));
    if (!err) return TG_OK;
    TG_HIP(hipMemset(e->net->split_ctl.p, 0, (size_t)TOWER_SPLIT_CTL_WORDS * 4));
    return fail(TG_ERR_HIP, err == 2 ? "k_tower_split: the workgroups of one position ran on different XCDs (the same-L2 exchange is not safe on this "
                                       "device: set TG_SPLIT_AGENT_FENCES=1 or TG_NO_SPLIT_TOWER=1); the forward's results are invalid"
                                     : "k_tower_split: a workgroup waited for its position's siblings beyond the bound; the forward's results are invalid");
}

int net_profile_enable(TgEngine* e, int sample_every) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    if (sample_every < 0) return fail(TG_ERR_INVALID_ARG, "sample_every must be >= 0");
    e->net->prof_every = sample_every;
    e->net->prof_counter = 0;
    return TG_OK;
}

int net_profile_read(TgEngine* e, TgProfile* out) {
    if (!e || !e->net || !out) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    Net* n = e->net;
    TG_HIP(hipSetDevice(e->cfg.device));
    TG_HIP(hipStreamSynchronize(e->stream));
    prof_collect(n);
    out->conv_launches = n->conv_n; out->conv_ms = n->conv_ms; out->forwards = n->fwd_n; out->forward_ms = n->fwd_ms;
    out->conv_rows = n->conv_rows; out->conv_flops = n->conv_flops; out->conv_flops_executed = n->conv_flops_exec;
    n->conv_n = n->fwd_n = 0;
    n->conv_ms = n->fwd_ms = 0.0;
    return TG_OK;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_net_set_tensor(TgEngine* e, const char* name, const float* data, size_t count) { return net_set_tensor(e, name, data, count); }
int tg_net_finalize(TgEngine* e) { return net_finalize(e); }

// Network::default() (net5.rs:29-73 / net6.rs:29-69): every layer with tch's default initialisers — conv2d: weight
// KaimingUniform = U(±1/sqrt(fan_in)), bias 0; batch_norm2d: weight U(0,1), bias 0, running mean 0 / var 1; linear: weight
// KaimingUniform, bias U(±1/sqrt(in)).  (tch 0.7 is not vendored in the reference: these are its documented defaults; only the
// distribution matters.)  Draws come from Philox(seed; tensor index, element index).
int tg_net_init_random(TgEngine* e, uint64_t seed) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    Net* n = e->net;
    const int F = e->cfg.filters, R = e->cfg.res_blocks, nsq = e->g.nsq;
    uint32_t tensor_id = 0;
    auto uniform = [&](const std::string& name, size_t count, float lo, float hi) {
        std::vector<float> v(count);
        for (size_t i = 0; i < count; i += 4) {
            U4 r = philox4x32_10(seed, tensor_id, (uint32_t)(i >> 2), (uint32_t)((uint64_t)i >> 34), 0x696e6974u);
            for (size_t k = 0; k < 4 && i + k < count; k++) v[i + k] = lo + (hi - lo) * ((float)(r.v[k] >> 8) * (1.0f / 16777216.0f));
        }
        n->tensors[name] = std::move(v);
        tensor_id++;
    };
    auto constant = [&](const std::string& name, size_t count, float c) { n->tensors[name] = std::vector<float>(count, c); };
    auto conv = [&](const std::string& name, int O, int I) {
        float b = 1.0f / std::sqrt((float)(I * 9));
        uniform(name + ".weight", (size_t)O * I * 9, -b, b);
        constant(name + ".bias", O, 0.0f);
    };
    auto bn = [&](const std::string& name) {
        uniform(name + ".weight", F, 0.0f, 1.0f);
        constant(name + ".bias", F, 0.0f);
        constant(name + ".running_mean", F, 0.0f);
        constant(name + ".running_var", F, 1.0f);
    };
    auto linear = [&](const std::string& name, int out, int in) {
        float b = 1.0f / std::sqrt((float)in);
        uniform(name + ".weight", (size_t)out * in, -b, b);
        uniform(name + ".bias", out, -b, b);
    };
    conv("conv0", F, e->cin);
    bn("bn0");
    for (int i = 0; i < R; i++) {
        std::string r = "res" + std::to_string(i);
        conv(r + ".conv1", F, F);
        conv(r + ".conv2", F, F);
        bn(r + ".bn1");
        bn(r + ".bn2");
    }
    if (e->cfg.policy_head == TG_HEAD_FC5) linear("policy", e->policy_size, F * nsq);
    else conv("policy", e->policy_size / nsq, F);
    linear("value", 1, F * nsq);
    n->ready = false;
    return TG_OK;
}

// the tensor as last given to tg_net_set_tensor / tg_net_init_random / tg_train_commit (Network::save reads these)
int tg_net_get_tensor(TgEngine* e, const char* name, float* out, size_t count) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    if (!name || !out) return fail(TG_ERR_INVALID_ARG, "tg_net_get_tensor: null argument");
    auto it = e->net->tensors.find(name);
    if (it == e->net->tensors.end()) return fail(TG_ERR_INVALID_ARG, (std::string("tg_net_get_tensor: no tensor ") + name).c_str());
    if (it->second.size() != count) return fail(TG_ERR_INVALID_ARG, (std::string("tg_net_get_tensor: ") + name + " has " + std::to_string(it->second.size()) + " elements").c_str());
    std::memcpy(out, it->second.data(), count * sizeof(float));
    return TG_OK;
}
int tg_net_set_precision(TgEngine* e, int precision) {
    if (!e || !e->net) return fail(TG_ERR_STATE, "engine has no network (evaluator is not TG_EVAL_RESNET)");
    if (precision != TG_PRECISION_F32 && precision != TG_PRECISION_BF16X3) return fail(TG_ERR_INVALID_ARG, "unknown precision");
    e->net->precision = precision;
    e->net->ready = false;  // takes effect at the next tg_net_finalize
    return TG_OK;
}
int tg_profile_enable(TgEngine* e, int sample_every) { return net_profile_enable(e, sample_every); }
int tg_profile_read(TgEngine* e, TgProfile* out) { return net_profile_read(e, out); }

// Network::policy_eval (net5.rs:120-130): encode on the device straight into NHWC, forward, copy out
int tg_policy_eval_dev(TgEngine* e, int n, const void* d_states, float* d_policy, float* d_eval) {
    if (!net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    if (n < 0 || n > e->cfg.max_batch) return fail(TG_ERR_INVALID_ARG, "tg_policy_eval_dev: n out of range");
    if (n == 0) return TG_OK;
    // (logits-only forwards — d_policy = nullptr — are the search's private contract with the FC's gather epilogue)
    if (!d_states || !d_policy || !d_eval) return fail(TG_ERR_INVALID_ARG, "tg_policy_eval_dev: null argument");
    TG_HIP(hipSetDevice(e->cfg.device));
    return net_forward_states_dev(e, n, (const uint8_t*)d_states, d_policy, d_eval);
}

int tg_policy_eval(TgEngine* e, int n, const void* states, float* policy, float* eval) {
    if (!e) return fail(TG_ERR_INVALID_ARG, "null engine");
    if (n < 0 || (n > 0 && (!states || !policy || !eval))) return fail(TG_ERR_INVALID_ARG, "tg_policy_eval: bad arguments");
    if (n == 0) return TG_OK;  // net5.rs:121-123
    if (!net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    TG_HIP(hipSetDevice(e->cfg.device));
    const size_t sb = e->g.bytes, P = (size_t)e->policy_size;
    TG_HIP(e->s_policy.ensure((size_t)e->cfg.max_batch * P * 4));
    TG_HIP(e->s_eval.ensure((size_t)e->cfg.max_batch * 4));
    for (int off = 0; off < n; off += e->cfg.max_batch) {
        int k = std::min(e->cfg.max_batch, n - off);
        TG_HIP(hipMemcpyAsync(e->s_states.p, (const uint8_t*)states + (size_t)off * sb, (size_t)k * sb, hipMemcpyHostToDevice, e->stream));
        int rc = tg_policy_eval_dev(e, k, e->s_states.p, e->s_policy.as<float>(), e->s_eval.as<float>());
        if (rc) return rc;
        TG_HIP(hipMemcpyAsync(policy + (size_t)off * P, e->s_policy.p, (size_t)k * P * 4, hipMemcpyDeviceToHost, e->stream));
        TG_HIP(hipMemcpyAsync(eval + off, e->s_eval.p, (size_t)k * 4, hipMemcpyDeviceToHost, e->stream));
        TG_HIP(hipStreamSynchronize(e->stream));
        rc = net_poll_errors(e);
        if (rc) return rc;
    }
    return TG_OK;
}

// Network::forward_mcts (net5.rs:106-111) on caller-encoded NCHW planes
int tg_forward_mcts(TgEngine* e, int n, const float* planes, float* policy, float* eval) {
    if (!e) return fail(TG_ERR_INVALID_ARG, "null engine");
    if (n < 0 || (n > 0 && (!planes || !policy || !eval))) return fail(TG_ERR_INVALID_ARG, "tg_forward_mcts: bad arguments");
    if (n == 0) return TG_OK;
    if (!net_ready(e)) return fail(TG_ERR_STATE, "network weights not finalized (tg_net_finalize)");
    TG_HIP(hipSetDevice(e->cfg.device));
    Net* net = e->net;
    const size_t per = (size_t)e->cin * e->g.nsq, P = (size_t)e->policy_size;
    TG_HIP(net->planes_nchw.ensure((size_t)e->cfg.max_batch * per * 4));
    TG_HIP(e->s_policy.ensure((size_t)e->cfg.max_batch * P * 4));
    TG_HIP(e->s_eval.ensure((size_t)e->cfg.max_batch * 4));
    for (int off = 0; off < n; off += e->cfg.max_batch) {
        int k = std::min(e->cfg.max_batch, n - off);
        TG_HIP(hipMemcpyAsync(net->planes_nchw.p, planes + (size_t)off * per, (size_t)k * per * 4, hipMemcpyHostToDevice, e->stream));
        TG_HIP(launch_nchw_to_nhwc(e->stream, net->planes_nchw.as<float>(), k, e->cin, e->g.nsq, net->cin_pad, net->planes_nhwc.as<float>()));
        int rc = net_forward_dev(e, k, net->planes_nhwc.as<float>(), e->s_policy.as<float>(), e->s_eval.as<float>());
        if (rc) return rc;
        TG_HIP(hipMemcpyAsync(policy + (size_t)off * P, e->s_policy.p, (size_t)k * P * 4, hipMemcpyDeviceToHost, e->stream));
        TG_HIP(hipMemcpyAsync(eval + off, e->s_eval.p, (size_t)k * 4, hipMemcpyDeviceToHost, e->stream));
        TG_HIP(hipStreamSynchronize(e->stream));
    }
    return TG_OK;
}

}  // extern "C"
