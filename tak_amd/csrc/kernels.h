// kernels.h — launchers of the HIP kernels (definitions in *_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace tg {

// The ONE reader of the library's TG_* environment switches (engine.hip: the table of names is there, and tg_debug_switches lists
// the ones that are on).  env_on: set to anything but "" or "0".  env_int: the value, 0 = off / unset.
bool env_on(const char* name);
int env_int(const char* name);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): one `static LdsAttr` per launcher remembers the
// largest size it has set on each device of the process, under a lock (engines on several devices, trainers on several host threads).
struct LdsAttr {
    std::mutex guard;
    size_t configured[16] = {};
    hipError_t ensure(const void* kernel, size_t lds) {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(guard);
        if (dev >= 0 && dev < 16 && lds <= configured[dev]) return hipSuccess;
        if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) configured[dev] = lds;
        return hipSuccess;
    }
};

// board_kernels.hip
void launch_movegen(hipStream_t st, const uint8_t* states, int count, int n, uint16_t* moves, int32_t* counts);
void launch_play(hipStream_t st, uint8_t* states, int count, int n, const uint16_t* moves, uint8_t* status);
void launch_result(hipStream_t st, const uint8_t* states, int count, int n, uint8_t* results);
void launch_encode(hipStream_t st, const uint8_t* states, int count, int n, float* planes, bool nhwc);
void launch_encode_nhwc(hipStream_t st, const uint8_t* states, int count, int n, float* planes, int cstride);
void launch_board_pass(hipStream_t st, const uint8_t* states, const uint16_t* moves, int count, int n, uint8_t* out_states,
                       uint8_t* results, int32_t* counts, float* planes, int cstride);
void launch_augment(hipStream_t st, const uint8_t* states, const int32_t* n_moves, const uint16_t* moves, const uint32_t* visits, int count,
                    int n, int P, bool legacy5, const int16_t* lut5, uint8_t* out_states, float* pi);
void launch_move_index(hipStream_t st, const uint16_t* moves, int count, int n, bool legacy5, const int16_t* lut5, int32_t* index);
void launch_perft_count(hipStream_t st, const uint8_t* states, int count, int n, int32_t* nchild, uint8_t* terminal);
void launch_perft_expand(hipStream_t st, const uint8_t* states, int count, int n, const int64_t* offsets, const int32_t* root_of,
                         uint8_t* next_states, int32_t* next_root);

// net_kernels.hip
// stats_part / stats_blocks (optional, training forward): when the launch can emit BatchNorm's column sums from its
// accumulators (the halo kernel, all channels in one workgroup column) it writes *stats_blocks partial rows
// part[(block·2 + {Σ, Σ²})·CoutP + channel] as doubles and sets *stats_blocks > 0; otherwise *stats_blocks = 0 and the caller
// reduces the output itself (launch_bn_stats)
// stats_part / stats_blocks (optional; honoured by the halo kernel for layers whose channels one workgroup column covers, else
// *stats_blocks = 0): per-channel column sums of the output leave the kernel as doubles, one partial row per (workgroup, row group),
// in k_col_reduce's layout part[(row·2 + {0, 1})·CoutP + channel] — Σz and Σz² (BatchNorm's batch statistics, training forward), or,
// with bnb set (the data-gradient convolution: the output is dy of the layer below), that layer's BatchNorm-backward sums Σg, Σg·x̂
struct ConvBnBwdIn { const float* y; const float* z; const float* mean; const float* invstd; };  // of the layer BELOW, rows as the output's
hipError_t launch_conv3x3(hipStream_t st, const float* in, const float* Wp, const float* bias, const float* res, float* out,
                          int M, int n, int Cpad, int CoutP, int out_stride, int cout_valid, bool relu,
                          double* stats_part = nullptr, int* stats_blocks = nullptr, const ConvBnBwdIn* bnb = nullptr);
// mean, 1/σ and the running statistics from such partial rows (Σz, Σz² in double): replaces launch_bn_stats' two passes over z
hipError_t launch_bn_stats_from_partials(hipStream_t st, const double* part, int nblk, int M, int F, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var);
// fused residual tower (k_tower): per-layer weight / bias pointers
struct TowerParams {
    const float* w[48];   // per layer: Wp[K/16][CoutP][16]
    const float* b[48];   // per layer: bias[CoutP]
    int nlayers;          // 1 + 2·R
    int cin_pad;          // channels of the input planes (layer 0)
    int cin_last_t;       // 2 / 3: w[0] and the LDS image of the planes carry the last 16-channel chunk permuted (conv_mainloop.cuh); 4: plain
    int F;                // channels of every later layer (= CoutP of every layer)
    // halo image of the large-batch kernel (k_tower_halo): tile slot → cell | row << 16, position stride in cells
    const uint32_t* slotmap;
    int halo_pw, halo_ps;
    // final activations in FRAGMENT-MAJOR order for the policy FC (FC-head networks): per (tile of 16 positions, 16-k
    // chunk) one contiguous KB — lane (position, q) of the FC's B operand at ((pos/16·K/16 + chunk)·64 + (pos%16)·4 + q)·16 B —
    // so that an FC wave's activation load is 8 whole cache lines instead of 16 half-used ones
    int frag_out;
    // Constant input planes as a bias (states entry only; `cb` = 1 enables it).  Of the C_in input planes only the first
    // board_channels(n) = 26 / 28 vary over the board; the reserves one-hots, the colour plane and the fcd plane (46 of 72 on
    // 5×5, 64 of 92 on 6×6; alpha-tak/src/repr/reserves.rs:4-28, repr/game.rs:28-50) hold ONE value per position.  Their
    // 3×3 convolution is therefore a per-position bias that depends only on which taps of a square stay on the board — 9
    // border classes — and layer 0 runs its MFMAs over the board planes alone (K = 9·32 instead of 9·80 / 9·96):
    //   w0_board      conv0 restricted to the board planes, Wp[9·32/16][CoutP][16], last chunk permuted (cb_last_t)
    //   cplane_sums   S[plane − board_channels][class][F] = Σ over the taps valid in that class of the folded conv0 weight
    // and the kernel adds bias + Σ_{planes set} S + fcd · S[fcd plane] (in that order) in the epilogue of layer 0.
    int cb;
    int cb_cin_pad;       // 32
    int cb_last_t;        // 3: ten / twelve real channels in the last 16-channel chunk
    const float* w0_board;
    const float* cplane_sums;
    // k_tower_split (small batches of 128-filter networks): one counter per position of a launch (zero between launches) and an
    // error word a bounded wait raises; nullptr: the split path is not used
    unsigned* split_flags;
    int* split_err;
};
constexpr int TOWER_SPLIT_CTL_WORDS = 128 * 32 + 32;  // TowerParams.split_flags: a 128-byte line per position, then the error word's line
constexpr int TOWER_SPLIT_MAX_BATCH = 128;  // 8 small workgroups per position: the whole grid (≤ 1024 of them on 5×5, ≤ 512 on 6×6) is resident at once
// geometry of the halo image for a supported topology (positions per workgroup, position stride) and the slot table
bool tower_halo_geometry(int n, int F, int* pw, int* ps);
void tower_halo_slotmap(int n, int pw, int ps, uint32_t* out /* ceil(pw·n²/16)·16 entries */);
bool tower_supported(int n, int F, int cin_pad);
hipError_t launch_tower(hipStream_t st, const float* in, const TowerParams& T, float* out, int B, int n);
hipError_t launch_tower_states(hipStream_t st, const uint8_t* states, const TowerParams& T, float* out, int B, int n, float* scratch = nullptr);
// net_s3_kernels.hip — split-bf16 ("bf16x3") tower
struct TowerS3Params {
    const void* w[48];    // per layer: [chunk = tap·KC + kc][cout tile][hi|lo][q][cout in tile][8 bf16]
    const float* b[48];   // per layer: bias[F]
    int nlayers;
    int cin_pad;          // channels per row of NHWC f32 input planes (planes entry only)
    int F;
    // optional conv policy head computed from the resident image (out_split launches only)
    const void* head_w;   // [chunk][cout tile][hi|lo][q][cout][8 bf16], head_cout output channels (multiple of 32)
    const float* head_b;  // bias[head_cout]
    float* head_out;      // logits [positions][squares][head_cout]
    int head_cout;
    // halo image (k_tower_s3_halo): tile slot → cell | row << 16 for this topology's workgroup, position stride in cells
    const uint32_t* slotmap;
    int halo_ps;
    // constant input planes as a per-position bias (states entry; see TowerParams.cb): layer 0 over one 32-channel chunk of
    // board planes (w0_board: split fragments with KC = 1) + PB from the f32 table cplane_sums
    int cb;
    const void* w0_board;
    const float* cplane_sums;
};
// positions per workgroup and position stride of the split tower's halo image
bool tower_s3_halo_geometry(int n, int F, int* pw, int* ps);
bool tower_s3_supported(int n, int F);
// out_split: write the final activations in the split row layout (per 8 channels 16 B hi, 16 B lo) for k_fc_s3
hipError_t launch_tower_s3(hipStream_t st, const float* planes, const TowerS3Params& T, float* out, int B, int n, bool out_split);
hipError_t launch_tower_s3_states(hipStream_t st, const uint8_t* states, const TowerS3Params& T, float* out, int B, int n, bool out_split);
bool fc_s3_supported(int K, int NP);
int fc_s3_cols(int NP);
// Wp: [chunk of 32 k][NP/208 column blocks][q][hi|lo][208 outputs][8 bf16], k = sq·F + c
// stats (optional, NP % 112 == 0): block-wise softmax statistics over columns < n_soft, [M][NP/112][2] (softmax.cuh)
struct FcGatherArgs;
hipError_t launch_fc_s3(hipStream_t st, const float* act_split, const void* Wp, const void* Wr, const float* bias, float* out, int M, int K, int NP,
                        int out_stride, int n_valid, float* stats = nullptr, int n_soft = 0, const FcGatherArgs* gather = nullptr);
bool fc_s3_ring_supported(int M, int K, int n_valid);
hipError_t launch_fc_stats(hipStream_t st, const float* logits, int ld, int M, int n_soft, float* stats);  // softmax.cuh's statistics of logits in memory
hipError_t launch_value_head_s3(hipStream_t st, const float* act_split, const float* wv, float bv, int B, int len, float* eval);
// a_frag: A is in the fragment-major order of TowerParams.frag_out (needs fc_frag_supported(K, NP))
// stats (optional, needs fc_stats_supported): the block-wise softmax statistics of softmax.cuh over columns < n_soft,
// [M][FC_STAT_STRIDE][2] floats (11 block pairs, then {value pre-activation = column n_soft, 0}) — emitted by the FC's epilogue
// (full batches) or by a small kernel behind it (≤ 512 rows): same bits
// gather (optional, needs stats and fc_gather_supported): no logits row is written; for every row the logits of the children
// of that row's leaf go to child_logit[row][child] (search iterations: the tree backup needs nothing else)
struct FcGatherArgs {
    const uint16_t* child_pidx;  // [M][stride] policy index per child, 0xFFFF = unmapped
    const uint32_t* leaf_rec;    // [M][2] {children block, child count}
    float* child_logit;          // [M][stride]
    int stride;
};
hipError_t launch_gemm(hipStream_t st, const float* A, int lda, const float* Wp, const float* bias, float* out, int M, int K,
                       int NP, int out_stride, int n_valid, bool a_frag = false, float* stats = nullptr, int n_soft = 0,
                       const FcGatherArgs* gather = nullptr, const float* Wlin = nullptr);  // Wlin: Wp with every (chunk, tile) block in lane order (k_fc_ring's LDS-DMA source)
bool fc_frag_supported(int K, int NP);
bool fc_stats_supported(int K, int NP, int out_stride);
bool fc_gather_supported(int M, int K, int NP);
hipError_t launch_softmax_stats(hipStream_t st, const float* logits, int row_stride, const float* stats, int blocks, int stat_stride, int P,
                                int B, float* policy, float* eval);
hipError_t launch_value_head(hipStream_t st, const float* act, const float* wv, float bv, int B, int len, float* eval);
struct ValueHeadArgs { const float* act; const float* wv; float bv; int len; float* eval; };  // k_value_head's inputs
hipError_t launch_softmax(hipStream_t st, const float* logits, int row_stride, bool conv_head, int nsq, int ch_stride, int P,
                          int B, float* policy, float* eval = nullptr, const ValueHeadArgs* value = nullptr, bool* value_done = nullptr);
hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* src, int B, int C, int nsq, int Cpad, float* dst);

// train_kernels.hip
hipError_t launch_pack_conv_fwd(hipStream_t st, const float* W, int O, int I, int Ipad, int OP, float* dst);
hipError_t launch_pack_conv_bwd(hipStream_t st, const float* W, int O, int I, int Opad, int IP, float* dst);
hipError_t launch_pack_fc_fwd(hipStream_t st, const float* W, int P, int F, int nsq, int NP, float* dst);
hipError_t launch_pack_fc_bwd(hipStream_t st, const float* W, int P, int F, int nsq, int Pp, int KP, float* dst);
hipError_t launch_pack_value(hipStream_t st, const float* W, int F, int nsq, float* dst);
hipError_t launch_pad_copy(hipStream_t st, const float* src, int n, int npad, float* dst);
int col_reduce_blocks(int M, int F, int* rows_per_block);
hipError_t launch_bn_stats(hipStream_t st, const float* z, int M, int F, float eps, float momentum, double* part, float* mean,
                           float* invstd, float* running_mean, float* running_var);
hipError_t launch_bn_fwd_apply(hipStream_t st, const float* z, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* skip, float* y, int M, int F);
hipError_t launch_bn_bwd(hipStream_t st, const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                         const float* gamma, int M, int F, double* part, double* mean_g, double* mean_gx, float* grad_gamma,
                         float* grad_beta, float* dz, float* gskip, float* grad_conv_bias = nullptr,  // grad_conv_bias += column sums of dz
                         int sums_in_part = 0,   // > 0: Σg, Σg·x̂ already in `part` (that many partial rows, from launch_conv3x3's ConvBnBwdIn)
                         double* colsum_part = nullptr, int* colsum_rows = nullptr);  // dz's column-sum partials go there, not finalised
hipError_t launch_colsum_finalize(hipStream_t st, const double* part, int rows, int F, int valid, float* grad);
hipError_t launch_colsum_acc(hipStream_t st, const float* a, int M, int Fp, int valid, double* part, float* grad);
hipError_t launch_policy_loss(hipStream_t st, const float* logits, int row_stride, bool conv_head, int nsq, int ch_stride, int P, int B,
                              const float* pi, float inv_b, float* dlogits, float* logp_out, float* loss_rows);
hipError_t launch_value_train(hipStream_t st, const float* act, const float* wv, const float* bv, int B, int len, const float* zt,
                              float inv_b, float* eval, float* dpre, float* loss_rows);
hipError_t launch_value_bwd(hipStream_t st, const float* act, const float* dpre, const float* wv, int B, int F, int nsq, float* ds,
                            double* part, float* grad_w, float* grad_b, hipStream_t st_grad = nullptr);
size_t wgrad_conv_workspace(int B, int n, int I, int O);
hipError_t launch_wgrad_conv(hipStream_t st, const float* X, int xs, int I, const float* G, int gs, int O, int B, int n, float* part,
                             float* grad);
size_t wgrad_fc_workspace(int B, int K, int P);
hipError_t launch_wgrad_fc(hipStream_t st, const float* S, int K, const float* G, int gs, int P, int B, int F, int nsq, float* part,
                           float* grad);
hipError_t launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, float bc1, float bc2_sqrt, float gscale);
hipError_t launch_sum_rows(hipStream_t st, const float* rows, int n, double* out);

// search_kernels.hip
struct SearchDev;
struct SelfPlayDev;
void launch_select(hipStream_t st, const SearchDev& S, const uint8_t* active);
void launch_backup(hipStream_t st, const SearchDev& S);
void launch_backup_select(hipStream_t st, const SearchDev& S);  // backup of iteration i + select of iteration i+1 (batch 1)
void launch_dirichlet(hipStream_t st, const SearchDev& S, const uint8_t* active, float alpha, float ratio);
void launch_apply_noise(hipStream_t st, const SearchDev& S, const uint8_t* active, const float* noise, float ratio);
void launch_reroot(hipStream_t st, const SearchDev& S, const int32_t* op);
void launch_root_stats(hipStream_t st, const SearchDev& S, uint16_t* moves, uint32_t* visits, float* prior, float* q, int32_t* counts,
                       uint32_t* root_visits, float* root_q);
void launch_play_move(hipStream_t st, const SearchDev& S, const uint16_t* moves, const uint8_t* active, int32_t* op);
void launch_sp_opening(hipStream_t st, const SearchDev& S);
void launch_sp_instant_win(hipStream_t st, const SearchDev& S, const SelfPlayDev& P);
void launch_sp_finish(hipStream_t st, const SearchDev& S, const SelfPlayDev& P, int32_t* op);
void launch_sp_noise_mask(hipStream_t st, const SearchDev& S, const SelfPlayDev& P);
void launch_sp_pick(hipStream_t st, const SearchDev& S, const SelfPlayDev& P, int32_t* op);
void launch_sp_count_ply(hipStream_t st, const SelfPlayDev& P);

}  // namespace tg
