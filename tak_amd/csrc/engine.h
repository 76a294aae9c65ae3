// engine.h — internal definition of the opaque TgEngine handle (host side of libtakgpu.so)
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/takgpu.h"
#include "board.cuh"

namespace tg {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define TG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return tg::fail(TG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

// RAII device allocation
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t nbytes) {
        if (nbytes <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, nbytes);
        if (e == hipSuccess) bytes = nbytes;
        return e;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct Net;     // net.hip
struct Search;  // search.hip
struct Trainer; // train.hip

}  // namespace tg

struct TgEngine {
    TgConfig cfg;
    tg::Geom g;
    hipStream_t stream = nullptr;
    hipStream_t half_stream[2] = {nullptr, nullptr};  // dual-stream rollouts (search.hip)
    hipEvent_t half_event[3] = {nullptr, nullptr, nullptr};
    int cin = 0;          // input channels
    int cin_pad = 0;      // channels per NHWC input row (multiple of 16, zero padded)
    int policy_size = 0;  // P
    bool legacy5 = false; // FC5 head → legacy 1575 LUT indices
    tg::DevBuf lut5;      // int16[25*4*32]
    // host-API scratch (sized for cfg.max_batch items)
    tg::DevBuf s_states, s_moves, s_counts, s_status, s_planes, s_policy, s_eval, s_index;
    tg::Net* net = nullptr;
    tg::Search* search = nullptr;
    tg::Trainer* trainer = nullptr;
    ~TgEngine();
};

namespace tg {
// engine.hip: host-side sanity check of caller-supplied packed states (TG_ERR_INVALID_ARG naming the first bad one)
int validate_states(const TgEngine* e, int k, const uint8_t* states, const char* who);
// net.hip
int net_create(TgEngine* e);
void net_destroy(Net* n);
// search.hip
void search_destroy(Search* s);
int search_poll_errors(TgEngine* e);  // after a stream sync: device error flags → status
int net_poll_errors(TgEngine* e);     // after a stream sync: the network kernels' error word → status
int net_set_tensor(TgEngine* e, const char* name, const float* data, size_t count);
int net_finalize(TgEngine* e);
bool net_ready(const TgEngine* e);
// forward on device: planes NHWC (n × nsq × C) → policy (n × P, softmax) and eval (n)
int net_forward_dev(TgEngine* e, int n, const float* d_planes_nhwc, float* d_policy, float* d_eval);
// same from packed states (device); encodes inside the fused tower when the topology allows, else via k_encode
int net_forward_states_dev(TgEngine* e, int n, const uint8_t* d_states, float* d_policy, float* d_eval);
bool net_takes_states(const TgEngine* e);  // true when the fused tower encodes in-kernel
struct FcGatherArgs;  // kernels.h
const float* net_fc_stats(const TgEngine* e, int* blocks, int* stride);  // block-wise softmax statistics that go with net_fc_logits' buffer ([batch][stride][2]), or nullptr
bool net_gather_ok(const TgEngine* e, int leaves);          // will a logits-only forward of `leaves` rows write the children's logits (net_set_gather)?
void net_set_gather(TgEngine* e, const FcGatherArgs* g);    // the search's child_pidx / leaf_rec / child_logit buffers, or nullptr
const float* net_fc_logits(const TgEngine* e, int* ld);  // FC head: logits buffer for logits-only forwards (d_policy = nullptr), else nullptr
int net_forward_states_at(TgEngine* e, int n, const uint8_t* d_states, float* d_policy, float* d_eval, hipStream_t st, int pos0);
bool net_profile_due(const TgEngine* e);
void net_profile_skip(TgEngine* e);
const std::map<std::string, std::vector<float>>* net_tensors(const TgEngine* e);  // tensors as given to tg_net_set_tensor
// train.hip
void trainer_destroy(Trainer* t);
}  // namespace tg
