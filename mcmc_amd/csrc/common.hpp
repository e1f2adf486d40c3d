// Shared host-side plumbing of libmcmc_hip.so: error text, context, launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mcmc_hip.h"
#include "layout.hpp"

struct mcmc_ctx {
    uint32_t magic;
    int device;
    int num_cu;
    size_t lds_per_block;
};

namespace mcmc {

constexpr uint32_t kCtxMagic = 0x6D636D63u;

int set_error(int code, const char* fmt, ...);

inline int check_ctx(const mcmc_ctx* ctx) {
    if (!ctx || ctx->magic != kCtxMagic) return set_error(MCMC_ESTATE, "invalid or destroyed context");
#ifdef MCMC_DRY_CTX
    // `make asan` only (never the product library): a context made by mcmc_ctx_create_dry stands for "some device", so the
    // argument validation behind this check can run under the host sanitizers on a box without a GPU (tests/abi_check.cpp)
    if (ctx->device == -1) return MCMC_OK;
#endif
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev != ctx->device)
        return set_error(MCMC_ESTATE, "context belongs to device %d but device %d is current", ctx->device, dev);
    return MCMC_OK;
}

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize: sticky per function and device) the first
// time a (kernel, device) pair needs `bytes` or more, not on every launch; thread-safe.  capi.cpp.
int ensure_dyn_lds(const void* kern, size_t bytes, const char* what);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(MCMC_EHIP, "%s: %s", what, hipGetErrorString(e));
    return MCMC_OK;
}

#define MCMC_REQUIRE(cond, ...)                                       \
    do {                                                              \
        if (!(cond)) return mcmc::set_error(MCMC_EINVAL, __VA_ARGS__); \
    } while (0)

// kernel launchers (one per .hip file)
int launch_pack_x(const float* X, int64_t n_rows, int F, int dtype, void* xpack, hipStream_t s);
int launch_sample_weights(const float* mu, const float* rho, int M, int S, int F, int64_t pair_begin, int pair_count,
                          int noise_mode, const float* eps, uint64_t seed, const uint64_t* seed_dev, int dtype, void* wpack,
                          hipStream_t s);
int launch_philox_eps(uint64_t seed, int64_t g_begin, int pair_count, int F, float* eps, hipStream_t s);
struct DropMask;  // dropmask.hpp; NULL or mode 0 = eval-mode forward
int launch_forward_f32(const float* xpack, int64_t n_rows, int F, const float* wpack, int pair_count, const DropMask* dm, float* Y,
                       int64_t ldY, hipStream_t s);
int launch_forward_bf16(const mcmc_ctx* ctx, bool fast, const void* xpack, int64_t n_rows, int F, const void* wpack, int pair_count,
                        const DropMask* dm, float* Y, int64_t ldY, hipStream_t s);
int launch_reduce_update(const mcmc_ctx* ctx, const float* Y, int64_t ldY, int n_pairs, int64_t n_rows, double* sum,
                         double* sumsq, const float* lo, const float* inv_w, uint16_t* hist, int nbins, hipStream_t s);
int launch_reduce_merge(int n_parts, int64_t n_rows, const double* sum_parts, const double* sumsq_parts, const uint16_t* hist_parts,
                        int nbins, double* sum, double* sumsq, uint16_t* hist, int* overflow, hipStream_t s);
int launch_set_edges(int64_t n_rows, int64_t n0, const double* sum, const double* sumsq, int nbins, float range_sigmas,
                     float* lo, float* inv_w, hipStream_t s);
int launch_finalize(int64_t n_rows, int64_t n, const double* sum, const double* sumsq, const uint16_t* hist,
                    const float* lo, const float* inv_w, int nbins, const double* q_levels, int n_q, float* mean,
                    float* std, float* quant, hipStream_t s);
int launch_loss_stats(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, double* stats, hipStream_t s);
int launch_loss_forward(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total,
                        const double* stats, float alpha, float beta, double* part, hipStream_t s);
int launch_loss_backward(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total,
                         const double* stats, const double* part, float alpha, float beta, float grad_scale,
                         float* grad, hipStream_t s);

int64_t loss_workspace_doubles();
int launch_loss_fused(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                      float* loss, float* grad, double* ws, hipStream_t s);

int loss_static_init(int device);
int launch_loss_fused_static(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                             float* loss, float* grad, hipStream_t s);

int launch_train_sigma(const float* rho, float* sigma, int64_t n, hipStream_t s);
int launch_train_early_stop(const double* epoch_loss, double* best, int64_t* bad, uint8_t* active, int M, int64_t patience, hipStream_t s);
int launch_train_sample(const float* mu, const float* sigma, int M, int S, int F, int noise_mode, const float* eps_in, uint64_t seed,
                        float* eps, float* W, float* WT, hipStream_t s);
// x_stride / y_stride: floats between consecutive members' batches (rows * F / rows when dense)
int launch_train_forward(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* WT, int mask_mode, const uint8_t* m0,
                         const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, float* Y, float* act, hipStream_t s);
int launch_train_loss(const float* Y, const float* yb, int64_t y_stride, int M, int S, int64_t rows, float alpha, float beta, float* delta,
                      float* loss, double* loss_acc, hipStream_t s);
// fuse != nullptr: the backward kernel computes the loss and dLoss/dy itself from Y / yb (no separate loss launch)
struct TrainLossFuse {
    const float* Y;
    const float* yb;
    int64_t y_stride;
    float alpha, beta;
    float* loss;
    double* loss_acc;
};
int launch_train_backward(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, const float* WT, int mask_mode,
                          const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, const float* delta,
                          int row_blocks, float* dW, const float* act, const TrainLossFuse* fuse, hipStream_t s);
// bf16-operand variant of the large-batch forward / backward (train_bf16.hip): W only ([out][in] fp32 from launch_train_sample)
int launch_train_forward_bf16(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, int mask_mode,
                              const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, int row_blocks,
                              float* Y, void* act, hipStream_t s);
int64_t train_act_bytes_bf16(int64_t rows);
int launch_train_backward_bf16(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, int mask_mode,
                               const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, const float* delta,
                               int row_blocks, float* dW, const void* act, hipStream_t s);
int launch_train_grad(int M, int S, int row_blocks, int F, const float* eps, const float* sigma, const float* dW, float* gmu,
                      float* grho, double* normsq_part, hipStream_t s);
int train_row_blocks(int64_t rows);
int launch_train_apply(float* mu, float* rho, float* sigma, int M, int F, const float* gmu, const float* grho, float* am, float* av,
                       double* normsq_part, bool recompute_norm, const uint8_t* active, float lr, float beta1, float beta2,
                       float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, hipStream_t s);
int launch_train_update(float* mu, float* rho, float* sigma, int M, int S, int F, const float* eps, const float* dW, float* gmu,
                        float* grho, float* am, float* av, double* normsq_part, const uint8_t* active, float lr, float beta1,
                        float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, hipStream_t s);
int train_grad_blocks();
int64_t train_act_floats(int64_t rows);  // act: [pairs][tiles] saved activations, forward -> backward (nullptr: recompute)

int tft_params(int F);
int tft_flat_blocks();
int64_t tft_act_floats();  // floats per window of the saved-activation buffer (act: forward writes, backward reads; nullptr: recompute)
int launch_tft_transpose(const float* P, float* PT, int F, hipStream_t s);
int launch_tft_forward(const float* X, int64_t n_rows, const int* start, int B, int T, int F, const float* P, const float* PT, int mask_mode,
                       const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, const uint8_t* m3, uint64_t seed, float p_drop, float* pred,
                       float* act, hipStream_t s);
// fuse != nullptr: the backward kernel derives dLoss/dpred itself from pred / yb (no separate loss launch)
struct TftLossFuse {
    const float* pred;  // [B] predictions of the forward kernel, or nullptr: read `delta`
    const float* yb;    // [B] targets
    float alpha, beta;
    float* loss;        // [1]
    double* loss_acc;   // optional [1]
};
int launch_tft_backward(const float* X, int64_t n_rows, const int* start, int B, int T, int F, const float* P, const float* PT, int mask_mode,
                        const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, const uint8_t* m3, uint64_t seed, float p_drop,
                        const float* delta, float* dP, const float* act, const TftLossFuse* fuse, hipStream_t s);
// PT != nullptr: param is a TFT parameter vector of F features and PT its transposed image, refreshed by the same kernel
int launch_flat_update(float* param, int64_t n, const float* dP, int B, float* grad, float* am, float* av, double* normsq_part, float lr,
                       float beta1, float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm,
                       float* PT, int F, hipStream_t s);

}  // namespace mcmc
