// extern "C" entry points of libmcmc_hip.so: argument validation + dispatch to the launchers.
// Declarations and the reference lines each one replaces: include/mcmc_hip.h.
#include <string.h>

#include <map>
#include <mutex>
#include <new>
#include <utility>

#include "common.hpp"
#include "dropmask.hpp"

namespace mcmc {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ensure_dyn_lds(const void* kern, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;  // (kernel, device) -> largest limit set so far
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{kern, dev}];
    if (bytes <= have) return MCMC_OK;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipFuncSetAttribute(%s, %zu B LDS): %s", what, bytes, hipGetErrorString(e));
    have = bytes;
    return MCMC_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_dims(int input_dim, int dtype) {
    MCMC_REQUIRE(input_dim >= 1 && input_dim <= kMaxInputDim, "input_dim must be in [1, %d] (got %d)", kMaxInputDim, input_dim);
    MCMC_REQUIRE(dtype == DT_F32 || dtype == DT_BF16 || dtype == DT_BF16_FAST, "dtype must be MCMC_DT_F32, MCMC_DT_BF16 or MCMC_DT_BF16_FAST (got %d)", dtype);
    return MCMC_OK;
}

}  // namespace mcmc

using namespace mcmc;

extern "C" {

const char* mcmc_last_error(void) { return g_err; }

int mcmc_abi_version(void) { return 21; }

int mcmc_ctx_create(int device, mcmc_ctx** out) {
    MCMC_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    MCMC_REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(MCMC_EHIP, "device %d is %s; this library holds gfx950 (MI355X) code only", device, prop.gcnArchName);
    {  // per-device address of the allocation-free entry point's scratch, taken with this device current
        int prev = -1;
        e = hipGetDevice(&prev);
        if (e != hipSuccess) return set_error(MCMC_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
        if (prev != device && (e = hipSetDevice(device)) != hipSuccess) return set_error(MCMC_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        const int rc = loss_static_init(device);
        if (prev != device) (void)hipSetDevice(prev);
        if (rc) return rc;
    }
    mcmc_ctx* c = new (std::nothrow) mcmc_ctx;
    if (!c) return set_error(MCMC_EHIP, "out of host memory");
    c->magic = kCtxMagic;
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->lds_per_block = prop.sharedMemPerBlock;
    *out = c;
    return MCMC_OK;
}

#ifdef MCMC_DRY_CTX
// Sanitizer build only (Makefile target `asan`; not declared in include/mcmc_hip.h, not in libmcmc_hip.so): a context without a
// device, so that every entry point's argument validation is reachable on the CPU box.  Nothing may be launched through it.
int mcmc_ctx_create_dry(mcmc_ctx** out) {
    MCMC_REQUIRE(out != nullptr, "out is NULL");
    mcmc_ctx* c = new (std::nothrow) mcmc_ctx;
    if (!c) return set_error(MCMC_EHIP, "out of host memory");
    c->magic = kCtxMagic;
    c->device = -1;
    c->num_cu = 256;
    c->lds_per_block = 160 * 1024;
    *out = c;
    return MCMC_OK;
}
#endif

int mcmc_ctx_destroy(mcmc_ctx* ctx) {
    if (!ctx || ctx->magic != kCtxMagic) return set_error(MCMC_ESTATE, "invalid or destroyed context");
    ctx->magic = 0;
    delete ctx;
    return MCMC_OK;
}

int64_t mcmc_param_count(int input_dim) {
    if (input_dim < 1 || input_dim > kMaxInputDim) return set_error(MCMC_EINVAL, "input_dim must be in [1, %d]", kMaxInputDim);
    return flat_param_count(input_dim);
}

int64_t mcmc_xpack_bytes(int64_t n_rows, int input_dim, int dtype) {
    if (int rc = check_dims(input_dim, dtype)) return rc;
    if (n_rows < 0) return set_error(MCMC_EINVAL, "n_rows < 0");
    return rows_pad(n_rows) * k0_pad(input_dim, dtype) * (dtype == DT_F32 ? 4 : 2);
}

int64_t mcmc_wpack_bytes(int input_dim, int dtype) {
    if (int rc = check_dims(input_dim, dtype)) return rc;
    return dtype == DT_F32 ? WPackF32(input_dim).total * 4 : dtype == DT_BF16 ? WPackBF16(input_dim).total : WPackBF16F(input_dim).total;
}

int mcmc_pack_x(mcmc_ctx* ctx, const float* X, int64_t n_rows, int input_dim, int dtype, void* xpack, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_dims(input_dim, dtype)) return rc;
    MCMC_REQUIRE(n_rows >= 0, "n_rows < 0");
    if (n_rows == 0) return MCMC_OK;
    MCMC_REQUIRE(X && xpack, "X / xpack is NULL");
    MCMC_REQUIRE(aligned16(xpack), "xpack must be 16-byte aligned");
    return launch_pack_x(X, n_rows, input_dim, dtype, xpack, (hipStream_t)stream);
}

static int sample_weights_impl(mcmc_ctx* ctx, const float* mu, const float* rho, int n_members, int n_samples, int input_dim,
                               int64_t pair_begin, int pair_count, int noise_mode, const float* eps, uint64_t seed,
                               const uint64_t* seed_dev, int dtype, void* wpack, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_dims(input_dim, dtype)) return rc;
    MCMC_REQUIRE(n_members >= 1 && n_samples >= 1, "n_members and n_samples must be >= 1");
    MCMC_REQUIRE((int64_t)n_members * n_samples <= 0x7FFFFFFFLL, "n_members * n_samples exceeds 2^31-1");
    MCMC_REQUIRE(pair_count >= 0 && pair_begin >= 0 && pair_begin + pair_count <= (int64_t)n_members * n_samples,
                 "pair range [%lld, +%d) outside 0..%lld", (long long)pair_begin, pair_count, (long long)n_members * n_samples);
    MCMC_REQUIRE(noise_mode >= MCMC_NOISE_NONE && noise_mode <= MCMC_NOISE_PHILOX, "bad noise_mode %d", noise_mode);
    if (pair_count == 0) return MCMC_OK;
    MCMC_REQUIRE(mu && rho && wpack, "mu / rho / wpack is NULL");
    MCMC_REQUIRE(noise_mode != MCMC_NOISE_INJECTED || eps, "MCMC_NOISE_INJECTED needs eps");
    MCMC_REQUIRE(aligned16(wpack), "wpack must be 16-byte aligned");
    return launch_sample_weights(mu, rho, n_members, n_samples, input_dim, pair_begin, pair_count, noise_mode, eps, seed, seed_dev,
                                 dtype, wpack, (hipStream_t)stream);
}

int mcmc_sample_weights(mcmc_ctx* ctx, const float* mu, const float* rho, int n_members, int n_samples, int input_dim,
                        int64_t pair_begin, int pair_count, int noise_mode, const float* eps, uint64_t seed, int dtype,
                        void* wpack, mcmc_stream_t stream) {
    return sample_weights_impl(ctx, mu, rho, n_members, n_samples, input_dim, pair_begin, pair_count, noise_mode, eps, seed, nullptr,
                               dtype, wpack, stream);
}

int mcmc_sample_weights_seedptr(mcmc_ctx* ctx, const float* mu, const float* rho, int n_members, int n_samples, int input_dim,
                                int64_t pair_begin, int pair_count, const uint64_t* seed_dev, int dtype, void* wpack,
                                mcmc_stream_t stream) {
    MCMC_REQUIRE(seed_dev != nullptr, "seed_dev is NULL");
    return sample_weights_impl(ctx, mu, rho, n_members, n_samples, input_dim, pair_begin, pair_count, MCMC_NOISE_PHILOX, nullptr, 0,
                               seed_dev, dtype, wpack, stream);
}

int mcmc_philox_eps(mcmc_ctx* ctx, uint64_t seed, int64_t g_begin, int pair_count, int input_dim, float* eps, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_dims(input_dim, DT_F32)) return rc;
    MCMC_REQUIRE(pair_count >= 0 && g_begin >= 0 && g_begin + pair_count <= 0x7FFFFFFFLL, "bad pair range");
    if (pair_count == 0) return MCMC_OK;
    MCMC_REQUIRE(eps, "eps is NULL");
    return launch_philox_eps(seed, g_begin, pair_count, input_dim, eps, (hipStream_t)stream);
}

static int forward_impl(mcmc_ctx* ctx, int dtype, const void* xpack, int64_t n_rows, int input_dim, const void* wpack, int pair_count,
                        const DropMask* dm, float* Y, int64_t ldY, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_dims(input_dim, dtype)) return rc;
    MCMC_REQUIRE(n_rows >= 0 && pair_count >= 0, "n_rows / pair_count < 0");
    if (n_rows == 0 || pair_count == 0) return MCMC_OK;
    MCMC_REQUIRE(xpack && wpack && Y, "xpack / wpack / Y is NULL");
    MCMC_REQUIRE(aligned16(xpack) && aligned16(wpack), "xpack and wpack must be 16-byte aligned");
    MCMC_REQUIRE(ldY >= n_rows && ldY % 4 == 0, "ldY must be >= n_rows and a multiple of 4 (got %lld)", (long long)ldY);
    if (dtype == DT_F32)
        return launch_forward_f32((const float*)xpack, n_rows, input_dim, (const float*)wpack, pair_count, dm, Y, ldY, (hipStream_t)stream);
    return launch_forward_bf16(ctx, dtype == DT_BF16_FAST, xpack, n_rows, input_dim, wpack, pair_count, dm, Y, ldY, (hipStream_t)stream);
}

int mcmc_forward(mcmc_ctx* ctx, int dtype, const void* xpack, int64_t n_rows, int input_dim, const void* wpack, int pair_count,
                 const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, float keep_scale, float* Y, int64_t ldY,
                 mcmc_stream_t stream) {
    const bool any = mask0 || mask1 || mask2, all = mask0 && mask1 && mask2;
    MCMC_REQUIRE(any == all, "dropout masks: pass all three or none");
    if (!any) return forward_impl(ctx, dtype, xpack, n_rows, input_dim, wpack, pair_count, nullptr, Y, ldY, stream);
    MCMC_REQUIRE((reinterpret_cast<uintptr_t>(mask0) & 3u) == 0 && (reinterpret_cast<uintptr_t>(mask1) & 3u) == 0 &&
                     (reinterpret_cast<uintptr_t>(mask2) & 3u) == 0,
                 "dropout masks must be 4-byte aligned");
    DropMask dm{};
    dm.m[0] = mask0;
    dm.m[1] = mask1;
    dm.m[2] = mask2;
    dm.mode = 1;
    dm.scale = keep_scale;
    dm.M = dm.S = 1;
    return forward_impl(ctx, dtype, xpack, n_rows, input_dim, wpack, pair_count, &dm, Y, ldY, stream);
}

int mcmc_forward_dropout(mcmc_ctx* ctx, int dtype, const void* xpack, int64_t n_rows, int input_dim, const void* wpack, int n_members,
                         int n_samples, int64_t pair_begin, int pair_count, uint64_t mask_seed, float p_drop, int64_t row_begin, float* Y,
                         int64_t ldY, mcmc_stream_t stream) {
    MCMC_REQUIRE(n_members >= 1 && n_samples >= 1, "n_members and n_samples must be >= 1");
    MCMC_REQUIRE(pair_begin >= 0 && pair_count >= 0 && pair_begin + pair_count <= (int64_t)n_members * n_samples,
                 "pair range [%lld, +%d) outside n_members*n_samples", (long long)pair_begin, pair_count);
    MCMC_REQUIRE(row_begin >= 0, "row_begin < 0");
    MCMC_REQUIRE((int64_t)n_members * n_samples <= 0x7FFFFFFFLL && n_rows <= 0xFFFFFFFFLL && row_begin <= 0xFFFFFFFFLL - n_rows,
                 "pair or row index exceeds the 32-bit Philox counter");
    MCMC_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "p_drop must be in [0, 1)");
    DropMask dm{};
    dm.seed = mask_seed;
    dm.mode = 2;
    dm.keep_prob = 1.0f - p_drop;
    dm.scale = (float)(1.0 / (1.0 - (double)p_drop));  // torch: scale = 1/(1-p) in double, applied as fp32
    dm.M = n_members;
    dm.S = n_samples;
    dm.pair_begin = pair_begin;
    dm.row_begin = row_begin;
    return forward_impl(ctx, dtype, xpack, n_rows, input_dim, wpack, pair_count, &dm, Y, ldY, stream);
}

static int check_nbins(int nbins) {
    MCMC_REQUIRE(nbins == 64 || nbins == 128 || nbins == 256 || nbins == 512, "nbins must be 64, 128, 256 or 512 (got %d)", nbins);
    return MCMC_OK;
}

int mcmc_reduce_update(mcmc_ctx* ctx, const float* Y, int64_t ldY, int n_pairs, int64_t n_rows, double* sum, double* sumsq,
                       const float* lo, const float* inv_w, uint16_t* hist, int nbins, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n_rows >= 0 && n_pairs >= 0, "n_rows / n_pairs < 0");
    if (n_rows == 0 || n_pairs == 0) return MCMC_OK;
    MCMC_REQUIRE(Y && sum && sumsq, "Y / sum / sumsq is NULL");
    MCMC_REQUIRE(ldY >= n_rows, "ldY < n_rows");
    if (hist) {
        if (int rc = check_nbins(nbins)) return rc;
        MCMC_REQUIRE(lo && inv_w, "hist needs lo and inv_w (mcmc_reduce_set_edges)");
        MCMC_REQUIRE((reinterpret_cast<uintptr_t>(hist) & 3u) == 0, "hist must be 4-byte aligned");
    }
    return launch_reduce_update(ctx, Y, ldY, n_pairs, n_rows, sum, sumsq, lo, inv_w, hist, nbins, (hipStream_t)stream);
}

int mcmc_reduce_merge(mcmc_ctx* ctx, int n_parts, int64_t n_rows, const double* sum_parts, const double* sumsq_parts,
                      const uint16_t* hist_parts, int nbins, double* sum, double* sumsq, uint16_t* hist, int32_t* overflow,
                      mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n_parts >= 1 && n_rows >= 0, "n_parts < 1 or n_rows < 0");
    if (n_rows == 0) return MCMC_OK;
    MCMC_REQUIRE(sum_parts && sumsq_parts && sum && sumsq, "NULL moments pointer");
    MCMC_REQUIRE((hist_parts == nullptr) == (hist == nullptr), "hist_parts and hist must both be given or both be NULL");
    if (hist) {
        if (int rc = check_nbins(nbins)) return rc;
        MCMC_REQUIRE(aligned16(hist_parts) && aligned16(hist), "hist_parts and hist must be 16-byte aligned");
    }
    return launch_reduce_merge(n_parts, n_rows, sum_parts, sumsq_parts, hist_parts, nbins, sum, sumsq, hist, overflow, (hipStream_t)stream);
}

int mcmc_reduce_set_edges(mcmc_ctx* ctx, int64_t n_rows, int64_t n0, const double* sum, const double* sumsq, int nbins,
                          float range_sigmas, float* lo, float* inv_w, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_nbins(nbins)) return rc;
    MCMC_REQUIRE(n_rows >= 0 && n0 >= 1, "n_rows < 0 or n0 < 1");
    MCMC_REQUIRE(range_sigmas > 0.0f, "range_sigmas must be > 0");
    if (n_rows == 0) return MCMC_OK;
    MCMC_REQUIRE(sum && sumsq && lo && inv_w, "NULL pointer");
    return launch_set_edges(n_rows, n0, sum, sumsq, nbins, range_sigmas, lo, inv_w, (hipStream_t)stream);
}

int mcmc_reduce_finalize(mcmc_ctx* ctx, int64_t n_rows, int64_t n, const double* sum, const double* sumsq, const uint16_t* hist,
                         const float* lo, const float* inv_w, int nbins, const double* q_levels, int n_q, float* mean, float* std,
                         float* quant, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n_rows >= 0 && n >= 1, "n_rows < 0 or n < 1");
    MCMC_REQUIRE(n_q >= 0 && n_q <= 8, "n_q must be in [0, 8]");
    if (n_rows == 0) return MCMC_OK;
    MCMC_REQUIRE(sum && sumsq && mean && std, "NULL pointer");
    if (n_q > 0) {
        if (int rc = check_nbins(nbins)) return rc;
        MCMC_REQUIRE(hist && lo && inv_w && q_levels && quant, "quantiles need hist, lo, inv_w, q_levels, quant");
        MCMC_REQUIRE(n <= 65535, "uint16 histogram holds at most 65,535 samples per row (n = %lld)", (long long)n);
        for (int i = 0; i < n_q; ++i) MCMC_REQUIRE(q_levels[i] >= 0.0 && q_levels[i] <= 1.0, "q_levels[%d] outside [0,1]", i);
    }
    return launch_finalize(n_rows, n, sum, sumsq, hist, lo, inv_w, nbins, q_levels, n_q, mean, std, quant, (hipStream_t)stream);
}

static int check_loss(const float* p, const float* t, int64_t n, int64_t n_total) {
    MCMC_REQUIRE(n >= 0 && n_total >= n && n_total >= 1, "need 0 <= n <= n_total and n_total >= 1");
    MCMC_REQUIRE(n == 0 || (p && t), "p / t is NULL");
    MCMC_REQUIRE(aligned16(p) && aligned16(t), "p and t must be 16-byte aligned");
    return MCMC_OK;
}

int mcmc_loss_stats(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, double* stats, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_loss(p, t, n, n > 0 ? n : 1)) return rc;
    MCMC_REQUIRE(stats, "stats is NULL");
    if (n == 0) return MCMC_OK;
    return launch_loss_stats(ctx, p, t, n, stats, (hipStream_t)stream);
}

int mcmc_loss_forward(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total, const double* stats, float alpha,
                      float beta, double* part, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_loss(p, t, n, n_total)) return rc;
    MCMC_REQUIRE(stats && part, "stats / part is NULL");
    if (n == 0) return MCMC_OK;
    return launch_loss_forward(ctx, p, t, n, n_total, stats, alpha, beta, part, (hipStream_t)stream);
}

int mcmc_loss_backward(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total, const double* stats,
                       const double* part, float alpha, float beta, float grad_scale, float* grad, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_loss(p, t, n, n_total)) return rc;
    MCMC_REQUIRE(stats && part, "stats / part is NULL");
    if (n == 0) return MCMC_OK;
    MCMC_REQUIRE(grad && aligned16(grad), "grad is NULL or not 16-byte aligned");
    return launch_loss_backward(ctx, p, t, n, n_total, stats, part, alpha, beta, grad_scale, grad, (hipStream_t)stream);
}

int64_t mcmc_loss_workspace_bytes(void) { return loss_workspace_doubles() * (int64_t)sizeof(double); }

int mcmc_loss_fused(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale, float* loss,
                    float* grad, void* workspace, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n >= 1, "n must be >= 1");
    if (int rc = check_loss(p, t, n, n)) return rc;
    MCMC_REQUIRE(loss, "loss is NULL");
    MCMC_REQUIRE(!grad || aligned16(grad), "grad must be 16-byte aligned");
    MCMC_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "workspace is NULL or not 8-byte aligned");
    return launch_loss_fused(ctx, p, t, n, alpha, beta, grad_scale, loss, grad, (double*)workspace, (hipStream_t)stream);
}

int mcmc_loss_fused_static(mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                           float* loss, float* grad, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n >= 1, "n must be >= 1");
    MCMC_REQUIRE(n <= MCMC_LOSS_STATIC_MAX_N, "n = %lld exceeds MCMC_LOSS_STATIC_MAX_N (%lld): use mcmc_loss_fused with a workspace",
                 (long long)n, (long long)MCMC_LOSS_STATIC_MAX_N);
    if (int rc = check_loss(p, t, n, n)) return rc;
    MCMC_REQUIRE(loss, "loss is NULL");
    MCMC_REQUIRE(!grad || aligned16(grad), "grad must be 16-byte aligned");
    return launch_loss_fused_static(ctx, p, t, n, alpha, beta, grad_scale, loss, grad, (hipStream_t)stream);
}

// ---- training step (row f1) ----------------------------------------------------------------------
static int check_train(int n_members, int n_samples, int input_dim, int64_t rows) {
    if (int rc = check_dims(input_dim, DT_F32)) return rc;
    MCMC_REQUIRE(n_members >= 1 && n_samples >= 1 && (int64_t)n_members * n_samples <= 65535,
                 "need n_members, n_samples >= 1 and n_members*n_samples <= 65535");
    MCMC_REQUIRE(rows >= 1, "rows must be >= 1");
    return MCMC_OK;
}

static int check_masks(int mask_mode, const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, float p_drop) {
    MCMC_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "mask_mode must be 0 (none), 1 (host) or 2 (philox)");
    MCMC_REQUIRE(mask_mode != 1 || (m0 && m1 && m2), "mask_mode 1 needs mask0, mask1 and mask2");
    MCMC_REQUIRE(mask_mode != 1 || ((reinterpret_cast<uintptr_t>(m0) | reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(m2)) & 3u) == 0,
                 "host keep-masks must be 4-byte aligned (the kernels read four flags per load)");
    MCMC_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "p_drop must be in [0, 1)");
    return MCMC_OK;
}

int mcmc_train_sigma(mcmc_ctx* ctx, const float* rho, float* sigma, int64_t n, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return MCMC_OK;
    MCMC_REQUIRE(rho && sigma, "rho / sigma is NULL");
    return launch_train_sigma(rho, sigma, n, (hipStream_t)stream);
}

int mcmc_train_sample(mcmc_ctx* ctx, const float* mu, const float* sigma, int n_members, int n_samples, int input_dim,
                      int noise_mode, const float* eps_in, uint64_t seed, float* eps, float* W, float* WT, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, 1)) return rc;
    MCMC_REQUIRE(noise_mode >= MCMC_NOISE_NONE && noise_mode <= MCMC_NOISE_PHILOX, "bad noise_mode %d", noise_mode);
    MCMC_REQUIRE(noise_mode != MCMC_NOISE_INJECTED || eps_in, "MCMC_NOISE_INJECTED needs eps_in");
    MCMC_REQUIRE(mu && sigma && eps && W && WT, "NULL pointer");
    return launch_train_sample(mu, sigma, n_members, n_samples, input_dim, noise_mode, eps_in, seed, eps, W, WT, (hipStream_t)stream);
}

static int train_forward_impl(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* WT,
                              int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                              float p_drop, float* Y, float* act, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, rows)) return rc;
    if (int rc = check_masks(mask_mode, mask0, mask1, mask2, p_drop)) return rc;
    MCMC_REQUIRE(xb && WT && Y, "NULL pointer");
    return launch_train_forward(xb, rows * input_dim, rows, input_dim, n_members, n_samples, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop, Y,
                                act, (hipStream_t)stream);
}

int mcmc_train_forward(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* WT,
                       int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                       float p_drop, float* Y, mcmc_stream_t stream) {
    return train_forward_impl(ctx, xb, rows, input_dim, n_members, n_samples, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop, Y, nullptr,
                              stream);
}

int mcmc_train_forward_save(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* WT,
                            int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed,
                            float p_drop, float* Y, float* act, mcmc_stream_t stream) {
    MCMC_REQUIRE(act, "act is NULL");
    return train_forward_impl(ctx, xb, rows, input_dim, n_members, n_samples, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop, Y, act,
                              stream);
}

int mcmc_train_loss(mcmc_ctx* ctx, const float* Y, const float* yb, int n_members, int n_samples, int64_t rows, float alpha,
                    float beta, float* delta, float* loss, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, 1, rows)) return rc;
    MCMC_REQUIRE(Y && yb && delta && loss, "NULL pointer");
    return launch_train_loss(Y, yb, rows, n_members, n_samples, rows, alpha, beta, delta, loss, nullptr, (hipStream_t)stream);
}

int mcmc_train_loss_acc(mcmc_ctx* ctx, const float* Y, const float* yb, int n_members, int n_samples, int64_t rows, float alpha,
                        float beta, float* delta, float* loss, double* loss_acc, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, 1, rows)) return rc;
    MCMC_REQUIRE(Y && yb && delta && loss && loss_acc, "NULL pointer");
    return launch_train_loss(Y, yb, rows, n_members, n_samples, rows, alpha, beta, delta, loss, loss_acc, (hipStream_t)stream);
}

static int train_backward_impl(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* W,
                               const float* WT, int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2,
                               uint64_t mask_seed, float p_drop, const float* delta, int row_blocks, float* dW, const float* act,
                               mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, rows)) return rc;
    if (int rc = check_masks(mask_mode, mask0, mask1, mask2, p_drop)) return rc;
    MCMC_REQUIRE(xb && W && WT && delta && dW, "NULL pointer");
    MCMC_REQUIRE(row_blocks >= 1 && row_blocks <= 65535, "row_blocks must be in [1, 65535] (got %d)", row_blocks);
    return launch_train_backward(xb, rows * input_dim, rows, input_dim, n_members, n_samples, W, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop,
                                 delta, row_blocks, dW, act, nullptr, (hipStream_t)stream);
}

int mcmc_train_backward(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* W,
                        const float* WT, int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2,
                        uint64_t mask_seed, float p_drop, const float* delta, float* dW, mcmc_stream_t stream) {
    return train_backward_impl(ctx, xb, rows, input_dim, n_members, n_samples, W, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop,
                               delta, 1, dW, nullptr, stream);
}

int mcmc_train_row_blocks(int64_t rows) { return train_row_blocks(rows < 0 ? 0 : rows); }

int mcmc_train_backward_blocks(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples,
                               const float* W, const float* WT, int mask_mode, const uint8_t* mask0, const uint8_t* mask1,
                               const uint8_t* mask2, uint64_t mask_seed, float p_drop, const float* delta, int row_blocks, float* dW,
                               mcmc_stream_t stream) {
    return train_backward_impl(ctx, xb, rows, input_dim, n_members, n_samples, W, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop,
                               delta, row_blocks, dW, nullptr, stream);
}

int mcmc_train_backward_saved(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples,
                              const float* W, const float* WT, int mask_mode, const uint8_t* mask0, const uint8_t* mask1,
                              const uint8_t* mask2, uint64_t mask_seed, float p_drop, const float* delta, int row_blocks, float* dW,
                              const float* act, mcmc_stream_t stream) {
    MCMC_REQUIRE(act, "act is NULL");
    MCMC_REQUIRE(row_blocks == train_row_blocks(rows), "row_blocks must be mcmc_train_row_blocks(rows) = %d: the saved activations are laid out "
                 "by the tile shape the forward launch chose from it (got %d)", train_row_blocks(rows), row_blocks);
    return train_backward_impl(ctx, xb, rows, input_dim, n_members, n_samples, W, WT, mask_mode, mask0, mask1, mask2, mask_seed, p_drop,
                               delta, row_blocks, dW, act, stream);
}

int mcmc_train_forward_bf16(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* W,
                            int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                            int row_blocks, float* Y, void* act, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, rows)) return rc;
    if (int rc = check_masks(mask_mode, mask0, mask1, mask2, p_drop)) return rc;
    MCMC_REQUIRE(xb && W && Y, "NULL pointer");
    MCMC_REQUIRE(aligned16(W) && aligned16(act), "W and act must be 16-byte aligned");
    MCMC_REQUIRE(row_blocks >= 1 && row_blocks <= 65535, "row_blocks must be in [1, 65535] (got %d)", row_blocks);
    return launch_train_forward_bf16(xb, rows * input_dim, rows, input_dim, n_members, n_samples, W, mask_mode, mask0, mask1, mask2, mask_seed,
                                     p_drop, row_blocks, Y, act, (hipStream_t)stream);
}

int mcmc_train_backward_bf16(mcmc_ctx* ctx, const float* xb, int64_t rows, int input_dim, int n_members, int n_samples, const float* W,
                             int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2, uint64_t mask_seed, float p_drop,
                             const float* delta, int row_blocks, float* dW, const void* act, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, rows)) return rc;
    if (int rc = check_masks(mask_mode, mask0, mask1, mask2, p_drop)) return rc;
    MCMC_REQUIRE(xb && W && delta && dW, "NULL pointer");
    MCMC_REQUIRE(aligned16(W) && aligned16(act), "W and act must be 16-byte aligned");
    MCMC_REQUIRE(row_blocks >= 1 && row_blocks <= 65535, "row_blocks must be in [1, 65535] (got %d)", row_blocks);
    return launch_train_backward_bf16(xb, rows * input_dim, rows, input_dim, n_members, n_samples, W, mask_mode, mask0, mask1, mask2, mask_seed,
                                      p_drop, delta, row_blocks, dW, act, (hipStream_t)stream);
}

int64_t mcmc_train_act_bytes_bf16(int64_t rows) { return rows < 1 ? 0 : train_act_bytes_bf16(rows); }

int mcmc_train_norm_blocks(void) { return train_grad_blocks(); }

int mcmc_train_early_stop(mcmc_ctx* ctx, const double* epoch_loss, double* best, int64_t* bad, uint8_t* active, int n_members,
                          int64_t patience, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n_members >= 1, "n_members must be >= 1");
    MCMC_REQUIRE(epoch_loss && best && bad && active, "NULL pointer");
    return launch_train_early_stop(epoch_loss, best, bad, active, n_members, patience, (hipStream_t)stream);
}

int64_t mcmc_train_act_floats(int64_t rows) { return rows < 1 ? 0 : train_act_floats(rows); }

int mcmc_train_update(mcmc_ctx* ctx, float* mu, float* rho, float* sigma, int n_members, int n_samples, int input_dim,
                      const float* eps, const float* dW, float* gmu, float* grho, float* adam_m, float* adam_v, double* norm_scratch,
                      const uint8_t* active, float lr, float beta1, float beta2, float adam_eps, float weight_decay, float max_norm,
                      int64_t step, float* gnorm, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, 1)) return rc;
    MCMC_REQUIRE(mu && rho && sigma && eps && dW && gmu && grho && adam_m && adam_v && norm_scratch, "NULL pointer");
    MCMC_REQUIRE(step >= 1, "step is 1-based");
    MCMC_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "betas must be in [0, 1)");
    return launch_train_update(mu, rho, sigma, n_members, n_samples, input_dim, eps, dW, gmu, grho, adam_m, adam_v, norm_scratch, active,
                               lr, beta1, beta2, adam_eps, weight_decay, max_norm, step, gnorm, (hipStream_t)stream);
}

int mcmc_train_grad(mcmc_ctx* ctx, int n_members, int n_samples, int row_blocks, int input_dim, const float* eps, const float* sigma,
                    const float* dW, float* gmu, float* grho, double* norm_scratch, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, 1)) return rc;
    MCMC_REQUIRE(row_blocks >= 1 && row_blocks <= 65535, "row_blocks must be in [1, 65535] (got %d)", row_blocks);
    MCMC_REQUIRE(eps && sigma && dW && gmu && grho && norm_scratch, "NULL pointer");
    return launch_train_grad(n_members, n_samples, row_blocks, input_dim, eps, sigma, dW, gmu, grho, norm_scratch, (hipStream_t)stream);
}

int mcmc_train_apply(mcmc_ctx* ctx, float* mu, float* rho, float* sigma, int n_members, int input_dim, const float* gmu,
                     const float* grho, float* adam_m, float* adam_v, double* norm_scratch, const uint8_t* active, float lr, float beta1,
                     float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, 1, input_dim, 1)) return rc;
    MCMC_REQUIRE(mu && rho && sigma && gmu && grho && adam_m && adam_v && norm_scratch, "NULL pointer");
    MCMC_REQUIRE(step >= 1, "step is 1-based");
    MCMC_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "betas must be in [0, 1)");
    return launch_train_apply(mu, rho, sigma, n_members, input_dim, gmu, grho, adam_m, adam_v, norm_scratch, true, active, lr, beta1,
                              beta2, adam_eps, weight_decay, max_norm, step, gnorm, (hipStream_t)stream);
}

int mcmc_train_step(mcmc_ctx* ctx, const mcmc_train_buffers* b, const mcmc_train_hyper* hp, const float* xb, int64_t x_member_stride,
                    const float* yb, int64_t y_member_stride, int64_t rows, int input_dim, int n_members, int n_samples, int noise_mode,
                    const float* eps_in, uint64_t seed, int mask_mode, const uint8_t* mask0, const uint8_t* mask1, const uint8_t* mask2,
                    uint64_t mask_seed, const uint8_t* active, float lr, float beta1, int64_t step, double* loss_acc,
                    mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_train(n_members, n_samples, input_dim, rows)) return rc;
    MCMC_REQUIRE(b && hp, "buffers / hyper is NULL");
    if (int rc = check_masks(mask_mode, mask0, mask1, mask2, hp->p_drop)) return rc;
    MCMC_REQUIRE(train_row_blocks(rows) == 1, "mcmc_train_step serves batches of one row block (rows < 2048, got %lld): use the "
                 "separate mcmc_train_* entry points", (long long)rows);
    MCMC_REQUIRE(noise_mode >= MCMC_NOISE_NONE && noise_mode <= MCMC_NOISE_PHILOX, "bad noise_mode %d", noise_mode);
    MCMC_REQUIRE(noise_mode != MCMC_NOISE_INJECTED || eps_in, "MCMC_NOISE_INJECTED needs eps_in");
    MCMC_REQUIRE(xb && yb, "xb / yb is NULL");
    MCMC_REQUIRE(x_member_stride >= rows * input_dim && y_member_stride >= rows, "member strides smaller than one member's batch");
    MCMC_REQUIRE(b->mu && b->rho && b->sigma && b->eps && b->W && b->WT && b->dW && b->gmu && b->grho && b->adam_m && b->adam_v &&
                 b->norm_scratch && b->Y && b->delta && b->loss, "NULL pointer in mcmc_train_buffers");
    MCMC_REQUIRE(step >= 1, "step is 1-based");
    MCMC_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && hp->beta2 >= 0.0f && hp->beta2 < 1.0f, "betas must be in [0, 1)");
    // small batches keep the forward pass's activations for the backward kernel when the caller provides room for them
    float* act = (b->act && b->act_floats >= (int64_t)n_members * n_samples * train_act_floats(rows)) ? b->act : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = launch_train_sample(b->mu, b->sigma, n_members, n_samples, input_dim, noise_mode, eps_in, seed, b->eps, b->W, b->WT, s))
        return rc;
    if (int rc = launch_train_forward(xb, x_member_stride, rows, input_dim, n_members, n_samples, b->WT, mask_mode, mask0, mask1, mask2,
                                      mask_seed, hp->p_drop, b->Y, act, s))
        return rc;
    // the loss (train_loss_kernel's arithmetic) runs in the prologue of the backward kernel: one launch fewer
    const TrainLossFuse fuse{b->Y, yb, y_member_stride, hp->alpha, hp->beta, b->loss, loss_acc};
    if (int rc = launch_train_backward(xb, x_member_stride, rows, input_dim, n_members, n_samples, b->W, b->WT, mask_mode, mask0, mask1,
                                       mask2, mask_seed, hp->p_drop, b->delta, 1, b->dW, act, &fuse, s))
        return rc;
    return launch_train_update(b->mu, b->rho, b->sigma, n_members, n_samples, input_dim, b->eps, b->dW, b->gmu, b->grho, b->adam_m,
                               b->adam_v, b->norm_scratch, active, lr, beta1, hp->beta2, hp->adam_eps, hp->weight_decay, hp->max_norm,
                               step, b->gnorm, s);
}

// ---- TFT (row f4) ---------------------------------------------------------------------------------------
static int check_tft(int n_windows, int seq_len, int num_features) {
    MCMC_REQUIRE(num_features >= 1 && num_features <= kMaxInputDim, "num_features must be in [1, %d] (got %d)", kMaxInputDim, num_features);
    MCMC_REQUIRE(seq_len >= 1 && seq_len <= 32, "seq_len must be in [1, 32] (got %d)", seq_len);
    MCMC_REQUIRE(n_windows >= 0, "n_windows < 0");
    return MCMC_OK;
}

static int check_tft_masks(int mask_mode, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, float p_drop) {
    MCMC_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "mask_mode must be 0 (eval), 1 (host) or 2 (philox)");
    MCMC_REQUIRE(mask_mode != 1 || (a && b && c && d), "mask_mode 1 needs all four masks");
    MCMC_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "p_drop must be in [0, 1)");
    return MCMC_OK;
}

int64_t mcmc_tft_param_count(int num_features) {
    if (num_features < 1 || num_features > kMaxInputDim) return set_error(MCMC_EINVAL, "num_features must be in [1, %d]", kMaxInputDim);
    return tft_params(num_features);
}

int mcmc_tft_transpose(mcmc_ctx* ctx, const float* P, float* PT, int num_features, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_tft(0, 1, num_features)) return rc;
    MCMC_REQUIRE(P && PT && P != PT, "P / PT is NULL or aliased");
    return launch_tft_transpose(P, PT, num_features, (hipStream_t)stream);
}

int64_t mcmc_tft_act_floats(void) { return tft_act_floats(); }

static int tft_forward_impl(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len, int num_features, const float* P,
                     const float* PT, int mask_mode, const uint8_t* mask_feature, const uint8_t* mask_temporal,
                     const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed, float p_drop, float* pred, float* act,
                     mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_tft(n_windows, seq_len, num_features)) return rc;
    if (int rc = check_tft_masks(mask_mode, mask_feature, mask_temporal, mask_attention, mask_final, p_drop)) return rc;
    if (n_windows == 0) return MCMC_OK;
    MCMC_REQUIRE(X && start && P && PT && pred, "NULL pointer");
    MCMC_REQUIRE(n_rows >= seq_len, "X has %lld rows, fewer than seq_len %d", (long long)n_rows, seq_len);
    return launch_tft_forward(X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal, mask_attention,
                              mask_final, seed, p_drop, pred, act, (hipStream_t)stream);
}

int mcmc_tft_forward(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len, int num_features,
                     const float* P, const float* PT, int mask_mode, const uint8_t* mask_feature, const uint8_t* mask_temporal,
                     const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed, float p_drop, float* pred,
                     mcmc_stream_t stream) {
    return tft_forward_impl(ctx, X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal,
                            mask_attention, mask_final, seed, p_drop, pred, nullptr, stream);
}

int mcmc_tft_forward_save(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len,
                          int num_features, const float* P, const float* PT, int mask_mode, const uint8_t* mask_feature,
                          const uint8_t* mask_temporal, const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                          float p_drop, float* pred, float* act, mcmc_stream_t stream) {
    MCMC_REQUIRE(act || n_windows == 0, "act is NULL");
    return tft_forward_impl(ctx, X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal,
                            mask_attention, mask_final, seed, p_drop, pred, act, stream);
}

static int tft_backward_impl(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len, int num_features, const float* P,
                      const float* PT, int mask_mode, const uint8_t* mask_feature, const uint8_t* mask_temporal,
                      const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed, float p_drop, const float* delta, float* dP,
                      const float* act, const TftLossFuse* fuse, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_tft(n_windows, seq_len, num_features)) return rc;
    if (int rc = check_tft_masks(mask_mode, mask_feature, mask_temporal, mask_attention, mask_final, p_drop)) return rc;
    if (n_windows == 0) return MCMC_OK;
    MCMC_REQUIRE(X && start && P && PT && (delta || fuse) && dP, "NULL pointer");
    MCMC_REQUIRE(n_rows >= seq_len, "X has %lld rows, fewer than seq_len %d", (long long)n_rows, seq_len);
    return launch_tft_backward(X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal, mask_attention,
                               mask_final, seed, p_drop, delta, dP, act, fuse, (hipStream_t)stream);
}

int mcmc_tft_backward(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len, int num_features,
                      const float* P, const float* PT, int mask_mode, const uint8_t* mask_feature, const uint8_t* mask_temporal,
                      const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed, float p_drop, const float* delta, float* dP,
                      mcmc_stream_t stream) {
    return tft_backward_impl(ctx, X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal,
                             mask_attention, mask_final, seed, p_drop, delta, dP, nullptr, nullptr, stream);
}

int mcmc_tft_backward_saved(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len,
                            int num_features, const float* P, const float* PT, int mask_mode, const uint8_t* mask_feature,
                            const uint8_t* mask_temporal, const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                            float p_drop, const float* delta, float* dP, const float* act, mcmc_stream_t stream) {
    MCMC_REQUIRE(act || n_windows == 0, "act is NULL");
    return tft_backward_impl(ctx, X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal,
                             mask_attention, mask_final, seed, p_drop, delta, dP, act, nullptr, stream);
}

int mcmc_tft_backward_loss(mcmc_ctx* ctx, const float* X, int64_t n_rows, const int32_t* start, int n_windows, int seq_len,
                           int num_features, const float* P, const float* PT, int mask_mode, const uint8_t* mask_feature,
                           const uint8_t* mask_temporal, const uint8_t* mask_attention, const uint8_t* mask_final, uint64_t seed,
                           float p_drop, const float* pred, const float* yb, float alpha, float beta, float* loss, double* loss_acc,
                           float* dP, const float* act, mcmc_stream_t stream) {
    MCMC_REQUIRE((pred && yb && loss) || n_windows == 0, "pred / yb / loss is NULL");
    const TftLossFuse fuse{pred, yb, alpha, beta, loss, loss_acc};
    return tft_backward_impl(ctx, X, n_rows, start, n_windows, seq_len, num_features, P, PT, mask_mode, mask_feature, mask_temporal,
                             mask_attention, mask_final, seed, p_drop, nullptr, dP, act, &fuse, stream);
}

int mcmc_flat_norm_blocks(void) { return tft_flat_blocks(); }

int mcmc_flat_adamw(mcmc_ctx* ctx, float* param, int64_t n, const float* dP, int n_partials, float* grad, float* adam_m, float* adam_v,
                    double* norm_scratch, float lr, float beta1, float beta2, float adam_eps, float weight_decay, float max_norm,
                    int64_t step, float* gnorm, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    MCMC_REQUIRE(n >= 1 && n_partials >= 1, "n and n_partials must be >= 1");
    MCMC_REQUIRE(param && dP && grad && adam_m && adam_v && norm_scratch, "NULL pointer");
    MCMC_REQUIRE(step >= 1, "step is 1-based");
    MCMC_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "betas must be in [0, 1)");
    return launch_flat_update(param, n, dP, n_partials, grad, adam_m, adam_v, norm_scratch, lr, beta1, beta2, adam_eps, weight_decay,
                              max_norm, step, gnorm, nullptr, 0, (hipStream_t)stream);
}

int mcmc_tft_adamw(mcmc_ctx* ctx, float* P, float* PT, int num_features, const float* dP, int n_partials, float* grad, float* adam_m,
                   float* adam_v, double* norm_scratch, float lr, float beta1, float beta2, float adam_eps, float weight_decay,
                   float max_norm, int64_t step, float* gnorm, mcmc_stream_t stream) {
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = check_tft(n_partials, 1, num_features)) return rc;
    MCMC_REQUIRE(n_partials >= 1, "n_partials must be >= 1");
    MCMC_REQUIRE(P && PT && dP && grad && adam_m && adam_v && norm_scratch, "NULL pointer");
    MCMC_REQUIRE(step >= 1, "step is 1-based");
    MCMC_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "betas must be in [0, 1)");
    return launch_flat_update(P, mcmc_tft_param_count(num_features), dP, n_partials, grad, adam_m, adam_v, norm_scratch, lr, beta1, beta2,
                              adam_eps, weight_decay, max_norm, step, gnorm, PT, num_features, (hipStream_t)stream);
}

}  // extern "C"
