// Host-sanitizer driver of the C ABI ("-fsanitize=address host build of the C-ABI shim").
// TEST INFRASTRUCTURE: built by `make asan` together with a sanitizer build of mcmc_amd/csrc (AddressSanitizer +
// UndefinedBehaviorSanitizer on the host code, -DMCMC_DRY_CTX) and run by tests/test_host_sanitizer.py on the CPU box.
// It drives every extern "C" entry point of include/mcmc_hip.h through its argument validation - a NULL context, then a
// device-less "dry" context with one bad argument after the other, then the empty-input fast paths - and the layout queries
// through their whole range.  No call here reaches a kernel launch: every one must return before it.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../include/mcmc_hip.h"

extern "C" int mcmc_ctx_create_dry(mcmc_ctx** out);  // sanitizer build only (capi.cpp, MCMC_DRY_CTX)

static int g_checks = 0, g_failed = 0;

#define EXPECT(expr, want)                                                                                          \
    do {                                                                                                            \
        const long long got_ = (long long)(expr);                                                                   \
        ++g_checks;                                                                                                 \
        if (got_ != (long long)(want)) {                                                                            \
            ++g_failed;                                                                                             \
            fprintf(stderr, "%s:%d: %s = %lld, expected %lld (%s)\n", __FILE__, __LINE__, #expr, got_, (long long)(want), \
                    mcmc_last_error());                                                                             \
        }                                                                                                           \
    } while (0)
// a failed validation must leave a message behind
#define EXPECT_EINVAL(expr)                                   \
    do {                                                      \
        EXPECT(expr, MCMC_EINVAL);                            \
        ++g_checks;                                           \
        if (!mcmc_last_error()[0]) {                          \
            ++g_failed;                                       \
            fprintf(stderr, "%s:%d: empty error text\n", __FILE__, __LINE__); \
        }                                                     \
    } while (0)

alignas(256) static unsigned char g_buf[8192];  // stands for device memory: validated (NULL, alignment), never dereferenced

int main() {
    float* f = reinterpret_cast<float*>(g_buf);
    double* d = reinterpret_cast<double*>(g_buf);
    void* v = g_buf;
    uint8_t* u8 = g_buf;
    uint16_t* u16 = reinterpret_cast<uint16_t*>(g_buf);
    int32_t* i32 = reinterpret_cast<int32_t*>(g_buf);
    int64_t* i64 = reinterpret_cast<int64_t*>(g_buf);
    uint64_t* u64 = reinterpret_cast<uint64_t*>(g_buf);
    float* f_odd = reinterpret_cast<float*>(g_buf + 4);      // 4-byte aligned only
    uint8_t* u8_odd = g_buf + 1;
    const mcmc_stream_t s = nullptr;

    // ---- host-only queries, over their whole range ----
    EXPECT(mcmc_abi_version() >= 21, 1);
    EXPECT(mcmc_last_error() != nullptr, 1);
    for (int F = -2; F <= 40; ++F) {
        const bool ok = F >= 1 && F <= 24;
        const int64_t P = mcmc_param_count(F);
        EXPECT(P, ok ? (int64_t)F * 256 + 256 + 256 * 128 + 128 + 128 * 64 + 64 + 64 + 1 : MCMC_EINVAL);
        for (int dt = -1; dt <= 3; ++dt) {
            const bool dok = dt >= MCMC_DT_F32 && dt <= MCMC_DT_BF16_FAST;
            const int64_t w = mcmc_wpack_bytes(F, dt);
            EXPECT(w > 0 && w % 256 == 0, ok && dok);
            if (!(ok && dok)) EXPECT(w, MCMC_EINVAL);
            for (int64_t rows : {(int64_t)-1, (int64_t)0, (int64_t)1, (int64_t)365, (int64_t)3650000}) {
                const int64_t x = mcmc_xpack_bytes(rows, F, dt);
                if (ok && dok && rows >= 0) EXPECT(x >= rows * (F + 1) * (dt == MCMC_DT_F32 ? 4 : 2), 1);
                else EXPECT(x, MCMC_EINVAL);
            }
        }
        const int64_t tp = mcmc_tft_param_count(F);
        EXPECT(tp > 0, ok);
    }
    EXPECT(mcmc_param_count(11), 44289);
    EXPECT(mcmc_tft_param_count(11), 67585);
    EXPECT(mcmc_loss_workspace_bytes() > 0 && mcmc_loss_workspace_bytes() % 8 == 0, 1);
    EXPECT(mcmc_train_norm_blocks() >= 1, 1);
    EXPECT(mcmc_flat_norm_blocks() >= 1, 1);
    EXPECT(mcmc_tft_act_floats() > 0, 1);
    EXPECT(mcmc_train_row_blocks(-5) >= 0, 1);
    EXPECT(mcmc_train_row_blocks(32), 1);
    EXPECT(mcmc_train_row_blocks(2047), 1);
    EXPECT(mcmc_train_row_blocks(65536) > 1 && mcmc_train_row_blocks(65536) <= 64, 1);
    EXPECT(mcmc_train_row_blocks((int64_t)1 << 40) <= 64, 1);
    EXPECT(mcmc_train_act_bytes_bf16(0), 0);
    EXPECT(mcmc_train_act_bytes_bf16(65536) > 0 && mcmc_train_act_bytes_bf16(65536) % 16 == 0, 1);
    EXPECT(mcmc_train_act_floats(0), 0);
    EXPECT(mcmc_train_act_floats(-3), 0);
    EXPECT(mcmc_train_act_floats(32) > 0, 1);
    EXPECT(mcmc_train_act_floats(65536) >= mcmc_train_act_floats(32), 1);

    // ---- contexts ----
    mcmc_ctx* none = nullptr;
    EXPECT_EINVAL(mcmc_ctx_create(0, nullptr));
    EXPECT(mcmc_ctx_create(-1, &none) < 0 && none == nullptr, 1);         // no device / out of range: an error code, no context
    EXPECT(mcmc_ctx_create(1 << 20, &none) < 0 && none == nullptr, 1);
    EXPECT(mcmc_ctx_destroy(nullptr), MCMC_ESTATE);
    mcmc_ctx* c = nullptr;
    EXPECT(mcmc_ctx_create_dry(&c), MCMC_OK);
    if (!c) return 2;
    mcmc_ctx* dead = nullptr;
    EXPECT(mcmc_ctx_create_dry(&dead), MCMC_OK);

    // every context-taking entry point refuses a NULL context first
#define ESTATE(call) EXPECT(call, MCMC_ESTATE)
    mcmc_ctx* z = nullptr;
    mcmc_train_buffers tb;
    memset(&tb, 0, sizeof tb);
    mcmc_train_hyper hp = {0.1f, 2.0f, 0.5f, 0.999f, 1e-8f, 0.01f, 1.0f};
    const double q1[1] = {0.5};
    ESTATE(mcmc_pack_x(z, f, 4, 11, 0, v, s));
    ESTATE(mcmc_sample_weights(z, f, f, 1, 1, 11, 0, 1, 0, nullptr, 0, 0, v, s));
    ESTATE(mcmc_sample_weights_seedptr(z, f, f, 1, 1, 11, 0, 1, u64, 0, v, s));
    ESTATE(mcmc_philox_eps(z, 1, 0, 1, 11, f, s));
    ESTATE(mcmc_forward(z, 0, v, 4, 11, v, 1, nullptr, nullptr, nullptr, 1.0f, f, 4, s));
    ESTATE(mcmc_forward_dropout(z, 0, v, 4, 11, v, 1, 1, 0, 1, 0, 0.1f, 0, f, 4, s));
    ESTATE(mcmc_reduce_update(z, f, 4, 1, 4, d, d, nullptr, nullptr, nullptr, 0, s));
    ESTATE(mcmc_reduce_merge(z, 1, 4, d, d, nullptr, 0, d, d, nullptr, nullptr, s));
    ESTATE(mcmc_reduce_set_edges(z, 4, 1, d, d, 128, 6.0f, f, f, s));
    ESTATE(mcmc_reduce_finalize(z, 4, 2, d, d, nullptr, nullptr, nullptr, 0, nullptr, 0, f, f, nullptr, s));
    ESTATE(mcmc_loss_stats(z, f, f, 4, d, s));
    ESTATE(mcmc_loss_forward(z, f, f, 4, 4, d, 2.0f, 0.5f, d, s));
    ESTATE(mcmc_loss_backward(z, f, f, 4, 4, d, d, 2.0f, 0.5f, 1.0f, f, s));
    ESTATE(mcmc_loss_fused(z, f, f, 4, 2.0f, 0.5f, 1.0f, f, f, v, s));
    ESTATE(mcmc_loss_fused_static(z, f, f, 4, 2.0f, 0.5f, 1.0f, f, f, s));
    ESTATE(mcmc_train_sigma(z, f, f, 4, s));
    ESTATE(mcmc_train_sample(z, f, f, 1, 1, 11, 0, nullptr, 0, f, f, f, s));
    ESTATE(mcmc_train_forward(z, f, 4, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, s));
    ESTATE(mcmc_train_forward_save(z, f, 4, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    ESTATE(mcmc_train_loss(z, f, f, 1, 1, 4, 2.0f, 0.5f, f, f, s));
    ESTATE(mcmc_train_loss_acc(z, f, f, 1, 1, 4, 2.0f, 0.5f, f, f, d, s));
    ESTATE(mcmc_train_backward(z, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    ESTATE(mcmc_train_backward_blocks(z, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 1, f, s));
    ESTATE(mcmc_train_backward_saved(z, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 1, f, f, s));
    ESTATE(mcmc_train_forward_bf16(z, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, 4, f, nullptr, s));
    ESTATE(mcmc_train_backward_bf16(z, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 4, f, nullptr, s));
    ESTATE(mcmc_train_update(z, f, f, f, 1, 1, 11, f, f, f, f, f, f, d, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    ESTATE(mcmc_train_grad(z, 1, 1, 1, 11, f, f, f, f, f, d, s));
    ESTATE(mcmc_train_apply(z, f, f, f, 1, 11, f, f, f, f, d, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    ESTATE(mcmc_train_step(z, &tb, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));
    ESTATE(mcmc_train_early_stop(z, d, d, i64, u8, 1, 25, s));
    ESTATE(mcmc_tft_transpose(z, f, f + 64, 11, s));
    ESTATE(mcmc_tft_forward(z, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, s));
    ESTATE(mcmc_tft_forward_save(z, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    ESTATE(mcmc_tft_backward(z, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    ESTATE(mcmc_tft_backward_saved(z, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, f, s));
    ESTATE(mcmc_tft_backward_loss(z, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, 2.0f, 0.5f, f, nullptr, f, nullptr, s));
    ESTATE(mcmc_flat_adamw(z, f, 4, f, 1, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    ESTATE(mcmc_tft_adamw(z, f, f, 11, f, 1, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    // ... and a destroyed one
    EXPECT(mcmc_ctx_destroy(dead), MCMC_OK);

    // ---- piece 1 ----
    EXPECT_EINVAL(mcmc_pack_x(c, f, 4, 0, 0, v, s));            // input_dim
    EXPECT_EINVAL(mcmc_pack_x(c, f, 4, 25, 1, v, s));
    EXPECT_EINVAL(mcmc_pack_x(c, f, 4, 11, 3, v, s));           // dtype
    EXPECT_EINVAL(mcmc_pack_x(c, f, -1, 11, 0, v, s));
    EXPECT_EINVAL(mcmc_pack_x(c, nullptr, 4, 11, 0, v, s));
    EXPECT_EINVAL(mcmc_pack_x(c, f, 4, 11, 0, f_odd, s));       // alignment
    EXPECT(mcmc_pack_x(c, nullptr, 0, 11, 0, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 0, 1, 11, 0, 1, 0, nullptr, 0, 0, v, s));
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 2, 3, 11, 4, 5, 2, nullptr, 0, 0, v, s));          // pair range
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 65536, 65536, 11, 0, 1, 2, nullptr, 0, 0, v, s));  // M*S > 2^31-1
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 2, 3, 11, 0, 1, 7, nullptr, 0, 0, v, s));          // noise mode
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 2, 3, 11, 0, 1, MCMC_NOISE_INJECTED, nullptr, 0, 0, v, s));
    EXPECT_EINVAL(mcmc_sample_weights(c, nullptr, f, 2, 3, 11, 0, 1, 2, nullptr, 0, 0, v, s));
    EXPECT_EINVAL(mcmc_sample_weights(c, f, f, 2, 3, 11, 0, 1, 2, nullptr, 0, 0, f_odd, s));
    EXPECT(mcmc_sample_weights(c, nullptr, nullptr, 2, 3, 11, 6, 0, 2, nullptr, 0, 0, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_sample_weights_seedptr(c, f, f, 2, 3, 11, 0, 1, nullptr, 0, v, s));
    EXPECT_EINVAL(mcmc_philox_eps(c, 1, -1, 1, 11, f, s));
    EXPECT_EINVAL(mcmc_philox_eps(c, 1, 0, 1, 11, nullptr, s));
    EXPECT_EINVAL(mcmc_philox_eps(c, 1, 0x7FFFFFFFLL, 1, 11, f, s));
    EXPECT(mcmc_philox_eps(c, 1, 0, 0, 11, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 100, 11, v, 1, nullptr, nullptr, nullptr, 1.0f, f, 50, s));   // ldY < rows
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 6, 11, v, 1, nullptr, nullptr, nullptr, 1.0f, f, 6, s));      // ldY % 4
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 4, 11, v, 1, u8, nullptr, u8, 1.0f, f, 4, s));                // two of three masks
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 4, 11, v, 1, u8, u8_odd, u8, 1.0f, f, 4, s));                 // mask alignment
    EXPECT_EINVAL(mcmc_forward(c, 0, f_odd, 4, 11, v, 1, nullptr, nullptr, nullptr, 1.0f, f, 4, s));
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 4, 11, v, -1, nullptr, nullptr, nullptr, 1.0f, f, 4, s));
    EXPECT_EINVAL(mcmc_forward(c, 0, v, 4, 11, nullptr, 1, nullptr, nullptr, nullptr, 1.0f, f, 4, s));
    EXPECT_EINVAL(mcmc_forward(c, MCMC_DT_BF16_FAST, v, 4, 11, v, 1, u8, u8, u8, 1.0f, f, 4, s));     // no mask stage (launcher-side check)
    EXPECT(mcmc_forward(c, 0, nullptr, 0, 11, nullptr, 0, nullptr, nullptr, nullptr, 1.0f, nullptr, 0, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_forward_dropout(c, 1, v, 4, 11, v, 2, 3, 5, 2, 0, 0.1f, 0, f, 4, s));          // pair range
    EXPECT_EINVAL(mcmc_forward_dropout(c, 1, v, 4, 11, v, 2, 3, 0, 1, 0, 1.0f, 0, f, 4, s));          // p_drop
    EXPECT_EINVAL(mcmc_forward_dropout(c, 1, v, 4, 11, v, 2, 3, 0, 1, 0, 0.1f, -1, f, 4, s));         // row_begin
    EXPECT_EINVAL(mcmc_forward_dropout(c, 1, v, 4, 11, v, 2, 3, 0, 1, 0, 0.1f, 0xFFFFFFFFLL, f, 4, s));  // row index past the 32-bit counter
    EXPECT_EINVAL(mcmc_forward_dropout(c, MCMC_DT_BF16_FAST, v, 64, 11, v, 2, 3, 0, 1, 0, 0.1f, 0, f, 64, s));
    EXPECT(mcmc_forward_dropout(c, 1, nullptr, 0, 11, nullptr, 2, 3, 0, 0, 0, 0.1f, 0, nullptr, 0, s), MCMC_OK);

    // ---- piece 2 ----
    EXPECT_EINVAL(mcmc_reduce_update(c, f, 4, 1, 8, d, d, nullptr, nullptr, nullptr, 0, s));          // ldY < rows
    EXPECT_EINVAL(mcmc_reduce_update(c, f, 64, 4, 64, d, d, f, f, u16, 100, s));                      // nbins
    EXPECT_EINVAL(mcmc_reduce_update(c, f, 64, 4, 64, d, d, nullptr, f, u16, 128, s));                // hist without edges
    EXPECT_EINVAL(mcmc_reduce_update(c, f, 64, 4, 64, d, d, f, f, reinterpret_cast<uint16_t*>(g_buf + 2), 128, s));
    EXPECT_EINVAL(mcmc_reduce_update(c, nullptr, 64, 4, 64, d, d, nullptr, nullptr, nullptr, 0, s));
    EXPECT_EINVAL(mcmc_reduce_update(c, f, 64, -1, 64, d, d, nullptr, nullptr, nullptr, 0, s));
    EXPECT(mcmc_reduce_update(c, nullptr, 0, 0, 64, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_reduce_merge(c, 0, 4, d, d, nullptr, 0, d, d, nullptr, nullptr, s));
    EXPECT_EINVAL(mcmc_reduce_merge(c, 2, 4, d, d, u16, 128, d, d, nullptr, nullptr, s));             // hist on one side only
    EXPECT_EINVAL(mcmc_reduce_merge(c, 2, 4, d, d, u16, 100, d, d, u16, nullptr, s));
    EXPECT_EINVAL(mcmc_reduce_merge(c, 2, 4, d, d, reinterpret_cast<uint16_t*>(g_buf + 8), 128, d, d, u16, nullptr, s));
    EXPECT_EINVAL(mcmc_reduce_merge(c, 2, 4, nullptr, d, nullptr, 0, d, d, nullptr, nullptr, s));
    EXPECT(mcmc_reduce_merge(c, 2, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_reduce_set_edges(c, 4, 1, d, d, 100, 6.0f, f, f, s));
    EXPECT_EINVAL(mcmc_reduce_set_edges(c, 4, 0, d, d, 128, 6.0f, f, f, s));
    EXPECT_EINVAL(mcmc_reduce_set_edges(c, 4, 1, d, d, 128, 0.0f, f, f, s));
    EXPECT_EINVAL(mcmc_reduce_set_edges(c, 4, 1, d, d, 128, 6.0f, nullptr, f, s));
    EXPECT(mcmc_reduce_set_edges(c, 0, 1, nullptr, nullptr, 128, 6.0f, nullptr, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 4, 0, d, d, nullptr, nullptr, nullptr, 0, nullptr, 0, f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 4, 2, d, d, nullptr, nullptr, nullptr, 0, nullptr, 9, f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 4, 2, d, d, nullptr, nullptr, nullptr, 128, q1, 1, f, f, f, s));       // quantiles without hist
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 64, 70000, d, d, u16, f, f, 128, q1, 1, f, f, f, s));                  // uint16 counts
    const double qbad[1] = {1.5};
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 4, 2, d, d, u16, f, f, 128, qbad, 1, f, f, f, s));
    EXPECT_EINVAL(mcmc_reduce_finalize(c, 4, 2, d, d, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, f, nullptr, s));
    EXPECT(mcmc_reduce_finalize(c, 0, 2, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, s), MCMC_OK);

    // ---- piece 3 ----
    EXPECT_EINVAL(mcmc_loss_stats(c, f, f, -1, d, s));
    EXPECT_EINVAL(mcmc_loss_stats(c, nullptr, f, 4, d, s));
    EXPECT_EINVAL(mcmc_loss_stats(c, f_odd, f, 4, d, s));
    EXPECT_EINVAL(mcmc_loss_stats(c, f, f, 4, nullptr, s));
    EXPECT(mcmc_loss_stats(c, nullptr, nullptr, 0, d, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_loss_forward(c, f, f, 8, 4, d, 2.0f, 0.5f, d, s));                             // n > n_total
    EXPECT_EINVAL(mcmc_loss_forward(c, f, f, 4, 4, nullptr, 2.0f, 0.5f, d, s));
    EXPECT(mcmc_loss_forward(c, nullptr, nullptr, 0, 4, d, 2.0f, 0.5f, d, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_loss_backward(c, f, f, 4, 4, d, d, 2.0f, 0.5f, 1.0f, f_odd, s));
    EXPECT_EINVAL(mcmc_loss_backward(c, f, f, 4, 4, d, d, 2.0f, 0.5f, 1.0f, nullptr, s));
    EXPECT_EINVAL(mcmc_loss_backward(c, f, f, 4, 0, d, d, 2.0f, 0.5f, 1.0f, f, s));
    EXPECT(mcmc_loss_backward(c, nullptr, nullptr, 0, 4, d, d, 2.0f, 0.5f, 1.0f, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_loss_fused(c, f, f, 0, 2.0f, 0.5f, 1.0f, f, f, v, s));
    EXPECT_EINVAL(mcmc_loss_fused(c, f, f, 4, 2.0f, 0.5f, 1.0f, nullptr, f, v, s));
    EXPECT_EINVAL(mcmc_loss_fused(c, f, f, 4, 2.0f, 0.5f, 1.0f, f, f_odd, v, s));
    EXPECT_EINVAL(mcmc_loss_fused(c, f, f, 4, 2.0f, 0.5f, 1.0f, f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_loss_fused(c, f, f, 4, 2.0f, 0.5f, 1.0f, f, f, g_buf + 4, s));
    EXPECT_EINVAL(mcmc_loss_fused_static(c, f, f, 0, 2.0f, 0.5f, 1.0f, f, f, s));
    EXPECT_EINVAL(mcmc_loss_fused_static(c, f, f, MCMC_LOSS_STATIC_MAX_N + 1, 2.0f, 0.5f, 1.0f, f, f, s));
    EXPECT_EINVAL(mcmc_loss_fused_static(c, f, nullptr, 4, 2.0f, 0.5f, 1.0f, f, f, s));
    EXPECT_EINVAL(mcmc_loss_fused_static(c, f, f, 4, 2.0f, 0.5f, 1.0f, nullptr, f, s));

    // ---- training step (row f1) ----
    EXPECT_EINVAL(mcmc_train_sigma(c, f, f, -1, s));
    EXPECT_EINVAL(mcmc_train_sigma(c, nullptr, f, 4, s));
    EXPECT(mcmc_train_sigma(c, nullptr, nullptr, 0, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 0, 1, 11, 0, nullptr, 0, f, f, f, s));
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 256, 256, 11, 0, nullptr, 0, f, f, f, s));               // M*S > 65535
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 1, 1, 30, 0, nullptr, 0, f, f, f, s));
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 1, 1, 11, 5, nullptr, 0, f, f, f, s));
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 1, 1, 11, MCMC_NOISE_INJECTED, nullptr, 0, f, f, f, s));
    EXPECT_EINVAL(mcmc_train_sample(c, f, f, 1, 1, 11, 0, nullptr, 0, f, nullptr, f, s));
    EXPECT_EINVAL(mcmc_train_forward(c, f, 0, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, s));   // rows
    EXPECT_EINVAL(mcmc_train_forward(c, f, 4, 11, 1, 1, f, 3, nullptr, nullptr, nullptr, 0, 0.1f, f, s));   // mask mode
    EXPECT_EINVAL(mcmc_train_forward(c, f, 4, 11, 1, 1, f, 1, u8, nullptr, u8, 0, 0.1f, f, s));
    EXPECT_EINVAL(mcmc_train_forward(c, f, 4, 11, 1, 1, f, 1, u8, u8_odd, u8, 0, 0.1f, f, s));
    EXPECT_EINVAL(mcmc_train_forward(c, f, 4, 11, 1, 1, f, 2, nullptr, nullptr, nullptr, 0, -0.5f, f, s));
    EXPECT_EINVAL(mcmc_train_forward(c, nullptr, 4, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, s));
    EXPECT_EINVAL(mcmc_train_forward_save(c, f, 4, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_forward_save(c, f, 4, 11, 0, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    EXPECT_EINVAL(mcmc_train_loss(c, f, f, 1, 1, 0, 2.0f, 0.5f, f, f, s));
    EXPECT_EINVAL(mcmc_train_loss(c, f, nullptr, 1, 1, 4, 2.0f, 0.5f, f, f, s));
    EXPECT_EINVAL(mcmc_train_loss_acc(c, f, f, 1, 1, 4, 2.0f, 0.5f, f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_loss_acc(c, f, f, 1, 0, 4, 2.0f, 0.5f, f, f, d, s));
    EXPECT_EINVAL(mcmc_train_backward(c, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, f, s));
    EXPECT_EINVAL(mcmc_train_backward(c, f, 4, 0, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, f, s));
    EXPECT_EINVAL(mcmc_train_backward_blocks(c, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 0, f, s));
    EXPECT_EINVAL(mcmc_train_backward_blocks(c, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 70000, f, s));
    EXPECT_EINVAL(mcmc_train_backward_saved(c, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 1, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_backward_saved(c, f, 4, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 3, f, f, s));   // not the library's split
    EXPECT_EINVAL(mcmc_train_forward_bf16(c, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, 0, f, nullptr, s));       // row_blocks
    EXPECT_EINVAL(mcmc_train_forward_bf16(c, f, 4096, 11, 1, 1, f_odd, 0, nullptr, nullptr, nullptr, 0, 0.1f, 4, f, nullptr, s));   // W alignment
    EXPECT_EINVAL(mcmc_train_forward_bf16(c, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, 4, f, f_odd, s));   // act alignment
    EXPECT_EINVAL(mcmc_train_forward_bf16(c, f, 4096, 11, 1, 1, f, 1, u8, u8, nullptr, 0, 0.1f, 4, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_forward_bf16(c, f, 4096, 25, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, 4, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_backward_bf16(c, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, 4, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_backward_bf16(c, f, 0, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 4, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_backward_bf16(c, f, 4096, 11, 1, 1, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 70000, f, nullptr, s));
    EXPECT_EINVAL(mcmc_train_update(c, f, f, f, 1, 1, 11, f, f, f, f, f, f, d, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 0, nullptr, s));   // step
    EXPECT_EINVAL(mcmc_train_update(c, f, f, f, 1, 1, 11, f, f, f, f, f, f, d, nullptr, 1e-3f, 1.0f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));   // beta1
    EXPECT_EINVAL(mcmc_train_update(c, f, f, f, 1, 1, 11, f, f, f, f, f, f, nullptr, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_train_grad(c, 1, 1, 0, 11, f, f, f, f, f, d, s));
    EXPECT_EINVAL(mcmc_train_grad(c, 1, 1, 1, 11, f, f, nullptr, f, f, d, s));
    EXPECT_EINVAL(mcmc_train_apply(c, f, f, f, 1, 11, f, f, f, f, d, nullptr, 1e-3f, 0.9f, 1.5f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_train_apply(c, f, f, f, 0, 11, f, f, f, f, d, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_train_step(c, nullptr, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_train_step(c, &tb, nullptr, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));   // NULL buffers
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 44, f, 4, 4096, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));   // more than one row block
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 10, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));     // stride < batch
    tb.mu = tb.rho = tb.sigma = tb.eps = tb.W = tb.WT = tb.dW = tb.gmu = tb.grho = tb.adam_m = tb.adam_v = tb.Y = tb.delta = tb.loss = f;
    tb.norm_scratch = d;
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 0, nullptr, s));     // step
    hp.beta2 = 1.0f;
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));     // beta2
    hp.beta2 = 0.999f;
    hp.p_drop = 1.0f;
    EXPECT_EINVAL(mcmc_train_step(c, &tb, &hp, f, 44, f, 4, 4, 11, 1, 1, 0, nullptr, 0, 2, nullptr, nullptr, nullptr, 0, nullptr, 1e-3f, 0.9f, 1, nullptr, s));     // p_drop
    EXPECT_EINVAL(mcmc_train_early_stop(c, d, d, i64, u8, 0, 25, s));
    EXPECT_EINVAL(mcmc_train_early_stop(c, d, nullptr, i64, u8, 1, 25, s));

    // ---- TFT (row f4) ----
    EXPECT_EINVAL(mcmc_tft_transpose(c, f, f, 11, s));           // aliased
    EXPECT_EINVAL(mcmc_tft_transpose(c, f, f + 64, 0, s));
    EXPECT_EINVAL(mcmc_tft_forward(c, f, 40, i32, 1, 33, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, s));   // seq_len
    EXPECT_EINVAL(mcmc_tft_forward(c, f, 40, i32, -1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, s));
    EXPECT_EINVAL(mcmc_tft_forward(c, f, 10, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, s));   // fewer rows than a window
    EXPECT_EINVAL(mcmc_tft_forward(c, f, 40, i32, 1, 30, 11, f, f, 1, u8, u8, nullptr, u8, 0, 0.1f, f, s));                  // three of four masks
    EXPECT_EINVAL(mcmc_tft_forward(c, f, 40, nullptr, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, s));
    EXPECT(mcmc_tft_forward(c, nullptr, 0, nullptr, 0, 30, 11, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_tft_forward_save(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, nullptr, s));
    EXPECT(mcmc_tft_forward_save(c, nullptr, 0, nullptr, 0, 30, 11, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_tft_backward(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, f, s));   // no delta
    EXPECT_EINVAL(mcmc_tft_backward(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 1.0f, f, f, s));
    EXPECT_EINVAL(mcmc_tft_backward_saved(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, nullptr, s));
    EXPECT_EINVAL(mcmc_tft_backward_saved(c, f, 40, i32, 1, 30, 25, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, f, s));
    EXPECT_EINVAL(mcmc_tft_backward_loss(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, f, 2.0f, 0.5f, f, nullptr, f, nullptr, s));
    EXPECT_EINVAL(mcmc_tft_backward_loss(c, f, 40, i32, 1, 30, 11, f, f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, f, f, 2.0f, 0.5f, f, nullptr, nullptr, nullptr, s));
    EXPECT(mcmc_tft_backward_loss(c, nullptr, 0, nullptr, 0, 30, 11, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.1f, nullptr, nullptr, 2.0f, 0.5f, nullptr, nullptr, nullptr, nullptr, s), MCMC_OK);
    EXPECT_EINVAL(mcmc_flat_adamw(c, f, 0, f, 1, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_flat_adamw(c, f, 4, f, 1, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 0, nullptr, s));
    EXPECT_EINVAL(mcmc_flat_adamw(c, f, 4, f, 1, f, nullptr, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_tft_adamw(c, f, f, 11, f, 0, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_tft_adamw(c, f, f, 0, f, 1, f, f, f, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));
    EXPECT_EINVAL(mcmc_tft_adamw(c, f, f, 11, f, 1, f, f, f, d, 1e-3f, -0.1f, 0.999f, 1e-8f, 0.01f, 1.0f, 1, nullptr, s));

    // an error text longer than the thread-local buffer must be truncated, not overrun (exercised through the longest message)
    EXPECT_EINVAL(mcmc_train_backward_saved(c, f, (int64_t)1 << 40, 11, 1, 1, f, f, 0, nullptr, nullptr, nullptr, 0, 0.1f, f, 7, f, f, s));
    EXPECT(strlen(mcmc_last_error()) < 512, 1);

    EXPECT(mcmc_ctx_destroy(c), MCMC_OK);

    printf("abi_check: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
