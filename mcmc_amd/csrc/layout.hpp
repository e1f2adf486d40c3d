// Memory layouts of the MC predictive path (host + device).
//
// Network: input_dim -> 256 -> 128 -> 64 -> 1 (/root/reference/models/bayesian_ensemble.py:53,61-70).
//
// FLAT layout (the reference's own, one member): per layer W [out,in] row-major then b [out],
// layers in forward order = the order the reference draws noise in (bayesian_ensemble.py:43-44).
// mu[M,P], rho[M,P], injected eps[M*S,P] all use it.
//
// WPACK layout (kernel-private, one (member,sample) pair of EFFECTIVE weights
// W = mu + eps*exp(rho)): MFMA A-operand fragments, already in the order the lanes read them.
// Both MFMA shapes used (f32 32x32x2, bf16 32x32x16) leave output feature
//     f(i,h) = (i&3) + 8*(i>>2) + 4*h          (i = accumulator register 0..15, h = lane>>5)
// of a 32-feature tile in register i of lane half h, with the batch row on lane&31.  The next
// layer consumes those registers directly as its B operand, so its weights are stored with the
// k index permuted to match (see kperm_*).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MCMC_HD __host__ __device__
#else
#define MCMC_HD
#endif

namespace mcmc {

constexpr int H1 = 256, H2 = 128, H3 = 64;
constexpr int kMaxInputDim = 24;
constexpr int DT_F32 = 0, DT_BF16 = 1, DT_BF16_FAST = 2;

MCMC_HD inline int64_t flat_w_off(int F, int layer) {
    switch (layer) {
        case 0: return 0;
        case 1: return (int64_t)F * H1 + H1;
        case 2: return (int64_t)F * H1 + H1 + H1 * H2 + H2;
        default: return (int64_t)F * H1 + H1 + H1 * H2 + H2 + H2 * H3 + H3;
    }
}
MCMC_HD inline int layer_in(int F, int layer) { return layer == 0 ? F : layer == 1 ? H1 : layer == 2 ? H2 : H3; }
MCMC_HD inline int layer_out(int layer) { return layer == 0 ? H1 : layer == 1 ? H2 : layer == 2 ? H3 : 1; }
MCMC_HD inline int64_t flat_b_off(int F, int layer) { return flat_w_off(F, layer) + (int64_t)layer_in(F, layer) * layer_out(layer); }
MCMC_HD inline int64_t flat_param_count(int F) { return flat_b_off(F, 3) + 1; }

// feature held by accumulator register i of lane half h (within a 32-feature tile)
MCMC_HD inline int acc_feature(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- input packing ------------------------------------------------------------------
// XPACK: [rows_pad][K0] row-major, K0 = padded input width; column F holds 1.0 (carries the
// layer-0 bias through the MFMA; MCMC_DT_BF16_FAST: column F+1 too when K0 has room, for the bias residual),
// further columns are 0.  rows_pad = rows rounded up to 32.
MCMC_HD inline int k0_pad(int F, int dtype) { return dtype != DT_F32 ? ((F + 1 + 15) / 16) * 16 : ((F + 1 + 1) / 2) * 2; }
MCMC_HD inline int64_t rows_pad(int64_t rows) { return (rows + 31) / 32 * 32; }

// ---- WPACK, fp32 (MFMA f32 32x32x2: one k-step = 2 k values, k = lane half) -----------
// L0 : [8 mo][K0/2 step][64 lane]   lane(r,h) = W0[32mo+r][2*step+h]      (bias in column F)
// L1 : [4 mo][128 step][64 lane]    step = 16*mt+i, lane(r,h) = W1[32mo+r][32mt+f(i,h)]
// L2 : [2 mo][64 step][64 lane]     same rule
// B1 : [4 mo][2 h][16 i] = b1[32mo+f(i,h)];  B2 : [2 mo][2 h][16 i];  W3 : [2 mo][2 h][16 i];  B3 : [4] (b3,0,0,0)
struct WPackF32 {
    int K0;
    int64_t l0, l1, l2, b1, b2, w3, b3, total;  // float offsets
    MCMC_HD explicit WPackF32(int F) {
        K0 = k0_pad(F, DT_F32);
        l0 = 0;
        l1 = l0 + (int64_t)8 * (K0 / 2) * 64;
        b1 = l1 + (int64_t)4 * 128 * 64;
        l2 = b1 + 4 * 2 * 16;
        b2 = l2 + (int64_t)2 * 64 * 64;
        w3 = b2 + 2 * 2 * 16;
        b3 = w3 + 2 * 2 * 16;
        total = ((b3 + 4 + 63) / 64) * 64;  // pair stride, multiple of 256 B
    }
};

constexpr float kLog2e = 1.4426950408889634f;

// ---- WPACK, bf16 exact ELU (MFMA bf16 32x32x16: one k-step = 16 k values, lane half h holds 8) --
// log2(e)-scaled domain: the kernel keeps a = log2e * z so that ELU needs one v_exp_f32:
//   phi(a) = max(a, log2e * 2^min(a,0) - log2e) = log2e * elu(z).
// L0 (and its bias column) is scaled by log2e, b1/b2 by log2e, w3 by 1/log2e; W1, W2 unscaled.
// L0 : [8 mo][K0/16 step][64 lane][8 j] bf16   lane(r,h) elem j = log2e*W0[32mo+r][16*step+8h+j]
// L1 : [4 mo][16 step][64 lane][8 j]           step = 2*mt+s, elem j = W1[32mo+r][32mt+16s+8(j>>2)+4h+(j&3)]
// L2 : [2 mo][8 step][64 lane][8 j]            same rule
// then fp32: B1 [4][2][16], B2 [2][2][16], W3 [2][2][16], B3 [4]
struct WPackBF16 {
    int K0;
    int64_t l0, l1, l2, b1, b2, w3, b3, total;  // BYTE offsets
    MCMC_HD explicit WPackBF16(int F) {
        K0 = k0_pad(F, DT_BF16);
        l0 = 0;
        l1 = l0 + (int64_t)8 * (K0 / 16) * 64 * 16;
        l2 = l1 + (int64_t)4 * 16 * 64 * 16;
        b1 = l2 + (int64_t)2 * 8 * 64 * 16;
        b2 = b1 + 4 * 2 * 16 * 4;
        w3 = b2 + 2 * 2 * 16 * 4;
        b3 = w3 + 2 * 2 * 16 * 4;
        total = ((b3 + 16 + 255) / 256) * 256;  // pair stride, multiple of 256 B
    }
};

// ---- WPACK, bf16 fast ELU (same fragment rules) --
// Shifted log2(e) domain: with c = log2e the accumulators of a hidden layer hold u = c*z + c and the
// activation handed to the next layer is H = c*(elu(z) + 1) > 0 (forward_bf16.hip), so
//   L0 : [8 mo][K0/16 step][64 lane][8 j] bf16   lane(r,h) elem j = c*W0[32mo+r][16*step+8h+j]; input column F
//        (x = 1) carries bf16(v), v = c*b0 + c, and - when K0 has room - column F+1 (x = 1) the residual v - bf16(v)
//   L1 : [4 mo][16 step][64 lane][8 j]           step = 2*mt+s, elem j = W1[32mo+r][32mt+16s+8(j>>2)+4h+(j&3)]
//   L2 : [2 mo][8 step][64 lane][8 j]            same rule
//   L3 : [4 step][64 lane][8 j]                  lanes r = 0 only: elem j = w3[k]/c (the other 31 output rows are zero)
//   then fp32: B1 [4][2][16] = c*(b1 + 1 - rowsum(bf16 W1)), B2 [2][2][16] likewise, B3 [4] = (b3 - c*sum(bf16(w3/c)), 0, 0, 0)
// (the row sums are taken over the bf16 values the MFMA multiplies by, so the constant part of H cancels exactly).
// centring of the exponent-field exponential in bf16 mantissa units (forward_bf16.hip, tools/elu_approx_study.py)
constexpr float kFieldShift = -7.5f;
struct WPackBF16F {
    int K0;
    bool bias_residual;  // input column F+1 exists and carries the bf16 residual of the layer-0 bias
    int64_t l0, l1, l2, l3, b1, b2, b3, total;  // BYTE offsets
    MCMC_HD explicit WPackBF16F(int F) {
        K0 = k0_pad(F, DT_BF16);
        bias_residual = F + 2 <= K0;
        l0 = 0;
        l1 = l0 + (int64_t)8 * (K0 / 16) * 64 * 16;
        l2 = l1 + (int64_t)4 * 16 * 64 * 16;
        l3 = l2 + (int64_t)2 * 8 * 64 * 16;
        b1 = l3 + (int64_t)4 * 64 * 16;
        b2 = b1 + 4 * 2 * 16 * 4;
        b3 = b2 + 2 * 2 * 16 * 4;
        total = ((b3 + 16 + 255) / 256) * 256;  // pair stride, multiple of 256 B
    }
};

MCMC_HD inline int kperm_bf16(int mt, int s, int h, int j) { return 32 * mt + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

}  // namespace mcmc
