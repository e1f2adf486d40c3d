// K0 pack_x and K1 sample_weights: build the MFMA operand images of the MC predictive path.
//
// K1 restates BayesianLinear.forward's sampling line
//     weight = weight_mu + randn_like(weight_mu) * exp(weight_rho)      (and the same for bias)
// (/root/reference/models/bayesian_ensemble.py:43-44) for a whole range of (member,sample)
// pairs at once, writing the result straight into the fragment order the forward kernels read
// (layout.hpp).  363 M weights per config-3 predict = ~1 ms; the kernel is a gather by design.
#include "common.hpp"
#include "philox.hpp"

namespace mcmc {

namespace {

constexpr float kLn2 = 0.6931471805599453f;

struct NoiseSrc {
    const float* mu;   // [M,P]
    const float* rho;  // [M,P]
    const float* eps;  // [M*S,P] or null
    uint64_t seed;
    const uint64_t* seed_dev;  // when set, the Philox key is read from device memory (graph replay with a new seed)
    int mode;
    int64_t P;
};

// effective weight for flat parameter idx of member m / global pair g
__device__ __forceinline__ float effective_weight(const NoiseSrc& ns, int m, uint32_t g, int64_t idx) {
    const float mu = ns.mu[(int64_t)m * ns.P + idx];
    if (ns.mode == MCMC_NOISE_NONE) return mu;
    const float e = ns.mode == MCMC_NOISE_INJECTED ? ns.eps[(int64_t)g * ns.P + idx] : philox_normal(ns.seed_dev ? *ns.seed_dev : ns.seed, g, (uint32_t)idx);
    const float sigma = expf(ns.rho[(int64_t)m * ns.P + idx]);
    // two roundings like torch's mul then add (no fma contraction)
    return __fadd_rn(mu, __fmul_rn(e, sigma));
}

// effective weights of the four consecutive flat parameters idx0 .. idx0+3 (idx0 % 4 == 0: one Philox block)
__device__ __forceinline__ void effective_weight4(const NoiseSrc& ns, int m, uint32_t g, int64_t idx0, float (&w)[4]) {
    const float* mu = ns.mu + (int64_t)m * ns.P + idx0;
    if (ns.mode == MCMC_NOISE_NONE) {
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = mu[t];
        return;
    }
    float e[4];
    if (ns.mode == MCMC_NOISE_INJECTED) {
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = ns.eps[(int64_t)g * ns.P + idx0 + t];
    } else {
        philox_normal4(ns.seed_dev ? *ns.seed_dev : ns.seed, g, (uint32_t)(idx0 >> 2), e);
    }
    const float* rho = ns.rho + (int64_t)m * ns.P + idx0;
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = __fadd_rn(mu[t], __fmul_rn(e[t], expf(rho[t])));
}

// flat index of the layer-0 A element (out, k): W0 column, the bias column at k == F, or -1 (zero pad)
__device__ __forceinline__ int64_t l0_flat(int F, int out, int k) {
    if (k < F) return (int64_t)out * F + k;
    if (k == F) return flat_b_off(F, 0) + out;
    return -1;
}

__global__ __launch_bounds__(256) void sample_weights_f32_kernel(NoiseSrc ns, int M, int S, int F, int64_t pair_begin,
                                                                 float* __restrict__ wpack) {
    const WPackF32 L(F);
    const int c = blockIdx.y;
    const int64_t q = pair_begin + c;
    const int m = (int)(q % M);
    const uint32_t g = (uint32_t)((int64_t)m * S + q / M);
    float* out = wpack + (int64_t)c * L.total;
    const int ks0 = L.K0 / 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < L.total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t idx = -1;
        if (e < L.l1) {
            const int lane = (int)(e & 63), step = (int)((e >> 6) % ks0), mo = (int)((e >> 6) / ks0);
            idx = l0_flat(F, 32 * mo + (lane & 31), 2 * step + (lane >> 5));
        } else if (e < L.b1) {
            continue;  // layer-1 weights: quad loop below
        } else if (e < L.l2) {
            const int l = (int)(e - L.b1);
            idx = flat_b_off(F, 1) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
        } else if (e < L.b2) {
            continue;  // layer-2 weights: quad loop below
        } else if (e < L.w3) {
            const int l = (int)(e - L.b2);
            idx = flat_b_off(F, 2) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
        } else if (e < L.b3) {
            const int l = (int)(e - L.w3);
            idx = flat_w_off(F, 3) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
        } else if (e == L.b3) {
            idx = flat_b_off(F, 3);
        }
        out[e] = idx >= 0 ? effective_weight(ns, m, g, idx) : 0.0f;
    }
    // Layers 1-2: four consecutive k-steps of one lane are four consecutive k of one weight row
    // (acc_feature(4a+t, h) = t + 8a + 4h) = one Philox block; the four stores sit 64 floats apart.
    const int64_t q1 = (L.b1 - L.l1) / 4, q2 = (L.b2 - L.l2) / 4;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < q1 + q2; u += (int64_t)gridDim.x * blockDim.x) {
        const bool first = u < q1;
        const int64_t v = first ? u : u - q1;
        const int lane = (int)(v & 63);
        const int sq = first ? (int)((v >> 6) & 31) : (int)((v >> 6) & 15);   // step quad: 128 / 64 k-steps per row tile
        const int mo = first ? (int)(v >> 11) : (int)(v >> 10);
        const int step = 4 * sq;
        const int k = 32 * (step >> 4) + acc_feature(step & 15, lane >> 5);
        const int64_t idx0 = first ? flat_w_off(F, 1) + (int64_t)(32 * mo + (lane & 31)) * H1 + k
                                   : flat_w_off(F, 2) + (int64_t)(32 * mo + (lane & 31)) * H2 + k;
        float w[4];
        effective_weight4(ns, m, g, idx0, w);
        float* dst = out + (first ? L.l1 : L.l2) + (((int64_t)mo * (first ? 128 : 64) + step) << 6) + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t << 6] = w[t];
    }
}

__device__ __forceinline__ uint32_t bf16_bits(float x) {
    const __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, b);
}

__global__ __launch_bounds__(256) void sample_weights_bf16_kernel(NoiseSrc ns, int M, int S, int F, int64_t pair_begin,
                                                                  uint32_t* __restrict__ wpack) {
    const WPackBF16 L(F);
    const int c = blockIdx.y;
    const int64_t q = pair_begin + c;
    const int m = (int)(q % M);
    const uint32_t g = (uint32_t)((int64_t)m * S + q / M);
    uint32_t* out = wpack + (int64_t)c * (L.total / 4);
    const int ks0 = L.K0 / 16;
    const int64_t ndw = L.total / 4;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t byte = d * 4;
        uint32_t word = 0;
        if (byte >= L.l1 && byte < L.b1) {
            // layers 1-2: elements j..j+3 (j = 0 or 4) of a lane's fragment are four consecutive k of one
            // weight row = one Philox block; the even dword of each 8-byte unit produces both dwords
            if (d & 1) continue;
            const int64_t el = byte / 2;
            int64_t idx0;
            if (byte < L.l2) {
                const int64_t l = el - L.l1 / 2;
                const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 15), mo = (int)(l >> 13);
                idx0 = flat_w_off(F, 1) + (int64_t)(32 * mo + (lane & 31)) * H1 + kperm_bf16(step >> 1, step & 1, lane >> 5, j);
            } else {
                const int64_t l = el - L.l2 / 2;
                const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 7), mo = (int)(l >> 12);
                idx0 = flat_w_off(F, 2) + (int64_t)(32 * mo + (lane & 31)) * H2 + kperm_bf16(step >> 1, step & 1, lane >> 5, j);
            }
            float w[4];
            effective_weight4(ns, m, g, idx0, w);
            out[d] = bf16_bits(w[0]) | (bf16_bits(w[1]) << 16);
            out[d + 1] = bf16_bits(w[2]) | (bf16_bits(w[3]) << 16);
            continue;
        }
        if (byte < L.b1) {
            // two consecutive bf16 elements j, j+1 of one lane's 8-element fragment
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t el = byte / 2 + u;
                int64_t idx;
                float scale = 1.0f;
                if (byte < L.l1) {
                    const int j = (int)(el & 7), lane = (int)((el >> 3) & 63);
                    const int step = (int)((el >> 9) % ks0), mo = (int)((el >> 9) / ks0);
                    idx = l0_flat(F, 32 * mo + (lane & 31), 16 * step + 8 * (lane >> 5) + j);
                    scale = kLog2e;
                } else if (byte < L.l2) {
                    const int64_t l = el - L.l1 / 2;
                    const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 15), mo = (int)(l >> 13);
                    const int k = kperm_bf16(step >> 1, step & 1, lane >> 5, j);
                    idx = flat_w_off(F, 1) + (int64_t)(32 * mo + (lane & 31)) * H1 + k;
                } else {
                    const int64_t l = el - L.l2 / 2;
                    const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 7), mo = (int)(l >> 12);
                    const int k = kperm_bf16(step >> 1, step & 1, lane >> 5, j);
                    idx = flat_w_off(F, 2) + (int64_t)(32 * mo + (lane & 31)) * H2 + k;
                }
                v[u] = idx >= 0 ? effective_weight(ns, m, g, idx) * scale : 0.0f;
            }
            word = bf16_bits(v[0]) | (bf16_bits(v[1]) << 16);
        } else {
            int64_t idx = -1;
            float scale = 1.0f;
            if (byte < L.b2) {
                const int l = (int)((byte - L.b1) / 4);
                idx = flat_b_off(F, 1) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
                scale = kLog2e;
            } else if (byte < L.w3) {
                const int l = (int)((byte - L.b2) / 4);
                idx = flat_b_off(F, 2) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
                scale = kLog2e;
            } else if (byte < L.b3) {
                const int l = (int)((byte - L.w3) / 4);
                idx = flat_w_off(F, 3) + 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
                scale = kLn2;
            } else if (byte == L.b3) {
                idx = flat_b_off(F, 3);
            }
            const float v = idx >= 0 ? effective_weight(ns, m, g, idx) * scale : 0.0f;
            word = __float_as_uint(v);
        }
        out[d] = word;
    }
}

__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

// Second pass over the MCMC_DT_BF16_FAST image: every bias absorbs c * (row sum of the bf16 weights the MFMA multiplies
// by), read back from the image itself (L2-resident: written a moment ago) - 193 rows per pair, one wave each, so no
// Philox draw is made twice.  Row `o` of layer 1 / 2 lives in fragment lane (o & 31, h) of output tile o >> 5: 16 bytes
// per k-step and lane half.  Fixed summation order: bitwise reproducible.
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

__global__ __launch_bounds__(256) void bias_fixup_bf16_fast_kernel(int F, unsigned char* __restrict__ wpack) {
    const WPackBF16F L(F);
    unsigned char* img = wpack + (int64_t)blockIdx.y * L.total;
    const int lane = threadIdx.x & 63;
    const int row = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));  // 0..127 layer 1, 128..191 layer 2, 192 layer 3
    if (row > H2 + H3) return;
    float s = 0.0f;
    float* dst;
    float scale_b = kLog2e;
    if (row < H2 + H3) {
        const bool first = row < H2;
        const int o = first ? row : row - H2;
        const int steps = first ? 16 : 8;                       // k-steps of 16
        const int64_t base = (first ? L.l1 : L.l2) + (int64_t)(o >> 5) * steps * 1024;
        // lane -> (k-step, lane half): 2 * steps fragments of 16 bytes
        if (lane < 2 * steps) {
            const int step = lane >> 1, h = lane & 1;
            const uint4 v = *reinterpret_cast<const uint4*>(img + base + (int64_t)step * 1024 + ((o & 31) + 32 * h) * 16);
            s = ((bf16_lo(v.x) + bf16_hi(v.x)) + (bf16_lo(v.y) + bf16_hi(v.y))) + ((bf16_lo(v.z) + bf16_hi(v.z)) + (bf16_lo(v.w) + bf16_hi(v.w)));
        }
        const int l = 32 * (o >> 5);
        // position of output o in the [mo][2 h][16 i] bias block: o = 32 mo + f(i, h)
        const int f = o & 31, hh = (f >> 2) & 1, i = (f & 3) + 4 * (f >> 3);
        dst = reinterpret_cast<float*>(img + (first ? L.b1 : L.b2)) + l + 16 * hh + i;
    } else {
        // layer 3: output row 0 = lanes 0 and 32 of each of the 4 k-steps
        if (lane < 8) {
            const int step = lane >> 1, h = lane & 1;
            const uint4 v = *reinterpret_cast<const uint4*>(img + L.l3 + (int64_t)step * 1024 + (32 * h) * 16);
            s = ((bf16_lo(v.x) + bf16_hi(v.x)) + (bf16_lo(v.y) + bf16_hi(v.y))) + ((bf16_lo(v.z) + bf16_hi(v.z)) + (bf16_lo(v.w) + bf16_hi(v.w)));
        }
        dst = reinterpret_cast<float*>(img + L.b3);
        scale_b = 1.0f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) *dst = scale_b * (*dst) - kLog2e * s;
}

// The MCMC_DT_BF16_FAST image in the shifted log2e domain (layout.hpp, forward_bf16.hip): hidden accumulators hold
// c*z + c and activations c*(elu + 1), so every bias also carries c*(1 - rowsum of its bf16 weight row).
__global__ __launch_bounds__(256) void sample_weights_bf16_fast_kernel(NoiseSrc ns, int M, int S, int F, int64_t pair_begin,
                                                                       uint32_t* __restrict__ wpack) {
    const WPackBF16F L(F);
    const int c = blockIdx.y;
    const int64_t q = pair_begin + c;
    const int m = (int)(q % M);
    const uint32_t g = (uint32_t)((int64_t)m * S + q / M);
    uint32_t* out = wpack + (int64_t)c * (L.total / 4);
    const int ks0 = L.K0 / 16;
    const int64_t ndw = L.total / 4;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t byte = d * 4;
        uint32_t word = 0;
        if (byte < L.l1) {
            // layer 0: two consecutive bf16 elements j, j+1 of one lane's 8-element fragment
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t el = byte / 2 + u;
                const int j = (int)(el & 7), lane = (int)((el >> 3) & 63);
                const int step = (int)((el >> 9) % ks0), mo = (int)((el >> 9) / ks0);
                const int o = 32 * mo + (lane & 31), k = 16 * step + 8 * (lane >> 5) + j;
                if (k < F) {
                    v[u] = effective_weight(ns, m, g, (int64_t)o * F + k) * kLog2e;
                } else if (k == F || (k == F + 1 && L.bias_residual)) {
                    const float b = fmaf(effective_weight(ns, m, g, flat_b_off(F, 0) + o), kLog2e, kLog2e);
                    v[u] = k == F ? b : b - bf16_round(b);
                } else {
                    v[u] = 0.0f;
                }
            }
            word = bf16_bits(v[0]) | (bf16_bits(v[1]) << 16);
        } else if (byte < L.l3) {
            // layers 1-2: elements j..j+3 (j = 0 or 4) of a lane's fragment are four consecutive k of one
            // weight row = one Philox block; the even dword of each 8-byte unit produces both dwords
            if (d & 1) continue;
            const int64_t el = byte / 2;
            int64_t idx0;
            if (byte < L.l2) {
                const int64_t l = el - L.l1 / 2;
                const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 15), mo = (int)(l >> 13);
                idx0 = flat_w_off(F, 1) + (int64_t)(32 * mo + (lane & 31)) * H1 + kperm_bf16(step >> 1, step & 1, lane >> 5, j);
            } else {
                const int64_t l = el - L.l2 / 2;
                const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)((l >> 9) & 7), mo = (int)(l >> 12);
                idx0 = flat_w_off(F, 2) + (int64_t)(32 * mo + (lane & 31)) * H2 + kperm_bf16(step >> 1, step & 1, lane >> 5, j);
            }
            float w[4];
            effective_weight4(ns, m, g, idx0, w);
            out[d] = bf16_bits(w[0]) | (bf16_bits(w[1]) << 16);
            out[d + 1] = bf16_bits(w[2]) | (bf16_bits(w[3]) << 16);
            continue;
        } else if (byte < L.b1) {
            // layer 3: output row 0 of a 32-row A tile (lanes 0 and 32), the other rows stay zero
            const int64_t l = (byte - L.l3) / 2;
            const int j = (int)(l & 7), lane = (int)((l >> 3) & 63), step = (int)(l >> 9);
            if ((lane & 31) == 0) {
                const int64_t idx = flat_w_off(F, 3) + kperm_bf16(step >> 1, step & 1, lane >> 5, j);
                word = bf16_bits(effective_weight(ns, m, g, idx) * kLn2) | (bf16_bits(effective_weight(ns, m, g, idx + 1) * kLn2) << 16);
            }
        } else if (byte < L.b3) {
            // b + 1 here; the fix-up kernel below subtracts the row sum of the bf16 weights and scales by c
            const bool first = byte < L.b2;
            const int l = (int)((byte - (first ? L.b1 : L.b2)) / 4);
            const int o = 32 * (l >> 5) + acc_feature(l & 15, (l >> 4) & 1);
            word = __float_as_uint(effective_weight(ns, m, g, flat_b_off(F, first ? 1 : 2) + o) + 1.0f);
        } else if (byte == L.b3) {
            word = __float_as_uint(effective_weight(ns, m, g, flat_b_off(F, 3)));
        }
        out[d] = word;
    }
}

__global__ __launch_bounds__(256) void philox_eps_kernel(uint64_t seed, int64_t g_begin, int64_t P, float* __restrict__ eps) {
    const uint32_t g = (uint32_t)(g_begin + blockIdx.y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
        eps[(int64_t)blockIdx.y * P + i] = philox_normal(seed, g, (uint32_t)i);
}

__global__ __launch_bounds__(256) void pack_x_f32_kernel(const float* __restrict__ X, int64_t n_rows, int F, int K0,
                                                         float* __restrict__ xp) {
    const int64_t total = rows_pad(n_rows) * K0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / K0;
        const int k = (int)(e % K0);
        float v = 0.0f;
        if (row < n_rows) v = k < F ? X[row * F + k] : (k == F ? 1.0f : 0.0f);
        xp[e] = v;
    }
}

__global__ __launch_bounds__(256) void pack_x_bf16_kernel(const float* __restrict__ X, int64_t n_rows, int F, int K0, bool residual_col,
                                                          unsigned short* __restrict__ xp) {
    const int64_t total = rows_pad(n_rows) * K0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / K0;
        const int k = (int)(e % K0);
        float v = 0.0f;
        if (row < n_rows) v = k < F ? X[row * F + k] : ((k == F || (k == F + 1 && residual_col)) ? 1.0f : 0.0f);
        xp[e] = (unsigned short)bf16_bits(v);
    }
}

inline int grid_for(int64_t n, int cap) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

int launch_pack_x(const float* X, int64_t n_rows, int F, int dtype, void* xpack, hipStream_t s) {
    const int K0 = k0_pad(F, dtype);
    const int64_t total = rows_pad(n_rows) * K0;
    if (dtype == DT_F32)
        hipLaunchKernelGGL(pack_x_f32_kernel, dim3(grid_for(total, 4096)), dim3(256), 0, s, X, n_rows, F, K0, (float*)xpack);
    else
        hipLaunchKernelGGL(pack_x_bf16_kernel, dim3(grid_for(total, 4096)), dim3(256), 0, s, X, n_rows, F, K0,
                           dtype == DT_BF16_FAST && WPackBF16F(F).bias_residual, (unsigned short*)xpack);
    return check_launch("pack_x");
}

int launch_sample_weights(const float* mu, const float* rho, int M, int S, int F, int64_t pair_begin, int pair_count,
                          int noise_mode, const float* eps, uint64_t seed, const uint64_t* seed_dev, int dtype, void* wpack,
                          hipStream_t s) {
    NoiseSrc ns{mu, rho, eps, seed, seed_dev, noise_mode, flat_param_count(F)};
    // gridDim.y is limited to 65535: walk the pair range in slabs
    for (int done = 0; done < pair_count;) {
        const int n = pair_count - done > 32768 ? 32768 : pair_count - done;
        if (dtype == DT_F32) {
            const WPackF32 L(F);
            hipLaunchKernelGGL(sample_weights_f32_kernel, dim3(grid_for(L.total, 64), n), dim3(256), 0, s, ns, M, S, F,
                               pair_begin + done, (float*)wpack + (int64_t)done * L.total);
        } else if (dtype == DT_BF16) {
            const WPackBF16 L(F);
            hipLaunchKernelGGL(sample_weights_bf16_kernel, dim3(grid_for(L.total / 4, 32), n), dim3(256), 0, s, ns, M, S, F,
                               pair_begin + done, (uint32_t*)wpack + (int64_t)done * (L.total / 4));
        } else {
            const WPackBF16F L(F);
            hipLaunchKernelGGL(sample_weights_bf16_fast_kernel, dim3(grid_for(L.total / 4, 32), n), dim3(256), 0, s, ns, M, S, F,
                               pair_begin + done, (uint32_t*)wpack + (int64_t)done * (L.total / 4));
            hipLaunchKernelGGL(bias_fixup_bf16_fast_kernel, dim3((H2 + H3 + 1 + 3) / 4, n), dim3(256), 0, s, F,
                               (unsigned char*)wpack + (int64_t)done * L.total);
        }
        done += n;
    }
    return check_launch("sample_weights");
}

int launch_philox_eps(uint64_t seed, int64_t g_begin, int pair_count, int F, float* eps, hipStream_t s) {
    const int64_t P = flat_param_count(F);
    for (int done = 0; done < pair_count;) {
        const int n = pair_count - done > 32768 ? 32768 : pair_count - done;
        hipLaunchKernelGGL(philox_eps_kernel, dim3(grid_for(P, 64), n), dim3(256), 0, s, seed, g_begin + done, P,
                           eps + (int64_t)done * P);
        done += n;
    }
    return check_launch("philox_eps");
}

}  // namespace mcmc
