// Row f1: one training step of the ensemble on the device.
//
// Restates the inner loop of GPUBayesianEnsemble.train
// (/root/reference/models/bayesian_ensemble.py:160-181) for ALL members at once instead of the
// reference's sequential member loop:
//   num_samples train-mode forwards per member (weight noise AND live dropout, :169-171),
//   pred_mean = mean over samples (:173-174), peak_weighted_loss (:176 -> utils.py:7-28),
//   backward through the reparameterisation (dmu = dW, drho = dW * eps * exp(rho)),
//   clip_grad_norm_(1.0) (:179), AdamW step (:180; lr / beta1 come from the host's OneCycleLR).
//
// The batch is 32 rows (:124), so every GEMM here is tiny: the step is launch- and latency-bound,
// not MFMA-bound.  One workgroup owns one (member, sample) pair and keeps all activations of a
// 32-row tile in LDS (141 KB); weights stream from L2 in two pre-sampled layouts (W [out][in] for
// the input gradients, WT [in][out] for the forward) so every global access is lane-contiguous.
// fp32 VALU arithmetic throughout - this path is about fidelity and launch count, not FLOP/s.
#include "common.hpp"
#include "philox.hpp"

namespace mcmc {

namespace {

constexpr int TR = 32;         // rows per LDS tile
constexpr int NT = 256;        // threads per workgroup
constexpr int XS = 28;         // LDS row stride of the input tile (>= kMaxInputDim, multiple of 4)

struct MaskSrc {
    const uint8_t* m0;  // host-supplied keep masks [pairs, rows, 256/128/64] or null
    const uint8_t* m1;
    const uint8_t* m2;
    uint64_t seed;      // Philox mask key
    int mode;           // 0 none, 1 host, 2 philox
    float keep_prob;    // 1 - p
    float scale;        // fp32(1/(1-p)), torch's dropout scale
};

// keep flag of activation element `feat` (0..447 over the three hidden layers) of (pair, row)
__device__ __forceinline__ bool keep_flag(const MaskSrc& ms, int pair, int64_t rows, int64_t row, int layer, int feat) {
    if (ms.mode == 0) return true;
    if (ms.mode == 1) {
        const uint8_t* m = layer == 0 ? ms.m0 : layer == 1 ? ms.m1 : ms.m2;
        const int width = layer == 0 ? H1 : layer == 1 ? H2 : H3;
        return m[((int64_t)pair * rows + row) * width + feat] != 0;
    }
    const int elem = (layer == 0 ? 0 : layer == 1 ? H1 : H1 + H2) + feat;
    // torch: mask = bernoulli(1-p), i.e. keep with probability keep_prob
    return philox_mask_uniform(ms.seed, (uint32_t)pair, (uint32_t)row, (uint32_t)elem) < ms.keep_prob;
}

// ---- dense layers on LDS tiles --------------------------------------------------------------
// z[r][o] = b[o] + sum_k hin[r][k] * WT[k][o];  then h = elu(z) * mask * scale stored in place.
// Masked elements are stored as -0.0f: arithmetic zero, and recognisable in the backward pass.
template <int IN, int OUT, int IN_STRIDE>
__device__ __forceinline__ void dense_forward(const float* __restrict__ hin, const float* __restrict__ WT,
                                              const float* __restrict__ bias, float* __restrict__ hout, const MaskSrc& ms,
                                              int pair, int64_t rows, int64_t row_base, int layer) {
    constexpr int G = NT / OUT, RG = TR / G;  // row groups, rows per thread
    const int o = threadIdx.x % OUT, g = threadIdx.x / OUT;
    float acc[RG];
    const float b = bias[o];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) acc[rr] = b;
    if constexpr (IN % 4 == 0) {
        for (int k = 0; k < IN; k += 4) {
            const float w0 = WT[(k + 0) * OUT + o], w1 = WT[(k + 1) * OUT + o], w2 = WT[(k + 2) * OUT + o], w3 = WT[(k + 3) * OUT + o];
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                const float4 h = *reinterpret_cast<const float4*>(hin + (g * RG + rr) * IN_STRIDE + k);
                acc[rr] = fmaf(h.x, w0, acc[rr]);
                acc[rr] = fmaf(h.y, w1, acc[rr]);
                acc[rr] = fmaf(h.z, w2, acc[rr]);
                acc[rr] = fmaf(h.w, w3, acc[rr]);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        const int r = g * RG + rr;
        const float z = acc[rr];
        float h = z > 0.0f ? z : expm1f(z);
        const bool live = row_base + r < rows;
        if (ms.mode != 0) h = (live && keep_flag(ms, pair, rows, row_base + r, layer, o)) ? h * ms.scale : -0.0f;
        hout[r * OUT + o] = live ? h : -0.0f;
    }
}

// layer 0: the input width is a runtime value (7..24), scalar k loop
template <int OUT>
__device__ __forceinline__ void dense_forward_in(const float* __restrict__ xs, int F, const float* __restrict__ WT,
                                                 const float* __restrict__ bias, float* __restrict__ hout, const MaskSrc& ms, int pair,
                                                 int64_t rows, int64_t row_base) {
    const int o = threadIdx.x;  // OUT == NT == 256
    float acc[TR];
    const float b = bias[o];
#pragma unroll
    for (int r = 0; r < TR; ++r) acc[r] = b;
    for (int k = 0; k < F; ++k) {
        const float w = WT[k * OUT + o];
#pragma unroll
        for (int r = 0; r < TR; ++r) acc[r] = fmaf(xs[r * XS + k], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const float z = acc[r];
        float h = z > 0.0f ? z : expm1f(z);
        const bool live = row_base + r < rows;
        if (ms.mode != 0) h = (live && keep_flag(ms, pair, rows, row_base + r, 0, o)) ? h * ms.scale : -0.0f;
        hout[r * OUT + o] = live ? h : -0.0f;
    }
}

// derivative factor d = elu'(z) * mask * scale recovered from the stored activation
__device__ __forceinline__ float dfactor(float h, float scale) {
    if (__float_as_uint(h) == 0x80000000u) return 0.0f;  // masked (or padding row)
    return h > 0.0f ? scale : h + scale;                 // z > 0: scale;  z <= 0: exp(z)*scale = h + scale
}

struct Lds {
    float* xs;   // [TR][XS]
    float* h1;   // [TR][256]
    float* h2;   // [TR][128]
    float* h3;   // [TR][64]
    float* ga;   // gradient scratch A: [TR][256]
    float* gb;   // gradient scratch B: [TR][128] + transposed [128][TR]
    float* gc;   // gradient scratch C: [TR][64] + transposed [64][TR]
    float* y;    // [TR]
};
constexpr int kLdsFloats = TR * XS + TR * (H1 + H2 + H3) + TR * H1 + 2 * TR * H2 + 2 * TR * H3 + TR;

__device__ __forceinline__ Lds carve(float* base) {
    Lds l;
    l.xs = base;
    l.h1 = l.xs + TR * XS;
    l.h2 = l.h1 + TR * H1;
    l.h3 = l.h2 + TR * H2;
    l.ga = l.h3 + TR * H3;
    l.gb = l.ga + TR * H1;
    l.gc = l.gb + 2 * TR * H2;
    l.y = l.gc + 2 * TR * H3;
    return l;
}

// forward of one 32-row tile of one pair; fills xs, h1, h2, h3, y
__device__ __forceinline__ void forward_tile(const Lds& l, const float* __restrict__ xb, int64_t rows, int F, int64_t row_base,
                                             const float* __restrict__ WT, const MaskSrc& ms, int pair) {
    for (int i = threadIdx.x; i < TR * XS; i += NT) {
        const int r = i / XS, k = i % XS;
        l.xs[i] = (row_base + r < rows && k < F) ? xb[(row_base + r) * F + k] : 0.0f;
    }
    __syncthreads();
    const float* wt0 = WT + flat_w_off(F, 0);
    dense_forward_in<H1>(l.xs, F, wt0, WT + flat_b_off(F, 0), l.h1, ms, pair, rows, row_base);
    __syncthreads();
    dense_forward<H1, H2, H1>(l.h1, WT + flat_w_off(F, 1), WT + flat_b_off(F, 1), l.h2, ms, pair, rows, row_base, 1);
    __syncthreads();
    dense_forward<H2, H3, H2>(l.h2, WT + flat_w_off(F, 2), WT + flat_b_off(F, 2), l.h3, ms, pair, rows, row_base, 2);
    __syncthreads();
    {  // layer 3: 64 -> 1; 8 threads per row
        const int r = threadIdx.x >> 3, part = threadIdx.x & 7;
        const float* w3 = WT + flat_w_off(F, 3);
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = fmaf(l.h3[r * H3 + part * 8 + k], w3[part * 8 + k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (part == 0) l.y[r] = a + WT[flat_b_off(F, 3)];
    }
    __syncthreads();
}

// T1: eps, effective weights in both layouts ------------------------------------------------------
__global__ __launch_bounds__(256) void train_sample_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, int S, int F,
                                                          int noise_mode, const float* __restrict__ eps_in, uint64_t seed,
                                                          float* __restrict__ eps, float* __restrict__ W, float* __restrict__ WT) {
    const int64_t P = flat_param_count(F);
    const int pair = blockIdx.y, m = pair / S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        float e = 0.0f;
        if (noise_mode == MCMC_NOISE_INJECTED) e = eps_in[(int64_t)pair * P + i];
        else if (noise_mode == MCMC_NOISE_PHILOX) e = philox_normal(seed, (uint32_t)pair, (uint32_t)i);
        const float w = noise_mode == MCMC_NOISE_NONE ? mu[m * P + i] : __fadd_rn(mu[m * P + i], __fmul_rn(e, sigma[m * P + i]));
        eps[(int64_t)pair * P + i] = e;
        W[(int64_t)pair * P + i] = w;
        // transposed position inside the same layer block
        int layer = 3;
        if (i < flat_w_off(F, 1)) layer = 0;
        else if (i < flat_w_off(F, 2)) layer = 1;
        else if (i < flat_w_off(F, 3)) layer = 2;
        const int64_t wo = flat_w_off(F, layer), bo = flat_b_off(F, layer);
        int64_t j = i;
        if (i < bo) {
            const int in = layer_in(F, layer), out = layer_out(layer);
            const int64_t l = i - wo;
            const int o = (int)(l / in), k = (int)(l % in);
            j = wo + (int64_t)k * out + o;
        }
        WT[(int64_t)pair * P + j] = w;
    }
}

// T2: train-mode forward, one workgroup per pair ---------------------------------------------------
__global__ __launch_bounds__(NT, 1) void train_forward_kernel(const float* __restrict__ xb, int64_t rows, int F, int S,
                                                              const float* __restrict__ WT, MaskSrc ms, float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S;
    const int64_t P = flat_param_count(F);
    for (int64_t row_base = 0; row_base < rows; row_base += TR) {
        forward_tile(l, xb + (int64_t)m * rows * F, rows, F, row_base, WT + (int64_t)pair * P, ms, pair);
        if (threadIdx.x < TR && row_base + threadIdx.x < rows) Y[(int64_t)pair * rows + row_base + threadIdx.x] = l.y[threadIdx.x];
        __syncthreads();
    }
}

// T3: pred_mean, peak-weighted loss and d loss / d pred (per sample), one workgroup per member -------
__global__ __launch_bounds__(NT) void train_loss_kernel(const float* __restrict__ Y, const float* __restrict__ yb, int S, int64_t rows,
                                                        float alpha, float beta, float* __restrict__ delta, float* __restrict__ loss) {
    __shared__ double red[4][NT / 64];
    __shared__ double tot[4];
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum4 = [&](double (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double x = v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0) red[i][wave] = x;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            double a = 0.0;
            for (int w = 0; w < NT / 64; ++w) a += red[threadIdx.x][w];
            tot[threadIdx.x] = a;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tot[i];
        __syncthreads();
    };
    auto pred = [&](int64_t r) {  // mean over samples, summed in sample order like torch.mean(0)
        float a = 0.0f;
        for (int s = 0; s < S; ++s) a += Y[((int64_t)m * S + s) * rows + r];
        return a / (float)S;
    };
    const double N = (double)rows;
    double v[4] = {0, 0, 0, 0};
    for (int64_t r = threadIdx.x; r < rows; r += NT) {
        const double t = (double)yb[(int64_t)m * rows + r];
        v[0] += t;
        v[1] += t * t;
        v[2] += (double)fabsf(pred(r) - yb[(int64_t)m * rows + r]);
    }
    block_sum4(v);
    const double mean_t = v[0] / N;
    const double var = (v[1] - v[0] * v[0] / N) / (N - 1.0);
    const double sd = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
    const float thr = (float)(mean_t + (double)beta * sd), mean_e = (float)(v[2] / N), am1 = alpha - 1.0f;
    double w2[4] = {0, 0, 0, 0};
    for (int64_t r = threadIdx.x; r < rows; r += NT) {
        const float t = yb[(int64_t)m * rows + r], d = pred(r) - t;
        const float k = t > thr ? 1.0f : 0.0f;
        const float g = 1.0f / (1.0f + expf(mean_e - fabsf(d)));
        w2[0] += (double)(d * d * (1.0f + am1 * k * (1.0f + g)));
        w2[1] += (double)(k * d * d * g * (1.0f - g));
    }
    block_sum4(w2);
    if (threadIdx.x == 0) loss[m] = (float)(w2[0] / N);
    const float g_over_n = (float)(w2[1] / N);
    const float scale = (float)(1.0 / (N * (double)S));  // d pred_mean / d pred_sample = 1/S
    for (int64_t r = threadIdx.x; r < rows; r += NT) {
        const float t = yb[(int64_t)m * rows + r], d = pred(r) - t;
        const float k = t > thr ? 1.0f : 0.0f;
        const float g = 1.0f / (1.0f + expf(mean_e - fabsf(d)));
        const float w = 1.0f + am1 * k * (1.0f + g);
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        delta[(int64_t)m * rows + r] = scale * (2.0f * d * w + am1 * sgn * (k * d * d * g * (1.0f - g) - g_over_n));
    }
}

// T4: backward of one pair: recompute the forward tile, then back-propagate -----------------------
__global__ __launch_bounds__(NT, 1) void train_backward_kernel(const float* __restrict__ xb, int64_t rows, int F, int S,
                                                               const float* __restrict__ W, const float* __restrict__ WT, MaskSrc ms,
                                                               const float* __restrict__ delta, float* __restrict__ dW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S, t = threadIdx.x;
    const int64_t P = flat_param_count(F);
    const float* w = W + (int64_t)pair * P;
    float* g = dW + (int64_t)pair * P;
    const float sc = ms.mode != 0 ? ms.scale : 1.0f;
    float* d2 = l.gc;            // dz2 [TR][64]
    float* d2t = l.gc + TR * H3; // dz2^T [64][TR]
    float* d1 = l.gb;            // dz1 [TR][128]
    float* d1t = l.gb + TR * H2; // dz1^T [128][TR]
    float* d0 = l.ga;            // dz0 [TR][256]
    for (int64_t row_base = 0; row_base < rows; row_base += TR) {
        const bool first = row_base == 0;
        forward_tile(l, xb + (int64_t)m * rows * F, rows, F, row_base, WT + (int64_t)pair * P, ms, pair);
        // ---- layer 3: dw3[k] = sum_r dy[r] h3[r][k], db3 = sum_r dy[r]; dz2 = dy * w3 * d(h3)
        if (t < TR) l.y[t] = row_base + t < rows ? delta[(int64_t)m * rows + row_base + t] : 0.0f;
        __syncthreads();
        if (t < H3) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < TR; ++r) a = fmaf(l.y[r], l.h3[r * H3 + t], a);
            float* p = g + flat_w_off(F, 3) + t;
            *p = first ? a : *p + a;
        } else if (t == H3) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < TR; ++r) a += l.y[r];
            float* p = g + flat_b_off(F, 3);
            *p = first ? a : *p + a;
        }
        for (int i = t; i < TR * H3; i += NT) {
            const int r = i / H3, k = i % H3;
            const float v = l.y[r] * w[flat_w_off(F, 3) + k] * dfactor(l.h3[i], sc);
            d2[i] = v;
            d2t[k * TR + r] = v;
        }
        __syncthreads();
        // ---- layer 2 weights: dW2[o][k] = sum_r dz2[r][o] h2[r][k]  (64 x 128), db2
        {
            const int k = t % H2, og = t / H2;  // 2 groups of 32 outputs
            float hk[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) hk[r] = l.h2[r * H2 + k];
            for (int o = og * 32; o < og * 32 + 32; ++o) {
                float a = 0.0f;
#pragma unroll
                for (int r4 = 0; r4 < TR; r4 += 4) {
                    const float4 z = *reinterpret_cast<const float4*>(d2t + o * TR + r4);
                    a = fmaf(z.x, hk[r4], a);
                    a = fmaf(z.y, hk[r4 + 1], a);
                    a = fmaf(z.z, hk[r4 + 2], a);
                    a = fmaf(z.w, hk[r4 + 3], a);
                }
                float* p = g + flat_w_off(F, 2) + (int64_t)o * H2 + k;
                *p = first ? a : *p + a;
            }
            if (t < H3) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < TR; ++r) a += d2t[t * TR + r];
                float* p = g + flat_b_off(F, 2) + t;
                *p = first ? a : *p + a;
            }
        }
        // ---- dz1[r][k] = (sum_o dz2[r][o] W2[o][k]) * d(h2[r][k])   (k < 128)
        {
            const int k = t % H2, rg = t / H2;  // 2 row groups of 16
            float a[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) a[rr] = 0.0f;
            const float* w2 = w + flat_w_off(F, 2);
            for (int o = 0; o < H3; o += 4) {
                const float w0 = w2[(o + 0) * H2 + k], w1 = w2[(o + 1) * H2 + k], w2v = w2[(o + 2) * H2 + k], w3 = w2[(o + 3) * H2 + k];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const float4 z = *reinterpret_cast<const float4*>(d2 + (rg * 16 + rr) * H3 + o);
                    a[rr] = fmaf(z.x, w0, a[rr]);
                    a[rr] = fmaf(z.y, w1, a[rr]);
                    a[rr] = fmaf(z.z, w2v, a[rr]);
                    a[rr] = fmaf(z.w, w3, a[rr]);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int r = rg * 16 + rr;
                const float v = a[rr] * dfactor(l.h2[r * H2 + k], sc);
                d1[r * H2 + k] = v;
                d1t[k * TR + r] = v;
            }
        }
        __syncthreads();
        // ---- layer 1 weights: dW1[o][k] = sum_r dz1[r][o] h1[r][k]  (128 x 256), db1
        {
            const int k = t;  // 256 threads = 256 inputs
            float hk[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) hk[r] = l.h1[r * H1 + k];
            for (int o = 0; o < H2; ++o) {
                float a = 0.0f;
#pragma unroll
                for (int r4 = 0; r4 < TR; r4 += 4) {
                    const float4 z = *reinterpret_cast<const float4*>(d1t + o * TR + r4);
                    a = fmaf(z.x, hk[r4], a);
                    a = fmaf(z.y, hk[r4 + 1], a);
                    a = fmaf(z.z, hk[r4 + 2], a);
                    a = fmaf(z.w, hk[r4 + 3], a);
                }
                float* p = g + flat_w_off(F, 1) + (int64_t)o * H1 + k;
                *p = first ? a : *p + a;
            }
            if (t < H2) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < TR; ++r) a += d1t[t * TR + r];
                float* p = g + flat_b_off(F, 1) + t;
                *p = first ? a : *p + a;
            }
        }
        // ---- dz0[r][k] = (sum_o dz1[r][o] W1[o][k]) * d(h1[r][k])   (k < 256)
        {
            const int k = t;
            float a[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) a[r] = 0.0f;
            const float* w1 = w + flat_w_off(F, 1);
            for (int o = 0; o < H2; o += 4) {
                const float w0 = w1[(o + 0) * H1 + k], w1v = w1[(o + 1) * H1 + k], w2v = w1[(o + 2) * H1 + k], w3 = w1[(o + 3) * H1 + k];
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    const float4 z = *reinterpret_cast<const float4*>(d1 + r * H2 + o);
                    a[r] = fmaf(z.x, w0, a[r]);
                    a[r] = fmaf(z.y, w1v, a[r]);
                    a[r] = fmaf(z.z, w2v, a[r]);
                    a[r] = fmaf(z.w, w3, a[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < TR; ++r) d0[r * H1 + k] = a[r] * dfactor(l.h1[r * H1 + k], sc);
        }
        __syncthreads();
        // ---- layer 0 weights: dW0[o][k] = sum_r dz0[r][o] x[r][k]  (256 x F), db0
        {
            const int o = t;
            float z[TR];
            float bsum = 0.0f;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                z[r] = d0[r * H1 + o];
                bsum += z[r];
            }
            for (int k = 0; k < F; ++k) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < TR; ++r) a = fmaf(z[r], l.xs[r * XS + k], a);
                float* p = g + (int64_t)o * F + k;
                *p = first ? a : *p + a;
            }
            float* p = g + flat_b_off(F, 0) + o;
            *p = first ? bsum : *p + bsum;
        }
        __syncthreads();
    }
}

// T5: gradients of (mu, rho) from the per-pair weight gradients + per-block squared norms ----------
__global__ __launch_bounds__(256) void train_grad_kernel(const float* __restrict__ dW, const float* __restrict__ eps,
                                                        const float* __restrict__ sigma, int S, int64_t P, float* __restrict__ gmu,
                                                        float* __restrict__ grho, double* __restrict__ normsq_part) {
    __shared__ double red[4];
    const int m = blockIdx.y;
    double nsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.0f, b = 0.0f;
        for (int s = 0; s < S; ++s) {  // fixed order: deterministic
            const float d = dW[((int64_t)m * S + s) * P + i];
            a += d;
            b = fmaf(d, eps[((int64_t)m * S + s) * P + i], b);
        }
        b *= sigma[m * P + i];
        gmu[m * P + i] = a;
        grho[m * P + i] = b;
        nsq += (double)a * a + (double)b * b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nsq += __shfl_down(nsq, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nsq;
    __syncthreads();
    if (threadIdx.x == 0) normsq_part[(int64_t)m * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// T5b: per-block squared norms of finished (mu, rho) gradients - the data-parallel step recomputes them
// after the gradient all-reduce (same grid and summation order as train_grad_kernel)
__global__ __launch_bounds__(256) void train_gradnorm_kernel(const float* __restrict__ gmu, const float* __restrict__ grho, int64_t P,
                                                            double* __restrict__ normsq_part) {
    __shared__ double red[4];
    const int m = blockIdx.y;
    double nsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = gmu[m * P + i], b = grho[m * P + i];
        nsq += (double)a * a + (double)b * b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nsq += __shfl_down(nsq, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nsq;
    __syncthreads();
    if (threadIdx.x == 0) normsq_part[(int64_t)m * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct AdamHyper {
    float lr, beta1, beta2, eps, weight_decay, max_norm;
    float bias1, bias2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
};

// T6: clip_grad_norm_ + AdamW (torch semantics), refresh sigma = exp(rho) ------------------------------
__global__ __launch_bounds__(256) void train_adamw_kernel(float* __restrict__ mu, float* __restrict__ rho, float* __restrict__ sigma,
                                                         const float* __restrict__ gmu, const float* __restrict__ grho,
                                                         float* __restrict__ am, float* __restrict__ av, int64_t P,
                                                         const double* __restrict__ normsq_part, int nparts,
                                                         const uint8_t* __restrict__ active, AdamHyper hp, float* __restrict__ gnorm) {
    const int m = blockIdx.y;
    if (active && !active[m]) return;
    double nsq = 0.0;
    for (int i = 0; i < nparts; ++i) nsq += normsq_part[(int64_t)m * nparts + i];  // same order in every thread
    const float total = (float)sqrt(nsq);
    if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm) gnorm[m] = total;
    float coef = hp.max_norm / (total + 1e-6f);  // torch.nn.utils.clip_grad_norm_
    coef = coef > 1.0f ? 1.0f : coef;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * P; i += (int64_t)gridDim.x * blockDim.x) {
        const bool is_rho = i >= P;
        const int64_t j = is_rho ? i - P : i;
        float* param = (is_rho ? rho : mu) + m * P + j;
        const float gr = (is_rho ? grho : gmu)[m * P + j] * coef;
        float p = *param;
        p *= 1.0f - hp.lr * hp.weight_decay;
        const int64_t si = (int64_t)m * 2 * P + i;
        const float mm = am[si] + (gr - am[si]) * (1.0f - hp.beta1);  // lerp_
        const float vv = av[si] * hp.beta2 + (1.0f - hp.beta2) * gr * gr;
        am[si] = mm;
        av[si] = vv;
        const float denom = sqrtf(vv) / hp.bias2_sqrt + hp.eps;
        p -= (hp.lr / hp.bias1) * (mm / denom);
        *param = p;
        if (is_rho) sigma[m * P + j] = expf(p);
    }
}

__global__ __launch_bounds__(256) void exp_kernel(const float* __restrict__ rho, float* __restrict__ sigma, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) sigma[i] = expf(rho[i]);
}

constexpr int kGradBlocks = 32;

}  // namespace

int launch_train_sigma(const float* rho, float* sigma, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, rho, sigma, n);
    return check_launch("train_sigma");
}

int launch_train_sample(const float* mu, const float* sigma, int M, int S, int F, int noise_mode, const float* eps_in, uint64_t seed,
                        float* eps, float* W, float* WT, hipStream_t s) {
    hipLaunchKernelGGL(train_sample_kernel, dim3(44, M * S), dim3(256), 0, s, mu, sigma, S, F, noise_mode, eps_in, seed, eps, W, WT);
    return check_launch("train_sample");
}

static MaskSrc make_mask(int mode, const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t seed, float p_drop) {
    MaskSrc ms;
    ms.m0 = m0;
    ms.m1 = m1;
    ms.m2 = m2;
    ms.seed = seed;
    ms.mode = mode;
    ms.keep_prob = 1.0f - p_drop;
    ms.scale = (float)(1.0 / (1.0 - (double)p_drop));  // torch: scale = 1/(1-p) in double, applied as fp32
    return ms;
}

int launch_train_forward(const float* xb, int64_t rows, int F, int M, int S, const float* WT, int mask_mode, const uint8_t* m0,
                         const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, float* Y, hipStream_t s) {
    const size_t lds = kLdsFloats * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipFuncSetAttribute(train_forward): %s", hipGetErrorString(e));
    hipLaunchKernelGGL(train_forward_kernel, dim3(M * S), dim3(NT), lds, s, xb, rows, F, S, WT,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), Y);
    return check_launch("train_forward");
}

int launch_train_loss(const float* Y, const float* yb, int M, int S, int64_t rows, float alpha, float beta, float* delta, float* loss,
                      hipStream_t s) {
    hipLaunchKernelGGL(train_loss_kernel, dim3(M), dim3(NT), 0, s, Y, yb, S, rows, alpha, beta, delta, loss);
    return check_launch("train_loss");
}

int launch_train_backward(const float* xb, int64_t rows, int F, int M, int S, const float* W, const float* WT, int mask_mode,
                          const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, const float* delta,
                          float* dW, hipStream_t s) {
    const size_t lds = kLdsFloats * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipFuncSetAttribute(train_backward): %s", hipGetErrorString(e));
    hipLaunchKernelGGL(train_backward_kernel, dim3(M * S), dim3(NT), lds, s, xb, rows, F, S, W, WT,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), delta, dW);
    return check_launch("train_backward");
}

int launch_train_grad(int M, int S, int F, const float* eps, const float* sigma, const float* dW, float* gmu, float* grho,
                      double* normsq_part, hipStream_t s) {
    const int64_t P = flat_param_count(F);
    hipLaunchKernelGGL(train_grad_kernel, dim3(kGradBlocks, M), dim3(256), 0, s, dW, eps, sigma, S, P, gmu, grho, normsq_part);
    return check_launch("train_grad");
}

int launch_train_apply(float* mu, float* rho, float* sigma, int M, int F, const float* gmu, const float* grho, float* am, float* av,
                       double* normsq_part, bool recompute_norm, const uint8_t* active, float lr, float beta1, float beta2,
                       float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, hipStream_t s) {
    const int64_t P = flat_param_count(F);
    if (recompute_norm) hipLaunchKernelGGL(train_gradnorm_kernel, dim3(kGradBlocks, M), dim3(256), 0, s, gmu, grho, P, normsq_part);
    AdamHyper hp;
    hp.lr = lr;
    hp.beta1 = beta1;
    hp.beta2 = beta2;
    hp.eps = adam_eps;
    hp.weight_decay = weight_decay;
    hp.max_norm = max_norm;
    hp.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
    hp.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(train_adamw_kernel, dim3(64, M), dim3(256), 0, s, mu, rho, sigma, gmu, grho, am, av, P, normsq_part, kGradBlocks,
                       active, hp, gnorm);
    return check_launch("train_apply");
}

int launch_train_update(float* mu, float* rho, float* sigma, int M, int S, int F, const float* eps, const float* dW, float* gmu,
                        float* grho, float* am, float* av, double* normsq_part, const uint8_t* active, float lr, float beta1,
                        float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, hipStream_t s) {
    if (int rc = launch_train_grad(M, S, F, eps, sigma, dW, gmu, grho, normsq_part, s)) return rc;
    return launch_train_apply(mu, rho, sigma, M, F, gmu, grho, am, av, normsq_part, false, active, lr, beta1, beta2, adam_eps,
                              weight_decay, max_norm, step, gnorm, s);
}

int train_grad_blocks() { return kGradBlocks; }

}  // namespace mcmc
