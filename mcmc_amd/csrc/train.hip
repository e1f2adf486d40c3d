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
// The batch is 32 rows (:124), so every GEMM here is one or a few 32x32 MFMA tiles.  One workgroup (4 or 8 waves,
// train_tiles.hpp) owns one (member, sample) pair and keeps the activations of a 32-row tile in LDS ([row][feature] tiles with
// odd strides, 62 KB: two workgroups per CU; the backward pass turns them into gradients in place); the weights stream from L2 in two pre-sampled layouts
// (WT [in][out] for the forward, W [out][in] for the input gradients) as MFMA A operands, 32 consecutive
// floats per lane half.  All GEMMs run on v_mfma_f32_32x32x2_f32: exact fp32 FMAs, so the step stays
// within 2e-5 of the reference's fp32 autograd.
#include "common.hpp"
#include "dropmask.hpp"
#include "elu.hpp"
#include "member_loss.hpp"

namespace mcmc {

namespace {

// LDS row strides of the [row][feature] tiles: odd, so a column read (32 rows of one feature: the MFMA's B
// operand) and a row read (32 consecutive features: its A operand) are both bank-conflict-free
constexpr int SX = 33, S1 = H1 + 1, S2 = H2 + 1, S3 = H3 + 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One k-step of v_mfma_f32_32x32x2_f32 (exact fp32 FMAs): D[32 x 32] += A[32 x 2] * B[2 x 32]; lane (r, h) feeds
// A[r][h] and B[h][r]; afterwards register i of lane (r, h) holds D[acc_feature(i, h)][r] (layout.hpp).
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); }

// derivative factor d = elu'(z) * mask * scale recovered from the stored activation
__device__ __forceinline__ float dfactor(float h, float scale) {
    if (__float_as_uint(h) == 0x80000000u) return 0.0f;  // masked (or padding row)
    return h > 0.0f ? scale : h + scale;                 // z > 0: scale;  z <= 0: exp(z)*scale = h + scale
}

// T3: pred_mean, peak-weighted loss and d loss / d pred (per sample) of one member: member_loss.hpp

// T4: backward of one pair: recompute (or reload) the forward tile, then back-propagate ----------------
// With LossFuse::Y set, the kernel derives dLoss/dy itself: every workgroup of a member repeats the member's two
// loss reductions (a few microseconds at the reference's batch size) instead of waiting for a separate loss launch.
struct LossFuse {
    const float* Y;   // [M*S, rows] predictions of the forward kernel, or nullptr: read `delta`
    const float* yb;  // targets, member stride y_stride
    int64_t y_stride;
    float alpha, beta;
    float* loss;       // [M]
    double* loss_acc;  // optional [M]
};

// The tile code is compiled for two workgroup sizes (train_tiles.hpp): four waves, two workgroups per CU, when a
// launch has more workgroups than the chip has CUs (large batches); eight waves when every workgroup has a CU to
// itself (the reference's batch sizes: 50 pairs), where the second wave per SIMD hides the first one's L2 latency.
#define MCMC_TRAIN_NT 256
#define MCMC_TRAIN_TR 32
namespace nt256 {
#include "train_tiles.hpp"
}
#undef MCMC_TRAIN_NT
#define MCMC_TRAIN_NT 512
namespace nt512 {
#include "train_tiles.hpp"
}
#undef MCMC_TRAIN_TR
#define MCMC_TRAIN_TR 64
namespace nt512r64 {
#include "train_tiles.hpp"
}
#undef MCMC_TRAIN_TR
#undef MCMC_TRAIN_NT
constexpr int kWideMaxWorkgroups = 256;  // CUs of an MI355X

// T1: eps, effective weights in both layouts ------------------------------------------------------
// One workgroup = one 32 x 32 tile of one layer's weight matrix of one pair (or the biases + output layer):
// noise, W = mu + eps * sigma and eps are written along the input index (the flat layout), the tile goes
// through LDS, and WT is written along the output index - both global writes are lane-contiguous.
// Layers 1-2: a thread owns four consecutive inputs of one output row = one Philox block (philox_normal4).
__device__ __forceinline__ float sample_one(const float* __restrict__ mu, const float* __restrict__ sigma, int noise_mode,
                                            const float* __restrict__ eps_in, uint64_t seed, int pair, int64_t P, int m, int64_t i,
                                            float& e) {
    e = 0.0f;
    if (noise_mode == MCMC_NOISE_INJECTED) e = eps_in[(int64_t)pair * P + i];
    else if (noise_mode == MCMC_NOISE_PHILOX) e = philox_normal(seed, (uint32_t)pair, (uint32_t)i);
    return noise_mode == MCMC_NOISE_NONE ? mu[m * P + i] : __fadd_rn(mu[m * P + i], __fmul_rn(e, sigma[m * P + i]));
}

constexpr int kSampleTiles = 8 + 32 + 8 + 1;  // layer 0 (8 x 1), layer 1 (4 x 8), layer 2 (2 x 4), the rest

__global__ __launch_bounds__(256) void train_sample_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, int S, int F,
                                                          int noise_mode, const float* __restrict__ eps_in, uint64_t seed,
                                                          float* __restrict__ eps, float* __restrict__ W, float* __restrict__ WT) {
    __shared__ float tl[32][33];
    const int64_t P = flat_param_count(F);
    const int pair = blockIdx.y, m = pair / S, t = threadIdx.x;
    const int64_t base = (int64_t)pair * P;
    int tile = blockIdx.x;
    if (tile >= 48) {  // biases of the hidden layers, then the output layer (w3 [1 x 64] is its own transpose)
        const int64_t lo[4] = {flat_b_off(F, 0), flat_b_off(F, 1), flat_b_off(F, 2), flat_w_off(F, 3)};
        const int n[4] = {H1, H2, H3, H3 + 1};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            for (int k = t; k < n[q]; k += 256) {
                float e;
                const float w = sample_one(mu, sigma, noise_mode, eps_in, seed, pair, P, m, lo[q] + k, e);
                eps[base + lo[q] + k] = e;
                W[base + lo[q] + k] = w;
                WT[base + lo[q] + k] = w;
            }
        return;
    }
    int layer = 0, n_in_tiles = 1;
    if (tile >= 40) {
        layer = 2;
        tile -= 40;
        n_in_tiles = H2 / 32;
    } else if (tile >= 8) {
        layer = 1;
        tile -= 8;
        n_in_tiles = H1 / 32;
    }
    const int to = tile / n_in_tiles, ti = tile % n_in_tiles;
    const int in = layer_in(F, layer), out = layer_out(layer);
    const int64_t wo = flat_w_off(F, layer);
    const int ol = t >> 3, kl0 = (t & 7) * 4;  // output row and first of four inputs inside the tile
    const int o = 32 * to + ol, k0 = 32 * ti + kl0;
    const int64_t i0 = wo + (int64_t)o * in + k0;
    if (layer > 0 && noise_mode == MCMC_NOISE_PHILOX) {
        float e[4];
        philox_normal4(seed, (uint32_t)pair, (uint32_t)(i0 >> 2), e);  // i0 % 4 == 0 for layers 1-2
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w = __fadd_rn(mu[m * P + i0 + j], __fmul_rn(e[j], sigma[m * P + i0 + j]));
            eps[base + i0 + j] = e[j];
            W[base + i0 + j] = w;
            tl[ol][kl0 + j] = w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < in) {
                float e;
                const float w = sample_one(mu, sigma, noise_mode, eps_in, seed, pair, P, m, i0 + j, e);
                eps[base + i0 + j] = e;
                W[base + i0 + j] = w;
                tl[ol][kl0 + j] = w;
            }
    }
    __syncthreads();
    // WT[k][o]: 32 consecutive outputs per input row
    const int ol2 = t & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kl = (t >> 5) + 8 * q, k = 32 * ti + kl;
        if (k < in) WT[base + wo + (int64_t)k * out + 32 * to + ol2] = tl[ol2][kl];
    }
}

// T5: gradients of (mu, rho) from the per-pair weight gradients + per-block squared norms ----------
__global__ __launch_bounds__(256) void train_grad_kernel(const float* __restrict__ dW, const float* __restrict__ eps,
                                                        const float* __restrict__ sigma, int S, int NB, int64_t P,
                                                        float* __restrict__ gmu, float* __restrict__ grho,
                                                        double* __restrict__ normsq_part) {
    __shared__ double red[4];
    const int m = blockIdx.y;
    double nsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.0f, b = 0.0f;
        if (NB == 1) {  // the reference's batch size: the loads of eight samples are issued together, then summed in sample order
            for (int s0 = 0; s0 < S; s0 += 8) {
                float d[8], e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = s0 + u < S;
                    d[u] = in ? dW[((int64_t)m * S + s0 + u) * P + i] : 0.0f;
                    e[u] = in ? eps[((int64_t)m * S + s0 + u) * P + i] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < S) {
                        a += d[u];
                        b = fmaf(d[u], e[u], b);
                    }
            }
        } else {
            for (int s = 0; s < S; ++s) {  // fixed order: deterministic
                float d = dW[((int64_t)m * S + s) * NB * P + i];
                for (int nb = 1; nb < NB; ++nb) d += dW[(((int64_t)m * S + s) * NB + nb) * P + i];  // row-block partials
                a += d;
                b = fmaf(d, eps[((int64_t)m * S + s) * P + i], b);
            }
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
    // squared norm from the per-block partials (nparts <= 256): one partial per thread, a fixed shuffle tree per
    // wave, then the four wave sums in order - the same value in every workgroup and every run
    __shared__ double wsum[4];
    double v = (int)threadIdx.x < nparts ? normsq_part[(int64_t)m * nparts + threadIdx.x] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    const double nsq = wsum[0] + wsum[1] + wsum[2] + wsum[3];
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

// Early stopping of bayesian_ensemble.py:183-193 for all members after an epoch (one thread per member): a member
// still active improves its best epoch loss or counts a bad epoch, and is frozen once it has `patience` of them.
__global__ void early_stop_kernel(const double* __restrict__ epoch_loss, double* __restrict__ best, int64_t* __restrict__ bad,
                                  uint8_t* __restrict__ active, int M, int64_t patience) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || !active[m]) return;
    const double el = epoch_loss[m];
    if (el < best[m]) {
        best[m] = el;
        bad[m] = 0;
    } else {
        bad[m] += 1;
    }
    if (bad[m] >= patience) active[m] = 0;
}

constexpr int kGradBlocks = 128;

}  // namespace

int launch_train_sigma(const float* rho, float* sigma, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, rho, sigma, n);
    return check_launch("train_sigma");
}

int launch_train_early_stop(const double* epoch_loss, double* best, int64_t* bad, uint8_t* active, int M, int64_t patience, hipStream_t s) {
    hipLaunchKernelGGL(early_stop_kernel, dim3((M + 63) / 64), dim3(64), 0, s, epoch_loss, best, bad, active, M, patience);
    return check_launch("train_early_stop");
}

int launch_train_sample(const float* mu, const float* sigma, int M, int S, int F, int noise_mode, const float* eps_in, uint64_t seed,
                        float* eps, float* W, float* WT, hipStream_t s) {
    hipLaunchKernelGGL(train_sample_kernel, dim3(kSampleTiles, M * S), dim3(256), 0, s, mu, sigma, S, F, noise_mode, eps_in, seed, eps, W,
                       WT);
    return check_launch("train_sample");
}

static DropMask make_mask(int mode, const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t seed, float p_drop) {
    DropMask dm{};
    dm.m[0] = m0;
    dm.m[1] = m1;
    dm.m[2] = m2;
    dm.seed = seed;
    dm.mode = mode;
    dm.keep_prob = 1.0f - p_drop;
    dm.scale = (float)(1.0 / (1.0 - (double)p_drop));  // torch: scale = 1/(1-p) in double, applied as fp32
    dm.M = 1;  // the pair index IS the global pair id (member * S + sample)
    dm.S = 1;
    dm.pair_begin = 0;
    return dm;
}

// Which build of the tile code runs a launch.  Forward and backward of one step must agree (the saved activations are
// laid out per tile), so both derive it from the same two numbers.
enum class Tiles { k256, k512, k512r64 };
static Tiles pick_tiles(int64_t pairs, int row_blocks) {
    if (row_blocks > 1) return Tiles::k512r64;  // large batches: 64-row tiles, eight waves, one workgroup per CU
    return pairs <= kWideMaxWorkgroups ? Tiles::k512 : Tiles::k256;
}

template <typename K>
static int set_dyn_lds(K kern, size_t lds, const char* what) {
    return ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, what);
}

#define MCMC_TRAIN_PICK(NS, KERNEL, mode) ((mode) == 1 ? NS::KERNEL<1> : (mode) == 2 ? NS::KERNEL<2> : NS::KERNEL<0>)

int launch_train_forward(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* WT, int mask_mode, const uint8_t* m0,
                         const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, float* Y, float* act, hipStream_t s) {
    const int row_blocks = train_row_blocks(rows);
    const Tiles tv = pick_tiles((int64_t)M * S, row_blocks);
    auto kern = tv == Tiles::k512r64 ? MCMC_TRAIN_PICK(nt512r64, train_forward_kernel, mask_mode)
                : tv == Tiles::k512  ? MCMC_TRAIN_PICK(nt512, train_forward_kernel, mask_mode)
                                     : MCMC_TRAIN_PICK(nt256, train_forward_kernel, mask_mode);
    const size_t lds = (tv == Tiles::k512r64 ? nt512r64::kLdsFloats : nt256::kLdsFloats) * sizeof(float);
    if (int rc = set_dyn_lds(kern, lds, "train_forward")) return rc;
    hipLaunchKernelGGL(kern, dim3(M * S, row_blocks), dim3(tv == Tiles::k256 ? 256 : 512), lds, s, xb, x_stride, rows, F, S, WT,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), Y, act);
    return check_launch("train_forward");
}

int launch_train_loss(const float* Y, const float* yb, int64_t y_stride, int M, int S, int64_t rows, float alpha, float beta, float* delta,
                      float* loss, double* loss_acc, hipStream_t s) {
    hipLaunchKernelGGL(nt256::train_loss_kernel, dim3(M), dim3(nt256::NT), 0, s, Y, yb, y_stride, S, rows, alpha, beta, delta, loss, loss_acc);
    return check_launch("train_loss");
}

int launch_train_backward(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, const float* WT, int mask_mode,
                          const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, const float* delta,
                          int row_blocks, float* dW, const float* act, const TrainLossFuse* fuse, hipStream_t s) {
    const Tiles tv = pick_tiles((int64_t)M * S, row_blocks);
    auto kern = tv == Tiles::k512r64 ? MCMC_TRAIN_PICK(nt512r64, train_backward_kernel, mask_mode)
                : tv == Tiles::k512  ? MCMC_TRAIN_PICK(nt512, train_backward_kernel, mask_mode)
                                     : MCMC_TRAIN_PICK(nt256, train_backward_kernel, mask_mode);
    const size_t lds = (tv == Tiles::k512r64 ? nt512r64::kLdsFloats : nt256::kLdsFloats) * sizeof(float);
    if (int rc = set_dyn_lds(kern, lds, "train_backward")) return rc;
    LossFuse lf{};
    if (fuse) lf = LossFuse{fuse->Y, fuse->yb, fuse->y_stride, fuse->alpha, fuse->beta, fuse->loss, fuse->loss_acc};
    hipLaunchKernelGGL(kern, dim3(M * S, row_blocks), dim3(tv == Tiles::k256 ? 256 : 512), lds, s, xb, x_stride, rows, F, S, W, WT,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), delta, dW, act, lf);
    return check_launch("train_backward");
}

int launch_train_grad(int M, int S, int row_blocks, int F, const float* eps, const float* sigma, const float* dW, float* gmu,
                      float* grho, double* normsq_part, hipStream_t s) {
    const int64_t P = flat_param_count(F);
    hipLaunchKernelGGL(train_grad_kernel, dim3(kGradBlocks, M), dim3(256), 0, s, dW, eps, sigma, S, row_blocks, P, gmu, grho,
                       normsq_part);
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
    const unsigned adam_blocks = (unsigned)((2 * P + 255) / 256);  // one element per thread: the loop's loads do not serialise
    hipLaunchKernelGGL(train_adamw_kernel, dim3(adam_blocks, M), dim3(256), 0, s, mu, rho, sigma, gmu, grho, am, av, P, normsq_part, kGradBlocks,
                       active, hp, gnorm);
    return check_launch("train_apply");
}

int launch_train_update(float* mu, float* rho, float* sigma, int M, int S, int F, const float* eps, const float* dW, float* gmu,
                        float* grho, float* am, float* av, double* normsq_part, const uint8_t* active, float lr, float beta1,
                        float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm, hipStream_t s) {
    if (int rc = launch_train_grad(M, S, 1, F, eps, sigma, dW, gmu, grho, normsq_part, s)) return rc;
    return launch_train_apply(mu, rho, sigma, M, F, gmu, grho, am, av, normsq_part, false, active, lr, beta1, beta2, adam_eps,
                              weight_decay, max_norm, step, gnorm, s);
}

int train_grad_blocks() { return kGradBlocks; }

// floats per pair of the saved-activation buffer for a batch of `rows` rows
// (sized for 64-row tiles, which also covers the 32-row builds' [tiles of 32 rows] layout of the same batch)
int64_t train_act_floats(int64_t rows) { return (rows + 63) / 64 * (int64_t)nt512r64::kActFloats; }

// Row blocks a batch of `rows` rows per member is split into (each pair then runs on that many workgroups):
// one up to 2,047 rows - the reference's 32-row batches - then one per 1,024 rows, at most 64.
int train_row_blocks(int64_t rows) {
    const int64_t nb = rows / 1024;
    return (int)(nb < 1 ? 1 : nb > 64 ? 64 : nb);
}

}  // namespace mcmc
