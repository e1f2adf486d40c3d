// Piece 3: fused peak-weighted loss, forward + closed-form backward (HBM-bound).
//
// Restates peak_weighted_loss (/root/reference/utils.py:7-28):
//   d = p - t;  m = mean(t);  s = std(t) (unbiased);  k = [t > m + beta*s]   (no grad)
//   e = |d|;  g = sigmoid(e - mean(e));  w = 1 + (alpha-1) k (1+g);  L = mean(d^2 w)
// and its autograd gradient in closed form:
//   dL/dp_j = (1/N) [ 2 d_j w_j + (alpha-1) sign(d_j) (k_j d_j^2 g'_j - G/N) ],  g' = g(1-g),
//   G = sum_i k_i d_i^2 g'_i      (the mean(e) inside the sigmoid carries gradient, utils.py:23)
// The reference runs ~12 elementwise/reduce kernels forward plus autograd backward; here three
// streaming passes, each reading p,t once with 16-B loads.  The global sums sit in device
// memory (fp64) between passes so that data-parallel ranks can all-reduce them.
#include <hip/hip_cooperative_groups.h>

#include "common.hpp"

namespace cg = cooperative_groups;

namespace mcmc {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

template <int NV>
__device__ __forceinline__ void block_accumulate(double (&v)[NV], double* __restrict__ out) {
    __shared__ double sm[NV][kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double w = wave_sum(v[i]);
        if (lane == 0) sm[i][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) a += sm[threadIdx.x][w];
        atomicAdd(out + threadIdx.x, a);  // global_atomic_add_f64
    }
}

struct LossConsts {
    float thr;     // m + beta*s   (NaN when N == 1: k == 0 everywhere, plain MSE)
    float mean_e;  // mean |p-t|
    float am1;     // alpha - 1
};

__device__ __forceinline__ LossConsts loss_consts(const double* __restrict__ stats, double N, float alpha, float beta) {
    const double st = stats[0], stt = stats[1], se = stats[2];
    const double m = st / N;
    const double var = (stt - st * st / N) / (N - 1.0);  // N == 1 -> 0/0 = NaN like torch.std
    const double s = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
    LossConsts c;
    c.thr = (float)(m + (double)beta * s);
    c.mean_e = (float)(se / N);
    c.am1 = alpha - 1.0f;
    return c;
}

// per-element terms shared by forward and backward
__device__ __forceinline__ void elem_terms(float p, float t, const LossConsts& c, float& d, float& w, float& kd2gp) {
    d = p - t;
    const float e = fabsf(d);
    const float k = t > c.thr ? 1.0f : 0.0f;
    const float g = 1.0f / (1.0f + expf(c.mean_e - e));
    w = 1.0f + c.am1 * k * (1.0f + g);
    kd2gp = k * d * d * g * (1.0f - g);
}

__global__ __launch_bounds__(kBlock) void loss_stats_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                            double* __restrict__ stats) {
    double acc[3] = {0.0, 0.0, 0.0};
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte load pairs in flight per thread
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = p4[i + v * stride];
            b[v] = t4[i + v * stride];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float pv[4] = {a[v].x, a[v].y, a[v].z, a[v].w}, tv[4] = {b[v].x, b[v].y, b[v].z, b[v].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double td = (double)tv[u];
                acc[0] += td;
                acc[1] = fma(td, td, acc[1]);
                acc[2] += (double)fabsf(pv[u] - tv[u]);
            }
        }
    }
    for (; i < n4; i += stride) {
        const float4 a = p4[i], b = t4[i];
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double td = (double)tv[u];
            acc[0] += td;
            acc[1] = fma(td, td, acc[1]);
            acc[2] += (double)fabsf(pv[u] - tv[u]);
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double td = (double)t[i];
        acc[0] += td;
        acc[1] = fma(td, td, acc[1]);
        acc[2] += (double)fabsf(p[i] - t[i]);
    }
    block_accumulate<3>(acc, stats);
}

__global__ __launch_bounds__(kBlock) void loss_forward_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                              double N, const double* __restrict__ stats, float alpha, float beta,
                                                              double* __restrict__ part) {
    const LossConsts c = loss_consts(stats, N, alpha, beta);
    double acc[2] = {0.0, 0.0};
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const float4 a = p4[i], b = t4[i];
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d, w, q;
            elem_terms(pv[u], tv[u], c, d, w, q);
            acc[0] += (double)(d * d * w);
            acc[1] += (double)q;
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float d, w, q;
        elem_terms(p[i], t[i], c, d, w, q);
        acc[0] += (double)(d * d * w);
        acc[1] += (double)q;
    }
    block_accumulate<2>(acc, part);
}

__device__ __forceinline__ float grad_elem(float p, float t, const LossConsts& c, float g_over_n, float scale) {
    float d, w, q;
    elem_terms(p, t, c, d, w, q);
    const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return scale * (2.0f * d * w + c.am1 * sgn * (q - g_over_n));
}

__global__ __launch_bounds__(kBlock) void loss_backward_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                               double N, const double* __restrict__ stats,
                                                               const double* __restrict__ part, float alpha, float beta,
                                                               float grad_scale, float* __restrict__ grad) {
    const LossConsts c = loss_consts(stats, N, alpha, beta);
    const float g_over_n = (float)(part[1] / N);
    const float scale = (float)((double)grad_scale / N);
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* g4 = reinterpret_cast<float4*>(grad);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const float4 a = p4[i], b = t4[i];
        float4 o;
        o.x = grad_elem(a.x, b.x, c, g_over_n, scale);
        o.y = grad_elem(a.y, b.y, c, g_over_n, scale);
        o.z = grad_elem(a.z, b.z, c, g_over_n, scale);
        o.w = grad_elem(a.w, b.w, c, g_over_n, scale);
        g4[i] = o;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        grad[i] = grad_elem(p[i], t[i], c, g_over_n, scale);
}


// ---- single-rank fused path: no atomics, no zeroing, one C call -------------------------------------------------
// Workspace (fp64): [3*G] pass-1 block partials, then [2*G] pass-2 block partials, G = grid size of all three
// launches.  Every block writes its own slots (nothing to zero) and every block of a later pass re-reduces the whole
// partial array in a fixed order (G <= 1024: a few L2-resident KB), so the sums are bitwise reproducible and no
// launch ends in serialised fp64 atomics.
template <int NV, int BLOCK = kBlock>
__device__ __forceinline__ void block_store(double (&v)[NV], double* __restrict__ slot) {
    __shared__ double sm[NV][BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double w = wave_sum(v[i]);
        if (lane == 0) sm[i][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) a += sm[threadIdx.x][w];
        slot[threadIdx.x] = a;
    }
}

// totals of NV interleaved partial columns over G blocks, the same bits in every block
// COHERENT: the partials were written by other workgroups of THIS launch (before a grid barrier): agent-scope loads
template <int NV, int BLOCK = kBlock, bool COHERENT = false>
__device__ __forceinline__ void reduce_partials(const double* part, int G, double (&tot)[NV]) {
    __shared__ double sm[NV][BLOCK / 64];
    __shared__ double res[NV];
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.0;
    for (int b = threadIdx.x; b < G; b += BLOCK)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            v[i] += COHERENT ? __hip_atomic_load(part + (int64_t)b * NV + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[(int64_t)b * NV + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double w = wave_sum(v[i]);
        if (lane == 0) sm[i][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) a += sm[threadIdx.x][w];
        res[threadIdx.x] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] = res[i];
    __syncthreads();  // sm / res are reused by the next call
}

__device__ __forceinline__ void stats_accumulate(const float* __restrict__ p, const float* __restrict__ t, int64_t n, double (&acc)[3]) {
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto take = [&](const float4& a, const float4& b) {
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double td = (double)tv[u];
            acc[0] += td;
            acc[1] = fma(td, td, acc[1]);
            acc[2] += (double)fabsf(pv[u] - tv[u]);
        }
    };
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte load pairs in flight per thread
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = p4[i + v * stride];
            b[v] = t4[i + v * stride];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) take(a[v], b[v]);
    }
    for (; i < n4; i += stride) take(p4[i], t4[i]);
    for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const double td = (double)t[j];
        acc[0] += td;
        acc[1] = fma(td, td, acc[1]);
        acc[2] += (double)fabsf(p[j] - t[j]);
    }
}

__global__ __launch_bounds__(kBlock) void fused_stats_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                             double* __restrict__ ws) {
    double acc[3] = {0.0, 0.0, 0.0};
    stats_accumulate(p, t, n, acc);
    block_store<3>(acc, ws + 3 * (int64_t)blockIdx.x);
}

__global__ __launch_bounds__(kBlock) void fused_forward_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                               float alpha, float beta, double* __restrict__ ws) {
    const int G = gridDim.x;
    double st[3];
    reduce_partials<3>(ws, G, st);
    const LossConsts c = loss_consts(st, (double)n, alpha, beta);
    double acc[2] = {0.0, 0.0};
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const int64_t stride = (int64_t)G * kBlock;
    auto take = [&](const float4& a, const float4& b) {
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d, w, q;
            elem_terms(pv[u], tv[u], c, d, w, q);
            acc[0] += (double)(d * d * w);
            acc[1] += (double)q;
        }
    };
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = p4[i + v * stride];
            b[v] = t4[i + v * stride];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) take(a[v], b[v]);
    }
    for (; i < n4; i += stride) take(p4[i], t4[i]);
    for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        float d, w, q;
        elem_terms(p[j], t[j], c, d, w, q);
        acc[0] += (double)(d * d * w);
        acc[1] += (double)q;
    }
    block_store<2>(acc, ws + 3 * (int64_t)G + 2 * (int64_t)blockIdx.x);
}

// grad == nullptr: forward only (one block suffices: it just publishes the loss)
__global__ __launch_bounds__(kBlock) void fused_backward_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                                float alpha, float beta, float grad_scale, int G,
                                                                const double* __restrict__ ws, float* __restrict__ loss,
                                                                float* __restrict__ grad) {
    double st[3], pt[2];
    reduce_partials<3>(ws, G, st);
    reduce_partials<2>(ws + 3 * (int64_t)G, G, pt);
    const double N = (double)n;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(pt[0] / N);
    if (!grad) return;
    const LossConsts c = loss_consts(st, N, alpha, beta);
    const float g_over_n = (float)(pt[1] / N);
    const float scale = (float)((double)grad_scale / N);
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* g4 = reinterpret_cast<float4*>(grad);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto one = [&](const float4& a, const float4& b) {
        float4 o;
        o.x = grad_elem(a.x, b.x, c, g_over_n, scale);
        o.y = grad_elem(a.y, b.y, c, g_over_n, scale);
        o.z = grad_elem(a.z, b.z, c, g_over_n, scale);
        o.w = grad_elem(a.w, b.w, c, g_over_n, scale);
        return o;
    };
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 a0 = p4[i], b0 = t4[i], a1 = p4[i + stride], b1 = t4[i + stride];
        g4[i] = one(a0, b0);
        g4[i + stride] = one(a1, b1);
    }
    for (; i < n4; i += stride) g4[i] = one(p4[i], t4[i]);
    for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride)
        grad[j] = grad_elem(p[j], t[j], c, g_over_n, scale);
}

// ---- one persistent launch that keeps part of the slice on chip across the three passes -------------------------
// The three passes are inherent (pass 2 needs mean|d| and the threshold, pass 3 needs G), but the second and third
// read of p,t need not come from HBM: one 512-thread workgroup per CU stays resident (cooperative launch) and keeps
// what it read in pass 1 - kKeepRegs float4 pairs per thread in registers and kKeepLds rounds of the workgroup in LDS
// (128 KB) - and only the rest of the slice is streamed again.  256 CUs hold 256 x 512 x (20 + 8) x 32 B = 117 MB:
// the whole input at the per-GPU size of config 5 (4.56e6 elements: 12 B/element of HBM traffic), 40 % of it at 3.65e7.
// Work is dealt in ROUNDS of gridDim.x * 512 float4s: round k is one coalesced sweep of the whole grid.  All accesses
// are raw buffer operations on descriptors of exactly n/4 float4s: a lane past the end loads zeros (which add exactly
// nothing to any of the five sums) and its store is dropped by the range check, so no round needs a per-lane branch
// and a group of four rounds is one straight run of independent loads.  Between the passes: per-workgroup partial
// sums in `ws` exactly as in the three-launch form, a grid barrier, and every workgroup re-reduces all partials in the
// same fixed order (deterministic, no atomics on the sums).
constexpr int kPB = 512;
constexpr int kKeepRegs = 20;  // multiple of 4
constexpr int kKeepLds = 8;    // multiple of 4
constexpr size_t kKeepLdsBytes = (size_t)kKeepLds * 2 * kPB * sizeof(float4);
constexpr int64_t kOnchipMaxN = (int64_t)1 << 29;  // byte offsets and the descriptors' record counts stay below 2^31

typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

template <int R, int L>
__global__ __launch_bounds__(kPB) void loss_onchip_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n, float alpha,
                                                          float beta, float grad_scale, double* ws, float* __restrict__ loss,
                                                          float* __restrict__ grad) {
    static_assert(R % 4 == 0 && L % 4 == 0, "rounds are handled four at a time");
    extern __shared__ float4 keep[];  // [L][2][kPB]
    cg::grid_group grid = cg::this_grid();
    const int G = gridDim.x;
    const unsigned gtid = blockIdx.x * kPB + threadIdx.x;
    const unsigned voff = gtid * 16u;                           // this lane's byte offset inside every round
    const int64_t n4 = n / 4;
    const unsigned round_bytes = (unsigned)G * kPB * 16u;        // <= 1024 x 512 x 16 = 8 MB
    const int bytes4 = (int)(n4 * 16);                           // < 2^31 (kOnchipMaxN)
    const int K = (int)(((unsigned)bytes4 + round_bytes - 1) / round_bytes);  // rounds (uniform)
    const __amdgpu_buffer_rsrc_t pbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes4, 0x00020000);
    const __amdgpu_buffer_rsrc_t tbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t), 0, bytes4, 0x00020000);
    const __amdgpu_buffer_rsrc_t gbuf = __builtin_amdgcn_make_buffer_rsrc(grad, 0, grad ? bytes4 : 0, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& buf, int k) {
        const u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(buf, (int)(voff + (unsigned)k * round_bytes), 0, 0);
        return __builtin_bit_cast(float4, v);
    };
    const int64_t j_tail = n4 * 4 + gtid;  // the n % 4 last elements (threads 0..2 of workgroup 0)

    // ---- pass 1: sums of t, t^2, |p - t|; fills the registers and the LDS rounds
    float4 rp[R], rt[R];
    double s1[3] = {0.0, 0.0, 0.0};
    auto stat4 = [&](const float4& a, const float4& b) {
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double td = (double)tv[u];
            s1[0] += td;
            s1[1] = fma(td, td, s1[1]);
            s1[2] += (double)fabsf(pv[u] - tv[u]);
        }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rp[r] = ld(pbuf, r);
        rt[r] = ld(tbuf, r);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) stat4(rp[r], rt[r]);
    __builtin_amdgcn_sched_barrier(0);  // later loads stay below: hoisted, they would push the kept rounds into scratch
#pragma unroll
    for (int q = 0; q < L; q += 4) {
        if (R + q >= K) break;  // uniform
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = ld(pbuf, R + q + v);
            b[v] = ld(tbuf, R + q + v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            stat4(a[v], b[v]);
            keep[(2 * (q + v)) * kPB + threadIdx.x] = a[v];
            keep[(2 * (q + v) + 1) * kPB + threadIdx.x] = b[v];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int k = R + L; k < K; k += 4) {  // rounds past K load zeros
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = ld(pbuf, k + v);
            b[v] = ld(tbuf, k + v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) stat4(a[v], b[v]);
    }
    if (j_tail < n) {
        const double td = (double)t[j_tail];
        s1[0] += td;
        s1[1] = fma(td, td, s1[1]);
        s1[2] += (double)fabsf(p[j_tail] - t[j_tail]);
    }
    block_store<3, kPB>(s1, ws + 3 * (int64_t)blockIdx.x);
    grid.sync();  // one release / acquire per workgroup inside (a fence in every wave costs tens of microseconds of L2 write-backs)
    double st[3];
    reduce_partials<3, kPB, true>(ws, G, st);
    const double N = (double)n;
    const LossConsts c = loss_consts(st, N, alpha, beta);

    // ---- pass 2: sum d^2 w and G
    double s2[2] = {0.0, 0.0};
    auto fwd4 = [&](const float4& a, const float4& b) {
        const float pv[4] = {a.x, a.y, a.z, a.w}, tv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d, w, q;
            elem_terms(pv[u], tv[u], c, d, w, q);
            s2[0] += (double)(d * d * w);
            s2[1] += (double)q;
        }
    };
#pragma unroll
    for (int q = 0; q < R; q += 4) {
        if (q >= K) break;  // uniform: small inputs skip the empty rounds
#pragma unroll
        for (int v = 0; v < 4; ++v) fwd4(rp[q + v], rt[q + v]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < L; q += 4) {
        if (R + q >= K) break;
#pragma unroll
        for (int v = 0; v < 4; ++v) fwd4(keep[(2 * (q + v)) * kPB + threadIdx.x], keep[(2 * (q + v) + 1) * kPB + threadIdx.x]);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int k = R + L; k < K; k += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = ld(pbuf, k + v);
            b[v] = ld(tbuf, k + v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) fwd4(a[v], b[v]);
    }
    if (j_tail < n) {
        float d, w, q;
        elem_terms(p[j_tail], t[j_tail], c, d, w, q);
        s2[0] += (double)(d * d * w);
        s2[1] += (double)q;
    }
    double* ws2 = ws + 3 * (int64_t)G;
    block_store<2, kPB>(s2, ws2 + 2 * (int64_t)blockIdx.x);
    grid.sync();
    double pt[2];
    reduce_partials<2, kPB, true>(ws2, G, pt);
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(pt[0] / N);
    if (!grad) return;

    // ---- pass 3: the gradient
    const float g_over_n = (float)(pt[1] / N);
    const float scale = (float)((double)grad_scale / N);
    auto one = [&](const float4& a, const float4& b) {
        float4 o;
        o.x = grad_elem(a.x, b.x, c, g_over_n, scale);
        o.y = grad_elem(a.y, b.y, c, g_over_n, scale);
        o.z = grad_elem(a.z, b.z, c, g_over_n, scale);
        o.w = grad_elem(a.w, b.w, c, g_over_n, scale);
        return o;
    };
    auto st4 = [&](int k, const float4& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, v), gbuf, (int)(voff + (unsigned)k * round_bytes), 0, 0);
    };
#pragma unroll
    for (int q = 0; q < R; q += 4) {
        if (q >= K) break;
#pragma unroll
        for (int v = 0; v < 4; ++v) st4(q + v, one(rp[q + v], rt[q + v]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < L; q += 4) {
        if (R + q >= K) break;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            st4(R + q + v, one(keep[(2 * (q + v)) * kPB + threadIdx.x], keep[(2 * (q + v) + 1) * kPB + threadIdx.x]));
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int k = R + L; k < K; k += 2) {
        const float4 a0 = ld(pbuf, k), b0 = ld(tbuf, k), a1 = ld(pbuf, k + 1), b1 = ld(tbuf, k + 1);
        st4(k, one(a0, b0));
        st4(k + 1, one(a1, b1));
    }
    if (j_tail < n) grad[j_tail] = grad_elem(p[j_tail], t[j_tail], c, g_over_n, scale);
}

constexpr int kMaxLossGrid = 1024;  // 4 workgroups per CU on 256 CUs
constexpr int kMaxDevices = 64;

// Scratch of the allocation-free entry point (mcmc_loss_fused_static): the workspace of mcmc_loss_fused as a file-scope
// device array, 40 KB on every GPU the library runs on, until the process ends.  In host code the NAME is a stand-in,
// not a device address: the per-device address comes from hipGetSymbolAddress at context creation and only that pointer
// is ever handed to a kernel.
static __device__ __align__(256) double scratch[5 * kMaxLossGrid];
static double* g_scratch_ptr[kMaxDevices];  // [device] -> address of that device's copy of `scratch`
static int g_onchip_blocks[kMaxDevices];     // [device] -> co-resident workgroups of loss_onchip_kernel (0: not usable there)

inline int loss_grid(const mcmc_ctx* ctx, int64_t n) {
    int64_t b = (n / 4 + kBlock - 1) / kBlock;
    // few, fat workgroups: every one ends with fp64 atomics on the same 2-3 addresses, which serialise
    const int64_t cap = (int64_t)(ctx->num_cu > 0 ? ctx->num_cu : 256) * 4;
    if (b > cap) b = cap;
    if (b > kMaxLossGrid) b = kMaxLossGrid;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

int launch_loss_stats(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, double* stats, hipStream_t s) {
    hipLaunchKernelGGL(loss_stats_kernel, dim3(loss_grid(ctx, n)), dim3(kBlock), 0, s, p, t, n, stats);
    return check_launch("loss_stats");
}

int launch_loss_forward(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total, const double* stats,
                        float alpha, float beta, double* part, hipStream_t s) {
    hipLaunchKernelGGL(loss_forward_kernel, dim3(loss_grid(ctx, n)), dim3(kBlock), 0, s, p, t, n, (double)n_total, stats, alpha, beta,
                       part);
    return check_launch("loss_forward");
}

int launch_loss_backward(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, int64_t n_total, const double* stats,
                         const double* part, float alpha, float beta, float grad_scale, float* grad, hipStream_t s) {
    hipLaunchKernelGGL(loss_backward_kernel, dim3(loss_grid(ctx, n)), dim3(kBlock), 0, s, p, t, n, (double)n_total, stats, part, alpha,
                       beta, grad_scale, grad);
    return check_launch("loss_backward");
}

int64_t loss_workspace_doubles() { return 5 * (int64_t)kMaxLossGrid; }

// once per device, with that device current (mcmc_ctx_create)
int loss_static_init(int device) {
    if (device < 0 || device >= kMaxDevices) return set_error(MCMC_EINVAL, "device %d beyond the %d this library keeps scratch for", device, kMaxDevices);
    if (g_scratch_ptr[device]) return MCMC_OK;
    double* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return set_error(MCMC_EHIP, "hipGetSymbolAddress(loss scratch) on device %d: %s", device, hipGetErrorString(e));
    g_scratch_ptr[device] = ptr;
    // the persistent loss kernel: 128 KB of dynamic LDS, and its grid must be co-resident (cooperative launch)
    int coop = 0, per_cu = 0, cus = 0;
    const void* fn = reinterpret_cast<const void*>(&loss_onchip_kernel<kKeepRegs, kKeepLds>);
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && coop &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kKeepLdsBytes) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kPB, kKeepLdsBytes) == hipSuccess && per_cu >= 1)
        g_onchip_blocks[device] = cus;  // one workgroup per CU
    (void)hipGetLastError();            // a device without cooperative launch keeps the three-launch form
    return MCMC_OK;
}

// the allocation-free entry point: the three launches on this device's static scratch, no allocation of any kind
static int launch_loss_three(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                             float* loss, float* grad, double* ws, hipStream_t s);

int launch_loss_fused_static(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                             float* loss, float* grad, hipStream_t s) {
    double* ws = (ctx->device >= 0 && ctx->device < kMaxDevices) ? g_scratch_ptr[ctx->device] : nullptr;
    if (!ws) return set_error(MCMC_ESTATE, "no static scratch for device %d (context not created by mcmc_ctx_create?)", ctx->device);
    return launch_loss_three(ctx, p, t, n, alpha, beta, grad_scale, loss, grad, ws, s);
}

// three launches on a caller-held workspace
static int launch_loss_three(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                             float* loss, float* grad, double* ws, hipStream_t s) {
    const int G = loss_grid(ctx, n);
    hipLaunchKernelGGL(fused_stats_kernel, dim3(G), dim3(kBlock), 0, s, p, t, n, ws);
    hipLaunchKernelGGL(fused_forward_kernel, dim3(G), dim3(kBlock), 0, s, p, t, n, alpha, beta, ws);
    hipLaunchKernelGGL(fused_backward_kernel, dim3(grad ? G : 1), dim3(kBlock), 0, s, p, t, n, alpha, beta, grad_scale, G, ws, loss, grad);
    return check_launch("loss_fused");
}

// One persistent cooperative launch (loss_onchip_kernel) where the device offers it, else the three launches.
int launch_loss_fused(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                      float* loss, float* grad, double* ws, hipStream_t s) {
    const int resident = (ctx->device >= 0 && ctx->device < kMaxDevices) ? g_onchip_blocks[ctx->device] : 0;
    if (resident < 1 || n >= kOnchipMaxN) return launch_loss_three(ctx, p, t, n, alpha, beta, grad_scale, loss, grad, ws, s);
    int64_t G = (n / 4 + kPB - 1) / kPB;
    if (G > resident) G = resident;
    if (G > kMaxLossGrid) G = kMaxLossGrid;
    if (G < 1) G = 1;
    void* args[] = {(void*)&p, (void*)&t, (void*)&n, (void*)&alpha, (void*)&beta, (void*)&grad_scale, (void*)&ws, (void*)&loss, (void*)&grad};
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&loss_onchip_kernel<kKeepRegs, kKeepLds>), dim3((unsigned)G),
                                              dim3(kPB), args, (unsigned)kKeepLdsBytes, s);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "loss_onchip (cooperative launch of %lld workgroups): %s", (long long)G, hipGetErrorString(e));
    return check_launch("loss_onchip");
}

}  // namespace mcmc

