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
#include "common.hpp"

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
template <int NV>
__device__ __forceinline__ void block_store(double (&v)[NV], double* __restrict__ slot) {
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
        slot[threadIdx.x] = a;
    }
}

// totals of NV interleaved partial columns over G blocks, the same bits in every block
template <int NV>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ part, int G, double (&tot)[NV]) {
    __shared__ double sm[NV][kBlock / 64];
    __shared__ double res[NV];
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.0;
    for (int b = threadIdx.x; b < G; b += kBlock)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += part[(int64_t)b * NV + i];
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

constexpr int kMaxLossGrid = 1024;  // 4 workgroups per CU on 256 CUs
constexpr int kMaxDevices = 64;

// Scratch of the allocation-free entry point (mcmc_loss_fused_static): the workspace of mcmc_loss_fused as a file-scope
// device array, 40 KB on every GPU the library runs on, until the process ends.  In host code the NAME is a stand-in,
// not a device address: the per-device address comes from hipGetSymbolAddress at context creation and only that pointer
// is ever handed to a kernel.
static __device__ __align__(256) double scratch[5 * kMaxLossGrid];
static double* g_scratch_ptr[kMaxDevices];  // [device] -> address of that device's copy of `scratch`

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
    return MCMC_OK;
}

// the three launches of launch_loss_fused on this device's static scratch: no allocation of any kind
int launch_loss_fused_static(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                             float* loss, float* grad, hipStream_t s) {
    double* ws = (ctx->device >= 0 && ctx->device < kMaxDevices) ? g_scratch_ptr[ctx->device] : nullptr;
    if (!ws) return set_error(MCMC_ESTATE, "no static scratch for device %d (context not created by mcmc_ctx_create?)", ctx->device);
    return launch_loss_fused(ctx, p, t, n, alpha, beta, grad_scale, loss, grad, ws, s);
}

int launch_loss_fused(const mcmc_ctx* ctx, const float* p, const float* t, int64_t n, float alpha, float beta, float grad_scale,
                      float* loss, float* grad, double* ws, hipStream_t s) {
    const int G = loss_grid(ctx, n);
    hipLaunchKernelGGL(fused_stats_kernel, dim3(G), dim3(kBlock), 0, s, p, t, n, ws);
    hipLaunchKernelGGL(fused_forward_kernel, dim3(G), dim3(kBlock), 0, s, p, t, n, alpha, beta, ws);
    hipLaunchKernelGGL(fused_backward_kernel, dim3(grad ? G : 1), dim3(kBlock), 0, s, p, t, n, alpha, beta, grad_scale, G, ws, loss, grad);
    return check_launch("loss_fused");
}

}  // namespace mcmc
