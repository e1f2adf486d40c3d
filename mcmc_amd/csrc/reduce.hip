// Piece 2: streaming per-timestep statistics of the MC samples (HBM-bound).
//
// Replaces torch.stack/reshape + mean(0) + std(0) of GPUBayesianEnsemble.predict
// (/root/reference/models/bayesian_ensemble.py:234-246), which materialises all M*S samples
// ([M*S, B] fp32 = 120 GB at config 3) and reads them twice.  Here a bounded chunk Y[C pairs, B]
// is consumed once: per row (sum, sum of squares) in fp64 plus a per-row histogram that yields
// interval quantiles (an extension; the reference computes none - its oracle is torch.quantile).
//
// update: workgroup = 64 rows x 4 waves; wave w takes pairs c = w, w+4, ...; lane = row, so a
//   wave-load is 256 contiguous bytes.  Histogram counts go to an LDS tile with ds_add_u32 on
//   packed uint16 pairs (row-rotated so rows falling in the same bin hit different banks), and
//   are added to the global uint16 histogram once per chunk - the tile owns its rows, so global
//   state is updated without atomics and the result is bitwise deterministic.
#include <type_traits>

#include "common.hpp"

namespace mcmc {

namespace {

constexpr int kRows = 64;   // rows per workgroup
constexpr int kWavesR = 4;  // waves per workgroup
constexpr int kUnroll = 8;  // independent loads in flight per lane

__device__ __forceinline__ int hist_bin(float y, float lo, float inv_w, int nbins) {
    // fp32, two roundings (no fma) so the CPU restatement bins identically
    float t = __fmul_rn(__fsub_rn(y, lo), inv_w);
    t = fminf(fmaxf(t, -1.0f), (float)(nbins - 2));  // NaN -> -1 (fmaxf returns the non-NaN)
    return (int)floorf(t) + 1;
}

template <bool HIST>
__global__ __launch_bounds__(kRows * kWavesR) void reduce_update_kernel(const float* __restrict__ Y, int64_t ldY, int n_pairs,
                                                                       int64_t n_rows, double* __restrict__ sum,
                                                                       double* __restrict__ sumsq,
                                                                       const float* __restrict__ lo_g,
                                                                       const float* __restrict__ inv_w_g,
                                                                       uint32_t* __restrict__ hist, int nbins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem);                                    // [2][kWavesR][kRows]
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem + 2 * kWavesR * kRows * sizeof(double));  // [kRows][nbins/2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * kRows + lane;
    const bool live = row < n_rows;
    const int words = nbins >> 1;  // uint32 words per row (power of two: nbins is 64..512)

    float lo = 0.0f, inv_w = 0.0f;
    if (HIST) {
        for (int i = threadIdx.x; i < kRows * words; i += kRows * kWavesR) lh[i] = 0u;
        if (live) {
            lo = lo_g[row];
            inv_w = inv_w_g[row];
        }
        __syncthreads();
    }

    double s1 = 0.0, s2 = 0.0;
    const float* yp = Y + row;
    int c = wave;
    for (; c + (kUnroll - 1) * kWavesR < n_pairs; c += kUnroll * kWavesR) {
        float v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = live ? __builtin_nontemporal_load(yp + (int64_t)(c + u * kWavesR) * ldY) : 0.0f;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const double d = (double)v[u];
            s1 += d;
            s2 = fma(d, d, s2);
            if (HIST && live) {
                const int b = hist_bin(v[u], lo, inv_w, nbins);
                atomicAdd(&lh[lane * words + (((b >> 1) + lane) & (words - 1))], 1u << ((b & 1) * 16));
            }
        }
    }
    for (; c < n_pairs; c += kWavesR) {
        const float v = live ? __builtin_nontemporal_load(yp + (int64_t)c * ldY) : 0.0f;
        const double d = (double)v;
        s1 += d;
        s2 = fma(d, d, s2);
        if (HIST && live) {
            const int b = hist_bin(v, lo, inv_w, nbins);
            atomicAdd(&lh[lane * words + (((b >> 1) + lane) & (words - 1))], 1u << ((b & 1) * 16));
        }
    }

    red[wave * kRows + lane] = s1;
    red[(kWavesR + wave) * kRows + lane] = s2;
    __syncthreads();
    if (wave == 0 && live) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesR; ++w) {  // fixed order: deterministic
            a += red[w * kRows + lane];
            b += red[(kWavesR + w) * kRows + lane];
        }
        sum[row] += a;
        sumsq[row] += b;
    }
    if (HIST) {
        // global uint16 counts, two per word; no carry can cross the halves while every count
        // stays <= 65535 (the caller bounds total samples per row)
        const int64_t row_base = (int64_t)blockIdx.x * kRows;
        for (int i = threadIdx.x; i < kRows * words; i += kRows * kWavesR) {
            const int rl = i / words, w = i & (words - 1);
            if (row_base + rl < n_rows) {
                const uint32_t add = lh[rl * words + ((w + rl) & (words - 1))];
                if (add) hist[(row_base + rl) * words + w] += add;
            }
        }
    }
}

// 16-byte variant (used whenever Y is 16-B aligned and ldY % 4 == 0): workgroup = 64 rows x 4
// waves like above, but a lane owns 4 adjacent rows (one float4 load) and the 64 lanes of a wave
// are 16 row-quads x 4 pairs, so each wave-load moves 4 x 256 B with a quarter of the load
// instructions.  The histogram tile stays 64 rows (16 KB at 128 bins): 8 workgroups per CU.
constexpr int kSlots = 4 * kWavesR;  // pairs in flight per workgroup step
constexpr int kUnroll4 = 4;

template <bool HIST>
__global__ __launch_bounds__(kRows * kWavesR) void reduce_update4_kernel(const float* __restrict__ Y, int64_t ldY, int n_pairs,
                                                                        int64_t n_rows, double* __restrict__ sum,
                                                                        double* __restrict__ sumsq, const float* __restrict__ lo_g,
                                                                        const float* __restrict__ inv_w_g,
                                                                        uint32_t* __restrict__ hist, int nbins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem);                                             // [2][kWavesR][kRows]
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem + 2 * kWavesR * kRows * sizeof(double));  // [kRows][nbins/2]
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rq = lane & 15, slot = wave * 4 + (lane >> 4);
    const int64_t row_base = (int64_t)blockIdx.x * kRows;
    const int64_t row0 = row_base + 4 * rq;
    const bool loadable = row0 + 3 < ldY;  // the float4 stays inside the row pitch
    const int words = nbins >> 1;

    const bool full_tile = row_base + kRows <= n_rows;  // block-uniform: no per-row bounds checks
    // The tile's state is read NOW and written at the end: its global round trip (the moments of the rows, and for nbins <=
    // 128 the tile's [64 x words] block of the histogram, 16 bytes x 4 per thread) hides behind the whole sample stream
    // instead of standing between the last sample and the first store.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kPre = 4;  // uint4 per thread held across the stream: covers 64 rows x 64 words (128 bins)
    const int n4 = kRows * words / 4;
    const bool prefetch = HIST && full_tile && n4 <= kPre * kRows * kWavesR;
    u32x4 gpre[kPre];
    if (prefetch) {
        const u32x4* hg = reinterpret_cast<const u32x4*>(hist + row_base * words);
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i4 = threadIdx.x + u * kRows * kWavesR;
            gpre[u] = i4 < n4 ? __builtin_nontemporal_load(hg + i4) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    double m1 = 0.0, m2 = 0.0;
    if (wave == 0 && row_base + lane < n_rows) {
        m1 = sum[row_base + lane];
        m2 = sumsq[row_base + lane];
    }
    float lo[4] = {0, 0, 0, 0}, inv_w[4] = {0, 0, 0, 0};
    if (HIST) {
        for (int i = threadIdx.x; i < kRows * words / 4; i += kRows * kWavesR) reinterpret_cast<uint4*>(lh)[i] = uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (row0 + j < n_rows) {
                lo[j] = lo_g[row0 + j];
                inv_w[j] = inv_w_g[row0 + j];
            }
        __syncthreads();
    }

    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const f32x4* yp = reinterpret_cast<const f32x4*>(Y + row0);
    const int64_t ld4 = ldY >> 2;
    // Histogram address arithmetic, 9 VALU instructions per sample: with u = bin + 4*rq (the row
    // quad rotates the word inside its row so the 16 quads of a lane group spread over the banks),
    // word = (u >> 1) mod words, half = u & 1 = bin & 1.
    const float hi_clamp = (float)(nbins - 2);
    const int lw = 31 - __builtin_clz(words);
    auto consume = [&](const f32x4& q, auto tail) {
        const float v[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double d = (double)v[j];
            s1[j] += d;
            s2[j] = fma(d, d, s2[j]);
            if (HIST && (!decltype(tail)::value || row0 + j < n_rows)) {
                // same arithmetic as hist_bin(): two roundings, clamp to [-1, nbins-2] (NaN -> -1), floor
                const float t = __builtin_amdgcn_fmed3f(__fmul_rn(__fsub_rn(v[j], lo[j]), inv_w[j]), -1.0f, hi_clamp);
                int bm1;
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(bm1) : "v"(t));
                const unsigned u = (unsigned)(bm1 + 1 + 4 * rq);
                const unsigned word = __builtin_amdgcn_ubfe(u, 1, lw);
                atomicAdd(&lh[(4 * rq + j) * words + word], (u & 1u) * 65535u + 1u);
            }
        }
    };
    using yes = std::true_type;
    using no = std::false_type;
    if (loadable) {
        if (full_tile) {
            int c = slot;
            for (; c + (kUnroll4 - 1) * kSlots < n_pairs; c += kUnroll4 * kSlots) {
                f32x4 q[kUnroll4];
#pragma unroll
                for (int u = 0; u < kUnroll4; ++u) q[u] = __builtin_nontemporal_load(yp + (int64_t)(c + u * kSlots) * ld4);
#pragma unroll
                for (int u = 0; u < kUnroll4; ++u) consume(q[u], no{});
            }
            for (; c < n_pairs; c += kSlots) consume(__builtin_nontemporal_load(yp + (int64_t)c * ld4), no{});
        } else {
            // the one ragged tile of the launch: per-row bounds checks, no unrolling (keeps the register
            // count of the kernel that of the full-tile loop)
#pragma unroll 1
            for (int c = slot; c < n_pairs; c += kSlots) consume(__builtin_nontemporal_load(yp + (int64_t)c * ld4), yes{});
        }
    }
    // the 4 pair-slots of a wave sit 16 lanes apart: fold them, then the waves through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s1[j] += __shfl_xor(s1[j], 16);
        s1[j] += __shfl_xor(s1[j], 32);
        s2[j] += __shfl_xor(s2[j], 16);
        s2[j] += __shfl_xor(s2[j], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[wave * kRows + 4 * rq + j] = s1[j];
            red[(kWavesR + wave) * kRows + 4 * rq + j] = s2[j];
        }
    }
    __syncthreads();
    if (wave == 0 && row_base + lane < n_rows) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesR; ++w) {  // fixed order: deterministic
            a += red[w * kRows + lane];
            b += red[(kWavesR + w) * kRows + lane];
        }
        sum[row_base + lane] = m1 + a;
        sumsq[row_base + lane] = m2 + b;
    }
    if (HIST) {
        if (prefetch) {
            u32x4* hg = reinterpret_cast<u32x4*>(hist + row_base * words);
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const int i4 = threadIdx.x + u * kRows * kWavesR;
                if (i4 < n4) {
                    const int rl = (4 * i4) / words, w = (4 * i4) & (words - 1), rot = 2 * (rl >> 2);
                    const uint32_t* lr = lh + rl * words;
                    u32x4 g = gpre[u];
                    g.x += lr[(w + rot) & (words - 1)];
                    g.y += lr[(w + 1 + rot) & (words - 1)];
                    g.z += lr[(w + 2 + rot) & (words - 1)];
                    g.w += lr[(w + 3 + rot) & (words - 1)];
                    __builtin_nontemporal_store(g, hg + i4);
                }
            }
        } else if (full_tile) {
            // the tile's 64 rows are one contiguous [64 x words] block of the global histogram: 16-byte
            // read-modify-write, four independent loads in flight per thread (the state pass is pure
            // latency otherwise: every workgroup ends with it)
            u32x4* hg = reinterpret_cast<u32x4*>(hist + row_base * words);
            for (int base = threadIdx.x; base < n4; base += 4 * kRows * kWavesR) {
                u32x4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i4 = base + u * kRows * kWavesR;
                    if (i4 < n4) g[u] = __builtin_nontemporal_load(hg + i4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i4 = base + u * kRows * kWavesR;
                    if (i4 < n4) {
                        const int rl = (4 * i4) / words, w = (4 * i4) & (words - 1), rot = 2 * (rl >> 2);
                        const uint32_t* lr = lh + rl * words;
                        g[u].x += lr[(w + rot) & (words - 1)];
                        g[u].y += lr[(w + 1 + rot) & (words - 1)];
                        g[u].z += lr[(w + 2 + rot) & (words - 1)];
                        g[u].w += lr[(w + 3 + rot) & (words - 1)];
                        __builtin_nontemporal_store(g[u], hg + i4);
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < kRows * words; i += kRows * kWavesR) {
                const int rl = i / words, w = i & (words - 1);
                if (row_base + rl < n_rows) {
                    const uint32_t add = lh[rl * words + ((w + 2 * (rl >> 2)) & (words - 1))];
                    if (add) hist[(row_base + rl) * words + w] += add;
                }
            }
        }
    }
}

// ---- merge of partial states (the local half of the multi-GPU reduce-scatter) ----
// parts[p] are W partial states of the SAME rows, one per rank, as they arrive from the row exchange; the sum runs
// in the fixed order p = 0..W-1 (deterministic, the same on every rank for its rows).
__global__ __launch_bounds__(256) void merge_moments_kernel(int n_parts, int64_t n, const double* __restrict__ parts_a,
                                                            const double* __restrict__ parts_b, double* __restrict__ out_a,
                                                            double* __restrict__ out_b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0, b = 0.0;
        for (int p = 0; p < n_parts; ++p) {
            a += parts_a[(int64_t)p * n + i];
            b += parts_b[(int64_t)p * n + i];
        }
        out_a[i] = a;
        out_b[i] = b;
    }
}

// uint16 counts, two per 32-bit word, four words per thread step; the halves are summed apart in 32 bits, so a
// total above 65,535 is seen (flag raised, count saturated) instead of carrying into its neighbour
typedef uint32_t u32x4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void merge_hist_kernel(int n_parts, int64_t n_vec, const u32x4m* __restrict__ parts,
                                                         u32x4m* __restrict__ out, int* __restrict__ overflow) {
    bool over = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        for (int p = 0; p < n_parts; ++p) {
            const u32x4m w = __builtin_nontemporal_load(parts + (int64_t)p * n_vec + i);
            const uint32_t v[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] += v[j] & 0xFFFFu;
                hi[j] += v[j] >> 16;
            }
        }
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            over |= lo[j] > 0xFFFFu || hi[j] > 0xFFFFu;
            r[j] = (lo[j] > 0xFFFFu ? 0xFFFFu : lo[j]) | ((hi[j] > 0xFFFFu ? 0xFFFFu : hi[j]) << 16);
        }
        out[i] = u32x4m{r[0], r[1], r[2], r[3]};
    }
    if (over && overflow) *overflow = 1;  // plain vector store of the same value from every lane that saw one
}

__global__ __launch_bounds__(256) void set_edges_kernel(int64_t n_rows, double n0, const double* __restrict__ sum,
                                                        const double* __restrict__ sumsq, int nbins, double range_sigmas,
                                                        float* __restrict__ lo, float* __restrict__ inv_w) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const double s1 = sum[row], s2 = sumsq[row];
    const double mean = s1 / n0;
    const double denom = n0 > 1.0 ? n0 - 1.0 : 1.0;
    double var = (s2 - s1 * s1 / n0) / denom;
    if (!(var > 0.0)) var = 0.0;
    double half = range_sigmas * sqrt(var);
    double tiny = fabs(mean) * 1e-6;
    if (tiny < 1e-12) tiny = 1e-12;
    if (!(half > tiny)) half = tiny;
    lo[row] = (float)(mean - half);
    inv_w[row] = (float)((double)(nbins - 2) / (2.0 * half));
}

struct QLevels {
    double q[8];
    int n;
};

// one wave per row: lane holds K = nbins/64 consecutive bins
template <int K>
__global__ __launch_bounds__(256) void finalize_kernel(int64_t n_rows, double n, const double* __restrict__ sum,
                                                       const double* __restrict__ sumsq, const uint16_t* __restrict__ hist,
                                                       const float* __restrict__ lo_g, const float* __restrict__ inv_w_g,
                                                       QLevels ql, float* __restrict__ mean_o, float* __restrict__ std_o,
                                                       float* __restrict__ quant) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int nbins = K * 64;
    for (int64_t row = wave_global; row < n_rows; row += n_waves) {
        if (lane == 0) {
            const double s1 = sum[row], s2 = sumsq[row];
            const double m = s1 / n;
            mean_o[row] = (float)m;
            double sd;
            if (n > 1.0) {
                double var = (s2 - s1 * s1 / n) / (n - 1.0);
                if (!(var > 0.0)) var = 0.0;
                sd = sqrt(var);
            } else {
                sd = __builtin_nan("");  // torch.std of one sample (bayesian_ensemble.py:246)
            }
            std_o[row] = (float)sd;
        }
        if (ql.n == 0) continue;
        uint32_t cnt[K], cum[K];
        uint32_t tot = 0;
        const uint16_t* hp = hist + row * nbins + lane * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            cnt[k] = hp[k];
            tot += cnt[k];
            cum[k] = tot;
        }
        uint32_t incl = tot;  // inclusive scan of lane totals
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const uint32_t excl = incl - tot;
        const double lo = (double)lo_g[row];
        const double width = 1.0 / (double)inv_w_g[row];
        // estimate of the j-th order statistic (0-based): uniform spread inside its bin
        auto order_stat = [&](double j) -> double {
            int kk = -1;
#pragma unroll
            for (int k = K - 1; k >= 0; --k)
                if ((double)(excl + cum[k]) > j) kk = k;
            const unsigned long long ball = __ballot(kk >= 0);
            const int first = ball ? __ffsll((long long)ball) - 1 : 63;
            if (kk < 0) kk = K - 1;  // j beyond all counts (cannot happen when hist sums to n)
            const int b = lane * K + kk;
            const double prev = (double)(excl + (kk > 0 ? cum[kk - 1] : 0u));
            const double cb = (double)(cnt[kk] > 0 ? cnt[kk] : 1u);
            double val = lo + ((double)(b - 1) + (j + 0.5 - prev) / cb) * width;
            if (b == 0) val = lo;
            if (b == nbins - 1) val = lo + (double)(nbins - 2) * width;
            return __shfl(val, first);
        };
        for (int qi = 0; qi < ql.n; ++qi) {
            // torch.quantile 'linear': position q*(n-1) between order statistics k and k+1
            const double pos = ql.q[qi] * (n - 1.0);
            const double k0 = floor(pos);
            const double frac = pos - k0;
            double val = order_stat(k0);
            if (frac > 0.0 && k0 + 1.0 < n) val += frac * (order_stat(k0 + 1.0) - val);
            if (lane == 0) quant[(int64_t)qi * n_rows + row] = (float)val;
        }
    }
}

}  // namespace

int launch_reduce_update(const mcmc_ctx* ctx, const float* Y, int64_t ldY, int n_pairs, int64_t n_rows, double* sum,
                         double* sumsq, const float* lo, const float* inv_w, uint16_t* hist, int nbins, hipStream_t s) {
    (void)ctx;
    const bool wide = (reinterpret_cast<uintptr_t>(Y) & 15u) == 0 && (ldY & 3) == 0;  // else: the 4-byte-per-lane kernel
    if (wide) {
        const int64_t blocks = (n_rows + kRows - 1) / kRows;
        const size_t red_bytes = 2 * kWavesR * kRows * sizeof(double);
        if (hist) {
            const size_t lds = red_bytes + (size_t)kRows * (nbins / 2) * sizeof(uint32_t);
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(reduce_update4_kernel<true>), lds, "reduce_update4")) return rc;
            hipLaunchKernelGGL(reduce_update4_kernel<true>, dim3((unsigned)blocks), dim3(kRows * kWavesR), lds, s, Y, ldY, n_pairs,
                               n_rows, sum, sumsq, lo, inv_w, reinterpret_cast<uint32_t*>(hist), nbins);
        } else {
            hipLaunchKernelGGL(reduce_update4_kernel<false>, dim3((unsigned)blocks), dim3(kRows * kWavesR), red_bytes, s, Y, ldY,
                               n_pairs, n_rows, sum, sumsq, nullptr, nullptr, nullptr, 0);
        }
        return check_launch("reduce_update");
    }
    const int64_t blocks = (n_rows + kRows - 1) / kRows;
    const size_t red_bytes = 2 * kWavesR * kRows * sizeof(double);
    if (hist) {
        const size_t lds = red_bytes + (size_t)kRows * (nbins / 2) * sizeof(uint32_t);
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(reduce_update_kernel<true>), lds, "reduce_update")) return rc;
        hipLaunchKernelGGL(reduce_update_kernel<true>, dim3((unsigned)blocks), dim3(kRows * kWavesR), lds, s, Y, ldY, n_pairs,
                           n_rows, sum, sumsq, lo, inv_w, reinterpret_cast<uint32_t*>(hist), nbins);
    } else {
        hipLaunchKernelGGL(reduce_update_kernel<false>, dim3((unsigned)blocks), dim3(kRows * kWavesR), red_bytes, s, Y, ldY,
                           n_pairs, n_rows, sum, sumsq, nullptr, nullptr, nullptr, 0);
    }
    return check_launch("reduce_update");
}

int launch_reduce_merge(int n_parts, int64_t n_rows, const double* sum_parts, const double* sumsq_parts, const uint16_t* hist_parts,
                        int nbins, double* sum, double* sumsq, uint16_t* hist, int* overflow, hipStream_t s) {
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(merge_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n_parts, n_rows, sum_parts, sumsq_parts, sum, sumsq);
    if (hist) {
        const int64_t n_vec = n_rows * nbins / 8;  // nbins is a multiple of 64: whole uint4s per row
        blocks = (n_vec + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(merge_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n_parts, n_vec, reinterpret_cast<const u32x4m*>(hist_parts),
                           reinterpret_cast<u32x4m*>(hist), overflow);
    }
    return check_launch("reduce_merge");
}

int launch_set_edges(int64_t n_rows, int64_t n0, const double* sum, const double* sumsq, int nbins, float range_sigmas,
                     float* lo, float* inv_w, hipStream_t s) {
    hipLaunchKernelGGL(set_edges_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, n_rows, (double)n0, sum, sumsq,
                       nbins, (double)range_sigmas, lo, inv_w);
    return check_launch("reduce_set_edges");
}

int launch_finalize(int64_t n_rows, int64_t n, const double* sum, const double* sumsq, const uint16_t* hist, const float* lo,
                    const float* inv_w, int nbins, const double* q_levels, int n_q, float* mean, float* std, float* quant,
                    hipStream_t s) {
    QLevels ql;
    ql.n = n_q;
    for (int i = 0; i < 8; ++i) ql.q[i] = i < n_q ? q_levels[i] : 0.0;
    int64_t blocks = (n_rows + 3) / 4;  // 4 waves per block, one row per wave per trip
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
#define MCMC_FIN(K)                                                                                                         \
    hipLaunchKernelGGL(finalize_kernel<K>, dim3((unsigned)blocks), dim3(256), 0, s, n_rows, (double)n, sum, sumsq, hist, lo, \
                       inv_w, ql, mean, std, quant)
    switch (n_q ? nbins : 64) {
        case 64: MCMC_FIN(1); break;
        case 128: MCMC_FIN(2); break;
        case 256: MCMC_FIN(4); break;
        case 512: MCMC_FIN(8); break;
        default: return set_error(MCMC_EINVAL, "nbins must be 64, 128, 256 or 512 (got %d)", nbins);
    }
#undef MCMC_FIN
    return check_launch("reduce_finalize");
}

}  // namespace mcmc
