// bf16 fused forward of BayesianNetwork for (member,sample) pairs - the throughput path.
//
// Restates /root/reference/models/bayesian_ensemble.py:73-92 in eval mode (what predict() runs,
// :215-216: dropout is the identity) on pre-sampled effective weights.  One workgroup = one
// pair x one row superblock; its 8 waves (2 per SIMD) keep the pair's 89 KB of bf16 weight
// fragments resident in LDS and stream 32-row tiles past them:
//     Z^T[out, batch] = W[out, k] . H^T[k, batch]          v_mfma_f32_32x32x16_bf16
// The accumulator of one layer (feature f(i,h) in register i, batch row on the lane) is
// converted to bf16 in place and IS the next layer's B operand - activations never touch LDS or
// HBM; only A fragments (weights, pre-permuted by sample_weights.hip) are read from LDS, one
// ds_read_b128 per MFMA.  88 MFMAs per 32 rows = 45,056 MAC/row for 43,840 algorithmic (97.3 %).
//
// ELU runs in the log2e-scaled domain (layout.hpp): a = log2e*z,
//     phi(a) = max(a, log2e * 2^min(a,0) - log2e) = log2e * elu(z)
// = v_exp_f32 (clamp) + v_fma_f32 + v_max_f32 per element, the budget that keeps the VALU under
// the MFMA time at 2 waves per SIMD.
#include <stdlib.h>

#include "common.hpp"
#include "dropmask.hpp"

namespace mcmc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr float kLog2e = 1.4426950408889634f;
// Waves per workgroup (= per CU: the 89 KB weight image admits one workgroup per CU).  Measured on
// MI355X (365k rows x 1024 pairs): 8 waves 917, 12 waves 963, 16 waves 973 TFLOP/s - the kernel is
// VALU-issue-bound (ELU), more resident waves only close the scheduling gaps.
constexpr int kDefaultWaves = 16;

// max(a, e) as one v_med3_f32 against FLT_MAX (saturating at FLT_MAX): a target intrinsic the
// compiler cannot rewrite into maxnum, so it neither canonicalises the MFMA output first (plain
// fmaxf costs a second v_max per element) nor pads an inline-asm boundary
__device__ __forceinline__ float max_nc(float a, float e) { return __builtin_amdgcn_fmed3f(a, e, 3.402823466e+38f); }

// phi over a 16-register accumulator tile, staged in groups of 8 (exps, then fmas, then maxes)
// so no instruction waits on the one before it
__device__ __forceinline__ void phi16(const f32x16& acc, float (&o)[16]) {
#if defined(MCMC_ABLATE) && MCMC_ABLATE == 1  // tools/fwd_ablate.hip: no activation at all
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = acc[i];
#else
#ifndef MCMC_PHI_GROUP
#define MCMC_PHI_GROUP 8
#endif
#pragma unroll
    for (int g = 0; g < 16; g += MCMC_PHI_GROUP) {
        float t[MCMC_PHI_GROUP];
#pragma unroll
        for (int i = 0; i < MCMC_PHI_GROUP; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(acc[g + i]), 0.0f, 1.0f);  // v_exp_f32 clamp
#pragma unroll
        for (int i = 0; i < MCMC_PHI_GROUP; ++i) t[i] = fmaf(t[i], kLog2e, -kLog2e);
#pragma unroll
        for (int i = 0; i < MCMC_PHI_GROUP; ++i) o[g + i] = max_nc(acc[g + i], t[i]);
    }
#endif
}

__device__ __forceinline__ void pack16(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
    float o[16];
    phi16(acc, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (__bf16)o[j];
        hi[j] = (__bf16)o[8 + j];
    }
}

// phi + dropout + bf16 pack of hidden-layer tile `mo` (MODE 0: exactly pack16)
template <int MODE>
__device__ __forceinline__ void pack16_drop(const f32x16& acc, bf16x8& lo, bf16x8& hi, const DropMask& dm, int layer, int mo, int h,
                                            int c, uint32_t g, int64_t n_rows, int64_t row) {
    float o[16];
    phi16(acc, o);
    drop16<MODE>(o, dm, layer, mo, h, c, g, n_rows, row);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (__bf16)o[j];
        hi[j] = (__bf16)o[8 + j];
    }
}

template <int KS0, int kWaves>
__global__ __launch_bounds__(64 * kWaves, kWaves / 4) void forward_bf16_kernel(const unsigned short* __restrict__ xpack,
                                                                      int64_t n_rows, int F, int rows_per_block,
                                                                      const unsigned char* __restrict__ wpack,
                                                                      float* __restrict__ Y, int64_t ldY) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const WPackBF16 L(F);
    constexpr int ks0 = KS0;
    // pair on blockIdx.x: the workgroups resident at one time share a row superblock, so its X
    // tile (512 KB) is an L2 hit for all but the first of them
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;

    {  // stage this pair's fragments: linear 16-B copy (the image is already lane-linear)
        const uint4* src = reinterpret_cast<const uint4*>(wpack + (int64_t)c * L.total);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(L.total / 16); i += 64 * kWaves) dst[i] = src[i];
    }
    __syncthreads();

    const bf16x8* A0 = reinterpret_cast<const bf16x8*>(lds + L.l0) + lane;
    const bf16x8* A1 = reinterpret_cast<const bf16x8*>(lds + L.l1) + lane;
    const bf16x8* A2 = reinterpret_cast<const bf16x8*>(lds + L.l2) + lane;
    const float* B1 = reinterpret_cast<const float*>(lds + L.b1) + h * 16;
    const float* B2 = reinterpret_cast<const float*>(lds + L.b2) + h * 16;
    const float* W3 = reinterpret_cast<const float*>(lds + L.w3) + h * 16;
    const float b3 = *reinterpret_cast<const float*>(lds + L.b3);

    const int64_t padded = rows_pad(n_rows);
    const int64_t blk_begin = (int64_t)blockIdx.y * rows_per_block;
    int64_t blk_end = blk_begin + rows_per_block;
    if (blk_end > padded) blk_end = padded;

    for (int64_t row0 = blk_begin + 32 * wave; row0 < blk_end; row0 += 32 * kWaves) {
        const int64_t row = row0 + r;
        // the weight image is loop-invariant: without this the compiler hoists all 88 A
        // fragments (352 VGPRs) out of the tile loop and spills
        asm volatile("" ::: "memory");
        bf16x8 xb[KS0];
#pragma unroll
        for (int s = 0; s < KS0; ++s)
            xb[s] = *reinterpret_cast<const bf16x8*>(xpack + row * L.K0 + 16 * s + 8 * h);

        // ---- layer 0: input_dim(+1) -> 256 ----
        bf16x8 hb1[16];
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) {
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < KS0; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[(mo * ks0 + s) * 64], xb[s], acc, 0, 0, 0);
            pack16(acc, hb1[2 * mo], hb1[2 * mo + 1]);
        }
        // ---- layer 1: 256 -> 128 ----
        bf16x8 hb2[8];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = B1[mo * 32 + i];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[(mo * 16 + s) * 64], hb1[s], acc, 0, 0, 0);
            pack16(acc, hb2[2 * mo], hb2[2 * mo + 1]);
        }
        // ---- layer 2: 128 -> 64, layer 3: 64 -> 1 on the VALU ----
        float y = 0.0f;
#pragma unroll
        for (int mo = 0; mo < 2; ++mo) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = B2[mo * 32 + i];
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[(mo * 8 + s) * 64], hb2[s], acc, 0, 0, 0);
            float o[16];
            phi16(acc, o);
#pragma unroll
            for (int i = 0; i < 16; ++i) y = fmaf(W3[mo * 32 + i], o[i], y);
        }
        y += __shfl_xor(y, 32);
        y += b3;
        if (h == 0 && row < n_rows) Y[(int64_t)c * ldY + row] = y;
    }
}

// Variant with T row tiles per wave: every A fragment read from LDS feeds T MFMAs (T x fewer LDS bytes and
// ds_read instructions per row), and the T independent tiles give the wave its own instruction-level
// parallelism, so fewer, fatter waves (256 registers at 8 waves per CU) keep the SIMD's issue port fed.
// Layers are interleaved explicitly: each 32-feature tile of a layer is activated, packed and consumed by the
// next layer's MFMAs at once, so only one tile's bf16 fragments are live (same accumulation order as above:
// bitwise-identical results).
template <int KS0, int kWaves, int T, int MODE>
__global__ __launch_bounds__(64 * kWaves, kWaves / 4) void forward_bf16_tiles_kernel(const unsigned short* __restrict__ xpack,
                                                                            int64_t n_rows, int F, int rows_per_block,
                                                                            const unsigned char* __restrict__ wpack,
                                                                            float* __restrict__ Y, int64_t ldY, DropMask dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const WPackBF16 L(F);
    constexpr int ks0 = KS0;
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    {
        const uint4* src = reinterpret_cast<const uint4*>(wpack + (int64_t)c * L.total);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(L.total / 16); i += 64 * kWaves) dst[i] = src[i];
    }
    __syncthreads();
    const bf16x8* A0 = reinterpret_cast<const bf16x8*>(lds + L.l0) + lane;
    const bf16x8* A1 = reinterpret_cast<const bf16x8*>(lds + L.l1) + lane;
    const bf16x8* A2 = reinterpret_cast<const bf16x8*>(lds + L.l2) + lane;
    const float* B1 = reinterpret_cast<const float*>(lds + L.b1) + h * 16;
    const float* B2 = reinterpret_cast<const float*>(lds + L.b2) + h * 16;
    const float* W3 = reinterpret_cast<const float*>(lds + L.w3) + h * 16;
    const float b3 = *reinterpret_cast<const float*>(lds + L.b3);

    const int64_t padded = rows_pad(n_rows);
    const int64_t blk_begin = (int64_t)blockIdx.y * rows_per_block;
    int64_t blk_end = blk_begin + rows_per_block;
    if (blk_end > padded) blk_end = padded;

    const uint32_t g = MODE == 2 ? drop_pair_id(dm, c) : 0u;
    for (int64_t row0 = blk_begin + 32 * T * wave; row0 < blk_end; row0 += 32 * T * kWaves) {
        asm volatile("" ::: "memory");
        bf16x8 xb[T][KS0];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int64_t row = row0 + 32 * t + r;
            if (row >= padded) row = padded - 1;  // a tile past the end recomputes the last row; nothing is stored
#pragma unroll
            for (int s = 0; s < KS0; ++s) xb[t][s] = *reinterpret_cast<const bf16x8*>(xpack + row * L.K0 + 16 * s + 8 * h);
        }
        // ---- layers 0 and 1 interleaved ----
        f32x16 acc1[T][4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float b = B1[mo * 32 + i];
#pragma unroll
                for (int t = 0; t < T; ++t) acc1[t][mo][i] = b;
            }
#pragma unroll
        for (int m0 = 0; m0 < 8; ++m0) {
            bf16x8 lo[T], hi[T];
            f32x16 a0[T];
#pragma unroll
            for (int t = 0; t < T; ++t) a0[t] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS0; ++s) {
                const bf16x8 fa = A0[(m0 * ks0 + s) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) a0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xb[t][s], a0[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) pack16_drop<MODE>(a0[t], lo[t], hi[t], dm, 0, m0, h, c, g, n_rows, row0 + 32 * t + r);
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) {
                const bf16x8 fa = A1[(mo * 16 + 2 * m0) * 64], fb = A1[(mo * 16 + 2 * m0 + 1) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) acc1[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lo[t], acc1[t][mo], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < T; ++t) acc1[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, hi[t], acc1[t][mo], 0, 0, 0);
            }
        }
        // ---- layers 1 (activation) and 2 interleaved ----
        f32x16 acc2[T][2];
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float b = B2[mo * 32 + i];
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][mo][i] = b;
            }
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) {
            bf16x8 lo[T], hi[T];
#pragma unroll
            for (int t = 0; t < T; ++t) pack16_drop<MODE>(acc1[t][m1], lo[t], hi[t], dm, 1, m1, h, c, g, n_rows, row0 + 32 * t + r);
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                const bf16x8 fa = A2[(mo * 8 + 2 * m1) * 64], fb = A2[(mo * 8 + 2 * m1 + 1) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lo[t], acc2[t][mo], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, hi[t], acc2[t][mo], 0, 0, 0);
            }
        }
        // ---- layer 2 activation, layer 3 on the VALU ----
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float y = 0.0f;
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                float o[16];
                phi16(acc2[t][mo], o);
                drop16<MODE>(o, dm, 2, mo, h, c, g, n_rows, row0 + 32 * t + r);
#pragma unroll
                for (int i = 0; i < 16; ++i) y = fmaf(W3[mo * 32 + i], o[i], y);
            }
            y += __shfl_xor(y, 32);
            y += b3;
            const int64_t row = row0 + 32 * t + r;
            if (h == 0 && row < n_rows && row0 + 32 * t < blk_end) Y[(int64_t)c * ldY + row] = y;
        }
    }
}

template <int KS0, int kWaves, int T = 1, int MODE = 0>
static int launch_variant(const void* xpack, int64_t n_rows, int F, const void* wpack, int pair_count, float* Y, int64_t ldY,
                          hipStream_t s, const DropMask& dm = DropMask{}) {
    const WPackBF16 L(F);
    const size_t lds_bytes = (size_t)L.total;
    const void* kern;
    constexpr bool kPlain = T == 1 && MODE == 0;  // the one-tile kernel without a mask stage
    if constexpr (kPlain) kern = reinterpret_cast<const void*>(forward_bf16_kernel<KS0, kWaves>);
    else kern = reinterpret_cast<const void*>(forward_bf16_tiles_kernel<KS0, kWaves, T, MODE>);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return set_error(MCMC_EHIP, "hipFuncSetAttribute(forward_bf16, %zu B LDS): %s", lds_bytes, hipGetErrorString(e));
    // Row superblock: large enough that staging 89 KB of weights is ~1 % of its MFMA time
    // (64 tiles per wave at 8 waves), small enough that pairs x superblocks >> 256 CUs.
    const int64_t padded = rows_pad(n_rows);
    const int step = 32 * kWaves * T;
#ifndef MCMC_ROWS_PER_BLOCK
#define MCMC_ROWS_PER_BLOCK 16384
#endif
    int rows_per_block = MCMC_ROWS_PER_BLOCK / step * step;
    if (padded < rows_per_block) rows_per_block = (int)((padded + step - 1) / step * step);
    const int64_t row_blocks = (padded + rows_per_block - 1) / rows_per_block;
    if (row_blocks > 65535) return set_error(MCMC_EINVAL, "n_rows too large for one launch (%lld row blocks)", (long long)row_blocks);
    for (int done = 0; done < pair_count;) {
        const int n = pair_count - done > 32768 ? 32768 : pair_count - done;
        const unsigned char* wp = (const unsigned char*)wpack + (int64_t)done * L.total;
        float* y = Y + (int64_t)done * ldY;
        if constexpr (kPlain) {
            hipLaunchKernelGGL((forward_bf16_kernel<KS0, kWaves>), dim3(n, (unsigned)row_blocks), dim3(64 * kWaves), lds_bytes, s,
                               (const unsigned short*)xpack, n_rows, F, rows_per_block, wp, y, ldY);
        } else {
            DropMask d = dm;  // this slab's pairs: host masks and the pair ids advance with it
            if (MODE == 1) {
                d.m[0] += (int64_t)done * n_rows * H1;
                d.m[1] += (int64_t)done * n_rows * H2;
                d.m[2] += (int64_t)done * n_rows * H3;
            }
            d.pair_begin += done;
            hipLaunchKernelGGL((forward_bf16_tiles_kernel<KS0, kWaves, T, MODE>), dim3(n, (unsigned)row_blocks), dim3(64 * kWaves),
                               lds_bytes, s, (const unsigned short*)xpack, n_rows, F, rows_per_block, wp, y, ldY, d);
        }
        done += n;
    }
    return check_launch("forward_bf16");
}

}  // namespace

int launch_forward_bf16(const mcmc_ctx* ctx, const void* xpack, int64_t n_rows, int F, const void* wpack, int pair_count,
                        const DropMask* dm, float* Y, int64_t ldY, hipStream_t s) {
    (void)ctx;
    if (dm && dm->mode) {  // train-mode dropout: the two-tile kernel with the mask stage compiled in
        const bool k16 = WPackBF16(F).K0 == 16;
        if (dm->mode == 1) {
            if (k16) return launch_variant<1, 8, 2, 1>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s, *dm);
            return launch_variant<2, 8, 2, 1>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s, *dm);
        }
        if (k16) return launch_variant<1, 8, 2, 2>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s, *dm);
        return launch_variant<2, 8, 2, 2>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s, *dm);
    }
    static const int waves = [] {
        const char* e = getenv("MCMC_FWD_WAVES");  // tuning knob; default chosen by measurement
        const int w = e ? atoi(e) : kDefaultWaves;
        return (w == 8 || w == 12 || w == 16) ? w : kDefaultWaves;  // 16 falls back to 12 when input_dim > 14
    }();
    const bool one = WPackBF16(F).K0 == 16;
    // Default: two row tiles per wave at 8 waves per workgroup (measured 33.4 vs 34.0 ms for one tile per wave at
    // 16 waves, 365k rows x 1024 pairs).  MCMC_FWD_TILES=1 selects the one-tile kernel (dev knob).
    static const int tiles = [] {
        const char* e = getenv("MCMC_FWD_TILES");
        return e ? atoi(e) : 2;
    }();
    if (tiles != 1) {
        if (one) return launch_variant<1, 8, 2>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s);
        return launch_variant<2, 8, 2>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s);
    }
#define MCMC_GO(KS, W) return launch_variant<KS, W>(xpack, n_rows, F, wpack, pair_count, Y, ldY, s)
    if (one) {
        if (waves == 16) MCMC_GO(1, 16);
        if (waves == 12) MCMC_GO(1, 12);
        MCMC_GO(1, 8);
    }
    if (waves >= 12) MCMC_GO(2, 12);  // two layer-0 k-steps need 160 VGPRs: 16 waves do not fit
    MCMC_GO(2, 8);
#undef MCMC_GO
}

}  // namespace mcmc
