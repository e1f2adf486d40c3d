// bf16 fused forward of BayesianNetwork for (member,sample) pairs - the throughput path.
//
// Restates /root/reference/models/bayesian_ensemble.py:73-92 in eval mode (what predict() runs,
// :215-216: dropout is the identity) on pre-sampled effective weights.  One workgroup = one
// pair x one row superblock; its 8 waves (2 per SIMD) keep the pair's ~90 KB of bf16 weight
// fragments resident in LDS and each wave streams two 32-row tiles at a time past them:
//     Z^T[out, batch] = W[out, k] . H^T[k, batch]          v_mfma_f32_32x32x16_bf16
// Every A fragment read from LDS (weights, pre-permuted by sample_weights.hip) feeds two MFMAs, one
// per tile.  The accumulator of one layer (feature f(i,h) in register i, batch row on the lane) is
// activated, packed to bf16 in place and IS the next layer's B operand - activations never touch
// LDS or HBM.  Layers are interleaved: a 32-feature tile of layer l is consumed by layer l+1's
// MFMAs at once, so only one tile's fragments are live.
//
// Two precisions of the activation, one kernel each (MCMC_DT_BF16 / MCMC_DT_BF16_FAST, layout.hpp):
//
// exact - ELU in the log2e-scaled domain: a = log2e*z, phi(a) = max(a, log2e*2^min(a,0) - log2e) = log2e*elu(z):
// v_exp_f32 (clamp) + v_fma_f32 + v_med3_f32 per element + half a v_cvt_pk_bf16_f32 = 18.75 issue cycles, in a kernel
// bound by the SIMD's vector issue port.  Layer 3 (64 -> 1) is 32 VALU FMAs.  Carries the dropout-mask
// stage (train-mode forward, MC-dropout predict) as separate template instances.
//
// fast - ELU in the SHIFTED log2e domain: the accumulators hold u = a + c (c = log2e) and the bf16 handed on is
//     H = max(u, c * 2^min(a,0)) = c * (elu(z) + 1)  > 0,
// whose constant part the next layer's bias absorbs.  H is positive, so on bf16 BIT PATTERNS the max is an integer
// max and the exponential can be built in the exponent field (Schraudolph 1999): bits(c*2^a) ~ 128*a + K.
// Per two elements: 2 v_fma_f32, v_cvt_pknorm_u16_f32 (round, saturate at 0, pack), v_pk_min_u16 (cap at bits(c):
// a > 0), v_cvt_pk_bf16_f32 (the linear branch), v_pk_max_i16 = 12.5 issue cycles per element, no transcendental.
// The constructed mantissa is linear where 2^f is not: a relative error of +-3 % of e^z (centred by kFieldShift),
// |elu error| <= 0.03; its effect on the predictive mean / std is measured in DESIGN.md 4.2.  Layer 3 runs on the
// MFMA too (output row 0 of a 32-row A tile), since H is already packed.
#include "common.hpp"
#include "dropmask.hpp"

namespace mcmc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 8;  // per workgroup = per CU: the weight image admits one workgroup per CU
#ifdef MCMC_FWD_STAMP
// Diagnostic build only (tools/fwd_ablate.hip -DMCMC_FWD_STAMP; never defined for the library): every workgroup stamps
// s_memtime / s_memrealtime once around its tile loop into this array, which nothing else reads - the in-kernel clock is
// d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long g_fwd_stamp[2 * 65536];
#endif
constexpr int kTiles = 2;  // 32-row tiles per wave

// bits(c * 2^a) ~ 128 * (a + log2(c) + 127) + kFieldShift,  a = u - c;  as the fraction of 65535 that
// v_cvt_pknorm_u16_f32 scales by
constexpr float kFieldK1 = 128.0f / 65535.0f;
constexpr float kFieldK0 = (float)((128.0 * (127.0 + 0.5287663729448977 - 1.4426950408889634) + (double)kFieldShift) / 65535.0);
constexpr unsigned short kCapBits = 0x3FB9;  // bf16(log2e), round to nearest

// v_cvt_pk_bf16_f32 as compiler-visible code (not inline asm: the hazard recognizer must see a VALU read of
// MFMA results to place the wait states the hardware does not interlock)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// max(a, e) as one v_med3_f32 against FLT_MAX: a target intrinsic the compiler cannot rewrite into maxnum,
// so it does not canonicalise the MFMA output first (plain fmaxf costs a second v_max per element)
__device__ __forceinline__ float max_nc(float a, float e) { return __builtin_amdgcn_fmed3f(a, e, 3.402823466e+38f); }

// phi(a) = max(a, c * 2^min(a,0) - c) = c * elu(z) over a 16-register accumulator tile, staged in groups of 8
// (exps, then fmas, then maxes) so no instruction waits on the one before it
__device__ __forceinline__ void phi16(const f32x16& acc, float (&o)[16]) {
#pragma unroll
    for (int g = 0; g < 16; g += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(acc[g + i]), 0.0f, 1.0f);  // v_exp_f32 clamp
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = fmaf(t[i], kLog2e, -kLog2e);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[g + i] = max_nc(acc[g + i], t[i]);
    }
}

// phi + dropout + bf16 pack of hidden-layer tile `mo` (exact path)
template <int MODE>
__device__ __forceinline__ void pack16_exact(const f32x16& acc, bf16x8& lo, bf16x8& hi, const DropMask& dm, int layer, int mo, int h,
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

template <int KS0, int MODE>
__global__ __launch_bounds__(64 * kWaves, kWaves / 4) void forward_bf16_exact_kernel(const unsigned short* __restrict__ xpack,
                                                                            int64_t n_rows, int F, int rows_per_block,
                                                                            const unsigned char* __restrict__ wpack,
                                                                            float* __restrict__ Y, int64_t ldY, DropMask dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int T = kTiles;
    const WPackBF16 L(F);
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
                const bf16x8 fa = A0[(m0 * KS0 + s) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) a0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xb[t][s], a0[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) pack16_exact<MODE>(a0[t], lo[t], hi[t], dm, 0, m0, h, c, g, n_rows, row0 + 32 * t + r);
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
            for (int t = 0; t < T; ++t) pack16_exact<MODE>(acc1[t][m1], lo[t], hi[t], dm, 1, m1, h, c, g, n_rows, row0 + 32 * t + r);
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

// two accumulator values u -> two bf16 of H (eval mode)
__device__ __forceinline__ unsigned elu_field2(float u0, float u1) {
    const float x0 = fmaf(u0, kFieldK1, kFieldK0), x1 = fmaf(u1, kFieldK1, kFieldK0);
    const u16x2 y = __builtin_bit_cast(u16x2, __builtin_amdgcn_cvt_pknorm_u16(x0, x1));
    const u16x2 e = __builtin_elementwise_min(y, u16x2{kCapBits, kCapBits});
    const i16x2 p = __builtin_bit_cast(i16x2, cvt_pk_bf16(u0, u1));  // negative when u < 0: loses the signed max
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(p, __builtin_bit_cast(i16x2, e)));
}

// exponent-field ELU + bf16 pack of a 16-register accumulator tile
__device__ __forceinline__ void pack16_field(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = elu_field2(acc[2 * j], acc[2 * j + 1]);
    lo = __builtin_bit_cast(bf16x8, u32x4v{w[0], w[1], w[2], w[3]});
    hi = __builtin_bit_cast(bf16x8, u32x4v{w[4], w[5], w[6], w[7]});
}

template <int KS0>
__global__ __launch_bounds__(64 * kWaves, kWaves / 4) void forward_bf16_fast_kernel(const unsigned short* __restrict__ xpack, int64_t n_rows,
                                                                               int F, int rows_per_block,
                                                                               const unsigned char* __restrict__ wpack,
                                                                               float* __restrict__ Y, int64_t ldY) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int T = kTiles;
    const WPackBF16F L(F);
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
    const bf16x8* A3 = reinterpret_cast<const bf16x8*>(lds + L.l3) + lane;
    const float* B1 = reinterpret_cast<const float*>(lds + L.b1) + h * 16;
    const float* B2 = reinterpret_cast<const float*>(lds + L.b2) + h * 16;
    const float b3 = *reinterpret_cast<const float*>(lds + L.b3);

    const int64_t padded = rows_pad(n_rows);
    const int64_t blk_begin = (int64_t)blockIdx.y * rows_per_block;
    int64_t blk_end = blk_begin + rows_per_block;
    if (blk_end > padded) blk_end = padded;

#ifdef MCMC_FWD_STAMP
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) alone
#endif
    for (int64_t row0 = blk_begin + 32 * T * wave; row0 < blk_end; row0 += 32 * T * kWaves) {
        // the weight image is loop-invariant: without this the compiler hoists the A fragments
        // (352 VGPRs) out of the tile loop and spills
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
                const bf16x8 fa = A0[(m0 * KS0 + s) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) a0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xb[t][s], a0[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) pack16_field(a0[t], lo[t], hi[t]);
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
            for (int t = 0; t < T; ++t) pack16_field(acc1[t][m1], lo[t], hi[t]);
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                const bf16x8 fa = A2[(mo * 8 + 2 * m1) * 64], fb = A2[(mo * 8 + 2 * m1 + 1) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lo[t], acc2[t][mo], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, hi[t], acc2[t][mo], 0, 0, 0);
            }
        }
        // ---- layer 2 (activation) and layer 3 (64 -> 1: output row 0 of a 32-row A tile, the rest zero) ----
        f32x16 acc3[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc3[t] = f32x16{0};
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            bf16x8 lo[T], hi[T];
#pragma unroll
            for (int t = 0; t < T; ++t) pack16_field(acc2[t][m2], lo[t], hi[t]);
            const bf16x8 fa = A3[(2 * m2) * 64], fb = A3[(2 * m2 + 1) * 64];
#pragma unroll
            for (int t = 0; t < T; ++t) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lo[t], acc3[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, hi[t], acc3[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t row = row0 + 32 * t + r;  // output feature 0 = accumulator register 0 of lane half 0
            if (h == 0 && row < n_rows && row0 + 32 * t < blk_end) Y[(int64_t)c * ldY + row] = acc3[t][0] + b3;
        }
    }
#ifdef MCMC_FWD_STAMP
    if (threadIdx.x == 0) {
        const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) & 65535u;
        g_fwd_stamp[2 * slot] = __builtin_amdgcn_s_memtime() - stamp_c0;
        g_fwd_stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
}

template <typename LAYOUT, typename KERNEL, typename... EXTRA>
static int launch_slabs(KERNEL kern, const void* xpack, int64_t n_rows, int F, const void* wpack, int pair_count, float* Y, int64_t ldY,
                        hipStream_t s, DropMask dm, bool with_masks) {
    const LAYOUT L(F);
    const size_t lds_bytes = (size_t)L.total;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_bytes, "forward_bf16")) return rc;
    // Row superblock: large enough that staging ~90 KB of weights is ~1 % of its MFMA time
    // (32 double tiles per wave), small enough that pairs x superblocks >> 256 CUs.
    const int64_t padded = rows_pad(n_rows);
    constexpr int step = 32 * kWaves * kTiles;
    int rows_per_block = 16384 / step * step;
    if (padded < rows_per_block) rows_per_block = (int)((padded + step - 1) / step * step);
    const int64_t row_blocks = (padded + rows_per_block - 1) / rows_per_block;
    if (row_blocks > 65535) return set_error(MCMC_EINVAL, "n_rows too large for one launch (%lld row blocks)", (long long)row_blocks);
    for (int done = 0; done < pair_count;) {  // gridDim.x slabs
        const int n = pair_count - done > 32768 ? 32768 : pair_count - done;
        const unsigned char* wp = (const unsigned char*)wpack + (int64_t)done * L.total;
        float* y = Y + (int64_t)done * ldY;
        if constexpr (sizeof...(EXTRA) == 0) {
            hipLaunchKernelGGL(kern, dim3(n, (unsigned)row_blocks), dim3(64 * kWaves), lds_bytes, s, (const unsigned short*)xpack, n_rows, F,
                               rows_per_block, wp, y, ldY);
        } else {
            DropMask d = dm;  // this slab's pairs: host masks and the pair ids advance with it
            if (with_masks && d.mode == 1) {
                d.m[0] += (int64_t)done * n_rows * H1;
                d.m[1] += (int64_t)done * n_rows * H2;
                d.m[2] += (int64_t)done * n_rows * H3;
            }
            d.pair_begin += done;
            hipLaunchKernelGGL(kern, dim3(n, (unsigned)row_blocks), dim3(64 * kWaves), lds_bytes, s, (const unsigned short*)xpack, n_rows, F,
                               rows_per_block, wp, y, ldY, d);
        }
        done += n;
    }
    return check_launch("forward_bf16");
}

}  // namespace

int launch_forward_bf16(const mcmc_ctx* ctx, bool fast, const void* xpack, int64_t n_rows, int F, const void* wpack, int pair_count,
                        const DropMask* dm, float* Y, int64_t ldY, hipStream_t s) {
    (void)ctx;
    const bool k16 = k0_pad(F, DT_BF16) == 16;
    const int mode = dm ? dm->mode : 0;
    const DropMask d = dm ? *dm : DropMask{};
    if (fast) {
        if (mode) return set_error(MCMC_EINVAL, "MCMC_DT_BF16_FAST has no dropout-mask stage: use MCMC_DT_BF16");
        if (k16) return launch_slabs<WPackBF16F>(forward_bf16_fast_kernel<1>, xpack, n_rows, F, wpack, pair_count, Y, ldY, s, d, false);
        return launch_slabs<WPackBF16F>(forward_bf16_fast_kernel<2>, xpack, n_rows, F, wpack, pair_count, Y, ldY, s, d, false);
    }
#define MCMC_GO(MODE)                                                                                                                   \
    return k16 ? launch_slabs<WPackBF16, decltype(&forward_bf16_exact_kernel<1, MODE>), DropMask>(forward_bf16_exact_kernel<1, MODE>, xpack, n_rows, F, \
                                                                                                 wpack, pair_count, Y, ldY, s, d, true)               \
               : launch_slabs<WPackBF16, decltype(&forward_bf16_exact_kernel<2, MODE>), DropMask>(forward_bf16_exact_kernel<2, MODE>, xpack, n_rows, F, \
                                                                                                 wpack, pair_count, Y, ldY, s, d, true)
    if (mode == 1) MCMC_GO(1);
    if (mode == 2) MCMC_GO(2);
    MCMC_GO(0);
#undef MCMC_GO
}

}  // namespace mcmc
