// Row f1, large batches (BASELINE config 5), opt-in bf16 variant: the train-mode forward and the backward pass of every
// (member, sample) pair on v_mfma_f32_32x32x16_bf16 - bf16 operands, fp32 accumulation, fp32 master weights and AdamW
// (mcmc_train_sample / mcmc_train_grad / mcmc_train_apply are shared with the fp32 path, train.hip).
//
// Restates the forward / loss.backward() pair of GPUBayesianEnsemble.train's inner loop
// (/root/reference/models/bayesian_ensemble.py:165-178) like train_tiles.hpp does, with bf16 GEMM operands: fp32 MFMA runs
// at the vector rate (1/16 of the bf16 rate), and at 10^5 .. 10^6 rows per step the fp32 kernels are bound by it.
//
// One workgroup (8 waves) = one pair x one row block.  The pair's effective weights (W = mu + eps * sigma, fp32 [out][in] from
// mcmc_train_sample) are converted once into bf16 row-major LDS images; the activations of a 32-row tile live in LDS as bf16
// [row][feature] tiles and are turned into the gradients w.r.t. their pre-activations in place, as in the fp32 kernels.
// Every GEMM of the step reads its operands from those images with one of two fragment loaders:
//   row fragment  (ds_read_b128):          lane (r, h) <- T[row0 + r][k0 + 8h .. 8h+7]        A = W rows, B = activation rows
//   transposed    (2 x ds_read_b64_tr_b16): lane (r, h) <- T[k0 + 8h .. 8h+7][col0 + r]        the same tiles read column-wise
//   forward      Z^T[out, row]  = W[out, k]       . H[row, k]        A row(W),   B row(H)
//   input grad   dH^T[in, row]  = W^T[in, out]    . dZ[row, out]     A tr(W),    B row(dZ)
//   weight grad  dW[out, in]   += dZ^T[out, row]  . H[row, in]       A tr(dZ),   B tr(H)
// so no tile is stored twice.  The weight gradients of a row block stay in accumulator registers across its tiles (96 per
// lane: wave w owns dW1[:, 32w .. 32w+31], one 32 x 32 tile of dW2 and one of dW0) and are written once, as the block's
// partial dW[pair * row_blocks + block] in the flat layout, for mcmc_train_grad to sum.  ELU, dropout masks (host or the
// 'MASK' Philox stream, dropmask.hpp) and the -0.0 marking of dropped elements are those of the fp32 kernels.
#include "common.hpp"
#include "dropmask.hpp"
#include "elu.hpp"

namespace mcmc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int NT = 512, NW = 8, TR = 32;
constexpr int KX = 32;  // width of the input tile: input columns, then the 1.0 that carries the layer-0 bias (F <= 24)
// row strides in bf16 elements: +8 keeps rows 16-byte aligned and spreads ds_read_b128 row reads over all banks
constexpr int SW0 = KX + 8, SW1 = H1 + 8, SW2 = H2 + 8;           // weight images  [256][SW0], [128][SW1], [64][SW2]
constexpr int SX = KX + 8, S1 = H1 + 8, S2 = H2 + 8, S3 = H3 + 8;  // activation tiles [TR][...]
constexpr int kSmallFloats = H2 + H3 + H3 + 4;                     // b1 | b2 | w3 | b3 (fp32)
constexpr size_t kLdsBytes = 2 * (size_t)(H1 * SW0 + H2 * SW1 + H3 * SW2 + TR * (SX + S1 + S2 + S3)) + 4 * (kSmallFloats + TR);

struct Lds {
    __bf16 *w0, *w1, *w2;      // weight images
    __bf16 *xs, *h1, *h2, *h3;  // tiles
    float *b1, *b2, *w3, *b3;  // fp32
    float* y;                  // [TR] outputs, then dLoss/dy
};

__device__ __forceinline__ Lds carve(unsigned char* base) {
    Lds l;
    l.w0 = reinterpret_cast<__bf16*>(base);
    l.w1 = l.w0 + H1 * SW0;
    l.w2 = l.w1 + H2 * SW1;
    l.xs = l.w2 + H3 * SW2;
    l.h1 = l.xs + TR * SX;
    l.h2 = l.h1 + TR * S1;
    l.h3 = l.h2 + TR * S2;
    l.b1 = reinterpret_cast<float*>(l.h3 + TR * S3);
    l.b2 = l.b1 + H2;
    l.w3 = l.b2 + H3;
    l.b3 = l.w3 + H3;
    l.y = l.b3 + 4;
    return l;
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }

// lane (r, h) <- T[row0 + r][k0 + 8h .. +7]
__device__ __forceinline__ bf16x8 row_frag(const __bf16* T, int stride, int row0, int k0) {
    const int lane = threadIdx.x & 63;
    return *reinterpret_cast<const bf16x8*>(T + (row0 + (lane & 31)) * stride + k0 + 8 * (lane >> 5));
}

// lane (r, h) <- T[k0 + 8h + j][col0 + r], j = 0..7: per 16-lane group two 4-row x 16-column blocks read column-major
// (ds_read_b64_tr_b16: lane 4q+p of a group supplies the address of block row q, columns 4p .. 4p+3, and receives column
// (lane & 15) of the four rows).  EXEC must be all ones: callers keep these reads out of lane-divergent code.
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* T, int stride, int k0, int col0) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16* a = T + (k0 + 8 * (g >> 1) + q) * stride + col0 + 16 * (g & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// derivative factor d = elu'(z) * mask * scale recovered from the stored activation (train.hip:dfactor)
__device__ __forceinline__ float dfactor(float h, float scale) {
    if (__float_as_uint(h) == 0x80000000u) return 0.0f;  // masked (or padding row)
    return h > 0.0f ? scale : h + scale;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// activation epilogue of one 32 x 32 forward unit: ELU, dropout, bf16, into hout[row][32 mo + feature]
template <int MODE>
__device__ __forceinline__ void store_activation(const f32x16& acc, __bf16* hout, int stride, int mo, const DropMask& dm, int layer, int pair,
                                                 int64_t rows, int64_t row) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // both branches evaluated, then a select: no exec-mask branch per element
        const float e = elu_neg(fminf(acc[i], 0.0f));
        o[i] = acc[i] > 0.0f ? acc[i] : e;
    }
    drop16<MODE>(o, dm, layer, mo, h, pair, (uint32_t)pair, rows, row);
    const bool live = row < rows;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // registers 4q .. 4q+3 hold features 8q + 4h + 0..3 (layout.hpp: acc_feature)
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (__bf16)(live ? o[4 * q + j] : -0.0f);
        *reinterpret_cast<bf16x4*>(hout + r * stride + 32 * mo + 8 * q + 4 * h) = v;
    }
}

// the pair's weights -> bf16 LDS images (once per workgroup)
__device__ __forceinline__ void stage_weights(const Lds& l, const float* __restrict__ w, int F) {
    const int t = threadIdx.x;
    for (int i = t; i < H1 * KX; i += NT) {
        const int o = i / KX, k = i % KX;
        const float v = k < F ? w[flat_w_off(F, 0) + (int64_t)o * F + k] : k == F ? w[flat_b_off(F, 0) + o] : 0.0f;
        l.w0[o * SW0 + k] = (__bf16)v;
    }
    for (int i = t; i < H2 * H1 / 4; i += NT) {
        const int o = i / (H1 / 4), k = 4 * (i % (H1 / 4));
        const float4 v = *reinterpret_cast<const float4*>(w + flat_w_off(F, 1) + (int64_t)o * H1 + k);
        *reinterpret_cast<bf16x4*>(l.w1 + o * SW1 + k) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
    for (int i = t; i < H3 * H2; i += NT) l.w2[(i / H2) * SW2 + i % H2] = (__bf16)w[flat_w_off(F, 2) + i];
    for (int i = t; i < H2; i += NT) l.b1[i] = w[flat_b_off(F, 1) + i];
    for (int i = t; i < H3; i += NT) {
        l.b2[i] = w[flat_b_off(F, 2) + i];
        l.w3[i] = w[flat_w_off(F, 3) + i];
    }
    if (t == 0) l.b3[0] = w[flat_b_off(F, 3)];
}

constexpr int XN = TR * KX / NT;  // input-tile elements per thread

// this thread's elements of the input tile that starts at row_base (column F carries the 1.0 of the layer-0 bias)
__device__ __forceinline__ void load_x(float (&xr)[XN], const float* __restrict__ xm, int64_t rows, int F, int64_t row_base) {
#pragma unroll
    for (int q = 0; q < XN; ++q) {
        const int i = threadIdx.x + q * NT, rr = i / KX, k = i % KX;
        const int64_t row = row_base + rr;
        float v = 0.0f;
        if (row < rows) v = k < F ? xm[row * F + k] : (k == F ? 1.0f : 0.0f);
        xr[q] = v;
    }
}
__device__ __forceinline__ void store_x(const Lds& l, const float (&xr)[XN]) {
#pragma unroll
    for (int q = 0; q < XN; ++q) {
        const int i = threadIdx.x + q * NT;
        l.xs[(i / KX) * SX + i % KX] = (__bf16)xr[q];
    }
}

// forward of one 32-row tile: fills xs (from xr), h1, h2, h3 (bf16) and y (fp32); xr is then refilled with the next tile's
// input rows (next_base >= 0), whose loads land under the layers instead of in front of the next tile
template <int MODE>
__device__ __forceinline__ void forward_tile(const Lds& l, float (&xr)[XN], const float* __restrict__ xm, int64_t rows, int F, int64_t row_base,
                                             int64_t next_base, const DropMask& dm, int pair) {
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 31, h = lane >> 5;
    store_x(l, xr);
    __syncthreads();
    if (next_base >= 0) load_x(xr, xm, rows, F, next_base);
    const int64_t row = row_base + r;
    {  // layer 0: 8 units, one per wave; the bias rides on input column F
        f32x16 acc = {0};
        const int ks0 = (F + 1 + 15) / 16;
        for (int s = 0; s < ks0; ++s) acc = mfma16(row_frag(l.w0, SW0, 32 * wave, 16 * s), row_frag(l.xs, SX, 0, 16 * s), acc);
        store_activation<MODE>(acc, l.h1, S1, wave, dm, 0, pair, rows, row);
    }
    __syncthreads();
    if (wave < H2 / 32) {  // layer 1: 4 units
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = l.b1[32 * wave + acc_feature(i, h)];
#pragma unroll 4
        for (int s = 0; s < H1 / 16; ++s) acc = mfma16(row_frag(l.w1, SW1, 32 * wave, 16 * s), row_frag(l.h1, S1, 0, 16 * s), acc);
        store_activation<MODE>(acc, l.h2, S2, wave, dm, 1, pair, rows, row);
    }
    __syncthreads();
    if (wave < H3 / 32) {  // layer 2: 2 units
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = l.b2[32 * wave + acc_feature(i, h)];
#pragma unroll 4
        for (int s = 0; s < H2 / 16; ++s) acc = mfma16(row_frag(l.w2, SW2, 32 * wave, 16 * s), row_frag(l.h2, S2, 0, 16 * s), acc);
        store_activation<MODE>(acc, l.h3, S3, wave, dm, 2, pair, rows, row);
    }
    __syncthreads();
    if (t < 8 * TR) {  // layer 3: 64 -> 1, eight threads per row
        const int rr = t >> 3, part = t & 7;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = fmaf((float)l.h3[rr * S3 + part * 8 + k], l.w3[part * 8 + k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (part == 0) l.y[rr] = a + l.b3[0];
    }
    __syncthreads();
}

__device__ __forceinline__ void block_rows(int64_t rows, int64_t& begin, int64_t& end) {
    const int64_t per = (((rows + gridDim.y - 1) / gridDim.y) + TR - 1) / TR * TR;
    begin = (int64_t)blockIdx.y * per;
    end = begin + per < rows ? begin + per : rows;
}

// h1 | h2 | h3 of a tile are contiguous in LDS: the forward kernel can leave them in `act` (29.5 KB per pair and 32 rows) for the
// backward kernel, which then neither recomputes the forward pass nor draws the masks again (dropped elements are -0.0)
constexpr int kActElems = TR * (S1 + S2 + S3);
static_assert((kActElems * 2) % 16 == 0, "activation tiles are copied as 16-byte pieces");
constexpr int AN = (kActElems * 2 / 16 + NT - 1) / NT;  // 16-byte pieces per thread

template <int MODE>
__global__ __launch_bounds__(NT, 2) void train_forward_bf16_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F, int S,
                                                                  const float* __restrict__ W, DropMask dm, float* __restrict__ Y,
                                                                  uint4* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S;
    int64_t begin, end;
    block_rows(rows, begin, end);
    if (begin >= end) return;
    const float* xm = xb + (int64_t)m * x_stride;
    float xr[XN];
    load_x(xr, xm, rows, F, begin);
    stage_weights(l, W + (int64_t)pair * flat_param_count(F), F);
    __syncthreads();
    const int64_t tiles = (rows + TR - 1) / TR;
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
        forward_tile<MODE>(l, xr, xm, rows, F, row_base, row_base + TR < end ? row_base + TR : -1, dm, pair);
        if (threadIdx.x < TR && row_base + threadIdx.x < rows) Y[(int64_t)pair * rows + row_base + threadIdx.x] = l.y[threadIdx.x];
        if (act) {
            uint4* dst = act + ((int64_t)pair * tiles + row_base / TR) * (kActElems * 2 / 16);
            const uint4* src = reinterpret_cast<const uint4*>(l.h1);
            for (int i = threadIdx.x; i < kActElems * 2 / 16; i += NT) dst[i] = src[i];
        }
        __syncthreads();
    }
}

// dH^T[in tile mi, rows] = W^T . dZ, times d(h_in), written over h_in in place: one 32 x 32 unit
__device__ __forceinline__ void input_grad_unit(const __bf16* Wimg, int sw, int OUT, const __bf16* dz, int sdz, __bf16* hin, int sin, int mi,
                                                float scale) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int s = 0; s < OUT / 16; ++s) acc = mfma16(tr_frag(Wimg, sw, 16 * s, 32 * mi), row_frag(dz, sdz, 0, 16 * s), acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4* p = reinterpret_cast<bf16x4*>(hin + r * sin + 32 * mi + 8 * q + 4 * h);
        const bf16x4 hv = *p;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)(acc[4 * q + j] * dfactor((float)hv[j], scale));
        *p = o;
    }
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void train_backward_bf16_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F, int S,
                                                                   const float* __restrict__ W, DropMask dm, const float* __restrict__ delta,
                                                                   float* __restrict__ dW, const uint4* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S, t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 31, h = lane >> 5;
    const int64_t P = flat_param_count(F);
    float* g = dW + ((int64_t)pair * gridDim.y + blockIdx.y) * P;
    int64_t begin, end;
    block_rows(rows, begin, end);
    if (begin >= end) {  // more blocks than tiles: this partial is zero
        for (int64_t i = t; i < P; i += NT) g[i] = 0.0f;
        return;
    }
    const float* xm = xb + (int64_t)m * x_stride;
    const int64_t tiles = (rows + TR - 1) / TR;
    float xr[XN];
    uint4 ar[AN];  // the next tile's saved activations, in flight under the current tile's arithmetic
    auto load_act = [&](int64_t row_base) {
        const uint4* src = act + ((int64_t)pair * tiles + row_base / TR) * (kActElems * 2 / 16);
#pragma unroll
        for (int q = 0; q < AN; ++q)
            if (t + q * NT < kActElems * 2 / 16) ar[q] = src[t + q * NT];
    };
    load_x(xr, xm, rows, F, begin);
    if (act) load_act(begin);
    stage_weights(l, W + (int64_t)pair * P, F);
    __syncthreads();
    const float sc = MODE != 0 ? dm.scale : 1.0f;
    // the block's weight gradients, in registers across its tiles
    f32x16 gw1[H2 / 32], gw2 = {0}, gw0 = {0};
#pragma unroll
    for (int mo = 0; mo < H2 / 32; ++mo) gw1[mo] = f32x16{0};
    float gcol = 0.0f;  // thread t: db1[t] (t < 128), db2[t - 128] (t < 192), dw3[t - 192] (t < 256), db3 (t == 256)
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
        const int64_t next_base = row_base + TR < end ? row_base + TR : -1;
        if (act) {  // the forward kernel saved this tile's h1 | h2 | h3: no recompute, no masks
            store_x(l, xr);
            uint4* dst = reinterpret_cast<uint4*>(l.h1);
#pragma unroll
            for (int q = 0; q < AN; ++q)
                if (t + q * NT < kActElems * 2 / 16) dst[t + q * NT] = ar[q];
            __syncthreads();
            if (next_base >= 0) {
                load_x(xr, xm, rows, F, next_base);
                load_act(next_base);
            }
        } else {
            forward_tile<MODE>(l, xr, xm, rows, F, row_base, next_base, dm, pair);
        }
        // ---- layer 3: dy; dw3, db3; dz2 = dy * w3 * d(h3) over h3
        if (t < TR) l.y[t] = row_base + t < rows ? delta[(int64_t)m * rows + row_base + t] : 0.0f;
        __syncthreads();
        if (t >= H2 + H3 && t < H2 + H3 + H3) {
            const int k = t - (H2 + H3);
            float a = 0.0f;
#pragma unroll 8
            for (int rr = 0; rr < TR; ++rr) a = fmaf(l.y[rr], (float)l.h3[rr * S3 + k], a);
            gcol += a;
        } else if (t == H2 + H3 + H3) {
            float a = 0.0f;
            for (int rr = 0; rr < TR; ++rr) a += l.y[rr];
            gcol += a;
        }
        __syncthreads();
        for (int i = t; i < TR * H3; i += NT) {
            const int rr = i / H3, k = i % H3;
            const float hv = (float)l.h3[rr * S3 + k];
            l.h3[rr * S3 + k] = (__bf16)(l.y[rr] * l.w3[k] * dfactor(hv, sc));
        }
        __syncthreads();
        // ---- layer 2: dW2 += dz2^T h2 (wave w: out tile w >> 2, in tile w & 3); db2; then dz1 over h2
#pragma unroll
        for (int ks = 0; ks < TR / 16; ++ks)
            gw2 = mfma16(tr_frag(l.h3, S3, 16 * ks, 32 * (wave >> 2)), tr_frag(l.h2, S2, 16 * ks, 32 * (wave & 3)), gw2);
        if (t >= H2 && t < H2 + H3) {
            float a = 0.0f;
#pragma unroll 8
            for (int rr = 0; rr < TR; ++rr) a += (float)l.h3[rr * S3 + t - H2];
            gcol += a;
        }
        __syncthreads();
        if (wave < H2 / 32) input_grad_unit(l.w2, SW2, H3, l.h3, S3, l.h2, S2, wave, sc);
        __syncthreads();
        // ---- layer 1: dW1[:, in tile w] += dz1^T h1; db1; then dz0 over h1
#pragma unroll
        for (int ks = 0; ks < TR / 16; ++ks) {
            const bf16x8 b = tr_frag(l.h1, S1, 16 * ks, 32 * wave);
#pragma unroll
            for (int mo = 0; mo < H2 / 32; ++mo) gw1[mo] = mfma16(tr_frag(l.h2, S2, 16 * ks, 32 * mo), b, gw1[mo]);
        }
        if (t < H2) {
            float a = 0.0f;
#pragma unroll 8
            for (int rr = 0; rr < TR; ++rr) a += (float)l.h2[rr * S2 + t];
            gcol += a;
        }
        __syncthreads();
        input_grad_unit(l.w1, SW1, H2, l.h2, S2, l.h1, S1, wave, sc);
        __syncthreads();
        // ---- layer 0: dW0 | db0 (column F of the input tile is 1) += dz0^T x  (wave w: out tile w)
#pragma unroll
        for (int ks = 0; ks < TR / 16; ++ks) gw0 = mfma16(tr_frag(l.h1, S1, 16 * ks, 32 * wave), tr_frag(l.xs, SX, 16 * ks, 0), gw0);
        __syncthreads();
    }
    // ---- the block's partial, flat layout: register i of lane (r, h) holds [out = 32 tile + acc_feature(i, h)][in = 32 tile + r]
#pragma unroll
    for (int mo = 0; mo < H2 / 32; ++mo)
#pragma unroll
        for (int i = 0; i < 16; ++i) g[flat_w_off(F, 1) + (int64_t)(32 * mo + acc_feature(i, h)) * H1 + 32 * wave + r] = gw1[mo][i];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[flat_w_off(F, 2) + (int64_t)(32 * (wave >> 2) + acc_feature(i, h)) * H2 + 32 * (wave & 3) + r] = gw2[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = 32 * wave + acc_feature(i, h);
        if (r < F) g[flat_w_off(F, 0) + (int64_t)o * F + r] = gw0[i];
        else if (r == F) g[flat_b_off(F, 0) + o] = gw0[i];
    }
    if (t < H2) g[flat_b_off(F, 1) + t] = gcol;
    else if (t < H2 + H3) g[flat_b_off(F, 2) + t - H2] = gcol;
    else if (t < H2 + H3 + H3) g[flat_w_off(F, 3) + t - (H2 + H3)] = gcol;
    else if (t == H2 + H3 + H3) g[flat_b_off(F, 3)] = gcol;
}

DropMask make_mask(int mode, const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t seed, float p_drop) {
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

#define MCMC_TB_PICK(KERNEL, mode) ((mode) == 1 ? KERNEL<1> : (mode) == 2 ? KERNEL<2> : KERNEL<0>)

}  // namespace

int launch_train_forward_bf16(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, int mask_mode,
                              const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, int row_blocks,
                              float* Y, void* act, hipStream_t s) {
    auto kern = MCMC_TB_PICK(train_forward_bf16_kernel, mask_mode);
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), kLdsBytes, "train_forward_bf16")) return rc;
    hipLaunchKernelGGL(kern, dim3(M * S, row_blocks), dim3(NT), kLdsBytes, s, xb, x_stride, rows, F, S, W,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), Y, reinterpret_cast<uint4*>(act));
    return check_launch("train_forward_bf16");
}

int launch_train_backward_bf16(const float* xb, int64_t x_stride, int64_t rows, int F, int M, int S, const float* W, int mask_mode,
                               const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, uint64_t mask_seed, float p_drop, const float* delta,
                               int row_blocks, float* dW, const void* act, hipStream_t s) {
    auto kern = MCMC_TB_PICK(train_backward_bf16_kernel, mask_mode);
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), kLdsBytes, "train_backward_bf16")) return rc;
    hipLaunchKernelGGL(kern, dim3(M * S, row_blocks), dim3(NT), kLdsBytes, s, xb, x_stride, rows, F, S, W,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), delta, dW, reinterpret_cast<const uint4*>(act));
    return check_launch("train_backward_bf16");
}

// bytes per pair of the bf16 saved-activation buffer for a batch of `rows` rows
int64_t train_act_bytes_bf16(int64_t rows) { return (rows + TR - 1) / TR * (int64_t)kActElems * 2; }

}  // namespace mcmc
