// Row f1, large batches (BASELINE config 5), opt-in bf16 variant: the train-mode forward and the backward pass of every
// (member, sample) pair on v_mfma_f32_32x32x16_bf16 - bf16 operands, fp32 accumulation, fp32 master weights and AdamW
// (mcmc_train_sample / mcmc_train_grad / mcmc_train_apply are shared with the fp32 path, train.hip).
//
// Restates the forward / loss.backward() pair of GPUBayesianEnsemble.train's inner loop
// (/root/reference/models/bayesian_ensemble.py:165-178) like train_tiles.hpp does, with bf16 GEMM operands: fp32 MFMA runs
// at the vector rate (1/16 of the bf16 rate), and at 10^5 .. 10^6 rows per step the fp32 kernels are bound by it.
//
// One workgroup (16 waves, four per SIMD) = one pair x one row block, 64 rows at a time.  The pair's effective weights
// (W = mu + eps * sigma, fp32 [out][in] from mcmc_train_sample) of layers 1 and 2 are converted once into bf16 row-major LDS
// images (layer 0's 256 x (F+1) sit in the registers of the wave that multiplies by them); the activations of a 64-row tile live
// in LDS as bf16 [row][feature] tiles and are turned into the gradients w.r.t. their pre-activations in place, as in the fp32
// kernels.  Every GEMM of the step reads its operands from those images with one of two fragment loaders:
//   row fragment  (ds_read_b128):          lane (r, h) <- T[row0 + r][k0 + 8h .. 8h+7]        A = W rows, B = activation rows
//   transposed    (2 x ds_read_b64_tr_b16): lane (r, h) <- T[k0 + 8h .. 8h+7][col0 + r]        the same tiles read column-wise
//   forward      Z^T[out, row]  = W[out, k]       . H[row, k]        A row(W),   B row(H)
//   input grad   dH^T[in, row]  = W^T[in, out]    . dZ[row, out]     A tr(W),    B row(dZ)
//   weight grad  dW[out, in]   += dZ^T[out, row]  . H[row, in]       A tr(dZ),   B tr(H)
// so no tile is stored twice.  The phases between two workgroup barriers are sized so that the 64 waves-per-CU-worth of MFMAs
// of the wide layers run together: the weight gradient of a layer and the input gradient that follows it only READ the tiles, so
// they share a phase, and the in-place epilogue that overwrites the activations with the next gradient is the phase after.
// The weight gradients of a row block stay in accumulator registers across its tiles (48 per lane: wave w owns two 32 x 32
// tiles of dW1 and one of dW2 (w < 8) or dW0 (w >= 8)) and are written once, as the block's partial
// dW[pair * row_blocks + block] in the flat layout, for mcmc_train_grad to sum; bias and output-layer gradients are per-thread
// partial sums reduced across the workgroup at the end of the block.  Dropout masks (host or the 'MASK' Philox stream,
// dropmask.hpp) and the -0.0 marking of dropped elements are those of the fp32 kernels.
// The forward kernel can leave h1 | h2 | h3 of every tile in global memory (the LDS image byte for byte); the backward kernel
// then reloads them by LDS-DMA, each part while its LDS region is dead (phases: see the kernel), instead of recomputing the
// tile's forward pass (second build of the kernel; identical results).
// History, phase timings and what bounds the kernels now: profiles/EXPERIMENTS.md section 4b.
#include "common.hpp"
#include "dropmask.hpp"

namespace mcmc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int NT = 1024, NW = 16, TR = 64;
constexpr int KX = 32;  // width of the input tile: input columns, then the 1.0 that carries the layer-0 bias (F <= 24)
// row strides in bf16 elements: +8 keeps rows 16-byte aligned and spreads ds_read_b128 row reads over all banks
constexpr int SW1 = H1 + 8, SW2 = H2 + 8;                          // weight images  [128][SW1], [64][SW2]
constexpr int SX = KX + 8, S1 = H1 + 8, S2 = H2 + 8, S3 = H3 + 8;  // activation tiles [TR][...]
constexpr int kSmallFloats = H2 + H3 + H3 + 4;                     // b1 | b2 | w3 | b3 (fp32)
constexpr size_t kLdsBytes = 2 * (size_t)(H2 * SW1 + H3 * SW2 + TR * (SX + S1 + S2 + S3)) + 4 * (kSmallFloats + TR);
static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU");

struct Lds {
    __bf16 *w1, *w2;            // weight images
    __bf16 *xs, *h1, *h2, *h3;  // tiles
    float *b1, *b2, *w3, *b3;   // fp32
    float* y;                   // [TR] dLoss/dy of the tile
};

__device__ __forceinline__ Lds carve(unsigned char* base) {
    Lds l;
    l.w1 = reinterpret_cast<__bf16*>(base);
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
__device__ __forceinline__ bf16x8 row_frag(int lane, const __bf16* T, int stride, int row0, int k0) {
    return *reinterpret_cast<const bf16x8*>(T + (row0 + (lane & 31)) * stride + k0 + 8 * (lane >> 5));
}

// Transposed fragment of the 16 tile rows k0 .. k0+15 at columns col0 .. col0+31: lane (r, h) <- T[row(h, j)][col0 + r], j = 0..7
// (two ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of one row's columns 4p .. 4p+3, and receives
// column (lane & 15) of the group's four rows).  EXEC must be all ones: callers keep these reads out of lane-divergent code.
//   SPREAD (default): row(h, j) = k0 + 4 (j & 3) + 2h + (j >> 2).  The four rows one 32-lane pass touches are then 4 apart: with
//     row strides of 4 mod 16 dwords (all of the tiles here) they fall into four different quarters of the 64 banks and the read
//     is conflict-free; with consecutive rows (row = k0 + 8h + j) it is 2- to 4-way conflicted, and the backward kernel's LDS
//     array was busy 71 % of the time, 57 % of it conflict cycles (PMC).  Which of the 16 rows lands in which k slot does not
//     matter when both operands of the MFMA come from this loader (the weight gradients: k = batch row); for the input
//     gradients the weight images are staged with their rows in this order (stage_weights<true>).
//   !SPREAD: row(h, j) = k0 + 8h + j, for a weight image in natural row order (the recomputing build, whose forward pass reads
//     the same image row-wise).
template <bool SPREAD = true>
__device__ __forceinline__ bf16x8 tr_frag(int lane, const __bf16* T, int stride, int k0, int col0) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16* a = T + (SPREAD ? k0 + 4 * q + 2 * (g >> 1) : k0 + 8 * (g >> 1) + q) * stride + col0 + 16 * (g & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + (SPREAD ? 1 : 4) * stride));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// where stage_weights<true> puts row o of a weight image: k slot 8h + 4e + q of its 16-row block -> row 4q + 2h + e
__device__ __forceinline__ int spread_row(int o) { return (o & ~15) + 4 * (o & 3) + 2 * ((o >> 3) & 1) + ((o >> 2) & 1); }

// The lane id, opaque to the optimiser: addresses derived from it are computed where they are used instead of being hoisted out
// of the tile loop, where two dozen of them would stay live beside the accumulators (128 registers per lane at 16 waves).
__device__ __forceinline__ int fresh_lane() {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    return lane;
}

// derivative factor d = elu'(z) * mask * scale recovered from the stored activation (train.hip:dfactor)
__device__ __forceinline__ float dfactor(float h, float scale) {
    if (__float_as_uint(h) == 0x80000000u) return 0.0f;  // masked (or padding row)
    return h > 0.0f ? scale : h + scale;
}

// activation epilogue of one 32 x 32 forward unit (feature tile mo, row tile rt): ELU, dropout, bf16, into hout[row][32 mo + feature]
template <int MODE>
__device__ __forceinline__ void store_activation(const f32x16& acc, __bf16* hout, int stride, int mo, int rt, const DropMask& dm, int layer,
                                                 int pair, int64_t rows, int64_t row_base) {
    const int lane = fresh_lane(), r = lane & 31, h = lane >> 5;
    const int64_t row = row_base + 32 * rt + r;
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // exp(z) - 1 as it stands: its absolute error (6e-8) is far below the bf16 rounding of the stored activation, so the
        // small-|z| polynomial of the fp32 kernels (elu.hpp) buys nothing here; both branches evaluated, then a select
        // (no exec-mask branch per element; exp of a large positive z is inf and not selected)
        const float e = __expf(acc[i]) - 1.0f;
        o[i] = acc[i] > 0.0f ? acc[i] : e;
    }
    drop16<MODE>(o, dm, layer, mo, h, pair, (uint32_t)pair, rows, row);
    if (row_base + TR > rows) {  // the batch's last, ragged tile (workgroup-uniform): rows past the end read as dropped
        const bool live = row < rows;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = live ? o[i] : -0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // registers 4q .. 4q+3 hold features 8q + 4h + 0..3 (layout.hpp: acc_feature)
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (__bf16)o[4 * q + j];
        *reinterpret_cast<bf16x4*>(hout + (32 * rt + r) * stride + 32 * mo + 8 * q + 4 * h) = v;
    }
}

// the pair's layer-1 / layer-2 weights -> bf16 LDS images, biases and the output layer -> fp32 (once per workgroup)
template <bool SPREAD>
__device__ __forceinline__ void stage_weights(const Lds& l, const float* __restrict__ w, int F) {
    const int t = threadIdx.x;
    for (int i = t; i < H2 * H1 / 4; i += NT) {
        const int o = i / (H1 / 4), k = 4 * (i % (H1 / 4));
        const float4 v = *reinterpret_cast<const float4*>(w + flat_w_off(F, 1) + (int64_t)o * H1 + k);
        *reinterpret_cast<bf16x4*>(l.w1 + (SPREAD ? spread_row(o) : o) * SW1 + k) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
    for (int i = t; i < H3 * H2 / 4; i += NT) {
        const int o = i / (H2 / 4), k = 4 * (i % (H2 / 4));
        const float4 v = *reinterpret_cast<const float4*>(w + flat_w_off(F, 2) + (int64_t)o * H2 + k);
        *reinterpret_cast<bf16x4*>(l.w2 + (SPREAD ? spread_row(o) : o) * SW2 + k) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
    if (t < H2) l.b1[t] = w[flat_b_off(F, 1) + t];
    else if (t < H2 + H3) l.b2[t - H2] = w[flat_b_off(F, 2) + t - H2];
    else if (t < H2 + 2 * H3) l.w3[t - H2 - H3] = w[flat_w_off(F, 3) + t - H2 - H3];
    else if (t == H2 + 2 * H3) l.b3[0] = w[flat_b_off(F, 3)];
}

// layer 0 as the A fragments of one output tile: lane (r, h) <- [W0 | b0 | 0][32 ot + r][16 s + 8h .. +7]
struct W0Frag {
    bf16x8 a[2];
};
__device__ __forceinline__ W0Frag load_w0(const float* __restrict__ w, int F, int ot) {
    const int lane = threadIdx.x & 63, o = 32 * ot + (lane & 31), h = lane >> 5;
    W0Frag f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            f.a[s][j] = (__bf16)(k < F ? w[flat_w_off(F, 0) + (int64_t)o * F + k] : k == F ? w[flat_b_off(F, 0) + o] : 0.0f);
        }
    return f;
}

// Workgroup barrier that leaves VMEM in flight: __syncthreads() drains vmcnt while an LDS-DMA load is outstanding
// (cdna_hip_programming.md, 'Pipelining across barriers'); LDS traffic is retired by the lgkmcnt wait.
__device__ __forceinline__ void barrier_lds_only() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <bool RAW>
__device__ __forceinline__ void tile_barrier() {
    if (RAW) barrier_lds_only();
    else __syncthreads();
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

// forward of one 64-row tile: fills xs (from xr), h1, h2, h3 (bf16); writes the outputs to Y (when given); xr is then refilled
// with the next tile's input rows (next_base >= 0), whose loads land under the layers instead of in front of the next tile.
// Four barriers; the tiles are complete after the third, and callers must put a barrier between this call and the next write
// of xs / h1 (the next call's own first barrier does that for a loop of calls).
template <int MODE>
__device__ __forceinline__ void forward_tile(const Lds& l, const W0Frag& w0, float (&xr)[XN], const float* __restrict__ xm, int64_t rows, int F,
                                             int64_t row_base, int64_t next_base, const DropMask& dm, int pair, float* __restrict__ Ypair,
                                             uint64_t* stamps = nullptr) {
#ifdef MCMC_TB_STAMP  // diagnostic build: s_memtime at the phase boundaries of the tile the caller picked
#define FT_STAMP(i) \
    if (stamps) stamps[i] = __builtin_amdgcn_s_memtime()
#else
#define FT_STAMP(i)
#endif
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), h = lane >> 5;
    FT_STAMP(0);
    store_x(l, xr);
    barrier_lds_only();
    FT_STAMP(1);
    if (next_base >= 0) load_x(xr, xm, rows, F, next_base);
    {  // layer 0: 8 feature tiles x 2 row tiles, one unit per wave; the bias rides on input column F
        const int mo = wave & 7, rt = wave >> 3, ln = fresh_lane();
        f32x16 acc = {0};
        acc = mfma16(w0.a[0], row_frag(ln, l.xs, SX, 32 * rt, 0), acc);
        if (F + 1 > 16) acc = mfma16(w0.a[1], row_frag(ln, l.xs, SX, 32 * rt, 16), acc);
        store_activation<MODE>(acc, l.h1, S1, mo, rt, dm, 0, pair, rows, row_base);
    }
    barrier_lds_only();
    FT_STAMP(2);
    if (wave < 2 * (H2 / 32)) {  // layer 1: 4 x 2 units
        const int mo = wave & 3, rt = wave >> 2, ln = fresh_lane();
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = l.b1[32 * mo + acc_feature(i, h)];
#pragma unroll 4
        for (int s = 0; s < H1 / 16; ++s) acc = mfma16(row_frag(ln, l.w1, SW1, 32 * mo, 16 * s), row_frag(ln, l.h1, S1, 32 * rt, 16 * s), acc);
        store_activation<MODE>(acc, l.h2, S2, mo, rt, dm, 1, pair, rows, row_base);
    }
    barrier_lds_only();
    FT_STAMP(3);
    if (wave < 2 * (H3 / 32)) {  // layer 2: 2 x 2 units
        const int mo = wave & 1, rt = wave >> 1, ln = fresh_lane();
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = l.b2[32 * mo + acc_feature(i, h)];
#pragma unroll 4
        for (int s = 0; s < H2 / 16; ++s) acc = mfma16(row_frag(ln, l.w2, SW2, 32 * mo, 16 * s), row_frag(ln, l.h2, S2, 32 * rt, 16 * s), acc);
        store_activation<MODE>(acc, l.h3, S3, mo, rt, dm, 2, pair, rows, row_base);
    }
    barrier_lds_only();
    FT_STAMP(4);
    if (Ypair && t < 8 * TR) {  // layer 3: 64 -> 1, eight threads per row
        const int rr = t >> 3, part = t & 7;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = fmaf((float)l.h3[rr * S3 + part * 8 + k], l.w3[part * 8 + k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (part == 0 && row_base + rr < rows) Ypair[row_base + rr] = a + l.b3[0];
    }
    FT_STAMP(5);
}

__device__ __forceinline__ void block_rows(int64_t rows, int64_t& begin, int64_t& end) {
    const int64_t per = (((rows + gridDim.y - 1) / gridDim.y) + TR - 1) / TR * TR;
    begin = (int64_t)blockIdx.y * per;
    end = begin + per < rows ? begin + per : rows;
}

// h1 | h2 | h3 of a tile are contiguous in LDS: the forward kernel can leave them in `act` (59 KB per pair and 64 rows) for the
// backward kernel, which then neither recomputes the forward pass nor draws the masks again (dropped elements are -0.0)
constexpr int kActElems = TR * (S1 + S2 + S3);
static_assert((kActElems * 2) % 16 == 0, "activation tiles are copied as 16-byte pieces");
constexpr int kActPieces = kActElems * 2 / 16;
// in 1-KiB pieces (one LDS-DMA wave-instruction each): h1 is the first 33, h2 | h3 the other 26
static_assert(TR * S1 * 2 % 1024 == 0 && kActElems * 2 % 1024 == 0, "the saved tile splits into whole 1-KiB pieces at h1 | h2");
constexpr int kKiBH1 = TR * S1 * 2 / 1024, kKiBAll = kActElems * 2 / 1024;

template <int MODE>
__global__ __launch_bounds__(NT) void train_forward_bf16_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F, int S,
                                                               const float* __restrict__ W, DropMask dm, float* __restrict__ Y,
                                                               uint4* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S;
    int64_t begin, end;
    block_rows(rows, begin, end);
    if (begin >= end) return;
    const float* xm = xb + (int64_t)m * x_stride;
    const float* w = W + (int64_t)pair * flat_param_count(F);
    float xr[XN];
    load_x(xr, xm, rows, F, begin);
    const W0Frag w0 = load_w0(w, F, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 7);
    stage_weights<false>(l, w, F);
    __syncthreads();
    const int64_t tiles = (rows + TR - 1) / TR;
#ifdef MCMC_TB_STAMP
    uint64_t stamps[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
#ifdef MCMC_TB_STAMP
        uint64_t* st = blockIdx.x == 0 && blockIdx.y == 0 && row_base == begin + 2 * TR ? stamps : nullptr;
#else
        uint64_t* st = nullptr;
#endif
        forward_tile<MODE>(l, w0, xr, xm, rows, F, row_base, row_base + TR < end ? row_base + TR : -1, dm, pair, Y + (int64_t)pair * rows, st);
        if (act) {  // the tiles are complete; the next tile's first barrier comes before anything overwrites them
            uint4* dst = act + ((int64_t)pair * tiles + row_base / TR) * kActPieces;
            const uint4* src = reinterpret_cast<const uint4*>(l.h1);
            for (int i = threadIdx.x; i < kActPieces; i += NT) dst[i] = src[i];
        }
#ifdef MCMC_TB_STAMP
        if (st) st[6] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef MCMC_TB_STAMP  // the stamps replace the first outputs of pair 0 (diagnostic build only)
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        for (int i = 0; i < 7; ++i) reinterpret_cast<uint64_t*>(Y)[i] = stamps[i];
#endif
}

// dH^T[in tile mi, row tile rt] = W^T . dZ: the MFMA chain (reads only); SPREAD = the image's rows are in spread_row order
template <bool SPREAD>
__device__ __forceinline__ f32x16 input_grad_mfma(const __bf16* Wimg, int sw, int OUT, const __bf16* dz, int sdz, int mi, int rt) {
    const int ln = fresh_lane();
    f32x16 acc = {0};
#ifndef TB_IG_UNROLL
#define TB_IG_UNROLL 2
#endif
#pragma unroll TB_IG_UNROLL
    for (int s = 0; s < OUT / 16; ++s) acc = mfma16(tr_frag<SPREAD>(ln, Wimg, sw, 16 * s, 32 * mi), row_frag(ln, dz, sdz, 32 * rt, 16 * s), acc);
    return acc;
}
// ... times d(h_in), written over h_in in place
__device__ __forceinline__ void input_grad_store(const f32x16& acc, __bf16* hin, int sin, int mi, int rt, float scale) {
    const int lane = fresh_lane(), r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4* p = reinterpret_cast<bf16x4*>(hin + (32 * rt + r) * sin + 32 * mi + 8 * q + 4 * h);
        const bf16x4 hv = *p;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)(acc[4 * q + j] * dfactor((float)hv[j], scale));
        *p = o;
    }
}

// SAVED: the forward kernel left the tiles in `act`; otherwise each tile's forward pass is recomputed here (two builds, so that
// neither carries the other's registers: 128 per lane at four waves per SIMD)
typedef __attribute__((address_space(3))) char lds_char;
// One LDS-DMA load (global_load_lds_dwordx4: lane i's 16 bytes land at the wave-uniform LDS byte address lds_dst + 16 i), as an
// asm statement so that hipcc does not count it: with the builtin every LDS read that follows waits vmcnt(0) for it (the
// compiler cannot tell that the read does not alias the destination), which would expose the load in the very next phase.
// Completion is counted by hand: s_waitcnt vmcnt(0) in the issuing wave, then a barrier, then the reads
// (cdna_hip_programming.md, inline-asm rules: M0 is written in the statement that uses it and restored).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// 1-KiB pieces [p0, p1) of a saved tile, global -> LDS without registers (the saved image is the LDS image byte for byte),
// pieces dealt round-robin to the waves
__device__ __forceinline__ void dma_pieces(const uint4* src_tile, __bf16* lds_tile, int wave, int p0, int p1) {
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)(lds_char*)reinterpret_cast<char*>(lds_tile);
#ifdef TB_SKIP_ACT  // timing experiment (profiles/EXPERIMENTS.md): the kernel without its saved-activation traffic
    return;
#endif
    for (int p = p0 + wave; p < p1; p += NW) glds16(src_tile + p * 64 + lane, __builtin_amdgcn_readfirstlane(base + p * 1024));
}

// Diagnostic build (-DMCMC_TB_STAMP, tools/check_train_bf16.py --stamps): workgroup (0, 0) records s_memtime at the phase
// boundaries of its third tile and leaves the stamps in the first floats of its dW partial (which is then not a gradient).
#ifdef MCMC_TB_STAMP
#define TB_STAMP(i) \
    if (stamp_tile) stamps[i] = __builtin_amdgcn_s_memtime()
#else
#define TB_STAMP(i)
#endif

template <int MODE, bool SAVED>
__global__ __launch_bounds__(NT) void train_backward_bf16_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F, int S,
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
    const float* w = W + (int64_t)pair * P;
    const float* dm_row = delta + (int64_t)m * rows;
    const int64_t tiles = (rows + TR - 1) / TR;
    float xr[XN];
    float dyr = 0.0f;  // t < TR: dLoss/dy of this thread's row of the next tile
    auto load_dy = [&](int64_t row_base) {
        if (t < TR) dyr = row_base + t < rows ? dm_row[row_base + t] : 0.0f;
    };
    load_x(xr, xm, rows, F, begin);
    load_dy(begin);
    W0Frag w0 = {};
    if (!SAVED) w0 = load_w0(w, F, wave & 7);
    stage_weights<SAVED>(l, w, F);
    if (SAVED) dma_pieces(act + ((int64_t)pair * tiles + begin / TR) * kActPieces, l.h1, wave, kKiBH1, kKiBAll);
    __syncthreads();
    const float sc = MODE != 0 ? dm.scale : 1.0f;
    // the block's weight gradients, in registers across its tiles:
    //   gw1[j]: dW1[out tile 2 (wave >> 3) + j][in tile wave & 7];  gwx: dW2[out tile wave >> 2][in tile wave & 3] (wave < 8),
    //   dW0 | db0 [out tile wave - 8] (wave >= 8)
    f32x16 gw1[2] = {f32x16{0}, f32x16{0}}, gwx = {0};
    // per-thread partial sums: column t & 63 of dw3 and db2 over the rows this thread visits, db3 (lanes with column 0),
    // column t & 127 of db1 over a quarter of the rows (t < 512)
    float g_w3 = 0.0f, g_b2 = 0.0f, g_b3 = 0.0f, g_b1 = 0.0f;
#ifdef MCMC_TB_STAMP
    uint64_t stamps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
        const int64_t next_base = row_base + TR < end ? row_base + TR : -1;
#ifdef MCMC_TB_STAMP
        const bool stamp_tile = blockIdx.x == 0 && blockIdx.y == 0 && row_base == begin + 2 * TR;
#endif
        TB_STAMP(0);
        // ---- phase A: the tile's x, h1 | h2 | h3 and dy in LDS
        if (SAVED) {
            // The forward kernel saved this tile's h1 | h2 | h3: no recompute, no masks.  The tile arrives by LDS-DMA, each half
            // while its LDS region is dead: h2 | h3 were issued in the previous tile's phase F (here: retired by vmcnt(0), then
            // this phase's barrier, read from phase B on), h1 is issued now (retired before phase D's barrier, read in phase E).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_x(l, xr);
            if (t < TR) l.y[t] = dyr;
            dma_pieces(act + ((int64_t)pair * tiles + row_base / TR) * kActPieces, l.h1, wave, 0, kKiBH1);
            if (next_base >= 0) {
                load_x(xr, xm, rows, F, next_base);
                load_dy(next_base);
            }
            barrier_lds_only();
        } else {
            forward_tile<MODE>(l, w0, xr, xm, rows, F, row_base, next_base, dm, pair, nullptr);
            if (t < TR) l.y[t] = dyr;
            __syncthreads();
            if (next_base >= 0) load_dy(next_base);
        }
        TB_STAMP(1);
        // ---- phase B: output layer.  dw3 += dy h3, db3 += dy; dz2 = dy w3 d(h3) over h3, db2 += dz2
        {
            const int k = t & 63;
            const float w3k = l.w3[k];
#pragma unroll
            for (int j = 0; j < TR / NW; ++j) {
                const int rr = wave + NW * j;
                const float yv = l.y[rr];
                const float hv = (float)l.h3[rr * S3 + k];
                g_w3 = fmaf(yv, hv, g_w3);
                g_b3 += yv;
                const float dz = yv * w3k * dfactor(hv, sc);
                g_b2 += dz;
                l.h3[rr * S3 + k] = (__bf16)dz;
            }
        }
        tile_barrier<SAVED>();
        TB_STAMP(2);
        // ---- phase C: dW2 += dz2^T h2 (waves 0-7)  |  the MFMA chain of dz1 = W2^T dz2 (waves 8-15)
        f32x16 ig = {0};
        if (wave < 8) {
            const int ln = fresh_lane();
#pragma unroll
            for (int ks = 0; ks < TR / 16; ++ks)
                gwx = mfma16(tr_frag(ln, l.h3, S3, 16 * ks, 32 * (wave >> 2)), tr_frag(ln, l.h2, S2, 16 * ks, 32 * (wave & 3)), gwx);
        } else {
            ig = input_grad_mfma<SAVED>(l.w2, SW2, H3, l.h3, S3, (wave - 8) & 3, (wave - 8) >> 2);
        }
        tile_barrier<SAVED>();
        TB_STAMP(3);
        // ---- phase D: dz1 over h2
        if (wave >= 8) input_grad_store(ig, l.h2, S2, (wave - 8) & 3, (wave - 8) >> 2, sc);
        if (SAVED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of h1 have landed
        tile_barrier<SAVED>();
        TB_STAMP(4);
        // ---- phase E: dW1 += dz1^T h1, and the MFMA chain of dz0 = W1^T dz1 (all waves; both only read the tiles)
        {
            const int ln = fresh_lane();
#ifndef TB_DW1_UNROLL
#define TB_DW1_UNROLL 1
#endif
#pragma unroll TB_DW1_UNROLL
            for (int ks = 0; ks < TR / 16; ++ks) {
                const bf16x8 b = tr_frag(ln, l.h1, S1, 16 * ks, 32 * (wave & 7));
#pragma unroll
                for (int j = 0; j < 2; ++j) gw1[j] = mfma16(tr_frag(ln, l.h2, S2, 16 * ks, 32 * (2 * (wave >> 3) + j)), b, gw1[j]);
            }
        }
        ig = input_grad_mfma<SAVED>(l.w1, SW1, H2, l.h2, S2, wave & 7, wave >> 3);
        if (wave < 8) {  // db1 += column sums of dz1 (here, not in phase G: h2 is dead from this phase's barrier on)
            const int col = t & (H2 - 1), r0 = (t >> 7) * (TR / 4);
            float a = 0.0f;
#pragma unroll 8
            for (int rr = 0; rr < TR / 4; ++rr) a += (float)l.h2[(r0 + rr) * S2 + col];
            g_b1 += a;
        }
        tile_barrier<SAVED>();
        TB_STAMP(5);
        // ---- phase F: dz0 over h1; the next tile's h2 | h3 start to arrive in their (dead) regions
        if (SAVED && next_base >= 0) dma_pieces(act + ((int64_t)pair * tiles + next_base / TR) * kActPieces, l.h1, wave, kKiBH1, kKiBAll);
        input_grad_store(ig, l.h1, S1, wave & 7, wave >> 3, sc);
        tile_barrier<SAVED>();
        TB_STAMP(6);
        // ---- phase G: dW0 | db0 (column F of the input tile is 1) += dz0^T x (waves 8-15)
        if (wave >= 8) {
            const int ln = fresh_lane();
#pragma unroll
            for (int ks = 0; ks < TR / 16; ++ks) gwx = mfma16(tr_frag(ln, l.h1, S1, 16 * ks, 32 * (wave - 8)), tr_frag(ln, l.xs, SX, 16 * ks, 0), gwx);
        }
        TB_STAMP(7);
        tile_barrier<SAVED>();
        TB_STAMP(8);
    }
    // ---- the block's partial, flat layout: register i of lane (r, h) holds [out = 32 tile + acc_feature(i, h)][in = 32 tile + r]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            g[flat_w_off(F, 1) + (int64_t)(32 * (2 * (wave >> 3) + j) + acc_feature(i, h)) * H1 + 32 * (wave & 7) + r] = gw1[j][i];
    if (wave < 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g[flat_w_off(F, 2) + (int64_t)(32 * (wave >> 2) + acc_feature(i, h)) * H2 + 32 * (wave & 3) + r] = gwx[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 32 * (wave - 8) + acc_feature(i, h);
            if (r < F) g[flat_w_off(F, 0) + (int64_t)o * F + r] = gwx[i];
            else if (r == F) g[flat_b_off(F, 0) + o] = gwx[i];
        }
    }
    if (SAVED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // per-thread partial sums -> LDS (the h1 tile is free: the loop ended on a barrier) -> one thread per element
    float* red = reinterpret_cast<float*>(l.h1);  // [NW][64] dw3 | [NW][64] db2 | [NW] db3 | [4][128] db1
    float* red_w3 = red;
    float* red_b2 = red_w3 + NW * H3;
    float* red_b3 = red_b2 + NW * H3;
    float* red_b1 = red_b3 + NW;
    static_assert((2 * NW * H3 + NW + 4 * H2) * 4 <= TR * S1 * 2, "reduction scratch fits the h1 tile");
    red_w3[wave * H3 + lane] = g_w3;
    red_b2[wave * H3 + lane] = g_b2;
    if (lane == 0) red_b3[wave] = g_b3;
    if (t < 4 * H2) red_b1[t] = g_b1;
    __syncthreads();
    if (t < H2) {
        g[flat_b_off(F, 1) + t] = red_b1[t] + red_b1[H2 + t] + red_b1[2 * H2 + t] + red_b1[3 * H2 + t];
    } else if (t < H2 + H3) {
        float a = 0.0f;
        for (int v = 0; v < NW; ++v) a += red_b2[v * H3 + t - H2];
        g[flat_b_off(F, 2) + t - H2] = a;
    } else if (t < H2 + 2 * H3) {
        float a = 0.0f;
        for (int v = 0; v < NW; ++v) a += red_w3[v * H3 + t - H2 - H3];
        g[flat_w_off(F, 3) + t - H2 - H3] = a;
    } else if (t == H2 + 2 * H3) {
        float a = 0.0f;
        for (int v = 0; v < NW; ++v) a += red_b3[v];
        g[flat_b_off(F, 3)] = a;
    }
#ifdef MCMC_TB_STAMP
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0)
        for (int i = 0; i < 9; ++i) reinterpret_cast<uint64_t*>(g)[i] = stamps[i];
#endif
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
    auto kern = act ? (mask_mode == 1 ? train_backward_bf16_kernel<1, true> : mask_mode == 2 ? train_backward_bf16_kernel<2, true> : train_backward_bf16_kernel<0, true>)
                    : (mask_mode == 1 ? train_backward_bf16_kernel<1, false> : mask_mode == 2 ? train_backward_bf16_kernel<2, false> : train_backward_bf16_kernel<0, false>);
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), kLdsBytes, "train_backward_bf16")) return rc;
    hipLaunchKernelGGL(kern, dim3(M * S, row_blocks), dim3(NT), kLdsBytes, s, xb, x_stride, rows, F, S, W,
                       make_mask(mask_mode, m0, m1, m2, mask_seed, p_drop), delta, dW, reinterpret_cast<const uint4*>(act));
    return check_launch("train_backward_bf16");
}

// bytes per pair of the bf16 saved-activation buffer for a batch of `rows` rows
int64_t train_act_bytes_bf16(int64_t rows) { return (rows + TR - 1) / TR * (int64_t)kActElems * 2; }

}  // namespace mcmc
