// Row f4: the Temporal Fusion Transformer leg on the device.
//
// Restates /root/reference/models/temporal_fusion.py for gfx950:
//   TemporalFusionTransformer.forward  :120-153  input projection -> feature GRN -> temporal GRN ->
//       nn.MultiheadAttention(64, 4 heads, dropout 0.1) -> final GRN -> Linear(64 -> 1)
//   GatedResidualNetwork.forward       :62-89    fc1 -> ELU -> dropout -> fc2 -> ELU; gate = sigmoid(W[x;h]);
//       LayerNorm(gate*h + (1-gate)*x)
//   TFTModel.train inner step          :242-256  prediction = output[:, -1, 0]
//   TFTModel.predict                   :289-294  same read-out in eval mode
// Both callers read only the LAST timestep of every window.  The two leading GRNs and the key /
// value projections need all T rows, but the attention query, the output projection, the final
// GRN and the read-out are evaluated for the last row only - the gradient is the same and the
// reference's [T x T] attention collapses to one query per head.
//
// One workgroup (4 waves) owns one window; wave w owns rows 8w..8w+7 and lane = feature, so a
// LayerNorm row statistic is one wave reduction and every dense layer is a per-wave [8 x 64] tile
// whose inputs are broadcast LDS reads.  All activations the backward pass needs stay in LDS
// (153 KB); weights stream from L2 in two layouts (P [out][in] for input gradients, PT [in][out]
// for the forward) so every global access is lane-contiguous.  fp32 VALU arithmetic: windows are
// 30 x 64, the step is launch/latency bound.
#include "common.hpp"
#include "elu.hpp"
#include "member_loss.hpp"
#include "philox.hpp"

namespace mcmc {

namespace {

constexpr int HD = 64;     // hidden size (temporal_fusion.py:160)
constexpr int NH = 4;      // heads (:161)
constexpr int DH = HD / NH;
constexpr int TMAX = 32;   // rows per window tile (seq_length 30, :157)
constexpr int NT = 256;
constexpr int XS = 28;     // LDS stride of the input window (>= kMaxInputDim, multiple of 4)
constexpr float kLnEps = 1e-5f;

// ---- flat parameter layout: the reference module's named_parameters() order --------------------
constexpr int kGrnW1 = 0, kGrnB1 = 4096, kGrnW2 = 4160, kGrnB2 = 8256, kGrnWg = 8320, kGrnBg = 16512, kGrnGamma = 16576,
              kGrnBeta = 16640, kGrnSize = 16704;
constexpr int kAttInW = 0, kAttInB = 12288, kAttOutW = 12480, kAttOutB = 16576, kAttSize = 16640;
__host__ __device__ inline int tft_off_proj_b(int F) { return 64 * F; }
__host__ __device__ inline int tft_off_grn(int F, int g) { return 64 * F + 64 + (g < 2 ? g * kGrnSize : 2 * kGrnSize + kAttSize); }
__host__ __device__ inline int tft_off_att(int F) { return 64 * F + 64 + 2 * kGrnSize; }
__host__ __device__ inline int tft_off_out(int F) { return 64 * F + 64 + 3 * kGrnSize + kAttSize; }
__host__ __device__ inline int tft_param_count(int F) { return tft_off_out(F) + 65; }

struct TftMasks {
    const uint8_t* g[3];  // GRN keep masks [B,T,64] (feature, temporal, final) or null
    const uint8_t* at;    // attention keep mask [B*heads,T,T] or null
    uint64_t seed;
    int mode;             // 0 none (eval), 1 host, 2 philox
    float keep_prob, scale;
};

struct Win {              // one window
    int b;                // index in the batch (host masks)
    uint32_t key;         // Philox key: the window's start row (independent of batch composition)
    int T;
};

__device__ __forceinline__ bool keep_grn(const TftMasks& ms, const Win& w, int which, int r, int o) {
    if (ms.mode == 1) return ms.g[which][((int64_t)w.b * w.T + r) * HD + o] != 0;
    return philox_mask_uniform(ms.seed, w.key, (uint32_t)r, (uint32_t)(which * HD + o)) < ms.keep_prob;
}
__device__ __forceinline__ bool keep_att(const TftMasks& ms, const Win& w, int h, int j) {
    if (ms.mode == 1) return ms.at[(((int64_t)w.b * NH + h) * w.T + (w.T - 1)) * w.T + j] != 0;
    return philox_mask_uniform(ms.seed, w.key, (uint32_t)(w.T - 1), (uint32_t)(3 * HD + h * TMAX + j)) < ms.keep_prob;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float elu(float z) { return z > 0.0f ? z : elu_neg(z); }
__device__ __forceinline__ float sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }
// derivative factor of a stored ELU(+dropout) activation; masked elements are stored as -0.0f
__device__ __forceinline__ float dfactor(float h, float scale) {
    if (__float_as_uint(h) == 0x80000000u) return 0.0f;
    return h > 0.0f ? scale : h + scale;
}

// acc[rr] += sum_{j<J} in[(row0+rr)*stride + j] * Mat[j*ld + c]   (J multiple of 4; `in` in LDS, broadcast reads)
template <int J>
__device__ __forceinline__ void dense8(const float* in, int stride, int row0, const float* __restrict__ Mat, int ld, int c,
                                       float (&acc)[8]) {
#pragma unroll 2
    for (int j = 0; j < J; j += 4) {
        const float w0 = Mat[(j + 0) * ld + c], w1 = Mat[(j + 1) * ld + c], w2 = Mat[(j + 2) * ld + c], w3 = Mat[(j + 3) * ld + c];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const float4 v = *reinterpret_cast<const float4*>(in + (row0 + rr) * stride + j);
            acc[rr] = fmaf(v.x, w0, acc[rr]);
            acc[rr] = fmaf(v.y, w1, acc[rr]);
            acc[rr] = fmaf(v.z, w2, acc[rr]);
            acc[rr] = fmaf(v.w, w3, acc[rr]);
        }
    }
}

// The same product as dense8<64> for ALL 32 rows of a tile on the matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32
// FMAs): acc[rr] += sum_{j<64} in[(8*wave+rr)*64 + j] * Mat[j*ld + lane].  Every wave of the workgroup takes part:
// wave = (mo, kh) computes output features 32*mo.. for k = 32*kh..32*kh+31 (16 MFMAs, its 16 weight loads in flight
// together, activations as four 16-byte LDS reads per lane: lane (row, h) feeds k = 8j + 4h + i at step (j, i)); the
// two k-halves meet in `stage` [2][32][65] (LDS) and come back in dense8's layout - rows 8*wave.., lane = feature.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStageFloats = 2 * TMAX * 65;
__device__ __forceinline__ void dense32_mfma(const float* in, const float* __restrict__ Mat, int ld, float* stage, float (&acc)[8]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5, mo = wv & 1, kh = wv >> 1;
    float av[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) av[4 * j + i] = Mat[(32 * kh + 8 * j + 4 * h + i) * ld + 32 * mo + r];
    f32x16 d = {0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(in + r * HD + 32 * kh + 8 * j + 4 * h);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * j + 0], q.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * j + 1], q.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * j + 2], q.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * j + 3], q.w, d, 0, 0, 0);
    }
    float* st = stage + kh * (TMAX * 65) + r * 65 + 32 * mo;  // register i of lane (r, h) holds D[feature acc_feature(i, h)][row r]
#pragma unroll
    for (int i = 0; i < 16; ++i) st[acc_feature(i, h)] = d[i];
    __syncthreads();
    const float* s0 = stage + (wv * 8) * 65 + lane;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) acc[rr] += s0[rr * 65] + s0[TMAX * 65 + rr * 65];
    __syncthreads();
}

// ---- LDS ------------------------------------------------------------------------------------------------
struct GrnBuf {
    float *h1, *h2, *gt, *xh, *rstd;  // [rows][64] x4, [rows]
};
struct Lds {
    float* x;    // [32][XS]
    float* u[3]; // inputs of GRN 0, 1 and of the attention: [32][64]
    GrnBuf g[3]; // g[2] has 8 rows
    float *K, *V;             // [32][64]
    float *qs, *prob, *probd, *ctx;  // [64], [4][32], [4][32], [64]
    float* a_out;  // attention output = input of the final GRN: [8][64], row 0 live
    float* out3;   // final GRN output [8][64], row 0 live
    // backward scratch
    float *dU, *dz, *dzT;  // [32][64], [32][64], [64][32]
    float* red;            // [4][64] cross-wave sums
    float *vec0, *vec1;    // [64] small vectors (do / dctx, dq)
    float* ds;             // [4][32]
};
constexpr int kLdsFloats = TMAX * XS + 3 * TMAX * HD + 2 * (4 * TMAX * HD + TMAX) + (4 * 8 * HD + 8) + 2 * TMAX * HD + (HD + 2 * NH * TMAX + HD) +
                           2 * 8 * HD + 3 * TMAX * HD + 4 * HD + 2 * HD + NH * TMAX;

// Everything the forward pass leaves in LDS (x ... out3) is one contiguous block at the start of the allocation: the
// train-mode forward kernel can save it (124 KB per window) and the backward kernel reload it instead of
// recomputing the window's forward pass (autograd's saved tensors).
constexpr int kFwdFloats = TMAX * XS + 3 * TMAX * HD + 2 * (4 * TMAX * HD + TMAX) + (4 * 8 * HD + 8) + 2 * TMAX * HD + (HD + 2 * NH * TMAX + HD) +
                           2 * 8 * HD;
static_assert(kFwdFloats % 4 == 0, "saved activations are copied as float4");

__device__ __forceinline__ Lds carve(float* p) {
    Lds l;
    auto take = [&](int n) { float* q = p; p += n; return q; };
    l.x = take(TMAX * XS);
    for (int i = 0; i < 3; ++i) l.u[i] = take(TMAX * HD);
    for (int i = 0; i < 3; ++i) {
        const int rows = i < 2 ? TMAX : 8;
        l.g[i].h1 = take(rows * HD);
        l.g[i].h2 = take(rows * HD);
        l.g[i].gt = take(rows * HD);
        l.g[i].xh = take(rows * HD);
        l.g[i].rstd = take(rows);
    }
    static_assert(kStageFloats <= 2 * TMAX * HD + HD, "the backward pass's MFMA staging tile aliases K | V | qs");
    l.K = take(TMAX * HD);
    l.V = take(TMAX * HD);
    l.qs = take(HD);
    l.prob = take(NH * TMAX);
    l.probd = take(NH * TMAX);
    l.ctx = take(HD);
    l.a_out = take(8 * HD);
    l.out3 = take(8 * HD);
    static_assert(kStageFloats <= 3 * TMAX * HD, "the forward pass's MFMA staging tile aliases dU | dz | dzT");
    l.dU = take(TMAX * HD);
    l.dz = take(TMAX * HD);
    l.dzT = take(HD * TMAX);
    l.red = take(4 * HD);
    l.vec0 = take(HD);
    l.vec1 = take(HD);
    l.ds = take(NH * TMAX);
    return l;
}

// ---- GRN forward on a row tile: wave `wv` owns rows 8wv..8wv+7 when wv < n_waves --------------------
// u -> out (both [rows][64] in LDS); stores h1 (post dropout), h2, gate, xhat, rstd for the backward pass.
// `row_of(r)` gives the window row of tile row r (masks); `live(r)` whether the row exists.
// WIDE (all four waves active, 32-row tiles): the four dense layers run on the matrix pipe (dense32_mfma, `stage` =
// kStageFloats of LDS scratch); otherwise (the final GRN's single live row) on the VALU.
template <bool TRAIN, bool WIDE>
__device__ __forceinline__ void grn_forward(const float* u, const GrnBuf& gb, float* out, const float* __restrict__ PT, int n_waves,
                                            const TftMasks& ms, const Win& w, int which, int row_shift, int n_live, float* stage) {
    const int o = threadIdx.x & 63, wv = threadIdx.x >> 6, r0 = wv * 8;
    const bool act = wv < n_waves;
    float acc[8];
    if (act) {
        const float b = PT[kGrnB1 + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        if (!WIDE) dense8<HD>(u, HD, r0, PT + kGrnW1, HD, o, acc);
    }
    if (WIDE) dense32_mfma(u, PT + kGrnW1, HD, stage, acc);
    if (act) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + rr;
            float h = elu(acc[rr]);
            if (TRAIN && ms.mode != 0) h = (r < n_live && keep_grn(ms, w, which, r + row_shift, o)) ? h * ms.scale : -0.0f;
            gb.h1[r * HD + o] = r < n_live ? h : -0.0f;
        }
    }
    __syncthreads();
    float h2v[8];
    if (act) {
        const float b = PT[kGrnB2 + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        if (!WIDE) dense8<HD>(gb.h1, HD, r0, PT + kGrnW2, HD, o, acc);
    }
    if (WIDE) dense32_mfma(gb.h1, PT + kGrnW2, HD, stage, acc);
    if (act) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            h2v[rr] = r0 + rr < n_live ? elu(acc[rr]) : 0.0f;
            gb.h2[(r0 + rr) * HD + o] = h2v[rr];
        }
    }
    __syncthreads();
    if (act) {
        const float b = PT[kGrnBg + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        if (!WIDE) {
            dense8<HD>(u, HD, r0, PT + kGrnWg, HD, o, acc);               // gate input = [u | h2]  (transposed gate: [128][64])
            dense8<HD>(gb.h2, HD, r0, PT + kGrnWg + HD * HD, HD, o, acc);
        }
    }
    if (WIDE) {
        dense32_mfma(u, PT + kGrnWg, HD, stage, acc);
        dense32_mfma(gb.h2, PT + kGrnWg + HD * HD, HD, stage, acc);
    }
    if (act) {
        const float gamma = PT[kGrnGamma + o], beta = PT[kGrnBeta + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + rr;
            const float g = sigmoidf(acc[rr]);
            const float uv = u[r * HD + o];
            const float wgt = g * h2v[rr] + (1.0f - g) * uv;
            const float mean = wave_sum(wgt) * (1.0f / HD);
            const float d = wgt - mean;
            const float var = wave_sum(d * d) * (1.0f / HD);
            const float rstd = 1.0f / sqrtf(var + kLnEps);
            const float xh = d * rstd;
            gb.gt[r * HD + o] = g;
            gb.xh[r * HD + o] = xh;
            if (o == 0) gb.rstd[r] = rstd;
            out[r * HD + o] = r < n_live ? fmaf(xh, gamma, beta) : 0.0f;
        }
    }
    __syncthreads();
}

// Touch every 128-byte line of a parameter image once at kernel start (a handful of loads per thread, all in flight
// at once): the AdamW step rewrote the image since the last launch, so the first reader in each XCD misses its L2,
// and without this the layers would take those misses one dependent round trip after the other.
// Returns a value the caller must keep alive (see `keep`).
__device__ __forceinline__ float touch_params(const float* __restrict__ Pimg, int n) {
    constexpr int kSweeps = (64 * kMaxInputDim + 64 + 3 * kGrnSize + kAttSize + 65 + NT * 32 - 1) / (NT * 32);  // largest image
    float v[kSweeps];
#pragma unroll
    for (int k = 0; k < kSweeps; ++k) {
        const int i = (k * NT + (int)threadIdx.x) * 32;
        v[k] = i < n ? Pimg[i] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kSweeps; ++k) s += v[k];
    return s;
}
__device__ __forceinline__ void keep(float v) { asm volatile("" ::"v"(v)); }

// ---- GRN forward for a tile with ONE live row (the final GRN: tile row 0 <-> window row `row`) -----------------
// The three matrix-vector products are split over the four waves (a quarter of k each, 16 or 32 weight loads in
// flight per lane) and the partial sums meet in `red` [4][64]; every wave then holds the same row.  Rows 1..7 of the
// stored buffers get neutral finite values (their upstream gradient is zero in the backward pass).
template <int KQ>
__device__ __forceinline__ float matvec_part(const float* vec, const float* __restrict__ Mat, int ld, int o, int k0) {
    float wq[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) wq[k] = Mat[(k0 + k) * ld + o];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < KQ; ++k) s = fmaf(vec[k0 + k], wq[k], s);
    return s;
}
__device__ __forceinline__ float cross_wave(float part, float bias, float* red) {
    const int o = threadIdx.x & 63, wv = threadIdx.x >> 6;
    red[wv * HD + o] = part;
    __syncthreads();
    const float r = bias + ((red[o] + red[HD + o]) + (red[2 * HD + o] + red[3 * HD + o]));
    __syncthreads();
    return r;
}
template <bool TRAIN>
__device__ __forceinline__ void grn_forward_row(const float* u, const GrnBuf& gb, float* out, const float* __restrict__ PT, const TftMasks& ms,
                                                const Win& w, int which, int row, float* red) {
    const int o = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto store8 = [&](float* buf, float live, float rest) {  // tile rows wv and wv + 4 of an [8][64] buffer
        buf[wv * HD + o] = wv == 0 ? live : rest;
        buf[(wv + 4) * HD + o] = rest;
    };
    float h1 = elu(cross_wave(matvec_part<16>(u, PT + kGrnW1, HD, o, 16 * wv), PT[kGrnB1 + o], red));
    if (TRAIN && ms.mode != 0) h1 = keep_grn(ms, w, which, row, o) ? h1 * ms.scale : -0.0f;
    store8(gb.h1, h1, -0.0f);
    __syncthreads();
    const float h2 = elu(cross_wave(matvec_part<16>(gb.h1, PT + kGrnW2, HD, o, 16 * wv), PT[kGrnB2 + o], red));
    store8(gb.h2, h2, 0.0f);
    __syncthreads();
    // gate input = [u | h2] (transposed gate: [128][64]): waves 0-1 take u, waves 2-3 take h2
    const float part = wv < 2 ? matvec_part<32>(u, PT + kGrnWg, HD, o, 32 * wv)
                              : matvec_part<32>(gb.h2, PT + kGrnWg + HD * HD, HD, o, 32 * (wv - 2));
    const float g = sigmoidf(cross_wave(part, PT[kGrnBg + o], red));
    const float wgt = g * h2 + (1.0f - g) * u[o];
    const float mean = wave_sum(wgt) * (1.0f / HD);
    const float d = wgt - mean;
    const float var = wave_sum(d * d) * (1.0f / HD);
    const float rstd = 1.0f / sqrtf(var + kLnEps);
    const float xh = d * rstd;
    store8(gb.gt, g, 0.5f);
    store8(gb.xh, xh, 0.0f);
    if (o == 0) {
        gb.rstd[wv] = wv == 0 ? rstd : 1.0f;
        gb.rstd[wv + 4] = 1.0f;
    }
    store8(out, fmaf(xh, PT[kGrnGamma + o], PT[kGrnBeta + o]), 0.0f);
    __syncthreads();
}

// ---- whole forward of one window; returns the prediction in every thread -----------------------------
template <bool TRAIN>
__device__ __forceinline__ float tft_forward_window(const Lds& l, const float* __restrict__ xw, int64_t rows_left, int F, const float* __restrict__ P,
                                                    const float* __restrict__ PT, const TftMasks& ms, const Win& w) {
    const int t = threadIdx.x, o = t & 63, wv = t >> 6, r0 = wv * 8, T = w.T;
    for (int i = t; i < TMAX * XS; i += NT) {
        const int r = i / XS, k = i % XS;
        l.x[i] = (r < T && k < F && r < rows_left) ? xw[(int64_t)r * F + k] : 0.0f;  // rows past the end of X read as 0
    }
    __syncthreads();
    {  // input projection (:130)
        float acc[8];
        const float b = PT[tft_off_proj_b(F) + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        float wk[kMaxInputDim];  // all columns' weights first: one L2 round trip
#pragma unroll
        for (int k = 0; k < kMaxInputDim; ++k) wk[k] = k < F ? PT[k * HD + o] : 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxInputDim; ++k)
            if (k < F) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) acc[rr] = fmaf(l.x[(r0 + rr) * XS + k], wk[k], acc[rr]);
            }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) l.u[0][(r0 + rr) * HD + o] = r0 + rr < T ? acc[rr] : 0.0f;
    }
    __syncthreads();
    grn_forward<TRAIN, true>(l.u[0], l.g[0], l.u[1], PT + tft_off_grn(F, 0), 4, ms, w, 0, 0, T, l.dU);
    grn_forward<TRAIN, true>(l.u[1], l.g[1], l.u[2], PT + tft_off_grn(F, 1), 4, ms, w, 1, 0, T, l.dU);
    // ---- attention (:139-144), query = last row only
    const float* A = PT + tft_off_att(F);  // transposed in_proj: [64][192]; out_proj [64][64]
    {
        float acc[8];
        float b = A[kAttInB + HD + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        dense32_mfma(l.u[2], A + kAttInW + HD, 3 * HD, l.dU, acc);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) l.K[(r0 + rr) * HD + o] = r0 + rr < T ? acc[rr] : 0.0f;
        b = A[kAttInB + 2 * HD + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr] = b;
        dense32_mfma(l.u[2], A + kAttInW + 2 * HD, 3 * HD, l.dU, acc);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) l.V[(r0 + rr) * HD + o] = r0 + rr < T ? acc[rr] : 0.0f;
        // query of the last row: a matrix-vector product split over the four waves
        const float q = cross_wave(matvec_part<16>(l.u[2] + (T - 1) * HD, A + kAttInW, 3 * HD, o, 16 * wv), A[kAttInB + o], l.red);
        if (wv == 0) l.qs[o] = q * 0.25f;  // q * sqrt(1/head_dim), head_dim = 16
    }
    __syncthreads();
    if (t < NH * TMAX) {  // scores + softmax: 32 lanes per head
        const int h = t >> 5, j = t & 31;
        float s = -INFINITY;
        if (j < T) {
            s = 0.0f;
#pragma unroll
            for (int c = 0; c < DH; ++c) s = fmaf(l.qs[h * DH + c], l.K[j * HD + h * DH + c], s);
        }
        float m = s;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float e = j < T ? expf(s - m) : 0.0f;
        float sum = e;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float a = e / sum;
        l.prob[t] = a;
        float ad = a;
        if (TRAIN && ms.mode != 0) ad = (j < T && keep_att(ms, w, h, j)) ? a * ms.scale : 0.0f;
        l.probd[t] = ad;
    }
    __syncthreads();
    if (t < HD) {
        const int h = t >> 4;
        float c = 0.0f;
        for (int j = 0; j < T; ++j) c = fmaf(l.probd[h * TMAX + j], l.V[j * HD + t], c);
        l.ctx[t] = c;
    }
    __syncthreads();
    {  // attention output projection of the last row -> tile row 0 of a_out, rows 1..7 zero
        const float ao = cross_wave(matvec_part<16>(l.ctx, A + kAttOutW, HD, o, 16 * wv), A[kAttOutB + o], l.red);
        static_assert(8 * HD == 2 * NT, "a_out is written as two elements per thread");
        l.a_out[t] = wv == 0 ? ao : 0.0f;
        l.a_out[t + NT] = 0.0f;
    }
    __syncthreads();
    // final GRN on the last row (tile row 0 <-> window row T-1)
    grn_forward_row<TRAIN>(l.a_out, l.g[2], l.out3, PT + tft_off_grn(F, 2), ms, w, 2, T - 1, l.red);
    const float* Wo = P + tft_off_out(F);
    float y = wave_sum(l.out3[o] * Wo[o]) + Wo[HD];
    return y;
}

// ---- backward helpers ---------------------------------------------------------------------------------------
// dW[o][col0 + k] = sum_{r<R} dzT[o][r] * act[r*stride + k]  for the 16 o's of this wave; written (not added) to g
template <int R>
__device__ __forceinline__ void wgrad(const float* dzT, const float* act, int stride, int kmax, float* __restrict__ g, int ld, int col0) {
    const int k = threadIdx.x & 63, og = threadIdx.x >> 6;
    if (k >= kmax) return;
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = act[r * stride + k];
#pragma unroll 1
    for (int o = og * 16; o < og * 16 + 16; ++o) {
        float s = 0.0f;
#pragma unroll
        for (int r4 = 0; r4 < R; r4 += 4) {
            const float4 z = *reinterpret_cast<const float4*>(dzT + o * TMAX + r4);
            s = fmaf(z.x, a[r4], s);
            s = fmaf(z.y, a[r4 + 1], s);
            s = fmaf(z.z, a[r4 + 2], s);
            s = fmaf(z.w, a[r4 + 3], s);
        }
        g[o * ld + col0 + k] = s;
    }
}
// The same weight gradient for a full tile (R = 32 rows, 64 x 64 weights, activation stride 64) on the matrix pipe:
// D[o][k] = sum_r dzT[o][r] * act[r][k]; wave = (mo, nk) owns the 32 x 32 block of outputs 32*mo.. and inputs 32*nk..;
// lane (m, h) feeds dzT[o = 32*mo + m][r] and act[r][k = 32*nk + m] for r = 8j + 4h + i at step (j, i).
__device__ __forceinline__ void wgrad32_mfma(const float* dzT, const float* act, float* __restrict__ g, int ld, int col0) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 31, h = lane >> 5, mo = wv & 1, nk = wv >> 1;
    f32x16 d = {0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(dzT + (32 * mo + m) * TMAX + 8 * j + 4 * h);
        const float* b = act + (8 * j + 4 * h) * HD + 32 * nk + m;
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(q.y, b[HD], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(q.z, b[2 * HD], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(q.w, b[3 * HD], d, 0, 0, 0);
    }
    float* out = g + (32 * mo) * ld + col0 + 32 * nk + m;  // register i of lane (m, h): D[o = acc_feature(i, h)][k = m]
#pragma unroll
    for (int i = 0; i < 16; ++i) out[acc_feature(i, h) * ld] = d[i];
}
// bias gradient: g[o] = sum_r dzT[o][r]
template <int R>
__device__ __forceinline__ void bgrad(const float* dzT, float* __restrict__ g) {
    if (threadIdx.x < HD) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) s += dzT[threadIdx.x * TMAX + r];
        g[threadIdx.x] = s;
    }
}
// store a per-thread [8] tile into dz (row-major) and dzT (transposed); inactive waves store zeros
__device__ __forceinline__ void store_dz(const Lds& l, const float (&v)[8], bool act) {
    const int o = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 8;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const float x = act ? v[rr] : 0.0f;
        l.dz[(r0 + rr) * HD + o] = x;
        l.dzT[o * TMAX + r0 + rr] = x;
    }
}

// GRN backward: upstream l.dU (gradient w.r.t. the GRN output) -> l.dU (gradient w.r.t. its input u).
template <int R>
__device__ __forceinline__ void grn_backward(const Lds& l, const float* u, const GrnBuf& gb, const float* __restrict__ Pg, float* __restrict__ G,
                                             int n_waves, float scale) {
    const int o = threadIdx.x & 63, wv = threadIdx.x >> 6, r0 = wv * 8;
    const bool act = wv < n_waves;
    float du[8], dh2[8], tmp[8];
    float dgam = 0.0f, dbet = 0.0f;
    if (act) {
        const float gamma = Pg[kGrnGamma + o];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + rr;
            const float dy = l.dU[r * HD + o], xh = gb.xh[r * HD + o];
            dgam = fmaf(dy, xh, dgam);
            dbet += dy;
            const float dxh = dy * gamma;
            const float m1 = wave_sum(dxh) * (1.0f / HD);
            const float m2 = wave_sum(dxh * xh) * (1.0f / HD);
            const float dw = gb.rstd[r] * (dxh - m1 - xh * m2);
            const float g = gb.gt[r * HD + o], h2v = gb.h2[r * HD + o], uv = u[r * HD + o];
            tmp[rr] = dw * (h2v - uv) * g * (1.0f - g);  // d gate pre-activation
            dh2[rr] = dw * g;
            du[rr] = dw * (1.0f - g);
        }
    }
    l.red[wv * HD + o] = act ? dgam : 0.0f;
    store_dz(l, tmp, act);
    __syncthreads();
    if (threadIdx.x < HD) G[kGrnGamma + o] = l.red[o] + l.red[HD + o] + l.red[2 * HD + o] + l.red[3 * HD + o];
    __syncthreads();
    l.red[wv * HD + o] = act ? dbet : 0.0f;
    // gate: dWg = dgp^T [u | h2], dbg; d[u | h2] += dgp Wg
    if (R == TMAX) wgrad32_mfma(l.dzT, u, G + kGrnWg, 2 * HD, 0); else wgrad<R>(l.dzT, u, HD, HD, G + kGrnWg, 2 * HD, 0);
    if (R == TMAX) wgrad32_mfma(l.dzT, gb.h2, G + kGrnWg, 2 * HD, HD); else wgrad<R>(l.dzT, gb.h2, HD, HD, G + kGrnWg, 2 * HD, HD);
    bgrad<R>(l.dzT, G + kGrnBg);
    // input gradients: 32-row tiles on the matrix pipe (staging tile = K | V | qs, dead once the attention backward is
    // done), the final GRN's single row on the VALU
    constexpr bool WIDE = R == TMAX;
    float* stage = l.K;
    if (WIDE) {
        dense32_mfma(l.dz, Pg + kGrnWg, 2 * HD, stage, du);
        dense32_mfma(l.dz, Pg + kGrnWg + HD, 2 * HD, stage, dh2);
    }
    if (act) {
        if (!WIDE) {
            dense8<HD>(l.dz, HD, r0, Pg + kGrnWg, 2 * HD, o, du);
            dense8<HD>(l.dz, HD, r0, Pg + kGrnWg + HD, 2 * HD, o, dh2);
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const float h2v = gb.h2[(r0 + rr) * HD + o];
            tmp[rr] = dh2[rr] * (h2v > 0.0f ? 1.0f : h2v + 1.0f);  // d a2
        }
    }
    __syncthreads();
    if (threadIdx.x < HD) G[kGrnBeta + o] = l.red[o] + l.red[HD + o] + l.red[2 * HD + o] + l.red[3 * HD + o];
    store_dz(l, tmp, act);
    __syncthreads();
    if (R == TMAX) wgrad32_mfma(l.dzT, gb.h1, G + kGrnW2, HD, 0); else wgrad<R>(l.dzT, gb.h1, HD, HD, G + kGrnW2, HD, 0);
    bgrad<R>(l.dzT, G + kGrnB2);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) tmp[rr] = 0.0f;
    if (WIDE) dense32_mfma(l.dz, Pg + kGrnW2, HD, stage, tmp);
    if (act) {
        if (!WIDE) dense8<HD>(l.dz, HD, r0, Pg + kGrnW2, HD, o, tmp);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) tmp[rr] *= dfactor(gb.h1[(r0 + rr) * HD + o], scale);  // d a1
    }
    __syncthreads();
    store_dz(l, tmp, act);
    __syncthreads();
    if (R == TMAX) wgrad32_mfma(l.dzT, u, G + kGrnW1, HD, 0); else wgrad<R>(l.dzT, u, HD, HD, G + kGrnW1, HD, 0);
    bgrad<R>(l.dzT, G + kGrnB1);
    if (WIDE) dense32_mfma(l.dz, Pg + kGrnW1, HD, stage, du);
    if (act) {
        if (!WIDE) dense8<HD>(l.dz, HD, r0, Pg + kGrnW1, HD, o, du);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) l.dU[(r0 + rr) * HD + o] = du[rr];
    }
    __syncthreads();
}

// GRN backward for a tile with ONE live row (the final GRN): upstream l.dU row 0 -> l.dU row 0.  Weight gradients are
// outer products (wave wv writes output rows 16*wv..), the four input-gradient products are matrix-vector products
// split over the four waves like grn_forward_row.  `vec` = 64 floats of LDS scratch.
__device__ __forceinline__ void grn_backward_row(const Lds& l, const float* u, const GrnBuf& gb, const float* __restrict__ Pg,
                                                 float* __restrict__ G, float scale, float* vec) {
    const int o = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float dy = l.dU[o], xh = gb.xh[o];
    const float dxh = dy * Pg[kGrnGamma + o];
    const float m1 = wave_sum(dxh) * (1.0f / HD);
    const float m2 = wave_sum(dxh * xh) * (1.0f / HD);
    const float dw = gb.rstd[0] * (dxh - m1 - xh * m2);
    const float g = gb.gt[o], h2v = gb.h2[o], uv = u[o], h1v = gb.h1[o];
    const float dgp = dw * (h2v - uv) * g * (1.0f - g);  // d gate pre-activation
    float dh2 = dw * g, du = dw * (1.0f - g);
    if (wv == 0) {
        G[kGrnGamma + o] = dy * xh;
        G[kGrnBeta + o] = dy;
        G[kGrnBg + o] = dgp;
        vec[o] = dgp;
    }
    __syncthreads();
    // gate: dWg[oo][k] = dgp[oo] [u | h2][k];  d[u | h2] += dgp Wg
    for (int oo = 16 * wv; oo < 16 * wv + 16; ++oo) {
        const float d = vec[oo];
        G[kGrnWg + oo * 2 * HD + o] = d * uv;
        G[kGrnWg + oo * 2 * HD + HD + o] = d * h2v;
    }
    du += cross_wave(matvec_part<16>(vec, Pg + kGrnWg, 2 * HD, o, 16 * wv), 0.0f, l.red);
    dh2 += cross_wave(matvec_part<16>(vec, Pg + kGrnWg + HD, 2 * HD, o, 16 * wv), 0.0f, l.red);
    const float da2 = dh2 * (h2v > 0.0f ? 1.0f : h2v + 1.0f);
    if (wv == 0) {
        G[kGrnB2 + o] = da2;
        vec[o] = da2;  // every wave finished reading vec before cross_wave's barriers
    }
    __syncthreads();
    for (int oo = 16 * wv; oo < 16 * wv + 16; ++oo) G[kGrnW2 + oo * HD + o] = vec[oo] * h1v;
    const float da1 = cross_wave(matvec_part<16>(vec, Pg + kGrnW2, HD, o, 16 * wv), 0.0f, l.red) * dfactor(h1v, scale);
    if (wv == 0) {
        G[kGrnB1 + o] = da1;
        vec[o] = da1;
    }
    __syncthreads();
    for (int oo = 16 * wv; oo < 16 * wv + 16; ++oo) G[kGrnW1 + oo * HD + o] = vec[oo] * uv;
    du += cross_wave(matvec_part<16>(vec, Pg + kGrnW1, HD, o, 16 * wv), 0.0f, l.red);
    if (wv == 0) l.dU[o] = du;
    __syncthreads();
}

// ---- kernels ---------------------------------------------------------------------------------------------------
// index of flat parameter i in the transposed image PT (every weight matrix [out][in] stored as [in][out]; vectors in place)
__device__ __forceinline__ int tft_transposed_index(int i, int F) {
    // find the matrix this element belongs to: (base, rows=out, cols=in) or a vector
    int base = 0, rows = 0, cols = 0;
    auto in_mat = [&](int b, int r, int c) {
        if (i >= b && i < b + r * c) {
            base = b;
            rows = r;
            cols = c;
            return true;
        }
        return false;
    };
    bool m = in_mat(0, HD, F);
    for (int g = 0; g < 3 && !m; ++g) {
        const int gb = tft_off_grn(F, g);
        m = in_mat(gb + kGrnW1, HD, HD) || in_mat(gb + kGrnW2, HD, HD) || in_mat(gb + kGrnWg, HD, 2 * HD);
    }
    if (!m) m = in_mat(tft_off_att(F) + kAttInW, 3 * HD, HD) || in_mat(tft_off_att(F) + kAttOutW, HD, HD);
    if (!m) return i;
    const int e = i - base, r = e / cols, c = e % cols;
    return base + c * rows + r;
}

__global__ __launch_bounds__(256) void tft_transpose_kernel(const float* __restrict__ P, float* __restrict__ PT, int F) {
    const int n = tft_param_count(F);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) PT[tft_transposed_index(i, F)] = P[i];
}

template <bool TRAIN>
__global__ __launch_bounds__(NT, 1) void tft_forward_kernel(const float* __restrict__ X, int64_t n_rows, const int* __restrict__ start, int T, int F,
                                                            const float* __restrict__ P, const float* __restrict__ PT, TftMasks ms,
                                                            float* __restrict__ pred, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds l = carve(smem);
    Win w;
    w.b = blockIdx.x;
    w.key = (uint32_t)start[blockIdx.x];
    w.T = T;
    int64_t s0 = start[blockIdx.x];
    if (s0 < 0 || s0 > n_rows) s0 = n_rows;  // an invalid start reads an all-zero window instead of faulting
    const float warm = TRAIN ? touch_params(PT, tft_param_count(F)) : 0.0f;  // eval: repeated predicts find the image in L2
    const float y = tft_forward_window<TRAIN>(l, X + s0 * F, n_rows - s0, F, P, PT, ms, w);
    keep(warm);
    if (threadIdx.x == 0) pred[blockIdx.x] = y;
    if (act) {  // keep the window's activations for the backward kernel
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(act + (int64_t)blockIdx.x * kFwdFloats);
        const float4* src = reinterpret_cast<const float4*>(smem);
        for (int i = threadIdx.x; i < kFwdFloats / 4; i += NT) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(NT, 1) void tft_backward_kernel(const float* __restrict__ X, int64_t n_rows, const int* __restrict__ start, int T, int F,
                                                             const float* __restrict__ P, const float* __restrict__ PT, TftMasks ms,
                                                             const float* __restrict__ delta, float* __restrict__ dP,
                                                             const float* __restrict__ act, TftLossFuse lf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds l = carve(smem);
    Win w;
    w.b = blockIdx.x;
    w.key = (uint32_t)start[blockIdx.x];
    w.T = T;
    const int t = threadIdx.x, o = t & 63, wv = t >> 6, r0 = wv * 8;
    // dLoss/dpred of this window: read, or derived here - every workgroup repeats the batch's two loss reductions over the
    // n_windows predictions (a few microseconds) instead of waiting for a separate loss launch
    float dy;
    if (lf.pred) {
        const LossCoef c = member_loss<NT>(lf.pred, lf.yb, 0, 1, (int64_t)gridDim.x, lf.alpha, lf.beta, blockIdx.x == 0, lf.loss, lf.loss_acc);
        dy = loss_delta(c, lf.pred[blockIdx.x], lf.yb[blockIdx.x]);
    } else {
        dy = delta[blockIdx.x];
    }
    int64_t s0 = start[blockIdx.x];
    if (s0 < 0 || s0 > n_rows) s0 = n_rows;
    const float warm = (act ? 0.0f : touch_params(PT, tft_param_count(F))) + touch_params(P, tft_param_count(F));
    if (act) {  // the forward kernel saved this window's activations: reload them
        const float4* src = reinterpret_cast<const float4*>(act + (int64_t)blockIdx.x * kFwdFloats);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = t; i < kFwdFloats / 4; i += NT) dst[i] = src[i];
        __syncthreads();
    } else {
        tft_forward_window<true>(l, X + s0 * F, n_rows - s0, F, P, PT, ms, w);
    }
    keep(warm);
    float* G = dP + (int64_t)blockIdx.x * tft_param_count(F);
    const float sc = ms.mode != 0 ? ms.scale : 1.0f;
    // ---- read-out (:149-150, last row): dWout, dbout, upstream of the final GRN in tile row 0
    {
        const float* Wo = P + tft_off_out(F);
        float* Go = G + tft_off_out(F);
        if (t < HD) Go[t] = dy * l.out3[t];
        if (t == HD) Go[HD] = dy;
        for (int i = t; i < TMAX * HD; i += NT) l.dU[i] = i < HD ? dy * Wo[i] : 0.0f;
    }
    __syncthreads();
    grn_backward_row(l, l.a_out, l.g[2], P + tft_off_grn(F, 2), G + tft_off_grn(F, 2), sc, l.vec1);
    // ---- attention backward; l.dU row 0 = gradient w.r.t. the attention output of the last row
    const float* A = P + tft_off_att(F);
    float* GA = G + tft_off_att(F);
    if (t < HD) {
        const float d_o = l.dU[t];
        l.vec0[t] = d_o;
        GA[kAttOutB + t] = d_o;
    }
    __syncthreads();
    {  // dWo[o][c] = do[o] ctx[c];  dctx[c] = sum_o do[o] Wo[o][c]
        const float c = l.ctx[o];
        for (int oo = wv * 16; oo < wv * 16 + 16; ++oo) GA[kAttOutW + oo * HD + o] = l.vec0[oo] * c;
        const float dc = cross_wave(matvec_part<16>(l.vec0, A + kAttOutW, HD, o, 16 * wv), 0.0f, l.red);
        if (wv == 0) l.vec1[o] = dc;  // dctx
    }
    __syncthreads();
    if (t < NH * TMAX) {  // softmax backward, 32 lanes per head
        const int h = t >> 5, j = t & 31;
        float da = 0.0f;
        if (j < T) {
#pragma unroll
            for (int c = 0; c < DH; ++c) da = fmaf(l.vec1[h * DH + c], l.V[j * HD + h * DH + c], da);
            if (ms.mode != 0) da = keep_att(ms, w, h, j) ? da * ms.scale : 0.0f;
        }
        const float a = l.prob[t];
        float dot = a * da;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        l.ds[t] = a * (da - dot);
    }
    __syncthreads();
    {
        // dK[r][o] = ds[h][r] * qs[o];  dq[o] = 0.25 * sum_j ds[h][j] K[j][o]
        const int h = o >> 4;
        float dk[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) dk[rr] = r0 + rr < T ? l.ds[h * TMAX + r0 + rr] * l.qs[o] : 0.0f;
        store_dz(l, dk, true);
        if (wv == 0) {
            float s = 0.0f;
            for (int j = 0; j < T; ++j) s = fmaf(l.ds[h * TMAX + j], l.K[j * HD + o], s);
            s *= 0.25f;
            l.vec0[o] = s;  // dq (all reads of the old vec0 = do finished before the previous barrier)
            GA[kAttInB + o] = s;
        }
    }
    __syncthreads();
    {  // query rows of in_proj: dWq[o][k] = dq[o] u_last[k]
        const float ul = l.u[2][(T - 1) * HD + o];
        for (int oo = wv * 16; oo < wv * 16 + 16; ++oo) GA[kAttInW + oo * HD + o] = l.vec0[oo] * ul;
    }
    wgrad32_mfma(l.dzT, l.u[2], GA + kAttInW + HD * HD, HD, 0);  // key rows
    bgrad<TMAX>(l.dzT, GA + kAttInB + HD);
    float du3[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) du3[rr] = 0.0f;
    dense32_mfma(l.dz, A + kAttInW + HD * HD, HD, l.K, du3);           // dK Wk (K, V, qs are dead: staging tile)
    const float s = cross_wave(matvec_part<16>(l.vec0, A + kAttInW, HD, o, 16 * wv), 0.0f, l.red);
    if (wv == (T - 1) / 8) {                                          // + Wq^T dq on the last row
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
            if (r0 + rr == T - 1) du3[rr] += s;
    }
    __syncthreads();
    {  // dV[r][o] = probd[h][r] * dctx[o]
        const int h = o >> 4;
        float dv[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) dv[rr] = r0 + rr < T ? l.probd[h * TMAX + r0 + rr] * l.vec1[o] : 0.0f;
        store_dz(l, dv, true);
    }
    __syncthreads();
    wgrad32_mfma(l.dzT, l.u[2], GA + kAttInW + 2 * HD * HD, HD, 0);  // value rows
    bgrad<TMAX>(l.dzT, GA + kAttInB + 2 * HD);
    dense32_mfma(l.dz, A + kAttInW + 2 * HD * HD, HD, l.K, du3);           // + dV Wv
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) l.dU[(r0 + rr) * HD + o] = du3[rr];
    __syncthreads();
    grn_backward<TMAX>(l, l.u[1], l.g[1], P + tft_off_grn(F, 1), G + tft_off_grn(F, 1), 4, sc);
    grn_backward<TMAX>(l, l.u[0], l.g[0], P + tft_off_grn(F, 0), G + tft_off_grn(F, 0), 4, sc);
    // ---- input projection: dWp[o][k] = sum_r dU[r][o] x[r][k], dbp
    {
        float v[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) v[rr] = l.dU[(r0 + rr) * HD + o];
        __syncthreads();
        store_dz(l, v, true);
        __syncthreads();
        wgrad<TMAX>(l.dzT, l.x, XS, F, G, F, 0);
        bgrad<TMAX>(l.dzT, G + tft_off_proj_b(F));
    }
}

// ---- flat optimiser: grad = sum_b dP[b], clip_grad_norm_, AdamW ------------------------------------------------
constexpr int kFlatBlocks = 256;  // about one parameter per thread at 67,585 parameters; <= 256: one partial per thread in flat_adamw

__global__ __launch_bounds__(256) void flat_grad_kernel(const float* __restrict__ dP, int B, int64_t n, float* __restrict__ grad,
                                                       double* __restrict__ normsq_part) {
    __shared__ double red[4];
    double nsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.0f;
        for (int b0 = 0; b0 < B; b0 += 8) {  // eight windows' loads in flight, summed in window order: deterministic
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = b0 + u < B ? dP[(int64_t)(b0 + u) * n + i] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u < B) a += d[u];
        }
        grad[i] = a;
        nsq += (double)a * a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nsq += __shfl_down(nsq, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nsq;
    __syncthreads();
    if (threadIdx.x == 0) normsq_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct FlatHyper {
    float lr, beta1, beta2, eps, weight_decay, max_norm, bias1, bias2_sqrt;
};

__global__ __launch_bounds__(256) void flat_adamw_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ am,
                                                        float* __restrict__ av, int64_t n, const double* __restrict__ normsq_part,
                                                        int nparts, FlatHyper hp, float* __restrict__ gnorm, float* __restrict__ PT, int F) {
    // squared norm from the per-block partials (nparts <= 256): one partial per thread, a fixed shuffle tree per wave,
    // then the four wave sums in order - the same value in every workgroup and every run
    __shared__ double wsum[4];
    double v = (int)threadIdx.x < nparts ? normsq_part[threadIdx.x] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    const double nsq = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const float total = (float)sqrt(nsq);
    if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm) gnorm[0] = total;
    float coef = hp.max_norm / (total + 1e-6f);  // torch.nn.utils.clip_grad_norm_
    coef = coef > 1.0f ? 1.0f : coef;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gr = grad[i] * coef;
        float p = param[i] * (1.0f - hp.lr * hp.weight_decay);
        const float mm = am[i] + (gr - am[i]) * (1.0f - hp.beta1);
        const float vv = av[i] * hp.beta2 + (1.0f - hp.beta2) * gr * gr;
        am[i] = mm;
        av[i] = vv;
        p -= (hp.lr / hp.bias1) * (mm / (sqrtf(vv) / hp.bias2_sqrt + hp.eps));
        param[i] = p;
        if (PT) PT[tft_transposed_index((int)i, F)] = p;  // the TFT's transposed image stays in step with the parameters
    }
}

TftMasks make_masks(int mode, const uint8_t* g0, const uint8_t* g1, const uint8_t* at, const uint8_t* g2, uint64_t seed, float p_drop) {
    TftMasks ms;
    ms.g[0] = g0;
    ms.g[1] = g1;
    ms.g[2] = g2;
    ms.at = at;
    ms.seed = seed;
    ms.mode = mode;
    ms.keep_prob = 1.0f - p_drop;
    ms.scale = (float)(1.0 / (1.0 - (double)p_drop));
    return ms;
}

template <typename K>
int set_lds(K kernel, const char* what) {
    return ensure_dyn_lds(reinterpret_cast<const void*>(kernel), kLdsFloats * sizeof(float), what);
}

}  // namespace

int tft_params(int F) { return tft_param_count(F); }
int tft_flat_blocks() { return kFlatBlocks; }
int64_t tft_act_floats() { return kFwdFloats; }

int launch_tft_transpose(const float* P, float* PT, int F, hipStream_t s) {
    hipLaunchKernelGGL(tft_transpose_kernel, dim3(64), dim3(256), 0, s, P, PT, F);
    return check_launch("tft_transpose");
}

int launch_tft_forward(const float* X, int64_t n_rows, const int* start, int B, int T, int F, const float* P, const float* PT, int mask_mode,
                       const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, const uint8_t* m3, uint64_t seed, float p_drop, float* pred,
                       float* act, hipStream_t s) {
    const size_t lds = kLdsFloats * sizeof(float);
    const TftMasks ms = make_masks(mask_mode, m0, m1, m2, m3, seed, p_drop);
    if (mask_mode == 0) {
        if (int rc = set_lds(tft_forward_kernel<false>, "tft_forward")) return rc;
        hipLaunchKernelGGL(tft_forward_kernel<false>, dim3(B), dim3(NT), lds, s, X, n_rows, start, T, F, P, PT, ms, pred, act);
    } else {
        if (int rc = set_lds(tft_forward_kernel<true>, "tft_forward")) return rc;
        hipLaunchKernelGGL(tft_forward_kernel<true>, dim3(B), dim3(NT), lds, s, X, n_rows, start, T, F, P, PT, ms, pred, act);
    }
    return check_launch("tft_forward");
}

int launch_tft_backward(const float* X, int64_t n_rows, const int* start, int B, int T, int F, const float* P, const float* PT, int mask_mode,
                        const uint8_t* m0, const uint8_t* m1, const uint8_t* m2, const uint8_t* m3, uint64_t seed, float p_drop,
                        const float* delta, float* dP, const float* act, const TftLossFuse* fuse, hipStream_t s) {
    if (int rc = set_lds(tft_backward_kernel, "tft_backward")) return rc;
    const TftLossFuse lf = fuse ? *fuse : TftLossFuse{nullptr, nullptr, 0.0f, 0.0f, nullptr, nullptr};
    hipLaunchKernelGGL(tft_backward_kernel, dim3(B), dim3(NT), kLdsFloats * sizeof(float), s, X, n_rows, start, T, F, P, PT,
                       make_masks(mask_mode, m0, m1, m2, m3, seed, p_drop), delta, dP, act, lf);
    return check_launch("tft_backward");
}

int launch_flat_update(float* param, int64_t n, const float* dP, int B, float* grad, float* am, float* av, double* normsq_part, float lr,
                       float beta1, float beta2, float adam_eps, float weight_decay, float max_norm, int64_t step, float* gnorm,
                       float* PT, int F, hipStream_t s) {
    hipLaunchKernelGGL(flat_grad_kernel, dim3(kFlatBlocks), dim3(256), 0, s, dP, B, n, grad, normsq_part);
    FlatHyper hp;
    hp.lr = lr;
    hp.beta1 = beta1;
    hp.beta2 = beta2;
    hp.eps = adam_eps;
    hp.weight_decay = weight_decay;
    hp.max_norm = max_norm;
    hp.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
    hp.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(flat_adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, param, grad, am, av, n, normsq_part, kFlatBlocks, hp, gnorm, PT, F);
    return check_launch("flat_update");
}

}  // namespace mcmc
