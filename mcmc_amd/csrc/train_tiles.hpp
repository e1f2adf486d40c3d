// Tile code of the training step (train.hip includes this file once per (workgroup size, rows per tile), inside a
// namespace, with MCMC_TRAIN_NT and MCMC_TRAIN_TR defined): dense layers on LDS tiles, the forward tile, the loss reductions and the
// forward / loss / backward kernels.  No include guard on purpose.
constexpr int NT = MCMC_TRAIN_NT;  // threads per workgroup
constexpr int NW = NT / 64;
constexpr int kWgPerCu = NT <= 256 ? 2 : 1;
constexpr int TR = MCMC_TRAIN_TR;  // rows per LDS tile: 32, or 64 = two MFMA row tiles per pass of the tile code (large batches)
constexpr int RT = TR / 32;
static_assert(TR % 32 == 0 && NT >= 8 * TR, "layer 3 gives a row eight threads");

struct Lds {
    float* xs;   // [TR][SX]  input tile, columns >= F are zero
    float* h1;   // [TR][S1]  activations; the backward pass overwrites each tile in place with the gradient
    float* h2;   // [TR][S2]  w.r.t. its pre-activations once the weight gradient that needs the activations is done
    float* h3;   // [TR][S3]
    float* y;    // [TR] outputs, then dLoss/dy in the backward pass
};
constexpr int kLdsFloats = TR * (SX + S1 + S2 + S3 + 1);  // 62 KB at 32 rows: two workgroups per CU, forward and backward; 124 KB at 64
// h1, h2, h3 of one tile are contiguous in LDS: the forward kernel can save them (57.7 KB per pair and tile) so that
// the backward kernel of a small batch reloads them from L2 instead of recomputing the tile's forward pass
constexpr int kActFloats = TR * (S1 + S2 + S3);
constexpr int kActEarlyFloats = (H3 / 32) * RT < NW ? TR * (S1 + S2) : 0;  // h1 | h2: saved during layer 2 by its idle waves (forward_tile)
static_assert(kActFloats % 4 == 0 && (TR * SX) % 4 == 0 && kActEarlyFloats % 4 == 0, "activation tiles are copied as float4");

__device__ __forceinline__ Lds carve(float* base) {
    Lds l;
    l.xs = base;
    l.h1 = l.xs + TR * SX;
    l.h2 = l.h1 + TR * S1;
    l.h3 = l.h2 + TR * S2;
    l.y = l.h3 + TR * S3;
    return l;
}


// ---- dense layers on LDS tiles, MFMA ---------------------------------------------------------------
// Z^T[out, row] = b[out] + sum_k WT[k][out] * hin[row][k]; then h = elu(z) * mask * scale -> hout[row][out].
// A comes straight from the pre-sampled WT image in global memory / L2 (32 consecutive floats per lane half),
// B from the LDS tile.  Wave w owns output tiles w, w+4, ...  Masked elements and padding rows are stored as
// -0.0f: arithmetic zero, and recognisable in the backward pass.
template <int MODE, int OUT, int IN_STRIDE, int OUT_STRIDE>
__device__ __forceinline__ void dense_forward(const float* __restrict__ hin, int K, const float* __restrict__ WT,
                                              const float* __restrict__ bias, float* __restrict__ hout, const DropMask& dm, int pair,
                                              int64_t rows, int64_t row_base, int layer) {
    // the wave id as a scalar: tile bases then live in SGPRs and every weight load is (scalar base) + (one per-lane offset)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    // a unit = one 32 x 32 output tile: (feature tile mo, row tile rt); wave w owns units w, w + NW, ...
#pragma unroll 1
    for (int unit = wave; unit < (OUT / 32) * RT; unit += NW) {
        const int mo = unit / RT, rr = 32 * (unit % RT) + r;  // rr: this lane's row inside the LDS tile
        const bool live = row_base + rr < rows;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias[32 * mo + acc_feature(i, h)];
        const float* a = WT + 32 * mo;            // uniform
        const unsigned al = (unsigned)(h * OUT + r);  // this lane's element offset inside every k-step pair
        const float* b = hin + rr * IN_STRIDE;
        // every weight is used once per tile, so the A stream is bound by L2 latency: issue a batch of loads,
        // then the MFMAs that consume them
        if (K % 32) {  // layer 0: K = input width (<= kMaxInputDim), possibly odd; xs columns >= K are zero
            float av[(kMaxInputDim + 1) / 2], bv[(kMaxInputDim + 1) / 2];
#pragma unroll
            for (int st = 0; st < (kMaxInputDim + 1) / 2; ++st) {
                const int k = 2 * st + h;
                av[st] = k < K ? a[2 * st * OUT + al] : 0.0f;
                bv[st] = b[k];
            }
#pragma unroll
            for (int st = 0; st < (kMaxInputDim + 1) / 2; ++st)
                if (2 * st < K) acc = mfma2(av[st], bv[st], acc);
        } else {
            // two register batches of 32 weights: the loads of batch s+1 are in flight while the MFMAs of batch s run,
            // so only a tile's first batch waits for L2.  The scheduling fences keep hipcc from hoisting both batches'
            // LDS operand reads to the top (that spilled hundreds of registers).
            float av0[32], av1[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) av0[u] = a[2 * u * OUT + al];
#pragma unroll 1
            for (int s0 = 0; s0 < K / 2; s0 += 64) {
                const bool more1 = s0 + 32 < K / 2, more2 = s0 + 64 < K / 2;
                if (more1) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) av1[u] = a[2 * (s0 + 32 + u) * OUT + al];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 32; ++u) acc = mfma2(av0[u], b[2 * (s0 + u) + h], acc);
                __builtin_amdgcn_sched_barrier(0);
                if (more2) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) av0[u] = a[2 * (s0 + 64 + u) * OUT + al];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more1) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) acc = mfma2(av1[u], b[2 * (s0 + 32 + u) + h], acc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = acc[i] > 0.0f ? acc[i] : elu_neg(acc[i]);
        drop16<MODE>(o, dm, layer, mo, h, pair, (uint32_t)pair, rows, row_base + rr);
#pragma unroll
        for (int i = 0; i < 16; ++i) hout[rr * OUT_STRIDE + 32 * mo + acc_feature(i, h)] = live ? o[i] : -0.0f;
    }
}

// dW[out][in] (+)= sum_r dz[r][out] * hin[r][in] for one 32-row tile: both operands are row reads of LDS tiles.
template <int OUT, int DZ_STRIDE, int IN_STRIDE>
__device__ __forceinline__ void weight_grad(const float* __restrict__ dz, const float* __restrict__ hin, int IN, float* __restrict__ g,
                                            bool first) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int n_in = (IN + 31) / 32;
#pragma unroll 1
    for (int tile = wave; tile < (OUT / 32) * n_in; tile += NW) {
        const int mo = tile / n_in, ni = tile % n_in;
        f32x16 acc = {0};
        const float* a = dz + 32 * mo + r;
        const float* b = hin + 32 * ni + r;
        const int in = 32 * ni + r;
        // later tiles of a row block accumulate: the running sums are fetched before the MFMA chain and added
        // after it, so their L2 round trip hides behind the chain
        float prev[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = (!first && in < IN) ? g[(int64_t)(32 * mo + acc_feature(i, h)) * IN + in] : 0.0f;
#pragma unroll
        for (int k = 0; k < TR; k += 2) acc = mfma2(a[(k + h) * DZ_STRIDE], b[(k + h) * IN_STRIDE], acc);
        if (in < IN) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g[(int64_t)(32 * mo + acc_feature(i, h)) * IN + in] = prev[i] + acc[i];
        }
    }
}

// db[o] (+)= sum_r dz[r][o]
template <int OUT, int DZ_STRIDE>
__device__ __forceinline__ void bias_grad(const float* __restrict__ dz, float* __restrict__ g, bool first) {
    for (int o = threadIdx.x; o < OUT; o += NT) {
        float a = 0.0f;
#pragma unroll
        for (int r = 0; r < TR; ++r) a += dz[r * DZ_STRIDE + o];
        g[o] = first ? a : g[o] + a;
    }
}

// dzin[r][in] = (sum_o dz[r][o] * W[o][in]) * d(hin[r][in]): A = rows of W [out][in] from global memory / L2,
// B = columns of the dz tile.
// dzin may be hin itself: every element is read (for d(.)) and then written by the same lane.
template <int OUT, int IN, int DZ_STRIDE, int IN_STRIDE>
__device__ __forceinline__ void input_grad(const float* __restrict__ dz, const float* __restrict__ W, const float* hin, float scale,
                                           float* dzin) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int unit = wave; unit < (IN / 32) * RT; unit += NW) {
        const int mi = unit / RT, rr = 32 * (unit % RT) + r;
        f32x16 acc = {0};
        const float* a = W + 32 * mi;                // uniform
        const unsigned al = (unsigned)(h * IN + r);  // this lane's element offset inside every k-step pair
        const float* b = dz + rr * DZ_STRIDE;
        // two register batches, as in dense_forward
        float av0[32], av1[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) av0[u] = a[2 * u * IN + al];
#pragma unroll 1
        for (int s0 = 0; s0 < OUT / 2; s0 += 64) {
            const bool more1 = s0 + 32 < OUT / 2, more2 = s0 + 64 < OUT / 2;
            if (more1) {
#pragma unroll
                for (int u = 0; u < 32; ++u) av1[u] = a[2 * (s0 + 32 + u) * IN + al];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = mfma2(av0[u], b[2 * (s0 + u) + h], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) {
#pragma unroll
                for (int u = 0; u < 32; ++u) av0[u] = a[2 * (s0 + 64 + u) * IN + al];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more1) {
#pragma unroll
                for (int u = 0; u < 32; ++u) acc = mfma2(av1[u], b[2 * (s0 + 32 + u) + h], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int in = 32 * mi + acc_feature(i, h);
            dzin[rr * IN_STRIDE + in] = acc[i] * dfactor(hin[rr * IN_STRIDE + in], scale);
        }
    }
}

// What the forward KERNEL carries from tile to tile so that a tile's critical path has no global load behind a store
// (a wave's load waits for the acknowledgement of its earlier stores): the output layer's weights, read once per
// workgroup, and the next tile's input rows, fetched under layer 1 of the current tile into the then-idle xs tile.
struct TileCarry {
    float w3[8];        // this thread's eight output-layer weights (thread = (row, part of 8))
    float b3;
    bool xs_ready;      // l.xs already holds this tile's rows (the previous tile prefetched them)
    int64_t next_base;  // first row of the next tile to prefetch, or -1
};

__device__ __forceinline__ float xs_elem(const float* __restrict__ xb, int64_t rows, int F, int64_t row_base, int i) {
    const int r = i / SX, k = i % SX;
    return (row_base + r < rows && k < F) ? xb[(row_base + r) * F + k] : 0.0f;
}

// forward of one tile of one pair; fills xs, h1, h2, h3, y
template <int MODE>
__device__ __forceinline__ void forward_tile(const Lds& l, const float* __restrict__ xb, int64_t rows, int F, int64_t row_base,
                                             const float* __restrict__ WT, const DropMask& dm, int pair, float* __restrict__ act_tile = nullptr,
                                             const TileCarry* carry = nullptr) {
    constexpr int XN = (TR * SX + NT - 1) / NT;
    if (!(carry && carry->xs_ready)) {
        for (int i = threadIdx.x; i < TR * SX; i += NT) l.xs[i] = xs_elem(xb, rows, F, row_base, i);
        __syncthreads();
    }
    // layer 0: k runs over the F real columns only (the next flat entry after W0^T is the bias, never read as a weight)
    dense_forward<MODE, H1, SX, S1>(l.xs, F, WT + flat_w_off(F, 0), WT + flat_b_off(F, 0), l.h1, dm, pair, rows, row_base, 0);
    __syncthreads();
    const bool prefetch = carry && carry->next_base >= 0;
    float xr[XN];
    if (prefetch) {  // the xs tile is idle from here on (forward kernel only): the loads land under layer 1
#pragma unroll
        for (int q = 0; q < XN; ++q) xr[q] = threadIdx.x + q * NT < TR * SX ? xs_elem(xb, rows, F, carry->next_base, threadIdx.x + q * NT) : 0.0f;
    }
    dense_forward<MODE, H2, S1, S2>(l.h1, H1, WT + flat_w_off(F, 1), WT + flat_b_off(F, 1), l.h2, dm, pair, rows, row_base, 1);
    if (prefetch) {
#pragma unroll
        for (int q = 0; q < XN; ++q)
            if (threadIdx.x + q * NT < TR * SX) l.xs[threadIdx.x + q * NT] = xr[q];
    }
    __syncthreads();
    dense_forward<MODE, H3, S2, S3>(l.h2, H2, WT + flat_w_off(F, 2), WT + flat_b_off(F, 2), l.h3, dm, pair, rows, row_base, 2);
    // Saving h1 and h2 for the backward kernel is the job of the waves that have no output tile in layer 2: a wave's next
    // load waits for its earlier stores to be acknowledged (one counter for both), and these waves have no next load.
    constexpr int kBusy = (H3 / 32) * RT < NW ? (H3 / 32) * RT : NW;
    if (act_tile && kBusy < NW && (int)threadIdx.x >= kBusy * 64) {
        float4* dst = reinterpret_cast<float4*>(act_tile);
        const float4* src = reinterpret_cast<const float4*>(l.h1);
        for (int i = threadIdx.x - kBusy * 64; i < kActEarlyFloats / 4; i += NT - kBusy * 64) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x < 8 * TR) {  // layer 3: 64 -> 1; 8 threads per row
        const int r = threadIdx.x >> 3, part = threadIdx.x & 7;
        const float* w3 = WT + flat_w_off(F, 3);
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = fmaf(l.h3[r * S3 + part * 8 + k], carry ? carry->w3[k] : w3[part * 8 + k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (part == 0) l.y[r] = a + (carry ? carry->b3 : WT[flat_b_off(F, 3)]);
    }
    __syncthreads();
}

// T2: train-mode forward, one workgroup per pair ---------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, kWgPerCu) void train_forward_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F,
                                                              int S, const float* __restrict__ WT, DropMask dm,
                                                              float* __restrict__ Y, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S;
    const int64_t P = flat_param_count(F);
    // rows of a large batch are split over blockIdx.y in runs of whole tiles
    const int64_t per = (((rows + gridDim.y - 1) / gridDim.y) + TR - 1) / TR * TR;
    const int64_t begin = (int64_t)blockIdx.y * per, end = begin + per < rows ? begin + per : rows;
    TileCarry carry;
    {
        const float* w3 = WT + (int64_t)pair * P + flat_w_off(F, 3);
#pragma unroll
        for (int k = 0; k < 8; ++k) carry.w3[k] = w3[(threadIdx.x & 7) * 8 + k];
        carry.b3 = WT[(int64_t)pair * P + flat_b_off(F, 3)];
        carry.xs_ready = false;
    }
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
        const int64_t tiles = (rows + TR - 1) / TR;
        float* act_tile = act ? act + ((int64_t)pair * tiles + row_base / TR) * kActFloats : nullptr;
        carry.next_base = row_base + TR < end ? row_base + TR : -1;
        forward_tile<MODE>(l, xb + (int64_t)m * x_stride, rows, F, row_base, WT + (int64_t)pair * P, dm, pair, act_tile, &carry);
        carry.xs_ready = carry.next_base >= 0;
        if (threadIdx.x < TR && row_base + threadIdx.x < rows) Y[(int64_t)pair * rows + row_base + threadIdx.x] = l.y[threadIdx.x];
        if (act) {  // keep the rest of the tile's activations (h3; forward_tile saved h1 and h2) for the backward kernel
            float4* dst = reinterpret_cast<float4*>(act_tile);
            const float4* src = reinterpret_cast<const float4*>(l.h1);
            for (int i = kActEarlyFloats / 4 + threadIdx.x; i < kActFloats / 4; i += NT) dst[i] = src[i];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void train_loss_kernel(const float* __restrict__ Y, const float* __restrict__ yb, int64_t y_stride, int S,
                                                        int64_t rows, float alpha, float beta, float* __restrict__ delta,
                                                        float* __restrict__ loss, double* __restrict__ loss_acc) {
    const int m = blockIdx.x;
    const float* ym = yb + (int64_t)m * y_stride;
    const LossCoef c = member_loss<NT>(Y, ym, m, S, rows, alpha, beta, true, loss, loss_acc);
    for (int64_t r = threadIdx.x; r < rows; r += NT) delta[(int64_t)m * rows + r] = loss_delta(c, pred_mean(Y, m, S, rows, r), ym[r]);
}

template <int MODE>
__global__ __launch_bounds__(NT, kWgPerCu) void train_backward_kernel(const float* __restrict__ xb, int64_t x_stride, int64_t rows, int F,
                                                               int S, const float* __restrict__ W, const float* __restrict__ WT, DropMask dm,
                                                               const float* __restrict__ delta, float* __restrict__ dW,
                                                               const float* __restrict__ act, LossFuse lf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ LossCoef lcs;
    const Lds l = carve(smem);
    const int pair = blockIdx.x, m = pair / S, t = threadIdx.x;
    const int64_t P = flat_param_count(F);
    const float* w = W + (int64_t)pair * P;
    // row block blockIdx.y accumulates its own partial dW[pair * row_blocks + block] (summed by train_grad_kernel)
    float* g = dW + ((int64_t)pair * gridDim.y + blockIdx.y) * P;
    const float sc = MODE != 0 ? dm.scale : 1.0f;
    const int64_t per = (((rows + gridDim.y - 1) / gridDim.y) + TR - 1) / TR * TR;
    const int64_t begin = (int64_t)blockIdx.y * per, end = begin + per < rows ? begin + per : rows;
    if (begin >= end) {  // more blocks than tiles: this partial is zero
        for (int64_t i = t; i < P; i += NT) g[i] = 0.0f;
        return;
    }
    const float* ym = lf.Y ? lf.yb + (int64_t)m * lf.y_stride : nullptr;
    if (lf.Y) {
        const LossCoef c = member_loss<NT>(lf.Y, ym, m, S, rows, lf.alpha, lf.beta, pair % S == 0 && blockIdx.y == 0, lf.loss, lf.loss_acc);
        if (t == 0) lcs = c;  // kept in LDS: the back-propagation below has no registers to spare
        __syncthreads();
    }
    for (int64_t row_base = begin; row_base < end; row_base += TR) {
        const bool first = row_base == begin;
        if (act) {  // the forward kernel saved this tile's activations: reload them (and the input tile)
            const float* xm = xb + (int64_t)m * x_stride;
            for (int i = t; i < TR * SX; i += NT) {
                const int r = i / SX, k = i % SX;
                l.xs[i] = (row_base + r < rows && k < F) ? xm[(row_base + r) * F + k] : 0.0f;
            }
            const int64_t tiles = (rows + TR - 1) / TR;
            const float4* src = reinterpret_cast<const float4*>(act + ((int64_t)pair * tiles + row_base / TR) * kActFloats);
            float4* dst = reinterpret_cast<float4*>(l.h1);
            for (int i = t; i < kActFloats / 4; i += NT) dst[i] = src[i];
            __syncthreads();
        } else {
            forward_tile<MODE>(l, xb + (int64_t)m * x_stride, rows, F, row_base, WT + (int64_t)pair * P, dm, pair);
        }
        // ---- layer 3: dw3[k] = sum_r dy[r] h3[r][k], db3 = sum_r dy[r]; dz2 = dy * w3 * d(h3)
        if (t < TR) {
            const int64_t row = row_base + t;
            float dy = 0.0f;
            if (row < rows) dy = lf.Y ? loss_delta(lcs, pred_mean(lf.Y, m, S, rows, row), ym[row]) : delta[(int64_t)m * rows + row];
            l.y[t] = dy;
        }
        __syncthreads();
        if (t < H3) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < TR; ++r) a = fmaf(l.y[r], l.h3[r * S3 + t], a);
            float* p = g + flat_w_off(F, 3) + t;
            *p = first ? a : *p + a;
        } else if (t == H3) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < TR; ++r) a += l.y[r];
            float* p = g + flat_b_off(F, 3);
            *p = first ? a : *p + a;
        }
        __syncthreads();  // dw3 has read h3: dz2 may now replace it
        for (int i = t; i < TR * H3; i += NT) {
            const int r = i / H3, k = i % H3;
            l.h3[r * S3 + k] = l.y[r] * w[flat_w_off(F, 3) + k] * dfactor(l.h3[r * S3 + k], sc);
        }
        __syncthreads();
        // ---- layer 2: dW2 = dz2^T h2 (64 x 128), db2; then dz1 = (dz2 W2) * d(h2) replaces h2
        weight_grad<H3, S3, S2>(l.h3, l.h2, H2, g + flat_w_off(F, 2), first);
        bias_grad<H3, S3>(l.h3, g + flat_b_off(F, 2), first);
        __syncthreads();
        input_grad<H3, H2, S3, S2>(l.h3, w + flat_w_off(F, 2), l.h2, sc, l.h2);
        __syncthreads();
        // ---- layer 1: dW1 = dz1^T h1 (128 x 256), db1; then dz0 = (dz1 W1) * d(h1) replaces h1
        weight_grad<H2, S2, S1>(l.h2, l.h1, H1, g + flat_w_off(F, 1), first);
        bias_grad<H2, S2>(l.h2, g + flat_b_off(F, 1), first);
        __syncthreads();
        input_grad<H2, H1, S2, S1>(l.h2, w + flat_w_off(F, 1), l.h1, sc, l.h1);
        __syncthreads();
        // ---- layer 0: dW0 = dz0^T x (256 x F), db0
        weight_grad<H1, S1, SX>(l.h1, l.xs, F, g + flat_w_off(F, 0), first);
        bias_grad<H1, S1>(l.h1, g + flat_b_off(F, 0), first);
        __syncthreads();
    }
}

