// fp32 fused forward of BayesianNetwork for (member,sample) pairs - the parity path.
//
// Restates /root/reference/models/bayesian_ensemble.py:73-92 (3 x {linear, ELU, dropout}, then
// linear 64->1) on pre-sampled effective weights (sample_weights.hip).  One workgroup = one pair
// x 256 rows (8 waves, two per SIMD so one wave's ELU epilogue hides behind the other's MFMA chain
// and the staged weights serve twice the rows); one wave = 32 rows.  The whole chain stays in registers:
//     Z^T[out, batch] = W[out, k] . H^T[k, batch]      (v_mfma_f32_32x32x2_f32, exact fp32 FMAs)
// leaves feature f(i,h) of each 32-feature tile in accumulator register i, batch row on the
// lane - exactly the B operand of the next layer's MFMA, whose weights sample_weights.hip stored
// with the matching k permutation.  Weights are staged through LDS in two phases
// (L0+L1+b1 = 141 KB, then L2+b2+w3+b3 = 33 KB) because fp32 weights (177 KB) exceed the CU's
// 160 KB.  Dropout keep-masks (train mode) are applied in the activation epilogue.
#include "common.hpp"
#include "dropmask.hpp"

namespace mcmc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

__device__ __forceinline__ void stage(float* lds, const float* __restrict__ src, int64_t n_floats) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(lds);
    for (int64_t i = threadIdx.x; i < n_floats / 4; i += blockDim.x) d4[i] = s4[i];
}

constexpr int kWavesF32 = 8;                 // waves per workgroup
constexpr int kRowsF32 = 32 * kWavesF32;    // rows per workgroup

template <int MODE>
__global__ __launch_bounds__(64 * kWavesF32, 1) void forward_f32_kernel(const float* __restrict__ xpack, int64_t n_rows, int F,
                                                             const float* __restrict__ wpack, DropMask dm,
                                                             float* __restrict__ Y, int64_t ldY) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WPackF32 L(F);
    const int ks0 = L.K0 / 2;
    const int c = blockIdx.y;
    const float* wp = wpack + (int64_t)c * L.total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * kRowsF32 + 32 * wave;
    const int64_t row = row0 + r;
    const bool active = row0 < rows_pad(n_rows);  // wave-uniform
    const bool real_row = row < n_rows;
    const uint32_t g = MODE == 2 ? drop_pair_id(dm, c) : 0u;

    stage(lds, wp, L.l2);  // L0, L1, B1
    __syncthreads();

    float h1[8][16];
    float h2[4][16];
    if (active) {
        // ---- layer 0: K0/2 k-steps, bias rides on the 1.0 column of xpack ----
        float xv[kMaxInputDim / 2 + 1];
#pragma unroll
        for (int i = 0; i < kMaxInputDim / 2 + 1; ++i) xv[i] = i < ks0 ? xpack[row * L.K0 + 2 * i + h] : 0.0f;
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) {
            f32x16 acc = {0};
#pragma unroll
            for (int i = 0; i < kMaxInputDim / 2 + 1; ++i)
                if (i < ks0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L.l0 + (mo * ks0 + i) * 64 + lane], xv[i], acc, 0, 0, 0);
            float o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = elu1(acc[i]);
            drop16<MODE>(o, dm, 0, mo, h, c, g, n_rows, row);
#pragma unroll
            for (int i = 0; i < 16; ++i) h1[mo][i] = o[i];
        }
        // ---- layer 1: 128 k-steps per 32-feature output tile ----
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
            f32x16 acc;
            const float* bias = lds + L.b1 + (mo * 2 + h) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = bias[i];
            const float* a = lds + L.l1 + (int64_t)mo * 128 * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(mt * 16 + i) * 64], h1[mt][i], acc, 0, 0, 0);
            float o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = elu1(acc[i]);
            drop16<MODE>(o, dm, 1, mo, h, c, g, n_rows, row);
#pragma unroll
            for (int i = 0; i < 16; ++i) h2[mo][i] = o[i];
        }
    }
    __syncthreads();
    stage(lds, wp + L.l2, L.total - L.l2);  // L2, B2, W3, B3 now at offset 0
    __syncthreads();
    if (!active) return;

    const int64_t o_b2 = L.b2 - L.l2, o_w3 = L.w3 - L.l2, o_b3 = L.b3 - L.l2;
    float y = 0.0f;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo) {
        f32x16 acc;
        const float* bias = lds + o_b2 + (mo * 2 + h) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias[i];
        const float* a = lds + (int64_t)mo * 64 * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(mt * 16 + i) * 64], h2[mt][i], acc, 0, 0, 0);
        const float* w3 = lds + o_w3 + (mo * 2 + h) * 16;
        float o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = elu1(acc[i]);
        drop16<MODE>(o, dm, 2, mo, h, c, g, n_rows, row);
#pragma unroll
        for (int i = 0; i < 16; ++i) y = fmaf(w3[i], o[i], y);
    }
    y += __shfl_xor(y, 32);
    y += lds[o_b3];
    if (h == 0 && real_row) Y[(int64_t)c * ldY + row] = y;
}

}  // namespace

int launch_forward_f32(const float* xpack, int64_t n_rows, int F, const float* wpack, int pair_count, const DropMask* dmp, float* Y,
                       int64_t ldY, hipStream_t s) {
    const WPackF32 L(F);
    const size_t lds_bytes = (size_t)L.l2 * sizeof(float);
    const DropMask dm = dmp ? *dmp : DropMask{};
    auto kern = dm.mode == 1 ? forward_f32_kernel<1> : dm.mode == 2 ? forward_f32_kernel<2> : forward_f32_kernel<0>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_bytes, "forward_f32")) return rc;
    const int64_t row_blocks = (rows_pad(n_rows) + kRowsF32 - 1) / kRowsF32;
    for (int done = 0; done < pair_count;) {
        const int n = pair_count - done > 32768 ? 32768 : pair_count - done;
        DropMask d = dm;  // this slab's pairs: host masks and the pair ids advance with it
        if (dm.mode == 1) {
            d.m[0] += (int64_t)done * n_rows * H1;
            d.m[1] += (int64_t)done * n_rows * H2;
            d.m[2] += (int64_t)done * n_rows * H3;
        }
        d.pair_begin += done;
        hipLaunchKernelGGL(kern, dim3((unsigned)row_blocks, n), dim3(64 * kWavesF32), lds_bytes, s, xpack, n_rows, F,
                           wpack + (int64_t)done * L.total, d, Y + (int64_t)done * ldY, ldY);
        done += n;
    }
    return check_launch("forward_f32");
}

}  // namespace mcmc
