// Dev tool: the rate a bare v_mfma_f32_32x32x16_bf16 stream sustains on this chip as a function of the DATA it chews on.
// The clock an MI355X holds under an MFMA-dense kernel is power-managed, and the power of the matrix pipe depends on how many
// operand bits toggle: a loop over constant operands (tools/mfma_valu_mix.hip) therefore overstates the ceiling of a kernel that
// multiplies real activations.  Modes: 0 = the same two constant fragments every time, 1 = eight random fragments per operand
// rotated every instruction (every operand bit toggles like data), 2 = mode 1 with the A fragment read from LDS (ds_read_b128)
// once per two MFMAs, as forward_bf16_fast_kernel does.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_data_ceiling.hip -o /tmp/mfma_data_ceiling && /tmp/mfma_data_ceiling [launches timed back to back]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ bf16x8 random_fragment(unsigned key) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned h = mix32(key * 8u + j);
        f[j] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));  // uniform in [-1, 1): sign, exponent and mantissa all vary
    }
    return f;
}

template <int MODE>
__global__ __launch_bounds__(512) void stream_kernel(float* out, int iters) {
    __shared__ bf16x8 lds_a[8 * 64];
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x16{0};
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = random_fragment(MODE == 0 ? 1u : tid * 16u + r);
        b[r] = random_fragment(MODE == 0 ? 2u : tid * 16u + 8u + r);
    }
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64)
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_a[r * 64 + lane] = a[r];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ra = MODE == 0 ? 0 : r, rb = MODE == 0 ? 0 : (r * 3 + 1) & 7;
            bf16x8 af = a[ra];
            if (MODE == 2) af = lds_a[ra * 64 + lane];  // one 16-byte LDS read feeds two MFMAs (two row tiles)
            acc[(2 * r) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[rb], acc[(2 * r) & 3], 0, 0, 0);
            acc[(2 * r + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[(rb + 4) & 7], acc[(2 * r + 1) & 3], 0, 0, 0);
        }
        if (MODE == 2) asm volatile("" ::: "memory");
    }
    float s = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[m][i];
    out[tid] = s;
}

template <int MODE>
void run(int waves_per_simd, int iters, int reps) {
    const int blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) stream_kernel<MODE><<<blocks, threads>>>(out, iters);  // let the clock settle
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) stream_kernel<MODE><<<blocks, threads>>>(out, iters);  // back to back, like the predict's launches
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double per_simd = (double)iters * 16 * waves_per_simd;
    const char* names[] = {"constant operands", "random operands, rotated", "random operands, A from LDS per 2 MFMAs"};
    printf("%-42s waves/SIMD=%d  %6.1f ms  %.2f ns per MFMA per SIMD  %.0f TFLOP/s (%.3f of 2.5 PF)\n", names[MODE], waves_per_simd, ms,
           ms * 1e6 / per_simd, per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12, per_simd * 1024 * 32768.0 / (ms * 1e-3) / 2.5e15);
    fflush(stdout);
    hipFree(out);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 1;  // launches timed back to back (the clock keeps sagging for ~1 s under load)
    for (int w : {2, 1}) {
        const int iters = 200000 / w;  // ~50-60 ms per launch
        run<0>(w, iters, reps);
        run<1>(w, iters, reps);
        run<2>(w, iters, reps);
    }
    return 0;
}
