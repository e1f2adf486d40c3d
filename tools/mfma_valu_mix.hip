// Dev tool: how much VALU issue time fits beside v_mfma_f32_32x32x16_bf16 at W waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND, int ACC = 0>
__global__ void mix_kernel(float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(0.5f - j * 0.01f); }
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 0.01f - 0.5f;
    const float c = 1.4426950408889634f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (ACC == 1) {  // accumulators pinned to AccVGPRs
                if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            } else if (ACC == 2) {  // accumulators pinned to ArchVGPRs (asm, for a like-for-like comparison)
                if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            } else if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int i = (m * K + k) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(x[i]));
                if (KIND == 2) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND, int ACC = 0>
void run(int waves_per_simd) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    mix_kernel<K, KIND, ACC><<<blocks, threads>>>(out, iters);
    hipEventRecord(a, 0);
    mix_kernel<K, KIND, ACC><<<blocks, threads>>>(out, iters);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)iters * 4 * waves_per_simd;  // per SIMD
    const char* kn = KIND == 0 ? "v_fma" : KIND == 1 ? "v_exp" : "v_med3";
    printf("acc=%s waves/SIMD=%d  %2d x %-6s per MFMA: %.1f ns per MFMA slot (=%.1f cycles at 2.4 GHz), MFMA rate %.0f TFLOP/s\n", ACC == 1 ? "agpr" : ACC == 2 ? "vgpr(asm)" : "vgpr", waves_per_simd, K, kn,
           ms * 1e6 / mfmas, ms * 1e6 / mfmas * 2.4, mfmas * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0, 0>(w); run<2, 0>(w); run<4, 0>(w); run<6, 0>(w); run<8, 0>(w); run<12, 0>(w);
        run<1, 1>(w); run<2, 1>(w); run<4, 1>(w);
        run<2, 2>(w); run<4, 2>(w); run<8, 2>(w);
    }
    for (int w : {4}) {
        run<0, 0, 2>(w); run<4, 0, 2>(w); run<8, 0, 2>(w); run<12, 0, 2>(w);
        run<0, 0, 1>(w); run<2, 0, 1>(w); run<4, 0, 1>(w); run<6, 0, 1>(w); run<8, 0, 1>(w); run<12, 0, 1>(w);
        run<1, 1, 1>(w); run<2, 1, 1>(w); run<4, 1, 1>(w);
        run<2, 2, 1>(w); run<4, 2, 1>(w); run<8, 2, 1>(w);
    }
    return 0;
}
