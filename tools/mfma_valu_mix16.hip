// Dev tool: VALU issue time beside v_mfma_f32_16x16x32_bf16 (half the FLOPs per instruction of the 32x32x16 form),
// reported per 32,768 FLOP so the numbers compare with tools/mfma_valu_mix.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND, int SHAPE>
__global__ void mix_kernel(float* out, int iters) {
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x16 d0 = {0}, d1 = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(0.5f - j * 0.01f); }
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 0.01f - 0.5f;
    const float c = 1.4426950408889634f;
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 16) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {  // 8 x 16x16x32 = 4 x 32x32x16 FLOPs
                if ((m & 3) == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                if ((m & 3) == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                if ((m & 3) == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                if ((m & 3) == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K / 2; ++k) {
                    const int i = (m * (K / 2) + k) & 15;
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                    if (KIND == 1) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(x[i]));
                    if (KIND == 2) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
                else d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int i = (m * K + k) & 15;
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                    if (KIND == 1) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(x[i]));
                    if (KIND == 2) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + d0[i] + d1[i];
    s += c0[0] + c1[1] + c2[2] + c3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND, int SHAPE>
void run(int waves_per_simd) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    mix_kernel<K, KIND, SHAPE><<<blocks, threads>>>(out, iters);
    hipEventRecord(a, 0);
    mix_kernel<K, KIND, SHAPE><<<blocks, threads>>>(out, iters);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double slots = (double)iters * 4 * waves_per_simd;  // 32,768-FLOP slots per SIMD
    const char* kn = KIND == 0 ? "v_fma" : KIND == 1 ? "v_exp" : "v_med3";
    printf("shape=%dx%d waves/SIMD=%d  %2d x %-6s per 32768 FLOP: %.1f ns per slot, MFMA rate %.0f TFLOP/s\n", SHAPE, SHAPE, waves_per_simd, K, kn,
           ms * 1e6 / slots, slots * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

template <int SHAPE>
void all(int w) {
    run<0, 0, SHAPE>(w); run<4, 0, SHAPE>(w); run<8, 0, SHAPE>(w); run<12, 0, SHAPE>(w);
    run<2, 1, SHAPE>(w); run<4, 1, SHAPE>(w);
    run<4, 2, SHAPE>(w); run<8, 2, SHAPE>(w);
}

int main() {
    for (int w : {2, 4}) { all<32>(w); all<16>(w); }
    return 0;
}
