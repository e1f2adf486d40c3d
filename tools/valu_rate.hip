// Dev tool: VALU issue cost per instruction kind on gfx950 at W waves per SIMD (s_memtime cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x * 1e-3f + i * 0.01f - 0.5f;
    float c = 1.4426950408889634f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(x[i]));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (KIND == 2) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 5) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (KIND == 6) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 7) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(x[i]));
            if (KIND == 8) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(double*)&x[i & ~1]) : "v"(*(double*)&x[(i & ~1)]));
            if (KIND == 10) asm volatile("v_fmamk_f32 %0, %0, 0x3fb8aa3b, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 11) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[i]));
            if (KIND == 12) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(c));
            if (KIND == 13) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 14) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 15) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 16) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c));
            if (KIND == 17) asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(x[i]));
            if (KIND == 18) asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(x[i]) : "v"(c));
            if (KIND == 19) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 20) asm volatile("v_pk_mul_f32 %0, %1, %1" : "+v"(*(double*)&x[i & ~1]) : "v"(*(double*)&x[(i & ~1)]));
            if (KIND == 21) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 5) & 15]));
            if (KIND == 22) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 23) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 24) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(x[i]));
            if (KIND == 25) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 26) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 27) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 28) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 29) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x[i]));
            if (KIND == 30) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 31) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    const int iters = 4000;
    const int threads = waves_per_simd > 4 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);  // fill every CU evenly
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    rate_kernel<KIND><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(a, 0);
    rate_kernel<KIND><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static unsigned long long h[4096];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[i];
    avg /= blocks;
    // per SIMD: waves_per_simd waves each issue iters*16 instructions
    const double per_simd = (double)iters * 16 * waves_per_simd;
    printf("%-22s waves/SIMD=%d  %.2f memtime ticks, %.2f ns per wave-instruction on the SIMD (=%.2f cycles at 2.4 GHz)\n", name,
           waves_per_simd, avg / per_simd, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {4}) {
        run<24>("v_exp_legacy_f32", w);
        run<25>("v_maximum3_f32", w);
        run<26>("v_max_i32", w);
        run<27>("v_max3_f32", w);
        run<28>("v_ldexp_f32", w);
        run<29>("v_rcp_f32", w);
        run<30>("v_perm_b32", w);
        run<31>("v_pk_maximum3_f16", w);
        run<0>("v_exp_f32 clamp", w);
        run<11>("v_exp_f32 e32", w);
        run<1>("v_fma_f32", w);
        run<10>("v_fmamk_f32", w);
        run<2>("v_max_f32", w);
        run<3>("v_med3_f32", w);
        run<4>("v_cvt_pk_bf16_f32", w);
        run<5>("v_pk_fma_f16", w);
        run<6>("v_pk_max_f16", w);
        run<7>("v_exp_f16", w);
        run<8>("v_cvt_pkrtz_f16_f32", w);
        run<9>("v_pk_fma_f32", w);
        run<12>("v_mul_f32 clamp", w);
        run<13>("v_add_f32", w);
        run<14>("v_mul_f32", w);
        run<15>("v_min_f32", w);
        run<16>("v_cndmask_b32", w);
        run<17>("v_max_f32 0,x", w);
        run<18>("v_fma_f32 clamp", w);
        run<19>("v_and_b32", w);
        run<20>("v_pk_mul_f32", w);
        run<21>("v_max_f32 x,y", w);
        run<22>("v_pk_max_i16", w);
        run<23>("v_mov_b32", w);
    }
    return 0;
}
