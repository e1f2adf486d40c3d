// Dev tool: the rate a bare bf16 MFMA stream sustains on this chip as a function of the DATA it chews on and of the MFMA SHAPE.
// The clock an MI355X holds under an MFMA-dense kernel is power-managed, and the power of the matrix pipe depends on how many
// operand bits toggle: a loop over constant operands (tools/mfma_valu_mix.hip) therefore overstates the ceiling of a kernel that
// multiplies real activations, and ranks the two bf16 shapes by cycles only (MI355X_MICROARCH.md, DVFS give-back item 7).
// Shapes: v_mfma_f32_32x32x16_bf16 (4 accumulator tiles of 32x32 per wave) and v_mfma_f32_16x16x32_bf16 (16 tiles of 16x16):
// the same 64 accumulator registers, the same MACs per loop iteration, the same bytes read from LDS per MAC.
// Modes: 0 = the same two constant fragments every time, 1 = eight random fragments per operand rotated every instruction
// (every operand bit toggles like data), 2 = mode 1 with the A fragment read from LDS (ds_read_b128) once per 32,768 MACs, as
// forward_bf16_fast_kernel does (two 32-row tiles, or four 16-row tiles, per A fragment), 3 = mode 2 plus the exponent-field
// ELU's VALU instructions at the kernel's ratio (672 per 92 32x32x16-MFMAs = 7.3 per 16,384 MACs; here 7) on registers of their own.
// Every workgroup stamps s_memtime / s_memrealtime around its loop into a buffer nothing reads: the in-kernel clock
// (give-back item 6) is the median over workgroups of d(memtime) / d(memrealtime) x 100 MHz, in the LAST launch of the run.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_data_ceiling.hip -o /tmp/mfma_data_ceiling && /tmp/mfma_data_ceiling [launches timed back to back] [sweep]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

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

// the six instructions forward_bf16.hip's elu_field2 issues per two elements
__device__ __forceinline__ unsigned elu_field2(float u0, float u1) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float x0 = fmaf(u0, 128.0f / 65535.0f, 0.2465f), x1 = fmaf(u1, 128.0f / 65535.0f, 0.2465f);
    const u16x2 y = __builtin_bit_cast(u16x2, __builtin_amdgcn_cvt_pknorm_u16(x0, x1));
    const u16x2 e = __builtin_elementwise_min(y, u16x2{0x3FB9, 0x3FB9});
    const i16x2 p = __builtin_bit_cast(i16x2, __builtin_convertvector(f32x2{u0, u1}, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(p, __builtin_bit_cast(i16x2, e)));
}

struct Stamp {
    unsigned long long cycles, ticks;  // d(s_memtime), d(s_memrealtime: 100 MHz)
};

template <int SHAPE, int MODE, int NP = 16>  // SHAPE 32: 32x32x16, 16: 16x16x32; NP: element pairs of VALU mix per 8 steps (mode 3)
__global__ __launch_bounds__(512) void stream_kernel(float* out, int iters, Stamp* stamps) {
    __shared__ bf16x8 lds_a[8 * 64];
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
    float v[6];  // the VALU mix's own data: like accumulators, a few units around zero
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = 4.0f * (float)a[i][0];
    unsigned sink = 0;
    __syncthreads();
    f32x16 acc32[4];
    f32x4 acc16[16];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc32[m] = f32x16{0};
#pragma unroll
    for (int m = 0; m < 16; ++m) acc16[m] = f32x4{0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) alone: the stamps are back before the loop's own LDS waits
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ra = MODE == 0 ? 0 : r, rb = MODE == 0 ? 0 : (r * 3 + 1) & 7;
            bf16x8 af = a[ra];
            if (MODE >= 2) af = lds_a[ra * 64 + lane];  // one 16-byte LDS read per 32,768 MACs
            if (SHAPE == 32) {
                acc32[(2 * r) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[rb], acc32[(2 * r) & 3], 0, 0, 0);
                acc32[(2 * r + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[(rb + 4) & 7], acc32[(2 * r + 1) & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc16[(4 * r + t) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[(rb + 2 * t) & 7], acc16[(4 * r + t) & 15], 0, 0, 0);
            }
            if (MODE == 3) {
                // the kernel's 14.6 VALU instructions per 32,768 MACs: NP = 16 pairs of elements per 8 steps, 6 instructions each
                // + one xor that keeps the result alive = 14 per step; the empty asm statement makes the inputs opaque at no cost
#pragma unroll
                for (int e = 0; e < (NP * (r + 1)) / 8 - (NP * r) / 8; ++e) {
                    const int i = (2 * r + e) % 3;
                    asm volatile("" : "+v"(v[2 * i]), "+v"(v[2 * i + 1]));
                    sink ^= elu_field2(v[2 * i], v[2 * i + 1]);
                }
            }
        }
        if (MODE >= 2) asm volatile("" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = __uint_as_float(sink & 0x007FFFFFu);
    if (SHAPE == 32) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc32[m][i];
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += acc16[m][i];
    }
    out[tid] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

template <int SHAPE, int MODE, int NP = 16>
void run(int waves_per_simd, int iters, int reps) {
    const int blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    Stamp* stamps;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&stamps, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) stream_kernel<SHAPE, MODE, NP><<<blocks, threads>>>(out, iters, stamps);  // let the clock settle
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) stream_kernel<SHAPE, MODE, NP><<<blocks, threads>>>(out, iters, stamps);  // back to back, like the predict's launches
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<Stamp> h(blocks);
    hipMemcpy(h.data(), stamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (const Stamp& s : h)
        if (s.ticks) mhz.push_back(100.0 * (double)s.cycles / (double)s.ticks);
    std::sort(mhz.begin(), mhz.end());
    const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    // per SIMD and iteration: 16 MFMAs' worth of 16,384 MACs in either shape
    const double per_simd = (double)iters * 16 * waves_per_simd;
    const double cyc = h[blocks / 2].cycles / ((double)iters * 16 * waves_per_simd);  // cycles per 16,384 MACs per SIMD
    const char* names[] = {"constant operands", "random operands, rotated", "random, A from LDS per 32k MACs", "random, A from LDS, + field-ELU mix"};
    char label[96];
    snprintf(label, sizeof label, MODE == 3 ? "%s (%.2f VALU per 16k MACs)" : "%s", names[MODE], NP * 7 / 16.0);
    printf("%-9s %-62s waves/SIMD=%d  %6.1f ms  %.2f ns per 16k MACs per SIMD  %.0f TFLOP/s (%.3f of 2.5 PF)  in-kernel clock %.0f MHz  %.1f cycles per 16k MACs\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", label, waves_per_simd, ms, ms * 1e6 / per_simd,
           per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12, per_simd * 1024 * 32768.0 / (ms * 1e-3) / 2.5e15, clk, cyc);
    fflush(stdout);
    hipFree(out);
    hipFree(stamps);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 1;  // launches timed back to back (the clock keeps sagging for ~1 s under load)
    if (argc > 2) {  // sweep of the VALU share at two waves per SIMD: what one VALU instruction per MFMA costs either shape
        const int iters = 100000;
        run<32, 3, 0>(2, iters, reps);
        run<32, 3, 4>(2, iters, reps);
        run<32, 3, 8>(2, iters, reps);
        run<32, 3, 12>(2, iters, reps);
        run<32, 3, 14>(2, iters, reps);
        run<32, 3, 16>(2, iters, reps);
        run<32, 3, 20>(2, iters, reps);
        run<16, 3, 0>(2, iters, reps);
        run<16, 3, 4>(2, iters, reps);
        run<16, 3, 8>(2, iters, reps);
        run<16, 3, 12>(2, iters, reps);
        run<16, 3, 14>(2, iters, reps);
        run<16, 3, 16>(2, iters, reps);
        run<16, 3, 20>(2, iters, reps);
        return 0;
    }
    for (int w : {2, 1}) {
        const int iters = 200000 / w;  // ~50-60 ms per launch
        run<32, 0>(w, iters, reps);
        run<16, 0>(w, iters, reps);
        run<32, 1>(w, iters, reps);
        run<16, 1>(w, iters, reps);
        run<32, 2>(w, iters, reps);
        run<16, 2>(w, iters, reps);
        run<32, 3>(w, iters, reps);
        run<16, 3>(w, iters, reps);
    }
    return 0;
}
