// Dev tool: what ONE VALU instruction of a given kind costs beside an LDS-fed v_mfma_f32_32x32x16_bf16 stream on random data, two
// waves per SIMD - the regime of forward_bf16_fast_kernel (tools/mfma_data_ceiling.hip prices its whole ELU mix; this prices the
// parts).  Per 32,768 MACs (two MFMAs sharing one A fragment from LDS) the loop issues K independent instructions of one kind
// (inline asm on registers of their own, rotating over four destinations), or K as one dependent chain, or whole ELU sequences
// written sequentially / stage-wise.  Prints ns and in-kernel cycles per MFMA and the difference to the bare stream per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_beside_mfma.hip -o build/valu_beside_mfma && build/valu_beside_mfma [launches back to back]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bf16x8 random_fragment(unsigned key) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)(((int)(mix32(key * 8u + j) & 0xffff) - 32768) * (1.0f / 32768.0f));
    return f;
}

struct Stamp {
    unsigned long long cycles, ticks;
};

enum Kind { NONE, FMAMK, FMA, CVT_PKNORM_U16, CVT_PKNORM_I16, CVT_PK_BF16, PK_MIN_U16, PK_MAX_I16, PK_ADD_I16, MOV, PERM, MED3, EXP,
            CHAIN6, SEQ6, STAGED6, SEQ5 };

// K = 4 * REPT independent instructions of kind KD in ONE asm block (hipcc pads every asm statement with s_nops, so one statement
// per step): d[j] = op(a[j], b[j]), four destinations in rotation
#define VBM_BLOCK(INSTR)                                                                                                  \
    asm volatile(".rept %c12\n\t" INSTR(0, 4, 8) "\n\t" INSTR(1, 5, 9) "\n\t" INSTR(2, 6, 10) "\n\t" INSTR(3, 7, 11) "\n\t.endr"      \
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])                                                     \
                 : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "n"(REPT))
#define I_FMAMK(D, A, B) "v_fmamk_f32 %" #D ", %" #A ", 0x3b000080, %" #B
#define I_FMA(D, A, B) "v_fma_f32 %" #D ", %" #A ", %" #B ", %" #B
#define I_PKNORM_U(D, A, B) "v_cvt_pknorm_u16_f32 %" #D ", %" #A ", %" #B
#define I_PKNORM_I(D, A, B) "v_cvt_pknorm_i16_f32 %" #D ", %" #A ", %" #B
#define I_PK_BF16(D, A, B) "v_cvt_pk_bf16_f32 %" #D ", %" #A ", %" #B
#define I_PK_MIN(D, A, B) "v_pk_min_u16 %" #D ", %" #A ", %" #B
#define I_PK_MAX(D, A, B) "v_pk_max_i16 %" #D ", %" #A ", %" #B
#define I_PK_ADD(D, A, B) "v_pk_add_i16 %" #D ", %" #A ", %" #B " clamp"
#define I_MOV(D, A, B) "v_mov_b32 %" #D ", %" #A
#define I_PERM(D, A, B) "v_perm_b32 %" #D ", %" #A ", %" #B ", %" #B
#define I_MED3(D, A, B) "v_med3_f32 %" #D ", %" #A ", %" #B ", %" #B
#define I_EXP(D, A, B) "v_exp_f32 %" #D ", %" #A
#define I_CHAIN(D, A, B) "v_fma_f32 %0, %0, %" #A ", %" #B

template <int KD, int REPT>
__device__ __forceinline__ void block(unsigned (&d)[4], const unsigned (&u)[8]) {
    if (KD == FMAMK) VBM_BLOCK(I_FMAMK);
    if (KD == FMA) VBM_BLOCK(I_FMA);
    if (KD == CVT_PKNORM_U16) VBM_BLOCK(I_PKNORM_U);
    if (KD == CVT_PKNORM_I16) VBM_BLOCK(I_PKNORM_I);
    if (KD == CVT_PK_BF16) VBM_BLOCK(I_PK_BF16);
    if (KD == PK_MIN_U16) VBM_BLOCK(I_PK_MIN);
    if (KD == PK_MAX_I16) VBM_BLOCK(I_PK_MAX);
    if (KD == PK_ADD_I16) VBM_BLOCK(I_PK_ADD);
    if (KD == MOV) VBM_BLOCK(I_MOV);
    if (KD == PERM) VBM_BLOCK(I_PERM);
    if (KD == MED3) VBM_BLOCK(I_MED3);
    if (KD == EXP) VBM_BLOCK(I_EXP);
    if (KD == CHAIN6) VBM_BLOCK(I_CHAIN);  // every instruction reads the one before it
}

// the kernel's six-instruction field ELU of one element pair (u0, u1), as one asm block (sequential = one dependent chain)
__device__ __forceinline__ void elu6(unsigned& w, unsigned u0, unsigned u1, unsigned k0, unsigned cap) {
    unsigned x0, x1, p;
    asm volatile(
        "v_fmamk_f32 %0, %4, 0x3b000080, %6\n\t"
        "v_fmamk_f32 %1, %5, 0x3b000080, %6\n\t"
        "v_cvt_pknorm_u16_f32 %0, %0, %1\n\t"
        "v_cvt_pk_bf16_f32 %2, %4, %5\n\t"
        "v_pk_min_u16 %0, %0, %7\n\t"
        "v_pk_max_i16 %3, %2, %0"
        : "=&v"(x0), "=&v"(x1), "=&v"(p), "=v"(w)
        : "v"(u0), "v"(u1), "v"(k0), "v"(cap));
}
// four pairs stage by stage: no instruction depends on one of the three before it
__device__ __forceinline__ void elu6x4(unsigned (&w)[4], const unsigned (&u)[8], unsigned k0, unsigned cap) {
    unsigned x[8], p[4];
    asm volatile(
        "v_fmamk_f32 %0, %16, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %1, %17, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %2, %18, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %3, %19, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %4, %20, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %5, %21, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %6, %22, 0x3b000080, %24\n\t"
        "v_fmamk_f32 %7, %23, 0x3b000080, %24\n\t"
        "v_cvt_pk_bf16_f32 %8, %16, %17\n\t"
        "v_cvt_pk_bf16_f32 %9, %18, %19\n\t"
        "v_cvt_pk_bf16_f32 %10, %20, %21\n\t"
        "v_cvt_pk_bf16_f32 %11, %22, %23\n\t"
        "v_cvt_pknorm_u16_f32 %0, %0, %1\n\t"
        "v_cvt_pknorm_u16_f32 %2, %2, %3\n\t"
        "v_cvt_pknorm_u16_f32 %4, %4, %5\n\t"
        "v_cvt_pknorm_u16_f32 %6, %6, %7\n\t"
        "v_pk_min_u16 %0, %0, %25\n\t"
        "v_pk_min_u16 %2, %2, %25\n\t"
        "v_pk_min_u16 %4, %4, %25\n\t"
        "v_pk_min_u16 %6, %6, %25\n\t"
        "v_pk_max_i16 %12, %8, %0\n\t"
        "v_pk_max_i16 %13, %9, %2\n\t"
        "v_pk_max_i16 %14, %10, %4\n\t"
        "v_pk_max_i16 %15, %11, %6"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(p[0]), "=&v"(p[1]),
          "=&v"(p[2]), "=&v"(p[3]), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "v"(k0), "v"(cap));
}
// the five-instruction form (accumulators pre-scaled so that no v_fma is needed; round 2 measured no gain from it in the kernel)
__device__ __forceinline__ void elu5(unsigned& w, unsigned u0, unsigned u1, unsigned off, unsigned cap) {
    unsigned x, p;
    asm volatile(
        "v_cvt_pknorm_i16_f32 %0, %3, %4\n\t"
        "v_cvt_pk_bf16_f32 %1, %3, %4\n\t"
        "v_pk_add_i16 %0, %0, %5 clamp\n\t"
        "v_pk_min_i16 %0, %0, %6\n\t"
        "v_pk_max_i16 %2, %1, %0"
        : "=&v"(x), "=&v"(p), "=v"(w)
        : "v"(u0), "v"(u1), "v"(off), "v"(cap));
}

template <int KD, int K>  // K instructions (or K/6, K/5 sequences) per 32,768 MACs
__global__ __launch_bounds__(512) void stream_kernel(float* out, int iters, Stamp* stamps) {
    __shared__ bf16x8 lds_a[8 * 64];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = random_fragment(tid * 16u + r);
        b[r] = random_fragment(tid * 16u + 8u + r);
    }
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64)
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_a[r * 64 + lane] = a[r];
    unsigned u[8], d[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(4.0f * (float)a[i][0]);
    unsigned k0 = __float_as_uint(0.2465f), cap = 0x3FB93FB9u;
    asm volatile("" : "+v"(k0), "+v"(cap));
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x16{0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rb = (r * 3 + 1) & 7;
            const bf16x8 af = lds_a[r * 64 + lane];
            acc[(2 * r) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[rb], acc[(2 * r) & 3], 0, 0, 0);
            acc[(2 * r + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[(rb + 4) & 7], acc[(2 * r + 1) & 3], 0, 0, 0);
            if (KD == SEQ6) {
#pragma unroll
                for (int e = 0; e < K / 6; ++e) elu6(d[e & 3], u[(2 * e) & 7], u[(2 * e + 1) & 7], k0, cap);
            } else if (KD == STAGED6) {
                if (r & 1) elu6x4(d, u, k0, cap);  // 24 instructions per two steps = 12 per step
            } else if (KD == SEQ5) {
#pragma unroll
                for (int e = 0; e < K / 5; ++e) elu5(d[e & 3], u[(2 * e) & 7], u[(2 * e + 1) & 7], k0, cap);
            } else if (KD != NONE) {
                block<KD, K / 4>(d, u);
            }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = __uint_as_float((d[0] ^ d[1] ^ d[2] ^ d[3]) & 0x007FFFFFu);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[m][i];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0};
}

static double g_base_ns = 0.0;

template <int KD, int K>
void run(const char* name, int reps) {
    const int blocks = 256, threads = 512, iters = 100000;
    float* out;
    Stamp* stamps;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&stamps, sizeof(Stamp) * blocks * 8);
    hipMemset(stamps, 0, sizeof(Stamp) * blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) stream_kernel<KD, K><<<blocks, threads>>>(out, iters, stamps);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) stream_kernel<KD, K><<<blocks, threads>>>(out, iters, stamps);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<Stamp> h(blocks * 8);
    hipMemcpy(h.data(), stamps, sizeof(Stamp) * blocks * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (const Stamp& s : h)
        if (s.ticks) mhz.push_back(100.0 * (double)s.cycles / (double)s.ticks);
    std::sort(mhz.begin(), mhz.end());
    const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    const double per_simd = (double)iters * 16 * 2;       // MFMAs per SIMD and launch
    const double ns = ms * 1e6 / per_simd;                // wall time per MFMA per SIMD
    const double cyc = ns * clk * 1e-3;                   // at the in-kernel clock
    if (KD == NONE) g_base_ns = ns;
    const double per_mfma = K / 2.0;
    printf("%-44s %5.2f VALU/MFMA  %6.2f ns/MFMA  %5.1f cyc/MFMA  clock %4.0f MHz  %5.0f TFLOP/s", name, per_mfma, ns, cyc, clk,
           per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    if (KD != NONE) printf("   +%.2f ns = +%.1f cyc per instruction", (ns - g_base_ns) / per_mfma, (ns - g_base_ns) / per_mfma * clk * 1e-3);
    printf("\n");
    fflush(stdout);
    hipFree(out);
    hipFree(stamps);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    run<NONE, 0>("bare stream (A from LDS, random data)", reps);
    run<FMAMK, 12>("v_fmamk_f32 (literal)", reps);
    run<FMA, 12>("v_fma_f32", reps);
    run<CVT_PKNORM_U16, 12>("v_cvt_pknorm_u16_f32", reps);
    run<CVT_PKNORM_I16, 12>("v_cvt_pknorm_i16_f32", reps);
    run<CVT_PK_BF16, 12>("v_cvt_pk_bf16_f32", reps);
    run<PK_MIN_U16, 12>("v_pk_min_u16", reps);
    run<PK_MAX_I16, 12>("v_pk_max_i16", reps);
    run<PK_ADD_I16, 12>("v_pk_add_i16 clamp", reps);
    run<MOV, 12>("v_mov_b32", reps);
    run<PERM, 12>("v_perm_b32", reps);
    run<MED3, 12>("v_med3_f32", reps);
    run<EXP, 8>("v_exp_f32", reps);
    run<FMA, 4>("v_fma_f32 x4", reps);
    run<FMA, 8>("v_fma_f32 x8", reps);
    run<FMA, 16>("v_fma_f32 x16", reps);
    run<FMA, 20>("v_fma_f32 x20", reps);
    run<CHAIN6, 12>("v_fma_f32, one dependent chain", reps);
    run<SEQ6, 12>("field ELU, 2 pairs sequential (6 each)", reps);
    run<STAGED6, 12>("field ELU, 4 pairs stage-wise per 2 steps", reps);
    run<SEQ5, 10>("5-instruction form, 2 pairs sequential", reps);
    run<SEQ6, 18>("field ELU, 3 pairs sequential", reps);
    return 0;
}
