// Dev tool: time the product's reduce_update kernels on random samples without torch (ablations: edit a copy of the kernel
// source and build this harness against it with -DRED_SRC='"path"').
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Imcmc_amd/csrc tools/red_ablate.hip -o build/red_ablate
#ifndef RED_SRC
#define RED_SRC "../mcmc_amd/csrc/reduce.hip"
#endif
#include RED_SRC

#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace mcmc {
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}
int ensure_dyn_lds(const void* kern, size_t bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? 0 : set_error(-2, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
}
}  // namespace mcmc

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

__global__ void fill_kernel(float* y, int64_t n, unsigned seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
        y[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 3.0f + 0.3f;
    }
}
__global__ void edges_kernel(float* lo, float* inv_w, int64_t n, int nbins) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        lo[i] = -1.5f;
        inv_w[i] = (float)(nbins - 2) / 3.6f;
    }
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 3650000;
    const int pairs = argc > 2 ? atoi(argv[2]) : 1024;
    const int nbins = argc > 3 ? atoi(argv[3]) : 128;
    const int iters = argc > 4 ? atoi(argv[4]) : 7;
    const int64_t ldY = (rows + 63) / 64 * 64;
    float *Y, *lo, *inv_w;
    double *sum, *sumsq;
    uint16_t* hist;
    CK(hipMalloc(&Y, (size_t)pairs * ldY * 4));
    CK(hipMalloc(&lo, rows * 4));
    CK(hipMalloc(&inv_w, rows * 4));
    CK(hipMalloc(&sum, rows * 8));
    CK(hipMalloc(&sumsq, rows * 8));
    CK(hipMalloc(&hist, (size_t)rows * nbins * 2));
    fill_kernel<<<4096, 256>>>(Y, (int64_t)pairs * ldY, 1u);
    edges_kernel<<<1024, 256>>>(lo, inv_w, rows, nbins);
    CK(hipMemset(sum, 0, rows * 8));
    CK(hipMemset(sumsq, 0, rows * 8));
    CK(hipMemset(hist, 0, (size_t)rows * nbins * 2));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        for (int i = 0; i < iters + 1; ++i) {
            CK(hipMemsetAsync(hist, 0, (size_t)rows * nbins * 2, 0));  // uint16 counts: one chunk's worth per launch
            CK(hipEventRecord(a, 0));
            if (mcmc::launch_reduce_update(nullptr, Y, ldY, pairs, rows, sum, sumsq, mode ? lo : nullptr, mode ? inv_w : nullptr,
                                           mode ? hist : nullptr, nbins, 0))
                return 1;
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (i && ms < best) best = ms;
        }
        const double sb = 4.0 * pairs * rows, st = 2.0 * rows * (16 + (mode ? 4 + 2.0 * nbins : 0));
        printf("%s rows=%lld pairs=%d nbins=%d best %.3f ms  samples %.0f GB/s  samples+state %.0f GB/s\n", mode ? "hist   " : "moments",
               (long long)rows, pairs, nbins, best, sb / best / 1e6, (sb + st) / best / 1e6);
    }
    return 0;
}
