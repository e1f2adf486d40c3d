// Dev tool: time the product's forward_bf16_kernel on random operands without torch (ablations: edit a copy of the kernel
// source and build this harness against it with -DFWD_SRC='"path"').
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Imcmc_amd/csrc tools/fwd_ablate.hip -o build/fwd_ablate
// With -DMCMC_FWD_STAMP (a diagnostic build: the fast kernel stamps s_memtime / s_memrealtime around its tile loop) and a
// fifth argument N it also runs N launches back to back (>= 2 s) and prints the in-kernel clock of the last one.
#ifndef FWD_SRC
#define FWD_SRC "../mcmc_amd/csrc/forward_bf16.hip"
#endif
#include FWD_SRC

#include <stdio.h>
#include <string.h>
#include <algorithm>
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

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 365000;
    const int pairs = argc > 2 ? atoi(argv[2]) : 1024;
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const bool fast = argc > 4 ? atoi(argv[4]) != 0 : true;  // MCMC_DT_BF16_FAST (default) or the exact-ELU kernel
    const int F = 11;
    struct { int K0; int64_t total; } L = {mcmc::k0_pad(F, mcmc::DT_BF16), fast ? mcmc::WPackBF16F(F).total : mcmc::WPackBF16(F).total};
    const int64_t padded = mcmc::rows_pad(rows), ldY = (rows + 63) / 64 * 64;
    // random bf16 in roughly [-0.25, 0.25] for weights, [-2, 2] for inputs
    std::vector<unsigned short> hx(padded * L.K0), hw((size_t)pairs * L.total / 2);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    auto bf = [](float f) { union { float f; unsigned u; } c; c.f = f; const unsigned u = c.u; return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
    for (auto& v : hx) v = bf(4.0f * rnd());
    for (auto& v : hw) v = bf(0.5f * rnd());
    void *dx, *dw;
    float* dy;
    CK(hipMalloc(&dx, hx.size() * 2));
    CK(hipMalloc(&dw, hw.size() * 2));
    CK(hipMalloc(&dy, (size_t)pairs * ldY * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    if (mcmc::launch_forward_bf16(nullptr, fast, dx, rows, F, dw, pairs, nullptr, dy, ldY, 0)) return 1;
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a, 0));
        if (mcmc::launch_forward_bf16(nullptr, fast, dx, rows, F, dw, pairs, nullptr, dy, ldY, 0)) return 1;
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
#ifdef MCMC_FWD_STAMP
    const int burst = argc > 5 ? atoi(argv[5]) : 80;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < burst; ++i)
        if (mcmc::launch_forward_bf16(nullptr, fast, dx, rows, F, dw, pairs, nullptr, dy, ldY, 0)) return 1;
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    {
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> st(2 * 65536);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(mcmc::g_fwd_stamp), st.size() * 8));
        std::vector<double> mhz;
        for (int i = 0; i < 65536; ++i)
            if (st[2 * i + 1]) mhz.push_back(100.0 * (double)st[2 * i] / (double)st[2 * i + 1]);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty())
            printf("stamped build: %d launches back to back %.1f ms each (%.1f TFLOP/s alg); in-kernel clock median %.0f MHz (p10 %.0f, p90 %.0f; %zu workgroups)\n",
                   burst, ms / burst, (double)rows * pairs * 87680.0 / (ms / burst) / 1e9, mhz[mhz.size() / 2], mhz[mhz.size() / 10],
                   mhz[mhz.size() * 9 / 10], mhz.size());
    }
#endif
    const double units = (double)rows * pairs;
    printf("%s rows=%lld pairs=%d best %.3f ms  %.1f TFLOP/s(alg)\n", fast ? "fast " : "exact", (long long)rows, pairs, best, units * 87680.0 / best / 1e9);
    return 0;
}
