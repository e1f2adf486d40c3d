// The peak-weighted loss of one member's batch inside a training kernel (utils.py:7-28 with pred = the mean over the member's
// samples, bayesian_ensemble.py:173-176; temporal_fusion.py:249-251 is the same with one member and one sample): the member's
// global-batch statistics, and dLoss/d(one sample's prediction) of a row.  Shared by train.hip and tft.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmc {
namespace {

// Everything a row's dLoss/dy needs beyond its own (pred, target): the member's global-batch statistics.
struct LossCoef {
    float thr, mean_e, am1, g_over_n, scale;
};

// mean over samples, summed in sample order like torch.mean(0)
__device__ __forceinline__ float pred_mean(const float* __restrict__ Y, int m, int S, int64_t rows, int64_t r) {
    float a = 0.0f;
    for (int s = 0; s < S; ++s) a += Y[((int64_t)m * S + s) * rows + r];
    return a / (float)S;
}

// dLoss / d(one sample's prediction) of a row with sample-mean prediction p and target t
__device__ __forceinline__ float loss_delta(const LossCoef& c, float p, float t) {
    const float d = p - t;
    const float k = t > c.thr ? 1.0f : 0.0f;
    const float g = 1.0f / (1.0f + expf(c.mean_e - fabsf(d)));
    const float w = 1.0f + c.am1 * k * (1.0f + g);
    const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return c.scale * (2.0f * d * w + c.am1 * sgn * (k * d * d * g * (1.0f - g) - c.g_over_n));
}

// All NT threads of a workgroup: the two reduction passes of utils.py:17-28 over member m's batch.  `write` selects
// the one workgroup that stores loss[m] (and adds it to the epoch sum) when several compute the same member.
template <int NT>
__device__ __forceinline__ LossCoef member_loss(const float* __restrict__ Y, const float* __restrict__ ym, int m, int S, int64_t rows,
                                                float alpha, float beta, bool write, float* __restrict__ loss,
                                                double* __restrict__ loss_acc) {
    __shared__ double red[4][NT / 64];
    __shared__ double tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum4 = [&](double (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double x = v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0) red[i][wave] = x;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            double a = 0.0;
            for (int w = 0; w < NT / 64; ++w) a += red[threadIdx.x][w];
            tot[threadIdx.x] = a;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tot[i];
        __syncthreads();
    };
    const double N = (double)rows;
    // a thread's first row (its only one at the reference's batch sizes) is loaded once for both passes
    const bool has0 = threadIdx.x < rows;
    const float t0 = has0 ? ym[threadIdx.x] : 0.0f, p0 = has0 ? pred_mean(Y, m, S, rows, threadIdx.x) : 0.0f;
    auto for_rows = [&](auto&& f) {
        if (has0) f(t0, p0);
        for (int64_t r = (int64_t)threadIdx.x + NT; r < rows; r += NT) f(ym[r], pred_mean(Y, m, S, rows, r));
    };
    double v[4] = {0, 0, 0, 0};
    for_rows([&](float tf, float p) {
        const double t = (double)tf;
        v[0] += t;
        v[1] += t * t;
        v[2] += (double)fabsf(p - tf);
    });
    block_sum4(v);
    const double mean_t = v[0] / N;
    const double var = (v[1] - v[0] * v[0] / N) / (N - 1.0);
    const double sd = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
    LossCoef c;
    c.thr = (float)(mean_t + (double)beta * sd);
    c.mean_e = (float)(v[2] / N);
    c.am1 = alpha - 1.0f;
    double w2[4] = {0, 0, 0, 0};
    for_rows([&](float t, float p) {
        const float d = p - t;
        const float k = t > c.thr ? 1.0f : 0.0f;
        const float g = 1.0f / (1.0f + expf(c.mean_e - fabsf(d)));
        w2[0] += (double)(d * d * (1.0f + c.am1 * k * (1.0f + g)));
        w2[1] += (double)(k * d * d * g * (1.0f - g));
    });
    block_sum4(w2);
    if (write && threadIdx.x == 0) {
        const float lm = (float)(w2[0] / N);
        loss[m] = lm;
        if (loss_acc) loss_acc[m] += (double)lm;  // the epoch sum of bayesian_ensemble.py:181, kept on the device
    }
    c.g_over_n = (float)(w2[1] / N);
    c.scale = (float)(1.0 / (N * (double)S));  // d pred_mean / d pred_sample = 1/S
    return c;
}

}  // namespace
}  // namespace mcmc
