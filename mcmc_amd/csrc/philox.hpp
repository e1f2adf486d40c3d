// Counter-based noise for the MC predictive path (device side).
//
// The reference draws its weight noise with torch.randn_like on torch's sequential generator
// (/root/reference/models/bayesian_ensemble.py:43-44); a sequential stream cannot be sharded
// over workgroups or GPUs, so the device path keys Philox4x32-10 (Salmon et al., SC'11) by
// (seed; global pair id, parameter index): results do not depend on launch shape or world size.
// oracle/bnn_oracle.py:philox_normals restates exactly this keying on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmc {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;
constexpr uint32_t kTagNoise = 0x4D434D43u;  // 'MCMC': weight-noise stream
constexpr uint32_t kTagMask = 0x4D41534Bu;   // 'MASK': dropout-mask stream

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)kPhiloxM0 * c.x;
        const uint64_t p1 = (uint64_t)kPhiloxM1 * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return c;
}

// uint32 -> fp32 uniform in (0,1): (2*(x>>9)+1) * 2^-24, exact in fp32.
__host__ __device__ inline float u01(uint32_t x) {
    return ((float)(x >> 9) * 2.0f + 1.0f) * 5.9604644775390625e-08f;
}

// Standard normal for parameter index idx of global pair id `pair` (component idx&3 of the
// 4-wide Philox block idx>>2; two Box-Muller pairs per block).
__device__ inline float philox_normal(uint64_t seed, uint32_t pair, uint32_t idx) {
    const u32x4 x = philox4x32_10(u32x4{idx >> 2, pair, kTagNoise, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t comp = idx & 3u;
    const uint32_t xa = (comp & 2u) ? x.z : x.x;
    const uint32_t xb = (comp & 2u) ? x.w : x.y;
    // Box-Muller on the hardware transcendentals: v_log_f32, and v_sin_f32 / v_cos_f32, which take their argument
    // in revolutions (sin(2 pi u) is one instruction).  Within 2e-6 of the libm restatement in oracle/.
    const float r = sqrtf(-2.0f * __logf(u01(xa)));
    const float u = u01(xb);
    return (comp & 1u) ? r * __builtin_amdgcn_sinf(u) : r * __builtin_amdgcn_cosf(u);
}

// The four normals of Philox block `blk` (= parameter indices 4*blk .. 4*blk+3) at once: the same values
// philox_normal returns for them, for a quarter of the Philox work and half the log / sqrt.
__device__ inline void philox_normal4(uint64_t seed, uint32_t pair, uint32_t blk, float (&z)[4]) {
    const u32x4 x = philox4x32_10(u32x4{blk, pair, kTagNoise, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float r0 = sqrtf(-2.0f * __logf(u01(x.x))), u0 = u01(x.y);
    const float r1 = sqrtf(-2.0f * __logf(u01(x.z))), u1 = u01(x.w);
    z[0] = r0 * __builtin_amdgcn_cosf(u0);
    z[1] = r0 * __builtin_amdgcn_sinf(u0);
    z[2] = r1 * __builtin_amdgcn_cosf(u1);
    z[3] = r1 * __builtin_amdgcn_sinf(u1);
}

// Uniform in (0,1) for dropout element `elem` of row `row`, global pair id `pair`.
__device__ inline float philox_mask_uniform(uint64_t seed, uint32_t pair, uint32_t row, uint32_t elem) {
    const u32x4 x = philox4x32_10(u32x4{elem >> 2, pair, kTagMask, row}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t comp = elem & 3u;
    const uint32_t v = comp == 0 ? x.x : comp == 1 ? x.y : comp == 2 ? x.z : x.w;
    return u01(v);
}

}  // namespace mcmc
