// Train-mode dropout inside the fused forward kernels (bayesian_ensemble.py:71,86: nn.Dropout(0.1) after each
// ELU; identity in predict(), live in train() and in the opt-in MC-dropout predict).
//
// torch computes x * mask * fp32(1/(1-p)); here the keep flags of the four consecutive features that four
// consecutive accumulator registers of a lane hold (layout.hpp: f(i,h) = (i&3) + 8(i>>2) + 4h) come from one
// 4-byte load of the host mask, or from the 'MASK' Philox stream (oracle/bnn_oracle.py:philox_keep_masks restates it):
// 16 bits per element, so one Philox4x32-10 block serves the EIGHT elements a lane holds in eight accumulator registers.
// elem = position among the 448 hidden activations; block counter (((elem>>4)<<1) | ((elem>>2)&1), global pair id,
// 'MASK', row); slot = 4*((elem>>3)&1) + (elem&3) -> 16-bit half (slot&1) of output word slot>>1;
// keep iff that half < floor((1-p) * 65536)  (p = 0.1: 58,982 / 65,536 = 0.899994).  The mask arithmetic is VALU-bound
// (two 64-bit multiplies per round): 32 bits per element cost the large-batch training kernels twice the time.
#pragma once
#include "layout.hpp"
#include "philox.hpp"

namespace mcmc {

struct DropMask {
    const uint8_t* m[3];  // mode 1: keep-masks [pair_count, n_rows, 256 / 128 / 64], chunk-local pair index
    uint64_t seed;        // mode 2: Philox key
    int mode;             // 0 none, 1 host masks, 2 Philox
    float keep_prob;      // 1 - p
    float scale;          // fp32(1/(1-p))
    int M, S;             // mode 2: processing index q = pair_begin + c -> global pair id (q % M) * S + q / M
    int64_t pair_begin;
    int64_t row_begin;    // mode 2: global index of the launch's row 0 (a row block of a larger predict keys the masks of its
                          // rows by their global index, so the masks do not depend on how the rows are cut into launches)
};

__device__ __forceinline__ uint32_t drop_pair_id(const DropMask& dm, int c) {
    const int64_t q = dm.pair_begin + c;
    return (uint32_t)((q % dm.M) * dm.S + q / dm.M);
}

// o[0..15]: activated accumulator tile `mo` (32 features) of hidden layer `layer` for lane half h, batch row `row`.
template <int MODE>
__device__ __forceinline__ void drop16(float (&o)[16], const DropMask& dm, int layer, int mo, int h, int c, uint32_t g,
                                       int64_t n_rows, int64_t row) {
    if (MODE == 0) return;
    const int width = layer == 0 ? H1 : layer == 1 ? H2 : H3;
    const int base = layer == 0 ? 0 : layer == 1 ? H1 : H1 + H2;
    uint32_t words[4] = {0, 0, 0, 0};
    const uint32_t thr = (uint32_t)(dm.keep_prob * 65536.0f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int feat0 = 32 * mo + 8 * q + 4 * h;
        bool keep[4];
        if (MODE == 1) {
            // rows past n_rows are padding: computed, never stored, and have no mask bytes
            const uint32_t w = row < n_rows ? *reinterpret_cast<const uint32_t*>(dm.m[layer] + ((int64_t)c * n_rows + row) * width + feat0)
                                            : 0x01010101u;
#pragma unroll
            for (int j = 0; j < 4; ++j) keep[j] = ((w >> (8 * j)) & 0xFFu) != 0;
        } else {
            if ((q & 1) == 0) {  // one block for registers 4q .. 4q+7
                // The counter is made to "depend" on the activation it will mask: otherwise the scheduler hoists
                // every Philox block of the fully unrolled network to the top of the tile and spills.
                uint32_t ctr0 = (((uint32_t)(base + 32 * mo + 8 * q) >> 4) << 1) | (uint32_t)h;
                asm volatile("" : "+v"(ctr0) : "v"(o[4 * q]));
                const u32x4 x = philox4x32_10(u32x4{ctr0, g, kTagMask, (uint32_t)(dm.row_begin + row)}, (uint32_t)dm.seed, (uint32_t)(dm.seed >> 32));
                words[0] = x.x;
                words[1] = x.y;
                words[2] = x.z;
                words[3] = x.w;
            }
            const uint32_t wa = words[2 * (q & 1)], wb = words[2 * (q & 1) + 1];
            keep[0] = (wa & 0xFFFFu) < thr;
            keep[1] = (wa >> 16) < thr;
            keep[2] = (wb & 0xFFFFu) < thr;
            keep[3] = (wb >> 16) < thr;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * q + j] = keep[j] ? o[4 * q + j] * dm.scale : -0.0f;  // -0: zero, and marks "dropped" for train.hip's backward
    }
}

}  // namespace mcmc
