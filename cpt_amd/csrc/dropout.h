// Counter-based dropout masks for the training step (reference: nn.Dropout at
// /root/reference/Oscar/oscar/modeling/modeling_bert.py:57 (attention probabilities), :266 (region embeddings) and
// inside the third-party BertEmbeddings / BertSelfOutput / BertOutput blocks; p = config.hidden_dropout_prob /
// attention_probs_dropout_prob, set to --drop_out 0.1 by fewshot/refcoco_cpt.py:387,509-512).
//
// Nothing is stored: forward and backward regenerate the same bits from Philox4x32-10 keyed by the caller's seed, with
// the counter made of (element block, training step, site).  A mask is therefore a pure function of
// (seed, step, site, logical element index) -- independent of tile shapes, launch geometry and the number of GPUs.
//   hidden sites:     element e = row * H + col; one Philox call covers the 4 consecutive columns e & ~3 (32-bit uniforms)
//   attention sites:  4 x 4 blocks of the (query, key) plane of one (sequence, head); two calls per block (rows 0-1 /
//                     rows 2-3), eight 16-bit uniforms each: a lane that owns 4 consecutive keys of one query (forward,
//                     backward phase A) needs ONE call, a lane that owns 4 consecutive queries of one key (phase B) two.
// keep <=> uniform >= threshold;  kept values are scaled by 1 / (1 - p_eff) with p_eff the exactly representable rate.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace cpt {

struct DropSpec {          // by value into kernels; p == 0 <=> thresh == 0 <=> identity
    uint32_t k0, k1;       // Philox key  = 64-bit seed
    uint32_t step;         // counter word 2: optimizer step (a fresh mask every step)
    uint32_t site;         // counter word 3: which dropout of the model (see train.hip)
    uint32_t thresh;       // hidden: 32-bit threshold; attention: 16-bit threshold
    float scale;           // 1 / (1 - p_eff)
};

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                       uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// hidden dropout: keep flags of the 4 consecutive elements starting at e4 * 4 (e4 = element index / 4)
__host__ __device__ __forceinline__ void drop_hidden4(const DropSpec& d, uint64_t e4, bool (&keep)[4]) {
    uint32_t u[4];
    philox4x32_10((uint32_t)e4, (uint32_t)(e4 >> 32), d.step, d.site, d.k0, d.k1, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = u[i] >= d.thresh;
}

// attention dropout: uniforms of rows (qb*4 + 2*half, +1) x columns kb4*4 .. +3 of (sequence, head) bh:
// u16 index = (row & 1) * 4 + (col & 3)
__host__ __device__ __forceinline__ void drop_attn_call(const DropSpec& d, uint32_t bh, uint32_t qb4, uint32_t kb4, uint32_t half,
                                                        uint32_t (&u)[4]) {
    philox4x32_10((qb4 << 16) | (kb4 << 1) | half, bh, d.step, d.site, d.k0, d.k1, u);
}
__host__ __device__ __forceinline__ uint32_t drop_u16(const uint32_t (&u)[4], int idx) { return (u[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu; }

// keep flags of (query q, keys k4*4 .. k4*4 + 3): one call
__host__ __device__ __forceinline__ void drop_attn_row4(const DropSpec& d, uint32_t bh, int q, int k4, bool (&keep)[4]) {
    uint32_t u[4];
    drop_attn_call(d, bh, (uint32_t)q >> 2, (uint32_t)k4, ((uint32_t)q >> 1) & 1, u);
#pragma unroll
    for (int j = 0; j < 4; ++j) keep[j] = drop_u16(u, (q & 1) * 4 + j) >= d.thresh;
}
// keep flags of (queries q4*4 .. q4*4 + 3, key k): two calls
__host__ __device__ __forceinline__ void drop_attn_col4(const DropSpec& d, uint32_t bh, int q4, int k, bool (&keep)[4]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t u[4];
        drop_attn_call(d, bh, (uint32_t)q4, (uint32_t)k >> 2, half, u);
        keep[2 * half] = drop_u16(u, (k & 3)) >= d.thresh;
        keep[2 * half + 1] = drop_u16(u, 4 + (k & 3)) >= d.thresh;
    }
}
// keep bits of (queries 2 rp, 2 rp + 1) x (keys 32 kb .. 32 kb + 31): eight calls, bit i of w0 / w1 = key 32 kb + i of the even / odd query.
// The attention backward fills a [query][key / 32] bit plane in LDS with these once per (sequence, head) -- every uniform computed once per
// workgroup instead of once per lane that touches its 2 x 4 block in each of the kernel's passes (round 6: the Philox rounds were ~3/4 of that kernel's time).
__host__ __device__ __forceinline__ void drop_attn_bits2x32(const DropSpec& d, uint32_t bh, uint32_t rp, uint32_t kb, uint32_t& w0, uint32_t& w1) {
    w0 = 0; w1 = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        uint32_t u[4];
        drop_attn_call(d, bh, rp >> 1, kb * 8 + i, rp & 1, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w0 |= (drop_u16(u, j) >= d.thresh ? 1u : 0u) << (4 * i + j);
            w1 |= (drop_u16(u, 4 + j) >= d.thresh ? 1u : 0u) << (4 * i + j);
        }
    }
}
__host__ __device__ __forceinline__ bool drop_attn_one(const DropSpec& d, uint32_t bh, int q, int k) {
    uint32_t u[4];
    drop_attn_call(d, bh, (uint32_t)q >> 2, (uint32_t)k >> 2, ((uint32_t)q >> 1) & 1, u);
    return drop_u16(u, (q & 1) * 4 + (k & 3)) >= d.thresh;
}

}  // namespace cpt
