// HBM-bound row kernels of the CPT hot path: embedding gather + LayerNorm, (residual) LayerNorm,
// pad/cast of region features, row gather, cross-entropy over [MASK] rows.
// One 64-lane wavefront per row, 16-byte loads, statistics by wave shuffles; fp32 math.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace cpt {

constexpr int ROW_THREADS = 256;            // 4 waves = 4 rows per workgroup
constexpr int MAXV = 4;                     // float4 per lane: H <= 1024

// two-pass mean / biased variance over a row held as up to MAXV float4 per lane
template <int NA = MAXV>      // NA: float4 per lane the caller holds (MAXV = any H <= 1024; 3 = the H = 768 fast kernel: a third fewer registers)
__device__ __forceinline__ void ln_stats(const f32x4 (&v)[NA], int nv, int lane, int H, float& mean, float& rstd, float eps) {
    constexpr int MAXV = NA;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < H) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
}

// index (in 4-element quads) of columns [c, c + 4) of `row` in a tensor kept in the panel layout [rows / 32][H / 16][64][8] (gemm_prod.hip)
__device__ __forceinline__ size_t panel_quad(size_t row, int c, int H) {
    return ((size_t)panel_unit((int)row, c >> 3, H >> 4) << 1) + ((c >> 2) & 1);
}

template <typename LP, int NA = MAXV>
__device__ __forceinline__ void ln_write(const f32x4 (&v)[NA], int nv, int lane, int H, float mean, float rstd,
                                         const float* g, const float* b, float* of, LP* ol, signed char* olo = nullptr, long long panel_row = -1,
                                         const DropSpec* odrop = nullptr, size_t drop_row = 0) {
    // odrop (round 6, training forward): hidden dropout on the OUTPUT row (BertEmbeddings' dropout / modeling_bert.py:266 on the region rows), element index
    // drop_row * H + c of that site's mask -- the embedding launches apply it themselves instead of a dropout pass over all rows behind them
    constexpr int MAXV = NA;
    // olo != NULL (bf16 LP only): the row leaves in the 3-byte residual form (common.h r3_encode): hi -> ol, lo -> olo
    // panel_row >= 0 (round 5, with olo): ol / olo are the BASES of the panel-layout residual stream and the row lands at its quads there
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
            f32x4 y;
            if (g) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
            } else {
                y = v[i];
            }
            if (odrop && odrop->thresh != 0) {
                bool keep[4];
                drop_hidden4(*odrop, ((uint64_t)drop_row * H + c) >> 2, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = keep[j] ? y[j] * odrop->scale : 0.f;
            }
            if (of) *reinterpret_cast<f32x4*>(of + c) = y;
            if constexpr (sizeof(LP) == 2) {
                if (olo) {
                    u32x2_t hq; unsigned lq;
                    r3_encode(y, hq, lq);
                    if (panel_row >= 0) {
                        const size_t qd = panel_quad((size_t)panel_row, c, H);
                        reinterpret_cast<u32x2_t*>(ol)[qd] = hq;
                        reinterpret_cast<unsigned*>(olo)[qd] = lq;
                        continue;
                    }
                    *reinterpret_cast<u32x2_t*>(ol + c) = hq;
                    *reinterpret_cast<unsigned*>(olo + c) = lq;
                    continue;
                }
            }
            if (ol) {
                if constexpr (sizeof(LP) == 2) {
                    bf16x4 p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[j] = (bf16)y[j];
                    *reinterpret_cast<bf16x4*>(ol + c) = p;
                } else {
                    *reinterpret_cast<f32x4*>(ol + c) = y;
                }
            }
        }
    }
}

// Round 3: LN_RPW rows per wave with every row's loads issued before the first row is reduced (a wave with ONE row kept 3 KB in
// flight and the kernel ran at the latency of its load -> reduce -> store chain: 3.9-4.6 TB/s on Infinity-Cache-resident rows).
// Per-row arithmetic and order are unchanged: same bits.
// Chosen per launch: two rows per wave from 24576 rows on (> 256 MB working sets: 3.85 -> 4.9 TB/s for the bf16-only output form); below
// that one row per wave keeps twice the waves on the chip, which is what hides the latency of a small launch (7680 rows: 9.1 vs 11.3 us).
template <typename LP, bool GELU_IN, int LN_RPW, int NA = 4>      // NA = 3: the H = 768 instantiation (row and residual arrays of three float4 per lane instead of four)
__global__ __launch_bounds__(ROW_THREADS) void layernorm_rows_kernel(
    const float* x, const float* __restrict__ g, const float* __restrict__ bta, float eps,
    float* out_f32, LP* __restrict__ out_lp, int R, int H, int grp, int grp_stride, int grp_off,
    const float* resid, DropSpec dr, float* pre_out, signed char* __restrict__ out_lo, int x_parts, size_t x_stride, int out_panel, float* __restrict__ stat_out, RowMap drows, unsigned short* __restrict__ keep_out,
    const float* __restrict__ resid_stat, const float* __restrict__ resid_g, const float* __restrict__ resid_b, DropSpec odr) {      // odr (round 6): dropout on the OUTPUT rows (ln_write odrop; index = output row);   resid_stat / resid_g / resid_b (round 6): `resid` holds the PRE-LayerNorm rows of the LayerNorm in front; the residual is re-formed from them with its (mean, rstd), gain and shift -- ln_write's expression -- so that launch need not write its fp32 output at all (11.8 MB per launch at 3840 rows);   stat_out (round 6): [R][2] (mean, rstd) per row for the backward pass; x / out_f32 / resid / pre_out may alias (in-place calls of the training step): no __restrict__ on them
    // x_parts / x_stride (round 3): x is x_parts split-K partial matrices of the dense layer in front, x_stride elements apart; the row that is
    // processed is their sum in split order (training forward: no reduction launch between the GEMM and this pass)
    // resid / dr / pre_out (training forward of LN(dropout(dense) + residual), modeling_bert.py:85-86,145 with the third-party
    // BertSelfOutput / BertOutput): the row that is normalised is dropout(x) + resid, written to pre_out for the backward pass --
    // the same arithmetic, in the same order, as the dropout_rows pass this replaces
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6)) * LN_RPW;
    if (r0 >= R) return;
    constexpr int MAXV = NA;
    const int nv = (H + 255) / 256;
    f32x4 v[LN_RPW][MAXV], rr[LN_RPW][MAXV];
#pragma unroll
    for (int u = 0; u < LN_RPW; ++u) {
        const int r = r0 + u;
        if (r >= R) continue;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                v[u][i] = *reinterpret_cast<const f32x4*>(x + (size_t)r * H + c);
                // (eight partial matrices' loads in flight at a time, added in split order: a one-by-one loop serialised up to 15 memory round trips per row)
                for (int k0 = 1; k0 < x_parts; k0 += 8) {
                    f32x4 t[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        if (k0 + kk < x_parts) t[kk] = *reinterpret_cast<const f32x4*>(x + (size_t)(k0 + kk) * x_stride + (size_t)r * H + c);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        if (k0 + kk < x_parts) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[u][i][j] += t[kk][j];
                        }
                }
                if (resid) rr[u][i] = *reinterpret_cast<const f32x4*>(resid + (size_t)r * H + c);
            }
        }
        if (resid && resid_stat) {
            const float2 st = *reinterpret_cast<const float2*>(resid_stat + 2 * (size_t)r);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (lane + 64 * i) * 4;
                if (i < nv && c < H) {
                    const f32x4 gg = *reinterpret_cast<const f32x4*>(resid_g + c);
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(resid_b + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[u][i][j] = (rr[u][i][j] - st.x) * st.y * gg[j] + bb[j];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < LN_RPW; ++u) {
        const int r = r0 + u;
        if (r >= R) continue;
        unsigned kbits = 0;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                if (GELU_IN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][i][j] = gelu_erf(v[u][i][j]);
                }
                if (dr.thresh != 0) {
                    bool keep[4];
                    drop_hidden4(dr, ((uint64_t)rowmap_row(drows, r) * H + c) >> 2, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[u][i][j] = keep[j] ? v[u][i][j] * dr.scale : 0.f; kbits |= (keep[j] ? 1u : 0u) << (4 * i + j); }
                }
                if (resid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][i][j] += rr[u][i][j];
                }
                if (pre_out) *reinterpret_cast<f32x4*>(pre_out + (size_t)r * H + c) = v[u][i];
            }
        }
        if (keep_out) keep_out[(size_t)r * 64 + lane] = (unsigned short)kbits;
        float mean = 0.f, rstd = 1.f;
        if (g) ln_stats<NA>(v[u], nv, lane, H, mean, rstd, eps);
        if (stat_out && lane == 0) *reinterpret_cast<float2*>(stat_out + 2 * (size_t)r) = float2{mean, rstd};
        const size_t orow = (size_t)(r / grp) * grp_stride + grp_off + (r % grp);
        if (out_panel) ln_write<LP, NA>(v[u], nv, lane, H, mean, rstd, g, bta, nullptr, out_lp, out_lo, (long long)orow);
        else
        ln_write<LP, NA>(v[u], nv, lane, H, mean, rstd, g, bta, out_f32 ? out_f32 + orow * H : nullptr,
                     out_lp ? out_lp + orow * H : nullptr, out_lo ? out_lo + orow * H : nullptr, -1, &odr, orow);
    }
}

// Round 5: the plain LayerNorm of H = 768 rows (no residual, no dropout, no GELU, no split-K partials: the inference-side launches and
// bench.py's hbm_kernels) with RPW rows per wave and only the registers those rows need -- three float4 per lane per row, no residual
// copy: 2 rows per wave at the occupancy the general kernel has with one (its MAXV = 4 arrays of the row AND of the residual made RPW = 2
// cost half the waves, so the bytes in flight per CU never grew).  The same helpers, the same arithmetic in the same order: the same bits.
template <typename LP, int RPW>
__global__ __launch_bounds__(ROW_THREADS) void layernorm768_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                                   float* __restrict__ out_f32, LP* __restrict__ out_lp, int R, int grp, int grp_stride, int grp_off,
                                                                   signed char* __restrict__ out_lo, int out_panel) {
    constexpr int H = 768, NV = 3;
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6)) * RPW;
    if (r0 >= R) return;
    f32x4 v[RPW][NV];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int r = min(r0 + u, R - 1);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[u][i] = *reinterpret_cast<const f32x4*>(x + (size_t)r * H + (lane + 64 * i) * 4);
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int r = r0 + u;
        if (r >= R) continue;
        float mean = 0.f, rstd = 1.f;
        if (g) ln_stats<NV>(v[u], NV, lane, H, mean, rstd, eps);
        const size_t orow = (size_t)(r / grp) * grp_stride + grp_off + (r % grp);
        if (out_panel) ln_write<LP, NV>(v[u], NV, lane, H, mean, rstd, g, bta, nullptr, out_lp, out_lo, (long long)orow);
        else ln_write<LP, NV>(v[u], NV, lane, H, mean, rstd, g, bta, out_f32 ? out_f32 + orow * H : nullptr, out_lp ? out_lp + orow * H : nullptr,
                              out_lo ? out_lo + orow * H : nullptr);
    }
}

int layernorm_rows_ex(const float* x, const float* g, const float* bta, float eps, float* out_f32,
                      void* out_lp, int lp_dtype, int R, int H, int grp, int grp_stride, int grp_off,
                      int gelu_in, hipStream_t s, const float* resid, const DropSpec* drop, float* pre_out, void* out_lo, int x_parts, size_t x_stride, int out_panel, float* stat_out, const RowMap* drop_rows, unsigned short* keep_out,
                      const float* resid_stat, const float* resid_g, const float* resid_b, const DropSpec* out_drop) {
    if (x_parts < 1 || x_parts > 64) return CPT_ERR_SHAPE;
    const DropSpec odr = out_drop ? *out_drop : DropSpec{};
    if (odr.thresh != 0 && out_panel) return CPT_ERR_SHAPE;
    if (resid_stat && !(resid && resid_g && resid_b)) return CPT_ERR_NULL;
    if (out_panel && (!out_lo || out_f32 || H % 16)) return CPT_ERR_SHAPE;      // panel output: the 3-byte residual stream only (rows rounded up to 32 by the caller's buffer)
    if (out_lo && !(out_lp && lp_dtype == CPT_BF16)) return CPT_ERR_DTYPE;
    if (R <= 0 || H <= 0 || H % 4 || H > 256 * MAXV || grp <= 0) return CPT_ERR_SHAPE;
    if (!x || (!out_f32 && !out_lp)) return CPT_ERR_NULL;
    const DropSpec dr = drop ? *drop : DropSpec{};
    const RowMap drows = drop_rows ? *drop_rows : RowMap{nullptr, 0, 0};
    if (grp_stride == 0 && grp != R) grp_stride = grp;
    // H = 768 rows without residual / dropout / GELU / split-K partials: the lean kernel, one row per wave (measured against two and four rows per wave:
    // profiles/r05_ab_log.md -- 0.59 / 0.72 of 8 TB/s at 7680 / 61440 rows for the bf16 output against 0.58 / 0.71 and 0.49 / 0.67; the general kernel: 0.56 / 0.55)
    if (keep_out && dr.thresh == 0) keep_out = nullptr;
    if (H == 768 && !gelu_in && !resid && dr.thresh == 0 && !pre_out && x_parts == 1 && !stat_out && odr.thresh == 0) {
        dim3 g768((R + 3) / 4), b768(ROW_THREADS);
        if (out_lp && lp_dtype == CPT_BF16)
            layernorm768_kernel<bf16, 1><<<g768, b768, 0, s>>>(x, g, bta, eps, out_f32, (bf16*)out_lp, R, grp, grp_stride, grp_off, (signed char*)out_lo, out_panel);
        else
            layernorm768_kernel<float, 1><<<g768, b768, 0, s>>>(x, g, bta, eps, out_f32, (float*)out_lp, R, grp, grp_stride, grp_off, (signed char*)out_lo, out_panel);
        return CPT_OK;
    }
    const int rpw = R >= 24576 ? 2 : 1;
    dim3 grid((R + 4 * rpw - 1) / (4 * rpw)), block(ROW_THREADS);
    const bool lp16 = out_lp && lp_dtype == CPT_BF16;
#define LNK(LPT, GI)                                                                                                                              \
    do {                                                                                                                                          \
        if (rpw == 2) layernorm_rows_kernel<LPT, GI, 2><<<grid, block, 0, s>>>(x, g, bta, eps, out_f32, (LPT*)out_lp, R, H, grp, grp_stride, grp_off, resid, dr, pre_out, (signed char*)out_lo, x_parts, x_stride, out_panel, stat_out, drows, keep_out, resid_stat, resid_g, resid_b, odr); \
        else if (H == 768) layernorm_rows_kernel<LPT, GI, 1, 3><<<grid, block, 0, s>>>(x, g, bta, eps, out_f32, (LPT*)out_lp, R, H, grp, grp_stride, grp_off, resid, dr, pre_out, (signed char*)out_lo, x_parts, x_stride, out_panel, stat_out, drows, keep_out, resid_stat, resid_g, resid_b, odr); \
        else layernorm_rows_kernel<LPT, GI, 1><<<grid, block, 0, s>>>(x, g, bta, eps, out_f32, (LPT*)out_lp, R, H, grp, grp_stride, grp_off, resid, dr, pre_out, (signed char*)out_lo, x_parts, x_stride, out_panel, stat_out, drows, keep_out, resid_stat, resid_g, resid_b, odr);          \
    } while (0)
    if (lp16) { if (gelu_in) LNK(bf16, true); else LNK(bf16, false); }
    else      { if (gelu_in) LNK(float, true); else LNK(float, false); }
#undef LNK
    return CPT_OK;
}

int layernorm_rows(const float* x, const float* g, const float* bta, float eps, float* out_f32,
                   void* out_lp, int lp_dtype, int R, int H, int grp, int grp_stride, int grp_off,
                   hipStream_t s) {
    return layernorm_rows_ex(x, g, bta, eps, out_f32, out_lp, lp_dtype, R, H, grp, grp_stride, grp_off, 0, s);
}

// ---- BertEmbeddings: gather 3 rows, add, LayerNorm -------------------------------------------
struct EmbedArgs {
    const int64_t *ids, *tt, *pos;
    const float *word, *posw, *typew, *g, *bta;
    float eps;
    float* out_f32;
    void* out_lp;
    signed char* out_lo;
    int B, Lt, L, H, vocab, max_pos, type_vocab;
    int out_panel;      // round 5: out_lp / out_lo are the panel-layout residual stream (3-byte form)
    float* zero_f2; unsigned* zero_u1;      // round 6 (training forward): cleared by the first workgroup (kernels.h)
    DropSpec odrop;                         // round 6 (training forward): BertEmbeddings' dropout on the output rows (thresh 0: none)
};
// ERW rows per wave (round 6: 2 -- every gather of both rows in flight before the first row is reduced; with one row per wave the launch ran at the
// latency of its ids -> table rows -> reduce -> store chain, 0.47 of 8 TB/s on the unique-bytes model at 4480 rows).  Per-row arithmetic unchanged: same bits.
// Chosen per launch like the LayerNorm's rows per wave: measured 10.6 vs 8.9 us at 4480 rows (one resident round of waves either way: half the waves
// only lengthen each wave's chain) and 49.7 vs 53.3 us at 35840 rows -- two rows per wave from 16384 rows on.
constexpr int EMB_RPW2_ROWS = 16384;
template <typename LP, int NA = MAXV, int ERW = 1>      // NA = 3: the H = 768 instantiation (a quarter fewer row registers, as layernorm768_kernel)
__device__ __forceinline__ void embed_ln_block(const EmbedArgs& a, int block) {
    constexpr int MAXV = NA;
    const int64_t* __restrict__ ids = a.ids; const int64_t* __restrict__ tt = a.tt; const int64_t* __restrict__ pos = a.pos;
    const float* __restrict__ word = a.word; const float* __restrict__ posw = a.posw; const float* __restrict__ typew = a.typew;
    const float* __restrict__ g = a.g; const float* __restrict__ bta = a.bta;
    const float eps = a.eps;
    float* __restrict__ out_f32 = a.out_f32; LP* __restrict__ out_lp = (LP*)a.out_lp; signed char* __restrict__ out_lo = a.out_lo;
    const int B = a.B, Lt = a.Lt, L = a.L, H = a.H, vocab = a.vocab, max_pos = a.max_pos, type_vocab = a.type_vocab;
    const int lane = threadIdx.x & 63;
    if (block == 0) {
        if (a.zero_f2 && threadIdx.x < 2) a.zero_f2[threadIdx.x] = 0.f;
        if (a.zero_u1 && threadIdx.x == 2) *a.zero_u1 = 0u;
    }
    const int r0 = (block * (ROW_THREADS / 64) + (threadIdx.x >> 6)) * ERW;
    if (r0 >= B * Lt) return;
    const int nv = (H + 255) / 256;
    long wid[ERW], pid[ERW], tid[ERW];
#pragma unroll
    for (int u = 0; u < ERW; ++u) {
        const int r = min(r0 + u, B * Lt - 1);
        wid[u] = ids[r];
        pid[u] = pos ? pos[r] : r % Lt;
        tid[u] = tt ? tt[r] : 0;
    }
    f32x4 v[ERW][MAXV];
#pragma unroll
    for (int u = 0; u < ERW; ++u) {
        // out-of-range ids would be a host bug; clamp so the kernel never faults
        wid[u] = wid[u] < 0 ? 0 : (wid[u] >= vocab ? vocab - 1 : wid[u]);
        pid[u] = pid[u] < 0 ? 0 : (pid[u] >= max_pos ? max_pos - 1 : pid[u]);
        tid[u] = tid[u] < 0 ? 0 : (tid[u] >= type_vocab ? type_vocab - 1 : tid[u]);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                const f32x4 aw = *reinterpret_cast<const f32x4*>(word + (size_t)wid[u] * H + c);
                const f32x4 p = *reinterpret_cast<const f32x4*>(posw + (size_t)pid[u] * H + c);
                const f32x4 q = *reinterpret_cast<const f32x4*>(typew + (size_t)tid[u] * H + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][i][j] = aw[j] + p[j] + q[j];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < ERW; ++u) {
        const int r = r0 + u;
        if (r >= B * Lt) continue;
        const int b = r / Lt, t = r % Lt;
        float mean, rstd;
        ln_stats<NA>(v[u], nv, lane, H, mean, rstd, eps);
        const size_t orow = (size_t)b * L + t;
        if (a.out_panel) ln_write<LP, NA>(v[u], nv, lane, H, mean, rstd, g, bta, nullptr, out_lp, out_lo, (long long)orow);
        else
        ln_write<LP, NA>(v[u], nv, lane, H, mean, rstd, g, bta, out_f32 ? out_f32 + orow * H : nullptr,
                         out_lp ? out_lp + orow * H : nullptr, out_lo ? out_lo + orow * H : nullptr, -1, &a.odrop, orow);
    }
}
template <typename LP, int NA = MAXV, int ERW = 1>
__global__ __launch_bounds__(ROW_THREADS) void embed_ln_kernel(EmbedArgs a) { embed_ln_block<LP, NA, ERW>(a, blockIdx.x); }

int embed_ln(const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word,
             const float* posw, const float* typew, const float* g, const float* bta, float eps,
             float* out_f32, void* out_lp, int lp_dtype, int B, int Lt, int L, int H, int vocab,
             int max_pos, int type_vocab, hipStream_t s, void* out_lo, int out_panel, float* zero_f2, unsigned* zero_u1, const DropSpec* out_drop) {
    if (B <= 0 || Lt <= 0 || L < Lt || H % 4 || H > 256 * MAXV) return CPT_ERR_SHAPE;
    if (out_panel && (!out_lo || out_f32 || H % 16)) return CPT_ERR_SHAPE;
    if (out_lo && !(out_lp && lp_dtype == CPT_BF16)) return CPT_ERR_DTYPE;
    if (!ids || !word || !posw || !typew || !g || !bta) return CPT_ERR_NULL;
    if (!pos && Lt > max_pos) return CPT_ERR_SHAPE;
    const int erw = (B * Lt >= EMB_RPW2_ROWS && H == 768) ? 2 : 1;
    dim3 grid((B * Lt + 4 * erw - 1) / (4 * erw)), block(ROW_THREADS);
    EmbedArgs a{ids, tt, pos, word, posw, typew, g, bta, eps, out_f32, out_lp, (signed char*)out_lo, B, Lt, L, H, vocab, max_pos, type_vocab, out_panel ? 1 : 0, zero_f2, zero_u1, (out_drop && !out_panel) ? *out_drop : DropSpec{}};
    if (out_lp && lp_dtype == CPT_BF16) { if (H == 768 && erw == 2) embed_ln_kernel<bf16, 3, 2><<<grid, block, 0, s>>>(a); else if (H == 768) embed_ln_kernel<bf16, 3><<<grid, block, 0, s>>>(a); else embed_ln_kernel<bf16><<<grid, block, 0, s>>>(a); }
    else { a.out_lo = nullptr; if (H == 768) embed_ln_kernel<float, 3><<<grid, block, 0, s>>>(a); else embed_ln_kernel<float><<<grid, block, 0, s>>>(a); }
    return CPT_OK;
}

// ---- pad + cast: x[R][K] f32 -> out[R][Kp] T ---------------------------------------------------
// One thread per 8 output elements: four 8-byte loads (rows of x are 8-byte aligned when K is even, e.g. the
// 2054-float region features) and one 16-byte (bf16) / two 16-byte (f32) stores.
// Round 3: four chunks per thread (grid-stride apart, so a wave's accesses stay contiguous), all eight 16-byte loads in flight
// before the first store.  Measured: 13.7 vs 13.2 us for the 26 MB region-feature read -- no gain over one chunk per thread, so the
// pass is not bound by bytes in flight (its rows start on 8-byte boundaries: every 16-byte load straddles).
constexpr int PC_UNR = 4;
template <typename T>
__device__ __forceinline__ void pad_cast_block(const float* __restrict__ x, T* __restrict__ out, int R, int K, int Kp, int block, int nblocks) {
    const int cpr = Kp / 8;                                   // 8-element chunks per output row
    const size_t total = (size_t)R * cpr, stride = (size_t)nblocks * 256;
    const size_t idx0 = (size_t)block * 256 + threadIdx.x;
    float v[PC_UNR][8];
#pragma unroll
    for (int u = 0; u < PC_UNR; ++u) {
        const size_t idx = idx0 + u * stride;
        if (idx >= total) continue;
        const int r = (int)(idx / cpr), c = (int)(idx % cpr) * 8;
        const float* src = x + (size_t)r * K + c;
        if (c + 8 <= K && (K & 1) == 0) {
            // rows of K = 2054 floats start on 8-byte boundaries only: two 16-byte loads from 8-byte-aligned addresses (legal: the HSA target runs
            // in unaligned-access mode) instead of four 8-byte ones -- half the load instructions of this HBM-bound pass
            typedef f32x4 f32x4_a8 __attribute__((aligned(8)));
            const f32x4 t0 = *reinterpret_cast<const f32x4_a8*>(src), t1 = *reinterpret_cast<const f32x4_a8*>(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[u][e] = t0[e]; v[u][4 + e] = t1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = c + e < K ? src[e] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < PC_UNR; ++u) {
        const size_t idx = idx0 + u * stride;
        if (idx >= total) continue;
        const int r = (int)(idx / cpr), c = (int)(idx % cpr) * 8;
        T* dst = out + (size_t)r * Kp + c;
        if constexpr (sizeof(T) == 2) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[u][e];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[u][0], v[u][1], v[u][2], v[u][3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[u][4], v[u][5], v[u][6], v[u][7]};
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ x, T* __restrict__ out, int R, int K, int Kp) {
    pad_cast_block<T>(x, out, R, K, Kp, blockIdx.x, gridDim.x);
}
// Round 4: the text embedding and the region-feature pad + cast write disjoint rows of the model's input and depend on nothing: ONE launch
// runs both (the first `npad` workgroups convert the regions, the rest gather + normalise the text rows), so that the embedding's gathers
// ride under the HBM-bound conversion instead of a 10 us launch of their own in front of it.
static_assert(ROW_THREADS == 256, "embed_pad_kernel: both bodies are written for 256 threads");
template <int NA>
__global__ __launch_bounds__(256) void embed_pad_kernel(EmbedArgs a, const float* __restrict__ x, bf16* __restrict__ xo, int R, int K, int Kp, int npad) {
    if ((int)blockIdx.x < npad) pad_cast_block<bf16>(x, xo, R, K, Kp, blockIdx.x, npad);
    else embed_ln_block<bf16, NA>(a, blockIdx.x - npad);
}

int embed_ln_pad_cast(const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word, const float* posw, const float* typew,
                      const float* g, const float* bta, float eps, void* out_lp, void* out_lo, int B, int Lt, int L, int H, int vocab,
                      int max_pos, int type_vocab, const float* x, void* xo, int R, int K, int Kp, hipStream_t s, int out_panel,
                      float* out_f32, float* zero_f2, unsigned* zero_u1, const DropSpec* out_drop) {
    if (out_panel && (!out_lo || H % 16 || out_f32)) return CPT_ERR_SHAPE;
    if (B <= 0 || Lt <= 0 || L < Lt || H % 4 || H > 256 * MAXV || R <= 0 || K <= 0 || Kp < K || Kp % 8) return CPT_ERR_SHAPE;
    if (!ids || !word || !posw || !typew || !g || !bta || !out_lp || !x || !xo) return CPT_ERR_NULL;
    if (!pos && Lt > max_pos) return CPT_ERR_SHAPE;
    if (((uintptr_t)xo % 16) || ((uintptr_t)x % 8)) return CPT_ERR_ALIGN;
    const size_t n = (size_t)R * (Kp / 8);
    const int npad = (int)((n + 256 * PC_UNR - 1) / (256 * PC_UNR));
    EmbedArgs a{ids, tt, pos, word, posw, typew, g, bta, eps, out_f32, out_lp, (signed char*)out_lo, B, Lt, L, H, vocab, max_pos, type_vocab, out_panel ? 1 : 0, zero_f2, zero_u1, (out_drop && !out_panel) ? *out_drop : DropSpec{}};
    if (H == 768) embed_pad_kernel<3><<<dim3(npad + (B * Lt + 3) / 4), dim3(256), 0, s>>>(a, x, (bf16*)xo, R, K, Kp, npad);
    else embed_pad_kernel<MAXV><<<dim3(npad + (B * Lt + 3) / 4), dim3(256), 0, s>>>(a, x, (bf16*)xo, R, K, Kp, npad);
    return CPT_OK;
}

// generic fallback (Kp not a multiple of 8 or unaligned output): one thread per pair of output elements
template <typename T>
__global__ __launch_bounds__(256) void pad_cast_pairs_kernel(const float* __restrict__ x, T* __restrict__ out, int R, int K, int Kp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int hp = Kp / 2;
    if (idx >= (size_t)R * hp) return;
    const int r = (int)(idx / hp), c = (int)(idx % hp) * 2;
    const float a = c < K ? x[(size_t)r * K + c] : 0.f;
    const float b = c + 1 < K ? x[(size_t)r * K + c + 1] : 0.f;
    out[(size_t)r * Kp + c] = from_f32<T>(a);
    out[(size_t)r * Kp + c + 1] = from_f32<T>(b);
}

int pad_cast(const float* x, void* out, int dtype, int R, int K, int Kp, hipStream_t s) {
    if (R <= 0 || K <= 0 || Kp < K || Kp % 2) return CPT_ERR_SHAPE;
    if (dtype != CPT_BF16 && dtype != CPT_F32) return CPT_ERR_DTYPE;
    dim3 block(256);
    if (Kp % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)x % 8) == 0) {
        const size_t n = (size_t)R * (Kp / 8);
        dim3 grid((unsigned)((n + 256 * PC_UNR - 1) / (256 * PC_UNR)));
        if (dtype == CPT_BF16) pad_cast_kernel<bf16><<<grid, block, 0, s>>>(x, (bf16*)out, R, K, Kp);
        else pad_cast_kernel<float><<<grid, block, 0, s>>>(x, (float*)out, R, K, Kp);
        return CPT_OK;
    }
    const size_t n = (size_t)R * (Kp / 2);
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == CPT_BF16) pad_cast_pairs_kernel<bf16><<<grid, block, 0, s>>>(x, (bf16*)out, R, K, Kp);
    else pad_cast_pairs_kernel<float><<<grid, block, 0, s>>>(x, (float*)out, R, K, Kp);
    return CPT_OK;
}

// ---- split-operand copies of the bf16x3 parity mode: hi = bf16(x), lo = bf16(x - hi); row layout hi | hi | lo (activations)
// or hi | lo | hi (weights), so that one bf16 GEMM over K' = 3K gives hi.hi + hi.lo + lo.hi ------------------------------------
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ld, bf16* __restrict__ out, int R, int K, int worder) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 4-element chunk per thread
    const int kc = K / 4;
    if (i >= (size_t)R * kc) return;
    const int r = (int)(i / kc), c = (int)(i % kc) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c);
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
    bf16* o = out + (size_t)r * 3 * K + c;
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + K) = worder ? lo : hi;
    *reinterpret_cast<bf16x4*>(o + 2 * K) = worder ? hi : lo;
}

int split3(const float* x, int ld, void* out, int R, int K, int weight_order, hipStream_t s) {
    if (R <= 0 || K <= 0 || K % 4 || ld % 4 || ld < K) return CPT_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 7)) return CPT_ERR_ALIGN;
    const size_t n = (size_t)R * (K / 4);
    split3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(x, ld, (bf16*)out, R, K, weight_order);
    return CPT_OK;
}

// The same split for a product that contracts over ROWS (bf16x3 training: a weight gradient dY^T . X over the M rows, read by the TN
// kernel as stored): three row blocks of Rp rows each -- hi, hi, lo (dY) or hi, lo, hi (X) -- rows R..Rp-1 of every block zero.
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ x, int ld, bf16* __restrict__ out, int R, int Rp, int C, int worder) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cc = C / 4;
    if (i >= (size_t)Rp * cc) return;
    const int r = (int)(i / cc), c = (int)(i % cc) * 4;
    bf16x4 hi, lo;
    if (r < R) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (bf16)0.f; lo[e] = (bf16)0.f; }
    }
    bf16* o = out + (size_t)r * C + c;
    const size_t blk = (size_t)Rp * C;
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + blk) = worder ? lo : hi;
    *reinterpret_cast<bf16x4*>(o + 2 * blk) = worder ? hi : lo;
}

int split3_rows(const float* x, int ld, void* out, int R, int Rp, int C, int weight_order, hipStream_t s) {
    if (R <= 0 || Rp < R || C <= 0 || C % 4 || ld % 4 || ld < C) return CPT_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 7)) return CPT_ERR_ALIGN;
    const size_t n = (size_t)Rp * (C / 4);
    split3_rows_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(x, ld, (bf16*)out, R, Rp, C, weight_order);
    return CPT_OK;
}

// ---- fold a LayerNorm (gamma, beta) into the Linear that consumes its output ------------------------
// Wf[n][k] = bf16(gamma[k] * W[n][k]);  colc[n] = sum_k float(Wf[n][k]) (what the MFMA will see);
// cold[n] = sum_k beta[k] * W[n][k] + bias[n].  One wave per output row n.
__global__ __launch_bounds__(256) void fold_ln_weights_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              bf16* __restrict__ Wf, float* __restrict__ colc, float* __restrict__ cold,
                                                              int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float c = 0.f, d = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (bf16)(g[e] * w[e]);
            c += (float)o[e];
            d += b[e] * w[e];
        }
        *reinterpret_cast<bf16x4*>(Wf + (size_t)n * K + k) = o;
    }
    c = wave_sum(c);
    d = wave_sum(d);
    if (lane == 0) { colc[n] = c; cold[n] = d + (bias ? bias[n] : 0.f); }
}

int fold_ln_weights(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* colc,
                    float* cold, int N, int K, hipStream_t s) {
    if (N <= 0 || K <= 0 || K % 4) return CPT_ERR_SHAPE;
    if (!W || !gamma || !beta || !Wf_bf16 || !colc || !cold) return CPT_ERR_NULL;
    fold_ln_weights_kernel<<<dim3((N + 3) / 4), dim3(256), 0, s>>>(W, gamma, beta, bias, (bf16*)Wf_bf16, colc, cold, N, K);
    return CPT_OK;
}

// ---- K-tile-major copy of a bf16 weight: dst[K / 32][N][32] = src[N][K] -------------------------------------------------------
// The fused QKV + attention kernel of qkv_attn3.hip stages K-tiles of 32 (64 bytes per row): out of the row-major weight every
// LDS-DMA request is half a cache line; in this layout a piece's 16 rows are one contiguous KiB.
__global__ __launch_bounds__(256) void retile_k32_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int N, int K) {
    const int cpr = K / 8;                                     // 16-byte chunks per source row
    const size_t total = (size_t)N * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / cpr), c = (int)(i - (size_t)n * cpr);
        const int t = c >> 2, cc = c & 3;                      // K-tile, chunk inside it
        dst[((size_t)t * N + n) * 4 + cc] = src[i];
    }
}
int retile_k32(const void* src, void* dst, int N, int K, hipStream_t s) {
    if (N <= 0 || K <= 0 || K % 32) return CPT_ERR_SHAPE;
    if (!src || !dst) return CPT_ERR_NULL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return CPT_ERR_ALIGN;
    const size_t total = (size_t)N * (K / 8);
    retile_k32_kernel<<<dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s>>>((const uint4*)src, (uint4*)dst, N, K);
    return CPT_OK;
}

// ---- gather rows: out[b] = src[b*L + pos[b]] ---------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int64_t* __restrict__ pos,
                                                          uint4* __restrict__ out, int B, int L, int chunks, const int64_t* __restrict__ seq, int n_seq) {
    const int b = blockIdx.x;
    long p = pos ? pos[b] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    long q = seq ? seq[b] : b;                    // seq: output row b is position pos[b] of sequence seq[b] (label grids)
    q = q < 0 ? 0 : (q >= n_seq ? n_seq - 1 : q);
    const uint4* s = src + ((size_t)q * L + p) * chunks;
    for (int c = threadIdx.x; c < chunks; c += 256) out[(size_t)b * chunks + c] = s[c];
}

// pruned last layer of the training step: the head rows of two tensors in one launch (bf16 ctx rows, fp32 residual rows) ...
__global__ __launch_bounds__(256) void tail_gather2_kernel(const uint4* __restrict__ a, const uint4* __restrict__ bb, const int64_t* __restrict__ pos,
                                                           uint4* __restrict__ oa, uint4* __restrict__ ob, int L, int ca, int cb) {
    const int b = blockIdx.x;
    long p = pos ? pos[b] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const size_t row = (size_t)b * L + p;
    for (int c = threadIdx.x; c < ca + cb; c += 256) {
        if (c < ca) oa[(size_t)b * ca + c] = a[row * ca + c];
        else ob[(size_t)b * cb + (c - ca)] = bb[row * cb + (c - ca)];
    }
}
int tail_gather2(const void* a, const float* bb, const int64_t* pos, void* out_a, float* out_b, int B, int L, int H, hipStream_t s) {
    if (B <= 0 || L <= 0 || H % 8) return CPT_ERR_SHAPE;
    if (!a || !bb || !out_a || !out_b) return CPT_ERR_NULL;
    tail_gather2_kernel<<<dim3(B), dim3(256), 0, s>>>((const uint4*)a, (const uint4*)bb, pos, (uint4*)out_a, (uint4*)out_b, L, H * 2 / 16, H * 4 / 16);
    return CPT_OK;
}
// ... and the inverse for its backward: both [B][L][H] tensors zero except the head rows (every row written once: no fill launch, no atomics)
__global__ __launch_bounds__(256) void tail_scatter2_kernel(const uint4* __restrict__ ar, const uint4* __restrict__ br, const int64_t* __restrict__ pos,
                                                            uint4* __restrict__ za, uint4* __restrict__ zb, int L, int ca, int cb, int rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / L, t = row % L;
    long p = pos ? pos[b] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const bool hit = t == (int)p;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int c = lane; c < ca; c += 64) za[(size_t)row * ca + c] = hit ? ar[(size_t)b * ca + c] : z;
    for (int c = lane; c < cb; c += 64) zb[(size_t)row * cb + c] = hit ? br[(size_t)b * cb + c] : z;
}
int tail_scatter2(const void* a_r, const float* b_r, const int64_t* pos, void* za, float* zb, int B, int L, int H, hipStream_t s) {
    if (B <= 0 || L <= 0 || H % 8) return CPT_ERR_SHAPE;
    if (!a_r || !b_r || !za || !zb) return CPT_ERR_NULL;
    const int rows = B * L;
    tail_scatter2_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, s>>>((const uint4*)a_r, (const uint4*)b_r, pos, (uint4*)za, (uint4*)zb, L, H * 2 / 16, H * 4 / 16, rows);
    return CPT_OK;
}

int gather_rows(const void* src, int dtype, const int64_t* pos, void* out, int B, int L, int H, hipStream_t s, const int64_t* seq, int n_seq) {
    const int esz = dtype == CPT_BF16 ? 2 : 4;
    if (B <= 0 || L <= 0 || (H * esz) % 16) return CPT_ERR_SHAPE;
    gather_rows_kernel<<<dim3(B), dim3(256), 0, s>>>((const uint4*)src, pos, (uint4*)out, B, L, H * esz / 16, seq, seq ? n_seq : B);
    return CPT_OK;
}

// ---- 3-byte residual stream (common.h: r3_encode) -----------------------------------------------
// split: x fp32 [n] -> hi bf16 [n] + lo int8 [n] (the encoder input, once per forward);  merge: the inverse, either for
// every row or for one gathered row per sequence (pos NULL: row 0) -- what the heads read after the last layer.
__global__ __launch_bounds__(256) void r3_split_kernel(const f32x4* __restrict__ x, u32x2_t* __restrict__ hi, unsigned* __restrict__ lo, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        u32x2_t h; unsigned l;
        r3_encode(x[i], h, l);
        hi[i] = h; lo[i] = l;
    }
}
int r3_split(const float* x, void* hi, void* lo, size_t n, hipStream_t s) {
    if (!x || !hi || !lo) return CPT_ERR_NULL;
    if (n % 4 || (((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) & 15)) return CPT_ERR_ALIGN;
    const size_t n4 = n / 4;
    if (n4 == 0) return CPT_OK;
    const int blocks = (int)min((n4 + 255) / 256, (size_t)4096);
    r3_split_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const f32x4*)x, (u32x2_t*)hi, (unsigned*)lo, n4);
    return CPT_OK;
}
__global__ __launch_bounds__(256) void r3_merge_kernel(const u32x2_t* __restrict__ hi, const unsigned* __restrict__ lo, const int64_t* __restrict__ pos,
                                                       f32x4* __restrict__ out, int R, int L, int gather, int q, int src_panel) {
    // one block per output row; gather: row r reads source row r * L + clamp(pos[r]) (pos NULL: 0), else source row r
    // src_panel (round 5): hi / lo are the panel-layout residual stream
    const int r = blockIdx.x;
    size_t src = r;
    if (gather) {
        long p = pos ? pos[r] : 0;
        p = p < 0 ? 0 : (p >= L ? L - 1 : p);
        src = (size_t)r * L + p;
    }
    for (int c = threadIdx.x; c < q; c += 256) {
        const size_t qd = src_panel ? panel_quad(src, c * 4, q * 4) : src * q + c;
        out[(size_t)r * q + c] = r3_decode(hi[qd], lo[qd]);
    }
}
int r3_merge(const void* hi, const void* lo, const int64_t* pos, float* out, int R, int L, int H, int gather, hipStream_t s, int src_panel) {
    if (!hi || !lo || !out) return CPT_ERR_NULL;
    if (R <= 0 || H % 4 || (gather && L <= 0) || (src_panel && H % 16)) return CPT_ERR_SHAPE;
    r3_merge_kernel<<<dim3(R), dim3(256), 0, s>>>((const u32x2_t*)hi, (const unsigned*)lo, pos, (f32x4*)out, R, L, gather, H / 4, src_panel);
    return CPT_OK;
}

// ---- MLM head on the [MASK] rows, bf16 throughput path (round 3) -----------------------------------------------------------
// head_rows_ln3: row b of the output = LayerNorm(decode(hi, lo)[b * L + pos[b]]) as bf16: the gather / merge pass and the LayerNorm pass
// of the head's input rows in one launch (one wave per row).  The same arithmetic as r3_merge + layernorm_rows.
__global__ __launch_bounds__(ROW_THREADS) void head_rows_ln3_kernel(const u32x2_t* __restrict__ hi, const unsigned* __restrict__ lo, const int64_t* __restrict__ pos,
                                                                    const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                                    bf16* __restrict__ out, int R, int L, int H,
                                                                    const void* __restrict__ pf, size_t pf_bytes, int pf_blocks, int src_panel) {
    // The launch has R / 4 blocks of real work (16 at B = 64) on a 256-CU chip: pf_blocks LEADING blocks stream the vocabulary decoder's
    // weight table (47 MB, read once by the GEMM three launches later) into the Infinity Cache meanwhile (common.h prefetch_region)
    __shared__ __attribute__((aligned(16))) unsigned char pf_scratch[4 * 1024];
    if ((int)blockIdx.x < pf_blocks) {
        prefetch_region(pf, pf_bytes, blockIdx.x, pf_blocks, threadIdx.x, ROW_THREADS, pf_scratch);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x - pf_blocks) * (ROW_THREADS / 64) + (threadIdx.x >> 6);
    if (r >= R) return;
    long p = pos ? pos[r] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const size_t src = ((size_t)r * L + p) * (H / 4);
    const int nv = (H + 255) / 256;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (i < nv && c * 4 < H) {
            const size_t qd = src_panel ? panel_quad((size_t)r * L + p, c * 4, H) : src + c;
            v[i] = r3_decode(hi[qd], lo[qd]);
        }
    }
    float mean, rstd;
    ln_stats(v, nv, lane, H, mean, rstd, eps);
    ln_write<bf16>(v, nv, lane, H, mean, rstd, g, bta, nullptr, out + (size_t)r * H);
}
int head_rows_ln3(const void* hi, const void* lo, const int64_t* pos, const float* g, const float* bta, float eps, void* out_bf16, int R, int L, int H, hipStream_t s,
                  const void* pf, size_t pf_bytes, int src_panel) {
    if (!hi || !lo || !g || !bta || !out_bf16) return CPT_ERR_NULL;
    if (R <= 0 || L <= 0 || H <= 0 || H % 4 || H > 256 * MAXV || (src_panel && H % 16)) return CPT_ERR_SHAPE;
    const int nb = (R + 3) / 4;
    const int pfb = (pf && pf_bytes && !((uintptr_t)pf & 15) && nb < 224) ? 224 - (nb & ~7) : 0;      // (a hint: dropped when misaligned or when the rows fill the chip)
    head_rows_ln3_kernel<<<dim3(nb + pfb), dim3(ROW_THREADS), 0, s>>>((const u32x2_t*)hi, (const unsigned*)lo, pos, g, bta, eps, (bf16*)out_bf16, R, L, H,
                                                                       pfb ? pf : nullptr, pf_bytes, pfb, src_panel);
    return CPT_OK;
}
// head_finish: row r of the output = LayerNorm(gelu(sum over the S split-K partial matrices of row r)) as bf16 (the bias rides in partial 0):
// the reduction of the K-split transform GEMM, BertPredictionHeadTransform's GELU and its LayerNorm in one launch.  Partials are added
// in split order (deterministic); GELU = the bf16 path's gelu_fast, as the GEMM epilogue it replaces.
__global__ __launch_bounds__(ROW_THREADS) void head_finish_kernel(const f32x4* __restrict__ part, int S, const float* __restrict__ g, const float* __restrict__ bta,
                                                                  float eps, bf16* __restrict__ out, int R, int H,
                                                                  const void* __restrict__ pf, size_t pf_bytes, int pf_blocks) {
    // round 4: this launch too has R / 4 blocks of real work: pf_blocks leading blocks stream the SECOND part of the decoder's weight table
    // into the Infinity Cache (the first part rides on head_rows_ln3; the whole table there made that launch as long as its 47 MB take)
    __shared__ __attribute__((aligned(16))) unsigned char pf_scratch[4 * 1024];
    if ((int)blockIdx.x < pf_blocks) {
        prefetch_region(pf, pf_bytes, blockIdx.x, pf_blocks, threadIdx.x, ROW_THREADS, pf_scratch);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x - pf_blocks) * (ROW_THREADS / 64) + (threadIdx.x >> 6);
    if (r >= R) return;
    const int nv = (H + 255) / 256, q = H / 4;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (i < nv && c * 4 < H) {
            // all partials of this chunk requested before the first add (a loop over a run-time S issued them one round trip at a time)
            constexpr int SMAX = 8;
            f32x4 pb[SMAX];
#pragma unroll
            for (int k = 0; k < SMAX; ++k) pb[k] = part[((size_t)min(k, S - 1) * R + r) * q + c];
            f32x4 a = pb[0];
#pragma unroll
            for (int k = 1; k < SMAX; ++k)
                if (k < S) { a[0] += pb[k][0]; a[1] += pb[k][1]; a[2] += pb[k][2]; a[3] += pb[k][3]; }
            for (int k = SMAX; k < S; ++k) { const f32x4 b = part[((size_t)k * R + r) * q + c]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
            const f32x2 g0 = gelu_fast2(f32x2{a[0], a[1]}), g1 = gelu_fast2(f32x2{a[2], a[3]});
            v[i] = f32x4{g0[0], g0[1], g1[0], g1[1]};
        }
    }
    float mean, rstd;
    ln_stats(v, nv, lane, H, mean, rstd, eps);
    ln_write<bf16>(v, nv, lane, H, mean, rstd, g, bta, nullptr, out + (size_t)r * H);
}
int head_finish(const float* partials, int S, const float* g, const float* bta, float eps, void* out_bf16, int R, int H, hipStream_t s,
                const void* pf, size_t pf_bytes) {
    if (!partials || !g || !bta || !out_bf16) return CPT_ERR_NULL;
    if (R <= 0 || S <= 0 || H <= 0 || H % 4 || H > 256 * MAXV) return CPT_ERR_SHAPE;
    const int nb = (R + 3) / 4;
    const int pfb = (pf && pf_bytes && !((uintptr_t)pf & 15) && nb < 224) ? 224 - (nb & ~7) : 0;
    head_finish_kernel<<<dim3(nb + pfb), dim3(ROW_THREADS), 0, s>>>((const f32x4*)partials, S, g, bta, eps, (bf16*)out_bf16, R, H, pfb ? pf : nullptr, pf_bytes, pfb);
    return CPT_OK;
}

// ---- last encoder layer on the head's rows only (round 5) -------------------------------------------------------------------
// With only the [MASK] (or [CLS]) rows read after the encoder, everything of the LAST layer behind its attention -- attention output + LayerNorm,
// FFN, LayerNorm -- is row-wise work whose other rows nobody reads: cpt_model_fwd runs it on the R = B head rows (cpt_abi.hip, `tail`).
// tail_rows: row r of the outputs = source row r * L + pos[r] (pos NULL: 0) of
//   ctx_out  [R][H] bf16 : the attention context (panel layout or row-major [M][H])
//   resid_out[R][H] fp32 : LayerNorm(decode(hi, lo); g, bta) -- the residual operand of the attention-output dense layer, i.e. the previous
//                          layer's output LayerNorm, which the folded encoder never materialises
// One wave per row; the same arithmetic as head_rows_ln3.
__global__ __launch_bounds__(ROW_THREADS) void tail_rows_kernel(const u32x2_t* __restrict__ hi, const unsigned* __restrict__ lo, const int64_t* __restrict__ pos,
                                                                const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                                const u32x2_t* __restrict__ ctx, bf16* __restrict__ ctx_out, float* __restrict__ resid_out,
                                                                int R, int L, int H, int src_panel, int ctx_panel,
                                                                const void* __restrict__ pf, size_t pf_bytes, int pf_blocks) {
    __shared__ __attribute__((aligned(16))) unsigned char pf_scratch[4 * 1024];
    if ((int)blockIdx.x < pf_blocks) {
        prefetch_region(pf, pf_bytes, blockIdx.x, pf_blocks, threadIdx.x, ROW_THREADS, pf_scratch);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x - pf_blocks) * (ROW_THREADS / 64) + (threadIdx.x >> 6);
    if (r >= R) return;
    long p = pos ? pos[r] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const size_t row = (size_t)r * L + p;
    const int nv = (H + 255) / 256;
    f32x4 v[MAXV];
    u32x2_t cq[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (i < nv && c * 4 < H) {
            const size_t qd = src_panel ? panel_quad(row, c * 4, H) : row * (H / 4) + c;
            v[i] = r3_decode(hi[qd], lo[qd]);
            cq[i] = ctx[ctx_panel ? panel_quad(row, c * 4, H) : row * (H / 4) + c];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (i < nv && c * 4 < H) reinterpret_cast<u32x2_t*>(ctx_out + (size_t)r * H)[c] = cq[i];
    }
    float mean, rstd;
    ln_stats(v, nv, lane, H, mean, rstd, eps);
    ln_write<float>(v, nv, lane, H, mean, rstd, g, bta, resid_out + (size_t)r * H, (float*)nullptr);
}
int tail_rows(const void* hi, const void* lo, const int64_t* pos, const float* g, const float* bta, float eps, const void* ctx, void* ctx_out, float* resid_out,
              int R, int L, int H, hipStream_t s, int src_panel, int ctx_panel, const void* pf, size_t pf_bytes) {
    if (!hi || !lo || !g || !bta || !ctx || !ctx_out || !resid_out) return CPT_ERR_NULL;
    if (R <= 0 || L <= 0 || H <= 0 || H % 4 || H > 256 * MAXV || ((src_panel || ctx_panel) && H % 16)) return CPT_ERR_SHAPE;
    const int nb = (R + 3) / 4;
    const int pfb = (pf && pf_bytes && !((uintptr_t)pf & 15) && nb < 224) ? 224 - (nb & ~7) : 0;
    tail_rows_kernel<<<dim3(nb + pfb), dim3(ROW_THREADS), 0, s>>>((const u32x2_t*)hi, (const unsigned*)lo, pos, g, bta, eps, (const u32x2_t*)ctx, (bf16*)ctx_out, resid_out,
                                                                   R, L, H, src_panel, ctx_panel, pfb ? pf : nullptr, pf_bytes, pfb);
    return CPT_OK;
}
// tail_finish: row r of the outputs = LayerNorm(sum over the S split-K partial matrices of row r (bias in partial 0) + resid[r]) -- fp32 (the next
// residual operand, optional) and bf16 (the next GEMM operand).  ONE WORKGROUP PER ROW: wave w adds partials [w S / 4, (w + 1) S / 4) in split order with
// every load in flight at once (layernorm_rows_ex walks its x_parts one memory round trip at a time: 24 of them behind the FFN-down), wave 0 adds the
// four wave sums in wave order, the residual, and normalises.  The order depends on S only.
template <int NA>
__global__ __launch_bounds__(256) void tail_finish_kernel(const float* __restrict__ part, int S, size_t stride, const float* __restrict__ resid,
                                                          const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                          float* __restrict__ out_f32, bf16* __restrict__ out_lp, int R, int H,
                                                          const void* __restrict__ pf, size_t pf_bytes, int pf_blocks) {
    __shared__ __attribute__((aligned(16))) float red[3][256 * NA];
    if ((int)blockIdx.x < pf_blocks) {
        prefetch_region(pf, pf_bytes, blockIdx.x, pf_blocks, threadIdx.x, 256, red);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x - pf_blocks;
    const int nv = (H + 255) / 256;
    const int per = (S + 3) / 4, k0 = wave * per, k1 = min(S, k0 + per);
    constexpr int KMAX = 6;                      // partials per wave held in flight (S <= 24)
    f32x4 v[NA], rr[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = f32x4{0, 0, 0, 0};
        if (i < nv && c < H) {
            f32x4 pb[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) pb[k] = *reinterpret_cast<const f32x4*>(part + (size_t)min(k0 + k, S - 1) * stride + (size_t)r * H + c);
            if (wave == 0 && resid) rr[i] = *reinterpret_cast<const f32x4*>(resid + (size_t)r * H + c);
            if (k0 < k1) v[i] = pb[0];
#pragma unroll
            for (int k = 1; k < KMAX; ++k)
                if (k0 + k < k1) { v[i][0] += pb[k][0]; v[i][1] += pb[k][1]; v[i][2] += pb[k][2]; v[i][3] += pb[k][3]; }
            for (int k = k0 + KMAX; k < k1; ++k) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(part + (size_t)k * stride + (size_t)r * H + c);
                v[i][0] += b[0]; v[i][1] += b[1]; v[i][2] += b[2]; v[i][3] += b[3];
            }
            if (wave > 0) *reinterpret_cast<f32x4*>(&red[wave - 1][c]) = v[i];
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
                if ((w + 1) * per < S) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&red[w][c]);
                    v[i][0] += b[0]; v[i][1] += b[1]; v[i][2] += b[2]; v[i][3] += b[3];
                }
            if (resid) { v[i][0] += rr[i][0]; v[i][1] += rr[i][1]; v[i][2] += rr[i][2]; v[i][3] += rr[i][3]; }
        }
    }
    float mean, rstd;
    ln_stats<NA>(v, nv, lane, H, mean, rstd, eps);
    ln_write<bf16, NA>(v, nv, lane, H, mean, rstd, g, bta, out_f32 ? out_f32 + (size_t)r * H : nullptr, out_lp ? out_lp + (size_t)r * H : nullptr);
}
int tail_finish(const float* partials, int S, const float* resid, const float* g, const float* bta, float eps, float* out_f32, void* out_bf16, int R, int H, hipStream_t s,
                const void* pf, size_t pf_bytes) {
    if (!partials || !g || !bta || (!out_bf16 && !out_f32)) return CPT_ERR_NULL;
    if (R <= 0 || S <= 0 || H <= 0 || H % 4 || H > 256 * MAXV) return CPT_ERR_SHAPE;
    const int pfb = (pf && pf_bytes && !((uintptr_t)pf & 15) && R < 224) ? 224 - (R & ~7) : 0;
    const size_t stride = (size_t)R * H;
    if (H <= 768)
        tail_finish_kernel<3><<<dim3(R + pfb), dim3(256), 0, s>>>(partials, S, stride, resid, g, bta, eps, out_f32, (bf16*)out_bf16, R, H, pfb ? pf : nullptr, pf_bytes, pfb);
    else
        tail_finish_kernel<4><<<dim3(R + pfb), dim3(256), 0, s>>>(partials, S, stride, resid, g, bta, eps, out_f32, (bf16*)out_bf16, R, H, pfb ? pf : nullptr, pf_bytes, pfb);
    return CPT_OK;
}
// gelu_parts: out[r][c] (bf16) = gelu(sum over the S split-K partial matrices part[k][r][c]) (bias in partial 0), added in split order: the reduction
// + activation behind the FFN-up GEMM of the tail rows (gelu_fast2, as the FFN-up epilogue of the big launches)
// split_q > 0 (bf16x3 parity mode): rows of split_q quads; the value leaves as the [hi | hi | lo] split copy the next three-term GEMM reads (row pitch 12 split_q)
__global__ __launch_bounds__(256) void gelu_parts_kernel(const f32x4* __restrict__ part, int S, bf16* __restrict__ out, size_t n4,
                                                         const void* __restrict__ pf, size_t pf_bytes, int pf_blocks, int split_q) {
    __shared__ __attribute__((aligned(16))) unsigned char pf_scratch[4 * 1024];
    if ((int)blockIdx.x < pf_blocks) {
        prefetch_region(pf, pf_bytes, blockIdx.x, pf_blocks, threadIdx.x, 256, pf_scratch);
        return;
    }
    const size_t i = (size_t)(blockIdx.x - pf_blocks) * 256 + threadIdx.x;
    if (i >= n4) return;
    constexpr int SMAX = 8;
    f32x4 pb[SMAX];
#pragma unroll
    for (int k = 0; k < SMAX; ++k) pb[k] = part[(size_t)min(k, S - 1) * n4 + i];
    f32x4 a = pb[0];
#pragma unroll
    for (int k = 1; k < SMAX; ++k)
        if (k < S) { a[0] += pb[k][0]; a[1] += pb[k][1]; a[2] += pb[k][2]; a[3] += pb[k][3]; }
    for (int k = SMAX; k < S; ++k) { const f32x4 b = part[(size_t)k * n4 + i]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
    const f32x2 g0 = gelu_fast2(f32x2{a[0], a[1]}), g1 = gelu_fast2(f32x2{a[2], a[3]});
    const float gv[4] = {g0[0], g0[1], g1[0], g1[1]};
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)gv[e];
    if (split_q > 0) {
        bf16x4 lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = (bf16)(gv[e] - (float)o[e]);
        const size_t r = i / (size_t)split_q, c = i - r * (size_t)split_q;
        bf16x4* row = reinterpret_cast<bf16x4*>(out) + r * 3 * (size_t)split_q;
        row[c] = o; row[(size_t)split_q + c] = o; row[2 * (size_t)split_q + c] = lo;
        return;
    }
    reinterpret_cast<bf16x4*>(out)[i] = o;
}
int gelu_parts(const float* partials, int S, void* out_bf16, size_t n, hipStream_t s, const void* pf, size_t pf_bytes, int split_cols) {
    if (!partials || !out_bf16) return CPT_ERR_NULL;
    if (S <= 0 || n == 0 || n % 4 || split_cols < 0 || split_cols % 4 || (split_cols && n % (size_t)split_cols)) return CPT_ERR_SHAPE;
    const size_t n4 = n / 4;
    const size_t nb = (n4 + 255) / 256;
    if (nb > (size_t)1 << 24) return CPT_ERR_SHAPE;
    const int pfb = (pf && pf_bytes && !((uintptr_t)pf & 15) && nb < 224) ? 224 - ((int)nb & ~7) : 0;
    gelu_parts_kernel<<<dim3((unsigned)nb + pfb), dim3(256), 0, s>>>((const f32x4*)partials, S, (bf16*)out_bf16, n4, pfb ? pf : nullptr, pf_bytes, pfb, split_cols / 4);
    return CPT_OK;
}

// ---- decoder on a LIST of vocabulary columns (round 6; SURVEY a11 "or just colour columns") --------------------------------------
// The zero- and few-shot drivers read a handful of colour-token columns of the [B][30522] prediction scores (zeroshot/refcoco_cpt.py:219,
// fewshot/refcoco_cpt.py:272-291, gqa_cpt.py:598-600): out[r][j] = t[r] . W[cols[j]] + b[cols[j]] reads n rows of the decoder table instead of
// streaming all 47 MB of it and writes B x n scores instead of B x V.  One workgroup per row, one column per wave at a time, fp32 accumulation.
// MODE 0: bf16 rows x bf16 table [V][H];  1: fp32 x fp32;  2 (bf16x3): fp32 rows x the split table [V][hi | lo | hi] -- the weight is re-formed as hi + lo.
template <int MODE>
__global__ __launch_bounds__(256) void decoder_cols_kernel(const void* __restrict__ t_, const void* __restrict__ W_, const float* __restrict__ bias,
                                                           const int64_t* __restrict__ cols, int n, float* __restrict__ out, int H, int V) {
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < n; j += 4) {
        long c = cols[j];
        c = c < 0 ? 0 : (c >= V ? V - 1 : c);       // (the host checks the list; never fault)
        float acc = 0.f;
        if constexpr (MODE == 0) {
            const bf16* t = (const bf16*)t_ + (size_t)r * H; const bf16* w = (const bf16*)W_ + (size_t)c * H;
            for (int k = lane * 4; k < H; k += 256) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(t + k), b = *reinterpret_cast<const bf16x4*>(w + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += (float)a[q] * (float)b[q];
            }
        } else if constexpr (MODE == 1) {
            const float* t = (const float*)t_ + (size_t)r * H; const float* w = (const float*)W_ + (size_t)c * H;
            for (int k = lane * 4; k < H; k += 256) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(t + k), b = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += a[q] * b[q];
            }
        } else {
            const float* t = (const float*)t_ + (size_t)r * H; const bf16* w = (const bf16*)W_ + (size_t)c * 3 * H;
            for (int k = lane * 4; k < H; k += 256) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(t + k);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(w + k), lo = *reinterpret_cast<const bf16x4*>(w + H + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += a[q] * ((float)hi[q] + (float)lo[q]);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(size_t)r * n + j] = acc + (bias ? bias[c] : 0.f);
    }
}
int decoder_cols(const void* t, int mode, const void* W, const float* bias, const int64_t* cols, int n, float* out, int R, int H, int V, hipStream_t s) {
    if (R <= 0 || n <= 0 || H <= 0 || H % 4 || V <= 0) return CPT_ERR_SHAPE;
    if (!t || !W || !cols || !out) return CPT_ERR_NULL;
    if (mode == 0) decoder_cols_kernel<0><<<dim3(R), dim3(256), 0, s>>>(t, W, bias, cols, n, out, H, V);
    else if (mode == 1) decoder_cols_kernel<1><<<dim3(R), dim3(256), 0, s>>>(t, W, bias, cols, n, out, H, V);
    else if (mode == 2) decoder_cols_kernel<2><<<dim3(R), dim3(256), 0, s>>>(t, W, bias, cols, n, out, H, V);
    else return CPT_ERR_DTYPE;
    return CPT_OK;
}

// ---- cross entropy over rows (ignore_index = -1) ----------------------------------------------
// the last workgroup to finish (ticket) turns the totals into the mean and a second copy of them: the training forward's loss needs no divide kernel
// and no device-to-device copy behind this launch (round 6).  Every workgroup passes here, ignored rows too.
__device__ __forceinline__ void ce_finish(float* __restrict__ loss, unsigned* __restrict__ ticket, float* __restrict__ mean_out, float* __restrict__ loss_copy) {
    if (!ticket || threadIdx.x != 0) return;
    __threadfence();
    if (atomicAdd(ticket, 1u) != gridDim.x - 1) return;
    __threadfence();
    const float sum = atomicAdd(&loss[0], 0.f), cnt = atomicAdd(&loss[1], 0.f);      // (read at the device's coherence point)
    if (mean_out) mean_out[0] = sum / cnt;       // 0 / 0 = NaN when every row is ignored, as torch's mean over no rows
    if (loss_copy) { loss_copy[0] = sum; loss_copy[1] = cnt; }
    *ticket = 0u;
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss, float* __restrict__ dlogits, int R, int V,
                                                      unsigned* __restrict__ ticket, float* __restrict__ mean_out, float* __restrict__ loss_copy) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const long lab = labels[r];
    const float* x = logits + (size_t)r * V;
    float* d = dlogits ? dlogits + (size_t)r * V : nullptr;
    if (lab < 0 || lab >= V) {            // ignored row: no loss, zero gradient
        if (d) for (int c = threadIdx.x; c < V; c += 256) d[c] = 0.f;
        ce_finish(loss, ticket, mean_out, loss_copy);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, x[c]);
    m = wave_max(m);
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) sum += expf(x[c] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + w] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    const float lse = m + logf(sum);
    if (threadIdx.x == 0) {
        atomicAdd(&loss[0], lse - x[lab]);
        atomicAdd(&loss[1], 1.0f);
    }
    ce_finish(loss, ticket, mean_out, loss_copy);
    if (d) {
        const float inv = 1.0f / sum;
        for (int c = threadIdx.x; c < V; c += 256) d[c] = expf(x[c] - m) * inv - (c == lab ? 1.f : 0.f);
    }
}

// V <= 32 K (every vocabulary of section 8): 1024 threads per row, the row read ONCE into registers (32 values per thread) for the
// maximum, the exponential sum and the gradient -- the 256-thread three-pass form above took 54 us for 32 rows of 30522
constexpr int CE_NV = 32;
__global__ __launch_bounds__(1024) void ce_rows_reg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           float* __restrict__ loss, float* __restrict__ dlogits, int R, int V,
                                                           unsigned* __restrict__ ticket, float* __restrict__ mean_out, float* __restrict__ loss_copy) {
    __shared__ float red[32];
    const int r = blockIdx.x, tid = threadIdx.x;
    const long lab = labels[r];
    const float* x = logits + (size_t)r * V;
    float* d = dlogits ? dlogits + (size_t)r * V : nullptr;
    if (lab < 0 || lab >= V) {            // ignored row: no loss, zero gradient
        if (d) for (int c = tid; c < V; c += 1024) d[c] = 0.f;
        ce_finish(loss, ticket, mean_out, loss_copy);
        return;
    }
    const int lane = tid & 63, w = tid >> 6;
    float xv[CE_NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < CE_NV; ++i) {
        const int c = i * 1024 + tid;
        xv[i] = c < V ? x[c] : -INFINITY;
        m = fmaxf(m, xv[i]);
    }
    m = wave_max(m);
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CE_NV; ++i) { xv[i] = expf(xv[i] - m); sum += xv[i]; }      // exp(-inf) = 0 beyond V
    sum = wave_sum(sum);
    if (lane == 0) red[16 + w] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += red[16 + k];
    if (tid == 0) {
        atomicAdd(&loss[0], m + logf(sum) - x[lab]);
        atomicAdd(&loss[1], 1.0f);
    }
    ce_finish(loss, ticket, mean_out, loss_copy);
    if (d) {
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < CE_NV; ++i) {
            const int c = i * 1024 + tid;
            if (c < V) d[c] = xv[i] * inv - (c == lab ? 1.f : 0.f);
        }
    }
}

int ce_rows(const float* logits, const int64_t* labels, float* loss, float* dlogits, int R, int V, hipStream_t s, unsigned* ticket, float* mean_out, float* loss_copy) {
    if (R <= 0 || V <= 0) return CPT_ERR_SHAPE;
    if (!logits || !labels || !loss) return CPT_ERR_NULL;
    if (V <= CE_NV * 1024) ce_rows_reg_kernel<<<dim3(R), dim3(1024), 0, s>>>(logits, labels, loss, dlogits, R, V, ticket, mean_out, loss_copy);
    else ce_rows_kernel<<<dim3(R), dim3(256), 0, s>>>(logits, labels, loss, dlogits, R, V, ticket, mean_out, loss_copy);
    return CPT_OK;
}

// ---- score extraction on the device (row a15 / SURVEY 8(f).3): only indices return to the host --------------
// torch.argmax semantics: first maximum wins, NaN counts as the maximum.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;                       // a NaN beats any number
    if (!vn && v != bv) return v > bv;
    return i < bi;                                 // equal (or both NaN): the earlier position
}

// One wave per query.  The query's proposal sequences are query_first[q] .. query_first[q+1]-1; sequence s contributes
// the logits at its colour ids color_ids[s][0..C) (entries < 0 are padding) in order, optionally divided by its
// "none" logit; the winner is the argmax position inside that concatenation.
__global__ __launch_bounds__(64) void select_regions_kernel(const float* __restrict__ logits, int V, const int64_t* __restrict__ color_ids,
                                                            int C, const int* __restrict__ query_first, int64_t none_id,
                                                            int divide_by_none, int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int s0 = query_first[q], s1 = query_first[q + 1];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    int base = 0;                                  // concatenation offset of the current sequence (uniform)
    for (int s = s0; s < s1; ++s) {
        int n = 0;                                 // valid colour ids of this sequence (padding is trailing)
        while (n < C && color_ids[(size_t)s * C + n] >= 0) ++n;
        const float* row = logits + (size_t)s * V;
        const float none = divide_by_none ? row[none_id] : 1.0f;
        for (int c = lane; c < n; c += 64) {
            float v = row[color_ids[(size_t)s * C + c]];
            if (divide_by_none) v = v / none;
            if (bi == 0x7fffffff || better(v, base + c, bv, bi)) { bv = v; bi = base + c; }
        }
        base += n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        out_idx[q] = bi == 0x7fffffff ? -1 : bi;
        if (out_score) out_score[q] = bv;
    }
}

int select_regions(const float* logits, int V, const int64_t* color_ids, int C, const int* query_first, int Q,
                   int64_t none_id, int divide_by_none, int64_t* out_idx, float* out_score, hipStream_t s) {
    if (Q <= 0 || V <= 0 || C <= 0 || none_id < 0 || none_id >= V) return CPT_ERR_SHAPE;
    if (!logits || !color_ids || !query_first || !out_idx) return CPT_ERR_NULL;
    select_regions_kernel<<<dim3(Q), dim3(64), 0, s>>>(logits, V, color_ids, C, query_first, none_id, divide_by_none, out_idx, out_score);
    return CPT_OK;
}

// One wave per row: argmax over logits[row][ids[0..n_ids)] (gqa_cpt.py:598-601: answer scored at its first token)
__global__ __launch_bounds__(64) void argmax_columns_kernel(const float* __restrict__ logits, int V, const int64_t* __restrict__ ids, int n_ids,
                                                            int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const float* row = logits + (size_t)r * V;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < n_ids; c += 64) {
        const float v = row[ids[c]];
        if (bi == 0x7fffffff || better(v, c, bv, bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        out_idx[r] = bi == 0x7fffffff ? -1 : bi;
        if (out_val) out_val[r] = bv;
    }
}

int argmax_columns(const float* logits, int V, const int64_t* ids, int n_ids, int R, int64_t* out_idx, float* out_val, hipStream_t s) {
    if (R <= 0 || V <= 0 || n_ids <= 0) return CPT_ERR_SHAPE;
    if (!logits || !ids || !out_idx) return CPT_ERR_NULL;
    argmax_columns_kernel<<<dim3(R), dim3(64), 0, s>>>(logits, V, ids, n_ids, out_idx, out_val);
    return CPT_OK;
}

}  // namespace cpt
