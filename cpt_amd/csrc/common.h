// Shared device helpers for the CPT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "switches.h"

namespace cpt {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// A 16-byte LDS/global chunk: 8 bf16 or 4 f32.
template <typename T> struct Chunk;
template <> struct Chunk<bf16> { static constexpr int N = 8; };
template <> struct Chunk<float> { static constexpr int N = 4; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// exact erf GELU (third-party ACT2FN["gelu"])
__device__ __forceinline__ float gelu_erf(float x) {
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
// erf by Abramowitz-Stegun 7.1.26 (5-term, one v_rcp + one v_exp): |error| < 7e-7 in fp32, i.e. below
// bf16 resolution by four orders of magnitude.  Used by the bf16 throughput path, where libm's
// erff (~60 VALU ops with branches) cost more than the FFN-up GEMM's whole main loop.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = fmaf(-p, __expf(-ax * ax), 1.0f);
    return copysignf(r, x);
}
// erf-GELU of the bf16 throughput path, by Abramowitz-Stegun 7.1.28: erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16,
// |error| <= 3e-7.  With z = |x|/sqrt(2) and the 0.5 of the GELU folded into the polynomial (2^(1/16) scale):
//   gelu(x) = max(x, 0) - |x| * r^16,   r = 1 / P(|x|)
// One v_rcp_f32 and no exp; everything else is fma/mul, written on float pairs so that it compiles to the packed
// fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32).  |gelu error| < 1e-6 absolute over all x (checked against
// scipy erf on 2M points, tests/test_gpu_ops.py checks the kernel): four orders below bf16 resolution.
// (Every multiply-add is an EXPLICIT fma: the same sequence of roundings in every kernel that inlines this, whatever the
// compiler's contraction choices in that context -- rows of a batch must not depend on which tile shape served them.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    f32x2 p = __builtin_elementwise_fma(ax, splat2(5.6212996640e-06f), splat2(5.1055209009e-05f));
    p = __builtin_elementwise_fma(p, ax, splat2(3.9686137011e-05f));
    p = __builtin_elementwise_fma(p, ax, splat2(3.4227392389e-03f));
    p = __builtin_elementwise_fma(p, ax, splat2(2.2076998457e-02f));
    p = __builtin_elementwise_fma(p, ax, splat2(5.2075163037e-02f));
    p = __builtin_elementwise_fma(p, ax, splat2(1.0442737824e+00f));
    f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    r = r * r; r = r * r; r = r * r; r = r * r;
    const f32x2 m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-ax, r, m);
}
// The same arithmetic on NP independent pairs, written step by step across the pairs (round 4): a wave that runs alone on its SIMD
// (gemm_ffn4.hip) cannot hide the latency of gelu_fast2's dependent chain behind another wave's work; NP chains side by side can.
// Per pair bit-identical to gelu_fast2 (the same operations in the same order).
template <int NP>
__device__ __forceinline__ void gelu_fast2_n(f32x2 (&x)[NP]) {
    f32x2 ax[NP], p[NP], r[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) ax[i] = f32x2{fabsf(x[i][0]), fabsf(x[i][1])};
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(ax[i], splat2(5.6212996640e-06f), splat2(5.1055209009e-05f));
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], ax[i], splat2(3.9686137011e-05f));
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], ax[i], splat2(3.4227392389e-03f));
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], ax[i], splat2(2.2076998457e-02f));
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], ax[i], splat2(5.2075163037e-02f));
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], ax[i], splat2(1.0442737824e+00f));
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = f32x2{__builtin_amdgcn_rcpf(p[i][0]), __builtin_amdgcn_rcpf(p[i][1])};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NP; ++i) r[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const f32x2 m = {fmaxf(x[i][0], 0.f), fmaxf(x[i][1], 0.f)};
        x[i] = __builtin_elementwise_fma(-ax[i], r[i], m);
    }
}
// Folded LayerNorm, shared by every kernel that applies it, with explicit fma for the same reason:
//   (mean, rstd) from a row's sums;   x = rstd * (acc - mean * c) + d
__device__ __forceinline__ void ln_mean_rstd(float sum, float sq, float inv_h, float eps, float& mu, float& rs) {
    mu = sum * inv_h;
    rs = rsqrtf(fmaxf(fmaf(-mu, mu, sq * inv_h), 0.f) + eps);
}
__device__ __forceinline__ float ln_fold(float acc, float mu, float rs, float c, float d) { return fmaf(rs, fmaf(-mu, c, acc), d); }
//   LayerNorm applied on the fly to a stored pre-LayerNorm value (the producers' residual):  (r - mean) * rstd * gain + shift
__device__ __forceinline__ float ln_apply(float r, float mu, float rs, float g, float b) { return fmaf((r - mu) * rs, g, b); }
__device__ __forceinline__ float gelu_fast(float x) {
    const f32x2 g = gelu_fast2(f32x2{x, x});
    return g[0];
}
// exact-erf GELU for the fp32 parity path, fast-erf GELU for the bf16 path
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<bf16>(float x) { return gelu_fast(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_for(float x);
// d/dx gelu of the bf16 training path: Phi(x) + x phi(x) with Phi from the same A&S 7.1.28 form as gelu_fast2
// (r^16 already carries the factor 0.5: Phi = 1 - r^16 for x >= 0, r^16 for x < 0) and phi by one v_exp_f32;
// |error| < 1e-6.  The fp32 parity path keeps libm (gelu_erf_grad).
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float ax = fabsf(x);
    float p = fmaf(ax, 5.6212996640e-06f, 5.1055209009e-05f);
    p = fmaf(p, ax, 3.9686137011e-05f);
    p = fmaf(p, ax, 3.4227392389e-03f);
    p = fmaf(p, ax, 2.2076998457e-02f);
    p = fmaf(p, ax, 5.2075163037e-02f);
    p = fmaf(p, ax, 1.0442737824e+00f);
    float r = __builtin_amdgcn_rcpf(p);
    r = r * r; r = r * r; r = r * r; r = r * r;
    const float cdf = x >= 0.f ? 1.0f - r : r;
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return fmaf(x, pdf, cdf);
}
// d/dx gelu_erf
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
template <> __device__ __forceinline__ float gelu_grad_for<float>(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_grad_for<bf16>(float x) { return gelu_grad_fast(x); }

// Butterfly over the 64 lanes, partner distances 32, 16, 8, 4, 2, 1 (round 5: the SAME pairings in the same order as the __shfl_xor loop this
// replaces, so the same bits, but without six dependent ds_bpermute round trips -- the row kernels run one row per wave and sat at the
// latency of their reduce -> reduce -> store chain): v_permlane32_swap for 32, ds_swizzle (no address operand) for 16 and 4, DPP row
// rotate / quad permutes for 8, 2 and 1.
// CONTRACT: 1-D workgroups whose size is a multiple of 64 (lane id = threadIdx.x & 63) with the WHOLE wave active -- the DPP / swizzle
// steps read inactive lanes as 0 and lane_xor32 picks its half by threadIdx.x.  Every call site (23) is a one-row-per-wave or full-wave pass.
__device__ __forceinline__ float lane_xor32(float v) {
    const int i = __float_as_int(v);
    const auto s = __builtin_amdgcn_permlane32_swap(i, i, false, false);      // {[lo, lo], [hi, hi]} of the input's two half-waves
    return (threadIdx.x & 32) ? __int_as_float((int)s[0]) : __int_as_float((int)s[1]);
}
template <int PATTERN> __device__ __forceinline__ float lane_swizzle(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN)); }
template <int CTRL> __device__ __forceinline__ float lane_dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_xor16(float v) { return lane_swizzle<0x401f>(v); }     // bit mode: and 0x1f, or 0, xor 0x10
__device__ __forceinline__ float lane_xor8(float v) { return lane_dpp<0x128>(v); }           // row_ror:8 (rotation by half a row of 16 = xor 8)
__device__ __forceinline__ float lane_xor4(float v) { return lane_swizzle<0x101f>(v); }     // xor 0x04
__device__ __forceinline__ float lane_xor2(float v) { return lane_dpp<0x4e>(v); }            // quad_perm [2, 3, 0, 1]
__device__ __forceinline__ float lane_xor1(float v) { return lane_dpp<0xb1>(v); }            // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor32(v); v += lane_xor16(v); v += lane_xor8(v); v += lane_xor4(v); v += lane_xor2(v); v += lane_xor1(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor32(v)); v = fmaxf(v, lane_xor16(v)); v = fmaxf(v, lane_xor8(v));
    v = fmaxf(v, lane_xor4(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, lane_xor1(v));
    return v;
}

// One MFMA "k-step" on 16-byte operand chunks.
//   bf16: v_mfma_f32_32x32x16_bf16, lane l holds row (l&31), k = 8*(l>>5)+[0..7]
//   f32 : 4 x v_mfma_f32_32x32x2_f32, lane l holds row (l&31); the chunk's 4 floats
//         are consumed one per instruction (k pairs {j, 4+j} across the two half-waves)
__device__ __forceinline__ void mfma_chunk(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_chunk(f32x16& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
template <typename T> struct FragOf;
template <> struct FragOf<bf16> { typedef bf16x8 type; };
template <> struct FragOf<float> { typedef f32x4 type; };

// ---- 3-byte residual stream (bf16 encoder, round 2) ---------------------------------------------------------------
// The LayerNorm producers used to write every pre-LayerNorm sum twice (fp32 for the next residual add + bf16 for the
// next GEMM's A operand: 6 B out, 4 B in per element) and their epilogues run at the memory system's limit, so only fewer
// bytes help.  A value x is now kept as T = its fp32 pattern rounded to the top 24 bits (sign, exponent, 15 mantissa
// bits; round half away, carry runs into the exponent like any IEEE rounding), split into
//   hi = (T + 0x80) >> 8   the bf16 operand of the next GEMM (round-half-away of T: a plain bf16 tensor), and
//   lo = T & 0xff          one signed byte, T - (hi << 8),
// 3 B out + 3 B in per element; readers rebuild x' = ((hi << 8) + (int8)lo) << 8 with |x' - x| <= 2^-17 |x| (the bf16
// operands beside it carry 2^-9).  Inf and NaN survive (the low byte of an infinity is 0).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void r3_encode(const f32x4& v, u32x2_t& hi, unsigned& lo) {
    unsigned t[4], x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ve = v[e];
        const unsigned b = __float_as_uint(ve);
        x[e] = b + 0x80u;            // byte 1 = lo
        t[e] = b + 0x8080u;          // bytes 3:2 = hi
    }
    hi[0] = __builtin_amdgcn_perm(t[1], t[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(t[3], t[2], 0x07060302u);
    const unsigned p01 = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0501u);
    const unsigned p23 = __builtin_amdgcn_perm(x[3], x[2], 0x0c0c0501u);
    lo = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}
__device__ __forceinline__ f32x4 r3_decode(const u32x2_t& hi, unsigned lo) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned h = (e & 1) ? (hi[e >> 1] & 0xffff0000u) : (hi[e >> 1] << 16);
        const int s = __builtin_amdgcn_sbfe((int)lo, 8 * e, 8);
        r[e] = __uint_as_float((unsigned)(s << 8) + h);
    }
    return r;
}
__device__ __forceinline__ float r3_decode1(bf16 hi, signed char lo) {
    unsigned short hb; __builtin_memcpy(&hb, &hi, 2);
    const unsigned h = (unsigned)hb << 16;
    return __uint_as_float(h + (unsigned)((int)lo << 8));
}
__device__ __forceinline__ void r3_encode1(float v, bf16& hi, signed char& lo) {
    const unsigned b = __float_as_uint(v);
    const unsigned short hb = (unsigned short)((b + 0x8080u) >> 16);
    __builtin_memcpy(&hi, &hb, 2);
    lo = (signed char)(((b + 0x80u) >> 8) & 0xffu);
}

// ---- partial row sums of the folded LayerNorm (gemm.hip "EpiX") ---------------------------------------------
// Sum of the partial row sums of `row` in slot order.  Table layout [M][slots][2] with `slots` = parts rounded up to
// an even number (ln_stat_slots): the partial sums of one row are contiguous (64 bytes for hidden = 768), so a reader
// fetches them as 16-byte pairs of slots in one unrolled, branch-free batch.  Slots past `parts` hold garbage and
// are skipped by a select.
template <int MAXQ>   // MAXQ 16-byte loads = 2*MAXQ slots
__device__ __forceinline__ void sum_parts_n(const float* __restrict__ st, int parts, int slots, int row, float& sum, float& sq) {
    const f32x4* base = reinterpret_cast<const f32x4*>(st + (size_t)row * slots * 2);
    const int nq = slots >> 1;
    f32x4 v[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) v[q] = base[min(q, nq - 1)];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const bool u0 = 2 * q < parts, u1 = 2 * q + 1 < parts;      // (select, not multiply: an unused slot may hold NaN)
        sum += u0 ? v[q][0] : 0.f; sq += u0 ? v[q][1] : 0.f;
        sum += u1 ? v[q][2] : 0.f; sq += u1 ? v[q][3] : 0.f;
    }
}
__device__ __forceinline__ void sum_parts(const float* __restrict__ st, int parts, int row, float& sum, float& sq) {
    const int slots = (parts + 1) & ~1;
    sum = 0.f; sq = 0.f;
    if (parts <= 8) sum_parts_n<4>(st, parts, slots, row, sum, sq);
    else if (parts <= 12) sum_parts_n<6>(st, parts, slots, row, sum, sq);
    else {
        for (int p = 0; p < parts; ++p) {
            const float2 v = *reinterpret_cast<const float2*>(st + ((size_t)row * slots + p) * 2);
            sum += v.x; sq += v.y;
        }
    }
}

// ---- the ONE summation order of a LayerNorm producer's partial row sums (round 5) -----------------------------------------------
// A row's 96-column block (one slot of the [M][slots][2] table) is summed the same way by every epilogue, whatever its register layout,
// so that a row's statistics -- and with them every later bit -- do not depend on the kernel that served it:
//   the block's 24 four-column chunks c = 8 j + 2 q + h  (j = 32-column MFMA block 0..2, q = register quad 0..3, h = half-wave 0 / 1: the
//   accumulator layout of the swapped-operand MFMA, lane = row) form 12 chunk pairs (j, qq, h) = chunks (j, 2 qq, h) and (j, 2 qq + 1, h);
//   P(j, qq, h) = the pair's 8 values added one by one (rowsum_chunk_pair; squares by fma);
//   T(qq, h) = (P(0, qq, h) + P(1, qq, h)) + P(2, qq, h);   S(h) = T(0, h) + T(1, h);   S = S(0) + S(1).
// Register-direct epilogues (lane = row, half-wave h) keep T in registers and add the two half-waves; slab epilogues give the four
// lanes of a row the roles (h, qq) = (part & 1, part >> 1) and add across qq, then across h.
__device__ __forceinline__ void rowsum_chunk_pair(const f32x4& a, const f32x4& b, float& sm, float& sq) {
    sm = a[0]; sq = a[0] * a[0];
#pragma unroll
    for (int e = 1; e < 4; ++e) { sm += a[e]; sq = fmaf(a[e], a[e], sq); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm += b[e]; sq = fmaf(b[e], b[e], sq); }
}

// ---- round 3: fragment-major "panel" layout of a GEMM A operand + write-through output stores ------------------------------
// panel[M / 32][K / 16][64][8] bf16 (gemm_prod.hip): the 16-byte unit holding columns [8 c8, 8 c8 + 8) of `row`; k16 = K / 16
__device__ __forceinline__ unsigned panel_unit(int row, int c8, int k16) {
    return (unsigned)(((row >> 5) * k16 + (c8 >> 1)) * 64 + (c8 & 1) * 32 + (row & 31));
}
// The big kernels of the encoder store their outputs WRITE-THROUGH (sc1): a plain store leaves the line dirty in the XCD's L2 and
// the end-of-kernel release writes all of them back before the next kernel may start -- measured on the LayerNorm producers
// (tools/panel_bench.py): 3.7 us between a launch's last workgroup and the next launch's first with plain stores, 1.9 us with
// write-through stores or with no stores at all.  -DCPT_WT=0 restores plain stores (A/B builds).
#ifndef CPT_WT
#define CPT_WT 1
#endif
constexpr int CPT_ST_AUX = CPT_WT ? 16 : 0;          // aux field of the buffer-store builtins: sc1
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Prefetch workgroups (round 3).  Every step streams the 170 MB of bf16 weights once, so each launch finds its weight matrix in HBM,
// and a K loop that keeps ~150 KB per CU in flight runs at (bytes in flight) / (HBM latency): measured on the LayerNorm producers,
// 1076 ticks per K-tile with Infinity-Cache-resident operands, 1560-1830 with either operand in HBM, 1464 in the model
// (tools/panel_bench.py).  A launch whose tiles leave CUs idle (240 tiles on 256 CUs) therefore carries up to 16 extra workgroups
// that do nothing but read the NEXT launch's weight matrix (part `part` of `parts`), which pulls it into the memory-side
// Infinity Cache ~20-50 us before it is needed.  Results are discarded; correctness cannot depend on them.
// The loads are LDS-DMA into a scratch KiB of the (otherwise unused) LDS of the prefetch workgroup: no VGPR destination, so nothing
// the compiler may reuse while a load is in flight (an asm load into a "dead" register returned late and overwrote the next address).
__device__ __forceinline__ void prefetch_region(const void* p, size_t bytes, int part, int parts, int tid, int nthreads, void* lds_scratch) {
    const size_t n16 = bytes >> 4;
    const size_t per = (n16 + parts - 1) / parts;
    const size_t lo = (size_t)part * per;
    const size_t hi = lo + per < n16 ? lo + per : n16;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    auto lds = (__attribute__((address_space(3))) void*)((unsigned char*)lds_scratch + (tid >> 6) * 1024);
    for (size_t i = lo + tid; i < hi; i += nthreads)
        __builtin_amdgcn_global_load_lds(q + i, lds, 16, 0, 0);
}
constexpr int CPT_PREFETCH_WGS = 16;

// 32x32 accumulator element r of lane l sits at (row, col):
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

}  // namespace cpt
