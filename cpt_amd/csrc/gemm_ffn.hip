// FFN-up (BertIntermediate: dense + GELU, /root/reference/Oscar/oscar/modeling/modeling_bert.py:144 -> third-party
// BertIntermediate) of the bf16 fused encoder as ONE kernel with its epilogue software-pipelined under the K loop:
//   out[M][N] = gelu( rstd[m] * (A.Wf^T - mean[m] * colc[n]) + cold[n] )        (LayerNorm folded, see gemm.hip / DESIGN.md 5c)
//
// Why a dedicated kernel (measured on MI355X, tools/gemm_cu_timeline.py): with 128x192 tiles the K loop is bound by the
// CU's LDS-DMA path (40 KB of operands per 768 MFMA cycles) and the GELU epilogue (about as many VALU issue cycles as
// the MFMAs take matrix cycles) ran after it, un-overlapped.  Here one workgroup owns a 384 x 256 output tile and computes
// it as TWO passes of 192 x 256 (56 KB of operands per 1536 MFMA cycles): pass 1's K loop carries pass 0's finished
// accumulators in a second register set and retires them -- LayerNorm fold, GELU, bf16 pack, 16-byte stores -- one
// register-quad pair per K-tile, in the issue slots the matrix pipe leaves free.  Only the second half's epilogue is
// exposed.  240 workgroups at M = 7680, N = 3072: one per CU, one round.
//
// Layout: 8 waves as 2 (M) x 4 (N), wave tile 96 x 64 = 3 x 2 MFMA 32x32x16 blocks with SWAPPED operands, so the
// accumulators are transposed (lane = output row, register quad = 4 consecutive columns; gemm.hip "direct epilogue").
// LDS rings (LDS-DMA, source-side XOR swizzle, counted vmcnt): three stages of 56 KB do not fit beside the side data, and with two the single tile in flight exposed the
// whole L2/MALL latency every K-tile (2720 cycles per K-tile measured, MFMA time 1536).  So the operands ride rings of
// different depth: A half-tiles (24 KB) 2 deep, W tiles (32 KB) 3 deep -- the W tile of K-tile t+3 is requested two
// iterations ahead, only the 24 KB A half-tile of t+2 has to arrive within one.  Plus the tile's 256-column c/d vectors
// and its 384 rows' (mean, rstd): 149 KB.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace cpt {
namespace {

constexpr int RB = 128;                       // bytes per operand-tile row (64 bf16)
constexpr int HM = 192, TN = 256, TM = 2 * HM;
constexpr int NWV = 8;
constexpr int A_SLOT = HM * RB, W_SLOT = TN * RB;            // 24 KB, 32 KB
constexpr int NA = 2, NW_ = 3;                                // ring depths: A half-tiles 2, W tiles 3 (see "rings" below)
constexpr int GA = HM / 8 / NWV, GW = TN / 8 / NWV;           // LDS-DMA pieces (1 KiB = 8 rows) per wave per tile: 3 of A, 4 of W
constexpr int MI = 3, NJ = 2;
constexpr int W_RING = NA * A_SLOT, SIDE_C = W_RING + NW_ * W_SLOT, SIDE_D = SIDE_C + TN * 4, SIDE_ST = SIDE_D + TN * 4,
              LDS_BYTES = SIDE_ST + TM * 8;                   // 48 + 96 + 5 KB
static_assert(GA * 8 * NWV == HM && GW * 8 * NWV == TN && LDS_BYTES <= 160 * 1024, "tile must split evenly over the waves and fit the LDS");

__device__ __forceinline__ int ldsoff(int row, int chunk) { return row * RB + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// PANEL: out is the fragment-major panel copy [M / 32][N / 16][64][8] the FFN-down producer reads (gemm_prod.hip), rows padded to a
// multiple of 32 by the caller; every 16-byte store of a half-wave then lands in one contiguous 512 bytes.
// PIPE: pass 0's epilogue retires under pass 1's K loop (round 2).  Measured in the model in round 3 (ablation build): the pieces ADD
// 17 k ticks to pass 1 (46.0 k vs 29.1 k without them) while the same work takes 8.8 k ticks when it runs exposed behind the loop
// (VALU issued between dependent MFMAs costs more than its issue slots, and both waves of a SIMD pay it) -- PIPE = false runs both
// epilogues behind pass 1.
#ifndef CPT_FFN_PIPE
#define CPT_FFN_PIPE 0
#endif
#ifndef CPT_FFN_EPI_PAIR
#define CPT_FFN_EPI_PAIR 1          // round 4: exposed epilogues by quad PAIRS (epi_pair); 0 = one quad at a time (A/B builds)
#endif
// GELU = false (round 3): the same kernel as a plain LayerNorm-consumer GEMM -- the stand-alone QKV projection of the shapes whose attention
// does not fuse (L > 128: GQA 165 + 45, VCR 165 + 100), which ran on the 384 x 192 pipe kernel at 2/3 of this kernel's rate per CU.
// APANEL (round 5): A is the residual stream's hi part in the panel layout [M / 32][K / 16][64][8] (gemm_prod.hip RP; lda ignored, M % 32 == 0): every
// LDS-DMA piece is one contiguous KiB unit of the panel and keeps that layout in LDS, so the A fragment reads are lane-linear (no swizzle).
// (First version: the row-major LDS image gathered from the panel -- eight 128-byte runs per piece, sixteen cache lines per 16 lanes: +2.8 us per launch.)
template <int NT, bool LATE = true, bool PANEL = false, bool GELU = true, bool PIPE = (CPT_FFN_PIPE != 0), bool APANEL = false>       // K-tiles per pass: K = 64 NT; LATE: the refill DMA is issued behind the first k-step after the barrier
__global__ __launch_bounds__(512, 2) void ffn_up_2pass_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                                                              bf16* __restrict__ out, int ldo, int M, int N,
                                                              const float* __restrict__ st_in, int st_parts, const float* __restrict__ colc,
                                                              const float* __restrict__ cold, float eps, float inv_h,
                                                              long long* __restrict__ trace_arg, int abl_arg,
                                                              const void* __restrict__ pf, size_t pf_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the first gridDim.x - tiles workgroups prefetch the next launch's weights into the Infinity Cache (common.h prefetch_region; gemm_prod.hip)
    const int npf = gridDim.x - ((M + TM - 1) / TM) * (N / TN);
    if ((int)blockIdx.x < npf) {
        if (pf) prefetch_region(pf, pf_bytes, blockIdx.x, npf, threadIdx.x, 512, smem);
        return;
    }
#ifdef CPT_ABLATION
    long long* __restrict__ trace = trace_arg;
    const int abl = abl_arg;          // diagnostic build: 1 no operand DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue under pass 1
#else
    constexpr long long* trace = nullptr;
    constexpr int abl = 0;
    (void)trace_arg; (void)abl_arg;
#endif
    long long tr0 = 0, tr1 = 0, tr2 = 0, trp = 0, s_wait = 0, s_bar = 0, s_dma = 0, ta = 0, tb = 0, tc = 0;
    if (trace) tr0 = clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;

    // XCD-first tile order, 4 row tiles per group (as gemm.hip)
    int m0, n0;
    {
        const int tm = (M + TM - 1) / TM, tn = N / TN;
        const int nwg = tm * tn, bid = blockIdx.x - npf;      // (prefetch workgroups come first)
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int gm = 4, per_group = gm * tn;
        const int g = lid / per_group, first_m = g * gm;
        const int gsz = min(tm - first_m, gm);
        const int in_g = lid - g * per_group;
        m0 = (first_m + in_g % gsz) * TM;
        n0 = (in_g / gsz) * TN;
    }

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)(APANEL ? ((M + 31) & ~31) : M) * (APANEL ? 64 * NT : lda) * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)min((size_t)((M + 31) & ~31) * (PANEL ? N : ldo) * 2, (size_t)0x7fffffff), 0x00020000);
    const int rbase = wave * 8 + (lane >> 3);
    const unsigned c16 = (unsigned)(((lane & 7) ^ ((rbase >> 1) & 7)) * 16);
    // K-tile `it` of the 2 NT-tile stream (pass = it / NT): A half-tile into A slot `sa`, W tile into W slot `sw`
    auto stage_a = [&](int sa, int it) {
        if (abl & 1) return;
        const int pass = it >= NT ? 1 : 0;
        const int soff = (it - pass * NT) * (APANEL ? 4096 : 64 * 2);
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            auto lds = (__attribute__((address_space(3))) void*)(smem + sa * A_SLOT + (i * NWV + wave) * 1024);
            if constexpr (APANEL) {
                // piece p = i NWV + wave IS one 1 KiB unit of the panel: row block p / 4 of the half-tile, k16 unit p % 4 of the K-tile; the LDS image keeps
                // the unit as it is (lane l = row l % 32, k half l / 32: the MFMA operand order), so the fragment reads below are lane-linear
                const int p = i * NWV + wave;
                const unsigned rb = (unsigned)min((m0 + pass * HM) / 32 + (p >> 2), (M >> 5) - 1);
                const unsigned vo = (rb * (unsigned)(NT * 4) + (unsigned)(p & 3)) * 1024u + (unsigned)lane * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lds, 16, vo, soff, 0, 0);
            } else {
            const unsigned vo = (unsigned)min(m0 + pass * HM + rbase + i * NWV * 8, M - 1) * (unsigned)(lda * 2) + c16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lds, 16, vo, soff, 0, 0);
            }
        }
    };
    auto stage_w = [&](int sw, int it) {
        if (abl & 1) return;
        const int pass = it >= NT ? 1 : 0;
        const int soff = (it - pass * NT) * 64 * 2;
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            auto lds = (__attribute__((address_space(3))) void*)(smem + W_RING + sw * W_SLOT + (i * NWV + wave) * 1024);
            const unsigned vo = (unsigned)min(n0 + rbase + i * NWV * 8, N - 1) * (unsigned)(ldw * 2) + c16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds, 16, vo, soff, 0, 0);
        }
    };

    // Side data (the tile's column vectors and its 384 rows' partial LayerNorm sums): requested by INLINE ASM right behind the first
    // K-tile's DMA and parked in LDS after the first barrier (round 3).  Round 2 loaded them with plain loads AHEAD of the DMA -- hipcc
    // waits vmcnt(0) for a plain load's first use while LDS-DMA is in flight, so they had to come first, and the wave sat through
    // their round trip before it issued the first operand tile: two round trips in the prologue instead of one.
    float* side_c = reinterpret_cast<float*>(smem + SIDE_C);
    float* side_d = reinterpret_cast<float*>(smem + SIDE_D);
    float2* side_st = reinterpret_cast<float2*>(smem + SIDE_ST);
    constexpr int NSIDE = 7;                       // asm loads per lane: one column-vector quad + six quads of partial sums (always issued)
    f32x4 cd, sv[6];
    stage_a(0, 0); stage_w(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    {
        const float* cdp = tid < 64 ? colc + n0 + tid * 4 : cold + n0 + (min(tid, 127) - 64) * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cd) : "v"(cdp));
        const int row = min(m0 + min(tid, TM - 1), M - 1);
        const int slots = (st_parts + 1) & ~1, nq = slots >> 1;
        const f32x4* base = reinterpret_cast<const f32x4*>(st_in + (size_t)row * slots * 2);
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sv[q]) : "v"(base + min(q, nq - 1)));   // up to 12 slots (hidden <= 1152)
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_a(1, 1); stage_w(1, 1);
    stage_w(2, 2);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc0[MI][NJ], acc1[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // Fragments: A blocks single-buffered, each re-read for the next k-step right after the MFMAs that used it have been
    // issued; the two W blocks (used by all three A blocks) double-buffered and read a whole k-step ahead.  With two waves
    // per SIMD and 192 accumulator registers this is what keeps LDS latency off the matrix pipe (one buffer for all five:
    // 2840 cycles per K-tile measured; MFMA time 1536).
    bf16x8 fa[MI], fb[2][NJ];
    // fragment addresses: every row this lane reads is (multiple of 16) + fr, so the swizzle term (row >> 1) & 7 is the lane
    // constant (fr >> 1) & 7: one base per operand, the blocks 32 rows apart sit at immediate offsets
    const unsigned abase = (unsigned)(wm * 96 + fr) * RB, wbase = (unsigned)W_RING + (unsigned)(wn * 64 + fr) * RB;
    const unsigned sx = (unsigned)((fr >> 1) & 7);
    auto coff = [&](int ks) { return (((unsigned)(ks * 2 + fh)) ^ sx) << 4; };
    auto rd_a = [&](int i, int sa, int ks) {
        if (abl & 2) return;
        if constexpr (APANEL) fa[i] = *reinterpret_cast<const bf16x8*>(smem + (unsigned)(sa * A_SLOT) + (unsigned)(((wm * 3 + i) * 4 + ks) * 1024) + (unsigned)lane * 16u);
        else fa[i] = *reinterpret_cast<const bf16x8*>(smem + abase + coff(ks) + (unsigned)(sa * A_SLOT) + i * 32 * RB);
    };
    auto rd_b = [&](int b, int sw, int ks) {
        if (abl & 2) return;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[b][j] = *reinterpret_cast<const bf16x8*>(smem + wbase + coff(ks) + (unsigned)(sw * W_SLOT) + j * 32 * RB);
    };
#define CPT_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef CPT_STAGGER
#define CPT_STAGGER 0
#endif
    // experiment: the second wave of every SIMD (waves 4-7) leaves each K-tile barrier CPT_STAGGER x 64 clocks late
#define CPT_STAG() do { if (CPT_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(CPT_STAGGER); } while (0)
#define CPT_T(x) asm volatile("" : "+v"(x))
    // one k-step on accumulator set ACC with W buffer B; the fragments of k-step (NSLOT, NKS) are fetched meanwhile if MORE
#define CPT_MF(D, X, Y) do { if (!(abl & 4)) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Y, D, 0, 0, 0); } while (0)
#define CPT_KSTEP(ACC, B, NSA, NSW, NKS, MORE)                                                                              \
    do {                                                                                                                  \
        if (MORE) { rd_b((B) ^ 1, NSW, NKS); CPT_SB(); }                                                                  \
        CPT_T(fa[0]); CPT_T(fb[B][0]); CPT_T(fb[B][1]); CPT_SB();                                                         \
        CPT_MF(ACC[0][0], fb[B][0], fa[0]);                         \
        CPT_MF(ACC[0][1], fb[B][1], fa[0]); CPT_SB();               \
        if (MORE) { rd_a(0, NSA, NKS); CPT_SB(); }                                                                        \
        CPT_T(fa[1]); CPT_SB();                                                                                           \
        CPT_MF(ACC[1][0], fb[B][0], fa[1]);                         \
        CPT_MF(ACC[1][1], fb[B][1], fa[1]); CPT_SB();               \
        if (MORE) { rd_a(1, NSA, NKS); CPT_SB(); }                                                                        \
        CPT_T(fa[2]); CPT_SB();                                                                                           \
        CPT_MF(ACC[2][0], fb[B][0], fa[2]);                         \
        CPT_MF(ACC[2][1], fb[B][1], fa[2]); CPT_SB();               \
        if (MORE) { rd_a(2, NSA, NKS); CPT_SB(); }                                                                        \
    } while (0)
    // all fragment reads of the current tile retired (before the barrier that frees its slot)
#define CPT_RETIRE(B) do { CPT_T(fa[0]); CPT_T(fa[1]); CPT_T(fa[2]); CPT_T(fb[B][0]); CPT_T(fb[B][1]); CPT_SB(); } while (0)

    // ---- epilogue pieces: pair p (0..11) of a pass = row block i = p / 4, column block j = (p / 2) % 2, quad pair gp = p % 2
    auto epi_quad = [&](const f32x16& a, int pass, int p, int h) -> u32x2 {
        const int i = p >> 2, j = (p >> 1) & 1, g = 2 * (p & 1) + h;
        // The side data is read with inline-asm ds_read: for an LDS load it can see, hipcc waits vmcnt(0) first whenever an
        // LDS-DMA is in flight (it cannot tell the side area from the ring), which drained the operand pipeline once per
        // K-tile (measured: +1070 cycles per K-tile of pass 1).  The wait below also retires the older fragment reads.
        const int lc = wn * 64 + j * 32 + 8 * g + 4 * fh;
        float2 ms; f32x4 c4, d4;
#ifndef CPT_SIDE_MODE
#define CPT_SIDE_MODE 0
#endif
#if CPT_SIDE_MODE == 0
        ms = side_st[pass * HM + wm * 96 + i * 32 + fr];
        c4 = *reinterpret_cast<const f32x4*>(side_c + lc);
        d4 = *reinterpret_cast<const f32x4*>(side_d + lc);
#else
#if CPT_SIDE_MODE == 2
        wait_vm<0>();
#endif
        asm volatile("ds_read_b64 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ms), "=&v"(c4), "=&v"(d4)
                     : "v"((unsigned)(SIDE_ST + (pass * HM + wm * 96 + i * 32 + fr) * 8)), "v"((unsigned)(SIDE_C + lc * 4)), "v"((unsigned)(SIDE_D + lc * 4))
                     : "memory");
#if CPT_SIDE_MODE == 3
        {   // debug: the same values through compiler-visible loads; mismatches counted in the last trace slot
            const float2 ms2 = side_st[pass * HM + wm * 96 + i * 32 + fr];
            const f32x4 c42 = *reinterpret_cast<const f32x4*>(side_c + lc);
            const f32x4 d42 = *reinterpret_cast<const f32x4*>(side_d + lc);
            bool bad = __float_as_uint(ms2.x) != __float_as_uint(ms.x) || __float_as_uint(ms2.y) != __float_as_uint(ms.y);
            for (int e = 0; e < 4; ++e) bad = bad || __float_as_uint(c42[e]) != __float_as_uint(c4[e]) || __float_as_uint(d42[e]) != __float_as_uint(d4[e]);
            if (bad && trace) atomicAdd(reinterpret_cast<unsigned long long*>(trace) + 4096 * 8 - 1, 1ull);
        }
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);
        f32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ln_fold(a[4 * g + e], ms.x, ms.y, c4[e], d4[e]);
        if constexpr (!GELU) {
            bf16x4 p4 = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
            return __builtin_bit_cast(u32x2, p4);
        }
        const f32x2 g0 = gelu_fast2(f32x2{x[0], x[1]}), g1 = gelu_fast2(f32x2{x[2], x[3]});
        bf16x4 p4 = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
        return __builtin_bit_cast(u32x2, p4);
    };
    // round 4: both quads of a pair through the fold and the GELU side by side (four independent pairs per wave: the dependent chain of
    // gelu_fast2 no longer waits on itself); same operations per element: same bits
    auto epi_pair = [&](const f32x16& a, int pass, int p, u32x2& k0, u32x2& k1) {
        const int i = p >> 2, j = (p >> 1) & 1;
        const float2 ms = side_st[pass * HM + wm * 96 + i * 32 + fr];
        f32x2 x[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = 2 * (p & 1) + h;
            const int lc = wn * 64 + j * 32 + 8 * g + 4 * fh;
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(side_c + lc);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(side_d + lc);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[2 * h + (e >> 1)][e & 1] = ln_fold(a[4 * g + e], ms.x, ms.y, c4[e], d4[e]);
        }
        if constexpr (GELU) gelu_fast2_n<4>(x);
        const bf16x4 p0 = {(bf16)x[0][0], (bf16)x[0][1], (bf16)x[1][0], (bf16)x[1][1]};
        const bf16x4 p1 = {(bf16)x[2][0], (bf16)x[2][1], (bf16)x[3][0], (bf16)x[3][1]};
        k0 = __builtin_bit_cast(u32x2, p0); k1 = __builtin_bit_cast(u32x2, p1);
    };
    auto epi_store = [&](int pass, int p, u32x2 k0, u32x2 k1) {
        const int i = p >> 2, j = (p >> 1) & 1, gp = p & 1;
        // half-wave exchange: lanes 0-31 get columns [16 gp, 16 gp + 8) of the block, lanes 32-63 the next 8 (guide T21)
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(k0[0], k1[0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(k0[1], k1[1], false, false);
        const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
        const int row = m0 + pass * HM + wm * 96 + i * 32 + fr;
        const int col = n0 + wn * 64 + j * 32 + 16 * gp + 8 * fh;
        if (row < M) {      // write-through (common.h CPT_ST_AUX): nothing of the 47 MB is left dirty in L2 for the end-of-kernel release
            if constexpr (PANEL) __builtin_amdgcn_raw_buffer_store_b128(w, rsO, panel_unit(row, col >> 3, N >> 4) * 16u, 0, CPT_ST_AUX);
            else *reinterpret_cast<u32x4*>(out + (size_t)row * ldo + col) = w;      // row-major: 32 rows x 32 B per instruction -- write-through of partial lines measured SLOWER (51.2 vs 48 us)
        }
    };

    // issued so far per wave: A0 W0 | side | A1 W1 W2 = 3 4 | 7 | 3 4 4; tile 0 is complete when at most 7 + 11 are outstanding
    wait_vm<NSIDE + GA + 2 * GW>();
    __builtin_amdgcn_s_barrier();                               // tile 0 visible to every wave
    if (trace) tr1 = clock64();
    CPT_SB();
    rd_b(0, 0, 0); rd_a(0, 0, 0); rd_a(1, 0, 0); rd_a(2, 0, 0);
    CPT_SB();
    // the side data (issued with tile 0) has landed too: reduce the partial sums in slot order (bit-reproducible, gemm.hip sum_parts) and
    // park everything in LDS; its first readers (pass 0's epilogue pieces under pass 1) sit behind many K-tile barriers
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(cd), "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]) : "n"(GA + 2 * GW) : "memory");
    CPT_SB();
    if (tid < 64) *reinterpret_cast<f32x4*>(side_c + tid * 4) = cd;
    else if (tid < 128) *reinterpret_cast<f32x4*>(side_d + (tid - 64) * 4) = cd;
    if (tid < TM) {
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const bool u0 = 2 * q < st_parts, u1 = 2 * q + 1 < st_parts;
            sum += u0 ? sv[q][0] : 0.f; sq += u0 ? sv[q][1] : 0.f;
            sum += u1 ? sv[q][2] : 0.f; sq += u1 ? sv[q][3] : 0.f;
        }
        float mu, rs;
        ln_mean_rstd(sum, sq, inv_h, eps, mu, rs);
        side_st[tid] = float2{mu, rs};
    }
    CPT_SB();

    // Steady state of iteration `it` (K-tile it of 2 NT): after its last fragment reads are retired, wait for tile it+1
    // (A half-tile it+1 was issued one iteration ago together with, and ahead of, W tile it+2, which may stay in flight: 4
    // pieces, plus the epilogue store of this iteration), barrier, then request A(it+2) into the A slot just freed and
    // W(it+3) into the W slot just freed.  k-steps 0..3 of a tile use W fragment buffers 0, 1, 0, 1.
    // ---- pass 0: rows [m0, m0 + 192)
    int sa = 0, sw = 0;
    for (int kt = 0; kt < NT; ++kt) {
        const int sw1 = sw == NW_ - 1 ? 0 : sw + 1;
        CPT_KSTEP(acc0, 0, sa, sw, 1, true);
        CPT_KSTEP(acc0, 1, sa, sw, 2, true);
        CPT_KSTEP(acc0, 0, sa, sw, 3, true);
        if (trace) ta = clock64();
        CPT_RETIRE(1);                             // this wave's reads of the tile are retired
        wait_vm<GW>();                             // tile kt+1 landed (W(kt+2) may still be in flight)
        CPT_SB();
        if (trace) tb = clock64();
        __builtin_amdgcn_s_barrier();              // ... for every wave; nobody still reads tile kt's slots
        CPT_STAG();
        CPT_SB();
        if (trace) tc = clock64();
        if (!LATE) {
            stage_a(sa, kt + 2);                   // (kt + 3 < 2 NT always holds in pass 0: NT >= 3)
            stage_w(sw, kt + 3);
            CPT_SB();
        }
        if (trace) { const long long td = clock64(); s_wait += tb - ta; s_bar += tc - tb; s_dma += td - tc; }
        CPT_KSTEP(acc0, 1, sa ^ 1, sw1, 0, true);
        if (LATE) {                                // refill behind this wave's MFMAs: the eight waves' DMA issue does not sit between
            stage_a(sa, kt + 2);                   // the barrier release and the first MFMA (same issue order: A(kt+2), then W(kt+3))
            stage_w(sw, kt + 3);
            CPT_SB();
        }
        sa ^= 1; sw = sw1;
    }
    if (trace) trp = clock64();
    // ---- pass 1: rows [m0 + 192, m0 + 384); pass 0's accumulators retire under it, one quad pair per K-tile
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        constexpr int NP = MI * NJ * 2;            // 12 pairs
        const int it = NT + kt;
        const int a_s = it & 1, w_s = it % NW_, w_s1 = (it + 1) % NW_;
        const bool epi = PIPE && kt < NP && !(abl & 8);
        const bool more = kt + 1 < NT;
        u32x2 k0 = {0, 0}, k1 = {0, 0};
        CPT_KSTEP(acc1, 0, a_s, w_s, 1, true);
        if (epi) { k0 = epi_quad(acc0[(kt % NP) >> 2][((kt % NP) >> 1) & 1], 0, kt % NP, 0); CPT_SB(); }
        CPT_KSTEP(acc1, 1, a_s, w_s, 2, true);
        if (epi) { k1 = epi_quad(acc0[(kt % NP) >> 2][((kt % NP) >> 1) & 1], 0, kt % NP, 1); CPT_SB(); }
        CPT_KSTEP(acc1, 0, a_s, w_s, 3, true);
        if (epi) { epi_store(0, kt % NP, k0, k1); CPT_SB(); }
        if (trace) ta = clock64();
        CPT_RETIRE(1);
        if (more) {
            // Loads younger than A(it+1): the 4 pieces of W(it+2), if it exists.  Loads complete in issue order, stores do NOT
            // keep order with loads on gfx950 (one vmcnt for both): the epilogue's stores must not be counted as "younger
            // ops that may stay in flight" -- a store that completes early would let an older DMA piece slip through.  At
            // most GW outstanding => every load older than W(it+2) has landed, whatever the stores did.
            if (kt + 2 < NT) wait_vm<GW>(); else wait_vm<0>();
            CPT_SB();
            if (trace) tb = clock64();
            __builtin_amdgcn_s_barrier();
            CPT_STAG();
            CPT_SB();
            if (trace) tc = clock64();
            if (!LATE) {
                if (kt + 2 < NT) stage_a(a_s, it + 2);
                if (kt + 3 < NT) stage_w(w_s, it + 3);
                CPT_SB();
            }
            if (trace) { const long long td = clock64(); s_wait += tb - ta; s_bar += tc - tb; s_dma += td - tc; }
        }
        CPT_KSTEP(acc1, 1, a_s ^ 1, w_s1, 0, more);
        if (LATE && more) {
            if (kt + 2 < NT) stage_a(a_s, it + 2);
            if (kt + 3 < NT) stage_w(w_s, it + 3);
            CPT_SB();
        }
    }
    if (trace) tr2 = clock64();
    // pairs of pass 0 that did not fit under pass 1 (NT < 12), then pass 1's own epilogue (exposed)
#pragma unroll
    for (int p = PIPE ? NT : 0; p < MI * NJ * 2; ++p) {
        u32x2 k0, k1;
        if (CPT_FFN_EPI_PAIR) epi_pair(acc0[p >> 2][(p >> 1) & 1], 0, p, k0, k1);
        else { k0 = epi_quad(acc0[p >> 2][(p >> 1) & 1], 0, p, 0); k1 = epi_quad(acc0[p >> 2][(p >> 1) & 1], 0, p, 1); }
        epi_store(0, p, k0, k1);
    }
#pragma unroll
    for (int p = 0; p < MI * NJ * 2; ++p) {
        u32x2 k0, k1;
        if (CPT_FFN_EPI_PAIR) epi_pair(acc1[p >> 2][(p >> 1) & 1], 1, p, k0, k1);
        else { k0 = epi_quad(acc1[p >> 2][(p >> 1) & 1], 1, p, 0); k1 = epi_quad(acc1[p >> 2][(p >> 1) & 1], 1, p, 1); }
        epi_store(1, p, k0, k1);
    }
#undef CPT_SB
#undef CPT_T
#undef CPT_KSTEP
#undef CPT_RETIRE
    if (trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* t = trace + (size_t)(blockIdx.x - npf) * 8;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = trp; t[4] = clock64();
        t[5] = s_wait; t[6] = s_bar; t[7] = s_dma;         // wave 0: sums over the K-tiles (fragment retire + vmcnt wait, barrier, DMA issue)
    }
#endif
}

template <int NT, bool LATE = true, bool PANEL = false, bool GELU = true, bool APANEL = false>
int launch_2pass(const bf16* A, int lda, const bf16* W, int ldw, bf16* out, int ldo, int M, int N, const float* st_in, int st_parts,
                 const float* colc, const float* cold, float eps, float inv_h, long long* trace, int abl, hipStream_t s,
                 const void* pf = nullptr, size_t pf_bytes = 0) {
    auto kern = ffn_up_2pass_kernel<NT, LATE, PANEL, GELU, (CPT_FFN_PIPE != 0), APANEL>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    const int ntile = ((M + TM - 1) / TM) * (N / TN);
    const int npf = (pf && pf_bytes) ? (std::max(0, std::min(CPT_PREFETCH_WGS, 256 - ntile)) & ~7) : 0;
    if (!npf) pf = nullptr;
    kern<<<dim3(ntile + npf), dim3(512), LDS_BYTES, s>>>(A, lda, W, ldw, out, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, trace, abl, pf, pf_bytes);
    return CPT_OK;
}

}  // namespace

CPT_SWITCH(int g_ffn_2pass_min_tiles, 192);      // (cpt_set_tuning key 16: experiments with half batches on two streams)
void set_ffn_2pass_min_tiles(int v) { CPT_SWITCH_SET(g_ffn_2pass_min_tiles = v); (void)v; }
CPT_SWITCH(int g_ffn_dma_late, 1);      // A/B switch (cpt_set_tuning key 12): 0 = refill DMA issued right behind the barrier (round 2)
void set_ffn_dma_late(int v) { CPT_SWITCH_SET(g_ffn_dma_late = v); (void)v; }

// shapes the two-pass kernel can run / shapes it is the better choice for
int ffn_up_2pass_legal(int M, int N, int K) { return (K == 768 || K == 1024) && N % TN == 0 && M >= TM; }
int ffn_up_2pass_preferred(int M, int N, int K) {
    if (!ffn_up_2pass_legal(M, N, K)) return 0;
    // one workgroup per CU per round: worth it only when the 384 x 256 tiles fill most of the 256 CUs (B = 64 x L = 120:
    // 240 tiles); below that the 128 x 192 two-per-CU shape has 4x the workgroups
    return (long)((M + TM - 1) / TM) * (N / TN) >= g_ffn_2pass_min_tiles;
}

int gemm_ffn_up_2pass(const void* A, int lda, const void* Wf, int ldw, const float* st_in, int st_parts, const float* colc, const float* cold,
                      float eps, int hidden, void* out, int ldo, int M, int N, int K, void* trace, int abl, hipStream_t s, int out_panel,
                      const void* pf, size_t pf_bytes, int gelu, int a_panel) {
    if (!ffn_up_2pass_legal(M, N, K)) return CPT_ERR_SHAPE;
    if (a_panel && (!gelu || !out_panel || M % 32)) return CPT_ERR_SHAPE;          // the panel-in form exists for the fused encoder's FFN-up only
    if (!gelu) {        // plain LayerNorm-consumer form (QKV projection): row-major output only
        if (out_panel) return CPT_ERR_SHAPE;
        if (lda % 8 || ldw % 8 || ldo % 8 || (((uintptr_t)A | (uintptr_t)Wf | (uintptr_t)out | (uintptr_t)colc | (uintptr_t)cold) & 15)) return CPT_ERR_ALIGN;
        if (K == 768) return launch_2pass<12, true, false, false>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s);
        return launch_2pass<16, true, false, false>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s);
    }
    if ((size_t)((M + 31) & ~31) * (out_panel ? N : ldo) * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;      // 32-bit store offsets
    if (out_panel) {
        if (N % 16) return CPT_ERR_SHAPE;
        if ((uintptr_t)pf & 15) pf = nullptr;          // (a prefetch region is a hint: dropped when the 16-byte loads cannot take it)
        if (a_panel) {
            if (K == 768) return launch_2pass<12, true, true, true, true>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s, pf, pf_bytes);
            return launch_2pass<16, true, true, true, true>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s, pf, pf_bytes);
        }
        if (K == 768) return launch_2pass<12, true, true>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s, pf, pf_bytes);
        return launch_2pass<16, true, true>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, 1.0f / (float)hidden, (long long*)trace, abl, s, pf, pf_bytes);
    }
    if (lda % 8 || ldw % 8 || ldo % 8 || (((uintptr_t)A | (uintptr_t)Wf | (uintptr_t)out | (uintptr_t)colc | (uintptr_t)cold) & 15)) return CPT_ERR_ALIGN;
    const float inv_h = 1.0f / (float)hidden;
#ifdef CPT_ABLATION
    if (K == 768 && !g_ffn_dma_late) return launch_2pass<12, false>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, (long long*)trace, abl, s);
#endif
    if (K == 768) return launch_2pass<12>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, (long long*)trace, abl, s);
    return launch_2pass<16>((const bf16*)A, lda, (const bf16*)Wf, ldw, (bf16*)out, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, (long long*)trace, abl, s);
}

}  // namespace cpt
