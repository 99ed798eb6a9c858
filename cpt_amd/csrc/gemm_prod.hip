// LayerNorm-PRODUCER GEMMs of the fused bf16 encoder (BertSelfOutput.dense / BertOutput.dense + residual,
// /root/reference/Oscar/oscar/modeling/modeling_bert.py:85-86, :145 -> third-party BertSelfOutput / BertOutput) with the A operand
// read STRAIGHT INTO REGISTERS from a fragment-major ("panel") copy of the activation (round 3):
//   out = A . W^T + bias + LayerNorm_on_the_fly(resid)      in the 3-byte residual form, + partial row sums     (gemm.hip CPT_EPI_LNPROD3)
//
// Why: at M = 7680, N = 768 only 128 x 192 tiles fill the 256 CUs, and such a tile moves 40 KB of operands through the CU's
// LDS-DMA path per 768 MFMA cycles -- the K loop ran at the DMA rate (1460-1530 cycles per K-tile measured in the model,
// DESIGN.md 5h).  Each wave's A rows are private to a row of waves, so they need no LDS at all: the producer of A (the attention
// kernel for ctx, the FFN-up epilogue for h) writes it as MFMA fragments,
//   panel[M / 32][K / 16][64 lanes][8]:   element (row, k) at (((row / 32) (K / 16) + k / 16) 64 + ((k % 16) / 8) 32 + row % 32) 8 + k % 8
// i.e. the 1 KiB that lane l = 32 (k % 16 / 8) + row % 32 of one v_mfma_f32_32x32x16_bf16 A operand reads is contiguous, and a
// K-tile of 64 of a 32-row block is one contiguous 4 KiB.  A wave fetches its fragments with four buffer_load_dwordx4 per K-tile
// (inline asm: hipcc would drain the LDS-DMA queue in front of a plain load's first use), three K-tiles ahead, into a rotating
// set of four register buffers; only W (24 KB per K-tile) rides the LDS ring, four stages deep.  LDS-DMA per K-tile 40 -> 24 KB.
// Same MFMA order over K as the row-major kernel: results are bit-identical to cpt_gemm_ln_prod3 (tested).
#include <algorithm>

#include "common.h"
#include "kernels.h"

// A/B switches (developer builds, tools/build_variant.sh): CPT_PROD_SIDE 0 = side data loaded where the epilogue starts, 1 = requested
// ahead of the first tile, 2 (default) = right behind the first tile; CPT_PROD_AUX0 0 = first residual slice loaded where the epilogue starts
#ifndef CPT_PROD_SIDE
#define CPT_PROD_SIDE 2
#endif
#ifndef CPT_PROD_AUX0
#define CPT_PROD_AUX0 1
#endif
#ifndef CPT_PROD_RAMP
#define CPT_PROD_RAMP 1          // 8-wave shape: prologue requests two tiles, the rest of the ring fills under tile 0 (0: the whole ring in front of the first MFMA, rounds 3-4)
#endif

namespace cpt {
namespace {

constexpr int TM = 128, TN = 192, RB = 128;
constexpr int ST = 4;                          // W ring depth = A register buffers
constexpr int W_SLOT = TN * RB;                // 24 KB
constexpr int GA = 4;                          // A fragment loads per wave per K-tile (one per k-step)
constexpr int RING_BYTES = ST * W_SLOT;        // 96 KB; the epilogue's slabs reuse it
constexpr int GROUP_M = 4;
// Two wave shapes of the same tile (round 4), same MFMA order over K, same epilogue arithmetic -> the same bits:
//   NWV = 8: 4 x 2 waves of 32 x 96 (two per SIMD).  The two waves of a row pair load the SAME A fragments (TA / L1 traffic 2x).
//   NWV = 4: 4 x 1 waves of 32 x 192 (one per SIMD, up to 512 registers): every A fragment is loaded once, half the wave
//            instructions outside the MFMAs (profiles/r04_kloop_vs_hipblaslt.md: against hipBLASLt's 4-wave winner the 8-wave kernel
//            spends the same CYCLES but 1.5x the L1 accesses and 2.5x the waves, and the chip clocks 10 % lower under its power cap).
template <int NWV> struct Shape {
    static constexpr int WN = NWV / 4;                          // waves along N
    static constexpr int NJ = TN / 32 / WN;                     // 32-column blocks per wave (3 / 6)
    static constexpr int WCOLS = NJ * 32;                       // 96 / 192
    static constexpr int GW = TN / 8 / NWV;                     // LDS-DMA pieces per wave per K-tile (3 / 6)
    static constexpr int SIDE = 32 * 8 + 3 * WCOLS * 4;         // per-wave side data: (mean, rstd) of its 32 rows, residual LayerNorm gain / shift and bias of its columns
    static constexpr int LDS_BYTES = RING_BYTES + NWV * SIDE;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ldsoff(int row, int chunk) { return row * RB + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int IMM>
__device__ __forceinline__ void a_load(u32x4& d, unsigned voff, const i32x4& rs, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff), "n"(IMM));
}

// ABL (timing experiments only, cpt_set_tuning key 13; results are garbage for 1-3): 1 = the wn = 1 waves skip their A loads (half the
// A load instructions), 2 = no A loads, 3 = no W LDS-DMA, 4 = refill issued right behind the barrier (correct results)
// SITE (round 4): 0 = short contraction (K <= 1024: the attention-output projection), 1 = long (FFN-down).  Not used by the code: it gives
// the two launches of a layer DISTINCT kernel symbols, so that rocprofv3's per-kernel statistics separate them.
// RP (round 5): the RESIDUAL STREAM travels in the panel layout too -- resid_hi / out_hi as [M / 32][N / 16][64][8] bf16 (exactly the A operand
// panel the next GEMM reads), resid_lo / out_lo as [M / 32][N / 16][64][8] bytes -- and the MFMA operands are swapped (lane = output row,
// register quad = 4 consecutive columns), so the epilogue needs NO LDS slab: a lane's residual columns arrive as one 16-byte + one 8-byte
// load from a contiguous KiB per wave instruction, two v_permlane32_swap put them in the accumulator's quad order, per-row statistics are
// per-lane scalars, the row sums are in-lane chains + one cross-half add, and the stores leave the same way.  No barrier between the K loop
// and the epilogue (nothing reuses the ring).  Same arithmetic per element and the same order in the row sums as the slab epilogue:
// bit-identical results (tests/test_gpu_ops.py).
template <int ABL, int NWV, int SITE, bool RP = false>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void prod3_panel_kernel(
    const bf16* __restrict__ Ap, const bf16* __restrict__ W, int ldw, const float* __restrict__ bias,
    const bf16* __restrict__ resid_hi, const signed char* __restrict__ resid_lo, int ldr,
    const float* __restrict__ st_in, int st_in_parts, const float* __restrict__ g_in, const float* __restrict__ b_in, float eps, float inv_h,
    bf16* __restrict__ out_hi, signed char* __restrict__ out_lo, int ldo, float* __restrict__ st_out, int st_out_slots,
    int M, int N, int K, long long* __restrict__ trace,
    const void* __restrict__ pf0, size_t pf0_bytes, const void* __restrict__ pf1, size_t pf1_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Shape<NWV> SH;
    constexpr int WN = SH::WN, NJ = SH::NJ, WCOLS = SH::WCOLS, GW = SH::GW, SIDE = SH::SIDE;
    // The FIRST gridDim.x - tiles workgroups prefetch the next launches' weights into the Infinity Cache (common.h prefetch_region):
    // dispatched ahead of the tiles they start at once on CUs of their own (behind the tiles they queued for a busy CU and ran
    // as a tail: attn-out 22.3 -> 23.0 us); their count is a multiple of 8, so block id -> XCD is the same for the tiles.
    const int npf = gridDim.x - (M / TM) * (N / TN);
    if ((int)blockIdx.x < npf) {
        if (pf0) prefetch_region(pf0, pf0_bytes, blockIdx.x, npf, threadIdx.x, NWV * 64, smem);
        if (pf1) prefetch_region(pf1, pf1_bytes, blockIdx.x, npf, threadIdx.x, NWV * 64, smem);
        return;
    }
    long long tr0 = 0, tr1 = 0, tr2 = 0, tw0 = 0, tra = 0, trb = 0;
    if (trace) { tr0 = clock64(); tw0 = wall_clock64(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;

    // XCD-first, then GROUP_M row tiles per group (as gemm.hip)
    int m0, n0;
    {
        const int tm = M / TM, tn = N / TN;
        const int nwg = tm * tn, bid = blockIdx.x - npf;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int per_group = GROUP_M * tn;
        const int g = lid / per_group, first_m = g * GROUP_M;
        const int gsz = min(tm - first_m, GROUP_M);
        const int in_g = lid - g * per_group;
        m0 = (first_m + in_g % gsz) * TM;
        n0 = (in_g / gsz) * TN;
    }
    const int nt = K / 64;

    // W: LDS-DMA through a buffer descriptor, source-side XOR swizzle (gemm.hip)
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
    unsigned voffw[GW];
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int r = (i * NWV + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffw[i] = (unsigned)(((size_t)(n0 + r) * ldw + c * 8) * 2);
    }
    auto stage_w = [&](int slot, int t) {
        if (ABL == 3) return;
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            auto lds = (__attribute__((address_space(3))) void*)(smem + slot * W_SLOT + (i * NWV + wave) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds, 16, voffw[i], t * 128, 0, 0);
        }
    };
    // A: this wave's 32-row block of the panel copy; K-tile t = 4 KiB at a_base + 4096 t, k-step ks at + 1024 ks, lane l at + 16 l
    i32x4 rsA;
    {
        const unsigned long long pa = (unsigned long long)(uintptr_t)Ap;
        rsA[0] = (int)(unsigned)(pa & 0xffffffffull);
        rsA[1] = (int)(unsigned)((pa >> 32) & 0xffffull);
        rsA[2] = (int)min((size_t)M * K * 2, (size_t)0x7fffffff);
        rsA[3] = 0x00020000;
    }
    const unsigned voffa = (unsigned)lane * 16u;
    const int a_base = __builtin_amdgcn_readfirstlane(((m0 >> 5) + wm) * (K >> 4) * 1024);
    u32x4 afr[ST][GA];
#define CPT_A_LOAD(BUF, T)                                                       \
    do {                                                                         \
        if (ABL == 2 || (ABL == 1 && wn == 1)) break;                             \
        const int so_ = a_base + (T) * 4096;                                      \
        a_load<0>(afr[BUF][0], voffa, rsA, so_);                                  \
        a_load<1024>(afr[BUF][1], voffa, rsA, so_);                               \
        a_load<2048>(afr[BUF][2], voffa, rsA, so_);                               \
        a_load<3072>(afr[BUF][3], voffa, rsA, so_);                               \
    } while (0)
#define CPT_A_TOUCH(BUF) asm volatile("" : "+v"(afr[BUF][0]), "+v"(afr[BUF][1]), "+v"(afr[BUF][2]), "+v"(afr[BUF][3]))
#define CPT_SB() __builtin_amdgcn_sched_barrier(0)

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Side data of the epilogue (bias, the residual LayerNorm's gain / shift, the partial row sums of this wave's 32 rows): requested in the
    // prologue right BEHIND the first tile (ahead of it their HBM round trip held the first tile back: prologue 5.4 k -> 7.9 k ticks), by
    // inline asm (for a load it can see hipcc waits vmcnt(0) while LDS-DMA is in flight); always NSIDE loads (absent operands read a
    // dummy address) so that the counted waits are constants; parked in the LDS side area behind the ring after iteration 0's wait.
    const int wrow0 = m0 + wm * 32, wcol0 = n0 + wn * WCOLS;
    const bool fold_resid = g_in != nullptr;
    constexpr int NSIDE = 7;
    f32x4 sd_b, sd_g, sd_t, sd_s[4];
    float2 ms_reg = {0.f, 1.f};        // RP: (mean, rstd) of this lane's row of the residual's LayerNorm
#define CPT_SIDE_LOADS()                                                                                                        \
    do {                                                                                                                        \
        const int c4 = wcol0 + min(lane, WCOLS / 4 - 1) * 4;                                                                              \
        const float* dummy = reinterpret_cast<const float*>(W);                                                                  \
        const int slots = (st_in_parts + 1) & ~1, nq = slots >> 1;                                                               \
        const f32x4* base = fold_resid ? reinterpret_cast<const f32x4*>(st_in + (size_t)(wrow0 + (lane & 31)) * slots * 2)      \
                                       : reinterpret_cast<const f32x4*>(dummy);                                                 \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sd_b) : "v"(bias ? bias + c4 : dummy));                            \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sd_g) : "v"(fold_resid ? g_in + c4 : dummy));                      \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sd_t) : "v"(fold_resid ? b_in + c4 : dummy));                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                            \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sd_s[q]) : "v"(base + (fold_resid ? min(q, nq - 1) : 0)));     \
    } while (0)
    // Issue order (per wave), the one every counted wait below assumes:
    //   A(0) W(0) side A(1) A(2) W(1) W(2) W(3) | iteration t: A(t+3) W(t+4) ...
    // Loads complete in issue order, so "tile t+1 has landed" (its A fragments, older: its W pieces) = at most the ops issued
    // after A(t+1) outstanding: W(t+2), A(t+2), W(t+3) = 2 GW + GA -- except in iteration 0, where W(1) is the younger one of
    // tile 1's two parts and only W(2) W(3) may stay in flight.
    // RAMP (round 5, 8 waves): the prologue requests TILES 0 AND 1 ONLY -- 14 instead of 31 KiB-requests per wave in front of the first MFMA (the full fill of
    // the ring took 6.5 k ticks per workgroup: a quarter of the attention-output launch); the side data, A(2), W(2) and W(3) are issued between the k-steps
    // of tile 0, under its MFMAs.  Issue order:  A(0) W(0) A(1) W(1) | side A(2) W(2) W(3) inside iteration 0 | iteration t: A(t+3) W(t+4) ...
    //   first wait (tile 0): younger than W(0) are A(1) W(1) = GA + GW;  iteration 0 (KIND 7): tile 1's younger part is W(1), younger than it side A(2) W(2) W(3)
    //   = NSIDE + GA + 2 GW;  iteration 1 (KIND 6: the side data, older than A(2), has landed -> parked): younger than W(2) are W(3) A(3) W(4) = 2 GW + GA, the
    //   steady-state count, and from there on the sequence IS the steady state.
    constexpr bool RAMP = NWV == 8 && CPT_PROD_SIDE == 2 && ABL == 0 && CPT_PROD_RAMP;
    if (trace) tra = clock64();
    if (RAMP) {
        CPT_A_LOAD(0, 0); CPT_SB(); stage_w(0, 0); CPT_SB();
        CPT_A_LOAD(1, 1); CPT_SB(); stage_w(1, 1); CPT_SB();
    } else {
    if (CPT_PROD_SIDE == 1) { CPT_SIDE_LOADS(); CPT_SB(); }
    CPT_A_LOAD(0, 0); CPT_SB(); stage_w(0, 0); CPT_SB();
    if (CPT_PROD_SIDE == 2) { CPT_SIDE_LOADS(); CPT_SB(); }
    CPT_A_LOAD(1, 1); CPT_A_LOAD(2, 2); CPT_SB();
    stage_w(1, 1); stage_w(2, 2); stage_w(3, 3); CPT_SB();
    }
    if (trace) trb = clock64();

    bf16x8 fb[4][NJ];
    auto ldfrag = [&](int slot, int ks, int pb) {
        const unsigned char* sw = smem + slot * W_SLOT;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[pb][j] = *reinterpret_cast<const bf16x8*>(sw + ldsoff(wn * WCOLS + j * 32 + fr, ks * 2 + fh));
    };
    auto touch = [&](int pb) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(fb[pb][j]));
    };
#define CPT_MMA(BUF, KS)                                                                                           \
    do {                                                                                                           \
        const bf16x8 a_ = __builtin_bit_cast(bf16x8, afr[BUF][KS]);                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                           \
            acc[j_] = RP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[KS][j_], a_, acc[j_], 0, 0, 0)                \
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, fb[KS][j_], acc[j_], 0, 0, 0);               \
    } while (0)

    if (RAMP) wait_vm<GA + GW>();
    else wait_vm<(CPT_PROD_SIDE == 2 ? NSIDE : 0) + 2 * GA + 3 * GW>();            // tile 0 landed: younger than W(0) are the side data, A(1) A(2), W(1) W(2) W(3)
    __builtin_amdgcn_s_barrier();
    CPT_SB();
    CPT_A_TOUCH(0);
    ldfrag(0, 0, 0); ldfrag(0, 1, 1);
    CPT_SB();
    if (trace) tr1 = clock64();

    // the side data has landed with tile 1 (it is older): (mean, rstd) of the wave's rows and the column vectors go to the LDS side area
#define CPT_SIDE_PARK()                                                                                                         \
    do {                                                                                                                        \
        asm volatile("" : "+v"(sd_b), "+v"(sd_g), "+v"(sd_t), "+v"(sd_s[0]), "+v"(sd_s[1]), "+v"(sd_s[2]), "+v"(sd_s[3]));       \
        unsigned char* side_ = smem + RING_BYTES + wave * SIDE;                                                                  \
        if (RP || lane < 32) {                                                                                                   \
            float2 ms = {0.f, 1.f};                                                                                              \
            if (fold_resid) {     /* the arithmetic of sum_parts_n<4> (slot order, unused slots skipped by select) */           \
                float sum = 0.f, sq = 0.f;                                                                                       \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                  \
                    const bool u0 = 2 * q < st_in_parts, u1 = 2 * q + 1 < st_in_parts;                                           \
                    sum += u0 ? sd_s[q][0] : 0.f; sq += u0 ? sd_s[q][1] : 0.f;                                                   \
                    sum += u1 ? sd_s[q][2] : 0.f; sq += u1 ? sd_s[q][3] : 0.f;                                                   \
                }                                                                                                                \
                ln_mean_rstd(sum, sq, inv_h, eps, ms.x, ms.y);                                                                   \
            }                                                                                                                    \
            if (RP) ms_reg = ms; else reinterpret_cast<float2*>(side_)[lane] = ms;                                                \
        }                                                                                                                        \
        if (lane < WCOLS / 4) {                                                                                                  \
            *reinterpret_cast<f32x4*>(side_ + 256 + lane * 16) = fold_resid ? sd_g : f32x4{1.f, 1.f, 1.f, 1.f};                  \
            *reinterpret_cast<f32x4*>(side_ + 256 + WCOLS * 4 + lane * 16) = fold_resid ? sd_t : f32x4{0.f, 0.f, 0.f, 0.f};      \
            *reinterpret_cast<f32x4*>(side_ + 256 + WCOLS * 8 + lane * 16) = bias ? sd_b : f32x4{0.f, 0.f, 0.f, 0.f};            \
        }                                                                                                                        \
    } while (0)
    // residual rows of the first epilogue slice (16 rows x 96 columns per wave: 16-byte hi + 8-byte lo per lane, three per lane), by asm loads
    constexpr int C8 = WCOLS / 8, NIT = 16 * C8 / 64;
    u32x4 ax0h[NIT]; u32x2_t ax0l[NIT];
    // RP: the wave tile's residual = KB consecutive KiB (hi) / half-KiB (lo) units of the panel, one unit per 16 columns; lane l takes bytes 16 l / 8 l
    constexpr int KB = WCOLS / 16;
    u32x4 rph[RP ? KB : 1]; u32x2_t rpl[RP ? KB : 1];
    const size_t rp_unit0 = (size_t)(wrow0 >> 5) * (size_t)(N >> 4) + (size_t)(wcol0 >> 4);        // first unit of this wave's tile (same for resid and out)
#define CPT_AUX0()                                                                                                              \
    do {                                                                                                                        \
        if constexpr (RP) {                                                                                                     \
            const unsigned char* ph_ = reinterpret_cast<const unsigned char*>(resid_hi) + rp_unit0 * 1024 + lane * 16;           \
            const unsigned char* pl_ = reinterpret_cast<const unsigned char*>(resid_lo) + rp_unit0 * 512 + lane * 8;             \
            _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) {                                                                  \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rph[kb]) : "v"(ph_ + kb * 1024));                          \
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rpl[kb]) : "v"(pl_ + kb * 512));                           \
            }                                                                                                                   \
            break;                                                                                                              \
        }                                                                                                                       \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                                    \
            const int idx = it * 64 + lane, rr = idx / C8, c8 = idx % C8;                                                         \
            const size_t off = (size_t)(wrow0 + rr) * ldr + wcol0 + c8 * 8;                                                       \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ax0h[it]) : "v"(resid_hi + off));                               \
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ax0l[it]) : "v"(resid_lo + off));                               \
        }                                                                                                                       \
    } while (0)
    constexpr bool LATE_ = ABL < 4;          // ABL 4: round-3 first version, refill issued right behind the barrier
    // one K-tile: tile t sits in W slot B and A buffer B (B = t % 4).  KIND 0: steady state (issues A(t+3), W(t+4)); 5: the same, t = 0; 1: issues
    // A(t+3) only (t = nt - 4); 2: t = nt - 3; 3: t = nt - 2; 4: last tile (no successor)
#define CPT_TILE(B, KIND, T)                                                                                       \
    do {                                                                                                           \
        constexpr int NB_ = ((B) + 1) & 3, PB_ = ((B) + 3) & 3;                                                     \
        touch(0); CPT_SB(); ldfrag(B, 2, 2); CPT_SB(); CPT_MMA(B, 0); CPT_SB();                                     \
        if ((KIND) == 7) { CPT_SIDE_LOADS(); CPT_SB(); CPT_A_LOAD(2, 2); CPT_SB(); }       /* RAMP: the rest of the ring's first fill, under tile 0 */ \
        touch(1); CPT_SB(); ldfrag(B, 3, 3); CPT_SB(); CPT_MMA(B, 1); CPT_SB();                                     \
        if ((KIND) == 7) { stage_w(2, 2); stage_w(3, 3); CPT_SB(); }                                                \
        touch(2); touch(3); CPT_SB();              /* this wave's reads of tile t are retired before the barrier */ \
        if ((KIND) != 4) {                                                                                         \
            if ((KIND) == 5) wait_vm<2 * GW>(); else if ((KIND) == 7) wait_vm<NSIDE + GA + 2 * GW>();               \
            else if ((KIND) <= 1 || (KIND) == 6) wait_vm<2 * GW + GA>();                                            \
            else if ((KIND) == 2) wait_vm<GW + GA>(); else wait_vm<0>();                                            \
            CPT_SB();                                                                                              \
            __builtin_amdgcn_s_barrier();          /* tile t+1 visible to all waves; nobody still reads tile t */   \
            CPT_SB();                                                                                              \
            CPT_A_TOUCH(NB_);                                                                                      \
            if (((KIND) == 5 && CPT_PROD_SIDE != 0) || (KIND) == 6) { CPT_SIDE_PARK(); CPT_SB(); }                   \
            if ((KIND) == 3 && CPT_PROD_AUX0) { CPT_AUX0(); CPT_SB(); }   /* nothing else is in flight: the first residual slice rides under the last tile */ \
        }                                                                                                          \
        if (LATE_) { CPT_MMA(B, 2); CPT_SB(); }                                                                     \
        if ((KIND) != 4) {                                                                                         \
            if ((KIND) <= 1 || (KIND) >= 5) { CPT_A_LOAD(PB_, (T) + 3); CPT_SB(); }                                 \
            if (!LATE_ && ((KIND) == 0 || (KIND) >= 5)) { stage_w(B, (T) + 4); CPT_SB(); }                          \
            ldfrag(NB_, 0, 0); CPT_SB();                                                                            \
        }                                                                                                          \
        if (!LATE_) { CPT_MMA(B, 2); CPT_SB(); }                                                                    \
        if (LATE_) { CPT_MMA(B, 3); CPT_SB(); }                                                                     \
        if (LATE_ && ((KIND) == 0 || (KIND) >= 5)) { stage_w(B, (T) + 4); CPT_SB(); }                               \
        if ((KIND) != 4) { ldfrag(NB_, 1, 1); CPT_SB(); }                                                           \
        if (!LATE_) { CPT_MMA(B, 3); CPT_SB(); }                                                                    \
    } while (0)

    // ---- NWV = 4 (round 4): one wave per SIMD, so nothing but this wave's own stream can keep the matrix pipe busy: every LDS read,
    // A load and LDS-DMA piece is issued BETWEEN two MFMAs (an MFMA holds the pipe for 32 cycles; the few issue slots of the filler
    // behind it are free), instead of in blocks between the six-MFMA groups (first version: 1227 ticks per K-tile against 768 of MFMA;
    // the blocks' issue time was exposed).  Per K-tile of tile t in slot B:
    //   k-step 0: MFMA j | reads of tile t's k-steps 2 AND 3 (two per MFMA): all of tile t's reads are issued 6+ MFMAs ahead of the barrier
    //   k-step 1: MFMA j | A(t+3) fragment loads; before the last MFMA: lgkmcnt(0) (reads retired) + counted vmcnt (tile t+1 landed);
    //             s_barrier right behind the last MFMA (it runs while the waves meet)
    //   k-step 2: MFMA j | read of tile t+1's k-step 0, fragment j | W(t+4) piece j into slot B (dead since the barrier)
    //   k-step 3: MFMA j | read of tile t+1's k-step 1, fragment j
    // Issue order per wave: A(0) W(0) side A(1) A(2) W(1) W(2) W(3) | iteration t: A(t+3) ... W(t+4).  "Tile t+1 landed" leaves in flight what is
    // younger than A(t+1): W(t+2) A(t+2) W(t+3) A(t+3) = 2 GW + 2 GA (t = 0: younger than W(1): W(2) W(3) A(3) = 2 GW + GA);
    // t = nt-3: W(nt-1) A(nt-1) = GW + GA; t = nt-2: nothing.
#define CPT_MF1(BUF, KS, J) acc[J] = RP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[KS][J], __builtin_bit_cast(bf16x8, afr[BUF][KS]), acc[J], 0, 0, 0) \
                                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[BUF][KS]), fb[KS][J], acc[J], 0, 0, 0)
#define CPT_RD1(SLOT, KS, J) fb[KS][J] = *reinterpret_cast<const bf16x8*>(smem + (SLOT) * W_SLOT + ldsoff((J) * 32 + fr, (KS) * 2 + fh))
#define CPT_AL1(BUF, T, KS) do { if (ABL != 2) a_load<(KS) * 1024>(afr[BUF][KS], voffa, rsA, a_base + (T) * 4096); } while (0)
#define CPT_SW1(SLOT, T, I)                                                                                                      \
    do {                                                                                                                         \
        if (ABL != 3) {                                                                                                          \
            auto lds_ = (__attribute__((address_space(3))) void*)(smem + (SLOT) * W_SLOT + ((I) * NWV + wave) * 1024);           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds_, 16, voffw[I], (T) * 128, 0, 0);                                   \
        }                                                                                                                        \
    } while (0)
#define CPT_K0(B, J) CPT_MF1(B, 0, J); CPT_SB(); CPT_RD1(B, 2, J); CPT_RD1(B, 3, J); CPT_SB();
#define CPT_K1(B, KIND, T, J) CPT_MF1(B, 1, J); CPT_SB(); if ((KIND) <= 1 || (KIND) == 5) { CPT_AL1(((B) + 3) & 3, (T) + 3, J); CPT_SB(); }
#define CPT_K2(B, KIND, T, J) CPT_MF1(B, 2, J); CPT_SB(); if ((KIND) != 4) { CPT_RD1(((B) + 1) & 3, 0, J); } if ((KIND) == 0 || (KIND) == 5) { CPT_SW1(B, (T) + 4, J); } CPT_SB();
#define CPT_K3(B, KIND, J) CPT_MF1(B, 3, J); CPT_SB(); if ((KIND) != 4) { CPT_RD1(((B) + 1) & 3, 1, J); CPT_SB(); }
#define CPT_TILE4(B, KIND, T)                                                                                      \
    do {                                                                                                           \
        constexpr int NB_ = ((B) + 1) & 3;                                                                          \
        CPT_K0(B, 0) CPT_K0(B, 1) CPT_K0(B, 2) CPT_K0(B, 3) CPT_K0(B, 4) CPT_K0(B, 5)                               \
        CPT_K1(B, KIND, T, 0) CPT_K1(B, KIND, T, 1) CPT_K1(B, KIND, T, 2) CPT_K1(B, KIND, T, 3)                     \
        CPT_MF1(B, 1, 4); CPT_SB();                                                                                 \
        if ((KIND) != 4) {                                                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* this wave's reads of tile t are retired */   \
            if ((KIND) == 5) wait_vm<2 * GW + GA>(); else if ((KIND) <= 1) wait_vm<2 * GW + 2 * GA>();              \
            else if ((KIND) == 2) wait_vm<GW + GA>(); else wait_vm<0>();                                            \
            CPT_SB();                                                                                              \
        }                                                                                                          \
        CPT_MF1(B, 1, 5); CPT_SB();                                                                                 \
        if ((KIND) != 4) {                                                                                         \
            __builtin_amdgcn_s_barrier();          /* tile t+1 visible to all waves; nobody still reads tile t */   \
            CPT_SB();                                                                                              \
            CPT_A_TOUCH(NB_);                                                                                      \
            if ((KIND) == 5 && CPT_PROD_SIDE != 0) { CPT_SIDE_PARK(); CPT_SB(); }                                    \
            if ((KIND) == 3 && CPT_PROD_AUX0) { CPT_AUX0(); CPT_SB(); }                                              \
        }                                                                                                          \
        CPT_K2(B, KIND, T, 0) CPT_K2(B, KIND, T, 1) CPT_K2(B, KIND, T, 2) CPT_K2(B, KIND, T, 3) CPT_K2(B, KIND, T, 4) CPT_K2(B, KIND, T, 5) \
        CPT_K3(B, KIND, 0) CPT_K3(B, KIND, 1) CPT_K3(B, KIND, 2) CPT_K3(B, KIND, 3) CPT_K3(B, KIND, 4) CPT_K3(B, KIND, 5)  \
    } while (0)

    if constexpr (NWV == 4) {
    // KIND 2 waits for tile nt-2's successor nt-1: younger than A(nt-1) nothing was issued -> but W(nt-1) is OLDER than A(nt-2)?
    // Order near the end: ... A(nt-3) W(nt-2) | A(nt-2) W(nt-1) | A(nt-1).  Tile t+1 needs A(t+1) and W(t+1):
    //   t = nt-4 (KIND 1): younger than A(nt-3): W(nt-2) A(nt-2) W(nt-1)      = 2 GW + GA   (A(nt-1) is issued after this wait)
    //   t = nt-3 (KIND 2): younger than A(nt-2): W(nt-1) A(nt-1)              = GW + GA
    //   t = nt-2 (KIND 3): younger than A(nt-1): nothing                      = 0
    const int groups = nt / 4 - 1;
    CPT_TILE4(0, 5, 0);
    CPT_TILE4(1, 0, 1);
    CPT_TILE4(2, 0, 2);
    CPT_TILE4(3, 0, 3);
    int t = 4;
    for (int g = 1; g < groups; ++g, t += 4) {
        CPT_TILE4(0, 0, t);
        CPT_TILE4(1, 0, t + 1);
        CPT_TILE4(2, 0, t + 2);
        CPT_TILE4(3, 0, t + 3);
    }
    CPT_TILE4(0, 1, t);
    CPT_TILE4(1, 2, t + 1);
    CPT_TILE4(2, 3, t + 2);
    CPT_TILE4(3, 4, t + 3);
    } else {
    // KIND 2 waits for tile nt-2's successor nt-1: younger than A(nt-1) nothing was issued -> but W(nt-1) is OLDER than A(nt-2)?
    // Order near the end: ... A(nt-3) W(nt-2) | A(nt-2) W(nt-1) | A(nt-1).  Tile t+1 needs A(t+1) and W(t+1):
    //   t = nt-4 (KIND 1): younger than A(nt-3): W(nt-2) A(nt-2) W(nt-1)      = 2 GW + GA   (A(nt-1) is issued after this wait)
    //   t = nt-3 (KIND 2): younger than A(nt-2): W(nt-1) A(nt-1)              = GW + GA
    //   t = nt-2 (KIND 3): younger than A(nt-1): nothing                      = 0
    const int groups = nt / 4 - 1;
    if constexpr (RAMP) { CPT_TILE(0, 7, 0); CPT_TILE(1, 6, 1); }
    else { CPT_TILE(0, 5, 0); CPT_TILE(1, 0, 1); }
    CPT_TILE(2, 0, 2);
    CPT_TILE(3, 0, 3);
    int t = 4;
    for (int g = 1; g < groups; ++g, t += 4) {
        CPT_TILE(0, 0, t);
        CPT_TILE(1, 0, t + 1);
        CPT_TILE(2, 0, t + 2);
        CPT_TILE(3, 0, t + 3);
    }
    CPT_TILE(0, 1, t);
    CPT_TILE(1, 2, t + 1);
    CPT_TILE(2, 3, t + 2);
    CPT_TILE(3, 4, t + 3);
    }
#undef CPT_TILE
#undef CPT_TILE4
#undef CPT_K0
#undef CPT_K1
#undef CPT_K2
#undef CPT_K3
#undef CPT_MF1
#undef CPT_RD1
#undef CPT_AL1
#undef CPT_SW1
#undef CPT_AUX0
#undef CPT_SIDE_PARK
#undef CPT_SIDE_LOADS
#undef CPT_MMA
#undef CPT_A_LOAD
#undef CPT_A_TOUCH
#undef CPT_SB
    if (trace) tr2 = clock64();

    // ---- epilogue: gemm.hip's per-wave slab epilogue (CPT_EPI_LNPROD3, interior-tile instance): the same arithmetic per element and the
    // same order in the row sums (bit-identical outputs), but EIGHT columns per lane in the read-back instead of four: half the
    // vector-memory instructions, 16-byte hi / 8-byte lo accesses, and the stores write-through (common.h CPT_ST_AUX).
    if constexpr (RP) {
        // ---- register-direct epilogue on the panel residual stream (see the kernel's head comment) ----
        unsigned char* side = smem + RING_BYTES + wave * SIDE;
        const float* side_g = reinterpret_cast<const float*>(side + 256);
        const float* side_t = side_g + WCOLS;
        const float* side_b = side_t + WCOLS;
        if (CPT_PROD_SIDE == 0) {     // A/B: side data fetched here
            float2 ms = {0.f, 1.f};
            if (fold_resid) {
                float sum, sq;
                sum_parts(st_in, st_in_parts, wrow0 + fr, sum, sq);
                ln_mean_rstd(sum, sq, inv_h, eps, ms.x, ms.y);
            }
            ms_reg = ms;
            if (lane < WCOLS / 4) {
                float* sg = reinterpret_cast<float*>(side + 256);
                *reinterpret_cast<f32x4*>(sg + lane * 4) = fold_resid ? *reinterpret_cast<const f32x4*>(g_in + wcol0 + lane * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
                *reinterpret_cast<f32x4*>(sg + WCOLS + lane * 4) = fold_resid ? *reinterpret_cast<const f32x4*>(b_in + wcol0 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(sg + 2 * WCOLS + lane * 4) = bias ? *reinterpret_cast<const f32x4*>(bias + wcol0 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!CPT_PROD_AUX0) {
            const unsigned char* ph = reinterpret_cast<const unsigned char*>(resid_hi) + rp_unit0 * 1024 + lane * 16;
            const unsigned char* pl = reinterpret_cast<const unsigned char*>(resid_lo) + rp_unit0 * 512 + lane * 8;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                rph[kb] = *reinterpret_cast<const u32x4*>(ph + kb * 1024);
                rpl[kb] = *reinterpret_cast<const u32x2_t*>(pl + kb * 512);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(rph[kb]), "+v"(rpl[kb]));
        }
        const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)out_hi, 0, (int)min((size_t)M * N * 2, (size_t)0x7fffffff), 0x00020000);
        const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)out_lo, 0, (int)min((size_t)M * N, (size_t)0x7fffffff), 0x00020000);
        const unsigned so_h = (unsigned)__builtin_amdgcn_readfirstlane((int)(rp_unit0 * 1024)), so_l = (unsigned)__builtin_amdgcn_readfirstlane((int)(rp_unit0 * 512));
        const float mu = ms_reg.x, rs = ms_reg.y;
#pragma unroll
        for (int hb = 0; hb < WCOLS / 96; ++hb) {
            float tsm[2], tsq[2];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = hb * 3 + jj;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int kb = j * 2 + kk;
                    // panel unit -> the two accumulator quads of this lane: q = 2 kk (columns 16 kk + 4 h ..) and q = 2 kk + 1 (+ 8)
                    const auto s0 = __builtin_amdgcn_permlane32_swap(rph[kb][0], rph[kb][2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(rph[kb][1], rph[kb][3], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(rpl[kb][0], rpl[kb][1], false, false);
                    const f32x4 rA = r3_decode(u32x2_t{s0[0], s1[0]}, sl[0]);
                    const f32x4 rB = r3_decode(u32x2_t{s0[1], s1[1]}, sl[1]);
                    const int lcA = j * 32 + 16 * kk + 4 * fh, lcB = lcA + 8;
                    const f32x4 bA = *reinterpret_cast<const f32x4*>(side_b + lcA), bB = *reinterpret_cast<const f32x4*>(side_b + lcB);
                    const f32x4 gA = *reinterpret_cast<const f32x4*>(side_g + lcA), gB = *reinterpret_cast<const f32x4*>(side_g + lcB);
                    const f32x4 tA = *reinterpret_cast<const f32x4*>(side_t + lcA), tB = *reinterpret_cast<const f32x4*>(side_t + lcB);
                    f32x4 xA, xB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[j][8 * kk + e] + bA[e];
                        a += ln_apply(rA[e], mu, rs, gA[e], tA[e]);
                        xA[e] = a;
                        float b2 = acc[j][8 * kk + 4 + e] + bB[e];
                        b2 += ln_apply(rB[e], mu, rs, gB[e], tB[e]);
                        xB[e] = b2;
                    }
                    float ps, pq;
                    rowsum_chunk_pair(xA, xB, ps, pq);
                    if (jj == 0) { tsm[kk] = ps; tsq[kk] = pq; } else { tsm[kk] += ps; tsq[kk] += pq; }
                    u32x2_t hA, hB; unsigned lA, lB;
                    r3_encode(xA, hA, lA);
                    r3_encode(xB, hB, lB);
                    const auto t0 = __builtin_amdgcn_permlane32_swap(hA[0], hB[0], false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(hA[1], hB[1], false, false);
                    const auto tl = __builtin_amdgcn_permlane32_swap(lA, lB, false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{t0[0], t1[0], t0[1], t1[1]}, rsH, (unsigned)lane * 16u + kb * 1024u, so_h, CPT_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{tl[0], tl[1]}, rsL, (unsigned)lane * 8u + kb * 512u, so_l, CPT_ST_AUX);
                }
            }
            float sm = tsm[0] + tsm[1], sq = tsq[0] + tsq[1];
            sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
            if (fh == 0)
                *reinterpret_cast<float2*>(st_out + 2 * ((size_t)(wrow0 + fr) * st_out_slots + wcol0 / 96 + hb)) = float2{sm, sq};
        }
    } else {
    __syncthreads();                                  // every wave is done reading the W ring
    constexpr int CPW = WCOLS * 4 + 16, CH = WCOLS / 4, NSL = 2, P = 3;
    static_assert(16 * CPW * NWV <= RING_BYTES, "per-wave slabs must fit in the ring");
    static_assert(C8 == WCOLS / 8 && NIT * 64 == 16 * C8 && (64 * P) % C8 == 0, "read-back fills whole waves; column chunk repeats with period P");
    unsigned char* slab = smem + wave * (16 * CPW);
    unsigned char* side = smem + RING_BYTES + wave * SIDE;
    const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)out_hi, 0, (int)min((size_t)M * ldo * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)out_lo, 0, (int)min((size_t)M * ldo, (size_t)0x7fffffff), 0x00020000);
    float2* side_row = reinterpret_cast<float2*>(side);
    float* side_g = reinterpret_cast<float*>(side + 256);
    float* side_t = side_g + WCOLS;
    float* side_b = side_t + WCOLS;
    if (CPT_PROD_SIDE == 0) {     // A/B: side data fetched here, as gemm.hip does
        if (lane < 32) {
            float2 ms = {0.f, 1.f};
            if (fold_resid) {
                float sum, sq;
                sum_parts(st_in, st_in_parts, wrow0 + lane, sum, sq);
                ln_mean_rstd(sum, sq, inv_h, eps, ms.x, ms.y);
            }
            side_row[lane] = ms;
        }
        if (lane < CH) {
            *reinterpret_cast<f32x4*>(side_g + lane * 4) = fold_resid ? *reinterpret_cast<const f32x4*>(g_in + wcol0 + lane * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
            *reinterpret_cast<f32x4*>(side_t + lane * 4) = fold_resid ? *reinterpret_cast<const f32x4*>(b_in + wcol0 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(side_b + lane * 4) = bias ? *reinterpret_cast<const f32x4*>(bias + wcol0 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 bv[P][2];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int lc = ((q * 64 + lane) % C8) * 8;
        bv[q][0] = *reinterpret_cast<const f32x4*>(side_b + lc);
        bv[q][1] = *reinterpret_cast<const f32x4*>(side_b + lc + 4);
    }
    struct Aux { u32x4 h[NIT]; u32x2_t l[NIT]; };
    auto load_aux = [&](int sl, Aux& a) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 64 + lane, rr = idx / C8, c8 = idx % C8;
            const size_t off = (size_t)(wrow0 + sl * 16 + rr) * ldr + wcol0 + c8 * 8;
            a.h[it] = *reinterpret_cast<const u32x4*>(resid_hi + off);
            a.l[it] = *reinterpret_cast<const u32x2_t*>(resid_lo + off);
        }
    };
    Aux aux_a, aux_b;
    if (CPT_PROD_AUX0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) { asm volatile("" : "+v"(ax0h[it]), "+v"(ax0l[it])); aux_a.h[it] = ax0h[it]; aux_a.l[it] = ax0l[it]; }
    } else load_aux(0, aux_a);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int rr = (r8 & 3) + 8 * (r8 >> 2) + 4 * (lane >> 5);
                *reinterpret_cast<float*>(slab + rr * CPW + (j * 32 + (lane & 31)) * 4) = acc[j][sl * 8 + r8];
            }
        if (sl + 1 < NSL) load_aux(sl + 1, aux_b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        Aux& ax = (sl & 1) ? aux_b : aux_a;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 64 + lane, rr = idx / C8, c8 = idx % C8;
            const int row = wrow0 + sl * 16 + rr, col = wcol0 + c8 * 8;
            const float2 ms = side_row[sl * 16 + rr];
            const float mu = ms.x, rs = ms.y;
            u32x2_t hq[2]; unsigned lq[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {          // the two 4-column quads of this lane's 8 columns
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + rr * CPW + c8 * 32 + hf * 16);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(side_g + c8 * 8 + hf * 4);
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(side_t + c8 * 8 + hf * 4);
                const f32x4 rr4 = r3_decode(u32x2_t{ax.h[it][2 * hf], ax.h[it][2 * hf + 1]}, ax.l[it][hf]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + bv[it % P][hf][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e];
                    x += ln_apply(rr4[e], mu, rs, g4[e], t4[e]);
                    v[e] = x;
                }
                r3_encode(v, hq[hf], lq[hf]);
                *reinterpret_cast<f32x4*>(slab + rr * CPW + c8 * 32 + hf * 16) = v;      // finished values back for the row sums
            }
            const unsigned eo = (unsigned)row * (unsigned)ldo + (unsigned)col;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{hq[0][0], hq[0][1], hq[1][0], hq[1][1]}, rsH, eo * 2u, 0, CPT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{lq[0], lq[1]}, rsL, eo, 0, CPT_ST_AUX);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // partial row sums per 96-COLUMN block (one table slot each), whatever the wave's width: 4 lanes per row, 6 chunks each, then two
        // shuffles -- the same additions in the same order for both wave shapes
        const int r16 = lane >> 2, part = lane & 3;
        const int srow = wrow0 + sl * 16 + r16;
#pragma unroll
        for (int hb = 0; hb < WCOLS / 96; ++hb) {
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {          // the library's one order of a row's partial sums (common.h rowsum_chunk_pair): lane `part` = (h, qq)
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(slab + r16 * CPW + (hb * 24 + j * 8 + 4 * (part >> 1) + (part & 1)) * 16);
                const f32x4 f1 = *reinterpret_cast<const f32x4*>(slab + r16 * CPW + (hb * 24 + j * 8 + 4 * (part >> 1) + 2 + (part & 1)) * 16);
                float ps, pq;
                rowsum_chunk_pair(f0, f1, ps, pq);
                if (j == 0) { sm = ps; sq = pq; } else { sm += ps; sq += pq; }
            }
            sm += __shfl_xor(sm, 2, 64); sq += __shfl_xor(sq, 2, 64);
            sm += __shfl_xor(sm, 1, 64); sq += __shfl_xor(sq, 1, 64);
            if (part == 0)
                *reinterpret_cast<float2*>(st_out + 2 * ((size_t)srow * st_out_slots + wcol0 / 96 + hb)) = float2{sm, sq};
        }
    }
    }
    if (trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* tp = trace + (size_t)(blockIdx.x - npf) * 8;
        tp[0] = tr0; tp[1] = tr1; tp[2] = tr2; tp[3] = tw0; tp[4] = clock64();
        tp[5] = wall_clock64();                                  // (3, 5: the 100 MHz chip-wide counter at start / end)
        tp[6] = tra - tr0;                                      // prologue split: set-up (index math, descriptors) ...
        tp[7] = trb - tra;                                      // ... issue of the prologue's loads (then, up to stamp 1: wait for tile 0)
    }
#endif
}

// row-major [M][ld] <-> panel; one unit = 8 consecutive elements of a row (16 bytes of bf16, 8 bytes of int8)
template <typename U>
__global__ __launch_bounds__(256) void panel_pack_kernel(const U* __restrict__ src, int ld8, U* __restrict__ dst, int M, int K16, int to_panel) {
    const size_t n = (size_t)(M / 32) * K16 * 64;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < n; u += (size_t)gridDim.x * 256) {
        const int l = (int)(u & 63);
        const size_t q = u >> 6;
        const int kk = (int)(q % K16);
        const int rb = (int)(q / K16);
        const size_t rm = (size_t)(rb * 32 + (l & 31)) * ld8 + kk * 2 + (l >> 5);
        if (to_panel) dst[u] = src[rm]; else dst[rm] = src[u];
    }
}

}  // namespace

extern long long* g_gemm_trace;
extern int g_trace_k, g_trace_epi;       // diagnostics: stamp only launches of this K (0: all) / this epilogue id (-1: all; the producers are 11)
CPT_SWITCH(int g_prod_abl, 0);          // timing experiments (cpt_set_tuning key 13), see prod3_panel_kernel
void set_prod_abl(int v) { CPT_SWITCH_SET(g_prod_abl = v); (void)v; }
CPT_SWITCH(int g_prod_waves, 0);        // wave shape of the tile (cpt_set_tuning key 24): 8 = 4 x 2 waves of 32 x 96, 4 = 4 x 1 waves of 32 x 192, 0 = by shape (4 when the tiles run several rounds); same bits
void set_prod_waves(int v) { CPT_SWITCH_SET(g_prod_waves = (v == 4 || v == 0) ? v : 8); (void)v; }

int panel_eligible(int M, int N, int K) { return M > 0 && M % TM == 0 && N > 0 && N % TN == 0 && K >= 512 && K % 256 == 0 && (size_t)M * K * 2 <= (size_t)0x7fffffff; }

int panel_pack(const void* src, int ld, void* dst, int M, int K, int to_panel, hipStream_t s, int elem_bytes) {
    if (M <= 0 || K <= 0 || M % 32 || K % 16 || ld % 8 || (elem_bytes != 2 && elem_bytes != 1)) return CPT_ERR_SHAPE;
    if (!src || !dst) return CPT_ERR_NULL;
    if (((uintptr_t)src | (uintptr_t)dst) & (elem_bytes == 2 ? 15 : 7)) return CPT_ERR_ALIGN;
    const size_t n = (size_t)M * K / 8;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (elem_bytes == 2) panel_pack_kernel<uint4><<<dim3(blocks), dim3(256), 0, s>>>((const uint4*)src, ld / 8, (uint4*)dst, M, K / 16, to_panel);
    else panel_pack_kernel<uint2><<<dim3(blocks), dim3(256), 0, s>>>((const uint2*)src, ld / 8, (uint2*)dst, M, K / 16, to_panel);
    return CPT_OK;
}

int gemm_ln_prod3_panel(const void* A_panel, const void* W, int ldw, const float* bias, const void* resid_hi, const void* resid_lo, int ldr,
                        const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                        void* out_hi, void* out_lo, float* st_out, int ldo, int M, int N, int K, hipStream_t s,
                        const void* pf0, size_t pf0_bytes, const void* pf1, size_t pf1_bytes, int resid_panel) {
    if (!panel_eligible(M, N, K) || ldw % 8 || (st_in && ln_stat_parts(hidden) > 8)) return CPT_ERR_SHAPE;     // (the prologue fetches 8 slots of partial row sums)
    if ((uintptr_t)pf0 & 15) pf0 = nullptr;          // (a prefetch region is a hint: one the 16-byte loads cannot take is dropped, not an error)
    if ((uintptr_t)pf1 & 15) pf1 = nullptr;
    if (!A_panel || !W || !resid_hi || !resid_lo || !out_hi || !out_lo || !st_out) return CPT_ERR_NULL;
    if ((size_t)M * ldo * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;      // 32-bit store offsets
    if (ldo % 8 || ldr % 8 || (((uintptr_t)A_panel | (uintptr_t)W | (uintptr_t)out_hi | (uintptr_t)out_lo | (uintptr_t)resid_hi | (uintptr_t)resid_lo |
                                 (uintptr_t)bias | (uintptr_t)g_in | (uintptr_t)b_in) & 15))
        return CPT_ERR_ALIGN;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        for (const void* k : {(const void*)prod3_panel_kernel<0, 8, 0, true>, (const void*)prod3_panel_kernel<0, 8, 1, true>, (const void*)prod3_panel_kernel<0, 8, 0>, (const void*)prod3_panel_kernel<0, 8, 1>
#ifdef CPT_ABLATION
                              , (const void*)prod3_panel_kernel<1, 8, 1>, (const void*)prod3_panel_kernel<2, 8, 1>, (const void*)prod3_panel_kernel<3, 8, 1>, (const void*)prod3_panel_kernel<4, 8, 1>
#endif
             }) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, Shape<8>::LDS_BYTES);
            if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        }
        for (const void* k : {(const void*)prod3_panel_kernel<0, 4, 0, true>, (const void*)prod3_panel_kernel<0, 4, 1, true>, (const void*)prod3_panel_kernel<0, 4, 0>, (const void*)prod3_panel_kernel<0, 4, 1>
#ifdef CPT_ABLATION
                              , (const void*)prod3_panel_kernel<2, 4, 1>, (const void*)prod3_panel_kernel<3, 4, 1>
#endif
             }) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, Shape<4>::LDS_BYTES);
            if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        }
        attr_done = true;
    }
    const int ntile = (M / TM) * (N / TN);
    // extra workgroups on the CUs the tiles leave idle: they read the next launches' weights (prefetch_region)
    const int npf = ((pf0 && pf0_bytes) || (pf1 && pf1_bytes)) ? (std::max(0, std::min(CPT_PREFETCH_WGS, 256 - ntile)) & ~7) : 0;
    if (!npf) { pf0 = nullptr; pf1 = nullptr; }
    const int nwg = ntile + npf;
#define CPT_LAUNCH(ABL, NWV) CPT_LAUNCH3(ABL, NWV, 1)
#define CPT_LAUNCH3(ABL, NWV, SITE) CPT_LAUNCH4(ABL, NWV, SITE, false)
#define CPT_LAUNCH4(ABL, NWV, SITE, RP) prod3_panel_kernel<ABL, NWV, SITE, RP><<<dim3(nwg), dim3(NWV * 64), Shape<NWV>::LDS_BYTES, s>>>(                                  \
        (const bf16*)A_panel, (const bf16*)W, ldw, bias, (const bf16*)resid_hi, (const signed char*)resid_lo, ldr, st_in, ln_stat_parts(hidden), g_in, b_in, \
        eps, 1.0f / (float)hidden, (bf16*)out_hi, (signed char*)out_lo, ldo, st_out, ln_stat_slots(N), M, N, K, ((g_trace_epi < 0 || g_trace_epi == 11) && (g_trace_k == 0 || g_trace_k == K)) ? g_gemm_trace : nullptr, \
        pf0, pf0_bytes, pf1, pf1_bytes)
    // by default 4 waves when the tiles run several rounds (GQA shape: 1680 tiles; measured attn-out 1.66 -> 1.56, FFN-down 3.79 -> 3.63 ms
    // per step at B = 256, L = 210).  In one round (the bench shape, 240 tiles) the 4-wave FFN-down launch is 2.5 us shorter by its own
    // brackets, but the step is not (1.706 vs 1.696 ms, 1.741 vs 1.739 on a second box): the chip sits at its power cap and the denser
    // launch takes clock from its neighbours (profiles/r04_kloop_vs_hipblaslt.md), so the 8-wave shape stays there.
    const int prod_waves = call_override().prod_waves >= 0 ? call_override().prod_waves : g_prod_waves;      // (per-call test override, kernels.h)
    if (resid_panel) {      // residual stream in the panel layout, register-direct epilogue (round 5)
        if ((size_t)M * N * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;
        const bool w4 = prod_waves == 4 || (prod_waves == 0 && ntile > 256);
        if (w4) { if (K <= 1024) CPT_LAUNCH4(0, 4, 0, true); else CPT_LAUNCH4(0, 4, 1, true); }
        else { if (K <= 1024) CPT_LAUNCH4(0, 8, 0, true); else CPT_LAUNCH4(0, 8, 1, true); }
    } else
    if (prod_waves == 4 || (prod_waves == 0 && ntile > 256)) {
        switch (g_prod_abl) {
#ifdef CPT_ABLATION
            case 2: CPT_LAUNCH(2, 4); break;
            case 3: CPT_LAUNCH(3, 4); break;
#endif
            default: if (K <= 1024) CPT_LAUNCH3(0, 4, 0); else CPT_LAUNCH3(0, 4, 1); break;
        }
    } else
    switch (g_prod_abl) {
#ifdef CPT_ABLATION
        case 1: CPT_LAUNCH(1, 8); break;
        case 2: CPT_LAUNCH(2, 8); break;
        case 3: CPT_LAUNCH(3, 8); break;
        case 4: CPT_LAUNCH(4, 8); break;
#endif
        default: if (K <= 1024) CPT_LAUNCH3(0, 8, 0); else CPT_LAUNCH3(0, 8, 1); break;
    }
#undef CPT_LAUNCH4
#undef CPT_LAUNCH3
#undef CPT_LAUNCH
    return CPT_OK;
}

}  // namespace cpt
