// Fused QKV projection + self-attention of the bf16 encoder, ONE workgroup per (sequence, group of three heads)  [round 3].
//   q|k|v = x W^T + b (or the LayerNorm-folded form)           /root/reference/Oscar/oscar/modeling/modeling_bert.py:38-40
//   ctx   = softmax(q k^T / 8 + mask) v, per head               modeling_bert.py:42-67
// Replaces the (sequence, head) form of gemm.hip (CPT_EPI_ATTN*, 768 workgroups of 128 x 192 at B = 64): that grid ran as
// 1.5 rounds of two co-resident workgroups per CU (the third unit of every CU ran alone for 43 % of the launch,
// profiles/r02_kernel_timelines_128x192.txt) and streamed every sequence's 128 activation rows twelve times.  Here
//   * the grid is B x heads / 3 = 256 workgroups at B = 64: one per CU, one round;
//   * a workgroup computes the 128 x 576 tile [tokens of the sequence] x [Q | K | V of three heads]: the activation rows are
//     streamed four times instead of twelve (operand bytes through LDS-DMA 277 MB instead of 377 MB per launch);
//   * 12 waves as 2 (M) x 6 (N), wave tile 64 x 96 = 2 x 3 MFMA 32x32x16 blocks: 5 fragment reads per 6 MFMAs (the 8-wave
//     128 x 192 tile reads 4 per 3), which takes the loop's LDS time (fragment reads + LDS-DMA writes: 1660 LDS cycles per
//     64-deep K step) below its MFMA time (2304 cycles);
//   * the three heads' attention runs as 12 (head, 32-query block) tasks, one per wave, out of Q / K / V tiles that never
//     leave the CU (attn_core.h, the same code the stand-alone attention kernel runs -> the same bits).
// A 128 x 576 x 64 operand stage is 88 KB, so the ring holds K-tiles of 32 (64-byte rows, 44 KB per stage, three stages);
// the rows are XOR-swizzled on the SOURCE side of the LDS-DMA (chunk ^ ((row >> 2) & 3)) so that every ds_read_b128 fragment
// read is bank-conflict free, as in gemm.hip.  MFMA operands are swapped (lane = token, register quad = 4 consecutive
// output columns), the k-steps of a row's dot product run in the same order as in every other GEMM of the library: the
// outputs are bit-identical to the (sequence, head) kernel and to the two-kernel path (tests/test_gpu_model.py).
#include <type_traits>

#include "common.h"
#include "attn_core.h"
#include "kernels.h"

namespace cpt {
namespace {

constexpr int Q3_TM = 128, Q3_TN = 576;               // tokens x (3 heads x (Q | K | V) x 64)
constexpr int Q3_RB = 64;                             // bytes per operand row per stage (32 bf16)
constexpr int Q3_ROWS = Q3_TM + Q3_TN;                // 704 rows = 44 pieces of 16 rows
constexpr int Q3_STAGE = Q3_ROWS * Q3_RB;             // 45056 B
constexpr int Q3_STAGES = 3;
constexpr int Q3_NW = 12, Q3_G = 4;                   // 12 waves x 4 pieces = 48: the last piece of waves 8..11 is a dummy
constexpr int Q3_DUMMY = Q3_STAGES * Q3_STAGE;        // 4 KB landing area of the dummy pieces
constexpr int Q3_TILE = 128 * 128;                    // one Q / K / V tile: 128 tokens x 64 bf16 (V in the swizzled 128-byte form)
constexpr int Q3_MASK = 9 * Q3_TILE;                  // additive key mask [128] fp32
constexpr int Q3_CD = Q3_MASK + 512;                  // the tile's 576 column constants: c (LayerNorm fold) then d (shift / bias), fp32
constexpr int Q3_ST = Q3_CD + 2 * Q3_TN * 4;          // the tile rows' partial LayerNorm sums (<= 8 slots = 64 B per row): 8 KB
constexpr int Q3_LDS = Q3_ST + 8192;                  // 160768 B (ring + dummy area = 139264)
static_assert(Q3_DUMMY + 4096 <= Q3_LDS, "ring must fit");

template <int N> __device__ __forceinline__ void q3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// 64-byte rows, four rows per 256-byte bank row: a 16-lane ds_read_b128 group touches 16 distinct 16-byte slots
__device__ __forceinline__ int q3_off(int row, int chunk) { return row * Q3_RB + ((chunk ^ ((row >> 2) & 3)) << 4); }

struct Q3Args {
    const bf16* A; int lda;                // activations [M][K] (bf16 hi part of the residual stream)
    int a_panel;                           // round 5: A is the residual stream's hi part in the panel layout [M rounded up to 32 / 32][K / 16][64][8] (gemm_prod.hip RP), lda ignored
    const bf16* W; int ldw;                // fused [3 * heads * 64][K] weight (LayerNorm gain folded in when st_in != NULL)
    int w_tiled;                           // 1: W is the K-tile-major copy [K / 32][3 * heads * 64][32] (cpt_retile_k32): a piece's 16 rows are 1 KiB contiguous
    const float* bias;                     // [3 * heads * 64] (plain form) or NULL
    const float* st_in; int st_parts;      // folded LayerNorm: partial row sums of A's rows
    const float* colc; const float* cold;
    float eps, inv_h;
    const int64_t* mask;                   // [B][L] or NULL
    bf16* ctx; int ldo;                    // PANEL instantiation: ctx is the fragment-major panel copy [M / 32][hidden / 16][64][8] (gemm_prod.hip), ldo ignored
    int M, K, L, heads;
    long long* trace;                      // diagnostics (cpt_debug_gemm_trace): 8 int64 per workgroup, shader-clock stamps + HW ids
};

// ABL (diagnostic instantiations, cpt_set_tuning key 1): 1 = no operand DMA after the prologue, 2 = no attention phase,
// 4 = no MFMA in the K loop, 8 = no fragment reads in the K loop
template <bool LN, int ABL = 0, bool PANEL = false>
__global__ __launch_bounds__(Q3_NW * 64, 3) void qkv3_attn_kernel(Q3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, wc0 = 0;
    if (a.trace) { tr0 = clock64(); wc0 = wall_clock64(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 6, wn = wave - wm * 6;
    const int fr = lane & 31, fh = lane >> 5;

    // XCD x owns a contiguous range of (sequence, head group) units: the four groups of a sequence share one L2
    const int groups = a.heads / 3;
    int seq, h0;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        seq = lid / groups;
        h0 = (lid - seq * groups) * 3;
    }
    const int m0 = seq * a.L;
    const int M = a.M, hd64 = a.heads * 64;

    // ---- LDS-DMA pieces: piece p = stage rows [16 p, 16 p + 16) (rows 0..127 = tokens, 128.. = weight rows), lane l -> row
    // 16 p + (l >> 2), destination chunk l & 3 (linear), SOURCE chunk (l & 3) ^ swizzle(row).  Wave w issues pieces w, w + 12,
    // w + 24 and w + 36; pieces 44..47 (waves 8..11) do not exist: those waves re-read their first piece into a dummy area,
    // so that every wave has the same four loads per stage in flight (one counted vmcnt for all).
    const int n3 = 3 * hd64;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)min((size_t)(a.a_panel ? ((M + 31) & ~31) : M) * (a.a_panel ? a.K : a.lda) * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)min((size_t)n3 * a.ldw * 2, (size_t)0x7fffffff), 0x00020000);
    const bool big = wave < 8;             // wave-uniform: has a real fourth piece; its first piece is an activation piece
    const auto rs0 = big ? rsA : rsW;      // (scalar select: the main loop stays one basic block)
    // row-major weight: row stride ldw, K-tile t at byte offset 64 t;  K-tile-major copy: row stride 64 B, K-tile t at n3 * 64 t
    const unsigned w_row = a.w_tiled ? (unsigned)Q3_RB : (unsigned)a.ldw * 2u;
    const int w_kstep = a.w_tiled ? n3 * Q3_RB : Q3_RB;
    // (A from the panel copy: a K-tile of 32 is two 1 KiB units of the row block further on; a piece's 16 rows x 64 B become 256-byte runs of one unit half)
    const int k0step = big ? (a.a_panel ? 2048 : Q3_RB) : w_kstep;       // first piece: activations (waves 0..7) or weights
    unsigned voff[Q3_G];
#pragma unroll
    for (int i = 0; i < Q3_G; ++i) {
        const int p = (i == 3 && !big) ? wave : wave + 12 * i;
        const int row = p * 16 + (lane >> 2);
        const int sc = (lane & 3) ^ ((row >> 2) & 3);
        if (p < 8) {
            const int rg = min(m0 + row, M - 1);
            voff[i] = a.a_panel ? (unsigned)((((rg >> 5) * (a.K >> 4) + (sc >> 1)) * 64 + (sc & 1) * 32 + (rg & 31)) * 16)
                                : (unsigned)(((size_t)rg * a.lda) * 2 + sc * 16);
        }
        else {
            const int rw = row - Q3_TM;                        // 0..575: head rw / 192, (q | k | v) block, row in head
            const int hh = rw / 192, r2 = rw - hh * 192;
            voff[i] = (unsigned)((r2 >> 6) * hd64 + (h0 + hh) * 64 + (r2 & 63)) * w_row + (unsigned)(sc * 16);
        }
    }
    auto stage = [&](int slot, int t) {
#pragma unroll
        for (int i = 0; i < Q3_G; ++i) {
            const int p = wave + 12 * i;
            const int dst = (i == 3 && !big) ? Q3_DUMMY + (wave - 8) * 1024 : slot * Q3_STAGE + p * 1024;
            auto lds = (__attribute__((address_space(3))) void*)(smem + dst);
            if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, lds, 16, voff[i], t * k0step, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds, 16, voff[i], t * w_kstep, 0, 0);
        }
    };

    // Folded LayerNorm: the partial row sums of the tile's 128 rows are one contiguous region of st_in (<= 8 KB).  They ride
    // into LDS in the dummy slots of the first two stages (waves 8..11, four 1 KiB pieces per stage), so the epilogue finds
    // them on chip instead of waiting for a cold global load behind the K loop.
    const int st_stride = ((a.st_parts + 1) & ~1) * 8;         // bytes per row of the table
    const auto rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.st_in, 0, LN ? (int)min((size_t)M * st_stride, (size_t)0x7fffffff) : 0, 0x00020000);
    auto stage_first = [&](int slot, int t) {              // t = 0, 1
#pragma unroll
        for (int i = 0; i < Q3_G; ++i) {
            const int p = wave + 12 * i;
            if (LN && i == 3 && !big) {
                const int q = t * 4 + (wave - 8);
                const unsigned vo = (unsigned)min((long long)m0 * st_stride + q * 1024 + lane * 16, (long long)M * st_stride - 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(smem + Q3_ST + q * 1024), 16, vo, 0, 0, 0);
                continue;
            }
            const int dst = (i == 3 && !big) ? Q3_DUMMY + (wave - 8) * 1024 : slot * Q3_STAGE + p * 1024;
            auto lds = (__attribute__((address_space(3))) void*)(smem + dst);
            if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, lds, 16, voff[i], t * k0step, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds, 16, voff[i], t * w_kstep, 0, 0);
        }
    };
    const int nt = a.K / 32;
    stage_first(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // column constants of the tile's 576 columns (c, d of the LayerNorm fold, or the bias): two loads per lane issued BEHIND the first
    // operand stage and ahead of the other two, by INLINE ASM with their own counted wait -- for a load it can see, hipcc waits
    // vmcnt(0) while LDS-DMA is in flight (it does not count the DMA), which would hold the loop start until all three stages
    // have landed.  Loads return in order: vmcnt(8) = the two stages issued behind them may stay in flight.  Parked in LDS.
    f32x4 cq = {0.f, 0.f, 0.f, 0.f}, dq = cq;
    const float* pc = LN ? a.colc : a.bias;
    const float* pd = LN ? a.cold : a.bias;
    const bool have_cd = pc != nullptr;               // (kernel argument: wave-uniform)
    if (have_cd) {
        const int col = min(tid, Q3_TN / 4 - 1) * 4, seg = col >> 6, hh = seg / 3, which = seg - hh * 3;
        const int gcol = which * hd64 + (h0 + hh) * 64 + (col & 63);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cq) : "v"(pc + gcol) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dq) : "v"(pd + gcol) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_first(1, 1);                                     // (K >= 96: three K-tiles at least)
    stage(2, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (have_cd) {
        if (nt >= Q3_STAGES) asm volatile("s_waitcnt vmcnt(8)" : "+v"(cq), "+v"(dq) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(cq), "+v"(dq) : : "memory");
    }
    if (!LN) cq = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < Q3_TN / 4) {
        *reinterpret_cast<f32x4*>(smem + Q3_CD + tid * 16) = cq;
        *reinterpret_cast<f32x4*>(smem + Q3_CD + Q3_TN * 4 + tid * 16) = dq;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;


    // fragments: every row this lane reads is (a multiple of 32) + fr, so the swizzle term is the lane constant (fr >> 2) & 3
    bf16x8 fa[2][2], fb[2][3];
    const unsigned sx = (unsigned)((fr >> 2) & 3);
    const unsigned abase = (unsigned)(wm * 64 + fr) * Q3_RB, bbase = (unsigned)(Q3_TM + wn * 96 + fr) * Q3_RB;
    auto ldfrag = [&](int slot, int ks, int pb) {
        if (ABL & 8) return;
        const unsigned co = (((unsigned)(ks * 2 + fh)) ^ sx) << 4;
        const unsigned char* st = smem + slot * Q3_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[pb][i] = *reinterpret_cast<const bf16x8*>(st + abase + co + i * 32 * Q3_RB);
#pragma unroll
        for (int j = 0; j < 3; ++j) fb[pb][j] = *reinterpret_cast<const bf16x8*>(st + bbase + co + j * 32 * Q3_RB);
    };
    auto mma = [&](int pb) {
        if (ABL & 4) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][j], fa[pb][i], acc[i][j], 0, 0, 0);   // transposed: lane = token
    };
    auto touch = [&](int pb) {     // places the compiler's lgkmcnt wait for fragment set pb here
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(fa[pb][i]));
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(fb[pb][j]));
    };
    // tile t has landed once at most (last - t) younger tiles are in flight (four loads each)
    auto wait_tile = [&](int t, int last) {
        const int after = last - t;
        if (after >= 2) q3_wait_vm<2 * Q3_G>();
        else if (after == 1) q3_wait_vm<Q3_G>();
        else q3_wait_vm<0>();
    };
#define Q3_SB() __builtin_amdgcn_sched_barrier(0)
    if (nt > 0) {
        wait_tile(0, min(nt, Q3_STAGES) - 1);
        __builtin_amdgcn_s_barrier();
        Q3_SB();
        ldfrag(0, 0, 0);
        Q3_SB();
    }
    if (a.trace) tr1 = clock64();
    int slot = 0;
    // Pipeline (as gemm.hip, FD = 2): RAW -- a wave reads tile t + 1 only after its own counted vmcnt wait AND the barrier
    // behind it; WAR -- every wave has RETIRED its reads of tile t (touch) before it arrives at that barrier, and the refill
    // of tile t's slot is issued after it.  Main iterations are one basic block (exact waitcnt bookkeeping by the compiler).
    auto body = [&](int t, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const bool more = MAIN || (t + 1 < nt);
        int nslot = slot + 1;
        if (nslot == Q3_STAGES) nslot = 0;
        touch(0); Q3_SB(); ldfrag(slot, 1, 1); Q3_SB(); mma(0); Q3_SB();
        touch(1); Q3_SB();
        if (more) {
            if (MAIN) q3_wait_vm<(Q3_STAGES - 2) * Q3_G>(); else wait_tile(t + 1, min(nt, t + Q3_STAGES) - 1);
            Q3_SB();
            __builtin_amdgcn_s_barrier();
            Q3_SB();
            ldfrag(nslot, 0, 0);
            Q3_SB();
        }
        mma(1); Q3_SB();
        // refill of tile t's slot: behind the barrier (WAR) and behind this wave's MFMAs, so that the twelve waves' DMA issue
        // does not sit between the barrier release and the first MFMA
        if (more && !(ABL & 1) && (MAIN || t + Q3_STAGES < nt)) { stage(slot, t + Q3_STAGES); Q3_SB(); }
        slot = nslot;
    };
    const int t_main = max(nt - Q3_STAGES, 0);
    for (int t = 0; t < t_main; ++t) body(t, std::true_type{});
    for (int t = t_main; t < nt; ++t) body(t, std::false_type{});
#undef Q3_SB
    if (a.trace) tr2 = clock64();

    // ---- Q | K | V of the three heads -> bf16 tiles in LDS (the ring is dead: every wave's DMA has landed and been read)
    float mu[2] = {0.f, 0.f}, rs[2] = {1.f, 1.f};
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // the arithmetic of sum_parts_n<4> (slot order, unused slots skipped by select), read from the LDS copy
            const f32x4* sp = reinterpret_cast<const f32x4*>(smem + Q3_ST + (wm * 64 + i * 32 + fr) * st_stride);
            const int nq = st_stride >> 4;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = sp[min(q, nq - 1)];
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool u0 = 2 * q < a.st_parts, u1 = 2 * q + 1 < a.st_parts;
                sm += u0 ? v[q][0] : 0.f; sq += u0 ? v[q][1] : 0.f;
                sm += u1 ? v[q][2] : 0.f; sq += u1 ? v[q][3] : 0.f;
            }
            ln_mean_rstd(sm, sq, a.inv_h, a.eps, mu[i], rs[i]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int seg = (wn * 96 + j * 32) >> 6;              // 64-column segment of the tile: head seg / 3, (q | k | v) = seg % 3
        const int hh = seg / 3, which = seg - hh * 3;
        unsigned char* tile = smem + (which * 3 + hh) * Q3_TILE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cc = ((wn * 96 + j * 32) & 63) + 8 * g + 4 * fh;           // column inside the head
            const int tcol = wn * 96 + j * 32 + 8 * g + 4 * fh;                  // column of the tile
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(smem + Q3_CD + tcol * 4);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(smem + Q3_CD + Q3_TN * 4 + tcol * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int trow = wm * 64 + i * 32 + fr;
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x;
                    if constexpr (LN) x = ln_fold(acc[i][j][4 * g + e], mu[i], rs[i], c4[e], d4[e]);
                    else x = acc[i][j][4 * g + e] + d4[e];
                    pk[e] = (bf16)x;
                }
                unsigned char* dst = which == 2 ? tile + att_voff_swz(trow, cc >> 3) + (cc & 7) * 2
                                                : tile + att_koff16(trow, cc >> 3) + (cc & 7) * 2;
                *reinterpret_cast<bf16x4*>(dst) = pk;
            }
        }
    }
    float* sMask = reinterpret_cast<float*>(smem + Q3_MASK);
    const int Ls = a.L;
    if (tid < 128) {
        float mv = -INFINITY;                                  // rows past the sequence (the next sequence's tokens): not keys
        if (tid < Ls) mv = a.mask ? (1.0f - (float)a.mask[(size_t)seq * Ls + tid]) * (-10000.0f * ATT_LOG2E) : 0.f;
        sMask[tid] = mv;
    }
    __syncthreads();
    if (a.trace) tr3 = clock64();
    // ---- attention: wave w = (head w / 4, queries 32 (w % 4) ..)
    {
        const int hh = wave >> 2, qb = wave & 3;
        if (qb * 32 < Ls && !(ABL & 2)) {
            const int q = qb * 32 + fr;
            const unsigned char* sQ = smem + hh * Q3_TILE;
            const unsigned char* sK = smem + (3 + hh) * Q3_TILE;
            const unsigned char* sV = smem + (6 + hh) * Q3_TILE;
            bf16x8 fq[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const bf16x8*>(sQ + att_koff16(q, 2 * ks + fh));
            if constexpr (PANEL) {
                attn_core_bf16<4, true, true>(fq, sK, sV, sMask, lane, q < Ls, nullptr, nullptr, Ls, DropSpec{}, 0, 0, nullptr,
                                              (void*)a.ctx, (int)min((size_t)((M + 31) & ~31) * hd64 * 2, (size_t)0x7fffffff), m0 + min(q, Ls - 1), (h0 + hh) * 8, hd64 >> 4);
            } else {
            bf16* crow = a.ctx + ((size_t)m0 + min(q, Ls - 1)) * a.ldo + (h0 + hh) * 64;
            attn_core_bf16<4, true>(fq, sK, sV, sMask, lane, q < Ls, crow, nullptr, Ls);
            }
        }
    }
    if (a.trace && tid == 0) {
        long long* t = a.trace + (size_t)blockIdx.x * 8;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = tr3; t[4] = clock64();
        t[5] = wc0; t[6] = wall_clock64();                      // constant-rate (100 MHz) device-wide counter: start, end
        t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    }
#endif
}

template <bool LN, int ABL = 0, bool PANEL = false>
int q3_launch(const Q3Args& a, int B, hipStream_t s) {
    auto kern = qkv3_attn_kernel<LN, ABL, PANEL>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q3_LDS);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    kern<<<dim3(B * (a.heads / 3)), dim3(Q3_NW * 64), Q3_LDS, s>>>(a);
    return CPT_OK;
}

}  // namespace

CPT_SWITCH(int g_q3_abl, 0);
extern int g_trace_epi;
long long* g_q3_trace = nullptr;
void set_q3_trace(void* p) { g_q3_trace = (long long*)p; }
// (K = hidden <= 768: the rows' partial LayerNorm sums fit 8 slots = the 8 KB the kernel parks them in)
int qkv_attn3_eligible(int L, int heads, int K) { return L > 0 && L <= 128 && heads > 0 && heads % 3 == 0 && K >= 96 && K <= 768 && K % 32 == 0; }

// Same contract as gemm_qkv_attn (gemm.hip): st_in == NULL -> x W^T + bias, else the LayerNorm-folded form.
int gemm_qkv_attn3(const void* A, int lda, const void* W, int ldw, const float* bias, const float* st_in, const float* colc,
                   const float* cold, float eps, int hidden, const int64_t* mask, void* ctx, int ldo, int B, int L, int heads,
                   int K, hipStream_t s, int w_tiled, int ctx_panel, int a_panel) {
    if (a_panel && K % 16) return CPT_ERR_SHAPE;
    if (B <= 0 || !qkv_attn3_eligible(L, heads, K) || lda % 8 || ldw % 8 || ldo % 4 || (st_in && ln_stat_slots(hidden) > 8)) return CPT_ERR_SHAPE;
    if (!A || !W || !ctx || (st_in && (!colc || !cold))) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)colc | (uintptr_t)cold) & 15) || ((uintptr_t)ctx & (ctx_panel ? 15 : 7))) return CPT_ERR_ALIGN;
    if (ctx_panel && (heads * 64) % 16) return CPT_ERR_SHAPE;
    Q3Args a;
    a.A = (const bf16*)A; a.lda = lda; a.a_panel = a_panel ? 1 : 0; a.W = (const bf16*)W; a.ldw = ldw; a.bias = bias; a.w_tiled = w_tiled ? 1 : 0;
    if (w_tiled && ldw != K) return CPT_ERR_SHAPE;
    a.st_in = st_in; a.st_parts = ln_stat_parts(hidden); a.colc = colc; a.cold = cold; a.eps = eps; a.inv_h = 1.0f / (float)hidden;
    a.trace = (g_trace_epi < 0 || g_trace_epi == 10) ? g_q3_trace : nullptr;      // (diagnostics: trace filter by epilogue id, 10 = fused QKV + attention)
    a.mask = mask; a.ctx = (bf16*)ctx; a.ldo = ldo; a.M = B * L; a.K = K; a.L = L; a.heads = heads;
#ifdef CPT_ABLATION      // diagnostic builds only (tools/abl_sweep.sh, cpt_set_tuning(1, bits)): timing instantiations with GARBAGE results; row-major ctx only
    if (st_in && g_q3_abl && ctx_panel) return CPT_ERR_SHAPE;
    if (st_in) switch (g_q3_abl) {
        case 1: return q3_launch<true, 1>(a, B, s);
        case 2: return q3_launch<true, 2>(a, B, s);
        case 3: return q3_launch<true, 3>(a, B, s);
        case 4: return q3_launch<true, 4>(a, B, s);
        case 8: return q3_launch<true, 8>(a, B, s);
        case 12: return q3_launch<true, 12>(a, B, s);
        case 13: return q3_launch<true, 13>(a, B, s);
        case 15: return q3_launch<true, 15>(a, B, s);
        default: break;
    }
#endif
    if (ctx_panel) return st_in ? q3_launch<true, 0, true>(a, B, s) : q3_launch<false, 0, true>(a, B, s);
    return st_in ? q3_launch<true>(a, B, s) : q3_launch<false>(a, B, s);
}
void set_q3_abl(int v) { CPT_SWITCH_SET(g_q3_abl = v); (void)v; }

}  // namespace cpt
