// Round 4: the LayerNorm-CONSUMER GEMMs of the fused bf16 encoder (FFN-up = BertIntermediate dense + GELU,
// /root/reference/Oscar/oscar/modeling/modeling_bert.py:144, and the stand-alone QKV projection of sequences longer than 128, :38-40) with ONE
// wave per SIMD and the operand stream issued between the MFMAs:
//   out[M][N] = [gelu]( rstd[m] * (A.Wf^T - mean[m] * colc[n]) + cold[n] )          (LayerNorm folded, gemm.hip / DESIGN.md 5c)
//
// Why (profiles/r04_kloop_vs_hipblaslt.md): the two-pass 384 x 256 kernel (gemm_ffn.hip, 8 waves, two per SIMD running the same stream
// between the same barriers) spends 2700-2900 cycles per K-tile against 1536 of MFMA; hipBLASLt's winners are 4-wave workgroups, and the
// 4-wave form of the LayerNorm producers (gemm_prod.hip CPT_TILE4) runs its K loop at 87 % matrix-pipe occupancy once every LDS read and
// LDS-DMA piece sits BETWEEN two MFMAs of the wave's own stream.  Same here:
//   tile 192 x 256 (one pass), 4 waves as 2 (M) x 2 (N), wave tile 96 x 128 = 3 x 4 MFMA 32x32x16 blocks with swapped operands
//   (accumulators transposed: lane = output row, register quad = 4 consecutive columns; 192 accumulator registers, 7 fragment reads per
//   12 MFMAs instead of 5 per 6), operands by LDS-DMA into rings of different depth (A tiles 24 KB x 2, W tiles 32 KB x 3, as gemm_ffn.hip),
//   one barrier per K-tile behind the last MFMA of k-step 2, refill + next tile's first fragments under k-step 3.
// 480 workgroups at M = 7680, N = 3072 (two rounds of one per CU, the second dispatched as the first drains).
// Same MFMA order over K and the same epilogue arithmetic as gemm_ffn.hip / the generic consumer: bit-identical outputs (tested).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace cpt {
extern long long* g_gemm_trace;      // diagnostics (cpt_debug_gemm_trace)
namespace {

constexpr int RB = 128;                       // bytes per operand-tile row (64 bf16)
constexpr int TM = 192, TN = 256;
constexpr int NWV = 4;
constexpr int A_SLOT = TM * RB, W_SLOT = TN * RB;            // 24 KB, 32 KB
constexpr int NA = 2, NWR = 3;                                // ring depths
constexpr int GA = TM / 8 / NWV, GW = TN / 8 / NWV;           // LDS-DMA pieces (1 KiB = 8 rows) per wave per tile: 6 of A, 8 of W
constexpr int MI = 3, NJ = 4;
constexpr int W_RING = NA * A_SLOT, SIDE_C = W_RING + NWR * W_SLOT, SIDE_D = SIDE_C + TN * 4, SIDE_ST = SIDE_D + TN * 4,
              LDS_BYTES = SIDE_ST + TM * 8;                   // 48 + 96 + 2 + 1.5 KB
static_assert(GA * 8 * NWV == TM && GW * 8 * NWV == TN && LDS_BYTES <= 160 * 1024, "tile must split evenly over the waves and fit the LDS");

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// NT: K-tiles (K = 64 NT).  PANEL: out is the fragment-major panel copy [M / 32][N / 16][64][8] the FFN-down producer reads (gemm_prod.hip).
// GELU = false: plain LayerNorm-consumer GEMM (stand-alone QKV projection).
template <int NT, bool PANEL, bool GELU>
__global__ __launch_bounds__(256, 1) void lncons4_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                                                          bf16* __restrict__ out, int ldo, int M, int N,
                                                          const float* __restrict__ st_in, int st_parts, const float* __restrict__ colc,
                                                          const float* __restrict__ cold, float eps, float inv_h,
                                                          const void* __restrict__ pf, size_t pf_bytes, int pf_first, int npf, long long* __restrict__ trace) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (trace) tr0 = clock64();
    // workgroups [pf_first, pf_first + npf) prefetch the next launch's weights into the Infinity Cache (common.h prefetch_region): leading
    // ones when the tiles leave CUs idle in their only round, else the first ones of the second round
    int bid = blockIdx.x;
    if (npf) {
        if (bid >= pf_first && bid < pf_first + npf) {
            if (pf) prefetch_region(pf, pf_bytes, bid - pf_first, npf, threadIdx.x, 256, smem);
            return;
        }
        if (bid >= pf_first + npf) bid -= npf;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;

    // XCD-first tile order, 4 row tiles per group (as gemm.hip)
    int m0, n0;
    {
        const int tm = (M + TM - 1) / TM, tn = N / TN;
        const int nwg = tm * tn;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int gm = 4, per_group = gm * tn;
        const int g = lid / per_group, first_m = g * gm;
        const int gsz = min(tm - first_m, gm);
        const int in_g = lid - g * per_group;
        m0 = (first_m + in_g % gsz) * TM;
        n0 = (in_g / gsz) * TN;
    }

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)M * lda * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)min((size_t)((M + 31) & ~31) * (PANEL ? N : ldo) * 2, (size_t)0x7fffffff), 0x00020000);
    const int rbase = wave * 8 + (lane >> 3);
    const unsigned c16 = (unsigned)(((lane & 7) ^ ((rbase >> 1) & 7)) * 16);      // source-side XOR swizzle (rows rbase + 32 i: the same row parity term)
    unsigned voa[GA], vow[GW];
#pragma unroll
    for (int i = 0; i < GA; ++i) voa[i] = (unsigned)min(m0 + rbase + i * NWV * 8, M - 1) * (unsigned)(lda * 2) + c16;
#pragma unroll
    for (int i = 0; i < GW; ++i) vow[i] = (unsigned)min(n0 + rbase + i * NWV * 8, N - 1) * (unsigned)(ldw * 2) + c16;
#define CPT_DA(SLOT, T, I)                                                                                                                 \
    do {                                                                                                                                   \
        auto lds_ = (__attribute__((address_space(3))) void*)(smem + (SLOT) * A_SLOT + ((I) * NWV + wave) * 1024);                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lds_, 16, voa[I], (T) * 128, 0, 0);                                                   \
    } while (0)
#define CPT_DW(SLOT, T, I)                                                                                                                 \
    do {                                                                                                                                   \
        auto lds_ = (__attribute__((address_space(3))) void*)(smem + W_RING + (SLOT) * W_SLOT + ((I) * NWV + wave) * 1024);                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds_, 16, vow[I], (T) * 128, 0, 0);                                                   \
    } while (0)
#define CPT_SB() __builtin_amdgcn_sched_barrier(0)
    auto stage_a = [&](int slot, int t) {
        CPT_DA(slot, t, 0); CPT_DA(slot, t, 1); CPT_DA(slot, t, 2); CPT_DA(slot, t, 3); CPT_DA(slot, t, 4); CPT_DA(slot, t, 5);
    };
    auto stage_w = [&](int slot, int t) {
        CPT_DW(slot, t, 0); CPT_DW(slot, t, 1); CPT_DW(slot, t, 2); CPT_DW(slot, t, 3); CPT_DW(slot, t, 4); CPT_DW(slot, t, 5); CPT_DW(slot, t, 6); CPT_DW(slot, t, 7);
    };

    // Side data (the tile's column vectors and its 192 rows' partial LayerNorm sums): inline-asm loads right behind the first K-tile's DMA,
    // parked in LDS after the first barrier (as gemm_ffn.hip)
    float* side_c = reinterpret_cast<float*>(smem + SIDE_C);
    float* side_d = reinterpret_cast<float*>(smem + SIDE_D);
    float2* side_st = reinterpret_cast<float2*>(smem + SIDE_ST);
    constexpr int NSIDE = 7;
    f32x4 cd, sv[6];
    stage_a(0, 0); stage_w(0, 0);
    CPT_SB();
    {
        const float* cdp = tid < 64 ? colc + n0 + tid * 4 : cold + n0 + (min(tid, 127) - 64) * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cd) : "v"(cdp));
        const int row = min(m0 + min(tid, TM - 1), M - 1);
        const int slots = (st_parts + 1) & ~1, nq = slots >> 1;
        const f32x4* base = reinterpret_cast<const f32x4*>(st_in + (size_t)row * slots * 2);
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sv[q]) : "v"(base + min(q, nq - 1)));
    }
    CPT_SB();
    stage_a(1, 1); stage_w(1, 1);
    stage_w(2, 2);
    CPT_SB();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments, double-buffered by k-step parity: fa[p][i] (A block i), fb[p][j] (W block j)
    bf16x8 fa[2][MI], fb[2][NJ];
    const unsigned abase = (unsigned)(wm * 96 + fr) * RB, wbase = (unsigned)W_RING + (unsigned)(wn * 128 + fr) * RB;
    const unsigned sx = (unsigned)((fr >> 1) & 7);
#define CPT_COFF(KS) ((((unsigned)((KS) * 2 + fh)) ^ sx) << 4)
#define CPT_RA(P, I, SA, KS) fa[P][I] = *reinterpret_cast<const bf16x8*>(smem + abase + CPT_COFF(KS) + (unsigned)((SA) * A_SLOT) + (I) * 32 * RB)
#define CPT_RB(P, J, SW, KS) fb[P][J] = *reinterpret_cast<const bf16x8*>(smem + wbase + CPT_COFF(KS) + (unsigned)((SW) * W_SLOT) + (J) * 32 * RB)
#define CPT_MF(P, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][J], fa[P][I], acc[I][J], 0, 0, 0)
    // one k-step on fragment buffer P; READ: fetch the fragments of k-step (SA, SW, KS) into buffer P ^ 1 meanwhile, one read behind each of
    // the first seven MFMAs, in the order the next k-step's MFMAs need them (fb0 fa0 fb1 fb2 fb3 fa1 fa2)
#define CPT_KSTEP(P, READ, SA, SW, KS)                                                                                 \
    do {                                                                                                               \
        CPT_MF(P, 0, 0); CPT_SB(); if (READ) { CPT_RB((P) ^ 1, 0, SW, KS); CPT_SB(); }                                  \
        CPT_MF(P, 0, 1); CPT_SB(); if (READ) { CPT_RA((P) ^ 1, 0, SA, KS); CPT_SB(); }                                  \
        CPT_MF(P, 0, 2); CPT_SB(); if (READ) { CPT_RB((P) ^ 1, 1, SW, KS); CPT_SB(); }                                  \
        CPT_MF(P, 0, 3); CPT_SB(); if (READ) { CPT_RB((P) ^ 1, 2, SW, KS); CPT_SB(); }                                  \
        CPT_MF(P, 1, 0); CPT_SB(); if (READ) { CPT_RB((P) ^ 1, 3, SW, KS); CPT_SB(); }                                  \
        CPT_MF(P, 1, 1); CPT_SB(); if (READ) { CPT_RA((P) ^ 1, 1, SA, KS); CPT_SB(); }                                  \
        CPT_MF(P, 1, 2); CPT_SB(); if (READ) { CPT_RA((P) ^ 1, 2, SA, KS); CPT_SB(); }                                  \
        CPT_MF(P, 1, 3); CPT_SB();                                                                                      \
        CPT_MF(P, 2, 0); CPT_SB();                                                                                      \
        CPT_MF(P, 2, 1); CPT_SB();                                                                                      \
        CPT_MF(P, 2, 2); CPT_SB();                                                                                      \
    } while (0)

    // issued so far per wave: A0 W0 | side | A1 W1 W2 = 6 8 | 7 | 6 8 8; tile 0 is complete when at most 7 + 22 are outstanding
    wait_vm<NSIDE + GA + 2 * GW>();
    __builtin_amdgcn_s_barrier();                               // tile 0 visible to every wave
    if (trace) tr1 = clock64();
    CPT_SB();
    CPT_RB(0, 0, 0, 0); CPT_RA(0, 0, 0, 0); CPT_RB(0, 1, 0, 0); CPT_RB(0, 2, 0, 0); CPT_RB(0, 3, 0, 0); CPT_RA(0, 1, 0, 0); CPT_RA(0, 2, 0, 0);
    CPT_SB();
    // the side data (issued with tile 0, older than A1 W1 W2) has landed too: reduce the partial sums in slot order and park everything in LDS
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

    // K-tile t: A slot t % 2, W slot t % 3.  Issue order per wave: ... | iteration t: A(t+2) W(t+3).  "Tile t+1 landed" leaves in flight what is
    // younger than A(t+1): W(t+2) = GW pieces (none when it does not exist).
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int sa = t & 1, sw = t % NWR, sa1 = sa ^ 1, sw1 = (t + 1) % NWR;
        const bool more = t + 1 < NT;
        CPT_KSTEP(0, true, sa, sw, 1);
        CPT_MF(0, 2, 3); CPT_SB();
        CPT_KSTEP(1, true, sa, sw, 2);
        CPT_MF(1, 2, 3); CPT_SB();
        CPT_KSTEP(0, true, sa, sw, 3);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of tile t are retired
            if (t + 2 < NT) wait_vm<GW>(); else wait_vm<0>();        // tile t+1 landed
            CPT_SB();
        }
        CPT_MF(0, 2, 3); CPT_SB();
        if (more) {
            __builtin_amdgcn_s_barrier();                           // ... for every wave; nobody still reads tile t's slots
            CPT_SB();
        }
        // k-step 3 (buffer 1): the next tile's first fragments and the refill of the two slots just freed ride between its MFMAs
        if (more) {
            CPT_MF(1, 0, 0); CPT_SB(); CPT_RB(0, 0, sw1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 0); CPT_SB(); }
            CPT_MF(1, 0, 1); CPT_SB(); CPT_RA(0, 0, sa1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 1); CPT_SB(); }
            CPT_MF(1, 0, 2); CPT_SB(); CPT_RB(0, 1, sw1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 2); CPT_SB(); }
            CPT_MF(1, 0, 3); CPT_SB(); CPT_RB(0, 2, sw1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 3); CPT_SB(); }
            CPT_MF(1, 1, 0); CPT_SB(); CPT_RB(0, 3, sw1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 4); CPT_SB(); }
            CPT_MF(1, 1, 1); CPT_SB(); CPT_RA(0, 1, sa1, 0); CPT_SB(); if (t + 2 < NT) { CPT_DA(sa, t + 2, 5); CPT_SB(); }
            CPT_MF(1, 1, 2); CPT_SB(); CPT_RA(0, 2, sa1, 0); CPT_SB(); if (t + 3 < NT) { CPT_DW(sw, t + 3, 0); CPT_SB(); }
            CPT_MF(1, 1, 3); CPT_SB(); if (t + 3 < NT) { CPT_DW(sw, t + 3, 1); CPT_SB(); }
            CPT_MF(1, 2, 0); CPT_SB(); if (t + 3 < NT) { CPT_DW(sw, t + 3, 2); CPT_DW(sw, t + 3, 3); CPT_SB(); }
            CPT_MF(1, 2, 1); CPT_SB(); if (t + 3 < NT) { CPT_DW(sw, t + 3, 4); CPT_DW(sw, t + 3, 5); CPT_SB(); }
            CPT_MF(1, 2, 2); CPT_SB(); if (t + 3 < NT) { CPT_DW(sw, t + 3, 6); CPT_DW(sw, t + 3, 7); CPT_SB(); }
            CPT_MF(1, 2, 3); CPT_SB();
        } else {
            CPT_KSTEP(1, false, 0, 0, 0);
            CPT_MF(1, 2, 3); CPT_SB();
        }
    }
#undef CPT_KSTEP
#undef CPT_MF
#undef CPT_RA
#undef CPT_RB
#undef CPT_COFF
#undef CPT_DA
#undef CPT_DW

    if (trace) tr2 = clock64();
    // ---- epilogue (the arithmetic of gemm_ffn.hip's epi_quad / epi_store): block (i, j), quad pair gp: LayerNorm fold, GELU, bf16 pack,
    // half-wave exchange, one 16-byte store per lane
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rl = wm * 96 + i * 32 + fr;
        const float2 ms = side_st[rl];
        const int row = m0 + rl;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // the block's 16 values of this lane (four quads) go through the fold and the GELU side by side: eight independent pairs keep the
            // VALU busy although this wave runs alone on its SIMD (one quad at a time: 15.6 k ticks per tile, tools/cons_bench.py)
            f32x2 x[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = wn * 128 + j * 32 + 8 * g + 4 * fh;
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(side_c + lc);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(side_d + lc);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[2 * g + (e >> 1)][e & 1] = ln_fold(acc[i][j][4 * g + e], ms.x, ms.y, c4[e], d4[e]);
            }
            if constexpr (GELU) gelu_fast2_n<8>(x);
            u32x2 k[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x4 p4 = {(bf16)x[2 * g][0], (bf16)x[2 * g][1], (bf16)x[2 * g + 1][0], (bf16)x[2 * g + 1][1]};
                k[g] = __builtin_bit_cast(u32x2, p4);
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // half-wave exchange: lanes 0-31 get columns [16 gp, 16 gp + 8) of the block, lanes 32-63 the next 8
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(k[2 * gp][0], k[2 * gp + 1][0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(k[2 * gp][1], k[2 * gp + 1][1], false, false);
                const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                const int col = n0 + wn * 128 + j * 32 + 16 * gp + 8 * fh;
                if (row < M) {
                    if constexpr (PANEL) __builtin_amdgcn_raw_buffer_store_b128(w, rsO, panel_unit(row, col >> 3, N >> 4) * 16u, 0, CPT_ST_AUX);
                    else *reinterpret_cast<u32x4*>(out + (size_t)row * ldo + col) = w;
                }
            }
        }
    }
    if (trace && tid == 0) {        // diagnostics (cpt_debug_gemm_trace): per-workgroup shader-clock stamps start / first tile landed / K loop done / end
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* tp = trace + (size_t)bid * 8;
        tp[0] = tr0; tp[1] = tr1; tp[2] = tr2; tp[3] = wall_clock64(); tp[4] = clock64();
    }
#undef CPT_SB
#endif
}

template <int NT, bool PANEL, bool GELU>
int launch4(const bf16* A, int lda, const bf16* W, int ldw, bf16* out, int ldo, int M, int N, const float* st_in, int st_parts,
            const float* colc, const float* cold, float eps, float inv_h, hipStream_t s, const void* pf, size_t pf_bytes) {
    auto kern = lncons4_kernel<NT, PANEL, GELU>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    const int ntile = ((M + TM - 1) / TM) * (N / TN);
    // prefetch workgroups (the next launch's weights): leading when the tiles leave CUs idle in a single round, else at the head of the second
    // round (in both cases a multiple of 8, so block id -> XCD is unchanged for the tiles)
    int npf = 0, pf_first = 0;
    if (pf && pf_bytes && !((uintptr_t)pf & 15)) {
        if (ntile <= 256 - 8) npf = std::min(CPT_PREFETCH_WGS, 256 - ntile) & ~7;
        else if (ntile > 256 && ntile <= 512 - CPT_PREFETCH_WGS) { npf = CPT_PREFETCH_WGS; pf_first = 256; }
    }
    kern<<<dim3(ntile + npf), dim3(256), LDS_BYTES, s>>>(A, lda, W, ldw, out, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, npf ? pf : nullptr, pf_bytes, pf_first, npf, g_gemm_trace);
    return CPT_OK;
}

}  // namespace

CPT_SWITCH(int g_lncons4, 1);      // cpt_set_tuning key 29: 1 (default) = the 4-wave consumer kernel where the two-pass kernel ran, 0 = the two-pass kernel (round 3)
void set_lncons4(int v) { CPT_SWITCH_SET(g_lncons4 = v); (void)v; }
int lncons4_enabled() { return g_lncons4; }

// same shapes as the two-pass kernel (gemm_ffn.hip ffn_up_2pass_legal): K = 768 or 1024, N % 256 == 0
int gemm_lncons4(const void* A, int lda, const void* Wf, int ldw, const float* st_in, int st_parts, const float* colc, const float* cold,
                 float eps, int hidden, void* out, int ldo, int M, int N, int K, hipStream_t s, int out_panel, const void* pf, size_t pf_bytes, int gelu) {
    if (!(K == 768 || K == 1024) || N % TN || M < TM) return CPT_ERR_SHAPE;
    if (!A || !Wf || !st_in || !colc || !cold || !out) return CPT_ERR_NULL;
    if (st_parts > 12) return CPT_ERR_SHAPE;
    if ((size_t)((M + 31) & ~31) * (out_panel ? N : ldo) * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;      // 32-bit store offsets
    if (lda % 8 || ldw % 8 || ldo % 8 || (((uintptr_t)A | (uintptr_t)Wf | (uintptr_t)out | (uintptr_t)colc | (uintptr_t)cold) & 15)) return CPT_ERR_ALIGN;
    if (out_panel && (!gelu || N % 16)) return CPT_ERR_SHAPE;
    const float inv_h = 1.0f / (float)hidden;
    const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)Wf; bf16* o = (bf16*)out;
#define CPT_L4(NT, PANEL, GELU) launch4<NT, PANEL, GELU>(a, lda, w, ldw, o, ldo, M, N, st_in, st_parts, colc, cold, eps, inv_h, s, pf, pf_bytes)
    if (K == 768) {
        if (!gelu) return CPT_L4(12, false, false);
        return out_panel ? CPT_L4(12, true, true) : CPT_L4(12, false, true);
    }
    if (!gelu) return CPT_L4(16, false, false);
    return out_panel ? CPT_L4(16, true, true) : CPT_L4(16, false, true);
#undef CPT_L4
}

}  // namespace cpt
