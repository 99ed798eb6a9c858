// MFMA GEMM for the dense projections of the CPT hot path:
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T + bias[N] (+ resid[M][N]) )
// W is in nn.Linear layout (out_features x in_features, row-major), so both operands are
// K-contiguous and feed MFMA fragments with 16-byte reads.  Replaces the torch Linear calls at
// /root/reference/Oscar/oscar/modeling/modeling_bert.py:38-40 (Q,K,V fused into N=3H), :85
// (BertSelfOutput.dense), :144 (BertIntermediate.dense + gelu), :145 (BertOutput.dense), :261
// (img_embedding) and the BertLMPredictionHead / BertPooler denses (modeling_rec.py:143,
// modeling_bert.py:275).
//
// gfx950 design: 128x128 tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2
// MFMA 32x32 blocks), K-tile of 128 bytes per row (64 bf16 / 32 f32) staged through LDS with a
// 16-row XOR swizzle so ds_read_b128 fragment reads are bank-conflict free, register-staged
// double buffering (next tile's global loads are issued before the current tile's MFMAs and
// written to the other LDS buffer afterwards: one barrier per K-tile).
//   T = bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   T = f32  : v_mfma_f32_32x32x2_f32 (exact fp32; parity mode)
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "attn_core.h"
#include "kernels.h"

namespace cpt {

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB = bytes per tile row
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 128-byte rows, two rows per 256-byte bank row: conflict-free when the 16 lanes of a
    // ds_read_b128 group touch rows that are distinct mod 16.
    return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T, int EPI, typename OT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
    OT* __restrict__ out, int ldo, int M, int N, int K) {
    typedef typename FragOf<T>::type frag_t;
    constexpr int CE = Chunk<T>::N;          // elements per 16-byte chunk
    constexpr int BK = ROWB / (int)sizeof(T);  // elements per K-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BM * ROWB];  // [buf][A|W]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging map: 1024 chunks per operand tile, 4 per thread
    const int sc = tid & 7;       // chunk within the row
    const int sr = tid >> 3;      // row 0..31 (+32*i)
    uint4 ra[4], rw[4];

    auto gload = [&](int k0) {
        const int kc = k0 + sc * CE;
        const bool kok = kc < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 32 * i;
            const int gm = m0 + r, gn = n0 + r;
            ra[i] = (kok && gm < M) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + kc)
                                    : make_uint4(0, 0, 0, 0);
            rw[i] = (kok && gn < N) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * ldw + kc)
                                    : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 32 * i;
            *reinterpret_cast<uint4*>(&smem[buf][0][lds_off(r, sc)]) = ra[i];
            *reinterpret_cast<uint4*>(&smem[buf][1][lds_off(r, sc)]) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload((t + 1) * BK);
        const unsigned char* sa = smem[buf][0];
        const unsigned char* sw = smem[buf][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const frag_t*>(sa + lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                fb[i] = *reinterpret_cast<const frag_t*>(sw + lds_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk(acc[i][j], fa[i], fb[j]);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + acc_col(lane);
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == CPT_EPI_GELU) v = gelu_erf(v);
                if (EPI == CPT_EPI_TANH) v = tanhf(v);
                if (EPI == CPT_EPI_RESID) v += resid[(size_t)row * ldr + col];
                out[(size_t)row * ldo + col] = from_f32<OT>(v);
            }
        }
    }
}


constexpr int GROUP_M = 4;             // row tiles per group in the XCD-first workgroup order (time within 2.5 % for 1..15; doubled for wide outputs)
constexpr int CPT_EPI_ATOMIC = 4;      // internal: split-K partial tiles added with fp32 atomics
constexpr int CPT_EPI_RESID_LP = 5;    // internal: residual operand is in the compute dtype T (bf16 residual stream)
constexpr int CPT_EPI_LNPROD = 6;      // internal: + residual (optionally LayerNorm'ed on the fly), writes fp32 + T copies and row sums
constexpr int CPT_EPI_LNCONS = 7;      // internal: A operand is a pre-LayerNorm tensor; LayerNorm folded into the epilogue
constexpr int CPT_EPI_LNCONS_GELU = 8; // internal: same + GELU
constexpr int CPT_EPI_ATTN = 9;        // internal: fused QKV projection + self-attention of one (sequence, head) per workgroup
constexpr int CPT_EPI_ATTN_LN = 10;    // internal: same, A operand is a pre-LayerNorm tensor (LayerNorm folded like LNCONS)
constexpr int CPT_EPI_GELU2 = 12;      // internal (training forward): writes u = A.W^T + bias (bf16, to EpiX.out_lp) AND gelu(u) (to out): BertIntermediate
constexpr int CPT_EPI_GELUGRAD = 13;   // internal (training backward): out = (A.W) * gelu'(u), u (compute dtype) passed in the residual slot: dgrad of BertOutput.dense + gelu backward
constexpr int CPT_EPI_GELU_X3 = 14;    // internal (bf16x3 parity mode): gelu(A.W^T + bias) written as the [hi | hi | lo] bf16 split copy the NEXT three-term GEMM reads (out [M][3N], EpiX.x3_k = N)
constexpr int CPT_EPI_LNPROD3 = 11;    // internal: LNPROD with the residual stream in the 3-byte form (bf16 hi + int8 lo, see r3_encode): in and out

// ---------------------------------------------------------------------------------------------
// Pipelined kernel: STAGES-deep LDS ring fed by LDS-DMA with COUNTED vmcnt waits (tiles stay in
// flight across the raw s_barrier; one barrier per K-tile), generic tile/wave shape, and an
// epilogue staged through LDS so that bias / GELU / residual / stores run on whole rows with
// 16-byte accesses instead of 2-byte scattered stores.
//   tile TBM x TBN, waves WM x WN, each wave (TBM/WM) x (TBN/WN) as MI x NJ MFMA 32x32 blocks
// ---------------------------------------------------------------------------------------------
// extra epilogue operands of the LayerNorm-folding modes (see cpt_abi.hip, "folded LayerNorm")
struct EpiX {
    // Row statistics travel as PARTIAL sums, one slot per 96-column block of the producing GEMM: [M][slots][2]
    // (sum, sum of squares).  Every producer wave owns exactly one slot per row (plain stores: no atomics, nothing to
    // zero), and every reader adds the `parts` slots in index order, so the statistics are bit-reproducible.
    const float* st_in;    // partial row sums of the LayerNorm INPUT this GEMM reads / of the residual source
    int st_in_parts;       // number of slots in st_in
    const float* g_in;     // LNPROD: gain of the LayerNorm applied to the residual source on the fly (NULL: residual used as is)
    const float* b_in;
    float* st_out;         // LNPROD: partial row sums of this GEMM's output, slot = output column / 96
    int st_out_slots;      // slots per row of st_out (ln_stat_slots of this GEMM's N)
    void* out_lp;          // LNPROD: copy of the output in the compute dtype
    const signed char* resid_lo;  // LNPROD3: low bytes of the residual (its bf16 hi part travels in `resid`)
    signed char* out_lo;          // LNPROD3: low bytes of the output (hi part -> out_lp; no fp32 copy is written)
    const float* colc;     // LNCONS: c[n] = sum_k W'[n][k]  (W' = gain-folded weight as the MFMA sees it)
    const float* cold;     // LNCONS: d[n] = sum_k beta[k] W[n][k] + bias[n]
    float eps, inv_h;      // LayerNorm eps, 1 / hidden
    const int64_t* mask;   // ATTN: [B][L] attention mask (1 keep / 0 drop) or NULL
    int seq_len, heads;    // ATTN: tokens per sequence (<= 128), attention heads
    int w_rows;            // NN form: rows of W that exist (0: K); rows beyond read as zero (K rounded up to a K-tile multiple)
    size_t split_stride;   // split-K partial matrices: elements between two splits' outputs (0: M * ldo)
    int x3_k;              // GELU_X3: columns of one part of the split copy (= N)
    float* colsum;         // GELUGRAD (training backward): += column sums of the finished output = the gradient of the bias in front of the GELU
    int colsum_rows;       // 0: colsum[N] takes atomic adds; > 0 (round 6): colsum is [colsum_rows][N] partial rows, one per 32-row wave block of the output (row = first row / 32), written with plain stores and added up by a later launch's column-sum job (kernels.h ColJob)
    int skew;              // two-workgroups-per-CU shapes: start delay of every second workgroup (see skew_start)
};

constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// TN = 1 (round 2, weight gradients): out[M][N] = sum_k A[k][m] . W[k][n] -- both operands are stored with the CONTRACTION index as
// the slow dimension (A = dY [rows][M], W = X [rows][N], K = rows), which is how the backward pass holds them.  A K-tile is 64
// rows of each operand (the same bytes per stage as the NT form: TBM * 128 + TBN * 128), staged by LDS-DMA as row-major
// [64][TBM] / [64][TBN] tiles whose 16-byte chunks are XOR-swizzled inside aligned groups of 8 by 2 * (row & 3) on the SOURCE
// side, and the MFMA fragments come out of them through ds_read_b64_tr_b16 (two transpose reads = 8 consecutive contraction
// rows of one output row/column).  A half-wave's transpose read touches 4 rows x 64 bytes (two 16-lane groups, 32 B each per row):
// tn_swz places the four rows in the four 64-byte quarters of the 256-byte bank row for both tile pitches.  Replaces the two
// explicit transposes per weight gradient (11 % of the training step in round 1).  Requires M % TBM == 0, N % TBN == 0,
// K % 64 == 0 (the caller falls back to the transposed-operand form otherwise).
// chunk-index XOR of tile row `row` (chunks per row cpr = 16: pitch 256 B, rows alias -> quarter = row & 3; cpr = 24: pitch 384 B,
// rows alternate between the two halves of the bank row -> one more bit from row >> 1); always inside an aligned group of 8 chunks
__device__ __forceinline__ int tn_swz(int row, int cpr) { return (cpr % 16 == 0) ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }

// The kernel body as an inlined device function: gemm_pipe_kernel runs it on the launch's one problem (bid_x = blockIdx.x),
// gemm_tn_group2_kernel (weight gradients, round 3) on one of TWO problems that share a launch.
template <typename T, int EPI, typename OT, int TBM, int TBN, int WM, int WN, int STAGES, int EP = 1, int FD = 4, int OCC = 1, int TN = 0>
__device__ __forceinline__ void gemm_pipe_body(
    const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
    OT* __restrict__ out, int ldo, int M, int N, int K, int splitk, long long* __restrict__ trace_arg, int abl_arg, EpiX ex, const int bid_x) {
#if defined(__HIP_DEVICE_COMPILE__)   // body uses gfx950-only builtins/types (buffer rsrc, "v" asm): device pass only
    // Ablation bits (cpt_set_tuning key 1) exist only in -DCPT_ABLATION builds: as run-time tests they put nine scalar
    // branches into every K-loop iteration (measured: FFN-up +7 us), so the shipped kernels compile them away.
#ifdef CPT_ABLATION
    const int abl = abl_arg;
    long long* __restrict__ trace = trace_arg;
#else
    constexpr int abl = 0;
    constexpr long long* trace = nullptr;
    (void)abl_arg; (void)trace_arg;
#endif
    typedef typename FragOf<T>::type frag_t;
    constexpr int CE = Chunk<T>::N;
    constexpr int BK = ROWB / (int)sizeof(T);
    constexpr int NW = WM * WN, NT = NW * 64;
    long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;
    if (trace) tr0 = clock64();
    constexpr int MI = TBM / WM / 32, NJ = TBN / WN / 32;
    constexpr int ROWS = TBM + TBN;
    constexpr int G = ROWS / 8 / NW;                  // LDS-DMA pieces per wave per stage
    constexpr int STAGE_BYTES = ROWS * ROWB;
    constexpr int CP = TBN * 4 + 16;                  // epilogue staging pitch (bytes)
    static_assert(ROWS % (8 * NW) == 0, "stage must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // Two co-resident workgroups per CU start together and would run their phases in lockstep (both in the K loop,
    // then both in the epilogue): delaying every second one by about half a K loop puts one's epilogue (VALU, LDS,
    // stores) under the other's MFMAs, and the offset persists through the later rounds.  skew: low byte = delay in
    // units of 1024 cycles, bits 8.. = log2 of the block-id stride that separates the two co-resident workgroups.
    if constexpr (OCC == 2) {
        const int amount = ex.skew & 255, sh = ex.skew >> 8;
        if (amount > 0 && ((bid_x >> sh) & 1))
            for (int i = 0; i < amount; ++i) __builtin_amdgcn_s_sleep(16);
    }

    // XCD-first, then GROUP_M row tiles per group (see tile_of_block)
    constexpr bool ATTN = EPI == CPT_EPI_ATTN || EPI == CPT_EPI_ATTN_LN;
    // LayerNorm-folding epilogues of the bf16 encoder run straight from the accumulator registers (see "direct epilogue")
    // The LayerNorm PRODUCER keeps the slab epilogue: its residual loads and fp32 stores are row-per-lane in the direct form
    // (32 rows x 32 B per instruction instead of 1 KB runs), measured slower on MI355X (attn-out 24.7 -> 28.2 us, FFN-down
    // 51.5 -> 55.3 us; tools/epi_bench.hip: 7.0 vs 5.3 us for the fp32 tile stores alone).  The direct code path below still
    // handles it (DIRECT_LNPROD) for A/B runs.
    constexpr bool DIRECT_LNPROD = false;
    constexpr bool DIRECT = ((EPI == CPT_EPI_LNPROD && DIRECT_LNPROD) || EPI == CPT_EPI_LNCONS || EPI == CPT_EPI_LNCONS_GELU) && sizeof(T) == 2;
    int m0, n0, split;
    int att_b = 0, att_h = 0;
    if constexpr (ATTN) {
        // one workgroup per (sequence, head): rows = the sequence's tokens, columns = that head's Q | K | V slices.
        // XCD x owns a contiguous range of (sequence, head) pairs, so a sequence's 12 heads share one L2.
        static_assert(TBM == 128 && TBN == 192 && MI == 1 && NJ == 3, "fused attention: 128 tokens x (64 Q | 64 K | 64 V)");
        const int nwg = gridDim.x, bid = bid_x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        att_b = lid0 / ex.heads;
        att_h = lid0 - att_b * ex.heads;
        m0 = att_b * ex.seq_len;
        n0 = 0;
        split = 0;
    } else
    {
        const int tm = (M + TBM - 1) / TBM, tn = (N + TBN - 1) / TBN;
        const int nwg = tm * tn * splitk, bid = bid_x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int lid0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int lid = lid0 / splitk;
        split = lid0 - lid * splitk;
        const int gm = tn >= 16 ? 2 * GROUP_M : GROUP_M;     // wide outputs (FFN-up): taller groups keep the A panel in L2 (PMC: 120 vs 142 MB)
        const int per_group = gm * tn;
        const int g = lid / per_group, first_m = g * gm;
        const int gsz = min(tm - first_m, gm);
        const int in_g = lid - g * per_group;
        m0 = (first_m + in_g % gsz) * TBM;
        n0 = (in_g / gsz) * TBN;
    }

    // LDS-DMA through buffer descriptors: the per-lane byte offset of every piece is computed once
    // (row clamp + source-side swizzle), the K position travels in the scalar offset, so issuing a
    // stage costs two SALU ops + one buffer_load...lds per 1 KiB piece and no VALU.
    constexpr int GA = TBM / 8 / NW;                  // pieces i < GA come from A, the rest from W
    static_assert((TBM / 8) % NW == 0, "A pieces must split evenly over the waves");
    // TN = 2 (data gradients: out[M][N] = sum_k A[m][k] . W[k][n], W = the nn.Linear weight [out_features = k][in_features = n] as
    // stored): A is staged and read as in the NT form, W as in the TN form -- no transposed weight copies.
    constexpr bool TA = TN == 1, TW = TN != 0;        // which operands have the contraction index as their slow dimension
    static_assert(!TN || (sizeof(T) == 2 && TBM % 64 == 0 && TBN % 64 == 0), "TN / NN forms: bf16");
    static_assert(TN != 1 || EPI == CPT_EPI_NONE, "TN form: plain epilogue");
    static_assert(TN != 2 || EPI == CPT_EPI_NONE || EPI == CPT_EPI_RESID || EPI == CPT_EPI_GELUGRAD, "NN form: plain, residual or GELU-gradient epilogue");
    const int k_rows = (TN && ex.w_rows > 0) ? ex.w_rows : K;     // rows of a transposed-read operand that exist (K rounded up to a K-tile multiple: the rest reads as zero)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)(TA ? k_rows : M) * lda * sizeof(T), (size_t)0x7fffffff), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)(TN ? k_rows : N) * ldw * sizeof(T), (size_t)0x7fffffff), 0x00020000);
    constexpr int CPRA = TBM / 8, CPRB = TBN / 8;      // TN: 16-byte chunks per tile row
    // G <= 6: one VGPR per piece, computed once.  Bigger tiles (registers go to the accumulators): the offsets are rebuilt at
    // every issue from the piece's row (2 VALU per 1 KiB piece); with an even wave count the swizzle term is the same for
    // every piece of a lane.
    constexpr bool VOFF_ARRAY = G <= 6 || ATTN || TN != 0;
    static_assert(!TN || VOFF_ARRAY, "TN form keeps its piece offsets in registers");
    static_assert(VOFF_ARRAY || NW % 2 == 0, "rebuilt offsets need an even wave count");
    const int rbase = wave * 8 + (lane >> 3);
    const unsigned c16 = (unsigned)(((lane & 7) ^ ((rbase >> 1) & 7)) * 16);
    unsigned voff[VOFF_ARRAY ? G : 1];
    if constexpr (!VOFF_ARRAY) voff[0] = 0;
    else
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int g = i * NW + wave;
        const int r = g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if ((i < GA && TA) || (i >= GA && TW)) {
            // piece g = 64 consecutive chunk positions of the row-major [64][CPR] tile; position (row, ch) holds source chunk ch ^ tn_swz(row)
            const int idx = (i < GA ? g : g - TBM / 8) * 64 + lane;
            const int cpr = i < GA ? CPRA : CPRB;
            const int row = idx / cpr, ch = idx - row * cpr;
            const int sch = ch ^ tn_swz(row, cpr);
            voff[i] = i < GA ? (unsigned)(((size_t)row * lda + m0 + sch * 8) * sizeof(T))
                             : (unsigned)(((size_t)row * ldw + n0 + sch * 8) * sizeof(T));
        } else
        if (i < GA) voff[i] = (unsigned)(((size_t)min(m0 + r, M - 1) * lda + c * CE) * sizeof(T));
        else if constexpr (ATTN) {
            const int rw = r - TBM;                   // 0..191 -> row of the fused [3H][K] weight: (q|k|v) block, this head, row in head
            voff[i] = (unsigned)(((size_t)((rw >> 6) * (ex.heads * 64) + att_h * 64 + (rw & 63)) * ldw + c * CE) * sizeof(T));
        }
        else        voff[i] = (unsigned)(((size_t)min(n0 + r - TBM, N - 1) * ldw + c * CE) * sizeof(T));
    }
    auto stage = [&](int slot, int k0) {
        if (abl & 1) return;                          // ablation: no operand traffic
        const int soff = k0 * (int)sizeof(T);
        const int soffA = TA ? k0 * lda * (int)sizeof(T) : soff, soffW = TW ? k0 * ldw * (int)sizeof(T) : soff;   // transposed operand: k0 counts rows
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int g = i * NW + wave;
            auto lds = (__attribute__((address_space(3))) void*)(smem + slot * STAGE_BYTES + g * 1024);
            unsigned vo;
            if constexpr (VOFF_ARRAY) vo = voff[i];
            else if (i < GA) vo = (unsigned)min(m0 + rbase + i * NW * 8, M - 1) * (unsigned)(lda * (int)sizeof(T)) + c16;
            else vo = (unsigned)min(n0 + rbase + i * NW * 8 - TBM, N - 1) * (unsigned)(ldw * (int)sizeof(T)) + c16;
            if (i < GA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lds, 16, vo, soffA, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lds, 16, vo, soffW, 0, 0);
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K: this workgroup owns K-tiles [kt0, kt0 + nt) and adds its partial tile atomically
    const int nt_all = K / BK, nt_per = (nt_all + splitk - 1) / splitk;
    const int kt0 = split * nt_per;
    const int nt = max(0, min(nt_per, nt_all - kt0));
    const int kbase = kt0 * BK;
    // all STAGES slots are filled up front; afterwards tile t+STAGES is issued into tile t's slot right
    // after the mid-iteration barrier of iteration t (every wave has issued its last reads of tile t by then)
#pragma unroll
    for (int p = 0; p < STAGES; ++p)
        if (p < nt) stage(p, kbase + p * BK);

    const int fr = lane & 31, fh = lane >> 5;
    // Register-resident fragments for all four k-steps of a tile; reads run TWO k-steps ahead of the
    // MFMAs that consume them (across the tile boundary too), so LDS latency (~300 cycles with 8
    // waves reading) hides behind two k-steps of matrix work of this wave and its SIMD partner.
    // FD = 4: all four k-steps of a tile live in registers, reads run TWO k-steps ahead.
    // FD = 2: ping-pong buffers, reads run one k-step ahead (for big wave tiles, where six MFMAs
    //         per k-step per wave already cover the LDS latency and registers are the scarce resource).
    frag_t fa[FD][MI], fb[FD][NJ];
    auto ldfrag = [&](int slot, int ks, int pb) {
        if (abl & 2) return;                          // ablation: no LDS fragment reads
        const unsigned char* sa = smem + slot * STAGE_BYTES;
        const unsigned char* sw = sa + TBM * ROWB;
        if constexpr (TN) {
            // contraction rows ks * 16 + 8 fh + [0, 8) of tile column (block start + lane & 31): two transpose reads, 4 rows apart
            // (same row & 3, hence the same swizzle term and a constant second address)
            const int row = ks * 16 + 8 * fh + ((lane & 15) >> 2);
            const int cl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            // The reads are INLINE ASM: through the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of every transpose read while
            // an LDS-DMA is in flight (it cannot tell the tile being read from the one being filled), which serialises the ring
            // (measured: 2200 instead of 1450 cycles per K-tile).  The compiler therefore does not track these reads either:
            // touch() waits for them with explicit counted lgkmcnt.
            auto tr8 = [&](const unsigned char* tile, auto cpr_tag, int col) {
                constexpr int cpr = decltype(cpr_tag)::value;
                const unsigned q = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(tile + (row * cpr + ((col >> 3) ^ tn_swz(row, cpr))) * 16 + (col & 7) * 2);
                u32x2_t lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(q));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(q), "n"(4 * cpr * 16));
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t o = {lo[0], lo[1], hi[0], hi[1]};
                return __builtin_bit_cast(bf16x8, o);
            };
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (TA) fa[pb][i] = tr8(sa, std::integral_constant<int, CPRA>{}, wm * (MI * 32) + i * 32 + cl);
                else {      // NN form: the A fragment as in the NT form, but asm too (every LDS read of the loop must be counted by touch())
                    const unsigned q = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(sa + lds_off(wm * (MI * 32) + i * 32 + fr, ks * 2 + fh));
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    u32x4_t v;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(q));
                    fa[pb][i] = __builtin_bit_cast(bf16x8, v);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[pb][j] = tr8(sw, std::integral_constant<int, CPRB>{}, wn * (NJ * 32) + j * 32 + cl);
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[pb][i] = *reinterpret_cast<const frag_t*>(sa + lds_off(wm * (MI * 32) + i * 32 + fr, ks * 2 + fh));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[pb][j] = *reinterpret_cast<const frag_t*>(sw + lds_off(wn * (NJ * 32) + j * 32 + fr, ks * 2 + fh));
    };
    auto mma = [&](int pb) {
        if (abl & 4) return;                          // ablation: no MFMA
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (ATTN || DIRECT) mfma_chunk(acc[i][j], fb[pb][j], fa[pb][i]);   // transposed tile: lane = row (token), registers = output columns
                else mfma_chunk(acc[i][j], fa[pb][i], fb[pb][j]);
            }
    };
    // make the compiler place its lgkmcnt wait for buffer `pb` HERE (before younger ds_reads are
    // issued) instead of in front of the MFMAs that consume it
    auto touch = [&](int pb) {
        if constexpr (TN) {
            // asm transpose reads (see ldfrag): buffers are read in the order 0, 1, 2, 3, 0, ... and LDS reads return in order,
            // so buffer pb has landed once at most the reads of the buffer issued after it are outstanding (none after buffer 3)
            static_assert(!TN || FD == 4, "TN form: four fragment buffers");
            if (pb < 3) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((TA ? 2 : 1) * MI + 2 * NJ));
            else asm volatile("s_waitcnt lgkmcnt(0)");
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(fa[pb][i]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(fb[pb][j]));
    };
    // wait until tile `t` has landed, called when tiles up to index `last` have been issued:
    // (last - t) younger tiles may stay in flight
    auto wait_tile = [&](int t, int last) {
        const int after = last - t;
        if (after >= 3) wait_vmcnt<3 * G>();
        else if (after == 2) wait_vmcnt<2 * G>();
        else if (after == 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
    };
#define CPT_SB() __builtin_amdgcn_sched_barrier(0)

    if (trace) tr1 = clock64();
    if (nt > 0) {
        wait_tile(0, min(nt, STAGES) - 1);
        __builtin_amdgcn_s_barrier();
        CPT_SB();
        ldfrag(0, 0, 0);
        if (FD == 4) ldfrag(0, 1, 1);
        CPT_SB();
    }
    int slot = 0;
    // MAIN iterations (every condition true) are one basic block, so the compiler's waitcnt pass
    // counts outstanding LDS reads exactly; the last STAGES-1 tiles run the guarded version.
    auto body = [&](int t, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const bool more = MAIN || (t + 1 < nt);
        int nslot = slot + 1;
        if (nslot == STAGES) nslot = 0;
        auto refill = [&]() {      // tile t+STAGES into the slot tile t occupies (pipeline hazards)
            if (MAIN || t + STAGES < nt) stage(slot, kbase + (t + STAGES) * BK);
        };
        if constexpr (FD == 4) {
            touch(0); CPT_SB(); ldfrag(slot, 2, 2); CPT_SB(); mma(0); CPT_SB();
            touch(1); CPT_SB(); ldfrag(slot, 3, 3); CPT_SB(); mma(1); CPT_SB();
            // WAR safety by ORDERING, not by latency: this wave's last reads of tile t (k-steps 2, 3) are RETIRED
            // (lgkmcnt wait placed by touch) before it arrives at the barrier, and no wave refills tile t's slot
            // before every wave has arrived.  (Round 1 retired them after the refill had been issued.)
            touch(2); touch(3); CPT_SB();
            if (more) {
                if (MAIN) wait_vmcnt<(STAGES - 2) * G>(); else wait_tile(t + 1, min(nt, t + STAGES) - 1);
                CPT_SB();
                __builtin_amdgcn_s_barrier();      // tile t+1 visible to all waves; nobody still reads tile t
                CPT_SB();
                refill();
                CPT_SB();
            }
            if (more) ldfrag(nslot, 0, 0); CPT_SB(); mma(2); CPT_SB();
            if (more) ldfrag(nslot, 1, 1); CPT_SB(); mma(3); CPT_SB();
        } else if constexpr (FD == 1) {
            // one fragment buffer (big tiles at four waves per SIMD, where registers go to the accumulators and the other
            // three waves of the SIMD cover this wave's LDS latency): read k-step ks+1 right after the MFMAs of ks issue
            touch(0); CPT_SB(); mma(0); CPT_SB(); ldfrag(slot, 1, 0); CPT_SB();
            touch(0); CPT_SB(); mma(0); CPT_SB(); ldfrag(slot, 2, 0); CPT_SB();
            touch(0); CPT_SB(); mma(0); CPT_SB(); ldfrag(slot, 3, 0); CPT_SB();
            touch(0); CPT_SB();                    // this wave's reads of tile t are all retired
            if (more) {
                if (MAIN) wait_vmcnt<(STAGES - 2) * G>(); else wait_tile(t + 1, min(nt, t + STAGES) - 1);
                CPT_SB();
                __builtin_amdgcn_s_barrier();      // tile t+1 visible to all waves; nobody still reads tile t
                CPT_SB();
                refill();
                CPT_SB();
            }
            mma(0); CPT_SB();
            if (more) { ldfrag(nslot, 0, 0); CPT_SB(); }
        } else {
            touch(0); CPT_SB(); ldfrag(slot, 1, 1); CPT_SB(); mma(0); CPT_SB();
            touch(1); CPT_SB(); ldfrag(slot, 2, 0); CPT_SB(); mma(1); CPT_SB();
            touch(0); CPT_SB(); ldfrag(slot, 3, 1); CPT_SB(); mma(0); CPT_SB();
            touch(1); CPT_SB();                    // this wave's reads of tile t are all done
            if (more) {
                if (MAIN) wait_vmcnt<(STAGES - 2) * G>(); else wait_tile(t + 1, min(nt, t + STAGES) - 1);
                CPT_SB();
                __builtin_amdgcn_s_barrier();      // tile t+1 visible to all waves; nobody still reads tile t
                CPT_SB();
                refill();
                CPT_SB();
                ldfrag(nslot, 0, 0);
                CPT_SB();
            }
            mma(1); CPT_SB();
        }
        slot = nslot;
    };
    const int t_main = max(nt - STAGES, 0);      // iterations that still have a tile to issue
    for (int t = 0; t < t_main; ++t) body(t, std::true_type{});
    for (int t = t_main; t < nt; ++t) body(t, std::false_type{});
#undef CPT_SB

    // ---- epilogue: per-WAVE staging through LDS, no workgroup barriers --------------------------------
    // Each wave owns a private 16-row x (NJ*32)-column fp32 slab in LDS.  For every 16-row slice of its
    // accumulators it writes the slab (ds_write_b32, MFMA layout), waits for its own writes, and reads it
    // back as whole rows (ds_read_b128) to apply bias / GELU / residual and store 8-16 bytes per lane on
    // contiguous row segments.  Waves run this independently; residual loads are software-pipelined one
    // slice ahead so no load is ever issued behind a store (stores count in vmcnt on gfx950).
    if (trace) tr2 = clock64();
    constexpr int WCOLS = NJ * 32;                    // columns of this wave's sub-tile
    constexpr int CPW = WCOLS * 4 + 16;               // slab pitch (bytes)
    constexpr int CH = WCOLS / 4;                     // float4 chunks per slab row
    constexpr int NIT = (16 * CH + 63) / 64;          // read-back iterations per slice
    constexpr int NSL = MI * 2;                       // 16-row slices per wave
    constexpr int SIDE = MI * 32 * 8 + 2 * WCOLS * 4; // per-wave side area: (mean, rstd) per row + two column vectors
    static_assert(DIRECT || (16 * CPW + SIDE) * NW <= STAGES * STAGE_BYTES, "per-wave slabs must fit in the ring");
    static_assert(DIRECT || (MI * 32 <= 64 && CH <= 64), "side area is filled by one wave pass");
    if (split != 0) bias = nullptr;                   // split-K: the bias is added once
    if constexpr (TN != 0 || EPI == CPT_EPI_NONE) out += (size_t)split * (ex.split_stride ? ex.split_stride : (size_t)M * ldo); // split-K (TN / NN forms, gemm_nt_split): every split writes its own partial matrix; split = 0 when K is not split: every split writes its own partial matrix (reduced in slot order afterwards)
    if constexpr (DIRECT) {
        // ---- direct epilogue: the accumulators are TRANSPOSED (operands swapped in mma): lane = output row
        // (wrow0 + 32 i + (lane & 31)), register quad g of block j = the four consecutive columns
        // wcol0 + 32 j + 8 g + 4 (lane >> 5) + [0, 4).  Per-row LayerNorm statistics are therefore per-lane scalars, the
        // per-column vectors come from a 1 KB per-wave LDS side area as 16-byte broadcasts, every output quad is
        // finished in registers and leaves as 16-byte stores (fp32 quads as they are; bf16 quads of the two half-waves
        // paired by v_permlane32_swap).  No slab round trip, no LDS waits in the math, all quads independent.
        constexpr bool LNPROD = EPI == CPT_EPI_LNPROD;
        constexpr bool GELU = EPI == CPT_EPI_LNCONS_GELU;
        static_assert(NW * 3 * NJ * 32 * 4 <= STAGES * STAGE_BYTES, "side areas must fit in the ring");
        const int wrow0 = m0 + wm * (MI * 32), wcol0 = n0 + wn * (NJ * 32);
        constexpr int WC = NJ * 32;
        const bool fold_resid = LNPROD && ex.g_in != nullptr;
        float* sd = reinterpret_cast<float*>(smem) + wave * (3 * WC);
        // operand loads are issued ahead of the barrier that frees the ring
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0;
        if constexpr (LNPROD) v1 = f32x4{1.f, 1.f, 1.f, 1.f};
        {
            const int c = wcol0 + lane * 4;
            if (lane < WC / 4 && c < N) {
                if constexpr (LNPROD) {
                    if (bias) v0 = *reinterpret_cast<const f32x4*>(bias + c);
                    if (fold_resid) { v1 = *reinterpret_cast<const f32x4*>(ex.g_in + c); v2 = *reinterpret_cast<const f32x4*>(ex.b_in + c); }
                } else {
                    v0 = *reinterpret_cast<const f32x4*>(ex.colc + c);
                    v1 = *reinterpret_cast<const f32x4*>(ex.cold + c);
                }
            }
        }
        // row statistics -> (mean, rstd) of row block i; block 0 ahead of the barrier, the others inside the row loop
        // (one block's 64 bytes of partial sums in flight at a time: registers belong to the accumulators here)
        auto row_stats = [&](int i, float& m_o, float& r_o) {
            m_o = 0.f; r_o = 1.f;
            if (!LNPROD || fold_resid) {
                float sum, sq;
                sum_parts(ex.st_in, ex.st_in_parts, min(wrow0 + i * 32 + fr, M - 1), sum, sq);
                ln_mean_rstd(sum, sq, ex.inv_h, ex.eps, m_o, r_o);
            }
        };
        float mu0, rs0;
        row_stats(0, mu0, rs0);
        __syncthreads();                              // every wave is done reading the operand ring
        if (trace) tr3 = clock64();
        if (lane < WC / 4) {
            *reinterpret_cast<f32x4*>(sd + lane * 4) = v0;
            *reinterpret_cast<f32x4*>(sd + WC + lane * 4) = v1;
            if constexpr (LNPROD) *reinterpret_cast<f32x4*>(sd + 2 * WC + lane * 4) = v2;
        }
        // (LDS operations of one wave execute in issue order: its reads below see these writes)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        T* out_lp = LNPROD ? reinterpret_cast<T*>(ex.out_lp) : reinterpret_cast<T*>(out);
        auto run = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wrow0 + i * 32 + fr;
                const bool rok = FULL || row < M;
                const size_t rowc = (size_t)(FULL ? row : min(row, M - 1));
                f32x4 rq[LNPROD ? NJ * 4 : 1];
                if constexpr (LNPROD) {
                    if (!(abl & 128)) {
#pragma unroll
                    for (int q = 0; q < NJ * 4; ++q) {
                        const int c = wcol0 + (q >> 2) * 32 + 8 * (q & 3) + 4 * fh;
                        rq[q] = *reinterpret_cast<const f32x4*>(resid + rowc * ldr + (FULL ? c : min(c, N - 4)));
                    }
                    }
                }
                float sm = 0.f, sq = 0.f;
                float tsm[2] = {0.f, 0.f}, tsq[2] = {0.f, 0.f};      // row sums in the library's one order (common.h rowsum_chunk_pair)
                float m_i = mu0, r_i = rs0;
                if (i > 0) row_stats(i, m_i, r_i);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        u32x2 pk[2];
                        f32x4 xs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int g = 2 * gp + h;
                            const int lc = j * 32 + 8 * g + 4 * fh, c = wcol0 + lc;
                            const bool cok = FULL || c < N;
                            f32x4 x;
                            if constexpr (LNPROD) {
                                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sd + lc);
                                const f32x4 g4 = *reinterpret_cast<const f32x4*>(sd + WC + lc);
                                const f32x4 t4 = *reinterpret_cast<const f32x4*>(sd + 2 * WC + lc);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    x[e] = acc[i][j][4 * g + e] + b4[e] + ln_apply(rq[j * 4 + g][e], m_i, r_i, g4[e], t4[e]);
                                if (rok && cok && !(abl & 64)) *reinterpret_cast<f32x4*>(out + (size_t)row * ldo + c) = x;
#pragma unroll
                                for (int e = 0; e < 4; ++e) xs[h][e] = cok ? x[e] : 0.f;
                            } else {
                                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sd + lc);
                                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + WC + lc);
#pragma unroll
                                for (int e = 0; e < 4; ++e) x[e] = ln_fold(acc[i][j][4 * g + e], m_i, r_i, c4[e], d4[e]);
                                if constexpr (GELU) {
                                    if (!(abl & 16)) {
                                    const f32x2 g0 = gelu_fast2(f32x2{x[0], x[1]}), g1 = gelu_fast2(f32x2{x[2], x[3]});
                                    x = f32x4{g0[0], g0[1], g1[0], g1[1]};
                                    }
                                }
                            }
                            bf16x4 p4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) p4[e] = (bf16)x[e];
                            pk[h] = __builtin_bit_cast(u32x2, p4);
                        }
                        // half-wave exchange: lanes 0-31 end up with columns [16 gp, 16 gp + 8) of block j, lanes 32-63 with the next 8
                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                        const int c8 = wcol0 + j * 32 + 16 * gp + 8 * fh;
                        if (rok && (FULL || c8 < N) && !(abl & 64)) *reinterpret_cast<u32x4*>(out_lp + (size_t)row * ldo + c8) = w;
                        if constexpr (LNPROD) {
                            float ps, pq;
                            rowsum_chunk_pair(xs[0], xs[1], ps, pq);
                            if (j == 0) { tsm[gp] = ps; tsq[gp] = pq; } else { tsm[gp] += ps; tsq[gp] += pq; }
                        }
                    }
                if constexpr (LNPROD) {
                    sm = tsm[0] + tsm[1]; sq = tsq[0] + tsq[1];
                    sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
                    static_assert(!LNPROD || WC == 96, "statistics slots are 96 columns wide");
                    if (fh == 0 && rok)
                        *reinterpret_cast<float2*>(ex.st_out + 2 * ((size_t)row * ex.st_out_slots + wcol0 / WC)) = float2{sm, sq};
                }
            }
        };
        if (abl & 8) { if (acc[0][0][0] == 12345.678f) out[0] = from_f32<OT>(1.f); }     // ablation: no epilogue
        else if (wrow0 + MI * 32 <= M && wcol0 + WC <= N) run(std::true_type{});
        else run(std::false_type{});
    } else {
    __syncthreads();                                  // every wave is done reading the operand ring
    if (trace) tr3 = clock64();
    if constexpr (ATTN) {
        // ---- fused attention: the finished Q | K | V tiles go to LDS as bf16 (never to HBM), then four waves run
        // the attention core on 32 queries each.  The accumulators are TRANSPOSED (lane = token, register quads =
        // four consecutive output columns), so every LDS write is one 8-byte bf16x4.
        unsigned char* sQ = smem;
        unsigned char* sK = smem + 128 * 128;
        unsigned char* sV = smem + 2 * 128 * 128;
        float* sMask = reinterpret_cast<float*>(smem + 2 * 128 * 128 + 128 * ATT_VP16);
        static_assert(2 * 128 * 128 + 128 * ATT_VP16 + 128 * 4 <= STAGES * STAGE_BYTES, "attention tiles must fit in the ring");
        const int Ls = ex.seq_len, hd = ex.heads;
        const int trow = wm * 32 + fr;                // token row inside the tile
        float mu = 0.f, rs = 1.f;
        if constexpr (EPI == CPT_EPI_ATTN_LN) {
            float sm, sq;
            sum_parts(ex.st_in, ex.st_in_parts, min(m0 + trow, M - 1), sm, sq);
            ln_mean_rstd(sm, sq, ex.inv_h, ex.eps, mu, rs);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 96 + j * 32 + 8 * g + 4 * fh;               // tile column of this register quad
                const int which = col >> 6, cc = col & 63;                       // 0 Q, 1 K, 2 V; column inside the head
                const int gcol = which * (hd * 64) + att_h * 64 + cc;            // row of the fused weight / bias vectors
                f32x4 x;
                if constexpr (EPI == CPT_EPI_ATTN_LN) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(ex.colc + gcol);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(ex.cold + gcol);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = ln_fold(acc[0][j][4 * g + e], mu, rs, c4[e], d4[e]);
                } else {
                    const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + gcol) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = acc[0][j][4 * g + e] + b4[e];
                }
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16)x[e];
                unsigned char* dst = which == 2 ? sV + trow * ATT_VP16 + cc * 2
                                                : (which == 1 ? sK : sQ) + att_koff16(trow, cc >> 3) + (cc & 7) * 2;
                *reinterpret_cast<bf16x4*>(dst) = pk;
            }
        if (tid < 128) {
            float mv = -INFINITY;                     // rows past the sequence (next sequence's tokens): not keys
            if (tid < Ls) mv = ex.mask ? (1.0f - (float)ex.mask[(size_t)att_b * Ls + tid]) * (-10000.0f * ATT_LOG2E) : 0.f;
            sMask[tid] = mv;
        }
        __syncthreads();
        if (wave < 4 && wave * 32 < Ls && !(abl & 32)) {   // one attention wave per SIMD (abl 32: projection only)
            const int q = wave * 32 + fr;
            bf16x8 fq[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const bf16x8*>(sQ + att_koff16(q, 2 * ks + fh));
            bf16* crow = reinterpret_cast<bf16*>(out) + ((size_t)m0 + min(q, Ls - 1)) * ldo + att_h * 64;
            attn_core_bf16<4>(fq, sK, sV, sMask, lane, q < Ls, crow, nullptr, Ls);
        }
    } else {
    unsigned char* slab = smem + wave * (16 * CPW);
    unsigned char* side = smem + NW * (16 * CPW) + wave * SIDE;
    const int wrow0 = m0 + wm * (MI * 32), wcol0 = n0 + wn * WCOLS;
    constexpr bool R3 = EPI == CPT_EPI_LNPROD3;                  // residual stream in the 3-byte form (r3_encode), in and out
    constexpr bool LNPROD = EPI == CPT_EPI_LNPROD || R3;
    constexpr bool LNCONS = EPI == CPT_EPI_LNCONS || EPI == CPT_EPI_LNCONS_GELU;
    constexpr bool GG = EPI == CPT_EPI_GELUGRAD;       // the "residual" operand is u (compute dtype): the result is multiplied by gelu'(u)
    constexpr bool HAS_RESID = EPI == CPT_EPI_RESID || EPI == CPT_EPI_RESID_LP || LNPROD || GG;
    constexpr bool GELU2 = EPI == CPT_EPI_GELU2;      // pre-activation stored too; the GELU is taken of the STORED (bf16-rounded) value, as the two-kernel form did
    static_assert(!GELU2 || (sizeof(T) == 2 && sizeof(OT) == 2), "GELU2: bf16 in, bf16 out");
    constexpr bool X3 = EPI == CPT_EPI_GELU_X3;
    static_assert(!X3 || (sizeof(T) == 2 && sizeof(OT) == 2), "GELU_X3: bf16 in, bf16 split copy out");
    constexpr bool DO_GELU = EPI == CPT_EPI_GELU || EPI == CPT_EPI_LNCONS_GELU || GELU2 || X3;
    const T* resid_lp = reinterpret_cast<const T*>(resid);      // EPI_RESID_LP: same rows, compute dtype; LNPROD3: the hi part
    const bool fold_resid = LNPROD && ex.g_in != nullptr;       // residual = LayerNorm(resid; st_in, g_in, b_in)
    // fp32 outputs without residual (e.g. the vocabulary decoder, ldo = 30522): rows are only 8-byte aligned, but 16-byte
    // stores to dword-aligned addresses are legal (the HSA target runs in unaligned-access mode) -- no alignment demand.
    // Everything else keeps natural alignment of its 8/16-byte accesses.
    typedef f32x4 f32x4_u __attribute__((aligned(4)));
    const bool out_ok = (sizeof(OT) == 4 && !HAS_RESID) ? true
                        : ((N % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)out) % 16 == 0));
    const bool vec_ok = out_ok && (!HAS_RESID || (ldr % 4 == 0 && ((uintptr_t)resid) % 16 == 0)) &&
                        (!bias || ((uintptr_t)bias) % 16 == 0);
    static_assert((16 * CH) % 64 == 0 && CH % 4 == 0, "slab read-back must fill whole waves");
    // row statistics -> (mean, rstd)
    auto stats_of = [&](int row, float& mu, float& rs) {      // adds the partial sums of `row` in slot order
        float sum, sq;
        sum_parts(ex.st_in, ex.st_in_parts, row, sum, sq);
        ln_mean_rstd(sum, sq, ex.inv_h, ex.eps, mu, rs);
    };
    // element-wise finish shared by the guarded path: a = accumulator, returns the output value
    auto finish1 = [&](float a, int row, int col) {
        float x;
        if constexpr (LNCONS) {
            float mu, rs;
            stats_of(row, mu, rs);
            x = ln_fold(a, mu, rs, ex.colc[col], ex.cold[col]);
        } else {
            x = a + (bias ? bias[col] : 0.f);
        }
        if constexpr (GELU2) {
            const T ub = from_f32<T>(x);
            reinterpret_cast<T*>(ex.out_lp)[(size_t)row * ldo + col] = ub;
            x = to_f32(ub);
        }
        if (DO_GELU) x = gelu_for<T>(x);
        if (EPI == CPT_EPI_TANH) x = tanhf(x);
        if (EPI == CPT_EPI_RESID) x += resid[(size_t)row * ldr + col];
        if (EPI == CPT_EPI_RESID_LP) x += to_f32(resid_lp[(size_t)row * ldr + col]);
        if constexpr (GG) x *= gelu_grad_for<T>(to_f32(resid_lp[(size_t)row * ldr + col]));
        if constexpr (LNPROD) {
            float r;
            if constexpr (R3) r = r3_decode1(reinterpret_cast<const bf16*>(resid)[(size_t)row * ldr + col], ex.resid_lo[(size_t)row * ldr + col]);
            else r = resid[(size_t)row * ldr + col];
            if (fold_resid) {
                float mu, rs;
                stats_of(row, mu, rs);
                r = ln_apply(r, mu, rs, ex.g_in[col], ex.b_in[col]);
            }
            x += r;
        }
        return x;
    };
    // Three instances.  MODE 1 (interior sub-tile, aligned pointers): one basic block, no guards, so the compiler's vmcnt
    // bookkeeping is exact and stores issue back to back.  MODE 2 (sub-tile that crosses the matrix edge, aligned pointers,
    // N % 4 == 0): the SAME vector arithmetic with clamped load addresses and predicated stores -- a row must get the
    // same bits whether its sub-tile is interior or not (round 1's element-wise edge path differed in the last bit for
    // the LayerNorm producers when M % 32 != 0).  MODE 0: element-wise path for unaligned operands / ragged quads.
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FULL = MODE != 0;             // vector path
        constexpr bool GUARD = MODE == 2;
        auto ccol = [&](int c) { return GUARD ? min(c, N - 4) : c; };
        auto crow = [&](int r) { return GUARD ? min(r, M - 1) : r; };
        // per-column operands: the column chunk of read-back iteration `it` is ((it*64 + lane) % CH), which
        // repeats with period P = CH / gcd(64, CH) in `it` -- P register sets serve all NIT iterations
        constexpr int P = CH / gcd_c(64, CH);
        f32x4 bv[P], cv[LNCONS ? P : 1];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int col = ccol(wcol0 + ((q * 64 + lane) % CH) * 4);
            if constexpr (LNCONS) {
                bv[q] = FULL ? *reinterpret_cast<const f32x4*>(ex.cold + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                cv[q] = FULL ? *reinterpret_cast<const f32x4*>(ex.colc + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                bv[q] = (FULL && bias) ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // LayerNorm-folding modes: (mean, rstd) of this wave's rows and the residual LayerNorm's gain/bias go to a
        // small per-wave LDS side area once, ahead of every store; the slice loop reads them with ds_read
        float2* side_row = reinterpret_cast<float2*>(side);
        float* side_g = reinterpret_cast<float*>(side + MI * 32 * 8);
        float* side_t = side_g + WCOLS;
        if constexpr (FULL && (LNCONS || LNPROD)) {
            if (lane < MI * 32) {
                float2 ms = {0.f, 1.f};
                if (LNCONS || fold_resid) {
                    stats_of(crow(wrow0 + lane), ms.x, ms.y);
                }
                side_row[lane] = ms;
            }
            if constexpr (LNPROD) {
                if (lane < CH) {
                    const f32x4 g4 = fold_resid ? *reinterpret_cast<const f32x4*>(ex.g_in + ccol(wcol0 + lane * 4)) : f32x4{1.f, 1.f, 1.f, 1.f};
                    const f32x4 t4 = fold_resid ? *reinterpret_cast<const f32x4*>(ex.b_in + ccol(wcol0 + lane * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(side_g + lane * 4) = g4;
                    *reinterpret_cast<f32x4*>(side_t + lane * 4) = t4;
                }
            }
        }
        // residual rows, loaded one slice ahead of their use
        struct Aux { f32x4 r[(HAS_RESID && !R3) ? NIT : 1]; u32x2_t h[R3 ? NIT : 1]; unsigned l[R3 ? NIT : 1]; };
        auto load_aux = [&](int sl, Aux& a) {
            if (abl & 128) return;                    // ablation: no residual loads
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane, rr = idx / CH, ch = idx % CH;
                const int row = crow(wrow0 + sl * 16 + rr);
                if constexpr (HAS_RESID) {
                    const size_t off = (size_t)row * ldr + ccol(wcol0 + ch * 4);
                    if constexpr (R3) {                // raw loads only: decoding here would wait for them one slice early
                        a.h[it] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16*>(resid) + off);
                        a.l[it] = *reinterpret_cast<const unsigned*>(ex.resid_lo + off);
                    } else
                    if constexpr ((EPI == CPT_EPI_RESID_LP || GG) && sizeof(T) == 2) {
                        const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(resid_lp + off);
                        a.r[it] = f32x4{(float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]};
                    } else if constexpr (EPI == CPT_EPI_RESID_LP || GG) {
                        a.r[it] = *reinterpret_cast<const f32x4*>(resid_lp + off);
                    } else {
                        a.r[it] = *reinterpret_cast<const f32x4*>(resid + off);
                    }
                }
            }
        };
        Aux aux_a, aux_b;
        if constexpr (FULL && HAS_RESID) load_aux(0, aux_a);
        // GELUGRAD: column sums of the finished values.  A lane meets only P distinct column chunks over the read-back iterations
        // (as for bv above), so the sums ride in P register quads through all slices; the lanes that share a chunk are combined
        // once at the end through the wave's LDS side area (P * 4 ds_add_f32 per lane -- LDS float atomics inside the loop cost
        // 57 us per launch), then one global atomic per column.  MEASURED (FFN-down data gradient, M = 3840): the 184 k global atomics
        // of the launch (1900 per 128-byte line, all at the end of its one round) add 16 us, the stand-alone column-sum launch
        // costs 7.7 -- so the training step leaves this off (cpt_set_tuning key 18 bit 0) and keeps the colsum launch for b_in
        f32x4 cs[GG ? P : 1];
#pragma unroll
        for (int q = 0; q < (GG ? P : 1); ++q) cs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int i = sl >> 1, half = sl & 1;
            if (!(abl & 256)) {                       // ablation 256: no slab writes
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int rr = (r8 & 3) + 8 * (r8 >> 2) + 4 * (lane >> 5);
                    *reinterpret_cast<float*>(slab + rr * CPW + (j * 32 + (lane & 31)) * 4) = acc[i][j][half * 8 + r8];
                }
            } else { asm volatile("" :: "v"(acc[i][0][half * 8])); }
            if constexpr (FULL && HAS_RESID) {
                if (sl + 1 < NSL) { if (sl & 1) load_aux(sl + 1, aux_a); else load_aux(sl + 1, aux_b); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's slab writes have landed
            Aux& ax = (sl & 1) ? aux_b : aux_a;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane, rr = idx / CH, ch = idx % CH;
                const int row = wrow0 + sl * 16 + rr, col = wcol0 + ch * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + rr * CPW + ch * 16);
                if constexpr (FULL) {
                  if (!(abl & 16)) {                  // ablation 16: no finishing math
                    float2 ms = {0.f, 1.f};
                    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (LNCONS || LNPROD) ms = side_row[sl * 16 + rr];
                    if constexpr (LNPROD) {
                        g4 = *reinterpret_cast<const f32x4*>(side_g + ch * 4);
                        t4 = *reinterpret_cast<const f32x4*>(side_t + ch * 4);
                    }
                    const float mu = ms.x, rs = ms.y;
                    f32x4 rr4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (R3) rr4 = r3_decode(ax.h[it], ax.l[it]);
                    else if constexpr (HAS_RESID) rr4 = ax.r[it];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (LNCONS) v[e] = ln_fold(v[e], mu, rs, cv[it % P][e], bv[it % P][e]);
                        else v[e] = v[e] + bv[it % P][e];
                    }
                    if constexpr (GELU2) {
                        bf16x4 ub;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ub[e] = (bf16)v[e]; v[e] = (float)ub[e]; }
                        if (!GUARD || (row < M && col < N)) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(ex.out_lp) + (size_t)row * ldo + col) = ub;
                    }
                    if constexpr (DO_GELU && sizeof(T) == 2) {        // bf16 path: packed-fp32 fast GELU on pairs
                        const f32x2 g0 = gelu_fast2(f32x2{v[0], v[1]}), g1 = gelu_fast2(f32x2{v[2], v[3]});
                        v = f32x4{g0[0], g0[1], g1[0], g1[1]};
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e];
                        if constexpr (DO_GELU && sizeof(T) != 2) x = gelu_for<T>(x);
                        if (EPI == CPT_EPI_TANH) x = tanhf(x);
                        if constexpr (LNPROD) x += ln_apply(rr4[e], mu, rs, g4[e], t4[e]);     // mu=0, rs=1, g=1, b=0 when not folded
                        else if constexpr (GG) x *= gelu_grad_for<T>(rr4[e]);
                        else if constexpr (HAS_RESID) x += rr4[e];
                        v[e] = x;
                    }
                  }
                  const bool in_mat = !GUARD || (row < M && col < N);
                  if (abl & 64) { asm volatile("" :: "v"(v)); } else if (in_mat) {   // ablation 64: no global stores
                    if constexpr (R3) {
                        u32x2_t hq; unsigned lq;
                        r3_encode(v, hq, lq);
                        *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16*>(ex.out_lp) + (size_t)row * ldo + col) = hq;
                        *reinterpret_cast<unsigned*>(ex.out_lo + (size_t)row * ldo + col) = lq;
                    } else
                    if constexpr (X3) {          // hi = bf16(x), lo = bf16(x - hi); row layout hi | hi | lo (rowops.hip split3, activation order)
                        bf16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
                        OT* o3 = out + (size_t)row * ldo + col;
                        *reinterpret_cast<bf16x4*>(o3) = hi;
                        *reinterpret_cast<bf16x4*>(o3 + ex.x3_k) = hi;
                        *reinterpret_cast<bf16x4*>(o3 + 2 * ex.x3_k) = lo;
                    } else
                    if constexpr (sizeof(OT) == 2) {
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (bf16)v[e];
                        *reinterpret_cast<bf16x4*>(out + (size_t)row * ldo + col) = pk;
                    } else if constexpr (EPI == CPT_EPI_ATOMIC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(reinterpret_cast<float*>(out) + (size_t)row * ldo + col + e, v[e]);
                    } else {
                        *reinterpret_cast<f32x4_u*>(out + (size_t)row * ldo + col) = v;
                    }
                    if constexpr (GG) {
                        if (ex.colsum) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) cs[it % P][e] += v[e];
                        }
                    }
                    if constexpr (LNPROD && !R3) {
                        T* olp = reinterpret_cast<T*>(ex.out_lp) + (size_t)row * ldo + col;
                        if constexpr (sizeof(T) == 2) {
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (bf16)v[e];
                            *reinterpret_cast<bf16x4*>(olp) = pk;
                        } else {
                            *reinterpret_cast<f32x4*>(olp) = v;
                        }
                    }
                  }
                    if constexpr (LNPROD) {
                        if (GUARD && !in_mat) v = f32x4{0.f, 0.f, 0.f, 0.f};           // outside the matrix: nothing to sum
                        *reinterpret_cast<f32x4*>(slab + rr * CPW + ch * 16) = v;      // finished values back for the row sums
                    }
                } else {
                    f32x4 fin = {0.f, 0.f, 0.f, 0.f};                  // finished values; 0 outside the matrix
                    if (row < M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e < N) {
                                const float x = finish1(v[e], row, col + e);
                                if constexpr (R3) {
                                    bf16 hq; signed char lq;
                                    r3_encode1(x, hq, lq);
                                    reinterpret_cast<bf16*>(ex.out_lp)[(size_t)row * ldo + col + e] = hq;
                                    ex.out_lo[(size_t)row * ldo + col + e] = lq;
                                } else {
                                if constexpr (EPI == CPT_EPI_ATOMIC && sizeof(OT) == 4) atomicAdd(reinterpret_cast<float*>(out) + (size_t)row * ldo + col + e, x);
                                else if constexpr (X3) {
                                    const OT hi1 = from_f32<OT>(x);
                                    OT* o3 = out + (size_t)row * ldo + col + e;
                                    o3[0] = hi1; o3[ex.x3_k] = hi1; o3[2 * ex.x3_k] = from_f32<OT>(x - to_f32(hi1));
                                }
                                else out[(size_t)row * ldo + col + e] = from_f32<OT>(x);
                                if constexpr (LNPROD) reinterpret_cast<T*>(ex.out_lp)[(size_t)row * ldo + col + e] = from_f32<T>(x);
                                if constexpr (GG) { if (ex.colsum && ex.colsum_rows == 0) atomicAdd(ex.colsum + col + e, x); }      // (the partial-row form needs the vector epilogues: gemm_nn checks their conditions)
                                }
                                fin[e] = x;
                            }
                        }
                    }
                    if constexpr (LNPROD) *reinterpret_cast<f32x4*>(slab + rr * CPW + ch * 16) = fin;
                }
            }
            if constexpr (LNPROD) {
                // row sums of the finished slice: 4 lanes per row, CH/4 chunks each, two shuffles, then ONE 8-byte store
                // into this wave's own slot (column block wcol0 / 96) of the partial-sum table
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int r16 = lane >> 2, part = lane & 3;
                static_assert(!LNPROD || WCOLS == 96, "statistics slots are 96 columns wide");
                // the library's ONE order of a row's partial sums over a 96-column block (common.h rowsum_chunk_pair): lane `part` = (h, qq)
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 f0 = *reinterpret_cast<const f32x4*>(slab + r16 * CPW + (j * 8 + 4 * (part >> 1) + (part & 1)) * 16);
                    const f32x4 f1 = *reinterpret_cast<const f32x4*>(slab + r16 * CPW + (j * 8 + 4 * (part >> 1) + 2 + (part & 1)) * 16);
                    float ps, pq;
                    rowsum_chunk_pair(f0, f1, ps, pq);
                    if (j == 0) { sm = ps; sq = pq; } else { sm += ps; sq += pq; }
                }
                sm += __shfl_xor(sm, 2, 64); sq += __shfl_xor(sq, 2, 64);
                sm += __shfl_xor(sm, 1, 64); sq += __shfl_xor(sq, 1, 64);
                const int srow = wrow0 + sl * 16 + r16;
                if (part == 0 && srow < M)
                    *reinterpret_cast<float2*>(ex.st_out + 2 * ((size_t)srow * ex.st_out_slots + wcol0 / WCOLS)) = float2{sm, sq};
            }
            // (LDS operations of one wave execute in issue order: the next slice's writes cannot pass these reads)
        }
        if constexpr (FULL && GG) {
            if (ex.colsum) {
                // round 6: the P quads of every lane go to the wave's (now free) slab as [P][64][4]; column c = chunk ch, element e then adds the lanes
                // that met chunk ch -- lanes l = (ch - 64 q) mod CH, + CH, ... of quad q -- with plain LDS reads (no LDS atomics)
                static_assert(!GG || P * 64 * 16 <= 16 * CPW, "column sums: the wave's slab holds the lanes' quads");
                float* stq = reinterpret_cast<float*>(slab);
#pragma unroll
                for (int q = 0; q < P; ++q) *reinterpret_cast<f32x4*>(stq + (q * 64 + lane) * 4) = cs[q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int c = lane; c < WCOLS; c += 64) {
                    const int ch = c >> 2, e = c & 3;
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const int l0 = ((ch - (q * 64) % CH) + CH) % CH;
                        for (int l = l0; l < 64; l += CH) t += stq[(q * 64 + l) * 4 + e];
                    }
                    if (GUARD && wcol0 + c >= N) continue;
                    if (ex.colsum_rows > 0) {      // this wave's 32 x WCOLS block owns row wrow0 / 32 of the partial table
                        static_assert(!GG || MI <= 2, "column-sum partial rows: one or two 32-row blocks per wave");
                        if ((wrow0 >> 5) < ex.colsum_rows) ex.colsum[(size_t)(wrow0 >> 5) * N + wcol0 + c] = t;
                        if (MI == 2 && (wrow0 >> 5) + 1 < ex.colsum_rows) ex.colsum[(size_t)((wrow0 >> 5) + 1) * N + wcol0 + c] = 0.f;      // (the table has one row per 32 output rows: this wave's 64 rows own two, the sum sits in the first)
                    } else {
                        atomicAdd(ex.colsum + wcol0 + c, t);
                    }
                }
            }
        }
    };
    if (abl & 8) { if (acc[0][0][0] == 12345.678f) out[0] = from_f32<OT>(1.f); }     // ablation: no epilogue
    else if (vec_ok && wrow0 + MI * 32 <= M && wcol0 + WCOLS <= N) epilogue(std::integral_constant<int, 1>{});
    else if (vec_ok && N % 4 == 0 && N >= 4 && EPI != CPT_EPI_ATOMIC) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
    }   // !ATTN
    }   // !DIRECT
    if (trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* t = trace + (size_t)bid_x * 8;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = tr3; t[4] = clock64();
        t[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
        t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        t[7] = bid_x;
    }
#endif
}

template <typename T, int EPI, typename OT, int TBM, int TBN, int WM, int WN, int STAGES, int EP = 1, int FD = 4, int OCC = 1, int TN = 0>
__global__ __launch_bounds__(WM * WN * 64, OCC * ((WM * WN + 3) / 4)) void gemm_pipe_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
    OT* __restrict__ out, int ldo, int M, int N, int K, int splitk, long long* __restrict__ trace_arg, int abl_arg, EpiX ex) {
    gemm_pipe_body<T, EPI, OT, TBM, TBN, WM, WN, STAGES, EP, FD, OCC, TN>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, splitk, trace_arg, abl_arg, ex,
                                                                          (int)blockIdx.x);
}

// Two TN problems (weight gradients of one encoder layer that share the contraction rows) in ONE launch: workgroups [0, wg0) take
// problem 0, the rest problem 1.  96 + 96 tiles (the two FFN weights) fill 192 CUs with the whole contraction each instead of two
// launches of 96 tiles x 2 splits + two reduction launches; 24 + 72 tiles (attention output + Q|K|V) split the contraction
// in two instead of in eight and in three.  wg0 is a multiple of 8, so a workgroup keeps its XCD under the per-problem tile map.
struct TnProb { const bf16* A; const bf16* W; float* out; int lda, ldw, ldo, M, N; size_t split_stride; };
template <int TBN>
__global__ __launch_bounds__(512, 2) void gemm_tn_group2_kernel(TnProb p0, TnProb p1, int K, int splitk, int wg0, EpiX ex) {
    const bool second = (int)blockIdx.x >= wg0;
    const TnProb& p = second ? p1 : p0;
    ex.split_stride = p.split_stride;
    gemm_pipe_body<bf16, CPT_EPI_NONE, float, 128, TBN, 4, 2, 3, 1, 4, 1, 1>(p.A, p.lda, p.W, p.ldw, nullptr, nullptr, 0, p.out, p.ldo, p.M, p.N, K, splitk,
                                                                            nullptr, 0, ex, (int)blockIdx.x - (second ? wg0 : 0));
}

// out[i] = part[0][i] + part[1][i] + ... in split order (reduce_partials_kernel's arithmetic), by `nb` workgroups of NT threads; four elements' loads
// (4 S of them) in flight per thread
template <int NT>
__device__ __forceinline__ void reduce_job_block(const ReduceJob& jb, int b, int nb) {
    const f32x4* __restrict__ part = (const f32x4*)jb.part;
    f32x4* __restrict__ out = (f32x4*)jb.out;
    const size_t n4 = jb.n4, step = (size_t)nb * NT;
    const int S = jb.S;
    for (size_t i0 = (size_t)b * NT + threadIdx.x; i0 < n4; i0 += 4 * step) {
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t i = i0 + u * step; if (i < n4) a[u] = part[i]; }
        for (int k = 1; k < S; ++k) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const size_t i = i0 + u * step; if (i < n4) t[u] = part[(size_t)k * n4 + i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u][0] += t[u][0]; a[u][1] += t[u][1]; a[u][2] += t[u][2]; a[u][3] += t[u][3]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t i = i0 + u * step; if (i < n4) out[i] = a[u]; }
    }
}
__global__ __launch_bounds__(256) void reduce_job_kernel(ReduceJob jb) { reduce_job_block<256>(jb, (int)blockIdx.x, (int)gridDim.x); }
int reduce_job_flush(const ReduceJob& job, hipStream_t s) {
    if (job.S <= 1) return CPT_OK;
    reduce_job_kernel<<<dim3((unsigned)std::min<size_t>((job.n4 + 1023) / 1024, 1024)), dim3(256), 0, s>>>(job);
    return CPT_OK;
}

// ... and three: FFN down | FFN up | attention output (96 + 96 + 24 tiles at hidden 768: 216 CUs with the whole contraction each)
// round 6: workgroups from nmain on (the CUs the tiles leave idle) add up the partial matrices of an EARLIER K-split launch (ReduceJob, kernels.h)
template <int TBN>
__global__ __launch_bounds__(512, 2) void gemm_tn_group3_kernel(TnProb p0, TnProb p1, TnProb p2, int K, int wg0, int wg1, EpiX ex, ReduceJob job, int nmain) {
    const int bid = (int)blockIdx.x;
    if (bid >= nmain) { reduce_job_block<512>(job, bid - nmain, (int)gridDim.x - nmain); return; }
    const int which = bid >= wg1 ? 2 : (bid >= wg0 ? 1 : 0);
    const TnProb& p = which == 2 ? p2 : (which == 1 ? p1 : p0);
    gemm_pipe_body<bf16, CPT_EPI_NONE, float, 128, TBN, 4, 2, 3, 1, 4, 1, 1>(p.A, p.lda, p.W, p.ldw, nullptr, nullptr, 0, p.out, p.ldo, p.M, p.N, K, 1,
                                                                            nullptr, 0, ex, bid - (which == 2 ? wg1 : (which == 1 ? wg0 : 0)));
}

long long* g_gemm_trace = nullptr;
int g_trace_epi = -1, g_trace_k = 0;       // diagnostic builds: stamp only launches of this epilogue / this K (-1 / 0: all)
void set_gemm_trace_filter(int epi, int k) { g_trace_epi = epi == 255 ? -1 : epi; g_trace_k = k; }
CPT_SWITCH(int g_gemm_abl, 0);
CPT_SWITCH(int g_gemm_skew, 0);
void set_gemm_skew(int v) { CPT_SWITCH_SET(g_gemm_skew = v); (void)v; }

template <typename T, int EPI, typename OT, int TBM, int TBN, int WM, int WN, int STAGES, int EP = 1, int FD = 4, int OCC = 1, int TN = 0>
static int launch_pipe(const T* A, int lda, const T* W, int ldw, const float* bias, const float* resid, int ldr,
                       OT* out, int ldo, int M, int N, int K, hipStream_t s, int splitk = 1, const EpiX* ex = nullptr) {
    constexpr int LDS = STAGES * (TBM + TBN) * ROWB;     // the epilogue's per-wave slabs reuse the ring
    auto kern = gemm_pipe_kernel<T, EPI, OT, TBM, TBN, WM, WN, STAGES, EP, FD, OCC, TN>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (LDS > 64 * 1024 && !attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    EpiX e = {};
    if (ex) e = *ex;
    e.skew = g_gemm_skew;
    const int nwg = ((M + TBM - 1) / TBM) * ((N + TBN - 1) / TBN) * splitk;
    long long* tr = ((g_trace_epi < 0 || g_trace_epi == EPI) && (g_trace_k == 0 || g_trace_k == K)) ? g_gemm_trace : nullptr;
    kern<<<dim3(nwg), dim3(WM * WN * 64), LDS, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, splitk, tr, g_gemm_abl, e);
    return CPT_OK;
}

CallOverride& call_override() { static thread_local CallOverride o; return o; }
CPT_SWITCH(int g_gemm_variant_sw, 3);
#define g_gemm_variant (call_override().gemm_variant >= 0 ? call_override().gemm_variant : g_gemm_variant_sw)
// g_gemm_variant: 0: generic register-staged kernel only; 3: pipelined kernel, tile shape chosen per GEMM; 10/11/13/14/15: one fixed shape

// the five tile configurations of the pipelined kernel
#define CPT_CFG_128x192 128, 192, 4, 2, 3
#define CPT_CFG_192x192 192, 192, 6, 2, 3, 2
#define CPT_CFG_128x384 128, 384, 2, 4, 2, 2
#define CPT_CFG_384x192 384, 192, 6, 2, 2, 6, 2
#define CPT_CFG_128x192_OCC2 128, 192, 4, 2, 2, 1, 2, 2
// (round 4: the 4-wave 64x96 form at two workgroups per CU and the 256x192 form, variants 16 / 17, lost every A/B of rounds 1-3
// (profiles/r01_gemm_ladder.md, r03_gemm_variants_standalone.txt) and were removed)
#define CPT_CFG_64x192 64, 192, 2, 2, 3                   // 4 waves of 32x96: twice the workgroups when M is small
#define CPT_CFG_384x256 384, 256, 4, 2, 2, 1, 1, 1        // 8 waves of 96x128 (192 accumulator registers), the whole LDS as a 2-stage ring

CPT_SWITCH(int g_narrow_tiles, 1);      // (values > 1, development build: the workgroup-count threshold itself instead of 128)
#define NARROW_MAX (g_narrow_tiles > 1 ? (long)g_narrow_tiles : 128L)
CPT_SWITCH(int g_narrow_dummy_, 0);      // cpt_set_tuning(39, v): 64 x 96 / 64 x 128 tiles for the FFN-up forward / GELU-gradient GEMMs where 64 x 192 tiles fill at most half the chip
void set_narrow_tiles(int v) { CPT_SWITCH_SET(g_narrow_tiles = v); (void)v; }
template <typename T, int EPI, typename OT>
static void launch_fast(int variant, const T* A, int lda, const T* W, int ldw, const float* bias,
                        const float* resid, int ldr, OT* out, int ldo, int M, int N, int K, hipStream_t s,
                        const EpiX* ex = nullptr) {
    int pick = 0;
#ifndef CPT_DECODER_64x128
#define CPT_DECODER_64x128 1
#endif
    // A few rows against a long table (the MLM decoder: <= 64 [MASK] rows x 30522 x 768, 47 MB of weights streamed once): 64 x 128 tiles = 239
    // workgroups pull the table through 239 CUs instead of the 159 of 64 x 192 tiles; same K order, same bits.
    if constexpr (EPI == CPT_EPI_NONE && sizeof(T) == 2 && CPT_DECODER_64x128) {
        if (variant == 3 && M <= 64 && N >= 8192) {
            launch_pipe<T, EPI, OT, 64, 128, 2, 2, 3>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex);
            return;
        }
    }
    // round 6, few rows (the training forward's FFN-up at 4 sequences per GPU: M = 480, N = 3072 is 128 tiles of 64 x 192): 64 x 96 tiles (2 waves of 32 x 96,
    // the same wave tile) put 256 workgroups on the chip with 20 instead of 32 KB of operands per K-tile
    if constexpr (EPI == CPT_EPI_GELU2 || (EPI == CPT_EPI_NONE && sizeof(T) == 2)) {      // (... and the plain epilogue: the forward Q|K|V at few rows, 96 tiles of 64 x 192)
        if (variant == 3 && g_narrow_tiles && N % 96 == 0 && (long)((M + 63) / 64) * ((N + 191) / 192) <= NARROW_MAX) {
            launch_pipe<T, EPI, OT, 64, 96, 2, 1, 3>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex);
            return;
        }
    }
    if (variant == 3) {
        // Tile shape by a two-term model measured on MI355X (tools/ubench.hip): the K loop is
        // bound by LDS port time (LDS-DMA writes + fragment reads) unless the tile does >= 3.6 MFMA per KiB of
        // operands, and a launch takes ceil(workgroups / slots) rounds.
        struct Cand { int bm, bn, slots, round_cost; };   // round_cost ~ cycles for one round of `slots` workgroups
        constexpr bool big_ok = EPI == CPT_EPI_LNCONS_GELU;   // 384x256 exists for the direct FFN-up epilogue only
        const Cand cand[9] = {{128, 192, 256, 320}, {192, 192, 256, 384}, {128, 384, 256, 512},
                              {384, 192, 256, 645 /* MFMA-bound */},
                              {128, 192, 512, 460 /* two co-resident workgroups per CU: epilogue under the other's K loop */},
                              {0, 0, 1, 0}, {0, 0, 1, 0} /* 5, 6: removed shapes */,
                              {64, 192, 256, 200 /* small M: fills the chip with half-height tiles */},
                              {big_ok ? 384 : 0, 256, 256, 860 /* MFMA-bound, 26 B of operands per MFMA cycle */}};
        long best_cost = -1;
        // residual-type epilogues do not fit the 168 (12 waves) / 128 (two workgroups per CU) register caps without
        // scratch spills, and a spilling instance runs 3-5x slower: those shapes are not candidates for them
        constexpr bool heavy_epi = EPI == CPT_EPI_RESID || EPI == CPT_EPI_RESID_LP || EPI == CPT_EPI_LNPROD || EPI == CPT_EPI_LNPROD3;
        for (int i = 0; i < 9; ++i) {
            // (round 4: the two-per-CU form of the 3-byte LayerNorm producer spills 29 registers, all of them in its epilogue: behind a long
            // contraction it still wins where 128 x 192 tiles would run two rounds -- Oscar-large FFN-down, 402 tiles: 2.65 -> 2.45 ms per step)
#ifndef CPT_OCC2_PROD
#define CPT_OCC2_PROD 1
#endif
            const bool occ2_ok = CPT_OCC2_PROD && EPI == CPT_EPI_LNPROD3 && K >= 2048;
            if (cand[i].bm == 0 || (heavy_epi && (i == 3 || (i == 4 && !occ2_ok)))) continue;
            const long wgs = (long)((M + cand[i].bm - 1) / cand[i].bm) * ((N + cand[i].bn - 1) / cand[i].bn);
            long rc = cand[i].round_cost;
            // round 3 (measured on the Oscar-large shapes, M = 8480, N = 1024): with a residual-type epilogue the 128 x 384 shape (8 waves of
            // 64 x 96 on a 2-stage ring) runs a round in ~2.2x the time of a 128 x 192 round, not 1.6x: attn-out 65 vs 43 us, FFN-down
            // 133 vs 108 us for one round of it against two rounds of 128 x 192
            if (heavy_epi && i == 2) rc = 720;
            const long cost = ((wgs + cand[i].slots - 1) / cand[i].slots) * rc;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; pick = i; }
        }
    } else if (variant == 11) pick = 1;
    else if (variant == 10) pick = 2;
    else if (variant == 14) pick = 3;
    else if (variant == 15) pick = 4;
    else if (variant == 18) pick = 7;
    else if (variant == 19) pick = 8;
    if constexpr (EPI == CPT_EPI_LNCONS_GELU) {
        if (pick == 8) { launch_pipe<T, EPI, OT, CPT_CFG_384x256>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); return; }
    }
    if (pick == 8) pick = 0;
    // (round 4: the shapes whose register caps the residual-type epilogues overflow -- 384 x 192 at 114-254 spilled registers, the two-per-CU form
    // except for the 3-byte producer -- are not instantiated for them any more: a forced variant 14 / 15 runs the 128 x 192 kernel there)
    constexpr bool spills = EPI == CPT_EPI_RESID || EPI == CPT_EPI_RESID_LP || EPI == CPT_EPI_LNPROD || EPI == CPT_EPI_LNPROD3;
    if constexpr (spills) { if (pick == 3 || (pick == 4 && EPI != CPT_EPI_LNPROD3)) pick = 0; }
    if constexpr (!spills) {
        if (pick == 3) { launch_pipe<T, EPI, OT, CPT_CFG_384x192>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); return; }
    }
    if constexpr (!spills || EPI == CPT_EPI_LNPROD3) {
        if (pick == 4) { launch_pipe<T, EPI, OT, CPT_CFG_128x192_OCC2>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); return; }
    }
    switch (pick) {
        case 1: launch_pipe<T, EPI, OT, CPT_CFG_192x192>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); break;
        case 2: launch_pipe<T, EPI, OT, CPT_CFG_128x384>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); break;
        case 7: launch_pipe<T, EPI, OT, CPT_CFG_64x192>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); break;
        default: launch_pipe<T, EPI, OT, CPT_CFG_128x192>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s, 1, ex); break;
    }
}

template <typename T, typename OT>
static int launch_epi(int epi, const T* A, int lda, const T* W, int ldw, const float* bias,
                      const float* resid, int ldr, OT* out, int ldo, int M, int N, int K,
                      hipStream_t s) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(GEMM_THREADS);
    constexpr int BKE = ROWB / (int)sizeof(T);
    if (g_gemm_variant > 0 && K % BKE == 0) {
        switch (epi) {
            case CPT_EPI_NONE: launch_fast<T, CPT_EPI_NONE, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_GELU: launch_fast<T, CPT_EPI_GELU, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_TANH: launch_fast<T, CPT_EPI_TANH, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_RESID:
                if (!resid) return CPT_ERR_SHAPE;
                launch_fast<T, CPT_EPI_RESID, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_RESID_LP:
                if (!resid || g_gemm_variant < 3) return CPT_ERR_SHAPE;
                launch_fast<T, CPT_EPI_RESID_LP, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            default: return CPT_ERR_SHAPE;
        }
    }
    switch (epi) {
        case CPT_EPI_NONE:
            gemm_kernel<T, CPT_EPI_NONE, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_GELU:
            gemm_kernel<T, CPT_EPI_GELU, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_TANH:
            gemm_kernel<T, CPT_EPI_TANH, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_RESID:
            if (!resid) return CPT_ERR_SHAPE;
            gemm_kernel<T, CPT_EPI_RESID, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        default:
            return CPT_ERR_SHAPE;
    }
    return CPT_OK;
}

int gemm(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias,
         const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
         hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return CPT_ERR_SHAPE;
    const int ce = dtype == CPT_BF16 ? 8 : 4;
    if (K % ce || lda % ce || ldw % ce) return CPT_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return CPT_ERR_ALIGN;
    if (dtype == CPT_BF16) {
        if (out_dtype == CPT_BF16)
            return launch_epi<bf16, bf16>(epi, (const bf16*)A, lda, (const bf16*)W, ldw, bias, resid, ldr, (bf16*)out, ldo, M, N, K, s);
        return launch_epi<bf16, float>(epi, (const bf16*)A, lda, (const bf16*)W, ldw, bias, resid, ldr, (float*)out, ldo, M, N, K, s);
    }
    if (dtype == CPT_F32) {
        if (out_dtype != CPT_F32) return CPT_ERR_DTYPE;
        return launch_epi<float, float>(epi, (const float*)A, lda, (const float*)W, ldw, bias, resid, ldr, (float*)out, ldo, M, N, K, s);
    }
    return CPT_ERR_DTYPE;
}

// Region projection of the bf16 path (modeling_bert.py:261 img_embedding: M = B * regions rows, K = 2054 padded to 2112, N = hidden): K split
// over TWO workgroups per 128 x 192 tile -- 200 workgroups of 16-17 K-tiles at B = 64 instead of 200 of 33 with 64-row tiles -- each
// writing its own fp32 partial matrix (out, then out + M * ldo; bias in the first); the LayerNorm pass that follows adds the two
// (layernorm_rows_ex resid).  Always split, whatever M: a row's bits must not depend on the batch it travels in.
int gemm_img_proj(const void* A, int lda, const void* W, int ldw, const float* bias, float* out2, int ldo, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K < 128 || K % 64 || lda % 8 || ldw % 8 || N % 4 || ldo % 4) return CPT_ERR_SHAPE;
    if (!A || !W || !out2) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out2 | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    return launch_pipe<bf16, CPT_EPI_NONE, float, 128, 192, 4, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, out2, ldo, M, N, K, s, 2);
}

// Head transform of the bf16 path (BertPredictionHeadTransform.dense on the B [MASK] rows: 64 x 768 x 768 at B = 64 ran as FOUR workgroups
// of 12 K-tiles in sequence, 14 us): K split over S workgroups per 64 x 192 tile, each writing its own fp32 partial matrix
// partials[S][M][N] (bias in the first); rowops.hip head_finish adds them in split order, applies GELU and the LayerNorm.  S depends on K
// only (two K-tiles per split), never on the batch.
int head_transform_splits(int K) { const int nt = K / 64; return nt >= 4 ? nt / 2 : 1; }
int gemm_head_transform(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K < 64 || K % 64 || lda % 8 || ldw % 8 || N % 4) return CPT_ERR_SHAPE;
    if (!A || !W || !partials) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)partials | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    return launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, partials, N, M, N, K, s, head_transform_splits(K));
}

// The dense layers of the LAST encoder layer on the head's rows only (cpt_abi.hip `tail`, rowops.hip tail_rows): a few rows against a whole weight
// matrix is the head transform's problem again -- K split over S workgroups per 64 x 192 tile, fp32 partial matrices partials[S][M][N] (bias in the
// first) added in split order by the row pass behind (layernorm_rows_ex x_parts / gelu_parts).  S depends on K only (two K-tiles per split, at
// most 24 splits): a row's bits do not depend on the batch it travels in.
int rows_gemm_splits(int K) {
    const int nt = K / 64;
    if (nt < 4) return 1;
    int S = nt / 2;
    while (S > 24 || nt % S) --S;
    return S;
}
int gemm_rows_split(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K < 64 || K % 64 || lda % 8 || ldw % 8 || N % 4) return CPT_ERR_SHAPE;
    if (!A || !W || !partials) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)partials | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    return launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, partials, N, M, N, K, s, rows_gemm_splits(K));
}

CPT_SWITCH(int g_splitk_target, 384);
void set_splitk_target(int v) { CPT_SWITCH_SET(g_splitk_target = v); (void)v; }

// out[M][N] (fp32, caller-zeroed or holding a partial sum) += A[M][K] . W[N][K]^T, K split over
// enough workgroups to fill the chip; used for weight gradients (M, N small; K = rows of the batch)
int gemm_splitk_accum(int dtype, const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return CPT_ERR_SHAPE;
    const int bke = dtype == CPT_BF16 ? 64 : 32;
    if (K % bke || lda % 8 || ldw % 8 || (((uintptr_t)A | (uintptr_t)W) & 15)) return CPT_ERR_ALIGN;
    const int tiles = ((M + 127) / 128) * ((N + 191) / 192);
    const int nt = K / bke;
    int splitk = (g_splitk_target + tiles - 1) / tiles;
    if (splitk > nt) splitk = nt;
    if (splitk < 1) splitk = 1;
    if (dtype == CPT_BF16)
        return launch_pipe<bf16, CPT_EPI_ATOMIC, float, 128, 192, 4, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, nullptr, nullptr, 0, out, ldo, M, N, K, s, splitk);
    if (dtype == CPT_F32)
        return launch_pipe<float, CPT_EPI_ATOMIC, float, 128, 192, 4, 2, 3>((const float*)A, lda, (const float*)W, ldw, nullptr, nullptr, 0, out, ldo, M, N, K, s, splitk);
    return CPT_ERR_DTYPE;
}

// ---- TN form: out[M][N] (fp32) = sum over the K rows of A[k][m] . W[k][n]  (weight gradients: A = dY, W = X) ----------------
// K is split over enough workgroups to fill the chip when the output has few tiles (768 x 768: 24); every split writes its
// own partial matrix into `partials` and a second kernel adds them in split order (deterministic, no atomics, no zeroing).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ out, size_t n4, int S,
                                                              const f32x4* __restrict__ resid = nullptr) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = part[i];
        // (round 6: eight partial matrices' loads in flight at a time, added in split order -- the one-by-one loop made a 16-way split of the head's 64 x 768
        // data gradient a 20 us launch: sixteen memory round trips in sequence per thread)
        for (int k0 = 1; k0 < S; k0 += 8) {
            f32x4 t[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (k0 + kk < S) t[kk] = part[(size_t)(k0 + kk) * n4 + i];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (k0 + kk < S) { a[0] += t[kk][0]; a[1] += t[kk][1]; a[2] += t[kk][2]; a[3] += t[kk][3]; }
        }
        if (resid) { const f32x4 r = resid[i]; a[0] += r[0]; a[1] += r[1]; a[2] += r[2]; a[3] += r[3]; }     // same order as the epilogue: sum, then + residual
        out[i] = a;
    }
}

int gemm_tn_eligible(int M, int N, int K, int lda, int ldw, int ldo) {
    return M > 0 && N > 0 && K > 0 && M % 128 == 0 && (N % 192 == 0 || N % 128 == 0) && K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldo == N;
}

int gemm_tn(const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K, void* partials, size_t partial_bytes,
            hipStream_t s, int k_rows, ReduceJob* defer) {
    if (defer) defer->S = 0;
    if (!gemm_tn_eligible(M, N, K, lda, ldw, ldo)) return CPT_ERR_SHAPE;
    if (k_rows < 0 || k_rows > K) return CPT_ERR_SHAPE;
    EpiX ex = {};
    ex.w_rows = k_rows;        // rows of A / W that exist (0: K); K - k_rows < 64 rows of padding read as zero through the buffer bounds
    if (!A || !W || !out) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)partials) & 15)) return CPT_ERR_ALIGN;
    if ((size_t)K * (size_t)std::max(lda, ldw) * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;      // 32-bit buffer offsets
    const bool w192 = N % 192 == 0;
    const int tiles = (M / 128) * (N / (w192 ? 192 : 128)), nt = K / 64;
    int S = 256 / tiles;
    if (S > 8) S = 8;
    if (S > nt / 6) S = nt / 6;                        // at least six K-tiles per split: a short contraction (4 sequences per GPU: 8 K-tiles) is all
                                                       // pipeline fill, splitting it only adds the partial matrices and the reduction launch
    const size_t mat = (size_t)M * N * 4;
    if (!partials) S = 1;
    while (S > 1 && (size_t)S * mat > partial_bytes) --S;
    if (S < 1) S = 1;
    float* dst = S > 1 ? (float*)partials : out;
    const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)W;
    int rc;
    // (round 6: a short contraction that is not split and fills at most half the chip -- the Q|K|V weight gradient at 4 sequences per GPU, 72 tiles over
    // 8 K-tiles -- runs 64 x 192 tiles: 144 workgroups)
    if (w192 && S == 1 && g_narrow_tiles && tiles <= NARROW_MAX)
        rc = launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3, 1, 4, 1, 1>(a, lda, w, ldw, nullptr, nullptr, 0, dst, ldo, M, N, K, s, S, &ex);
    else
    if (w192) rc = launch_pipe<bf16, CPT_EPI_NONE, float, 128, 192, 4, 2, 3, 1, 4, 1, 1>(a, lda, w, ldw, nullptr, nullptr, 0, dst, ldo, M, N, K, s, S, &ex);
    else      rc = launch_pipe<bf16, CPT_EPI_NONE, float, 128, 128, 4, 2, 3, 1, 4, 1, 1>(a, lda, w, ldw, nullptr, nullptr, 0, dst, ldo, M, N, K, s, S, &ex);
    if (rc != CPT_OK) return rc;
    if (S > 1) {
        const size_t n4 = mat / 16;
        if (defer) { *defer = ReduceJob{(const float*)partials, out, n4, S}; return CPT_OK; }      // (added up by a later launch's spare workgroups)
        const int blocks = (int)std::min<size_t>((n4 + 255) / 256, 2048);
        reduce_partials_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const f32x4*)partials, (f32x4*)out, n4, S);
    }
    return CPT_OK;
}

// two partial-matrix sets reduced by one launch (the two problems of gemm_tn_pair)
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const f32x4* __restrict__ part, size_t stride4, int S, f32x4* __restrict__ out0, size_t n40,
                                                               f32x4* __restrict__ out1, size_t n41) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n40 + n41; i += (size_t)gridDim.x * 256) {
        f32x4 a = part[i];
        for (int k = 1; k < S; ++k) { const f32x4 b = part[(size_t)k * stride4 + i]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
        if (i < n40) out0[i] = a; else out1[i - n40] = a;
    }
}

// Two weight gradients that contract over the same K rows in one launch (gemm_tn_group2_kernel).  CPT_ERR_SHAPE: not applicable
// (shapes of different tile classes, first problem's workgroups not a multiple of 8, ...) -- the caller runs two gemm_tn instead.
int gemm_tn_pair(const void* A0, int lda0, const void* W0, int ldw0, float* out0, int M0, int N0,
                 const void* A1, int lda1, const void* W1, int ldw1, float* out1, int M1, int N1,
                 int K, int k_rows, void* partials, size_t partial_bytes, hipStream_t s) {
    if (!gemm_tn_eligible(M0, N0, K, lda0, ldw0, N0) || !gemm_tn_eligible(M1, N1, K, lda1, ldw1, N1)) return CPT_ERR_SHAPE;
    if (k_rows < 0 || k_rows > K) return CPT_ERR_SHAPE;
    if (!A0 || !W0 || !out0 || !A1 || !W1 || !out1) return CPT_ERR_NULL;
    if ((((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)out0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)out1 | (uintptr_t)partials) & 15)) return CPT_ERR_ALIGN;
    if ((size_t)K * (size_t)std::max(std::max(lda0, ldw0), std::max(lda1, ldw1)) * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;
    const bool w192 = N0 % 192 == 0 && N1 % 192 == 0;
    if (!w192 && (N0 % 128 || N1 % 128)) return CPT_ERR_SHAPE;
    const int tbn = w192 ? 192 : 128;
    const int t0 = (M0 / 128) * (N0 / tbn), t1 = (M1 / 128) * (N1 / tbn), nt = K / 64;
    int S = 256 / (t0 + t1);
    if (S > 8) S = 8;
    if (S > nt / 6) S = nt / 6;
    const size_t mat0 = (size_t)M0 * N0, mat1 = (size_t)M1 * N1;
    if (!partials) S = 1;
    while (S > 1 && (size_t)S * (mat0 + mat1) * 4 > partial_bytes) --S;
    if (S < 1) S = 1;
    if ((t0 * S) % 8) return CPT_ERR_SHAPE;
    EpiX ex = {};
    ex.w_rows = k_rows;
    ex.skew = g_gemm_skew;
    TnProb p0 = {(const bf16*)A0, (const bf16*)W0, out0, lda0, ldw0, N0, M0, N0, 0};
    TnProb p1 = {(const bf16*)A1, (const bf16*)W1, out1, lda1, ldw1, N1, M1, N1, 0};
    if (S > 1) {
        p0.out = (float*)partials; p1.out = (float*)partials + mat0;
        p0.split_stride = p1.split_stride = mat0 + mat1;
    }
    const int nwg = (t0 + t1) * S;
    static bool attr_done_dev[2][CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[w192 ? 0 : 1][current_device_slot()];
#define CPT_TN_G2(TBN_)                                                                                                            \
    do {                                                                                                                           \
        constexpr int LDS = 3 * (128 + TBN_) * ROWB;                                                                               \
        auto kern = gemm_tn_group2_kernel<TBN_>;                                                                                   \
        if (LDS > 64 * 1024 && !attr_done) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            if (e != hipSuccess) return CPT_ERR_HIP - (int)e;                                                                      \
            attr_done = true;                                                                                                      \
        }                                                                                                                          \
        kern<<<dim3(nwg), dim3(512), LDS, s>>>(p0, p1, K, S, t0 * S, ex);                                                          \
    } while (0)
    if (w192) CPT_TN_G2(192); else CPT_TN_G2(128);
#undef CPT_TN_G2
    if (S > 1) {
        const size_t n40 = mat0 / 4, n41 = mat1 / 4;
        const int blocks = (int)std::min<size_t>((n40 + n41 + 255) / 256, 2048);
        reduce_partials2_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const f32x4*)partials, (mat0 + mat1) / 4, S, (f32x4*)out0, n40, (f32x4*)out1, n41);
    }
    return CPT_OK;
}

// Three weight gradients over the same K rows in one launch, each workgroup with the whole contraction (no partial matrices): applicable
// when the three problems' 128 x 192 tiles fit one round of the chip and the block ranges start on multiples of 8 (XCD map).
int gemm_tn_triple_eligible(int M0, int N0, int M1, int N1, int M2, int N2, int K) {
    if (!gemm_tn_eligible(M0, N0, K, 8, 8, N0) || !gemm_tn_eligible(M1, N1, K, 8, 8, N1) || !gemm_tn_eligible(M2, N2, K, 8, 8, N2)) return 0;
    if (N0 % 192 || N1 % 192 || N2 % 192) return 0;
    const int t0 = (M0 / 128) * (N0 / 192), t1 = (M1 / 128) * (N1 / 192), t2 = (M2 / 128) * (N2 / 192);
    return t0 + t1 + t2 <= 256 && t0 % 8 == 0 && (t0 + t1) % 8 == 0;
}
int gemm_tn_triple(const void* A0, int lda0, const void* W0, int ldw0, float* out0, int M0, int N0,
                   const void* A1, int lda1, const void* W1, int ldw1, float* out1, int M1, int N1,
                   const void* A2, int lda2, const void* W2, int ldw2, float* out2, int M2, int N2, int K, int k_rows, hipStream_t s, const ReduceJob* carry) {
    if (!gemm_tn_triple_eligible(M0, N0, M1, N1, M2, N2, K)) return CPT_ERR_SHAPE;
    if (lda0 % 8 || ldw0 % 8 || lda1 % 8 || ldw1 % 8 || lda2 % 8 || ldw2 % 8 || k_rows < 0 || k_rows > K) return CPT_ERR_SHAPE;
    if (!A0 || !W0 || !out0 || !A1 || !W1 || !out1 || !A2 || !W2 || !out2) return CPT_ERR_NULL;
    if ((((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)out0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)out1 | (uintptr_t)A2 | (uintptr_t)W2 | (uintptr_t)out2) & 15)) return CPT_ERR_ALIGN;
    const int mx = std::max(std::max(std::max(lda0, ldw0), std::max(lda1, ldw1)), std::max(lda2, ldw2));
    if ((size_t)K * (size_t)mx * 2 > (size_t)0x7fffffff) return CPT_ERR_SHAPE;
    const int t0 = (M0 / 128) * (N0 / 192), t1 = (M1 / 128) * (N1 / 192), t2 = (M2 / 128) * (N2 / 192);
    EpiX ex = {};
    ex.w_rows = k_rows;
    ex.skew = g_gemm_skew;
    TnProb p0 = {(const bf16*)A0, (const bf16*)W0, out0, lda0, ldw0, N0, M0, N0, 0};
    TnProb p1 = {(const bf16*)A1, (const bf16*)W1, out1, lda1, ldw1, N1, M1, N1, 0};
    TnProb p2 = {(const bf16*)A2, (const bf16*)W2, out2, lda2, ldw2, N2, M2, N2, 0};
    constexpr int LDS = 3 * (128 + 192) * ROWB;
    auto kern = gemm_tn_group3_kernel<192>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    const int nmain = t0 + t1 + t2;
    ReduceJob job = {};
    int extra = 0;
    if (carry && carry->S > 1) {
        extra = 256 - nmain;                 // the CUs the tiles leave idle (one workgroup per CU: the ring takes 120 of 160 KB)
        if (extra < 8) {                     // nothing idle: the job runs as its own launch in front
            int rf = reduce_job_flush(*carry, s);
            if (rf != CPT_OK) return rf;
            extra = 0;
        } else job = *carry;
    }
    kern<<<dim3(nmain + extra), dim3(512), LDS, s>>>(p0, p1, p2, K, t0, t0 + t1, ex, job, nmain);
    return CPT_OK;
}

// Training forward of BertIntermediate (modeling_bert.py:144): u = A.W^T + bias (bf16, kept for the backward pass) and h = gelu(u)
// from one GEMM (the GELU used to be its own pass over the M x I tensor)
int gemm_gelu2(const void* A, int lda, const void* W, int ldw, const float* bias, void* u_out, void* h_out, int ldo, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8 || ldo % 8 || N % 8) return CPT_ERR_SHAPE;
    if (!A || !W || !u_out || !h_out) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)u_out | (uintptr_t)h_out | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    EpiX ex = {};
    ex.out_lp = u_out;
    launch_fast<bf16, CPT_EPI_GELU2, bf16>(g_gemm_variant >= 3 ? g_gemm_variant : 3, (const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0,
                                          (bf16*)h_out, ldo, M, N, K, s, &ex);
    return CPT_OK;
}

// NT GEMM with fp32 output for SMALL row counts (the few-shot step at 4 sequences per GPU: 480 rows = 32 tiles of 64 x 192 on 256 CUs,
// 48 K-tiles in sequence for the FFN-down): K split over workgroups, partial matrices added in split order, then + resid.
// Returns CPT_ERR_SHAPE when the problem does not call for it (the caller runs the plain GEMM).
int gemm_nt_split(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr, float* out, int ldo, int M, int N, int K,
                  void* partials, size_t partial_bytes, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8 || N % 4 || ldo != N || (resid && ldr != N) || !partials) return CPT_ERR_SHAPE;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)partials | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    const long tiles = (long)((M + 63) / 64) * ((N + 191) / 192);
    const int nt = K / 64;
    int S = (int)(256 / tiles);
    if (S > 16) S = 16;
    if (S > nt / 3) S = nt / 3;
    const size_t mat = (size_t)M * N * 4;
    while (S > 1 && (size_t)S * mat > partial_bytes) --S;
    if (S < 3) return CPT_ERR_SHAPE;
    int rc = launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, (float*)partials, ldo, M, N, K, s, S);
    if (rc != CPT_OK) return rc;
    const size_t n4 = mat / 16;
    reduce_partials_kernel<<<dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s>>>((const f32x4*)partials, (f32x4*)out, n4, S, (const f32x4*)resid);
    return CPT_OK;
}

// The same decisions as gemm_nt_split / the training forward's two-way split, but the partial matrices are LEFT for the consumer (the
// dropout + residual + LayerNorm pass adds them in split order): few rows -> 64 x 192 tiles, K over up to 16 workgroups per tile;
// 2048..6144 rows with a long K -> 128 x 192 tiles, K over two workgroups (M = 3840: 240 workgroups instead of 120).
int gemm_nt_partials(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, size_t partial_bytes, int M, int N, int K,
                     hipStream_t s, int* S_out) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8 || N % 4 || !partials || !S_out) return CPT_ERR_SHAPE;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)partials | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    const size_t mat = (size_t)M * N * 4;
    const int nt = K / 64;
    if (M >= 2048 && M <= 6144 && K >= 2048 && K % 128 == 0 && N % 192 == 0 && 2 * mat <= partial_bytes) {
        *S_out = 2;
        return launch_pipe<bf16, CPT_EPI_NONE, float, 128, 192, 4, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, partials, N, M, N, K, s, 2);
    }
    const long tiles = (long)((M + 63) / 64) * ((N + 191) / 192);
    int S = (int)(256 / tiles);
    if (S > 16) S = 16;
    if (S > nt / 3) S = nt / 3;
    while (S > 1 && (size_t)S * mat > partial_bytes) --S;
    if (S < 3) return CPT_ERR_SHAPE;
    *S_out = S;
    // (round 6: where even the split leaves half the chip idle -- a short contraction, K = 768: 32 tiles x 4 splits -- 64 x 96 tiles, launch_fast's rule)
    if (g_narrow_tiles && N % 96 == 0 && tiles * S <= NARROW_MAX)
        return launch_pipe<bf16, CPT_EPI_NONE, float, 64, 96, 2, 1, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, partials, N, M, N, K, s, S);
    return launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3>((const bf16*)A, lda, (const bf16*)W, ldw, bias, nullptr, 0, partials, N, M, N, K, s, S);
}

// ---- NN form: out[M][N] = A[M][K] . W[K][N] (+ resid), W = an nn.Linear weight [out_features = K][in_features = N] as stored:
// the data gradients dX = dY . W of the backward pass without a transposed weight copy --------------------------------------
CPT_SWITCH(int g_nn_tile256, 1);     // cpt_set_tuning(35, v): 1 (default) = the GELU-gradient data-gradient GEMM on 256 x 192 tiles where 128-row tiles would make 1..2 rounds
void set_nn_tile256(int v) { CPT_SWITCH_SET(g_nn_tile256 = v); (void)v; }
CPT_SWITCH(int g_nn_split2, 1);      // cpt_set_tuning(34, v): 1 (default) = the data-gradient GEMMs in front of a LayerNorm backward at 2048..6144 rows as two K-split bf16 partial matrices of 128 x 192 tiles
void set_nn_split2(int v) { CPT_SWITCH_SET(g_nn_split2 = v); (void)v; }
int gemm_nn_eligible(int M, int N, int K, int lda, int ldw) {
    return M > 0 && N > 0 && K > 0 && (N % 192 == 0 || N % 128 == 0) && K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && (size_t)K * ldw * 2 <= (size_t)0x7fffffff;
}

int gemm_nn(const void* A, int lda, const void* W, int ldw, const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
            hipStream_t s, int w_rows, void* partials, size_t partial_bytes, const void* gelu_u, int ldu, float* gelu_colsum, int colsum_rows, int* S_out, int split2_bf16) {
    if (S_out) *S_out = 1;
    if (!gemm_nn_eligible(M, N, K, lda, ldw)) return CPT_ERR_SHAPE;
    if (colsum_rows > 0 && (colsum_rows < (M + 31) / 32 || ldo % 4 || ldu % 8)) return CPT_ERR_SHAPE;      // partial-row bias sums: one row per 32-row wave block, vector epilogue
    if (!A || !W || !out) return CPT_ERR_NULL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)partials) & 15)) return CPT_ERR_ALIGN;
    if (w_rows < 0 || w_rows > K) return CPT_ERR_SHAPE;
    const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)W;
    // 64-row tiles when 128-row tiles would leave half the chip idle (M = 3840, N = 768: 120 vs 240 workgroups)
    const bool n192 = N % 192 == 0;                   // hidden 768 / 3072; otherwise 128-column tiles (Oscar-large: 1024 / 4096)
    const long wg128 = (long)((M + 127) / 128) * (N / (n192 ? 192 : 128));
    const bool small = wg128 < 200 && n192;
    EpiX ex = {};
    ex.w_rows = w_rows;
    ex.colsum = gelu_u ? gelu_colsum : nullptr;
    ex.colsum_rows = gelu_u && gelu_colsum ? colsum_rows : 0;
    // round 6 (split2_bf16, with S_out): 2048..6144 rows, a long contraction and a consumer that adds partial matrices itself (the LayerNorm backward behind
    // the data-gradient GEMMs): 128 x 192 tiles with K split over two workgroups instead of 64 x 192 tiles over the whole K.  The 64 x 192 tile is bound by
    // its LDS-DMA operand stream (1.5 MFMAs per KiB of operands against 2.4); the two partial matrices leave as bf16 (the same bytes as one fp32 matrix; the
    // fp32 residual is added by the consumer, not here) -- with fp32 partials the extra 35 MB per launch pair ate the gain (profiles/r06_ab_log.md)
    if (S_out && split2_bf16 && g_nn_split2 && out_dtype == CPT_F32 && !gelu_u && partials && ldo == N && n192 && (!resid || ldr == N) && M >= 2048 && M <= 6144 &&
        K >= 1536 && K % 128 == 0 && 2 * (size_t)M * N * 2 <= partial_bytes) {
        int rc = launch_pipe<bf16, CPT_EPI_NONE, bf16, 128, 192, 4, 2, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, nullptr, 0, (bf16*)partials, ldo, M, N, K, s, 2, &ex);
        if (rc != CPT_OK) return rc;
        *S_out = 2;
        return CPT_OK;
    }
    // few output tiles and a long contraction (the decoder's data gradient: 32 x 768 outputs over K = 30528 ran on 4 CUs for 220 us):
    // split K over up to 64 workgroups per tile, partial matrices added in split order
    if (out_dtype == CPT_F32 && (!resid || ldr == N) && partials && ldo == N && n192) {
        const long tiles = (long)((M + 63) / 64) * (N / 192);
        const int nt = K / 64;
        int S = (int)(256 / tiles);
        if (S > 64) S = 64;
        if (S > nt / 3) S = nt / 3;
        const size_t mat = (size_t)M * N * 4;
        while (S > 1 && (size_t)S * mat > partial_bytes) --S;
        if (S >= 4) {
            int rc = launch_pipe<bf16, CPT_EPI_NONE, float, 64, 192, 2, 2, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, nullptr, 0, (float*)partials, ldo, M, N, K, s, S, &ex);
            if (rc != CPT_OK) return rc;
            if (S_out) { *S_out = S; return CPT_OK; }      // the caller's consumer adds the S partial matrices (and the residual) itself
            const size_t n4 = mat / 16;
            reduce_partials_kernel<<<dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s>>>((const f32x4*)partials, (f32x4*)out, n4, S,
                                                                                                                  (const f32x4*)resid);
            return CPT_OK;
        }
    }
#define CPT_NN(EPI, OT, CFG_TBM, CFG_WM) launch_pipe<bf16, EPI, OT, CFG_TBM, 192, CFG_WM, 2, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, resid, ldr, (OT*)out, ldo, M, N, K, s, 1, &ex)
#define CPT_NN128(EPI, OT) launch_pipe<bf16, EPI, OT, 128, 128, 4, 2, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, resid, ldr, (OT*)out, ldo, M, N, K, s, 1, &ex)
    if (!n192) {
        if (out_dtype == CPT_BF16) {
            if (resid) return CPT_ERR_DTYPE;
            if (gelu_u) {
                if (ldu % 8 || ((uintptr_t)gelu_u & 15)) return CPT_ERR_ALIGN;
                resid = (const float*)gelu_u; ldr = ldu;
                return CPT_NN128(CPT_EPI_GELUGRAD, bf16);
            }
            return CPT_NN128(CPT_EPI_NONE, bf16);
        }
        if (out_dtype != CPT_F32 || gelu_u) return CPT_ERR_DTYPE;
        return resid ? CPT_NN128(CPT_EPI_RESID, float) : CPT_NN128(CPT_EPI_NONE, float);
    }
    if (out_dtype == CPT_BF16) {
        if (resid) return CPT_ERR_DTYPE;
        if (gelu_u) {       // out = (A.W) * gelu'(u): the GELU backward rides in the epilogue (u bf16 [M][ldu])
            if (ldu % 8 || ((uintptr_t)gelu_u & 15)) return CPT_ERR_ALIGN;
            resid = (const float*)gelu_u; ldr = ldu;
            // round 6: where 128-row tiles make between one and two rounds (M = 3840, N = 3072: 480 workgroups = 1.875 rounds on 256 CUs) and 256-row tiles make at
            // most one, take 256 x 192 (8 waves of 64 x 96, 2-stage ring of 112 KB): 240 workgroups, 1.71 instead of 2.4... MFMAs per KiB of operands UP (2.7)
            const long wg256 = (long)((M + 255) / 256) * (N / 192);
            if (g_nn_tile256 && M >= 2048 && wg128 > 256 && wg128 < 512 && wg256 <= 256)
                return launch_pipe<bf16, CPT_EPI_GELUGRAD, bf16, 256, 192, 4, 2, 2, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, resid, ldr, (bf16*)out, ldo, M, N, K, s, 1, &ex);
            // round 6, few rows (4 sequences per GPU: M = 480, N = 3072 is 128 tiles of 64 x 192 -- half the chip idle, and the launch lasts as long as one
            // workgroup's twelve K-tiles): 64 x 128 tiles (4 waves of 32 x 64) put 192 workgroups on the chip with 24 instead of 32 KB of operands per K-tile
            if (small && g_narrow_tiles && N % 128 == 0 && (long)((M + 63) / 64) * (N / 192) <= NARROW_MAX)
                return launch_pipe<bf16, CPT_EPI_GELUGRAD, bf16, 64, 128, 2, 2, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, resid, ldr, (bf16*)out, ldo, M, N, K, s, 1, &ex);
            return small ? CPT_NN(CPT_EPI_GELUGRAD, bf16, 64, 2) : CPT_NN(CPT_EPI_GELUGRAD, bf16, 128, 4);
        }
        // (round 6: the attention output's data gradient at few rows is 32 tiles of 64 x 192: 64 x 64 tiles -- 2 waves of 32 x 64 -- make it 96 workgroups)
        if (small && g_narrow_tiles && N % 64 == 0 && (long)((M + 63) / 64) * (N / 192) <= NARROW_MAX / 2)
            return launch_pipe<bf16, CPT_EPI_NONE, bf16, 64, 64, 2, 1, 3, 1, 4, 1, 2>(a, lda, w, ldw, nullptr, resid, ldr, (bf16*)out, ldo, M, N, K, s, 1, &ex);
        return small ? CPT_NN(CPT_EPI_NONE, bf16, 64, 2) : CPT_NN(CPT_EPI_NONE, bf16, 128, 4);
    }
    if (gelu_u) return CPT_ERR_DTYPE;
    if (out_dtype != CPT_F32) return CPT_ERR_DTYPE;
    if (resid) return small ? CPT_NN(CPT_EPI_RESID, float, 64, 2) : CPT_NN(CPT_EPI_RESID, float, 128, 4);
    return small ? CPT_NN(CPT_EPI_NONE, float, 64, 2) : CPT_NN(CPT_EPI_NONE, float, 128, 4);
#undef CPT_NN
#undef CPT_NN128
}

// ---- LayerNorm folded into the GEMMs around it (bf16 throughput path) -------------------------------
// Producer: out_f32 = A.W^T + bias + R, where R = resid or LayerNorm(resid; st_in, g_in, b_in) computed on
// the fly; also writes the bf16 copy and accumulates the row sums (sum, sum of squares) of out into st_out.
int ln_stat_parts(int n_cols) { return (n_cols + 95) / 96; }     // 96-column blocks of an n_cols-wide producer
int ln_stat_slots(int n_cols) { return (ln_stat_parts(n_cols) + 1) & ~1; }   // slots per row of its table [M][slots][2]

int gemm_ln_prod(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr,
                 const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                 float* out_f32, void* out_lp, float* st_out, int ldo, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8) return CPT_ERR_SHAPE;
    if (!A || !W || !resid || !out_f32 || !out_lp || !st_out) return CPT_ERR_NULL;
    // direct epilogue: 16-byte accesses on rows of the outputs, the residual and the column vectors
    if (N % 8 || ldo % 8 || ldr % 4 || (((uintptr_t)out_f32 | (uintptr_t)out_lp | (uintptr_t)resid | (uintptr_t)bias | (uintptr_t)g_in | (uintptr_t)b_in) & 15))
        return CPT_ERR_ALIGN;
    EpiX ex = {};
    ex.st_in = st_in; ex.st_in_parts = ln_stat_parts(hidden); ex.g_in = g_in; ex.b_in = b_in; ex.st_out = st_out; ex.st_out_slots = ln_stat_slots(N); ex.out_lp = out_lp;
    ex.eps = eps; ex.inv_h = 1.0f / (float)hidden;
    launch_fast<bf16, CPT_EPI_LNPROD, float>(g_gemm_variant >= 3 ? g_gemm_variant : 3, (const bf16*)A, lda, (const bf16*)W, ldw, bias,
                                            resid, ldr, out_f32, ldo, M, N, K, s, &ex);
    return CPT_OK;
}

// Same producer with the residual stream in the 3-byte form (r3_encode): residual = (resid_hi bf16, resid_lo int8) [M][ldr],
// output = (out_hi bf16 = the next GEMM's A operand, out_lo int8) [M][ldo]; no fp32 tensor is read or written.
int gemm_ln_prod3(const void* A, int lda, const void* W, int ldw, const float* bias, const void* resid_hi, const void* resid_lo, int ldr,
                  const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                  void* out_hi, void* out_lo, float* st_out, int ldo, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8) return CPT_ERR_SHAPE;
    if (!A || !W || !resid_hi || !resid_lo || !out_hi || !out_lo || !st_out) return CPT_ERR_NULL;
    if (N % 8 || ldo % 8 || ldr % 8 || (((uintptr_t)out_hi | (uintptr_t)out_lo | (uintptr_t)resid_hi | (uintptr_t)resid_lo | (uintptr_t)bias | (uintptr_t)g_in | (uintptr_t)b_in) & 15))
        return CPT_ERR_ALIGN;
    EpiX ex = {};
    ex.st_in = st_in; ex.st_in_parts = ln_stat_parts(hidden); ex.g_in = g_in; ex.b_in = b_in; ex.st_out = st_out; ex.st_out_slots = ln_stat_slots(N); ex.out_lp = out_hi;
    ex.resid_lo = (const signed char*)resid_lo; ex.out_lo = (signed char*)out_lo;
    ex.eps = eps; ex.inv_h = 1.0f / (float)hidden;
    launch_fast<bf16, CPT_EPI_LNPROD3, float>(g_gemm_variant >= 3 ? g_gemm_variant : 3, (const bf16*)A, lda, (const bf16*)W, ldw, bias,
                                             (const float*)resid_hi, ldr, (float*)nullptr, ldo, M, N, K, s, &ex);
    return CPT_OK;
}

// Consumer: A is the bf16 copy of a pre-LayerNorm tensor, Wf the gain-folded weight;
// out = [gelu]( rstd[m] * (A.Wf^T - mean[m] * colc[n]) + cold[n] )  ==  [gelu]( LayerNorm(A) . W^T + bias )
// bf16x3 parity mode, FFN-up (round 3): gelu(A'.W'^T + bias) over K' = 3K written straight as the [M][hi | hi | lo] split copy of h that the
// FFN-down's three-term GEMM reads -- no fp32 h, no stand-alone split3 pass over the M x I tensor
int gemm_gelu_x3(const void* A3, int lda, const void* W3, int ldw, const float* bias, void* out_split, int M, int N, int K3, hipStream_t s) {
    if (M <= 0 || N <= 0 || K3 <= 0 || K3 % 64 || lda % 8 || ldw % 8 || N % 8) return CPT_ERR_SHAPE;
    if (!A3 || !W3 || !out_split) return CPT_ERR_NULL;
    if ((((uintptr_t)A3 | (uintptr_t)W3 | (uintptr_t)out_split | (uintptr_t)bias) & 15)) return CPT_ERR_ALIGN;
    EpiX ex = {};
    ex.x3_k = N;
    launch_fast<bf16, CPT_EPI_GELU_X3, bf16>(g_gemm_variant >= 3 ? g_gemm_variant : 3, (const bf16*)A3, lda, (const bf16*)W3, ldw, bias, nullptr, 0,
                                            (bf16*)out_split, 3 * N, M, N, K3, s, &ex);
    return CPT_OK;
}

CPT_SWITCH(int g_qkv_2pass, 1);
void set_qkv_2pass(int v) { CPT_SWITCH_SET(g_qkv_2pass = v); (void)v; }

int gemm_ln_cons(const void* A, int lda, const void* Wf, int ldw, const float* st_in, const float* colc, const float* cold,
                 float eps, int hidden, int gelu, void* out_lp, int ldo, int M, int N, int K, hipStream_t s, int out_panel, const void* pf, size_t pf_bytes, int a_panel) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8) return CPT_ERR_SHAPE;
    if (a_panel && (!out_panel || lncons4_enabled() >= 2)) return CPT_ERR_SHAPE;      // A from the panel residual stream: the two-pass kernel with panel output only (ln_cons_takes_panel_a)
    if (!A || !Wf || !st_in || !colc || !cold || !out_lp) return CPT_ERR_NULL;
    if (N % 8 || ldo % 8 || (((uintptr_t)out_lp | (uintptr_t)colc | (uintptr_t)cold) & 15)) return CPT_ERR_ALIGN;
    EpiX ex = {};
    ex.st_in = st_in; ex.st_in_parts = ln_stat_parts(hidden); ex.colc = colc; ex.cold = cold; ex.eps = eps; ex.inv_h = 1.0f / (float)hidden;
    const int v = g_gemm_variant >= 3 ? g_gemm_variant : 3;
    if (out_panel) {      // output in the panel layout of the FFN-down producer (gemm_prod.hip): the two-pass kernel only
        if (!gelu || !ffn_up_2pass_legal(M, N, K)) return CPT_ERR_SHAPE;
        if (lncons4_enabled() >= 2) return gemm_lncons4(A, lda, Wf, ldw, st_in, ln_stat_parts(hidden), colc, cold, eps, hidden, out_lp, ldo, M, N, K, s, 1, pf, pf_bytes, 1);
        void* tr = ((g_trace_epi < 0 || g_trace_epi == CPT_EPI_LNCONS_GELU) && (g_trace_k == 0 || g_trace_k == K)) ? (void*)g_gemm_trace : nullptr;
        return gemm_ffn_up_2pass(A, lda, Wf, ldw, st_in, ln_stat_parts(hidden), colc, cold, eps, hidden, out_lp, ldo, M, N, K, tr, g_gemm_abl, s, 1, pf, pf_bytes, 1, a_panel);
    }
    // round 4: the 4-wave consumer kernel (gemm_ffn4.hip, 192 x 256 tiles) where ITS tiles fill their rounds clearly better than the two-pass
    // kernel's 384 x 256 ones: Oscar-large FFN-up, M = 8480, N = 4096: 720 tiles = 2.8 rounds (94 % of three) against 368 = 1.44 (72 % of two):
    // 2.41 -> 2.13 ms per step (VCR shape).  At equal round efficiency the two-pass kernel stays: the 4-wave K loop is denser (1980 against
    // 2700-2900 ticks per K-tile) but one wave per SIMD takes 13 k ticks for a tile's GELU epilogue, 1.6x the two-pass kernel's per element
    // (bench shape 44 -> 49 us per launch; GQA-shape QKV projection equal).  cpt_set_tuning key 29: 0 never, 2 always (where legal);
    // variant 21 forces it (tests).
    const long t4 = (long)((M + 191) / 192) * (N / 256), r4 = (t4 + 255) / 256;
    const long t2 = (long)((M + 383) / 384) * (N / 256), r2 = (t2 + 255) / 256;
    const bool fills_better = t4 * r2 * 10 >= t2 * r4 * 11;            // round efficiency t / (256 r): at least 10 % better
    if ((v == 21 && ffn_up_2pass_legal(M, N, K) && N % 256 == 0) ||
        (v == 3 && ffn_up_2pass_preferred(M, N, K) && (gelu || (g_qkv_2pass && t4 * 5 >= r4 * 256 * 4)) && (lncons4_enabled() >= 2 || (lncons4_enabled() == 1 && fills_better))))
        return gemm_lncons4(A, lda, Wf, ldw, st_in, ln_stat_parts(hidden), colc, cold, eps, hidden, out_lp, ldo, M, N, K, s, 0, nullptr, 0, gelu);
    if (gelu && ((v == 3 && ffn_up_2pass_preferred(M, N, K)) || (v == 20 && ffn_up_2pass_legal(M, N, K)))) {     // variant 20: forced (tests)
        void* tr = ((g_trace_epi < 0 || g_trace_epi == CPT_EPI_LNCONS_GELU) && (g_trace_k == 0 || g_trace_k == K)) ? (void*)g_gemm_trace : nullptr;
        return gemm_ffn_up_2pass(A, lda, Wf, ldw, st_in, ln_stat_parts(hidden), colc, cold, eps, hidden, out_lp, ldo, M, N, K, tr, g_gemm_abl, s);
    }
    // round 3: the stand-alone QKV projection (attention not fused: L > 128) through the GELU-less two-pass kernel when its 384 x 256 tiles
    // fill the chip (key 20 = 0: the 384 x 192 pipe kernel as before); same accumulation order, same bits
    // (only when its tiles fill their rounds: M = 8480, N = 3072 makes 23 x 12 = 276 tiles = 1.08 rounds, where 192 x 192 tiles take 1.70 vs 1.97 ms per step)
    // (round 4: not where 128 x 192 tiles run many rounds -- there the two-per-CU form of the pipe kernel is ahead of both LayerNorm-consumer
    // kernels: GQA shape, 5040 tiles: 2.74 against 2.87 (two-pass) and 2.81 ms per step (4-wave); the tile model would take 384 x 192: 3.98)
    const long t2p = (long)((M + 383) / 384) * (N / 256), r2p = (t2p + 255) / 256;
    const long t128 = (long)((M + 127) / 128) * ((N + 191) / 192);
#ifndef CPT_QKV_OCC2_MIN
#define CPT_QKV_OCC2_MIN 1024
#endif
    if (!gelu && v == 3 && g_qkv_2pass && t128 >= CPT_QKV_OCC2_MIN) {
        launch_pipe<bf16, CPT_EPI_LNCONS, bf16, CPT_CFG_128x192_OCC2>((const bf16*)A, lda, (const bf16*)Wf, ldw, nullptr, nullptr, 0, (bf16*)out_lp, ldo, M, N, K, s, 1, &ex);
        return CPT_OK;
    }
    if (!gelu && g_qkv_2pass && ((v == 3 && ffn_up_2pass_preferred(M, N, K) && t2p * 5 >= r2p * 256 * 4) || (v == 20 && ffn_up_2pass_legal(M, N, K))))
        return gemm_ffn_up_2pass(A, lda, Wf, ldw, st_in, ln_stat_parts(hidden), colc, cold, eps, hidden, out_lp, ldo, M, N, K, nullptr, g_gemm_abl, s, 0, nullptr, 0, 0);
    if (gelu) launch_fast<bf16, CPT_EPI_LNCONS_GELU, bf16>(v, (const bf16*)A, lda, (const bf16*)Wf, ldw, nullptr, nullptr, 0, (bf16*)out_lp, ldo, M, N, K, s, &ex);
    else launch_fast<bf16, CPT_EPI_LNCONS, bf16>(v, (const bf16*)A, lda, (const bf16*)Wf, ldw, nullptr, nullptr, 0, (bf16*)out_lp, ldo, M, N, K, s, &ex);
    return CPT_OK;
}

// Fused QKV projection + self-attention (bf16, seq_len <= 128, head_dim 64): one workgroup per (sequence, head)
// computes that head's Q | K | V for the sequence's tokens (A rows b*L .., the three 64-row slices of the fused
// [3H][K] weight) and runs softmax(QK^T/8 + mask)V on them from LDS; only the context rows reach HBM.
//   st_in == NULL: plain projection (x W^T + bias);  else LayerNorm folded as in gemm_ln_cons (W = gain-folded).
template <int EPI, int STAGES, int FD, int OCC>
static int launch_qkv_attn(const bf16* A, int lda, const bf16* W, int ldw, const float* bias, bf16* ctx, int ldo,
                           int M, int N, int K, const EpiX& ex, int B, hipStream_t s) {
    constexpr int LDS = STAGES * (128 + 192) * ROWB;
    auto kern = gemm_pipe_kernel<bf16, EPI, bf16, 128, 192, 4, 2, STAGES, 1, FD, OCC>;
    static bool attr_done_dev[CPT_MAX_DEV] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (LDS > 64 * 1024 && !attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        attr_done = true;
    }
    EpiX e = ex;
    e.skew = g_gemm_skew;
    long long* tr = (g_trace_epi < 0 || g_trace_epi == EPI) ? g_gemm_trace : nullptr;
    kern<<<dim3(B * ex.heads), dim3(512), LDS, s>>>(A, lda, W, ldw, bias, nullptr, 0, ctx, ldo, M, N, K, 1, tr, g_gemm_abl, e);
    return CPT_OK;
}

int gemm_qkv_attn(const void* A, int lda, const void* W, int ldw, const float* bias, const float* st_in, const float* colc,
                  const float* cold, float eps, int hidden, const int64_t* mask, void* ctx, int ldo, int B, int L, int heads,
                  int K, int config, hipStream_t s) {
    if (B <= 0 || L <= 0 || L > 128 || heads <= 0 || K <= 0 || K % 64 || lda % 8 || ldw % 8 || ldo % 4) return CPT_ERR_SHAPE;
    if (!A || !W || !ctx || (st_in && (!colc || !cold))) return CPT_ERR_NULL;
    EpiX ex = {};
    ex.st_in = st_in; ex.st_in_parts = ln_stat_parts(hidden); ex.colc = colc; ex.cold = cold; ex.eps = eps; ex.inv_h = 1.0f / (float)hidden;
    ex.mask = mask; ex.seq_len = L; ex.heads = heads;
    const int M = B * L, N = 3 * heads * 64;
    const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)W; bf16* c = (bf16*)ctx;
    if (config == 2) {      // one workgroup per CU, 3-stage ring
        return st_in ? launch_qkv_attn<CPT_EPI_ATTN_LN, 3, 4, 1>(a, lda, w, ldw, bias, c, ldo, M, N, K, ex, B, s)
                     : launch_qkv_attn<CPT_EPI_ATTN, 3, 4, 1>(a, lda, w, ldw, bias, c, ldo, M, N, K, ex, B, s);
    }
    // two co-resident workgroups per CU: one's softmax (VALU) runs under the other's K loop (MFMA)
    return st_in ? launch_qkv_attn<CPT_EPI_ATTN_LN, 2, 2, 2>(a, lda, w, ldw, bias, c, ldo, M, N, K, ex, B, s)
                 : launch_qkv_attn<CPT_EPI_ATTN, 2, 2, 2>(a, lda, w, ldw, bias, c, ldo, M, N, K, ex, B, s);
}

void set_gemm_variant(int v) { CPT_SWITCH_SET(g_gemm_variant_sw = v); (void)v; }
void set_gemm_abl(int v) { CPT_SWITCH_SET(g_gemm_abl = v); (void)v; }
void set_gemm_trace(void* p) { g_gemm_trace = (long long*)p; }

}  // namespace cpt
