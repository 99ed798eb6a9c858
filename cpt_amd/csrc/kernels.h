// Internal launcher declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cpt_hip.h"
#include "../../include/cpt_hip_debug.h"
#include "dropout.h"
#include "switches.h"

namespace cpt {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: the launchers remember it per (kernel, device), not per process
constexpr int CPT_MAX_DEV = 64;
inline int current_device_slot() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CPT_MAX_DEV) d = 0; return d; }

int gemm(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias,
         const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
         hipStream_t s);

int embed_ln(const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word,
             const float* posw, const float* typew, const float* g, const float* bta, float eps,
             float* out_f32, void* out_lp, int lp_dtype, int B, int Lt, int L, int H, int vocab,
             int max_pos, int type_vocab, hipStream_t s, void* out_lo = nullptr, int out_panel = 0, float* zero_f2 = nullptr, unsigned* zero_u1 = nullptr, const struct DropSpec* out_drop = nullptr);   // out_drop (round 6): hidden dropout on the output rows (site mask at element index row * H + col, as dropout_rows)   // zero_f2 / zero_u1 (round 6, training forward): two floats and one word cleared by the launch's first workgroup -- the loss accumulator and the finish ticket of the cross-entropy launch at the END of the same forward (no memset launch);   out_lo: rows leave in the 3-byte residual form (hi -> out_lp); out_panel: at their panel positions

// bf16 inference: embed_ln (3-byte or bf16 output rows) and pad_cast(bf16) of the region features in ONE launch (they touch disjoint data)
int embed_ln_pad_cast(const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word, const float* posw, const float* typew,
                      const float* g, const float* bta, float eps, void* out_lp, void* out_lo, int B, int Lt, int L, int H, int vocab,
                      int max_pos, int type_vocab, const float* x, void* xo, int R, int K, int Kp, hipStream_t s, int out_panel = 0,
                      float* out_f32 = nullptr, float* zero_f2 = nullptr, unsigned* zero_u1 = nullptr, const struct DropSpec* out_drop = nullptr);     // out_f32 / zero_*: the training forward's fp32 residual rows and cleared words (embed_ln above);   out_panel (round 5): out_lp / out_lo = the panel-layout residual stream

int layernorm_rows(const float* x, const float* g, const float* bta, float eps, float* out_f32,
                   void* out_lp, int lp_dtype, int R, int H, int grp, int grp_stride, int grp_off,
                   hipStream_t s);

int attention(int dtype, const void* qkv, const int64_t* attn_mask, void* ctx, void* probs, int B,
              int L, int heads, hipStream_t s, const DropSpec* drop = nullptr, int mask_3d = 0,   // mask_3d: attn_mask is [B][L][L]
              int ctx_panel = 0,    // ctx_panel (bf16 inference): ctx leaves in the panel layout of the attn-out producer (gemm_prod.hip)
              float* stats = nullptr);   // stats (bf16 training forward, round 6): [B * heads][L][2] (row max in base 2, 1 / row sum) of every query, read by attention_bwd

// bf16x3 parity mode: attention on bf16 MFMA with split operands (three-term products); ctx fp32 [M][H] or, with ctx_split, the split copy
// [M][hi | hi | lo] bf16 (ld 3H) for the attention-output GEMM; 2-D masks, L <= 288 (attention_x3_supported)
int attention_x3_supported(int L);
int attention_x3(const float* qkv, const int64_t* attn_mask, float* ctx, void* ctx_split, int B, int L, int heads, hipStream_t s,
                 const DropSpec* drop = nullptr);      // drop: dropout on the probabilities (the training forward of the mode)

int attention_max_len(int inference);      // 1024 for the inference kernels (beyond 288: the one-wave-per-query coverage kernel), 288 for the training step
int pad_cast(const float* x, void* out, int dtype, int R, int K, int Kp, hipStream_t s);
int split3(const float* x, int ld, void* out_bf16, int R, int K, int weight_order, hipStream_t s);
// [3][Rp][C] row-stacked split (blocks hi, hi, lo or hi, lo, hi; rows beyond R zero): operands of a product contracting over rows
int split3_rows(const float* x, int ld, void* out_bf16, int R, int Rp, int C, int weight_order, hipStream_t s);

int gather_rows(const void* src, int dtype, const int64_t* pos, void* out, int B, int L, int H,
                hipStream_t s, const int64_t* seq = nullptr, int n_seq = 0);      // seq: row b = position pos[b] of sequence seq[b] (of n_seq)

int ce_rows(const float* logits, const int64_t* labels, float* loss, float* dlogits, int R, int V,
            hipStream_t s, unsigned* ticket = nullptr, float* mean_out = nullptr, float* loss_copy = nullptr);
// ticket (round 6, a word holding 0, left at 0): the LAST workgroup to finish reads the totals back and writes mean_out[0] = sum / count and
// loss_copy[0..1] = {sum, count} -- no divide kernel on the caller's side, no device-to-device copy of the totals

int transpose_cast(const void* in, int in_dtype, int ldi, void* out, int out_dtype, int ldo, int R, int C, hipStream_t s);
int colsum(const void* x, int dtype, int ld, float* out, int R, int C, hipStream_t s);
int gelu_fwd(const void* u, void* h, int dtype, size_t n, hipStream_t s);
int gelu_bwd(const void* dh, const void* u, void* du, int dtype, size_t n, hipStream_t s);
// column-sum jobs (round 6): dst[c / seg][c % seg] += sum over `rows` of src[r * row_stride + c]; up to four ride in the spare workgroups of a
// LayerNorm backward launch (LnBwdExtra::jobs) or go out on their own (col_jobs_flush) -- the deferred second stage of earlier launches' partial sums
struct ColJob { const float* src; float *dst0, *dst1, *dst2; int rows, cols, row_stride, seg; };
struct ColJobs { ColJob j[4]; int n; };
int col_jobs_flush(const ColJobs& jobs, hipStream_t s);
inline bool col_jobs_add(ColJobs& q, const float* src, int rows, int cols, int row_stride, int seg, float* d0, float* d1 = nullptr, float* d2 = nullptr) {
    if (q.n >= 4) return false;
    q.j[q.n++] = ColJob{src, d0, d1, d2, rows, cols, row_stride, seg};
    return true;
}
// Rows of a compact [B][H] matrix that ARE rows (b, pos[b]) of the [B][L][H] tensor (the head rows of the pruned last layer, round 6): the hidden
// dropout masks are functions of the element index in the full tensor, so the compact passes regenerate them at row b * L + pos[b].
struct RowMap { const int64_t* pos; int L; int on; };      // pos NULL: position 0 ([CLS]); clamped like gather_rows
__host__ __device__ inline size_t rowmap_row(const RowMap& m, int r) {
    if (!m.on) return (size_t)r;
    long p = m.pos ? m.pos[r] : 0;
    p = p < 0 ? 0 : (p >= m.L ? m.L - 1 : p);
    return (size_t)r * m.L + (size_t)p;
}
struct LnBwdExtra {
    const float* stats;       // [R][2] (mean, rstd) of the forward (layernorm_rows_ex stat_out): the row statistics are not recomputed
    int dy_parts;             // > 1: dy holds that many split-K partial matrices dy_stride elements apart (compact rows only), added in split order
    int dy_parts_bf16;        // the partial matrices are bf16 (gemm_nn split2_bf16), not fp32
    size_t dy_stride;
    const float* dy_resid;    // + this (the residual of the data-gradient GEMM whose partials dy holds)
    const ColJobs* jobs;      // column-sum jobs of earlier launches, run by extra workgroups of this one
    int defer_reduce;         // 1: leave this launch's partial rows [ln_bwd_part_rows(R)][2 or 3][H] in `part` for a later job instead of launching the reduction
    RowMap drop_rows;         // dropout masks taken at these rows of the full tensor (compact head rows)
    DropSpec in_drop;         // round 6: hidden dropout applied to the INCOMING gradient rows as they are read (element index (placed row) * H + col of that site's mask:
                              // the embedding dropout's backward on the region rows, instead of a dropout pass over all rows in front); thresh 0: none
    const unsigned short* keep_bits;   // [R][64] the forward pass's keep bits (layernorm_rows_ex keep_out): bit 4 i + j of word [row][lane] = element (lane + 64 i) * 4 + j;
                              // the mask is then read, not regenerated (the Philox rounds were ~1/3 of the launch's VALU time); H <= 1024
};
int ln_bwd_part_rows(int R);
int ln_bwd(const float* dy, const float* x, const float* g, float eps, float* dx, void* dx_lp, int lp_dtype,
           float* dg, float* db, int R, int H, int grp, int grp_stride, int grp_off, int gelu_in, hipStream_t s,
           float* part = nullptr, size_t part_bytes = 0,       // part: scratch for two-stage column sums ((R / 4) * 3 * H floats)
           const DropSpec* drop = nullptr, float* dbias = nullptr,    // drop: dx_lp = dx through that hidden-site mask; dbias += column sums of it
           const LnBwdExtra* ext = nullptr);
int embed_bwd(const float* dy, const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word,
              const float* posw, const float* typew, const float* g, float eps, float* dword, float* dposw,
              float* dtypew, float* dg, float* db, int B, int Lt, int L, int H, int vocab, int max_pos,
              int type_vocab, hipStream_t s, const DropSpec* in_drop = nullptr);     // in_drop (round 6): BertEmbeddings' dropout applied to dy as it is read (element index (b L + t) H + col)
// pruned last layer (round 6): out_a[b] = a[b][pos[b]] (bf16 rows), out_b[b] = bb[b][pos[b]] (fp32 rows) in one launch; and its inverse for the
// backward: za [B][L][H] bf16 and zb [B][L][H] fp32 zero except row (b, pos[b]) = a_r[b] / b_r[b] (one launch: fill + rows)
int tail_gather2(const void* a, const float* bb, const int64_t* pos, void* out_a, float* out_b, int B, int L, int H, hipStream_t s);
int tail_scatter2(const void* a_r, const float* b_r, const int64_t* pos, void* za, float* zb, int B, int L, int H, hipStream_t s);
int scatter_rows_add(const float* src, const int64_t* pos, float* dst, int B, int L, int H, hipStream_t s, const int64_t* seq = nullptr, int n_seq = 0);
int tanh_bwd(const float* dy, const float* y, float* dx, void* dx_lp, int lp_dtype, size_t n, hipStream_t s);
int unpad_add(const float* src, float* dst, int R, int K, int Kp, hipStream_t s);     // dst = src without the padding columns (overwrites)
constexpr int ZS_MAX = 224;
struct ZeroSegs { float* p[ZS_MAX]; unsigned n[ZS_MAX]; int count; };     // 2.7 KB of kernel arguments
int zero_segments(const ZeroSegs& z, hipStream_t s);
int scale_cast(const float* x, const float* loss_acc, float scale, const float* dscale, void* out, int out_dtype, int R, int C, int ldo,
               hipStream_t s, float* colsum_out = nullptr);     // colsum_out (round 6): [C] += column sums of the rounded output (the bias gradient behind it), same launch at R <= 64
int attention_bwd(int dtype, const void* qkv, const int64_t* attn_mask, const void* dctx, void* dqkv, int B, int L, int heads, hipStream_t s,
                  const DropSpec* drop = nullptr, float* dbias = nullptr, int mask_3d = 0,
                  const void* ctx = nullptr, const float* stats = nullptr);     // ctx + stats (round 6): the forward's context rows and softmax statistics -- the L <= 128 bf16 kernel then skips its statistics pass;     dbias [3H]: += column sums of dqkv (the stacked Q|K|V bias gradient); mask_3d: attn_mask is [B][L][L]
// y = dropout(x) (+ resid): x, y fp32 [R][H] (in place allowed), y_lp optional copy in lp_dtype; element index of the mask =
// row * H + col.  Forward of the hidden dropouts and, with resid = NULL, their backward (the mask applied to a gradient).
// bf16x3 training: the MFMA backward on split fp32 operands (bwd.hip attn_bwd_x3_kernel / attn_bwd_x3_long_kernel), L <= 288, per-key masks
int attention_bwd_x3_supported(int L, int mask_3d);
int attention_bwd_x3(const float* qkv, const int64_t* attn_mask, const float* dctx, float* dqkv, int B, int L, int heads, hipStream_t s,
                     const DropSpec* drop, float* dbias);
int attention_bwd_supported(int dtype, int L, int has_drop, int mask_3d = 0);
int dropout_rows(const float* x, const float* resid, float* y, void* y_lp, int lp_dtype, int R, int H, const DropSpec& d, hipStream_t s);
// keep-mask export (tests): kind 0 hidden [R][H]; kind 1 attention [BH][L][L]; out = 1 keep / 0 drop
int dropout_mask(int kind, unsigned char* out, int n0, int n1, int n2, const DropSpec& d, hipStream_t s);
int adamw_flat(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16, size_t n,
               double lr, double beta1, double beta2, double eps, double wd, int step, float grad_scale, hipStream_t s, int flags = 0);     // flags: CPT_ADAMW_* (cpt_hip.h)
int layernorm_rows_ex(const float* x, const float* g, const float* bta, float eps, float* out_f32, void* out_lp,
                      int lp_dtype, int R, int H, int grp, int grp_stride, int grp_off, int gelu_in, hipStream_t s,
                      const float* resid = nullptr, const DropSpec* drop = nullptr, float* pre_out = nullptr, void* out_lo = nullptr,
                      int x_parts = 1, size_t x_stride = 0, int out_panel = 0, float* stat_out = nullptr, const RowMap* drop_rows = nullptr,
                      unsigned short* keep_out = nullptr, const float* resid_stat = nullptr, const float* resid_g = nullptr, const float* resid_b = nullptr, const DropSpec* out_drop = nullptr);     // out_drop (round 6): hidden dropout on the OUTPUT rows, element index (output row) * H + col;     // resid_stat / resid_g / resid_b (round 6): resid = the pre-LayerNorm rows of the LayerNorm in front, whose output is re-formed here from its [R][2] (mean, rstd), gain and shift (that launch then skips its fp32 output);   keep_out (round 6, with drop): [R][64] keep bits of the row's dropout mask for ln_bwd (LnBwdExtra::keep_bits)     // stat_out (round 6): [R][2] (mean, rstd) of every row, for ln_bwd     // out_panel (round 5, with out_lo): out_lp / out_lo are the bases of the panel-layout residual stream; x_parts > 1: x holds that many split-K partial matrices, x_stride elements apart; the row processed is their sum in split order
// resid / drop / pre_out: normalise dropout(x) + resid (element index row * H + col of the hidden-site mask) and store that sum

// out[M][N] fp32 = sum_k A[k][m] W[k][n]: bf16 operands with the contraction index as the slow dimension (weight gradients
// without operand transposes); needs gemm_tn_eligible; `partials` (optional) holds the split-K partial matrices
int gemm_tn_eligible(int M, int N, int K, int lda, int ldw, int ldo);
// ReduceJob (round 6): the addition of a K-split launch's S partial matrices (n4 float4 each, S > 1), in split order, LEFT for spare workgroups of a later
// launch (gemm_tn_triple: 216 of 256 CUs busy at hidden 768) instead of a reduction launch of its own; S <= 1: no job
struct ReduceJob { const float* part; float* out; size_t n4; int S; };
int reduce_job_flush(const ReduceJob& job, hipStream_t s);     // a launch of its own (nothing left to carry it)
int gemm_tn(const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K, void* partials, size_t partial_bytes,
            hipStream_t s, int k_rows = 0, ReduceJob* defer = nullptr);     // defer: a split launch fills *defer instead of launching the reduction;   k_rows: rows of A / W that exist when K was rounded up to a multiple of 64
// two such products over the same K rows in one launch (the weight gradients of a layer's two FFN matrices, or of its attention
// output and Q|K|V matrices); outputs dense (ldo = N).  CPT_ERR_SHAPE: not applicable, run two gemm_tn
int gemm_tn_pair(const void* A0, int lda0, const void* W0, int ldw0, float* out0, int M0, int N0,
                 const void* A1, int lda1, const void* W1, int ldw1, float* out1, int M1, int N1,
                 int K, int k_rows, void* partials, size_t partial_bytes, hipStream_t s);
// ... and three (FFN down | FFN up | attention output), every workgroup over the whole K; gemm_tn_triple_eligible says whether the shapes fit one round
int gemm_tn_triple_eligible(int M0, int N0, int M1, int N1, int M2, int N2, int K);
int gemm_tn_triple(const void* A0, int lda0, const void* W0, int ldw0, float* out0, int M0, int N0,
                   const void* A1, int lda1, const void* W1, int ldw1, float* out1, int M1, int N1,
                   const void* A2, int lda2, const void* W2, int ldw2, float* out2, int M2, int N2, int K, int k_rows, hipStream_t s, const ReduceJob* carry = nullptr);     // carry: run by the workgroups behind the tiles (CPT_ERR_SHAPE untouched: the caller flushes it)
// out[M][N] = A[M][K] . W[K][N] (+ resid): W stored with the contraction index as its slow dimension (data gradients against an
// nn.Linear weight as stored); bf16 operands, out fp32 (optionally + fp32 resid) or bf16
int gemm_nn_eligible(int M, int N, int K, int lda, int ldw);
int gemm_nn(const void* A, int lda, const void* W, int ldw, const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
            hipStream_t s, int w_rows = 0, void* partials = nullptr, size_t partial_bytes = 0, const void* gelu_u = nullptr, int ldu = 0,
            float* gelu_colsum = nullptr,      // gelu_colsum [N] (with gelu_u): += column sums of out (fp32, before the bf16 rounding) = the gradient of the bias in front of the GELU
            int colsum_rows = 0,               // > 0 (round 6): gelu_colsum is [colsum_rows >= M / 32][N] partial rows (plain stores, one per 32-row wave block) for a later column-sum job
            int* S_out = nullptr,              // non-NULL (round 6): where the launch splits K, leave the *S_out partial matrices in `partials` (no reduction launch, resid NOT added); *S_out = 1: out is complete
            int split2_bf16 = 0);              // with S_out: the caller's consumer also takes TWO BF16 partial matrices (2048..6144 rows, long K: 128 x 192 tiles, K split in two); then *S_out = 2 means bf16 partials
// gelu_u (bf16 out only): out = (A.W) * gelu'(u) with u bf16 [M][ldu] -- the GELU backward fused into the data-gradient GEMM
// w_rows: rows of W that exist when K was rounded up to a multiple of 64 (A's extra columns must be zero); partials: split-K scratch
// u = A.W^T + bias (bf16) and h = gelu(u) (bf16) from one bf16 GEMM (training forward of BertIntermediate)
int gemm_gelu2(const void* A, int lda, const void* W, int ldw, const float* bias, void* u_out, void* h_out, int ldo, int M, int N, int K, hipStream_t s);
// bf16 NT GEMM, fp32 out (+ bias, + resid), K split for small row counts; CPT_ERR_SHAPE = not worth it / not applicable (caller: plain gemm)
int gemm_nt_split(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr, float* out, int ldo, int M, int N, int K,
                  void* partials, size_t partial_bytes, hipStream_t s);
// bf16 NT GEMM into split-K partial matrices partials[S][M][N] (bias in the first) for a consumer that adds them itself (layernorm_rows_ex x_parts):
// returns S >= 2 in *S_out, or CPT_ERR_SHAPE when the problem does not call for a split (the caller runs the plain GEMM)
int gemm_nt_partials(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, size_t partial_bytes, int M, int N, int K,
                     hipStream_t s, int* S_out);
// bf16 region projection: two fp32 partial matrices out2[2][M][ldo] (K split in two; bias in the first), summed by the LayerNorm pass behind it
int gemm_img_proj(const void* A, int lda, const void* W, int ldw, const float* bias, float* out2, int ldo, int M, int N, int K, hipStream_t s);
// MLM head on the [MASK] rows, bf16 path (round 3): gather + 3-byte merge + LayerNorm in one launch; transform GEMM with K split over
// workgroups (partials[S][M][N], S = head_transform_splits(K)); reduction + GELU + LayerNorm in one launch
int head_rows_ln3(const void* hi, const void* lo, const int64_t* pos, const float* g, const float* bta, float eps, void* out_bf16, int R, int L, int H, hipStream_t s,
                  const void* pf = nullptr, size_t pf_bytes = 0, int src_panel = 0);     // src_panel (round 5): hi / lo in the panel layout; pf: region (the decoder's weight table) that leading blocks of the launch read into the Infinity Cache
// decoder scores of a list of vocabulary columns only: out[R][n] = t[R][H] . W[cols[j]][:] + bias[cols[j]]   (mode 0 bf16 x bf16, 1 fp32 x fp32, 2 fp32 x the bf16x3 split table)
int decoder_cols(const void* t, int mode, const void* W, const float* bias, const int64_t* cols, int n, float* out, int R, int H, int V, hipStream_t s);
int head_transform_splits(int K);
int gemm_head_transform(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, int M, int N, int K, hipStream_t s);
int head_finish(const float* partials, int S, const float* g, const float* bta, float eps, void* out_bf16, int R, int H, hipStream_t s,
                const void* pf = nullptr, size_t pf_bytes = 0);     // pf: as head_rows_ln3 (the rest of the decoder table)
// last encoder layer on the head's rows only (round 5; rowops.hip / gemm.hip): gather + previous LayerNorm of the head rows, K-split dense layers on them
int tail_rows(const void* hi, const void* lo, const int64_t* pos, const float* g, const float* bta, float eps, const void* ctx, void* ctx_out, float* resid_out,
              int R, int L, int H, hipStream_t s, int src_panel, int ctx_panel, const void* pf = nullptr, size_t pf_bytes = 0);
int gelu_parts(const float* partials, int S, void* out_bf16, size_t n, hipStream_t s, const void* pf = nullptr, size_t pf_bytes = 0,
               int split_cols = 0);     // split_cols = N > 0: n = R * N values written as the bf16x3 mode's [R][hi | hi | lo] split copy (row pitch 3 N)
int tail_finish(const float* partials, int S, const float* resid, const float* g, const float* bta, float eps, float* out_f32, void* out_bf16, int R, int H, hipStream_t s,
                const void* pf = nullptr, size_t pf_bytes = 0);
int rows_gemm_splits(int K);
int gemm_rows_split(const void* A, int lda, const void* W, int ldw, const float* bias, float* partials, int M, int N, int K, hipStream_t s);
int gemm_splitk_accum(int dtype, const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K, hipStream_t s);
int gemm_ln_prod(const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr,
                 const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                 float* out_f32, void* out_lp, float* st_out, int ldo, int M, int N, int K, hipStream_t s);
// the same producer with the residual stream in the 3-byte form (common.h: r3_encode), in and out
int gemm_ln_prod3(const void* A, int lda, const void* W, int ldw, const float* bias, const void* resid_hi, const void* resid_lo, int ldr,
                  const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                  void* out_hi, void* out_lo, float* st_out, int ldo, int M, int N, int K, hipStream_t s);
// round 3: the same producer with A read from its fragment-major ("panel") copy straight into registers (gemm_prod.hip);
// panel[M / 32][K / 16][64][8] bf16, see panel_pack; needs panel_eligible(M, N, K)
int panel_eligible(int M, int N, int K);
int panel_pack(const void* src, int ld, void* dst, int M, int K, int to_panel, hipStream_t s, int elem_bytes = 2);     // to_panel 0: the inverse; elem_bytes 1: the residual stream's lo bytes (units of 8 bytes)
int gemm_ln_prod3_panel(const void* A_panel, const void* W, int ldw, const float* bias, const void* resid_hi, const void* resid_lo, int ldr,
                        const float* st_in, const float* g_in, const float* b_in, float eps, int hidden,
                        void* out_hi, void* out_lo, float* st_out, int ldo, int M, int N, int K, hipStream_t s,
                        const void* pf0 = nullptr, size_t pf0_bytes = 0, const void* pf1 = nullptr, size_t pf1_bytes = 0, int resid_panel = 0);
// resid_panel (round 5): resid_hi / resid_lo / out_hi / out_lo are in the panel layout [M / 32][N / 16][64][8] (ldr, ldo ignored); register-direct epilogue
// pf0 / pf1: regions (the NEXT launches' weight matrices) that the launch's spare workgroups read into the Infinity Cache (common.h prefetch_region)
int r3_split(const float* x, void* hi_bf16, void* lo_i8, size_t n, hipStream_t s);
int r3_merge(const void* hi_bf16, const void* lo_i8, const int64_t* pos, float* out, int R, int L, int H, int gather, hipStream_t s, int src_panel = 0);
int ln_stat_parts(int n_cols);      // 96-column blocks of a gemm_ln_prod of n_cols columns
int ln_stat_slots(int n_cols);      // slots per row of the partial row-sum table [M][slots][2] it fills
int gemm_ln_cons(const void* A, int lda, const void* Wf, int ldw, const float* st_in, const float* colc, const float* cold,
                 float eps, int hidden, int gelu, void* out_lp, int ldo, int M, int N, int K, hipStream_t s, int out_panel = 0,
                 const void* pf = nullptr, size_t pf_bytes = 0, int a_panel = 0);      // pf: prefetch region of the panel form (as gemm_ln_prod3_panel)
// out_panel (gelu form, ffn_up_2pass_legal shapes): out_lp leaves in the fragment-major panel layout (gemm_prod.hip) instead of row-major
// fused QKV projection + self-attention (bf16, L <= 128); st_in NULL: plain bias, else LayerNorm folded (colc/cold);
// config 1: two workgroups per CU (2-stage ring), 2: one workgroup per CU (3-stage ring)
int gemm_qkv_attn(const void* A, int lda, const void* W, int ldw, const float* bias, const float* st_in, const float* colc,
                  const float* cold, float eps, int hidden, const int64_t* mask, void* ctx, int ldo, int B, int L, int heads,
                  int K, int config, hipStream_t s);
// round 3: the same fused launch with one workgroup per (sequence, three heads) (qkv_attn3.hip); needs qkv_attn3_eligible
int qkv_attn3_eligible(int L, int heads, int K);
void set_q3_trace(void* p);  // cpt_debug_gemm_trace: per-workgroup phase stamps of the kernel
void set_q3_abl(int v);     // diagnostic builds (-DCPT_ABLATION): ablation bits of the kernel, see qkv_attn3.hip
int gemm_qkv_attn3(const void* A, int lda, const void* W, int ldw, const float* bias, const float* st_in, const float* colc,
                   const float* cold, float eps, int hidden, const int64_t* mask, void* ctx, int ldo, int B, int L, int heads,
                   int K, hipStream_t s, int w_tiled = 0, int ctx_panel = 0, int a_panel = 0);     // a_panel (round 5): A = the residual stream's hi part in the panel layout     // w_tiled: W is the K-tile-major copy made by retile_k32; ctx_panel: ctx leaves in the panel layout (gemm_prod.hip), rows padded to a multiple of 32
// dst[K / 32][N][32] = src[N][K] (bf16): every K-tile of 32 of all rows contiguous (64 bytes per row, rows adjacent)
int retile_k32(const void* src, void* dst, int N, int K, hipStream_t s);
int select_regions(const float* logits, int V, const int64_t* color_ids, int C, const int* query_first, int Q,
                   int64_t none_id, int divide_by_none, int64_t* out_idx, float* out_score, hipStream_t s);
int argmax_columns(const float* logits, int V, const int64_t* ids, int n_ids, int R, int64_t* out_idx, float* out_val, hipStream_t s);
int fold_ln_weights(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* colc,
                    float* cold, int N, int K, hipStream_t s);
void set_splitk_target(int v);
void set_wgrad_pair(int v);  // training backward: the layer's weight gradients as two paired launches (1, default) or four single ones (0)
void set_attn_qt_all(int v); // stand-alone attention, L > 128: the query tiles of a (sequence, head) as neighbouring workgroups of one XCD (1, default)
// bf16x3 parity mode: FFN-up over the split operands (K3 = 3K) with the GELU epilogue writing h's [M][hi | hi | lo] split copy (ld 3N)
int gemm_gelu_x3(const void* A3, int lda, const void* W3, int ldw, const float* bias, void* out_split, int M, int N, int K3, hipStream_t s);
void set_fwd_split2(int v);  // training forward: FFN-down as two split-K partial matrices summed by the LayerNorm pass (1, default)
void set_qkv_2pass(int v);   // stand-alone LayerNorm-consumer QKV projection through the GELU-less two-pass kernel (1, default)
void set_nn_tile256(int v);
void set_nn_split2(int v);   // training backward: data gradients in front of a LayerNorm backward as two K-split bf16 partial matrices (1, default)
void set_ln_lean(int v);     // training forward: LayerNorm launches without fp32 output, residual re-formed by the next row pass (1, default)
void set_attn_bwd_split(int v);   // attention backward: one workgroup per PHASE of a (sequence, head) pair where the launch fills less than half the chip (1, default)
void set_narrow_tiles(int v);   // FFN-up forward / GELU-gradient GEMMs at few rows on 64 x 96 / 64 x 128 tiles (1, default)
void set_qkv_defer(int v);   // training backward: Q|K|V weight-gradient partial sums inside the next layer's three-problem weight-gradient launch (1, default)
void set_train_tail(int v);  // training step: the last encoder layer behind the attention on the head rows only (1, default)
void set_bias_fuse(int v);   // training backward: bias-gradient column sums inside their producers (bit 0 b_in, bit 1 b_qkv)
void set_lnb_rpb(int v);     // LayerNorm backward: rows per workgroup of the two-stage column-sum form (experiments)
void set_wgrad_tn(int v);    // 1 (default): bf16 weight gradients through the TN GEMM; 0: explicit operand transposes
void set_attn_bwd_variant(int v);
void set_gemm_variant(int v);
void set_gemm_abl(int v);
void set_gemm_skew(int v);
// FFN-up as one 384 x 256 two-pass kernel with its epilogue pipelined under the second pass (gemm_ffn.hip)
int ffn_up_2pass_legal(int M, int N, int K);
int ffn_up_2pass_preferred(int M, int N, int K);
int gemm_ffn_up_2pass(const void* A, int lda, const void* Wf, int ldw, const float* st_in, int st_parts, const float* colc, const float* cold,
                      float eps, int hidden, void* out, int ldo, int M, int N, int K, void* trace, int abl, hipStream_t s, int out_panel = 0,
                      const void* pf = nullptr, size_t pf_bytes = 0, int gelu = 1, int a_panel = 0);     // gelu 0: plain LayerNorm-consumer GEMM (stand-alone QKV projection); a_panel: A = the panel residual stream (round 5)
// round 4: the same consumer GEMMs on one wave per SIMD (gemm_ffn4.hip); same shapes, same bits
int gemm_lncons4(const void* A, int lda, const void* Wf, int ldw, const float* st_in, int st_parts, const float* colc, const float* cold,
                 float eps, int hidden, void* out, int ldo, int M, int N, int K, hipStream_t s, int out_panel, const void* pf, size_t pf_bytes, int gelu);
int lncons4_enabled();
void set_lncons4(int v);
void set_ffn_dma_late(int v);
void set_prod_abl(int v);    // timing experiments of the panel producer (gemm_prod.hip)
void set_prod_waves(int v);  // wave shape of the panel producer: 8 (4 x 2 waves of 32 x 96) or 4 (4 x 1 waves of 32 x 192)
// Per-CALL kernel choice for the operator-level test entry points (cpt_gemm_tile, cpt_gemm_ln_cons_tile, cpt_gemm_ln_prod3_panel_waves, the
// waves argument of cpt_gemm_ln_prod3_rpanel): thread-local, set for the duration of one entry-point call by an OverrideScope and restored on
// return -- no process-global state, other threads and later calls are untouched.  -1 = the library's own choice.
struct CallOverride { int gemm_variant = -1; int prod_waves = -1; };
CallOverride& call_override();
struct OverrideScope {
    CallOverride saved;
    OverrideScope(int gemm_variant, int prod_waves) : saved(call_override()) { call_override().gemm_variant = gemm_variant; call_override().prod_waves = prod_waves; }
    ~OverrideScope() { call_override() = saved; }
};
void set_gemm_trace(void* p);
void set_gemm_trace_filter(int epi, int k);

}  // namespace cpt
