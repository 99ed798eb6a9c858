// Few-shot training step of the CPT hot path: forward with saved activations, explicit backward,
// fused AdamW.  Replaces autograd over /root/reference/Oscar/oscar/modeling/modeling_rec.py:137-152 +
// modeling_bert.py:199-279 as driven by Oscar/oscar/fewshot/refcoco_cpt.py:231-249.
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>

#include "kernels.h"

namespace cpt { int abi_fail(int code, const char* fmt, ...); int abi_check(int rc, const char* what); }
using cpt::abi_check;
using cpt::abi_fail;

namespace cpt { CPT_SWITCH(int g_wgrad_tn, 1); void set_wgrad_tn(int v) { CPT_SWITCH_SET(g_wgrad_tn = v); (void)v; } }
using cpt::g_wgrad_tn;
// cpt_set_tuning(18, bits): bias-gradient column sums inside their producers -- bit 0: b_in in the GELU-gradient epilogue of
// dgrad(ffn down) (round 6: as partial rows added up by a later launch's column-sum job; round 3's end-of-launch atomics cost more than the colsum launch, gemm.hip), bit 1: b_qkv in the attention
// backward kernel (on: 43.3 vs 42.0 us per launch, no colsum launch); a cleared bit runs the stand-alone colsum launch instead
// cpt_set_tuning(19, v): 2 (default) = FFN down | FFN up | attention output in ONE launch (216 workgroups, whole contraction each) + Q|K|V alone
// with its own three-way split, where the shapes allow it (else as 1); 1 = two paired launches (gemm_tn_pair); 0 = four single ones
namespace cpt { CPT_SWITCH(int g_wgrad_pair, 2); void set_wgrad_pair(int v) { CPT_SWITCH_SET(g_wgrad_pair = v); (void)v; } }
using cpt::g_wgrad_pair;
// cpt_set_tuning(22, v): 1 (default) = the FFN-down of the training forward runs on 128 x 192 tiles with K split in two (gemm_img_proj's
// kernel) and the dropout + residual + LayerNorm pass adds the two partial matrices, 0 = 64 x 192 tiles over the whole K
namespace cpt { CPT_SWITCH(int g_fwd_split2, 1); void set_fwd_split2(int v) { CPT_SWITCH_SET(g_fwd_split2 = v); (void)v; } }
using cpt::g_fwd_split2;
// cpt_set_tuning(33, v): 1 (default) = the LAST encoder layer's attention output, FFN and both LayerNorms run on the head's rows only (one per
// sequence: the [MASK] row, or [CLS] for the NSP head) in the training forward AND backward -- every other row of that layer's output is dead
// (the loss reads B rows; modeling_rec.py:142-150) and so is its gradient; 0 = all rows
// cpt_set_tuning(36, v): 1 (default) = with hidden dropout the LayerNorm launches of the training forward write no fp32 output; the row pass behind
// re-forms the residual from the pre-LayerNorm rows kept for the backward, 0 = fp32 outputs written and read back
namespace cpt { CPT_SWITCH(int g_ln_lean, 1); void set_ln_lean(int v) { CPT_SWITCH_SET(g_ln_lean = v); (void)v; } }
using cpt::g_ln_lean;
// cpt_set_tuning(37, v): 1 (default) = the K-split partial matrices of a layer's Q|K|V weight gradient are added up by the spare workgroups of the NEXT layer's
// three-problem weight-gradient launch (no reduction launch), 0 = reduction launch behind the split launch
namespace cpt { CPT_SWITCH(int g_qkv_defer, 1); void set_qkv_defer(int v) { CPT_SWITCH_SET(g_qkv_defer = v); (void)v; } }
using cpt::g_qkv_defer;
namespace cpt { CPT_SWITCH(int g_train_tail, 1); void set_train_tail(int v) { CPT_SWITCH_SET(g_train_tail = v); (void)v; } }
using cpt::g_train_tail;
namespace cpt { CPT_SWITCH(int g_bias_fuse, 3); void set_bias_fuse(int v) { CPT_SWITCH_SET(g_bias_fuse = v); (void)v; } }
using cpt::g_bias_fuse;

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline int up64(int x) { return (x + 63) / 64 * 64; }

struct TrainLayout {
    size_t x_f32, a_f32, layer0, layer_stride;
    size_t o_xin, o_qkv, o_ctx, o_pre1, o_a, o_u, o_h, o_pre2, o_st1, o_st2, o_ast, o_kb1, o_kb2;     // offsets inside a layer block (st1 / st2, round 6: (mean, rstd) per row of the two LayerNorms, read by their backward)
    size_t xout, imgp, imgpre, rows, uh, t2, dlogits, loss;
    size_t dx, dpre, da, dpre_lp, dlp2, dctx, dbig, tA, tB, wT, gimg, dl_lp, dt2, duh, duh_lp, drows, dimg, dimg_lp, dmask, dmask_lp;
    size_t t_ctx, t_xres, t_pre1, t_a, t_af, t_u, t_h, t_pre2, t_st1, t_st2, t_kb1, t_kb2, t_dpre, t_dlpF, t_dlpA, t_dbig, t_da, t_dctx;      // round 6: the pruned last layer's compact [B][.] activations and gradients
    size_t qkvp, qkvp_bytes;      // round 6: partial matrices of the Q|K|V weight gradient's K-split launch, added up by spare workgroups of the NEXT layer's three-problem weight-gradient launch
    size_t lnp[2], lnp_bytes, csp; int csp_rows;      // round 6: partial column sums left for a later launch's column-sum job -- two alternating LayerNorm-backward tables, the FFN-up bias rows of the GELU-gradient GEMM
    size_t sA, sW, sA_bytes, sW_bytes;      // bf16x3: split copies of a GEMM's two fp32 operands ([rows][hi | hi | lo] and [rows][hi | lo | hi])
    size_t total, tA_bytes, tB_bytes;
    int Mp, Bp, Vp, Rp;
};

// Rh: rows the head runs on -- the labelled positions (cpt_batch.n_rows), or one per sequence
TrainLayout train_layout(const cpt_dims& d, int B, int Lt, int Li, int Rh) {
    const size_t es = d.dtype == CPT_BF16 ? 2 : 4;
    const size_t L = (size_t)Lt + Li, M = (size_t)B * L, H = d.hidden, I = d.inter, V = d.vocab, Dp = d.img_dim_pad;
    const size_t R = (size_t)B * Li;
    TrainLayout w;
    w.Mp = up64((int)M); w.Bp = up64(Rh); w.Vp = up64((int)V); w.Rp = up64((int)std::max<size_t>(R, 1));
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t p = o; o += al(bytes); return p; };
    w.x_f32 = take(M * H * 4);
    w.a_f32 = take(M * H * 4);
    {   // one block per layer
        size_t q = 0;
        auto sub = [&](size_t bytes) { size_t p = q; q += al(bytes); return p; };
        w.o_xin = sub(M * H * es); w.o_qkv = sub(M * 3 * H * es); w.o_ctx = sub(M * H * es); w.o_pre1 = sub(M * H * 4);
        w.o_a = sub(M * H * es); w.o_u = sub(M * I * es); w.o_h = sub(M * I * es); w.o_pre2 = sub(M * H * 4);
        w.o_st1 = sub(M * 2 * 4); w.o_st2 = sub(M * 2 * 4);
        w.o_kb1 = sub(M * 64 * 2); w.o_kb2 = sub(M * 64 * 2);      // keep bits of the two hidden dropouts (one 16-bit word per lane of the row pass), read back by the LayerNorm backward
        w.o_ast = sub((size_t)B * d.heads * L * 2 * 4);      // softmax statistics (row max, 1 / row sum) of every (sequence, head, query): the one-pass attention backward reads them
        w.layer_stride = q;
        w.layer0 = take(q * d.layers);
    }
    w.xout = take(M * H * es);
    w.imgp = take(R * Dp * es);
    w.imgpre = take(R * H * 4);
    w.rows = take((size_t)Rh * H * es);
    w.uh = take((size_t)Rh * H * 4);
    w.t2 = take((size_t)Rh * H * es);
    w.dlogits = take((size_t)Rh * V * 4);
    w.loss = take(256);
    // backward temporaries
    w.dx = take(M * H * 4);
    w.dpre = take(M * H * 4);
    w.da = take(M * H * 4);
    w.dpre_lp = take(M * H * es);
    w.dlp2 = take(M * H * es);        // round 3: the attention-side LayerNorm backward's low-precision output when the FFN-side one must outlive it (three-problem weight-gradient launch)
    w.dctx = take(M * H * es);
    w.dbig = take(M * std::max(3 * H, I) * es);
    const size_t cols = std::max<size_t>({(size_t)w.Mp, (size_t)w.Bp, (size_t)w.Rp});
    w.tA_bytes = std::max<size_t>({3 * H, I, V}) * cols * es;
    w.tA = take(w.tA_bytes);
    w.tB_bytes = std::max<size_t>({I, Dp, H}) * cols * es;      // also the scratch of ln_bwd's two-stage column sums
    w.tB = take(w.tB_bytes);
    w.wT = take(std::max<size_t>({3 * H * H, I * H, H * (size_t)w.Vp}) * es);
    w.gimg = take(H * Dp * 4);
    w.dl_lp = take((size_t)Rh * w.Vp * es);
    w.dt2 = take((size_t)Rh * H * 4);
    w.duh = take((size_t)Rh * H * 4);
    w.duh_lp = take((size_t)Rh * H * es);
    w.drows = take((size_t)Rh * H * 4);
    w.dimg = take(R * H * 4);
    w.dimg_lp = take(R * H * es);
    w.dmask = take(M * H * 4);        // dropout: masked gradient of a dense output (the unmasked one feeds the residual path)
    w.dmask_lp = take(M * H * es);
    w.lnp_bytes = (M / 4 + 2) * 3 * H * 4;      // (ln_bwd_part_rows(M) <= M / 4 + 1 rows of [3][H])
    w.lnp[0] = take(w.lnp_bytes); w.lnp[1] = take(w.lnp_bytes);
    w.csp_rows = (int)((M + 31) / 32);
    w.csp = take((size_t)w.csp_rows * I * 4);
    {
        // (gemm_tn splits K at most 256 / tiles ways, at most eight)
        const size_t tiles = (3 * H / 128) * (H % 192 == 0 ? H / 192 : std::max<size_t>(H / 128, 1));
        const size_t smax = (d.dtype == CPT_BF16 && tiles > 0) ? std::min<size_t>(8, 256 / std::max<size_t>(tiles, 1)) : 0;
        w.qkvp_bytes = smax > 1 ? smax * 3 * H * H * 4 : 0;
        w.qkvp = take(w.qkvp_bytes);
    }
    {
        const size_t Bq = (size_t)B;
        w.t_ctx = take(Bq * H * es); w.t_xres = take(Bq * H * 4); w.t_pre1 = take(Bq * H * 4); w.t_a = take(Bq * H * es); w.t_af = take(Bq * H * 4);
        w.t_u = take(Bq * I * es); w.t_h = take(Bq * I * es); w.t_pre2 = take(Bq * H * 4); w.t_st1 = take(Bq * 8); w.t_st2 = take(Bq * 8); w.t_kb1 = take(Bq * 128); w.t_kb2 = take(Bq * 128);
        w.t_dpre = take(Bq * H * 4); w.t_dlpF = take(Bq * H * es); w.t_dlpA = take(Bq * H * es); w.t_dbig = take(Bq * I * es); w.t_da = take(Bq * H * 4);
        w.t_dctx = take(Bq * H * es);
    }
    w.sA = w.sW = 0; w.sA_bytes = w.sW_bytes = 0;
    if (d.dtype == CPT_BF16X3_MASTERS) {
        // A operands: activations [M][<= max(3H, I)], transposed gradients [<= max(3H, I)][Mp], the head's [Rh][Vp] / [V][Bp], region rows [R][Dp] / [H][Rp]
        const size_t wide = std::max<size_t>(3 * H, I);
        const size_t a_el = std::max<size_t>({wide * (size_t)w.Mp, (size_t)w.Vp * (size_t)w.Bp, (size_t)w.Rp * Dp, H * (size_t)w.Rp});
        // W operands: weights [<= max(3H, I, V)][<= max(H, I)] and their transposes, transposed activations [<= max(I, Dp, H)][cols]
        const size_t w_el = std::max<size_t>({3 * H * H, I * H, H * (size_t)w.Vp, H * Dp, std::max<size_t>({I, Dp, H}) * cols});
        w.sA_bytes = 6 * a_el; w.sW_bytes = 6 * w_el;
        w.sA = take(w.sA_bytes); w.sW = take(w.sW_bytes);
    }
    w.total = o;
    return w;
}

int check_common(const cpt_model* m, const cpt_batch* b, void* ws, size_t ws_bytes, const TrainLayout& w, const char* who) {
    const cpt_dims& d = m->dims;
    if (b->B <= 0 || b->Lt <= 0 || b->Li < 0) return abi_fail(CPT_ERR_SHAPE, "%s: bad batch shape", who);
    if (b->n_rows < 0 || (long)b->n_rows > (long)b->B * (b->Lt + b->Li) || (b->n_rows > 0 && !b->row_seq))
        return abi_fail(CPT_ERR_SHAPE, "%s: n_rows %d needs row_seq and at most B * L = %ld rows", who, b->n_rows, (long)b->B * (b->Lt + b->Li));
    if (d.heads <= 0 || d.hidden != d.heads * 64) return abi_fail(CPT_ERR_SHAPE, "%s: head_dim must be 64", who);
    if (d.dtype == CPT_BF16X3)
        return abi_fail(CPT_ERR_DTYPE, "%s: CPT_BF16X3 describes the standing split weight copies cpt_model_fwd reads; the training step takes the fp32 "
                        "master weights under CPT_BF16X3_MASTERS (ABI 6)", who);
    if (d.dtype != CPT_F32 && d.dtype != CPT_BF16 && d.dtype != CPT_BF16X3_MASTERS) return abi_fail(CPT_ERR_DTYPE, "%s: dtype %d", who, d.dtype);
    if (d.img_dim_pad < d.img_dim || d.img_dim_pad % 64) return abi_fail(CPT_ERR_ALIGN, "%s: img_dim_pad must be a multiple of 64 for training", who);
    if (d.hidden % 64 || d.inter % 64) return abi_fail(CPT_ERR_ALIGN, "%s: hidden/intermediate sizes must be multiples of 64", who);
    const bool nsp = !m->w_tr && !m->w_dec && m->w_pool && m->w_rel && d.n_rel > 0;
    if (!b->labels || (!nsp && !b->mask_pos)) return abi_fail(CPT_ERR_NULL, "%s: labels (and, for the MLM head, mask_pos) are required", who);
    if (b->Li > 0 && !b->img_feats) return abi_fail(CPT_ERR_NULL, "%s: img_feats is NULL", who);
    if (!nsp && (!m->w_tr || !m->w_dec)) return abi_fail(CPT_ERR_NULL, "%s: model has neither the MLM head nor the NSP head (pooler + seq_relationship)", who);
    if (nsp && d.n_rel > 64) return abi_fail(CPT_ERR_SHAPE, "%s: n_rel %d > 64", who, d.n_rel);
    if (ws_bytes < w.total) return abi_fail(CPT_ERR_WORKSPACE, "%s: workspace %zu < required %zu bytes", who, ws_bytes, w.total);
    if ((uintptr_t)ws & 255) return abi_fail(CPT_ERR_ALIGN, "%s: workspace must be 256-byte aligned", who);
    return CPT_OK;
}

// Dropout sites of the model (counter word 3 of the Philox stream, dropout.h): 0 = embeddings (text and region rows,
// after their LayerNorms); layer l: 1 + 3l attention probabilities, 2 + 3l BertSelfOutput, 3 + 3l BertOutput.
cpt::DropSpec drop_spec(const cpt_dropout* d, int site, bool attn) {
    cpt::DropSpec s = {};
    if (!d) return s;
    const double p = attn ? d->p_attn : d->p_hidden;
    if (!(p > 0.0)) return s;
    s.k0 = (uint32_t)d->seed; s.k1 = (uint32_t)(d->seed >> 32);
    s.step = (uint32_t)d->step; s.site = (uint32_t)site;
    const double full = attn ? 65536.0 : 4294967296.0;
    double t = p * full;
    if (t < 1.0) t = 1.0;
    if (t > full - 1.0) t = full - 1.0;
    s.thresh = (uint32_t)(t + 0.5);
    s.scale = (float)(1.0 / (1.0 - (double)s.thresh / full));      // exact effective rate: E[mask * scale] = 1
    return s;
}
int check_drop(const cpt_dropout* d, const char* who) {
    if (!d) return CPT_OK;
    if (!(d->p_hidden >= 0.f && d->p_hidden < 1.f) || !(d->p_attn >= 0.f && d->p_attn < 1.f))
        return abi_fail(CPT_ERR_SHAPE, "%s: dropout probabilities must be in [0, 1)", who);
    return CPT_OK;
}

// the pruned last layer (g_train_tail): bf16 step, one head row per sequence (no label grid), tile-aligned hidden sizes, 2-D mask
bool train_tail_on(const cpt_dims& d, const cpt_batch* b) {
    return g_train_tail && g_wgrad_tn && d.dtype == CPT_BF16 && b->n_rows == 0 && d.layers >= 1 && d.hidden % 192 == 0 && d.inter % 192 == 0 &&
           (long)b->B * 4 <= (long)b->B * (b->Lt + b->Li);
}

}  // namespace

// One Linear-shaped product out[Mr][N] = A[Mr][K] . W[N][K]^T of the training step.  fp32 / bf16: the library GEMM of that type.  bf16x3 (the parity mode at
// MFMA-bf16 rates, include/cpt_hip.h): both fp32 operands are split on the spot into [hi | hi | lo] / [hi | lo | hi] bf16 copies and the bf16 kernel runs
// over K' = 3K -- a.w ~ hi.hi + hi.lo + lo.hi with fp32 accumulation; everything outside the GEMMs runs as in fp32 mode.  (The weights change every
// step, so unlike inference there is no standing split copy of them.)
#define CPT_TRAIN_GM                                                                                                                                    \
    auto gm = [&](int epi, const void* A, int lda, const void* W, int ldw, const float* bias, const float* resid, int ldr, void* out, int out_dt,      \
                  int ldo, int Mr, int N, int K, hipStream_t st) -> int {                                                                               \
        if (!x3) return cpt::gemm(dt, epi, A, lda, W, ldw, bias, resid, ldr, out, out_dt, ldo, Mr, N, K, st);                                      \
        if (out_dt != CPT_F32) return CPT_ERR_DTYPE;                                                                                                    \
        if ((size_t)Mr * K * 6 > w.sA_bytes || (size_t)N * K * 6 > w.sW_bytes) return CPT_ERR_WORKSPACE;                                               \
        int r3 = cpt::split3((const float*)A, lda, ws + w.sA, Mr, K, 0, st);                                                                           \
        if (r3 != CPT_OK) return r3;                                                                                                                    \
        r3 = cpt::split3((const float*)W, ldw, ws + w.sW, N, K, 1, st);                                                                                \
        if (r3 != CPT_OK) return r3;                                                                                                                    \
        return cpt::gemm(CPT_BF16, epi, ws + w.sA, 3 * K, ws + w.sW, 3 * K, bias, resid, ldr, out, CPT_F32, ldo, Mr, N, 3 * K, st);                   \
    }

#define TRY(expr, what)                        \
    do {                                       \
        int rc__ = abi_check((expr), what);    \
        if (rc__ != CPT_OK) return rc__;       \
    } while (0)

extern "C" {

size_t cpt_train_workspace_bytes(const cpt_dims* d, int B, int Lt, int Li) {
    if (!d || B <= 0 || Lt <= 0 || Li < 0) return 0;
    return train_layout(*d, B, Lt, Li, B).total;
}
size_t cpt_train_workspace_bytes_rows(const cpt_dims* d, int B, int Lt, int Li, int n_rows) {
    if (!d || B <= 0 || Lt <= 0 || Li < 0 || n_rows < 0 || (long)n_rows > (long)B * (Lt + Li)) return 0;
    return train_layout(*d, B, Lt, Li, n_rows > 0 ? n_rows : B).total;
}

int cpt_train_fwd(const cpt_model* m, const cpt_batch* b, const cpt_outputs* o, void* workspace,
                  size_t workspace_bytes, void* stream) {
    return cpt_train_fwd_ex(m, b, o, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}

// bf16x3 training: ONE predicate decides whether a step's attention runs on the mode's split-operand MFMA kernels -- forward AND backward
// (the forward used to pick by a macro and the backward on its own, and the shape gate asked the fp32 kernel either way).
static inline bool train_x3_attention(const cpt_dims& d, int L, int m3d) {
    return d.dtype == CPT_BF16X3_MASTERS && !m3d && cpt::attention_x3_supported(L) && cpt::attention_bwd_x3_supported(L, m3d);
}

int cpt_train_fwd_ex(const cpt_model* m, const cpt_batch* b, const cpt_outputs* o, void* workspace,
                     size_t workspace_bytes, void* stream, cpt_bucket_fn before_bucket, void* user, const cpt_dropout* drop) {
    if (!m || !b || !o || !workspace) return abi_fail(CPT_ERR_NULL, "cpt_train_fwd: null argument");
    if (int rcd = check_drop(drop, "cpt_train_fwd")) return rcd;
    const bool ph = drop && drop->p_hidden > 0.f, pa = drop && drop->p_attn > 0.f;
    const cpt_dims& d = m->dims;
    const TrainLayout w = train_layout(d, b->B, b->Lt, b->Li, b->n_rows > 0 ? b->n_rows : b->B);
    int rc = check_common(m, b, workspace, workspace_bytes, w, "cpt_train_fwd");
    if (rc) return rc;
    const bool nsp = !m->w_tr && !m->w_dec;
    if (nsp && b->n_rows > 0) return abi_fail(CPT_ERR_SHAPE, "cpt_train_fwd: label grids (n_rows) belong to the MLM head, not the NSP head");
    if (!o->loss || (nsp ? !o->rel : !o->logits)) return abi_fail(CPT_ERR_NULL, "cpt_train_fwd: loss and logits (NSP head: rel) outputs are required");
    hipStream_t s = (hipStream_t)stream;
    const bool x3 = d.dtype == CPT_BF16X3_MASTERS;
    const int B = b->B, Lt = b->Lt, Li = b->Li, L = Lt + Li, M = B * L, H = d.hidden, I = d.inter, dt = x3 ? CPT_F32 : d.dtype;
    if (L > 288) return abi_fail(CPT_ERR_SHAPE, "cpt_train_fwd: sequence length %d > 288", L);
    const int m3d = (b->mask_3d && b->attn_mask) ? 1 : 0;
    const bool x3_attn = train_x3_attention(d, L, m3d);
    const bool tail = train_tail_on(d, b) && !m3d;
    if (!x3_attn && !cpt::attention_bwd_supported(dt, L, pa ? 1 : 0, m3d))       // reject here, not after the forward has run (the backward's attention kernel sets the limit)
    {
        int lmax = L;
        while (lmax > 0 && !train_x3_attention(d, lmax, m3d) && !cpt::attention_bwd_supported(dt, lmax, pa ? 1 : 0, m3d)) --lmax;
        return abi_fail(CPT_ERR_SHAPE, "cpt_train_fwd: no attention backward for dtype %d at sequence length %d%s (longest supported: %d)", dt, L,
                        pa ? " with attention dropout" : "", lmax);
    }
    unsigned char* ws = (unsigned char*)workspace;
    CPT_TRAIN_GM;
    float* x_f32 = (float*)(ws + w.x_f32);
    float* a_f32 = (float*)(ws + w.a_f32);
    auto LB = [&](int l, size_t off) { return (void*)(ws + w.layer0 + (size_t)l * w.layer_stride + off); };
    // parameter buckets (data-parallel training): the host callback runs BEFORE the first launch that reads the
    // bucket's parameters, so the caller can make `stream` wait for that bucket's all-gather (include/cpt_hip.h)
    auto need = [&](int bucket) { if (before_bucket) before_bucket(user, bucket); };

    need(0);
    // (round 6: the embedding launch also clears the loss accumulator and the finish ticket of the cross-entropy launch at the end of this forward --
    // no memset launch; with regions in bf16 it carries the region features' pad + cast as well, as the inference forward's first launch does)
    float* loss_ws = (float*)(ws + w.loss);
    unsigned* ce_ticket = (unsigned*)(ws + w.loss + 16);
    const bool emb_pad = Li > 0 && dt == CPT_BF16 && d.img_dim_pad % 8 == 0 && !((uintptr_t)b->img_feats & 7);
    // (round 6: BertEmbeddings' dropout on the text rows and modeling_bert.py:266 on the region rows are applied by the launches that write those rows --
    // the same site mask at the same element indices as the dropout pass over all rows they replace)
    const cpt::DropSpec edrop = drop_spec(drop, 0, false);
    if (emb_pad)
        TRY(cpt::embed_ln_pad_cast(b->input_ids, b->token_type, b->position_ids, m->word_emb, m->pos_emb, m->type_emb, m->emb_ln_g, m->emb_ln_b, d.ln_eps,
                                   LB(0, w.o_xin), nullptr, B, Lt, L, H, d.vocab, d.max_pos, d.type_vocab, b->img_feats, ws + w.imgp, B * Li, d.img_dim, d.img_dim_pad, s, 0,
                                   x_f32, o->loss, ce_ticket, ph ? &edrop : nullptr), "embed_ln + dropout + pad_cast(img_feats)");
    else
    TRY(cpt::embed_ln(b->input_ids, b->token_type, b->position_ids, m->word_emb, m->pos_emb, m->type_emb, m->emb_ln_g,
                      m->emb_ln_b, d.ln_eps, x_f32, LB(0, w.o_xin), dt, B, Lt, L, H, d.vocab, d.max_pos, d.type_vocab, s, nullptr, 0, o->loss, ce_ticket, ph ? &edrop : nullptr), "embed_ln + dropout");
    if (Li > 0) {
        void* imgp = ws + w.imgp;
        float* imgpre = (float*)(ws + w.imgpre);
        if (!emb_pad) TRY(cpt::pad_cast(b->img_feats, imgp, dt, B * Li, d.img_dim, d.img_dim_pad, s), "pad_cast(img_feats)");
        TRY(gm(CPT_EPI_NONE, imgp, d.img_dim_pad, m->w_img, d.img_dim_pad, m->b_img, nullptr, 0, imgpre, CPT_F32, H,
                      B * Li, H, d.img_dim_pad, s), "gemm(img_embedding)");
        const bool iln = d.use_img_ln && m->img_ln_g;       // use_img_layernorm = 0 (modeling_bert.py:263): the projection is used as is
        TRY(cpt::layernorm_rows_ex(imgpre, iln ? m->img_ln_g : nullptr, iln ? m->img_ln_b : nullptr, d.img_ln_eps, x_f32, LB(0, w.o_xin), dt, B * Li, H, Li, L, Lt, 0, s,
                                   nullptr, nullptr, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ph ? &edrop : nullptr),
            "layernorm(img) + dropout");
    }
    // dense layers with fp32 output (attn-out, FFN-down): at small row counts K is split over workgroups (gemm_nt_split)
    auto dense_f32 = [&](const void* A, int lda, const void* W, int K, const float* bias, const float* resid, void* out, int N, const char* what) -> int {
        if (dt == CPT_BF16 && g_wgrad_tn) {
            const int rs = cpt::gemm_nt_split(A, lda, W, K, bias, resid, N, (float*)out, N, M, N, K, ws + w.tA, w.tA_bytes, s);
            if (rs == CPT_OK) return abi_check(CPT_OK, what);
            if (rs != CPT_ERR_SHAPE) return abi_check(rs, what);
        }
        return abi_check(gm(resid ? CPT_EPI_RESID : CPT_EPI_NONE, A, lda, W, K, bias, resid, N, out, CPT_F32, N, M, N, K, s), what);
    };
    // round 6: with hidden dropout every residual add sits in a row pass, and that pass can re-form the LayerNorm output it adds from the
    // pre-LayerNorm rows the backward keeps anyway (+ (mean, rstd), gain, shift): the LayerNorm launches then write no fp32 output
    // (11.8 of 53-65 MB per launch at 3840 rows).  x_f32 is still written where the pruned last layer gathers its residual rows from it.
    const bool lean = ph && g_ln_lean && M >= 2048;      // (at few rows the row passes are pure latency chains: the extra (mean, rstd), gain and shift loads cost what the 1.5 MB buy)
    for (int l = 0; l < d.layers; ++l) {
        const cpt_layer& y = m->layers[l];
        need(1 + l);
        void* xin = LB(l, w.o_xin);
        const bool lean1 = lean && l > 0;       // (layer 0 adds the embedding rows: a real buffer)
        const float* res1 = lean1 ? (const float*)LB(l - 1, w.o_pre2) : x_f32;
        const float* res1_st = lean1 ? (const float*)LB(l - 1, w.o_st2) : nullptr;
        const float* res1_g = lean1 ? m->layers[l - 1].ln2_g : nullptr; const float* res1_b = lean1 ? m->layers[l - 1].ln2_b : nullptr;
        const float* res2 = lean ? (const float*)LB(l, w.o_pre1) : a_f32;
        const float* res2_st = lean ? (const float*)LB(l, w.o_st1) : nullptr;
        const float* res2_g = lean ? y.ln1_g : nullptr; const float* res2_b = lean ? y.ln1_b : nullptr;
        float* a_out = lean ? nullptr : a_f32;
        float* x_out = (lean && !(tail && l == d.layers - 2)) ? nullptr : x_f32;
        void* xnext = l + 1 < d.layers ? LB(l + 1, w.o_xin) : (void*)(ws + w.xout);
        TRY(gm(CPT_EPI_NONE, xin, H, y.w_qkv, H, y.b_qkv, nullptr, 0, LB(l, w.o_qkv), dt, 3 * H, M, 3 * H, H, s), "gemm(qkv)");
        const cpt::DropSpec da_spec = drop_spec(drop, 1 + 3 * l, true);
        if (x3_attn)      // bf16x3: the mode's split-operand MFMA attention (fp32 ctx out), same dropout stream
            TRY(cpt::attention_x3((const float*)LB(l, w.o_qkv), b->attn_mask, (float*)LB(l, w.o_ctx), nullptr, B, L, d.heads, s, pa ? &da_spec : nullptr), "attention (split operands)");
        else
        TRY(cpt::attention(dt, LB(l, w.o_qkv), b->attn_mask, LB(l, w.o_ctx), nullptr, B, L, d.heads, s, pa ? &da_spec : nullptr, m3d, 0,
                           (dt == CPT_BF16 && !m3d) ? (float*)LB(l, w.o_ast) : nullptr), "attention");
        if (tail && l == d.layers - 1) {
            // ---- round 6: the rest of the last layer on the B head rows (row b = position pos[b] of sequence b) ----
            const cpt::RowMap rm = {nsp ? nullptr : b->mask_pos, L, 1};
            void* ctx_r = ws + w.t_ctx; float* xres_r = (float*)(ws + w.t_xres); float* pre1_r = (float*)(ws + w.t_pre1);
            void* a_r = ws + w.t_a; float* af_r = (float*)(ws + w.t_af); float* pre2_r = (float*)(ws + w.t_pre2);
            TRY(cpt::tail_gather2(LB(l, w.o_ctx), x_f32, rm.pos, ctx_r, xres_r, B, L, H, s), "gather(head rows of ctx, residual)");
            const cpt::DropSpec sp1 = drop_spec(drop, 2 + 3 * l, false), sp2 = drop_spec(drop, 3 + 3 * l, false);
            float* part = (float*)(ws + w.tA);
            int S1 = 0;
            int r1 = cpt::gemm_nt_partials(ctx_r, H, y.w_ao, H, y.b_ao, part, w.tA_bytes, B, H, H, s, &S1);
            if (r1 == CPT_ERR_SHAPE) {
                S1 = 1;
                r1 = gm(CPT_EPI_NONE, ctx_r, H, y.w_ao, H, y.b_ao, nullptr, 0, part, CPT_F32, H, B, H, H, s);
            }
            TRY(r1, "gemm(attn out, head rows)");
            TRY(cpt::layernorm_rows_ex(part, y.ln1_g, y.ln1_b, d.ln_eps, af_r, a_r, dt, B, H, B, 0, 0, 0, s, xres_r, ph ? &sp1 : nullptr, pre1_r, nullptr,
                                       S1, (size_t)B * H, 0, (float*)(ws + w.t_st1), &rm, (unsigned short*)(ws + w.t_kb1)), "dropout(attn out)+residual+layernorm (head rows)");
            TRY(cpt::gemm_gelu2(a_r, H, y.w_in, H, y.b_in, ws + w.t_u, ws + w.t_h, I, B, I, H, s), "gemm(ffn up)+gelu (head rows)");
            int S2 = 0;
            int r2 = cpt::gemm_nt_partials(ws + w.t_h, I, y.w_out, I, y.b_out, part, w.tA_bytes, B, H, I, s, &S2);
            if (r2 == CPT_ERR_SHAPE) {
                S2 = 1;
                r2 = gm(CPT_EPI_NONE, ws + w.t_h, I, y.w_out, I, y.b_out, nullptr, 0, part, CPT_F32, H, B, H, I, s);
            }
            TRY(r2, "gemm(ffn down, head rows)");
            // ... whose LayerNorm output IS the head's input (no gather of [MASK] / [CLS] rows below)
            TRY(cpt::layernorm_rows_ex(part, y.ln2_g, y.ln2_b, d.ln_eps, nullptr, ws + w.rows, dt, B, H, B, 0, 0, 0, s, af_r, ph ? &sp2 : nullptr, pre2_r, nullptr,
                                       S2, (size_t)B * H, 0, (float*)(ws + w.t_st2), &rm, (unsigned short*)(ws + w.t_kb2)), "dropout(ffn down)+residual+layernorm (head rows)");
            continue;
        }
        // round 3: where the dense layer's K is split over workgroups (few rows; or 2048..6144 rows with a long K), its partial matrices go
        // straight to the row pass behind it, which adds them in split order -- no reduction launch in between (cpt_set_tuning key 22)
        int Sp = 0;
        auto dense_parts = [&](const void* A, int lda, const void* W, int K, const float* bias, const char* what) -> int {      // CPT_OK: Sp partial matrices at tA
            Sp = 0;
            if (!(g_fwd_split2 && dt == CPT_BF16 && g_wgrad_tn)) return CPT_ERR_SHAPE;
            const int r = cpt::gemm_nt_partials(A, lda, W, K, bias, (float*)(ws + w.tA), w.tA_bytes, M, H, K, s, &Sp);
            if (r == CPT_ERR_SHAPE) return r;
            return abi_check(r, what);
        };
        const float* part = (const float*)(ws + w.tA);
        if (int rp = dense_parts(LB(l, w.o_ctx), H, y.w_ao, H, y.b_ao, "gemm(attn out, K split)"); rp != CPT_ERR_SHAPE) {
            if (rp) return rp;
            const cpt::DropSpec sp = drop_spec(drop, 2 + 3 * l, false);
            TRY(cpt::layernorm_rows_ex(part, y.ln1_g, y.ln1_b, d.ln_eps, a_out, LB(l, w.o_a), dt, M, H, M, 0, 0, 0, s,
                                       res1, ph ? &sp : nullptr, (float*)LB(l, w.o_pre1), nullptr, Sp, (size_t)M * H, 0, (float*)LB(l, w.o_st1), nullptr, (unsigned short*)LB(l, w.o_kb1),
                                       res1_st, res1_g, res1_b), "partials+dropout(attn out)+residual+layernorm");
        } else
        if (ph) {   // LN(dropout(dense(ctx)) + x): the residual add moves from the GEMM epilogue into the dropout pass
            if (int r_ = dense_f32(LB(l, w.o_ctx), H, y.w_ao, H, y.b_ao, nullptr, LB(l, w.o_pre1), H, "gemm(attn out)")) return r_;
            // dropout + residual + LayerNorm in one row pass (pre1 = dropout(dense) + x is stored for the backward pass)
            const cpt::DropSpec sp = drop_spec(drop, 2 + 3 * l, false);
            TRY(cpt::layernorm_rows_ex((const float*)LB(l, w.o_pre1), y.ln1_g, y.ln1_b, d.ln_eps, a_out, LB(l, w.o_a), dt, M, H, M, 0, 0, 0, s,
                                       res1, &sp, (float*)LB(l, w.o_pre1), nullptr, 1, 0, 0, (float*)LB(l, w.o_st1), nullptr, (unsigned short*)LB(l, w.o_kb1),
                                       res1_st, res1_g, res1_b), "dropout(attn out)+residual+layernorm");
        } else {
        if (int r_ = dense_f32(LB(l, w.o_ctx), H, y.w_ao, H, y.b_ao, x_f32, LB(l, w.o_pre1), H, "gemm(attn out)")) return r_;
        TRY(cpt::layernorm_rows_ex((const float*)LB(l, w.o_pre1), y.ln1_g, y.ln1_b, d.ln_eps, a_f32, LB(l, w.o_a), dt, M, H, M, 0, 0, 0, s,
                                   nullptr, nullptr, nullptr, nullptr, 1, 0, 0, (float*)LB(l, w.o_st1)), "layernorm(attn)");
        }
        if (dt == CPT_BF16 && H % 64 == 0 && I % 8 == 0) {
            TRY(cpt::gemm_gelu2(LB(l, w.o_a), H, y.w_in, H, y.b_in, LB(l, w.o_u), LB(l, w.o_h), I, M, I, H, s), "gemm(ffn up)+gelu");
        } else {
        TRY(gm(CPT_EPI_NONE, LB(l, w.o_a), H, y.w_in, H, y.b_in, nullptr, 0, LB(l, w.o_u), dt, I, M, I, H, s), "gemm(ffn up)");
        TRY(cpt::gelu_fwd(LB(l, w.o_u), LB(l, w.o_h), dt, (size_t)M * I, s), "gelu");
        }
        // (FFN-down the same way: at M = 3840 two 128 x 192 half-K workgroups per tile, 240 workgroups at twice the arithmetic intensity of the
        // 64 x 192 tiles; at 480 rows eight K slices per 64 x 192 tile)
        if (int rp = dense_parts(LB(l, w.o_h), I, y.w_out, I, y.b_out, "gemm(ffn down, K split)"); rp != CPT_ERR_SHAPE) {
            if (rp) return rp;
            const cpt::DropSpec sp = drop_spec(drop, 3 + 3 * l, false);
            TRY(cpt::layernorm_rows_ex(part, y.ln2_g, y.ln2_b, d.ln_eps, x_out, xnext, dt, M, H, M, 0, 0, 0, s,
                                       res2, ph ? &sp : nullptr, (float*)LB(l, w.o_pre2), nullptr, Sp, (size_t)M * H, 0, (float*)LB(l, w.o_st2), nullptr, (unsigned short*)LB(l, w.o_kb2),
                                       res2_st, res2_g, res2_b), "partials+dropout(ffn down)+residual+layernorm");
        } else
        if (ph) {
            if (int r_ = dense_f32(LB(l, w.o_h), I, y.w_out, I, y.b_out, nullptr, LB(l, w.o_pre2), H, "gemm(ffn down)")) return r_;
            const cpt::DropSpec sp = drop_spec(drop, 3 + 3 * l, false);
            TRY(cpt::layernorm_rows_ex((const float*)LB(l, w.o_pre2), y.ln2_g, y.ln2_b, d.ln_eps, x_out, xnext, dt, M, H, M, 0, 0, 0, s,
                                       res2, &sp, (float*)LB(l, w.o_pre2), nullptr, 1, 0, 0, (float*)LB(l, w.o_st2), nullptr, (unsigned short*)LB(l, w.o_kb2),
                                       res2_st, res2_g, res2_b), "dropout(ffn down)+residual+layernorm");
        } else {
        if (int r_ = dense_f32(LB(l, w.o_h), I, y.w_out, I, y.b_out, a_f32, LB(l, w.o_pre2), H, "gemm(ffn down)")) return r_;
        TRY(cpt::layernorm_rows_ex((const float*)LB(l, w.o_pre2), y.ln2_g, y.ln2_b, d.ln_eps, x_f32, xnext, dt, M, H, M, 0, 0, 0, s,
                                   nullptr, nullptr, nullptr, nullptr, 1, 0, 0, (float*)LB(l, w.o_st2)), "layernorm(ffn)");
        }
    }
    // head on the [MASK] rows
    void* rows = ws + w.rows;
    float* uh = (float*)(ws + w.uh);
    void* t2 = ws + w.t2;
    need(d.layers + 1);
    if (nsp) {
        // NSPCPT: pooled = tanh([CLS] W_pool^T + b_pool) (kept in `uh`), rel = pooled W_rel^T + b_rel, CE over n_rel classes
        if (!tail) TRY(cpt::gather_rows(ws + w.xout, dt, nullptr, rows, B, L, H, s), "gather([CLS])");
        TRY(gm(CPT_EPI_TANH, rows, H, m->w_pool, H, m->b_pool, nullptr, 0, uh, CPT_F32, H, B, H, H, s), "gemm(pooler)");
        const void* pin = uh;
        if (dt == CPT_BF16) {
            TRY(cpt::layernorm_rows(uh, nullptr, nullptr, 0.f, nullptr, t2, dt, B, H, B, 0, 0, s), "cast(pooled)");
            pin = t2;
        }
        TRY(gm(CPT_EPI_NONE, pin, H, m->w_rel, H, m->b_rel, nullptr, 0, o->rel, CPT_F32, d.n_rel, B, d.n_rel, H, s), "gemm(seq_relationship)");
        TRY(cpt::ce_rows(o->rel, b->labels, o->loss, (float*)(ws + w.dlogits), B, d.n_rel, s, ce_ticket, o->loss_mean, loss_ws), "ce_rows(rel)");
        return CPT_OK;
    }
    // Rh head rows: the [MASK] position of every sequence, or the n_rows labelled positions of a label grid (row_seq, mask_pos)
    const int Rh = b->n_rows > 0 ? b->n_rows : B;
    if (!tail) TRY(cpt::gather_rows(ws + w.xout, dt, b->mask_pos, rows, Rh, L, H, s, b->n_rows > 0 ? b->row_seq : nullptr, B), "gather([MASK])");
    TRY(gm(CPT_EPI_NONE, rows, H, m->w_tr, H, m->b_tr, nullptr, 0, uh, CPT_F32, H, Rh, H, H, s), "gemm(head transform)");
    TRY(cpt::layernorm_rows_ex(uh, m->tr_ln_g, m->tr_ln_b, d.ln_eps, dt == CPT_F32 ? (float*)t2 : nullptr,
                               dt == CPT_F32 ? nullptr : t2, dt, Rh, H, Rh, 0, 0, 1, s), "gelu+layernorm(head)");
    TRY(gm(CPT_EPI_NONE, t2, H, m->w_dec, H, m->b_dec, nullptr, 0, o->logits, CPT_F32, d.vocab, Rh, d.vocab, H, s), "gemm(decoder)");
    // (the launch's last workgroup leaves {sum, count} in the workspace for the backward and the mean in o->loss_mean: no memset, copy or divide launch)
    TRY(cpt::ce_rows(o->logits, b->labels, o->loss, (float*)(ws + w.dlogits), Rh, d.vocab, s, ce_ticket, o->loss_mean, loss_ws), "ce_rows");
    return CPT_OK;
}

int cpt_train_bwd(const cpt_model* m, const cpt_batch* b, const cpt_model_grads* g, float loss_scale,
                  void* workspace, size_t workspace_bytes, void* stream) {
    return cpt_train_bwd_ex(m, b, g, loss_scale, nullptr, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}

int cpt_train_bwd_ex(const cpt_model* m, const cpt_batch* b, const cpt_model_grads* g, float loss_scale, const float* loss_scale_dev,
                     void* workspace, size_t workspace_bytes, void* stream, cpt_bucket_fn grads_ready, void* user, const cpt_dropout* drop) {
    if (!m || !b || !g || !workspace) return abi_fail(CPT_ERR_NULL, "cpt_train_bwd: null argument");
    if (int rcd = check_drop(drop, "cpt_train_bwd")) return rcd;
    const bool ph = drop && drop->p_hidden > 0.f, pa = drop && drop->p_attn > 0.f;
    const cpt_dims& d = m->dims;
    const TrainLayout w = train_layout(d, b->B, b->Lt, b->Li, b->n_rows > 0 ? b->n_rows : b->B);
    int rc = check_common(m, b, workspace, workspace_bytes, w, "cpt_train_bwd");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool x3 = d.dtype == CPT_BF16X3_MASTERS;
    const int B = b->B, Lt = b->Lt, Li = b->Li, L = Lt + Li, M = B * L, H = d.hidden, I = d.inter, V = d.vocab, dt = x3 ? CPT_F32 : d.dtype;
    const int Mp = w.Mp, Bp = w.Bp, Vp = w.Vp, Rp = w.Rp, Dp = d.img_dim_pad;
    unsigned char* ws = (unsigned char*)workspace;
    CPT_TRAIN_GM;
    auto LB = [&](int l, size_t off) { return (void*)(ws + w.layer0 + (size_t)l * w.layer_stride + off); };
    float* dx = (float*)(ws + w.dx);
    float* dpre = (float*)(ws + w.dpre);
    float* da = (float*)(ws + w.da);
    void* dpre_lp = ws + w.dpre_lp;
    void* dctx = ws + w.dctx;
    void* dbig = ws + w.dbig;
    void* tA = ws + w.tA;
    void* tB = ws + w.tB;
    void* wT = ws + w.wT;
    // out[Nout][Kout] (fp32 gradient of a Linear weight) = dY^T . X over the M rows
    auto wgrad = [&](const void* dY, int dY_dt, int ldy, int Nout, const void* X, int ldx, int Kout, int rows, int rows_p,
                     float* out, int ldo, const char* what, cpt::ReduceJob* defer = nullptr) -> int {
        // bf16, tile-aligned shapes: the TN form of the GEMM reads dY and X as they are (transpose reads in LDS); tA serves as
        // its split-K partial buffer.  Everything else (fp32 mode, head-sized problems) goes through explicit transposes.
        // (rows_p = rows rounded up to a K-tile: the missing rows read as zero through the buffer bounds)
        if (g_wgrad_tn && dt == CPT_BF16 && dY_dt == CPT_BF16 && cpt::gemm_tn_eligible(Nout, Kout, rows_p, ldy, ldx, ldo)) {
            // (defer: the partial matrices go to their own buffer -- tA is reused right away -- and are added up by a later launch's spare workgroups)
            if (defer && w.qkvp_bytes) TRY(cpt::gemm_tn(dY, ldy, X, ldx, out, ldo, Nout, Kout, rows_p, ws + w.qkvp, w.qkvp_bytes, s, rows, defer), what);
            else
            TRY(cpt::gemm_tn(dY, ldy, X, ldx, out, ldo, Nout, Kout, rows_p, tA, w.tA_bytes, s, rows), what);
            return CPT_OK;
        }
        // bf16x3: the same TN kernel over row-stacked split copies of the two fp32 operands (contraction 3 rows_p: hi.hi + hi.lo + lo.hi)
        if (x3 && g_wgrad_tn && dY_dt == CPT_F32 && cpt::gemm_tn_eligible(Nout, Kout, 3 * rows_p, Nout, Kout, ldo) &&
            (size_t)6 * rows_p * Nout <= w.sA_bytes && (size_t)6 * rows_p * Kout <= w.sW_bytes) {
            TRY(cpt::split3_rows((const float*)dY, ldy, ws + w.sA, rows, rows_p, Nout, 0, s), what);
            TRY(cpt::split3_rows((const float*)X, ldx, ws + w.sW, rows, rows_p, Kout, 1, s), what);
            TRY(cpt::gemm_tn(ws + w.sA, Nout, ws + w.sW, Kout, out, ldo, Nout, Kout, 3 * rows_p, tA, w.tA_bytes, s, 3 * rows_p), what);
            return CPT_OK;
        }
        TRY(cpt::transpose_cast(dY, dY_dt, ldy, tA, dt, rows_p, rows, Nout, s), what);
        TRY(cpt::transpose_cast(X, dt, ldx, tB, dt, rows_p, rows, Kout, s), what);
        // (split-K with fp32 atomics was measured slower at B=32: 11.7-16.6 ms/step vs 10.4 -- see DESIGN.md)
        TRY(gm(CPT_EPI_NONE, tA, rows_p, tB, rows_p, nullptr, nullptr, 0, out, CPT_F32, ldo, Nout, Kout, rows_p, s), what);
        return CPT_OK;
    };
    // two weight gradients over the same rows in one launch (bf16, tile-aligned shapes); false = not applicable, the caller runs two wgrad
    auto wgrad_pair = [&](const void* dY0, int ldy0, int Nout0, const void* X0, int ldx0, int Kout0, float* out0,
                          const void* dY1, int ldy1, int Nout1, const void* X1, int ldx1, int Kout1, float* out1, int& rc_out, const char* what) -> bool {
        rc_out = CPT_OK;
        if (!(g_wgrad_tn && g_wgrad_pair && dt == CPT_BF16)) return false;
        const int r = cpt::gemm_tn_pair(dY0, ldy0, X0, ldx0, out0, Nout0, Kout0, dY1, ldy1, X1, ldx1, out1, Nout1, Kout1, Mp, M, tA, w.tA_bytes, s);
        if (r == CPT_ERR_SHAPE) return false;
        rc_out = abi_check(r, what);
        return true;
    };
    // out[rows][Kout] = dY[rows][Nout] . Wt[Nout][Kout]  (+ resid), Wt given as the Linear weight [Nout][Kout]
    constexpr int NOT_FUSED = 1 << 20;       // dgrad(..., gelu_u): the shape has no fused GELU-gradient epilogue
    // colsum_rows > 0: gelu_colsum is a table of partial rows (gemm_nn); S_out: a K-split launch leaves its *S_out partial matrices in tA for the
    // LayerNorm backward behind it, which adds them and the residual itself (*S_out = 1: `out` is complete)
    auto dgrad = [&](const void* dY, int ldy, int Nout_p, const void* Wt, int ldw, int Nout, int Kout, int rows,
                     const float* resid, void* out, int out_dt, const char* what, const void* gelu_u = nullptr, float* gelu_colsum = nullptr,
                     int colsum_rows = 0, int* S_out = nullptr, int split2_bf16 = 0) -> int {
        if (S_out) *S_out = 1;
        // bf16, tile-aligned: the NN form reads the weight as stored (transpose reads in LDS)
        // (Nout < Nout_p: the K-tile padding columns of dY are zero and the weight rows beyond Nout read as zero)
        if (g_wgrad_tn && dt == CPT_BF16 && cpt::gemm_nn_eligible(rows, Kout, Nout_p, ldy, ldw)) {
            TRY(cpt::gemm_nn(dY, ldy, Wt, ldw, resid, Kout, out, out_dt, Kout, rows, Kout, Nout_p, s, Nout, tA, w.tA_bytes, gelu_u, Kout, gelu_colsum, colsum_rows, S_out, split2_bf16), what);
            return CPT_OK;
        }
        if (gelu_u) return NOT_FUSED;   // caller runs the unfused pair (dgrad, then gelu_bwd)
        // bf16x3: the same NN kernel on split operands -- dY as [rows][hi | hi | lo], the weight AS STORED split into three row blocks
        // (hi; lo; hi): contraction 3 Nout_p, no transposed weight copy
        if (x3 && g_wgrad_tn && out_dt == CPT_F32 && Nout_p % 4 == 0 && cpt::gemm_nn_eligible(rows, Kout, 3 * Nout_p, 3 * Nout_p, Kout) &&
            (size_t)6 * rows * Nout_p <= w.sA_bytes && (size_t)6 * Nout_p * Kout <= w.sW_bytes) {
            TRY(cpt::split3((const float*)dY, ldy, ws + w.sA, rows, Nout_p, 0, s), what);
            TRY(cpt::split3_rows((const float*)Wt, ldw, ws + w.sW, Nout, Nout_p, Kout, 1, s), what);
            TRY(cpt::gemm_nn(ws + w.sA, 3 * Nout_p, ws + w.sW, Kout, resid, Kout, out, CPT_F32, Kout, rows, Kout, 3 * Nout_p, s, 3 * Nout_p, tA, w.tA_bytes,
                             nullptr, 0, nullptr), what);
            return CPT_OK;
        }
        TRY(cpt::transpose_cast(Wt, dt, ldw, wT, dt, Nout_p, Nout, Kout, s), what);
        TRY(gm(resid ? CPT_EPI_RESID : CPT_EPI_NONE, dY, ldy, wT, Nout_p, nullptr, resid, Kout, out, out_dt, Kout, rows, Kout, Nout_p, s), what);
        return CPT_OK;
    };

    // ---- head ------------------------------------------------------------------------------------
    void* dl_lp = ws + w.dl_lp;
    float* dt2 = (float*)(ws + w.dt2);
    float* duh = (float*)(ws + w.duh);
    void* duh_lp = ws + w.duh_lp;
    float* drows = (float*)(ws + w.drows);
    const bool nsp = !m->w_tr && !m->w_dec;
    const bool tail = train_tail_on(d, b) && !(b->mask_3d && b->attn_mask);      // the forward's pruned last layer (same predicate)
    const int Rh = (!nsp && b->n_rows > 0) ? b->n_rows : B;      // head rows (see cpt_train_fwd)
    if (nsp) {
        if (!g->w_pool || !g->b_pool || !g->w_rel || !g->b_rel) return abi_fail(CPT_ERR_NULL, "cpt_train_bwd: NSP head needs the w_pool / b_pool / w_rel / b_rel gradient tensors");
        const int NR = d.n_rel, NRp = 64;
        const float* pooled = (const float*)(ws + w.uh);
        const void* pin = dt == CPT_BF16 ? (const void*)(ws + w.t2) : (const void*)pooled;
        TRY(cpt::scale_cast((const float*)(ws + w.dlogits), (const float*)(ws + w.loss), loss_scale, loss_scale_dev, dl_lp, dt, B, NR, NRp, s, g->b_rel), "scale(drel) + colsum(cls.bias)");
        rc = dgrad(dl_lp, NRp, NRp, m->w_rel, H, NR, H, B, nullptr, dt2, CPT_F32, "dgrad(seq_relationship)");
        if (rc) return rc;
        rc = wgrad(dl_lp, dt, NRp, NR, pin, H, H, B, Bp, g->w_rel, H, "wgrad(seq_relationship)");
        if (rc) return rc;
        TRY(cpt::tanh_bwd(dt2, pooled, duh, dt == CPT_BF16 ? duh_lp : nullptr, dt, (size_t)B * H, s), "tanh_bwd(pooler)");
        const void* dpo = dt == CPT_BF16 ? (const void*)duh_lp : (const void*)duh;
        TRY(cpt::colsum(duh, CPT_F32, H, g->b_pool, B, H, s), "colsum(pooler bias)");
        rc = wgrad(dpo, dt, H, H, ws + w.rows, H, H, B, Bp, g->w_pool, H, "wgrad(pooler)");
        if (rc) return rc;
        rc = dgrad(dpo, H, H, m->w_pool, H, H, H, B, nullptr, drows, CPT_F32, "dgrad(pooler)");
        if (rc) return rc;
    } else {
    TRY(cpt::scale_cast((const float*)(ws + w.dlogits), (const float*)(ws + w.loss), loss_scale, loss_scale_dev, dl_lp, dt, Rh, V, Vp, s, g->b_dec), "scale(dlogits) + colsum(cls.bias)");
    rc = dgrad(dl_lp, Vp, Vp, m->w_dec, H, V, H, Rh, nullptr, dt2, CPT_F32, "dgrad(decoder)");
    if (rc) return rc;
    rc = wgrad(dl_lp, dt, Vp, V, ws + w.t2, H, H, Rh, Bp, g->word_emb, H, "wgrad(decoder)");
    if (rc) return rc;
    TRY(cpt::ln_bwd(dt2, (const float*)(ws + w.uh), m->tr_ln_g, d.ln_eps, duh, dt == CPT_BF16 ? duh_lp : nullptr, dt, g->tr_ln_g,
                    g->tr_ln_b, Rh, H, Rh, 0, 0, 1, s, nullptr, 0, nullptr, g->b_tr), "ln_bwd(head) + transform bias sums");      // (round 6: the bias gradient = column sums of this launch's output, summed inside it)
    const void* duh_in = dt == CPT_BF16 ? duh_lp : (const void*)duh;
    rc = wgrad(duh_in, dt, H, H, ws + w.rows, H, H, Rh, Bp, g->w_tr, H, "wgrad(head transform)");
    if (rc) return rc;
    rc = dgrad(duh_in, H, H, m->w_tr, H, H, H, Rh, nullptr, drows, CPT_F32, "dgrad(head transform)");
    if (rc) return rc;
    }
    if (!tail) {      // (pruned last layer: drows IS the gradient of that layer's compact output)
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)M * H * 4, s);
    if (e != hipSuccess) return abi_fail(CPT_ERR_HIP - (int)e, "zero dx: %s", hipGetErrorString(e));
    TRY(cpt::scatter_rows_add(drows, nsp ? nullptr : b->mask_pos, dx, Rh, L, H, s, b->n_rows > 0 ? b->row_seq : nullptr, B), "scatter([MASK] / [CLS] rows)");
    }
    // gradient buckets: the host callback runs right AFTER the last launch that writes the bucket's gradients has
    // been enqueued on `stream` (the tied word-embedding table belongs to bucket 0: its lookup gradient comes last)
    auto ready = [&](int bucket) { if (grads_ready) grads_ready(user, bucket); };
    ready(d.layers + 1);

    // ---- encoder layers, last to first -----------------------------------------------------------
    // with hidden dropout the gradient that enters a dense layer is the masked one (dmask); the unmasked one (dpre)
    // keeps feeding the residual path
    float* dmask = (float*)(ws + w.dmask);
    void* dmask_lp = ws + w.dmask_lp;
    const void* dpre_in = ph ? (dt == CPT_BF16 ? (const void*)dmask_lp : (const void*)dmask) : (dt == CPT_BF16 ? (const void*)dpre_lp : (const void*)dpre);
    const float* dpre_f = ph ? dmask : dpre;
    cpt::ColJobs pend = {};      // round 6: column-sum jobs waiting for a carrier launch (kernels.h)
    // round 6: the Q|K|V weight gradient's K-split partial matrices of layer l + 1 wait for the three-problem weight-gradient launch of layer l, whose tiles
    // leave 40 of 256 CUs idle at hidden 768 (cpt_set_tuning key 37); the bucket of layer l + 1 is announced behind that launch
    cpt::ReduceJob qjob = {};
    int ready_after_triple = -1;
    const bool qkv_defer = g_qkv_defer && g_wgrad_tn && g_wgrad_pair >= 2 && dt == CPT_BF16 && cpt::gemm_tn_triple_eligible(H, I, I, H, H, H, Mp);
    int lnp_turn = 0, dxS = 1, dxB = 0;      // dxB: those partial matrices are bf16 (two of them, gemm_nn split2_bf16)
      // dxS > 1: the gradient entering the layer lies in tA as that many K-split partial matrices (+ residual dpre)
    for (int l = d.layers - 1; l >= 0; --l) {
        const cpt_layer& y = m->layers[l];
        const cpt_layer_grads& gy = g->layers[l];
        const bool fuse_db = dt == CPT_BF16;
        bool triple = false;
        if (tail && l == d.layers - 1) {
            // ---- round 6: the pruned last layer -- BertOutput, BertIntermediate, BertSelfOutput backward on the B head rows; the weight gradients
            // contract over B rows (one K-tile), the data gradient re-enters the full tensors at the head rows (everything else is zero) ----
            const cpt::RowMap rm = {nsp ? nullptr : b->mask_pos, L, 1};
            const cpt::DropSpec sp2 = drop_spec(drop, 3 + 3 * l, false), sp1 = drop_spec(drop, 2 + 3 * l, false);
            float* dpre_r = (float*)(ws + w.t_dpre); float* da_r = (float*)(ws + w.t_da);
            void* dlpF = ws + w.t_dlpF; void* dlpA = ws + w.t_dlpA; void* dbig_r = ws + w.t_dbig; void* dctx_r = ws + w.t_dctx;
            cpt::LnBwdExtra e2 = {};
            e2.stats = (const float*)(ws + w.t_st2); e2.drop_rows = rm; e2.keep_bits = (const unsigned short*)(ws + w.t_kb2);
            TRY(cpt::ln_bwd(drows, (const float*)(ws + w.t_pre2), y.ln2_g, d.ln_eps, dpre_r, dlpF, dt, gy.ln2_g, gy.ln2_b, B, H, B, 0, 0, 0, s, nullptr, 0,
                            ph ? &sp2 : nullptr, gy.b_out, &e2), "ln_bwd(ffn, head rows)");
            rc = dgrad(dlpF, H, H, y.w_out, I, H, I, B, nullptr, dbig_r, dt, "dgrad(ffn down)+gelu_bwd+bias (head rows)", ws + w.t_u, gy.b_in);
            if (rc) return rc == NOT_FUSED ? abi_fail(CPT_ERR_SHAPE, "cpt_train_bwd: pruned last layer without the fused GELU-gradient GEMM") : rc;
            int daS = 1;
            rc = dgrad(dbig_r, I, I, y.w_in, H, I, H, B, dpre_r, da_r, CPT_F32, "dgrad(ffn up)+residual (head rows)", nullptr, nullptr, 0, &daS);
            if (rc) return rc;
            cpt::LnBwdExtra e1 = {};
            e1.stats = (const float*)(ws + w.t_st1); e1.drop_rows = rm; e1.keep_bits = (const unsigned short*)(ws + w.t_kb1);
            if (daS > 1) { e1.dy_parts = daS; e1.dy_stride = (size_t)B * H; e1.dy_resid = dpre_r; }
            TRY(cpt::ln_bwd(daS > 1 ? (const float*)tA : da_r, (const float*)(ws + w.t_pre1), y.ln1_g, d.ln_eps, dpre_r, dlpA, dt, gy.ln1_g, gy.ln1_b, B, H, B, 0, 0, 0, s,
                            nullptr, 0, ph ? &sp1 : nullptr, gy.b_ao, &e1), "ln_bwd(attn, head rows)");
            const int Kp = up64(B);
            const int r3 = cpt::gemm_tn_triple(dlpF, H, ws + w.t_h, I, gy.w_out, H, I, dbig_r, I, ws + w.t_a, H, gy.w_in, I, H,
                                               dlpA, H, ws + w.t_ctx, H, gy.w_ao, H, H, Kp, B, s);
            if (r3 == CPT_ERR_SHAPE) {
                rc = wgrad(dlpF, dt, H, H, ws + w.t_h, I, I, B, Kp, gy.w_out, I, "wgrad(ffn down, head rows)");
                if (rc) return rc;
                rc = wgrad(dbig_r, dt, I, I, ws + w.t_a, H, H, B, Kp, gy.w_in, H, "wgrad(ffn up, head rows)");
                if (rc) return rc;
                rc = wgrad(dlpA, dt, H, H, ws + w.t_ctx, H, H, B, Kp, gy.w_ao, H, "wgrad(attn out, head rows)");
                if (rc) return rc;
            } else TRY(r3, "wgrad(ffn down | ffn up | attn out, head rows)");
            rc = dgrad(dlpA, H, H, y.w_ao, H, H, H, B, nullptr, dctx_r, dt, "dgrad(attn out, head rows)");
            if (rc) return rc;
            TRY(cpt::tail_scatter2(dctx_r, dpre_r, rm.pos, dctx, dpre, B, L, H, s), "scatter(head rows of dctx, dpre)");
            triple = true;      // (below: Q|K|V's weight gradient alone)
        } else {
        // x_out = LN2(pre2); pre2 = h W_out^T + b_out + a
        // bf16: the LayerNorm backward also applies the dense output's dropout mask to the gradient that goes on into the dense
        // layer (written as bf16 only) and sums its columns for the bias gradient: no dropout_rows / colsum launches
        if (fuse_db) {
            // round 6: the forward's row statistics; the incoming gradient possibly as the K-split partial matrices of the previous layer's
            // dgrad(qkv) (+ its residual dpre -- the row this launch overwrites with its own output only after reading it); the column sums of this
            // launch stay behind as partial rows and are added up by spare workgroups of the NEXT LayerNorm backward (pend), as this one adds the last one's
            const cpt::DropSpec sp2 = drop_spec(drop, 3 + 3 * l, false);
            cpt::LnBwdExtra e = {};
            e.stats = (const float*)LB(l, w.o_st2); e.jobs = &pend; e.defer_reduce = 1; e.keep_bits = (const unsigned short*)LB(l, w.o_kb2);
            if (dxS > 1) { e.dy_parts = dxS; e.dy_stride = (size_t)M * H; e.dy_resid = dpre; e.dy_parts_bf16 = dxB; }
            float* part = (float*)(ws + w.lnp[lnp_turn]);
            TRY(cpt::ln_bwd(dxS > 1 ? (const float*)tA : dx, (const float*)LB(l, w.o_pre2), y.ln2_g, d.ln_eps, dpre, ph ? dmask_lp : dpre_lp, dt, gy.ln2_g, gy.ln2_b,
                            M, H, M, 0, 0, 0, s, part, w.lnp_bytes, ph ? &sp2 : nullptr, gy.b_out, &e), "ln_bwd(ffn)+dropout+bias");
            pend = cpt::ColJobs{};
            cpt::col_jobs_add(pend, part, cpt::ln_bwd_part_rows(M), 3 * H, 3 * H, H, gy.ln2_g, gy.ln2_b, gy.b_out);
            lnp_turn ^= 1;
            if (l + 1 < d.layers) {      // the layer above is complete now: this launch added its attention-side LayerNorm's sums
                if (qjob.S > 1) ready_after_triple = 1 + l + 1;      // (... but for its Q|K|V weight gradient, still in partial matrices)
                else ready(1 + l + 1);
            }
        } else {
        TRY(cpt::ln_bwd(dx, (const float*)LB(l, w.o_pre2), y.ln2_g, d.ln_eps, dpre, dt == CPT_BF16 ? dpre_lp : nullptr, dt, gy.ln2_g, gy.ln2_b,
                        M, H, M, 0, 0, 0, s, (float*)tB, w.tB_bytes), "ln_bwd(ffn)");
        if (ph) TRY(cpt::dropout_rows(dpre, nullptr, dmask, dt == CPT_BF16 ? dmask_lp : nullptr, dt, M, H, drop_spec(drop, 3 + 3 * l, false), s), "dropout_bwd(ffn down)");
        TRY(cpt::colsum(dpre_f, CPT_F32, H, gy.b_out, M, H, s), "colsum(b_out)");
        }
        // (weight gradients: the two FFN matrices share one launch behind the GELU backward, attention output and Q|K|V one behind
        // the attention backward -- their operands stay untouched until then; shapes the pair kernel does not take run one by one)
        // h = gelu(u); u = a W_in^T + b_in: bf16 runs the GELU backward in the epilogue of the data-gradient GEMM
        // ... and sums its columns into the bias gradient (round 3: no colsum launch over the M x I tensor)
        // round 6: ... as PARTIAL rows (one per 32-row wave block, plain stores -- the end-of-launch atomics of round 3's form cost more than the
        // colsum launch), added up by the attention-side LayerNorm backward's spare workgroups (bit 0 of key 18 cleared: the colsum launch)
        const bool bin_parts = (g_bias_fuse & 1) != 0;
        rc = dt == CPT_BF16 ? dgrad(dpre_in, H, H, y.w_out, I, H, I, M, nullptr, dbig, dt, "dgrad(ffn down)+gelu_bwd+bias", LB(l, w.o_u),
                                    bin_parts ? (float*)(ws + w.csp) : nullptr, w.csp_rows) : NOT_FUSED;
        if (rc == NOT_FUSED) {
            rc = dgrad(dpre_in, H, H, y.w_out, I, H, I, M, nullptr, dbig, dt, "dgrad(ffn down)");
            if (rc) return rc;
            TRY(cpt::gelu_bwd(dbig, LB(l, w.o_u), dbig, dt, (size_t)M * I, s), "gelu_bwd");
            TRY(cpt::colsum(dbig, dt, I, gy.b_in, M, I, s), "colsum(b_in)");
        } else if (rc) return rc;
        else if (!bin_parts) TRY(cpt::colsum(dbig, dt, I, gy.b_in, M, I, s), "colsum(b_in)");
        else cpt::col_jobs_add(pend, (const float*)(ws + w.csp), (M + 31) / 32, I, I, I, gy.b_in);
        // round 3: where the three problems' tiles fit one round (hidden 768: 96 + 96 + 24), the two FFN weight gradients wait for the attention
        // output's and share ONE launch with it behind the attention-side LayerNorm backward (which then writes its low-precision output
        // to a second buffer: the FFN-side one is still an operand); Q|K|V runs alone behind the attention backward
        triple = g_wgrad_tn && g_wgrad_pair >= 2 && dt == CPT_BF16 && cpt::gemm_tn_triple_eligible(H, I, I, H, H, H, Mp);
        if (triple) {
        } else
        if (wgrad_pair(dpre_in, H, H, LB(l, w.o_h), I, I, gy.w_out, dbig, I, I, LB(l, w.o_a), H, H, gy.w_in, rc, "wgrad(ffn down | ffn up)")) {
            if (rc) return rc;
        } else {
            rc = wgrad(dpre_in, dt, H, H, LB(l, w.o_h), I, I, M, Mp, gy.w_out, I, "wgrad(ffn down)");
            if (rc) return rc;
            rc = wgrad(dbig, dt, I, I, LB(l, w.o_a), H, H, M, Mp, gy.w_in, H, "wgrad(ffn up)");
            if (rc) return rc;
        }
        int daS = 1;
        rc = dgrad(dbig, I, I, y.w_in, H, I, H, M, dpre, da, CPT_F32, "dgrad(ffn up)+residual", nullptr, nullptr, 0, fuse_db ? &daS : nullptr, 1);
        const int daB = (daS == 2 && M >= 2048) ? 1 : 0;      // (two partial matrices at 2048..6144 rows: the bf16 pair of gemm_nn's split2_bf16; at few rows K splits >= 4 ways in fp32)
        if (rc) return rc;
        // a = LN1(pre1); pre1 = ctx W_ao^T + b_ao + x_in
        if (fuse_db) {
            const cpt::DropSpec sp1 = drop_spec(drop, 2 + 3 * l, false);
            cpt::LnBwdExtra e = {};
            e.stats = (const float*)LB(l, w.o_st1); e.jobs = &pend; e.defer_reduce = 1; e.keep_bits = (const unsigned short*)LB(l, w.o_kb1);
            if (daS > 1) { e.dy_parts = daS; e.dy_stride = (size_t)M * H; e.dy_resid = dpre; e.dy_parts_bf16 = daB; }
            float* part = (float*)(ws + w.lnp[lnp_turn]);
            TRY(cpt::ln_bwd(daS > 1 ? (const float*)tA : da, (const float*)LB(l, w.o_pre1), y.ln1_g, d.ln_eps, dpre, triple ? (void*)(ws + w.dlp2) : (ph ? dmask_lp : dpre_lp), dt, gy.ln1_g, gy.ln1_b,
                            M, H, M, 0, 0, 0, s, part, w.lnp_bytes, ph ? &sp1 : nullptr, gy.b_ao, &e), "ln_bwd(attn)+dropout+bias");
            pend = cpt::ColJobs{};
            cpt::col_jobs_add(pend, part, cpt::ln_bwd_part_rows(M), 3 * H, 3 * H, H, gy.ln1_g, gy.ln1_b, gy.b_ao);
            lnp_turn ^= 1;
        } else {
        TRY(cpt::ln_bwd(da, (const float*)LB(l, w.o_pre1), y.ln1_g, d.ln_eps, dpre, dt == CPT_BF16 ? dpre_lp : nullptr, dt, gy.ln1_g, gy.ln1_b,
                        M, H, M, 0, 0, 0, s, (float*)tB, w.tB_bytes), "ln_bwd(attn)");
        if (ph) TRY(cpt::dropout_rows(dpre, nullptr, dmask, dt == CPT_BF16 ? dmask_lp : nullptr, dt, M, H, drop_spec(drop, 2 + 3 * l, false), s), "dropout_bwd(attn out)");
        TRY(cpt::colsum(dpre_f, CPT_F32, H, gy.b_ao, M, H, s), "colsum(b_ao)");
        }
        const void* dao_in = triple ? (const void*)(ws + w.dlp2) : dpre_in;      // gradient entering the attention output's dense layer
        if (triple) {
            TRY(cpt::gemm_tn_triple(dpre_in, H, LB(l, w.o_h), I, gy.w_out, H, I, dbig, I, LB(l, w.o_a), H, gy.w_in, I, H,
                                    dao_in, H, LB(l, w.o_ctx), H, gy.w_ao, H, H, Mp, M, s, &qjob), "wgrad(ffn down | ffn up | attn out) + partial sums of the layer above's wgrad(qkv)");
            qjob = cpt::ReduceJob{};
        } else if (qjob.S > 1) {
            TRY(cpt::reduce_job_flush(qjob, s), "partial sums of wgrad(qkv)");
            qjob = cpt::ReduceJob{};
        }
        if (ready_after_triple >= 0) { ready(ready_after_triple); ready_after_triple = -1; }
        rc = dgrad(dao_in, H, H, y.w_ao, H, H, H, M, nullptr, dctx, dt, "dgrad(attn out)");
        if (rc) return rc;
        }
        const cpt::DropSpec da_spec = drop_spec(drop, 1 + 3 * l, true);
        if (train_x3_attention(d, L, (b->mask_3d && b->attn_mask) ? 1 : 0))
            TRY(cpt::attention_bwd_x3((const float*)LB(l, w.o_qkv), b->attn_mask, (const float*)dctx, (float*)dbig, B, L, d.heads, s, pa ? &da_spec : nullptr,
                                      (g_bias_fuse & 2) ? gy.b_qkv : nullptr), "attention_bwd (split operands)");
        else
        TRY(cpt::attention_bwd(dt, LB(l, w.o_qkv), b->attn_mask, dctx, dbig, B, L, d.heads, s, pa ? &da_spec : nullptr, (g_bias_fuse & 2) ? gy.b_qkv : nullptr, (b->mask_3d && b->attn_mask) ? 1 : 0,
                               (dt == CPT_BF16 && !(b->mask_3d && b->attn_mask)) ? LB(l, w.o_ctx) : nullptr, (dt == CPT_BF16 && !(b->mask_3d && b->attn_mask)) ? (const float*)LB(l, w.o_ast) : nullptr), "attention_bwd+bias");
        if (!(g_bias_fuse & 2)) TRY(cpt::colsum(dbig, dt, 3 * H, gy.b_qkv, M, 3 * H, s), "colsum(b_qkv)");
        if (triple) {
            rc = wgrad(dbig, dt, 3 * H, 3 * H, LB(l, w.o_xin), H, H, M, Mp, gy.w_qkv, H, "wgrad(qkv)", (qkv_defer && fuse_db) ? &qjob : nullptr);
            if (rc) return rc;
        } else
        if (wgrad_pair(dpre_in, H, H, LB(l, w.o_ctx), H, H, gy.w_ao, dbig, 3 * H, 3 * H, LB(l, w.o_xin), H, H, gy.w_qkv, rc, "wgrad(attn out | qkv)")) {
            if (rc) return rc;
        } else {
            rc = wgrad(dpre_in, dt, H, H, LB(l, w.o_ctx), H, H, M, Mp, gy.w_ao, H, "wgrad(attn out)");
            if (rc) return rc;
            rc = wgrad(dbig, dt, 3 * H, 3 * H, LB(l, w.o_xin), H, H, M, Mp, gy.w_qkv, H, "wgrad(qkv)");
            if (rc) return rc;
        }
        // (layer 0's result feeds the embedding passes: complete; above that a K-split launch leaves its partial matrices to the next LayerNorm backward)
        dxS = 1;
        rc = dgrad(dbig, 3 * H, 3 * H, y.w_qkv, H, 3 * H, H, M, dpre, dx, CPT_F32, "dgrad(qkv)+residual", nullptr, nullptr, 0, (fuse_db && l > 0) ? &dxS : nullptr, 1);
        dxB = (dxS == 2 && M >= 2048) ? 1 : 0;
        if (rc) return rc;
        if (!fuse_db) ready(1 + l);
    }
    if (dt == CPT_BF16) {      // the last attention-side LayerNorm's sums (and whatever else still waits for a carrier launch)
        if (qjob.S > 1) { TRY(cpt::reduce_job_flush(qjob, s), "partial sums of wgrad(qkv), first layer"); qjob = cpt::ReduceJob{}; }
        if (Li <= 0) {      // (with regions the jobs ride on the region LayerNorm's backward below: no launch of their own)
        TRY(cpt::col_jobs_flush(pend, s), "column-sum jobs");
        pend = cpt::ColJobs{};
        ready(1);
        }
    }

    // ---- region projection and text embeddings ----------------------------------------------------
    // (round 6: the embedding dropout's backward is applied by the two launches that read dx -- the region LayerNorm's backward and the embedding backward -- as
    // they load their rows: no dropout pass over all rows in front of them)
    const cpt::DropSpec edrop = drop_spec(drop, 0, false);
    if (Li > 0) {
        const int R = B * Li;
        float* dimg = (float*)(ws + w.dimg);
        void* dimg_lp = ws + w.dimg_lp;
        const bool iln = d.use_img_ln && m->img_ln_g;
        cpt::LnBwdExtra ei = {};
        if (ph) ei.in_drop = edrop;
        if (dt == CPT_BF16) ei.jobs = &pend;      // the first layer's attention-side column sums (and whatever else still waits): spare workgroups of this launch
        TRY(cpt::ln_bwd(dx, (const float*)(ws + w.imgpre), iln ? m->img_ln_g : nullptr, d.img_ln_eps, dimg, dt == CPT_BF16 ? dimg_lp : nullptr, dt,
                        iln ? g->img_ln_g : nullptr, iln ? g->img_ln_b : nullptr, R, H, Li, L, Lt, 0, s, nullptr, 0, nullptr, iln ? g->b_img : nullptr, &ei), "ln_bwd(img) + b_img sums + column-sum jobs");
        if (dt == CPT_BF16) { pend = cpt::ColJobs{}; ready(1); }
        if (!iln) TRY(cpt::colsum(dimg, CPT_F32, H, g->b_img, R, H, s), "colsum(b_img)");      // (no region LayerNorm: the launch above is a plain row gather)
        float* gimg = (float*)(ws + w.gimg);      // (written whole by the weight-gradient GEMM: nothing to clear)
        rc = wgrad(dt == CPT_BF16 ? dimg_lp : (const void*)dimg, dt, H, H, ws + w.imgp, Dp, Dp, R, Rp, gimg, Dp, "wgrad(img_embedding)");
        if (rc) return rc;
        TRY(cpt::unpad_add(gimg, g->w_img, H, d.img_dim, Dp, s), "unpad(img weight grad)");
    }
    TRY(cpt::embed_bwd(dx, b->input_ids, b->token_type, b->position_ids, m->word_emb, m->pos_emb, m->type_emb, m->emb_ln_g,
                       d.ln_eps, g->word_emb, g->pos_emb, g->type_emb, g->emb_ln_g, g->emb_ln_b, B, Lt, L, H, d.vocab, d.max_pos,
                       d.type_vocab, s, ph ? &edrop : nullptr), "embed_bwd + dropout_bwd(embeddings)");
    ready(0);
    return CPT_OK;
}

// What cpt_train_bwd needs cleared, in one launch (+ one memset for a table the chosen head does not overwrite): Linear weight
// gradients are WRITTEN by their GEMMs (and the tied word-embedding table by the decoder's), so only the vectors and small tables
// that take atomic adds are cleared -- 0.3 MB instead of the 447 MB of a whole-buffer fill per step.
int cpt_train_zero_grads(const cpt_model* m, const cpt_model_grads* g, int Li, void* stream) {
    if (!m || !g || !g->layers) return abi_fail(CPT_ERR_NULL, "cpt_train_zero_grads: null argument");
    const cpt_dims& d = m->dims;
    hipStream_t s = (hipStream_t)stream;
    const bool nsp = !m->w_tr && !m->w_dec;
    const size_t H = d.hidden, I = d.inter;
    cpt::ZeroSegs z;
    z.count = 0;
    bool ovf = false;
    auto seg = [&](float* p, size_t n) {
        if (!p || !n) return;
        if (z.count >= cpt::ZS_MAX) { ovf = true; return; }
        z.p[z.count] = p; z.n[z.count] = (unsigned)n; ++z.count;
    };
    auto big = [&](float* p, size_t n) -> int {
        if (!p || !n) return CPT_OK;
        hipError_t e = hipMemsetAsync(p, 0, n * 4, s);
        return e == hipSuccess ? CPT_OK : abi_fail(CPT_ERR_HIP - (int)e, "cpt_train_zero_grads: %s", hipGetErrorString(e));
    };
    seg(g->pos_emb, (size_t)d.max_pos * H); seg(g->type_emb, (size_t)d.type_vocab * H);
    seg(g->emb_ln_g, H); seg(g->emb_ln_b, H); seg(g->b_img, H); seg(g->img_ln_g, H); seg(g->img_ln_b, H);
    for (int l = 0; l < d.layers; ++l) {
        const cpt_layer_grads& y = g->layers[l];
        seg(y.b_qkv, 3 * H); seg(y.b_ao, H); seg(y.ln1_g, H); seg(y.ln1_b, H);
        seg(y.b_in, I); seg(y.b_out, H); seg(y.ln2_g, H); seg(y.ln2_b, H);
    }
    seg(g->b_tr, H); seg(g->tr_ln_g, H); seg(g->tr_ln_b, H); seg(g->b_dec, (size_t)d.vocab);
    seg(g->b_pool, H); seg(g->b_rel, (size_t)d.n_rel);
    if (ovf) return abi_fail(CPT_ERR_SHAPE, "cpt_train_zero_grads: more than %d gradient vectors (layers = %d)", cpt::ZS_MAX, d.layers);
    TRY(cpt::zero_segments(z, s), "zero_segments");
    int rc;
    if (nsp) {      // no decoder GEMM writes the word table: the lookup gradient adds into it
        if ((rc = big(g->word_emb, (size_t)d.vocab * H))) return rc;
        if ((rc = big(g->w_tr, H * H))) return rc;
    }
    // (the MLM path leaves the pooler / relation head without a gradient: nothing writes those regions of the caller's gradient buffer and the
    // optimizer skips them (code 0) -- round 6: no per-step fill of them either)
    if (Li <= 0 && (rc = big(g->w_img, H * (size_t)d.img_dim))) return rc;      // no regions: the projection gets no gradient
    return CPT_OK;
}

int cpt_dropout_mask(const cpt_dropout* drop, int site, int is_attn, unsigned char* out, int n0, int n1, int n2, void* stream) {
    if (!drop || !out) return abi_fail(CPT_ERR_NULL, "cpt_dropout_mask: null argument");
    if (int rcd = check_drop(drop, "cpt_dropout_mask")) return rcd;
    const cpt::DropSpec sp = drop_spec(drop, site, is_attn != 0);
    if (sp.thresh == 0) return abi_fail(CPT_ERR_SHAPE, "cpt_dropout_mask: the dropout probability of this site is 0");
    return abi_check(cpt::dropout_mask(is_attn ? 1 : 0, out, n0, n1, n2, sp, (hipStream_t)stream), "cpt_dropout_mask");
}

// ---- operator-level backward entry points (ABI 6): thin wrappers over the kernels cpt_train_bwd launches, for per-operator parity tests ----
int cpt_attention_bwd(int dtype, const void* qkv, const int64_t* attn_mask, int mask_3d, const void* dctx, void* dqkv, float* dbias_qkv,
                      int B, int L, int heads, const cpt_dropout* drop, int site, void* stream) {
    if (!qkv || !dctx || !dqkv) return abi_fail(CPT_ERR_NULL, "cpt_attention_bwd: null argument");
    if (B <= 0 || L <= 0 || heads <= 0) return abi_fail(CPT_ERR_SHAPE, "cpt_attention_bwd: bad shape B=%d L=%d heads=%d", B, L, heads);
    if (int rcd = check_drop(drop, "cpt_attention_bwd")) return rcd;
    const bool pa = drop && drop->p_attn > 0.f;
    const cpt::DropSpec sp = drop_spec(drop, site, true);
    const int m3 = (mask_3d && attn_mask) ? 1 : 0;
    if (dtype == CPT_BF16X3 || dtype == CPT_BF16X3_MASTERS) {
        if (!cpt::attention_bwd_x3_supported(L, m3))
            return abi_fail(CPT_ERR_SHAPE, "cpt_attention_bwd: no split-operand attention backward at L = %d%s", L, m3 ? " with a 3-D mask" : "");
        return abi_check(cpt::attention_bwd_x3((const float*)qkv, attn_mask, (const float*)dctx, (float*)dqkv, B, L, heads, (hipStream_t)stream,
                                               pa ? &sp : nullptr, dbias_qkv), "cpt_attention_bwd (split operands)");
    }
    if (dtype != CPT_F32 && dtype != CPT_BF16) return abi_fail(CPT_ERR_DTYPE, "cpt_attention_bwd: dtype %d", dtype);
    if (!cpt::attention_bwd_supported(dtype, L, pa ? 1 : 0, m3))
        return abi_fail(CPT_ERR_SHAPE, "cpt_attention_bwd: no attention backward for dtype %d at L = %d%s", dtype, L, pa ? " with attention dropout" : "");
    return abi_check(cpt::attention_bwd(dtype, qkv, attn_mask, dctx, dqkv, B, L, heads, (hipStream_t)stream, pa ? &sp : nullptr, dbias_qkv, m3), "cpt_attention_bwd");
}

int cpt_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx, void* dx_lp, int lp_dtype, float* dg, float* db,
                      int R, int H, const cpt_dropout* drop, int site, float* dbias, void* scratch, size_t scratch_bytes, void* stream) {
    if (!dy || !x || !g || !dx || !dg || !db) return abi_fail(CPT_ERR_NULL, "cpt_layernorm_bwd: null argument");
    if (R <= 0 || H <= 0 || H % 4) return abi_fail(CPT_ERR_SHAPE, "cpt_layernorm_bwd: bad shape R=%d H=%d", R, H);
    if (dx_lp && lp_dtype != CPT_BF16 && lp_dtype != CPT_F32) return abi_fail(CPT_ERR_DTYPE, "cpt_layernorm_bwd: lp_dtype %d", lp_dtype);
    if (int rcd = check_drop(drop, "cpt_layernorm_bwd")) return rcd;
    const bool ph = drop && drop->p_hidden > 0.f;
    const cpt::DropSpec sp = drop_spec(drop, site, false);
    return abi_check(cpt::ln_bwd(dy, x, g, eps, dx, dx_lp, lp_dtype, dg, db, R, H, R, 0, 0, 0, (hipStream_t)stream, (float*)scratch, scratch_bytes,
                                 ph ? &sp : nullptr, dbias), "cpt_layernorm_bwd");
}

int cpt_embed_ln_bwd(const float* dy, const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word, const float* posw, const float* typew,
                     const float* g, float eps, float* dword, float* dposw, float* dtypew, float* dg, float* db, int B, int Lt, int L, int H,
                     int vocab, int max_pos, int type_vocab, void* stream) {
    if (!dy || !ids || !word || !posw || !typew || !g || !dword || !dposw || !dtypew || !dg || !db) return abi_fail(CPT_ERR_NULL, "cpt_embed_ln_bwd: null argument");
    if (B <= 0 || Lt <= 0 || L < Lt || H <= 0 || H % 4) return abi_fail(CPT_ERR_SHAPE, "cpt_embed_ln_bwd: bad shape B=%d Lt=%d L=%d H=%d", B, Lt, L, H);
    return abi_check(cpt::embed_bwd(dy, ids, tt, pos, word, posw, typew, g, eps, dword, dposw, dtypew, dg, db, B, Lt, L, H, vocab, max_pos, type_vocab,
                                    (hipStream_t)stream), "cpt_embed_ln_bwd");
}

int cpt_adamw(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16,
              size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
              float grad_scale, void* stream) {
    return abi_check(cpt::adamw_flat(p, g, m, v, code, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                                     (hipStream_t)stream), "cpt_adamw");
}

int cpt_adamw_ex(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16,
                 size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                 float grad_scale, int flags, void* stream) {
    if (flags & ~(CPT_ADAMW_HF | CPT_ADAMW_NO_BIAS_CORRECTION)) return abi_fail(CPT_ERR_SHAPE, "cpt_adamw_ex: unknown flags %d", flags);
    return abi_check(cpt::adamw_flat(p, g, m, v, code, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                                     (hipStream_t)stream, flags), "cpt_adamw_ex");
}

}  // extern "C"
