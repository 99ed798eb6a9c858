/* cpt_hip.h -- C ABI of libcpt_hip.so: the MI355X (gfx950) implementation of CPT's
 * data-parallel hot path, the Oscar/BertImg forward/backward pass that scores colour-word
 * [MASK] logits over text-token + VinVL region-feature sequences.
 *
 * The reference has no native FFI for this path: it is PyTorch eager code.  Each entry point
 * below replaces the reference Python interface cited next to it (paths relative to
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in practice) unless marked host.  The library never
 *     allocates device memory and keeps no pointer after a call returns.
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void* here so the
 *     header needs no HIP include); no internal threads; the compute entry points keep no state between calls and are
 *     re-entrant.  Exceptions, all process-global and meant for A/B measurements and diagnostics only: cpt_set_tuning,
 *     cpt_prof_enable / cpt_prof_read, cpt_debug_gemm_trace.  One process drives one GPU (as torch.distributed launches the
 *     reference): kernel attributes (dynamic LDS sizes) are set once per process.
 *   - return value: CPT_OK (0) or a negative status; cpt_last_error() gives a host string for
 *     the calling thread.  No C++ exception crosses the boundary.
 *   - dtype: CPT_F32 runs every GEMM on v_mfma_f32_32x32x2_f32 (exact fp32; parity mode, matches
 *     the reference CPU path to ~1e-5); CPT_BF16 feeds bf16 operands to v_mfma_f32_32x32x16_bf16
 *     with fp32 accumulation and keeps the residual stream, LayerNorm, softmax, GELU and all
 *     reductions in fp32 (throughput mode); CPT_BF16X3 is the parity mode at MFMA-bf16 rates:
 *     every GEMM operand is split into bf16 hi + lo parts and the product a.w is computed as hi.hi + hi.lo + lo.hi
 *     (three bf16 MFMA terms, fp32 accumulate; relative error ~2^-16 per product instead of 2^-8), laid out as ONE bf16
 *     GEMM over a tripled K: activations [M][hi | hi | lo], weights [N][hi | lo | hi] (cpt_split3).  Everything
 *     outside the GEMMs runs as in CPT_F32 mode.  cpt_model_fwd takes the weight matrices of a CPT_BF16X3 model as
 *     standing split copies; cpt_train_fwd / cpt_train_bwd take them as plain fp32 (as in CPT_F32 mode: they change every
 *     step) under the tag CPT_BF16X3_MASTERS -- a descriptor of the other kind is refused, not misread -- and split both operands of every GEMM -- forward, input gradient, weight gradient -- on the spot.
 */
#ifndef CPT_HIP_H
#define CPT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPT_ABI_VERSION 8     /* 2: CPT_ATTN_MASK_3D, cpt_gemm_ln_prod3 / cpt_resid3_*, cpt_gemm_tn / cpt_gemm_nn (round 2); 3: cpt_layer_fold.w_qkv_t, cpt_retile_k32; 4: cpt_panel_pack, cpt_gemm_ln_prod3_panel (round 3); 5: cpt_train_zero_grads, cpt_batch.n_rows / mask_3d / row_seq;
                               * 6: CPT_BF16X3 in cpt_train_* (fp32 master weights there, NOT the split copies cpt_model_fwd reads under the same tag: the training step's own tag is CPT_BF16X3_MASTERS),
                               *    cpt_set_tuning / cpt_prof_* / cpt_debug_gemm_trace declared in cpt_hip_debug.h, operator-level backward entry points, cpt_comm_*;
                               * 7: cpt_adamw / cpt_adamw_ex take lr, betas, eps and weight decay as doubles (round 6);
                               * 8: cpt_outputs.loss_mean, cpt_outputs.logit_cols / n_logit_cols (round 6) */

enum { CPT_F32 = 0, CPT_BF16 = 1, CPT_BF16X3 = 2,
       CPT_BF16X3_MASTERS = 3 /* ABI 6, cpt_dims.dtype for cpt_train_* only: CPT_BF16X3 arithmetic on plain fp32 weight matrices (split per GEMM) */ };
enum { CPT_EPI_NONE = 0, CPT_EPI_GELU = 1, CPT_EPI_TANH = 2, CPT_EPI_RESID = 3 };
enum {
    CPT_OK = 0,
    CPT_ERR_SHAPE = -1,      /* unsupported / inconsistent sizes */
    CPT_ERR_DTYPE = -2,
    CPT_ERR_ALIGN = -3,      /* pointer not 16-byte aligned or K/ld not a multiple of 16 bytes */
    CPT_ERR_ARCH = -4,       /* device is not gfx950 */
    CPT_ERR_WORKSPACE = -5,  /* workspace too small */
    CPT_ERR_NULL = -6,
    CPT_ERR_HIP = -1000      /* -1000 - hipError_t */
};

int cpt_version(void);
const char* cpt_last_error(void);
/* 0 when device `dev` is gfx950, CPT_ERR_ARCH otherwise. */
int cpt_check_device(int dev);

/* ------------------------------------------------------------------------------------------
 * Model description: what BertImgModel / REC_MLM_CPT hold as nn.Module state
 * (Oscar/oscar/modeling/modeling_bert.py:153-183, modeling_rec.py:101-109).  Matrices are
 * nn.Linear layout (out x in, row-major) in the COMPUTE dtype (fp32 master tensors in CPT_F32
 * mode, a bf16 shadow in CPT_BF16 mode); vectors (biases, LayerNorm gain/shift) and the
 * embedding tables used by the gather are always fp32 -- i.e. the tensors of the state dict.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t hidden;        /* config.hidden_size (multiple of 64) */
    int32_t heads;         /* config.num_attention_heads; hidden / heads must be 64 */
    int32_t inter;         /* config.intermediate_size */
    int32_t layers;        /* config.num_hidden_layers */
    int32_t vocab;         /* config.vocab_size */
    int32_t img_dim;       /* config.img_feature_dim (2054) */
    int32_t img_dim_pad;   /* img_dim rounded up to a multiple of 8: leading dim of w_img */
    int32_t max_pos;       /* config.max_position_embeddings */
    int32_t type_vocab;    /* config.type_vocab_size */
    int32_t use_img_ln;    /* config.use_img_layernorm */
    int32_t n_rel;         /* rows of cls.seq_relationship (num_contrast_classes), 0 if absent */
    int32_t dtype;         /* CPT_F32 | CPT_BF16 | CPT_BF16X3 (matrices then are [N][3K] bf16 split copies, cpt_split3) | CPT_BF16X3_MASTERS (cpt_train_*) */
    float ln_eps;          /* config.layer_norm_eps */
    float img_ln_eps;      /* config.img_layer_norm_eps */
} cpt_dims;

typedef struct {           /* bert.encoder.layer.{i}.* */
    const void* w_qkv;     /* [3H][H]: attention.self.{query,key,value}.weight stacked */
    const float* b_qkv;    /* [3H] */
    const void* w_ao;      /* attention.output.dense.weight [H][H] */
    const float* b_ao;
    const float* ln1_g;    /* attention.output.LayerNorm.{weight,bias} */
    const float* ln1_b;
    const void* w_in;      /* intermediate.dense.weight [I][H] */
    const float* b_in;
    const void* w_out;     /* output.dense.weight [H][I] */
    const float* b_out;
    const float* ln2_g;    /* output.LayerNorm.{weight,bias} */
    const float* ln2_b;
} cpt_layer;

typedef struct {           /* optional LayerNorm-folded operands of layer i (CPT_BF16 only) */
    const void* w_qkv_f;   /* [3H][H] bf16 = output.LayerNorm.weight of layer i-1 (columns) * w_qkv; NULL for layer 0 */
    const float* c_qkv;    /* [3H] sum_k w_qkv_f[n][k] */
    const float* d_qkv;    /* [3H] sum_k output.LayerNorm.bias(i-1)[k] * w_qkv[n][k] + b_qkv[n] */
    const void* w_in_f;    /* [I][H] bf16 = attention.output.LayerNorm.weight of layer i * w_in */
    const float* c_in;     /* [I] */
    const float* d_in;     /* [I] */
    const void* w_qkv_t;   /* ABI 3, optional (NULL: not used): K-tile-major copy [H / 32][3H][32] bf16 (cpt_retile_k32) of the QKV weight this
                            * layer's fused QKV + attention launch reads -- w_qkv_f, or the plain w_qkv for layer 0 */
} cpt_layer_fold;

typedef struct {
    cpt_dims dims;
    const float* word_emb;   /* bert.embeddings.word_embeddings.weight [V][H] fp32 */
    const float* pos_emb;    /* position_embeddings.weight [P][H] */
    const float* type_emb;   /* token_type_embeddings.weight [T][H] */
    const float* emb_ln_g;   /* embeddings.LayerNorm */
    const float* emb_ln_b;
    const void* w_img;       /* bert.img_embedding.weight padded to [H][img_dim_pad], compute dtype */
    const float* b_img;
    const float* img_ln_g;   /* bert.LayerNorm (NULL unless use_img_ln) */
    const float* img_ln_b;
    const cpt_layer* layers; /* HOST array of dims.layers entries */
    const void* w_pool;      /* bert.pooler.dense [H][H] */
    const float* b_pool;
    const void* w_tr;        /* cls.transform.dense [H][H] */
    const float* b_tr;
    const float* tr_ln_g;    /* cls.transform.LayerNorm */
    const float* tr_ln_b;
    const void* w_dec;       /* cls.decoder.weight [V][H] (tied word embeddings), compute dtype */
    const float* b_dec;      /* cls.bias [V] */
    const void* w_rel;       /* cls.seq_relationship.weight [n_rel][H] compute dtype, or NULL */
    const float* b_rel;
    const cpt_layer_fold* fold; /* HOST array of dims.layers entries, or NULL: with it (and CPT_BF16) the encoder's
                                 * LayerNorms are folded into the GEMMs around them instead of running as kernels */
} cpt_model;

/* One batch as the reference drivers hand it to the model
 * (Oscar/oscar/zeroshot/refcoco_cpt.py:213-218): int64 ids / mask, fp32 region features. */
typedef struct {
    int32_t B, Lt, Li;            /* sequences, text length (70), region slots (50); L = Lt + Li */
    const int64_t* input_ids;     /* [B][Lt] */
    const int64_t* token_type;    /* [B][Lt] or NULL (zeros) */
    const int64_t* position_ids;  /* [B][Lt] or NULL (arange) */
    const int64_t* attn_mask;     /* [B][L] 1 = attend, or NULL (ones) */
    const float* img_feats;       /* [B][Li][img_dim] fp32, or NULL when Li == 0 */
    const int64_t* mask_pos;      /* [B] position of [MASK] per sequence (rows mode) or NULL */
    const int64_t* labels;        /* rows mode: [B] target id (-1 ignored); all-rows: [B][L]; or NULL */
    /* ABI 5, training only (0 / NULL: one labelled position per sequence, as above): an arbitrary label grid
     * (masked_lm_labels of modeling_rec.py:147-150 with any number of labelled positions per sequence) as n_rows labelled
     * positions -- position mask_pos[r] of sequence row_seq[r] carries labels[r]; the MLM head, the loss and o->logits
     * [n_rows][V] run over those rows (cpt_train_workspace_bytes_rows sizes the workspace) */
    int32_t n_rows;
    int32_t mask_3d;              /* ABI 5, training only: 1 = attn_mask is [B][L][L], one mask row per query (modeling_bert.py:215-216; inference
                                   * passes CPT_ATTN_MASK_3D instead); the attention backward then runs its generic kernel */
    const int64_t* row_seq;       /* [n_rows] sequence index of every labelled row, ascending; NULL with n_rows = 0 */
} cpt_batch;

enum {
    CPT_OUT_SEQ = 1,        /* sequence_output [B][L][H] fp32 (BertImgModel.forward()[0]) */
    CPT_OUT_POOLED = 2,     /* pooled_output [B][H] fp32 (modeling_bert.py:275) */
    CPT_OUT_MASK_LOGITS = 4,/* prediction scores of the [MASK] rows only [B][V] fp32 */
    CPT_OUT_ALL_LOGITS = 8, /* prediction scores of every position [B][L][V] (modeling_rec.py:143) */
    CPT_OUT_LOSS = 16,      /* CrossEntropy(ignore_index=-1) (modeling_rec.py:147-150) */
    CPT_OUT_REL = 32,       /* cls.seq_relationship(pooled) [B][n_rel] (modeling_vcr.py NSPCPT) */
    CPT_ATTN_MASK_3D = 256  /* input flag: cpt_batch.attn_mask is [B][L][L], one mask row per query (modeling_bert.py:215-216);
                               inference only, attention runs as its own kernel (no QKV fusion) */
};  /* (training keeps its activations through cpt_train_fwd / cpt_train_bwd below, not through a flag here) */

typedef struct {
    float* seq;          /* CPT_OUT_SEQ */
    float* pooled;       /* CPT_OUT_POOLED */
    float* logits;       /* CPT_OUT_MASK_LOGITS [B][V] or CPT_OUT_ALL_LOGITS [B][L][V] */
    float* loss;         /* CPT_OUT_LOSS: [2] = {sum of labelled-row losses, labelled-row count} */
    float* rel;          /* CPT_OUT_REL */
    float* loss_mean;    /* ABI 8, cpt_train_fwd(_ex) only, optional (NULL: not written): [1] = loss[0] / loss[1], the value REC_MLM_CPT.forward /
                          * NSPCPT.forward return (modeling_rec.py:147-150, modeling_vcr.py:126-128), written by the cross-entropy launch itself so that
                          * the host needs no divide kernel behind the forward; cpt_model_fwd ignores the field */
    const int64_t* logit_cols;   /* ABI 8, cpt_model_fwd with CPT_OUT_MASK_LOGITS only, optional (NULL: every column): DEVICE list of n_logit_cols vocabulary ids;
                                  * `logits` is then [B][n_logit_cols] -- the scores of those columns only, in list order.  What the zero- / few-shot
                                  * drivers read of the [B][V] scores is a handful of colour-token columns (zeroshot/refcoco_cpt.py:219,
                                  * fewshot/refcoco_cpt.py:272-291, gqa_cpt.py:598-600): the 47 MB decoder table is not streamed and B x V floats are not written.
                                  * Not with CPT_OUT_LOSS / CPT_OUT_ALL_LOGITS (CPT_ERR_SHAPE).  Ids outside [0, V) are the caller's bug (clamped, never a fault) */
    int64_t n_logit_cols;
} cpt_outputs;

/* Workspace the caller must supply for cpt_model_fwd with these flags (bytes). */
size_t cpt_fwd_workspace_bytes(const cpt_dims* d, int B, int Lt, int Li, int flags);

/* REC_MLM_CPT.forward / BertImgModel.forward (modeling_rec.py:137-152, modeling_bert.py:199-279):
 * embeddings + region projection written into one [B][L][H] buffer, N encoder layers, then the
 * outputs selected by `flags`. */
int cpt_model_fwd(const cpt_model* m, const cpt_batch* b, const cpt_outputs* o, int flags,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Few-shot training step (Oscar/oscar/fewshot/refcoco_cpt.py:231-249): forward that keeps the
 * activations, backward into caller-owned fp32 gradient tensors laid out like the parameters,
 * fused AdamW.  [MASK]-rows mode only (the loss of modeling_rec.py:147-150 sees only those rows).
 * The plain forms run without dropout; the _ex forms below take the dropout description.
 * ---------------------------------------------------------------------------------------- */
typedef struct {           /* gradients of cpt_layer, all fp32, same shapes */
    float* w_qkv; float* b_qkv; float* w_ao; float* b_ao; float* ln1_g; float* ln1_b;
    float* w_in; float* b_in; float* w_out; float* b_out; float* ln2_g; float* ln2_b;
} cpt_layer_grads;

typedef struct {           /* gradients of cpt_model: the MLM-head fields (w_tr .. b_dec) or the NSP-head fields (w_pool .. b_rel) */
    float* word_emb;       /* [V][H]: tied table, decoder + embedding-lookup contributions */
    float* pos_emb; float* type_emb; float* emb_ln_g; float* emb_ln_b;
    float* w_img;          /* [H][img_dim] (unpadded) */
    float* b_img; float* img_ln_g; float* img_ln_b;
    const cpt_layer_grads* layers;   /* HOST array */
    float* w_tr; float* b_tr; float* tr_ln_g; float* tr_ln_b; float* b_dec;
    float* w_pool; float* b_pool;    /* NSP head (model without w_tr / w_dec but with w_pool and w_rel): bert.pooler.dense */
    float* w_rel; float* b_rel;      /* ... and the relation Linear [n_rel][H] (NSPCPT.cls, modeling_vcr.py:90-92) */
} cpt_model_grads;

size_t cpt_train_workspace_bytes(const cpt_dims* d, int B, int Lt, int Li);
size_t cpt_train_workspace_bytes_rows(const cpt_dims* d, int B, int Lt, int Li, int n_rows);   /* with cpt_batch.n_rows labelled rows (0: B) */
/* forward: o->logits [B][V] and o->loss[2] = {sum of row losses, labelled-row count} are written;
 * b->mask_pos and b->labels ([B], -1 = ignored) are required.
 * A model WITHOUT the MLM head (w_tr, w_dec NULL) but with w_pool and w_rel trains the NSP-CPT head instead
 * (NSPCPT.forward, modeling_vcr.py:115-129; fewshot/vcr_nsp_cpt.py:425-470): pooled [CLS] -> tanh -> Linear(H, n_rel) ->
 * CrossEntropyLoss(ignore_index=-1) over b->labels [B]; o->rel [B][n_rel] is written instead of o->logits, b->mask_pos is not used. */
int cpt_train_fwd(const cpt_model* m, const cpt_batch* b, const cpt_outputs* o, void* workspace,
                  size_t workspace_bytes, void* stream);
/* backward of loss = loss_scale * mean over labelled rows; uses the workspace cpt_train_fwd filled.
 * WRITE / ACCUMULATE contract of g's tensors (ABI 5):
 *   WRITTEN whole by every call (previous contents ignored, never added to): every Linear weight gradient -- w_qkv, w_ao, w_in, w_out of each
 *     layer, w_img, w_pool, w_tr, w_rel -- and, with the MLM head, the tied word-embedding / decoder table (decoder gradient written, lookup
 *     gradients added behind it);
 *   ACCUMULATED into with atomics (must hold zeros, or whatever the caller wants added to): every bias and LayerNorm gain / shift vector, the
 *     position and token-type tables, and the word table when the model has no MLM head.
 * cpt_train_zero_grads clears exactly the second group.  A caller that accumulates micro-batches must therefore ADD the written tensors
 * itself between calls (cpt_amd/train.py keeps a copy and adds it); skipping the zero launch sums the vectors but OVERWRITES the matrices. */
int cpt_train_bwd(const cpt_model* m, const cpt_batch* b, const cpt_model_grads* g, float loss_scale,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Clears what cpt_train_bwd ADDS into -- the bias / LayerNorm gradient vectors and the small embedding tables (atomic
 * accumulation), plus any table the model's head does not overwrite -- in one launch.  Linear weight gradients (and, with the MLM
 * head, the tied word-embedding table) are WRITTEN by cpt_train_bwd, so a caller that runs this before every backward never
 * clears them: 0.3 MB per step instead of the whole 447 MB gradient buffer (replaces optimizer.zero_grad(),
 * Oscar/oscar/fewshot/refcoco_cpt.py:247-249).  Li = region slots of the batch that follows (0: w_img is cleared too). */
int cpt_train_zero_grads(const cpt_model* m, const cpt_model_grads* g, int Li, void* stream);
/* Data-parallel fine-tuning (the reference wraps the model in DistributedDataParallel, whose bucketed gradient
 * all-reduce overlaps backward: Oscar/oscar/fewshot/refcoco_cpt.py:516-522).  The parameters form
 * dims.layers + 2 BUCKETS: 0 = embedding tables + embeddings.LayerNorm + region projection (+ its LayerNorm),
 * 1 + l = encoder layer l, dims.layers + 1 = pooler + MLM head.  The _ex forms call back on the HOST, from inside
 * the call, on the calling thread:
 *   cpt_train_fwd_ex: before_bucket(user, k) BEFORE the first launch that reads bucket k's parameters is enqueued
 *                     (order 0, 1 .. layers, layers + 1) -- the caller makes `stream` wait for k's parameter all-gather;
 *   cpt_train_bwd_ex: grads_ready(user, k) AFTER the last launch that writes bucket k's gradients is enqueued
 *                     (order layers + 1, layers .. 1, 0) -- the caller records an event on `stream` and starts
 *                     bucket k's reduce-scatter on its communication stream while backward continues.
 * The library itself owns no communicator: collectives stay in torch.distributed (RCCL).  NULL callbacks = the plain forms. */
typedef void (*cpt_bucket_fn)(void* user, int bucket);

/* Dropout of the training step (reference: nn.Dropout at modeling_bert.py:57 on the attention probabilities, :266 on the
 * region embeddings, and inside BertEmbeddings / BertSelfOutput / BertOutput; p = config.hidden_dropout_prob /
 * attention_probs_dropout_prob, --drop_out 0.1 in fewshot/refcoco_cpt.py:387,509-512).  Masks are never stored: forward
 * and backward regenerate them from Philox4x32-10 keyed by `seed` with the counter (element block, step, site), so a mask
 * is a pure function of (seed, step, site, element) whatever the tile shapes or the number of GPUs.  Pass the SAME
 * struct to the forward and the backward of a step; ranks of a data-parallel job use different seeds (their batches
 * differ).  NULL, or both probabilities 0, runs exactly the dropout-free kernels.  The rate actually applied is p rounded
 * to 2^-32 (hidden) / 2^-16 (attention), and kept values are scaled by 1 / (1 - that rate). */
typedef struct {
    float p_hidden;      /* embeddings, region embeddings, BertSelfOutput, BertOutput */
    float p_attn;        /* attention probabilities */
    uint64_t seed;
    uint64_t step;       /* training step: a fresh mask every step */
} cpt_dropout;
/* Keep-mask of one dropout site (tests / oracle): site 0 = embeddings; layer l: 1 + 3l attention probabilities,
 * 2 + 3l BertSelfOutput, 3 + 3l BertOutput.  is_attn 0: out[n0 = rows][n1 = hidden]; 1: out[n0 = B*heads][n1 = L][n2 = L].
 * out[i] = 1 keep / 0 drop. */
int cpt_dropout_mask(const cpt_dropout* drop, int site, int is_attn, unsigned char* out, int n0, int n1, int n2, void* stream);

int cpt_train_fwd_ex(const cpt_model* m, const cpt_batch* b, const cpt_outputs* o, void* workspace,
                     size_t workspace_bytes, void* stream, cpt_bucket_fn before_bucket, void* user, const cpt_dropout* drop);
/* loss_scale_dev (optional): DEVICE scalar multiplied into loss_scale -- autograd's incoming gradient of the loss
 * without a host synchronisation. */
int cpt_train_bwd_ex(const cpt_model* m, const cpt_batch* b, const cpt_model_grads* g, float loss_scale,
                     const float* loss_scale_dev, void* workspace, size_t workspace_bytes, void* stream,
                     cpt_bucket_fn grads_ready, void* user, const cpt_dropout* drop);

/* torch.optim.AdamW update (fewshot/refcoco_cpt.py:343,249) over flat buffers of n fp32 elements
 * (n % 4 == 0).  code[i]: 0 = no gradient on this path (skipped), 1 = weight decay, 2 = no decay
 * (fewshot/refcoco_cpt.py:320-338).  grad is multiplied by grad_scale first (1/world after a
 * sum all-reduce).  shadow_bf16 (optional) receives the bf16 copy of the updated parameters.
 * ABI 7 (round 6): lr, beta1, beta2, eps, weight_decay are DOUBLES, as the reference's optimizers hold them (Python floats): the derived scalars
 * (1 - beta, 1 - lr * weight_decay, lr / (1 - beta1^t), sqrt(1 - beta2^t)) are formed in double on the host and rounded to fp32 ONCE, where torch
 * rounds them -- with float betas 1 - 0.999f was 1.3e-5 off 0.001 (found by the float64 golden vectors of tests/golden/hf_adamw.npz). */
int cpt_adamw(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16,
              size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
              float grad_scale, void* stream);

/* Round 5 (ABI 6): the same launch with the arithmetic of the GQA / VCR few-shot drivers' optimizer, pytorch_transformers.AdamW
 * (Oscar/oscar/fewshot/vcr_nsp_cpt.py:385, gqa_cpt.py:342; transformers@067923d optimization.py, not vendored):
 *   CPT_ADAMW_HF: m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;  p -= lr sqrt(1 - beta2^t) / (1 - beta1^t) m / (sqrt(v) + eps);
 *                 then, for code 1, p -= lr weight_decay p (on the UPDATED p).  eps sits outside the bias correction, the decay behind the step: both differ
 *                 from torch.optim.AdamW (flags = 0 = cpt_adamw) in the last digits only, but a drop-in follows its driver's optimizer.
 *   CPT_ADAMW_NO_BIAS_CORRECTION: correct_bias = False (both corrections 1). */
enum { CPT_ADAMW_HF = 1, CPT_ADAMW_NO_BIAS_CORRECTION = 2 };
int cpt_adamw_ex(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16,
                 size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                 float grad_scale, int flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (the kernels cpt_model_fwd is built from; also what the parity
 * tests call one by one).
 * ---------------------------------------------------------------------------------------- */

/* out = epi(A[M][K] . W[N][K]^T + bias (+ resid)); A, W in `dtype`; out in `out_dtype`
 * (CPT_F32 always allowed; CPT_BF16 only with dtype CPT_BF16).  Replaces nn.Linear (+gelu /
 * tanh / residual add) at modeling_bert.py:38-40,85,144,145,261,275 and modeling_rec.py:143. */
int cpt_gemm(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias,
             const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
             void* stream);

/* BertEmbeddings.forward (call site modeling_bert.py:244-245): LN(word[ids]+pos[pid]+type[tt])
 * for B*Lt tokens, written to rows b*L + t of out_f32 [B*L][H] (and out_lp when non-NULL). */
int cpt_embed_ln(const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word,
                 const float* posw, const float* typew, const float* g, const float* bta, float eps,
                 float* out_f32, void* out_lp, int lp_dtype, int B, int Lt, int L, int H, int vocab,
                 int max_pos, int type_vocab, void* stream);

/* BertLayerNorm over rows of x[R][H] (fp32), biased variance, eps inside the sqrt.  Input row r
 * goes to output row (r / grp) * grp_stride + grp_off + r % grp (grp = R, stride 0, off 0 for a
 * plain LayerNorm; grp = Li, stride = L, off = Lt places region rows behind the text rows,
 * replacing torch.cat at modeling_bert.py:269).  g == NULL skips the affine+norm (copy/cast). */
int cpt_layernorm_rows(const float* x, const float* g, const float* bta, float eps, float* out_f32,
                       void* out_lp, int lp_dtype, int R, int H, int grp, int grp_stride,
                       int grp_off, void* stream);

/* CaptionBertSelfAttention core (modeling_bert.py:42-67) for head_dim 64: qkv [B*L][3H] ->
 * ctx [B*L][H]; softmax(QK^T/8 + (1-mask)*-10000) V, scores never leave the chip.  L <= 288: MFMA kernels (score strip in registers); 288 < L <= 1024
 * (round 5; the reference accepts 512 text positions + regions, no CPT configuration needs more than 265): a one-wave-per-query coverage kernel, inference only.
 * probs (optional, [B][heads][L][L] in `dtype`) is written only for the backward pass. */
int cpt_attention(int dtype, const void* qkv, const int64_t* attn_mask, void* ctx, void* probs,
                  int B, int L, int heads, void* stream);

/* x[R][K] fp32 -> out[R][Kp] in `dtype`, zero-padded (region features / img weight, K=2054). */
int cpt_pad_cast(const float* x, void* out, int dtype, int R, int K, int Kp, void* stream);

/* Score extraction on the device: only indices go back to the host.
 * cpt_select_regions: per query q, its proposal sequences are rows query_first[q] .. query_first[q+1]-1 of
 * mask_logits [S][V]; sequence s contributes the logits at color_ids[s][0..C) (entries < 0 = trailing padding), in
 * order, divided by its "none" logit when divide_by_none (fewshot/refcoco_cpt.py:291) or raw
 * (zeroshot/refcoco_cpt.py:242); out_idx[q] = position of the maximum inside that concatenation with torch.argmax
 * semantics (first maximum; NaN is maximal), -1 if the query has no colour; out_score[q] (may be NULL) its value.
 * cpt_argmax_columns: out_idx[r] = argmax_j logits[r][ids[j]] (gqa_cpt.py:598-601). */
int cpt_select_regions(const float* mask_logits, int V, const int64_t* color_ids, int C, const int32_t* query_first, int Q,
                       int64_t none_id, int divide_by_none, int64_t* out_idx, float* out_score, void* stream);
int cpt_argmax_columns(const float* logits, int V, const int64_t* ids, int n_ids, int R, int64_t* out_idx, float* out_val,
                       void* stream);

/* Fold LayerNorm(gamma, beta) into the Linear (W [N][K] fp32, bias [N] or NULL) that consumes its output:
 * Wf = bf16(gamma[k] * W[n][k]), colc[n] = sum_k Wf[n][k], cold[n] = sum_k beta[k] W[n][k] + bias[n]. */
int cpt_fold_ln_weights(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_bf16,
                        float* colc, float* cold, int N, int K, void* stream);

/* K-tile-major copy of a bf16 weight: dst[K / 32][N][32] = src[N][K] (K % 32 == 0): the layout cpt_layer_fold.w_qkv_t holds.  The
 * fused QKV + attention launch stages K-tiles of 32; out of this copy every 16-row LDS-DMA piece is one contiguous KiB. */
int cpt_retile_k32(const void* src_bf16, void* dst_bf16, int N, int K, void* stream);

/* The two GEMM forms of the fused bf16 encoder, as stand-alone operators (bf16 operands, fp32 accumulate).
 * Row statistics travel as partial sums: st[M][slots][2] (sum, sum of squares), one slot per 96-column block of the
 * `hidden`-wide producer, slots = number of blocks rounded up to even; readers add the slots in index order.
 *   consumer:  out = [gelu]( LayerNorm(A; st_in) . W^T + bias ), with the LayerNorm folded: Wf = gamma (.) W in bf16,
 *              colc / cold from cpt_fold_ln_weights; BertIntermediate (modeling_bert.py:144) and the Q|K|V projections.
 *   producer:  out_f32 = A . W^T + bias + R, R = resid or LayerNorm(resid; st_in, g_in, b_in) when g_in != NULL; also writes
 *              the bf16 copy and st_out (partial row sums of out_f32); BertSelfOutput / BertOutput (:85-86, :145). */
int cpt_gemm_ln_cons(const void* A_bf16, int lda, const void* Wf_bf16, int ldw, const float* st_in, const float* colc,
                     const float* cold, float eps, int hidden, int gelu, void* out_bf16, int ldo, int M, int N, int K, void* stream);
int cpt_gemm_ln_prod(const void* A_bf16, int lda, const void* W_bf16, int ldw, const float* bias, const float* resid, int ldr,
                     const float* st_in, const float* g_in, const float* b_in, float eps, int hidden, float* out_f32,
                     void* out_bf16, float* st_out, int ldo, int M, int N, int K, void* stream);

/* The producer with the residual stream in the 3-byte form (round 2; cpt_model_fwd's default in bf16 mode, cpt_set_tuning(9, 0)
 * restores the fp32 + bf16 pair): a pre-LayerNorm sum x travels as T = its fp32 pattern rounded to 24 bits (round half away),
 * hi = (T + 0x80) >> 8 as a plain bf16 tensor (the next GEMM's A operand) and lo = the signed byte T - (hi << 8);
 * x' = ((hi << 8) + lo) << 8, |x' - x| <= 2^-17 |x|.  3 + 3 bytes per element through the epilogue instead of 4 + 6.
 *   cpt_gemm_ln_prod3: as cpt_gemm_ln_prod with resid = (resid_hi bf16, resid_lo int8) [M][ldr], out = (out_hi bf16, out_lo int8) [M][ldo]
 *   cpt_resid3_split:  x fp32 [n] -> hi, lo (n % 4 == 0, 16-byte aligned)
 *   cpt_resid3_merge:  the inverse into out fp32 [R][H]; gather = 0: source row r; gather = 1: source row r * L + pos[r]
 *                      (pos NULL: r * L), the rows the heads read (modeling_rec.py:143, modeling_bert.py:275) */
int cpt_gemm_ln_prod3(const void* A_bf16, int lda, const void* W_bf16, int ldw, const float* bias, const void* resid_hi, const void* resid_lo,
                      int ldr, const float* st_in, const float* g_in, const float* b_in, float eps, int hidden, void* out_hi, void* out_lo,
                      float* st_out, int ldo, int M, int N, int K, void* stream);
int cpt_resid3_split(const float* x, void* hi_bf16, void* lo_i8, size_t n, void* stream);
/* Round 3 (ABI 4): the producer with its A operand in the fragment-major "panel" layout
 *   panel[M / 32][K / 16][64][8] bf16: element (row, k) at (((row / 32) (K / 16) + k / 16) 64 + ((k % 16) / 8) 32 + row % 32) 8 + k % 8
 * (every v_mfma_f32_32x32x16_bf16 A operand of a 32-row block is one contiguous KiB), read straight into registers: only W rides
 * the LDS ring.  Inside cpt_model_fwd the attention kernel and the FFN-up epilogue write ctx / h in this layout; results are
 * bit-identical to cpt_gemm_ln_prod3 on the row-major tensor.  M % 128 == 0, N % 192 == 0, K % 256 == 0, K >= 512.
 *   cpt_panel_pack: row-major bf16 [M][ld] -> panel (to_panel = 1) or back (to_panel = 0, ld = leading dimension of the row-major side);
 *                   M % 32 == 0, K % 16 == 0. */
int cpt_panel_pack(const void* src_bf16, int ld, void* dst_bf16, int M, int K, int to_panel, void* stream);
int cpt_gemm_ln_prod3_panel(const void* A_panel, const void* W_bf16, int ldw, const float* bias, const void* resid_hi, const void* resid_lo,
                            int ldr, const float* st_in, const float* g_in, const float* b_in, float eps, int hidden, void* out_hi, void* out_lo,
                            float* st_out, int ldo, int M, int N, int K, void* stream);
int cpt_resid3_merge(const void* hi_bf16, const void* lo_i8, const int64_t* pos, float* out, int R, int L, int H, int gather, void* stream);
/* Round 5 (ABI 6): the producer with the RESIDUAL STREAM in the panel layout as well -- resid_hi / out_hi as [M / 32][N / 16][64][8] bf16 (the very
 * A-operand panel the next GEMM reads), resid_lo / out_lo as [M / 32][N / 16][64][8] bytes (cpt_panel_pack_bytes: the same index arithmetic on
 * one-byte elements) -- with a register-direct epilogue (swapped MFMA operands, lane = output row; no LDS slab, no barrier behind the K loop).
 * Element for element and slot for slot the same bits as cpt_gemm_ln_prod3_panel / cpt_gemm_ln_prod3 on the row-major tensors.  Inside
 * cpt_model_fwd it is what the fused bf16 encoder runs at panel-eligible shapes (BertSelfOutput / BertOutput, modeling_bert.py:85-86, :145). */
int cpt_panel_pack_bytes(const void* src_i8, int ld, void* dst_i8, int M, int K, int to_panel, void* stream);
int cpt_gemm_ln_prod3_rpanel(const void* A_panel, const void* W_bf16, int ldw, const float* bias, const void* resid_hi_panel, const void* resid_lo_panel,
                             const float* st_in, const float* g_in, const float* b_in, float eps, int hidden, void* out_hi_panel, void* out_lo_panel,
                             float* st_out, int M, int N, int K, int waves, void* stream);
/* (waves: 0 = the library's choice by shape -- 4 x 2 waves of 32 x 96 when the tiles fit one round, 4 x 1 waves of 32 x 192 over several rounds --,
 *  8 / 4 = that shape for THIS call; same bits either way) */

/* Weight-gradient GEMM in the TN form (what cpt_train_bwd runs for dW = dY^T . X, fewshot/refcoco_cpt.py:248's autograd of
 * every nn.Linear): out[M][N] fp32 = sum over k < K of A[k][m] * W[k][n], A bf16 [K][lda], W bf16 [K][ldw] -- both operands
 * as the backward pass holds them (rows = tokens), no transposed copies.  M % 128 == 0, N % 128 == 0 or N % 192 == 0,
 * K % 64 == 0, ldo == N.  k_rows (0: K): token rows that exist when K was rounded up to a multiple of 64 (the rest read as zero).
 * partials (optional, partial_bytes): scratch for split-K partial matrices (up to 8 * M * N * 4
 * bytes are used when the output has few tiles); they are added in split order, so the result is reproducible. */
int cpt_gemm_tn(const void* A_bf16, int lda, const void* W_bf16, int ldw, float* out, int ldo, int M, int N, int K, int k_rows,
                void* partials, size_t partial_bytes, void* stream);
/* Data-gradient GEMM in the NN form (dX = dY . W against an nn.Linear weight as stored): out[M][N] = A[M][K] . W[K][N] (+ resid),
 * A bf16 [M][lda], W bf16 [K][ldw] (row = contraction index = out_features), out fp32 (optionally + fp32 resid [M][ldr]) or bf16.
 * N % 192 == 0 or N % 128 == 0, K % 64 == 0.  w_rows (0: K): rows of W that exist when K was rounded up to a multiple of 64 -- rows beyond read as
 * zero, A's extra columns must hold zeros (the vocabulary-sized decoder, 30522 rows).  partials (optional): scratch for split-K when the
 * output has few tiles and K is long (fp32 output without residual; up to 64 * M * N * 4 bytes used, added in split order). */
int cpt_gemm_nn(const void* A_bf16, int lda, const void* W_bf16, int ldw, const float* resid, int ldr, void* out, int out_dtype, int ldo,
                int M, int N, int K, int w_rows, void* partials, size_t partial_bytes, void* stream);

/* Split-operand copy for CPT_BF16X3: x fp32 [R][K] (leading dimension ld) -> out bf16 [R][3K] holding, per row, the blocks
 * hi | hi | lo (weight_order 0: activations) or hi | lo | hi (weight_order 1: nn.Linear weights), hi = bf16(x),
 * lo = bf16(x - hi).  A bf16 GEMM of the two over K' = 3K is x.w to ~2^-16 relative. */
int cpt_split3(const float* x, int ld, void* out_bf16, int R, int K, int weight_order, void* stream);

/* out[b][:] = src[b*L + pos[b]][:] (pos NULL = row 0): the [MASK] rows
 * (zeroshot/refcoco_cpt.py:219) and the [CLS] rows of BertPooler. */
int cpt_gather_rows(const void* src, int dtype, const int64_t* pos, void* out, int B, int L, int H,
                    void* stream);

/* CrossEntropyLoss(ignore_index=-1) over rows of logits[R][V] (modeling_rec.py:147-150).
 * loss[0] += sum of row losses, loss[1] += labelled-row count (caller zeroes loss first);
 * dlogits (optional) = softmax - onehot for labelled rows, 0 otherwise (caller scales by 1/count). */
int cpt_ce_rows(const float* logits, const int64_t* labels, float* loss, float* dlogits, int R, int V,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level BACKWARD entry points (ABI 6; SURVEY 8(b): attention_bwd, bias_residual_ln_bwd, embed_ln_bwd): the kernels cpt_train_bwd
 * launches, one by one -- what autograd runs under loss.backward() (Oscar/oscar/fewshot/refcoco_cpt.py:248) for the blocks of
 * Oscar/oscar/modeling/modeling_bert.py:30-70 (self-attention), :85-86 / :145 (BertSelfOutput / BertOutput LayerNorm) and :244-245 (BertEmbeddings).
 * `drop` / `site` as in cpt_train_*_ex (NULL or p = 0: no dropout): masks are regenerated from (seed, step, site), never stored.
 *
 * cpt_attention_bwd: qkv [B*L][3H] and dctx [B*L][H] -> dqkv [B*L][3H] (written) for ctx = dropout(softmax(q k^T / 8 + mask)) v per head, the probabilities
 *   recomputed.  dtype CPT_F32: fp32 tensors; CPT_BF16: bf16 tensors (MFMA kernel, L <= 288); CPT_BF16X3: fp32 tensors through the split-operand
 *   MFMA kernel.  dbias_qkv (optional, [3H] fp32): += column sums of dqkv (the stacked Q | K | V bias gradient).  mask_3d: attn_mask is [B][L][L].
 * cpt_layernorm_bwd: y = LayerNorm(x; g, b) over rows of x [R][H] fp32 (x = the stored pre-LayerNorm sum): dx [R][H] fp32 written; dg, db += their sums
 *   (atomics: zero them first); dx_lp (optional): the copy of dx in lp_dtype THROUGH the hidden-site dropout mask of `site` when drop is given
 *   (the gradient entering the dense layer in front of the residual add), dbias (optional, [H]) += its column sums.  scratch (optional,
 *   >= (R / 4) * 3 * H * 4 bytes): two-stage column sums instead of per-block atomics.
 * cpt_embed_ln_bwd: backward of cpt_embed_ln for dy = rows b * L + t of a [B*L][H] fp32 gradient: LayerNorm backward, then scatter-add into dword
 *   (rows of padding_idx 0 skipped, as nn.Embedding), dposw, dtypew; dg, db += (zero everything first).
 * ---------------------------------------------------------------------------------------- */
int cpt_attention_bwd(int dtype, const void* qkv, const int64_t* attn_mask, int mask_3d, const void* dctx, void* dqkv, float* dbias_qkv,
                      int B, int L, int heads, const cpt_dropout* drop, int site, void* stream);
int cpt_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx, void* dx_lp, int lp_dtype, float* dg, float* db,
                      int R, int H, const cpt_dropout* drop, int site, float* dbias, void* scratch, size_t scratch_bytes, void* stream);
int cpt_embed_ln_bwd(const float* dy, const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word, const float* posw, const float* typew,
                     const float* g, float eps, float* dword, float* dposw, float* dtypew, float* dg, float* db, int B, int Lt, int L, int H,
                     int vocab, int max_pos, int type_vocab, void* stream);

/* ------------------------------------------------------------------------------------------
 * Communicator block (ABI 6) for hosts that are NOT PyTorch.  cpt_amd itself keeps its collectives in torch.distributed
 * (backend "nccl" = RCCL over xGMI) and starts them from the bucket callbacks of cpt_train_fwd_ex / cpt_train_bwd_ex; a host without
 * torch.distributed drives the same data-parallel step through these calls instead: sum the flat gradient (cpt_allreduce_grads, or
 * cpt_reduce_scatter + cpt_allgather around a sharded cpt_adamw) where the reference's DistributedDataParallel all-reduces it
 * (Oscar/oscar/fewshot/refcoco_cpt.py:516-522), and gather the per-rank result arrays where the reference all_gathers pickled dicts
 * (Oscar/oscar/utils/comm.py:102-142).  RCCL is bound at run time (dlopen of librccl.so.1: no link-time dependency; inside a PyTorch
 * process the soname resolves to the copy torch loaded).  ONE communicator per process, on the device current at cpt_comm_init; rank 0
 * creates the 128-byte id with cpt_comm_unique_id and the host ships it to the other ranks out of band (a file, a socket, MPI).
 * All collectives are asynchronous on `stream`; dtype CPT_F32 or CPT_BF16; counts in elements; in-place allowed where RCCL allows it.
 * ---------------------------------------------------------------------------------------- */
int cpt_comm_unique_id(void* id128);                              /* host buffer of 128 bytes */
int cpt_comm_init(int rank, int nranks, const void* id128);
int cpt_comm_rank(int* rank, int* nranks);
int cpt_allreduce_grads(void* buf, size_t count, int dtype, void* stream);                                     /* in-place sum over ranks */
int cpt_reduce_scatter(const void* send, void* recv, size_t recv_count, int dtype, void* stream);              /* send holds nranks * recv_count elements */
int cpt_allgather(const void* send, void* recv, size_t send_count, int dtype, void* stream);                   /* recv holds nranks * send_count elements */
int cpt_comm_destroy(void);

/* Per-kernel event timing, the A/B switches of the kernels (cpt_set_tuning) and the per-workgroup trace live in cpt_hip_debug.h: they are
 * measurement and development entry points of the same library, not part of the interface a host binds for the hot path. */

#ifdef __cplusplus
}
#endif
#endif /* CPT_HIP_H */
