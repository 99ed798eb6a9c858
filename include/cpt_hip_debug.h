/* libcpt_hip.so -- measurement and development entry points (round 4: split out of cpt_hip.h).
 *
 * Exported by the same library; NOT part of the hot-path interface of cpt_hip.h.  bench.py's roofline leg uses the per-kernel event
 * timing; tools/ and the A/B tests use cpt_set_tuning and the per-workgroup trace.  A product host never needs to call any of them.
 * Round 5: the kernel-variant switches are process-global state only in the DEVELOPMENT build of the library (-DCPT_ABLATION,
 * cpt_amd/libcpt_hip_abl.so, selected by CPT_AMD_ABLATION=1); in the product build they are compile-time constants, the code behind the
 * losing settings is not compiled, and cpt_set_tuning refuses every key but -1 (cpt_build_info tells the two apart).
 */
#ifndef CPT_HIP_DEBUG_H
#define CPT_HIP_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * ---------------------------------------------------------------------------------------- */
enum { CPT_K_GEMM_QKV = 0, CPT_K_ATTN, CPT_K_GEMM_AO, CPT_K_LN, CPT_K_GEMM_FFN1, CPT_K_GEMM_FFN2,
       CPT_K_EMBED, CPT_K_IMG, CPT_K_HEAD, CPT_K_OP /* any operator-level call */, CPT_K_COUNT };
int cpt_prof_enable(int on);                          /* resets accumulators */
int cpt_prof_read(int kernel_id, double* total_ms, int64_t* launches); /* synchronises */

/* Operator-level entry points with a PER-CALL kernel choice (round 5; product and development build alike): which of the SHIPPED tile
 * configurations / wave shapes serves a GEMM is normally the library's choice by shape; the parity and race tests walk through all of them on
 * small problems with these.  The choice is an argument of the call (thread-local for its duration): no state survives it.
 *   tile (cpt_gemm_tile, cpt_gemm_ln_cons_tile): 3 = by shape; 0 = generic register-staged kernel; 13 = 128x192 (3-stage), 11 = 192x192, 10 = 128x384,
 *     14 = 384x192, 15 = 128x192 two workgroups per CU, 18 = 64x192, 19 = 384x256 one pass; consumers also 20 = two-pass 384x256, 21 = 4-wave 192x256
 *   waves (cpt_gemm_ln_prod3_panel_waves; also an argument of cpt_gemm_ln_prod3_rpanel in cpt_hip.h): 0 = by shape, 8 = 4x2 waves of 32x96, 4 = 4x1 waves of 32x192 */
int cpt_gemm_tile(int tile, int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias,
                  const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K, void* stream);
int cpt_gemm_ln_cons_tile(int tile, const void* A_bf16, int lda, const void* Wf_bf16, int ldw, const float* st_in, const float* colc,
                          const float* cold, float eps, int hidden, int gelu, void* out_bf16, int ldo, int M, int N, int K, void* stream);
int cpt_gemm_ln_prod3_panel_waves(int waves, const void* A_panel, const void* W_bf16, int ldw, const float* bias, const void* resid_hi, const void* resid_lo,
                                  int ldr, const float* st_in, const float* g_in, const float* b_in, float eps, int hidden, void* out_hi, void* out_lo,
                                  float* st_out, int ldo, int M, int N, int K, void* stream);

/* 0 = product build (libcpt_hip.so: every kernel-variant switch below is a compile-time constant, cpt_set_tuning refuses every key but -1);
 * bit 0 set = the CPT_ABLATION development build (libcpt_hip_abl.so), where the switches are process-global (round 5). */
int cpt_build_info(void);

/* Kernel-variant switches for A/B measurements (CPT_ABLATION build only); defaults are the shipped configuration.
 *   key 0  GEMM (20 / 21: force the two-pass / the 4-wave LayerNorm-consumer kernel where legal): 0 = generic register-staged kernel only; 3 (default) = pipelined LDS-DMA kernel, tile shape chosen
 *          per GEMM; fixed shapes 13 = 128x192 (3-stage), 11 = 192x192, 10 = 128x384, 14 = 384x192,
 *          15 = 128x192 two workgroups per CU, 18 = 64x192 (small M)
 *   key 1  GEMM ablation bits: 1 no operand LDS-DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue,
 *          32 fused QKV + attention kernel without its attention phase
 *   key 2  attention backward: 0 = generic fp32-math kernel, 1 (default) = MFMA kernel for bf16, L <= 128
 *   key 3  split-K target for the generic path
 *   key 4  1 = bf16 residual stream in the kernel-per-op bf16 encoder (default 0: fp32 residual)
 *   key 5  0 = run the encoder LayerNorms as kernels even when cpt_model.fold is given (default 1: folded)
 *   key 6  QKV projection + attention: 0 = two kernels, 1 = fused, one workgroup per (sequence, head), two per CU,
 *          2 = the same, one per CU, 3 (default) = fused, one workgroup per (sequence, three heads) where heads % 3 == 0,
 *          else 1 (bf16, L <= 128 only; otherwise always two kernels)
 *   key 9  residual stream of the fused bf16 encoder: 1 (default) = 3-byte form (cpt_gemm_ln_prod3), 0 = fp32 + bf16 copies
 *   key 10 bf16 weight gradients of cpt_train_bwd: 1 (default) = TN GEMM (operands read as stored, split-K partials reduced in
 *          order), 0 = explicit operand transposes + NT GEMM
 *   key 11 fused QKV + attention, form 3: 1 (default) = read cpt_layer_fold.w_qkv_t when given, 0 = always the row-major weight
 *   key 12 FFN-up two-pass kernel: 1 (default) = refill DMA issued behind the first k-step after the barrier, 0 = right behind it
 *   key 13 timing experiments of the panel producer (gemm_prod.hip); 0 (default) = the shipped kernel
 *   key 14 panel mode of the fused bf16 encoder: 1 (default) = ctx / FFN activation in the fragment-major panel layout where shapes allow
 *          (cpt_gemm_ln_prod3_panel), 0 = row-major tensors (cpt_gemm_ln_prod3)
 *   key 15 panel mode: 1 (default) = launches that leave CUs idle carry 16 workgroups that read the next launch's weights into the
 *          Infinity Cache, 0 = no prefetch workgroups
 *   key 16 FFN-up two-pass kernel (and with it panel mode) from this many 384 x 256 tiles on (default 192; experiments with small batches)
 *   key 17 LayerNorm backward, two-stage column-sum form: rows per workgroup (default 0 = 8 from 2048 rows on, else 4; experiments)
 *   key 18 training backward, bias-gradient column sums inside their producers: bit 0 = b_in in the GELU-gradient epilogue (round 6: as partial rows that a
 *          later LayerNorm-backward launch adds up in spare workgroups), bit 1 = b_qkv in the attention backward kernel (default 3); a cleared bit runs the
 *          stand-alone column-sum launch
 *   key 19 training backward, a layer's weight gradients: 2 (default) = FFN down | FFN up | attention output in one launch + Q|K|V alone where the
 *          shapes fit one round (else as 1), 1 = two paired launches (the two FFN matrices; attention output + Q|K|V), 0 = four launches with
 *          their own split-K reductions
 *   key 20 stand-alone QKV projection with the LayerNorm folded (sequences longer than 128: attention runs as its own kernel): 1 (default) =
 *          the GELU-less form of the two-pass 384 x 256 kernel when its tiles fill the chip, 0 = the 384 x 192 pipelined kernel
 *   key 21 stand-alone attention kernel, sequences longer than 128: 1 (default) = the 128-query tiles of a (sequence, head) are neighbouring
 *          workgroups of one XCD (their shared K / V rows are fetched from memory once), 0 = the (pair, tile) grid
 *   key 22 training forward, FFN-down at 2048..6144 rows: 1 (default) = 128 x 192 tiles with K split over two workgroups, the two partial
 *          matrices added by the dropout + residual + LayerNorm pass behind it; 0 = 64 x 192 tiles over the whole K
 *   key 23 bf16x3 parity mode: 1 (default) = the FFN-up's GELU epilogue writes the split copy of its output that the FFN-down reads, 0 = an
 *          fp32 tensor and a stand-alone cpt_split3 pass
 *   key 24 wave shape of the panel LayerNorm producers' 128 x 192 tile (gemm_prod.hip; same bits either way): 0 (default) = by shape -- 4 x 1 waves of
 *          32 x 192 with the operand stream interleaved between the MFMAs when the tiles run several rounds, else 4 x 2 waves of 32 x 96 (in one
 *          round the denser 4-wave launch takes clock from its neighbours under the power cap: no gain for the step); 4 / 8 force one shape
 *   key 25 fused bf16 encoder: 1 (default) = text embedding and region-feature pad + cast in ONE launch, 0 = two launches
 *   key 26 MLM head on the [MASK] rows: percent of the decoder weight table prefetched by the gather + LayerNorm launch (default 40; the rest rides
 *          on the reduce + GELU + LayerNorm launch; 100 = round 3's form)
 *   key 27 bf16x3 parity mode: 1 (default) = attention on bf16 MFMA with split operands (three-term products), ctx written as the split copy the
 *          attention-output GEMM reads; 0 = the fp32 MFMA attention kernel and a cpt_split3 pass over ctx
 *   key 28 panel layout of the FFN activation when the producers run several rounds of tiles: 1 (default), 0 = row-major there (round 3)
 *   key 29 the 4-wave 192 x 256 LayerNorm-consumer kernel with the operand stream between the MFMAs (gemm_ffn4.hip) in place of the two-pass
 *          384 x 256 kernel (gemm_ffn.hip): 1 (default) = where its tiles fill their rounds at least 10 % better (Oscar-large FFN-up), 2 = wherever
 *          legal, 0 = nowhere; same bits
 *   key 30 fused bf16 encoder at the full-panel shapes: 1 (default) = the residual stream itself in the panel layout and the LayerNorm producers'
 *          register-direct epilogue (cpt_gemm_ln_prod3_rpanel), 0 = row-major 3-byte stream + slab epilogue (round 3); same bits
 *   key 31 fused bf16 encoder, only the [MASK] rows (or only the [CLS] rows) read behind it: 1 (default) = the last layer's attention output, FFN and
 *          LayerNorms on those rows alone (rowops.hip tail_rows / tail_finish, gemm.hip gemm_rows_split), 0 = every row through the last layer;
 *          same values to bf16 accuracy, not the same bits (different kernels behind the last attention)
 *   key 32 fused bf16 encoder, batches whose row count the full panel mode does not take: 1 (default) = the encoder's tensors are sized and launched for the
 *          next row count it takes (at most 1.5x the real rows; the padded rows belong to no sequence), 0 = such batches run the row-major kernels; same bits
 *   key 33 training step, last encoder layer: 1 (default) = everything behind its attention (attention output, FFN, both LayerNorms; forward and backward) on
 *          the head's rows only -- one per sequence ([MASK], or [CLS] for the NSP head) -- when the batch carries no label grid; 0 = all rows
 *   key 34 training backward, data-gradient GEMMs in front of a LayerNorm backward at 2048..6144 rows: 1 (default) = 128 x 192 tiles with K split over two
 *          workgroups, two BF16 partial matrices added (with the fp32 residual) by the LayerNorm backward; 0 = 64 x 192 tiles over the whole K, fp32 out
 *   key 35 training backward, the GELU-gradient data-gradient GEMM (FFN-down): 1 (default) = 256 x 192 tiles where 128-row tiles would make between one and two
 *          rounds of workgroups and 256-row tiles at most one (M = 3840: 240 instead of 480 workgroups), 0 = 128 x 192 tiles
 *   key 36 training forward with hidden dropout: 1 (default) = the LayerNorm launches write no fp32 output, the row pass behind re-forms the residual it adds
 *          from the pre-LayerNorm rows kept for the backward (+ (mean, rstd), gain, shift); 0 = fp32 outputs written and read back
 *   key 37 training backward: 1 (default) = the K-split partial matrices of a layer's Q|K|V weight gradient are added up by the workgroups the NEXT layer's
 *          three-problem weight-gradient launch leaves idle (216 of 256 CUs busy at hidden 768); 0 = a reduction launch of their own
 *   key 38 attention backward with the forward's statistics (bf16 training step, L <= 128): 1 (default) = where two workgroups per (sequence, head) still fit
 *          one per CU (2 B heads <= 256: 4 sequences per GPU) the two phases (dQ | dK, dV) run in a workgroup each, side by side; 0 = one workgroup runs both
 *   key 39 training step at few rows (4 sequences per GPU): 1 (default) = the FFN-up forward (u and gelu(u) out) on 64 x 96 tiles and the GELU-gradient
 *          data-gradient GEMM on 64 x 128 tiles where 64 x 192 tiles fill at most half the chip (also: forward Q|K|V and attention output on 64 x 96, the
 *          attention output's data gradient on 64 x 64, the unsplit TN weight gradient on 64 x 192); 0 = 64 x 192 / 128 x 192 tiles; v > 1 = the
 *          workgroup-count threshold itself instead of 128 (256 at 32 sequences: 5.315 / 5.307 vs 5.321 / 5.316 ms -- nothing, the default stays 128)
 *   key -1 restores the default of every key (value ignored) */
int cpt_set_tuning(int key, int value);
/* Debug: when buf != NULL the pipelined GEMM writes 8 int64 per workgroup (shader-clock stamps at
 * start / after prologue issue / after K loop / after staging / end, and the XCC id). */
int cpt_debug_gemm_trace(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* CPT_HIP_DEBUG_H */
