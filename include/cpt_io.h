/* cpt_io.h -- C ABI of the region-feature wire-format decoder, host side.
 *
 * The reference stores the VinVL region features of every (image, proposal) pair as base64 text of
 * float32[2054] inside a JSON object on one TSV line (writer: prompt_feat/maskrcnn_benchmark/engine/inference_ref.py:157-191)
 * and decodes them per box in Python: json.loads -> base64.b64decode -> np.frombuffer -> np.stack -> torch.Tensor
 * -> zero-pad to img_seq_len rows (Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:161-180 and :119-120).
 * At the GPU's rate (3e4 sequences/s x 0.55 MB of base64 each) that Python path is the bottleneck.  These entry
 * points decode straight into caller-owned (pinned) host memory; nothing is allocated or retained.
 *
 * Round 5: on CPUs with AVX2 (checked once per process) the base64 inner loop decodes 32 characters per iteration (csrc/b64_avx2.cpp); results and
 * error reports are those of the scalar loop, which still serves tails, padded groups and every block that holds an invalid character.  The
 * environment variable CPT_B64_SCALAR=1 (read once, at load time) keeps the scalar loop everywhere: a test / measurement hook, not an interface.
 *
 * Part of libcpt_hip.so; plain pointers and sizes only; every function returns CPT_OK (0) or a negative
 * status from cpt_hip.h (CPT_ERR_NULL / CPT_ERR_SHAPE), with the message in cpt_last_error().
 */
#ifndef CPT_IO_H
#define CPT_IO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One region: base64 text (standard alphabet, '=' padding, no whitespace) of exactly dim float32 values
 * (little endian, as np.float32.tobytes()) -> out[dim].  Replaces np.frombuffer(base64.b64decode(s), np.float32)
 * (refcoco_zsl_cpt_dataset.py:173).  Unlike Python's lenient b64decode, characters outside the alphabet and a
 * decoded size other than 4*dim bytes are errors (the reference's writer never produces them). */
int cpt_b64_decode_f32(const char* b64, size_t len, float* out, int dim);

/* One proposal sequence: n_regions base64 strings -> rows 0..n_regions-1 of out[max_regions][dim], rows
 * n_regions.. zero (torch.cat([feat, zeros]), refcoco_zsl_cpt_dataset.py:119-120); mask_img (may be NULL)
 * receives 1 for real regions and 0 for padding (the image part of input_mask, tokenize()). */
int cpt_decode_regions(const char* const* b64, const size_t* lens, int n_regions, int dim, int max_regions,
                       float* out, int64_t* mask_img);

/* n_seq sequences at once.  String i is the len[i] characters at base + offsets[i] (e.g. inside the raw TSV row,
 * see cpt_json_find_strings); the strings of sequence s are numbers first[s] .. first[s] + n_regions[s] - 1.
 * out[n_seq][max_regions][dim] zero padded, mask_img[n_seq][max_regions] or NULL; n_threads host threads
 * (<= 0: one). */
int cpt_decode_regions_batch(const char* base, const size_t* offsets, const size_t* lens, const int* first,
                             const int* n_regions, int n_seq, int dim, int max_regions, float* out,
                             int64_t* mask_img, int n_threads);

/* One raw TSV payload (the JSON text of a row): locates every string value of the key `key` (e.g. "feature") in
 * document order -- offsets[i] / lens[i] of the characters between its quotes, up to max_values -- and copies the
 * JSON to `stripped` with those values replaced by "" so that the caller's json.loads no longer touches the bulk
 * of the line (stripped_cap >= len suffices).  A minimal JSON string scanner (backslash escapes honoured), so the
 * key text inside another string value is not mistaken for a key.  Replaces the json.loads of the full row in
 * decode_features (refcoco_zsl_cpt_dataset.py:162-163); the located strings go to cpt_decode_regions_batch. */
int cpt_json_find_strings(const char* json, size_t len, const char* key, size_t* offsets, size_t* lens,
                          int max_values, char* stripped, size_t stripped_cap, int* n_values, size_t* stripped_len);

/* Many TSV payloads in one call, with native threads (what the GPU's rate needs: Python threads serialise on the
 * interpreter lock).  For every row: scan as cpt_json_find_strings, group the `key` values by the array that
 * encloses their objects (the per-proposal box lists of "objects"[0]: sibling arrays of the first one found,
 * empty ones included), and decode group g of row r into sequence seq0 + g of out[max_seqs][max_regions][dim]
 * (zero padded; mask_img[max_seqs][max_regions] or NULL), where seq0 is the number of groups of the rows before.
 *   stripped[r] / stripped_cap[r]: per-row buffers for the JSON without the values; stripped_len[r] out
 *   seqs_per_row[n_rows], regions_per_seq[max_seqs]: out.  Temporary host vectors only; nothing is retained.  * Outputs on a non-OK return: UNDEFINED -- rows are scanned and decoded in one pass by the worker threads, so the rows in front of the one that
 * fails (or of the max_seqs overflow) have already been written and later rows may have been decoded at a wrong base; nothing is written out of
 * bounds.  A ring slot that saw an error must be refilled before it is used. */
int cpt_decode_tsv_rows(const char* const* rows, const size_t* lens, int n_rows, const char* key, int dim,
                        int max_regions, int max_seqs, float* out, int64_t* mask_img, char* const* stripped,
                        const size_t* stripped_cap, size_t* stripped_len, int* seqs_per_row, int* regions_per_seq,
                        int n_threads);

/* ---- Round 5: decode on the DEVICE ---------------------------------------------------------------------------------------
 * At the GPU's rate the host decoder needs most of a 16-core host (4.4 MB of base64 per TSV row, 22 GB/s at 4e4 sequences/s).  The text can
 * travel to the GPU as it stands instead: the host only LOCATES the strings and copies them into (pinned) memory, a HIP kernel decodes them at
 * the memory system's rate on the stream the H2D copy ran on, straight into the [n_seq][max_regions][dim] float32 tensor cpt_batch.img_feats
 * points at.  Same strictness and the same bytes as cpt_b64_decode_f32 (tests compare the two bit for bit). */

/* Characters of one region's base64 string: 4 * ceil(4 * dim / 3)  (10956 for dim = 2054). */
size_t cpt_b64_chars(int dim);

/* Host half: cpt_decode_tsv_rows with the decode left out.  Region i of sequence s is COPIED to text[s][i][0 .. cpt_b64_chars(dim)) of
 * text[max_seqs][max_regions][cpt_b64_chars(dim)] (a value of any other length is an error; slots without a region are not written),
 * mask_img[max_seqs][max_regions] (required) receives 1 / 0 per slot; stripped / seqs_per_row / regions_per_seq as cpt_decode_tsv_rows. */
int cpt_pack_tsv_rows(const char* const* rows, const size_t* lens, int n_rows, const char* key, int dim,
                      int max_regions, int max_seqs, char* text, int64_t* mask_img, char* const* stripped,
                      const size_t* stripped_cap, size_t* stripped_len, int* seqs_per_row, int* regions_per_seq,
                      int n_threads);

/* Device half (one launch on `stream`; all pointers DEVICE memory): text_dev[n_seq][max_regions][cpt_b64_chars(dim)] + mask_img_dev[n_seq][max_regions]
 * -> out_dev[n_seq][max_regions][dim] float32, rows of slots whose mask is 0 zeroed.  *err_dev (8 bytes, zero before the first launch; stays zero
 * while every string is valid) receives ~((slot << 32) | character) of the FIRST invalid character (lowest slot = s * max_regions + i, then lowest
 * offset) -- read it back where the host decoder would have returned its error; the slot's output is undefined then. */
int cpt_b64_decode_regions_device(const void* text_dev, const int64_t* mask_img_dev, int n_seq, int dim, int max_regions, float* out_dev,
                                  unsigned long long* err_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPT_IO_H */
