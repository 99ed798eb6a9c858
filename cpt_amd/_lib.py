"""ctypes binding of libcpt_hip.so (include/cpt_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CPT_AMD_ABLATION=1: the development build with live kernel-variant switches (include/cpt_hip_debug.h; `python -m cpt_amd.build --ablation`);
# CPT_LIB_PATH: developer A/B builds (tools/ only)
LIB_PATH = os.environ.get("CPT_LIB_PATH") or os.path.join(HERE, "libcpt_hip_abl.so" if os.environ.get("CPT_AMD_ABLATION", "0") not in ("", "0") else "libcpt_hip.so")

CPT_F32, CPT_BF16, CPT_BF16X3, CPT_BF16X3_MASTERS = 0, 1, 2, 3
EPI_NONE, EPI_GELU, EPI_TANH, EPI_RESID = 0, 1, 2, 3
OUT_SEQ, OUT_POOLED, OUT_MASK_LOGITS, OUT_ALL_LOGITS, OUT_LOSS, OUT_REL = 1, 2, 4, 8, 16, 32
ATTN_MASK_3D = 256        # input flag: attention mask is (B, L, L)
ADAMW_HF, ADAMW_NO_BIAS_CORRECTION = 1, 2      # cpt_adamw_ex flags
K_NAMES = ["gemm_qkv", "attention", "gemm_attn_out", "layernorm", "gemm_ffn_up", "gemm_ffn_down",
           "embed_ln", "img_proj", "head", "op"]

vp, i32, i64p, f32p = C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p
BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)      # cpt_bucket_fn (host callback of cpt_train_{fwd,bwd}_ex)
NULL_CB = C.cast(None, BUCKET_CB)                        # "no callback"


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("hidden", "heads", "inter", "layers", "vocab", "img_dim", "img_dim_pad", "max_pos",
                 "type_vocab", "use_img_ln", "n_rel", "dtype")] + [("ln_eps", C.c_float), ("img_ln_eps", C.c_float)]


class Layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_in", "b_in", "w_out", "b_out",
                 "ln2_g", "ln2_b")]


class LayerFold(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_qkv_f", "c_qkv", "d_qkv", "w_in_f", "c_in", "d_in", "w_qkv_t")]


class Model(C.Structure):
    _fields_ = [("dims", Dims)] + [(n, C.c_void_p) for n in
                ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "w_img", "b_img", "img_ln_g",
                 "img_ln_b")] + [("layers", C.POINTER(Layer))] + [(n, C.c_void_p) for n in
                ("w_pool", "b_pool", "w_tr", "b_tr", "tr_ln_g", "tr_ln_b", "w_dec", "b_dec", "w_rel", "b_rel")] + \
               [("fold", C.POINTER(LayerFold))]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("Lt", C.c_int32), ("Li", C.c_int32)] + [(n, C.c_void_p) for n in
                ("input_ids", "token_type", "position_ids", "attn_mask", "img_feats", "mask_pos", "labels")] + \
               [("n_rows", C.c_int32), ("mask_3d", C.c_int32), ("row_seq", C.c_void_p)]


class LayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_in", "b_in", "w_out", "b_out",
                 "ln2_g", "ln2_b")]


class ModelGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "w_img", "b_img",
                                          "img_ln_g", "img_ln_b")] + [("layers", C.POINTER(LayerGrads))] + \
               [(n, C.c_void_p) for n in ("w_tr", "b_tr", "tr_ln_g", "tr_ln_b", "b_dec", "w_pool", "b_pool", "w_rel", "b_rel")]


class Dropout(C.Structure):
    _fields_ = [("p_hidden", C.c_float), ("p_attn", C.c_float), ("seed", C.c_uint64), ("step", C.c_uint64)]


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("seq", "pooled", "logits", "loss", "rel", "loss_mean", "logit_cols")] + [("n_logit_cols", C.c_int64)]


_SIGS = {
    "cpt_version": (C.c_int, []),
    "cpt_last_error": (C.c_char_p, []),
    "cpt_check_device": (C.c_int, [C.c_int]),
    "cpt_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.c_int]),
    "cpt_model_fwd": (C.c_int, [C.POINTER(Model), C.POINTER(Batch), C.POINTER(Outputs), C.c_int, vp, C.c_size_t, vp]),
    "cpt_train_workspace_bytes": (C.c_size_t, [C.POINTER(Dims), C.c_int, C.c_int, C.c_int]),
    "cpt_train_workspace_bytes_rows": (C.c_size_t, [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.c_int]),
    "cpt_train_fwd": (C.c_int, [C.POINTER(Model), C.POINTER(Batch), C.POINTER(Outputs), vp, C.c_size_t, vp]),
    "cpt_train_bwd": (C.c_int, [C.POINTER(Model), C.POINTER(Batch), C.POINTER(ModelGrads), C.c_float, vp, C.c_size_t, vp]),
    "cpt_train_fwd_ex": (C.c_int, [C.POINTER(Model), C.POINTER(Batch), C.POINTER(Outputs), vp, C.c_size_t, vp, BUCKET_CB, vp,
                                   C.POINTER(Dropout)]),
    "cpt_train_bwd_ex": (C.c_int, [C.POINTER(Model), C.POINTER(Batch), C.POINTER(ModelGrads), C.c_float, vp, vp, C.c_size_t, vp,
                                   BUCKET_CB, vp, C.POINTER(Dropout)]),
    "cpt_train_zero_grads": (C.c_int, [C.POINTER(Model), C.POINTER(ModelGrads), C.c_int, vp]),
    "cpt_dropout_mask": (C.c_int, [C.POINTER(Dropout), C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_attention_bwd": (C.c_int, [C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Dropout), C.c_int, vp]),
    "cpt_layernorm_bwd": (C.c_int, [vp, vp, vp, C.c_float, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.POINTER(Dropout), C.c_int, vp, vp, C.c_size_t, vp]),
    "cpt_embed_ln_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_comm_unique_id": (C.c_int, [vp]),
    "cpt_comm_init": (C.c_int, [C.c_int, C.c_int, vp]),
    "cpt_comm_rank": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cpt_allreduce_grads": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "cpt_reduce_scatter": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp]),
    "cpt_allgather": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp]),
    "cpt_comm_destroy": (C.c_int, []),
    "cpt_adamw": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                            C.c_int, C.c_float, vp]),
    "cpt_adamw_ex": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_int, C.c_float, C.c_int, vp]),
    "cpt_gemm": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int,
                           C.c_int, C.c_int, C.c_int, vp]),
    "cpt_embed_ln": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_layernorm_rows": (C.c_int, [vp, vp, vp, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, vp]),
    "cpt_attention": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_pad_cast": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_fold_ln_weights": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_cons": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_prod": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                   C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_prod3": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                    C.c_int, C.c_int, vp]),
    "cpt_panel_pack": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_prod3_panel": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                          C.c_int, C.c_int, vp]),
    "cpt_panel_pack_bytes": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_prod3_rpanel": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_tile": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_cons_tile": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_ln_prod3_panel_waves": (C.c_int, [C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                                C.c_int, C.c_int, vp]),
    "cpt_resid3_split": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "cpt_resid3_merge": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_gemm_nn": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "cpt_gemm_tn": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "cpt_split3": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_select_regions": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int64, C.c_int, vp, vp, vp]),
    "cpt_argmax_columns": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]),
    "cpt_gather_rows": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "cpt_ce_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "cpt_retile_k32": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "cpt_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "cpt_build_info": (C.c_int, []),
    "cpt_debug_gemm_trace": (C.c_int, [vp]),
    "cpt_prof_enable": (C.c_int, [C.c_int]),
    "cpt_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    # include/cpt_io.h (host-side wire-format decoder)
    "cpt_b64_decode_f32": (C.c_int, [C.c_char_p, C.c_size_t, vp, C.c_int]),
    "cpt_decode_regions": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, vp, vp]),
    "cpt_decode_regions_batch": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]),
    "cpt_decode_tsv_rows": (C.c_int, [C.POINTER(C.c_char_p), vp, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp,
                                      vp, vp, vp, vp, vp, C.c_int]),
    "cpt_pack_tsv_rows": (C.c_int, [C.POINTER(C.c_char_p), vp, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp,
                                    vp, vp, vp, vp, vp, C.c_int]),
    "cpt_b64_chars": (C.c_size_t, [C.c_int]),
    "cpt_b64_decode_regions_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "cpt_json_find_strings": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, vp, vp, C.c_int, vp, C.c_size_t,
                                        C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
}

_lib = None


def _bind_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (same soname as
    /opt/rocm's).  Load torch's copy first so libcpt_hip.so's NEEDED entry resolves to it --
    streams and device pointers handed over from torch must belong to the same runtime."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def exported_symbols():
    return sorted(_SIGS)


def ablation_build():
    """True when the loaded library is the CPT_ABLATION development build (cpt_set_tuning works)."""
    return bool(lib().cpt_build_info() & 1)


def lib():
    """The loaded library; raises if it has not been built (python -m cpt_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "cpt_amd: %s is missing -- the HIP extension is required (no CPU fallback). "
                "Build it with `python -m cpt_amd.build` (hipcc --offload-arch=gfx950)." % LIB_PATH)
        _bind_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)        # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if l.cpt_version() != 8:
            raise RuntimeError("cpt_amd: libcpt_hip ABI version %d, expected 8" % l.cpt_version())
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cpt_last_error().decode("utf-8", "replace")
        raise RuntimeError("cpt_amd %s failed (status %d): %s" % (what, rc, msg))


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()
