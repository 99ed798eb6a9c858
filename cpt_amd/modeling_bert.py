"""BertImgModel / BertImgForPreTraining with the reference's constructor, forward signature and
state-dict keys, executing on the HIP library.

Drop-in for /root/reference/Oscar/oscar/modeling/modeling_bert.py:150-279 (BertImgModel) and
:914-1021 (BertPreTrainingHeads, BertImgForPreTraining).  The nn.Module tree below only HOLDS
parameters under the reference's names (so ``Oscar/pretrained_models/.../pytorch_model.bin`` loads
unchanged); the arithmetic runs in libcpt_hip.so via cpt_amd.engine.  There is no eager/CPU
fallback: calling forward without the library or with CPU tensors raises.
"""
import torch
from torch import nn

from . import _lib as L
from .engine import PackedModel
from .modeling_utils import PreTrainedModel, ImgPreTrainedModel, BertPreTrainedModel  # noqa: F401

BertLayerNorm = nn.LayerNorm


# ---- parameter containers (names = state-dict keys of SURVEY.md section 8b) ------------------
class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _SelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _DenseLN(nn.Module):
    def __init__(self, n_in, n_out, eps):
        super().__init__()
        self.dense = nn.Linear(n_in, n_out)
        self.LayerNorm = BertLayerNorm(n_out, eps=eps)


class _Dense(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.dense = nn.Linear(n_in, n_out)


class _Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = _SelfAttention(config)
        self.output = _DenseLN(config.hidden_size, config.hidden_size, config.layer_norm_eps)


class CaptionBertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = _Attention(config)
        self.intermediate = _Dense(config.hidden_size, config.intermediate_size)
        self.output = _DenseLN(config.intermediate_size, config.hidden_size, config.layer_norm_eps)


class CaptionBertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(_Dense):
    def __init__(self, config):
        super().__init__(config.hidden_size, config.hidden_size)


class BertPredictionHeadTransform(_DenseLN):
    def __init__(self, config):
        super().__init__(config.hidden_size, config.hidden_size, config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))


class BertPreTrainingHeads(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)
        n = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.seq_relationship = nn.Linear(config.hidden_size, n)


def _check_unsupported(config, head_mask, encoder_history_states):
    if head_mask is not None:
        raise NotImplementedError("cpt_amd: head_mask is not used by any CPT driver and is not implemented")
    if encoder_history_states:
        raise NotImplementedError("cpt_amd: encoder_history_states (captioning) is out of scope")
    if getattr(config, "output_attentions", False) or getattr(config, "output_hidden_states", False):
        raise NotImplementedError("cpt_amd: output_attentions/output_hidden_states are not implemented")


class _EngineMixin(object):
    """compute dtype switch + lazily built PackedModel."""
    _head = "none"

    def _engine(self):
        eng = self.__dict__.get("_cpt_engine")
        if eng is None:
            eng = PackedModel(self, self.config, self._head)
            self.__dict__["_cpt_engine"] = eng
        return eng

    def set_compute_dtype(self, dtype):
        """'fp32' (exact-fp32 MFMA, parity mode; default), 'bf16' (bf16 MFMA operands, fp32 accumulate / residual /
        LayerNorm / softmax; throughput mode) or 'bf16x3' (parity mode at bf16-MFMA rates, inference and training -- every GEMM
        operand split into bf16 hi + lo, three MFMA terms, everything else as in fp32 mode)."""
        if dtype not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("compute dtype must be 'fp32', 'bf16' or 'bf16x3'")
        self._engine().dtype = dtype
        return self

    def invalidate_packed(self):
        """Call after writing parameters in place through ``.data`` (see PackedModel.invalidate)."""
        self._engine().invalidate()
        return self

    def state_dict(self, *args, **kwargs):
        eng = self.__dict__.get("_cpt_engine")
        if eng is not None and eng.pending is not None:
            eng.complete_pending()          # data parallel: a parameter all-gather may still be in flight
        return super().state_dict(*args, **kwargs)


class BertImgModel(_EngineMixin, BertPreTrainedModel):
    """modeling_bert.py:150-279."""

    def __init__(self, config):
        super().__init__(config)
        if config.img_feature_type in ("dis_code", "dis_code_t", "dis_code_scale"):
            raise NotImplementedError("cpt_amd: img_feature_type %r (discrete codes) is not used by CPT"
                                      % config.img_feature_type)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = getattr(config, "use_img_layernorm", None)
        self.img_embedding = nn.Linear(self.img_dim, config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self.apply(self.init_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                img_feats=None, encoder_history_states=None):
        _check_unsupported(self.config, head_mask, encoder_history_states)
        out = self._engine().forward(input_ids, token_type_ids, attention_mask, position_ids, img_feats,
                                     flags=L.OUT_SEQ | L.OUT_POOLED)
        return (out["seq"], out["pooled"])


class BertImgForPreTraining(_EngineMixin, ImgPreTrainedModel):
    """modeling_bert.py:927-1021: the class the pre-trained checkpoint is loaded with."""
    _head = "pretrain"

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertImgModel(config)
        self.cls = BertPreTrainingHeads(config)
        self.num_seq_relations = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.apply(self.init_weights)
        self.tie_weights()

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, masked_lm_labels=None,
                next_sentence_label=None, position_ids=None, head_mask=None, img_feats=None):
        _check_unsupported(self.config, head_mask, None)
        flags = L.OUT_ALL_LOGITS | L.OUT_REL
        want_loss = masked_lm_labels is not None and next_sentence_label is not None
        if want_loss:
            flags |= L.OUT_LOSS
        out = self._engine().forward(input_ids, token_type_ids, attention_mask, position_ids, img_feats,
                                     labels=masked_lm_labels if want_loss else None, flags=flags)
        outputs = (out["logits"], out["rel"])
        if want_loss:
            outputs = (out["loss"],) + outputs + (out["loss"],)
        return outputs
