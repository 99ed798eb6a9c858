"""NSPCPT: the VCR colour-prompt scoring wrapper (pooled [CLS] -> pre-trained seq_relationship head), on the
HIP library.

Drop-in for /root/reference/Oscar/oscar/modeling/modeling_vcr.py:79-129: same
constructor, ``copy_from_pretraining_model`` (``self.cls`` BECOMES the pre-training model's
``cls.seq_relationship`` Linear, so the saved state dict is ``bert.*`` + ``cls.{weight,bias}``), forward argument
order and outputs ``(loss?, seq_relationship_score)``.  The encoder is the same HIP path as REC_MLM_CPT; the head
is the C ABI's CPT_OUT_REL output (pooler GEMM + tanh, then Linear(H, num_contrast_classes)).

Fine-tuning (fewshot/vcr_nsp_cpt.py:425-470) runs through the same HIP training step as REC_MLM_CPT with the NSP head
(pooler + tanh + Linear(H, 3) + cross entropy) in place of the MLM head: cpt_train_fwd / cpt_train_bwd select it when the
model carries w_pool / w_rel and no MLM head.
"""
import torch
from torch import nn

from . import _lib as L
from . import ops
from .modeling_bert import BertImgModel, BertLMPredictionHead, _EngineMixin, _check_unsupported
from .modeling_utils import BertPreTrainedModel


class NSPCPT(_EngineMixin, BertPreTrainedModel):
    _head = "nsp"

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertImgModel(config)
        self.cls = BertLMPredictionHead(config)          # as the reference: replaced by copy_from_pretraining_model
        self.num_seq_relations = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.apply(self.init_weights)
        self.tie_weights()

    def copy_from_pretraining_model(self, model, possible_colors=[]):
        """modeling_vcr.py:90-92."""
        self.bert = model.bert
        self.cls = model.cls.seq_relationship
        self.__dict__.pop("_cpt_engine", None)
        if "_cpt_engine" in model.__dict__:
            self._engine().dtype = model.__dict__["_cpt_engine"].dtype

    def tie_weights(self):
        """modeling_vcr.py:108-113 (a no-op once ``cls`` is the relation head)."""
        if hasattr(self.cls, "decoder"):
            self._tie_or_clone_weights(self.cls.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, next_sentence_label=None,
                position_ids=None, head_mask=None, img_feats=None):
        _check_unsupported(self.config, head_mask, None)
        if not isinstance(self.cls, nn.Linear):
            raise RuntimeError("cpt_amd: NSPCPT needs copy_from_pretraining_model() first (cls must be the "
                               "seq_relationship Linear, modeling_vcr.py:90-92)")
        if torch.is_grad_enabled() and next_sentence_label is not None and self.cls.weight.requires_grad:
            # few-shot fine-tuning (fewshot/vcr_nsp_cpt.py:425-470): loss with gradients through the HIP backward
            from .train import mlm_loss_with_grad
            return mlm_loss_with_grad(self, input_ids, token_type_ids, attention_mask, next_sentence_label.view(-1),
                                      position_ids, img_feats, None)
        out = self._engine().forward(input_ids, token_type_ids, attention_mask, position_ids, img_feats, flags=L.OUT_REL)
        rel = out["rel"]
        outputs = (rel,)
        if next_sentence_label is not None:
            # CrossEntropyLoss(ignore_index=-1) over (B, num_seq_relations), modeling_vcr.py:124-127
            acc = ops.ce_rows(rel.view(-1, self.num_seq_relations).contiguous(),
                              next_sentence_label.view(-1).to(torch.int64).contiguous())
            outputs = (acc[0] / acc[1],) + outputs
        return outputs
