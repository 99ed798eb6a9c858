"""REC_MLM_CPT: the CPT colour-prompt scoring wrapper, on the HIP library.

Drop-in for /root/reference/Oscar/oscar/modeling/modeling_rec.py:100-152.  Same constructor,
``copy_from_pretraining_model``, ``tie_weights``, forward argument order and state-dict keys
(``bert.*``, ``cls.{bias,transform.*,decoder.weight}``).

One extension: ``mask_token_pos``.  The reference pushes all L positions through the vocabulary
head (B x L x 30522 fp32 = 938 MB at B=64) and every caller then keeps only the [MASK] row
(zeroshot/refcoco_cpt.py:219, fewshot/refcoco_cpt.py:268, gqa_cpt.py:598-600).  Passing
``mask_token_pos`` (B,) returns exactly those rows, (B, V); without it the full
(B, L, V) tensor is produced as in the reference.
"""
import torch
from torch import nn

from . import _lib as L
from .modeling_bert import (BertImgModel, BertLMPredictionHead, _EngineMixin, _check_unsupported)
from .modeling_utils import BertPreTrainedModel


class REC_MLM_CPT(_EngineMixin, BertPreTrainedModel):
    _head = "cpt"

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertImgModel(config)
        self.cls = BertLMPredictionHead(config)
        self.num_seq_relations = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.apply(self.init_weights)
        self.tie_weights()

    def copy_from_pretraining_model(self, model, possible_colors=[]):
        """modeling_rec.py:111-114."""
        self.bert = model.bert
        self.cls = model.cls.predictions
        self.tie_weights()
        self.__dict__.pop("_cpt_engine", None)
        if "_cpt_engine" in model.__dict__:
            self._engine().dtype = model.__dict__["_cpt_engine"].dtype

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, masked_lm_labels=None,
                position_ids=None, head_mask=None, img_feats=None, mask_token_pos=None, vocab_columns=None):
        # vocab_columns (extension, inference with mask_token_pos only): 1-D tensor of vocabulary ids -- the returned scores are (B, len(vocab_columns)),
        # those columns of the (B, V) prediction scores in list order (the drivers read colour-token columns only: zeroshot/refcoco_cpt.py:219)
        _check_unsupported(self.config, head_mask, None)
        rows = mask_token_pos is not None
        flags = L.OUT_MASK_LOGITS if rows else L.OUT_ALL_LOGITS
        labels = None
        if masked_lm_labels is not None:
            flags |= L.OUT_LOSS
            labels = masked_lm_labels
            if rows and labels.dim() == 2:       # (B, L) label grid of fewshot/refcoco_cpt.py:231-233
                labels = labels[torch.arange(labels.size(0), device=labels.device), mask_token_pos]
        if vocab_columns is not None and (not rows or masked_lm_labels is not None):
            raise NotImplementedError("cpt_amd: vocab_columns goes with mask_token_pos and without masked_lm_labels")
        if torch.is_grad_enabled() and masked_lm_labels is not None and self.bert.img_embedding.weight.requires_grad:
            from .train import mlm_loss_with_grad
            if mask_token_pos is None:
                # reference call form: a (B, L) grid with -1 everywhere except the [MASK] slot
                # (fewshot/refcoco_cpt.py:231-233, 245-247) -> recover the slot per row
                grid = masked_lm_labels != -1
                if not bool((grid.sum(1) == 1).all()):
                    # any label grid (modeling_rec.py:147-150 takes whatever the (B, L) tensor holds, e.g. several masked words
                    # per sequence): the head runs on the labelled positions, in row-major order of the grid
                    idx = grid.nonzero()
                    if idx.size(0) == 0:
                        raise ValueError("cpt_amd: masked_lm_labels holds no labelled position (the reference's loss would be NaN)")
                    seq, pos = idx[:, 0].contiguous(), idx[:, 1].contiguous()
                    return mlm_loss_with_grad(self, input_ids, token_type_ids, attention_mask, masked_lm_labels[seq, pos], position_ids,
                                              img_feats, pos, row_seq=seq)
                mask_token_pos = grid.long().argmax(1)
                labels = masked_lm_labels[torch.arange(grid.size(0), device=grid.device), mask_token_pos]
            return mlm_loss_with_grad(self, input_ids, token_type_ids, attention_mask, labels, position_ids,
                                      img_feats, mask_token_pos)
        out = self._engine().forward(input_ids, token_type_ids, attention_mask, position_ids, img_feats,
                                     mask_pos=mask_token_pos, labels=labels, flags=flags, logit_cols=vocab_columns)
        outputs = (out["logits"],)
        if masked_lm_labels is not None:
            outputs = (out["loss"],) + outputs
        return outputs
