"""Checkpoint surface: HF-style ``from_pretrained`` / ``save_pretrained``.

Same observable behaviour as the reference loader
(/root/reference/Oscar/oscar/modeling/modeling_utils.py:689-875): reads ``config.json`` +
``pytorch_model.bin`` (a plain state dict, ``torch.load(map_location='cpu')``), renames legacy
``gamma``/``beta`` keys to ``weight``/``bias`` (:811-823), loads with or without the ``bert.``
prefix (:843-851), tolerates only the ``cls.seq_relationship`` size mismatch (:858-860), re-ties
the decoder to the word embeddings (:865-866) and returns the model in eval mode (:869).
``save_pretrained`` writes the layout ``utils/save_model.py:4-12`` produces.
"""
import logging
import os

import torch
from torch import nn

from .config import BertConfig, WEIGHTS_NAME

logger = logging.getLogger(__name__)


class PreTrainedModel(nn.Module):
    config_class = BertConfig
    base_model_prefix = "bert"

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    # weight init of modeling_rec.py:116-128 / modeling_bert.py:991-1003
    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def _tie_or_clone_weights(self, first_module, second_module):
        if getattr(self.config, "torchscript", False):
            first_module.weight = nn.Parameter(second_module.weight.clone())
        else:
            first_module.weight = second_module.weight

    def tie_weights(self):
        pass

    def save_pretrained(self, save_directory):
        assert os.path.isdir(save_directory), "save_pretrained needs an existing directory"
        model_to_save = self.module if hasattr(self, "module") else self
        model_to_save.config.save_pretrained(save_directory)
        sd = {k: v.detach().to("cpu").clone() for k, v in model_to_save.state_dict().items()}
        torch.save(sd, os.path.join(save_directory, WEIGHTS_NAME))

    # ---- checkpoint loading ---------------------------------------------------------------------------------------
    # One pass over a name -> tensor table instead of nn.Module's recursive loader: the tensors this model owns (parameters and
    # buffers under their state-dict names; tied weights appear under every name they have, all aliasing one storage, which in
    # a packed model is a slice of the flat buffer) are looked up in the checkpoint after its keys have been normalised, and
    # copied in place.  Observable behaviour follows the reference loader (Oscar/oscar/modeling/modeling_utils.py:689-875).
    _TOLERATED_MISMATCH = "cls.seq_relationship."     # a 2-class checkpoint head under a 3-class config (or the reverse) is re-initialised, :858-860

    @classmethod
    def _checkpoint_table(cls, raw, model):
        """Checkpoint keys -> this model's names: legacy LayerNorm names (``gamma`` / ``beta``, :811-823) become ``weight`` /
        ``bias``; a checkpoint of the bare encoder loads into ``model.bert`` and a prefixed checkpoint into a bare encoder
        (:843-851).  Returns (table, names of the model this checkpoint is expected to fill)."""
        table = {}
        for key, tensor in raw.items():
            # the reference's rule (:811-823): ONE substitution on the original key -- `beta` -> `bias` when the key holds "beta",
            # else `gamma` -> `weight`
            name = key.replace("beta", "bias") if "beta" in key else (key.replace("gamma", "weight") if "gamma" in key else key)
            table[name] = tensor
        prefix = cls.base_model_prefix + "."
        prefixed = any(k.startswith(cls.base_model_prefix) for k in table)
        own = list(model.state_dict(keep_vars=True))
        if hasattr(model, cls.base_model_prefix) and not prefixed:
            table = {prefix + k: t for k, t in table.items()}
            own = [n for n in own if n.startswith(prefix)]
        elif not hasattr(model, cls.base_model_prefix) and prefixed:
            table = {(k[len(prefix):] if k.startswith(prefix) else "\0" + k): t for k, t in table.items()}   # head keys cannot match
        return table, own

    def adopt_state_dict(self, raw):
        """Copies a checkpoint's tensors into this model in place; returns {missing_keys, unexpected_keys, error_msgs}."""
        table, own = self._checkpoint_table(raw, self)
        # Tied weights (cls.predictions.decoder.weight = bert.embeddings.word_embeddings.weight) appear under both names and alias one
        # storage; names are visited in state-dict order, so when a checkpoint holds two different tensors for them the LATER name
        # (the decoder) is what remains -- as in the reference, whose constructors tie before loading (modeling_bert.py:980,
        # modeling_rec.py:109) and whose per-module loader visits `bert` before `cls`.
        mine = self.state_dict(keep_vars=True)
        missing, errors = [], []
        with torch.no_grad():
            for name in own:
                src = table.get(name)
                if src is None:
                    missing.append(name)
                elif tuple(src.shape) != tuple(mine[name].shape):
                    errors.append("size mismatch for {}: copying a param with shape {} from checkpoint, the shape in current model is {}."
                                  .format(name, tuple(src.shape), tuple(mine[name].shape)))
                else:
                    mine[name].copy_(src)
        wanted = set(own)
        unexpected = [k.lstrip("\0") for k in table if k not in wanted]
        return {"missing_keys": missing, "unexpected_keys": unexpected, "error_msgs": errors}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        kwargs.pop("cache_dir", None)
        want_info = kwargs.pop("output_loading_info", False)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints are not supported")
        if config is None:
            config, kwargs = cls.config_class.from_pretrained(pretrained_model_name_or_path, return_unused_kwargs=True, **kwargs)
        model = cls(config, *model_args, **kwargs)
        if state_dict is None:
            path = pretrained_model_name_or_path
            state_dict = torch.load(os.path.join(path, WEIGHTS_NAME) if os.path.isdir(path) else path, map_location="cpu")
        info = model.adopt_state_dict(state_dict)
        who = model.__class__.__name__
        if info["missing_keys"]:
            logger.info("Weights of %s not initialized from pretrained model: %s", who, info["missing_keys"])
        if info["unexpected_keys"]:
            logger.info("Weights from pretrained model not used in %s: %s", who, info["unexpected_keys"])
        if info["error_msgs"]:
            text = "Error(s) in loading state_dict for {}:\n\t{}".format(who, "\n\t".join(info["error_msgs"]))
            if all(m.startswith("size mismatch for " + cls._TOLERATED_MISMATCH) for m in info["error_msgs"]):
                logger.info(text)
            else:
                raise RuntimeError(text)
        model.tie_weights()
        model.eval()
        return (model, info) if want_info else model


ImgPreTrainedModel = PreTrainedModel
BertPreTrainedModel = PreTrainedModel
