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

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        kwargs.pop("cache_dir", None)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints are not supported")
        output_loading_info = kwargs.pop("output_loading_info", False)
        if config is None:
            config, model_kwargs = cls.config_class.from_pretrained(
                pretrained_model_name_or_path, return_unused_kwargs=True, **kwargs)
        else:
            model_kwargs = kwargs
        if os.path.isdir(pretrained_model_name_or_path):
            archive_file = os.path.join(pretrained_model_name_or_path, WEIGHTS_NAME)
        else:
            archive_file = pretrained_model_name_or_path
        model = cls(config, *model_args, **model_kwargs)
        if state_dict is None:
            state_dict = torch.load(archive_file, map_location="cpu")
        state_dict = dict(state_dict)
        for key in list(state_dict.keys()):
            new_key = None
            if "gamma" in key:
                new_key = key.replace("gamma", "weight")
            if "beta" in key:
                new_key = key.replace("beta", "bias")
            if new_key:
                state_dict[new_key] = state_dict.pop(key)

        missing_keys, unexpected_keys, error_msgs = [], [], []

        def load(module, prefix=""):
            module._load_from_state_dict(state_dict, prefix, {}, True, missing_keys, unexpected_keys, error_msgs)
            for name, child in module._modules.items():
                if child is not None:
                    load(child, prefix + name + ".")

        start_prefix = ""
        model_to_load = model
        has_prefix = any(s.startswith(cls.base_model_prefix) for s in state_dict.keys())
        if not hasattr(model, cls.base_model_prefix) and has_prefix:
            start_prefix = cls.base_model_prefix + "."
        if hasattr(model, cls.base_model_prefix) and not has_prefix:
            model_to_load = getattr(model, cls.base_model_prefix)
        load(model_to_load, prefix=start_prefix)
        if missing_keys:
            logger.info("Weights of %s not initialized from pretrained model: %s", model.__class__.__name__, missing_keys)
        if unexpected_keys:
            logger.info("Weights from pretrained model not used in %s: %s", model.__class__.__name__, unexpected_keys)
        if len(error_msgs) == 2 and "size mismatch for cls.seq_relationship.weight" in error_msgs[0]:
            logger.info("Error(s) in loading state_dict for %s:\n\t%s", model.__class__.__name__, "\n\t".join(error_msgs))
        elif error_msgs:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(
                model.__class__.__name__, "\n\t".join(error_msgs)))
        if hasattr(model, "tie_weights"):
            model.tie_weights()
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": missing_keys, "unexpected_keys": unexpected_keys, "error_msgs": error_msgs}
        return model


ImgPreTrainedModel = PreTrainedModel
BertPreTrainedModel = PreTrainedModel
