"""BertConfig mirror for the CPT hot path.

Mirrors the attribute-bag behaviour of the third-party ``BertConfig`` the
reference reads (attributes consumed at
/root/reference/Oscar/oscar/modeling/modeling_bert.py:159-181 and set by callers at
Oscar/oscar/run_oscarplus_pretrain.py:238-249): ``config.json`` keys become
attributes, unknown keys are kept, ``save_pretrained`` dumps them back.
"""
import copy
import json
import os

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "pytorch_model.bin"


class BertConfig(object):
    _defaults = dict(
        vocab_size=30522, hidden_size=768, num_hidden_layers=12,
        num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        max_position_embeddings=512, type_vocab_size=2,
        initializer_range=0.02, layer_norm_eps=1e-12,
        output_attentions=False, output_hidden_states=False,
        torchscript=False, num_labels=2,
        # Oscar extras (run_oscarplus_pretrain.py:238-249)
        img_feature_dim=2054, img_feature_type="faster_r-cnn",
        use_img_layernorm=1, img_layer_norm_eps=1e-12, num_contrast_classes=3)

    def __init__(self, **kwargs):
        for k, v in self._defaults.items():
            setattr(self, k, v)
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls(**json.load(f))

    @classmethod
    def from_pretrained(cls, path, *args, **kwargs):
        kwargs.pop("cache_dir", None)
        return_unused = kwargs.pop("return_unused_kwargs", False)
        fn = os.path.join(path, CONFIG_NAME) if os.path.isdir(path) else path
        cfg = cls.from_json_file(fn)
        unused = {}
        for k, v in kwargs.items():
            if hasattr(cfg, k):
                setattr(cfg, k, v)
            else:
                unused[k] = v
        return (cfg, unused) if return_unused else cfg

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory):
        assert os.path.isdir(save_directory)
        with open(os.path.join(save_directory, CONFIG_NAME), "w", encoding="utf-8") as f:
            f.write(self.to_json_string())

    def __repr__(self):
        return "BertConfig " + self.to_json_string()


def oscar_base(**over):
    return BertConfig(**over)


def oscar_large(**over):
    d = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    d.update(over)
    return BertConfig(**d)


def tiny(**over):
    """Small shape used by parity tests and golden fixtures (head_dim stays 64)."""
    d = dict(vocab_size=518, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
             intermediate_size=512, max_position_embeddings=32, img_feature_dim=38)
    d.update(over)
    return BertConfig(**d)
