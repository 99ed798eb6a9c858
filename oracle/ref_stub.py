"""Stand-in for the un-vendored third-party module the reference imports.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_golden.py`` in the build
container to import the reference's own files
(``/root/reference/Oscar/oscar/modeling/{modeling_bert,modeling_rec,modeling_utils}.py``)
so that golden vectors can be generated from them.  Never imported by the
product path (``cpt_amd``), by ``bench.py`` or by the GPU tests.

Why it exists: the reference's block arithmetic (embeddings, LayerNorm, GELU,
the Linear blocks, the MLM head) lives in ``huggingface/transformers`` at
commit 067923d3267325f525f4e46f357360c191ba562e (package
``pytorch_transformers``), pinned by ``/root/reference/install.sh:30-34`` and
imported at ``Oscar/oscar/modeling/modeling_bert.py:10-16``,
``modeling_rec.py:11-17`` and ``modeling_utils.py:10-15``.  That source is not
under /root/reference and cannot be fetched (no network).  The classes below
restate its published semantics; ``make_golden.py``
cross-checks every block against the installed ``transformers==5.15``
``models.bert.modeling_bert`` modules with copied weights so the restatement is
not self-referential.

Nothing here is copied from /root/reference or from the third-party package.
"""
import copy
import json
import math
import os
import sys
import types

import torch
from torch import nn


class BertConfig(object):
    """Attribute bag with BERT-base defaults; unknown keys become attributes."""

    _defaults = dict(
        vocab_size=30522, hidden_size=768, num_hidden_layers=12,
        num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        max_position_embeddings=512, type_vocab_size=2,
        initializer_range=0.02, layer_norm_eps=1e-12,
        output_attentions=False, output_hidden_states=False,
        torchscript=False, num_labels=2)

    def __init__(self, **kwargs):
        for k, v in self._defaults.items():
            setattr(self, k, v)
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path, *args, **kwargs):
        kwargs.pop("cache_dir", None)
        ret_unused = kwargs.pop("return_unused_kwargs", False)
        fn = os.path.join(path, "config.json") if os.path.isdir(path) else path
        with open(fn, "r", encoding="utf-8") as f:
            cfg = cls(**json.load(f))
        unused = {}
        for k, v in kwargs.items():
            if hasattr(cfg, k):
                setattr(cfg, k, v)
            else:
                unused[k] = v
        return (cfg, unused) if ret_unused else cfg

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def save_pretrained(self, d):
        with open(os.path.join(d, "config.json"), "w", encoding="utf-8") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)


def gelu(x):
    # exact erf form
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


ACT2FN = {"gelu": gelu, "relu": torch.relu}

BertLayerNorm = nn.LayerNorm


class PreTrainedModel(nn.Module):
    config_class = BertConfig
    base_model_prefix = ""
    pretrained_model_archive_map = {}

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def _tie_or_clone_weights(self, first_module, second_module):
        if getattr(self.config, "torchscript", False):
            first_module.weight = nn.Parameter(second_module.weight.clone())
        else:
            first_module.weight = second_module.weight

    def save_pretrained(self, save_directory):
        model_to_save = self.module if hasattr(self, "module") else self
        model_to_save.config.save_pretrained(save_directory)
        torch.save(model_to_save.state_dict(),
                   os.path.join(save_directory, "pytorch_model.bin"))


class BertPreTrainedModel(PreTrainedModel):
    config_class = BertConfig
    base_model_prefix = "bert"

    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        seq_length = input_ids.size(1)
        if position_ids is None:
            position_ids = torch.arange(seq_length, dtype=torch.long, device=input_ids.device)
            position_ids = position_ids.unsqueeze(0).expand_as(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        e = (self.word_embeddings(input_ids) + self.position_embeddings(position_ids)
             + self.token_type_embeddings(token_type_ids))
        return self.dropout(self.LayerNorm(e))


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def transpose_for_scores(self, x):
        x = x.view(*(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)))
        return x.permute(0, 2, 1, 3)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.intermediate_act_fn = ACT2FN[config.hidden_act] if isinstance(config.hidden_act, str) else config.hidden_act

    def forward(self, hidden_states):
        return self.intermediate_act_fn(self.dense(hidden_states))


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return self.activation(self.dense(hidden_states[:, 0]))


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.transform_act_fn = ACT2FN[config.hidden_act] if isinstance(config.hidden_act, str) else config.hidden_act
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        return self.LayerNorm(self.transform_act_fn(self.dense(hidden_states)))


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))

    def forward(self, hidden_states):
        return self.decoder(self.transform(hidden_states)) + self.bias


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


def install():
    """Register the stand-in under the module names the reference imports."""
    import transformers  # real 5.x package: only serves as the parent namespace

    pkg = types.ModuleType("transformers.pytorch_transformers")
    mb = types.ModuleType("transformers.pytorch_transformers.modeling_bert")
    mu = types.ModuleType("transformers.pytorch_transformers.modeling_utils")
    fu = types.ModuleType("transformers.pytorch_transformers.file_utils")
    for name in ("BertEmbeddings", "BertSelfAttention", "BertAttention", "BertEncoder",
                 "BertLayer", "BertSelfOutput", "BertIntermediate", "BertOutput",
                 "BertPooler", "BertLayerNorm", "BertPreTrainedModel",
                 "BertPredictionHeadTransform", "BertOnlyMLMHead",
                 "BertLMPredictionHead", "BertConfig"):
        setattr(mb, name, globals()[name])
    mb.BERT_PRETRAINED_MODEL_ARCHIVE_MAP = {}
    mb.load_tf_weights_in_bert = None
    mu.PreTrainedModel = PreTrainedModel
    mu.WEIGHTS_NAME = "pytorch_model.bin"
    mu.TF_WEIGHTS_NAME = "model.ckpt"
    fu.cached_path = lambda p, cache_dir=None: p
    pkg.BertConfig = BertConfig
    pkg.modeling_bert, pkg.modeling_utils, pkg.file_utils = mb, mu, fu
    sys.modules["transformers.pytorch_transformers"] = pkg
    sys.modules["transformers.pytorch_transformers.modeling_bert"] = mb
    sys.modules["transformers.pytorch_transformers.modeling_utils"] = mu
    sys.modules["transformers.pytorch_transformers.file_utils"] = fu
    sys.modules.setdefault("anytree", types.ModuleType("anytree"))
    transformers.pytorch_transformers = pkg
    return pkg
