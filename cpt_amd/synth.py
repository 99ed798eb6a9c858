"""Deterministic synthetic weights and RefCOCO-shaped CPT batches.

There is no checkpoint, vocab or dataset on the build/GPU boxes, so bench, smoke
and the parity tests use inputs of the shape the reference dataset produces
(/root/reference/Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:85-159,211-302:
text padded to 70 ids, regions padded to 50 x 2054, attention mask over both) and
random-init weights of the reference architecture (init per
Oscar/oscar/modeling/modeling_rec.py:116-128).  Everything is drawn from numpy's
PCG64 so the same seed gives the same bytes on every box.
"""
import numpy as np
import torch

CLS, SEP, MASK, PAD = 101, 102, 103, 0
# Placeholder colour-word ids (no vocab.txt on the boxes; real ids come from the
# checkpoint's vocab at run time).  bert-base-uncased: red, purple, green, yellow,
# blue, (none) -- fixed here so goldens are reproducible.
COLOR_IDS = (2417, 6379, 2665, 3756, 2630)
NONE_ID = 3904


def param_specs(cfg, head="cpt"):
    """[(state-dict key, shape, kind)] in the reference's registration order.

    Keys are the ones listed in SURVEY.md section 8(b) (verified there by
    instantiating the reference).  kind: 'w' Linear/Embedding weight, 'b' bias,
    'g' LayerNorm weight.  head: 'cpt' (REC_MLM_CPT, modeling_rec.py:100-109),
    'pretrain' (BertImgForPreTraining, modeling_bert.py:981-991), 'nsp' (NSPCPT, modeling_vcr.py:79-92), 'none'.
    """
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s = [("bert.embeddings.word_embeddings.weight", (V, H), "w"),
         ("bert.embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H), "w"),
         ("bert.embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H), "w"),
         ("bert.embeddings.LayerNorm.weight", (H,), "g"),
         ("bert.embeddings.LayerNorm.bias", (H,), "b")]
    for i in range(cfg.num_hidden_layers):
        p = "bert.encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            s += [(p + "attention.self.%s.weight" % nm, (H, H), "w"),
                  (p + "attention.self.%s.bias" % nm, (H,), "b")]
        s += [(p + "attention.output.dense.weight", (H, H), "w"),
              (p + "attention.output.dense.bias", (H,), "b"),
              (p + "attention.output.LayerNorm.weight", (H,), "g"),
              (p + "attention.output.LayerNorm.bias", (H,), "b"),
              (p + "intermediate.dense.weight", (I, H), "w"),
              (p + "intermediate.dense.bias", (I,), "b"),
              (p + "output.dense.weight", (H, I), "w"),
              (p + "output.dense.bias", (H,), "b"),
              (p + "output.LayerNorm.weight", (H,), "g"),
              (p + "output.LayerNorm.bias", (H,), "b")]
    s += [("bert.pooler.dense.weight", (H, H), "w"), ("bert.pooler.dense.bias", (H,), "b"),
          ("bert.img_embedding.weight", (H, cfg.img_feature_dim), "w"),
          ("bert.img_embedding.bias", (H,), "b")]
    if getattr(cfg, "use_img_layernorm", None):
        s += [("bert.LayerNorm.weight", (H,), "g"), ("bert.LayerNorm.bias", (H,), "b")]
    if head == "none":
        return s
    if head == "nsp":          # NSPCPT after copy_from_pretraining_model: cls IS the seq_relationship Linear (modeling_vcr.py:90-92)
        n = getattr(cfg, "num_contrast_classes", 2)
        return s + [("cls.weight", (n, H), "w"), ("cls.bias", (n,), "b")]
    hp = "cls." if head == "cpt" else "cls.predictions."
    s += [(hp + "bias", (V,), "b"),
          (hp + "transform.dense.weight", (H, H), "w"), (hp + "transform.dense.bias", (H,), "b"),
          (hp + "transform.LayerNorm.weight", (H,), "g"), (hp + "transform.LayerNorm.bias", (H,), "b"),
          (hp + "decoder.weight", (V, H), "tied")]
    if head == "pretrain":
        n = getattr(cfg, "num_contrast_classes", 2)
        s += [("cls.seq_relationship.weight", (n, H), "w"), ("cls.seq_relationship.bias", (n,), "b")]
    return s


def init_state_dict(cfg, seed=88, head="cpt", randomize_all=True):
    """name -> fp32 CPU tensor.  Weights ~ N(0, initializer_range).  With
    ``randomize_all`` biases and LayerNorm parameters are perturbed too (the
    reference init leaves them 0/1, which would hide a missing bias or gain in a
    parity test)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    std = cfg.initializer_range
    sd = {}
    for name, shape, kind in param_specs(cfg, head):
        if kind == "tied":
            sd[name] = sd["bert.embeddings.word_embeddings.weight"]
            continue
        if kind == "w":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif kind == "b":
            a = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
                 if randomize_all else np.zeros(shape, np.float32))
        else:
            a = (np.float32(1.0) + rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
                 if randomize_all else np.ones(shape, np.float32))
        sd[name] = torch.from_numpy(np.ascontiguousarray(a))
    return sd


def make_batch(B, cfg, seed=88, max_seq_len=70, img_seq_len=50, n_regions=None,
               vary_regions=False):
    """One flattened batch of B (query x proposal) sequences, as ``test_collate``
    (Oscar/oscar/zeroshot/refcoco_cpt.py:159-172) hands it to ``val``.

    Returns dict of CPU tensors: img_feats (B,img_seq_len,D) f32, input_ids /
    segment_ids (B,max_seq_len) i64, attention_mask (B,max_seq_len+img_seq_len) i64,
    mask_token_pos (B,) i64, colors (B,) i64 (a GT colour id per sequence, for the
    few-shot loss).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    D = cfg.img_feature_dim
    V = cfg.vocab_size
    hi = min(30000, V)
    lo = min(1000, max(4, V // 4))
    special = [t for t in (CLS, SEP, MASK) if t < V]
    cls_id, sep_id, mask_id = (CLS, SEP, MASK) if len(special) == 3 else (1, 2, 3)
    if n_regions is None:
        n_regions = img_seq_len
    img = np.zeros((B, img_seq_len, D), np.float32)
    ids = np.zeros((B, max_seq_len), np.int64)
    seg = np.zeros((B, max_seq_len), np.int64)
    att = np.zeros((B, max_seq_len + img_seq_len), np.int64)
    mpos = np.zeros((B,), np.int64)
    colors = np.zeros((B,), np.int64)
    color_pool = [c for c in COLOR_IDS if c < V] or [5, 6, 7, 8, 9]
    for b in range(B):
        nr = n_regions
        if vary_regions:
            nr = int(rng.integers(max(1, n_regions // 2), n_regions + 1))
        nbox = D - 2048 if D > 2048 else min(6, D)
        nfeat = D - nbox
        f = np.maximum(rng.standard_normal((nr, nfeat), dtype=np.float32), 0)
        img[b, :nr, :nfeat] = f
        if nbox >= 6:
            x1 = rng.random(nr, dtype=np.float32) * 0.7
            y1 = rng.random(nr, dtype=np.float32) * 0.7
            w = rng.random(nr, dtype=np.float32) * (1 - x1 - 0.05) + 0.05
            h = rng.random(nr, dtype=np.float32) * (1 - y1 - 0.05) + 0.05
            img[b, :nr, nfeat:nfeat + 6] = np.stack([x1, y1, x1 + w, y1 + h, w, h], 1)
        # text a: [CLS] caption is in [MASK] color . [SEP]; text b: od labels + one colour word [SEP]
        cap = rng.integers(lo, hi, size=min(int(rng.integers(2, 11)), max(1, max_seq_len - 10)))
        tail = rng.integers(lo, hi, size=4)          # "is in", "color", "."
        a = [cls_id] + list(cap) + [int(tail[0]), int(tail[1]), mask_id, int(tail[2]), int(tail[3]), sep_id]
        n_b = max(1, min(nr + 1, max_seq_len - len(a) - 1))
        bt = list(rng.integers(lo, hi, size=n_b))
        bt[int(rng.integers(0, n_b))] = int(color_pool[int(rng.integers(0, len(color_pool)))])
        toks = a + bt + [sep_id]
        ids[b, :len(toks)] = toks
        seg[b, len(a):len(toks)] = 1
        att[b, :len(toks)] = 1
        att[b, max_seq_len:max_seq_len + nr] = 1
        mpos[b] = a.index(mask_id)
        colors[b] = int(color_pool[int(rng.integers(0, len(color_pool)))])
    return dict(img_feats=torch.from_numpy(img), input_ids=torch.from_numpy(ids),
                segment_ids=torch.from_numpy(seg), attention_mask=torch.from_numpy(att),
                mask_token_pos=torch.from_numpy(mpos), colors=torch.from_numpy(colors))


def make_prediction_rows(n_rows, proposals=8, boxes=50, seed=0, dim=2054):
    """Synthetic rows of a RefCOCO ``predictions.tsv`` in the reference's wire format (written by
    ``inference_ref.py:157-191``, read by ``refcoco_zsl_cpt_dataset.py:161-180``): per row ``proposals`` painted copies of an image, each
    a list of ``boxes`` detections whose ``feature`` is the base64 text of float32[dim] (non-negative like post-ReLU pooled CNN features).
    Returns the rows' JSON payloads as bytes (about 4.4 MB each at 8 x 50); every row carries different features."""
    import base64
    import json
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n_rows):
        feats = np.maximum(rng.standard_normal((proposals, boxes, dim), dtype=np.float32), 0)
        objs = [[{"rect": [1.0, 2.0, 30.0, 40.0], "bbox_id": j, "class": "dog", "conf": 0.9, "feature": base64.b64encode(feats[p, j].tobytes()).decode()}
                 for j in range(boxes)] for p in range(proposals)]
        rows.append(json.dumps({"objects": [objs, "a dog on the left", [["red"]] * proposals, [[[1, 2, 30, 40]]] * proposals]}).encode())
    return rows


def write_predictions_tsv(path, n_rows, proposals=8, boxes=50, seed=0):
    """``make_prediction_rows`` as ``key \\t json`` lines plus the ``.lineidx`` companion the readers seek through."""
    import os
    from cpt_amd import io
    with open(path, "wb") as f:
        for i, r in enumerate(make_prediction_rows(n_rows, proposals, boxes, seed)):
            f.write(b"img_%d\t" % i + r + b"\n")
    io.generate_lineidx_file(path, os.path.splitext(path)[0] + ".lineidx")
    return path
