"""CPU oracle for the CPT [MASK]-scoring hot path (Oscar/BertImg forward/backward).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file, and only as the checker.  ``cpt_amd`` never imports it and has no CPU
fallback: without the HIP library the product path raises.

What it is: a plain fp32 CPU restatement (functional PyTorch on CPU tensors, no
``nn.Module``) of the reference algorithm, written from the reference call
sites cited on each function.  Paths are relative to /root/reference.

Parity pinning: the reference has NO tests, golden vectors or fixtures for this
path, and the block arithmetic lives in an un-vendored
third-party dependency: huggingface/transformers @ 067923d3267325f525f4e46f357360c191ba562e
(package ``pytorch_transformers``; install.sh:30-34).  The oracle is therefore
pinned by fixtures generated in the build container by ``oracle/make_golden.py``:
the reference's OWN files (Oscar/oscar/modeling/modeling_bert.py, modeling_rec.py,
modeling_utils.py) imported from /root/reference and executed on top of a
restatement of that dependency (``oracle/ref_stub.py``), with every restated
block cross-checked against the installed transformers==5.15 BERT modules.
The fixtures live in ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks
this file against all of them.
"""
import math

import torch
import torch.nn.functional as F

MASK_NEG = -10000.0


def gelu_erf(x):
    """third-party ``gelu`` (ACT2FN['gelu'], used by BertIntermediate and
    BertPredictionHeadTransform): exact erf form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):
    """BertLayerNorm == torch.nn.LayerNorm: biased variance, eps inside sqrt
    (ctor call sites: Oscar/oscar/modeling/modeling_bert.py:181)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def extended_mask(attention_mask, dtype=torch.float32):
    """Oscar/oscar/modeling/modeling_bert.py:213-226."""
    if attention_mask.dim() == 2:
        ext = attention_mask[:, None, None, :]
    elif attention_mask.dim() == 3:
        ext = attention_mask[:, None, :, :]
    else:
        raise NotImplementedError
    return (1.0 - ext.to(dtype)) * MASK_NEG


def text_embeddings(sd, cfg, input_ids, token_type_ids, position_ids=None, prefix="bert."):
    """BertEmbeddings.forward (third-party); call site modeling_bert.py:244-245."""
    B, Lt = input_ids.shape
    if position_ids is None:
        position_ids = torch.arange(Lt, dtype=torch.long).unsqueeze(0).expand(B, Lt)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    p = prefix + "embeddings."
    e = (sd[p + "word_embeddings.weight"][input_ids]
         + sd[p + "position_embeddings.weight"][position_ids]
         + sd[p + "token_type_embeddings.weight"][token_type_ids])
    return layer_norm(e, sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], cfg["layer_norm_eps"])


def image_embeddings(sd, cfg, img_feats, prefix="bert."):
    """modeling_bert.py:178,261-266 (dropout is identity in eval)."""
    y = F.linear(img_feats, sd[prefix + "img_embedding.weight"], sd[prefix + "img_embedding.bias"])
    if cfg.get("use_img_layernorm"):
        y = layer_norm(y, sd[prefix + "LayerNorm.weight"], sd[prefix + "LayerNorm.bias"],
                       cfg["img_layer_norm_eps"])
    return y


def _drop(x, drop, key):
    """nn.Dropout in training mode with a GIVEN mask: ``drop[key]`` is the multiplier tensor (0 or 1/(1-p)), broadcastable
    to x.  drop None / key absent = eval mode (identity)."""
    if drop is None or key not in drop:
        return x
    return x * drop[key].to(x.dtype).view(x.shape)


def self_attention(sd, cfg, x, ext_mask, p, drop=None, layer=0):
    """CaptionBertSelfAttention.forward, modeling_bert.py:30-70 (dropout on the probabilities: :57)."""
    B, L, H = x.shape
    nh = cfg["num_attention_heads"]
    d = H // nh
    q = F.linear(x, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(x, sd[p + "key.weight"], sd[p + "key.bias"])
    v = F.linear(x, sd[p + "value.weight"], sd[p + "value.bias"])
    q = q.view(B, L, nh, d).permute(0, 2, 1, 3)
    k = k.view(B, L, nh, d).permute(0, 2, 1, 3)
    v = v.view(B, L, nh, d).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    s = s + ext_mask
    pr = _drop(torch.softmax(s, dim=-1), drop, ("attn", layer))
    ctx = torch.matmul(pr, v)
    return ctx.permute(0, 2, 1, 3).contiguous().view(B, L, H)


def encoder_layer(sd, cfg, x, ext_mask, i, prefix="bert.", drop=None):
    """CaptionBertLayer.forward modeling_bert.py:139-147; CaptionBertAttention
    82-87; BertSelfOutput / BertIntermediate / BertOutput (third-party)."""
    p = "%sencoder.layer.%d." % (prefix, i)
    eps = cfg["layer_norm_eps"]
    ctx = self_attention(sd, cfg, x, ext_mask, p + "attention.self.", drop, i)
    a = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    a = layer_norm(_drop(a, drop, ("ao", i)) + x, sd[p + "attention.output.LayerNorm.weight"],
                   sd[p + "attention.output.LayerNorm.bias"], eps)
    h = gelu_erf(F.linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    o = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return layer_norm(_drop(o, drop, ("out", i)) + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def bert_img_forward(sd, cfg, input_ids, token_type_ids=None, attention_mask=None,
                     position_ids=None, img_feats=None, prefix="bert.", all_hidden=False, drop=None):
    """BertImgModel.forward, modeling_bert.py:199-279 -> (sequence_output, pooled_output[, hiddens])."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    ext = extended_mask(attention_mask)
    x = text_embeddings(sd, cfg, input_ids, token_type_ids, position_ids, prefix)
    if img_feats is not None:
        x = torch.cat((x, image_embeddings(sd, cfg, img_feats, prefix)), 1)   # :269
    x = _drop(x, drop, "emb")        # BertEmbeddings' dropout (text rows) and modeling_bert.py:266 (region rows)
    hiddens = [x]
    for i in range(cfg["num_hidden_layers"]):                                   # :100-126
        x = encoder_layer(sd, cfg, x, ext, i, prefix, drop)
        hiddens.append(x)
    pooled = torch.tanh(F.linear(x[:, 0], sd[prefix + "pooler.dense.weight"],
                                 sd[prefix + "pooler.dense.bias"]))             # :275
    if all_hidden:
        return x, pooled, hiddens
    return x, pooled


def lm_head(sd, cfg, x, prefix="cls."):
    """BertLMPredictionHead (third-party) as wired by REC_MLM_CPT
    (modeling_rec.py:104-105,130-135,143): decoder weight tied to word embeddings."""
    t = gelu_erf(F.linear(x, sd[prefix + "transform.dense.weight"], sd[prefix + "transform.dense.bias"]))
    t = layer_norm(t, sd[prefix + "transform.LayerNorm.weight"], sd[prefix + "transform.LayerNorm.bias"],
                   cfg["layer_norm_eps"])
    return F.linear(t, sd[prefix + "decoder.weight"]) + sd[prefix + "bias"]


def rec_mlm_cpt_forward(sd, cfg, input_ids, token_type_ids=None, attention_mask=None,
                        masked_lm_labels=None, position_ids=None, img_feats=None,
                        mask_rows_only=None, drop=None):
    """REC_MLM_CPT.forward, modeling_rec.py:137-152.

    ``mask_rows_only``: optional LongTensor (B,) of [MASK] positions; when given
    only those rows go through the head -> (B, V).  Every reference consumer
    keeps exactly those rows (zeroshot/refcoco_cpt.py:219, fewshot/refcoco_cpt.py:268).
    """
    seq, _ = bert_img_forward(sd, cfg, input_ids, token_type_ids, attention_mask,
                              position_ids, img_feats, drop=drop)
    if mask_rows_only is not None:
        rows = seq[torch.arange(seq.size(0)), mask_rows_only]
        scores = lm_head(sd, cfg, rows)
    else:
        scores = lm_head(sd, cfg, seq)
    out = (scores,)
    if masked_lm_labels is not None:
        V = cfg["vocab_size"]
        if mask_rows_only is not None:
            lab = masked_lm_labels[torch.arange(seq.size(0)), mask_rows_only]
            loss = F.cross_entropy(scores.view(-1, V), lab.view(-1), ignore_index=-1)
        else:
            loss = F.cross_entropy(scores.view(-1, V), masked_lm_labels.view(-1), ignore_index=-1)
        out = (loss,) + out
    return out


def nsp_cpt_scores(sd, cfg, input_ids, token_type_ids, attention_mask, img_feats):
    """NSPCPT scoring head (Oscar/oscar/modeling/modeling_vcr.py:79-129): pooled
    [CLS] -> pretrained cls.seq_relationship Linear(H,3)."""
    _, pooled = bert_img_forward(sd, cfg, input_ids, token_type_ids, attention_mask, None, img_feats)
    return F.linear(pooled, sd["cls.seq_relationship.weight"], sd["cls.seq_relationship.bias"])


def nsp_cpt_forward(sd, cfg, input_ids, token_type_ids, attention_mask, img_feats, next_sentence_label=None,
                    w_key="cls.seq_relationship.weight", b_key="cls.seq_relationship.bias"):
    """NSPCPT.forward (Oscar/oscar/modeling/modeling_vcr.py:115-129): relation scores on the pooled [CLS]
    and, with labels, CrossEntropyLoss(ignore_index=-1) over the num_contrast_classes scores."""
    _, pooled = bert_img_forward(sd, cfg, input_ids, token_type_ids, attention_mask, None, img_feats)
    rel = F.linear(pooled, sd[w_key], sd[b_key])
    if next_sentence_label is None:
        return (rel,)
    loss = F.cross_entropy(rel.view(-1, rel.size(-1)), next_sentence_label.view(-1), ignore_index=-1)
    return (loss, rel)


def nsp_choice_labels(labels, interval, n_seq):
    """fewshot/vcr_nsp_cpt.py:433-436: class 0 ("is next") for the correct answer of each question, 1 elsewhere."""
    cls_labels = torch.ones([n_seq], dtype=torch.long)
    for i, lb in enumerate(labels):
        cls_labels[i * interval + int(lb)] = 0
    return cls_labels


def nsp_choose(rel, interval):
    """fewshot/vcr_nsp_cpt.py:597-604: score = 1 - softmax(rel)[:, 1]; first-max argmax inside each question's
    `interval` answer choices.  Returns (choice scores, [pred per question])."""
    logits = 1 - (rel.softmax(-1)[:, 1].view(-1))
    n = rel.size(0) // interval
    return logits, [int(logits[q * interval:(q + 1) * interval].argmax()) for q in range(n)]


# ---- a15: score extraction (callers) ---------------------------------------

def select_region_zeroshot(mask_scores, color_id_sets, none_id):
    """zeroshot/refcoco_cpt.py:224-246: per query, gather colour logits of each
    proposal sequence (drop the trailing "none"), concat, first-max argmax."""
    collected = []
    for row, ids in zip(mask_scores, color_id_sets):
        cur = row[list(ids) + [none_id]]
        collected.append(cur[0:-1])
    collected = torch.cat(collected, -1)
    return int(collected.argmax()), collected


def select_region_fewshot(mask_scores, color_id_sets, none_id):
    """fewshot/refcoco_cpt.py:277-295: as above but colour logit / none logit."""
    collected = []
    for row, ids in zip(mask_scores, color_id_sets):
        cur = row[list(ids) + [none_id]]
        collected.append(cur[0:-1] / cur[-1])
    collected = torch.cat(collected, -1)
    return int(collected.argmax()), collected


def compute_iou(box1, box2):
    """Oscar/oscar/utils/iou.py:1-12, boxes [x, y, w, h], inclusive pixels."""
    ix1 = max(box1[0], box2[0]); iy1 = max(box1[1], box2[1])
    ix2 = min(box1[0] + box1[2] - 1, box2[0] + box2[2] - 1)
    iy2 = min(box1[1] + box1[3] - 1, box2[1] + box2[3] - 1)
    inter = (ix2 - ix1 + 1) * (iy2 - iy1 + 1) if (ix1 < ix2 and iy1 < iy2) else 0
    return float(inter) / (box1[2] * box1[3] + box2[2] * box2[3] - inter)


# ---- a14: schedule / optimizer (few-shot step) ------------------------------

def warmup_linear(step, warmup_step, tot_step):
    """Oscar/oscar/utils/optim_sched.py:16-20."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, learning_rate, warmup_steps, num_train_steps):
    """Oscar/oscar/utils/optim_sched.py:39-45 (floor 1e-8)."""
    lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    return 1e-8 if lr <= 0 else lr


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """torch.optim.AdamW single-tensor update (fewshot/refcoco_cpt.py:343,249):
    decoupled decay first, bias-corrected Adam, eps added after sqrt/bias2."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def adamw_step_hf(p, g, m, v, step, lr, beta1, beta2, eps, wd, correct_bias=True):
    """pytorch_transformers.AdamW single-tensor update, the optimizer of the GQA / VCR few-shot drivers (Oscar/oscar/fewshot/vcr_nsp_cpt.py:385,
    gqa_cpt.py:342).  Its source (transformers@067923d optimization.py) is NOT under /root/reference: restated from the published algorithm
     -- exp_avg.mul_(b1).add_(1 - b1, grad); exp_avg_sq.mul_(b2).addcmul_(1 - b2, grad, grad); denom = exp_avg_sq.sqrt() + eps;
    step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) when correct_bias; p.addcdiv_(-step_size, exp_avg, denom); then p.add_(-lr * wd, p).
    Parity unpinned beyond that restatement (no vendored source, no golden vector): tests compare the HIP kernel with THIS function."""
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    denom = v.sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p - step_size * (m / denom)
    if wd > 0.0:
        p = p - lr * wd * p
    return p, m, v


def warmup_linear_schedule(step, warmup_steps, t_total):
    """pytorch_transformers.WarmupLinearSchedule's multiplier (fewshot/vcr_nsp_cpt.py:386)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def train_step_grads(sd, cfg, batch, names=None, drop=None):
    """loss.backward() of fewshot/refcoco_cpt.py:231-248 with dropout disabled (or, with ``drop``, with the given masks):
    returns (loss, {name: grad}).  The tied decoder/word-embedding tensor gets the
    sum of both contributions, exactly as autograd does for the shared Parameter."""
    tied = sd["cls.decoder.weight"].data_ptr() == sd["bert.embeddings.word_embeddings.weight"].data_ptr()
    leaves = {}
    for k, t in sd.items():
        if k == "cls.decoder.weight" and tied:
            continue
        leaves[k] = t.detach().clone().requires_grad_(True)
    work = dict(leaves)
    if tied:
        work["cls.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    B = batch["input_ids"].size(0)
    labels = torch.full(batch["attention_mask"].shape, -1, dtype=torch.long)
    labels[torch.arange(B), batch["mask_token_pos"]] = batch["colors"]
    loss, _ = rec_mlm_cpt_forward(work, cfg, batch["input_ids"], batch["segment_ids"],
                                  batch["attention_mask"], masked_lm_labels=labels,
                                  img_feats=batch["img_feats"],
                                  mask_rows_only=batch["mask_token_pos"], drop=drop)
    loss.backward()
    grads = {k: (t.grad if t.grad is not None else None) for k, t in leaves.items()}
    return loss.detach(), grads


# ---- dropout masks: CPU restatement of the counter scheme of cpt_amd/csrc/dropout.h (to pin the exported masks) --------

def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on numpy uint32 arrays -> four uint32 arrays."""
    import numpy as np
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    k0 = np.uint64(k0 & 0xFFFFFFFF)
    k1 = np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & mask
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & mask
        c0, c1, c2, c3 = n0 & mask, n1, n2 & mask, n3
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def dropout_thresh_scale(p, attn):
    full = 65536.0 if attn else 4294967296.0
    t = min(max(p * full, 1.0), full - 1.0)
    thresh = int(t + 0.5)
    return thresh, 1.0 / (1.0 - thresh / full)


def dropout_keep_hidden(seed, step, site, R, H, p):
    """keep mask [R][H] (uint8) of a hidden dropout site: element e = r*H + c, one Philox call per 4 consecutive elements."""
    import numpy as np
    thresh, _ = dropout_thresh_scale(p, False)
    n4 = R * H // 4
    e4 = np.arange(n4, dtype=np.uint64)
    u = philox4x32_10(e4 & 0xFFFFFFFF, e4 >> np.uint64(32), np.full(n4, step), np.full(n4, site), seed & 0xFFFFFFFF, seed >> 32)
    keep = np.stack([w >= np.uint32(thresh) for w in u], 1).reshape(R, H)
    return keep.astype(np.uint8)


def dropout_keep_attn(seed, step, site, BH, L, p):
    """keep mask [BH][L][L] of an attention dropout site: 4 x 4 blocks of the (query, key) plane, two calls per block
    (query rows 0-1 / 2-3), eight 16-bit uniforms each, index (q & 1) * 4 + (k & 3)."""
    import numpy as np
    thresh, _ = dropout_thresh_scale(p, True)
    bh, q, k = np.meshgrid(np.arange(BH, dtype=np.uint64), np.arange(L, dtype=np.uint64), np.arange(L, dtype=np.uint64), indexing="ij")
    c0 = ((q >> np.uint64(2)) << np.uint64(16)) | ((k >> np.uint64(2)) << np.uint64(1)) | ((q >> np.uint64(1)) & np.uint64(1))
    u = philox4x32_10(c0.ravel(), bh.ravel(), np.full(c0.size, step), np.full(c0.size, site), seed & 0xFFFFFFFF, seed >> 32)
    idx = ((q & np.uint64(1)) * np.uint64(4) + (k & np.uint64(3))).ravel().astype(np.int64)
    words = np.stack(u, 1)                                         # [n][4]
    w = words[np.arange(idx.size), idx >> 1]
    u16 = (w >> ((idx & 1) * 16).astype(np.uint32)) & np.uint32(0xFFFF)
    return (u16 >= np.uint32(thresh)).reshape(BH, L, L).astype(np.uint8)
