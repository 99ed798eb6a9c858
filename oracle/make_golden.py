"""Generate tests/golden/* from the reference itself.  BUILD-CONTAINER ONLY.

Imports the reference's own model files from /root/reference (read-only) on top
of ``oracle/ref_stub.py`` (restatement of the un-vendored
transformers@067923d ``pytorch_transformers`` blocks), runs them on CPU/fp32 on
seeded inputs, cross-checks the restated blocks against the installed
transformers==5.15 BERT modules, and writes inputs + expected outputs (data
only) under tests/golden/.  /root/reference does not exist on the GPU box, so
nothing at test/bench time runs this script.

    python oracle/make_golden.py            # writes tests/golden/*
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(ROOT, "tests", "golden")

import ref_stub  # noqa: E402

ref_stub.install()
sys.path.insert(0, "/root/reference/Oscar")
from oscar.modeling.modeling_bert import BertImgModel, BertImgForPreTraining  # noqa: E402
from oscar.modeling.modeling_rec import REC_MLM_CPT  # noqa: E402
from oscar.utils.optim_sched import get_lr_sched  # noqa: E402
from oscar.utils.iou import computeIoU  # noqa: E402

from cpt_amd import config as cfgmod  # noqa: E402
from cpt_amd import synth  # noqa: E402


def to_ref_cfg(cfg, **over):
    d = cfg.to_dict()
    d.update(over)
    return ref_stub.BertConfig(**d)


def build_ref(cfg, seed, **over):
    """reference BertImgForPreTraining + REC_MLM_CPT carrying synth weights
    (load path of zeroshot/refcoco_cpt.py:444-447)."""
    rc = to_ref_cfg(cfg, **over)
    pre = BertImgForPreTraining(rc)
    sd = synth.init_state_dict(cfg, seed, head="pretrain")
    missing, unexpected = pre.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    pre.tie_weights()
    m = REC_MLM_CPT(rc)
    m.copy_from_pretraining_model(pre)
    m.eval()
    return m, pre


def labels_for(batch):
    B = batch["input_ids"].size(0)
    lab = torch.full(batch["attention_mask"].shape, -1, dtype=torch.long)
    lab[torch.arange(B), batch["mask_token_pos"]] = batch["colors"]     # fewshot/refcoco_cpt.py:231-233
    return lab


def hf_crosscheck(cfg, seed):
    """Restated third-party blocks vs installed transformers 5.x modules, same weights."""
    from transformers import BertConfig as HFConfig
    from transformers.models.bert import modeling_bert as hf
    hc = HFConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                  num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                  intermediate_size=cfg.intermediate_size, hidden_dropout_prob=0.0,
                  attention_probs_dropout_prob=0.0, max_position_embeddings=cfg.max_position_embeddings,
                  type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu")
    hc._attn_implementation = "eager"
    m, _ = build_ref(cfg, seed)
    res = {}
    torch.manual_seed(0)
    B, L, H = 2, 12, cfg.hidden_size
    x = torch.randn(B, L, H)
    am = torch.ones(B, L)
    am[0, -3:] = 0
    ext = (1.0 - am[:, None, None, :]) * -10000.0
    with torch.no_grad():
        # encoder layer
        ref_layer = m.bert.encoder.layer[0]
        hl = hf.BertLayer(hc).eval()
        hl.load_state_dict(ref_layer.state_dict())
        hf_out = hl(x, attention_mask=ext)
        hf_out = hf_out[0] if isinstance(hf_out, tuple) else hf_out
        res["BertLayer"] = float((ref_layer(x, ext)[0] - hf_out).abs().max())
        # embeddings
        he = hf.BertEmbeddings(hc).eval()
        he.load_state_dict(m.bert.embeddings.state_dict(), strict=False)
        ids = torch.randint(1, cfg.vocab_size, (B, 7))
        tt = torch.randint(0, 2, (B, 7))
        res["BertEmbeddings"] = float((m.bert.embeddings(ids, token_type_ids=tt)
                                       - he(input_ids=ids, token_type_ids=tt)).abs().max())
        # LM head
        hh = hf.BertLMPredictionHead(hc).eval()
        sdh = dict(m.cls.state_dict())
        sdh["decoder.bias"] = sdh["bias"]
        hh.load_state_dict(sdh, strict=False)
        res["BertLMPredictionHead"] = float((m.cls(x) - hh(x)).abs().max())
        # pooler
        hp = hf.BertPooler(hc).eval()
        hp.load_state_dict(m.bert.pooler.state_dict())
        res["BertPooler"] = float((m.bert.pooler(x) - hp(x)).abs().max())
    return res


def tiny_case():
    cfg = cfgmod.tiny()
    seed = 1234
    m, pre = build_ref(cfg, seed, output_hidden_states=True)
    batch = synth.make_batch(3, cfg, seed=7, max_seq_len=20, img_seq_len=6, n_regions=6, vary_regions=True)
    lab = labels_for(batch)
    out = m(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
            img_feats=batch["img_feats"], masked_lm_labels=lab)
    loss, scores, hiddens = out[0], out[1], out[2]
    m.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
    with torch.no_grad():
        seq, pooled = m.bert(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
                             img_feats=batch["img_feats"])[:2]
        nsp = pre.cls.seq_relationship(pooled)
    g = {"in_" + k: v.numpy() for k, v in batch.items()}
    g["loss"] = loss.detach().numpy()
    g["scores"] = scores.detach().numpy()
    g["pooled"] = pooled.numpy()
    g["nsp_scores"] = nsp.numpy()
    for i, h in enumerate(hiddens):
        g["hidden_%d" % i] = h.detach().numpy()
    for k, v in grads.items():
        g["grad_" + k] = v
    np.savez_compressed(os.path.join(OUT, "tiny_fwd_bwd.npz"), **g)

    # --- 3-step few-shot trace: AdamW groups of fewshot/refcoco_cpt.py:318-343, LR 236-243
    m2, _ = build_ref(cfg, seed)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = [(n, p) for n, p in m2.named_parameters() if "classifier" not in n]
    lr0, wd, betas = 3e-5 * 100, 0.01, (0.9, 0.98)      # lr scaled up so 3 steps move the loss visibly
    groups = [{"params": [], "lr": lr0, "weight_decay": wd}, {"params": [], "lr": lr0, "weight_decay": 0.0},
              {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": wd},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    groups = [gr for gr in groups if len(gr["params"])]
    opt = torch.optim.AdamW(groups, lr=lr0, betas=betas)

    class O(object):
        learning_rate = lr0
        warmup_steps = 1
        num_train_steps = 3
    losses, lrs = [], []
    for step in range(3):
        lr = get_lr_sched(step, O)
        for gr in opt.param_groups:
            gr["lr"] = lr
        opt.zero_grad()
        ls = m2(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
                img_feats=batch["img_feats"], masked_lm_labels=lab)[0]
        ls.backward()
        opt.step()
        losses.append(float(ls))
        lrs.append(lr)
    after = {k: v.detach().numpy().copy() for k, v in m2.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "tiny_train3.npz"), losses=np.array(losses, np.float64),
                        lrs=np.array(lrs, np.float64), lr0=lr0, wd=wd, beta1=betas[0], beta2=betas[1],
                        **{"after_" + k: v for k, v in after.items() if k in (
                            "bert.encoder.layer.1.output.dense.weight", "bert.embeddings.word_embeddings.weight",
                            "bert.encoder.layer.0.attention.self.query.bias", "cls.transform.LayerNorm.weight",
                            "bert.img_embedding.weight", "cls.bias")})

    # --- checkpoint surface: legacy gamma/beta names, loaded by the reference's own from_pretrained
    ck = os.path.join(OUT, "tiny_ckpt")
    os.makedirs(ck, exist_ok=True)
    sd = synth.init_state_dict(cfg, seed, head="pretrain")
    legacy = {}
    for k, v in sd.items():
        if "LayerNorm.weight" in k:
            k = k.replace("LayerNorm.weight", "LayerNorm.gamma")
        elif "LayerNorm.bias" in k:
            k = k.replace("LayerNorm.bias", "LayerNorm.beta")
        legacy[k] = v.clone()
    torch.save(legacy, os.path.join(ck, "pytorch_model.bin"))
    cfg.save_pretrained(ck)
    rc = ref_stub.BertConfig.from_pretrained(ck)
    pre2 = BertImgForPreTraining.from_pretrained(ck, config=rc)        # modeling_utils.py:689-875
    m3 = REC_MLM_CPT(rc)
    m3.copy_from_pretraining_model(pre2)
    m3.eval()
    with torch.no_grad():
        sc3 = m3(batch["input_ids"], batch["segment_ids"], batch["attention_mask"], img_feats=batch["img_feats"])[0]
    np.savez_compressed(os.path.join(OUT, "tiny_ckpt_expected.npz"), scores=sc3.numpy(),
                        keys=np.array(sorted(m3.state_dict().keys())))
    return float((sc3 - scores.detach()).abs().max())


def base_case(name, B, n_regions, seed_w=88, seed_b=88, with_grads=False, vary=False, Lt=70, Li=50):
    cfg = cfgmod.oscar_base()
    m, pre = build_ref(cfg, seed_w)
    batch = synth.make_batch(B, cfg, seed=seed_b, max_seq_len=Lt, img_seq_len=Li, n_regions=n_regions, vary_regions=vary)
    lab = labels_for(batch)
    ids_sub = sorted(set(list(synth.COLOR_IDS) + [synth.NONE_ID] +
                         [int(i) for i in np.random.Generator(np.random.PCG64(5)).integers(0, cfg.vocab_size, 64)]))
    g = dict(B=B, n_regions=n_regions, seed_w=seed_w, seed_b=seed_b, vary=int(vary), ids_sub=np.array(ids_sub), Lt=Lt, Li=Li)
    if with_grads:
        out = m(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
                img_feats=batch["img_feats"], masked_lm_labels=lab)
        loss, scores = out[0], out[1]
        m.zero_grad()
        loss.backward()
        gn = {}
        for k, p in m.named_parameters():
            gn[k] = float(p.grad.double().norm()) if p.grad is not None else -1.0
        g["grad_names"] = np.array(list(gn.keys()))
        g["grad_norms"] = np.array(list(gn.values()), np.float64)
        g["grad_sample_qw"] = m.bert.encoder.layer[11].attention.self.query.weight.grad[:8, :16].numpy().copy()
        g["grad_sample_img"] = m.bert.img_embedding.weight.grad[:8, 2040:2054].numpy().copy()
        g["grad_sample_emb_mask"] = m.bert.embeddings.word_embeddings.weight.grad[synth.MASK, :32].numpy().copy()
        g["loss"] = float(loss)
        scores = scores.detach()
    else:
        with torch.no_grad():
            out = m(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
                    img_feats=batch["img_feats"], masked_lm_labels=lab)
        g["loss"] = float(out[0])
        scores = out[1]
    rows = scores[torch.arange(B), batch["mask_token_pos"]]                 # zeroshot/refcoco_cpt.py:219
    g["mask_logits_sub"] = rows[:, ids_sub].numpy()
    g["mask_logits_argmax"] = rows.argmax(-1).numpy()
    g["mask_logits_absmax"] = float(rows.abs().max())
    # a second row (position 0 = [CLS]) pins the all-rows head too
    g["cls_row_logits_sub"] = scores[:, 0][:, ids_sub].numpy()
    with torch.no_grad():
        seq, pooled = m.bert(batch["input_ids"], batch["segment_ids"], batch["attention_mask"],
                             img_feats=batch["img_feats"])[:2]
        g["seq_sample"] = seq[:, ::17, ::29].numpy()
        g["pooled_sample"] = pooled[:, ::13].numpy()
        g["nsp_scores"] = pre.cls.seq_relationship(pooled).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)


def vcr_nsp_case():
    """Section 8(f).1: the reference's NSPCPT (modeling_vcr.py:79-129) on the tiny config, 2 questions x 4 answer
    choices: relation scores, CE loss against the driver's label construction (fewshot/vcr_nsp_cpt.py:433-436) and
    the driver's choice rule 1 - softmax[:,1] -> argmax per question (:597-604)."""
    from oscar.modeling.modeling_vcr import NSPCPT
    cfg = cfgmod.tiny()
    rc = to_ref_cfg(cfg)
    pre = BertImgForPreTraining(rc)
    sd = synth.init_state_dict(cfg, 4321, head="pretrain")
    pre.load_state_dict(sd, strict=True)
    pre.tie_weights()
    m = NSPCPT(rc)
    m.copy_from_pretraining_model(pre)
    m.eval()
    interval, labels = 4, [2, 0]
    batch = synth.make_batch(8, cfg, seed=17, max_seq_len=20, img_seq_len=6, n_regions=6, vary_regions=True)
    cls_labels = torch.ones([8], dtype=torch.long)
    for i, lb in enumerate(labels):
        cls_labels[i * interval + lb] = 0
    loss, rel = m(batch["input_ids"], batch["segment_ids"], batch["attention_mask"], next_sentence_label=cls_labels,
                  img_feats=batch["img_feats"])[:2]
    m.zero_grad()
    loss.backward()
    logits = 1 - (rel[:, :].softmax(-1)[:, 1].view(-1))
    preds = [int(logits[n * interval:(n + 1) * interval].argmax()) for n in range(2)]
    g = {"in_" + k: v.numpy() for k, v in batch.items()}
    g.update(rel=rel.detach().numpy(), loss=loss.detach().numpy(), cls_labels=cls_labels.numpy(),
             choice_logits=logits.detach().numpy(), preds=np.array(preds), interval=np.array(interval),
             grad_cls_weight=m.cls.weight.grad.numpy(), grad_cls_bias=m.cls.bias.grad.numpy(),
             grad_pooler_weight=m.bert.pooler.dense.weight.grad.numpy(),
             grad_q0_weight=m.bert.encoder.layer[0].attention.self.query.weight.grad.numpy(),
             keys=np.array(sorted(m.state_dict().keys())))
    np.savez_compressed(os.path.join(OUT, "tiny_vcr_nsp.npz"), **g)


def large_vcr_case(name="large_vcr_b2_l265", B=2, Lt=165, Li=100, seed_w=88, seed_b=21):
    """BASELINE configs[4] shape: the reference's NSPCPT (modeling_vcr.py:79-129) on the Oscar-large config (24 layers, hidden 1024,
    16 heads), L = 165 + 100, ragged region counts: relation scores, CE loss against the driver's labels
    (fewshot/vcr_nsp_cpt.py:433-436), four gradient samples + every gradient norm under autograd (dropout off: eval mode)."""
    from oscar.modeling.modeling_vcr import NSPCPT
    cfg = cfgmod.oscar_large()
    rc = to_ref_cfg(cfg)
    pre = BertImgForPreTraining(rc)
    sd = synth.init_state_dict(cfg, seed_w, head="pretrain")
    pre.load_state_dict(sd, strict=True)
    pre.tie_weights()
    m = NSPCPT(rc)
    m.copy_from_pretraining_model(pre)
    m.eval()
    del pre
    batch = synth.make_batch(B, cfg, seed=seed_b, max_seq_len=Lt, img_seq_len=Li, n_regions=Li, vary_regions=True)
    cls_labels = torch.tensor([(i * 2 + 1) % 3 for i in range(B)], dtype=torch.long)
    loss, rel = m(batch["input_ids"], batch["segment_ids"], batch["attention_mask"], next_sentence_label=cls_labels,
                  img_feats=batch["img_feats"])[:2]
    m.zero_grad()
    loss.backward()
    gn = {k: (float(p.grad.double().norm()) if p.grad is not None else -1.0) for k, p in m.named_parameters()}
    with torch.no_grad():
        seq, pooled = m.bert(batch["input_ids"], batch["segment_ids"], batch["attention_mask"], img_feats=batch["img_feats"])[:2]
    g = dict(B=B, Lt=Lt, Li=Li, seed_w=seed_w, seed_b=seed_b, rel=rel.detach().numpy(), loss=float(loss), cls_labels=cls_labels.numpy(),
             choice_logits=(1 - rel.detach().softmax(-1)[:, 1]).numpy(),
             seq_sample=seq[:, ::23, ::37].numpy(), pooled_sample=pooled[:, ::13].numpy(),
             grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values()), np.float64),
             grad_cls_weight=m.cls.weight.grad.numpy().copy(),
             grad_sample_pooler=m.bert.pooler.dense.weight.grad[:8, :16].numpy().copy(),
             grad_sample_q23=m.bert.encoder.layer[23].attention.self.query.weight.grad[:8, :16].numpy().copy(),
             grad_sample_ffn0=m.bert.encoder.layer[0].intermediate.dense.weight.grad[:8, :16].numpy().copy(),
             grad_sample_img=m.bert.img_embedding.weight.grad[:8, 2040:2054].numpy().copy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)


def tsv_rows_case():
    """Section 8(f).2: two rows in the reference's predictions.tsv format (writer: zeroshot/inference_ref.py:157-191,
    SURVEY Appendix B) with small random features, read back with the reference's TSVFile and decoded with the
    reference's decode_features; the TSV itself and the decoded arrays are the fixture."""
    import base64
    from oscar.utils.tsv_file import TSVFile
    import oscar.datasets.refcoco_zsl_cpt_dataset as D
    rng = np.random.Generator(np.random.PCG64(2054))
    tsv = os.path.join(OUT, "tiny_rows.tsv")
    rows = []
    for r, (P, nb) in enumerate(((3, (4, 2, 5)), (2, (1, 3)))):
        objs = []
        for p in range(P):
            boxes = []
            for j in range(nb[p]):
                f = rng.standard_normal(2054).astype(np.float32)
                f[:2048] = np.maximum(f[:2048], 0)
                boxes.append({"rect": [float(v) for v in rng.integers(0, 400, 4)], "bbox_id": j,
                              "class": ["dog", "man", "frisbee", "tree", "car"][j % 5], "conf": float(rng.random()),
                              "feature": base64.b64encode(f.tobytes()).decode("utf-8")})
            objs.append(boxes)
        caption = 'the "feature": "dog" on the left\\' if r == 0 else "man in red"
        rows.append(("img_%d.jpg" % r, json.dumps({"objects": [objs, caption, [["red"]] * P,
                                                                [[[1, 2, 30, 40]]] * P]})))
    with open(tsv, "w") as f:
        for k, v in rows:
            f.write(k + "\t" + v + "\n")
    t = TSVFile(tsv, generate_lineidx=True)
    g = {"n_rows": np.array(t.num_rows())}
    for i in range(t.num_rows()):
        img_name, od_labels, im_feats, caption, colors, rect_lists = D.ZSLColorFinetuneDataset.decode_features(None, t, i)
        g["r%d_name" % i] = np.array(img_name)
        g["r%d_caption" % i] = np.array(caption)
        g["r%d_od_labels" % i] = np.array(od_labels)
        g["r%d_counts" % i] = np.array([f.size(0) for f in im_feats])
        g["r%d_feats" % i] = torch.cat(im_feats, 0).numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_rows_expected.npz"), **g)


PROMPT_VOCAB = (["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] +
                ["the", "a", "dog", "man", "woman", "on", "left", "right", "in", "red", "blue", "green", "is", "color", ".", ",",
                 "fr", "##is", "##bee", "tree", "car", "none", "shirt", "black", "of", "and", "standing", "next", "to", "big",
                 "small", "##s", "##ing", "holding", "traffic", "light", "feature", ":", "\"", "\\", "person", "hat"])


def prompt_tokenizer(tmpdir):
    """A WordPiece tokenizer over a small vocabulary laid out like bert-base-uncased ([MASK] = 103, which the reference
    hard-codes at refcoco_zsl_cpt_dataset.py:116).  The reference takes any object with tokenize / convert_tokens_to_ids."""
    from transformers import BertTokenizer
    vf = os.path.join(tmpdir, "vocab.txt")
    with open(vf, "w") as f:
        f.write("\n".join(PROMPT_VOCAB) + "\n")
    return BertTokenizer(vf, do_lower_case=True)


def prompt_case():
    """Section 8(f).2, prompt assembly: the reference's own templates, tokenize() and ZSLColorFinetuneDataset.__getitem__
    (refcoco_zsl_cpt_dataset.py:18-54, 85-159, 211-302) on a three-row predictions file with varied colours / rectangles /
    annotated boxes, in evaluation and in few-shot (sampling) mode.  The data set object is built without its __init__
    (which wants the RefCOCO annotation tree on disk); everything __getitem__ touches is set by hand."""
    import base64
    import random
    import tempfile
    from oscar.utils.tsv_file import TSVFile
    import oscar.datasets.refcoco_zsl_cpt_dataset as D
    rng = np.random.Generator(np.random.PCG64(70))
    tmp = tempfile.mkdtemp()
    tok = prompt_tokenizer(tmp)
    g = {"vocab": np.array(PROMPT_VOCAB)}
    # (a) templates + tokenize() alone, incl. both truncation branches and the no-text_b form
    cases = [("the dog on the left", "dog man frisbee", 4, 1), ("man in red shirt standing next to the big tree and the small car " * 3, "tree car " * 20, 7, 2),
             ("a woman holding a frisbee", "", 0, 3), ("the person", "red traffic light hat " * 12, 50, 1), ("dogs, trees and cars.", "zebra car", 2, 2)]
    for i, (cap, tb, nf, t) in enumerate(cases):
        text_a = D.tmp_list[t](cap, 0)
        ids, msk, seg, lab = D.tokenize(tok, text_a=text_a, text_b=tb, img_feat=torch.zeros(nf, 2054), max_img_seq_len=50,
                                        max_seq_a_len=40, max_seq_len=70, cls_token_segment_id=0, pad_token_segment_id=0,
                                        sequence_a_segment_id=0, sequence_b_segment_id=1)
        g["tok%d_in" % i] = np.array([cap, tb, str(nf), str(t)])
        g["tok%d_text_a" % i] = np.array(text_a)
        g["tok%d_ids" % i], g["tok%d_mask" % i], g["tok%d_seg" % i], g["tok%d_lab" % i] = ids.numpy(), msk.numpy(), seg.numpy(), lab.numpy()
    for t in (4, 5, 6):
        g["tmpl%d" % t] = np.array([D.tmp_list[t]("man in red shirt", [3, 10]), D.tmp_list[t]("man in red", [10])])
    # (b) whole rows
    tsv = os.path.join(OUT, "tiny_prompt_rows.tsv")
    spec = [("17", "The dog on the left.", (("red", "blue"), ("red",), ("green", "red")),
             (([10, 10, 110, 110], [200, 200, 260, 260]), ([12, 8, 108, 112],), ([300, 0, 320, 40], [0, 0, 50, 50])), [10, 10, 101, 101]),
            ("23", "man in red shirt.", (("red",), ("red",)), (([5, 5, 25, 25],), ([100, 100, 150, 180],)), [100, 100, 51, 81]),
            ("31", "a woman holding a frisbee", (("blue",), ("blue", "green"), ("blue",), ("blue",)),
             (([0, 0, 10, 10],), ([50, 50, 90, 90], [52, 48, 92, 95]), ([400, 400, 420, 420],), ([51, 51, 89, 91],)), [50, 50, 41, 41])]
    rows, anns, dets = [], {}, {}
    for name, cap, colors, rects, gtb in spec:
        objs = []
        for pi, cs in enumerate(colors):
            boxes = []
            for j in range(len(cs)):
                f = rng.standard_normal(2054).astype(np.float32)
                boxes.append({"rect": [float(v) for v in rects[pi][j]], "bbox_id": j, "class": "x", "conf": 0.5,
                              "feature": base64.b64encode(f.tobytes()).decode("utf-8")})
            objs.append(boxes)
        rows.append((name, json.dumps({"objects": [objs, cap, [list(c) for c in colors], [[list(r) for r in rs] for rs in rects]]})))
        anns[name] = {"id": int(name), "bbox": gtb}
        dets[name] = ["dog", "man", "frisbee", "traffic light", "person"][: len(colors) + 1]
    with open(tsv, "w") as f:
        for k, v in rows:
            f.write(k + "\t" + v + "\n")
    g["anns"] = np.array(json.dumps(anns))
    g["dets"] = np.array(json.dumps(dets))
    for is_train in (False, True):
        ds = object.__new__(D.ZSLColorFinetuneDataset)
        ds.tokenizer, ds.txt_seq_len, ds.img_seq_len = tok, 70, 50
        ds.corpus_tsvfile = TSVFile(tsv, generate_lineidx=True)
        ds.anns_dic, ds.det_dic, ds.is_train, ds.template = anns, dets, is_train, D.tmp_list[2]
        random.seed(1234)
        for i in range(len(rows)):
            img_name, feats, ids, msk, seg, mpos, gts, colors, rects = ds[i]
            k = "%s%d_" % ("tr" if is_train else "ev", i)
            g[k + "ids"], g[k + "mask"], g[k + "seg"] = torch.stack(ids).numpy(), torch.stack(msk).numpy(), torch.stack(seg).numpy()
            g[k + "mpos"], g[k + "gts"] = np.array(mpos), np.array(gts)
            g[k + "nfeat"] = np.array([f.size(0) for f in feats])           # padded to img_seq_len by the reference
            g[k + "feat_sum"] = np.array([float(f.double().sum()) for f in feats])
    np.savez_compressed(os.path.join(OUT, "tiny_prompts.npz"), **g)


def caller_goldens():
    """a15 + iou: outputs of the reference helper functions on seeded inputs."""
    rng = np.random.Generator(np.random.PCG64(99))
    boxes = rng.integers(0, 400, size=(64, 2, 4)).astype(np.float64)
    boxes[:, :, 2:] += 5
    ious = np.array([computeIoU(list(b[0]), list(b[1])) for b in boxes])
    np.savez_compressed(os.path.join(OUT, "iou.npz"), boxes=boxes, ious=ious)

    class O(object):
        learning_rate = 3e-5
        warmup_steps = 50
        num_train_steps = 500
    steps = np.arange(0, 520, 7)
    np.savez_compressed(os.path.join(OUT, "lr_sched.npz"), steps=steps,
                        lrs=np.array([get_lr_sched(int(s), O) for s in steps], np.float64))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-prompts" in sys.argv:      # add the prompt-assembly fixture without rewriting the others
        prompt_case()
        return
    if "--only-vcr" in sys.argv:          # add the section 8(f).1 fixture without rewriting the others
        vcr_nsp_case()
        return
    if "--only-tsv" in sys.argv:          # section 8(f).2 fixture
        tsv_rows_case()
        return
    if "--only-cfg45" in sys.argv:        # round 4: BASELINE configs[3] / configs[4] shapes without rewriting the others
        base_case("base_gqa_b2_l210", B=2, n_regions=45, seed_b=11, with_grads=True, vary=True, Lt=165, Li=45)
        large_vcr_case()
        return
    meta = {"reference": "thunlp/CPT @ /root/reference (v1)",
            "third_party_restated": "huggingface/transformers@067923d3267325f525f4e46f357360c191ba562e (pytorch_transformers)",
            "torch": torch.__version__}
    meta["hf_crosscheck_maxabs_tiny"] = hf_crosscheck(cfgmod.tiny(), 1234)
    meta["hf_crosscheck_maxabs_base"] = hf_crosscheck(cfgmod.oscar_base(), 88)
    print("hf cross-check:", meta["hf_crosscheck_maxabs_tiny"], meta["hf_crosscheck_maxabs_base"])
    meta["tiny_ckpt_vs_direct_maxabs"] = tiny_case()
    caller_goldens()
    vcr_nsp_case()
    prompt_case()
    tsv_rows_case()
    base_case("base_cfg1_b2_r36", B=2, n_regions=36)                       # BASELINE config 1 shape
    base_case("base_cfg2_b4_r50", B=4, n_regions=50, with_grads=True)      # config 2 shape (+ grads for config 3)
    base_case("base_ragged_b3", B=3, n_regions=50, seed_b=3, vary=True)
    base_case("base_gqa_b2_l210", B=2, n_regions=45, seed_b=11, with_grads=True, vary=True, Lt=165, Li=45)   # configs[3] shape (Oscar/cmds/gqa/_cpt_fsl_base.sh:19,27)
    large_vcr_case()                                                                                         # configs[4] shape (Oscar-large, 100 regions)
    with open(os.path.join(OUT, "META.json"), "w") as f:
        json.dump(meta, f, indent=2, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
