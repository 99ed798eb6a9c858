"""GPU parity at the other BASELINE.json config shapes (the bench line is configs[1]; these are the
parity-test cases): GQA-CPT shape (L = 165 + 45, answer-id gather), Oscar-large blocks with the NSP-style
relation head (H = 1024, 16 heads, L = 165 + 100), and the driver loops (val / train_batch)."""
import os

import numpy as np
import pytest
import torch

from cpt_amd import config as cfgmod
from cpt_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _pair(cfg, seed, dev, head="cpt"):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    from cpt_amd.modeling_bert import BertImgForPreTraining
    sd = synth.init_state_dict(cfg, seed, head="pretrain")
    pre = BertImgForPreTraining(cfg)
    pre.load_state_dict(sd)
    pre.tie_weights()
    m = REC_MLM_CPT(cfg)
    m.copy_from_pretraining_model(pre)
    m.to(dev).eval()
    pre.to(dev).eval()
    osd = {k.replace("cls.predictions.", "cls."): v for k, v in sd.items()}
    return m, pre, osd


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gqa_shape_answer_gather(dev, mode):
    """config 4 shape: 165 text + 45 regions; logits at the [MASK] slot gathered over 1853 answer ids
    (Oscar/oscar/fewshot/gqa_cpt.py:598-612)."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.oscar_base(num_hidden_layers=3)
    m, _, osd = _pair(cfg, 11, dev)
    m.set_compute_dtype(mode)
    b = synth.make_batch(5, cfg, seed=2, max_seq_len=165, img_seq_len=45, vary_regions=True)
    d = {k: v.to(dev) for k, v in b.items()}
    ans = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).choice(cfg.vocab_size, 1853, replace=False))
    with torch.no_grad():
        ref = O.rec_mlm_cpt_forward(osd, cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"],
                                    img_feats=b["img_feats"], mask_rows_only=b["mask_token_pos"])[0][:, ans]
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"],
                mask_token_pos=d["mask_token_pos"])[0][:, ans.to(dev)].cpu()
    err = (got - ref).abs().max().item()
    print("gqa shape %s: max|d| %.3e" % (mode, err))
    assert err < (1e-3 if mode == "fp32" else 0.1)
    if mode == "fp32":
        assert (got.argmax(1) == ref.argmax(1)).all()
    # the answer gather + argmax on the device (cpt_argmax_columns) picks what the host rule picks from the same logits
    from cpt_amd import scoring
    with torch.no_grad():
        full = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"],
                 mask_token_pos=d["mask_token_pos"])[0]
    assert scoring.argmax_columns_device(full, ans.tolist()).cpu().tolist() == full[:, ans.to(dev)].argmax(1).cpu().tolist()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_oscar_large_blocks_nsp_head(dev, mode):
    """config 5 shape: hidden 1024, 16 heads, intermediate 4096, L = 165 + 100; NSPCPT scoring =
    pooled [CLS] -> cls.seq_relationship (Oscar/oscar/modeling/modeling_vcr.py:79-129), 4 choices."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.oscar_large(num_hidden_layers=2)
    m, pre, osd = _pair(cfg, 5, dev)
    pre.set_compute_dtype(mode)
    b = synth.make_batch(4, cfg, seed=8, max_seq_len=165, img_seq_len=100, vary_regions=True)
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        ref = O.nsp_cpt_scores(osd, cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"], b["img_feats"])
        got = pre(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"])[1].cpu()
    err = (got - ref).abs().max().item()
    print("oscar-large NSP %s: max|d| %.3e" % (mode, err))
    assert err < (1e-3 if mode == "fp32" else 0.05)
    score = lambda x: 1 - torch.softmax(x, -1)[:, 1]            # vcr_nsp_cpt scoring
    if mode == "fp32":
        assert int(score(got).argmax()) == int(score(ref).argmax())


def _queries(cfg, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    qs = []
    for i in range(n):
        P = int(rng.integers(1, 5))
        b = synth.make_batch(P, cfg, seed=seed * 100 + i, vary_regions=True)
        colors = [[int(c) for c in rng.choice(synth.COLOR_IDS, size=int(rng.integers(1, 4)), replace=False)] for _ in range(P)]
        rects = [[[10 * j, 5 * k, 10 * j + 30, 5 * k + 40] for k in range(len(colors[j]))] for j in range(P)]
        q = {k: b[k] for k in ("img_feats", "input_ids", "segment_ids", "attention_mask", "mask_token_pos")}
        q["colors"], q["rects"] = colors, rects
        qs.append(q)
    return qs


def test_val_driver_matches_oracle_selection(dev):
    """Region selection of the val() counterpart == the oracle's (reference algorithm) on the same queries."""
    from cpt_amd import drivers
    from oracle import cpt_oracle as O
    cfg = cfgmod.oscar_base(num_hidden_layers=2)
    m, _, osd = _pair(cfg, 21, dev)
    qs = _queries(cfg, 7, 3)
    for few_shot in (False, True):
        got = drivers.val_queries(m, qs, synth.NONE_ID, dev, few_shot=few_shot, batch_queries=3)
        for gi, q in enumerate(qs):
            with torch.no_grad():
                rows = O.rec_mlm_cpt_forward(osd, cfg.to_dict(), q["input_ids"], q["segment_ids"], q["attention_mask"],
                                             img_feats=q["img_feats"], mask_rows_only=q["mask_token_pos"])[0]
            sel = O.select_region_fewshot if few_shot else O.select_region_zeroshot
            idx, sc = sel(rows, q["colors"], synth.NONE_ID)
            top2 = sc.topk(min(2, sc.numel())).values
            margin = float(top2[0] - top2[-1]) if sc.numel() > 1 else 1.0
            assert got[gi][0] == idx or margin < 1e-3, (gi, got[gi][0], idx, margin)


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_decoder_on_a_column_list_equals_those_columns_of_the_full_scores(dev, mode):
    """Round 6 (SURVEY a11 "or just colour columns"; zeroshot/refcoco_cpt.py:219, gqa_cpt.py:598-600): cpt_outputs.logit_cols -- the decoder scores a list of
    vocabulary columns only.  Against the same columns of the full (B, V) scores of the same mode (a dot product in fp32 instead of the MFMA tile's order),
    duplicates and unsorted ids included; and the drivers' selections are the same with and without the column list."""
    from cpt_amd import drivers
    cfg = cfgmod.oscar_base(num_hidden_layers=2)
    m, _, _ = _pair(cfg, 23, dev)
    m.set_compute_dtype(mode)
    b = {k: v.to(dev) for k, v in synth.make_batch(9, cfg, seed=5, vary_regions=True).items()}
    ids = list(synth.COLOR_IDS) + [synth.NONE_ID, 0, cfg.vocab_size - 1, synth.COLOR_IDS[0]]
    cols = torch.tensor(ids, dtype=torch.int64, device=dev)
    with torch.no_grad():
        full = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
        part = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"], vocab_columns=cols)[0]
    assert tuple(part.shape) == (9, len(ids))
    err = float((part - full[:, cols]).abs().max())
    print("column decoder %s: max |d| %.3e" % (mode, err))
    assert err < {"fp32": 2e-5, "bf16": 2e-5, "bf16x3": 2e-4}[mode]      # (bf16: the same bf16 operands, fp32 sums in another order; bf16x3: hi + lo weights against the three-term product)
    with pytest.raises(Exception):
        m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], vocab_columns=cols)        # all-row scores: no column list
    qs = _queries(cfg, 7, 3)
    for few_shot in (False, True):
        a = drivers.val_queries(m, qs, synth.NONE_ID, dev, few_shot=few_shot, batch_queries=3, colour_columns_only=True)
        c = drivers.val_queries(m, qs, synth.NONE_ID, dev, few_shot=few_shot, batch_queries=3, colour_columns_only=False)
        assert {k: v[0] for k, v in a.items()} == {k: v[0] for k, v in c.items()}


def test_train_batch_driver_runs_and_learns(dev):
    from cpt_amd import drivers
    from cpt_amd.modeling_rec import REC_MLM_CPT
    from cpt_amd.train import build_optimizer
    cfg = cfgmod.tiny()
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 3, head="cpt"))
    m.tie_weights()
    m.to(dev)

    class Opts(object):
        learning_rate, weight_decay, betas = 2e-3, 0.01, (0.9, 0.98)
        warmup_steps, num_train_steps = 2, 12
    opt = build_optimizer(m, Opts)
    batch = synth.make_batch(6, cfg, seed=4, max_seq_len=20, img_seq_len=6)
    step, losses = drivers.train_batch(m, opt, [batch] * 12, Opts, dev)
    assert step == 12 and len(losses) == 12
    assert float(losses[-1]) < float(losses[0]) - 0.3


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_vcr_nspcpt_golden(dev, golden_dir, mode):
    """Section 8(f).1: NSPCPT (modeling_vcr.py:79-129) through the HIP path against the fixture produced by the
    reference's own class: relation scores, CE loss (fewshot/vcr_nsp_cpt.py:433-436 labels) and the driver's
    choice rule (:597-604)."""
    from cpt_amd.modeling_bert import BertImgForPreTraining
    from cpt_amd.modeling_vcr import NSPCPT
    from cpt_amd import scoring
    g = np.load(os.path.join(golden_dir, "tiny_vcr_nsp.npz"))
    cfg = cfgmod.tiny()
    pre = BertImgForPreTraining(cfg)
    pre.load_state_dict(synth.init_state_dict(cfg, 4321, head="pretrain"))
    pre.tie_weights()
    m = NSPCPT(cfg)
    m.copy_from_pretraining_model(pre)
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    m.to(dev).eval().set_compute_dtype(mode)
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    interval = int(g["interval"])
    lab = scoring.nsp_choice_labels([2, 0], interval, 8, device=dev)
    assert (lab.cpu().numpy() == g["cls_labels"]).all()
    with torch.no_grad():
        loss, rel = m(b["input_ids"], b["segment_ids"], b["attention_mask"], next_sentence_label=lab, img_feats=b["img_feats"])
        rel_only = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"])[0]
    tol = 2e-5 if mode == "fp32" else 5e-3
    assert (rel.cpu() - torch.from_numpy(g["rel"])).abs().max().item() < tol
    assert torch.equal(rel, rel_only)
    assert abs(loss.item() - float(g["loss"])) < (1e-5 if mode == "fp32" else 5e-3)
    logits, preds = scoring.nsp_choose(rel.cpu(), interval)
    assert (logits - torch.from_numpy(g["choice_logits"])).abs().max().item() < tol
    if mode == "fp32":
        assert preds == list(g["preds"])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_vcr_nspcpt_finetune_gradients(dev, golden_dir, mode):
    """NSPCPT fine-tuning (fewshot/vcr_nsp_cpt.py:425-470): loss.backward() through the HIP training step with the NSP head
    against the gradients the reference's own NSPCPT produced under autograd (fixture), every other gradient against
    the oracle's autograd, and a few AdamW steps that must lower the loss."""
    from cpt_amd.modeling_bert import BertImgForPreTraining
    from cpt_amd.modeling_vcr import NSPCPT
    from cpt_amd import scoring
    from cpt_amd.train import FusedAdamW
    from oracle import cpt_oracle as O
    g = np.load(os.path.join(golden_dir, "tiny_vcr_nsp.npz"))
    cfg = cfgmod.tiny()
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
    pre = BertImgForPreTraining(cfg)
    sd0 = synth.init_state_dict(cfg, 4321, head="pretrain")
    pre.load_state_dict(sd0)
    pre.tie_weights()
    m = NSPCPT(cfg)
    m.copy_from_pretraining_model(pre)
    m.to(dev).train().set_compute_dtype(mode)
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    lab = scoring.nsp_choice_labels([2, 0], int(g["interval"]), 8, device=dev)
    loss, rel = m(b["input_ids"], b["segment_ids"], b["attention_mask"], next_sentence_label=lab, img_feats=b["img_feats"])
    loss.backward()
    ltol, gtol = {"fp32": (1e-5, 2e-4), "bf16x3": (2e-5, 5e-4), "bf16": (5e-3, 8e-2)}[mode]
    assert abs(loss.item() - float(g["loss"])) < ltol
    assert (rel.detach().cpu() - torch.from_numpy(g["rel"])).abs().max().item() < {"fp32": 2e-5, "bf16x3": 1e-4, "bf16": 5e-3}[mode]
    named = dict(m.named_parameters())

    def rel_err(name, ref):
        got = named[name].grad.double().cpu().flatten()
        ref = torch.as_tensor(ref).double().flatten()
        return float((got - ref).norm() / (ref.norm() + 1e-30))
    # the reference's own autograd (oracle/make_golden.py: vcr_nsp_case)
    for name, key in (("cls.weight", "grad_cls_weight"), ("cls.bias", "grad_cls_bias"), ("bert.pooler.dense.weight", "grad_pooler_weight"),
                      ("bert.encoder.layer.0.attention.self.query.weight", "grad_q0_weight")):
        assert rel_err(name, g[key]) < gtol, name
    # everything else: the oracle's autograd on the same inputs
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ob = {k: v.cpu() for k, v in b.items()}
    oloss, _ = O.nsp_cpt_forward(osd, cfg.to_dict(), ob["input_ids"], ob["segment_ids"], ob["attention_mask"], ob["img_feats"],
                                 next_sentence_label=lab.cpu(), w_key="cls.weight", b_key="cls.bias")
    oloss.backward()
    worst = 0.0
    for name, prm in named.items():
        ref = osd[name].grad
        if ref is None or prm.grad is None:
            assert ref is None or float(ref.abs().max()) == 0.0, name
            continue
        if float(ref.abs().max()) < 1e-6 and float(prm.grad.abs().max()) < (1e-3 if mode == "bf16" else 1e-5):
            continue                     # key bias: true gradient 0 (softmax is shift-invariant), rounding noise on both sides
        e = rel_err(name, ref)
        worst = max(worst, e)
        assert e < gtol, (name, e)
    print("NSPCPT %s: loss %.5f, worst relative gradient error %.2e" % (mode, loss.item(), worst))
    # a few optimizer steps
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01)
    losses = [loss.item()]
    for _ in range(4):
        opt.step()
        opt.zero_grad()
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], next_sentence_label=lab, img_feats=b["img_feats"])
        loss.backward()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses


def test_region_stager_matches_host_decode(dev):
    """Section 8(f).2: features decoded by the C decoder into pinned memory and staged on the side stream arrive on
    the GPU bit-identical to the Python decode, and drive the model to the same logits as the host tensor."""
    import base64
    from cpt_amd import io
    rng = np.random.default_rng(11)
    counts = [50, 36, 0, 17]
    data = [[np.maximum(rng.standard_normal(2054), 0).astype(np.float32) for _ in range(c)] for c in counts]
    lists = [[base64.b64encode(a.tobytes()).decode() for a in seq] for seq in data]
    st = io.RegionStager(max_seqs=8, device=str(dev), threads=2)
    for _ in range(3):                                   # cycles through the pinned ring
        feats, mask, ev = st.stage(lists)
        torch.cuda.current_stream().wait_event(ev)
        ref = torch.zeros(4, 50, 2054)
        for p, seq in enumerate(data):
            if seq:
                ref[p, :len(seq)] = torch.from_numpy(np.stack(seq))
        assert torch.equal(feats.cpu(), ref)
        assert mask.cpu().sum(1).tolist() == counts
        st.release()
    # write-after-read: a slow consumer and a host that runs `depth` batches ahead -- the side stream must
    # not overwrite slot k before the consumer's reads of it (enqueued behind a long kernel) have run
    outs, refs = [], []
    for it in range(6):
        data_it = [[np.full(2054, 100.0 * it + p + 0.25 * j, dtype=np.float32) for j in range(c)] for p, c in enumerate(counts)]
        lists_it = [[base64.b64encode(a.tobytes()).decode() for a in seq] for seq in data_it]
        feats, mask = st.stage_and_wait(lists_it)
        torch.cuda._sleep(40_000_000)                     # ~20 ms of consumer-stream work ahead of the read
        outs.append(feats.clone())
        st.release()
        r = torch.zeros(4, 50, 2054)
        for p, seq in enumerate(data_it):
            if seq:
                r[p, :len(seq)] = torch.from_numpy(np.stack(seq))
        refs.append(r)
    for o, r in zip(outs, refs):
        assert torch.equal(o.cpu(), r)


@pytest.mark.parametrize("shape", [("base", 1, 70, 50), ("base", 5, 78, 50), ("base", 3, 28, 5), ("large", 3, 70, 50)])
def test_fused_bf16_encoder_odd_shapes(dev, shape):
    """The fused bf16 forms (folded LayerNorm, QKV + attention in one kernel) at the shapes the bench does not visit:
    a single sequence (M < one tile), L = 128 exactly, L = 33 (last query block nearly empty), and 16 heads / H = 1024
    (11 statistics slots: an odd count).  Fusing attention alone must not change a bit; the fully fused encoder
    must stay inside the bf16 band of the fp32 parity mode."""
    from cpt_amd import _lib as L
    from cpt_amd.modeling_rec import REC_MLM_CPT
    kind, B, Lt, Li = shape
    cfg = cfgmod.oscar_base(num_hidden_layers=3) if kind == "base" else cfgmod.oscar_large(num_hidden_layers=2)
    cfg.max_position_embeddings = max(cfg.max_position_embeddings, Lt)
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 7, head="cpt"))
    m.tie_weights()
    m.to(dev).eval()
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=9, max_seq_len=Lt, img_seq_len=Li, vary_regions=True).items()}

    def run():
        with torch.no_grad():
            return m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                     mask_token_pos=b["mask_token_pos"])[0].float().cpu()
    m.set_compute_dtype("fp32")
    ref = run()
    m.set_compute_dtype("bf16")
    res = {}
    abl = L.ablation_build()          # development build: also walk the two-kernel / fused attention forms (cpt_set_tuning key 6); product: the shipped form
    try:
        for fold, fuse in (((False, 0), (False, 1), (True, 0), (True, 1)) if abl else ((False, None), (True, None))):
            m._engine().fold_ln = fold
            if fuse is not None:
                L.check(L.lib().cpt_set_tuning(6, fuse), "cpt_set_tuning")
            res[(fold, fuse)] = run()
    finally:
        L.lib().cpt_set_tuning(-1, 0)
    if abl:
        assert torch.equal(res[(False, 0)], res[(False, 1)])
        assert torch.equal(res[(True, 0)], res[(True, 1)])
    band = (res[(False, 0 if abl else None)] - ref).abs().max().item()
    for k, v in res.items():
        err = (v - ref).abs().max().item()
        print(shape, k, "max |bf16 - fp32| = %.3e" % err)
        assert err < max(2.5 * band, 0.05)
