"""GPU parity at the other BASELINE.json config shapes (the bench line is configs[1]; these are the
parity-test cases): GQA-CPT shape (L = 165 + 45, answer-id gather), Oscar-large blocks with the NSP-style
relation head (H = 1024, 16 heads, L = 165 + 100), and the driver loops (val / train_batch)."""
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
