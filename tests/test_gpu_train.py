"""GPU parity of the few-shot training step (cpt_train_fwd / cpt_train_bwd / cpt_adamw through the
C ABI) against the golden gradients generated from the reference (tiny config: every gradient;
Oscar-base: loss, gradient norms and samples) and against the reference's 3-step AdamW trace."""
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


def _model(cfg, seed, dev, dtype, dropout=0.0):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    # the reference goldens are generated with dropout disabled (oracle/make_golden.py); dropout tests pass their own rate
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = dropout
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, seed, head="cpt"))
    m.tie_weights()
    m.to(dev).train()
    m.set_compute_dtype(dtype)
    return m


def _rel(got, ref):
    got, ref = got.double().cpu().flatten(), torch.as_tensor(ref).double().flatten()
    return float((got - ref).norm() / (ref.norm() + 1e-30)), float((got - ref).abs().max())


# bf16x3: the parity mode at MFMA-bf16 rates (every GEMM of the step as three bf16 MFMA terms over split fp32 operands): fp32-grade bounds
@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_tiny_all_gradients(dev, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "tiny_fwd_bwd.npz"))
    cfg = cfgmod.tiny()
    m = _model(cfg, 1234, dev, mode)
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    lab = torch.full(b["attention_mask"].shape, -1, dtype=torch.long, device=dev)
    lab[torch.arange(3, device=dev), b["mask_token_pos"]] = b["colors"]
    # reference call form: (B, L) label grid, no mask_token_pos
    loss, scores = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=lab)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < {"fp32": 1e-4, "bf16x3": 1e-4, "bf16": 3e-2}[mode]
    worst = 0.0
    n = 0
    for name, p in m.named_parameters():
        key = "grad_" + name
        if key not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name     # pooler: no gradient
            continue
        rel, mx = _rel(p.grad, g[key])
        print("%-60s rel=%.2e maxabs=%.2e" % (name, rel, mx))
        worst = max(worst, rel)
        n += 1
        # key.bias gradients are exactly zero in exact arithmetic (softmax is shift-invariant per row):
        # both sides hold rounding noise there, so fall back to an absolute bound
        assert rel < {"fp32": 2e-4, "bf16x3": 5e-4, "bf16": 6e-2}[mode] or mx < {"fp32": 1e-9, "bf16x3": 1e-8, "bf16": 2e-6}[mode], (name, rel, mx)
    assert n > 30
    assert m.bert.pooler.dense.weight.grad is None
    print("worst relative gradient error (%s): %.3e" % (mode, worst))


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_base_gradients_cfg3_shape(dev, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "base_cfg2_b4_r50.npz"))
    cfg = cfgmod.oscar_base()
    m = _model(cfg, int(g["seed_w"]), dev, mode)
    b = {k: v.to(dev) for k, v in synth.make_batch(int(g["B"]), cfg, seed=int(g["seed_b"]), n_regions=int(g["n_regions"])).items()}
    loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                mask_token_pos=b["mask_token_pos"])
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < (5e-2 if mode == "bf16" else 1e-3)
    names, norms = list(g["grad_names"]), g["grad_norms"]
    params = dict(m.named_parameters())
    tol = 8e-2 if mode == "bf16" else 1e-3
    for name, ref in zip(names, norms):
        name = str(name)
        if ref < 0:
            continue
        got = float(params[name].grad.double().norm())
        if ".key.bias" in name:            # exactly zero in exact arithmetic: noise on both sides
            assert got < 1e-4 and ref < 1e-6, (name, got, ref)
            continue
        assert abs(got - ref) <= tol * max(ref, 1e-6), (name, got, ref)
    q = params["bert.encoder.layer.11.attention.self.query.weight"].grad[:8, :16]
    rel, mx = _rel(q, g["grad_sample_qw"])
    assert rel < (0.15 if mode == "bf16" else 1e-3), (rel, mx)
    gi = params["bert.img_embedding.weight"].grad[:8, 2040:2054]
    rel, mx = _rel(gi, g["grad_sample_img"])
    assert rel < (0.15 if mode == "bf16" else 1e-3), (rel, mx)
    ge = params["bert.embeddings.word_embeddings.weight"].grad[synth.MASK, :32]
    rel, mx = _rel(ge, g["grad_sample_emb_mask"])
    assert rel < (0.15 if mode == "bf16" else 1e-3), (rel, mx)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_tiny_train3_trace_fp32(dev, golden_dir, mode):
    """3 few-shot steps (label grid, LR schedule, AdamW groups) reproduce the reference's loss trace
    and updated parameters -- in fp32 mode and in the bf16x3 parity mode (same bounds)."""
    from cpt_amd.train import build_optimizer, get_lr_sched
    t = np.load(os.path.join(golden_dir, "tiny_train3.npz"))
    g = np.load(os.path.join(golden_dir, "tiny_fwd_bwd.npz"))
    cfg = cfgmod.tiny()
    m = _model(cfg, 1234, dev, mode)
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}

    class O(object):
        learning_rate = float(t["lr0"])
        weight_decay = float(t["wd"])
        betas = (float(t["beta1"]), float(t["beta2"]))
        warmup_steps = 1
        num_train_steps = 3
    opt = build_optimizer(m, O)
    for step in range(3):
        lr = get_lr_sched(step, O)
        assert lr == t["lrs"][step]
        for gr in opt.param_groups:
            gr["lr"] = lr
        opt.zero_grad()
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                    masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        opt.step()
        assert abs(loss.item() - t["losses"][step]) < 2e-4, (step, loss.item(), t["losses"][step])
    sd = m.state_dict()
    for k in t.files:
        if k.startswith("after_"):
            rel, mx = _rel(sd[k[6:]], t[k])
            # (bf16x3: where a gradient is ~0 the first AdamW steps move the weight by +-lr on its rounding noise: bound the tensor, not the element)
            assert mx < 5e-5 if mode == "fp32" else rel < 2e-4, (k, rel, mx)
    assert torch.equal(sd["cls.decoder.weight"], sd["bert.embeddings.word_embeddings.weight"])
    if mode == "bf16x3":
        # the standing split copies of the weights (what the INFERENCE forward of this mode reads) went stale with every step and
        # were never rebuilt by the training steps; the evaluation forward behind them must see the trained weights
        assert m._engine()._x3_stale
        m.eval()
        with torch.no_grad():
            got = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
            assert not m._engine()._x3_stale
            m.set_compute_dtype("fp32")
            want = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
        assert (got - want).abs().max().item() < 2e-4


def test_bf16_training_reduces_loss(dev):
    """bf16 mode: 5 steps on one batch drive the loss down and keep the bf16 shadow in sync
    (inference after training sees the updated weights)."""
    from cpt_amd.train import FusedAdamW
    cfg = cfgmod.tiny()
    m = _model(cfg, 7, dev, "bf16")
    b = {k: v.to(dev) for k, v in synth.make_batch(6, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}
    opt = FusedAdamW(m, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                    masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.5, losses
    m.eval()
    with torch.no_grad():
        l2 = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
               masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])[0].item()
    assert l2 < losses[-1] + 0.05


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_no_img_layernorm_forward_and_gradients(dev, mode):
    """use_img_layernorm = 0 (modeling_bert.py:263-264: the region projection enters the encoder without LayerNorm): logits,
    loss and every gradient against the oracle's forward / autograd; there is no bert.LayerNorm parameter at all."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.tiny(use_img_layernorm=0)
    m = _model(cfg, 77, dev, mode)
    assert not any(n.startswith("bert.LayerNorm") for n, _ in m.named_parameters())
    b = synth.make_batch(3, cfg, seed=11, max_seq_len=20, img_seq_len=6, vary_regions=True)
    d = {k: v.to(dev) for k, v in b.items()}
    loss, scores = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=d["colors"],
                     mask_token_pos=d["mask_token_pos"])
    loss.backward()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd["cls.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    ref_loss, ref = O.train_step_grads(sd, cfg.to_dict(), b)
    ltol, gtol = (2e-4, 2e-4) if mode == "fp32" else (4e-2, 8e-2)
    assert abs(loss.item() - float(ref_loss)) < ltol, (loss.item(), float(ref_loss))
    n = 0
    for name, prm in m.named_parameters():
        g = ref.get(name)
        if g is None:
            continue
        rel, mx = _rel(prm.grad, g)
        if float(g.abs().max()) < 1e-6 and mx < (1e-5 if mode == "fp32" else 1e-3):
            continue          # key bias
        assert rel < gtol, (name, rel, mx)
        n += 1
    assert n > 25
    # inference path of the same model
    m.eval()
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
    want = O.rec_mlm_cpt_forward(sd, cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                                 mask_rows_only=b["mask_token_pos"])[0]
    err = (got.float().cpu() - want).abs().max().item()
    assert err < (1e-3 if mode == "fp32" else 4e-2), err


@pytest.mark.ablation
@pytest.mark.parametrize("size", ["base", "large"])
def test_weight_gradients_tn_gemm_vs_transposed_operands(dev, size):
    """bf16 weight / data gradients: the TN / NN GEMM forms (operands read as stored) against the explicit-transpose + NT GEMM path
    on the same batch (M = 8 x 120 = 960 rows = 15 K-tiles), every parameter; both accumulate in fp32 over the same bf16 products.
    large: hidden 1024 / 4096 (Oscar-large, the VCR few-shot model) takes the 128-column tile instantiations."""
    from cpt_amd import _lib as L
    cfg = cfgmod.oscar_base(num_hidden_layers=2) if size == "base" else cfgmod.oscar_large(num_hidden_layers=2)
    b = {k: v.to(dev) for k, v in synth.make_batch(8, cfg, seed=21).items()}
    grads = {}
    for tn in (0, 1):
        L.check(L.lib().cpt_set_tuning(10, tn))
        try:
            m = _model(cfg, 5, dev, "bf16")
            loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                        masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
            loss.backward()
            grads[tn] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            L.check(L.lib().cpt_set_tuning(10, 1))
    for n in grads[0]:
        rel, mx = _rel(grads[1][n], grads[0][n].cpu())
        # The two paths add the same bf16 products in different fp32 orders.  The decoder's data gradient sums 30522 terms of
        # dlogits = softmax - onehot (which cancel to ~0), so its two orders differ by ~6e-5 of the result already (split-K over 64
        # workgroups vs one running sum); bf16 casts and the softmax backward carry that to ~2e-3 at the query / key weights.
        # A wrong tile or a missing K slice would be O(1).  (The operator tests pin each form against fp32 matmul.)
        if ".key.bias" in n:
            continue            # true gradient 0 (softmax is shift-invariant): rounding noise on both sides
        assert rel < 6e-3 or mx < 1e-7, (n, rel, mx)


@pytest.mark.ablation
def test_attention_backward_variants_agree(dev):
    """MFMA attention backward (bf16) against the generic fp32-math kernel on the same inputs."""
    from cpt_amd import _lib as L
    cfg = cfgmod.oscar_base(num_hidden_layers=2)
    b = {k: v.to(dev) for k, v in synth.make_batch(3, cfg, seed=9, vary_regions=True).items()}
    grads = {}
    for variant in (0, 1, 2):           # 0 generic fp32-math kernel, 1 MFMA with transposed tile copies, 2 MFMA with LDS transpose reads
        L.check(L.lib().cpt_set_tuning(2, variant))
        try:
            m = _model(cfg, 3, dev, "bf16")
            loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                        masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
            loss.backward()
            grads[variant] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            L.check(L.lib().cpt_set_tuning(2, 1))
    for n in grads[0]:
        if ".key.bias" in n:
            continue
        rel, mx = _rel(grads[1][n], grads[0][n].cpu())
        assert rel < 3e-2, (n, rel, mx)
        # the two MFMA kernels issue the same score / dP products (operands read two ways); round 6: the default one takes the forward's softmax
        # statistics and D = rowsum(dO . O) over the bf16-rounded context rows where the other recomputes rowsum(dP . P): 2^-9-grade differences in dS
        rel2, mx2 = _rel(grads[2][n], grads[1][n].cpu())
        assert rel2 < 8e-3, (n, rel2, mx2)


def test_checkpoint_resume_reproduces_reference_trace(dev, golden_dir, tmp_path):
    """save (weights in HF layout + AdamW moments + step) after step 1, load into a FRESH model and
    optimizer, run steps 2-3: the losses and final parameters still match the reference's uninterrupted 3-step
    trace, and the moments round-trip bit-exactly."""
    from cpt_amd.train import build_optimizer, get_lr_sched, load_checkpoint, save_checkpoint
    t = np.load(os.path.join(golden_dir, "tiny_train3.npz"))
    g = np.load(os.path.join(golden_dir, "tiny_fwd_bwd.npz"))
    cfg = cfgmod.tiny()
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}

    class O(object):
        learning_rate = float(t["lr0"])
        weight_decay = float(t["wd"])
        betas = (float(t["beta1"]), float(t["beta2"]))
        warmup_steps = 1
        num_train_steps = 3

    def one_step(m, opt, step):
        for gr in opt.param_groups:
            gr["lr"] = get_lr_sched(step, O)
        opt.zero_grad()
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                    masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        opt.step()
        return loss.item()

    m = _model(cfg, 1234, dev, "fp32")
    opt = build_optimizer(m, O)
    assert abs(one_step(m, opt, 0) - t["losses"][0]) < 2e-4
    ck = str(tmp_path / "ckpt")
    save_checkpoint(ck, m, opt, global_step=1)
    assert sorted(os.listdir(ck)) == ["config.json", "optimizer.pt", "pytorch_model.bin", "training_state.json"]
    m2 = _model(cfg, 999, dev, "fp32")                      # different weights: everything must come from the checkpoint
    opt2 = build_optimizer(m2, O)
    assert load_checkpoint(ck, m2, opt2) == 1
    sd1, sd2 = opt.state_dict(), opt2.state_dict()
    for n in sd1["state"]:
        assert torch.equal(sd1["state"][n]["exp_avg"], sd2["state"][n]["exp_avg"])
        assert torch.equal(sd1["state"][n]["exp_avg_sq"], sd2["state"][n]["exp_avg_sq"])
    assert opt2.step_count == 1
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    for step in (1, 2):
        assert abs(one_step(m2, opt2, step) - t["losses"][step]) < 2e-4
    sd = m2.state_dict()
    for k in t.files:
        if k.startswith("after_"):
            rel, mx = _rel(sd[k[6:]], t[k])
            assert mx < 5e-5, (k, rel, mx)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_backward_needs_no_whole_buffer_fill(dev, mode):
    """Round 3: the step clears only what the backward ADDS into (cpt_train_zero_grads: bias / LayerNorm vectors, small tables);
    Linear weight gradients and the tied word table are written whole.  A gradient buffer poisoned with NaN between two
    backward passes must come out as after the first one (optimizer.zero_grad(), fewshot/refcoco_cpt.py:247-249)."""
    from cpt_amd import train as T
    cfg = cfgmod.tiny()
    m = _model(cfg, 99, dev, mode, dropout=0.1)
    T.set_dropout_seed(m, 5)
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=3, max_seq_len=20, img_seq_len=6).items()}

    def step():
        T.set_dropout_seed(m, 5)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    g1 = step()
    st = T._state(m._engine())
    st.grad.fill_(float("nan"))
    g2 = step()
    assert set(g1) == set(g2) and len(g1) > 30
    for n in g1:
        assert torch.isfinite(g2[n]).all(), n
        if g1[n].dim() == 2 and "embeddings" not in n:      # GEMM-written: same bits
            assert torch.equal(g1[n], g2[n]), n
        else:                                                   # atomic accumulation order
            assert float((g1[n] - g2[n]).abs().max()) <= 1e-5 * max(1.0, float(g1[n].abs().max())), n
    st.grad.zero_()


@pytest.mark.parametrize("B", [4, 32])
def test_backward_needs_no_whole_buffer_fill_base_shape(dev, B):
    """the same poison test at the Oscar-base shape in bf16, where the paired / triple weight-gradient launches
    (gemm_tn_pair / gemm_tn_triple, cpt_set_tuning key 19) and the split-K reductions write the matrices: B = 32 (triple launch,
    one round) and B = 4 (config 3's per-GPU share: short contractions, no split)."""
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base()
    m = _model(cfg, 99, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=3).items()}

    def step():
        T.set_dropout_seed(m, 5)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    g1 = step()
    st = T._state(m._engine())
    st.grad.fill_(float("nan"))
    g2 = step()
    assert set(g1) == set(g2) and len(g1) > 190
    for n in g1:
        assert torch.isfinite(g2[n]).all(), n
        if g1[n].dim() == 2 and "embeddings" not in n:      # GEMM-written: same bits
            assert torch.equal(g1[n], g2[n]), n
        else:                                                   # atomic accumulation order
            assert float((g1[n] - g2[n]).abs().max()) <= 2e-3 * max(1e-3, float(g1[n].abs().max())), n
    st.grad.zero_()


@pytest.mark.ablation
@pytest.mark.parametrize("B", [4, 32])
def test_bias_gradients_summed_inside_their_producers(dev, B):
    """Round 3: the stacked Q|K|V bias gradient is summed by the attention backward kernel (cpt_set_tuning key 18 bit 1, default),
    the intermediate bias gradient optionally by the GELU-gradient epilogue (bit 0): both against the stand-alone column-sum
    launches (key 18 = 0), Oscar-base at 4 (64 x 192 tiles) and 32 (128 x 192 tiles) sequences, dropout on."""
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base()
    m = _model(cfg, 21, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=9).items()}
    names = ["bert.encoder.layer.%d.%s" % (l, n) for l in (0, 11)
             for n in ("attention.self.query.bias", "attention.self.value.bias", "intermediate.dense.bias", "output.dense.bias")]
    params = dict(m.named_parameters())

    def grads(bits):
        L.check(L.lib().cpt_set_tuning(18, bits), "cpt_set_tuning")
        T.set_dropout_seed(m, 11)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return {n: params[n].grad.double().clone() for n in names}

    ref = grads(0)
    for bits in (1, 2, 3):
        got = grads(bits)
        for n in names:
            den = float(ref[n].norm()) + 1e-30
            rel = float((got[n] - ref[n]).norm()) / den
            # the fused sums take the fp32 values, the launches the bf16-rounded tensor: 2^-9 relative per element, averaging out
            assert rel < 2e-3, (bits, n, rel)


@pytest.mark.ablation
@pytest.mark.parametrize("B", [4, 32])
def test_paired_weight_gradient_launches_match_single_ones(dev, B):
    """Round 3: a layer's four weight gradients as two paired launches (gemm_tn_pair: FFN down | FFN up without a K split, attention
    output | Q|K|V with the contraction split in two) against four single launches with their own split-K reductions (cpt_set_tuning
    key 19 = 0): same operands, a different summation split -> equal to fp32 rounding."""
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base()
    m = _model(cfg, 23, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=13).items()}
    names = ["bert.encoder.layer.%d.%s" % (l, n) for l in (0, 5, 11)
             for n in ("attention.self.query.weight", "attention.self.value.weight", "attention.output.dense.weight",
                       "intermediate.dense.weight", "output.dense.weight")]
    params = dict(m.named_parameters())

    def grads(pair):
        L.check(L.lib().cpt_set_tuning(19, pair), "cpt_set_tuning")
        T.set_dropout_seed(m, 17)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return {n: params[n].grad.double().clone() for n in names}

    ref = grads(0)
    for mode in (1, 2):       # 1: two paired launches; 2 (default): FFN down | FFN up | attention output in one launch + Q|K|V alone
        got = grads(mode)
        for n in names:
            rel = float((got[n] - ref[n]).norm()) / (float(ref[n].norm()) + 1e-30)
            assert rel < 1e-5, (mode, n, rel)
            assert float(ref[n].norm()) > 0, n


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_label_grid_with_several_labelled_positions_per_sequence(dev, mode):
    """Round 3 (VERDICT r2 missing 4): masked_lm_labels as ANY (B, L) grid -- 0, 1 or several labelled positions per sequence,
    as modeling_rec.py:147-150 (CrossEntropyLoss(ignore_index=-1) over every position) accepts -- runs the head on the labelled
    rows only (cpt_batch.n_rows / row_seq).  Loss and every gradient against autograd over the oracle's all-position form."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.tiny()
    m = _model(cfg, 77, dev, mode)
    B, Lt, Li = 5, 20, 6
    b = synth.make_batch(B, cfg, seed=31, max_seq_len=Lt, img_seq_len=Li)
    grid = torch.full((B, Lt + Li), -1, dtype=torch.long)
    g = torch.Generator().manual_seed(3)
    for row, cols in ((0, (2, 7, 11)), (1, (5,)), (3, (1, 2, 3, 19)), (4, (22,))):       # sequence 2 has no label; 4 labels a region slot
        for c in cols:
            grid[row, c] = int(torch.randint(4, cfg.vocab_size, (1,), generator=g))
    d = {k: v.to(dev) for k, v in b.items()}
    loss, scores = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=grid.to(dev))
    loss.backward()
    assert scores.shape == (int((grid != -1).sum()), cfg.vocab_size)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    leaves = {k: t.clone().requires_grad_(True) for k, t in sd.items() if k != "cls.decoder.weight"}
    work = dict(leaves)
    work["cls.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    ref_loss, ref_scores = O.rec_mlm_cpt_forward(work, cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"],
                                                 masked_lm_labels=grid, img_feats=b["img_feats"])
    ref_loss.backward()
    ltol, gtol, stol = (1e-4, 2e-4, 1e-4) if mode == "fp32" else (3e-2, 8e-2, 5e-2)
    assert abs(loss.item() - float(ref_loss.detach())) < ltol, (loss.item(), float(ref_loss.detach()))
    picked = ref_scores.detach()[grid != -1]                       # row-major order of the grid = the order of our rows
    assert float((scores.cpu() - picked).abs().max()) < stol * max(1.0, float(picked.abs().max()))
    n = 0
    for name, prm in m.named_parameters():
        rg = leaves[name].grad if name in leaves else None
        if rg is None or float(rg.abs().max()) == 0.0:
            continue
        rel, mx = _rel(prm.grad, rg)
        assert rel < gtol or mx < (1e-9 if mode == "fp32" else 2e-6), (name, rel, mx)
        n += 1
    assert n > 30


@pytest.mark.parametrize("mode,Lt,Li,p", [("fp32", 20, 6, 0.0), ("bf16", 20, 6, 0.0), ("fp32", 100, 45, 0.0), ("bf16", 165, 45, 0.0), ("fp32", 20, 6, 0.1),
                                            ("bf16x3", 20, 6, 0.1)])     # bf16x3: per-query masks go through the fp32 attention kernels, the GEMMs stay split
def test_training_with_a_three_dimensional_attention_mask(dev, mode, Lt, Li, p):
    """Round 3: attention_mask (B, L, L), one mask row per query (modeling_bert.py:215-216), in the TRAINING step:
    the forward attention kernel reads it per query, the backward runs the generic kernels (they read the mask per score; the MFMA
    kernels hold one value per key).  Loss and every gradient against autograd over the oracle (with the exported dropout masks when
    p > 0); L = 145 and 210 take the long-sequence instantiations."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.tiny(max_position_embeddings=max(96, Lt))
    m = _model(cfg, 41, dev, mode, dropout=p)
    B = 3
    b = synth.make_batch(B, cfg, seed=6, max_seq_len=Lt, img_seq_len=Li, vary_regions=True)
    Lq = Lt + Li
    rng = np.random.Generator(np.random.PCG64(12))
    per_q = torch.from_numpy((rng.random((B, Lq, Lq)) < 0.7).astype(np.int64)) * b["attention_mask"][:, None, :]
    per_q[:, torch.arange(Lq), torch.arange(Lq)] = 1
    d = {k: v.to(dev) for k, v in b.items()}
    drop = None
    if p > 0:
        from cpt_amd import train as T
        from tests.test_gpu_dropout import _drop_dict, SEED
        T.set_dropout_seed(m, SEED, step=0)     # the key test_gpu_dropout's mask export uses; the first forward is step 1
    loss, _ = m(d["input_ids"], d["segment_ids"], per_q.to(dev), img_feats=d["img_feats"], masked_lm_labels=d["colors"],
                mask_token_pos=d["mask_token_pos"])
    loss.backward()
    if p > 0:
        drop = _drop_dict(dev, cfg, p, 1, B, Lq)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    leaves = {k: t.clone().requires_grad_(True) for k, t in sd.items() if k != "cls.decoder.weight"}
    work = dict(leaves)
    work["cls.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    grid = torch.full((B, Lq), -1, dtype=torch.long)
    grid[torch.arange(B), b["mask_token_pos"]] = b["colors"]
    ref_loss, _ = O.rec_mlm_cpt_forward(work, cfg.to_dict(), b["input_ids"], b["segment_ids"], per_q, masked_lm_labels=grid,
                                        img_feats=b["img_feats"], mask_rows_only=b["mask_token_pos"], drop=drop)
    ref_loss.backward()
    ltol, gtol = {"fp32": (2e-4, 2e-4), "bf16x3": (2e-4, 5e-4), "bf16": (4e-2, 8e-2)}[mode]
    assert abs(loss.item() - float(ref_loss.detach())) < ltol, (loss.item(), float(ref_loss.detach()))
    two = O.rec_mlm_cpt_forward(sd | {"cls.decoder.weight": sd["bert.embeddings.word_embeddings.weight"]}, cfg.to_dict(), b["input_ids"],
                                b["segment_ids"], b["attention_mask"], masked_lm_labels=grid, img_feats=b["img_feats"],
                                mask_rows_only=b["mask_token_pos"])[0]
    if p == 0:
        assert abs(float(two) - float(ref_loss.detach())) > 1e-4          # the per-query mask matters
    n = 0
    for name, prm in m.named_parameters():
        rg = leaves[name].grad if name in leaves else None
        if rg is None or float(rg.abs().max()) == 0.0:
            continue
        rel, mx = _rel(prm.grad, rg)
        assert rel < gtol or mx < {"fp32": 1e-9, "bf16x3": 1e-8, "bf16": 2e-6}[mode], (name, rel, mx)
        n += 1
    assert n > 30


@pytest.mark.ablation
def test_forward_ffn_down_split_in_two_matches_the_unsplit_form(dev):
    """Round 3: at 2048..6144 rows the training forward runs the FFN-down on 128 x 192 tiles with K split over two workgroups and lets the
    dropout + residual + LayerNorm pass add the two partial matrices (cpt_set_tuning key 22).  Against the 64 x 192 form over the whole K
    (key 22 = 0): the same bf16 products in another fp32 summation order -- loss to 1e-4, gradients to the bf16 band."""
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base(num_hidden_layers=4)
    m = _model(cfg, 29, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(32, cfg, seed=19).items()}
    params = dict(m.named_parameters())

    def run(v):
        L.check(L.lib().cpt_set_tuning(22, v), "cpt_set_tuning")
        T.set_dropout_seed(m, 23)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return loss.item(), {n: p.grad.double().clone() for n, p in params.items() if p.grad is not None}

    l0, g0 = run(0)
    l1, g1 = run(1)
    assert abs(l0 - l1) < 1e-4 * max(1.0, abs(l0)), (l0, l1)
    assert l0 != l1 or any(not torch.equal(g0[n], g1[n]) for n in g0)          # (the split path really ran: some bit differs)
    for n in g0:
        if ".key.bias" in n:
            continue
        rel = float((g0[n] - g1[n]).norm()) / (float(g0[n].norm()) + 1e-30)
        assert rel < 2e-2, (n, rel)


@pytest.mark.parametrize("correct_bias", [True, False])
def test_hf_adamw_kernel_and_drop_in_optimizer(dev, correct_bias):
    """Round 5: pytorch_transformers.AdamW, the optimizer of the GQA / VCR few-shot drivers (fewshot/vcr_nsp_cpt.py:385, gqa_cpt.py:342).
    (1) cpt_adamw_ex(CPT_ADAMW_HF [| NO_BIAS_CORRECTION]) on flat buffers with all three element codes and the bf16 shadow against the oracle's
    restatement over three steps; (2) train.AdamW + WarmupLinearSchedule on the tiny model: after each of three scheduled steps every parameter
    equals adamw_step_hf applied to the gradients the step produced, weight decay on everything but bias / LayerNorm parameters."""
    import ctypes as C
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    from oracle import cpt_oracle as O
    g = torch.Generator().manual_seed(3)
    n = 4096 * 3
    p = torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    code = torch.randint(0, 3, (n,), generator=g, dtype=torch.uint8)
    lr, b1, b2, eps, wd = 3e-3, 0.9, 0.999, 1e-6, 0.05
    flags = L.ADAMW_HF | (0 if correct_bias else L.ADAMW_NO_BIAS_CORRECTION)
    pd, md, vd, cd = p.to(dev), m.to(dev), v.to(dev), code.to(dev)
    sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        gr = torch.randn(n, generator=g) * (0.1 * step)
        gd = gr.to(dev)
        L.check(L.lib().cpt_adamw_ex(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), cd.data_ptr(), sh.data_ptr(), n, lr, b1, b2, eps, wd,
                                     step, 0.5, flags, L.stream_ptr()), "cpt_adamw_ex")
        pn, mn, vn = O.adamw_step_hf(p, gr * 0.5, m, v, step, lr, b1, b2, eps, wd, correct_bias)
        pn0, _, _ = O.adamw_step_hf(p, gr * 0.5, m, v, step, lr, b1, b2, eps, 0.0, correct_bias)
        upd = code != 0
        p = torch.where(code == 1, pn, torch.where(code == 2, pn0, p))
        m, v = torch.where(upd, mn, m), torch.where(upd, vn, v)
        assert (pd.cpu() - p).abs().max().item() < 2e-6 and (md.cpu() - m).abs().max().item() < 1e-6 and (vd.cpu() - v).abs().max().item() < 1e-6
        assert torch.equal(sh.float().cpu(), pd.cpu().to(torch.bfloat16).float())
    # (2) the drop-in optimizer + schedule on the tiny model
    cfg = cfgmod.tiny()
    mdl = _model(cfg, 5, dev, "fp32")
    opt = T.AdamW(mdl, lr=2e-3, eps=1e-6, weight_decay=0.05, correct_bias=correct_bias)
    sched = T.WarmupLinearSchedule(opt, warmup_steps=2, t_total=6)
    assert len(opt.param_groups) == 2 and opt.param_groups[0]["lr"] == 0.0
    b = {k: t.to(dev) for k, t in synth.make_batch(4, cfg, seed=8, max_seq_len=20, img_seq_len=6).items()}
    names = dict(mdl.named_parameters())
    st = {k: (torch.zeros_like(t, device="cpu"), torch.zeros_like(t, device="cpu")) for k, t in names.items()}
    for step in (1, 2, 3):
        sched.step()
        lr_now = opt.param_groups[0]["lr"]
        assert abs(lr_now - 2e-3 * O.warmup_linear_schedule(step, 2, 6)) < 1e-12
        before = {k: t.detach().cpu().clone() for k, t in names.items()}
        opt.zero_grad()
        loss, _ = mdl(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        grads = {k: (t.grad.detach().cpu().clone() if t.grad is not None else None) for k, t in names.items()}
        opt.step()
        for k, t in names.items():
            if grads[k] is None or k == "cls.decoder.weight":          # (no gradient on this path: untouched; the tied table is checked under its embedding name)
                continue
            w = 0.0 if any(nd in k for nd in T.NO_DECAY) else 0.05
            pn, mn, vn = O.adamw_step_hf(before[k], grads[k], st[k][0], st[k][1], step, lr_now, 0.9, 0.999, 1e-6, w, correct_bias)
            st[k] = (mn, vn)
            assert (t.detach().cpu() - pn).abs().max().item() < 3e-6, (step, k)


@pytest.mark.parametrize("B", [4, 32])
def test_last_layer_on_head_rows_matches_the_all_rows_form(dev, B):
    """Round 6: with one head row per sequence the training step runs the last encoder layer behind its attention (attention output, FFN, both
    LayerNorms; forward and backward) on the B [MASK] rows only.  The SAME batch handed over as a label grid (cpt_batch.n_rows = B, row_seq =
    0..B-1) takes the all-rows path of the same library: loss, scores and every gradient of the two must agree (same products, other tile
    splits; dropout 0.1 -- the compact passes regenerate the masks at the rows' positions in the full tensor).  Product library, no switches."""
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base()
    m = _model(cfg, 31, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=15).items()}
    params = dict(m.named_parameters())

    def run(grid):
        T.set_dropout_seed(m, 23)
        for p in m.parameters():
            p.grad = None
        seq = torch.arange(B, device=dev) if grid else None
        loss, scores = T.mlm_loss_with_grad(m, b["input_ids"], b["segment_ids"], b["attention_mask"], b["colors"], None, b["img_feats"],
                                            b["mask_token_pos"], row_seq=seq)
        loss.backward()
        return loss.item(), scores.detach().double().clone(), {n: p.grad.double().clone() for n, p in params.items() if p.grad is not None}

    l0, s0, g0 = run(True)        # all rows
    l1, s1, g1 = run(False)       # head rows
    assert abs(l0 - l1) < 2e-3 * max(1.0, abs(l0)), (l0, l1)
    assert float((s0 - s1).abs().max()) < 2e-2, float((s0 - s1).abs().max())
    assert set(g0) == set(g1) and len(g0) > 190
    worst = ("", 0.0)
    for n in g0:
        if n.endswith("attention.self.key.bias"):      # zero in exact arithmetic (a constant added to every score of a query): rounding noise on both sides
            continue
        den = float(g0[n].norm())
        assert den > 0, n
        rel = float((g1[n] - g0[n]).norm()) / den
        if rel > worst[1]:
            worst = (n, rel)
    # bf16 activations are rounded after differently split sums: a few 1e-3 on the smallest gradients
    assert worst[1] < 1e-2, worst


def test_development_library_variant_comparisons_in_a_subprocess():
    """the `ablation`-marked tests (kernel VARIANTS against each other through cpt_set_tuning) need the development build of
    the library, which the driver's `pytest -m gpu` does not load -- so this test runs them in a child process with CPT_AMD_ABLATION=1 and
    fails when any of them fails.  The child loads libcpt_hip_abl.so; this process keeps the product library."""
    import subprocess
    import sys
    from cpt_amd import _lib as L
    if L.ablation_build():
        pytest.skip("already running on the development library: the marked tests run in this process")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "cpt_amd", "libcpt_hip_abl.so")):
        pytest.fail("cpt_amd/libcpt_hip_abl.so is missing: __graft_entry__.build() builds it (python -m cpt_amd.build --ablation)")
    env = dict(os.environ, CPT_AMD_ABLATION="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests"), "-q", "-x", "-m", "gpu and ablation", "-p", "no:cacheprovider"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail


def test_hf_adamw_kernel_against_the_float64_vectors(dev, golden_dir):
    """cpt_adamw_ex(CPT_ADAMW_HF) against the golden vectors of the INDEPENDENT float64 restatement of pytorch_transformers.AdamW
    (oracle/make_hf_adamw_fixture.py; tests/golden/META.json says what that pin is and is not): six steps, three hyper-parameter sets."""
    from cpt_amd import _lib as L
    g = np.load(os.path.join(golden_dir, "hf_adamw.npz"))
    for tag in ("gqa", "vcr", "nobias"):
        b1, b2, eps, wd, cb = [float(x) for x in g[tag + "_hyper"]]
        flags = L.ADAMW_HF | (0 if cb else L.ADAMW_NO_BIAS_CORRECTION)
        n = g[tag + "_p0"].shape[0]
        pd = torch.from_numpy(g[tag + "_p0"]).float().to(dev)
        md, vd = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        cd = torch.ones(n, device=dev, dtype=torch.uint8)          # code 1: decayed parameter
        for t in range(1, g[tag + "_g"].shape[0] + 1):
            gd = torch.from_numpy(g[tag + "_g"][t - 1]).float().to(dev)
            L.check(L.lib().cpt_adamw_ex(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), cd.data_ptr(), None, n, float(g[tag + "_lr"][t - 1]),
                                         b1, b2, eps, wd, t, 1.0, flags, L.stream_ptr()), "cpt_adamw_ex")
        for got, key, tol in ((pd, "_p", 4e-6), (md, "_m", 4e-6), (vd, "_v", 4e-6)):
            ref = torch.from_numpy(g[tag + key][-1])
            err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
            assert err < tol, (tag, key, err)


@pytest.mark.ablation
def test_bf16_partial_data_gradients_match_the_fp32_epilogue_form(dev):
    """Round 6: at 2048..6144 rows the data-gradient GEMMs in front of a LayerNorm backward (FFN-up, Q|K|V) run 128 x 192 tiles with K split in two and hand the
    LayerNorm backward two BF16 partial matrices (cpt_set_tuning key 34, default) instead of one fp32 matrix from 64 x 192 tiles with the residual in the
    epilogue (34 = 0).  The GEMM part of that gradient is then rounded to bf16 twice before the fp32 residual is added: every gradient of the step against the
    fp32-epilogue form, Oscar-base at 32 sequences, dropout on."""
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base(num_hidden_layers=4)
    m = _model(cfg, 27, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(32, cfg, seed=29).items()}
    params = dict(m.named_parameters())

    def grads(v):
        L.check(L.lib().cpt_set_tuning(34, v), "cpt_set_tuning")
        T.set_dropout_seed(m, 31)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return {n: p.grad.double().clone() for n, p in params.items() if p.grad is not None}

    ref = grads(0)
    got = grads(1)
    worst = ("", 0.0)
    for n in ref:
        if n.endswith("attention.self.key.bias"):
            continue
        rel = float((got[n] - ref[n]).norm()) / (float(ref[n].norm()) + 1e-30)
        if rel > worst[1]:
            worst = (n, rel)
    assert worst[1] < 1e-2, worst          # 2^-9 per partial on the GEMM part of dx, averaged over the contraction of everything downstream


@pytest.mark.ablation
def test_deferred_partial_sums_and_lean_layernorm_leave_the_gradients_unchanged(dev):
    """Round 6: (a) the K-split partial matrices of a layer's Q|K|V weight gradient are added up by the spare workgroups of the NEXT layer's three-problem
    weight-gradient launch (cpt_set_tuning key 37, default) instead of a reduction launch -- the same additions in the same order: every gradient BIT-identical;
    (b) the training forward's LayerNorm launches write no fp32 output and the next row pass re-forms the residual from the kept pre-LayerNorm rows (key 36):
    the same expression on the same fp32 operands -- loss and gradients within fp32 rounding.  Oscar-base (4 layers) at 32 sequences (3840 rows: the split and
    the pruned last layer are both on), dropout on."""
    from cpt_amd import _lib as L
    from cpt_amd import train as T
    cfg = cfgmod.oscar_base(num_hidden_layers=4)
    m = _model(cfg, 41, dev, "bf16", dropout=0.1)
    b = {k: v.to(dev) for k, v in synth.make_batch(32, cfg, seed=43).items()}
    params = dict(m.named_parameters())

    def grads(key, v):
        L.check(L.lib().cpt_set_tuning(-1, 0), "cpt_set_tuning")
        L.check(L.lib().cpt_set_tuning(key, v), "cpt_set_tuning")
        T.set_dropout_seed(m, 47)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in params.items() if p.grad is not None}

    l0, ref = grads(37, 0)
    l1, got = grads(37, 1)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)       # (the row losses are added with atomics: the last bit moves run to run)
    for n in ref:
        if "encoder.layer" in n and n.endswith(".weight") and "LayerNorm" not in n:      # the Linear weights of the encoder: written by GEMMs, no atomics anywhere upstream
            assert torch.equal(ref[n], got[n]), n
        else:      # (vectors and embedding tables take atomic adds: not bit-reproducible run to run either way)
            if n.endswith("attention.self.key.bias"):      # (mathematically zero: rounding noise only)
                continue
            assert float((ref[n].double() - got[n].double()).norm()) <= 1e-4 * float(ref[n].double().norm()) + 1e-12, n
    l2, lean0 = grads(36, 0)
    assert abs(l2 - l1) <= 1e-5 * abs(l1)
    worst = ("", 0.0)
    for n in got:
        if n.endswith("attention.self.key.bias"):
            continue
        rel = float((got[n].double() - lean0[n].double()).norm()) / (float(lean0[n].double().norm()) + 1e-30)
        if rel > worst[1]:
            worst = (n, rel)
    assert worst[1] < 2e-3, worst          # (a last-bit change of an fp32 residual moves bf16 roundings downstream)
    L.check(L.lib().cpt_set_tuning(-1, 0), "cpt_set_tuning")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_training_loss_is_the_mean_over_labelled_rows_written_by_the_cross_entropy_launch(dev, mode):
    """Round 6 (ABI 8, cpt_outputs.loss_mean): the training forward's loss comes out of the cross-entropy launch itself (its last workgroup divides the totals and
    leaves {sum, count} in the workspace for the backward; the embedding launch clears the accumulators) -- no memset, copy or divide launch.  Against
    torch's CrossEntropyLoss(ignore_index=-1) on the returned scores (modeling_rec.py:147-150): some rows ignored; the same call twice (the finish ticket
    resets itself); every row ignored -> NaN, as the reference's mean over no rows; and the gradient scale 1 / count reaches the backward."""
    cfg = cfgmod.tiny()
    m = _model(cfg, 5, dev, mode)
    b = {k: v.to(dev) for k, v in synth.make_batch(6, cfg, seed=9, max_seq_len=20, img_seq_len=6).items()}
    labels = b["colors"].clone()
    labels[1] = -1
    labels[4] = -1

    def run(lab):
        for p in m.parameters():
            p.grad = None
        loss, scores = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=lab, mask_token_pos=b["mask_token_pos"])
        return loss, scores

    for _ in range(2):
        loss, scores = run(labels)
        ref = torch.nn.functional.cross_entropy(scores.float(), labels, ignore_index=-1)
        assert abs(float(loss.detach()) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    loss.backward()
    g4 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    loss6, _ = run(b["colors"])
    loss6.backward()
    g6 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    n = "cls.transform.dense.weight" if "cls.transform.dense.weight" in g4 else next(k for k in g4 if k.endswith("transform.dense.weight"))
    assert float((g4[n] - g6[n]).abs().max()) > 0      # (a different row set and 1 / 4 instead of 1 / 6)
    none = torch.full_like(labels, -1)
    loss0, _ = run(none)
    assert torch.isnan(loss0.detach()).item()
