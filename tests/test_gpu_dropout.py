"""GPU: dropout of the training step (cpt_train_fwd_ex / cpt_train_bwd_ex with a cpt_dropout) -- the reference trains
with nn.Dropout(p=0.1) on the attention probabilities (/root/reference/Oscar/oscar/modeling/modeling_bert.py:57), the
region embeddings (:266) and inside BertEmbeddings / BertSelfOutput / BertOutput (fewshot/refcoco_cpt.py:387,509-512).

Bitwise parity with torch's RNG stream is impossible, so the checks are:
  * the exported masks equal a CPU restatement of the counter scheme (oracle.dropout_keep_*; its Philox4x32-10 is pinned
    by the published Random123 known-answer vectors in tests/test_oracle_golden.py) bit for bit, keep rate within 4 sigma;
  * with THOSE masks fed to the oracle, loss and every gradient agree (fp32 mode to 2e-4 relative, bf16 in its band),
    at L = 26 (one key block) and L = 100 (four key blocks: every mask lane mapping of the MFMA kernels);
  * p = 0 runs the dropout-free kernels (bit-identical), a fixed seed reproduces, another step / seed does not.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from cpt_amd import config as cfgmod
from cpt_amd import synth
from oracle import cpt_oracle as O

pytestmark = pytest.mark.gpu
SEED = 0x5EEDC0FFEE123457


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(cfg, dev, mode, p):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    from cpt_amd.train import set_dropout_seed
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = p
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 1234, head="cpt"))
    m.tie_weights()
    m.to(dev).train()
    m.set_compute_dtype(mode)
    set_dropout_seed(m, SEED, step=0)
    return m


def _export(dev, p, step, site, attn, n0, n1, n2=0):
    from cpt_amd import _lib as L
    d = L.Dropout(p_hidden=p, p_attn=p, seed=SEED, step=step)
    out = torch.empty((n0, n1, n2) if attn else (n0, n1), dtype=torch.uint8, device=dev)
    L.check(L.lib().cpt_dropout_mask(C.byref(d), site, 1 if attn else 0, out.data_ptr(), n0, n1, n2, L.stream_ptr()), "cpt_dropout_mask")
    return out.cpu()


def _drop_dict(dev, cfg, p, step, B, L):
    """Multiplier tensors for the oracle from the masks the LIBRARY exports for (SEED, step)."""
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    _, sh = O.dropout_thresh_scale(p, False)
    _, sa = O.dropout_thresh_scale(p, True)
    sh, sa = np.float32(sh), np.float32(sa)
    drop = {"emb": _export(dev, p, step, 0, False, B * L, H).view(B, L, H).float() * float(sh)}
    for i in range(cfg.num_hidden_layers):
        drop[("attn", i)] = _export(dev, p, step, 1 + 3 * i, True, B * nh, L, L).view(B, nh, L, L).float() * float(sa)
        drop[("ao", i)] = _export(dev, p, step, 2 + 3 * i, False, B * L, H).view(B, L, H).float() * float(sh)
        drop[("out", i)] = _export(dev, p, step, 3 + 3 * i, False, B * L, H).view(B, L, H).float() * float(sh)
    return drop


def test_exported_masks_match_cpu_restatement_and_rate(dev):
    p = 0.1
    for step, site in ((1, 0), (7, 5)):
        got = _export(dev, p, step, site, False, 96, 128).numpy()
        ref = O.dropout_keep_hidden(SEED, step, site, 96, 128, p)
        assert (got == ref).all()
    for step, site in ((1, 1), (3, 4)):
        got = _export(dev, p, step, site, True, 6, 45, 45).numpy()
        ref = O.dropout_keep_attn(SEED, step, site, 6, 45, p)
        assert (got == ref).all()
    big = _export(dev, p, 2, 3, False, 4096, 768).float()
    n = big.numel()
    assert abs(big.mean().item() - 0.9) < 4 * (0.09 / n) ** 0.5
    att = _export(dev, 0.3, 2, 4, True, 96, 120, 120).float()          # GQA / VCR drivers use 0.3
    assert abs(att.mean().item() - (1 - round(0.3 * 65536) / 65536)) < 4 * (0.21 / att.numel()) ** 0.5
    # different step / site / seed -> different masks
    a = _export(dev, p, 1, 0, False, 64, 128)
    assert not torch.equal(a, _export(dev, p, 2, 0, False, 64, 128))
    assert not torch.equal(a, _export(dev, p, 1, 2, False, 64, 128))


# (165, 45) and (165, 100): the GQA and VCR few-shot sequence lengths (Oscar/cmds/gqa/_cpt_fsl_base.sh:19,27; BASELINE config 5):
# bf16 runs the transpose-read MFMA attention backward (csrc/bwd.hip, 128 < L <= 288), fp32 at L = 145 the generic kernel,
# fp32 at L = 210 / 265 its 16-query-block form that reads V from global memory
@pytest.mark.parametrize("mode,Lt,Li", [("fp32", 20, 6), ("bf16", 20, 6), ("fp32", 60, 40), ("bf16", 60, 40),
                                        ("fp32", 100, 45), ("bf16", 100, 45), ("bf16", 165, 45), ("bf16", 165, 100),
                                        ("fp32", 165, 45), ("fp32", 165, 100),      # fp32 beyond L = 176: V read from global memory (round 3)
                                        # bf16x3 training (round 4): the split-operand MFMA attention backward -- all eight tiles in LDS at L <= 128,
                                        # the two-phase form that keeps two tensors per phase in LDS at L = 145 / 210 / 265
                                        ("bf16x3", 20, 6), ("bf16x3", 60, 40), ("bf16x3", 100, 45), ("bf16x3", 165, 45), ("bf16x3", 165, 100)])
def test_loss_and_gradients_match_oracle_with_the_same_masks(dev, mode, Lt, Li):
    p = 0.1
    cfg = cfgmod.tiny(max_position_embeddings=max(96, Lt))
    m = _model(cfg, dev, mode, p)
    B = 3
    b = synth.make_batch(B, cfg, seed=5, max_seq_len=Lt, img_seq_len=Li, vary_regions=True)
    d = {k: v.to(dev) for k, v in b.items()}
    loss, _ = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=d["colors"],
                mask_token_pos=d["mask_token_pos"])
    loss.backward()
    drop = _drop_dict(dev, cfg, p, 1, B, Lt + Li)                        # first training forward -> step 1
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd["cls.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    ref_loss, ref = O.train_step_grads(sd, cfg.to_dict(), b, drop=drop)
    ltol, gtol = {"fp32": (2e-4, 2e-4), "bf16x3": (2e-4, 5e-4), "bf16": (4e-2, 8e-2)}[mode]
    assert abs(loss.item() - float(ref_loss)) < ltol, (loss.item(), float(ref_loss))
    # and the loss differs from the dropout-free one (the masks really were applied)
    free, _ = O.train_step_grads(sd, cfg.to_dict(), b)
    assert abs(float(free) - float(ref_loss)) > 2e-4
    worst = 0.0
    for name, prm in m.named_parameters():
        g = ref.get(name)
        if g is None:
            continue
        got = prm.grad.double().cpu().flatten()
        rf = g.double().flatten()
        rel = float((got - rf).norm() / (rf.norm() + 1e-30))
        if float(rf.abs().max()) < 1e-6 and float(got.abs().max()) < (1e-3 if mode == "bf16" else 1e-5):
            continue          # key bias: its true gradient is 0 (softmax is shift-invariant), both sides hold rounding noise
        worst = max(worst, rel)
        assert rel < gtol, (name, rel)
    print("dropout %s L=%d: loss %.5f (oracle %.5f), worst relative gradient error %.2e" % (mode, Lt + Li, loss.item(), float(ref_loss), worst))


def test_p0_is_bit_identical_and_seed_reproduces(dev):
    cfg = cfgmod.tiny()
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=9, max_seq_len=20, img_seq_len=6).items()}

    def run(m, steps=1):
        out = []
        for _ in range(steps):
            for prm in m.parameters():
                prm.grad = None
            loss, logits = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                             mask_token_pos=b["mask_token_pos"])
            loss.backward()
            # tensors whose bits are run-to-run deterministic: the [MASK]-row logits and GEMM-produced weight gradients (the
            # scalar loss, the embedding-table and LayerNorm gain/bias gradients are accumulated with fp32 atomics)
            out.append((logits.detach().clone(), m.bert.encoder.layer[0].attention.self.query.weight.grad.clone(),
                        m.bert.encoder.layer[1].intermediate.dense.weight.grad.clone()))
        return out
    for mode in ("fp32", "bf16"):
        m0 = _model(cfgmod.tiny(), dev, mode, 0.0)
        a = run(m0)[0]
        m0.eval()                                                      # eval mode with p > 0 in the config: also no dropout
        m0.config.hidden_dropout_prob = m0.config.attention_probs_dropout_prob = 0.1
        for prm in m0.parameters():
            prm.requires_grad_(True)
        e = run(m0)[0]
        assert all(torch.equal(x, y) for x, y in zip(a, e))
        m1 = _model(cfgmod.tiny(), dev, mode, 0.1)
        r1 = run(m1, 2)
        m2 = _model(cfgmod.tiny(), dev, mode, 0.1)
        r2 = run(m2, 2)
        for s in range(2):
            assert all(torch.equal(x, y) for x, y in zip(r1[s], r2[s]))            # same seed, same step: same bits
        assert not torch.equal(r1[0][1], r1[1][1])                                   # next step: fresh masks
        assert not torch.equal(r1[0][1], a[1])                                       # and not the dropout-free gradient
