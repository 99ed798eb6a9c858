"""Edges of the hot path's input space, HIP path against the oracle (empty and ragged inputs, maximum sizes):
text-only sequences (img_feats = None, modeling_bert.py:261), the smallest batch there is, the longest sequence the kernels take and
the first one they refuse, sequences whose keys are ALL masked, explicit position ids and the last row of every table."""
import pytest
import torch

from cpt_amd import config as cfgmod
from cpt_amd import synth

pytestmark = pytest.mark.gpu
TOL = {"fp32": 1e-3, "bf16x3": 1e-3, "bf16": 0.04}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(cfg, seed, dev, dtype, train=False):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, seed, head="cpt"))
    m.tie_weights()
    m.to(dev)
    m.train() if train else m.eval()
    m.set_compute_dtype(dtype)
    return m


def _sd(m):
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd["cls.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd


def _oracle_logits(m, cfg, b, **kw):
    from oracle import cpt_oracle as O
    with torch.no_grad():
        return O.rec_mlm_cpt_forward(_sd(m), cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"],
                                     img_feats=b.get("img_feats"), mask_rows_only=b["mask_token_pos"], **kw)[0]


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_text_only_sequences(dev, mode):
    """No region features at all (img_feats = None): the encoder runs on the text rows alone, inference and one training step."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.tiny()
    b = synth.make_batch(5, cfg, seed=3, max_seq_len=24, img_seq_len=4)
    b = {k: (v[:, :24] if k == "attention_mask" else v) for k, v in b.items() if k != "img_feats"}
    m = _model(cfg, 41, dev, mode)
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=None, mask_token_pos=d["mask_token_pos"])[0]
    want = _oracle_logits(m, cfg, b)
    assert got.shape == want.shape
    assert (got.float().cpu() - want).abs().max().item() < TOL[mode]
    # training on text alone: loss and every gradient the oracle's autograd produces (the region projection gets none)
    m.train()
    loss, _ = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=None, masked_lm_labels=d["colors"], mask_token_pos=d["mask_token_pos"])
    loss.backward()
    ref_loss, ref = O.train_step_grads(_sd(m), cfg.to_dict(), dict(b, img_feats=None))
    ltol, gtol = {"fp32": (2e-4, 2e-4), "bf16x3": (2e-4, 5e-4), "bf16": (4e-2, 8e-2)}[mode]
    assert abs(loss.item() - float(ref_loss)) < ltol
    n = 0
    for name, prm in m.named_parameters():
        g = ref.get(name)
        if g is None or float(g.abs().max()) < 1e-6:
            assert prm.grad is None or float(prm.grad.abs().max()) < (1e-3 if mode == "bf16" else 1e-5), name
            continue
        rel = float((prm.grad.double().cpu() - g.double()).norm() / g.double().norm())
        assert rel < gtol, (name, rel)
        n += 1
    assert n > 25
    assert m.bert.img_embedding.weight.grad is None or float(m.bert.img_embedding.weight.grad.abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_one_sequence_of_minimal_length(dev, mode):
    """B = 1 with a two-token text ([CLS]-like id + [MASK]) and a single region: the smallest input the reference accepts."""
    cfg = cfgmod.tiny()
    m = _model(cfg, 43, dev, mode)
    g = torch.Generator().manual_seed(9)
    V = cfg.vocab_size
    b = {"input_ids": torch.tensor([[min(synth.CLS, V - 2), synth.MASK if synth.MASK < V else 3]]), "segment_ids": torch.zeros(1, 2, dtype=torch.long),
         "attention_mask": torch.ones(1, 3, dtype=torch.long), "mask_token_pos": torch.tensor([1]),
         "img_feats": torch.rand(1, 1, cfg.img_feature_dim, generator=g)}
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
    want = _oracle_logits(m, cfg, b)
    assert got.shape == (1, cfg.vocab_size)
    assert (got.float().cpu() - want).abs().max().item() < TOL[mode]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_longest_mfma_sequence_and_the_first_beyond_it(dev, mode):
    """L = 288 (188 text + 100 regions) is the longest sequence the MFMA attention kernels take (score strip in registers); from L = 289 on the
    inference path runs the one-wave-per-query coverage kernel (round 5) -- same numbers against the oracle -- and the TRAINING step refuses
    with an error that says so, before anything is launched."""
    cfg = cfgmod.tiny(max_position_embeddings=192)
    m = _model(cfg, 47, dev, mode)
    for Lt in (188, 189):
        b = synth.make_batch(2, cfg, seed=6, max_seq_len=Lt, img_seq_len=100, vary_regions=True)
        d = {k: v.to(dev) for k, v in b.items()}
        with torch.no_grad():
            got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
        want = _oracle_logits(m, cfg, b)
        assert (got.float().cpu() - want).abs().max().item() < TOL[mode], Lt
    m.train()
    with pytest.raises(RuntimeError, match="288"):
        m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=d["colors"], mask_token_pos=d["mask_token_pos"])
    m.eval()


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("Lt,Li", [(300, 50), (512, 100)])
def test_sequences_as_long_as_the_reference_accepts(dev, mode, Lt, Li):
    """The reference takes up to max_position_embeddings = 512 text positions plus the region slots (modeling_bert.py:244-269).  L = 350 and L = 612
    through the whole inference path against the oracle: logits of the [MASK] rows, a ragged 2-D mask, and (fp32) a 3-D mask on a short batch."""
    cfg = cfgmod.tiny(max_position_embeddings=512)
    m = _model(cfg, 49, dev, mode)
    b = synth.make_batch(2, cfg, seed=7, max_seq_len=Lt, img_seq_len=Li, vary_regions=True)
    b["attention_mask"][1, Lt - 37:Lt] = 0
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
    want = _oracle_logits(m, cfg, b)
    assert torch.isfinite(got).all()
    assert (got.float().cpu() - want).abs().max().item() < TOL[mode]
    if mode == "fp32" and Lt == 300:
        L = Lt + Li
        m3 = torch.tril(torch.ones(L, L, dtype=torch.long)).unsqueeze(0).repeat(2, 1, 1) * b["attention_mask"][:, None, :]
        b3 = dict(b, attention_mask=m3)
        with torch.no_grad():
            got3 = m(d["input_ids"], d["segment_ids"], m3.to(dev), img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
        assert (got3.float().cpu() - _oracle_logits(m, cfg, b3)).abs().max().item() < TOL[mode]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_sequence_with_every_key_masked(dev, mode):
    """attention_mask all zero for one sequence: every score gets -10000 and the softmax is the plain softmax of the raw scores
    (modeling_bert.py:213-218, 53-58 add the mask, they do not exclude keys) -- same numbers as the oracle, nothing NaN, and the other
    sequences of the batch are untouched."""
    cfg = cfgmod.tiny()
    m = _model(cfg, 53, dev, mode)
    b = synth.make_batch(4, cfg, seed=8, max_seq_len=20, img_seq_len=6)
    ref_rows = _oracle_logits(m, cfg, b)
    b["attention_mask"][2] = 0
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0].float().cpu()
    want = _oracle_logits(m, cfg, b)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() < TOL[mode]
    keep = [0, 1, 3]
    assert (got[keep] - ref_rows[keep]).abs().max().item() < TOL[mode]


def test_explicit_position_ids_and_last_table_rows(dev):
    """position_ids given explicitly (modeling_bert.py:244 passes them through), pointing at the LAST row of the position table; token
    ids at the last row of the vocabulary; token types at the last type: gathers at the table ends read the right rows."""
    from oracle import cpt_oracle as O
    cfg = cfgmod.tiny()
    m = _model(cfg, 59, dev, "fp32")
    b = synth.make_batch(3, cfg, seed=10, max_seq_len=20, img_seq_len=6)
    b["input_ids"][:, 3] = cfg.vocab_size - 1
    b["segment_ids"][:, 5] = cfg.type_vocab_size - 1
    pos = torch.arange(20).repeat(3, 1)
    pos[:, -1] = cfg.max_position_embeddings - 1
    pos[1] = torch.flip(pos[1], dims=[0])
    d = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], position_ids=pos.to(dev), img_feats=d["img_feats"],
                mask_token_pos=d["mask_token_pos"])[0]
        want = O.rec_mlm_cpt_forward(_sd(m), cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"], position_ids=pos,
                                     img_feats=b["img_feats"], mask_rows_only=b["mask_token_pos"])[0]
        base = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
    assert (got.float().cpu() - want).abs().max().item() < 1e-3
    assert (got - base).abs().max().item() > 1e-3          # the explicit ids were used
