"""Operator-level backward parity (attention_bwd, bias_residual_ln_bwd, embed_ln_bwd): the kernels
cpt_train_bwd launches, called one by one through the C ABI (cpt_attention_bwd, cpt_layernorm_bwd, cpt_embed_ln_bwd) and compared with
autograd over the ORACLE's own blocks -- self_attention (modeling_bert.py:30-70), layer_norm (BertSelfOutput / BertOutput, :85-86, :145),
text_embeddings (BertEmbeddings, :244-245) -- with the dropout masks the library exports for the same (seed, step, site)."""
import ctypes as C

import numpy as np
import pytest
import torch

from cpt_amd import config as cfgmod
from oracle import cpt_oracle as O

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED1234


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _mask(dev, p, step, site, attn, n0, n1, n2=0):
    from cpt_amd import _lib as L
    d = L.Dropout(p_hidden=p, p_attn=p, seed=SEED, step=step)
    out = torch.empty((n0, n1, n2) if attn else (n0, n1), dtype=torch.uint8, device=dev)
    L.check(L.lib().cpt_dropout_mask(C.byref(d), site, 1 if attn else 0, out.data_ptr(), n0, n1, n2, L.stream_ptr()), "cpt_dropout_mask")
    return out.cpu()


def _rel(got, ref):
    return ((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("L", [120, 210, 265])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16x3"])
def test_attention_bwd_against_oracle_autograd(dev, mode, L, p):
    """dqkv of cpt_attention_bwd against autograd through oracle.self_attention.  The oracle block owns the Q / K / V projections, so the
    comparison runs one linear map further: from the kernel's dqkv the test forms dx = dqkv . W, dW = dqkv^T . x and db = column sums in
    fp64 and compares them with the oracle's x.grad / weight / bias gradients (x has full row rank, B L <= H: dW determines dqkv)."""
    from cpt_amd import _lib as Lb
    cfg = cfgmod.oscar_base()
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    B, step, layer = 2, 3, 4
    site = 1 + 3 * layer
    g = torch.Generator().manual_seed(100 * L + int(p * 10))
    x = torch.randn(B, L, H, generator=g).requires_grad_(True)
    pre = "bert.encoder.layer.%d.attention.self." % layer
    sd = {}
    for nm in ("query", "key", "value"):
        sd[pre + nm + ".weight"] = (torch.randn(H, H, generator=g) * 0.05).requires_grad_(True)
        sd[pre + nm + ".bias"] = (torch.randn(H, generator=g) * 0.1).requires_grad_(True)
    am = torch.ones(B, L, dtype=torch.long)
    am[1, L - 17:] = 0
    drop = None
    if p > 0:
        _, sa = O.dropout_thresh_scale(p, True)
        drop = {("attn", layer): _mask(dev, p, step, site, True, B * nh, L, L).view(B, nh, L, L).float() * float(np.float32(sa))}
    ctx = O.self_attention(sd, cfg.to_dict(), x, O.extended_mask(am), pre, drop, layer)
    dctx = torch.randn(B, L, H, generator=g)
    ctx.backward(dctx)
    with torch.no_grad():
        W = torch.cat([sd[pre + n + ".weight"] for n in ("query", "key", "value")], 0)          # [3H][H]
        bias = torch.cat([sd[pre + n + ".bias"] for n in ("query", "key", "value")], 0)
        qkv = torch.nn.functional.linear(x, W, bias).reshape(B * L, 3 * H)
    lp = mode == "bf16"
    tdt = torch.bfloat16 if lp else torch.float32
    q_d = qkv.to(tdt).to(dev).contiguous()
    d_d = dctx.reshape(B * L, H).to(tdt).to(dev).contiguous()
    am_d = am.to(dev)
    dq_d = torch.empty(B * L, 3 * H, dtype=tdt, device=dev)
    db_d = torch.zeros(3 * H, dtype=torch.float32, device=dev)
    dr = Lb.Dropout(p_hidden=p, p_attn=p, seed=SEED, step=step)
    code = {"fp32": Lb.CPT_F32, "bf16": Lb.CPT_BF16, "bf16x3": Lb.CPT_BF16X3}[mode]
    Lb.check(Lb.lib().cpt_attention_bwd(code, q_d.data_ptr(), am_d.data_ptr(), 0, d_d.data_ptr(), dq_d.data_ptr(), db_d.data_ptr(), B, L, nh,
                                        C.byref(dr) if p > 0 else None, site, Lb.stream_ptr()), "cpt_attention_bwd")
    dqkv = dq_d.float().cpu().double()
    x2 = x.detach().reshape(B * L, H).double()
    tol = {"fp32": 2e-4, "bf16x3": 1e-3, "bf16": 4e-2}[mode]
    gW = torch.cat([sd[pre + n + ".weight"].grad for n in ("query", "key", "value")], 0)
    gb = torch.cat([sd[pre + n + ".bias"].grad for n in ("query", "key", "value")], 0)
    assert _rel(dqkv @ W.double(), x.grad.reshape(B * L, H)) < tol
    assert _rel(dqkv.t() @ x2, gW) < tol
    assert _rel(dqkv.sum(0), gb) < tol
    assert _rel(db_d.cpu(), gb) < tol                     # the kernel's own bias-gradient sums


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("lp", ["bf16", "fp32"])
@pytest.mark.parametrize("R", [480, 3840])
def test_layernorm_bwd_against_oracle_autograd(dev, R, lp, p):
    """y = layer_norm(dropout(o) + resid) as BertSelfOutput / BertOutput compute it: dx (= the residual's gradient), dgamma, dbeta, and -- with
    dropout -- the masked gradient of the dense output with its column sums, against autograd over oracle.layer_norm with the exported mask.
    With and without the two-stage column-sum scratch."""
    from cpt_amd import _lib as Lb
    H, step, site = 768, 5, 2 + 3 * 7
    g = torch.Generator().manual_seed(R + int(p * 10))
    o = torch.randn(R, H, generator=g).requires_grad_(True)
    resid = (torch.randn(R, H, generator=g) * 1.5 + 0.2).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(H, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(H, generator=g)).requires_grad_(True)
    drop = None
    if p > 0:
        _, sh = O.dropout_thresh_scale(p, False)
        drop = {"k": _mask(dev, p, step, site, False, R, H).float() * float(np.float32(sh))}
    pre = O._drop(o, drop, "k") + resid
    y = O.layer_norm(pre, gam, bet, 1e-12)
    dy = torch.randn(R, H, generator=g)
    y.backward(dy)
    lpt = torch.bfloat16 if lp == "bf16" else torch.float32
    dy_d, pre_d, gam_d = dy.to(dev), pre.detach().to(dev), gam.detach().to(dev)      # (held: a temporary's memory would be reused by the next copy)
    for scratch_on in (False, True):
        dx = torch.empty(R, H, device=dev)
        dxl = torch.empty(R, H, device=dev, dtype=lpt)
        dg, db, dbias = (torch.zeros(H, device=dev) for _ in range(3))
        scratch = torch.empty(((R + 3) // 4) * 3 * H, device=dev) if scratch_on else None
        dr = Lb.Dropout(p_hidden=p, p_attn=p, seed=SEED, step=step)
        Lb.check(Lb.lib().cpt_layernorm_bwd(dy_d.data_ptr(), pre_d.data_ptr(), gam_d.data_ptr(), 1e-12, dx.data_ptr(), dxl.data_ptr(),
                                            Lb.CPT_BF16 if lp == "bf16" else Lb.CPT_F32, dg.data_ptr(), db.data_ptr(), R, H, C.byref(dr) if p > 0 else None, site,
                                            dbias.data_ptr(), scratch.data_ptr() if scratch_on else None, scratch.numel() * 4 if scratch_on else 0, Lb.stream_ptr()),
                 "cpt_layernorm_bwd")
        assert _rel(dx.cpu(), resid.grad) < 2e-5
        assert _rel(dg.cpu(), gam.grad) < 1e-4 and _rel(db.cpu(), bet.grad) < 1e-4
        assert _rel(dxl.float().cpu(), o.grad) < (1e-2 if lp == "bf16" else 2e-5)
        assert _rel(dbias.cpu(), o.grad.sum(0)) < 1e-4


def test_embed_ln_bwd_against_oracle_autograd(dev):
    """BertEmbeddings backward: LayerNorm backward + scatter-add into the word / position / token-type tables, against autograd over
    oracle.text_embeddings; rows of padding_idx 0 receive no gradient (nn.Embedding(padding_idx=0) of the third-party BertEmbeddings:
    the oracle indexes the table directly, so its row 0 is cleared before the comparison); explicit position ids included."""
    from cpt_amd import _lib as Lb
    cfg = cfgmod.tiny()
    H, V, P, T = cfg.hidden_size, cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size
    B, Lt, Li = 5, 9, 4
    L = Lt + Li
    g = torch.Generator().manual_seed(7)
    sd = {"bert.embeddings.word_embeddings.weight": torch.randn(V, H, generator=g).requires_grad_(True),
          "bert.embeddings.position_embeddings.weight": torch.randn(P, H, generator=g).requires_grad_(True),
          "bert.embeddings.token_type_embeddings.weight": torch.randn(T, H, generator=g).requires_grad_(True),
          "bert.embeddings.LayerNorm.weight": (1 + 0.1 * torch.randn(H, generator=g)).requires_grad_(True),
          "bert.embeddings.LayerNorm.bias": (0.1 * torch.randn(H, generator=g)).requires_grad_(True)}
    ids = torch.randint(0, V, (B, Lt), generator=g)
    ids[:, -2:] = 0                                        # padding tokens
    ids[0, :3] = ids[1, :3]                                # repeated ids: the scatter must add
    tt = torch.randint(0, T, (B, Lt), generator=g)
    for pos in (None, torch.randint(0, P, (B, Lt), generator=g)):
        for t in sd.values():
            t.grad = None
        y = O.text_embeddings(sd, cfg.to_dict(), ids, tt, pos)
        dy_full = torch.randn(B, L, H, generator=g)
        y.backward(dy_full[:, :Lt])
        ref_w = sd["bert.embeddings.word_embeddings.weight"].grad.clone()
        ref_w[0] = 0
        d = {k: v.detach().to(dev) for k, v in sd.items()}
        dy_d, ids_d, tt_d, pos_d = dy_full.to(dev), ids.to(dev), tt.to(dev), (pos.to(dev) if pos is not None else None)
        dw, dp, dt = torch.zeros(V, H, device=dev), torch.zeros(P, H, device=dev), torch.zeros(T, H, device=dev)
        dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
        Lb.check(Lb.lib().cpt_embed_ln_bwd(dy_d.data_ptr(), ids_d.data_ptr(), tt_d.data_ptr(), pos_d.data_ptr() if pos_d is not None else None,
                                           d["bert.embeddings.word_embeddings.weight"].data_ptr(), d["bert.embeddings.position_embeddings.weight"].data_ptr(),
                                           d["bert.embeddings.token_type_embeddings.weight"].data_ptr(), d["bert.embeddings.LayerNorm.weight"].data_ptr(),
                                           cfg.layer_norm_eps, dw.data_ptr(), dp.data_ptr(), dt.data_ptr(), dg.data_ptr(), db.data_ptr(), B, Lt, L, H, V, P, T,
                                           Lb.stream_ptr()), "cpt_embed_ln_bwd")
        assert _rel(dw.cpu(), ref_w) < 2e-5 and float(dw[0].abs().max()) == 0.0
        assert _rel(dp.cpu(), sd["bert.embeddings.position_embeddings.weight"].grad) < 2e-5
        assert _rel(dt.cpu(), sd["bert.embeddings.token_type_embeddings.weight"].grad) < 2e-5
        assert _rel(dg.cpu(), sd["bert.embeddings.LayerNorm.weight"].grad) < 2e-5
        assert _rel(db.cpu(), sd["bert.embeddings.LayerNorm.bias"].grad) < 2e-5
