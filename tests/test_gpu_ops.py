"""GPU parity of each HIP operator (through the C ABI) against the CPU oracle / plain fp32 torch
on the same seeded inputs.  fp32 mode must agree to ~1e-5; bf16 mode to bf16-rounding level."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cpt_amd import _lib
    _lib.check(_lib.lib().cpt_check_device(0), "cpt_check_device")
    return torch.device("cuda:0")


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _t(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * scale).astype(np.float32))


def _stats(name, got, ref):
    d = (got.double() - ref.double()).abs()
    print("%s: max_abs=%.3e mean_abs=%.3e ref_absmax=%.3e" % (name, d.max().item(), d.mean().item(), ref.abs().max().item()))
    return d.max().item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 40), (7680 // 8, 2304, 768), (64, 30522, 768), (77, 3, 128),
                                   (300, 768, 2056)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gemm_bias(dev, M, N, K, mode):
    from cpt_amd import ops
    rng = _rng(M * 7 + N * 3 + K)
    a, w, b = _t(rng, M, K), _t(rng, N, K, scale=0.05), _t(rng, N)
    # asymmetric operands: a transposed/permuted fragment layout cannot pass
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    ad, wd = a.to(dev).to(dt), w.to(dev).to(dt)
    out = ops.gemm(ad, wd, b.to(dev), out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = ad.float().cpu().double() @ wd.float().cpu().double().T + b.double()
    err = _stats("gemm %s %dx%dx%d" % (mode, M, N, K), out.cpu(), ref)
    assert err < (2e-5 if mode == "fp32" else 2e-4) * max(1.0, math.sqrt(K / 64))   # inputs identical: only accumulation order differs


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gemm_epilogues(dev, mode):
    from cpt_amd import ops, _lib as L
    from oracle import cpt_oracle as O
    rng = _rng(5)
    M, N, K = 250, 384, 256
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    a, w, b, r = _t(rng, M, K), _t(rng, N, K, scale=0.05), _t(rng, N), _t(rng, M, N)
    ad, wd = a.to(dev).to(dt), w.to(dev).to(dt)
    base = ad.float().cpu() @ wd.float().cpu().T + b
    tol = 3e-5 if mode == "fp32" else 3e-4
    g = ops.gemm(ad, wd, b.to(dev), epi=L.EPI_GELU, out_dtype=torch.float32).cpu()
    assert _stats("gelu", g, O.gelu_erf(base)) < tol
    t = ops.gemm(ad, wd, b.to(dev), epi=L.EPI_TANH, out_dtype=torch.float32).cpu()
    assert _stats("tanh", t, torch.tanh(base)) < tol
    rs = ops.gemm(ad, wd, b.to(dev), epi=L.EPI_RESID, resid=r.to(dev), out_dtype=torch.float32).cpu()
    assert _stats("resid", rs, base + r) < tol
    if mode == "bf16":
        lo = ops.gemm(ad, wd, b.to(dev), epi=L.EPI_GELU, out_dtype=torch.bfloat16).cpu()
        assert _stats("gelu->bf16", lo.float(), O.gelu_erf(base)) < 2e-2


def test_gemm_rejects_bad_alignment(dev):
    from cpt_amd import ops
    a = torch.zeros(8, 30, device=dev)
    w = torch.zeros(8, 30, device=dev)
    with pytest.raises(RuntimeError):
        ops.gemm(a, w)


@pytest.mark.parametrize("H", [128, 768, 1024])
def test_layernorm_rows(dev, H):
    from cpt_amd import ops
    from oracle import cpt_oracle as O
    rng = _rng(H)
    R = 301
    x, g, b = _t(rng, R, H, scale=3.0) + 1.5, _t(rng, H) + 1.0, _t(rng, H)
    out, lp = ops.layernorm_rows(x.to(dev), g.to(dev), b.to(dev), 1e-12, lp_dtype=torch.bfloat16)
    ref = O.layer_norm(x, g, b, 1e-12)
    assert _stats("ln H=%d" % H, out.cpu(), ref) < 1e-5
    assert _stats("ln bf16 shadow", lp.float().cpu(), ref) < 4e-2
    # grouped placement: rows of group i land at i*stride + off (torch.cat replacement)
    grp, stride, off = 7, 12, 5
    dst = torch.zeros((R // grp) * stride + stride, H, device=dev)
    ops.layernorm_rows(x[: (R // grp) * grp].contiguous().to(dev), g.to(dev), b.to(dev), 1e-12, out=dst, grp=grp,
                       grp_stride=stride, grp_off=off)
    dst = dst.cpu()
    for i in (0, 3, R // grp - 1):
        assert torch.allclose(dst[i * stride + off: i * stride + off + grp], ref[i * grp:(i + 1) * grp], atol=1e-5)
    assert dst[0:off].abs().max() == 0


def test_embed_ln(dev):
    from cpt_amd import ops
    from oracle import cpt_oracle as O
    rng = _rng(11)
    V, P, H, B, Lt, L = 523, 40, 768, 5, 17, 23
    word, posw, typew = _t(rng, V, H), _t(rng, P, H), _t(rng, 2, H)
    g, b = _t(rng, H) + 1.0, _t(rng, H)
    ids = torch.from_numpy(rng.integers(0, V, (B, Lt)))
    tt = torch.from_numpy(rng.integers(0, 2, (B, Lt)))
    sd = {"bert.embeddings.word_embeddings.weight": word, "bert.embeddings.position_embeddings.weight": posw,
          "bert.embeddings.token_type_embeddings.weight": typew, "bert.embeddings.LayerNorm.weight": g,
          "bert.embeddings.LayerNorm.bias": b}
    ref = O.text_embeddings(sd, {"layer_norm_eps": 1e-12}, ids, tt)
    out, lp = ops.embed_ln(ids.to(dev), tt.to(dev), None, word.to(dev), posw.to(dev), typew.to(dev), g.to(dev),
                           b.to(dev), 1e-12, L, lp_dtype=torch.bfloat16)
    assert _stats("embed", out[:, :Lt].cpu(), ref) < 1e-5
    assert out[:, Lt:].abs().max().item() == 0
    assert _stats("embed bf16", lp[:, :Lt].float().cpu(), ref) < 4e-2


@pytest.mark.parametrize("L,B,heads", [(26, 3, 2), (120, 4, 12), (120, 2, 1), (210, 2, 12), (265, 2, 16)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_attention(dev, L, B, heads, mode):
    from cpt_amd import ops
    rng = _rng(L + heads)
    H = heads * 64
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    qkv = _t(rng, B * L, 3 * H).to(dt)
    mask = torch.ones(B, L, dtype=torch.int64)
    for b in range(B):
        mask[b, L - int(rng.integers(0, L // 2)):] = 0
    mask[0, 3] = 0
    ctx, probs = ops.attention(qkv.to(dev), mask.to(dev), B, L, heads, want_probs=True)
    x = qkv.float().view(B, L, 3, heads, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / 8.0 + ((1.0 - mask.float()) * -10000.0)[:, None, None, :]
    p = torch.softmax(s, -1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    tol = 2e-5 if mode == "fp32" else 3e-2
    assert _stats("attn probs %s" % mode, probs.float().cpu(), p) < (1e-5 if mode == "fp32" else 1e-2)
    assert _stats("attn ctx %s L=%d" % (mode, L), ctx.float().cpu(), ref) < tol


def test_pad_cast_gather_ce(dev):
    from cpt_amd import ops
    rng = _rng(3)
    x = _t(rng, 37, 2054)
    p = ops.pad_cast(x.to(dev), 2056, torch.bfloat16).cpu()
    assert torch.equal(p[:, :2054], x.to(torch.bfloat16)) and p[:, 2054:].abs().max() == 0
    p32 = ops.pad_cast(x.to(dev), 2056, torch.float32).cpu()
    assert torch.equal(p32[:, :2054], x) and p32[:, 2054:].abs().max() == 0
    B, L, H = 6, 11, 768
    src = _t(rng, B * L, H)
    pos = torch.from_numpy(rng.integers(0, L, (B,)))
    g = ops.gather_rows(src.to(dev), pos.to(dev), B, L).cpu()
    assert torch.equal(g, src.view(B, L, H)[torch.arange(B), pos])
    g0 = ops.gather_rows(src.to(dev).to(torch.bfloat16), None, B, L).cpu()
    assert torch.equal(g0, src.to(torch.bfloat16).view(B, L, H)[:, 0])
    R, V = 9, 30522
    lg = _t(rng, R, V, scale=2.0)
    lab = torch.from_numpy(rng.integers(0, V, (R,)))
    lab[2] = -1
    lab[7] = -1
    loss, d = ops.ce_rows(lg.to(dev), lab.to(dev), want_grad=True)
    loss = loss.cpu()
    ref = torch.nn.functional.cross_entropy(lg, lab, ignore_index=-1, reduction="sum")
    assert loss[1].item() == 7
    assert abs(loss[0].item() - ref.item()) < 1e-3
    lgr = lg.clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(lgr, lab, ignore_index=-1, reduction="sum").backward()
    assert _stats("ce grad", d.cpu(), lgr.grad) < 1e-6


@pytest.mark.parametrize("variant", [0, 3, 10, 11, 13, 14, 15, 18])
def test_gemm_variants_agree(dev, variant):
    """Every GEMM kernel variant (tile shape / pipeline depth) gives the same answer, including
    ragged M/N edges and the unaligned-ldo decoder shape."""
    from cpt_amd import ops, _lib as L
    rng = _rng(77)
    if True:
        for (M, N, K, epi) in [(7680 // 4, 2304, 768, L.EPI_NONE), (333, 200, 128, L.EPI_GELU), (64, 30522, 768, L.EPI_NONE),
                               (500, 768, 3072, L.EPI_RESID)]:
            a, w, b = _t(rng, M, K).to(torch.bfloat16), _t(rng, N, K, scale=0.05).to(torch.bfloat16), _t(rng, N)
            r = _t(rng, M, N) if epi == L.EPI_RESID else None
            out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), epi=epi, resid=None if r is None else r.to(dev),
                           out_dtype=torch.float32, tile=variant).cpu()
            ref = a.float() @ w.float().T + b
            if epi == L.EPI_GELU:
                ref = ref * 0.5 * (1.0 + torch.erf(ref / math.sqrt(2.0)))
            if r is not None:
                ref = ref + r
            assert _stats("variant %d %dx%dx%d" % (variant, M, N, K), out, ref) < 3e-4 * max(1.0, math.sqrt(K / 64))
            a32, w32 = a.float().to(dev), w.float().to(dev)
            o32 = ops.gemm(a32, w32, b.to(dev), epi=epi, resid=None if r is None else r.to(dev), tile=variant).cpu()
            assert _stats("variant %d fp32" % variant, o32, ref) < 3e-5 * max(1.0, math.sqrt(K / 64))


def test_select_regions_device_matches_reference_rule(dev):
    """Row a15 / 8(f).3 on the device: same index as the host rule of zeroshot/refcoco_cpt.py:224-246 (raw logits) and
    fewshot/refcoco_cpt.py:277-295 (ratio to the "none" logit), including torch.argmax's first-max tie-break,
    ragged colour sets, a query whose maximum sits in a later sequence, and NaN."""
    from cpt_amd import scoring
    from oracle import cpt_oracle as O
    g = torch.Generator().manual_seed(3)
    V, none_id = 3000, 2999
    n_seq = [3, 1, 5, 2, 4]
    sets, first = [], [0]
    for n in n_seq:
        for _ in range(n):
            k = int(torch.randint(1, 6, (1,), generator=g))
            sets.append([int(v) for v in torch.randperm(V - 1, generator=g)[:k]])
        first.append(first[-1] + n)
    S = first[-1]
    scores = torch.randn(S, V, generator=g)
    scores[:, none_id] = torch.rand(S, generator=g) + 0.5
    # ties: two equal maxima in query 2 (first must win); NaN in query 3
    scores[first[2] + 1, sets[first[2] + 1][0]] = 50.0
    scores[first[2] + 3, sets[first[2] + 3][0]] = 50.0
    scores[first[3] + 1, sets[first[3] + 1][0]] = float("nan")
    sd = scores.to(dev)
    for few in (False, True):
        got, val = scoring.select_regions_device(sd, sets, first, none_id, few_shot=few, return_scores=True)
        for q in range(len(n_seq)):
            rows = scores[first[q]:first[q + 1]]
            cs = sets[first[q]:first[q + 1]]
            ref_idx, ref_sc = (O.select_region_fewshot if few else O.select_region_zeroshot)(rows, cs, none_id)
            assert int(got[q]) == ref_idx, (few, q, int(got[q]), ref_idx)
            if ref_sc[ref_idx] == ref_sc[ref_idx]:
                assert float(val[q]) == float(ref_sc[ref_idx])          # bit-identical fp32 (IEEE division)
    ids = [5, 17, 2998, 40, 41]
    sc2 = torch.randn(7, V, generator=g)
    sc2[3, 40] = sc2[3, 41] = 9.0
    got = scoring.argmax_columns_device(sc2.to(dev), ids)
    assert got.cpu().tolist() == sc2[:, ids].argmax(1).tolist()


@pytest.mark.parametrize("M,N,K,variant", [(7680, 3072, 768, 3), (1000, 3072, 768, 20), (500, 512, 1024, 20), (100, 3072, 768, 3),
                                           (1000, 3072, 768, 15), (1000, 2304, 768, 14), (640, 1024, 1024, 19), (300, 200, 128, 3)])
def test_gemm_ln_consumer(dev, M, N, K, variant):
    """The LayerNorm-consumer GEMM of the fused bf16 encoder, gelu( LayerNorm(x) . W^T + bias ) with the LayerNorm folded
    (modeling_bert.py:144 + the LayerNorm of :86 before it), on every kernel that serves it: the two-pass 384 x 256 FFN-up
    kernel (gemm_ffn.hip; K = 768 and 1024, full and ragged row tiles), the direct-epilogue 128 x 192 / 384 x 192 / 384 x 256
    tile shapes of gemm.hip, and shapes that fall back to guarded edge tiles.  Reference: fp64 on the bf16-rounded operands."""
    from cpt_amd import ops, _lib as L
    rng = _rng(5 + M + K)
    x = _t(rng, M, K, scale=1.3) + 0.4
    W = _t(rng, N, K, scale=0.04)
    gamma, beta, bias = 1.0 + _t(rng, K, scale=0.1), _t(rng, K, scale=0.1), _t(rng, N, scale=0.1)
    eps = 1e-12
    a = x.to(torch.bfloat16)
    wf = (W * gamma).to(torch.bfloat16)
    colc = wf.float().sum(1)                              # as the MFMA sees the folded weight
    cold = (W.double() @ beta.double() + bias.double()).float()
    st = ops.row_stats_table(x.to(dev))
    mu = x.double().mean(1, keepdim=True)
    rs = 1.0 / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + eps)
    for gelu in (True, False):
        pre = rs * (a.double() @ wf.double().T - mu * colc.double()) + cold.double()
        ref = pre * 0.5 * (1.0 + torch.erf(pre / math.sqrt(2.0))) if gelu else pre
        got = ops.gemm_ln_cons(a.to(dev), wf.to(dev), st, colc.to(dev), cold.to(dev), eps, K, gelu, tile=variant).float().cpu()
        again = ops.gemm_ln_cons(a.to(dev), wf.to(dev), st, colc.to(dev), cold.to(dev), eps, K, gelu, tile=variant).float().cpu()
        assert torch.equal(got, again)                    # no run-to-run variation (counted-vmcnt pipeline)
        d = (got.double() - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 2e-3                # bf16 output rounding (half an ulp = 2^-9 relative) + fp32 accumulation
        print("ln-consumer %dx%dx%d variant %d gelu=%d: max|d| %.3e (ref absmax %.2f)" % (M, N, K, variant, gelu, d.max().item(), ref.abs().max().item()))
        assert bool((d <= tol).all()), (d - tol).max().item()


def test_ln_consumer_kernels_are_bit_identical(dev):
    """Which kernel serves the FFN-up GEMM depends on the batch size (two-pass 384 x 256 tiles when they fill the chip,
    128 x 192 two-per-CU tiles otherwise, 384 x 256 single pass as an option): every one of them must produce the same BITS,
    or rows of a batch would depend on the batch they sit in (the accumulation order over K and the epilogue arithmetic
    -- explicit fma -- are the same by construction)."""
    from cpt_amd import ops, _lib as L
    rng = _rng(99)
    M, N, K = 7680, 3072, 768
    x = _t(rng, M, K, scale=1.3) + 0.4
    a = x.to(torch.bfloat16).to(dev)
    st = ops.row_stats_table(x.to(dev))
    wf = _t(rng, N, K, scale=0.04).to(torch.bfloat16).to(dev)
    colc = wf.float().sum(1).contiguous()
    cold = _t(rng, N, scale=0.1).to(dev)
    outs = {}
    for v in (3, 15, 19, 20, 21):         # (3: the library's own choice here; 20 / 21: the two-pass / 4-wave kernels)
        outs[v] = ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, K, True, tile=v)
        # the first 960 rows as their own (small) problem: other tile shape choices, ragged last tiles
        outs[(v, "small")] = ops.gemm_ln_cons(a[:960], wf, st[:960], colc, cold, 1e-12, K, True, tile=v)
    for v in (15, 19, 20, 21):
        assert torch.equal(outs[3], outs[v]), "variant %d differs from the default" % v
    for v in (3, 15, 19, 20, 21):
        assert torch.equal(outs[3][:960], outs[(v, "small")]), "variant %d, 960-row problem" % v


@pytest.mark.parametrize("M,N,K", [(53760, 2304, 768), (8480, 3072, 1024), (7680, 2304, 768)])
def test_qkv_projection_two_pass_kernel_is_bit_identical(dev, M, N, K):
    """Round 3: the stand-alone QKV projection with the LayerNorm folded (sequences longer than 128, whose attention does not fuse: the
    GQA and VCR shapes, Oscar-base and Oscar-large widths) runs the GELU-less form of the two-pass 384 x 256 kernel when its tiles fill
    the chip; cpt_set_tuning(20, 0) keeps the 384 x 192 pipelined kernel.  Same accumulation order and epilogue arithmetic: same BITS,
    also for a 960-row slice (which takes the small-problem tile shapes)."""
    from cpt_amd import ops, _lib as L
    rng = _rng(M + N)
    x = _t(rng, M, K, scale=1.3) + 0.4
    a = x.to(torch.bfloat16).to(dev)
    st = ops.row_stats_table(x.to(dev))
    wf = _t(rng, N, K, scale=0.04).to(torch.bfloat16).to(dev)
    colc = wf.float().sum(1).contiguous()
    cold = _t(rng, N, scale=0.1).to(dev)
    two = ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, K, False)
    small = ops.gemm_ln_cons(a[:960], wf, st[:960], colc, cold, 1e-12, K, False)
    for forced in (20, 21, 14):       # the two-pass and the 4-wave consumer kernels and the 384 x 192 pipelined kernel, whatever the library chose
        assert torch.equal(ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, K, False, tile=forced), two), forced
    assert torch.equal(two[:960], small)
    xs = x[:256]
    ref = ((xs.to(torch.bfloat16).float() - xs.mean(1, keepdim=True)) / xs.var(1, unbiased=False, keepdim=True).add(1e-12).sqrt()) @ wf.float().cpu().T
    # (loose sanity bound against fp32 math of the folded form; the operator's parity test is test_gemm_ln_cons)
    got = two[:256].float().cpu() - cold.cpu()
    assert float((got - ref).abs().max()) < 0.05 * float(ref.abs().max()) + 0.05


@pytest.mark.parametrize("K,M,N", [(3840, 768, 768), (3840, 2304, 768), (3840, 768, 3072), (1600, 768, 2112), (960, 1024, 1024),
                                   (192, 128, 192), (64, 128, 128), (53760, 768, 768), (2120, 1024, 4096), (100, 128, 128)])
def test_gemm_tn_weight_gradient_form(dev, K, M, N):
    """out = A^T W with both bf16 operands stored rows = contraction index (dY [tokens][out], X [tokens][in]): the TN GEMM reads
    them through LDS transpose reads.  Against torch fp32 matmul of the same bf16 values; with and without split-K scratch the
    result must agree to fp32 summation-order noise, and two runs are bit-equal (partials are added in split order)."""
    from cpt_amd import ops
    rng = _rng(K + M + N)
    # the operands are the first K rows of buffers whose tails hold NaN: token counts that are not a multiple of the 64-row K-tile
    # (2120 = 8 x 265, 100) must read the missing rows as zero through the buffer bounds, not whatever follows in memory
    abuf = torch.full((K + 64, M), float("nan"), dtype=torch.bfloat16, device=dev)
    wbuf = torch.full((K + 64, N), float("nan"), dtype=torch.bfloat16, device=dev)
    abuf[:K] = _t(rng, K, M).to(torch.bfloat16).to(dev)
    wbuf[:K] = _t(rng, K, N).to(torch.bfloat16).to(dev)
    a, w = abuf[:K], wbuf[:K]
    a[:, 5] = 0.0
    a[7, 5] = 1.0                                  # output row 5 = row 7 of w exactly: catches any row / column permutation
    ref = a.float().t() @ w.float()
    got = ops.gemm_tn(a, w)
    tol = 2e-5 * (K ** 0.5) + 1e-4
    assert (got - ref).abs().max().item() < tol * 4, (got - ref).abs().max().item()
    assert torch.equal(got[5], w[7].float())
    assert torch.equal(got, ops.gemm_tn(a, w))
    one = ops.gemm_tn(a, w, split_scratch=False)   # no scratch: one split
    assert (one - ref).abs().max().item() < tol * 4
    # strided operands (a column block of a wider tensor, as dqkv / the FFN activations are)
    wide = _t(rng, K, N + 64).to(torch.bfloat16).to(dev)
    got2 = ops.gemm_tn(a, wide[:, 64:])
    assert (got2 - a.float().t() @ wide[:, 64:].float()).abs().max().item() < tol * 4


@pytest.mark.parametrize("M,N,K,resid,odt", [(3840, 768, 3072, True, torch.float32), (3840, 3072, 768, False, torch.bfloat16),
                                             (3840, 768, 768, False, torch.bfloat16), (3840, 768, 2304, True, torch.float32),
                                             (1000, 768, 768, False, torch.float32), (77, 192, 64, True, torch.float32),
                                             (2120, 1024, 4096, True, torch.float32), (2120, 4096, 1024, False, torch.bfloat16),
                                             (53760, 768, 768, False, torch.bfloat16)])
def test_gemm_nn_data_gradient_form(dev, M, N, K, resid, odt):
    """out = A W (+ resid) with W stored [K][N] (an nn.Linear weight, rows = the contraction index): A staged as in the NT form, W read
    through LDS transpose reads; 64-row and 128-row tile instantiations, ragged M, every epilogue the backward pass uses."""
    from cpt_amd import ops
    rng = _rng(M + N + K)
    a = _t(rng, M, K).to(torch.bfloat16).to(dev)
    w = _t(rng, K, N, scale=0.05).to(torch.bfloat16).to(dev)
    w[:, 3] = 0.0
    w[11, 3] = 1.0                                  # output column 3 = column 11 of a exactly
    r = _t(rng, M, N).to(dev) if resid else None
    ref = a.float() @ w.float() + (r if resid else 0.0)
    got = ops.gemm_nn(a, w, r, odt)
    tol = (2e-5 * (K ** 0.5) + 1e-4) * 4 if odt == torch.float32 else 0.02 * (K ** 0.5) * 0.05 + 0.02
    assert (got.float() - ref).abs().max().item() < tol, (got.float() - ref).abs().max().item()
    if not resid:
        assert torch.equal(got[:, 3].float(), a[:, 11].float())
    assert torch.equal(got, ops.gemm_nn(a, w, r, odt))


def test_gemm_nn_split_k_and_row_bound(dev):
    """The decoder's data gradient: 32 x 768 outputs over the vocabulary (K = 30522 rounded up to 30528 with zero columns in A; the
    weight has 30522 rows and is followed by other live memory): split-K partials added in order, rows beyond the bound read as
    zero -- also when what follows the weight in memory is NaN."""
    from cpt_amd import ops
    rng = _rng(30522)
    V, Vp, H, B = 30522, 30528, 768, 32
    a = torch.zeros(B, Vp, dtype=torch.bfloat16, device=dev)
    a[:, :V] = (_t(rng, B, V, scale=0.01)).to(torch.bfloat16).to(dev)
    buf = torch.full((Vp + 8, H), float("nan"), dtype=torch.bfloat16, device=dev)       # the weight's rows, then NaN
    buf[:V] = _t(rng, V, H, scale=0.05).to(torch.bfloat16).to(dev)
    w = buf[:V]
    ref = a[:, :V].float() @ w.float()
    for scratch in (True, False):
        got = ops.gemm_nn(a, w, None, torch.float32, split_scratch=scratch)
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < 2e-3, (scratch, (got - ref).abs().max().item())
    assert torch.equal(ops.gemm_nn(a, w, None, torch.float32, split_scratch=True), ops.gemm_nn(a, w, None, torch.float32, split_scratch=True))


def _r3_encode_np(x):
    """CPU restatement of the 3-byte residual code (csrc/common.h r3_encode): T = fp32 pattern rounded half away to 24 bits,
    hi = (T + 0x80) >> 8 (a bf16 pattern), lo = int8(T - (hi << 8))."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    hi = (((b + 0x8080) >> 16) & 0xffff).astype(np.uint16)
    lo = (((b + 0x80) >> 8) & 0xff).astype(np.uint8).view(np.int8)
    return hi, lo


def _r3_decode_np(hi, lo):
    v = ((hi.astype(np.int64) << 16) + (lo.astype(np.int64) << 8)) & 0xffffffff
    return v.astype(np.uint32).view(np.float32)


def test_resid3_codec_matches_restatement(dev):
    """3-byte residual stream: the device split / merge kernels equal the numpy restatement bit for bit (incl. zeros, denormals,
    infinities, values that round up into the next exponent, lo = -128 / 127), hi is a legal bf16 within one bf16 ulp of RNE,
    and decode(encode(x)) is within 2^-17 relative."""
    from cpt_amd import ops
    rng = _rng(31)
    x = (rng.standard_normal(1 << 16) * np.exp(rng.uniform(-20, 20, 1 << 16))).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-40, -1e-40, 3.4028235e38, 1.0, -1.0], np.float32)
    pats = np.array([0x3f807f80, 0x3f808080, 0x3f80807f, 0x3f7fff80, 0x3f7fffff, 0x7f7fff7f, 0x00000080, 0x3f800080, 0xbf80ff80], np.uint32).view(np.float32)
    x[:special.size] = special
    x[special.size:special.size + pats.size] = pats
    hi_ref, lo_ref = _r3_encode_np(x)
    xt = torch.from_numpy(x).to(dev).view(256, 256)
    hi, lo = ops.resid3_split(xt)
    assert np.array_equal(hi.view(torch.int16).cpu().numpy().view(np.uint16).ravel(), hi_ref)
    assert np.array_equal(lo.cpu().numpy().ravel(), lo_ref)
    back = ops.resid3_merge(hi, lo).cpu().numpy().ravel()
    ref_back = _r3_decode_np(hi_ref, lo_ref)
    assert np.array_equal(back.view(np.uint32), ref_back.view(np.uint32))
    fin = np.isfinite(x) & (np.abs(x) > 1e-37) & (np.abs(x) < 1e38)
    assert np.max(np.abs(back[fin] - x[fin]) / np.abs(x[fin])) <= 2.0 ** -16
    assert np.array_equal(back[~np.isfinite(x)], x[~np.isfinite(x)])
    rne = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16).astype(np.int64)
    ok = np.isfinite(x)
    assert np.max(np.abs(hi_ref.astype(np.int64)[ok] - rne[ok])) <= 1
    # gathered rows: row pos[r] of every group of L rows
    pos = torch.tensor([3, 0, 7, 15] * 4, dtype=torch.int64, device=dev)
    g = ops.resid3_merge(hi, lo, pos, 16).cpu().numpy()
    want = back.reshape(16, 16, 256)[np.arange(16), pos.cpu().numpy()]
    assert np.array_equal(g, want)


@pytest.mark.parametrize("K", [768, 3072])
def test_gemm_ln_prod3_matches_fp32_producer(dev, K):
    """The 3-byte producer against the fp32 producer on the same operands: with the residual given as (hi, lo) and as the fp32
    values those decode to, the outputs agree to the code's resolution (decode(out3) vs out_f32 within 2^-16 relative + the
    residual's own 2^-17), out_hi is the code's hi of out_f32 up to that resolution, the row sums are equal to 1e-6 relative;
    rows are batch invariant bit for bit (ragged M included)."""
    from cpt_amd import ops
    rng = _rng(K)
    M, H = 1000, 768
    x = (_t(rng, M, H, scale=1.2) + 0.3).to(dev)
    hi, lo = ops.resid3_split(x)
    xq = ops.resid3_merge(hi, lo)                      # what the (hi, lo) pair holds exactly
    st = ops.row_stats_table(x)
    a = _t(rng, M, K).to(torch.bfloat16).to(dev)
    w = _t(rng, H, K, scale=0.03).to(torch.bfloat16).to(dev)
    bias, g, bt = _t(rng, H, scale=0.1).to(dev), (1 + _t(rng, H, scale=0.1)).to(dev), _t(rng, H, scale=0.1).to(dev)
    for fold in (True, False):
        gi, bi, si = (g, bt, st) if fold else (None, None, None)
        o_f32, o_bf16, st_f32 = ops.gemm_ln_prod(a, w, bias, xq, si, gi, bi, 1e-12, H)
        o_hi, o_lo, st3 = ops.gemm_ln_prod3(a, w, bias, hi, lo, si, gi, bi, 1e-12, H)
        dec = ops.resid3_merge(o_hi, o_lo)
        err = ((dec - o_f32).abs() / o_f32.abs().clamp_min(1e-3)).max().item()
        assert err <= 2.0 ** -16, err
        assert torch.equal(st3, st_f32)                # row sums are taken from the unrounded values in both forms
        dh = (o_hi.view(torch.int16).int() - o_bf16.view(torch.int16).int()).abs().max().item()
        assert dh <= 1, dh
        for Ms in (77, 120):
            s_hi, s_lo, s_st = ops.gemm_ln_prod3(a[:Ms].contiguous(), w, bias, hi[:Ms].contiguous(), lo[:Ms].contiguous(),
                                                 si[:Ms].contiguous() if fold else None, gi, bi, 1e-12, H)
            assert torch.equal(s_hi, o_hi[:Ms]) and torch.equal(s_lo, o_lo[:Ms]) and torch.equal(s_st, st3[:Ms])


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("K,H,M", [(768, 768, 640), (3072, 768, 640), (512, 384, 128), (1024, 192, 256)])
def test_gemm_ln_prod3_panel_is_bit_identical(dev, K, H, M, waves):
    """Round 3: the producer that reads A from its fragment-major panel copy straight into registers (gemm_prod.hip) against the
    row-major producer on the same operands: hi, lo and the partial row sums bit for bit, with and without the on-the-fly residual
    LayerNorm; the panel copy round-trips through cpt_panel_pack; 20 back-to-back launches give identical bits.  Both wave shapes of the
    tile (round 4, cpt_set_tuning key 24: 4 x 2 waves of 32 x 96, 4 x 1 waves of 32 x 192) give the same bits."""
    from cpt_amd import ops, _lib as L
    rng = _rng(K + 5)          # (K = 512: the shortest K loop the kernel runs, 8 K-tiles; one row tile; 1, 2 and 4 column tiles)
    x = (_t(rng, M, H, scale=1.2) + 0.3).to(dev)
    hi, lo = ops.resid3_split(x)
    st = ops.row_stats_table(x)
    a = _t(rng, M, K).to(torch.bfloat16).to(dev)
    w = _t(rng, H, K, scale=0.03).to(torch.bfloat16).to(dev)
    bias, g, bt = _t(rng, H, scale=0.1).to(dev), (1 + _t(rng, H, scale=0.1)).to(dev), _t(rng, H, scale=0.1).to(dev)
    ap = ops.panel_pack(a)
    assert torch.equal(ops.panel_pack(ap, to_panel=False, K=K), a)
    # the documented layout: element (row, k) at (((row / 32) (K / 16) + k / 16) 64 + ((k % 16) / 8) 32 + row % 32) 8 + k % 8
    rows = torch.arange(M, device=dev)[:, None]
    ks = torch.arange(K, device=dev)[None, :]
    idx = (((rows // 32) * (K // 16) + ks // 16) * 64 + ((ks % 16) // 8) * 32 + rows % 32) * 8 + ks % 8
    assert torch.equal(ap[idx.reshape(-1)].view(M, K), a)
    for fold in (True, False):
        gi, bi, si = (g, bt, st) if fold else (None, None, None)
        r_hi, r_lo, r_st = ops.gemm_ln_prod3(a, w, bias, hi, lo, si, gi, bi, 1e-12, H)
        p_hi, p_lo, p_st = ops.gemm_ln_prod3_panel(ap, K, w, bias, hi, lo, si, gi, bi, 1e-12, H, waves=waves)
        assert torch.equal(p_hi, r_hi) and torch.equal(p_lo, r_lo) and torch.equal(p_st, r_st)
        for _ in range(20):
            q_hi, q_lo, q_st = ops.gemm_ln_prod3_panel(ap, K, w, bias, hi, lo, si, gi, bi, 1e-12, H, waves=waves)
            assert torch.equal(q_hi, r_hi) and torch.equal(q_lo, r_lo) and torch.equal(q_st, r_st)


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("K,H,M", [(768, 768, 640), (3072, 768, 640), (512, 384, 128), (1024, 192, 256)])
def test_gemm_ln_prod3_rpanel_is_bit_identical(dev, K, H, M, waves):
    """Round 5: the producer with the RESIDUAL STREAM in the panel layout as well and a register-direct epilogue (swapped MFMA operands, no
    LDS slab) against the row-major producer: hi, lo (unpacked from their panels) and the partial row sums bit for bit, with and without
    the on-the-fly residual LayerNorm, both wave shapes; the byte panel round-trips; 20 back-to-back launches give identical bits."""
    from cpt_amd import ops, _lib as L
    rng = _rng(K + 11)
    x = (_t(rng, M, H, scale=1.2) + 0.3).to(dev)
    hi, lo = ops.resid3_split(x)
    hi, lo = hi.view(M, H), lo.view(M, H)
    st = ops.row_stats_table(x)
    a = _t(rng, M, K).to(torch.bfloat16).to(dev)
    w = _t(rng, H, K, scale=0.03).to(torch.bfloat16).to(dev)
    bias, g, bt = _t(rng, H, scale=0.1).to(dev), (1 + _t(rng, H, scale=0.1)).to(dev), _t(rng, H, scale=0.1).to(dev)
    ap, hp, lp = ops.panel_pack(a), ops.panel_pack(hi), ops.panel_pack_bytes(lo)
    assert torch.equal(ops.panel_pack_bytes(lp, to_panel=False, K=H), lo)
    rows = torch.arange(M, device=dev)[:, None]
    ks = torch.arange(H, device=dev)[None, :]
    idx = (((rows // 32) * (H // 16) + ks // 16) * 64 + ((ks % 16) // 8) * 32 + rows % 32) * 8 + ks % 8
    assert torch.equal(lp[idx.reshape(-1)].view(M, H), lo)
    for fold in (True, False):
        gi, bi, si = (g, bt, st) if fold else (None, None, None)
        r_hi, r_lo, r_st = ops.gemm_ln_prod3(a, w, bias, hi, lo, si, gi, bi, 1e-12, H)
        for rep in range(21):
            q_hi, q_lo, q_st = ops.gemm_ln_prod3_rpanel(ap, K, w, bias, hp, lp, si, gi, bi, 1e-12, H, waves=waves)
            assert torch.equal(q_st, r_st), "row sums (fold %s, launch %d)" % (fold, rep)
            assert torch.equal(ops.panel_pack(q_hi, to_panel=False, K=H), r_hi) and torch.equal(ops.panel_pack_bytes(q_lo, to_panel=False, K=H), r_lo)


@pytest.mark.parametrize("Mbig,Msmall", [(7680, 840), (7680, 120), (1000, 77)])
def test_operators_are_batch_invariant(dev, Mbig, Msmall):
    """Rows [0, Msmall) of a big problem equal the same rows run as their own problem, bit for bit, for the four GEMM forms of
    the fused encoder -- also when Msmall is not a multiple of the 32-row wave sub-tile, where the guarded per-element
    epilogue finishes the last rows (round 1 only ever compared multiples of 32 and missed a contraction difference there)."""
    from cpt_amd import ops
    rng = _rng(Mbig + Msmall)
    H, I = 768, 3072
    x = (_t(rng, Mbig, H, scale=1.2) + 0.3).to(dev)
    a = x.to(torch.bfloat16)
    st = ops.row_stats_table(x)
    for N, gelu in ((3 * H, False), (I, True)):
        wf = _t(rng, N, H, scale=0.03).to(torch.bfloat16).to(dev)
        colc = wf.float().sum(1).contiguous()
        cold = _t(rng, N, scale=0.1).to(dev)
        big = ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, H, gelu)
        small = ops.gemm_ln_cons(a[:Msmall].contiguous(), wf, st[:Msmall].contiguous(), colc, cold, 1e-12, H, gelu)
        assert torch.equal(big[:Msmall], small), "consumer N=%d" % N
    for K in (H, I):
        ak = _t(rng, Mbig, K).to(torch.bfloat16).to(dev)
        w = _t(rng, H, K, scale=0.03).to(torch.bfloat16).to(dev)
        bias, g, bt = _t(rng, H, scale=0.1).to(dev), (1 + _t(rng, H, scale=0.1)).to(dev), _t(rng, H, scale=0.1).to(dev)
        o1 = ops.gemm_ln_prod(ak, w, bias, x, st, g, bt, 1e-12, H)
        o2 = ops.gemm_ln_prod(ak[:Msmall].contiguous(), w, bias, x[:Msmall].contiguous(), st[:Msmall].contiguous(), g, bt, 1e-12, H)
        for nm, p, q in zip(("fp32", "bf16", "row sums"), o1, o2):
            assert torch.equal(p[:Msmall], q), "producer K=%d: %s" % (K, nm)


def test_retile_k32_matches_reshape(dev):
    """cpt_retile_k32: dst[K / 32][N][32] = src[N][K] (the K-tile-major weight copy the (sequence, three heads) QKV + attention launch reads)."""
    from cpt_amd import _lib as L
    torch.manual_seed(3)
    for N, K in ((2304, 768), (96, 64), (7, 32)):
        src = torch.randn(N, K, device=dev).to(torch.bfloat16)
        dst = torch.empty(K // 32, N, 32, device=dev, dtype=torch.bfloat16)
        L.check(L.lib().cpt_retile_k32(src.data_ptr(), dst.data_ptr(), N, K, L.stream_ptr()), "cpt_retile_k32")
        want = src.view(N, K // 32, 32).permute(1, 0, 2).contiguous()
        assert torch.equal(dst.view(torch.int16), want.view(torch.int16))
    with pytest.raises(RuntimeError):
        L.check(L.lib().cpt_retile_k32(src.data_ptr(), dst.data_ptr(), 7, 40, L.stream_ptr()), "cpt_retile_k32")
