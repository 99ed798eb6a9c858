"""GPU race screen of the pipelined GEMM kernels: the LDS-DMA operand rings are synchronised by
counted s_waitcnt vmcnt + one raw s_barrier per K-tile, i.e. by ORDERING; a mistake there shows up as rare wrong tiles
that depend on memory timing.  Every tile configuration the hot path uses is launched hundreds of times back to back
while a second stream thrashes HBM / L2 / MALL, and every result must equal the first launch bit for bit (no kernel on
the inference path uses atomics, so bitwise equality is the specification)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
LAUNCHES = 500


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _screen(dev, launch, n=LAUNCHES, ring=4):
    """launch(slot) -> tuple of output tensors written into per-slot buffers.  Runs n launches under memory pressure and
    compares every result with the first one (slots rotate so a result is checked before its buffer is reused)."""
    side = torch.cuda.Stream(device=dev)
    junk_a = torch.empty(192 * 1024 * 1024 // 4, device=dev)
    junk_b = torch.empty_like(junk_a)
    ref = [t.clone() for t in launch(0)]
    torch.cuda.synchronize()
    bad = torch.zeros((), device=dev, dtype=torch.int64)
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(n // 4 + 1):                # ~0.1 ms per copy pair: runs alongside the whole screen
            junk_b.copy_(junk_a)
            junk_a.copy_(junk_b)
        stop.record(side)
    for i in range(n):
        outs = launch(i % ring)
        for o, r in zip(outs, ref):
            bad += (o.view(torch.int16 if o.dtype == torch.bfloat16 else torch.int32) != r.view(torch.int16 if r.dtype == torch.bfloat16 else torch.int32)).any().long()
    torch.cuda.synchronize()
    return int(bad.item())


@pytest.mark.parametrize("variant,M,N,K,epi", [(3, 7680, 768, 768, "resid"), (3, 7680, 768, 3072, "resid"), (14, 7680, 2304, 768, "none"),
                                               (15, 7680, 3072, 768, "gelu"), (10, 1920, 3072, 768, "gelu"), (11, 1920, 768, 768, "none"),
                                               (18, 640, 768, 2112, "none")])
def test_gemm_ring_race_screen(dev, variant, M, N, K, epi):
    from cpt_amd import ops, _lib as L
    torch.manual_seed(variant + K)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if epi == "resid" else None
    code = {"resid": L.EPI_RESID, "gelu": L.EPI_GELU, "none": L.EPI_NONE}[epi]
    odt = torch.float32 if epi == "resid" else torch.bfloat16
    bad = _screen(dev, lambda s: (ops.gemm(a, w, b, epi=code, resid=r, out_dtype=odt, tile=variant),), n=LAUNCHES if K < 3000 else 300)
    assert bad == 0, "%d of the launches differed from the first one" % bad


@pytest.mark.parametrize("form,M,N,K", [("tn", 768, 768, 3840), ("tn", 3072, 768, 3840), ("tn", 1024, 1024, 960), ("nn", 3840, 768, 3072),
                                        ("nn", 3840, 3072, 768), ("nn", 1000, 768, 768)])
def test_gradient_gemm_forms_race_screen(dev, form, M, N, K):
    """TN (weight gradients, split-K partials added in order) and NN (data gradients) forms: operand tiles read by inline-asm
    transpose reads behind explicit counted lgkmcnt waits -- exactly the kind of synchronisation a timing change would break."""
    from cpt_amd import ops
    torch.manual_seed(M + K)
    if form == "tn":
        a = torch.randn(K, M, device=dev).to(torch.bfloat16)
        w = torch.randn(K, N, device=dev).to(torch.bfloat16)
        bad = _screen(dev, lambda s: (ops.gemm_tn(a, w),), n=300)
    else:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev)
        bad = _screen(dev, lambda s: (ops.gemm_nn(a, w, r, torch.float32),), n=300)
    assert bad == 0, "%d of the launches differed from the first one" % bad


@pytest.mark.parametrize("variant,M,N,K", [(3, 7680, 3072, 768), (20, 2000, 4096, 1024), (15, 7680, 3072, 768), (19, 7680, 3072, 768), (14, 7680, 2304, 768)])
def test_ln_consumer_race_screen(dev, variant, M, N, K):
    """Two-pass FFN-up kernel (variant 3, K = 768 and 1024 incl. a ragged last row tile) and the direct-epilogue tile shapes."""
    from cpt_amd import ops, _lib as L
    torch.manual_seed(variant + K)
    x = torch.randn(M, K, device=dev) + 0.2
    a = x.to(torch.bfloat16)
    st = ops.row_stats_table(x)
    wf = (torch.randn(N, K, device=dev) * 0.04).to(torch.bfloat16)
    colc = wf.float().sum(1).contiguous()
    cold = torch.randn(N, device=dev) * 0.1
    bad = _screen(dev, lambda s: (ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, K, True, tile=variant),))
    assert bad == 0, "%d of the launches differed from the first one" % bad


@pytest.mark.parametrize("K", [768, 3072])
def test_ln_producer_race_screen(dev, K):
    """attn-out / FFN-down form: + bias + LayerNorm-on-the-fly residual, fp32 + bf16 outputs + partial row sums."""
    from cpt_amd import ops
    M, N = 7680, 768
    torch.manual_seed(K)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) * 0.1
    resid = torch.randn(M, N, device=dev)
    st_in = ops.row_stats_table(resid)
    g, bt = 1.0 + 0.1 * torch.randn(N, device=dev), 0.1 * torch.randn(N, device=dev)
    bad = _screen(dev, lambda s: ops.gemm_ln_prod(a, w, bias, resid, st_in, g, bt, 1e-12, N), n=LAUNCHES if K < 3000 else 300)
    assert bad == 0, "%d of the launches differed from the first one" % bad
    # the same producer with the 3-byte residual stream (what the fused encoder runs)
    hi, lo = ops.resid3_split(resid)
    bad = _screen(dev, lambda s: ops.gemm_ln_prod3(a, w, bias, hi, lo, st_in, g, bt, 1e-12, N), n=300)
    assert bad == 0, "3-byte producer: %d of the launches differed from the first one" % bad
    # round 3: the panel producer (A straight into registers by asm loads, W through the LDS ring, ONE hand-counted vmcnt for both
    # queues, side data and the first residual slice prefetched by asm loads): the kind of synchronisation a timing change would break
    ap = ops.panel_pack(a)
    ref = ops.gemm_ln_prod3(a, w, bias, hi, lo, st_in, g, bt, 1e-12, N)
    got = ops.gemm_ln_prod3_panel(ap, K, w, bias, hi, lo, st_in, g, bt, 1e-12, N)
    assert all(torch.equal(u, v) for u, v in zip(ref, got))
    bad = _screen(dev, lambda s: ops.gemm_ln_prod3_panel(ap, K, w, bias, hi, lo, st_in, g, bt, 1e-12, N), n=LAUNCHES if K < 3000 else 300)
    assert bad == 0, "panel producer: %d of the launches differed from the first one" % bad


def test_fused_qkv_attention_race_screen(dev):
    """The whole bf16 forward (fused QKV + attention kernel, producers, consumers, head) 200 times under memory pressure."""
    from cpt_amd import config as cfgmod, synth
    from cpt_amd.modeling_rec import REC_MLM_CPT
    cfg = cfgmod.oscar_base()
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
    m.tie_weights()
    m.to(dev).eval().set_compute_dtype("bf16")
    b = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=3, vary_regions=True).items()}

    def fwd(_):
        with torch.no_grad():
            return (m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0].clone(),)
    bad = _screen(dev, fwd, n=200)
    assert bad == 0, "%d of the forwards differed from the first one" % bad


@pytest.mark.parametrize("B", [32, 4])
def test_training_step_weight_gradients_race_screen(dev, B):
    """Round 3: the paired weight-gradient launches (two TN problems per launch, split-K partials reduced by one launch), the fused
    Q|K|V bias sums, the split-K FFN-down forward and the segment clear that replaced the whole-buffer fill, as the TRAINING STEP runs
    them: 60 forward + backward passes of one batch under the memory-thrashing side stream; every GEMM-written gradient and the loss
    must equal the first pass bit for bit (the vector gradients and the loss sum take atomics: fp32 rounding only)."""
    from cpt_amd import config as cfgmod, synth, train as T
    from cpt_amd.modeling_rec import REC_MLM_CPT
    cfg = cfgmod.oscar_base(num_hidden_layers=2)
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.1
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 3, head="cpt"))
    m.tie_weights()
    m.to(dev).train()
    m.set_compute_dtype("bf16")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=2).items()}
    names = [n for n, p in m.named_parameters() if p.dim() == 2 and "embeddings" not in n and "pooler" not in n]
    params = dict(m.named_parameters())

    def launch(_slot):
        T.set_dropout_seed(m, 77)
        for p in m.parameters():
            p.grad = None
        loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                    mask_token_pos=b["mask_token_pos"])
        loss.backward()
        losses.append(loss.detach().clone())
        return tuple(params[n].grad.clone() for n in names)

    losses = []
    bad = _screen(dev, launch, n=60, ring=1)
    assert bad == 0, "%d tensors of the later passes differed from the first pass" % bad
    ls = torch.stack(losses).double()
    assert float((ls - ls[0]).abs().max()) <= 1e-6 * float(ls[0].abs()), ls        # row losses are added with atomics
