"""GPU: the PRODUCT data-parallel training step (train.FusedAdamW with world_size 2: rank-0 broadcast at construction,
per-bucket reduce-scatter started from cpt_train_bwd_ex's callbacks on a side stream, sharded cpt_adamw, parameter
all-gather awaited bucket by bucket inside cpt_train_fwd_ex) against a single-process step on the concatenated batch.
Both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the collectives are the same
torch.distributed calls).  Reference: DistributedDataParallel at /root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:516-522."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from cpt_amd import config as cfgmod
from cpt_amd import synth

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(cfg, seed, dev, mode, wire=None):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    from cpt_amd.train import FusedAdamW
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0      # ranks draw different masks: compare without dropout
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, seed, head="cpt"))
    m.tie_weights()
    m.to(dev).train()
    m.set_compute_dtype(mode)
    opt = FusedAdamW(m, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.01, grad_wire=wire)
    return m, opt


def _run(m, opt, b, rows, steps=STEPS):
    losses = []
    d = {k: v[rows] for k, v in b.items()}
    for _ in range(steps):
        opt.zero_grad()
        loss, _ = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"],
                    masked_lm_labels=d["colors"], mask_token_pos=d["mask_token_pos"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


def _worker(rank, world, port, tmp, mode, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = cfgmod.tiny()
        m, opt = _make(cfg, 1234 + 17 * rank, dev, mode, wire)      # rank 1 starts from OTHER weights: the broadcast must fix that
        assert opt.sync is not None and opt.m.numel() * world == m._engine().flat.numel()      # moments are sharded
        b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}
        per = 4 // world
        losses = _run(m, opt, b, slice(rank * per, (rank + 1) * per))
        m.eval()
        with torch.no_grad():                                        # a parameter all-gather is still pending here
            logits = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                       mask_token_pos=b["mask_token_pos"])[0]
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        osd = opt.state_dict()                                       # collective: gathers the sharded moments
        torch.save({"sd": sd, "losses": losses, "logits": logits.cpu(), "opt": osd}, os.path.join(tmp, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,wire", [("fp32", None), ("bf16", None), ("fp32", "bf16"), ("bf16x3", None)])
def test_product_dp_step_world2_matches_single_process(mode, wire):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), tmp, mode, wire), nprocs=world, join=True)
        r = [torch.load(os.path.join(tmp, "r%d.pt" % i)) for i in range(world)]
    dev = torch.device("cuda:0")
    cfg = cfgmod.tiny()
    m, opt = _make(cfg, 1234, dev, mode)
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}
    losses = _run(m, opt, b, slice(0, 4))
    m.eval()
    with torch.no_grad():
        logits = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                   mask_token_pos=b["mask_token_pos"])[0].cpu()
    exact = mode == "fp32" and wire is None
    x3 = mode == "bf16x3"          # split-operand GEMMs: two half batches against one whole batch differ by ~1e-5 of the gradient scale
    ptol, ltol = (2e-6, 1e-5) if exact else ((2e-4, 2e-4) if x3 else (4e-3, 5e-2))
    # replicas identical to each other bit for bit, and equal to the single-process run on the concatenated batch
    for k, v in m.state_dict().items():
        assert torch.equal(r[0]["sd"][k], r[1]["sd"][k]), k
        if k.endswith(".key.bias"):
            continue      # its true gradient is 0 (softmax is shift-invariant): Adam normalises pure rounding noise there
        err = (r[0]["sd"][k] - v.cpu()).abs().max().item()
        assert err < ptol, (k, err)
    assert torch.equal(r[0]["logits"], r[1]["logits"])
    assert (r[0]["logits"] - logits).abs().max().item() < (1e-4 if exact else (1e-3 if x3 else 0.15))
    for s in range(STEPS):
        mean = 0.5 * (r[0]["losses"][s] + r[1]["losses"][s])        # equal labelled-row counts per rank
        assert abs(mean - losses[s]) < ltol, (s, mean, losses[s])
    ref = opt.state_dict()
    for k in ref["state"]:
        e = (r[0]["opt"]["state"][k]["exp_avg"] - ref["state"][k]["exp_avg"]).abs().max().item()
        assert e < (1e-6 if exact else (1e-4 if x3 else 1e-2)), (k, e)
    assert r[0]["opt"]["step_count"] == STEPS


# ---- the RCCL branch itself ("nccl" backend: communication stream, events, work.wait() as a stream dependency) -----------------
# World size 2 needs two devices (RCCL refuses two ranks on one GPU) and the GPU test box has one, so the branch is driven
# at world size 1 with force_collectives: every bucket goes through reduce_scatter_tensor / all_gather_into_tensor on the
# communication stream exactly as on eight GPUs; the sum over one rank is the identity, so the run must reproduce the plain
# single-process step BIT FOR BIT.  (replaces DistributedDataParallel at fewshot/refcoco_cpt.py:516-522.)

def _nccl1_worker(rank, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch.distributed as dist
    from cpt_amd.modeling_rec import REC_MLM_CPT
    from cpt_amd.train import FusedAdamW
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    try:
        cfg = cfgmod.tiny()
        b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}

        def make(mode, **kw):
            c = cfgmod.tiny()
            c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
            m = REC_MLM_CPT(c)
            m.load_state_dict(synth.init_state_dict(c, 1234, head="cpt"))
            m.tie_weights()
            m.to(dev).train()
            m.set_compute_dtype(mode)
            return m, FusedAdamW(m, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.01, force_collectives=True, **kw)

        for mode, wire in (("fp32", None), ("bf16", None), ("fp32", "bf16")):
            m, opt = make(mode, grad_wire=wire)
            assert opt.sync is not None and opt.sync.collectives and opt.sync.comm is not None
            l1 = _run(m, opt, b, slice(0, 4), steps=1)
            m.eval()
            with torch.no_grad():      # (waits for the pending parameter all-gather bucket by bucket)
                m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])
            m.train()
            sd1 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            losses = l1 + _run(m, opt, b, slice(0, 4), steps=STEPS - 1)
            sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            out[(mode, wire)] = {"losses": losses, "sd": sd, "sd1": sd1}
        # comm profiling (bench.py --mode train prints this block): events around every collective and around every wait for one
        m, opt = make("bf16")
        _run(m, opt, b, slice(0, 4), steps=2)
        opt.sync.profile = True
        opt.sync.comm_report(1)                       # (clears)
        _run(m, opt, b, slice(0, 4), steps=3)
        out["comm"] = opt.sync.comm_report(3)
        opt.sync.profile = False
        # stress: 50 steps back to back while a second stream thrashes HBM / MALL (tests/test_gpu_race.py's pressure)
        m, opt = make("bf16")
        side = torch.cuda.Stream(device=dev)
        ja = torch.empty(128 * 1024 * 1024 // 4, device=dev)
        jb = torch.empty_like(ja)
        with torch.cuda.stream(side):
            for _ in range(200):
                jb.copy_(ja)
                ja.copy_(jb)
        losses = _run(m, opt, b, slice(0, 4), steps=50)
        m.eval()
        with torch.no_grad():          # the last step's all-gather is still pending here: the forward must wait for it per bucket
            logits = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
        torch.cuda.synchronize()
        out["stress"] = {"losses": losses, "sd": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "logits": logits.cpu()}
        # an in-place edit of p.grad after backward cannot reach the update once the reduce-scatter is in flight: step() refuses
        m, opt = make("fp32")
        d = b
        opt.zero_grad()
        loss, _ = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=d["colors"],
                    mask_token_pos=d["mask_token_pos"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1e-3)       # the reference's VCR loop does this (fewshot/vcr_nsp_cpt.py:461)
        try:
            opt.step()
            out["edit_raises"] = False
        except RuntimeError as e:
            out["edit_raises"] = "clip_grad_norm_" in str(e)
        # defer_reduce=True: local gradients stay editable, the reduce-scatter runs inside step()
        m, opt = make("fp32", defer_reduce=True)
        opt.zero_grad()
        loss, _ = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], masked_lm_labels=d["colors"],
                    mask_token_pos=d["mask_token_pos"])
        loss.backward()
        for p_ in m.parameters():
            if p_.grad is not None:
                p_.grad.mul_(0.5)
        opt.step()
        m.eval()
        out["deferred"] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        # no_sync(): the first micro-step of an accumulation window sends nothing; the sum is reduced once
        m, opt = make("fp32")
        opt.zero_grad()
        for i, rows in enumerate((slice(0, 2), slice(2, 4))):
            dd = {k: v[rows] for k, v in b.items()}
            ctx = opt.no_sync() if i == 0 else None
            if ctx is not None:
                ctx.__enter__()
            loss, _ = m(dd["input_ids"], dd["segment_ids"], dd["attention_mask"], img_feats=dd["img_feats"], masked_lm_labels=dd["colors"],
                        mask_token_pos=dd["mask_token_pos"])
            loss.backward()
            if ctx is not None:
                ctx.__exit__(None, None, None)
                assert not opt.sync.reduced                       # nothing was sent by the first micro-step
        opt.step()
        m.eval()
        out["accum"] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        torch.save(out, os.path.join(tmp, "nccl1.pt"))
    finally:
        dist.destroy_process_group()


def test_rccl_path_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_nccl1_worker, args=(_free_port(), tmp), nprocs=1, join=True)
        r = torch.load(os.path.join(tmp, "nccl1.pt"))
    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfgmod.tiny(), seed=5, max_seq_len=20, img_seq_len=6).items()}
    # The embedding scatter-add and the bias / LayerNorm column sums of the backward pass use fp32 atomics (order-dependent last
    # bits), so two runs of the SAME step agree bit for bit only on gradients that do not pass through them: after ONE step
    # every encoder weight matrix must be identical to the plain step (the collectives of one rank are the identity), the
    # rest to rounding; later steps to the tolerances of the world-2 test.
    for mode, wire in (("fp32", None), ("bf16", None), ("fp32", "bf16")):
        m, opt = _make(cfgmod.tiny(), 1234, dev, mode)
        assert opt.sync is None
        l1 = _run(m, opt, b, slice(0, 4), steps=1)
        got = r[(mode, wire)]
        ptol, ltol = (2e-6, 1e-5) if (mode == "fp32" and wire is None) else (4e-3, 5e-2)
        if wire is None:
            # (the mean over the labelled rows is accumulated with fp32 atomics in ce_rows: the row order, hence the last bit, may differ between runs)
            assert abs(got["losses"][0] - l1[0]) <= 2e-7 * max(1.0, abs(l1[0])), (mode, got["losses"], l1)
            for k, v in m.state_dict().items():
                if k.startswith("bert.encoder.") and k.endswith("weight") and v.dim() == 2:      # GEMM-produced gradients: no atomics on their path
                    assert torch.equal(got["sd1"][k], v.cpu()), (mode, k)
                else:
                    assert (got["sd1"][k] - v.cpu()).abs().max().item() < ptol, (mode, k)
        losses = l1 + _run(m, opt, b, slice(0, 4), steps=STEPS - 1)
        for a_, b_ in zip(got["losses"], losses):
            assert abs(a_ - b_) < ltol, (mode, wire, got["losses"], losses)
        for k, v in m.state_dict().items():
            if not k.endswith(".key.bias"):
                assert (got["sd"][k] - v.cpu()).abs().max().item() < ptol, (mode, wire, k)
    # stress run: finite, tracks the plain run, and the eval forward right behind the last step saw exactly the final
    # parameters (its logits are reproduced bit for bit by a fresh model loaded with the final state dict: no stale shadow
    # copy, no parameter read ahead of its all-gather)
    m, opt = _make(cfgmod.tiny(), 1234, dev, "bf16")
    losses = _run(m, opt, b, slice(0, 4), steps=50)
    assert all(l == l and abs(l) < 1e4 for l in r["stress"]["losses"])
    assert max(abs(a_ - b_) for a_, b_ in zip(r["stress"]["losses"], losses)) < 0.15, (r["stress"]["losses"][-3:], losses[-3:])
    from cpt_amd.modeling_rec import REC_MLM_CPT
    c = cfgmod.tiny()
    fresh = REC_MLM_CPT(c)
    fresh.load_state_dict(r["stress"]["sd"])
    fresh.tie_weights()
    fresh.to(dev).eval().set_compute_dtype("bf16")
    with torch.no_grad():
        again = fresh(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0].cpu()
    assert torch.equal(again, r["stress"]["logits"])
    assert r["edit_raises"] is True
    # the comm block of bench.py --mode train: one reduce-scatter and one all-gather per bucket per step, positive times, waits <= totals
    c = r["comm"]
    assert c["ranks"] == 1 and c["buckets"] == len(c["per_bucket"]) >= 3
    assert c["reduce_scatter_ms_per_step"] > 0 and c["all_gather_ms_per_step"] > 0
    assert c["reduce_scatter_bytes_per_rank_per_step"] == 4 * c["elements"] == c["all_gather_bytes_per_rank_per_step"]
    assert all(v["rs_ms"] > 0 and v["ag_ms"] > 0 for v in c["per_bucket"].values())
    assert c["fraction_hidden"]["reduce_scatter"] <= 1.0 and c["fraction_hidden"]["all_gather"] <= 1.0
    # defer_reduce: the halved local gradients reached the update -> equal to the plain step with the same edit
    m, opt = _make(cfgmod.tiny(), 1234, dev, "fp32")
    opt.zero_grad()
    loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"],
                mask_token_pos=b["mask_token_pos"])
    loss.backward()
    for p_ in m.parameters():
        if p_.grad is not None:
            p_.grad.mul_(0.5)
    opt.step()
    for k, v in m.state_dict().items():
        assert (r["deferred"][k] - v.cpu()).abs().max().item() < 2e-6, k
    # accumulation over two micro-batches
    m, opt = _make(cfgmod.tiny(), 1234, dev, "fp32")
    opt.zero_grad()
    for rows in (slice(0, 2), slice(2, 4)):
        dd = {k: v[rows] for k, v in b.items()}
        loss, _ = m(dd["input_ids"], dd["segment_ids"], dd["attention_mask"], img_feats=dd["img_feats"], masked_lm_labels=dd["colors"],
                    mask_token_pos=dd["mask_token_pos"])
        loss.backward()
    opt.step()
    for k, v in m.state_dict().items():
        assert (r["accum"][k] - v.cpu()).abs().max().item() < 2e-6, k


def _nccl2_worker(rank, world, port, tmp, mode, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch.distributed as dist
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg = cfgmod.tiny()
        m, opt = _make(cfg, 1234 + 17 * rank, dev, mode, wire)
        b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}
        per = 4 // world
        losses = _run(m, opt, b, slice(rank * per, (rank + 1) * per))
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        torch.save({"sd": sd, "losses": losses}, os.path.join(tmp, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,wire", [("fp32", None), ("fp32", "bf16")])
def test_product_dp_step_two_gpus_rccl(mode, wire):
    """the same comparison over RCCL with one rank per device; skipped on boxes with fewer than two GPUs."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_nccl2_worker, args=(world, _free_port(), tmp, mode, wire), nprocs=world, join=True)
        r = [torch.load(os.path.join(tmp, "r%d.pt" % i)) for i in range(world)]
    dev = torch.device("cuda:0")
    m, opt = _make(cfgmod.tiny(), 1234, dev, mode)
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfgmod.tiny(), seed=5, max_seq_len=20, img_seq_len=6).items()}
    losses = _run(m, opt, b, slice(0, 4))
    ptol, ltol = (2e-6, 1e-5) if wire is None else (4e-3, 5e-2)
    for k, v in m.state_dict().items():
        assert torch.equal(r[0]["sd"][k], r[1]["sd"][k]), k
        if not k.endswith(".key.bias"):
            assert (r[0]["sd"][k] - v.cpu()).abs().max().item() < ptol, k
    for s_ in range(STEPS):
        assert abs(0.5 * (r[0]["losses"][s_] + r[1]["losses"][s_]) - losses[s_]) < ltol


def test_stale_training_forward_raises():
    """loss_a = model(a); loss_b = model(b); loss_a.backward() must not consume b's activations."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    cfg = cfgmod.tiny()
    m, opt = _make(cfg, 1234, dev, "fp32")
    b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=5, max_seq_len=20, img_seq_len=6).items()}

    def fwd(rows):
        d = {k: v[rows] for k, v in b.items()}
        return m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"],
                 masked_lm_labels=d["colors"], mask_token_pos=d["mask_token_pos"])[0]
    la = fwd(slice(0, 2))
    lb = fwd(slice(2, 4))
    with pytest.raises(RuntimeError, match="stale training forward"):
        la.backward()
    lb.backward()                                                   # the latest forward still has its activations


# ---- inference: the val loop sharded over two ranks ---------------------------------------------------------
# Whole queries are sharded (cdist.shard_range), no collective runs inside the model, ONE fixed-shape gather of the chosen indices closes the
# loop (replaces the pickled-dict all_gather of utils/comm.py:102-142 called from zeroshot/refcoco_cpt.py:256).  Both ranks share cuda:0
# over gloo; 7 queries do not divide by 2, so the shards differ in size.

def _val_queries(cfg, n, seed):
    import numpy as np
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


def _val_model(cfg, dev, mode):
    from cpt_amd.modeling_rec import REC_MLM_CPT
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 21, head="cpt"))
    m.tie_weights()
    m.to(dev).eval()
    m.set_compute_dtype(mode)
    return m


def _val_worker(rank, world, port, tmp, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpt_amd import drivers
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = cfgmod.oscar_base(num_hidden_layers=2)
        m = _val_model(cfg, dev, mode)
        qs = _val_queries(cfg, 7, 3)
        out = {fs: drivers.val_queries(m, qs, synth.NONE_ID, dev, few_shot=fs, batch_queries=3) for fs in (False, True)}
        torch.save(out, os.path.join(tmp, "v%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_val_queries_two_ranks_equal_single_process(mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cpt_amd import drivers
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_val_worker, args=(world, _free_port(), tmp, mode), nprocs=world, join=True)
        r = [torch.load(os.path.join(tmp, "v%d.pt" % i)) for i in range(world)]
    dev = torch.device("cuda:0")
    cfg = cfgmod.oscar_base(num_hidden_layers=2)
    m = _val_model(cfg, dev, mode)
    qs = _val_queries(cfg, 7, 3)
    for fs in (False, True):
        single = drivers.val_queries(m, qs, synth.NONE_ID, dev, few_shot=fs, batch_queries=3)
        assert r[0][fs] == r[1][fs], "every rank holds the whole result"
        # sequences are independent and every kernel is batch-invariant: the sharded loop scores each query with the same bits
        assert r[0][fs] == single, (fs, r[0][fs], single)
        assert sorted(single) == list(range(7)) and all(v[0] >= 0 and v[1] is not None for v in single.values())


# ---- the C ABI's own communicator block (cpt_comm_*, include/cpt_hip.h): RCCL bound at run time, for hosts without torch.distributed ----

def _comm_worker(rank, world, tmp):
    import ctypes as C
    import time
    from cpt_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:%d" % rank)
    torch.cuda.set_device(dev)
    idf = os.path.join(tmp, "id.bin")
    buf = C.create_string_buffer(128)
    if rank == 0:
        L.check(lib.cpt_comm_unique_id(buf), "cpt_comm_unique_id")
        open(idf + ".tmp", "wb").write(buf.raw)
        os.replace(idf + ".tmp", idf)                       # the id travels out of band: here a file
    else:
        for _ in range(600):
            if os.path.exists(idf):
                break
            time.sleep(0.05)
        buf.raw = open(idf, "rb").read()
    assert lib.cpt_allreduce_grads(torch.zeros(4, device=dev).data_ptr(), 4, L.CPT_F32, L.stream_ptr()) != 0      # no communicator yet
    assert b"cpt_comm_init" in lib.cpt_last_error()
    L.check(lib.cpt_comm_init(rank, world, buf), "cpt_comm_init")
    assert lib.cpt_comm_init(rank, world, buf) != 0                                                                # one communicator per process
    r, n = C.c_int(-1), C.c_int(-1)
    L.check(lib.cpt_comm_rank(C.byref(r), C.byref(n)), "cpt_comm_rank")
    assert (r.value, n.value) == (rank, world)
    N = 1 << 20
    g = torch.arange(N, device=dev, dtype=torch.float32) % 97 + rank
    L.check(lib.cpt_allreduce_grads(g.data_ptr(), N, L.CPT_F32, L.stream_ptr()), "cpt_allreduce_grads")
    want = (torch.arange(N, device=dev, dtype=torch.float32) % 97) * world + world * (world - 1) / 2
    assert torch.equal(g, want)
    gb = ((torch.arange(N, device=dev) % 13).float() + rank).to(torch.bfloat16)
    L.check(lib.cpt_allreduce_grads(gb.data_ptr(), N, L.CPT_BF16, L.stream_ptr()), "cpt_allreduce_grads(bf16)")
    assert torch.equal(gb.float(), (torch.arange(N, device=dev) % 13).float() * world + world * (world - 1) / 2)
    # reduce-scatter -> "update" -> all-gather: the sharded form of the same step
    src = torch.arange(N, device=dev, dtype=torch.float32) % 31 - rank
    shard = torch.empty(N // world, device=dev)
    L.check(lib.cpt_reduce_scatter(src.data_ptr(), shard.data_ptr(), N // world, L.CPT_F32, L.stream_ptr()), "cpt_reduce_scatter")
    full = (torch.arange(N, device=dev, dtype=torch.float32) % 31) * world - world * (world - 1) / 2
    assert torch.equal(shard, full[rank * (N // world):(rank + 1) * (N // world)])
    out = torch.empty(N, device=dev)
    L.check(lib.cpt_allgather(shard.data_ptr(), out.data_ptr(), N // world, L.CPT_F32, L.stream_ptr()), "cpt_allgather")
    assert torch.equal(out, full)
    assert lib.cpt_allgather(shard.data_ptr(), out.data_ptr(), N // world, 5, L.stream_ptr()) != 0                  # bad dtype
    torch.cuda.synchronize(dev)
    L.check(lib.cpt_comm_destroy(), "cpt_comm_destroy")
    L.check(lib.cpt_comm_destroy(), "cpt_comm_destroy (idempotent)")
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


@pytest.mark.parametrize("world", [1, 2])
def test_c_abi_communicator_block(world):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_comm_worker, args=(world, tmp), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(tmp, "ok%d" % r)) for r in range(world))
