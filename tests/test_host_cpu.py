"""CPU: host-side logic of the drop-in surface -- scoring/IoU vs golden vectors and the oracle,
LR schedule, optimizer grouping codes, config/checkpoint round trip, data-parallel helpers with
world_size 2 on gloo."""
import math
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cpt_amd import config as cfgmod
from cpt_amd import scoring, synth, dist as cdist
from oracle import cpt_oracle as O


def test_iou_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "iou.npz"))
    got = np.array([scoring.compute_iou(list(b[0]), list(b[1])) for b in g["boxes"]])
    assert (got == g["ious"]).all()


def test_lr_schedule_matches_reference_vectors(golden_dir):
    from cpt_amd.train import get_lr_sched
    s = np.load(os.path.join(golden_dir, "lr_sched.npz"))

    class Opt(object):
        learning_rate, warmup_steps, num_train_steps = 3e-5, 50, 500
    got = np.array([get_lr_sched(int(t), Opt) for t in s["steps"]])
    assert (got == s["lrs"]).all()


def test_select_region_matches_oracle_and_ties():
    rng = np.random.Generator(np.random.PCG64(4))
    V = 7000
    for trial in range(20):
        P = int(rng.integers(1, 6))
        scores = torch.from_numpy(rng.standard_normal((P, V)).astype(np.float32))
        sets = [list(rng.choice(synth.COLOR_IDS, size=int(rng.integers(1, 4)), replace=False)) for _ in range(P)]
        rects = [[[i, j, i + 5, j + 7] for j in range(len(s))] for i, s in enumerate(sets)]
        i1, r1, sc1 = scoring.select_region(scores, sets, rects, synth.NONE_ID)
        i0, sc0 = O.select_region_zeroshot(scores, sets, synth.NONE_ID)
        assert i1 == i0 and torch.equal(sc1, sc0)
        i1, r1, sc1 = scoring.select_region(scores, sets, rects, synth.NONE_ID, few_shot=True)
        i0, sc0 = O.select_region_fewshot(scores, sets, synth.NONE_ID)
        assert i1 == i0 and torch.equal(sc1, sc0)
    # first-max tie-break, as torch.argmax in the reference
    s = torch.zeros(2, V)
    s[0, synth.COLOR_IDS[1]] = 3.0
    s[1, synth.COLOR_IDS[0]] = 3.0
    idx, rect, _ = scoring.select_region(s, [[synth.COLOR_IDS[0], synth.COLOR_IDS[1]], [synth.COLOR_IDS[0]]],
                                         [[[0, 0, 1, 1], [1, 1, 2, 2]], [[2, 2, 3, 3]]], synth.NONE_ID)
    assert idx == 1 and rect == [1, 1, 2, 2]


def test_accuracy_counts_iou_over_half():
    preds = {"a": [10, 10, 50, 50], "b": [0, 0, 10, 10]}
    gts = {"a": [10, 10, 41, 41], "b": [100, 100, 20, 20]}
    assert scoring.accuracy(preds, gts) == 50.0


def test_state_dict_keys_and_checkpoint_roundtrip(golden_dir):
    from cpt_amd.modeling_bert import BertImgForPreTraining
    from cpt_amd.modeling_rec import REC_MLM_CPT
    ck = os.path.join(golden_dir, "tiny_ckpt")
    cfg = cfgmod.BertConfig.from_pretrained(ck)
    pre, info = BertImgForPreTraining.from_pretrained(ck, config=cfg, output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info["error_msgs"]
    assert not pre.training                                           # eval() as the reference loader
    e = np.load(os.path.join(golden_dir, "tiny_ckpt_expected.npz"))
    m = REC_MLM_CPT(cfg)
    m.copy_from_pretraining_model(pre)
    assert sorted(m.state_dict().keys()) == list(e["keys"])
    assert m.cls.decoder.weight is m.bert.embeddings.word_embeddings.weight      # tied
    with tempfile.TemporaryDirectory() as d:
        m.save_pretrained(d)
        assert sorted(os.listdir(d)) == ["config.json", "pytorch_model.bin"]
        m2 = REC_MLM_CPT.from_pretrained(d)                            # fewshot/refcoco_cpt.py:503-505
        for k, v in m.state_dict().items():
            assert torch.equal(v, m2.state_dict()[k]), k
        assert m2.config.img_feature_dim == cfg.img_feature_dim
    # size mismatch on cls.seq_relationship is tolerated (modeling_utils.py:858-860), anything else raises
    cfg2 = cfgmod.BertConfig.from_pretrained(ck)
    cfg2.num_contrast_classes = 2
    BertImgForPreTraining.from_pretrained(ck, config=cfg2)
    cfg3 = cfgmod.BertConfig.from_pretrained(ck)
    cfg3.intermediate_size = 256
    with pytest.raises(RuntimeError):
        BertImgForPreTraining.from_pretrained(ck, config=cfg3)


def test_checkpoint_prefix_forms(golden_dir):
    """modeling_utils.py:843-851: a bare-encoder checkpoint fills ``model.bert`` of a head model, a head-model checkpoint fills a
    bare encoder (its head keys are reported unused); legacy gamma / beta names are accepted in both."""
    from cpt_amd.modeling_bert import BertImgForPreTraining, BertImgModel
    ck = os.path.join(golden_dir, "tiny_ckpt")
    cfg = cfgmod.BertConfig.from_pretrained(ck)
    full = BertImgForPreTraining.from_pretrained(ck, config=cfg)
    sd = full.state_dict()
    bare = {k[len("bert."):].replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v
            for k, v in sd.items() if k.startswith("bert.")}
    torch.manual_seed(5)
    m, info = BertImgForPreTraining.from_pretrained(None, config=cfg, state_dict=bare, output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info["error_msgs"]
    for k, v in sd.items():
        if k.startswith("bert."):
            assert torch.equal(m.state_dict()[k], v), k
    enc, info = BertImgModel.from_pretrained(None, config=cfg, state_dict=sd, output_loading_info=True)
    assert not info["missing_keys"] and not info["error_msgs"]
    assert sorted(info["unexpected_keys"]) == sorted(k for k in sd if not k.startswith("bert."))
    for k, v in enc.state_dict().items():
        assert torch.equal(v, sd["bert." + k]), k
    # a key the model does not have is reported, a missing one too
    part = dict(sd)
    part.pop("bert.pooler.dense.bias")
    part["bert.not_a_parameter"] = torch.zeros(1)
    _, info = BertImgForPreTraining.from_pretrained(None, config=cfg, state_dict=part, output_loading_info=True)
    assert info["missing_keys"] == ["bert.pooler.dense.bias"] and info["unexpected_keys"] == ["bert.not_a_parameter"]


def test_unsupported_surface_raises():
    from cpt_amd.modeling_rec import REC_MLM_CPT
    m = REC_MLM_CPT(cfgmod.tiny())
    ids = torch.zeros(1, 4, dtype=torch.long)
    with pytest.raises(NotImplementedError):
        m(ids, head_mask=torch.ones(2))
    with pytest.raises(NotImplementedError):
        m.bert(ids, encoder_history_states=[ids])


def test_optimizer_codes_follow_reference_groups():
    """decay / no-decay split of fewshot/refcoco_cpt.py:320-338 and skip of gradient-less params."""
    from cpt_amd.train import NO_DECAY, NO_GRAD_PREFIXES
    from cpt_amd.engine import pack_order
    cfg = cfgmod.tiny()
    names = pack_order(cfg, "cpt")
    decay = [n for n in names if not n.startswith(NO_GRAD_PREFIXES) and not any(nd in n for nd in NO_DECAY)]
    nodecay = [n for n in names if not n.startswith(NO_GRAD_PREFIXES) and any(nd in n for nd in NO_DECAY)]
    assert "bert.encoder.layer.0.attention.self.query.weight" in decay
    assert "bert.img_embedding.weight" in decay and "bert.embeddings.word_embeddings.weight" in decay
    assert "cls.bias" in nodecay and "bert.LayerNorm.weight" in nodecay and "bert.encoder.layer.1.output.dense.bias" in nodecay
    assert all(n.startswith("bert.pooler.") for n in names if n.startswith(NO_GRAD_PREFIXES))
    # q/k/v weights adjacent, then q/k/v biases adjacent (fused N=3H GEMM)
    i = names.index("bert.encoder.layer.0.attention.self.query.weight")
    assert names[i:i + 6] == ["bert.encoder.layer.0.attention.self.%s.%s" % (a, b) for b in ("weight", "bias")
                              for a in ("query", "key", "value")]


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = cdist.shard_range(n, r, world)
                cover += list(range(lo, hi))
            assert cover == list(range(n))
            sizes = [cdist.shard_range(n, r, world)[1] - cdist.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 11
        lo, hi = cdist.shard_range(n, rank, world)
        # inference: each rank scores its own queries; fixed-shape gather rebuilds global order
        local = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1.0, 10.0]])
        allv = cdist.gather_fixed(local, n)
        assert torch.equal(allv, torch.arange(n, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1.0, 10.0]]))
        # training: one all-reduce over the flat gradient = mean of per-rank gradients
        g = torch.full((1000,), float(rank + 1))
        cdist.allreduce_mean_(g)
        assert torch.allclose(g, torch.full((1000,), (1 + world) / 2.0))
        p = torch.full((10,), float(rank))
        cdist.broadcast_(p, 0)
        assert float(p.sum()) == 0.0
        # DP-averaged AdamW step == single-process step on the concatenated batch gradient mean
        torch.manual_seed(0)
        w0 = torch.randn(64)
        grads = [torch.randn(64) for _ in range(world)]
        gl = grads[rank].clone()
        cdist.allreduce_mean_(gl)
        p1, m1, v1 = O.adamw_step(w0, gl, torch.zeros(64), torch.zeros(64), 1, 1e-3, 0.9, 0.98, 1e-8, 0.01)
        p_ref, _, _ = O.adamw_step(w0, sum(grads) / world, torch.zeros(64), torch.zeros(64), 1, 1e-3, 0.9, 0.98, 1e-8, 0.01)
        assert torch.allclose(p1, p_ref, atol=1e-7)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_data_parallel_helpers_world2_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_dp_worker, args=(2, port, tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok0")) and os.path.exists(os.path.join(tmp, "ok1"))


def test_nspcpt_surface_and_choice_rule(golden_dir):
    """Section 8(f).1 host side: NSPCPT's state-dict surface after copy_from_pretraining_model equals the
    reference's (fixture keys), the label construction and the 1 - softmax[:,1] choice rule reproduce the
    reference's outputs from the reference's own relation scores, and there is no CPU fallback."""
    import numpy as np
    import pytest
    from cpt_amd import config as cfgmod, scoring
    from cpt_amd.modeling_bert import BertImgForPreTraining
    from cpt_amd.modeling_vcr import NSPCPT
    g = np.load(os.path.join(golden_dir, "tiny_vcr_nsp.npz"))
    cfg = cfgmod.tiny()
    m = NSPCPT(cfg)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, dtype=torch.long))                  # before copy_from_pretraining_model
    m.copy_from_pretraining_model(BertImgForPreTraining(cfg))
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    interval = int(g["interval"])
    lab = scoring.nsp_choice_labels([2, 0], interval, 8)
    assert (lab.numpy() == g["cls_labels"]).all()
    logits, preds = scoring.nsp_choose(torch.from_numpy(g["rel"]), interval)
    np.testing.assert_allclose(logits.numpy(), g["choice_logits"], atol=1e-6, rtol=0)
    assert preds == list(g["preds"])
    with pytest.raises(RuntimeError):                             # parameters on the CPU: the HIP path refuses
        with torch.no_grad():
            m(torch.from_numpy(g["in_input_ids"]), torch.from_numpy(g["in_segment_ids"]),
              torch.from_numpy(g["in_attention_mask"]), img_feats=torch.from_numpy(g["in_img_feats"]))


def test_train_batch_helpers_accept_what_the_reference_accepts():
    """ADVICE r4 (fewshot/refcoco_cpt.py:231-243): the label grid is built by index assignment -- int32 and negative [MASK] positions,
    any integer colour dtype -- and only a FIFTH parameter group raises; an optimizer with one to three groups trains."""
    from cpt_amd import drivers
    mask = torch.ones(3, 7, dtype=torch.long)
    batch = {"attention_mask": mask, "mask_token_pos": torch.tensor([2, -1, 0], dtype=torch.int32), "colors": torch.tensor([5, 6, 7], dtype=torch.int32)}
    grid = drivers._label_grid(batch)
    want = torch.full((3, 7), -1, dtype=torch.long)
    want[0, 2], want[1, 6], want[2, 0] = 5, 6, 7
    assert grid.dtype == torch.long and torch.equal(grid, want)
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(5)]
    for n in (1, 2, 3, 4):
        opt = torch.optim.AdamW([{"params": [p]} for p in ps[:n]], lr=1.0)
        drivers._apply_lr(opt, 0.5, 10.0)
        assert [g["lr"] for g in opt.param_groups] == ([5.0, 5.0, 0.5, 0.5])[:n]
    with pytest.raises(ValueError):
        drivers._apply_lr(torch.optim.AdamW([{"params": [p]} for p in ps], lr=1.0), 0.5, 10.0)


def test_warmup_schedules_match_torch_lambdalr_and_transformers():
    """The GQA / VCR few-shot drivers schedule with pytorch_transformers.WarmupLinearSchedule / WarmupConstantSchedule (fewshot/vcr_nsp_cpt.py:386,
    gqa_cpt.py:346-348; not vendored).  cpt_amd.train's classes against torch.optim.lr_scheduler.LambdaLR driven by the INSTALLED transformers'
    get_linear_schedule_with_warmup / get_constant_schedule_with_warmup (today's names of the same lambdas) and against the oracle's multiplier:
    identical learning rates at construction and after every step, for two parameter groups (these were checked by nothing)."""
    import transformers
    from cpt_amd import train as T

    class Opt(object):          # what the schedules need of FusedAdamW / AdamW: param_groups
        def __init__(self):
            self.param_groups = [{"lr": 5e-5, "weight_decay": 0.05}, {"lr": 5e-5, "weight_decay": 0.0}]

    for warm, total in ((0, 20), (4, 20), (7, 7), (30, 20)):
        ps = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        ref_opt = torch.optim.SGD([{"params": [ps[0]]}, {"params": [ps[1]]}], lr=5e-5)
        ref = transformers.get_linear_schedule_with_warmup(ref_opt, warm, total)
        mine = T.WarmupLinearSchedule(Opt(), warmup_steps=warm, t_total=total)
        for step in range(total + 4):
            got = [g["lr"] for g in mine.optimizer.param_groups]
            assert got == [g["lr"] for g in ref_opt.param_groups], (warm, total, step)
            assert got[0] == 5e-5 * O.warmup_linear_schedule(step, warm, total)
            ref_opt.step()
            ref.step()
            mine.step()
        ref_opt = torch.optim.SGD([{"params": [ps[0]]}, {"params": [ps[1]]}], lr=5e-5)
        ref = transformers.get_constant_schedule_with_warmup(ref_opt, warm)
        mine = T.WarmupConstantSchedule(Opt(), warmup_steps=warm)
        for step in range(total):
            assert [g["lr"] for g in mine.optimizer.param_groups] == [g["lr"] for g in ref_opt.param_groups], (warm, step)
            ref_opt.step()
            ref.step()
            mine.step()


def test_oracle_hf_adamw_against_closed_forms():
    """oracle.adamw_step_hf (pytorch_transformers.AdamW restated): with eps = 0 and no decay the first step is p - lr * sign(g) whatever the betas
    (bias corrections cancel); without correct_bias it is p - lr (1 - b1) / sqrt(1 - b2) * sign(g); the decay multiplies the UPDATED parameter; and
    against torch.optim.AdamW the difference is only where eps enters (denominator before vs after the sqrt(bc2) scaling) and the order of the decay."""
    g = torch.tensor([0.3, -2.0, 1e-3, 7.0])
    p = torch.tensor([1.0, -1.0, 0.5, 2.0])
    z = torch.zeros(4)
    lr, b1, b2 = 1e-2, 0.9, 0.999
    p1, m1, v1 = O.adamw_step_hf(p, g, z, z, 1, lr, b1, b2, 0.0, 0.0)
    assert torch.allclose(p1, p - lr * torch.sign(g), atol=1e-7)
    p2, _, _ = O.adamw_step_hf(p, g, z, z, 1, lr, b1, b2, 0.0, 0.0, correct_bias=False)
    assert torch.allclose(p2, p - lr * (1 - b1) / math.sqrt(1 - b2) * torch.sign(g), atol=1e-6)
    p3, _, _ = O.adamw_step_hf(p, g, z, z, 1, lr, b1, b2, 0.0, 0.1)
    assert torch.allclose(p3, p1 * (1 - lr * 0.1), atol=1e-7)
    # three steps against torch.optim.AdamW with eps -> 0 and wd = 0: the two algorithms coincide there
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pt], lr=lr, betas=(b1, b2), eps=1e-30, weight_decay=0.0)
    ph, mh, vh = p.clone(), z.clone(), z.clone()
    for t in range(1, 4):
        gt = g * t
        pt.grad = gt.clone()
        opt.step()
        ph, mh, vh = O.adamw_step_hf(ph, gt, mh, vh, t, lr, b1, b2, 1e-30, 0.0)
        assert torch.allclose(ph, pt.detach(), atol=1e-6), t


def test_hf_adamw_restatements_agree():
    """pytorch_transformers.AdamW has no source to execute here (absent from /root/reference and from the installed transformers),
    so the oracle's float32 restatement (the implementation's in-place call sequence) is held against an independent float64 one written from the
    papers' form (oracle/make_hf_adamw_fixture.py -> tests/golden/hf_adamw.npz): six steps, changing learning rate, gradients over ten orders of
    magnitude, zero-gradient parameters, three hyper-parameter sets incl. correct_bias = False."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_adamw.npz"))
    for tag in ("gqa", "vcr", "nobias"):
        b1, b2, eps, wd, cb = [float(x) for x in g[tag + "_hyper"]]
        p = torch.from_numpy(g[tag + "_p0"]).double()
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for t in range(1, g[tag + "_g"].shape[0] + 1):
            gr = torch.from_numpy(g[tag + "_g"][t - 1]).double()
            p, m, v = O.adamw_step_hf(p, gr, m, v, t, float(g[tag + "_lr"][t - 1]), b1, b2, eps, wd, correct_bias=bool(cb))
            for got, key in ((p, "_p"), (m, "_m"), (v, "_v")):
                ref = torch.from_numpy(g[tag + key][t - 1])
                err = float((got - ref).abs().max() / (ref.abs().max() + 1e-300))
                assert err < 1e-12, (tag, t, key, err)      # same algorithm in float64 through two different algebraic forms
        # ... and in float32, as the tests of the HIP kernel call it
        p32 = torch.from_numpy(g[tag + "_p0"]).float()
        m32 = torch.zeros_like(p32)
        v32 = torch.zeros_like(p32)
        for t in range(1, g[tag + "_g"].shape[0] + 1):
            p32, m32, v32 = O.adamw_step_hf(p32, torch.from_numpy(g[tag + "_g"][t - 1]).float(), m32, v32, t, float(g[tag + "_lr"][t - 1]), b1, b2, eps, wd,
                                            correct_bias=bool(cb))
        ref = torch.from_numpy(g[tag + "_p"][-1])
        assert float((p32.double() - ref).abs().max() / ref.abs().max()) < 3e-6, tag
