"""Few-shot training step on the HIP library: loss with gradients, fused AdamW, LR schedule.

Counterparts of the reference's few-shot driver pieces:
  * ``loss, _ = model(..., masked_lm_labels=mlm_labels); loss.backward()``
    (/root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:245-248) -> ``mlm_loss_with_grad`` (called by
    ``REC_MLM_CPT.forward`` when gradients are enabled): cpt_train_fwd / cpt_train_bwd.
  * ``build_optimizer`` (:318-343, torch.optim.AdamW with 4 groups) -> ``build_optimizer`` /
    ``FusedAdamW`` (one cpt_adamw launch over the flat parameter buffer).
  * ``get_lr_sched`` / ``warmup_linear`` (Oscar/oscar/utils/optim_sched.py:16-20,39-45).
Dropout is not applied in the HIP training path (the reference trains with p=0.1; bitwise parity
under dropout is impossible, gradient parity is tested with dropout disabled).
"""
import ctypes as C

import torch

from . import _lib as L

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")       # fewshot/refcoco_cpt.py:320
# parameters that receive no gradient on the MLM path (pooler is computed but unused by the loss;
# the reference wraps the model with find_unused_parameters=True for exactly this reason)
NO_GRAD_PREFIXES = ("bert.pooler.", "cls.seq_relationship.")


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, opts):
    lr_this_step = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    if lr_this_step <= 0:
        lr_this_step = 1e-8
    return lr_this_step


class _TrainState(object):
    """Per-engine training buffers: flat gradient (same layout as the flat parameters), workspace."""

    def __init__(self, eng):
        self.eng = eng
        self.grad = None
        self.gdesc = None
        self.ws = None
        self.saved = None

    def ensure(self):
        eng = self.eng
        if self.grad is None or self.grad.numel() != eng.flat.numel() or self.grad.device != eng.flat.device:
            self.grad = torch.zeros_like(eng.flat)
            self.gdesc = None
        if self.gdesc is None:
            cfg = eng.cfg
            base = self.grad.data_ptr()

            def gp(n):
                return base + eng.offsets[n][0] * 4

            layers = (L.LayerGrads * cfg.num_hidden_layers)()
            for i in range(cfg.num_hidden_layers):
                p = "bert.encoder.layer.%d." % i
                y = layers[i]
                y.w_qkv = gp(p + "attention.self.query.weight")
                y.b_qkv = gp(p + "attention.self.query.bias")
                y.w_ao = gp(p + "attention.output.dense.weight")
                y.b_ao = gp(p + "attention.output.dense.bias")
                y.ln1_g = gp(p + "attention.output.LayerNorm.weight")
                y.ln1_b = gp(p + "attention.output.LayerNorm.bias")
                y.w_in = gp(p + "intermediate.dense.weight")
                y.b_in = gp(p + "intermediate.dense.bias")
                y.w_out = gp(p + "output.dense.weight")
                y.b_out = gp(p + "output.dense.bias")
                y.ln2_g = gp(p + "output.LayerNorm.weight")
                y.ln2_b = gp(p + "output.LayerNorm.bias")
            g = L.ModelGrads()
            g.word_emb = gp("bert.embeddings.word_embeddings.weight")
            g.pos_emb = gp("bert.embeddings.position_embeddings.weight")
            g.type_emb = gp("bert.embeddings.token_type_embeddings.weight")
            g.emb_ln_g = gp("bert.embeddings.LayerNorm.weight")
            g.emb_ln_b = gp("bert.embeddings.LayerNorm.bias")
            g.w_img = gp("bert.img_embedding.weight")
            g.b_img = gp("bert.img_embedding.bias")
            if "bert.LayerNorm.weight" in eng.offsets:
                g.img_ln_g = gp("bert.LayerNorm.weight")
                g.img_ln_b = gp("bert.LayerNorm.bias")
            g.layers = C.cast(layers, C.POINTER(L.LayerGrads))
            hp = "cls." if eng.head == "cpt" else "cls.predictions."
            g.w_tr = gp(hp + "transform.dense.weight")
            g.b_tr = gp(hp + "transform.dense.bias")
            g.tr_ln_g = gp(hp + "transform.LayerNorm.weight")
            g.tr_ln_b = gp(hp + "transform.LayerNorm.bias")
            g.b_dec = gp(hp + "bias")
            self.gdesc = (g, layers)

    def workspace(self, B, Lt, Li):
        eng = self.eng
        m, _ = eng.descriptor()
        need = L.lib().cpt_train_workspace_bytes(C.byref(m.dims), B, Lt, Li)
        if need == 0:
            raise RuntimeError("cpt_amd: cpt_train_workspace_bytes rejected the batch shape")
        if self.ws is None or self.ws.numel() < need or self.ws.device != eng.flat.device:
            self.ws = torch.empty(need, device=eng.flat.device, dtype=torch.uint8)
        return self.ws


def _state(eng):
    st = eng.__dict__.get("_train_state")
    if st is None:
        st = _TrainState(eng)
        eng.__dict__["_train_state"] = st
    return st


def _named_grad_views(eng, st):
    named = eng._named()
    out = []
    for n, p in named.items():
        if n.startswith(NO_GRAD_PREFIXES):
            out.append((p, None))
        else:
            off, num = eng.offsets[n]
            out.append((p, st.grad[off:off + num].view(p.shape)))
    return out


class _MLMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, eng, tensors):
        ctx.eng = eng
        st = _state(eng)
        ids, seg, mask, pos, feats, mpos, labels = tensors
        B, Lt = ids.shape
        Li = feats.size(1) if feats is not None else 0
        m, _ = eng.descriptor()
        dev = eng.flat.device
        logits = torch.empty((B, eng.cfg.vocab_size), device=dev, dtype=torch.float32)
        loss_acc = torch.empty(2, device=dev, dtype=torch.float32)
        o = L.Outputs(logits=logits.data_ptr(), loss=loss_acc.data_ptr())
        bt = L.Batch(B=B, Lt=Lt, Li=Li, input_ids=ids.data_ptr(), token_type=L.ptr(seg), position_ids=L.ptr(pos),
                     attn_mask=L.ptr(mask), img_feats=L.ptr(feats), mask_pos=mpos.data_ptr(), labels=labels.data_ptr())
        ws = st.workspace(B, Lt, Li)
        L.check(L.lib().cpt_train_fwd(C.byref(m), C.byref(bt), C.byref(o), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                "cpt_train_fwd")
        st.saved = (bt, tensors)          # keep the input tensors alive until backward
        ctx.logits = logits
        ctx.mark_non_differentiable(logits)
        return loss_acc[0] / loss_acc[1], logits

    @staticmethod
    def backward(ctx, grad_loss, _grad_logits):
        eng = ctx.eng
        st = _state(eng)
        if st.saved is None:
            raise RuntimeError("cpt_amd: backward called twice (activations of the training forward were released)")
        bt, tensors = st.saved
        st.ensure()
        m, _ = eng.descriptor()
        views = _named_grad_views(eng, st)
        accumulate = any(p.grad is not None for p, v in views if v is not None)
        if accumulate:
            keep = st.grad.clone()
        st.grad.zero_()
        g, _ = st.gdesc
        L.check(L.lib().cpt_train_bwd(C.byref(m), C.byref(bt), C.byref(g), float(grad_loss), st.ws.data_ptr(), st.ws.numel(),
                                      L.stream_ptr()), "cpt_train_bwd")
        if accumulate:
            st.grad.add_(keep)
        for p, v in views:
            p.grad = v
        st.saved = None
        return None, None, None


def mlm_loss_with_grad(model, input_ids, token_type_ids, attention_mask, labels, position_ids, img_feats, mask_token_pos):
    """(loss, prediction_scores) with autograd history, as REC_MLM_CPT.forward returns them
    (modeling_rec.py:147-152).  ``mask_token_pos`` is required: the loss only sees the [MASK] rows
    (fewshot/refcoco_cpt.py:231-233 puts -1 everywhere else)."""
    if mask_token_pos is None:
        raise NotImplementedError("cpt_amd: training needs mask_token_pos (the (B, L) label grid of the reference has "
                                  "exactly one labelled position per row: pass it as mask_token_pos)")
    eng = model._engine()
    eng.ensure_packed()
    eng.refresh_shadow()
    st = _state(eng)
    st.ensure()

    def prep(t, dt, name):
        if t is None:
            return None
        if not t.is_cuda:
            raise RuntimeError("cpt_amd: %s must be on the GPU" % name)
        return t.to(dt).contiguous()

    if attention_mask is not None and attention_mask.dim() != 2:
        raise NotImplementedError("cpt_amd: only 2-D attention_mask is supported")
    tensors = (prep(input_ids, torch.int64, "input_ids"), prep(token_type_ids, torch.int64, "token_type_ids"),
               prep(attention_mask, torch.int64, "attention_mask"), prep(position_ids, torch.int64, "position_ids"),
               prep(img_feats, torch.float32, "img_feats"), prep(mask_token_pos, torch.int64, "mask_token_pos"),
               prep(labels, torch.int64, "labels"))
    trigger = torch.zeros((), device=eng.flat.device, requires_grad=True)
    loss, logits = _MLMLoss.apply(trigger, eng, tensors)
    return (loss, logits)


class FusedAdamW(object):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt) as ONE
    kernel over the flat parameter / gradient / moment buffers; also refreshes the bf16 shadow.
    Exposes ``param_groups`` with the reference's four groups so ``param_group['lr'] = ...``
    scheduling code (fewshot/refcoco_cpt.py:237-243) keeps working; with data parallelism the flat
    gradient is all-reduced once (sum) and averaged inside the update."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.model = model
        self.eng = model._engine()
        self.betas = betas
        self.eps = eps
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "params": []},
                             {"lr": lr, "weight_decay": 0.0, "params": []},
                             {"lr": lr, "weight_decay": weight_decay, "params": []},
                             {"lr": lr, "weight_decay": 0.0, "params": []}]
        self.step_count = 0
        self.m = self.v = self.code = None

    def _ensure(self):
        eng = self.eng
        eng.ensure_packed()
        if self.m is None or self.m.numel() != eng.flat.numel() or self.m.device != eng.flat.device:
            self.m = torch.zeros_like(eng.flat)
            self.v = torch.zeros_like(eng.flat)
            code = torch.zeros(eng.flat.numel(), dtype=torch.uint8)
            for n in eng.offsets:
                off, num = eng.offsets[n]
                if n.startswith(NO_GRAD_PREFIXES):
                    c = 0
                elif any(nd in n for nd in NO_DECAY):
                    c = 2
                else:
                    c = 1
                code[off:off + num] = c
            self.code = code.to(eng.flat.device)

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def step(self):
        import torch.distributed as dist
        eng = self.eng
        self._ensure()
        st = _state(eng)
        if st.grad is None:
            raise RuntimeError("cpt_amd: optimizer.step() before any backward")
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(st.grad)                     # ONE collective over all 111.68 M gradients
            scale = 1.0 / dist.get_world_size()
        self.step_count += 1
        lr = self.param_groups[2]["lr"]
        wd = self.param_groups[2]["weight_decay"]
        shadow = eng.flat_lp.data_ptr() if (eng.dtype == "bf16" and eng.flat_lp is not None) else None
        L.check(L.lib().cpt_adamw(eng.flat.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                  self.code.data_ptr(), shadow, eng.flat.numel(), lr, self.betas[0], self.betas[1], self.eps,
                                  wd, self.step_count, scale, L.stream_ptr()), "cpt_adamw")
        eng.weights_updated(shadow_fresh=shadow is not None)


    # ---- checkpointing: the reference saves no optimizer state (utils/save_model.py), so a few-shot
    # run cannot resume; here the moments travel with the model, keyed by parameter NAME ------------------------------
    def state_dict(self):
        """{"state": {name: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...], "betas", "eps"}: the same
        per-parameter entries as torch.optim.AdamW.state_dict(), as CPU tensors."""
        self._ensure()
        state = {}
        for n, (off, num) in self.eng.offsets.items():
            shape = tuple(self.eng._named()[n].shape) if hasattr(self.eng, "_named") else (num,)
            state[n] = {"step": self.step_count,
                        "exp_avg": self.m[off:off + num].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": self.v[off:off + num].view(shape).detach().cpu().clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"state": state, "param_groups": groups, "betas": tuple(self.betas), "eps": self.eps,
                "step_count": self.step_count}

    def load_state_dict(self, sd):
        self._ensure()
        missing = [n for n in self.eng.offsets if n not in sd["state"]]
        if missing:
            raise RuntimeError("cpt_amd: optimizer state lacks %d parameters, e.g. %s" % (len(missing), missing[0]))
        for n, (off, num) in self.eng.offsets.items():
            e = sd["state"][n]
            if e["exp_avg"].numel() != num:
                raise RuntimeError("cpt_amd: optimizer state of %s has %d elements, expected %d" % (n, e["exp_avg"].numel(), num))
            self.m[off:off + num].copy_(e["exp_avg"].reshape(-1))
            self.v[off:off + num].copy_(e["exp_avg_sq"].reshape(-1))
        self.step_count = int(sd.get("step_count", next(iter(sd["state"].values()))["step"]))
        self.betas = tuple(sd.get("betas", self.betas))
        self.eps = sd.get("eps", self.eps)
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in src.items() if k != "params"})


def save_checkpoint(save_dir, model, optimizer=None, global_step=0, extra=None):
    """HF-layout checkpoint (config.json + pytorch_model.bin via save_pretrained, as utils/save_model.py) PLUS
    optimizer.pt and training_state.json, so that a few-shot run resumes where it stopped."""
    import json
    import os
    os.makedirs(save_dir, exist_ok=True)
    model.save_pretrained(save_dir)
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(save_dir, "optimizer.pt"))
    with open(os.path.join(save_dir, "training_state.json"), "w") as f:
        json.dump({"global_step": int(global_step), "extra": extra or {}}, f)


def load_checkpoint(save_dir, model, optimizer=None):
    """Loads weights in place (through the packed views, so the HIP path sees them) and the optimizer moments;
    returns the saved global step."""
    import json
    import os
    sd = torch.load(os.path.join(save_dir, "pytorch_model.bin"), map_location="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if [k for k in missing if "decoder.weight" not in k]:
        raise RuntimeError("cpt_amd: checkpoint lacks parameters: %s" % missing[:3])
    model.tie_weights()
    if optimizer is not None:
        optimizer.load_state_dict(torch.load(os.path.join(save_dir, "optimizer.pt"), map_location="cpu"))
    p = os.path.join(save_dir, "training_state.json")
    return json.load(open(p))["global_step"] if os.path.exists(p) else 0


def build_optimizer(model, opts):
    """fewshot/refcoco_cpt.py:318-343 (opts.learning_rate, opts.weight_decay, opts.betas)."""
    return FusedAdamW(model, lr=opts.learning_rate, betas=tuple(opts.betas), weight_decay=opts.weight_decay)
