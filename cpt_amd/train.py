"""Few-shot training step on the HIP library: loss with gradients, fused AdamW, LR schedule.

Counterparts of the reference's few-shot driver pieces:
  * ``loss, _ = model(..., masked_lm_labels=mlm_labels); loss.backward()``
    (/root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:245-248) -> ``mlm_loss_with_grad`` (called by
    ``REC_MLM_CPT.forward`` when gradients are enabled): cpt_train_fwd / cpt_train_bwd.
  * ``build_optimizer`` (:318-343, torch.optim.AdamW with 4 groups) -> ``build_optimizer`` /
    ``FusedAdamW`` (one cpt_adamw launch over the flat parameter buffer).
  * ``get_lr_sched`` / ``warmup_linear`` (Oscar/oscar/utils/optim_sched.py:16-20,39-45).
Dropout (the reference trains with --drop_out 0.1) is applied when the module is in training mode: counter-based masks
regenerated in backward (csrc/dropout.h); gradient parity against the reference is tested with dropout disabled
(bit-identical to the dropout-free kernels) and, with dropout on, against a CPU restatement fed the exported masks.
"""
import ctypes as C

import torch

from . import _lib as L

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")       # fewshot/refcoco_cpt.py:320
# parameters that receive no gradient on the MLM path (pooler is computed but unused by the loss;
# the reference wraps the model with find_unused_parameters=True for exactly this reason)
NO_GRAD_PREFIXES = ("bert.pooler.", "cls.seq_relationship.")


def no_grad_prefixes(head):
    """Parameters outside the loss's graph: the MLM heads never touch the pooler / relation head; the NSP-CPT head
    (NSPCPT, modeling_vcr.py:115-129) trains every parameter it holds."""
    return () if head == "nsp" else NO_GRAD_PREFIXES


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
        self.gen = 0            # stamps every training forward: a backward may only consume ITS forward's activations
        self.drop_seed = None   # Philox key of the dropout masks (default: torch.initial_seed() mixed with the rank)
        self.drop_step = 0      # counter word: one fresh mask set per training forward
        self.sync = None        # dist.ShardedGradSync when an optimizer runs this engine data-parallel
        self.defer = False      # data parallel: backward leaves the reduce-scatter to optimizer.step() (FusedAdamW.no_sync / defer_reduce)
        self.sent_version = None   # version counter of `grad` when its reduce-scatter started: a later in-place edit cannot reach the update

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
            if eng.head == "nsp":
                g.w_pool = gp("bert.pooler.dense.weight")
                g.b_pool = gp("bert.pooler.dense.bias")
                g.w_rel = gp("cls.weight")
                g.b_rel = gp("cls.bias")
            else:
                hp = "cls." if eng.head == "cpt" else "cls.predictions."
                g.w_tr = gp(hp + "transform.dense.weight")
                g.b_tr = gp(hp + "transform.dense.bias")
                g.tr_ln_g = gp(hp + "transform.LayerNorm.weight")
                g.tr_ln_b = gp(hp + "transform.LayerNorm.bias")
                g.b_dec = gp(hp + "bias")
            self.gdesc = (g, layers)

    def workspace(self, B, Lt, Li, n_rows=0):
        eng = self.eng
        m, _ = eng.descriptor(train=True)
        need = L.lib().cpt_train_workspace_bytes_rows(C.byref(m.dims), B, Lt, Li, n_rows)
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
    skip = no_grad_prefixes(eng.head)
    for n, p in named.items():
        if skip and n.startswith(skip):
            out.append((p, None))
        else:
            off, num = eng.offsets[n]
            out.append((p, st.grad[off:off + num].view(p.shape)))
    return out


class _MLMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, eng, tensors, drop):
        ctx.eng = eng
        st = _state(eng)
        ids, seg, mask, pos, feats, mpos, labels, rseq = tensors
        B, Lt = ids.shape
        R = int(rseq.numel()) if rseq is not None else 0        # label grid: R labelled rows (else one per sequence)
        Li = feats.size(1) if feats is not None else 0
        m, _ = eng.descriptor(train=True)
        dev = eng.flat.device
        loss_acc = torch.empty(3, device=dev, dtype=torch.float32)      # {sum, count, mean}: the mean is written by the cross-entropy launch (cpt_outputs.loss_mean)
        if eng.head == "nsp":       # relation scores of the pooled [CLS] instead of vocabulary logits of the [MASK] row
            logits = torch.empty((B, m.dims.n_rel), device=dev, dtype=torch.float32)
            o = L.Outputs(rel=logits.data_ptr(), loss=loss_acc.data_ptr(), loss_mean=loss_acc.data_ptr() + 8)
        else:
            logits = torch.empty((R if R else B, eng.cfg.vocab_size), device=dev, dtype=torch.float32)
            o = L.Outputs(logits=logits.data_ptr(), loss=loss_acc.data_ptr(), loss_mean=loss_acc.data_ptr() + 8)
        bt = L.Batch(B=B, Lt=Lt, Li=Li, input_ids=ids.data_ptr(), token_type=L.ptr(seg), position_ids=L.ptr(pos),
                     attn_mask=L.ptr(mask), img_feats=L.ptr(feats), mask_pos=L.ptr(mpos), labels=labels.data_ptr(),
                     n_rows=R, row_seq=L.ptr(rseq), mask_3d=1 if (mask is not None and mask.dim() == 3) else 0)
        if st.saved is not None and st.ws is not None:
            # the activations of an earlier training forward are still waiting for their backward; the workspace is
            # per engine, so this forward overwrites them (their backward will raise instead of using the wrong ones)
            pass
        ws = st.workspace(B, Lt, Li, R)
        errs = []

        def before_bucket(_user, k):
            # data parallel: bucket k's parameter all-gather (queued by the last optimizer step) must have landed,
            # and its bf16 shadow / padded copies must be current, before the first kernel that reads it
            try:
                eng.complete_pending(k)
            except Exception as e:          # never let an exception cross the C frame
                errs.append(e)
        cb = L.BUCKET_CB(before_bucket) if eng.pending is not None else L.NULL_CB
        L.check(L.lib().cpt_train_fwd_ex(C.byref(m), C.byref(bt), C.byref(o), ws.data_ptr(), ws.numel(), L.stream_ptr(),
                                         cb, None, C.byref(drop) if drop is not None else None), "cpt_train_fwd")
        if errs:
            raise errs[0]
        st.gen += 1
        ctx.gen = st.gen
        st.saved = (bt, tensors, st.gen, drop)    # keep the input tensors alive until backward; same masks in backward
        ctx.logits = logits
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)      # (no zero-filled (R, V) gradient tensor for the scores on every backward)
        return loss_acc[2], logits

    @staticmethod
    def backward(ctx, grad_loss, _grad_logits):
        eng = ctx.eng
        st = _state(eng)
        if st.saved is None:
            raise RuntimeError("cpt_amd: backward called twice (activations of the training forward were released)")
        bt, tensors, gen, drop = st.saved
        if gen != ctx.gen:
            raise RuntimeError("cpt_amd: backward of a stale training forward: the engine keeps the activations of the LATEST "
                               "training forward only (one workspace per model); call backward() before the next "
                               "training forward of the same model")
        st.ensure()
        m, _ = eng.descriptor(train=True)
        views = _named_grad_views(eng, st)
        accumulate = any(p.grad is not None for p, v in views if v is not None)
        if st.sync is not None:
            st.sync.begin_backward()      # reduce-scatters of an unconsumed earlier backward may still read st.grad: wait BEFORE touching it
        if accumulate:
            keep = st.grad.clone()
        g, _ = st.gdesc
        # only what the backward ADDS into (bias / LayerNorm vectors, small tables) is cleared; weight gradients are written whole
        L.check(L.lib().cpt_train_zero_grads(C.byref(m), C.byref(g), int(bt.Li), L.stream_ptr()), "cpt_train_zero_grads")
        # the early reduce-scatter only when this backward completes the gradient: not while accumulating, not under no_sync()
        sync = st.sync if not (accumulate or st.defer) else None
        st.sent_version = None
        errs = []

        def grads_ready(_user, k):
            try:
                sync.grads_ready(st.grad, k)
            except Exception as e:
                errs.append(e)
        cb = L.BUCKET_CB(grads_ready) if sync is not None else L.NULL_CB     # (None: buckets are reduced in optimizer.step())
        gl = grad_loss.to(torch.float32).contiguous()          # device scalar: no host synchronisation between fwd and bwd
        L.check(L.lib().cpt_train_bwd_ex(C.byref(m), C.byref(bt), C.byref(g), 1.0, gl.data_ptr(), st.ws.data_ptr(), st.ws.numel(),
                                         L.stream_ptr(), cb, None, C.byref(drop) if drop is not None else None), "cpt_train_bwd")
        if errs:
            raise errs[0]
        if accumulate:
            st.grad.add_(keep)
            if st.sync is not None and not st.defer:
                # the closing micro-step of an accumulation window: the sum is final now -- start every bucket's reduce-scatter here
                # (it then runs under whatever the caller does before step(); nothing of this backward was left to hide it under)
                sync = st.sync
                for k in sync.order:
                    sync.grads_ready(st.grad, k)
        for p, v in views:
            p.grad = v
        if sync is not None:
            # The p.grad views alias the buffer whose reduce-scatter is already in flight: an in-place edit now (e.g.
            # torch.nn.utils.clip_grad_norm_, fewshot/vcr_nsp_cpt.py:461) would race with it and could never reach the update.
            # Views share their base's version counter, so step() can tell and refuse (FusedAdamW.step).
            st.sent_version = st.grad._version
        st.saved = None
        return None, None, None, None


def mlm_loss_with_grad(model, input_ids, token_type_ids, attention_mask, labels, position_ids, img_feats, mask_token_pos, row_seq=None):
    """(loss, prediction_scores) with autograd history, as REC_MLM_CPT.forward returns them
    (modeling_rec.py:147-152).  ``mask_token_pos`` is required: the loss only sees the [MASK] rows
    (fewshot/refcoco_cpt.py:231-233 puts -1 everywhere else).  For NSPCPT (head "nsp", modeling_vcr.py:115-129)
    ``labels`` are the (B,) next-sentence labels (-1 ignored), ``mask_token_pos`` is None and the second result is
    the (B, num_contrast_classes) relation scores.
    ``row_seq`` (label grids with any number of labelled positions per sequence): ``labels`` / ``mask_token_pos`` then list the R
    labelled positions, ``row_seq`` (R,) the sequence each belongs to; the second result is the (R, V) scores of those rows."""
    eng = model._engine()
    if mask_token_pos is None and eng.head != "nsp":
        raise NotImplementedError("cpt_amd: training needs mask_token_pos (the (B, L) label grid of the reference has "
                                  "exactly one labelled position per row: pass it as mask_token_pos)")
    if eng.dtype not in ("fp32", "bf16", "bf16x3"):
        raise NotImplementedError("cpt_amd: training runs in 'fp32', 'bf16x3' or 'bf16' compute mode (not '%s')" % eng.dtype)
    eng.ensure_packed()
    if eng.pending is None:          # (data parallel: a pending parameter all-gather is awaited bucket by bucket in the forward)
        eng.refresh_shadow(train=True)
    st = _state(eng)
    st.ensure()

    def prep(t, dt, name):
        if t is None:
            return None
        if not t.is_cuda:
            raise RuntimeError("cpt_amd: %s must be on the GPU" % name)
        return t.to(dt).contiguous()

    if attention_mask is not None and attention_mask.dim() not in (2, 3):
        raise NotImplementedError("cpt_amd: attention_mask must be (B, L) or (B, L, L) (modeling_bert.py:213-218)")
    tensors = (prep(input_ids, torch.int64, "input_ids"), prep(token_type_ids, torch.int64, "token_type_ids"),
               prep(attention_mask, torch.int64, "attention_mask"), prep(position_ids, torch.int64, "position_ids"),
               prep(img_feats, torch.float32, "img_feats"), prep(mask_token_pos, torch.int64, "mask_token_pos"),
               prep(labels, torch.int64, "labels"), prep(row_seq, torch.int64, "row_seq"))
    trigger = getattr(st, "trigger", None)      # one leaf per engine, made once: autograd needs an input that requires grad, backward returns None for it
    if trigger is None or trigger.device != eng.flat.device:
        trigger = st.trigger = torch.zeros((), device=eng.flat.device, requires_grad=True)
    loss, logits = _MLMLoss.apply(trigger, eng, tensors, dropout_for(model, st))
    return (loss, logits)


def dropout_for(model, st):
    """The cpt_dropout of this training forward: config.hidden_dropout_prob / attention_probs_dropout_prob when the
    module is in training mode (nn.Dropout semantics: identity in eval), a key derived from torch's seed and the rank
    (so `torch.manual_seed(args.seed)` of the reference drivers, zeroshot/refcoco_cpt.py:376, controls it and ranks draw
    different masks), and a counter that advances with every training forward.  None = no dropout."""
    cfg = model.config
    ph = float(getattr(cfg, "hidden_dropout_prob", 0.0) or 0.0) if model.training else 0.0
    pa = float(getattr(cfg, "attention_probs_dropout_prob", 0.0) or 0.0) if model.training else 0.0
    if ph <= 0.0 and pa <= 0.0:
        return None
    if st.drop_seed is None:
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        st.drop_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    st.drop_step += 1
    return L.Dropout(p_hidden=ph, p_attn=pa, seed=st.drop_seed, step=st.drop_step)


def set_dropout_seed(model, seed, step=0):
    """Pin the dropout stream of `model` (tests, reproducible runs): masks become a function of (seed, forward count)."""
    st = _state(model._engine())
    st.drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    st.drop_step = int(step)


class FusedAdamW(object):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt) as ONE kernel over the
    flat parameter / gradient / moment buffers; also refreshes the bf16 shadow.  Exposes ``param_groups`` with the
    reference's four groups so ``param_group['lr'] = ...`` scheduling code (fewshot/refcoco_cpt.py:237-243) keeps
    working.

    Data parallel (torch.distributed initialised with world size > 1; the reference's DistributedDataParallel wrap at
    fewshot/refcoco_cpt.py:516-522 is replaced by this optimizer, do NOT also wrap the model in DDP):
      * construction broadcasts rank 0's flat parameters (what DDP does at wrap time);
      * backward reduce-scatters each parameter bucket as soon as its gradients are complete, on a side stream,
        under the rest of backward (dist.ShardedGradSync);
      * step() runs AdamW on this rank's 1/N shard of every bucket (moments are sharded: 1/N of the state per GPU)
        and queues the parameter all-gather, which the next forward waits for bucket by bucket.
      ``p.grad`` views hold this rank's LOCAL gradients (the averaged ones exist only as shards) and are READ-ONLY once
      backward has returned: their reduce-scatter is already in flight, so step() raises if they were edited in place.  Use
      ``clip_grad_norm_`` below instead of torch.nn.utils.clip_grad_norm_ (fewshot/vcr_nsp_cpt.py:461), or
      ``defer_reduce=True`` (the reduce-scatter runs inside step(), local gradients stay editable, no overlap with backward);
      wrap the leading micro-steps of a gradient-accumulation window in ``no_sync()``.
      ``state_dict()`` / ``load_state_dict()`` / ``train.save_checkpoint`` are COLLECTIVE in this mode (the sharded moments are
      gathered): every rank must call them, unlike the model's own ``state_dict()``.
    grad_wire: None (fp32 on the wire) or "bf16" (gradients cast to bf16 for the reduce-scatter: half the bytes).
    force_collectives: run the collectives even at world size 1 (exercises the RCCL stream choreography on one GPU)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_wire=None, defer_reduce=False,
                 force_collectives=False, hf_arithmetic=False, correct_bias=True):
        self.model = model
        # hf_arithmetic: the update of pytorch_transformers.AdamW (the GQA / VCR few-shot drivers' optimizer) instead of torch.optim.AdamW's --
        # eps outside the bias correction, decay applied to the updated parameter (include/cpt_hip.h cpt_adamw_ex); see the AdamW class below
        self.flags = (L.ADAMW_HF if hf_arithmetic else 0) | (0 if correct_bias else L.ADAMW_NO_BIAS_CORRECTION)
        if not correct_bias and not hf_arithmetic:
            raise ValueError("correct_bias=False belongs to the pytorch_transformers arithmetic (hf_arithmetic=True)")
        self.defer_reduce = bool(defer_reduce)
        self.force_collectives = bool(force_collectives)
        self.eng = model._engine()
        self.betas = betas
        self.eps = eps
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "params": []},
                             {"lr": lr, "weight_decay": 0.0, "params": []},
                             {"lr": lr, "weight_decay": weight_decay, "params": []},
                             {"lr": lr, "weight_decay": 0.0, "params": []}]
        self.step_count = 0
        self.m = self.v = self.code = None
        self.grad_wire = {None: None, "fp32": None, "bf16": torch.bfloat16}[grad_wire]
        self.sync = None
        self._clip = 1.0
        self._stale_shadow = set()
        import torch.distributed as dist
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.dp = self.world > 1 or self.force_collectives
        if self.dp and next(model.parameters()).is_cuda:
            self._ensure()

    def no_sync(self):
        """Context manager (DistributedDataParallel.no_sync): backward passes inside it do not start the gradient
        reduce-scatter -- for the leading micro-steps of a gradient-accumulation window, whose partial sums would be sent for
        nothing.  The backward that closes the window (outside the context) adds its gradients to the kept partial sum and THEN starts
        the buckets' reduce-scatters (they cannot overlap that backward: the sum is only final behind it); with ``defer_reduce`` they
        run inside step()."""
        opt = self

        class _NoSync(object):
            def __enter__(self_inner):
                st = _state(opt.eng)
                self_inner.prev = st.defer
                st.defer = True

            def __exit__(self_inner, *exc):
                _state(opt.eng).defer = self_inner.prev or opt.defer_reduce
                return False
        return _NoSync()

    def _ensure(self):
        eng = self.eng
        eng.ensure_packed()
        dev = eng.flat.device
        if self.dp and (self.sync is None or self.sync.device != dev):
            from . import dist as cdist
            self.sync = cdist.ShardedGradSync(eng.buckets, dev, wire=self.grad_wire, force_collectives=self.force_collectives)
            cdist.broadcast_(eng.flat, 0)             # replicas start from rank 0's parameters
            eng.weights_updated()
            _state(eng).sync = self.sync
            _state(eng).defer = self.defer_reduce
            self.m = None
        n_state = self.sync.shard_elems if self.sync is not None else eng.flat.numel()
        if self.m is None or self.m.numel() != n_state or self.m.device != dev:
            self.m = torch.zeros(n_state, device=dev, dtype=torch.float32)
            self.v = torch.zeros(n_state, device=dev, dtype=torch.float32)
            code = torch.zeros(eng.flat.numel(), dtype=torch.uint8)
            for n in eng.offsets:
                off, num = eng.offsets[n]
                if no_grad_prefixes(eng.head) and n.startswith(no_grad_prefixes(eng.head)):
                    c = 0
                elif any(nd in n for nd in NO_DECAY):
                    c = 2
                else:
                    c = 1
                code[off:off + num] = c
            code = code.to(dev)
            if self.sync is not None:
                sh = torch.zeros(n_state, device=dev, dtype=torch.uint8)
                self.sync.scatter_full(code, sh)
                code = sh
            self.code = code

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping of the AVERAGED gradient (gqa_cpt.py:454 clips at 1.0; the RefCOCO path does not clip).
        Call between backward() and step(); the coefficient is folded into the update.  Returns the norm (a host
        float: this synchronises)."""
        self._ensure()
        st = _state(self.eng)
        if self.sync is not None:
            import torch.distributed as dist
            self.sync.finish_reduce(st.grad)
            sq = (self.sync.gshard.double() ** 2).sum() / float(self.world) ** 2
            dist.all_reduce(sq)
        else:
            sq = (st.grad.double() ** 2).sum()
        norm = float(sq.sqrt())
        self._clip = min(1.0, max_norm / (norm + 1e-6))
        return norm

    def _adamw(self, p_ptr, g_ptr, m_ptr, v_ptr, code_ptr, shadow_ptr, n, lr, wd, scale):
        if self.flags:
            L.check(L.lib().cpt_adamw_ex(p_ptr, g_ptr, m_ptr, v_ptr, code_ptr, shadow_ptr, n, lr, self.betas[0], self.betas[1], self.eps,
                                         wd, self.step_count, scale, self.flags, L.stream_ptr()), "cpt_adamw_ex")
            return
        L.check(L.lib().cpt_adamw(p_ptr, g_ptr, m_ptr, v_ptr, code_ptr, shadow_ptr, n, lr, self.betas[0], self.betas[1], self.eps,
                                  wd, self.step_count, scale, L.stream_ptr()), "cpt_adamw")

    def step(self):
        eng = self.eng
        self._ensure()
        st = _state(eng)
        if st.grad is None:
            raise RuntimeError("cpt_amd: optimizer.step() before any backward")
        self.step_count += 1
        lr = self.param_groups[2]["lr"]
        wd = self.param_groups[2]["weight_decay"]
        lp = eng.dtype == "bf16" and eng.flat_lp is not None
        if self.sync is None:
            shadow = eng.flat_lp.data_ptr() if lp else None
            self._adamw(eng.flat.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.code.data_ptr(), shadow,
                        eng.flat.numel(), lr, wd, self._clip)
            self._clip = 1.0
            eng.weights_updated(shadow_fresh=shadow is not None)
            return
        # data parallel: summed gradient shards -> AdamW on the shard (mean folded in as 1/world) -> all-gather
        sync = self.sync
        if eng.pending is not None:
            eng.complete_pending()
        if st.sent_version is not None and st.grad._version != st.sent_version:
            raise RuntimeError(
                "cpt_amd: a parameter's .grad was modified in place after backward() had already started the data-parallel "
                "reduce-scatter of the gradients (the edit cannot reach the update and races with the transfer).  For global-norm "
                "clipping call optimizer.clip_grad_norm_(max_norm) instead of torch.nn.utils.clip_grad_norm_; to edit local "
                "gradients build the optimizer with defer_reduce=True (the reduce-scatter then runs inside step()).")
        sync.finish_reduce(st.grad)
        for k in sync.order:
            slo, shi = sync.shard_range(k)
            so = sync.soff[k]
            self._adamw(eng.flat.data_ptr() + slo * 4, sync.gshard.data_ptr() + so * 4, self.m.data_ptr() + so * 4,
                        self.v.data_ptr() + so * 4, self.code.data_ptr() + so, None, shi - slo, lr, wd, self._clip / self.world)
        self._clip = 1.0
        st.sent_version = None            # consumed: a later in-place edit of the (now stale) gradient views is the caller's business
        sync.all_gather_params(eng.flat)
        self._stale_shadow = set(sync.order)
        eng.pending = self

    def _params_arrived(self, bucket=None):
        """engine.complete_pending: bucket's all-gather has landed on the current stream -> refresh its derived copies."""
        ks = list(self.sync.order) if bucket is None else [bucket]
        for k in ks:
            if k in self._stale_shadow:
                self.sync.wait_params(k)
                self.eng.refresh_bucket_shadow(k)
                self._stale_shadow.discard(k)
        if not self._stale_shadow:
            self.eng.pending = None
            self.eng._sig = {self.eng.dtype: self.eng._versions()}

    # ---- checkpointing: the reference saves no optimizer state (utils/save_model.py), so a few-shot
    # run cannot resume; here the moments travel with the model, keyed by parameter NAME ------------------------------
    def _full_moments(self):
        if self.sync is None:
            return self.m, self.v
        n = self.eng.flat.numel()
        return self.sync.gather_full(self.m, n), self.sync.gather_full(self.v, n)

    def state_dict(self):
        """{"state": {name: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...], "betas", "eps"}: the same
        per-parameter entries as torch.optim.AdamW.state_dict(), as CPU tensors (data parallel: a collective call,
        the sharded moments are gathered)."""
        self._ensure()
        m, v = self._full_moments()
        state = {}
        for n, (off, num) in self.eng.offsets.items():
            shape = tuple(self.eng._named()[n].shape) if hasattr(self.eng, "_named") else (num,)
            state[n] = {"step": self.step_count,
                        "exp_avg": m[off:off + num].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": v[off:off + num].view(shape).detach().cpu().clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"state": state, "param_groups": groups, "betas": tuple(self.betas), "eps": self.eps,
                "step_count": self.step_count}

    def load_state_dict(self, sd):
        self._ensure()
        missing = [n for n in self.eng.offsets if n not in sd["state"]]
        if missing:
            raise RuntimeError("cpt_amd: optimizer state lacks %d parameters, e.g. %s" % (len(missing), missing[0]))
        n_all = self.eng.flat.numel()
        dev = self.eng.flat.device
        m = torch.zeros(n_all, device=dev) if self.sync is not None else self.m
        v = torch.zeros(n_all, device=dev) if self.sync is not None else self.v
        for n, (off, num) in self.eng.offsets.items():
            e = sd["state"][n]
            if e["exp_avg"].numel() != num:
                raise RuntimeError("cpt_amd: optimizer state of %s has %d elements, expected %d" % (n, e["exp_avg"].numel(), num))
            m[off:off + num].copy_(e["exp_avg"].reshape(-1))
            v[off:off + num].copy_(e["exp_avg_sq"].reshape(-1))
        if self.sync is not None:
            self.sync.scatter_full(m, self.m)
            self.sync.scatter_full(v, self.v)
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
    import torch.distributed as dist
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    # Data parallel: COLLECTIVE -- every rank calls this and rank 0 writes, which is how the reference's save_model is called
    # too (fewshot/refcoco_cpt.py:548 on every rank, the rank test inside utils/save_model.py:6); the optimizer's sharded
    # moments are gathered by optimizer.state_dict(), so guarding THIS call with `if rank == 0` would hang the other ranks.
    opt_sd = optimizer.state_dict() if optimizer is not None else None
    if rank != 0:
        return
    os.makedirs(save_dir, exist_ok=True)
    model.save_pretrained(save_dir)
    if opt_sd is not None:
        torch.save(opt_sd, os.path.join(save_dir, "optimizer.pt"))
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


class AdamW(FusedAdamW):
    """Drop-in for ``pytorch_transformers.AdamW`` as the GQA / VCR few-shot drivers build it (fewshot/vcr_nsp_cpt.py:380-385, gqa_cpt.py:337-342:
    ``AdamW(optimizer_grouped_parameters, lr=args.learning_rate, eps=args.adam_epsilon)`` with weight decay on everything but ``bias`` /
    ``LayerNorm.weight``): same defaults (betas (0.9, 0.999), eps 1e-6, weight_decay 0.0, correct_bias True) and the SAME update arithmetic
    (``cpt_adamw_ex(CPT_ADAMW_HF)``), one launch over the flat buffers.  Takes the MODEL where the reference passes its two parameter groups --
    the no-decay rule of those groups is the one ``FusedAdamW`` applies -- and exposes two ``param_groups`` (decay, no decay) for
    ``WarmupLinearSchedule`` / ``WarmupConstantSchedule`` below."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, **kw):
        FusedAdamW.__init__(self, model, lr, betas=betas, eps=eps, weight_decay=weight_decay, hf_arithmetic=True, correct_bias=correct_bias, **kw)
        # FusedAdamW reads lr / weight_decay from param_groups[2]: keep its four-slot list, show the two groups the reference has
        self.param_groups = [self.param_groups[2], self.param_groups[3]]

    def step(self):
        g = self.param_groups
        self.param_groups = [g[0], g[1], g[0], g[1]]          # (FusedAdamW.step reads slot 2: the decay group)
        try:
            FusedAdamW.step(self)
        finally:
            self.param_groups = g


class _WarmupSchedule(object):
    """torch.optim.lr_scheduler.LambdaLR's behaviour on an optimizer that only has ``param_groups`` (FusedAdamW / AdamW): construction sets
    lr = base_lr x lambda(0), every ``step()`` moves to the next epoch.  ``step(epoch)`` is accepted as pytorch_transformers 1.x callers pass it."""

    def __init__(self, optimizer, last_epoch=-1):
        self.optimizer = optimizer
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def lr_lambda(self, step):
        raise NotImplementedError

    def get_lr(self):
        return [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr


class WarmupLinearSchedule(_WarmupSchedule):
    """``pytorch_transformers.WarmupLinearSchedule`` (fewshot/vcr_nsp_cpt.py:386, gqa_cpt.py:348): the multiplier rises linearly from 0 to 1 over
    ``warmup_steps`` and falls linearly to 0 at ``t_total`` (transformers.get_linear_schedule_with_warmup is today's name of the same lambda:
    tests/test_host_cpu.py checks them against each other)."""

    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps, self.t_total = warmup_steps, t_total
        _WarmupSchedule.__init__(self, optimizer, last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))


class WarmupConstantSchedule(_WarmupSchedule):
    """``pytorch_transformers.WarmupConstantSchedule`` (gqa_cpt.py:346): linear warm-up, then 1."""

    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        self.warmup_steps = warmup_steps
        _WarmupSchedule.__init__(self, optimizer, last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1.0, self.warmup_steps))
        return 1.0


def build_optimizer(model, opts):
    """fewshot/refcoco_cpt.py:318-343 (opts.learning_rate, opts.weight_decay, opts.betas)."""
    return FusedAdamW(model, lr=opts.learning_rate, betas=tuple(opts.betas), weight_decay=opts.weight_decay)
