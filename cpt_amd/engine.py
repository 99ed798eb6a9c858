"""Host-side engine: packs a model's parameters into flat device buffers, builds the
``cpt_model`` descriptor of include/cpt_hip.h and launches the HIP forward.

MI355X-first layout: every fp32 parameter lives in ONE contiguous buffer (the nn.Parameters of
the module tree are views into it, so ``state_dict`` stays the source of truth), with
query/key/value weights adjacent so the three projections run as one N=3H GEMM.  The same
layout is mirrored in a bf16 shadow for the MFMA operands, in a flat gradient buffer (one RCCL
all-reduce covers all gradients) and in flat AdamW moments (one fused optimizer kernel).
"""
import ctypes as C

import torch

from . import _lib as L
from .synth import param_specs

ALIGN = 64  # elements


BUCKET_ALIGN = 512  # elements: every parameter bucket splits evenly over 1, 2, 4 or 8 ranks into 64-element-aligned shards


def bucket_of(name, n_layers):
    """Data-parallel bucket of a parameter (include/cpt_hip.h, cpt_train_bwd_ex): 0 = embedding tables, their
    LayerNorm and the region projection; 1 + l = encoder layer l; n_layers + 1 = pooler + task head."""
    if name.startswith("bert.encoder.layer."):
        return 1 + int(name.split(".")[3])
    if name.startswith(("bert.pooler.", "cls.")):
        return n_layers + 1
    return 0


def pack_order(cfg, head):
    """Parameter names in flat-buffer order: grouped by data-parallel bucket (each bucket is one contiguous range, so
    its gradients reduce-scatter as ONE message), inside a layer q/k/v weights adjacent, then q/k/v biases adjacent
    (the fused QKV GEMM reads them as [3H][H] and [3H])."""
    names = [n for n, _, kind in param_specs(cfg, head) if kind != "tied"]
    final = []
    for n in names:
        if ".attention.self." in n:
            continue
        if n.endswith("attention.output.dense.weight"):
            p = n[: -len("output.dense.weight")] + "self."
            final += [p + "query.weight", p + "key.weight", p + "value.weight",
                      p + "query.bias", p + "key.bias", p + "value.bias"]
        final.append(n)
    nl = cfg.num_hidden_layers
    return sorted(final, key=lambda n: bucket_of(n, nl))     # stable: registration order inside a bucket


def bucket_table(cfg, head, numel=None):
    """(offsets, buckets, total) of the flat parameter / gradient / moment buffers WITHOUT allocating them: offsets[name] = (first
    element, elements), buckets[k] = [lo, hi) of data-parallel bucket k (every bucket starts on a BUCKET_ALIGN boundary: it splits evenly
    over 1, 2, 4 or 8 ranks).  numel(name) -> elements (default: from synth.param_specs).  The layout PackedModel.ensure_packed builds."""
    if numel is None:
        shapes = {n: shape for n, shape, kind in param_specs(cfg, head) if kind != "tied"}
        numel = lambda n: int(torch.Size(shapes[n]).numel())
    total, offsets, starts = 0, {}, {}
    nl = cfg.num_hidden_layers
    for n in pack_order(cfg, head):
        k = bucket_of(n, nl)
        if k not in starts:                       # a new bucket starts on a BUCKET_ALIGN boundary
            total = (total + BUCKET_ALIGN - 1) // BUCKET_ALIGN * BUCKET_ALIGN
            starts[k] = total
        offsets[n] = (total, numel(n))
        total += (numel(n) + ALIGN - 1) // ALIGN * ALIGN
    total = (total + BUCKET_ALIGN - 1) // BUCKET_ALIGN * BUCKET_ALIGN
    ks = sorted(starts)
    buckets = {k: (starts[k], starts[ks[i + 1]] if i + 1 < len(ks) else total) for i, k in enumerate(ks)}
    return offsets, buckets, total


class PackedModel(object):
    """Flat parameter storage + C descriptor for one nn.Module (BertImgModel or a head wrapper)."""

    def __init__(self, module, cfg, head, prefix_map=None):
        self.module = module
        self.cfg = cfg
        self.head = head            # 'cpt' | 'pretrain' | 'none'
        self.flat = None
        self.flat_lp = None
        self.grad = None
        self.offsets = {}
        self._ptrs = None
        self._sig = None
        self._desc = {}
        self._ws = {}
        self.dtype = "fp32"
        self.fold_ln = True         # bf16 inference: fold the encoder LayerNorms into the GEMMs around them
        self._fold = None
        self._fold_stale = True
        self.pending = None         # data-parallel optimizer whose parameter all-gather is still in flight (train.FusedAdamW)

    # ---- packing -------------------------------------------------------------------------
    def _named(self):
        named = dict(self.module.named_parameters())
        pre = "" if self.head != "none" else "bert."
        out = {}
        for n in pack_order(self.cfg, self.head):
            key = n[len(pre):] if pre and n.startswith(pre) else n
            if key not in named:
                raise RuntimeError("cpt_amd: parameter %s missing from module" % key)
            out[n] = named[key]
        return out

    def _packed_ok(self):
        # cheap per-call check: Module.to()/load paths replace every .data or none
        if self.flat is None:
            return False
        first, last, last_off = self._probe
        base = self.flat.data_ptr()
        return first.data_ptr() == base and last.data_ptr() == base + last_off * 4

    def ensure_packed(self):
        if self._packed_ok():
            return
        named = self._named()
        dev = next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError("cpt_amd: model parameters are on %s; move the model to the GPU "
                               "(the HIP path has no CPU fallback)" % dev)
        for n, p in named.items():
            if p.dtype != torch.float32:
                raise RuntimeError("cpt_amd: parameter %s is %s; master weights must be fp32" % (n, p.dtype))
        # [lo, hi) element range of bucket k in the flat parameter / gradient / moment buffers (gaps hold zeros)
        self.offsets, self.buckets, total = bucket_table(self.cfg, self.head, numel=lambda n: named[n].numel())
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for n, p in named.items():
            off, num = self.offsets[n]
            flat[off:off + num].copy_(p.data.reshape(-1))
            p.data = flat[off:off + num].view(p.shape)
        self.flat = flat
        self._plist = list(named.values())
        order = list(named)
        self._probe = (named[order[0]], named[order[-1]], self.offsets[order[-1]][0])
        self.flat_lp = None
        self.grad = None
        self._sig = None
        self._desc = {}

    def view(self, name, flat=None):
        off, num = self.offsets[name]
        return (self.flat if flat is None else flat)[off:off + num]

    def _versions(self):
        return sum(p._version for p in self._plist)

    def refresh_shadow(self, force=False, train=False):
        """bf16 copy of the flat buffer + zero-padded img weight; redone whenever a parameter
        was written in place (load_state_dict, an optimizer step done outside this engine).
        train (bf16x3 mode): the training step splits its operands itself, so the standing split copies of the weights
        are only marked stale here and rebuilt by the next inference forward."""
        if self.pending is not None:
            self.complete_pending()
        sig = self._versions()
        if self._sig is None:
            self._sig = {}
        if not force and self._sig.get(self.dtype) == sig:
            if self.dtype == "bf16x3" and not train and getattr(self, "_x3_stale", False):
                self._build_x3(L.stream_ptr())
                self._x3_stale = False
            return
        self._sig[self.dtype] = sig
        self._fold_stale = True
        cfg = self.cfg
        H, D = cfg.hidden_size, cfg.img_feature_dim
        Dp = (D + 63) // 64 * 64
        st = L.stream_ptr()
        w_img = self.view("bert.img_embedding.weight")
        n = self.flat.numel()
        if self.dtype == "bf16":
            if self.flat_lp is None or self.flat_lp.device != self.flat.device:
                self.flat_lp = torch.empty(n, device=self.flat.device, dtype=torch.bfloat16)
                self.img_pad_lp = torch.empty((H, Dp), device=self.flat.device, dtype=torch.bfloat16)
                self._desc = {}
            L.check(L.lib().cpt_pad_cast(self.flat.data_ptr(), self.flat_lp.data_ptr(), L.CPT_BF16, 1, n, n, st),
                    "cpt_pad_cast(shadow)")
            L.check(L.lib().cpt_pad_cast(w_img.data_ptr(), self.img_pad_lp.data_ptr(), L.CPT_BF16, H, D, Dp, st),
                    "cpt_pad_cast(w_img)")
        else:
            if getattr(self, "img_pad_f32", None) is None or self.img_pad_f32.device != self.flat.device:
                self.img_pad_f32 = torch.empty((H, Dp), device=self.flat.device, dtype=torch.float32)
                self._desc = {}
            L.check(L.lib().cpt_pad_cast(w_img.data_ptr(), self.img_pad_f32.data_ptr(), L.CPT_F32, H, D, Dp, st),
                    "cpt_pad_cast(w_img)")
            if self.dtype == "bf16x3":
                if train:
                    self._x3_stale = True
                else:
                    self._build_x3(st)
                    self._x3_stale = False

    def _x3_matrices(self):
        """(key, first parameter name, rows, cols) of every matrix the forward multiplies by (Q|K|V stacked as one)."""
        cfg = self.cfg
        H, I = cfg.hidden_size, cfg.intermediate_size
        out = []
        for i in range(cfg.num_hidden_layers):
            p = "bert.encoder.layer.%d." % i
            out += [(p + "qkv", p + "attention.self.query.weight", 3 * H, H), (p + "ao", p + "attention.output.dense.weight", H, H),
                    (p + "in", p + "intermediate.dense.weight", I, H), (p + "out", p + "output.dense.weight", H, I)]
        out.append(("pool", "bert.pooler.dense.weight", H, H))
        if self.head == "nsp":
            out.append(("rel", "cls.weight", getattr(cfg, "num_contrast_classes", 2), H))
        if self.head in ("cpt", "pretrain"):
            hp = "cls." if self.head == "cpt" else "cls.predictions."
            out.append(("tr", hp + "transform.dense.weight", H, H))
            out.append(("dec", "bert.embeddings.word_embeddings.weight", cfg.vocab_size, H))
        if self.head == "pretrain":
            out.append(("rel", "cls.seq_relationship.weight", getattr(cfg, "num_contrast_classes", 2), H))
        return out

    def _build_x3(self, st):
        """bf16x3 parity mode: [N][hi | lo | hi] bf16 split copies of the weight matrices (include/cpt_hip.h cpt_split3)."""
        cfg = self.cfg
        H, D = cfg.hidden_size, cfg.img_feature_dim
        Dp = (D + 63) // 64 * 64
        mats = self._x3_matrices()
        total = sum(n * 3 * k for _, _, n, k in mats) + H * 3 * Dp
        dev = self.flat.device
        if getattr(self, "x3_buf", None) is None or self.x3_buf.device != dev or self.x3_buf.numel() != total:
            self.x3_buf = torch.empty(total, device=dev, dtype=torch.bfloat16)
            self._desc = {}
        self.x3_off = {}
        off = 0
        base = self.x3_buf.data_ptr()
        for key, pname, n, k in mats:
            src = self.flat.data_ptr() + self.offsets[pname][0] * 4
            L.check(L.lib().cpt_split3(src, k, base + off * 2, n, k, 1, st), "cpt_split3(%s)" % key)
            self.x3_off[key] = off
            off += n * 3 * k
        L.check(L.lib().cpt_split3(self.img_pad_f32.data_ptr(), Dp, base + off * 2, H, Dp, 1, st), "cpt_split3(w_img)")
        self.x3_off["img"] = off

    def invalidate(self):
        """Tell the engine the parameters were written behind its back.  In-place writes through ``.data``
        (``p.data.copy_/mul_/normal_``, which init_weights-style code uses) do not bump the Parameter's version
        counter, so the bf16 shadow, the LayerNorm-folded weights and the padded img_embedding copy would go stale
        silently; load_state_dict / optimizer paths are detected automatically, anything else calls this."""
        self._sig = None
        self._fold_stale = True

    def complete_pending(self, bucket=None):
        """Data parallel: make the current stream wait for the parameter all-gather the last optimizer step queued
        (bucket None: all of it) and refresh the copies derived from those parameters."""
        opt = self.pending
        if opt is None:
            return
        opt._params_arrived(bucket)

    def refresh_bucket_shadow(self, k):
        """bf16 shadow (and the padded img_embedding copy for bucket 0) of ONE parameter bucket."""
        lo, hi = self.buckets[k]
        st = L.stream_ptr()
        cfg = self.cfg
        H, D = cfg.hidden_size, cfg.img_feature_dim
        Dp = (D + 63) // 64 * 64
        if self.dtype == "bf16" and self.flat_lp is not None:
            L.check(L.lib().cpt_pad_cast(self.flat.data_ptr() + lo * 4, self.flat_lp.data_ptr() + lo * 2, L.CPT_BF16, 1, hi - lo,
                                         hi - lo, st), "cpt_pad_cast(shadow bucket)")
        if k == 0:
            w_img = self.view("bert.img_embedding.weight")
            if self.dtype == "bf16" and getattr(self, "img_pad_lp", None) is not None:
                L.check(L.lib().cpt_pad_cast(w_img.data_ptr(), self.img_pad_lp.data_ptr(), L.CPT_BF16, H, D, Dp, st), "cpt_pad_cast(w_img)")
            elif self.dtype != "bf16" and getattr(self, "img_pad_f32", None) is not None:
                L.check(L.lib().cpt_pad_cast(w_img.data_ptr(), self.img_pad_f32.data_ptr(), L.CPT_F32, H, D, Dp, st), "cpt_pad_cast(w_img)")
        if self.dtype == "bf16x3":
            self._x3_stale = True
        self._fold_stale = True

    def weights_updated(self, shadow_fresh=False):
        """Called after the fused optimizer wrote the flat buffer through raw pointers (which does
        not bump tensor versions): the padded img weight must be rebuilt; the bf16 shadow too unless
        the optimizer kernel already refreshed it."""
        cfg = self.cfg
        H, D = cfg.hidden_size, cfg.img_feature_dim
        Dp = (D + 63) // 64 * 64
        st = L.stream_ptr()
        w_img = self.view("bert.img_embedding.weight")
        self._sig = {}
        self._fold_stale = True
        if self.dtype == "bf16":
            if self.flat_lp is None or not shadow_fresh:
                self.refresh_shadow(force=True)
                return
            L.check(L.lib().cpt_pad_cast(w_img.data_ptr(), self.img_pad_lp.data_ptr(), L.CPT_BF16, H, D, Dp, st),
                    "cpt_pad_cast(w_img)")
        else:
            self.refresh_shadow(force=True, train=True)      # (bf16x3: the split copies wait for the next inference forward)
            return
        self._sig[self.dtype] = self._versions()

    def ensure_fold(self):
        """LayerNorm-folded copies of the two GEMM weights that consume a LayerNorm output
        (include/cpt_hip.h cpt_layer_fold), rebuilt lazily after any weight change.  bf16 inference only."""
        if not self._fold_stale and self._fold is not None:
            return
        cfg = self.cfg
        H, I, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        dev = self.flat.device
        if self._fold is None or self._fold["w"].device != dev:
            self._fold = {"w": torch.empty(nl * (3 * H * H + I * H), device=dev, dtype=torch.bfloat16),
                          "v": torch.empty(nl * 2 * (3 * H + I), device=dev, dtype=torch.float32),
                          # K-tile-major copies of the QKV weights for the (sequence, three heads) fused launch (cpt_retile_k32)
                          "t": torch.empty(nl * 3 * H * H, device=dev, dtype=torch.bfloat16) if H % 32 == 0 else None,
                          "arr": (L.LayerFold * nl)()}
            self._desc = {}
        wbuf, vbuf, arr = self._fold["w"], self._fold["v"], self._fold["arr"]
        st = L.stream_ptr()
        base = self.flat.data_ptr()

        def fp(n):
            return base + self.offsets[n][0] * 4
        wo, vo = 0, 0
        for i in range(nl):
            p = "bert.encoder.layer.%d." % i
            f = arr[i]
            f.w_in_f = wbuf.data_ptr() + wo * 2
            wo += I * H
            f.c_in = vbuf.data_ptr() + vo * 4
            f.d_in = vbuf.data_ptr() + (vo + I) * 4
            vo += 2 * I
            L.check(L.lib().cpt_fold_ln_weights(fp(p + "intermediate.dense.weight"), fp(p + "attention.output.LayerNorm.weight"),
                                                fp(p + "attention.output.LayerNorm.bias"), fp(p + "intermediate.dense.bias"),
                                                f.w_in_f, f.c_in, f.d_in, I, H, st), "cpt_fold_ln_weights(ffn up)")
            if i > 0:
                q = "bert.encoder.layer.%d." % (i - 1)
                f.w_qkv_f = wbuf.data_ptr() + wo * 2
                f.c_qkv = vbuf.data_ptr() + vo * 4
                f.d_qkv = vbuf.data_ptr() + (vo + 3 * H) * 4
                L.check(L.lib().cpt_fold_ln_weights(fp(p + "attention.self.query.weight"), fp(q + "output.LayerNorm.weight"),
                                                    fp(q + "output.LayerNorm.bias"), fp(p + "attention.self.query.bias"),
                                                    f.w_qkv_f, f.c_qkv, f.d_qkv, 3 * H, H, st), "cpt_fold_ln_weights(qkv)")
            if self._fold["t"] is not None:
                # (layer 0 reads the plain weight: its K-tile-major copy comes from the bf16 shadow)
                src = f.w_qkv_f if i > 0 else self.flat_lp.data_ptr() + self.offsets[p + "attention.self.query.weight"][0] * 2
                f.w_qkv_t = self._fold["t"].data_ptr() + i * 3 * H * H * 2
                L.check(L.lib().cpt_retile_k32(src, f.w_qkv_t, 3 * H, H, st), "cpt_retile_k32(qkv)")
            wo += 3 * H * H
            vo += 2 * 3 * H
        self._fold_stale = False

    # ---- descriptor ----------------------------------------------------------------------
    def descriptor(self, train=False):
        """cpt_model of the current compute mode.  train: the descriptor cpt_train_fwd / cpt_train_bwd take -- in bf16x3 mode the
        fp32 master weights (split per GEMM by the step itself) instead of the standing split copies inference reads."""
        x3 = self.dtype == "bf16x3"
        x3_train = x3 and bool(train)
        key = (self.dtype, bool(self.fold_ln), x3_train)
        if key in self._desc:
            return self._desc[key]
        cfg = self.cfg
        lp = self.dtype == "bf16"
        x3 = x3 and not x3_train
        esz = 2 if lp else 4
        mat_base = (self.flat_lp if lp else self.flat).data_ptr()
        vec_base = self.flat.data_ptr()
        x3_key = {}
        if x3:
            for xkey, pname, _, _ in self._x3_matrices():
                x3_key.setdefault(pname, xkey)

        def mat(n):
            if x3:
                return self.x3_buf.data_ptr() + self.x3_off[x3_key[n]] * 2
            return mat_base + self.offsets[n][0] * esz

        def vec(n):
            return vec_base + self.offsets[n][0] * 4

        D = cfg.img_feature_dim
        d = L.Dims(hidden=cfg.hidden_size, heads=cfg.num_attention_heads, inter=cfg.intermediate_size,
                   layers=cfg.num_hidden_layers, vocab=cfg.vocab_size, img_dim=D, img_dim_pad=(D + 63) // 64 * 64,
                   max_pos=cfg.max_position_embeddings, type_vocab=cfg.type_vocab_size,
                   use_img_ln=1 if getattr(cfg, "use_img_layernorm", None) else 0,
                   n_rel=getattr(cfg, "num_contrast_classes", 2) if self.head in ("pretrain", "nsp") else 0,
                   dtype=L.CPT_BF16 if lp else (L.CPT_BF16X3_MASTERS if x3_train else (L.CPT_BF16X3 if x3 else L.CPT_F32)), ln_eps=cfg.layer_norm_eps,
                   img_ln_eps=getattr(cfg, "img_layer_norm_eps", cfg.layer_norm_eps))
        layers = (L.Layer * cfg.num_hidden_layers)()
        for i in range(cfg.num_hidden_layers):
            p = "bert.encoder.layer.%d." % i
            y = layers[i]
            y.w_qkv = mat(p + "attention.self.query.weight")
            y.b_qkv = vec(p + "attention.self.query.bias")
            y.w_ao = mat(p + "attention.output.dense.weight")
            y.b_ao = vec(p + "attention.output.dense.bias")
            y.ln1_g = vec(p + "attention.output.LayerNorm.weight")
            y.ln1_b = vec(p + "attention.output.LayerNorm.bias")
            y.w_in = mat(p + "intermediate.dense.weight")
            y.b_in = vec(p + "intermediate.dense.bias")
            y.w_out = mat(p + "output.dense.weight")
            y.b_out = vec(p + "output.dense.bias")
            y.ln2_g = vec(p + "output.LayerNorm.weight")
            y.ln2_b = vec(p + "output.LayerNorm.bias")
        m = L.Model()
        m.dims = d
        m.word_emb = vec("bert.embeddings.word_embeddings.weight")
        m.pos_emb = vec("bert.embeddings.position_embeddings.weight")
        m.type_emb = vec("bert.embeddings.token_type_embeddings.weight")
        m.emb_ln_g = vec("bert.embeddings.LayerNorm.weight")
        m.emb_ln_b = vec("bert.embeddings.LayerNorm.bias")
        m.w_img = (self.x3_buf.data_ptr() + self.x3_off["img"] * 2) if x3 else (self.img_pad_lp if lp else self.img_pad_f32).data_ptr()
        m.b_img = vec("bert.img_embedding.bias")
        if d.use_img_ln:
            m.img_ln_g = vec("bert.LayerNorm.weight")
            m.img_ln_b = vec("bert.LayerNorm.bias")
        m.layers = C.cast(layers, C.POINTER(L.Layer))
        m.w_pool = mat("bert.pooler.dense.weight")
        m.b_pool = vec("bert.pooler.dense.bias")
        if self.head == "nsp":
            m.w_rel = mat("cls.weight")
            m.b_rel = vec("cls.bias")
        elif self.head != "none":
            hp = "cls." if self.head == "cpt" else "cls.predictions."
            m.w_tr = mat(hp + "transform.dense.weight")
            m.b_tr = vec(hp + "transform.dense.bias")
            m.tr_ln_g = vec(hp + "transform.LayerNorm.weight")
            m.tr_ln_b = vec(hp + "transform.LayerNorm.bias")
            m.w_dec = mat("bert.embeddings.word_embeddings.weight")
            m.b_dec = vec(hp + "bias")
        if self.head == "pretrain":
            m.w_rel = mat("cls.seq_relationship.weight")
            m.b_rel = vec("cls.seq_relationship.bias")
        if lp and self.fold_ln and self._fold is not None:
            m.fold = C.cast(self._fold["arr"], C.POINTER(L.LayerFold))
        self._desc[key] = (m, layers)     # keep `layers` alive
        return self._desc[key]

    # ---- forward -------------------------------------------------------------------------
    def workspace(self, B, Lt, Li, flags):
        m, _ = self.descriptor()
        need = L.lib().cpt_fwd_workspace_bytes(C.byref(m.dims), B, Lt, Li, flags)
        if need == 0:
            raise RuntimeError("cpt_amd: cpt_fwd_workspace_bytes rejected B=%d Lt=%d Li=%d" % (B, Lt, Li))
        ws = self._ws.get("fwd")
        if ws is None or ws.numel() < need or ws.device != self.flat.device:
            ws = torch.empty(need, device=self.flat.device, dtype=torch.uint8)
            self._ws["fwd"] = ws
        return ws

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, img_feats=None,
                mask_pos=None, labels=None, flags=0, logit_cols=None):
        """Returns dict with the outputs selected by `flags` (see _lib.OUT_*).
        logit_cols (1-D int64 tensor of vocabulary ids, with OUT_MASK_LOGITS and no loss): ``logits`` holds those columns only, (B, len(logit_cols)),
        in list order -- the decoder then reads len(logit_cols) rows of its table instead of streaming it (cpt_outputs.logit_cols)."""
        self.ensure_packed()
        self.refresh_shadow()
        if self.dtype == "bf16" and self.fold_ln:
            self.ensure_fold()
        dev = self.flat.device

        def prep(t, dt, name):
            if t is None:
                return None
            if not t.is_cuda:
                raise RuntimeError("cpt_amd: %s is on %s; inputs must be on the GPU" % (name, t.device))
            if t.dtype != dt:
                t = t.to(dt)
            return t.contiguous()

        input_ids = prep(input_ids, torch.int64, "input_ids")
        token_type_ids = prep(token_type_ids, torch.int64, "token_type_ids")
        position_ids = prep(position_ids, torch.int64, "position_ids")
        mask3 = attention_mask is not None and attention_mask.dim() == 3
        if attention_mask is not None and attention_mask.dim() not in (2, 3):
            raise NotImplementedError("cpt_amd: attention_mask must be 2-D or 3-D (modeling_bert.py:213-218)")
        attention_mask = prep(attention_mask, torch.int64, "attention_mask")
        img_feats = prep(img_feats, torch.float32, "img_feats")
        mask_pos = prep(mask_pos, torch.int64, "mask_token_pos")
        labels = prep(labels, torch.int64, "labels")
        B, Lt = input_ids.shape
        Li = img_feats.size(1) if img_feats is not None else 0
        Lseq = Lt + Li
        if img_feats is not None and img_feats.size(2) != self.cfg.img_feature_dim:
            raise RuntimeError("cpt_amd: img_feats last dim %d != config.img_feature_dim %d"
                               % (img_feats.size(2), self.cfg.img_feature_dim))
        want = (B, Lseq, Lseq) if mask3 else (B, Lseq)
        if attention_mask is not None and tuple(attention_mask.shape) != want:
            raise RuntimeError("cpt_amd: attention_mask shape %s != %s" % (tuple(attention_mask.shape), want))
        call_flags = flags | (L.ATTN_MASK_3D if mask3 else 0)      # (B, L, L): one mask row per query (modeling_bert.py:215-216)
        m, _ = self.descriptor()
        H, V = self.cfg.hidden_size, self.cfg.vocab_size
        out = {}
        o = L.Outputs()
        if flags & L.OUT_SEQ:
            out["seq"] = torch.empty((B, Lseq, H), device=dev, dtype=torch.float32)
            o.seq = out["seq"].data_ptr()
        if flags & L.OUT_POOLED:
            out["pooled"] = torch.empty((B, H), device=dev, dtype=torch.float32)
            o.pooled = out["pooled"].data_ptr()
        if logit_cols is not None:
            if not (flags & L.OUT_MASK_LOGITS) or (flags & L.OUT_LOSS):
                raise RuntimeError("cpt_amd: logit_cols goes with the [MASK]-row scores and without a loss")
            logit_cols = prep(logit_cols.reshape(-1), torch.int64, "logit_cols")
            if logit_cols.numel() == 0:
                raise RuntimeError("cpt_amd: logit_cols is empty")
        if flags & L.OUT_MASK_LOGITS:
            out["logits"] = torch.empty((B, V if logit_cols is None else logit_cols.numel()), device=dev, dtype=torch.float32)
            o.logits = out["logits"].data_ptr()
            if logit_cols is not None:
                o.logit_cols = logit_cols.data_ptr()
                o.n_logit_cols = logit_cols.numel()
        if flags & L.OUT_ALL_LOGITS:
            out["logits"] = torch.empty((B, Lseq, V), device=dev, dtype=torch.float32)
            o.logits = out["logits"].data_ptr()
        if flags & L.OUT_LOSS:
            out["loss_acc"] = torch.empty(2, device=dev, dtype=torch.float32)
            o.loss = out["loss_acc"].data_ptr()
        if flags & L.OUT_REL:
            out["rel"] = torch.empty((B, m.dims.n_rel), device=dev, dtype=torch.float32)
            o.rel = out["rel"].data_ptr()
        bt = L.Batch(B=B, Lt=Lt, Li=Li, input_ids=input_ids.data_ptr(), token_type=L.ptr(token_type_ids),
                     position_ids=L.ptr(position_ids), attn_mask=L.ptr(attention_mask), img_feats=L.ptr(img_feats),
                     mask_pos=L.ptr(mask_pos), labels=L.ptr(labels))
        ws = self.workspace(B, Lt, Li, flags)
        L.check(L.lib().cpt_model_fwd(C.byref(m), C.byref(bt), C.byref(o), call_flags, ws.data_ptr(), ws.numel(),
                                      L.stream_ptr()), "cpt_model_fwd")
        if flags & L.OUT_LOSS:
            acc = out["loss_acc"]
            out["loss"] = acc[0] / acc[1]       # mean over labelled rows (CrossEntropyLoss default)
        return out


def profile_read():
    """{kernel name: (total_ms, launches)} accumulated since cpt_prof_enable(1)."""
    res = {}
    for i, name in enumerate(L.K_NAMES):
        t, n = C.c_double(0), C.c_int64(0)
        L.check(L.lib().cpt_prof_read(i, C.byref(t), C.byref(n)), "cpt_prof_read")
        res[name] = (t.value, n.value)
    return res
