"""Operator-level wrappers over the C ABI (torch tensors in, torch tensors out).

These mirror the entry points of include/cpt_hip.h one to one; the parity tests drive
them directly.  All tensors must live on the GPU; nothing here computes on the host.
"""
import torch

from . import _lib as L


def _dt(t):
    if t.dtype == torch.float32:
        return L.CPT_F32
    if t.dtype == torch.bfloat16:
        return L.CPT_BF16
    raise TypeError("cpt_amd: unsupported dtype %s" % t.dtype)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cpt_amd: tensors must be on the GPU (the HIP path has no CPU fallback)")


def gemm(a, w, bias=None, epi=L.EPI_NONE, resid=None, out_dtype=None, tile=None):
    """epi(a[M,K] @ w[N,K].T + bias (+ resid)).  tile: one of the shipped tile configurations for THIS call (cpt_gemm_tile; None: by shape)."""
    _need_cuda(a, w, bias, resid)
    assert a.dim() == 2 and w.dim() == 2 and a.size(1) == w.size(1) and a.dtype == w.dtype
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.size(0)
    out_dtype = out_dtype or a.dtype
    out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    if resid is not None:
        assert resid.dtype == torch.float32 and resid.shape == (M, N) and resid.stride(1) == 1
    args = (_dt(a), epi, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), L.ptr(bias),
            L.ptr(resid), resid.stride(0) if resid is not None else 0, out.data_ptr(),
            L.CPT_BF16 if out_dtype == torch.bfloat16 else L.CPT_F32, out.stride(0), M, N, K, L.stream_ptr())
    if tile is None:
        L.check(L.lib().cpt_gemm(*args), "cpt_gemm")
    else:
        L.check(L.lib().cpt_gemm_tile(int(tile), *args), "cpt_gemm_tile")
    return out


def embed_ln(ids, tt, pos, word, posw, typew, g, b, eps, L_total, lp_dtype=None):
    _need_cuda(ids, word)
    B, Lt = ids.shape
    H = word.size(1)
    out = torch.zeros((B, L_total, H), device=ids.device, dtype=torch.float32)
    out_lp = torch.zeros((B, L_total, H), device=ids.device, dtype=lp_dtype) if lp_dtype is not None else None
    L.check(L.lib().cpt_embed_ln(ids.data_ptr(), L.ptr(tt), L.ptr(pos), word.data_ptr(), posw.data_ptr(),
                                 typew.data_ptr(), g.data_ptr(), b.data_ptr(), float(eps), out.data_ptr(),
                                 L.ptr(out_lp), L.CPT_BF16 if lp_dtype == torch.bfloat16 else L.CPT_F32, B, Lt,
                                 L_total, H, word.size(0), posw.size(0), typew.size(0), L.stream_ptr()),
            "cpt_embed_ln")
    return out, out_lp


def layernorm_rows(x, g, b, eps, lp_dtype=None, out=None, out_lp=None, grp=None, grp_stride=0, grp_off=0):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    R, H = x.shape
    if out is None and grp is None:
        out = torch.empty_like(x)
    if lp_dtype is not None and out_lp is None:
        out_lp = torch.empty((R, H), device=x.device, dtype=lp_dtype)
    L.check(L.lib().cpt_layernorm_rows(x.data_ptr(), L.ptr(g), L.ptr(b), float(eps), L.ptr(out), L.ptr(out_lp),
                                       L.CPT_BF16 if (out_lp is not None and out_lp.dtype == torch.bfloat16) else L.CPT_F32,
                                       R, H, grp or R, grp_stride, grp_off, L.stream_ptr()), "cpt_layernorm_rows")
    return out, out_lp


def attention(qkv, attn_mask, B, Lseq, heads, want_probs=False):
    _need_cuda(qkv, attn_mask)
    assert qkv.is_contiguous() and qkv.shape == (B * Lseq, 3 * heads * 64)
    ctx = torch.empty((B * Lseq, heads * 64), device=qkv.device, dtype=qkv.dtype)
    probs = torch.zeros((B, heads, Lseq, Lseq), device=qkv.device, dtype=qkv.dtype) if want_probs else None
    L.check(L.lib().cpt_attention(_dt(qkv), qkv.data_ptr(), L.ptr(attn_mask), ctx.data_ptr(), L.ptr(probs), B,
                                  Lseq, heads, L.stream_ptr()), "cpt_attention")
    return (ctx, probs) if want_probs else ctx


def pad_cast(x, Kp, dtype):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    R, K = x.shape
    out = torch.empty((R, Kp), device=x.device, dtype=dtype)
    L.check(L.lib().cpt_pad_cast(x.data_ptr(), out.data_ptr(), _dt(out), R, K, Kp, L.stream_ptr()), "cpt_pad_cast")
    return out


def gather_rows(src, pos, B, Lseq):
    _need_cuda(src, pos)
    H = src.size(-1)
    out = torch.empty((B, H), device=src.device, dtype=src.dtype)
    L.check(L.lib().cpt_gather_rows(src.data_ptr(), _dt(src), L.ptr(pos), out.data_ptr(), B, Lseq, H,
                                    L.stream_ptr()), "cpt_gather_rows")
    return out


def ce_rows(logits, labels, want_grad=False):
    _need_cuda(logits, labels)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.int64
    R, V = logits.shape
    loss = torch.zeros(2, device=logits.device, dtype=torch.float32)
    d = torch.empty_like(logits) if want_grad else None
    L.check(L.lib().cpt_ce_rows(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), L.ptr(d), R, V,
                                L.stream_ptr()), "cpt_ce_rows")
    return (loss, d) if want_grad else loss


def ln_stat_slots(hidden):
    """Slots per row of the partial row-sum table of a `hidden`-wide LayerNorm producer (96-column blocks, rounded up to even)."""
    parts = (hidden + 95) // 96
    return (parts + 1) & ~1


def row_stats_table(x, hidden=None):
    """Partial row sums [M][slots][2] (sum, sum of squares per 96-column block) of fp32 x[M][hidden], as gemm_ln_prod writes them."""
    M, H = x.shape
    slots = ln_stat_slots(H)
    st = torch.zeros((M, slots, 2), device=x.device, dtype=torch.float32)
    for p in range((H + 95) // 96):
        blk = x[:, p * 96:(p + 1) * 96].float()
        st[:, p, 0] = blk.sum(1)
        st[:, p, 1] = (blk * blk).sum(1)
    return st


def gemm_nn(a, w, resid=None, out_dtype=torch.float32, split_scratch=False):
    """out[M, N] = a @ w (+ resid) for bf16 a[M, K], w[Kw, N] (w = an nn.Linear weight as stored: the data-gradient form).
    Kw <= K: K is a's column count rounded up to a multiple of 64 with zero columns; w rows beyond Kw count as zero."""
    _need_cuda(a, w, resid)
    M, K = a.shape
    N = w.size(1)
    out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    part = torch.empty((64 * M * N,), device=a.device, dtype=torch.float32) if split_scratch else None
    L.check(L.lib().cpt_gemm_nn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), L.ptr(resid), resid.stride(0) if resid is not None else 0,
                                out.data_ptr(), L.CPT_BF16 if out_dtype == torch.bfloat16 else L.CPT_F32, N, M, N, K, w.size(0),
                                L.ptr(part), part.numel() * 4 if part is not None else 0, L.stream_ptr()), "cpt_gemm_nn")
    return out


def gemm_tn(a, w, split_scratch=True):
    """out[M, N] fp32 = a.T @ w for bf16 a[K, M], w[K, N] (the weight-gradient form: contraction over rows, no transposed copies)."""
    _need_cuda(a, w)
    K, M = a.shape
    N = w.size(1)
    Kp = (K + 63) // 64 * 64            # rows beyond K read as zero (buffer bounds)
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    part = torch.empty((8 * M * N,), device=a.device, dtype=torch.float32) if split_scratch else None
    L.check(L.lib().cpt_gemm_tn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), N, M, N, Kp, K,
                                L.ptr(part), part.numel() * 4 if part is not None else 0, L.stream_ptr()), "cpt_gemm_tn")
    return out


def resid3_split(x):
    """fp32 x -> (hi bf16, lo int8) of the 3-byte residual stream (include/cpt_hip.h cpt_resid3_split)."""
    _need_cuda(x)
    x = x.contiguous()
    hi = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    lo = torch.empty(x.shape, device=x.device, dtype=torch.int8)
    L.check(L.lib().cpt_resid3_split(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), L.stream_ptr()), "cpt_resid3_split")
    return hi, lo


def resid3_merge(hi, lo, pos=None, L_rows=None):
    """(hi, lo)[R*][H] -> fp32; with L_rows: one row per group of L_rows source rows, row pos[r] of group r (pos None: row 0)."""
    _need_cuda(hi, lo)
    H = hi.size(-1)
    if L_rows is None:
        R = hi.numel() // H
        out = torch.empty((R, H), device=hi.device, dtype=torch.float32)
        L.check(L.lib().cpt_resid3_merge(hi.data_ptr(), lo.data_ptr(), None, out.data_ptr(), R, 1, H, 0, L.stream_ptr()), "cpt_resid3_merge")
        return out
    R = hi.numel() // H // L_rows
    out = torch.empty((R, H), device=hi.device, dtype=torch.float32)
    L.check(L.lib().cpt_resid3_merge(hi.data_ptr(), lo.data_ptr(), L.ptr(pos), out.data_ptr(), R, L_rows, H, 1, L.stream_ptr()), "cpt_resid3_merge")
    return out


def gemm_ln_prod3(a, w, bias, resid_hi, resid_lo, st_in=None, g_in=None, b_in=None, eps=1e-12, hidden=None):
    """gemm_ln_prod with the residual and the output in the 3-byte form: returns (out_hi bf16, out_lo int8, st_out)."""
    _need_cuda(a, w, bias, resid_hi, resid_lo)
    M, K = a.shape
    N = w.size(0)
    hidden = hidden or N
    out_hi = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    out_lo = torch.empty((M, N), device=a.device, dtype=torch.int8)
    st_out = torch.zeros((M, ln_stat_slots(N), 2), device=a.device, dtype=torch.float32)
    L.check(L.lib().cpt_gemm_ln_prod3(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), L.ptr(bias), resid_hi.data_ptr(), resid_lo.data_ptr(),
                                      resid_hi.stride(0), L.ptr(st_in), L.ptr(g_in), L.ptr(b_in), float(eps), hidden, out_hi.data_ptr(),
                                      out_lo.data_ptr(), st_out.data_ptr(), out_hi.stride(0), M, N, K, L.stream_ptr()), "cpt_gemm_ln_prod3")
    return out_hi, out_lo, st_out


def panel_pack(a, to_panel=True, K=None):
    """Row-major bf16 [M, K] -> the fragment-major panel copy [M/32][K/16][64][8] the panel producers read (flat tensor), or back."""
    _need_cuda(a)
    if to_panel:
        M, K = a.shape
        out = torch.empty(M * K, device=a.device, dtype=torch.bfloat16)
        L.check(L.lib().cpt_panel_pack(a.data_ptr(), a.stride(0), out.data_ptr(), M, K, 1, L.stream_ptr()), "cpt_panel_pack")
        return out
    M = a.numel() // K
    out = torch.empty((M, K), device=a.device, dtype=torch.bfloat16)
    L.check(L.lib().cpt_panel_pack(a.data_ptr(), K, out.data_ptr(), M, K, 0, L.stream_ptr()), "cpt_panel_pack")
    return out


def gemm_ln_prod3_panel(a_panel, K, w, bias, resid_hi, resid_lo, st_in=None, g_in=None, b_in=None, eps=1e-12, hidden=None, waves=None):
    """gemm_ln_prod3 with A given as its panel copy (panel_pack): A goes straight to registers, only W through LDS.  waves: 8 / 4 = that wave
    shape of the tile for this call (cpt_gemm_ln_prod3_panel_waves; None: by shape)."""
    _need_cuda(a_panel, w, bias, resid_hi, resid_lo)
    M = a_panel.numel() // K
    N = w.size(0)
    hidden = hidden or N
    out_hi = torch.empty((M, N), device=w.device, dtype=torch.bfloat16)
    out_lo = torch.empty((M, N), device=w.device, dtype=torch.int8)
    st_out = torch.zeros((M, ln_stat_slots(N), 2), device=w.device, dtype=torch.float32)
    args = (a_panel.data_ptr(), w.data_ptr(), w.stride(0), L.ptr(bias), resid_hi.data_ptr(), resid_lo.data_ptr(),
            resid_hi.stride(0), L.ptr(st_in), L.ptr(g_in), L.ptr(b_in), float(eps), hidden, out_hi.data_ptr(),
            out_lo.data_ptr(), st_out.data_ptr(), out_hi.stride(0), M, N, K, L.stream_ptr())
    if waves is None:
        L.check(L.lib().cpt_gemm_ln_prod3_panel(*args), "cpt_gemm_ln_prod3_panel")
    else:
        L.check(L.lib().cpt_gemm_ln_prod3_panel_waves(int(waves), *args), "cpt_gemm_ln_prod3_panel_waves")
    return out_hi, out_lo, st_out


def panel_pack_bytes(a, to_panel=True, K=None):
    """panel_pack for one-byte elements (the residual stream's lo bytes): int8 [M, K] <-> [M/32][K/16][64][8] bytes (flat)."""
    _need_cuda(a)
    if to_panel:
        M, K = a.shape
        out = torch.empty(M * K, device=a.device, dtype=torch.int8)
        L.check(L.lib().cpt_panel_pack_bytes(a.data_ptr(), a.stride(0), out.data_ptr(), M, K, 1, L.stream_ptr()), "cpt_panel_pack_bytes")
        return out
    M = a.numel() // K
    out = torch.empty((M, K), device=a.device, dtype=torch.int8)
    L.check(L.lib().cpt_panel_pack_bytes(a.data_ptr(), K, out.data_ptr(), M, K, 0, L.stream_ptr()), "cpt_panel_pack_bytes")
    return out


def gemm_ln_prod3_rpanel(a_panel, K, w, bias, resid_hi_panel, resid_lo_panel, st_in=None, g_in=None, b_in=None, eps=1e-12, hidden=None, waves=0):
    """gemm_ln_prod3_panel with the residual stream (input and output) in the panel layout too (round 5): returns the PANEL copies
    (out_hi bf16 flat, out_lo int8 flat) and st_out."""
    _need_cuda(a_panel, w, bias, resid_hi_panel, resid_lo_panel)
    M = a_panel.numel() // K
    N = w.size(0)
    hidden = hidden or N
    out_hi = torch.empty(M * N, device=w.device, dtype=torch.bfloat16)
    out_lo = torch.empty(M * N, device=w.device, dtype=torch.int8)
    st_out = torch.zeros((M, ln_stat_slots(N), 2), device=w.device, dtype=torch.float32)
    L.check(L.lib().cpt_gemm_ln_prod3_rpanel(a_panel.data_ptr(), w.data_ptr(), w.stride(0), L.ptr(bias), resid_hi_panel.data_ptr(), resid_lo_panel.data_ptr(),
                                             L.ptr(st_in), L.ptr(g_in), L.ptr(b_in), float(eps), hidden, out_hi.data_ptr(), out_lo.data_ptr(),
                                             st_out.data_ptr(), M, N, K, int(waves), L.stream_ptr()), "cpt_gemm_ln_prod3_rpanel")
    return out_hi, out_lo, st_out


def gemm_ln_cons(a, wf, st_in, colc, cold, eps, hidden, gelu, tile=None):
    """[gelu]( rstd * (a @ wf.T - mean * colc) + cold ): bf16 a[M,K], wf[N,K]; st_in from row_stats_table / gemm_ln_prod.  tile: one of the
    shipped kernels / tile configurations for THIS call (cpt_gemm_ln_cons_tile; None: by shape)."""
    _need_cuda(a, wf, st_in, colc, cold)
    M, K = a.shape
    N = wf.size(0)
    out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    args = (a.data_ptr(), a.stride(0), wf.data_ptr(), wf.stride(0), st_in.data_ptr(), colc.data_ptr(), cold.data_ptr(),
            float(eps), hidden, 1 if gelu else 0, out.data_ptr(), out.stride(0), M, N, K, L.stream_ptr())
    if tile is None:
        L.check(L.lib().cpt_gemm_ln_cons(*args), "cpt_gemm_ln_cons")
    else:
        L.check(L.lib().cpt_gemm_ln_cons_tile(int(tile), *args), "cpt_gemm_ln_cons_tile")
    return out


def gemm_ln_prod(a, w, bias, resid, st_in=None, g_in=None, b_in=None, eps=1e-12, hidden=None):
    """out = a @ w.T + bias + R with R = resid, or LayerNorm(resid; st_in, g_in, b_in) when g_in is given.
    Returns (out_f32, out_bf16, st_out) -- st_out = partial row sums of out (the table the next consumer reads)."""
    _need_cuda(a, w, bias, resid)
    M, K = a.shape
    N = w.size(0)
    hidden = hidden or N
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    out_lp = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    st_out = torch.zeros((M, ln_stat_slots(N), 2), device=a.device, dtype=torch.float32)
    L.check(L.lib().cpt_gemm_ln_prod(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), L.ptr(bias), resid.data_ptr(), resid.stride(0),
                                     L.ptr(st_in), L.ptr(g_in), L.ptr(b_in), float(eps), hidden, out.data_ptr(), out_lp.data_ptr(),
                                     st_out.data_ptr(), out.stride(0), M, N, K, L.stream_ptr()), "cpt_gemm_ln_prod")
    return out, out_lp, st_out
