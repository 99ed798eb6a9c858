#!/usr/bin/env python
"""bench.py -- prompted (image,query) pairs/sec through the CPT [MASK]-scoring hot path.

Workload (BASELINE.json configs[1]): Oscar-base CPT RefCOCO inference, batch 64 sequences per
GPU, 50 region features + 70 text tokens (L = 120), bf16 MFMA operands, synthetic region
features of the RefCOCO shape, random-init weights (no checkpoint on the box).  One step = one
REC_MLM_CPT forward over one batch already resident in HBM, producing the [MASK]-row logits
(B x 30522 fp32) every reference consumer keeps (zeroshot/refcoco_cpt.py:219).

    python bench.py [--gpus N --steps K --warmup W] [--batch 64] [--dtype bf16|fp32] [--all-rows]
N > 1: one rank per GPU under torch.distributed.run (RCCL) -- the driver launches it that way, and a
bare `python bench.py --gpus N` re-executes itself the same way.  Inference shards sequences across
ranks with no data-path collective (weak scaling: 64 sequences per GPU); --mode train runs the
few-shot step data-parallel (bucketed reduce-scatter under backward, sharded AdamW, all-gather).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DLOGIT_LIMIT = 0.03       # bf16 throughput mode against the fp32 oracle on the bench batch: beyond this the run exits non-zero
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}      # MI355X_MICROARCH.md: dense MFMA peaks
def _latest(name):
    """newest committed round artefact profiles/rNN_<name>"""
    for r in ("r06", "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", "%s_%s" % (r, name))
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", "r06_" + name)


PMC_FILE = _latest("pmc_bench_summary.json")
LIVE_MFMA_BUSY = {}        # per kernel family, filled by live_pmc_traffic: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of this run's counter pass
YARDSTICK_FILE = _latest("yardstick.json")


def yardstick_us():
    """Best hipBLASLt time (us) of the plain bf16 GEMM of each encoder projection at the bench shape, from the committed
    tools/yardstick.hip run (profiles/r03_yardstick.json); {} when absent."""
    try:
        d = json.load(open(YARDSTICK_FILE))["hipblaslt_bf16"]
        return {k.split(" ")[0]: v["best_us"] for k, v in d.items()}
    except Exception:
        return {}


def yardstick_live(timeout_s=180):
    """hipBLASLt's best plain bf16 GEMM of the four encoder projections measured IN THIS RUN on this box (tools/yardstick.bin, built by
    __graft_entry__.build(); a measurement tool, never on the product path): ({family: best_us}, note) or ({}, why)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "yardstick.bin")
    if not os.path.exists(exe):
        return {}, "tools/yardstick.bin not built"
    try:
        r = subprocess.run([exe], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return {}, "tools/yardstick.bin failed (rc %d)" % r.returncode
        d = json.loads(r.stdout)["hipblaslt_bf16"]
        return {k.split(" ")[0]: v["best_us"] for k, v in d.items()}, "tools/yardstick.bin run inside this bench.py run, same box (hipBLASLt, plain GEMM, best of the heuristic's algorithms, back-to-back launches)"
    except Exception as e:
        return {}, "tools/yardstick.bin: %r" % (e,)


def fwd_gflop_per_seq(cfg, Lt, Li, mlm_head=True):
    """Algorithmic forward GFLOP per sequence as SURVEY.md 8(d) counts it (2 m n k; padding not discounted; head on one row)."""
    H, I, L, nl = cfg.hidden_size, cfg.intermediate_size, Lt + Li, cfg.num_hidden_layers
    layer = 2.0 * L * (4 * H * H + 2 * H * I) + 4.0 * L * L * H
    img = 2.0 * Li * cfg.img_feature_dim * H
    head = 2.0 * (H * H + H * cfg.vocab_size) if mlm_head else 2.0 * (H * H + H * 3)
    return (nl * layer + img + head + 2.0 * H * H) / 1e9


def pmc_traffic_bytes(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r02_pmc_bench_summary.json, produced by tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in
    separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams on gfx950).
    None when the summary is absent or the batch differs from the profiled one."""
    fn = PMC_FILE
    try:
        d = json.load(open(fn))
        fam = {"gemm_ffn_up": "gemm_ffn_up(+gelu)", "gemm_qkv": "gemm_qkv_attn", "gemm_attn_out": "gemm_attn_out",
               "gemm_ffn_down": "gemm_ffn_down"}.get(kernel)
        if fam is None or fam not in d:
            return None
        return int((2.0 * d[fam]["FETCH_SIZE"]["mean_per_launch"] + d[fam]["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    except Exception:
        return None


def live_pmc_traffic(kernel, timeout_s=240):
    """HBM-side traffic of `kernel` (bytes per launch) MEASURED DURING THIS RUN: two child passes of this same command under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes), a few steps each, summed per kernel family by tools/pmc_summary.py;
    2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for 16 B/lane streams), both in KiB.  Returns (bytes, note) or (None, why)."""
    import shutil
    import subprocess
    import tempfile
    fam = {"gemm_ffn_up": "gemm_ffn_up(+gelu)", "gemm_qkv": "gemm_qkv_attn", "gemm_attn_out": "gemm_attn_out", "gemm_ffn_down": "gemm_ffn_down"}.get(kernel)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if fam is None or not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    out = {}
    tmp = tempfile.mkdtemp(prefix="cpt_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
            d = os.path.join(tmp, counter.split(" ")[0])
            cmd = [exe, "--pmc"] + counter.split(" ") + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                    "--steps", "4", "--warmup", "2", "--no-cpu", "--no-roofline", "--no-extra", "--no-io", "--no-sustained"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
            if r.returncode != 0:
                if counter.startswith("SQ_"):
                    continue            # (the MFMA-busy pass is a side figure: the traffic figure stands without it)
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (counter, r.returncode)
        with open(os.devnull, "w") as devnull:
            stdout, sys.stdout = sys.stdout, devnull
            try:
                pmc_summary.main(tmp)
            finally:
                sys.stdout = stdout
        d = json.load(open(os.path.join(tmp, "summary.json")))
        if fam not in d or "FETCH_SIZE" not in d[fam] or "WRITE_SIZE" not in d[fam]:
            return None, "kernel family %s missing from the counter passes" % fam
        f, w = d[fam]["FETCH_SIZE"], d[fam]["WRITE_SIZE"]
        global LIVE_MFMA_BUSY
        # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md);
        # GRBM_GUI_ACTIVE = the launch's cycles summed over the 8 XCDs -> busy share of all SIMD cycles = MFMA_BUSY / (GRBM / 8 * 1024)
        LIVE_MFMA_BUSY = {k: round(v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (128.0 * v["GRBM_GUI_ACTIVE"]["mean_per_launch"]), 4)
                          for k, v in d.items() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", {}).get("mean_per_launch")}
        return int((2.0 * f["mean_per_launch"] + w["mean_per_launch"]) * 1024), \
            "measured in this run: two child passes of this command under rocprofv3 --pmc (FETCH_SIZE: %d launches, WRITE_SIZE: %d launches; 2 x FETCH + WRITE, MI355X_MICROARCH.md)" \
            % (f["launches"], w["launches"])
    except Exception as e:
        return None, "live PMC passes failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gemm_flops(kind, M, H, I):
    return {"gemm_qkv": 2.0 * M * 3 * H * H, "gemm_attn_out": 2.0 * M * H * H,
            "gemm_ffn_up": 2.0 * M * I * H, "gemm_ffn_down": 2.0 * M * I * H}[kind]


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()                      # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:                                                                         # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(cfg, seed, threads):
    """The oracle (CPU fp32 restatement of the reference path) on this box's host cores, as SURVEY.md 8(d) prescribes:
    the bench workload itself (B=64, 50 regions) and configs[0] (1 query x 36 regions, L_img padded to 50), 2 warm-ups,
    median of 5; [MASK]-row head (algorithmic) and, once, the all-row head as the reference computes it."""
    from cpt_amd import synth
    from oracle import cpt_oracle as O
    torch.set_num_threads(threads)
    sd = synth.init_state_dict(cfg, seed, head="cpt", randomize_all=False)
    cd = cfg.to_dict()

    last = {}

    def timed(B, n_regions, warm, iters, all_rows=False):
        b = synth.make_batch(B, cfg, seed=seed, n_regions=n_regions)

        def run():
            with torch.no_grad():
                return O.rec_mlm_cpt_forward(sd, cd, b["input_ids"], b["segment_ids"], b["attention_mask"],
                                             img_feats=b["img_feats"],
                                             mask_rows_only=None if all_rows else b["mask_token_pos"])[0]
        for _ in range(warm):
            run()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            last["logits"] = run()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return B / ts[len(ts) // 2]
    v64 = timed(64, 50, 2, 5)
    ref_logits = last["logits"]            # the oracle's [MASK]-row logits of the bench batch (seed, weights and shape of rank 0's batch): the parity check of main()
    v1 = timed(1, 36, 2, 5)
    vall = timed(64, 50, 0, 1, all_rows=True)
    return ref_logits, {"value": round(v64, 2), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "oracle (torch CPU fp32) Oscar-base forward, [MASK]-row head, 2 warm-ups + median of 5: B=64 x 50 regions "
                      "(the bench workload) %.2f pairs/s; configs[0] B=1 x 36 regions %.2f pairs/s; B=64 as the reference "
                      "computes it (all-row head, 1 run) %.2f pairs/s" % (v64, v1, vall)}


def parity_block(mode, got, ref):
    """The timed mode's [MASK]-row logits of the bench batch against the oracle's (fp32 CPU, same batch, same weights): max |d logit| and
    colour-argmax flips per sequence under both selection rules of the reference -- zero-shot: argmax of the raw colour logits
    (zeroshot/refcoco_cpt.py:242); few-shot: argmax of colour logit / "none" logit (fewshot/refcoco_cpt.py:291)."""
    from cpt_amd import synth
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    cols = torch.tensor(list(synth.COLOR_IDS))
    gc, rc = got[:, cols], ref[:, cols]
    gr, rr = gc / got[:, synth.NONE_ID][:, None], rc / ref[:, synth.NONE_ID][:, None]
    top2 = rc.topk(2, 1).values
    return {"mode": mode, "reference": "oracle (fp32 CPU restatement, pinned to the reference by tests/golden) on the same batch and weights",
            "sequences": int(got.size(0)), "max_abs_dlogit": float((got - ref).abs().max()),
            "max_abs_dlogit_colour_ids": float((gc - rc).abs().max()),
            "colour_argmax_flips_zsl": int((gc.argmax(1) != rc.argmax(1)).sum()),
            "colour_argmax_flips_fsl": int((gr.argmax(1) != rr.argmax(1)).sum()),
            "vocab_argmax_flips": int((got.argmax(1) != ref.argmax(1)).sum()),
            "median_colour_margin_of_reference": float((top2[:, 0] - top2[:, 1]).median()),
            "bar": "north_star: 1e-3 and identical argmax (met by the fp32 and bf16x3 modes; bf16 is the throughput mode, see extra.parity_modes)"}


def io_pipeline_leg(dev, model, b, seconds=2.0, workers=None, threads=2, device_decode=False):
    """The input side of the hot path INSIDE this run (zeroshot/refcoco_cpt.py:213-218, utils/tsv_file.py:20-85,
    refcoco_zsl_cpt_dataset.py:161-180): generated predictions.tsv rows -> worker processes -> shared pinned ring -> H2D on a side stream ->
    forward, every step on freshly decoded and freshly copied region features, timed for `seconds` of steady state beside `seconds / 2` of the
    forward alone on a resident batch.  device_decode (round 5): the workers only locate and copy the base64 strings (cpt_pack_tsv_rows), the
    text travels to the GPU and cpt_b64_decode_regions_device decodes it on the side stream; False: cpt_decode_tsv_rows on the host cores."""
    import tempfile
    from cpt_amd import io, synth
    B, rows_per_batch, n_rows = 64, 8, 48
    tsv_path = os.path.join(tempfile.gettempdir(), "cpt_bench_predictions_%d.tsv" % os.getuid())
    if not (os.path.exists(tsv_path) and os.path.exists(os.path.splitext(tsv_path)[0] + ".lineidx")
            and io.TSVFile(tsv_path).num_rows() == n_rows):
        synth.write_predictions_tsv(tsv_path, n_rows, rows_per_batch, 50, seed=88)
    workers = workers or max(2, min(8, usable_cores() // 2))

    def fwd(feats, mask):
        with torch.no_grad():
            return model(b["input_ids"], b["segment_ids"], mask, img_feats=feats, mask_token_pos=b["mask_token_pos"])[0]
    pool = io.DecodePool(tsv_path, max_seqs=B, workers=workers, slots=2 * workers, threads=threads, device_decode=device_decode)
    try:
        side = torch.cuda.Stream(dev)
        dfe = [torch.empty((B, 50, 2054), device=dev) for _ in range(2)]
        dma = [b["attention_mask"].clone() for _ in range(2)]
        dtx = [torch.empty((B, 50, io.b64_chars(2054)), dtype=torch.uint8, device=dev) for _ in range(2)] if device_decode else None
        dmi = [torch.zeros((B, 50), dtype=torch.int64, device=dev) for _ in range(2)]
        derr = torch.zeros(1, dtype=torch.int64, device=dev)
        consumed = [None, None]
        state = {"next": 0, "wait": 0.0}

        def pump():
            while pool.can_submit():
                i = state["next"]
                pool.submit([(rows_per_batch * i + j) % n_rows for j in range(rows_per_batch)])
                state["next"] += 1

        def step(k):
            tw = time.perf_counter()
            slot, names, infos, spr, regions = pool.next()
            state["wait"] += time.perf_counter() - tw
            S = sum(spr)
            with torch.cuda.stream(side):
                if consumed[k] is not None:
                    side.wait_event(consumed[k])            # the forward that read this device buffer is done
                dmi[k][:S].copy_(pool.masks[slot][:S], non_blocking=True)
                dma[k][:S, 70:].copy_(dmi[k][:S], non_blocking=True)
                if device_decode:
                    dtx[k][:S].copy_(pool.feats[slot][:S], non_blocking=True)
                    io.decode_text_device(dtx[k][:S], dmi[k][:S], dfe[k][:S], derr, stream=side)
                else:
                    dfe[k][:S].copy_(pool.feats[slot][:S], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
            out = fwd(dfe[k][:S], dma[k][:S])
            consumed[k] = torch.cuda.Event()
            consumed[k].record()
            ev.synchronize()                                # the copy out of the pinned slot is done: the slot is free
            pool.release(slot)
            pump()
            return out
        pump()
        step(0)
        torch.cuda.synchronize()
        # what reached the device in step 0 is bit for bit what a plain host decode of the same rows gives (rows 0..7 of the file)
        tsv = io.TSVFile(tsv_path)
        _, hf, hm, _, _ = io.decode_rows([tsv.seek_raw(i)[1].strip() for i in range(rows_per_batch)], 50, max_seqs=B)
        check = bool(torch.equal(dfe[0].cpu(), hf) and torch.equal(dma[0][:, 70:].cpu(), hm))
        for i in range(2 * workers + 8):                    # workers started, every slot touched, ring in steady state
            step((i + 1) & 1)
        torch.cuda.synchronize()
        # the forward alone on the resident batch, same clocks
        n_f, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 2:
            for _ in range(20):
                fwd(b["img_feats"], b["attention_mask"])
            torch.cuda.synchronize()
            n_f += 20
        fwd_only = B * n_f / (time.perf_counter() - t0)
        for i in range(8):
            step(i & 1)
        torch.cuda.synchronize()
        state["wait"] = 0.0
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            step(n & 1)
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        io.check_device_decode(derr, 50)
    finally:
        pool.close()
    rate = B * n / dt
    return {"measured_in_this_run": True, "end_to_end_pairs_per_s": round(rate, 1), "forward_only_pairs_per_s": round(fwd_only, 1),
            "fraction_of_forward_only": round(rate / fwd_only, 4), "workers": workers, "threads_per_worker": threads, "steps": n,
            "ms_per_step": round(dt / n * 1e3, 4), "ms_waiting_for_decode_per_step": round(state["wait"] / n * 1e3, 4),
            "decode": "device" if device_decode else "host",
            "h2d_MB_per_step": round(B * 50 * (io.b64_chars(2054) if device_decode else 2054 * 4) / 1e6, 1), "distinct_batches_in_file": n_rows // rows_per_batch, "host_cores": usable_cores(),
            "device_batch_equals_host_decode": check,
            "path": ("generated predictions.tsv (48 rows x 8 proposals x 50 boxes, base64 float32[2054]) -> cpt_amd.io.DecodePool(device_decode=True): worker "
                     "processes locate + copy the base64 strings into the shared pinned ring -> hipMemcpyAsync of the TEXT on a side stream -> "
                     "cpt_b64_decode_regions_device on that stream -> REC_MLM_CPT forward; 64 sequences per step, fresh features every step") if device_decode else
                    ("generated predictions.tsv (48 rows x 8 proposals x 50 boxes, base64 float32[2054]) -> cpt_amd.io.DecodePool (C decoder in worker processes, "
                     "shared pinned ring) -> hipMemcpyAsync on a side stream -> REC_MLM_CPT forward; 64 sequences per step, fresh features every step")}


def hbm_kernels(cfg, B, dev, iters=20):
    """HBM-bound kernels of the path run stand-alone on the bench shapes (the fused bf16 encoder folds its LayerNorms
    into GEMM epilogues, so they do not appear in the step): algorithmic bytes / HIP-event time vs 8 TB/s."""
    from cpt_amd import _lib as L, ops
    M, H = B * 120, cfg.hidden_size
    out = {}

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    def entry(name, nbytes, sec):
        out[name] = {"bytes": int(nbytes), "us": round(sec * 1e6, 2), "GB/s": round(nbytes / sec / 1e9, 1),
                     "frac_of_8TB/s": round(nbytes / sec / 8e12, 3)}
    g = torch.randn(H, device=dev)
    bt = torch.randn(H, device=dev)
    word = torch.randn(cfg.vocab_size, H, device=dev)
    posw = torch.randn(512, H, device=dev)
    typw = torch.randn(2, H, device=dev)
    # Two working sets per row kernel: the bench shape (M = 7680 rows: 35-59 MB, resident in the 256 MB Infinity Cache and short
    # enough that launch latency is a third of the time) and 8x the rows (> 256 MB in + out: what HBM itself sustains).
    for scale, tag in ((1, "bench shape, Infinity-Cache resident"), (8, "8x rows, > 256 MB working set")):
        Ms, Bs = M * scale, B * scale
        x = torch.randn(Ms, H, device=dev)
        o32 = torch.empty_like(x)
        o16 = torch.empty(Ms, H, device=dev, dtype=torch.bfloat16)
        entry("layernorm_rows (fp32 in, fp32 + bf16 out) [%s]" % tag, Ms * H * (4 + 4 + 2),
              timeit(lambda: ops.layernorm_rows(x, g, bt, 1e-12, out=o32, out_lp=o16)))
        entry("layernorm_rows (fp32 in, bf16 out) [%s]" % tag, Ms * H * (4 + 2),
              timeit(lambda: L.check(L.lib().cpt_layernorm_rows(x.data_ptr(), g.data_ptr(), bt.data_ptr(), 1e-12, None, o16.data_ptr(),
                                                                 L.CPT_BF16, Ms, H, Ms, 0, 0, L.stream_ptr()))))
        ids = torch.randint(1000, 30000, (Bs, 70), device=dev)
        tt = torch.zeros(Bs, 70, dtype=torch.long, device=dev)
        e32 = torch.empty(Bs, 120, H, device=dev)             # (preallocated: ops.embed_ln would add two fill kernels per call)
        e16 = torch.empty(Bs, 120, H, device=dev, dtype=torch.bfloat16)
        # unique-bytes model: every DISTINCT table row is read from memory once (repeats hit the caches), every output row is written
        uniq = int(torch.unique(ids).numel())
        entry("embed_ln (unique rows: %d word + 70 position + 1 type, fp32 + bf16 out) [%s]" % (uniq, tag), (uniq + 70 + 1) * H * 4 + Bs * 70 * H * (4 + 2),
              timeit(lambda: L.check(L.lib().cpt_embed_ln(ids.data_ptr(), tt.data_ptr(), None, word.data_ptr(), posw.data_ptr(), typw.data_ptr(),
                                                           g.data_ptr(), bt.data_ptr(), 1e-12, e32.data_ptr(), e16.data_ptr(), L.CPT_BF16, Bs, 70, 120, H,
                                                           cfg.vocab_size, 512, 2, L.stream_ptr()))))
        # round 5: base64 text of the region features -> float32 on the device (include/cpt_io.h cpt_b64_decode_regions_device): text read once, floats written once
        from cpt_amd import io
        chars = io.b64_chars(2054)
        txt = torch.randint(65, 91, (Bs, 50, chars), dtype=torch.uint8, device=dev)
        txt[:, :, chars - 1] = 61                                     # '=': the one padding character of a float32[2054] string
        mk = torch.ones(Bs, 50, dtype=torch.int64, device=dev)
        fo = torch.empty(Bs, 50, 2054, device=dev)
        derr = torch.zeros(1, dtype=torch.int64, device=dev)
        entry("b64_decode_regions (base64 text in, fp32 out; %d x 50 regions of float32[2054]) [%s]" % (Bs, tag), Bs * 50 * (chars + 2054 * 4),
              timeit(lambda: io.decode_text_device(txt, mk, fo, derr)))
        io.check_device_decode(derr, 50)
        del x, o32, o16, e32, e16, txt, fo
    n = 111_680_000 // 64 * 64
    p, gr, m, v = (torch.randn(n, device=dev) for _ in range(4))
    v.abs_()
    code = torch.ones(n, device=dev, dtype=torch.uint8)
    step = [0]

    def adam():
        step[0] += 1
        L.check(L.lib().cpt_adamw(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), code.data_ptr(), None, n, 3e-5, 0.9, 0.98,
                                  1e-8, 0.01, step[0], 1.0, L.stream_ptr()))
    entry("adamw (111.68 M params, 28 B/param)", n * 28, timeit(adam))
    return out


def extra_configs(dev, model, cfg, seed):
    """The other BASELINE configurations at their stated per-GPU size, a few steps each (not the headline metric; the driver's
    one bench line then carries a measured number for every configuration): configs[2] few-shot training step at 32 and at
    4 sequences per GPU (its share at DP = 8), configs[3] GQA-shape inference B = 256, configs[4] Oscar-large VCR B = 32."""
    from cpt_amd import config as cfgmod, synth, _lib, engine
    from cpt_amd.train import FusedAdamW
    out = {}

    def timed(fn, warm, steps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t_w = time.perf_counter()            # untimed: keep stepping until 0.3 s have passed since the warm-up (model construction and
        while time.perf_counter() - t_w < 0.3:      # host work between configurations let the clocks drop)
            fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def gemm_fracs(fn, steps, M, H, I):
        _lib.lib().cpt_prof_enable(1)
        for _ in range(steps):
            fn()
        prof = engine.profile_read()
        _lib.lib().cpt_prof_enable(0)
        gem = {k: prof[k] for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn_up", "gemm_ffn_down") if prof[k][1]}
        fr = {k: round(gemm_flops(k, M, H, I) / (gem[k][0] / gem[k][1] * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"], 4) for k in gem}
        return fr, (max(gem, key=lambda k: gem[k][0]) if gem else None)

    def infer_entry(m, c, B, Lt, Li, call, mlm_head):
        b = {k: v.to(dev) for k, v in synth.make_batch(B, c, seed=seed, max_seq_len=Lt, img_seq_len=Li).items()}

        def fn():
            with torch.no_grad():
                return call(m, b)
        dt = timed(fn, 2, 10)
        gf = fwd_gflop_per_seq(c, Lt, Li, mlm_head)
        fr, dom = gemm_fracs(fn, 3, B * (Lt + Li), c.hidden_size, c.intermediate_size)
        return {"pairs/s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": B, "seq_len": "%d+%d" % (Lt, Li), "steps": 10,
                "fwd_GFLOP_per_seq": round(gf, 2), "frac_of_bf16_peak_end_to_end": round(B * gf / dt / 1e3 / PEAK_TFLOPS["bf16"], 4),
                "dominant_kernel": dom, "dominant_kernel_frac": fr.get(dom), "gemm_fracs": fr}

    # parity modes on the headline workload: the two modes that meet north_star's 1e-3 / identical-argmax bar
    hb = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=seed, max_seq_len=70, img_seq_len=50).items()}
    pm, plog = {}, {}
    for md in ("bf16x3", "fp32"):
        model.set_compute_dtype(md)

        def fn():
            with torch.no_grad():
                return model(hb["input_ids"], hb["segment_ids"], hb["attention_mask"], img_feats=hb["img_feats"], mask_token_pos=hb["mask_token_pos"])[0]
        dt = timed(fn, 2, 5)
        plog[md] = fn().float().cpu()
        pm[md] = {"pairs/s": round(64 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": 64, "steps": 5}
    model.set_compute_dtype("bf16")
    out["parity_modes"] = pm
    out["_parity_logits"] = plog                # (compared with the oracle's logits once the CPU leg has produced them; removed from the line)
    # ragged batches of the headline workload (the reference's batches are sum-of-proposals long, zeroshot/refcoco_cpt.py:213-218): 63 and 48 sequences
    # run the full panel mode on rows padded up to its next shape
    rag = {}
    for Br in (63, 48):
        rb = {k: v.to(dev) for k, v in synth.make_batch(Br, cfg, seed=seed, max_seq_len=70, img_seq_len=50).items()}

        def fn():
            with torch.no_grad():
                return model(rb["input_ids"], rb["segment_ids"], rb["attention_mask"], img_feats=rb["img_feats"], mask_token_pos=rb["mask_token_pos"])[0]
        dt = timed(fn, 3, 20)
        rag["b%d" % Br] = {"pairs/s": round(Br / dt, 1), "ms_per_step": round(dt * 1e3, 4), "batch": Br, "steps": 20}
    out["config1_ragged_batches"] = rag
    # configs[3]: GQA shape on the Oscar-base model of the headline run
    out["config3_gqa_infer_b256"] = infer_entry(
        model, cfg, 256, 165, 45, lambda m, b: m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                                                 mask_token_pos=b["mask_token_pos"])[0], True)
    # configs[4]: Oscar-large (24 layers, hidden 1024), NSP-CPT head, 100 regions; weights initialised on the device
    from cpt_amd.modeling_bert import BertImgForPreTraining
    from cpt_amd.modeling_vcr import NSPCPT
    lc = cfgmod.oscar_large()
    with torch.device(dev):
        pre = BertImgForPreTraining(lc)
        big = NSPCPT(lc)
    big.copy_from_pretraining_model(pre)
    big.to(dev).eval().set_compute_dtype("bf16")
    out["config4_vcr_large_infer_b32"] = infer_entry(
        big, lc, 32, 165, 100, lambda m, b: m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"])[0], False)
    del big, pre
    torch.cuda.empty_cache()
    # configs[2]: few-shot training step (forward + backward + AdamW, dropout 0.1) on the Oscar-base model
    model.train()
    opt = FusedAdamW(model, lr=3e-5, betas=(0.9, 0.98), weight_decay=0.01)
    gf3 = 3.0 * fwd_gflop_per_seq(cfg, 70, 50, True)
    for B in (32, 4):
        b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=seed, max_seq_len=70, img_seq_len=50).items()}

        def fn():
            opt.zero_grad()
            loss, _ = model(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                            masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
            loss.backward()
            opt.step()
        dt = timed(fn, 2, 5)
        out["config2_train_step_%dseq_per_gpu" % B] = {
            "pairs/s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": B, "steps": 5,
            "fwd_bwd_GFLOP_per_seq": round(gf3, 2), "frac_of_bf16_peak_end_to_end": round(B * gf3 / dt / 1e3 / PEAK_TFLOPS["bf16"], 4),
            "note": "forward + backward + AdamW on one GPU, dropout 0.1" + ("; configs[2]'s per-GPU share at DP = 8" if B == 4 else "")}
    # the same step in the parity-grade modes (the reference's few-shot loop is fp32, fewshot/refcoco_cpt.py:245-249): bf16x3 = every GEMM of the
    # step as three bf16 MFMA terms over split fp32 operands, everything else as in fp32 mode
    b = {k: v.to(dev) for k, v in synth.make_batch(32, cfg, seed=seed, max_seq_len=70, img_seq_len=50).items()}
    for mode in ("bf16x3", "fp32"):
        try:      # (a side figure: it must never take the line down with it)
            model.set_compute_dtype(mode)

            def fn():
                opt.zero_grad()
                loss, _ = model(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                                masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
                loss.backward()
                opt.step()
            dt = timed(fn, 1, 3)
            out["config2_train_step_32seq_per_gpu_%s" % mode] = {"pairs/s": round(32 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": 32, "steps": 3}
        except Exception as e:
            out["config2_train_step_32seq_per_gpu_%s" % mode] = {"error": repr(e)[:200]}
    model.set_compute_dtype("bf16")
    model.eval()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--all-rows", action="store_true", help="vocabulary head on all 120 rows, as the reference computes it")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: BASELINE configs[1] (default, the headline metric); train: few-shot step "
                         "(configs[2]: forward+backward+grad all-reduce+AdamW, 32 sequences per GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profiles/ summary instead of two rocprofv3 --pmc child passes of this run")
    ap.add_argument("--no-io", action="store_true", help="skip the measured input-pipeline leg (extra.io_pipeline_measured)")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configurations (the `extra` object of the line)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained run behind the K timed steps")
    ap.add_argument("--tune", default="", help="debug: comma list of key=value passed to cpt_set_tuning (loads the CPT_ABLATION development build of the library)")
    ap.add_argument("--no-check", action="store_true", help="debug: skip the finite-output check (ablation runs)")
    ap.add_argument("--workload", default="refcoco", choices=["refcoco", "gqa", "vcr"],
                    help="refcoco: BASELINE configs[1] (default; the headline metric).  gqa: configs[3] shape, Oscar-base L=165+45, "
                         "default batch 256.  vcr: configs[4], Oscar-large 24 layers, NSP-CPT head, L=165+100, default batch 32")
    ap.add_argument("--print-launch", action="store_true", help="print the multi-rank launch command instead of running it")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"], help="train mode: gradient dtype on the wire")
    ap.add_argument("--force-collectives", action="store_true",
                    help="debug, train mode at one rank: run the reduce-scatter / all-gather of the data-parallel step anyway (RCCL with one rank) so that the "
                         "`comm` block of the line can be exercised on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: one rank per GPU under torch.distributed.run (the same command line the driver uses)
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        fwd = [a for a in sys.argv[1:] if a != "--print-launch"]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + fwd
        if args.print_launch:
            print(" ".join(cmd))
            return
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        os.execvpe(sys.executable, cmd, env)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks = 1
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29547")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                    # the ranks RCCL actually joined: every rank contributes 1 over the wire
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)
        try:     # RCCL's version banner (C stdio, buffered on a pipe) leaves every rank's buffer NOW, long before rank 0 prints the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    n_gpus = world

    if args.tune:
        os.environ.setdefault("CPT_AMD_ABLATION", "1")      # A/B switches exist in the development build of the library only (cpt_amd/libcpt_hip_abl.so)
    from cpt_amd import config as cfgmod, synth, _lib, engine
    from cpt_amd.modeling_rec import REC_MLM_CPT
    _lib.check(_lib.lib().cpt_check_device(local), "cpt_check_device")
    for kv in [t for t in args.tune.split(",") if t]:
        k, v = kv.split("=")
        _lib.check(_lib.lib().cpt_set_tuning(int(k), int(v)), "cpt_set_tuning")
    seed = 88
    train = args.mode == "train"
    Lt, Li = 70, 50
    if args.workload == "vcr":
        from cpt_amd.modeling_bert import BertImgForPreTraining
        from cpt_amd.modeling_vcr import NSPCPT
        cfg = cfgmod.oscar_large()
        Lt, Li = 165, 100
        pre = BertImgForPreTraining(cfg)
        pre.load_state_dict(synth.init_state_dict(cfg, seed, head="pretrain", randomize_all=False))
        pre.tie_weights()
        model = NSPCPT(cfg)
        model.copy_from_pretraining_model(pre)
        if args.batch == 64:
            args.batch = 32
        if args.all_rows:
            raise SystemExit("--workload vcr has no all-row head (NSP-CPT relation head)")
        if train and args.batch == 32:
            args.batch = 8          # configs[4] few-shot step: 4 choices x 2 questions per GPU
    else:
        cfg = cfgmod.oscar_base()
        if args.workload == "gqa":
            Lt, Li = 165, 45
            if args.batch == 64 and not train:
                args.batch = 256
        model = REC_MLM_CPT(cfg)
        model.load_state_dict(synth.init_state_dict(cfg, seed, head="cpt", randomize_all=False))
        model.tie_weights()
    model.to(dev).eval().set_compute_dtype(args.dtype)
    if train and args.batch == 64:
        args.batch = 32
    B = args.batch
    Lseq = Lt + Li
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=seed + rank, max_seq_len=Lt, img_seq_len=Li).items()}
    if train:
        from cpt_amd.train import FusedAdamW
        model.train()
        opt = FusedAdamW(model, lr=3e-5, betas=(0.9, 0.98), weight_decay=0.01,       # fewshot/refcoco_cpt.py:509-513
                         grad_wire=args.grad_wire, force_collectives=args.force_collectives and world == 1)
    mpos = None if args.all_rows else b["mask_token_pos"]
    nsp_labels = (torch.arange(B, device=dev) % 3).to(torch.int64)       # VCR: relation label per (question, choice) sequence

    def step():
        if train and args.workload == "vcr":     # fewshot/vcr_nsp_cpt.py:425-470: relation-label cross entropy through the NSP-CPT head
            opt.zero_grad()
            loss, logits = model(b["input_ids"], b["segment_ids"], b["attention_mask"], nsp_labels, img_feats=b["img_feats"])
            loss.backward()
            opt.step()
            return logits
        if train:
            opt.zero_grad()
            loss, logits = model(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                                 masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
            loss.backward()
            opt.step()                      # world > 1: sharded AdamW + parameter all-gather (reduce-scatter ran under backward)
            return logits
        with torch.no_grad():
            if args.workload == "vcr":
                return model(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"])[0]
            return model(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"],
                         mask_token_pos=mpos)[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    assert args.no_check or torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * n_gpus * args.steps / dt
    gpu_logits = out.detach().float().cpu() if (not train and args.workload == "refcoco" and not args.all_rows) else None

    # sustained figure: the same step for >= 2 s of wall time (the K timed steps above last tens of milliseconds), same bracketing
    sustained = None
    if not args.no_sustained:
        barrier()
        t0 = time.perf_counter()
        n_sus = 0
        while True:
            for _ in range(50):
                step()
            n_sus += 50
            torch.cuda.synchronize()
            flag = torch.tensor([1.0 if time.perf_counter() - t0 >= 2.0 else 0.0], device=dev)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)       # every rank runs the same number of steps
            if flag.item() > 0:
                break
        barrier()
        dts = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"pairs_per_s": round(B * n_gpus * n_sus / dts, 1), "steps": n_sus, "seconds": round(dts, 3), "ms_per_step": round(dts / n_sus * 1e3, 4)}

    comm = None
    if train and getattr(opt, "sync", None) is not None and opt.sync.collectives and opt.sync.cuda:
        # communication of the data-parallel step, measured over a few more steps with events on the communication stream (dist.ShardedGradSync.comm_report)
        opt.sync.profile = True
        opt.sync.comm_report(1)
        n_c = min(args.steps, 10)
        for _ in range(n_c):
            step()
        comm = opt.sync.comm_report(n_c)
        opt.sync.profile = False

    roof, breakdown, breakdown_note = None, None, None
    if rank == 0 and not args.no_roofline and not train:
        # per-kernel durations: HIP events recorded by the library on the launch stream around every
        # launch, over a second pass of the same K steps (events off in the timed region above)
        _lib.lib().cpt_prof_enable(1)
        for _ in range(args.steps):
            step()
        prof = engine.profile_read()
        _lib.lib().cpt_prof_enable(0)
        M, H, I = B * Lseq, cfg.hidden_size, cfg.intermediate_size
        breakdown = {k: round(t / args.steps, 4) for k, (t, n) in prof.items() if n}
        n_launch = sum(n for _, (t, n) in prof.items()) / args.steps
        ksum = sum(breakdown.values())
        breakdown_note = {"sum_ms": round(ksum, 4), "launch_brackets_per_step": round(n_launch, 1),
                          "sum_minus_ms_per_step": round(ksum - ms_per_step, 4),
                          "note": "each figure is a pair of HIP events around one launch bracket, recorded in a SECOND pass of the same steps; the pair adds "
                                  "about %.1f us per bracket over the un-instrumented step (sum - ms_per_step over the brackets), so the breakdown sums to more "
                                  "than ms_per_step; rocprofv3's kernel durations (profiles/) are the un-bracketed ones" % ((ksum - ms_per_step) / max(n_launch, 1) * 1e3)}
        gem = {k: prof[k] for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn_up", "gemm_ffn_down") if prof[k][1]}
        dom = max(gem, key=lambda k: gem[k][0])
        avg_ms = gem[dom][0] / gem[dom][1]
        ach = gemm_flops(dom, M, H, I) / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                "traffic": pmc_traffic_bytes(dom) if (B == 64 and args.dtype == "bf16" and args.workload == "refcoco") else None,
                "traffic_source": "profiles/%s: rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled, see MI355X_MICROARCH.md), committed "
                                  "with the round's artefacts -- NOT measured inside this run" % os.path.basename(PMC_FILE),
                "_want_live_pmc": bool(B == 64 and args.dtype == "bf16" and args.workload == "refcoco" and n_gpus == 1 and not args.no_live_pmc and not args.tune),
                "peak_note": "2.5 PF/s is the 2.4 GHz spec figure; on this workload the package power limiter (PPT) is active for 25-47 % of the step's time and the "
                             "shader clock averages 1.95-2.02 GHz at 1.27-1.32 kW (profiles/r06_throttle_step_vs_chain.txt, r06_power_step.txt; rounds 4 and 6, three boxes)",
                "avg_launch_ms": round(avg_ms, 5),
                "flop_per_launch": gemm_flops(dom, M, H, I),
                # every encoder GEMM against the same peak (gemm_qkv: projection flops only; its launches also run the attention)
                "all_kernels_frac": {k: round(gemm_flops(k, M, H, I) / (gem[k][0] / gem[k][1] * 1e-3) / 1e12 / peak, 4) for k in gem}}
        ys, ys_note = {}, None
        if B == 64 and args.dtype == "bf16" and args.workload == "refcoco":
            if n_gpus == 1 and not args.tune and not args.no_live_pmc:
                ys, ys_note = yardstick_live()
            if not ys:
                ys, ys_note = yardstick_us(), "profiles/%s (hipBLASLt, plain GEMM, stand-alone back-to-back launches; ANOTHER box -- %s)" % (os.path.basename(YARDSTICK_FILE), ys_note or "live run skipped")
        if ys:
            # the same launches against hipBLASLt's best PLAIN bf16 GEMM of the shape on this chip (tools/yardstick.hip; the fused
            # launches also do bias / GELU / LayerNorm / residual / attention, so 1.0 is not the bar, the trend is)
            roof["yardstick"] = {"source": ys_note,
                                 "hipblaslt_us": {k: ys[k] for k in gem if k in ys},
                                 "hipblaslt_frac_of_peak": {k: round(gemm_flops(k, M, H, I) / (ys[k] * 1e-6) / 1e12 / peak, 4) for k in gem if k in ys},
                                 "ours_us": {k: round(gem[k][0] / gem[k][1] * 1e3, 2) for k in gem},
                                 "time_ratio_hipblaslt_over_ours": {k: round(ys[k] / (gem[k][0] / gem[k][1] * 1e3), 3) for k in gem if k in ys}}
    if world > 1:
        dist.barrier()

    if rank == 0:
        line = {"metric": "prompted (image,query) pairs/sec at Oscar-base L=70+50", "value": round(value, 1),
                "unit": "pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": ("%s few-shot training step (fwd+bwd+grad all-reduce+AdamW), batch %d/GPU, "
                                        "%d regions, seq_len %d+%d, %s, dropout %.2g" % ("Oscar-large (24 layers) VCR NSP-CPT" if args.workload == "vcr" else
                                                                                         "Oscar-base CPT", B, Li, Lt, Li, args.dtype, cfg.hidden_dropout_prob)) if train else
                                       {"refcoco": "Oscar-base CPT RefCOCO inference, batch %d/GPU, 50 regions, seq_len 120, %s, [MASK]-row logits%s",
                                        "gqa": "Oscar-base CPT GQA inference (BASELINE configs[3] shape), batch %d/GPU, 45 regions, seq_len 165+45, %s, "
                                               "[MASK]-row logits%s",
                                        "vcr": "Oscar-large (24 layers, hidden 1024) VCR NSP-CPT scoring (BASELINE configs[4]), batch %d/GPU, "
                                               "100 regions, seq_len 165+100, %s, relation scores%s"}[args.workload]
                                       % (B, args.dtype, " (all-row head)" if args.all_rows else ""),
                           "global_batch": B * n_gpus, "seq_len": Lseq, "parallelism": "dp%d" % n_gpus,
                           "weights": "random-init N(0,0.02), seed 88"},
                "rccl_ranks": rccl_ranks, "sustained_2s": sustained,
                "roofline": roof, "kernel_ms_per_step": breakdown, "kernel_ms_note": breakdown_note if breakdown else None}
        if comm is not None:
            line["comm"] = comm
        if args.workload != "refcoco":
            line["metric"] = "prompted (image,query) pairs/sec, %s workload (not the headline configuration)" % args.workload
        if n_gpus == 1 and not args.no_roofline and not train and args.workload == "refcoco":
            line["hbm_kernels"] = hbm_kernels(cfg, B, dev)
        # (the other GPU configurations before the CPU leg: the GPU idles through the 20-30 s of the oracle and takes ~1 s to come back
        # to its clocks -- a 5-step measurement right behind it read 16.3 ms for a 14.0 ms step)
        extra = None
        if n_gpus == 1 and not args.no_extra and not train and args.workload == "refcoco" and args.dtype == "bf16" and not args.tune:
            try:
                extra = extra_configs(dev, model, cfg, seed)
            except Exception as e:      # side figures: a failure there is reported, the headline line still goes out
                extra = {"error": repr(e)[:300]}
                model.set_compute_dtype(args.dtype)
                model.eval()
            if not args.no_io and B == 64:
                # the input side measured in THIS run (default-on, about 3 s timed): decode workers -> pinned ring -> side-stream H2D -> forward
                try:
                    extra["io_pipeline_measured"] = io_pipeline_leg(dev, model, b, device_decode=False)
                except Exception as e:
                    extra["io_pipeline_measured"] = {"error": repr(e)[:300]}
                try:        # round 5: the same leg with the base64 TEXT copied to the GPU and decoded there (35.1 instead of 26.3 MB per step over PCIe), half as long
                    extra["io_pipeline_device_decode"] = io_pipeline_leg(dev, model, b, seconds=1.0, device_decode=True)
                except Exception as e:
                    extra["io_pipeline_device_decode"] = {"error": repr(e)[:300]}
        ref_logits = None
        if n_gpus == 1 and not args.no_cpu and not train and args.workload == "refcoco":
            ref_logits, line["cpu_baseline"] = cpu_baseline(cfg, seed, min(usable_cores(), 64))
        else:
            line["cpu_baseline"] = None
        # parity of the timed mode (and of the parity modes of `extra`) against the oracle's logits of the same batch, from the CPU leg
        plog = extra.pop("_parity_logits", {}) if extra is not None else {}
        if ref_logits is not None and gpu_logits is not None and B == 64:
            line["parity"] = parity_block(args.dtype, gpu_logits, ref_logits)
            for md, lg in plog.items():
                pb = parity_block(md, lg, ref_logits)
                extra["parity_modes"][md].update({k: pb[k] for k in ("max_abs_dlogit", "colour_argmax_flips_zsl", "colour_argmax_flips_fsl", "vocab_argmax_flips")})
        else:
            line["parity"] = None
        roof_ = line.get("roofline")
        if isinstance(roof_, dict) and roof_.pop("_want_live_pmc", False):
            # HBM-side traffic of the dominant kernel measured in THIS run (two rocprofv3 --pmc child passes; the parent idles meanwhile); the committed
            # profiles/ figure stays beside it, and stands in when the passes cannot run on the box
            tb, note = live_pmc_traffic(roof_["kernel"])
            if tb is not None:
                roof_["traffic_committed_artefact"] = roof_["traffic"]
                roof_["traffic"] = tb
                roof_["traffic_source"] = note
                if LIVE_MFMA_BUSY:
                    # north_star: "evidenced by ... MFMA utilisation": share of the chip's SIMD cycles in which the matrix pipe is busy, per kernel family of
                    # the step: SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) from a third rocprofv3 --pmc child pass of this run (the counter pass runs
                    # at a slightly lower clock than the timed region: MI355X_MICROARCH.md, DVFS note)
                    roof_["mfma_busy_frac_of_simd_cycles"] = LIVE_MFMA_BUSY
            else:
                roof_["traffic_live_note"] = note
        if extra is not None:
            line["extra"] = extra
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out LAST: RCCL writes its version banner through C stdio, which sits in the C library's buffer until it is flushed (or
        # the process exits) when stdout is a pipe -- flushed here first, the banner can only precede the line, never follow it
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
        par = line.get("parity")
        if par and par["mode"] == "bf16" and not par["max_abs_dlogit"] <= BF16_DLOGIT_LIMIT:
            # loud: the throughput mode drifted out of its band against the oracle (observed 1.0-1.9e-2; tests/test_gpu_model.py BF16_TOL 0.025)
            sys.stderr.write("bench.py: PARITY FAILURE: bf16 max |d logit| %.4g against the oracle exceeds %.3g -- the rate above is not a valid measurement\n"
                             % (par["max_abs_dlogit"], BF16_DLOGIT_LIMIT))
            sys.exit(3)


if __name__ == "__main__":
    main()
