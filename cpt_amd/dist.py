"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the MI355X node, "gloo" in the CPU tests).

The hot path shards over independent (query x proposal) sequences:
  * inference: contiguous shards of whole QUERIES per rank (all proposals of a query stay together,
    the argmax is per query), NO collective inside the model; one fixed-shape gather of the chosen
    indices / scores at the end -- replaces the pickled-dict all_gather of
    /root/reference/Oscar/oscar/utils/comm.py:102-142 (called from zeroshot/refcoco_cpt.py:256).
  * training: replicated weights; per parameter bucket a reduce-scatter of the flat gradient under backward, AdamW on
    the 1/N shard, an all-gather of the updated parameters under the next forward (ShardedGradSync below; replaces
    DistributedDataParallel's bucketed all-reduce + replicated optimizer, fewshot/refcoco_cpt.py:516-522).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """env:// rendezvous as torch.distributed.launch / torchrun set it up
    (zeroshot/refcoco_cpt.py:114-124)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    return world


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) shard of n_items for this rank; sizes differ by at most one and every
    item is owned exactly once (the reference's DistributedSampler pads by repetition instead and
    then asserts duplicate predictions agree, zeroshot/refcoco_cpt.py:259)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_fixed(local, n_total, fill=0):
    """All ranks contribute a (n_local, ...) tensor for their shard_range; every rank gets the
    (n_total, ...) concatenation in global order.  Fixed-shape all_gather (padded to the largest
    shard), no pickling."""
    rank, world = rank_world()
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.full((mx,) + tuple(local.shape[1:]), fill, dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def allreduce_mean_(flat):
    """In-place mean over ranks of one flat tensor (the whole gradient buffer)."""
    rank, world = rank_world()
    if world > 1:
        dist.all_reduce(flat)
        flat.div_(world)
    return flat


def broadcast_(flat, src=0):
    """Replicate rank `src`'s flat parameter buffer (what DDP does at construction)."""
    rank, world = rank_world()
    if world > 1:
        dist.broadcast(flat, src)
    return flat


class ShardedGradSync(object):
    """Bucketed, overlapped gradient synchronisation for the flat buffers of the training step.

    The reference wraps the model in DistributedDataParallel (fewshot/refcoco_cpt.py:516-522): 25 MB buckets of an
    all-reduce that overlaps backward, then every rank runs the whole optimizer.  On the MI355X node a ring
    all-reduce of the 447 MB gradient is bound by ONE xGMI link (2*(7/8)*447 MB / 153 GB/s = 5.1 ms); reduce-scatter
    + all-gather uses all seven links of the fully connected mesh (2*(447/8) MB / 153 GB/s = 0.73 ms), and between the
    two every rank only owns 1/N of the AdamW work (ZeRO-1): this class does that, per parameter bucket
    (engine.bucket_of: embeddings | one per encoder layer | head), so that

      backward :  bucket k's reduce-scatter starts on the communication stream as soon as cpt_train_bwd_ex reports its
                  gradients complete (head first, layer N-1 .. 0, embeddings last) and runs under the remaining backward;
      step     :  AdamW on this rank's shard of every bucket, then the buckets' all-gathers are queued in FORWARD
                  order (embeddings, layer 0 ..) on the communication stream;
      forward  :  cpt_train_fwd_ex asks for bucket k right before its first use -> the compute stream waits for that
                  all-gather only, so the parameter all-gather overlaps the next forward.

    buckets: {k: (lo, hi)} element ranges of the flat buffers, (hi - lo) % world == 0.  Works on CPU tensors with gloo
    (tests) and on device tensors with RCCL ("nccl") or gloo; collectives are torch.distributed's.
    wire: None = fp32 gradients on the wire, torch.bfloat16 = gradients cast to bf16 for the reduce-scatter
    (half the bytes; the sum is then a bf16 sum).
    """

    def __init__(self, buckets, device, wire=None, group=None, force_collectives=False):
        self.rank, self.world = rank_world()
        self.group = group
        # world size 1 normally short-circuits every collective (a copy); force_collectives sends them through
        # torch.distributed anyway (needs an initialised process group), so that the stream / event choreography of the
        # RCCL path can be exercised on ONE GPU (tests/test_gpu_dp.py::test_rccl_path_on_one_gpu)
        self.collectives = self.world > 1 or bool(force_collectives)
        if self.collectives and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("cpt_amd.dist: force_collectives needs torch.distributed.init_process_group first")
        self.buckets = dict(buckets)
        self.order = sorted(self.buckets)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.wire = wire
        for k, (lo, hi) in self.buckets.items():
            if (hi - lo) % self.world:
                raise ValueError("bucket %d has %d elements, not a multiple of world size %d" % (k, hi - lo, self.world))
        # this rank's shard of bucket k lives at [soff[k], soff[k] + n_k / world) of the shard-sized buffers
        self.soff, o = {}, 0
        for k in self.order:
            lo, hi = self.buckets[k]
            self.soff[k] = o
            o += (hi - lo) // self.world
        self.shard_elems = o
        self.gshard = torch.zeros(o, device=self.device, dtype=torch.float32)    # reduced gradient shards
        self.pstage = torch.empty(o, device=self.device, dtype=torch.float32)    # updated parameter shards (all-gather source)
        self.wire_full = self.wire_shard = None
        self.comm = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._rs = {}          # bucket -> (work, event) of a reduce-scatter in flight
        self._ag = {}          # bucket -> (work, event) of an all-gather in flight
        self.reduced = set()
        # comm profiling (bench.py --mode train, `comm` block): timing events around every collective on the communication stream
        # and around every wait of the compute stream for one -- see comm_report()
        self.profile = False
        self._prof = {"rs": [], "ag": [], "wait_rs": [], "wait_ag": []}

    # -- ranges ------------------------------------------------------------------------------------------------
    def shard_range(self, k):
        """(lo, hi) of this rank's shard of bucket k inside the FLAT buffers."""
        lo, hi = self.buckets[k]
        n = (hi - lo) // self.world
        return lo + self.rank * n, lo + (self.rank + 1) * n

    def shard_view(self, buf, k):
        lo, hi = self.buckets[k]
        n = (hi - lo) // self.world
        return buf[self.soff[k]: self.soff[k] + n]

    # -- backward: reduce-scatter -------------------------------------------------------------------------------
    def begin_backward(self):
        """Call BEFORE the gradient buffer is touched: reduce-scatters of an earlier backward that no step() consumed may still
        be reading it (device: the compute stream waits for their events; host-driven backends: the works are waited for)."""
        for work, done in self._rs.values():
            if work is not None:
                work.wait()
            if done is not None:
                torch.cuda.current_stream(self.device).wait_event(done)
        self.reduced = set()
        self._rs = {}

    def grads_ready(self, grad, k):
        """Bucket k of the flat gradient `grad` is final on the current stream: start its reduce-scatter (sum)."""
        if k in self.reduced:
            return
        self.reduced.add(k)
        lo, hi = self.buckets[k]
        out = self.shard_view(self.gshard, k)
        if not self.collectives:
            out.copy_(grad[lo:hi])
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                t0 = self._tick(self.comm)
                work = self._reduce_scatter(grad, k, lo, hi, out)
                if work is not None:
                    work.wait()        # RCCL: the communication stream waits for the collective (no host block)
                done = torch.cuda.Event(enable_timing=self.profile)
                done.record(self.comm)
                if t0 is not None:
                    self._prof["rs"].append((k, t0, done))
            self._rs[k] = (None, done)
        else:
            self._rs[k] = (self._reduce_scatter(grad, k, lo, hi, out), None)

    def _reduce_scatter(self, grad, k, lo, hi, out):
        if self.wire is None:
            return dist.reduce_scatter_tensor(out, grad[lo:hi], group=self.group, async_op=True)
        if self.wire_full is None:
            self.wire_full = torch.empty(max(h - l for l, h in self.buckets.values()) if not self.cuda else grad.numel(),
                                         device=self.device, dtype=self.wire)
            self.wire_shard = torch.empty(self.shard_elems, device=self.device, dtype=self.wire)
        src = self.wire_full[lo:hi] if self.cuda else self.wire_full[: hi - lo]
        src.copy_(grad[lo:hi])
        wout = self.shard_view(self.wire_shard, k)
        work = dist.reduce_scatter_tensor(wout, src, group=self.group, async_op=True)
        if not self.cuda:
            work.wait()
            out.copy_(wout)
            return None
        work.wait()            # stream-level dependency on the communication stream, not a host wait
        out.copy_(wout)
        return None

    def finish_reduce(self, grad):
        """All buckets reduced and visible to the current stream (buckets whose callback never came are reduced now)."""
        for k in self.order:
            if k not in self.reduced:
                self.grads_ready(grad, k)
        cur = torch.cuda.current_stream(self.device) if self.cuda else None
        t0 = self._tick(cur)
        for k, (work, done) in self._rs.items():
            if work is not None:
                work.wait()
            if done is not None:
                cur.wait_event(done)
        if t0 is not None:
            self._prof["wait_rs"].append((t0, self._tick(cur)))
        self._rs = {}

    # -- step: sharded update + all-gather ------------------------------------------------------------------------
    def all_gather_params(self, flat, after_update=None):
        """Queue the all-gather of every bucket of `flat` (this rank's shard already updated in place), buckets in
        forward order.  after_update(k) runs on the current stream before bucket k is sent (unused by default)."""
        self._ag = {}
        if not self.collectives:
            return
        for k in self.order:
            lo, hi = self.buckets[k]
            slo, shi = self.shard_range(k)
            stage = self.shard_view(self.pstage, k)
            stage.copy_(flat[slo:shi])
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ev)
                    t0 = self._tick(self.comm)
                    work = dist.all_gather_into_tensor(flat[lo:hi], stage, group=self.group, async_op=True)
                    work.wait()
                    done = torch.cuda.Event(enable_timing=self.profile)
                    done.record(self.comm)
                    if t0 is not None:
                        self._prof["ag"].append((k, t0, done))
                self._ag[k] = (None, done)
            else:
                self._ag[k] = (dist.all_gather_into_tensor(flat[lo:hi], stage, group=self.group, async_op=True), None)

    def params_pending(self):
        return bool(self._ag)

    def wait_params(self, k=None):
        """The current stream may read bucket k's parameters (k None: all buckets) after this returns."""
        ks = self.order if k is None else [k]
        for kk in ks:
            ent = self._ag.pop(kk, None)
            if ent is None:
                continue
            work, done = ent
            if work is not None:
                work.wait()
            if done is not None:
                cur = torch.cuda.current_stream(self.device)
                t0 = self._tick(cur)
                cur.wait_event(done)
                if t0 is not None:
                    self._prof["wait_ag"].append((t0, self._tick(cur)))

    # -- comm profiling ----------------------------------------------------------------------------------------------
    def _tick(self, stream):
        if not (self.profile and self.cuda and stream is not None):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def comm_report(self, steps):
        """Per-step communication figures of the profiled steps (set .profile = True first; device + collectives only): bytes per
        bucket, reduce-scatter / all-gather time on the communication stream, the time the COMPUTE stream spent waiting for them
        (finish_reduce in step(), wait_params in the forward), and the fraction of the communication time that was hidden.  Clears the
        recorded events.  Bound for comparison: 2 x (7/8) x 447 MB over seven xGMI links x 153 GB/s = 0.73 ms."""
        if not (self.cuda and self.collectives):
            return None
        torch.cuda.synchronize(self.device)
        el = lambda a, b: a.elapsed_time(b)
        wire_b = 2 if self.wire is torch.bfloat16 else 4
        per = {}
        for k, a, b in self._prof["rs"]:
            per.setdefault(k, {"rs_ms": 0.0, "ag_ms": 0.0})["rs_ms"] += el(a, b)
        for k, a, b in self._prof["ag"]:
            per.setdefault(k, {"rs_ms": 0.0, "ag_ms": 0.0})["ag_ms"] += el(a, b)
        rs = sum(v["rs_ms"] for v in per.values()) / steps
        ag = sum(v["ag_ms"] for v in per.values()) / steps
        wrs = sum(el(a, b) for a, b in self._prof["wait_rs"]) / steps
        wag = sum(el(a, b) for a, b in self._prof["wait_ag"]) / steps
        n_el = sum(hi - lo for lo, hi in self.buckets.values())
        rep = {"ranks": self.world, "buckets": len(self.order), "elements": n_el,
               "reduce_scatter_bytes_per_rank_per_step": n_el * wire_b, "all_gather_bytes_per_rank_per_step": n_el * 4,
               "reduce_scatter_ms_per_step": round(rs, 4), "all_gather_ms_per_step": round(ag, 4),
               "compute_stream_wait_ms_per_step": {"reduce_scatter (in step())": round(wrs, 4), "all_gather (in the next forward)": round(wag, 4)},
               "fraction_hidden": {"reduce_scatter": round(1.0 - wrs / rs, 4) if rs > 0 else None, "all_gather": round(1.0 - wag / ag, 4) if ag > 0 else None},
               "per_bucket": {str(k): {"bytes_rs": (self.buckets[k][1] - self.buckets[k][0]) * wire_b, "bytes_ag": (self.buckets[k][1] - self.buckets[k][0]) * 4,
                                       "rs_ms": round(v["rs_ms"] / steps, 4), "ag_ms": round(v["ag_ms"] / steps, 4)} for k, v in sorted(per.items())},
               "bound_ms_7_links_fp32": round(2 * (self.world - 1) / max(self.world, 1) * n_el * 4 / (7 * 153e9) * 1e3, 3) if self.world > 1 else None}
        self._prof = {"rs": [], "ag": [], "wait_rs": [], "wait_ag": []}
        return rep

    # -- optimizer-state helpers -----------------------------------------------------------------------------------
    def gather_full(self, shard_buf, total):
        """Full flat tensor from per-rank shard buffers (checkpointing the sharded AdamW moments)."""
        full = torch.zeros(total, device=self.device, dtype=shard_buf.dtype)
        for k in self.order:
            lo, hi = self.buckets[k]
            sv = self.shard_view(shard_buf, k).contiguous()
            if not self.collectives:
                full[lo:hi].copy_(sv)
            else:
                dist.all_gather_into_tensor(full[lo:hi], sv, group=self.group)
        return full

    def scatter_full(self, full, shard_buf):
        for k in self.order:
            slo, shi = self.shard_range(k)
            self.shard_view(shard_buf, k).copy_(full[slo:shi])
