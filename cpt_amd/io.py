"""Region-feature input pipeline: TSV + lineidx reader, wire-format decode into pinned
host memory through the C ABI (include/cpt_io.h), and staging to the GPU on a side stream.

Counterparts in the reference:
  * ``TSVFile``            Oscar/oscar/utils/tsv_file.py:8-85 (seek by byte offset from the .lineidx companion)
  * ``decode_features``    Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:161-180 (json -> per-box base64 -> float32)
  * zero padding + image part of the attention mask   refcoco_zsl_cpt_dataset.py:119-120 and tokenize()
The reference decodes box by box in Python; here one C call per TSV row decodes every ``feature`` string of the row
straight into a pinned ``(P, img_seq_len, 2054)`` buffer and hands json.loads a copy of the row without them.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib as L


def _line_starts(path, chunk_bytes=64 << 20):
    """Byte offset of the start of every line of `path` (numpy int64), found by scanning the file for newlines in
    bounded chunks; a final line without a newline still counts, an empty tail after the last newline does not."""
    size = os.path.getsize(path)
    pieces = [np.zeros(1, dtype=np.int64)] if size else []
    with open(path, "rb") as f:
        done = 0
        while done < size:
            blk = np.frombuffer(f.read(chunk_bytes), dtype=np.uint8)
            nl = np.flatnonzero(blk == 10).astype(np.int64)
            pieces.append(nl + (done + 1))
            done += blk.size
    starts = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int64)
    return starts[starts < size]


def generate_lineidx_file(filein, idxout):
    """Writes the `.lineidx` companion the reference reads and writes (Oscar/oscar/utils/tsv_file.py:8-18): one decimal
    byte offset per line of `filein`.  Written to a temporary name first so that a concurrent reader never sees half a file."""
    starts = _line_starts(filein)
    tmp = "%s.%d.tmp" % (idxout, os.getpid())
    with open(tmp, "w") as out:
        out.write("".join("%d\n" % o for o in starts.tolist()))
    os.replace(tmp, idxout)


class TSVFile(object):
    """Random access to the rows of a TSV file through its `.lineidx` companion: the public surface of the reference's
    reader (Oscar/oscar/utils/tsv_file.py:20-85: ``TSVFile(path, generate_lineidx)``, ``num_rows`` / ``len``, ``seek`` / ``[]``,
    ``seek_first_column``) on top of a read-only memory map and a numpy offset table, which is also what the native decoder
    wants: ``row_span(i)`` gives the byte range of a row inside ``buffer()`` so that 4 MB rows are decoded in place out of the
    page cache and never become Python ``bytes`` (``DecodePool``).  ``seek_raw`` returns the columns as ``bytes``."""

    def __init__(self, tsv_file, generate_lineidx=False):
        self.tsv_file = tsv_file
        self.lineidx = os.path.splitext(tsv_file)[0] + ".lineidx"
        self._offsets = None        # int64 [rows + 1]: row starts, then the file size
        self._ascending = True      # the .lineidx offsets increase (offsets()): the next entry bounds a row's newline search
        self._map = None
        self._map_owner = None      # the process that created the map (a forked child maps again)
        if generate_lineidx and not os.path.isfile(self.lineidx):
            generate_lineidx_file(self.tsv_file, self.lineidx)

    def __repr__(self):
        return "TSVFile(tsv_file='{}')".format(self.tsv_file)

    # ---- offsets / map -------------------------------------------------------------------------------------------
    def offsets(self):
        """int64 array of the row start offsets followed by the file size (len = num_rows + 1)."""
        if self._offsets is None:
            with open(self.lineidx, "rb") as f:
                txt = f.read().split()
            table = np.empty(len(txt) + 1, dtype=np.int64)
            if txt:
                table[:-1] = np.array(txt, dtype=np.int64)
            size = os.path.getsize(self.tsv_file)
            table[-1] = size
            starts = table[:-1]
            if len(starts) and (starts.min() < 0 or starts.max() >= max(size, 1)):
                raise ValueError("%s holds a row offset outside %s (%d bytes)" % (self.lineidx, self.tsv_file, size))
            # The reference reader (tsv_file.py:60-66: seek to the offset, readline) also serves .lineidx files that list a
            # subset of the rows or list them out of order, so a row ends at ITS newline, found on demand (row_span); an
            # ascending table only bounds that search by the next entry.
            self._ascending = bool(len(starts) == 0 or (np.diff(table) > 0).all())
            self._offsets = table
        return self._offsets

    def buffer(self):
        """The whole file as a read-only memory map (shared page cache; re-mapped in a forked worker)."""
        if self._map is None or self._map_owner != os.getpid():
            import mmap
            with open(self.tsv_file, "rb") as f:
                self._map = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if os.fstat(f.fileno()).st_size else b""
            self._map_owner = os.getpid()
        return self._map

    def row_span(self, idx):
        """(first byte, one past the last byte) of row `idx` in ``buffer()``, line terminator included."""
        offs = self.offsets()
        n = len(offs) - 1
        if not -n <= idx < n:
            raise IndexError("row %d of a %d-row TSV file" % (idx, n))
        idx %= n
        lo = int(offs[idx])
        buf = self.buffer()
        # exact for any file size and any table: memchr over one row is negligible next to its base64 decode
        hi = int(offs[idx + 1]) if self._ascending else len(buf)
        nl = buf.find(b"\n", lo, hi)
        return lo, (nl + 1 if nl >= 0 else hi)

    # ---- the reference's surface ---------------------------------------------------------------------------------
    def num_rows(self):
        return len(self.offsets()) - 1

    __len__ = num_rows

    def seek_raw(self, idx):
        lo, hi = self.row_span(idx)
        return self.buffer()[lo:hi].split(b"\t")

    def seek(self, idx):
        return [c.decode("utf-8").strip() for c in self.seek_raw(idx)]

    __getitem__ = seek

    def seek_first_column(self, idx):
        lo, hi = self.row_span(idx)
        buf = self.buffer()
        tab = buf.find(b"\t", lo, hi)
        if tab < 0:                       # a one-column row: the column without its line terminator
            return buf[lo:hi].decode("utf-8").rstrip("\r\n")
        return buf[lo:tab].decode("utf-8")


def b64_to_f32(s, dim=2054):
    """One region: np.frombuffer(base64.b64decode(s), np.float32) of refcoco_zsl_cpt_dataset.py:173."""
    if isinstance(s, str):
        s = s.encode("ascii")
    out = torch.empty(dim, dtype=torch.float32)
    L.check(L.lib().cpt_b64_decode_f32(s, len(s), out.data_ptr(), dim), "cpt_b64_decode_f32")
    return out


def _batch_decode(base_ptr, offsets, lens, counts, img_seq_len, dim, out, mask, threads):
    P = len(counts)
    if P == 0:
        return
    cnt = np.asarray(counts, dtype=np.int32)
    first = np.zeros(max(P, 1), dtype=np.int32)
    if P > 1:
        np.cumsum(cnt[:-1], out=first[1:P])
    L.check(L.lib().cpt_decode_regions_batch(base_ptr, offsets.ctypes.data, lens.ctypes.data, first.ctypes.data,
                                             cnt.ctypes.data if P else first.ctypes.data, P, dim, img_seq_len,
                                             out.data_ptr(), mask.data_ptr() if mask is not None else None, threads),
            "cpt_decode_regions_batch")


def decode_regions(feature_lists, img_seq_len=50, dim=2054, out=None, mask=None, threads=1):
    """feature_lists: per proposal sequence, the list of base64 strings of its boxes.  Returns
    (feats (P, img_seq_len, dim) f32 zero padded, mask (P, img_seq_len) int64).  ``out`` / ``mask`` may be
    preallocated (pinned) tensors."""
    P = len(feature_lists)
    if out is None:
        out = torch.empty((P, img_seq_len, dim), dtype=torch.float32)
    if mask is None:
        mask = torch.empty((P, img_seq_len), dtype=torch.int64)
    assert out.is_contiguous() and mask.is_contiguous() and tuple(out.shape) == (P, img_seq_len, dim)
    flat = [s.encode("ascii") if isinstance(s, str) else s for fl in feature_lists for s in fl]
    blob = b"".join(flat)
    lens = np.fromiter((len(s) for s in flat), dtype=np.uint64, count=len(flat))
    offsets = np.zeros(max(len(flat), 1), dtype=np.uint64)
    if len(flat) > 1:
        np.cumsum(lens[:-1], out=offsets[1:len(flat)])
    if len(flat) == 0:
        lens = np.zeros(1, dtype=np.uint64)
    buf = C.c_char_p(blob)
    _batch_decode(C.cast(buf, C.c_void_p), offsets, lens, [len(fl) for fl in feature_lists], img_seq_len, dim, out, mask, threads)
    return out, mask


def decode_row(payload, img_seq_len=50, dim=2054, out=None, mask=None, key=b"feature", threads=1):
    """One TSV payload (the JSON text of a row, bytes): returns (info, feats, mask, counts) where ``info`` is the
    parsed JSON with every feature string replaced by "", ``feats`` (P, img_seq_len, dim) the zero-padded features of
    the P box lists in ``info["objects"][0]``, ``mask`` (P, img_seq_len) and ``counts`` the box-list lengths.
    The base64 text is decoded in place from the row: it never becomes a Python string."""
    if isinstance(payload, str):
        payload = payload.encode("utf-8")
    # every value we are after is the base64 of dim float32 -> at least 16*dim/3 characters: bounds their number
    max_values = len(payload) // ((16 * dim) // 3) + 1
    offsets = np.empty(max_values, dtype=np.uint64)
    lens = np.empty(max_values, dtype=np.uint64)
    stripped = np.empty(len(payload) + 1, dtype=np.uint8)
    n_val, slen = C.c_int(0), C.c_size_t(0)
    L.check(L.lib().cpt_json_find_strings(payload, len(payload), key, offsets.ctypes.data, lens.ctypes.data, max_values,
                                          stripped.ctypes.data, len(payload) + 1, C.byref(n_val), C.byref(slen)),
            "cpt_json_find_strings")
    info = json.loads(stripped[:slen.value].tobytes())
    objs = info["objects"][0]
    counts = [len(bl) for bl in objs]
    if sum(counts) != n_val.value:
        raise RuntimeError("cpt_amd.io: %d feature strings for %d boxes" % (n_val.value, sum(counts)))
    P = len(objs)
    if out is None:
        out = torch.empty((P, img_seq_len, dim), dtype=torch.float32)
    if mask is None:
        mask = torch.empty((P, img_seq_len), dtype=torch.int64)
    buf = C.c_char_p(payload)
    _batch_decode(C.cast(buf, C.c_void_p), offsets, lens, counts, img_seq_len, dim, out[:P], mask[:P], threads)
    return info, out[:P], mask[:P], counts


def decode_rows(payloads, img_seq_len=50, dim=2054, out=None, mask=None, key=b"feature", threads=4, max_seqs=None,
                parse=True):
    """Many TSV payloads in ONE C call with native threads (cpt_decode_tsv_rows): returns
    (infos, feats (S, img_seq_len, dim), mask (S, img_seq_len), seqs_per_row, regions_per_seq) with S the total number
    of proposal sequences; sequence order = row order, then box-list order inside the row.  ``parse=False`` returns the
    stripped JSON texts instead of parsed objects (a driver may parse them on another process / later)."""
    payloads = [p.encode("utf-8") if isinstance(p, str) else p for p in payloads]
    n = len(payloads)
    rows = (C.c_char_p * max(n, 1))(*payloads)
    lens = np.fromiter((len(p) for p in payloads), dtype=np.uint64, count=n) if n else np.zeros(1, np.uint64)
    return decode_rows_at(rows, lens, n, img_seq_len, dim, out, mask, key, threads, max_seqs, parse)


def b64_chars(dim=2054):
    """Characters of one region's base64 string (cpt_b64_chars): 10956 for float32[2054]."""
    return int(L.lib().cpt_b64_chars(int(dim)))


def decode_rows_at(rows, lens, n, img_seq_len=50, dim=2054, out=None, mask=None, key=b"feature", threads=4, max_seqs=None,
                   parse=True, text=None):
    """decode_rows on row texts given by ADDRESS: ``rows`` a ctypes array of n pointers (c_char_p / c_void_p), ``lens``
    their byte lengths (uint64 array).  Lets a caller decode straight out of a memory-mapped predictions file without
    materialising 4 MB ``bytes`` objects per row (DecodePool's workers).

    ``text`` (round 5, a uint8 tensor (max_seqs, img_seq_len, b64_chars(dim))): the device-decode form -- the located strings are COPIED
    there as text (cpt_pack_tsv_rows) instead of decoded; the returned feats is then that tensor's first S sequences, to be copied to the
    GPU and decoded there (``decode_text_device``)."""
    if max_seqs is None:
        ref = text if text is not None else out
        max_seqs = ref.size(0) if ref is not None else int(sum(int(l) // ((16 * dim) // 3) + 1 for l in lens[:n]))
    if out is None and text is None:
        out = torch.empty((max_seqs, img_seq_len, dim), dtype=torch.float32)
    if text is not None:
        assert text.dtype == torch.uint8 and text.is_contiguous() and tuple(text.shape[1:]) == (img_seq_len, b64_chars(dim)) and text.size(0) >= max_seqs
        out = text
    if mask is None:
        mask = torch.empty((max_seqs, img_seq_len), dtype=torch.int64)
    assert out.is_contiguous() and mask.is_contiguous() and out.size(0) >= max_seqs and mask.size(0) >= max_seqs
    sbufs = [np.empty(int(lens[i]) + 1, dtype=np.uint8) for i in range(n)]
    sptr = np.fromiter((b.ctypes.data for b in sbufs), dtype=np.uint64, count=n) if n else np.zeros(1, np.uint64)
    scap = lens + 1
    slen = np.zeros(max(n, 1), dtype=np.uint64)
    seqs_per_row = np.zeros(max(n, 1), dtype=np.int32)
    regions = np.zeros(max(max_seqs, 1), dtype=np.int32)
    fn = L.lib().cpt_pack_tsv_rows if text is not None else L.lib().cpt_decode_tsv_rows
    L.check(fn(C.cast(rows, C.POINTER(C.c_char_p)), lens.ctypes.data, n, key, dim, img_seq_len, max_seqs, out.data_ptr(),
               mask.data_ptr(), sptr.ctypes.data, scap.ctypes.data, slen.ctypes.data,
               seqs_per_row.ctypes.data, regions.ctypes.data, threads), "cpt_pack_tsv_rows" if text is not None else "cpt_decode_tsv_rows")
    infos = [sbufs[i][:int(slen[i])].tobytes() for i in range(n)]      # the row's JSON without the feature strings
    if parse:
        infos = [json.loads(b) for b in infos]
    S = int(seqs_per_row[:n].sum())
    return infos, out[:S], mask[:S], seqs_per_row[:n].tolist(), regions[:S].tolist()


def decode_text_device(text_dev, mask_dev, out_dev, err_dev, stream=None):
    """Device half of the device decode (cpt_b64_decode_regions_device): text_dev uint8 (S, R, b64_chars(dim)) and mask_dev int64 (S, R) on the GPU
    -> out_dev float32 (S, R, dim), one launch on ``stream`` (default: the current stream).  ``err_dev``: a zero-initialised int64 device scalar that
    stays zero while every string is valid (``check_device_decode`` turns it into the host decoder's exception)."""
    S, R, dim = out_dev.shape
    assert text_dev.dtype == torch.uint8 and text_dev.is_contiguous() and tuple(text_dev.shape) == (S, R, b64_chars(dim))
    assert mask_dev.dtype == torch.int64 and mask_dev.is_contiguous() and tuple(mask_dev.shape) == (S, R)
    assert out_dev.dtype == torch.float32 and out_dev.is_contiguous() and err_dev.dtype == torch.int64 and err_dev.numel() >= 1
    st = stream if stream is not None else torch.cuda.current_stream(out_dev.device)
    L.check(L.lib().cpt_b64_decode_regions_device(text_dev.data_ptr(), mask_dev.data_ptr(), S, dim, R, out_dev.data_ptr(), err_dev.data_ptr(),
                                                  C.c_void_p(st.cuda_stream)), "cpt_b64_decode_regions_device")
    return out_dev


def check_device_decode(err_dev, img_seq_len=50):
    """Raises what the host decoder would have raised if a launch of decode_text_device met an invalid string (synchronises on err_dev)."""
    code = int(err_dev.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    if code:
        key = (~code) & 0xFFFFFFFFFFFFFFFF
        slot, ch = key >> 32, key & 0xFFFFFFFF
        raise RuntimeError("cpt_amd.io: device base64 decode: sequence %d region %d: character %d is outside the alphabet or the padding is wrong"
                           % (slot // img_seq_len, slot % img_seq_len, ch))


def decode_features(tsv, img_idx, img_seq_len=50, dim=2054):
    """decode_features of refcoco_zsl_cpt_dataset.py:161-180: (img_name, od_labels, im_feats, caption, colors,
    rect_lists); im_feats is the list of per-proposal (n_boxes, dim) tensors the reference returns (views of one
    decoded buffer)."""
    cols = tsv.seek_raw(img_idx)
    img_name = cols[0].decode("utf-8").strip()
    info, feats, _, counts = decode_row(cols[1].strip(), img_seq_len, dim)
    objs, caption, colors, rect_lists = info["objects"]
    im_feats = [feats[p, :c] for p, c in enumerate(counts)]
    od_labels = [" ".join(o["class"] for o in boxlist) for boxlist in objs]
    return img_name, od_labels, im_feats, caption, colors, rect_lists


class RegionStager(object):
    """Pinned host buffers + a side stream: decode batch i+1 on the host while batch i's features travel to the GPU
    (the 26 MB/step host buffer of DESIGN.md section 7).

    Slot k (= call number mod depth) is reused every `depth` calls, and BOTH directions of the dependency are tracked:
      copy -> consumer : stage() returns an event; the consumer stream must ``wait_event(event)`` before reading;
      consumer -> copy : before slot k's device buffer is overwritten, the side stream waits for the event the
                         consumer recorded after ITS last read of that slot -- pass the consumer stream to stage()
                         (default: the current stream) and call ``release(slot)`` once the forward that reads the
                         slot has been enqueued, or use ``stage_and_wait`` which does wait + release bookkeeping for
                         the common "one forward per staged batch" loop.  Without that event a fast host could
                         overwrite batch i's features while pad_cast of batch i is still reading them."""

    def __init__(self, max_seqs, img_seq_len=50, dim=2054, device="cuda:0", depth=2, threads=4):
        self.dev = torch.device(device)
        self.shape = (max_seqs, img_seq_len, dim)
        self.threads = threads
        self.host = [torch.empty(self.shape, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.hmask = [torch.empty(self.shape[:2], dtype=torch.int64).pin_memory() for _ in range(depth)]
        self.devb = [torch.empty(self.shape, dtype=torch.float32, device=self.dev) for _ in range(depth)]
        self.dmask = [torch.empty(self.shape[:2], dtype=torch.int64, device=self.dev) for _ in range(depth)]
        self.stream = torch.cuda.Stream(self.dev)
        self.events = [None] * depth          # H2D copy of slot k done
        self.consumed = [None] * depth        # consumer's reads of slot k done
        self.unreleased = [False] * depth
        self.i = 0
        self.last_slot = None

    def release(self, slot=None, stream=None):
        """Record "the consumer has finished reading `slot`" on the consumer stream (call after enqueueing the forward)."""
        k = self.last_slot if slot is None else slot
        if k is None:
            return
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.dev))
        self.consumed[k] = ev
        self.unreleased[k] = False

    def stage(self, feature_lists, consumer_stream=None):
        """Decode + enqueue the H2D copy; returns (feats_dev, mask_dev, event).  The consumer stream must
        ``wait_event(event)`` before reading and ``release()`` the slot after its reads are enqueued."""
        k = self.i % len(self.host)
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()              # the previous copy out of this pinned buffer is done
        if self.unreleased[k]:
            # the consumer never told us when it finished with this slot: fall back to "everything enqueued on its
            # stream so far" (safe, possibly later than necessary)
            self.release(k, consumer_stream)
        P = len(feature_lists)
        decode_regions(feature_lists, self.shape[1], self.shape[2], self.host[k][:P], self.hmask[k][:P], self.threads)
        with torch.cuda.stream(self.stream):
            if self.consumed[k] is not None:
                self.stream.wait_event(self.consumed[k])      # write-after-read: the model is done with this slot
            self.devb[k][:P].copy_(self.host[k][:P], non_blocking=True)
            self.dmask[k][:P].copy_(self.hmask[k][:P], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[k] = ev
        self.unreleased[k] = True
        self.last_slot = k
        return self.devb[k][:P], self.dmask[k][:P], ev

    def stage_and_wait(self, feature_lists, stream=None):
        """stage() + make `stream` (default: current) wait for the copy; the caller enqueues its forward and then calls
        ``release()``."""
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        feats, mask, ev = self.stage(feature_lists, st)
        st.wait_event(ev)
        return feats, mask


# ---- worker processes + shared pinned ring -------------------------------------------------------------------------
# The reference feeds its model from torch DataLoader worker PROCESSES (zeroshot/refcoco_cpt.py: DataLoader(num_workers=...)),
# each decoding rows in Python and pickling (P, 50, 2054) tensors back to the main process.  One Python process cannot keep
# up with the HIP forward (decode + json + launch contend for the interpreter: 9 k sequences/s, profiles/r01_io_decode.json),
# so the decode moves to worker processes here too -- but they write the decoded features straight into a ring of SHARED,
# PINNED host buffers (one batch per slot): nothing is pickled but the rows' small JSON, and the main process only issues the
# H2D copy of a finished slot and the forward.

def _pool_worker(tsv_path, img_seq_len, dim, threads, feats, masks, tasks, results, my_slots):
    as_text = feats.dtype == torch.uint8      # device-decode ring: the slots hold base64 text
    for k in my_slots:                  # map this worker's slots now: first-touch page faults stay out of the steady state
        feats[k].zero_()
        masks[k].zero_()
    tsv = TSVFile(tsv_path)
    mm = tsv.buffer()                                             # rows are decoded in place out of the page cache
    base = np.frombuffer(mm, dtype=np.uint8).ctypes.data
    while True:
        job = tasks.get()
        if job is None:
            return
        ticket, slot, rows = job
        try:
            n = len(rows)
            ptrs = (C.c_void_p * max(n, 1))()
            lens = np.zeros(max(n, 1), dtype=np.uint64)
            names = []
            for j, i in enumerate(rows):
                lo, hi = tsv.row_span(i)                          # (IndexError for a row that does not exist: reported below)
                tab = mm.find(b"\t", lo, hi)
                if tab < 0:
                    raise ValueError("row %d has no tab-separated payload" % i)
                names.append(mm[lo:tab].decode("utf-8").strip())
                lo = tab + 1                                      # (two-column rows, as inference_ref.py writes them: name \t json;
                #  scanning 4 MB for a further tab with mmap.find costs more than the decode itself)
                while hi > lo and mm[hi - 1:hi] in (b" ", b"\t", b"\r", b"\n"):
                    hi -= 1
                while lo < hi and mm[lo:lo + 1] in (b" ", b"\t", b"\r", b"\n"):
                    lo += 1
                if hi - lo < 2 or mm[hi - 1:hi] != b"}":
                    raise ValueError("row %d: the payload column is not one JSON object" % i)
                ptrs[j] = base + lo
                lens[j] = hi - lo
            infos, f, m, seqs_per_row, regions = decode_rows_at(ptrs, lens, n, img_seq_len, dim, out=None if as_text else feats[slot], mask=masks[slot],
                                                                threads=threads, max_seqs=feats.size(1), parse=False, text=feats[slot] if as_text else None)
            results.put((ticket, slot, names, infos, seqs_per_row, regions, None))
        except Exception as e:          # the main process re-raises
            results.put((ticket, slot, None, None, None, None, "%s: %s" % (type(e).__name__, e)))


class DecodePool(object):
    """``workers`` processes decode whole batches of TSV rows into a ring of ``slots`` shared pinned buffers.

        pool = DecodePool(tsv_path, max_seqs=64, workers=4, slots=6)
        for rows in batches: pool.submit(rows)                 # row indices of one batch; up to `slots` in flight
        slot, names, infos, seqs_per_row, regions = pool.next()  # in submission order
        feats, mask = pool.feats[slot][:S], pool.masks[slot][:S]   # pinned host views: copy H2D, then pool.release(slot)

    ``infos`` are the rows' JSON texts WITHOUT the feature strings (bytes; json.loads them where the captions are needed,
    e.g. prompts.PromptBuilder).  A slot is reused only after ``release(slot)``: call it once the H2D copy out of the slot
    has completed (e.g. after the copy event's synchronize())."""

    def __init__(self, tsv_path, max_seqs, img_seq_len=50, dim=2054, workers=4, slots=None, threads=2, pin=None, device_decode=False):
        """device_decode (round 5): the ring holds the regions' base64 TEXT (uint8 (slots, max_seqs, img_seq_len, b64_chars(dim))): the workers only
        locate and copy the strings (cpt_pack_tsv_rows); the consumer copies a slot to the GPU and decodes it there (``decode_text_device``)."""
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")               # never fork a process that holds a HIP context
        self.slots = slots or 2 * workers
        if self.slots < workers:
            raise ValueError("DecodePool: at least one slot per worker")
        self.device_decode = bool(device_decode)
        if self.device_decode:
            self.feats = torch.empty((self.slots, max_seqs, img_seq_len, b64_chars(dim)), dtype=torch.uint8).share_memory_()
        else:
            self.feats = torch.empty((self.slots, max_seqs, img_seq_len, dim), dtype=torch.float32).share_memory_()
        self.masks = torch.empty((self.slots, max_seqs, img_seq_len), dtype=torch.int64).share_memory_()
        self.pinned = False
        if pin is None:
            pin = torch.cuda.is_available()
        if pin:                                      # page-lock the shared pages for the DMA engine (hipHostRegister)
            rt = torch.cuda.cudart()
            for t in (self.feats, self.masks):
                rc = rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
                if int(rc) != 0:
                    raise RuntimeError("cpt_amd.io: hipHostRegister of the shared ring failed (%s)" % rc)
            self.pinned = True
        # slot k always belongs to worker k % workers (own task queue): a process only ever maps and touches its own slots
        self.workers = workers
        self.tasks, self.results = [ctx.Queue() for _ in range(workers)], ctx.Queue()
        self.procs = [ctx.Process(target=_pool_worker, args=(tsv_path, img_seq_len, dim, threads, self.feats, self.masks,
                                                            self.tasks[w], self.results, list(range(w, self.slots, workers))),
                                  daemon=True) for w in range(workers)]
        for p in self.procs:
            p.start()
        self.free = list(range(self.slots))
        self.ticket = self.next_ticket = 0
        self.done = {}

    def can_submit(self):
        return bool(self.free)

    def submit(self, rows):
        if not self.free:
            raise RuntimeError("cpt_amd.io.DecodePool: every slot is in flight; release() one first")
        # prefer a slot of the worker with the fewest batches in flight
        busy = [0] * self.workers
        for k in range(self.slots):
            if k not in self.free:
                busy[k % self.workers] += 1
        slot = min(self.free, key=lambda k: (busy[k % self.workers], k))
        self.free.remove(slot)
        self.tasks[slot % self.workers].put((self.ticket, slot, list(rows)))
        self.ticket += 1
        return slot

    def next(self, timeout=120.0):
        """Result of the oldest unreturned batch: (slot, names, infos, seqs_per_row, regions_per_seq)."""
        while self.next_ticket not in self.done:
            t, slot, names, infos, spr, regions, err = self.results.get(timeout=timeout)
            self.done[t] = (slot, names, infos, spr, regions, err)
        slot, names, infos, spr, regions, err = self.done.pop(self.next_ticket)
        self.next_ticket += 1
        if err is not None:
            self.free.append(slot)
            raise RuntimeError("cpt_amd.io.DecodePool worker: " + err)
        return slot, names, infos, spr, regions

    def release(self, slot):
        self.free.append(slot)

    def close(self):
        for q in self.tasks:
            q.put(None)
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        if self.pinned:
            rt = torch.cuda.cudart()
            for t in (self.feats, self.masks):
                rt.cudaHostUnregister(t.data_ptr())
            self.pinned = False
        self.procs = []

    def __del__(self):
        try:
            if self.procs:
                self.close()
        except Exception:
            pass
