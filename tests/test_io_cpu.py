"""CPU: the region-feature wire-format decoder (include/cpt_io.h, cpt_amd/io.py) against the
fixture decoded by the reference's own TSVFile + decode_features, against the Python restatement (oracle/io_oracle.py)
on random inputs, and on the edge cases (empty / full box lists, bad characters, wrong sizes)."""
import base64
import json
import os
import shutil

import numpy as np
import pytest
import torch

from cpt_amd import io
from oracle import io_oracle as IO


def _b64(a):
    return base64.b64encode(np.asarray(a, np.float32).tobytes()).decode("ascii")


def test_oracle_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_rows_expected.npz"))
    tsv, idx = os.path.join(golden_dir, "tiny_rows.tsv"), os.path.join(golden_dir, "tiny_rows.lineidx")
    for i in range(int(g["n_rows"])):
        name, od, feats, caption, colors, rects = IO.decode_features(IO.tsv_seek(tsv, idx, i))
        assert name == str(g["r%d_name" % i]) and caption == str(g["r%d_caption" % i])
        assert od == list(g["r%d_od_labels" % i])
        assert [f.size(0) for f in feats] == list(g["r%d_counts" % i])
        assert np.array_equal(torch.cat(feats, 0).numpy(), g["r%d_feats" % i])


def test_decoder_matches_reference_fixture(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "tiny_rows_expected.npz"))
    src = os.path.join(golden_dir, "tiny_rows.tsv")
    t = io.TSVFile(src)
    assert len(t) == int(g["n_rows"])
    for i in range(len(t)):
        name, od, feats, caption, colors, rects = io.decode_features(t, i, img_seq_len=6)
        assert name == str(g["r%d_name" % i]) and caption == str(g["r%d_caption" % i])
        assert od == list(g["r%d_od_labels" % i])
        assert [f.size(0) for f in feats] == list(g["r%d_counts" % i])
        assert np.array_equal(torch.cat(feats, 0).numpy(), g["r%d_feats" % i])          # bit exact
        assert t[i][0] == IO.tsv_seek(src, t.lineidx, i)[0] and t.seek_first_column(i) == name
        # padded form + image part of the attention mask == the reference's torch.cat with zeros
        cols = t.seek_raw(i)
        info, padded, mask, counts = io.decode_row(cols[1].strip(), img_seq_len=6)
        ref_p, ref_m = IO.pad_regions(IO.decode_features(IO.tsv_seek(src, t.lineidx, i))[2], 6)
        assert torch.equal(padded, ref_p) and torch.equal(mask, ref_m)
        assert all(o["feature"] == "" for bl in info["objects"][0] for o in bl)
    # .lineidx generation == the one the reference's generate_lineidx_file wrote into the fixture
    cp = str(tmp_path / "copy.tsv")
    shutil.copy(src, cp)
    t2 = io.TSVFile(cp, generate_lineidx=True)
    assert open(t2.lineidx).read() == open(t.lineidx).read()


@pytest.mark.parametrize("threads", [1, 3])
def test_decode_regions_random_ragged(threads):
    rng = np.random.default_rng(5)
    counts = [0, 1, 50, 7, 50, 0, 13]
    data = [[rng.standard_normal(2054).astype(np.float32) for _ in range(c)] for c in counts]
    data[2][3][:5] = [np.nan, np.inf, -np.inf, -0.0, np.float32(1e-45)]          # every bit pattern survives
    lists = [[_b64(a) for a in seq] for seq in data]
    feats, mask = io.decode_regions(lists, img_seq_len=50, threads=threads)
    ref_f, ref_m = IO.pad_regions([torch.Tensor(np.stack(s)) if s else torch.zeros(0, 2054) for s in data], 50)
    assert feats.numpy().tobytes() == ref_f.numpy().tobytes()
    assert torch.equal(mask, ref_m)
    # other dims / paddings of the base64 tail ('=' and '==')
    for dim in (1, 2, 3, 5, 38):
        a = rng.standard_normal(dim).astype(np.float32)
        assert io.b64_to_f32(_b64(a), dim).numpy().tobytes() == a.tobytes()
    assert io.decode_regions([], img_seq_len=4)[0].shape == (0, 4, 2054)


def test_decoder_rejects_bad_input():
    good = _b64(np.arange(2054))
    with pytest.raises(RuntimeError, match="outside the alphabet"):
        io.b64_to_f32(good[:100] + "!" + good[101:])
    with pytest.raises(RuntimeError, match="does not decode to 2054"):
        io.b64_to_f32(_b64(np.arange(2053)))
    with pytest.raises(RuntimeError, match="does not decode"):
        io.b64_to_f32(good[:-1])
    with pytest.raises(RuntimeError, match="do not fit max_regions"):
        io.decode_regions([[good] * 5], img_seq_len=4)
    with pytest.raises(RuntimeError, match="outside the alphabet"):          # error raised inside a worker thread
        io.decode_regions([[good], [good], [good[:7] + "\n" + good[8:]], [good]], img_seq_len=2, threads=4)
    row = json.dumps({"objects": [[[{"class": "a", "feature": good}]], "c", [["red"]], [[[0, 0, 1, 1]]]]})
    with pytest.raises(RuntimeError, match="unterminated"):
        io.decode_row(row[:-40].encode() if False else (row[:row.index(good) + 50]).encode())
    info, feats, mask, counts = io.decode_row(row.encode(), img_seq_len=3)
    assert counts == [1] and feats[0, 0, 7].item() == 7.0 and mask.tolist() == [[1, 0, 0]]


@pytest.mark.parametrize("threads", [1, 4])
def test_decode_rows_native_batch(golden_dir, threads):
    """cpt_decode_tsv_rows: whole rows in one call, values grouped by their enclosing box list (empty lists kept)."""
    g = np.load(os.path.join(golden_dir, "tiny_rows_expected.npz"))
    t = io.TSVFile(os.path.join(golden_dir, "tiny_rows.tsv"))
    payloads = [t.seek_raw(i)[1].strip() for i in range(len(t))]
    rng = np.random.default_rng(3)
    extra = [[rng.standard_normal(2054).astype(np.float32) for _ in range(c)] for c in (0, 2, 0, 1, 0)]
    row3 = json.dumps({"objects": [[[{"rect": [[1, 2], [3, 4]], "class": "x", "feature": _b64(a)} for a in bl] for bl in extra],
                                   "caption [with] brackets ]", [["red"]] * 5, [[[0, 0, 1, 1]]] * 5]})
    infos, feats, mask, seqs_per_row, regions = io.decode_rows(payloads + [row3], img_seq_len=6, threads=threads)
    assert seqs_per_row == [3, 2, 5]
    assert regions == list(g["r0_counts"]) + list(g["r1_counts"]) + [0, 2, 0, 1, 0]
    ref = np.concatenate([g["r0_feats"], g["r1_feats"]] + [np.stack(bl) for bl in extra if bl])
    got = np.concatenate([feats[s, :c].numpy() for s, c in enumerate(regions) if c])
    assert got.tobytes() == ref.tobytes()
    assert mask.sum(1).tolist() == regions and float(feats[5, :].abs().max()) == 0.0
    for s, c in enumerate(regions):
        assert float(feats[s, c:].abs().max() if c < 6 else 0.0) == 0.0
    assert infos[0]["objects"][1] == str(g["r0_caption"]) and infos[2]["objects"][1] == "caption [with] brackets ]"
    with pytest.raises(RuntimeError, match="do not fit max_seqs"):
        io.decode_rows(payloads, img_seq_len=6, max_seqs=4)
    with pytest.raises(RuntimeError, match="do not fit max_regions"):
        io.decode_rows(payloads, img_seq_len=4)


def test_decode_pool_worker_processes_match_direct_decode(golden_dir):
    """Section 8(f).2: worker processes decoding batches of rows into the shared ring give, slot by slot and in submission
    order, exactly what the in-process decoder gives for the same rows."""
    import json
    tsv_path = os.path.join(golden_dir, "tiny_prompt_rows.tsv")
    tsv = io.TSVFile(tsv_path)
    n = tsv.num_rows()
    pool = io.DecodePool(tsv_path, max_seqs=10, img_seq_len=50, workers=2, slots=3, threads=1, pin=False)
    try:
        batches = [[0, 1], [2], [1, 2, 0], [0], [2, 1]]
        got = []
        it = iter(batches)
        pending = 0
        for rows in it:
            while not pool.can_submit():
                slot, names, infos, spr, regions = pool.next()
                S = sum(spr)
                got.append((names, [json.loads(b) for b in infos], pool.feats[slot][:S].clone(), pool.masks[slot][:S].clone(), spr, regions))
                pool.release(slot)
                pending -= 1
            pool.submit(rows)
            pending += 1
        while pending:
            slot, names, infos, spr, regions = pool.next()
            S = sum(spr)
            got.append((names, [json.loads(b) for b in infos], pool.feats[slot][:S].clone(), pool.masks[slot][:S].clone(), spr, regions))
            pool.release(slot)
            pending -= 1
        assert len(got) == len(batches)
        for rows, (names, infos, f, m, spr, regions) in zip(batches, got):
            cols = [tsv.seek_raw(i) for i in rows]
            assert names == [c[0].decode().strip() for c in cols]
            ref_infos, rf, rm, rspr, rreg = io.decode_rows([c[1].strip() for c in cols], 50, max_seqs=10)
            assert spr == rspr and regions == rreg and infos == ref_infos
            assert torch.equal(f, rf) and torch.equal(m, rm)
        with pytest.raises(RuntimeError):
            pool.submit([n + 5])
            pool.next()
    finally:
        pool.close()


def test_tsvfile_subset_and_reordered_lineidx(tmp_path):
    """the reference reader (utils/tsv_file.py:60-66: seek to the offset, readline) also serves .lineidx files that list a
    subset of the rows or list them out of order; a row then ends at its newline, not at the next table entry."""
    from cpt_amd.io import TSVFile
    rows = [("k%d" % i, "payload %d." % i + "x" * (3 * i)) for i in range(6)]
    tsv = tmp_path / "t.tsv"
    tsv.write_text("".join("%s\t%s\n" % r for r in rows) + "lonely\n")
    full = TSVFile(str(tsv), generate_lineidx=True)
    assert len(full) == 7 and full._ascending
    assert [full.seek(i) for i in range(6)] == [list(r) for r in rows]
    assert full.seek_first_column(6) == "lonely" and full.seek(6) == ["lonely"]
    offs = [int(o) for o in (tmp_path / "t.lineidx").read_text().split()]
    for name, pick in (("subset", [0, 2, 5]), ("reordered", [4, 1, 3, 0])):
        sub = tmp_path / (name + ".tsv")
        sub.write_bytes(tsv.read_bytes())
        (tmp_path / (name + ".lineidx")).write_text("".join("%d\n" % offs[i] for i in pick))
        t = TSVFile(str(sub))
        assert len(t) == len(pick)
        assert [t.seek(j) for j in range(len(pick))] == [list(rows[i]) for i in pick]
        assert [t.seek_first_column(j) for j in range(len(pick))] == [rows[i][0] for i in pick]
    # an ascending subset that starts at row 0 and skips rows in between (what a sampled check of a tens-of-GB file
    # would have classified as "every row listed"): the skipped lines must not be merged into the row in front of them
    sub = tmp_path / "skips.tsv"
    sub.write_bytes(tsv.read_bytes())
    (tmp_path / "skips.lineidx").write_text("".join("%d\n" % offs[i] for i in [0, 1, 4, 6]))
    t = TSVFile(str(sub))
    assert [t.seek(j) for j in range(3)] == [list(rows[i]) for i in [0, 1, 4]] and t.seek(3) == ["lonely"]
    assert [t.row_span(j)[1] - t.row_span(j)[0] for j in range(3)] == [len("%s\t%s\n" % rows[i]) for i in [0, 1, 4]]
    bad = tmp_path / "bad.tsv"
    bad.write_bytes(tsv.read_bytes())
    (tmp_path / "bad.lineidx").write_text("0\n99999\n")
    with pytest.raises(ValueError):
        TSVFile(str(bad)).seek(0)


def test_base64_vector_and_scalar_loops_agree(tmp_path):
    """Round 5: the AVX2 base64 loop (csrc/b64_avx2.cpp, taken when the CPU has AVX2) and the table-driven scalar loop (CPT_B64_SCALAR=1) decode the same
    bits and report the same error positions: random float32 payloads incl. NaN / Inf / denormal patterns, every length class of the 32-character blocks,
    '=' padding, and a character outside the alphabet at the first / a middle / the last position of a block."""
    import subprocess
    import sys
    script = r'''
import base64, ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
from cpt_amd import _lib as L
lib = L.lib()
rng = np.random.default_rng(5)
out = []
for dim in (1, 2, 5, 6, 7, 8, 23, 24, 25, 48, 2054):
    x = rng.standard_normal(dim).astype(np.float32)
    special = np.array([0x7fc00000, 0x7f800000, 0xff800000, 0x00000001], dtype=np.uint32)
    x.view(np.uint32)[: min(dim, 4)] = special[: min(dim, 4)]
    if dim > 40:
        x.view(np.uint32)[dim - 4:] = special
    s = base64.b64encode(x.tobytes())
    got = np.empty(dim, np.float32)
    rc = lib.cpt_b64_decode_f32(s, len(s), got.ctypes.data, dim)
    out.append([dim, rc, got.view(np.uint32).tolist() == x.view(np.uint32).tolist()])
    for pos in (0, len(s) // 2, max(len(s) - 5, 0), 33 if len(s) > 40 else 1):
        bad = bytearray(s)
        bad[pos] = ord("*")
        rc = lib.cpt_b64_decode_f32(bytes(bad), len(bad), got.ctypes.data, dim)
        out.append([dim, pos, rc, lib.cpt_last_error().decode()])
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for scalar in ("0", "1"):
        env = dict(os.environ, CPT_B64_SCALAR=scalar)
        p = subprocess.run([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[scalar] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["0"] == res["1"]
    ok = [r for r in res["0"] if len(r) == 3]
    assert ok and all(r[1] == 0 and r[2] for r in ok)
    errs = [r for r in res["0"] if len(r) == 4]
    assert errs and all(r[2] != 0 and ("character %d " % (r[1] // 4 * 4)) in r[3] for r in errs)      # (the decoder names the 4-character group of the offender)


@pytest.mark.parametrize("threads", [1, 4])
def test_pack_rows_for_the_device_decoder(golden_dir, threads):
    """Round 5, host half of the device decode (cpt_pack_tsv_rows): the located strings arrive as TEXT in [sequence][region][b64_chars] slots, with
    the masks, counts and stripped JSON of cpt_decode_tsv_rows; decoding the packed text with Python's base64 gives the reference's features."""
    g = np.load(os.path.join(golden_dir, "tiny_rows_expected.npz"))
    t = io.TSVFile(os.path.join(golden_dir, "tiny_rows.tsv"))
    payloads = [t.seek_raw(i)[1].strip() for i in range(len(t))]
    n = len(payloads)
    import ctypes as C
    rows = (C.c_char_p * n)(*payloads)
    lens = np.fromiter((len(p) for p in payloads), dtype=np.uint64, count=n)
    chars = io.b64_chars(2054)
    assert chars == 10956 and io.b64_chars(1) == 8 and io.b64_chars(3) == 16
    text = torch.full((8, 6, chars), 0x21, dtype=torch.uint8)          # '!': a slot that is not written stays invalid text
    infos, packed, mask, seqs_per_row, regions = io.decode_rows_at(rows, lens, n, img_seq_len=6, threads=threads, text=text)
    ref_infos, feats, ref_mask, ref_spr, ref_regions = io.decode_rows(payloads, img_seq_len=6, threads=threads)
    assert seqs_per_row == ref_spr and regions == ref_regions and torch.equal(mask, ref_mask) and infos == ref_infos
    assert packed.data_ptr() == text.data_ptr() and packed.shape == (sum(seqs_per_row), 6, chars)
    for s, c in enumerate(regions):
        for i in range(6):
            raw = bytes(packed[s, i].numpy())
            if i < c:
                assert np.frombuffer(base64.b64decode(raw), np.float32).tobytes() == feats[s, i].numpy().tobytes()
            else:
                assert raw == b"!" * chars
    # a value of another length is the host decoder's size error
    short = json.dumps({"objects": [[[{"class": "a", "feature": _b64(np.arange(2053))}]], "c", [["red"]], [[[0, 0, 1, 1]]]]}).encode()
    rows1 = (C.c_char_p * 1)(short)
    with pytest.raises(RuntimeError, match="does not decode to 2054"):
        io.decode_rows_at(rows1, np.array([len(short)], np.uint64), 1, img_seq_len=6, threads=threads, text=text)
    with pytest.raises(RuntimeError, match="do not fit max_regions"):
        io.decode_rows_at(rows, lens, n, img_seq_len=4, threads=threads, text=torch.zeros((8, 4, chars), dtype=torch.uint8))
