"""GPU: the device half of the region-feature wire format (include/cpt_io.h cpt_b64_decode_regions_device):
base64 text decoded on the GPU, bit for bit what the host decoder (and Python's base64 + np.frombuffer, refcoco_zsl_cpt_dataset.py:173) gives."""
import base64
import json
import os

import numpy as np
import pytest
import torch

from cpt_amd import io

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _b64(a):
    return base64.b64encode(np.asarray(a, np.float32).tobytes())


def _text(lists, R, dim):
    chars = io.b64_chars(dim)
    t = torch.full((len(lists), R, chars), 0x21, dtype=torch.uint8)
    m = torch.zeros((len(lists), R), dtype=torch.int64)
    for s, seq in enumerate(lists):
        for i, a in enumerate(seq):
            t[s, i] = torch.frombuffer(bytearray(_b64(a)), dtype=torch.uint8)
            m[s, i] = 1
    return t, m


@pytest.mark.parametrize("dim", [2054, 1, 2, 3, 5, 6, 38, 2052])
def test_device_decode_equals_host_decode(dev, dim):
    """Every tail form of the base64 string (4 dim mod 3 = 0 / 1 / 2: no padding, '==', '='), ragged sequences (0 .. R regions: the other slots hold
    invalid text and must come out as zero rows), special bit patterns."""
    rng = np.random.default_rng(dim)
    R = 7
    counts = [0, 1, R, 3, R, 0, 5]
    data = [[rng.standard_normal(dim).astype(np.float32) for _ in range(c)] for c in counts]
    data[2][3][:1] = [np.nan]
    data[4][0][-1:] = [-0.0]
    if dim >= 5:
        data[2][1][:5] = [np.inf, -np.inf, np.float32(1e-45), np.float32(3.4e38), np.float32(-1e-38)]
    t, m = _text(data, R, dim)
    ref = torch.zeros((len(counts), R, dim))
    for s, seq in enumerate(data):
        for i, a in enumerate(seq):
            ref[s, i] = torch.from_numpy(a)
    host, hmask = io.decode_regions([[_b64(a).decode() for a in seq] for seq in data], img_seq_len=R, dim=dim)
    assert host.numpy().tobytes() == ref.numpy().tobytes() and torch.equal(hmask, m)
    out = torch.full((len(counts), R, dim), 7.0, device=dev)
    err = torch.zeros(1, dtype=torch.int64, device=dev)
    io.decode_text_device(t.to(dev), m.to(dev), out, err)
    io.check_device_decode(err, R)
    assert out.cpu().numpy().tobytes() == ref.numpy().tobytes()


def test_device_decode_reports_the_first_invalid_character(dev):
    dim, R = 2054, 3
    rng = np.random.default_rng(1)
    data = [[rng.standard_normal(dim).astype(np.float32) for _ in range(R)] for _ in range(4)]
    t, m = _text(data, R, dim)
    chars = io.b64_chars(dim)
    cases = [((2, 1, 5000), ord("!")), ((1, 2, 16 * 684 + 3), ord("\n")), ((3, 0, chars - 1), ord("A")),        # a character of the last whole group; the '=' missing
             ((0, 2, 100), ord("=")), ((1, 0, chars - 2), ord("="))]                                            # '=' inside the string; one '=' too many
    for (s, i, ch), c in cases:
        bad = t.clone()
        bad[s, i, ch] = c
        bad[3, 2, 7777] = ord("?")                                            # a LATER slot's error never wins
        err = torch.zeros(1, dtype=torch.int64, device=dev)
        out = torch.empty((4, R, dim), device=dev)
        io.decode_text_device(bad.to(dev), m.to(dev), out, err)
        with pytest.raises(RuntimeError, match="sequence %d region %d: character %d " % (s, i, ch)):
            io.check_device_decode(err, R)
        # the host decoder refuses the same string
        with pytest.raises(RuntimeError):
            io.b64_to_f32(bytes(bad[s, i].numpy()).decode("latin1"), dim)
    # a masked-out slot is never read as text
    m2 = m.clone()
    m2[2, 1] = 0
    bad = t.clone()
    bad[2, 1, 5000] = ord("!")
    err = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.empty((4, R, dim), device=dev)
    io.decode_text_device(bad.to(dev), m2.to(dev), out, err)
    io.check_device_decode(err, R)
    assert float(out[2, 1].abs().max()) == 0.0


def test_device_decode_pool_matches_host_pool(dev, golden_dir):
    """Worker processes that only pack text (DecodePool(device_decode=True)) + the device decoder == the host-decoding pool, batch by batch."""
    tsv_path = os.path.join(golden_dir, "tiny_prompt_rows.tsv")
    batches = [[0, 1], [2], [1, 2, 0], [0]]
    res = {}
    for mode in (False, True):
        pool = io.DecodePool(tsv_path, max_seqs=10, img_seq_len=50, workers=2, slots=3, threads=1, device_decode=mode)
        try:
            got = []
            err = torch.zeros(1, dtype=torch.int64, device=dev)
            for rows in batches:
                pool.submit(rows)
                slot, names, infos, spr, regions = pool.next()
                S = sum(spr)
                if mode:
                    assert pool.feats.dtype == torch.uint8
                    out = torch.empty((S, 50, 2054), device=dev)
                    io.decode_text_device(pool.feats[slot][:S].to(dev), pool.masks[slot][:S].to(dev), out, err)
                    f = out.cpu()
                else:
                    f = pool.feats[slot][:S].clone()
                got.append((names, infos, spr, regions, f, pool.masks[slot][:S].clone()))
                pool.release(slot)
            io.check_device_decode(err, 50)
            res[mode] = got
        finally:
            pool.close()
    for a, b in zip(res[False], res[True]):
        assert a[:4] == b[:4] and torch.equal(a[5], b[5])
        assert a[4].numpy().tobytes() == b[4].numpy().tobytes()
