"""CPU: the C-ABI library builds, loads, and exports every entry point include/*.h declares
(no compute calls without a GPU); the ctypes binding covers the same set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            src = open(os.path.join(ROOT, "include", h)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(cpt_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_symbols_exported():
    from cpt_amd import build, _lib
    build.build(verbose=False)
    names = _declared()
    assert len(names) >= 15
    lib = _lib.lib()                                  # dlopen: every undefined symbol would fail here
    for n in names:
        assert hasattr(lib, n), "libcpt_hip.so does not export %s" % n
    assert set(_lib.exported_symbols()) == set(names), set(_lib.exported_symbols()) ^ set(names)
    assert lib.cpt_version() == 8
    assert isinstance(lib.cpt_last_error(), bytes)


def test_struct_layouts_match_header():
    from cpt_amd import _lib as L
    assert ctypes.sizeof(L.Dims) == 12 * 4 + 2 * 4
    assert ctypes.sizeof(L.Layer) == 12 * 8
    assert ctypes.sizeof(L.Batch) == 3 * 4 + 4 + 7 * 8 + 4 + 4 + 8     # 3 ints, padding, 7 pointers, n_rows, padding, row_seq (ABI 5)
    assert ctypes.sizeof(L.Outputs) == 8 * 8
    assert L.Model.layers.offset == ctypes.sizeof(L.Dims) + 9 * 8


def test_product_path_has_no_cpu_fallback():
    """Model on CPU tensors must raise, and nothing under cpt_amd imports the oracle."""
    import torch
    from cpt_amd import config as cfgmod
    from cpt_amd.modeling_rec import REC_MLM_CPT
    m = REC_MLM_CPT(cfgmod.tiny())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, dtype=torch.long))
    for dp, _, files in os.walk(os.path.join(ROOT, "cpt_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# noqa", ""), "%s mentions the oracle" % f
