"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/t2amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "t2amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t2_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tacotron2_subword_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n


def test_binding_lists_every_declared_symbol():
    from tacotron2_subword_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_layout_query_and_error_path_without_gpu():
    from tacotron2_subword_amd import _lib as L
    from oracle import tacotron2_oracle as O
    hp = O.default_hparams()
    dims = L.dims_from_hparams(hp)
    lay = L.decoder_layout(dims, 64, 400, 100, 60)
    assert lay.total_floats * 4 < 8 << 30
    assert lay.din % 4 == 0 and lay.dout % 4 == 0
    bad = L.dims_from_hparams(dict(hp, prenet_dim=100))
    out = L.DecoderLayout()
    rc = L.lib().t2_decoder_layout_query(ctypes.byref(bad), 2, 3, 4, 5, ctypes.byref(out))
    assert rc != 0 and b"multiples of 64" in L.lib().t2_last_error()


def test_no_cpu_fallback():
    """CPU tensors are refused loudly."""
    import pytest
    import torch
    from tacotron2_subword_amd import _lib as L
    with pytest.raises(RuntimeError):
        L.ptr(torch.zeros(4))
