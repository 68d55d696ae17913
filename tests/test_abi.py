"""The C-ABI library loads on a CPU-only box and exports every symbol include/filtlong_hip.h declares
(no compute calls here: there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

from filtlong_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "filtlong_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_and_loader_agree():
    assert header_functions() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), "missing export: " + name


def test_version_and_layout_helpers(lib):
    L = _lib.load()
    assert L.flx_abi_version() == 2
    assert b"gfx950" in L.flx_version()
    import numpy as np
    from filtlong_amd import api
    plane, offsets, lengths = api.pack_reads([b"ABC", b"", b"x" * 16, b"y" * 17])
    assert list(offsets) == [0, 16, 16, 32] and plane.nbytes == 64 and list(lengths) == [3, 0, 16, 17]
    assert bytes(plane[:3]) == b"ABC" and bytes(plane[32:49]) == b"y" * 17
    assert list(api.length_order(np.array([5, 9, 9, 1], dtype=np.int32))) == [1, 2, 0, 3]


def test_no_gpu_fails_loudly():
    """Without a gfx950 device the context must refuse (no silent CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from filtlong_amd import api
    with pytest.raises(api.FlxError):
        api.Context(0)
