r"""CPU: libbbhip.so loads and exports every symbol include/bbhip.h declares; the product
refuses to run without a GPU instead of falling back."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def _declared():
    text = (REPO / "include" / "bbhip.h").read_text()
    return sorted(set(re.findall(r"\b(bbh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    so = REPO / "bblean_amd" / "libbbhip.so"
    assert so.is_file(), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(str(so))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_python_prototypes_cover_header():
    from bblean_amd import _lib

    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared()


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bblean_amd import BitBirch, _lib
    from bblean_amd.similarity import jt_sim_packed

    with pytest.raises(_lib.BBHipError):
        jt_sim_packed(np.zeros((4, 256), np.uint8), np.zeros(256, np.uint8))
    with pytest.raises(_lib.BBHipError):
        BitBirch().fit(np.zeros((4, 256), np.uint8))
