"""CPU: the C-ABI shared library loads and exports every symbol include/vistracker.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vistracker.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    syms = header_symbols()
    for need in ("vt_smplh_forward", "vt_smplh_backward", "vt_query_forward", "vt_query_backward", "vt_sil_forward",
                 "vt_chamfer_ragged", "vt_adam_step", "vt_so3_project_forward", "vt_landmarks_forward", "vt_mahalanobis"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from vistracker_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    l = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(l, s)]
    assert not missing, missing
    # and the ctypes signature table covers the header
    uncovered = [s for s in header_symbols() if s not in _lib.SIGNATURES]
    assert not uncovered, uncovered


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU tensor (no silent CPU route)."""
    import torch
    from vistracker_amd import _lib
    with pytest.raises(_lib.VtError):
        _lib.dptr(torch.zeros(3))
