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


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing in the package (Python or HIP/C sources, Makefile) imports, includes, links or runs it."""
    import re
    pkg = os.path.join(ROOT, "vistracker_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|libvt_oracle|vt_oracle\.h|/oracle/|\boracle\.(oracle|vt_)", re.M)
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")) or f == "Makefile":
                src = open(os.path.join(d, f), errors="ignore").read()
                if pat.search(src):
                    bad.append(os.path.relpath(os.path.join(d, f), ROOT))
    assert not bad, bad
    # and bench.py uses it only inside cpu_baseline
    b = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle", b)]
    lo = b.index("def cpu_baseline"); hi = b.index("\ndef ", lo + 1)
    assert uses and all(lo < u < hi for u in uses), uses
