"""not-gpu: the C-ABI library loads and exports every symbol include/ganrev.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ganrev.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    import ganrev._lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in ganrev.h but not exported: {missing}"
    # and the ctypes binding covers the same set
    assert sorted(L.EXPORTED_SYMBOLS) == syms, sorted(set(syms) ^ set(L.EXPORTED_SYMBOLS))


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device the product path must raise, never compute on the CPU."""
    import pytest
    import torch
    import ganrev._lib as L
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(L.GanrevError):
        L.Context(0)
    from ganrev import models, synth
    R = models.create_R((1, 8, 8), 4)
    with pytest.raises(L.GanrevError):
        R.forward(synth.uniform((2, 1, 8, 8), 1))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan-reverser_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "ganrev_oracle" not in src and "from oracle" not in src and "import oracle" not in src, os.path.join(dp, f)
