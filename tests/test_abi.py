"""The C-ABI library loads and exports every symbol include/vsgpu.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "vsgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libvsgpu.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vsgpu.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SYMBOLS) == declared
    P.load()


def test_no_cpu_fallback_without_device():
    """Product path must fail loudly when there is no HIP device (this container) instead of computing on the CPU."""
    import pytest
    import pgvectorscale_amd as P
    try:
        ctx = P.Context(0)
    except P.VsError as e:
        assert e.code == -2 and "no CPU fallback" in str(e)
    else:  # on a GPU box the context simply works
        ctx.close()
        pytest.skip("HIP device present")


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pgvectorscale_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "vs_oracle" not in txt, f
