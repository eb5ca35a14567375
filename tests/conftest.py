import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end GPU tier)")
    if os.environ.get("VS_EMU"):
        # tests/test_emu.py re-runs `-m gpu` tests in a child process against the wave64 lockstep interpreter
        # (tests/emu/): same kernel sources, compiled for the host.  Test infrastructure only — the product package
        # itself knows nothing about it.
        from pgvectorscale_amd import _lib
        assert os.path.exists(EMU_LIB), "run `make -C tests/emu` first"
        _lib.LIB_PATH = EMU_LIB


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import pgvectorscale_amd as P
    ctx = P.Context(0)
    yield ctx
    ctx.close()
