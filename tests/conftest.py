import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end GPU tier)")


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
