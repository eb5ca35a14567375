"""pgvectorscale_amd — MI355X (gfx950) implementation of pgvectorscale's StreamingDiskANN search hot path.

The compute lives in libvsgpu.so (hand-written HIP kernels behind the C ABI of include/vsgpu.h); this package is
the thin host-side mirror of the reference's scan interface (index.py) plus the synthetic corpus generator used by
tests and bench (datagen.py).
"""
from ._lib import VS_COSINE, VS_INVALID_NODE, VS_IP, VS_L2, VsError, load  # noqa: F401
from .index import Broker, Context, DiskAnnIndex, IndexScan, ScanPool, ShmClient, ShmServer, get_option, set_option  # noqa: F401
from .multi import Comm, MultiIndex  # noqa: F401  (several GPUs: one process with N devices / one process per device over RCCL)
