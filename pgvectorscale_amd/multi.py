"""Host-side mirror of the multi-GPU entry points of libvsgpu.so (include/vsgpu.h, "multi-GPU").

The path shards by query — every device holds the whole index, device g takes a contiguous block of the batch, one gather of the
[nq, k] blocks ends the step (the reference runs one scan per backend and has nothing to mirror: AM/mod.rs:63).  Two deployments,
both inside the C library, neither needs torch:

  MultiIndex  one process owns N devices (vs_multi_*): the index is replicated device to device over xGMI, one host thread per
              device runs its shard of a host batch and writes its rows into the caller's buffers at the shard's offset
  Comm        one process per device (vs_comm_*): RCCL all-gather of the device-resident id / distance blocks, RCCL broadcast to
              replicate an index (or single arrays of it) from the rank that built it
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Stats, check
from .index import DEFAULT_QUERY_RESCORE, DEFAULT_QUERY_SEARCH_LIST_SIZE, Context, DiskAnnIndex, _p

VS_MULTI_COPY_ALWAYS = 1
VS_COMM_ID_BYTES = 128


def shard_range(nq_total, world, rank):
    """Contiguous block [begin, end) of rank `rank` (vs_shard_range: blocks differ by at most one query)."""
    b, e = C.c_uint32(), C.c_uint32()
    check(_lib.load().vs_shard_range(nq_total, world, rank, C.byref(b), C.byref(e)))
    return int(b.value), int(e.value)


def replicate(index, ctx):
    """vs_index_replicate: a full copy of `index` on ctx's device, device to device (hipMemcpyPeerAsync)."""
    h = C.c_void_p()
    check(index._L.vs_index_replicate(index.h, ctx.h, C.byref(h)))
    return DiskAnnIndex(ctx, h)


class _Borrowed:
    """a context owned by a vs_multi (never destroyed from Python)"""

    def __init__(self, L, h, device):
        self._L, self.h, self.device = L, h, device

    def sync(self):
        check(self._L.vs_ctx_sync(self.h))


class MultiIndex:
    """One copy of `index` per entry of `devices` inside this process (vs_multi_create)."""

    def __init__(self, index, devices, copy_always=False):
        self._L = index._L
        self.src = index
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        check(self._L.vs_multi_create(index.h, devs, len(devices), VS_MULTI_COPY_ALWAYS if copy_always else 0, C.byref(h)))
        self.h = h
        self.devices = list(devices)
        self.dim_full = index.desc.dim_full

    def __len__(self):
        return int(self._L.vs_multi_size(self.h))

    def index(self, i):
        """shard i's index (borrowed: closing it is the MultiIndex's business)"""
        ctx = _Borrowed(self._L, C.c_void_p(self._L.vs_multi_ctx(self.h, i)), self.devices[i])
        ix = DiskAnnIndex(ctx, C.c_void_p(self._L.vs_multi_index(self.h, i)))
        ix.close = lambda: None
        return ix

    def search_batch(self, queries, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE, k=10, qlabels=None):
        """vs_search_batch over all devices: the rows one device returns for the whole batch, in the same order."""
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim_full)
        nq = q.shape[0]
        lv, lo = DiskAnnIndex._label_keys(qlabels, nq)
        ids = np.empty((nq, k), np.uint32)
        tids = np.empty((nq, k), np.uint64)
        dist = np.empty((nq, k), np.float32)
        st = Stats()
        check(self._L.vs_multi_search_batch(self.h, _p(q), _p(lv), _p(lo), nq, search_list_size, rescore, k, _p(ids), _p(tids), _p(dist),
                                            C.byref(st)))
        return ids, tids, dist, st.as_dict()

    def stream_batch(self, queries, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, m=59, qlabels=None):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim_full)
        nq = q.shape[0]
        lv, lo = DiskAnnIndex._label_keys(qlabels, nq)
        ids = np.empty((nq, m), np.uint32)
        ham = np.empty((nq, m), np.uint32)
        st = Stats()
        check(self._L.vs_multi_stream_batch(self.h, _p(q), _p(lv), _p(lo), nq, search_list_size, m, _p(ids), _p(ham), C.byref(st)))
        return ids, ham, st.as_dict()

    def close(self):
        if self.h:
            self._L.vs_multi_destroy(self.h)
            self.h = None


def comm_unique_id():
    """vs_comm_unique_id (ncclGetUniqueId): one rank calls it, the bytes reach every rank by the host's own means"""
    buf = (C.c_uint8 * VS_COMM_ID_BYTES)()
    check(_lib.load().vs_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """One rank of a process-per-device job (vs_comm_create = ncclCommInitRank on ctx's device; collective)."""

    def __init__(self, ctx, unique_id, rank, world):
        assert len(unique_id) == VS_COMM_ID_BYTES
        self._L = ctx._L
        self.ctx = ctx
        buf = (C.c_uint8 * VS_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(self._L.vs_comm_create(ctx.h, buf, rank, world, C.byref(h)))
        self.h = h
        self.rank, self.world = rank, world

    def gather_topk(self, d_ids, d_dist, nq_local, nq_total, k, d_out_ids, d_out_dist):
        """this rank's device-resident [nq_local, k] blocks -> [nq_total, k] on every rank (enqueued on the context's stream)"""
        check(self._L.vs_comm_gather_topk(self.h, d_ids, d_dist, nq_local, nq_total, k, d_out_ids, d_out_dist))

    def bcast(self, d_buf, nbytes, root=0):
        check(self._L.vs_comm_bcast(self.h, d_buf, nbytes, root))

    def replicate_index(self, index, root=0):
        check(self._L.vs_comm_replicate_index(self.h, index.h, root))
        index._refresh()

    def close(self):
        if self.h:
            self._L.vs_comm_destroy(self.h)
            self.h = None


__all__ = ["MultiIndex", "Comm", "comm_unique_id", "shard_range", "replicate", "Context"]
