"""Host-side mirror of the reference's scan interface for the `diskann` access method.

Names follow the reference: `DiskAnnIndex` stands for the index relation (MetaPage + SbqNode pages + SbqMeans,
AM/meta_page.rs, AM/sbq/node.rs), `IndexScan` for the IndexScanDesc whose lifecycle is
ambeginscan -> amrescan -> amgettuple* -> amendscan (AM/scan.rs:308-456).  Everything is computed by libvsgpu.so on
the MI355X; this module only marshals numpy arrays into the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VS_COSINE, VS_INVALID_NODE, VS_IP, VS_L2, IndexDesc, IndexHost, Stats, check  # noqa: F401

# GUC defaults (AM/guc.rs:3-4) and index defaults (AM/meta_page.rs:284-323)
DEFAULT_QUERY_SEARCH_LIST_SIZE = 100
DEFAULT_QUERY_RESCORE = 50
DEFAULT_NUM_NEIGHBORS = 50


def default_bits(dims_to_index):
    return 2 if dims_to_index < 900 else 1


def quantized_size(dims, bits):
    return (dims * bits + 63) // 64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_option(name, value):
    """a tuning option of libvsgpu through the C ABI (vs_set_option); value None unsets it; wins over the environment"""
    check(_lib.load().vs_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    """the option's value as the library sees it (vs_set_option, else the VS_* environment snapshot), or None"""
    buf = C.create_string_buffer(256)
    r = check(_lib.load().vs_get_option(name.encode(), buf, 256))
    return buf.value.decode() if r == 1 else None


class Context:
    """One MI355X + its streams and pinned staging buffers."""

    def __init__(self, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.vs_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def device_name(self):
        buf = C.create_string_buffer(256)
        check(self._L.vs_ctx_device_name(self.h, buf, 256))
        return buf.value.decode()

    def mem_info(self):
        f, t = C.c_uint64(), C.c_uint64()
        check(self._L.vs_ctx_mem_info(self.h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def sync(self):
        check(self._L.vs_ctx_sync(self.h))

    def profile_enable(self, on=True):
        check(self._L.vs_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        """{kernel: (total_ms, launches)} from HIP events recorded on the ctx stream."""
        p = _lib.Profile()
        check(self._L.vs_profile_read(self.h, C.byref(p), int(reset)))
        names = ["prepare_queries", "search", "rerank", "resort", "search_fallback", "scan"]
        return {n: (float(p.ms[k]), int(p.launches[k])) for k, n in enumerate(names)}

    def ws_probe(self, d_mem, nbytes, iters=600):
        """milliseconds the search kernel's private-state request shapes take on this device region (vs_ws_probe)"""
        ms = C.c_float(0)
        check(self._L.vs_ws_probe(self.h, d_mem, nbytes, iters, C.byref(ms)))
        return float(ms.value)

    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self._L.vs_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        check(self._L.vs_dev_free(self.h, p))

    def upload(self, dev_ptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self._L.vs_dev_upload(self.h, dev_ptr, _p(arr), arr.nbytes))

    def download(self, dev_ptr, arr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self._L.vs_dev_download(self.h, _p(arr), dev_ptr, arr.nbytes))
        return arr

    def close(self):
        if self.h:
            self._L.vs_ctx_destroy(self.h)
            self.h = None


class DiskAnnIndex:
    """A `diskann` index resident in HBM."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._L = ctx._L
        self.h = handle
        d = IndexDesc()
        check(self._L.vs_index_get_desc(self.h, C.byref(d)))
        self.desc = d

    def view(self, ctx):
        """A second handle on the same device arrays with its own workspace, bound to `ctx` (another stream of the same device):
        batches submitted through different views overlap on the GPU (vs_index_view).  Close every view before the index."""
        h = C.c_void_p()
        check(self._L.vs_index_view(self.h, ctx.h, C.byref(h)))
        return DiskAnnIndex(ctx, h)

    def ws_probe_mix(self, d_mem, nbytes, iters=400):
        """milliseconds of the search kernel's whole request mix with its private state on this device region (vs_ws_probe_mix)"""
        ms = C.c_float(0)
        check(self._L.vs_ws_probe_mix(self.h, d_mem, nbytes, iters, C.byref(ms)))
        return float(ms.value)

    def set_slab(self, d_mem, nbytes):
        """the caller's device memory as this handle's workspace slab (vs_index_set_slab; before its first search)"""
        check(self._L.vs_index_set_slab(self.h, d_mem, nbytes))

    def prepare_workspace(self):
        """chooses the workspace slab now instead of inside the first search (vs_index_prepare_workspace)"""
        check(self._L.vs_index_prepare_workspace(self.h))

    # -- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def upload(cls, ctx, *, codes, nbrs, heap_tids, vecs, mean, m2, count, bits, dim_index, num_neighbors,
               distance_type, default_start, label_off=None, label_val=None, label_starts=None, storage_type=_lib.VS_STORAGE_SBQ):
        if storage_type == _lib.VS_STORAGE_PLAIN:
            return cls._upload_plain(ctx, nbrs=nbrs, heap_tids=heap_tids, vecs=vecs, num_neighbors=num_neighbors,
                                     distance_type=distance_type, default_start=default_start, dim_index=dim_index)
        codes = np.ascontiguousarray(codes, np.uint64)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        heap_tids = np.ascontiguousarray(heap_tids, np.uint64)
        vecs = None if vecs is None else np.ascontiguousarray(vecs, np.float32)
        mean = np.ascontiguousarray(mean, np.float32)
        m2 = None if m2 is None else np.ascontiguousarray(m2, np.float32)
        n, w = codes.shape
        # the C side reads n rows of every array: a shorter one would be read past its end
        if nbrs.ndim != 2 or nbrs.shape[0] != n or nbrs.shape[1] < num_neighbors:
            raise ValueError(f"nbrs must be [{n}][>= {num_neighbors}], got {nbrs.shape}")
        if heap_tids.shape != (n,):
            raise ValueError(f"heap_tids must have shape ({n},), got {heap_tids.shape}")
        if vecs is not None and (vecs.ndim != 2 or vecs.shape[0] != n or vecs.shape[1] < dim_index):
            raise ValueError(f"vecs must be [{n}][>= {dim_index}], got {vecs.shape}")
        if mean.shape != (dim_index,) or (m2 is not None and m2.shape != (dim_index,)):
            raise ValueError(f"mean / m2 must have shape ({dim_index},)")
        if label_off is not None and np.asarray(label_off).shape != (n + 1,):
            raise ValueError(f"label_off must have shape ({n + 1},)")
        d = IndexDesc()
        d.n, d.dim_index, d.bits, d.words = n, dim_index, bits, w
        d.dim_full = dim_index if vecs is None else vecs.shape[1]
        d.num_neighbors, d.distance_type = num_neighbors, distance_type
        d.has_labels = int(label_off is not None)
        d.default_start = default_start
        ls = sorted((label_starts or {}).items())
        d.n_label_starts = len(ls)
        lsl = np.array([k for k, _ in ls], np.int16)
        lsn = np.array([v for _, v in ls], np.uint32)
        lo = None if label_off is None else np.ascontiguousarray(label_off, np.uint32)
        lv = None if label_val is None else np.ascontiguousarray(label_val, np.int16)
        h = IndexHost()
        h.codes, h.nbrs, h.nbr_stride = _p(codes).value, _p(nbrs).value, nbrs.shape[1]
        h.heap_tids = _p(heap_tids).value
        h.vecs = None if vecs is None else _p(vecs).value
        h.mean = _p(mean).value
        h.m2 = None if m2 is None else _p(m2).value
        h.count = count
        h.label_off = None if lo is None else _p(lo).value
        h.label_val = None if lv is None else _p(lv).value
        h.label_start_labels = _p(lsl).value if len(ls) else None
        h.label_start_nodes = _p(lsn).value if len(ls) else None
        out = C.c_void_p()
        check(ctx._L.vs_index_upload(ctx.h, C.byref(d), C.byref(h), C.byref(out)))
        return cls(ctx, out)

    @classmethod
    def _upload_plain(cls, ctx, *, nbrs, heap_tids, vecs, num_neighbors, distance_type, default_start, dim_index=None):
        """A `plain` storage index (AM/plain/storage.rs): vectors + neighbor lists, no SBQ codes."""
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        heap_tids = np.ascontiguousarray(heap_tids, np.uint64)
        vecs = np.ascontiguousarray(vecs, np.float32)
        d = IndexDesc()
        d.n, d.dim_full, d.dim_index = vecs.shape[0], vecs.shape[1], dim_index or vecs.shape[1]
        d.bits, d.words = 1, quantized_size(d.dim_index, 1)
        d.num_neighbors, d.distance_type, d.has_labels = num_neighbors, distance_type, 0
        d.default_start, d.n_label_starts, d.storage_type = default_start, 0, _lib.VS_STORAGE_PLAIN
        h = IndexHost()
        h.nbrs, h.nbr_stride, h.heap_tids, h.vecs = _p(nbrs).value, nbrs.shape[1], _p(heap_tids).value, _p(vecs).value
        out = C.c_void_p()
        check(ctx._L.vs_index_upload(ctx.h, C.byref(d), C.byref(h), C.byref(out)))
        return cls(ctx, out)

    @classmethod
    def alloc(cls, ctx, *, n, dim_full, dim_index=None, bits=None, num_neighbors=DEFAULT_NUM_NEIGHBORS,
              distance_type=VS_L2, with_vecs=True):
        dim_index = dim_index or dim_full
        bits = bits or default_bits(dim_index)
        d = IndexDesc()
        d.n, d.dim_full, d.dim_index, d.bits = n, dim_full, dim_index, bits
        d.words = quantized_size(dim_index, bits)
        d.num_neighbors, d.distance_type, d.has_labels = num_neighbors, distance_type, 0
        d.default_start, d.n_label_starts = VS_INVALID_NODE, 0
        out = C.c_void_p()
        check(ctx._L.vs_index_alloc(ctx.h, C.byref(d), int(with_vecs), C.byref(out)))
        return cls(ctx, out)

    def _refresh(self):
        check(self._L.vs_index_get_desc(self.h, C.byref(self.desc)))

    def array(self, which):
        p, s = C.c_void_p(), C.c_uint32()
        check(self._L.vs_index_array(self.h, which, C.byref(p), C.byref(s)))
        return p, int(s.value)

    def set_quantizer(self, mean, m2, count):
        mean = np.ascontiguousarray(mean, np.float32)
        m2 = None if m2 is None else np.ascontiguousarray(m2, np.float32)
        check(self._L.vs_index_set_quantizer(self.h, _p(mean), _p(m2), count))

    def get_quantizer(self):
        mean = np.empty(self.desc.dim_index, np.float32)
        m2 = np.empty(self.desc.dim_index, np.float32)
        cnt = C.c_uint64()
        check(self._L.vs_index_get_quantizer(self.h, _p(mean), _p(m2), C.byref(cnt)))
        return mean, m2, int(cnt.value)

    def set_start_nodes(self, default_start, label_starts=None):
        ls = sorted((label_starts or {}).items())
        lsl = np.array([k for k, _ in ls], np.int16)
        lsn = np.array([v for _, v in ls], np.uint32)
        check(self._L.vs_index_set_start_nodes(self.h, default_start, _p(lsl), _p(lsn), len(ls)))
        self._refresh()

    def set_labels(self, label_off, label_val):
        lo = np.ascontiguousarray(label_off, np.uint32)
        lv = np.ascontiguousarray(label_val, np.int16)
        check(self._L.vs_index_set_labels(self.h, _p(lo), _p(lv)))
        self._refresh()

    def set_visibility(self, visible):
        """Heap visibility under the scans' snapshot: uint8 [n], 0 = the heap fetch of get_full_distance_for_resort finds no
        visible tuple (the candidate is counted and dropped before the rescore window, AM/scan.rs:268-272); None = all visible."""
        if visible is None:
            check(self._L.vs_index_set_visibility(self.h, None))
            return
        v = np.ascontiguousarray(visible, np.uint8)
        if v.shape != (self.desc.n,):
            raise ValueError(f"visibility mask must have shape ({self.desc.n},), got {v.shape}")
        check(self._L.vs_index_set_visibility(self.h, _p(v)))

    def download(self, codes=True, nbrs=True, tids=True, vecs=False, row_begin=0, row_count=None):
        n = self.desc.n if row_count is None else row_count
        out = {}
        c = np.empty((n, self.desc.words), np.uint64) if codes else None
        nb = np.empty((n, self.desc.num_neighbors), np.uint32) if nbrs else None
        t = np.empty(n, np.uint64) if tids else None
        v = np.empty((n, self.desc.dim_full), np.float32) if vecs else None
        check(self._L.vs_index_download(self.h, _p(c), _p(nb), _p(t), _p(v), row_begin, n))
        out.update(codes=c, nbrs=nb, heap_tids=t, vecs=v)
        return out

    def mark_deleted(self, nodes):
        a = np.ascontiguousarray(nodes, np.uint32)
        check(self._L.vs_index_mark_deleted(self.h, _p(a), a.size))

    def refresh_norms(self):
        check(self._L.vs_index_refresh_norms(self.h))

    # -- build-side helpers -----------------------------------------------------------------------------------------
    def sbq_train(self):
        check(self._L.vs_sbq_train(self.h))

    def sbq_quantize_corpus(self):
        check(self._L.vs_sbq_quantize_corpus(self.h))

    def build_graph(self, search_list_size=100, max_alpha=1.2, batch_max=0, seed=0):
        check(self._L.vs_build_graph(self.h, search_list_size, max_alpha, batch_max, seed))
        self._refresh()

    def build_unreachable(self):
        """nodes the last build_graph left unreachable from the default start node (0 on well-formed input)"""
        return int(self._L.vs_index_build_unreachable(self.h))

    # -- single kernels ------------------------------------------------------------------------------------------------
    def save_graph(self, path):
        """Dump the neighbor array (device layout [n][nbr_stride] u32) to a file (benchmark convenience: the on-device
        build of a 50M-node index takes minutes and is deterministic)."""
        ptr, stride = self.array(_lib.ARR_NBRS)
        arr = np.empty((self.desc.n, stride), np.uint32)
        self.ctx.download(ptr, arr)
        arr.tofile(path)

    def load_graph(self, path, default_start=0):
        ptr, stride = self.array(_lib.ARR_NBRS)
        arr = np.fromfile(path, np.uint32)
        if arr.size != self.desc.n * stride:
            raise ValueError(f"{path}: {arr.size} u32, expected {self.desc.n} x {stride}")
        self.ctx.upload(ptr, arr.reshape(self.desc.n, stride))
        self.set_start_nodes(default_start)

    def quantize(self, q):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, self.desc.dim_index)
        out = np.empty((q.shape[0], self.desc.words), np.uint64)
        check(self._L.vs_quantize(self.h, _p(q), q.shape[0], _p(out)))
        return out

    @staticmethod
    def _csr(lists):
        off = np.zeros(len(lists) + 1, np.uint32)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        flat = np.concatenate([np.asarray(l, np.uint32) for l in lists]) if len(lists) and off[-1] else np.zeros(0, np.uint32)
        return np.ascontiguousarray(flat, np.uint32), off

    def hamming_gather(self, qcodes, id_lists):
        qcodes = np.ascontiguousarray(qcodes, np.uint64).reshape(-1, self.desc.words)
        ids, off = self._csr(id_lists)
        out = np.empty(ids.size, np.uint32)
        check(self._L.vs_hamming_gather(self.h, _p(qcodes), _p(ids), _p(off), qcodes.shape[0], _p(out)))
        return [out[off[i]:off[i + 1]] for i in range(len(id_lists))]

    def rerank(self, queries, id_lists):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.desc.dim_full)
        ids, off = self._csr(id_lists)
        out = np.empty(ids.size, np.float32)
        check(self._L.vs_rerank(self.h, _p(q), _p(ids), _p(off), q.shape[0], _p(out)))
        return [out[off[i]:off[i + 1]] for i in range(len(id_lists))]

    def scan_topk(self, qcodes, k, qlabels=None, live_only=False):
        """flat SBQ scan: exact top-k of the Hamming distance, order (hamming, node id); qlabels (one label list per query) /
        live_only restrict it to the rows a label-filtered scan may return"""
        qcodes = np.ascontiguousarray(qcodes, np.uint64).reshape(-1, self.desc.words)
        nq = qcodes.shape[0]
        ids = np.empty((nq, k), np.uint32)
        ham = np.empty((nq, k), np.uint32)
        if qlabels is None and not live_only:
            check(self._L.vs_scan_topk(self.h, _p(qcodes), nq, k, _p(ids), _p(ham)))
            return ids, ham
        lv, lo = self._label_keys(qlabels, nq) if qlabels is not None else (None, None)
        check(self._L.vs_scan_topk_filtered(self.h, _p(qcodes), _p(lv), _p(lo), int(live_only), nq, k, _p(ids), _p(ham)))
        return ids, ham

    # -- batched scans ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _label_keys(qlabels, nq):
        if qlabels is None:
            return None, None
        assert len(qlabels) == nq
        off = np.zeros(nq + 1, np.uint32)
        vals = []
        for i, l in enumerate(qlabels):
            vals.extend(int(x) for x in l)
            off[i + 1] = len(vals)
        return np.array(vals, np.int16), off

    def search_batch(self, queries, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE,
                     k=10, qlabels=None):
        """For each query: the rows of the first k amgettuple calls (node ids, heap TIDs, reranked distances)."""
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.desc.dim_full)
        nq = q.shape[0]
        lv, lo = self._label_keys(qlabels, nq)
        ids = np.empty((nq, k), np.uint32)
        tids = np.empty((nq, k), np.uint64)
        dist = np.empty((nq, k), np.float32)
        st = Stats()
        check(self._L.vs_search_batch(self.h, _p(q), _p(lv), _p(lo), nq, search_list_size, rescore, k, _p(ids), _p(tids),
                                      _p(dist), C.byref(st)))
        return ids, tids, dist, st.as_dict()

    def stream_batch(self, queries, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, m=59, qlabels=None):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.desc.dim_full)
        nq = q.shape[0]
        lv, lo = self._label_keys(qlabels, nq)
        ids = np.empty((nq, m), np.uint32)
        ham = np.empty((nq, m), np.uint32)
        st = Stats()
        check(self._L.vs_stream_batch(self.h, _p(q), _p(lv), _p(lo), nq, search_list_size, m, _p(ids), _p(ham),
                                      C.byref(st)))
        return ids, ham, st.as_dict()

    def search_batch_dev(self, d_queries, nq, search_list_size, rescore, k, d_out_ids, d_out_tids=None, d_out_dist=None,
                         d_qlabels=None, d_qlabel_off=None):
        """d_qlabels / d_qlabel_off: device CSR of the label keys (sorted, de-duplicated per query) or None"""
        check(self._L.vs_search_batch_dev(self.h, d_queries, d_qlabels, d_qlabel_off, nq, search_list_size, rescore, k, d_out_ids,
                                          d_out_tids, d_out_dist))

    def search_batch_dev_finish(self):
        st = Stats()
        check(self._L.vs_search_batch_dev_finish(self.h, C.byref(st)))
        return st.as_dict()

    def autotune(self, d_queries, nq, search_list_size, rescore, k, d_qlabels=None, d_qlabel_off=None, reps=2, skip=()):
        """vs_index_autotune: times every applicable exact launch variant of the search kernel on this device-resident batch,
        disqualifies any whose rows / distance bits / counters differ from the default's, keeps the fastest -> report (list of dicts)"""
        from ._lib import TuneEntry
        rep = (TuneEntry * 32)()
        n = C.c_uint32(0)
        check(self._L.vs_index_autotune(self.h, d_queries, d_qlabels, d_qlabel_off, nq, search_list_size, rescore, k, reps,
                                        ",".join(skip).encode() if skip else None, rep, 32, C.byref(n)))
        return [rep[i].as_dict() for i in range(min(int(n.value), 32))]

    def set_variant(self, name):
        check(self._L.vs_index_set_variant(self.h, name.encode()))

    def variant(self):
        buf = C.create_string_buffer(64)
        check(self._L.vs_index_get_variant(self.h, buf, 64))
        return buf.value.decode()

    def bruteforce_topk(self, d_queries, nq, k):
        ids = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        check(self._L.vs_bruteforce_topk(self.h, d_queries, nq, k, _p(ids), _p(dist)))
        return ids, dist

    # -- the access-method surface ------------------------------------------------------------------------------------------
    def beginscan(self):
        """ambeginscan (AM/scan.rs:308-333)"""
        return IndexScan(self)

    def close(self):
        if self.h:
            self._L.vs_index_free(self.h)
            self.h = None


class ScanPool:
    """Many streamed scans of one index continued by launches they share (vs_scanpool_*, csrc/vs_scanpool.cpp): slot i is one
    backend's amrescan / amgettuple cursor; fetch() serves the next k rows of every listed slot with ONE resumed search launch (+ one
    rerank launch) per round.  All slots share search_list_size / rescore."""

    def __init__(self, index, capacity, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE, kmax=64, rows_cap=0):
        self.index = index
        self._L = index._L
        self.kmax = kmax
        h = C.c_void_p()
        check(self._L.vs_scanpool_create(index.h, capacity, search_list_size, rescore, kmax, rows_cap, C.byref(h)))
        self.h = h

    def rescan(self, slot, query, labels=None):
        q = None if query is None else np.ascontiguousarray(query, np.float32).reshape(self.index.desc.dim_full)
        lab = None if labels is None else np.ascontiguousarray(labels, np.int16)
        check(self._L.vs_scanpool_rescan(self.h, slot, _p(q), _p(lab), 0 if lab is None else lab.size, int(labels is not None)))

    def endscan(self, slot):
        check(self._L.vs_scanpool_endscan(self.h, slot))

    def fetch(self, slots, k):
        """-> (rows [n] int32 (negative: that slot's error code), ids [n][k], tids [n][k], dist [n][k])"""
        sl = np.ascontiguousarray(slots, np.uint32)
        n = sl.size
        ids = np.full((n, k), 0xFFFFFFFF, np.uint32)
        tids = np.zeros((n, k), np.uint64)
        dist = np.zeros((n, k), np.float32)
        rows = np.zeros(n, np.int32)
        check(self._L.vs_scanpool_fetch(self.h, _p(sl), n, k, _p(tids), _p(ids), _p(dist), _p(rows)))
        return rows, ids, tids, dist

    def stats(self, slot):
        st = _lib.Stats()
        check(self._L.vs_scanpool_get_stats(self.h, slot, C.byref(st)))
        return st.as_dict()

    def work(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.vs_scanpool_get_work(self.h, C.byref(a), C.byref(b)))
        return {"launches": int(a.value), "rounds": int(b.value)}

    def close(self):
        if self.h:
            self._L.vs_scanpool_free(self.h)
            self.h = None


class Broker:
    """Coalesces the scans of many client threads into batched launches (vs_broker_*; threads stand in for PostgreSQL
    backends).  search() may be called from any thread; the library's dispatcher thread is the only one that touches the
    device context."""

    def __init__(self, index, max_batch=0, max_wait_us=200, cursor_lanes=0):
        """cursor_lanes > 0: the continuations of the scans' cursors run on that many lanes (a thread, a HIP stream and a view of
        the index each) instead of on the dispatcher thread between two shared launches"""
        self.index = index
        self._L = index._L
        cfg = _lib.BrokerConfig(max_batch, max_wait_us, cursor_lanes)
        h = C.c_void_p()
        check(self._L.vs_broker_create(index.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def search(self, query, labels=None, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE, k=10,
               snapshot=0):
        """snapshot: id of the visibility mask the scan runs under (snapshot_put; 0 = every heap tuple visible)"""
        q = None if query is None else np.ascontiguousarray(query, np.float32).reshape(self.index.desc.dim_full)
        lab = None if labels is None else np.ascontiguousarray(labels, np.int16)
        ids = np.empty(k, np.uint32)
        tids = np.empty(k, np.uint64)
        dist = np.empty(k, np.float32)
        check(self._L.vs_broker_search_snapshot(self.h, _p(q), _p(lab), 0 if lab is None else lab.size, int(labels is not None),
                                                search_list_size, rescore, k, snapshot, _p(ids), _p(tids), _p(dist)))
        return ids, tids, dist

    def snapshot_put(self, snapshot, visible):
        """hands the dispatcher the per-node visibility mask of snapshot id 1..15 (None drops it); any thread"""
        v = None if visible is None else np.ascontiguousarray(visible, np.uint8)
        if v is not None and v.shape != (self.index.desc.n,):
            raise ValueError("one byte per node")
        check(self._L.vs_broker_snapshot_put(self.h, snapshot, _p(v)))

    def beginscan(self):
        """ambeginscan for a backend whose scans go through this broker"""
        return IndexScan(self.index, broker=self)

    def stats(self):
        st = _lib.BrokerStats()
        check(self._L.vs_broker_get_stats(self.h, C.byref(st)))
        return {"batches": int(st.batches), "scans": int(st.scans), "max_batch": int(st.max_batch), "tasks": int(st.tasks),
                "cursors": int(st.cursors)}

    def close(self):
        if self.h:
            self._L.vs_broker_destroy(self.h)
            self.h = None


class ShmServer:
    """The dispatcher side of the cross-process request queue (vs_shm_server_*): lives in the one process that owns the device
    context; client processes post scans into the POSIX shared-memory segment `name` and get the rows of vs_search_batch."""

    def __init__(self, index, name, nslots=256, kmax=64, max_batch=0, max_wait_us=200, cursor_lanes=0, cursor_pool=0):
        self.index = index
        self._L = index._L
        # cursor_lanes: streamed scans are served on that many lanes; cursor_pool: ... or out of scan pools of that many slots, the
        # continuations of one dispatcher round sharing their launches
        cfg = _lib.BrokerConfig(max_batch, max_wait_us, cursor_lanes, cursor_pool)
        h = C.c_void_p()
        check(self._L.vs_shm_server_create(index.h, name.encode(), nslots, kmax, C.byref(cfg), C.byref(h)))
        self.h = h

    def snapshot_put(self, snapshot, visible):
        """the serving process's per-node visibility mask of snapshot id 1..15 (None drops it)"""
        v = None if visible is None else np.ascontiguousarray(visible, np.uint8)
        if v is not None and v.shape != (self.index.desc.n,):
            raise ValueError("one byte per node")
        check(self._L.vs_shm_server_snapshot_put(self.h, snapshot, _p(v)))

    def stats(self):
        st = _lib.BrokerStats()
        check(self._L.vs_shm_server_get_stats(self.h, C.byref(st)))
        return {"batches": int(st.batches), "scans": int(st.scans), "max_batch": int(st.max_batch), "tasks": int(st.tasks),
                "cursors": int(st.cursors)}

    def pool_stats(self):
        """scan pools of the server: pools alive, shared fetch rounds, pools re-keyed at the cap, scans living in pools"""
        out = (C.c_uint64 * 4)()
        check(self._L.vs_shm_server_pool_stats(self.h, out))
        return {"pools": int(out[0]), "rounds": int(out[1]), "retired": int(out[2]), "scans": int(out[3])}

    def close(self):
        if self.h:
            self._L.vs_shm_server_destroy(self.h)
            self.h = None


class ShmClient:
    """A backend's end of the queue (vs_shm_client_*): needs no GPU and no device context, only the segment's name."""

    def __init__(self, name):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.vs_shm_client_open(name.encode(), C.byref(h)))
        self.h = h
        self.dim = int(self._L.vs_shm_client_dim(h))

    def search(self, query, labels=None, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE, k=10,
               snapshot=0):
        q = None if query is None else np.ascontiguousarray(query, np.float32).reshape(self.dim)
        lab = None if labels is None else np.ascontiguousarray(labels, np.int16)
        ids = np.empty(k, np.uint32)
        tids = np.empty(k, np.uint64)
        dist = np.empty(k, np.float32)
        check(self._L.vs_shm_client_search_snapshot(self.h, _p(q), _p(lab), 0 if lab is None else lab.size, int(labels is not None),
                                                    search_list_size, rescore, k, snapshot, _p(ids), _p(tids), _p(dist)))
        return ids, tids, dist

    def fetch(self, scan_id, query, skip, k, labels=None, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE,
              snapshot=0):
        """rows [skip, skip + k) of the scan this process calls scan_id, continued on the serving process's cursor for it (fewer
        than k rows: the scan has ended)"""
        q = None if query is None else np.ascontiguousarray(query, np.float32).reshape(self.dim)
        lab = None if labels is None else np.ascontiguousarray(labels, np.int16)
        ids = np.empty(k, np.uint32)
        tids = np.empty(k, np.uint64)
        dist = np.empty(k, np.float32)
        n = C.c_uint32(0)
        check(self._L.vs_shm_client_fetch(self.h, scan_id, _p(q), _p(lab), 0 if lab is None else lab.size, int(labels is not None),
                                          search_list_size, rescore, snapshot, skip, k, _p(ids), _p(tids), _p(dist), C.byref(n)))
        return ids[:n.value], tids[:n.value], dist[:n.value]

    def end_scan(self, scan_id):
        check(self._L.vs_shm_client_end_scan(self.h, scan_id))

    def beginscan(self, chunk=16):
        """the amrescan / amgettuple surface of a backend process: the first `chunk` rows come from a launch shared with the other
        backends' scans, every later chunk continues the scan's cursor in the serving process"""
        return ShmScan(self, chunk)

    def close(self):
        if self.h:
            self._L.vs_shm_client_close(self.h)
            self.h = None


class ShmScan:
    """IndexScanDesc of a backend process that reaches the device through a ShmClient."""

    _next_id = 1

    def __init__(self, client, chunk=16):
        self.client = client
        self.chunk = chunk
        self.scan_id = ShmScan._next_id
        ShmScan._next_id += 1
        self._args = None
        self._rows = []
        self._pos = 0
        self._streamed = False
        self._ended = True

    def rescan(self, query, labels=None, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE, snapshot=0):
        if self._streamed:
            self.client.end_scan(self.scan_id)
        self.scan_id = ShmScan._next_id  # (a new scan: the serving process must not mistake it for the old one)
        ShmScan._next_id += 1
        self._args = dict(query=None if query is None else np.array(query, np.float32), labels=labels,
                          search_list_size=search_list_size, rescore=rescore, snapshot=snapshot)
        self._rows, self._pos, self._streamed, self._ended = [], 0, False, False

    def gettuple(self):
        """(heap_tid, node, distance) or None at end of scan"""
        if self._pos >= len(self._rows) and not self._ended:
            a = self._args
            if not self._rows:  # the first rows: a shared launch
                ids, tids, dist = self.client.search(a["query"], a["labels"], a["search_list_size"], a["rescore"], self.chunk, a["snapshot"])
                keep = ids != 0xFFFFFFFF
                n = int(keep.argmin()) if not keep.all() else len(ids)
                ids, tids, dist = ids[:n], tids[:n], dist[:n]
                self._ended = n < self.chunk
            else:
                ids, tids, dist = self.client.fetch(self.scan_id, a["query"], len(self._rows), self.chunk, a["labels"],
                                                    a["search_list_size"], a["rescore"], a["snapshot"])
                self._streamed = True
                self._ended = len(ids) < self.chunk
            self._rows.extend(zip(tids.tolist(), ids.tolist(), dist.tolist()))
        if self._pos >= len(self._rows):
            return None
        r = self._rows[self._pos]
        self._pos += 1
        return r

    def endscan(self):
        if self._streamed:
            self.client.end_scan(self.scan_id)
            self._streamed = False


class IndexScan:
    """IndexScanDesc + TSVScanState: rescan() = amrescan, gettuple() = amgettuple, endscan() = amendscan."""

    def __init__(self, index, broker=None):
        self.index = index
        self._L = index._L
        h = C.c_void_p()
        if broker is None:
            check(self._L.vs_beginscan(index.h, C.byref(h)))
        else:  # windows of rows are fetched through the broker: scans of many threads share launches
            check(self._L.vs_beginscan_on_broker(broker.h, C.byref(h)))
        self.h = h

    def rescan(self, query, labels=None, search_list_size=DEFAULT_QUERY_SEARCH_LIST_SIZE, rescore=DEFAULT_QUERY_RESCORE):
        """query=None is the SQL NULL query; labels=None means no scan key (nkeys == 0)."""
        q = None if query is None else np.ascontiguousarray(query, np.float32).reshape(-1)
        if q is not None and q.size != self.index.desc.dim_full:  # (vs_rescan reads dim_full floats)
            raise ValueError(f"query has {q.size} dimensions, the index {self.index.desc.dim_full}")
        lv = None if labels is None else np.array(labels, np.int16)
        check(self._L.vs_rescan(self.h, _p(q), _p(lv), 0 if lv is None else lv.size, int(labels is not None),
                                search_list_size, rescore))

    def gettuple(self):
        """Returns (heap_tid, node, distance) or None at end of scan."""
        tid, node, d = C.c_uint64(), C.c_uint32(), C.c_float()
        r = check(self._L.vs_gettuple(self.h, C.byref(tid), C.byref(node), C.byref(d)))
        if r == 0:
            return None
        return int(tid.value), int(node.value), np.float32(d.value)

    @property
    def xs_recheck(self):
        return bool(self._L.vs_scan_xs_recheck(self.h))

    def stats(self):
        st = Stats()
        check(self._L.vs_scan_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def set_snapshot(self, snapshot):
        """(scan on a broker) the visibility mask the scan runs under from its next rescan() on"""
        check(self._L.vs_scan_set_snapshot(self.h, snapshot))

    def prefetch(self, rows):
        """hint: the executor will pull `rows` rows in all"""
        check(self._L.vs_scan_prefetch(self.h, rows))

    def work(self):
        """what the device really did for this scan since rescan() (prefetch and restarts included) + its launch count"""
        st, n = Stats(), C.c_uint32()
        check(self._L.vs_scan_get_work(self.h, C.byref(st), C.byref(n)))
        d = st.as_dict()
        d["launches"] = int(n.value)
        return d

    def endscan(self):
        if self.h:
            self._L.vs_endscan(self.h)
            self.h = None
