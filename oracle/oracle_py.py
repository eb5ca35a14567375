"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference search path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
INVALID = 0xFFFFFFFF
COSINE, L2, IP = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH) for f in ("vs_oracle.cpp", "vs_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class VsoIndex(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("dim_full", C.c_uint32), ("dim_index", C.c_uint32), ("bits", C.c_uint32),
        ("words", C.c_uint32), ("num_neighbors", C.c_uint32), ("nbr_stride", C.c_uint32),
        ("distance_type", C.c_uint32), ("has_labels", C.c_uint32), ("default_start", C.c_uint32),
        ("n_label_starts", C.c_uint32),
        ("label_start_labels", C.c_void_p), ("label_start_nodes", C.c_void_p),
        ("codes", C.c_void_p), ("nbrs", C.c_void_p), ("heap_tids", C.c_void_p), ("vecs", C.c_void_p),
        ("label_off", C.c_void_p), ("label_val", C.c_void_p), ("mean", C.c_void_p), ("m2", C.c_void_p),
        ("count", C.c_uint64), ("storage_plain", C.c_uint32), ("visible", C.c_void_p),
    ]


class VsoStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "calls", "node_reads", "node_heap_reads", "quantized_distance_comparisons", "full_distance_comparisons",
        "visited_nodes", "candidate_nodes", "next_calls", "next_calls_with_resort")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp, u64p, u32p, i16p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_int16), C.POINTER(C.c_uint8))
    vp = C.c_void_p
    sz = C.c_size_t
    L.vso_distance_xor.restype = C.c_uint64
    L.vso_distance_xor.argtypes = [vp, vp, sz]
    for name in ("vso_distance_l2", "vso_inner_product", "vso_distance_inner_product", "vso_distance_cosine",
                 "vso_distance_l2_unoptimized", "vso_inner_product_unoptimized", "vso_distance_cosine_unoptimized",
                 "vso_distance_l2_avx2", "vso_inner_product_avx2"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [vp, vp, sz]
    L.vso_have_avx2.restype = C.c_int
    L.vso_micro_bench.restype = C.c_double
    L.vso_micro_bench.argtypes = [C.c_int, C.c_uint64]
    L.vso_preprocess_cosine.restype = C.c_int
    L.vso_preprocess_cosine.argtypes = [vp, sz]
    L.vso_distance_by_type.restype = C.c_float
    L.vso_distance_by_type.argtypes = [C.c_int, vp, vp, sz]
    L.vso_quantized_size.restype = sz
    L.vso_quantized_size.argtypes = [sz, C.c_uint]
    L.vso_default_bits.restype = C.c_uint
    L.vso_default_bits.argtypes = [sz]
    L.vso_quantize.restype = None
    L.vso_quantize.argtypes = [vp, vp, C.c_uint64, C.c_uint, vp, sz, vp]
    L.vso_train.restype = None
    L.vso_train.argtypes = [vp, vp, u64p, C.c_uint, vp, sz, sz]
    L.vso_labelset_from.restype = sz
    L.vso_labelset_from.argtypes = [vp, sz]
    L.vso_labels_overlap.restype = C.c_int
    L.vso_labels_overlap.argtypes = [vp, sz, vp, sz]
    L.vso_labels_contains_intersection.restype = C.c_int
    L.vso_labels_contains_intersection.argtypes = [vp, sz, vp, sz, vp, sz]
    L.vso_smallint_array_overlap.restype = C.c_int
    L.vso_smallint_array_overlap.argtypes = [vp, vp, sz, vp, vp, sz]
    L.vso_scan_begin.restype = vp
    L.vso_scan_begin.argtypes = [C.POINTER(VsoIndex), vp, vp, sz, C.c_int, C.c_uint32, C.c_uint32]
    L.vso_scan_gettuple.restype = C.c_int
    L.vso_scan_gettuple.argtypes = [vp, u32p, u64p, fp]
    L.vso_scan_next_sbq.restype = C.c_int
    L.vso_scan_next_sbq.argtypes = [vp, u32p, u64p, u32p]
    L.vso_scan_xs_recheck.restype = C.c_int
    L.vso_scan_xs_recheck.argtypes = [vp]
    L.vso_scan_stats.restype = None
    L.vso_scan_stats.argtypes = [vp, C.POINTER(VsoStats)]
    L.vso_scan_end.restype = None
    L.vso_scan_end.argtypes = [vp]
    L.vso_search_batch.restype = None
    L.vso_search_batch.argtypes = [C.POINTER(VsoIndex), vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_uint32, vp, vp, C.POINTER(VsoStats)]
    L.vso_stream_batch.restype = None
    L.vso_stream_batch.argtypes = [C.POINTER(VsoIndex), vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   vp, vp, C.POINTER(VsoStats)]
    L.vso_heap_replay.restype = sz
    L.vso_heap_replay.argtypes = [vp, sz, vp]
    L.vso_build_graph.restype = None
    L.vso_build_graph.argtypes = [C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, u32p]
    L.vso_build_graph_labeled.restype = C.c_uint32
    L.vso_build_graph_labeled.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp,
                                          u32p, vp, vp]
    L.vso_bruteforce_topk.restype = None
    L.vso_bruteforce_topk.argtypes = [C.POINTER(VsoIndex), vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.vso_hamming_scan_topk.restype = None
    L.vso_hamming_scan_topk.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.vso_hamming_scan_topk_filtered.restype = None
    L.vso_hamming_scan_topk_filtered.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- L1 arithmetic -------------------------------------------------------------------------------
def distance_xor(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(lib().vso_distance_xor(_p(a), _p(b), a.size))


def _dist(name, a, b):
    a, b = _f32(a), _f32(b)
    assert a.size == b.size
    return np.float32(getattr(lib(), name)(_p(a), _p(b), a.size))


def distance_l2(a, b): return _dist("vso_distance_l2", a, b)
def inner_product(a, b): return _dist("vso_inner_product", a, b)
def distance_inner_product(a, b): return _dist("vso_distance_inner_product", a, b)
def distance_cosine(a, b): return _dist("vso_distance_cosine", a, b)
def distance_l2_unoptimized(a, b): return _dist("vso_distance_l2_unoptimized", a, b)
def inner_product_unoptimized(a, b): return _dist("vso_inner_product_unoptimized", a, b)
def distance_cosine_unoptimized(a, b): return _dist("vso_distance_cosine_unoptimized", a, b)
def distance_l2_avx2(a, b): return _dist("vso_distance_l2_avx2", a, b)
def inner_product_avx2(a, b): return _dist("vso_inner_product_avx2", a, b)
def have_avx2(): return bool(lib().vso_have_avx2())


def micro_bench(iters=200000):
    """ns per call on the inputs of the reference's criterion benches (benches/distance.rs:144-161,299-338), one thread"""
    names = ["distance_l2_2000d", "distance_cosine_2000d", "inner_product_2000d", "distance_xor_optimized_1536bit"]
    return {n: round(float(lib().vso_micro_bench(i, iters * (20 if i == 3 else 1))), 2) for i, n in enumerate(names)}



def distance_by_type(t, a, b):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().vso_distance_by_type(int(t), _p(a), _p(b), a.size))


def preprocess_cosine(v):
    """Returns (normalised copy, changed?)."""
    v = _f32(v).copy()
    ch = lib().vso_preprocess_cosine(_p(v), v.size)
    return v, bool(ch)


# ---- SBQ ----------------------------------------------------------------------------------------------
def quantized_size(dims, bits): return int(lib().vso_quantized_size(dims, bits))
def default_bits(dims): return int(lib().vso_default_bits(dims))


def train(rows, bits):
    rows = _f32(rows)
    n, d = rows.shape
    mean = np.zeros(d, np.float32)
    m2 = np.zeros(d, np.float32)
    cnt = C.c_uint64(0)
    lib().vso_train(_p(mean), _p(m2), C.byref(cnt), bits, _p(rows), n, d)
    return mean, m2, int(cnt.value)


def quantize(mean, m2, count, bits, v):
    v = _f32(v)
    mean = _f32(mean)
    m2 = _f32(m2) if m2 is not None else np.zeros_like(mean)
    single = v.ndim == 1
    v2 = v.reshape(1, -1) if single else v
    w = quantized_size(v2.shape[1], bits)
    out = np.zeros((v2.shape[0], w), np.uint64)
    for i in range(v2.shape[0]):
        row = np.ascontiguousarray(v2[i])
        lib().vso_quantize(_p(mean), _p(m2), count, bits, _p(row), row.size, C.c_void_p(out[i].ctypes.data))
    return out[0] if single else out


# ---- labels -------------------------------------------------------------------------------------------
def labelset(labels):
    a = np.array(labels, dtype=np.int16)
    n = lib().vso_labelset_from(_p(a), a.size) if a.size else 0
    return a[:n].copy()


def labels_overlap(a, b):
    a, b = labelset(a), labelset(b)
    return bool(lib().vso_labels_overlap(_p(a), a.size, _p(b), b.size))


def labels_contains_intersection(c, a, b):
    c, a, b = labelset(c), labelset(a), labelset(b)
    return bool(lib().vso_labels_contains_intersection(_p(c), c.size, _p(a), a.size, _p(b), b.size))


def smallint_array_overlap(left, right):
    """left/right: python lists with None for SQL NULL."""
    def enc(x):
        v = np.array([0 if e is None else e for e in x], dtype=np.int16)
        m = np.array([1 if e is None else 0 for e in x], dtype=np.uint8)
        return v, m
    lv, lm = enc(left)
    rv, rm = enc(right)
    return bool(lib().vso_smallint_array_overlap(_p(lv), _p(lm), lv.size, _p(rv), _p(rm), rv.size))


# ---- flat index -----------------------------------------------------------------------------------------
class OracleIndex:
    """Flat-array diskann index as the oracle sees it. Keeps numpy arrays alive for the C struct."""

    def __init__(self, *, codes, nbrs, heap_tids, vecs, mean, m2, count, bits, dim_index, num_neighbors,
                 distance_type, default_start, label_off=None, label_val=None, label_starts=None, storage_plain=False,
                 visible=None):
        self.codes = np.ascontiguousarray(codes, np.uint64)
        self.nbrs = np.ascontiguousarray(nbrs, np.uint32)
        self.heap_tids = np.ascontiguousarray(heap_tids, np.uint64)
        self.vecs = _f32(vecs)
        self.mean = _f32(mean)
        self.m2 = _f32(m2) if m2 is not None else np.zeros_like(self.mean)
        self.label_off = None if label_off is None else np.ascontiguousarray(label_off, np.uint32)
        self.label_val = None if label_val is None else np.ascontiguousarray(label_val, np.int16)
        ls = sorted((label_starts or {}).items())
        self.ls_labels = np.array([k for k, _ in ls], np.int16)
        self.ls_nodes = np.array([v for _, v in ls], np.uint32)
        n, w = self.codes.shape
        s = VsoIndex()
        s.n, s.dim_full, s.dim_index, s.bits, s.words = n, self.vecs.shape[1], dim_index, bits, w
        s.num_neighbors, s.nbr_stride = num_neighbors, self.nbrs.shape[1]
        s.distance_type, s.has_labels = distance_type, int(label_off is not None)
        s.default_start, s.n_label_starts = default_start, len(ls)
        s.label_start_labels, s.label_start_nodes = _p(self.ls_labels).value, _p(self.ls_nodes).value
        s.codes, s.nbrs, s.heap_tids, s.vecs = (_p(self.codes).value, _p(self.nbrs).value,
                                                _p(self.heap_tids).value, _p(self.vecs).value)
        s.label_off = None if self.label_off is None else _p(self.label_off).value
        s.label_val = None if self.label_val is None else _p(self.label_val).value
        s.mean, s.m2, s.count = _p(self.mean).value, _p(self.m2).value, count
        s.storage_plain = int(storage_plain)
        self.visible = None
        s.visible = None
        self.c = s
        self.set_visibility(visible)
        self.n, self.words, self.dim_full = n, w, self.vecs.shape[1]

    def set_visibility(self, visible):
        """per-node result of the heap fetch under the scan's snapshot (None: everything visible)"""
        self.visible = None if visible is None else np.ascontiguousarray(visible, np.uint8)
        assert self.visible is None or self.visible.shape == (self.c.n,)
        self.c.visible = None if self.visible is None else _p(self.visible).value

    def _labels_csr(self, qlabels, nq):
        if qlabels is None:
            return None, None
        off = np.zeros(nq + 1, np.uint32)
        vals = []
        for i, l in enumerate(qlabels):
            vals.extend(l)
            off[i + 1] = len(vals)
        return np.array(vals, np.int16), off

    def search_batch(self, queries, L=100, rescore=50, k=10, qlabels=None, threads=1):
        q = _f32(queries)
        nq = q.shape[0]
        lv, lo = self._labels_csr(qlabels, nq)
        nodes = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        st = VsoStats()
        lib().vso_search_batch(C.byref(self.c), _p(q), _p(lv), _p(lo), nq, L, rescore, k, threads, _p(nodes), _p(dist),
                               C.byref(st))
        return nodes, dist, st.as_dict()

    def stream_batch(self, queries, L=100, m=60, qlabels=None, threads=1):
        q = _f32(queries)
        nq = q.shape[0]
        lv, lo = self._labels_csr(qlabels, nq)
        nodes = np.empty((nq, m), np.uint32)
        ham = np.empty((nq, m), np.uint32)
        st = VsoStats()
        lib().vso_stream_batch(C.byref(self.c), _p(q), _p(lv), _p(lo), nq, L, m, threads, _p(nodes), _p(ham), C.byref(st))
        return nodes, ham, st.as_dict()

    def bruteforce(self, queries, k=10, threads=1):
        q = _f32(queries)
        nq = q.shape[0]
        nodes = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        lib().vso_bruteforce_topk(C.byref(self.c), _p(q), nq, k, threads, _p(nodes), _p(dist))
        return nodes, dist

    def scan(self, query, labels=None, L=100, rescore=50):
        return OracleScan(self, query, labels, L, rescore)


class OracleScan:
    """amrescan + amgettuple, one row at a time (AM/scan.rs:336-436)."""

    def __init__(self, idx, query, labels, L, rescore):
        self.idx = idx
        q = None if query is None else _f32(query)
        lv = None if labels is None else np.array(labels, np.int16)
        self._keep = (q, lv)
        self.h = lib().vso_scan_begin(C.byref(idx.c), _p(q), _p(lv), 0 if lv is None else lv.size,
                                      int(labels is not None), L, rescore)

    def gettuple(self):
        node, tid, d = C.c_uint32(), C.c_uint64(), C.c_float()
        ok = lib().vso_scan_gettuple(self.h, C.byref(node), C.byref(tid), C.byref(d))
        return (int(node.value), int(tid.value), np.float32(d.value)) if ok else None

    def next_sbq(self):
        node, tid, ham = C.c_uint32(), C.c_uint64(), C.c_uint32()
        ok = lib().vso_scan_next_sbq(self.h, C.byref(node), C.byref(tid), C.byref(ham))
        return (int(node.value), int(tid.value), int(ham.value)) if ok else None

    @property
    def xs_recheck(self):
        return bool(lib().vso_scan_xs_recheck(self.h))

    def stats(self):
        st = VsoStats()
        lib().vso_scan_stats(self.h, C.byref(st))
        return st.as_dict()

    def close(self):
        if self.h:
            lib().vso_scan_end(self.h)
            self.h = None

    def __del__(self):
        self.close()


def build_graph(codes, num_neighbors=50, nbr_stride=None, search_list_size=100, max_alpha=1.2):
    codes = np.ascontiguousarray(codes, np.uint64)
    n, w = codes.shape
    stride = nbr_stride or num_neighbors
    nbrs = np.empty((n, stride), np.uint32)
    start = C.c_uint32()
    lib().vso_build_graph(n, w, _p(codes), num_neighbors, stride, search_list_size, max_alpha, _p(nbrs), C.byref(start))
    return nbrs, int(start.value)


def heap_replay(ops):
    """ops: [(key, id)] pushes, (0xFFFFFFFF, 0) pops -> ids in pop order (the rest is popped at the end)"""
    a = np.ascontiguousarray(ops, np.uint32).reshape(-1, 2)
    out = np.empty(a.shape[0], np.uint32)
    k = lib().vso_heap_replay(_p(a), a.shape[0], _p(out))
    return out[:k].tolist()


def build_graph_labeled(codes, label_off, label_val, num_neighbors=50, nbr_stride=None, search_list_size=100, max_alpha=1.2):
    """Graph::insert over a labeled vector set -> (nbrs, default start, {label: start node})"""
    codes = np.ascontiguousarray(codes, np.uint64)
    lo = np.ascontiguousarray(label_off, np.uint32)
    lv = np.ascontiguousarray(label_val, np.int16)
    n, w = codes.shape
    stride = nbr_stride or num_neighbors
    nbrs = np.empty((n, stride), np.uint32)
    start = C.c_uint32()
    nl = max(len(set(lv.tolist())), 1)
    sl, sn = np.zeros(nl, np.int16), np.zeros(nl, np.uint32)
    k = lib().vso_build_graph_labeled(n, w, _p(codes), _p(lo), _p(lv), num_neighbors, stride, search_list_size, max_alpha,
                                      _p(nbrs), C.byref(start), _p(sl), _p(sn))
    return nbrs, int(start.value), {int(sl[i]): int(sn[i]) for i in range(k)}


def hamming_scan_topk(codes, qcodes, k, label_off=None, label_val=None, heap_tids=None, qlabels=None):
    """exact SBQ top-k, order (hamming, id); with label sets + one key per query (and heap tids) only the rows a label-filtered
    scan may return are ranked"""
    codes = np.ascontiguousarray(codes, np.uint64)
    qcodes = np.ascontiguousarray(qcodes, np.uint64)
    nq = qcodes.shape[0]
    nodes = np.empty((nq, k), np.uint32)
    ham = np.empty((nq, k), np.uint32)
    if qlabels is None and heap_tids is None:
        lib().vso_hamming_scan_topk(_p(codes), codes.shape[0], codes.shape[1], _p(qcodes), nq, k, _p(nodes), _p(ham))
        return nodes, ham
    lo = None if label_off is None else np.ascontiguousarray(label_off, np.uint32)
    lv = None if label_val is None else np.ascontiguousarray(label_val, np.int16)
    ht = None if heap_tids is None else np.ascontiguousarray(heap_tids, np.uint64)
    qo = qv = None
    if qlabels is not None:
        qo = np.zeros(nq + 1, np.uint32)
        flat = []
        for i, l in enumerate(qlabels):
            flat += sorted(set(l))
            qo[i + 1] = len(flat)
        qv = np.array(flat, np.int16)
    lib().vso_hamming_scan_topk_filtered(_p(codes), codes.shape[0], codes.shape[1], _p(lo), _p(lv), _p(ht), _p(qcodes), _p(qv), _p(qo),
                                         nq, k, _p(nodes), _p(ham))
    return nodes, ham
