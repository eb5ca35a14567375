"""Index relation pages -> an index resident in HBM (host-side mirror of the vs_pages_* calls of include/vsgpu.h).

`IndexPages` stands for the main fork of a `diskann` index relation (Meta chain on block 0, SbqMeans chain, SbqNode
pages; UT/page.rs:28-39).  Blocks are appended in block order, decoded by libvsgpu on the host cores into the flat
arrays of `vs_index_host`, and `upload()` streams them to the device through the pinned staging ring.  What is NOT
read from the pages is what the reference itself keeps elsewhere or only the Rust side can decode: the heap's vector
column (`vecs`, needed for rerank).  The MetaPage body (geometry, start nodes, the SbqMeans pointer) is decoded by
`meta()` / `upload_from_meta()` (vs_pages_meta), or passed by hand to `upload()`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HeapAttr, HeapInfo, IndexDesc, IndexHost, MetaLayout, MetaPage, NodeLayout, PagesInfo, check

BLCKSZ = 8192
PAGE_SBQ_MEANS, PAGE_META = 7, 8


def _meta_layout(layout):
    """None, or a {"root_size": .., field: offset} dict as oracle/pages_py.meta_layout() gives / offset_of! prints"""
    if layout is None:
        return None
    lay = MetaLayout()
    lay.root_size = layout["root_size"]
    for k, _ in MetaLayout._fields_[1:]:
        setattr(lay, k, layout[k[4:]])
    return lay


def decode_meta_page(data, layout=None):
    """rkyv::from_bytes::<MetaPage> over the payload of the chain at (0, 2) -> (fields, {label: (block, offset)})"""
    L = _lib.load()
    buf = np.frombuffer(bytes(data), np.uint8)
    lay = _meta_layout(layout)
    m = MetaPage()
    check(L.vs_meta_page_decode(buf.ctypes.data_as(C.c_void_p), buf.size, None if lay is None else C.byref(lay), C.byref(m), None,
                                None, None, 0))
    n = int(m.n_labeled_start_nodes)
    lab, blk, off = np.empty(n, np.int16), np.empty(n, np.uint32), np.empty(n, np.uint32)
    check(L.vs_meta_page_decode(buf.ctypes.data_as(C.c_void_p), buf.size, None if lay is None else C.byref(lay), C.byref(m),
                                lab.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), n))
    return m.as_dict(), {int(l): (int(b), int(o)) for l, b, o in zip(lab, blk, off)}


def _meta_of(fn, handle, layout):
    lay = _meta_layout(layout)
    m, d = MetaPage(), IndexDesc()
    check(fn(handle, None if lay is None else C.byref(lay), C.byref(m), C.byref(d), None, None, 0))
    n = int(d.n_label_starts)
    lab, nodes = np.empty(n, np.int16), np.empty(n, np.uint32)
    check(fn(handle, None if lay is None else C.byref(lay), C.byref(m), C.byref(d), lab.ctypes.data_as(C.c_void_p),
             nodes.ctypes.data_as(C.c_void_p), n))
    return m.as_dict(), d, {int(l): int(v) for l, v in zip(lab, nodes)}


class HeapColumn:
    """The heap's vector column, staged from the pages of the table and its TOAST relation (vs_heap_*): `vecs[node]` = the
    vector of the heap tuple node's heap TID names — what the reference fetches per rescore candidate."""

    def __init__(self, attrs, vector_attno, dim, heap_tids, page_size=BLCKSZ):
        """attrs: [(attlen, attalign)] of the table's columns in attnum order; vector_attno 1-based"""
        self._L = _lib.load()
        self.page_size = page_size
        self._attrs = (HeapAttr * len(attrs))(*[HeapAttr(int(l), a.encode()) for l, a in attrs])
        self._tids = np.ascontiguousarray(heap_tids, np.uint64)
        self.vecs = np.empty((self._tids.size, dim), np.float32)
        h = C.c_void_p()
        check(self._L.vs_heap_open(page_size, self._attrs, len(attrs), vector_attno, dim, self._tids.ctypes.data_as(C.c_void_p),
                                   self._tids.size, self.vecs.ctypes.data_as(C.c_void_p), dim, C.byref(h)))
        self.h = h
        self._nb = [0, 0]

    def _add(self, fn, which, pages, first_block):
        buf = np.frombuffer(pages, np.uint8)
        if buf.size % self.page_size:
            raise ValueError(f"{buf.size} bytes is not a whole number of {self.page_size}-byte pages")
        nb = buf.size // self.page_size
        fb = self._nb[which] if first_block is None else first_block
        check(fn(self.h, fb, buf.ctypes.data_as(C.c_void_p), nb))
        self._nb[which] += nb

    def add(self, pages, first_block=None):
        self._add(self._L.vs_heap_add, 0, pages, first_block)

    def toast_add(self, pages, first_block=None):
        self._add(self._L.vs_heap_toast_add, 1, pages, first_block)

    def finish(self):
        info = HeapInfo()
        found = np.zeros(self._tids.size, np.uint8)
        check(self._L.vs_heap_finish(self.h, C.byref(info), found.ctypes.data_as(C.c_void_p)))
        return info.as_dict(), found

    def close(self):
        if self.h:
            self._L.vs_heap_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IndexPages:
    def __init__(self, has_labels=False, page_size=BLCKSZ, layout=None, threads=0, plain=False):
        """plain=True: a `plain` storage index (PlainNode items: full-precision vectors in the nodes, no SBQ codes, no labels)"""
        self._L = _lib.load()
        h = C.c_void_p()
        lay = None
        if layout is not None:
            lay = NodeLayout(*layout)
        self.plain = bool(plain)
        if plain:
            check(self._L.vs_pages_open_plain(page_size, None if lay is None else C.byref(lay), threads, C.byref(h)))
        else:
            check(self._L.vs_pages_open(page_size, int(has_labels), None if lay is None else C.byref(lay), threads, C.byref(h)))
        self.h = h
        self.page_size = page_size
        self.has_labels = bool(has_labels)
        self.n_blocks = 0
        self.info = None

    @staticmethod
    def default_layout(has_labels):
        lay = NodeLayout()
        check(_lib.load().vs_node_layout_default(int(has_labels), C.byref(lay)))
        return tuple(int(getattr(lay, k)) for k, _ in NodeLayout._fields_)

    def add(self, pages, first_block=None):
        """Append whole blocks (bytes-like, a multiple of page_size long)."""
        buf = np.frombuffer(pages, np.uint8)
        if buf.size % self.page_size:
            raise ValueError(f"{buf.size} bytes is not a whole number of {self.page_size}-byte pages")
        nb = buf.size // self.page_size
        fb = self.n_blocks if first_block is None else first_block
        check(self._L.vs_pages_add(self.h, fb, buf.ctypes.data_as(C.c_void_p), nb))
        self.n_blocks += nb

    def add_file(self, path, chunk_blocks=16384):
        """Append a relation segment file (base/<db>/<relfilenode>[.N]) chunk by chunk."""
        with open(path, "rb") as f:
            while True:
                b = f.read(chunk_blocks * self.page_size)
                if not b:
                    break
                self.add(b)

    def finish(self):
        info = PagesInfo()
        check(self._L.vs_pages_finish(self.h, C.byref(info)))
        self.info = info
        return info

    def node_of(self, block, offset):
        """IndexPointer -> node id (e.g. the MetaPage's start nodes)."""
        out = C.c_uint32()
        check(self._L.vs_pages_node_of(self.h, block, offset, C.byref(out)))
        return int(out.value)

    def item_pointer_of(self, node):
        b, o = C.c_uint32(), C.c_uint32()
        check(self._L.vs_pages_item_pointer_of(self.h, node, C.byref(b), C.byref(o)))
        return int(b.value), int(o.value)

    def read_chain(self, block, offset, page_type):
        n = C.c_size_t()
        check(self._L.vs_pages_read_chain(self.h, block, offset, page_type, None, 0, C.byref(n)))
        buf = np.empty(int(n.value), np.uint8)
        check(self._L.vs_pages_read_chain(self.h, block, offset, page_type, buf.ctypes.data_as(C.c_void_p), buf.size,
                                          C.byref(n)))
        return buf.tobytes()

    def sbq_means(self, block, offset):
        """SbqMeans::load at MetaPage.quantizer_metadata -> (count, mean, m2)"""
        dim, cnt = C.c_uint32(), C.c_uint64()
        check(self._L.vs_pages_sbq_means(self.h, block, offset, None, None, 0, C.byref(dim), C.byref(cnt)))
        mean = np.empty(dim.value, np.float32)
        m2 = np.empty(dim.value, np.float32)
        check(self._L.vs_pages_sbq_means(self.h, block, offset, mean.ctypes.data_as(C.c_void_p),
                                         m2.ctypes.data_as(C.c_void_p), dim.value, C.byref(dim), C.byref(cnt)))
        return int(cnt.value), mean, m2

    def meta(self, layout=None):
        """MetaPage::fetch (AM/meta_page.rs:380-403): -> (MetaPage fields, vs_index_desc with node-id start, {label: start node})"""
        if self.info is None:
            self.finish()
        return _meta_of(self._L.vs_pages_meta, self.h, layout)

    def upload_from_meta(self, ctx, vecs=None, meta_layout=None):
        """everything but the heap's vector column comes from the relation: geometry and start nodes from the MetaPage, the
        quantizer from the SbqMeans chain it points at, nodes from the SbqNode pages"""
        m, d, starts = self.meta(meta_layout)
        assert bool(m["has_labels"]) == self.has_labels
        has_q = m["quantizer_block"] != 0xFFFFFFFF
        return self.upload(ctx, dim_index=d.dim_index, bits=d.bits, distance_type=d.distance_type,
                           default_start=None if d.default_start == _lib.VS_INVALID_NODE else int(d.default_start),
                           quantizer_metadata=(m["quantizer_block"], m["quantizer_offset"]) if has_q else None,
                           vecs=vecs, label_starts=starts)

    def _host(self):
        if self.info is None:
            self.finish()
        h = IndexHost()
        check(self._L.vs_pages_host(self.h, C.byref(h)))
        return h

    def arrays(self):
        """Copies of the decoded flat arrays (tests / inspection)."""
        h = self._host()
        n, W, R = self.info.n_nodes, self.info.words, self.info.num_neighbors

        def view(ptr, dtype, count):
            if not ptr or count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), (count,)).copy()

        out = {"nbrs": view(h.nbrs, np.uint32, n * R).reshape(n, R), "heap_tids": view(h.heap_tids, np.uint64, n)}
        if self.plain:
            out["vecs"] = view(h.vecs, np.float32, n * W).reshape(n, W)  # PlainNode.vector
        else:
            out["codes"] = view(h.codes, np.uint64, n * W).reshape(n, W)
        if self.has_labels:
            out["label_off"] = view(h.label_off, np.uint32, n + 1)
            out["label_val"] = view(h.label_val, np.int16, int(self.info.n_label_vals))
        return out

    def upload_plain(self, ctx, *, distance_type, default_start, vecs=None):
        """a `plain` storage index in HBM: the graph from the pages; `vecs` = the heap's vector column (HeapColumn) or, when
        None, the node vectors themselves (the cosine-normalised index slice: enough when num_dimensions_to_index = num_dimensions)"""
        from .index import DiskAnnIndex
        assert self.plain
        a = self.arrays()
        node = self.node_of(*default_start) if isinstance(default_start, tuple) else default_start
        v = a["vecs"] if vecs is None else np.ascontiguousarray(vecs, np.float32)
        return DiskAnnIndex._upload_plain(ctx, nbrs=a["nbrs"], heap_tids=a["heap_tids"], vecs=v, num_neighbors=self.info.num_neighbors,
                                          distance_type=distance_type, default_start=_lib.VS_INVALID_NODE if node is None else int(node),
                                          dim_index=a["vecs"].shape[1])

    def upload(self, ctx, *, dim_index, bits, distance_type, default_start, quantizer_metadata=None, mean=None, m2=None,
               count=0, vecs=None, label_starts=None):
        """vs_index_upload of the decoded arrays.  default_start / label_starts values are IndexPointers (block, offset)
        or node ids; quantizer_metadata is the IndexPointer of the SbqMeans chain (or pass mean / m2 / count)."""
        from .index import DiskAnnIndex
        h = self._host()
        info = self.info
        if quantizer_metadata is not None:
            count, mean, m2 = self.sbq_means(*quantizer_metadata)
        mean = np.ascontiguousarray(mean, np.float32)
        m2 = None if m2 is None else np.ascontiguousarray(m2, np.float32)
        vecs = None if vecs is None else np.ascontiguousarray(vecs, np.float32)

        def node(x):
            return self.node_of(*x) if isinstance(x, tuple) else int(x)

        d = IndexDesc()
        d.n, d.dim_index, d.bits, d.words = info.n_nodes, dim_index, bits, info.words
        d.dim_full = dim_index if vecs is None else vecs.shape[1]
        d.num_neighbors, d.distance_type, d.has_labels = info.num_neighbors, distance_type, int(self.has_labels)
        d.default_start = _lib.VS_INVALID_NODE if default_start is None else node(default_start)
        ls = sorted((int(k), node(v)) for k, v in (label_starts or {}).items())
        d.n_label_starts = len(ls)
        lsl = np.array([k for k, _ in ls], np.int16)
        lsn = np.array([v for _, v in ls], np.uint32)
        h.vecs = None if vecs is None else vecs.ctypes.data
        h.mean = mean.ctypes.data
        h.m2 = None if m2 is None else m2.ctypes.data
        h.count = count
        h.label_start_labels = lsl.ctypes.data if ls else None
        h.label_start_nodes = lsn.ctypes.data if ls else None
        out = C.c_void_p()
        check(ctx._L.vs_index_upload(ctx.h, C.byref(d), C.byref(h), C.byref(out)))
        return DiskAnnIndex(ctx, out)

    def close(self):
        if self.h:
            self._L.vs_pages_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePages:
    """The same relation, decoded on the device (vs_pages_dev_*): blocks are copied to HBM as they are, a kernel decodes
    the SbqNode items (label sets included) into the index arrays.  memory_optimized indexes."""

    def __init__(self, ctx, n_blocks_total, page_size=BLCKSZ, layout=None):
        self.ctx = ctx
        self._L = ctx._L
        lay = None if layout is None else NodeLayout(*layout)
        h = C.c_void_p()
        check(self._L.vs_pages_dev_open(ctx.h, page_size, None if lay is None else C.byref(lay), n_blocks_total, C.byref(h)))
        self.h = h
        self.page_size = page_size
        self.n_blocks = 0
        self.info = None

    def add(self, pages, first_block=None):
        buf = np.frombuffer(pages, np.uint8)
        if buf.size % self.page_size:
            raise ValueError(f"{buf.size} bytes is not a whole number of {self.page_size}-byte pages")
        nb = buf.size // self.page_size
        fb = self.n_blocks if first_block is None else first_block
        check(self._L.vs_pages_dev_add(self.h, fb, buf.ctypes.data_as(C.c_void_p), nb))
        self.n_blocks += nb

    def node_of(self, block, offset):
        out = C.c_uint32()
        check(self._L.vs_pages_dev_node_of(self.h, block, offset, C.byref(out)))
        return int(out.value)

    def sbq_means(self, block, offset):
        dim, cnt = C.c_uint32(), C.c_uint64()
        check(self._L.vs_pages_dev_sbq_means(self.h, block, offset, None, None, 0, C.byref(dim), C.byref(cnt)))
        mean = np.empty(dim.value, np.float32)
        m2 = np.empty(dim.value, np.float32)
        check(self._L.vs_pages_dev_sbq_means(self.h, block, offset, mean.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p),
                                             dim.value, C.byref(dim), C.byref(cnt)))
        return int(cnt.value), mean, m2

    def meta(self, layout=None):
        """MetaPage::fetch on the metadata pages the host kept -> (fields, vs_index_desc, {label: start node})"""
        return _meta_of(self._L.vs_pages_dev_meta, self.h, layout)

    def build_from_meta(self, vecs=None, meta_layout=None):
        m, d, starts = self.meta(meta_layout)
        has_q = m["quantizer_block"] != 0xFFFFFFFF
        return self.build(words=d.words, num_neighbors=d.num_neighbors, dim_index=d.dim_index, bits=d.bits,
                          distance_type=d.distance_type,
                          default_start=None if d.default_start == _lib.VS_INVALID_NODE else int(d.default_start),
                          quantizer_metadata=(m["quantizer_block"], m["quantizer_offset"]) if has_q else None, vecs=vecs,
                          has_labels=bool(m["has_labels"]), label_starts=starts)

    def build(self, *, words, num_neighbors, dim_index, bits, distance_type, default_start, quantizer_metadata=None, mean=None,
              m2=None, count=0, vecs=None, has_labels=False, label_starts=None):
        """default_start / label_starts values: IndexPointer (block, offset) or node id; quantizer_metadata: IndexPointer of
        the SbqMeans chain."""
        from .index import DiskAnnIndex
        if quantizer_metadata is not None:
            count, mean, m2 = self.sbq_means(*quantizer_metadata)
        mean = np.ascontiguousarray(mean, np.float32)
        m2 = None if m2 is None else np.ascontiguousarray(m2, np.float32)
        vecs = None if vecs is None else np.ascontiguousarray(vecs, np.float32)
        d = IndexDesc()
        d.dim_index, d.bits, d.words, d.num_neighbors = dim_index, bits, words, num_neighbors
        d.dim_full = dim_index if vecs is None else vecs.shape[1]

        def node(x):
            return self.node_of(*x) if isinstance(x, tuple) else int(x)

        ls = sorted((int(k), node(v)) for k, v in (label_starts or {}).items())
        lsl = np.array([k for k, _ in ls], np.int16)
        lsn = np.array([v for _, v in ls], np.uint32)
        d.distance_type, d.has_labels, d.n_label_starts, d.storage_type = distance_type, int(bool(has_labels)), len(ls), 0
        d.default_start = _lib.VS_INVALID_NODE if default_start is None else node(default_start)
        h = IndexHost()
        h.label_start_labels = lsl.ctypes.data if ls else None
        h.label_start_nodes = lsn.ctypes.data if ls else None
        h.vecs = None if vecs is None else vecs.ctypes.data
        h.mean = mean.ctypes.data
        h.m2 = None if m2 is None else m2.ctypes.data
        h.count = count
        info = PagesInfo()
        out = C.c_void_p()
        check(self._L.vs_pages_dev_build(self.h, C.byref(d), C.byref(h), C.byref(info), C.byref(out)))
        self.info = info
        return DiskAnnIndex(self.ctx, out)

    def close(self):
        if self.h:
            self._L.vs_pages_dev_close(self.h)
            self.h = None
