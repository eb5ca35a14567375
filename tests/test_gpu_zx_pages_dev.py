"""vs_pages_dev_*: index relation pages decoded ON the device (a wave per node page) — see tests/test_gpu_pages.py for the host
reader.  (Sorted late in the GPU tier on purpose: the decode kernel is newer than the last hardware run.)"""
import numpy as np
import pytest

from helpers import TestIndex
from oracle import pages_py as PG

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return np.all(nan | (np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-30) + 1e-12))


def test_pages_decoded_on_the_device(gpu_ctx, oracle):
    """vs_pages_dev_*: the blocks go to HBM as they are and a kernel decodes the SbqNode items; the index arrays must be the
    ones the pages were written from, scans must return the oracle's rows, malformed pages must be refused with a reason."""
    import struct

    import pgvectorscale_amd as P
    from pgvectorscale_amd.pages import DevicePages
    O = oracle
    ti = TestIndex(n=1400, dim_full=96, dim_index=64, bits=2, R=24, distance=O.L2, seed=18, kind="gauss", deleted_frac=0.1, L_build=50)
    w = PG.write_index(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, mean=ti.mean, m2=ti.m2, count=ti.count, zero_page_every=300,
                       means_first=False)
    data = w.rel.tobytes()
    nblk = len(w.rel.pages)

    def stage(raw):
        dp = DevicePages(gpu_ctx, nblk)
        third = (nblk // 3) * PG.BLCKSZ
        dp.add(raw[:third])
        dp.add(raw[third:])
        return dp

    dp = stage(data)
    ix = dp.build(words=ti.codes.shape[1], num_neighbors=ti.R, dim_index=ti.dim_index, bits=ti.bits, distance_type=ti.distance,
                  default_start=w.node_ptrs[ti.start], quantizer_metadata=w.means_ptr, vecs=ti.vecs)
    assert dp.info.n_nodes == ti.n and dp.node_of(*w.node_ptrs[5]) == 5
    dp.close()
    dev = ix.download()
    assert (dev["codes"] == ti.codes).all() and (dev["nbrs"] == ti.nbrs).all() and (dev["heap_tids"] == ti.tids).all()
    mean, m2, cnt = ix.get_quantizer()
    assert (mean == ti.mean).all() and (m2 == ti.m2).all() and cnt == ti.count
    q = ti.queries(32, seed=6, kind="gauss")
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=40, rescore=20, k=10)
    oi, od, ost = ti.oracle.search_batch(q, L=40, rescore=20, k=10)
    assert (gi == oi).all() and _close(gd, od) and gst["visited_nodes"] == ost["visited_nodes"]
    ix.close()

    # malformed input: a neighbor that points at the meta page, a code of another width, a dead line pointer
    blk, off = w.node_ptrs[40]
    s, l = w.rel.item_span(blk, off)
    base = blk * PG.BLCKSZ
    cases = []
    bad = bytearray(data)
    fld = base + s + l - 32 + 16
    rel_off, _ = struct.unpack_from("<iI", bad, fld)
    bad[fld + rel_off:fld + rel_off + 8] = PG.rkyv_item_pointer(0, 1)
    cases.append((bytes(bad), "not an SbqNode item"))
    bad = bytearray(data)
    struct.pack_into("<I", bad, base + s + l - 32 + 8 + 4, 1)
    cases.append((bytes(bad), "code width"))
    bad = bytearray(data)
    lp = struct.unpack_from("<I", bad, base + 24 + 4 * (off - 1))[0]
    struct.pack_into("<I", bad, base + 24 + 4 * (off - 1), lp | (3 << 15))
    cases.append((bytes(bad), "LP_NORMAL"))
    for raw, why in cases:
        dp = stage(raw)
        with pytest.raises(P.VsError, match=why):
            dp.build(words=ti.codes.shape[1], num_neighbors=ti.R, dim_index=ti.dim_index, bits=ti.bits, distance_type=ti.distance,
                     default_start=w.node_ptrs[ti.start], quantizer_metadata=w.means_ptr, vecs=ti.vecs)
        dp.close()


def test_labeled_pages_decoded_on_the_device(gpu_ctx, oracle):
    """LabeledSbqNode items: the label sets come off the staged pages in two more passes (count, copy); label-filtered scans
    return the oracle's rows; a label set that is not strictly increasing is refused."""
    import struct

    import pgvectorscale_amd as P
    from pgvectorscale_amd.pages import DevicePages
    O = oracle
    ti = TestIndex(n=1300, dim_full=80, dim_index=64, bits=2, R=20, distance=O.COSINE, seed=23, kind="gauss", n_labels=6,
                   deleted_frac=0.05, L_build=50)
    w = PG.write_index(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, mean=ti.mean, m2=ti.m2, count=ti.count,
                       label_off=ti.label_off, label_val=ti.label_val, zero_page_every=400)
    data = w.rel.tobytes()
    nblk = len(w.rel.pages)

    def build(raw):
        dp = DevicePages(gpu_ctx, nblk)
        dp.add(raw)
        try:
            return dp.build(words=ti.codes.shape[1], num_neighbors=ti.R, dim_index=ti.dim_index, bits=ti.bits,
                            distance_type=ti.distance, default_start=w.node_ptrs[ti.start], quantizer_metadata=w.means_ptr,
                            vecs=ti.vecs, has_labels=True, label_starts={l: w.node_ptrs[v] for l, v in ti.label_starts.items()})
        finally:
            dp.close()

    ix = build(data)
    dev = ix.download()
    assert (dev["codes"] == ti.codes).all() and (dev["nbrs"] == ti.nbrs).all() and (dev["heap_tids"] == ti.tids).all()
    from pgvectorscale_amd import _lib
    lo = gpu_ctx.download(ix.array(_lib.ARR_LABEL_OFF)[0], np.empty(ti.n + 1, np.uint32))
    lv = gpu_ctx.download(ix.array(_lib.ARR_LABEL_VAL)[0], np.empty(len(ti.label_val), np.int16))
    assert (lo == ti.label_off).all() and (lv == ti.label_val).all()
    q = ti.queries(40, seed=8, kind="gauss")
    rng = np.random.default_rng(4)
    keys = [sorted(set(int(x) for x in rng.integers(1, 7, int(rng.integers(1, 3))))) for _ in range(40)]
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=40, rescore=20, k=10, qlabels=keys)
    oi, od, ost = ti.oracle.search_batch(q, L=40, rescore=20, k=10, qlabels=keys)
    assert (gi == oi).all() and _close(gd, od) and gst["visited_nodes"] == ost["visited_nodes"]
    ix.close()

    # a node with at least two labels: swap the first two
    node = next(i for i in range(ti.n) if ti.label_off[i + 1] - ti.label_off[i] >= 2)
    blk, off = w.node_ptrs[node]
    s, l = w.rel.item_span(blk, off)
    lay = P.pages.IndexPages.default_layout(True)
    fld = blk * PG.BLCKSZ + s + l - lay[0] + lay[4]
    rel_off, cnt = struct.unpack_from("<iI", data, fld)
    assert cnt == ti.label_off[node + 1] - ti.label_off[node]
    bad = bytearray(data)
    a, b = struct.unpack_from("<hh", bad, fld + rel_off)
    struct.pack_into("<hh", bad, fld + rel_off, b, a)
    with pytest.raises(P.VsError, match="strictly increasing"):
        build(bytes(bad))
