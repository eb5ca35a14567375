"""The staging path end to end: index relation pages (written byte by byte the way the reference lays them out) ->
vs_pages_* decode on the host -> vs_index_upload (pinned ring, hipMemcpyAsync) -> scans on the MI355X; the rows must be
the oracle's rows for the arrays the pages were written from."""
import numpy as np
import pytest

from helpers import TestIndex
from oracle import pages_py as PG

pytestmark = pytest.mark.gpu


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return np.all(nan | (np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-30) + 1e-12))


@pytest.mark.parametrize("labeled", [False, True])
def test_index_from_pages_searches_like_the_oracle(gpu_ctx, oracle, labeled):
    from pgvectorscale_amd.pages import IndexPages
    O = oracle
    ti = TestIndex(n=1400, dim_full=96, dim_index=64, bits=2, R=24, distance=O.COSINE if labeled else O.L2, seed=17,
                   kind="gauss", n_labels=5 if labeled else 0, deleted_frac=0.1, L_build=50)
    w = PG.write_index(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, mean=ti.mean, m2=ti.m2, count=ti.count,
                       label_off=ti.label_off, label_val=ti.label_val, zero_page_every=500)
    pages = IndexPages(has_labels=labeled)
    data = w.rel.tobytes()
    half = (len(w.rel.pages) // 2) * PG.BLCKSZ
    pages.add(data[:half])      # blocks arrive in pieces, as a reader looping over the relation would hand them over
    pages.add(data[half:])
    info = pages.finish()
    assert (info.n_nodes, info.words, info.num_neighbors) == (ti.n, ti.codes.shape[1], ti.R)
    ix = pages.upload(gpu_ctx, dim_index=ti.dim_index, bits=ti.bits, distance_type=ti.distance,
                      default_start=w.node_ptrs[ti.start],                      # StartNodes.default_node as an IndexPointer
                      label_starts={l: w.node_ptrs[v] for l, v in ti.label_starts.items()},
                      quantizer_metadata=w.means_ptr, vecs=ti.vecs)
    pages.close()  # the decoded host arrays are no longer needed once the index is in HBM
    # what landed in HBM is what the pages were written from
    dev = ix.download()
    assert (dev["codes"] == ti.codes).all() and (dev["nbrs"] == ti.nbrs).all() and (dev["heap_tids"] == ti.tids).all()
    mean, m2, cnt = ix.get_quantizer()
    assert (mean == ti.mean).all() and (m2 == ti.m2).all() and cnt == ti.count
    q = ti.queries(40, seed=5, kind="gauss")
    keys = None
    if labeled:
        rng = np.random.default_rng(3)
        keys = [sorted(set(int(x) for x in rng.integers(1, 6, int(rng.integers(1, 3))))) for _ in range(40)]
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=40, rescore=20, k=10, qlabels=keys)
    oi, od, ost = ti.oracle.search_batch(q, L=40, rescore=20, k=10, qlabels=keys)
    assert (gi == oi).all()
    assert _close(gd, od)
    assert (gt == ti.tids[np.minimum(gi, ti.n - 1)])[gi != 0xFFFFFFFF].all()
    assert gst["visited_nodes"] == ost["visited_nodes"]
    assert gst["quantized_distance_comparisons"] == ost["quantized_distance_comparisons"]
    ix.close()



@pytest.mark.parametrize("on_device", [False, True])
def test_index_from_the_relation_alone(gpu_ctx, oracle, on_device):
    """nothing is passed by hand but the heap's vector column: geometry, distance type, default and labeled start nodes and the
    pointer to the SbqMeans chain are decoded from the MetaPage (vs_pages_meta / vs_pages_dev_meta), host and device decoders"""
    from pgvectorscale_amd.pages import DevicePages, IndexPages
    O = oracle
    ti = TestIndex(n=1200, dim_full=80, dim_index=64, bits=2, R=20, distance=O.COSINE, seed=23, kind="gauss", n_labels=5,
                   deleted_frac=0.05, L_build=40)
    meta = dict(num_dimensions=ti.dim_full, num_dimensions_to_index=ti.dim_index, bq_num_bits_per_dimension=ti.bits,
                distance_type=ti.distance, num_neighbors=ti.R, default_start=ti.start, labeled_starts=dict(ti.label_starts))
    w = PG.write_index(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, mean=ti.mean, m2=ti.m2, count=ti.count,
                       label_off=ti.label_off, label_val=ti.label_val, meta=meta)
    data = w.rel.tobytes()
    if on_device:
        pages = DevicePages(gpu_ctx, len(w.rel.pages))
        pages.add(data)
        ix = pages.build_from_meta(vecs=ti.vecs)
    else:
        pages = IndexPages(has_labels=True)
        pages.add(data)
        ix = pages.upload_from_meta(gpu_ctx, vecs=ti.vecs)
    pages.close()
    d = ix.desc
    assert (d.n, d.dim_full, d.dim_index, d.bits, d.num_neighbors, d.distance_type, d.default_start) == \
        (ti.n, ti.dim_full, ti.dim_index, ti.bits, ti.R, ti.distance, ti.start)
    q = ti.queries(24, seed=6, kind="gauss")
    rng = np.random.default_rng(4)
    keys = [sorted(set(int(x) for x in rng.integers(1, 6, int(rng.integers(1, 3))))) for _ in range(24)]
    for kk in (None, keys):
        gi, _, gd, gst = ix.search_batch(q, search_list_size=30, rescore=15, k=10, qlabels=kk)
        oi, od, ost = ti.oracle.search_batch(q, L=30, rescore=15, k=10, qlabels=kk)
        assert (gi == oi).all() and _close(gd, od)
        assert gst["visited_nodes"] == ost["visited_nodes"]
    ix.close()
