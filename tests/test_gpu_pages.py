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

