"""`plain` storage (AM/plain/storage.rs): the same streaming beam search, but candidates are scored with
the full-precision distance to the vector stored in the node (the reference's AVX2 accumulation order) instead of SBQ
Hamming, and there is no resort.  Rows (ids AND f32 distances, bit for bit) and work counters must equal the oracle's."""
import numpy as np
import pytest

from helpers import make_vectors
from oracle import oracle_py as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


class PlainIndex:
    def __init__(self, n, dim, R, distance, seed, kind, deleted_frac=0.0, dim_index=None):
        self.vecs = make_vectors(n, dim, seed, kind)
        if kind == "gauss":
            self.vecs *= np.random.default_rng(seed).uniform(0.2, 3.0, (n, 1)).astype(np.float32)  # un-normalised rows
            self.vecs[3] = 0                                                                             # a zero vector
        # any connected graph will do for parity: use the SBQ builder of the oracle on 2-bit codes of the same vectors
        mean, m2, cnt = O.train(self.vecs, 2)
        codes = O.quantize(mean, m2, cnt, 2, self.vecs)
        self.nbrs, self.start = O.build_graph(codes, num_neighbors=R, search_list_size=50)
        self.tids = ((np.arange(n, dtype=np.uint64) + 11) << np.uint64(16)) | np.uint64(1)
        if deleted_frac:
            self.tids[np.random.default_rng(seed + 1).random(n) < deleted_frac] &= ~np.uint64(0xFFFF)
        self.n, self.dim, self.R, self.distance, self.dim_index = n, dim, R, distance, dim_index or dim
        if self.dim_index != dim:  # the oracle derives its code width from dim_index
            mean, m2, cnt = O.train(np.ascontiguousarray(self.vecs[:, :self.dim_index]), 2)
            codes = O.quantize(mean, m2, cnt, 2, np.ascontiguousarray(self.vecs[:, :self.dim_index]))
        self.oracle = O.OracleIndex(codes=codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=mean, m2=m2, count=cnt,
                                    bits=2, dim_index=self.dim_index, num_neighbors=R, distance_type=distance,
                                    default_start=self.start, storage_plain=True)

    def upload(self, ctx):
        import pgvectorscale_amd as P
        from pgvectorscale_amd import _lib
        return P.DiskAnnIndex.upload(ctx, codes=None, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=None, m2=None,
                                     count=0, bits=None, dim_index=self.dim_index, num_neighbors=self.R, distance_type=self.distance,
                                     default_start=self.start, storage_type=_lib.VS_STORAGE_PLAIN)


@pytest.mark.parametrize("dim,R,distance,kind,L", [(128, 32, O.L2, "uniform", 40), (100, 24, O.COSINE, "gauss", 30),
                                                   (36, 50, O.IP, "gauss", 25), (768, 50, O.COSINE, "clustered", 30)])
def test_plain_storage_rows_match_the_oracle(gpu_ctx, oracle, dim, R, distance, kind, L):
    pi = PlainIndex(n=1200, dim=dim, R=R, distance=distance, seed=dim, kind=kind, deleted_frac=0.1)
    ix = pi.upload(gpu_ctx)
    q = make_vectors(40, dim, 9, kind)
    k = 25
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=L, rescore=50, k=k)  # rescore is ignored: no resort for plain storage
    oi, od, ost = pi.oracle.search_batch(q, L=L, rescore=50, k=k)
    assert (gi == oi).all()
    assert (gd.view(np.uint32) == od.view(np.uint32)).all()
    live = gi != 0xFFFFFFFF
    assert (gt[live] == pi.tids[gi[live]]).all()
    for key in ("visited_nodes", "candidate_nodes", "full_distance_comparisons", "node_reads", "next_calls"):
        assert gst[key] == ost[key], key
    assert gst["quantized_distance_comparisons"] == 0
    # the raw stream: same ids, and the graph distances are the f32 distances
    si, sd, _ = ix.stream_batch(q, search_list_size=L, m=k)
    assert (si == oi).all() and (sd[si != 0xFFFFFFFF] == od.view(np.uint32)[si != 0xFFFFFFFF]).all()
    # the amgettuple mirror, including a NULL query (AM/build.rs:2015-2044)
    scan = ix.beginscan()
    scan.rescan(q[0], search_list_size=L, rescore=50)
    os_ = pi.oracle.scan(q[0], L=L, rescore=50)
    for _ in range(40):
        r, o = scan.gettuple(), os_.gettuple()
        assert (r is None) == (o is None)
        if r is None:
            break
        assert r[1] == o[0] and np.float32(r[2]).view(np.uint32) == np.float32(o[2]).view(np.uint32)
    scan.rescan(None, search_list_size=5, rescore=0)
    os_ = pi.oracle.scan(None, L=5, rescore=0)
    cnt = 0
    while True:  # every live row the start node reaches, in the oracle's order
        r, o = scan.gettuple(), os_.gettuple()
        assert (r is None) == (o is None)
        if r is None:
            break
        assert r[1] == o[0]
        cnt += 1
    assert cnt > 0
    scan.endscan()
    # label keys are refused, as in the reference
    import pgvectorscale_amd as P
    with pytest.raises(P.VsError, match="label"):
        ix.search_batch(q[:2], search_list_size=L, rescore=0, k=5, qlabels=[[1], [2]])
    ix.close()


@pytest.mark.parametrize("distance", [O.COSINE, O.L2])
def test_plain_storage_with_truncated_index_dimensions(gpu_ctx, oracle, distance):
    """num_dimensions_to_index < num_dimensions: the graph search compares index slices (normalised on their own for cosine),
    then the rows are resorted on the full vectors like an SBQ scan (amgettuple, Plain arm, AM/scan.rs:392-401)."""
    pi = PlainIndex(n=1000, dim=96, R=24, distance=distance, seed=5, kind="gauss", deleted_frac=0.1, dim_index=64)
    ix = pi.upload(gpu_ctx)
    q = make_vectors(32, 96, 3, "gauss")
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=30, rescore=15, k=12)
    oi, od, ost = pi.oracle.search_batch(q, L=30, rescore=15, k=12)
    assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    for key in ("visited_nodes", "candidate_nodes", "full_distance_comparisons", "node_reads"):
        assert gst[key] == ost[key], key
    ix.close()


def test_plain_index_from_relation_pages(gpu_ctx, oracle):
    """a `plain` storage index staged from its relation's pages (PageType::Node, PlainNode items): the graph and the node vectors
    come from vs_pages_open_plain, geometry and start node from the MetaPage; scans equal the oracle's bit for bit"""
    from oracle import pages_py as PG
    from pgvectorscale_amd.pages import IndexPages
    pi = PlainIndex(n=900, dim=64, R=24, distance=O.L2, seed=5, kind="uniform", deleted_frac=0.05)
    meta = dict(num_dimensions=64, storage_type=0, bq_num_bits_per_dimension=1, distance_type=O.L2, num_neighbors=24, default_start=pi.start)
    w = PG.write_plain_index(vectors=pi.vecs, nbrs=pi.nbrs, heap_tids=pi.tids, meta=meta)
    pages = IndexPages(plain=True)
    pages.add(w.rel.tobytes())
    pages.finish()
    m, d, _ = pages.meta()
    ix = pages.upload_plain(gpu_ctx, distance_type=d.distance_type, default_start=int(d.default_start))
    pages.close()
    q = make_vectors(24, 64, 3, "uniform")
    gi, _, gd, gst = ix.search_batch(q, search_list_size=30, rescore=0, k=15)
    oi, od, ost = pi.oracle.search_batch(q, L=30, rescore=0, k=15)
    assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert gst["visited_nodes"] == ost["visited_nodes"] and gst["full_distance_comparisons"] == ost["full_distance_comparisons"]
    ix.close()
