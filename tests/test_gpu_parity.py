"""GPU parity tests proper: every result of libvsgpu.so (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bars (BASELINE.json north_star): bit-exact for SBQ codes / Hamming distances / SBQ-ranked ids;
f32 rerank distances within 1e-5 relative (the kernels replay the reference's accumulation order, so they are in
fact expected to be bit-identical — reported, and asserted at the 1e-5 bar)."""
import numpy as np
import pytest

from helpers import TestIndex, cached_index, make_vectors

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: "f32 rerank distances within 1e-5 relative"


def _close(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)  # rows past the end of a scan carry NaN on both sides
    return np.all(nan | (np.abs(a - b) <= REL_TOL * np.maximum(np.abs(b), 1e-30) + 1e-12))


@pytest.mark.parametrize("dims,bits", [(128, 2), (768, 2), (1536, 1), (65, 1), (100, 3), (33, 2), (16000 // 8, 1)])
def test_quantize_bit_exact(gpu_ctx, oracle, dims, bits):
    O = oracle
    rng = np.random.default_rng(dims * 10 + bits)
    X = rng.standard_normal((300, dims)).astype(np.float32)
    X[:, 0] = 0.5  # zero-variance column: std_dev = 0 -> NaN / inf index paths (AM/sbq/quantize.rs:64-87)
    mean, m2, cnt = O.train(X, bits)
    w = O.quantized_size(dims, bits)
    import pgvectorscale_amd as P
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=np.zeros((4, w), np.uint64), nbrs=np.full((4, 4), 0xFFFFFFFF, np.uint32),
                               heap_tids=np.ones(4, np.uint64), vecs=None, mean=mean, m2=m2, count=cnt, bits=bits,
                               dim_index=dims, num_neighbors=4, distance_type=P.VS_L2, default_start=0)
    Q = rng.standard_normal((97, dims)).astype(np.float32)
    Q[0] = mean                      # exactly on the mean
    Q[1, 0] = 0.5                    # v == mean with std 0 -> NaN
    Q[2, 0] = 0.6                    # v > mean with std 0 -> +inf
    Q[3] = 1e30
    Q[4] = -1e30
    got = ix.quantize(Q)
    want = O.quantize(mean, m2, cnt, bits, Q)
    assert got.shape == want.shape and (got == want).all()
    ix.close()


@pytest.mark.parametrize("words", [1, 2, 3, 4, 12, 24, 25, 48, 250])
def test_hamming_gather_bit_exact(gpu_ctx, oracle, words):
    import pgvectorscale_amd as P
    rng = np.random.default_rng(words)
    n = 3000
    codes = rng.integers(0, 2 ** 64, (n, words), dtype=np.uint64)
    codes[5] = 0
    codes[6] = np.uint64(0xFFFFFFFFFFFFFFFF)
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=codes, nbrs=np.full((n, 4), 0xFFFFFFFF, np.uint32),
                               heap_tids=np.ones(n, np.uint64), vecs=None, mean=np.zeros(words * 64, np.float32),
                               m2=None, count=1, bits=1, dim_index=words * 64, num_neighbors=4, distance_type=P.VS_L2,
                               default_start=0)
    nq = 9
    qcodes = rng.integers(0, 2 ** 64, (nq, words), dtype=np.uint64)
    qcodes[0] = 0
    lists = [rng.integers(0, n, m).astype(np.uint32) for m in (0, 1, 15, 16, 17, 50, 64, 200, 1)]  # empty + ragged
    lists[8] = np.array([6], np.uint32)
    got = ix.hamming_gather(qcodes, lists)
    for qi in range(nq):
        want = [oracle.distance_xor(codes[i], qcodes[qi]) for i in lists[qi]]
        assert list(got[qi]) == want
    ix.close()


@pytest.mark.parametrize("distance", ["L2", "COSINE", "IP"])
@pytest.mark.parametrize("dims", [128, 768, 100, 36])
def test_rerank_matches_reference_order(gpu_ctx, oracle, distance, dims):
    O = oracle
    dt = getattr(O, distance)
    rng = np.random.default_rng(dims)
    n = 500
    X = (rng.standard_normal((n, dims)) * rng.uniform(0.1, 3, (n, 1))).astype(np.float32)
    X[3] = 0                                              # zero vector: left alone by preprocess_cosine
    X[4] = X[4] / np.linalg.norm(X[4])                    # already unit norm
    import pgvectorscale_amd as P
    w = O.quantized_size(dims, 1)
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=np.zeros((n, w), np.uint64), nbrs=np.full((n, 4), 0xFFFFFFFF, np.uint32),
                               heap_tids=np.ones(n, np.uint64), vecs=X, mean=np.zeros(dims, np.float32), m2=None,
                               count=1, bits=1, dim_index=dims, num_neighbors=4, distance_type=dt, default_start=0)
    Q = rng.standard_normal((6, dims)).astype(np.float32)
    lists = [rng.integers(0, n, m).astype(np.uint32) for m in (59, 0, 1, 8, 33, 300)]
    lists[2] = np.array([3], np.uint32)
    lists[3][:2] = [3, 4]
    got = ix.rerank(Q, lists)
    exact = total = 0
    for qi in range(6):
        q = Q[qi]
        if dt == O.COSINE:
            q = O.preprocess_cosine(q)[0]
        for j, node in enumerate(lists[qi]):
            v = X[node]
            if dt == O.COSINE:
                v = O.preprocess_cosine(v)[0]
            want = O.distance_by_type(dt, v, q)
            assert _close(got[qi][j], want), (qi, j, got[qi][j], want)
            exact += got[qi][j].tobytes() == want.tobytes()
            total += 1
    print(f"rerank {distance} d={dims}: {exact}/{total} bit-identical to the AVX2-order oracle")
    assert exact == total  # stronger than the bar: the kernel replays the accumulation order exactly
    ix.close()


CONFIGS = {
    # name: (index kwargs, query kind, L, rescore, k)
    "cfg1_like_l2": (dict(n=4000, dim_full=128, bits=2, R=50, distance=1, seed=1, kind="uniform", L_build=100), "uniform", 100, 50, 10),
    "cosine_768": (dict(n=1500, dim_full=768, R=50, distance=0, seed=3, kind="clustered", L_build=100), "clustered", 100, 50, 10),
    "ip_small": (dict(n=1200, dim_full=64, bits=2, R=32, distance=2, seed=5, kind="gauss", L_build=64), "gauss", 40, 20, 7),
    "odd_words_1bit": (dict(n=1000, dim_full=200, bits=1, R=20, distance=1, seed=6, kind="gauss", L_build=50), "gauss", 30, 10, 5),
    "three_bits": (dict(n=1000, dim_full=50, bits=3, R=24, distance=1, seed=7, kind="uniform", L_build=50), "uniform", 64, 0, 12),
    "matryoshka_cos": (dict(n=1000, dim_full=96, dim_index=64, bits=2, R=24, distance=0, seed=8, kind="gauss", L_build=50), "gauss", 60, 30, 10),
    "big_R": (dict(n=1500, dim_full=64, bits=2, R=80, distance=1, seed=9, kind="uniform", L_build=100), "uniform", 50, 25, 10),
    "tiny_L": (dict(n=800, dim_full=32, bits=2, R=16, distance=1, seed=2, kind="uniform", L_build=50), "uniform", 1, 3, 20),
}


@pytest.fixture(scope="module")
def uploaded(gpu_ctx):
    cache = {}

    def get(name):
        if name not in cache:
            kw = CONFIGS[name][0]
            ti = cached_index(**kw)
            cache[name] = (ti, ti.upload(gpu_ctx))
        return cache[name]
    yield get
    for _, ix in cache.values():
        ix.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_sbq_stream_bit_exact(uploaded, name):
    """The SBQ-ordered stream (ids AND Hamming distances AND work counters) must equal the oracle's exactly: this pins
    the Rust-BinaryHeap tie order, the visited-list insertion rule and the dedup semantics."""
    ti, ix = uploaded(name)
    _, qkind, L, rescore, k = CONFIGS[name]
    q = ti.queries(48, seed=123, kind=qkind)
    m = rescore + k + 5
    gi, gh, gst = ix.stream_batch(q, search_list_size=L, m=m)
    oi, oh, ost = ti.oracle.stream_batch(q, L=L, m=m)
    assert (gi == oi).all()
    assert (gh == oh).all()
    for a, b in (("visited_nodes", "visited_nodes"), ("candidate_nodes", "candidate_nodes"),
                 ("quantized_distance_comparisons", "quantized_distance_comparisons"), ("node_reads", "node_reads"),
                 ("next_calls", "next_calls")):
        assert gst[a] == ost[b], (a, gst[a], ost[b])


@pytest.mark.parametrize("name", list(CONFIGS))
def test_search_rows_match_oracle(uploaded, name):
    """Rows of the first k amgettuple calls: ids bit-exact, distances within 1e-5 relative."""
    ti, ix = uploaded(name)
    _, qkind, L, rescore, k = CONFIGS[name]
    q = ti.queries(64, seed=321, kind=qkind)
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=L, rescore=rescore, k=k)
    oi, od, ost = ti.oracle.search_batch(q, L=L, rescore=rescore, k=k)
    assert (gi == oi).all()
    assert (gt == ti.tids[np.minimum(gi, ti.n - 1)])[gi != 0xFFFFFFFF].all()
    if rescore == 0:
        assert np.isnan(gd).all() and np.isnan(od).all()
    else:
        assert _close(gd, od)
        print(f"{name}: {(gd.view(np.uint32) == od.view(np.uint32)).mean():.3f} of distances bit-identical")
    assert gst["full_distance_comparisons"] == ost["full_distance_comparisons"]
    assert gst["visited_nodes"] == ost["visited_nodes"]


def test_deleted_label_null_and_exhaustive(gpu_ctx, oracle):
    O = oracle
    ti = TestIndex(n=900, dim_full=48, bits=2, R=20, distance=O.L2, seed=41, kind="uniform", n_labels=6, deleted_frac=0.2)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(12, seed=77)
    # unfiltered with deletions
    gi, _, gd, _ = ix.search_batch(q, search_list_size=30, rescore=10, k=15)
    oi, od, _ = ti.oracle.search_batch(q, L=30, rescore=10, k=15)
    assert (gi == oi).all() and _close(gd, od)
    dead = set(np.nonzero((ti.tids & np.uint64(0xFFFF)) == 0)[0].tolist())
    assert not (set(gi.ravel().tolist()) & dead)
    # label keys: one label, two labels, unsorted with duplicates, a label nobody has, the empty key
    keys = [[3], [2, 5], [5, 2, 5], [99], [], [1], [6, 1], [4], [3, 4], [2], [5], [1, 2, 3, 4, 5, 6]]
    gi, _, gd, gst = ix.search_batch(q, search_list_size=30, rescore=10, k=15, qlabels=keys)
    oi, od, ost = ti.oracle.search_batch(q, L=30, rescore=10, k=15, qlabels=keys)
    assert (gi == oi).all() and _close(gd, od)
    assert (gi[3] == 0xFFFFFFFF).all() and (gi[4] == 0xFFFFFFFF).all()
    assert gst["quantized_distance_comparisons"] == ost["quantized_distance_comparisons"]
    # exhaustive scan through the amgettuple mirror: every live row exactly once, same order as the oracle
    scan = ix.beginscan()
    scan.rescan(q[0], search_list_size=2, rescore=4)
    os_ = ti.oracle.scan(q[0], L=2, rescore=4)
    rows = []
    while True:
        r = scan.gettuple()
        o = os_.gettuple()
        assert (r is None) == (o is None)
        if r is None:
            break
        assert r[1] == o[0] and r[0] == o[1]
        rows.append(r[1])
    assert len(rows) == 900 - len(dead) and len(set(rows)) == len(rows)
    assert not scan.xs_recheck
    # NULL query (AM/build.rs:2015-2044) and a labelled rescan on the same scan descriptor
    scan.rescan(None, search_list_size=5, rescore=50)
    c = 0
    while scan.gettuple() is not None:
        c += 1
    assert c == 900 - len(dead)
    scan.rescan(q[1], labels=[2, 5], search_list_size=30, rescore=10)
    assert scan.xs_recheck
    os_ = ti.oracle.scan(q[1], labels=[2, 5], L=30, rescore=10)
    for _ in range(40):
        r, o = scan.gettuple(), os_.gettuple()
        assert (r is None) == (o is None)
        if r is None:
            break
        assert r[1] == o[0] and _close(r[2], o[2])
    scan.endscan()
    # vacuum-style deletion after upload
    victims = [int(x) for x in gi[0][:3] if x != 0xFFFFFFFF]
    ix.mark_deleted(victims)
    ti.tids[victims] &= ~np.uint64(0xFFFF)
    gi2, _, _, _ = ix.search_batch(q[:1], search_list_size=30, rescore=10, k=15)
    oi2, _, _ = ti.oracle.search_batch(q[:1], L=30, rescore=10, k=15)
    assert (gi2 == oi2).all() and not (set(gi2.ravel().tolist()) & set(victims))
    ix.close()


def test_empty_graph_and_error_paths(gpu_ctx):
    import pgvectorscale_amd as P
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=np.zeros((1, 4), np.uint64), nbrs=np.full((1, 50), 0xFFFFFFFF, np.uint32),
                               heap_tids=np.ones(1, np.uint64), vecs=np.zeros((1, 128), np.float32),
                               mean=np.zeros(128, np.float32), m2=np.ones(128, np.float32), count=1, bits=2, dim_index=128,
                               num_neighbors=50, distance_type=P.VS_L2, default_start=P.VS_INVALID_NODE)
    gi, _, _, _ = ix.search_batch(np.zeros((2, 128), np.float32), k=3)
    assert (gi == 0xFFFFFFFF).all()   # ListSearchResult::empty() (AM/graph/mod.rs:337-341)
    with pytest.raises(P.VsError):
        ix.search_batch(np.zeros((1, 128), np.float32), search_list_size=0)
    with pytest.raises(P.VsError):
        ix.search_batch(np.zeros((1, 128), np.float32), qlabels=[[1]])  # label key on an unlabeled index
    ix.close()
    # duplicate ids in one neighbor list are rejected at upload
    nb = np.full((3, 4), 0xFFFFFFFF, np.uint32)
    nb[0, :2] = [1, 1]
    with pytest.raises(P.VsError):
        P.DiskAnnIndex.upload(gpu_ctx, codes=np.zeros((3, 1), np.uint64), nbrs=nb, heap_tids=np.ones(3, np.uint64),
                              vecs=None, mean=np.zeros(64, np.float32), m2=None, count=1, bits=1, dim_index=64,
                              num_neighbors=4, distance_type=P.VS_L2, default_start=0)
