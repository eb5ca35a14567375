"""K5 flat SBQ scan (vs_scan_topk) and the exact brute force (vs_bruteforce_topk) against numpy / the oracle.
Hamming ranking is bit-exact with ties broken by node id; brute-force distances are bit-identical to the oracle's
AVX2-order distance (asserted at the 1e-5 bar of north_star)."""
import ctypes as C

import numpy as np
import pytest

from helpers import cached_index

pytestmark = pytest.mark.gpu


def _popcount(a):
    a = a.copy()
    a = a - ((a >> np.uint64(1)) & np.uint64(0x5555555555555555))
    a = (a & np.uint64(0x3333333333333333)) + ((a >> np.uint64(2)) & np.uint64(0x3333333333333333))
    a = (a + (a >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return ((a * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.uint32)


def _flat_index(gpu_ctx, codes):
    import pgvectorscale_amd as P
    n, words = codes.shape
    return P.DiskAnnIndex.upload(gpu_ctx, codes=codes, nbrs=np.full((n, 4), 0xFFFFFFFF, np.uint32),
                                 heap_tids=np.ones(n, np.uint64), vecs=None, mean=np.zeros(words * 64, np.float32),
                                 m2=None, count=1, bits=1, dim_index=words * 64, num_neighbors=4, distance_type=P.VS_L2,
                                 default_start=0)


@pytest.mark.parametrize("words,n,nq,k", [(24, 20011, 19, 10), (4, 5000, 8, 64), (12, 70001, 3, 1), (48, 3001, 9, 17),
                                          (3, 100, 2, 10), (24, 7, 1, 10)])
def test_scan_topk_exact(gpu_ctx, words, n, nq, k):
    rng = np.random.default_rng(words * 1000 + n)
    # few distinct values per word => many Hamming ties, so the (hamming, id) tie rule is exercised
    codes = rng.integers(0, 4, (n, words), dtype=np.uint64) * np.uint64(0x0101010101010101)
    qcodes = rng.integers(0, 4, (nq, words), dtype=np.uint64) * np.uint64(0x0101010101010101)
    ix = _flat_index(gpu_ctx, codes)
    ids, ham = ix.scan_topk(qcodes, k)
    for q in range(nq):
        d = _popcount(codes ^ qcodes[q]).sum(axis=1).astype(np.int64)
        order = np.lexsort((np.arange(n), d))[:k]
        want_ids = np.full(k, 0xFFFFFFFF, np.uint32)
        want_ham = np.full(k, 0xFFFFFFFF, np.uint32)
        want_ids[:len(order)] = order
        want_ham[:len(order)] = d[order]
        assert (ids[q] == want_ids).all(), (q, ids[q], want_ids)
        assert (ham[q] == want_ham).all()
    ix.close()


def test_scan_topk_matches_oracle_hamming(gpu_ctx, oracle):
    ti = cached_index(n=3000, dim_full=96, R=24, seed=5)
    ix = ti.upload(gpu_ctx)
    Q = ti.queries(11, seed=3)
    qcodes = oracle.quantize(ti.mean, ti.m2, ti.count, ti.bits, Q)
    ids, ham = ix.scan_topk(qcodes, 10)
    want_ids, want_ham = oracle.hamming_scan_topk(ti.codes, qcodes, 10)
    assert (ids == want_ids).all()
    assert (ham == want_ham).all()
    ix.close()


@pytest.mark.parametrize("distance", ["l2", "cosine", "ip"])
def test_bruteforce_topk_matches_oracle(gpu_ctx, oracle, distance):
    O = oracle
    dt = {"l2": O.L2, "cosine": O.COSINE, "ip": O.IP}[distance]
    ti = cached_index(n=2500, dim_full=72, R=16, seed=9, distance=dt, kind="gauss")
    ix = ti.upload(gpu_ctx)
    Q = ti.queries(7, seed=4, kind="gauss")
    dq = gpu_ctx.alloc(Q.nbytes)
    gpu_ctx.upload(dq, Q)
    ids, dist = ix.bruteforce_topk(dq, len(Q), 10)
    gt_ids, gt_d = ti.oracle.bruteforce(Q, k=10)
    assert (ids == gt_ids).all()
    assert np.allclose(dist, gt_d, rtol=1e-5, atol=0)
    assert (dist.view(np.uint32) == np.asarray(gt_d, np.float32).view(np.uint32)).all()
    gpu_ctx.free(dq)
    ix.close()


def test_scan_topk_filtered_is_the_exact_filtered_ranking(gpu_ctx, oracle):
    """vs_scan_topk_filtered: exact SBQ top-k among the rows a label-filtered scan may return (label sets overlap, an empty key
    filters nothing, deleted tuples skipped with live_only) — against the oracle twin, with keys from common to absent labels,
    and as the upper bound of what the label-filtered graph walk finds (its stream is a subsequence of this ranking's rows)."""
    ti = cached_index(n=3000, dim_full=64, bits=2, R=32, distance=1, seed=11, kind="gauss", L_build=64, n_labels=6, deleted_frac=0.1)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(24, seed=3, kind="gauss")
    qcodes = ix.quantize(q)
    rng = np.random.default_rng(4)
    keys = [sorted(set(int(v) for v in rng.integers(1, 7, int(rng.integers(0, 3))))) for _ in range(len(q))]
    keys[0], keys[1], keys[2] = [], [6], [99]  # no filter / one label / a label nobody carries
    for live in (False, True):
        for k in (1, 10, 64):
            gi, gh = ix.scan_topk(qcodes, k, qlabels=keys, live_only=live)
            oi, oh = oracle.hamming_scan_topk(ti.codes, qcodes, k, label_off=ti.label_off, label_val=ti.label_val,
                                              heap_tids=ti.tids if live else None, qlabels=keys)
            assert (gi == oi).all() and (gh == oh).all()
    assert (gi[2] == 0xFFFFFFFF).all()
    # no key at all = the plain flat scan
    a, _ = ix.scan_topk(qcodes, 10)
    b, _ = ix.scan_topk(qcodes, 10, qlabels=[[] for _ in keys])
    assert (a == b).all()
    # the graph walk's SBQ-ordered stream only ever returns rows of the filtered set, at Hamming distances no smaller than the
    # exact ranking's at the same position
    si, sh, _ = ix.stream_batch(q[3:], search_list_size=100, m=10, qlabels=keys[3:])
    gi, gh = ix.scan_topk(qcodes[3:], 10, qlabels=keys[3:], live_only=True)
    live = sh != 0xFFFFFFFF
    assert (sh[live] >= gh[live]).all()
    ix.close()
