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
