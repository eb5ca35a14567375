#!/usr/bin/env python3
"""Differential fuzzing of the search path against the oracle on the wave64 interpreter (tests/emu; no GPU needed).

Random index geometries (size, dimensions, bits, R, distance, labels, deleted tuples, matryoshka slice) x random scan
parameters (L, rescore, k, label keys, NULL queries) x the kernels and regimes the library can be steered into
(k_search_fast regimes, the general kernel).  Every case must reproduce the oracle: ids bit for bit, distances
within 1e-5, the SBQ stream (ids + Hamming distances) and the work counters exactly.

  make -C tests/emu && python scripts/fuzz_emu.py --seconds 600 [--seed 1] [--gpu]

--gpu runs the same cases against libvsgpu.so on a real device instead.  A failing case prints the line that reproduces
it (`--only <case seed>`)."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

REGIMES = [
    {},
    {"VS_F_LDS_MAX_INS": "0"},
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VR": "8", "VS_F_MINW": "4"},
    {"VS_F_LH": "256"},
    {"VS_F_VR": "0", "VS_F_VCAP": "64"},
    {"VS_F_HL": "63"},
    {"VS_F_HL": "63", "VS_F_LDS_MAX_INS": "0"},
    {"VS_F_LH": "256", "VS_F_POOL": "0.01"},
    {"VS_FAST": "0"},
    {"VS_FAST": "0", "VS_HL": "64", "VS_G0": "256"},
    {"VS_F_LDS_MAX_INS": "0", "VS_F_GCAP": "512"},                      # tiny global dedup table: second attempt of k_search_fast
    {"VS_F_LDS_MAX_INS": "0", "VS_F_GCAP": "512", "VS_F_RETRY": "0"},   # ... or straight to the general kernel
]
# drawn on top of the four regimes of a case (by a generator of their own, so that the cases of older seeds stay what they were)
EXTRA_REGIMES = [
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "1"},                      # written-bucket bitmap instead of cleared dedup tables
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "1", "VS_F_HL": "63"},
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "1", "VS_F_GCAP": "2048"},  # ... tight: chains of full buckets, second attempts
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "2"},                      # occupancy bit per slot (linear probing, loads only behind occupied home slots)
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "2", "VS_F_GCAP": "2048"},  # ... tight: long occupied runs across groups, wrap-around, second attempts
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "3"},                      # 16-bit entries in buckets of eight + overflow table
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "3", "VS_F_GCAP": "2048"},  # ... tight: full buckets, overflow inserts / lookups, second attempts
    {"VS_F_LDS_MAX_INS": "0", "VS_F_VIRGIN": "3", "VS_F_GCAP": "1024", "VS_F_GLOAD_PCT": "90"},
]
TUNING = sorted({k for r in REGIMES + EXTRA_REGIMES for k in r})


def close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-30) + 1e-12)))


def gettuple_mirror(ix, oracle, q, keys, L, rescore, rng, where, exact_dist):
    """amrescan / amgettuple one row at a time: two scans on one handle (the second a rescan, sometimes a NULL query),
    a random number of rows each, sometimes to exhaustion"""
    # one scan in three runs behind a dispatcher: first rows from a shared launch, the rest from its cursor on the dispatcher thread
    broker = None
    if np.random.default_rng(int(rng.bit_generator.state["state"]["state"]) % (2 ** 32) + 3).random() < 0.35:
        import pgvectorscale_amd as P
        broker = P.Broker(ix, max_batch=4, max_wait_us=0)
    scan = ix.beginscan() if broker is None else broker.beginscan()
    try:
        for rnd in range(2):
            i = int(rng.integers(0, q.shape[0]))
            null = rng.random() < 0.2
            lab = None if keys is None else keys[i]
            rows = int(rng.choice([1, 7, 60, 10 ** 9]))
            scan.rescan(None if null else q[i], labels=lab, search_list_size=L, rescore=rescore)
            osc = oracle.scan(None if null else q[i], labels=lab, L=L, rescore=rescore)
            got = 0
            while got < rows:
                r, o = scan.gettuple(), osc.gettuple()
                assert (r is None) == (o is None), f"{where}: gettuple row {got} of scan {rnd}: end of scan differs"
                if r is None:
                    break
                assert r[1] == o[0] and r[0] == o[1], f"{where}: gettuple row {got} of scan {rnd}: {r} vs {o}"
                if exact_dist:
                    assert np.float32(r[2]).view(np.uint32) == np.float32(o[2]).view(np.uint32), f"{where}: gettuple distance bits"
                else:
                    assert close(r[2], o[2]), f"{where}: gettuple distance {r[2]} vs {o[2]}"
                got += 1
                if got in (1, 2, 9, 33) or got % 61 == 0:  # the cursor's GreedySearchStats after this many amgettuple calls
                    gs, os_ = scan.stats(), osc.stats()
                    for c in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons",
                              "node_reads", "node_heap_reads", "next_calls"):
                        assert gs[c] == os_[c], f"{where}: cursor counter {c} after {got} rows of scan {rnd}: {gs[c]} != {os_[c]}"
            gs, os_ = scan.stats(), osc.stats()
            for c in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "next_calls"):
                assert gs[c] == os_[c], f"{where}: cursor counter {c} at the end of scan {rnd}: {gs[c]} != {os_[c]}"
    finally:
        scan.endscan()
        if broker is not None:
            broker.close()


def plain_case(ctx, O, case_seed, verbose):
    """`plain` storage: f32 distances in the graph, ids and distance bits must equal the oracle's"""
    from test_gpu_zy_plain import PlainIndex
    from helpers import make_vectors
    rng = np.random.default_rng(case_seed)
    dim = int(rng.choice([3, 8, 17, 36, 64, 100, 128, 384]))
    n = int(rng.choice([1, 2, 40, 300, 1200]))
    R = int(rng.choice([4, 16, 24, 50]))
    distance = int(rng.choice([0, 1, 2]))
    kind = str(rng.choice(["uniform", "gauss", "clustered"]))
    dim_index = None if rng.random() < 0.7 or dim < 8 else int(rng.integers(2, dim))
    pi = PlainIndex(n=max(n, 4) if kind == "gauss" else n, dim=dim, R=R, distance=distance, seed=int(rng.integers(1, 1 << 30)), kind=kind,
                    deleted_frac=float(rng.choice([0.0, 0.1, 0.5])), dim_index=dim_index)
    nq = int(rng.choice([1, 5, 33]))
    q = make_vectors(nq, dim, int(rng.integers(1, 1 << 30)), kind)
    L = int(rng.choice([1, 5, 30, 100]))
    k = int(rng.choice([1, 10, 40]))
    where = f"plain case {case_seed}: n={pi.n} dim={dim}/{pi.dim_index} R={R} dist={distance} {kind} nq={nq} L={L} k={k}"
    if verbose:
        print(where, flush=True)
    for v in TUNING:
        os.environ.pop(v, None)
    ix = pi.upload(ctx)
    try:
        gi, gt, gd, gst = ix.search_batch(q, search_list_size=L, rescore=50, k=k)
        oi, od, ost = pi.oracle.search_batch(q, L=L, rescore=50, k=k)
        assert (gi == oi).all(), f"{where}: ids differ"
        assert (gd.view(np.uint32) == od.view(np.uint32)).all(), f"{where}: distance bits differ"
        for c in ("visited_nodes", "candidate_nodes", "full_distance_comparisons"):
            assert gst[c] == ost[c], f"{where}: counter {c}"
        gettuple_mirror(ix, pi.oracle, q, None, L, 50, np.random.default_rng(case_seed + 1), where, exact_dist=True)
    finally:
        ix.close()


def pages_case(ctx, O, case_seed, verbose):
    """index relation pages (written byte by byte in the reference's layout, oracle/pages_py.py) -> host decode and device
    decode -> the arrays the pages were written from, and scans equal to the oracle's"""
    from helpers import TestIndex
    from oracle import pages_py as PG
    from pgvectorscale_amd import _lib
    from pgvectorscale_amd.pages import DevicePages, IndexPages
    rng = np.random.default_rng(case_seed)
    dim = int(rng.choice([8, 33, 64, 128, 200, 768]))
    n = int(rng.choice([1, 3, 60, 400, 1300]))
    R = int(rng.choice([4, 15, 24, 50]))
    n_labels = int(rng.choice([0, 0, 5, 32]))
    ti = TestIndex(n=n, dim_full=dim, R=R, distance=int(rng.choice([0, 1])), seed=int(rng.integers(1, 1 << 30)), kind="gauss",
                   n_labels=n_labels, deleted_frac=float(rng.choice([0.0, 0.2])), L_build=30, label_zipf=bool(rng.random() < 0.5))
    zp = int(rng.choice([0, 7, 100]))
    mf = bool(rng.random() < 0.5)
    where = f"pages case {case_seed}: n={n} dim={dim} bits={ti.bits} R={R} labels={n_labels} zero_page_every={zp} means_first={mf}"
    if verbose:
        print(where, flush=True)
    for v in TUNING:
        os.environ.pop(v, None)
    w = PG.write_index(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, mean=ti.mean, m2=ti.m2, count=ti.count, label_off=ti.label_off,
                       label_val=ti.label_val, zero_page_every=zp, means_first=mf)
    data = w.rel.tobytes()
    nblk = len(w.rel.pages)
    cut = int(rng.integers(0, nblk + 1)) * PG.BLCKSZ
    starts = {l: w.node_ptrs[v] for l, v in ti.label_starts.items()}
    common = dict(dim_index=ti.dim_index, bits=ti.bits, distance_type=ti.distance, default_start=w.node_ptrs[ti.start],
                  quantizer_metadata=w.means_ptr, vecs=ti.vecs, label_starts=starts)
    hp = IndexPages(has_labels=bool(n_labels))
    hp.add(data[:cut])
    hp.add(data[cut:])
    hp.finish()
    ix_h = hp.upload(ctx, **common)
    hp.close()
    dp = DevicePages(ctx, nblk)
    dp.add(data[:cut])
    dp.add(data[cut:])
    ix_d = dp.build(words=ti.codes.shape[1], num_neighbors=R, has_labels=bool(n_labels), **common)
    dp.close()
    try:
        q = ti.queries(9, seed=int(rng.integers(1, 1 << 30)), kind="gauss")
        keys = None
        if n_labels:
            keys = [[int(x) for x in rng.integers(1, n_labels + 1, int(rng.integers(1, 3)))] for _ in range(9)]
        oi, od, ost = ti.oracle.search_batch(q, L=40, rescore=20, k=10, qlabels=keys)
        for name, ix in (("host decode", ix_h), ("device decode", ix_d)):
            dev = ix.download()
            assert (dev["codes"] == ti.codes).all() and (dev["nbrs"] == ti.nbrs).all() and (dev["heap_tids"] == ti.tids).all(), \
                f"{where}: {name}: arrays differ"
            if n_labels:
                lo = ctx.download(ix.array(_lib.ARR_LABEL_OFF)[0], np.empty(n + 1, np.uint32))
                lv = ctx.download(ix.array(_lib.ARR_LABEL_VAL)[0], np.empty(max(len(ti.label_val), 1), np.int16))[:len(ti.label_val)]
                assert (lo == ti.label_off).all() and (lv == ti.label_val).all(), f"{where}: {name}: label sets differ"
            gi, gt, gd, gst = ix.search_batch(q, search_list_size=40, rescore=20, k=10, qlabels=keys)
            assert (gi == oi).all() and close(gd, od), f"{where}: {name}: rows differ"
    finally:
        ix_h.close()
        ix_d.close()


def kernels_case(ctx, O, case_seed, verbose):
    """the standalone kernels of the C ABI: vs_quantize, vs_hamming_gather, vs_rerank, vs_scan_topk, vs_bruteforce_topk"""
    from helpers import TestIndex
    rng = np.random.default_rng(case_seed)
    dim = int(rng.choice([3, 8, 33, 64, 100, 128, 384, 768, 1536]))
    bits = int(rng.choice([0, 1, 2, 3]))
    if bits and dim * bits > 1600:
        bits = 1
    n = int(rng.choice([1, 7, 100, 1000, 5000]))
    distance = int(rng.choice([0, 1, 2]))
    kind = str(rng.choice(["uniform", "gauss", "clustered"]))
    ti = TestIndex(n=n, dim_full=dim, bits=bits or None, R=8, distance=distance, seed=int(rng.integers(1, 1 << 30)), kind=kind, L_build=10)
    if kind == "gauss" and n > 8:  # rows of very different norms, a zero row
        ti.vecs[::3] *= 2.5
        ti.vecs[5] = 0
    where = f"kernels case {case_seed}: n={n} dim={dim} bits={ti.bits} dist={distance} {kind}"
    if verbose:
        print(where, flush=True)
    for v in TUNING:
        os.environ.pop(v, None)
    ix = ti.upload(ctx)
    try:
        nq = int(rng.choice([1, 5, 40]))
        Q = ti.queries(nq, seed=int(rng.integers(1, 1 << 30)), kind=kind)
        Q[0, 0] = np.float32("inf") if rng.random() < 0.1 else Q[0, 0]
        if rng.random() < 0.1:
            Q[-1] = np.float32("nan")
        # SbqQuantizer::quantize
        qc = ix.quantize(Q)
        assert (qc == O.quantize(ti.mean, ti.m2, ti.count, ti.bits, Q)).all(), f"{where}: vs_quantize"
        # distance_xor_optimized over gathered nodes
        lists = [rng.integers(0, n, int(rng.choice([0, 1, 9, 130]))).astype(np.uint32) for _ in range(nq)]
        got = ix.hamming_gather(qc, lists)
        for i in range(nq):
            want = np.array([O.distance_xor(qc[i], ti.codes[v]) for v in lists[i]], np.uint32)
            assert (got[i] == want).all(), f"{where}: vs_hamming_gather query {i}"
        # exact (hamming, id) top-k of the flat scan
        k = int(rng.choice([1, 10, 64]))
        ids, ham = ix.scan_topk(qc, k)
        wi, wh = O.hamming_scan_topk(ti.codes, qc, k)
        assert (ids == wi).all() and (ham == wh).all(), f"{where}: vs_scan_topk k={k}"
        # full-precision distances in the reference's accumulation order (finite queries only)
        fin = [i for i in range(nq) if np.isfinite(Q[i]).all()]
        if fin:
            Qf = Q[fin]
            rl = [lists[i] for i in fin]
            gotd = ix.rerank(Qf, rl)
            for a, qi in enumerate(fin):
                q = O.preprocess_cosine(Q[qi])[0] if distance == 0 else Q[qi]
                for j, node in enumerate(rl[a]):
                    v = O.preprocess_cosine(ti.vecs[node])[0] if distance == 0 else ti.vecs[node]
                    want = O.distance_by_type(distance, v, q)
                    assert np.float32(gotd[a][j]).tobytes() == np.float32(want).tobytes(), \
                        f"{where}: vs_rerank query {qi} node {node}: {gotd[a][j]} vs {want}"
            dq = ctx.alloc(Qf.nbytes)
            ctx.upload(dq, Qf)
            kb = min(k, 32)
            bi, bd = ix.bruteforce_topk(dq, len(Qf), kb)
            ctx.free(dq)
            # the oracle's brute force scores the uploaded vectors (ti.oracle holds the unmodified copy: rebuild its view)
            oidx = O.OracleIndex(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, vecs=ti.vecs, mean=ti.mean, m2=ti.m2, count=ti.count,
                                 bits=ti.bits, dim_index=ti.dim_index, num_neighbors=ti.R, distance_type=distance, default_start=ti.start)
            oi, od = oidx.bruteforce(Qf, k=kb)
            assert (bi == oi).all(), f"{where}: vs_bruteforce_topk ids"
            assert (bd.view(np.uint32) == np.asarray(od, np.float32).view(np.uint32)).all(), f"{where}: vs_bruteforce_topk distances"
    finally:
        ix.close()


def reachable(nbrs, start):
    n = nbrs.shape[0]
    seen = np.zeros(n, bool)
    seen[start] = True
    stack = [int(start)]
    while stack:
        v = stack.pop()
        for u in nbrs[v]:
            if u != 0xFFFFFFFF and not seen[u]:
                seen[u] = True
                stack.append(int(u))
    return int(seen.sum())


def build_case(ctx, O, case_seed, verbose):
    """index manufacture on the device: SBQ training and corpus quantisation bit for bit against the oracle; the graph the
    device builds is well formed, identical when built twice, and searched identically by the device and the oracle"""
    import pgvectorscale_amd as P
    from helpers import make_vectors
    rng = np.random.default_rng(case_seed)
    dim = int(rng.choice([8, 40, 64, 96, 128, 384]))
    bits = int(rng.choice([0, 1, 2, 3]))
    if bits and dim * bits > 1200:
        bits = 1
    n = int(rng.choice([1, 2, 50, 700, 2500]))
    R = int(rng.choice([4, 8, 16, 32, 50]))
    distance = int(rng.choice([0, 1, 2]))
    dim_index = None if rng.random() < 0.7 else int(rng.integers(2, dim))
    kind = str(rng.choice(["uniform", "gauss", "clustered"]))
    Lb = int(rng.choice([1, 10, 40, 100]))
    where = f"build case {case_seed}: n={n} dim={dim}/{dim_index} bits={bits} R={R} dist={distance} {kind} L_build={Lb}"
    if verbose:
        print(where, flush=True)
    for v in TUNING:
        os.environ.pop(v, None)
    X = make_vectors(n, dim, int(rng.integers(1, 1 << 30)), kind)
    scaled = kind == "gauss" and n > 8  # rows of very different norms: the cosine rescale path of the quantiser runs; such a
    if scaled:                           # corpus prunes to a sparse graph (the oracle's builder too), so no navigability claim
        X[::3] *= 2.5
        X[7] = 0
    graphs = []
    for rep in range(2):
        ix = P.DiskAnnIndex.alloc(ctx, n=n, dim_full=dim, dim_index=dim_index, bits=bits or None, num_neighbors=R, distance_type=distance)
        try:
            ctx.upload(ix.array(P._lib.ARR_VECS)[0], X) if ix.array(P._lib.ARR_VECS)[1] == dim else None
            if ix.array(P._lib.ARR_VECS)[1] != dim:  # padded rows on the device
                vp, stride = ix.array(P._lib.ARR_VECS)
                Xp = np.zeros((n, stride), np.float32)
                Xp[:, :dim] = X
                ctx.upload(vp, Xp)
            ix.refresh_norms()
            ix.sbq_train()
            ix.sbq_quantize_corpus()
            ix.build_graph(search_list_size=Lb, max_alpha=1.2)
            host = ix.download(vecs=True)
            assert (host["vecs"] == X).all(), f"{where}: vectors did not round-trip"
            di = dim_index or dim
            b = ix.desc.bits
            sl = np.ascontiguousarray(X[:, :di]).copy()
            if distance == 0:
                for i in range(n):
                    sl[i] = O.preprocess_cosine(sl[i])[0]
            mean, m2, cnt = O.train(sl, b)
            gmean, gm2, gcnt = ix.get_quantizer()
            assert gcnt == cnt == n and gmean.tobytes() == mean.tobytes(), f"{where}: SBQ means differ"
            if b > 1:
                assert gm2.tobytes() == m2.tobytes(), f"{where}: SBQ m2 differs"
            assert (host["codes"] == O.quantize(mean, m2, cnt, b, sl)).all(), f"{where}: codes differ"
            nb = host["nbrs"]
            graphs.append(nb.copy())
            for i in range(n):
                row = nb[i]
                live = row[row != 0xFFFFFFFF]
                assert (row[:len(live)] != 0xFFFFFFFF).all() and len(set(live.tolist())) == len(live) and i not in live \
                    and (live < n).all(), f"{where}: neighbor list of node {i} malformed: {row}"
            if rep == 0:
                oidx = O.OracleIndex(codes=host["codes"], nbrs=nb, heap_tids=host["heap_tids"], vecs=X, mean=mean, m2=m2, count=cnt,
                                     bits=b, dim_index=di, num_neighbors=R, distance_type=distance, default_start=ix.desc.default_start)
                q = make_vectors(16, dim, int(rng.integers(1, 1 << 30)), kind)
                gi, gt, gd, gst = ix.search_batch(q, search_list_size=50, rescore=25, k=10)
                oi, od, ost = oidx.search_batch(q, L=50, rescore=25, k=10)
                assert (gi == oi).all() and close(gd, od), f"{where}: rows on the device-built graph differ"
                # navigable: an exhaustive scan reaches every node (AM/build.rs:1254-1269) wherever the reference's own
                # sequential algorithm (the oracle's builder) achieves that on these codes, and never fewer nodes than it does
                sc = oidx.scan(q[0], L=2, rescore=0)
                seen = 0
                while sc.next_sbq() is not None:
                    seen += 1
                onb, ostart = O.build_graph(host["codes"], num_neighbors=R, search_list_size=Lb)
                want = reachable(onb, ostart)
                assert seen == reachable(nb, ix.desc.default_start), f"{where}: the scan and a BFS disagree"
                # (within the reference's option ranges: num_neighbors >= 10, search_list_size in 10..1000, AM/options.rs:55-64,
                # 213-232; below them neither builder makes a navigable graph and the case only checks form and parity)
                # (a corpus on which the sequential builder itself reaches under 90 % of the nodes — SBQ codes of rows with very
                # different norms around a zero vector under L2 — is not navigable either way)
                assert seen >= want or R < 10 or Lb < 10 or want < 0.9 * n, \
                    f"{where}: exhaustive scan reached {seen} of {n} nodes, the sequential builder's graph {want}"
        finally:
            ix.close()
    assert (graphs[0] == graphs[1]).all(), f"{where}: the build is not deterministic"


def one_case(ctx, O, case_seed, verbose):
    from helpers import TestIndex
    if case_seed % 11 == 10:
        return build_case(ctx, O, case_seed, verbose)
    if case_seed % 13 == 12:
        return kernels_case(ctx, O, case_seed, verbose)
    if case_seed % 5 == 4:
        return plain_case(ctx, O, case_seed, verbose)
    if case_seed % 7 == 6:
        return pages_case(ctx, O, case_seed, verbose)
    rng = np.random.default_rng(case_seed)
    dim_full = int(rng.choice([3, 8, 17, 32, 48, 64, 65, 100, 128, 200, 384, 768]))
    bits = int(rng.choice([0, 1, 2, 3]))  # 0 = the reference's default for the dimension count
    if bits and dim_full * bits > 1600:
        bits = 1
    dim_index = dim_full if rng.random() < 0.8 or dim_full < 8 else int(rng.integers(2, dim_full))
    n = int(rng.choice([1, 2, 5, 40, 300, 900, 1500, 2500]))
    R = int(rng.choice([4, 8, 16, 20, 32, 50, 64, 80]))
    distance = int(rng.choice([0, 1, 2]))
    n_labels = int(rng.choice([0, 0, 3, 8, 32]))
    deleted = float(rng.choice([0.0, 0.0, 0.1, 0.6]))
    kind = str(rng.choice(["uniform", "gauss", "clustered"]))
    ti = TestIndex(n=n, dim_full=dim_full, dim_index=dim_index, bits=bits or None, R=R, distance=distance, seed=int(rng.integers(1, 1 << 30)),
                   kind=kind, n_labels=n_labels, deleted_frac=deleted, L_build=int(rng.choice([10, 50, 100])),
                   label_zipf=bool(rng.random() < 0.5))
    desc = f"n={n} dim={dim_full}/{dim_index} bits={ti.bits} R={R} dist={distance} labels={n_labels} del={deleted} {kind}"
    nq = int(rng.choice([1, 3, 17, 64]))
    q = ti.queries(nq, seed=int(rng.integers(1, 1 << 30)), kind=kind)
    if rng.random() < 0.2:  # duplicates of corpus rows: distance ties
        for i in range(0, nq, 2):
            q[i] = ti.vecs[int(rng.integers(0, n))]
    keys = None
    if n_labels and rng.random() < 0.8:
        keys = []
        for _ in range(nq):
            kk = int(rng.choice([0, 1, 1, 2, 2, 5]))
            keys.append([int(x) for x in rng.integers(1, n_labels + 2, kk)])  # unsorted, duplicates, one label nobody has
    L = int(rng.choice([1, 2, 10, 30, 64, 100, 150, 300, 1000]))
    rescore = int(rng.choice([0, 1, 10, 50, 115, 400]))
    k = int(rng.choice([1, 5, 10, 40, 200]))
    m = int(rng.choice([1, 20, 75]))
    regimes = [REGIMES[i] for i in rng.choice(len(REGIMES), 4, replace=False)]
    rng_extra = np.random.default_rng(case_seed + 7)
    if rng_extra.random() < 0.5:
        regimes.append(EXTRA_REGIMES[int(rng_extra.integers(0, len(EXTRA_REGIMES)))])
    oi, od, ost = ti.oracle.search_batch(q, L=L, rescore=rescore, k=k, qlabels=keys)
    si, sh, sst = ti.oracle.stream_batch(q, L=L, m=m, qlabels=keys)
    for reg in regimes:
        for v in TUNING:
            os.environ.pop(v, None)
        os.environ.update(reg)
        where = f"case {case_seed}: {desc} nq={nq} L={L} rescore={rescore} k={k} m={m} keys={'yes' if keys else 'no'} regime={reg}"
        if verbose:
            print(where, flush=True)
        ix = ti.upload(ctx)
        try:
            for rep in range(2):  # the second call runs with launch sizes adapted to the first one's statistics
                gi, gt, gd, gst = ix.search_batch(q, search_list_size=L, rescore=rescore, k=k, qlabels=keys)
                assert (gi == oi).all(), f"{where}: ids differ (rep {rep})"
                assert close(gd, od), f"{where}: distances differ (rep {rep})"
                live = gi != 0xFFFFFFFF
                assert (gt[live] == ti.tids[gi[live]]).all(), f"{where}: heap tids differ"
                for c in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons"):
                    assert gst[c] == ost[c], f"{where}: counter {c} {gst[c]} != {ost[c]}"
            gi2, gh2, gst2 = ix.stream_batch(q, search_list_size=L, m=m, qlabels=keys)
            if not ((gi2 == si).all() and (gh2 == sh).all()):
                bad = np.argwhere((gi2 != si) | (gh2 != sh))
                q0, p0 = (int(x) for x in bad[0])
                raise AssertionError(f"{where}: stream differs: {len(bad)} of {gi2.size} entries, first at query {q0} row {p0}: got "
                                     f"({gi2[q0, p0]}, {gh2[q0, p0]}) want ({si[q0, p0]}, {sh[q0, p0]}); got row {gi2[q0].tolist()} "
                                     f"ham {gh2[q0].tolist()} want row {si[q0].tolist()} ham {sh[q0].tolist()}; counters {gst2} vs {sst}")
            assert gst2["candidate_nodes"] == sst["candidate_nodes"], f"{where}: stream counters differ"
            gettuple_mirror(ix, ti.oracle, q, keys, L, rescore, np.random.default_rng(case_seed + 1), where, exact_dist=False)
            if reg is regimes[0] and n > 1:  # tuples deleted after the upload (vs_index_mark_deleted): skipped by the scans from now on
                dead = np.unique(rng.integers(0, n, max(1, n // 7))).astype(np.uint32)
                ix.mark_deleted(dead)
                tids2 = ti.tids.copy()
                tids2[dead] &= ~np.uint64(0xFFFF)
                o2 = O.OracleIndex(codes=ti.codes, nbrs=ti.nbrs, heap_tids=tids2, vecs=ti.vecs, mean=ti.mean, m2=ti.m2, count=ti.count,
                                   bits=ti.bits, dim_index=ti.dim_index, num_neighbors=ti.R, distance_type=ti.distance,
                                   default_start=ti.start, label_off=ti.label_off, label_val=ti.label_val, label_starts=ti.label_starts)
                gi3, gt3, gd3, _ = ix.search_batch(q, search_list_size=L, rescore=rescore, k=k, qlabels=keys)
                oi3, od3, _ = o2.search_batch(q, L=L, rescore=rescore, k=k, qlabels=keys)
                assert (gi3 == oi3).all() and close(gd3, od3), f"{where}: rows after vs_index_mark_deleted differ"
                assert not (set(gi3.ravel().tolist()) & set(dead.tolist())), f"{where}: a deleted tuple was returned"
        finally:
            ix.close()
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None, help="run this one case seed")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--kind", default="any", choices=["any", "search", "plain", "pages", "build", "kernels"], help="only cases of this kind")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--lib", default=None, help="with --gpu: this libvsgpu build instead of pgvectorscale_amd/libvsgpu.so (bisecting)")
    ap.add_argument("--repeat", type=int, default=1, help="with --only: run the case this many times (rare hardware-only failures)")
    args = ap.parse_args()
    from pgvectorscale_amd import _lib
    if args.gpu and args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    if not args.gpu:
        _lib.LIB_PATH = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")
        assert os.path.exists(_lib.LIB_PATH), "make -C tests/emu first"
    import pgvectorscale_amd as P
    from oracle import oracle_py as O
    O.build()
    ctx = P.Context(0)
    t0 = time.time()
    cases = failures = 0
    seeds = [args.only] * args.repeat if args.only is not None else (args.seed * 1_000_000 + i for i in range(1 << 30))
    def kind_of(cs):
        return ("build" if cs % 11 == 10 else "kernels" if cs % 13 == 12 else "plain" if cs % 5 == 4 else "pages" if cs % 7 == 6
                else "search")

    for cs in seeds:
        if args.only is None and time.time() - t0 > args.seconds:
            break
        if args.only is None and args.kind != "any" and kind_of(cs) != args.kind:
            continue
        try:
            one_case(ctx, O, cs, args.verbose or (args.only is not None and args.repeat == 1))
        except AssertionError as e:
            failures += 1
            print("FAIL", e, flush=True)
        except Exception:
            failures += 1
            print(f"ERROR in case {cs}:", flush=True)
            traceback.print_exc()
        cases += 1
    print(f"{cases} cases, {failures} failures, {time.time() - t0:.0f} s")
    ctx.close()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
