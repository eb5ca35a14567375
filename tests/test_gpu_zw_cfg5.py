"""BASELINE.json configs[4] in miniature: 1536-d unit-norm mixture ("OpenAI-like"), SBQ 1 bit (24-word codes), 32 labels with
Zipf frequencies and 1-3 labels per vector, label-filtered scans with one and with two labels in the key
(Filtered-DiskANN predicate = LabelSet overlap, AM/labels/mod.rs:124-142; start nodes per label, AM/graph/start_nodes.rs:39-48).
Rows and work counters must equal the oracle's in the LDS-table and in the table-less regime of k_search_fast."""
import os

import numpy as np
import pytest

from helpers import cached_index

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

KW = dict(n=3000, dim_full=1536, R=50, distance=0, seed=8, kind="clustered", L_build=100, n_labels=32, label_zipf=True,
          deleted_frac=0.02)


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return np.all(nan | (np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-30) + 1e-12))


@pytest.mark.parametrize("regime", ["lds_table", "tableless"])
def test_label_filtered_1536d_one_bit(gpu_ctx, oracle, regime, monkeypatch):
    ti = cached_index(**KW)
    assert ti.bits == 1 and ti.codes.shape[1] == 24
    if regime == "tableless":
        monkeypatch.setenv("VS_F_LDS_MAX_INS", "0")
    ix = ti.upload(gpu_ctx)
    nq = 48
    q = ti.queries(nq, seed=9, kind="clustered")
    rng = np.random.default_rng(10)
    pz = 1.0 / np.arange(1, 33)
    pz /= pz.sum()
    keys = [sorted(set(int(x) + 1 for x in rng.choice(32, 1 if i % 2 == 0 else 2, p=pz))) for i in range(nq)]
    keys[5] = [32]  # the rarest label
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=100, rescore=50, k=10, qlabels=keys)
    oi, od, ost = ti.oracle.search_batch(q, L=100, rescore=50, k=10, qlabels=keys)
    assert (gi == oi).all() and _close(gd, od)
    for c in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons"):
        assert gst[c] == ost[c], c
    # every returned row satisfies the predicate and is live
    for i in range(nq):
        for v in gi[i][gi[i] != 0xFFFFFFFF]:
            assert set(ti.label_val[ti.label_off[v]:ti.label_off[v + 1]].tolist()) & set(keys[i])
            assert ti.tids[v] & np.uint64(0xFFFF)
    # the SBQ-ordered stream (ids + Hamming distances), before the rerank
    si, sh, sst = ix.stream_batch(q, search_list_size=100, m=65, qlabels=keys)
    ti_i, ti_h, _ = ti.oracle.stream_batch(q, L=100, m=65, qlabels=keys)
    assert (si == ti_i).all() and (sh == ti_h).all()
    ix.close()
