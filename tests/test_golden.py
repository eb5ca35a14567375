"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py): small indexes with their
queries and the expected SBQ-ordered streams, rows and rerank distances.  The CPU test checks that the oracle still
reproduces them (the oracle is the definition the HIP path is held to); the GPU test uploads the SAVED arrays through
the C ABI and checks the HIP path against the same files — no oracle involved on that side."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
                if not os.path.basename(p).startswith("pages_"))  # (pages_golden.npz: byte-layout fixtures, tests/test_golden_pages.py)


def _qlabels(g):
    if "qlabel_off" not in g:
        return None
    off, val = g["qlabel_off"], g["qlabel_val"]
    return [list(int(v) for v in val[off[i]:off[i + 1]]) for i in range(len(off) - 1)]


def _label_starts(g):
    if "label_start_labels" not in g:
        return {}
    return {int(a): int(b) for a, b in zip(g["label_start_labels"], g["label_start_nodes"])}


def test_fixtures_exist():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    O = oracle
    g = np.load(path)
    m2 = g["m2"] if g["m2"].size else None
    oidx = O.OracleIndex(codes=g["codes"], nbrs=g["nbrs"], heap_tids=g["tids"], vecs=g["vecs"], mean=g["mean"], m2=m2,
                         count=int(g["count"]), bits=int(g["bits"]), dim_index=int(g["dim_index"]),
                         num_neighbors=int(g["R"]), distance_type=int(g["distance"]), default_start=int(g["start"]),
                         label_off=g["label_off"] if "label_off" in g else None,
                         label_val=g["label_val"] if "label_val" in g else None, label_starts=_label_starts(g))
    ql = _qlabels(g)
    si, sh, st = oidx.stream_batch(g["queries"], L=int(g["L"]), m=int(g["m"]), qlabels=ql)
    assert (si == g["stream_ids"]).all() and (sh == g["stream_ham"]).all()
    assert st["visited_nodes"] == int(g["visited_nodes"])
    ri, rd, _ = oidx.search_batch(g["queries"], L=int(g["L"]), rescore=int(g["rescore"]), k=int(g["k"]), qlabels=ql)
    assert (ri == g["rows_ids"]).all()
    assert (rd.view(np.uint32) == g["rows_dist"].view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_reproduces_golden(gpu_ctx, path):
    import pgvectorscale_amd as P
    g = np.load(path)
    m2 = g["m2"] if g["m2"].size else None
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=g["codes"], nbrs=g["nbrs"], heap_tids=g["tids"], vecs=g["vecs"],
                               mean=g["mean"], m2=m2, count=int(g["count"]), bits=int(g["bits"]),
                               dim_index=int(g["dim_index"]), num_neighbors=int(g["R"]), distance_type=int(g["distance"]),
                               default_start=int(g["start"]), label_off=g["label_off"] if "label_off" in g else None,
                               label_val=g["label_val"] if "label_val" in g else None, label_starts=_label_starts(g))
    ql = _qlabels(g)
    si, sh, st = ix.stream_batch(g["queries"], search_list_size=int(g["L"]), m=int(g["m"]), qlabels=ql)
    assert (si == g["stream_ids"]).all() and (sh == g["stream_ham"]).all()
    assert st["visited_nodes"] == int(g["visited_nodes"])
    assert st["quantized_distance_comparisons"] == int(g["quantized_distance_comparisons"])
    ri, _, rd, _ = ix.search_batch(g["queries"], search_list_size=int(g["L"]), rescore=int(g["rescore"]), k=int(g["k"]),
                                   qlabels=ql)
    assert (ri == g["rows_ids"]).all()
    want = g["rows_dist"]
    nan = np.isnan(want)
    assert (np.isnan(rd) == nan).all()
    assert np.allclose(rd[~nan], want[~nan], rtol=1e-5, atol=0)          # the bar north_star states
    assert (rd.view(np.uint32)[~nan] == want.view(np.uint32)[~nan]).all()  # and in fact bit-identical
    ix.close()
