"""Committed byte-layout fixtures (tests/golden/pages_golden.npz, written by tests/golden/make_golden_pages.py): an archived
MetaPage, heap + TOAST pages with vectors in every storage form, and the GreedySearchStats trajectory of one streamed scan.  The CPU
tests hold libvsgpu's host decoders AND the pure-Python readers to the frozen bytes; the GPU test holds the amgettuple cursor to the
frozen rows and counters — no encoder and no oracle involved on the decoding side."""
import os

import numpy as np
import pytest

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pages_golden.npz")
ATTRS = [(8, "d"), (2, "s"), (-1, "i"), (-1, "i"), (4, "i")]


@pytest.fixture(scope="module")
def g():
    return np.load(PATH)


def test_meta_page_bytes_decode_to_the_frozen_fields(g):
    from oracle import pages_py as PG
    from pgvectorscale_amd.pages import decode_meta_page
    data = g["meta_bytes"].tobytes()
    want_starts = {int(l): (int(b), int(o)) for l, b, o in zip(g["meta_start_labels"], g["meta_start_blocks"], g["meta_start_offsets"])}
    for fields, starts in (decode_meta_page(data), (lambda d: (d, d["labeled_starts"]))(PG.parse_meta_page(data))):
        assert fields["extension_version_when_built"] == "0.8.0-golden+fixture"
        assert (fields["distance_type"], fields["num_dimensions"], fields["num_dimensions_to_index"], fields["bq_num_bits_per_dimension"],
                fields["storage_type"], fields["num_neighbors"], fields["search_list_size"], fields["max_alpha"]) == (0, 1536, 768, 2, 2, 50, 100, 1.2)
        assert starts == want_starts and len(starts) == 300
    f, _ = decode_meta_page(data)
    assert (f["default_start_block"], f["default_start_offset"], f["quantizer_block"], f["quantizer_offset"], f["has_labels"]) == (17, 3, 1, 1, 1)


@pytest.mark.parametrize("dim", [5, 100, 768])
def test_heap_pages_decode_to_the_frozen_vectors(g, dim):
    from oracle import heap_py as HP
    from pgvectorscale_amd.pages import HeapColumn
    tids, want = g[f"heap{dim}_tids"], g[f"heap{dim}_vecs"]
    hc = HeapColumn(ATTRS, 4, dim, tids)
    hc.add(g[f"heap{dim}_bytes"].tobytes())
    if g[f"toast{dim}_bytes"].size:
        hc.toast_add(g[f"toast{dim}_bytes"].tobytes())
    info, found = hc.finish()
    assert found.all() and hc.vecs.tobytes() == want.tobytes()
    assert (info["n_external"] > 0) == (dim == 768)
    hc.close()
    # the independent reader on the same bytes
    t = HP.Table(ATTRS)
    t.heap.pages = [bytearray(g[f"heap{dim}_bytes"][i:i + HP.BLCKSZ]) for i in range(0, g[f"heap{dim}_bytes"].size, HP.BLCKSZ)]
    t.toast.pages = [bytearray(g[f"toast{dim}_bytes"][i:i + HP.BLCKSZ]) for i in range(0, g[f"toast{dim}_bytes"].size, HP.BLCKSZ)]
    ref = HP.read_vector_column(t, tids, 3, dim)
    assert all(r.tobytes() == w.tobytes() for r, w in zip(ref, want))


KEYS = ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
        "node_heap_reads", "next_calls")


def _cursor_index(g, O):
    return dict(codes=g["cur_codes"], nbrs=g["cur_nbrs"], heap_tids=g["cur_tids"], vecs=g["cur_vecs"], mean=g["cur_mean"], m2=g["cur_m2"],
                count=int(g["cur_count"]), bits=2, dim_index=32, num_neighbors=12, distance_type=1, default_start=int(g["cur_start"]))


def test_oracle_reproduces_the_frozen_cursor_trajectory(g, oracle):
    oidx = oracle.OracleIndex(**_cursor_index(g, oracle))
    sc = oidx.scan(g["cur_query"], L=8, rescore=6)
    for i in range(40):
        assert sc.gettuple()[0] == int(g["cur_rows"][i])
        st = sc.stats()
        assert [st[k] for k in KEYS] == g["cur_stats"][i].tolist(), i


@pytest.mark.gpu
def test_cursor_reproduces_the_frozen_trajectory(g, gpu_ctx):
    import pgvectorscale_amd as P
    ix = P.DiskAnnIndex.upload(gpu_ctx, **_cursor_index(g, None))
    scan = ix.beginscan()
    scan.rescan(g["cur_query"], search_list_size=8, rescore=6)
    for i in range(40):
        assert scan.gettuple()[1] == int(g["cur_rows"][i])
        st = scan.stats()
        assert [st[k] for k in KEYS] == g["cur_stats"][i].tolist(), i
    scan.endscan()
    ix.close()
