#!/usr/bin/env python3
"""Generates tests/golden/pages_golden.npz: frozen BYTES of the page-layer formats the readers decode — an archived MetaPage
(out-of-line version string, 300 labeled start nodes = a B-tree with an inner root), a three-block heap + its TOAST relation with
vectors in every storage form — and the values they must decode to; plus the GreedySearchStats trajectory of one streamed scan
(the counters after 1, 2, 3, ... amgettuple calls).  Written by the test-infrastructure encoders (oracle/pages_py.py,
oracle/heap_py.py) and the oracle: like the other fixtures they FREEZE behaviour, they do not pin it against the reference.

  python tests/golden/make_golden_pages.py     # rewrites the file (only for a deliberate change of a byte layout)
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import TestIndex  # noqa: E402
from oracle import heap_py as HP  # noqa: E402
from oracle import pages_py as PG  # noqa: E402

ATTRS = [(8, "d"), (2, "s"), (-1, "i"), (-1, "i"), (4, "i")]  # id bigint, label smallint, note text, embedding vector, extra int4


def main():
    out = {}
    starts = {l: (1000 + (l * 7) % 911, 1 + l % 13) for l in range(-150, 150)}
    out["meta_bytes"] = np.frombuffer(PG.rkyv_meta_page(
        extension_version="0.8.0-golden+fixture", distance_type=0, num_dimensions=1536, num_dimensions_to_index=768,
        bq_num_bits_per_dimension=2, storage_type=2, num_neighbors=50, search_list_size=100, max_alpha=1.2, default_start=(17, 3),
        labeled_starts=starts, quantizer=(1, 1), has_labels=True), np.uint8)
    out["meta_start_labels"] = np.array(sorted(starts), np.int16)
    out["meta_start_blocks"] = np.array([starts[l][0] for l in sorted(starts)], np.uint32)
    out["meta_start_offsets"] = np.array([starts[l][1] for l in sorted(starts)], np.uint32)
    rng = np.random.default_rng(11)
    for dim in (5, 100, 768):  # 1-byte varlena header / 4-byte header in line / out of line in two TOAST chunks
        t = HP.Table(ATTRS)
        tids, vecs = [], []
        for i in range(8):
            v = rng.standard_normal(dim).astype(np.float32)
            note = None if i % 3 == 0 else ("inline", rng.bytes([5, 140, 400][i % 3]))
            label = None if i % 4 == 1 else struct.pack("<h", i)
            tids.append(t.insert([struct.pack("<q", i), label, note, ("inline", HP.vector_datum_body(v)), struct.pack("<i", -i)]))
            vecs.append(v)
        out[f"heap{dim}_bytes"] = np.frombuffer(t.heap.tobytes(), np.uint8)
        out[f"toast{dim}_bytes"] = np.frombuffer(t.toast.tobytes(), np.uint8)
        out[f"heap{dim}_tids"] = np.array(tids, np.uint64)
        out[f"heap{dim}_vecs"] = np.stack(vecs)
    ti = TestIndex(n=500, dim_full=32, bits=2, R=12, distance=1, seed=77, kind="gauss", deleted_frac=0.05, L_build=40)
    q = ti.queries(1, seed=5, kind="gauss")[0]
    sc = ti.oracle.scan(q, L=8, rescore=6)
    traj, rows = [], []
    keys = ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
            "node_heap_reads", "next_calls")
    for _ in range(40):
        r = sc.gettuple()
        rows.append(r[0])
        st = sc.stats()
        traj.append([st[k] for k in keys])
    out.update(cur_codes=ti.codes, cur_nbrs=ti.nbrs, cur_tids=ti.tids, cur_vecs=ti.vecs, cur_mean=ti.mean, cur_m2=ti.m2,
               cur_count=np.uint64(ti.count), cur_start=np.uint32(ti.start), cur_query=q, cur_rows=np.array(rows, np.uint32),
               cur_stats=np.array(traj, np.uint64))
    np.savez_compressed(os.path.join(HERE, "pages_golden.npz"), **out)
    print("wrote pages_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
