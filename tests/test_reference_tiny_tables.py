"""Row counts the reference's own pg_tests assert on tiny labeled tables (exact numbers, so they pin the filter semantics of
the scan: start nodes per label, LabelSet overlap, the empty key, labels nobody carries, unsorted keys, NULL elements):
  * test_tiny_labeled_index        AM/labels/filtering_tests.rs:663-715
  * test_null_and_empty_labels     AM/labels/filtering_tests.rs:23-109
  * test_build_index_on_nonempty_table   AM/labels/filtering_tests.rs:112-167
  * test_label_size_bounds         AM/labels/filtering_tests.rs:718-793 (labels 0, 32767 and -1 are legal smallints; the
    reference checks them with the `&&` operator alone, here they also go through the index scan)
and on tiny unlabeled ones:
  * test_l2_sanity_check / test_ip_sanity_check   AM/build.rs:1476-1556 (the first row of ORDER BY ... LIMIT 1)
  * test_empty_table_insert / test_insert_empty_insert   AM/build.rs:1559-1611 (counts; deleted rows stay in the graph with
    a dead heap pointer)
The graph of a 3-4 row table with num_neighbors = 15 is complete; start node of a label = the first row carrying it
(AM/graph/start_nodes.rs).  The oracle must give the reference's counts (CPU tier), the HIP path must give the oracle's rows
(GPU tier)."""
import numpy as np
import pytest

from oracle import oracle_py as O

TINY = dict(vecs=[[1, 2, 3], [4, 5, 6], [7, 8, 10]], labels=[[1, 2], [1, 3], [2, 3]],
            expect=[(None, 3), ([1], 2), ([2], 2), ([3], 2), ([1, 3], 3), ([1, 2, 3], 3), ([4], 0), ([1, 4], 2), ([4, 1], 2)])
# NULL array and '{}' are the empty label set; a NULL element is dropped ('{1,NULL,3}' -> {1,3})
NULLS = dict(vecs=[[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12]], labels=[[1, 2], [], [], [1, 3]],
             expect=[([1], 2), ([], 0), ([3], 1), (None, 4)])
NONEMPTY = dict(vecs=[[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12], [13, 14, 15]], labels=[[1, 2], [1, 3], [2, 3], [4, 5], []],
                expect=[([1], 2), ([2, 3], 3), ([5], 1), (None, 5)])
BOUNDS = dict(vecs=[[1, 2, 3], [7, 8, 9]], labels=[[0, 32767], [-1]],
              expect=[([32767], 1), ([-1], 1), ([0], 1), ([-1, 32767], 2), ([-32768], 0), (None, 2)])
TABLES = [TINY, NULLS, NONEMPTY, BOUNDS]
IDS = ["test_tiny_labeled_index", "test_null_and_empty_labels", "test_build_index_on_nonempty_table", "test_label_size_bounds"]


class TinyTable:
    def __init__(self, vecs, labels, R=15, distance=O.COSINE, deleted=()):
        self.vecs = np.array(vecs, np.float32)
        n, dim = self.vecs.shape
        self.n, self.dim, self.R, self.distance = n, dim, R, distance
        # `<=>`: the index stores the normalised vector's code
        unit = np.stack([O.preprocess_cosine(v)[0] for v in self.vecs]) if distance == O.COSINE else self.vecs
        self.bits = O.default_bits(dim)
        self.mean, self.m2, self.count = O.train(unit, self.bits)
        self.codes = O.quantize(self.mean, self.m2, self.count, self.bits, unit)
        self.nbrs = np.full((n, R), 0xFFFFFFFF, np.uint32)
        for i in range(n):
            others = [j for j in range(n) if j != i]
            self.nbrs[i, :len(others)] = others
        self.tids = ((np.arange(n, dtype=np.uint64) + 1) << np.uint64(16)) | np.uint64(1)
        for i in deleted:
            self.tids[i] &= ~np.uint64(0xFFFF)
        self.label_off = self.label_val = None
        self.label_starts = {}
        if labels is not None:
            self.label_off = np.zeros(n + 1, np.uint32)
            vals = []
            for i, ls in enumerate(labels):
                vals.extend(sorted(set(ls)))
                self.label_off[i + 1] = len(vals)
                for l in ls:
                    self.label_starts.setdefault(l, i)
            self.label_val = np.array(vals, np.int16)
        self.oracle = O.OracleIndex(codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=self.mean, m2=self.m2,
                                    count=self.count, bits=self.bits, dim_index=dim, num_neighbors=R, distance_type=distance,
                                    default_start=0, label_off=self.label_off, label_val=self.label_val,
                                    label_starts=self.label_starts)

    def upload(self, ctx):
        import pgvectorscale_amd as P
        return P.DiskAnnIndex.upload(ctx, codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=self.mean,
                                     m2=self.m2, count=self.count, bits=self.bits, dim_index=self.dim, num_neighbors=self.R,
                                     distance_type=self.distance, default_start=0, label_off=self.label_off, label_val=self.label_val,
                                     label_starts=self.label_starts)


def drain(scan):
    rows = []
    while True:
        r = scan.gettuple()
        if r is None:
            return rows
        rows.append(r)


@pytest.mark.parametrize("table", TABLES, ids=IDS)
def test_oracle_gives_the_reference_counts(oracle, table):
    t = TinyTable(table["vecs"], table["labels"])
    q = np.zeros(3, np.float32)  # ORDER BY embedding <=> '[0,0,0]'
    for key, count in table["expect"]:
        rows = drain(t.oracle.scan(q, labels=key, L=100, rescore=50))  # the session defaults of the GUCs
        assert len(rows) == count, (key, rows)
        assert len({r[0] for r in rows}) == count
        for node, _, _ in rows:
            assert key is None or set(table["labels"][node]) & set(key)


@pytest.mark.gpu
@pytest.mark.parametrize("table", TABLES, ids=IDS)
def test_hip_path_gives_the_reference_counts(gpu_ctx, oracle, table):
    t = TinyTable(table["vecs"], table["labels"])
    ix = t.upload(gpu_ctx)
    q = np.zeros(3, np.float32)
    scan = ix.beginscan()
    for key, count in table["expect"]:
        scan.rescan(q, labels=key, search_list_size=100, rescore=50)
        rows = drain(scan)
        want = drain(t.oracle.scan(q, labels=key, L=100, rescore=50))
        assert len(rows) == count
        assert [r[1] for r in rows] == [w[0] for w in want] and [r[0] for r in rows] == [w[1] for w in want]
        assert scan.xs_recheck == (key is not None)
    scan.endscan()
    ix.close()


# (distance, rows, [(query, the row ORDER BY ... LIMIT 1 returns)])
SANITY = {
    "test_l2_sanity_check": (O.L2, [[1, 1, 1], [2, 2, 2], [3, 3, 3]], [([1, 1, 1], 0), ([2, 2, 2], 1), ([3, 3, 3], 2)]),
    "test_ip_sanity_check": (O.IP, [[1, 1, 1], [2, 2, 2], [3, 3, 3]], [([1, 1, 1], 2), ([2, 2, 2], 2), ([3, 3, 3], 2)]),
}
# (rows, deleted rows, count of `order by embedding <=> '[0,0,0]'`)
COUNTS = {
    "test_empty_table_insert": ([[1, 2, 3], [4, 5, 6], [7, 8, 10]], (), 3),
    "test_insert_empty_insert": ([[1, 2, 3], [4, 5, 6], [7, 8, 10], [1, 2, 3], [14, 15, 16]], (0, 1, 2), 2),
}


@pytest.mark.parametrize("name", list(SANITY))
def test_oracle_first_row_of_the_sanity_checks(oracle, name):
    distance, rows, expect = SANITY[name]
    t = TinyTable(rows, None, R=10, distance=distance)
    for q, first in expect:
        r = t.oracle.scan(np.array(q, np.float32), L=100, rescore=50).gettuple()
        assert r is not None and r[0] == first


@pytest.mark.parametrize("name", list(COUNTS))
def test_oracle_counts_with_deleted_rows(oracle, name):
    rows, deleted, count = COUNTS[name]
    t = TinyTable(rows, None, deleted=deleted)
    got = drain(t.oracle.scan(np.zeros(3, np.float32), L=100, rescore=50))
    assert len(got) == count and not ({r[0] for r in got} & set(deleted))


@pytest.mark.gpu
def test_hip_path_on_the_unlabeled_tiny_tables(gpu_ctx, oracle):
    for name, (distance, rows, expect) in SANITY.items():
        t = TinyTable(rows, None, R=10, distance=distance)
        ix = t.upload(gpu_ctx)
        scan = ix.beginscan()
        for q, first in expect:
            scan.rescan(np.array(q, np.float32), search_list_size=100, rescore=50)
            r = scan.gettuple()
            assert r is not None and r[1] == first, name
        scan.endscan()
        gi, _, _, _ = ix.search_batch(np.array([q for q, _ in expect], np.float32), search_list_size=100, rescore=50, k=1)
        assert gi[:, 0].tolist() == [f for _, f in expect], name
        ix.close()
    for name, (rows, deleted, count) in COUNTS.items():
        t = TinyTable(rows, None, deleted=deleted)
        ix = t.upload(gpu_ctx)
        scan = ix.beginscan()
        scan.rescan(np.zeros(3, np.float32), search_list_size=100, rescore=50)
        got = drain(scan)
        want = drain(t.oracle.scan(np.zeros(3, np.float32), L=100, rescore=50))
        assert len(got) == count and [r[1] for r in got] == [w[0] for w in want], name
        scan.endscan()
        ix.close()
