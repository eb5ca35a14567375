"""Row counts the reference's own pg_tests assert on tiny labeled tables (exact numbers, so they pin the filter semantics of
the scan: start nodes per label, LabelSet overlap, the empty key, labels nobody carries, unsorted keys, NULL elements):
  * test_tiny_labeled_index        AM/labels/filtering_tests.rs:663-715
  * test_null_and_empty_labels     AM/labels/filtering_tests.rs:23-109
  * test_build_index_on_nonempty_table   AM/labels/filtering_tests.rs:112-167
  * test_label_size_bounds         AM/labels/filtering_tests.rs:718-793 (labels 0, 32767 and -1 are legal smallints; the
    reference checks them with the `&&` operator alone, here they also go through the index scan)
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
    def __init__(self, vecs, labels, R=15):
        self.vecs = np.array(vecs, np.float32)
        n, dim = self.vecs.shape
        self.n, self.dim, self.R = n, dim, R
        unit = np.stack([O.preprocess_cosine(v)[0] for v in self.vecs])  # `<=>`: the index stores the normalised vector's code
        self.bits = O.default_bits(dim)
        self.mean, self.m2, self.count = O.train(unit, self.bits)
        self.codes = O.quantize(self.mean, self.m2, self.count, self.bits, unit)
        self.nbrs = np.full((n, R), 0xFFFFFFFF, np.uint32)
        for i in range(n):
            others = [j for j in range(n) if j != i]
            self.nbrs[i, :len(others)] = others
        self.tids = ((np.arange(n, dtype=np.uint64) + 1) << np.uint64(16)) | np.uint64(1)
        self.label_off = np.zeros(n + 1, np.uint32)
        vals = []
        self.label_starts = {}
        for i, ls in enumerate(labels):
            vals.extend(sorted(set(ls)))
            self.label_off[i + 1] = len(vals)
            for l in ls:
                self.label_starts.setdefault(l, i)
        self.label_val = np.array(vals, np.int16)
        self.oracle = O.OracleIndex(codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=self.mean, m2=self.m2,
                                    count=self.count, bits=self.bits, dim_index=dim, num_neighbors=R, distance_type=O.COSINE,
                                    default_start=0, label_off=self.label_off, label_val=self.label_val,
                                    label_starts=self.label_starts)

    def upload(self, ctx):
        import pgvectorscale_amd as P
        return P.DiskAnnIndex.upload(ctx, codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs, mean=self.mean,
                                     m2=self.m2, count=self.count, bits=self.bits, dim_index=self.dim, num_neighbors=self.R,
                                     distance_type=O.COSINE, default_start=0, label_off=self.label_off, label_val=self.label_val,
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
