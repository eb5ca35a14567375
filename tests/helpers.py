"""Shared builders for parity tests: small `diskann` indexes manufactured with the ORACLE (training, quantisation,
Vamana graph) so that the GPU path and the oracle search exactly the same flat arrays."""
import functools

import numpy as np

from oracle import oracle_py as O


def make_vectors(n, d, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":  # mirrors the reference tests' random() data (AM/build.rs:1226)
        return rng.random((n, d), dtype=np.float32)
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "clustered":
        c = rng.standard_normal((32, d)).astype(np.float32)
        x = c[rng.integers(0, 32, n)] + 0.4 * rng.standard_normal((n, d)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    raise ValueError(kind)


class TestIndex:
    """Flat arrays of one index + the oracle view of it."""
    __test__ = False

    def __init__(self, n=2000, dim_full=64, dim_index=None, bits=None, R=32, distance=O.L2, seed=1, kind="uniform",
                 n_labels=0, deleted_frac=0.0, L_build=50, label_zipf=False):
        dim_index = dim_index or dim_full
        bits = bits or O.default_bits(dim_index)
        self.n, self.dim_full, self.dim_index, self.bits, self.R, self.distance = n, dim_full, dim_index, bits, R, distance
        X = make_vectors(n, dim_full, seed, kind)
        self.vecs = X
        # the index stores quantize(normalised index slice) (AM/pg_vector.rs:143-157)
        idx_slice = np.ascontiguousarray(X[:, :dim_index]).copy()
        if distance == O.COSINE:
            for i in range(n):
                idx_slice[i] = O.preprocess_cosine(idx_slice[i])[0]
        self.mean, self.m2, self.count = O.train(idx_slice, bits)
        self.codes = O.quantize(self.mean, self.m2, self.count, bits, idx_slice)
        self.nbrs, self.start = O.build_graph(self.codes, num_neighbors=R, search_list_size=L_build)
        rng = np.random.default_rng(seed + 1000)
        self.tids = ((np.arange(n, dtype=np.uint64) + 7) << np.uint64(16)) | np.uint64(1)
        if deleted_frac > 0:
            dele = rng.random(n) < deleted_frac
            self.tids[dele] &= ~np.uint64(0xFFFF)
        self.label_off = self.label_val = None
        self.label_starts = {}
        if n_labels:
            off = np.zeros(n + 1, np.uint32)
            vals = []
            pz = 1.0 / np.arange(1, n_labels + 1)  # Zipf, s = 1
            pz /= pz.sum()
            for i in range(n):
                k = int(rng.integers(1, 4))
                if label_zipf:
                    ls = sorted(set(int(v) + 1 for v in rng.choice(n_labels, k, p=pz)))
                else:
                    ls = sorted(set(int(v) for v in rng.integers(1, n_labels + 1, k)))
                vals.extend(ls)
                off[i + 1] = len(vals)
                for l in ls:  # first node carrying a label becomes that label's start node
                    self.label_starts.setdefault(l, i)
            self.label_off, self.label_val = off, np.array(vals, np.int16)
        self.oracle = O.OracleIndex(codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs,
                                    mean=self.mean, m2=self.m2, count=self.count, bits=bits, dim_index=dim_index,
                                    num_neighbors=R, distance_type=distance, default_start=self.start,
                                    label_off=self.label_off, label_val=self.label_val, label_starts=self.label_starts)

    def upload(self, ctx):
        import pgvectorscale_amd as P
        return P.DiskAnnIndex.upload(ctx, codes=self.codes, nbrs=self.nbrs, heap_tids=self.tids, vecs=self.vecs,
                                     mean=self.mean, m2=self.m2, count=self.count, bits=self.bits,
                                     dim_index=self.dim_index, num_neighbors=self.R, distance_type=self.distance,
                                     default_start=self.start, label_off=self.label_off, label_val=self.label_val,
                                     label_starts=self.label_starts)

    def queries(self, nq, seed=99, kind="uniform"):
        return make_vectors(nq, self.dim_full, seed, kind)


@functools.lru_cache(maxsize=None)
def cached_index(**kw):
    return TestIndex(**kw)
