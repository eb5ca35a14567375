"""Deterministic synthetic corpora ("Cohere-like" / "OpenAI-like" stand-ins: a clustered mixture with low intrinsic
dimensionality embedded in `dim` dimensions, unit norm).  The stream is defined by integer arithmetic (see
csrc/vs_extra.hip) so that `fill_device` (HIP kernel, for corpora that only fit in HBM) and `rows_numpy` (this file,
for CPU-side tests) produce bit-identical f32 rows for the same (params, row index)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
K1 = 0x9E3779B97F4A7C15
K2 = 0xC2B2AE3D27D4EB4F
K3 = 0x165667B19E3779F9


@dataclass(frozen=True)
class DatagenParams:
    seed: int
    dim: int = 768
    latent_dim: int = 32
    n_clusters: int = 1024
    intra_pct: int = 60
    noise_pct: int = 10
    normalize: int = 1

    def c_struct(self):
        p = _lib.DatagenParams()
        p.seed, p.dim, p.latent_dim, p.n_clusters = self.seed, self.dim, self.latent_dim, self.n_clusters
        p.intra_pct, p.noise_pct, p.normalize = self.intra_pct, self.noise_pct, self.normalize
        return p


def _hash(seed, a, b):
    with np.errstate(over="ignore"):
        x = (np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.asarray(a, np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.asarray(b, np.uint64) * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(0x94D049BB133111EB))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _gauss(h):
    b = np.ascontiguousarray(h, np.uint64).view(np.uint8).reshape(h.shape + (8,))
    return b.sum(axis=-1, dtype=np.int64) - 1020


def _isqrt(v):
    r = 0
    while (r + 1) * (r + 1) <= v:
        r += 1
    return r


def rows_numpy(p: DatagenParams, first_row: int, rows: int) -> np.ndarray:
    """CPU twin of vs_datagen_fill: rows [first_row, first_row+rows) as float32 [rows][dim]."""
    r = np.arange(first_row, first_row + rows, dtype=np.uint64)
    j = np.arange(p.latent_dim, dtype=np.uint64)
    i = np.arange(p.dim, dtype=np.uint64)
    proj = _gauss(_hash(p.seed ^ K2, j[:, None], i[None, :]))                       # [latent][dim]
    cl = (_hash(p.seed, r, np.uint64(0)) % np.uint64(p.n_clusters)).astype(np.uint64)
    centers = _gauss(_hash(p.seed ^ K1, cl[:, None], j[None, :]))                   # [rows][latent]
    z = 100 * centers + p.intra_pct * _gauss(_hash(p.seed, r[:, None], np.uint64(1) + j[None, :]))
    noise_mult = p.noise_pct * _isqrt(p.latent_dim) * 209
    out = np.empty((rows, p.dim), np.float32)
    step = max(1, (1 << 22) // p.dim)
    for a in range(0, rows, step):
        b = min(rows, a + step)
        acc = z[a:b].astype(np.int64) @ proj.astype(np.int64)
        acc += noise_mult * _gauss(_hash(p.seed ^ K3, r[a:b, None], i[None, :]))
        xs = acc >> 10
        if p.normalize:
            ss = (xs * xs).sum(axis=1)
            out[a:b] = (xs.astype(np.float64) / np.sqrt(ss.astype(np.float64))[:, None]).astype(np.float32)
        else:
            out[a:b] = (xs.astype(np.float64) * (1.0 / 16384.0)).astype(np.float32)
    return out


def fill_device(ctx, p: DatagenParams, first_row: int, rows: int, dev_ptr):
    """Generate rows straight into HBM at dev_ptr (float32, row stride = dim)."""
    cp = p.c_struct()
    _lib.check(ctx._L.vs_datagen_fill(ctx.h, C.byref(cp), first_row, rows, dev_ptr))
