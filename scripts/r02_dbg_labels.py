#!/usr/bin/env python3
"""Which kernel path disagrees with the oracle on label-filtered scans at scale?  (diagnostics)
Builds a labeled index on the device, runs the same queries through the default path, with the second fast attempt off, and on the
general kernel only, and compares each with the CPU oracle."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import pgvectorscale_amd as P
from pgvectorscale_amd import _lib
from pgvectorscale_amd.datagen import DatagenParams, fill_device
from bench import zipf_labels, label_start_nodes
from oracle import oracle_py as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim, NL, nq, k = 1536, 32, 32768, 10
L, S = 75, 108
ctx = P.Context(0)
ix = P.DiskAnnIndex.alloc(ctx, n=n, dim_full=dim, num_neighbors=50, distance_type=P.VS_COSINE)
gp = DatagenParams(seed=8, dim=dim)
vp, _ = ix.array(_lib.ARR_VECS)
fill_device(ctx, gp, 0, n, vp)
ix.refresh_norms(); ix.sbq_train(); ix.sbq_quantize_corpus()
lo, lv = zipf_labels(np, n, NL, 108, 1, 3)
starts = label_start_nodes(np, lo, lv)
ix.set_labels(lo, lv)
t0 = time.time(); ix.build_graph(search_list_size=100, max_alpha=1.2); print("build", time.time() - t0, "unreachable", ix.build_unreachable(), flush=True)
ix.set_start_nodes(0, starts)
q = ctx.alloc(nq * dim * 4)
fill_device(ctx, gp, 1 << 40, nq, q)
rng = np.random.default_rng(5)
keys = [[int(rng.integers(1, NL + 1))] if i % 2 == 0 else sorted(set(int(v) for v in rng.integers(1, NL + 1, 2))) for i in range(nq)]
off = np.zeros(nq + 1, np.uint32); vals = []
for i, kk in enumerate(keys):
    vals += kk; off[i + 1] = len(vals)
vals = np.array(vals, np.int16)
d_val = ctx.alloc(vals.size * 2); d_off = ctx.alloc(off.size * 4)
ctx.upload(d_val, vals); ctx.upload(d_off, off)
out = ctx.alloc(nq * k * 4); outd = ctx.alloc(nq * k * 4)
host = ix.download(vecs=True)
mean, m2, cnt = ix.get_quantizer()
oidx = O.OracleIndex(codes=host["codes"], nbrs=host["nbrs"], heap_tids=host["heap_tids"], vecs=host["vecs"], mean=mean, m2=m2, count=cnt,
                     bits=ix.desc.bits, dim_index=dim, num_neighbors=50, distance_type=O.COSINE, default_start=0, label_off=lo, label_val=lv,
                     label_starts=starts)
qh = ctx.download(q, np.empty((nq, dim), np.float32))
t0 = time.time(); oi, od, ost = oidx.search_batch(qh, L=L, rescore=S, k=k, threads=16, qlabels=keys); print("oracle", time.time() - t0, flush=True)
for name, env in (("default", {}), ("retry_off", {"VS_F_RETRY": "0"}), ("general_only", {"VS_FAST": "0"}), ("default_again", {})):
    for k_, v_ in env.items():
        os.environ[k_] = v_
    os.environ["VS_DEBUG_STATUS"] = "1"
    for rep in range(2):
        ix.search_batch_dev(q, nq, L, S, k, out, d_out_dist=outd, d_qlabels=d_val, d_qlabel_off=d_off) if False else \
            ix.search_batch_dev(q, nq, L, S, k, out, None, outd, d_qlabels=d_val, d_qlabel_off=d_off)
        st = ix.search_batch_dev_finish()
    gi = ctx.download(out, np.empty((nq, k), np.uint32))
    gd = ctx.download(outd, np.empty((nq, k), np.float32))
    bad = np.nonzero((gi != oi).any(1))[0]
    print(f"{name:14s}: {len(bad)} of {nq} scans differ; fallback_scans={st['fallback_scans']} retries={st['retries']} visits/q={st['visited_nodes'] / nq:.1f}", flush=True)
    for b in bad[:6]:
        first = int(np.nonzero(gi[b] != oi[b])[0][0])
        print("   q", int(b), "key", keys[b], "first diff at", first, "gpu", gi[b][first:first + 3], gd[b][first:first + 3], "oracle", oi[b][first:first + 3], od[b][first:first + 3])
    for k_ in env:
        os.environ.pop(k_)
