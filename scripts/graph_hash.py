import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pgvectorscale_amd as P
from pgvectorscale_amd import _lib
from pgvectorscale_amd.datagen import DatagenParams, fill_device
ctx = P.Context(0)
n = 1_000_000
ix = P.DiskAnnIndex.alloc(ctx, n=n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
vp, _ = ix.array(_lib.ARR_VECS)
fill_device(ctx, DatagenParams(seed=3, dim=768), 0, n, vp)
ix.refresh_norms(); ix.sbq_train(); ix.sbq_quantize_corpus()
t0 = time.time(); ix.build_graph(search_list_size=100, max_alpha=1.2); ctx.sync(); t = time.time() - t0
ptr, stride = ix.array(_lib.ARR_NBRS)
arr = np.empty((n, stride), np.uint32); ctx.download(ptr, arr)
print(f"build {t:.3f}s graph sha1 {hashlib.sha1(arr.tobytes()).hexdigest()}")
