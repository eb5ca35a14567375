"""The workspace slab (WsSlab in csrc/vs_internal.h): the persistent grid's dedup tables and heap spill arrays live in one
grow-only slab per index that the library CHOOSES among probed candidates (device memory is not uniform for the search kernel's request
mix), shared with the index's views; vs_index_set_slab hands the library the caller's memory instead; vs_ws_probe / vs_ws_probe_mix are
the probes.  Whatever the placement, rows, distance bits and counters are the oracle's; nothing leaks."""
import os

import numpy as np
import pytest

import pgvectorscale_amd as P
from helpers import cached_index

pytestmark = pytest.mark.gpu

KW = dict(n=2500, dim_full=64, bits=2, R=24, distance=1, seed=17, kind="gauss", L_build=48)


def _check(ix, ti, q, L=40, S=20):
    gi, _, gd, gst = ix.search_batch(q, search_list_size=L, rescore=S, k=10)
    oi, od, ost = ti.oracle.search_batch(q, L=L, rescore=S, k=10)
    assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    for c in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons"):
        assert gst[c] == ost[c], c


@pytest.mark.parametrize("mode", ["chosen_among_candidates", "callers_memory", "no_slab"])
def test_placement_never_changes_a_row(gpu_ctx, oracle, monkeypatch, mode):
    ti = cached_index(**KW)
    monkeypatch.setenv("VS_F_LDS_MAX_INS", "0")  # the table-less regime (the persistent grid's regions are what the slab holds)
    free0, _ = gpu_ctx.mem_info()
    mem = None
    if mode == "chosen_among_candidates":
        monkeypatch.setenv("VS_WS_SLAB_MIN_N", "1")   # (indexes of 4M nodes and more get one by default)
        monkeypatch.setenv("VS_WS_SLAB_MB", "128")
        monkeypatch.setenv("VS_WS_SLAB_CANDIDATES", "3")
    elif mode == "no_slab":
        monkeypatch.setenv("VS_WS_SLAB_MB", "0")
    ix = ti.upload(gpu_ctx)
    ctx2 = P.Context(0)
    try:
        if mode == "callers_memory":
            nbytes = 96 << 20
            mem = gpu_ctx.alloc(nbytes)
            ix.set_slab(mem, nbytes)
        q = ti.queries(40, seed=5, kind="gauss")
        _check(ix, ti, q)
        vw = ix.view(ctx2)  # a view sub-allocates from the same slab
        try:
            _check(vw, ti, q)
            _check(ix, ti, q, L=80, S=50)  # a larger operating point: the arrays grow inside the slab
        finally:
            vw.close()
        if mode == "callers_memory":
            with pytest.raises(P.VsError):  # only before the handle's first search
                ix.set_slab(mem, 96 << 20)
    finally:
        ix.close()
        ctx2.close()
        if mem is not None:
            gpu_ctx.free(mem)
    if not os.environ.get("VS_EMU"):
        free1, _ = gpu_ctx.mem_info()
        assert free1 >= free0 - (64 << 20), (free0, free1)  # candidates, spacers and the slab itself went back to the device


def test_probes_run_on_any_region(gpu_ctx, oracle):
    ti = cached_index(**KW)
    ix = ti.upload(gpu_ctx)
    nbytes = 128 << 20
    mem = gpu_ctx.alloc(nbytes)
    try:
        a = gpu_ctx.ws_probe(mem, nbytes, iters=4)
        b = ix.ws_probe_mix(mem, nbytes, iters=4)
        assert a > 0 and b > 0
        with pytest.raises(P.VsError):
            gpu_ctx.ws_probe(mem, 1 << 20, iters=4)  # at least 64 MiB
    finally:
        gpu_ctx.free(mem)
        ix.close()


def test_prepare_workspace_chooses_the_slab_before_the_first_search(gpu_ctx, oracle, monkeypatch):
    """vs_index_prepare_workspace (advisor, round 5: the probing's transient allocations at a moment of the host's choosing, bounded to a
    share of the free memory): the slab exists after the call — the first search allocates no slab of its own — and rows are the oracle's"""
    ti = cached_index(**KW)
    monkeypatch.setenv("VS_F_LDS_MAX_INS", "0")
    monkeypatch.setenv("VS_WS_SLAB_MIN_N", "1")
    monkeypatch.setenv("VS_WS_SLAB_MB", "128")
    monkeypatch.setenv("VS_WS_SLAB_CANDIDATES", "3")
    monkeypatch.setenv("VS_WS_SLAB_PROBE_PCT", "10")
    ix = ti.upload(gpu_ctx)
    try:
        free0, _ = gpu_ctx.mem_info()
        ix.prepare_workspace()
        free1, _ = gpu_ctx.mem_info()
        real = not os.environ.get("VS_EMU")  # (the interpreter's hipMemGetInfo does not move)
        if real:
            assert 64 << 20 <= free0 - free1 <= 2 * (128 << 20) + (8 << 20), (free0, free1)  # one slab (or a pair) stays, the other candidates and the spacers went back
        ix.prepare_workspace()  # a second call is a no-op
        assert gpu_ctx.mem_info()[0] == free1
        q = ti.queries(24, seed=6, kind="gauss")
        _check(ix, ti, q)
        free2, _ = gpu_ctx.mem_info()
        assert not real or free1 - free2 < (64 << 20), (free1, free2)  # the search's regions came out of the slab: no second slab-sized allocation
    finally:
        ix.close()
