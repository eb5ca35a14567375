"""vs_index_autotune / tune_probe: the launch variants of k_search_fast are held to the default's rows on the caller's own batch
before one of them may become an index's choice, and whatever is chosen returns the ORACLE's rows.

On hardware the variants are launched the way the product recommends to a caller that cannot afford a misbehaving kernel in its own
process: first in a child process under a timeout (pgvectorscale_amd/tune_probe.py), then in this process without the ones that
did not come back clean.  On the interpreter (VS_EMU=1, CPU tier) every variant must be exact."""
import os

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = pytest.mark.gpu

EMU = bool(os.environ.get("VS_EMU"))
REGIME = {"VS_F_LDS_MAX_INS": "0", "VS_F_VR": "0"}  # the table-less regime of large indexes (the variants exist only there)
NAMES = ["default", "bucket_bitmap", "bucket_bitmap_16k", "cleared_tables", "slot_bitmap", "table_less", "table_less_bitmap", "lds_table_ring"]
LDS_REGIME_ONLY = NAMES[-3:]  # candidates for indexes whose default keeps the dedup table in LDS


@pytest.fixture(scope="module")
def probe_skip():
    """names the child-process probe did not clear (None: the child did not finish cleanly -> no variant is launched here)"""
    from conftest import EMU_LIB
    from pgvectorscale_amd import tune_probe
    kw = dict(n=600, nq=16, rescore=20, build_l=20, lib=EMU_LIB, timeout=600) if EMU else dict(n=20000, nq=2048, timeout=300)
    skip, rep = tune_probe.run(**kw)
    print("tune_probe:", rep)
    if EMU:
        assert rep["ok"] and skip == [], rep
        assert all(leg["ok"] for leg in rep["legs"].values()) and len(rep["legs"]) == 5, rep["legs"]
        assert sorted(rep["variants"]) == sorted(NAMES)
        assert all(v["applicable"] and v["rows_identical"] and not v["error"] and v["legs"] == 5 and not v["failed_legs"]
                   for nm, v in rep["variants"].items() if nm not in LDS_REGIME_ONLY), rep
    return skip


@pytest.fixture()
def regime():
    saved = {k: os.environ.get(k) for k in list(REGIME) + ["VS_TUNE_SABOTAGE"]}
    os.environ.update(REGIME)
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _setup(gpu_ctx, nq=24):
    ti = TestIndex(n=1500, dim_full=768, bits=2, R=50, distance=1, seed=29, kind="gauss", L_build=60)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(nq, seed=8, kind="gauss")
    dq = gpu_ctx.alloc(q.nbytes)
    gpu_ctx.upload(dq, q)
    return ti, ix, q, dq


def test_autotune_holds_every_variant_to_the_defaults_rows(gpu_ctx, probe_skip, regime):
    if probe_skip is None:
        pytest.skip("the probe child did not finish cleanly on this box: no variant is launched in this process")
    ti, ix, q, dq = _setup(gpu_ctx)
    try:
        oi, od, ost = ti.oracle.search_batch(q, L=25, rescore=30, k=10)
        rep = ix.autotune(dq, len(q), 25, 30, 10, reps=1, skip=probe_skip)
        assert [e["name"] for e in rep] == NAMES
        assert rep[0]["applicable"] and rep[0]["rows_identical"] and rep[0]["step_ms"] > 0
        assert sum(e["chosen"] for e in rep) == 1
        chosen = next(e for e in rep if e["chosen"])
        assert chosen["applicable"] and chosen["rows_identical"] and chosen["error"] == 0
        assert ix.variant() == chosen["name"]
        for e in rep[1:]:
            if e["name"] in probe_skip or e["name"] in LDS_REGIME_ONLY:
                assert not e["applicable"] and not e["chosen"]
            elif EMU:  # on the interpreter every variant exists for this geometry and is exact
                assert e["applicable"] and e["rows_identical"] and e["error"] == 0, e
            elif e["applicable"] and not e["rows_identical"]:
                import warnings
                warnings.warn(f"variant {e['name']} does not reproduce the default's rows on this hardware (disqualified, never chosen)")
        # the index's choice — and every variant that qualified — returns the oracle's rows through the ordinary entry point
        for e in rep:
            if e["applicable"] and e["rows_identical"] and not e["error"]:
                ix.set_variant(e["name"])
                gi, _, gd, gst = ix.search_batch(q, search_list_size=25, rescore=30, k=10)
                assert (gi == oi).all(), e["name"]
                assert (gd.view(np.uint32) == od.view(np.uint32)).all(), e["name"]
                for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "node_reads"):
                    assert gst[key] == ost[key], (e["name"], key)
    finally:
        gpu_ctx.free(dq)
        ix.close()


def test_a_variant_whose_rows_differ_is_disqualified(gpu_ctx, probe_skip, regime):
    if not EMU:
        pytest.skip("VS_TUNE_SABOTAGE is compiled into the interpreter build of the test tier only (-DVS_TEST_HOOKS), not into libvsgpu.so")
    if probe_skip is None:
        pytest.skip("the probe child did not finish cleanly on this box: no variant is launched in this process")
    victim = next(n for n in NAMES[1:] if n not in probe_skip and "16k" not in n)
    os.environ["VS_TUNE_SABOTAGE"] = victim  # one id of that variant's downloaded rows is flipped before the comparison
    ti, ix, q, dq = _setup(gpu_ctx)
    try:
        rep = {e["name"]: e for e in ix.autotune(dq, len(q), 25, 30, 10, reps=1, skip=probe_skip)}
        assert rep[victim]["applicable"] and not rep[victim]["rows_identical"] and not rep[victim]["chosen"]
        assert ix.variant() != victim
        assert sum(e["chosen"] for e in rep.values()) == 1
    finally:
        gpu_ctx.free(dq)
        ix.close()


def test_small_scans_try_the_table_less_regime(gpu_ctx, probe_skip):
    """a small scan keeps its dedup table in LDS by default: there the candidates are the table-less regime (plain and with the
    bitmap) and nothing else; both return the default's rows, and the oracle's"""
    ti = TestIndex(n=1500, dim_full=64, bits=2, R=32, distance=1, seed=3, kind="gauss", L_build=50)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(16, seed=2, kind="gauss")
    dq = gpu_ctx.alloc(q.nbytes)
    gpu_ctx.upload(dq, q)
    skip = list(probe_skip or [])
    if probe_skip is None:
        skip.append("table_less_bitmap")  # (its kernel has not been seen to come back on this box)
    try:
        # (a scan of a few hundred inserted ids: below VS_F_LDS_MAX_INS = 1024 expected inserts the table stays in LDS)
        oi, od, _ = ti.oracle.search_batch(q, L=6, rescore=3, k=5)
        rep = ix.autotune(dq, len(q), 6, 3, 5, reps=1, skip=skip)
        assert rep[0]["applicable"] and sum(e["chosen"] for e in rep) == 1
        for e in rep[1:]:
            if e["name"] not in LDS_REGIME_ONLY:
                assert not e["applicable"], e
            elif e["name"] not in skip:
                assert e["applicable"] and (e["rows_identical"] or not EMU), e
        for e in rep:
            if e["applicable"] and e["rows_identical"]:
                ix.set_variant(e["name"])
                gi, _, gd, _ = ix.search_batch(q, search_list_size=6, rescore=3, k=5)
                assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all(), e["name"]
    finally:
        gpu_ctx.free(dq)
        ix.close()


def test_set_variant_by_name(gpu_ctx):
    import pgvectorscale_amd as P
    ti = TestIndex(n=300, dim_full=64, bits=2, R=16, distance=1, seed=4, kind="gauss", L_build=30)
    ix = ti.upload(gpu_ctx)
    try:
        assert ix.variant() == "default"
        ix.set_variant("cleared_tables")
        assert ix.variant() == "cleared_tables"
        with pytest.raises(P.VsError):
            ix.set_variant("no_such_variant")
        assert ix.variant() == "cleared_tables"
        ix.set_variant("default")
        # a view inherits the choice of the index it was made from
        ix.set_variant("bucket_bitmap")
        v = ix.view(gpu_ctx)
        assert v.variant() == "bucket_bitmap"
        v.close()
    finally:
        ix.close()
