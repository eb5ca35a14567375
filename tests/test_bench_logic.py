"""Host logic of bench.py that can be checked without a GPU: the operating-point search and the graph-cache naming."""
import math
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _model(noise=0.0):
    calls = []

    def run_sample(L, S):
        calls.append((L, S))
        recall = 1.0 - 0.9 * math.exp(-L / 60.0) * math.exp(-S / 45.0) - 0.004
        visits = 1.1 * L + (S + 9 if S else 10) + 8
        return recall, {"visited_nodes": visits * 1000, "queries": 1000}
    return run_sample, calls


def test_choose_operating_point_takes_the_cheapest_passing_point():
    run_sample, calls = _model()
    log = []
    L, S, rec = bench.choose_operating_point(run_sample, 10, 0.99, log)
    assert rec >= 0.99
    # exhaustive check on a fine grid: nothing that passes is more than a few percent cheaper
    def cost(l, s):
        return 1.1 * l + (s + 9) + 8 + 0.12 * (s + 9)
    best = min(cost(l, s) for l in (50, 75, 100, 150, 200, 400) for s in range(1, 401) if run_sample(l, s)[0] >= 0.99)
    assert cost(L, S) <= 1.08 * best
    assert len(log) == len(set((a, b) for a, b, _ in log)) and len(log) <= 70  # 12 list sizes x 5 windows at most, plus the bisection


def test_choose_operating_point_reports_the_best_point_when_the_target_is_out_of_reach():
    def run_sample(L, S):
        return 0.5 + L / 1000.0 + S / 10000.0, {"visited_nodes": 1000 * L, "queries": 1000}
    L, S, rec = bench.choose_operating_point(run_sample, 10, 0.99, [])
    assert (L, S) == (400, 400) and rec < 0.99


def test_choose_operating_point_survives_failing_launches():
    class Boom(Exception):
        pass

    def run_sample(L, S):
        if L < 100:
            raise Boom("capacity")
        return (0.995 if S >= 50 else 0.9), {"visited_nodes": 1000 * (L + S), "queries": 1000}
    L, S, rec = bench.choose_operating_point(run_sample, 10, 0.99, [], Boom)
    assert L == 100 and 25 < S <= 50 and rec >= 0.99


def test_graph_cache_path():
    a = types.SimpleNamespace(graph_cache="none", distance="l2", build_l=100)
    assert bench.graph_cache_path(a, 50_000_000, 768, 6, 2, 50) is None
    a.graph_cache = "/tmp/x"
    assert bench.graph_cache_path(a, 1_000_000, 768, 3, 2, 50) == "/tmp/x.1000000x768.l2.b2.R50.L100.s3"
    a.graph_cache = "auto"
    assert bench.graph_cache_path(a, 1_000_000, 768, 3, 2, 50) is None  # small builds are not worth caching
    p = bench.graph_cache_path(a, 10_000_000, 768, 5, 2, 50)
    assert p is None or ("vs_graph_cache_" in p and p.endswith(".10000000x768.l2.b2.R50.L100.s5"))


def test_usable_cores_respects_affinity_and_quota():
    """cpu_baseline.cores must describe what ran: the affinity mask capped by the cgroup quota, not os.cpu_count()"""
    c = bench.usable_cores()
    assert 1 <= c["usable"] <= c["affinity"] <= max(c["os_cpu_count"], c["affinity"])
    if c["cgroup_quota"]:
        assert c["usable"] <= int(c["cgroup_quota"] + 0.5) or c["usable"] == 1


def test_label_workload_helpers():
    """--labels: label sets are sorted sets of 1..NL with 1-3 members, start nodes are first carriers, masks match the CSR"""
    import numpy as np
    import bench
    off, val = bench.zipf_labels(np, 5000, 32, 7, 1, 3)
    assert off[0] == 0 and off[-1] == val.size and val.dtype == np.int16
    cnt = np.diff(off.astype(np.int64))
    assert cnt.min() >= 1 and cnt.max() <= 3 and val.min() >= 1 and val.max() <= 32
    for i in range(0, 5000, 97):
        row = val[off[i]:off[i + 1]].tolist()
        assert row == sorted(set(row))
    freq = np.bincount(val, minlength=33)
    assert freq[1] > freq[4] > freq[16] > 0  # Zipf: frequencies fall with the label number
    starts = bench.label_start_nodes(np, off, val)
    for l, v in starts.items():
        assert l in val[off[v]:off[v + 1]]
        assert not any(l in val[off[u]:off[u + 1]] for u in range(v))
    m = bench.label_masks(np, off, val)
    for i in (0, 1, 4999):
        assert m[i] == sum(1 << int(x) for x in val[off[i]:off[i + 1]])
    off2, val2 = bench.zipf_labels(np, 5000, 32, 7, 1, 3)
    assert (off2 == off).all() and (val2 == val).all()  # deterministic
