"""Oracle vs every known-answer test / golden vector the reference holds for this path.
CPU only."""
import os

import numpy as np
import pytest


def _seq_norm(v):
    s = np.float32(0)
    for x in v:
        s = np.float32(s + np.float32(x * x))
    return np.float32(np.sqrt(s))


def test_distances_equal_kat(oracle):
    """AM/distance/distance_x86.rs:41-62 `distances_equal`: |simd - scalar| < 1e-6 on 2000-d normalised ramps."""
    O = oracle
    r = np.arange(2000, dtype=np.float32) + 1
    l = np.arange(2000, dtype=np.float32) + 2
    r = (r / _seq_norm(r)).astype(np.float32)
    l = (l / _seq_norm(l)).astype(np.float32)
    assert abs(float(O.distance_cosine(r, l)) - float(O.distance_cosine_unoptimized(r, l))) < 1e-6
    assert abs(float(O.distance_l2(r, l)) - float(O.distance_l2_unoptimized(r, l))) < 1e-6


def test_simd_lane_emulation_matches_real_avx2(oracle):
    """The scalar restatement of the 4x8-lane AVX2 accumulation must be bit-identical to real AVX2/FMA intrinsics."""
    O = oracle
    if not O.have_avx2():
        pytest.skip("host without AVX2+FMA")
    rng = np.random.default_rng(5)
    for n in (1, 7, 31, 32, 33, 64, 100, 128, 768, 1536, 2000):
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        assert O.distance_l2(a, b).tobytes() == O.distance_l2_avx2(a, b).tobytes()
        assert O.inner_product(a, b).tobytes() == O.inner_product_avx2(a, b).tobytes()


def test_bench_inputs_with_derivable_answers(oracle):
    """benches/distance.rs:299-305: 1536-bit patterns i%2==0 vs i%3==0 -> Hamming = 768+512-2*256 = 768."""
    O = oracle
    a = np.array([i % 2 == 0 for i in range(1536)])
    b = np.array([i % 3 == 0 for i in range(1536)])
    pack = lambda x: np.packbits(x, bitorder="little").view(np.uint64)
    assert pack(a).size == 24
    assert O.distance_xor(pack(a), pack(b)) == 768
    # benches/distance.rs:145-146 ramps: exact L2^2 = 2000 * 1000.1^2 in real arithmetic; f32 result within 1e-3 rel
    r = (np.arange(2000, dtype=np.float32) + np.float32(1000.1)).astype(np.float32)
    l = (np.arange(2000, dtype=np.float32) + np.float32(2000.2)).astype(np.float32)
    exact = float(np.sum((r.astype(np.float64) - l.astype(np.float64)) ** 2))
    assert abs(float(O.distance_l2(r, l)) - exact) / exact < 1e-5


@pytest.mark.parametrize("a,b,expect", [
    ([], [1, 2, 3], False),                        # test_overlaps_empty           AM/labels/mod.rs:253-259
    ([1, 2], [2, 3], True),                        # test_overlaps_non_empty       :261-267
    ([1, 2], [3, 4], False),                       # test_overlaps_no_overlap      :269-275
    ([1, 2, 3, 4, 5], [1, 2, 3, 4], True),         # test_overlaps_longer          :277-283
    ([1, 2, 3, 4, 5], [6, 7, 8, 9, 10], False),    # test_overlaps_non_empty_no_overlap :285-291
    ([1, 2, 3, 4, 5], [2, 3, 4, 5, 6], True),      # test_overlaps_non_empty_overlap    :293-299
    ([1, 3, 5, 10, 11], [2, 4, 6, 8, 11], True),   # test_overlaps_interleavings   :301-307
])
def test_label_overlaps_kat(oracle, a, b, expect):
    assert oracle.labels_overlap(a, b) is expect
    assert oracle.labels_overlap(b, a) is expect


def test_contains_intersection_kat(oracle):
    """AM/labels/mod.rs:309-330 (first three contains_intersection tests)."""
    O = oracle
    a, b = [1, 3, 5, 10, 11], [2, 4, 6, 8, 11]
    c = list(range(1, 12))
    assert O.labels_contains_intersection(c, a, b) and O.labels_contains_intersection(c, b, a)
    assert O.labels_contains_intersection([1, 2, 3], [], [1, 2, 3])
    assert O.labels_contains_intersection([1, 2, 3], [1, 2, 3], [])
    assert O.labels_contains_intersection([7], [1, 2, 3], [4, 5, 6])  # empty intersection is contained in anything


@pytest.mark.parametrize("l,r,expect", [
    ([None, None, None], [None, None, None], False),   # test_empty_overlap   AM/mod.rs:324-334
    ([3, 1, 2], [6, 2, 4], True),                      # test_simple_overlap  :336-346
    ([3, 1, 2], [8, 4, 6], False),                     # test_no_overlap      :348-358
    ([2, 1, 3, 2], [6, 4, 2], True),                   # test_repeated_overlap :360-370
    ([19, 3, 2, 15, 7, 1, 14, 10, 11, 8, 13, 12, 9, 16, 17, 18, 2], [8, 6, 2, 4, 7], True),   # :372-383
    ([33, 2, 30, 5, 10, 20, 23, 24, 25, 26, 27, 28, 29, 1, 31, 32, 3], [9, 4, 8, 6], False),  # :385-396
])
def test_smallint_array_overlap_kat(oracle, l, r, expect):
    assert oracle.smallint_array_overlap(l, r) is expect


def test_labelset_from_sorts_and_dedups(oracle):
    assert list(oracle.labelset([3, 1, 2, 3, 1])) == [1, 2, 3]  # AM/labels/mod.rs:30-37


def test_quantized_size_and_default_bits(oracle):
    O = oracle
    assert O.quantized_size(128, 2) == 4 and O.quantized_size(768, 2) == 24 and O.quantized_size(1536, 1) == 24
    assert O.quantized_size(65, 1) == 2 and O.quantized_size(16000, 1) == 250
    assert O.default_bits(768) == 2 and O.default_bits(899) == 2 and O.default_bits(900) == 1 and O.default_bits(1536) == 1


def test_quantize_semantics(oracle):
    """No KAT exists in the reference for SbqQuantizer::quantize; pin the restated rules on hand-computable cases
    (AM/sbq/quantize.rs:52-102)."""
    O = oracle
    # 1 bit: strictly greater than the mean, LSB-first
    mean = np.zeros(70, np.float32)
    v = np.zeros(70, np.float32)
    v[0] = 1
    v[63] = 1
    v[64] = 2
    v[69] = -1
    c = O.quantize(mean, None, 10, 1, v)
    assert c[0] == (1 | (1 << 63)) and c[1] == 1
    # 2 bits: std=1 (m2/count = 1), thresholds at z=-2+4/3 and z=-2+8/3
    mean = np.zeros(4, np.float32)
    m2 = np.full(4, 10, np.float32)
    v = np.array([-1.0, -0.6, 0.7, 5.0], np.float32)   # index = (z+2)/(4/3): 0.75, 1.05, 2.025, 5.25
    c = O.quantize(mean, m2, 10, 2, v)
    assert c[0] == 0b11_11_01_00
    # std = 0 and v == mean -> NaN index -> `NaN < 1.0` false -> `NaN as usize` = 0 ones; v > mean -> +inf -> all ones
    m2 = np.zeros(2, np.float32)
    c = O.quantize(np.zeros(2, np.float32), m2, 10, 2, np.array([0.0, 1.0], np.float32))
    assert c[0] == 0b11_00


def test_welford_training(oracle):
    """add_sample (AM/sbq/quantize.rs:115-148) restated in f32: compare with a straight numpy transcription."""
    O = oracle
    rng = np.random.default_rng(3)
    X = rng.random((200, 5), dtype=np.float32)
    mean, m2, cnt = O.train(X, 2)
    m = np.zeros(5, np.float32)
    s2 = np.zeros(5, np.float32)
    for i, s in enumerate(X):
        c = np.float32(i + 1)
        delta = (s - m).astype(np.float32)
        m = (m + ((s - m).astype(np.float32) / c).astype(np.float32)).astype(np.float32)
        delta2 = (s - m).astype(np.float32)
        s2 = (s2 + (delta * delta2).astype(np.float32)).astype(np.float32)
    assert cnt == 200 and mean.tobytes() == m.tobytes() and m2.tobytes() == s2.tobytes()


def test_preprocess_cosine(oracle):
    O = oracle
    z = np.zeros(16, np.float32)
    out, ch = O.preprocess_cosine(z)
    assert not ch and not out.any()                      # zero vector left alone (AM/distance/mod.rs:231-233)
    u = np.zeros(16, np.float32)
    u[3] = 1
    assert not O.preprocess_cosine(u)[1]                 # already unit (:235-237)
    v = np.arange(1, 17, dtype=np.float32)
    out, ch = O.preprocess_cosine(v)
    assert ch and abs(float(np.dot(out, out)) - 1) < 1e-5
    assert not O.preprocess_cosine(out)[1]               # idempotent (debug_assert :246-249)


def test_micro_bench_runs(oracle):
    """the timing loop over the reference's criterion bench inputs (benches/distance.rs:144-161,299-338) that bench.py reports"""
    m = oracle.micro_bench(iters=2000)
    assert set(m) == {"distance_l2_2000d", "distance_cosine_2000d", "inner_product_2000d", "distance_xor_optimized_1536bit"}
    assert all(0 < v < 1e6 for v in m.values())


# ---- known answers generated BY THE REFERENCE (oracle/ref_kat.rs, one `cargo test` for anyone with its toolchain) ----------
REF_KAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kat.txt")


def _kat_sample(i, d):
    return np.float32(np.float32((i * 31 + d * 17) % 97) / np.float32(97.0)) - np.float32(0.5)


def _heap_ops(seed, steps, nkeys):
    """the push / pop sequence of ref_kat_binary_heap_tie_order (a pop on an empty heap becomes a push there too)"""
    ops, size, x = [], 0, seed
    for i in range(steps):
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        if (x >> 16) % 3 != 0 or size == 0:
            ops.append(((x >> 8) % nkeys, i))
            size += 1
        else:
            ops.append((0xFFFFFFFF, 0))
            size -= 1
    return ops


def check_ref_kat_lines(oracle, lines):
    """every `KAT ...` line of oracle/ref_kat.rs against the oracle; returns the number of lines checked"""
    O = oracle
    from oracle import pages_py as PG
    trained = {}
    checked = 0
    for ln in lines:
        f = dict(kv.split("=", 1) for kv in ln.split()[2:])
        kind = ln.split()[1]
        if kind == "train":
            dims, bits, n = int(f["dims"]), int(f["bits"]), int(f["n"])
            rows = np.array([[_kat_sample(i, d) for d in range(dims)] for i in range(n)], np.float32)
            mean, m2, cnt = O.train(rows, bits)
            trained[(dims, bits, n)] = (mean, m2, cnt)
            assert cnt == int(f["count"])
            assert [f"{v:08x}" for v in mean.view(np.uint32)] == f["mean"].split(",")
            if bits > 1:
                assert [f"{v:08x}" for v in m2.view(np.uint32)] == f["m2"].split(",")
        elif kind == "code":
            dims, bits, n, j = int(f["dims"]), int(f["bits"]), int(f["n"]), int(f["q"])
            mean, m2, cnt = trained[(dims, bits, n)]
            v = np.array([[np.float32(_kat_sample(1000 + 7 * j, d) * np.float32(1.5)) for d in range(dims)]], np.float32)
            code = O.quantize(mean, m2, cnt, bits, v)[0]
            assert [f"{w:016x}" for w in code] == f["words"].split(",")
        elif kind == "heap":
            want = [int(x) for x in f["pops"].split(",")]
            assert O.heap_replay(_heap_ops(int(f["seed"]), int(f["steps"]), int(f["nkeys"]))) == want
        elif kind == "node":
            layout = (int(f["size"]), int(f["off_heap"]), int(f["off_code"]), int(f["off_nbrs"]), int(f["off_last"]))
            inv = (0xFFFFFFFF, 0)
            got = PG.rkyv_sbq_node((7, 3), [0x0123456789abcdef, 0xfedcba9876543210, 0x00000000ffffffff],
                                   [(1, 1), (2, 5), inv, inv], labels=[2, 5, 9] if f["kind"] == "labeled" else None, layout=layout)
            assert got.hex() == f["bytes"]
        elif kind == "simd":
            d = int(f["d"])
            a = np.array([np.float32(np.float32(i * 37 % 101) / np.float32(101.0)) - np.float32(0.3) for i in range(d)], np.float32)
            b = np.array([np.float32(np.float32(i * 53 % 103) / np.float32(103.0)) - np.float32(0.7) for i in range(d)], np.float32)
            assert f"{np.float32(O.distance_l2(a, b)).view(np.uint32):08x}" == f["l2"]
            assert f"{np.float32(O.inner_product(a, b)).view(np.uint32):08x}" == f["ip"]
        elif kind == "meta":
            # the archived MetaPage as the reference's rkyv really writes it: the pure-Python encoder must produce exactly these
            # bytes under the printed field offsets (the C decoder is held to the Python one in tests/test_pages.py)
            from oracle import pages_py as PG
            offs = [int(x) for x in f["offs"].split(",")]
            lay = dict(zip(PG.META_FIELDS, offs), root_size=int(f["size"]))
            base = dict(num_dimensions=768, num_dimensions_to_index=512, bq_num_bits_per_dimension=2, distance_type=1, storage_type=2,
                        num_neighbors=50, search_list_size=100, max_alpha=1.2, quantizer=(3, 1), layout=lay)
            cases = {
                "none_inline": dict(extension_version="0.8.0"),
                "some_empty_outofline": dict(extension_version="0.8.0-rc1+build.77", default_start=(7, 1)),
                "some_four_labels": dict(extension_version="0.8.0", default_start=(7, 1), has_labels=True,
                                         labeled_starts={l: (100 + (l & 0xFFFFFFFF) % 50, 1 + (l & 0xFFFFFFFF) % 7) for l in (5, -3, 300, 17)}),
                "some_thousand_labels": dict(extension_version="0.8.0", default_start=(7, 1), has_labels=True,
                                             labeled_starts={l: (l + 1000, 1 + (l + 500) % 90) for l in range(-500, 500)}),
            }
            want = bytes.fromhex(f["bytes"])
            assert PG.rkyv_meta_page(**base, **cases[f["case"]]) == want, f["case"]
            got = PG.parse_meta_page(want, lay)
            assert got["default_start"] == cases[f["case"]].get("default_start")
        else:
            raise AssertionError(f"unknown KAT line: {ln[:60]}")
        checked += 1
    return checked


def test_reference_generated_kats(oracle):
    if not os.path.exists(REF_KAT):
        pytest.skip("tests/golden/ref_kat.txt absent: generate it with the reference's toolchain (oracle/ref_kat.rs)")
    lines = [ln.strip() for ln in open(REF_KAT) if ln.startswith("KAT ")]
    assert check_ref_kat_lines(oracle, lines) == len(lines) > 0


def test_ref_kat_loader_on_oracle_made_lines(oracle):
    """the loader itself is exercised on lines in the generator's format produced from the oracle (self-consistency only: it
    proves the parsing and the closed-form inputs, not the reference)"""
    O = oracle
    from oracle import pages_py as PG
    lines = []
    for dims, bits, n in ((10, 2, 50), (70, 1, 33)):
        rows = np.array([[_kat_sample(i, d) for d in range(dims)] for i in range(n)], np.float32)
        mean, m2, cnt = O.train(rows, bits)
        lines.append(f"KAT train dims={dims} bits={bits} n={n} count={cnt} mean=" + ",".join(f"{v:08x}" for v in mean.view(np.uint32)) +
                     " m2=" + ",".join(f"{v:08x}" for v in (m2.view(np.uint32) if bits > 1 else [])))
        v = np.array([[np.float32(_kat_sample(1000, d) * np.float32(1.5)) for d in range(dims)]], np.float32)
        code = O.quantize(mean, m2, cnt, bits, v)[0]
        lines.append(f"KAT code dims={dims} bits={bits} n={n} q=0 words=" + ",".join(f"{w:016x}" for w in code))
    pops = O.heap_replay(_heap_ops(1, 400, 7))
    assert sorted(pops) == sorted(i for i, (kk, _) in enumerate(_heap_ops(1, 400, 7)) if kk != 0xFFFFFFFF)
    lines.append("KAT heap seed=1 steps=400 nkeys=7 pops=" + ",".join(str(x) for x in pops))
    inv = (0xFFFFFFFF, 0)
    b = PG.rkyv_sbq_node((7, 3), [0x0123456789abcdef, 0xfedcba9876543210, 0x00000000ffffffff], [(1, 1), (2, 5), inv, inv], labels=[2, 5, 9])
    lines.append(f"KAT node kind=labeled size=32 off_heap=0 off_code=8 off_nbrs=16 off_last=24 bytes={b.hex()}")
    lay = PG.DEFAULT_META_LAYOUT
    offs = ",".join(str(lay[k]) for k in PG.META_FIELDS)
    mb = PG.rkyv_meta_page(num_dimensions=768, num_dimensions_to_index=512, quantizer=(3, 1), extension_version="0.8.0-rc1+build.77",
                           default_start=(7, 1))
    lines.append(f"KAT meta case=some_empty_outofline size={lay['root_size']} offs={offs} bytes={mb.hex()}")
    assert check_ref_kat_lines(oracle, lines) == len(lines)
