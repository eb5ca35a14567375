"""Index relation pages -> flat arrays (vs_pages_*, host code of libvsgpu; no device needed).

The page writer of oracle/pages_py.py restates the reference's page layer; it is first pinned against the reference's
own KATs (`tape_resume`, UT/tape.rs:100-171; `test_chain_tape`, UT/chain.rs:217-294) and then used to manufacture
whole index relations that libvsgpu's reader must decode to exactly the arrays they were made from (also compared
with the independent pure-Python reader).
"""
import struct

import numpy as np
import pytest

from oracle import pages_py as PG


def _reader(**kw):
    from pgvectorscale_amd.pages import IndexPages
    return IndexPages(**kw)


def random_index(n, W, R, seed, n_labels=0, deleted_frac=0.1, full_rows=False):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 1 << 63, (n, W), dtype=np.uint64) | (rng.integers(0, 2, (n, W), dtype=np.uint64) << np.uint64(63))
    nbrs = np.full((n, R), 0xFFFFFFFF, np.uint32)
    for i in range(n):
        k = R if full_rows else int(rng.integers(0, R + 1))
        k = min(k, n - 1)
        cand = rng.choice(n - 1, size=k, replace=False)
        cand[cand >= i] += 1  # no self loops
        nbrs[i, :k] = cand
    tids = ((rng.integers(0, 1 << 31, n, dtype=np.uint64)) << np.uint64(16)) | rng.integers(1, 200, n, dtype=np.uint64)
    tids[rng.random(n) < deleted_frac] &= ~np.uint64(0xFFFF)
    label_off = label_val = None
    if n_labels:
        label_off = np.zeros(n + 1, np.uint32)
        vals = []
        for i in range(n):
            ls = sorted(set(int(v) for v in rng.integers(-5, n_labels, int(rng.integers(0, 5)))))
            vals.extend(ls)
            label_off[i + 1] = len(vals)
        label_val = np.array(vals, np.int16)
    D = W * 32
    mean = rng.standard_normal(D).astype(np.float32)
    m2 = rng.random(D).astype(np.float32)
    return dict(codes=codes, nbrs=nbrs, heap_tids=tids, mean=mean, m2=m2, count=n, label_off=label_off, label_val=label_val)


def decode(w, chunks=None, threads=0, layout=None):
    rd = _reader(has_labels=w.has_labels, threads=threads, layout=layout)
    data = w.rel.tobytes()
    nb = len(w.rel.pages)
    if chunks is None:
        rd.add(data)
    else:
        b = 0
        while b < nb:
            e = min(nb, b + chunks)
            rd.add(data[b * PG.BLCKSZ:e * PG.BLCKSZ], first_block=b)
            b = e
    info = rd.finish()
    return rd, info, rd.arrays()


def check_equal(src, arr, has_labels):
    assert (arr["codes"] == src["codes"]).all()
    assert (arr["heap_tids"] == src["heap_tids"]).all()
    assert (arr["nbrs"] == src["nbrs"]).all()
    if has_labels:
        assert (arr["label_off"] == src["label_off"]).all()
        assert (arr["label_val"] == src["label_val"]).all()


# ---- the reference's own KATs for the page layer ----------------------------------------------------------------------
def test_tape_resume_kat():
    """UT/tape.rs:100-171"""
    rel = PG.Relation()
    PG.ChainTapeWriter(rel, PG.PT_META)  # block 0 of every index relation
    tape = PG.Tape(rel, PG.PT_NODE)
    node_page = tape.current
    assert tape.write(bytes([1, 2, 3])) == (node_page, 1)
    ip = tape.write(bytes([4, 5, 6]))
    assert ip[0] == node_page and tape.current == node_page

    tape = PG.Tape.resume(rel, PG.PT_PQ_VEC)
    ip = tape.write(bytes([99]))
    assert ip[0] == tape.current == node_page + 1, "An unseen page type must create a new page"

    tape = PG.Tape.resume(rel, PG.PT_NODE)
    ip = tape.write(bytes([7, 8, 9]))
    assert ip[0] == tape.current
    tape.write(bytes([10, 11, 12]))
    assert tape.current == node_page, "Data should be written to existing page when there is room"
    assert rel.aligned_free_space(tape.current) == 8104  # the one number the reference pins for the page arithmetic

    tape = PG.Tape.resume(rel, PG.PT_NODE)
    ip = tape.write(bytes([42]) * 8109)
    assert ip[0] == tape.current and tape.current != node_page, "Writing more than available forces a new page"


def test_chain_tape_kat():
    """UT/chain.rs:217-294: small items, then sizes around 1x / 2x / 3x BLCKSZ, read back through the pure-Python
    iterator AND through libvsgpu's vs_pages_read_chain."""
    rel = PG.Relation()
    meta = PG.ChainTapeWriter(rel, PG.PT_META)
    assert meta.write(PG.rkyv_meta_header()) == (0, 1) and meta.write(b"\0" * 100) == (0, 2)  # AM/meta_page.rs:357-364
    expected = []
    tape = PG.ChainTapeWriter(rel, PG.PT_SBQ_MEANS)
    for i in range(100):
        data = f"hello world {i}".encode()
        ip = tape.write(data)
        assert PG.read_chain(rel, *ip, PG.PT_SBQ_MEANS) == data
        expected.append((ip, data))
    for mult in (1, 2, 3):
        for data_size in range(mult * PG.BLCKSZ - 100, mult * PG.BLCKSZ + 100, 1 if mult == 1 else 7):
            big = bytes(i % 256 for i in range(data_size))
            tape = PG.ChainTapeWriter(rel, PG.PT_SBQ_MEANS)
            for _ in range(10 if mult == 1 else 3):
                ip = tape.write(big)
                expected.append((ip, big))
    for ip, data in expected[100::17]:
        assert PG.read_chain(rel, *ip, PG.PT_SBQ_MEANS) == data
    rd = _reader()
    rd.add(rel.tobytes())
    info = rd.finish()
    assert info.n_nodes == 0 and info.pages_by_type[PG.PT_SBQ_MEANS] == len(rel.pages) - 1
    for ip, data in expected:
        assert rd.read_chain(ip[0], ip[1], PG.PT_SBQ_MEANS) == data
    # a chain is typed: reading it as the other chained page type is the reference's assert (UT/chain.rs:170)
    from pgvectorscale_amd import VsError
    with pytest.raises(VsError, match="page type"):
        rd.read_chain(expected[0][0][0], expected[0][0][1], PG.PT_META)
    rd.close()


# ---- node items ----------------------------------------------------------------------------------------------------------
def test_node_item_geometry_matches_the_survey():
    """W=24, R=50: 192 + 400 + 32 = 624-byte items, 12 per 8 KB page"""
    src = random_index(40, 24, 50, seed=1, full_rows=True)
    w = PG.write_index(**src)
    blk, off = w.node_ptrs[0]
    assert w.rel.item_span(blk, off)[1] == 624
    per_page = [w.rel.max_offset(b) for b in range(len(w.rel.pages)) if w.rel.page_type(b) == PG.PT_SBQ_NODE]
    assert per_page == [12, 12, 12, 4]
    rd, info, arr = decode(w)
    assert (info.n_nodes, info.words, info.num_neighbors) == (40, 24, 50)
    check_equal(src, arr, False)
    rd.close()


@pytest.mark.parametrize("n,W,R,n_labels", [(700, 4, 32, 0), (333, 24, 50, 0), (500, 2, 20, 12), (64, 250, 7, 3), (1, 1, 1, 0)])
def test_relation_round_trip(n, W, R, n_labels):
    src = random_index(n, W, R, seed=n + W, n_labels=n_labels)
    w = PG.write_index(**src, means_first=(n % 2 == 0), zero_page_every=97 if n > 100 else 0)
    rd, info, arr = decode(w)
    assert (info.n_nodes, info.words, info.num_neighbors, info.has_labels) == (n, W, R, int(n_labels > 0))
    assert info.n_blocks == len(w.rel.pages)
    assert info.meta_magic == PG.TSV_MAGIC_NUMBER and info.meta_version == PG.TSV_VERSION
    assert info.n_deleted == int((src["heap_tids"] & np.uint64(0xFFFF) == 0).sum())
    assert info.pages_by_type[PG.PT_META] == 1
    assert info.new_pages == sum(1 for p in w.rel.pages if struct.unpack_from("<H", p, 14)[0] == 0)
    check_equal(src, arr, n_labels > 0)
    # the independent pure-Python reader sees the same index
    codes, nbrs, tids, labs = PG.read_index(w)
    assert (np.array(codes) == arr["codes"]).all() and (np.array(tids, np.uint64) == arr["heap_tids"]).all()
    for i in range(n):
        row = arr["nbrs"][i]
        k = int((row != 0xFFFFFFFF).sum())
        assert list(row[:k]) == nbrs[i] and (row[k:] == 0xFFFFFFFF).all()
        if n_labels:
            assert list(arr["label_val"][arr["label_off"][i]:arr["label_off"][i + 1]]) == list(labs[i])
    # IndexPointer <-> node id
    for i in (0, n // 2, n - 1):
        assert rd.node_of(*w.node_ptrs[i]) == i
        assert rd.item_pointer_of(i) == w.node_ptrs[i]
    # SbqMeans through the chain (W=250 -> 8000 dims -> 64 KB, nine pages)
    cnt, mean, m2 = rd.sbq_means(*w.means_ptr)
    assert cnt == src["count"] and (mean == src["mean"]).all() and (m2 == src["m2"]).all()
    if W == 250:
        assert info.pages_by_type[PG.PT_SBQ_MEANS] >= 8
    rd.close()


def test_chunked_and_threaded_adds_agree():
    src = random_index(900, 6, 24, seed=5, n_labels=9)
    w = PG.write_index(**src)
    ref = decode(w, threads=1)
    for chunks, threads in ((1, 1), (3, 8), (64, 4)):
        rd, info, arr = decode(w, chunks=chunks, threads=threads)
        for k in arr:
            assert (arr[k] == ref[2][k]).all(), (k, chunks, threads)
        rd.close()
    ref[0].close()


def test_custom_node_layout():
    """The archived node's field order is a parameter (rkyv 0.7 archives are repr(Rust)): a permuted layout written by
    the page writer is decoded with the matching vs_node_layout, and mis-decoded or rejected without it."""
    from pgvectorscale_amd import VsError
    src = random_index(200, 3, 10, seed=11, n_labels=6)
    layout = (40, 32, 16, 0, 8)  # root size 40: neighbors | labels | bq_vector | (pad) | heap pointer
    w = PG.write_index(**src, layout=layout)
    rd, info, arr = decode(w, layout=layout)
    check_equal(src, arr, True)
    rd.close()
    with pytest.raises(VsError):
        decode(w)
    from pgvectorscale_amd.pages import IndexPages
    assert IndexPages.default_layout(True) == (32, 0, 8, 16, 24)
    assert IndexPages.default_layout(False) == (32, 0, 8, 16, 0xFFFFFFFF)


def test_neighbor_list_ends_at_first_invalid_pointer():
    """ArchivedSbqNode::num_neighbors: slots after the first InvalidBlockNumber are ignored even if they hold bytes
    (AM/sbq/node.rs:260-285)"""
    src = random_index(50, 2, 8, seed=3, full_rows=True)
    w = PG.write_index(**src)
    blk, off = w.node_ptrs[7]
    s, l = w.rel.item_span(blk, off)
    fld = s + l - 32 + 16
    rel_off, cnt = struct.unpack_from("<iI", w.rel.pages[blk], fld)
    at = fld + rel_off
    w.rel.pages[blk][at + 8 * 3:at + 8 * 3 + 8] = PG.rkyv_item_pointer(PG.INVALID_BLOCK, 0)  # slot 3 ends the list
    rd, info, arr = decode(w)
    assert (arr["nbrs"][7, :3] == src["nbrs"][7, :3]).all() and (arr["nbrs"][7, 3:] == 0xFFFFFFFF).all()
    assert (arr["nbrs"][8] == src["nbrs"][8]).all()
    rd.close()


# ---- malformed input ------------------------------------------------------------------------------------------------------
def _expect_error(rel_bytes, match, has_labels=False, finish=True):
    from pgvectorscale_amd import VsError
    rd = _reader(has_labels=has_labels)
    with pytest.raises(VsError, match=match):
        rd.add(rel_bytes)
        if finish:
            rd.finish()
    rd.close()


def test_malformed_pages_are_rejected_with_a_reason():
    from pgvectorscale_amd import VsError
    src = random_index(60, 2, 6, seed=2, n_labels=4)
    w = PG.write_index(**src)
    good = bytearray(w.rel.tobytes())
    node_blk, node_off = w.node_ptrs[20]
    base = node_blk * PG.BLCKSZ

    bad = bytearray(good)
    struct.pack_into("<H", bad, base + PG.BLCKSZ - 8 + 2, 0x1234)  # page_id magic (UT/page.rs:101-103)
    _expect_error(bytes(bad), "not the diskann magic", True)

    bad = bytearray(good)
    bad[base + PG.BLCKSZ - 8] = 9  # PageType::from_u8 panics on unknown numbers (UT/page.rs:44-56)
    _expect_error(bytes(bad), "Unknown PageType|unknown PageType", True)

    bad = bytearray(good)
    struct.pack_into("<H", bad, base + 18, 4096 | 4)
    _expect_error(bytes(bad), "pd_pagesize_version", True)

    bad = bytearray(good)
    struct.pack_into("<H", bad, base + 14, 20)  # pd_upper < pd_lower
    _expect_error(bytes(bad), "inconsistent page header", True)

    # a neighbor pointing at an item that does not exist
    bad = bytearray(good)
    s, l = w.rel.item_span(node_blk, node_off)
    fld = base + s + l - 32 + 16
    rel_off, cnt = struct.unpack_from("<iI", bad, fld)
    bad[fld + rel_off:fld + rel_off + 8] = PG.rkyv_item_pointer(node_blk, 200)
    _expect_error(bytes(bad), "not an SbqNode item", True)
    # ... or at the meta page
    bad[fld + rel_off:fld + rel_off + 8] = PG.rkyv_item_pointer(0, 1)
    _expect_error(bytes(bad), "not an SbqNode item", True)

    # an ArchivedVec that leaves the item
    bad = bytearray(good)
    struct.pack_into("<i", bad, base + s + l - 32 + 8, -100000)
    _expect_error(bytes(bad), "points outside the item", True)

    # a code of another width
    bad = bytearray(good)
    struct.pack_into("<I", bad, base + s + l - 32 + 8 + 4, 1)
    _expect_error(bytes(bad), "code width", True)

    # an unsorted label set
    lab_node = next(i for i in range(60) if src["label_off"][i + 1] - src["label_off"][i] >= 2)
    lb, lo = w.node_ptrs[lab_node]
    s2, l2 = w.rel.item_span(lb, lo)
    fld2 = lb * PG.BLCKSZ + s2 + l2 - 32 + 24
    r2, c2 = struct.unpack_from("<iI", good, fld2)
    bad = bytearray(good)
    a, b = struct.unpack_from("<hh", bad, fld2 + r2)
    struct.pack_into("<hh", bad, fld2 + r2, b, a)
    _expect_error(bytes(bad), "strictly increasing", True)

    # a dead line pointer on a node page
    bad = bytearray(good)
    lp = struct.unpack_from("<I", bad, base + 24 + 4 * (node_off - 1))[0]
    struct.pack_into("<I", bad, base + 24 + 4 * (node_off - 1), (lp & ~(3 << 15)) | (3 << 15))
    _expect_error(bytes(bad), "LP_NORMAL", True)

    # wrong meta magic
    bad = bytearray(good)
    s0, l0 = w.rel.item_span(0, 1)
    struct.pack_into("<I", bad, s0 + 8, 1)
    _expect_error(bytes(bad), "magic", True)

    # call-sequence errors; a failed add leaves the reader usable
    rd = _reader(has_labels=True)
    with pytest.raises(VsError, match="in order"):
        rd.add(bytes(good[:PG.BLCKSZ]), first_block=3)
    bad = bytearray(good)
    bad[base + PG.BLCKSZ - 8] = 9
    with pytest.raises(VsError):
        rd.add(bytes(bad))
    rd.n_blocks = 0
    rd.add(bytes(good))
    info = rd.finish()
    assert info.n_nodes == 60
    check_equal(src, rd.arrays(), True)
    with pytest.raises(VsError, match="after vs_pages_finish"):
        rd.add(bytes(good[:PG.BLCKSZ]), first_block=info.n_blocks)
    with pytest.raises(VsError, match="not an SbqNode item"):
        rd.node_of(0, 1)
    rd.close()

    with pytest.raises(ValueError):
        _reader().add(b"\0" * 100)
    with pytest.raises(VsError, match="block size"):
        _reader(page_size=1000)


def test_plain_storage_relations_are_refused():
    rel = PG.Relation()
    meta = PG.ChainTapeWriter(rel, PG.PT_META)
    meta.write(PG.rkyv_meta_header())
    meta.write(b"\0" * 64)
    PG.Tape(rel, PG.PT_NODE).write(b"\1" * 300)
    _expect_error(rel.tobytes(), "plain")


def test_empty_relation():
    rel = PG.Relation()
    meta = PG.ChainTapeWriter(rel, PG.PT_META)
    meta.write(PG.rkyv_meta_header())
    meta.write(b"\0" * 64)
    rd = _reader()
    rd.add(rel.tobytes())
    info = rd.finish()
    assert info.n_nodes == 0 and info.n_blocks == 1 and info.meta_version == 3
    assert rd.arrays()["codes"].size == 0
    rd.close()


# ---- the MetaPage body (AM/meta_page.rs:176-210) ----------------------------------------------------------------------
def test_meta_page_round_trip_through_both_decoders():
    """rkyv archive of MetaPage (String inline / out of line, Option<StartNodes> None / Some, BTreeMap of 0 .. 65536 labeled start
    nodes = one leaf .. three levels of nodes): libvsgpu's decoder and the pure-Python one must both give back what was written"""
    from pgvectorscale_amd.pages import decode_meta_page
    cases = [
        dict(num_dimensions=768),
        dict(num_dimensions=1536, num_dimensions_to_index=512, bq_num_bits_per_dimension=1, distance_type=0, storage_type=2,
             num_neighbors=64, search_list_size=77, max_alpha=1.35, default_start=(3, 9), quantizer=(1, 1)),
        dict(num_dimensions=128, extension_version="0.8.0-rc1+build.77", default_start=(7, 1), labeled_starts={5: (9, 2)},
             has_labels=True),
        dict(num_dimensions=64, storage_type=0, default_start=(2, 2), has_labels=True,
             labeled_starts={int(l): (1000 + i, 1 + i % 90) for i, l in enumerate(range(-32768, 32768, 97))}),
        dict(num_dimensions=64, default_start=(2, 2), has_labels=True,
             labeled_starts={l: (40000 + l, 1 + (l % 7)) for l in range(-32768, 32768)}),
    ]
    for kw in cases:
        data = PG.rkyv_meta_page(**kw)
        want = PG.parse_meta_page(data)
        got, starts = decode_meta_page(data)
        assert got["magic_number"] == PG.TSV_MAGIC_NUMBER and got["version"] == PG.TSV_VERSION
        for key in ("distance_type", "num_dimensions", "num_dimensions_to_index", "bq_num_bits_per_dimension", "storage_type",
                    "num_neighbors", "search_list_size", "max_alpha"):
            assert got[key] == want[key], key
        assert got["extension_version_when_built"] == want["extension_version_when_built"][:63]
        assert bool(got["has_labels"]) == want["has_labels"] == bool(kw.get("has_labels"))
        assert (got["quantizer_block"], got["quantizer_offset"]) == want["quantizer_metadata"]
        if want["default_start"] is None:
            assert not got["has_start_nodes"] and starts == {}
        else:
            assert (got["default_start_block"], got["default_start_offset"]) == want["default_start"]
            assert starts == want["labeled_starts"] == (kw.get("labeled_starts") or {})
            assert list(starts) == sorted(starts)


def test_meta_page_with_another_field_order():
    """rkyv 0.7 archives are repr(Rust): the field offsets are a parameter of both decoders"""
    from pgvectorscale_amd.pages import decode_meta_page
    order = ("max_alpha", "start_nodes", "quantizer_metadata", "extension_version_when_built", "magic_number", "version",
             "num_dimensions", "num_dimensions_to_index", "num_neighbors", "search_list_size", "distance_type",
             "bq_num_bits_per_dimension", "storage_type", "has_labels")  # largest alignment first, as rustc tends to do
    lay = PG.meta_layout(order)
    assert lay["root_size"] == 80 and lay["magic_number"] == 44 and lay != PG.DEFAULT_META_LAYOUT
    data = PG.rkyv_meta_page(num_dimensions=100, default_start=(4, 4), labeled_starts={1: (5, 5), 2: (6, 6)}, layout=lay,
                             has_labels=True, max_alpha=1.0625)
    got, starts = decode_meta_page(data, layout=lay)
    assert got["num_dimensions"] == 100 and got["max_alpha"] == 1.0625 and starts == {1: (5, 5), 2: (6, 6)}
    from pgvectorscale_amd import VsError
    with pytest.raises(VsError):  # the default layout on these bytes: the magic number is not where it is looked for
        decode_meta_page(data)


def test_malformed_meta_pages_are_rejected():
    from pgvectorscale_amd import VsError
    from pgvectorscale_amd.pages import decode_meta_page
    good = bytearray(PG.rkyv_meta_page(num_dimensions=64, default_start=(2, 2), has_labels=True, extension_version="0.8.0-long-version",
                                       labeled_starts={l: (9, 1) for l in range(600)}))
    lay = PG.DEFAULT_META_LAYOUT
    root = len(good) - lay["root_size"]
    sn = root + lay["start_nodes"]

    def broken(mut):
        b = bytearray(good)
        mut(b)
        with pytest.raises(VsError):
            decode_meta_page(bytes(b))

    decode_meta_page(bytes(good))
    with pytest.raises(VsError):
        decode_meta_page(bytes(good[-40:]))                                       # shorter than the root object
    broken(lambda b: struct.pack_into("<I", b, root, 12345))                       # magic
    broken(lambda b: b.__setitem__(sn, 7))                                         # Option tag
    broken(lambda b: struct.pack_into("<i", b, sn + 16, -(1 << 30)))               # B-tree root outside the archive
    broken(lambda b: struct.pack_into("<I", b, sn + 12, 599))                      # length field != entries
    broken(lambda b: struct.pack_into("<I", b, sn + 12, 70000))                    # more labels than smallints
    broken(lambda b: struct.pack_into("<i", b, root + lay["extension_version_when_built"] + 4, -(1 << 29)))  # string target
    # a cycle: the root's first child pointer led back to the root itself
    rootnode = sn + 16 + struct.unpack_from("<i", good, sn + 16)[0]
    assert struct.unpack_from("<H", good, rootnode)[0] & 0x8000
    broken(lambda b: struct.pack_into("<i", b, rootnode + 8, -8))
    # unsorted keys
    leaf0 = rootnode + 8 + struct.unpack_from("<i", good, rootnode + 8)[0]
    broken(lambda b: struct.pack_into("<h", b, leaf0 + 12, 500))


@pytest.mark.parametrize("n_labels,big_map", [(0, False), (6, False), (6, True)])
def test_index_desc_from_the_relation_alone(n_labels, big_map):
    """MetaPage::fetch over a manufactured relation: geometry, default start node, labeled start nodes (IndexPointers translated to
    node ids) and the quantizer pointer all come from the pages — vs_pages_meta yields a complete vs_index_desc without Rust"""
    n, W, R = 700, 3, 9
    ix = random_index(n, W, R, seed=12, n_labels=n_labels)
    rng = np.random.default_rng(1)
    starts = {}
    if n_labels:
        labs = range(-5, n_labels) if not big_map else range(-2000, 2000)  # 4000 entries: the meta chain leaves block 0
        starts = {int(l): int(rng.integers(0, n)) for l in labs}
    meta = dict(num_dimensions=W * 32, bq_num_bits_per_dimension=2, distance_type=1, num_neighbors=R, search_list_size=100,
                max_alpha=1.2, default_start=17, labeled_starts=starts)
    w = PG.write_index(**ix, meta=meta, zero_page_every=97)
    rd = _reader(has_labels=bool(n_labels))
    rd.add(w.rel.tobytes())
    rd.finish()
    m, d, st = rd.meta()
    assert (d.n, d.dim_full, d.dim_index, d.bits, d.words, d.num_neighbors, d.distance_type) == (n, W * 32, W * 32, 2, W, R, 1)
    assert d.has_labels == int(bool(n_labels)) and d.storage_type == 0 and d.default_start == 17 and d.n_label_starts == len(starts)
    assert st == starts
    assert (m["quantizer_block"], m["quantizer_offset"]) == w.means_ptr
    cnt, mean, m2 = rd.sbq_means(m["quantizer_block"], m["quantizer_offset"])
    assert cnt == n and (mean == ix["mean"]).all() and (m2 == ix["m2"]).all()
    if big_map:
        assert len(PG.read_chain(w.rel, 0, 2, PG.PT_META)) > PG.BLCKSZ
    # an index that never saw a row: start_nodes is None -> VS_INVALID_NODE
    empty = PG.write_index(codes=np.zeros((0, W), np.uint64), nbrs=np.zeros((0, R), np.uint32), heap_tids=np.zeros(0, np.uint64),
                           mean=ix["mean"], m2=ix["m2"], count=0, num_neighbors=R,
                           meta=dict(num_dimensions=W * 32, num_neighbors=R))
    rd2 = _reader()
    rd2.add(empty.rel.tobytes())
    rd2.finish()
    m2_, d2, st2 = rd2.meta()
    assert d2.n == 0 and d2.default_start == 0xFFFFFFFF and st2 == {} and not m2_["has_start_nodes"]


# ---- `plain` storage relations (PageType::Node pages of PlainNode items, AM/plain/node.rs:15-22) -------------------------------
def test_plain_storage_relation_round_trip():
    rng = np.random.default_rng(3)
    n, D, R = 900, 96, 12
    vecs = rng.standard_normal((n, D)).astype(np.float32)
    ix = random_index(n, 2, R, seed=8)
    meta = dict(num_dimensions=D, storage_type=0, bq_num_bits_per_dimension=1, distance_type=0, num_neighbors=R, default_start=5)
    w = PG.write_plain_index(vectors=vecs, nbrs=ix["nbrs"], heap_tids=ix["heap_tids"], meta=meta)
    rd = _reader(plain=True, threads=3)
    data = w.rel.tobytes()
    half = (len(w.rel.pages) // 2) * PG.BLCKSZ
    rd.add(data[:half])
    rd.add(data[half:])
    info = rd.finish()
    assert (info.n_nodes, info.words, info.num_neighbors, info.has_labels) == (n, D, R, 0)
    assert info.pages_by_type[PG.PT_NODE] > 0 and info.pages_by_type[PG.PT_SBQ_NODE] == 0
    a = rd.arrays()
    assert a["vecs"].tobytes() == vecs.tobytes() and (a["nbrs"] == ix["nbrs"]).all() and (a["heap_tids"] == ix["heap_tids"]).all()
    m, d, st = rd.meta()
    assert (d.n, d.dim_full, d.dim_index, d.num_neighbors, d.distance_type, d.storage_type, d.default_start) == (n, D, D, R, 0, 1, 5)
    # the wrong reader for the relation is told so
    from pgvectorscale_amd import VsError
    sbq = _reader()
    sbq.add(data)
    with pytest.raises(VsError, match="plain"):
        sbq.finish()
    # another field order of the archived PlainNode
    lay = (32, 0, 8, 24, None)  # heap_item_pointer, vector, (pq_vector at 16), neighbor_index_pointers
    w2 = PG.write_plain_index(vectors=vecs[:50], nbrs=np.minimum(ix["nbrs"][:50], 49) | np.where(ix["nbrs"][:50] == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), np.uint32(0)),
                              heap_tids=ix["heap_tids"][:50], layout=lay)
    rd2 = _reader(plain=True, layout=(32, 0, 8, 24, 0xFFFFFFFF))
    rd2.add(w2.rel.tobytes())
    rd2.finish()
    assert rd2.arrays()["vecs"].tobytes() == vecs[:50].tobytes()
