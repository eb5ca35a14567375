"""The heap's vector column, staged from table pages (vs_heap_*, host code of libvsgpu; no device needed).

oracle/heap_py.py restates PostgreSQL's heap / TOAST page and tuple layout and manufactures the table a `diskann` index points into:
rows with several columns before the vector (fixed-length with alignment padding, NULLs, short and long varlenas), vectors in
line (short 1-byte header, 4-byte header) and out of line (TOAST chunks, 1 996 bytes each), pruned HOT chains, dead line pointers,
deleted index tuples.  libvsgpu's reader must hand back, for every index node, exactly the vector `heap_getattr` + detoasting
would — compared with the source arrays and with the independent pure-Python reader."""
import struct

import numpy as np
import pytest

from oracle import heap_py as HP

# id bigint, label smallint, tags smallint[] (varlena), note text (varlena), embedding vector, extra int4
ATTRS = [(8, "d"), (2, "s"), (-1, "i"), (-1, "i"), (-1, "i"), (4, "i")]
VEC_ATT = 5  # 1-based attnum of `embedding`


def make_table(n, dim, seed, null_frac=0.05):
    rng = np.random.default_rng(seed)
    t = HP.Table(ATTRS)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    tids, expect = [], []
    for i in range(n):
        tags = None if rng.random() < 0.3 else ("inline", struct.pack("<iiiii", 1, 0, 21, 3, 1) + rng.bytes(int(rng.integers(0, 9)) * 2))
        note = None if rng.random() < 0.2 else ("inline", rng.bytes(int(rng.choice([0, 3, 60, 126, 127, 200, 900]))))
        vec = None if rng.random() < null_frac else ("inline", HP.vector_datum_body(vecs[i]))
        label = None if rng.random() < 0.1 else struct.pack("<h", int(rng.integers(-5, 100)))
        natts = len(ATTRS) if rng.random() < 0.8 else int(rng.integers(VEC_ATT, len(ATTRS) + 1))  # rows older than ADD COLUMN extra
        vals = [struct.pack("<q", i), label, tags, note, vec, struct.pack("<i", i)]
        tids.append(t.insert(vals, natts=natts))
        expect.append(None if vec is None else vecs[i])
    return t, np.array(tids, np.uint64), expect


def read_with_lib(t, tids, dim, chunk_blocks=None):
    from pgvectorscale_amd.pages import HeapColumn
    hc = HeapColumn(ATTRS, VEC_ATT, dim, tids)
    heap, toast = t.heap.tobytes(), t.toast.tobytes()
    step = (chunk_blocks or max(1, len(t.heap.pages))) * HP.BLCKSZ
    for at in range(0, len(heap), step):
        hc.add(heap[at:at + step])
    step = (chunk_blocks or max(1, len(t.toast.pages))) * HP.BLCKSZ
    for at in range(0, len(toast), step):
        hc.toast_add(toast[at:at + step])
    info, found = hc.finish()
    vecs = hc.vecs.copy()
    hc.close()
    return vecs, info, found


@pytest.mark.parametrize("dim,chunk", [(3, None), (30, 1), (31, None), (128, 3), (768, 2), (1536, None)])
def test_vector_column_round_trip(dim, chunk):
    """dim 3 / 30: 1-byte varlena header (<= 127 bytes, unaligned); 31 / 128: 4-byte header in line; 768 / 1536: 3 / 6 KB, moved to the
    TOAST relation in two / four chunks — every form heap_fill_tuple and the toaster produce for pgvector's `external` storage"""
    n = 400
    t, tids, expect = make_table(n, dim, seed=dim)
    # index order is not heap order: shuffle, and mark a few index tuples deleted (vacuum: offset 0)
    rng = np.random.default_rng(1)
    order = rng.permutation(n)
    tids, expect = tids[order], [expect[i] for i in order]
    dele = rng.random(n) < 0.05
    tids_idx = tids.copy()
    tids_idx[dele] &= ~np.uint64(0xFFFF)
    vecs, info, found = read_with_lib(t, tids_idx, dim, chunk)
    ref = HP.read_vector_column(t, tids, VEC_ATT - 1, dim)
    n_null = 0
    for i in range(n):
        if dele[i]:
            assert not found[i] and not vecs[i].any()
            continue
        assert (ref[i] is None) == (expect[i] is None)
        if expect[i] is None:
            n_null += 1
            assert not found[i] and not vecs[i].any()
        else:
            assert found[i] and vecs[i].tobytes() == expect[i].tobytes() == ref[i].tobytes(), i
    assert info["n_deleted"] == int(dele.sum()) and info["n_null"] == n_null
    assert info["n_inline"] + info["n_external"] == int(found.sum())
    assert (info["n_external"] > 0) == (dim >= 768) and (info["n_inline"] > 0) == (dim < 768)
    assert info["n_dead_line_pointer"] == info["n_not_found"] == info["n_toast_incomplete"] == 0
    if dim >= 768:
        assert info["n_chunks"] == info["n_external"] * -(-(4 + 4 * dim) // HP.toast_max_chunk_size())


def test_redirects_dead_pointers_and_missing_blocks():
    dim = 16
    t, tids, expect = make_table(60, dim, seed=7, null_frac=0.0)
    heap = t.heap
    # a pruned HOT chain: the root line pointer of row 5 redirects to a later version of the row on the same page (same vector)
    blk, off = int(tids[5]) >> 16, int(tids[5]) & 0xFFFF
    vals = [struct.pack("<q", 5), None, None, None, ("inline", HP.vector_datum_body(expect[5])), struct.pack("<i", 5)]
    nb, no = heap.add_item(HP.form_tuple(ATTRS, vals))
    if nb == blk:
        heap.set_line_pointer(blk, off, HP.LP_REDIRECT, lp_off=no)
    # rows 6 and 7: vacuumed away under the index's feet
    for i, fl in ((6, HP.LP_DEAD), (7, HP.LP_UNUSED)):
        heap.set_line_pointer(int(tids[i]) >> 16, int(tids[i]) & 0xFFFF, fl)
    # a TID beyond the relation
    tids = np.append(tids, np.uint64((9999 << 16) | 1))
    vecs, info, found = read_with_lib(t, tids, dim)
    assert found[5] and vecs[5].tobytes() == expect[5].tobytes()
    assert not found[6] and not found[7] and not found[-1]
    assert info["n_dead_line_pointer"] == 2 and info["n_not_found"] == 1
    ok = [i for i in range(60) if i not in (6, 7)]
    assert all(found[i] and vecs[i].tobytes() == expect[i].tobytes() for i in ok)


def test_compressed_vectors_are_decoded(monkeypatch):
    """a column whose storage was ALTERed to `extended`: pglz-compressed in line (VARATT_IS_4B_C) and compressed + external"""
    dim = 600
    v = np.zeros(dim, np.float32)
    v[::7] = 1.5  # compressible
    body = HP.vector_datum_body(v)
    comp = pglz_compress(body)
    assert len(comp) < len(body) // 4
    t = HP.Table(ATTRS)
    inline_c = ("raw", struct.pack("<II", ((len(comp) + 8) << 2) | 2, len(body)) + comp)  # 4-byte header (compressed), tcinfo, data
    tid_a = t.insert([struct.pack("<q", 1), None, None, None, inline_c, struct.pack("<i", 1)])
    # external + compressed: the TOAST relation holds tcinfo + data; va_rawsize counts the raw datum, va_extinfo the stored bytes
    stored = struct.pack("<I", len(body)) + comp
    vid = t.next_value
    t.next_value += 1
    for seq, at in enumerate(range(0, len(stored), t.chunk)):
        t.toast.add_item(HP.form_tuple(HP.TOAST_ATTRS, [struct.pack("<I", vid), struct.pack("<i", seq), ("inline", stored[at:at + t.chunk])]))
    tid_b = t.insert([struct.pack("<q", 2), None, None, None, ("external", len(body) + 4, len(stored), vid, t.toast_relid), struct.pack("<i", 2)])
    vecs, info, found = read_with_lib(t, np.array([tid_a, tid_b], np.uint64), dim)
    assert found.all() and vecs[0].tobytes() == v.tobytes() == vecs[1].tobytes()
    assert info["n_inline"] == 1 and info["n_external"] == 1


def pglz_compress(data):
    """a plain greedy pglz encoder (test infrastructure): control byte per 8 items; match = 2 bytes (+1 for lengths >= 18)"""
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        ctrl_at = len(out)
        out.append(0)
        for bit in range(8):
            if i >= n:
                break
            best_len, best_off = 0, 0
            for off in range(1, min(i, 4095) + 1):
                l = 0
                while l < 273 and i + l < n and data[i + l - off] == data[i + l]:
                    l += 1
                if l > best_len:
                    best_len, best_off = l, off
                    if l == 273:
                        break
            if best_len >= 3:
                out[ctrl_at] |= 1 << bit
                if best_len >= 18:
                    out += bytes([((best_off >> 4) & 0xF0) | 0x0F, best_off & 0xFF, best_len - 18])
                else:
                    out += bytes([((best_off >> 4) & 0xF0) | (best_len - 3), best_off & 0xFF])
                i += best_len
            else:
                out.append(data[i])
                i += 1
    return bytes(out)


def test_malformed_heap_pages_are_rejected():
    from pgvectorscale_amd import VsError
    dim = 64
    t, tids, expect = make_table(40, dim, seed=3, null_frac=0.0)
    good = bytearray(t.heap.tobytes())
    blk, off = int(tids[0]) >> 16, int(tids[0]) & 0xFFFF
    lp_at = blk * HP.BLCKSZ + HP.SIZE_OF_PAGE_HEADER + 4 * (off - 1)
    lp = struct.unpack_from("<I", good, lp_at)[0]
    tup = blk * HP.BLCKSZ + (lp & 0x7FFF)

    def broken(mut, match=None):
        b = bytearray(good)
        mut(b)
        t2 = HP.Table(ATTRS)
        t2.heap.pages = [b[i:i + HP.BLCKSZ] for i in range(0, len(b), HP.BLCKSZ)]
        t2.toast = t.toast
        with pytest.raises(VsError, match=match):
            read_with_lib(t2, tids, dim)

    read_with_lib(t, tids, dim)
    broken(lambda b: struct.pack_into("<H", b, blk * HP.BLCKSZ + 18, 4096 | 4), "pagesize")          # another page size
    broken(lambda b: struct.pack_into("<I", b, lp_at, (lp & ~0x7FFF) | 8), "outside")                 # item inside the header
    broken(lambda b: b.__setitem__(tup + 22, 255), "t_hoff")                                          # t_hoff not MAXALIGNed
    broken(lambda b: b.__setitem__(tup + 22, 200))                                                    # t_hoff into the data: caught downstream
    spans = HP.parse_tuple(ATTRS, bytes(good[tup:tup + (lp >> 17)]))
    vs, vl = spans[VEC_ATT - 1]
    broken(lambda b: struct.pack_into("<h", b, tup + vs + 4, dim + 1), "dimensions")                  # a vector of another width
    broken(lambda b: struct.pack_into("<I", b, tup + vs, (vl + 4000) << 2), "runs past")             # varlena longer than the tuple
    # an external pointer whose chunks never arrive is reported, not invented
    t3, tids3, exp3 = make_table(5, 768, seed=9, null_frac=0.0)
    from pgvectorscale_amd.pages import HeapColumn
    hc = HeapColumn(ATTRS, VEC_ATT, 768, tids3)
    hc.add(t3.heap.tobytes())
    hc.toast_add(t3.toast.tobytes()[:HP.BLCKSZ])  # only the first TOAST block
    info, found = hc.finish()
    assert info["n_toast_incomplete"] > 0 and found.sum() + info["n_toast_incomplete"] == 5
    hc.close()
