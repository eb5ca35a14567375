"""TEST INFRASTRUCTURE ONLY — a restatement of how the reference lays a `diskann` index out on PostgreSQL pages, used
to manufacture index relations byte by byte (there is no PostgreSQL / Rust toolchain in this image) and to read them
back independently of libvsgpu's reader (pgvectorscale_amd/csrc/vs_pages.cpp).  Only tests/ may import this module.

Restated pieces ("UT/" = /root/reference/pgvectorscale/src/util/, "AM/" = .../src/access_method/):
  * PageInit / PageAddItemExtended / PageGetFreeSpace of PostgreSQL's bufpage.c as WritablePage drives them
    (UT/page.rs:107-215): 24-byte header, 4-byte line pointers (lp_off:15, lp_flags:2, lp_len:15), items MAXALIGNed
    downwards from pd_special, special area = TsvPageOpaqueData {page_type u8, reserved u8, page_id u16 = 0xAE24}
    (UT/page.rs:24-76);
  * Tape::write (UT/tape.rs:53-76) and ChainTapeWriter::write / ChainItemIterator (UT/chain.rs:72-185);
  * the MetaPage body (String, Option<StartNodes> with its BTreeMap) the same way, see rkyv_meta_page below;
  * rkyv 0.7 `to_bytes` for the structs on this path (size_32, little endian): out-of-line data of each field in
    field order, each aligned to its element type, then the root object aligned to its own alignment at the END;
    ArchivedVec = {i32 offset relative to the field, u32 len}; archived_root = last size_of::<Archived<T>>() bytes
    (pgvectorscale_derive/src/lib.rs:35-40).

Pin status: the page arithmetic is pinned by the reference's own KAT (`tape_resume`, UT/tape.rs:100-171: free space
8104 after four 3-byte items; an 8109-byte item forces a new page) and by `test_chain_tape` (UT/chain.rs:217-294,
round trips around 1x/2x/3x BLCKSZ) — see tests/test_pages.py.  The field ORDER inside ArchivedClassicSbqNode /
ArchivedLabeledSbqNode / ArchivedSbqMeans is declaration order here; rkyv 0.7 archives are repr(Rust), so that order is
"parity unpinned" (no fixture of real index pages exists in the reference); libvsgpu takes the offsets as a parameter.
"""
import struct

import numpy as np

BLCKSZ = 8192
SIZE_OF_PAGE_HEADER = 24
TSV_PAGE_ID = 0xAE24
TSV_MAGIC_NUMBER = 768756476  # AM/meta_page.rs:22
TSV_VERSION = 3               # AM/meta_page.rs:23
INVALID_BLOCK = 0xFFFFFFFF
INVALID_OFFSET = 0
# PageType (UT/page.rs:28-39)
PT_META_V1, PT_NODE, PT_PQ_DEF, PT_PQ_VEC, PT_SBQ_MEANS_V1, PT_SBQ_NODE, PT_META_V2, PT_SBQ_MEANS, PT_META = range(9)
CHAIN_ITEM_HEADER_SIZE = 8  # size_of::<ArchivedChainItemHeader>() = ArchivedItemPointer (UT/chain.rs:27-33)


def maxalign(x):
    return (x + 7) & ~7


def is_chained(page_type):  # UT/page.rs:60-62
    return page_type in (PT_SBQ_MEANS, PT_META)


class Relation:
    """The main fork of an index relation: a growable list of BLCKSZ pages."""

    def __init__(self):
        self.pages = []

    def new_page(self, page_type):
        """WritablePage::new + reinit (UT/page.rs:107-135): PageInit with a 4-byte special area."""
        p = bytearray(BLCKSZ)
        special = BLCKSZ - maxalign(4)
        struct.pack_into("<HHHH", p, 12, SIZE_OF_PAGE_HEADER, special, special, BLCKSZ | 4)  # lower, upper, special, size|version
        struct.pack_into("<BBH", p, special, page_type, 0, TSV_PAGE_ID)
        self.pages.append(p)
        return len(self.pages) - 1

    def add_zero_page(self):
        self.pages.append(bytearray(BLCKSZ))
        return len(self.pages) - 1

    # -- page accessors ------------------------------------------------------------------------------------------
    def _hdr(self, blk):
        return struct.unpack_from("<HHHH", self.pages[blk], 12)

    def page_type(self, blk):
        lower, upper, special, _ = self._hdr(blk)
        pt, _, pid = struct.unpack_from("<BBH", self.pages[blk], special)
        assert pid == TSV_PAGE_ID
        return pt

    def free_space(self, blk):
        """PageGetFreeSpace"""
        lower, upper, _, _ = self._hdr(blk)
        space = upper - lower
        return 0 if space < 4 else space - 4

    def aligned_free_space(self, blk):
        """WritablePage::get_aligned_free_space (UT/page.rs:187-190)"""
        fs = self.free_space(blk)
        return fs - fs % 8

    def max_offset(self, blk):
        lower = self._hdr(blk)[0]
        return 0 if lower <= SIZE_OF_PAGE_HEADER else (lower - SIZE_OF_PAGE_HEADER) // 4

    def add_item(self, blk, data):
        """PageAddItemExtended(page, item, size, InvalidOffsetNumber, 0) -> OffsetNumber (UT/page.rs:142-163)"""
        p = self.pages[blk]
        lower, upper, special, _ = self._hdr(blk)
        size = len(data)
        assert size < BLCKSZ
        off = self.max_offset(blk) + 1
        new_lower = lower + 4
        new_upper = upper - maxalign(size)
        assert new_lower <= new_upper, "PageAddItemExtended: no room"
        p[new_upper:new_upper + size] = data
        struct.pack_into("<I", p, SIZE_OF_PAGE_HEADER + 4 * (off - 1), new_upper | (1 << 15) | (size << 17))  # LP_NORMAL
        struct.pack_into("<HH", p, 12, new_lower, new_upper)
        return off

    def item_span(self, blk, off):
        """PageGetItemId + PageGetItem (UT/ports.rs:56-77): (start, len) of the item inside the page"""
        assert 1 <= off <= self.max_offset(blk)
        lp = struct.unpack_from("<I", self.pages[blk], SIZE_OF_PAGE_HEADER + 4 * (off - 1))[0]
        return lp & 0x7FFF, lp >> 17

    def item(self, blk, off):
        s, l = self.item_span(blk, off)
        return bytes(self.pages[blk][s:s + l])

    def tobytes(self):
        return b"".join(bytes(p) for p in self.pages)


class Tape:
    """UT/tape.rs:15-76"""

    def __init__(self, rel, page_type, current=None):
        assert not is_chained(page_type)
        self.rel, self.page_type = rel, page_type
        self.current = rel.new_page(page_type) if current is None else current

    @classmethod
    def resume(cls, rel, page_type):
        """Tape::resume (UT/tape.rs:30-51): continue on the newest page of this type, else start a new one"""
        for blk in reversed(range(len(rel.pages))):
            if rel.page_type(blk) == page_type:
                return cls(rel, page_type, current=blk)
        return cls(rel, page_type)

    def write(self, data):
        size = len(data)
        assert size < BLCKSZ
        if self.rel.aligned_free_space(self.current) < size:  # don't split data over pages
            self.current = self.rel.new_page(self.page_type)
            assert self.rel.aligned_free_space(self.current) >= size, "Not enough free space on new page"
        return self.current, self.rel.add_item(self.current, data)


def rkyv_item_pointer(block, offset):
    return struct.pack("<IHH", block, offset, 0)


class ChainTapeWriter:
    """UT/chain.rs:35-131"""

    def __init__(self, rel, page_type):
        assert is_chained(page_type)
        self.rel, self.page_type = rel, page_type
        self.current = rel.new_page(page_type)

    @classmethod
    def reinit(cls, rel, page_type, block):
        """ChainTapeWriter::reinit (UT/chain.rs:51-70): start again on an existing block, which is initialised afresh"""
        w = cls.__new__(cls)
        w.rel, w.page_type = rel, page_type
        fresh = Relation()
        fresh.new_page(page_type)
        rel.pages[block] = fresh.pages[0]
        w.current = block
        return w

    def write(self, data):
        rel = self.rel
        cur = self.current
        if rel.aligned_free_space(cur) < CHAIN_ITEM_HEADER_SIZE + 1:
            cur = rel.new_page(self.page_type)
        result = None
        while CHAIN_ITEM_HEADER_SIZE + len(data) > rel.aligned_free_space(cur):
            nxt = rel.new_page(self.page_type)
            data_size = rel.aligned_free_space(cur) - CHAIN_ITEM_HEADER_SIZE
            off = rel.add_item(cur, rkyv_item_pointer(nxt, 1) + data[:data_size])
            if result is None:
                result = (cur, off)
            cur = nxt
            data = data[data_size:]
        off = rel.add_item(cur, rkyv_item_pointer(INVALID_BLOCK, INVALID_OFFSET) + data)
        if result is None:
            result = (cur, off)
        self.current = cur
        return result


def read_chain(rel, block, offset, page_type):
    """ChainItemIterator (UT/chain.rs:159-185)"""
    out = b""
    while block != INVALID_BLOCK:
        assert rel.page_type(block) == page_type
        it = rel.item(block, offset)
        assert len(it) > CHAIN_ITEM_HEADER_SIZE
        block, offset, _ = struct.unpack_from("<IHH", it, 0)
        out += it[CHAIN_ITEM_HEADER_SIZE:]
    return out


# ---- rkyv 0.7 to_bytes of the structs on this path ------------------------------------------------------------------
class _Ser:
    def __init__(self):
        self.b = bytearray()

    def align(self, a):
        while len(self.b) % a:
            self.b.append(0)
        return len(self.b)

    def write(self, data):
        pos = len(self.b)
        self.b += data
        return pos


def _vec_field(field_pos, data_pos, n):
    return struct.pack("<iI", data_pos - field_pos, n)


DEFAULT_NODE_LAYOUT = (32, 0, 8, 16, 24)  # root size, heap_item_pointer, bq_vector, neighbor_index_pointers, last field


def rkyv_sbq_node(heap_ptr, code, nbr_ptrs, labels=None, layout=DEFAULT_NODE_LAYOUT):
    """to_bytes(ClassicSbqNode | LabeledSbqNode) (AM/sbq/node.rs:26-42): bq_vector data, neighbor ItemPointers, then the
    labels (labeled) — `_neighbor_vectors` of a classic node is always empty (AM/sbq/node.rs:82) — then the 32-byte root:
    heap_item_pointer | bq_vector | neighbor_index_pointers | _neighbor_vectors or labels."""
    s = _Ser()
    p_code = s.align(8)
    s.write(np.asarray(code, "<u8").tobytes())
    p_nbr = s.align(4)
    s.write(b"".join(rkyv_item_pointer(b, o) for (b, o) in nbr_ptrs))
    if labels is None:
        p_last, n_last = s.align(4), 0  # empty Vec<Vec<u64>>
    else:
        p_last, n_last = s.align(2), len(labels)
        s.write(np.asarray(labels, "<i2").tobytes())
    root = s.align(4)
    size, o_heap, o_code, o_nbr, o_last = layout
    s.write(b"\0" * size)
    s.b[root + o_heap:root + o_heap + 8] = rkyv_item_pointer(*heap_ptr)
    s.b[root + o_code:root + o_code + 8] = _vec_field(root + o_code, p_code, len(code))
    s.b[root + o_nbr:root + o_nbr + 8] = _vec_field(root + o_nbr, p_nbr, len(nbr_ptrs))
    s.b[root + o_last:root + o_last + 8] = _vec_field(root + o_last, p_last, n_last)
    return bytes(s.b)


DEFAULT_PLAIN_LAYOUT = (32, 24, 0, 16, None)  # root size, heap_item_pointer, vector, neighbor_index_pointers, (no labels)


def rkyv_plain_node(heap_ptr, vector, nbr_ptrs, layout=DEFAULT_PLAIN_LAYOUT):
    """to_bytes(PlainNode {vector: Vec<f32>, pq_vector: Vec<u8>, neighbor_index_pointers: Vec<ItemPointer>, heap_item_pointer})
    (AM/plain/node.rs:15-22): the vector's floats, the (always empty) pq_vector, the neighbor ItemPointers, then the 32-byte root"""
    s = _Ser()
    p_vec = s.align(4)
    s.write(np.asarray(vector, "<f4").tobytes())
    p_pq = s.align(1)
    p_nbr = s.align(4)
    s.write(b"".join(rkyv_item_pointer(b, o) for (b, o) in nbr_ptrs))
    root = s.align(4)
    size, o_heap, o_vec, o_nbr, _ = layout
    s.write(b"\0" * size)
    o_pq = next(o for o in (0, 8, 16, 24) if o not in (o_heap, o_vec, o_nbr))
    s.b[root + o_heap:root + o_heap + 8] = rkyv_item_pointer(*heap_ptr)
    s.b[root + o_vec:root + o_vec + 8] = _vec_field(root + o_vec, p_vec, len(vector))
    s.b[root + o_pq:root + o_pq + 8] = _vec_field(root + o_pq, p_pq, 0)
    s.b[root + o_nbr:root + o_nbr + 8] = _vec_field(root + o_nbr, p_nbr, len(nbr_ptrs))
    return bytes(s.b)


def write_plain_index(*, vectors, nbrs, heap_tids, num_neighbors=None, meta=None, layout=DEFAULT_PLAIN_LAYOUT):
    """a `plain` storage index relation: block 0 = Meta chain, PlainNode items on PageType::Node pages written through a Tape
    (AM/plain/node.rs:66-70), neighbor lists patched in place afterwards as the reference's build does"""
    n, D = vectors.shape
    R = num_neighbors or nbrs.shape[1]
    rel = Relation()
    mw = ChainTapeWriter(rel, PT_META)
    assert mw.write(rkyv_meta_header()) == (0, 1)
    body = b"\x00" * 120 if meta is None else rkyv_meta_page(**dict(meta, default_start=None, labeled_starts=None))
    assert mw.write(body) == (0, 2)
    tape = Tape(rel, PT_NODE)
    empty = [(INVALID_BLOCK, INVALID_OFFSET)] * R
    ptrs = []
    for i in range(n):
        tid = int(heap_tids[i])
        ptrs.append(tape.write(rkyv_plain_node((tid >> 16, tid & 0xFFFF), vectors[i], empty, layout)))
    for i in range(n):
        blk, off = ptrs[i]
        s, l = rel.item_span(blk, off)
        page = rel.pages[blk]
        fld = s + l - layout[0] + layout[3]
        rel_off, cnt = struct.unpack_from("<iI", page, fld)
        assert cnt == R
        at = fld + rel_off
        for j in range(R):
            v = int(nbrs[i, j]) if j < nbrs.shape[1] else INVALID_BLOCK
            if v == INVALID_BLOCK:
                break
            page[at + 8 * j:at + 8 * j + 8] = rkyv_item_pointer(*ptrs[v])
    if meta is not None:
        final = dict(meta)
        ds = final.pop("default_start", None)
        final.pop("labeled_starts", None)
        body = rkyv_meta_page(default_start=None if ds is None else ptrs[ds], **final)
        again = ChainTapeWriter.reinit(rel, PT_META, 0)
        assert again.write(rkyv_meta_header()) == (0, 1)
        assert again.write(body) == (0, 2)
    return WrittenIndex(rel, ptrs, None, False, layout)


def rkyv_sbq_means(count, mean, m2):
    """to_bytes(SbqMeans {count: u64, means: Vec<f32>, m2: Vec<f32>}) (AM/sbq/mod.rs:62-69)"""
    s = _Ser()
    p_mean = s.align(4)
    s.write(np.asarray(mean, "<f4").tobytes())
    p_m2 = s.align(4)
    s.write(np.asarray(m2, "<f4").tobytes())
    root = s.align(8)
    s.write(struct.pack("<Q", count))
    s.write(_vec_field(root + 8, p_mean, len(mean)))
    s.write(_vec_field(root + 16, p_m2, len(m2)))
    return bytes(s.b)


def rkyv_meta_header(magic=TSV_MAGIC_NUMBER, version=TSV_VERSION):
    """to_bytes(MetaPageHeader {magic_number: u32, version: u32}) (AM/meta_page.rs:166-174)"""
    return struct.pack("<II", magic, version)


# ---- MetaPage (AM/meta_page.rs:176-210) ------------------------------------------------------------------------------
# rkyv 0.7.43 (size_32, little endian; un-vendored dependency, restated from its published sources — "parity unpinned": no
# fixture of a real meta page exists in the reference, oracle/ref_kat.rs prints one for a maintainer with the toolchain):
#   * ArchivedString, 8 bytes (string/repr.rs): up to 7 bytes inline {bytes[7], len u8}; longer {len u32, offset i32 LE
#     relative to the START of the repr} with the bytes written before the struct — the offset is negative, so the sign bit
#     of the last byte tells the two apart (is_inline: last byte & 0x80 == 0);
#   * ArchivedOption<T>: #[repr(u8)] enum {None = 0, Some(T) = 1}: the tag byte, then T at T's alignment;
#   * ArchivedBTreeMap<K, V> (collections/btree_map): {len u32, root RelPtr i32 relative to the root field}; nodes =
#     NodeHeader {meta u16 (bit 15 = inner node, low 15 bits = entries), size u32 (bytes of the node), ptr i32 (leaf: next
#     leaf in key order, 0 = none; inner: the child below the first key)} followed by the entries, LeafNodeEntry {key, value}
#     or InnerNodeEntry {ptr i32, key}; built from the keys in REVERSE order, a node is closed once it has reached 4096
#     bytes, so the leaf with the largest keys is written first and the root last;
#   * ArchivedItemPointer {block_number u32, offset u16} = 8 bytes, 4-aligned (UT/mod.rs:17-23);
#   * the root object's fields in declaration order with C alignment rules (rkyv 0.7 archives are repr(Rust): the order is
#     a parameter of the decoders, DEFAULT_META_LAYOUT).
META_FIELDS = ("magic_number", "version", "extension_version_when_built", "distance_type", "num_dimensions",
               "num_dimensions_to_index", "bq_num_bits_per_dimension", "storage_type", "num_neighbors", "search_list_size",
               "max_alpha", "start_nodes", "quantizer_metadata", "has_labels")
_META_SIZE_ALIGN = {"magic_number": (4, 4), "version": (4, 4), "extension_version_when_built": (8, 4), "distance_type": (2, 2),
                    "num_dimensions": (4, 4), "num_dimensions_to_index": (4, 4), "bq_num_bits_per_dimension": (1, 1),
                    "storage_type": (1, 1), "num_neighbors": (4, 4), "search_list_size": (4, 4), "max_alpha": (8, 8),
                    "start_nodes": (20, 4), "quantizer_metadata": (8, 4), "has_labels": (1, 1)}


def meta_layout(order=META_FIELDS):
    """{"root_size": n, field: offset, ...} for the fields laid out in `order` with C alignment rules"""
    pos, out = 0, {}
    for f in order:
        size, al = _META_SIZE_ALIGN[f]
        pos = (pos + al - 1) // al * al
        out[f] = pos
        pos += size
    out["root_size"] = (pos + 7) // 8 * 8
    return out


DEFAULT_META_LAYOUT = meta_layout()
BTREE_MAX_NODE_SIZE = 4096
_BT_HEADER = 12   # NodeHeader {u16 meta, (pad), u32 size, i32 ptr}
_BT_LEAF_ENTRY = 12   # LeafNodeEntry<i16, ArchivedItemPointer> {i16 key, (pad), u32 block, u16 offset, (pad)}
_BT_INNER_ENTRY = 8   # InnerNodeEntry<i16> {i32 ptr, i16 key, (pad)}


def _btree_nodes(s, entries):
    """BTreeMap::serialize -> ArchivedBTreeMap::serialize_from_reverse_iter: writes the nodes, returns the root's position"""
    level = []  # (first key, node position), built back to front
    rev = list(reversed(entries))
    i = 0
    next_leaf = None
    while i < len(rev):
        block_start = len(s.b)
        grp = [rev[i]]
        i += 1
        while True:
            est = len(s.b) - block_start + _BT_HEADER + len(grp) * _BT_LEAF_ENTRY
            if est >= BTREE_MAX_NODE_SIZE and len(grp) >= 1:
                break
            if i < len(rev):
                grp.append(rev[i])
                i += 1
            else:
                break
        pos = s.align(4)
        grp.reverse()  # entries are stored in key order
        size = _BT_HEADER + len(grp) * _BT_LEAF_ENTRY
        ptr = 0 if next_leaf is None else next_leaf - (pos + 8)
        s.write(struct.pack("<HHIi", len(grp), 0, size, ptr))
        for key, (blk, off) in grp:
            s.write(struct.pack("<hHIHH", key, 0, blk, off, 0))
        next_leaf = pos
        level.append((grp[0][0], pos))
    while len(level) > 1:
        nxt = []
        j = 0
        while j < len(level):  # `level` is still in reverse key order
            block_start = len(s.b)
            grp = [level[j]]
            j += 1
            while True:
                est = len(s.b) - block_start + _BT_HEADER + len(grp) * _BT_INNER_ENTRY
                if est >= BTREE_MAX_NODE_SIZE and len(grp) >= 2:
                    break
                if j < len(level):
                    grp.append(level[j])
                    j += 1
                else:
                    break
            # the smallest child hangs off the header, the others become (ptr, key) entries in key order
            grp.reverse()
            first_key, first_pos = grp[0]
            rest = grp[1:]
            if not rest and j >= len(level) and not nxt:  # a lone child: it is the root itself
                nxt.append((first_key, first_pos))
                continue
            pos = s.align(4)
            size = _BT_HEADER + len(rest) * _BT_INNER_ENTRY
            s.write(struct.pack("<HHIi", 0x8000 | len(rest), 0, size, first_pos - (pos + 8)))
            for n_, (key, cpos) in enumerate(rest):
                epos = pos + _BT_HEADER + n_ * _BT_INNER_ENTRY
                s.write(struct.pack("<ihH", cpos - epos, key, 0))
            nxt.append((first_key, pos))
        level = nxt
    return level[0][1]


def rkyv_meta_page(*, extension_version="0.8.0", distance_type=1, num_dimensions, num_dimensions_to_index=None,
                   bq_num_bits_per_dimension=2, storage_type=2, num_neighbors=50, search_list_size=100, max_alpha=1.2,
                   default_start=None, labeled_starts=None, quantizer=(INVALID_BLOCK, INVALID_OFFSET), has_labels=False,
                   magic=TSV_MAGIC_NUMBER, version=TSV_VERSION, layout=DEFAULT_META_LAYOUT):
    """to_bytes(MetaPage) (AM/meta_page.rs:176-210, store: :344-365).  default_start None -> start_nodes = None (an index that
    never saw a row); labeled_starts = {label: (block, offset)} (StartNodes.labeled_nodes, AM/graph/start_nodes.rs:14-22)."""
    s = _Ser()
    ver = extension_version.encode()
    p_ver = None
    if len(ver) > 7:
        p_ver = s.write(ver)
    root_pos = None
    entries = sorted((labeled_starts or {}).items())
    if default_start is not None and entries:
        root_pos = _btree_nodes(s, entries)
    root = s.align(8)
    s.write(b"\0" * layout["root_size"])
    b = s.b

    def put(name, data):
        b[root + layout[name]:root + layout[name] + len(data)] = data

    put("magic_number", struct.pack("<I", magic))
    put("version", struct.pack("<I", version))
    f = root + layout["extension_version_when_built"]
    if p_ver is None:
        put("extension_version_when_built", ver.ljust(7, b"\0") + bytes([len(ver)]))
    else:
        put("extension_version_when_built", struct.pack("<Ii", len(ver), p_ver - f))
    put("distance_type", struct.pack("<H", distance_type))
    put("num_dimensions", struct.pack("<I", num_dimensions))
    put("num_dimensions_to_index", struct.pack("<I", num_dimensions if num_dimensions_to_index is None else num_dimensions_to_index))
    put("bq_num_bits_per_dimension", bytes([bq_num_bits_per_dimension]))
    put("storage_type", bytes([storage_type]))
    put("num_neighbors", struct.pack("<I", num_neighbors))
    put("search_list_size", struct.pack("<I", search_list_size))
    put("max_alpha", struct.pack("<d", max_alpha))
    if default_start is not None:
        sn = root + layout["start_nodes"]
        root_field = sn + 4 + 8 + 4
        put("start_nodes", bytes([1, 0, 0, 0]) + rkyv_item_pointer(*default_start) +
            struct.pack("<Ii", len(entries), 0 if root_pos is None else root_pos - root_field))
    put("quantizer_metadata", rkyv_item_pointer(*quantizer))
    put("has_labels", bytes([1 if has_labels else 0]))
    return bytes(b)


def parse_meta_page(data, layout=DEFAULT_META_LAYOUT):
    """rkyv::from_bytes::<MetaPage> (AM/meta_page.rs:367-378), independent of libvsgpu's decoder -> dict"""
    root = len(data) - layout["root_size"]
    assert root >= 0

    def at(name):
        return root + layout[name]

    out = {"magic_number": struct.unpack_from("<I", data, at("magic_number"))[0],
           "version": struct.unpack_from("<I", data, at("version"))[0]}
    f = at("extension_version_when_built")
    if data[f + 7] & 0x80 == 0:
        out["extension_version_when_built"] = bytes(data[f:f + data[f + 7]]).decode()
    else:
        ln, off = struct.unpack_from("<Ii", data, f)
        assert 0 <= f + off and f + off + ln <= len(data)
        out["extension_version_when_built"] = bytes(data[f + off:f + off + ln]).decode()
    out["distance_type"] = struct.unpack_from("<H", data, at("distance_type"))[0]
    out["num_dimensions"] = struct.unpack_from("<I", data, at("num_dimensions"))[0]
    out["num_dimensions_to_index"] = struct.unpack_from("<I", data, at("num_dimensions_to_index"))[0]
    out["bq_num_bits_per_dimension"] = data[at("bq_num_bits_per_dimension")]
    out["storage_type"] = data[at("storage_type")]
    out["num_neighbors"] = struct.unpack_from("<I", data, at("num_neighbors"))[0]
    out["search_list_size"] = struct.unpack_from("<I", data, at("search_list_size"))[0]
    out["max_alpha"] = struct.unpack_from("<d", data, at("max_alpha"))[0]
    sn = at("start_nodes")
    if data[sn] == 0:
        out["default_start"], out["labeled_starts"] = None, {}
    else:
        b_, o_, _ = struct.unpack_from("<IHH", data, sn + 4)
        out["default_start"] = (b_, o_)
        ln, roff = struct.unpack_from("<Ii", data, sn + 12)
        starts = {}

        def walk(pos, depth):
            assert depth < 8 and 0 <= pos and pos + _BT_HEADER <= len(data)
            meta, _, _, ptr = struct.unpack_from("<HHIi", data, pos)
            cnt = meta & 0x7FFF
            if meta & 0x8000:
                walk(pos + 8 + ptr, depth + 1)
                for e in range(cnt):
                    ep = pos + _BT_HEADER + e * _BT_INNER_ENTRY
                    walk(ep + struct.unpack_from("<i", data, ep)[0], depth + 1)
            else:
                for e in range(cnt):
                    k_, _, bb, oo, _ = struct.unpack_from("<hHIHH", data, pos + _BT_HEADER + e * _BT_LEAF_ENTRY)
                    starts[k_] = (bb, oo)

        if ln:
            walk(sn + 16 + roff, 0)
        assert len(starts) == ln and list(starts) == sorted(starts)
        out["labeled_starts"] = starts
    b_, o_, _ = struct.unpack_from("<IHH", data, at("quantizer_metadata"))
    out["quantizer_metadata"] = (b_, o_)
    out["has_labels"] = bool(data[at("has_labels")])
    return out


def _archived_vec(item, field, elem):
    off, n = struct.unpack_from("<iI", item, field)
    start = field + off
    if n:
        assert 0 <= start and start + n * elem <= len(item)
    return start, n


def parse_sbq_node(item, has_labels, layout=DEFAULT_NODE_LAYOUT):
    """rkyv::archived_root::<SbqNode> + the accessors of ArchivedSbqNode (AM/sbq/node.rs:260-311)"""
    size, o_heap, o_code, o_nbr, o_last = layout
    root = len(item) - size
    hb, ho, _ = struct.unpack_from("<IHH", item, root + o_heap)
    s, n = _archived_vec(item, root + o_code, 8)
    code = np.frombuffer(item, "<u8", n, s).copy()
    s, n_slots = _archived_vec(item, root + o_nbr, 8)
    nbrs = []
    for j in range(n_slots):  # iter_neighbors: take(num_neighbors()) = up to the first InvalidBlockNumber
        b, o, _ = struct.unpack_from("<IHH", item, s + 8 * j)
        if b == INVALID_BLOCK:
            break
        nbrs.append((b, o))
    labels = None
    if has_labels:
        s, n = _archived_vec(item, root + o_last, 2)
        labels = np.frombuffer(item, "<i2", n, s).copy()
    return (hb, ho), code, nbrs, n_slots, labels


def parse_sbq_means(data):
    root = len(data) - 24
    count = struct.unpack_from("<Q", data, root)[0]
    s, n = _archived_vec(data, root + 8, 4)
    mean = np.frombuffer(data, "<f4", n, s).copy()
    s, n = _archived_vec(data, root + 16, 4)
    m2 = np.frombuffer(data, "<f4", n, s).copy()
    return count, mean, m2


# ---- a whole index -------------------------------------------------------------------------------------------------
class WrittenIndex:
    def __init__(self, rel, node_ptrs, means_ptr, has_labels, layout):
        self.rel, self.node_ptrs, self.means_ptr, self.has_labels, self.layout = rel, node_ptrs, means_ptr, has_labels, layout


def write_index(*, codes, nbrs, heap_tids, mean, m2, count, label_off=None, label_val=None, num_neighbors=None,
                means_first=True, meta_body=b"\x00" * 120, meta=None, zero_page_every=0,
                layout=DEFAULT_NODE_LAYOUT):
    """Lay an index out the way the reference's build does: block 0 = Meta chain (header item 1, MetaPage item 2,
    AM/meta_page.rs:344-365), a chained SbqMeans item (AM/sbq/mod.rs:123-137) and SbqNode items written through a Tape
    (AM/sbq/node.rs:118-123).  Nodes are first written with empty neighbor lists (SbqNode::new, AM/sbq/node.rs:55-90)
    and patched in place afterwards, like the reference's set_neighbors_on_disk.  `nbrs` holds dense ids; the list of a
    node ends at the first 0xFFFFFFFF.  `meta` = keyword arguments of rkyv_meta_page with default_start / labeled_starts given
    as dense node ids: the MetaPage is first stored without start nodes and quantizer pointer (MetaPage::create,
    AM/meta_page.rs:297-359) and stored again on a re-initialised block 0 once they are known (store(index, false)), chaining
    onto fresh pages when it has outgrown the block; without `meta` the body is the opaque placeholder `meta_body`."""
    n, W = codes.shape
    R = num_neighbors or nbrs.shape[1]
    has_labels = label_off is not None
    rel = Relation()
    meta_w = meta
    meta = ChainTapeWriter(rel, PT_META)
    assert meta.write(rkyv_meta_header()) == (0, 1)
    if meta_w is not None:
        first = dict(meta_w, default_start=None, labeled_starts=None)
        first.pop("quantizer", None)
        meta_body = rkyv_meta_page(has_labels=has_labels, **first)
    assert meta.write(meta_body) == (0, 2)
    means_ptr = None
    if means_first:
        means_ptr = ChainTapeWriter(rel, PT_SBQ_MEANS).write(rkyv_sbq_means(count, mean, m2))
    tape = Tape(rel, PT_SBQ_NODE)
    empty = [(INVALID_BLOCK, INVALID_OFFSET)] * R
    ptrs = []
    for i in range(n):
        if zero_page_every and i and i % zero_page_every == 0:
            rel.add_zero_page()  # a block the relation was extended by but never initialised
        tid = int(heap_tids[i])
        labels = None if not has_labels else label_val[label_off[i]:label_off[i + 1]]
        ptrs.append(tape.write(rkyv_sbq_node((tid >> 16, tid & 0xFFFF), codes[i], empty, labels, layout)))
    for i in range(n):
        blk, off = ptrs[i]
        s, l = rel.item_span(blk, off)
        page = rel.pages[blk]
        fld = s + l - layout[0] + layout[3]
        rel_off, cnt = struct.unpack_from("<iI", page, fld)
        assert cnt == R
        at = fld + rel_off
        for j in range(R):
            v = int(nbrs[i, j]) if j < nbrs.shape[1] else INVALID_BLOCK
            if v == INVALID_BLOCK:
                break
            page[at + 8 * j:at + 8 * j + 8] = rkyv_item_pointer(*ptrs[v])
    if not means_first:
        means_ptr = ChainTapeWriter(rel, PT_SBQ_MEANS).write(rkyv_sbq_means(count, mean, m2))
    if meta_w is not None:
        final = dict(meta_w)
        ds = final.pop("default_start", None)
        ls = final.pop("labeled_starts", None) or {}
        final.setdefault("quantizer", means_ptr)
        body = rkyv_meta_page(has_labels=has_labels, default_start=None if ds is None else ptrs[ds],
                              labeled_starts={k: ptrs[v] for k, v in ls.items()}, **final)
        again = ChainTapeWriter.reinit(rel, PT_META, 0)
        assert again.write(rkyv_meta_header()) == (0, 1)
        assert again.write(body) == (0, 2)
    return WrittenIndex(rel, ptrs, means_ptr, has_labels, layout)


def read_index(w):
    """Independent (pure Python) read-back of a written relation into flat arrays with the dense numbering of
    vs_pages.cpp: node id = SbqNode items on earlier blocks + offset - 1."""
    rel = w.rel
    base, ids = {}, 0
    for blk in range(len(rel.pages)):
        if struct.unpack_from("<H", rel.pages[blk], 14)[0] == 0:  # PageIsNew
            continue
        if rel.page_type(blk) == PT_SBQ_NODE:
            base[blk] = ids
            ids += rel.max_offset(blk)
    codes, nbrs, tids, labs = [], [], [], []
    for blk in sorted(base):
        for off in range(1, rel.max_offset(blk) + 1):
            (hb, ho), code, nb, _, labels = parse_sbq_node(rel.item(blk, off), w.has_labels, w.layout)
            codes.append(code)
            tids.append((hb << 16) | ho)
            nbrs.append([base[b] + o - 1 for (b, o) in nb])
            labs.append(labels)
    return codes, nbrs, tids, labs
