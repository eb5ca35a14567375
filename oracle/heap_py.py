"""TEST INFRASTRUCTURE ONLY — a restatement of how PostgreSQL lays a heap relation (and its TOAST relation) out on pages, used to
manufacture the table a `diskann` index points into byte by byte (there is no PostgreSQL in this image) and to read it back
independently of libvsgpu's reader (pgvectorscale_amd/csrc/vs_heap.cpp).  Only tests/ may import this module.

What the reference does with these pages: the rescore window fetches the heap tuple an index node points at
(`table_index_fetch_tuple`, UT/table_slot.rs:19-42), takes the vector column (`slot_getattr` + `pg_detoast_datum_copy`,
AM/pg_vector.rs:125-135, AM/sbq/storage.rs:304-328) and computes the full-precision distance.  This port stages that column
once, in bulk, into HBM (`vs_index_host.vecs`); the reader must therefore find the same bytes `heap_getattr` + detoasting find.

Restated pieces of PostgreSQL (access/htup_details.h, access/heaptoast.h, postgres.h / varatt.h, storage/bufpage.h; PG 13-17,
little endian, 8 KB pages, MAXALIGN 8):
  * page header 24 bytes, 4-byte line pointers (lp_off:15, lp_flags:2, lp_len:15; LP_UNUSED 0, LP_NORMAL 1, LP_REDIRECT 2 with
    lp_off = the line pointer redirected to, LP_DEAD 3), tuples MAXALIGNed downwards from pd_special (no special area in a heap);
  * HeapTupleHeaderData: t_xmin 4, t_xmax 4, t_cid 4, t_ctid 6, t_infomask2 2 (low 11 bits = number of attributes),
    t_infomask 2 (HEAP_HASNULL 0x0001), t_hoff 1, then the null bitmap (1 = NOT null) when HEAP_HASNULL, padded to MAXALIGN;
  * attribute layout of heap_fill_tuple: fixed-length attributes aligned to attalign ('c' 1, 's' 2, 'i' 4, 'd' 8); a varlena
    whose total size (with a 1-byte header) is <= 127 is stored with the 1-byte header and NO alignment padding, otherwise with a
    4-byte header at attalign; readers tell the two apart by the first byte (a pad byte is 0, a 1-byte header never is);
  * varlena headers (little endian): 4-byte uncompressed = len << 2; 1-byte = (len << 1) | 1; external = 0x01 then the tag byte
    VARTAG_ONDISK = 18 then varatt_external {int32 va_rawsize (incl. the 4-byte header), uint32 va_extinfo (external size, top two
    bits = compression method), Oid va_valueid, Oid va_toastrelid}, 18 bytes, unaligned;
  * TOAST relation rows (chunk_id oid, chunk_seq int4, chunk_data bytea), TOAST_MAX_CHUNK_SIZE = 1996 data bytes per chunk;
  * pgvector's `vector` (vector.h): varlena header, int16 dim, int16 unused, float4 x[dim]; its typstorage is `external`
    (out of line when the row is too wide, never compressed).
"""
import struct

import numpy as np

BLCKSZ = 8192
SIZE_OF_PAGE_HEADER = 24
HEAP_HASNULL = 0x0001
HEAP_HASVARWIDTH = 0x0002
HEAP_HASEXTERNAL = 0x0004
HEAP_XMIN_COMMITTED = 0x0100
HEAP_XMAX_INVALID = 0x0800
VARTAG_ONDISK = 18
LP_UNUSED, LP_NORMAL, LP_REDIRECT, LP_DEAD = 0, 1, 2, 3
TOAST_TUPLE_THRESHOLD = 2032  # MaximumBytesPerTuple(4)
ALIGN = {"c": 1, "s": 2, "i": 4, "d": 8}


def maxalign(x):
    return (x + 7) & ~7


def toast_max_chunk_size(page_size=BLCKSZ):
    """EXTERN_TUPLE_MAX_SIZE - MAXALIGN(SizeofHeapTupleHeader) - sizeof(Oid) - sizeof(int32) - VARHDRSZ (heaptoast.h)"""
    per_tuple = ((page_size - maxalign(SIZE_OF_PAGE_HEADER + 4 * 4)) // 4) & ~7
    return per_tuple - maxalign(23) - 4 - 4 - 4


class HeapRelation:
    """main fork of a heap (or TOAST) relation: PageInit without a special area + PageAddItem"""

    def __init__(self, page_size=BLCKSZ):
        self.page_size = page_size
        self.pages = []

    def new_page(self):
        p = bytearray(self.page_size)
        struct.pack_into("<HHHH", p, 12, SIZE_OF_PAGE_HEADER, self.page_size, self.page_size, self.page_size | 4)
        self.pages.append(p)
        return len(self.pages) - 1

    def _hdr(self, blk):
        return struct.unpack_from("<HHHH", self.pages[blk], 12)

    def free_space(self, blk):
        lower, upper, _, _ = self._hdr(blk)
        return max(0, upper - lower - 4)

    def max_offset(self, blk):
        return (self._hdr(blk)[0] - SIZE_OF_PAGE_HEADER) // 4

    def add_item(self, data):
        """RelationGetBufferForTuple in miniature: the last page, or a new one -> (block, offset)"""
        need = maxalign(len(data))
        if not self.pages or self.free_space(len(self.pages) - 1) < need:
            self.new_page()
        blk = len(self.pages) - 1
        p = self.pages[blk]
        lower, upper, _, _ = self._hdr(blk)
        off = self.max_offset(blk) + 1
        upper -= need
        p[upper:upper + len(data)] = data
        struct.pack_into("<I", p, SIZE_OF_PAGE_HEADER + 4 * (off - 1), upper | (LP_NORMAL << 15) | (len(data) << 17))
        struct.pack_into("<HH", p, 12, lower + 4, upper)
        return blk, off

    def set_line_pointer(self, blk, off, flags, lp_off=0, lp_len=0):
        struct.pack_into("<I", self.pages[blk], SIZE_OF_PAGE_HEADER + 4 * (off - 1), lp_off | (flags << 15) | (lp_len << 17))

    def tobytes(self):
        return b"".join(bytes(p) for p in self.pages)


def varlena_4b(data):
    return struct.pack("<I", (len(data) + 4) << 2) + data


def vector_datum_body(v):
    """the data portion of a pgvector datum: int16 dim, int16 unused, float4[dim]"""
    v = np.asarray(v, "<f4")
    return struct.pack("<hh", v.size, 0) + v.tobytes()


def form_tuple(attrs, values, ctid=(0, 0), natts=None):
    """heap_form_tuple / heap_fill_tuple.  attrs: [(attlen, attalign)], values: per attribute None (NULL), bytes of a fixed-length
    value, or for a varlena ("inline", data) | ("external", rawsize, extsize, valueid, toastrelid) | ("raw", bytes as they are)."""
    natts = len(attrs) if natts is None else natts
    hasnull = any(v is None for v in values[:natts])
    hoff = 23 + ((natts + 7) // 8 if hasnull else 0)
    hoff = maxalign(hoff)
    data = bytearray()
    infomask = HEAP_XMIN_COMMITTED | HEAP_XMAX_INVALID | (HEAP_HASNULL if hasnull else 0)
    for (attlen, attalign), val in list(zip(attrs, values))[:natts]:
        if val is None:
            continue
        if attlen == -1:
            infomask |= HEAP_HASVARWIDTH
            kind = val[0]
            if kind == "external":
                _, rawsize, extsize, valueid, toastrelid = val
                infomask |= HEAP_HASEXTERNAL
                data += bytes([0x01, VARTAG_ONDISK]) + struct.pack("<iIII", rawsize, extsize, valueid, toastrelid)
            elif kind == "raw":
                data += val[1]
            else:
                body = val[1]
                if len(body) + 1 <= 127:  # short header, no alignment (heap_fill_tuple: VARATT_CAN_MAKE_SHORT)
                    data += bytes([((len(body) + 1) << 1) | 1]) + body
                else:
                    while len(data) % ALIGN[attalign]:
                        data.append(0)
                    data += varlena_4b(body)
        else:
            while len(data) % ALIGN[attalign]:
                data.append(0)
            assert attlen < 0 or len(val) == attlen
            data += val
    hdr = bytearray(hoff)
    struct.pack_into("<IIIIHHHB", hdr, 0, 700, 0, 0, ctid[0], ctid[1], natts, infomask, hoff)
    if hasnull:
        for i, v in enumerate(values[:natts]):
            if v is not None:
                hdr[23 + (i >> 3)] |= 1 << (i & 7)
    return bytes(hdr) + bytes(data)


TOAST_ATTRS = [(4, "i"), (4, "i"), (-1, "i")]


class Table:
    """a heap + its TOAST relation; insert() returns the heap TID of the row"""

    def __init__(self, attrs, toast_relid=16999, page_size=BLCKSZ):
        self.attrs = attrs
        self.heap = HeapRelation(page_size)
        self.toast = HeapRelation(page_size)
        self.toast_relid = toast_relid
        self.next_value = 20000
        self.chunk = toast_max_chunk_size(page_size)

    def toast_value(self, body):
        """toast_save_datum: the data portion goes to the TOAST relation in chunks -> the external varlena of the heap tuple"""
        vid = self.next_value
        self.next_value += 1
        for seq, at in enumerate(range(0, len(body), self.chunk)):
            part = body[at:at + self.chunk]
            self.toast.add_item(form_tuple(TOAST_ATTRS, [struct.pack("<I", vid), struct.pack("<i", seq), ("inline", part)]))
        return ("external", len(body) + 4, len(body), vid, self.toast_relid)

    def insert(self, values, force_external=None, natts=None):
        """values as in form_tuple; varlena values given as ("inline", body) are moved out of line, widest first, while the tuple
        is wider than TOAST_TUPLE_THRESHOLD (heap_toast_insert_or_update for `external` / `extended` columns without compression)"""
        vals = list(values)
        for i in (force_external or ()):
            vals[i] = self.toast_value(vals[i][1])
        while len(form_tuple(self.attrs, vals, natts=natts)) > TOAST_TUPLE_THRESHOLD:
            cand = [(len(v[1]), i) for i, v in enumerate(vals) if isinstance(v, tuple) and v[0] == "inline" and self.attrs[i][0] == -1]
            if not cand:
                break
            _, i = max(cand)
            vals[i] = self.toast_value(vals[i][1])
        t = form_tuple(self.attrs, vals, natts=natts)
        blk, off = self.heap.add_item(t)
        return (blk << 16) | off


def parse_tuple(attrs, tup):
    """heap_deform_tuple in miniature -> per attribute None | (start, size) of the stored bytes inside `tup`"""
    natts = struct.unpack_from("<H", tup, 18)[0] & 0x07FF
    infomask = struct.unpack_from("<H", tup, 20)[0]
    hoff = tup[22]
    out, off = [], hoff
    for i, (attlen, attalign) in enumerate(attrs):
        if i >= natts or ((infomask & HEAP_HASNULL) and not (tup[23 + (i >> 3)] >> (i & 7)) & 1):
            out.append(None)
            continue
        if attlen == -1:
            if tup[off] == 0:
                off = hoff + ((off - hoff + ALIGN[attalign] - 1) // ALIGN[attalign]) * ALIGN[attalign]
            b0 = tup[off]
            if b0 == 0x01:
                size = 2 + 16
            elif b0 & 1:
                size = (b0 >> 1) & 0x7F
            else:
                size = (struct.unpack_from("<I", tup, off)[0] >> 2) & 0x3FFFFFFF
        else:
            off = hoff + ((off - hoff + ALIGN[attalign] - 1) // ALIGN[attalign]) * ALIGN[attalign]
            size = attlen
        out.append((off, size))
        off += size
    return out


def read_vector_column(table, tids, vec_att, dim):
    """independent (pure Python) read-back: for each heap TID the vector of attribute `vec_att` (0-based) or None"""
    chunks = {}
    for blk in range(len(table.toast.pages)):
        for off in range(1, table.toast.max_offset(blk) + 1):
            lp = struct.unpack_from("<I", table.toast.pages[blk], SIZE_OF_PAGE_HEADER + 4 * (off - 1))[0]
            if (lp >> 15) & 3 != LP_NORMAL:
                continue
            t = bytes(table.toast.pages[blk][lp & 0x7FFF:(lp & 0x7FFF) + (lp >> 17)])
            spans = parse_tuple(TOAST_ATTRS, t)
            vid = struct.unpack_from("<I", t, spans[0][0])[0]
            seq = struct.unpack_from("<i", t, spans[1][0])[0]
            s, l = spans[2]
            body = t[s + 1:s + l] if t[s] & 1 else t[s + 4:s + l]
            chunks.setdefault(vid, {})[seq] = body
    out = []
    for tid in tids:
        blk, off = int(tid) >> 16, int(tid) & 0xFFFF
        res = None
        hops = 0
        while blk < len(table.heap.pages) and 1 <= off <= table.heap.max_offset(blk) and hops < 4:
            lp = struct.unpack_from("<I", table.heap.pages[blk], SIZE_OF_PAGE_HEADER + 4 * (off - 1))[0]
            fl = (lp >> 15) & 3
            if fl == LP_REDIRECT:
                off = lp & 0x7FFF
                hops += 1
                continue
            if fl == LP_NORMAL:
                t = bytes(table.heap.pages[blk][lp & 0x7FFF:(lp & 0x7FFF) + (lp >> 17)])
                span = parse_tuple(table.attrs, t)[vec_att]
                if span is not None:
                    s, l = span
                    if t[s] == 0x01:
                        raw, ext, vid, _ = struct.unpack_from("<iIII", t, s + 2)
                        body = b"".join(chunks[vid][k] for k in sorted(chunks[vid]))
                        assert len(body) == ext == raw - 4
                    elif t[s] & 1:
                        body = t[s + 1:s + l]
                    else:
                        body = t[s + 4:s + l]
                    d = struct.unpack_from("<h", body, 0)[0]
                    assert d == dim
                    res = np.frombuffer(body, "<f4", d, 4).copy()
            break
        out.append(res)
    return out
