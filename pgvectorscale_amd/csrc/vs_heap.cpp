// vs_heap.cpp — the heap's vector column, staged in bulk (host-only; the last array of vs_index_host: `vecs`).
//
// The reference computes every full-precision distance of the rescore window from the HEAP tuple an index node points at:
// table_index_fetch_tuple under the scan's snapshot (UT/table_slot.rs:19-42), slot_getattr of the indexed column and
// pg_detoast_datum_copy (AM/pg_vector.rs:125-135), then distance_fn (AM/sbq/storage.rs:304-328) — one buffer pin, one tuple
// deform and, for any vector wider than ~500 dimensions, one TOAST index scan PER CANDIDATE.  Here the column is read once:
// the blocks of the heap's main fork stream past (vs_heap_add), the tuple each node's heap TID names is deformed up to the vector
// attribute exactly as heap_deform_tuple would (null bitmap, attalign padding, 1-byte / 4-byte varlena headers; LP_REDIRECT
// line pointers of pruned HOT chains are followed), an inline vector is copied to row `node` of the output at once, an external
// one (varatt_external, VARTAG_ONDISK) is noted by its TOAST value id; then the blocks of the TOAST relation stream past
// (vs_heap_toast_add) and every chunk (chunk_id, chunk_seq, chunk_data) of a noted value lands in its row at
// chunk_seq * TOAST_MAX_CHUNK_SIZE.  Nothing is kept of the pages; memory is the output array + 24 bytes per node.
//
// Restated from PostgreSQL 13-17 (little endian): storage/bufpage.h + itemid.h (page header, line pointers),
// access/htup_details.h (HeapTupleHeaderData, t_hoff, HEAP_HASNULL, att_isnull, att_align_pointer), varatt.h / postgres.h
// (VARATT_IS_1B / _1B_E / _4B_U / _4B_C, VARSIZE_*, VARTAG_ONDISK, varatt_external), access/heaptoast.h
// (TOAST_MAX_CHUNK_SIZE), utils/pg_lzcompress (pglz_decompress, for a column whose storage was altered to `extended`);
// pgvector's vector.h (int16 dim, int16 unused, float4 x[]).  Visibility is NOT decided here: the row a TID names is exported
// whatever its xmin / xmax say (a HOT-updated tuple keeps its indexed column; what a snapshot may see is the visibility
// mask's business, vs_index_set_visibility); a dead or unused line pointer leaves its row zero and is reported.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vsgpu.h"

void vs_set_error(const char* fmt, ...);

namespace {

constexpr uint32_t kPageHeader = 24;
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    vs_set_error("%s", buf);
    return VS_ERR_INVALID;
}

uint32_t align_of(char c) { return c == 'd' ? 8u : (c == 'i' ? 4u : (c == 's' ? 2u : 1u)); }

struct Page {
    const uint8_t* p;
    uint32_t size, lower, upper, special, nitems;
    bool is_new;
};
bool view(const uint8_t* p, uint32_t page_size, Page& v, std::string& err) {
    v.p = p;
    v.size = page_size;
    v.lower = rd16(p + 12);
    v.upper = rd16(p + 14);
    v.special = rd16(p + 16);
    v.is_new = v.upper == 0;  // PageIsNew
    v.nitems = 0;
    if (v.is_new) return true;
    char buf[160];
    if ((uint32_t)(rd16(p + 18) & 0xFF00) != page_size) {
        snprintf(buf, sizeof buf, "pd_pagesize_version 0x%04x does not say %u-byte pages", rd16(p + 18), page_size);
        err = buf;
        return false;
    }
    if (!(v.lower >= kPageHeader && v.lower <= v.upper && v.upper <= v.special && v.special <= page_size)) {
        snprintf(buf, sizeof buf, "inconsistent page header (pd_lower %u, pd_upper %u, pd_special %u)", v.lower, v.upper, v.special);
        err = buf;
        return false;
    }
    v.nitems = (v.lower - kPageHeader) / 4;
    return true;
}
enum { LP_UNUSED = 0, LP_NORMAL = 1, LP_REDIRECT = 2, LP_DEAD = 3 };

// one attribute located inside a tuple
struct Span {
    const uint8_t* at = nullptr;
    uint32_t size = 0;
    bool is_null = false;
};

// heap_deform_tuple up to attribute `want` (0-based) of a tuple of `len` bytes.  false + err on a malformed tuple.
bool locate_attr(const uint8_t* t, uint32_t len, const vs_heap_attr* attrs, uint32_t natts_desc, uint32_t want, Span& out,
                 std::string& err) {
    char buf[200];
    if (len < 23) {
        err = "tuple shorter than its header";
        return false;
    }
    const uint32_t natts = rd16(t + 18) & 0x07FFu;  // HeapTupleHeaderGetNatts
    const uint16_t infomask = rd16(t + 20);
    const uint32_t hoff = t[22];
    const bool hasnull = (infomask & 0x0001) != 0;
    if (hoff < 23 + (hasnull ? (natts + 7) / 8 : 0) || hoff > len || (hoff & 7)) {
        snprintf(buf, sizeof buf, "t_hoff %u does not fit a %u-byte tuple of %u attributes", hoff, len, natts);
        err = buf;
        return false;
    }
    if (want >= natts) {  // a column added after the row was written: NULL (or its missing-value default; a vector has none here)
        out.is_null = true;
        return true;
    }
    uint32_t off = 0;  // relative to t + hoff (which is MAXALIGNed, so alignment of `off` is alignment in memory)
    const uint8_t* tp = t + hoff;
    const uint32_t dl = len - hoff;
    for (uint32_t a = 0; a <= want; ++a) {
        if (a >= natts_desc) {
            err = "the tuple descriptor has fewer attributes than the vector's attribute number";
            return false;
        }
        if (hasnull && !((t[23 + (a >> 3)] >> (a & 7)) & 1)) {  // att_isnull
            if (a == want) {
                out.is_null = true;
                return true;
            }
            continue;
        }
        const int attlen = attrs[a].attlen;
        const uint32_t al = align_of(attrs[a].attalign);
        uint32_t size;
        if (attlen == -1) {
            if (off >= dl) {
                err = "attribute starts past the end of the tuple";
                return false;
            }
            if (tp[off] == 0) off = (off + al - 1) / al * al;  // att_align_pointer: a pad byte, so a 4-byte header follows at attalign
            if (off >= dl) {
                err = "attribute starts past the end of the tuple";
                return false;
            }
            const uint8_t b0 = tp[off];
            if (b0 == 0x01) {  // VARATT_IS_1B_E: external, VARTAG_SIZE(tag)
                if (off + 2 > dl) {
                    err = "truncated external varlena";
                    return false;
                }
                const uint8_t tag = tp[off + 1];
                if (tag != 18) {  // VARTAG_ONDISK; indirect / expanded pointers never reach disk
                    snprintf(buf, sizeof buf, "varlena tag %u on disk", tag);
                    err = buf;
                    return false;
                }
                size = 2 + 16;
            } else if (b0 & 0x01) {
                size = (b0 >> 1) & 0x7Fu;  // VARSIZE_1B (includes the header byte)
                if (size == 0) {
                    err = "1-byte varlena header of size 0";
                    return false;
                }
            } else {
                if (off + 4 > dl) {
                    err = "truncated varlena header";
                    return false;
                }
                size = (rd32(tp + off) >> 2) & 0x3FFFFFFFu;  // VARSIZE_4B
                if (size < 4) {
                    err = "4-byte varlena of fewer than 4 bytes";
                    return false;
                }
            }
        } else if (attlen == -2) {  // cstring
            off = (off + al - 1) / al * al;
            uint32_t e = off;
            while (e < dl && tp[e]) ++e;
            if (e >= dl) {
                err = "unterminated cstring attribute";
                return false;
            }
            size = e - off + 1;
        } else if (attlen > 0) {
            off = (off + al - 1) / al * al;
            size = (uint32_t)attlen;
        } else {
            err = "attribute length 0 / below -2 in the tuple descriptor";
            return false;
        }
        if ((uint64_t)off + size > dl) {
            snprintf(buf, sizeof buf, "attribute %u (%u bytes at +%u) runs past the tuple's %u data bytes", a + 1, size, off, dl);
            err = buf;
            return false;
        }
        if (a == want) {
            out.at = tp + off;
            out.size = size;
            out.is_null = false;
            return true;
        }
        off += size;
    }
    return true;
}

// pglz_decompress (common/pg_lzcompress.c): control byte, 8 items each: literal byte, or a (len 3..18+, offset 1..4095) match
bool pglz_decompress(const uint8_t* src, uint32_t slen, uint8_t* dst, uint32_t rawsize) {
    const uint8_t* sp = src;
    const uint8_t* send = src + slen;
    uint8_t* dp = dst;
    uint8_t* dend = dst + rawsize;
    while (sp < send && dp < dend) {
        uint8_t ctrl = *sp++;
        for (int c = 0; c < 8 && sp < send && dp < dend; ++c, ctrl >>= 1) {
            if (ctrl & 1) {
                if (sp + 2 > send) return false;
                int32_t len = (sp[0] & 0x0f) + 3;
                const int32_t off = ((sp[0] & 0xf0) << 4) | sp[1];
                sp += 2;
                if (len == 18) {
                    if (sp >= send) return false;
                    len += *sp++;
                }
                if (off == 0 || off > dp - dst) return false;
                len = (int32_t)std::min<int64_t>(len, dend - dp);
                for (int32_t i = 0; i < len; ++i, ++dp) *dp = dp[-off];  // (overlapping copies are the point of the format)
            } else {
                *dp++ = *sp++;
            }
        }
    }
    return dp == dend && sp == send;
}

}  // namespace

enum : uint8_t { ST_PENDING = 0, ST_DONE = 1, ST_WAIT_TOAST = 2, ST_NULL = 3, ST_DEAD_LP = 4, ST_NOT_FOUND = 5, ST_INCOMPLETE = 6 };

struct ToastRef {
    uint32_t valueid;
    uint32_t node;
    uint32_t extsize;   // bytes stored in the TOAST relation
    uint32_t rawsize;   // bytes of the datum's data portion once decompressed (== extsize when not compressed)
    uint32_t got = 0;   // chunk bytes received so far
    uint32_t cmethod = 0;  // 0 pglz, 1 lz4 (only meaningful when extsize < rawsize)
};

struct vs_heap {
    uint32_t page_size = VS_BLCKSZ, vec_att = 0, dim = 0, n = 0, stride = 0;
    std::vector<vs_heap_attr> attrs;
    float* out = nullptr;
    std::vector<std::pair<uint64_t, uint32_t>> by_tid;  // (heap tid, node), sorted: the heap streams past in block order
    size_t cursor = 0;
    std::vector<uint8_t> state;
    std::vector<ToastRef> refs;       // sorted by valueid once the heap pass is over
    std::vector<std::vector<uint8_t>> packed;  // compressed values are assembled here (rare), indexed like refs
    bool heap_done = false, toast_sorted = false;
    uint32_t next_heap_block = 0, next_toast_block = 0, chunk = 0;
    vs_heap_info info{};
};

static int take_inline(vs_heap* h, uint32_t node, const uint8_t* body, uint32_t blen, uint32_t blk, uint32_t off) {
    if (blen < 4) return fail("heap (%u,%u): vector datum of %u data bytes", blk, off, blen);
    const int dim = (int16_t)rd16(body);
    if ((uint32_t)dim != h->dim) return fail("heap (%u,%u): vector of %d dimensions, the index has %u", blk, off, dim, h->dim);
    if (blen != 4 + 4 * h->dim) return fail("heap (%u,%u): %u data bytes for %u dimensions", blk, off, blen, h->dim);
    memcpy(h->out + (size_t)node * h->stride, body + 4, (size_t)h->dim * 4);
    return VS_OK;
}

extern "C" {

int vs_heap_open(uint32_t page_size, const vs_heap_attr* attrs, uint32_t natts, uint32_t vector_attno, uint32_t dim,
                 const uint64_t* heap_tids, uint32_t n, float* out_vecs, uint32_t out_stride, vs_heap** out) {
    if (!attrs || !out || natts == 0 || vector_attno < 1 || vector_attno > natts || dim == 0 || dim > 16000 || (n && (!heap_tids || !out_vecs)) ||
        out_stride < dim)
        return fail("vs_heap_open: bad arguments");
    if (page_size < 512 || page_size > 32768 || (page_size & (page_size - 1))) return fail("vs_heap_open: page size %u", page_size);
    if (attrs[vector_attno - 1].attlen != -1) return fail("vs_heap_open: attribute %u is not a varlena (pgvector's vector is)", vector_attno);
    *out = nullptr;
    vs_heap* h = nullptr;
    try {
        h = new vs_heap();
        h->page_size = page_size;
        h->vec_att = vector_attno - 1;
        h->dim = dim;
        h->n = n;
        h->stride = out_stride;
        h->attrs.assign(attrs, attrs + natts);
        h->out = out_vecs;
        h->state.assign(n, ST_PENDING);
        h->by_tid.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            // deleted tuples (offset 0 = InvalidOffsetNumber, AM/scan.rs:231-234) have no heap row to fetch
            if ((heap_tids[i] & 0xFFFFull) == 0) h->state[i] = ST_NULL, h->info.n_deleted++;
            else h->by_tid.emplace_back(heap_tids[i], i);
        }
        std::sort(h->by_tid.begin(), h->by_tid.end());
        // EXTERN_TUPLE_MAX_SIZE - MAXALIGN(SizeofHeapTupleHeader) - sizeof(Oid) - sizeof(int32) - VARHDRSZ (access/heaptoast.h)
        const uint32_t per_tuple = ((page_size - ((kPageHeader + 4 * 4 + 7) & ~7u)) / 4) & ~7u;
        h->chunk = per_tuple - 24 - 4 - 4 - 4;
    } catch (const std::bad_alloc&) {
        delete h;
        vs_set_error("vs_heap_open: out of host memory");
        return VS_ERR_OOM;
    }
    for (uint32_t i = 0; i < n; ++i) memset(out_vecs + (size_t)i * out_stride, 0, (size_t)dim * 4);
    *out = h;
    return VS_OK;
}

int vs_heap_add(vs_heap* h, uint32_t first_block, const void* pages, uint32_t n_blocks) {
    if (!h || (!pages && n_blocks)) return fail("vs_heap_add: null argument");
    if (h->heap_done) return fail("vs_heap_add: the TOAST pass has begun");
    if (first_block != h->next_heap_block) return fail("vs_heap_add: expected block %u, got %u (blocks come in order)", h->next_heap_block, first_block);
    const uint8_t* base = static_cast<const uint8_t*>(pages);
    try {
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const uint32_t blk = first_block + b;
            // the nodes whose tuple lives on this block
            size_t lo = h->cursor;
            while (lo < h->by_tid.size() && (h->by_tid[lo].first >> 16) < blk) {
                h->state[h->by_tid[lo].second] = ST_NOT_FOUND;  // (cannot happen: blocks arrive in order)
                ++lo;
            }
            size_t hi = lo;
            while (hi < h->by_tid.size() && (h->by_tid[hi].first >> 16) == blk) ++hi;
            h->cursor = hi;
            if (lo == hi) continue;
            Page pg;
            std::string err;
            if (!view(base + (size_t)b * h->page_size, h->page_size, pg, err)) return fail("heap block %u: %s", blk, err.c_str());
            for (size_t e = lo; e < hi; ++e) {
                const uint32_t node = h->by_tid[e].second;
                uint32_t off = (uint32_t)(h->by_tid[e].first & 0xFFFF);
                uint8_t st = ST_NOT_FOUND;
                for (int hop = 0; hop < 8; ++hop) {
                    if (pg.is_new || off < 1 || off > pg.nitems) break;
                    const uint32_t lp = rd32(pg.p + kPageHeader + 4 * (off - 1));
                    const uint32_t lp_off = lp & 0x7FFF, fl = (lp >> 15) & 3, lp_len = lp >> 17;
                    if (fl == LP_REDIRECT) {  // the root of a pruned HOT chain: the live member is where it points
                        off = lp_off;
                        continue;
                    }
                    if (fl != LP_NORMAL) {
                        st = ST_DEAD_LP;
                        break;
                    }
                    if (lp_off < pg.upper || lp_off + lp_len > pg.special || lp_len < 23)
                        return fail("heap (%u,%u): item (off %u, len %u) lies outside pd_upper..pd_special", blk, off, lp_off, lp_len);
                    const uint8_t* t = pg.p + lp_off;
                    Span sp;
                    if (!locate_attr(t, lp_len, h->attrs.data(), (uint32_t)h->attrs.size(), h->vec_att, sp, err))
                        return fail("heap (%u,%u): %s", blk, off, err.c_str());
                    if (sp.is_null) {
                        st = ST_NULL;
                        break;
                    }
                    const uint8_t b0 = sp.at[0];
                    if (b0 == 0x01) {  // external: note the TOAST value
                        ToastRef r{};
                        const int32_t va_rawsize = rdi32(sp.at + 2);
                        const uint32_t va_extinfo = rd32(sp.at + 6);
                        r.valueid = rd32(sp.at + 10);
                        r.node = node;
                        r.extsize = va_extinfo & 0x3FFFFFFFu;
                        r.cmethod = va_extinfo >> 30;
                        if (va_rawsize < 4) return fail("heap (%u,%u): va_rawsize %d", blk, off, va_rawsize);
                        r.rawsize = (uint32_t)va_rawsize - 4;
                        if (r.rawsize != 4 + 4 * h->dim)
                            return fail("heap (%u,%u): external vector of %u data bytes, %u dimensions need %u", blk, off, r.rawsize, h->dim,
                                        4 + 4 * h->dim);
                        if (r.extsize > r.rawsize) return fail("heap (%u,%u): external size %u above raw size %u", blk, off, r.extsize, r.rawsize);
                        if (r.extsize < r.rawsize && r.cmethod != 0)
                            return fail("heap (%u,%u): vector compressed with method %u (only pglz is decoded here; pgvector's own storage "
                                        "is `external`, i.e. never compressed)", blk, off, r.cmethod);
                        h->refs.push_back(r);
                        st = ST_WAIT_TOAST;
                    } else if (b0 & 0x01) {  // short header: body unaligned, memcpy takes care
                        int rc = take_inline(h, node, sp.at + 1, sp.size - 1, blk, off);
                        if (rc != VS_OK) return rc;
                        st = ST_DONE;
                        h->info.n_inline++;
                    } else if ((rd32(sp.at) & 0x03) == 0x02) {  // VARATT_IS_4B_C: compressed in line (tcinfo: rawsize | method << 30)
                        if (sp.size < 8) return fail("heap (%u,%u): truncated compressed varlena", blk, off);
                        const uint32_t tc = rd32(sp.at + 4);
                        const uint32_t raw = tc & 0x3FFFFFFFu;
                        if ((tc >> 30) != 0) return fail("heap (%u,%u): vector compressed with method %u (only pglz is decoded)", blk, off, tc >> 30);
                        if (raw != 4 + 4 * h->dim) return fail("heap (%u,%u): compressed vector of %u data bytes", blk, off, raw);
                        std::vector<uint8_t> tmp(raw);
                        if (!pglz_decompress(sp.at + 8, sp.size - 8, tmp.data(), raw)) return fail("heap (%u,%u): corrupt pglz data", blk, off);
                        int rc = take_inline(h, node, tmp.data(), raw, blk, off);
                        if (rc != VS_OK) return rc;
                        st = ST_DONE;
                        h->info.n_inline++;
                    } else {
                        int rc = take_inline(h, node, sp.at + 4, sp.size - 4, blk, off);
                        if (rc != VS_OK) return rc;
                        st = ST_DONE;
                        h->info.n_inline++;
                    }
                    break;
                }
                h->state[node] = st;
            }
        }
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_heap_add: out of host memory");
        return VS_ERR_OOM;
    }
    h->next_heap_block = first_block + n_blocks;
    return VS_OK;
}

int vs_heap_toast_add(vs_heap* h, uint32_t first_block, const void* pages, uint32_t n_blocks) {
    if (!h || (!pages && n_blocks)) return fail("vs_heap_toast_add: null argument");
    if (first_block != h->next_toast_block)
        return fail("vs_heap_toast_add: expected block %u, got %u (blocks come in order)", h->next_toast_block, first_block);
    static const vs_heap_attr toast_attrs[3] = {{4, 'i'}, {4, 'i'}, {-1, 'i'}};  // chunk_id oid, chunk_seq int4, chunk_data bytea
    try {
        if (!h->toast_sorted) {
            h->heap_done = true;
            for (size_t e = h->cursor; e < h->by_tid.size(); ++e) h->state[h->by_tid[e].second] = ST_NOT_FOUND;  // blocks never seen
            h->cursor = h->by_tid.size();
            std::sort(h->refs.begin(), h->refs.end(), [](const ToastRef& a, const ToastRef& b) { return a.valueid < b.valueid; });
            for (size_t i = 1; i < h->refs.size(); ++i)
                if (h->refs[i].valueid == h->refs[i - 1].valueid)
                    return fail("TOAST value %u is referenced by two heap tuples (nodes %u and %u)", h->refs[i].valueid, h->refs[i - 1].node,
                                h->refs[i].node);
            h->packed.resize(h->refs.size());
            h->toast_sorted = true;
        }
        const uint8_t* base = static_cast<const uint8_t*>(pages);
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const uint32_t blk = first_block + b;
            Page pg;
            std::string err;
            if (!view(base + (size_t)b * h->page_size, h->page_size, pg, err)) return fail("TOAST block %u: %s", blk, err.c_str());
            for (uint32_t off = 1; off <= pg.nitems; ++off) {
                const uint32_t lp = rd32(pg.p + kPageHeader + 4 * (off - 1));
                const uint32_t lp_off = lp & 0x7FFF, fl = (lp >> 15) & 3, lp_len = lp >> 17;
                if (fl != LP_NORMAL) continue;
                if (lp_off < pg.upper || lp_off + lp_len > pg.special || lp_len < 23)
                    return fail("TOAST (%u,%u): item lies outside pd_upper..pd_special", blk, off);
                const uint8_t* t = pg.p + lp_off;
                Span id, seq, data;
                if (!locate_attr(t, lp_len, toast_attrs, 3, 0, id, err) || !locate_attr(t, lp_len, toast_attrs, 3, 1, seq, err) ||
                    !locate_attr(t, lp_len, toast_attrs, 3, 2, data, err))
                    return fail("TOAST (%u,%u): %s", blk, off, err.c_str());
                if (id.is_null || seq.is_null || data.is_null) return fail("TOAST (%u,%u): NULL in a chunk row", blk, off);
                const uint32_t vid = rd32(id.at);
                auto it = std::lower_bound(h->refs.begin(), h->refs.end(), vid, [](const ToastRef& r, uint32_t v) { return r.valueid < v; });
                if (it == h->refs.end() || it->valueid != vid) continue;  // a value of another column / row
                h->info.n_chunks++;
                const int32_t s = rdi32(seq.at);
                const uint8_t* body;
                uint32_t blen;
                if (data.at[0] == 0x01) return fail("TOAST (%u,%u): external chunk data", blk, off);
                if (data.at[0] & 1) {
                    body = data.at + 1;
                    blen = data.size - 1;
                } else {
                    if ((rd32(data.at) & 3) != 0) return fail("TOAST (%u,%u): compressed chunk data", blk, off);
                    body = data.at + 4;
                    blen = data.size - 4;
                }
                const uint64_t at = (uint64_t)(s < 0 ? 0 : s) * h->chunk;
                if (s < 0 || at + blen > it->extsize || (blen != h->chunk && at + blen != it->extsize))
                    return fail("TOAST (%u,%u): chunk %d of value %u holds %u bytes at %llu of %u", blk, off, s, vid, blen,
                                (unsigned long long)at, it->extsize);
                const size_t ri = (size_t)(it - h->refs.begin());
                if (it->extsize == it->rawsize) {  // the usual case: the chunk's floats go straight to the node's row
                    // value bytes: [0,2) dim, [2,4) unused, [4, ..) the floats
                    uint8_t* row = reinterpret_cast<uint8_t*>(h->out + (size_t)it->node * h->stride);
                    uint64_t vb = at, n_ = blen;
                    const uint8_t* src = body;
                    if (vb < 4) {
                        if (vb == 0 && n_ >= 2 && (uint32_t)(int16_t)rd16(src) != h->dim)
                            return fail("TOAST value %u: vector of %d dimensions, the index has %u", vid, (int16_t)rd16(src), h->dim);
                        const uint64_t skip = std::min<uint64_t>(4 - vb, n_);
                        src += skip;
                        n_ -= skip;
                        vb += skip;
                    }
                    if (n_) memcpy(row + (vb - 4), src, n_);
                } else {
                    std::vector<uint8_t>& pk = h->packed[ri];
                    if (pk.empty()) pk.resize(it->extsize);
                    memcpy(pk.data() + at, body, blen);
                }
                it->got += blen;
                if (it->got > it->extsize) return fail("TOAST value %u: more chunk bytes than its %u", vid, it->extsize);
                if (it->got == it->extsize) {
                    if (it->extsize != it->rawsize) {  // varattrib compressed: 4-byte tcinfo header is NOT part of the toasted bytes' count
                        std::vector<uint8_t>& pk = h->packed[ri];
                        // (the toasted form of a compressed datum starts with the 4-byte rawsize/method word, toast_compress_header)
                        if (pk.size() < 4) return fail("TOAST value %u: compressed datum of %zu bytes", vid, pk.size());
                        std::vector<uint8_t> raw(it->rawsize);
                        if (!pglz_decompress(pk.data() + 4, (uint32_t)pk.size() - 4, raw.data(), it->rawsize))
                            return fail("TOAST value %u: corrupt pglz data", vid);
                        int rc = take_inline(h, it->node, raw.data(), it->rawsize, blk, off);
                        if (rc != VS_OK) return rc;
                        std::vector<uint8_t>().swap(pk);
                    }
                    h->state[it->node] = ST_DONE;
                    h->info.n_external++;
                }
            }
        }
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_heap_toast_add: out of host memory");
        return VS_ERR_OOM;
    }
    h->next_toast_block = first_block + n_blocks;
    return VS_OK;
}

int vs_heap_finish(vs_heap* h, vs_heap_info* info, uint8_t* found) {
    if (!h) return fail("vs_heap_finish: null reader");
    if (!h->heap_done) {
        h->heap_done = true;
        for (size_t e = h->cursor; e < h->by_tid.size(); ++e) h->state[h->by_tid[e].second] = ST_NOT_FOUND;
        h->cursor = h->by_tid.size();
    }
    vs_heap_info& I = h->info;
    I.n_nodes = h->n;
    I.heap_blocks = h->next_heap_block;
    I.toast_blocks = h->next_toast_block;
    I.n_null = I.n_dead_line_pointer = I.n_not_found = I.n_toast_incomplete = 0;
    for (uint32_t i = 0; i < h->n; ++i) {
        uint8_t st = h->state[i];
        if (st == ST_WAIT_TOAST) st = ST_INCOMPLETE;
        if (found) found[i] = st == ST_DONE ? 1 : 0;
        if (st == ST_NULL) I.n_null++;
        else if (st == ST_DEAD_LP) I.n_dead_line_pointer++;
        else if (st == ST_NOT_FOUND || st == ST_PENDING) I.n_not_found++;
        else if (st == ST_INCOMPLETE) I.n_toast_incomplete++;
    }
    I.n_null -= I.n_deleted;  // (deleted index tuples were parked in the NULL state at open)
    if (info) *info = I;
    return VS_OK;
}

void vs_heap_close(vs_heap* h) { delete h; }

}  // extern "C"
