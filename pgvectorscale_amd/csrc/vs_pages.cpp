// vs_pages.cpp — index relation pages -> the flat arrays of vs_index_host (host side of the staging path).
//
// The reference keeps every index node as one line-pointer item on an 8 KB PostgreSQL page and reaches it through
// the buffer manager (ItemPointer::read_bytes, UT/mod.rs:152-155 -> ReadablePage::get_item_unchecked,
// UT/page.rs:270-283 -> rkyv::archived_root, pgvectorscale_derive/src/lib.rs:35-40).  This file reads the same bytes
// in bulk: blocks of the main fork are handed over in order (vs_pages_add), every SbqNode item is decoded into the
// code / neighbor / heap-tid / label arrays, and neighbor ItemPointers become dense node ids
// (id = items on earlier SbqNode pages + offset - 1; the reference never frees a node's line pointer, vacuum only
// invalidates heap_item_pointer, AM/sbq/node.rs:134-155).  vs_index_upload then streams the arrays to HBM through the
// pinned ring.  Host code only: nothing here needs a device.
//
// Byte layouts restated (no PostgreSQL / rkyv headers exist in this image):
//   PageHeaderData (storage/bufpage.h): pd_lsn 8 | pd_checksum 2 | pd_flags 2 | pd_lower 2 | pd_upper 2 | pd_special 2 |
//     pd_pagesize_version 2 | pd_prune_xid 4 | ItemIdData pd_linp[] (4 B each: lp_off:15, lp_flags:2, lp_len:15)
//   special area: TsvPageOpaqueData {u8 page_type, u8 reserved, u16 page_id = 0xAE24} (UT/page.rs:24-76)
//   rkyv 0.7 (size_32, little endian): root object at the END of the item (len - size_of::<Archived<T>>()),
//     ArchivedVec<T> = {i32 offset relative to the field's own address, u32 len}
//   ArchivedItemPointer {u32 block_number, u16 offset, 2 B pad} (UT/mod.rs:17-23)
//   ArchivedClassicSbqNode / ArchivedLabeledSbqNode (AM/sbq/node.rs:26-42): four 8-byte fields in declaration order —
//     the one layout fact that is NOT pinned by anything in the reference (the archived struct is repr(Rust)); the
//     caller may pass the real offsets (core::mem::offset_of! on the Rust side) through vs_node_layout.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/vsgpu.h"

void vs_set_error(const char* fmt, ...);

namespace {

constexpr uint32_t kPageHeaderSize = 24;       // offsetof(PageHeaderData, pd_linp)
constexpr uint16_t kTsvPageId = 0xAE24;        // UT/page.rs:24
constexpr uint32_t kInvalidBlock = 0xFFFFFFFFu;  // InvalidBlockNumber
constexpr uint32_t kTsvMagic = 768756476u;     // AM/meta_page.rs:22
constexpr uint32_t kChainHeader = 8;           // size_of::<ArchivedChainItemHeader>() (UT/chain.rs:27-33)
constexpr uint64_t kNoNeighbor = ~0ull;

inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct PageView {
    const uint8_t* p = nullptr;
    uint32_t lower = 0, upper = 0, special = 0;
    uint32_t n_items = 0;  // PageGetMaxOffsetNumber
    int type = -1;         // PageType, -1 = new (all-zero) page
};

// Header + special-area checks of one page; `err` receives the reason when false is returned.
bool view_page(const uint8_t* p, uint32_t page_size, PageView& v, std::string& err) {
    v.p = p;
    v.lower = rd16(p + 12);
    v.upper = rd16(p + 14);
    v.special = rd16(p + 16);
    const uint16_t psv = rd16(p + 18);
    if (v.upper == 0) {  // PageIsNew(): a block the relation was extended by but that was never initialised
        v.type = -1;
        v.n_items = 0;
        return true;
    }
    char buf[160];
    if ((uint32_t)(psv & 0xFF00) != page_size) {
        snprintf(buf, sizeof buf, "pd_pagesize_version 0x%04x does not say %u-byte pages", psv, page_size);
        err = buf;
        return false;
    }
    if (!(v.lower >= kPageHeaderSize && v.lower <= v.upper && v.upper <= v.special && v.special + 4 <= page_size)) {
        snprintf(buf, sizeof buf, "inconsistent page header (pd_lower %u, pd_upper %u, pd_special %u)", v.lower, v.upper,
                 v.special);
        err = buf;
        return false;
    }
    const uint8_t* sp = p + v.special;  // TsvPageOpaqueData::read_from_page + verify (UT/page.rs:92-103)
    if (rd16(sp + 2) != kTsvPageId) {
        snprintf(buf, sizeof buf, "page_id 0x%04x is not the diskann magic 0x%04x", rd16(sp + 2), kTsvPageId);
        err = buf;
        return false;
    }
    if (sp[0] > VS_PAGE_META) {
        snprintf(buf, sizeof buf, "unknown PageType number %u", sp[0]);
        err = buf;
        return false;
    }
    v.type = sp[0];
    v.n_items = (v.lower - kPageHeaderSize) / 4;
    return true;
}

// PageGetItemId + PageGetItem (UT/ports.rs:56-77) with the bounds the reference leaves to PostgreSQL
bool get_item(const PageView& v, uint32_t offset, const uint8_t*& data, uint32_t& len, std::string& err) {
    char buf[160];
    if (offset < 1 || offset > v.n_items) {
        snprintf(buf, sizeof buf, "offset %u outside 1..%u", offset, v.n_items);
        err = buf;
        return false;
    }
    const uint32_t lp = rd32(v.p + kPageHeaderSize + 4 * (offset - 1));
    const uint32_t lp_off = lp & 0x7FFF, lp_flags = (lp >> 15) & 3, lp_len = lp >> 17;
    if (lp_flags != 1 /* LP_NORMAL */ || lp_len == 0) {
        snprintf(buf, sizeof buf, "line pointer %u is not LP_NORMAL (flags %u, len %u)", offset, lp_flags, lp_len);
        err = buf;
        return false;
    }
    if (lp_off < v.upper || lp_off + lp_len > v.special) {
        snprintf(buf, sizeof buf, "item %u (off %u, len %u) lies outside pd_upper..pd_special", offset, lp_off, lp_len);
        err = buf;
        return false;
    }
    data = v.p + lp_off;
    len = lp_len;
    return true;
}

// ArchivedVec<T> at byte position `field` of an item of `len` bytes: start of the elements and their count
bool archived_vec(const uint8_t* item, uint32_t len, uint32_t field, uint32_t elem, const uint8_t*& first, uint32_t& count,
                  std::string& err) {
    const int64_t target = (int64_t)field + rdi32(item + field);
    count = rd32(item + field + 4);
    if (count == 0) {
        first = item;
        return true;
    }
    if (target < 0 || (uint64_t)target + (uint64_t)count * elem > len) {
        char buf[160];
        snprintf(buf, sizeof buf, "ArchivedVec at +%u points outside the item (target %lld, %u x %u B, item %u B)", field,
                 (long long)target, count, elem, len);
        err = buf;
        return false;
    }
    first = item + target;
    return true;
}

template <class F>
void parallel_for(uint32_t threads, uint64_t n, F&& fn) {
    if (threads <= 1 || n < 2) {
        for (uint64_t i = 0; i < n; i++) fn(i);
        return;
    }
    const uint32_t t = (uint32_t)std::min<uint64_t>(threads, n);
    std::atomic<uint64_t> next{0};
    const uint64_t grain = std::max<uint64_t>(1, n / (t * 8ull));
    std::vector<std::thread> pool;
    pool.reserve(t);
    for (uint32_t k = 0; k < t; k++)
        pool.emplace_back([&] {
            for (;;) {
                const uint64_t b = next.fetch_add(grain);
                if (b >= n) return;
                const uint64_t e = std::min(n, b + grain);
                for (uint64_t i = b; i < e; i++) fn(i);
            }
        });
    for (auto& th : pool) th.join();
}

}  // namespace

struct vs_pages {
    uint32_t page_size = VS_BLCKSZ;
    bool has_labels = false;
    bool plain = false;  // `plain` storage: PageType::Node pages of PlainNode items (AM/plain/node.rs:15-22); W = dimensions, pvecs instead of codes
    std::vector<float> pvecs;  // [n][W] PlainNode.vector
    vs_node_layout lay{};
    uint32_t threads = 1;
    bool finished = false;
    bool headers_only = false;  // vs_pages_headers_only: block table + metadata pages only (the device decodes the nodes)
    // per block
    std::vector<uint32_t> blk_base;  // dense id of the block's first node
    std::vector<uint32_t> blk_cnt;   // SbqNode items on the block (0 for every other page type)
    std::vector<int8_t> blk_type;
    uint32_t by_type[9] = {0};
    uint32_t n_new = 0;
    std::unordered_map<uint32_t, std::vector<uint8_t>> kept;  // chained / metadata pages, by block
    // per node
    bool geometry_known = false;
    uint32_t W = 0, R = 0;
    uint64_t n = 0;
    std::vector<uint64_t> codes;     // [n][W]
    std::vector<uint64_t> nbr_raw;   // [n][R] (block << 16) | offset, kNoNeighbor from the first invalid slot on
    std::vector<uint32_t> nbrs;      // [n][R] dense ids (after finish)
    std::vector<uint64_t> tids;      // [n]
    std::vector<uint32_t> label_off; // [n+1]
    std::vector<int16_t> label_val;
    uint32_t n_deleted = 0;
    uint32_t meta_magic = 0, meta_version = 0;
};

static int fail(const char* fmt, ...) {
    char buf[900];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    vs_set_error("%s", buf);
    return VS_ERR_INVALID;
}

extern "C" {

int vs_node_layout_default(int has_labels, vs_node_layout* out) {
    if (!out) return fail("vs_node_layout_default: null output");
    // declaration order of ClassicSbqNode / LabeledSbqNode (AM/sbq/node.rs:26-42); every archived field is 8 bytes, align 4
    out->root_size = 32;
    out->off_heap_item_pointer = 0;
    out->off_bq_vector = 8;
    out->off_neighbor_index_pointers = 16;
    out->off_labels = has_labels ? 24 : 0xFFFFFFFFu;  // Classic: +24 is _neighbor_vectors (unused, AM/sbq/node.rs:32)
    return VS_OK;
}

int vs_pages_open(uint32_t page_size, int has_labels, const vs_node_layout* layout, uint32_t threads, vs_pages** out) {
    if (!out) return fail("vs_pages_open: null output");
    *out = nullptr;
    if (page_size < 512 || page_size > 32768 || (page_size & (page_size - 1)))
        return fail("vs_pages_open: page_size %u is not a PostgreSQL block size", page_size);
    vs_pages* p = new (std::nothrow) vs_pages();
    if (!p) {
        vs_set_error("vs_pages_open: out of memory");
        return VS_ERR_OOM;
    }
    p->page_size = page_size;
    p->has_labels = has_labels != 0;
    if (layout)
        p->lay = *layout;
    else
        vs_node_layout_default(has_labels, &p->lay);
    const vs_node_layout& l = p->lay;
    const uint32_t offs[4] = {l.off_heap_item_pointer, l.off_bq_vector, l.off_neighbor_index_pointers, l.off_labels};
    for (int i = 0; i < (has_labels ? 4 : 3); i++)
        if (offs[i] > l.root_size || l.root_size - offs[i] < 8 || l.root_size > 4096) {
            delete p;
            return fail("vs_pages_open: field offset %u does not fit a %u-byte archived node", offs[i], l.root_size);
        }
    if (threads == 0) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    p->threads = threads;
    *out = p;
    return VS_OK;
}

int vs_plain_layout_default(vs_node_layout* out) {
    if (!out) return fail("vs_plain_layout_default: null output");
    // declaration order of PlainNode (AM/plain/node.rs:15-22): vector, pq_vector, neighbor_index_pointers, heap_item_pointer
    out->root_size = 32;
    out->off_bq_vector = 0;  // (the field that holds the node's vector: ArchivedVec<f32>)
    out->off_neighbor_index_pointers = 16;
    out->off_heap_item_pointer = 24;
    out->off_labels = 0xFFFFFFFFu;  // plain storage carries no labels (AM/plain/storage.rs:262)
    return VS_OK;
}

int vs_pages_open_plain(uint32_t page_size, const vs_node_layout* layout, uint32_t threads, vs_pages** out) {
    vs_node_layout lay;
    if (layout) lay = *layout;
    else vs_plain_layout_default(&lay);
    int rc = vs_pages_open(page_size, 0, &lay, threads, out);
    if (rc == VS_OK) (*out)->plain = true;
    return rc;
}

void vs_pages_close(vs_pages* p) { delete p; }

int vs_pages_add(vs_pages* p, uint32_t first_block, const void* pages, uint32_t n_blocks) {
    if (!p || (!pages && n_blocks)) return fail("vs_pages_add: null argument");
    if (p->finished) {
        vs_set_error("vs_pages_add after vs_pages_finish");
        return VS_ERR_STATE;
    }
    if (first_block != p->blk_cnt.size())
        return fail("vs_pages_add: blocks must arrive in order (expected block %zu, got %u)", p->blk_cnt.size(), first_block);
    if ((uint64_t)first_block + n_blocks >= kInvalidBlock) return fail("vs_pages_add: block number overflow");
    const uint8_t* base = static_cast<const uint8_t*>(pages);
    const uint32_t ps = p->page_size;
    const vs_node_layout lay = p->lay;

    // pass 1 (serial, headers only): page types, item counts, dense-id bases, copies of the metadata pages.
    // Nothing is committed to the reader until the whole call has succeeded.
    std::vector<PageView> views(n_blocks);
    std::vector<uint32_t> l_base(n_blocks), l_cnt(n_blocks, 0);
    std::vector<int8_t> l_type(n_blocks);
    uint32_t l_by_type[9] = {0}, l_new = 0;
    const uint64_t n0 = p->n;
    uint64_t n1 = n0;
    std::string err;
    for (uint32_t b = 0; b < n_blocks; b++) {
        PageView& v = views[b];
        if (!view_page(base + (size_t)b * ps, ps, v, err)) return fail("block %u: %s", first_block + b, err.c_str());
        if (v.type < 0)
            l_new++;
        else {
            l_by_type[v.type]++;
            if (v.type == (p->plain ? VS_PAGE_NODE : VS_PAGE_SBQ_NODE)) l_cnt[b] = v.n_items;
        }
        l_base[b] = (uint32_t)n1;
        l_type[b] = (int8_t)v.type;
        n1 += l_cnt[b];
        if (n1 >= VS_INVALID_NODE) return fail("more than %u index nodes", VS_INVALID_NODE - 1);
    }
    auto commit_blocks = [&]() -> int {
        try {
            for (uint32_t b = 0; b < n_blocks; b++) {
                const int t = l_type[b];
                if (t == VS_PAGE_SBQ_MEANS || t == VS_PAGE_META || t == VS_PAGE_SBQ_MEANS_V1 || t == VS_PAGE_META_V2)
                    p->kept[first_block + b].assign(views[b].p, views[b].p + ps);
            }
            p->blk_base.insert(p->blk_base.end(), l_base.begin(), l_base.end());
            p->blk_cnt.insert(p->blk_cnt.end(), l_cnt.begin(), l_cnt.end());
            p->blk_type.insert(p->blk_type.end(), l_type.begin(), l_type.end());
        } catch (const std::bad_alloc&) {
            vs_set_error("vs_pages_add: out of host memory");
            return VS_ERR_OOM;
        }
        for (int t = 0; t < 9; t++) p->by_type[t] += l_by_type[t];
        p->n_new += l_new;
        return VS_OK;
    };
    if (n1 == n0) return commit_blocks();
    if (p->headers_only) {  // node items are decoded elsewhere (vs_pages_dev_*): only the block table is kept
        const int rc0 = commit_blocks();
        if (rc0 == VS_OK) p->n = n1;
        return rc0;
    }

    // geometry (W, R) is fixed for an index: every node is written with num_neighbors slots (AM/sbq/node.rs:62-66) and
    // a quantized_size()-word code (AM/sbq/quantize.rs:37-45); take it from the first node seen
    const bool geometry_was_known = p->geometry_known;
    if (!p->geometry_known) {
        for (uint32_t b = 0; b < n_blocks && !p->geometry_known; b++) {
            if (l_cnt[b] == 0) continue;
            const uint8_t* it;
            uint32_t len;
            if (!get_item(views[b], 1, it, len, err)) return fail("block %u: %s", first_block + b, err.c_str());
            if (len < lay.root_size) return fail("block %u item 1: %u bytes cannot hold a %u-byte archived node", first_block + b, len, lay.root_size);
            const uint32_t root = len - lay.root_size;
            const uint8_t* f;
            uint32_t w, r;
            if (!archived_vec(it, len, root + lay.off_bq_vector, p->plain ? 4 : 8, f, w, err) ||
                !archived_vec(it, len, root + lay.off_neighbor_index_pointers, 8, f, r, err))
                return fail("block %u item 1: %s", first_block + b, err.c_str());
            if (w == 0 || w > (p->plain ? 16000u : 1024u))
                return fail("block %u item 1: %s of %u %s", first_block + b, p->plain ? "vector" : "bq_vector", w, p->plain ? "dimensions" : "words");
            if (r == 0 || r > 4096) return fail("block %u item 1: %u neighbor slots", first_block + b, r);
            p->W = w;
            p->R = r;
            p->geometry_known = true;
        }
    }
    const uint32_t W = p->W, R = p->R;
    try {
        if (p->plain) p->pvecs.resize((size_t)n1 * W);
        else p->codes.resize((size_t)n1 * W);
        p->nbr_raw.resize((size_t)n1 * R);
        p->tids.resize(n1);
        if (p->has_labels) p->label_off.resize(n1 + 1, 0);
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_pages_add: out of host memory for %llu nodes", (unsigned long long)n1);
        return VS_ERR_OOM;
    }

    // pass 2 (parallel over blocks): decode every SbqNode item
    std::vector<std::pair<const uint8_t*, uint32_t>> lab;  // per new node: its label slice inside `pages`
    if (p->has_labels) lab.resize(n1 - n0);
    std::atomic<int> failed{0};
    std::string first_err;
    std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
    std::atomic<uint32_t> deleted{0};
    auto report = [&](uint32_t blk, uint32_t off, const std::string& why) {
        if (failed.exchange(1) == 0) {
            while (err_lock.test_and_set()) {}
            char buf[400];
            snprintf(buf, sizeof buf, "block %u item %u: %s", blk, off, why.c_str());
            first_err = buf;
            err_lock.clear();
        }
    };
    parallel_for(p->threads, n_blocks, [&](uint64_t b) {
        const uint32_t blk = first_block + (uint32_t)b;
        const uint32_t cnt = l_cnt[b];
        if (cnt == 0 || failed.load(std::memory_order_relaxed)) return;
        const PageView& v = views[b];
        std::string e;
        uint32_t del = 0;
        for (uint32_t off = 1; off <= cnt; off++) {
            const uint64_t node = (uint64_t)l_base[b] + off - 1;
            const uint8_t* it;
            uint32_t len;
            if (!get_item(v, off, it, len, e)) return report(blk, off, e);
            if (len < lay.root_size) return report(blk, off, "item shorter than the archived node");
            const uint32_t root = len - lay.root_size;  // rkyv::archived_root: the root object is the tail of the buffer
            // heap_item_pointer
            const uint8_t* hp = it + root + lay.off_heap_item_pointer;
            const uint32_t hblk = rd32(hp);
            const uint32_t hoff = rd16(hp + 4);
            p->tids[node] = ((uint64_t)hblk << 16) | hoff;
            if (hoff == 0) del++;  // is_deleted(): offset == InvalidOffsetNumber (AM/sbq/node.rs:158-160)
            // bq_vector
            const uint8_t* f;
            uint32_t c;
            if (!archived_vec(it, len, root + lay.off_bq_vector, p->plain ? 4 : 8, f, c, e)) return report(blk, off, e);
            if (c != W) return report(blk, off, p->plain ? "vector length differs from the index's dimensions" : "bq_vector length differs from the index's code width");
            if (p->plain) memcpy(&p->pvecs[node * W], f, (size_t)W * 4);  // PlainNode.vector (AM/plain/node.rs:18)
            else memcpy(&p->codes[node * W], f, (size_t)W * 8);
            // neighbor_index_pointers: the list ends at the first InvalidBlockNumber (AM/sbq/node.rs:260-285)
            if (!archived_vec(it, len, root + lay.off_neighbor_index_pointers, 8, f, c, e)) return report(blk, off, e);
            if (c != R) return report(blk, off, "neighbor slot count differs from the index's num_neighbors");
            uint64_t* out = &p->nbr_raw[node * R];
            bool ended = false;
            for (uint32_t j = 0; j < R; j++) {
                const uint32_t nb = rd32(f + 8 * j);
                const uint32_t no = rd16(f + 8 * j + 4);
                if (nb == kInvalidBlock) ended = true;
                out[j] = ended ? kNoNeighbor : (((uint64_t)nb << 16) | no);
            }
            // labels (LabeledSbqNode only): sorted, de-duplicated i16 (AM/labels/mod.rs:15-37)
            if (p->has_labels) {
                if (!archived_vec(it, len, root + lay.off_labels, 2, f, c, e)) return report(blk, off, e);
                for (uint32_t j = 1; j < c; j++)
                    if ((int16_t)rd16(f + 2 * j) <= (int16_t)rd16(f + 2 * j - 2))
                        return report(blk, off, "label set is not strictly increasing");
                lab[node - n0] = {f, c};
            }
        }
        deleted += del;
    });
    auto rollback = [&]() {  // leave the reader as it was before this call
        if (p->plain) p->pvecs.resize((size_t)n0 * W);
        else p->codes.resize((size_t)n0 * W);
        p->nbr_raw.resize((size_t)n0 * R);
        p->tids.resize(n0);
        if (p->has_labels) p->label_off.resize(n0 + 1);
        p->geometry_known = geometry_was_known;
    };
    if (failed.load()) {
        rollback();
        return fail("%s", first_err.c_str());
    }
    if (p->has_labels) {
        uint64_t tot = p->label_val.size();
        for (uint64_t i = n0; i < n1; i++) {
            p->label_off[i] = (uint32_t)tot;
            tot += lab[i - n0].second;
        }
        if (tot >= 0xFFFFFFFFull) {
            rollback();
            return fail("label CSR exceeds 2^32 entries");
        }
        p->label_off[n1] = (uint32_t)tot;
        try {
            p->label_val.resize(tot);
        } catch (const std::bad_alloc&) {
            rollback();
            vs_set_error("vs_pages_add: out of host memory");
            return VS_ERR_OOM;
        }
        parallel_for(p->threads, n1 - n0, [&](uint64_t i) {
            if (lab[i].second) memcpy(&p->label_val[p->label_off[n0 + i]], lab[i].first, (size_t)lab[i].second * 2);
        });
    }
    int rc = commit_blocks();
    if (rc != VS_OK) {
        rollback();
        if (p->has_labels) p->label_val.resize(p->label_off[n0]);
        return rc;
    }
    p->n_deleted += deleted.load();
    p->n = n1;
    return VS_OK;
}

static int find_item(const vs_pages* p, uint32_t block, uint32_t offset, int want_type_a, int want_type_b, const uint8_t*& data,
                     uint32_t& len) {
    auto it = p->kept.find(block);
    if (it == p->kept.end()) return fail("block %u is not a metadata page of this relation (or was not added)", block);
    PageView v;
    std::string err;
    if (!view_page(it->second.data(), p->page_size, v, err)) return fail("block %u: %s", block, err.c_str());
    if (v.type != want_type_a && v.type != want_type_b)
        return fail("block %u has page type %d, expected %d", block, v.type, want_type_a);  // assert!(page.get_type() == self.page_type), UT/chain.rs:170
    if (!get_item(v, offset, data, len, err)) return fail("block %u: %s", block, err.c_str());
    return VS_OK;
}

int vs_pages_read_chain(const vs_pages* p, uint32_t block, uint32_t offset, int page_type, void* buf, size_t cap, size_t* len) {
    if (!p || !len) return fail("vs_pages_read_chain: null argument");
    if (page_type != VS_PAGE_SBQ_MEANS && page_type != VS_PAGE_META)
        return fail("vs_pages_read_chain: page type %d is not chained (UT/page.rs:60-62)", page_type);
    // ChainItemIterator::next (UT/chain.rs:159-185)
    size_t total = 0;
    uint32_t hops = 0;
    uint8_t* out = static_cast<uint8_t*>(buf);
    while (block != kInvalidBlock) {
        if (++hops > (1u << 20)) return fail("vs_pages_read_chain: chain does not terminate");
        const uint8_t* it;
        uint32_t l;
        int r = find_item(p, block, offset, page_type, page_type, it, l);
        if (r != VS_OK) return r;
        if (l <= kChainHeader) return fail("block %u item %u: chain item of %u bytes has no payload", block, offset, l);
        const uint32_t nb = rd32(it);
        const uint32_t no = rd16(it + 4);
        const size_t part = l - kChainHeader;
        if (out && total + part <= cap) memcpy(out + total, it + kChainHeader, part);
        total += part;
        block = nb;
        offset = no;
    }
    *len = total;
    if (out && total > cap) return fail("vs_pages_read_chain: %zu bytes needed, buffer holds %zu", total, cap);
    return VS_OK;
}

int vs_pages_sbq_means(const vs_pages* p, uint32_t block, uint32_t offset, float* mean, float* m2, uint32_t dim_cap, uint32_t* dim,
                       uint64_t* count) {
    if (!p || !dim || !count) return fail("vs_pages_sbq_means: null argument");
    auto kit = p->kept.find(block);
    if (kit == p->kept.end()) return fail("block %u is not a metadata page of this relation (or was not added)", block);
    std::vector<uint8_t> bytes;
    const int type = p->blk_type[block];
    if (type == VS_PAGE_SBQ_MEANS) {  // SbqMeans::load, chained (AM/sbq/mod.rs:102-112)
        size_t l = 0;
        int r = vs_pages_read_chain(p, block, offset, VS_PAGE_SBQ_MEANS, nullptr, 0, &l);
        if (r != VS_OK) return r;
        bytes.resize(l);
        r = vs_pages_read_chain(p, block, offset, VS_PAGE_SBQ_MEANS, bytes.data(), l, &l);
        if (r != VS_OK) return r;
    } else if (type == VS_PAGE_SBQ_MEANS_V1) {  // SbqMeansV1::load, one plain item (AM/sbq/mod.rs:43-60)
        const uint8_t* it;
        uint32_t l;
        int r = find_item(p, block, offset, VS_PAGE_SBQ_MEANS_V1, VS_PAGE_SBQ_MEANS_V1, it, l);
        if (r != VS_OK) return r;
        bytes.assign(it, it + l);
    } else {
        return fail("Invalid page type %d for SbqMeans", type);  // AM/sbq/mod.rs:114-116
    }
    // ArchivedSbqMeans {u64 count, ArchivedVec<f32> means, ArchivedVec<f32> m2} = 24 bytes at the tail
    const uint32_t len = (uint32_t)bytes.size();
    if (len < 24) return fail("SbqMeans of %u bytes", len);
    const uint32_t root = len - 24;
    std::string err;
    const uint8_t *fm, *f2;
    uint32_t cm, c2;
    if (!archived_vec(bytes.data(), len, root + 8, 4, fm, cm, err) || !archived_vec(bytes.data(), len, root + 16, 4, f2, c2, err))
        return fail("SbqMeans: %s", err.c_str());
    if (c2 != cm && c2 != 0) return fail("SbqMeans: %u means but %u m2 entries", cm, c2);
    *dim = cm;
    *count = rd64(bytes.data() + root);
    if (cm > dim_cap) {
        if (mean || m2) return fail("SbqMeans has %u dimensions, buffers hold %u", cm, dim_cap);
        return VS_OK;
    }
    if (mean) memcpy(mean, fm, (size_t)cm * 4);
    if (m2) {
        if (c2) memcpy(m2, f2, (size_t)c2 * 4);
        else memset(m2, 0, (size_t)cm * 4);
    }
    return VS_OK;
}

int vs_meta_layout_default(vs_meta_layout* out) {
    if (!out) return fail("vs_meta_layout_default: null argument");
    // declaration order (AM/meta_page.rs:179-210), C alignment: u32 u32 String(8,4) u16 u32 u32 u8 u8 u32 u32 f64 Option<StartNodes>(20,4)
    // ItemPointer(8,4) bool
    *out = vs_meta_layout{80, 0, 4, 8, 16, 20, 24, 28, 29, 32, 36, 40, 48, 68, 76};
    return VS_OK;
}

namespace {
struct MetaCursor {
    const uint8_t* b;
    size_t len;
    bool in(int64_t pos, size_t n) const { return pos >= 0 && (uint64_t)pos + n <= len; }
};
constexpr uint32_t kBtHeader = 12, kBtLeafEntry = 12, kBtInnerEntry = 8;

// in-order walk of an ArchivedBTreeMap<i16, ArchivedItemPointer> node; false + err on a malformed tree
bool walk_btree(const MetaCursor& c, int64_t pos, int depth, std::vector<int16_t>& keys, std::vector<uint64_t>& vals, uint64_t limit,
                std::string& err) {
    char buf[160];
    if (depth > 8 || !c.in(pos, kBtHeader)) {
        snprintf(buf, sizeof buf, "B-tree node at %lld lies outside the archive (depth %d)", (long long)pos, depth);
        err = buf;
        return false;
    }
    const uint16_t meta = rd16(c.b + pos);
    const uint32_t cnt = meta & 0x7FFFu;
    if (meta & 0x8000u) {
        if (!c.in(pos, kBtHeader + (size_t)cnt * kBtInnerEntry)) {
            err = "inner B-tree node runs past the archive";
            return false;
        }
        if (!walk_btree(c, pos + 8 + rdi32(c.b + pos + 8), depth + 1, keys, vals, limit, err)) return false;
        for (uint32_t e = 0; e < cnt; ++e) {
            const int64_t ep = pos + kBtHeader + (int64_t)e * kBtInnerEntry;
            if (!walk_btree(c, ep + rdi32(c.b + ep), depth + 1, keys, vals, limit, err)) return false;
        }
        return true;
    }
    if (!c.in(pos, kBtHeader + (size_t)cnt * kBtLeafEntry)) {
        err = "leaf B-tree node runs past the archive";
        return false;
    }
    for (uint32_t e = 0; e < cnt; ++e) {
        const uint8_t* q = c.b + pos + kBtHeader + (size_t)e * kBtLeafEntry;
        if (keys.size() >= limit) {
            err = "B-tree holds more entries than its length field says";
            return false;
        }
        keys.push_back((int16_t)rd16(q));
        vals.push_back(((uint64_t)rd32(q + 4) << 16) | rd16(q + 8));
    }
    return true;
}
}  // namespace

int vs_meta_page_decode(const void* bytes, size_t len, const vs_meta_layout* layout, vs_meta_page* out, int16_t* start_labels,
                        uint32_t* start_blocks, uint32_t* start_offsets, uint32_t cap) {
    if (!bytes || !out) return fail("vs_meta_page_decode: null argument");
    vs_meta_layout L;
    if (layout) L = *layout;
    else vs_meta_layout_default(&L);
    const uint32_t offs[] = {L.off_magic_number + 4, L.off_version + 4, L.off_extension_version_when_built + 8, L.off_distance_type + 2,
                             L.off_num_dimensions + 4, L.off_num_dimensions_to_index + 4, L.off_bq_num_bits_per_dimension + 1,
                             L.off_storage_type + 1, L.off_num_neighbors + 4, L.off_search_list_size + 4, L.off_max_alpha + 8,
                             L.off_start_nodes + 20, L.off_quantizer_metadata + 8, L.off_has_labels + 1};
    for (uint32_t e : offs)
        if (e > L.root_size) return fail("vs_meta_layout: a field ends at byte %u of a %u-byte root object", e, L.root_size);
    if (len < L.root_size) return fail("MetaPage archive of %zu bytes is shorter than its %u-byte root object", len, L.root_size);
    const uint8_t* b = static_cast<const uint8_t*>(bytes);
    const size_t root = len - L.root_size;  // rkyv::archived_root: the root object is the tail of the buffer
    const MetaCursor c{b, len};
    try {
        memset(out, 0, sizeof *out);
        out->magic_number = rd32(b + root + L.off_magic_number);
        out->version = rd32(b + root + L.off_version);
        if (out->magic_number != kTsvMagic) return fail("MetaPage magic %u is not %u", out->magic_number, kTsvMagic);
        {  // ArchivedString
            const size_t f = root + L.off_extension_version_when_built;
            const uint8_t* src;
            size_t n;
            if ((b[f + 7] & 0x80) == 0) {
                n = b[f + 7];
                if (n > 7) return fail("MetaPage: inline string of %zu bytes", n);
                src = b + f;
            } else {
                n = rd32(b + f);
                const int64_t at = (int64_t)f + rdi32(b + f + 4);
                if (!c.in(at, n)) return fail("MetaPage: extension_version_when_built points outside the archive");
                src = b + at;
            }
            const size_t m = std::min(n, sizeof(out->extension_version_when_built) - 1);
            memcpy(out->extension_version_when_built, src, m);
        }
        out->distance_type = rd16(b + root + L.off_distance_type);
        out->num_dimensions = rd32(b + root + L.off_num_dimensions);
        out->num_dimensions_to_index = rd32(b + root + L.off_num_dimensions_to_index);
        out->bq_num_bits_per_dimension = b[root + L.off_bq_num_bits_per_dimension];
        out->storage_type = b[root + L.off_storage_type];
        out->num_neighbors = rd32(b + root + L.off_num_neighbors);
        out->search_list_size = rd32(b + root + L.off_search_list_size);
        memcpy(&out->max_alpha, b + root + L.off_max_alpha, 8);
        out->quantizer_block = rd32(b + root + L.off_quantizer_metadata);
        out->quantizer_offset = rd16(b + root + L.off_quantizer_metadata + 4);
        out->has_labels = b[root + L.off_has_labels] ? 1u : 0u;
        const size_t sn = root + L.off_start_nodes;  // ArchivedOption<ArchivedStartNodes>
        if (b[sn] > 1) return fail("MetaPage: Option tag %u", b[sn]);
        out->has_start_nodes = b[sn];
        out->default_start_block = kInvalidBlock;
        if (b[sn]) {
            out->default_start_block = rd32(b + sn + 4);
            out->default_start_offset = rd16(b + sn + 8);
            const uint32_t cnt = rd32(b + sn + 12);
            out->n_labeled_start_nodes = cnt;
            if (cnt > 65536) return fail("MetaPage: %u labeled start nodes (labels are smallints)", cnt);
            std::vector<int16_t> keys;
            std::vector<uint64_t> vals;
            if (cnt) {
                std::string err;
                keys.reserve(cnt);
                vals.reserve(cnt);
                if (!walk_btree(c, (int64_t)sn + 16 + rdi32(b + sn + 16), 0, keys, vals, cnt, err)) return fail("MetaPage: %s", err.c_str());
                if (keys.size() != cnt) return fail("MetaPage: B-tree holds %zu entries, its length field says %u", keys.size(), cnt);
                for (size_t i = 1; i < keys.size(); ++i)
                    if (keys[i - 1] >= keys[i]) return fail("MetaPage: start-node labels are not strictly increasing at entry %zu", i);
            }
            if (start_labels || start_blocks || start_offsets) {
                if (cnt > cap) return fail("MetaPage has %u labeled start nodes, the buffers hold %u", cnt, cap);
                for (uint32_t i = 0; i < cnt; ++i) {
                    if (start_labels) start_labels[i] = keys[i];
                    if (start_blocks) start_blocks[i] = (uint32_t)(vals[i] >> 16);
                    if (start_offsets) start_offsets[i] = (uint32_t)(vals[i] & 0xFFFF);
                }
            }
        }
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_meta_page_decode: out of host memory");
        return VS_ERR_OOM;
    }
    return VS_OK;
}

int vs_pages_meta(const vs_pages* p, const vs_meta_layout* layout, vs_meta_page* meta, vs_index_desc* desc, int16_t* start_labels,
                  uint32_t* start_nodes, uint32_t cap) {
    if (!p || !desc) return fail("vs_pages_meta: null argument");
    if (!p->finished) return fail("vs_pages_meta: call vs_pages_finish first");
    if (p->blk_type.empty() || p->blk_type[0] != VS_PAGE_META)
        return fail("block 0 has page type %d: only PageType::Meta (format version 3) relations carry a chained MetaPage",
                    p->blk_type.empty() ? -1 : p->blk_type[0]);
    try {
        size_t l = 0;
        int r = vs_pages_read_chain(p, 0, 2, VS_PAGE_META, nullptr, 0, &l);  // META_OFFSET = 2 (AM/meta_page.rs:28)
        if (r != VS_OK) return r;
        std::vector<uint8_t> bytes(l);
        r = vs_pages_read_chain(p, 0, 2, VS_PAGE_META, bytes.data(), l, &l);
        if (r != VS_OK) return r;
        vs_meta_page m;
        r = vs_meta_page_decode(bytes.data(), l, layout, &m, nullptr, nullptr, nullptr, 0);
        if (r != VS_OK) return r;
        std::vector<int16_t> lab(m.n_labeled_start_nodes);
        std::vector<uint32_t> blk(m.n_labeled_start_nodes), off(m.n_labeled_start_nodes);
        r = vs_meta_page_decode(bytes.data(), l, layout, &m, lab.data(), blk.data(), off.data(), m.n_labeled_start_nodes);
        if (r != VS_OK) return r;
        if (m.version != p->meta_version) return fail("MetaPage version %u, MetaPageHeader version %u", m.version, p->meta_version);
        if (m.distance_type > 2) return fail("Unknown DistanceType number %u", m.distance_type);       // DistanceType::from_u16
        if (m.storage_type != 0 && m.storage_type != 2) return fail("Invalid storage type %u", m.storage_type);  // StorageType::from_u8
        if (m.num_dimensions_to_index == 0 || m.num_dimensions_to_index > m.num_dimensions)
            return fail("MetaPage: %u of %u dimensions indexed", m.num_dimensions_to_index, m.num_dimensions);
        memset(desc, 0, sizeof *desc);
        desc->n = (uint32_t)p->n;
        desc->dim_full = m.num_dimensions;
        desc->dim_index = m.num_dimensions_to_index;
        desc->bits = m.bq_num_bits_per_dimension;
        desc->words = (uint32_t)(((uint64_t)m.num_dimensions_to_index * m.bq_num_bits_per_dimension + 63) / 64);  // AM/sbq/quantize.rs:37-45
        desc->num_neighbors = m.num_neighbors;
        desc->distance_type = m.distance_type;
        desc->has_labels = m.has_labels;
        desc->storage_type = m.storage_type == 0 ? VS_STORAGE_PLAIN : VS_STORAGE_SBQ;
        desc->default_start = VS_INVALID_NODE;
        if (p->n && !p->headers_only) {
            if ((uint32_t)p->has_labels != m.has_labels)
                return fail("the reader was opened with has_labels = %d, the MetaPage says %u", p->has_labels, m.has_labels);
            if ((desc->storage_type == VS_STORAGE_PLAIN) != p->plain)
                return fail("the MetaPage says storage type %u, the reader was opened for %s nodes", m.storage_type, p->plain ? "plain" : "SBQ");
            if (desc->storage_type == VS_STORAGE_SBQ && (p->W != desc->words || p->R != m.num_neighbors))
                return fail("SbqNode items hold %u-word codes and %u neighbor slots, the MetaPage implies %u and %u", p->W, p->R,
                            desc->words, m.num_neighbors);
            if (desc->storage_type == VS_STORAGE_PLAIN && (p->W != m.num_dimensions_to_index || p->R != m.num_neighbors))
                return fail("PlainNode items hold %u-dimensional vectors and %u neighbor slots, the MetaPage implies %u and %u", p->W, p->R,
                            m.num_dimensions_to_index, m.num_neighbors);
        }
        if (m.has_start_nodes) {
            r = vs_pages_node_of(p, m.default_start_block, m.default_start_offset, &desc->default_start);
            if (r != VS_OK) return r;
            desc->n_label_starts = m.n_labeled_start_nodes;
            if (start_labels || start_nodes) {
                if (m.n_labeled_start_nodes > cap)
                    return fail("MetaPage has %u labeled start nodes, the buffers hold %u", m.n_labeled_start_nodes, cap);
                for (uint32_t i = 0; i < m.n_labeled_start_nodes; ++i) {
                    uint32_t node;
                    r = vs_pages_node_of(p, blk[i], off[i], &node);
                    if (r != VS_OK) return r;
                    if (start_labels) start_labels[i] = lab[i];
                    if (start_nodes) start_nodes[i] = node;
                }
            }
        }
        if (meta) *meta = m;
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_pages_meta: out of host memory");
        return VS_ERR_OOM;
    }
    return VS_OK;
}

int vs_pages_finish(vs_pages* p, vs_pages_info* info) {
    if (!p) return fail("vs_pages_finish: null reader");
    if (!p->finished) {
        // MetaPageHeader {magic_number, version} = item 1 of block 0 (AM/meta_page.rs:26-27,166-174,351-356)
        if (!p->blk_type.empty() && (p->blk_type[0] == VS_PAGE_META || p->blk_type[0] == VS_PAGE_META_V2)) {
            uint8_t hdr[16];
            size_t l = 0;
            if (p->blk_type[0] == VS_PAGE_META) {
                int r = vs_pages_read_chain(p, 0, 1, VS_PAGE_META, hdr, sizeof hdr, &l);
                if (r != VS_OK) return r;
            } else {
                const uint8_t* it;
                uint32_t il;
                int r = find_item(p, 0, 1, VS_PAGE_META_V2, VS_PAGE_META_V2, it, il);
                if (r != VS_OK) return r;
                l = std::min<size_t>(il, sizeof hdr);
                memcpy(hdr, it, l);
            }
            if (l != 8) return fail("MetaPageHeader of %zu bytes", l);
            p->meta_magic = rd32(hdr);
            p->meta_version = rd32(hdr + 4);
            if (p->meta_magic != kTsvMagic) return fail("meta page magic %u is not %u", p->meta_magic, kTsvMagic);
        }
        if (!p->plain && p->n == 0 && p->by_type[VS_PAGE_NODE] > 0)
            return fail("the relation holds `plain` storage nodes (PageType::Node): open the reader with vs_pages_open_plain");
        if (p->plain && p->n == 0 && p->by_type[VS_PAGE_SBQ_NODE] > 0)
            return fail("the relation holds memory_optimized (SBQ) nodes: open the reader with vs_pages_open");
        if (p->headers_only) {
            p->finished = true;
        } else {
        // neighbor ItemPointers -> dense ids
        const uint32_t R = p->R;
        const uint64_t n = p->n;
        try {
            p->nbrs.resize((size_t)n * R);
        } catch (const std::bad_alloc&) {
            vs_set_error("vs_pages_finish: out of host memory");
            return VS_ERR_OOM;
        }
        const uint32_t nblk = (uint32_t)p->blk_cnt.size();
        std::atomic<uint64_t> bad{~0ull};
        parallel_for(p->threads, n, [&](uint64_t i) {
            const uint64_t* in = &p->nbr_raw[i * R];
            uint32_t* out = &p->nbrs[i * R];
            for (uint32_t j = 0; j < R; j++) {
                const uint64_t v = in[j];
                if (v == kNoNeighbor) {
                    out[j] = VS_INVALID_NODE;
                    continue;
                }
                const uint32_t b = (uint32_t)(v >> 16), o = (uint32_t)(v & 0xFFFF);
                if (b >= nblk || o < 1 || o > p->blk_cnt[b]) {
                    uint64_t exp = ~0ull;
                    bad.compare_exchange_strong(exp, i * R + j);
                    out[j] = VS_INVALID_NODE;
                    continue;
                }
                out[j] = p->blk_base[b] + o - 1;
            }
        });
        if (bad.load() != ~0ull) {
            const uint64_t at = bad.load();
            const uint64_t v = p->nbr_raw[at];
            p->nbrs.clear();
            return fail("node %llu neighbor slot %llu points at (%u,%u), which is not an SbqNode item of this relation",
                        (unsigned long long)(at / R), (unsigned long long)(at % R), (uint32_t)(v >> 16), (uint32_t)(v & 0xFFFF));
        }
        std::vector<uint64_t>().swap(p->nbr_raw);
        p->finished = true;
        }
    }
    if (info) {
        memset(info, 0, sizeof *info);
        info->n_blocks = (uint32_t)p->blk_cnt.size();
        info->n_nodes = (uint32_t)p->n;
        info->words = p->W;
        info->num_neighbors = p->R;
        info->has_labels = p->has_labels;
        info->n_deleted = p->n_deleted;
        info->n_label_vals = p->label_val.size();
        for (int t = 0; t < 9; t++) info->pages_by_type[t] = p->by_type[t];
        info->new_pages = p->n_new;
        info->meta_magic = p->meta_magic;
        info->meta_version = p->meta_version;
    }
    return VS_OK;
}

int vs_pages_headers_only(vs_pages* p) {
    if (!p) return fail("vs_pages_headers_only: null reader");
    if (!p->blk_cnt.empty()) {
        vs_set_error("vs_pages_headers_only after blocks were added");
        return VS_ERR_STATE;
    }
    p->headers_only = true;
    return VS_OK;
}

int vs_pages_block_table(const vs_pages* p, const uint32_t** blk_base, const uint32_t** blk_cnt, uint32_t* n_blocks) {
    if (!p || !blk_base || !blk_cnt || !n_blocks) return fail("vs_pages_block_table: null argument");
    *blk_base = p->blk_base.data();
    *blk_cnt = p->blk_cnt.data();
    *n_blocks = (uint32_t)p->blk_cnt.size();
    return VS_OK;
}

int vs_pages_host(const vs_pages* p, vs_index_host* host) {
    if (!p || !host) return fail("vs_pages_host: null argument");
    if (p->headers_only) {
        vs_set_error("vs_pages_host: this reader only keeps the block table (vs_pages_headers_only)");
        return VS_ERR_STATE;
    }
    if (!p->finished) {
        vs_set_error("vs_pages_host before vs_pages_finish");
        return VS_ERR_STATE;
    }
    memset(host, 0, sizeof *host);
    if (p->plain) host->vecs = p->pvecs.data();  // PlainNode.vector: the (cosine-normalised) index slice, [n][vs_pages_info.words] floats
    else host->codes = p->codes.data();
    host->nbrs = p->nbrs.data();
    host->nbr_stride = p->R;
    host->heap_tids = p->tids.data();
    if (p->has_labels) {
        host->label_off = p->label_off.data();
        host->label_val = p->label_val.data();
    }
    return VS_OK;
}

int vs_pages_node_of(const vs_pages* p, uint32_t block, uint32_t offset, uint32_t* node) {
    if (!p || !node) return fail("vs_pages_node_of: null argument");
    if (block >= p->blk_cnt.size() || offset < 1 || offset > p->blk_cnt[block])
        return fail("(%u,%u) is not an SbqNode item of this relation", block, offset);
    *node = p->blk_base[block] + offset - 1;
    return VS_OK;
}

int vs_pages_item_pointer_of(const vs_pages* p, uint32_t node, uint32_t* block, uint32_t* offset) {
    if (!p || !block || !offset) return fail("vs_pages_item_pointer_of: null argument");
    if (node >= p->n) return fail("node %u of %llu", node, (unsigned long long)p->n);
    // last block whose base <= node and which holds nodes
    auto it = std::upper_bound(p->blk_base.begin(), p->blk_base.end(), node);
    size_t b = (size_t)(it - p->blk_base.begin()) - 1;
    while (p->blk_cnt[b] == 0) b--;  // blocks without nodes share the base of the next node page
    *block = (uint32_t)b;
    *offset = node - p->blk_base[b] + 1;
    return VS_OK;
}

}  // extern "C"
