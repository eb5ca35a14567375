// vs_pages_dev.hip — index relation pages decoded ON the device (the MI355X-first form of the staging path).
//
// vs_pages.cpp decodes SbqNode items on the host cores and uploads flat arrays.  Here the host only copies: the blocks of
// the relation go through the pinned ring to HBM as they are (hipMemcpyAsync), the host keeps nothing but the block table
// it reads off the page headers on the way past (page type + item count per block -> dense id of each block's first
// node; vs_pages_headers_only) and copies of the few metadata pages, and one kernel — a wave per node page — walks the
// line pointers, follows the rkyv relative pointers of every archived node (same layout facts and the same bounds checks
// as the host reader, vs_pages.cpp) and writes codes / neighbor ids / heap tids straight into the index arrays, neighbor
// IndexPointers translated through the block table.  33 GB of pages for a 50M-node index is 0.6 s of PCIe instead of
// tens of seconds of host decoding.  Label sets (LabeledSbqNode) take two more passes over the staged pages: count, then copy.
#include <algorithm>
#include <vector>

#include "vs_device.h"

struct vs_pages_dev {
    vs_ctx* ctx = nullptr;
    vs_pages* hdr = nullptr;  // host side: block table + metadata pages
    uint8_t* d_pages = nullptr;
    uint32_t page_size = VS_BLCKSZ;
    uint32_t cap_blocks = 0, n_blocks = 0;
    vs_node_layout lay{};
    bool layout_given = false;
};

enum { PE_OK = 0, PE_LINE_POINTER = 1, PE_ITEM_BOUNDS = 2, PE_SHORT_ITEM = 3, PE_VEC_BOUNDS = 4, PE_CODE_WIDTH = 5, PE_NEIGHBOR_SLOTS = 6,
       PE_DANGLING = 7, PE_LABELS = 8 };

__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {  // items are MAXALIGNed: 4-byte aligned words
    return *reinterpret_cast<const uint32_t*>(p);
}

// err[0] = first error code (0 = none), err[1] = block, err[2] = item, err[3] = detail
__device__ __forceinline__ void page_error(uint32_t* err, uint32_t code, uint32_t blk, uint32_t item, uint32_t detail) {
    if (atomicCAS(&err[0], 0u, code) == 0u) {
        err[1] = blk;
        err[2] = item;
        err[3] = detail;
    }
}

// one wave per block; blocks that hold no SbqNode items are skipped
__global__ __launch_bounds__(WAVE) void k_pages_decode(const uint8_t* __restrict__ pages, uint32_t page_size, uint32_t n_blocks,
                                                       const uint32_t* __restrict__ blk_base, const uint32_t* __restrict__ blk_cnt,
                                                       vs_node_layout lay, uint32_t W, uint32_t R, uint64_t* __restrict__ codes,
                                                       uint32_t code_stride, uint32_t* __restrict__ nbrs, uint32_t nbr_stride,
                                                       uint64_t* __restrict__ tids, uint32_t* __restrict__ err) {
    const int lane = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint32_t cnt = blk_cnt[b];
        if (cnt == 0) continue;
        const uint8_t* page = pages + (size_t)b * page_size;
        const uint32_t upper = ld16(page + 14), special = ld16(page + 16);
        for (uint32_t off = 1; off <= cnt; ++off) {
            const uint32_t node = blk_base[b] + off - 1;
            // PageGetItemId / PageGetItem (UT/ports.rs:56-77)
            const uint32_t lp = ld32(page + 24 + 4 * (off - 1));
            const uint32_t lp_off = lp & 0x7FFFu, lp_flags = (lp >> 15) & 3u, len = lp >> 17;
            if (lp_flags != 1u || len == 0) {
                if (lane == 0) page_error(err, PE_LINE_POINTER, b, off, lp);
                continue;
            }
            if (lp_off < upper || lp_off + len > special || (lp_off & 3u)) {
                if (lane == 0) page_error(err, PE_ITEM_BOUNDS, b, off, lp);
                continue;
            }
            if (len < lay.root_size) {
                if (lane == 0) page_error(err, PE_SHORT_ITEM, b, off, len);
                continue;
            }
            const uint8_t* item = page + lp_off;
            const uint32_t root = len - lay.root_size;  // rkyv::archived_root: the root object is the tail of the item
            // heap_item_pointer
            if (lane == 0) {
                const uint8_t* hp = item + root + lay.off_heap_item_pointer;
                tids[node] = ((uint64_t)ld32(hp) << 16) | ld16(hp + 4);
            }
            // bq_vector: ArchivedVec<u64> = {i32 offset relative to the field, u32 len}
            {
                const uint32_t fld = root + lay.off_bq_vector;
                const int64_t tgt = (int64_t)fld + (int32_t)ld32(item + fld);
                const uint32_t n = ld32(item + fld + 4);
                if (n != W) {
                    if (lane == 0) page_error(err, PE_CODE_WIDTH, b, off, n);
                    continue;
                }
                if (tgt < 0 || (uint64_t)tgt + (uint64_t)n * 8 > len || (tgt & 3)) {
                    if (lane == 0) page_error(err, PE_VEC_BOUNDS, b, off, fld);
                    continue;
                }
                for (uint32_t w = lane; w < W; w += WAVE) {
                    const uint8_t* src = item + tgt + 8 * w;
                    codes[(size_t)node * code_stride + w] = (uint64_t)ld32(src) | ((uint64_t)ld32(src + 4) << 32);
                }
            }
            // neighbor_index_pointers: ArchivedVec<ArchivedItemPointer {u32 block, u16 offset, pad}>, the list ends at the
            // first InvalidBlockNumber (AM/sbq/node.rs:260-285)
            {
                const uint32_t fld = root + lay.off_neighbor_index_pointers;
                const int64_t tgt = (int64_t)fld + (int32_t)ld32(item + fld);
                const uint32_t n = ld32(item + fld + 4);
                if (n != R) {
                    if (lane == 0) page_error(err, PE_NEIGHBOR_SLOTS, b, off, n);
                    continue;
                }
                if (tgt < 0 || (uint64_t)tgt + (uint64_t)n * 8 > len || (tgt & 3)) {
                    if (lane == 0) page_error(err, PE_VEC_BOUNDS, b, off, fld);
                    continue;
                }
                bool ended = false;
                for (uint32_t j0 = 0; j0 < R && !ended; j0 += WAVE) {
                    const uint32_t j = j0 + (uint32_t)lane;
                    uint32_t nb = 0xFFFFFFFFu, no = 0;
                    if (j < R) {
                        nb = ld32(item + tgt + 8 * j);
                        no = ld16(item + tgt + 8 * j + 4);
                    }
                    const uint64_t inval = __ballot(nb == 0xFFFFFFFFu);
                    const uint32_t nvalid = inval ? (uint32_t)__builtin_ctzll(inval) : WAVE;
                    if (nvalid < WAVE) ended = true;
                    if ((uint32_t)lane < nvalid) {
                        if (nb >= n_blocks || no < 1 || no > blk_cnt[nb]) page_error(err, PE_DANGLING, b, off, j);
                        else nbrs[(size_t)node * nbr_stride + j] = blk_base[nb] + no - 1;
                    }
                }
            }
        }
    }
}

// LabeledSbqNode.labels (ArchivedLabelSet = ArchivedVec<i16>, sorted and de-duplicated, AM/labels/mod.rs:15-37): WRITE = false
// counts the labels of every node (and checks them), WRITE = true copies them to label_val[label_off[node] ..].  The items
// were located and bounds-checked by k_pages_decode before.
template <bool WRITE>
__global__ __launch_bounds__(WAVE) void k_pages_labels(const uint8_t* __restrict__ pages, uint32_t page_size, uint32_t n_blocks,
                                                       const uint32_t* __restrict__ blk_base, const uint32_t* __restrict__ blk_cnt,
                                                       vs_node_layout lay, uint32_t* __restrict__ label_cnt,
                                                       const uint32_t* __restrict__ label_off, int16_t* __restrict__ label_val,
                                                       uint32_t* __restrict__ err) {
    const int lane = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint32_t cnt = blk_cnt[b];
        if (cnt == 0) continue;
        const uint8_t* page = pages + (size_t)b * page_size;
        for (uint32_t off = 1; off <= cnt; ++off) {
            const uint32_t node = blk_base[b] + off - 1;
            const uint32_t lp = ld32(page + 24 + 4 * (off - 1));
            const uint32_t lp_off = lp & 0x7FFFu, len = lp >> 17;
            const uint8_t* item = page + lp_off;
            const uint32_t fld = len - lay.root_size + lay.off_labels;
            const int64_t tgt = (int64_t)fld + (int32_t)ld32(item + fld);
            const uint32_t n = ld32(item + fld + 4);
            if (n && (tgt < 0 || (uint64_t)tgt + (uint64_t)n * 2 > len || (tgt & 1))) {
                if (lane == 0) page_error(err, PE_VEC_BOUNDS, b, off, fld);
                if (!WRITE && lane == 0) label_cnt[node] = 0;
                continue;
            }
            if (!WRITE) {
                for (uint32_t j = 1 + (uint32_t)lane; j < n; j += WAVE)
                    if ((int16_t)ld16(item + tgt + 2 * j) <= (int16_t)ld16(item + tgt + 2 * j - 2)) page_error(err, PE_LABELS, b, off, j);
                if (lane == 0) label_cnt[node] = n;
            } else {
                const uint32_t o = label_off[node];
                for (uint32_t j = lane; j < n; j += WAVE) label_val[o + j] = (int16_t)ld16(item + tgt + 2 * j);
            }
        }
    }
}

extern "C" int vs_pages_dev_open(vs_ctx* ctx, uint32_t page_size, const vs_node_layout* layout, uint32_t n_blocks_total,
                                 vs_pages_dev** out) {
    VS_REQUIRE(ctx && out, "vs_pages_dev_open: bad args");
    *out = nullptr;
    vs_pages_dev* d = new vs_pages_dev();
    d->ctx = ctx;
    d->page_size = page_size;
    int r = vs_pages_open(page_size, 0, layout, 1, &d->hdr);
    if (r == VS_OK) r = vs_pages_headers_only(d->hdr);
    if (r != VS_OK) {
        vs_pages_close(d->hdr);
        delete d;
        return r;
    }
    d->layout_given = layout != nullptr;
    if (layout) d->lay = *layout;
    else vs_node_layout_default(0, &d->lay);
    d->cap_blocks = std::max<uint32_t>(n_blocks_total, 1);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&d->d_pages, (size_t)d->cap_blocks * page_size);
    if (e != hipSuccess) {
        vs_set_error("vs_pages_dev_open: %zu bytes for the raw pages: %s", (size_t)d->cap_blocks * page_size, hipGetErrorString(e));
        vs_pages_close(d->hdr);
        delete d;
        return e == hipErrorOutOfMemory ? VS_ERR_OOM : VS_ERR_HIP;
    }
    *out = d;
    return VS_OK;
}

extern "C" void vs_pages_dev_close(vs_pages_dev* d) {
    if (!d) return;
    if (d->d_pages) (void)hipFree(d->d_pages);
    vs_pages_close(d->hdr);
    delete d;
}

extern "C" int vs_pages_dev_add(vs_pages_dev* d, uint32_t first_block, const void* pages, uint32_t n_blocks) {
    VS_REQUIRE(d && (pages || !n_blocks), "vs_pages_dev_add: bad args");
    VS_REQUIRE(d->d_pages, "vs_pages_dev_add after vs_pages_dev_build");
    VS_REQUIRE((uint64_t)first_block + n_blocks <= d->cap_blocks, "vs_pages_dev_add: more blocks than vs_pages_dev_open was told (%u)",
               d->cap_blocks);
    VS_TRY(vs_pages_add(d->hdr, first_block, pages, n_blocks));  // header checks, block table, metadata pages (host)
    VS_TRY(vs_dev_upload(d->ctx, d->d_pages + (size_t)first_block * d->page_size, pages, (size_t)n_blocks * d->page_size));
    d->n_blocks = first_block + n_blocks;
    return VS_OK;
}

extern "C" int vs_pages_dev_node_of(const vs_pages_dev* d, uint32_t block, uint32_t offset, uint32_t* node) {
    VS_REQUIRE(d, "vs_pages_dev_node_of: null reader");
    return vs_pages_node_of(d->hdr, block, offset, node);
}

extern "C" int vs_pages_dev_sbq_means(const vs_pages_dev* d, uint32_t block, uint32_t offset, float* mean, float* m2, uint32_t dim_cap,
                                      uint32_t* dim, uint64_t* count) {
    VS_REQUIRE(d, "vs_pages_dev_sbq_means: null reader");
    return vs_pages_sbq_means(d->hdr, block, offset, mean, m2, dim_cap, dim, count);
}

extern "C" int vs_pages_dev_meta(vs_pages_dev* d, const vs_meta_layout* layout, vs_meta_page* meta, vs_index_desc* desc,
                                 int16_t* start_labels, uint32_t* start_nodes, uint32_t cap) {
    VS_REQUIRE(d, "vs_pages_dev_meta: null reader");
    VS_TRY(vs_pages_finish(d->hdr, nullptr));  // (block table complete: every block has been added)
    return vs_pages_meta(d->hdr, layout, meta, desc, start_labels, start_nodes, cap);
}

// desc: the MetaPage fields (n is taken from the pages); extras: vecs / mean / m2 / count / start-node arrays (node ids)
extern "C" int vs_pages_dev_build(vs_pages_dev* d, const vs_index_desc* desc, const vs_index_host* extras, vs_pages_info* info,
                                  vs_index** out) {
    VS_REQUIRE(d && desc && extras && out, "vs_pages_dev_build: bad args");
    VS_REQUIRE(d->d_pages, "vs_pages_dev_build: already built");
    if (!d->layout_given) vs_node_layout_default(desc->has_labels ? 1 : 0, &d->lay);
    VS_REQUIRE(!desc->has_labels || (d->lay.off_labels <= d->lay.root_size && d->lay.root_size - d->lay.off_labels >= 8),
               "vs_pages_dev_build: the node layout has no labels field");
    VS_REQUIRE(desc->storage_type == VS_STORAGE_SBQ, "vs_pages_dev_build: memory_optimized (SBQ) indexes only");
    *out = nullptr;
    vs_pages_info pi{};
    VS_TRY(vs_pages_finish(d->hdr, &pi));
    if (info) *info = pi;
    VS_REQUIRE(pi.n_nodes > 0 || pi.pages_by_type[VS_PAGE_NODE] == 0,
               "the relation holds `plain` storage nodes (PageType::Node); this path reads memory_optimized (SBQ) indexes");
    vs_ctx* c = d->ctx;
    VS_HIP(hipSetDevice(c->device));
    vs_index_desc dd = *desc;
    dd.n = pi.n_nodes;
    vs_index* ix = nullptr;
    VS_TRY(vs_index_alloc(c, &dd, extras->vecs != nullptr, &ix));  // neighbor rows start as all-sentinel, codes as zero
    const uint32_t *h_base = nullptr, *h_cnt = nullptr;
    uint32_t nb = 0;
    uint32_t *d_base = nullptr, *d_cnt = nullptr, *d_err = nullptr;
    int r = vs_pages_block_table(d->hdr, &h_base, &h_cnt, &nb);
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (r == VS_OK && e != hipSuccess) {
            vs_set_error("vs_pages_dev_build: %s: %s", what, hipGetErrorString(e));
            r = e == hipErrorOutOfMemory ? VS_ERR_OOM : VS_ERR_HIP;
        }
    };
    uint32_t herr[4] = {0, 0, 0, 0};
    if (r == VS_OK && nb > 0 && pi.n_nodes > 0) {
        hip_ok(hipMalloc(&d_base, (size_t)nb * 4), "block table");
        hip_ok(hipMalloc(&d_cnt, (size_t)nb * 4), "block table");
        hip_ok(hipMalloc(&d_err, 16), "error record");
        if (r == VS_OK) r = vs_dev_upload(c, d_base, h_base, (size_t)nb * 4);
        if (r == VS_OK) r = vs_dev_upload(c, d_cnt, h_cnt, (size_t)nb * 4);
        if (r == VS_OK) hip_ok(hipMemsetAsync(d_err, 0, 16, c->stream), "error record");
        if (r == VS_OK) {
            const uint32_t grid = std::min<uint32_t>(nb, 1u << 20);
            hipLaunchKernelGGL(k_pages_decode, dim3(grid), dim3(WAVE), 0, c->stream, d->d_pages, d->page_size, nb, d_base, d_cnt,
                               d->lay, dd.words, dd.num_neighbors, ix->codes, ix->code_stride, ix->nbrs, ix->nbr_stride, ix->tids,
                               d_err);
            hip_ok(hipGetLastError(), "k_pages_decode");
            hip_ok(hipMemcpyAsync(herr, d_err, 16, hipMemcpyDeviceToHost, c->stream), "error record");
            hip_ok(hipStreamSynchronize(c->stream), "k_pages_decode");
        }
    }
    // labels: count per node, prefix sum on the host (4 bytes per node), then the values
    if (r == VS_OK && herr[0] == PE_OK && desc->has_labels && pi.n_nodes > 0) {
        const uint32_t n = pi.n_nodes;
        const uint32_t grid = std::min<uint32_t>(nb, 1u << 20);
        uint32_t* d_lcnt = nullptr;
        hip_ok(hipMalloc(&d_lcnt, (size_t)n * 4), "label counts");
        if (r == VS_OK) {
            hipLaunchKernelGGL((k_pages_labels<false>), dim3(grid), dim3(WAVE), 0, c->stream, d->d_pages, d->page_size, nb, d_base, d_cnt,
                               d->lay, d_lcnt, (const uint32_t*)nullptr, (int16_t*)nullptr, d_err);
            hip_ok(hipGetLastError(), "k_pages_labels");
        }
        std::vector<uint32_t> off((size_t)n + 1, 0);
        if (r == VS_OK) r = vs_dev_download(c, off.data() + 1, d_lcnt, (size_t)n * 4);
        if (r == VS_OK) hip_ok(hipMemcpy(herr, d_err, 16, hipMemcpyDeviceToHost), "error record");
        if (d_lcnt) (void)hipFree(d_lcnt);
        if (r == VS_OK && herr[0] == PE_OK) {
            uint64_t tot = 0;
            for (uint32_t i = 0; i < n; ++i) {
                tot += off[i + 1];
                off[i + 1] = (uint32_t)tot;
            }
            if (tot >= 0xFFFFFFFFull) {
                vs_set_error("label CSR exceeds 2^32 entries");
                r = VS_ERR_INVALID;
            }
            if (r == VS_OK) {
                hip_ok(hipMalloc(&ix->label_off, ((size_t)n + 1) * 4), "label offsets");
                hip_ok(hipMalloc(&ix->label_val, std::max<uint64_t>(tot, 1) * 2), "label values");
            }
            if (r == VS_OK) r = vs_dev_upload(c, ix->label_off, off.data(), ((size_t)n + 1) * 4);
            if (r == VS_OK) {
                hipLaunchKernelGGL((k_pages_labels<true>), dim3(grid), dim3(WAVE), 0, c->stream, d->d_pages, d->page_size, nb, d_base,
                                   d_cnt, d->lay, (uint32_t*)nullptr, (const uint32_t*)ix->label_off, ix->label_val, d_err);
                hip_ok(hipGetLastError(), "k_pages_labels");
                hip_ok(hipStreamSynchronize(c->stream), "k_pages_labels");
                ix->n_label_vals = tot;
                ix->d.has_labels = 1;
                if (r == VS_OK) r = vs_refresh_label_masks(ix);
            }
        }
    }
    if (d_base) (void)hipFree(d_base);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_err) (void)hipFree(d_err);
    if (r == VS_OK && herr[0] != PE_OK) {
        static const char* what[] = {"", "line pointer is not LP_NORMAL", "item lies outside pd_upper..pd_special",
                                     "item shorter than the archived node", "ArchivedVec points outside the item",
                                     "bq_vector length differs from the index's code width",
                                     "neighbor slot count differs from the index's num_neighbors",
                                     "neighbor points at something that is not an SbqNode item of this relation",
                                     "label set is not strictly increasing"};
        vs_set_error("block %u item %u: %s (detail %u)", herr[1], herr[2], what[herr[0] <= PE_LABELS ? herr[0] : 0], herr[3]);
        r = VS_ERR_INVALID;
    }
    // the raw pages are no longer needed
    (void)hipFree(d->d_pages);
    d->d_pages = nullptr;
    if (r == VS_OK && extras->vecs) {
        void* dv = nullptr;
        uint32_t stride = 0;
        r = vs_index_array(ix, VS_ARR_VECS, &dv, &stride);
        if (r == VS_OK) r = vs_upload_rows(c, dv, (size_t)stride * 4, extras->vecs, (size_t)dd.dim_full * 4, (size_t)dd.dim_full * 4, dd.n);
    }
    if (r == VS_OK && extras->mean) r = vs_index_set_quantizer(ix, extras->mean, extras->m2, extras->count);
    if (r == VS_OK)
        r = vs_index_set_start_nodes(ix, desc->default_start, extras->label_start_labels, extras->label_start_nodes, desc->n_label_starts);
    if (r == VS_OK) r = vs_validate_graph(ix);
    if (r == VS_OK) r = vs_index_refresh_norms(ix);
    if (r != VS_OK) {
        vs_index_free(ix);
        return r;
    }
    *out = ix;
    return VS_OK;
}
