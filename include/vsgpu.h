/*
 * vsgpu.h — C ABI of libvsgpu.so: the MI355X (gfx950) implementation of pgvectorscale's StreamingDiskANN
 * *search* hot path (SBQ Hamming candidate scoring inside the greedy graph search + f32 rerank), i.e. the work
 * done between `amrescan` and `amgettuple` of the `diskann` access method.
 *
 * Citations are relative to /root/reference/pgvectorscale/src/access_method/ ("AM/").
 *
 * The reference has NO FFI seam on this path (storages / distance functions are statically dispatched Rust,
 * AM/storage.rs:41-142, AM/distance/mod.rs:8); its only C-ABI surface is the IndexAmRoutine filled in by
 * `amhandler` (AM/mod.rs:27-92).  This header therefore declares what a PGRX shim would bind from inside those
 * callbacks (the Rust `extern "C"` block is shown in INTEGRATION.md):
 *
 *   ambeginscan (AM/scan.rs:308-333) -> vs_beginscan        amrescan  (AM/scan.rs:335-367) -> vs_rescan
 *   amgettuple  (AM/scan.rs:369-436) -> vs_gettuple         amendscan (AM/scan.rs:438-456) -> vs_endscan
 *
 * plus the batched entry points (many backends' queries at once; what a GPU broker process would call) and the
 * individual kernels K1..K5 of SURVEY.md §2.
 *
 * Rules: plain C types only; every call returns 0 (VS_OK) or a negative vs_status and never throws / longjmps;
 * vs_last_error() gives a thread-local message (the Rust side turns it into pgrx::error!); the caller owns every
 * host buffer; the library owns device memory and pinned staging buffers; calls are synchronous unless their name
 * ends in _async; one thread per vs_ctx at a time.  There is NO CPU fallback: without a HIP device every compute
 * entry point fails with VS_ERR_HIP.
 */
#ifndef VSGPU_H
#define VSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_INVALID_NODE 0xFFFFFFFFu /* InvalidBlockNumber sentinel ending a neighbor list, AM/sbq/node.rs:260-285 */

enum vs_distance_type { VS_COSINE = 0, VS_L2 = 1, VS_IP = 2 }; /* AM/distance/mod.rs:11-15 */
/* memory_optimized (SbqSpeedupStorage: SBQ Hamming in the graph search + f32 rerank, the hot path of this library) or
 * plain (PlainStorage, AM/plain/storage.rs: the graph search scores candidates with the full-precision distance to the
 * vector stored in the node (its index slice when num_dimensions_to_index < num_dimensions, then the rows are resorted on the
 * full vectors); no label filters) */
enum vs_storage_type { VS_STORAGE_SBQ = 0, VS_STORAGE_PLAIN = 1 };

enum vs_status {
    VS_OK = 0,
    VS_ERR_INVALID = -1,  /* bad argument / malformed index (e.g. duplicate ids inside one neighbor list) */
    VS_ERR_HIP = -2,      /* HIP runtime error or no device */
    VS_ERR_OOM = -3,      /* device or pinned-host allocation failed */
    VS_ERR_CAPACITY = -4, /* a per-query search structure overflowed even after the automatic retries */
    VS_ERR_STATE = -5     /* call sequence error (e.g. vs_gettuple before vs_rescan) */
};

typedef struct vs_ctx vs_ctx;     /* one HIP device + streams + staging buffers */
typedef struct vs_index vs_index; /* one diskann index resident in HBM */
typedef struct vs_scan vs_scan;   /* one IndexScanDesc's opaque state (TSVScanState, AM/scan.rs:40-89) */

/* Index geometry = the MetaPage fields that parameterise the search (AM/meta_page.rs:179-210). */
typedef struct vs_index_desc {
    uint32_t n;             /* index nodes; node id = dense position (stands in for the ItemPointer)            */
    uint32_t dim_full;      /* num_dimensions                                                                 */
    uint32_t dim_index;     /* num_dimensions_to_index (<= dim_full)                                          */
    uint32_t bits;          /* num_bits_per_dimension; default 2 if dim_index < 900 else 1 (:312-323)         */
    uint32_t words;         /* W = ceil(dim_index*bits/64) u64 words per SBQ code (AM/sbq/quantize.rs:37-45) */
    uint32_t num_neighbors; /* R (default 50, :284-294)                                                       */
    uint32_t distance_type; /* vs_distance_type                                                               */
    uint32_t has_labels;    /* MetaPage.has_labels                                                            */
    uint32_t default_start; /* StartNodes.default_node or VS_INVALID_NODE for an empty graph                  */
    uint32_t n_label_starts;/* entries of StartNodes.labeled_nodes (AM/graph/start_nodes.rs:17-22)            */
    uint32_t storage_type;  /* vs_storage_type: which Storage the index was built with (AM/storage.rs)        */
} vs_index_desc;

/* Host-side flat arrays an exporter produces from the index relation's pages (SbqNode items, AM/sbq/node.rs:26-42;
 * SbqMeans chain, AM/sbq/mod.rs:62-121) and from the heap's vector column. */
typedef struct vs_index_host {
    const uint64_t* codes;     /* [n][words]            ArchivedSbqNode.bq_vector                             */
    const uint32_t* nbrs;      /* [n][nbr_stride]       neighbor_index_pointers, list ends at VS_INVALID_NODE */
    uint32_t nbr_stride;       /* row stride of `nbrs` in elements (>= num_neighbors)                         */
    const uint64_t* heap_tids; /* [n] (block<<16)|offset; offset==0 (InvalidOffsetNumber) => deleted tuple    */
    const float* vecs;         /* [n][dim_full] heap vectors (raw); NULL => no rerank possible (rescore must be 0) */
    const float* mean;         /* [dim_index]           SbqMeans.means                                        */
    const float* m2;           /* [dim_index]           SbqMeans.m2 (bits>1; may be NULL when bits==1)        */
    uint64_t count;            /* SbqMeans.count                                                              */
    const uint32_t* label_off; /* [n+1] CSR offsets into label_val (has_labels)                               */
    const int16_t* label_val;  /* per-node sorted, de-duplicated label sets (AM/labels/mod.rs:15-37)          */
    const int16_t* label_start_labels; /* [n_label_starts] sorted keys of StartNodes.labeled_nodes            */
    const uint32_t* label_start_nodes; /* [n_label_starts] their start nodes                                  */
} vs_index_host;

/* GreedySearchStats + TSVResponseIterator counters (AM/stats.rs:68-125, AM/scan.rs:461-472), summed over the call */
typedef struct vs_stats {
    uint64_t queries;
    uint64_t visited_nodes;                  /* "visits"       */
    uint64_t candidate_nodes;                /* "candidate"    */
    uint64_t quantized_distance_comparisons; /* "d_quantized"  */
    uint64_t full_distance_comparisons;      /* "d_full"       */
    uint64_t node_reads;                     /* "reads_index"  */
    uint64_t node_heap_reads;                /* "reads_heap"   */
    uint64_t next_calls;                     /* "next"         */
    uint64_t retries;                        /* capacity-overflow relaunches (ours; 0 in steady state) */
    uint64_t fallback_scans;                 /* scans the LDS-resident fast kernel handed to the general kernel (ours) */
    uint64_t fallback_visited_nodes;         /* their share of visited_nodes */
    uint64_t fallback_quantized_distance_comparisons; /* their share of quantized_distance_comparisons */
} vs_stats;

const char* vs_last_error(void);

/* Tuning options (launch variants, capacities, slab placement, diagnostics), process wide.  `name` is one of the
 * VS_* names listed there, `value` its text ("3", "0", a path ...; NULL unsets it again).  An option set here wins over the environment
 * variable of the same name; the environment itself is only SNAPSHOT — at the first lookup and again when the process's VS_* variables
 * change — never searched on the launch path.  No reference counterpart: these are the knobs of this implementation (the reference's
 * own GUCs, diskann.query_search_list_size and diskann.query_rescore, are arguments of vs_rescan / vs_search_batch).  Thread safe.
 * vs_get_option: 1 = set (copied to out, truncated to cap), 0 = set nowhere (out = ""). */
int vs_set_option(const char* name, const char* value);
int vs_get_option(const char* name, char* out, size_t cap);
const char* vs_version(void);

/* ---- context ------------------------------------------------------------------------------------------------- */
int vs_ctx_create(int device, vs_ctx** out);            /* staging ring: 2 x 32 MiB of pinned host memory */
/* the same with a staging ring of 2 x staging_bytes: the contexts of cursor lanes and other handles that move a few rows per call
 * (vs_broker_config.cursor_lanes creates up to 64 of them per broker) take 2 x 1 MiB instead of 2 x 32 MiB */
int vs_ctx_create_staging(int device, size_t staging_bytes, vs_ctx** out);
void vs_ctx_destroy(vs_ctx* ctx);
int vs_ctx_sync(vs_ctx* ctx);          /* wait for the compute stream */
void* vs_ctx_stream(vs_ctx* ctx);      /* the hipStream_t all kernels of this ctx are launched on */
int vs_ctx_device_name(vs_ctx* ctx, char* buf, size_t len);
int vs_ctx_mem_info(vs_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

/* per-kernel timing with HIP events recorded on the ctx stream around every launch of the batched-scan pipeline
 * (what bench.py's roofline figure is computed from).  kind: 0 prepare_queries, 1 search (the LDS-resident fast
 * kernel; the general kernel when the fast path is off), 2 rerank, 3 resort, 4 search fallback (general kernel re-running
 * the scans the fast kernel handed over), 5 flat SBQ scan (vs_scan_topk). */
typedef struct vs_profile {
    double ms[8];        /* accumulated kernel time per kind */
    uint64_t launches[8];
} vs_profile;
int vs_profile_enable(vs_ctx* ctx, int on);
int vs_profile_read(vs_ctx* ctx, vs_profile* out, int reset); /* synchronises the stream */

/* raw device buffers (so callers can keep inputs resident in HBM across calls) */
int vs_dev_alloc(vs_ctx* ctx, size_t bytes, void** out);
int vs_dev_free(vs_ctx* ctx, void* p);
int vs_dev_upload(vs_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);   /* pinned staging + hipMemcpyAsync */
int vs_dev_download(vs_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- index ---------------------------------------------------------------------------------------------------- */
/* Stage host arrays chunk-wise through pinned buffers to HBM (hipMemcpyAsync, double-buffered) and validate them. */
int vs_index_upload(vs_ctx* ctx, const vs_index_desc* desc, const vs_index_host* host, vs_index** out);
/* Allocate an index whose arrays are produced on the device (vs_datagen_*, vs_sbq_train, vs_sbq_quantize_corpus,
 * vs_build_graph); contents are undefined until filled. */
int vs_index_alloc(vs_ctx* ctx, const vs_index_desc* desc, int with_vecs, vs_index** out);
void vs_index_free(vs_index* idx);
/* A second handle on the SAME device arrays with its own search workspace, bound to `ctx` (another HIP stream of the same device):
 * batches submitted through different views run concurrently — the bandwidth-bound rerank of one under the latency-bound search
 * of the next.  The reference has one backend per scan and nothing to share (AM/scan.rs:308-333); this is the handle a GPU broker
 * gives each of its submission lanes.  The arrays stay owned by `src`: no upload / build / visibility change while a view has a
 * batch in flight, and every view is freed before `src`. */
int vs_index_view(vs_index* src, vs_ctx* ctx, vs_index** out);
/* The search workspace's randomly accessed part (the persistent grid's per-workgroup dedup tables and heap spill arrays) lives in ONE
 * grow-only slab per index, shared with its views (VS_WS_SLAB_MB, default 2 GiB per kind from 4M nodes; chosen by the first search among VS_WS_SLAB_CANDIDATES probed allocations).
 * vs_index_set_slab hands the library the CALLER's device memory for it instead (before the handle's first search; the memory stays the
 * caller's and must outlive the handle and the views made of it afterwards) — for a host that manages HBM itself, or one that has
 * measured where those regions run fastest: vs_ws_probe times the same request shapes (random 16-byte loads, 4-byte stores, 8-byte
 * loads in 24 private shares per CU) on any device region of at least 64 MiB.  No reference counterpart (PostgreSQL's buffer manager
 * places the reference's pages); placement moves k_search_fast by several percent at identical bytes ("State"). */
int vs_index_set_slab(vs_index* idx, void* d_mem, size_t bytes);
int vs_ws_probe(vs_ctx* ctx, void* d_mem, size_t bytes, uint32_t iters, float* ms_out);
/* ... and vs_ws_probe_mix the kernel's WHOLE request mix, with the neighbor rows and code rows read from this index's arrays and the
 * private state on the region (tables at its start, heap arrays in its second half): what the library itself uses to choose its slab
 * among VS_WS_SLAB_CANDIDATES allocations (default 8, spread over the free device memory by spacers) at the first search of an index of 4M nodes or more. */
int vs_ws_probe_mix(vs_index* idx, void* d_mem, size_t bytes, uint32_t iters, float* ms_out);
/* Chooses the slab NOW instead of inside the first search (after the index's arrays are uploaded / loaded; a no-op for an index that
 * has its slab, has none by the size rule, or is a view).  The probing holds its candidates and spacers for a fraction of a second:
 * at most VS_WS_SLAB_PROBE_PCT (default 50) per cent of the device memory that is free at that moment and never the last
 * VS_WS_SLAB_KEEP_FREE_MB (default 12 GiB) — a host that shares the device calls this at index load, when nobody else allocates. */
int vs_index_prepare_workspace(vs_index* idx);
int vs_index_get_desc(const vs_index* idx, vs_index_desc* out);
enum vs_array { VS_ARR_CODES = 0, VS_ARR_NBRS = 1, VS_ARR_TIDS = 2, VS_ARR_VECS = 3, VS_ARR_MEAN = 4, VS_ARR_M2 = 5,
                VS_ARR_VNORM = 6, VS_ARR_LABEL_OFF = 7, VS_ARR_LABEL_VAL = 8 };
/* device pointer + row stride (in elements) of one of the index arrays */
int vs_index_array(const vs_index* idx, int which, void** dev_ptr, uint32_t* row_stride);
int vs_index_set_quantizer(vs_index* idx, const float* mean, const float* m2, uint64_t count);
int vs_index_set_start_nodes(vs_index* idx, uint32_t default_start, const int16_t* labels, const uint32_t* nodes, uint32_t n);
int vs_index_set_labels(vs_index* idx, const uint32_t* label_off, const int16_t* label_val);
/* Heap visibility under the scan's snapshot, one byte per node (0 = index_fetch_tuple finds no tuple this snapshot can see).
 * Replaces the `None` arm of get_full_distance_for_resort (AM/sbq/storage.rs:313-317, AM/plain/storage.rs:181-184, reached
 * from next_with_resort, AM/scan.rs:258-272): with query_rescore > 0 such a candidate is fetched (counted as a heap read and a
 * full-distance comparison) and dropped BEFORE it enters the rescore window; with query_rescore = 0 the access method does not
 * look at the heap (the executor does) and the mask is ignored.  `visible` is a host array of n bytes, NULL = every tuple
 * visible (the default).  The mask applies to every scan started afterwards (a PGRX shim sets it per snapshot / batch). */
int vs_index_set_visibility(vs_index* idx, const uint8_t* visible);
/* the same with the mask already in device memory (n bytes, kept by the caller until replaced; NULL clears it) */
int vs_index_set_visibility_dev(vs_index* idx, const uint8_t* d_visible);
/* Shared launches (vs_broker / vs_shm) serve backends with DIFFERENT snapshots, so one mask per index is not enough there: the
 * index keeps up to VS_MAX_SNAPSHOTS - 1 masks (ids 1 .. VS_MAX_SNAPSHOTS - 1; `visible` = n host bytes, NULL drops the mask) and
 * a launch runs under the one its scans name; id 0 = every tuple visible.  vs_index_snapshot_use makes snapshot `id` the
 * mask of the launches that follow and returns the mask that was in force (restore it with vs_index_set_visibility_dev).
 * Same threading rule as every index call: with a broker attached only its dispatcher may call these — clients go through
 * vs_broker_snapshot_put / vs_shm_server_snapshot_put. */
#define VS_MAX_SNAPSHOTS 16
int vs_index_snapshot_put(vs_index* idx, uint32_t snapshot, const uint8_t* visible);
int vs_index_snapshot_use(vs_index* idx, uint32_t snapshot, const uint8_t** previous /* may be NULL */);
/* (for the lanes of a broker) a view takes over the snapshot masks its source holds right now — the masks stay the source's */
int vs_index_snapshot_share(vs_index* view, const vs_index* src);
int vs_index_device(const vs_index* idx); /* the HIP device of the index's context */
/* 1 when the label masks of every node's neighbors are cached next to the neighbor rows: label-filtered scans on an index of
 * <= 64 distinct labels and more than 8M nodes build the cache on first use when device memory allows (+6.7 % at 20M, +0.6 % at
 * 5M, where the masks are cache resident anyway); VS_F_NBRMASK=1 / 0 forces it on / off */
int vs_index_has_neighbor_masks(const vs_index* idx);
int vs_index_get_quantizer(const vs_index* idx, float* mean, float* m2, uint64_t* count);
/* copy index arrays back to host (tests / cpu_baseline leg); any pointer may be NULL */
int vs_index_download(const vs_index* idx, uint64_t* codes, uint32_t* nbrs /*[n][num_neighbors]*/, uint64_t* heap_tids,
                      float* vecs, uint32_t row_begin, uint32_t row_count);
/* (re)compute the per-node cosine divisor cache from the vector column (exact preprocess_cosine semantics,
 * AM/distance/mod.rs:225-253); called by upload automatically */
int vs_index_refresh_norms(vs_index* idx);
/* mark heap tuples deleted (what ambulkdelete does to heap_item_pointer, AM/vacuum.rs:24-78) */
int vs_index_mark_deleted(vs_index* idx, const uint32_t* nodes, uint32_t n);

/* ---- index relation pages -> vs_index_host (the exporter the arrays above come from) ------
 * The reference reaches a node through the buffer manager, one page pin per neighbor (ItemPointer::read_bytes,
 * util/mod.rs:152-155; ReadablePage::get_item_unchecked, util/page.rs:270-283; rkyv::archived_root,
 * pgvectorscale_derive/src/lib.rs:35-40).  Here the blocks of the index relation's main fork are handed over in bulk,
 * in block order (from ReadBufferExtended copies, or straight from the relation's segment files after a CHECKPOINT);
 * every SbqNode item (AM/sbq/node.rs:26-42) is decoded on the host cores into the flat arrays, neighbor ItemPointers
 * become dense node ids (node id = SbqNode items on earlier blocks + offset - 1), and vs_index_upload streams the
 * result to HBM.  Host-only code: these calls work without a device. */
#define VS_BLCKSZ 8192u
enum vs_page_type { /* PageType, util/page.rs:28-39 */
    VS_PAGE_META_V1 = 0, VS_PAGE_NODE = 1, VS_PAGE_PQ_QUANTIZER_DEF = 2, VS_PAGE_PQ_QUANTIZER_VECTOR = 3,
    VS_PAGE_SBQ_MEANS_V1 = 4, VS_PAGE_SBQ_NODE = 5, VS_PAGE_META_V2 = 6, VS_PAGE_SBQ_MEANS = 7, VS_PAGE_META = 8 };

/* Byte offsets of the fields inside the archived root object of an SbqNode item (the last root_size bytes of the
 * item).  rkyv 0.7 archives ClassicSbqNode / LabeledSbqNode as repr(Rust) structs of four 8-byte, 4-aligned fields;
 * vs_node_layout_default() assumes declaration order.  Nothing in the reference pins that order, so a PGRX shim
 * should pass core::mem::offset_of!(ArchivedLabeledSbqNode, ...) values instead. */
typedef struct vs_node_layout {
    uint32_t root_size;                   /* size_of::<ArchivedClassicSbqNode>() = 32                       */
    uint32_t off_heap_item_pointer;       /* ArchivedItemPointer {u32 block_number, u16 offset} (util/mod.rs:17-23) */
    uint32_t off_bq_vector;               /* ArchivedVec<u64>  = {i32 relative offset, u32 len}              */
    uint32_t off_neighbor_index_pointers; /* ArchivedVec<ArchivedItemPointer>                                */
    uint32_t off_labels;                  /* ArchivedLabelSet = ArchivedVec<i16> (LabeledSbqNode); 0xFFFFFFFF otherwise */
} vs_node_layout;
int vs_node_layout_default(int has_labels, vs_node_layout* out);

typedef struct vs_pages_info {
    uint32_t n_blocks;
    uint32_t n_nodes;          /* -> vs_index_desc.n                                                          */
    uint32_t words;            /* bq_vector.len() of every node -> vs_index_desc.words                        */
    uint32_t num_neighbors;    /* neighbor_index_pointers.len() of every node -> vs_index_desc.num_neighbors  */
    uint32_t has_labels;
    uint32_t n_deleted;        /* nodes whose heap_item_pointer.offset == InvalidOffsetNumber                 */
    uint64_t n_label_vals;
    uint32_t pages_by_type[9]; /* indexed by vs_page_type                                                     */
    uint32_t new_pages;        /* all-zero blocks (PageIsNew)                                                 */
    uint32_t meta_magic;       /* MetaPageHeader (block 0, item 1; AM/meta_page.rs:166-174), 0 if block 0 is no meta page */
    uint32_t meta_version;
} vs_pages_info;

typedef struct vs_pages vs_pages; /* reader over the main fork of one diskann index relation */
/* has_labels = MetaPage.has_labels (selects LabeledSbqNode); layout NULL = vs_node_layout_default; threads 0 = all cores (<= 32) */
int vs_pages_open(uint32_t page_size, int has_labels, const vs_node_layout* layout, uint32_t threads, vs_pages** out);
/* `plain` storage (PageType::Node pages of PlainNode items, AM/plain/node.rs:15-22): the same reader, with vs_pages_info.words =
 * the dimensions of PlainNode.vector and vs_index_host.vecs = those vectors (the cosine-normalised index slice the graph
 * distances are computed on, AM/plain/storage.rs:239-247) instead of codes; no labels (AM/plain/storage.rs:262).  layout NULL =
 * vs_plain_layout_default (off_bq_vector names the vector field, off_labels is unused). */
int vs_plain_layout_default(vs_node_layout* out);
int vs_pages_open_plain(uint32_t page_size, const vs_node_layout* layout, uint32_t threads, vs_pages** out);
/* append blocks first_block .. first_block+n_blocks-1 (must continue where the previous call stopped; the bytes are
 * not referenced after the call returns).  A failing call leaves the reader unchanged. */
int vs_pages_add(vs_pages* p, uint32_t first_block, const void* pages, uint32_t n_blocks);
/* translate neighbor ItemPointers to node ids, check the MetaPageHeader magic; info may be NULL */
int vs_pages_finish(vs_pages* p, vs_pages_info* info);
/* borrowed pointers (valid until vs_pages_close): codes, nbrs, nbr_stride, heap_tids, label_off, label_val; the
 * caller fills vecs (heap column), mean / m2 / count (vs_pages_sbq_means) and the start-node arrays, then calls
 * vs_index_upload */
int vs_pages_host(const vs_pages* p, vs_index_host* host);
/* IndexPointer <-> node id (start nodes of the MetaPage, AM/graph/start_nodes.rs:17-22) */
int vs_pages_node_of(const vs_pages* p, uint32_t block, uint32_t offset, uint32_t* node);
int vs_pages_item_pointer_of(const vs_pages* p, uint32_t node, uint32_t* block, uint32_t* offset);
/* ChainItemIterator (util/chain.rs:159-185): concatenated payload of the chain that starts at (block, offset);
 * page_type VS_PAGE_SBQ_MEANS or VS_PAGE_META; buf may be NULL to query *len */
int vs_pages_read_chain(const vs_pages* p, uint32_t block, uint32_t offset, int page_type, void* buf, size_t cap, size_t* len);
/* SbqMeans::load (AM/sbq/mod.rs:85-121) at MetaPage.quantizer_metadata: chained SbqMeans or single-item SbqMeansV1 */
int vs_pages_sbq_means(const vs_pages* p, uint32_t block, uint32_t offset, float* mean, float* m2, uint32_t dim_cap,
                       uint32_t* dim, uint64_t* count);
void vs_pages_close(vs_pages* p);
/* keep only the block table (page type / item count per block) and the metadata pages: node items are decoded elsewhere */
int vs_pages_headers_only(vs_pages* p);
int vs_pages_block_table(const vs_pages* p, const uint32_t** blk_base, const uint32_t** blk_cnt, uint32_t* n_blocks);

/* ---- the MetaPage body (AM/meta_page.rs:176-210): item 2 of block 0, a chained rkyv archive (store / load, :344-378) ----
 * Byte offsets of the fields inside ArchivedMetaPage (the last root_size bytes of the chain's payload).  rkyv 0.7 archives are
 * repr(Rust): vs_meta_layout_default() lays the fields out in declaration order with C alignment; a PGRX shim passes
 * core::mem::offset_of!(ArchivedMetaPage, ...) instead (oracle/ref_kat.rs prints them).  The archived forms
 * restated from rkyv 0.7.43: ArchivedString = 8 bytes (<= 7 bytes inline, else {u32 len, i32 offset}); ArchivedOption =
 * {u8 tag, T at T's alignment}; ArchivedStartNodes = {ArchivedItemPointer default_node, ArchivedBTreeMap<i16, ItemPointer>
 * {u32 len, i32 root}} with B-tree nodes {u16 meta (bit 15: inner), u32 size, i32 ptr} + LeafNodeEntry {i16 key, ItemPointer}
 * / InnerNodeEntry {i32 ptr, i16 key}; relative pointers count from the position of the pointer itself. */
typedef struct vs_meta_layout {
    uint32_t root_size;                    /* size_of::<ArchivedMetaPage>() = 80 */
    uint32_t off_magic_number, off_version, off_extension_version_when_built, off_distance_type, off_num_dimensions,
        off_num_dimensions_to_index, off_bq_num_bits_per_dimension, off_storage_type, off_num_neighbors, off_search_list_size,
        off_max_alpha, off_start_nodes, off_quantizer_metadata, off_has_labels;
} vs_meta_layout;
int vs_meta_layout_default(vs_meta_layout* out);
typedef struct vs_meta_page {
    uint32_t magic_number, version;
    char extension_version_when_built[64];  /* NUL terminated (longer strings are cut)                                     */
    uint32_t distance_type;                 /* DistanceType as stored (u16): 0 cosine, 1 l2, 2 inner product                 */
    uint32_t num_dimensions, num_dimensions_to_index, bq_num_bits_per_dimension;
    uint32_t storage_type;                  /* StorageType as stored (u8): 0 plain, 1 (retired), 2 SbqCompression           */
    uint32_t num_neighbors, search_list_size;
    double max_alpha;
    uint32_t has_start_nodes;               /* Option<StartNodes> is Some                                                    */
    uint32_t default_start_block, default_start_offset;   /* StartNodes.default_node                                        */
    uint32_t n_labeled_start_nodes;         /* StartNodes.labeled_nodes.len()                                                */
    uint32_t quantizer_block, quantizer_offset;            /* quantizer_metadata (SbqMeans chain; vs_pages_sbq_means)        */
    uint32_t has_labels;
} vs_meta_page;
/* rkyv::from_bytes::<MetaPage> over `bytes` (the chain payload).  The labeled start nodes come out in key order:
 * start_labels / start_blocks / start_offsets hold `cap` entries each (any may be NULL); more entries than cap is an error
 * unless all three are NULL (query n_labeled_start_nodes first).  layout NULL = vs_meta_layout_default.  Host-only. */
int vs_meta_page_decode(const void* bytes, size_t len, const vs_meta_layout* layout, vs_meta_page* out, int16_t* start_labels,
                        uint32_t* start_blocks, uint32_t* start_offsets, uint32_t cap);
/* MetaPage::fetch for a reader (after vs_pages_finish): reads the chain at (0, 2), decodes it and fills the geometry of
 * `desc` (n from the pages, distance / dims / bits / words / R / storage / has_labels from the MetaPage, default_start and the
 * labeled start nodes translated to node ids; an index without rows has default_start = VS_INVALID_NODE).  start_labels /
 * start_nodes: cap entries (may be NULL to query desc->n_label_starts).  meta may be NULL.  Only PageType::Meta (version 3)
 * relations: MetaV1 / MetaV2 pages are what the reference itself rewrites on first use. */
int vs_pages_meta(const vs_pages* p, const vs_meta_layout* layout, vs_meta_page* meta, vs_index_desc* desc, int16_t* start_labels,
                  uint32_t* start_nodes, uint32_t cap);

/* ---- the heap's vector column (vs_heap.cpp; host-only) ----------------------------------------------------------------------
 * vs_index_host.vecs — what the rescore window reads through table_index_fetch_tuple + slot_getattr + pg_detoast_datum_copy per
 * candidate (UT/table_slot.rs:19-42, AM/pg_vector.rs:125-135, AM/sbq/storage.rs:304-328) — is staged once, in bulk, from the
 * pages of the table the index points into: the heap's main fork streams past in block order (vs_heap_add), then its TOAST
 * relation's (vs_heap_toast_add; a 768-dimensional vector is 3 KB and lives there, chunked).  For every index node the tuple
 * its heap TID names is deformed up to the vector attribute exactly as heap_deform_tuple does (null bitmap, alignment padding,
 * 1-byte / 4-byte / external / pglz-compressed varlena forms; LP_REDIRECT line pointers are followed) and the float4s land in
 * row `node` of out_vecs.  Visibility is not decided here (vs_index_set_visibility). */
typedef struct vs_heap_attr {  /* pg_attribute.attlen / attalign of one column of the table, in attnum order */
    int16_t attlen;   /* > 0 fixed length, -1 varlena, -2 cstring */
    char attalign;    /* 'c' 1, 's' 2, 'i' 4, 'd' 8 */
} vs_heap_attr;
typedef struct vs_heap_info {
    uint32_t n_nodes, heap_blocks, toast_blocks;
    uint32_t n_inline;            /* vectors found inside their heap tuple                                            */
    uint32_t n_external;          /* vectors assembled from TOAST chunks                                              */
    uint32_t n_deleted;           /* index tuples with an invalid heap offset (vacuumed): no row, left zero           */
    uint32_t n_null;              /* the row exists, its vector column is NULL                                        */
    uint32_t n_dead_line_pointer; /* the TID names an unused / dead line pointer                                      */
    uint32_t n_not_found;         /* the TID lies beyond the blocks that were added                                   */
    uint32_t n_toast_incomplete;  /* an external vector whose chunks did not all arrive                               */
    uint64_t n_chunks;            /* TOAST chunk rows consumed                                                        */
} vs_heap_info;
typedef struct vs_heap vs_heap;
/* attrs / natts: the table's tuple descriptor; vector_attno: the indexed column (1-based attnum, IndexInfo.ii_IndexAttrNumbers);
 * heap_tids [n]: ArchivedSbqNode.heap_item_pointer of every node, (block << 16) | offset, as vs_pages_host returns them;
 * out_vecs [n][out_stride] floats (out_stride >= dim), zeroed here, filled by the calls below; kept by the caller. */
int vs_heap_open(uint32_t page_size, const vs_heap_attr* attrs, uint32_t natts, uint32_t vector_attno, uint32_t dim,
                 const uint64_t* heap_tids, uint32_t n, float* out_vecs, uint32_t out_stride, vs_heap** out);
int vs_heap_add(vs_heap* h, uint32_t first_block, const void* pages, uint32_t n_blocks);        /* heap main fork, in block order */
int vs_heap_toast_add(vs_heap* h, uint32_t first_block, const void* pages, uint32_t n_blocks);  /* then the TOAST relation's       */
/* found [n] (may be NULL): 1 where the node's vector was read.  Everything else is counted in info. */
int vs_heap_finish(vs_heap* h, vs_heap_info* info, uint8_t* found);
void vs_heap_close(vs_heap* h);

/* The same, decoded ON the device (vs_pages_dev.hip): the blocks are copied to HBM as they are (pinned ring,
 * hipMemcpyAsync), the host only reads the page headers on the way past, and one kernel (a wave per node page) walks the
 * line pointers and rkyv relative pointers and writes codes / neighbor ids / heap tids into the index arrays; label sets
 * (desc->has_labels, LabeledSbqNode) take a count pass and a copy pass over the same staged pages.  memory_optimized
 * indexes.  layout NULL = the default layout for desc->has_labels.  n_blocks_total = RelationGetNumberOfBlocks (the raw pages stay in HBM until the build). */
typedef struct vs_pages_dev vs_pages_dev;
int vs_pages_dev_open(vs_ctx* ctx, uint32_t page_size, const vs_node_layout* layout, uint32_t n_blocks_total, vs_pages_dev** out);
int vs_pages_dev_add(vs_pages_dev* d, uint32_t first_block, const void* pages, uint32_t n_blocks); /* in block order */
int vs_pages_dev_node_of(const vs_pages_dev* d, uint32_t block, uint32_t offset, uint32_t* node);
int vs_pages_dev_sbq_means(const vs_pages_dev* d, uint32_t block, uint32_t offset, float* mean, float* m2, uint32_t dim_cap,
                           uint32_t* dim, uint64_t* count);
/* MetaPage::fetch on the metadata pages the host kept (after the last vs_pages_dev_add): as vs_pages_meta */
int vs_pages_dev_meta(vs_pages_dev* d, const vs_meta_layout* layout, vs_meta_page* meta, vs_index_desc* desc, int16_t* start_labels,
                      uint32_t* start_nodes, uint32_t cap);
/* desc: the MetaPage fields (vs_pages_dev_meta, or by hand: n is taken from the pages, default_start is a node id from vs_pages_dev_node_of);
 * extras: vecs / mean / m2 / count / label_start_labels / label_start_nodes (node ids); frees the raw pages */
int vs_pages_dev_build(vs_pages_dev* d, const vs_index_desc* desc, const vs_index_host* extras, vs_pages_info* info, vs_index** out);
void vs_pages_dev_close(vs_pages_dev* d);

/* ---- K4: SBQ quantisation of queries (SbqQuantizer::quantize, AM/sbq/quantize.rs:52-102) --------------------- */
/* q: host [nq][dim_index], already cosine-normalised by the caller if applicable; out: host [nq][words] */
int vs_quantize(vs_index* idx, const float* q, uint32_t nq, uint64_t* out_codes);

/* ---- K1: XOR+popcount of query codes against gathered node codes (distance_xor_optimized, AM/distance/mod.rs:266-323)
 * ids/off: CSR — query i is scored against ids[off[i] .. off[i+1]).  out: one u32 per id. */
int vs_hamming_gather(vs_index* idx, const uint64_t* qcodes, const uint32_t* ids, const uint32_t* off, uint32_t nq,
                      uint32_t* out);

/* ---- K2: full-precision rerank distances (get_full_distance_for_resort, AM/sbq/storage.rs:304-328 with
 * distance_l2 / distance_cosine / distance_inner_product in the reference's AVX2 accumulation order,
 * AM/distance/mod.rs:325-435).  q_full: host [nq][dim_full] RAW queries (cosine-normalised inside, as
 * PgVector::from_datum does, AM/pg_vector.rs:153-155). */
int vs_rerank(vs_index* idx, const float* q_full, const uint32_t* ids, const uint32_t* off, uint32_t nq, float* out);

/* ---- K5: flat scan — top-k of Hamming distance over ALL codes, order (hamming asc, node id asc) ------------- */
int vs_scan_topk(vs_index* idx, const uint64_t* qcodes, uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham);
/* The same with the scan's predicate: only rows whose label set overlaps the query's key (qlabels / qlabel_off: CSR per query, an
 * empty key filters nothing — LabelSetView::overlaps, AM/labels/mod.rs:124-142, AM/scan.rs:189) and, with live_only, whose heap
 * tuple is not deleted (AM/scan.rs:231-234) are ranked: the EXACT filtered SBQ top-k, what the label-filtered graph walk
 * approximates.  An explicit entry point (ground truth of the SBQ ranking, exact mode for very selective keys); the access-method
 * callbacks never switch to it on their own, because their rows must be the reference's graph walk's. */
int vs_scan_topk_filtered(vs_index* idx, const uint64_t* qcodes, const int16_t* qlabels, const uint32_t* qlabel_off, int live_only,
                          uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham);

/* ---- K3 (+K2 + resort window): batched scans ------------------------------------------------------------------
 * For each query: exactly the rows the reference returns from the first k amgettuple calls after amrescan
 * (TSVResponseIterator::next_with_resort, AM/scan.rs:244-305) with GUCs diskann.query_search_list_size =
 * search_list_size and diskann.query_rescore = rescore (AM/guc.rs:3-4).
 *   queries   host [nq][dim_full] raw f32
 *   qlabels / qlabel_off: CSR of the smallint[] scan keys, qlabel_off == NULL => no scan key on any query.  A key may hold any
 *             number of labels, as in the reference (LabelSet, AM/labels/mod.rs:19-37): up to 64 distinct labels ride in on-chip
 *             memory, a wider key is read from global memory by the general kernel (the LDS-resident kernel hands such scans
 *             over: vs_stats.fallback_scans).  Only the label-aware BUILD limits a node to 64 labels (VS_ERR_INVALID beyond).
 *   out_ids   [nq][k] node ids (VS_INVALID_NODE past the end of a scan), out_tids [nq][k] heap TIDs (may be NULL),
 *   out_dist  [nq][k] reranked f32 distance (NaN when rescore == 0) (may be NULL)                               */
int vs_search_batch(vs_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                    uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids,
                    float* out_dist, vs_stats* stats);
/* the raw SBQ-ordered stream: first m results of TSVResponseIterator::next (AM/scan.rs:210-242) */
int vs_stream_batch(vs_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                    uint32_t search_list_size, uint32_t m, uint32_t* out_ids, uint32_t* out_ham, vs_stats* stats);
/* device-resident variant: d_queries [nq][dim_full] and the outputs are device pointers (vs_dev_alloc); label keys
 * must already be sorted + de-duplicated per query.  Enqueues on the ctx stream and returns; vs_ctx_sync() (or
 * vs_search_batch_dev_finish for the stats / error check) completes it. */
int vs_search_batch_dev(vs_index* idx, const float* d_queries, const int16_t* d_qlabels, const uint32_t* d_qlabel_off,
                        uint32_t nq, uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* d_out_ids,
                        uint64_t* d_out_tids, float* d_out_dist);
int vs_search_batch_dev_finish(vs_index* idx, vs_stats* stats);

/* ---- launch-variant selection (ours; the reference has no counterpart — its only query-time knobs are the two GUCs above,
 * AM/guc.rs:3-43, and they stay the caller's).  The search kernel exists in several EXACT instantiations that differ only in
 * how a scan keeps its private state (dedup tables cleared per scan, a written-bucket bitmap, an occupancy bit per slot; for small
 * scans: dedup table in LDS or not; DESIGN.md 3.1 / 4, docs/LAB_NOTEBOOK.md 10b).  Which is fastest depends on the index size and on the box, so it is measured
 * where it runs: vs_index_autotune runs every applicable variant on the caller's own device-resident batch (the arguments of
 * vs_search_batch_dev), `reps` timed steps each after one warm-up, holds every row, every distance bit and every work counter of
 * a variant to the library default's on the same batch, DISQUALIFIES a variant that differs anywhere (rows_identical = 0) and
 * makes the fastest qualified one the index's choice when it beats the default by at least 3 % and does so again on a second timing.  VS_F_* environment
 * variables still override the choice per call.  report (may be NULL) receives one entry per variant, the default first.
 * A caller that cannot afford a misbehaving kernel in its own process probes the variants in a child process first
 * (pgvectorscale_amd/tune_probe.py: a small index of the same code width, every variant, a hard timeout) and passes the ones
 * that did not come back clean as `skip`. */
typedef struct vs_tune_entry {
    char name[40];
    float step_ms;          /* best device time of one step (search + rerank + rescore window), HIP events on the ctx stream */
    float search_ms;        /* of which the search kernel(s) of the first attempt */
    uint32_t applicable;    /* 0: the variant does not exist for this index / operating point (launched the default's kernel) */
    uint32_t rows_identical;/* ids, distance bits and work counters equal the default's on the whole batch */
    uint32_t chosen;
    int32_t error;          /* VS_OK, or the error the variant's launch returned (it is then not eligible) */
} vs_tune_entry;
int vs_index_autotune(vs_index* idx, const float* d_queries, const int16_t* d_qlabels, const uint32_t* d_qlabel_off,
                      uint32_t nq, uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t reps,
                      const char* skip /* comma-separated variant names not to launch at all, or NULL */,
                      vs_tune_entry* report, uint32_t report_cap, uint32_t* n_report);
/* name the variant by hand ("default", or a name vs_index_autotune reports) / read the current choice */
int vs_index_set_variant(vs_index* idx, const char* name);
int vs_index_get_variant(vs_index* idx, char* buf, size_t len);

/* ---- the amrescan / amgettuple mirror (one row at a time) ----------------------------------------------------
 * A scan keeps what TSVScanState keeps between amgettuple calls (lsr + resort_buffer, AM/scan.rs:162-174) on the device and
 * CONTINUES the beam search when the executor asks for more rows (AM/scan.rs:370-405): it is never run again from the start
 * (except after a capacity overflow, vs_stats.retries), and the rows prefetched ahead of the executor stay within ~6 % of
 * what was pulled.  One thread per scan; scans of one index share its vs_ctx stream. */
int vs_beginscan(vs_index* idx, vs_scan** out);                                  /* ambeginscan */
/* query == NULL is the SQL-NULL query (zero vector, labels ignored; AM/labels/mod.rs:214-216).
 * has_label_key: nkeys == 1 (sets xs_recheck, AM/scan.rs:350-352); labels may be unsorted / contain duplicates. */
int vs_rescan(vs_scan* scan, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
              uint32_t search_list_size, uint32_t rescore);                       /* amrescan */
/* returns 1 and fills the outputs for the next row, 0 at end of scan, <0 on error */
int vs_gettuple(vs_scan* scan, uint64_t* heap_tid, uint32_t* node, float* dist); /* amgettuple */
/* optional hint: the executor will pull `rows` rows in all (a LIMIT it knows): they are produced by one continuation instead
 * of several.  Fewer rows become available when the scan ends first; never an error to ask for more than exist. */
int vs_scan_prefetch(vs_scan* scan, uint32_t rows);
int vs_scan_xs_recheck(const vs_scan* scan);
/* GreedySearchStats as the reference's scan holds them after the amgettuple calls made so far (AM/stats.rs:68-125,
 * AM/scan.rs:461-472): the counters are recorded per emitted row, so prefetched rows do not show (zero on a broker scan). */
int vs_scan_get_stats(const vs_scan* scan, vs_stats* out);
/* what the device really did for the scan since vs_rescan (prefetched rows and restarts included); launches may be NULL */
int vs_scan_get_work(const vs_scan* scan, vs_stats* out, uint32_t* launches);
void vs_endscan(vs_scan* scan);                                                   /* amendscan */

/* ---- many streamed scans continued TOGETHER (vs_scanpool.cpp) ------------------------------------------------
 * A single cursor continues its scan with a launch of its own: one wave on the chip and half a millisecond of launch / copy
 * overhead per continuation, and 64 backends streaming at once are 64 such launches per round.  A scan pool keeps what
 * TSVResponseIterator keeps per scan (lsr + resort_buffer, AM/scan.rs:162-174) for up to `capacity` scans in pooled device arrays,
 * so that ONE resumed launch (plus one rerank launch) continues every scan that asked for rows in this round — what a GPU broker
 * does with the amgettuple calls of many backends that arrive together.  Rows and GreedySearchStats are the single cursor's, row
 * for row.  All scans of a pool share the GUCs (search_list_size, rescore) and the visibility mask in force; scan keys are per
 * scan (slot).  kmax = most rows one fetch asks for; rows_cap = most stream rows a pooled scan may produce (0: 4096) — a scan that
 * needs more fails with VS_ERR_CAPACITY in its out_rows entry and is continued by a vs_scan of its own.  One thread at a time. */
typedef struct vs_scan_pool vs_scan_pool;
int vs_scanpool_create(vs_index* idx, uint32_t capacity, uint32_t search_list_size, uint32_t rescore, uint32_t kmax, uint32_t rows_cap,
                       vs_scan_pool** out);
void vs_scanpool_free(vs_scan_pool* pool);
int vs_scanpool_rescan(vs_scan_pool* pool, uint32_t slot, const float* query, const int16_t* labels, uint32_t n_labels,
                       int has_label_key);                                                                   /* amrescan of one slot */
int vs_scanpool_endscan(vs_scan_pool* pool, uint32_t slot);
/* amgettuple x k for the n listed slots at once: out_*[i][0 .. out_rows[i]) are the next rows of slots[i] (fewer than k: its scan
 * has ended; a negative VS_ERR_*: that scan failed, the others are served).  out_tids / out_ids / out_dist may be NULL. */
int vs_scanpool_fetch(vs_scan_pool* pool, const uint32_t* slots, uint32_t n, uint32_t k, uint64_t* out_tids, uint32_t* out_ids,
                      float* out_dist, int32_t* out_rows);
int vs_scanpool_get_stats(const vs_scan_pool* pool, uint32_t slot, vs_stats* out);  /* as vs_scan_get_stats */
int vs_scanpool_get_work(const vs_scan_pool* pool, uint64_t* launches, uint64_t* rounds);  /* shared search launches / rounds so far */

/* ---- coalescing concurrent scans into batched launches (vs_broker.cpp) -------------------
 * The reference serves one query per single-threaded backend (amcanparallel = false, AM/mod.rs:63); a GPU needs thousands
 * of scans per launch.  A broker owns the index (its dispatcher thread is the only thread that touches the vs_ctx); any
 * number of client threads call vs_broker_search(), which blocks until the scan's rows are ready.  The dispatcher
 * gathers requests for at most max_wait_us after the oldest one (or until max_batch are waiting) and runs every group
 * that shares (search_list_size, rescore, k, label key present, snapshot) as one vs_search_batch().  Heap visibility: scans of
 * different backends see different snapshots, so a request names the snapshot mask it runs under (vs_broker_search_snapshot;
 * masks are handed to the dispatcher with vs_broker_snapshot_put) and only scans of one snapshot share a launch; plain
 * vs_broker_search = snapshot 0 = every tuple visible.  The index-level mask of vs_index_set_visibility is not consulted by a
 * broker (it is left as it was).  In a PGRX deployment the queue lives in shared memory and the dispatcher is a background
 * worker. */
typedef struct vs_broker vs_broker;
typedef struct vs_broker_config {
    uint32_t max_batch;   /* scans per launch at most (0 = 8192)                                  */
    uint32_t max_wait_us; /* how long the oldest waiting scan may be held back to let others join */
    uint32_t cursor_lanes; /* 0 (default): the continuations of the scans' cursors (amgettuple past the shared first rows) run on
                            * the dispatcher thread, one at a time, between two shared launches.  n > 0: on n lanes — threads of the
                            * broker with a HIP stream and a view of the index each — so that n scans continue concurrently on the
                            * device and none of them waits behind a shared launch (a scan stays on its lane).  vs_shm_server:
                            * the streamed scans of client processes, by (client pid, scan id).  Cost per lane: a context
                            * with 2 x 1 MiB of pinned staging memory and the device workspace of the scans it continues.
                            * (VS_BROKER_LANES in the environment: the lane count of brokers created with 0 — how the test tier
                            * runs every broker test on lanes; leave it unset in production.)                         */
    uint32_t cursor_pool; /* (vs_shm_server) n > 0: the streamed scans of the client processes live in scan pools of n slots
                           * (vs_scanpool_*: one pool per (search_list_size, rescore, snapshot) in use, four at most) and the
                           * amgettuple continuations that arrive in one dispatcher round are served by SHARED launches — 64
                           * backends streaming at once cost about what one costs.  Scans a pool cannot take (no free pool for
                           * their GUCs, more than 4096 stream rows) fall back to a cursor of their own.  (VS_SHM_CURSOR_POOL in
                           * the environment: the pool size of servers created with 0 — test tier.)                          */
} vs_broker_config;
typedef struct vs_broker_stats {
    uint64_t batches;   /* vs_search_batch calls made                */
    uint64_t scans;     /* scans served                              */
    uint64_t max_batch; /* largest number of scans in one launch     */
    uint64_t tasks;     /* single-scan pieces of work run by the dispatcher between launches (cursor continuations) */
    uint64_t cursors;   /* (vs_shm_server) scan cursors the serving process holds open right now */
} vs_broker_stats;
int vs_broker_create(vs_index* idx, const vs_broker_config* cfg /* NULL = defaults */, vs_broker** out);
/* one scan: the rows of its first k amgettuple calls (as vs_search_batch).  query == NULL: the SQL-NULL query (label keys
 * ignored).  Thread safe; blocks.  out_tids / out_dist may be NULL. */
int vs_broker_search(vs_broker* b, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                     uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids,
                     float* out_dist);
/* the same under snapshot mask `snapshot` (0 = every tuple visible; an id without a mask fails with VS_ERR_STATE) */
int vs_broker_search_snapshot(vs_broker* b, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                              uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t snapshot, uint32_t* out_ids,
                              uint64_t* out_tids, float* out_dist);
/* hands a snapshot's mask (n host bytes; NULL drops it) to the dispatcher; returns when it is in place.  Thread safe. */
int vs_broker_snapshot_put(vs_broker* b, uint32_t snapshot, const uint8_t* visible);
int vs_broker_get_stats(vs_broker* b, vs_broker_stats* out);
vs_index* vs_broker_index(vs_broker* b);
/* the amrescan / amgettuple mirror on top of a broker: like vs_beginscan, but the scan's first 16 rows come out of a launch
 * shared with the scans of other threads (backends) — a LIMIT <= 16 never needs more — and an executor that keeps pulling gets a
 * cursor of its own on the device, opened, continued and released on the dispatcher thread (vs_broker_call): from then on the
 * scan is continued, never re-run, and vs_scan_get_stats is exact as for a direct scan (for a scan that only ever used the
 * shared launch it replays the scan on a cursor when asked).  vs_rescan / vs_gettuple / vs_endscan as usual; end a broker's
 * scans before vs_broker_destroy. */
int vs_beginscan_on_broker(vs_broker* b, vs_scan** out);
/* the visibility mask (vs_broker_snapshot_put) a scan on a broker runs under, from its next vs_rescan on (0 = every tuple
 * visible, the default).  A direct scan runs under the index's current mask instead. */
int vs_scan_set_snapshot(vs_scan* scan, uint32_t snapshot);
/* runs fn(arg) on the dispatcher thread between two launches and returns its result (the error text comes along): the way work
 * on ONE scan's device state reaches the only thread that may touch the index.  Thread safe; blocks. */
int vs_broker_call(vs_broker* b, int (*fn)(void*), void* arg);
void vs_broker_destroy(vs_broker* b); /* serves what is queued, then stops the dispatcher */

/* ---- the same across PROCESSES (vs_shm.cpp): PostgreSQL backends are processes, so the request queue is a POSIX shared-memory
 * segment (`name`, "/..."): one slot per in-flight scan (query vector, keys, GUCs, the k result rows, a futex).  The process that
 * owns the vs_ctx / vs_index creates the segment and runs the dispatcher (groups posted scans exactly like vs_broker); a client
 * needs no HIP: it maps the segment, posts a scan with vs_shm_client_search() and sleeps until its rows are in the slot.  In a
 * PGRX deployment the segment is a DSM segment, the futex a latch, the dispatcher a background worker. */
typedef struct vs_shm_server vs_shm_server;
typedef struct vs_shm_client vs_shm_client;
int vs_shm_server_create(vs_index* idx, const char* name, uint32_t nslots /* scans in flight at most (max_connections) */,
                         uint32_t kmax /* rows per scan at most */, const vs_broker_config* cfg /* NULL = defaults */,
                         vs_shm_server** out);
int vs_shm_server_get_stats(vs_shm_server* s, vs_broker_stats* out);
/* the scan pools of a server (cfg->cursor_pool != 0): out[0] = pools alive, out[1] = shared fetch rounds so far, out[2] = pools re-keyed
 * at the cap of four (an empty pool gives way to a new (search_list_size, rescore, snapshot) combination), out[3] = scans living in pools */
int vs_shm_server_pool_stats(vs_shm_server* s, uint64_t out[4]);
void vs_shm_server_destroy(vs_shm_server* s); /* fails what is still posted, unlinks the segment */
int vs_shm_client_open(const char* name, vs_shm_client** out);
uint32_t vs_shm_client_dim(const vs_shm_client* c); /* dim_full of the index behind the segment */
/* one scan: the rows of its first k amgettuple calls (as vs_search_batch).  query == NULL: the SQL-NULL query (label keys
 * ignored).  Blocks; one call at a time per client handle.  out_tids / out_dist may be NULL. */
int vs_shm_client_search(vs_shm_client* c, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                         uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids,
                         float* out_dist);
/* the same under snapshot mask `snapshot` of the serving process (vs_shm_server_snapshot_put; 0 = every tuple visible) */
int vs_shm_client_search_snapshot(vs_shm_client* c, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                                  uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t snapshot, uint32_t* out_ids,
                                  uint64_t* out_tids, float* out_dist);
/* amgettuple beyond the first rows, across processes: rows [skip, skip + k) of the scan this process calls `scan_id` (k <= kmax;
 * *n_rows < k: the scan has ended).  The serving process keeps a cursor on the device for the scan (lsr + resort_buffer of
 * AM/scan.rs:162-174) and continues it from request to request; every request carries the whole scan description, so a cursor
 * the server no longer has — or never had, because the first rows came from vs_shm_client_search — is opened and fast-forwarded
 * to `skip` (one replay) and continued from there.  vs_shm_client_end_scan drops the cursor (so does the death of the client). */
int vs_shm_client_fetch(vs_shm_client* c, uint64_t scan_id, const float* query, const int16_t* labels, uint32_t n_labels,
                        int has_label_key, uint32_t search_list_size, uint32_t rescore, uint32_t snapshot, uint32_t skip, uint32_t k,
                        uint32_t* out_ids, uint64_t* out_tids, float* out_dist, uint32_t* n_rows);
int vs_shm_client_end_scan(vs_shm_client* c, uint64_t scan_id);
/* serving process: a snapshot's mask (n host bytes, copied; NULL drops it); in place before the next group is formed */
int vs_shm_server_snapshot_put(vs_shm_server* s, uint32_t snapshot, const uint8_t* visible);
void vs_shm_client_close(vs_shm_client* c);

/* ---- multi-GPU (BASELINE.json north_star: "query batches shard embarrassingly across the 8 GPUs of one node
 * with RCCL over xGMI used only for a final top-k gather").  The reference has nothing to mirror here — AM/mod.rs:63
 * amcanparallel = false, one backend runs one scan (AM/scan.rs:308-456) — so this is the seam a PGRX host binds when one
 * PostgreSQL instance fronts several devices.  The path shards by QUERY: every device holds the whole index, device g takes a
 * contiguous block of the batch (vs_shard_range), no collective touches the data path, one gather of the [nq][k] blocks ends
 * the step.  Two deployments, neither needs torch:
 *   vs_multi_*  ONE process owns N devices (a broker / background worker): the index is replicated device to device
 *               (hipMemcpyPeerAsync over xGMI: one build or upload, N - 1 copies), one host thread per device runs its shard of a
 *               host batch and writes its rows into the caller's buffers at the shard's offset.
 *   vs_comm_*   one PROCESS per device: RCCL (ncclAllGather of the id / distance blocks; ncclBroadcast to replicate an index
 *               from the rank that holds it).  librccl is dlopen'ed by the first vs_comm_* call (VS_RCCL_LIB overrides the
 *               name); the 128-byte communicator id travels between the processes by the host's own means. */
int vs_shard_range(uint32_t nq_total, uint32_t world, uint32_t rank, uint32_t* begin, uint32_t* end); /* blocks differ by <= 1 */
/* a full copy of `src` (arrays, quantizer, label sets + start map, visibility masks) on dst_ctx's device, device to device;
 * the copy is an ordinary index (vs_index_free).  Works between two contexts of one device as well. */
int vs_index_replicate(vs_index* src, vs_ctx* dst_ctx, vs_index** out);

typedef struct vs_multi vs_multi;
#define VS_MULTI_COPY_ALWAYS 1u /* also the source's own device gets a replica instead of a view of the source's arrays */
/* one context + one copy of `src` per entry of devices[] (the first entry naming src's own device reads src's arrays through
 * a view; src must outlive the vs_multi and must not be mutated while it exists) */
int vs_multi_create(vs_index* src, const int* devices, uint32_t n_devices, uint32_t flags, vs_multi** out);
uint32_t vs_multi_size(const vs_multi* m);
vs_index* vs_multi_index(vs_multi* m, uint32_t i); /* shard i's index / context: device-resident work is driven per device */
vs_ctx* vs_multi_ctx(vs_multi* m, uint32_t i);
/* vs_search_batch / vs_stream_batch over all devices: same arguments, same rows in the same order as one device returns for the
 * whole batch (scans are independent); stats are summed over the shards */
int vs_multi_search_batch(vs_multi* m, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                          uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids,
                          float* out_dist, vs_stats* stats);
int vs_multi_stream_batch(vs_multi* m, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                          uint32_t search_list_size, uint32_t mrows, uint32_t* out_ids, uint32_t* out_ham, vs_stats* stats);
void vs_multi_destroy(vs_multi* m);

typedef struct vs_comm vs_comm;
#define VS_COMM_ID_BYTES 128
int vs_comm_unique_id(uint8_t* id /* [VS_COMM_ID_BYTES] */);  /* one rank calls it (ncclGetUniqueId), every rank gets the bytes */
int vs_comm_create(vs_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t world, vs_comm** out); /* collective: ncclCommInitRank */
uint32_t vs_comm_rank(const vs_comm* c);
uint32_t vs_comm_world(const vs_comm* c);
/* the final top-k gather: this rank's device-resident [nq_local][k] blocks -> [nq_total][k] on every rank, shards in rank order
 * (nq_local must be this rank's vs_shard_range of nq_total; d_dist / d_out_dist may both be NULL).  Enqueued on ctx's stream
 * behind the search that produced the blocks (vs_search_batch_dev); vs_ctx_sync completes it.  No host synchronisation. */
int vs_comm_gather_topk(vs_comm* c, const uint32_t* d_ids, const float* d_dist, uint32_t nq_local, uint32_t nq_total, uint32_t k,
                        uint32_t* d_out_ids, float* d_out_dist);
int vs_comm_bcast(vs_comm* c, void* d_buf, size_t bytes, uint32_t root); /* one device array from root to every rank (enqueued) */
/* every rank passes an index of the same geometry on the communicator's device (root: the built / uploaded one, the others:
 * vs_index_alloc); on return every rank holds root's index (arrays, quantizer, label sets + masks, start map, visibility) */
int vs_comm_replicate_index(vs_comm* c, vs_index* idx, uint32_t root);
void vs_comm_destroy(vs_comm* c);

/* ---- build-side helpers (SURVEY.md §8f "next" rows; needed to manufacture device-resident indexes) ---------- */
/* Welford pass over rows [0,n) in heap order, bit-exact to SbqQuantizer::add_sample (AM/sbq/quantize.rs:115-148):
 * one lane per dimension, sequential over rows.  Uses the (cosine-normalised) first dim_index dims of the vectors. */
int vs_sbq_train(vs_index* idx);
/* codes[i] = quantize(normalised index slice of vecs[i]) for all nodes */
int vs_sbq_quantize_corpus(vs_index* idx);
/* Batched Vamana build over the SBQ codes (greedy search + robust prune with alpha ladder, Hamming distances),
 * the GPU counterpart of Graph::insert / prune_neighbors (AM/graph/mod.rs:392-488,637-717).  With label sets attached
 * (vs_index_set_labels before the call) the build is label-aware as Graph::insert is: a filtered pass from the label start
 * nodes, an unfiltered pass from the default start node, contains_intersection in the pruning rule, and a node becomes the
 * start node of every label it is the first to carry (at most 64 labels per node). */
int vs_build_graph(vs_index* idx, uint32_t search_list_size, double max_alpha, uint32_t batch_max, uint64_t seed);
/* nodes the last vs_build_graph could not make reachable from the default start node (its repair pass gives every such
 * node an in-edge where that strands nobody else; 0 on well-formed input, 0xFFFFFFFF = graph deeper than the pass can judge).
 * The reference's build gives no reachability guarantee either (AM/graph/mod.rs:700-715 only warns about orphans). */
uint32_t vs_index_build_unreachable(const vs_index* idx);

/* ---- synthetic corpora generated in HBM (bench / tests; bit-reproducible on the CPU, see pgvectorscale_amd/datagen.py) */
typedef struct vs_datagen_params {
    uint64_t seed;
    uint32_t dim;          /* vector dimensionality                                   */
    uint32_t latent_dim;   /* intrinsic dimensionality of the mixture                 */
    uint32_t n_clusters;   /* mixture components                                      */
    uint32_t intra_pct;    /* within-cluster latent spread, percent of between-cluster spread */
    uint32_t noise_pct;    /* isotropic ambient noise, percent of signal scale        */
    uint32_t normalize;    /* 1 => unit L2 norm                                       */
} vs_datagen_params;
/* fills d_out[row_begin .. row_begin+rows) (row stride = dim floats) with rows `first_row + i` of the stream */
int vs_datagen_fill(vs_ctx* ctx, const vs_datagen_params* p, uint64_t first_row, uint64_t rows, float* d_out);
/* exact brute-force f32 top-k on the device (ground truth for recall): d_queries [nq][dim_full] raw */
int vs_bruteforce_topk(vs_index* idx, const float* d_queries, uint32_t nq, uint32_t k, uint32_t* out_ids, float* out_dist);

#ifdef __cplusplus
}
#endif
#endif /* VSGPU_H */
