// vs_internal.h — shared declarations of libvsgpu.so (HIP / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/vsgpu.h"

#define VS_EMPTY 0xFFFFFFFFu

void vs_set_error(const char* fmt, ...);
// vs_options.cpp: the option table (vs_set_option, else a snapshot of the VS_* environment refreshed only when it changes); nullptr = unset
const char* vs_opt_get(const char* name);

#define VS_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            vs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return (_e == hipErrorOutOfMemory) ? VS_ERR_OOM : VS_ERR_HIP;                              \
        }                                                                                              \
    } while (0)

#define VS_REQUIRE(cond, ...)         \
    do {                              \
        if (!(cond)) {                \
            vs_set_error(__VA_ARGS__);\
            return VS_ERR_INVALID;    \
        }                             \
    } while (0)

#define VS_REQUIRE_OOM(cond, ...)     \
    do {                              \
        if (!(cond)) {                \
            vs_set_error(__VA_ARGS__);\
            return VS_ERR_OOM;        \
        }                             \
    } while (0)

#define VS_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r != VS_OK) return _r; \
    } while (0)

// extern "C" entry points never let a C++ exception cross the boundary (vsgpu.h: errors are return codes)
#include <exception>
#include <new>
template <class F>
static inline int vs_guard(const char* what, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        vs_set_error("%s: out of host memory", what);
        return VS_ERR_OOM;
    } catch (const std::exception& e) {
        vs_set_error("%s: %s", what, e.what());
        return VS_ERR_INVALID;
    }
}

__host__ __device__ static inline uint32_t round_up_u32(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }
static inline uint32_t next_pow2_u32(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

struct vs_ctx {
    int device = -1;
    hipStream_t stream = nullptr;       // compute
    hipStream_t copy_stream = nullptr;  // H2D/D2H staging
    void* pinned[2] = {nullptr, nullptr};
    size_t pinned_bytes = 0;
    hipEvent_t pinned_ev[2] = {nullptr, nullptr};
    hipDeviceProp_t prop;
    // optional per-kernel event timing (vs_profile_*)
    bool profiling = false;
    struct ProfSpan { int kind; hipEvent_t a, b; };
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[8] = {0};
    uint64_t prof_launches[8] = {0};
};
enum { PK_PREPARE = 0, PK_SEARCH = 1, PK_RERANK = 2, PK_RESORT = 3, PK_SEARCH_FB = 4, PK_SCAN = 5 };
hipEvent_t prof_begin(vs_ctx* c);
void prof_end(vs_ctx* c, int kind, hipEvent_t a);

// growable device scratch buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool in_slab = false;  // a chunk of the index's WsSlab (never freed on its own)
};

// The hot, randomly accessed per-workgroup regions of the search workspace (dedup tables, heap spill arrays of the persistent grid:
// 0.7 GB at 50M) run fastest packed at the start of ONE large device allocation: allocation history moved k_search_fast by +-10 % at
// identical bytes while the regions lived in allocations of their own size (profiles/r04/s7_diag_state_50m.txt,
// s11_diag_spread_50m.txt).  One grow-only slab per index, allocated by the first search that needs it and shared with every view
// of the index (cursor lanes, second streams, the own-device shard of a vs_multi); chunks are bump-allocated, never returned one
// by one, and the slab goes when the last handle that holds it does.
struct WsSlab {
    std::mutex mu;
    // two regions: [0] the dedup tables, [1] the heap spill arrays — halves of one allocation, or two allocations when the probe found a
    // pair of places that beats every single one
    void* base[2] = {nullptr, nullptr};
    size_t bytes[2] = {0, 0}, used[2] = {0, 0};
    void* owned[2] = {nullptr, nullptr};  // what is hipFree'd with the slab (nothing for the caller's memory)
    int refs = 1;
    int device = 0;
    bool tried = false;  // the allocation failed once: handles fall back to allocations of their own
    bool external = false;  // the caller's memory (vs_index_set_slab): used whatever VS_WS_SLAB_MB / the size rule say
};

struct SearchWorkspace {
    DevBuf q_full, qcodes, qlabels, qlabel_off, hash, heap_g, heap_g4, ghash4, heap_g4b, ghash4b, pool_ctr, fb_flag, phase, timeline, stream_ids, stream_ham, stream_cnt, stats, status,
        rr_dist, out_ids, out_tids, out_dist, resort_heap, raw_q, misc, q_index,
        raw_q2, out_ids2, out_tids2, out_dist2;  // second set of a pipelined host batch (search_host)
    // pending async call (vs_search_batch_dev)
    bool fb_valid = false;  // fb_flag holds the fallback marks of the last chunk
    bool pending = false;
    uint32_t pend_nq = 0;
    uint32_t pend_m = 0;
    uint32_t pend_L = 0;
    void* pend_blob = nullptr;  // PendingBatch of vs_api.hip (plan + capacities + output pointers of the batch in flight)
};

// what the last batches needed (per search_list_size / stream length): sizes the LDS dedup table of the next launch
struct ScanObs {
    bool valid = false;
    uint32_t L = 0, M = 0;
    double ins_mean = 0, ins_max = 0;  // inserted ids per scan
    double ov_frac = 1.0;              // fraction of scans that outgrew the LDS table
};

// the launch variant of k_search_fast an index prefers (vs_index_autotune / vs_index_set_variant): -1 = the library default;
// a VS_F_* environment variable still overrides the field it names
struct TuneVariant {
    int virgin = -1, minw = -1, persist = -1;
    int lds_max_ins = -1;  // 0: the table-less regime even for scans whose dedup table would fit LDS
    int vr = -1;           // 0: the LDS-ring visited list also where the register-resident one is the default (LDS-table regime)
    uint32_t gcap = 0;
    char name[40] = "default";
};
// what the last first-attempt launch of k_search_fast really was (a variant that does not exist for an index / operating point
// silently launches the default's instantiation: the autotuner reads this to tell)
struct FastSig {
    uint32_t vwords = 0, minw = 0, gcap = 0, lh = 0, vr = 0, ran = 0;
    bool operator==(const FastSig& o) const {
        return vwords == o.vwords && minw == o.minw && gcap == o.gcap && lh == o.lh && vr == o.vr && ran == o.ran;
    }
};

struct vs_index {
    vs_ctx* ctx = nullptr;
    bool is_view = false;  // vs_index_view: the device arrays belong to another handle
    vs_index* view_of = nullptr;  // ... that one (never dereferenced: the owner may be gone)
    uint64_t owner_id = 0;        // key of the owner in the registry of live views (unique per vs_index_alloc / replica, never reused)
    WsSlab* slab = nullptr;       // shared by an index and its views (reference counted)
    vs_index_desc d{};
    uint32_t code_stride = 0;  // u64 words per code row (W rounded up to even, zero padded)
    uint32_t nbr_stride = 0;   // u32 per neighbor row (R rounded up to 16)
    uint32_t vec_stride = 0;   // floats per vector row (dim_full rounded up to 4)
    uint64_t* codes = nullptr;
    uint32_t* nbrs = nullptr;
    uint64_t* tids = nullptr;
    uint32_t build_unreachable = 0;    // nodes the last vs_build_graph left unreachable from the start node (0xFFFFFFFF: not judged)
    const uint8_t* visible = nullptr;  // per node, 0 = the heap fetch finds nothing under the scan's snapshot (nullptr: all visible)
    uint8_t* visible_own = nullptr;    // the library's own copy (vs_index_set_visibility)
    uint8_t* snap[VS_MAX_SNAPSHOTS] = {nullptr};  // per-snapshot masks of shared launches (vs_index_snapshot_put); slot 0 unused
    float* vecs = nullptr;
    float* vnorm = nullptr;  // per node: 0 => leave vector alone, else divisor sqrt(norm) (preprocess_cosine)
    float* vnorm_idx = nullptr;  // the same for the index slice (plain storage, cosine, num_dimensions_to_index < num_dimensions)
    float* mean = nullptr;
    float* m2 = nullptr;
    uint64_t count = 0;
    uint32_t* label_off = nullptr;
    int16_t* label_val = nullptr;
    uint64_t* label_mask = nullptr;  // per node: bit label_bit[l] set <=> label l in its set; only when the index uses <= 64 distinct labels
    uint8_t* label_bit = nullptr;    // [65536] label (as u16) -> its bit, 0xFF = the label occurs nowhere in the index
    // [n][nbr_stride] label masks of every node's neighbors in list order (a cache derived from nbrs + label_mask for the
    // label-filtered scans; rebuilt lazily by vs_refresh_neighbor_masks when nbr_mask_valid is false)
    uint64_t* nbr_mask = nullptr;
    bool nbr_mask_valid = false;
    bool nbr_mask_tried = false;     // the array did not fit the device: the scans load the masks per neighbor
    uint64_t n_label_vals = 0;
    int16_t* ls_labels = nullptr;
    uint32_t* ls_nodes = nullptr;
    SearchWorkspace ws;
    ScanObs obs;
    uint32_t last_ins_limit = 0;  // LDS-table admission limit of the last fast launch
    TuneVariant tune;
    FastSig last_fast;
    vs_stats last_stats{};
};

// "done once per device" for function-scope statics (hipFuncSetAttribute is per device; one thread per device may launch the same
// instantiation at once through vs_multi_*): pending() is true where the calling device's bit is not set yet, done() records it.
// Setting an attribute twice is harmless, so two threads of one device racing through the unset state are fine.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool pending(int device) const { return device < 0 || device >= 64 || !((mask.load(std::memory_order_acquire) >> device) & 1ull); }
    void done(int device) {
        if (device >= 0 && device < 64) mask.fetch_or(1ull << device, std::memory_order_acq_rel);
    }
};
int devbuf_reserve(vs_ctx* ctx, DevBuf& b, size_t bytes);
int devbuf_reserve_hot(vs_index* ix, DevBuf& b, size_t bytes, int which);  // from region `which` of the index's slab when it has one (else as devbuf_reserve)
uint64_t vs_new_owner_id();
WsSlab* vs_slab_new(int device);
void vs_slab_release(WsSlab* s);
int vs_index_live_views(vs_index* ix);  // views made of ix that have not been freed yet
// row-wise staging through the pinned ring (device rows may be wider than host rows) / neighbor-list validation
int vs_upload_rows(vs_ctx* c, void* dst, size_t dev_row_bytes, const void* src, size_t host_row_bytes, size_t copy_bytes, size_t rows);
int vs_validate_graph(vs_index* ix);
bool vs_neighbor_masks_wanted(const vs_index* ix);  // policy: > 8M nodes, or VS_F_NBRMASK=1 / 0
int vs_refresh_neighbor_masks(vs_index* ix);  // (re)derives nbr_mask when it is stale (no-op for a view or when it does not fit)
int vs_refresh_label_masks(vs_index* ix);  // (re)derives label_mask / label_bit from the label CSR, or drops them when the index uses more than 64 distinct labels
void devbuf_free(DevBuf& b);
// ---- shared by the host-side translation units (vs_api / vs_slab / vs_batch / vs_tune / vs_cursor .hip) ----------
// an option as a number (vs_options.cpp: vs_set_option, else the snapshot of the VS_* environment), else the default
static inline uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = vs_opt_get(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}
size_t slab_bytes_wanted(const vs_index* ix);                      // vs_slab.hip: 0 = this index gets no slab
void slab_select(vs_index* ix, WsSlab* s, size_t slab_bytes);     // vs_slab.hip: probes the candidates (caller holds s->mu)
hipEvent_t pool_event(vs_ctx* c);                                  // vs_api.hip: an event from the context's pool
int download_async_rows(vs_ctx* c, void* dst, const void* src, size_t bytes);  // vs_api.hip: rows back through the pinned ring
int vs_search_batch_dev_impl(vs_index* ix, const float* d_queries, const int16_t* d_qlabels, const uint32_t* d_qlabel_off, uint32_t nq,
                             uint32_t L, uint32_t rescore, uint32_t k, uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist);  // vs_batch.hip
int vs_search_batch_dev_finish_impl(vs_index* ix, vs_stats* stats);  // vs_batch.hip

// ---- kernel launch wrappers (defined in the .hip files) -------------------------------------------------------
struct SearchLaunch {
    uint32_t nq, L, M;
    uint32_t hl;       // candidate-heap entries resident in LDS (heap positions [0, hl))
    uint32_t hcap;     // total candidate-heap capacity; positions [hl, hcap) live in heap_g
    uint32_t vcap;     // visited-list capacity (LDS)
    uint32_t lh;       // slots of the LDS-resident exact dedup table (power of two, 0 = none)
    uint32_t hashcap;  // slots reserved per scan for the global overflow dedup ladder
    uint32_t g0;       // slots of the ladder's first level (power of two); level j has g0 << j slots
    const uint64_t* qcodes;        // [nq][code_stride]
    const int16_t* qlabels;        // may be null
    const uint32_t* qlabel_off;    // may be null => no label keys
    uint64_t* heap_g;              // [nq][hcap - hl]  (untouched unless a heap outgrows LDS)
    uint32_t* hash;                // [nq][hashcap]    (lazily cleared by the kernel when first needed)
    uint32_t* out_ids;             // [nq][M]
    uint32_t* out_ham;             // [nq][M]
    uint32_t* out_cnt;             // [nq]
    uint32_t* stats;               // [nq][8]
    uint32_t* status;              // [nq]
    uint32_t only_failed = 0;      // 1: run only the scans whose status[q] != 0 (left over by the fast kernel)
    uint32_t* pool_counter = nullptr;  // non-null: heap_g / hash hold pool_slots regions claimed with an atomic counter
    uint32_t pool_slots = 0;
    uint32_t* fb_flag = nullptr;   // [nq] set to 1 for every scan this launch ran in only_failed mode
    const uint8_t* visible = nullptr;  // non-null: rows whose node has visible[node] == 0 are counted and left out of the stream
    // resumable scans (the amgettuple cursor, AM/scan.rs:162-174,370-405): resume[q * resume_stride ..] is scan q's saved state —
    // header RS_* + the LDS image (heap top, LDS dedup table, visited list); its heap spill array / dedup ladder are region q of
    // heap_g / hash and live on between launches.  A launch continues the scan for M more rows (written to out_ids[q][0..M)).
    uint32_t* resume = nullptr;
    uint32_t resume_stride = 0;
    uint32_t* row_stats = nullptr;     // [nq][M][ST_N] the work counters as they stood when each row was emitted (or null)
};
// header of a saved scan (u32 words), followed by the LDS image
enum { RS_INIT = 0, RS_HLEN, RS_HMAX, RS_VLEN, RS_GLEV, RS_NINS_L, RS_NINS_G, RS_NINS_TOP, RS_VISITS, RS_CAND, RS_DQ, RS_READS,
       RS_NEXT, RS_INVIS, RS_STATUS, RS_EMITTED, RS_HDR = 16 };
size_t search_resume_words(const SearchLaunch& s);  // u32 words of one saved scan at these capacities
// fast path (vs_search_fast.hip): all hot state in LDS
struct FastLaunch {
    uint32_t nq, L, M;
    uint32_t hl;       // heap positions resident in LDS, 2^k - 1
    uint32_t hcap;     // total heap capacity; positions [hl, hcap) live in heap_g
    uint32_t gstride;  // u32 per scan in heap_g (even, >= hcap - hl + 2)
    uint32_t lh;       // slots of the LDS dedup table (multiple of 4)
    uint32_t gcap;     // slots of the per-scan global overflow dedup table (a multiple of 256), handles lh .. lh + gcap - 1
    uint32_t glimit = 0;  // ids the global table may hold when a visit starts (gcap x load limit - one wave of inserts); a scan beyond it is handed to the second attempt
    uint32_t sb;       // bits of a slot handle inside a heap entry (lh + gcap <= 1 << sb)
    uint32_t vr;       // visited list: 8 = eight register pairs (512 entries), 0 = LDS ring of vcap entries
    uint32_t vcap;     // visited ring capacity (vr == 0)
    uint32_t minw;     // register cap variant: waves per SIMD to leave room for (1 = unconstrained)
    uint32_t rc = 0;   // entries of the LDS cache of ids known to be in the table (table-less regime; 0 or a power of two)
    uint32_t persist = 0;  // != 0: persistent grid of `persist` workgroups taking scans from scan_counter; regions of heap_g / ghash are per workgroup
    uint32_t vwords = 0;  // != 0: words of the LDS bitmap of written buckets (one bit per four slots of gcap): tables are neither cleared nor read before their first write
    uint32_t vslot = 0;   // with vwords: the bitmap has one bit per SLOT (gcap / 32 words) and the table is probed slot by slot (linear probing)
                          // 2: ... and the table holds 16-BIT entries in buckets of eight (quotienting, see fast_scan VG == 3)
    // vslot == 2: an id is mapped by a bijection on qd bits to x; bucket = x >> qk (gcap / 8 buckets, a power of two), the entry is the
    // remainder x & (2^qk - 1) (qk <= 16); ids whose bucket is full go to an overflow table of ocap 32-bit slots behind the buckets
    // (handles gcap ..); gregion = u32 words per region (gcap / 2 + ocap)
    uint32_t qd = 0, qk = 0, ocap = 0, gregion = 0;
    uint32_t build = 0; // 1: greedy_search_for_build (the visited list is the output; needs vr == 0)
    uint32_t flags = 0; // FAST_* (measurement switches)
    // second attempt of the scans a first launch gave up on (bigger capacities): only scans whose status[q] != 0 run, their
    // heap spill array comes with the dedup table region claimed from the pool (heap_g is [pool_slots][gstride] then), and
    // fb_flag[q] is set for the statistics
    uint32_t only_failed = 0;
    uint32_t* fb_flag = nullptr;
    const uint8_t* visible = nullptr;  // non-null: rows whose node has visible[node] == 0 are counted and left out of the stream
    const uint64_t* qcodes;
    const int16_t* qlabels;
    const uint32_t* qlabel_off;
    uint32_t* heap_g;  // [nq][gstride] heap positions >= hl
    // the global dedup overflow table is claimed on first need from a pool of pool_slots tables
    uint32_t* ghash;   // [pool_slots][gcap] (cleared by the claiming wave)
    uint32_t* pool_counter;
    uint32_t pool_slots;
    uint32_t* out_ids;
    uint32_t* out_ham;
    uint32_t* out_cnt;
    uint32_t* stats;
    uint32_t* status;
    uint64_t* phase = nullptr;  // optional [nq][8] per-phase shader-clock sums (VS_PHASE=1, diagnostics only)
    uint32_t* scan_counter = nullptr;  // persist: next scan to run (zero before the launch)
    uint64_t* timeline = nullptr;  // optional [nq][2] start / end of every scan in 100 MHz ticks (VS_TIMELINE=1, diagnostics only)
    // resumable scans (the scan pools; as SearchLaunch::resume): resume[q * resume_stride ..] is scan q's saved state — header RSF_* + the
    // LDS image; region q of heap_g / ghash lives on between launches; status[q] != 0 on entry marks the scans that run.  A launch
    // continues a scan for M more rows (out_ids[q][0..M)); row_stats: [nq][M][ST_N] the work counters as each row was emitted.
    uint32_t* resume = nullptr;
    uint32_t resume_stride = 0;
    uint32_t* row_stats = nullptr;
};
enum { RSF_INIT = 0, RSF_HLEN, RSF_VHEAD, RSF_VLEN, RSF_NINS_G, RSF_N_OVF, RSF_HMAX, RSF_VISITS, RSF_CAND, RSF_POPS, RSF_INVIS, RSF_STATUS,
       RSF_NEXT, RSF_ENDED, RSF_HDR = 16 };
size_t fast_resume_words(const FastLaunch& s);
enum {
    // (1 was FAST_PLAIN_ROW_LOADS until round 5: code rows through the normal cache policy; the loads are non-temporal at compile time now)
    FAST_FULL_VARIANT = 8,     // run the instantiation that handles label keys and a visibility mask even when the batch has neither
};
size_t fast_lds_bytes(const vs_index* idx, const FastLaunch& s);
int launch_search_fast(vs_index* idx, const FastLaunch& s);
int fast_resident_scans(vs_index* idx, const FastLaunch& s, uint32_t* out);  // size of a persistent grid for this instantiation
enum { ST_VISITS = 0, ST_CAND = 1, ST_DQ = 2, ST_READS = 3, ST_NEXT = 4, ST_GSPILL = 5, ST_INVIS = 6, ST_N = 8 };
enum { OVF_HEAP = 1, OVF_VISITED = 2, OVF_HASH = 4, OVF_POOL = 8,
       OVF_KEY = 16 };  // (fast kernel only) the scan key has more labels than its LDS slot holds: the general kernel runs the scan
size_t search_lds_bytes(const vs_index* idx, const SearchLaunch& s);

int launch_prepare_queries(vs_index* idx, const float* d_raw, uint32_t nq, float* d_q_full, uint64_t* d_qcodes);
int launch_quantize_rows(vs_index* idx, const float* d_rows, uint32_t row_stride, uint32_t nrows, uint64_t* d_codes,
                         uint32_t code_stride);
int launch_hamming_gather(vs_index* idx, const uint64_t* d_qcodes, const uint32_t* d_ids, const uint32_t* d_off,
                          uint32_t nq, uint32_t* d_out);
int launch_rerank(vs_index* idx, const float* d_q_full, const uint32_t* d_ids, const uint32_t* d_off,
                  const uint32_t* d_cnt, uint32_t fixed_m, uint32_t nq, float* d_out, uint32_t row_base = 0);
int launch_search(vs_index* idx, const SearchLaunch& s, bool build_mode = false);
int launch_resort(vs_index* idx, uint32_t nq, uint32_t M, uint32_t rescore, uint32_t k, const uint32_t* d_stream_ids,
                  const uint32_t* d_cnt, const float* d_dist, uint64_t* d_heap_ws, uint32_t* d_out_ids,
                  uint64_t* d_out_tids, float* d_out_dist);
int launch_resort_cursor(vs_index* idx, uint32_t n, bool exhausted, uint32_t rescore, uint32_t k, const uint32_t* d_stream,
                         const float* d_dist, const uint32_t* d_keys, uint64_t* d_heap, uint32_t* d_cur, uint32_t* d_out_ids,
                         uint64_t* d_out_tids, float* d_out_dist);
int launch_resort_cursor_batch(vs_index* idx, uint32_t n, const uint32_t* d_list, uint32_t rescore, uint32_t k, const uint32_t* d_stream,
                               const float* d_dist, const uint32_t* d_keys, uint32_t row_stride, uint64_t* d_heap, uint32_t* d_cur,
                               uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist, uint32_t out_stride);
int launch_pool_append(vs_index* idx, uint32_t nq, const uint32_t* d_cnt, const uint32_t* d_off, const uint32_t* d_stage, uint32_t stage_kind_stride, uint32_t M,
                       uint32_t* d_all, uint32_t all_kind_stride, uint32_t rows_cap);
int launch_row_norms(vs_index* idx);
int launch_slice_norms(vs_index* idx, float* d_out);  // divisor of the first dim_index dims of every heap vector
int launch_prepare_index_slice(vs_index* idx, const float* d_raw, uint32_t nq, float* d_q_index);
int launch_validate_nbrs(vs_index* idx, uint32_t* d_flag);
int launch_scan_topk(vs_index* idx, const uint64_t* d_qcodes, uint32_t nq, uint32_t k, uint32_t* d_out_ids,
                     uint32_t* d_out_ham, const int16_t* d_qlabels = nullptr, const uint32_t* d_qlabel_off = nullptr,
                     bool live_only = false);
