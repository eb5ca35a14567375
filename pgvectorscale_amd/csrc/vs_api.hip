// vs_api.hip — host side of libvsgpu.so: the C ABI declared in include/vsgpu.h.
// Context / staging / index residency / batched search pipeline.  No CPU compute fallback anywhere: every compute
// entry point needs a HIP device and fails with VS_ERR_HIP otherwise.
#include <thread>
#include <cstdarg>
#include <cmath>
#include <algorithm>
#include <cstdlib>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "vs_internal.h"

static thread_local char g_err[1024] = "";

void vs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vs_last_error(void) { return g_err; }
extern "C" const char* vs_version(void) { return "vsgpu 0.1 (gfx950)"; }

int devbuf_reserve(vs_ctx* ctx, DevBuf& b, size_t bytes) {
    (void)ctx;
    if (bytes <= b.bytes) return VS_OK;
    if (b.p && !b.in_slab) VS_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    b.in_slab = false;
    size_t want = bytes + bytes / 8 + 256;
    VS_HIP(hipMalloc(&b.p, want));
    b.bytes = want;
    return VS_OK;
}
void devbuf_free(DevBuf& b) {
    if (b.p && !b.in_slab) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    b.in_slab = false;
}

// ---------------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------------
static const size_t kPinnedBytes = 32u << 20;  // 2 x 32 MiB staging ring

extern "C" int vs_ctx_create_staging(int device, size_t staging_bytes, vs_ctx** out);
extern "C" int vs_ctx_create(int device, vs_ctx** out) { return vs_ctx_create_staging(device, kPinnedBytes, out); }

extern "C" int vs_ctx_create_staging(int device, size_t staging_bytes, vs_ctx** out) {
    VS_REQUIRE(out != nullptr, "vs_ctx_create: out is NULL");
    VS_REQUIRE(staging_bytes >= 4096 && staging_bytes <= ((size_t)1 << 32), "vs_ctx_create_staging: %zu bytes per staging buffer outside [4 KiB, 4 GiB]",
               staging_bytes);
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        vs_set_error("no HIP device available (%s); libvsgpu has no CPU fallback", hipGetErrorString(e));
        return VS_ERR_HIP;
    }
    VS_REQUIRE(device >= 0 && device < ndev, "vs_ctx_create: device %d out of range [0,%d)", device, ndev);
    VS_HIP(hipSetDevice(device));
    vs_ctx* c = new vs_ctx();
    c->device = device;
    VS_HIP(hipGetDeviceProperties(&c->prop, device));
    VS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    VS_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    c->pinned_bytes = staging_bytes;
    for (int i = 0; i < 2; ++i) {
        VS_HIP(hipHostMalloc(&c->pinned[i], c->pinned_bytes, hipHostMallocDefault));
        VS_HIP(hipEventCreateWithFlags(&c->pinned_ev[i], hipEventDisableTiming));
    }
    *out = c;
    return VS_OK;
}

extern "C" void vs_ctx_destroy(vs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_stream);
    for (auto& sp : c->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
        if (c->pinned_ev[i]) (void)hipEventDestroy(c->pinned_ev[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

extern "C" int vs_ctx_sync(vs_ctx* c) {
    VS_REQUIRE(c, "vs_ctx_sync: ctx is NULL");
    VS_HIP(hipStreamSynchronize(c->stream));
    return VS_OK;
}
extern "C" void* vs_ctx_stream(vs_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int vs_ctx_device_name(vs_ctx* c, char* buf, size_t len) {
    VS_REQUIRE(c && buf && len, "vs_ctx_device_name: bad args");
    snprintf(buf, len, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return VS_OK;
}
extern "C" int vs_ctx_mem_info(vs_ctx* c, uint64_t* free_b, uint64_t* total_b) {
    VS_REQUIRE(c, "vs_ctx_mem_info: ctx is NULL");
    size_t f = 0, t = 0;
    VS_HIP(hipMemGetInfo(&f, &t));
    if (free_b) *free_b = f;
    if (total_b) *total_b = t;
    return VS_OK;
}

hipEvent_t pool_event(vs_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
hipEvent_t prof_begin(vs_ctx* c) {
    if (!c->profiling) return nullptr;
    hipEvent_t a = pool_event(c);
    (void)hipEventRecord(a, c->stream);
    return a;
}
void prof_end(vs_ctx* c, int kind, hipEvent_t a) {
    if (!c->profiling || !a) return;
    hipEvent_t b = pool_event(c);
    (void)hipEventRecord(b, c->stream);
    c->spans.push_back({kind, a, b});
}
extern "C" int vs_profile_enable(vs_ctx* c, int on) {
    VS_REQUIRE(c, "vs_profile_enable: ctx is NULL");
    c->profiling = on != 0;
    return VS_OK;
}
extern "C" int vs_profile_read(vs_ctx* c, vs_profile* out, int reset) {
    VS_REQUIRE(c && out, "vs_profile_read: bad args");
    VS_HIP(hipStreamSynchronize(c->stream));
    for (auto& sp : c->spans) {
        float ms = 0.f;
        VS_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        c->prof_ms[sp.kind] += ms;
        c->prof_launches[sp.kind] += 1;
        c->event_pool.push_back(sp.a);
        c->event_pool.push_back(sp.b);
    }
    c->spans.clear();
    for (int i = 0; i < 8; ++i) {
        out->ms[i] = c->prof_ms[i];
        out->launches[i] = c->prof_launches[i];
        if (reset) {
            c->prof_ms[i] = 0;
            c->prof_launches[i] = 0;
        }
    }
    return VS_OK;
}

extern "C" int vs_dev_alloc(vs_ctx* c, size_t bytes, void** out) {
    VS_REQUIRE(c && out, "vs_dev_alloc: bad args");
    VS_HIP(hipSetDevice(c->device));
    VS_HIP(hipMalloc(out, bytes ? bytes : 16));
    return VS_OK;
}
extern "C" int vs_dev_free(vs_ctx* c, void* p) {
    VS_REQUIRE(c, "vs_dev_free: ctx is NULL");
    if (p) VS_HIP(hipFree(p));
    return VS_OK;
}

// Host -> HBM through the pinned ring: memcpy into pinned buffer i while buffer 1-i is in flight (hipMemcpyAsync on
// the copy stream).  The final event is waited on by the compute stream so kernels see the data.
// pageable <-> pinned copies of the staging ring.  One thread moves ~13 GB/s, a quarter of what the PCIe link behind the pinned
// buffer takes (50M x 768: 62 ms of a 273 ms PCIe-inclusive step were this memcpy, profiles/r03/bench_50m.json), so chunks of
// 8 MiB and more are split over a few threads (VS_STAGE_THREADS, default 4; 1 = the plain memcpy).
static unsigned stage_threads() {
    static const unsigned nt_cfg = [] {
        const char* e = vs_opt_get("VS_STAGE_THREADS");
        const unsigned v = e && *e ? (unsigned)strtoul(e, nullptr, 10) : 4u;
        return std::min(std::max(v, 1u), 16u);
    }();
    return nt_cfg;
}
static void stage_copy(void* dst, const void* src, size_t n) {
    const unsigned nt_cfg = stage_threads();
    if (n < (8u << 20) || nt_cfg == 1) {
        memcpy(dst, src, n);
        return;
    }
    const size_t part = (((n + nt_cfg - 1) / nt_cfg) + 4095) & ~(size_t)4095;  // (nt_cfg parts cover n)
    std::thread th[16];
    unsigned started = 0;
    for (unsigned t = 1; t < nt_cfg; ++t) {
        const size_t off = (size_t)t * part;
        if (off >= n) break;
        const size_t len = std::min(part, n - off);
        try {
            th[started] = std::thread([=] { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
            started++;
        } catch (...) {  // no thread to be had: this one does the part itself
            memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len);
        }
    }
    memcpy(dst, src, std::min(part, n));
    for (unsigned t = 0; t < started; ++t) th[t].join();
}

extern "C" int vs_dev_upload(vs_ctx* c, void* dst, const void* src, size_t bytes) {
    VS_REQUIRE(c && (bytes == 0 || (dst && src)), "vs_dev_upload: bad args");
    const char* s = static_cast<const char*>(src);
    char* d = static_cast<char*>(dst);
    int slot = 0;
    size_t off = 0;
    while (off < bytes) {
        size_t n = std::min(c->pinned_bytes, bytes - off);
        VS_HIP(hipEventSynchronize(c->pinned_ev[slot]));  // buffer free again?
        stage_copy(c->pinned[slot], s + off, n);
        VS_HIP(hipMemcpyAsync(d + off, c->pinned[slot], n, hipMemcpyHostToDevice, c->copy_stream));
        VS_HIP(hipEventRecord(c->pinned_ev[slot], c->copy_stream));
        off += n;
        slot ^= 1;
    }
    VS_HIP(hipStreamSynchronize(c->copy_stream));
    return VS_OK;
}

extern "C" int vs_dev_download(vs_ctx* c, void* dst, const void* src, size_t bytes) {
    VS_REQUIRE(c && (bytes == 0 || (dst && src)), "vs_dev_download: bad args");
    VS_HIP(hipStreamSynchronize(c->stream));
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    size_t off = 0;
    int slot = 0;
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    while (off < bytes || pend_n[0] || pend_n[1]) {
        if (pend_n[slot]) {  // drain the older transfer in this slot
            VS_HIP(hipEventSynchronize(c->pinned_ev[slot]));
            stage_copy(d + pend_off[slot], c->pinned[slot], pend_n[slot]);
            pend_n[slot] = 0;
        }
        if (off < bytes) {
            size_t n = std::min(c->pinned_bytes, bytes - off);
            VS_HIP(hipMemcpyAsync(c->pinned[slot], s + off, n, hipMemcpyDeviceToHost, c->copy_stream));
            VS_HIP(hipEventRecord(c->pinned_ev[slot], c->copy_stream));
            pend_off[slot] = off;
            pend_n[slot] = n;
            off += n;
        }
        slot ^= 1;
    }
    return VS_OK;
}

// D2H through the pinned ring WITHOUT waiting for the compute stream: the caller has already synchronised with the kernels that
// produced `src`, and later launches on the compute stream (the next chunk of a pipelined batch) do not touch it
int download_async_rows(vs_ctx* c, void* dst, const void* src, size_t bytes) {
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    size_t off = 0;
    int slot = 0;
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    while (off < bytes || pend_n[0] || pend_n[1]) {
        if (pend_n[slot]) {
            VS_HIP(hipEventSynchronize(c->pinned_ev[slot]));
            stage_copy(d + pend_off[slot], c->pinned[slot], pend_n[slot]);
            pend_n[slot] = 0;
        }
        if (off < bytes) {
            const size_t n = std::min(c->pinned_bytes, bytes - off);
            VS_HIP(hipEventSynchronize(c->pinned_ev[slot]));
            VS_HIP(hipMemcpyAsync(c->pinned[slot], s + off, n, hipMemcpyDeviceToHost, c->copy_stream));
            VS_HIP(hipEventRecord(c->pinned_ev[slot], c->copy_stream));
            pend_off[slot] = off;
            pend_n[slot] = n;
            off += n;
        }
        slot ^= 1;
    }
    return VS_OK;
}

// strided upload: host rows of `row_bytes` into device rows of `dev_row_bytes` (zero padded)
static int upload_rows(vs_ctx* c, void* dst, size_t dev_row_bytes, const void* src, size_t host_row_bytes,
                       size_t copy_bytes, size_t rows) {
    if (rows == 0) return VS_OK;
    if (dev_row_bytes == host_row_bytes && copy_bytes == host_row_bytes)
        return vs_dev_upload(c, dst, src, rows * host_row_bytes);
    const size_t rows_per_chunk = std::max<size_t>(1, c->pinned_bytes / dev_row_bytes);
    int slot = 0;
    for (size_t r0 = 0; r0 < rows; r0 += rows_per_chunk) {
        size_t nr = std::min(rows_per_chunk, rows - r0);
        VS_HIP(hipEventSynchronize(c->pinned_ev[slot]));
        char* p = static_cast<char*>(c->pinned[slot]);
        // (rows of a chunk are independent: a chunk of 8 MiB and more is padded + copied by several threads, like stage_copy)
        auto fill = [=](size_t ra, size_t rb) {
            for (size_t r = ra; r < rb; ++r) {
                char* drow = p + r * dev_row_bytes;
                memcpy(drow, static_cast<const char*>(src) + (r0 + r) * host_row_bytes, copy_bytes);
                if (dev_row_bytes > copy_bytes) memset(drow + copy_bytes, 0, dev_row_bytes - copy_bytes);
            }
        };
        const unsigned nt = nr * dev_row_bytes >= (8u << 20) ? stage_threads() : 1u;
        if (nt <= 1) {
            fill(0, nr);
        } else {
            std::thread th[16];
            unsigned started = 0;
            const size_t per = (nr + nt - 1) / nt;
            for (unsigned t = 1; t < nt && (size_t)t * per < nr; ++t) {
                const size_t ra = (size_t)t * per, rb = std::min(nr, ra + per);
                try {
                    th[started] = std::thread(fill, ra, rb);
                    started++;
                } catch (...) {
                    fill(ra, rb);
                }
            }
            fill(0, std::min(per, nr));
            for (unsigned t = 0; t < started; ++t) th[t].join();
        }
        VS_HIP(hipMemcpyAsync(static_cast<char*>(dst) + r0 * dev_row_bytes, p, nr * dev_row_bytes,
                              hipMemcpyHostToDevice, c->copy_stream));
        VS_HIP(hipEventRecord(c->pinned_ev[slot], c->copy_stream));
        slot ^= 1;
    }
    VS_HIP(hipStreamSynchronize(c->copy_stream));
    return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------------------------
static int check_desc(const vs_index_desc* d) {
    VS_REQUIRE(d, "index desc is NULL");
    VS_REQUIRE(d->dim_full >= 1 && d->dim_index >= 1 && d->dim_index <= d->dim_full, "bad dimensions %u/%u",
               d->dim_index, d->dim_full);
    VS_REQUIRE(d->bits >= 1 && d->bits <= 8, "num_bits_per_dimension %u unsupported", d->bits);
    uint64_t nbits = (uint64_t)d->dim_index * d->bits;
    VS_REQUIRE(d->words == (nbits + 63) / 64, "words=%u does not match ceil(dim_index*bits/64)=%llu", d->words,
               (unsigned long long)((nbits + 63) / 64));
    VS_REQUIRE(nbits <= (1u << 24), "dim_index*bits = %llu: Hamming distances must stay exactly representable in f32",
               (unsigned long long)nbits);
    VS_REQUIRE(d->num_neighbors >= 1 && d->num_neighbors <= 1024, "num_neighbors %u unsupported", d->num_neighbors);
    VS_REQUIRE(d->distance_type <= VS_IP, "unknown distance type %u", d->distance_type);
    VS_REQUIRE(d->default_start == VS_INVALID_NODE || d->default_start < d->n, "default_start out of range");
    return VS_OK;
}

extern "C" void vs_index_free(vs_index* ix);

static int index_alloc_arrays(vs_ctx* c, const vs_index_desc* desc, bool with_vecs, vs_index* ix) {
    const size_t n = std::max<uint32_t>(desc->n, 1);
    VS_HIP(hipMalloc(&ix->codes, n * ix->code_stride * sizeof(uint64_t)));
    VS_HIP(hipMalloc(&ix->nbrs, n * ix->nbr_stride * sizeof(uint32_t)));
    VS_HIP(hipMalloc(&ix->tids, n * sizeof(uint64_t)));
    VS_HIP(hipMalloc(&ix->mean, desc->dim_index * sizeof(float)));
    VS_HIP(hipMalloc(&ix->m2, desc->dim_index * sizeof(float)));
    VS_HIP(hipMemsetAsync(ix->m2, 0, desc->dim_index * sizeof(float), c->stream));
    VS_HIP(hipMemsetAsync(ix->mean, 0, desc->dim_index * sizeof(float), c->stream));
    if (with_vecs) {
        VS_HIP(hipMalloc(&ix->vecs, n * ix->vec_stride * sizeof(float)));
        VS_HIP(hipMalloc(&ix->vnorm, n * sizeof(float)));
        VS_HIP(hipMemsetAsync(ix->vnorm, 0, n * sizeof(float), c->stream));
    }
    return VS_OK;
}

static int index_alloc_common(vs_ctx* c, const vs_index_desc* desc, bool with_vecs, vs_index** out) {
    VS_REQUIRE(c && out, "index alloc: bad args");
    *out = nullptr;
    VS_TRY(check_desc(desc));
    VS_HIP(hipSetDevice(c->device));
    vs_index* ix = new vs_index();
    ix->ctx = c;
    ix->owner_id = vs_new_owner_id();
    ix->slab = vs_slab_new(c->device);
    ix->d = *desc;
    // (VS_WS_SLAB_EARLY=1: the slab is the index's FIRST device allocation instead of being made by the first search that needs it)
    if (env_u32("VS_WS_SLAB_EARLY", 0) && slab_bytes_wanted(ix)) {
        std::lock_guard<std::mutex> lk(ix->slab->mu);
        slab_select(ix, ix->slab, slab_bytes_wanted(ix));  // (no arrays yet: one allocation, no probe)
    }
    ix->code_stride = round_up_u32(desc->words, 2);
    ix->nbr_stride = round_up_u32(desc->num_neighbors, 16);
    ix->vec_stride = round_up_u32(desc->dim_full, 4);
    const int r = index_alloc_arrays(c, desc, with_vecs, ix);
    if (r != VS_OK) {  // a 150 GB vector array that does not fit must not leave the other arrays behind
        vs_index_free(ix);
        return r;
    }
    *out = ix;
    return VS_OK;
}

// views point into their source's arrays: the count of live views is what lets the entry points that free or move those arrays
// (label sets, start map) refuse while a lane / a second stream / a vs_multi shard could still launch on the old pointers
static std::mutex vs_view_mu;
// keyed by the owner's id, not its address: an index allocated at the address of a freed one must not inherit (or lose) its count
static std::unordered_map<uint64_t, int> vs_view_count;  // owner id -> live views (owners without views have no entry)
uint64_t vs_new_owner_id() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}
int vs_index_live_views(vs_index* ix) {
    if (ix->is_view) return 0;  // (the mutators below refuse view handles outright: VS_REQUIRE_OWNER)
    std::lock_guard<std::mutex> lk(vs_view_mu);
    const auto it = vs_view_count.find(ix->owner_id);
    return it == vs_view_count.end() ? 0 : it->second;
}
#define VS_REQUIRE_OWNER(ix, what)                                                                                                     \
    do {                                                                                                                                \
        if ((ix)->is_view) {                                                                                                            \
            vs_set_error("%s: this handle is a view; the arrays belong to the index it was made from", what);                         \
            return VS_ERR_STATE;                                                                                                        \
        }                                                                                                                               \
    } while (0)
#define VS_REQUIRE_NO_VIEWS(ix, what)                                                                                                  \
    do {                                                                                                                                \
        const int _nv = vs_index_live_views(ix);                                                                                        \
        if (_nv > 0) {                                                                                                                  \
            vs_set_error("%s: %d view(s) of this index are alive (cursor lanes, a second stream, a vs_multi shard) and hold its device " \
                         "pointers; free them first",                                                                                  \
                         what, _nv);                                                                                                    \
            return VS_ERR_STATE;                                                                                                        \
        }                                                                                                                               \
    } while (0)

extern "C" void vs_index_free(vs_index* ix) {
    if (!ix) return;
    {
        std::lock_guard<std::mutex> lk(vs_view_mu);
        if (ix->is_view) {
            const auto it = vs_view_count.find(ix->owner_id);  // (gone when the owner was freed first)
            if (it != vs_view_count.end() && --it->second <= 0) vs_view_count.erase(it);
        } else {
            const auto it = vs_view_count.find(ix->owner_id);
            if (it != vs_view_count.end()) {
                fprintf(stderr, "[libvsgpu] vs_index_free: %d view(s) of this index are still alive; they must not be used any more\n", it->second);
                vs_view_count.erase(it);
            }
        }
    }
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    void* ptrs[] = {ix->codes, ix->nbrs, ix->tids, ix->vecs, ix->vnorm, ix->vnorm_idx, ix->mean, ix->m2, ix->visible_own,
                    ix->label_off, ix->label_val, ix->label_mask, ix->label_bit, ix->nbr_mask, ix->ls_labels, ix->ls_nodes};
    if (!ix->is_view) {  // (a view shares the arrays of the index it was made from)
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        for (uint8_t* p : ix->snap)
            if (p) (void)hipFree(p);
    }
    SearchWorkspace& w = ix->ws;
    DevBuf* bufs[] = {&w.q_full, &w.qcodes, &w.qlabels, &w.qlabel_off, &w.hash, &w.heap_g, &w.heap_g4, &w.ghash4, &w.heap_g4b, &w.ghash4b, &w.pool_ctr, &w.fb_flag, &w.phase, &w.timeline, &w.raw_q2, &w.out_ids2, &w.out_tids2, &w.out_dist2, &w.stream_ids,
                      &w.stream_ham, &w.stream_cnt, &w.stats, &w.status, &w.rr_dist, &w.out_ids, &w.out_tids,
                      &w.out_dist, &w.resort_heap, &w.raw_q, &w.misc, &w.q_index};
    for (DevBuf* b : bufs) devbuf_free(*b);
    free(w.pend_blob);
    w.pend_blob = nullptr;
    vs_slab_release(ix->slab);  // (after the stream synchronisation above; the last handle frees the allocation)
    delete ix;
}

static int vs_index_view_impl(vs_index* src, vs_ctx* c, vs_index** out) {
    VS_REQUIRE(src && c && out, "vs_index_view: bad args");
    VS_REQUIRE(c->device == src->ctx->device, "vs_index_view: the context is on device %d, the index on device %d", c->device,
               src->ctx->device);
    if (src->nbr_mask_valid) VS_HIP(hipStreamSynchronize(src->ctx->stream));  // (derived arrays were filled on the source's stream)
    vs_index* v = new vs_index(*src);  // the pointers and the geometry; the workspace below is this handle's own
    v->ctx = c;
    v->is_view = true;
    {
        std::lock_guard<std::mutex> lk(vs_view_mu);
        v->view_of = src->is_view ? src->view_of : src;  // (a view of a view is a view of the owner)
        vs_view_count[v->owner_id]++;                    // (owner_id was copied from the source)
    }
    if (env_u32("VS_WS_SLAB_PRIVATE", 0)) {  // (measurement: a slab of the view's own, allocated by its first search)
        v->slab = vs_slab_new(c->device);
    } else if (v->slab) {  // the view's hot regions come out of the owner's slab
        std::lock_guard<std::mutex> lk(v->slab->mu);
        v->slab->refs++;
    }
    v->visible_own = nullptr;
    v->ws = SearchWorkspace{};
    v->last_stats = vs_stats{};
    *out = v;
    return VS_OK;
}
extern "C" int vs_index_view(vs_index* src, vs_ctx* c, vs_index** out) {
    return vs_guard("vs_index_view", [&] { return vs_index_view_impl(src, c, out); });
}

static int index_alloc_fill(vs_ctx* c, const vs_index_desc* desc, vs_index* ix) {
    // empty graph, live tuples with tid = (node<<16)|1 until told otherwise
    VS_HIP(hipMemsetAsync(ix->nbrs, 0xFF, (size_t)std::max<uint32_t>(desc->n, 1) * ix->nbr_stride * 4, c->stream));
    VS_HIP(hipMemsetAsync(ix->codes, 0, (size_t)std::max<uint32_t>(desc->n, 1) * ix->code_stride * 8, c->stream));
    std::vector<uint64_t> t(desc->n);
    for (uint32_t i = 0; i < desc->n; ++i) t[i] = ((uint64_t)i << 16) | 1u;
    VS_TRY(vs_dev_upload(c, ix->tids, t.data(), t.size() * 8));
    VS_HIP(hipStreamSynchronize(c->stream));
    return VS_OK;
}

extern "C" int vs_index_alloc(vs_ctx* c, const vs_index_desc* desc, int with_vecs, vs_index** out) {
    VS_TRY(index_alloc_common(c, desc, with_vecs != 0, out));
    const int r = index_alloc_fill(c, desc, *out);
    if (r != VS_OK) {
        vs_index_free(*out);
        *out = nullptr;
    }
    return r;
}

extern "C" int vs_index_set_quantizer(vs_index* ix, const float* mean, const float* m2, uint64_t count) {
    VS_REQUIRE(ix && mean, "vs_index_set_quantizer: bad args");
    VS_REQUIRE(ix->d.bits == 1 || m2 != nullptr, "m2 is required when num_bits_per_dimension > 1");
    VS_TRY(vs_dev_upload(ix->ctx, ix->mean, mean, ix->d.dim_index * sizeof(float)));
    if (m2) VS_TRY(vs_dev_upload(ix->ctx, ix->m2, m2, ix->d.dim_index * sizeof(float)));
    ix->count = count;
    return VS_OK;
}

extern "C" int vs_index_get_quantizer(const vs_index* ix, float* mean, float* m2, uint64_t* count) {
    VS_REQUIRE(ix, "vs_index_get_quantizer: index is NULL");
    if (mean) VS_TRY(vs_dev_download(ix->ctx, mean, ix->mean, ix->d.dim_index * sizeof(float)));
    if (m2) VS_TRY(vs_dev_download(ix->ctx, m2, ix->m2, ix->d.dim_index * sizeof(float)));
    if (count) *count = ix->count;
    return VS_OK;
}

static int vs_index_set_start_nodes_impl(vs_index* ix, uint32_t default_start, const int16_t* labels,
                                        const uint32_t* nodes, uint32_t n) {
    VS_REQUIRE(ix, "vs_index_set_start_nodes: index is NULL");
    VS_REQUIRE_OWNER(ix, "vs_index_set_start_nodes");
    VS_REQUIRE_NO_VIEWS(ix, "vs_index_set_start_nodes");
    VS_REQUIRE(default_start == VS_INVALID_NODE || default_start < ix->d.n, "default_start out of range");
    for (uint32_t i = 0; i < n; ++i) {
        VS_REQUIRE(nodes[i] < ix->d.n, "label start node out of range");
        VS_REQUIRE(i == 0 || labels[i - 1] < labels[i], "label start map must be sorted by label, unique");
    }
    ix->d.default_start = default_start;
    if (ix->ls_labels) VS_HIP(hipFree(ix->ls_labels));
    if (ix->ls_nodes) VS_HIP(hipFree(ix->ls_nodes));
    ix->ls_labels = nullptr;
    ix->ls_nodes = nullptr;
    ix->d.n_label_starts = n;
    if (n) {
        VS_HIP(hipMalloc(&ix->ls_labels, n * sizeof(int16_t)));
        VS_HIP(hipMalloc(&ix->ls_nodes, n * sizeof(uint32_t)));
        VS_TRY(vs_dev_upload(ix->ctx, ix->ls_labels, labels, n * sizeof(int16_t)));
        VS_TRY(vs_dev_upload(ix->ctx, ix->ls_nodes, nodes, n * sizeof(uint32_t)));
    }
    return VS_OK;
}
extern "C" int vs_index_set_start_nodes(vs_index* ix, uint32_t default_start, const int16_t* labels,
                                        const uint32_t* nodes, uint32_t n) {
    return vs_guard("vs_index_set_start_nodes", [&] { return vs_index_set_start_nodes_impl(ix, default_start, labels, nodes, n); });
}


static int vs_index_set_labels_impl(vs_index* ix, const uint32_t* label_off, const int16_t* label_val) {
    VS_REQUIRE(ix && label_off, "vs_index_set_labels: bad args");
    VS_REQUIRE_OWNER(ix, "vs_index_set_labels");
    VS_REQUIRE_NO_VIEWS(ix, "vs_index_set_labels");
    const uint32_t n = ix->d.n;
    VS_REQUIRE(label_off[0] == 0, "label_off[0] must be 0");
    for (uint32_t i = 0; i < n; ++i) {
        VS_REQUIRE(label_off[i] <= label_off[i + 1], "label_off must be non-decreasing");
        for (uint32_t j = label_off[i] + 1; j < label_off[i + 1]; ++j)
            VS_REQUIRE(label_val[j - 1] < label_val[j], "node %u: label set must be sorted and de-duplicated", i);
    }
    if (ix->label_off) VS_HIP(hipFree(ix->label_off));
    if (ix->label_val) VS_HIP(hipFree(ix->label_val));
    ix->n_label_vals = label_off[n];
    VS_HIP(hipMalloc(&ix->label_off, ((size_t)n + 1) * 4));
    VS_HIP(hipMalloc(&ix->label_val, std::max<uint64_t>(ix->n_label_vals, 1) * 2));
    VS_TRY(vs_dev_upload(ix->ctx, ix->label_off, label_off, ((size_t)n + 1) * 4));
    if (ix->n_label_vals) VS_TRY(vs_dev_upload(ix->ctx, ix->label_val, label_val, ix->n_label_vals * 2));
    ix->d.has_labels = 1;
    return vs_refresh_label_masks(ix);
}
extern "C" int vs_index_set_labels(vs_index* ix, const uint32_t* label_off, const int16_t* label_val) {
    return vs_guard("vs_index_set_labels", [&] { return vs_index_set_labels_impl(ix, label_off, label_val); });
}


extern "C" uint32_t vs_index_build_unreachable(const vs_index* ix) { return ix ? ix->build_unreachable : 0xFFFFFFFFu; }

extern "C" int vs_index_set_visibility_dev(vs_index* ix, const uint8_t* d_visible) {
    VS_REQUIRE(ix, "vs_index_set_visibility_dev: index is NULL");
    ix->visible = d_visible;
    return VS_OK;
}

extern "C" int vs_index_set_visibility(vs_index* ix, const uint8_t* visible) {
    VS_REQUIRE(ix, "vs_index_set_visibility: index is NULL");
    if (!visible) {
        ix->visible = nullptr;
        return VS_OK;
    }
    if (!ix->visible_own) VS_HIP(hipMalloc(&ix->visible_own, std::max<size_t>(ix->d.n, 1)));
    if (ix->d.n) VS_TRY(vs_dev_upload(ix->ctx, ix->visible_own, visible, ix->d.n));
    ix->visible = ix->visible_own;
    return VS_OK;
}

extern "C" int vs_index_snapshot_put(vs_index* ix, uint32_t snapshot, const uint8_t* visible) {
    VS_REQUIRE(ix && snapshot >= 1 && snapshot < VS_MAX_SNAPSHOTS, "vs_index_snapshot_put: snapshot id outside [1,%d]", VS_MAX_SNAPSHOTS - 1);
    VS_HIP(hipSetDevice(ix->ctx->device));
    if (!visible) {
        if (ix->snap[snapshot]) {
            VS_HIP(hipStreamSynchronize(ix->ctx->stream));  // no launch may still read it
            if (ix->visible == ix->snap[snapshot]) ix->visible = nullptr;
            VS_HIP(hipFree(ix->snap[snapshot]));
            ix->snap[snapshot] = nullptr;
        }
        return VS_OK;
    }
    if (!ix->snap[snapshot]) VS_HIP(hipMalloc(&ix->snap[snapshot], std::max<size_t>(ix->d.n, 1)));
    else VS_HIP(hipStreamSynchronize(ix->ctx->stream));  // (replacing a mask a launch may still be reading)
    if (ix->d.n) VS_TRY(vs_dev_upload(ix->ctx, ix->snap[snapshot], visible, ix->d.n));
    return VS_OK;
}

extern "C" int vs_index_snapshot_use(vs_index* ix, uint32_t snapshot, const uint8_t** previous) {
    VS_REQUIRE(ix && snapshot < VS_MAX_SNAPSHOTS, "vs_index_snapshot_use: snapshot id outside [0,%d]", VS_MAX_SNAPSHOTS - 1);
    if (snapshot && !ix->snap[snapshot]) {
        vs_set_error("snapshot %u has no visibility mask (vs_index_snapshot_put)", snapshot);
        return VS_ERR_STATE;
    }
    if (previous) *previous = ix->visible;
    ix->visible = snapshot ? ix->snap[snapshot] : nullptr;
    return VS_OK;
}

extern "C" int vs_index_snapshot_share(vs_index* view, const vs_index* src) {
    VS_REQUIRE(view && src && view->is_view, "vs_index_snapshot_share: needs a view and its source");
    for (int i = 0; i < VS_MAX_SNAPSHOTS; ++i) view->snap[i] = src->snap[i];
    return VS_OK;
}
extern "C" int vs_index_device(const vs_index* ix) { return ix ? ix->ctx->device : -1; }

extern "C" int vs_index_refresh_norms(vs_index* ix) {
    VS_REQUIRE(ix, "vs_index_refresh_norms: index is NULL");
    VS_TRY(launch_row_norms(ix));
    VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    return VS_OK;
}

static int validate_graph(vs_index* ix) {
    uint32_t* d_flag = nullptr;
    VS_HIP(hipMalloc(&d_flag, 4));
    int r = VS_OK;
    if (hipMemsetAsync(d_flag, 0, 4, ix->ctx->stream) != hipSuccess) {
        vs_set_error("validate_graph: hipMemsetAsync failed");
        r = VS_ERR_HIP;
    }
    if (r == VS_OK) r = launch_validate_nbrs(ix, d_flag);
    uint32_t flag = 0;
    if (r == VS_OK) {
        hipError_t e = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ix->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->ctx->stream);
        if (e != hipSuccess) {
            vs_set_error("validate_graph: %s", hipGetErrorString(e));
            r = VS_ERR_HIP;
        }
    }
    (void)hipFree(d_flag);
    VS_TRY(r);
    VS_REQUIRE(!(flag & 2u), "neighbor list refers to a node id >= n");
    VS_REQUIRE(!(flag & 1u), "a neighbor list contains the same node twice");
    return VS_OK;
}

// (used by vs_pages_dev.hip)
int vs_upload_rows(vs_ctx* c, void* dst, size_t dev_row_bytes, const void* src, size_t host_row_bytes, size_t copy_bytes, size_t rows) {
    return upload_rows(c, dst, dev_row_bytes, src, host_row_bytes, copy_bytes, rows);
}
int vs_validate_graph(vs_index* ix) { return validate_graph(ix); }

static int vs_index_upload_impl(vs_ctx* c, const vs_index_desc* desc, const vs_index_host* h, vs_index** out) {
    VS_REQUIRE(h && out && desc, "vs_index_upload: bad args");
    const bool plain = desc->storage_type == VS_STORAGE_PLAIN;
    VS_REQUIRE(desc->storage_type == VS_STORAGE_SBQ || plain, "vs_index_upload: unknown storage_type %u", desc->storage_type);
    if (plain) {
        // PlainNode = vector + neighbor pointers + heap pointer (AM/plain/node.rs); no quantizer, no labels
        VS_REQUIRE(h->nbrs && h->heap_tids && h->vecs, "vs_index_upload: plain storage needs nbrs / heap_tids / vecs");
        VS_REQUIRE(desc->dim_index <= desc->dim_full, "num_dimensions_to_index > num_dimensions");
        VS_REQUIRE(!desc->has_labels, "Plain storage does not support label filters");
    } else {
        VS_REQUIRE(h->codes && h->nbrs && h->heap_tids && h->mean, "vs_index_upload: codes/nbrs/heap_tids/mean required");
    }
    VS_REQUIRE(h->nbr_stride >= desc->num_neighbors, "nbr_stride < num_neighbors");
    VS_REQUIRE(!desc->has_labels || (h->label_off && h->label_val), "has_labels set but no label arrays");
    vs_index* ix = nullptr;
    VS_TRY(index_alloc_common(c, desc, h->vecs != nullptr, &ix));
    int r = VS_OK;
    const size_t n = desc->n;
    do {
        if (h->codes) {
            if ((r = upload_rows(c, ix->codes, ix->code_stride * 8ull, h->codes, desc->words * 8ull, desc->words * 8ull, n))) break;
        } else if (hipMemsetAsync(ix->codes, 0, (size_t)n * ix->code_stride * 8, c->stream) != hipSuccess) {
            r = VS_ERR_HIP;
            break;
        }
        // neighbor rows: copy R ids, pad the device row with the end-of-list sentinel
        {
            std::vector<uint32_t> row_buf;
            const size_t rows_per_chunk = std::max<size_t>(1, (8u << 20) / (ix->nbr_stride * 4));
            row_buf.resize(rows_per_chunk * ix->nbr_stride);
            for (size_t r0 = 0; r0 < n && r == VS_OK; r0 += rows_per_chunk) {
                size_t nr = std::min(rows_per_chunk, n - r0);
                std::fill(row_buf.begin(), row_buf.begin() + nr * ix->nbr_stride, VS_INVALID_NODE);
                for (size_t i = 0; i < nr; ++i)
                    memcpy(&row_buf[i * ix->nbr_stride], h->nbrs + (r0 + i) * h->nbr_stride, desc->num_neighbors * 4ull);
                r = vs_dev_upload(c, ix->nbrs + r0 * ix->nbr_stride, row_buf.data(), nr * ix->nbr_stride * 4ull);
            }
            if (r) break;
        }
        if ((r = vs_dev_upload(c, ix->tids, h->heap_tids, n * 8))) break;
        if (h->vecs)
            if ((r = upload_rows(c, ix->vecs, ix->vec_stride * 4ull, h->vecs, desc->dim_full * 4ull, desc->dim_full * 4ull, n))) break;
        if (h->mean)
            if ((r = vs_index_set_quantizer(ix, h->mean, h->m2, h->count))) break;
        if (desc->has_labels)
            if ((r = vs_index_set_labels(ix, h->label_off, h->label_val))) break;
        if ((r = vs_index_set_start_nodes(ix, desc->default_start, h->label_start_labels, h->label_start_nodes,
                                          desc->n_label_starts))) break;
        if ((r = validate_graph(ix))) break;
        if ((r = vs_index_refresh_norms(ix))) break;
        if (plain && desc->dim_index < desc->dim_full && desc->distance_type == VS_COSINE) {  // norms of the stored index slices
            if (hipMalloc(&ix->vnorm_idx, (size_t)std::max<uint32_t>(desc->n, 1) * 4) != hipSuccess) {
                vs_set_error("vs_index_upload: out of device memory");
                r = VS_ERR_OOM;
                break;
            }
            if ((r = launch_slice_norms(ix, ix->vnorm_idx))) break;
            if (hipStreamSynchronize(c->stream) != hipSuccess) {
                r = VS_ERR_HIP;
                break;
            }
        }
    } while (0);
    if (r != VS_OK) {
        vs_index_free(ix);
        return r;
    }
    *out = ix;
    return VS_OK;
}
extern "C" int vs_index_upload(vs_ctx* c, const vs_index_desc* desc, const vs_index_host* h, vs_index** out) {
    return vs_guard("vs_index_upload", [&] { return vs_index_upload_impl(c, desc, h, out); });
}


extern "C" int vs_index_get_desc(const vs_index* ix, vs_index_desc* out) {
    VS_REQUIRE(ix && out, "vs_index_get_desc: bad args");
    *out = ix->d;
    return VS_OK;
}

extern "C" int vs_index_array(const vs_index* ix, int which, void** p, uint32_t* stride) {
    VS_REQUIRE(ix && p, "vs_index_array: bad args");
    uint32_t s = 1;
    switch (which) {
        case VS_ARR_CODES: *p = ix->codes; s = ix->code_stride; break;
        case VS_ARR_NBRS:  // (the caller may write through this pointer: whatever was derived from the neighbor lists is stale;
            // a caller that keeps the pointer and writes again later must ask for it again before the next scan)
            const_cast<vs_index*>(ix)->nbr_mask_valid = false;
            *p = ix->nbrs;
            s = ix->nbr_stride;
            break;
        case VS_ARR_TIDS: *p = ix->tids; break;
        case VS_ARR_VECS: *p = ix->vecs; s = ix->vec_stride; break;
        case VS_ARR_MEAN: *p = ix->mean; break;
        case VS_ARR_M2: *p = ix->m2; break;
        case VS_ARR_VNORM: *p = ix->vnorm; break;
        case VS_ARR_LABEL_OFF: *p = ix->label_off; break;
        case VS_ARR_LABEL_VAL: *p = ix->label_val; break;
        default: vs_set_error("vs_index_array: unknown array %d", which); return VS_ERR_INVALID;
    }
    if (stride) *stride = s;
    return VS_OK;
}

extern "C" int vs_index_download(const vs_index* ix, uint64_t* codes, uint32_t* nbrs, uint64_t* heap_tids, float* vecs,
                                 uint32_t row_begin, uint32_t row_count) {
    VS_REQUIRE(ix, "vs_index_download: index is NULL");
    VS_REQUIRE((uint64_t)row_begin + row_count <= ix->d.n, "vs_index_download: row range out of bounds");
    vs_ctx* c = ix->ctx;
    VS_HIP(hipStreamSynchronize(c->stream));
    const size_t nr = row_count;
    if (codes)
        VS_HIP(hipMemcpy2D(codes, ix->d.words * 8ull, ix->codes + (size_t)row_begin * ix->code_stride,
                           ix->code_stride * 8ull, ix->d.words * 8ull, nr, hipMemcpyDeviceToHost));
    if (nbrs)
        VS_HIP(hipMemcpy2D(nbrs, ix->d.num_neighbors * 4ull, ix->nbrs + (size_t)row_begin * ix->nbr_stride,
                           ix->nbr_stride * 4ull, ix->d.num_neighbors * 4ull, nr, hipMemcpyDeviceToHost));
    if (heap_tids) VS_HIP(hipMemcpy(heap_tids, ix->tids + row_begin, nr * 8, hipMemcpyDeviceToHost));
    if (vecs) {
        VS_REQUIRE(ix->vecs, "index has no vector column");
        VS_HIP(hipMemcpy2D(vecs, ix->d.dim_full * 4ull, ix->vecs + (size_t)row_begin * ix->vec_stride,
                           ix->vec_stride * 4ull, ix->d.dim_full * 4ull, nr, hipMemcpyDeviceToHost));
    }
    return VS_OK;
}

extern "C" int vs_index_mark_deleted(vs_index* ix, const uint32_t* nodes, uint32_t n) {
    VS_REQUIRE(ix && (n == 0 || nodes), "vs_index_mark_deleted: bad args");
    VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    for (uint32_t i = 0; i < n; ++i) {
        VS_REQUIRE(nodes[i] < ix->d.n, "node id out of range");
        uint64_t t = 0;
        VS_HIP(hipMemcpy(&t, ix->tids + nodes[i], 8, hipMemcpyDeviceToHost));
        t &= ~0xFFFFull;  // heap_item_pointer.offset = InvalidOffsetNumber
        VS_HIP(hipMemcpy(ix->tids + nodes[i], &t, 8, hipMemcpyHostToDevice));
    }
    return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// single-kernel entry points (host pointers in/out)
// ---------------------------------------------------------------------------------------------------------------
static int vs_quantize_impl(vs_index* ix, const float* q, uint32_t nq, uint64_t* out_codes) {
    VS_REQUIRE(ix && (nq == 0 || (q && out_codes)), "vs_quantize: bad args");
    if (nq == 0) return VS_OK;
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    const uint32_t di = ix->d.dim_index;
    VS_TRY(devbuf_reserve(c, w.raw_q, (size_t)nq * di * 4));
    VS_TRY(devbuf_reserve(c, w.qcodes, (size_t)nq * ix->code_stride * 8));
    VS_TRY(vs_dev_upload(c, w.raw_q.p, q, (size_t)nq * di * 4));
    VS_TRY(launch_quantize_rows(ix, (const float*)w.raw_q.p, di, nq, (uint64_t*)w.qcodes.p, ix->code_stride));
    VS_HIP(hipStreamSynchronize(c->stream));
    VS_HIP(hipMemcpy2D(out_codes, ix->d.words * 8ull, w.qcodes.p, ix->code_stride * 8ull, ix->d.words * 8ull, nq,
                       hipMemcpyDeviceToHost));
    return VS_OK;
}
extern "C" int vs_quantize(vs_index* ix, const float* q, uint32_t nq, uint64_t* out_codes) {
    return vs_guard("vs_quantize", [&] { return vs_quantize_impl(ix, q, nq, out_codes); });
}


static int upload_qcodes(vs_index* ix, const uint64_t* qcodes, uint32_t nq) {
    SearchWorkspace& w = ix->ws;
    VS_TRY(devbuf_reserve(ix->ctx, w.qcodes, (size_t)nq * ix->code_stride * 8));
    return upload_rows(ix->ctx, w.qcodes.p, ix->code_stride * 8ull, qcodes, ix->d.words * 8ull, ix->d.words * 8ull, nq);
}

static int vs_hamming_gather_impl(vs_index* ix, const uint64_t* qcodes, const uint32_t* ids, const uint32_t* off,
                                 uint32_t nq, uint32_t* out) {
    VS_REQUIRE(ix && (nq == 0 || (qcodes && off)), "vs_hamming_gather: bad args");
    if (nq == 0) return VS_OK;
    VS_REQUIRE(off[0] == 0, "off[0] must be 0");
    const uint32_t total = off[nq];
    for (uint32_t i = 0; i < nq; ++i) VS_REQUIRE(off[i] <= off[i + 1], "off must be non-decreasing");
    for (uint32_t i = 0; i < total; ++i) VS_REQUIRE(ids[i] < ix->d.n, "node id %u out of range", ids[i]);
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    VS_TRY(upload_qcodes(ix, qcodes, nq));
    VS_TRY(devbuf_reserve(c, w.stream_ids, std::max<size_t>(total, 1) * 4));
    VS_TRY(devbuf_reserve(c, w.stream_ham, std::max<size_t>(total, 1) * 4));
    VS_TRY(devbuf_reserve(c, w.qlabel_off, ((size_t)nq + 1) * 4));
    VS_TRY(vs_dev_upload(c, w.stream_ids.p, ids, (size_t)total * 4));
    VS_TRY(vs_dev_upload(c, w.qlabel_off.p, off, ((size_t)nq + 1) * 4));
    VS_TRY(launch_hamming_gather(ix, (const uint64_t*)w.qcodes.p, (const uint32_t*)w.stream_ids.p,
                                 (const uint32_t*)w.qlabel_off.p, nq, (uint32_t*)w.stream_ham.p));
    VS_TRY(vs_dev_download(c, out, w.stream_ham.p, (size_t)total * 4));
    return VS_OK;
}
extern "C" int vs_hamming_gather(vs_index* ix, const uint64_t* qcodes, const uint32_t* ids, const uint32_t* off,
                                 uint32_t nq, uint32_t* out) {
    return vs_guard("vs_hamming_gather", [&] { return vs_hamming_gather_impl(ix, qcodes, ids, off, nq, out); });
}


static int vs_rerank_impl(vs_index* ix, const float* q_full, const uint32_t* ids, const uint32_t* off, uint32_t nq,
                         float* out) {
    VS_REQUIRE(ix && (nq == 0 || (q_full && off)), "vs_rerank: bad args");
    if (nq == 0) return VS_OK;
    VS_REQUIRE(ix->vecs, "index has no vector column: rerank impossible");
    VS_REQUIRE(off[0] == 0, "off[0] must be 0");
    const uint32_t total = off[nq];
    for (uint32_t i = 0; i < nq; ++i) VS_REQUIRE(off[i] <= off[i + 1], "off must be non-decreasing");
    for (uint32_t i = 0; i < total; ++i) VS_REQUIRE(ids[i] < ix->d.n, "node id %u out of range", ids[i]);
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    VS_TRY(devbuf_reserve(c, w.raw_q, (size_t)nq * ix->d.dim_full * 4));
    VS_TRY(devbuf_reserve(c, w.q_full, (size_t)nq * ix->vec_stride * 4));
    VS_TRY(devbuf_reserve(c, w.qcodes, (size_t)nq * ix->code_stride * 8));
    VS_TRY(devbuf_reserve(c, w.stream_ids, std::max<size_t>(total, 1) * 4));
    VS_TRY(devbuf_reserve(c, w.rr_dist, std::max<size_t>(total, 1) * 4));
    VS_TRY(devbuf_reserve(c, w.qlabel_off, ((size_t)nq + 1) * 4));
    VS_TRY(vs_dev_upload(c, w.raw_q.p, q_full, (size_t)nq * ix->d.dim_full * 4));
    VS_TRY(vs_dev_upload(c, w.stream_ids.p, ids, (size_t)total * 4));
    VS_TRY(vs_dev_upload(c, w.qlabel_off.p, off, ((size_t)nq + 1) * 4));
    VS_TRY(launch_prepare_queries(ix, (const float*)w.raw_q.p, nq, (float*)w.q_full.p, (uint64_t*)w.qcodes.p));
    VS_TRY(launch_rerank(ix, (const float*)w.q_full.p, (const uint32_t*)w.stream_ids.p, (const uint32_t*)w.qlabel_off.p,
                         nullptr, 0, nq, (float*)w.rr_dist.p));
    VS_TRY(vs_dev_download(c, out, w.rr_dist.p, (size_t)total * 4));
    return VS_OK;
}
extern "C" int vs_rerank(vs_index* ix, const float* q_full, const uint32_t* ids, const uint32_t* off, uint32_t nq,
                         float* out) {
    return vs_guard("vs_rerank", [&] { return vs_rerank_impl(ix, q_full, ids, off, nq, out); });
}
