// vs_multi.hip — the multi-GPU half of the path behind the C ABI (the reference has no counterpart:
// AM/mod.rs:63 amcanparallel = false, one backend = one scan).
//
// The path shards by QUERY: scans are independent and read-only, every device holds the whole index, device g takes a
// contiguous block of the batch, and one gather of the [nq, k] id / distance blocks closes the step.  Nothing here is a data-path
// collective.  Two deployments, both without torch:
//
//   * vs_multi_*  — ONE process that owns N devices (what a PGRX background worker / broker process would be): the index is
//     replicated to the other devices with hipMemcpyPeerAsync over xGMI (vs_index_replicate: no N builds, no N uploads), a host
//     batch is cut into N contiguous shards, one host thread per device runs vs_search_batch on its shard and writes its rows
//     straight into the caller's buffers at the shard's offset — the "gather" of a host batch is that placement.
//   * vs_comm_*   — one PROCESS per device (torchrun-style ranks, or one broker per GPU): RCCL over xGMI, loaded at run time
//     (dlopen; a process that never creates a vs_comm never maps librccl).  ncclAllGather of the id and distance blocks closes a
//     step (grouped broadcasts when the shards are uneven), ncclBroadcast replicates an index (or single arrays of it) from the
//     rank that built or uploaded it.  The 128-byte communicator id travels between the processes by the host's own means
//     (PostgreSQL: shared memory; bench.py: its launcher's store).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <system_error>
#include <thread>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>

#include "vs_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// shard arithmetic (the same as pgvectorscale_amd/sharding.py::shard_range): blocks differ by at most one query
// ---------------------------------------------------------------------------------------------------------------
static void shard_of(uint32_t nq_total, uint32_t world, uint32_t rank, uint32_t* begin, uint32_t* end) {
    const uint32_t base = nq_total / world, rem = nq_total % world;
    const uint32_t b = rank * base + std::min(rank, rem);
    *begin = b;
    *end = b + base + (rank < rem ? 1u : 0u);
}
extern "C" int vs_shard_range(uint32_t nq_total, uint32_t world, uint32_t rank, uint32_t* begin, uint32_t* end) {
    VS_REQUIRE(world >= 1 && rank < world && begin && end, "vs_shard_range: bad args (world %u, rank %u)", world, rank);
    shard_of(nq_total, world, rank, begin, end);
    return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// vs_index_replicate: a full copy of an index on another context's device, device to device
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct PeerCopy {
    vs_ctx* dst;
    int src_dev;
    // (a copy between two allocations of ONE device is an ordinary device-to-device copy: the test tier runs two contexts on one GPU)
    int operator()(void* d, const void* s, size_t bytes) const {
        if (!bytes) return VS_OK;
        if (src_dev == dst->device) VS_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst->stream));
        else VS_HIP(hipMemcpyPeerAsync(d, dst->device, s, src_dev, bytes, dst->stream));
        return VS_OK;
    }
};
template <class T>
int clone_array(const PeerCopy& cp, T*& dst, const T* src, size_t count) {
    dst = nullptr;
    if (!src) return VS_OK;
    VS_HIP(hipMalloc(&dst, std::max<size_t>(count, 1) * sizeof(T)));
    return cp(dst, src, count * sizeof(T));
}
}  // namespace

static int vs_index_replicate_impl(vs_index* src, vs_ctx* c, vs_index** out) {
    VS_REQUIRE(src && c && out, "vs_index_replicate: bad args");
    *out = nullptr;
    const int sdev = src->ctx->device;
    // everything the source's streams still have in flight (upload, build, derived arrays) is in its arrays before they are read
    VS_HIP(hipSetDevice(sdev));
    VS_HIP(hipStreamSynchronize(src->ctx->stream));
    VS_HIP(hipStreamSynchronize(src->ctx->copy_stream));
    VS_HIP(hipSetDevice(c->device));
    if (sdev != c->device) {
        int can = 0;
        VS_HIP(hipDeviceCanAccessPeer(&can, c->device, sdev));
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(sdev, 0);  // direct xGMI copies; without it the runtime stages through the host
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) VS_HIP(e);
            (void)hipGetLastError();
        }
    }
    vs_index* ix = new vs_index();
    ix->ctx = c;
    ix->owner_id = vs_new_owner_id();
    ix->slab = vs_slab_new(c->device);
    ix->d = src->d;
    ix->code_stride = src->code_stride;
    ix->nbr_stride = src->nbr_stride;
    ix->vec_stride = src->vec_stride;
    ix->count = src->count;
    ix->n_label_vals = src->n_label_vals;
    ix->build_unreachable = src->build_unreachable;
    ix->tune = src->tune;
    ix->obs = src->obs;
    const PeerCopy cp{c, sdev};
    const size_t n = std::max<uint32_t>(src->d.n, 1);
    auto all = [&]() -> int {
        VS_TRY(clone_array(cp, ix->codes, src->codes, n * src->code_stride));
        VS_TRY(clone_array(cp, ix->nbrs, src->nbrs, n * src->nbr_stride));
        VS_TRY(clone_array(cp, ix->tids, src->tids, n));
        VS_TRY(clone_array(cp, ix->mean, src->mean, src->d.dim_index));
        VS_TRY(clone_array(cp, ix->m2, src->m2, src->d.dim_index));
        VS_TRY(clone_array(cp, ix->vecs, src->vecs, n * src->vec_stride));
        VS_TRY(clone_array(cp, ix->vnorm, src->vnorm, n));
        VS_TRY(clone_array(cp, ix->vnorm_idx, src->vnorm_idx, n));
        VS_TRY(clone_array(cp, ix->label_off, src->label_off, n + 1));
        VS_TRY(clone_array(cp, ix->label_val, src->label_val, (size_t)src->n_label_vals));
        VS_TRY(clone_array(cp, ix->label_mask, src->label_mask, n));
        VS_TRY(clone_array(cp, ix->label_bit, src->label_bit, 65536));
        VS_TRY(clone_array(cp, ix->ls_labels, src->ls_labels, src->d.n_label_starts));
        VS_TRY(clone_array(cp, ix->ls_nodes, src->ls_nodes, src->d.n_label_starts));
        // the visibility mask in force (the scan's snapshot, AM/scan.rs:268-272) and the per-snapshot masks of shared launches
        if (src->visible) {
            VS_TRY(clone_array(cp, ix->visible_own, src->visible, n));
            ix->visible = ix->visible_own;
        }
        for (uint32_t s = 0; s < VS_MAX_SNAPSHOTS; ++s) VS_TRY(clone_array(cp, ix->snap[s], (const uint8_t*)src->snap[s], n));
        // nbr_mask is a cache derived from nbrs + label_mask: the replica rebuilds it on its own device when a filtered scan wants it
        ix->nbr_mask = nullptr;
        ix->nbr_mask_valid = false;
        ix->nbr_mask_tried = false;
        VS_HIP(hipStreamSynchronize(c->stream));
        return VS_OK;
    };
    const int r = all();
    if (r != VS_OK) {
        vs_index_free(ix);
        return r;
    }
    *out = ix;
    return VS_OK;
}
extern "C" int vs_index_replicate(vs_index* src, vs_ctx* c, vs_index** out) {
    return vs_guard("vs_index_replicate", [&] { return vs_index_replicate_impl(src, c, out); });
}

// ---------------------------------------------------------------------------------------------------------------
// vs_multi: one process, N devices
// ---------------------------------------------------------------------------------------------------------------
// One PERSISTENT worker thread per device after the first (round 6; a thread was spawned per device per batch until round 5): the
// thread that runs a device's shard is the same from batch to batch — HIP keeps per-thread state (current device, the error slot),
// thread creation costs tens of microseconds a batch, and a deployment wants to pin it.  A worker sleeps on its condition variable
// between batches; a worker that could not be started leaves its shard to the caller's thread.
struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> task;  // non-empty: posted, not yet finished
    bool busy = false, stop = false, started = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || (busy && task); });
            if (stop) return;
            std::function<void()> t = std::move(task);
            task = nullptr;
            lk.unlock();
            t();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    bool post(std::function<void()> t) {
        if (!started) return false;
        std::lock_guard<std::mutex> g(mu);
        task = std::move(t);
        busy = true;
        cv.notify_all();
        return true;
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy; });
    }
    void shutdown() {
        if (!started) return;
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
        started = false;
    }
};

struct vs_multi {
    struct Dev {
        vs_ctx* ctx = nullptr;
        vs_index* ix = nullptr;
        bool own_ix = true;
        std::unique_ptr<MultiWorker> worker;  // (none for the first device: its shard runs on the caller's thread)
    };
    std::vector<Dev> devs;
    std::mutex batch_mu;  // one batch at a time per vs_multi (the workers hold one task each)
};

extern "C" void vs_multi_destroy(vs_multi* m) {
    if (!m) return;
    for (auto& d : m->devs)
        if (d.worker) d.worker->shutdown();
    for (auto& d : m->devs) {
        if (d.ix && d.own_ix) vs_index_free(d.ix);
        if (d.ctx) vs_ctx_destroy(d.ctx);
    }
    delete m;
}

static int vs_multi_create_impl(vs_index* src, const int* devices, uint32_t n, uint32_t flags, vs_multi** out) {
    VS_REQUIRE(src && devices && n >= 1 && n <= 64 && out, "vs_multi_create: bad args");
    VS_REQUIRE((flags & ~(uint32_t)VS_MULTI_COPY_ALWAYS) == 0, "vs_multi_create: unknown flags 0x%x", flags);
    *out = nullptr;
    vs_multi* m = new vs_multi();
    m->devs.resize(n);
    auto fail = [&](int r) {
        vs_multi_destroy(m);
        return r;
    };
    bool view_made = false;
    for (uint32_t i = 0; i < n; ++i) {
        vs_multi::Dev& d = m->devs[i];
        int r = vs_ctx_create(devices[i], &d.ctx);
        if (r != VS_OK) return fail(r);
        // the source's own device: the first shard there reads the source's arrays through a view (own stream, own workspace);
        // every other entry — another device, or the same device again — gets a replica
        if (devices[i] == src->ctx->device && !view_made && !(flags & VS_MULTI_COPY_ALWAYS)) {
            r = vs_index_view(src, d.ctx, &d.ix);
            view_made = true;
        } else {
            r = vs_index_replicate(src, d.ctx, &d.ix);
        }
        if (r != VS_OK) return fail(r);
    }
    for (uint32_t i = 1; i < n; ++i) {
        vs_multi::Dev& d = m->devs[i];
        d.worker.reset(new MultiWorker());
        try {
            d.worker->th = std::thread([w = d.worker.get()] { w->loop(); });
            d.worker->started = true;
        } catch (const std::system_error&) {
            d.worker.reset();  // (its shard runs on the caller's thread)
        }
    }
    *out = m;
    return VS_OK;
}
extern "C" int vs_multi_create(vs_index* src, const int* devices, uint32_t n, uint32_t flags, vs_multi** out) {
    return vs_guard("vs_multi_create", [&] { return vs_multi_create_impl(src, devices, n, flags, out); });
}
extern "C" uint32_t vs_multi_size(const vs_multi* m) { return m ? (uint32_t)m->devs.size() : 0; }
extern "C" vs_index* vs_multi_index(vs_multi* m, uint32_t i) { return (m && i < m->devs.size()) ? m->devs[i].ix : nullptr; }
extern "C" vs_ctx* vs_multi_ctx(vs_multi* m, uint32_t i) { return (m && i < m->devs.size()) ? m->devs[i].ctx : nullptr; }

static void add_stats(vs_stats& a, const vs_stats& b) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(&b);
    uint64_t* d = reinterpret_cast<uint64_t*>(&a);
    for (size_t i = 0; i < sizeof(vs_stats) / 8; ++i) d[i] += s[i];
}

static int vs_multi_search_impl(vs_multi* m, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                                uint32_t L, uint32_t rescore, uint32_t k, bool stream_only, uint32_t* out_ids, uint64_t* out_tids,
                                float* out_dist, uint32_t* out_ham, vs_stats* stats) {
    VS_REQUIRE(m && !m->devs.empty(), "vs_multi_search_batch: no devices");
    VS_REQUIRE(nq == 0 || (queries && out_ids), "vs_multi_search_batch: bad args");
    if (stats) memset(stats, 0, sizeof(*stats));
    const uint32_t world = (uint32_t)m->devs.size();
    const uint32_t dim = m->devs[0].ix->d.dim_full;
    std::vector<int> rc(world, VS_OK);
    std::vector<std::string> err(world);
    std::vector<vs_stats> st(world);
    auto work = [&](uint32_t g) {
        try {
        uint32_t b, e;
        shard_of(nq, world, g, &b, &e);
        memset(&st[g], 0, sizeof(vs_stats));
        if (b == e) return;
        // the shard's label keys: the CSR offsets rebased to the shard's first key
        std::vector<uint32_t> off;
        const int16_t* ql = nullptr;
        if (qlabel_off) {
            off.resize(e - b + 1);
            for (uint32_t q = b; q <= e; ++q) off[q - b] = qlabel_off[q] - qlabel_off[b];
            ql = qlabels + qlabel_off[b];
        }
        vs_index* ix = m->devs[g].ix;
        int r;
        if (stream_only)
            r = vs_stream_batch(ix, queries + (size_t)b * dim, ql, qlabel_off ? off.data() : nullptr, e - b, L, k, out_ids + (size_t)b * k,
                                out_ham ? out_ham + (size_t)b * k : nullptr, &st[g]);
        else
            r = vs_search_batch(ix, queries + (size_t)b * dim, ql, qlabel_off ? off.data() : nullptr, e - b, L, rescore, k,
                                out_ids + (size_t)b * k, out_tids ? out_tids + (size_t)b * k : nullptr,
                                out_dist ? out_dist + (size_t)b * k : nullptr, &st[g]);
        rc[g] = r;
        if (r != VS_OK) err[g] = vs_last_error();  // (thread-local: carried to the caller's thread below)
        } catch (const std::exception& ex) {  // (nothing may leave a worker thread)
            rc[g] = VS_ERR_OOM;
            try {
                err[g] = ex.what();
            } catch (...) {
            }
        }
    };
    // every shard but the first on its device's persistent worker; a device without one (thread creation failed) on this thread
    std::lock_guard<std::mutex> batch(m->batch_mu);
    std::vector<uint32_t> inline_shards, posted;
    for (uint32_t g = 1; g < world; ++g) {
        MultiWorker* w = m->devs[g].worker.get();
        if (w && w->post([&work, g] { work(g); })) posted.push_back(g);
        else inline_shards.push_back(g);
    }
    work(0);
    for (uint32_t g : inline_shards) work(g);
    for (uint32_t g : posted) m->devs[g].worker->wait();
    for (uint32_t g = 0; g < world; ++g) {
        if (rc[g] != VS_OK) {
            vs_set_error("vs_multi_search_batch: device %d (shard %u of %u): %s", m->devs[g].ctx->device, g, world, err[g].c_str());
            return rc[g];
        }
        if (stats) add_stats(*stats, st[g]);
    }
    return VS_OK;
}
extern "C" int vs_multi_search_batch(vs_multi* m, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                                     uint32_t L, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids, float* out_dist,
                                     vs_stats* stats) {
    return vs_guard("vs_multi_search_batch", [&] {
        return vs_multi_search_impl(m, queries, qlabels, qlabel_off, nq, L, rescore, k, false, out_ids, out_tids, out_dist, nullptr, stats);
    });
}
extern "C" int vs_multi_stream_batch(vs_multi* m, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                                     uint32_t L, uint32_t mrows, uint32_t* out_ids, uint32_t* out_ham, vs_stats* stats) {
    return vs_guard("vs_multi_stream_batch", [&] {
        return vs_multi_search_impl(m, queries, qlabels, qlabel_off, nq, L, 0, mrows, true, out_ids, nullptr, nullptr, out_ham, stats);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// vs_comm: one process per device, RCCL over xGMI.  The entry points of librccl this file uses, with the C ABI rccl.h declares
// for them (rccl.h:40-43 ncclUniqueId, :187 ncclGetUniqueId, :220 ncclCommInitRank, :260 ncclCommDestroy, :339 ncclGetErrorString,
// ncclAllGather, ncclBroadcast, ncclGroupStart / ncclGroupEnd); the header itself is not needed.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclUint8 = 1, kNcclUint32 = 3 };
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::string path;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return VS_OK;
    // VS_RCCL_LIB names the library (the CPU test tier points it at tests/emu/libfakerccl.so); otherwise a librccl this process
    // has already mapped is reused (one RCCL per process), then the ROCm installation's
    const char* env = vs_opt_get("VS_RCCL_LIB");
    void* h = nullptr;
    std::string tried;
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        tried = env;
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names)
            if (!h) h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char* nm : names) {
            if (!h) h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            tried += std::string(tried.empty() ? "" : ", ") + nm;
        }
    }
    if (!h) {
        vs_set_error("vs_comm: RCCL is not loadable (%s): %s", tried.c_str(), dlerror());
        return VS_ERR_HIP;
    }
    Rccl r;
    r.h = h;
    bool ok = true;
    auto sym = [&](const char* nm) -> void* {
        void* p = dlsym(h, nm);
        if (!p) {
            vs_set_error("vs_comm: %s has no symbol %s", tried.c_str(), nm);
            ok = false;
        }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    if (!ok) {
        dlclose(h);
        return VS_ERR_HIP;
    }
    g_rccl = r;
    return VS_OK;
}
}  // namespace

#define VS_NCCL(expr)                                                                                             \
    do {                                                                                                          \
        const int _e = (expr);                                                                                    \
        if (_e != kNcclSuccess) {                                                                                 \
            vs_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_e), __FILE__, __LINE__);          \
            return VS_ERR_HIP;                                                                                    \
        }                                                                                                         \
    } while (0)

struct vs_comm {
    vs_ctx* ctx = nullptr;
    NcclComm comm = nullptr;
    uint32_t rank = 0, world = 1;
    DevBuf scratch;  // a few words on the device (sizes travel through it before the arrays they describe)
};

extern "C" int vs_comm_unique_id(uint8_t* id) {
    return vs_guard("vs_comm_unique_id", [&]() -> int {
        VS_REQUIRE(id, "vs_comm_unique_id: id is NULL");
        VS_TRY(rccl_load());
        NcclId u;
        memset(&u, 0, sizeof(u));
        VS_NCCL(g_rccl.GetUniqueId(&u));
        memcpy(id, u.internal, VS_COMM_ID_BYTES);
        return VS_OK;
    });
}

extern "C" void vs_comm_destroy(vs_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    devbuf_free(c->scratch);
    delete c;
}

extern "C" int vs_comm_create(vs_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t world, vs_comm** out) {
    return vs_guard("vs_comm_create", [&]() -> int {
        VS_REQUIRE(ctx && id && out && world >= 1 && rank < world, "vs_comm_create: bad args (rank %u of %u)", rank, world);
        *out = nullptr;
        VS_TRY(rccl_load());
        VS_HIP(hipSetDevice(ctx->device));  // ncclCommInitRank binds the communicator to the calling thread's device
        NcclId u;
        memcpy(u.internal, id, VS_COMM_ID_BYTES);
        vs_comm* c = new vs_comm();
        c->ctx = ctx;
        c->rank = rank;
        c->world = world;
        const int e = g_rccl.CommInitRank(&c->comm, (int)world, u, (int)rank);
        if (e != kNcclSuccess) {
            vs_set_error("ncclCommInitRank(rank %u of %u, device %d) failed: %s", rank, world, ctx->device, g_rccl.GetErrorString(e));
            c->comm = nullptr;
            vs_comm_destroy(c);
            return VS_ERR_HIP;
        }
        const int r = devbuf_reserve(ctx, c->scratch, 8192);
        if (r != VS_OK) {
            vs_comm_destroy(c);
            return r;
        }
        *out = c;
        return VS_OK;
    });
}
extern "C" uint32_t vs_comm_rank(const vs_comm* c) { return c ? c->rank : 0; }
extern "C" uint32_t vs_comm_world(const vs_comm* c) { return c ? c->world : 0; }

// chunks of at most 1 GiB: one collective per chunk keeps RCCL's element counts and its staging well inside what it is tested with
static int bcast_bytes(vs_comm* c, void* d_buf, size_t bytes, uint32_t root) {
    const size_t kChunk = (size_t)1 << 30;
    for (size_t o = 0; o < bytes; o += kChunk) {
        const size_t nb = std::min(kChunk, bytes - o);
        VS_NCCL(g_rccl.Broadcast((const char*)d_buf + o, (char*)d_buf + o, nb, kNcclUint8, (int)root, c->comm, c->ctx->stream));
    }
    return VS_OK;
}
extern "C" int vs_comm_bcast(vs_comm* c, void* d_buf, size_t bytes, uint32_t root) {
    return vs_guard("vs_comm_bcast", [&]() -> int {
        VS_REQUIRE(c && (d_buf || bytes == 0) && root < c->world, "vs_comm_bcast: bad args");
        VS_HIP(hipSetDevice(c->ctx->device));
        return bcast_bytes(c, d_buf, bytes, root);
    });
}

// The final top-k gather (north_star: "RCCL over xGMI used only for a final top-k gather").  d_ids / d_dist: this rank's
// [nq_local][k] blocks; d_out_ids / d_out_dist: [nq_total][k] on every rank, shards in rank order.  Shard sizes are arithmetic
// (vs_shard_range of nq_total): nothing is exchanged to learn them and nothing synchronises with the host.  Enqueued on the
// context's stream (after the search that produced the blocks); vs_ctx_sync completes it.
static int vs_comm_gather_topk_impl(vs_comm* c, const uint32_t* d_ids, const float* d_dist, uint32_t nq_local, uint32_t nq_total,
                                    uint32_t k, uint32_t* d_out_ids, float* d_out_dist) {
    VS_REQUIRE(c && k >= 1 && (nq_total == 0 || d_out_ids), "vs_comm_gather_topk: bad args");
    uint32_t b, e;
    shard_of(nq_total, c->world, c->rank, &b, &e);
    VS_REQUIRE(e - b == nq_local, "vs_comm_gather_topk: rank %u of %u holds %u rows, the shard of a %u-row batch is %u", c->rank,
               c->world, nq_local, nq_total, e - b);
    VS_REQUIRE(nq_local == 0 || d_ids, "vs_comm_gather_topk: d_ids is NULL");
    VS_REQUIRE((d_dist == nullptr) == (d_out_dist == nullptr), "vs_comm_gather_topk: distances in and out go together");
    if (nq_total == 0) return VS_OK;
    VS_HIP(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    const bool even = nq_total % c->world == 0;
    VS_NCCL(g_rccl.GroupStart());  // ids and distances: one fused launch
    int rc = kNcclSuccess;
    if (even) {
        rc = g_rccl.AllGather(d_ids, d_out_ids, (size_t)nq_local * k, kNcclUint32, c->comm, s);
        if (rc == kNcclSuccess && d_dist) rc = g_rccl.AllGather(d_dist, d_out_dist, (size_t)nq_local * k, kNcclUint32, c->comm, s);
    } else {  // all-gather-v: one broadcast per rank of that rank's block into its place
        for (uint32_t r = 0; r < c->world && rc == kNcclSuccess; ++r) {
            uint32_t rb, re;
            shard_of(nq_total, c->world, r, &rb, &re);
            if (re == rb) continue;
            const size_t cnt = (size_t)(re - rb) * k;
            rc = g_rccl.Broadcast(r == c->rank ? (const void*)d_ids : (const void*)(d_out_ids + (size_t)rb * k), d_out_ids + (size_t)rb * k, cnt,
                                  kNcclUint32, (int)r, c->comm, s);
            if (rc == kNcclSuccess && d_dist)
                rc = g_rccl.Broadcast(r == c->rank ? (const void*)d_dist : (const void*)(d_out_dist + (size_t)rb * k),
                                      d_out_dist + (size_t)rb * k, cnt, kNcclUint32, (int)r, c->comm, s);
        }
    }
    const int rg = g_rccl.GroupEnd();
    VS_NCCL(rc);
    VS_NCCL(rg);
    return VS_OK;
}
extern "C" int vs_comm_gather_topk(vs_comm* c, const uint32_t* d_ids, const float* d_dist, uint32_t nq_local, uint32_t nq_total, uint32_t k,
                                   uint32_t* d_out_ids, float* d_out_dist) {
    return vs_guard("vs_comm_gather_topk", [&] { return vs_comm_gather_topk_impl(c, d_ids, d_dist, nq_local, nq_total, k, d_out_ids, d_out_dist); });
}

// Replicate the root's index into the index every other rank allocated with the same geometry (vs_index_alloc): the arrays travel
// HBM to HBM over xGMI; what has a data-dependent size (label CSR, start map) is announced first through a few words.
static int vs_comm_replicate_index_impl(vs_comm* c, vs_index* ix, uint32_t root) {
    VS_REQUIRE(c && ix && root < c->world, "vs_comm_replicate_index: bad args");
    VS_REQUIRE(ix->ctx->device == c->ctx->device, "vs_comm_replicate_index: the index lives on device %d, the communicator on %d",
               ix->ctx->device, c->ctx->device);
    VS_REQUIRE(!ix->is_view, "vs_comm_replicate_index: a view does not own its arrays");
    VS_REQUIRE(c->rank == root || vs_index_live_views(ix) == 0, "vs_comm_replicate_index: views of the receiving index are alive");
    VS_HIP(hipSetDevice(c->ctx->device));
    const bool is_root = c->rank == root;
    hipStream_t s = c->ctx->stream;
    if (ix->ctx != c->ctx) VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    // ---- header: geometry check + the variable sizes
    uint64_t hdr[24] = {0};
    if (is_root) {
        const uint64_t v[] = {ix->d.n, ix->d.dim_full, ix->d.dim_index, ix->d.bits, ix->d.words, ix->d.num_neighbors, ix->d.distance_type,
                              ix->d.has_labels, ix->d.default_start, ix->d.n_label_starts, ix->d.storage_type, ix->count, ix->n_label_vals,
                              (uint64_t)(ix->vecs != nullptr), (uint64_t)(ix->vnorm_idx != nullptr), (uint64_t)(ix->label_off != nullptr),
                              (uint64_t)(ix->visible != nullptr), ix->build_unreachable};
        memcpy(hdr, v, sizeof(v));
        VS_HIP(hipMemcpyAsync(c->scratch.p, hdr, sizeof(hdr), hipMemcpyHostToDevice, s));
    }
    VS_TRY(bcast_bytes(c, c->scratch.p, sizeof(hdr), root));
    VS_HIP(hipMemcpyAsync(hdr, c->scratch.p, sizeof(hdr), hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    // ---- every rank checks the header against its own index and makes every allocation the transfer needs, then the ranks exchange
    // one word each: a rank that cannot go on (another geometry, out of device memory) fails the call on ALL ranks here, before
    // anyone is inside a broadcast the others would wait in forever
    const size_t n = std::max<uint32_t>(ix->d.n, 1);
    const uint32_t nls = (uint32_t)hdr[9];
    int my_rc = VS_OK;
    auto prepare = [&]() -> int {
        if (is_root) return VS_OK;
        VS_REQUIRE(hdr[0] == ix->d.n && hdr[1] == ix->d.dim_full && hdr[2] == ix->d.dim_index && hdr[3] == ix->d.bits && hdr[4] == ix->d.words &&
                       hdr[5] == ix->d.num_neighbors && hdr[6] == ix->d.distance_type && hdr[10] == ix->d.storage_type,
                   "vs_comm_replicate_index: rank %u allocated another geometry than the root's (n %u vs %llu, dim %u vs %llu)", c->rank,
                   ix->d.n, (unsigned long long)hdr[0], ix->d.dim_full, (unsigned long long)hdr[1]);
        VS_REQUIRE((hdr[13] != 0) == (ix->vecs != nullptr), "vs_comm_replicate_index: the root %s the heap vectors, rank %u %s",
                   hdr[13] ? "holds" : "does not hold", c->rank, ix->vecs ? "allocated them" : "did not allocate them");
        if (hdr[14] && !ix->vnorm_idx) VS_HIP(hipMalloc(&ix->vnorm_idx, n * 4));
        // label sets (AM/labels/mod.rs:15-37): the root's, or none when the root has none
        if (ix->label_off) VS_HIP(hipFree(ix->label_off));
        if (ix->label_val) VS_HIP(hipFree(ix->label_val));
        ix->label_off = nullptr;
        ix->label_val = nullptr;
        ix->n_label_vals = 0;
        ix->d.has_labels = 0;
        if (!hdr[15]) {  // (the derived masks of an earlier label set go with it)
            if (ix->label_mask) VS_HIP(hipFree(ix->label_mask));
            if (ix->label_bit) VS_HIP(hipFree(ix->label_bit));
            if (ix->nbr_mask) VS_HIP(hipFree(ix->nbr_mask));
            ix->label_mask = nullptr;
            ix->label_bit = nullptr;
            ix->nbr_mask = nullptr;
        }
        ix->nbr_mask_valid = false;
        ix->nbr_mask_tried = false;
        if (hdr[15]) {
            VS_HIP(hipMalloc(&ix->label_off, (n + 1) * 4));
            VS_HIP(hipMalloc(&ix->label_val, std::max<uint64_t>(hdr[12], 1) * 2));
        }
        if (ix->ls_labels) VS_HIP(hipFree(ix->ls_labels));
        if (ix->ls_nodes) VS_HIP(hipFree(ix->ls_nodes));
        ix->ls_labels = nullptr;
        ix->ls_nodes = nullptr;
        ix->d.n_label_starts = 0;
        if (nls) {
            VS_HIP(hipMalloc(&ix->ls_labels, (size_t)nls * 2));
            VS_HIP(hipMalloc(&ix->ls_nodes, (size_t)nls * 4));
        }
        if (hdr[16] && !ix->visible_own) VS_HIP(hipMalloc(&ix->visible_own, n));
        if (!hdr[16]) ix->visible = nullptr;  // the root scans without a mask: so does the replica
        return VS_OK;
    };
    my_rc = prepare();
    const std::string my_err = my_rc == VS_OK ? std::string() : std::string(vs_last_error());
    if (my_rc != VS_OK) (void)hipGetLastError();
    {
        VS_REQUIRE((size_t)c->world * 4 + 1024 + 4 <= c->scratch.bytes, "vs_comm_replicate_index: world of %u ranks", c->world);
        uint32_t* words = reinterpret_cast<uint32_t*>((char*)c->scratch.p + 1024);  // [world] gathered, then this rank's own word
        const uint32_t mine = my_rc == VS_OK ? 1u : 0u;
        VS_HIP(hipMemcpyAsync(words + c->world, &mine, 4, hipMemcpyHostToDevice, s));
        VS_NCCL(g_rccl.AllGather(words + c->world, words, 1, kNcclUint32, c->comm, s));
        std::vector<uint32_t> ok(c->world);
        VS_HIP(hipMemcpyAsync(ok.data(), words, (size_t)c->world * 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        for (uint32_t r = 0; r < c->world; ++r) {
            if (ok[r]) continue;
            if (my_rc != VS_OK) vs_set_error("%s", my_err.c_str());
            else vs_set_error("vs_comm_replicate_index: rank %u could not take the root's index (see its error); nothing was transferred", r);
            return my_rc != VS_OK ? my_rc : VS_ERR_STATE;
        }
    }
    if (!is_root) {
        ix->d.default_start = (uint32_t)hdr[8];
        ix->count = hdr[11];
        ix->build_unreachable = (uint32_t)hdr[17];
        ix->n_label_vals = hdr[15] ? hdr[12] : 0;
        ix->d.has_labels = hdr[15] ? (uint32_t)hdr[7] : 0;
        ix->d.n_label_starts = nls;
    }
    VS_TRY(bcast_bytes(c, ix->codes, n * ix->code_stride * 8, root));
    VS_TRY(bcast_bytes(c, ix->nbrs, n * ix->nbr_stride * 4, root));
    VS_TRY(bcast_bytes(c, ix->tids, n * 8, root));
    VS_TRY(bcast_bytes(c, ix->mean, (size_t)ix->d.dim_index * 4, root));
    VS_TRY(bcast_bytes(c, ix->m2, (size_t)ix->d.dim_index * 4, root));
    if (ix->vecs) {
        VS_TRY(bcast_bytes(c, ix->vecs, n * ix->vec_stride * 4, root));
        VS_TRY(bcast_bytes(c, ix->vnorm, n * 4, root));
    }
    if (hdr[14]) VS_TRY(bcast_bytes(c, ix->vnorm_idx, n * 4, root));
    if (hdr[15]) {  // CSR travels; the derived masks are rebuilt locally
        VS_TRY(bcast_bytes(c, ix->label_off, ((size_t)ix->d.n + 1) * 4, root));
        VS_TRY(bcast_bytes(c, ix->label_val, (size_t)ix->n_label_vals * 2, root));
        if (!is_root) {
            VS_HIP(hipStreamSynchronize(s));
            VS_TRY(vs_refresh_label_masks(ix));
        }
    }
    if (nls) {
        VS_TRY(bcast_bytes(c, ix->ls_labels, (size_t)nls * 2, root));
        VS_TRY(bcast_bytes(c, ix->ls_nodes, (size_t)nls * 4, root));
    }
    if (hdr[16]) {  // the visibility mask in force
        VS_TRY(bcast_bytes(c, is_root ? (void*)ix->visible : (void*)ix->visible_own, n, root));
        if (!is_root) ix->visible = ix->visible_own;
    }
    VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}
extern "C" int vs_comm_replicate_index(vs_comm* c, vs_index* ix, uint32_t root) {
    return vs_guard("vs_comm_replicate_index", [&] { return vs_comm_replicate_index_impl(c, ix, root); });
}
