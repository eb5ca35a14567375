// vs_shm.cpp — the cross-process half of the broker: PostgreSQL serves every connection from its own
// single-threaded backend PROCESS (amcanparallel = false, AM/mod.rs:63; one TSVScanState per IndexScanDesc,
// AM/scan.rs:308-333), so the scans that must share a launch arrive from different address spaces.  This file is the
// transport: a POSIX shared-memory segment with one slot per in-flight scan, a dispatcher (in the one process that owns the
// vs_ctx / vs_index) that gathers posted slots and runs every group sharing (search_list_size, rescore, k, label key present)
// as one vs_search_batch(), and a client side that needs no HIP at all — a backend maps the segment, posts its query, sleeps on
// a futex in its slot and wakes with the rows of its first k amgettuple calls.  In a PGRX deployment the segment is a DSM
// segment, the futex a latch and the dispatcher a background worker; slots, states, grouping and batching are what is built
// and tested here (tests/test_gpu_zu_shm.py: client PROCESSES against one dispatcher).
//
// Slot life cycle (state word, all transitions by compare-and-swap or release stores):
//   FREE -> CLAIMED (client fills the request) -> READY (posted) -> RUNNING (dispatcher took it) -> DONE (results in the slot)
//   -> FREE (client copied them out).  A slot whose owner process died is reclaimed by the dispatcher (CLAIMED / DONE ->
//   REAPING -> FREE, owner pid cleared before the slot is free again, so a slot is never freed under a live client); a client
//   whose dispatcher PROCESS died (no orderly vs_shm_server_destroy) notices through the pid in the header and fails with
//   VS_ERR_STATE instead of sleeping forever.  The dispatcher trusts nothing it reads from a slot, and reads it ONCE: when a
//   posted slot is taken (READY -> RUNNING) its request — k, the GUCs, the label key, the query vector, the scan id — is copied
//   into the dispatcher's own memory and validated THERE; everything that follows (grouping, launches, how many rows are
//   written back and where) uses that copy and the geometry the dispatcher wrote, never the slot or the header again.
//
// Streaming (amgettuple beyond the first rows): a slot request is either OP_SEARCH — the first k rows, out of a launch shared
// with the other backends' scans — or OP_FETCH: rows [skip, skip + k) of the scan (owner pid, scan_id), which the dispatcher
// serves from a cursor it keeps for that scan on the device (a direct vs_scan of its own index: lsr + resort_buffer of
// AM/scan.rs:162-174 live there between requests).  Every OP_FETCH carries the whole scan description, so a cursor the dispatcher
// no longer has (evicted, or never opened because the first rows came from OP_SEARCH) is opened again and fast-forwarded to
// `skip` — one replay — and then continued; OP_CLOSE drops it (so does the death of the owner, and the least recently used one
// when more than max_cursors are open).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <shared_mutex>
#include <mutex>
#include <cerrno>
#include <chrono>
#include <climits>
#include <csignal>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/vsgpu.h"

void vs_set_error(const char* fmt, ...);
const char* vs_opt_get(const char* name);  // vs_options.cpp: the option table (vs_set_option, snapshot of the VS_* environment)

namespace {

constexpr uint32_t SHM_MAGIC = 0x56534851u;  // "VSHQ"
constexpr uint32_t SHM_VERSION = 3;
constexpr size_t SHM_MAX_CURSORS = 1024;
enum : uint32_t { OP_SEARCH = 0, OP_FETCH = 1, OP_CLOSE = 2 };
constexpr uint32_t SHM_MAX_LABELS = 64;
enum : uint32_t { S_FREE = 0, S_CLAIMED = 1, S_READY = 2, S_RUNNING = 3, S_DONE = 4, S_REAPING = 5 };

struct ShmHeader {
    uint32_t magic, version;
    uint32_t nslots, dim_full, kmax;
    uint32_t slot_bytes;
    std::atomic<uint32_t> serving;   // 1 while a dispatcher is attached
    std::atomic<uint32_t> work_seq;  // bumped (and futex-woken) by every post
    int32_t server_pid;
    uint32_t pad[7];
};

struct SlotHead {
    std::atomic<uint32_t> state;
    int32_t owner_pid;
    uint32_t L, rescore, k, n_labels, has_label_key, null_query;
    int32_t rc;
    uint32_t snapshot;  // visibility mask of the serving process the scan runs under (0 = every tuple visible)
    uint32_t op;        // OP_*
    uint32_t skip;      // OP_FETCH: rows of the scan the client already has
    uint32_t n_rows;    // OP_FETCH, out: rows returned (< k: the scan has ended)
    uint32_t pad0;
    uint64_t scan_id;   // OP_FETCH / OP_CLOSE: the client's name for the scan (unique per client process)
    char err[168];
    int16_t labels[SHM_MAX_LABELS];
    // followed by: float query[dim_full]; uint64_t out_tids[kmax]; uint32_t out_ids[kmax]; float out_dist[kmax]
};

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
size_t slot_size(uint32_t dim_full, uint32_t kmax) {
    return align16(sizeof(SlotHead) + (size_t)dim_full * 4 + (size_t)kmax * 16);
}

struct Mapping {
    void* base = nullptr;
    size_t bytes = 0;
    // the segment's geometry as this process read it ONCE (server: as it wrote it): the header lives in memory every client can
    // write, so no address is ever computed from the header again
    uint32_t nslots = 0, dim_full = 0, kmax = 0, slot_bytes = 0;
    void pin() {
        const ShmHeader* h = hdr();
        nslots = h->nslots;
        dim_full = h->dim_full;
        kmax = h->kmax;
        slot_bytes = h->slot_bytes;
    }
    ShmHeader* hdr() const { return static_cast<ShmHeader*>(base); }
    SlotHead* slot(uint32_t i) const {
        return reinterpret_cast<SlotHead*>(static_cast<char*>(base) + align16(sizeof(ShmHeader)) + (size_t)i * slot_bytes);
    }
    static float* query(SlotHead* s) { return reinterpret_cast<float*>(reinterpret_cast<char*>(s) + sizeof(SlotHead)); }
    uint64_t* tids(SlotHead* s) const { return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(query(s)) + align16((size_t)dim_full * 4)); }
    uint32_t* ids(SlotHead* s) const { return reinterpret_cast<uint32_t*>(tids(s) + kmax); }
    float* dist(SlotHead* s) const { return reinterpret_cast<float*>(ids(s) + kmax); }
};

int futex_wait(std::atomic<uint32_t>* addr, uint32_t expected, int timeout_us) {
    timespec ts{timeout_us / 1000000, (timeout_us % 1000000) * 1000L};
    return (int)syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAIT, expected, timeout_us >= 0 ? &ts : nullptr, nullptr, 0);
}
void futex_wake(std::atomic<uint32_t>* addr, int n) { syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAKE, n, nullptr, nullptr, 0); }

// the dispatcher process is gone (crashed or was killed: `serving` is only cleared by an orderly shutdown)
bool server_dead(const ShmHeader* h) { return h->server_pid > 0 && kill(h->server_pid, 0) != 0 && errno == ESRCH; }

}  // namespace

struct vs_shm_server {
    vs_index* ix = nullptr;
    vs_index_desc d{};
    vs_broker_config cfg{};
    std::string name;
    Mapping m;
    std::atomic<bool> stop{false};
    std::thread dispatcher;
    std::atomic<uint64_t> batches{0}, scans{0}, max_batch{0};
    // snapshot masks waiting to be installed by the dispatcher (the only thread that touches the index)
    struct PendingPut { uint32_t snapshot; bool drop; std::vector<uint8_t> mask; int rc = 1; std::string err; };
    std::mutex put_mu;
    std::condition_variable put_cv;
    std::vector<std::shared_ptr<PendingPut>> puts;  // (shared: a caller that gives up at shutdown leaves nothing dangling behind)
    std::atomic<int> put_callers{0};               // threads inside vs_shm_server_snapshot_put (destroy waits for them)
    // cursors of the scans that are being streamed (OP_FETCH), touched by the dispatcher thread only
    struct Cursor {
        int32_t pid = 0;
        uint64_t scan_id = 0, sig = 0, last_use = 0;
        uint32_t pos = 0;  // rows handed out so far
        vs_scan* scan = nullptr;
    };
    // a table of cursors and the handle their scans run through: the dispatcher's own (the index itself), or one per cursor lane
    struct CursorTable {
        vs_index* ix = nullptr;
        std::vector<Cursor> cursors;
        uint64_t use_clock = 0;
    };
    CursorTable main_tab;
    // the dispatcher's private copy of a taken slot's request (see the header comment): filled by take(), read by run_group /
    // run_fetch (a lane reads the copy of the slot it was handed: the slot is RUNNING until that lane marks it DONE)
    struct Req {
        uint32_t L = 0, rescore = 0, k = 0, n_labels = 0, has_label_key = 0, null_query = 0, snapshot = 0, op = 0, skip = 0;
        uint64_t scan_id = 0;
        int32_t owner_pid = 0;
        int16_t labels[SHM_MAX_LABELS] = {0};
        std::vector<float> query;
    };
    std::vector<Req> reqs;
    // dispatcher-private: the slot's last accepted request has not been marked DONE by the SERVER yet.  A client that writes READY over
    // the state word of its RUNNING slot is not taken again (its request would be copied over the one a lane is still reading, and the
    // slot would be queued twice): the slot is looked at again once the server itself has finished it
    std::unique_ptr<std::atomic<uint8_t>[]> in_flight;
    bool posted(uint32_t slot) const {
        return m.slot(slot)->state.load(std::memory_order_acquire) == S_READY && !in_flight[slot].load(std::memory_order_acquire);
    }
    void finish(uint32_t slot) {  // DONE, by the server: the slot may be taken again
        SlotHead* s = m.slot(slot);
        in_flight[slot].store(0, std::memory_order_release);
        s->state.store(S_DONE, std::memory_order_release);
        futex_wake(&s->state, 1);
    }
    const char* take(uint32_t slot);  // copies into a local, validates, publishes into reqs[slot] on success; nullptr = accepted, else why not
    // cursor lanes (vs_broker_config.cursor_lanes): a thread, a context (HIP stream) and a view of the index each; a streamed scan
    // is served by the lane its (client pid, scan id) hashes to, concurrently with the other lanes and with the shared launches
    struct Lane {
        vs_ctx* ctx = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<uint32_t> q;  // slots (already S_RUNNING) waiting for this lane
        bool stop = false;
        CursorTable tab;
    };
    // scan pools (vs_broker_config.cursor_pool): the streamed scans of the clients kept in pooled device arrays, the continuations of
    // one dispatcher round served by shared launches (vs_scanpool.cpp); dispatcher thread only
    struct PoolCursor {
        int32_t pid = 0;
        uint64_t scan_id = 0, sig = 0, last_use = 0;
        uint64_t round = 0;  // the dispatcher round that last put a request of this scan on its list (never evicted within that round)
        uint32_t pos = 0;  // rows handed out so far
        bool used = false;
    };
    struct Pool {
        vs_scan_pool* h = nullptr;
        uint32_t L = 0, rescore = 0, snapshot = 0;
        std::vector<PoolCursor> cur;  // by pool slot
        uint64_t use_clock = 0;
        uint64_t last_round = 0;  // the last dispatcher round that served a request out of this pool
    };
    std::vector<Pool> pools;
    uint64_t pool_round_no = 0;
    static constexpr size_t MAX_POOLS = 4;
    std::atomic<uint64_t> pool_rounds{0};
    void run_fetch_pooled(const std::vector<uint32_t>& slots);
    void reap_pools();
    void free_pools();
    std::vector<std::unique_ptr<Lane>> lanes;
    std::shared_mutex snap_mu;        // lanes: shared for the length of a request; a mask is replaced exclusively
    std::atomic<int> put_waiting{0};  // ... and lanes do not start a request while one waits (readers would starve the writer)
    std::atomic<uint64_t> fetches{0}, cursor_opens{0}, open_cursors{0}, pools_retired{0}, pools_alive{0}, pooled_scans{0};
    void apply_puts();
    void run();
    void run_lane(Lane& ln);
    void run_group(const std::vector<uint32_t>& grp);
    void run_fetch(uint32_t slot, CursorTable& t);
    void drop_cursor(CursorTable& t, size_t i);
    void reap_cursors(CursorTable& t);
};

struct vs_shm_client {
    Mapping m;
};

const char* vs_shm_server::take(uint32_t slot) {
    SlotHead* s = m.slot(slot);
    Req r;  // (a rejected request leaves reqs[slot] as it was: nothing unvalidated is ever visible to run_group / run_fetch)
    r.L = s->L;
    r.rescore = s->rescore;
    r.k = s->k;
    r.n_labels = s->n_labels;
    r.has_label_key = s->has_label_key;
    r.null_query = s->null_query;
    r.snapshot = s->snapshot;
    r.op = s->op;
    r.skip = s->skip;
    r.scan_id = s->scan_id;
    r.owner_pid = s->owner_pid;
    if (r.k == 0 || r.k > m.kmax) return "k outside [1, kmax]";
    if (r.n_labels > SHM_MAX_LABELS) return "more labels than a slot holds";
    if (r.L < 1 || r.L > 10000) return "diskann.query_search_list_size outside [1,10000]";
    if (r.rescore > 1000) return "diskann.query_rescore outside [0,1000]";
    if (r.snapshot >= VS_MAX_SNAPSHOTS) return "snapshot id out of range";
    if (r.op > OP_CLOSE) return "unknown request kind";
    if (r.op == OP_FETCH && (uint64_t)r.skip + r.k > (1u << 30)) return "row position out of range";
    memcpy(r.labels, s->labels, (size_t)r.n_labels * sizeof(int16_t));
    r.query = std::move(reqs[slot].query);  // (the slot's buffer is kept from request to request: no allocation per cursor request)
    r.query.resize(d.dim_full);
    if (!r.null_query) memcpy(r.query.data(), Mapping::query(s), (size_t)d.dim_full * 4);
    reqs[slot] = std::move(r);
    in_flight[slot].store(1, std::memory_order_release);
    return nullptr;
}

void vs_shm_server::run_group(const std::vector<uint32_t>& grp) {
    const uint32_t nq = (uint32_t)grp.size();
    const Req& head = reqs[grp[0]];
    const uint32_t k = head.k;
    const bool keys = head.has_label_key != 0;
    int rc = VS_OK;
    std::string err;
    std::vector<float> q;
    std::vector<int16_t> lab;
    std::vector<uint32_t> off, ids;
    std::vector<uint64_t> tids;
    std::vector<float> dist;
    try {
        q.assign((size_t)nq * d.dim_full, 0.0f);  // a NULL query is the zero vector (AM/labels/mod.rs:214-216)
        off.assign(nq + 1, 0);
        for (uint32_t i = 0; i < nq; ++i) {
            const Req& r = reqs[grp[i]];
            if (!r.null_query) memcpy(&q[(size_t)i * d.dim_full], r.query.data(), (size_t)d.dim_full * 4);
            if (keys && !r.null_query) lab.insert(lab.end(), r.labels, r.labels + r.n_labels);
            off[i + 1] = (uint32_t)lab.size();
        }
        ids.assign((size_t)nq * k, 0);
        tids.assign((size_t)nq * k, 0);
        dist.assign((size_t)nq * k, 0.0f);
        vs_stats st{};
        const uint8_t* prev = nullptr;  // the group's snapshot mask for the duration of the launch
        rc = vs_index_snapshot_use(ix, head.snapshot, &prev);
        if (rc == VS_OK) {
            rc = vs_search_batch(ix, q.data(), keys ? lab.data() : nullptr, keys ? off.data() : nullptr, nq, head.L, head.rescore, k,
                                 ids.data(), tids.data(), dist.data(), &st);
            if (rc != VS_OK) err = vs_last_error();
            (void)vs_index_set_visibility_dev(ix, prev);
        } else {
            err = vs_last_error();
        }
    } catch (const std::bad_alloc&) {
        rc = VS_ERR_OOM;
        err = "vs_shm: out of host memory while assembling a batch";
    }
    batches++;
    scans += nq;
    if (nq > max_batch.load()) max_batch = nq;
    for (uint32_t i = 0; i < nq; ++i) {
        SlotHead* s = m.slot(grp[i]);
        if (rc == VS_OK) {
            memcpy(m.ids(s), &ids[(size_t)i * k], (size_t)k * 4);
            memcpy(m.tids(s), &tids[(size_t)i * k], (size_t)k * 8);
            memcpy(m.dist(s), &dist[(size_t)i * k], (size_t)k * 4);
        }
        s->rc = rc;
        snprintf(s->err, sizeof(s->err), "%s", err.c_str());
        finish(grp[i]);
    }
}

void vs_shm_server::drop_cursor(CursorTable& t, size_t i) {
    vs_endscan(t.cursors[i].scan);
    t.cursors[i] = t.cursors.back();
    t.cursors.pop_back();
    open_cursors--;
}

// the cursors of clients that died give their device memory back
void vs_shm_server::reap_cursors(CursorTable& t) {
    for (size_t i = 0; i < t.cursors.size();)
        if (t.cursors[i].pid > 0 && kill(t.cursors[i].pid, 0) != 0 && errno == ESRCH) drop_cursor(t, i);
        else ++i;
}

// what identifies the scan a cursor belongs to besides (pid, scan_id): a client that reuses an id for another scan gets a new cursor
template <class R>
static uint64_t scan_signature(const R* s, const float* q, uint32_t dim) {
    // (eight bytes per step: every cursor request carries its query and is hashed again — byte by byte that was 3 us per request, the
    // largest single item of a dispatcher round with 64 backends streaming; the value only has to tell scans apart inside one server)
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, b + i, 8);
            h = (h ^ w) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
        }
        for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    const uint32_t g[6] = {s->L, s->rescore, s->has_label_key, s->null_query, s->snapshot, s->n_labels};
    mix(g, sizeof(g));
    if (!s->null_query) mix(q, (size_t)dim * 4);
    mix(s->labels, (size_t)std::min<uint32_t>(s->n_labels, SHM_MAX_LABELS) * 2);
    return h;
}

void vs_shm_server::run_fetch(uint32_t slot, CursorTable& t) {
    std::vector<Cursor>& cursors = t.cursors;
    vs_index* const ix = t.ix;  // (shadows the member: the handle this table's scans run through)
    SlotHead* const out = m.slot(slot);  // written (results), never read
    const Req* const s = &reqs[slot];    // the request as it was taken
    const uint32_t req_k = s->k, req_skip = s->skip;  // (validated by take(): k <= kmax, the rows fit the slot)
    int rc = VS_OK;
    std::string err;
    uint32_t got = 0;
    const int32_t pid = s->owner_pid;
    size_t at = cursors.size();
    for (size_t i = 0; i < cursors.size(); ++i)
        if (cursors[i].pid == pid && cursors[i].scan_id == s->scan_id) at = i;
    try {
        if (s->op == OP_CLOSE) {
            if (at < cursors.size()) drop_cursor(t, at);
        } else {
            const uint64_t sig = scan_signature(s, s->query.data(), d.dim_full);
            const uint8_t* prev = nullptr;
            rc = vs_index_snapshot_use(ix, s->snapshot, &prev);
            if (rc == VS_OK) {
                if (at < cursors.size() && (cursors[at].sig != sig || cursors[at].pos != req_skip)) {
                    drop_cursor(t, at);  // another scan under the same id, or a client that is somewhere else in it: start over
                    at = cursors.size();
                }
                if (at == cursors.size()) {
                    if (cursors.size() >= SHM_MAX_CURSORS) {  // make room: the least recently used cursor goes
                        size_t lru = 0;
                        for (size_t i = 1; i < cursors.size(); ++i)
                            if (cursors[i].last_use < cursors[lru].last_use) lru = i;
                        drop_cursor(t, lru);
                    }
                    Cursor c;
                    c.pid = pid;
                    c.scan_id = s->scan_id;
                    c.sig = sig;
                    rc = vs_beginscan(ix, &c.scan);
                    if (rc == VS_OK)
                        rc = vs_rescan(c.scan, s->null_query ? nullptr : s->query.data(), s->labels, s->n_labels,
                                       (int)s->has_label_key, s->L, s->rescore);
                    if (rc == VS_OK) rc = vs_scan_prefetch(c.scan, req_skip + req_k);  // (one launch for the replay and the new rows)
                    for (uint32_t i = 0; rc == VS_OK && i < req_skip; ++i) {  // fast-forward: the client has these rows
                        const int r = vs_gettuple(c.scan, nullptr, nullptr, nullptr);
                        if (r < 0) rc = r;
                        if (r <= 0) break;
                        c.pos++;
                    }
                    if (rc == VS_OK) {
                        cursors.push_back(c);
                        open_cursors++;
                        at = cursors.size() - 1;
                        cursor_opens++;
                    } else {
                        err = vs_last_error();
                        if (c.scan) vs_endscan(c.scan);
                    }
                } else {
                    rc = vs_scan_prefetch(cursors[at].scan, cursors[at].pos + req_k);
                    if (rc != VS_OK) err = vs_last_error();
                }
                if (rc == VS_OK) {
                    Cursor& c = cursors[at];
                    c.last_use = ++t.use_clock;
                    // (a fast-forward that fell short: the scan has fewer rows than the client skipped — nothing left to return)
                    while (c.pos >= req_skip && got < req_k) {
                        const int r = vs_gettuple(c.scan, m.tids(out) + got, m.ids(out) + got, m.dist(out) + got);
                        if (r < 0) {
                            rc = r;
                            err = vs_last_error();
                            break;
                        }
                        if (r == 0) break;
                        got++;
                        c.pos++;
                    }
                }
                (void)vs_index_set_visibility_dev(ix, prev);
            } else {
                err = vs_last_error();
            }
        }
    } catch (const std::bad_alloc&) {
        rc = VS_ERR_OOM;
        err = "vs_shm: out of host memory while serving a scan cursor";
    }
    fetches++;
    out->n_rows = rc == VS_OK ? got : 0;
    out->rc = rc;
    snprintf(out->err, sizeof(out->err), "%s", err.c_str());
    finish(slot);
}

// ---- pooled continuations ---------------------------------------------------------------------------------------------------
void vs_shm_server::free_pools() {
    for (Pool& p : pools) {
        for (PoolCursor& c : p.cur)
            if (c.used) open_cursors--;
        vs_scanpool_free(p.h);
    }
    pools.clear();
}
void vs_shm_server::reap_pools() {
    for (Pool& p : pools)
        for (uint32_t i = 0; i < p.cur.size(); ++i) {
            PoolCursor& c = p.cur[i];
            if (c.used && c.pid > 0 && kill(c.pid, 0) != 0 && errno == ESRCH) {
                (void)vs_scanpool_endscan(p.h, i);
                c = PoolCursor{};
                open_cursors--;
            }
        }
}
// Every OP_FETCH / OP_CLOSE request taken in this dispatcher round.  A scan lives in the pool of its (search_list_size, rescore,
// snapshot); the requests of one pool that ask for the same number of rows are ONE vs_scanpool_fetch — one resumed search launch, one
// rerank launch for all of them.  What a pool cannot take is served by a cursor of its own (run_fetch).
void vs_shm_server::run_fetch_pooled(const std::vector<uint32_t>& slots) {
    struct Item { uint32_t slot, pool, pslot; };
    std::vector<Item> items, ff;  // requests to serve / new scans to fast-forward first
    std::vector<uint32_t> single;  // requests for the single-cursor path
    const uint64_t round_no = ++pool_round_no;
    // host allocations below (the lists of a round) must not take the dispatcher thread down: whatever has not been answered when
    // memory runs out is answered with VS_ERR_OOM (a slot is in flight from take() until finish())
    try {
    auto in_main_tab = [&](const Req& r) {
        for (const Cursor& c : main_tab.cursors)
            if (c.pid == r.owner_pid && c.scan_id == r.scan_id) return true;
        return false;
    };
    for (uint32_t slot : slots) {
        const Req& r = reqs[slot];
        // (a scan that fell back to a cursor of its own stays there)
        if (in_main_tab(r)) {
            single.push_back(slot);
            continue;
        }
        // where the scan lives, if anywhere
        int pi = -1, ci = -1;
        for (size_t a = 0; a < pools.size() && pi < 0; ++a)
            for (size_t b = 0; b < pools[a].cur.size(); ++b)
                if (pools[a].cur[b].used && pools[a].cur[b].pid == r.owner_pid && pools[a].cur[b].scan_id == r.scan_id) {
                    pi = (int)a;
                    ci = (int)b;
                    break;
                }
        if (r.op == OP_CLOSE) {
            if (pi >= 0) {
                (void)vs_scanpool_endscan(pools[pi].h, (uint32_t)ci);
                pools[pi].cur[ci] = PoolCursor{};
                open_cursors--;
                pooled_scans--;  // (before the client is answered: it may read the counters right away)
                SlotHead* out = m.slot(slot);
                out->n_rows = 0;
                out->rc = VS_OK;
                out->err[0] = 0;
                fetches++;
                finish(slot);
            } else {
                single.push_back(slot);
            }
            continue;
        }
        const uint64_t sig = scan_signature(&r, r.query.data(), d.dim_full);
        if (pi >= 0 && (pools[pi].L != r.L || pools[pi].rescore != r.rescore || pools[pi].snapshot != r.snapshot || pools[pi].cur[ci].sig != sig ||
                        pools[pi].cur[ci].pos != r.skip)) {
            // another scan under the same id, or a client that is somewhere else in it: start over
            (void)vs_scanpool_endscan(pools[pi].h, (uint32_t)ci);
            pools[pi].cur[ci] = PoolCursor{};
            open_cursors--;
            pi = ci = -1;
        }
        if (pi < 0) {
            for (size_t a = 0; a < pools.size(); ++a)
                if (pools[a].L == r.L && pools[a].rescore == r.rescore && pools[a].snapshot == r.snapshot) pi = (int)a;
            if (pi < 0 && pools.size() >= MAX_POOLS) {
                // at the cap: the pool that has served nothing for the longest time AND holds no live scan is re-keyed in place (its
                // index stays valid for the lists of this round: a pool without live scans is on none of them).  Snapshot ids come and
                // go in a PostgreSQL deployment — without this the fifth (L, rescore, snapshot) would fall back to single cursors for good.
                int victim = -1;
                for (size_t a = 0; a < pools.size(); ++a) {
                    bool live = false;
                    for (const PoolCursor& c : pools[a].cur) live = live || c.used;
                    if (!live && pools[a].last_round != round_no && (victim < 0 || pools[a].last_round < pools[victim].last_round)) victim = (int)a;
                }
                if (victim >= 0) {
                    Pool& vp = pools[victim];
                    vs_scan_pool* nh = nullptr;
                    if (vs_scanpool_create(ix, cfg.cursor_pool, r.L, r.rescore, m.kmax, 0, &nh) == VS_OK) {
                        vs_scanpool_free(vp.h);
                        vp.h = nh;
                        vp.L = r.L;
                        vp.rescore = r.rescore;
                        vp.snapshot = r.snapshot;
                        vp.cur.assign(cfg.cursor_pool, PoolCursor{});
                        vp.use_clock = 0;
                        pools_retired++;
                        pi = victim;
                    }
                }
            }
            if (pi < 0 && pools.size() < MAX_POOLS) {
                Pool np;
                np.L = r.L;
                np.rescore = r.rescore;
                np.snapshot = r.snapshot;
                if (vs_scanpool_create(ix, cfg.cursor_pool, r.L, r.rescore, m.kmax, 0, &np.h) == VS_OK) {
                    np.cur.resize(cfg.cursor_pool);
                    pools.push_back(np);
                    pi = (int)pools.size() - 1;
                }
            }
            if (pi < 0) {
                single.push_back(slot);
                continue;
            }
            Pool& p = pools[pi];
            for (size_t b = 0; b < p.cur.size() && ci < 0; ++b)
                if (!p.cur[b].used) ci = (int)b;
            if (ci < 0) {
                // make room: the least recently used scan of the pool goes (its client's next request replays it) — but never a scan
                // that already has a request on this round's list: two requests would share one pool slot, the second rescan would
                // overwrite the first scan, and vs_scanpool_fetch would be handed the slot twice
                for (size_t b = 0; b < p.cur.size(); ++b)
                    if (p.cur[b].round != round_no && (ci < 0 || p.cur[b].last_use < p.cur[ci].last_use)) ci = (int)b;
                if (ci < 0) {  // every slot is spoken for in this round: a cursor of its own
                    single.push_back(slot);
                    continue;
                }
                (void)vs_scanpool_endscan(p.h, (uint32_t)ci);
                p.cur[ci] = PoolCursor{};
                open_cursors--;
            }
            const int rc = vs_scanpool_rescan(p.h, (uint32_t)ci, r.null_query ? nullptr : r.query.data(), r.labels, r.n_labels, (int)r.has_label_key);
            if (rc != VS_OK) {
                p.cur[ci] = PoolCursor{};
                single.push_back(slot);
                continue;
            }
            PoolCursor& c = p.cur[ci];
            c.used = true;
            c.pid = r.owner_pid;
            c.scan_id = r.scan_id;
            c.sig = sig;
            c.pos = 0;
            open_cursors++;
            cursor_opens++;
            // a scan that is already under way (its first rows came out of a shared OP_SEARCH launch, or its cursor was evicted): the
            // rows the client has are reproduced once by the deterministic scan and skipped, as a single cursor does it — for all
            // the scans that start in this round together (below)
            if (r.skip > 0) ff.push_back(Item{slot, (uint32_t)pi, (uint32_t)ci});
        }
        pools[pi].cur[ci].last_use = ++pools[pi].use_clock;
        pools[pi].cur[ci].round = round_no;
        pools[pi].last_round = round_no;
        items.push_back(Item{slot, (uint32_t)pi, (uint32_t)ci});
    }
    // fast-forward of the scans that start in this round: shared fetches whose rows are thrown away, rounds of at most kmax rows
    {
        std::vector<bool> ffd(ff.size(), false);
        for (size_t a = 0; a < ff.size(); ++a) {
            if (ffd[a]) continue;
            Pool& p = pools[ff[a].pool];
            const uint32_t skip = reqs[ff[a].slot].skip;
            std::vector<size_t> grp;
            for (size_t b = a; b < ff.size(); ++b)
                if (!ffd[b] && ff[b].pool == ff[a].pool && reqs[ff[b].slot].skip == skip) {
                    ffd[b] = true;
                    grp.push_back(b);
                }
            const uint8_t* prev = nullptr;
            const bool snap_set = vs_index_snapshot_use(ix, p.snapshot, &prev) == VS_OK;
            bool ok = snap_set;
            std::vector<uint32_t> live;  // pool slots still being forwarded
            for (size_t g : grp) live.push_back(ff[g].pslot);
            uint32_t left = skip;
            while (ok && left && !live.empty()) {
                const uint32_t kk = std::min<uint32_t>(left, m.kmax);
                std::vector<int32_t> got(live.size(), 0);
                ok = vs_scanpool_fetch(p.h, live.data(), (uint32_t)live.size(), kk, nullptr, nullptr, nullptr, got.data()) == VS_OK;
                if (!ok) break;
                std::vector<uint32_t> next;
                for (size_t i = 0; i < live.size(); ++i) {
                    if (got[i] < 0) continue;  // (outgrew the pool during the replay: pos stays short, handled below)
                    p.cur[live[i]].pos += (uint32_t)got[i];
                    if ((uint32_t)got[i] == kk) next.push_back(live[i]);  // else: fewer rows than the client skipped — the scan is over
                }
                live.swap(next);
                left -= kk;
            }
            if (snap_set) (void)vs_index_set_visibility_dev(ix, prev);  // (also after a failed fetch: the index-level mask is left as it was)
            for (size_t g : grp) {
                PoolCursor& c = p.cur[ff[g].pslot];
                if (!ok) c.pos = 0xFFFFFFFFu;  // (marks "could not be forwarded": a cursor of its own)
            }
        }
    }
    // requests whose scan could not be brought to the client's position inside the pool go to a cursor of their own
    for (size_t a = 0; a < items.size();) {
        Pool& p = pools[items[a].pool];
        PoolCursor& c = p.cur[items[a].pslot];
        if (c.pos == 0xFFFFFFFFu) {
            (void)vs_scanpool_endscan(p.h, items[a].pslot);
            c = PoolCursor{};
            open_cursors--;
            single.push_back(items[a].slot);
            items.erase(items.begin() + (long)a);
            continue;
        }
        if (c.pos < reqs[items[a].slot].skip) {  // (the fast-forward fell short: the scan is over)
            SlotHead* out = m.slot(items[a].slot);
            out->n_rows = 0;
            out->rc = VS_OK;
            out->err[0] = 0;
            fetches++;
            finish(items[a].slot);
            items.erase(items.begin() + (long)a);
            continue;
        }
        ++a;
    }
    // one shared fetch per (pool, rows asked)
    std::vector<bool> done(items.size(), false);
    for (size_t a = 0; a < items.size(); ++a) {
        if (done[a]) continue;
        const uint32_t k = reqs[items[a].slot].k;
        Pool& p = pools[items[a].pool];
        std::vector<size_t> grp;
        std::vector<uint32_t> pslots;
        for (size_t b = a; b < items.size(); ++b)
            if (!done[b] && items[b].pool == items[a].pool && reqs[items[b].slot].k == k) {
                done[b] = true;
                grp.push_back(b);
                pslots.push_back(items[b].pslot);
            }
        int rc = VS_OK;
        std::string err;
        std::vector<int32_t> rows(grp.size(), 0);
        std::vector<uint32_t> ids;
        std::vector<uint64_t> tids;
        std::vector<float> dist;
        try {
            ids.assign(grp.size() * k, 0);
            tids.assign(grp.size() * k, 0);
            dist.assign(grp.size() * k, 0.0f);
            const uint8_t* prev = nullptr;
            rc = vs_index_snapshot_use(ix, p.snapshot, &prev);
            if (rc == VS_OK) {
                rc = vs_scanpool_fetch(p.h, pslots.data(), (uint32_t)pslots.size(), k, tids.data(), ids.data(), dist.data(), rows.data());
                if (rc != VS_OK) err = vs_last_error();
                (void)vs_index_set_visibility_dev(ix, prev);
            } else {
                err = vs_last_error();
            }
        } catch (const std::bad_alloc&) {
            rc = VS_ERR_OOM;
            err = "vs_shm: out of host memory while serving pooled scan cursors";
        }
        pool_rounds++;
        for (size_t g = 0; g < grp.size(); ++g) {
            const Item& it = items[grp[g]];
            PoolCursor& c = p.cur[it.pslot];
            if (rc == VS_OK && rows[g] < 0) {  // this scan outgrew the pool: it continues on a cursor of its own (one replay)
                (void)vs_scanpool_endscan(p.h, it.pslot);
                c = PoolCursor{};
                open_cursors--;
                single.push_back(it.slot);
                continue;
            }
            SlotHead* out = m.slot(it.slot);
            if (rc == VS_OK) {
                const uint32_t got = (uint32_t)rows[g];
                memcpy(m.ids(out), &ids[g * k], (size_t)got * 4);
                memcpy(m.tids(out), &tids[g * k], (size_t)got * 8);
                memcpy(m.dist(out), &dist[g * k], (size_t)got * 4);
                out->n_rows = got;
                c.pos += got;
            } else {
                out->n_rows = 0;
            }
            out->rc = rc;
            snprintf(out->err, sizeof(out->err), "%s", err.c_str());
            fetches++;
            finish(it.slot);
        }
    }
    for (uint32_t slot : single) run_fetch(slot, main_tab);
    uint64_t live_scans = 0;
    for (const Pool& p : pools)
        for (const PoolCursor& c : p.cur) live_scans += c.used ? 1 : 0;
    pools_alive.store(pools.size());
    pooled_scans.store(live_scans);
    } catch (const std::bad_alloc&) {
        for (uint32_t slot : slots) {
            if (!in_flight[slot].load(std::memory_order_acquire)) continue;
            SlotHead* out = m.slot(slot);
            out->n_rows = 0;
            out->rc = VS_ERR_OOM;
            snprintf(out->err, sizeof(out->err), "vs_shm: out of host memory while serving pooled scan cursors");
            fetches++;
            finish(slot);
        }
    }
}

void vs_shm_server::apply_puts() {
    std::unique_lock<std::mutex> lk(put_mu);
    if (puts.empty()) return;
    std::vector<std::shared_ptr<PendingPut>> mine;
    mine.swap(puts);
    lk.unlock();
    for (const std::shared_ptr<PendingPut>& p : mine) {
        int r;
        {
            struct Waiting {
                std::atomic<int>& n;
                explicit Waiting(std::atomic<int>& n_) : n(n_) { n.fetch_add(1, std::memory_order_acq_rel); }
                ~Waiting() { n.fetch_sub(1, std::memory_order_acq_rel); }
            } waiting(put_waiting);
            std::unique_lock<std::shared_mutex> xl(snap_mu);  // (no lane is inside a request)
            r = vs_index_snapshot_put(ix, p->snapshot, p->drop ? nullptr : p->mask.data());
        }
        const std::string e = r == VS_OK ? "" : vs_last_error();
        lk.lock();
        p->err = e;
        p->rc = r;
        lk.unlock();
    }
    put_cv.notify_all();
}

void vs_shm_server::run_lane(Lane& ln) {
    std::unique_lock<std::mutex> lk(ln.mu);
    auto last_reap = std::chrono::steady_clock::now();
    for (;;) {
        ln.cv.wait_for(lk, std::chrono::milliseconds(200), [&] { return ln.stop || !ln.q.empty(); });
        if (ln.q.empty() && ln.stop) break;
        if (ln.q.empty() || std::chrono::steady_clock::now() - last_reap > std::chrono::milliseconds(200)) {
            lk.unlock();
            reap_cursors(ln.tab);
            last_reap = std::chrono::steady_clock::now();
            lk.lock();
            if (ln.q.empty()) continue;
        }
        const uint32_t slot = ln.q.front();
        ln.q.pop_front();
        lk.unlock();
        {
            while (put_waiting.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
            std::shared_lock<std::shared_mutex> sl(snap_mu);
            (void)vs_index_snapshot_share(ln.tab.ix, ix);  // the masks the index holds right now
            run_fetch(slot, ln.tab);
        }
        lk.lock();
    }
    lk.unlock();
    while (!ln.tab.cursors.empty()) drop_cursor(ln.tab, ln.tab.cursors.size() - 1);
}

void vs_shm_server::run() {
    ShmHeader* h = m.hdr();
    std::vector<uint32_t> ready;
    auto last_reap = std::chrono::steady_clock::now();
    while (!stop.load()) {
        const uint32_t seq = h->work_seq.load(std::memory_order_acquire);
        apply_puts();
        ready.clear();
        for (uint32_t i = 0; i < m.nslots; ++i)
            if (posted(i)) ready.push_back(i);
        if (ready.empty()) {
            futex_wait(&h->work_seq, seq, 50000);  // (50 ms: also the cadence of the dead-owner check below)
        } else {
            // gather: until max_wait_us have passed since the first posted request was seen, or max_batch are posted (every post
            // bumps work_seq and wakes this thread: look again and keep waiting for the rest of the window)
            // (only scans that can share a launch wait for company: a cursor request is served at once)
            bool any_search = false;
            for (uint32_t i : ready) any_search |= m.slot(i)->op == OP_SEARCH || (cfg.cursor_pool != 0 && open_cursors.load() > 1);  // (a hint for how long to gather; nothing is run on it — with scan pools the cursor requests share launches too)
            if (any_search && cfg.max_wait_us && ready.size() < cfg.max_batch) {
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(cfg.max_wait_us);
                for (;;) {
                    const uint32_t seq2 = h->work_seq.load(std::memory_order_acquire);
                    ready.clear();
                    for (uint32_t i = 0; i < m.nslots; ++i)
                        if (posted(i)) ready.push_back(i);
                    const auto now = std::chrono::steady_clock::now();
                    if (ready.size() >= cfg.max_batch || now >= deadline || stop.load()) break;
                    // scans streamed out of the pools move in step: once every open cursor's backend has posted its next request (and no
                    // new scan is waiting for company) the window has nobody left to wait for
                    if (cfg.cursor_pool) {
                        size_t nfetch = 0;
                        bool search_posted = false;
                        for (uint32_t i : ready) {
                            const uint32_t op = m.slot(i)->op;
                            nfetch += op == OP_FETCH;
                            search_posted |= op == OP_SEARCH;
                        }
                        if (!search_posted && nfetch >= open_cursors.load()) break;
                    }
                    const auto left = std::chrono::duration_cast<std::chrono::microseconds>(deadline - now).count();
                    futex_wait(&h->work_seq, seq2, (int)std::max<long long>(left, 1));
                }
            }
            // every posted slot is TAKEN here: READY -> RUNNING first (the client may no longer touch a RUNNING slot; one that does
            // anyway changes nothing the dispatcher will read), then its request is copied and the copy validated; a request
            // outside the segment's limits is failed, not run
            for (size_t a = 0; a < ready.size();) {
                SlotHead* sa = m.slot(ready[a]);
                uint32_t exp = S_READY;
                if (!sa->state.compare_exchange_strong(exp, S_RUNNING, std::memory_order_acq_rel)) {  // (reaped or withdrawn meanwhile)
                    ready.erase(ready.begin() + (long)a);
                    continue;
                }
                const char* why = take(ready[a]);
                if (!why) {
                    ++a;
                    continue;
                }
                sa->rc = VS_ERR_INVALID;
                snprintf(sa->err, sizeof(sa->err), "vs_shm: request rejected by the dispatcher: %s", why);
                sa->state.store(S_DONE, std::memory_order_release);
                futex_wake(&sa->state, 1);
                ready.erase(ready.begin() + (long)a);
            }
            std::vector<bool> taken(ready.size(), false);
            if (cfg.cursor_pool) {  // cursor requests of this round: served together out of the scan pools
                std::vector<uint32_t> fetch_slots;
                for (size_t a = 0; a < ready.size(); ++a)
                    if (reqs[ready[a]].op != OP_SEARCH) {
                        taken[a] = true;
                        fetch_slots.push_back(ready[a]);
                    }
                if (!fetch_slots.empty()) run_fetch_pooled(fetch_slots);
            }
            for (size_t a = 0; a < ready.size(); ++a) {  // cursor requests: one scan each, served one after the other
                const Req* ha = &reqs[ready[a]];
                if (ha->op == OP_SEARCH || taken[a]) continue;
                taken[a] = true;
                if (lanes.empty()) {
                    run_fetch(ready[a], main_tab);
                } else {  // the lane this scan lives on: (client pid, scan id) -> lane, the same for every request of the scan
                    const uint64_t key = ((uint64_t)(uint32_t)ha->owner_pid * 0x9E3779B97F4A7C15ull) ^ (ha->scan_id * 0xC2B2AE3D27D4EB4Full);
                    Lane& ln = *lanes[(size_t)((key >> 17) % lanes.size())];
                    {
                        std::lock_guard<std::mutex> g(ln.mu);
                        ln.q.push_back(ready[a]);
                    }
                    ln.cv.notify_one();
                }
            }
            for (size_t a = 0; a < ready.size(); ++a) {
                if (taken[a]) continue;
                const Req* ha = &reqs[ready[a]];
                std::vector<uint32_t> grp;
                for (size_t b = a; b < ready.size() && grp.size() < cfg.max_batch; ++b) {
                    const Req* hb = &reqs[ready[b]];
                    if (!taken[b] && hb->L == ha->L && hb->rescore == ha->rescore && hb->k == ha->k && hb->has_label_key == ha->has_label_key &&
                        hb->snapshot == ha->snapshot) {
                        taken[b] = true;
                        grp.push_back(ready[b]);
                    }
                }
                run_group(grp);
            }
        }
        // a backend that died between CLAIMED and FREE would hold its slot forever
        const auto now = std::chrono::steady_clock::now();
        if (now - last_reap > std::chrono::milliseconds(200)) {
            last_reap = now;
            for (uint32_t i = 0; i < m.nslots; ++i) {
                SlotHead* s = m.slot(i);
                uint32_t st = s->state.load(std::memory_order_acquire);
                const int32_t pid = s->owner_pid;
                if ((st == S_CLAIMED || st == S_DONE) && pid > 0 && kill(pid, 0) != 0 && errno == ESRCH &&
                    s->state.compare_exchange_strong(st, S_REAPING, std::memory_order_acq_rel)) {
                    // (a claimer writes its pid only after its FREE -> CLAIMED swap: the dead owner's pid must be gone before the
                    // slot can be claimed again, or the next pass would free it under the new, live owner)
                    if (s->owner_pid == pid) s->owner_pid = 0;
                    s->state.store(S_FREE, std::memory_order_release);
                }
            }
            reap_cursors(main_tab);  // ... and its cursors their device memory (the lanes look after their own tables)
            reap_pools();
        }
    }
    while (!main_tab.cursors.empty()) drop_cursor(main_tab, main_tab.cursors.size() - 1);
    free_pools();
    // the lanes serve what they were handed, drop their cursors and end
    for (auto& l : lanes) {
        {
            std::lock_guard<std::mutex> g(l->mu);
            l->stop = true;
        }
        l->cv.notify_all();
        if (l->th.joinable()) l->th.join();
    }
    // shutting down: fail what is still posted so that no client sleeps forever
    for (uint32_t i = 0; i < m.nslots; ++i) {
        SlotHead* s = m.slot(i);
        uint32_t exp = S_READY;
        if (s->state.compare_exchange_strong(exp, S_RUNNING)) {
            s->rc = VS_ERR_STATE;
            snprintf(s->err, sizeof(s->err), "vs_shm: the dispatcher is shutting down");
            s->state.store(S_DONE, std::memory_order_release);
            futex_wake(&s->state, 1);
        }
    }
}

// the views and contexts of the lanes (their threads are not running: not started yet, or joined by run())
static void free_lanes(vs_shm_server* s) {
    for (auto& l : s->lanes) {
        if (l->tab.ix) vs_index_free(l->tab.ix);
        if (l->ctx) vs_ctx_destroy(l->ctx);
    }
    s->lanes.clear();
}

extern "C" {

int vs_shm_server_create(vs_index* idx, const char* name, uint32_t nslots, uint32_t kmax, const vs_broker_config* cfg,
                         vs_shm_server** out) {
    if (!idx || !name || !out || nslots == 0 || kmax == 0 || name[0] != '/') {
        vs_set_error("vs_shm_server_create: bad arguments (the name must start with '/')");
        return VS_ERR_INVALID;
    }
    *out = nullptr;
    vs_shm_server* s = new (std::nothrow) vs_shm_server();
    if (!s) {
        vs_set_error("vs_shm_server_create: out of memory");
        return VS_ERR_OOM;
    }
    s->ix = idx;
    s->name = name;
    int rc = vs_index_get_desc(idx, &s->d);
    if (rc != VS_OK) {
        delete s;
        return rc;
    }
    s->cfg.max_batch = cfg && cfg->max_batch ? cfg->max_batch : 8192;
    s->cfg.max_wait_us = cfg ? cfg->max_wait_us : 200;
    s->cfg.cursor_lanes = cfg ? std::min<uint32_t>(cfg->cursor_lanes, 64) : 0;
    s->cfg.cursor_pool = cfg ? std::min<uint32_t>(cfg->cursor_pool, 1024) : 0;
    if (!s->cfg.cursor_pool)
        if (const char* e = vs_opt_get("VS_SHM_CURSOR_POOL")) s->cfg.cursor_pool = std::min<uint32_t>((uint32_t)strtoul(e, nullptr, 10), 1024);
    if (!s->cfg.cursor_lanes)
        if (const char* e = vs_opt_get("VS_BROKER_LANES")) s->cfg.cursor_lanes = std::min<uint32_t>((uint32_t)strtoul(e, nullptr, 10), 64);
    s->main_tab.ix = idx;
    auto drop_lanes = [s] { free_lanes(s); };  // (no lane thread has been started yet)
    for (uint32_t i = 0; i < s->cfg.cursor_lanes; ++i) {
        std::unique_ptr<vs_shm_server::Lane> ln(new (std::nothrow) vs_shm_server::Lane());
        rc = ln ? vs_ctx_create_staging(vs_index_device(idx), (size_t)1 << 20, &ln->ctx) : VS_ERR_OOM;  // (a lane moves one query in and a few rows out)
        if (rc == VS_OK) rc = vs_index_view(idx, ln->ctx, &ln->tab.ix);
        if (ln) s->lanes.push_back(std::move(ln));
        if (rc != VS_OK) {
            drop_lanes();
            delete s;
            return rc;
        }
    }
    const size_t sb = slot_size(s->d.dim_full, kmax);
    const size_t bytes = align16(sizeof(ShmHeader)) + sb * nslots;
    shm_unlink(name);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
        vs_set_error("vs_shm_server_create: shm_open/ftruncate(%s, %zu): %s", name, bytes, strerror(errno));
        if (fd >= 0) close(fd);
        free_lanes(s);
        delete s;
        return VS_ERR_OOM;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        vs_set_error("vs_shm_server_create: mmap: %s", strerror(errno));
        shm_unlink(name);
        free_lanes(s);
        delete s;
        return VS_ERR_OOM;
    }
    memset(p, 0, bytes);
    s->m.base = p;
    s->m.bytes = bytes;
    ShmHeader* h = s->m.hdr();
    h->version = SHM_VERSION;
    h->nslots = nslots;
    h->dim_full = s->d.dim_full;
    h->kmax = kmax;
    h->slot_bytes = (uint32_t)sb;
    h->server_pid = (int32_t)getpid();
    h->serving.store(1);
    std::atomic_thread_fence(std::memory_order_release);
    h->magic = SHM_MAGIC;  // last: a client that sees the magic sees a complete header
    s->m.pin();
    s->reqs.resize(nslots);
    s->in_flight.reset(new std::atomic<uint8_t>[nslots]);
    for (uint32_t i = 0; i < nslots; ++i) s->in_flight[i].store(0);
    for (auto& l : s->lanes) {
        vs_shm_server::Lane* lp = l.get();
        lp->th = std::thread([s, lp] { s->run_lane(*lp); });
    }
    s->dispatcher = std::thread([s] { s->run(); });
    *out = s;
    return VS_OK;
}

int vs_shm_server_get_stats(vs_shm_server* s, vs_broker_stats* out) {
    if (!s || !out) {
        vs_set_error("vs_shm_server_get_stats: null argument");
        return VS_ERR_INVALID;
    }
    out->batches = s->batches.load();
    out->scans = s->scans.load();
    out->max_batch = s->max_batch.load();
    out->tasks = s->fetches.load();
    out->cursors = s->open_cursors.load();
    return VS_OK;
}

// (read by the caller's thread while the dispatcher works: the counters are atomics, the pool list is only sized)
int vs_shm_server_pool_stats(vs_shm_server* s, uint64_t out[4]) {
    if (!s || !out) {
        vs_set_error("vs_shm_server_pool_stats: null argument");
        return VS_ERR_INVALID;
    }
    out[0] = s->pools_alive.load();
    out[1] = s->pool_rounds.load();
    out[2] = s->pools_retired.load();
    out[3] = s->pooled_scans.load();
    return VS_OK;
}

void vs_shm_server_destroy(vs_shm_server* s) {
    if (!s) return;
    s->m.hdr()->serving.store(0);
    s->stop.store(true);
    s->m.hdr()->work_seq.fetch_add(1);
    futex_wake(&s->m.hdr()->work_seq, INT_MAX);
    if (s->dispatcher.joinable()) s->dispatcher.join();  // (run() joins the lanes before it returns)
    free_lanes(s);
    // callers still inside vs_shm_server_snapshot_put see `stop`, give up and leave before the object goes away
    while (s->put_callers.load(std::memory_order_acquire) > 0) {
        {
            std::lock_guard<std::mutex> g(s->put_mu);
        }
        s->put_cv.notify_all();
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    munmap(s->m.base, s->m.bytes);
    shm_unlink(s->name.c_str());
    delete s;
}

int vs_shm_client_open(const char* name, vs_shm_client** out) {
    if (!name || !out) {
        vs_set_error("vs_shm_client_open: null argument");
        return VS_ERR_INVALID;
    }
    *out = nullptr;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) {
        vs_set_error("vs_shm_client_open: shm_open(%s): %s", name, strerror(errno));
        return VS_ERR_STATE;
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(ShmHeader)) {
        vs_set_error("vs_shm_client_open: %s is not a request segment", name);
        close(fd);
        return VS_ERR_INVALID;
    }
    void* p = mmap(nullptr, (size_t)sb.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        vs_set_error("vs_shm_client_open: mmap: %s", strerror(errno));
        return VS_ERR_OOM;
    }
    const ShmHeader* h = static_cast<const ShmHeader*>(p);
    if (h->magic != SHM_MAGIC || h->version != SHM_VERSION ||
        align16(sizeof(ShmHeader)) + (size_t)h->slot_bytes * h->nslots > (size_t)sb.st_size || h->slot_bytes != slot_size(h->dim_full, h->kmax)) {
        vs_set_error("vs_shm_client_open: %s has no valid header (magic %08x, version %u)", name, h->magic, h->version);
        munmap(p, (size_t)sb.st_size);
        return VS_ERR_INVALID;
    }
    vs_shm_client* c = new (std::nothrow) vs_shm_client();
    if (!c) {
        munmap(p, (size_t)sb.st_size);
        vs_set_error("vs_shm_client_open: out of memory");
        return VS_ERR_OOM;
    }
    c->m.base = p;
    c->m.bytes = (size_t)sb.st_size;
    c->m.pin();  // (validated above against the size of the mapping)
    *out = c;
    return VS_OK;
}

uint32_t vs_shm_client_dim(const vs_shm_client* c) { return c ? c->m.dim_full : 0; }

int vs_shm_server_snapshot_put(vs_shm_server* s, uint32_t snapshot, const uint8_t* visible) {
    if (!s || snapshot < 1 || snapshot >= VS_MAX_SNAPSHOTS) {
        vs_set_error("vs_shm_server_snapshot_put: snapshot id outside [1,%d]", VS_MAX_SNAPSHOTS - 1);
        return VS_ERR_INVALID;
    }
    struct Inside {
        std::atomic<int>& n;
        explicit Inside(std::atomic<int>& n_) : n(n_) { n.fetch_add(1, std::memory_order_acq_rel); }
        ~Inside() { n.fetch_sub(1, std::memory_order_acq_rel); }
    } inside(s->put_callers);
    std::shared_ptr<vs_shm_server::PendingPut> p;
    try {
        p = std::make_shared<vs_shm_server::PendingPut>();
        p->snapshot = snapshot;
        p->drop = visible == nullptr;
        if (visible) p->mask.assign(visible, visible + s->d.n);
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_shm_server_snapshot_put: out of host memory");
        return VS_ERR_OOM;
    }
    std::unique_lock<std::mutex> lk(s->put_mu);
    if (s->stop.load()) {
        vs_set_error("vs_shm_server_snapshot_put: the dispatcher is shutting down");
        return VS_ERR_STATE;
    }
    s->puts.push_back(p);
    s->m.hdr()->work_seq.fetch_add(1);
    futex_wake(&s->m.hdr()->work_seq, INT_MAX);
    s->put_cv.wait(lk, [&] { return p->rc <= 0 || s->stop.load(); });
    if (p->rc > 0) {  // the dispatcher stopped first (an entry it had already taken is its own: the shared_ptr keeps it alive)
        for (auto it = s->puts.begin(); it != s->puts.end(); ++it)
            if (it->get() == p.get()) {
                s->puts.erase(it);
                break;
            }
        vs_set_error("vs_shm_server_snapshot_put: the dispatcher is shutting down");
        return VS_ERR_STATE;
    }
    if (p->rc != VS_OK) vs_set_error("%s", p->err.c_str());
    return p->rc;
}

int vs_shm_client_search(vs_shm_client* c, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                         uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids,
                         float* out_dist) {
    return vs_shm_client_search_snapshot(c, query, labels, n_labels, has_label_key, search_list_size, rescore, k, 0, out_ids, out_tids,
                                         out_dist);
}

static int client_request(vs_shm_client* c, uint32_t op, uint64_t scan_id, uint32_t skip, const float* query, const int16_t* labels,
                          uint32_t n_labels, int has_label_key, uint32_t search_list_size, uint32_t rescore, uint32_t k,
                          uint32_t snapshot, uint32_t* out_ids, uint64_t* out_tids, float* out_dist, uint32_t* n_rows) {
    if (!c || (!out_ids && op != OP_CLOSE) || k == 0 || snapshot >= VS_MAX_SNAPSHOTS) {
        vs_set_error("vs_shm_client_search: bad arguments");
        return VS_ERR_INVALID;
    }
    ShmHeader* h = c->m.hdr();
    if (k > c->m.kmax) {
        vs_set_error("vs_shm_client_search: k = %u exceeds the segment's %u rows per scan", k, c->m.kmax);
        return VS_ERR_INVALID;
    }
    if (has_label_key && query && n_labels > SHM_MAX_LABELS) {
        vs_set_error("vs_shm_client_search: more than %u labels in one scan key", SHM_MAX_LABELS);
        return VS_ERR_INVALID;
    }
    // claim a slot (every backend holds at most one; the segment is sized for max_connections)
    SlotHead* s = nullptr;
    const uint32_t start = (uint32_t)getpid() % c->m.nslots;
    for (int spin = 0; !s; ++spin) {
        if (!h->serving.load(std::memory_order_acquire) || server_dead(h)) {
            vs_set_error("vs_shm_client_search: no dispatcher is attached to the segment");
            return VS_ERR_STATE;
        }
        for (uint32_t i = 0; i < c->m.nslots && !s; ++i) {
            SlotHead* cand = c->m.slot((start + i) % c->m.nslots);
            uint32_t exp = S_FREE;
            if (cand->state.compare_exchange_strong(exp, S_CLAIMED, std::memory_order_acq_rel)) s = cand;
        }
        if (!s) usleep(spin < 10 ? 50 : 1000);  // every slot busy: more scans in flight than slots
    }
    s->owner_pid = (int32_t)getpid();
    s->op = op;
    s->scan_id = scan_id;
    s->skip = skip;
    s->n_rows = 0;
    s->L = search_list_size;
    s->rescore = rescore;
    s->k = k;
    s->snapshot = snapshot;
    s->null_query = query ? 0u : 1u;
    // a NULL query ignores its keys (amrescan: LabeledVector::from_scan_key_data with a NULL vector)
    s->has_label_key = (has_label_key && query) ? 1u : 0u;
    s->n_labels = s->has_label_key ? n_labels : 0u;
    if (s->n_labels) memcpy(s->labels, labels, (size_t)s->n_labels * 2);
    if (query) memcpy(Mapping::query(s), query, (size_t)c->m.dim_full * 4);
    s->rc = VS_OK;
    s->state.store(S_READY, std::memory_order_release);
    h->work_seq.fetch_add(1, std::memory_order_acq_rel);
    futex_wake(&h->work_seq, 1);
    for (;;) {
        const uint32_t st = s->state.load(std::memory_order_acquire);
        if (st == S_DONE) break;
        const bool gone = !h->serving.load(std::memory_order_acquire) || server_dead(h);
        if (gone && (st == S_READY || (st == S_RUNNING && server_dead(h)))) {
            // the dispatcher went away before taking the request, or died with it in hand: nobody will ever complete the slot
            uint32_t exp = st;
            if (s->state.compare_exchange_strong(exp, S_CLAIMED)) {
                s->owner_pid = 0;
                s->state.store(S_FREE, std::memory_order_release);
                vs_set_error("vs_shm_client_search: the dispatcher went away");
                return VS_ERR_STATE;
            }
            continue;
        }
        futex_wait(&s->state, st, 100000);  // (the timeout is the cadence of the liveness check above)
    }
    const int rc = s->rc;
    if (rc == VS_OK && op != OP_CLOSE) {
        const uint32_t rows = op == OP_FETCH ? std::min(s->n_rows, k) : k;
        memcpy(out_ids, c->m.ids(s), (size_t)rows * 4);
        if (out_tids) memcpy(out_tids, c->m.tids(s), (size_t)rows * 8);
        if (out_dist) memcpy(out_dist, c->m.dist(s), (size_t)rows * 4);
        if (n_rows) *n_rows = rows;
    } else if (rc != VS_OK) {
        vs_set_error("%s", s->err);
    }
    s->owner_pid = 0;
    s->state.store(S_FREE, std::memory_order_release);
    return rc;
}

int vs_shm_client_search_snapshot(vs_shm_client* c, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                                  uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t snapshot, uint32_t* out_ids,
                                  uint64_t* out_tids, float* out_dist) {
    return client_request(c, OP_SEARCH, 0, 0, query, labels, n_labels, has_label_key, search_list_size, rescore, k, snapshot, out_ids,
                          out_tids, out_dist, nullptr);
}

int vs_shm_client_fetch(vs_shm_client* c, uint64_t scan_id, const float* query, const int16_t* labels, uint32_t n_labels,
                        int has_label_key, uint32_t search_list_size, uint32_t rescore, uint32_t snapshot, uint32_t skip, uint32_t k,
                        uint32_t* out_ids, uint64_t* out_tids, float* out_dist, uint32_t* n_rows) {
    if (!n_rows) {
        vs_set_error("vs_shm_client_fetch: n_rows is NULL");
        return VS_ERR_INVALID;
    }
    *n_rows = 0;
    return client_request(c, OP_FETCH, scan_id, skip, query, labels, n_labels, has_label_key, search_list_size, rescore, k, snapshot,
                          out_ids, out_tids, out_dist, n_rows);
}

int vs_shm_client_end_scan(vs_shm_client* c, uint64_t scan_id) {
    return client_request(c, OP_CLOSE, scan_id, 0, nullptr, nullptr, 0, 0, 1, 0, 1, 0, nullptr, nullptr, nullptr, nullptr);
}

void vs_shm_client_close(vs_shm_client* c) {
    if (!c) return;
    munmap(c->m.base, c->m.bytes);
    delete c;
}

}  // extern "C"
