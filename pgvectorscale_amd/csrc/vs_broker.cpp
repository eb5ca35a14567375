// vs_broker.cpp — coalescing many concurrent scans into batched launches (host side).
//
// PostgreSQL runs one single-threaded backend per connection and the reference serves one query per backend
// (amcanparallel = false, AM/mod.rs:63; one TSVScanState per IndexScanDesc, AM/scan.rs:308-333).  A GPU does nothing
// useful with one scan at a time — a launch of the search kernel is efficient from a few thousand scans up — and a HIP
// context per backend is prohibitive.  The broker is the piece in between: ONE dispatcher thread owns the vs_index (and
// with it the vs_ctx: the "one thread per ctx" rule of vsgpu.h holds by construction); any number of client threads call
// vs_broker_search(), which enqueues the scan and blocks; the dispatcher gathers what arrived within a short window
// (max_wait_us after the oldest waiting request, or max_batch requests), runs every group of scans that share the same
// GUCs (search_list_size, rescore, k, label key present or not) as one vs_search_batch(), and hands each client its
// rows.  Results are those of vs_search_batch, i.e. exactly the rows of the first k amgettuple calls.
//
// In a PGRX deployment the client side of this queue lives in shared memory (one slot per backend, a latch per slot)
// and the dispatcher is a background worker; the queueing / grouping / batching logic is the same and is what this
// file implements and tests (threads stand in for backends).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vsgpu.h"

void vs_set_error(const char* fmt, ...);
const char* vs_opt_get(const char* name);  // vs_options.cpp: the option table (vs_set_option, snapshot of the VS_* environment)

namespace {

struct Request {
    const float* query;  // dim_full floats or nullptr (SQL NULL query)
    std::vector<int16_t> labels;
    bool has_label_key;
    uint32_t L, rescore, k;
    uint32_t snapshot = 0;             // visibility mask the scan runs under (0 = every tuple visible)
    bool is_put = false;               // control request: install `put_mask` (n bytes or nullptr) as mask `snapshot`
    const uint8_t* put_mask = nullptr;
    int (*task)(void*) = nullptr;      // control request: run task(task_arg) on the dispatcher thread (scan cursors: vs_broker_call)
    int (*task_via)(void*, vs_index*) = nullptr;  // ... or task_via(task_arg, <the handle of the thread that runs it>) (vs_broker_call_lane)
    void* task_arg = nullptr;
    uint32_t* out_ids;
    uint64_t* out_tids;
    float* out_dist;
    std::chrono::steady_clock::time_point t_arrive;
    // completion
    bool done = false;
    int rc = VS_OK;
    std::string err;
    std::condition_variable cv;

    bool same_group(const Request& o) const {
        // (scans of different snapshots never share a launch: a launch runs under ONE visibility mask)
        return !is_put && !o.is_put && !task && !o.task && !task_via && !o.task_via && L == o.L && rescore == o.rescore && k == o.k && has_label_key == o.has_label_key &&
               snapshot == o.snapshot;
    }
};

// One cursor lane (vs_broker_config.cursor_lanes): a thread with a context (HIP stream) and a view of the index of its own.  The
// continuations of the scans assigned to it run here, one after the other; different lanes run concurrently with each other and
// with the dispatcher's shared launches.
struct Lane {
    vs_ctx* ctx = nullptr;
    vs_index* view = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Request*> q;
    bool stop = false;
};

}  // namespace

struct vs_broker {
    vs_index* ix = nullptr;
    vs_index_desc d{};
    vs_broker_config cfg{};
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<Request*> queue;
    uint32_t ntasks = 0;  // single-scan tasks in the queue (always at its front: they are served before the gathering goes on)
    bool stop = false;
    std::thread dispatcher;
    vs_broker_stats st{};
    std::vector<std::unique_ptr<Lane>> lanes;
    std::atomic<uint32_t> next_lane{0};
    // a lane works under the snapshot masks of the broker's index (shared for the length of a task); vs_broker_snapshot_put
    // replaces one exclusively, i.e. when no lane is inside a task that could be reading it
    std::shared_mutex snap_mu;
    std::atomic<int> put_waiting{0};  // lanes do not start a task while a mask waits to be replaced (readers would starve the writer)

    void run_lane(Lane& ln);

    void run();
    void run_group(std::vector<Request*>& grp);
    void run_group_launch(std::vector<Request*>& grp, std::vector<float>& q, std::vector<int16_t>& lab, std::vector<uint32_t>& off,
                          std::vector<uint32_t>& ids, std::vector<uint64_t>& tids, std::vector<float>& dist, int& rc, std::string& err);
};

void vs_broker::run_group(std::vector<Request*>& grp) {
    const uint32_t nq = (uint32_t)grp.size();
    const Request& head = *grp[0];
    const uint32_t k = head.k;
    int rc = VS_OK;
    std::string err;
    std::vector<float> q, dist;
    std::vector<int16_t> lab;
    std::vector<uint32_t> off, ids;
    std::vector<uint64_t> tids;
    try {  // (this is the dispatcher thread: an exception here would terminate the process with every client still blocked)
        run_group_launch(grp, q, lab, off, ids, tids, dist, rc, err);
    } catch (const std::bad_alloc&) {
        rc = VS_ERR_OOM;
        err = "vs_broker: out of host memory while assembling a batch";
    }
    std::lock_guard<std::mutex> lk(mu);
    st.batches++;
    st.scans += nq;
    st.max_batch = std::max<uint64_t>(st.max_batch, nq);
    for (uint32_t i = 0; i < nq; ++i) {
        Request* r = grp[i];
        if (rc == VS_OK) {
            memcpy(r->out_ids, &ids[(size_t)i * k], (size_t)k * 4);
            if (r->out_tids) memcpy(r->out_tids, &tids[(size_t)i * k], (size_t)k * 8);
            if (r->out_dist) memcpy(r->out_dist, &dist[(size_t)i * k], (size_t)k * 4);
        }
        r->rc = rc;
        r->err = err;
        r->done = true;
        r->cv.notify_one();
    }
}

void vs_broker::run_group_launch(std::vector<Request*>& grp, std::vector<float>& q, std::vector<int16_t>& lab,
                                 std::vector<uint32_t>& off, std::vector<uint32_t>& ids, std::vector<uint64_t>& tids,
                                 std::vector<float>& dist, int& rc, std::string& err) {
    const uint32_t nq = (uint32_t)grp.size();
    const Request& head = *grp[0];
    const uint32_t k = head.k;
    q.assign((size_t)nq * d.dim_full, 0.0f);  // a NULL query is the zero vector (AM/labels/mod.rs:214-216)
    off.assign(nq + 1, 0);
    for (uint32_t i = 0; i < nq; ++i) {
        if (grp[i]->query) memcpy(&q[(size_t)i * d.dim_full], grp[i]->query, (size_t)d.dim_full * 4);
        // (label keys are ignored for a NULL query, as amrescan does)
        if (head.has_label_key && grp[i]->query) lab.insert(lab.end(), grp[i]->labels.begin(), grp[i]->labels.end());
        off[i + 1] = (uint32_t)lab.size();
    }
    ids.assign((size_t)nq * k, 0);
    tids.assign((size_t)nq * k, 0);
    dist.assign((size_t)nq * k, 0.0f);
    // scans with a label key and NULL-query scans (no key) cannot share a launch: the caller keeps them in separate groups
    const bool keys = head.has_label_key;
    vs_stats stats{};
    // the group's snapshot mask for the duration of the launch (the index-level mask of direct callers is put back)
    const uint8_t* prev = nullptr;
    rc = vs_index_snapshot_use(ix, head.snapshot, &prev);
    if (rc == VS_OK) {
        rc = vs_search_batch(ix, q.data(), keys ? lab.data() : nullptr, keys ? off.data() : nullptr, nq, head.L, head.rescore, k,
                             ids.data(), tids.data(), dist.data(), &stats);
        if (rc != VS_OK) err = vs_last_error();
        (void)vs_index_set_visibility_dev(ix, prev);
    } else {
        err = vs_last_error();
    }
}

void vs_broker::run() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) {
            if (stop) return;
            continue;
        }
        // gather: until the oldest request has waited max_wait_us or max_batch requests are queued
        const auto deadline = queue.front()->t_arrive + std::chrono::microseconds(cfg.max_wait_us);
        // (a task that arrives meanwhile is at the front of the queue: it is run at once and the gathering resumes — the oldest
        // scan's deadline is unchanged)
        while (!stop && !ntasks && queue.size() < cfg.max_batch && std::chrono::steady_clock::now() < deadline)
            cv_work.wait_until(lk, deadline);
        // one group = the scans that share the oldest request's GUCs (a NULL query never carries a label key)
        std::vector<Request*> grp;
        Request* head = queue.front();
        if (head->is_put || head->task || head->task_via) {  // a snapshot mask to install, or a piece of work on one scan's cursor: done here, on
                                            // the only thread that touches the index
            queue.pop_front();
            lk.unlock();
            int prc;
            std::string perr;
            try {
                if (head->task_via) {
                    prc = head->task_via(head->task_arg, ix);
                } else if (head->task) {
                    prc = head->task(head->task_arg);
                } else {
                    struct Waiting {
                        std::atomic<int>& n;
                        explicit Waiting(std::atomic<int>& n_) : n(n_) { n.fetch_add(1, std::memory_order_acq_rel); }
                        ~Waiting() { n.fetch_sub(1, std::memory_order_acq_rel); }
                    } waiting(put_waiting);
                    std::unique_lock<std::shared_mutex> xl(snap_mu);  // (no lane is inside a task)
                    prc = vs_index_snapshot_put(ix, head->snapshot, head->put_mask);
                }
                if (prc != VS_OK) perr = vs_last_error();
            } catch (const std::bad_alloc&) {
                prc = VS_ERR_OOM;
                perr = "vs_broker: out of host memory in a dispatcher task";
            }
            lk.lock();
            if (head->task || head->task_via) {
                st.tasks++;
                ntasks--;
            }
            head->rc = prc;
            head->err = perr;
            head->done = true;
            head->cv.notify_one();
            continue;
        }
        for (auto it = queue.begin(); it != queue.end() && grp.size() < cfg.max_batch;) {
            if ((*it)->is_put) break;  // scans posted after a mask change run after it
            if ((*it)->same_group(*head)) {
                grp.push_back(*it);
                it = queue.erase(it);
            } else {
                ++it;
            }
        }
        lk.unlock();
        run_group(grp);
        lk.lock();
    }
}

void vs_broker::run_lane(Lane& ln) {
    std::unique_lock<std::mutex> lk(ln.mu);
    for (;;) {
        ln.cv.wait(lk, [&] { return ln.stop || !ln.q.empty(); });
        if (ln.q.empty()) return;  // (stop: what was queued has been served)
        Request* r = ln.q.front();
        ln.q.pop_front();
        lk.unlock();
        int rc;
        std::string err;
        try {
            while (put_waiting.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
            std::shared_lock<std::shared_mutex> sl(snap_mu);
            rc = vs_index_snapshot_share(ln.view, ix);
            if (rc == VS_OK) rc = r->task_via(r->task_arg, ln.view);
            if (rc != VS_OK) err = vs_last_error();
        } catch (const std::bad_alloc&) {
            rc = VS_ERR_OOM;
            err = "vs_broker: out of host memory in a cursor lane";
        }
        {
            std::lock_guard<std::mutex> g(mu);  // (the request's completion flag and the statistics live under the broker's lock)
            st.tasks++;
            r->rc = rc;
            r->err = err;
            r->done = true;
            r->cv.notify_one();
        }
        lk.lock();
    }
}

extern "C" {

int vs_broker_create(vs_index* idx, const vs_broker_config* cfg, vs_broker** out) {
    if (!idx || !out) {
        vs_set_error("vs_broker_create: null argument");
        return VS_ERR_INVALID;
    }
    *out = nullptr;
    vs_broker* b = new (std::nothrow) vs_broker();
    if (!b) {
        vs_set_error("vs_broker_create: out of memory");
        return VS_ERR_OOM;
    }
    b->ix = idx;
    int rc = vs_index_get_desc(idx, &b->d);
    if (rc != VS_OK) {
        delete b;
        return rc;
    }
    b->cfg.max_batch = cfg && cfg->max_batch ? cfg->max_batch : 8192;
    b->cfg.max_wait_us = cfg ? cfg->max_wait_us : 200;
    b->cfg.cursor_lanes = cfg ? std::min<uint32_t>(cfg->cursor_lanes, 64) : 0;
    if (!b->cfg.cursor_lanes)  // (VS_BROKER_LANES: the default for brokers created without a lane count — how the whole broker
        if (const char* e = vs_opt_get("VS_BROKER_LANES")) b->cfg.cursor_lanes = std::min<uint32_t>((uint32_t)strtoul(e, nullptr, 10), 64);  // test tier runs on lanes)
    for (uint32_t i = 0; i < b->cfg.cursor_lanes; ++i) {
        std::unique_ptr<Lane> ln(new (std::nothrow) Lane());
        rc = ln ? vs_ctx_create_staging(vs_index_device(idx), (size_t)1 << 20, &ln->ctx) : VS_ERR_OOM;  // (a lane moves one query in and a few rows out)
        if (rc == VS_OK) rc = vs_index_view(idx, ln->ctx, &ln->view);
        if (rc != VS_OK) {  // (what was created so far goes away again; no thread has been started yet)
            if (ln && ln->ctx) vs_ctx_destroy(ln->ctx);
            for (auto& l : b->lanes) {
                vs_index_free(l->view);
                vs_ctx_destroy(l->ctx);
            }
            delete b;
            return rc;
        }
        b->lanes.push_back(std::move(ln));
    }
    for (auto& l : b->lanes) {
        Lane* lp = l.get();
        lp->th = std::thread([b, lp] { b->run_lane(*lp); });
    }
    b->dispatcher = std::thread([b] { b->run(); });
    *out = b;
    return VS_OK;
}

int vs_broker_search_snapshot(vs_broker* b, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                              uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t snapshot, uint32_t* out_ids,
                              uint64_t* out_tids, float* out_dist) {
    if (!b || !out_ids || k == 0 || snapshot >= VS_MAX_SNAPSHOTS) {
        vs_set_error("vs_broker_search: bad arguments");
        return VS_ERR_INVALID;
    }
    Request r;
    r.query = query;
    // a NULL query ignores its keys (amrescan: LabeledVector::from_scan_key_data with a NULL vector)
    r.has_label_key = has_label_key != 0 && query != nullptr;
    if (r.has_label_key && labels) r.labels.assign(labels, labels + n_labels);
    r.L = search_list_size;
    r.rescore = rescore;
    r.k = k;
    r.snapshot = snapshot;
    r.out_ids = out_ids;
    r.out_tids = out_tids;
    r.out_dist = out_dist;
    r.t_arrive = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) {
        vs_set_error("vs_broker_search: the broker is shutting down");
        return VS_ERR_STATE;
    }
    b->queue.push_back(&r);
    b->cv_work.notify_one();
    r.cv.wait(lk, [&] { return r.done; });
    lk.unlock();
    if (r.rc != VS_OK) vs_set_error("%s", r.err.c_str());
    return r.rc;
}

int vs_broker_search(vs_broker* b, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                     uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t* out_ids, uint64_t* out_tids, float* out_dist) {
    return vs_broker_search_snapshot(b, query, labels, n_labels, has_label_key, search_list_size, rescore, k, 0, out_ids, out_tids,
                                     out_dist);
}

int vs_broker_snapshot_put(vs_broker* b, uint32_t snapshot, const uint8_t* visible) {
    if (!b || snapshot < 1 || snapshot >= VS_MAX_SNAPSHOTS) {
        vs_set_error("vs_broker_snapshot_put: snapshot id outside [1,%d]", VS_MAX_SNAPSHOTS - 1);
        return VS_ERR_INVALID;
    }
    Request r;
    r.query = nullptr;
    r.has_label_key = false;
    r.L = r.rescore = r.k = 0;
    r.out_ids = nullptr;
    r.out_tids = nullptr;
    r.out_dist = nullptr;
    r.snapshot = snapshot;
    r.is_put = true;
    r.put_mask = visible;
    // (no waiting for company: the gather window of run() ends at once for a request that arrived max_wait_us ago)
    r.t_arrive = std::chrono::steady_clock::now() - std::chrono::hours(1);
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) {
        vs_set_error("vs_broker_snapshot_put: the broker is shutting down");
        return VS_ERR_STATE;
    }
    b->queue.push_back(&r);  // behind the scans already queued: they run under the mask they were posted with
    b->cv_work.notify_one();
    r.cv.wait(lk, [&] { return r.done; });
    lk.unlock();
    if (r.rc != VS_OK) vs_set_error("%s", r.err.c_str());
    return r.rc;
}

// Runs fn(arg) on the dispatcher thread, between two launches, and returns its result: how work that belongs to ONE scan (the
// continuation of its cursor on the device, the release of its device buffers) reaches the only thread allowed to touch the index.
// A non-zero result carries the dispatcher thread's vs_last_error() text over to the caller's.
int vs_broker_call(vs_broker* b, int (*fn)(void*), void* arg) {
    if (!b || !fn) {
        vs_set_error("vs_broker_call: null argument");
        return VS_ERR_INVALID;
    }
    if (std::this_thread::get_id() == b->dispatcher.get_id()) return fn(arg);  // (already there)
    Request r;
    r.query = nullptr;
    r.has_label_key = false;
    r.L = r.rescore = r.k = 0;
    r.out_ids = nullptr;
    r.out_tids = nullptr;
    r.out_dist = nullptr;
    r.task = fn;
    r.task_arg = arg;
    r.t_arrive = std::chrono::steady_clock::now() - std::chrono::hours(1);  // (no waiting for company)
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) {
        vs_set_error("vs_broker_call: the broker is shutting down");
        return VS_ERR_STATE;
    }
    // ahead of queued scans that are still gathering company: a continuation is one short launch and its backend is waiting
    b->queue.push_front(&r);
    b->ntasks++;
    b->cv_work.notify_one();
    r.cv.wait(lk, [&] { return r.done; });
    lk.unlock();
    if (r.rc != VS_OK) vs_set_error("%s", r.err.c_str());
    return r.rc;
}

// fn(arg, handle) on the lane `lane_key` names (its thread, its view of the index), or — a broker without lanes — on the dispatcher
// thread with the broker's own index; blocks until it is done.  Work of ONE scan: the scan stays on the lane it was assigned.
int vs_broker_call_lane(vs_broker* b, uint32_t lane_key, int (*fn)(void*, vs_index*), void* arg) {
    if (!b || !fn) {
        vs_set_error("vs_broker_call_lane: null argument");
        return VS_ERR_INVALID;
    }
    Request r;
    r.query = nullptr;
    r.has_label_key = false;
    r.L = r.rescore = r.k = 0;
    r.out_ids = nullptr;
    r.out_tids = nullptr;
    r.out_dist = nullptr;
    r.task_via = fn;
    r.task_arg = arg;
    r.t_arrive = std::chrono::steady_clock::now() - std::chrono::hours(1);  // (no waiting for company)
    if (b->lanes.empty()) {
        if (std::this_thread::get_id() == b->dispatcher.get_id()) return fn(arg, b->ix);  // (already there)
        std::unique_lock<std::mutex> lk(b->mu);
        if (b->stop) {
            vs_set_error("vs_broker_call_lane: the broker is shutting down");
            return VS_ERR_STATE;
        }
        b->queue.push_front(&r);  // ahead of queued scans that are still gathering company
        b->ntasks++;
        b->cv_work.notify_one();
        r.cv.wait(lk, [&] { return r.done; });
    } else {
        Lane& ln = *b->lanes[lane_key % b->lanes.size()];
        {
            std::lock_guard<std::mutex> lk(b->mu);
            if (b->stop) {
                vs_set_error("vs_broker_call_lane: the broker is shutting down");
                return VS_ERR_STATE;
            }
        }
        {
            std::lock_guard<std::mutex> g(ln.mu);
            if (ln.stop) {
                vs_set_error("vs_broker_call_lane: the broker is shutting down");
                return VS_ERR_STATE;
            }
            ln.q.push_back(&r);
        }
        ln.cv.notify_one();
        std::unique_lock<std::mutex> lk(b->mu);
        r.cv.wait(lk, [&] { return r.done; });
    }
    if (r.rc != VS_OK) vs_set_error("%s", r.err.c_str());
    return r.rc;
}

// round robin over the lanes (0 without lanes)
uint32_t vs_broker_assign_lane(vs_broker* b) { return b && !b->lanes.empty() ? b->next_lane.fetch_add(1) % (uint32_t)b->lanes.size() : 0u; }

vs_index* vs_broker_index(vs_broker* b) { return b ? b->ix : nullptr; }

int vs_broker_get_stats(vs_broker* b, vs_broker_stats* out) {
    if (!b || !out) {
        vs_set_error("vs_broker_get_stats: null argument");
        return VS_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(b->mu);
    *out = b->st;
    return VS_OK;
}

void vs_broker_destroy(vs_broker* b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    if (b->dispatcher.joinable()) b->dispatcher.join();  // drains what is queued first (run() only returns on an empty queue)
    for (auto& l : b->lanes) {  // the lanes serve what they were handed, then end; their views go before the index does
        {
            std::lock_guard<std::mutex> g(l->mu);
            l->stop = true;
        }
        l->cv.notify_all();
        if (l->th.joinable()) l->th.join();
        vs_index_free(l->view);
        vs_ctx_destroy(l->ctx);
    }
    delete b;
}

}  // extern "C"
