// vs_cursor.hip — host side of libvsgpu.so: the amrescan / amgettuple mirror (vs_beginscan / vs_rescan / vs_gettuple / vs_endscan).
// Split out of vs_api.hip in round 6 (code motion only).
#include <thread>
#include <cstdarg>
#include <cmath>
#include <algorithm>
#include <cstdlib>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "vs_internal.h"


// ---------------------------------------------------------------------------------------------------------------
// amrescan / amgettuple mirror (AM/scan.rs:308-456).
//
// A scan on an index keeps what the reference keeps between amgettuple calls — the ListSearchResult (`lsr`) and the
// resort_buffer of TSVResponseIterator (AM/scan.rs:162-174) — on the device: the candidate heap, the dedup set and the visited
// list of ITS OWN resumable launch of the general kernel (k_search, state saved in `state`, spill regions heap_g / hash), the
// rows emitted so far (all_ids / all_ham / all_dist) and the BinaryHeap of the rescore window (resort_heap).  A call that runs
// out of prefetched rows CONTINUES the scan for a few more rows (vs_search.hip, SearchLaunch::resume) instead of running it
// again, reranks only the new rows and continues the window (k_resort_cursor).  The work counters are recorded per emitted
// row, so vs_scan_get_stats reports what the reference's GreedySearchStats hold after the same number of amgettuple calls,
// however far the prefetch has run ahead.  A scan whose structures outgrow their capacities is started again with larger ones
// and fast-forwarded (rare: capacities are sized for ~1000 rows beyond the list size).
//
// A scan on a broker fetches windows through vs_broker_search (shared launches); there a longer window re-runs the
// deterministic scan (four times larger each time), as before.
// ---------------------------------------------------------------------------------------------------------------
struct ScanCursor {
    bool open = false;        // state on the device belongs to the current rescan
    bool started = false;     // at least one launch ran (state blob initialised)
    bool exhausted = false;   // the stream has ended (a next() came back empty)
    uint32_t hl = 0, hcap = 0, vcap = 0, lh = 0, hashcap = 0, g0 = 0;
    uint32_t rows = 0;        // stream rows emitted so far (valid prefix of all_ids)
    uint32_t rows_cap = 0;    // capacity of all_ids / all_ham / all_dist (rows)
    uint32_t restarts = 0;    // times the scan was started again with larger capacities (since the rescan)
    uint32_t launches = 0;
    bool masked = false;      // the launches ran under a heap-visibility mask (rows hidden by it still cost a heap fetch)
    DevBuf raw_q, q_full, q_index, qcodes, qlabels, qlabel_off, heap_g, hash, state, cnt, stats, status, row_stats, all_ids, all_ham,
        all_dist, resort_heap, cur, out_ids, out_tids, out_dist;
    std::vector<uint32_t> row_stats_h;  // [rows][ST_N] counters at the emission of each row
    uint32_t final_counters[ST_N] = {0};  // the counters when the stream ended (incl. the next() that found nothing)
    // what the launches of this scan really did: the current run's counters + those of runs given up for a restart
    uint64_t run_visits = 0, run_dq = 0, run_cand = 0, run_reads = 0, lost_visits = 0, lost_dq = 0, lost_cand = 0, lost_reads = 0;
    void free_all() {
        for (DevBuf* b : {&raw_q, &q_full, &q_index, &qcodes, &qlabels, &qlabel_off, &heap_g, &hash, &state, &cnt, &stats, &status,
                          &row_stats, &all_ids, &all_ham, &all_dist, &resort_heap, &cur, &out_ids, &out_tids, &out_dist})
            devbuf_free(*b);
    }
};

struct vs_scan {
    vs_index* ix = nullptr;
    vs_broker* broker = nullptr;  // non-null: the first window comes from a shared launch, the rest from a cursor on the dispatcher thread
    uint32_t lane = 0;            // (broker scans) the cursor lane the scan's continuations run on
    uint32_t snapshot = 0;        // (broker scans) visibility mask the scan runs under
    uint32_t snapshot_next = 0;   // ... from the next vs_rescan on (vs_scan_set_snapshot)
    bool active = false;
    bool null_query = false;
    std::vector<float> query;
    std::vector<int16_t> labels;
    bool has_label_key = false;
    uint32_t L = 100, rescore = 50;
    uint32_t window = 0;                 // rows fetched so far
    uint32_t cursor = 0;                 // rows handed out
    uint32_t calls_after_end = 0;        // amgettuple calls that found the scan already exhausted
    bool exhausted = false;              // the fetched window reached the end of the scan
    std::vector<uint32_t> ids;
    std::vector<uint64_t> tids;
    std::vector<float> dist;
    vs_stats stats{};
    ScanCursor cur;
    ~vs_scan() { cur.free_all(); }
};
extern "C" int vs_broker_call(vs_broker* b, int (*fn)(void*), void* arg);
extern "C" int vs_broker_call_lane(vs_broker* b, uint32_t lane_key, int (*fn)(void*, vs_index*), void* arg);
extern "C" uint32_t vs_broker_assign_lane(vs_broker* b);

extern "C" int vs_beginscan(vs_index* ix, vs_scan** out) {
    VS_REQUIRE(ix && out, "vs_beginscan: bad args");
    vs_scan* s = new (std::nothrow) vs_scan();
    VS_REQUIRE_OOM(s, "vs_beginscan: out of host memory");
    s->ix = ix;
    *out = s;
    return VS_OK;
}

extern "C" int vs_beginscan_on_broker(vs_broker* b, vs_scan** out) {
    VS_REQUIRE(b && out, "vs_beginscan_on_broker: bad args");
    vs_scan* s = new (std::nothrow) vs_scan();
    VS_REQUIRE_OOM(s, "vs_beginscan_on_broker: out of host memory");
    s->ix = vs_broker_index(b);
    s->broker = b;
    s->lane = vs_broker_assign_lane(b);
    *out = s;
    return VS_OK;
}

// grow a device array of `elem`-byte rows to at least `rows` rows, keeping its first `keep` rows
static int devbuf_grow_keep(vs_ctx* c, DevBuf& b, size_t rows, size_t keep, size_t elem) {
    if (rows * elem <= b.bytes) return VS_OK;
    void* np = nullptr;
    const size_t want = rows * elem + 256;
    VS_HIP(hipMalloc(&np, want));
    if (b.p && keep) {
        hipError_t e = hipMemcpyAsync(np, b.p, keep * elem, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(np);
            VS_HIP(e);
        }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.bytes = want;
    return VS_OK;
}

static uint32_t effective_rescore(const vs_index* ix, uint32_t rescore) {
    // amgettuple, Plain arm: num_dimensions == num_dimensions_to_index => "no need to resort" (AM/scan.rs:392-399)
    return (ix->d.storage_type == VS_STORAGE_PLAIN && ix->d.dim_index == ix->d.dim_full) ? 0u : rescore;
}

// (re)initialises the device side of a scan: query preparation, label key, capacities, empty state
static int cursor_open(vs_scan* s, uint32_t min_rows) {
    vs_index* ix = s->ix;
    vs_ctx* c = ix->ctx;
    ScanCursor& k = s->cur;
    const bool keys = s->has_label_key && !s->null_query;
    if (ix->d.storage_type == VS_STORAGE_PLAIN) VS_REQUIRE(!keys, "Plain storage does not support label filters");  // AM/plain/storage.rs:262
    VS_HIP(hipSetDevice(c->device));
    // capacities: room for `horizon` rows beyond the list (a scan that goes further is restarted with four times the room)
    const uint64_t horizon = std::max<uint64_t>(env_u32("VS_CURSOR_HORIZON", 1024), 4ull * min_rows) << (2 * std::min<uint32_t>(k.restarts, 6));
    const uint64_t visits = 2ull * s->L + horizon + 32;
    const uint64_t pushes = visits * ix->d.num_neighbors;
    k.hl = env_u32("VS_HL", 1024);
    k.lh = 0;
    k.g0 = env_u32("VS_G0", 4096);
    k.hcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pushes, k.hl), 1u << 24);
    k.vcap = (uint32_t)std::min<uint64_t>((2ull * s->L + 256) << std::min<uint32_t>(k.restarts, 6), 1u << 20);
    k.hashcap = std::max<uint32_t>(next_pow2_u32(std::min<uint64_t>(2ull * pushes, 1u << 26)), k.g0);
    SearchLaunch probe{};
    probe.hl = k.hl;
    probe.lh = k.lh;
    probe.vcap = k.vcap;
    VS_TRY(devbuf_reserve(c, k.raw_q, (size_t)ix->d.dim_full * 4));
    VS_TRY(devbuf_reserve(c, k.q_full, (size_t)ix->vec_stride * 4));
    VS_TRY(devbuf_reserve(c, k.qcodes, (size_t)ix->code_stride * 8 + 16));
    VS_TRY(devbuf_reserve(c, k.heap_g, std::max<size_t>((size_t)(k.hcap > k.hl ? k.hcap - k.hl : 0) * 8, 16)));
    VS_TRY(devbuf_reserve(c, k.hash, (size_t)k.hashcap * 4));
    VS_TRY(devbuf_reserve(c, k.state, search_resume_words(probe) * 4));
    VS_TRY(devbuf_reserve(c, k.cnt, 16));
    VS_TRY(devbuf_reserve(c, k.stats, ST_N * 4));
    VS_TRY(devbuf_reserve(c, k.status, 16));
    VS_TRY(devbuf_reserve(c, k.cur, 16));
    VS_HIP(hipMemsetAsync(k.state.p, 0, RS_HDR * 4, c->stream));
    VS_HIP(hipMemsetAsync(k.cur.p, 0, 16, c->stream));
    VS_TRY(vs_dev_upload(c, k.raw_q.p, s->query.data(), (size_t)ix->d.dim_full * 4));
    VS_TRY(launch_prepare_queries(ix, (const float*)k.raw_q.p, 1, (float*)k.q_full.p, (uint64_t*)k.qcodes.p));
    if (ix->d.storage_type == VS_STORAGE_PLAIN && ix->d.dim_index < ix->d.dim_full) {
        VS_TRY(devbuf_reserve(c, k.q_index, (size_t)ix->vec_stride * 4));
        VS_TRY(launch_prepare_index_slice(ix, (const float*)k.raw_q.p, 1, (float*)k.q_index.p));
    }
    if (keys) {
        VS_REQUIRE(ix->d.has_labels && ix->label_off, "label scan keys on an index without labels");
        std::vector<int16_t> l(s->labels);  // LabelSet::from(Vec<Label>): sort_unstable + dedup (AM/labels/mod.rs:30-37)
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        const uint32_t off[2] = {0, (uint32_t)l.size()};
        VS_TRY(devbuf_reserve(c, k.qlabels, std::max<size_t>(l.size(), 1) * 2));
        VS_TRY(devbuf_reserve(c, k.qlabel_off, 8));
        if (!l.empty()) VS_TRY(vs_dev_upload(c, k.qlabels.p, l.data(), l.size() * 2));
        VS_TRY(vs_dev_upload(c, k.qlabel_off.p, off, 8));
    }
    k.open = true;
    k.started = false;
    k.exhausted = false;
    k.rows = 0;
    k.row_stats_h.clear();
    memset(k.final_counters, 0, sizeof(k.final_counters));
    k.run_visits = k.run_dq = k.run_cand = k.run_reads = 0;
    return VS_OK;
}

// continues the scan on the device until `want_rows` stream rows exist (or the stream ends)
static int cursor_extend(vs_scan* s, uint32_t want_rows) {
    vs_index* ix = s->ix;
    vs_ctx* c = ix->ctx;
    ScanCursor& k = s->cur;
    const uint32_t S = effective_rescore(ix, s->rescore);
    const bool keys = s->has_label_key && !s->null_query;
    const bool plain = ix->d.storage_type == VS_STORAGE_PLAIN;
    while (k.rows < want_rows && !k.exhausted) {
        const uint32_t M = want_rows - k.rows;
        if (k.rows + M > k.rows_cap) {
            const uint32_t ncap = std::max<uint32_t>(k.rows + M, std::max<uint32_t>(256, 2 * k.rows_cap));
            VS_TRY(devbuf_grow_keep(c, k.all_ids, ncap, k.rows, 4));
            VS_TRY(devbuf_grow_keep(c, k.all_ham, ncap, k.rows, 4));
            VS_TRY(devbuf_grow_keep(c, k.all_dist, ncap, k.rows, 4));
            k.rows_cap = ncap;
        }
        VS_TRY(devbuf_reserve(c, k.row_stats, (size_t)M * ST_N * 4));
        SearchLaunch sl;
        sl.nq = 1;
        sl.L = s->L;
        sl.M = M;
        sl.hl = k.hl;
        sl.hcap = k.hcap;
        sl.vcap = k.vcap;
        sl.lh = k.lh;
        sl.hashcap = k.hashcap;
        sl.g0 = k.g0;
        sl.qcodes = (const uint64_t*)k.qcodes.p;
        sl.qlabels = keys ? (const int16_t*)k.qlabels.p : nullptr;
        sl.qlabel_off = keys ? (const uint32_t*)k.qlabel_off.p : nullptr;
        sl.heap_g = (uint64_t*)k.heap_g.p;
        sl.hash = (uint32_t*)k.hash.p;
        sl.out_ids = (uint32_t*)k.all_ids.p + k.rows;
        sl.out_ham = (uint32_t*)k.all_ham.p + k.rows;
        sl.out_cnt = (uint32_t*)k.cnt.p;
        sl.stats = (uint32_t*)k.stats.p;
        sl.status = (uint32_t*)k.status.p;
        sl.visible = S > 0 ? ix->visible : nullptr;  // the heap is only fetched for the rescore window
        k.masked = sl.visible != nullptr;
        sl.resume = (uint32_t*)k.state.p;
        sl.resume_stride = 0;
        sl.row_stats = (uint32_t*)k.row_stats.p;
        // the plain-storage kernel reads its prepared query from the batch workspace slot: point it at this scan's
        void* const ws_q_full = ix->ws.q_full.p;
        void* const ws_q_index = ix->ws.q_index.p;
        if (plain) {
            ix->ws.q_full.p = k.q_full.p;
            ix->ws.q_index.p = k.q_index.p;
        }
        hipEvent_t ev = prof_begin(c);
        const int lr = launch_search(ix, sl);
        prof_end(c, PK_SEARCH, ev);
        if (plain) {
            ix->ws.q_full.p = ws_q_full;
            ix->ws.q_index.p = ws_q_index;
        }
        VS_TRY(lr);
        k.launches++;
        k.started = true;
        uint32_t hdr[RS_HDR];
        uint32_t cnt = 0;
        VS_HIP(hipMemcpyAsync(hdr, k.state.p, sizeof(hdr), hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipMemcpyAsync(&cnt, k.cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
        k.run_visits = hdr[RS_VISITS];
        k.run_dq = hdr[RS_DQ];
        k.run_cand = hdr[RS_CAND];
        k.run_reads = hdr[RS_READS];
        if (hdr[RS_STATUS] != 0) {
            // a structure outgrew its capacity: start again with more room; the caller fast-forwards (the rows already handed
            // out are reproduced by the deterministic scan and skipped)
            VS_REQUIRE(k.restarts < 8, "scan structures overflowed (flags 0x%x) at hcap=%u vcap=%u hashcap=%u", hdr[RS_STATUS],
                       k.hcap, k.vcap, k.hashcap);
            k.lost_visits += k.run_visits;
            k.lost_dq += k.run_dq;
            k.lost_cand += k.run_cand;
            k.lost_reads += k.run_reads;
            k.restarts++;
            VS_TRY(cursor_open(s, want_rows));
            continue;
        }
        if (cnt) {
            const size_t base = k.row_stats_h.size();
            k.row_stats_h.resize(base + (size_t)cnt * ST_N);
            VS_HIP(hipMemcpyAsync(k.row_stats_h.data() + base, k.row_stats.p, (size_t)cnt * ST_N * 4, hipMemcpyDeviceToHost, c->stream));
            if (S > 0) {  // get_full_distance_for_resort of the new rows only (AM/sbq/storage.rs:304-328)
                VS_REQUIRE(ix->vecs, "diskann.query_rescore > 0 needs the heap vector column on the device");
                hipEvent_t ev2 = prof_begin(c);
                VS_TRY(launch_rerank(ix, (const float*)k.q_full.p, (const uint32_t*)k.all_ids.p + k.rows, nullptr,
                                     (const uint32_t*)k.cnt.p, M, 1, (float*)k.all_dist.p + k.rows));
                prof_end(c, PK_RERANK, ev2);
            }
            VS_HIP(hipStreamSynchronize(c->stream));
        }
        k.rows += cnt;
        if (cnt < M) {
            k.exhausted = true;
            for (int i = 0; i < ST_N; ++i) k.final_counters[i] = 0;
            k.final_counters[ST_VISITS] = hdr[RS_VISITS];
            k.final_counters[ST_CAND] = hdr[RS_CAND];
            k.final_counters[ST_DQ] = hdr[RS_DQ];
            k.final_counters[ST_READS] = hdr[RS_READS];
            k.final_counters[ST_NEXT] = hdr[RS_NEXT];
            k.final_counters[ST_INVIS] = hdr[RS_INVIS];
        }
    }
    return VS_OK;
}

// makes rows [s->ids.size(), target) of the scan available in the host vectors (fewer when the scan ends first)
static int cursor_fetch(vs_scan* s, uint32_t target) {
    vs_index* ix = s->ix;
    vs_ctx* c = ix->ctx;
    ScanCursor& k = s->cur;
    const uint32_t S = effective_rescore(ix, s->rescore);
    if (!k.open) VS_TRY(cursor_open(s, target));
    const uint32_t need = S > 0 ? S + target - 1 : target;  // stream rows behind `target` amgettuple calls
    VS_TRY(cursor_extend(s, need));
    // (after a restart the window state on the device starts at row 0 again: the rows already handed out are reproduced)
    uint32_t curh[4] = {0, 0, 0, 0};
    VS_HIP(hipMemcpyAsync(curh, k.cur.p, 16, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    uint32_t have = curh[2];
    VS_REQUIRE(have <= s->ids.size(), "scan cursor out of step");
    if (S > 0) VS_TRY(devbuf_reserve(c, k.resort_heap, (size_t)S * 8));
    while (have < target) {
        const uint32_t kk = std::min<uint32_t>(target - have, 4096);
        VS_TRY(devbuf_reserve(c, k.out_ids, (size_t)kk * 4));
        VS_TRY(devbuf_reserve(c, k.out_tids, (size_t)kk * 8));
        VS_TRY(devbuf_reserve(c, k.out_dist, (size_t)kk * 4));
        hipEvent_t ev = prof_begin(c);
        VS_TRY(launch_resort_cursor(ix, k.rows, k.exhausted, S, kk, (const uint32_t*)k.all_ids.p, (const float*)k.all_dist.p,
                                    (const uint32_t*)k.all_ham.p, (uint64_t*)k.resort_heap.p, (uint32_t*)k.cur.p,
                                    (uint32_t*)k.out_ids.p, (uint64_t*)k.out_tids.p, (float*)k.out_dist.p));
        prof_end(c, PK_RESORT, ev);
        VS_HIP(hipMemcpyAsync(curh, k.cur.p, 16, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
        const uint32_t got = curh[3];
        if (got) {
            std::vector<uint32_t> ids(got);
            std::vector<uint64_t> tids(got);
            std::vector<float> dist(got);
            VS_HIP(hipMemcpy(ids.data(), k.out_ids.p, (size_t)got * 4, hipMemcpyDeviceToHost));
            VS_HIP(hipMemcpy(tids.data(), k.out_tids.p, (size_t)got * 8, hipMemcpyDeviceToHost));
            VS_HIP(hipMemcpy(dist.data(), k.out_dist.p, (size_t)got * 4, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < got; ++i) {
                if (have + i < s->ids.size()) continue;  // handed out before a restart
                s->ids.push_back(ids[i]);
                s->tids.push_back(tids[i]);
                s->dist.push_back(dist[i]);
            }
        }
        have += got;
        if (got < kk) break;  // the scan has ended (or, never: the window could not be filled)
    }
    s->window = (uint32_t)s->ids.size();
    s->exhausted = k.exhausted && s->window < target;
    return VS_OK;
}

static int scan_fetch(vs_scan* s, uint32_t window) {
    vs_index* ix = s->ix;
    s->ids.assign(window, VS_INVALID_NODE);
    s->tids.assign(window, 0);
    s->dist.assign(window, 0.f);
    const bool keys = s->has_label_key && !s->null_query;
    VS_TRY(vs_broker_search_snapshot(s->broker, s->null_query ? nullptr : s->query.data(), s->labels.data(), (uint32_t)s->labels.size(),
                                     keys ? 1 : 0, s->L, s->rescore, window, s->snapshot, s->ids.data(), s->tids.data(), s->dist.data()));
    s->stats = vs_stats{};  // the counters of a shared launch are not attributed to single scans
    (void)ix;
    s->window = window;
    s->exhausted = false;
    for (uint32_t i = 0; i < window; ++i)
        if (s->ids[i] == VS_INVALID_NODE) {
            s->exhausted = true;
            s->window = i;
            break;
        }
    s->ids.resize(s->window);  // (only rows of the scan: a cursor that takes over appends to them)
    s->tids.resize(s->window);
    s->dist.resize(s->window);
    return VS_OK;
}

// A scan on a broker continues on the dispatcher thread (the only one that may touch the index): its cursor is opened there, run
// under the scan's snapshot mask, and released there.  The rows a shared launch already produced are reproduced by the
// deterministic scan once and skipped (cursor_fetch), after that the scan is only ever continued.
struct BrokerCursorTask {
    vs_scan* s;
    uint32_t target;
    bool release;
};
// `via`: the handle the work runs through — the broker's index on its dispatcher thread, or the view of the lane the scan lives on
// (vs_broker_config.cursor_lanes); the scan's owner is blocked in vs_broker_call_lane meanwhile, so its `ix` can be lent out
static int broker_cursor_task(void* p, vs_index* via) {
    BrokerCursorTask* t = static_cast<BrokerCursorTask*>(p);
    vs_scan* s = t->s;
    if (t->release) {
        s->cur.free_all();
        return VS_OK;
    }
    vs_index* const own = s->ix;
    s->ix = via;
    const int rc = vs_guard("vs_gettuple", [&] {
        const uint8_t* prev = nullptr;
        VS_TRY(vs_index_snapshot_use(s->ix, s->snapshot, &prev));
        const int r = cursor_fetch(s, t->target);
        (void)vs_index_set_visibility_dev(s->ix, prev);  // (leaves the error text of a failed fetch alone)
        return r;
    });
    s->ix = own;
    return rc;
}
static int broker_cursor_fetch(vs_scan* s, uint32_t target) {
    BrokerCursorTask t{s, target, false};
    return vs_broker_call_lane(s->broker, s->lane, broker_cursor_task, &t);
}

static int vs_rescan_impl(vs_scan* s, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                         uint32_t L, uint32_t rescore) {
    VS_REQUIRE(s, "vs_rescan: scan is NULL");
    VS_REQUIRE(L >= 1 && L <= 10000, "diskann.query_search_list_size %u outside [1,10000]", L);
    VS_REQUIRE(rescore <= 1000, "diskann.query_rescore %u outside [0,1000]", rescore);
    vs_index* ix = s->ix;
    s->null_query = query == nullptr;
    if (query) s->query.assign(query, query + ix->d.dim_full);
    else s->query.assign(ix->d.dim_full, 0.0f);  // PgVector::zeros (AM/labels/mod.rs:214-216)
    s->labels.assign(labels ? labels : nullptr, labels ? labels + n_labels : nullptr);
    s->has_label_key = has_label_key != 0;
    s->L = L;
    s->rescore = rescore;
    s->snapshot = s->snapshot_next;
    s->cursor = 0;
    s->window = 0;
    s->calls_after_end = 0;
    s->exhausted = false;
    s->active = true;
    s->ids.clear();
    s->tids.clear();
    s->dist.clear();
    s->stats = vs_stats{};
    s->cur.open = false;  // the device state is rebuilt by the first amgettuple
    s->cur.restarts = 0;
    s->cur.launches = 0;
    s->cur.lost_visits = s->cur.lost_dq = s->cur.lost_cand = s->cur.lost_reads = 0;
    s->cur.run_visits = s->cur.run_dq = s->cur.run_cand = s->cur.run_reads = 0;
    return VS_OK;
}
extern "C" int vs_rescan(vs_scan* s, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key,
                         uint32_t L, uint32_t rescore) {
    return vs_guard("vs_rescan", [&] { return vs_rescan_impl(s, query, labels, n_labels, has_label_key, L, rescore); });
}


static int vs_gettuple_impl(vs_scan* s, uint64_t* heap_tid, uint32_t* node, float* dist) {
    if (!s || !s->active) {
        vs_set_error("vs_gettuple before vs_rescan");
        return VS_ERR_STATE;
    }
    if (s->cursor >= s->window && !s->exhausted) {
        int r;
        // continue the scan on the device for a few rows more than asked for (1/16 of what was pulled so far, 8..256): the
        // launch overhead is shared by those rows and the scan never runs more than ~6 % ahead of the executor
        const uint32_t ahead = std::min<uint32_t>(256, std::max<uint32_t>(8, s->cursor / 16));
        if (s->broker && s->window == 0 && !s->cur.open) {
            // the first rows of a scan on a broker come out of a launch shared with the other backends' scans (a LIMIT <= 16 never
            // needs more); an executor that keeps pulling gets a cursor of its own, which runs the first rows once more
            r = scan_fetch(s, 16u);
        } else if (s->broker) {
            r = broker_cursor_fetch(s, s->cursor + ahead);
        } else {
            r = cursor_fetch(s, s->cursor + ahead);
        }
        if (r != VS_OK) return r;
    }
    if (s->cursor >= s->window) {
        s->calls_after_end++;
        return 0;
    }
    if (heap_tid) *heap_tid = s->tids[s->cursor];
    if (node) *node = s->ids[s->cursor];
    if (dist) *dist = s->dist[s->cursor];
    s->cursor++;
    return 1;
}
extern "C" int vs_gettuple(vs_scan* s, uint64_t* heap_tid, uint32_t* node, float* dist) {
    return vs_guard("vs_gettuple", [&] { return vs_gettuple_impl(s, heap_tid, node, dist); });
}


extern "C" int vs_scan_xs_recheck(const vs_scan* s) { return (s && s->has_label_key) ? 1 : 0; }  // AM/scan.rs:350-352

// GreedySearchStats as the reference's scan holds them after the amgettuple calls made so far (AM/stats.rs:68-125): the work
// counters were recorded when each stream row was emitted, so rows the library prefetched beyond the executor's position are
// not in them.  After j calls the reference has pulled rescore + j - 1 rows out of next() (j when there is no window); once
// the stream has ended every further call asks next() once more in vain.
extern "C" int vs_scan_get_stats(const vs_scan* s, vs_stats* out) {
    VS_REQUIRE(s && out, "vs_scan_get_stats: bad args");
    if (!s->cur.open && (!s->broker || !s->active)) {
        *out = s->stats;
        return VS_OK;
    }
    const ScanCursor& k = s->cur;
    const vs_index* ix = s->ix;
    vs_stats st{};
    const uint32_t S = effective_rescore(ix, s->rescore);
    const uint64_t calls = (uint64_t)s->cursor + s->calls_after_end;
    if (calls == 0) {
        *out = st;
        return VS_OK;
    }
    const uint64_t need = S > 0 ? S + calls - 1 : calls;  // rows those calls asked next() for
    if (s->broker && (!k.open || (need > k.rows && !k.exhausted))) {
        // a scan on a broker whose rows so far came out of a shared launch (whose counters belong to no single scan): the scan is
        // replayed on a cursor of its own up to the executor's position, which is where the reference's counters stand
        vs_scan* m = const_cast<vs_scan*>(s);
        VS_TRY(broker_cursor_fetch(m, std::max<uint32_t>(m->cursor + (m->calls_after_end ? 1u : 0u), 1u)));
    }
    const uint32_t* r;
    uint64_t rows_used, next_calls;
    if (need <= k.rows) {
        r = k.row_stats_h.data() + (size_t)(need - 1) * ST_N;
        rows_used = need;
        next_calls = r[ST_NEXT];
    } else {
        VS_REQUIRE(k.exhausted, "vs_scan_get_stats: scan cursor out of step");
        r = k.final_counters;
        rows_used = k.rows;
        // calls j with rescore + j - 1 > rows (j > rows when there is no window) each found next() empty once; the kernel's
        // final counters hold the first of them
        const uint64_t first_empty = S > 0 ? (k.rows + 1 > S ? k.rows + 1 - S : 0) + 1 : (uint64_t)k.rows + 1;
        const uint64_t empty_calls = calls >= first_empty ? calls - first_empty + 1 : 0;
        next_calls = (uint64_t)r[ST_NEXT] - 1 + std::max<uint64_t>(empty_calls, 1);
    }
    st.queries = 1;
    st.visited_nodes = r[ST_VISITS];
    st.candidate_nodes = r[ST_CAND];
    if (ix->d.storage_type == VS_STORAGE_PLAIN) st.full_distance_comparisons = r[ST_DQ];
    else st.quantized_distance_comparisons = r[ST_DQ];
    st.node_reads = r[ST_READS];
    st.next_calls = next_calls;
    if (S > 0) {
        const uint64_t nr = rows_used + (k.masked ? r[ST_INVIS] : 0u);
        st.full_distance_comparisons += nr;
        st.node_heap_reads += nr;
    }
    st.retries = k.restarts;
    *out = st;
    return VS_OK;
}

// What the device really did for this scan since the last vs_rescan (prefetched rows and restarts included): the reference's
// counters of vs_scan_get_stats never exceed these, and the difference is the price of prefetching.
extern "C" int vs_scan_get_work(const vs_scan* s, vs_stats* out, uint32_t* launches) {
    VS_REQUIRE(s && out, "vs_scan_get_work: bad args");
    vs_stats st{};
    const ScanCursor& k = s->cur;
    st.queries = 1;
    st.visited_nodes = k.run_visits + k.lost_visits;
    st.candidate_nodes = k.run_cand + k.lost_cand;
    if (s->ix->d.storage_type == VS_STORAGE_PLAIN) st.full_distance_comparisons = k.run_dq + k.lost_dq;
    else st.quantized_distance_comparisons = k.run_dq + k.lost_dq;
    st.node_reads = k.run_reads + k.lost_reads;
    st.retries = k.restarts;
    *out = st;
    if (launches) *launches = k.launches;
    return VS_OK;
}

extern "C" void vs_endscan(vs_scan* s) {
    if (!s) return;
    if (s->broker && (s->cur.open || s->cur.state.p)) {  // the cursor's device buffers go where they came from: the dispatcher thread
        BrokerCursorTask t{s, 0, true};
        (void)vs_broker_call_lane(s->broker, s->lane, broker_cursor_task, &t);  // (a broker that is shutting down: freed below, by this thread)
    }
    delete s;
}

extern "C" int vs_scan_set_snapshot(vs_scan* s, uint32_t snapshot) {
    VS_REQUIRE(s && snapshot < VS_MAX_SNAPSHOTS, "vs_scan_set_snapshot: snapshot id outside [0,%d]", VS_MAX_SNAPSHOTS - 1);
    VS_REQUIRE(s->broker, "vs_scan_set_snapshot: a direct scan runs under the index's current mask (vs_index_set_visibility)");
    s->snapshot_next = snapshot;
    return VS_OK;
}

static int vs_scan_prefetch_impl(vs_scan* s, uint32_t rows) {
    if (!s || !s->active) {
        vs_set_error("vs_scan_prefetch before vs_rescan");
        return VS_ERR_STATE;
    }
    if (rows <= s->window || s->exhausted) return VS_OK;
    if (s->broker && s->window == 0 && !s->cur.open) {
        VS_TRY(scan_fetch(s, std::min<uint32_t>(rows, 1024u)));  // (a shared launch hands out up to 1024 rows per scan)
        if (rows <= s->window || s->exhausted) return VS_OK;
    }
    return s->broker ? broker_cursor_fetch(s, rows) : cursor_fetch(s, rows);
}
extern "C" int vs_scan_prefetch(vs_scan* s, uint32_t rows) {
    return vs_guard("vs_scan_prefetch", [&] { return vs_scan_prefetch_impl(s, rows); });
}
