// vs_scanpool.cpp — BATCHED cursor continuations: many backends' scans that are being streamed (amgettuple past the first rows,
// AM/scan.rs:369-436) continued by launches they SHARE.
//
// A single cursor (vs_beginscan / vs_gettuple, vs_api.hip) continues its scan with one resumable launch of the general kernel per
// fetch: one wave on the whole chip and about half a millisecond of launch / synchronise / copy overhead.  64 backends streaming at
// once therefore cost 64 launches per round, serialised over the few hardware queues HIP maps its streams onto: 22 x one cursor's wall
// time in round 4 (profiles/r04/cursor_concurrency_1m.txt).  A scan pool keeps the device state of up to `capacity` scans — what the
// reference keeps per scan in TSVResponseIterator: `lsr` + `resort_buffer` (AM/scan.rs:162-174) — in POOLED arrays, one region per
// slot, so that one resumed k_search launch (nq = slots, the scans that need rows marked in status[], SearchLaunch::only_failed), one
// k_rerank launch and one round of k_resort_cursor launches serve every scan that asked for rows in this round.  Nothing here is a new
// kernel: the launch wrappers of vs_internal.h are the single cursor's.
//
// Semantics are the single cursor's, row for row (tests/test_gpu_zt_scanpool.py holds the pool to the oracle and to vs_gettuple):
// rows in next_with_resort order (AM/scan.rs:244-305), GreedySearchStats as they stand after exactly the calls made so far (the work
// counters are recorded per emitted stream row).  All scans of a pool share the GUCs (diskann.query_search_list_size, query_rescore:
// session-level in PostgreSQL) and the snapshot mask in force; scan keys are per scan.  A scan that outgrows the pool's fixed
// capacities (more than rows_cap stream rows) fails with VS_ERR_CAPACITY and is continued by a cursor of its own.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "vs_internal.h"

namespace {

struct Slot {
    bool active = false, null_query = false, keys = false, exhausted = false, failed = false, masked = false;
    std::vector<int16_t> labels;  // sorted, distinct (LabelSet::from, AM/labels/mod.rs:30-37)
    uint32_t rows = 0;            // stream rows emitted so far
    uint32_t handed = 0;          // rows handed to the caller
    uint32_t calls_after_end = 0;
    std::vector<uint32_t> row_stats;  // [rows][ST_N]
    uint32_t final_counters[ST_N] = {0};
    uint32_t launches = 0;
};

}  // namespace

struct vs_scan_pool {
    vs_index* ix = nullptr;
    uint32_t cap = 0, L = 0, rescore = 0, S = 0, kmax = 0, rows_cap = 0, mmax = 0;
    uint32_t hl = 0, hcap = 0, vcap = 0, lh = 0, hashcap = 0, g0 = 0, rw = 0;
    // round 6: the continuations run the RESUMABLE instantiation of the fast kernel (k_search_fast, OPT_RS: heap top, occupancy bits and
    // visited ring on chip, 16-bit dedup tables) where the index and the capacities allow it — the general kernel otherwise (plain
    // storage, codes wider than the register-resident widths, a state that does not fit LDS, VS_POOL_FAST=0).  `fl` holds the geometry.
    bool fast = false;
    FastLaunch fl{};
    std::vector<Slot> slots;
    bool csr_dirty = true;
    DevBuf raw_q, q_full, q_index, qcodes, qlabels, qlabel_off, heap_g, hash, state, cnt, stats, status, row_stats, stage, all, resort_heap,
        out_tids, list, roff;
    uint64_t* d_tids = nullptr;
    uint32_t *d_cur = nullptr, *d_ids = nullptr;
    float* d_dist = nullptr;
    // round 6: a round that is PREFETCHED — launched on a stream of its own at the end of a fetch for the scans that are being streamed,
    // not waited for — so that the search of the next rows runs while the backends consume the rows they have (see prefetch_round).
    // One round is in flight at most; its results come back into pinned host memory and are booked when a later call finds it done
    // (or needs its rows).  The staging arrays of a round (cnt, status, stats, row_stats, stage, roff) belong to the round in flight.
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_round = nullptr;
    uint32_t* h_pin = nullptr;  // [kst G x ST_N | kstatus 2 G | cnt G | rst G x mmax x ST_N]
    struct {
        bool active = false, masked = false;
        std::vector<uint32_t> run;
        uint32_t M = 0, nq = 0;
    } fly;
    uint64_t launches = 0, rounds = 0, fetches = 0, scans_served = 0, prefetched = 0, prefetch_waits = 0;
    double t_search = 0, t_append = 0, t_resort = 0;  // (VS_POOL_DEBUG) host seconds: launch .. counters back / rerank + append / resort + rows back
    void free_all() {
        if (stream2) {
            (void)hipStreamSynchronize(stream2);
            (void)hipStreamDestroy(stream2);
            stream2 = nullptr;
        }
        if (ev_round) (void)hipEventDestroy(ev_round);
        ev_round = nullptr;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        for (DevBuf* b : {&raw_q, &q_full, &q_index, &qcodes, &qlabels, &qlabel_off, &heap_g, &hash, &state, &cnt, &stats, &status, &row_stats,
                          &stage, &all, &resort_heap, &out_tids, &list, &roff}) {
            if (b->p && !b->in_slab) (void)hipFree(b->p);
            b->p = nullptr;
            b->bytes = 0;
        }
    }
};

static int settle_prefetch(vs_scan_pool* p, bool wait);

static uint32_t pool_env_u32(const char* name, uint32_t dflt) {
    const char* v = vs_opt_get(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

static int scanpool_create_impl(vs_index* ix, uint32_t capacity, uint32_t L, uint32_t rescore, uint32_t kmax, uint32_t rows_cap, vs_scan_pool** out) {
    VS_REQUIRE(ix && out && capacity >= 1 && capacity <= 4096, "vs_scanpool_create: bad args (1..4096 slots)");
    VS_REQUIRE(L >= 1 && L <= 10000, "diskann.query_search_list_size %u outside [1,10000]", L);
    VS_REQUIRE(rescore <= 1000, "diskann.query_rescore %u outside [0,1000]", rescore);
    VS_REQUIRE(kmax >= 1 && kmax <= 4096, "vs_scanpool_create: rows per fetch outside [1,4096]");
    *out = nullptr;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    vs_scan_pool* p = new vs_scan_pool();
    p->ix = ix;
    p->cap = capacity;
    p->L = L;
    p->rescore = rescore;
    // amgettuple, Plain arm: num_dimensions == num_dimensions_to_index => "no need to resort" (AM/scan.rs:392-399)
    p->S = (ix->d.storage_type == VS_STORAGE_PLAIN && ix->d.dim_index == ix->d.dim_full) ? 0u : rescore;
    p->kmax = kmax;
    p->rows_cap = std::max<uint32_t>(rows_cap ? rows_cap : 4096, p->S + kmax);
    p->mmax = p->S + kmax;  // the first fetch of a scan needs rescore + k - 1 stream rows
    // capacities of the general kernel for a scan of rows_cap rows (as cursor_open sizes a single cursor for its horizon)
    const uint64_t visits = 2ull * L + p->rows_cap + 32;
    const uint64_t pushes = visits * ix->d.num_neighbors;
    p->hl = pool_env_u32("VS_HL", 1024);
    p->lh = 0;
    p->g0 = pool_env_u32("VS_G0", 4096);
    p->hcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pushes, p->hl), 1u << 24);
    p->vcap = (uint32_t)std::min<uint64_t>(2ull * L + 256, 1u << 20);
    p->hashcap = std::max<uint32_t>(next_pow2_u32(std::min<uint64_t>(2ull * pushes, 1u << 26)), p->g0);
    SearchLaunch probe{};
    probe.hl = p->hl;
    probe.lh = p->lh;
    probe.vcap = p->vcap;
    p->rw = (uint32_t)search_resume_words(probe);
    if (ix->d.storage_type != VS_STORAGE_PLAIN && pool_env_u32("VS_POOL_FAST", 1) && ix->d.n > 0) {
        FastLaunch f{};
        f.hl = 511;
        f.hcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pushes, f.hl), 1u << 22);
        f.gstride = round_up_u32(f.hcap - f.hl + 2, 2);
        f.vcap = round_up_u32(std::max<uint32_t>(2u * (uint32_t)std::min<uint64_t>((uint64_t)L + L / 2 + 32, 1u << 16), 64), 64);
        f.minw = 1;
        f.gcap = std::max<uint32_t>(next_pow2_u32(std::min<uint64_t>(pushes * 4 / 3 + 256, 1u << 22)), 1024);
        f.ocap = std::max<uint32_t>(round_up_u32(f.gcap / 16, 32), 256);
        f.vwords = (f.gcap + f.ocap) / 32;
        f.vslot = 2;
        while ((1ull << f.sb) < (uint64_t)f.gcap + f.ocap) f.sb++;
        uint32_t qd = 1, lb = 0;
        while ((1ull << qd) < (uint64_t)std::max<uint32_t>(ix->d.n, 2)) qd++;
        while ((1u << lb) < (f.gcap >> 3)) lb++;
        if (qd < lb + 3) qd = lb + 3;
        f.qd = qd;
        f.qk = qd - lb;
        f.gregion = (f.gcap >> 1) + f.ocap;
        f.glimit = (uint32_t)((uint64_t)f.gcap * 3 / 4) - 64u;
        const uint64_t nbits = (uint64_t)ix->d.dim_index * ix->d.bits;
        const uint32_t nch = (ix->code_stride + 7) / 8;
        if (f.qk <= 16 && f.qd <= 32 && nbits < (1ull << (32 - f.sb)) && nch <= 6 && fast_lds_bytes(ix, f) <= 96 * 1024) {
            p->fast = true;
            p->fl = f;
            p->rw = (uint32_t)fast_resume_words(f);
        }
    }
    p->slots.resize(capacity);
    const size_t G = capacity;
    auto all = [&]() -> int {
        VS_TRY(devbuf_reserve(c, p->raw_q, G * ix->d.dim_full * 4));
        VS_TRY(devbuf_reserve(c, p->q_full, G * ix->vec_stride * 4));
        VS_TRY(devbuf_reserve(c, p->qcodes, G * ix->code_stride * 8 + 16));
        if (ix->d.storage_type == VS_STORAGE_PLAIN && ix->d.dim_index < ix->d.dim_full) VS_TRY(devbuf_reserve(c, p->q_index, G * ix->vec_stride * 4));
        VS_TRY(devbuf_reserve(c, p->qlabel_off, (G + 1) * 4));
        VS_TRY(devbuf_reserve(c, p->qlabels, 64));
        if (p->fast) {
            VS_TRY(devbuf_reserve(c, p->heap_g, std::max<size_t>(G * (size_t)p->fl.gstride * 4, 16)));
            VS_TRY(devbuf_reserve(c, p->hash, G * (size_t)p->fl.gregion * 4));
        } else {
            VS_TRY(devbuf_reserve(c, p->heap_g, std::max<size_t>(G * (size_t)(p->hcap > p->hl ? p->hcap - p->hl : 0) * 8, 16)));
            VS_TRY(devbuf_reserve(c, p->hash, G * (size_t)p->hashcap * 4));
        }
        VS_TRY(devbuf_reserve(c, p->state, G * (size_t)p->rw * 4));
        VS_TRY(devbuf_reserve(c, p->cnt, G * 4));
        VS_TRY(devbuf_reserve(c, p->stats, G * ST_N * 4));
        VS_TRY(devbuf_reserve(c, p->status, 2 * G * 4));  // (one run mask per launch of a round: unkeyed / keyed scans)
        VS_TRY(devbuf_reserve(c, p->row_stats, G * (size_t)p->mmax * ST_N * 4));
        VS_TRY(devbuf_reserve(c, p->stage, 3 * G * (size_t)p->mmax * 4));     // [ids | ham | dist][slot][M of the round]
        VS_TRY(devbuf_reserve(c, p->all, 3 * G * (size_t)p->rows_cap * 4));   // [ids | ham | dist][slot][rows_cap]
        VS_TRY(devbuf_reserve(c, p->resort_heap, std::max<size_t>(G * (size_t)p->S * 8, 16)));
        // what a fetch brings back in ONE copy: [tids G x kmax u64 | cur G x 4 u32 | ids G x kmax u32 | dist G x kmax f32]
        VS_TRY(devbuf_reserve(c, p->out_tids, G * (size_t)kmax * 16 + G * 16));
        p->d_tids = (uint64_t*)p->out_tids.p;
        p->d_cur = (uint32_t*)(p->d_tids + G * (size_t)kmax);
        p->d_ids = p->d_cur + G * 4;
        p->d_dist = (float*)(p->d_ids + G * (size_t)kmax);
        VS_TRY(devbuf_reserve(c, p->list, G * 12));  // (slot, rows, exhausted) of the scans a fetch lists
        VS_TRY(devbuf_reserve(c, p->roff, G * 4));   // where a round's new rows go in every slot's stream arrays
        VS_HIP(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
        VS_HIP(hipEventCreateWithFlags(&p->ev_round, hipEventDisableTiming));
        VS_HIP(hipHostMalloc((void**)&p->h_pin, (G * ST_N + 2 * G + G + G * (size_t)p->mmax * ST_N) * 4, hipHostMallocDefault));
        return VS_OK;
    };
    const int r = all();
    if (r != VS_OK) {
        p->free_all();
        delete p;
        return r;
    }
    *out = p;
    return VS_OK;
}
extern "C" int vs_scanpool_create(vs_index* ix, uint32_t capacity, uint32_t L, uint32_t rescore, uint32_t kmax, uint32_t rows_cap, vs_scan_pool** out) {
    return vs_guard("vs_scanpool_create", [&] { return scanpool_create_impl(ix, capacity, L, rescore, kmax, rows_cap, out); });
}

extern "C" void vs_scanpool_free(vs_scan_pool* p) {
    if (!p) return;
    if (pool_env_u32("VS_POOL_DEBUG", 0))
        fprintf(stderr, "[VS_POOL_DEBUG] pool L=%u rescore=%u (%s kernel): %llu fetch calls serving %llu scan chunks in %llu rounds (%llu search launches); host ms: "
                        "search %.1f, rerank+append %.1f, resort+rows %.1f\n", p->L, p->rescore, p->fast ? "resumable fast" : "general", (unsigned long long)p->fetches,
                (unsigned long long)p->scans_served, (unsigned long long)p->rounds, (unsigned long long)p->launches, p->t_search * 1e3, p->t_append * 1e3,
                p->t_resort * 1e3);
    if (pool_env_u32("VS_POOL_DEBUG", 0))
        fprintf(stderr, "[VS_POOL_DEBUG]   %llu of the rounds were prefetched, %llu of those were waited for\n", (unsigned long long)p->prefetched,
                (unsigned long long)p->prefetch_waits);
    (void)hipSetDevice(p->ix->ctx->device);
    (void)hipStreamSynchronize(p->ix->ctx->stream);
    p->free_all();
    delete p;
}

// amrescan of one slot (AM/scan.rs:335-367): prepares the query (normalise, SBQ code), stores the key, empties the slot's saved scan
static int scanpool_rescan_impl(vs_scan_pool* p, uint32_t slot, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key) {
    VS_REQUIRE(p && slot < p->cap, "vs_scanpool_rescan: bad slot");
    vs_index* ix = p->ix;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    VS_TRY(settle_prefetch(p, true));  // (a round in flight may be continuing this slot's former scan)
    Slot& s = p->slots[slot];
    const bool keys = has_label_key != 0 && query != nullptr;
    if (ix->d.storage_type == VS_STORAGE_PLAIN) VS_REQUIRE(!keys, "Plain storage does not support label filters");  // AM/plain/storage.rs:262
    if (keys) VS_REQUIRE(ix->d.has_labels && ix->label_off, "label scan keys on an index without labels");
    s = Slot{};
    s.active = true;
    s.null_query = query == nullptr;
    s.keys = keys;
    if (keys) {
        s.labels.assign(labels, labels + n_labels);
        std::sort(s.labels.begin(), s.labels.end());
        s.labels.erase(std::unique(s.labels.begin(), s.labels.end()), s.labels.end());
    }
    p->csr_dirty = true;
    std::vector<float> q(ix->d.dim_full, 0.0f);  // a NULL query is the zero vector (AM/labels/mod.rs:214-216)
    if (query) memcpy(q.data(), query, (size_t)ix->d.dim_full * 4);
    float* rq = (float*)p->raw_q.p + (size_t)slot * ix->d.dim_full;
    VS_TRY(vs_dev_upload(c, rq, q.data(), (size_t)ix->d.dim_full * 4));
    VS_TRY(launch_prepare_queries(ix, rq, 1, (float*)p->q_full.p + (size_t)slot * ix->vec_stride, (uint64_t*)p->qcodes.p + (size_t)slot * ix->code_stride));
    if (p->q_index.p) VS_TRY(launch_prepare_index_slice(ix, rq, 1, (float*)p->q_index.p + (size_t)slot * ix->vec_stride));
    VS_HIP(hipMemsetAsync((uint32_t*)p->state.p + (size_t)slot * p->rw, 0, RS_HDR * 4, c->stream));
    VS_HIP(hipMemsetAsync(p->d_cur + (size_t)slot * 4, 0, 16, c->stream));
    return VS_OK;
}
extern "C" int vs_scanpool_rescan(vs_scan_pool* p, uint32_t slot, const float* query, const int16_t* labels, uint32_t n_labels, int has_label_key) {
    return vs_guard("vs_scanpool_rescan", [&] { return scanpool_rescan_impl(p, slot, query, labels, n_labels, has_label_key); });
}

extern "C" int vs_scanpool_endscan(vs_scan_pool* p, uint32_t slot) {
    VS_REQUIRE(p && slot < p->cap, "vs_scanpool_endscan: bad slot");
    if (p->fly.active && std::find(p->fly.run.begin(), p->fly.run.end(), slot) != p->fly.run.end()) {
        (void)hipSetDevice(p->ix->ctx->device);
        VS_TRY(settle_prefetch(p, true));  // (its region of the pooled arrays is being written)
    }
    p->slots[slot] = Slot{};
    p->csr_dirty = true;
    return VS_OK;
}

// the label keys of every slot as one CSR (slots without a key: an empty range; they are not part of a keyed launch)
static int upload_csr(vs_scan_pool* p) {
    if (!p->csr_dirty) return VS_OK;
    vs_ctx* c = p->ix->ctx;
    std::vector<uint32_t> off(p->cap + 1, 0);
    std::vector<int16_t> val;
    for (uint32_t i = 0; i < p->cap; ++i) {
        const Slot& s = p->slots[i];
        if (s.active && s.keys) val.insert(val.end(), s.labels.begin(), s.labels.end());
        off[i + 1] = (uint32_t)val.size();
    }
    VS_TRY(devbuf_reserve(c, p->qlabels, std::max<size_t>(val.size(), 1) * 2));
    if (!val.empty()) VS_TRY(vs_dev_upload(c, p->qlabels.p, val.data(), val.size() * 2));
    VS_TRY(vs_dev_upload(c, p->qlabel_off.p, off.data(), off.size() * 4));
    p->csr_dirty = false;
    return VS_OK;
}

// One round: every slot of `run` is continued for M more stream rows by launches they share (keyed and unkeyed scans are two launches:
// "labels is Some" is a property of a launch, AM/labels/mod.rs:222-236), their new rows are reranked together and appended to the
// slots' stream arrays.
// round_launch puts the whole round on the context's current stream, the results' way back into the pinned block included, and does not
// wait; round_finish books what came back.
static int round_launch(vs_scan_pool* p, const std::vector<uint32_t>& run, uint32_t M) {
    vs_index* ix = p->ix;
    vs_ctx* c = ix->ctx;
    const uint32_t G = p->cap;
    const bool plain = ix->d.storage_type == VS_STORAGE_PLAIN;
    uint32_t nq = 0;
    for (uint32_t q : run) nq = std::max(nq, q + 1);
    VS_TRY(upload_csr(p));
    VS_HIP(hipMemsetAsync(p->cnt.p, 0, (size_t)G * 4, c->stream));
    uint32_t* const stage_ids = (uint32_t*)p->stage.p;
    uint32_t* const stage_ham = stage_ids + (size_t)G * p->mmax;
    float* const stage_dist = (float*)(stage_ham + (size_t)G * p->mmax);
    for (int keyed = 0; keyed < 2; ++keyed) {
        std::vector<uint32_t> mask(nq, 0);
        bool any = false;
        for (uint32_t q : run)
            if ((int)p->slots[q].keys == keyed) {
                mask[q] = 1;
                any = true;
            }
        if (!any) continue;
        uint32_t* const d_status = (uint32_t*)p->status.p + (size_t)keyed * G;
        VS_HIP(hipMemcpyAsync(d_status, mask.data(), (size_t)nq * 4, hipMemcpyHostToDevice, c->stream));  // (pageable source: staged before the call returns)
        if (p->fast) {
            FastLaunch f = p->fl;
            f.nq = nq;
            f.L = p->L;
            f.M = M;
            f.qcodes = (const uint64_t*)p->qcodes.p;
            f.qlabels = keyed ? (const int16_t*)p->qlabels.p : nullptr;
            f.qlabel_off = keyed ? (const uint32_t*)p->qlabel_off.p : nullptr;
            f.heap_g = (uint32_t*)p->heap_g.p;
            f.ghash = (uint32_t*)p->hash.p;
            f.pool_counter = nullptr;
            f.pool_slots = G;
            f.out_ids = stage_ids;  // [slot][M]
            f.out_ham = stage_ham;
            f.out_cnt = (uint32_t*)p->cnt.p;
            f.stats = (uint32_t*)p->stats.p;
            f.status = d_status;  // the run mask on entry, the scan's outcome on exit
            f.visible = p->S > 0 ? ix->visible : nullptr;
            f.resume = (uint32_t*)p->state.p;
            f.resume_stride = p->rw;
            f.row_stats = (uint32_t*)p->row_stats.p;
            hipEvent_t ev = prof_begin(c);
            const int lr = launch_search_fast(ix, f);
            prof_end(c, PK_SEARCH, ev);
            VS_TRY(lr);
            p->launches++;
            continue;
        }
        SearchLaunch sl;
        sl.nq = nq;
        sl.L = p->L;
        sl.M = M;
        sl.hl = p->hl;
        sl.hcap = p->hcap;
        sl.vcap = p->vcap;
        sl.lh = p->lh;
        sl.hashcap = p->hashcap;
        sl.g0 = p->g0;
        sl.qcodes = (const uint64_t*)p->qcodes.p;
        sl.qlabels = keyed ? (const int16_t*)p->qlabels.p : nullptr;
        sl.qlabel_off = keyed ? (const uint32_t*)p->qlabel_off.p : nullptr;
        sl.heap_g = (uint64_t*)p->heap_g.p;
        sl.hash = (uint32_t*)p->hash.p;
        sl.out_ids = stage_ids;  // [slot][M]
        sl.out_ham = stage_ham;
        sl.out_cnt = (uint32_t*)p->cnt.p;
        sl.stats = (uint32_t*)p->stats.p;
        sl.status = d_status;
        sl.only_failed = 1;  // only the slots marked above run; the others return at once and keep their saved state
        sl.visible = p->S > 0 ? ix->visible : nullptr;  // the heap is only fetched for the rescore window
        sl.resume = (uint32_t*)p->state.p;
        sl.resume_stride = p->rw;
        sl.row_stats = (uint32_t*)p->row_stats.p;  // [slot][M][ST_N]
        // the plain-storage kernel reads its prepared queries from the batch workspace slots: point them at the pool's
        void* const ws_q_full = ix->ws.q_full.p;
        void* const ws_q_index = ix->ws.q_index.p;
        if (plain) {
            ix->ws.q_full.p = p->q_full.p;
            ix->ws.q_index.p = p->q_index.p;
        }
        hipEvent_t ev = prof_begin(c);
        const int lr = launch_search(ix, sl);
        prof_end(c, PK_SEARCH, ev);
        if (plain) {
            ix->ws.q_full.p = ws_q_full;
            ix->ws.q_index.p = ws_q_index;
        }
        VS_TRY(lr);
        p->launches++;
    }
    p->rounds++;
    // get_full_distance_for_resort of the new rows (AM/sbq/storage.rs:304-328: one launch for every slot), then ids, Hamming keys and
    // distances go from the round's staging rows to the slots' stream arrays — both take the row counts from the device, so the whole
    // round is ONE trip to the host (round 6; until then: counts back, then a 2-D copy per slot, then a second synchronisation)
    if (p->S > 0) {
        VS_REQUIRE(ix->vecs, "diskann.query_rescore > 0 needs the heap vector column on the device");
        hipEvent_t ev2 = prof_begin(c);
        VS_TRY(launch_rerank(ix, (const float*)p->q_full.p, stage_ids, nullptr, (const uint32_t*)p->cnt.p, M, nq, stage_dist));
        prof_end(c, PK_RERANK, ev2);
    }
    {
        std::vector<uint32_t> off(nq, p->rows_cap);  // (a slot that is not part of the round has no rows to move: its count is zero too)
        for (uint32_t q : run) off[q] = p->slots[q].rows;
        VS_HIP(hipMemcpyAsync(p->roff.p, off.data(), (size_t)nq * 4, hipMemcpyHostToDevice, c->stream));  // (pageable source: staged before the call returns)
        VS_TRY(launch_pool_append(ix, nq, (const uint32_t*)p->cnt.p, (const uint32_t*)p->roff.p, stage_ids, (uint32_t)((size_t)G * p->mmax), M, (uint32_t*)p->all.p,
                                  (uint32_t)((size_t)G * p->rows_cap), p->rows_cap));
    }
    // outcome, counters and row counts of the slots that ran (what the kernel publishes per scan next to the saved state)
    uint32_t* const kst = p->h_pin;
    uint32_t* const kstatus = kst + (size_t)G * ST_N;
    uint32_t* const cnt = kstatus + 2 * (size_t)G;
    uint32_t* const rst = cnt + G;
    VS_HIP(hipMemcpyAsync(kst, p->stats.p, (size_t)nq * ST_N * 4, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(kstatus, p->status.p, (size_t)2 * G * 4, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(cnt, p->cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(rst, p->row_stats.p, (size_t)nq * M * ST_N * 4, hipMemcpyDeviceToHost, c->stream));
    return VS_OK;
}

static void round_finish(vs_scan_pool* p, const std::vector<uint32_t>& run, uint32_t M, bool masked) {
    const uint32_t G = p->cap;
    const uint32_t* const kst = p->h_pin;
    const uint32_t* const kstatus = kst + (size_t)G * ST_N;
    const uint32_t* const cnt = kstatus + 2 * (size_t)G;
    const uint32_t* const rst = cnt + G;
    for (uint32_t q : run) {
        Slot& s = p->slots[q];
        if (!s.active) continue;  // (ended while the round was in flight)
        const uint32_t* h = kst + (size_t)q * ST_N;
        s.launches++;
        s.masked = masked;
        if (kstatus[(size_t)(s.keys ? G : 0) + q] != 0) {  // a structure outgrew the pool's capacities: the scan continues on a cursor of its own
            s.failed = true;
            continue;
        }
        const uint32_t n = std::min(cnt[q], M);
        s.row_stats.insert(s.row_stats.end(), rst + (size_t)q * M * ST_N, rst + ((size_t)q * M + n) * ST_N);
        s.rows += n;
        if (n < M) {
            s.exhausted = true;
            memset(s.final_counters, 0, sizeof(s.final_counters));
            s.final_counters[ST_VISITS] = h[ST_VISITS];
            s.final_counters[ST_CAND] = h[ST_CAND];
            s.final_counters[ST_DQ] = h[ST_DQ];
            s.final_counters[ST_READS] = h[ST_READS];
            s.final_counters[ST_NEXT] = h[ST_NEXT];
            s.final_counters[ST_INVIS] = h[ST_INVIS];
        }
    }
}

// the round in flight, if any: booked when it has finished — or waited for when `wait` (a listed scan needs its rows, another round wants
// the staging arrays, a slot is rescanned)
static int settle_prefetch(vs_scan_pool* p, bool wait) {
    if (!p->fly.active) return VS_OK;
    if (!wait) {
        const hipError_t e = hipEventQuery(p->ev_round);
        if (e == hipErrorNotReady) return VS_OK;
        VS_HIP(e);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        VS_HIP(hipEventSynchronize(p->ev_round));
        p->t_search += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        p->prefetch_waits++;
    }
    p->fly.active = false;
    round_finish(p, p->fly.run, p->fly.M, p->fly.masked);
    return VS_OK;
}

// a round the caller waits for (the first rows of a scan, scans that are not streamed ahead, whatever a prefetched round did not cover)
static int pool_round(vs_scan_pool* p, const std::vector<uint32_t>& run, uint32_t M) {
    vs_ctx* c = p->ix->ctx;
    VS_TRY(settle_prefetch(p, true));  // (the staging arrays are one round's)
    const auto tp0 = std::chrono::steady_clock::now();
    VS_TRY(round_launch(p, run, M));
    VS_HIP(hipStreamSynchronize(c->stream));
    const auto tp1 = std::chrono::steady_clock::now();
    p->t_search += std::chrono::duration<double>(tp1 - tp0).count();
    round_finish(p, run, M, p->S > 0 && p->ix->visible != nullptr);
    p->t_append += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
    return VS_OK;
}

// The next round of the scans that are being streamed, launched WITHOUT waiting at the end of a fetch (round 6).  A backend that pulls a
// long scan chunk by chunk leaves the GPU idle between its fetches, and the round its fetch eventually needs costs the life of one wave
// (0.6 ms for 64 rows at search_list_size 100) during which every backend of that fetch waits: with 64 backends streaming 1 000 rows each the
// rounds were 15 of the 35 ms.  Launched one fetch early on a stream of its own, the round runs while the rows already there are
// handed out.  What keeps it simple: ONE round in flight; only scans without label keys, only while no snapshot mask is in force
// (a mask may be replaced while the round runs) and only on the resumable fast kernel — everything else takes the rounds it waits
// for; a scan is carried at most two rounds ahead of its executor and never past its row budget; the per-row counters hide what the
// executor has not pulled.
static int prefetch_round(vs_scan_pool* p, const uint32_t* slots, uint32_t n, uint32_t k) {
    vs_index* ix = p->ix;
    vs_ctx* c = ix->ctx;
    if (p->fly.active || !p->fast || ix->visible != nullptr || !pool_env_u32("VS_POOL_PREFETCH", 1)) return VS_OK;
    const uint32_t S = p->S;
    const uint32_t M = std::min<uint32_t>(4 * k, p->mmax);
    std::vector<uint32_t> run;
    for (uint32_t i = 0; i < n; ++i) {
        const Slot& s = p->slots[slots[i]];
        if (!s.active || s.failed || s.exhausted || s.keys || s.launches < 2) continue;
        const uint64_t need = S > 0 ? (uint64_t)S + s.handed + k - 1 : (uint64_t)s.handed + k;  // what the NEXT fetch of k rows stands on
        if ((uint64_t)s.rows + M > p->rows_cap || s.rows >= need + M) continue;  // (no room / a round ahead already)
        run.push_back(slots[i]);
    }
    if (run.empty()) return VS_OK;
    std::sort(run.begin(), run.end());
    if (p->csr_dirty) {  // (the keys of the other slots changed: their upload goes over the context's own stream and is finished first)
        VS_TRY(upload_csr(p));
        VS_HIP(hipStreamSynchronize(c->stream));
    }
    hipStream_t const own = c->stream;
    c->stream = p->stream2;  // (the launch wrappers take the stream from the context; the dispatcher thread is its only user)
    const int r = round_launch(p, run, M);
    const hipError_t e = r == VS_OK ? hipEventRecord(p->ev_round, p->stream2) : hipSuccess;
    c->stream = own;
    VS_TRY(r);
    VS_HIP(e);
    p->fly.active = true;
    p->fly.masked = false;
    p->fly.run.swap(run);
    p->fly.M = M;
    p->prefetched++;
    return VS_OK;
}

// amgettuple x k for many scans at once: rows [handed, handed + k) of every listed slot (fewer when its scan ends).  out_*: [n][k];
// out_rows[i] = rows produced for slots[i] (or a negative VS_ERR_* for a slot that failed: the others are served).
static int scanpool_fetch_impl(vs_scan_pool* p, const uint32_t* slots, uint32_t n, uint32_t k, uint64_t* out_tids, uint32_t* out_ids,
                               float* out_dist, int32_t* out_rows) {
    VS_REQUIRE(p && slots && out_rows && n >= 1 && k >= 1 && k <= p->kmax, "vs_scanpool_fetch: bad args (k <= the pool's rows per fetch)");
    vs_index* ix = p->ix;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    const uint32_t G = p->cap, S = p->S;
    std::vector<uint8_t> listed(G, 0);
    for (uint32_t i = 0; i < n; ++i) {
        VS_REQUIRE(slots[i] < G && p->slots[slots[i]].active, "vs_scanpool_fetch: slot %u is not an open scan", slots[i]);
        VS_REQUIRE(!listed[slots[i]], "vs_scanpool_fetch: slot %u listed twice", slots[i]);
        listed[slots[i]] = 1;
    }
    VS_TRY(settle_prefetch(p, false));  // (a prefetched round that has finished meanwhile is booked; one that has not stays in flight)
    // ---- rounds of shared launches until every listed scan has the stream rows its k calls need (or has ended)
    for (;;) {
        std::vector<uint32_t> run;
        uint32_t M = 0;
        for (uint32_t i = 0; i < n; ++i) {
            Slot& s = p->slots[slots[i]];
            if (s.failed || s.exhausted) continue;
            const uint64_t need = S > 0 ? (uint64_t)S + s.handed + k - 1 : (uint64_t)s.handed + k;  // stream rows behind those calls
            if (need <= s.rows) continue;
            if (need > p->rows_cap) {
                s.failed = true;
                continue;
            }
            run.push_back(slots[i]);
            M = std::max<uint32_t>(M, (uint32_t)(need - s.rows));
        }
        if (run.empty()) break;
        if (p->fly.active) {  // the rows may be on their way already (and the staging arrays are the round's in flight either way)
            VS_TRY(settle_prefetch(p, true));
            continue;
        }
        // a scan that keeps being streamed is carried ahead of its executor: from its third continuation on a round produces up to
        // four chunks of rows (bounded by the round's staging rows), so most of its later fetches find their rows already there.  The
        // counters are recorded per row: what the executor has not pulled does not show in them.
        bool streaming = true;
        for (uint32_t q : run) streaming = streaming && p->slots[q].launches >= 2;
        if (streaming) M = std::max(M, std::min<uint32_t>(4 * k, p->mmax));
        M = std::min(M, p->mmax);
        // ... and the round carries the OTHER listed scans that are being streamed ahead as well (round 6).  A round costs the life of its
        // longest scan whether it continues four scans or forty, and every backend of the fetch waits for it; when only the scans that
        // had run out of rows took part, backends streaming side by side drifted into different phases and nearly every fetch paid for
        // a round (64 backends x 1 000 rows: 99 rounds per 127 fetches where one scan alone needs 18).  Taking everybody along keeps
        // them in phase: the rounds a group needs are the rounds its neediest scan needs.  Bounded: a scan is carried at most two
        // rounds ahead of its executor, never past its row budget, and what the executor has not pulled shows in no counter.
        if (streaming && pool_env_u32("VS_POOL_TOPUP", 1)) {
            std::vector<uint8_t> in_run(G, 0);
            for (uint32_t q : run) in_run[q] = 1;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t q = slots[i];
                const Slot& s = p->slots[q];
                if (in_run[q] || s.failed || s.exhausted || s.launches < 2) continue;
                const uint64_t need = S > 0 ? (uint64_t)S + s.handed + k - 1 : (uint64_t)s.handed + k;
                if ((uint64_t)s.rows + M > p->rows_cap || s.rows >= need + M) continue;  // (no room / already a round ahead)
                run.push_back(q);
            }
            std::sort(run.begin(), run.end());
        }
        // (a scan that needs fewer rows than the round's M is simply further ahead afterwards — rows are handed out by the window
        // below and the counters are recorded per row — but no scan may pass its row budget)
        for (uint32_t q : run) M = std::min(M, p->rows_cap - p->slots[q].rows);
        VS_REQUIRE(M >= 1, "vs_scanpool_fetch: internal: empty round");
        VS_TRY(pool_round(p, run, M));
    }
    // ---- next_with_resort x k per scan (AM/scan.rs:244-305): one k_resort_cursor_batch launch, a wave per listed scan
    const auto tr0 = std::chrono::steady_clock::now();
    p->fetches++;
    p->scans_served += n;
    uint32_t* const all_ids = (uint32_t*)p->all.p;
    uint32_t* const all_ham = all_ids + (size_t)G * p->rows_cap;
    float* const all_dist = (float*)(all_ham + (size_t)G * p->rows_cap);
    // one wave per listed scan in ONE launch (round 6; until round 5: one single-thread k_resort_cursor launch per scan, dealt over eight
    // streams — 64 launches per fetch of 64 scans)
    {
        std::vector<uint32_t> list;
        list.reserve((size_t)n * 3);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t q = slots[i];
            const Slot& s = p->slots[q];
            if (s.failed) continue;
            list.push_back(q);
            list.push_back(s.rows);
            list.push_back(s.exhausted ? 1u : 0u);
        }
        if (!list.empty()) {
            VS_HIP(hipMemcpyAsync(p->list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, c->stream));  // (pageable source: staged before the call returns)
            VS_TRY(launch_resort_cursor_batch(ix, (uint32_t)(list.size() / 3), (const uint32_t*)p->list.p, S, k, all_ids, all_dist, all_ham, p->rows_cap,
                                              (uint64_t*)p->resort_heap.p, p->d_cur, p->d_ids, p->d_tids, p->d_dist, p->kmax));
        }
    }
    std::vector<uint64_t> pack((size_t)G * p->kmax * 2 + (size_t)G * 2);  // (u64 words of the layout above)
    VS_HIP(hipMemcpyAsync(pack.data(), p->out_tids.p, pack.size() * 8, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    const uint64_t* const tids = pack.data();
    const uint32_t* const cur = (const uint32_t*)(tids + (size_t)G * p->kmax);
    const uint32_t* const ids = cur + (size_t)G * 4;
    const float* const dist = (const float*)(ids + (size_t)G * p->kmax);
    p->t_resort += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t q = slots[i];
        Slot& s = p->slots[q];
        if (s.failed) {
            out_rows[i] = VS_ERR_CAPACITY;
            continue;
        }
        const uint32_t got = cur[(size_t)q * 4 + 3];
        VS_REQUIRE(cur[(size_t)q * 4 + 2] == s.handed + got && got <= k, "vs_scanpool_fetch: cursor of slot %u out of step", q);
        for (uint32_t j = 0; j < got; ++j) {
            if (out_ids) out_ids[(size_t)i * k + j] = ids[(size_t)q * p->kmax + j];
            if (out_tids) out_tids[(size_t)i * k + j] = tids[(size_t)q * p->kmax + j];
            if (out_dist) out_dist[(size_t)i * k + j] = dist[(size_t)q * p->kmax + j];
        }
        s.handed += got;
        if (got < k) s.calls_after_end += 1;  // the call that found the scan at its end (the executor stops there)
        out_rows[i] = (int32_t)got;
    }
    return prefetch_round(p, slots, n, k);  // the next rows of the scans that are being streamed: searched while these are consumed
}
extern "C" int vs_scanpool_fetch(vs_scan_pool* p, const uint32_t* slots, uint32_t n, uint32_t k, uint64_t* out_tids, uint32_t* out_ids,
                                 float* out_dist, int32_t* out_rows) {
    return vs_guard("vs_scanpool_fetch", [&] { return scanpool_fetch_impl(p, slots, n, k, out_tids, out_ids, out_dist, out_rows); });
}

// GreedySearchStats of one pooled scan as the reference's scan holds them after the amgettuple calls made so far (vs_scan_get_stats,
// vs_api.hip, has the derivation: after j calls the reference has pulled rescore + j - 1 rows out of next(); once the stream has
// ended every further call asks next() once more in vain)
extern "C" int vs_scanpool_get_stats(const vs_scan_pool* p, uint32_t slot, vs_stats* out) {
    VS_REQUIRE(p && out && slot < p->cap, "vs_scanpool_get_stats: bad args");
    const Slot& s = p->slots[slot];
    const vs_index* ix = p->ix;
    vs_stats st{};
    const uint32_t S = p->S;
    const uint64_t calls = (uint64_t)s.handed + s.calls_after_end;
    if (!s.active || calls == 0) {
        *out = st;
        return VS_OK;
    }
    const uint64_t need = S > 0 ? S + calls - 1 : calls;
    const uint32_t* r;
    uint64_t rows_used, next_calls;
    if (need <= s.rows) {
        r = s.row_stats.data() + (size_t)(need - 1) * ST_N;
        rows_used = need;
        next_calls = r[ST_NEXT];
    } else {
        VS_REQUIRE(s.exhausted, "vs_scanpool_get_stats: scan out of step");
        r = s.final_counters;
        rows_used = s.rows;
        const uint64_t first_empty = S > 0 ? (s.rows + 1 > S ? s.rows + 1 - S : 0) + 1 : (uint64_t)s.rows + 1;
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
        const uint64_t nr = rows_used + (s.masked ? r[ST_INVIS] : 0u);
        st.full_distance_comparisons += nr;
        st.node_heap_reads += nr;
    }
    *out = st;
    return VS_OK;
}

// shared launches / rounds since the pool was created (diagnostics: a round serves every scan that asked)
extern "C" int vs_scanpool_get_work(const vs_scan_pool* p, uint64_t* launches, uint64_t* rounds) {
    VS_REQUIRE(p, "vs_scanpool_get_work: pool is NULL");
    if (launches) *launches = p->launches;
    if (rounds) *rounds = p->rounds;
    return VS_OK;
}
