// vs_tune.hip — host side of libvsgpu.so: launch-variant selection (vs_index_autotune, vs_index_set_variant).
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
// Launch-variant selection (include/vsgpu.h: vs_index_autotune).  Every variant is an EXACT instantiation of k_search_fast
// (same rows, same counters); they differ in where a scan keeps its private state, and which of them is fastest depends on the
// index size and the box (docs/LAB_NOTEBOOK.md 11b.13-18) — so it is measured on the caller's own batch, and a variant has to reproduce the
// default's output on that batch bit for bit before it may be chosen.
// ---------------------------------------------------------------------------------------------------------------
struct TuneCand {
    const char* name;
    int virgin, minw;
    uint32_t gcap;
    int lds_max_ins = -1;  // 0: a candidate for indexes whose default is the LDS-table regime (the table-less regime there)
    int vr = -1;           // 0: likewise — the LDS table stays, the visited list moves from registers to the LDS ring
    bool for_lds_regime() const { return lds_max_ins == 0 || vr == 0; }
};
static const TuneCand kTuneCands[] = {
    {"default", -1, -1, 0},
    {"bucket_bitmap", 1, -1, 0},         // no clear, no read of a bucket the scan has not written (11b.16)
    {"bucket_bitmap_16k", 1, -1, 16384}, // ... with a sparser table (more first-touch buckets per probe, more lines)
    {"cleared_tables", 0, -1, 0},        // round 3's default: every scan clears its table, every probe loads a bucket
    {"slot_bitmap", 2, -1, 0},           // round 4's default: 4-byte entries, an occupancy bit per slot, linear probing
    // (the library default in the table-less regime since round 5: 16-BIT entries in buckets of eight with an occupancy bit per slot,
    // VS_F_VIRGIN=3 — where an index's id width does not fit 16-bit remainders the slot bitmap runs instead)
    // (no longer candidates: the two-row gather at 5 waves per SIMD, 2.8-7.3 % slower at 10M / 50M, profiles/r04/s1_ab_virgin_*.txt
    // (VS_F_MINW=5 still selects it by hand).  Deleted: the epoch-tagged tables — exact on hardware in round 4's first session,
    // profiles/r04/s1_fuzz_gpu_epoch*.txt, but no faster than the bitmaps and not compatible with the persistent grid's per-workgroup
    // regions — and the software-pipelined visits, three times slower, profiles/r03/ab_autotune_10m.json)
    // small scans (dedup table in LDS by default: 3-4 times fewer scans per CU): the table-less regime instead, plain and with the
    // bitmap (1M x 768 at search_list_size 3 / rescore 53: -37.7 % / -36.2 %, profiles/r03/ab_autotune_1m.json)
    {"table_less", 0, -1, 0, 0},
    {"table_less_bitmap", 1, -1, 0, 0},
    // ... or the LDS table with the LDS-ring visited list (the register-resident list is what costs the default its occupancy:
    // 141 VGPRs; exact on the interpreter, not timed yet)
    {"lds_table_ring", 0, -1, 0, -1, 0},
};
static const uint32_t kNTuneCands = sizeof(kTuneCands) / sizeof(kTuneCands[0]);

static void tune_apply(vs_index* ix, const TuneCand& c) {
    ix->tune.virgin = c.virgin;
    ix->tune.minw = c.minw;
    ix->tune.gcap = c.gcap;
    ix->tune.lds_max_ins = c.lds_max_ins;
    ix->tune.vr = c.vr;
    snprintf(ix->tune.name, sizeof(ix->tune.name), "%s", c.name);
}

extern "C" int vs_index_set_variant(vs_index* ix, const char* name) {
    VS_REQUIRE(ix && name, "vs_index_set_variant: bad args");
    for (uint32_t i = 0; i < kNTuneCands; ++i)
        if (!strcmp(name, kTuneCands[i].name)) {
            tune_apply(ix, kTuneCands[i]);
            return VS_OK;
        }
    vs_set_error("vs_index_set_variant: unknown variant '%s'", name);
    return VS_ERR_INVALID;
}
extern "C" int vs_index_get_variant(vs_index* ix, char* buf, size_t len) {
    VS_REQUIRE(ix && buf && len, "vs_index_get_variant: bad args");
    snprintf(buf, len, "%s", ix->tune.name);
    return VS_OK;
}

struct TuneRun {
    float step_ms = 0.f, search_ms = 0.f;
    vs_stats st{};
    FastSig sig{};
};

// one step of the caller's batch under the index's current variant: device time of the whole step (events on the ctx stream
// around everything the step enqueues) and of the first-attempt search kernel (the profile spans)
static int tune_step(vs_index* ix, const float* d_q, const int16_t* d_ql, const uint32_t* d_qo, uint32_t nq, uint32_t L,
                     uint32_t rescore, uint32_t k, uint32_t* d_ids, float* d_dist, TuneRun* out) {
    vs_ctx* c = ix->ctx;
    hipEvent_t a = pool_event(c), b = pool_event(c);
    VS_REQUIRE(a && b, "vs_index_autotune: no HIP events");
    vs_profile p;
    VS_TRY(vs_profile_read(c, &p, 1));
    VS_HIP(hipEventRecord(a, c->stream));
    int rc = vs_search_batch_dev_impl(ix, d_q, d_ql, d_qo, nq, L, rescore, k, d_ids, nullptr, d_dist);
    const FastSig sig = ix->last_fast;
    if (rc == VS_OK) {
        (void)hipEventRecord(b, c->stream);
        rc = vs_search_batch_dev_finish_impl(ix, &out->st);
    }
    if (rc != VS_OK) {
        ix->ws.pending = false;
        (void)hipStreamSynchronize(c->stream);
        (void)vs_profile_read(c, &p, 1);
        c->event_pool.push_back(a);
        c->event_pool.push_back(b);
        return rc;
    }
    VS_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    VS_HIP(hipEventElapsedTime(&ms, a, b));
    c->event_pool.push_back(a);
    c->event_pool.push_back(b);
    VS_TRY(vs_profile_read(c, &p, 1));
    out->step_ms = ms;
    out->search_ms = (float)p.ms[PK_SEARCH];
    out->sig = sig;
    return VS_OK;
}

static bool tune_same_counters(const vs_stats& a, const vs_stats& b) {
    return a.queries == b.queries && a.visited_nodes == b.visited_nodes && a.candidate_nodes == b.candidate_nodes &&
           a.quantized_distance_comparisons == b.quantized_distance_comparisons &&
           a.full_distance_comparisons == b.full_distance_comparisons && a.node_reads == b.node_reads &&
           a.node_heap_reads == b.node_heap_reads && a.next_calls == b.next_calls;
}

static int vs_index_autotune_impl(vs_index* ix, const float* d_q, const int16_t* d_ql, const uint32_t* d_qo, uint32_t nq,
                                  uint32_t L, uint32_t rescore, uint32_t k, uint32_t reps, const char* skip,
                                  vs_tune_entry* report, uint32_t report_cap, uint32_t* n_report) {
    VS_REQUIRE(ix && d_q && nq >= 1 && k >= 1, "vs_index_autotune: bad args");
    const std::string skip_list = std::string(",") + (skip ? skip : "") + ",";
    VS_REQUIRE(!ix->ws.pending, "vs_index_autotune: a batch is in flight (vs_search_batch_dev_finish first)");
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    reps = std::min<uint32_t>(std::max<uint32_t>(reps, 1), 16);
    const size_t out_n = (size_t)nq * k;
    DevBuf ids0, dist0, ids1, dist1;
    struct Cleanup {
        DevBuf *a, *b, *c_, *d;
        ~Cleanup() {
            devbuf_free(*a);
            devbuf_free(*b);
            devbuf_free(*c_);
            devbuf_free(*d);
        }
    } cleanup{&ids0, &dist0, &ids1, &dist1};
    VS_TRY(devbuf_reserve(c, ids0, out_n * 4));
    VS_TRY(devbuf_reserve(c, dist0, out_n * 4));
    VS_TRY(devbuf_reserve(c, ids1, out_n * 4));
    VS_TRY(devbuf_reserve(c, dist1, out_n * 4));
    std::vector<uint32_t> h_ids0(out_n), h_ids1(out_n), h_d0(out_n), h_d1(out_n);
    // the caller's profile accumulators are put back afterwards
    vs_profile saved;
    VS_TRY(vs_profile_read(c, &saved, 1));
    const bool was_profiling = c->profiling;
    c->profiling = true;
    const TuneVariant before = ix->tune;
    std::vector<vs_tune_entry> rep(kNTuneCands);
    int rc_all = VS_OK;
    TuneRun base{};
    const bool w24 = (ix->code_stride + 7) / 8 == 3;
#ifdef VS_TEST_HOOKS  // (the interpreter build of the test tier: the named variant's rows are damaged before the comparison)
    const char* const sabotage_opt = vs_opt_get("VS_TUNE_SABOTAGE");  // (the pointer lives until this thread's next lookup: copied)
    const std::string sabotage_s = sabotage_opt ? sabotage_opt : "";
    const char* sabotage = sabotage_opt ? sabotage_s.c_str() : nullptr;
#else
    const char* sabotage = nullptr;
#endif
    for (uint32_t ci = 0; ci < kNTuneCands && rc_all == VS_OK; ++ci) {
        const TuneCand& cand = kTuneCands[ci];
        vs_tune_entry& e = rep[ci];
        memset(&e, 0, sizeof(e));
        snprintf(e.name, sizeof(e.name), "%s", cand.name);
        if (ci > 0) {
            // a variant that cannot be told from the default here is not launched at all
            if (!base.sig.ran) continue;                                           // no LDS-resident kernel for this index
            if ((base.sig.lh != 0) != cand.for_lds_regime()) continue;             // table-less variants / LDS-table regime: the other's candidates
            if (cand.minw >= 0 && !w24) continue;                                  // built for 17..24-word codes only
            if (cand.gcap && cand.gcap <= base.sig.gcap) continue;                 // not sparser than the fitted table
            if (skip_list.find(std::string(",") + cand.name + ",") != std::string::npos) continue;  // the caller's veto
        }
        tune_apply(ix, cand);
        uint32_t* d_ids = (uint32_t*)(ci == 0 ? ids0.p : ids1.p);
        float* d_dist = (float*)(ci == 0 ? dist0.p : dist1.p);
        TuneRun best{};
        int rc = VS_OK;
        // the first step of the default also tells the table fit what a scan of this operating point inserts (ScanObs): two
        // warm-ups there, one for every other variant
        const uint32_t warm = ci == 0 ? 2u : 1u;
        bool have = false;
        for (uint32_t r = 0; r < warm + reps; ++r) {
            TuneRun t{};
            rc = tune_step(ix, d_q, d_ql, d_qo, nq, L, rescore, k, d_ids, d_dist, &t);
            if (rc != VS_OK) break;
            if (r == 0 && ci > 0 && t.sig == base.sig) break;  // launched the default's instantiation: nothing to compare
            if (r >= warm && (!have || t.step_ms < best.step_ms)) {
                best = t;
                have = true;
            }
        }
        if (rc != VS_OK) {
            if (ci == 0) {
                rc_all = rc;  // the default itself fails: the caller's arguments are at fault
                break;
            }
            e.error = rc;
            continue;
        }
        if (!have) continue;  // not applicable (same launch as the default)
        e.applicable = 1;
        e.step_ms = best.step_ms;
        e.search_ms = best.search_ms;
        std::vector<uint32_t>& hi = ci == 0 ? h_ids0 : h_ids1;
        std::vector<uint32_t>& hd = ci == 0 ? h_d0 : h_d1;
        hipError_t he = hipMemcpyAsync(hi.data(), d_ids, out_n * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(hd.data(), d_dist, out_n * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) {
            vs_set_error("vs_index_autotune: %s", hipGetErrorString(he));
            rc_all = VS_ERR_HIP;
            break;
        }
        if (ci == 0) {
            base = best;
            e.rows_identical = 1;
        } else {
            if (sabotage && !strcmp(sabotage, cand.name)) hi[out_n / 2] ^= 1u;
            e.rows_identical = (memcmp(hi.data(), h_ids0.data(), out_n * 4) == 0 && memcmp(hd.data(), h_d0.data(), out_n * 4) == 0 &&
                                tune_same_counters(best.st, base.st))
                                   ? 1u
                                   : 0u;
            if (!e.rows_identical)
                fprintf(stderr, "[libvsgpu] vs_index_autotune: variant '%s' does NOT reproduce the default's rows on this batch — disqualified\n",
                        cand.name);
        }
    }
    uint32_t pick = 0;
    if (rc_all == VS_OK) {
        // the default once more at the end (a box drifts over the seconds this takes): its time is the better of the two
        tune_apply(ix, kTuneCands[0]);
        for (uint32_t r = 0; r < reps; ++r) {
            TuneRun t{};
            if (tune_step(ix, d_q, d_ql, d_qo, nq, L, rescore, k, (uint32_t*)ids1.p, (float*)dist1.p, &t) != VS_OK) break;
            if (t.step_ms < rep[0].step_ms) {
                rep[0].step_ms = t.step_ms;
                rep[0].search_ms = t.search_ms;
            }
        }
        for (uint32_t ci = 1; ci < kNTuneCands; ++ci)
            if (rep[ci].applicable && rep[ci].rows_identical && !rep[ci].error && rep[ci].step_ms < rep[pick].step_ms) pick = ci;
        // A variant replaces the default only when it is at least 3 % faster AND still is when timed a second time: best-of-`reps`
        // times of ONE kernel differ by up to ~1.5 % between two rounds on one box (profiles/r03/ab_autotune_10m.json: a variant that
        // was 4.1 % slower in one session won a 1 % threshold by 1.3 % in the next), so anything inside that band is noise
        if (pick && !(rep[pick].step_ms < 0.97f * rep[0].step_ms)) pick = 0;
        if (pick) {
            tune_apply(ix, kTuneCands[pick]);
            float again = 0.f;
            bool have = false;
            for (uint32_t r = 0; r < reps + 1; ++r) {
                TuneRun t{};
                if (tune_step(ix, d_q, d_ql, d_qo, nq, L, rescore, k, (uint32_t*)ids1.p, (float*)dist1.p, &t) != VS_OK) {
                    have = false;
                    break;
                }
                if (r >= 1 && (!have || t.step_ms < again)) {
                    again = t.step_ms;
                    have = true;
                }
            }
            if (!have || !(again < 0.97f * rep[0].step_ms)) pick = 0;
            else rep[pick].step_ms = std::max(rep[pick].step_ms, again);  // (reported: the slower of its two measurements)
        }
        rep[pick].chosen = 1;
        tune_apply(ix, kTuneCands[pick]);
        // a sparser-table candidate grew the table array for everyone: give it back unless it won (the next launch sizes it anew)
        if (!kTuneCands[pick].gcap) {
            devbuf_free(ix->ws.ghash4);
        }
    } else {
        ix->tune = before;
    }
    c->profiling = was_profiling;
    {
        vs_profile drop;
        (void)vs_profile_read(c, &drop, 1);
        for (int i = 0; i < 8; ++i) {
            c->prof_ms[i] = saved.ms[i];
            c->prof_launches[i] = saved.launches[i];
        }
    }
    if (rc_all != VS_OK) return rc_all;
    if (n_report) *n_report = kNTuneCands;
    if (report)
        for (uint32_t i = 0; i < std::min(report_cap, kNTuneCands); ++i) report[i] = rep[i];
    return VS_OK;
}
extern "C" int vs_index_autotune(vs_index* ix, const float* d_q, const int16_t* d_ql, const uint32_t* d_qo, uint32_t nq, uint32_t L,
                                 uint32_t rescore, uint32_t k, uint32_t reps, const char* skip, vs_tune_entry* report,
                                 uint32_t report_cap, uint32_t* n_report) {
    return vs_guard("vs_index_autotune", [&] { return vs_index_autotune_impl(ix, d_q, d_ql, d_qo, nq, L, rescore, k, reps, skip, report, report_cap, n_report); });
}
