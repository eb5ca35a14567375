// vs_batch.hip — host side of libvsgpu.so: the batched search pipeline behind vs_search_batch / vs_stream_batch / vs_search_batch_dev
// (capacities of a launch, k_search_fast -> retry of the failed scans on the general kernel -> rerank -> resort; host batches cut
// into chunks that overlap staging, search and the copy back).  Split out of vs_api.hip in round 6 (code motion only).
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
// batched scans
// ---------------------------------------------------------------------------------------------------------------
struct Caps {
    uint32_t hl, hcap, vcap, lh, hashcap, g0;  // general kernel (vs_search.hip)
    // fast kernel (vs_search_fast.hip); f_lh == 0: no LDS dedup table (every id in the global table)
    bool f_on;
    uint32_t f_hl, f_hcap, f_gstride, f_lh, f_gcap, f_sb, f_vr, f_vcap;
    double f_pool_frac;  // share of the scans expected to need a global dedup-overflow table
};

// a launch knob: the environment variable when set, else the index's tuned variant (vs_index_autotune), else the default
static uint32_t knob_u32(const char* name, int tuned, uint32_t dflt) {
    const char* v = vs_opt_get(name);
    if (v && *v) return (uint32_t)strtoul(v, nullptr, 10);
    return tuned >= 0 ? (uint32_t)tuned : dflt;
}

static uint32_t gload_pct() { return std::min<uint32_t>(std::max<uint32_t>(env_u32("VS_F_GLOAD_PCT", 75), 25), 90); }

static Caps initial_caps(const vs_index* ix, uint32_t L, uint32_t M) {
    // visits ~ 1.1-2 L before the first row + one per further row; each visit pushes <= R candidates.
    uint64_t visits = 2ull * L + M + 32;
    uint64_t pushes = visits * ix->d.num_neighbors;
    Caps c;
    // general kernel: LDS holds the top `hl` heap positions and the visited list, the rest spills to per-scan global
    // arrays that cost address space only.  Overflows are retried with doubled caps.
    c.hl = env_u32("VS_HL", 1024);
    c.lh = env_u32("VS_LH", 0);
    c.g0 = env_u32("VS_G0", 4096);
    c.hcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pushes, c.hl), 1u << 22);
    c.vcap = (uint32_t)std::min<uint64_t>(3ull * L + M + 64, 1u << 20);
    c.hashcap = std::max<uint32_t>(next_pow2_u32(std::min<uint64_t>(2ull * pushes, 1u << 23)), c.g0);
    // fast kernel: everything in LDS, sized for the typical scan (about 8-10 new candidates per visit, 1.1 L + M
    // visits); the rare scan that outgrows it is re-run by the general kernel.
    // (about 8-10 new candidates per visit at 1M nodes, 1.1 L + M visits); bigger graphs overlap less, so the table is
    // sized from what the previous batches with the same (L, M) actually inserted once that is known.
    const uint64_t typ_visits = (uint64_t)L + L / 4 + M + 16;
    uint64_t typ_ins = typ_visits * std::min<uint64_t>(ix->d.num_neighbors, 16);
    if (ix->obs.valid && ix->obs.L == L && ix->obs.M == M) typ_ins = (uint64_t)(ix->obs.ins_mean * 1.75) + 96;
    // Two operating points.  Small scans (typ_ins up to ~3K ids): the whole dedup table lives in LDS (~10 KB / scan).
    // Large scans: an LDS table for all ids would leave 3-4 scans per CU, and measurements (10M x 768: 148 ms vs 97 ms
    // per 65536 scans) show that occupancy beats on-chip latency there, so the table shrinks to a 256-slot stub, ids go
    // to the per-scan global table (L2 atomics) and the CU holds 16+ scans.
    // (1024 since the end of round 3, 3072 before: at 1M x 768, search_list_size 3 / rescore 53 — about 1 100 inserted ids per scan —
    // the table-less regime runs the search kernel in 35.0 ms per 262 144 scans against 56.2 ms with the table in LDS
    // (profiles/r03/ab_autotune_1m.json): the LDS-table instantiation keeps its visited list in registers, 141 VGPRs, 12 scans per
    // CU against 24.  Below ~500 inserted ids per scan the table is a kilobyte and nothing has been measured: it stays in LDS.)
    const bool lds_table = typ_ins <= knob_u32("VS_F_LDS_MAX_INS", ix->tune.lds_max_ins, 1024);
    c.f_lh = env_u32("VS_F_LH", lds_table ? (uint32_t)round_up_u32((uint32_t)typ_ins, 64) : 0u);
    c.f_pool_frac = !lds_table ? 1.0
                    : (ix->obs.valid && ix->obs.L == L && ix->obs.M == M) ? std::min(1.0, 2.0 * ix->obs.ov_frac + 0.03) : 1.0;
    if (const char* e = vs_opt_get("VS_F_POOL")) c.f_pool_frac = std::min(1.0, std::max(0.01, atof(e)));
    // LDS heap levels: spilling the bottom level to global memory costs every pop / push an L2 round trip, so the heap
    // gets LDS for about 3/4 of the ids a scan inserts (its typical final size) once that is known
    uint32_t hl_auto = 1023;
    if (lds_table && ix->obs.valid && ix->obs.L == L && ix->obs.M == M) {
        const double want = 0.75 * ix->obs.ins_mean;
        hl_auto = want > 2047 ? 4095 : (want > 1023 ? 2047 : 1023);
    }
    // table-less regime: 80 VGPRs (6 waves per SIMD = 24 scans per CU) need 6.6 KB of LDS per scan at most: a 511-entry heap
    // top (measured: 105.2 vs 108.1 ms at 50M against 5 waves with 1023 entries)
    if (!lds_table) hl_auto = 511;
    c.f_hl = env_u32("VS_F_HL", hl_auto);
    const uint32_t want_v = (uint32_t)std::min<uint64_t>((uint64_t)L + L / 2 + 32, 1u << 20);
    // visited list: register resident (8 VGPR pairs) while LDS is the limiter; in the table-less regime registers are,
    // and the LDS ring variant needs 87 VGPRs instead of 141 (5 instead of 3 waves per SIMD)
    c.f_vr = knob_u32("VS_F_VR", ix->tune.vr, (lds_table && want_v <= 512) ? 8 : 0);
    // (sizing the ring from the lists of earlier batches — 21 instead of 18 scans per CU at the reference's default list size — was
    // measured in round 4 and bought nothing: profiles/r04/s6_summary.txt)
    c.f_vcap = c.f_vr ? 512 : round_up_u32(std::max<uint32_t>(env_u32("VS_F_VCAP", 2 * want_v), 64), 64);
    c.f_on = env_u32("VS_FAST", 1) != 0 && ix->d.storage_type != VS_STORAGE_PLAIN;  // the LDS-resident kernels score SBQ codes
    if (c.f_on) {
        if (c.f_lh) c.f_lh = round_up_u32(std::max<uint32_t>(c.f_lh, 256), 4);
        c.f_hl = std::max<uint32_t>(next_pow2_u32(c.f_hl + 1), 64) - 1;
        // overflow table: room for every candidate the worst scan could insert beyond the LDS table
        c.f_gcap = next_pow2_u32(std::min<uint64_t>(std::max<uint64_t>(std::min<uint64_t>(pushes, 4 * typ_ins), 1024), 1u << 22));
        // table-less regime: the tables of the scans in flight (24 per CU x 64 KB = 400 MB at 50M) compete for the 256 MB of
        // Infinity Cache — half the table is 5 % faster, twice the table 10 % slower (profiles/r03/ab_epoch_*.txt) — so once the
        // previous batches with this (L, M) have shown what the largest scan inserts, the table is sized for exactly that
        // (load limit 75 %, a few per cent of slack; a scan that still outgrows it takes the second attempt) instead of the
        // next power of two
        if (!lds_table && ix->obs.valid && ix->obs.L == L && ix->obs.M == M && env_u32("VS_F_GCAP_FIT", 1)) {
            // (load limit: 75 %; VS_F_GLOAD_PCT moves it — a denser table is a smaller cache footprint and longer probe runs)
            const uint64_t need = (uint64_t)((ix->obs.ins_max * 1.04 + 128) * 100.0 / gload_pct()) + 64;
            c.f_gcap = (uint32_t)std::min<uint64_t>(c.f_gcap, std::max<uint64_t>(round_up_u32((uint32_t)std::min<uint64_t>(need, 1u << 22), 256), 1024));
        }
        if (const uint32_t g = env_u32("VS_F_GCAP", lds_table ? 0 : ix->tune.gcap)) c.f_gcap = round_up_u32(std::max<uint32_t>(g, 256), 256);
        c.f_sb = 0;
        while ((1ull << c.f_sb) < (uint64_t)c.f_lh + c.f_gcap) c.f_sb++;
        c.f_hcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pushes, c.f_hl), 1u << 22);
        c.f_gstride = round_up_u32(c.f_hcap - c.f_hl + 2, 2);
        const uint64_t nbits = (uint64_t)ix->d.dim_index * ix->d.bits;
        FastLaunch probe{};
        probe.hl = c.f_hl;
        probe.lh = c.f_lh;
        probe.vr = c.f_vr;
        probe.vcap = c.f_vcap;
        if (nbits >= (1ull << (32 - c.f_sb)) || fast_lds_bytes(ix, probe) > 64 * 1024) c.f_on = false;
    }
    return c;
}

static uint32_t fast_pool_slots(uint32_t nq, double frac) {
    const uint64_t floor_slots = vs_opt_get("VS_F_POOL") ? 1 : 256;  // (the override exists to exercise pool exhaustion in tests)
    return (uint32_t)std::min<uint64_t>(nq, std::max<uint64_t>(floor_slots, (uint64_t)(frac * nq) + 1));
}
static uint32_t general_pool_slots(uint32_t nq) { return std::max<uint32_t>(64, nq / 64); }

static bool grow_caps(Caps& c, uint32_t ovf) {
    bool grew = (ovf & (OVF_POOL | OVF_KEY)) != 0;  // pool exhausted / wide label key: the relaunch (general kernel) takes them
    if ((ovf & OVF_HEAP) && c.hcap < (1u << 24)) {
        c.hcap *= 2;
        grew = true;
    }
    if (ovf & OVF_VISITED) {
        c.vcap *= 2;
        grew = true;
    }
    if ((ovf & OVF_HASH) && c.hashcap < (1u << 26)) {
        c.hashcap *= 2;
        grew = true;
    }
    return grew;
}

// runs prepare -> search (-> rerank -> resort) for nq queries already on the device.  Outputs land in the workspace
// (or the caller's device buffers).  Synchronous w.r.t. overflow retries when `allow_sync` is set.
struct BatchPlan {
    uint32_t nq, L, rescore, k, M;
    bool stream_only;  // vs_stream_batch: no rerank
};

// rerank + rescore window over the streams the search kernels left in the workspace
struct PendingBatch {
    BatchPlan bp;
    Caps caps;
    const int16_t* d_qlabels;
    const uint32_t* d_qlabel_off;
    uint32_t* d_out_ids;
    uint64_t* d_out_tids;
    float* d_out_dist;
};

static int run_post_search(vs_index* ix, const BatchPlan& bp, uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist) {
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    const uint32_t nq = bp.nq, M = bp.M;
    if (bp.stream_only) return VS_OK;
    if (bp.rescore > 0) {
        VS_REQUIRE(ix->vecs, "diskann.query_rescore > 0 needs the heap vector column on the device");
        VS_TRY(devbuf_reserve(c, w.rr_dist, (size_t)nq * M * 4));
        VS_TRY(devbuf_reserve(c, w.resort_heap, (size_t)nq * bp.rescore * 8));
        hipEvent_t ev = prof_begin(c);
        VS_TRY(launch_rerank(ix, (const float*)w.q_full.p, (const uint32_t*)w.stream_ids.p, nullptr,
                             (const uint32_t*)w.stream_cnt.p, M, nq, (float*)w.rr_dist.p));
        prof_end(c, PK_RERANK, ev);
    }
    hipEvent_t ev = prof_begin(c);
    VS_TRY(launch_resort(ix, nq, M, bp.rescore, bp.k, (const uint32_t*)w.stream_ids.p, (const uint32_t*)w.stream_cnt.p,
                         bp.rescore ? (const float*)w.rr_dist.p : nullptr, (uint64_t*)w.resort_heap.p, d_out_ids,
                         d_out_tids, d_out_dist));
    prof_end(c, PK_RESORT, ev);
    return VS_OK;
}

// (re)runs the general kernel over the scans whose status is non-zero until none is left; synchronises the stream
static int retry_failed_scans(vs_index* ix, const BatchPlan& bp, const int16_t* d_qlabels, const uint32_t* d_qlabel_off,
                              Caps& caps, vs_stats* st) {
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    const uint32_t nq = bp.nq, M = bp.M;
    std::vector<uint32_t> status(nq);
    for (int attempt = 0;; ++attempt) {
        VS_HIP(hipMemcpyAsync(status.data(), w.status.p, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
        uint32_t ovf = 0, nbad = 0;
        for (uint32_t v : status) {
            ovf |= v;
            nbad += v != 0;
        }
        if (!ovf) return VS_OK;
        if (st) st->retries++;
        if (attempt >= 8 || !grow_caps(caps, ovf)) {
            vs_set_error("search structures overflowed in %u of %u scans (flags 0x%x) at hcap=%u vcap=%u hashcap=%u", nbad, nq,
                         ovf, caps.hcap, caps.vcap, caps.hashcap);
            return VS_ERR_CAPACITY;
        }
        const size_t hg = caps.hcap > caps.hl ? caps.hcap - caps.hl : 0;
        const uint32_t gslots = std::min<uint32_t>(nq, std::max<uint32_t>(general_pool_slots(nq), nbad));
        VS_TRY(devbuf_reserve(c, w.hash, (size_t)gslots * caps.hashcap * 4));
        VS_TRY(devbuf_reserve(c, w.heap_g, std::max<size_t>((size_t)gslots * hg * 8, 16)));
        VS_TRY(devbuf_reserve(c, w.pool_ctr, 64));
        VS_HIP(hipMemsetAsync((char*)w.pool_ctr.p + 32, 0, 4, c->stream));
        SearchLaunch s;
        s.nq = nq;
        s.L = bp.L;
        s.M = M;
        s.hl = caps.hl;
        s.hcap = caps.hcap;
        s.vcap = caps.vcap;
        s.lh = caps.lh;
        s.hashcap = caps.hashcap;
        s.g0 = caps.g0;
        s.qcodes = (const uint64_t*)w.qcodes.p;
        s.qlabels = d_qlabels;
        s.qlabel_off = d_qlabel_off;
        s.heap_g = (uint64_t*)w.heap_g.p;
        s.hash = (uint32_t*)w.hash.p;
        s.out_ids = (uint32_t*)w.stream_ids.p;
        s.out_ham = (uint32_t*)w.stream_ham.p;
        s.out_cnt = (uint32_t*)w.stream_cnt.p;
        s.stats = (uint32_t*)w.stats.p;
        s.status = (uint32_t*)w.status.p;
        s.only_failed = 1;
        s.fb_flag = w.fb_valid ? (uint32_t*)w.fb_flag.p : nullptr;
        s.pool_counter = (uint32_t*)((char*)w.pool_ctr.p + 32);
        s.pool_slots = gslots;
        s.visible = (!bp.stream_only && bp.rescore > 0) ? ix->visible : nullptr;
        hipEvent_t ev = prof_begin(c);
        VS_TRY(launch_search(ix, s));
        prof_end(c, PK_SEARCH_FB, ev);
    }
}

static int run_search_chunk(vs_index* ix, const BatchPlan& bp, const float* d_raw_q, const int16_t* d_qlabels,
                            const uint32_t* d_qlabel_off, uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist,
                            Caps& caps, bool check_now, vs_stats* st) {
    vs_ctx* c = ix->ctx;
    SearchWorkspace& w = ix->ws;
    const uint32_t nq = bp.nq, M = bp.M;
    VS_TRY(devbuf_reserve(c, w.q_full, (size_t)nq * ix->vec_stride * 4));
    VS_TRY(devbuf_reserve(c, w.qcodes, (size_t)nq * ix->code_stride * 8));
    VS_TRY(devbuf_reserve(c, w.stream_ids, (size_t)nq * M * 4));
    VS_TRY(devbuf_reserve(c, w.stream_ham, (size_t)nq * M * 4));
    VS_TRY(devbuf_reserve(c, w.stream_cnt, (size_t)nq * 4));
    VS_TRY(devbuf_reserve(c, w.stats, (size_t)nq * ST_N * 4));
    VS_TRY(devbuf_reserve(c, w.status, (size_t)nq * 4));
    {
        hipEvent_t ev = prof_begin(c);
        VS_TRY(launch_prepare_queries(ix, d_raw_q, nq, (float*)w.q_full.p, (uint64_t*)w.qcodes.p));
        if (ix->d.storage_type == VS_STORAGE_PLAIN && ix->d.dim_index < ix->d.dim_full) {
            VS_TRY(devbuf_reserve(c, w.q_index, (size_t)nq * ix->vec_stride * 4));
            VS_TRY(launch_prepare_index_slice(ix, d_raw_q, nq, (float*)w.q_index.p));
        }
        prof_end(c, PK_PREPARE, ev);
    }
    bool fast_done = false;
    ix->last_fast = FastSig{};
    if (caps.f_on) {
        uint32_t fslots = fast_pool_slots(nq, caps.f_pool_frac);
        ix->last_ins_limit = caps.f_lh ? caps.f_lh - caps.f_lh / 8 - 64 : 0xFFFFFFFFu;
        // Persistent grid (VS_F_PERSIST, default on): as many single-wave workgroups as the device holds at once, each taking scan
        // after scan from a counter and reusing ITS region of the heap spill array and of the dedup tables — the workspace is
        // (resident scans) x (region) instead of nq x (region): 0.6 GB instead of 26 GB for 262 144 scans of the 50M index
        FastLaunch f;
        f.nq = nq;
        f.L = bp.L;
        f.M = M;
        f.hl = caps.f_hl;
        f.hcap = caps.f_hcap;
        f.gstride = caps.f_gstride;
        f.vr = caps.f_vr;
        f.gcap = caps.f_gcap;
        f.glimit = (uint32_t)((uint64_t)caps.f_gcap * gload_pct() / 100) - 64u;
        f.lh = caps.f_lh;
        f.minw = knob_u32("VS_F_MINW", (caps.f_lh == 0 && !caps.f_vr) ? ix->tune.minw : -1, caps.f_lh == 0 ? (caps.f_vr ? 4 : 6) : 1);
        f.flags = env_u32("VS_F_FLAGS", 0);
        f.sb = caps.f_sb;
        f.vcap = caps.f_vcap;
        f.qlabels = d_qlabels;
        f.qlabel_off = d_qlabel_off;
        f.visible = (!bp.stream_only && bp.rescore > 0) ? ix->visible : nullptr;  // the heap is only fetched for the rescore window
        f.rc = caps.f_lh == 0 ? env_u32("VS_F_RC", 0) : 0;  // (measurement: LDS id cache in front of the dedup table in HBM)
        if (f.rc) f.rc = next_pow2_u32(f.rc);
        // written-bucket bitmap (VS_F_VIRGIN=1, table-less regime): 128 slots of the table per LDS word; tables of more than
        // 64 Ki slots keep the clear (the bitmap would cost occupancy)
        // ... or (VS_F_VIRGIN=2) one bit per SLOT: linear probing at slot granularity with the occupancy known on chip, so most new
        // ids are stored without a load of the table; 32 slots per LDS word — taken only while it costs no scans per CU (else the
        // bucket bitmap runs)
        // Default since round 4's third GPU session: the slot bitmap — 161.1 ms per 262 144 scans at 50M against 167.9 with the bucket
        // bitmap and 171.2 with cleared tables, 125.8 / 129.7 / 130.1 at 10M (profiles/r04/s3_ab_slotmap_*.txt); 639 device fuzz cases.
        // Default since round 5: the 16-bit tables below (VS_F_VIRGIN=3) — 139.7 ms per 262 144 scans at 50M against 153.5 with the 4-byte
        // slot-bitmap tables, same session, same slab (profiles/r05/s10_ab_q16_50m.txt); 300 device fuzz runs, regimes green on hardware.
        const uint32_t vmode = knob_u32("VS_F_VIRGIN", ix->tune.virgin, 3);
        bool slot_eligible = false;  // one occupancy bit per slot is possible for this table (whether or not the 4-byte slot bitmap is taken)
        if (caps.f_lh == 0 && !f.vr && vmode && !env_u32("VS_PHASE", 0) && f.gcap <= (1u << 16)) {
            f.vwords = (f.gcap + 127) / 128;
            if (vmode >= 2 && !f.rc && f.gcap % 32 == 0) {
                slot_eligible = true;
                FastLaunch g = f;
                g.vwords = f.gcap / 32;
                g.vslot = 1;
                uint32_t res_b = 0, res_s = 0;
                VS_TRY(fast_resident_scans(ix, f, &res_b));
                VS_TRY(fast_resident_scans(ix, g, &res_s));
                if (res_s >= res_b || env_u32("VS_F_SLOTMAP_FORCE", 0)) {
                    f.vwords = g.vwords;
                    f.vslot = 1;
                }
            }
        }
        // ... or (VS_F_VIRGIN=3) 16-BIT entries: buckets of eight slots (one 16-byte load), the entry is the remainder of a bijective
        // hash of the node id given its bucket (quotienting), a small overflow table of whole ids behind the buckets.  Half the bytes
        // per slot: the tables of the scans in flight are the largest part of the kernel's hot private state (fast_scan, VG == 3).
        // Needs a power-of-two number of buckets and ceil(log2 n) - log2(buckets) <= 16 remainder bits.
        // Round 6 (profiles/r06/s20-s22): the rule used to be "only while it costs no scans per CU", against the 4-byte slot bitmap, which
        // itself had to cost none against the bucket bitmap — and at search_list_size 100 (a visited ring of 3 KB per scan) it does:
        // 19 resident scans per CU against 23.  So the reference's default GUCs, the label-filtered configuration and every other long
        // list ran round 3's bucket-bitmap tables and none of the round-5 / round-6 kernel work (rocprofv3 names the instantiation:
        // k_search_fast<3,0,false,6,false,*,1,0>).  Measured at 10M, 100 / 50: 100.8 ms per 262 144 scans (bucket bitmap, 23 per CU)
        // against 90.4 (16-bit tables, 19 per CU) and 86.2 (16-bit tables with the heap top below, 22 per CU); label keys at 100 / 90:
        // 119.1 / 115.5 / 110.6.  The 16-bit tables are now taken while they keep at least four fifths of the resident scans of whatever
        // the rules above chose — two thirds for label-filtered scans (s24, 10M x 1536, 100 / 90: 110.2 ms at 15 scans per CU against 119.0
        // at 22).  Not below that for unfiltered scans: the `mid` corpus at 100 / 592 (tables of 32 Ki slots, 15 against 21 per CU, the
        // heap spill arrays carrying most of the traffic) runs 436.6 ms with them against 382.2 without (s24).
        uint32_t gregion = f.gcap;
        if (vmode == 3 && f.vwords && slot_eligible && caps.f_lh == 0) {
            uint32_t qd = 1;
            while ((1ull << qd) < (uint64_t)std::max<uint32_t>(ix->d.n, 2)) qd++;
            const uint32_t gcap16 = std::max<uint32_t>(next_pow2_u32(f.gcap), 1024);
            uint32_t lb = 0;
            while ((1u << lb) < (gcap16 >> 3)) lb++;
            if (qd < lb + 3) qd = lb + 3;  // (a small index: more hash bits than id bits — the bijection works on any width)
            const uint32_t qk = qd - lb;
            FastLaunch g = f;
            g.gcap = gcap16;
            g.ocap = std::max<uint32_t>(round_up_u32(gcap16 / 16, 32), 256);
            g.vwords = (g.gcap + g.ocap) / 32;
            g.vslot = 2;
            g.sb = 0;
            while ((1ull << g.sb) < (uint64_t)g.gcap + g.ocap) g.sb++;
            const uint64_t nbits = (uint64_t)ix->d.dim_index * ix->d.bits;
            uint32_t res_s = 0, res_q = 0;
            if (qk <= 16 && qd <= 32 && nbits < (1ull << (32 - g.sb))) {
                g.qd = qd;
                g.qk = qk;
                g.gregion = (g.gcap >> 1) + g.ocap;
                g.glimit = (uint32_t)((uint64_t)g.gcap * gload_pct() / 100) - 64u;
                VS_TRY(fast_resident_scans(ix, f, &res_s));
                VS_TRY(fast_resident_scans(ix, g, &res_q));
                if (env_u32("VS_WS_DEBUG", 0))
                    fprintf(stderr, "[VS_WS_DEBUG] resident scans: 4-byte tables %u, 16-bit tables %u; gcap %u -> %u\n", res_s, res_q, f.gcap, g.gcap);
                // (label-filtered scans mark ~50 ids per ~9 scored rows, AM/sbq/storage.rs:148-172: the table is most of what they touch)
                const bool keyed = f.qlabel_off != nullptr;
                if ((keyed ? 3ull * res_q >= 2ull * res_s : 5ull * res_q >= 4ull * res_s) || env_u32("VS_F_SLOTMAP_FORCE", 0)) {
                    f = g;
                    gregion = g.gregion;
                }
            }
        }
        // LDS-bound launches: where the LDS per scan — not the register cap of the instantiation — limits the resident scans, a heap top
        // of 255 entries instead of 511 (1 KB less) is taken when it buys at least a tenth more of them (s22: 90.4 -> 86.2 ms at 100 / 50,
        // 115.5 -> 110.6 with label keys, 22 instead of 19 per CU; where the registers are the limit — search_list_size 3, 24 per CU —
        // nothing changes: there a smaller heap top only costs).  An explicit VS_F_HL stands.
        if (caps.f_lh == 0 && !f.vr && f.hl == 511 && f.hcap > 511 && !env_u32("VS_PHASE", 0)) {
            const char* const hl_opt = vs_opt_get("VS_F_HL");
            if (!(hl_opt && *hl_opt)) {
                FastLaunch h = f;
                h.hl = 255;
                h.gstride = round_up_u32(h.hcap - h.hl + 2, 2);
                uint32_t res_511 = 0, res_255 = 0;
                VS_TRY(fast_resident_scans(ix, f, &res_511));
                VS_TRY(fast_resident_scans(ix, h, &res_255));
                if (env_u32("VS_WS_DEBUG", 0)) fprintf(stderr, "[VS_WS_DEBUG] resident scans: heap top 511 %u, 255 %u\n", res_511, res_255);
                if (10ull * res_255 >= 11ull * res_511) {
                    f = h;
                    caps.f_hl = h.hl;
                    caps.f_gstride = h.gstride;
                }
            }
        }
        // ... and likewise the visited ring: room for 1.4 instead of 2 times the list's expected length (1 KB less at search_list_size 100)
        // where that buys at least a tenth more resident scans.  The scans that outgrow the smaller ring are finished by the second
        // attempt below, a few milliseconds on the critical path — worth it for label-filtered scans (s21 / s22, 10M x 1536, 100 / 90: 98 of
        // 262 144 scans, 3.5 ms, for a first attempt of 101.5 instead of 110.6 ms: 17 -> 19 scans per CU), a wash for unfiltered ones at
        // 100 / 50 (83.5 + 1.8 against 86.2 ms; 22 -> 24 per CU is under the threshold).  An explicit VS_F_VCAP stands.
        if (caps.f_lh == 0 && !f.vr && f.minw != 7 && !env_u32("VS_PHASE", 0)) {
            const char* const vc_opt = vs_opt_get("VS_F_VCAP");
            const uint32_t want_v = (uint32_t)std::min<uint64_t>((uint64_t)bp.L + bp.L / 2 + 32, 1u << 20);
            const uint32_t lean_v = round_up_u32(std::max<uint32_t>(want_v + 2 * want_v / 5, 64), 64);
            if (!(vc_opt && *vc_opt) && lean_v < f.vcap) {
                FastLaunch v = f;
                v.vcap = lean_v;
                uint32_t res_wide = 0, res_lean = 0;
                VS_TRY(fast_resident_scans(ix, f, &res_wide));
                VS_TRY(fast_resident_scans(ix, v, &res_lean));
                if (env_u32("VS_WS_DEBUG", 0))
                    fprintf(stderr, "[VS_WS_DEBUG] resident scans: visited ring %u entries %u, %u entries %u\n", f.vcap, res_wide, v.vcap, res_lean);
                if (10ull * res_lean >= 11ull * res_wide) f = v;
            }
        }
        // (VS_F_MINW=7 with the 16-bit tables: 28 scans per CU when a scan's LDS fits 5 632 B — the visited ring is then sized in steps
        // of 16 entries instead of 64)
        if (f.minw == 7 && f.vslot == 2 && !f.vr && !env_u32("VS_F_VCAP", 0)) {
            const uint32_t want_v = (uint32_t)std::min<uint64_t>((uint64_t)bp.L + bp.L / 2 + 32, 1u << 20);
            f.vcap = round_up_u32(std::max<uint32_t>(2 * want_v, 64), 16);
        }
        if (env_u32("VS_PHASE", 0)) f.phase = (uint64_t*)16;  // (selects the instantiation; the buffer is set below)
        if (knob_u32("VS_F_PERSIST", ix->tune.persist, 1)) {
            uint32_t res = 0;
            VS_TRY(fast_resident_scans(ix, f, &res));
            f.persist = std::max<uint32_t>(1, (uint32_t)((uint64_t)res * env_u32("VS_F_PERSIST_PCT", 100) / 100));
            fslots = std::min(f.persist, nq);
        }
        // (persistent grid: the two randomly accessed arrays live in the index's slab, dedup tables first)
        if (f.persist) {
            const uint32_t what = env_u32("VS_WS_SLAB_WHAT", 3);  // (measurement: 1 = only the dedup tables, 2 = only the heap spill arrays)
            if (what & 1) VS_TRY(devbuf_reserve_hot(ix, w.ghash4, (size_t)fslots * gregion * 4, 0));
            else VS_TRY(devbuf_reserve(c, w.ghash4, (size_t)fslots * gregion * 4));
            if (what & 2) VS_TRY(devbuf_reserve_hot(ix, w.heap_g4, std::max<size_t>((size_t)fslots * caps.f_gstride * 4, 16), 1));
            else VS_TRY(devbuf_reserve(c, w.heap_g4, std::max<size_t>((size_t)fslots * caps.f_gstride * 4, 16)));
        } else {
            VS_TRY(devbuf_reserve(c, w.heap_g4, std::max<size_t>((size_t)nq * caps.f_gstride * 4, 16)));
            VS_TRY(devbuf_reserve(c, w.ghash4, (size_t)fslots * gregion * 4));
        }
        if (env_u32("VS_WS_DEBUG", 0))  // diagnostics: where the hot arrays live (scripts/diag_state.py --placement)
            fprintf(stderr, "[VS_WS_DEBUG] ghash4 %p (%zu B%s) heap_g4 %p (%zu B%s) region bytes: table %zu heap %zu x %u regions; stream_ids %p qcodes %p\n", w.ghash4.p,
                    w.ghash4.bytes, w.ghash4.in_slab ? ", slab" : "", w.heap_g4.p, w.heap_g4.bytes, w.heap_g4.in_slab ? ", slab" : "",
                    (size_t)gregion * 4, (size_t)caps.f_gstride * 4, fslots, w.stream_ids.p, w.qcodes.p);
        VS_TRY(devbuf_reserve(c, w.pool_ctr, 64));
        VS_HIP(hipMemsetAsync(w.pool_ctr.p, 0, 64, c->stream));
        VS_TRY(devbuf_reserve(c, w.fb_flag, (size_t)nq * 4));
        VS_HIP(hipMemsetAsync(w.fb_flag.p, 0, (size_t)nq * 4, c->stream));
        f.heap_g = (uint32_t*)w.heap_g4.p;
        f.ghash = (uint32_t*)w.ghash4.p;
        f.pool_counter = (uint32_t*)w.pool_ctr.p;
        f.scan_counter = (uint32_t*)w.pool_ctr.p + 2;
        f.pool_slots = fslots;
        f.phase = nullptr;
        f.qcodes = (const uint64_t*)w.qcodes.p;
        f.out_ids = (uint32_t*)w.stream_ids.p;
        f.out_ham = (uint32_t*)w.stream_ham.p;
        f.out_cnt = (uint32_t*)w.stream_cnt.p;
        f.stats = (uint32_t*)w.stats.p;
        f.status = (uint32_t*)w.status.p;
        if (env_u32("VS_PHASE", 0)) {
            VS_TRY(devbuf_reserve(c, w.phase, (size_t)nq * 64));
            VS_HIP(hipMemsetAsync(w.phase.p, 0, (size_t)nq * 64, c->stream));
            f.phase = (uint64_t*)w.phase.p;
        }
        const char* const tl_opt = vs_opt_get("VS_TIMELINE");  // diagnostics: start / end of every scan of this launch, dumped to a file
        const std::string tl_s = tl_opt ? tl_opt : "";  // (the option's pointer lives until this thread's next lookup)
        const char* const tl_path = tl_s.c_str();
        if (*tl_path) {
            VS_TRY(devbuf_reserve(c, w.timeline, (size_t)nq * 16));
            VS_HIP(hipMemsetAsync(w.timeline.p, 0, (size_t)nq * 16, c->stream));
            f.timeline = (uint64_t*)w.timeline.p;
        }
        hipEvent_t ev = prof_begin(c);
        VS_TRY(launch_search_fast(ix, f));
        prof_end(c, PK_SEARCH, ev);
        fast_done = true;
        if (f.timeline) {
            std::vector<uint64_t> tl((size_t)nq * 2);
            VS_HIP(hipMemcpyAsync(tl.data(), w.timeline.p, tl.size() * 8, hipMemcpyDeviceToHost, c->stream));
            VS_HIP(hipStreamSynchronize(c->stream));
            if (FILE* fp = fopen(tl_path, "wb")) {
                fwrite(tl.data(), 8, tl.size(), fp);
                fclose(fp);
            }
        }
        ix->last_fast = FastSig{f.vwords, f.minw, f.gcap, f.lh, f.vr, 1u};
        // second attempt of the scans that outgrew these capacities (a handful per launch at the tail of the distribution):
        // the same kernel with a four times larger dedup table, twice the heap and visited-list room, regions from a small
        // pool.  Scans finished above return at once; what still does not fit goes to the general kernel below.
        if (env_u32("VS_F_RETRY", 1)) {
            FastLaunch r = f;
            r.vwords = 0;  // (its own, smaller table array: cleared by the few scans that run)
            r.vslot = 0;
            r.persist = 0;  // (one workgroup per scan: nearly all of them return at once; regions from the pool)
            r.timeline = nullptr;
            r.only_failed = 1;
            r.fb_flag = (uint32_t*)w.fb_flag.p;
            r.phase = nullptr;
            r.gcap = (uint32_t)std::min<uint64_t>(4ull * f.gcap, 1u << 22);
            r.glimit = 0;  // (75 % of the larger table)
            r.hcap = (uint32_t)std::min<uint64_t>(2ull * f.hcap, 1u << 22);
            r.gstride = round_up_u32(r.hcap - r.hl + 2, 2);
            if (!r.vr) r.vcap = 2 * f.vcap;
            r.sb = 0;
            while ((1ull << r.sb) < (uint64_t)r.lh + r.gcap) r.sb++;
            r.pool_slots = general_pool_slots(nq);
            r.pool_counter = (uint32_t*)((char*)w.pool_ctr.p + 16);
            const uint64_t nbits = (uint64_t)ix->d.dim_index * ix->d.bits;
            if (nbits < (1ull << (32 - r.sb)) && fast_lds_bytes(ix, r) <= 64 * 1024) {
                VS_TRY(devbuf_reserve(c, w.heap_g4b, (size_t)r.pool_slots * r.gstride * 4));
                VS_TRY(devbuf_reserve(c, w.ghash4b, (size_t)r.pool_slots * r.gcap * 4));
                r.heap_g = (uint32_t*)w.heap_g4b.p;
                r.ghash = (uint32_t*)w.ghash4b.p;
                hipEvent_t ev2 = prof_begin(c);
                VS_TRY(launch_search_fast(ix, r));
                prof_end(c, PK_SEARCH_FB, ev2);
            }
        }
        if (env_u32("VS_DEBUG_STATUS", 0)) {  // diagnostics: which flags did the fast kernel leave behind?
            std::vector<uint32_t> stv(nq);
            VS_HIP(hipMemcpyAsync(stv.data(), w.status.p, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
            uint32_t ctr[2] = {0, 0};
            VS_HIP(hipMemcpyAsync(ctr, w.pool_ctr.p, 4, hipMemcpyDeviceToHost, c->stream));
            VS_HIP(hipStreamSynchronize(c->stream));
            uint32_t hist[16] = {0};
            for (uint32_t v : stv) hist[v & 15]++;
            fprintf(stderr, "[VS_DEBUG_STATUS] fast kernel: lh=%u gcap=%u vr=%u minw=%u bitmap_words=%u (per %s); pool claims=%u of %u;",
                    f.lh, f.gcap, f.vr, f.minw, f.vwords, f.vslot == 2 ? "slot, 16-bit entries" : f.vslot ? "slot" : "bucket", ctr[0], fast_pool_slots(nq, caps.f_pool_frac));
            for (int i = 0; i < 16; ++i)
                if (hist[i]) fprintf(stderr, " status[%d]=%u", i, hist[i]);
            fprintf(stderr, "\n");
        }
    }
    w.fb_valid = fast_done;
    for (int attempt = 0;; ++attempt) {
        const size_t hg = caps.hcap > caps.hl ? caps.hcap - caps.hl : 0;
        // after the fast kernel only a few scans are left: they claim their regions from a small pool
        const uint32_t gslots = fast_done ? general_pool_slots(nq) : nq;
        VS_TRY(devbuf_reserve(c, w.hash, (size_t)gslots * caps.hashcap * 4));
        VS_TRY(devbuf_reserve(c, w.heap_g, std::max<size_t>((size_t)gslots * hg * 8, 16)));
        VS_TRY(devbuf_reserve(c, w.pool_ctr, 64));
        if (fast_done) VS_HIP(hipMemsetAsync((char*)w.pool_ctr.p + 32, 0, 4, c->stream));
        SearchLaunch s;
        s.nq = nq;
        s.L = bp.L;
        s.M = M;
        s.hl = caps.hl;
        s.hcap = caps.hcap;
        s.vcap = caps.vcap;
        s.lh = caps.lh;
        s.hashcap = caps.hashcap;
        s.g0 = caps.g0;
        s.qcodes = (const uint64_t*)w.qcodes.p;
        s.qlabels = d_qlabels;
        s.qlabel_off = d_qlabel_off;
        s.heap_g = (uint64_t*)w.heap_g.p;
        s.hash = (uint32_t*)w.hash.p;
        s.out_ids = (uint32_t*)w.stream_ids.p;
        s.out_ham = (uint32_t*)w.stream_ham.p;
        s.out_cnt = (uint32_t*)w.stream_cnt.p;
        s.stats = (uint32_t*)w.stats.p;
        s.status = (uint32_t*)w.status.p;
        // after the fast kernel (or a failed attempt) only the scans whose status is non-zero are (re)run
        s.only_failed = (fast_done || attempt > 0) ? 1u : 0u;
        s.fb_flag = fast_done ? (uint32_t*)w.fb_flag.p : nullptr;
        s.pool_counter = fast_done ? (uint32_t*)((char*)w.pool_ctr.p + 32) : nullptr;
        s.pool_slots = gslots;
        s.visible = (!bp.stream_only && bp.rescore > 0) ? ix->visible : nullptr;
        {
            hipEvent_t ev = prof_begin(c);
            VS_TRY(launch_search(ix, s));
            prof_end(c, fast_done ? PK_SEARCH_FB : PK_SEARCH, ev);
        }
        break;
    }
    if (check_now) VS_TRY(retry_failed_scans(ix, bp, d_qlabels, d_qlabel_off, caps, st));
    return run_post_search(ix, bp, d_out_ids, d_out_tids, d_out_dist);
}

static int collect_stats(vs_index* ix, uint32_t nq, uint32_t M, uint32_t rescore, bool stream_only, vs_stats* st,
                         uint32_t obs_L = 0) {
    if (!st) return VS_OK;
    SearchWorkspace& w = ix->ws;
    std::vector<uint32_t> hs((size_t)nq * ST_N), cnt(nq), fb(nq, 0);
    VS_HIP(hipMemcpyAsync(hs.data(), w.stats.p, hs.size() * 4, hipMemcpyDeviceToHost, ix->ctx->stream));
    if (w.fb_valid) VS_HIP(hipMemcpyAsync(fb.data(), w.fb_flag.p, fb.size() * 4, hipMemcpyDeviceToHost, ix->ctx->stream));
    VS_HIP(hipMemcpyAsync(cnt.data(), w.stream_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, ix->ctx->stream));
    VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    if (env_u32("VS_PHASE", 0) && w.phase.p && w.fb_valid) {
        std::vector<uint64_t> ph((size_t)nq * 8);
        VS_HIP(hipMemcpy(ph.data(), w.phase.p, ph.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0};
        uint64_t visits = 0;
        for (uint32_t q = 0; q < nq; ++q) {
            for (int k = 0; k < 8; ++k) sum[k] += (double)ph[(size_t)q * 8 + k];
            visits += hs[(size_t)q * ST_N + ST_VISITS];
        }
        const char* names[8] = {"pop", "row_wait", "visited", "dedup", "gather", "push", "other", "-"};
        fprintf(stderr, "[VS_PHASE] shader clocks per visit:");
        for (int k = 0; k < 7; ++k) fprintf(stderr, " %s=%.0f", names[k], sum[k] / (double)std::max<uint64_t>(visits, 1));
        fprintf(stderr, "\n");
    }
    if (w.fb_valid && ix->last_ins_limit) {  // what this batch needed: sizes the next launch with the same (L, M)
        double sum = 0, mx = 0;
        uint32_t cnt_fast = 0, ov = 0;
        for (uint32_t q = 0; q < nq; ++q) {
            const double v = hs[(size_t)q * ST_N + 7];
            if (fb[q]) {  // (finished by a second attempt: its insert count still tells how big a table the batch needs)
                ov++;
                mx = std::max(mx, v);
                continue;
            }
            sum += v;
            mx = std::max(mx, v);
            cnt_fast++;
            ov += v > ix->last_ins_limit;
        }
        if (cnt_fast) {
            ScanObs& o = ix->obs;
            const bool same = o.valid && o.L == obs_L && o.M == M;
            const double a = same ? 0.5 : 1.0;  // exponential average over batches
            o.ins_mean = (1 - a) * o.ins_mean + a * (sum / cnt_fast);
            o.ins_max = same ? std::max(o.ins_max, mx) : mx;
            o.ov_frac = (1 - a) * (same ? o.ov_frac : 0.0) + a * ((double)ov / nq);
            o.L = obs_L;
            o.M = M;
            o.valid = true;
        }
    }
    for (uint32_t q = 0; q < nq; ++q) {
        st->queries++;
        st->visited_nodes += hs[(size_t)q * ST_N + ST_VISITS];
        st->candidate_nodes += hs[(size_t)q * ST_N + ST_CAND];
        if (ix->d.storage_type == VS_STORAGE_PLAIN) st->full_distance_comparisons += hs[(size_t)q * ST_N + ST_DQ];
        else st->quantized_distance_comparisons += hs[(size_t)q * ST_N + ST_DQ];
        st->node_reads += hs[(size_t)q * ST_N + ST_READS];
        uint64_t next_calls = hs[(size_t)q * ST_N + ST_NEXT];
        if (!stream_only && rescore > 0 && cnt[q] < M && next_calls > 0) {
            // an exhausted stream under next_with_resort (AM/scan.rs:244-305): every amgettuple call that finds the window short asks
            // `next` once more and gets None again.  The batch stands for min(k, rows + 1) calls (the executor stops at the first call
            // without a row); the first call that runs into the end is number max(1, rows - rescore + 2)
            const int64_t C = cnt[q], S = rescore, kk = (int64_t)M - rescore + 1;
            const int64_t J = std::min<int64_t>(kk, C + 1), j0 = std::max<int64_t>(1, C - S + 2);
            next_calls = next_calls - 1 + (uint64_t)std::max<int64_t>(J - j0 + 1, 1);
        }
        st->next_calls += next_calls;
        if (fb[q]) {
            st->fallback_scans++;
            st->fallback_visited_nodes += hs[(size_t)q * ST_N + ST_VISITS];
            st->fallback_quantized_distance_comparisons += hs[(size_t)q * ST_N + ST_DQ];
        }
        if (!stream_only && rescore > 0) {
            // every row handed to the rescore window was fetched from the heap; so was every candidate the snapshot cannot
            // see (counted by the kernel, AM/scan.rs:258 + UT/table_slot.rs:45)
            const uint32_t nr = std::min(cnt[q], M) + (ix->visible ? hs[(size_t)q * ST_N + ST_INVIS] : 0u);
            st->full_distance_comparisons += nr;
            st->node_heap_reads += nr;
        }
    }
    return VS_OK;
}

static uint32_t stream_len(uint32_t rescore, uint32_t k) { return rescore > 0 ? rescore + k - 1 : k; }

// how many queries fit one launch given the workspace budget
static uint32_t chunk_queries(const vs_index* ix, const Caps& c, uint32_t M, uint32_t nq) {
    const size_t general = (size_t)c.hashcap * 4 + (size_t)(c.hcap > c.hl ? c.hcap - c.hl : 0) * 8;
    size_t per_q = (size_t)M * 12 + ix->vec_stride * 4ull + ix->code_stride * 8ull + 256;
    // (persistent grid: the dedup tables and heap spill arrays are per resident workgroup — at most 32 per CU — not per scan)
    const bool persist = c.f_on && knob_u32("VS_F_PERSIST", ix->tune.persist, 1);
    size_t fixed = 0;
    if (persist) {
        fixed = (size_t)ix->ctx->prop.multiProcessorCount * 32 * ((size_t)c.f_gcap * 4 + (size_t)c.f_gstride * 4);
        per_q += general / 64 + 64;
    } else if (c.f_on) {
        per_q += (size_t)((double)c.f_gcap * 4 * c.f_pool_frac) + (size_t)c.f_gstride * 4 + general / 64 + 64;
    } else {
        per_q += general;
    }
    // workspace budget: half of what is free on the device right now (plus what the workspace already holds), <= 64 GiB
    size_t budget = 24ull << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const SearchWorkspace& w = ix->ws;
        const size_t held = w.hash.bytes + w.heap_g.bytes + w.heap_g4.bytes + w.ghash4.bytes + w.stream_ids.bytes +
                            w.stream_ham.bytes + w.rr_dist.bytes + w.q_full.bytes + w.qcodes.bytes;
        budget = std::min<size_t>((free_b + held) / 2, 64ull << 30);
        budget = std::max<size_t>(budget, 1ull << 30);
    }
    budget = budget > fixed + (budget >> 2) ? budget - fixed : budget >> 2;
    uint32_t m = (uint32_t)std::max<size_t>(1, std::min<size_t>(budget / per_q, 1u << 20));
    return std::min(m, nq);
}

static int upload_label_keys(vs_index* ix, const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq,
                             const int16_t** d_labels, const uint32_t** d_off) {
    *d_labels = nullptr;
    *d_off = nullptr;
    if (!qlabel_off) return VS_OK;
    VS_REQUIRE(ix->d.has_labels && ix->label_off, "label scan keys on an index without labels");
    // LabelSet::from(Vec<Label>): sort_unstable + dedup (AM/labels/mod.rs:30-37)
    std::vector<int16_t> vals;
    std::vector<uint32_t> off(nq + 1, 0);
    for (uint32_t q = 0; q < nq; ++q) {
        VS_REQUIRE(qlabel_off[q] <= qlabel_off[q + 1], "qlabel_off must be non-decreasing");
        std::vector<int16_t> l(qlabels + qlabel_off[q], qlabels + qlabel_off[q + 1]);
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        vals.insert(vals.end(), l.begin(), l.end());
        off[q + 1] = (uint32_t)vals.size();
    }
    SearchWorkspace& w = ix->ws;
    VS_TRY(devbuf_reserve(ix->ctx, w.qlabels, std::max<size_t>(vals.size(), 1) * 2));
    VS_TRY(devbuf_reserve(ix->ctx, w.qlabel_off, off.size() * 4));
    if (!vals.empty()) VS_TRY(vs_dev_upload(ix->ctx, w.qlabels.p, vals.data(), vals.size() * 2));
    VS_TRY(vs_dev_upload(ix->ctx, w.qlabel_off.p, off.data(), off.size() * 4));
    *d_labels = (const int16_t*)w.qlabels.p;
    *d_off = (const uint32_t*)w.qlabel_off.p;
    return VS_OK;
}

static int search_host(vs_index* ix, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                       uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k, bool stream_only, uint32_t* out_ids,
                       uint64_t* out_tids, float* out_dist, uint32_t* out_ham, vs_stats* stats) {
    VS_REQUIRE(ix && (nq == 0 || queries), "search: bad args");
    VS_REQUIRE(L >= 1 && L <= 10000, "diskann.query_search_list_size %u outside [1,10000]", L);  // AM/guc.rs:11-26
    VS_REQUIRE(rescore <= 1000, "diskann.query_rescore %u outside [0,1000]", rescore);           // AM/guc.rs:28-43
    VS_REQUIRE(k >= 1, "k must be >= 1");
    if (ix->d.storage_type == VS_STORAGE_PLAIN) {
        VS_REQUIRE(!qlabel_off, "Plain storage does not support label filters");  // AM/plain/storage.rs:262
        // amgettuple, Plain arm: num_dimensions == num_dimensions_to_index => "no need to resort" (AM/scan.rs:392-399)
        if (ix->d.dim_index == ix->d.dim_full) rescore = 0;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (nq == 0) return VS_OK;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    SearchWorkspace& w = ix->ws;
    const uint32_t M = stream_only ? k : stream_len(rescore, k);
    Caps caps = initial_caps(ix, L, M);
    const int16_t* d_labels_all = nullptr;
    const uint32_t* d_off_all = nullptr;
    VS_TRY(upload_label_keys(ix, qlabels, qlabel_off, nq, &d_labels_all, &d_off_all));
    // Chunks of the batch run as a pipeline: while the device searches chunk i the host stages chunk i + 1 into the pinned ring and
    // hipMemcpyAsync moves it (copy stream), and the rows of chunk i - 1 go back to the caller — the PCIe time of a call is the
    // first chunk's way in and the last chunk's way out.  A batch that fits one launch is still cut into a few chunks when it is
    // large enough for that to pay (VS_HOST_CHUNKS: chunks to aim for, default 4; chunks of fewer than 32 768 scans do not fill
    // the device for long enough).  The query keys (AM/scan.rs:336-367) arrive on the host; nothing else does.
    uint32_t chunk = chunk_queries(ix, caps, M, nq);
    {
        const uint32_t want = std::max<uint32_t>(env_u32("VS_HOST_CHUNKS", 4), 1);
        const uint32_t floor_q = env_u32("VS_HOST_CHUNK_MIN", 32768);
        const uint32_t piece = std::max<uint32_t>((nq + want - 1) / want, floor_q);
        chunk = std::min(chunk, std::max<uint32_t>(piece, 1));
    }
    const size_t qrow = (size_t)ix->d.dim_full * 4;
    DevBuf* rawq[2] = {&w.raw_q, &w.raw_q2};
    DevBuf* oids[2] = {&w.out_ids, &w.out_ids2};
    DevBuf* otids[2] = {&w.out_tids, &w.out_tids2};
    DevBuf* odist[2] = {&w.out_dist, &w.out_dist2};
    const uint32_t nchunks = (nq + chunk - 1) / chunk;
    auto cq_of = [&](uint32_t ci) { return std::min(chunk, nq - ci * chunk); };
    auto stage_in = [&](uint32_t ci) -> int {
        const uint32_t cq = cq_of(ci);
        VS_TRY(devbuf_reserve(c, *rawq[ci & 1], (size_t)chunk * qrow));
        return vs_dev_upload(c, rawq[ci & 1]->p, queries + (size_t)ci * chunk * ix->d.dim_full, (size_t)cq * qrow);
    };
    auto launch = [&](uint32_t ci, BatchPlan& bp) -> int {
        const uint32_t cq = cq_of(ci), q0 = ci * chunk;
        VS_TRY(devbuf_reserve(c, *oids[ci & 1], (size_t)chunk * k * 4));
        VS_TRY(devbuf_reserve(c, *otids[ci & 1], (size_t)chunk * k * 8));
        VS_TRY(devbuf_reserve(c, *odist[ci & 1], (size_t)chunk * k * 4));
        bp = BatchPlan{cq, L, rescore, k, M, stream_only};
        // label CSR offsets are absolute into d_labels_all, so a chunk just offsets the off pointer
        return run_search_chunk(ix, bp, (const float*)rawq[ci & 1]->p, d_labels_all, d_off_all ? d_off_all + q0 : nullptr,
                                (uint32_t*)oids[ci & 1]->p, (uint64_t*)otids[ci & 1]->p, (float*)odist[ci & 1]->p, caps, false, stats);
    };
    // the scans of a launch that outgrew every pool are re-run (synchronously, growing capacities) and the window is redone
    auto finish = [&](uint32_t ci, const BatchPlan& bp) -> int {
        const uint32_t q0 = ci * chunk;
        std::vector<uint32_t> status(bp.nq);
        VS_HIP(hipMemcpyAsync(status.data(), w.status.p, (size_t)bp.nq * 4, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
        uint32_t ovf = 0;
        for (uint32_t v : status) ovf |= v;
        if (ovf) {
            VS_TRY(retry_failed_scans(ix, bp, d_labels_all, d_off_all ? d_off_all + q0 : nullptr, caps, stats));
            VS_TRY(run_post_search(ix, bp, (uint32_t*)oids[ci & 1]->p, (uint64_t*)otids[ci & 1]->p, (float*)odist[ci & 1]->p));
            VS_HIP(hipStreamSynchronize(c->stream));  // (stage_out does not wait for the compute stream)
        }
        return collect_stats(ix, bp.nq, M, rescore, stream_only, stats, L);
    };
    // a stream-only chunk hands back the workspace's own stream arrays: they go out before the next launch overwrites them
    auto stage_out = [&](uint32_t ci) -> int {
        const uint32_t cq = cq_of(ci), q0 = ci * chunk;
        if (stream_only) {
            VS_TRY(vs_dev_download(c, out_ids + (size_t)q0 * k, w.stream_ids.p, (size_t)cq * k * 4));
            if (out_ham) {
                VS_TRY(vs_dev_download(c, out_ham + (size_t)q0 * k, w.stream_ham.p, (size_t)cq * k * 4));
                if (ix->d.storage_type == VS_STORAGE_PLAIN)  // keys -> the f32 distances, bit for bit (rows past the end keep 0xFFFFFFFF)
                    for (size_t i = (size_t)q0 * k; i < ((size_t)q0 + cq) * k; ++i)
                        if (out_ids[i] != VS_INVALID_NODE) {
                            int32_t b = (int32_t)(out_ham[i] ^ 0x80000000u);
                            b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
                            out_ham[i] = (uint32_t)b;
                        }
            }
            return VS_OK;
        }
        VS_TRY(download_async_rows(c, out_ids + (size_t)q0 * k, oids[ci & 1]->p, (size_t)cq * k * 4));
        if (out_tids) VS_TRY(download_async_rows(c, out_tids + (size_t)q0 * k, otids[ci & 1]->p, (size_t)cq * k * 8));
        if (out_dist) VS_TRY(download_async_rows(c, out_dist + (size_t)q0 * k, odist[ci & 1]->p, (size_t)cq * k * 4));
        return VS_OK;
    };
    BatchPlan bp_cur{}, bp_next{};
    VS_TRY(stage_in(0));
    VS_TRY(launch(0, bp_cur));
    for (uint32_t ci = 0; ci < nchunks; ++ci) {
        if (ci + 1 < nchunks) VS_TRY(stage_in(ci + 1));  // (the device is busy with chunk ci)
        VS_TRY(finish(ci, bp_cur));
        if (stream_only) VS_TRY(stage_out(ci));
        if (ci + 1 < nchunks) VS_TRY(launch(ci + 1, bp_next));
        if (!stream_only) VS_TRY(stage_out(ci));  // (... and with chunk ci + 1 while these rows travel)
        bp_cur = bp_next;
    }
    if (stats) ix->last_stats = *stats;
    return VS_OK;
}

static int vs_search_batch_impl(vs_index* ix, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                               uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k, uint32_t* out_ids,
                               uint64_t* out_tids, float* out_dist, vs_stats* stats) {
    VS_REQUIRE(nq == 0 || out_ids, "vs_search_batch: out_ids is NULL");
    return search_host(ix, queries, qlabels, qlabel_off, nq, L, rescore, k, false, out_ids, out_tids, out_dist, nullptr,
                       stats);
}
extern "C" int vs_search_batch(vs_index* ix, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                               uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k, uint32_t* out_ids,
                               uint64_t* out_tids, float* out_dist, vs_stats* stats) {
    return vs_guard("vs_search_batch", [&] { return vs_search_batch_impl(ix, queries, qlabels, qlabel_off, nq, L, rescore, k, out_ids, out_tids, out_dist, stats); });
}


static int vs_stream_batch_impl(vs_index* ix, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                               uint32_t nq, uint32_t L, uint32_t m, uint32_t* out_ids, uint32_t* out_ham, vs_stats* stats) {
    VS_REQUIRE(nq == 0 || out_ids, "vs_stream_batch: out_ids is NULL");
    return search_host(ix, queries, qlabels, qlabel_off, nq, L, 0, m, true, out_ids, nullptr, nullptr, out_ham, stats);
}
extern "C" int vs_stream_batch(vs_index* ix, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                               uint32_t nq, uint32_t L, uint32_t m, uint32_t* out_ids, uint32_t* out_ham, vs_stats* stats) {
    return vs_guard("vs_stream_batch", [&] { return vs_stream_batch_impl(ix, queries, qlabels, qlabel_off, nq, L, m, out_ids, out_ham, stats); });
}


int vs_search_batch_dev_impl(vs_index* ix, const float* d_queries, const int16_t* d_qlabels,
                                   const uint32_t* d_qlabel_off, uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k,
                                   uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist) {
    VS_REQUIRE(ix && (nq == 0 || (d_queries && d_out_ids)), "vs_search_batch_dev: bad args");
    VS_REQUIRE(L >= 1 && L <= 10000 && rescore <= 1000 && k >= 1, "vs_search_batch_dev: GUC out of range");
    if (ix->d.storage_type == VS_STORAGE_PLAIN) {
        VS_REQUIRE(!d_qlabel_off, "Plain storage does not support label filters");
        if (ix->d.dim_index == ix->d.dim_full) rescore = 0;
    }
    SearchWorkspace& w = ix->ws;
    w.pending = false;
    if (nq == 0) return VS_OK;
    VS_HIP(hipSetDevice(ix->ctx->device));
    const uint32_t M = stream_len(rescore, k);
    Caps caps = initial_caps(ix, L, M);
    VS_REQUIRE(chunk_queries(ix, caps, M, nq) == nq, "vs_search_batch_dev: batch of %u queries exceeds the workspace budget", nq);
    BatchPlan bp{nq, L, rescore, k, M, false};
    VS_TRY(run_search_chunk(ix, bp, d_queries, d_qlabels, d_qlabel_off, d_out_ids, d_out_tids, d_out_dist, caps, false,
                            nullptr));
    w.pending = true;
    w.pend_nq = nq;
    w.pend_m = M;
    w.pend_L = L;
    {
        const PendingBatch pbv{bp, caps, d_qlabels, d_qlabel_off, d_out_ids, d_out_tids, d_out_dist};
        free(w.pend_blob);  // trivially copyable record
        w.pend_blob = malloc(sizeof(PendingBatch));
        VS_REQUIRE(w.pend_blob, "out of host memory");
        memcpy(w.pend_blob, &pbv, sizeof(pbv));
    }
    ix->last_stats = vs_stats{};
    return VS_OK;
}
extern "C" int vs_search_batch_dev(vs_index* ix, const float* d_queries, const int16_t* d_qlabels,
                                   const uint32_t* d_qlabel_off, uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k,
                                   uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist) {
    return vs_guard("vs_search_batch_dev", [&] { return vs_search_batch_dev_impl(ix, d_queries, d_qlabels, d_qlabel_off, nq, L, rescore, k, d_out_ids, d_out_tids, d_out_dist); });
}


int vs_search_batch_dev_finish_impl(vs_index* ix, vs_stats* stats) {
    VS_REQUIRE(ix, "vs_search_batch_dev_finish: index is NULL");
    SearchWorkspace& w = ix->ws;
    if (!w.pending) {
        vs_set_error("vs_search_batch_dev_finish: no batch in flight");
        return VS_ERR_STATE;
    }
    w.pending = false;
    const uint32_t nq = w.pend_nq, M = w.pend_m;
    VS_REQUIRE(w.pend_blob, "vs_search_batch_dev_finish: no batch descriptor");
    PendingBatch pb;
    memcpy(&pb, w.pend_blob, sizeof(pb));
    std::vector<uint32_t> status(nq);
    VS_HIP(hipMemcpyAsync(status.data(), w.status.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ix->ctx->stream));
    VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    uint32_t ovf = 0;
    for (uint32_t v : status) ovf |= v;
    vs_stats st{};
    if (ovf) {
        // some scans outgrew even the fallback pools of the asynchronous launch: re-run exactly those (synchronously,
        // with growing capacities), then redo the rerank / rescore window so the outputs are complete
        VS_TRY(retry_failed_scans(ix, pb.bp, pb.d_qlabels, pb.d_qlabel_off, pb.caps, &st));
        VS_TRY(run_post_search(ix, pb.bp, pb.d_out_ids, pb.d_out_tids, pb.d_out_dist));
        VS_HIP(hipStreamSynchronize(ix->ctx->stream));
    }
    VS_TRY(collect_stats(ix, nq, M, pb.bp.rescore, false, &st, w.pend_L));
    ix->last_stats = st;
    if (stats) *stats = st;
    return VS_OK;
}
extern "C" int vs_search_batch_dev_finish(vs_index* ix, vs_stats* stats) {
    return vs_guard("vs_search_batch_dev_finish", [&] { return vs_search_batch_dev_finish_impl(ix, stats); });
}
