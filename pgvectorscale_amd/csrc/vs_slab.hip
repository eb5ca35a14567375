// vs_slab.hip — host side of libvsgpu.so: the workspace slab (WsSlab, vs_internal.h).  The private state of the scans in flight (dedup
// tables, heap spill arrays) is sub-allocated from one allocation per index, CHOSEN among probed candidates: device memory comes in
// kinds and k_search_fast is 6-9 % slower when its private state shares a kind with the code rows (LAB_NOTEBOOK 11d).
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
// the workspace slab (WsSlab, vs_internal.h)
// ---------------------------------------------------------------------------------------------------------------
WsSlab* vs_slab_new(int device) {
    WsSlab* s = new WsSlab();
    s->device = device;
    return s;
}
void vs_slab_release(WsSlab* s) {
    if (!s) return;
    bool last;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        last = --s->refs <= 0;
    }
    if (!last) return;
    for (void* o : s->owned)
        if (o) (void)hipFree(o);
    delete s;
}
// The caller's own device memory as the slab of this handle (and of the views made of it afterwards): a host that manages HBM itself,
// or one that has probed where the hot regions run fastest (vs_ws_probe).  Before the handle's first search; the memory stays the
// caller's and must outlive the handle and its views.
static int vs_index_set_slab_impl(vs_index* ix, void* p, size_t bytes) {
    VS_REQUIRE(ix && p && bytes >= (1u << 20), "vs_index_set_slab: bad args (at least 1 MiB)");
    VS_REQUIRE(!ix->ws.ghash4.p && !ix->ws.heap_g4.p, "vs_index_set_slab: the handle has searched already (its workspace exists)");
    WsSlab* s = vs_slab_new(ix->ctx->device);
    const size_t half = bytes / 2 / 65536 * 65536;
    s->base[0] = p;
    s->base[1] = (char*)p + half;
    s->bytes[0] = s->bytes[1] = half;
    s->tried = true;  // (nothing owned: the memory stays the caller's)
    s->external = true;
    vs_slab_release(ix->slab);
    ix->slab = s;
    return VS_OK;
}
extern "C" int vs_index_set_slab(vs_index* ix, void* p, size_t bytes) {
    return vs_guard("vs_index_set_slab", [&] { return vs_index_set_slab_impl(ix, p, bytes); });
}

// The private-state traffic of k_search_fast in miniature, on an arbitrary device region: 24 single-wave workgroups per CU, each with
// its own contiguous share of the region; per iteration 28 random 16-byte loads, 31 random 4-byte stores and 56 random 8-byte loads
// inside that share (the dedup group loads, the dedup inserts, the heap's child pairs).  Milliseconds for `iters` iterations: where a
// region is slow for this shape, the search kernel is slow with its workspace there ("State").
__global__ __launch_bounds__(64) void k_ws_probe(uint8_t* base, size_t share, uint32_t iters, uint64_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint8_t* tab = base + (size_t)blockIdx.x * share;
    const uint32_t u16 = (uint32_t)(share / 16), u8 = (uint32_t)(share / 8), u4 = (uint32_t)(share / 4);
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 99u;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t h = ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u;
        h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
        if (lane < 28) {
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)(uint32_t)(((uint64_t)h * u16) >> 32) * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (lane < 56) acc += *reinterpret_cast<const uint64_t*>(tab + (size_t)(uint32_t)(((uint64_t)(h * 0x9E3779B1u) * u8) >> 32) * 8);
        if (lane < 31 && acc != 0x123456789abcull) *reinterpret_cast<uint32_t*>(tab + (size_t)(uint32_t)(((uint64_t)(h ^ 0x5bd1e995u) * u4) >> 32) * 4) = h;
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
static int vs_ws_probe_impl(vs_ctx* c, void* p, size_t bytes, uint32_t iters, float* ms_out) {
    VS_REQUIRE(c && p && ms_out && bytes >= (64u << 20) && iters > 0, "vs_ws_probe: bad args (a region of at least 64 MiB)");
    VS_HIP(hipSetDevice(c->device));
    const uint32_t waves = (uint32_t)c->prop.multiProcessorCount * 24;
    const size_t share = bytes / waves / 16 * 16;
    static DeviceOnce attr_set;
    if (attr_set.pending(c->device)) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done(c->device);
    }
    const size_t lds = (160 * 1024) / 24 - 64;  // pins 24 workgroups per CU
    hipEvent_t e0, e1;
    VS_HIP(hipEventCreate(&e0));
    VS_HIP(hipEventCreate(&e1));
    uint64_t* sink = nullptr;
    VS_HIP(hipMalloc(&sink, 8));
    hipLaunchKernelGGL(k_ws_probe, dim3(waves), dim3(64), lds, c->stream, (uint8_t*)p, share, std::max(iters / 8, 1u), sink);  // warm-up
    VS_HIP(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(k_ws_probe, dim3(waves), dim3(64), lds, c->stream, (uint8_t*)p, share, iters, sink);
    VS_HIP(hipEventRecord(e1, c->stream));
    VS_HIP(hipEventSynchronize(e1));
    VS_HIP(hipEventElapsedTime(ms_out, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return VS_OK;
}
extern "C" int vs_ws_probe(vs_ctx* c, void* p, size_t bytes, uint32_t iters, float* ms_out) {
    return vs_guard("vs_ws_probe", [&] { return vs_ws_probe_impl(c, p, bytes, iters, ms_out); });
}

// The same with the rest of the search kernel's request mix around it, read from THIS index's arrays: per iteration one random neighbor
// row (50 x 4-byte non-temporal loads), two passes of 16 random code rows (4 lanes x 16-byte non-temporal loads per 64 bytes of a row),
// and the private-state requests above with the tables at the region's start and the heap arrays in its second half.  Device memory
// is not uniform for this mix: the same launch takes 30.3 or 32.9 ms depending on which allocation holds the private state
// (scripts/microbench/placemix.hip, profiles/r05/s6_placemix.txt), a property of the allocation, not of offsets inside it — and
// k_search_fast follows (156 / 170 ms, profiles/r05/s4_placement_map_50m.txt).  So the slab is CHOSEN: see slab_select below.
struct WsMixArgs {
    const uint8_t* codes;
    const uint32_t* nbrs;
    uint64_t nrows;
    uint32_t code_row_bytes, nbr_stride, R;
    uint8_t* tab_base;
    uint8_t* heap_base;
    uint32_t tab_bytes, heap_bytes, iters;
    uint64_t* sink;
};
__global__ __launch_bounds__(64) void k_ws_probe_mix(WsMixArgs a) {
    const uint32_t lane = threadIdx.x, l4 = lane & 3, grp = lane >> 2;
    uint8_t* tab = a.tab_base + (size_t)blockIdx.x * a.tab_bytes;
    uint8_t* heap = a.heap_base + (size_t)blockIdx.x * a.heap_bytes;
    const uint32_t t16 = a.tab_bytes / 16, t4 = a.tab_bytes / 4, h8 = a.heap_bytes / 8;
    const uint32_t pieces = std::min<uint32_t>(a.code_row_bytes / 64, 3u);  // 64-byte pieces of a code row a 4-lane group reads
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 12345u;
    auto mix = [](uint32_t x) {
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        return x;
    };
    for (uint32_t it = 0; it < a.iters; ++it) {
        const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
        const uint64_t nrow = ((uint64_t)mix(ctr ^ (it * 0x9E3779B1u)) * a.nrows) >> 32;
        if (lane < a.R) acc += __builtin_nontemporal_load(a.nbrs + nrow * a.nbr_stride + lane);
        if (lane < 28) {
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)(uint32_t)(((uint64_t)h * t16) >> 32) * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (lane < 31 && acc != 0x123456789abcull)
            *reinterpret_cast<uint16_t*>(tab + (size_t)(uint32_t)(((uint64_t)(h ^ 0x5bd1e995u) * t4) >> 32) * 4) = (uint16_t)h;
        if (lane < 56) acc += *reinterpret_cast<const uint64_t*>(heap + (size_t)(uint32_t)(((uint64_t)(h * 0x9E3779B1u) * h8) >> 32) * 8);
        for (uint32_t p = 0; p < 2; ++p) {
            const uint64_t row = ((uint64_t)mix(ctr ^ ((it * 64u + p * 16u + grp) * 0x9E3779B1u) ^ 0xabcdefu) * a.nrows) >> 32;
            const uint8_t* r = a.codes + row * a.code_row_bytes + 16u * l4;
            for (uint32_t t = 0; t < pieces; ++t) {
                const __uint128_t v = __builtin_nontemporal_load(reinterpret_cast<const __uint128_t*>(r + 64u * t));
                acc += (uint64_t)__popcll((unsigned long long)v) + (uint64_t)__popcll((unsigned long long)(v >> 64));
            }
        }
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) a.sink[0] = acc;
}
// tables on [tab, tab + half), heap arrays on [heap, heap + half)
static int ws_probe_mix(vs_index* ix, void* tab, void* heap, size_t half, uint32_t iters, float* ms_out) {
    vs_ctx* c = ix->ctx;
    VS_REQUIRE(tab && heap && ms_out && half >= (32u << 20) && iters > 0 && ix->d.n > 0 && ix->codes && ix->nbrs, "vs_ws_probe_mix: bad args");
    VS_HIP(hipSetDevice(c->device));
    const uint32_t waves = (uint32_t)c->prop.multiProcessorCount * 24;
    WsMixArgs a;
    a.codes = reinterpret_cast<const uint8_t*>(ix->codes);
    a.nbrs = ix->nbrs;
    a.nrows = ix->d.n;
    a.code_row_bytes = ix->code_stride * 8;
    a.nbr_stride = ix->nbr_stride;
    a.R = std::min<uint32_t>(ix->d.num_neighbors, 64);
    a.tab_base = (uint8_t*)tab;
    a.heap_base = (uint8_t*)heap;
    a.tab_bytes = (uint32_t)std::min<size_t>(half / waves / 16 * 16, 36864);   // (a 16-bit table of 16 Ki slots + its overflow table)
    a.heap_bytes = (uint32_t)std::min<size_t>(half / waves / 16 * 16, 46368);
    a.iters = iters;
    static DeviceOnce attr_set;
    if (attr_set.pending(c->device)) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws_probe_mix), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done(c->device);
    }
    const size_t lds = (160 * 1024) / 24 - 64;  // pins 24 workgroups per CU
    hipEvent_t e0, e1;
    VS_HIP(hipEventCreate(&e0));
    VS_HIP(hipEventCreate(&e1));
    uint64_t* sink = nullptr;
    VS_HIP(hipMalloc(&sink, 8));
    a.sink = sink;
    WsMixArgs w = a;
    w.iters = std::max(iters / 8, 1u);
    hipLaunchKernelGGL(k_ws_probe_mix, dim3(waves), dim3(64), lds, c->stream, w);  // warm-up
    VS_HIP(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(k_ws_probe_mix, dim3(waves), dim3(64), lds, c->stream, a);
    VS_HIP(hipEventRecord(e1, c->stream));
    VS_HIP(hipEventSynchronize(e1));
    VS_HIP(hipEventElapsedTime(ms_out, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return VS_OK;
}
extern "C" int vs_ws_probe_mix(vs_index* ix, void* p, size_t bytes, uint32_t iters, float* ms_out) {
    return vs_guard("vs_ws_probe_mix", [&]() -> int {
        VS_REQUIRE(ix != nullptr && p != nullptr, "vs_ws_probe_mix: bad args");
        const size_t half = bytes / 2 / 4096 * 4096;
        return ws_probe_mix(ix, p, (char*)p + half, half, iters, ms_out);
    });
}

// VS_WS_SLAB_MB (default 2048; 0: no slab) for indexes of VS_WS_SLAB_MIN_N nodes and more (default 4M: smaller indexes run the
// LDS-table regime or tables of a few MB in all, where placement was never seen to matter)
size_t slab_bytes_wanted(const vs_index* ix) {
    if (ix->d.n < env_u32("VS_WS_SLAB_MIN_N", 4u << 20)) return 0;
    return (size_t)env_u32("VS_WS_SLAB_MB", 2048) << 20;
}
// Device memory is not uniform for the search kernel's request mix (k_ws_probe_mix): up to VS_WS_SLAB_CANDIDATES allocations of the
// slab's size are made (all held until the choice, so that each lands somewhere else) and every PAIR (tables on candidate i, heap
// arrays on candidate j, i == j: the two halves of one allocation) is timed with the mix probe against THIS index's arrays (a few ms
// each); the best pair is kept, the other candidates go back to the device.
void slab_select(vs_index* ix, WsSlab* s, size_t slab_bytes) {
    s->tried = true;
    uint32_t ncand = std::max<uint32_t>(1, std::min<uint32_t>(env_u32("VS_WS_SLAB_CANDIDATES", 8), 8));
    const size_t half = slab_bytes / 2 / 65536 * 65536;
    void* cand[8] = {nullptr};
    void* spacer[8] = {nullptr};
    uint32_t got = 0;
    const bool probe = ncand > 1 && ix->codes && ix->nbrs && ix->d.n > 0 && half >= ((size_t)64 << 20);
    // The kernel is slow where its private state lives in the same kind of memory as the code rows and fast elsewhere, and the kinds
    // come in stretches of tens of GB in allocation order (profiles/r05/s7, s12, s13): candidates made back to back would all be of
    // one kind, so spacers (returned right after the choice) spread them over what the device has free.
    size_t sp = 0;
    if (probe) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // Transient footprint (advisor, round 5): candidates and spacers are all held at once for the ~0.3 s of the probes, while
            // other users of the device (other processes, torch, other indexes, scan pools allocating at the same moment) may want
            // memory too.  So the whole transient set stays within VS_WS_SLAB_PROBE_PCT (default 50) per cent of what is free NOW and
            // never touches the last VS_WS_SLAB_KEEP_FREE_MB; a host that wants the probing at a moment of its own choosing calls
            // vs_index_prepare_workspace() after loading the index.
            const size_t keep = (size_t)env_u32("VS_WS_SLAB_KEEP_FREE_MB", 12288) << 20;  // what the probing never touches
            const size_t pct = std::min<uint32_t>(env_u32("VS_WS_SLAB_PROBE_PCT", 50), 100);
            const size_t budget = std::min<size_t>(free_b / 100 * pct, free_b > keep ? free_b - keep : 0);
            if (slab_bytes && budget / slab_bytes < ncand) ncand = (uint32_t)std::max<size_t>(1, budget / slab_bytes);  // (fewer candidates on a full device)
            const size_t need = (size_t)ncand * slab_bytes;
            if (budget > need && ncand > 1) sp = std::min<size_t>((budget - need) / (ncand - 1), (size_t)env_u32("VS_WS_SLAB_SPACER_MB", 16384) << 20);
            sp = sp / ((size_t)2 << 20) * ((size_t)2 << 20);
        } else {
            (void)hipGetLastError();
        }
    }
    for (uint32_t i = 0; i < (probe ? ncand : 1u); ++i) {
        if (hipMalloc(&cand[i], slab_bytes) != hipSuccess) {  // best effort: what the device can spare
            (void)hipGetLastError();
            cand[i] = nullptr;
            break;
        }
        got = i + 1;
        if (sp >= ((size_t)64 << 20) && i + 1 < ncand && hipMalloc(&spacer[i], sp) != hipSuccess) {
            (void)hipGetLastError();
            spacer[i] = nullptr;
        }
    }
    for (void* p : spacer)
        if (p) (void)hipFree(p);
    if (!got) return;
    uint32_t bi = 0, bj = 0;
    float ms[8][8];
    if (probe && got > 1) {
        const size_t ph = std::min<size_t>(half, (size_t)512 << 20);
        float best = 1e30f;
        for (uint32_t i = 0; i < got; ++i)
            for (uint32_t j = 0; j < got; ++j) {
                // (i == j: the heap arrays in the second half of the same allocation; i != j: at the start of the other one)
                void* hb = i == j ? (void*)((char*)cand[j] + half) : cand[j];
                if (ws_probe_mix(ix, cand[i], hb, ph, 200, &ms[i][j]) != VS_OK) ms[i][j] = 1e30f;
                // a pair of two allocations has to beat the best single one by 0.5 %: it costs the device a second slab
                const float v = i == j ? ms[i][j] : ms[i][j] * 1.005f;
                if (v < best) {
                    best = v;
                    bi = i;
                    bj = j;
                }
            }
    }
    s->base[0] = cand[bi];
    s->owned[0] = cand[bi];
    if (bi == bj) {
        s->base[1] = (char*)cand[bi] + half;
        s->bytes[0] = s->bytes[1] = half;
    } else {  // two allocations, each whole for its kind
        s->base[1] = cand[bj];
        s->owned[1] = cand[bj];
        s->bytes[0] = s->bytes[1] = slab_bytes;
    }
    for (uint32_t i = 0; i < got; ++i)
        if (i != bi && i != bj) (void)hipFree(cand[i]);
    if (env_u32("VS_WS_DEBUG", 0)) {
        fprintf(stderr, "[VS_WS_DEBUG] workspace slab: %u candidates of %zu MB (spacers of %zu MB), tables on %u, heap arrays on %u", got, slab_bytes >> 20,
                sp >> 20, bi, bj);
        if (probe && got > 1) {
            fprintf(stderr, "; mix probe ms [tables][heaps]:");
            for (uint32_t i = 0; i < got; ++i) {
                fprintf(stderr, " [");
                for (uint32_t j = 0; j < got; ++j) fprintf(stderr, "%s%.2f", j ? " " : "", ms[i][j]);
                fprintf(stderr, "]");
            }
        }
        fprintf(stderr, "\n");
    }
}
extern "C" int vs_index_prepare_workspace(vs_index* ix) {
    return vs_guard("vs_index_prepare_workspace", [&]() -> int {
        VS_REQUIRE(ix != nullptr, "vs_index_prepare_workspace: index is NULL");
        WsSlab* s = ix->slab;
        const size_t slab_bytes = s ? slab_bytes_wanted(ix) : 0;
        if (!slab_bytes || ix->is_view) return VS_OK;
        VS_HIP(hipSetDevice(ix->ctx->device));
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->base[0] && !s->tried && !s->external) slab_select(ix, s, slab_bytes);
        return VS_OK;
    });
}
int devbuf_reserve_hot(vs_index* ix, DevBuf& b, size_t bytes, int which) {
    if (bytes <= b.bytes) return VS_OK;
    WsSlab* s = ix->slab;
    const size_t slab_bytes = s ? slab_bytes_wanted(ix) : 0;
    if (slab_bytes || (s && s->external)) {  // (the caller's memory is used whatever the size rule says)
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->base[0] && !s->tried) slab_select(ix, s, slab_bytes);
        if (s->base[which]) {
            char* const base = (char*)s->base[which];
            const size_t kAlign = 1u << 16;
            const size_t want = (bytes + bytes / 8 + kAlign - 1) / kAlign * kAlign;
            // the newest chunk of a region grows in place
            if (b.in_slab && (char*)b.p + b.bytes == base + s->used[which] && (size_t)((char*)b.p - base) + want <= s->bytes[which]) {
                s->used[which] = (size_t)((char*)b.p - base) + want;
                b.bytes = want;
                return VS_OK;
            }
            if (s->used[which] + want <= s->bytes[which]) {
                if (b.p && !b.in_slab) VS_HIP(hipFree(b.p));  // (synchronises: nothing in flight reads the old array)
                b.p = base + s->used[which];
                b.bytes = want;
                b.in_slab = true;
                s->used[which] += want;
                return VS_OK;
            }
            // chunks are bump-allocated and never returned (views that come and go, growth that cannot extend in place): a region
            // that has run out sends the array to an allocation of its own — correct, but without the placement the slab exists for
            if (env_u32("VS_WS_DEBUG", 0))
                fprintf(stderr, "[VS_WS_DEBUG] workspace slab: region %d exhausted (%zu of %zu B used, %zu wanted): own allocation\n", which,
                        s->used[which], s->bytes[which], want);
        }
    }
    return devbuf_reserve(ix->ctx, b, bytes);
}
