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
static uint32_t env_u32(const char* name, uint32_t dflt);
static size_t slab_bytes_wanted(const vs_index* ix) {
    if (ix->d.n < env_u32("VS_WS_SLAB_MIN_N", 4u << 20)) return 0;
    return (size_t)env_u32("VS_WS_SLAB_MB", 2048) << 20;
}
// Device memory is not uniform for the search kernel's request mix (k_ws_probe_mix): up to VS_WS_SLAB_CANDIDATES allocations of the
// slab's size are made (all held until the choice, so that each lands somewhere else) and every PAIR (tables on candidate i, heap
// arrays on candidate j, i == j: the two halves of one allocation) is timed with the mix probe against THIS index's arrays (a few ms
// each); the best pair is kept, the other candidates go back to the device.
static void slab_select(vs_index* ix, WsSlab* s, size_t slab_bytes) {
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
        }
    }
    return devbuf_reserve(ix->ctx, b, bytes);
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

static hipEvent_t pool_event(vs_ctx* c) {
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
static int download_async_rows(vs_ctx* c, void* dst, const void* src, size_t bytes) {
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

static uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = vs_opt_get(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

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
        if (caps.f_lh == 0 && !f.vr && vmode && !env_u32("VS_PHASE", 0) && f.gcap <= (1u << 16)) {
            f.vwords = (f.gcap + 127) / 128;
            if (vmode >= 2 && !f.rc && f.gcap % 32 == 0) {
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
        uint32_t gregion = f.gcap;
        if (vmode == 3 && f.vslot == 1 && caps.f_lh == 0) {
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
                if (res_q >= res_s || env_u32("VS_F_SLOTMAP_FORCE", 0)) {  // (taken only while it costs no scans per CU)
                    f = g;
                    gregion = g.gregion;
                }
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


static int vs_search_batch_dev_impl(vs_index* ix, const float* d_queries, const int16_t* d_qlabels,
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


static int vs_search_batch_dev_finish_impl(vs_index* ix, vs_stats* stats) {
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
