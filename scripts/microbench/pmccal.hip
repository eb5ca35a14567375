// pmccal.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of k_search_fast.
//
// Each kernel below issues ONE request shape of the search kernel, a known number of times, over a footprint far beyond the
// caches (the 256 MB Infinity Cache included), with the search kernel's launch shape (single-wave workgroups, 24 per CU):
//   cal_stream16   coalesced streaming read, 16 B per lane                      (the guide's case: FETCH_SIZE = bytes / 2)
//   cal_rows192    random 192-byte code rows: 4 lanes x 3 non-temporal 16-byte loads   (ham_row_reg)
//   cal_rows200    random neighbor rows: lanes 0..49 x non-temporal 4-byte loads, 256-byte row stride  (load_stream32)
//   cal_group16    random 16-byte loads inside a 56-KB table private to the wave        (dedup group loads)
//   cal_pair8      random 8-byte loads inside a 32-KB array private to the wave         (heap spill child pairs)
//   cal_store4     random 4-byte stores into the wave's 56-KB table                     (dedup inserts; ~31 per "expansion")
//   cal_store2     random 2-byte stores into the wave's 28-KB table                     (16-bit table entries)
//   cal_wstream16  coalesced streaming write, 16 B per lane
// scripts/pmc_calibrate.sh runs the binary plain (kernel order + request counts), then once per counter, and
// scripts/pmc_calibrate.py divides: counter bytes per request of each shape -> profiles/r05/pmc_calibration_randmem.json,
// which scripts/pmc_traffic.py uses instead of a modelled request mix.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void cal_stream16(const v2u64* __restrict__ p, size_t n16, uint64_t* sink) {
    uint64_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 64) {
        const v2u64 v = __builtin_nontemporal_load(p + i);
        acc += v.x ^ v.y;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
__global__ __launch_bounds__(64) void cal_wstream16(v2u64* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 64) {
        v2u64 v;
        v.x = i;
        v.y = ~i;
        p[i] = v;
    }
}
// 16 rows per iteration (4 lanes per row)
__global__ __launch_bounds__(64) void cal_rows192(const uint8_t* __restrict__ codes, uint64_t nrows, uint32_t iters, uint64_t* sink) {
    const uint32_t lane = threadIdx.x, l4 = lane & 3, grp = lane >> 2;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t h = mix(ctr ^ ((it * 64u + grp) * 0x9E3779B1u));
        const uint64_t row = ((uint64_t)h * nrows) >> 32;
        const uint8_t* r = codes + row * 192 + 16u * l4;
        const v2u64 a = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r));
        const v2u64 b = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 64));
        const v2u64 c = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 128));
        acc += __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y);
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
// one row per iteration
__global__ __launch_bounds__(64) void cal_rows200(const uint32_t* __restrict__ nbrs, uint64_t nrows, uint32_t iters, uint64_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 777u;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t h = mix(ctr ^ (it * 0x9E3779B1u));
        const uint64_t row = ((uint64_t)h * nrows) >> 32;
        if (lane < 50) acc += __builtin_nontemporal_load(nbrs + row * 64 + lane);
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
// `per` lanes per iteration each load BYTES bytes at a random aligned place of the wave's private array of tab_bytes
template <int BYTES>
__global__ __launch_bounds__(64) void cal_small_load(const uint8_t* __restrict__ tables, uint32_t tab_bytes, uint32_t iters, uint32_t per,
                                                      uint64_t* sink) {
    const uint32_t lane = threadIdx.x;
    const uint8_t* tab = tables + (size_t)blockIdx.x * tab_bytes;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 99u;
    const uint32_t units = tab_bytes / BYTES;
    for (uint32_t it = 0; it < iters; ++it) {
        if (lane < per) {
            const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
            const uint32_t u = (uint32_t)(((uint64_t)h * units) >> 32);
            if (BYTES == 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)u * 16);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {
                acc += *reinterpret_cast<const uint64_t*>(tab + (size_t)u * 8);
            }
        }
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(64) void cal_small_store(T* __restrict__ tables, uint32_t tab_units, uint32_t iters, uint32_t per) {
    const uint32_t lane = threadIdx.x;
    T* tab = tables + (size_t)blockIdx.x * tab_units;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 4242u;
    for (uint32_t it = 0; it < iters; ++it) {
        if (lane < per) {
            const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
            tab[(uint32_t)(((uint64_t)h * tab_units) >> 32)] = (T)h;
        }
        ctr += 0x632be5abu;
    }
}

static hipEvent_t e0, e1;
static void begin() { CK(hipEventRecord(e0)); }
static void end(const char* name, double requests, double known_bytes, const char* unit) {
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // one line per kernel, in dispatch order: scripts/pmc_calibrate.py joins them with the counter CSVs by kernel name
    printf("CAL %s requests=%.0f known_bytes=%.0f unit=%s ms=%.3f GBps=%.0f\n", name, requests, known_bytes, unit, ms, known_bytes / ms / 1e6);
    fflush(stdout);
}

int main() {
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t nrows = 50000000ull;
    const uint32_t waves = 256 * 24;  // the persistent grid of k_search_fast: 24 single-wave workgroups per CU
    const size_t lds = (160 * 1024) / 24 - 64;  // pins the number of resident workgroups per CU
    uint8_t* codes;
    CK(hipMalloc(&codes, nrows * 192));
    CK(hipMemset(codes, 0x5a, nrows * 192));
    uint32_t* nbrs;
    CK(hipMalloc(&nbrs, nrows * 256));
    CK(hipMemset(nbrs, 0x11, nrows * 256));
    uint8_t* tables;
    const uint32_t tab_bytes = 14336 * 4;  // the fitted dedup table of the headline point: 14 K four-byte slots
    CK(hipMalloc(&tables, (size_t)waves * tab_bytes));
    CK(hipMemset(tables, 0x22, (size_t)waves * tab_bytes));
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    CK(hipDeviceSynchronize());
    const void* fns[] = {(const void*)cal_rows192, (const void*)cal_rows200, (const void*)cal_small_load<16>, (const void*)cal_small_load<8>,
                         (const void*)cal_small_store<uint32_t>, (const void*)cal_small_store<uint16_t>};
    for (const void* f : fns) CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    const size_t n16 = nrows * 192 / 16;
    begin();
    hipLaunchKernelGGL(cal_stream16, dim3(waves * 4), dim3(64), 0, 0, (const v2u64*)codes, n16, sink);
    end("cal_stream16", (double)n16 / 8, (double)n16 * 16, "128B_line");

    const uint32_t it_rows = 6000;
    begin();
    hipLaunchKernelGGL(cal_rows192, dim3(waves), dim3(64), lds, 0, codes, nrows, it_rows, sink);
    end("cal_rows192", (double)waves * it_rows * 16, (double)waves * it_rows * 16 * 192, "row");

    const uint32_t it_nb = 24000;
    begin();
    hipLaunchKernelGGL(cal_rows200, dim3(waves), dim3(64), lds, 0, nbrs, nrows, it_nb, sink);
    end("cal_rows200", (double)waves * it_nb, (double)waves * it_nb * 200, "row");

    const uint32_t it_small = 8000;
    begin();
    hipLaunchKernelGGL(cal_small_load<16>, dim3(waves), dim3(64), lds, 0, tables, tab_bytes, it_small, 28u, sink);
    end("cal_small_load<16>", (double)waves * it_small * 28, (double)waves * it_small * 28 * 16, "load");
    begin();
    hipLaunchKernelGGL(cal_small_load<8>, dim3(waves), dim3(64), lds, 0, tables, 32768u, it_small, 56u, sink);
    end("cal_small_load<8>", (double)waves * it_small * 56, (double)waves * it_small * 56 * 8, "load");

    begin();
    hipLaunchKernelGGL(cal_small_store<uint32_t>, dim3(waves), dim3(64), lds, 0, (uint32_t*)tables, tab_bytes / 4, it_small, 31u);
    end("cal_small_store<unsigned int>", (double)waves * it_small * 31, (double)waves * it_small * 31 * 4, "store");
    begin();
    hipLaunchKernelGGL(cal_small_store<uint16_t>, dim3(waves), dim3(64), lds, 0, (uint16_t*)tables, tab_bytes / 4, it_small, 31u);
    end("cal_small_store<unsigned short>", (double)waves * it_small * 31, (double)waves * it_small * 31 * 2, "store");

    begin();
    hipLaunchKernelGGL(cal_wstream16, dim3(waves * 4), dim3(64), 0, 0, (v2u64*)codes, n16);
    end("cal_wstream16", (double)n16 / 8, (double)n16 * 16, "128B_line");
    CK(hipDeviceSynchronize());
    return 0;
}
