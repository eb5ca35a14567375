// randmem.hip — what the memory system of one MI355X sustains for the access pattern of the streaming beam search
// (test infrastructure, not product): random 192-byte code rows (4 lanes x 16 B x 3 per row, 16 rows per wave pass),
// random 4-byte probes (CAS / load) of per-wave dedup tables, and both together at the search kernel's ratio.
//   hipcc -O3 --offload-arch=gfx950 -o randmem randmem.hip && ./randmem
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));           \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// MODE bit 0: gather rows; bit 1: probe table.  PROBE: 0 CAS, 1 load only, 2 load then CAS when empty, 3 CAS that fails (table pre-filled)
template <int MODE, int PROBE, int NT>
__global__ __launch_bounds__(64) void k_mix(const uint8_t* __restrict__ codes, uint64_t nrows, uint32_t row_bytes,
                                            uint32_t* tables, uint32_t tab_words /*per wave, pow2*/, uint32_t iters,
                                            uint32_t rows_per_iter /*multiple of 16*/, uint32_t probes_per_iter,
                                            uint64_t* sink) {
    extern __shared__ unsigned char pad_lds[];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const uint32_t l4 = lane & 3, grp = lane >> 2;
    uint32_t* tab = tables + (size_t)wave * tab_words;
    uint64_t acc = 0;
    uint32_t ctr = wave * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        if (MODE & 2) {
            if (lane < probes_per_iter) {
                const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
                const uint32_t slot = h & (tab_words - 1);
                const uint32_t nid = (h >> 4) | 1u;
                uint32_t old;
                if (PROBE == 0 || PROBE == 3) old = atomicCAS(&tab[slot], PROBE == 3 ? 0xFFFFFFFEu : 0xFFFFFFFFu, nid);
                else if (PROBE == 1) old = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (PROBE == 4) {
                    const uint4 b = *reinterpret_cast<const uint4*>(tab + (slot & ~3u));
                    old = b.x ^ b.y ^ b.z ^ b.w;
                    if ((h >> 24) < 156u) tab[slot] = nid;  // 61 % of the probes insert
                } else {
                    old = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == 0xFFFFFFFFu) old = atomicCAS(&tab[slot], 0xFFFFFFFFu, nid);
                }
                acc += old;
            }
        }
        if (MODE & 1) {
            for (uint32_t p = 0; p < rows_per_iter; p += 16) {
                const uint32_t h = mix(ctr ^ ((it * 64u + p + grp) * 0x9E3779B1u));
                const uint64_t row = ((uint64_t)h * nrows) >> 32;
                const uint8_t* r = codes + row * row_bytes + 16u * l4;
                ulonglong2 a, b, c;
                if (NT) {
                    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
                    const v2u64 a_ = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r));
                    const v2u64 b_ = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 64));
                    const v2u64 c_ = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 128));
                    a = make_ulonglong2(a_.x, a_.y); b = make_ulonglong2(b_.x, b_.y); c = make_ulonglong2(c_.x, c_.y);
                } else {
                    a = *reinterpret_cast<const ulonglong2*>(r);
                    b = *reinterpret_cast<const ulonglong2*>(r + 64);
                    c = *reinterpret_cast<const ulonglong2*>(r + 128);
                }
                acc += __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y);
            }
        }
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// The same requests as the mix above, but issued the way a scan issues them: one expansion = a neighbor row (256 B, address
// known only after the previous expansion), then the bucket loads + stores (addresses from that row), then the code rows
// (addresses from the buckets' contents), then `pad` dependent ALU steps standing in for the heap work, whose result names the
// next row.  DEPTH independent chains per wave (1 = what k_search_fast does; 2, 4 = what overlapping expansions would give).
template <int DEPTH>
__global__ __launch_bounds__(64) void k_chain(const uint8_t* __restrict__ codes, uint64_t nrows, const uint32_t* __restrict__ nbrs,
                                              uint64_t nnodes, uint32_t* tables, uint32_t tab_words, uint32_t iters,
                                              uint32_t rows_per_iter, uint32_t probes_per_iter, uint32_t pad, uint64_t* sink) {
    extern __shared__ unsigned char pad_lds[];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const uint32_t l4 = lane & 3, grp = lane >> 2;
    uint32_t* tab = tables + (size_t)wave * tab_words;
    uint64_t acc = 0;
    uint32_t dep[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) dep[d] = mix(wave * 0x9E3779B9u + 12345u + d * 0x51ed27u);
    for (uint32_t it = 0; it < iters; it += DEPTH) {
        uint32_t nb[DEPTH], pr[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // stage 1: the neighbor row of the node the last expansion chose
            const uint64_t node = ((uint64_t)dep[d] * nnodes) >> 32;
            nb[d] = nbrs[node * 64 + lane];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // stage 2: one bucket per neighbor id, a store for 61 % of them
            const uint32_t h = mix(nb[d] + dep[d] + lane * 0x85ebca6bu);
            pr[d] = h;
            if (lane < probes_per_iter) {
                const uint32_t slot = h & (tab_words - 1);
                const uint4 b = *reinterpret_cast<const uint4*>(tab + (slot & ~3u));
                pr[d] = h ^ ((b.x ^ b.y ^ b.z ^ b.w) & 1u);
                if ((h >> 24) < 156u) tab[slot] = (h >> 4) | 1u;
            }
        }
        uint32_t sum[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // stage 3: the code rows of the ids that were new
            sum[d] = 0;
            for (uint32_t p = 0; p < rows_per_iter; p += 16) {
                const uint32_t src = __shfl(pr[d], (int)(p + grp) & 63);
                const uint64_t row = ((uint64_t)mix(src ^ (p * 0x9E3779B1u)) * nrows) >> 32;
                const uint8_t* r = codes + row * 192 + 16u * l4;
                typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
                const v2u64 a = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r));
                const v2u64 b = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 64));
                const v2u64 c = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 128));
                sum[d] += __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // stage 4: the heap work (dependent ALU steps), then the next node is known
            uint32_t x = sum[d] + pr[d];
            for (uint32_t k = 0; k < pad; ++k) x = mix(x + k);
            x = (uint32_t)__shfl((int)x, 0);
            dep[d] = mix(x + it);
            acc += x;
        }
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}

template <int DEPTH>
static double run_chain(const uint8_t* codes, uint64_t nrows, const uint32_t* nbrs, uint64_t nnodes, uint32_t* tables,
                        uint32_t tab_words, uint32_t waves_per_cu, uint32_t iters, uint32_t pad, uint64_t* sink) {
    const uint32_t nwaves = 256 * waves_per_cu;
    const size_t lds = (160 * 1024) / waves_per_cu - 64;
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    k_fill<<<2048, 256>>>(tables, (size_t)nwaves * tab_words, 0xFFFFFFFFu);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<DEPTH>), dim3(nwaves), dim3(64), lds, 0, codes, nrows, nbrs, nnodes, tables, tab_words, iters, 32u, 50u,
                       pad, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = (double)nwaves * iters * 32;
    printf("chain depth=%d pad=%4u ALU steps              w/CU=%2u tabKB=%4u  %8.2f ms  rows %6.2f G/s (%6.0f GB/s alg)  %6.2f us per expansion\n",
           DEPTH, pad, waves_per_cu, tab_words / 256, ms, rows / ms / 1e6, rows * 192.0 / ms / 1e6, ms * 1e3 / iters * DEPTH);
    fflush(stdout);
    return ms;
}

struct Res { double ms; };

template <int MODE, int PROBE, int NT>
static double run(const char* name, const uint8_t* codes, uint64_t nrows, uint32_t row_bytes, uint32_t* tables, uint32_t tab_words,
                  uint32_t waves_per_cu, uint32_t iters, uint32_t rows_per_iter, uint32_t probes_per_iter, uint64_t* sink,
                  uint32_t fillv) {
    const uint32_t nwaves = 256 * waves_per_cu;
    const size_t lds = (160 * 1024) / waves_per_cu - 64;  // pins the number of resident waves per CU
    static bool attr[8][5][2];
    if (!attr[MODE][PROBE][NT]) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix<MODE, PROBE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[MODE][PROBE][NT] = true;
    }
    if (MODE & 2) {
        k_fill<<<2048, 256>>>(tables, (size_t)nwaves * tab_words, fillv);
        CK(hipDeviceSynchronize());
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<MODE, PROBE, NT>), dim3(nwaves), dim3(64), lds, 0, codes, nrows, row_bytes, tables, tab_words, iters,
                       rows_per_iter, probes_per_iter, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = (MODE & 1) ? (double)nwaves * iters * rows_per_iter : 0.0;
    const double probes = (MODE & 2) ? (double)nwaves * iters * probes_per_iter : 0.0;
    printf("%-44s w/CU=%2u rowB=%3u tabKB=%4u  %8.2f ms  rows %6.2f G/s (%6.0f GB/s alg)  probes %6.2f G/s\n", name, waves_per_cu,
           row_bytes, tab_words / 256, ms, rows / ms / 1e6, rows * 192.0 / ms / 1e6, probes / ms / 1e6);
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv) {
    const uint64_t nrows = 50000000ull;
    uint8_t* codes;
    CK(hipMalloc(&codes, nrows * 256));
    CK(hipMemset(codes, 0x5a, nrows * 256));
    uint32_t* tables;
    const size_t tab_max = (size_t)256 * 32 * 65536;  // up to 32 waves / CU x 128 KB
    CK(hipMalloc(&tables, tab_max * 4));
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    const uint32_t E = 0xFFFFFFFFu;
    const uint32_t it = 4000;
    if (argc > 1 && argv[1][0] == 'c') {  // "chain": the closed loop of one scan, see k_chain
        const uint64_t nnodes = 50000000ull;
        uint32_t* nbrs;
        CK(hipMalloc(&nbrs, nnodes * 64 * 4));
        k_fill<<<2048, 256>>>(nbrs, nnodes * 64, 0x1234567u);
        CK(hipDeviceSynchronize());
        const uint32_t itc = 2000;
        for (uint32_t w : {12u, 20u, 24u, 32u})
            for (uint32_t pad : {0u, 64u, 256u}) {
                run_chain<1>(codes, nrows, nbrs, nnodes, tables, 16384, w, itc, pad, sink);
                run_chain<2>(codes, nrows, nbrs, nnodes, tables, 16384, w, itc, pad, sink);
                run_chain<4>(codes, nrows, nbrs, nnodes, tables, 16384, w, itc, pad, sink);
            }
        return 0;
    }
    if (argc > 1) {  // table-size sweep of the non-atomic bucket scheme (16-byte load + 4-byte store for 61 % of the probes)
        for (uint32_t w : {20u, 12u}) {
            for (uint32_t tw : {1024u, 2048u, 4096u, 8192u, 16384u, 32768u, 65536u})
                run<3, 4, 1>("mix 50 bucket-load/store + 32 rows nt", codes, nrows, 192, tables, tw, w, it, 32, 50, sink, E);
            run<1, 0, 1>("rows only nt", codes, nrows, 192, tables, 1024, w, it, 32, 0, sink, E);
            run<2, 4, 1>("bucket-load/store only", codes, nrows, 192, tables, 2048, w, it, 0, 50, sink, E);
            run<2, 4, 1>("bucket-load/store only", codes, nrows, 192, tables, 16384, w, it, 0, 50, sink, E);
        }
        run<3, 4, 1>("mix 50 bucket-load/store + 32 rows nt", codes, nrows, 192, tables, 8192, 32, it, 32, 50, sink, E);
        run<3, 4, 1>("mix 50 bucket-load/store + 32 rows nt", codes, nrows, 192, tables, 16384, 32, it, 32, 50, sink, E);
        return 0;
    }
    // ---- rows only: stride, occupancy, non-temporal
    for (uint32_t w : {8u, 16u, 20u, 32u}) run<1, 0, 0>("rows only", codes, nrows, 192, tables, 16384, w, it, 32, 0, sink, E);
    run<1, 0, 0>("rows only, 256-B stride", codes, nrows, 256, tables, 16384, 20, it, 32, 0, sink, E);
    run<1, 0, 0>("rows only, 256-B stride", codes, nrows, 256, tables, 16384, 32, it, 32, 0, sink, E);
    run<1, 0, 1>("rows only, nontemporal", codes, nrows, 192, tables, 16384, 20, it, 32, 0, sink, E);
    run<1, 0, 0>("rows only, 1M rows (L2/MALL resident)", codes, 1000000, 192, tables, 16384, 20, it, 32, 0, sink, E);
    // ---- probes only: kind x table size x occupancy (footprint = waves x table)
    for (uint32_t tw : {4096u, 8192u, 16384u, 32768u}) {
        run<2, 0, 0>("probes only, CAS (mostly succeeds)", codes, nrows, 192, tables, tw, 20, it, 0, 50, sink, E);
        run<2, 3, 0>("probes only, CAS that fails", codes, nrows, 192, tables, tw, 20, it, 0, 50, sink, 0x11111111u);
        run<2, 1, 0>("probes only, load", codes, nrows, 192, tables, tw, 20, it, 0, 50, sink, E);
        run<2, 2, 0>("probes only, load then CAS if empty", codes, nrows, 192, tables, tw, 20, it, 0, 50, sink, E);
    }
    run<2, 0, 0>("probes only, CAS", codes, nrows, 192, tables, 16384, 8, it, 0, 50, sink, E);
    run<2, 0, 0>("probes only, CAS", codes, nrows, 192, tables, 16384, 32, it, 0, 50, sink, E);
    // ---- the search kernel's mix: 50 probes + 32 rows per expansion
    for (uint32_t w : {8u, 20u, 32u}) {
        run<3, 0, 0>("mix 50 CAS + 32 rows", codes, nrows, 192, tables, 16384, w, it, 32, 50, sink, E);
        run<3, 2, 0>("mix 50 load/CAS + 32 rows", codes, nrows, 192, tables, 16384, w, it, 32, 50, sink, E);
    }
    run<3, 0, 1>("mix 50 CAS + 32 rows nontemporal", codes, nrows, 192, tables, 16384, 20, it, 32, 50, sink, E);
    run<3, 2, 1>("mix 50 load/CAS + 32 rows nontemporal", codes, nrows, 192, tables, 16384, 20, it, 32, 50, sink, E);
    run<3, 0, 0>("mix 50 CAS + 32 rows, 16-KB tables", codes, nrows, 192, tables, 4096, 20, it, 32, 50, sink, E);
    run<3, 0, 0>("mix 50 CAS + 32 rows, 32-KB tables", codes, nrows, 192, tables, 8192, 20, it, 32, 50, sink, E);
    run<3, 0, 0>("mix 50 CAS + 32 rows, 256-B stride", codes, nrows, 256, tables, 16384, 20, it, 32, 50, sink, E);
    return 0;
}
