// placemix.hip — does the PLACEMENT of the search kernel's private state matter for its traffic MIX?
//
// k_search_fast runs the same 262 144-scan launch in 152, 156 or 170 ms depending on where its 0.7 GB of per-workgroup dedup tables and
// heap spill arrays were allocated (profiles/r05/s4_placement_map_50m.txt), while the private-state requests alone (vs_ws_probe) cost the
// same everywhere.  This microbenchmark issues the kernel's whole request mix — per "expansion": one random 256-byte neighbor row, two
// passes of 16 random 192-byte code rows (non-temporal), 28 random 16-byte loads + 31 random 4-byte stores in the workgroup's table
// region, 56 random 8-byte loads in its heap region — with the private regions on each of many 1-GB chunks in turn, next to arrays the
// size of the 50M index.  mode 0: region index = blockIdx (what the library does); mode 1: region index = (hardware CU, local rank), i.e.
// the 24 workgroups of a CU use adjacent regions.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

struct Args {
    const uint8_t* codes;
    const uint32_t* nbrs;
    uint64_t nrows;
    uint8_t* tab_base;
    uint8_t* heap_base;
    uint32_t tab_bytes, heap_bytes, iters, mode;
    uint32_t* cu_ctr;   // [4096] per hardware CU: workgroups that have arrived (mode 1)
    uint32_t* hw_hist;  // [4096] (diagnostics) workgroups seen per hardware CU id
    uint64_t* sink;
};

__device__ __forceinline__ uint32_t hw_cu_linear() {
    // HW_REG_HW_ID (id 4): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; HW_REG_XCC_ID (id 20): xcc_id[3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return (((xcc & 15u) * 8u + se) * 2u + sh) * 16u + cu;  // < 4096
}

__global__ __launch_bounds__(64) void k_placemix(Args a) {
    extern __shared__ unsigned char pad_lds[];
    const uint32_t lane = threadIdx.x, l4 = lane & 3, grp = lane >> 2;
    uint32_t region = blockIdx.x;
    {
        uint32_t r = 0;
        if (lane == 0) {
            const uint32_t cu = hw_cu_linear();
            const uint32_t rank = atomicAdd(&a.cu_ctr[cu], 1u);
            atomicAdd(&a.hw_hist[cu], 1u);
            r = cu * 32u + (rank & 31u);
        }
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        if (a.mode == 1) region = r;
    }
    uint8_t* tab = a.tab_base + (size_t)region * a.tab_bytes;
    uint8_t* heap = a.heap_base + (size_t)region * a.heap_bytes;
    const uint32_t t16 = a.tab_bytes / 16, t4 = a.tab_bytes / 4, h8 = a.heap_bytes / 8;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < a.iters; ++it) {
        const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
        // neighbor row (wave-uniform row, lanes 0..49)
        const uint32_t hr = mix(ctr ^ (it * 0x9E3779B1u));
        const uint64_t nrow = ((uint64_t)hr * a.nrows) >> 32;
        if (lane < 50) acc += __builtin_nontemporal_load(a.nbrs + nrow * 64 + lane);
        // dedup: 28 group loads + 31 stores
        if (lane < 28) {
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)(uint32_t)(((uint64_t)h * t16) >> 32) * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (lane < 31 && acc != 0x123456789abcull)
            *reinterpret_cast<uint32_t*>(tab + (size_t)(uint32_t)(((uint64_t)(h ^ 0x5bd1e995u) * t4) >> 32) * 4) = h;
        // heap: 56 child-pair loads
        if (lane < 56) acc += *reinterpret_cast<const uint64_t*>(heap + (size_t)(uint32_t)(((uint64_t)(h * 0x9E3779B1u) * h8) >> 32) * 8);
        // code rows: 2 passes x 16 rows
        for (uint32_t p = 0; p < 2; ++p) {
            const uint32_t hc = mix(ctr ^ ((it * 64u + p * 16u + grp) * 0x9E3779B1u) ^ 0xabcdefu);
            const uint64_t row = ((uint64_t)hc * a.nrows) >> 32;
            const uint8_t* r = a.codes + row * 192 + 16u * l4;
            const v2u64 x = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r));
            const v2u64 y = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 64));
            const v2u64 z = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 128));
            acc += __popcll(x.x) + __popcll(x.y) + __popcll(y.x) + __popcll(y.y) + __popcll(z.x) + __popcll(z.y);
        }
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) a.sink[0] = acc;
}

int main(int argc, char** argv) {
    const int nchunks = argc > 1 ? atoi(argv[1]) : 40;
    const uint32_t iters = argc > 2 ? atoi(argv[2]) : 1500;
    const size_t filler_gb = argc > 3 ? atoi(argv[3]) : 150;  // stands for the vector column (153.6 GB at 50M x 768)
    const uint64_t nrows = 50000000ull;
    uint8_t *codes, *filler = nullptr;
    uint32_t* nbrs;
    CK(hipMalloc(&codes, nrows * 192));
    CK(hipMalloc(&nbrs, nrows * 256));
    if (filler_gb) CK(hipMalloc(&filler, filler_gb << 30));
    CK(hipMemset(codes, 0x5a, nrows * 192));
    CK(hipMemset(nbrs, 0x11, nrows * 256));
    uint32_t *cu_ctr, *hw_hist;
    uint64_t* sink;
    CK(hipMalloc(&cu_ctr, 4096 * 4));
    CK(hipMalloc(&hw_hist, 4096 * 4));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(hw_hist, 0, 4096 * 4));
    const uint32_t tab_bytes = 58752, heap_bytes = 53248;
    const size_t chunk = (size_t)2 << 30;  // room for 4096 x 32 regions of both kinds in mode 1 is NOT needed: only touched regions matter
    std::vector<uint8_t*> chunks;
    for (int i = 0; i < nchunks; ++i) {
        uint8_t* p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) break;
        chunks.push_back(p);
    }
    printf("%zu chunks of %zu MB\n", chunks.size(), chunk >> 20);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_placemix), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t waves = 256 * 24;
    const size_t lds = (160 * 1024) / 24 - 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](uint8_t* tb, uint8_t* hb, uint32_t mode) -> float {
        Args a{codes, nbrs, nrows, tb, hb, tab_bytes, heap_bytes, iters, mode, cu_ctr, hw_hist, sink};
        float best = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(cu_ctr, 0, 4096 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_placemix, dim3(waves), dim3(64), lds, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    // mode 0: tables at the chunk's start, heaps 361 MB in (the library's slab layout).  mode 1 needs 4096 x 32 regions of address space
    // per kind (7.3 + 6.6 GB): tables on chunks[i .. i+3], heaps on chunks[i+4 .. i+7] are NOT contiguous, so mode 1 uses one big block.
    for (size_t i = 0; i < chunks.size(); ++i) {
        const float ms = run(chunks[i], chunks[i] + ((size_t)361 << 20), 0);
        printf("mode0 chunk %2zu %p  %8.3f ms\n", i, (void*)chunks[i], ms);
        fflush(stdout);
    }
    // independent placements: tables on chunk i, heaps on chunk j
    for (size_t i = 0; i + 1 < chunks.size() && i < 12; i += 2) {
        const float ms = run(chunks[i], chunks[i + 1], 0);
        printf("mode0 tables on chunk %2zu, heaps on chunk %2zu  %8.3f ms\n", i, i + 1, ms);
    }
    // mode 1: CU-local regions inside one block of 16 GB (tables in the first half)
    const size_t MB = (size_t)1 << 20;
    const size_t shifts[] = {0, 2, 4, 6, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 3072};
    for (int k = 0; k < 4; ++k) {
        uint8_t* big = nullptr;
        if (hipMalloc(&big, (size_t)16 << 30) != hipSuccess) break;
        const float m1 = run(big, big + ((size_t)8 << 30), 1);
        const float m0 = run(big, big + ((size_t)8 << 30), 0);
        printf("block %d %p: mode1 (CU-local regions) %8.3f ms   mode0 (same block) %8.3f ms\n", k, (void*)big, m1, m0);
        printf("  tables shifted (heaps at +8 GB):");
        for (size_t sh : shifts) printf(" +%zuM %.2f", sh, run(big + sh * MB, big + ((size_t)8 << 30), 0));
        printf("\n  heaps shifted (tables at +0):    ");
        for (size_t sh : shifts) printf(" +%zuM %.2f", sh, run(big, big + ((size_t)8 << 30) + sh * MB, 0));
        printf("\n  both shifted (heaps at tables + 361 MB):");
        for (size_t sh : shifts) printf(" +%zuM %.2f", sh, run(big + sh * MB, big + sh * MB + 361 * MB, 0));
        printf("\n");
        fflush(stdout);
        chunks.push_back(big);  // (kept: the next block lands elsewhere)
    }
    std::vector<uint32_t> hist(4096);
    CK(hipMemcpy(hist.data(), hw_hist, 4096 * 4, hipMemcpyDeviceToHost));
    uint32_t used = 0, mx = 0;
    for (uint32_t v : hist) {
        used += v != 0;
        if (v > mx) mx = v;
    }
    printf("hardware CU ids seen: %u distinct (of 4096 encodable); ", used);
    uint32_t xccs = 0, ses = 0;
    for (uint32_t i = 0; i < 4096; ++i)
        if (hist[i]) {
            xccs |= 1u << (i >> 8);
            ses |= 1u << ((i >> 5) & 7);
        }
    printf("xcc mask %#x se mask %#x\n", xccs, ses);
    return 0;
}
