// regionbw.hip — is HBM bandwidth uniform over device memory for random row gathers?  (profiles/r05/s7: the search kernel's private state
// runs 5 % faster in some physical regions than in others, in contiguous stretches of tens of GB.)  If the address space is interleaved
// over the HBM stacks only within large regions, a gather confined to one region sees a fraction of the chip's bandwidth and the same
// gather spread over several regions sees more.  24 chunks of 8 GB; random 192-byte rows (4 lanes x 3 x 16 B, non-temporal) from
// (a) one chunk at a time, (b) pairs, (c) all chunks; 24 single-wave workgroups per CU.
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
struct Bases { const uint8_t* p[32]; uint32_t n; };

__global__ __launch_bounds__(64) void k_gather(Bases b, uint64_t rows_per_chunk, uint32_t iters, uint64_t* sink) {
    extern __shared__ unsigned char pad_lds[];
    const uint32_t lane = threadIdx.x, l4 = lane & 3, grp = lane >> 2;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t h = mix(ctr ^ ((it * 64u + grp) * 0x9E3779B1u));
        const uint32_t h2 = mix(h ^ 0x5bd1e995u);
        const uint8_t* base = b.p[h2 % b.n];
        const uint64_t row = ((uint64_t)h * rows_per_chunk) >> 32;
        const uint8_t* r = base + row * 192 + 16u * l4;
        const v2u64 x = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r));
        const v2u64 y = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 64));
        const v2u64 z = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(r + 128));
        acc += __popcll(x.x) + __popcll(x.y) + __popcll(y.x) + __popcll(y.y) + __popcll(z.x) + __popcll(z.y);
        ctr += 0x632be5abu;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const v2u64* __restrict__ p, size_t n16, uint64_t* sink) {
    uint64_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const v2u64 v = __builtin_nontemporal_load(p + i);
        acc += v.x ^ v.y;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int nchunks = argc > 1 ? atoi(argv[1]) : 24;
    const size_t chunk = (size_t)8 << 30;
    std::vector<uint8_t*> ch;
    for (int i = 0; i < nchunks; ++i) {
        uint8_t* p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) break;
        CK(hipMemset(p, 0x5a, chunk));
        ch.push_back(p);
    }
    printf("%zu chunks of 8 GB\n", ch.size());
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t iters = 3000;
    auto gather = [&](const std::vector<int>& ids, uint32_t wpc) -> double {
        Bases b;
        b.n = (uint32_t)ids.size();
        for (size_t i = 0; i < ids.size(); ++i) b.p[i] = ch[ids[i]];
        const uint32_t waves = 256 * wpc;
        const size_t lds = (160 * 1024) / wpc - 64;
        float best = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3(waves), dim3(64), lds, 0, b, chunk / 192, iters, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return (double)waves * iters * 16 * 192 / best / 1e6;  // GB/s
    };
    for (size_t i = 0; i < ch.size(); ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, (const v2u64*)ch[i], chunk / 16, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("chunk %2zu %p: stream %6.0f GB/s   gather 24 w/CU %6.0f GB/s   32 w/CU %6.0f GB/s\n", i, (void*)ch[i], chunk / ms / 1e6, gather({(int)i}, 24),
               gather({(int)i}, 32));
        fflush(stdout);
    }
    const int n = (int)ch.size();
    for (int d : {1, 2, 4, 8, 12}) {
        if (d >= n) break;
        printf("pairs at distance %2d:", d);
        for (int i = 0; i + d < n; i += std::max(1, n / 6)) printf(" (%d,%d) %5.0f", i, i + d, gather({i, i + d}, 32));
        printf("\n");
    }
    std::vector<int> all, quarter, every3;
    for (int i = 0; i < n; ++i) {
        all.push_back(i);
        if (i % 4 == 0) quarter.push_back(i);
        if (i % 3 == 0) every3.push_back(i);
    }
    printf("all %d chunks: 24 w/CU %6.0f GB/s  32 w/CU %6.0f GB/s;  every 4th: %6.0f;  every 3rd: %6.0f\n", n, gather(all, 24), gather(all, 32), gather(quarter, 32),
           gather(every3, 32));
    return 0;
}
