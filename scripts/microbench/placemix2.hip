// placemix2.hip — the search kernel's request mix (see placemix.hip) over a MATRIX of placements: K candidate allocations for the streamed
// arrays (code rows 9.6 GB, neighbor rows 12.8 GB) x M candidate allocations for the private state (tables + heap arrays).  Question
// (profiles/r05/s11: a process where every private-state placement is slow): is what matters the RELATION between where the streams
// live and where the private state lives — and can the index arrays be placed too?
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
    const uint8_t* codes2;  // non-null: odd rows are read from this second array (row interleaving over two allocations)
    const uint32_t* nbrs2;
    const uint8_t* codes;
    const uint32_t* nbrs;
    uint64_t nrows;
    uint8_t* tab_base;
    uint8_t* heap_base;
    uint32_t tab_bytes, heap_bytes, iters;
    uint64_t* sink;
};

__global__ __launch_bounds__(64) void k_placemix(Args a) {
    extern __shared__ unsigned char pad_lds[];
    const uint32_t lane = threadIdx.x, l4 = lane & 3, grp = lane >> 2;
    uint8_t* tab = a.tab_base + (size_t)blockIdx.x * a.tab_bytes;
    uint8_t* heap = a.heap_base + (size_t)blockIdx.x * a.heap_bytes;
    const uint32_t t16 = a.tab_bytes / 16, t2 = a.tab_bytes / 2, h8 = a.heap_bytes / 8;
    uint64_t acc = 0;
    uint32_t ctr = blockIdx.x * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < a.iters; ++it) {
        const uint32_t h = mix(ctr + lane * 0x85ebca6bu + it * 0xc2b2ae35u);
        const uint32_t hr = mix(ctr ^ (it * 0x9E3779B1u));
        const uint64_t nrow = ((uint64_t)hr * a.nrows) >> 32;
        if (lane < 50) acc += __builtin_nontemporal_load(((a.nbrs2 && (nrow & 1)) ? a.nbrs2 : a.nbrs) + nrow * 64 + lane);
        if (lane < 28) {
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)(uint32_t)(((uint64_t)h * t16) >> 32) * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (lane < 31 && acc != 0x123456789abcull)
            *reinterpret_cast<uint16_t*>(tab + (size_t)(uint32_t)(((uint64_t)(h ^ 0x5bd1e995u) * t2) >> 32) * 2) = (uint16_t)h;
        if (lane < 56) acc += *reinterpret_cast<const uint64_t*>(heap + (size_t)(uint32_t)(((uint64_t)(h * 0x9E3779B1u) * h8) >> 32) * 8);
        for (uint32_t p = 0; p < 2; ++p) {
            const uint32_t hc = mix(ctr ^ ((it * 64u + p * 16u + grp) * 0x9E3779B1u) ^ 0xabcdefu);
            const uint64_t row = ((uint64_t)hc * a.nrows) >> 32;
            const uint8_t* r = ((a.codes2 && (row & 1)) ? a.codes2 : a.codes) + row * 192 + 16u * l4;
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
    const int K = argc > 1 ? atoi(argv[1]) : 6, M = argc > 2 ? atoi(argv[2]) : 10;
    const uint32_t iters = 1200;
    const uint64_t nrows = 50000000ull;
    std::vector<uint8_t*> codes, priv;
    std::vector<uint32_t*> nbrs;
    // interleave the allocations so that stream sets and private chunks land all over the device
    for (int i = 0; i < K || i < M; ++i) {
        if (i < M) {
            uint8_t* p;
            CK(hipMalloc(&p, (size_t)2 << 30));
            priv.push_back(p);
        }
        if (i < K) {
            uint8_t* c;
            uint32_t* n;
            CK(hipMalloc(&c, nrows * 192));
            CK(hipMalloc(&n, nrows * 256));
            codes.push_back(c);
            nbrs.push_back(n);
        }
        if (i < M) {  // a filler that stays: spreads the candidates
            uint8_t* f;
            if (hipMalloc(&f, (size_t)6 << 30) != hipSuccess) (void)hipGetLastError();
        }
    }
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_placemix), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t waves = 256 * 24;
    const size_t lds = (160 * 1024) / 24 - 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int c2 = -1, n2 = -1;
    auto run = [&](int ci, int ni, int pi, int hi) -> float {
        Args a{c2 >= 0 ? codes[c2] : nullptr, n2 >= 0 ? nbrs[n2] : nullptr, codes[ci], nbrs[ni], nrows, priv[pi], priv[hi] + (pi == hi ? ((size_t)1 << 30) : 0), 36864, 46368, iters, sink};
        float best = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {
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
    printf("addresses: ");
    for (int i = 0; i < K; ++i) printf("codes%d %p nbrs%d %p  ", i, (void*)codes[i], i, (void*)nbrs[i]);
    printf("\n           ");
    for (int j = 0; j < M; ++j) printf("priv%d %p ", j, (void*)priv[j]);
    printf("\nrows: streams (codes i + nbrs i); columns: private state on chunk j (tables + heaps)\n");
    for (int i = 0; i < K; ++i) {
        printf("streams %d:", i);
        for (int j = 0; j < M; ++j) printf(" %6.2f", run(i, i, j, j));
        printf("\n");
        fflush(stdout);
    }
    printf("split streams: codes i, nbrs i' (private on chunk 0 / chunk M-1)\n");
    for (int i = 0; i < K; ++i) {
        printf("codes %d:", i);
        for (int i2 = 0; i2 < K; ++i2) printf("  nbrs%d %6.2f/%6.2f", i2, run(i, i2, 0, 0), run(i, i2, M - 1, M - 1));
        printf("\n");
        fflush(stdout);
    }
    printf("split private: tables on chunk j, heaps on chunk j' (streams 0)\n");
    for (int j = 0; j < M; ++j) {
        printf("tables %d:", j);
        for (int j2 = 0; j2 < M; ++j2) printf(" %6.2f", run(0, 0, j, j2));
        printf("\n");
        fflush(stdout);
    }
    printf("row-interleaved streams: even rows from set i, odd rows from set i' (codes AND nbrs); private: tables on chunk j / heaps on chunk j'\n");
    const int pj[4][2] = {{0, 0}, {3, 3}, {0, 3}, {3, 0}};
    for (int i = 0; i < K; ++i)
        for (int i2 = i; i2 < K; ++i2) {
            c2 = i2;
            n2 = i2;
            printf("sets %d+%d:", i, i2);
            for (auto& pp : pj) printf("  t%d/h%d %6.2f", pp[0], pp[1], run(i, i, pp[0], pp[1]));
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
