// TEST INFRASTRUCTURE ONLY — a wave64 lockstep interpreter for this repository's own HIP kernels.
//
// This header shadows <hip/hip_runtime.h> when tests/emu/Makefile compiles the UNMODIFIED product sources
// (pgvectorscale_amd/csrc/*.hip) as host C++ into tests/emu/libvsgpu_emu.so.  Every GPU thread becomes a fiber; the 64
// fibers of a wave run until their next cross-lane operation (ballot, readlane, DPP, bpermute, shuffle, wave barrier) or
// __syncthreads, exchange values there, and continue — so data-dependent control flow, LDS data structures and atomics
// behave as on the hardware as long as cross-lane operations are reached in wave-uniform control flow (violations are
// detected and abort with the two source lines involved).  It exists so that kernel SOURCE changes can be checked
// against the oracle in a container without a GPU (tests/test_emu.py); it is not a product path, is never loaded by
// pgvectorscale_amd/, and proves nothing about performance, register pressure or memory-model races.
#pragma once
#include <x86intrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>

#define VS_WAVE64_EMULATOR 1

// ---- language surface -----------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local  // one block at a time per host thread; block-scope => static storage per host thread

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }

namespace emu {
struct Idx { unsigned x, y, z; };
struct Wave;
struct Block;
struct Fiber {
    void* sp;
    Idx tid, bid, bdim, gdim;
    int lane;        // lane inside the wave
    Wave* wave;
    Block* block;
    int state;       // 0 runnable, 1 waiting on the wave, 2 waiting on the block, 3 done
    uint32_t wait_gen;
    char* stack;
};
struct Snap {
    const uint64_t* vals;  // value published by each lane (undefined where !active)
    uint64_t active;       // lanes that took part
};
extern thread_local Fiber* cur;
Snap wave_exchange(uint64_t v, const char* file, int line);
void block_barrier(const char* file, int line);
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*thunk)(void*), void* ctx);
static inline int lane_id() { return cur->lane; }
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)

// ---- cross-lane operations (wave64) ---------------------------------------------------------------------------------
#define EMU_X(v) emu::wave_exchange((uint64_t)(v), __FILE__, __LINE__)

static inline uint64_t emu_ballot(bool p, const char* f, int l) {
    const emu::Snap s = emu::wave_exchange(p ? 1u : 0u, f, l);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i)
        if (((s.active >> i) & 1) && s.vals[i]) m |= 1ull << i;
    return m;
}
#define __ballot(p) emu_ballot((p), __FILE__, __LINE__)
// a wave-uniform lane mask back as a per-lane condition (s_and_saveexec on the mask itself: no vector instruction on the device)
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((uint64_t)(m) >> emu::lane_id()) & 1ull) != 0)

static inline int emu_readfirstlane(int v, const char* f, int l) {
    const emu::Snap s = emu::wave_exchange((uint32_t)v, f, l);
    return (int)(uint32_t)s.vals[__builtin_ctzll(s.active)];
}
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((v), __FILE__, __LINE__)

static inline int emu_readlane(int v, int lane, const char* f, int l) {
    const emu::Snap s = emu::wave_exchange((uint32_t)v, f, l);
    return ((s.active >> (lane & 63)) & 1) ? (int)(uint32_t)s.vals[lane & 63] : 0;
}
#define __builtin_amdgcn_readlane(v, lane) emu_readlane((v), (lane), __FILE__, __LINE__)

// DPP controls used by this code base: quad_perm (0x00-0xFF), row_shl:n (0x101-0x10F), row_shr:n (0x111-0x11F),
// row_ror:n (0x121-0x12F), row_bcast:15 / 31 (0x142, 0x143), wave_shl:1 (0x130),
// wave_shr:1 (0x138).  A lane whose source is out of range or inactive keeps `old` (or gets 0 with bound_ctrl).
static inline int emu_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, const char* f, int l) {
    const emu::Snap s = emu::wave_exchange((uint32_t)src, f, l);
    const int me = emu::lane_id();
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 0xF; from = (me & 15) + n < 16 ? me + n : -1; }      // row_shl:n
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 0xF; from = (me & 15) >= n ? me - n : -1; }           // row_shr:n
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 0xF; from = (me & ~15) | (((me & 15) - n) & 15); }    // row_ror:n
    else if (ctrl == 0x142) from = me >= 16 ? ((me >> 4) - 1) * 16 + 15 : -1;  // row_bcast:15 (lane 15 of a row to every lane of the next row)
    else if (ctrl == 0x143) from = me >= 32 ? 31 : -1;                         // row_bcast:31
    else if (ctrl == 0x130) from = me + 1 < 64 ? me + 1 : -1;
    else if (ctrl == 0x138) from = me >= 1 ? me - 1 : -1;
    else { fprintf(stderr, "emu: DPP control 0x%x not implemented (%s:%d)\n", ctrl, f, l); abort(); }
    const bool row_ok = (row_mask >> (me >> 4)) & 1, bank_ok = (bank_mask >> ((me >> 2) & 3)) & 1;
    if (!row_ok || !bank_ok) return old;
    if (from < 0 || !((s.active >> from) & 1)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)s.vals[from];
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_dpp((old), (src), (ctrl), (rm), (bm), (bc), __FILE__, __LINE__)
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) emu_dpp(0, (src), (ctrl), (rm), (bm), (bc), __FILE__, __LINE__)

static inline int emu_bpermute(int addr, int src, const char* f, int l) {
    const emu::Snap s = emu::wave_exchange((uint32_t)src, f, l);
    const int from = ((uint32_t)addr >> 2) & 63;
    return ((s.active >> from) & 1) ? (int)(uint32_t)s.vals[from] : 0;
}
#define __builtin_amdgcn_ds_bpermute(addr, src) emu_bpermute((addr), (src), __FILE__, __LINE__)

template <class T>
static inline T emu_shfl(T v, int from, const char* f, int l) {
    static_assert(sizeof(T) <= 8, "shuffle of a wide type");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const emu::Snap s = emu::wave_exchange(bits, f, l);
    from &= 63;
    const uint64_t r = ((s.active >> from) & 1) ? s.vals[from] : bits;
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
// (the optional width argument is accepted and must be the full wave)
#define __shfl(v, lane, ...) emu_shfl((v), (lane), __FILE__, __LINE__)
#define __shfl_xor(v, mask, ...) emu_shfl((v), emu::lane_id() ^ (mask), __FILE__, __LINE__)

#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_exchange(0, __FILE__, __LINE__))
// A fence executed by a wave orders the memory operations of ALL its lanes: in lockstep every lane's earlier stores are
// issued before any lane's later access.  Lane-after-lane execution only has that property at a rendezvous, so fences of
// workgroup / agent scope are one (wavefront-scope fences only ever bracket wave_barrier, which already is).
#define __builtin_amdgcn_fence(order, scope) \
    (((scope)[0] == 'w' && (scope)[1] == 'a') ? (void)0 : (void)emu::wave_exchange(0, __FILE__, __LINE__))
#define __syncthreads() emu::block_barrier(__FILE__, __LINE__)

// ---- scalar intrinsics ----------------------------------------------------------------------------------------------
#define __popcll(x) __builtin_popcountll((unsigned long long)(x))
#define __popc(x) __builtin_popcount((unsigned)(x))
#define __builtin_readcyclecounter() ((uint64_t)__rdtsc())
static inline uint64_t wall_clock64() {  // 100 MHz on the device
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10u;
}
#define __builtin_nontemporal_load(p) (*(p))
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline uint32_t __float_as_uint(float f) { uint32_t i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(uint32_t i) { float f; memcpy(&f, &i, 4); return f; }

template <class T>
static inline T atomicCAS(T* p, T cmp, T val) {
    __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}
template <class T, class U>
static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicMax(T* p, U v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <class T, class U>
static inline T atomicMin(T* p, U v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
using std::max;
using std::min;

// ---- runtime API (synchronous; "device" memory is host memory) -------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorPeerAccessAlreadyEnabled = 704 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct emu_stream;
typedef emu_stream* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; unsigned lag = 0; };
typedef emu_event* hipEvent_t;
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError"); }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// VS_EMU_DEVICES=N: N "devices" that are all this host's memory (the multi-device entry points: one context per device, peer copies)
static inline hipError_t hipGetDeviceCount(int* n) {
    const char* e = getenv("VS_EMU_DEVICES");
    *n = e && atoi(e) > 0 ? atoi(e) : 1;
    return hipSuccess;
}
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "wave64 lockstep emulator (host CPU)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950-emulated");
    p->multiProcessorCount = 4;
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
// "Device" allocations end right before an inaccessible guard page, so an access past the end of a buffer faults here
// the way it would fault (or silently corrupt a neighbour) on the GPU.  256-byte granularity like hipMalloc.
namespace emu {
void* guarded_alloc(size_t bytes);
void guarded_free(void* p);
}  // namespace emu
template <class T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
    void* q = emu::guarded_alloc(bytes);
    if (!q) return hipErrorOutOfMemory;
    *p = static_cast<T*>(q);
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) { emu::guarded_free(p); return hipSuccess; }
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
static inline hipError_t hipHostFree(void* p) { emu::guarded_free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
static inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)6 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event(); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->t = std::chrono::steady_clock::now();
    // (work runs to completion inside the launch call here; VS_EMU_EVENT_LAG=n makes the first n queries of a recorded event answer "not
    // ready", so that the callers' paths for work still in flight are walked too)
    const char* lag = getenv("VS_EMU_EVENT_LAG");
    e->lag = lag ? (unsigned)atoi(lag) : 0u;
    return hipSuccess;
}
static inline hipError_t hipEventQuery(hipEvent_t e) {
    if (e->lag) {
        e->lag--;
        return (hipError_t)hipErrorNotReady;
    }
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) { e->lag = 0; return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// two workgroups per "CU": a persistent grid of 8 workgroups, so every workgroup of a test batch runs several scans
template <class K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }

// ---- kernel launch ---------------------------------------------------------------------------------------------------
namespace emu {
template <class... P>
struct LaunchCtx {
    void (*k)(P...);
    std::tuple<typename std::decay<P>::type...> args;
    static void thunk(void* self) {
        auto* c = static_cast<LaunchCtx*>(self);
        auto copy = c->args;  // kernel parameters are per-thread copies
        std::apply(c->k, std::move(copy));
    }
};
template <class... P, class... A>
static inline void launch(void (*k)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t, A&&... a) {
    LaunchCtx<P...> c{k, std::tuple<typename std::decay<P>::type...>(std::forward<A>(a)...)};
    run_grid(grid, block, lds, &LaunchCtx<P...>::thunk, &c);
}
}  // namespace emu
#define hipLaunchKernelGGL(k, grid, block, lds, stream, ...) emu::launch((k), (grid), (block), (lds), (stream), __VA_ARGS__)
