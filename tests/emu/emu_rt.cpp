// TEST INFRASTRUCTURE ONLY — scheduler of the wave64 lockstep interpreter (see fake/hip/hip_runtime.h).
//
// One GPU thread = one fiber with its own stack; the fibers of a workgroup are scheduled round-robin on ONE host thread
// and only switch at cross-lane operations / barriers, so everything between two such points runs lane after lane — the
// same result as lockstep execution provided lanes do not communicate through memory without a wave barrier in between
// (the product kernels put wave_sync() there because the compiler needs it too).  Workgroups are independent and are
// spread over host threads (VS_EMU_THREADS, default: all cores); global atomics are real atomics.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <ucontext.h>

#include <mutex>
#include <thread>

namespace emu {

thread_local Fiber* cur = nullptr;

struct Wave {
    uint64_t vals[2][64];
    uint64_t snap_active[2];
    uint64_t arrived = 0, live = 0;
    uint32_t gen = 0;
    const char* file = nullptr;
    int line = 0;
};
struct Block {
    uint32_t live = 0, arrived = 0, gen = 0;
    const char* file = nullptr;
    int line = 0;
};

}  // namespace emu

// the dynamic LDS segment every kernel declares as `extern __shared__ unsigned char smem[]`
thread_local __attribute__((aligned(64))) unsigned char smem[160 * 1024];

// void emu_switch(void** save_sp, void* load_sp): callee-saved registers + stack pointer (System V x86-64)
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");

namespace emu {

// ---- guarded "device" memory ------------------------------------------------------------------------------------------
static std::mutex g_alloc_mu;
struct AllocInfo { void* base; size_t map_bytes; };
static std::vector<std::pair<void*, AllocInfo>> g_allocs;

// VS_EMU_ARENA=<MiB>: device memory comes out of ONE arena instead of a guarded mapping per allocation — freed blocks keep their
// address range and their contents, nothing is filled on reuse.  That is how a device allocator behaves and the guarded mode does
// not: hipFree + a larger hipMalloc can hand back the SAME address, with the old block's bytes at the front and whatever other
// freed buffers left behind in the extension (what a "was the buffer reallocated?" test by pointer comparison does not see).
struct ArenaSeg { size_t off, size; bool used; };
static char* g_arena = nullptr;
static size_t g_arena_bytes = 0;
static std::vector<ArenaSeg> g_segs;  // the blocks handed out so far, live or free

static bool arena_mode() {
    static const size_t mib = [] {
        const char* e = getenv("VS_EMU_ARENA");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0;
    }();
    if (!mib) return false;
    if (!g_arena) {
        g_arena_bytes = mib << 20;
        g_arena = static_cast<char*>(mmap(nullptr, g_arena_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (g_arena == MAP_FAILED) {
            fprintf(stderr, "emu: cannot map a %zu MiB arena\n", mib);
            abort();
        }
        // never-used device memory is not zero either: small pseudo-random words, the kind other kernels leave behind
        uint64_t x = 0x9E3779B97F4A7C15ull;
        uint32_t* w = reinterpret_cast<uint32_t*>(g_arena);
        for (size_t i = 0; i < g_arena_bytes / 4; ++i) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            w[i] = (uint32_t)(x >> 16) & ((i & 3) == 0 ? 0xFFFFFFFFu : ((i & 3) == 1 ? 0x00000FFFu : ((i & 3) == 2 ? 0x000FFFFFu : 0x0FFFFFFFu)));
        }
    }
    return true;
}

// granules of 2 MiB, like a device allocator that maps memory in large fragments: a freed block's address range is kept and handed
// to the next request that fits it, most recently freed first
static size_t g_arena_top = 0;
static std::vector<size_t> g_free_order;  // indices into g_segs, oldest first

static void* arena_alloc(size_t bytes) {
    const size_t gran = (size_t)2 << 20;
    const size_t rounded = (std::max<size_t>(bytes, 1) + gran - 1) / gran * gran;
    for (size_t k = g_free_order.size(); k-- > 0;) {
        ArenaSeg& sg = g_segs[g_free_order[k]];
        if (sg.size < rounded) continue;
        sg.used = true;
        g_free_order.erase(g_free_order.begin() + (long)k);
        return g_arena + sg.off;
    }
    if (g_arena_top + rounded > g_arena_bytes) return nullptr;
    g_segs.push_back({g_arena_top, rounded, true});
    g_arena_top += rounded;
    return g_arena + g_segs.back().off;
}

static bool arena_free(void* p) {
    const size_t off = (size_t)(static_cast<char*>(p) - g_arena);
    for (size_t i = 0; i < g_segs.size(); ++i) {
        if (g_segs[i].off != off || !g_segs[i].used) continue;
        g_segs[i].used = false;
        g_free_order.push_back(i);
        return true;
    }
    return false;
}

void* guarded_alloc(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        if (arena_mode()) return arena_alloc(bytes);
    }
    const size_t page = 4096;
    const size_t rounded = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    const size_t body = (rounded + page - 1) / page * page;
    const size_t map_bytes = body + 2 * page;
    char* base = static_cast<char*>(mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (base == MAP_FAILED) return nullptr;
    mprotect(base, page, PROT_NONE);                // below the buffer
    mprotect(base + page + body, page, PROT_NONE);  // right after its last 256-byte unit
    char* p = base + page + (body - rounded);
    // device memory is not zero initialised: 0xA5 by default; VS_EMU_POISON=<byte, hex> picks another filler and VS_EMU_POISON=rand
    // a different pseudo-random word for every 4 bytes (what a hipMalloc that recycles another kernel's memory hands out)
    {
        const char* e = getenv("VS_EMU_POISON");
        if (e && !strcmp(e, "rand")) {
            static std::atomic<uint64_t> seq{0x9E3779B97F4A7C15ull};
            uint64_t x = seq.fetch_add(0xD1B54A32D192ED03ull);
            uint32_t* w = reinterpret_cast<uint32_t*>(p);
            for (size_t i = 0; i < rounded / 4; ++i) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                w[i] = (uint32_t)(x >> 16) & ((i & 3) == 0 ? 0xFFFFFFFFu : ((i & 3) == 1 ? 0x00000FFFu : ((i & 3) == 2 ? 0x000FFFFFu : 0x0FFFFFFFu)));
            }
        } else {
            memset(p, e ? (int)strtoul(e, nullptr, 16) : 0xA5, rounded);
        }
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_allocs.push_back({p, AllocInfo{base, map_bytes}});
    return p;
}

void guarded_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    if (g_arena && static_cast<char*>(p) >= g_arena && static_cast<char*>(p) < g_arena + g_arena_bytes) {
        if (arena_free(p)) return;
        fprintf(stderr, "emu: hipFree of a pointer hipMalloc never returned (%p)\n", p);
        abort();
    }
    for (size_t i = 0; i < g_allocs.size(); ++i)
        if (g_allocs[i].first == p) {
            munmap(g_allocs[i].second.base, g_allocs[i].second.map_bytes);
            g_allocs[i] = g_allocs.back();
            g_allocs.pop_back();
            return;
        }
    fprintf(stderr, "emu: hipFree of a pointer hipMalloc never returned (%p)\n", p);
    abort();
}

}  // namespace emu
// test hook (tests/test_emu.py, device-memory hygiene): device + pinned allocations that are live right now
extern "C" size_t vs_emu_live_allocations(void) {
    std::lock_guard<std::mutex> lk(emu::g_alloc_mu);
    size_t n = emu::g_allocs.size();
    for (const auto& sg : emu::g_segs) n += sg.used ? 1 : 0;
    return n;
}
namespace emu {

static constexpr size_t kStackBytes = 512 * 1024;
static constexpr int kMaxThreadsPerBlock = 1024;

struct Worker {
    void* sched_sp = nullptr;
    char* stacks = nullptr;  // kMaxThreadsPerBlock stacks, committed lazily
    void (*thunk)(void*) = nullptr;
    void* ctx = nullptr;
    ~Worker() {
        if (stacks) munmap(stacks, kStackBytes * kMaxThreadsPerBlock);
    }
};
static thread_local Worker tl_worker;

[[noreturn]] static void die(const char* what, const char* f1, int l1, const char* f2, int l2) {
    fprintf(stderr, "emu: %s\n  at %s:%d\n  vs %s:%d\n", what, f1 ? f1 : "?", l1, f2 ? f2 : "?", l2);
    abort();
}

static void yield_to_scheduler() { emu_switch(&cur->sp, tl_worker.sched_sp); }

static void wave_release(Wave* w) {
    w->snap_active[w->gen & 1] = w->arrived;
    w->arrived = 0;
    w->gen++;
}

Snap wave_exchange(uint64_t v, const char* file, int line) {
    Fiber* f = cur;
    Wave* w = f->wave;
    const uint32_t g = w->gen;
    const int buf = g & 1;
    if (w->arrived == 0) {
        w->file = file;
        w->line = line;
    } else if (w->line != line || w->file != file) {
        die("lanes of one wave reached different cross-lane operations (divergent control flow around a wave collective)", file, line,
            w->file, w->line);
    }
    w->vals[buf][f->lane] = v;
    w->arrived |= 1ull << f->lane;
    if (w->arrived == w->live) {
        wave_release(w);
    } else {
        f->state = 1;
        f->wait_gen = g;
        yield_to_scheduler();
    }
    return Snap{w->vals[buf], w->snap_active[buf]};
}

void block_barrier(const char* file, int line) {
    Fiber* f = cur;
    Block* b = f->block;
    const uint32_t g = b->gen;
    if (b->arrived == 0) {
        b->file = file;
        b->line = line;
    } else if (b->line != line || b->file != file) {
        die("threads of one workgroup reached different __syncthreads()", file, line, b->file, b->line);
    }
    b->arrived++;
    if (b->arrived == b->live) {
        b->arrived = 0;
        b->gen++;
    } else {
        f->state = 2;
        f->wait_gen = g;
        yield_to_scheduler();
    }
}

static void fiber_main() {
    Fiber* f = cur;
    tl_worker.thunk(tl_worker.ctx);
    // the thread has left the kernel: it no longer takes part in collectives / barriers
    f = cur;
    Wave* w = f->wave;
    w->live &= ~(1ull << f->lane);
    if (w->arrived && w->arrived == w->live) wave_release(w);
    Block* b = f->block;
    b->live--;
    if (b->arrived && b->arrived == b->live) {
        b->arrived = 0;
        b->gen++;
    }
    f->state = 3;
    yield_to_scheduler();
    abort();  // a finished fiber is never resumed
}

static void run_block(Worker& wk, Idx bid, dim3 block, dim3 grid, std::vector<Fiber>& fibers, std::vector<Wave>& waves) {
    const uint32_t nthreads = block.x * block.y * block.z;
    const uint32_t nwaves = (nthreads + 63) / 64;
    Block blk;
    blk.live = nthreads;
    for (uint32_t w = 0; w < nwaves; ++w) {
        waves[w] = Wave();
        const uint32_t in_wave = std::min(64u, nthreads - w * 64);
        waves[w].live = in_wave == 64 ? ~0ull : ((1ull << in_wave) - 1);
    }
    for (uint32_t t = 0; t < nthreads; ++t) {
        Fiber& f = fibers[t];
        f.tid = Idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        f.bid = bid;
        f.bdim = Idx{block.x, block.y, block.z};
        f.gdim = Idx{grid.x, grid.y, grid.z};
        f.lane = (int)(t & 63);
        f.wave = &waves[t >> 6];
        f.block = &blk;
        f.state = 0;
        f.stack = wk.stacks + (size_t)t * kStackBytes;
        // initial frame: six callee-saved registers, the entry point, and a slot that keeps the ABI's stack alignment
        uint64_t* top = reinterpret_cast<uint64_t*>(f.stack + kStackBytes);
        top[-1] = 0;
        top[-2] = reinterpret_cast<uint64_t>(&fiber_main);
        for (int r = 3; r <= 8; ++r) top[-r] = 0;
        f.sp = top - 8;
    }
    // VS_EMU_ORDER=reverse runs the lanes highest first: anything two lanes exchange through memory WITHOUT a rendezvous
    // in between then sees the opposite order, so a result that survives both orders does not depend on it
    // VS_EMU_ORDER=shuffle: another pseudo-random permutation of the lanes in every scheduling pass
    static const char order = [] {
        const char* e = getenv("VS_EMU_ORDER");
        return e ? e[0] : 'f';
    }();
    const bool reverse = order == 'r';
    uint32_t done = 0, pass = 0;
    while (done < nthreads) {
        bool progress = false;
        uint32_t stride = 1, offset = 0;
        if (order == 's') {
            static const uint32_t primes[] = {7919, 104729, 15485863, 32452843, 49979687, 67867967};
            ++pass;
            stride = primes[pass % 6] % nthreads;
            auto gcd = [](uint32_t a, uint32_t b) {
                while (b) {
                    const uint32_t r = a % b;
                    a = b;
                    b = r;
                }
                return a;
            };
            while (stride == 0 || gcd(stride, nthreads) != 1) ++stride;
            offset = (pass * 2654435761u >> 7) % nthreads;
        }
        for (uint32_t tt = 0; tt < nthreads; ++tt) {
            const uint32_t t = reverse ? nthreads - 1 - tt : (uint32_t)(((uint64_t)tt * stride + offset) % nthreads);
            Fiber& f = fibers[t];
            if (f.state == 3) continue;
            if (f.state == 1 && f.wave->gen == f.wait_gen) continue;
            if (f.state == 2 && blk.gen == f.wait_gen) continue;
            f.state = 0;
            cur = &f;
            emu_switch(&wk.sched_sp, f.sp);
            progress = true;
            if (f.state == 3) done++;
        }
        if (!progress) {
            // every remaining thread waits for someone who will never arrive
            for (uint32_t t = 0; t < nthreads; ++t)
                if (fibers[t].state == 1)
                    die("deadlock: part of a wave waits at a cross-lane operation the other lanes never reach", fibers[t].wave->file,
                        fibers[t].wave->line, blk.file, blk.line);
            die("deadlock at __syncthreads()", blk.file, blk.line, nullptr, 0);
        }
    }
    cur = nullptr;
}

static void on_segv(int sig, siginfo_t* si, void* uc_) {
    ucontext_t* uc = static_cast<ucontext_t*>(uc_);
    Dl_info di{};
    void* pc = (void*)uc->uc_mcontext.gregs[REG_RIP];
    dladdr(pc, &di);
    fprintf(stderr, "emu: fault address %p, pc %p = %s+0x%lx (%s)\n", si->si_addr, pc, di.dli_fname ? di.dli_fname : "?",
            (unsigned long)((char*)pc - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
    void** sp = (void**)uc->uc_mcontext.gregs[REG_RSP];
    for (int i = 0; i < 6; ++i) {
        Dl_info d2{};
        if (dladdr(sp[i], &d2) && d2.dli_fname)
            fprintf(stderr, "emu:   stack[%d] = %p = %s+0x%lx (%s)\n", i, sp[i], d2.dli_fname, (unsigned long)((char*)sp[i] - (char*)d2.dli_fbase),
                    d2.dli_sname ? d2.dli_sname : "?");
    }
    void* frames[48];
    const int n = backtrace(frames, 48);
    fprintf(stderr, "emu: signal %d in GPU thread (%u,%u,%u) of workgroup (%u,%u,%u)\n", sig, cur ? cur->tid.x : 0, cur ? cur->tid.y : 0,
            cur ? cur->tid.z : 0, cur ? cur->bid.x : 0, cur ? cur->bid.y : 0, cur ? cur->bid.z : 0);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*thunk)(void*), void* ctx) {
    static const bool trace = [] {
        const char* e = getenv("VS_EMU_TRACE");
        if (e && *e) {
            static char altstack[1 << 16];
            stack_t ss{};
            ss.ss_sp = altstack;
            ss.ss_size = sizeof altstack;
            sigaltstack(&ss, nullptr);
            struct sigaction sa{};
            sa.sa_sigaction = on_segv;
            sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
            sigaction(SIGSEGV, &sa, nullptr);
            sigaction(SIGBUS, &sa, nullptr);
            return true;
        }
        return false;
    }();
    if (trace) fprintf(stderr, "emu: launch grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", grid.x, grid.y, grid.z, block.x, block.y, block.z, lds_bytes);
    const uint32_t nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > (uint32_t)kMaxThreadsPerBlock || lds_bytes > sizeof(smem)) {
        fprintf(stderr, "emu: unsupported launch (%u threads per workgroup, %zu B LDS)\n", nthreads, lds_bytes);
        abort();
    }
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    static const uint32_t max_threads = [] {
        const char* e = getenv("VS_EMU_THREADS");
        const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
        return e && *e ? std::max(1u, (uint32_t)atoi(e)) : hw;
    }();
    std::atomic<uint64_t> next{0};
    auto body = [&]() {
        Worker& wk = tl_worker;
        if (!wk.stacks) {
            void* m = mmap(nullptr, kStackBytes * kMaxThreadsPerBlock, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (m == MAP_FAILED) {
                perror("emu: mmap of fiber stacks");
                abort();
            }
            wk.stacks = static_cast<char*>(m);
        }
        wk.thunk = thunk;
        wk.ctx = ctx;
        std::vector<Fiber> fibers(nthreads);
        std::vector<Wave> waves((nthreads + 63) / 64);
        for (;;) {
            const uint64_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            memset(smem, 0xCD, std::min(sizeof(smem), lds_bytes + 4096));  // LDS is not zero initialised
            const Idx bid{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y))};
            run_block(wk, bid, block, grid, fibers, waves);
            // the words right after the launch's dynamic LDS segment must be untouched (an out-of-range LDS store is
            // dropped by the hardware; here it would silently land in the next bytes)
            for (size_t i = lds_bytes; i < std::min(sizeof(smem), lds_bytes + 4096); ++i)
                if (smem[i] != 0xCD) {
                    fprintf(stderr, "emu: workgroup (%u,%u,%u) wrote LDS byte %zu, past the %zu bytes of the launch\n", bid.x, bid.y, bid.z, i,
                            lds_bytes);
                    abort();
                }
        }
    };
    const uint32_t T = (uint32_t)std::min<uint64_t>(max_threads, nblocks);
    if (T <= 1) {
        body();
        return;
    }
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < T; ++t) pool.emplace_back(body);
    for (auto& th : pool) th.join();
}

}  // namespace emu
