// TEST INFRASTRUCTURE ONLY — a stand-in for librccl in the CPU test tier (VS_RCCL_LIB=tests/emu/libfakerccl.so): the entry
// points libvsgpu's vs_comm_* resolve with dlsym, implemented between PROCESSES over one POSIX shared-memory segment per
// communicator.  "Device" pointers are host pointers here (the wave64 interpreter's device memory is host memory) and every
// call is synchronous, so a stream argument is ignored.  Semantics kept: ncclGetUniqueId by one rank, ncclCommInitRank as a
// rendezvous of `nranks` callers, ncclBroadcast / ncclAllGather with element counts and NCCL's datatype codes, every rank
// issuing the same sequence of collectives.  Never loaded by the product unless VS_RCCL_LIB says so.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr size_t kChunk = 1u << 20;
struct Seg {
    std::atomic<uint32_t> attached, arrived, gen, broken;
    uint32_t pad[12];
    unsigned char data[kChunk];
};
struct Comm {
    Seg* seg;
    int rank, world;
};
const size_t kTypeSize[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};  // ncclInt8, Uint8, Int32, Uint32, Int64, Uint64, Float16, Float32, Float64, Bfloat16

bool wait_for(const std::atomic<uint32_t>& v, uint32_t unlike, const std::atomic<uint32_t>& broken) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; v.load(std::memory_order_acquire) == unlike; ++spin) {
        if (broken.load()) return false;
        if (spin < 64) sched_yield();
        else usleep(100);
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
    }
    return true;
}
bool barrier(Comm* c) {
    Seg* s = c->seg;
    const uint32_t g = s->gen.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->gen.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    if (!wait_for(s->gen, g, s->broken)) {
        s->broken.store(1);
        return false;
    }
    return true;
}
int bcast(Comm* c, const void* send, void* recv, size_t bytes, int root) {
    for (size_t o = 0; o < bytes; o += kChunk) {
        const size_t nb = bytes - o < kChunk ? bytes - o : kChunk;
        if (c->rank == root) memcpy(c->seg->data, (const char*)send + o, nb);
        if (!barrier(c)) return 2;
        if (c->rank != root) memcpy((char*)recv + o, c->seg->data, nb);
        else if (recv != send) memmove((char*)recv + o, (const char*)send + o, nb);
        if (!barrier(c)) return 2;
    }
    return 0;
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };
typedef Comm* ncclComm_t;

const char* ncclGetErrorString(int e) { return e == 0 ? "no error" : (e == 2 ? "fake rccl: a peer did not arrive (system error)" : "fake rccl: invalid argument"); }

int ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<uint32_t> ctr{0};
    memset(id, 0, sizeof(*id));
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof(id->internal), "/vsfakerccl_%d_%u_%llx", (int)getpid(), ctr.fetch_add(1), (unsigned long long)now);
    return 0;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;
    id.internal[127] = 0;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Seg)) != 0) {
        close(fd);
        return 2;
    }
    void* p = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Comm* c = new Comm{(Seg*)p, rank, nranks};
    c->seg->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->seg->attached.load() < (uint32_t)nranks) {
        usleep(200);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            shm_unlink(id.internal);
            munmap(p, sizeof(Seg));
            delete c;
            return 2;
        }
    }
    if (!barrier(c)) return 2;
    if (rank == 0) shm_unlink(id.internal);  // (every rank has it mapped)
    *out = c;
    return 0;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    munmap(c->seg, sizeof(Seg));
    delete c;
    return 0;
}

int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, ncclComm_t c, void*) {
    if (!c || dtype < 0 || dtype > 9 || root < 0 || root >= c->world) return 4;
    return bcast(c, send, recv, count * kTypeSize[dtype], root);
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, void*) {
    if (!c || dtype < 0 || dtype > 9) return 4;
    const size_t bytes = count * kTypeSize[dtype];
    for (int r = 0; r < c->world; ++r) {
        const int e = bcast(c, send, (char*)recv + (size_t)r * bytes, bytes, r);
        if (e) return e;
    }
    return 0;
}
}
