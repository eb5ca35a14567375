// vs_shm_lat — measurement harness (not part of libvsgpu's ABI): N BACKEND PROCESSES, each a plain C client of the shared-memory
// server (vs_shm_client_*: no HIP, no device context — what a PostgreSQL backend is to the GPU broker), run
// `LIMIT k` scans in a closed loop and record the latency of every scan.  What the reference publishes for this path are p95 latencies
// and throughput ratios of exactly such scans (/root/reference/README.md:17-21; AM/scan.rs:369-436 is the call sequence a scan makes):
// bench.py's `latency` extra starts the server, runs this program at 1 / 8 / 64 / 512 backends and puts its JSON line next to the
// oracle's single-thread latency on the same queries.
//
//   vs_shm_lat <segment> <queries.f32> <dim> <n_queries> <backends> <scans_per_backend> <search_list_size> <rescore> <k>
//   vs_shm_lat <segment> <queries.f32> <dim> <n_queries> <backends> 1 <search_list_size> <rescore> <chunk> <rows_per_scan> [ids_out.u32]
//
// The second form STREAMS (round 6; the cursor measurements used Python backends until then, whose interpreter time per fetch was
// of the order of the fetch itself): backend b runs ONE scan of query b and pulls rows_per_scan rows in chunks — the first chunk out of
// a shared OP_SEARCH launch, every later one an OP_FETCH continuation, then OP_CLOSE — three times over, the third pass timed (the first
// two pay the serving process's allocations).  wall_ms is the timed pass of all backends; ids_out receives [backends][rows_per_scan]
// node ids (0xFFFFFFFF where a scan ended early).
//
// The parent forks the backends, releases them together once every one has mapped the segment and run two warm-up scans, and
// aggregates: p50 / p95 / p99 / mean / max latency in microseconds over all timed scans, whole-run throughput, a checksum of the
// returned node ids (the same queries give the same rows at any concurrency).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vsgpu.h"

namespace {

struct Shared {
    std::atomic<uint32_t> ready, go, failed;
    std::atomic<uint64_t> checksum;
    char err[256];
};

double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

double pct(const std::vector<float>& sorted, double p) {
    if (sorted.empty()) return 0.0;
    const double pos = p * (double)(sorted.size() - 1);
    const size_t i = (size_t)pos;
    const double f = pos - (double)i;
    return i + 1 < sorted.size() ? sorted[i] * (1.0 - f) + sorted[i + 1] * f : sorted[i];
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 10) {
        fprintf(stderr, "usage: %s <segment> <queries.f32> <dim> <n_queries> <backends> <scans_per_backend> <L> <rescore> <k>\n", argv[0]);
        return 2;
    }
    const char* seg = argv[1];
    const char* qpath = argv[2];
    const uint32_t dim = (uint32_t)atoi(argv[3]), nqf = (uint32_t)atoi(argv[4]), nb = (uint32_t)atoi(argv[5]), reps = (uint32_t)atoi(argv[6]);
    const uint32_t L = (uint32_t)atoi(argv[7]), S = (uint32_t)atoi(argv[8]), k = (uint32_t)atoi(argv[9]);
    const uint32_t stream_rows = argc > 10 ? (uint32_t)atoi(argv[10]) : 0u;
    const char* rows_out = argc > 11 ? argv[11] : nullptr;
    if (!dim || !nqf || !nb || !reps || !k || nb > 4096) {
        fprintf(stderr, "vs_shm_lat: bad arguments\n");
        return 2;
    }
    // the queries: one read-only mapping shared by every backend
    const int fd = open(qpath, O_RDONLY);
    if (fd < 0) {
        perror("vs_shm_lat: queries");
        return 2;
    }
    const size_t qbytes = (size_t)nqf * dim * 4;
    const float* queries = (const float*)mmap(nullptr, qbytes, PROT_READ, MAP_SHARED, fd, 0);
    if (queries == MAP_FAILED) {
        perror("vs_shm_lat: mmap queries");
        return 2;
    }
    const size_t lat_bytes = (size_t)nb * reps * sizeof(float);
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    float* lat = (float*)mmap(nullptr, lat_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED || lat == MAP_FAILED) {
        perror("vs_shm_lat: mmap");
        return 2;
    }
    uint32_t* srows = nullptr;  // (stream mode) [backends][rows_per_scan] node ids
    if (stream_rows) {
        srows = (uint32_t*)mmap(nullptr, (size_t)nb * stream_rows * 4, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (srows == MAP_FAILED) {
            perror("vs_shm_lat: mmap rows");
            return 2;
        }
    }
    new (sh) Shared();
    std::vector<pid_t> kids;
    for (uint32_t b = 0; b < nb; ++b) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("vs_shm_lat: fork");
            sh->failed.fetch_add(1);
            break;
        }
        if (pid == 0) {  // ---- a backend
            vs_shm_client* c = nullptr;
            std::vector<uint32_t> ids(k);
            std::vector<uint64_t> tids(k);
            std::vector<float> dist(k);
            int rc = vs_shm_client_open(seg, &c);
            for (int w = 0; w < 2 && rc == VS_OK && !stream_rows; ++w)  // warm-up: the serving process's lazy allocations are not a backend's latency
                rc = vs_shm_client_search(c, queries + (size_t)((b * 7919u + w) % nqf) * dim, nullptr, 0, 0, L, S, k, ids.data(), tids.data(), dist.data());
            if (rc != VS_OK) {
                if (!sh->failed.fetch_add(1)) snprintf(sh->err, sizeof(sh->err), "backend %u: %s", b, vs_last_error());
                sh->ready.fetch_add(1);
                _exit(1);
            }
            if (stream_rows) {  // ---- one streamed scan per backend, three passes, the third timed
                const float* q = queries + (size_t)(b % nqf) * dim;
                uint64_t sum = 0;
                for (uint32_t pass = 0; pass < 3 && rc == VS_OK; ++pass) {
                    const uint64_t sid = 1000ull * (pass + 1) + b;
                    if (pass == 2) {
                        sh->ready.fetch_add(1);
                        while (!sh->go.load(std::memory_order_acquire)) usleep(50);
                    }
                    uint32_t have = 0;
                    rc = vs_shm_client_search(c, q, nullptr, 0, 0, L, S, k, ids.data(), tids.data(), dist.data());
                    if (rc != VS_OK) break;
                    for (uint32_t j = 0; j < k && have < stream_rows; ++j) srows[(size_t)b * stream_rows + have++] = ids[j];
                    while (have < stream_rows) {
                        uint32_t n = 0;
                        rc = vs_shm_client_fetch(c, sid, q, nullptr, 0, 0, L, S, 0, have, k, ids.data(), tids.data(), dist.data(), &n);
                        if (rc != VS_OK) break;
                        for (uint32_t j = 0; j < n && have < stream_rows; ++j) srows[(size_t)b * stream_rows + have++] = ids[j];
                        if (n < k) break;
                    }
                    if (rc == VS_OK) rc = vs_shm_client_end_scan(c, sid);
                    for (uint32_t j = have; j < stream_rows; ++j) srows[(size_t)b * stream_rows + j] = 0xFFFFFFFFu;
                    if (pass == 2)
                        for (uint32_t j = 0; j < have; ++j) sum += (uint64_t)srows[(size_t)b * stream_rows + j] * (uint64_t)(j + 1);
                }
                if (rc != VS_OK) {
                    if (!sh->failed.fetch_add(1)) snprintf(sh->err, sizeof(sh->err), "backend %u (streamed scan): %s", b, vs_last_error());
                    sh->ready.fetch_add(1);
                    _exit(1);
                }
                sh->checksum.fetch_add(sum);
                vs_shm_client_close(c);
                _exit(0);
            }
            sh->ready.fetch_add(1);
            while (!sh->go.load(std::memory_order_acquire)) usleep(200);
            uint64_t sum = 0;
            for (uint32_t r = 0; r < reps; ++r) {
                const float* q = queries + (size_t)((b * reps + r) % nqf) * dim;
                const double t0 = now_us();
                rc = vs_shm_client_search(c, q, nullptr, 0, 0, L, S, k, ids.data(), tids.data(), dist.data());
                lat[(size_t)b * reps + r] = (float)(now_us() - t0);
                if (rc != VS_OK) {
                    if (!sh->failed.fetch_add(1)) snprintf(sh->err, sizeof(sh->err), "backend %u scan %u: %s", b, r, vs_last_error());
                    _exit(1);
                }
                for (uint32_t j = 0; j < k; ++j) sum += (uint64_t)ids[j] * (uint64_t)(j + 1);
            }
            sh->checksum.fetch_add(sum);
            vs_shm_client_close(c);
            _exit(0);
        }
        kids.push_back(pid);
    }
    const double tw0 = now_us();
    while (sh->ready.load() < kids.size() && now_us() - tw0 < 300e6) usleep(1000);
    const double t0 = now_us();
    sh->go.store(1, std::memory_order_release);
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
    }
    const double wall_us = now_us() - t0;
    if (sh->failed.load() || kids.size() != nb) {
        printf("{\"error\": \"%s\"}\n", sh->err[0] ? sh->err : "a backend could not be started");
        return 1;
    }
    if (stream_rows) {
        if (rows_out) {
            FILE* f = fopen(rows_out, "wb");
            if (!f || fwrite(srows, 4, (size_t)nb * stream_rows, f) != (size_t)nb * stream_rows) {
                printf("{\"error\": \"cannot write %s\"}\n", rows_out);
                return 1;
            }
            fclose(f);
        }
        printf("{\"backends\": %u, \"rows_per_scan\": %u, \"chunk\": %u, \"search_list_size\": %u, \"rescore\": %u, \"wall_ms\": %.3f, \"ids_checksum\": %llu}\n",
               nb, stream_rows, k, L, S, wall_us * 1e-3, (unsigned long long)sh->checksum.load());
        return 0;
    }
    std::vector<float> all(lat, lat + (size_t)nb * reps);
    std::sort(all.begin(), all.end());
    double mean = 0;
    for (float v : all) mean += v;
    mean /= (double)all.size();
    printf("{\"backends\": %u, \"scans_per_backend\": %u, \"search_list_size\": %u, \"rescore\": %u, \"k\": %u, \"p50_us\": %.1f, \"p95_us\": %.1f, "
           "\"p99_us\": %.1f, \"mean_us\": %.1f, \"max_us\": %.1f, \"wall_ms\": %.2f, \"scans_per_s\": %.1f, \"ids_checksum\": %llu}\n",
           nb, reps, L, S, k, pct(all, 0.50), pct(all, 0.95), pct(all, 0.99), mean, (double)all.back(), wall_us * 1e-3,
           (double)nb * reps / (wall_us * 1e-6), (unsigned long long)sh->checksum.load());
    return 0;
}
