// vs_scan.hip — K5: flat SBQ scan.  Streams every code row once per tile of queries and keeps, for each query, the k
// smallest (hamming, node id) pairs — distance_xor_optimized (AM/distance/mod.rs:266-323) over the whole corpus instead
// of over the graph's neighbor gathers.  This is the bandwidth-bound form of the SBQ candidate scan: algorithmic
// traffic = n * 8 * code_stride bytes per query TILE (the tile's query codes sit in LDS), which is what the
// "SBQ-scan achieved HBM GB/s" figure of BASELINE.json is measured on.
//
// Layout: 4 lanes x 16 B cover one 64-B sector of a code row per load instruction, 16 rows per wave per pass, PASSES
// passes in flight per wave so enough bytes are outstanding to cover HBM latency.  Each wave owns a contiguous range of
// rows (DRAM page locality) and a private sorted top-k list per query in LDS; a row only touches the list when its key
// (hamming << 32 | id) beats the list's current k-th key, which after the first few hundred rows is rare.  A second,
// tiny kernel merges the per-wave lists.  Order of the result: hamming ascending, node id ascending (exact).
#include <cstdlib>

#include "vs_device.h"

#define SCAN_QMAX 16    // queries per tile (template parameter Q: 4, 8 or 16)
#define SCAN_WAVES 4    // waves per workgroup
#define SCAN_PASSES 4   // 16-row passes in flight per wave
#define SCAN_KMAX 64

struct ScanArgs {
    const uint64_t* codes;
    uint32_t code_stride, n;
    const uint64_t* qcodes;  // [nq][code_stride]
    uint32_t nq, k;
    uint32_t rows_per_wave;  // multiple of 16 * SCAN_PASSES
    uint64_t* partial;       // [tiles][waves_total][Q][k]
    uint32_t q_tile;         // Q (for the merge kernel)
    uint32_t waves_total;
    // optional predicate (vs_scan_topk_filtered): a row is admitted for query q only when its label set overlaps the key
    // qlabels[qlabel_off[q] .. qlabel_off[q + 1]) (LabelSetView::overlaps, AM/labels/mod.rs:124-142; an empty key filters
    // nothing, AM/scan.rs:189) and, with live_only, its heap tid is not deleted (AM/scan.rs:231-234)
    const uint32_t* label_off = nullptr;
    const int16_t* label_val = nullptr;
    const int16_t* qlabels = nullptr;
    const uint32_t* qlabel_off = nullptr;
    const uint64_t* tids = nullptr;
};

// (row and query are wave-uniform: scalar loads)
__device__ __forceinline__ bool scan_row_admitted(const ScanArgs& a, uint32_t row, uint32_t q) {
    if (a.tids && (a.tids[row] & 0xFFFFull) == 0) return false;
    if (!a.qlabel_off) return true;
    uint32_t i = a.qlabel_off[q];
    const uint32_t ie = a.qlabel_off[q + 1];
    if (i == ie) return true;
    uint32_t j = a.label_off[row];
    const uint32_t je = a.label_off[row + 1];
    while (i < ie && j < je) {
        const int16_t x = a.qlabels[i], y = a.label_val[j];
        if (x == y) return true;
        if (x < y) ++i;
        else ++j;
    }
    return false;
}

// insert key into the wave's sorted list (ascending, k entries, list[i] in LDS); wave-uniform key
__device__ __forceinline__ void list_insert(uint64_t* list, uint32_t k, uint64_t key, int lane) {
    uint64_t b = ~0ull;
    if ((uint32_t)lane < k) b = list[lane];
    const uint32_t pos = (uint32_t)__popcll(__ballot((uint32_t)lane < k && b < key));
    if (pos >= k) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if ((uint32_t)lane >= pos && (uint32_t)lane + 1 < k) list[lane + 1] = b;
    if ((uint32_t)lane == pos) list[pos] = key;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NCH, int SCAN_Q>
__global__ __launch_bounds__(SCAN_WAVES* WAVE) void k_scan_topk(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* qc = reinterpret_cast<uint64_t*>(smem);                       // [SCAN_Q][code_stride]
    uint64_t* lists = qc + (size_t)SCAN_Q * a.code_stride;                   // [SCAN_WAVES][SCAN_Q][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.y;
    const uint32_t q0 = tile * SCAN_Q;
    const uint32_t nqt = min((uint32_t)SCAN_Q, a.nq - q0);
    for (uint32_t i = tid; i < SCAN_Q * a.code_stride; i += blockDim.x) {
        const uint32_t qi = i / a.code_stride;
        qc[i] = qi < nqt ? a.qcodes[(size_t)(q0 + qi) * a.code_stride + (i - qi * a.code_stride)] : 0ull;
    }
    uint64_t* mylists = lists + (size_t)wave * SCAN_Q * a.k;
    for (uint32_t i = lane; i < SCAN_Q * a.k; i += WAVE) mylists[i] = ~0ull;
    __syncthreads();

    const uint32_t gw = blockIdx.x * SCAN_WAVES + wave;  // global wave index inside the tile
    const uint32_t r_begin = (uint32_t)min<uint64_t>((uint64_t)gw * a.rows_per_wave, a.n);
    const uint32_t r_end = (uint32_t)min<uint64_t>((uint64_t)r_begin + a.rows_per_wave, a.n);
    const int l4 = lane & 3, grp = lane >> 2;
    // thresholds: the hamming of the current k-th entry of each query's list (wave-uniform, kept in SGPRs).  A wave
    // sees its rows in increasing id order, so a row whose hamming EQUALS the threshold has a larger id than every
    // listed entry of that hamming and cannot enter: "hamming < threshold" is the exact admission test.
    // (queries past the end of the last tile keep threshold 0 and never admit anything)
    uint32_t th[SCAN_Q];
#pragma unroll
    for (int qi = 0; qi < SCAN_Q; ++qi) th[qi] = (uint32_t)qi < nqt ? 0xFFFFFFFFu : 0u;

    for (uint32_t r0 = r_begin; r0 < r_end; r0 += 16 * SCAN_PASSES) {
        ulonglong2 rows[SCAN_PASSES][NCH];
        uint64_t okm[SCAN_PASSES];  // lanes that report a row of this pass (one per 4-lane group)
#pragma unroll
        for (int p = 0; p < SCAN_PASSES; ++p) {
            const uint32_t row = r0 + (uint32_t)p * 16 + grp;
            const bool ok = row < r_end;
            okm[p] = __ballot(ok && l4 == 0);
            const uint64_t* rp = a.codes + (size_t)(ok ? row : r_begin) * a.code_stride;
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
                rows[p][t] = (ok && w < a.code_stride) ? *reinterpret_cast<const ulonglong2*>(rp + w) : make_ulonglong2(0, 0);
            }
        }
#pragma unroll
        for (int qi = 0; qi < SCAN_Q; ++qi) {
            ulonglong2 qv[NCH];
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
                qv[t] = w < a.code_stride ? *reinterpret_cast<const ulonglong2*>(qc + (size_t)qi * a.code_stride + w)
                                          : make_ulonglong2(0, 0);
            }
            uint32_t ham[SCAN_PASSES];
            uint64_t hit[SCAN_PASSES];
            uint64_t any = 0;
#pragma unroll
            for (int p = 0; p < SCAN_PASSES; ++p) {
                uint32_t acc = 0;
#pragma unroll
                for (int t = 0; t < NCH; ++t)
                    acc += (uint32_t)__popcll(rows[p][t].x ^ qv[t].x) + (uint32_t)__popcll(rows[p][t].y ^ qv[t].y);
                ham[p] = quad_sum(acc);
                hit[p] = __ballot(ham[p] < th[qi]) & okm[p];
                any |= hit[p];
            }
            if (any) {  // rare once the list has warmed up
#pragma unroll
                for (int p = 0; p < SCAN_PASSES; ++p) {
                    uint64_t h = hit[p];
                    while (h) {
                        const int src = __builtin_ctzll(h);
                        h &= h - 1;
                        const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)ham[p], src);
                        const uint32_t rid = r0 + (uint32_t)p * 16 + (uint32_t)(src >> 2);
                        if (hk < th[qi] && scan_row_admitted(a, rid, q0 + (uint32_t)qi)) {
                            const uint64_t kk = ((uint64_t)hk << 32) | rid;
                            uint64_t* lst = mylists + (size_t)qi * a.k;
                            list_insert(lst, a.k, kk, lane);
                            th[qi] = rfl((uint32_t)(lst[a.k - 1] >> 32));
                        }
                    }
                }
            }
        }
    }
    // publish this wave's lists
    uint64_t* out = a.partial + ((size_t)tile * a.waves_total + gw) * SCAN_Q * a.k;
    for (uint32_t i = lane; i < SCAN_Q * a.k; i += WAVE) out[i] = mylists[i];
}

// merge: one wave per query; repeatedly extract the smallest key greater than the last one taken (keys are unique)
__global__ __launch_bounds__(WAVE) void k_scan_merge(const uint64_t* __restrict__ partial, uint32_t waves_total, uint32_t nq,
                                                     uint32_t k, uint32_t Q, uint32_t* __restrict__ out_ids,
                                                     uint32_t* __restrict__ out_ham) {
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const int lane = threadIdx.x;
    const uint32_t tile = q / Q, qi = q - tile * Q;
    const uint64_t* base = partial + (size_t)tile * waves_total * Q * k;
    const uint32_t total = waves_total * k;
    uint64_t lastkey = 0;
    bool first = true;
    for (uint32_t j = 0; j < k; ++j) {
        uint64_t best = ~0ull;
        for (uint32_t i = lane; i < total; i += WAVE) {
            const uint32_t w = i / k, e = i - w * k;
            const uint64_t v = base[((size_t)w * Q + qi) * k + e];
            if ((first || v > lastkey) && v < best) best = v;
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, sh, WAVE);
            const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), sh, WAVE);
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            best = o < best ? o : best;
        }
        if (lane == 0) {
            out_ids[(size_t)q * k + j] = best == ~0ull ? VS_INVALID_NODE : (uint32_t)best;
            out_ham[(size_t)q * k + j] = best == ~0ull ? 0xFFFFFFFFu : (uint32_t)(best >> 32);
        }
        lastkey = best;
        first = false;
        if (best == ~0ull) {
            for (uint32_t jj = j + 1; jj < k; ++jj)
                if (lane == 0) {
                    out_ids[(size_t)q * k + jj] = VS_INVALID_NODE;
                    out_ham[(size_t)q * k + jj] = 0xFFFFFFFFu;
                }
            break;
        }
    }
}

template <int NCH, int Q>
static int launch_scan_tq(vs_index* idx, const ScanArgs& a, dim3 grid, size_t lds) {
    static DeviceOnce attr_set;
    const int attr_dev = idx->ctx->device;
    if (attr_set.pending(attr_dev)) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_topk<NCH, Q>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set.done(attr_dev);
    }
    hipLaunchKernelGGL((k_scan_topk<NCH, Q>), grid, dim3(SCAN_WAVES * WAVE), lds, idx->ctx->stream, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
template <int NCH>
static int launch_scan_t(vs_index* idx, const ScanArgs& a, dim3 grid, size_t lds) {
    switch (a.q_tile) {
        case 4: return launch_scan_tq<NCH, 4>(idx, a, grid, lds);
        case 16: return launch_scan_tq<NCH, 16>(idx, a, grid, lds);
        default: return launch_scan_tq<NCH, 8>(idx, a, grid, lds);
    }
}

// d_qcodes: device [nq][code_stride]; d_out_*: device [nq][k]
int launch_scan_topk(vs_index* idx, const uint64_t* d_qcodes, uint32_t nq, uint32_t k, uint32_t* d_out_ids, uint32_t* d_out_ham,
                     const int16_t* d_qlabels, const uint32_t* d_qlabel_off, bool live_only) {
    if (nq == 0) return VS_OK;
    VS_REQUIRE(k >= 1 && k <= SCAN_KMAX, "vs_scan_topk: k must be in [1, %d]", SCAN_KMAX);
    const uint32_t nch = (idx->code_stride + 7) / 8;
    VS_REQUIRE(nch >= 1 && nch <= 6, "vs_scan_topk: codes of %u words are not supported by the flat scan (max 48)", idx->code_stride);
    vs_ctx* c = idx->ctx;
    SearchWorkspace& w = idx->ws;
    const uint32_t n = idx->d.n;
    // queries per tile: the codes are streamed once per tile, so wide tiles amortise HBM traffic until the xor/popcount
    // work per byte makes the kernel issue bound (about 8 queries at 24-word codes)
    uint32_t Q = nq <= 4 ? 4u : 8u;
    if (const char* e = vs_opt_get("VS_SCAN_Q")) {
        const uint32_t v = (uint32_t)strtoul(e, nullptr, 10);
        if (v == 4 || v == 8 || v == 16) Q = v;
    }
    // enough waves to fill the chip (256 CUs x 8 waves) but at least one 64-row step each
    const uint32_t step = 16 * SCAN_PASSES;
    uint32_t waves = (uint32_t)c->prop.multiProcessorCount * 8u;
    uint32_t rows_per_wave = (uint32_t)(((uint64_t)n + waves - 1) / waves);
    rows_per_wave = std::max(step, round_up_u32(rows_per_wave, step));
    waves = (uint32_t)(((uint64_t)n + rows_per_wave - 1) / rows_per_wave);
    const uint32_t blocks = (waves + SCAN_WAVES - 1) / SCAN_WAVES;
    const uint32_t waves_total = blocks * SCAN_WAVES;
    const uint32_t tiles = (nq + Q - 1) / Q;
    VS_TRY(devbuf_reserve(c, w.misc, (size_t)tiles * waves_total * Q * k * 8));
    ScanArgs a;
    a.codes = idx->codes;
    a.code_stride = idx->code_stride;
    a.n = n;
    a.qcodes = d_qcodes;
    a.nq = nq;
    a.k = k;
    a.rows_per_wave = rows_per_wave;
    a.partial = (uint64_t*)w.misc.p;
    a.waves_total = waves_total;
    a.q_tile = Q;
    if (d_qlabel_off) {
        VS_REQUIRE(idx->label_off && idx->label_val, "vs_scan_topk_filtered: the index has no label sets");
        a.label_off = idx->label_off;
        a.label_val = idx->label_val;
        a.qlabels = d_qlabels;
        a.qlabel_off = d_qlabel_off;
    }
    a.tids = live_only ? idx->tids : nullptr;
    const size_t lds = ((size_t)Q * idx->code_stride + (size_t)SCAN_WAVES * Q * k) * 8;
    const dim3 grid(blocks, tiles);
    hipEvent_t ev = prof_begin(c);
    int rc;
    switch (nch) {
        case 1: rc = launch_scan_t<1>(idx, a, grid, lds); break;
        case 2: rc = launch_scan_t<2>(idx, a, grid, lds); break;
        case 3: rc = launch_scan_t<3>(idx, a, grid, lds); break;
        case 4: rc = launch_scan_t<4>(idx, a, grid, lds); break;
        default: rc = launch_scan_t<6>(idx, a, grid, lds); break;
    }
    prof_end(c, PK_SCAN, ev);
    VS_TRY(rc);
    hipLaunchKernelGGL(k_scan_merge, dim3(nq), dim3(WAVE), 0, c->stream, (const uint64_t*)w.misc.p, waves_total, nq, k, Q, d_out_ids,
                       d_out_ham);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
