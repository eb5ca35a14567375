// vs_extra.hip — build-side kernels that manufacture device-resident indexes (SURVEY.md §8f "next" rows) and the
// synthetic corpus generator.  None of this is on the reference's *search* path; it exists so that corpora far
// larger than host RAM (50M x 768 f32 = 153.6 GB) can be generated, trained, quantised and indexed in HBM.
#include <algorithm>
#include <cmath>

#include "vs_internal.h"

#define WAVE 64

// ===============================================================================================================
// Synthetic corpus: clustered mixture with low intrinsic dimensionality, embedded by a fixed random projection.
// All arithmetic is integer (64-bit) until the final scaling, which uses IEEE f64 sqrt/div: the stream is
// bit-reproducible on the CPU (pgvectorscale_amd/datagen.py holds the numpy twin).
//   h(seed, a, b)     = splitmix64-style mix
//   G(h)              = (sum of the 8 bytes of h) - 1020          ~ N(0, 209^2), integer in [-1020, 1020]
//   cluster(r)        = h(seed, r, 0) mod n_clusters
//   z[r][j]           = 100 * G(h(seed^K1, cluster, j)) + intra_pct * G(h(seed, r, 1 + j))
//   x[r][i]           = sum_j z[r][j] * G(h(seed^K2, j, i)) + noise_pct * isqrt(latent) * 209 * G(h(seed^K3, r, i))
//   xs                = x >> 10 (arithmetic)        ss = sum_i xs^2 (exact int64)
//   out[r][i]         = normalize ? (float)((double)xs / sqrt((double)ss)) : (float)((double)xs * 2^-14)
// ===============================================================================================================
#define DG_K1 0x9E3779B97F4A7C15ull
#define DG_K2 0xC2B2AE3D27D4EB4Full
#define DG_K3 0x165667B19E3779F9ull

__host__ __device__ static inline uint64_t dg_hash(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t x = seed + a * 0x9E3779B97F4A7C15ull + b * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ static inline int32_t dg_gauss(uint64_t h) {
    // byte sum via SWAR
    uint64_t s = (h & 0x00FF00FF00FF00FFull) + ((h >> 8) & 0x00FF00FF00FF00FFull);
    s = (s & 0x0000FFFF0000FFFFull) + ((s >> 16) & 0x0000FFFF0000FFFFull);
    s = (s & 0xFFFFFFFFull) + (s >> 32);
    return (int32_t)s - 1020;
}
static uint32_t isqrt_u32(uint32_t v) {
    uint32_t r = 0;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

__global__ void k_dg_tables(uint64_t seed, uint32_t dim, uint32_t latent, uint32_t n_clusters,
                            int16_t* __restrict__ proj /*[latent][dim]*/, int16_t* __restrict__ centers /*[nc][latent]*/) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < latent * dim) {
        uint32_t j = t / dim, i = t - j * dim;
        proj[t] = (int16_t)dg_gauss(dg_hash(seed ^ DG_K2, j, i));
    }
    if (t < n_clusters * latent) {
        uint32_t c = t / latent, j = t - c * latent;
        centers[t] = (int16_t)dg_gauss(dg_hash(seed ^ DG_K1, c, j));
    }
}

// one workgroup (256 threads) per row
__global__ __launch_bounds__(256) void k_dg_fill(uint64_t seed, uint32_t dim, uint32_t latent, uint32_t n_clusters,
                                                 uint32_t intra_pct, int64_t noise_mult, uint32_t normalize,
                                                 const int16_t* __restrict__ proj, const int16_t* __restrict__ centers,
                                                 uint64_t first_row, uint64_t rows, float* __restrict__ out,
                                                 uint32_t out_stride) {
    __shared__ int32_t z[256];
    __shared__ long long red[256];
    for (uint64_t rr = blockIdx.x; rr < rows; rr += gridDim.x) {
        const uint64_t r = first_row + rr;
        const uint32_t c = (uint32_t)(dg_hash(seed, r, 0) % n_clusters);
        if (threadIdx.x < latent)
            z[threadIdx.x] = 100 * (int32_t)centers[c * latent + threadIdx.x] +
                             (int32_t)intra_pct * dg_gauss(dg_hash(seed, r, 1 + threadIdx.x));
        __syncthreads();
        long long ss = 0;
        // each thread owns dims i = tid, tid+256, ... ; keep xs in registers (dim <= 16 * 256)
        long long xs_loc[16];
        int cnt = 0;
        for (uint32_t i = threadIdx.x; i < dim; i += 256, ++cnt) {
            long long acc = 0;
            for (uint32_t j = 0; j < latent; ++j) acc += (long long)z[j] * (long long)proj[j * dim + i];
            acc += noise_mult * (long long)dg_gauss(dg_hash(seed ^ DG_K3, r, i));
            long long xs = acc >> 10;
            xs_loc[cnt] = xs;
            ss += xs * xs;
        }
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        const long long tot = red[0];
        const double denom = sqrt((double)tot);
        cnt = 0;
        for (uint32_t i = threadIdx.x; i < dim; i += 256, ++cnt) {
            double v = normalize ? ((double)xs_loc[cnt] / denom) : ((double)xs_loc[cnt] * (1.0 / 16384.0));
            out[rr * out_stride + i] = (float)v;
        }
        __syncthreads();
    }
}

extern "C" int vs_datagen_fill(vs_ctx* c, const vs_datagen_params* p, uint64_t first_row, uint64_t rows, float* d_out) {
    VS_REQUIRE(c && p && (rows == 0 || d_out), "vs_datagen_fill: bad args");
    VS_REQUIRE(p->dim >= 1 && p->dim <= 4096, "vs_datagen_fill: dim %u outside [1,4096]", p->dim);
    VS_REQUIRE(p->latent_dim >= 1 && p->latent_dim <= 128, "vs_datagen_fill: latent_dim %u outside [1,128]", p->latent_dim);
    VS_REQUIRE(p->n_clusters >= 1 && p->intra_pct <= 100 && p->noise_pct <= 100, "vs_datagen_fill: bad mixture params");
    if (rows == 0) return VS_OK;
    VS_HIP(hipSetDevice(c->device));
    int16_t *proj = nullptr, *centers = nullptr;
    VS_HIP(hipMalloc(&proj, (size_t)p->latent_dim * p->dim * 2));
    {
        hipError_t ea = hipMalloc(&centers, (size_t)p->n_clusters * p->latent_dim * 2);
        if (ea != hipSuccess) {
            (void)hipFree(proj);
            VS_HIP(ea);
        }
    }
    uint32_t tmax = std::max(p->latent_dim * p->dim, p->n_clusters * p->latent_dim);
    hipLaunchKernelGGL(k_dg_tables, dim3((tmax + 255) / 256), dim3(256), 0, c->stream, p->seed, p->dim, p->latent_dim,
                       p->n_clusters, proj, centers);
    int64_t noise_mult = (int64_t)p->noise_pct * isqrt_u32(p->latent_dim) * 209;
    uint32_t grid = (uint32_t)std::min<uint64_t>(rows, 1u << 20);
    hipLaunchKernelGGL(k_dg_fill, dim3(grid), dim3(256), 0, c->stream, p->seed, p->dim, p->latent_dim, p->n_clusters,
                       p->intra_pct, noise_mult, p->normalize, proj, centers, first_row, rows, d_out, p->dim);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(proj);
    (void)hipFree(centers);
    VS_HIP(e);
    return VS_OK;
}

// ===============================================================================================================
// SBQ training: SbqQuantizer::add_sample (AM/sbq/quantize.rs:115-148) over all rows in heap order.  The Welford
// recurrence is sequential in f32 per dimension, so parallelism is across dimensions only: lane = dimension,
// 64 dims per wave, rows prefetched 32 at a time (double-buffered registers) to keep the HBM latency off the
// dependent chain.  Bit-exact with the reference by construction.
// ===============================================================================================================
__global__ __launch_bounds__(WAVE) void k_sbq_train(const float* __restrict__ vecs, uint32_t vec_stride, uint32_t n,
                                                    uint32_t dim_index, uint32_t bits, const float* __restrict__ rnorm,
                                                    float* __restrict__ mean, float* __restrict__ m2) {
    const uint32_t d = blockIdx.x * WAVE + threadIdx.x;
    const bool act = d < dim_index;
    const uint32_t dd = act ? d : 0;
    float mu = 0.0f, s2 = 0.0f;
    constexpr int T = 32;
    float cur[T], nxt[T];
    auto load_tile = [&](uint32_t r0, float* buf) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t r = r0 + t;
            float v = 0.0f;
            if (r < n) {
                v = vecs[(size_t)r * vec_stride + dd];
                if (rnorm) {
                    float s = rnorm[r];
                    if (s != 0.0f) v = v / s;
                }
            }
            buf[t] = v;
        }
    };
    load_tile(0, cur);
    for (uint32_t r0 = 0; r0 < n; r0 += T) {
        load_tile(r0 + T, nxt);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t r = r0 + t;
            if (r < n) {
                float c = (float)(uint64_t)(r + 1);  // count as f32
                float s = cur[t];
                float delta = s - mu;
                mu = mu + (s - mu) / c;
                if (bits > 1) {
                    float delta2 = s - mu;
                    float p = delta * delta2;
                    s2 = s2 + p;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) cur[t] = nxt[t];
    }
    if (act) {
        mean[d] = mu;
        m2[d] = s2;
    }
}

// row divisor for the index slice when dim_index != dim_full (same rule as k_row_norms in vs_kernels.hip)
__global__ __launch_bounds__(WAVE) void k_slice_norms(const float* __restrict__ vecs, uint32_t vec_stride, uint32_t dim,
                                                      uint32_t n, float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    for (uint32_t row0 = blockIdx.x * 64u; row0 < n; row0 += gridDim.x * 64u) {
        float norm = 0.0f;
        for (uint32_t d0 = 0; d0 < dim; d0 += 64) {
            for (int r = 0; r < 64; ++r) {
                uint32_t row = row0 + r, d = d0 + lane;
                tile[r][lane] = (row < n && d < dim) ? vecs[(size_t)row * vec_stride + d] : 0.0f;
            }
            __syncthreads();
            uint32_t lim = min(64u, dim - d0);
            for (uint32_t c = 0; c < lim; ++c) {
                float v = tile[lane][c];
                float p = v * v;
                norm = norm + p;
            }
            __syncthreads();
        }
        uint32_t row = row0 + lane;
        if (row < n) {
            const float eps = 1.1920929e-07f;
            float adj = eps * (float)dim;
            float s = 0.0f;
            if (!(norm < eps) && !(norm >= 1.0f - adj && norm <= 1.0f + adj)) s = sqrtf(norm);
            out[row] = s;
        }
    }
}

int launch_slice_norms(vs_index* ix, float* d_out) {
    if (ix->d.n == 0) return VS_OK;
    const uint32_t blocks = std::min<uint32_t>((ix->d.n + 63) / 64, 8192);
    hipLaunchKernelGGL(k_slice_norms, dim3(blocks), dim3(WAVE), 0, ix->ctx->stream, ix->vecs, ix->vec_stride, ix->d.dim_index, ix->d.n, d_out);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// the divisor array to use for the *index slice* of each heap vector, or nullptr when no normalisation applies
static int index_slice_norms(vs_index* ix, float** out, bool* owned) {
    *out = nullptr;
    *owned = false;
    if (ix->d.distance_type != VS_COSINE) return VS_OK;
    if (ix->d.dim_index == ix->d.dim_full) {
        VS_TRY(launch_row_norms(ix));
        *out = ix->vnorm;
        return VS_OK;
    }
    float* tmp = nullptr;
    VS_HIP(hipMalloc(&tmp, (size_t)std::max<uint32_t>(ix->d.n, 1) * 4));
    uint32_t blocks = std::min<uint32_t>((ix->d.n + 63) / 64, 8192);
    hipLaunchKernelGGL(k_slice_norms, dim3(blocks), dim3(WAVE), 0, ix->ctx->stream, ix->vecs, ix->vec_stride,
                       ix->d.dim_index, ix->d.n, tmp);
    *out = tmp;
    *owned = true;
    return VS_OK;
}

extern "C" int vs_sbq_train(vs_index* ix) {
    VS_REQUIRE(ix && ix->vecs, "vs_sbq_train: needs the vector column on the device");
    VS_HIP(hipSetDevice(ix->ctx->device));
    float* rn = nullptr;
    bool owned = false;
    VS_TRY(index_slice_norms(ix, &rn, &owned));
    hipLaunchKernelGGL(k_sbq_train, dim3((ix->d.dim_index + WAVE - 1) / WAVE), dim3(WAVE), 0, ix->ctx->stream, ix->vecs,
                       ix->vec_stride, ix->d.n, ix->d.dim_index, ix->d.bits, rn, ix->mean, ix->m2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ix->ctx->stream);
    if (owned) (void)hipFree(rn);
    VS_HIP(e);
    ix->count = ix->d.n;
    return VS_OK;
}

// ===============================================================================================================
// Corpus quantisation: codes[i] = SbqQuantizer::quantize(normalised index slice of vecs[i]).  One wave per row,
// lane = output bit (ballot packs 64 bits per word).
// ===============================================================================================================
__global__ __launch_bounds__(WAVE) void k_quantize_corpus(const float* __restrict__ vecs, uint32_t vec_stride, uint32_t n,
                                                          uint32_t dims, uint32_t bits, const float* __restrict__ rnorm,
                                                          const float* __restrict__ mean, const float* __restrict__ m2,
                                                          float count_f, uint32_t words, uint32_t code_stride,
                                                          uint64_t* __restrict__ codes) {
    const int lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float* v = vecs + (size_t)r * vec_stride;
        const float s = rnorm ? rnorm[r] : 0.0f;
        uint64_t* out = codes + (size_t)r * code_stride;
        for (uint32_t w = 0; w < code_stride; ++w) {
            uint64_t word = 0;
            if (w < words) {
                uint32_t g = w * 64u + (uint32_t)lane;
                uint32_t dim = g / bits;
                uint32_t j = g - dim * bits;
                bool bit = false;
                if (dim < dims) {
                    float x = v[dim];
                    if (s != 0.0f) x = x / s;
                    float mu = mean[dim];
                    if (bits == 1) {
                        bit = x > mu;
                    } else {
                        float variance = m2[dim] / count_f;
                        float std_dev = sqrtf(variance);
                        float ranges = (float)(bits + 1);
                        float z = (x - mu) / std_dev;
                        float index = (z + 2.0f) / (4.0f / ranges);
                        uint32_t ones = 0;
                        if (!(index < 1.0f)) {
                            float fl = floorf(index);
                            if (fl != fl) ones = 0;
                            else if (fl >= (float)bits) ones = bits;
                            else if (fl <= 0.0f) ones = 0;
                            else ones = (uint32_t)fl;
                        }
                        bit = j < ones;
                    }
                }
                word = __ballot(bit);
            }
            if (lane == 0) out[w] = word;
        }
    }
}

extern "C" int vs_sbq_quantize_corpus(vs_index* ix) {
    VS_REQUIRE(ix && ix->vecs, "vs_sbq_quantize_corpus: needs the vector column on the device");
    VS_REQUIRE(ix->count > 0, "vs_sbq_quantize_corpus: quantizer not trained");
    VS_HIP(hipSetDevice(ix->ctx->device));
    float* rn = nullptr;
    bool owned = false;
    VS_TRY(index_slice_norms(ix, &rn, &owned));
    uint32_t grid = std::min<uint32_t>(ix->d.n, 1u << 20);
    if (grid)
        hipLaunchKernelGGL(k_quantize_corpus, dim3(grid), dim3(WAVE), 0, ix->ctx->stream, ix->vecs, ix->vec_stride, ix->d.n,
                           ix->d.dim_index, ix->d.bits, rn, ix->mean, ix->m2, (float)ix->count, ix->d.words,
                           ix->code_stride, ix->codes);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ix->ctx->stream);
    if (owned) (void)hipFree(rn);
    VS_HIP(e);
    return VS_OK;
}

// ===============================================================================================================
// K5 host entry: flat SBQ scan (kernels in vs_scan.hip)
// ===============================================================================================================
static int scan_topk_host(vs_index* ix, const uint64_t* qcodes, const int16_t* qlabels, const uint32_t* qlabel_off, int live_only,
                          uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham);

static int vs_scan_topk_impl(vs_index* ix, const uint64_t* qcodes, uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham) {
    return scan_topk_host(ix, qcodes, nullptr, nullptr, 0, nq, k, out_ids, out_ham);
}
extern "C" int vs_scan_topk(vs_index* ix, const uint64_t* qcodes, uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham) {
    return vs_guard("vs_scan_topk", [&] { return vs_scan_topk_impl(ix, qcodes, nq, k, out_ids, out_ham); });
}


static int vs_scan_topk_filtered_impl(vs_index* ix, const uint64_t* qcodes, const int16_t* qlabels, const uint32_t* qlabel_off,
                                     int live_only, uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham) {
    return scan_topk_host(ix, qcodes, qlabels, qlabel_off, live_only, nq, k, out_ids, out_ham);
}
extern "C" int vs_scan_topk_filtered(vs_index* ix, const uint64_t* qcodes, const int16_t* qlabels, const uint32_t* qlabel_off,
                                     int live_only, uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham) {
    return vs_guard("vs_scan_topk_filtered", [&] { return vs_scan_topk_filtered_impl(ix, qcodes, qlabels, qlabel_off, live_only, nq, k, out_ids, out_ham); });
}


static int scan_topk_host(vs_index* ix, const uint64_t* qcodes, const int16_t* qlabels, const uint32_t* qlabel_off, int live_only,
                          uint32_t nq, uint32_t k, uint32_t* out_ids, uint32_t* out_ham) {
    VS_REQUIRE(ix && (nq == 0 || (qcodes && out_ids)), "vs_scan_topk: bad args");
    if (nq == 0) return VS_OK;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    SearchWorkspace& w = ix->ws;
    const uint32_t W = ix->d.words, cs = ix->code_stride;
    // repack [nq][W] -> [nq][code_stride] (zero padded) through the pinned staging path
    std::vector<uint64_t> padded((size_t)nq * cs, 0);
    for (uint32_t q = 0; q < nq; ++q) memcpy(&padded[(size_t)q * cs], qcodes + (size_t)q * W, (size_t)W * 8);
    VS_TRY(devbuf_reserve(c, w.qcodes, padded.size() * 8));
    VS_TRY(devbuf_reserve(c, w.out_ids, (size_t)nq * k * 4));
    VS_TRY(devbuf_reserve(c, w.stream_ham, (size_t)nq * k * 4));
    VS_TRY(vs_dev_upload(c, w.qcodes.p, padded.data(), padded.size() * 8));
    const int16_t* d_ql = nullptr;
    const uint32_t* d_qo = nullptr;
    if (qlabel_off) {  // keys as LabelSet::from makes them: sorted, de-duplicated (AM/labels/mod.rs:30-37)
        VS_REQUIRE(ix->d.has_labels && ix->label_off, "vs_scan_topk_filtered: label keys on an index without labels");
        std::vector<int16_t> vals;
        std::vector<uint32_t> off(nq + 1, 0);
        for (uint32_t q = 0; q < nq; ++q) {
            VS_REQUIRE(qlabel_off[q] <= qlabel_off[q + 1], "qlabel_off must be non-decreasing");
            std::vector<int16_t> l(qlabels + qlabel_off[q], qlabels + qlabel_off[q + 1]);
            std::sort(l.begin(), l.end());
            l.erase(std::unique(l.begin(), l.end()), l.end());
            vals.insert(vals.end(), l.begin(), l.end());
            off[q + 1] = (uint32_t)vals.size();
        }
        VS_TRY(devbuf_reserve(c, w.qlabels, std::max<size_t>(vals.size(), 1) * 2));
        VS_TRY(devbuf_reserve(c, w.qlabel_off, off.size() * 4));
        if (!vals.empty()) VS_TRY(vs_dev_upload(c, w.qlabels.p, vals.data(), vals.size() * 2));
        VS_TRY(vs_dev_upload(c, w.qlabel_off.p, off.data(), off.size() * 4));
        d_ql = (const int16_t*)w.qlabels.p;
        d_qo = (const uint32_t*)w.qlabel_off.p;
    }
    VS_TRY(launch_scan_topk(ix, (const uint64_t*)w.qcodes.p, nq, k, (uint32_t*)w.out_ids.p, (uint32_t*)w.stream_ham.p, d_ql, d_qo,
                            live_only != 0));
    VS_TRY(vs_dev_download(c, out_ids, w.out_ids.p, (size_t)nq * k * 4));
    if (out_ham) VS_TRY(vs_dev_download(c, out_ham, w.stream_ham.p, (size_t)nq * k * 4));
    return VS_OK;
}

// ===============================================================================================================
// Exact brute force (ground truth for recall): the f32 distance of every row in the reference's accumulation order
// (k_rerank on contiguous row chunks) folded into a running top-k per query, order (distance total_cmp, node id).
// One wave per query; its list lives in global memory between chunks.
// ===============================================================================================================
__device__ __forceinline__ uint32_t bf_total_key(float f) {  // monotone u32 image of f32::total_cmp
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(WAVE) void k_bf_fold(const float* __restrict__ dist, const uint64_t* __restrict__ tids, uint32_t chunk,
                                                  uint32_t row_base, uint32_t nq, uint32_t k,
                                                  uint64_t* __restrict__ lists /*[nq][k] ascending*/) {
    __shared__ uint64_t lst[64];
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const int lane = threadIdx.x;
    if ((uint32_t)lane < k) lst[lane] = lists[(size_t)q * k + lane];
    __syncthreads();
    uint64_t th = lst[k - 1];
    const float* d = dist + (size_t)q * chunk;
    for (uint32_t i0 = 0; i0 < chunk; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        uint64_t key = ~0ull;
        // deleted tuples (InvalidOffsetNumber) are never returned by a scan: not part of the ground truth either
        if (i < chunk && (tids[row_base + i] & 0xFFFFull) != 0) key = ((uint64_t)bf_total_key(d[i]) << 32) | (row_base + i);
        uint64_t hit = __ballot(key < th);
        while (hit) {
            const int src = __builtin_ctzll(hit);
            hit &= hit - 1;
            const uint64_t kk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
            if (kk < th) {
                uint64_t b = ~0ull;
                if ((uint32_t)lane < k) b = lst[lane];
                const uint32_t pos = (uint32_t)__popcll(__ballot((uint32_t)lane < k && b < kk));
                __syncthreads();
                if ((uint32_t)lane >= pos && (uint32_t)lane + 1 < k) lst[lane + 1] = b;
                if ((uint32_t)lane == pos) lst[pos] = kk;
                __syncthreads();
                th = lst[k - 1];
            }
        }
    }
    if ((uint32_t)lane < k) lists[(size_t)q * k + lane] = lst[lane];
}

extern "C" int vs_bruteforce_topk(vs_index* ix, const float* d_queries, uint32_t nq, uint32_t k, uint32_t* out_ids, float* out_dist) {
    VS_REQUIRE(ix && (nq == 0 || (d_queries && out_ids)), "vs_bruteforce_topk: bad args");
    VS_REQUIRE(ix->vecs, "vs_bruteforce_topk: needs the vector column on the device");
    VS_REQUIRE(k >= 1 && k <= 64, "vs_bruteforce_topk: k must be in [1, 64]");
    if (nq == 0) return VS_OK;
    vs_ctx* c = ix->ctx;
    VS_HIP(hipSetDevice(c->device));
    SearchWorkspace& w = ix->ws;
    const uint32_t n = ix->d.n;
    // query preparation exactly as for a scan (cosine normalisation of the full slice)
    VS_TRY(devbuf_reserve(c, w.q_full, (size_t)nq * ix->vec_stride * 4));
    VS_TRY(devbuf_reserve(c, w.qcodes, (size_t)nq * ix->code_stride * 8));
    VS_TRY(launch_prepare_queries(ix, d_queries, nq, (float*)w.q_full.p, (uint64_t*)w.qcodes.p));
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(n, 1), std::max<uint64_t>(4096, (256ull << 20) / nq));
    VS_TRY(devbuf_reserve(c, w.rr_dist, (size_t)nq * chunk * 4));
    VS_TRY(devbuf_reserve(c, w.resort_heap, (size_t)nq * k * 8));
    VS_HIP(hipMemsetAsync(w.resort_heap.p, 0xFF, (size_t)nq * k * 8, c->stream));
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - r0);
        VS_TRY(launch_rerank(ix, (const float*)w.q_full.p, nullptr, nullptr, nullptr, m, nq, (float*)w.rr_dist.p, (uint32_t)r0));
        hipLaunchKernelGGL(k_bf_fold, dim3(nq), dim3(WAVE), 0, c->stream, (const float*)w.rr_dist.p, ix->tids, m, (uint32_t)r0, nq, k,
                           (uint64_t*)w.resort_heap.p);
        VS_HIP(hipGetLastError());
    }
    std::vector<uint64_t> lists((size_t)nq * k);
    VS_TRY(vs_dev_download(c, lists.data(), w.resort_heap.p, lists.size() * 8));
    for (size_t i = 0; i < lists.size(); ++i) {
        const uint64_t e = lists[i];
        if (e == ~0ull) {
            out_ids[i] = VS_INVALID_NODE;
            if (out_dist) out_dist[i] = NAN;
        } else {
            out_ids[i] = (uint32_t)e;
            uint32_t kb = (uint32_t)(e >> 32);
            kb = (kb & 0x80000000u) ? (kb & 0x7FFFFFFFu) : ~kb;
            float f;
            memcpy(&f, &kb, 4);
            if (out_dist) out_dist[i] = f;
        }
    }
    return VS_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Label sets as 64-bit masks (an index whose labels all lie in 0..63 — the usual smallint tags): the overlap test of the scan
// (LabelSetView::overlaps, AM/labels/mod.rs:124-142) becomes one 8-byte load and an AND instead of two dependent loads and a merge.
// ---------------------------------------------------------------------------------------------------------------
// label_bit[(uint16_t)label] = the bit that stands for the label in the masks (0xFF: the label occurs nowhere in the index)
__global__ __launch_bounds__(256) void k_label_presence(const int16_t* __restrict__ val, uint64_t nvals, uint8_t* __restrict__ present) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvals; i += (uint64_t)gridDim.x * blockDim.x)
        present[(uint16_t)val[i]] = 1;  // (benign race: every writer stores 1)
}
__global__ __launch_bounds__(256) void k_label_masks(const uint32_t* __restrict__ off, const int16_t* __restrict__ val, uint32_t n,
                                                     const uint8_t* __restrict__ label_bit, uint64_t* __restrict__ mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t m = 0;
    for (uint32_t j = off[i]; j < off[i + 1]; ++j) m |= 1ull << label_bit[(uint16_t)val[j]];
    mask[i] = m;
}

// nbr_mask[i][j] = label_mask[nbrs[i][j]] (0 past the end of the list): what a label-filtered visit needs of its fresh neighbors,
// laid out next to the neighbor row so that one coalesced load brings all of it
__global__ __launch_bounds__(256) void k_nbr_masks(const uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t n,
                                                   const uint64_t* __restrict__ label_mask, uint64_t* __restrict__ out) {
    const uint64_t total = (uint64_t)n * nbr_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = nbrs[i];
        out[i] = v < n ? label_mask[v] : 0ull;
    }
}

extern "C" int vs_index_has_neighbor_masks(const vs_index* ix) { return ix && ix->nbr_mask_valid ? 1 : 0; }

bool vs_neighbor_masks_wanted(const vs_index* ix) {
    const char* e = vs_opt_get("VS_F_NBRMASK");
    if (e && *e == '0') return false;
    if (e && *e == '1') return true;
    return ix->d.n > (8u << 20);
}

int vs_refresh_neighbor_masks(vs_index* ix) {
    if (ix->nbr_mask_valid || ix->is_view || !ix->label_mask || !ix->nbrs || ix->d.n == 0) return VS_OK;
    // worth it once the node masks (8 bytes per node) no longer sit in the caches: +6.7 % at 20M x 1536 (77.9 -> 72.7 ms per 131072
    // scans, profiles/r03/bench_cfg5*.json), +0.6 % at 5M (s6_nbrmask_5m_summary.txt), for 8 x nbr_stride bytes per node.  So: built
    // for indexes of more than 8M nodes (64 MB of masks), VS_F_NBRMASK=1 / 0 forces it on / off.
    if (!vs_neighbor_masks_wanted(ix)) return VS_OK;
    vs_ctx* c = ix->ctx;
    const size_t bytes = (size_t)ix->d.n * ix->nbr_stride * 8;
    if (!ix->nbr_mask) {
        if (ix->nbr_mask_tried) return VS_OK;
        ix->nbr_mask_tried = true;
        size_t free_b = 0, total_b = 0;
        // a cache, not a requirement: it is only built when it leaves the search workspace plenty of room
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bytes + total_b / 16) return VS_OK;
        if (hipMalloc(&ix->nbr_mask, bytes) != hipSuccess) {
            (void)hipGetLastError();
            ix->nbr_mask = nullptr;
            return VS_OK;
        }
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)ix->d.n * ix->nbr_stride + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(k_nbr_masks, dim3(grid), dim3(256), 0, c->stream, ix->nbrs, ix->nbr_stride, ix->d.n, ix->label_mask, ix->nbr_mask);
    VS_HIP(hipGetLastError());
    ix->nbr_mask_valid = true;  // (same stream as the searches that follow: ordered)
    return VS_OK;
}

// When the index uses at most 64 DISTINCT labels (any smallint values), every node's label set becomes a 64-bit mask through a
// per-index label -> bit table, and the overlap test of a scan is one load and an AND; with more distinct labels the scans keep
// the sorted-merge test on the CSR (LabelSetView::overlaps, AM/labels/mod.rs:124-142).
int vs_refresh_label_masks(vs_index* ix) {
    vs_ctx* c = ix->ctx;
    ix->nbr_mask_valid = false;  // (derived from the masks below)
    ix->nbr_mask_tried = false;
    if (ix->label_mask) {
        VS_HIP(hipFree(ix->label_mask));
        ix->label_mask = nullptr;
    }
    if (ix->label_bit) {
        VS_HIP(hipFree(ix->label_bit));
        ix->label_bit = nullptr;
    }
    if (!ix->label_off || !ix->label_val || ix->d.n == 0) return VS_OK;
    uint8_t* tab = nullptr;
    VS_HIP(hipMalloc(&tab, 65536));
    std::vector<uint8_t> h(65536, 0);
    hipError_t e = hipMemsetAsync(tab, 0, 65536, c->stream);
    if (e == hipSuccess && ix->n_label_vals) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((ix->n_label_vals + 255) / 256, 65536);
        hipLaunchKernelGGL(k_label_presence, dim3(grid), dim3(256), 0, c->stream, ix->label_val, ix->n_label_vals, tab);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), tab, 65536, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        (void)hipFree(tab);
        VS_HIP(e);
    }
    // bits are dealt in the order of the labels as signed smallints (only "which bit" matters, not the order)
    uint32_t distinct = 0;
    for (int v = -32768; v <= 32767; ++v) {
        uint8_t& slot = h[(uint16_t)(int16_t)v];
        if (slot) slot = distinct < 64 ? (uint8_t)distinct : 0xFF, ++distinct;
        else slot = 0xFF;
    }
    if (distinct > 64) {  // the scans keep the merge
        (void)hipFree(tab);
        return VS_OK;
    }
    uint64_t* m = nullptr;
    e = hipMalloc(&m, (size_t)ix->d.n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(tab, h.data(), 65536, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_label_masks, dim3((ix->d.n + 255) / 256), dim3(256), 0, c->stream, ix->label_off, ix->label_val, ix->d.n, tab, m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        (void)hipFree(tab);
        if (m) (void)hipFree(m);
        VS_HIP(e);
    }
    ix->label_mask = m;
    ix->label_bit = tab;
    return VS_OK;
}
