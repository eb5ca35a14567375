// vs_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the StreamingDiskANN search path.
//
//   K4 k_prepare_queries / k_quantize_rows : SbqQuantizer::quantize            (AM/sbq/quantize.rs:52-102)
//   K1 k_hamming_gather                    : distance_xor_optimized on gathers (AM/distance/mod.rs:266-323)
//   K2 k_rerank                            : distance_l2/cosine/inner_product in the reference's AVX2 accumulation
//                                            order (AM/distance/mod.rs:325-435, AM/sbq/storage.rs:304-328)
//   K3 k_search (vs_search.hip)            : ListSearchResult + greedy_search_iterate + visit_lsn_internal +
//                                            TSVResponseIterator::next (AM/graph/mod.rs:74-185,357-385,
//                                            AM/sbq/storage.rs:135-190, AM/scan.rs:210-242)
//      k_resort                            : the rescore window of next_with_resort (AM/scan.rs:244-305)
//
// All of this is HBM-latency / bandwidth bound integer + f32 dot work: no MFMA.  One wave64 owns one query in
// K3 (the search is a serial chain of dependent expansions), lanes cooperate on the R gathered neighbor codes
// (4 lanes x 16 B per code row), the candidate heap / visited list live in LDS, the dedup hash set in L2.
// Compiled with -ffp-contract=off: the reference's L2 kernel uses separate mul+add, its dot kernel FMA.
#include "vs_internal.h"
#include "vs_device.h"

// ---------------------------------------------------------------------------------------------------------------
// preprocess_cosine on a vector held in LDS (AM/distance/mod.rs:225-253): sequential f32 sum of squares (lane 0),
// then every lane divides.  Returns nothing; buf is normalised in place.
// ---------------------------------------------------------------------------------------------------------------
__device__ void lds_preprocess_cosine(float* buf, uint32_t n, int lane, float* bcast /* LDS scratch [1] */) {
    __syncthreads();
    if (lane == 0) {
        float norm = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            float p = buf[i] * buf[i];
            norm = norm + p;
        }
        const float eps = 1.1920929e-07f;  // f32::EPSILON
        float adj = eps * (float)n;
        float s = 0.0f;  // 0 => leave alone
        if (!(norm < eps) && !(norm >= 1.0f - adj && norm <= 1.0f + adj)) s = sqrtf(norm);
        *bcast = s;
    }
    __syncthreads();
    float s = *bcast;
    if (s != 0.0f) {
        for (uint32_t i = lane; i < n; i += blockDim.x) buf[i] = buf[i] / s;
    }
    __syncthreads();
}

// SbqQuantizer::quantize of a vector in LDS, one wave: lane = bit position inside the output word (ballot packs).
__device__ void wave_quantize(const float* v, uint32_t dims, uint32_t bits, const float* __restrict__ mean,
                              const float* __restrict__ m2, float count_f, uint64_t* out, uint32_t words,
                              uint32_t out_stride, int lane) {
    for (uint32_t w = 0; w < out_stride; ++w) {
        uint64_t word = 0;
        if (w < words) {
            uint32_t g = w * 64u + (uint32_t)lane;  // global bit index
            uint32_t dim = g / bits;
            uint32_t j = g - dim * bits;
            bool bit = false;
            if (dim < dims) {
                float x = v[dim];
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
                    if (!(index < 1.0f)) {  // NaN falls through like Rust's `if index < 1.0 {} else {..}`
                        float fl = floorf(index);
                        // `fl as usize` saturating, NaN -> 0; then min(bits)
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

// ---------------------------------------------------------------------------------------------------------------
// K4a: query preparation = PgVector::from_datum(index=true, full=true) (AM/pg_vector.rs:162-199) +
//      SbqSearchDistanceMeasure::new (AM/sbq/mod.rs:145-148).  One wave per query.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_prepare_queries(const float* __restrict__ raw, uint32_t nq,
                                                          uint32_t dim_full, uint32_t dim_index, uint32_t vec_stride,
                                                          uint32_t distance_type, uint32_t bits,
                                                          const float* __restrict__ mean, const float* __restrict__ m2,
                                                          float count_f, uint32_t words, uint32_t code_stride,
                                                          float* __restrict__ q_full, uint64_t* __restrict__ qcodes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* full = reinterpret_cast<float*>(smem);
    float* idxv = full + round_up_u32(dim_full, 4);
    float* bc = idxv + round_up_u32(dim_index, 4);
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const float* src = raw + (size_t)q * dim_full;
    for (uint32_t i = lane; i < dim_full; i += WAVE) full[i] = src[i];
    const bool same = dim_full == dim_index;
    if (!same)
        for (uint32_t i = lane; i < dim_index; i += WAVE) idxv[i] = src[i];
    __syncthreads();
    if (distance_type == VS_COSINE) {
        lds_preprocess_cosine(full, dim_full, lane, bc);
        if (!same) lds_preprocess_cosine(idxv, dim_index, lane, bc);
    }
    float* qf = q_full + (size_t)q * vec_stride;
    for (uint32_t i = lane; i < vec_stride; i += WAVE) qf[i] = i < dim_full ? full[i] : 0.0f;
    wave_quantize(same ? full : idxv, dim_index, bits, mean, m2, count_f, qcodes + (size_t)q * code_stride, words,
                  code_stride, lane);
}

// plain storage with num_dimensions_to_index < num_dimensions: the graph search compares the INDEX slice of the query,
// cosine-normalised on its own (PgVector::from_datum, AM/pg_vector.rs:143-157), with the stored index-slice vectors
__global__ __launch_bounds__(WAVE) void k_prepare_index_slice(const float* __restrict__ raw, uint32_t nq, uint32_t dim_full,
                                                              uint32_t dim_index, uint32_t vec_stride, uint32_t distance_type,
                                                              float* __restrict__ q_index) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* idxv = reinterpret_cast<float*>(smem);
    float* bc = idxv + round_up_u32(dim_index, 4);
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    for (uint32_t i = lane; i < dim_index; i += WAVE) idxv[i] = raw[(size_t)q * dim_full + i];
    __syncthreads();
    if (distance_type == VS_COSINE) lds_preprocess_cosine(idxv, dim_index, lane, bc);
    for (uint32_t i = lane; i < vec_stride; i += WAVE) q_index[(size_t)q * vec_stride + i] = i < dim_index ? idxv[i] : 0.0f;
}

// K4b: quantize rows that are already prepared (normalised if cosine): one wave per row.
__global__ __launch_bounds__(WAVE) void k_quantize_rows(const float* __restrict__ rows, uint32_t row_stride,
                                                        uint32_t nrows, uint32_t dims, uint32_t bits,
                                                        const float* __restrict__ mean, const float* __restrict__ m2,
                                                        float count_f, uint32_t words, uint32_t code_stride,
                                                        uint64_t* __restrict__ codes) {
    const int lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < nrows; r += gridDim.x)
        wave_quantize(rows + (size_t)r * row_stride, dims, bits, mean, m2, count_f, codes + (size_t)r * code_stride,
                      words, code_stride, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// K1: Hamming distances of gathered code rows.  One wave per query; 16 rows per pass.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_hamming_gather(const uint64_t* __restrict__ codes, uint32_t code_stride,
                                                         const uint64_t* __restrict__ qcodes,
                                                         const uint32_t* __restrict__ ids,
                                                         const uint32_t* __restrict__ off, uint32_t nq,
                                                         uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* qc = reinterpret_cast<uint64_t*>(smem);
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    for (uint32_t w = lane; w < code_stride; w += WAVE) qc[w] = qcodes[(size_t)q * code_stride + w];
    __syncthreads();
    const uint32_t b = off[q], e = off[q + 1];
    for (uint32_t base = b; base < e; base += 16) {
        uint32_t j = base + (uint32_t)(lane >> 2);
        bool valid = j < e;
        uint32_t id = valid ? ids[j] : 0;
        uint32_t d = ham_row4(codes + (size_t)id * code_stride, qc, lane & 3, code_stride, valid);
        if (valid && (lane & 3) == 0) out[j] = d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K2: rerank.  One workgroup (4 waves) per query, query vector staged in LDS; 8 lanes per candidate row, float4
// loads: lane l8 owns elements 32t+4*l8..+3 of every 32-float step, i.e. exactly 4 of the 32 "virtual AVX2 lanes"
// (4 accumulators x 8 lanes) of distance_l2_simd_body!/inner_product_simd_body! (AM/distance/mod.rs:325-435).
// Final reduction replays horizontal_add_ps per accumulator and the left-to-right sum of the 4 accumulators, so the
// result is the same f32 the AVX2 reference produces (bit-for-bit, given the same hadd lane order).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, WAVE); }

__global__ __launch_bounds__(256) void k_rerank(const float* __restrict__ vecs, uint32_t vec_stride, uint32_t dim_full,
                                                const float* __restrict__ vnorm, uint32_t distance_type,
                                                const float* __restrict__ q_full, const uint32_t* __restrict__ ids,
                                                const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt,
                                                uint32_t fixed_m, uint32_t nq, float* __restrict__ out, uint32_t row_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qv = reinterpret_cast<float*>(smem);
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    for (uint32_t i = threadIdx.x; i < vec_stride; i += blockDim.x) qv[i] = q_full[(size_t)q * vec_stride + i];
    __syncthreads();
    uint32_t b, e;
    if (off) {
        b = off[q];
        e = off[q + 1];
    } else {
        b = q * fixed_m;
        e = b + (cnt ? min(cnt[q], fixed_m) : fixed_m);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l8 = lane & 7, grp = lane >> 3;
    const uint32_t steps = dim_full / 32;
    for (uint32_t base = b; base < e; base += 32) {
        uint32_t j = base + (uint32_t)(wave * 8 + grp);
        bool valid = j < e;
        // ids == nullptr: the contiguous rows row_base .. row_base + fixed_m - 1 (exact brute force, vs_bruteforce_topk)
        uint32_t id = valid ? (ids ? ids[j] : row_base + (j - b)) : VS_INVALID_NODE;
        if (id == VS_INVALID_NODE) valid = false;
        const float* row = vecs + (size_t)(valid ? id : 0) * vec_stride;
        float s = 0.0f;
        if (valid && distance_type == VS_COSINE) s = vnorm[id];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (valid) {
            if (distance_type == VS_L2) {
                for (uint32_t t = 0; t < steps; ++t) {
                    float4 x = *reinterpret_cast<const float4*>(row + 32 * t + 4 * l8);
                    float4 y = *reinterpret_cast<const float4*>(qv + 32 * t + 4 * l8);
                    float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                    float p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2, p3 = d3 * d3;
                    a0 = a0 + p0;
                    a1 = a1 + p1;
                    a2 = a2 + p2;
                    a3 = a3 + p3;
                }
            } else {
                for (uint32_t t = 0; t < steps; ++t) {
                    float4 x = *reinterpret_cast<const float4*>(row + 32 * t + 4 * l8);
                    float4 y = *reinterpret_cast<const float4*>(qv + 32 * t + 4 * l8);
                    if (s != 0.0f) {
                        x.x = x.x / s;
                        x.y = x.y / s;
                        x.z = x.z / s;
                        x.w = x.w / s;
                    }
                    a0 = __builtin_fmaf(x.x, y.x, a0);
                    a1 = __builtin_fmaf(x.y, y.y, a1);
                    a2 = __builtin_fmaf(x.z, y.z, a2);
                    a3 = __builtin_fmaf(x.w, y.w, a3);
                }
            }
        }
        // horizontal_add_ps of accumulator j lives on lanes (2j, 2j+1) of the 8-lane group:
        // s_c = a_c + a_{c+4}; h = (s0+s1)+(s2+s3)
        float s0 = a0 + shfl_f(a0, lane ^ 1);
        float s1 = a1 + shfl_f(a1, lane ^ 1);
        float s2 = a2 + shfl_f(a2, lane ^ 1);
        float s3 = a3 + shfl_f(a3, lane ^ 1);
        float t0 = s0 + s1;
        float t1 = s2 + s3;
        float h = t0 + t1;
        const int g0 = lane & ~7;
        float h0 = shfl_f(h, g0 + 0), h1 = shfl_f(h, g0 + 2), h2 = shfl_f(h, g0 + 4), h3 = shfl_f(h, g0 + 6);
        float dist = h0 + h1;
        dist = dist + h2;
        dist = dist + h3;
        if (valid && l8 == 0) {
            for (uint32_t i = steps * 32; i < dim_full; ++i) {  // scalar tail, in element order
                float x = row[i];
                if (distance_type == VS_L2) {
                    float diff = x - qv[i];
                    float p = diff * diff;
                    dist = dist + p;
                } else {
                    if (s != 0.0f) x = x / s;
                    float p = x * qv[i];
                    dist = dist + p;
                }
            }
            float r;
            if (distance_type == VS_L2) r = dist;
            else if (distance_type == VS_IP) r = -dist;
            else r = fmaxf(1.0f - dist, 0.0f);
            out[j] = r;
        }
    }
}

// per-node cosine divisor cache: exact preprocess_cosine_get_norm (sequential f32 sum).  A wave owns 64 rows;
// a [64 rows][64 dims] tile is staged through LDS so global reads stay coalesced while each lane walks one row
// in element order.
__global__ __launch_bounds__(WAVE) void k_row_norms(const float* __restrict__ vecs, uint32_t vec_stride, uint32_t dim,
                                                    uint32_t n, float* __restrict__ vnorm) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    for (uint32_t row0 = blockIdx.x * 64u; row0 < n; row0 += gridDim.x * 64u) {
        float norm = 0.0f;
        for (uint32_t d0 = 0; d0 < dim; d0 += 64) {
            for (int r = 0; r < 64; ++r) {
                uint32_t row = row0 + r;
                uint32_t d = d0 + lane;
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
            vnorm[row] = s;
        }
    }
}

// duplicate ids inside one neighbor list would make the wave-parallel dedup order-dependent: reject them at upload.
__global__ void k_validate_nbrs(const uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R, uint32_t n,
                                uint32_t* flag) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint32_t* r = nbrs + (size_t)row * nbr_stride;
    uint32_t deg = 0;
    while (deg < R && r[deg] != VS_INVALID_NODE) {
        if (r[deg] >= n) atomicOr(flag, 2u);
        ++deg;
    }
    for (uint32_t i = 1; i < deg; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (r[i] == r[j]) atomicOr(flag, 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// Rescore window of next_with_resort (AM/scan.rs:244-305): BinaryHeap<ResortData> with
// cmp(self, other) = other.distance.total_cmp(self.distance)  (AM/scan.rs:111-117).  One thread per query.
// heap entries: (total_cmp key as i32 in the high word, stream position in the low word).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t total_key(float f) {
    int32_t b = __float_as_int(f);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

__global__ void k_resort(uint32_t nq, uint32_t M, uint32_t rescore, uint32_t k, const uint32_t* __restrict__ stream,
                         const uint32_t* __restrict__ cnt, const float* __restrict__ dist, const uint64_t* __restrict__ tids,
                         uint64_t* __restrict__ heap_ws, uint32_t* __restrict__ out_ids, uint64_t* __restrict__ out_tids,
                         float* __restrict__ out_dist, const uint32_t* __restrict__ plain_keys) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t n = min(cnt[q], M);
    const uint32_t* sid = stream + (size_t)q * M;
    const float* sd = dist ? dist + (size_t)q * M : nullptr;
    uint32_t produced = 0;
    if (rescore == 0) {  // resort_buffer.capacity() == 0 -> plain next()
        for (; produced < k && produced < n; ++produced) {
            uint32_t id = sid[produced];
            out_ids[(size_t)q * k + produced] = id;
            if (out_tids) out_tids[(size_t)q * k + produced] = tids[id];
            if (out_dist) {
                float d = __int_as_float(0x7fc00000);
                if (plain_keys) {  // plain storage: the graph distance IS the full-precision distance (key = total_cmp image)
                    int32_t b = (int32_t)(plain_keys[(size_t)q * M + produced] ^ 0x80000000u);
                    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
                    d = __int_as_float(b);
                }
                out_dist[(size_t)q * k + produced] = d;
            }
        }
    } else {
        uint64_t* h = heap_ws + (size_t)q * rescore;
        uint32_t len = 0, pos = 0;
        // le(a,b) (Rust a <= b for ResortData) == key(b) <= key(a)
        auto kof = [](uint64_t e) { return (int32_t)(uint32_t)(e >> 32); };
        auto sift_up = [&](uint32_t p, uint64_t elem) {
            while (p > 0) {
                uint32_t parent = (p - 1) >> 1;
                uint64_t pe = h[parent];
                if (kof(pe) <= kof(elem)) break;  // elem <= parent
                h[p] = pe;
                p = parent;
            }
            h[p] = elem;
        };
        while (produced < k) {
            while (len < rescore && pos < n) {
                uint64_t e = ((uint64_t)(uint32_t)total_key(sd[pos]) << 32) | pos;
                uint32_t p = len++;
                sift_up(p, e);
                ++pos;
            }
            if (len == 0) break;
            uint64_t item = h[--len];
            uint64_t top = item;
            if (len > 0) {
                top = h[0];
                uint32_t end = len, p = 0, child = 1;
                uint32_t lim = end >= 2 ? end - 2 : 0;
                while (child <= lim) {
                    uint64_t le = h[child], ri = h[child + 1];
                    uint32_t pick = (kof(ri) <= kof(le)) ? 1u : 0u;  // data[child] <= data[child+1]
                    child += pick;
                    h[p] = pick ? ri : le;
                    p = child;
                    child = 2 * p + 1;
                }
                if (child == end - 1) {
                    h[p] = h[child];
                    p = child;
                }
                sift_up(p, item);
            }
            uint32_t sp = (uint32_t)top;
            uint32_t id = sid[sp];
            out_ids[(size_t)q * k + produced] = id;
            if (out_tids) out_tids[(size_t)q * k + produced] = tids[id];
            if (out_dist) out_dist[(size_t)q * k + produced] = sd[sp];
            ++produced;
        }
    }
    for (; produced < k; ++produced) {
        out_ids[(size_t)q * k + produced] = VS_INVALID_NODE;
        if (out_tids) out_tids[(size_t)q * k + produced] = 0;
        if (out_dist) out_dist[(size_t)q * k + produced] = __int_as_float(0x7fc00000);
    }
}

// The same window, resumable (the amgettuple cursor): the BinaryHeap<ResortData> of ONE scan lives in heap_ws between calls,
// cur[0] = its length, cur[1] = stream rows pushed so far, cur[2] = rows handed out so far.  `stream` / `dist` / `keys` hold every
// row the scan has emitted so far (n of them; `exhausted` = there will be no more).  Produces up to k more rows; a window that
// cannot be refilled because the rows are not there yet (n too small, scan not exhausted) stops early — the host fetches so
// that this never happens (rows >= rescore + handed out + k - 1).  cur[3] = rows produced by this call.
__global__ void k_resort_cursor(uint32_t n, uint32_t exhausted, uint32_t rescore, uint32_t k, const uint32_t* __restrict__ stream,
                                const float* __restrict__ dist, const uint32_t* __restrict__ keys, uint32_t plain_keys,
                                const uint64_t* __restrict__ tids, uint64_t* __restrict__ h, uint32_t* __restrict__ cur,
                                uint32_t* __restrict__ out_ids, uint64_t* __restrict__ out_tids, float* __restrict__ out_dist) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t len = cur[0], pos = cur[1];
    uint32_t produced = 0;
    if (rescore == 0) {  // resort_buffer.capacity() == 0 -> plain next()
        for (; produced < k && pos < n; ++produced, ++pos) {
            const uint32_t id = stream[pos];
            out_ids[produced] = id;
            out_tids[produced] = tids[id];
            float d = __int_as_float(0x7fc00000);
            if (plain_keys) {
                int32_t b = (int32_t)(keys[pos] ^ 0x80000000u);
                b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
                d = __int_as_float(b);
            }
            out_dist[produced] = d;
        }
    } else {
        auto kof = [](uint64_t e) { return (int32_t)(uint32_t)(e >> 32); };
        auto sift_up = [&](uint32_t p, uint64_t elem) {
            while (p > 0) {
                const uint32_t parent = (p - 1) >> 1;
                const uint64_t pe = h[parent];
                if (kof(pe) <= kof(elem)) break;
                h[p] = pe;
                p = parent;
            }
            h[p] = elem;
        };
        while (produced < k) {
            while (len < rescore && pos < n) {
                const uint64_t e = ((uint64_t)(uint32_t)total_key(dist[pos]) << 32) | pos;
                sift_up(len++, e);
                ++pos;
            }
            if (len < rescore && !exhausted) break;  // the window cannot be filled yet
            if (len == 0) break;
            const uint64_t item = h[--len];
            uint64_t top = item;
            if (len > 0) {
                top = h[0];
                const uint32_t end = len;
                uint32_t p = 0, child = 1;
                const uint32_t lim = end >= 2 ? end - 2 : 0;
                while (child <= lim) {
                    const uint64_t le = h[child], ri = h[child + 1];
                    const uint32_t pick = (kof(ri) <= kof(le)) ? 1u : 0u;
                    child += pick;
                    h[p] = pick ? ri : le;
                    p = child;
                    child = 2 * p + 1;
                }
                if (child == end - 1) {
                    h[p] = h[child];
                    p = child;
                }
                sift_up(p, item);
            }
            const uint32_t sp = (uint32_t)top;
            const uint32_t id = stream[sp];
            out_ids[produced] = id;
            out_tids[produced] = tids[id];
            out_dist[produced] = dist[sp];
            ++produced;
        }
    }
    cur[0] = len;
    cur[1] = pos;
    cur[2] += produced;
    cur[3] = produced;
}

// The resumable window for MANY scans in one launch (the scan pools' fetch, vs_scanpool.cpp): one wave per listed scan, scans side by
// side on the chip instead of one single-thread launch per scan.  list[3 b ..] = (pool slot, stream rows the scan has emitted so
// far, exhausted); the arrays of slot q are row q of the pool's 2-D arrays (strides below).  Inside a scan the heap mechanics are
// serial — Rust's BinaryHeap<ResortData>, AM/scan.rs:111-117,244-305, replayed exactly as in k_resort_cursor — but they run on an LDS
// copy of the heap (one coalesced load, one store) against keys the wave fetched ahead (total_cmp images of the distances the
// refills will push), and the k result rows are gathered by k lanes at once afterwards (stream id, heap tid, distance: two
// dependent round trips in all instead of three per row).  cur[] as in k_resort_cursor.
__global__ __launch_bounds__(WAVE) void k_resort_cursor_batch(const uint32_t* __restrict__ list, uint32_t rescore, uint32_t k,
                                                               const uint32_t* __restrict__ stream_base, const float* __restrict__ dist_base,
                                                               const uint32_t* __restrict__ keys_base, uint32_t row_stride, uint32_t plain_keys,
                                                               const uint64_t* __restrict__ tids, uint64_t* __restrict__ heap_base,
                                                               uint32_t* __restrict__ cur_base, uint32_t* __restrict__ out_ids_base,
                                                               uint64_t* __restrict__ out_tids_base, float* __restrict__ out_dist_base,
                                                               uint32_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* hh = reinterpret_cast<uint64_t*>(smem);                 // rescore entries
    uint32_t* kk = reinterpret_cast<uint32_t*>(hh + rescore);        // rescore + k keys of the rows the refills will push
    uint32_t* sps = kk + rescore + k;                                // k stream positions of the rows produced
    const uint32_t lane = threadIdx.x;
    const uint32_t q = list[3 * blockIdx.x], n = list[3 * blockIdx.x + 1], exhausted = list[3 * blockIdx.x + 2];
    const uint32_t* stream = stream_base + (size_t)q * row_stride;
    const float* dist = dist_base + (size_t)q * row_stride;
    const uint32_t* keys = keys_base + (size_t)q * row_stride;
    uint64_t* h = heap_base + (size_t)q * rescore;
    uint32_t* cur = cur_base + (size_t)q * 4;
    uint32_t* out_ids = out_ids_base + (size_t)q * out_stride;
    uint64_t* out_tids = out_tids_base + (size_t)q * out_stride;
    float* out_dist = out_dist_base + (size_t)q * out_stride;
    uint32_t len = cur[0], pos = cur[1];
    uint32_t produced = 0;
    if (rescore == 0) {  // resort_buffer.capacity() == 0 -> plain next(): rows pos .. pos + k - 1 as they are
        produced = min(k, n > pos ? n - pos : 0u);
        for (uint32_t j = lane; j < produced; j += WAVE) sps[j] = pos + j;
        pos += produced;
    } else {
        const uint32_t pos0 = pos;
        const uint32_t ahead = min(n > pos ? n - pos : 0u, rescore - len + k);  // (a call pushes at most that many rows)
        for (uint32_t i = lane; i < len; i += WAVE) hh[i] = h[i];
        for (uint32_t i = lane; i < ahead; i += WAVE) kk[i] = (uint32_t)total_key(dist[pos0 + i]);
        __syncthreads();
        if (lane == 0) {
            auto kof = [](uint64_t e) { return (int32_t)(uint32_t)(e >> 32); };
            auto sift_up = [&](uint32_t p, uint64_t elem) {
                while (p > 0) {
                    const uint32_t parent = (p - 1) >> 1;
                    const uint64_t pe = hh[parent];
                    if (kof(pe) <= kof(elem)) break;
                    hh[p] = pe;
                    p = parent;
                }
                hh[p] = elem;
            };
            while (produced < k) {
                while (len < rescore && pos < n) {
                    const uint64_t e = ((uint64_t)kk[pos - pos0] << 32) | pos;
                    sift_up(len++, e);
                    ++pos;
                }
                if (len < rescore && !exhausted) break;  // the window cannot be filled yet
                if (len == 0) break;
                const uint64_t item = hh[--len];
                uint64_t top = item;
                if (len > 0) {
                    top = hh[0];
                    const uint32_t end = len;
                    uint32_t p = 0, child = 1;
                    const uint32_t lim = end >= 2 ? end - 2 : 0;
                    while (child <= lim) {
                        const uint64_t le = hh[child], ri = hh[child + 1];
                        const uint32_t pick = (kof(ri) <= kof(le)) ? 1u : 0u;
                        child += pick;
                        hh[p] = pick ? ri : le;
                        p = child;
                        child = 2 * p + 1;
                    }
                    if (child == end - 1) {
                        hh[p] = hh[child];
                        p = child;
                    }
                    sift_up(p, item);
                }
                sps[produced++] = (uint32_t)top;
            }
            sps[k] = produced;  // (one word past the k positions: the hand-over to the other lanes)
            sps[k + 1] = len;
            sps[k + 2] = pos;
        }
        __syncthreads();
        produced = sps[k];
        len = sps[k + 1];
        pos = sps[k + 2];
        for (uint32_t i = lane; i < len; i += WAVE) h[i] = hh[i];
    }
    __syncthreads();
    for (uint32_t j = lane; j < produced; j += WAVE) {
        const uint32_t sp = sps[j];
        const uint32_t id = stream[sp];
        out_ids[j] = id;
        out_tids[j] = tids[id];
        float d = __int_as_float(0x7fc00000);
        if (rescore != 0) d = dist[sp];
        else if (plain_keys) {
            int32_t b = (int32_t)(keys[sp] ^ 0x80000000u);
            b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
            d = __int_as_float(b);
        }
        out_dist[j] = d;
    }
    if (lane == 0) {
        cur[0] = len;
        cur[1] = pos;
        cur[2] += produced;
        cur[3] = produced;
    }
}

// ===============================================================================================================
// launch wrappers
// ===============================================================================================================
static float count_as_f32(uint64_t c) { return (float)c; }

int launch_prepare_queries(vs_index* idx, const float* d_raw, uint32_t nq, float* d_q_full, uint64_t* d_qcodes) {
    if (nq == 0) return VS_OK;
    const vs_index_desc& d = idx->d;
    size_t lds = (round_up_u32(d.dim_full, 4) + round_up_u32(d.dim_index, 4) + 4) * sizeof(float);
    VS_REQUIRE(lds <= 160 * 1024, "query too large for LDS staging (%u dims)", d.dim_full);
    hipLaunchKernelGGL(k_prepare_queries, dim3(nq), dim3(WAVE), lds, idx->ctx->stream, d_raw, nq, d.dim_full,
                       d.dim_index, idx->vec_stride, d.distance_type, d.bits, idx->mean, idx->m2,
                       count_as_f32(idx->count), d.storage_type == VS_STORAGE_PLAIN ? 0u : d.words /* plain: no SBQ code */,
                       idx->code_stride, d_q_full, d_qcodes);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_prepare_index_slice(vs_index* idx, const float* d_raw, uint32_t nq, float* d_q_index) {
    if (nq == 0) return VS_OK;
    const vs_index_desc& d = idx->d;
    const size_t lds = (round_up_u32(d.dim_index, 4) + 4) * sizeof(float);
    hipLaunchKernelGGL(k_prepare_index_slice, dim3(nq), dim3(WAVE), lds, idx->ctx->stream, d_raw, nq, d.dim_full, d.dim_index,
                       idx->vec_stride, d.distance_type, d_q_index);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_quantize_rows(vs_index* idx, const float* d_rows, uint32_t row_stride, uint32_t nrows, uint64_t* d_codes,
                         uint32_t code_stride) {
    if (nrows == 0) return VS_OK;
    const vs_index_desc& d = idx->d;
    uint32_t grid = nrows < 65536u * 16 ? nrows : 65536u * 16;
    hipLaunchKernelGGL(k_quantize_rows, dim3(grid), dim3(WAVE), 0, idx->ctx->stream, d_rows, row_stride, nrows,
                       d.dim_index, d.bits, idx->mean, idx->m2, count_as_f32(idx->count), d.words, code_stride, d_codes);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_hamming_gather(vs_index* idx, const uint64_t* d_qcodes, const uint32_t* d_ids, const uint32_t* d_off,
                          uint32_t nq, uint32_t* d_out) {
    if (nq == 0) return VS_OK;
    size_t lds = (size_t)idx->code_stride * 8;
    hipLaunchKernelGGL(k_hamming_gather, dim3(nq), dim3(WAVE), lds, idx->ctx->stream, idx->codes, idx->code_stride,
                       d_qcodes, d_ids, d_off, nq, d_out);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_rerank(vs_index* idx, const float* d_q_full, const uint32_t* d_ids, const uint32_t* d_off,
                  const uint32_t* d_cnt, uint32_t fixed_m, uint32_t nq, float* d_out, uint32_t row_base) {
    if (nq == 0) return VS_OK;
    VS_REQUIRE(idx->vecs != nullptr, "index has no vector column: rerank impossible");
    size_t lds = (size_t)idx->vec_stride * 4;
    hipLaunchKernelGGL(k_rerank, dim3(nq), dim3(256), lds, idx->ctx->stream, idx->vecs, idx->vec_stride,
                       idx->d.dim_full, idx->vnorm, idx->d.distance_type, d_q_full, d_ids, d_off, d_cnt, fixed_m, nq,
                       d_out, row_base);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_resort(vs_index* idx, uint32_t nq, uint32_t M, uint32_t rescore, uint32_t k, const uint32_t* d_stream_ids,
                  const uint32_t* d_cnt, const float* d_dist, uint64_t* d_heap_ws, uint32_t* d_out_ids,
                  uint64_t* d_out_tids, float* d_out_dist) {
    if (nq == 0) return VS_OK;
    // plain storage without truncation: the graph distance IS the full-precision distance (no resort, AM/scan.rs:392-399)
    const uint32_t* plain_keys = (idx->d.storage_type == VS_STORAGE_PLAIN && idx->d.dim_index == idx->d.dim_full)
                                     ? (const uint32_t*)idx->ws.stream_ham.p : nullptr;
    hipLaunchKernelGGL(k_resort, dim3((nq + 63) / 64), dim3(64), 0, idx->ctx->stream, nq, M, rescore, k, d_stream_ids,
                       d_cnt, d_dist, idx->tids, d_heap_ws, d_out_ids, d_out_tids, d_out_dist, plain_keys);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_resort_cursor(vs_index* idx, uint32_t n, bool exhausted, uint32_t rescore, uint32_t k, const uint32_t* d_stream,
                         const float* d_dist, const uint32_t* d_keys, uint64_t* d_heap, uint32_t* d_cur, uint32_t* d_out_ids,
                         uint64_t* d_out_tids, float* d_out_dist) {
    const uint32_t plain_keys = (idx->d.storage_type == VS_STORAGE_PLAIN && idx->d.dim_index == idx->d.dim_full) ? 1u : 0u;
    hipLaunchKernelGGL(k_resort_cursor, dim3(1), dim3(64), 0, idx->ctx->stream, n, exhausted ? 1u : 0u, rescore, k, d_stream, d_dist,
                       d_keys, plain_keys, idx->tids, d_heap, d_cur, d_out_ids, d_out_tids, d_out_dist);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// The new stream rows of one scan-pool round, moved from the round's staging rows to the slots' stream arrays: workgroup q copies
// cnt[q] rows of each of the three kinds (ids, Hamming keys, distances) from stage[kind][q][0..] to all[kind][q][off[q]..].  The row
// counts are read on the device, so the round needs no host round trip between its search launch and this one (round 6; until then:
// one hipMemcpy2DAsync per slot after the counts had come back).  A slot that did not run, failed or ended at once has cnt 0.
__global__ __launch_bounds__(WAVE) void k_pool_append(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, const uint32_t* __restrict__ stage,
                                                      uint32_t stage_kind_stride, uint32_t M, uint32_t* __restrict__ all, uint32_t all_kind_stride,
                                                      uint32_t rows_cap) {
    const uint32_t q = blockIdx.x;
    const uint32_t n = min(cnt[q], M), o = off[q];
    if (n == 0 || o >= rows_cap) return;
    const uint32_t m = min(n, rows_cap - o);
    for (uint32_t kind = 0; kind < 3; ++kind) {
        const uint32_t* src = stage + (size_t)kind * stage_kind_stride + (size_t)q * M;
        uint32_t* dst = all + (size_t)kind * all_kind_stride + (size_t)q * rows_cap + o;
        for (uint32_t i = threadIdx.x; i < m; i += WAVE) dst[i] = src[i];
    }
}
int launch_pool_append(vs_index* idx, uint32_t nq, const uint32_t* d_cnt, const uint32_t* d_off, const uint32_t* d_stage, uint32_t stage_kind_stride, uint32_t M,
                       uint32_t* d_all, uint32_t all_kind_stride, uint32_t rows_cap) {
    if (nq == 0) return VS_OK;
    hipLaunchKernelGGL(k_pool_append, dim3(nq), dim3(WAVE), 0, idx->ctx->stream, d_cnt, d_off, d_stage, stage_kind_stride, M, d_all, all_kind_stride, rows_cap);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// d_list: [n][3] (pool slot, rows emitted, exhausted) on the device; the per-scan arrays are rows of 2-D arrays (see the kernel)
int launch_resort_cursor_batch(vs_index* idx, uint32_t n, const uint32_t* d_list, uint32_t rescore, uint32_t k, const uint32_t* d_stream,
                               const float* d_dist, const uint32_t* d_keys, uint32_t row_stride, uint64_t* d_heap, uint32_t* d_cur,
                               uint32_t* d_out_ids, uint64_t* d_out_tids, float* d_out_dist, uint32_t out_stride) {
    if (n == 0) return VS_OK;
    const uint32_t plain_keys = (idx->d.storage_type == VS_STORAGE_PLAIN && idx->d.dim_index == idx->d.dim_full) ? 1u : 0u;
    const size_t lds = (size_t)rescore * 8 + ((size_t)rescore + k) * 4 + ((size_t)k + 4) * 4;
    VS_REQUIRE(lds <= 64 * 1024, "resort window of %u rows / %u rows per fetch does not fit LDS", rescore, k);
    hipLaunchKernelGGL(k_resort_cursor_batch, dim3(n), dim3(WAVE), lds, idx->ctx->stream, d_list, rescore, k, d_stream, d_dist, d_keys, row_stride,
                       plain_keys, idx->tids, d_heap, d_cur, d_out_ids, d_out_tids, d_out_dist, out_stride);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_row_norms(vs_index* idx) {
    if (!idx->vecs || idx->d.n == 0) return VS_OK;
    uint32_t blocks = (idx->d.n + 63) / 64;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_row_norms, dim3(blocks), dim3(WAVE), 0, idx->ctx->stream, idx->vecs, idx->vec_stride,
                       idx->d.dim_full, idx->d.n, idx->vnorm);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int launch_validate_nbrs(vs_index* idx, uint32_t* d_flag) {
    if (idx->d.n == 0) return VS_OK;
    hipLaunchKernelGGL(k_validate_nbrs, dim3((idx->d.n + 255) / 256), dim3(256), 0, idx->ctx->stream, idx->nbrs,
                       idx->nbr_stride, idx->d.num_neighbors, idx->d.n, d_flag);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
