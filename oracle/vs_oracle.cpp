/*
 * vs_oracle.cpp — CPU ORACLE (test infrastructure; see vs_oracle.h for the usage contract).
 *
 * A C++17 restatement of the StreamingDiskANN search path of timescale/pgvectorscale v0.9.0.
 * Every function cites the reference lines it follows; "AM/" = pgvectorscale/src/access_method/.
 * Build: see oracle/Makefile (-O3 -mavx2 -mfma -mpopcnt -ffp-contract=off: the reference is built
 * with +avx2,+fma, .cargo/config.toml:5-6; Rust never contracts a*b+c, hence -ffp-contract=off).
 */
#include "vs_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <thread>
#include <vector>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define VSO_AVX2 1
#else
#define VSO_AVX2 0
#endif


namespace {

/* ------------------------------------------------------------------------------------------------
 * f32::total_cmp (Rust core): compare the sign-magnitude-fixed bit patterns as i32.
 * Used by DistanceWithTieBreak::cmp (AM/graph/neighbor_with_distance.rs:74-83) and
 * ResortData::cmp (AM/scan.rs:111-117).
 * ---------------------------------------------------------------------------------------------- */
inline int32_t total_key(float f) {
    int32_t b;
    std::memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
inline int total_cmp(float a, float b) {
    int32_t ka = total_key(a), kb = total_key(b);
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

/* ------------------------------------------------------------------------------------------------
 * Rust std::collections::BinaryHeap<T> (max-heap by `le`), restated from the std source:
 *   push  = Vec::push + sift_up(0, old_len)
 *   pop   = Vec::pop; if non-empty swap with data[0]; sift_down_to_bottom(0)
 *   sift_up(start,pos): while pos>start { parent=(pos-1)/2; if elem <= data[parent] break; move hole up }
 *   sift_down_to_bottom(pos): while child <= end.saturating_sub(2) { child += (data[child] <= data[child+1]);
 *        move hole to child; child = 2*pos+1 }  if child == end-1 { move hole to child }  then sift_up(start,pos)
 * Tie order among equal keys depends on exactly these mechanics (call sites AM/graph/mod.rs:75,146,167;
 * AM/scan.rs:168,279-304).  No reference test pins it -> "parity unpinned" (oracle/README.md).
 * `Le(a,b)` must implement Rust's `a <= b` for the element type.
 * ---------------------------------------------------------------------------------------------- */
template <typename T, typename Le>
struct RustBinaryHeap {
    std::vector<T> data;
    Le le;
    bool empty() const { return data.empty(); }
    size_t size() const { return data.size(); }
    const T& peek() const { return data[0]; }
    void clear() { data.clear(); }
    void push(const T& item) {
        size_t old_len = data.size();
        data.push_back(item);
        sift_up(0, old_len);
    }
    T pop() {
        T item = data.back();
        data.pop_back();
        if (!data.empty()) {
            std::swap(item, data[0]);
            sift_down_to_bottom(0);
        }
        return item;
    }
    size_t sift_up(size_t start, size_t pos) {
        T elem = data[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(elem, data[parent])) break;
            data[pos] = data[parent];
            pos = parent;
        }
        data[pos] = elem;
        return pos;
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = data.size();
        size_t start = pos;
        T elem = data[pos];
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
        while (child <= lim) {
            child += le(data[child], data[child + 1]) ? 1 : 0;
            data[pos] = data[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            data[pos] = data[child];
            pos = child;
        }
        data[pos] = elem;
        sift_up(start, pos);
    }
};

/* ListSearchNeighbor ordered by DistanceWithTieBreak (AM/graph/mod.rs:22-48).  For query distances the
 * tie-break is the constant 0 (with_query, AM/graph/neighbor_with_distance.rs:31-43), so cmp is:
 * both == 0.0 -> Equal, else total_cmp. */
struct LSN {
    uint32_t id;
    float dist;
};
inline int lsn_cmp(const LSN& a, const LSN& b) {
    if (a.dist == 0.0f && b.dist == 0.0f) return 0;
    return total_cmp(a.dist, b.dist);
}
/* BinaryHeap<Reverse<LSN>>: Reverse(a) <= Reverse(b)  <=>  b <= a */
struct ReverseLsnLe {
    bool operator()(const LSN& a, const LSN& b) const { return lsn_cmp(b, a) <= 0; }
};
/* ResortData: cmp(self, other) = other.distance.total_cmp(self.distance)  (AM/scan.rs:111-117) */
struct Resort {
    uint64_t heap_tid;
    uint32_t node;
    float distance;
};
struct ResortLe {
    bool operator()(const Resort& a, const Resort& b) const { return total_cmp(b.distance, a.distance) <= 0; }
};

/* HashSet<ItemPointer> `inserted` (AM/graph/mod.rs:77,126-128): membership only, so any set will do. */
struct U32Set {
    std::vector<uint32_t> slots;
    size_t count = 0;
    explicit U32Set(size_t cap_hint = 64) {
        size_t c = 64;
        while (c < cap_hint * 2) c <<= 1;
        slots.assign(c, 0xFFFFFFFFu);
    }
    static inline uint32_t hash(uint32_t x) {
        x ^= x >> 16;
        x *= 0x7feb352dU;
        x ^= x >> 15;
        x *= 0x846ca68bU;
        x ^= x >> 16;
        return x;
    }
    void grow() {
        std::vector<uint32_t> old;
        old.swap(slots);
        slots.assign(old.size() * 2, 0xFFFFFFFFu);
        count = 0;
        for (uint32_t v : old)
            if (v != 0xFFFFFFFFu) insert(v);
    }
    /* returns true if newly inserted (HashSet::insert semantics) */
    bool insert(uint32_t key) {
        if ((count + 1) * 2 > slots.size()) grow();
        size_t mask = slots.size() - 1;
        size_t i = hash(key) & mask;
        while (true) {
            uint32_t v = slots[i];
            if (v == key) return false;
            if (v == 0xFFFFFFFFu) {
                slots[i] = key;
                ++count;
                return true;
            }
            i = (i + 1) & mask;
        }
    }
};

inline float hadd8(const float* a) {
    /* simdeez 1.0.8 Avx2::horizontal_add_ps (restated from memory; see header):
     * lo+hi halves, then movehdup+add, movehl+add_ss:  ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)) */
    float s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
    float t0 = s0 + s1;
    float t1 = s2 + s3;
    return t0 + t1;
}

}  // namespace

extern "C" {

/* AM/distance/mod.rs:255-323 — every match arm computes the same sum over a.len() words */
uint64_t vso_distance_xor(const uint64_t* a, const uint64_t* b, size_t words) {
    uint64_t s = 0;
    for (size_t i = 0; i < words; ++i) s += (uint64_t)__builtin_popcountll(a[i] ^ b[i]);
    return s;
}

/* AM/distance/mod.rs:325-377 with S = Avx2 (VF32_WIDTH = 8), distance_x86.rs:21-25.
 * 4 accumulators x 8 lanes; per 32 floats acc_j += (x-y)*(x-y) (separate mul, add — no FMA);
 * dist = hadd(acc0)+hadd(acc1)+hadd(acc2)+hadd(acc3) left to right; scalar tail; no sqrt. */
float vso_distance_l2(const float* x, const float* y, size_t n) {
    float acc[4][8];
    for (int j = 0; j < 4; ++j)
        for (int l = 0; l < 8; ++l) acc[j][l] = 0.0f;
    size_t i = 0;
    while (n - i >= 32) {
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) {
                float d = x[i + 8 * j + l] - y[i + 8 * j + l];
                float p = d * d;
                acc[j][l] = acc[j][l] + p;
            }
        i += 32;
    }
    float dist = hadd8(acc[0]) + hadd8(acc[1]);
    dist = dist + hadd8(acc[2]);
    dist = dist + hadd8(acc[3]);
    for (; i < n; ++i) {
        float diff = x[i] - y[i];
        float p = diff * diff;
        dist += p;
    }
    return dist;
}

/* AM/distance/mod.rs:380-435 with S = Avx2: acc_j = fmadd(x, y, acc_j); tail `dist += x*y` (mul then add). */
float vso_inner_product(const float* x, const float* y, size_t n) {
    float acc[4][8];
    for (int j = 0; j < 4; ++j)
        for (int l = 0; l < 8; ++l) acc[j][l] = 0.0f;
    size_t i = 0;
    while (n - i >= 32) {
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) acc[j][l] = __builtin_fmaf(x[i + 8 * j + l], y[i + 8 * j + l], acc[j][l]);
        i += 32;
    }
    float dist = hadd8(acc[0]) + hadd8(acc[1]);
    dist = dist + hadd8(acc[2]);
    dist = dist + hadd8(acc[3]);
    for (; i < n; ++i) {
        float p = x[i] * y[i];
        dist += p;
    }
    return dist;
}

float vso_distance_inner_product(const float* a, const float* b, size_t n) { return -vso_inner_product(a, b, n); }

/* distance_x86.rs:34-36: (1.0 - dot).max(0.0); Rust f32::max returns the non-NaN operand */
float vso_distance_cosine(const float* a, const float* b, size_t n) {
    float r = 1.0f - vso_inner_product(a, b, n);
    return std::fmax(r, 0.0f);
}

/* AM/distance/mod.rs:106-117 — sequential f32 sum */
float vso_distance_l2_unoptimized(const float* a, const float* b, size_t n) {
    float norm = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        norm += p;
    }
    return norm;
}
float vso_inner_product_unoptimized(const float* a, const float* b, size_t n) {
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        s += p;
    }
    return s;
}
float vso_distance_cosine_unoptimized(const float* a, const float* b, size_t n) {
    float r = 1.0f - vso_inner_product_unoptimized(a, b, n);
    return std::fmax(r, 0.0f);
}

int vso_have_avx2(void) { return VSO_AVX2; }

/* Micro timings in the style of the reference's criterion benches (benches/distance.rs:144-161: D = 2000 f32 ramps
 * v+1000.1 / v+2000.2 through distance_l2 / distance_cosine / inner product; :299-338: 1536-bit patterns i%2==0 vs i%3==0
 * through distance_xor_optimized).  which: 0 = l2, 1 = cosine, 2 = inner product, 3 = xor.  Returns ns per call. */
double vso_micro_bench(int which, uint64_t iters) {
    std::vector<float> r(2000), l(2000);
    for (int v = 0; v < 2000; ++v) {
        r[v] = (float)v + 1000.1f;
        l[v] = (float)v + 2000.2f;
    }
    uint64_t a[24] = {0}, b[24] = {0};
    for (int i = 0; i < 1536; ++i) {
        if (i % 2 == 0) a[i / 64] |= 1ull << (i % 64);
        if (i % 3 == 0) b[i / 64] |= 1ull << (i % 64);
    }
    const float* rp = r.data();
    const float* lp = l.data();
    const uint64_t *ap = a, *bp = b;
    double sink = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < iters; ++it) {
        __asm__ volatile("" : "+r"(rp), "+r"(lp), "+r"(ap), "+r"(bp) : : "memory");  // black_box
        switch (which) {
            case 0: sink += vso_distance_l2(rp, lp, 2000); break;
            case 1: sink += vso_distance_cosine(rp, lp, 2000); break;
            case 2: sink += vso_inner_product(rp, lp, 2000); break;
            default: sink += (double)vso_distance_xor(ap, bp, 24); break;
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    __asm__ volatile("" : : "r"(&sink) : "memory");
    return std::chrono::duration<double, std::nano>(t1 - t0).count() / (double)(iters ? iters : 1);
}

#if VSO_AVX2
static inline float hadd_ps_avx2(__m256 a) {
    __m128 vlow = _mm256_castps256_ps128(a);
    __m128 vhigh = _mm256_extractf128_ps(a, 1);
    vlow = _mm_add_ps(vlow, vhigh);
    __m128 shuf = _mm_movehdup_ps(vlow);
    __m128 sums = _mm_add_ps(vlow, shuf);
    shuf = _mm_movehl_ps(shuf, sums);
    sums = _mm_add_ss(sums, shuf);
    return _mm_cvtss_f32(sums);
}
float vso_distance_l2_avx2(const float* x, const float* y, size_t n) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    size_t i = 0;
    while (n - i >= 32) {
        __m256 d0 = _mm256_sub_ps(_mm256_loadu_ps(x + i), _mm256_loadu_ps(y + i));
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(x + i + 8), _mm256_loadu_ps(y + i + 8));
        __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(x + i + 16), _mm256_loadu_ps(y + i + 16));
        __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(x + i + 24), _mm256_loadu_ps(y + i + 24));
        a0 = _mm256_add_ps(a0, _mm256_mul_ps(d0, d0));
        a1 = _mm256_add_ps(a1, _mm256_mul_ps(d1, d1));
        a2 = _mm256_add_ps(a2, _mm256_mul_ps(d2, d2));
        a3 = _mm256_add_ps(a3, _mm256_mul_ps(d3, d3));
        i += 32;
    }
    float dist = hadd_ps_avx2(a0) + hadd_ps_avx2(a1);
    dist = dist + hadd_ps_avx2(a2);
    dist = dist + hadd_ps_avx2(a3);
    for (; i < n; ++i) {
        float diff = x[i] - y[i];
        float p = diff * diff;
        dist += p;
    }
    return dist;
}
float vso_inner_product_avx2(const float* x, const float* y, size_t n) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    size_t i = 0;
    while (n - i >= 32) {
        a0 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i), _mm256_loadu_ps(y + i), a0);
        a1 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 8), _mm256_loadu_ps(y + i + 8), a1);
        a2 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 16), _mm256_loadu_ps(y + i + 16), a2);
        a3 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 24), _mm256_loadu_ps(y + i + 24), a3);
        i += 32;
    }
    float dist = hadd_ps_avx2(a0) + hadd_ps_avx2(a1);
    dist = dist + hadd_ps_avx2(a2);
    dist = dist + hadd_ps_avx2(a3);
    for (; i < n; ++i) {
        float p = x[i] * y[i];
        dist += p;
    }
    return dist;
}
#else
float vso_distance_l2_avx2(const float*, const float*, size_t) { return std::numeric_limits<float>::quiet_NaN(); }
float vso_inner_product_avx2(const float*, const float*, size_t) { return std::numeric_limits<float>::quiet_NaN(); }
#endif

/* AM/distance/mod.rs:225-253 */
int vso_preprocess_cosine(float* v, size_t n) {
    float norm = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = v[i] * v[i];
        norm += p;
    }
    const float eps = std::numeric_limits<float>::epsilon();
    float adj_epsilon = eps * (float)n;
    if (norm < eps) return 0;
    if (norm >= 1.0f - adj_epsilon && norm <= 1.0f + adj_epsilon) return 0;
    float s = std::sqrt(norm);
    for (size_t i = 0; i < n; ++i) v[i] /= s;
    return 1;
}

float vso_distance_by_type(int t, const float* a, const float* b, size_t n) {
    switch (t) { /* DistanceType::get_distance_function, AM/distance/mod.rs:44-50 */
        case VSO_COSINE: return vso_distance_cosine(a, b, n);
        case VSO_L2: return vso_distance_l2(a, b, n);
        default: return vso_distance_inner_product(a, b, n);
    }
}

/* ---- SBQ quantizer ------------------------------------------------------------------------------ */
size_t vso_quantized_size(size_t dims, unsigned bits) { /* AM/sbq/quantize.rs:37-45 */
    size_t num_bits = dims * bits;
    return (num_bits % 64 == 0) ? num_bits / 64 : num_bits / 64 + 1;
}

unsigned vso_default_bits(size_t dims_to_index) { return dims_to_index < 900 ? 2 : 1; } /* AM/meta_page.rs:312-323 */

void vso_quantize(const float* mean, const float* m2, uint64_t count, unsigned bits, const float* v, size_t dims,
                  uint64_t* out) { /* AM/sbq/quantize.rs:52-89 (use_mean is always true, :18) */
    size_t w = vso_quantized_size(dims, bits);
    for (size_t i = 0; i < w; ++i) out[i] = 0;
    if (bits == 1) {
        for (size_t i = 0; i < dims; ++i)
            if (v[i] > mean[i]) out[i / 64] |= (uint64_t)1 << (i % 64);
    } else {
        for (size_t i = 0; i < dims; ++i) {
            float mu = mean[i];
            float variance = m2[i] / (float)count;
            float std_dev = std::sqrt(variance);
            unsigned ranges = bits + 1;
            float v_z_score = (v[i] - mu) / std_dev;
            float index = (v_z_score + 2.0f) / (4.0f / (float)ranges);
            size_t bit_position = i * bits;
            if (index < 1.0f) {
                /* all zeros */
            } else {
                /* `index.floor() as usize`: saturating cast, NaN -> 0 */
                float fl = std::floor(index);
                size_t ones;
                if (std::isnan(fl)) ones = 0;
                else if (fl <= 0.0f) ones = 0;
                else if (fl >= 1.8e19f) ones = std::numeric_limits<size_t>::max();
                else ones = (size_t)fl;
                if (ones > bits) ones = bits;
                for (size_t j = 0; j < ones; ++j) out[(bit_position + j) / 64] |= (uint64_t)1 << ((bit_position + j) % 64);
            }
        }
    }
}

void vso_train(float* mean, float* m2, uint64_t* count, unsigned bits, const float* rows, size_t n, size_t dims) {
    /* AM/sbq/quantize.rs:115-148 (Welford, all in f32, count cast `as f32` each step) */
    for (size_t r = 0; r < n; ++r) {
        const float* s = rows + r * dims;
        *count += 1;
        float c = (float)*count;
        if (bits > 1) {
            for (size_t i = 0; i < dims; ++i) {
                float delta = s[i] - mean[i];
                mean[i] += (s[i] - mean[i]) / c;
                float delta2 = s[i] - mean[i];
                float p = delta * delta2;
                m2[i] += p;
            }
        } else {
            for (size_t i = 0; i < dims; ++i) mean[i] += (s[i] - mean[i]) / c;
        }
    }
}

/* ---- labels --------------------------------------------------------------------------------------- */
size_t vso_labelset_from(int16_t* labels, size_t n) { /* AM/labels/mod.rs:30-37 */
    std::sort(labels, labels + n);
    return (size_t)(std::unique(labels, labels + n) - labels);
}

int vso_labels_overlap(const int16_t* a, size_t na, const int16_t* b, size_t nb) { /* AM/labels/mod.rs:124-142 */
    size_t i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) return 1;
        if (a[i] < b[j]) ++i;
        else ++j;
    }
    return 0;
}

int vso_labels_contains_intersection(const int16_t* c, size_t nc, const int16_t* a, size_t na, const int16_t* b,
                                     size_t nb) { /* AM/labels/mod.rs:85-111 */
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) {
            while (k < nc && c[k] < a[i]) ++k;
            if (k == nc || c[k] > a[i]) return 0;
            ++i;
            ++j;
        } else if (a[i] < b[j]) ++i;
        else ++j;
    }
    return 1;
}

int vso_smallint_array_overlap(const int16_t* l, const uint8_t* ln, size_t nl, const int16_t* r, const uint8_t* rn,
                               size_t nr) { /* AM/mod.rs:283-314: both branches = "any non-NULL element in common" */
    if (nl == 0 || nr == 0) return 0;
    for (size_t i = 0; i < nl; ++i) {
        if (ln && ln[i]) continue;
        for (size_t j = 0; j < nr; ++j) {
            if (rn && rn[j]) continue;
            if (l[i] == r[j]) return 1;
        }
    }
    return 0;
}

}  // extern "C"

/* ==================================================================================================
 * Scan state: TSVScanState / TSVResponseIterator / ListSearchResult
 * ================================================================================================ */
struct vso_scan {
    const vso_index* idx;
    std::vector<float> q_full, q_index; /* PgVector full / index slices, AM/pg_vector.rs:162-199 */
    bool labels_some = false;           /* LabeledVector.labels is Some (AM/labels/mod.rs:222-236) */
    std::vector<int16_t> qlabels;
    bool has_label_filter = false; /* AM/scan.rs:189 */
    bool xs_recheck = false;
    std::vector<uint64_t> qcode; /* SbqSearchDistanceMeasure.vec, AM/sbq/mod.rs:145-148 */
    /* ListSearchResult, AM/graph/mod.rs:74-82 */
    RustBinaryHeap<LSN, ReverseLsnLe> candidates;
    std::vector<LSN> visited;
    U32Set inserted;
    uint32_t search_list_size;
    /* resort window, AM/scan.rs:162-174 */
    uint32_t resort_size;
    RustBinaryHeap<Resort, ResortLe> resort_buffer;
    vso_stats st{};
    std::vector<float> scratch; /* heap vector copy (pg_detoast_datum_copy, AM/pg_vector.rs:133) */

    explicit vso_scan(size_t hint) : inserted(hint) {}

    const uint64_t* code(uint32_t id) const { return idx->codes + (size_t)id * idx->words; }
    const int16_t* node_labels(uint32_t id, size_t* n) const {
        uint32_t a = idx->label_off[id], b = idx->label_off[id + 1];
        *n = b - a;
        return idx->label_val + a;
    }
    float bq_distance(uint32_t id) { /* AM/sbq/mod.rs:150-158 */
        if (idx->storage_plain) return plain_distance(id);
        st.quantized_distance_comparisons++;
        return (float)vso_distance_xor(code(id), qcode.data(), idx->words);
    }
    /* PlainDistanceMeasure::calculate_distance(distance_fn, query.to_index_slice(), node.vector) (AM/plain/storage.rs:239-247,
     * 273-281).  PlainNode.vector is the index slice of the inserted vector, cosine-normalised at insert time
     * (PgVector::from_datum, AM/pg_vector.rs:143-157); the flat arrays keep the raw heap column, so it is re-derived here. */
    float plain_distance(uint32_t id) {
        st.full_distance_comparisons++;
        const float* hv = idx->vecs + (size_t)id * idx->dim_full;
        const float* v = hv;
        if (idx->distance_type == VSO_COSINE) {
            scratch.assign(hv, hv + idx->dim_index);
            vso_preprocess_cosine(scratch.data(), idx->dim_index);
            v = scratch.data();
        }
        return vso_distance_by_type((int)idx->distance_type, q_index.data(), v, idx->dim_index);
    }
    void insert_neighbor(const LSN& n) { /* AM/graph/mod.rs:144-147 */
        st.candidate_nodes++;
        candidates.push(n);
    }
    /* AM/sbq/storage.rs:365-391 */
    void create_lsn_for_start_node(uint32_t id) {
        if (!inserted.insert(id)) return;
        st.node_reads++;
        LSN n{id, bq_distance(id)};
        insert_neighbor(n);
    }
    /* AM/graph/mod.rs:153-170 */
    bool visit_closest(size_t pos_limit, size_t* out_idx) {
        if (candidates.empty()) return false;
        if (visited.size() > pos_limit) {
            const LSN& node_at_pos = visited[pos_limit - 1];
            const LSN& head = candidates.peek();
            if (lsn_cmp(head, node_at_pos) >= 0) return false; /* head.0 >= *node_at_pos */
        }
        LSN head = candidates.pop();
        /* partition_point(|x| *x < head.0): first index whose element is NOT < head */
        size_t lo = 0, hi = visited.size();
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (lsn_cmp(visited[mid], head) < 0) lo = mid + 1;
            else hi = mid;
        }
        visited.insert(visited.begin() + (ptrdiff_t)lo, head);
        *out_idx = lo;
        return true;
    }
    /* AM/sbq/storage.rs:135-190 (GraphNeighborStore::Disk arm) */
    void visit_lsn(size_t lsn_idx, bool no_filter) {
        /* (plain storage: assert!(no_filter, "Plain storage does not support label filters"), AM/plain/storage.rs:262 —
         * vso_scan_begin refuses label keys on a plain index) */
        uint32_t visiting = visited[lsn_idx].id;
        st.node_reads++;
        const uint32_t* nb = idx->nbrs + (size_t)visiting * idx->nbr_stride;
        for (uint32_t s = 0; s < idx->num_neighbors; ++s) {
            uint32_t nid = nb[s];
            if (nid == VSO_INVALID_NODE) break; /* AM/sbq/node.rs:260-285 */
            if (!inserted.insert(nid)) continue; /* prepare_insert — marks BEFORE the label check */
            st.node_reads++;
            if (labels_some) {
                if (!no_filter) {
                    size_t nl;
                    const int16_t* l = node_labels(nid, &nl);
                    if (!vso_labels_overlap(qlabels.data(), qlabels.size(), l, nl)) continue;
                }
            }
            LSN n{nid, bq_distance(nid)};
            insert_neighbor(n);
        }
    }
    /* AM/graph/mod.rs:357-385 */
    void greedy_search_iterate(size_t visit_n_closest, bool no_filter) {
        size_t i;
        while (visit_closest(visit_n_closest, &i)) {
            st.visited_nodes++;
            visit_lsn(i, no_filter);
        }
    }
    /* AM/graph/mod.rs:174-184 + return_lsn AM/sbq/storage.rs:404-414 */
    bool consume(uint64_t* heap_tid, LSN* lsn) {
        if (visited.empty()) return false;
        *lsn = visited.front();
        visited.erase(visited.begin());
        st.node_reads++;
        *heap_tid = idx->heap_tids[lsn->id];
        return true;
    }
    /* AM/scan.rs:210-242 */
    bool next(uint64_t* heap_tid, LSN* lsn) {
        st.next_calls++;
        while (true) {
            greedy_search_iterate(search_list_size, !has_label_filter);
            if (!consume(heap_tid, lsn)) return false;
            if ((*heap_tid & 0xFFFFu) == 0) continue; /* InvalidOffsetNumber: deleted tuple */
            return true;
        }
    }
    /* AM/sbq/storage.rs:304-328 + AM/pg_vector.rs:125-157 (heap vector copied, cosine-normalised) */
    float full_distance(uint32_t node) {
        st.node_heap_reads++;
        st.full_distance_comparisons++;
        const float* hv = idx->vecs + (size_t)node * idx->dim_full;
        const float* v = hv;
        if (idx->distance_type == VSO_COSINE) {
            scratch.assign(hv, hv + idx->dim_full);
            vso_preprocess_cosine(scratch.data(), idx->dim_full);
            v = scratch.data();
        }
        return vso_distance_by_type((int)idx->distance_type, v, q_full.data(), idx->dim_full);
    }
    /* AM/scan.rs:244-305 */
    bool next_with_resort(uint32_t* node, uint64_t* heap_tid, float* dist) {
        st.next_calls_with_resort++;
        if (resort_size == 0) { /* BinaryHeap::with_capacity(0).capacity() == 0 */
            LSN l;
            if (!next(heap_tid, &l)) return false;
            *node = l.id;
            *dist = std::numeric_limits<float>::quiet_NaN();
            return true;
        }
        while (resort_buffer.size() < resort_size) {
            LSN l;
            uint64_t tid;
            if (!next(&tid, &l)) break;
            if (idx->visible && !idx->visible[l.id]) {
                /* TableSlot::from_index_heap_pointer found no tuple this snapshot can see: the heap read is recorded
                 * (UT/table_slot.rs:45), the candidate never enters the window (AM/scan.rs:268-272) */
                st.node_heap_reads++;
                st.full_distance_comparisons++; /* AM/scan.rs:258: counted before the fetch */
                continue;
            }
            float d = full_distance(l.id);
            resort_buffer.push(Resort{tid, l.id, d});
        }
        if (resort_buffer.empty()) return false;
        Resort r = resort_buffer.pop();
        *node = r.node;
        *heap_tid = r.heap_tid;
        *dist = r.distance;
        return true;
    }
};

extern "C" {

vso_scan* vso_scan_begin(const vso_index* idx, const float* query, const int16_t* labels, size_t n_labels,
                         int has_label_key, uint32_t search_list_size, uint32_t rescore) {
    vso_scan* s = new vso_scan((size_t)search_list_size * idx->num_neighbors);
    s->idx = idx;
    s->search_list_size = search_list_size;
    s->resort_size = rescore;
    s->xs_recheck = has_label_key != 0; /* AM/scan.rs:350-352 */
    /* LabeledVector::from_scan_key_data, AM/labels/mod.rs:209-238 */
    if (query == nullptr) {
        s->q_full.assign(idx->dim_full, 0.0f); /* PgVector::zeros */
        s->q_index.assign(idx->dim_index, 0.0f);
        s->labels_some = false;
    } else {
        /* PgVector::from_datum(index=true, full=true), AM/pg_vector.rs:162-199 */
        s->q_full.assign(query, query + idx->dim_full);
        if (idx->dim_full == idx->dim_index) {
            if (idx->distance_type == VSO_COSINE) vso_preprocess_cosine(s->q_full.data(), idx->dim_full);
            s->q_index = s->q_full;
        } else {
            s->q_index.assign(query, query + idx->dim_index);
            if (idx->distance_type == VSO_COSINE) {
                vso_preprocess_cosine(s->q_index.data(), idx->dim_index);
                vso_preprocess_cosine(s->q_full.data(), idx->dim_full);
            }
        }
        if (has_label_key) {
            s->labels_some = true;
            s->qlabels.assign(labels, labels + n_labels);
            s->qlabels.resize(vso_labelset_from(s->qlabels.data(), s->qlabels.size()));
        }
    }
    s->has_label_filter = s->labels_some && !s->qlabels.empty(); /* AM/scan.rs:189 */

    /* Graph::greedy_search_streaming_init, AM/graph/mod.rs:331-354 */
    if (idx->default_start == VSO_INVALID_NODE || idx->n == 0) return s; /* ListSearchResult::empty() */
    /* StartNodes::get_for_node, AM/graph/start_nodes.rs:39-48 */
    std::vector<uint32_t> starts;
    if (s->labels_some) {
        for (int16_t l : s->qlabels) {
            const int16_t* b = idx->label_start_labels;
            const int16_t* e = b + idx->n_label_starts;
            const int16_t* p = std::lower_bound(b, e, l);
            if (p != e && *p == l) starts.push_back(idx->label_start_nodes[p - b]);
        }
    } else {
        starts.push_back(idx->default_start);
    }
    /* SbqSearchDistanceMeasure::new, AM/sbq/mod.rs:145-148 (plain storage: PlainDistanceMeasure::Full(query)) */
    if (!idx->storage_plain) {
        s->qcode.resize(idx->words);
        vso_quantize(idx->mean, idx->m2, idx->count, idx->bits, s->q_index.data(), idx->dim_index, s->qcode.data());
    }
    /* ListSearchResult::new, AM/graph/mod.rs:97-124 */
    s->st.calls++;
    s->candidates.data.reserve((size_t)search_list_size * idx->num_neighbors);
    s->visited.reserve((size_t)search_list_size * 2);
    for (uint32_t sn : starts) s->create_lsn_for_start_node(sn);
    return s;
}

int vso_scan_gettuple(vso_scan* s, uint32_t* node, uint64_t* heap_tid, float* dist) {
    uint32_t n = VSO_INVALID_NODE;
    uint64_t t = 0;
    float d = std::numeric_limits<float>::quiet_NaN();
    bool ok;
    if (s->idx->storage_plain && s->idx->dim_full == s->idx->dim_index) {
        /* amgettuple, Plain arm without truncation: "no need to resort" -> iter.next (AM/scan.rs:392-399).  The distance
         * handed back here is the graph distance (already full precision); the reference returns none to the executor. */
        LSN l{VSO_INVALID_NODE, 0.0f};
        ok = s->next(&t, &l);
        if (ok) {
            n = l.id;
            d = l.dist;
        }
    } else {
        ok = s->next_with_resort(&n, &t, &d);
    }
    if (node) *node = n;
    if (heap_tid) *heap_tid = t;
    if (dist) *dist = d;
    return ok ? 1 : 0;
}

int vso_scan_next_sbq(vso_scan* s, uint32_t* node, uint64_t* heap_tid, uint32_t* ham) {
    LSN l{VSO_INVALID_NODE, 0.0f};
    uint64_t t = 0;
    bool ok = s->next(&t, &l);
    if (node) *node = ok ? l.id : VSO_INVALID_NODE;
    if (heap_tid) *heap_tid = t;
    if (ham) {
        if (!ok) *ham = 0xFFFFFFFFu;
        else if (s->idx->storage_plain) std::memcpy(ham, &l.dist, 4); /* the f32 graph distance, bit for bit */
        else *ham = (uint32_t)l.dist;
    }
    return ok ? 1 : 0;
}

int vso_scan_xs_recheck(const vso_scan* s) { return s->xs_recheck ? 1 : 0; }
void vso_scan_stats(const vso_scan* s, vso_stats* out) { *out = s->st; }
void vso_scan_end(vso_scan* s) { delete s; }

}  // extern "C"

static void stats_add(vso_stats* a, const vso_stats& b) {
    a->calls += b.calls;
    a->node_reads += b.node_reads;
    a->node_heap_reads += b.node_heap_reads;
    a->quantized_distance_comparisons += b.quantized_distance_comparisons;
    a->full_distance_comparisons += b.full_distance_comparisons;
    a->visited_nodes += b.visited_nodes;
    a->candidate_nodes += b.candidate_nodes;
    a->next_calls += b.next_calls;
    a->next_calls_with_resort += b.next_calls_with_resort;
}

template <typename F>
static void parallel_queries(uint32_t nq, uint32_t n_threads, F&& fn) {
    if (n_threads <= 1 || nq <= 1) {
        for (uint32_t q = 0; q < nq; ++q) fn(q, 0u);
        return;
    }
    std::atomic<uint32_t> nextq{0};
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            while (true) {
                uint32_t q = nextq.fetch_add(1);
                if (q >= nq) break;
                fn(q, t);
            }
        });
    for (auto& x : th) x.join();
}

extern "C" {

void vso_search_batch(const vso_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                      uint32_t nq, uint32_t L, uint32_t rescore, uint32_t k, uint32_t n_threads, uint32_t* out_nodes,
                      float* out_dist, vso_stats* stats_sum) {
    std::vector<vso_stats> per(std::max(1u, n_threads));
    for (auto& p : per) std::memset(&p, 0, sizeof(p));
    parallel_queries(nq, n_threads, [&](uint32_t q, uint32_t t) {
        const int16_t* ql = nullptr;
        size_t nl = 0;
        if (qlabel_off) {
            ql = qlabels + qlabel_off[q];
            nl = qlabel_off[q + 1] - qlabel_off[q];
        }
        vso_scan* s = vso_scan_begin(idx, queries + (size_t)q * idx->dim_full, ql, nl, qlabel_off != nullptr, L, rescore);
        for (uint32_t i = 0; i < k; ++i) {
            uint32_t node;
            uint64_t tid;
            float d;
            int ok = vso_scan_gettuple(s, &node, &tid, &d);
            out_nodes[(size_t)q * k + i] = ok ? node : VSO_INVALID_NODE;
            if (out_dist) out_dist[(size_t)q * k + i] = ok ? d : std::numeric_limits<float>::quiet_NaN();
            if (!ok) {
                for (uint32_t j = i + 1; j < k; ++j) {
                    out_nodes[(size_t)q * k + j] = VSO_INVALID_NODE;
                    if (out_dist) out_dist[(size_t)q * k + j] = std::numeric_limits<float>::quiet_NaN();
                }
                break;
            }
        }
        stats_add(&per[t], s->st);
        vso_scan_end(s);
    });
    if (stats_sum) {
        std::memset(stats_sum, 0, sizeof(*stats_sum));
        for (auto& p : per) stats_add(stats_sum, p);
    }
}

void vso_stream_batch(const vso_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                      uint32_t nq, uint32_t L, uint32_t m, uint32_t n_threads, uint32_t* out_nodes, uint32_t* out_ham,
                      vso_stats* stats_sum) {
    std::vector<vso_stats> per(std::max(1u, n_threads));
    for (auto& p : per) std::memset(&p, 0, sizeof(p));
    parallel_queries(nq, n_threads, [&](uint32_t q, uint32_t t) {
        const int16_t* ql = nullptr;
        size_t nl = 0;
        if (qlabel_off) {
            ql = qlabels + qlabel_off[q];
            nl = qlabel_off[q + 1] - qlabel_off[q];
        }
        vso_scan* s = vso_scan_begin(idx, queries + (size_t)q * idx->dim_full, ql, nl, qlabel_off != nullptr, L, 0);
        bool done = false;
        for (uint32_t i = 0; i < m; ++i) {
            uint32_t node = VSO_INVALID_NODE, ham = 0xFFFFFFFFu;
            uint64_t tid;
            if (!done && !vso_scan_next_sbq(s, &node, &tid, &ham)) done = true;
            out_nodes[(size_t)q * m + i] = done ? VSO_INVALID_NODE : node;
            if (out_ham) out_ham[(size_t)q * m + i] = done ? 0xFFFFFFFFu : ham;
        }
        stats_add(&per[t], s->st);
        vso_scan_end(s);
    });
    if (stats_sum) {
        std::memset(stats_sum, 0, sizeof(*stats_sum));
        for (auto& p : per) stats_add(stats_sum, p);
    }
}

}  // extern "C"

/* ==================================================================================================
 * Test-graph manufacture (NOT the reference search path).  Sequential Vamana with SBQ Hamming distances,
 * modelled on Graph::insert (AM/graph/mod.rs:637-717), add_neighbors (:212-266), prune_neighbors (:392-488).
 * Ties are broken deterministically by (distance, node id).
 * ================================================================================================ */
namespace {
struct Cand {
    uint32_t id;
    uint32_t d;
};
inline bool cand_lt(const Cand& a, const Cand& b) { return a.d != b.d ? a.d < b.d : a.id < b.id; }

struct Builder {
    uint32_t n, w, R, stride, L;
    double max_alpha;
    const uint64_t* codes;
    uint32_t* nbrs;
    /* labeled vector sets (nullptr: none): CSR label sets; per-label start nodes are assigned as nodes arrive */
    const uint32_t* loff = nullptr;
    const int16_t* lval = nullptr;
    std::map<int16_t, uint32_t> label_start;
    const int16_t* lab(uint32_t i) const { return lval + loff[i]; }
    size_t nlab(uint32_t i) const { return loff[i + 1] - loff[i]; }
    const uint64_t* code(uint32_t i) const { return codes + (size_t)i * w; }
    uint32_t ham(uint32_t a, uint32_t b) const { return (uint32_t)vso_distance_xor(code(a), code(b), w); }
    uint32_t degree(uint32_t i) const {
        const uint32_t* r = nbrs + (size_t)i * stride;
        uint32_t d = 0;
        while (d < R && r[d] != VSO_INVALID_NODE) ++d;
        return d;
    }
    void set_neighbors(uint32_t i, const std::vector<Cand>& l) {
        uint32_t* r = nbrs + (size_t)i * stride;
        for (uint32_t s = 0; s < stride; ++s) r[s] = s < l.size() ? l[s].id : VSO_INVALID_NODE;
    }
    /* prune_neighbors with alpha ladder 1.0, 1.2, ... <= max_alpha; distance factor as get_factor (ratio, 0-safe) */
    std::vector<Cand> prune(std::vector<Cand> cands, uint32_t of) {
        std::sort(cands.begin(), cands.end(), cand_lt);
        std::vector<Cand> results;
        std::vector<double> max_factors(cands.size(), 0.0);
        double alpha = 1.0;
        while (alpha <= max_alpha && results.size() < R) {
            for (size_t i = 0; i < cands.size(); ++i) {
                if (results.size() >= R) return results;
                if (max_factors[i] > alpha) continue;
                max_factors[i] = std::numeric_limits<double>::max();
                results.push_back(cands[i]);
                for (size_t j = i + 1; j < cands.size(); ++j) {
                    if (max_factors[j] > max_alpha) continue;
                    /* "Does it contain essential labels?"  A candidate is only occluded by an existing neighbor that carries
                     * every label the candidate shares with the point (AM/graph/mod.rs:442-456) */
                    if (loff && !vso_labels_contains_intersection(lab(cands[i].id), nlab(cands[i].id), lab(cands[j].id),
                                                                  nlab(cands[j].id), lab(of), nlab(of)))
                        continue;
                    uint32_t dce = ham(cands[j].id, cands[i].id);
                    double factor;
                    if (dce == 0) factor = cands[j].d == 0 ? 1.0 : std::numeric_limits<double>::max();
                    else factor = (double)cands[j].d / (double)dce;
                    max_factors[j] = std::max(max_factors[j], factor);
                }
            }
            alpha *= 1.2;
        }
        return results;
    }
    /* greedy_search_for_build: all visited nodes, stop rule identical to visit_closest */
    /* filter: visit_lsn_internal's label test with the new node's own labels as the query (insert_internal with
     * no_filter = false, AM/graph/mod.rs:664-672) */
    std::vector<Cand> search(uint32_t q, const std::vector<uint32_t>& starts, bool filter) {
        RustBinaryHeap<LSN, ReverseLsnLe> cand;
        std::vector<LSN> visited;
        U32Set inserted(L * R);
        std::vector<Cand> out;
        for (uint32_t start : starts) {
            if (!inserted.insert(start)) continue;
            cand.push(LSN{start, (float)ham(q, start)});
        }
        while (!cand.empty()) {
            if (visited.size() > L && lsn_cmp(cand.peek(), visited[L - 1]) >= 0) break;
            LSN head = cand.pop();
            size_t lo = 0, hi = visited.size();
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (lsn_cmp(visited[mid], head) < 0) lo = mid + 1;
                else hi = mid;
            }
            visited.insert(visited.begin() + (ptrdiff_t)lo, head);
            out.push_back(Cand{head.id, (uint32_t)head.dist});
            const uint32_t* r = nbrs + (size_t)head.id * stride;
            for (uint32_t s = 0; s < R && r[s] != VSO_INVALID_NODE; ++s) {
                if (!inserted.insert(r[s])) continue; /* marked before the label test (AM/sbq/storage.rs:148-172) */
                if (filter && !vso_labels_overlap(lab(q), nlab(q), lab(r[s]), nlab(r[s]))) continue;
                cand.push(LSN{r[s], (float)ham(q, r[s])});
            }
        }
        return out;
    }
    /* add_neighbors(neighbors_of, additional) */
    std::vector<Cand> add_neighbors(uint32_t of, const std::vector<Cand>& additional) {
        std::vector<Cand> cands;
        const uint32_t* r = nbrs + (size_t)of * stride;
        for (uint32_t s = 0; s < R && r[s] != VSO_INVALID_NODE; ++s) cands.push_back(Cand{r[s], ham(of, r[s])});
        for (const Cand& c : additional) {
            if (c.id == of) continue;
            bool dup = false;
            for (const Cand& e : cands)
                if (e.id == c.id) {
                    dup = true;
                    break;
                }
            if (!dup) cands.push_back(c);
        }
        std::vector<Cand> nl = cands.size() > R ? prune(cands, of) : cands;
        set_neighbors(of, nl);
        return nl;
    }
    /* insert_internal (AM/graph/mod.rs:664-717) */
    void insert_internal(uint32_t p, const std::vector<uint32_t>& starts, bool filter) {
        std::vector<Cand> v = search(p, starts, filter);
        std::vector<Cand> nl = add_neighbors(p, v);
        for (const Cand& q : nl) add_neighbors(q.id, std::vector<Cand>{Cand{p, q.d}});
    }
    void run() {
        for (size_t i = 0; i < (size_t)n * stride; ++i) nbrs[i] = VSO_INVALID_NODE;
        for (uint32_t p = 0; p < n; ++p) {
            /* update_start_nodes (AM/graph/mod.rs:490-531): node 0 is the default start node; a node is the start node of
             * every label it is the first to carry */
            if (loff)
                for (size_t t = 0; t < nlab(p); ++t) label_start.emplace(lab(p)[t], p);
            if (loff) { /* from the label start nodes, with the label filter (Graph::insert, AM/graph/mod.rs:649-652) */
                std::vector<uint32_t> starts;
                for (size_t t = 0; t < nlab(p); ++t) starts.push_back(label_start[lab(p)[t]]);
                insert_internal(p, starts, true);
            }
            if (p > 0) insert_internal(p, std::vector<uint32_t>{0u}, false); /* from the default start node, no filter */
        }
    }
};
}  // namespace

extern "C" {

/* Replays a push / pop sequence on the candidate heap (BinaryHeap<Reverse<ListSearchNeighbor>>, Ord on the distance only):
 * ops[i] = (key, id), key == 0xFFFFFFFF means pop; the remaining entries are popped at the end.  Writes the ids in pop order
 * and returns their number.  (What oracle/ref_kat.rs prints for the real std::collections::BinaryHeap.) */
size_t vso_heap_replay(const uint32_t* ops, size_t n_ops, uint32_t* out_ids) {
    RustBinaryHeap<LSN, ReverseLsnLe> h;
    size_t k = 0;
    for (size_t i = 0; i < n_ops; ++i) {
        if (ops[2 * i] == 0xFFFFFFFFu) {
            if (!h.empty()) out_ids[k++] = h.pop().id;
        } else {
            h.push(LSN{ops[2 * i + 1], (float)ops[2 * i]});
        }
    }
    while (!h.empty()) out_ids[k++] = h.pop().id;
    return k;
}

void vso_build_graph(uint32_t n, uint32_t words, const uint64_t* codes, uint32_t num_neighbors, uint32_t nbr_stride,
                     uint32_t search_list_size, double max_alpha, uint32_t* nbrs, uint32_t* default_start) {
    Builder b{n, words, num_neighbors, nbr_stride, search_list_size, max_alpha, codes, nbrs, nullptr, nullptr, {}};
    b.run();
    if (default_start) *default_start = n ? 0 : VSO_INVALID_NODE; /* first inserted node, AM/graph/mod.rs:490-540 */
}

/* Graph::insert over a labeled vector set (AM/graph/mod.rs:637-662): every node is inserted twice — from the start nodes of
 * its labels with the label filter, then from the default start node without it — and pruning keeps a candidate that shares a
 * label with the point which the occluding neighbor lacks.  start_labels / start_nodes (room for every distinct label, in
 * ascending label order) receive the per-label start nodes; returns their number. */
uint32_t vso_build_graph_labeled(uint32_t n, uint32_t words, const uint64_t* codes, const uint32_t* label_off,
                                 const int16_t* label_val, uint32_t num_neighbors, uint32_t nbr_stride,
                                 uint32_t search_list_size, double max_alpha, uint32_t* nbrs, uint32_t* default_start,
                                 int16_t* start_labels, uint32_t* start_nodes) {
    Builder b{n, words, num_neighbors, nbr_stride, search_list_size, max_alpha, codes, nbrs, nullptr, nullptr, {}};
    b.loff = label_off;
    b.lval = label_val;
    b.run();
    if (default_start) *default_start = n ? 0 : VSO_INVALID_NODE;
    uint32_t k = 0;
    for (const auto& kv : b.label_start) {
        start_labels[k] = kv.first;
        start_nodes[k] = kv.second;
        ++k;
    }
    return k;
}

void vso_bruteforce_topk(const vso_index* idx, const float* queries, uint32_t nq, uint32_t k, uint32_t n_threads,
                         uint32_t* out_nodes, float* out_dist) {
    parallel_queries(nq, n_threads, [&](uint32_t q, uint32_t) {
        std::vector<float> qf(queries + (size_t)q * idx->dim_full, queries + (size_t)(q + 1) * idx->dim_full);
        if (idx->distance_type == VSO_COSINE) vso_preprocess_cosine(qf.data(), idx->dim_full);
        std::vector<std::pair<float, uint32_t>> best; /* kept sorted ascending by (total_key, id) */
        std::vector<float> tmp(idx->dim_full);
        auto lt = [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
            int c = total_cmp(a.first, b.first);
            return c != 0 ? c < 0 : a.second < b.second;
        };
        for (uint32_t i = 0; i < idx->n; ++i) {
            if ((idx->heap_tids[i] & 0xFFFFu) == 0) continue;
            const float* v = idx->vecs + (size_t)i * idx->dim_full;
            if (idx->distance_type == VSO_COSINE) {
                std::memcpy(tmp.data(), v, sizeof(float) * idx->dim_full);
                vso_preprocess_cosine(tmp.data(), idx->dim_full);
                v = tmp.data();
            }
            float d = vso_distance_by_type((int)idx->distance_type, v, qf.data(), idx->dim_full);
            std::pair<float, uint32_t> e{d, i};
            if (best.size() < k) {
                best.insert(std::upper_bound(best.begin(), best.end(), e, lt), e);
            } else if (lt(e, best.back())) {
                best.pop_back();
                best.insert(std::upper_bound(best.begin(), best.end(), e, lt), e);
            }
        }
        for (uint32_t i = 0; i < k; ++i) {
            out_nodes[(size_t)q * k + i] = i < best.size() ? best[i].second : VSO_INVALID_NODE;
            if (out_dist) out_dist[(size_t)q * k + i] = i < best.size() ? best[i].first : std::numeric_limits<float>::quiet_NaN();
        }
    });
}

void vso_hamming_scan_topk(const uint64_t* codes, uint32_t n, uint32_t words, const uint64_t* qcodes, uint32_t nq,
                           uint32_t k, uint32_t* out_nodes, uint32_t* out_ham) {
    vso_hamming_scan_topk_filtered(codes, n, words, nullptr, nullptr, nullptr, qcodes, nullptr, nullptr, nq, k, out_nodes, out_ham);
}

/* exact SBQ top-k among the rows a label-filtered scan may return: label sets overlap (an empty key filters nothing), heap tuple
 * not deleted (heap_tids == nullptr: not checked) */
void vso_hamming_scan_topk_filtered(const uint64_t* codes, uint32_t n, uint32_t words, const uint32_t* label_off,
                                    const int16_t* label_val, const uint64_t* heap_tids, const uint64_t* qcodes,
                                    const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq, uint32_t k,
                                    uint32_t* out_nodes, uint32_t* out_ham) {
    for (uint32_t q = 0; q < nq; ++q) {
        std::vector<uint64_t> best; /* (ham<<32 | id) ascending */
        const uint64_t* qc = qcodes + (size_t)q * words;
        for (uint32_t i = 0; i < n; ++i) {
            if (heap_tids && (heap_tids[i] & 0xFFFFu) == 0) continue;
            if (qlabel_off && qlabel_off[q + 1] > qlabel_off[q] &&
                !vso_labels_overlap(qlabels + qlabel_off[q], qlabel_off[q + 1] - qlabel_off[q], label_val + label_off[i],
                                    label_off[i + 1] - label_off[i]))
                continue;
            uint64_t key = (vso_distance_xor(codes + (size_t)i * words, qc, words) << 32) | i;
            if (best.size() < k) best.insert(std::upper_bound(best.begin(), best.end(), key), key);
            else if (key < best.back()) {
                best.pop_back();
                best.insert(std::upper_bound(best.begin(), best.end(), key), key);
            }
        }
        for (uint32_t i = 0; i < k; ++i) {
            out_nodes[(size_t)q * k + i] = i < best.size() ? (uint32_t)best[i] : VSO_INVALID_NODE;
            out_ham[(size_t)q * k + i] = i < best.size() ? (uint32_t)(best[i] >> 32) : 0xFFFFFFFFu;
        }
    }
}

}  // extern "C"
