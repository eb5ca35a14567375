// vs_search.hip — K3: the streaming beam search of the `diskann` scan, one wave64 per scan.
//
//   ListSearchResult (candidates / visited / inserted)          AM/graph/mod.rs:74-185
//   greedy_search_streaming_init / greedy_search_iterate        AM/graph/mod.rs:331-385
//   SbqSpeedupStorage::visit_lsn_internal (Disk arm)            AM/sbq/storage.rs:135-190
//   create_lsn_for_start_node / return_lsn                      AM/sbq/storage.rs:365-414
//   TSVResponseIterator::next (skip deleted)                    AM/scan.rs:210-242
//   greedy_search_for_build (BUILD=true)                        AM/graph/mod.rs:285-327
//
// Why one wave per scan: a scan is a serial chain of dependent expansions (pop closest -> read its neighbor list ->
// score the unseen neighbors -> push); the parallelism inside one expansion is <= R gathered code rows, which 64
// lanes cover (4 lanes x 16 B per 192-B row, 16 rows per pass).  Throughput comes from thousands of concurrent
// scans hiding each other's HBM latency, so per-scan on-chip state is kept small:
//   * candidate heap: 8-byte entries (hamming << 32 | node), positions [0, hl) in LDS, deeper positions in a
//     per-scan global array (only touched when a heap outgrows hl);
//   * dedup set ("inserted"): exact open-addressing table in LDS first (ds atomic CAS), overflowing into a
//     per-scan global table that is cleared lazily by the wave itself;
//   * visited list: sorted (hamming, node) arrays in LDS.
// Latency: after each pop the new heap root is the most likely next expansion, so its neighbor row is requested
// right away and is (usually) in registers when the next iteration starts — the row fetch overlaps the code gathers.
//
// Exactness: the candidate heap replays Rust std's BinaryHeap (push = sift_up; pop = swap last into the root,
// sift_down_to_bottom, sift_up) because the order in which EQUAL Hamming distances are expanded depends on it;
// comparisons look at the Hamming field only (DistanceWithTieBreak with the constant tie-break of with_query,
// AM/graph/neighbor_with_distance.rs:31-43,74-83).  visited.insert goes before equal elements (partition_point,
// AM/graph/mod.rs:167-168).  prepare_insert marks a neighbor before the label check (AM/sbq/storage.rs:148-172).
#include "vs_device.h"

struct SearchArgs {
    const uint64_t* codes;
    const uint32_t* nbrs;
    const uint64_t* tids;
    const uint32_t* label_off;
    const int16_t* label_val;
    const int16_t* ls_labels;
    const uint32_t* ls_nodes;
    uint32_t code_stride, nbr_stride, R, n, n_ls, default_start;
    SearchLaunch s;
    // `plain` storage (AM/plain/storage.rs): candidates are scored with the full-precision distance to the node's vector
    // (appended so that the kernel-argument layout of everything above is the one the SBQ kernels were measured with)
    const float* vecs;
    const float* vnorm;
    const float* q_full;  // [nq][vec_stride] prepared (cosine-normalised) queries
    uint32_t vec_stride, dim, distance_type;
};

#define MAX_QLABELS 64

// monotone u32 image of f32::total_cmp (DistanceWithTieBreak compares distances with total_cmp,
// AM/graph/neighbor_with_distance.rs:74-83): heap / visited keys of the plain-storage search
__device__ __forceinline__ uint32_t plain_key(float f) {
    int32_t b = __float_as_int(f);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return (uint32_t)b ^ 0x80000000u;
}

// PlainDistanceMeasure::calculate_distance (AM/plain/storage.rs:239-247,273-281): distance_fn(query index slice, node
// vector) for the row each 8-lane group points at, in the reference's AVX2 accumulation order — the same arithmetic as
// k_rerank (vs_kernels.hip): lane l8 owns elements 32t + 4 l8 .. +3, i.e. 4 of the 32 virtual AVX2 lanes; L2 is mul + add,
// dot is FMA; horizontal_add_ps per accumulator, the four accumulators summed left to right, then the scalar tail.  The
// stored vector is the cosine-normalised insert-time vector: the raw row divided by its cached norm.  Valid on l8 == 0.
__device__ __forceinline__ float plain_dist8(const float* __restrict__ row, float sdiv, const float* qv, uint32_t dim,
                                             uint32_t distance_type, int lane, bool valid) {
    const int l8 = lane & 7;
    const uint32_t steps = dim / 32;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (valid) {
        if (distance_type == VS_L2) {
            for (uint32_t t = 0; t < steps; ++t) {
                const float4 x = *reinterpret_cast<const float4*>(row + 32 * t + 4 * l8);
                const float4 y = *reinterpret_cast<const float4*>(qv + 32 * t + 4 * l8);
                const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                const float p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2, p3 = d3 * d3;
                a0 = a0 + p0;
                a1 = a1 + p1;
                a2 = a2 + p2;
                a3 = a3 + p3;
            }
        } else {
            for (uint32_t t = 0; t < steps; ++t) {
                float4 x = *reinterpret_cast<const float4*>(row + 32 * t + 4 * l8);
                const float4 y = *reinterpret_cast<const float4*>(qv + 32 * t + 4 * l8);
                if (sdiv != 0.0f) {
                    x.x = x.x / sdiv;
                    x.y = x.y / sdiv;
                    x.z = x.z / sdiv;
                    x.w = x.w / sdiv;
                }
                a0 = __builtin_fmaf(x.x, y.x, a0);
                a1 = __builtin_fmaf(x.y, y.y, a1);
                a2 = __builtin_fmaf(x.z, y.z, a2);
                a3 = __builtin_fmaf(x.w, y.w, a3);
            }
        }
    }
    const float s0 = a0 + __shfl(a0, lane ^ 1, WAVE);
    const float s1 = a1 + __shfl(a1, lane ^ 1, WAVE);
    const float s2 = a2 + __shfl(a2, lane ^ 1, WAVE);
    const float s3 = a3 + __shfl(a3, lane ^ 1, WAVE);
    const float t0 = s0 + s1;
    const float t1 = s2 + s3;
    const float h = t0 + t1;
    const int g0 = lane & ~7;
    const float h0 = __shfl(h, g0 + 0, WAVE), h1 = __shfl(h, g0 + 2, WAVE), h2 = __shfl(h, g0 + 4, WAVE), h3 = __shfl(h, g0 + 6, WAVE);
    float dist = h0 + h1;
    dist = dist + h2;
    dist = dist + h3;
    float r = 0.0f;
    if (valid && l8 == 0) {
        for (uint32_t i = steps * 32; i < dim; ++i) {  // scalar tail, in element order
            float x = row[i];
            if (distance_type == VS_L2) {
                const float diff = x - qv[i];
                const float p = diff * diff;
                dist = dist + p;
            } else {
                if (sdiv != 0.0f) x = x / sdiv;
                const float p = x * qv[i];
                dist = dist + p;
            }
        }
        if (distance_type == VS_L2) r = dist;
        else if (distance_type == VS_IP) r = -dist;
        else r = fmaxf(1.0f - dist, 0.0f);
    }
    return r;
}

__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    uint32_t lo = rfl((uint32_t)v), hi = rfl((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t gload64(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-served: never a stale L1 line
}
__device__ __forceinline__ void gstore64(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// BinaryHeap<Reverse<ListSearchNeighbor>> (AM/graph/mod.rs:75).  Heap position i lives at l[i + 1] (so that sibling
// pairs are 16-byte aligned) while i < hl, else at g[i - hl].  All indices are wave-uniform.
struct WaveHeap {
    uint64_t* l;
    uint64_t* g;
    uint32_t hl, len, maxlen;
    __device__ __forceinline__ static uint32_t key(uint64_t e) { return (uint32_t)(e >> 32); }
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return rfl64(i < hl ? l[i + 1] : gload64(g + (i - hl))); }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v, int lane) {
        if (lane == 0) {
            if (i < hl) l[i + 1] = v;
            else gstore64(g + (i - hl), v);
        }
    }
    // sift_up(0, pos): while pos > 0 { parent = (pos-1)/2; if elem <= parent break; ... }; Reverse => parent.d <= elem.d
    __device__ __forceinline__ void sift_up(uint32_t pos, uint64_t elem, int lane) {
        const uint32_t ek = key(elem);
        while (pos > 0) {
            uint32_t parent = (pos - 1) >> 1;
            uint64_t pe = get(parent);
            if (key(pe) <= ek) break;
            set(pos, pe, lane);
            pos = parent;
        }
        set(pos, elem, lane);
    }
    __device__ __forceinline__ void push(uint64_t elem, int lane) {
        uint32_t pos = len;
        len = pos + 1;
        maxlen = max(maxlen, len);
        sift_up(pos, elem, lane);
    }
    // pop(): Vec::pop, swap with data[0], sift_down_to_bottom(0) (always to a leaf), then sift_up
    __device__ __forceinline__ uint64_t pop(int lane) {
        uint64_t item = get(len - 1);
        len -= 1;
        if (len == 0) return item;
        uint64_t top = get(0);
        const uint32_t end = len;
        uint32_t pos = 0, child = 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            uint64_t le, ri;
            if (child + 1 < hl) {
                const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(l + child + 1);
                le = rfl64(pr.x);
                ri = rfl64(pr.y);
            } else {
                le = get(child);
                ri = get(child + 1);
            }
            // child += (data[child] <= data[child+1]); Reverse => right.d <= left.d picks the right child
            uint32_t pick = (key(ri) <= key(le)) ? 1u : 0u;
            child += pick;
            set(pos, pick ? ri : le, lane);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            set(pos, get(child), lane);
            pos = child;
        }
        sift_up(pos, item, lane);
        return top;
    }
};

template <bool BUILD, bool PLAIN>
__global__ __launch_bounds__(WAVE) void k_search(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const SearchLaunch& s = a.s;
    if (q >= s.nq) return;
    if (s.only_failed) {  // follow-up of the fast kernel: only the scans it handed over
        if (s.status[q] == 0) return;
        if (threadIdx.x == 0 && s.fb_flag) s.fb_flag[q] = 1;
    }
    uint32_t wslot = q;  // which region of heap_g / hash this scan uses
    if (s.pool_counter) {
        if (threadIdx.x == 0) wslot = atomicAdd(s.pool_counter, 1u);
        wslot = rfl(wslot);
        if (wslot >= s.pool_slots) {  // pool exhausted: the host re-launches the scans that are still marked
            if (threadIdx.x == 0) {
                s.status[q] = OVF_POOL;
                s.out_cnt[q] = 0;  // an empty stream: the rerank / resort kernels behind this launch stay in bounds
            }
            return;
        }
    }

    // ---- LDS carve (every offset a multiple of 16 B) ----
    uint64_t* heap_l = reinterpret_cast<uint64_t*>(smem);                        // hl + 2 entries
    uint32_t* lhash = reinterpret_cast<uint32_t*>(heap_l + round_up_u32(s.hl + 2, 2));
    uint32_t* vdist = lhash + round_up_u32(s.lh, 4);
    uint32_t* vid = vdist + round_up_u32(s.vcap, 4);
    uint32_t* surv_id = vid + round_up_u32(s.vcap, 4);
    uint32_t* surv_d = surv_id + 64;
    uint64_t* qc = reinterpret_cast<uint64_t*>(surv_d + 64);
    int16_t* ql = reinterpret_cast<int16_t*>(qc + a.code_stride);
    float* qf = reinterpret_cast<float*>(ql + MAX_QLABELS);  // PLAIN only: the prepared query, vec_stride floats (16-B aligned)

    if (PLAIN) {
        for (uint32_t i = lane; i < a.vec_stride; i += WAVE) qf[i] = a.q_full[(size_t)q * a.vec_stride + i];
    } else {
        for (uint32_t w = lane; w < a.code_stride; w += WAVE) qc[w] = s.qcodes[(size_t)q * a.code_stride + w];
    }
    // resumable scan: state saved by an earlier launch (header + LDS image of heap top / dedup table / visited list)
    uint32_t* rs = s.resume ? s.resume + (size_t)q * s.resume_stride : nullptr;
    const bool resumed = rs != nullptr && rfl(rs[RS_INIT]) == 1u;
    const uint32_t image_words = 2u * round_up_u32(s.hl + 2, 2) + round_up_u32(s.lh, 4) + 2u * round_up_u32(s.vcap, 4);
    if (resumed) {
        uint32_t* img = reinterpret_cast<uint32_t*>(smem);
        for (uint32_t i = lane; i < image_words; i += WAVE) img[i] = rs[RS_HDR + i];
    } else {
        for (uint32_t i = lane; i < s.lh; i += WAVE) lhash[i] = VS_EMPTY;
    }
    const bool labels_some = s.qlabel_off != nullptr;  // LabeledVector.labels is Some (AM/labels/mod.rs:222-236)
    // the key's labels (LabelSet: sorted, distinct — any number of them, AM/labels/mod.rs:19-37): up to MAX_QLABELS are staged
    // in LDS, a wider key is read where it lies in global memory
    uint32_t nql = 0;
    const int16_t* qlp = ql;
    if (labels_some) {
        uint32_t lb = s.qlabel_off[q], le = s.qlabel_off[q + 1];
        nql = le - lb;
        if (nql <= (uint32_t)MAX_QLABELS) {
            for (uint32_t i = lane; i < nql; i += WAVE) ql[i] = s.qlabels[lb + i];
        } else {
            qlp = s.qlabels + lb;
        }
    }
    const bool has_label_filter = labels_some && nql > 0;  // AM/scan.rs:189 ; no_filter = !has_label_filter
    __syncthreads();

    // global dedup overflow: a ladder of tables (level j has g0 << j slots at offset (g0 << j) - g0); a new level is
    // opened (and cleared by this wave) when the current one is half full, so small scans stay cache-resident and
    // nothing is ever re-hashed.  Membership = present in ANY level; inserts go to the top level.
    uint32_t* ghash = s.hash + (size_t)wslot * s.hashcap;
    const uint32_t g0 = s.g0;
    const uint32_t lmask = s.lh ? s.lh - 1 : 0;
    const uint32_t l_full_at = s.lh ? (s.lh / 4) * 3 : 0;  // stop inserting into the LDS table at 75 % load
    int glev = -1;                                            // top level in use (-1: global ladder untouched)
    uint32_t nins_l = 0, nins_g = 0, nins_top = 0;

    WaveHeap heap{heap_l, s.heap_g + (size_t)wslot * (s.hcap > s.hl ? s.hcap - s.hl : 0), s.hl, 0, 0};
    uint32_t vlen = 0, emitted = 0, status = 0;
    uint32_t st_visits = 0, st_cand = 0, st_dq = 0, st_reads = 0, st_next = 0, st_invis = 0;
    if (resumed) {  // (all wave-uniform)
        heap.len = rfl(rs[RS_HLEN]);
        heap.maxlen = rfl(rs[RS_HMAX]);
        vlen = rfl(rs[RS_VLEN]);
        glev = (int)rfl(rs[RS_GLEV]) - 1;
        nins_l = rfl(rs[RS_NINS_L]);
        nins_g = rfl(rs[RS_NINS_G]);
        nins_top = rfl(rs[RS_NINS_TOP]);
        st_visits = rfl(rs[RS_VISITS]);
        st_cand = rfl(rs[RS_CAND]);
        st_dq = rfl(rs[RS_DQ]);
        st_reads = rfl(rs[RS_READS]);
        st_next = rfl(rs[RS_NEXT]);
        st_invis = rfl(rs[RS_INVIS]);
        status = rfl(rs[RS_STATUS]);
    }

    auto open_level = [&]() {  // uniform
        glev++;
        const uint32_t size = g0 << glev, off = size - g0;
        if ((uint64_t)off + size > s.hashcap) {
            status |= OVF_HASH;
            glev--;
            return;
        }
        for (uint32_t i = 4u * lane; i < size; i += 4u * WAVE)
            *reinterpret_cast<uint4*>(ghash + off + i) = make_uint4(VS_EMPTY, VS_EMPTY, VS_EMPTY, VS_EMPTY);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        nins_top = 0;
    };

    // HashSet::insert for the active lanes' node ids; returns true where the id was not present before.
    auto prepare_insert = [&](uint32_t nid, bool act) -> bool {
        bool fresh = false, need_g = false;
        if (act) {
            if (s.lh) {
                uint32_t slot = hash_u32(nid) & lmask;
                if (nins_l + WAVE <= l_full_at) {
                    for (uint32_t probe = 0; probe <= lmask; ++probe) {
                        uint32_t old = atomicCAS(&lhash[slot], VS_EMPTY, nid);  // ds_cmpst_rtn_b32
                        if (old == VS_EMPTY) { fresh = true; break; }
                        if (old == nid) break;
                        slot = (slot + 1) & lmask;
                    }
                } else {  // LDS table frozen: membership test only, misses continue in the global ladder
                    need_g = true;
                    for (uint32_t probe = 0; probe <= lmask; ++probe) {
                        uint32_t v = lhash[slot];
                        if (v == nid) { need_g = false; break; }
                        if (v == VS_EMPTY) break;
                        slot = (slot + 1) & lmask;
                    }
                }
            } else {
                need_g = true;
            }
        }
        nins_l += (uint32_t)__popcll(__ballot(fresh));
        if (__ballot(need_g)) {
            if (glev < 0 || (nins_top + WAVE) * 2u > (g0 << glev)) open_level();
            if (glev < 0) return fresh;  // OVF_HASH raised
            const uint32_t hh = hash_u32(nid ^ 0x5bd1e995u);
            // older levels: read-only probes served by L2 (entries were written by this wave's L2 atomics)
            for (int lev = 0; lev < glev && __ballot(need_g); ++lev) {
                if (need_g) {
                    const uint32_t size = g0 << lev, off = size - g0, m = size - 1;
                    uint32_t slot = hh & m;
                    for (uint32_t probe = 0; probe <= m; ++probe) {
                        uint32_t v = __hip_atomic_load(&ghash[off + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v == nid) { need_g = false; break; }
                        if (v == VS_EMPTY) break;
                        slot = (slot + 1) & m;
                    }
                }
            }
            bool gfresh = false;
            if (need_g) {
                const uint32_t size = g0 << glev, off = size - g0, m = size - 1;
                uint32_t slot = hh & m;
                for (uint32_t probe = 0; probe <= m; ++probe) {
                    uint32_t old = atomicCAS(&ghash[off + slot], VS_EMPTY, nid);
                    if (old == VS_EMPTY) { gfresh = true; break; }
                    if (old == nid) break;
                    slot = (slot + 1) & m;
                }
            }
            const uint32_t ng = (uint32_t)__popcll(__ballot(gfresh));
            nins_g += ng;
            nins_top += ng;
            fresh = fresh || gfresh;
        }
        return fresh;
    };

    // ---- ListSearchResult::new: start nodes (AM/graph/mod.rs:97-124, AM/graph/start_nodes.rs:39-48) ----
    if (!resumed) {
        uint32_t nstarts = labels_some ? nql : 1u;
        if (a.default_start == VS_INVALID_NODE || a.n == 0) nstarts = 0;  // ListSearchResult::empty()
        for (uint32_t si = 0; si < nstarts; ++si) {
            uint32_t sn = VS_INVALID_NODE;
            if (!labels_some) {
                sn = a.default_start;
            } else {
                int16_t lab = qlp[si];
                int lo = 0, hi = (int)a.n_ls;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (a.ls_labels[mid] < lab) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < (int)a.n_ls && a.ls_labels[lo] == lab) sn = a.ls_nodes[lo];
            }
            sn = rfl(sn);
            if (sn == VS_INVALID_NODE) continue;
            // create_lsn_for_start_node (AM/sbq/storage.rs:365-391)
            bool fr = prepare_insert(sn, lane == 0);
            if (!rfl(fr ? 1u : 0u)) continue;
            st_reads++;
            uint32_t d;
            if (PLAIN) {
                const float sdiv = a.distance_type == VS_COSINE ? a.vnorm[sn] : 0.0f;
                d = rfl(plain_key(plain_dist8(a.vecs + (size_t)sn * a.vec_stride, sdiv, qf, a.dim, a.distance_type, lane, lane < 8)));
            } else {
                d = rfl(ham_row4(a.codes + (size_t)sn * a.code_stride, qc, lane & 3, a.code_stride, lane < 4));
            }
            st_dq++;
            st_cand++;
            heap.push(((uint64_t)d << 32) | sn, lane);
        }
    }

    uint32_t pf_node = VS_INVALID_NODE, pf_val = VS_INVALID_NODE;  // prefetched first chunk of a neighbor row

    // ---- TSVResponseIterator::next, repeated until M rows are emitted (AM/scan.rs:210-242) ----
    while (emitted < s.M && status == 0) {
        st_next++;
        bool got = false;
        while (true) {  // "Iterate until we find a non-deleted tuple"
            // ---- greedy_search_iterate (AM/graph/mod.rs:357-385) ----
            while (true) {
                // visit_closest(L) (AM/graph/mod.rs:153-170)
                if (heap.len == 0) break;
                if (vlen > s.L) {
                    uint32_t node_at_pos = rfl(vdist[s.L - 1]);
                    if (WaveHeap::key(heap.get(0)) >= node_at_pos) break;
                }
                const uint64_t head = heap.pop(lane);
                const uint32_t hd = WaveHeap::key(head);
                const uint32_t node = (uint32_t)head;
                // neighbor row (first 64 slots): prefetched during the previous expansion if the prediction held
                const uint32_t* nrow = a.nbrs + (size_t)node * a.nbr_stride;
                uint32_t row0;
                if (node == pf_node) {
                    row0 = pf_val;
                    (void)0;
                } else {
                    row0 = ((uint32_t)lane < a.R) ? nrow[lane] : VS_INVALID_NODE;
                }
                // request the row of the new heap root: the most likely next expansion
                if (heap.len > 0) {
                    pf_node = (uint32_t)heap.get(0);
                    pf_val = ((uint32_t)lane < a.R) ? a.nbrs[(size_t)pf_node * a.nbr_stride + lane] : VS_INVALID_NODE;
                } else {
                    pf_node = VS_INVALID_NODE;
                }
                // visited.insert(partition_point(|x| *x < head), head): before the first element >= head
                if (vlen + 1 > s.vcap) {
                    if (BUILD) vlen = s.vcap - 1;  // build mode keeps the closest vcap visited nodes as prune candidates
                    else { status |= OVF_VISITED; break; }
                }
                {
                    uint32_t cntlt = 0;
                    for (uint32_t base = 0; base < vlen; base += WAVE) {
                        uint32_t i = base + lane;
                        bool lt = i < vlen && vdist[i] < hd;
                        cntlt += (uint32_t)__popcll(__ballot(lt));
                    }
                    const uint32_t idx = cntlt;
                    uint32_t hi = vlen;
                    while (hi > idx) {  // shift [idx, vlen) right by one, top chunk first
                        uint32_t lo = (hi - idx > WAVE) ? hi - WAVE : idx;
                        uint32_t i = lo + lane;
                        uint32_t td = 0, ti = 0;
                        if (i < hi) { td = vdist[i]; ti = vid[i]; }
                        __syncthreads();
                        if (i < hi) { vdist[i + 1] = td; vid[i + 1] = ti; }
                        __syncthreads();
                        hi = lo;
                    }
                    if (lane == 0) { vdist[idx] = hd; vid[idx] = node; }
                    vlen++;
                    __syncthreads();
                }
                st_visits++;
                // ---- visit_lsn_internal, Disk arm (AM/sbq/storage.rs:135-190) ----
                st_reads++;  // SbqNode::read(visiting)
                bool list_ended = false;
                for (uint32_t c0 = 0; c0 < a.R && !list_ended && status == 0; c0 += WAVE) {
                    uint32_t slot = c0 + lane;
                    uint32_t nid = c0 == 0 ? row0 : ((slot < a.R) ? nrow[slot] : VS_INVALID_NODE);
                    // list ends at the first InvalidBlockNumber (AM/sbq/node.rs:260-285)
                    uint64_t inval = __ballot(nid == VS_INVALID_NODE);
                    uint32_t nvalid = inval ? (uint32_t)__builtin_ctzll(inval) : WAVE;
                    if (nvalid < WAVE) list_ended = true;
                    bool act = (uint32_t)lane < nvalid;
                    // prepare_insert (marks BEFORE the label check, AM/sbq/storage.rs:148-172)
                    bool fresh = prepare_insert(nid, act);
                    st_reads += (uint32_t)__popcll(__ballot(fresh));  // SbqNode::read(neighbor)
                    if (status) break;
                    // label filter: query.labels.overlaps(node.labels) (AM/labels/mod.rs:124-142)
                    bool pass = fresh;
                    if (fresh && has_label_filter) {
                        uint32_t lb = a.label_off[nid], le = a.label_off[nid + 1];
                        uint32_t i = 0, j = lb;
                        bool ov = false;
                        while (i < nql && j < le) {
                            int16_t x = qlp[i], y = a.label_val[j];
                            if (x == y) { ov = true; break; }
                            if (x < y) ++i;
                            else ++j;
                        }
                        pass = ov;
                    }
                    uint64_t pm = __ballot(pass);
                    uint32_t c = (uint32_t)__popcll(pm);
                    if (c == 0) continue;
                    // compact survivors in neighbor-list order
                    if (pass) surv_id[__popcll(pm & ((1ull << lane) - 1ull))] = nid;
                    __syncthreads();
                    if (PLAIN) {  // full-precision distances: 8 lanes per vector row, 8 rows per pass
                        for (uint32_t p0 = 0; p0 < c; p0 += 8) {
                            const uint32_t j = p0 + (uint32_t)(lane >> 3);
                            const bool valid = j < c;
                            const uint32_t id = valid ? surv_id[j] : 0;
                            const float sdiv = (valid && a.distance_type == VS_COSINE) ? a.vnorm[id] : 0.0f;
                            const float r = plain_dist8(a.vecs + (size_t)id * a.vec_stride, sdiv, qf, a.dim, a.distance_type, lane, valid);
                            if (valid && (lane & 7) == 0) surv_d[j] = plain_key(r);
                        }
                    } else
                    // distances: 4 lanes per code row, 16 rows per pass, all loads of the chunk issued up front
#pragma unroll
                    for (int pass_i = 0; pass_i < 4; ++pass_i) {
                        uint32_t j = (uint32_t)pass_i * 16u + (uint32_t)(lane >> 2);
                        bool valid = j < c;
                        uint32_t id = valid ? surv_id[j] : 0;
                        uint32_t d = ham_row4(a.codes + (size_t)id * a.code_stride, qc, lane & 3, a.code_stride, valid);
                        if (valid && (lane & 3) == 0) surv_d[j] = d;
                    }
                    st_dq += c;
                    st_cand += c;
                    if (heap.len + c > s.hcap) { status |= OVF_HEAP; break; }
                    __syncthreads();
                    // insert_neighbor in list order (AM/graph/mod.rs:144-147)
                    for (uint32_t j = 0; j < c; ++j) {
                        uint32_t d = rfl(surv_d[j]);
                        uint32_t id = rfl(surv_id[j]);
                        heap.push(((uint64_t)d << 32) | id, lane);
                    }
                }
                if (status) break;
            }
            if (status) break;
            if (BUILD) break;
            // ---- consume (AM/graph/mod.rs:174-184) + return_lsn (AM/sbq/storage.rs:404-414) ----
            if (vlen == 0) break;  // None
            __syncthreads();
            const uint32_t fd = rfl(vdist[0]);
            const uint32_t fnode = rfl(vid[0]);
            {  // visited.remove(0)
                uint32_t lo = 1;
                while (lo < vlen) {
                    uint32_t i = lo + lane;
                    uint32_t hi = min(lo + WAVE, vlen);
                    uint32_t td = 0, ti = 0;
                    if (i < hi) { td = vdist[i]; ti = vid[i]; }
                    __syncthreads();
                    if (i < hi) { vdist[i - 1] = td; vid[i - 1] = ti; }
                    __syncthreads();
                    lo = hi;
                }
                vlen--;
            }
            st_reads++;
            const uint64_t tid = a.tids[fnode];
            if ((tid & 0xFFFFull) == 0) continue;  // InvalidOffsetNumber: deleted tuple (AM/scan.rs:231-234)
            if (s.visible && s.visible[fnode] == 0) {  // get_full_distance_for_resort -> None (AM/scan.rs:268-272)
                st_invis++;
                st_next++;  // the caller's loop asks `next` again
                continue;
            }
            if (lane == 0) {
                s.out_ids[(size_t)q * s.M + emitted] = fnode;
                s.out_ham[(size_t)q * s.M + emitted] = fd;
                if (s.row_stats) {  // the counters as the reference's stood when this row left next()
                    uint32_t* r = s.row_stats + ((size_t)q * s.M + emitted) * ST_N;
                    r[ST_VISITS] = st_visits;
                    r[ST_CAND] = st_cand;
                    r[ST_DQ] = st_dq;
                    r[ST_READS] = st_reads;
                    r[ST_NEXT] = st_next;
                    r[ST_GSPILL] = heap.maxlen;
                    r[ST_INVIS] = st_invis;
                    r[7] = nins_g;
                }
            }
            emitted++;
            got = true;
            break;
        }
        if (!got) break;
    }
    if (BUILD) {
        __syncthreads();
        emitted = min(vlen, s.M);
        for (uint32_t i = lane; i < emitted; i += WAVE) {
            s.out_ids[(size_t)q * s.M + i] = vid[i];
            s.out_ham[(size_t)q * s.M + i] = vdist[i];
        }
    }
    for (uint32_t i = emitted + lane; i < s.M; i += WAVE) {
        s.out_ids[(size_t)q * s.M + i] = VS_INVALID_NODE;
        s.out_ham[(size_t)q * s.M + i] = 0xFFFFFFFFu;
    }
    if (rs) {  // save the scan for the next launch (a failed scan keeps its flag: the host restarts it with larger capacities)
        __syncthreads();
        const uint32_t* img = reinterpret_cast<const uint32_t*>(smem);
        for (uint32_t i = lane; i < image_words; i += WAVE) rs[RS_HDR + i] = img[i];
        if (lane == 0) {
            rs[RS_INIT] = 1u;
            rs[RS_HLEN] = heap.len;
            rs[RS_HMAX] = heap.maxlen;
            rs[RS_VLEN] = vlen;
            rs[RS_GLEV] = (uint32_t)(glev + 1);
            rs[RS_NINS_L] = nins_l;
            rs[RS_NINS_G] = nins_g;
            rs[RS_NINS_TOP] = nins_top;
            rs[RS_VISITS] = st_visits;
            rs[RS_CAND] = st_cand;
            rs[RS_DQ] = st_dq;
            rs[RS_READS] = st_reads;
            rs[RS_NEXT] = st_next;
            rs[RS_INVIS] = st_invis;
            rs[RS_STATUS] = status;
            rs[RS_EMITTED] = (resumed ? rs[RS_EMITTED] : 0u) + (status ? 0u : emitted);
        }
    }
    if (lane == 0) {
        s.out_cnt[q] = status ? 0 : emitted;  // a failed scan publishes an empty stream (it is re-run or reported)
        s.status[q] = status;
        uint32_t* st = s.stats + (size_t)q * ST_N;
        st[ST_VISITS] = st_visits;
        st[ST_CAND] = st_cand;
        st[ST_DQ] = st_dq;
        st[ST_READS] = st_reads;
        st[ST_NEXT] = st_next;
        st[ST_GSPILL] = heap.maxlen;
        st[ST_INVIS] = st_invis;
        st[7] = nins_g;
    }
}

size_t search_resume_words(const SearchLaunch& s) {
    return RS_HDR + 2 * (size_t)round_up_u32(s.hl + 2, 2) + round_up_u32(s.lh, 4) + 2 * (size_t)round_up_u32(s.vcap, 4);
}

size_t search_lds_bytes(const vs_index* idx, const SearchLaunch& s) {
    size_t b = (size_t)round_up_u32(s.hl + 2, 2) * 8 + (size_t)round_up_u32(s.lh, 4) * 4 +
               2 * (size_t)round_up_u32(s.vcap, 4) * 4 + 128 * 4 + (size_t)idx->code_stride * 8 + MAX_QLABELS * 2 + 16;
    if (idx->d.storage_type == VS_STORAGE_PLAIN) b += (size_t)idx->vec_stride * 4;
    return (b + 15) / 16 * 16;
}

int launch_search(vs_index* idx, const SearchLaunch& s, bool build_mode) {
    if (s.nq == 0) return VS_OK;
    SearchArgs a;
    a.codes = idx->codes;
    a.nbrs = idx->nbrs;
    a.tids = idx->tids;
    a.label_off = idx->label_off;
    a.label_val = idx->label_val;
    a.ls_labels = idx->ls_labels;
    a.ls_nodes = idx->ls_nodes;
    a.code_stride = idx->code_stride;
    a.nbr_stride = idx->nbr_stride;
    a.R = idx->d.num_neighbors;
    a.n = idx->d.n;
    a.n_ls = idx->d.n_label_starts;
    a.default_start = idx->d.default_start;
    const bool plain = idx->d.storage_type == VS_STORAGE_PLAIN;
    a.vecs = idx->vecs;
    const bool truncated = plain && idx->d.dim_index < idx->d.dim_full;  // index slice: its own norm cache / prepared query
    a.vnorm = truncated ? idx->vnorm_idx : idx->vnorm;
    a.q_full = (const float*)(truncated ? idx->ws.q_index.p : idx->ws.q_full.p);
    a.vec_stride = idx->vec_stride;
    a.dim = idx->d.dim_index;
    a.distance_type = idx->d.distance_type;
    VS_REQUIRE(!plain || (!build_mode && idx->vecs && a.q_full && !s.qlabel_off),
               "plain storage search needs the vector column and takes no label keys (AM/plain/storage.rs:262)");
    a.s = s;
    size_t lds = search_lds_bytes(idx, s);
    if (lds > 160 * 1024) {
        vs_set_error("search state does not fit LDS (%zu B): search_list_size / visited capacity too large", lds);
        return VS_ERR_CAPACITY;
    }
    if ((s.lh & (s.lh - 1)) != 0 || (s.g0 & (s.g0 - 1)) != 0 || s.g0 < 256 || s.hashcap < s.g0) {
        vs_set_error("dedup table sizes must be powers of two");
        return VS_ERR_INVALID;
    }
    static DeviceOnce attr_set;
    const int attr_dev = idx->ctx->device;
    if (attr_set.pending(attr_dev)) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<false, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<true, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done(attr_dev);
    }
    if (plain) hipLaunchKernelGGL((k_search<false, true>), dim3(s.nq), dim3(WAVE), lds, idx->ctx->stream, a);
    else if (build_mode) hipLaunchKernelGGL((k_search<true, false>), dim3(s.nq), dim3(WAVE), lds, idx->ctx->stream, a);
    else hipLaunchKernelGGL((k_search<false, false>), dim3(s.nq), dim3(WAVE), lds, idx->ctx->stream, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
