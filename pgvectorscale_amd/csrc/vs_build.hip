// vs_build.hip — batched Vamana construction over SBQ codes on the GPU (a "next" row: index build is
// NOT the reference's search path, but a device-resident index of 1M..50M nodes cannot be manufactured any other way
// inside a benchmark run).  It is the batch-parallel counterpart of
//   Graph::insert / insert_internal        AM/graph/mod.rs:637-717
//   greedy_search_for_build                AM/graph/mod.rs:285-327   (k_search<BUILD=true>)
//   add_neighbors + prune_neighbors        AM/graph/mod.rs:212-266,392-488 (alpha ladder 1.0, 1.2 .. max_alpha)
//   update_back_pointer                    AM/graph/mod.rs:719-735
// with SBQ Hamming distances between nodes exactly like SbqNodeDistanceMeasure (AM/sbq/mod.rs:161-190).
// Nodes are inserted in heap order; node 0 is the default start node (first inserted node, as in the reference).
// Batches: sizes double until `batch_max`; every node of a batch searches the graph of all previous batches, its
// out-edges are pruned, then back-edges are grouped per target (radix sort) and each target's list is re-pruned once.
// The result is deterministic for a given (codes, parameters); it is not claimed to be edge-identical to the
// reference's sequential build (which is itself HashSet-order dependent, AM/graph/mod.rs:317-326).
// Labeled vector sets (vs_index_set_labels before the build): as in Graph::insert (AM/graph/mod.rs:637-662) every node is
// inserted twice — from the start nodes of its labels with the label filter on, then from the default start node without
// it, the second pass merging into the list the first one left — a node is the start node of every label it is the first
// to carry (update_start_nodes, AM/graph/mod.rs:490-531), and prune_neighbors only lets an existing neighbor occlude a
// candidate when it carries every label the candidate shares with the point (contains_intersection,
// AM/labels/mod.rs:85-111, used at AM/graph/mod.rs:442-456).
#include <cstdlib>
#include <map>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "vs_internal.h"

#define WAVE 64

__device__ __forceinline__ uint32_t ham_words(const uint64_t* a, const uint64_t* b, uint32_t stride) {
    uint32_t acc = 0;
    if (stride == 24) {  // 768 x 2 bit / 1536 x 1 bit: fully unrolled so the 24 loads are in flight together
        ulonglong2 x[12], y[12];
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            x[t] = *reinterpret_cast<const ulonglong2*>(a + 2 * t);
            y[t] = *reinterpret_cast<const ulonglong2*>(b + 2 * t);
        }
#pragma unroll
        for (int t = 0; t < 12; ++t) acc += (uint32_t)__popcll(x[t].x ^ y[t].x) + (uint32_t)__popcll(x[t].y ^ y[t].y);
        return acc;
    }
    for (uint32_t w = 0; w < stride; w += 2) {
        const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(a + w);
        const ulonglong2 y = *reinterpret_cast<const ulonglong2*>(b + w);
        acc += (uint32_t)__popcll(x.x ^ y.x) + (uint32_t)__popcll(x.y ^ y.y);
    }
    return acc;
}

// copy the codes of C candidates into LDS: all 16-byte chunks of all rows are spread over the wave, 4 loads in flight per
// lane (a row-at-a-time loop pays one HBM latency per candidate, which dominated the build)
__device__ __forceinline__ void stage_codes(uint64_t* ccode, const uint64_t* __restrict__ codes, const uint32_t* cid,
                                            uint32_t C, uint32_t stride, int lane) {
    const uint32_t cpr = stride >> 1;  // 16-byte chunks per row (stride is even)
    const uint32_t total = C * cpr;
    for (uint32_t base = 0; base < total; base += 4 * WAVE) {
        ulonglong2 v[4];
        uint32_t dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t idx = base + (uint32_t)u * WAVE + (uint32_t)lane;
            dst[u] = 0xFFFFFFFFu;
            if (idx < total) {
                const uint32_t j = idx / cpr, w = 2u * (idx - j * cpr);
                v[u] = *reinterpret_cast<const ulonglong2*>(codes + (size_t)cid[j] * stride + w);
                dst[u] = j * stride + w;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] != 0xFFFFFFFFu) *reinterpret_cast<ulonglong2*>(ccode + dst[u]) = v[u];
    }
}

// contains_intersection(existing, candidate, point) for label sets relative to ONE point: bit t of a node's mask says that
// the point's t-th label is in the node's set, so  (candidate ∩ point) ⊆ existing  <=>  (mask[candidate] & ~mask[existing]) == 0.
// (a point carries at most 64 labels: vs_build_graph checks)
__device__ __forceinline__ uint64_t label_pmask(const uint32_t* __restrict__ label_off, const int16_t* __restrict__ label_val,
                                                uint32_t of, uint32_t node) {
    const uint32_t ob = label_off[of], oe = label_off[of + 1];
    uint32_t j = label_off[node];
    const uint32_t je = label_off[node + 1];
    uint64_t m = 0;
    for (uint32_t t = ob; t < oe && j < je;) {
        const int16_t x = label_val[t], y = label_val[j];
        if (x == y) {
            m |= 1ull << (t - ob);
            ++t;
            ++j;
        } else if (x < y) {
            ++t;
        } else {
            ++j;
        }
    }
    return m;
}

// prune_neighbors for one node by one wave.  cand_id/cand_d: C candidates sorted ascending by (distance, id)
// (LDS).  ccode: optional LDS copy of the candidate codes [C][stride] (nullptr => read codes from global).
// Writes up to R selected candidate *positions* into sel[] (LDS) and returns their number.
__device__ uint32_t wave_prune(const uint32_t* cand_id, const uint32_t* cand_d, uint32_t C, const uint64_t* ccode,
                               const uint64_t* __restrict__ codes, uint32_t stride, uint32_t R, float max_alpha,
                               float* maxf /*LDS [C]*/, uint32_t* sel /*LDS [R]*/, int lane,
                               const uint64_t* pm = nullptr /*LDS [C] label masks (label_pmask) or nullptr*/) {
    const float FMAX = 3.0e38f;
    if (ccode && stride == 24 && C <= WAVE) {
        // register form of the same loop (768 x 2 bit / 1536 x 1 bit codes, at most one candidate per lane): lane j keeps
        // candidate j's code, distance and max-factor in registers; only the selected candidate's code is read from LDS
        // (a broadcast read).  Same arithmetic in the same order as the general loop below.
        const bool mine_ok = (uint32_t)lane < C;
        ulonglong2 mine[12];
#pragma unroll
        for (int t = 0; t < 12; ++t)
            mine[t] = mine_ok ? *reinterpret_cast<const ulonglong2*>(ccode + (size_t)lane * 24 + 2 * t) : make_ulonglong2(0, 0);
        const uint32_t myd = mine_ok ? cand_d[lane] : 0u;
        const uint64_t mypm = (pm && mine_ok) ? pm[lane] : 0ull;
        float mymax = 0.0f;
        uint32_t nres = 0;
        float alpha = 1.0f;
        while (alpha <= max_alpha && nres < R) {
            for (uint32_t i = 0; i < C && nres < R; ++i) {
                const float mf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mymax), (int)i));
                if (mf > alpha) continue;
                if ((uint32_t)lane == i) mymax = FMAX;
                if (lane == 0) sel[nres] = i;
                nres++;
                // "Does it contain essential labels?" (AM/graph/mod.rs:442-456)
                const bool essential = pm ? (mypm & ~pm[i]) != 0 : false;
                if ((uint32_t)lane > i && mine_ok && !(mymax > max_alpha) && !essential) {
                    const uint64_t* ci = ccode + (size_t)i * 24;
                    uint32_t dij = 0;
#pragma unroll
                    for (int t = 0; t < 12; ++t) {
                        const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(ci + 2 * t);
                        dij += (uint32_t)__popcll(mine[t].x ^ c.x) + (uint32_t)__popcll(mine[t].y ^ c.y);
                    }
                    float factor;
                    if (dij == 0) factor = myd == 0 ? 1.0f : FMAX;
                    else factor = (float)myd / (float)dij;
                    mymax = fmaxf(mymax, factor);
                }
            }
            alpha *= 1.2f;
        }
        __syncthreads();
        return nres;
    }
    for (uint32_t j = lane; j < C; j += WAVE) maxf[j] = 0.0f;
    __syncthreads();
    uint32_t nres = 0;
    float alpha = 1.0f;
    while (alpha <= max_alpha && nres < R) {
        for (uint32_t i = 0; i < C && nres < R; ++i) {
            float mf = maxf[i];
            if (mf > alpha) continue;
            __syncthreads();
            if (lane == 0) {
                maxf[i] = FMAX;
                sel[nres] = i;
            }
            nres++;
            const uint64_t* ci = ccode ? ccode + (size_t)i * stride : codes + (size_t)cand_id[i] * stride;
            for (uint32_t j = i + 1 + lane; j < C; j += WAVE) {
                float mj = maxf[j];
                if (mj > max_alpha) continue;
                if (pm && (pm[j] & ~pm[i]) != 0) continue;  // "Does it contain essential labels?" (AM/graph/mod.rs:442-456)
                const uint64_t* cj = ccode ? ccode + (size_t)j * stride : codes + (size_t)cand_id[j] * stride;
                uint32_t dij = ham_words(cj, ci, stride);
                float factor;
                if (dij == 0) factor = cand_d[j] == 0 ? 1.0f : FMAX;
                else factor = (float)cand_d[j] / (float)dij;
                maxf[j] = fmaxf(mj, factor);
            }
            __syncthreads();
        }
        alpha *= 1.2f;
    }
    __syncthreads();
    return nres;
}

// ---- out-edges of the new nodes of one batch ------------------------------------------------------------------
// one wave per new node p = b0 + blockIdx.x.  Input: its visited list (sorted) from k_search<BUILD>.
__global__ __launch_bounds__(WAVE) void k_build_prune_new(const uint64_t* __restrict__ codes, uint32_t stride,
                                                          uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R,
                                                          float max_alpha, uint32_t b0, uint32_t bn,
                                                          const uint32_t* __restrict__ vis_ids,
                                                          const uint32_t* __restrict__ vis_d,
                                                          const uint32_t* __restrict__ vis_cnt, uint32_t vmax,
                                                          uint32_t use_lds_codes, uint64_t* __restrict__ edge_q,
                                                          uint64_t* __restrict__ edge_pd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= bn) return;
    const uint32_t p = b0 + b;
    uint32_t C = min(vis_cnt[b], vmax);
    uint32_t* cid = reinterpret_cast<uint32_t*>(smem);
    uint32_t* cd = cid + vmax;
    float* maxf = reinterpret_cast<float*>(cd + vmax);
    uint32_t* sel = reinterpret_cast<uint32_t*>(maxf + vmax);
    uint64_t* ccode = reinterpret_cast<uint64_t*>(sel + round_up_u32(R, 4));
    for (uint32_t j = lane; j < C; j += WAVE) {
        cid[j] = vis_ids[(size_t)b * vmax + j];
        cd[j] = vis_d[(size_t)b * vmax + j];
    }
    __syncthreads();
    if (use_lds_codes) {
        stage_codes(ccode, codes, cid, C, stride, lane);
        __syncthreads();
    }
    uint32_t nres;
    if (C <= R) {  // Graph::add_neighbors prunes only a candidate list longer than num_neighbors (AM/graph/mod.rs:243-256)
        for (uint32_t t = lane; t < C; t += WAVE) sel[t] = t;
        nres = C;
        __syncthreads();
    } else {
        nres = wave_prune(cid, cd, C, use_lds_codes ? ccode : nullptr, codes, stride, R, max_alpha, maxf, sel, lane);
    }
    uint32_t* row = nbrs + (size_t)p * nbr_stride;
    for (uint32_t t = lane; t < nbr_stride; t += WAVE) row[t] = t < nres ? cid[sel[t]] : VS_INVALID_NODE;
    // back-edge requests (q <- p, d)
    for (uint32_t t = lane; t < R; t += WAVE) {
        size_t e = (size_t)b * R + t;
        if (t < nres) {
            edge_q[e] = ((uint64_t)cid[sel[t]] << 32) | cd[sel[t]];  // sort key: target, then distance (closest requests first)
            edge_pd[e] = ((uint64_t)cd[sel[t]] << 32) | p;
        } else {
            edge_q[e] = ~0ull;
            edge_pd[e] = 0;
        }
    }
}

// segment heads of the sorted back-edge list
__global__ void k_seg_heads(const uint64_t* __restrict__ q_sorted, uint32_t ne, uint32_t* __restrict__ seg_start,
                            uint32_t* __restrict__ nseg) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    uint32_t q = (uint32_t)(q_sorted[i] >> 32);
    if (q == VS_INVALID_NODE) return;
    if (i == 0 || (uint32_t)(q_sorted[i - 1] >> 32) != q) seg_start[atomicAdd(nseg, 1u)] = i;
}

// in-LDS bitonic sort of u64 keys (n padded to pow2 with ~0)
__device__ void wave_bitonic_sort(uint64_t* keys, uint32_t npow2, int lane) {
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = lane; i < npow2; i += WAVE) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = keys[i], b = keys[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- out-edges of the new nodes of one batch, labeled vector sets -------------------------------------------------
// Same job as k_build_prune_new with what Graph::insert does for a labeled vector (AM/graph/mod.rs:637-662, 212-266): the
// node itself is dropped from its candidates (it is the start node of a label it is the first to carry), the second pass
// (merge_existing) adds the neighbors the filtered pass left in the row, and pruning uses the label rule.  Candidates are
// re-sorted by (distance, id) after the merge.  cap = power of two >= vmax + R.
__global__ __launch_bounds__(WAVE) void k_build_prune_merge(const uint64_t* __restrict__ codes, uint32_t stride,
                                                            uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R,
                                                            float max_alpha, uint32_t b0, uint32_t bn,
                                                            const uint32_t* __restrict__ vis_ids,
                                                            const uint32_t* __restrict__ vis_d,
                                                            const uint32_t* __restrict__ vis_cnt, uint32_t vmax, uint32_t cap,
                                                            uint32_t use_lds_codes, uint32_t merge_existing,
                                                            const uint32_t* __restrict__ label_off,
                                                            const int16_t* __restrict__ label_val,
                                                            uint64_t* __restrict__ edge_q, uint64_t* __restrict__ edge_pd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= bn) return;
    const uint32_t p = b0 + b;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);  // [cap]
    uint64_t* pm = keys + cap;                            // [cap]
    uint32_t* cid = reinterpret_cast<uint32_t*>(pm + cap);
    uint32_t* cd = cid + cap;
    float* maxf = reinterpret_cast<float*>(cd + cap);
    uint32_t* sel = reinterpret_cast<uint32_t*>(maxf + cap);
    uint64_t* ccode = reinterpret_cast<uint64_t*>(sel + round_up_u32(R, 4));
    const uint32_t C0 = min(vis_cnt[b], vmax);
    for (uint32_t t = lane; t < cap; t += WAVE) {
        uint64_t key = ~0ull;
        if (t < C0) {
            const uint32_t id = vis_ids[(size_t)b * vmax + t];
            if (id != p) key = ((uint64_t)vis_d[(size_t)b * vmax + t] << 32) | id;  // "remove myself"
        }
        keys[t] = key;
    }
    __syncthreads();
    uint32_t* row = nbrs + (size_t)p * nbr_stride;
    if (merge_existing) {  // add_neighbors: the list so far + the new candidates, each id once
        const uint64_t* cp = codes + (size_t)p * stride;
        for (uint32_t t = lane; t < R; t += WAVE) {
            const uint32_t id = row[t];
            if (id == VS_INVALID_NODE || id == p) continue;
            bool dup = false;
            for (uint32_t u = 0; u < C0 && !dup; ++u) dup = (uint32_t)keys[u] == id && keys[u] != ~0ull;
            if (!dup) keys[C0 + t] = ((uint64_t)ham_words(codes + (size_t)id * stride, cp, stride) << 32) | id;
        }
        __syncthreads();
    }
    wave_bitonic_sort(keys, cap, lane);
    uint32_t T = 0;
    for (uint32_t t0 = 0; t0 < cap; t0 += WAVE) T += (uint32_t)__popcll(__ballot(keys[t0 + lane] != ~0ull));
    for (uint32_t t = lane; t < T; t += WAVE) {
        cid[t] = (uint32_t)keys[t];
        cd[t] = (uint32_t)(keys[t] >> 32);
        pm[t] = label_off ? label_pmask(label_off, label_val, p, (uint32_t)keys[t]) : 0ull;
    }
    __syncthreads();
    if (use_lds_codes) {
        stage_codes(ccode, codes, cid, T, stride, lane);
        __syncthreads();
    }
    uint32_t nres;
    if (T <= R) {  // Graph::add_neighbors prunes only a candidate list longer than num_neighbors (AM/graph/mod.rs:243-256)
        for (uint32_t t = lane; t < T; t += WAVE) sel[t] = t;
        nres = T;
        __syncthreads();
    } else {
        nres = wave_prune(cid, cd, T, use_lds_codes ? ccode : nullptr, codes, stride, R, max_alpha, maxf, sel, lane,
                          label_off ? pm : nullptr);
    }
    for (uint32_t t = lane; t < nbr_stride; t += WAVE) row[t] = t < nres ? cid[sel[t]] : VS_INVALID_NODE;
    for (uint32_t t = lane; t < R; t += WAVE) {  // back-edge requests (q <- p, d)
        const size_t e = (size_t)b * R + t;
        if (t < nres) {
            edge_q[e] = ((uint64_t)cid[sel[t]] << 32) | cd[sel[t]];  // sort key: target, then distance (closest requests first)
            edge_pd[e] = ((uint64_t)cd[sel[t]] << 32) | p;
        } else {
            edge_q[e] = ~0ull;
            edge_pd[e] = 0;
        }
    }
}

// ---- back-edges: one wave per target node q ---------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_build_backedges(const uint64_t* __restrict__ codes, uint32_t stride,
                                                          uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R,
                                                          float max_alpha, const uint64_t* __restrict__ q_sorted,
                                                          const uint64_t* __restrict__ pd_sorted, uint32_t ne,
                                                          const uint32_t* __restrict__ seg_start,
                                                          const uint32_t* __restrict__ nseg_p, uint32_t cmax,
                                                          uint32_t use_lds_codes, const uint32_t* __restrict__ label_off,
                                                          const int16_t* __restrict__ label_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t nseg = *nseg_p;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);  // [cmax] (pow2)
    uint64_t* pm = keys + cmax;                           // [cmax] label masks relative to the target (labeled sets only)
    uint32_t* cid = reinterpret_cast<uint32_t*>(pm + cmax);
    uint32_t* cd = cid + cmax;
    float* maxf = reinterpret_cast<float*>(cd + cmax);
    uint32_t* sel = reinterpret_cast<uint32_t*>(maxf + cmax);
    uint64_t* ccode = reinterpret_cast<uint64_t*>(sel + round_up_u32(R, 4));
    for (uint32_t sidx = blockIdx.x; sidx < nseg; sidx += gridDim.x) {
        const uint32_t e0 = seg_start[sidx];
        const uint32_t q = (uint32_t)(q_sorted[e0] >> 32);
        uint32_t* row = nbrs + (size_t)q * nbr_stride;
        // existing degree
        uint32_t deg = 0;
        for (uint32_t c0 = 0; c0 < R; c0 += WAVE) {
            uint32_t t = c0 + lane;
            uint32_t v = t < R ? row[t] : VS_INVALID_NODE;
            uint64_t inval = __ballot(v == VS_INVALID_NODE);
            if (inval) {
                deg = c0 + (uint32_t)__builtin_ctzll(inval);
                break;
            }
            deg = c0 + WAVE;
        }
        deg = min(deg, R);
        // segment length
        uint32_t m = 0;
        while (e0 + m < ne && (uint32_t)(q_sorted[e0 + m] >> 32) == q) ++m;  // uniform scalar loop (short)
        // add_neighbors takes every id once (AM/graph/mod.rs:227-235): a source that already is in the list — the second
        // pass of a labeled set asks again for the back-edges its first pass created — is dropped
        auto in_row = [&](uint32_t pid) -> bool {
            for (uint32_t u = 0; u < deg; ++u)
                if (row[u] == pid) return true;
            return false;
        };
        if (deg + m <= R) {  // room: append in (sorted) order
            uint32_t w = deg;
            for (uint32_t base = 0; base < m; base += WAVE) {
                const uint32_t t = base + lane;
                const uint32_t pid = t < m ? (uint32_t)pd_sorted[e0 + t] : VS_INVALID_NODE;
                const bool ok = t < m && !in_row(pid);
                const uint64_t okm = __ballot(ok);
                if (ok) row[w + (uint32_t)__popcll(okm & ((1ull << lane) - 1ull))] = pid;
                w += (uint32_t)__popcll(okm);
            }
            continue;
        }
        // candidates = existing neighbors (distance computed) + new sources.  A hub can receive more requests in one batch than
        // the candidate array holds: the requests of a target are sorted by distance, so the ones kept are its closest
        const uint32_t take_new = min(m, cmax - deg);
        const uint32_t T = deg + take_new;
        uint32_t np2 = 1;
        while (np2 < T) np2 <<= 1;
        const uint64_t* cq = codes + (size_t)q * stride;
        for (uint32_t t = lane; t < np2; t += WAVE) {
            uint64_t key = ~0ull;
            if (t < deg) {
                uint32_t id = row[t];
                key = ((uint64_t)ham_words(codes + (size_t)id * stride, cq, stride) << 32) | id;
            } else if (t < T) {
                uint64_t pd = pd_sorted[e0 + (t - deg)];
                if (!in_row((uint32_t)pd)) key = pd;  // (dist << 32) | p
            }
            keys[t] = key;
        }
        __syncthreads();
        wave_bitonic_sort(keys, np2, lane);
        uint32_t Tv = 0;  // candidates left once the repeated sources are gone (they sorted to the end)
        for (uint32_t t0 = 0; t0 < np2; t0 += WAVE) Tv += (uint32_t)__popcll(__ballot(t0 + lane < np2 && keys[t0 + lane] != ~0ull));
        for (uint32_t t = lane; t < Tv; t += WAVE) {
            cid[t] = (uint32_t)keys[t];
            cd[t] = (uint32_t)(keys[t] >> 32);
            if (label_off) pm[t] = label_pmask(label_off, label_val, q, (uint32_t)keys[t]);  // add_neighbors(q, from_labels = q's)
        }
        __syncthreads();
        if (use_lds_codes) {
            stage_codes(ccode, codes, cid, Tv, stride, lane);
            __syncthreads();
        }
        uint32_t nres;
        if (Tv <= R) {  // (only repeated sources made the list look too long)
            for (uint32_t t = lane; t < Tv; t += WAVE) sel[t] = t;
            nres = Tv;
            __syncthreads();
        } else {
            nres = wave_prune(cid, cd, Tv, use_lds_codes ? ccode : nullptr, codes, stride, R, max_alpha, maxf, sel, lane,
                              label_off ? pm : nullptr);
        }
        for (uint32_t t = lane; t < nbr_stride; t += WAVE) row[t] = t < nres ? cid[sel[t]] : VS_INVALID_NODE;
        __syncthreads();
    }
}

// ---- repair pass: nodes no scan can reach ---------------------------------------------------------------------------
// Nodes of one batch do not see each other, so of two (near-)identical vectors that arrive together one can lose every
// back-edge to the other (the target prunes it as covered) without gaining the edge between the two that sequential
// insertion gives; a node that cannot be reached from the start node is never returned by a scan.  After the last batch
// the reachable set is computed (level-synchronous sweeps over the neighbor array) and every node outside it is given a
// slot in the list of its closest reachable out-neighbor (vs_build_graph).
__global__ void k_reach_sweep(const uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R, uint32_t n,
                              uint8_t* __restrict__ reached, uint32_t level, uint32_t* __restrict__ changed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * R) return;
    if (reached[i / R] != level) return;  // reached[] holds 1 + the BFS level: only the frontier expands
    const uint32_t v = nbrs[(i / R) * nbr_stride + (i % R)];
    if (v != VS_INVALID_NODE && !reached[v]) {
        reached[v] = (uint8_t)(level + 1);
        *changed = 1;
    }
}

// in-edges that come from reachable nodes of a strictly LOWER BFS level than their target: such a source is reached on a
// path that does not pass through the target, so a target that keeps one of these edges stays reachable whatever else it loses
__global__ void k_count_reached_sources(const uint32_t* __restrict__ nbrs, uint32_t nbr_stride, uint32_t R, uint32_t n,
                                        const uint8_t* __restrict__ reached, uint32_t* __restrict__ indeg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * R) return;
    const uint8_t ls = reached[i / R];
    if (!ls) return;
    const uint32_t v = nbrs[(i / R) * nbr_stride + (i % R)];
    if (v != VS_INVALID_NODE && reached[v] > ls) atomicAdd(&indeg[v], 1u);
}

struct BuildBufs {
    uint32_t *vis_ids = nullptr, *vis_d = nullptr, *vis_cnt = nullptr, *stats = nullptr, *status = nullptr;
    uint32_t* hash = nullptr;
    uint64_t* heap_g = nullptr;
    uint32_t *seg_start = nullptr, *nseg = nullptr;
    uint64_t *edge_q = nullptr, *edge_q_sorted = nullptr;  // (target << 32) | distance
    uint64_t *edge_pd = nullptr, *edge_pd_sorted = nullptr;
    void* cub_tmp = nullptr;
    size_t cub_bytes = 0;
    uint32_t *f_ghash = nullptr, *f_heap = nullptr, *f_pool = nullptr;  // fast-kernel overflow table / heap spill / pool counter
    uint8_t* mark = nullptr;                                             // repair pass: node is reachable from the start node
    void free_all() {
        void* ps[] = {vis_ids, vis_d, vis_cnt, stats, status, hash, heap_g, edge_q, edge_q_sorted, seg_start, nseg,
                      edge_pd, edge_pd_sorted, cub_tmp, f_ghash, f_heap, f_pool, mark};
        for (void* p : ps)
            if (p) (void)hipFree(p);
    }
};

static int build_graph_impl(vs_index* ix, uint32_t L, double max_alpha_d, uint32_t batch_max, BuildBufs& B) {
    vs_ctx* c = ix->ctx;
    hipStream_t st = c->stream;
    const uint32_t n = ix->d.n, R = ix->d.num_neighbors, stride = ix->code_stride;
    const float max_alpha = (float)max_alpha_d;
    VS_HIP(hipMemsetAsync(ix->nbrs, 0xFF, (size_t)std::max(n, 1u) * ix->nbr_stride * 4, st));
    if (n == 0) {
        ix->d.default_start = VS_INVALID_NODE;
        return VS_OK;
    }
    ix->d.default_start = 0;
    const bool labeled = ix->label_off != nullptr;
    if (labeled) {
        // update_start_nodes (AM/graph/mod.rs:490-531): a node is the start node of every label it is the first to carry;
        // nodes arrive in id order, so that is the smallest id per label
        std::vector<uint32_t> off((size_t)n + 1);
        VS_HIP(hipMemcpy(off.data(), ix->label_off, off.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int16_t> val(std::max<size_t>(off[n], 1));
        if (off[n]) VS_HIP(hipMemcpy(val.data(), ix->label_val, (size_t)off[n] * 2, hipMemcpyDeviceToHost));
        std::map<int16_t, uint32_t> first;
        for (uint32_t i = 0; i < n; ++i) {
            VS_REQUIRE(off[i + 1] - off[i] <= 64, "vs_build_graph: node %u carries more than 64 labels", i);
            for (uint32_t j = off[i]; j < off[i + 1]; ++j) first.emplace(val[j], i);
        }
        std::vector<int16_t> sl;
        std::vector<uint32_t> sn;
        for (const auto& kv : first) {
            sl.push_back(kv.first);
            sn.push_back(kv.second);
        }
        VS_TRY(vs_index_set_start_nodes(ix, 0, sl.data(), sn.data(), (uint32_t)sl.size()));
    }
    if (batch_max == 0) batch_max = std::min<uint32_t>(65536, std::max<uint32_t>(1024, n / 64));
    // capacities of the build-mode search
    uint32_t vmax = std::max<uint32_t>(round_up_u32(3 * L + 64, 64), 128);      // visited list cap (candidates of prune)
    uint32_t hl = 512, lh = 0;                                       // LDS-resident parts of the search state
    uint32_t hcap = (2 * L + 64) * R;                                // heap capacity (global spill beyond hl)
    uint32_t hashcap = std::max<uint32_t>(next_pow2_u32(2ull * hcap), 256);
    uint32_t cmax = 1;
    while (cmax < R + 128) cmax <<= 1;  // back-edge candidate cap (pow2, >= R + new sources kept)
    const size_t code_bytes = (size_t)stride * 8;
    const uint32_t use_lds_new = (vmax * code_bytes + vmax * 12 + R * 4 + 64 <= 150 * 1024) ? 1 : 0;
    const uint32_t use_lds_back = (cmax * code_bytes + cmax * 28 + R * 4 + 64 <= 150 * 1024) ? 1 : 0;
    const size_t lds_new = (size_t)vmax * 12 + round_up_u32(R, 4) * 4 + (use_lds_new ? vmax * code_bytes : 0) + 64;
    const size_t lds_back = (size_t)cmax * 28 + round_up_u32(R, 4) * 4 + (use_lds_back ? cmax * code_bytes : 0) + 64;
    const uint32_t mcap = next_pow2_u32((uint64_t)vmax + R);  // candidates of k_build_prune_merge
    const uint32_t use_lds_merge = (mcap * code_bytes + mcap * 28 + R * 4 + 64 <= 150 * 1024) ? 1 : 0;
    const size_t lds_merge = (size_t)mcap * 28 + round_up_u32(R, 4) * 4 + (use_lds_merge ? mcap * code_bytes : 0) + 64;
    VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_prune_new),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_backedges),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_prune_merge),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    const size_t bm = batch_max;
    VS_HIP(hipMalloc(&B.vis_ids, bm * vmax * 4));
    VS_HIP(hipMalloc(&B.vis_d, bm * vmax * 4));
    VS_HIP(hipMalloc(&B.vis_cnt, bm * 4));
    VS_HIP(hipMalloc(&B.stats, bm * ST_N * 4));
    VS_HIP(hipMalloc(&B.status, bm * 4));
    VS_HIP(hipMalloc(&B.edge_q, bm * R * 8));
    VS_HIP(hipMalloc(&B.edge_q_sorted, bm * R * 8));
    VS_HIP(hipMalloc(&B.edge_pd, bm * R * 8));
    VS_HIP(hipMalloc(&B.edge_pd_sorted, bm * R * 8));
    VS_HIP(hipMalloc(&B.seg_start, bm * R * 4));
    VS_HIP(hipMalloc(&B.nseg, 4));
    VS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, B.cub_bytes, B.edge_q, B.edge_q_sorted, B.edge_pd,
                                              B.edge_pd_sorted, (int)(bm * R), 0, 64, st));
    VS_HIP(hipMalloc(&B.cub_tmp, B.cub_bytes + 16));
    size_t hash_alloc = 0, ids_alloc = 0;

    // The searches of a batch run on the LDS-resident kernel (vs_search_fast.hip, BUILD variant); the general kernel only
    // re-runs the scans that outgrow it.  Same operating-point rule as for queries: dedup table in LDS for small graphs,
    // table-less (global table, high occupancy) once a search inserts more ids than an LDS table should hold.
    FastLaunch f{};
    bool use_fast = vs_opt_get("VS_BUILD_FAST") ? atoi(vs_opt_get("VS_BUILD_FAST")) != 0 : true;
    {
        const uint64_t nbits = (uint64_t)ix->d.dim_index * ix->d.bits;
        const uint32_t typ_ins = (L + L / 4 + 16) * std::min<uint32_t>(R, 16);
        const bool lds_table = n <= 4000000u && typ_ins <= 3072;
        f.L = L;
        f.M = vmax;
        f.hl = 1023;
        f.hcap = std::max(hcap, f.hl);  // small L * R: the whole heap fits the LDS part (the launcher wants hcap >= hl)
        f.gstride = round_up_u32(f.hcap - f.hl + 2, 2);
        f.lh = lds_table ? std::max<uint32_t>(round_up_u32(typ_ins, 64), 256) : 0;
        f.gcap = next_pow2_u32(std::max<uint32_t>(4 * typ_ins, 1024));
        f.sb = 0;
        while ((1ull << f.sb) < (uint64_t)f.lh + f.gcap) f.sb++;
        f.vr = 0;
        f.vcap = vmax + 64;
        f.minw = 1;
        f.build = 1;
        if (nbits >= (1ull << (32 - f.sb)) || fast_lds_bytes(ix, f) > 64 * 1024) use_fast = false;
        if (use_fast) {
            VS_HIP(hipMalloc(&B.f_ghash, bm * f.gcap * 4));
            VS_HIP(hipMalloc(&B.f_heap, bm * f.gstride * 4));
            VS_HIP(hipMalloc(&B.f_pool, 64));
        }
    }

    // one batch: searches for the new nodes b0 .. b0 + bn - 1, their out-edges, the back-edges
    // filtered: the pass from the label start nodes with the label filter (labeled sets only); otherwise from the default
    // start node.  A labeled set's unfiltered pass merges into the rows the filtered pass wrote.
    auto insert_batch = [&](uint32_t b0, uint32_t bn, bool filtered) -> int {
        const uint64_t* qcodes = ix->codes + (size_t)b0 * stride;
        // label keys of the searches = the new nodes' own label sets (CSR offsets are absolute into label_val)
        const int16_t* ql = filtered ? ix->label_val : nullptr;
        const uint32_t* qlo = filtered ? ix->label_off + b0 : nullptr;
        for (int attempt = 0;; ++attempt) {
            if ((size_t)bn * hashcap * 4 > hash_alloc || !B.hash) {
                if (B.hash) VS_HIP(hipFree(B.hash));
                B.hash = nullptr;
                hash_alloc = (size_t)batch_max * hashcap * 4;
                VS_HIP(hipMalloc(&B.hash, hash_alloc));
            }
            const size_t hg = hcap > hl ? hcap - hl : 0;
            if ((size_t)bn * hg * 8 > ids_alloc || !B.heap_g) {
                if (B.heap_g) VS_HIP(hipFree(B.heap_g));
                B.heap_g = nullptr;
                ids_alloc = std::max<size_t>((size_t)batch_max * hg * 8, 16);
                VS_HIP(hipMalloc(&B.heap_g, ids_alloc));
            }
            SearchLaunch s;
            s.nq = bn;
            s.L = L;
            s.M = vmax;
            s.hl = hl;
            s.hcap = hcap;
            s.vcap = vmax + 64;
            s.lh = lh;
            s.hashcap = hashcap;
            s.g0 = std::min<uint32_t>(4096, hashcap);  // first level of the dedup ladder (small L * R: the whole table)
            s.qcodes = qcodes;
            s.qlabels = ql;
            s.qlabel_off = qlo;
            s.heap_g = B.heap_g;
            s.hash = B.hash;
            s.out_ids = B.vis_ids;
            s.out_ham = B.vis_d;
            s.out_cnt = B.vis_cnt;
            s.stats = B.stats;
            s.status = B.status;
            if (use_fast && attempt == 0) {
                f.nq = bn;
                f.hcap = std::max(std::min(hcap, f.hl + f.gstride - 2), f.hl);  // (the spill area was sized before the loop)
                f.qcodes = s.qcodes;
                f.qlabels = ql;
                f.qlabel_off = qlo;
                f.heap_g = B.f_heap;
                f.ghash = B.f_ghash;
                f.pool_counter = B.f_pool;
                f.pool_slots = bn;
                f.out_ids = B.vis_ids;
                f.out_ham = B.vis_d;
                f.out_cnt = B.vis_cnt;
                f.stats = B.stats;
                f.status = B.status;
                VS_HIP(hipMemsetAsync(B.f_pool, 0, 64, st));
                VS_TRY(launch_search_fast(ix, f));
            }
            // after the fast kernel (or a failed attempt) only the scans whose status is non-zero are (re)run
            s.only_failed = (use_fast || attempt > 0) ? 1u : 0u;
            VS_TRY(launch_search(ix, s, true));
            std::vector<uint32_t> status(bn);
            VS_HIP(hipMemcpyAsync(status.data(), B.status, (size_t)bn * 4, hipMemcpyDeviceToHost, st));
            VS_HIP(hipStreamSynchronize(st));
            uint32_t ovf = 0;
            for (uint32_t v : status) ovf |= v;
            if (!ovf) break;
            if (attempt >= 5) {
                vs_set_error("vs_build_graph: search structures overflowed (flags 0x%x)", ovf);
                return VS_ERR_CAPACITY;
            }
            if (ovf & OVF_VISITED) {
                // the visited list outgrew the prune candidate cap (the build-mode fast kernel truncates to the closest
                // entries itself; this is the general kernel's ring at vcap = vmax + 64): not a combination the LDS holds
                vs_set_error("vs_build_graph: visited list overflow (search_list_size too large for LDS)");
                return VS_ERR_CAPACITY;
            }
            if (ovf & OVF_HEAP) hcap *= 2;
            if (ovf & OVF_HASH) hashcap *= 2;
        }
        // out-edges of the new nodes + back-edge requests
        if (labeled)
            hipLaunchKernelGGL(k_build_prune_merge, dim3(bn), dim3(WAVE), lds_merge, st, ix->codes, stride, ix->nbrs,
                               ix->nbr_stride, R, max_alpha, b0, bn, B.vis_ids, B.vis_d, B.vis_cnt, vmax, mcap, use_lds_merge,
                               filtered ? 0u : 1u, ix->label_off, ix->label_val, B.edge_q, B.edge_pd);
        else
            hipLaunchKernelGGL(k_build_prune_new, dim3(bn), dim3(WAVE), lds_new, st, ix->codes, stride, ix->nbrs,
                               ix->nbr_stride, R, max_alpha, b0, bn, B.vis_ids, B.vis_d, B.vis_cnt, vmax, use_lds_new,
                               B.edge_q, B.edge_pd);
        VS_HIP(hipGetLastError());
        const uint32_t ne = bn * R;
        size_t tmp_bytes = B.cub_bytes;
        VS_HIP(hipcub::DeviceRadixSort::SortPairs(B.cub_tmp, tmp_bytes, B.edge_q, B.edge_q_sorted, B.edge_pd,
                                                  B.edge_pd_sorted, (int)ne, 0, 64, st));
        VS_HIP(hipMemsetAsync(B.nseg, 0, 4, st));
        hipLaunchKernelGGL(k_seg_heads, dim3((ne + 255) / 256), dim3(256), 0, st, B.edge_q_sorted, ne, B.seg_start, B.nseg);
        VS_HIP(hipGetLastError());
        uint32_t grid = std::min<uint32_t>(ne, 16384);
        hipLaunchKernelGGL(k_build_backedges, dim3(grid), dim3(WAVE), lds_back, st, ix->codes, stride, ix->nbrs,
                           ix->nbr_stride, R, max_alpha, B.edge_q_sorted, B.edge_pd_sorted, ne, B.seg_start, B.nseg, cmax,
                           use_lds_back, labeled ? ix->label_off : nullptr, labeled ? ix->label_val : nullptr);
        VS_HIP(hipGetLastError());
        return VS_OK;
    };

    uint32_t b0 = 1;  // node 0 is the start node and has no one to link to yet
    uint32_t bsz = 1;
    while (b0 < n) {
        const uint32_t bn = std::min<uint32_t>(std::min<uint32_t>(bsz, batch_max), n - b0);
        if (labeled) VS_TRY(insert_batch(b0, bn, true));  // Graph::insert: first with the label filter ...
        VS_TRY(insert_batch(b0, bn, false));              // ... then from the default start node without it
        b0 += bn;
        if (bsz < batch_max) bsz = std::min<uint32_t>(batch_max, bsz * 2);
    }
    VS_HIP(hipStreamSynchronize(st));

    // repair pass (see k_reach_sweep): rounds of { reachable set, in-edges for the nodes outside it } until a sweep finds every
    // node (at most eight); what the last sweep still could not reach is reported by vs_index_build_unreachable()
    const char* rep_env = vs_opt_get("VS_BUILD_REPAIR");
    if (n > 2 && !(rep_env && *rep_env == '0')) {
        VS_HIP(hipMalloc(&B.mark, (size_t)n + 8));  // n flags, then (4-byte aligned) the `changed` word
        uint32_t* d_changed = reinterpret_cast<uint32_t*>(B.mark + (((size_t)n + 3) & ~(size_t)3));
        std::vector<uint8_t> reached(n);
        std::vector<uint32_t> lost, indeg, row0(R);
        const size_t cells = (size_t)n * R;
        const dim3 cgrid((unsigned)((cells + 255) / 256));
        const uint32_t start = ix->d.default_start;
        ix->build_unreachable = 0;
        for (int round = 0; round < 8; ++round) {
            VS_HIP(hipMemsetAsync(B.mark, 0, (size_t)n + 8, st));
            const uint8_t one = 1;
            VS_HIP(hipMemcpyAsync(B.mark + start, &one, 1, hipMemcpyHostToDevice, st));
            bool converged = false;
            for (uint32_t level = 1; level < 255 && !converged; ++level) {  // one BFS level per sweep
                uint32_t changed = 0;
                VS_HIP(hipMemsetAsync(d_changed, 0, 4, st));
                hipLaunchKernelGGL(k_reach_sweep, cgrid, dim3(256), 0, st, ix->nbrs, ix->nbr_stride, R, n, B.mark, level, d_changed);
                VS_HIP(hipGetLastError());
                VS_HIP(hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, st));
                VS_HIP(hipStreamSynchronize(st));
                converged = changed == 0;
            }
            if (!converged) {  // more than 254 levels deep: not a graph this pass can judge
                ix->build_unreachable = 0xFFFFFFFFu;
                break;
            }
            VS_HIP(hipMemcpyAsync(reached.data(), B.mark, n, hipMemcpyDeviceToHost, st));
            VS_HIP(hipStreamSynchronize(st));
            lost.clear();
            for (uint32_t i = 0; i < n; ++i)
                if (!reached[i]) lost.push_back(i);
            ix->build_unreachable = (uint32_t)lost.size();  // (what the last completed sweep found; 0 when the loop ends here)
            if (lost.empty() || round == 7) break;
            // Each of them takes a slot in the list of its closest reachable out-neighbor — a free one, else that of the last
            // entry that keeps an in-edge from a node of a strictly lower BFS level (indeg[] counts only those: a lower-level
            // source is reached without passing through the entry, so the entry cannot be stranded by losing this edge —
            // counting every reachable source would let two nodes that only reach each other vouch for one another).  Rare
            // (none on the bench corpora), so this runs on the host, in node order; the sweep of the next round re-checks.
            uint32_t* d_indeg = nullptr;
            VS_HIP(hipMalloc(&d_indeg, (size_t)n * 4));
            indeg.assign(n, 0);
            int r = VS_OK;
            auto hip_ok = [&](hipError_t e) {
                if (r == VS_OK && e != hipSuccess) {
                    vs_set_error("vs_build_graph (repair): %s", hipGetErrorString(e));
                    r = VS_ERR_HIP;
                }
            };
            hip_ok(hipMemsetAsync(d_indeg, 0, (size_t)n * 4, st));
            hipLaunchKernelGGL(k_count_reached_sources, cgrid, dim3(256), 0, st, ix->nbrs, ix->nbr_stride, R, n, B.mark, d_indeg);
            hip_ok(hipGetLastError());
            hip_ok(hipMemcpyAsync(indeg.data(), d_indeg, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            hip_ok(hipStreamSynchronize(st));
            (void)hipFree(d_indeg);
            // the lists of the lost nodes, so that what becomes reachable through a node that was just given a way in is known
            // without another sweep
            std::vector<uint32_t> lrows(lost.size() * (size_t)R), lidx(n, 0xFFFFFFFFu), stack, level(n);
            for (uint32_t i = 0; i < n; ++i) level[i] = reached[i];  // 1 + BFS level (0: not reached); grows past 255 on the host
            for (size_t oi = 0; oi < lost.size() && r == VS_OK; ++oi) {
                lidx[lost[oi]] = (uint32_t)oi;
                hip_ok(hipMemcpy(&lrows[oi * R], ix->nbrs + (size_t)lost[oi] * ix->nbr_stride, (size_t)R * 4, hipMemcpyDeviceToHost));
            }
            for (bool progress = true; progress && r == VS_OK;) {
                progress = false;
                for (size_t oi = 0; oi < lost.size() && r == VS_OK; ++oi) {
                    const uint32_t x = lost[oi];
                    if (reached[x]) continue;
                    const uint32_t* rowx = &lrows[oi * R];
                    bool placed = false;
                    for (uint32_t c = 0; c < R && r == VS_OK && !placed && rowx[c] != VS_INVALID_NODE; ++c) {  // closest first
                        const uint32_t n0 = rowx[c];
                        if (!reached[n0]) continue;
                        uint32_t* rown = lidx[n0] != 0xFFFFFFFFu ? &lrows[(size_t)lidx[n0] * R] : row0.data();
                        if (rown == row0.data())
                            hip_ok(hipMemcpy(row0.data(), ix->nbrs + (size_t)n0 * ix->nbr_stride, (size_t)R * 4, hipMemcpyDeviceToHost));
                        if (r != VS_OK) break;
                        int slot = -1;
                        for (uint32_t t = 0; t < R && slot < 0; ++t)
                            if (rown[t] == VS_INVALID_NODE) slot = (int)t;
                        for (int t = (int)R - 1; t >= 0 && slot < 0; --t) {
                            const uint32_t y = rown[t];
                            const uint32_t mine = (reached[y] && level[n0] < level[y]) ? 1u : 0u;  // is n0 -> y one of the counted edges?
                            if (indeg[y] >= mine + 1u) slot = t;
                        }
                        if (slot < 0) continue;
                        if (rown[slot] != VS_INVALID_NODE && reached[rown[slot]] && level[n0] < level[rown[slot]]) indeg[rown[slot]]--;
                        rown[slot] = x;
                        level[x] = level[n0] + 1;
                        indeg[x]++;
                        placed = true;
                        hip_ok(hipMemcpy(ix->nbrs + (size_t)n0 * ix->nbr_stride, rown, (size_t)R * 4, hipMemcpyHostToDevice));
                    }
                    if (!placed) continue;
                    progress = true;
                    reached[x] = 1;
                    stack.assign(1, x);
                    while (!stack.empty()) {  // everything x leads to is reachable now
                        const uint32_t u = stack.back();
                        stack.pop_back();
                        const uint32_t* rowu = &lrows[(size_t)lidx[u] * R];
                        for (uint32_t t = 0; t < R && rowu[t] != VS_INVALID_NODE; ++t) {
                            const uint32_t v = rowu[t];
                            if (!reached[v]) {
                                reached[v] = 1;
                                level[v] = level[u] + 1;
                                indeg[v]++;
                                stack.push_back(v);
                            } else if (level[u] < level[v]) {
                                indeg[v]++;
                            }
                        }
                    }
                }
            }
            VS_TRY(r);
        }
    }
    return VS_OK;
}

static int vs_build_graph_impl(vs_index* ix, uint32_t search_list_size, double max_alpha, uint32_t batch_max, uint64_t seed) {
    (void)seed;
    VS_REQUIRE(ix, "vs_build_graph: index is NULL");
    VS_REQUIRE(search_list_size >= 1 && search_list_size <= 1000, "vs_build_graph: search_list_size outside [1,1000]");
    VS_REQUIRE(max_alpha >= 1.0 && max_alpha <= 5.0, "vs_build_graph: max_alpha outside [1,5]");
    VS_HIP(hipSetDevice(ix->ctx->device));
    BuildBufs B;
    ix->nbr_mask_valid = false;  // (the neighbor lists are about to change: what was derived from them is stale)
    int r = build_graph_impl(ix, search_list_size, max_alpha, batch_max, B);
    (void)hipStreamSynchronize(ix->ctx->stream);
    B.free_all();
    // the scan kernels rely on every neighbor list naming a node at most once (as vs_index_upload checks for staged
    // indexes): a list that does not is a defect of this builder, not something to search on
    if (r == VS_OK) r = vs_validate_graph(ix);
    return r;
}
extern "C" int vs_build_graph(vs_index* ix, uint32_t search_list_size, double max_alpha, uint32_t batch_max, uint64_t seed) {
    return vs_guard("vs_build_graph", [&] { return vs_build_graph_impl(ix, search_list_size, max_alpha, batch_max, seed); });
}

