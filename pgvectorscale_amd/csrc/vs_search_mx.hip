// vs_search_mx.hip — K3, "four scans per wave" form of the streaming beam search (VS_MX=1).
//
// Same semantics and citations as vs_search_fast.hip / vs_search.hip; what changes is the mapping.  k_search_fast gives a
// whole wave64 to one scan and, at the occupancy it reaches in the table-less regime, is bound by instruction issue:
// most of what a scan does between two memory phases (heap sift-up / sift-down, visited-list insert, loop control) is
// wave-uniform bookkeeping that occupies a 64-lane instruction slot per step.  Here a scan owns
// one DPP row of 16 lanes and a wave advances four scans in lockstep, so every such instruction does the bookkeeping of
// four scans; the phases that were already lane-parallel (dedup probes, code gather) keep one lane per item.
//
//   * candidate heap (BinaryHeap<Reverse<ListSearchNeighbor>>, AM/graph/mod.rs:75): same 4-byte entries
//     (hamming << sb | dedup slot) and the same LDS / spill split as k_search_fast; sift_up = lane r of the row compares
//     with the r-th ancestor (one read, one 16-bit ballot, one store; heaps of < 2^15 entries); the pushes of a visit run
//     against an LDS-staged copy of all their ancestors (one memory round trip per run); sift_down_to_bottom evaluates a
//     4-level subtree per round (both children of every node in one 8-byte read).  Rust std's array mechanics are
//     replayed exactly.
//   * dedup set: the per-scan global table of the table-less regime — private to its scan, so no atomics: buckets of four
//     slots (one 16-byte load per probe), plain stores, lanes that want the same bucket told apart by per-scan LDS rank
//     counters (vs_search_fast.hip has the reasoning); one lane per neighbor, all four steps of a list against one snapshot.
//   * visited list: sorted array in registers, entry i = lane i % 16 of register i / 16; insert / remove(0) are DPP row
//     shift (row_shr) with the carry between registers taken by a row rotate; remove(0) advances a head offset.
//   * distances: 4 lanes per code row, 4 rows per row-of-16 per step, GD steps in flight.
//   * the id of the next node to visit and the heap tid of the visited list's front are requested a step ahead.
//
// Scans whose state outgrows the kernel set the same status flags as in k_search_fast and are re-run by the general
// kernel.  Results (streams, Hamming distances, GreedySearchStats counters) are bit-identical to k_search_fast's.
#include <algorithm>
#include <cstdlib>

#include "vs_device.h"

#define MX_MAX_QLABELS 64
#define MX_G 16  // lanes per scan (one DPP row)
#define MX_STG 152  // staged heap words per scan during a run of pushes (MxHeap::push_run_staged)
#define MX_SURV 72  // survivor slots per scan (64 used; the stride keeps the rows' arrays in different LDS banks)
#define MX_ARB 64   // rank counters of the dedup table per scan (zero between uses)

struct MxArgs {
    const uint64_t* codes;
    const uint32_t* nbrs;
    const uint64_t* tids;
    const uint32_t* label_off;
    const int16_t* label_val;
    const int16_t* ls_labels;
    const uint32_t* ls_nodes;
    uint32_t code_stride, nbr_stride, R, n, n_ls, default_start;
    FastLaunch s;
    uint32_t persist;  // 1: rows fetch scans from `queue` until it is empty (grid = resident waves); 0: one scan per row
    uint32_t* queue;   // next scan id (zeroed by the host before the launch)
};

namespace {

__device__ __forceinline__ void mx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// per-scan global state (heap spill array, dedup table) is private to one wave: plain accesses (see vs_search_fast.hip)
__device__ __forceinline__ uint32_t mx_gload32(const uint32_t* p) { return *p; }
__device__ __forceinline__ void mx_gstore32(uint32_t* p, uint32_t v) { *p = v; }
__device__ __forceinline__ uint64_t mx_gload64(const uint64_t* p) { return *p; }
// DPP row operations (a row = the 16 lanes of one scan)
// lane i <- lane i-1 of its row; lane 0 of the row keeps `first`
__device__ __forceinline__ uint32_t row_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x111, 0xF, 0xF, false);
}
// lane i <- lane (i-1) mod 16 of its row
__device__ __forceinline__ uint32_t row_ror1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);
}

struct Lane {
    int lane, gl, gbase;  // lane in the wave, lane in the row, first lane of the row
    // the row's 16 bits of a wave ballot
    __device__ __forceinline__ uint32_t gballot(bool p) const { return (uint32_t)(__ballot(p) >> gbase) & 0xFFFFu; }
    // value held by lane `src` (0..15, row-uniform) of this row
    __device__ __forceinline__ uint32_t gbcast(uint32_t v, uint32_t src) const {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)gbase + (src & 15u)) << 2), (int)v);
    }
    // sum over the row, result in every lane (inclusive prefix by row shifts, then the last lane's value)
    __device__ __forceinline__ uint32_t gsum(uint32_t v) const {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
        return gbcast(v, 15);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Heap of one scan.  Position i lives at l[i + 1] while i < hl (hl = 2^k - 1), else at g[i - hl]; l[0] is a sentinel with
// key 0 ("ancestor of the root").  Every field is row-uniform; all four rows execute every step, `act` selects the rows
// the operation applies to.
// ---------------------------------------------------------------------------------------------------------------
struct MxHeap {
    uint32_t* l;
    uint32_t* g;
    uint32_t hl, sb, len;

    __device__ __forceinline__ uint32_t get1(uint32_t idx) const { return idx <= hl ? l[idx] : mx_gload32(g + (idx - 1 - hl)); }
    __device__ __forceinline__ void set1(uint32_t idx, uint32_t v) const {
        if (idx <= hl) l[idx] = v;
        else mx_gstore32(g + (idx - 1 - hl), v);
    }
    // sift_up(0, p1 - 1) of `elem` (not stored yet): while elem < parent (Reverse => smaller distance) the parent moves down
    __device__ __forceinline__ void place(const Lane& L, uint32_t p1, uint32_t elem, bool act) const {
        const uint32_t r = (uint32_t)L.gl;  // lane r looks at the r-th ancestor (1-based index p1 >> r; 0 = sentinel)
        uint32_t e = 0;
        if (act && r >= 1) e = get1(p1 >> r);
        const bool cmp = act && r >= 1 && (elem >> sb) < (e >> sb);
        const uint32_t bal = L.gballot(cmp) >> 1;                       // bit r-1 <-> ancestor r
        const uint32_t t = (uint32_t)__builtin_ctz(~bal | 0x4000u);     // leading run of ancestors that move down (<= 14)
        // lanes 1 .. t move their ancestor one rank down, lane t + 1 stores the element into the rank below itself
        if (act && r >= 1 && r <= t + 1) set1(p1 >> (r - 1), r <= t ? e : elem);
        mx_wave_sync();
    }
    __device__ __forceinline__ void push(const Lane& L, uint32_t elem, bool act) {
        place(L, len + 1, elem, act);
        len += act ? 1u : 0u;
    }
    // ---- insert_neighbor x c (AM/graph/mod.rs:144-147): elements e[0..c) are pushed one after another, but memory is
    // touched once per run.  The leaves of the run are positions p1f .. p1l (1-based), so their rank-r ancestors are the
    // contiguous range (p1f >> r) .. (p1l >> r): every such range (and the leaves themselves, rank 0) is staged in LDS
    // (stg, MX_STG words per row), the pushes run against the staged copy — lane r always works on rank r, so its slot
    // is (p1 >> r) + a per-lane constant — and the ranges are written back afterwards.  One memory round trip per run
    // instead of one per push when the bottom levels live in the spill array.  The caller keeps a run on ONE heap level
    // (c <= 64 leaves of equal depth), so the ranges of different ranks lie on different levels and never overlap.
    static __device__ __forceinline__ uint32_t stg_off(uint32_t r) {
        return r == 0 ? 0u : r == 1 ? 64u : r == 2 ? 97u : r == 3 ? 114u : r == 4 ? 123u : r == 5 ? 128u : 131u + 2u * (r - 6u);
    }
    __device__ __forceinline__ void push_run_staged(const Lane& L, uint32_t* stg, uint32_t* e, uint32_t* cl, uint32_t c, uint32_t cmax) {
        const bool row = c > 0;
        const uint32_t p1f = len + 1, p1l = len + c;
        const uint32_t r = (uint32_t)L.gl;
        // deepest rank any row needs (ranks past the root read the sentinel)
        uint32_t depth = row ? 32u - (uint32_t)__builtin_clz(p1l) : 0u;  // ranks 1 .. depth - 1 are real ancestors
        depth = max(max((uint32_t)__builtin_amdgcn_readlane((int)depth, 0), (uint32_t)__builtin_amdgcn_readlane((int)depth, 16)),
                    max((uint32_t)__builtin_amdgcn_readlane((int)depth, 32), (uint32_t)__builtin_amdgcn_readlane((int)depth, 48)));
        // stage in.  Rank k holds at most ceil(64 / 2^k) + 1 positions: ranks 1 and 2 take three and two steps of 16 lanes,
        // ranks 3..5 one step each, ranks 6..13 (two positions each) one step together, ranks 14, 15 one more when the
        // heap is that deep.  (Ranks past the root copy the sentinel l[0].)
        auto in_rank = [&](uint32_t k, uint32_t i) {
            const uint32_t b = p1f >> k, n = row ? (p1l >> k) - b + 1u : 0u;
            if (i < n) stg[stg_off(k) + i] = get1(b + i);
        };
        in_rank(1, r);
        in_rank(1, r + 16);
        in_rank(1, r + 32);
        in_rank(2, r);
        in_rank(2, r + 16);
        in_rank(3, r);
        in_rank(4, r);
        in_rank(5, r);
        in_rank(6 + (r >> 1), r & 1u);
        if (depth >= 14) in_rank(14 + ((r >> 1) & 1u), r < 4 ? (r & 1u) : 2u);
        mx_wave_sync();
        const uint32_t cst = stg_off(r) - (p1f >> r);         // slot of rank-r index x = x + cst (mod 2^32)
        const uint32_t cst_w = row_shr1(cst, 0);               // the same for rank r - 1 (lane r >= 1 writes there)
        // An element that is not smaller than the ORIGINAL parent of its leaf stays on the leaf whatever the earlier pushes
        // of the run do (they can only lower that parent), and no push ever reads a leaf of the run: such elements — about
        // a third — are stored right away, 16 per step; the others are compacted in run order (element into e[], its run
        // index into cl[]) and only they go through the sequential loop.
        uint32_t ncl = 0;
        for (uint32_t j0 = 0; j0 < cmax; j0 += MX_G) {
            const uint32_t j = j0 + r;
            const bool in = j < c;
            const uint32_t el = e[j];  // (in-bounds of the row arrays for every j < cmax + 16)
            const uint32_t par = stg[64u + ((p1f + j) >> 1) - (p1f >> 1)];
            const bool climb = in && (el >> sb) < (par >> sb);
            const uint32_t cm = L.gballot(climb);
            if (in && !climb) stg[j] = el;
            if (climb) {
                const uint32_t k = ncl + (uint32_t)__builtin_popcount(cm & ((1u << r) - 1u));
                e[k] = el;
                cl[k] = j;
            }
            ncl += (uint32_t)__builtin_popcount(cm);
            mx_wave_sync();
        }
        const uint32_t nclmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)ncl, 0), (uint32_t)__builtin_amdgcn_readlane((int)ncl, 16)),
                                    max((uint32_t)__builtin_amdgcn_readlane((int)ncl, 32), (uint32_t)__builtin_amdgcn_readlane((int)ncl, 48)));
        for (uint32_t k = 0; k < nclmax; ++k) {
            const bool on = k < ncl;
            // (both reads are unconditional: k < 64 and the slot of any (p1, r) lie inside the row's arrays)
            const uint32_t elem = e[k];
            const uint32_t p1 = p1f + (cl[k] & 63u);
            const uint32_t a = stg[(p1 >> r) + cst];
            const bool cmp = on && r >= 1 && (elem >> sb) < (a >> sb);
            const uint32_t bal = L.gballot(cmp) >> 1;                    // bit r-1 <-> ancestor r
            const uint32_t t = (uint32_t)__builtin_ctz(~bal | 0x4000u);  // leading run of ancestors that move down (<= 14)
            // lanes 1 .. t move their ancestor one rank down; lane t + 1 (the first ancestor that stays, or the sentinel)
            // drops the new element into the rank below itself.  Heaps of < 2^15 entries: rank 15 is the sentinel at most.
            if (on && r >= 1 && r <= t + 1) stg[(p1 >> (r - 1)) + cst_w] = r <= t ? a : elem;
            mx_wave_sync();
        }
        // stage out (same steps, plus the new leaves = rank 0; the sentinel, position 0, is never written back)
        auto out_rank = [&](uint32_t k, uint32_t i) {
            const uint32_t b = p1f >> k, n = row ? (p1l >> k) - b + 1u : 0u;
            if (i < n && b + i >= 1u) set1(b + i, stg[stg_off(k) + i]);
        };
        out_rank(0, r);
        out_rank(0, r + 16);
        out_rank(0, r + 32);
        out_rank(0, r + 48);
        out_rank(1, r);
        out_rank(1, r + 16);
        out_rank(1, r + 32);
        out_rank(2, r);
        out_rank(2, r + 16);
        out_rank(3, r);
        out_rank(4, r);
        out_rank(5, r);
        out_rank(6 + (r >> 1), r & 1u);
        if (depth >= 14) out_rank(14 + ((r >> 1) & 1u), r < 4 ? (r & 1u) : 2u);
        len += c;
        mx_wave_sync();
    }
    // ---- BinaryHeap::pop after the caller has read data[0]: Vec::pop, swap with data[0], sift_down_to_bottom(0), sift_up.
    // sift_down_to_bottom: "which child moves up" is local to a node, so the row evaluates it for a whole 4-level subtree
    // at once (lane j < 15 = node with relative heap index j, one 8-byte read of both children per lane, from LDS or from
    // the spill array); a lane is on the root-to-leaf path iff the choices of its ancestors inside the subtree lead to
    // it, and all moves of a round are one masked store.  Four levels per memory round trip.
    uint32_t lvl, offm1, amask, dpat;  // per-lane constants of the subtree mapping
    __device__ __forceinline__ void init_tree(const Lane& L) {
        const uint32_t j1 = (uint32_t)L.gl + 1u;
        lvl = 31u - (uint32_t)__builtin_clz(j1);
        offm1 = L.gl < 15 ? j1 - (1u << lvl) - 1u : 0x40000000u;
        amask = 0;
        dpat = 0;
        for (uint32_t k = 1; k <= lvl; ++k) {
            const uint32_t anc = (j1 >> k) - 1u, dir = (j1 >> (k - 1)) & 1u;
            amask |= 1u << anc;
            dpat |= dir << anc;
        }
        if (L.gl >= 15) { amask = 0; dpat = 1; }  // never matches
    }
    __device__ __forceinline__ void pop(const Lane& L, bool act) {
        act = act && len > 0;
        const uint32_t last = act ? len - 1 : 0;
        uint32_t item = 0;
        if (act) item = get1(last + 1);  // only needed once the hole has reached a leaf: stays in flight meanwhile
        len = act ? last : len;
        const bool go = act && len > 0;  // a heap that is empty now: nothing to restore
        const uint32_t end = len;
        uint32_t root = 0, pos = 0;
        uint32_t pkey = 0;  // key of the value now stored in the parent of the hole (0 at the heap root: never moves)
        bool going = go;
        while (__ballot(going)) {
            const uint32_t aidx = ((root + 1) << lvl) + offm1;
            const uint32_t c = 2 * aidx + 1;
            const bool exists = going && aidx < end, have1 = going && c < end, have2 = going && c + 1 < end;
            uint32_t le = 0, ri = 0;
            if (have1) {
                if (c < hl) {  // the pair (2a+1, 2a+2) is one aligned 8-byte word in LDS and in the spill array
                    const uint2 p = *reinterpret_cast<const uint2*>(l + c + 1);
                    le = p.x;
                    ri = p.y;
                } else {
                    const uint64_t p = mx_gload64(reinterpret_cast<const uint64_t*>(g + (c - hl)));
                    le = (uint32_t)p;
                    ri = (uint32_t)(p >> 32);
                }
            }
            // child += (data[child] <= data[child + 1]); Reverse => right.d <= left.d picks the right child
            const bool pick = have2 && (ri >> sb) <= (le >> sb);
            const uint32_t cv = pick ? ri : le;
            const uint32_t B = L.gballot(pick);
            // a node is on the path iff every ancestor inside this subtree chose the child leading to it
            const bool onpath = exists && ((B & amask) == dpat);
            const uint32_t pm = L.gballot(onpath);
            if (onpath && have1) set1(aidx + 1, cv);
            mx_wave_sync();
            const uint32_t jd = 31u - (uint32_t)__builtin_clz(pm | 1u);  // deepest path node (the subtree root exists)
            const uint32_t ad = L.gbcast(aidx, jd);
            const uint32_t h1 = L.gballot(have1);
            const bool leaf = !((h1 >> jd) & 1u);  // the hole ends here
            const uint32_t par = L.gbcast(cv, leaf ? ((jd > 0 ? jd - 1 : 0) >> 1) : jd);  // what the hole's parent just received
            if (going) {
                if (leaf) {
                    pos = ad;
                    pkey = jd > 0 ? par >> sb : pkey;
                    going = false;
                } else {
                    pkey = par >> sb;
                    root = 2 * ad + 1 + ((B >> jd) & 1u);  // jd is on the subtree's last level: descend
                }
            }
        }
        // sift_up(0, pos) of the former last element: it only moves when it is smaller than the new parent value
        const bool climb = go && (item >> sb) < pkey;
        if (__ballot(climb)) {
            place(L, pos + 1, item, go);
        } else {
            if (go && L.gl == 0) set1(pos + 1, item);
            mx_wave_sync();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// visited: Vec<ListSearchNeighbor> kept sorted (AM/graph/mod.rs:76,167-168,181): entry i = lane i % 16 of (h[i / 16], n[i / 16])
// ---------------------------------------------------------------------------------------------------------------
template <int VRR>
struct MxVisited {
    // entry i lives at position head + i; position p = lane p % 16 of (h[p / 16], n[p / 16]).  remove(0) only advances
    // `head`; once a whole register has been consumed the registers rotate down by one (plain moves).
    uint32_t h[VRR], n[VRR];
    uint32_t len, head;

    __device__ __forceinline__ void init() {
        len = 0;
        head = 0;
#pragma unroll
        for (int r = 0; r < VRR; ++r) { h[r] = 0; n[r] = 0; }
    }
    static __device__ __forceinline__ uint32_t capacity() { return 16u * (VRR - 1); }  // head < 16 always
    __device__ __forceinline__ uint32_t at(const Lane& L, const uint32_t (&arr)[VRR], uint32_t p) const {
        uint32_t sel = 0;
#pragma unroll
        for (int r = 0; r < VRR; ++r) sel = (p >> 4) == (uint32_t)r ? arr[r] : sel;
        return L.gbcast(sel, p & 15u);
    }
    // hamming of entry i (row-uniform, i < len)
    __device__ __forceinline__ uint32_t ham_at(const Lane& L, uint32_t i) const { return at(L, h, head + i); }
    // visited.insert(partition_point(|x| *x < new), new): before the first element >= new
    __device__ __forceinline__ void insert(const Lane& L, uint32_t hd, uint32_t node, bool act) {
        const uint32_t lo = head, hi = head + len;  // occupied positions [lo, hi)
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < VRR; ++r) {
            const uint32_t gi = (uint32_t)r * 16u + (uint32_t)L.gl;
            cnt += (gi >= lo && gi < hi && h[r] < hd) ? 1u : 0u;
        }
        const uint32_t at_pos = lo + L.gsum(cnt);
#pragma unroll
        for (int r = VRR - 1; r >= 0; --r) {
            const uint32_t ch = r > 0 ? row_ror1(h[r > 0 ? r - 1 : 0]) : 0u;  // lane 0 <- lane 15 of the previous register
            const uint32_t cn = r > 0 ? row_ror1(n[r > 0 ? r - 1 : 0]) : 0u;
            const uint32_t sh = row_shr1(h[r], ch), sn = row_shr1(n[r], cn);
            const uint32_t gi = (uint32_t)r * 16u + (uint32_t)L.gl;
            const bool moved = act && gi > at_pos && gi <= hi, here = act && gi == at_pos;
            h[r] = moved ? sh : (here ? hd : h[r]);
            n[r] = moved ? sn : (here ? node : n[r]);
        }
        len += act ? 1u : 0u;
    }
    // visited.remove(0)
    __device__ __forceinline__ void pop_front(const Lane& L, uint32_t& hd, uint32_t& node, bool act) {
        hd = L.gbcast(h[0], head);
        node = L.gbcast(n[0], head);
        head += act ? 1u : 0u;
        len -= act ? 1u : 0u;
        if (__ballot(head == 16u)) {  // the first register is used up in some row: its registers move down by one
            const bool rot = head == 16u;
#pragma unroll
            for (int r = 0; r + 1 < VRR; ++r) {
                h[r] = rot ? h[r + 1] : h[r];
                n[r] = rot ? n[r + 1] : n[r];
            }
            head = rot ? 0u : head;
        }
    }
};

template <int NCH>
__device__ __forceinline__ uint32_t mx_ham_row(const uint64_t* __restrict__ row, const ulonglong2 (&qv)[NCH], int l4, uint32_t code_stride,
                                               bool active) {
    uint32_t acc = 0;
    if (active) {
        ulonglong2 r[NCH];
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
            r[t] = w < code_stride ? load_stream16(row + w) : make_ulonglong2(0, 0);
        }
#pragma unroll
        for (int t = 0; t < NCH; ++t) acc += (uint32_t)__popcll(r[t].x ^ qv[t].x) + (uint32_t)__popcll(r[t].y ^ qv[t].y);
    }
    return quad_sum(acc);
}

}  // namespace

// NCH = 16-byte chunks of a code row per lane (4 lanes per row), VRR = visited-list registers (16 entries each)
// GD = gather depth: steps of 4 code rows per scan whose loads are in flight together; MINW = waves per SIMD the register
// allocator leaves room for (3: <= 168 VGPRs)
template <int NCH, int VRR, int GD, int MINW>
__global__ __launch_bounds__(WAVE, MINW) void k_search_mx(MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FastLaunch& s = a.s;
    Lane L;
    L.lane = threadIdx.x;
    L.gl = L.lane & 15;
    L.gbase = L.lane & 48;
    const int g = L.lane >> 4;
    // ---- LDS carve (per row) ----
    // (row strides are padded so that the four rows' arrays start in different LDS banks: the rows execute the same
    // instruction and touch the same relative index of their own array most of the time)
    const uint32_t hp_stride = s.hl + 1 + 16;                                   // keeps the 8-byte alignment of sibling pairs
    uint32_t* hp_all = reinterpret_cast<uint32_t*>(smem);                       // 4 x hp_stride
    uint32_t* surv_id_all = hp_all + 4 * hp_stride;                             // 4 x MX_SURV
    uint32_t* surv_e_all = surv_id_all + 4 * MX_SURV;                           // 4 x MX_SURV
    uint32_t* stg_all = surv_e_all + 4 * MX_SURV;                               // 4 x MX_STG
    uint32_t* arb_all = stg_all + 4 * MX_STG;                                   // 4 x MX_ARB
    int16_t* ql_all = reinterpret_cast<int16_t*>(arb_all + 4 * MX_ARB);         // 4 x MX_MAX_QLABELS
    uint32_t* hp = hp_all + (size_t)g * hp_stride;
    uint32_t* surv_id = surv_id_all + g * MX_SURV;
    uint32_t* surv_e = surv_e_all + g * MX_SURV;  // first the dedup slot, then (hamming << sb | slot)
    uint32_t* stg = stg_all + g * MX_STG;
    uint32_t* arb = arb_all + g * MX_ARB;
    int16_t* ql = ql_all + g * MX_MAX_QLABELS;

    const int l4 = L.gl & 3;
    ulonglong2 qv[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) qv[t] = make_ulonglong2(0, 0);
    if (L.gl == 0) hp[0] = 0;  // heap sentinel
    for (uint32_t i = (uint32_t)L.gl; i < MX_ARB; i += MX_G) arb[i] = 0;
    const bool labels_some = s.qlabel_off != nullptr;  // LabeledVector.labels is Some (AM/labels/mod.rs:222-236)
    mx_wave_sync();

    MxHeap heap;
    heap.l = hp;
    heap.g = s.heap_g;
    heap.hl = s.hl;
    heap.sb = s.sb;
    heap.len = 0;
    heap.init_tree(L);
    MxVisited<VRR> vis;
    vis.init();

    const uint32_t smask = (1u << s.sb) - 1u;
    const uint32_t gmask = s.gcap - 1;
    // heaps stay below 2^15 entries (depth <= 15: the lane of rank 15 only ever sees the sentinel, MxHeap::place); a scan
    // that needs more is handed to the general kernel like any other overflow
    const uint32_t hcap = min(s.hcap, 32767u);

    // ---- the scan this row is working on.  A row is idle (no scan), or busy with a scan that is alive or has failed;
    // `done` = nothing left to do for it but publish.  Rows take scans from a queue until it is empty (persist), so a wave
    // does not idle three rows while its longest scan finishes.
    uint32_t q = 0, nql = 0;
    bool has_label_filter = false;  // AM/scan.rs:189
    bool busy = false, alive = false, done = true;
    bool more = true;  // the queue may still hold a scan for this row
    uint32_t emitted = 0, status = 0, nins_g = 0, hmax = 0;
    uint32_t st_visits = 0, st_cand = 0, st_dq = 0, st_reads = 0, st_pfhit = 0;
    uint32_t* ghash = s.ghash;
    uint64_t ft_val = 1;     // heap tid of the visited list's front entry
    uint32_t next_node = 0;  // node id of the heap root

    // ---- the scan's dedup table: buckets of four slots, the probe sequence of an id moves on only past a FULL bucket
    auto bucket_of = [&](uint32_t nid) -> uint32_t { return hash_u32(nid ^ 0x5bd1e995u) & gmask & ~3u; };
    auto bucket_load = [&](uint32_t b0) -> uint4 { return *reinterpret_cast<const uint4*>(ghash + b0); };
    auto hit_mask = [](const uint4& v, uint32_t x) -> uint32_t {
        return (v.x == x ? 1u : 0u) | (v.y == x ? 2u : 0u) | (v.z == x ? 4u : 0u) | (v.w == x ? 8u : 0u);
    };
    // HashSet::insert of ONE id per scan (lane 0 of the row: start nodes); true where the id was not present before
    auto dedup_insert_one = [&](uint32_t nid, bool act, uint32_t& slot_out) -> bool {
        bool fresh = false;
        if (act) {
            uint32_t b0 = bucket_of(nid);
            for (;;) {
                const uint4 v = bucket_load(b0);
                const uint32_t hit = hit_mask(v, nid);
                if (hit) { slot_out = b0 + (uint32_t)__builtin_ctz(hit); break; }
                const uint32_t em = hit_mask(v, VS_EMPTY);
                if (em) {
                    slot_out = b0 + (uint32_t)__builtin_ctz(em);
                    ghash[slot_out] = nid;
                    fresh = true;
                    break;
                }
                b0 = (b0 + 4u) & gmask;
            }
        }
        nins_g += (uint32_t)__builtin_popcount(L.gballot(fresh));
        mx_wave_sync();
        return fresh;
    };
    auto fail = [&](bool cond, uint32_t flag) {  // the scan is handed to the general kernel
        if (alive && cond) {
            status |= flag;
            alive = false;
        }
    };

    // ---- a row begins scan `qn` (everything is predicated on `ini`: the other rows are in the middle of theirs) ----
    auto start_scan = [&](bool ini, uint32_t qn) {
        q = ini ? qn : q;
        busy = busy || ini;
        alive = ini ? true : alive;
        done = ini ? false : done;
        emitted = ini ? 0u : emitted;
        status = ini ? 0u : status;
        nins_g = ini ? 0u : nins_g;
        hmax = ini ? 0u : hmax;
        st_visits = ini ? 0u : st_visits;
        st_cand = ini ? 0u : st_cand;
        st_dq = ini ? 0u : st_dq;
        st_reads = ini ? 0u : st_reads;
        ft_val = ini ? 1ull : ft_val;
        heap.len = ini ? 0u : heap.len;
        heap.g = ini ? s.heap_g + (size_t)qn * s.gstride : heap.g;
        vis.len = ini ? 0u : vis.len;
        vis.head = ini ? 0u : vis.head;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
            if (ini) qv[t] = w < a.code_stride ? *reinterpret_cast<const ulonglong2*>(s.qcodes + (size_t)qn * a.code_stride + w)
                                               : make_ulonglong2(0, 0);
        }
        nql = ini ? 0u : nql;
        if (labels_some && ini) {
            const uint32_t lb = s.qlabel_off[qn], le = s.qlabel_off[qn + 1];
            nql = min(le - lb, (uint32_t)MX_MAX_QLABELS);
            for (uint32_t i = L.gl; i < nql; i += MX_G) ql[i] = s.qlabels[lb + i];
        }
        has_label_filter = ini ? (labels_some && nql > 0) : has_label_filter;
        mx_wave_sync();
        // the scan's dedup table ("inserted", HashSet<ItemPointer>): claimed from the pool and cleared up front
        {
            uint32_t slot = 0;
            if (ini && L.gl == 0) slot = atomicAdd(s.pool_counter, 1u);
            slot = L.gbcast(slot, 0);
            if (ini && slot >= s.pool_slots) {
                status |= OVF_POOL;
                alive = false;
            }
            const bool clr = ini && alive;
            ghash = clr ? s.ghash + (size_t)slot * s.gcap : ghash;
            if (clr)
                for (uint32_t i = 4u * (uint32_t)L.gl; i < s.gcap; i += 4u * MX_G)
                    *reinterpret_cast<uint4*>(ghash + i) = make_uint4(VS_EMPTY, VS_EMPTY, VS_EMPTY, VS_EMPTY);
            mx_wave_sync();
        }
        // ListSearchResult::new: start nodes (AM/graph/mod.rs:97-124, AM/graph/start_nodes.rs:39-48)
        uint32_t nstarts = labels_some ? nql : 1u;
        if (a.default_start == VS_INVALID_NODE || a.n == 0) nstarts = 0;  // ListSearchResult::empty()
        for (uint32_t si = 0; __ballot(ini && alive && si < nstarts); ++si) {
            const bool on = ini && alive && si < nstarts;
            uint32_t sn = VS_INVALID_NODE;
            if (on) {
                if (!labels_some) {
                    sn = a.default_start;
                } else {
                    const int16_t lab = ql[si];
                    int lo = 0, hi = (int)a.n_ls;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (a.ls_labels[mid] < lab) lo = mid + 1;
                        else hi = mid;
                    }
                    if (lo < (int)a.n_ls && a.ls_labels[lo] == lab) sn = a.ls_nodes[lo];
                }
            }
            const bool have = on && sn != VS_INVALID_NODE;
            fail(have && (nins_g + MX_G) * 4u > s.gcap * 3u, OVF_HASH);
            // create_lsn_for_start_node (AM/sbq/storage.rs:365-391)
            uint32_t slot = 0;
            const bool fr_lane = dedup_insert_one(sn, have && alive && L.gl == 0, slot);
            const bool fr = L.gbcast(fr_lane ? 1u : 0u, 0) != 0 && have && alive;
            slot = L.gbcast(slot, 0);
            st_reads += fr ? 1u : 0u;
            const uint32_t d = L.gbcast(
                mx_ham_row<NCH>(a.codes + (size_t)(fr ? sn : 0u) * a.code_stride, qv, l4, a.code_stride, fr && L.gl < 4), 0);
            st_dq += fr ? 1u : 0u;
            st_cand += fr ? 1u : 0u;
            fail(fr && heap.len + 1 > hcap, OVF_HEAP);
            heap.push(L, (d << s.sb) | slot, fr && alive);
        }
        if (ini && alive && heap.len > 0) next_node = mx_gload32(ghash + (hp[1] & smask));
    };
    // ---- a row publishes its finished (or failed) scan and becomes idle ----
    auto finish_scan = [&](bool fin) {
        // one `next` call per emitted row, plus the call that found the stream exhausted
        const uint32_t st_next = emitted + ((emitted < s.M && status == 0) ? 1u : 0u);
        if (fin) {
            if (status == 0) {
                for (uint32_t i = emitted + (uint32_t)L.gl; i < s.M; i += MX_G) {
                    s.out_ids[(size_t)q * s.M + i] = VS_INVALID_NODE;
                    s.out_ham[(size_t)q * s.M + i] = 0xFFFFFFFFu;
                }
            }
            if (L.gl == 0) {
                s.status[q] = status;
                s.out_cnt[q] = status ? 0 : emitted;  // a failed scan publishes an empty stream until the fallback re-runs it
                if (status == 0) {
                    uint32_t* st = s.stats + (size_t)q * ST_N;
                    st[ST_VISITS] = st_visits;
                    st[ST_CAND] = st_cand;
                    st[ST_DQ] = st_dq;
                    st[ST_READS] = st_reads;
                    st[ST_NEXT] = st_next;
                    st[ST_GSPILL] = hmax;
                    st[ST_PFHIT] = st_pfhit;
                    st[7] = nins_g;
                }
            }
        }
        busy = fin ? false : busy;
    };

    // ---- TSVResponseIterator::next until M rows are emitted (AM/scan.rs:210-242): every round, each scan first consumes
    // rows while it cannot visit (consume, AM/graph/mod.rs:174-184), then all scans that can do one visit_closest()
    // expansion (greedy_search_iterate, AM/graph/mod.rs:357-385) ----
    for (;;) {
        done = done || !alive;
        {  // publish what is finished, fetch what is next
            const bool fin = busy && done;
            if (__ballot(fin)) finish_scan(fin);
            const bool want = !busy && more;
            if (__ballot(want)) {
                uint32_t qn = blockIdx.x * 4u + (uint32_t)g;  // one scan per row ...
                if (a.persist) {                               // ... or the next one in the queue
                    qn = 0xFFFFFFFFu;
                    if (want && L.gl == 0) qn = atomicAdd(a.queue, 1u);
                    qn = L.gbcast(qn, 0);
                }
                const bool got = want && qn < s.nq;
                more = want ? (got && a.persist != 0) : more;
                start_scan(got, qn);
                done = done || !alive;
            }
            if (!__ballot(busy)) break;
        }
        // can this scan visit?  visit_closest(L) stop rule (AM/graph/mod.rs:153-170)
        auto can_visit_now = [&]() -> bool {
            bool cv = !done && heap.len > 0;
            uint32_t lim = 0;
            const bool need = cv && vis.len > s.L;
            if (__ballot(need)) lim = vis.ham_at(L, need ? s.L - 1 : 0);
            if (need) cv = (hp[1] >> s.sb) < lim;
            return cv;
        };
        bool cv = can_visit_now();
        while (__ballot(!done && !cv)) {
            const bool con = !done && !cv;
            // ---- consume + return_lsn (AM/sbq/storage.rs:404-414) ----
            const bool ended = con && vis.len == 0;  // None: the stream has ended
            done = done || ended;
            const bool take = con && !ended;
            uint32_t fd, fnode;
            vis.pop_front(L, fd, fnode, take);
            st_reads += take ? 1u : 0u;
            const uint64_t tid = ft_val;  // requested when this entry became the front
            if (__ballot(take && vis.len > 0)) {  // the new front's heap tid, should the scan consume again right away
                const uint32_t fn = L.gbcast(vis.n[0], vis.head);
                if (take && vis.len > 0) ft_val = a.tids[fn];
            }
            const bool live_row = take && (tid & 0xFFFFull) != 0;  // InvalidOffsetNumber: deleted tuple (AM/scan.rs:231-234)
            if (live_row && L.gl == 0) {
                s.out_ids[(size_t)q * s.M + emitted] = fnode;
                s.out_ham[(size_t)q * s.M + emitted] = fd;
            }
            emitted += live_row ? 1u : 0u;
            done = done || (live_row && emitted == s.M);
            cv = can_visit_now();
        }
        if (!__ballot(!done)) continue;  // nothing to expand: publish / refill
        const bool ex = !done && cv;  // this scan expands now
        hmax = ex ? max(hmax, heap.len) : hmax;
        const uint32_t top = ex ? hp[1] : 0u;
        const uint32_t hd = top >> s.sb;
        // handle -> node id: requested when the root last changed (end of the previous step), so it has arrived by now
        const uint32_t node = ex ? next_node : 0u;
        // the neighbor list (lane gl holds slots gl, gl + 16, gl + 32, gl + 48) is requested before the pop, which covers
        // part of its latency
        const uint32_t* nrow = a.nbrs + (size_t)node * a.nbr_stride;
        uint32_t nb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t sl = (uint32_t)k * 16u + (uint32_t)L.gl;
            nb[k] = (ex && sl < a.R) ? nrow[sl] : VS_INVALID_NODE;
        }
        heap.pop(L, ex);
        fail(ex && vis.len + 1 > vis.capacity(), OVF_VISITED);
        const bool ex2 = ex && alive;
        st_visits += ex2 ? 1u : 0u;
        st_reads += ex2 ? 1u : 0u;  // SbqNode::read(visiting)
        // visited.insert(partition_point(|x| *x < head), head)
        vis.insert(L, hd, node, ex2);
        // ---- visit_lsn_internal, Disk arm (AM/sbq/storage.rs:135-190) ----
        uint32_t c = 0;          // survivors of this visit, in neighbor-list order
        fail(ex2 && (nins_g + 64u) * 4u > s.gcap * 3u, OVF_HASH);  // room for every id of this list
        // prepare_insert marks BEFORE the label check (AM/sbq/storage.rs:148-172).  First the live slots of the list and
        // the first probe of every one of them (up to four L2 atomics per lane in flight) ...
        bool actk[4], freshk[4];
        uint32_t gsk[4];
        uint4 bkk[4];
        {
            bool open = ex2 && alive;  // the list has not ended yet
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // list ends at the first InvalidBlockNumber (AM/sbq/node.rs:260-285)
                const uint32_t inval = L.gballot(nb[k] == VS_INVALID_NODE);
                const uint32_t nvalid = inval ? (uint32_t)__builtin_ctz(inval) : 16u;
                actk[k] = open && (uint32_t)L.gl < nvalid;
                open = open && nvalid == 16u;
                gsk[k] = bucket_of(nb[k]);
                freshk[k] = false;
                bkk[k] = make_uint4(0, 0, 0, 0);
                if (actk[k]) bkk[k] = bucket_load(gsk[k]);
            }
        }
        // ... then every id is looked up in / added to its bucket.  All four steps work on ONE snapshot of the table (the
        // loads above), so the lanes that want an empty slot of a bucket — from whichever step — draw distinct ranks from
        // the scan's LDS counter of that bucket (other buckets sharing the counter only waste ranks) and take the rank-th
        // empty slot of the snapshot; the counters are reset only after all steps.  A lane whose rank is past the
        // bucket's empties, or whose bucket is full, takes part in the next round with a fresh snapshot.
        {
            bool pendk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pendk[k] = actk[k];
            for (;;) {
                mx_wave_sync();  // every lane holds its snapshot before any lane stores (the loads are earlier instructions)
                uint32_t ctrk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ctrk[k] = 0xFFFFFFFFu;
                    if (pendk[k]) {
                        const uint32_t nid = nb[k];
                        const uint32_t hit = hit_mask(bkk[k], nid);
                        if (hit) {
                            gsk[k] += (uint32_t)__builtin_ctz(hit);
                            pendk[k] = false;
                        } else {
                            const uint32_t em = hit_mask(bkk[k], VS_EMPTY);
                            if (em == 0) {
                                gsk[k] = (gsk[k] + 4u) & gmask;
                            } else {
                                ctrk[k] = (gsk[k] >> 2) & (MX_ARB - 1u);
                                const uint32_t rank = atomicAdd(&arb[ctrk[k]], 1u);  // ds_add_rtn_u32
                                if (rank < (uint32_t)__builtin_popcount(em)) {
                                    uint32_t m = em;
                                    if (rank > 0) m &= m - 1u;
                                    if (rank > 1) m &= m - 1u;
                                    if (rank > 2) m &= m - 1u;
                                    gsk[k] += (uint32_t)__builtin_ctz(m);
                                    ghash[gsk[k]] = nid;
                                    freshk[k] = true;
                                    pendk[k] = false;
                                }
                            }
                        }
                    }
                }
                mx_wave_sync();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ctrk[k] != 0xFFFFFFFFu) arb[ctrk[k]] = 0;
                mx_wave_sync();
                if (!__ballot(pendk[0] || pendk[1] || pendk[2] || pendk[3])) break;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (pendk[k]) bkk[k] = bucket_load(gsk[k]);
            }
        }
        // the label filter is applied and the survivors are compacted in list order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!__ballot(actk[k])) continue;
            const uint32_t nid = nb[k];
            const bool fresh = freshk[k];
            const uint32_t hslot = gsk[k];
            const uint32_t fm = L.gballot(fresh);
            nins_g += (uint32_t)__builtin_popcount(fm);
            st_reads += (uint32_t)__builtin_popcount(fm);  // SbqNode::read(neighbor)
            // label filter: query.labels.overlaps(node.labels) (AM/labels/mod.rs:124-142)
            bool pass = fresh;
            if (has_label_filter && fresh) {
                const uint32_t lb = a.label_off[nid], le = a.label_off[nid + 1];
                uint32_t i = 0, j = lb;
                bool ov = false;
                while (i < nql && j < le) {
                    const int16_t x = ql[i], y = a.label_val[j];
                    if (x == y) { ov = true; break; }
                    if (x < y) ++i;
                    else ++j;
                }
                pass = ov;
            }
            const uint32_t pm = L.gballot(pass);
            if (pass) {
                const uint32_t rank = c + (uint32_t)__builtin_popcount(pm & ((1u << L.gl) - 1u));
                surv_id[rank] = nid;
                surv_e[rank] = hslot;
            }
            c += (uint32_t)__builtin_popcount(pm);
        }
        c = (ex2 && alive) ? c : 0u;
        fail(c > 0 && heap.len + c > hcap, OVF_HEAP);
        c = alive ? c : 0u;
        mx_wave_sync();
        // ---- distances: 4 lanes per code row, 4 rows per scan per step ----
        st_dq += c;
        st_cand += c;
        // the largest c over the four rows (each row holds a uniform value: its first lane is enough)
        const uint32_t cmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)c, 0), (uint32_t)__builtin_amdgcn_readlane((int)c, 16)),
                                  max((uint32_t)__builtin_amdgcn_readlane((int)c, 32), (uint32_t)__builtin_amdgcn_readlane((int)c, 48)));
        for (uint32_t p0 = 0; p0 < cmax; p0 += 4u * GD) {  // GD steps (4 GD rows per scan, 16 GD per wave) in flight
            uint32_t jj[GD];
            bool vv[GD];
            ulonglong2 rr[GD][NCH];
#pragma unroll
            for (int u = 0; u < GD; ++u) {
                jj[u] = p0 + 4u * (uint32_t)u + (uint32_t)(L.gl >> 2);
                vv[u] = jj[u] < c;
                const uint32_t id = vv[u] ? surv_id[jj[u]] : 0u;
                const uint64_t* row = a.codes + (size_t)id * a.code_stride;
#pragma unroll
                for (int t = 0; t < NCH; ++t) {
                    const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
                    rr[u][t] = (vv[u] && w < a.code_stride) ? load_stream16(row + w) : make_ulonglong2(0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < GD; ++u) {
                uint32_t acc = 0;
#pragma unroll
                for (int t = 0; t < NCH; ++t) acc += (uint32_t)__popcll(rr[u][t].x ^ qv[t].x) + (uint32_t)__popcll(rr[u][t].y ^ qv[t].y);
                const uint32_t d = quad_sum(vv[u] ? acc : 0u);
                if (vv[u] && l4 == 0) surv_e[jj[u]] = (d << s.sb) | surv_e[jj[u]];
            }
        }
        mx_wave_sync();
        // ---- insert_neighbor in list order (AM/graph/mod.rs:144-147) ----
        if (cmax > 0 && !__ballot(c > 0 && heap.len < 64u)) {
            // runs stay on one heap level (all leaves of a run have the same depth, so the staged ranges of different
            // ranks can never name the same position): a visit that crosses into the next level takes two runs
            uint32_t jb = 0;
            while (__ballot(jb < c)) {
                const uint32_t p1f = heap.len + 1;
                const uint32_t room = (2u << (31u - (uint32_t)__builtin_clz(p1f))) - p1f;  // leaves left on this level
                const uint32_t n = jb < c ? min(c - jb, room) : 0u;
                const uint32_t nmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)n, 0), (uint32_t)__builtin_amdgcn_readlane((int)n, 16)),
                                          max((uint32_t)__builtin_amdgcn_readlane((int)n, 32), (uint32_t)__builtin_amdgcn_readlane((int)n, 48)));
                heap.push_run_staged(L, stg, surv_e + jb, surv_id, n, nmax);  // (surv_id is free after the gather)
                jb += n;
            }
        } else {  // a shallow heap somewhere (the first expansions of a scan): one push at a time
            for (uint32_t j = 0; j < cmax; ++j) {
                const bool on = j < c;
                const uint32_t elem = on ? surv_e[j] : 0u;
                heap.push(L, elem, on);
            }
        }
        // the root can only change in an expansion: ask for the id of the next node to visit now, and for the heap tid of
        // the visited list's front entry (what the next consume() returns)
        if (ex && alive && heap.len > 0) next_node = mx_gload32(ghash + (hp[1] & smask));
        {
            const uint32_t fn = L.gbcast(vis.n[0], vis.head);
            if (ex && alive && vis.len > 0) ft_val = a.tids[fn];
        }
    }

}

// ---- host side --------------------------------------------------------------------------------------------------------
static size_t mx_lds_bytes(const FastLaunch& s) {
    return ((size_t)4 * (s.hl + 1 + 16) * 4 + 2 * 4 * MX_SURV * 4 + 4 * MX_STG * 4 + 4 * MX_ARB * 4 + 4 * MX_MAX_QLABELS * 2 + 15) / 16 * 16;
}

// can this launch run on k_search_mx?  (table-less regime, query scans only, geometry the row-of-16 mapping covers)
bool search_mx_eligible(const vs_index* idx, const FastLaunch& s) {
    const uint32_t nch = (idx->code_stride + 7) / 8;
    const uint32_t want_v = s.L + s.L / 2 + 32;
    return s.lh == 0 && !s.build && !s.phase && idx->d.num_neighbors <= 64 && nch >= 1 && nch <= 6 &&
           want_v <= 16 * 32 && mx_lds_bytes(s) <= 64 * 1024;
}

template <int NCH, int VRR, int GD, int MINW>
static int launch_mx_ttt(vs_index* idx, const MxArgs& a, size_t lds) {
    static bool attr_set = false;
    if (!attr_set) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_mx<NCH, VRR, GD, MINW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // persistent rows: as many waves as the chip holds at once (LDS and the register cap decide), each row fetching scans
    // from the queue; VS_MX_PERSIST=0: one scan per row, (nq + 3) / 4 waves; VS_MX_GRID: number of waves (tests)
    uint32_t grid = (a.s.nq + 3) / 4;
    if (a.persist) {
        const uint32_t per_cu = (uint32_t)std::min<size_t>((size_t)MINW * 4, (160 * 1024) / std::max<size_t>(lds, 1));
        const uint32_t resident = (uint32_t)std::max(idx->ctx->prop.multiProcessorCount, 1) * std::max(per_cu, 1u);
        grid = std::min(grid, resident);
        const char* e = getenv("VS_MX_GRID");
        if (e && *e) grid = std::max(1u, std::min(grid, (uint32_t)strtoul(e, nullptr, 10)));
    }
    hipLaunchKernelGGL((k_search_mx<NCH, VRR, GD, MINW>), dim3(grid), dim3(WAVE), lds, idx->ctx->stream, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

template <int NCH, int VRR>
static int launch_mx_tt(vs_index* idx, const MxArgs& a, size_t lds) {
    // VS_MX_GD=4: 16 code rows per scan in flight (more registers: 2 waves per SIMD) instead of 8 — a tuning variant for
    // the headline code width, to be decided by measurement
    const char* e = getenv("VS_MX_GD");
    if (NCH == 3 && e && *e == '4') return launch_mx_ttt<3, VRR, 4, 2>(idx, a, lds);
    return launch_mx_ttt<NCH, VRR, 2, 3>(idx, a, lds);
}

template <int NCH>
static int launch_mx_t(vs_index* idx, const MxArgs& a, size_t lds) {
    const uint32_t want_v = a.s.L + a.s.L / 2 + 32;
    // capacity of the register-resident visited list = 16 (VRR - 1) entries
    if (want_v <= 16 * 12) return launch_mx_tt<NCH, 13>(idx, a, lds);
    if (want_v <= 16 * 20) return launch_mx_tt<NCH, 21>(idx, a, lds);
    return launch_mx_tt<NCH, 33>(idx, a, lds);
}

int launch_search_mx(vs_index* idx, const FastLaunch& s) {
    if (s.nq == 0) return VS_OK;
    VS_REQUIRE(search_mx_eligible(idx, s), "k_search_mx: launch outside the geometry this kernel covers");
    MxArgs a;
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
    a.s = s;
    {
        const char* e = getenv("VS_MX_PERSIST");
        a.persist = (e && *e == '0') ? 0u : 1u;
        a.queue = s.pool_counter + 4;  // the 64-byte counter block is zeroed before every launch (vs_api.hip)
    }
    const size_t lds = mx_lds_bytes(s);
    const uint32_t nch = (idx->code_stride + 7) / 8;
    switch (nch) {
        case 1: return launch_mx_t<1>(idx, a, lds);
        case 2: return launch_mx_t<2>(idx, a, lds);
        case 3: return launch_mx_t<3>(idx, a, lds);
        case 4: return launch_mx_t<4>(idx, a, lds);
        case 5:
        case 6: return launch_mx_t<6>(idx, a, lds);
        default: break;
    }
    vs_set_error("k_search_mx: code width not covered");
    return VS_ERR_INVALID;
}
