// vs_search_fast.hip — K3 fast path: the streaming beam search with all hot per-scan state on chip, one wave64 per scan.
//
// Same semantics and citations as vs_search.hip (the general kernel).  What limits this kernel is not bandwidth but
// the serial chain of one scan (pop -> neighbor row -> dedup -> code gather -> pushes -> pop ...) times the number of
// scans a CU can hold at once, which is set by LDS bytes per scan.  So the data structures are restated to keep the
// dependent chain short and the LDS footprint small:
//
//   * candidate heap  BinaryHeap<Reverse<ListSearchNeighbor>> (AM/graph/mod.rs:75): 4-byte entries
//     (hamming << sb | dedup slot) in LDS; the node id is looked up in the dedup table when the entry is popped.
//       - pop  = sift_down_to_bottom: "which child moves up" is local to a node, so the wave evaluates it for a whole
//         6-level subtree at once (one ds_read_b64 of both children per lane); a lane is on the root-to-leaf path iff
//         the choices of its ancestors lead to it — one AND + compare against per-lane constants — and all moves are
//         one masked LDS store.  Two such rounds cover 4095 entries.
//       - push = sift_up: lane r compares the new element with its r-th ancestor (one LDS read for the whole chain),
//         the climb is one ballot + ctz and the shift down the path one LDS store.  (Batching several pushes per
//         round was tried: ~70 % of the candidates climb and siblings then conflict, so it did not pay.)
//     Both replay Rust std's exact array mechanics (tie order of equal Hamming distances depends on them).
//   * dedup set ("inserted", HashSet<ItemPointer>): exact open-addressing table in LDS (ds_cmpst), never rehashed, so
//     a slot index is a stable handle for the node id.  When it reaches 87.5 % load it is frozen (read-only) and new
//     ids go to a per-scan global table (handles >= lh) that the wave clears lazily — the long tail of scans pays L2
//     latency for its last inserts instead of forcing every scan to reserve LDS for the worst case.  Big scans (lh == 0,
//     the "table-less" regime) keep every id there.  The global table belongs to ONE wave, so it needs no atomics (an L2
//     atomic costs a 64-byte write to the memory side each; scripts/microbench/randmem.hip: 26 G/s for the whole chip
//     against 60-100 G/s for loads): a probe is one 16-byte load of a bucket of four slots, an insert a 4-byte store, and
//     lanes that want the same bucket in the same step are told apart by an LDS counter.  The default of the table-less regime
//     since round 4 keeps one OCCUPANCY BIT PER SLOT in LDS instead (VG == 2): slot-granular linear probing, an id whose home slot
//     is free is new and is stored without a load, an occupied run is compared through one 16-byte load per 4-slot group, a free
//     slot is claimed with one ds_or; the tables are never cleared and never read before they are written.
//   * visited list (sorted Vec<ListSearchNeighbor>): sorted array in REGISTERS (entry i = lane i % 64 of register
//     i / 64); insert / remove(0) are DPP wave shifts, no LDS traffic.  (LDS ring buffer when it does not fit.)
//   * the query code lives in registers (4 lanes x 16 B per code row, NCH steps).
//   * neighbor rows are requested ahead of the pop that needs them (see the main loop).
//
// A scan whose state outgrows the LDS budget sets a status flag and is re-run by the general kernel
// (vs_search.hip, unbounded global spill) in a follow-up launch that skips every scan that completed here.
//
// THREE translation units are compiled from this source (round 6; VS_FAST_TU, set by the two-line files vs_search_fast_plain6.hip and
// vs_search_fast_keys6.hip that include it): 0 = the dispatcher and every instantiation but four; 1 = the two instantiations the
// unfiltered scans of the usual index run at six waves per SIMD; 2 = their two label-key counterparts.  The split exists so that each
// group can be built with the code-generation options measured best for IT: the compiler's scheduling strategy moves these kernels by
// 1-10 % and in opposite directions (profiles/r06/s17, s18; csrc/Makefile has the options and the numbers).
#ifndef VS_FAST_TU
#define VS_FAST_TU 0
#endif
#include <cstdlib>

#include "vs_device.h"

#define MAX_QLABELS 64
// odd multipliers of the 16-bit table's bijection (fast_scan, VG == 3) and their inverses mod 2^32
#define VS_Q16_A 0x9E3779B1u
#define VS_Q16_B 0x85EBCA6Bu
#define VS_Q16_AI 0x0E8B2F51u
#define VS_Q16_BI 0xA5CB9243u
static_assert((uint32_t)(VS_Q16_A * VS_Q16_AI) == 1u && (uint32_t)(VS_Q16_B * VS_Q16_BI) == 1u, "inverse multipliers");
#define ARB_SLOTS 128

struct FastArgs {
    const uint64_t* codes;
    const uint32_t* nbrs;
    const uint64_t* tids;
    const uint32_t* label_off;
    const int16_t* label_val;
    const uint64_t* label_mask;  // may be null
    const uint8_t* label_bit;    // with label_mask: [65536] label -> bit (0xFF: in no node's set)
    const uint64_t* nbr_mask;    // may be null: [n][nbr_stride] the label masks of every node's neighbors, in list order
    const int16_t* ls_labels;
    const uint32_t* ls_nodes;
    uint32_t code_stride, nbr_stride, R, n, n_ls, default_start;
    FastLaunch s;
};

__device__ __forceinline__ void wave_sync() {
    // single-wave workgroup: LDS operations of one wave execute in order, so cross-lane LDS communication only needs
    // the COMPILER not to cache / reorder across this point (no instructions are emitted)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Per-scan global state (heap spill array, dedup table) is private to one wave: plain loads and stores.  The vector memory
// operations of a wave reach its CU's L1 in program order and no other CU ever writes these lines, so a lane reads what any
// lane of the wave stored earlier (wave_sync() keeps the compiler from reordering across the hand-over); the lines stay
// write-back in L2 instead of costing a fabric write per store as agent-scope stores do.
// The address space is spelled out: heap.l / heap.g are members the compiler cannot always trace back to the __shared__ array
// and the kernel argument, and "position < hl ? LDS : spill array" then becomes ONE flat access through a selected pointer — LDS
// traffic through the vector memory path, and a wait on every outstanding load (row prefetches included) behind each of them.
typedef __attribute__((address_space(1))) uint32_t glb_u32;
typedef __attribute__((address_space(1))) uint64_t glb_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint16_t glb_u16;
__device__ __forceinline__ uint32_t gload32(const uint32_t* p) { return *(const glb_u32*)p; }
__device__ __forceinline__ void gstore32(uint32_t* p, uint32_t v) { *(glb_u32*)p = v; }
__device__ __forceinline__ uint64_t gload64u(const uint64_t* p) { return *(const glb_u64*)p; }
__device__ __forceinline__ uint32_t gload16(const uint16_t* p) { return (uint32_t)*(const glb_u16*)p; }
__device__ __forceinline__ void gstore16(uint16_t* p, uint32_t v) { *(glb_u16*)p = (uint16_t)v; }
__device__ __forceinline__ uint32_t lload32(const uint32_t* p) { return *(const lds_u32*)p; }
__device__ __forceinline__ void lstore32(uint32_t* p, uint32_t v) { *(lds_u32*)p = v; }
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
// lane i <- lane i-1, lane 0 <- carry   /   lane i <- lane i+1, lane 63 <- carry   (DPP wave shifts, gfx9 family)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x130, 0xF, 0xF, false);
}

// ---------------------------------------------------------------------------------------------------------------
// Heap position i lives at l[i + 1] while i < hl (hl = 2^k - 1, so a sibling pair (2a+1, 2a+2) is one aligned 8-byte
// word both in LDS and in the spill array g[i - hl]); l[0] is a sentinel with key 0 ("ancestor of the root").
// ---------------------------------------------------------------------------------------------------------------
// PK: the per-lane constants below are kept packed in one register and unpacked where they are used (the 64-VGPR variant)
template <bool PK>
struct FastHeap {
    uint32_t* l;
    uint32_t* g;
    uint32_t hl, sb, len;
    int lane;
    uint32_t lvl_, offm1_;  // this lane's level / (offset - 1) inside a 6-level subtree (lane 63: never a node)
    uint32_t pk;            // PK: wl_rank | wl_slot << 5 | wl_base << 10 | lvl << 16
    __device__ __forceinline__ uint32_t lvl() const { return PK ? (pk >> 16) : lvl_; }
    __device__ __forceinline__ uint32_t offm1() const {
        if (!PK) return offm1_;
        return lane < 63 ? (uint32_t)lane - (1u << (pk >> 16)) : 0x40000000u;
    }
    __device__ __forceinline__ uint32_t wl_rank() const { return PK ? (pk & 31u) : wl_rank_; }
    __device__ __forceinline__ uint32_t wl_slot() const { return PK ? ((pk >> 5) & 31u) : wl_slot_; }
    __device__ __forceinline__ uint32_t wl_base() const { return PK ? ((pk >> 10) & 63u) : wl_base_; }
    uint32_t amask, dpat;  // lane j is on the sift-down path iff (pick_bits & amask) == dpat (ancestor choices)

    __device__ __forceinline__ void init(int lane_) {
        lane = lane_;
        len = 0;
        const uint32_t j1 = (uint32_t)lane + 1u;
        const uint32_t lvl = 31u - (uint32_t)__builtin_clz(j1);
        lvl_ = lvl;
        offm1_ = lane < 63 ? j1 - (1u << lvl) - 1u : 0x40000000u;
        amask = 0;
        dpat = 0;
        for (uint32_t k = 1; k <= lvl; ++k) {
            const uint32_t anc = (j1 >> k) - 1u, dir = (j1 >> (k - 1)) & 1u;
            amask |= 1u << anc;
            dpat |= dir << anc;
        }
        if (lane >= 63) { amask = 0; dpat = 1; }  // never matches
        init_wide();
    }
    __device__ __forceinline__ uint32_t root() const { return rfl(l[1]); }

    __device__ __forceinline__ uint32_t get(uint32_t i) const { return i < hl ? lload32(l + i + 1) : gload32(g + (i - hl)); }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const {
        if (i < hl) lstore32(l + i + 1, v);
        else gstore32(g + (i - hl), v);
    }
    // ---- sift_up(0, pos) of `elem` (not yet stored): while elem < parent (Reverse => smaller distance) move parent down
    __device__ __forceinline__ void sift_up_lds(uint32_t pos, uint32_t elem) const {
        const uint32_t p1 = pos + 1;
        // lane r looks at the r-th ancestor, stored at l[p1 >> r] (l[0] = sentinel once the root is passed); lanes >= 32
        // alias lanes r - 32 (shift amounts are mod 32) and are ignored
        const uint32_t e = l[p1 >> (lane & 31)];
        const bool cmp = (elem >> sb) < (e >> sb);
        const uint32_t bal = (uint32_t)__ballot(cmp) >> 1;  // bit r-1 <-> ancestor r; bit 31 is always clear
        const uint32_t t = (uint32_t)__builtin_ctz(~bal);   // leading run of ancestors that move down
        if ((uint32_t)lane <= t) {
            const uint32_t dst = lane == 0 ? (p1 >> t) : (p1 >> (lane - 1));
            l[dst] = lane == 0 ? elem : e;
        }
        wave_sync();
    }
    __device__ __forceinline__ void sift_up_gen(uint32_t pos, uint32_t elem) const {
        const uint32_t p1 = pos + 1;
        const uint32_t r = (uint32_t)lane;
        const uint32_t ar1 = r < 32 ? (p1 >> r) : 0u;  // (r-th ancestor) + 1
        const bool valid = r >= 1 && ar1 >= 1;
        uint32_t e = 0;
        if (valid) e = get(ar1 - 1);
        const bool cmp = valid && (elem >> sb) < (e >> sb);
        const uint64_t bal = __ballot(cmp) >> 1;
        const uint32_t t = (uint32_t)__builtin_ctzll(~bal);
        if (r >= 1 && r <= t) set((p1 >> (r - 1)) - 1, e);
        if (lane == 0) set((p1 >> t) - 1, elem);
        wave_sync();
    }
    __device__ __forceinline__ void push(uint32_t elem) {
        if (len < hl) sift_up_lds(len, elem);
        else sift_up_gen(len, elem);
        len += 1;
    }

    // ---- insert_neighbor x c (AM/graph/mod.rs:144-147): elements 0..c-1 (lanes 0..c-1 of `entry`, neighbor-list order)
    // are pushed one after another, but memory is touched once per RUN of up to 32 leaves on one level (the ~31 candidates of a
    // typical visit are one run, i.e. one round trip to the spill array):
    //   * the distinct ancestors of such a run are few (<= 17 parents, 9 grandparents, 5, 3, then <= 2 per rank), so one
    //     wave-wide load (lane -> fixed (rank, slot) layout below) brings all of them in, from LDS or from the spill array;
    //   * consecutive leaves P, P+1 share their ancestors from rank sh = bitlen((P+1) xor (P+2)) upwards, and what push j
    //     did to that shared chain is known in registers (ranks r < t_j now hold the old rank r+1 value, rank t_j holds
    //     element j); the ranks below sh are positions no earlier push of the run can have touched, so push j+1 takes
    //     them from the wide load (ds_bpermute) and the rest from the patched chain of push j.
    // lane layout of the wide load: rank 1 -> lanes 0..16, rank 2 -> 17..25, rank 3 -> 26..30, rank 4 -> 31..33,
    // rank r >= 5 -> lanes 34 + 2 (r - 5), +1  (ranks up to 19: heaps of up to 2^19 entries)
    uint32_t wl_rank_, wl_slot_, wl_base_;  // this lane's (rank, slot) as a wide-load lane; first lane of rank `lane`
    __device__ __forceinline__ void init_wide() {
        const uint32_t i = (uint32_t)lane;
        uint32_t wl_rank, wl_slot;
        if (i < 17) { wl_rank = 1; wl_slot = i; }
        else if (i < 26) { wl_rank = 2; wl_slot = i - 17; }
        else if (i < 31) { wl_rank = 3; wl_slot = i - 26; }
        else if (i < 34) { wl_rank = 4; wl_slot = i - 31; }
        else { wl_rank = 5 + ((i - 34) >> 1); wl_slot = (i - 34) & 1u; }
        const uint32_t r = i;  // as a chain lane: where do rank-r values start
        const uint32_t wl_base = r <= 1 ? 0u : (r == 2 ? 17u : (r == 3 ? 26u : (r == 4 ? 31u : 34u + 2u * (r - 5u))));
        wl_rank_ = wl_rank;
        wl_slot_ = wl_slot;
        wl_base_ = wl_base;
        pk = (wl_rank & 31u) | (wl_slot << 5) | ((wl_base & 63u) << 10) | (lvl_ << 16);
    }
    // entry at index idx (= heap position + 1; 0 = sentinel)
    __device__ __forceinline__ uint32_t get1(uint32_t idx) const { return idx <= hl ? lload32(l + idx) : gload32(g + (idx - 1 - hl)); }
    __device__ __forceinline__ void set1(uint32_t idx, uint32_t v) const {
        if (idx <= hl) lstore32(l + idx, v);
        else gstore32(g + (idx - 1 - hl), v);
    }
    // geometry of the first run of c pushes starting at the current length (0 = "one at a time" path)
    __device__ __forceinline__ uint32_t first_run(uint32_t c) const {
        const uint32_t p1f = len + 1;
        if (len < 64 || p1f >= (1u << 19)) return 0;
        const uint32_t room = (2u << (31u - (uint32_t)__builtin_clz(p1f))) - p1f;
        return min(min(c, 32u), room);
    }
    // the wide ancestor load of a run of n leaves starting at the current length (only ISSUES the reads)
    __device__ __forceinline__ uint32_t wide_load(uint32_t n) const {
        const uint32_t p1f = len + 1, p1l = p1f + n - 1;
        const uint32_t idx = (p1f >> wl_rank()) + wl_slot();
        uint32_t anc = 0;
        if (idx <= (p1l >> wl_rank())) anc = p1l > hl ? get1(idx) : l[idx];
        return anc;
    }
    // ---- a run of n pushes, level by level instead of element by element -------------------------------------------------
    // A push is also a walk DOWN the root-to-leaf path with the element in hand: at the first node whose key is greater the
    // element stays and the node's old value is carried on, and from there on every node takes what is carried and hands its own
    // value down (that is sift_up's shift of the path by one position; "at the first greater key" is where sift_up stops
    // climbing).  Read that way, what push i meets at a node depends only on the pushes before it at that node and on what it
    // was handed from above — so all pushes of the run can do one LEVEL at a time.  At a node the pushes of its leaves arrive
    // in leaf order; the node holds the minimum seen so far (a carried value wins ties, an element loses them), i.e. an
    // exclusive prefix minimum inside the node's block of leaves seeded with the node's original value, and a push hands down
    // the loser of (what it brought, what the node held).  Lanes are the leaves (lane = leaf index mod 32-aligned window, so
    // the leaves of a rank-k node are an aligned block of 2^k lanes); ranks 1..5 are done this way, which settles everything
    // that stays inside the 32-leaf subtrees.  The few elements small enough to get above rank 5 first walk the part of the
    // path above it one after another (the chain-in-registers loop, with the rank-5 node in the role of the leaf).
    __device__ __forceinline__ void push_run_scan(uint32_t entry, uint32_t j, uint32_t n, uint32_t anc, bool spill) {
        const uint32_t IDENT = 0xFFFFFFFFu;
        const uint32_t p1f = len + 1;
        const uint32_t off = p1f & 31u;
        // Where rank k of the run lives: a run stays on one heap level and hl + 1 is a power of two, so the rank-k ancestors of its
        // leaves are all on ONE level too — every one of them in LDS or every one in the spill array, a wave-uniform test per rank
        // (round 6; the per-lane "LDS or spill array" of set1() cost two exec-mask regions per store)
        auto put = [&](const uint32_t k, uint32_t idx, uint32_t v) {
            if ((p1f >> k) > hl) gstore32(g + (idx - 1 - hl), v);
            else l[idx] = v;
        };
        const uint32_t i = (uint32_t)lane - off;            // this lane's element (leaf order); >= n: not a leaf of the run
        const bool in = i < n;
        const uint32_t p1 = (p1f & ~31u) + (uint32_t)lane;  // its leaf (position + 1)
        uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j + i) << 2), (int)entry);  // what the push carries
        // original value of the leaf's ancestor of rank k (wide-load lane layout, see push_run)
        auto anc_of = [&](uint32_t rank, uint32_t base) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + (p1 >> rank) - (p1f >> rank)) << 2), (int)anc);
        };
        // deepest rank (<= 5) any element can reach against the ORIGINAL ancestors (values only fall during the run), and the elements
        // that can get above rank 5 (bit e <-> element e) — rank by rank, stopping at the first rank nothing reaches, and the ancestors
        // are fetched AGAIN level by level below.  (Round 6 fetched all six ranks up front and kept them in registers — five address
        // computations and five waits on the LDS crossbar less per run that climbs high, and 22 % more time per label-filtered scan,
        // whose runs are short and mostly climb nowhere: 162.1 against 132.2 ms, profiles/r06/s8_ab_labels_10m.txt; rank 1 first and
        // the rest together: 137.4, s9.  The unfiltered regimes cannot tell the three apart.)
        uint32_t K = 0, hm;
        {
            const uint32_t kc = in ? c >> sb : IDENT;  // (every lane takes part in the fetches: they are cross-lane operations)
            auto below = [&](uint32_t rank, uint32_t base) -> uint64_t {
                const uint32_t ak = anc_of(rank, base);
                return __ballot(kc < (ak >> sb));
            };
            if (below(1, 0)) K = 1;
            if (K == 1 && below(2, 17)) K = 2;
            if (K == 2 && below(3, 26)) K = 3;
            if (K == 3 && below(4, 31)) K = 4;
            if (K == 4 && below(5, 34)) K = 5;
            hm = K == 5 ? (uint32_t)(below(6, 36) >> off) : 0u;
        }
        bool forced = false;
        // ---- above rank 5: the elements that can get there, one after another
        if (hm) {
            const uint32_t r = (uint32_t)lane;  // chain lane r <-> rank 5 + r (lane 0: what is handed down to rank 5)
            const uint32_t keymask = (1u << sb) - 1u;
            const uint32_t rsh = (5u + r) & 31u;
            const uint32_t fbase4 = (34u + 2u * r - (p1f >> rsh)) << 2;
            auto fresh_of = [&](uint32_t pp) -> uint32_t {
                return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((pp >> rsh) << 2) + fbase4), (int)anc);
            };
            uint32_t chain = fresh_of(p1f + (uint32_t)__builtin_ctz(hm));
            asm volatile("" : "+v"(chain));
            while (hm) {
                const uint32_t e = (uint32_t)__builtin_ctz(hm);
                hm &= hm - 1;
                const uint32_t elem = readlane_u32(entry, j + e);
                const uint32_t pp = p1f + e;
                const uint32_t ppn = p1f + (uint32_t)__builtin_ctz(hm | 0x80000000u);
                const uint32_t nxt = fresh_of(ppn);
                const bool cmp = (elem | keymask) < chain;
                const uint32_t bal = ((uint32_t)__ballot(cmp) >> 1) & 0x3FFFu;  // bit r-1 <-> rank 5 + r (ranks 6..19)
                const uint32_t t = (uint32_t)__builtin_ctz(~bal);
                const uint32_t up = wave_shl1(chain, 0);
                const uint32_t patched = r < t ? up : (r == t ? elem : chain);  // lane 0: the old rank-6 value when t >= 1
                if (r >= 1 && r <= t) {
                    const uint32_t dst = pp >> rsh;
                    if (spill) set1(dst, patched);
                    else l[dst] = patched;
                }
                const uint32_t out = readlane_u32(patched, 0);
                if ((uint32_t)lane == off + e && t >= 1) {
                    c = out;
                    forced = true;
                }
                const uint32_t shr = 32u - (uint32_t)__builtin_clz((pp ^ ppn) | 1u);  // ranks >= shr are shared (|1: clz(0))
                chain = 5u + r >= shr ? patched : nxt;
            }
            if (__ballot(forced)) K = 5;
        }
        // ---- ranks 5..1, all leaves at once
        auto level = [&](const uint32_t k, const uint32_t base) {
            const uint32_t ak = anc_of(k, base);  // the node's original value
            const uint32_t B1 = (1u << k) - 1u;
            const uint32_t comp = in ? (((c >> sb) << 7) | (forced ? 63u - i : 65u + i)) : IDENT;
            // exclusive prefix minimum inside the aligned block of 2^k lanes: the values shifted by one lane (a block's first lane starts
            // from the identity), then an inclusive scan whose steps cannot leave the block BY CONSTRUCTION (round 6; the row_shr steps
            // of round 3 needed a compare + select per step to cut them at the block boundaries): inside a quad quad_perm moves, across
            // the two quads of an 8-lane block the first quad's last lane broadcast into the second (bank mask 0b1010), whole 16-lane
            // rows with row_shr (a row boundary ends a DPP move anyway), and row 1 / 3 of a 32-lane block take lane 15 of the row below
            // (row_bcast:15, row mask 0b1010).  Every move is folded into its v_min_u32 (old = the identity).
#define VS_DMIN(v, src, ctrl, rmask, bmask) min((v), (uint32_t)__builtin_amdgcn_update_dpp((int)IDENT, (int)(src), (ctrl), (rmask), (bmask), false))
            uint32_t x = wave_shr1(comp, IDENT);
            if (((uint32_t)lane & B1) == 0) x = IDENT;
            if (k == 2 || k == 3) {
                x = VS_DMIN(x, x, 0x90 /* quad_perm [0,0,1,2] */, 0xF, 0xF);
                x = VS_DMIN(x, x, 0x44 /* quad_perm [0,1,0,1] */, 0xF, 0xF);
            }
            if (k == 3) {
                const uint32_t q3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xFF /* quad_perm [3,3,3,3] */, 0xF, 0xF, true);
                x = VS_DMIN(x, q3, 0x114 /* row_shr:4 */, 0xF, 0xA);
            }
            if (k >= 4) {
                x = VS_DMIN(x, x, 0x111, 0xF, 0xF);
                x = VS_DMIN(x, x, 0x112, 0xF, 0xF);
                x = VS_DMIN(x, x, 0x114, 0xF, 0xF);
                x = VS_DMIN(x, x, 0x118, 0xF, 0xF);
            }
            if (k >= 5) x = VS_DMIN(x, x, 0x142 /* row_bcast:15 */, 0xA, 0xF);
#undef VS_DMIN
            const uint32_t cur = min(x, ((ak >> sb) << 7) | 64u);  // what the node holds when this push arrives
            const uint32_t tb = cur & 127u;
            const uint32_t src = (tb < 64u ? 63u - tb : tb - 65u) + off;  // lane of the push that brought it
            const uint32_t hv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)c);
            const uint32_t curv = tb == 64u ? ak : hv;
            const bool wins = in && comp < cur;
            const bool last = in && ((((uint32_t)lane & B1) == B1) || i + 1 == n);  // last push of the node: its final value
            if (last) {
                const uint32_t fin = wins ? c : curv;
                if (fin != ak) put(k, p1 >> k, fin);
            }
            if (wins) {
                c = curv;
                forced = true;
            }
        };
        if (K >= 5) level(5, 34);
        if (K >= 4) level(4, 31);
        if (K >= 3) level(3, 26);
        if (K >= 2) level(2, 17);
        if (K >= 1) level(1, 0);
        if (in) put(0, p1, c);
    }
    // pre_n / pre_anc: a wide load the caller already issued for the first run (pre_n = first_run(c)), or pre_n = 0
    __device__ __forceinline__ void push_run(uint32_t entry, uint32_t c, uint32_t pre_n, uint32_t pre_anc) {
        uint32_t j = 0;
        while (j < c) {
            const uint32_t p1f = len + 1;  // (position of the run's first leaf) + 1
            // a run stays on one heap level, has at most 32 leaves, and needs a heap of >= 64 entries / depth <= 19
            const uint32_t room = (2u << (31u - (uint32_t)__builtin_clz(p1f))) - p1f;  // leaves left on this level
            const uint32_t n = min(min(c - j, 32u), room);
            if (len < 64 || p1f >= (1u << 19)) {  // tiny or huge heap: one at a time
                push(readlane_u32(entry, j));
                ++j;
                continue;
            }
            const uint32_t p1l = p1f + n - 1;
            const bool spill = p1l > hl;  // some leaf (hence possibly some parent) lives in the spill array
            // ---- wide load of every distinct ancestor of the run (the first run's may already be in flight)
            const uint32_t anc = (j == 0 && pre_n == n) ? pre_anc : wide_load(n);
            if (n >= 6) {  // level by level; short runs cost less element by element
                push_run_scan(entry, j, n, anc, spill);
                j += n;
                len += n;
                wave_sync();
                continue;
            }
            const uint32_t r = (uint32_t)lane;  // chain lane = ancestor rank (ranks 1..19 are ancestors; lane 0 stands for the leaf)
            // rank-r ancestor of leaf p1 as loaded at the start of the run: wide-load lane wl_base(r) + (p1 >> r) - (p1f >> r)
            // (any lane index is a legal bpermute source; what lanes without a rank read is never used)
            const uint32_t fbase4 = (wl_base() - (p1f >> (r & 31u))) << 2;
            auto fresh_of = [&](uint32_t p1) -> uint32_t {
                return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((p1 >> (r & 31u)) << 2) + fbase4), (int)anc);
            };
            // An element that is not smaller than the ORIGINAL parent of its leaf stays on the leaf whatever the earlier
            // pushes of the run do (a push only ever lowers the values on its path: a position receives the pushed element or
            // its own parent's old value), and no push reads a leaf of the run: those elements — about a third — are stored
            // right away, all at once; only the climbers go through the sequential loop below.  Between two climbers the
            // skipped leaves touch no ancestor, so the shared part of the chain carries over exactly as between neighbours.
            const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j + (uint32_t)lane) << 2), (int)entry);
            const uint32_t myp1 = p1f + (uint32_t)lane;
            const uint32_t par = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((myp1 >> 1) - (p1f >> 1)) << 2), (int)anc);
            const bool inrun = (uint32_t)lane < n;
            const bool climbs = inrun && (mine >> sb) < (par >> sb);
            uint32_t cm = (uint32_t)__ballot(climbs);  // (n <= 32)
            if (inrun && !climbs) {
                if (spill) set1(myp1, mine);
                else l[myp1] = mine;
            }
            if (cm) {
                const uint32_t keymask = (1u << sb) - 1u;
                uint32_t chain = fresh_of(p1f + (uint32_t)__builtin_ctz(cm));
                asm volatile("" : "+v"(chain));  // the chain is complete before the loop
                while (cm) {
                    const uint32_t e = (uint32_t)__builtin_ctz(cm);
                    cm &= cm - 1;
                    const uint32_t elem = readlane_u32(entry, j + e);
                    const uint32_t p1 = p1f + e;
                    const uint32_t p1n = p1f + (uint32_t)__builtin_ctz(cm | 0x80000000u);  // next climber's leaf (any leaf after the last)
                    const uint32_t nxt = fresh_of(p1n);  // in flight during this push
                    // key(elem) < key(chain)  <=>  (elem | keymask) < chain   (keys are the bits above the slot handle)
                    const bool cmp = (elem | keymask) < chain;
                    const uint32_t bal = ((uint32_t)__ballot(cmp) >> 1) & 0x7FFFFu;  // bit r-1 <-> ancestor r (ranks 1..19)
                    const uint32_t t = (uint32_t)__builtin_ctz(~bal);                  // leading run of ancestors that move down
                    // after the push position p1 >> k holds the old rank k+1 value for k < t and the element for k == t: lane k
                    // writes it, and the same values are the chain the next push sees from rank sh upwards
                    const uint32_t up = wave_shl1(chain, 0);  // rank r+1 value
                    const uint32_t patched = r < t ? up : (r == t ? elem : chain);
                    if (r <= t) {
                        const uint32_t dst = p1 >> (r & 31u);
                        if (spill) set1(dst, patched);
                        else l[dst] = patched;
                    }
                    const uint32_t sh = 32u - (uint32_t)__builtin_clz(p1 ^ p1n);
                    chain = r >= sh ? patched : nxt;
                }
            }
            j += n;
            len += n;
            wave_sync();
        }
    }

    // ---- BinaryHeap::pop: Vec::pop, swap with data[0], sift_down_to_bottom(0), sift_up.  (len > 0; the caller has
    // already read data[0])
    __device__ __forceinline__ void pop() {
        const uint32_t last = len - 1;
        const bool all_lds = last < hl;
        // the former last element is only needed once the hole has reached a leaf: its load (L2 when the heap spills)
        // stays in flight during the sift-down rounds
        const uint32_t item_v = all_lds ? l[last + 1] : get(last);
        len = last;
        if (len == 0) return;
        const uint32_t end = len;
        uint32_t root = 0, pos = 0;
        uint32_t pkey = 0;  // key of the value now stored in the parent of `root` (0 for the heap root: never moves)
        // (rounds cut at the LDS / spill-array boundary instead — node levels 0-5, 6-7, then 8-13 from the spill array — were tried in
        // round 4: exact, and no faster at 10M / 50M, profiles/r04/s3_ab_slotmap_*.txt; a heap of up to 8190 entries already needs only one
        // round that reads the spill array)
        for (;;) {
            // one 6-level subtree per iteration: lane j < 63 is the node with relative heap index j
            const uint32_t aidx = ((root + 1) << lvl()) + offm1();
            const uint32_t c = 2 * aidx + 1;
            const bool exists = aidx < end, have1 = c < end, have2 = c + 1 < end;
            // (the subtree under the heap root — nodes 0..62, children up to 126 — lies in LDS whenever hl >= 127: the first round of
            // every pop then takes the all-LDS path, a wave-uniform choice instead of a per-lane "LDS or spill array" around its load
            // and its store; round 6)
            const bool lo = all_lds || (root == 0 && hl >= 127u);
            uint32_t le = 0, ri = 0;
            if (have1) {
                if (lo || c < hl) {
                    const uint2 p = *reinterpret_cast<const uint2*>(l + c + 1);
                    le = p.x;
                    ri = p.y;
                } else {
                    const uint64_t p = gload64u(reinterpret_cast<const uint64_t*>(g + (c - hl)));
                    le = (uint32_t)p;
                    ri = (uint32_t)(p >> 32);
                }
            }
            // child += (data[child] <= data[child+1]); Reverse => right.d <= left.d picks the right child
            const bool pick = have2 && (ri >> sb) <= (le >> sb);
            const uint32_t cv = pick ? ri : le;
            const uint64_t B = __ballot(pick);
            // a node is on the path iff every ancestor inside this subtree chose the child leading to it
            const bool onpath = exists && (((uint32_t)B & amask) == dpat);
            const uint64_t pm = __ballot(onpath);
            if (onpath && have1) {
                if (lo) l[aidx + 1] = cv;
                else set(aidx, cv);
            }
            wave_sync();
            const uint32_t jd = 63u - (uint32_t)__builtin_clzll(pm);  // deepest path node (pm != 0: the root exists)
            const uint32_t ad = readlane_u32(aidx, jd);
            const uint64_t h1 = __ballot(have1);
            if (!((h1 >> jd) & 1ull)) {  // a leaf: the hole ends here
                pos = ad;
                if (jd > 0) pkey = readlane_u32(cv, (jd - 1) >> 1) >> sb;  // what its parent just received
                break;
            }
            pkey = readlane_u32(cv, jd) >> sb;
            root = 2 * ad + 1 + (uint32_t)((B >> jd) & 1ull);  // jd is on the subtree's last level: descend
            if (2 * root + 1 >= end) {  // ... unless the child is a leaf (a heap of 4096..8190 entries ends every path here)
                pos = root;
                break;
            }
        }
        // sift_up(0, pos) of the former last element: it only moves when it is smaller than the new parent value
        const uint32_t item = rfl(item_v);
        const uint32_t ikey = item >> sb;
        if (ikey < pkey) {
            if (all_lds) sift_up_lds(pos, item);
            else sift_up_gen(pos, item);
        } else {
            if (lane == 0) {
                if (all_lds) l[pos + 1] = item;
                else set(pos, item);
            }
            wave_sync();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// visited: Vec<ListSearchNeighbor> kept sorted (AM/graph/mod.rs:76,167-168,181).  VR > 0: in registers, entry i is
// lane i % 64 of (h[i / 64], n[i / 64]).  VR == 0: ring buffer of (flags << 62 | hamming << 32 | node) in LDS; the two flag
// bits (VIS_DEAD: heap tid deleted, VIS_HIDDEN: heap tuple invisible to the snapshot) are what consume() needs to know about
// the node: they are looked up when the node is VISITED, so a run of consume() calls never waits for memory.
#define VIS_DEAD 2u
#define VIS_HIDDEN 1u
#define VIS_HAM_MASK 0x3FFFFFFFu
// ---------------------------------------------------------------------------------------------------------------
template <int VR>
struct Visited {
    static constexpr int NR = VR > 0 ? VR : 1;
    uint32_t h[NR], n[NR];
    uint64_t* ring;
    uint32_t vcapv, head, len;  // ring capacity (any size), index of entry 0, number of entries
    int lane;

    __device__ __forceinline__ void init(int lane_, uint64_t* ring_, uint32_t vcap) {
        lane = lane_;
        ring = ring_;
        vcapv = vcap;
        head = 0;
        len = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) { h[r] = 0; n[r] = 0; }
    }
    __device__ __forceinline__ uint32_t capacity() const { return VR > 0 ? 64u * VR : vcapv; }
    // ring slot of entry i (i <= capacity; i == 0xFFFFFFFF means "one before entry 0")
    __device__ __forceinline__ uint32_t slot(uint32_t i) const {
        if (i == 0xFFFFFFFFu) return head ? head - 1 : vcapv - 1;
        const uint32_t x = head + i;
        return x >= vcapv ? x - vcapv : x;
    }
    // hamming of entry i (i < len, uniform)
    __device__ __forceinline__ uint32_t ham_at(uint32_t i) const {
        if (VR > 0) {
            uint32_t v = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if ((i >> 6) == (uint32_t)r) v = readlane_u32(h[r], i & 63u);
            return v;
        }
        return rfl((uint32_t)(ring[slot(i)] >> 32) & VIS_HAM_MASK);
    }
    // visited.insert(partition_point(|x| *x < new), new): before the first element >= new  (caller checked capacity)
    __device__ __forceinline__ void insert(uint32_t hd, uint32_t node, uint32_t flags = 0) {
        if (VR > 0) {
            uint32_t idx = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if ((uint32_t)r * 64u < len) {
                    const uint32_t gi = (uint32_t)r * 64u + (uint32_t)lane;
                    idx += (uint32_t)__popcll(__ballot(gi < len && h[r] < hd));
                }
            }
#pragma unroll
            for (int r = NR - 1; r >= 0; --r) {
                if ((uint32_t)r * 64u <= len && (uint32_t)r * 64u + 63u >= idx) {  // register holds a moved / new entry
                    const uint32_t ch = r > 0 ? readlane_u32(h[r > 0 ? r - 1 : 0], 63) : 0u;
                    const uint32_t cn = r > 0 ? readlane_u32(n[r > 0 ? r - 1 : 0], 63) : 0u;
                    const uint32_t sh = wave_shr1(h[r], ch), sn = wave_shr1(n[r], cn);
                    const uint32_t gi = (uint32_t)r * 64u + (uint32_t)lane;
                    h[r] = gi > idx ? sh : (gi == idx ? hd : h[r]);
                    n[r] = gi > idx ? sn : (gi == idx ? node : n[r]);
                }
            }
            len++;
            return;
        }
        if (len < WAVE) {
            // the whole list in one wave-wide read (the usual case: search_list_size + the rows not yet consumed stay below 64 for
            // the operating points of small lists): lane i holds entry i, the entries from the insertion point on move one slot back —
            // lane j writes slot j with the entry lane j - 1 read (a DPP shift), lane idx the new one.  One LDS read, one write, no loop.
            const uint32_t x = head + (uint32_t)lane;
            const uint32_t sl = x >= vcapv ? x - vcapv : x;  // (vcapv >= 64)
            uint64_t e = ~0ull;
            if ((uint32_t)lane < len) e = ring[sl];
            const bool lt = (uint32_t)lane < len && ((uint32_t)(e >> 32) & VIS_HAM_MASK) < hd;
            const uint32_t idx = (uint32_t)__popcll(__ballot(lt));
            const uint32_t plo = wave_shr1((uint32_t)e, 0), phi = wave_shr1((uint32_t)(e >> 32), 0);
            const uint64_t nw = ((uint64_t)(hd | (flags << 30)) << 32) | node;
            if ((uint32_t)lane >= idx && (uint32_t)lane <= len) ring[sl] = (uint32_t)lane == idx ? nw : (((uint64_t)phi << 32) | plo);
            len++;
            wave_sync();
            return;
        }
        // (a register-resident pass for lists of up to four chunks — every entry read once, the tail rewritten one slot further back — was
        // built and measured in round 6: 190.3 against 162.6 ms per 262 144 label-filtered scans at search_list_size 100, 10M x 1536,
        // profiles/r06/s7_ab_labels_10m.txt.  The loop below moves the SHORTER side of the insertion point, and a node that is visited
        // late in a scan lands near the front of a long list.)
        uint32_t idx = 0;
        for (uint32_t base = 0; base < len; base += WAVE) {
            const uint32_t i = base + lane;
            const bool lt = i < len && ((uint32_t)(ring[slot(i)] >> 32) & VIS_HAM_MASK) < hd;
            idx += (uint32_t)__popcll(__ballot(lt));
        }
        if (2 * idx < len) {  // move [0, idx) one slot towards the front, lowest chunk first
            for (uint32_t base = 0; base < idx; base += WAVE) {
                const uint32_t i = base + lane;
                uint64_t e = 0;
                if (i < idx) e = ring[slot(i)];
                wave_sync();
                if (i < idx) ring[slot(i - 1)] = e;
                wave_sync();
            }
            head = head ? head - 1 : vcapv - 1;
        } else {  // move [idx, len) one slot towards the back, highest chunk first
            uint32_t hi = len;
            while (hi > idx) {
                const uint32_t lo = (hi - idx > WAVE) ? hi - WAVE : idx;
                const uint32_t i = lo + lane;
                uint64_t e = 0;
                if (i < hi) e = ring[slot(i)];
                wave_sync();
                if (i < hi) ring[slot(i + 1)] = e;
                wave_sync();
                hi = lo;
            }
        }
        if (lane == 0) ring[slot(idx)] = ((uint64_t)(hd | (flags << 30)) << 32) | node;
        len++;
        wave_sync();
    }
    // visited.remove(0) (len > 0)
    __device__ __forceinline__ void pop_front(uint32_t& hd, uint32_t& node, uint32_t& flags) {
        flags = 0;
        if (VR > 0) {
            hd = readlane_u32(h[0], 0);
            node = readlane_u32(n[0], 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if ((uint32_t)r * 64u < len) {
                    const uint32_t ch = r + 1 < NR ? readlane_u32(h[r + 1 < NR ? r + 1 : 0], 0) : 0u;
                    const uint32_t cn = r + 1 < NR ? readlane_u32(n[r + 1 < NR ? r + 1 : 0], 0) : 0u;
                    h[r] = wave_shl1(h[r], ch);
                    n[r] = wave_shl1(n[r], cn);
                }
            }
            len--;
            return;
        }
        const uint64_t front = ring[head];
        hd = rfl((uint32_t)(front >> 32));
        flags = hd >> 30;
        hd &= VIS_HAM_MASK;
        node = rfl((uint32_t)front);
        head = head + 1 == vcapv ? 0 : head + 1;
        len--;
    }
};

// QL: the query code is read from its LDS copy (8 NCH words, zero padded) instead of 4 NCH registers per lane — the variant that
// has to fit 64 VGPRs (8 waves per SIMD)
// XW: the rows are exactly 8 NCH words wide (no lane ever reads past a row: the width tests fall away)
template <int NCH, bool QL = false, bool XW = false>
__device__ __forceinline__ uint32_t ham_row_reg(const uint64_t* __restrict__ row, const ulonglong2 (&qv)[NCH > 0 ? NCH : 1],
                                                const uint64_t* qc_l, int l4, uint32_t code_stride, bool active, bool stream) {
    uint32_t acc = 0;
    if (NCH > 0) {
        if (active) {
            ulonglong2 r[NCH > 0 ? NCH : 1];
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
                r[t] = (!XW && w >= code_stride) ? make_ulonglong2(0, 0)
                       : stream ? load_stream16(row + w) : *reinterpret_cast<const ulonglong2*>(row + w);
            }
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const ulonglong2 qq = QL ? *reinterpret_cast<const ulonglong2*>(qc_l + 2u * (uint32_t)l4 + 8u * (uint32_t)t) : qv[t];
                acc += (uint32_t)__popcll(r[t].x ^ qq.x) + (uint32_t)__popcll(r[t].y ^ qq.y);
            }
        }
        return quad_sum(acc);
    }
    return ham_row4(row, qc_l, l4, code_stride, active);
}

// two rows at once (all loads issued before the first popcount)
template <int NCH, bool QL = false>
__device__ __forceinline__ void ham_row_reg2(const uint64_t* __restrict__ row_a, const uint64_t* __restrict__ row_b,
                                             const ulonglong2 (&qv)[NCH > 0 ? NCH : 1], const uint64_t* qc_l, int l4, uint32_t code_stride, bool act_a,
                                             bool act_b, bool stream, uint32_t& da, uint32_t& db) {
    constexpr int N = NCH > 0 ? NCH : 1;
    ulonglong2 ra[N], rb[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
        const bool in = w < code_stride;
        ra[t] = !(act_a && in) ? make_ulonglong2(0, 0) : stream ? load_stream16(row_a + w) : *reinterpret_cast<const ulonglong2*>(row_a + w);
        rb[t] = !(act_b && in) ? make_ulonglong2(0, 0) : stream ? load_stream16(row_b + w) : *reinterpret_cast<const ulonglong2*>(row_b + w);
    }
    uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const ulonglong2 qq = QL ? *reinterpret_cast<const ulonglong2*>(qc_l + 2u * (uint32_t)l4 + 8u * (uint32_t)t) : qv[t];
        acc_a += (uint32_t)__popcll(ra[t].x ^ qq.x) + (uint32_t)__popcll(ra[t].y ^ qq.y);
        acc_b += (uint32_t)__popcll(rb[t].x ^ qq.x) + (uint32_t)__popcll(rb[t].y ^ qq.y);
    }
    da = quad_sum(act_a ? acc_a : 0u);
    db = quad_sum(act_b ? acc_b : 0u);
}

// minimum over the wave (DPP row reduction + 4 readlanes; no LDS)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111 /*row_shr:1*/, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112 /*row_shr:2*/, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114 /*row_shr:4*/, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118 /*row_shr:8*/, 0xF, 0xF, false));
    return min(min(readlane_u32(v, 15), readlane_u32(v, 31)), min(readlane_u32(v, 47), readlane_u32(v, 63)));
}

// MINW = waves per SIMD the register allocator must leave room for (1 = unconstrained): in the table-less regime the
// kernel is occupancy bound and LDS no longer limits it, so fewer VGPRs (some cold values in scratch) can pay.
// BUILD = greedy_search_for_build (AM/graph/mod.rs:285-327): no rows are consumed; the sorted visited list (capped at the
// ring capacity, farthest entry dropped) is the result.
// FULL = label keys and / or a visibility mask may be present; the plain instantiation (neither) leaves their pointers, counters
// and branches out of a kernel whose scalar registers are its tightest resource.
// One scan.  `slot` names the per-scan regions of the workspace (heap spill array, dedup table in HBM): the scan's own index when
// the launch has one workgroup per scan, the workgroup's index when the grid is persistent (s.persist: as many workgroups as the
// chip holds at once, each taking scan after scan from a counter — the regions are then reused by the scans a workgroup runs, the
// workspace is a few hundred MB whatever the batch size, and no region is claimed with an atomic).
// OPT: what the launch wrapper knows about the index and states at compile time (round 6: every one of these was a branch, a live
// scalar register or an exec-mask region in a kernel that spilled 226 of them): OPT_XW = code rows are exactly 8 NCH words,
// OPT_R1 = num_neighbors <= 64, i.e. a neighbor list is ONE wave-wide chunk, OPT_NS64 = neighbor rows are 64 ids apart.
// OPT_RS = RESUMABLE scans (the amgettuple continuations of the scan pools, AM/scan.rs:162-174,370-405): one workgroup per pool slot, the
// slot's regions of the dedup tables / heap spill array live on between launches, and the on-chip state (heap top, occupancy bits,
// visited ring, a header of counters) is restored from and saved to s.resume; a launch continues the scan for M more rows and records
// the work counters per emitted row (s.row_stats).  status[q] is the run mask on entry (only the marked slots run).
#define OPT_XW 1
#define OPT_R1 2
#define OPT_NS64 4
#define OPT_RS 8
#define OPT_G2 16
template <int NCH, int VR, bool TIMING, int MINW, bool BUILD, bool FULL, int VG, int OPT>
__device__ __forceinline__ void fast_scan(const FastArgs& a, const uint32_t q, const uint32_t slot) {
    constexpr bool XW = (OPT & OPT_XW) != 0 && NCH > 0;
    constexpr bool R1 = (OPT & OPT_R1) != 0;
    constexpr bool RS = (OPT & OPT_RS) != 0;
    static_assert(!RS || (VG == 3 && VR == 0 && !BUILD), "resumable scans: 16-bit tables, LDS-ring visited list");
    // (vector-register copies of the array bases, see in_vgpr; the row stride is a compile-time constant where the launch says so)
    // (the instantiation with label keys has the neighbors' masks in flight next to the rows: it keeps its vector registers for those)
    const uint64_t* const codes_v = FULL ? a.codes : in_vgpr(a.codes);
    const uint32_t* const nbrs_v = FULL ? a.nbrs : in_vgpr(a.nbrs);
    const uint64_t* const tids_v = FULL ? a.tids : in_vgpr(a.tids);
    const uint32_t cstride = XW ? 8u * (uint32_t)NCH : a.code_stride;
    const uint32_t nstride = (OPT & OPT_NS64) ? 64u : a.nbr_stride;
    const uint32_t R_v = in_vgpr(a.R);  // (only ever compared with lane indexes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int lane_v = threadIdx.x;
    // (persistent grid: what a scan derives from its lane index is derived again by the next scan instead of being carried
    // across the whole kernel — the register allocation of one scan stays what it is with one workgroup per scan)
    asm volatile("" : "+v"(lane_v));
    int lane = lane_v;  // (re-read as an opaque value at the top of every iteration of the main loop, see there)
    const FastLaunch& s = a.s;
    // (the bitmap variants only exist for the table-less regime: the LDS-table code paths are not compiled into them)
    const uint32_t lhv = VG ? 0u : s.lh;
    const uint32_t onlyfv = VG ? 0u : s.only_failed;  // (second attempts run the instantiation that clears its tables)
    const uint32_t rcv = VG >= 2 ? 0u : s.rc;
    if (onlyfv && s.status[q] == 0) return;  // (wave-uniform) finished by the first launch
    if (RS && s.status[q] == 0) return;      // (wave-uniform) not listed in this round
    uint32_t* const rs = RS ? s.resume + (size_t)q * s.resume_stride : nullptr;
    const bool resumed = RS && rfl(rs[RSF_INIT]) == 1u;
    if (s.timeline && lane == 0) s.timeline[2 * (size_t)q] = wall_clock64();

    // ---- LDS carve ----  Everything of a size known at compile time comes first, so its addresses are immediates of the ds
    // instructions instead of scalar registers (round 6); the arrays sized by the launch follow.
    // LEAN (the 7-waves-per-SIMD variant of the 16-bit tables: 28 scans per CU need <= 5 632 B of LDS each): a survivor's distance is
    // merged into its slot word instead of an array of its own, and the plain instantiation has no room for label keys it never reads
    constexpr bool LEAN = MINW == 7 && VG == 3;
    constexpr bool QL = NCH > 0 && MINW >= 6;
    uint32_t* surv_id = reinterpret_cast<uint32_t*>(smem);            // 64
    uint32_t* surv_slot = surv_id + 64;                               // 64
    uint32_t* surv_d = LEAN ? surv_slot : surv_slot + 64;             // 64 (LEAN: the same words)
    uint32_t* arb = surv_d + 64;                                      // ARB_SLOTS rank counters of the global dedup table (zero between uses; none with the slot bitmap)
    constexpr uint32_t ARB_N = VG >= 2 ? 0u : (uint32_t)ARB_SLOTS;
    uint64_t* qc_fix = reinterpret_cast<uint64_t*>(arb + ARB_N);      // 8 NCH words (QL: the register-capped variants' copy of the query code; 16-byte aligned)
    int16_t* ql = reinterpret_cast<int16_t*>(qc_fix + (QL ? 8 * NCH : 0));  // MAX_QLABELS
    uint32_t* hp = reinterpret_cast<uint32_t*>(ql + (LEAN && !FULL ? 0 : MAX_QLABELS));  // hl + 1 (hl + 1 is a power of two >= 64; 8-byte aligned)
    uint32_t* lhash = hp + (s.hl + 1);                                // lh (multiple of 4)
    uint64_t* ring = reinterpret_cast<uint64_t*>(lhash + lhv);        // vcap entries (VR == 0 only)
    // (optional) cache of ids known to be in the dedup table: a hit answers a duplicate probe without touching the table in HBM
    uint32_t* rc = reinterpret_cast<uint32_t*>(ring + (VR > 0 ? 0 : s.vcap));
    const uint32_t rcm = rcv - 1u;  // (rcv: 0 or a power of two)
    // (VG == 1) one bit per bucket of the dedup table in HBM: set once this scan has written the bucket.  A bucket whose bit is clear is
    // neither cleared nor read — its memory holds whatever an earlier scan left there — and counts as four empty slots.
    // (VG == 2) one bit per SLOT: the table is open addressing with linear probing at slot granularity, and which slots are occupied
    // is known on chip — an id whose home slot is free is new and is stored there without a load (most new ids: the expected share
    // is 1 - load factor), the others are compared with the occupied run that starts at their home slot (one 16-byte load per
    // 4-slot group the run touches), a free slot is claimed with one ds_or; no clears, no rank counters
    uint32_t* vmap = rc + rcv;  // s.vwords
    // the generic width's copy of the query code (code_stride words) is the one array of a size only the launch knows that wants 16 bytes
    uint64_t* qc_l = QL ? qc_fix : reinterpret_cast<uint64_t*>(smem + ((((vmap + s.vwords) - reinterpret_cast<uint32_t*>(smem)) * 4u + 15u) & ~15u));

    int l4 = lane & 3;
    // code rows are read once per scan: non-temporal loads, worth 10 % at 50M (profiles/r04/s5_ab_nt_rows_50m.txt).  A compile-time
    // constant since round 6: as a launch flag every row load was two instructions behind a branch.
    constexpr bool stream_rows = true;
    // (neighbor rows, the neighbors' label masks and heap tids are read once per scan too and go through non-temporal loads as well:
    // measured neutral at 50M — 159.94 against 159.93 ms, profiles/r04/s5_ab_nt_rows_50m.txt — where the code rows' are worth 10 %)
    constexpr bool G2 = NCH == 3 && VR == 0 && (MINW == 5 || (OPT & OPT_G2) != 0) && !BUILD && !TIMING;  // two code rows per 4-lane group in flight
    ulonglong2 qv[NCH > 0 ? NCH : 1];
    if (QL) {
        qv[0] = make_ulonglong2(0, 0);
        for (uint32_t w = lane; w < 8u * (uint32_t)NCH; w += WAVE)
            qc_l[w] = w < a.code_stride ? s.qcodes[(size_t)q * cstride + w] : 0ull;
    } else if (NCH > 0) {
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const uint32_t w = 2u * (uint32_t)l4 + 8u * (uint32_t)t;
            qv[t] = w < a.code_stride
                        ? *reinterpret_cast<const ulonglong2*>(s.qcodes + (size_t)q * cstride + w)
                        : make_ulonglong2(0, 0);
        }
    } else {
        for (uint32_t w = lane; w < a.code_stride; w += WAVE) qc_l[w] = s.qcodes[(size_t)q * cstride + w];
    }
    for (uint32_t i = 4u * lane; i < lhv; i += 4u * WAVE)
        *reinterpret_cast<uint4*>(lhash + i) = make_uint4(VS_EMPTY, VS_EMPTY, VS_EMPTY, VS_EMPTY);
    if (lane == 0 && !resumed) hp[0] = 0;  // heap sentinel
    for (uint32_t i = lane; i < ARB_N; i += WAVE) arb[i] = 0;
    for (uint32_t i = lane; i < rcv; i += WAVE) rc[i] = VS_EMPTY;
    if (VG && !resumed)
        for (uint32_t i = lane; i < s.vwords; i += WAVE) vmap[i] = 0;
    // (resumable scans) the image of a saved scan: everything from the heap top to the occupancy bits is one contiguous stretch of LDS
    const uint32_t image_words = RS ? (uint32_t)((vmap + s.vwords) - hp) : 0u;
    if (resumed)  // (16 bytes per lane: the image is padded to whole uint4, hp and the saved image are 16-byte aligned)
        for (uint32_t i = 4u * lane; i < image_words; i += 4u * WAVE) *reinterpret_cast<uint4*>(hp + i) = *reinterpret_cast<const uint4*>(rs + RSF_HDR + i);
    const uint8_t* const visible = FULL ? s.visible : nullptr;
    const bool labels_some = FULL && s.qlabel_off != nullptr;  // LabeledVector.labels is Some (AM/labels/mod.rs:222-236)
    uint32_t nql = 0;
    bool wide_key = false;  // a key of more than MAX_QLABELS labels does not fit the LDS slot: the general kernel runs the scan
    if (labels_some) {
        const uint32_t lb = s.qlabel_off[q], le = s.qlabel_off[q + 1];
        wide_key = le - lb > (uint32_t)MAX_QLABELS;
        nql = min(le - lb, (uint32_t)MAX_QLABELS);
        for (uint32_t i = lane; i < nql; i += WAVE) ql[i] = s.qlabels[lb + i];
    }
    const bool has_label_filter = labels_some && nql > 0;  // AM/scan.rs:189
    wave_sync();
    // the query's labels as a mask, for an index whose node label sets are masks (<= 64 distinct labels, each with its bit): a
    // query label that occurs nowhere in the index cannot be in any node's set and simply contributes no bit
    uint64_t qmask = 0;
    if (has_label_filter && a.label_mask) {
        uint64_t mine = 0;
        if ((uint32_t)lane < nql) {
            const uint32_t bit = a.label_bit[(uint16_t)ql[lane]];
            if (bit < 64u) mine = 1ull << bit;
        }
        // OR over the (<= 64) lanes: a butterfly of DPP-free shuffles is not worth it for something done once per scan
        for (int o = 32; o >= 1; o >>= 1) mine |= __shfl_xor(mine, o, WAVE);
        qmask = mine;
    }

    FastHeap<(MINW >= 7)> heap;
    heap.l = hp;
    // (second attempt with one workgroup per scan: set with the pool region)
    heap.g = (s.persist || RS) ? s.heap_g + (size_t)slot * s.gstride : (onlyfv ? s.heap_g : s.heap_g + (size_t)q * s.gstride);
    heap.hl = s.hl;
    heap.sb = s.sb;
    heap.init(lane);
    Visited<VR> vis;
    vis.init(lane, ring, VR > 0 ? 64u * VR : s.vcap);

    const uint32_t slot_limit = lhv - lhv / 8;  // stop at 87.5 % load: the scan is handed to the general kernel
    const uint32_t smask = (1u << s.sb) - 1u;
    uint32_t emitted = 0, status = wide_key ? (uint32_t)OVF_KEY : 0u, nins = 0, hmax = 0;
    // (work counters kept in scalar registers: visits, candidates (= quantized distance computations in this kernel: every candidate is
    // scored exactly once), and the pops of the visited list; SbqNode reads = pops + visits + ids that entered the dedup set, put
    // together when the scan ends — round 6: each dropped counter is a register and an add per visit less)
    uint32_t st_visits = 0, st_cand = 0, st_pops = 0;
    // optional phase clock (s_memtime): 0 pop, 1 row wait, 2 visited, 3 dedup, 4 gather, 5 push, 6 other
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tmark = TIMING ? __builtin_readcyclecounter() : 0;
    auto lap = [&](int k) {
        if (TIMING) {
            const uint64_t now = __builtin_readcyclecounter();
            ph[k] += now - tmark;
            tmark = now;
        }
    };

    // ---- HashSet::insert.  Normal mode: the first CAS is ISSUED by the caller so independent work can overlap its
    // latency, finish_insert completes the probe sequence.  Frozen mode (LDS table at its load limit): read-only
    // probe of the LDS table, then insert into the per-scan global overflow table.
    uint32_t* ghash = s.ghash;
    const uint32_t gbuckets = s.gcap >> 2;  // (any multiple of four slots: the home bucket is a multiply-shift, not a mask)
    uint32_t nins_g = 0;
    bool g_open = false, region = false;
    // global dedup overflow table: claimed from the pool on first need
    auto claim_region = [&]() -> bool {
        if (region) return true;
        if (s.persist || RS) {  // the workgroup's own region (the launch wrapper holds pool_slots >= the grid) / the pool slot's
            region = true;
            ghash = s.ghash + (size_t)slot * (VG == 3 ? s.gregion : s.gcap);
            return true;
        }
        uint32_t pslot = 0;
        if (lane == 0) pslot = atomicAdd(s.pool_counter, 1u);
        pslot = rfl(pslot);
        if (pslot >= s.pool_slots) {
            status |= OVF_POOL;
            return false;
        }
        region = true;
        ghash = s.ghash + (size_t)pslot * (VG == 3 ? s.gregion : s.gcap);
        if (onlyfv) heap.g = s.heap_g + (size_t)pslot * s.gstride;
        return true;
    };
    auto hash_home = [&](uint32_t nid) -> uint32_t { return (uint32_t)(((uint64_t)hash_u32(nid) * lhv) >> 32); };
    // handle -> node id.  node_load only ISSUES the read (LDS, or L2 for ids in the overflow table); the value is made
    // uniform with rfl() where it is needed, so the latency overlaps whatever runs in between.
    // entries are plain node ids; VS_EMPTY marks an empty slot of a table its claiming wave has cleared (no bitmap: second attempts,
    // build mode, LDS-table regime overflow)
    auto g_empty = [&](uint32_t v) -> bool { return v == VS_EMPTY; };
    // ---- VG == 3: 16-bit entries.  The id is mapped by a bijection on qd bits (multiply, xor-shift, multiply, xor-shift: each step is
    // invertible mod 2^qd) to x; the bucket of eight slots is x >> qk, the entry the remainder x & (2^qk - 1), the preferred slot x & 7.
    // (bucket, entry) names the id: q_inv gives it back, so a heap entry's handle (the slot) still finds the node without a second array.
    const uint32_t qdm = in_vgpr(VG == 3 ? (s.qd >= 32 ? 0xFFFFFFFFu : (1u << s.qd) - 1u) : 0u);
    const uint32_t qxs = in_vgpr((s.qd + 1u) >> 1);  // (2 qxs >= qd: x ^= x >> qxs is its own inverse)
    const uint32_t qk_v = in_vgpr(s.qk), rmask_v = in_vgpr((1u << s.qk) - 1u);
    const uint32_t ovf_lim_v = in_vgpr(s.ocap - s.ocap / 4);  // (the overflow table's load limit)
    // (the multipliers are compile-time constants — the kernel lives at its scalar-register limit; the launch wrapper checks that the
    // host's copies agree)
    auto q_fwd = [&](uint32_t id) -> uint32_t {
        uint32_t x = (id * VS_Q16_A) & qdm;
        x ^= x >> qxs;
        x = (x * VS_Q16_B) & qdm;
        x ^= x >> qxs;
        return x;
    };
    auto q_inv = [&](uint32_t x) -> uint32_t {
        x ^= x >> qxs;
        x = (x * VS_Q16_BI) & qdm;
        x ^= x >> qxs;
        return (x * VS_Q16_AI) & qdm;
    };
    auto node_load = [&](uint32_t handle) -> uint32_t {
        if (VG == 3) {  // (table-less regime only: lhv == 0)
            if (handle < s.gcap) return gload16(reinterpret_cast<const uint16_t*>(ghash) + handle);
            return gload32(ghash + (s.gcap >> 1) + (handle - s.gcap));
        }
        if (handle < lhv) return lload32(lhash + handle);
        return gload32(ghash + (handle - lhv));
    };
    // what node_load returned (made uniform by the caller) -> the node id
    auto node_of = [&](uint32_t handle, uint32_t raw) -> uint32_t {
        if (VG == 3 && handle < s.gcap) return rfl(q_inv(((handle >> 3) << qk_v) | raw));  // (vector instructions: the masks live in vector registers)
        return raw;
    };
    // true where the id was not present before; slot_out = its handle
    auto finish_insert = [&](uint32_t nid, bool act, uint32_t slot, uint32_t old, uint32_t& slot_out) -> bool {
        bool fresh = false;
        if (act) {
            for (;;) {
                if (old == VS_EMPTY) { fresh = true; break; }
                if (old == nid) break;
                slot = slot + 1 == lhv ? 0 : slot + 1;
                old = atomicCAS(&lhash[slot], VS_EMPTY, nid);  // ds_cmpst_rtn_b32
            }
            slot_out = slot;
        }
        nins += (uint32_t)__popcll(__ballot(fresh));
        return fresh;
    };
    // ---- the per-scan global table (handles lh .. lh + gcap - 1).  Buckets of four slots (one aligned 16-byte load); the
    // probe sequence of an id starts at its home bucket and moves to the next bucket only past a FULL one, so "the id is not
    // in this bucket and the bucket has an empty slot" means the id is absent.  Lanes that want a slot of their bucket in
    // the same step draw distinct ranks from an LDS counter (lanes of other buckets sharing the counter only waste ranks)
    // and take the rank-th empty slot; a lane whose rank is past the bucket's empties looks at the bucket again.
    const bool gmode = lhv == 0;  // table-less regime: every id lives in the global table
    auto ghash_home = [&](uint32_t nid) -> uint32_t {
        return (uint32_t)(((uint64_t)hash_u32(nid ^ 0x5bd1e995u) * gbuckets) >> 32) << 2;
    };
    auto bucket_load = [&](uint32_t b0) -> uint4 { return *reinterpret_cast<const uint4*>(ghash + b0); };
    // VG: virgin = the scan has not written this bucket yet; nothing is requested for it
    auto bucket_fetch = [&](uint32_t b0, bool& virgin) -> uint4 {
        if (VG) {
            const uint32_t b = b0 >> 2;
            virgin = ((vmap[b >> 5] >> (b & 31u)) & 1u) == 0;
            if (virgin) return make_uint4(VS_EMPTY, VS_EMPTY, VS_EMPTY, VS_EMPTY);
        }
        return bucket_load(b0);
    };
    // v = the bucket at b0 as fetched by the caller (where act).  true where the id was not present before
    auto global_insert = [&](uint32_t nid, bool act, uint32_t b0, uint4 v, bool virgin, uint32_t& slot_out) -> bool {
        bool fresh = false, pend = act;
        for (;;) {
            wave_sync();  // every lane holds its snapshot before any lane stores (the loads are earlier instructions)
            uint32_t em = 0;
            if (VG && pend && virgin) {
                // the id cannot be in a bucket nobody has written.  The one lane that flips the bucket's bit writes all four
                // slots (its id and three empties) with one store; a lane that lost the bit to another lane of this step looks
                // at the bucket again, now a written one
                const uint32_t b = b0 >> 2, bit = 1u << (b & 31u);
                if ((atomicOr(&vmap[b >> 5], bit) & bit) == 0) {  // ds_or_rtn_b32
                    *reinterpret_cast<uint4*>(ghash + b0) = make_uint4(nid, VS_EMPTY, VS_EMPTY, VS_EMPTY);
                    slot_out = lhv + b0;
                    fresh = true;
                    pend = false;
                }
                virgin = false;
            } else if (pend) {
                const uint32_t key = nid;
                const uint32_t hit = (v.x == key ? 1u : 0u) | (v.y == key ? 2u : 0u) | (v.z == key ? 4u : 0u) | (v.w == key ? 8u : 0u);
                if (hit) {
                    slot_out = lhv + b0 + (uint32_t)__builtin_ctz(hit);
                    pend = false;
                } else {
                    em = (g_empty(v.x) ? 1u : 0u) | (g_empty(v.y) ? 2u : 0u) | (g_empty(v.z) ? 4u : 0u) | (g_empty(v.w) ? 8u : 0u);
                    if (em == 0) b0 = b0 + 4u == s.gcap ? 0u : b0 + 4u;
                }
            }
            const bool want = pend && em != 0;
            uint32_t* ctr = arb + ((b0 >> 2) & (ARB_SLOTS - 1u));
            uint32_t rank = 0;
            if (want) rank = atomicAdd(ctr, 1u);  // ds_add_rtn_u32
            wave_sync();
            if (want) {
                *ctr = 0;
                if (rank < (uint32_t)__popc(em)) {
                    uint32_t m = em;
                    if (rank > 0) m &= m - 1u;
                    if (rank > 1) m &= m - 1u;
                    if (rank > 2) m &= m - 1u;
                    const uint32_t at = b0 + (uint32_t)__builtin_ctz(m);
                    ghash[at] = nid;
                    slot_out = lhv + at;
                    fresh = true;
                    pend = false;
                }
            }
            wave_sync();
            if (!__ballot(pend)) break;
            if (pend) v = bucket_fetch(b0, virgin);  // (this wave's stores of the round above are visible: same CU, program order)
        }
        nins_g += (uint32_t)__popcll(__ballot(fresh));
        return fresh;
    };
    // ---- VG == 2: the slot bitmap.  slot_run(pos) = occupied slots in a row from pos to the end of pos's 4-slot group (0: pos is free)
    auto slot_home = [&](uint32_t nid) -> uint32_t { return (uint32_t)(((uint64_t)hash_u32(nid ^ 0x5bd1e995u) * s.gcap) >> 32); };
    auto slot_run = [&](uint32_t pos) -> uint32_t {
        const uint32_t g = pos & ~3u;
        const uint32_t rel = ((vmap[g >> 5] >> (g & 31u)) & 0xFu) >> (pos & 3u);  // bit 0 = slot pos; the bit past the group's end is clear
        return (uint32_t)__builtin_ctz(~rel);
    };
    // v = the 4-slot group of pos as fetched by the caller where pre_t != 0 (pre_t = the occupied run at pos when it looked; nothing
    // has been inserted since).  An id is only ever compared with a run as it stood when the group was loaded: a slot that was free
    // then holds an earlier scan's leftovers.  true where the id was not present before
    auto slot_insert = [&](uint32_t nid, bool act, uint32_t pos, uint4 v, uint32_t pre_t, uint32_t& slot_out) -> bool {
        bool fresh = false, pend = act;
        for (;;) {
            wave_sync();  // every lane sees the bits claimed and the ids stored in the round before
            uint32_t t = pre_t;
            if (pend && !pre_t) {
                t = slot_run(pos);
                if (t) v = *reinterpret_cast<const uint4*>(ghash + (pos & ~3u));  // (a bit that is set: its id was stored before the bit could be seen)
            }
            pre_t = 0;
            if (pend && t == 0) {
                // a free slot ends the probe sequence: the id is new.  The lane that flips the slot's bit owns it; a lane that lost the
                // bit to another lane of this round (another id: a neighbor list holds no id twice) moves on
                const uint32_t bit = 1u << (pos & 31u);
                if ((atomicOr(&vmap[pos >> 5], bit) & bit) == 0) {  // ds_or_rtn_b32
                    ghash[pos] = nid;
                    slot_out = lhv + pos;
                    fresh = true;
                    pend = false;
                } else {
                    pos = pos + 1 == s.gcap ? 0u : pos + 1;
                }
            } else if (pend) {
                const uint32_t hit = ((v.x == nid ? 1u : 0u) | (v.y == nid ? 2u : 0u) | (v.z == nid ? 4u : 0u) | (v.w == nid ? 8u : 0u)) &
                                     (((1u << t) - 1u) << (pos & 3u));
                if (hit) {
                    slot_out = lhv + (pos & ~3u) + (uint32_t)__builtin_ctz(hit);
                    pend = false;
                } else {
                    pos += t;
                    if (pos == s.gcap) pos = 0;
                }
            }
            wave_sync();
            if (!__ballot(pend)) break;
        }
        nins_g += (uint32_t)__popcll(__ballot(fresh));
        return fresh;
    };
    // ---- VG == 3: buckets of eight 16-bit entries (one aligned 16-byte load), one occupancy bit per slot in LDS.  An id lives in its
    // bucket at the first slot that was free, in cyclic order from its preferred slot, when it was inserted; slots never become free
    // again, so (a) an id whose preferred slot is free is new — no load —, (b) otherwise it is in the occupied run that starts at its
    // preferred slot or it is new, and ONE load of the bucket decides, whatever other lanes insert meanwhile (a neighbor list holds no
    // id twice, so nothing inserted after the snapshot can be this id); (c) an id whose bucket is full goes to a small overflow table
    // of whole ids behind the buckets (linear probing, its own occupancy bits), which is also where a lookup continues when all
    // eight entries of the bucket are other ids.  Half the bytes per slot of the 4-byte table and never a second dependent load.
    auto b16_occ = [&](uint32_t hb) -> uint32_t { return (vmap[hb >> 2] >> ((hb & 3u) << 3)) & 0xFFu; };
    auto b16_run = [&](uint32_t x) -> uint32_t {  // occupied slots in a row, cyclic, from x's preferred slot (0: it is free, 8: full)
        const uint32_t occ = b16_occ(x >> qk_v);
        const uint32_t rot = ((occ | (occ << 8)) >> (x & 7u)) & 0xFFu;
        return (uint32_t)__builtin_ctz(~rot);
    };
    auto b16_load = [&](uint32_t x) -> uint4 { return *reinterpret_cast<const uint4*>(ghash + ((size_t)(x >> qk_v) << 2)); };
    uint32_t n_ovf = 0;
    auto ovf_insert = [&](uint32_t nid, bool act, uint32_t& slot_out) -> bool {  // (as slot_insert, on the overflow table)
        uint32_t* const ot = ghash + (s.gcap >> 1);
        uint32_t* const om = vmap + (s.gcap >> 5);
        auto orun = [&](uint32_t pos) -> uint32_t {
            const uint32_t g = pos & ~3u;
            const uint32_t rel = ((om[g >> 5] >> (g & 31u)) & 0xFu) >> (pos & 3u);
            return (uint32_t)__builtin_ctz(~rel);
        };
        bool fresh = false, pend = act;
        uint32_t pos = (uint32_t)(((uint64_t)hash_u32(nid ^ 0x5bd1e995u) * s.ocap) >> 32);
        for (;;) {
            wave_sync();
            uint32_t t = 0;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pend) {
                t = orun(pos);
                if (t) v = *reinterpret_cast<const uint4*>(ot + (pos & ~3u));
            }
            if (pend && t == 0) {
                const uint32_t bit = 1u << (pos & 31u);
                if ((atomicOr(&om[pos >> 5], bit) & bit) == 0) {
                    ot[pos] = nid;
                    slot_out = lhv + s.gcap + pos;
                    fresh = true;
                    pend = false;
                } else {
                    pos = pos + 1 == s.ocap ? 0u : pos + 1;
                }
            } else if (pend) {
                const uint32_t hit = ((v.x == nid ? 1u : 0u) | (v.y == nid ? 2u : 0u) | (v.z == nid ? 4u : 0u) | (v.w == nid ? 8u : 0u)) &
                                     (((1u << t) - 1u) << (pos & 3u));
                if (hit) {
                    slot_out = lhv + s.gcap + (pos & ~3u) + (uint32_t)__builtin_ctz(hit);
                    pend = false;
                } else {
                    pos += t;
                    if (pos == s.ocap) pos = 0;
                }
            }
            wave_sync();
            if (!__ballot(pend)) break;
        }
        n_ovf += (uint32_t)__popcll(__ballot(fresh));
        return fresh;
    };
    // v / pre_t: the bucket of x and the occupied run at its preferred slot as the caller found them (have_pre; nothing has been inserted
    // since), else both are looked up here.  true where the id was not present before
    auto b16_insert = [&](uint32_t nid, bool act, uint32_t x, uint4 v, uint32_t pre_t, bool have_pre, uint32_t& slot_out) -> bool {
        uint32_t t = pre_t;
        if (!have_pre) {
            wave_sync();
            t = act ? b16_run(x) : 0u;
            v = make_uint4(0, 0, 0, 0);
            if (t) v = b16_load(x);
        }
        const uint32_t hb = x >> qk_v, r = x & rmask_v, pref = x & 7u;
        // In the run of the snapshot, or new.  Straight-line for every lane (round 6; t == 0 is an empty run mask, a lane without a load
        // compares zeros): which of the eight entries equal the remainder — per 16-bit half min(entry ^ r, 1) is 0 exactly where they do
        // (v_pk_min_u16; the constant is kept opaque, or the compiler turns the minimum into eight compares and selects), the halves'
        // bits are shifted into entry order (entry 2 i: bit 2 i, entry 2 i + 1: bit 16 + 2 i, folded down by 15)
        const uint32_t rr = r | (r << 16);
        uint32_t one = 0x00010001u;
        asm volatile("" : "+v"(one));
        const uint32_t z = pk_min_u16(v.x ^ rr, one) | (pk_min_u16(v.y ^ rr, one) << 2) | (pk_min_u16(v.z ^ rr, one) << 4) | (pk_min_u16(v.w ^ rr, one) << 6);
        const uint32_t m = ~(z | (z >> 15)) & 0xFFu;
        uint32_t rm = ((1u << t) - 1u) << pref;
        rm = (rm | (rm >> 8)) & 0xFFu;
        const uint32_t hit = act ? (m & rm) : 0u;
        if (hit) slot_out = lhv + (hb << 3) + (uint32_t)__builtin_ctz(hit);
        // A new id takes the first free slot of its bucket, cyclic from its preferred one.  Which lanes are still looking, which found a
        // slot and which met a full bucket are wave-uniform lane masks kept by hand: as per-lane booleans carried through the loop each
        // cost three scalar instructions per merge point
        uint64_t pendm = __ballot(act && hit == 0), freshm = 0, ovfm = 0;
        // (the first attempt needs no second look at the occupancy bits: the occupied run found with the snapshot ends at the first free
        // slot, and nothing has been inserted since; only a lane that lost its slot to another lane of the wave reads them again — round 6)
        uint32_t t2 = t;
        while (pendm) {
            wave_sync();
            const uint32_t idx = (hb << 3) + ((pref + t2) & 7u);
            const uint32_t bit = 1u << (idx & 31u);
            const bool mine = lane_of(pendm);
            uint32_t old = 0xFFFFFFFFu;
            if (mine && t2 < 8u) old = atomicOr(&vmap[idx >> 5], bit);  // ds_or_rtn_b32: the lane that flips the bit owns the slot
            const bool won = (old & bit) == 0;
            if (won) {
                gstore16(reinterpret_cast<uint16_t*>(ghash) + idx, r);
                slot_out = lhv + idx;
            }
            const uint64_t wonm = __ballot(won), fullm = __ballot(mine && t2 >= 8u);  // full: the overflow table (where the id may also be already)
            freshm |= wonm;
            ovfm |= fullm;
            pendm &= ~(wonm | fullm);
            wave_sync();
            if (!pendm) break;
            const uint32_t occ = b16_occ(hb);
            t2 = (uint32_t)__builtin_ctz(~(((occ | (occ << 8)) >> pref) & 0xFFu));  // (8: the bucket is full)
        }
        bool fresh = lane_of(freshm);
        const uint32_t ov0 = n_ovf;
        if (ovfm) fresh = ovf_insert(nid, lane_of(ovfm), slot_out) || fresh;
        nins_g += (uint32_t)__popcll(freshm) + (n_ovf - ov0);
        return fresh;
    };
    auto open_table = [&]() -> bool {  // first use: this wave claims and clears its own table
        if (g_open) return true;
        if (!claim_region()) return false;
        g_open = true;
        if (!VG) {
            for (uint32_t i = 4u * lane; i < s.gcap; i += 4u * WAVE)
                *reinterpret_cast<uint4*>(ghash + i) = make_uint4(VS_EMPTY, VS_EMPTY, VS_EMPTY, VS_EMPTY);
            wave_sync();
        }
        return true;
    };
    // Frozen mode (LDS table at its load limit): read-only probe of the LDS table, then the global table
    auto frozen_insert = [&](uint32_t nid, bool act, uint32_t& slot_out) -> bool {
        bool need_g = act;
        if (act && lhv) {
            uint32_t slot = hash_home(nid);
            for (;;) {
                const uint32_t v = lhash[slot];
                if (v == nid) { need_g = false; break; }
                if (v == VS_EMPTY) break;
                slot = slot + 1 == lhv ? 0 : slot + 1;
            }
        }
        if (!__ballot(need_g)) return false;
        if (!open_table()) return false;
        if (nins_g > s.glimit) {
            status |= OVF_HASH;
            return false;
        }
        if (VG == 3) return b16_insert(nid, need_g, q_fwd(nid), make_uint4(0, 0, 0, 0), 0u, false, slot_out);
        if (VG == 2) return slot_insert(nid, need_g, slot_home(nid), make_uint4(0, 0, 0, 0), 0u, slot_out);
        const uint32_t b0 = ghash_home(nid);
        uint4 v = make_uint4(0, 0, 0, 0);
        bool virgin = false;
        if (need_g) v = bucket_fetch(b0, virgin);
        return global_insert(nid, need_g, b0, v, virgin, slot_out);
    };

    if (status) { /* handed over (wide label key): no region is claimed */ }
    else if (gmode) open_table();  // claimed and cleared up front
    else if (onlyfv) claim_region();  // (the heap spill array of a second attempt comes with the region)
    // (resumable scans) the scalars of a saved scan; `ended`: next() has returned None before (the stream is over for good)
    uint32_t next_base = 0, invis_base = 0;
    bool ended = false;
    if (resumed) {
        heap.len = rfl(rs[RSF_HLEN]);
        vis.head = rfl(rs[RSF_VHEAD]);
        vis.len = rfl(rs[RSF_VLEN]);
        nins_g = rfl(rs[RSF_NINS_G]);
        n_ovf = rfl(rs[RSF_N_OVF]);
        hmax = rfl(rs[RSF_HMAX]);
        st_visits = rfl(rs[RSF_VISITS]);
        st_cand = rfl(rs[RSF_CAND]);
        st_pops = rfl(rs[RSF_POPS]);
        invis_base = rfl(rs[RSF_INVIS]);
        status = rfl(rs[RSF_STATUS]);
        next_base = rfl(rs[RSF_NEXT]);
        ended = rfl(rs[RSF_ENDED]) != 0;
    }

    // ---- ListSearchResult::new: start nodes (AM/graph/mod.rs:97-124, AM/graph/start_nodes.rs:39-48) ----
    {
        uint32_t nstarts = labels_some ? nql : 1u;
        if (a.default_start == VS_INVALID_NODE || a.n == 0 || status || resumed) nstarts = 0;  // ListSearchResult::empty() (or a handed-over scan)
        for (uint32_t si = 0; si < nstarts; ++si) {
            uint32_t sn = VS_INVALID_NODE;
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
            sn = rfl(sn);
            if (sn == VS_INVALID_NODE) continue;
            // create_lsn_for_start_node (AM/sbq/storage.rs:365-391)
            uint32_t slot = hash_home(sn), old = VS_EMPTY;
            bool fr;
            if (nins + WAVE > slot_limit) {
                fr = frozen_insert(sn, lane == 0, slot);
                if (status) break;
            } else {
                if (lane == 0) old = atomicCAS(&lhash[slot], VS_EMPTY, sn);
                fr = finish_insert(sn, lane == 0, slot, old, slot);
            }
            if (!rfl(fr ? 1u : 0u)) continue;
            slot = rfl(slot);
            const uint32_t d =
                rfl(ham_row_reg<NCH, QL, XW>(codes_v + (size_t)sn * cstride, qv, qc_l, l4, cstride, lane < 4, stream_rows));
            st_cand++;
            if (heap.len + 1 > s.hcap) { status |= OVF_HEAP; break; }
            heap.push((d << s.sb) | slot);
        }
    }

    // Neighbor rows are requested ahead of the pop that needs them: slot A = the heap root left by the last pop (asked
    // for together with the code gather, so both latencies overlap), slot B = the best new candidate when it beats that
    // root (asked for as soon as its distance is known; the pushes / pop / visited insert cover the latency).
    uint32_t pfa_node = VS_INVALID_NODE, pfa_val = VS_INVALID_NODE, pfa_h = 0xFFFFFFFFu;  // (_h: the node's dedup handle)
    uint32_t pfb_node = VS_INVALID_NODE, pfb_val = VS_INVALID_NODE, pfb_h = 0xFFFFFFFFu;
    // label-filtered scans on an index with label masks: the masks of a node's neighbors sit next to its neighbor row (one
    // coalesced 8 x R byte load with the row) instead of one random 8-byte load per fresh neighbor
    const uint64_t* const nbr_mask = FULL ? a.nbr_mask : nullptr;
    uint64_t pfa_m = 0, pfb_m = 0;

    // ---- TSVResponseIterator::next until M rows are emitted (AM/scan.rs:210-242), flattened: every iteration is
    // either one visit_closest() expansion (greedy_search_iterate, AM/graph/mod.rs:357-385) or one consume() ----
    uint32_t ft_node = VS_INVALID_NODE;  // heap tid (and visibility) of the visited list's front entry, requested ahead of consume()
    uint64_t ft_val = 0;
    uint32_t ft_vis = 1, st_invis = invis_base;
    uint32_t hslot0 = 0;
    uint4 gbk0 = make_uint4(0, 0, 0, 0);
    bool rchit0 = false;  // this lane's id of the first chunk was found in the id cache
    bool virg0 = false;
    uint32_t pret0 = 0;  // (VG == 2) occupied run at the home slot of this lane's id of the first chunk when its group was requested
    while (status == 0 && !(RS && ended)) {
        // What a lane derives from its index (lane < c, (lane & 15) == 0, lane == 0 ... as exec masks in scalar register pairs) is loop
        // invariant, so the compiler computes some fifty of them ahead of the loop — and then has to park them in vector-register
        // lanes, two v_readlane each to get one back (round 6: 226 scalar spills).  An opaque lane index per iteration makes each a
        // one-instruction compare where it is used.
        asm volatile("" : "+v"(lane));
        heap.lane = lane;
        vis.lane = lane;
        l4 = lane & 3;
        hslot0 = 0;  // (these do not live across iterations)
        gbk0 = make_uint4(0, 0, 0, 0);
        rchit0 = false;
        virg0 = false;
        pret0 = 0;
        if (VR > 0 && vis.len > 0) {  // (the ring carries what consume() needs in its entries)
            const uint32_t fn = readlane_u32(vis.n[0], 0);
            if (fn != ft_node) {
                ft_node = fn;
                ft_val = load_stream64(tids_v + fn);
                if (visible) ft_vis = visible[fn];
            }
        }
        uint32_t top = 0;
        bool can_visit = heap.len > 0;
        if (can_visit) {
            top = heap.root();
            if (vis.len > s.L)  // visit_closest(L) stop rule (AM/graph/mod.rs:153-170)
                can_visit = (top >> s.sb) < vis.ham_at(s.L - 1);
        }
        if (!can_visit) {
            if (BUILD) break;  // greedy_search_for_build stops here: the visited list is the candidate set
            // ---- consume (AM/graph/mod.rs:174-184) + return_lsn (AM/sbq/storage.rs:404-414) ----
            if (vis.len == 0) {  // None: the stream has ended
                ended = true;
                break;
            }
            uint32_t fd, fnode, fflags;
            vis.pop_front(fd, fnode, fflags);
            st_pops++;
            if (VR > 0) {
                const uint64_t tid = fnode == ft_node ? ft_val : load_stream64(tids_v + fnode);
                fflags = (tid & 0xFFFFull) == 0 ? VIS_DEAD : 0u;
                if (visible) fflags |= rfl(fnode == ft_node ? ft_vis : (uint32_t)visible[fnode]) == 0 ? VIS_HIDDEN : 0u;
            }
            if (fflags & VIS_DEAD) continue;  // InvalidOffsetNumber: deleted tuple (AM/scan.rs:231-234)
            if (visible && (fflags & VIS_HIDDEN)) {  // get_full_distance_for_resort -> None: fetched, counted, never enters
                st_invis++;                              // the window (AM/scan.rs:268-272)
                continue;
            }
            if (lane == 0) {
                s.out_ids[(size_t)q * s.M + emitted] = fnode;
                s.out_ham[(size_t)q * s.M + emitted] = fd;
                if (RS && s.row_stats) {  // the counters as the reference's stood when this row left next() (vs_search.hip has the same)
                    uint32_t* r = s.row_stats + ((size_t)q * s.M + emitted) * ST_N;
                    r[ST_VISITS] = st_visits;
                    r[ST_CAND] = st_cand;
                    r[ST_DQ] = st_cand;
                    r[ST_READS] = st_pops + st_visits + nins + nins_g;
                    r[ST_NEXT] = next_base + emitted + 1u + (st_invis - invis_base);
                    r[ST_GSPILL] = hmax;
                    r[ST_INVIS] = st_invis;
                    r[7] = nins + nins_g;
                }
            }
            emitted++;
            lap(6);
            if (emitted == s.M) break;
            continue;
        }
        hmax = max(hmax, heap.len);
        lap(6);
        const uint32_t hd = top >> s.sb;
        // The node about to be visited is, almost always, one of the two whose neighbor rows were requested during the last
        // visit; its dedup handle says so without a memory access.  Then everything the visit needs from memory before it can
        // touch the heap again — heap tid, visibility, and (table-less regime) the dedup buckets of the row's ids — is requested
        // BEFORE the pop and arrives while the pop works.
        const uint32_t th = top & smask;
        const bool hit_a = th == pfa_h, hit_b = th == pfb_h;
        const bool hit = hit_a || hit_b;
        uint32_t node_v = hit_a ? pfa_node : pfb_node;
        if (!hit) node_v = node_load(th);
        uint32_t row0 = hit_a ? pfa_val : pfb_val;
        uint64_t rowm = hit_a ? pfa_m : pfb_m;
        uint64_t vtid = 1;
        uint32_t vvis = 1;
        bool early = false;
        if (hit) {
            if (VR == 0 && !BUILD) {
                vtid = load_stream64(tids_v + node_v);
                if (visible) vvis = visible[node_v];
            }
            if (gmode && nins_g <= s.glimit) {
                early = true;
                const uint64_t inval0 = __ballot(row0 == VS_INVALID_NODE);
                const bool act0 = (uint32_t)lane < (inval0 ? (uint32_t)__builtin_ctzll(inval0) : WAVE);
                virg0 = false;
                gbk0 = make_uint4(0, 0, 0, 0);
                if (VG == 3) {
                    hslot0 = q_fwd(row0);
                    pret0 = act0 ? b16_run(hslot0) : 0u;
                    if (pret0) gbk0 = b16_load(hslot0);
                } else if (VG == 2) {
                    hslot0 = slot_home(row0);
                    pret0 = act0 ? slot_run(hslot0) : 0u;
                    if (pret0) gbk0 = *reinterpret_cast<const uint4*>(ghash + (hslot0 & ~3u));
                } else {
                    hslot0 = ghash_home(row0);
                    rchit0 = rcv ? rc[hash_u32(row0 ^ 0x9e3779b9u) & rcm] == row0 : false;
                    if (act0 && !rchit0) gbk0 = bucket_fetch(hslot0, virg0);
                }
            }
        }
        heap.pop();
        const uint32_t node = hit ? rfl(node_v) : node_of(th, rfl(node_v));
        // what consume() will need to know about this node: requested now, folded into the ring entry at the insert below
        if (!hit && VR == 0 && !BUILD) {
            vtid = load_stream64(tids_v + node);
            if (visible) vvis = visible[node];
        }
        const uint32_t* nrow = nbrs_v + (size_t)node * nstride;
        if (!hit) {
            row0 = ((uint32_t)lane < R_v) ? load_stream32(nrow + lane) : VS_INVALID_NODE;
            if (nbr_mask) rowm = ((uint32_t)lane < R_v) ? load_stream64(nbr_mask + (size_t)node * nstride + lane) : 0ull;
        }
        lap(0);
        if (vis.len + 1 > vis.capacity()) {
            if (BUILD) vis.len = vis.capacity() - 1;  // build mode keeps the closest entries as prune candidates
            else { status |= OVF_VISITED; break; }
        }
        st_visits++;  // (also one SbqNode::read(visiting))
        // ---- visit_lsn_internal, Disk arm (AM/sbq/storage.rs:135-190) ----
        uint32_t root_after = 0xFFFFFFFFu;
        uint32_t root_node_v = VS_INVALID_NODE;  // id of the new root (slot A of the row prefetch), resolved at gather time
        auto after_pop = [&]() {
            root_after = heap.len > 0 ? heap.root() : 0xFFFFFFFFu;
            if (root_after != 0xFFFFFFFFu) root_node_v = node_load(root_after & smask);
        };
        after_pop();
        uint32_t best = 0xFFFFFFFFu;  // smallest (hamming << sb | slot) among this visit's new candidates
        uint32_t best_node = VS_INVALID_NODE;
        bool pfa_issued = false, pfb_issued = false, vis_done = false;
        bool list_ended = false;
        for (uint32_t c0 = 0; c0 < (R1 ? 1u : a.R) && !list_ended; c0 += WAVE) {
            const uint32_t slotidx = c0 + lane;
            const uint32_t nid = c0 == 0 ? row0 : ((slotidx < R_v) ? load_stream32(nrow + slotidx) : VS_INVALID_NODE);
            // list ends at the first InvalidBlockNumber (AM/sbq/node.rs:260-285)
            const uint64_t inval = __ballot(nid == VS_INVALID_NODE);
            const uint32_t nvalid = inval ? (uint32_t)__builtin_ctzll(inval) : WAVE;
            if (nvalid < WAVE) list_ended = true;
            const bool act = (uint32_t)lane < nvalid;
            lap(1);
            // prepare_insert (marks BEFORE the label check, AM/sbq/storage.rs:148-172): first probe issued, ...
            const bool frozen = !gmode && nins + WAVE > slot_limit;
            uint32_t hslot = gmode ? (VG == 3 ? q_fwd(nid) : VG == 2 ? slot_home(nid) : ghash_home(nid)) : hash_home(nid), old = VS_EMPTY;
            uint4 gbk = make_uint4(0, 0, 0, 0);
            bool rchit = false, virg = false;
            uint32_t pret = 0;  // (VG == 2) != 0: gbk holds the group of hslot, whose occupied run was pret slots long
            if (gmode && early && c0 == 0) {
                hslot = hslot0;  // requested before the pop
                gbk = gbk0;
                rchit = rchit0;
                virg = virg0;
                pret = pret0;
            } else if (gmode && VG == 3) {
                if (nins_g > s.glimit) { status |= OVF_HASH; break; }
                pret = act ? b16_run(hslot) : 0u;
                if (pret) gbk = b16_load(hslot);  // in flight during the visited insert
            } else if (gmode && VG == 2) {
                if (nins_g > s.glimit) { status |= OVF_HASH; break; }
                pret = act ? slot_run(hslot) : 0u;
                if (pret) gbk = *reinterpret_cast<const uint4*>(ghash + (hslot & ~3u));  // in flight during the visited insert
            } else if (gmode) {
                if (nins_g > s.glimit) { status |= OVF_HASH; break; }
                if (rcv && act) rchit = rc[hash_u32(nid ^ 0x9e3779b9u) & rcm] == nid;
                if (act && !rchit) gbk = bucket_fetch(hslot, virg);  // in flight during the visited insert
            } else if (!frozen && act) {
                old = atomicCAS(&lhash[hslot], VS_EMPTY, nid);
            }
            // ... visited.insert(partition_point(|x| *x < head), head) runs in registers meanwhile ...
            if (!vis_done) {
                vis_done = true;
                vis.insert(hd, node, ((vtid & 0xFFFFull) == 0 ? VIS_DEAD : 0u) | (vvis == 0 ? VIS_HIDDEN : 0u));
                lap(2);
            }
            // ... then the probe sequence is finished
            bool fresh;
            if (gmode && VG == 3) {
                fresh = b16_insert(nid, act, hslot, gbk, pret, true, hslot);
                if (__ballot(n_ovf + WAVE > ovf_lim_v)) status |= OVF_HASH;  // (the overflow table at its load limit: the second attempt)
            } else if (gmode && VG == 2) {
                fresh = slot_insert(nid, act, hslot, gbk, pret, hslot);
            } else if (gmode) {
                fresh = global_insert(nid, act && !rchit, hslot, gbk, virg, hslot);
                if (rcv && act && !rchit) rc[hash_u32(nid ^ 0x9e3779b9u) & rcm] = nid;  // (now in the table, new or not)
            } else if (frozen) {
                fresh = frozen_insert(nid, act, hslot);
                if (status) break;
            } else {
                fresh = finish_insert(nid, act, hslot, old, hslot);
            }
            // (SbqNode::read(neighbor) for every fresh id: counted by the insert as nins / nins_g)
            // label filter: query.labels.overlaps(node.labels) (AM/labels/mod.rs:124-142)
            bool pass = fresh;
            if (has_label_filter && nbr_mask) {
                const uint64_t nm = c0 == 0 ? rowm : ((slotidx < R_v) ? load_stream64(nbr_mask + (size_t)node * nstride + slotidx) : 0ull);
                pass = fresh && (nm & qmask) != 0;
            } else if (has_label_filter && a.label_mask) {
                if (fresh) pass = (a.label_mask[nid] & qmask) != 0;
            } else if (has_label_filter && fresh) {
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
            const uint64_t pm = __ballot(pass);
            const uint32_t c = (uint32_t)__popcll(pm);
            lap(3);
            if (c) {
                if (heap.len + c > s.hcap) { status |= OVF_HEAP; break; }
                // compact survivors in neighbor-list order
                if (pass) {
                    const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                    surv_id[rank] = nid;
                    surv_slot[rank] = hslot;
                }
                wave_sync();
            }
            if (c == 0) continue;
            // the heap positions the new candidates will get are already known: their ancestors' load (L2 when the heap
            // spills) is issued now and overlaps the code gather
            const uint32_t pre_n = heap.first_run(c);
            const uint32_t pre_anc = pre_n ? heap.wide_load(pre_n) : 0u;
            // distances: 4 lanes per code row, 16 rows per pass; the row of the current heap root rides along
            // (G2, the 5-waves-per-SIMD variant: two rows per 4-lane group are in flight at once, so the usual 17..32 new candidates of
            // a visit cost one round trip instead of two — paid for with 12 registers, i.e. 20 instead of 24 scans per CU)
            const uint32_t npass = (c + 15u) >> 4;
            for (uint32_t pass_i = 0; pass_i < npass; pass_i += G2 ? 2u : 1u) {
                const uint32_t j = pass_i * 16u + (uint32_t)(lane >> 2);
                const bool valid = j < c;
                const uint32_t id = valid ? surv_id[j] : 0;
                const uint64_t* crow = codes_v + (size_t)id * cstride;
                if (pass_i == 0 && !pfa_issued) {
                    pfa_issued = true;
                    pfa_node = VS_INVALID_NODE;
                    pfa_h = 0xFFFFFFFFu;
                    if (root_after != 0xFFFFFFFFu) {
                        pfa_node = node_of(root_after & smask, rfl(root_node_v));
                        pfa_h = root_after & smask;
                        pfa_val = ((uint32_t)lane < R_v) ? load_stream32(nbrs_v + (size_t)pfa_node * nstride + lane) : VS_INVALID_NODE;
                        if (nbr_mask) pfa_m = ((uint32_t)lane < R_v) ? load_stream64(nbr_mask + (size_t)pfa_node * nstride + lane) : 0ull;
                    }
                }
                if (G2) {
                    const uint32_t j2 = j + 16u;
                    const bool valid2 = j2 < c;
                    const uint64_t* crow2 = codes_v + (size_t)(valid2 ? surv_id[j2] : 0u) * cstride;
                    uint32_t d, d2;
                    ham_row_reg2<NCH, QL>(crow, crow2, qv, qc_l, l4, cstride, valid, valid2, stream_rows, d, d2);
                    if (valid && l4 == 0) surv_d[j] = LEAN ? ((d << s.sb) | surv_slot[j]) : d;
                    if (valid2 && l4 == 0) surv_d[j2] = LEAN ? ((d2 << s.sb) | surv_slot[j2]) : d2;
                    continue;
                }
                const uint32_t d = ham_row_reg<NCH, QL, XW>(crow, qv, qc_l, l4, cstride, valid, stream_rows);
                if (valid && l4 == 0) surv_d[j] = LEAN ? ((d << s.sb) | surv_slot[j]) : d;
            }
            st_cand += c;
            wave_sync();
            uint32_t entry = 0xFFFFFFFFu;
            if ((uint32_t)lane < c) entry = LEAN ? surv_slot[lane] : ((surv_d[lane] << s.sb) | surv_slot[lane]);
            if (TIMING) {
                if (__ballot(entry == 0xFFFFFFFEu) == ~0ull) status |= 0x100;
                lap(4);
            }
            // a new candidate with a strictly smaller key than the root's is the next pop for sure
            {
                const uint32_t m = wave_min_u32(entry);
                if (m < best) {
                    best = m;
                    best_node = rfl(surv_id[__builtin_ctzll(__ballot(entry == m))]);  // entry lane = survivor rank
                }
            }
            if ((R1 || list_ended || c0 + WAVE >= a.R) && !pfb_issued) {
                pfb_issued = true;
                pfb_node = VS_INVALID_NODE;
                pfb_h = 0xFFFFFFFFu;
                if (best != 0xFFFFFFFFu && (best >> s.sb) < (root_after >> s.sb)) {
                    pfb_node = best_node;  // a candidate of this visit: its id is known without a table lookup
                    pfb_h = best & smask;
                    pfb_val = ((uint32_t)lane < R_v) ? load_stream32(nbrs_v + (size_t)pfb_node * nstride + lane) : VS_INVALID_NODE;
                    if (nbr_mask) pfb_m = ((uint32_t)lane < R_v) ? load_stream64(nbr_mask + (size_t)pfb_node * nstride + lane) : 0ull;
                }
            }
            // insert_neighbor in list order (AM/graph/mod.rs:144-147)
            heap.push_run(entry, c, pre_n, pre_anc);
            lap(5);
        }
        if (status) break;
        if (!vis_done) vis.insert(hd, node, ((vtid & 0xFFFFull) == 0 ? VIS_DEAD : 0u) | (vvis == 0 ? VIS_HIDDEN : 0u));  // (an empty neighbor list)
        if (!pfa_issued) {  // nothing new to score: the old root is the next expansion
            pfa_node = VS_INVALID_NODE;
            pfa_h = 0xFFFFFFFFu;
            if (root_after != 0xFFFFFFFFu) {
                pfa_node = node_of(root_after & smask, rfl(root_node_v));
                pfa_h = root_after & smask;
                pfa_val = ((uint32_t)lane < R_v) ? load_stream32(nbrs_v + (size_t)pfa_node * nstride + lane) : VS_INVALID_NODE;
                if (nbr_mask) pfa_m = ((uint32_t)lane < R_v) ? load_stream64(nbr_mask + (size_t)pfa_node * nstride + lane) : 0ull;
            }
        }
        if (!pfb_issued) {
            pfb_node = VS_INVALID_NODE;
            pfb_h = 0xFFFFFFFFu;
        }
    }
    if (BUILD && VR == 0 && status == 0) {  // the visited list itself is the output (sorted by (hamming, recency))
        emitted = min(vis.len, s.M);
        for (uint32_t i = lane; i < emitted; i += WAVE) {
            const uint64_t e = vis.ring[vis.slot(i)];
            s.out_ids[(size_t)q * s.M + i] = (uint32_t)e;
            s.out_ham[(size_t)q * s.M + i] = (uint32_t)(e >> 32) & VIS_HAM_MASK;
        }
    }
    // one `next` call per emitted row, plus the call that found the stream exhausted
    const uint32_t st_next = next_base + emitted + (st_invis - invis_base) + ((emitted < s.M && status == 0) ? 1u : 0u);
    if (RS) {  // save the scan for the next launch (a failed scan keeps its flag: the pool hands it to a cursor of its own)
        wave_sync();
        for (uint32_t i = 4u * lane; i < image_words; i += 4u * WAVE) *reinterpret_cast<uint4*>(rs + RSF_HDR + i) = *reinterpret_cast<const uint4*>(hp + i);
        if (lane == 0) {
            rs[RSF_INIT] = 1u;
            rs[RSF_HLEN] = heap.len;
            rs[RSF_VHEAD] = vis.head;
            rs[RSF_VLEN] = vis.len;
            rs[RSF_NINS_G] = nins_g;
            rs[RSF_N_OVF] = n_ovf;
            rs[RSF_HMAX] = hmax;
            rs[RSF_VISITS] = st_visits;
            rs[RSF_CAND] = st_cand;
            rs[RSF_POPS] = st_pops;
            rs[RSF_INVIS] = st_invis;
            rs[RSF_STATUS] = status;
            rs[RSF_NEXT] = st_next;
            rs[RSF_ENDED] = ended ? 1u : 0u;
        }
    }
    if (status == 0) {
        for (uint32_t i = emitted + lane; i < s.M; i += WAVE) {
            s.out_ids[(size_t)q * s.M + i] = VS_INVALID_NODE;
            s.out_ham[(size_t)q * s.M + i] = 0xFFFFFFFFu;
        }
    }
    if (lane == 0) {
        if (onlyfv && s.fb_flag) s.fb_flag[q] = 1;
        s.status[q] = status;
        s.out_cnt[q] = status ? 0 : emitted;  // a failed scan publishes an empty stream until the fallback re-runs it
        if (status == 0) {
            uint32_t* st = s.stats + (size_t)q * ST_N;
            st[ST_VISITS] = st_visits;
            st[ST_CAND] = st_cand;
            st[ST_DQ] = st_cand;
            st[ST_READS] = st_pops + st_visits + nins + nins_g;
            st[ST_NEXT] = st_next;
            st[ST_GSPILL] = hmax;
            st[ST_INVIS] = st_invis;
            st[7] = nins + nins_g;
        }
        if (TIMING && s.phase) {
            lap(6);
            for (int k = 0; k < 8; ++k) s.phase[(size_t)q * 8 + k] = ph[k];
        }
        if (s.timeline) s.timeline[2 * (size_t)q + 1] = wall_clock64();
    }
}

template <int NCH, int VR, bool TIMING, int MINW, bool BUILD, bool FULL = true, int VG = 0, int OPT = 0>
__global__ __launch_bounds__(WAVE, MINW) void k_search_fast(FastArgs a) {
    // persistent grid (a.s.persist): the launch has no tail of its own beyond the last scans' lives, and the hardware never has to place a
    // new workgroup (LDS, registers, a wave slot) while the chip is full; otherwise one workgroup per scan.  ONE inlined copy of the scan
    // for both (round 6): as two copies the register allocator spilled differently in each, and a change that was neutral for the
    // launches of one kind cost those of the other 35 % in scratch traffic (profiles/r06/s12, s13).
    for (uint32_t it = 0;; ++it) {
        uint32_t q = blockIdx.x;
        if (a.s.persist) {
            if (threadIdx.x == 0) q = atomicAdd(a.s.scan_counter, 1u);
            q = rfl(q);
        } else if (it) {
            return;
        }
        if (q >= a.s.nq) return;
        fast_scan<NCH, VR, TIMING, MINW, BUILD, FULL, VG, OPT>(a, q, blockIdx.x);
        wave_sync();  // the next scan re-initialises the LDS state: every lane is done with this one's
    }
}

#if VS_FAST_TU == 0
// u32 words of one saved scan at these capacities (header + the LDS image: heap top, visited ring, occupancy bits)
size_t fast_resume_words(const FastLaunch& s) { return RSF_HDR + ((size_t)(s.hl + 1) + 2 * (size_t)s.vcap + s.vwords + 3) / 4 * 4; }

size_t fast_lds_bytes(const vs_index* idx, const FastLaunch& s) {
    const size_t nch = (idx->code_stride + 7) / 8;
    // LDS copy of the query code: the generic variant (NCH == 0), and the register-capped variants (minw >= 6: 8 NCH words, zero padded)
    const size_t qcopy = nch > 6 ? (size_t)idx->code_stride * 8 : (s.minw >= 6 && !s.build && !s.phase ? nch * 64 : 0);
    const bool lean = s.minw == 7 && s.vslot == 2 && nch == 3 && !s.build && !s.phase;  // (fast_scan: LEAN)
    const bool plain = !s.qlabel_off && !s.visible && !(s.flags & FAST_FULL_VARIANT);
    size_t b = (size_t)(s.hl + 1) * 4 + (size_t)s.lh * 4 + (lean ? 2 : 3) * 64 * 4 + (s.vslot ? 0 : ARB_SLOTS * 4) + (s.vr ? 0 : (size_t)s.vcap * 8) +
               (lean && plain ? 0 : MAX_QLABELS * 2) + qcopy + (size_t)s.rc * 4 + (size_t)s.vwords * 4 + (lean ? 16 : 32);
    return (b + 15) / 16 * 16;
}

#endif  // VS_FAST_TU == 0

template <int NCH, int VR, bool TIMING, int MINW, bool BUILD, bool FULL = true, int VG = 0, int OPT = 0>
static int launch_fast_tt(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* resident) {
    static DeviceOnce attr_set;
    const int attr_dev = idx->ctx->device;
    if (attr_set.pending(attr_dev)) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_fast<NCH, VR, TIMING, MINW, BUILD, FULL, VG, OPT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done(attr_dev);
    }
    if (resident) {  // not a launch: how many scans (= single-wave workgroups) of this instantiation does the device hold at once?
        int per_cu = 0;
        VS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_search_fast<NCH, VR, TIMING, MINW, BUILD, FULL, VG, OPT>, WAVE, lds));
        *resident = (uint32_t)std::max(per_cu, 1) * (uint32_t)idx->ctx->prop.multiProcessorCount;
        return VS_OK;
    }
    const uint32_t grid = a.s.persist ? std::min<uint32_t>(a.s.persist, a.s.nq) : a.s.nq;
    hipLaunchKernelGGL((k_search_fast<NCH, VR, TIMING, MINW, BUILD, FULL, VG, OPT>), dim3(grid), dim3(WAVE), lds, idx->ctx->stream, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// the four instantiations that live in translation units of their own (16-bit tables, 24-word code rows, six waves per SIMD; std_geom:
// num_neighbors <= 64 at a row pitch of 64 ids, the compile-time geometry of OPT_XW | OPT_R1 | OPT_NS64)
int launch_fast_plain6(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* res, bool std_geom);  // no label keys, no visibility mask
int launch_fast_keys6(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* res, bool std_geom);   // with
#if VS_FAST_TU == 1
int launch_fast_plain6(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* res, bool std_geom) {
    if (std_geom) return launch_fast_tt<3, 0, false, 6, false, false, 3, OPT_XW | OPT_R1 | OPT_NS64>(idx, a, lds, res);
    return launch_fast_tt<3, 0, false, 6, false, false, 3>(idx, a, lds, res);
}
#elif VS_FAST_TU == 2
int launch_fast_keys6(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* res, bool std_geom) {
    if (std_geom) return launch_fast_tt<3, 0, false, 6, false, true, 3, OPT_XW | OPT_R1 | OPT_NS64>(idx, a, lds, res);
    return launch_fast_tt<3, 0, false, 6, false, true, 3>(idx, a, lds, res);
}
#else
template <int NCH>
static int launch_fast_t(vs_index* idx, const FastArgs& a, size_t lds, uint32_t* res) {
    if (a.s.build) {
        VS_REQUIRE(a.s.vr == 0 && !a.s.phase, "build-mode search uses the LDS-ring visited list");
        return launch_fast_tt<NCH, 0, false, 1, true>(idx, a, lds, res);
    }
    if (a.s.phase) {
        VS_REQUIRE(NCH == 3, "VS_PHASE diagnostics are built for 17..24-word codes only");
        if (a.s.vr == 8) return launch_fast_tt<3, 8, true, 1, false>(idx, a, lds, res);
        return launch_fast_tt<3, 0, true, 1, false>(idx, a, lds, res);
    }
    if (a.s.resume) {  // resumable scans (the scan pools): one workgroup per pool slot, registers unconstrained
        VS_REQUIRE(a.s.vwords && a.s.vslot == 2 && !a.s.persist && a.s.vr == 0 && a.s.lh == 0 && a.s.rc == 0 && !a.s.only_failed && a.s.pool_slots >= a.s.nq,
                   "fast search: resumable scans run the 16-bit tables with one region per slot");
        return launch_fast_tt<NCH, 0, false, 1, false, true, 3, OPT_RS>(idx, a, lds, res);
    }
    if (a.s.vwords && a.s.vslot == 2) {  // 16-bit entries in buckets of eight + overflow table, occupancy bits in LDS (see fast_scan)
        VS_REQUIRE(a.s.vr == 0 && a.s.lh == 0 && a.s.rc == 0 && a.s.gcap >= 256 && (a.s.gcap & (a.s.gcap - 1)) == 0 && a.s.ocap % 32 == 0 && a.s.ocap >= 64 &&
                       (uint64_t)a.s.vwords * 32 >= (uint64_t)a.s.gcap + a.s.ocap && a.s.qk >= 3 && a.s.qk <= 16 && a.s.qd <= 32 &&
                       (1ull << a.s.qd) == ((uint64_t)(a.s.gcap >> 3) << a.s.qk) && (a.s.qd >= 32 || a.n <= (1ull << a.s.qd)) &&
                       a.s.gregion >= (a.s.gcap >> 1) + a.s.ocap,
                   "fast search: bad geometry of the 16-bit dedup table");
        const bool plain = !a.s.qlabel_off && !a.s.visible && !(a.s.flags & FAST_FULL_VARIANT);
        // (the usual index: 24-word code rows — 768 x 2 bit, 1536 x 1 bit — and num_neighbors <= 64)
        const bool std_geom = a.code_stride == 24 && a.R <= WAVE && a.nbr_stride == 64;
        // (OPT_G2 at six waves per SIMD — two code rows per 4-lane group in flight, 78 VGPRs, no spills — was timed at 50M in round 6:
        // 134.6 ms against 130.3, profiles/r06/s4_ab_50m.txt.  More requests in flight per scan do not help a launch whose 6 144 scans
        // already keep the memory system busy; the instantiation is not built.)
        if (NCH == 3 && a.s.minw == 6 && plain) return launch_fast_plain6(idx, a, lds, res, std_geom);
        if (NCH == 3 && a.s.minw == 6) return launch_fast_keys6(idx, a, lds, res, std_geom);
        if (NCH == 3 && a.s.minw == 7 && plain && std_geom) return launch_fast_tt<3, 0, false, 7, false, false, 3, OPT_XW | OPT_R1 | OPT_NS64>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 7 && plain) return launch_fast_tt<3, 0, false, 7, false, false, 3>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 7) return launch_fast_tt<3, 0, false, 7, false, true, 3>(idx, a, lds, res);
        return launch_fast_tt<NCH, 0, false, 1, false, true, 3>(idx, a, lds, res);
    }
    if (a.s.vwords && a.s.vslot) {  // occupancy bitmap of the dedup table's slots in LDS (table-less regime, LDS-ring visited list)
        VS_REQUIRE(a.s.vr == 0 && a.s.lh == 0 && (uint64_t)a.s.vwords * 32 >= a.s.gcap && a.s.rc == 0,
                   "fast search: the slot bitmap needs the table-less regime and one bit per slot");
        const bool plain = !a.s.qlabel_off && !a.s.visible && !(a.s.flags & FAST_FULL_VARIANT);
        if (NCH == 3 && a.s.minw == 6 && plain) return launch_fast_tt<3, 0, false, 6, false, false, 2>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 6) return launch_fast_tt<3, 0, false, 6, false, true, 2>(idx, a, lds, res);
        return launch_fast_tt<NCH, 0, false, 1, false, true, 2>(idx, a, lds, res);
    }
    if (a.s.vwords) {  // written-bucket bitmap in LDS instead of cleared tables (table-less regime, LDS-ring visited list)
        VS_REQUIRE(a.s.vr == 0 && a.s.lh == 0 && (uint64_t)a.s.vwords * 128 >= a.s.gcap,
                   "fast search: the written-bucket bitmap needs the table-less regime and one bit per bucket");
        const bool plain = !a.s.qlabel_off && !a.s.visible && !(a.s.flags & FAST_FULL_VARIANT);
        if (NCH == 3 && a.s.minw == 6 && plain) return launch_fast_tt<3, 0, false, 6, false, false, 1>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 6) return launch_fast_tt<3, 0, false, 6, false, true, 1>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 5 && plain) return launch_fast_tt<3, 0, false, 5, false, false, 1>(idx, a, lds, res);
        if (NCH == 3 && a.s.minw == 5) return launch_fast_tt<3, 0, false, 5, false, true, 1>(idx, a, lds, res);
        return launch_fast_tt<NCH, 0, false, 1, false, true, 1>(idx, a, lds, res);
    }
    if (a.s.vr == 8) {
        if (NCH == 3) {  // the headline geometry (768 x 2 bit, 1536 x 1 bit): register-capped variants for the occupancy-bound regime
            if (a.s.minw == 4) return launch_fast_tt<3, 8, false, 4, false>(idx, a, lds, res);
            if (a.s.minw == 5) return launch_fast_tt<3, 8, false, 5, false>(idx, a, lds, res);
            if (a.s.minw == 6) return launch_fast_tt<3, 8, false, 6, false>(idx, a, lds, res);
        }
        return launch_fast_tt<NCH, 8, false, 1, false>(idx, a, lds, res);
    }
    if (NCH == 3) {
        const bool plain = !a.s.qlabel_off && !a.s.visible && !(a.s.flags & FAST_FULL_VARIANT);  // no label keys, no visibility mask
        if (a.s.minw == 5 && plain) return launch_fast_tt<3, 0, false, 5, false, false>(idx, a, lds, res);
        if (a.s.minw == 5) return launch_fast_tt<3, 0, false, 5, false>(idx, a, lds, res);
        if (a.s.minw == 6 && plain) return launch_fast_tt<3, 0, false, 6, false, false>(idx, a, lds, res);
        if (a.s.minw == 6) return launch_fast_tt<3, 0, false, 6, false>(idx, a, lds, res);
        if (a.s.minw == 7) return launch_fast_tt<3, 0, false, 7, false>(idx, a, lds, res);
        if (a.s.minw == 8) return launch_fast_tt<3, 0, false, 8, false>(idx, a, lds, res);
    }
    return launch_fast_tt<NCH, 0, false, 1, false>(idx, a, lds, res);
}

static int fast_dispatch(vs_index* idx, const FastLaunch& s, uint32_t* res) {
    FastArgs a;
    a.codes = idx->codes;
    a.nbrs = idx->nbrs;
    a.tids = idx->tids;
    a.label_off = idx->label_off;
    a.label_val = idx->label_val;
    a.label_mask = idx->label_mask;
    a.label_bit = idx->label_bit;
    a.nbr_mask = nullptr;
    if (s.qlabel_off && idx->label_mask && !s.build) {  // label-filtered scans: the neighbors' masks next to the neighbor rows
        VS_TRY(vs_refresh_neighbor_masks(idx));
        if (idx->nbr_mask_valid && vs_neighbor_masks_wanted(idx)) a.nbr_mask = idx->nbr_mask;
    }
    a.ls_labels = idx->ls_labels;
    a.ls_nodes = idx->ls_nodes;
    a.code_stride = idx->code_stride;
    a.nbr_stride = idx->nbr_stride;
    a.R = idx->d.num_neighbors;
    a.n = idx->d.n;
    a.n_ls = idx->d.n_label_starts;
    a.default_start = idx->d.default_start;
    a.s = s;
    if (!a.s.glimit) a.s.glimit = s.gcap / 4 * 3 - WAVE;  // 75 % load, less the ids one visit can add
    const size_t lds = fast_lds_bytes(idx, s);
    VS_REQUIRE(lds <= 160 * 1024, "fast search state does not fit LDS (%zu B)", lds);
    VS_REQUIRE(((s.hl + 1) & s.hl) == 0 && s.hl >= 63, "fast search: hl must be 2^k - 1 >= 63");
    VS_REQUIRE(s.vr == 8 || (s.vr == 0 && s.vcap >= 64),
               "fast search: visited list must be 8 register pairs or a ring of >= 64 entries");
    VS_REQUIRE(s.lh % 4 == 0 && (s.lh == 0 || s.lh >= 256) && s.gcap % 4 == 0 && s.gcap >= 256 &&
                   (uint64_t)s.lh + s.gcap + (s.vslot == 2 ? s.ocap : 0u) <= (1ull << s.sb) && s.hcap >= s.hl && s.gstride % 2 == 0 &&
                   s.gstride >= s.hcap - s.hl + 2,
               "fast search: bad dedup table / spill geometry");
    const uint32_t nch = (idx->code_stride + 7) / 8;
    switch (nch) {
        case 1: return launch_fast_t<1>(idx, a, lds, res);
        case 2: return launch_fast_t<2>(idx, a, lds, res);
        case 3: return launch_fast_t<3>(idx, a, lds, res);
        case 4: return launch_fast_t<4>(idx, a, lds, res);
        case 5:
        case 6: return launch_fast_t<6>(idx, a, lds, res);
        default: return launch_fast_t<0>(idx, a, lds, res);
    }
}

int launch_search_fast(vs_index* idx, const FastLaunch& s) {
    if (s.nq == 0) return VS_OK;
    VS_REQUIRE(!s.persist || (s.scan_counter && s.pool_slots >= std::min(s.persist, s.nq)),
               "fast search: a persistent grid needs a scan counter and one region per workgroup");
    return fast_dispatch(idx, s, nullptr);
}

// scans of the instantiation `s` selects that are resident on the device at once (the size of a persistent grid)
int fast_resident_scans(vs_index* idx, const FastLaunch& s, uint32_t* out) { return fast_dispatch(idx, s, out); }
#endif  // VS_FAST_TU
