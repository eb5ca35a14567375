/*
 * vs_oracle.h — CPU ORACLE for the StreamingDiskANN search hot path of timescale/pgvectorscale.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load liboracle.so.  The product library (libvsgpu.so) never
 * links, loads or calls anything declared here.
 *
 * It is a C++17 restatement (flat arrays replace PostgreSQL pages) of the reference's Rust code.
 * Paths are relative to /root/reference/pgvectorscale/src/access_method/ ("AM/").
 *
 * Parity pin status (see oracle/README.md):
 *   - labels overlap / smallint[] && : pinned by the reference's own known-answer tests.
 *   - f32 distances: pinned at the reference's own tolerance (|simd - scalar| < 1e-6 on its
 *     2000-d normalised ramps, AM/distance/distance_x86.rs:41-62); BIT-level lane order of
 *     simdeez::horizontal_add_ps is restated from memory of simdeez 1.0.8 -> "bit parity unpinned".
 *   - SbqQuantizer::quantize / add_sample, distance_xor_optimized, ListSearchResult tie order
 *     (Rust std BinaryHeap mechanics): NO golden vector exists in the reference -> "parity unpinned":
 *     this restatement is the definition the GPU path is held to.
 */
#ifndef VS_ORACLE_H
#define VS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSO_INVALID_NODE 0xFFFFFFFFu /* InvalidBlockNumber sentinel in neighbor lists, AM/sbq/node.rs:260-285 */

enum { VSO_COSINE = 0, VSO_L2 = 1, VSO_IP = 2 }; /* AM/distance/mod.rs:11-15 */

/* ---- L1 distance arithmetic (AM/distance/mod.rs, distance_x86.rs) ------------------------------ */
uint64_t vso_distance_xor(const uint64_t* a, const uint64_t* b, size_t words);   /* :266-323 */
float vso_distance_l2(const float* a, const float* b, size_t n);                 /* :88-104,325-377 (AVX2 lane order) */
float vso_inner_product(const float* a, const float* b, size_t n);               /* :380-435 (AVX2+FMA lane order) */
float vso_distance_inner_product(const float* a, const float* b, size_t n);      /* :175-190  = -dot */
float vso_distance_cosine(const float* a, const float* b, size_t n);             /* distance_x86.rs:34-36 = max(0,1-dot) */
float vso_distance_l2_unoptimized(const float* a, const float* b, size_t n);     /* :106-117 */
float vso_inner_product_unoptimized(const float* a, const float* b, size_t n);   /* :211-214 */
float vso_distance_cosine_unoptimized(const float* a, const float* b, size_t n); /* :216-223 */
/* same arithmetic through real AVX2/FMA intrinsics (timing leg + cross-check of the scalar lane emulation).
 * Returns NaN if the build/CPU has no AVX2+FMA. */
float vso_distance_l2_avx2(const float* a, const float* b, size_t n);
float vso_inner_product_avx2(const float* a, const float* b, size_t n);
int vso_have_avx2(void);
/* ns per call of distance_l2 (0) / distance_cosine (1) / inner product (2) at D = 2000, distance_xor_optimized (3) at 1536
 * bits, on the inputs of the reference's criterion benches (benches/distance.rs:144-161,299-338) */
double vso_micro_bench(int which, uint64_t iters);
/* returns 1 if the vector was rescaled, 0 if left alone (AM/distance/mod.rs:225-253) */
int vso_preprocess_cosine(float* v, size_t n);
float vso_distance_by_type(int distance_type, const float* a, const float* b, size_t n);

/* ---- SBQ quantizer (AM/sbq/quantize.rs) --------------------------------------------------------- */
size_t vso_quantized_size(size_t dims, unsigned bits);                            /* :37-45 */
void vso_quantize(const float* mean, const float* m2, uint64_t count, unsigned bits,
                  const float* v, size_t dims, uint64_t* out /* [quantized_size] */); /* :52-102 */
/* Welford training pass over rows [0,n) in order; mean/m2 must be zero-initialised by the caller with
 * *count == 0 for a fresh training (or carry state to continue).  (:104-152) */
void vso_train(float* mean, float* m2, uint64_t* count, unsigned bits, const float* rows, size_t n, size_t dims);
/* default bits per dimension: 2 if dims_to_index < 900 else 1 (AM/meta_page.rs:312-323) */
unsigned vso_default_bits(size_t dims_to_index);

/* ---- labels (AM/labels/mod.rs) ------------------------------------------------------------------ */
size_t vso_labelset_from(int16_t* labels, size_t n);                              /* sort + dedup in place; :30-37 */
int vso_labels_overlap(const int16_t* a, size_t na, const int16_t* b, size_t nb); /* :124-142 */
int vso_labels_contains_intersection(const int16_t* self, size_t ns, const int16_t* a, size_t na,
                                     const int16_t* b, size_t nb);                 /* :85-111 */
/* the SQL operator smallint[] && smallint[] (AM/mod.rs:283-314); *_null[i] != 0 marks a NULL element */
int vso_smallint_array_overlap(const int16_t* l, const uint8_t* l_null, size_t nl,
                               const int16_t* r, const uint8_t* r_null, size_t nr);

/* ---- flat index ("pages" -> arrays) -------------------------------------------------------------- */
typedef struct {
    uint32_t n;             /* number of index nodes; node id = dense position                           */
    uint32_t dim_full;      /* MetaPage.num_dimensions                 (AM/meta_page.rs:179-210)         */
    uint32_t dim_index;     /* MetaPage.num_dimensions_to_index                                           */
    uint32_t bits;          /* num_bits_per_dimension                                                     */
    uint32_t words;         /* W = quantized_size(dim_index, bits)                                        */
    uint32_t num_neighbors; /* R, fixed slots per node                                                    */
    uint32_t nbr_stride;    /* row stride (>= R) of `nbrs` in u32 elements                                */
    uint32_t distance_type; /* VSO_COSINE | VSO_L2 | VSO_IP                                               */
    uint32_t has_labels;
    uint32_t default_start; /* StartNodes.default_node, VSO_INVALID_NODE if the graph is empty            */
    uint32_t n_label_starts;
    const int16_t* label_start_labels;  /* sorted; StartNodes.labeled_nodes keys   (AM/graph/start_nodes.rs) */
    const uint32_t* label_start_nodes;  /* parallel values                                                */
    const uint64_t* codes;     /* [n][words]                 ArchivedSbqNode.bq_vector                    */
    const uint32_t* nbrs;      /* [n][nbr_stride], list ends at first VSO_INVALID_NODE or after R slots   */
    const uint64_t* heap_tids; /* [n]  (block<<16)|offset ; offset==0 (InvalidOffsetNumber) => deleted    */
    const float* vecs;         /* [n][dim_full]  the heap table's vector column (raw, un-normalised)      */
    const uint32_t* label_off; /* [n+1] CSR offsets (has_labels)                                          */
    const int16_t* label_val;  /* sorted, dedup'ed label sets                                             */
    const float* mean;         /* [dim_index] SbqMeans                                                    */
    const float* m2;           /* [dim_index] (bits > 1)                                                  */
    uint64_t count;
    uint32_t storage_plain;    /* 1: `plain` storage (AM/plain/storage.rs): candidates are scored with the full-precision
                                  distance to the node's stored vector instead of SBQ Hamming; no label filters     */
    const uint8_t* visible;    /* [n] or NULL (= every tuple visible): what index_fetch_tuple(xs_snapshot) says about the node's
                                  heap tuple; 0 => get_full_distance_for_resort returns None (AM/sbq/storage.rs:313-317)   */
} vso_index;

typedef struct {
    uint64_t calls, node_reads, node_heap_reads, quantized_distance_comparisons, full_distance_comparisons,
        visited_nodes, candidate_nodes, next_calls, next_calls_with_resort;
} vso_stats; /* AM/stats.rs:68-125 + TSVResponseIterator counters AM/scan.rs:162-174 */

/* ---- one scan: amrescan + amgettuple (AM/scan.rs:336-436) ---------------------------------------- */
typedef struct vso_scan vso_scan;
/* query: dim_full raw floats, or NULL for the SQL-NULL query (zero vector, no labels; AM/labels/mod.rs:214-216).
 * labels: NULL => no scan key (nkeys==0); otherwise the smallint[] scan key (unsorted, may be empty).
 * search_list_size = GUC diskann.query_search_list_size, rescore = diskann.query_rescore (AM/guc.rs:3-4). */
vso_scan* vso_scan_begin(const vso_index* idx, const float* query, const int16_t* labels, size_t n_labels,
                         int has_label_key, uint32_t search_list_size, uint32_t rescore);
/* one amgettuple call: 1 = row produced, 0 = scan exhausted.  dist = reranked f32 distance (NaN when
 * rescore==0, where the reference never computes it). */
int vso_scan_gettuple(vso_scan* s, uint32_t* node, uint64_t* heap_tid, float* dist);
/* the raw SBQ-ordered stream: one TSVResponseIterator::next (AM/scan.rs:210-242). ham = Hamming distance. */
int vso_scan_next_sbq(vso_scan* s, uint32_t* node, uint64_t* heap_tid, uint32_t* ham);
int vso_scan_xs_recheck(const vso_scan* s); /* AM/scan.rs:350-352 */
void vso_scan_stats(const vso_scan* s, vso_stats* out);
void vso_scan_end(vso_scan* s);

/* batch drivers (used for parity sweeps and the cpu_baseline timing leg).  For each query writes the first
 * k amgettuple rows; rows past the end of a scan are filled with VSO_INVALID_NODE / NaN.
 * queries [nq][dim_full]; qlabel_off NULL => no label keys. n_threads<=1 => single thread. */
void vso_search_batch(const vso_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                      uint32_t nq, uint32_t search_list_size, uint32_t rescore, uint32_t k, uint32_t n_threads,
                      uint32_t* out_nodes, float* out_dist, vso_stats* stats_sum /* may be NULL */);
/* first m entries of the SBQ-ordered stream (rerank bypassed) */
void vso_stream_batch(const vso_index* idx, const float* queries, const int16_t* qlabels, const uint32_t* qlabel_off,
                      uint32_t nq, uint32_t search_list_size, uint32_t m, uint32_t n_threads,
                      uint32_t* out_nodes, uint32_t* out_ham, vso_stats* stats_sum);

/* ---- helpers that are NOT on the reference's search path (test-graph manufacture / ground truth) -- */
/* Sequential Vamana build with SBQ Hamming distances, modelled on Graph::insert / prune_neighbors
 * (AM/graph/mod.rs:392-488,637-717) — used only to manufacture test graphs; not claimed bit-identical
 * to the reference build (which itself is HashSet-order dependent, AM/graph/mod.rs:317-326). */
/* push / pop replay on the candidate heap; ops = [n_ops][2] (key, id), key 0xFFFFFFFF = pop; returns the number of ids written */
size_t vso_heap_replay(const uint32_t* ops, size_t n_ops, uint32_t* out_ids);
void vso_build_graph(uint32_t n, uint32_t words, const uint64_t* codes, uint32_t num_neighbors, uint32_t nbr_stride,
                     uint32_t search_list_size, double max_alpha, uint32_t* nbrs /* out [n][nbr_stride] */,
                     uint32_t* default_start /* out */);
/* the same over a labeled vector set: Graph::insert's two passes + label-aware pruning (AM/graph/mod.rs:392-488,637-662,
 * AM/labels/mod.rs:85-111); returns the number of per-label start nodes written (ascending label order) */
uint32_t vso_build_graph_labeled(uint32_t n, uint32_t words, const uint64_t* codes, const uint32_t* label_off,
                                 const int16_t* label_val, uint32_t num_neighbors, uint32_t nbr_stride,
                                 uint32_t search_list_size, double max_alpha, uint32_t* nbrs, uint32_t* default_start,
                                 int16_t* start_labels, uint32_t* start_nodes);
/* exact f32 brute-force top-k by the reference distance function (ground truth for recall) */
void vso_bruteforce_topk(const vso_index* idx, const float* queries, uint32_t nq, uint32_t k, uint32_t n_threads,
                         uint32_t* out_nodes, float* out_dist);
/* flat Hamming scan top-k with (hamming, node id) ascending order — oracle for the K5 scan kernel */
void vso_hamming_scan_topk(const uint64_t* codes, uint32_t n, uint32_t words, const uint64_t* qcodes, uint32_t nq,
                           uint32_t k, uint32_t* out_nodes, uint32_t* out_ham);
void vso_hamming_scan_topk_filtered(const uint64_t* codes, uint32_t n, uint32_t words, const uint32_t* label_off,
                                    const int16_t* label_val, const uint64_t* heap_tids, const uint64_t* qcodes,
                                    const int16_t* qlabels, const uint32_t* qlabel_off, uint32_t nq, uint32_t k,
                                    uint32_t* out_nodes, uint32_t* out_ham);

#ifdef __cplusplus
}
#endif
#endif
