// vs_extra.hip — K5 flat scan, build-side kernels and the synthetic corpus generator (filled in incrementally).
#include "vs_internal.h"

#define NOT_YET(name)                                      \
    do {                                                   \
        vs_set_error(name ": not implemented in this build"); \
        return VS_ERR_STATE;                               \
    } while (0)

extern "C" int vs_scan_topk(vs_index*, const uint64_t*, uint32_t, uint32_t, uint32_t*, uint32_t*) { NOT_YET("vs_scan_topk"); }
extern "C" int vs_sbq_train(vs_index*) { NOT_YET("vs_sbq_train"); }
extern "C" int vs_sbq_quantize_corpus(vs_index*) { NOT_YET("vs_sbq_quantize_corpus"); }
extern "C" int vs_build_graph(vs_index*, uint32_t, double, uint32_t, uint64_t) { NOT_YET("vs_build_graph"); }
extern "C" int vs_datagen_fill(vs_ctx*, const vs_datagen_params*, uint64_t, uint64_t, float*) { NOT_YET("vs_datagen_fill"); }
extern "C" int vs_bruteforce_topk(vs_index*, const float*, uint32_t, uint32_t, uint32_t*, float*) { NOT_YET("vs_bruteforce_topk"); }
