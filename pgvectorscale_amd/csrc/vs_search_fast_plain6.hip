// vs_search_fast_plain6.hip — translation unit 1 of vs_search_fast.hip (see its header): the instantiations of k_search_fast that the
// unfiltered scans of the usual index run (16-bit tables, six waves per SIMD), built with the options csrc/Makefile names for them.
#define VS_FAST_TU 1
#include "vs_search_fast.hip"
