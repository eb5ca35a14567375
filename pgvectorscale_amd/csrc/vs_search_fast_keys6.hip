// vs_search_fast_keys6.hip — translation unit 2 of vs_search_fast.hip (see its header): the label-key / visibility-mask counterparts
// of vs_search_fast_plain6.hip, built with the options csrc/Makefile names for them.
#define VS_FAST_TU 2
#include "vs_search_fast.hip"
