#!/bin/bash
# One GPU session producing everything profiles/ holds for the default bench configuration (50M x 768, L2):
#   1. python bench.py                      -> bench line (with the CPU oracle), also writes the graph cache
#   2. rocprofv3 --kernel-trace --stats     -> per-kernel summary of the same command (+ --graph-cache: the deterministic
#                                              6-minute index build is loaded instead of repeated)
#   3. scripts/pmc_traffic.sh               -> HBM traffic of k_search_fast (two --pmc passes, calibrated on the flat scan)
# usage: scripts/final_profile.sh [n] [distance] [L] [rescore]
N=${1:-50000000}; DIST=${2:-l2}; L=${3:-100}; S=${4:-200}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CACHE=/tmp/vs_graph
python bench.py --n $N --distance $DIST --graph-cache $CACHE 2> gpurun_out/final_bench.err > gpurun_out/final_bench.json
tail -4 gpurun_out/final_bench.err
rm -rf gpurun_out/prof_final
VS_BENCH_INPROC=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python bench.py --n $N --distance $DIST --skip-cpu --graph-cache $CACHE > gpurun_out/final_bench_prof.json 2> gpurun_out/final_bench_prof.err
python scripts/summarize_rocprof.py gpurun_out/prof_final/bench_kernel_stats.csv gpurun_out/final_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --skip-cpu --graph-cache ... ($N x 768 $DIST, 131072 scans per launch; index loaded from the cache the plain bench run wrote)"
head -8 gpurun_out/final_kernel_stats.csv
if [ "$DIST" = "l2" ]; then bash scripts/pmc_traffic.sh $N 131072 $L $S $CACHE 2>&1 | tail -25; fi
rm -f ${CACHE}.*
