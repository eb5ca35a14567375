#!/bin/bash
# gpurun --timeout 780 -- 'bash scripts/r02_stats.sh 3,196'
# Last session of the round: the GPU test tier and smoke() on the final tree, then the rocprofv3 kernel summary of the default
# bench at the operating point its sweep finds (argument: L,rescore), then the in-kernel phase clocks at that point.
LS=${1:-3,196}
L=${LS%,*}; S=${LS#*,}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
O=gpurun_out/r02
CACHE=/tmp/vs_graph
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee -a $O/gpu_tests.txt
timeout 500 python bench.py --skip-cpu --graph-cache $CACHE --fixed $LS --steps 4 --warmup 1 2> $O/bench_50m_fixed.err > $O/bench_50m_fixed.json
tail -3 $O/bench_50m_fixed.err; cut -c1-200 $O/bench_50m_fixed.json
rm -rf $O/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --skip-cpu --graph-cache $CACHE --fixed $LS > $O/bench_50m_under_rocprof.json 2> $O/bench_50m_prof.err
python scripts/summarize_rocprof.py $O/prof/bench_kernel_stats.csv $O/kernel_stats_50m.csv "rocprofv3 --kernel-trace --stats -- python bench.py --skip-cpu --graph-cache ... --fixed $LS (50M x 768 l2, 131072 scans per launch; index loaded from the cache the plain run wrote)"
head -8 $O/kernel_stats_50m.csv
rm -rf $O/prof
VS_PHASE=1 timeout 240 python scripts/perf_search.py --n 50000000 --nq 131072 --L $L --rescore $S --reps 2 --configs VS_FAST=1 --graph-cache $CACHE 2>&1 | grep -E "VS_PHASE|search " | tail -4 | tee $O/phase_50m.txt
rm -f ${CACHE}.*
