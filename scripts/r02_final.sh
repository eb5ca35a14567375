#!/bin/bash
# gpurun --timeout 3000 -- 'bash scripts/r02_final.sh'
# Everything profiles/r02 holds for the final tree, in one GPU session:
#   1. the GPU test tier
#   2. the default bench (50M x 768 L2, the configuration BASELINE.json quotes its metric on) with the CPU oracle; writes the graph cache
#   3. rocprofv3 --kernel-trace --stats of the same command at the operating point step 2 found (index loaded from the cache)
#   4. HBM traffic of k_search_fast at that operating point (two rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE) + one SQ pass
#   5. BASELINE configs[2] (10M cosine), configs[4] (20M x 1536, label-filtered), configs[1] (1M), and 10M of the SURVEY 8(d) corpus
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
O=gpurun_out/r02
CACHE=/tmp/vs_graph
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 1500 python bench.py --graph-cache $CACHE 2> $O/bench_50m.err > $O/bench_50m.json
tail -6 $O/bench_50m.err; cut -c1-400 $O/bench_50m.json
LS=$(python -c "import json;j=json.loads(open('$O/bench_50m.json').read().strip().splitlines()[-1]);print(str(j['config']['search_list_size'])+','+str(j['config']['rescore']))")
L=${LS%,*}; S=${LS#*,}
echo "operating point $L / $S"
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --skip-cpu --graph-cache $CACHE --fixed $LS > $O/bench_50m_prof.json 2> $O/bench_50m_prof.err
python scripts/summarize_rocprof.py $O/prof/bench_kernel_stats.csv $O/kernel_stats_50m.csv "rocprofv3 --kernel-trace --stats -- python bench.py --skip-cpu --graph-cache ... --fixed $LS (50M x 768 l2, 131072 scans per launch; index loaded from the cache the plain bench run wrote)"
head -8 $O/kernel_stats_50m.csv
rm -rf $O/prof
timeout 900 bash scripts/pmc_traffic.sh 50000000 131072 $L $S $CACHE 2>&1 | tail -30
cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m.json 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python scripts/perf_search.py --n 50000000 --nq 131072 --L $L --rescore $S --reps 2 --configs VS_FAST=1 --graph-cache $CACHE > $O/pmc_sq.log 2>&1 && python scripts/pmc_summary.py $O/pmc_sq/p_counter_collection.csv | tee $O/pmc_sq_50m.txt
rm -rf $O/pmc_sq ${CACHE}.*
timeout 600 python bench.py --n 10000000 --distance cosine --steps 6 --warmup 2 2> $O/bench_cfg3.err > $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
timeout 1200 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 4 --warmup 1 2> $O/bench_cfg5.err > $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err
timeout 300 python bench.py --n 1000000 --steps 8 --warmup 2 2> $O/bench_cfg2.err > $O/bench_cfg2.json; tail -3 $O/bench_cfg2.err
timeout 600 python bench.py --n 10000000 --corpus-kind survey --steps 4 --warmup 1 2> $O/bench_10m_survey.err > $O/bench_10m_survey.json; tail -4 $O/bench_10m_survey.err
ls -la $O
