#!/bin/bash
# round 3, the last evidence session (epoch tags off = the shipped default): GPU tier, the driver's bench command, the two PMC
# passes at its operating point, and — if the minutes last — the same command under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/final2
O=gpurun_out/final2
T0=$(date +%s)
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 420 python bench.py --steps 20 --warmup 5 --graph-cache /tmp/g > $O/bench_50m.json 2> $O/bench_50m.err
tail -2 $O/bench_50m.err
LS=$(python - <<'PY'
import json
j = json.loads(open("gpurun_out/final2/bench_50m.json").read().strip().splitlines()[-1])
print(j["config"]["search_list_size"], j["config"]["rescore"], j["config"]["queries_per_step_per_gpu"])
PY
)
set -- $LS; L=$1; S=$2; NQ=$3
echo "operating point L=$L rescore=$S nq=$NQ after $(( $(date +%s) - T0 )) s"
timeout 300 bash scripts/pmc_traffic.sh 50000000 $NQ $L $S /tmp/g 2>&1 | tail -30 > $O/pmc_traffic.log
cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m.json
echo "pmc done after $(( $(date +%s) - T0 )) s"
rm -rf gpurun_out/prof_final
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python bench.py --steps 8 --warmup 2 --skip-cpu --scan-nq 0 --pcie-steps 0 --fixed $L,$S --graph-cache /tmp/g > $O/bench_50m_under_rocprof.json 2> $O/bench_50m_under_rocprof.err
python scripts/summarize_rocprof.py gpurun_out/prof_final/bench_kernel_stats.csv $O/kernel_stats_50m.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 2 --skip-cpu --fixed $L,$S --graph-cache ... (50M x 768 l2, $NQ scans per launch; index loaded from the cache the plain bench run wrote)"
head -4 $O/kernel_stats_50m.csv
rm -f /tmp/g.*
echo "done after $(( $(date +%s) - T0 )) s"
