#!/bin/bash
# round 3, final evidence session on the frozen tree:
#   1. the GPU tier                                   -> gpu_tests.txt
#   2. python bench.py --steps 20 --warmup 5          -> bench_50m.json (the driver's command; writes the graph cache)
#   3. the same under rocprofv3 --kernel-trace --stats -> kernel_stats_50m.csv
#   4. two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) at the bench's operating point -> pmc_search_traffic_50m.json
#   5. configs[4], [1], [2] and the 10M line on the `mid` corpus
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 1800 python bench.py --steps 20 --warmup 5 --graph-cache /tmp/g > $O/bench_50m.json 2> $O/bench_50m.err
tail -3 $O/bench_50m.err
LS=$(python - <<'PY'
import json
j = json.loads(open("gpurun_out/final/bench_50m.json").read().strip().splitlines()[-1])
print(j["config"]["search_list_size"], j["config"]["rescore"], j["config"]["queries_per_step_per_gpu"])
PY
)
set -- $LS; L=$1; S=$2; NQ=$3
echo "operating point L=$L rescore=$S nq=$NQ"
rm -rf gpurun_out/prof_final
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python bench.py --steps 20 --warmup 5 --skip-cpu --graph-cache /tmp/g > $O/bench_50m_under_rocprof.json 2> $O/bench_50m_under_rocprof.err
python scripts/summarize_rocprof.py gpurun_out/prof_final/bench_kernel_stats.csv $O/kernel_stats_50m.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --skip-cpu --graph-cache ... (50M x 768 l2, $NQ scans per launch, L=$L rescore=$S; index loaded from the cache the plain bench run wrote)"
head -8 $O/kernel_stats_50m.csv
timeout 1500 bash scripts/pmc_traffic.sh 50000000 $NQ $L $S /tmp/g 2>&1 | tail -30 > $O/pmc_traffic.log
cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m.json
rm -f /tmp/g.*
timeout 1500 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 8 --warmup 2 --graph-cache /tmp/g5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
VS_F_NBRMASK=1 timeout 900 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 8 --warmup 2 --graph-cache /tmp/g5 --skip-cpu --scan-nq 0 --pcie-steps 0 > $O/bench_cfg5_nbrmask1.json 2> $O/bench_cfg5_nbrmask1.err
rm -f /tmp/g5.*
timeout 600 python scripts/fuzz_emu.py --gpu --seconds 200 --seed 777 2>&1 | tail -2 | tee $O/fuzz_gpu.txt
timeout 600 python scripts/cursor_latency.py --n 10000000 2>&1 | grep -v amdgpu.ids | tee $O/cursor_latency_10m.txt
timeout 600 python bench.py --n 1000000 --steps 20 --warmup 5 --graph-cache none > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --n 10000000 --distance cosine --steps 10 --warmup 3 --graph-cache none > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 900 python bench.py --n 10000000 --corpus-kind mid --steps 10 --warmup 3 --graph-cache none > $O/bench_10m_mid.json 2> $O/bench_10m_mid.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
          "recall", j["recall_at_k"], j["recall_validate"], j["recall_validate_lower95"], j["recall_heldout"], "met", j["recall_target_met"],
          "kernel ms", r["avg_kernel_ms"], "per131072", r.get("kernel_ms_per_131072_scans"), "frac", r["frac"], "traffic", r["traffic"],
          "cpu", (j.get("cpu_baseline") or {}).get("value"), "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done | tee $O/summary.txt
