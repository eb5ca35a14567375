#!/bin/bash
# round 4, SECOND final evidence session: the tree that reserves the search workspace together with the index arrays (the first final
# session, scripts/r04_final.sh on commit 575921a, found the run that builds the graph 8.8 % slower than the runs that load it;
# scripts/r04_s7.sh narrowed it down to the process that built).  Same steps 1-5 on the new frozen tree:
#   1. the GPU tier                                    -> gpu_tests.txt
#   2. python bench.py --steps 20 --warmup 5           -> bench_50m.json   (the driver's command; writes the graph cache)
#   3. the same under rocprofv3 --kernel-trace --stats -> kernel_stats_50m.csv (must agree with roofline.avg_kernel_ms)
#   4. two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) at the bench's operating point, library default variant
#                                                      -> pmc_search_traffic_50m.json (bench.py reads it as roofline.traffic)
#   5. bench.py once more (cached graph, no CPU leg): the line WITH roofline.traffic filled from step 4 -> bench_50m_with_traffic.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/final2
O=gpurun_out/final2
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|amdgpu.ids'
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -Ev "$NOBANNER" | tail -4 | tee $O/gpu_tests.txt
timeout 1800 python bench.py --steps 20 --warmup 5 --graph-cache /tmp/g > $O/bench_50m.json 2> $O/bench_50m.err
tail -3 $O/bench_50m.err
LS=$(python - <<'PY'
import json
j = json.loads(open("gpurun_out/final2/bench_50m.json").read().strip().splitlines()[-1])
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
mkdir -p profiles/r04 && cp $O/pmc_search_traffic_50m.json profiles/r04/pmc_search_traffic_50m.json
timeout 900 python bench.py --steps 20 --warmup 5 --skip-cpu --graph-cache /tmp/g > $O/bench_50m_with_traffic.json 2> $O/bench_50m_with_traffic.err
rm -f /tmp/g.*
python - <<'PY' | tee gpurun_out/final2/summary.txt
import json
for f in ("bench_50m", "bench_50m_under_rocprof", "bench_50m_with_traffic"):
    try:
        j = json.loads(open(f"gpurun_out/final2/{f}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
              "recall", j["recall_at_k"], j["recall_validate_lower95"], j["recall_heldout"], j.get("recall_heldout_lower95"), "met", j["recall_target_met"],
              "kernel ms", r["avg_kernel_ms"], "per131072", r.get("kernel_ms_per_131072_scans"), "frac", r["frac"], "traffic", r["traffic"],
              "src", (r.get("traffic_source") or {}).get("same_kernel_sources_as_this_build"),
              "pcie", (j.get("pcie_inclusive") or {}).get("value"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
