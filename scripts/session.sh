#!/bin/bash
# One parametrised GPU session script (replaces the per-session one-offs of rounds 2-4):
#   gpurun --timeout S -- 'bash scripts/session.sh <tag> step [step ...]'      outputs under gpurun_out/<tag>/
# steps: cal | tests[:<pytest -k expr>] | diag[:n] | bench[:extra args] | ab:<n>:<configs> | fuzz[:seconds] | prof | pmc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|amdgpu.ids'
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "=== $step ($(date +%T))"
  case $name in
    cal)   bash scripts/pmc_calibrate.sh > $O/pmc_calibrate.txt 2>&1; cp gpurun_out/pmccal/pmc_calibration_randmem.json $O/ 2>/dev/null; tail -5 $O/pmc_calibrate.txt ;;
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | grep -Ev "$NOBANNER" | tail -6 | tee $O/gpu_tests_subset.txt
           else timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -Ev "$NOBANNER" | tail -6 | tee $O/gpu_tests.txt; fi ;;
    diag)  N=${arg:-50000000}
           timeout 900 python scripts/diag_state.py --n $N --phase build --graph /tmp/diag_graph $DIAG_ARGS 2>&1 | grep -Ev "$NOBANNER" | sed 's/ *GPU\[0\].*//' | tee $O/diag_state_$N.txt
           [ -n "$DIAG_BUILD_ONLY" ] || timeout 600 python scripts/diag_state.py --n $N --phase load --graph /tmp/diag_graph $DIAG_ARGS 2>&1 | grep -Ev "$NOBANNER" | sed 's/ *GPU\[0\].*//' | tee -a $O/diag_state_$N.txt ;;
    place) N=${arg:-50000000}   # placement study in a process that LOADS the graph a diag step of this session wrote (or builds it)
           PH=load; [ -f /tmp/diag_graph ] || PH=build
           timeout 900 python scripts/diag_state.py --n $N --phase $PH --graph /tmp/diag_graph --placement ${PLACE_TRIALS:-24} 2>&1 | grep -Ev "$NOBANNER" | sed 's/ *GPU\[0\].*//' | uniq | tee $O/placement_$N.txt ;;
    pmap)  timeout 1200 python scripts/placement_map.py --n ${arg:-50000000} $PMAP_ARGS 2>&1 | grep -Ev "$NOBANNER" | tee $O/placement_map.txt ;;
    bench) timeout 2400 python bench.py $arg > $O/bench_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').json 2> $O/bench.err; tail -3 $O/bench.err; tail -c 1500 $O/bench_*.json ;;
    ab)    N=${arg%%:*}; CFG=${arg#*:}
           timeout 1500 python scripts/perf_search.py --n $N --nq 262144 --L 3 --rescore 195 --reps 4 --graph-cache /tmp/g --configs "$CFG" 2>&1 | grep -Ev "$NOBANNER" | tee $O/ab_$N.txt ;;
    pmcissue) bash scripts/pmc_issue.sh ${arg:-10000000} 262144 3 195 2>&1 | grep -Ev "$NOBANNER" | tee $O/pmc_issue.txt ;;
    cpool) timeout 900 python scripts/cursor_pool_concurrency.py --n ${arg:-1000000} 2>&1 | grep -Ev "$NOBANNER" | tee $O/cursor_pool_concurrency.txt ;;
    fuzzv) timeout $(( ${arg:-40} * 8 + 120 )) python scripts/fuzz_variants.py --gpu --cases ${arg:-40} --seed $RANDOM 2>&1 | tail -4 | tee $O/fuzz_variants_gpu.txt ;;
    fuzzv) timeout $(( ${arg:-40} * 8 + 120 )) python scripts/fuzz_variants.py --gpu --cases ${arg:-40} --seed $RANDOM 2>&1 | tail -4 | tee $O/fuzz_variants_gpu.txt ;;
    final) # the evidence set of the frozen tree: the driver's bench command (builds, writes the graph cache), the same under rocprofv3
           # --kernel-trace --stats, two --pmc passes at the bench's operating point, the bench once more WITH roofline.traffic
           timeout 2400 python bench.py --steps 20 --warmup 5 --graph-cache /tmp/g $FINAL_BENCH_ARGS > $O/bench_50m.json 2> $O/bench_50m.err; tail -3 $O/bench_50m.err
           LS=$(python - <<PY
import json
j = json.loads(open("$O/bench_50m.json").read().strip().splitlines()[-1])
print(j["config"]["search_list_size"], j["config"]["rescore"], j["config"]["queries_per_step_per_gpu"])
PY
)
           set -- $LS; L=$1; S=$2; NQ=$3; echo "operating point L=$L rescore=$S nq=$NQ"
           rm -rf gpurun_out/prof_final
           timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python bench.py --steps 20 --warmup 5 --skip-cpu --extras off --graph-cache /tmp/g > $O/bench_50m_under_rocprof.json 2> $O/bench_50m_under_rocprof.err
           python scripts/summarize_rocprof.py gpurun_out/prof_final/bench_kernel_stats.csv $O/kernel_stats_50m.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --skip-cpu --extras off --graph-cache ... (50M x 768 l2, $NQ scans per launch, L=$L rescore=$S; index loaded from the cache the plain bench run wrote)"
           head -8 $O/kernel_stats_50m.csv
           timeout 1500 bash scripts/pmc_traffic.sh 50000000 $NQ $L $S /tmp/g 2>&1 | tail -40 > $O/pmc_traffic.log
           cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m.json
           mkdir -p profiles/${PROFROUND:-r06} && cp $O/pmc_search_traffic_50m.json profiles/${PROFROUND:-r06}/pmc_search_traffic_50m.json
           timeout 900 python bench.py --steps 20 --warmup 5 --skip-cpu --extras off --graph-cache /tmp/g > $O/bench_50m_with_traffic.json 2> $O/bench_50m_with_traffic.err
           python - <<PY | tee $O/summary.txt
import json
for f in ("bench_50m", "bench_50m_under_rocprof", "bench_50m_with_traffic"):
    try:
        j = json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
              "recall", j["recall_at_k"], j["recall_validate_lower95"], j["recall_heldout"], j.get("recall_heldout_lower95"), "met", j["recall_target_met"],
              "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "traffic", r["traffic"],
              "src", (r.get("traffic_source") or {}).get("same_kernel_sources_as_this_build"),
              "pcie", (j.get("pcie_inclusive") or {}).get("value"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
        for x in ("default_gucs", "harder_corpus"):
            e = j.get(x)
            if e:
                print("  ", x, {k: e.get(k) for k in ("value", "search_list_size", "rescore", "recall_validate_lower95", "recall_heldout_lower95", "recall_target_met", "gpu_rows_identical", "seconds", "error")}, (e.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
           ;;
    fuzz)  timeout $(( ${arg:-150} + 120 )) python scripts/fuzz_emu.py --gpu --seconds ${arg:-150} --seed $RANDOM 2>&1 | tail -3 | tee $O/fuzz_gpu.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
