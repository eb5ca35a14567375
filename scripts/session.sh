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
    fuzzv) timeout $(( ${arg:-40} * 8 + 120 )) python scripts/fuzz_variants.py --gpu --cases ${arg:-40} --seed $RANDOM 2>&1 | tail -4 | tee $O/fuzz_variants_gpu.txt ;;
    fuzzv) timeout $(( ${arg:-40} * 8 + 120 )) python scripts/fuzz_variants.py --gpu --cases ${arg:-40} --seed $RANDOM 2>&1 | tail -4 | tee $O/fuzz_variants_gpu.txt ;;
    fuzz)  timeout $(( ${arg:-150} + 120 )) python scripts/fuzz_emu.py --gpu --seconds ${arg:-150} --seed $RANDOM 2>&1 | tail -3 | tee $O/fuzz_gpu.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
