#!/bin/bash
# round 3, GPU session 2: the GPU tier on the merged tree, 10M A/B of the two-stream pipeline, phase clocks + SQ counters of the
# merged k_search_fast, three candidate "mid" corpora at 10M, and the 50M line with one and two batches in flight.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s2
O=gpurun_out/s2
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_tests.txt
B="python bench.py --n 10000000 --graph-cache /tmp/g --skip-cpu"
timeout 600 $B --steps 8 --warmup 2 --pipeline 1 > $O/bench_10m_p1.json 2> $O/bench_10m_p1.err
timeout 600 $B --steps 8 --warmup 2 --pipeline 2 > $O/bench_10m_p2.json 2> $O/bench_10m_p2.err
VS_PHASE=1 timeout 600 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 2 --graph-cache /tmp/g > $O/phase_10m.txt 2>&1
timeout 900 bash scripts/pmc_issue.sh 10000000 131072 3 196 /tmp/g > $O/pmc_issue_10m.txt 2>&1
cp gpurun_out/pmc_issue_A.txt gpurun_out/pmc_issue_B.txt $O/ 2>/dev/null
for v in "mid" "mid --intra-pct 60 --noise-pct 20" "mid --latent-dim 128 --intra-pct 60 --noise-pct 20"; do
    tag=$(echo "$v" | tr -d ' -' )
    timeout 600 python bench.py --n 10000000 --corpus-kind $v --graph-cache none --skip-cpu --scan-nq 0 --pcie-steps 0 --steps 4 --warmup 1 \
        > $O/bench_10m_$tag.json 2> $O/bench_10m_$tag.err
done
timeout 1500 python bench.py --steps 20 --warmup 5 --pipeline 1 --graph-cache /tmp/g > $O/bench_50m_p1.json 2> $O/bench_50m_p1.err
timeout 900 python bench.py --steps 20 --warmup 5 --pipeline 2 --graph-cache /tmp/g --skip-cpu > $O/bench_50m_p2.json 2> $O/bench_50m_p2.err
rm -f /tmp/g.*
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
          "recall", j["recall_at_k"], j["recall_validate"], j["recall_validate_lower95"], j["recall_heldout"], "met", j["recall_target_met"],
          "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "retimed", j["retimed_after_heldout_check"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done | tee $O/summary.txt
