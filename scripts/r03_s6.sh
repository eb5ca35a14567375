#!/bin/bash
# round 3, GPU session 6: does the allocation history (index built in process vs loaded) change the search kernel's speed?
# + the neighbor-mask cache of the label-filtered scans at 5M x 1536 (A/B by environment switch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s6
O=gpurun_out/s6
timeout 1500 python scripts/alloc_history.py 2>&1 | grep -E "graph|search" | tee $O/alloc_history_50m.txt
for v in 1 0; do
  VS_F_NBRMASK=$v timeout 900 python bench.py --n 5000000 --dim 1536 --distance cosine --labels 32 --steps 6 --warmup 2 --graph-cache /tmp/g5 --skip-cpu --scan-nq 0 --pcie-steps 0 > $O/bench_cfg5_5m_nbrmask$v.json 2> $O/bench_cfg5_5m_nbrmask$v.err
done
rm -f /tmp/g5.*
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
          "recall", j["recall_heldout"], "met", j["recall_target_met"], "kernel ms", r["avg_kernel_ms"], "frac", r["frac"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done | tee $O/summary.txt
