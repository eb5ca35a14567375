#!/bin/bash
# round 3, GPU session 3: BASELINE configs[4] (20M x 1536, label-filtered, label masks), configs[1] (1M), configs[2] (10M cosine)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s3
O=gpurun_out/s3
timeout 1500 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 8 --warmup 2 --graph-cache none > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --n 1000000 --steps 20 --warmup 5 --graph-cache none > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --n 10000000 --distance cosine --steps 10 --warmup 3 --graph-cache none > $O/bench_cfg3.json 2> $O/bench_cfg3.err
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
          "recall", j["recall_at_k"], j["recall_validate"], j["recall_validate_lower95"], j["recall_heldout"], "met", j["recall_target_met"],
          "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "cpu", (j.get("cpu_baseline") or {}).get("value"),
          "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done | tee $O/summary.txt
tail -3 $O/*.err
