#!/bin/bash
# bench.py at ONE fixed operating point under several option sets (environment snapshot of vs_options), one cached graph:
#   gpurun --timeout 1800 -- 'bash scripts/ab_bench_env.sh <tag> "<bench.py args incl. --fixed L,S>" "X=1" "VS_F_SLOTMAP_FORCE=1 VS_F_HL=255" ...'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; ARGS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
i=0
for envs in "$@"; do
    i=$((i + 1))
    env $envs VS_WS_DEBUG=1 timeout 1200 python bench.py $ARGS --extras off --skip-cpu --scan-nq 0 --pcie-steps 0 --heldout-queries 0 --graph-cache /tmp/gabe > $O/bench_$i.json 2> $O/bench_$i.err
    grep -h "resident scans" $O/bench_$i.err | sort | uniq -c | sed 's/^/    /' | tee -a $O/ab.txt
    python - "$O/bench_$i.json" "$envs" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{sys.argv[2]:44s} kernel {r['avg_kernel_ms']:9.3f} ms  frac {r['frac']:.4f}  QPS {j['value']:11.1f}  L/S {j['config']['search_list_size']}/{j['config']['rescore']}  recall {j['recall_at_k']}  fallback {j['kernels'].get('fallback')}")
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
done
