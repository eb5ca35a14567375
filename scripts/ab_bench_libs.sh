#!/bin/bash
# bench.py at ONE fixed operating point under this tree's libvsgpu.so and every pgvectorscale_amd/libvsgpu_alt_*.so (VS_LIB_PATH), one
# graph (cached in /tmp): the A/B for configurations perf_search.py cannot build (label keys, cosine, other widths).
#   gpurun --timeout 1800 -- 'bash scripts/ab_bench_libs.sh <tag> "<bench.py args incl. --fixed L,S>"'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; ARGS=$2
O=gpurun_out/$TAG; mkdir -p $O
one() {  # $1 = label, $2 = lib path or ""
    VS_LIB_TOLERANT=1 VS_LIB_PATH=$2 timeout 1200 python bench.py $ARGS --extras off --skip-cpu --scan-nq 0 --pcie-steps 0 --heldout-queries 0 --graph-cache /tmp/gab > $O/bench_$1.json 2> $O/bench_$1.err
    python - "$O/bench_$1.json" "$1" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{sys.argv[2]:28s} kernel {r['avg_kernel_ms']:9.3f} ms  frac {r['frac']:.4f}  QPS {j['value']:11.1f}  L/S {j['config']['search_list_size']}/{j['config']['rescore']}  recall {j['recall_at_k']}")
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
}
one main ""
for lib in $(ls pgvectorscale_amd/libvsgpu_alt_*.so 2>/dev/null | sort); do
    one $(basename $lib .so | sed 's/libvsgpu_alt_//') $GRAFT_REPO_ROOT/$lib
done
one main_again ""
