#!/bin/bash
# the default bench + HBM traffic at the operating point it finds (first half of scripts/r02_final.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
O=gpurun_out/r02
CACHE=/tmp/vs_graph
timeout 1500 python bench.py --graph-cache $CACHE 2> $O/bench_50m.err > $O/bench_50m.json
grep -E "recall sweep L=(3|5|10) |operating point|timed results" $O/bench_50m.err | tail -30; cut -c1-300 $O/bench_50m.json
LS=$(python -c "import json;j=json.loads(open('$O/bench_50m.json').read().strip().splitlines()[-1]);print(str(j['config']['search_list_size'])+','+str(j['config']['rescore']))")
L=${LS%,*}; S=${LS#*,}
echo "operating point $L / $S"
timeout 900 bash scripts/pmc_traffic.sh 50000000 131072 $L $S $CACHE 2>&1 | tail -24
cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m.json 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write ${CACHE}.*
