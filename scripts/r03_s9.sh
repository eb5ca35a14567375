#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s9
O=gpurun_out/s9
run() { tag=$1; shift; echo "== $tag" | tee -a $O/repro3.txt; timeout 300 env "$@" python scripts/fuzz_emu.py --gpu --only 777000331 --repeat 8 2>&1 | grep -E "FAIL|ERROR|cases," | cut -c1-200 | sort | uniq -c | tee -a $O/repro3.txt; }
run "zero fill by kernel (new default)" VS_X=0
run "hipMemsetAsync (the old way)" VS_F_EPOCH_DBG=1
timeout 300 python scripts/fuzz_emu.py --gpu --seconds 150 --seed 779 2>&1 | tail -2 | tee -a $O/repro3.txt
