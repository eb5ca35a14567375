#!/bin/bash
# gpurun --timeout 1500 -- 'bash scripts/r02_ab.sh'   experiment: does the order of the queries inside a batch matter?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for O in asis cluster cluster_xcd; do
python scripts/perf_search.py --n 10000000 --nq 131072 --L 35 --rescore 106 --reps 3 --configs "VS_FAST=1" --order $O --graph-cache /tmp/vsg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab3_10m.txt
done
