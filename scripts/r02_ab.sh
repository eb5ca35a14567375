#!/bin/bash
# gpurun --timeout 1500 -- 'bash scripts/r02_ab.sh'   kernel-level A/B of the search kernels (one index build per corpus)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export VS_DEBUG_STATUS=1
CFG="VS_MX=0,VS_F_FLAGS=2,VS_F_FLAGS=4,VS_F_FLAGS=6,VS_F_FLAGS=0:VS_F_GCAP=8192,VS_F_GCAP=8192:VS_F_FLAGS=2"
F='^\[VS_DEBUG_STATUS\] fast kernel: pool claims=131072 of 131072; status\[0\]=131072$'
python scripts/perf_search.py --n ${1:-10000000} --nq 131072 --L ${2:-100} --rescore ${3:-100} --reps 3 --configs "$CFG" 2>&1 | grep -v "$F" | tee gpurun_out/ab_sens.txt
