#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python scripts/perf_search.py --n 10000000 --nq 131072 --L 35 --rescore 106 --reps 3 --configs "VS_FAST=1,VS_F_MINW=1:VS_F_HL=1023,VS_F_MINW=:VS_F_HL=:VS_PHASE=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab5_10m.txt
