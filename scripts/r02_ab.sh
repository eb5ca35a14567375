#!/bin/bash
# gpurun --timeout 1500 -- 'bash scripts/r02_ab.sh'   kernel-level A/B of k_search_fast variants (one index build per corpus)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
CFG="VS_FAST=1,VS_F_MINW=6,VS_F_MINW=6:VS_F_HL=511,VS_F_MINW=6:VS_F_HL=255,VS_F_MINW=:VS_F_HL=255"
python scripts/perf_search.py --n 10000000 --nq 131072 --L 35 --rescore 106 --reps 3 --configs "$CFG" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab4_10m.txt
python scripts/perf_search.py --n 50000000 --nq 131072 --L 25 --rescore 189 --reps 3 --configs "$CFG" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab4_50m.txt
