#!/bin/bash
# gpurun --timeout 300 -- 'bash scripts/r02_ab_minw8.sh'
# 10M x 768, L = 3 / rescore 196 (the shape of the headline operating point): waves per SIMD the table-less kernel is compiled for
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
timeout 60 python -m pytest tests/test_gpu_regimes.py -m gpu -q -x -k register_capped 2>&1 | tail -2
timeout 250 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 3 \
    --configs VS_FAST=1,VS_F_MINW=7,VS_F_MINW=8,VS_F_MINW=7:VS_F_HL=255,VS_F_MINW=6:VS_F_HL=511 2>&1 | grep -E "index ready|search " | tee gpurun_out/r02/ab_minw8_10m.txt
