#!/bin/bash
# gpurun --timeout 300 -- 'bash scripts/r02_ab_plain.sh'
# 10M x 768, L = 3 / rescore 196: the instantiation of k_search_fast without label keys / visibility mask (default for such batches)
# against the one that carries them (VS_F_FLAGS=8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
timeout 100 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_visibility.py -m gpu -q -x 2>&1 | tail -2
timeout 250 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 3 \
    --configs VS_FAST=1,VS_F_FLAGS=8,VS_F_FLAGS=0,VS_F_FLAGS=8:VS_F_MINW=7,VS_F_FLAGS=0:VS_F_MINW=6 2>&1 | grep -E "index ready|search " | tee gpurun_out/r02/ab_plain_variant_10m.txt
