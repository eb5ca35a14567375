#!/bin/bash
# gpurun --timeout 330 -- 'bash scripts/r02_chain.sh'
# (1) the closed-loop microbenchmark (scripts/microbench/randmem chain): what the chip sustains when the requests of an expansion are
#     issued as a scan issues them (row -> buckets -> code rows -> heap work -> next row), 1 / 2 / 4 independent chains per wave
# (2) 10M A/B at the headline operating point's shape (L = 3, rescore 196): LDS heap top 511 vs 1023 entries (both 24 scans per CU
#     at this list size), per-scan table 64 KB vs 32 KB
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02
O=gpurun_out/r02
timeout 100 scripts/microbench/randmem chain 2>&1 | tee $O/microbench_chain.txt
timeout 200 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 3 \
    --configs VS_FAST=1,VS_F_HL=1023,VS_F_HL=511,VS_F_HL=511:VS_F_GCAP=8192,VS_F_GCAP=0 2>&1 | grep -E "index ready|search " | tee $O/ab_heap_top_10m.txt
