#!/bin/bash
# round 3, GPU session 4: A/B of the epoch-tagged dedup table (and table size / occupancy / launch size with it) at 10M and 50M
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s4
O=gpurun_out/s4
CF="VS_F_EPOCH=0:VS_F_GCAP=0:VS_F_MINW=6,VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_EPOCH=1:VS_F_GCAP=8192:VS_F_MINW=6,VS_F_EPOCH=1:VS_F_GCAP=32768:VS_F_MINW=6,VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=7,VS_F_EPOCH=0:VS_F_GCAP=0:VS_F_MINW=6,VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=6"
timeout 900 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_epoch_10m.txt
timeout 1500 python scripts/perf_search.py --n 50000000 --nq 131072 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_epoch_50m.txt
timeout 900 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 2 --configs "VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=6" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_nq262144_50m.txt
rm -f /tmp/g.*
