#!/bin/bash
# round 3, GPU session 5: GPU tier on the tree with epoch tags + fitted tables; A/B of the fitted dedup table at 10M and 50M,
# launch sizes 131072 / 262144 / 393216
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s5
O=gpurun_out/s5
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_tests.txt
CF="VS_F_GCAP_FIT=0,VS_F_GCAP_FIT=1,VS_F_GCAP_FIT=0,VS_F_GCAP_FIT=1"
timeout 900 python scripts/perf_search.py --n 10000000 --nq 131072 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_fit_10m.txt
timeout 1500 python scripts/perf_search.py --n 50000000 --nq 131072 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_fit_50m.txt
timeout 900 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_fit_50m_nq262144.txt
timeout 900 python scripts/perf_search.py --n 50000000 --nq 393216 --L 3 --rescore 196 --reps 2 --configs "VS_F_GCAP_FIT=1" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_fit_50m_nq393216.txt
rm -f /tmp/g.*
