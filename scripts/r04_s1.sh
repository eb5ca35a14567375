#!/bin/bash
# round 4, GPU session 1: decide the dedup-table variant on hardware.  (a) the regimes / fuzz that were interpreter-only at the end of
# round 3 (written-bucket bitmap VS_F_VIRGIN=1, two-row gather VS_F_MINW=5, epoch tags VS_F_EPOCH=1 incl. case 777000331);
# (b) one A/B per variant on ONE cached graph at 10M and at 50M (3 repetitions each, default first AND last so that drift shows).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s1
O=gpurun_out/r04s1
VS_TEST_VIRGIN=1 timeout 300 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_zv_fuzz.py -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
timeout 150 python scripts/fuzz_variants.py --gpu --cases 20 --seed 9 2>&1 | tail -3 | tee $O/fuzz_variants_gpu.txt
VS_F_VIRGIN=1 VS_F_LDS_MAX_INS=0 timeout 130 python scripts/fuzz_emu.py --gpu --seconds 90 --seed 4041 2>&1 | tail -5 | tee $O/fuzz_gpu_virgin.txt
VS_F_EPOCH=1 timeout 60 python scripts/fuzz_emu.py --gpu --only 777000331 --repeat 3 2>&1 | tail -2 | tee $O/fuzz_gpu_epoch_case.txt
VS_F_EPOCH=1 timeout 130 python scripts/fuzz_emu.py --gpu --seconds 90 --seed 4042 2>&1 | tail -5 | tee $O/fuzz_gpu_epoch.txt
D="VS_F_EPOCH=0:VS_F_VIRGIN=0:VS_F_GCAP=0:VS_F_MINW=6"
CF="$D,VS_F_VIRGIN=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=1:VS_F_GCAP=16384:VS_F_MINW=6,VS_F_VIRGIN=0:VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=0:VS_F_EPOCH=0:VS_F_GCAP=0:VS_F_MINW=5,$D"
timeout 400 python scripts/perf_search.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_virgin_10m.txt
timeout 700 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_virgin_50m.txt
rm -f /tmp/g.*
