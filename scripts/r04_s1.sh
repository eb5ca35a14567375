#!/bin/bash
# round 4, GPU session 1 (prepared in round 3, which had no GPU minutes left to run it): the written-bucket bitmap of the table-less
# dedup table (VS_F_VIRGIN=1: no clears, no reads of buckets the scan has not written) — exact on the device
# first (the regimes that use it + the differential fuzzer), then timed against the shipped default at 10M and 50M, alone and with
# sparser tables (more never-written buckets per probe, more lines touched), the two-row gather variant (VS_F_MINW=5, §11b.17) and the
# seven waves per SIMD (the software-pipelined visits of §11b.18 were measured three times slower at the end of round 3 and deleted)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s1
O=gpurun_out/r04s1
VS_TEST_VIRGIN=1 timeout 300 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_zv_fuzz.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
timeout 200 python scripts/fuzz_variants.py --gpu --cases 25 --seed 9 2>&1 | tail -3 | tee $O/fuzz_variants_gpu.txt
VS_F_VIRGIN=1 VS_F_LDS_MAX_INS=0 timeout 200 python scripts/fuzz_emu.py --gpu --seconds 150 --seed 4041 2>&1 | tail -5 | tee $O/fuzz_gpu_virgin.txt
# the epoch tags with the reallocation fix: the case that failed in round 3, then random cases
VS_F_EPOCH=1 timeout 120 python scripts/fuzz_emu.py --gpu --only 777000331 2>&1 | tail -2 | tee $O/fuzz_gpu_epoch_case.txt
VS_F_EPOCH=1 timeout 200 python scripts/fuzz_emu.py --gpu --seconds 120 --seed 4042 2>&1 | tail -5 | tee $O/fuzz_gpu_epoch.txt
CF="VS_F_EPOCH=0:VS_F_VIRGIN=0:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=1:VS_F_GCAP=16384:VS_F_MINW=6,VS_F_VIRGIN=1:VS_F_GCAP=24576:VS_F_MINW=6,VS_F_VIRGIN=0:VS_F_GCAP=0:VS_F_MINW=5,VS_F_VIRGIN=1:VS_F_GCAP=0:VS_F_MINW=5,VS_F_VIRGIN=0:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=0:VS_F_EPOCH=1:VS_F_GCAP=0:VS_F_MINW=6,VS_F_VIRGIN=0:VS_F_EPOCH=0:VS_F_MINW=7,VS_F_VIRGIN=0:VS_F_MINW=6"
timeout 900 python scripts/perf_search.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_virgin_10m.txt
timeout 1500 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs "$CF" --graph-cache /tmp/g 2>&1 | grep -E "search |index ready" | tee $O/ab_virgin_50m.txt
# the product's own A/B (vs_index_autotune): the probe alone, then the bench with the selection on, and the PMC
# traffic of whatever it chose (scripts/pmc_traffic.sh reads the variant's switches from the environment)
timeout 300 python -m pgvectorscale_amd.tune_probe 2>&1 | tail -1 | tee $O/tune_probe.json
timeout 900 python bench.py --steps 20 --warmup 5 --graph-cache /tmp/g 2>$O/bench_50m.log | tee $O/bench_50m_autotune.json
read V L S < <(python -c "import json;j=json.load(open('$O/bench_50m_autotune.json'));print(j['roofline']['variant'], j['config']['search_list_size'], j['config']['rescore'])")
timeout 900 bash scripts/pmc_traffic.sh 50000000 262144 $L $S /tmp/g $V 2>&1 | tail -40 | tee $O/pmc_traffic_50m_$V.txt
cp gpurun_out/pmc_search_traffic.json $O/pmc_search_traffic_50m_$V.json
# BASELINE configs[2] at its full size against the oracle (opt-in test: 31 GB of vectors go to the host)
VS_TEST_FULL_10M=1 timeout 600 python -m pytest tests/test_gpu_zx_full_size.py -q -m gpu -x 2>&1 | tail -3 | tee $O/full_size_tests.txt
# where is the boundary between the LDS-table regime and the table-less one?  (1M x 768: -37.7 % without the LDS table at L = 3 /
# rescore 53, profiles/r03/ab_autotune_1m.json; the default moved from 3072 to 1024 expected inserts on that one measurement.)
# Small scans, each under: LDS table + register visited list (the old default), LDS table + LDS-ring visited list, table-less
for LS in "3 53" "3 10" "10 0" "25 20"; do set -- $LS
  timeout 120 python scripts/perf_search.py --n 1000000 --nq 262144 --L $1 --rescore $2 --reps 3 --configs "VS_F_LDS_MAX_INS=100000,VS_F_LDS_MAX_INS=100000:VS_F_VR=0,VS_F_LDS_MAX_INS=0:VS_F_VR=0" 2>&1 | grep -E "search " | sed "s/^/L=$1 rescore=$2 /" | tee -a $O/ab_regime_boundary_1m.txt
done
rm -f /tmp/g.*
