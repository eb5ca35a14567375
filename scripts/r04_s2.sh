#!/bin/bash
# round 4, GPU session 2: the persistent grid of k_search_fast (VS_F_PERSIST) on hardware — exact first (regimes + fuzz), then timed
# at 50M against the round-3 library on one cached graph, with and without the written-bucket bitmap, at three launch sizes, and
# the scan timeline (start / end of every scan) of a 131 072-scan launch with one workgroup per scan and with the persistent grid.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s2
O=gpurun_out/r04s2
VS_TEST_VIRGIN=1 timeout 400 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_zv_fuzz.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
timeout 100 python scripts/fuzz_emu.py --gpu --seconds 60 --seed 4051 --kind search 2>&1 | tail -3 | tee $O/fuzz_gpu_persist.txt
VS_F_VIRGIN=1 VS_F_LDS_MAX_INS=0 timeout 100 python scripts/fuzz_emu.py --gpu --seconds 60 --seed 4052 --kind search 2>&1 | tail -3 | tee $O/fuzz_gpu_persist_virgin.txt
B="VS_F_EPOCH=0:VS_F_GCAP=0:VS_F_MINW=6"
run() { timeout 900 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 3 --graph-cache /tmp/g "$@" 2>&1 | grep -E "search |index ready|^host "; }
echo "# round-3 library" | tee $O/ab_persist_50m.txt
run --lib pgvectorscale_amd/libvsgpu_alt_0_r03.so --configs "$B:VS_F_VIRGIN=0,$B:VS_F_VIRGIN=1,NQ=131072:$B:VS_F_VIRGIN=0" | tee -a $O/ab_persist_50m.txt
echo "# this tree" | tee -a $O/ab_persist_50m.txt
P0="$B:VS_F_PERSIST=0"; P1="$B:VS_F_PERSIST=1"
run --configs "$P0:VS_F_VIRGIN=0,$P1:VS_F_VIRGIN=0,$P1:VS_F_VIRGIN=1,$P0:VS_F_VIRGIN=1,$P1:VS_F_VIRGIN=1:VS_F_PERSIST_PCT=83,$P1:VS_F_VIRGIN=1:VS_F_PERSIST_PCT=67,VS_F_PERSIST_PCT=100:NQ=131072:$P0:VS_F_VIRGIN=0:VS_TIMELINE=$O/tl_p0.bin,NQ=131072:$P1:VS_F_VIRGIN=0:VS_TIMELINE=$O/tl_p1.bin,NQ=131072:$P1:VS_F_VIRGIN=1:VS_TIMELINE=$O/tl_p1v.bin,VS_TIMELINE=:NQ=131072:$P1:VS_F_VIRGIN=1,NQ=65536:$P1:VS_F_VIRGIN=1,NQ=32768:$P1:VS_F_VIRGIN=1,NQ=65536:$P0:VS_F_VIRGIN=0,NQ=262144:$P1:VS_F_VIRGIN=1" --host "VS_HOST_CHUNKS=1,VS_HOST_CHUNKS=2,VS_HOST_CHUNKS=4,VS_HOST_CHUNKS=8" | tee -a $O/ab_persist_50m.txt
python scripts/timeline_summary.py $O/tl_p0.bin $O/tl_p1.bin $O/tl_p1v.bin > $O/timeline_50m.txt 2>&1
rm -f $O/tl_*.bin /tmp/g.*
