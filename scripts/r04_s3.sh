#!/bin/bash
# round 4, GPU session 3: (a) the GPU tier of the tree — the multi-GPU entry points on two contexts of one GPU and the real librccl
# with a world of one, the slot-bitmap regimes, the re-cut pop rounds; (b) device fuzz with the slot bitmap; (c) on ONE cached 50M
# graph: the round-3 library against this tree with no bitmap / bucket bitmap / slot bitmap (fitted and sparser tables), persistent
# grid on; (d) the same A/B at 10M (where the bucket bitmap bought nothing).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s3
O=gpurun_out/r04s3
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.txt
VS_F_VIRGIN=2 VS_F_LDS_MAX_INS=0 timeout 120 python scripts/fuzz_emu.py --gpu --seconds 75 --seed 4061 --kind search 2>&1 | tail -2 | tee $O/fuzz_gpu_slotmap.txt
timeout 200 python scripts/fuzz_variants.py --gpu --cases 16 --seed 11 2>&1 | tail -2 | tee $O/fuzz_variants_gpu.txt
B="VS_F_EPOCH=0:VS_F_MINW=6"
run() { N=$1; shift; timeout 900 python scripts/perf_search.py --n $N --nq 262144 --L 3 --rescore 196 --reps 3 --graph-cache /tmp/g "$@" 2>&1 | grep -E "search |index ready|^host "; }
CF="$B:VS_F_GCAP=0:VS_F_VIRGIN=1,$B:VS_F_GCAP=0:VS_F_VIRGIN=2,$B:VS_F_GCAP=0:VS_F_VIRGIN=0,$B:VS_F_GCAP=16384:VS_F_VIRGIN=2,$B:VS_F_GCAP=20480:VS_F_VIRGIN=2,$B:VS_F_GCAP=32768:VS_F_VIRGIN=2:VS_F_SLOTMAP_FORCE=1,VS_F_SLOTMAP_FORCE=0:$B:VS_F_GCAP=0:VS_F_VIRGIN=2,NQ=131072:$B:VS_F_GCAP=0:VS_F_VIRGIN=2,NQ=262144:$B:VS_F_GCAP=0:VS_F_VIRGIN=1"
for N in 50000000 10000000; do
    echo "# round-3 library, $N" | tee -a $O/ab_slotmap_$N.txt
    run $N --lib pgvectorscale_amd/libvsgpu_alt_0_r03.so --configs "$B:VS_F_GCAP=0:VS_F_VIRGIN=0,$B:VS_F_GCAP=0:VS_F_VIRGIN=1" | tee -a $O/ab_slotmap_$N.txt
    echo "# this tree, $N" | tee -a $O/ab_slotmap_$N.txt
    run $N --configs "$CF" --host "VS_F_VIRGIN=2:VS_HOST_CHUNKS=4,VS_HOST_CHUNKS=3,VS_HOST_CHUNKS=6" | tee -a $O/ab_slotmap_$N.txt
    rm -f /tmp/g.*
done
