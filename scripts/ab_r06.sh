#!/bin/bash
# Round-6 A/B on one box, one cached graph: this tree's libvsgpu.so and every pgvectorscale_amd/libvsgpu_alt_*.so timed on the same
# index (perf_search.py --lib), then SQ instruction counters (one --pmc pass each) for the libraries named in PMC_LIBS.
#   gpurun --timeout 1500 -- 'bash scripts/ab_r06.sh <tag> <n> <L> <rescore> [configs]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; N=${2:-10000000}; L=${3:-3}; S=${4:-195}; CFGS=${5:-VS_FAST=1}
NQ=${NQ:-262144}
O=gpurun_out/$TAG; mkdir -p $O
NOB='^HIP version|^ROCm version|^Hostname|^Librccl path|amdgpu.ids'
run() { timeout 900 python scripts/perf_search.py --n $N --nq $NQ --L $L --rescore $S --reps ${REPS:-4} --graph-cache /tmp/g "$@" 2>&1 | grep -Ev "$NOB" | grep -E "search |index ready|rror"; }
echo "# this tree ($CFGS)" | tee $O/ab.txt
run --configs "$CFGS" | tee -a $O/ab.txt
for lib in $(ls pgvectorscale_amd/libvsgpu_alt_*.so 2>/dev/null | sort); do
    echo "# $lib" | tee -a $O/ab.txt
    run --configs "${ALT_CFGS:-VS_FAST=1}" --lib $lib | tee -a $O/ab.txt
done
echo "# this tree again" | tee -a $O/ab.txt
run --configs VS_FAST=1 | tee -a $O/ab.txt
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES"
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
for lib in ${PMC_LIBS:-main}; do
    LIBARG=""; [ "$lib" != main ] && LIBARG="--lib $lib"
    for pass in B ${PMC_PASS_A:+A}; do
        ctrs=${!pass}
        rm -rf gpurun_out/pmc_tmp
        echo "## pmc pass $pass: $lib" | tee -a $O/pmc.txt
        rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python scripts/perf_search.py --n $N --nq $NQ --L $L --rescore $S --reps 2 --graph-cache /tmp/g --configs VS_FAST=1 $LIBARG > $O/pmc_$pass.log 2>&1 \
            || { echo "pass failed"; tail -3 $O/pmc_$pass.log; continue; }
        grep -E "search " $O/pmc_$pass.log | tee -a $O/pmc.txt
        python scripts/pmc_summary.py gpurun_out/pmc_tmp/p_counter_collection.csv | grep -A9 "k_search_fast" | tee -a $O/pmc.txt
    done
done
rm -rf gpurun_out/pmc_tmp
