#!/bin/bash
# gpurun --timeout 1500 -- 'bash scripts/r02_pmc.sh [n] [L] [S]'   counter passes over both search kernels (separate rocprofv3 --pmc runs)
N=${1:-10000000}; L=${2:-100}; S=${3:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
CMD="python scripts/perf_search.py --n $N --nq 131072 --L $L --rescore $S --reps 2 --configs VS_MX=0,VS_MX=1 --graph-cache /tmp/vsg"
$CMD 2>&1 | tail -3 | tee gpurun_out/pmc/plain.txt
declare -A P
P[A]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P[B]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES"
P[C]="FETCH_SIZE GRBM_GUI_ACTIVE"
P[D]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P[E]="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
P[F]="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum"
P[G]="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
P[H]="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for pass in A B C D E F G H; do
    rm -rf gpurun_out/pmc/$pass
    rocprofv3 --pmc ${P[$pass]} --kernel-trace --output-format csv -d gpurun_out/pmc/$pass -o p -- $CMD > gpurun_out/pmc/$pass.log 2>&1 \
        || { echo "pass $pass failed"; tail -3 gpurun_out/pmc/$pass.log; continue; }
    python scripts/pmc_summary.py gpurun_out/pmc/$pass/p_counter_collection.csv | tee gpurun_out/pmc/$pass.txt
    rm -rf gpurun_out/pmc/$pass
done
