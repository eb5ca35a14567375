#!/bin/bash
# Is k_search_fast issue bound or latency bound?  Two rocprofv3 --pmc passes (8 SQ slots each, kernel-trace only, as
# MI355X_MICROARCH.md prescribes) over the same perf_search.py run:
#   pass A: where the wave cycles go     WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) +
#                                        ACTIVE_INST_ANY ~ WAVE_CYCLES; ACTIVE_INST_VALU / _SCA / _LDS / _VMEM split the last
#   pass B: how many instructions        INSTS_VALU / _SALU / _LDS / _VMEM_RD / _VMEM_WR / _SMEM, WAVES
# Reading: ACTIVE_INST_ANY / WAVE_CYCLES near the number of resident waves^-1 means the SIMDs are busy issuing (cut
# instructions, DESIGN.md section 11 / docs/LAB_NOTEBOOK.md section 11); a dominant WAIT_ANY means latency (raise occupancy / prefetch more).
# usage: scripts/pmc_issue.sh <n> <nq> <L> <rescore> [graph-cache-prefix]     (counter names: rocprofv3 -L)
N=${1:-1000000}; NQ=${2:-131072}; L=${3:-100}; S=${4:-50}; CACHE=${5:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python scripts/perf_search.py --n $N --nq $NQ --L $L --rescore $S --configs VS_FAST=1 --reps 2"
if [ -n "$CACHE" ]; then CMD="$CMD --graph-cache $CACHE"; fi
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES"
for pass in A B; do
    ctrs=${!pass}
    rm -rf gpurun_out/pmc_issue_$pass
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_issue_$pass -o p -- $CMD > gpurun_out/pmc_issue_$pass.log 2>&1 \
        || { echo "pass $pass failed (an unknown counter name? see gpurun_out/pmc_issue_$pass.log and rocprofv3 -L)"; tail -5 gpurun_out/pmc_issue_$pass.log; continue; }
    python scripts/pmc_summary.py gpurun_out/pmc_issue_$pass/p_counter_collection.csv | tee gpurun_out/pmc_issue_$pass.txt
done
