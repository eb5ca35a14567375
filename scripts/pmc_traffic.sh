#!/bin/bash
# HBM traffic of the search kernels from the TCC counters, collected as MI355X_MICROARCH.md prescribes: one rocprofv3
# --pmc pass per counter group (FETCH_SIZE / WRITE_SIZE), kernel-trace only.  The flat scan (k_scan_topk), whose byte
# count is known exactly and which uses the same 16-B-per-lane loads, runs in the same process to calibrate FETCH_SIZE
# (gfx950 tallies 128-B requests at 64 B).  usage: scripts/pmc_traffic.sh <n> <nq> <L> <rescore> [graph-cache-prefix] [variant]
# (variant: a name vs_index_autotune reports, e.g. the one a bench line carries in roofline.variant; default = the library default)
set -e
N=${1:-1000000}; NQ=${2:-131072}; L=${3:-100}; S=${4:-50}; CACHE=${5:-}; VARIANT=${6:-default}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python scripts/perf_search.py --n $N --nq $NQ --L $L --rescore $S --configs VS_FAST=1:VARIANT=$VARIANT --reps 2 --scan 64"
if [ -n "$CACHE" ]; then CMD="$CMD --graph-cache $CACHE"; fi
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $CMD > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $CMD > gpurun_out/pmc_write.log 2>&1
grep -E "QPS|scan:" gpurun_out/pmc_fetch.log
python scripts/pmc_traffic.py --n $N --nq $NQ --L $L --rescore $S --variant $VARIANT
