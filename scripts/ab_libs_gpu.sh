#!/bin/bash
# On the GPU box: this tree's libvsgpu.so against pgvectorscale_amd/libvsgpu_alt.so (scripts/ab_branch.sh) on one index geometry.
#   bash scripts/ab_libs_gpu.sh [n=10000000] [L=3] [rescore=196]
# Order: the parity tier with the ALTERNATIVE library first (it is the unproven one, and the first python process on a fresh box
# pays the minute-long import of torch — never put a short timeout on it), then both timings.
N=${1:-10000000}; L=${2:-3}; S=${3:-196}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ab
O=gpurun_out/ab/ab_${N}_${L}_${S}.txt
cp pgvectorscale_amd/libvsgpu.so /tmp/libvsgpu_main.so
cp pgvectorscale_amd/libvsgpu_alt.so pgvectorscale_amd/libvsgpu.so
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/ab/gpu_tests_alt.txt
echo "# alternative library" > $O
timeout 600 python scripts/perf_search.py --n $N --nq 131072 --L $L --rescore $S --reps 3 --configs VS_FAST=1 2>&1 | grep -E "search " | tee -a $O
cp /tmp/libvsgpu_main.so pgvectorscale_amd/libvsgpu.so
echo "# this tree" >> $O
timeout 600 python scripts/perf_search.py --n $N --nq 131072 --L $L --rescore $S --reps 3 --configs VS_FAST=1 2>&1 | grep -E "search " | tee -a $O
