#!/bin/bash
# On the GPU box: this tree's libvsgpu.so against every pgvectorscale_amd/libvsgpu_alt_*.so (scripts/ab_branch.sh), one index
# geometry, one session.   bash scripts/ab_libs_gpu.sh [n=10000000] [L=3] [rescore=196]
# Order: the parity tier with the LAST alternative library first (the branch tip is the unproven one, and the first python
# process on a fresh box pays the minute-long import of torch — never put a short timeout on it); then every library is timed on
# the same graph (built once, cached in /tmp).
N=${1:-10000000}; L=${2:-3}; S=${3:-196}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ab
O=gpurun_out/ab/ab_${N}_${L}_${S}.txt
LIBS=$(ls pgvectorscale_amd/libvsgpu_alt_*.so 2>/dev/null | sort -t_ -k3 -n)
TIP=$(echo "$LIBS" | tail -1)
if [ -n "$TIP" ]; then
    cp pgvectorscale_amd/libvsgpu.so /tmp/libvsgpu_main.so
    cp "$TIP" pgvectorscale_amd/libvsgpu.so
    timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/ab/gpu_tests_tip.txt
    cp /tmp/libvsgpu_main.so pgvectorscale_amd/libvsgpu.so
fi
CFGS=${AB_CONFIGS:-VS_FAST=1}
run() { timeout 600 python scripts/perf_search.py --n $N --nq 131072 --L $L --rescore $S --reps 3 --configs "$CFGS" --graph-cache /tmp/vs_ab_graph "$@" 2>&1 | grep -E "search "; }
echo "# this tree" | tee $O
run | tee -a $O
for lib in $LIBS; do
    echo "# $lib" | tee -a $O
    run --lib $lib | tee -a $O
done
if [ -n "$AB_TIP_CONFIGS" ] && [ -n "$TIP" ]; then
    echo "# $TIP with $AB_TIP_CONFIGS" | tee -a $O
    CFGS="$AB_TIP_CONFIGS" run --lib $TIP | tee -a $O
fi
CFGS=${AB_CONFIGS:-VS_FAST=1}
echo "# this tree again" | tee -a $O
run | tee -a $O
rm -f /tmp/vs_ab_graph.*
