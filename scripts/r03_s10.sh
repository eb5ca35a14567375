#!/bin/bash
# the shipped tree against the library of the last evidence session (3740f30: the same defaults + a diagnostics branch that spilled
# three registers in the headline kernel) on one 10M graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s10
O=gpurun_out/s10/ab_spill_10m.txt
run() { timeout 200 python scripts/perf_search.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 3 --configs VS_FAST=1 --graph-cache /tmp/g "$@" 2>&1 | grep -E "search "; }
echo "# this tree" | tee $O; run | tee -a $O
echo "# 3740f30 (3 spilled registers)" | tee -a $O; VS_LIB_TOLERANT=1 run --lib pgvectorscale_amd/libvsgpu_alt_1_3740f30.so | tee -a $O
echo "# this tree again" | tee -a $O; run | tee -a $O
