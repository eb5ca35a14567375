#!/bin/bash
# round 4, GPU session 9: where the wave cycles of the shipped k_search_fast (slot bitmap, persistent grid) go and how many
# instructions it executes — two rocprofv3 --pmc passes (SQ counters, kernel-trace only) at 10M, L = 3 / rescore 196, 262 144 scans
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s9
timeout 900 bash scripts/pmc_issue.sh 10000000 262144 3 196 /tmp/g 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s9/pmc_issue_10m.txt
grep -E "search |index ready" gpurun_out/pmc_issue_A.log gpurun_out/pmc_issue_B.log | tee -a gpurun_out/r04s9/pmc_issue_10m.txt
rm -f /tmp/g.*
