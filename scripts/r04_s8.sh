#!/bin/bash
# round 4, GPU session 8 (diagnostic at 10M, 3 minutes): does the state dependence of k_search_fast (50M: 168 ms in the process that
# built the graph, 152 ms in one that loaded it, 171 ms when the workspace is reserved with the index arrays) show at 10M too, where the
# next round could study it at 27 s per build?  build process / load process / load process with the workspace allocated first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s8
O=gpurun_out/r04s8
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
timeout 400 python scripts/diag_state.py --n 10000000 --rescore 196 --phase build --idle 5 2>&1 | grep -Ev "$NOBANNER" | cut -c1-130 | tee $O/diag_state_10m.txt
timeout 200 python scripts/diag_state.py --n 10000000 --rescore 196 --phase load 2>&1 | grep -Ev "$NOBANNER" | cut -c1-130 | tee -a $O/diag_state_10m.txt
timeout 200 python scripts/diag_state.py --n 10000000 --rescore 196 --phase load --early 2>&1 | grep -Ev "$NOBANNER" | cut -c1-130 | tee -a $O/diag_state_10m.txt
timeout 200 python scripts/diag_state.py --n 10000000 --rescore 196 --phase load 2>&1 | grep -Ev "$NOBANNER" | cut -c1-130 | tee -a $O/diag_state_10m.txt
rm -f /tmp/diag_graph*
