#!/bin/bash
# round 4, GPU session 7 (diagnostic, no source change): why the search kernel is ~8 % slower in the process that BUILT the 50M graph
# than in a process that loaded it (final session: 167.2 vs 153.6 / 154.1 ms for identical work) — scripts/diag_state.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s7
O=gpurun_out/r04s7
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
timeout 900 python scripts/diag_state.py --n 50000000 --phase build --idle 60 2>&1 | grep -Ev "$NOBANNER" | tee $O/diag_state_50m.txt
timeout 300 python scripts/diag_state.py --n 50000000 --phase load 2>&1 | grep -Ev "$NOBANNER" | tee -a $O/diag_state_50m.txt
rm -f /tmp/diag_graph*
