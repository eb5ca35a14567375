#!/bin/bash
# round 4, GPU session 10 (diagnostic at 10M): is the process-to-process spread of k_search_fast a matter of where things land in
# device memory?  One build, then the same batch in separate processes that hold a pad of a given size before the index arrays
# (shifts everything) or right before the first search (shifts the workspace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s10
O=gpurun_out/r04s10/diag_placement_10m.txt
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
run() { timeout 200 python scripts/diag_state.py --n 10000000 --rescore 196 "$@" 2>&1 | grep -Ev "$NOBANNER" | grep -E "^pad|^\(a\)|^\(c\)" | cut -c1-118 | tee -a $O; }
run --phase build --idle 1
for i in 1 2 3; do run --phase load; done
for p in 64 256 1024 4096; do run --phase load --pad-mb $p; done
for p in 64 256 1024 4096; do run --phase load --ws-pad-mb $p; done
run --phase load
rm -f /tmp/diag_graph*
