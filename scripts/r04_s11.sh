#!/bin/bash
# round 4, GPU session 11 (experiment, library built from docs/experiments/ws_spread.patch, NOT the shipped one): if the 10 % between
# the states of session 7 is WHERE the 0.7 GB of per-workgroup regions land (all of it behind one slice of the memory system), then
# spreading the regions of the persistent grid over tens of GB should bring the slow state (right after the build) to the fast one.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s11
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
VS_LIB_TOLERANT=1 timeout 700 python scripts/diag_state.py --n 50000000 --phase build --idle 1 --lib pgvectorscale_amd/libvsgpu_exp_spread.so \
    --spread "3:0,1:16384,2:16384,3:16384,3:32768,3:4096,3:0" 2>&1 | grep -Ev "$NOBANNER" | cut -c1-120 | tee gpurun_out/r04s11/diag_spread_50m.txt
rm -f /tmp/diag_graph*
