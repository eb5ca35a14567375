#!/bin/bash
# Calibrates FETCH_SIZE / WRITE_SIZE on the request shapes of k_search_fast (scripts/microbench/pmccal.hip): one plain run (request
# counts, in dispatch order) and one rocprofv3 --pmc pass per counter, kernel-trace only (MI355X_MICROARCH.md: separate passes).
#   gpurun -- 'bash scripts/pmc_calibrate.sh'   ->  gpurun_out/pmccal/pmc_calibration_randmem.json (copy to profiles/rNN/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/pmccal
B=scripts/microbench/pmccal
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B scripts/microbench/pmccal.hip
O=gpurun_out/pmccal
$B > $O/plain.txt 2>&1
rm -rf $O/fetch $O/write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
python scripts/pmc_calibrate.py --dir $O
