#!/bin/bash
# gpurun --timeout 600 -- 'bash scripts/microbench/run.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/mb
B=scripts/microbench/randmem
$B | tee gpurun_out/mb/randmem.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/mb/fetch -o p -- $B > gpurun_out/mb/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/mb/write -o p -- $B > gpurun_out/mb/write.log 2>&1
python - <<'PY'
import csv, glob
for tag in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/mb/{tag}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        print(tag, f, len(rows))
        for r in rows:
            if "k_mix" in r.get("Kernel_Name", ""):
                print(tag, r.get("Dispatch_Id"), r.get("Kernel_Name")[:40], r.get("Counter_Name"), r.get("Counter_Value"))
PY
