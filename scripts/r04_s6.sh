#!/bin/bash
# round 4, GPU session 6 (short): the GPU tier of the tree that sizes the visited list's LDS ring from the lists of earlier batches,
# and what that is worth where lists are long — the reference's default search_list_size of 100 on a label-filtered 5M x 1536 index
# (configs[4] in small) and on 10M x 768 cosine (configs[2]'s operating point): ring fitted (default) against the old 2 x (1.5 L + 32).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s6
O=gpurun_out/r04s6
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep -Ev "$NOBANNER" | tail -4 | tee $O/gpu_tests.txt
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"], "kernel ms", r["avg_kernel_ms"], "frac", r["frac"],
          "heldout", j["recall_heldout"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
C5="--n 5000000 --dim 1536 --distance cosine --labels 32 --steps 6 --warmup 2 --fixed 100,90 --skip-cpu --scan-nq 0 --pcie-steps 0 --graph-cache /tmp/g5"
timeout 600 python bench.py $C5 > $O/cfg5_5m_ring_fitted.json 2> $O/cfg5_5m_ring_fitted.err; line $O/cfg5_5m_ring_fitted.json | tee -a $O/summary.txt
VS_F_VCAP=384 timeout 600 python bench.py $C5 > $O/cfg5_5m_ring_384.json 2> $O/cfg5_5m_ring_384.err; line $O/cfg5_5m_ring_384.json | tee -a $O/summary.txt
rm -f /tmp/g5.*
C3="--n 10000000 --distance cosine --steps 6 --warmup 2 --fixed 100,50 --skip-cpu --scan-nq 0 --pcie-steps 0 --graph-cache /tmp/g3"
timeout 600 python bench.py $C3 > $O/cfg3_L100_ring_fitted.json 2> $O/cfg3_L100_ring_fitted.err; line $O/cfg3_L100_ring_fitted.json | tee -a $O/summary.txt
VS_F_VCAP=384 timeout 600 python bench.py $C3 > $O/cfg3_L100_ring_384.json 2> $O/cfg3_L100_ring_384.err; line $O/cfg3_L100_ring_384.json | tee -a $O/summary.txt
rm -f /tmp/g3.*
