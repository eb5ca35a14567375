#!/bin/bash
# round 4, GPU session 5: (a) neighbor rows / neighbor masks / heap tids through non-temporal loads (default) against the normal
# cache policy (VS_F_FLAGS=2) on one cached 50M graph, with the code rows' own switch (VS_F_FLAGS=1) for scale; (b) the `mid` corpus
# at 50M with the operating-point grid extended to the GUC's upper range (rescore 600 / 800 / 1000); (c) the parity tier of the files
# the change touches.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s5
O=gpurun_out/r04s5
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|^RCCL version|amdgpu.ids'
timeout 600 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_parity.py tests/test_gpu_zw_cfg5.py tests/test_gpu_visibility.py tests/test_gpu_zv_fuzz.py -q -m gpu -x 2>&1 | grep -Ev "$NOBANNER" | tail -3 | tee $O/gpu_tests.txt
B="VS_F_MINW=6:VS_F_VIRGIN=2"
timeout 900 python scripts/perf_search.py --n 50000000 --nq 262144 --L 3 --rescore 196 --reps 3 --graph-cache /tmp/g \
    --configs "$B:VS_F_FLAGS=0,$B:VS_F_FLAGS=2,$B:VS_F_FLAGS=0,$B:VS_F_FLAGS=2,$B:VS_F_FLAGS=1,$B:VS_F_FLAGS=3,$B:VS_F_FLAGS=0" 2>&1 | grep -E "search |index ready" | tee $O/ab_nt_rows_50m.txt
rm -f /tmp/g.*
timeout 1800 python bench.py --corpus-kind mid --steps 10 --warmup 3 --graph-cache none > $O/bench_50m_mid.json 2> $O/bench_50m_mid.err
grep -E "operating point|recall@10 of the timed|WARNING" $O/bench_50m_mid.err | tail -5
python - <<'PY' | tee gpurun_out/r04s5/summary.txt
import json
j = json.loads(open("gpurun_out/r04s5/bench_50m_mid.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("50m mid QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"], "nq", j["config"]["queries_per_step_per_gpu"],
      "recall", j["recall_at_k"], j["recall_validate_lower95"], j["recall_heldout"], j.get("recall_heldout_lower95"), "met", j["recall_target_met"],
      "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
PY
