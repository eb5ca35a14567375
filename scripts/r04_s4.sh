#!/bin/bash
# round 4, GPU session 4: the GPU tier of the tree whose default is the slot bitmap (incl. configs[2] at full size, now in the
# default tier), then one bench line of THIS tree for every configuration whose default changed — configs[1] (1M), configs[2]
# (10M cosine), configs[4] (20M x 1536, label-filtered) and the `mid` corpus at 50M — and two A/Bs on cached graphs: the
# label-filtered kernel's variants at configs[4], and the dedup table's load limit at 10M.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04s4
O=gpurun_out/r04s4
NOBANNER='^HIP version|^ROCm version|^Hostname|^Librccl path|amdgpu.ids'
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep -Ev "$NOBANNER" | tail -6 | tee $O/gpu_tests.txt
summ() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1], "QPS", j["value"], "ms/step", j["ms_per_step"], "L/S", j["config"]["search_list_size"], j["config"]["rescore"],
          "recall", j["recall_at_k"], j["recall_validate_lower95"], j["recall_heldout"], j.get("recall_heldout_lower95"), "met", j["recall_target_met"],
          "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "variant", r.get("variant"), "identical", (j.get("cpu_baseline") or {}).get("gpu_rows_identical"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
timeout 600 python bench.py --n 1000000 --steps 20 --warmup 5 --graph-cache none > $O/bench_cfg2.json 2> $O/bench_cfg2.err; summ $O/bench_cfg2.json | tee -a $O/summary.txt
timeout 900 python bench.py --n 10000000 --distance cosine --steps 10 --warmup 3 --graph-cache none > $O/bench_cfg3.json 2> $O/bench_cfg3.err; summ $O/bench_cfg3.json | tee -a $O/summary.txt
timeout 1500 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 8 --warmup 2 --graph-cache /tmp/g5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; summ $O/bench_cfg5.json | tee -a $O/summary.txt
LS=$(python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r04s4/bench_cfg5.json").read().strip().splitlines()[-1])
    print(f'{j["config"]["search_list_size"]},{j["config"]["rescore"]}')
except Exception:
    print("100,90")
PY
)
for V in "VS_F_VIRGIN=1" "VS_F_VIRGIN=0" "VS_F_MINW=5" "VS_F_MINW=5 VS_F_VIRGIN=1" "VS_F_NBRMASK=0"; do
    T=$(echo "$V" | tr ' =' '__')
    env $V timeout 600 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 6 --warmup 2 --graph-cache /tmp/g5 --fixed $LS --skip-cpu --scan-nq 0 --pcie-steps 0 > $O/bench_cfg5_$T.json 2> $O/bench_cfg5_$T.err
    summ $O/bench_cfg5_$T.json | tee -a $O/summary.txt
done
rm -f /tmp/g5.*
timeout 1800 python bench.py --corpus-kind mid --steps 10 --warmup 3 --graph-cache none > $O/bench_50m_mid.json 2> $O/bench_50m_mid.err; summ $O/bench_50m_mid.json | tee -a $O/summary.txt
B="VS_F_EPOCH=0:VS_F_MINW=6:VS_F_VIRGIN=2"
timeout 600 python scripts/perf_search.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 3 --graph-cache /tmp/g \
    --configs "$B:VS_F_GLOAD_PCT=75,$B:VS_F_GLOAD_PCT=85,$B:VS_F_GLOAD_PCT=90,$B:VS_F_GLOAD_PCT=65,$B:VS_F_GLOAD_PCT=75" 2>&1 | grep -E "search |index ready" | tee $O/ab_gload_10m.txt
rm -f /tmp/g.*
