#!/bin/bash
# The first GPU call of the next session, in one go (about 25 GPU-minutes): what this round could only check on the wave64
# interpreter gets measured.
#   gpurun --timeout 2400 -- 'bash scripts/next_gpu_session.sh'
# 1. the whole GPU tier (incl. k_search_mx and the page path, new since the last hardware run)
# 2. k_search_fast vs k_search_mx at 1M (forced table-less), 10M, with identical operating points
# 3. SQ issue / stall counters of both kernels at 10M (is the search issue bound?  DESIGN.md section 11)
# 4. the default bench (50M; canary + A/B decide the kernel) with its rocprofv3 summary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/s_tests.txt
timeout 400 python scripts/fuzz_emu.py --gpu --seconds 180 --seed 11 2>&1 | tail -8 | tee gpurun_out/s_fuzz.txt  # random cases on the device
for MX in 0 1; do
  VS_MX=$MX VS_F_LDS_MAX_INS=0 python bench.py --n 1000000 --fixed 100,50 --skip-cpu --scan-nq 0 2>gpurun_out/s_1m_mx$MX.err | tee gpurun_out/s_1m_mx$MX.json | cut -c1-400
  VS_MX=$MX python bench.py --n 10000000 --distance cosine --fixed 100,100 --skip-cpu --scan-nq 0 --graph-cache /tmp/vs_graph 2>gpurun_out/s_10m_mx$MX.err | tee gpurun_out/s_10m_mx$MX.json | cut -c1-400
done
# k_search_mx tuning variants at 10M: LDS heap top (255 / 511 / 1023 entries) x gather depth (8 / 16 rows per scan in flight)
for HL in 255 511 1023; do for GD in 2 4; do
  VS_MX=1 VS_F_HL=$HL VS_MX_GD=$GD python bench.py --n 10000000 --distance cosine --fixed 100,100 --skip-cpu --scan-nq 0 --graph-cache /tmp/vs_graph 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mx hl=$HL gd=$GD', j['value'], j['kernels']['search'])" | tee -a gpurun_out/s_mx_variants.txt
done; done
# ... and with one scan per row instead of the scan queue
VS_MX=1 VS_MX_PERSIST=0 python bench.py --n 10000000 --distance cosine --fixed 100,100 --skip-cpu --scan-nq 0 --graph-cache /tmp/vs_graph 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mx persist=0', j['value'], j['kernels']['search'])" | tee -a gpurun_out/s_mx_variants.txt
for MX in 0 1; do
  VS_MX=$MX bash scripts/pmc_issue.sh 10000000 131072 100 100 /tmp/vs_graph 2>&1 | tail -30 | tee gpurun_out/s_pmc_issue_mx$MX.txt
  for p in A B; do mv gpurun_out/pmc_issue_$p.txt gpurun_out/s_pmc_issue_${p}_mx$MX.txt 2>/dev/null; done
done
bash scripts/final_profile.sh 2>&1 | tail -40

# BASELINE configs[4] (20M x 1536, label-filtered): 123 GB of vectors + 3.8 GB codes; ~4 min of build
python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --skip-cpu --scan-nq 0 2>gpurun_out/s_cfg5.err | tee gpurun_out/s_cfg5.json | cut -c1-600
