#!/bin/bash
# gpurun --timeout 2400 -- 'bash scripts/r02_cfg.sh'   GPU test tier + BASELINE configs[2] (10M cosine) and configs[4] (20M x 1536, labels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/cfg
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/cfg/tests.txt
timeout 900 python bench.py --n 10000000 --distance cosine --steps 6 --warmup 2 2>gpurun_out/cfg/cfg3.err | tee gpurun_out/cfg/cfg3.json | cut -c1-300
tail -5 gpurun_out/cfg/cfg3.err
timeout 1500 python bench.py --n 20000000 --dim 1536 --distance cosine --labels 32 --steps 4 --warmup 1 2>gpurun_out/cfg/cfg5.err | tee gpurun_out/cfg/cfg5.json | cut -c1-300
tail -12 gpurun_out/cfg/cfg5.err
