#!/bin/bash
# round 3, GPU session 8: the one device-fuzz failure of the evidence session (case 777000331, regime VS_F_HL=63 + table-less,
# "stream differs"; passes on the interpreter with every memory filler): how often, and since which change?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s8
O=gpurun_out/s8
run() { tag=$1; shift; echo "== $tag" | tee -a $O/repro.txt; timeout 300 env "$@" python scripts/fuzz_emu.py --gpu --only 777000331 --repeat 12 $LIBARG 2>&1 | grep -E "FAIL|ERROR|cases," | sort | uniq -c | tee -a $O/repro.txt; }
LIBARG=""
run "this tree" VS_X=0
run "this tree, VS_F_EPOCH=0" VS_F_EPOCH=0
run "this tree, VS_F_GCAP_FIT=0" VS_F_GCAP_FIT=0
run "this tree, VS_F_EPOCH=0 VS_F_GCAP_FIT=0" VS_F_EPOCH=0 VS_F_GCAP_FIT=0
LIBARG="--lib pgvectorscale_amd/libvsgpu_alt_6d5053b.so"
run "6d5053b (merged push-loop + overlap, before epoch / fitted tables)" VS_LIB_TOLERANT=1
LIBARG="--lib pgvectorscale_amd/libvsgpu_alt_59e24fe.so"
run "59e24fe (end of round 2)" VS_LIB_TOLERANT=1
LIBARG=""
# and the neighbourhood of the failing seed on this tree, search cases only
timeout 400 python scripts/fuzz_emu.py --gpu --seconds 240 --seed 778 --kind search 2>&1 | tail -3 | tee $O/fuzz_search.txt
