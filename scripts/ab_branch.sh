#!/bin/bash
# Build the libvsgpu.so of another branch / commit next to this tree's, for a same-session A/B on the GPU box:
#   scripts/ab_branch.sh next/push-loop        (CPU: cross-compiles into pgvectorscale_amd/libvsgpu_alt.so, git-ignored)
#   gpurun --timeout 300 -- 'bash scripts/ab_libs_gpu.sh 10000000 3 196'
set -e
REF=${1:?branch or commit}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/vs_ab.XXXXXX)
git -C "$ROOT" archive "$REF" pgvectorscale_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/pgvectorscale_amd/csrc" -j8 -s
cp "$TMP/pgvectorscale_amd/libvsgpu.so" "$ROOT/pgvectorscale_amd/libvsgpu_alt.so"
rm -rf "$TMP"
echo "built $REF -> pgvectorscale_amd/libvsgpu_alt.so"
