#!/bin/bash
# Build one libvsgpu per commit of another branch next to this tree's, for a same-session A/B on the GPU box:
#   scripts/ab_branch.sh next/push-loop        (CPU: cross-compiles pgvectorscale_amd/libvsgpu_alt_<k>_<commit>.so, git-ignored,
#                                               k = 1 for the oldest commit that is not on the current branch)
#   gpurun --timeout 600 -- 'bash scripts/ab_libs_gpu.sh 10000000 3 196'
set -e
REF=${1:?branch or commit}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -f "$ROOT"/pgvectorscale_amd/libvsgpu_alt_*.so
k=0
for c in $(git -C "$ROOT" rev-list --reverse HEAD.."$REF"); do
    if git -C "$ROOT" diff --quiet "$c^" "$c" -- pgvectorscale_amd/csrc include; then continue; fi  # (no source change)
    k=$((k + 1))
    TMP=$(mktemp -d /tmp/vs_ab.XXXXXX)
    git -C "$ROOT" archive "$c" pgvectorscale_amd/csrc include | tar -x -C "$TMP"
    make -C "$TMP/pgvectorscale_amd/csrc" -j8 -s
    cp "$TMP/pgvectorscale_amd/libvsgpu.so" "$ROOT/pgvectorscale_amd/libvsgpu_alt_${k}_$(git -C "$ROOT" rev-parse --short "$c").so"
    rm -rf "$TMP"
    echo "built $(git -C "$ROOT" log --format='%h %s' -1 "$c" | cut -c1-100)"
done
ls -la "$ROOT"/pgvectorscale_amd/libvsgpu_alt_*.so
