#!/bin/bash
# Compile ONE instantiation of k_search_fast to gfx950 assembly (seconds instead of minutes for all 57) and print its
# code-object numbers + static instruction mix.  usage: scripts/isa_one.sh "3, 0, false, 6, false, false, 3" [out.s]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pgvectorscale_amd/csrc/vs_search_fast.hip
INST=${1:-"3, 0, false, 6, false, false, 3"}
OUT=${2:-/tmp/isa/one.s}
mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp /tmp/isa_one_XXXX.hip)
awk '/^#if VS_FAST_TU == 0/ {exit} {print}' "$SRC" > "$TMP"   # (everything above the host side of translation unit 0)
echo "template __global__ void k_search_fast<$INST>(FastArgs);" >> "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -I"$ROOT/pgvectorscale_amd/csrc" --cuda-device-only -S -o "$OUT" "$TMP" 2>/dev/null
rm -f "$TMP"
python3 - "$OUT" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
    m = re.search(r"\.%s:\s+(\d+)" % k, t)
    print(k, m.group(1) if m else "?")
body = t[t.index("k_search_fast"):]
ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and re.match(r"^\t[a-z_0-9]+(\s|$)", l) and not l.strip().startswith(".")]
def cnt(p): return sum(1 for o in ops if o.startswith(p))
print("instructions", len(ops), "salu", cnt("s_"), "valu", cnt("v_"), "lds", cnt("ds_"), "vmem", cnt("global_") + cnt("scratch_") + cnt("buffer_") + cnt("flat_"))
# loops of the first copy of the scan (the persistent grid's): the smallest one of >= 1000 instructions is `while (status == 0)`
lines = body.split("\n")
ins, labels = [], {}
for l in lines:
    if "s_endpgm" in l:
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if l.startswith("\t") and re.match(r"^\t[a-z_0-9]+(\s|$)", l) and not l.strip().startswith("."):
        ins.append(l.strip())
loops = []
for i, x in enumerate(ins):
    op = x.split()[0]
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = x.split()[-1]
        if tgt in labels and labels[tgt] <= i and 1300 <= i - labels[tgt] <= 2300:
            loops.append((i - labels[tgt] + 1, labels[tgt], i))
seen = set()
for n, a, b in sorted(loops):
    if a in seen:
        continue
    seen.add(a)
    L = [x.split()[0] for x in ins[a:b + 1]]
    c = lambda p: sum(1 for o in L if o.startswith(p))
    print("loop", n, "salu", c("s_"), "valu", c("v_"), "lds", c("ds_"), "vmem", c("global_"), "readlane", c("v_readlane"), "writelane", c("v_writelane"),
          "s_load", c("s_load"), "s_nop", c("s_nop"), "saveexec", sum(1 for o in L if "saveexec" in o), "branches", c("s_cbranch") + c("s_branch"))
print("v_writelane", cnt("v_writelane"), "v_readlane", cnt("v_readlane"), "s_nop", cnt("s_nop"), "branches", cnt("s_cbranch") + cnt("s_branch"),
      "saveexec", sum(1 for o in ops if "saveexec" in o), "scratch", cnt("scratch_"), "s_load", cnt("s_load"), "waitcnt", cnt("s_waitcnt"))
PY
