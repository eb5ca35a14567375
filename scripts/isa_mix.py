#!/usr/bin/env python3
"""Static instruction mix of one kernel's gfx950 ISA, per loop (no GPU needed: hipcc cross-compiles).

  python scripts/isa_mix.py pgvectorscale_amd/csrc/vs_search_fast.hip k_search_fastILi3ELi0ELb0ELi1ELb0E

Prints the register counts and, for every backward branch (= loop), the number of VALU / SALU / LDS / VMEM instructions
in its body.  Used to budget the instruction-issue-bound search kernel: the wave-uniform bookkeeping of
one scan runs on the scalar unit, whose issue rate per SIMD equals the vector unit's."""
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "--cuda-device-only", "-S"]


def cls(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", f.name, src], stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    m = re.search(r"^(_Z\w*%s\w*):.*?^\s*s_endpgm" % re.escape(pat), text, re.S | re.M)
    if not m:
        sys.exit(f"no kernel matching {pat}")
    name = m.group(1)
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), text, re.S)
    if meta:
        for k in ("sgpr_count", "vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            mm = re.search(r"\.%s:\s+(\d+)" % k, text[text.index(".name:           " + name):][:1500])
            if mm:
                print(f"{k}: {mm.group(1)}")
    ins, labels = [], {}
    for line in m.group(0).split("\n"):
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            labels[lm.group(1)] = len(ins)
            continue
        t = line.strip()
        if t and not t.startswith((";", ".")) and re.match(r"^[a-z_0-9]+$", t.split()[0]):
            ins.append(t.split()[0])
    tot = {}
    for op in ins:
        tot[cls(op)] = tot.get(cls(op), 0) + 1
    print(f"{name}: {len(ins)} instructions {tot}")
    loops = set()
    body = m.group(0).split("\n")
    idx = 0
    for line in body:
        t = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", line) or not t or t.startswith((";", ".")):
            continue
        if not re.match(r"^[a-z_0-9]+$", t.split()[0]):
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                loops.add((labels[tgt], idx, tgt))
        idx += 1
    for a, b, t in sorted(loops, key=lambda x: (x[0], -x[1])):
        c = {}
        for op in ins[a:b + 1]:
            c[cls(op)] = c.get(cls(op), 0) + 1
        marks = [k for k in ("ds_bpermute_b32", "v_readlane_b32", "ds_cmpst_rtn_b32", "global_atomic_cmpswap", "v_bcnt_u32_b32")
                 if any(op.startswith(k) for op in ins[a:b + 1])]
        print(f"  loop {t} [{a}..{b}] {b - a + 1:5d} instr {c} {' '.join(marks)}")


if __name__ == "__main__":
    main()
