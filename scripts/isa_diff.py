#!/usr/bin/env python3
"""Are the kernels two source trees compile to the same machine code?  (No GPU needed: hipcc cross-compiles.)

  python scripts/isa_diff.py <old vs_search_fast.hip> <new vs_search_fast.hip> [name-map-suffix]
  python scripts/isa_diff.py <old> <new> --dropped-last-param     # the new tree REMOVED the last (bool) template parameter

Compiles both to gfx950 assembly, cuts every k_search_fast instantiation out (label .. s_endpgm), drops comments and renumbers the
basic-block labels in order of appearance, and reports per instantiation whether the instruction streams are identical — the
evidence that a commit which only ADDS template parameters / opt-in instantiations left the shipped instantiations' code (and with
it every hardware measurement of them) untouched.  An instantiation of the old tree k_search_fast<A...> is matched with
k_search_fast<A..., false, false> (the parameters the new tree appended, default off) when its exact name is gone.  With
--dropped-last-param the old k_search_fast<A..., false> is matched with k_search_fast<A...> and the old <A..., true> ones are
expected to be gone (how the deletion of the software-pipelined instantiations at the end of round 3 was checked: the 20
register-capped instantiations byte-identical, the 19 unconstrained ones the same length with a few register numbers changed)."""
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "--cuda-device-only", "-S", "-Wno-unused-function"]


def kernels(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", f.name, src], stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    out = {}
    for m in re.finditer(r"^(_Z\w*k_search_fast\w*):[^\n]*\n(.*?^\s*s_endpgm)", text, re.S | re.M):
        body = []
        labels = {}
        for ln in m.group(2).splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith("."):
                if not re.match(r"\.LBB\w+:", ln):
                    continue
            ln = re.sub(r"\.LBB\d+_\d+", lambda g: labels.setdefault(g.group(0), f"L{len(labels)}"), ln)
            body.append(ln)
        out[m.group(1)] = body
    return out


def main():
    old, new = kernels(sys.argv[1]), kernels(sys.argv[2])
    dropped = len(sys.argv) > 3 and sys.argv[3] == "--dropped-last-param"
    suffix = sys.argv[3] if len(sys.argv) > 3 and not dropped else "Lb0ELb0E"
    same = diff = missing = 0
    for name, body in sorted(old.items()):
        if dropped:
            if name.endswith("Lb1EEv8FastArgs"):
                print(f"gone     {name}" if name not in new else f"STILL THERE {name}")
                continue
            cand = name[:-len("Lb0EEv8FastArgs")] + "Ev8FastArgs" if name.endswith("Lb0EEv8FastArgs") else name
        else:
            cand = name if name in new else name.replace("EEv8FastArgs", "E" + suffix + "Ev8FastArgs")
        if cand not in new:
            print(f"MISSING  {name}")
            missing += 1
            continue
        nb = new[cand]
        if nb == body:
            same += 1
            print(f"same     {name}  ({len(body)} lines)")
        else:
            diff += 1
            first = next((i for i, (a, b) in enumerate(zip(body, nb)) if a != b), min(len(body), len(nb)))
            print(f"DIFFERS  {name}: {len(body)} -> {len(nb)} lines, first difference at line {first}: "
                  f"{body[first] if first < len(body) else '<end>'!r} vs {nb[first] if first < len(nb) else '<end>'!r}")
    print(f"{same} identical, {diff} different, {missing} missing; {len(new) - same - diff} instantiations only in the new tree")


if __name__ == "__main__":
    main()
