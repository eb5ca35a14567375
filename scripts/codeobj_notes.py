#!/usr/bin/env python3
"""Code-object numbers of the gfx950 kernels inside a hipcc object file or libvsgpu.so: registers, spills, scratch, LDS, kernarg bytes
per kernel — from the AMDGPU metadata notes of the device ELF (llvm-readelf --notes), which is what the hardware runs.  rocprofv3's own
VGPR / LDS columns are launch-time fields (allocation granules, dynamic LDS left out) and disagreed with these in round 5.

  python scripts/codeobj_notes.py pgvectorscale_amd/csrc/vs_search_fast_plain6.o [name-filter]     (no GPU needed; k_search_fast lives in
  vs_search_fast.o, vs_search_fast_plain6.o and vs_search_fast_keys6.o since round 6)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_elf(path, tmp):
    fat = os.path.join(tmp, "fatbin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "copy.o")])
    out = os.path.join(tmp, "gfx950.co")
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--output={out}"], stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def kernels(path):
    """[{name, demangled, sgpr_count, sgpr_spill_count, vgpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size,
    kernarg_segment_size}]"""
    with tempfile.TemporaryDirectory() as tmp:
        co = device_elf(path, tmp)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        d = {}
        for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
                  "kernarg_segment_size"):
            m = re.search(r"\.%s:\s+(\d+)" % k, blk)
            d[k] = int(m.group(1)) if m else None
        m = re.search(r"\.name:\s+(\S+)", blk)
        d["name"] = m.group(1) if m else "?"
        out.append(d)
    for d, dm in zip(out, demangle([d["name"] for d in out])):
        d["demangled"] = dm
    return out


def line(d):
    return (f"SGPRs={d['sgpr_count']} (spilled {d['sgpr_spill_count']}) VGPRs={d['vgpr_count']} (spilled {d['vgpr_spill_count']}) "
            f"scratch={d['private_segment_fixed_size']} B static_LDS={d['group_segment_fixed_size']} B kernarg={d['kernarg_segment_size']} B")


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for d in kernels(path):
        if flt in d["demangled"]:
            print(f"{d['demangled']}: {line(d)}")


if __name__ == "__main__":
    main()
