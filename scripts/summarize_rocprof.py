#!/usr/bin/env python3
"""Trim a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short, committable summary
(kernel names cut at the first '(' / '<' template argument list)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<(?:true|false)>)?)", name)
    s = m.group(1) if m else name[:60]
    return s[-70:]


def full_launches(trace_path, kernel="k_search_fast", min_ms=1.0):
    """k_search_fast is also launched on tiny recall samples and as the (mostly empty) second attempt of a step, which the stats
    row averages in: the timed launches are the dispatches within a factor of two of the longest one (with the persistent grid the
    LARGEST grid is the second attempt's — one workgroup per scan, nearly all of which return at once)"""
    rows = [r for r in csv.DictReader(open(trace_path)) if kernel in r["Kernel_Name"]]
    if not rows:
        return None
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6  # noqa: E731
    dmax = max(dur(r) for r in rows)
    full = [r for r in rows if dur(r) >= max(0.5 * dmax, min_ms)]
    if not full:
        return None
    d = [dur(r) for r in full]
    # the timed region of bench.py is its last `steps` full launches (the earlier ones: warm-up, validation and held-out batches)
    r0 = max(full, key=dur)
    return {"name": re.sub(r"^void\s+", "", r0["Kernel_Name"]), "grid": int(r0["Grid_Size_X"]), "calls": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
            "last20_avg_ms": sum(d[-20:]) / len(d[-20:]), "vgpr": r0["VGPR_Count"], "sgpr": r0["SGPR_Count"], "lds": r0["LDS_Block_Size"],
            "scratch": r0["Scratch_Size"]}


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    import os
    trace = src.replace("_kernel_stats.csv", "_kernel_trace.csv")
    fl = full_launches(trace) if os.path.exists(trace) else None
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        if fl:
            f.write(f"# k_search_fast, full launches only (grid {fl['grid']} threads = {fl['grid'] // 64} single-wave workgroups, within 2x of the longest; from the kernel "
                    f"trace of the same run): calls={fl['calls']} avg_ms={fl['avg_ms']:.3f} min_ms={fl['min_ms']:.3f} max_ms={fl['max_ms']:.3f} "
                    f"last_20_avg_ms={fl['last20_avg_ms']:.3f} "
                    f"(rocprofv3's launch-time fields: VGPRs={fl['vgpr']} SGPRs={fl['sgpr']} LDS={fl['lds']} B scratch={fl['scratch']} B — allocation "
                    f"granules, dynamic LDS not included; the code object's own numbers follow)\n")
            # what the hardware runs: the AMDGPU metadata notes of the instantiation the trace names
            try:
                sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                import codeobj_notes
                root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                want = re.sub(r"\s+", "", fl["name"].split("(")[0])
                # (k_search_fast is built as three translation units since round 6: look in each object)
                objs = [o for o in ("vs_search_fast_plain6.o", "vs_search_fast_keys6.o", "vs_search_fast.o")
                        if os.path.exists(os.path.join(root, "pgvectorscale_amd", "csrc", o))]
                found = False
                for o in objs:
                    for d in codeobj_notes.kernels(os.path.join(root, "pgvectorscale_amd", "csrc", o)):
                        if re.sub(r"\s+", "", d["demangled"].split("(")[0].replace("void ", "")) == want:
                            f.write(f"# code object (llvm-readelf --notes on csrc/{o}), {d['demangled'].split('(')[0]}: {codeobj_notes.line(d)}\n")
                            found = True
                            break
                    if found:
                        break
                if not found:
                    f.write(f"# code object: no instantiation named {want} in {', '.join('csrc/' + o for o in objs)}\n")
            except Exception as e:  # noqa: BLE001
                f.write(f"# code object notes unavailable: {e!r}\n")
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},"
                    f"{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{float(r['Percentage']):.3f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
