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
    row averages in: the timed launches are the dispatches of the largest grid that ran longer than `min_ms`"""
    rows = [r for r in csv.DictReader(open(trace_path)) if kernel in r["Kernel_Name"]]
    if not rows:
        return None
    gmax = max(int(r["Grid_Size_X"]) for r in rows)
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if int(r["Grid_Size_X"]) == gmax]
    d = [x for x in d if x >= min_ms]
    if not d:
        return None
    d = [x for x in d if x >= 0.5 * max(d)]  # (a second attempt that had a scan or two to finish is not a full launch)
    r0 = max((r for r in rows if int(r["Grid_Size_X"]) == gmax), key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {"grid": gmax, "calls": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d), "vgpr": r0["VGPR_Count"],
            "sgpr": r0["SGPR_Count"], "lds": r0["LDS_Block_Size"], "scratch": r0["Scratch_Size"]}


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    import os
    trace = src.replace("_kernel_stats.csv", "_kernel_trace.csv")
    fl = full_launches(trace) if os.path.exists(trace) else None
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        if fl:
            f.write(f"# k_search_fast, full launches only (grid {fl['grid']} threads, > 1 ms; from the kernel trace of the same run): "
                    f"calls={fl['calls']} avg_ms={fl['avg_ms']:.3f} min_ms={fl['min_ms']:.3f} max_ms={fl['max_ms']:.3f} "
                    f"VGPRs={fl['vgpr']} SGPRs={fl['sgpr']} LDS={fl['lds']} B scratch={fl['scratch']} B\n")
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},"
                    f"{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{float(r['Percentage']):.3f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
