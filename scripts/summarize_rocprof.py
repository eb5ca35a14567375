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


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f},"
                    f"{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{float(r['Percentage']):.3f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
