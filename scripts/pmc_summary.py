#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv: mean counter value per dispatch, per (short) kernel name."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<(?:true|false)>)?)", name)
    return (m.group(1) if m else name[:60])[-60:]


def main(path, only=("k_search", "k_rerank", "k_scan")):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r.get("Kernel_Name", ""))
            if only and not any(o in k for o in only):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, ctrs in acc.items():
        n = max(len(v) for v in ctrs.values())
        print(f"{k}: {n} dispatches (mean per dispatch; last dispatches only = steady state)")
        for c, v in sorted(ctrs.items()):
            tail = v[-min(len(v), 4):]
            print(f"   {c:32s} {sum(tail) / len(tail):18.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
