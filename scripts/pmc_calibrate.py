#!/usr/bin/env python3
"""Counter bytes per request for each request shape of k_search_fast, from the runs of scripts/pmc_calibrate.sh
(scripts/microbench/pmccal.hip: known request counts over footprints far beyond the caches).  Output:
pmc_calibration_randmem.json — per kernel: requests, the bytes the requests ask for, FETCH_SIZE / WRITE_SIZE bytes, and the
counter bytes PER REQUEST.  scripts/pmc_traffic.py turns a search launch's counters into HBM bytes with these."""
import argparse
import csv
import glob
import json
import os
import re


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void\s+", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name).strip()
            out.setdefault(name, []).append(float(r["Counter_Value"]) * 1024.0)  # KiB
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    args = ap.parse_args()
    fetch = counters(os.path.join(args.dir, "fetch"), "FETCH_SIZE")
    write = counters(os.path.join(args.dir, "write"), "WRITE_SIZE")
    out = {"source": "scripts/microbench/pmccal.hip under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), one MI355X",
           "patterns": {}}
    for ln in open(os.path.join(args.dir, "plain.txt")):
        m = re.match(r"CAL (\S+(?: \S+)*?) requests=(\d+) known_bytes=(\d+) unit=(\S+) ms=([0-9.]+) GBps=(\d+)", ln)
        if not m:
            continue
        name, req, known, unit, ms = m.group(1), float(m.group(2)), float(m.group(3)), m.group(4), float(m.group(5))
        f = sum(fetch.get(name, [0.0])) / max(len(fetch.get(name, [0.0])), 1)
        w = sum(write.get(name, [0.0])) / max(len(write.get(name, [0.0])), 1)
        out["patterns"][name] = {"requests": req, "unit": unit, "requested_bytes": known, "requested_bytes_per_request": known / req,
                                 "ms": ms, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w,
                                 "fetch_bytes_per_request": round(f / req, 3), "write_bytes_per_request": round(w / req, 3),
                                 "fetch_over_requested": round(f / known, 4), "write_over_requested": round(w / known, 4)}
    dst = os.path.join(args.dir, "pmc_calibration_randmem.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
