#!/usr/bin/env python3
"""Reads a VS_TIMELINE dump of libvsgpu (start / end of every scan of one k_search_fast launch, 100 MHz ticks) and prints what the
launch looked like from the inside: its span, how many scans were in flight over time, how long a scan lived depending on when it
started.  Diagnostics for the launch-tail question.

  VS_TIMELINE=/tmp/tl.bin python scripts/perf_search.py ... ; python scripts/timeline_summary.py /tmp/tl.bin
"""
import sys

import numpy as np


def main():
    for path in sys.argv[1:]:
        t = np.fromfile(path, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
        ok = (t[:, 0] > 0) & (t[:, 1] >= t[:, 0])
        t = t[ok]
        if not len(t):
            print(f"{path}: no scans recorded")
            continue
        t0, t1 = t[:, 0].min(), t[:, 1].max()
        span = (t1 - t0) / 100.0  # microseconds
        life = (t[:, 1] - t[:, 0]) / 100.0
        print(f"{path}: {len(t)} scans, launch span {span / 1e3:.3f} ms, scan life mean {life.mean():.0f} us  p50 {np.median(life):.0f}  "
              f"p99 {np.percentile(life, 99):.0f}  max {life.max():.0f}; mean in flight {life.sum() / span:.0f}")
        nb = 40
        edges = np.linspace(t0, t1, nb + 1)
        # scans in flight at the middle of every bucket, scans started / finished per bucket, mean life of the scans started in it
        starts = np.sort(t[:, 0])
        ends = np.sort(t[:, 1])
        mids = (edges[:-1] + edges[1:]) / 2
        inflight = np.searchsorted(starts, mids, side="right") - np.searchsorted(ends, mids, side="right")
        b = np.clip(np.searchsorted(edges, t[:, 0], side="right") - 1, 0, nb - 1)
        fin = np.clip(np.searchsorted(edges, t[:, 1], side="right") - 1, 0, nb - 1)
        print("  bucket_ms  in_flight  started  finished  mean_life_us_of_started")
        for i in range(nb):
            m = b == i
            print(f"  {(mids[i] - t0) / 1e5:8.2f}  {inflight[i]:9d}  {m.sum():7d}  {(fin == i).sum():8d}  {life[m].mean() if m.any() else 0:10.0f}")


if __name__ == "__main__":
    main()
