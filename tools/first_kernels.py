#!/usr/bin/env python3
"""Why does the first kernel of a forward (stem2_kernel) take 80 us under HIP events and 100 us in rocprofv3's statistics?  (VERDICT r3 weak #10.)
From a `--kernel-trace` CSV of `bench.py --inflight 1`: for every stem2 launch its duration and the IDLE time of the device in front of it, the
durations binned by that idle time, and the first three kernels of a few forwards.

    python tools/first_kernels.py <kernel_trace.csv> [out.md]
"""
import csv
import re
import sys

import numpy as np


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    out = ["# The first kernels of a forward under rocprofv3 (`bench.py --inflight 1`)", ""]
    idx = [i for i, r in enumerate(rows) if "stem2_kernel" in r[2]]
    dur = np.array([(rows[i][1] - rows[i][0]) / 1e3 for i in idx])
    last_end = np.maximum.accumulate(np.array([r[1] for r in rows]))
    idle = np.array([max(0.0, (rows[i][0] - last_end[i - 1]) / 1e3) if i else 0.0 for i in idx])
    out.append("stem2_kernel: %d launches, duration min %.1f / median %.1f / mean %.1f / max %.1f us" % (len(dur), dur.min(), np.median(dur), dur.mean(), dur.max()))
    out += ["", "| device idle before the launch | launches | mean duration (us) | min | max |", "|---|---|---|---|---|"]
    for lo, hi in ((0, 5), (5, 50), (50, 200), (200, 1000), (1000, 1e12)):
        sel = (idle >= lo) & (idle < hi)
        if sel.any():
            out.append("| %g - %g us | %d | %.1f | %.1f | %.1f |" % (lo, min(hi, 1e6), int(sel.sum()), dur[sel].mean(), dur[sel].min(), dur[sel].max()))
    # what else runs WHILE the stem runs (kernels that start before it ends and end after it starts)
    import collections
    co = collections.Counter()
    alone = []
    for k, i in enumerate(idx):
        s0, e0 = rows[i][0], rows[i][1]
        others = [short(rows[j][2]) for j in range(max(0, i - 12), min(len(rows), i + 12)) if j != i and rows[j][0] < e0 and rows[j][1] > s0]
        for o in set(others):
            co[o] += 1
        if not others:
            alone.append(dur[k])
    out += ["", "Launches with no other kernel on the device at the same time: %d, mean %.1f us; with one: %d, mean %.1f us.  Kernels seen beside it: %s" %
            (len(alone), float(np.mean(alone)) if alone else 0.0, len(dur) - len(alone), float((dur.sum() - sum(alone)) / max(1, len(dur) - len(alone))),
             ", ".join("%s x%d" % kv for kv in co.most_common(6)))]
    out += ["", "First kernels of five forwards (name, start relative to the stem's start, duration):", ""]
    for i in idx[len(idx) // 2: len(idx) // 2 + 5]:
        t0 = rows[i][0]
        out.append("- " + "; ".join("%s +%.1f us, %.1f us" % (short(rows[j][2]), (rows[j][0] - t0) / 1e3, (rows[j][1] - rows[j][0]) / 1e3) for j in range(i, min(i + 3, len(rows))))
                   + " (idle before: %.0f us)" % idle[idx.index(i)])
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
