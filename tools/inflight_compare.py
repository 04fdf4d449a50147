"""Average kernel durations with one and with two batches in flight (two rocprofv3 --kernel-trace --stats runs):
python tools/inflight_compare.py <stats1.csv> <stats2.csv>   -> kernels sorted by the time they take per forward when overlapped"""
import csv, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[demangle(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, (n2, t2) in b.items():
    if k in a and "at::" not in k:
        n1, t1 = a[k]
        rows.append((t2 * n2, k, n1, t1, n2, t2))
rows.sort(reverse=True)
print("%-64s %6s %9s %9s %6s" % ("kernel", "calls", "alone us", "2-in-fl us", "x"))
for tot, k, n1, t1, n2, t2 in rows[:40]:
    print("%-64s %6d %9.1f %9.1f %6.2f" % (k[:64], n2, t1, t2, t2 / t1))
