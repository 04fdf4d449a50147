"""Per-kernel means of SQ counters from one rocprofv3 --pmc pass (counter_collection.csv).
usage: python tools/pmc_sq.py <dir> [kernel-substring]"""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", demangle(r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("void ", ""))
        a = agg[n][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, cs in agg.items():
    if flt not in n:
        continue
    wc = cs.get("SQ_WAVE_CYCLES", [0, 1]); wc = wc[0] / max(wc[1], 1)
    print(n[:70])
    for c, (v, k) in sorted(cs.items()):
        print("   %-26s %14.0f %s" % (c, v / k, ("%.3f of WAVE_CYCLES" % (v / k / wc)) if wc and c != "SQ_WAVE_CYCLES" else ""))
