"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, the TCC block cannot hold both)
into per-kernel HBM traffic per launch, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:

    traffic = FETCH_SIZE [KB] * 1024 * 2  +  WRITE_SIZE [KB] * 1024

(the x2 is the guide's gfx950 correction: FETCH_SIZE tallies 128-byte requests at 64 B for wide coalesced reads,
which is the access shape of every kernel here — 16-byte lanes; WRITE_SIZE is taken as reported).

    usage: python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json>

Collected with (on the GPU box, one pass per counter, no other trace domains):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o p -- \
        python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-autotune-cache
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle  # noqa: E402


def per_kernel(dirname, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + dirname)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = demangle(r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("void ", "")
            n = re.sub(r"\(.*$", "", n)
            agg[n][0] += float(r["Counter_Value"])
            agg[n][1] += 1
    return agg


def main(root, out):
    fetch = per_kernel(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    res = {}
    for n in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, fc = fetch[n]
        w, wc = write.get(n, [0.0, 1])
        fb, wb = f / fc * 1024 * 2, w / max(wc, 1) * 1024
        res[n] = dict(launches=fc, fetch_bytes_corrected=int(fb), write_bytes=int(wb), traffic_bytes=int(fb + wb))
    json.dump(res, open(out, "w"), indent=1)
    for n, v in list(res.items())[:40]:
        print("%-62s n=%4d  fetch %8.2f MB  write %8.2f MB" % (n[:62], v["launches"], v["fetch_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
