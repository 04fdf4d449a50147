#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts (VERDICT r2 task 7; MI355X_MICROARCH.md §HBM: FETCH_SIZE
reports half the bytes of a 16-byte-per-lane streaming read, "other access widths and WRITE_SIZE are uncalibrated").

  run     (under rocprofv3, one counter per pass — the TCC block cannot hold both):
          rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/cal/pmc_FETCH_SIZE -o p -- python tools/pmc_calibrate.py run
          rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/cal/pmc_WRITE_SIZE -o p -- python tools/pmc_calibrate.py run
  reduce  python tools/pmc_calibrate.py reduce gpurun_out/cal profiles/round3_pmc_calibration.json

`run` streams a 512 MiB buffer (twice the 256 MiB Infinity Cache: nothing is served on-die) through csrc/probe.hip:pmc_cal_kernel<kind> — 16- / 8- /
4-byte lane copies, a stride-2 read of 16-byte chunks, 16-byte reads with 2-byte writes, half-line (64-byte) segments — three launches each.
`reduce` divides what the counters report (KB) by what the kernels moved: factor = true bytes / reported bytes per access shape."""
import collections
import csv
import ctypes as C
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BYTES = 512 << 20
KINDS = {0: ("copy, 16-byte lanes", 1.0, 1.0), 1: ("copy, 8-byte lanes", 1.0, 1.0), 2: ("copy, 4-byte lanes", 1.0, 1.0),
         3: ("every other 16-byte chunk read, 16-byte writes", 0.5, 0.5), 4: ("16-byte reads, 2-byte writes", 1.0, 0.125),
         5: ("half lines: 64 of every 128 bytes read (MFMA activation fragment shape), 16-byte writes", 0.5, 0.5)}


def run():
    import torch
    from maf_yolo_amd import lib
    lib.load()                                        # the product library (the probe library links it)
    L = __import__("ctypes").CDLL(__import__("os").path.join(__import__("os").path.dirname(lib.LIB_PATH), "libmafyolo_probe.so"))   # `make -C maf-yolo_amd/csrc probe` (tools/probe.hip: not in the product)
    L.maf_probe_pmc_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]
    src = torch.randint(0, 255, (BYTES,), dtype=torch.uint8, device="cuda:0")
    dst = torch.empty(BYTES, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for kind in KINDS:
        for _ in range(3):
            lib.check(L.maf_probe_pmc_copy(st, kind, src.data_ptr(), dst.data_ptr(), BYTES))
    torch.cuda.synchronize()


def reduce(root, out):
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"pmc_cal_kernelILi(\d)E", r["Kernel_Name"]) or re.search(r"pmc_cal_kernel<(\d)>", r["Kernel_Name"])
                if m and r["Counter_Name"] == counter:
                    agg[int(m.group(1))].append(float(r["Counter_Value"]))
        for kind, vals in agg.items():
            name, rf, wf = KINDS[kind]
            true = BYTES * (rf if counter == "FETCH_SIZE" else wf)
            rep = sum(vals) / len(vals) * 1024
            e = res.setdefault(str(kind), {"access": name})
            e[counter] = {"true_bytes": int(true), "reported_bytes": int(rep), "factor_true_over_reported": round(true / rep, 4), "launches": len(vals)}
    json.dump({"buffer_bytes": BYTES, "note": "factor = bytes actually moved / (counter [KB] * 1024); multiply a reported value by it", "kinds": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items()):
        print(k, v["access"], {c: v[c]["factor_true_over_reported"] for c in v if c != "access"})


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        reduce(sys.argv[2], sys.argv[3])
