#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a tracked summary under profiles/.

    python tools/summarize_profile.py gpurun_out/prof5 r5 profiles/round1 [bench.json]

Copies <prefix>_kernel_stats.csv and writes <out>_kernel_stats.md (top kernels, per-forward time)."""
import csv
import json
import os
import shutil
import subprocess
import sys


_T = {"DF16_": "_Float16", "f": "float", "h": "unsigned char"}


def demangle(n):
    """c++filt does not know the _Float16 mangling (DF16_), so the few kernel templates of this library are decoded here."""
    import re
    m = re.match(r"_ZN12_GLOBAL__N_1\d+conv_mfma_kernelI(DF16_|f)Li(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])(?:ELb([01]))?EEEv", n)
    if m:
        tf = lambda g: "true" if m.group(g) == "1" else "false"
        return "conv_mfma_kernel<%s, %s, %s, %s, %s, %s, %s%s>" % (_T[m.group(1)], m.group(2), m.group(3), m.group(4), tf(5), tf(6), tf(7), ", " + tf(8) if m.group(8) else "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+dwconv_tile_kernelI(DF16_|f)Li(\d+)ELi(\d+)EEEv", n)
    if m:
        return "dwconv_tile_kernel<%s, %s, %s>" % (_T[m.group(1)], m.group(2), m.group(3))
    m = re.match(r"_ZN12_GLOBAL__N_1\d+stem2_kernelI(DF16_|f|h)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEEv", n)
    if m:
        return "stem2_kernel<%s, %s, %s>" % (m.group(2), m.group(3), m.group(5))
    m = re.match(r"_ZN12_GLOBAL__N_1\d+stem_kernelI(DF16_|f|h)(DF16_|f)Lb([01])EEEv", n)
    if m:
        return "stem_kernel<%s, %s, %s>" % (_T[m.group(1)], _T[m.group(2)], "true" if m.group(3) == "1" else "false")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(sppf_pool\w*_kernel)I(DF16_|f)", n)
    if m:
        return "%s<%s>" % (m.group(1), _T[m.group(2)])
    if n.startswith("_Z"):                                     # everything else: binutils' demangler (it does not know DF16_ = _Float16: spelled as `half`, then renamed)
        try:
            d = subprocess.run(["c++filt", n.replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip()
            if d and not d.startswith("_Z"):
                return d.replace("<half", "<_Float16").replace(" half", " _Float16")
        except Exception:
            pass
    return n


def main():
    d, prefix, out = sys.argv[1:4]
    src = os.path.join(d, prefix + "_kernel_stats.csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    shutil.copy(src, out + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(src)))
    if "train" in os.path.basename(out):                       # the commands of tools/profile_round.sh
        cmd = "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --train --scale n --batch 32 --steps 5 --warmup 2 --no-cpu-baseline` (the training step; the first steps time the conv variants of every layer shape, so call counts are not multiples of the step count)"
    else:
        cmd = "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --tune-file <chain>/tune.json --inflight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg` (one batch in flight: with two, launches of the two streams overlap and every kernel looks longer than it is alone)"
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % os.path.basename(out), "", "Command: " + cmd, ""]
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        try:
            j = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
            lines += ["bench line of the same command: value = %s %s, forward only = %s ms/step, roofline kernel = `%s` avg %s ms" %
                      (j["value"], j["unit"], j["forward_only"]["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["avg_launch_ms"]), ""]
        except Exception as e:
            lines += ["(bench line not parsed: %s)" % e, ""]
    lines += ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:45]:
        name = demangle(r["Name"]).replace("(anonymous namespace)::", "").replace("void ", "")
        if len(name) > 110:
            name = name[:107] + "..."
        lines.append("| `%s` | %s | %.2f | %.2f | %s |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    open(out + "_kernel_stats.md", "w").write("\n".join(lines) + "\n")
    print("wrote", out + "_kernel_stats.md")


if __name__ == "__main__":
    main()
