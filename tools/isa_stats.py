#!/usr/bin/env python3
"""Per-kernel statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only): registers, spills, MFMA count and how many MFMAs sit
directly behind a full `s_waitcnt ...cnt(0)` (an un-pipelined operand load: the matrix core idles for the whole memory / LDS round trip).

  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imaf-yolo_amd/csrc -S --cuda-device-only -o /tmp/k.s maf-yolo_amd/csrc/<file>.hip
  python tools/isa_stats.py /tmp/k.s [filter]
"""
import re
import subprocess
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
text = open(path).read()
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
    body = m.group(2)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, body) else -1
    meta[m.group(1)] = (g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"))
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    if flt and flt not in dem:
        continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    nm = sum(1 for l in lines if l.startswith("v_mfma"))
    stalled_l = stalled_v = 0
    for i, l in enumerate(lines):
        if l.startswith("v_mfma"):
            for j in range(max(0, i - 6), i):
                if lines[j].startswith("v_mfma"):
                    break_at = j
            window = []
            j = i - 1
            while j >= 0 and i - j <= 8 and not lines[j].startswith("v_mfma"):
                window.append(lines[j]); j -= 1
            if any(re.search(r"lgkmcnt\(0\)", w) for w in window):
                stalled_l += 1
            if any(re.search(r"vmcnt\(0\)", w) for w in window):
                stalled_v += 1
    v, sp, s, lds = meta.get(name, (-1, -1, -1, -1))
    print("%-70s vgpr %3d spill %3d  mfma %4d  behind lgkmcnt(0) %4d  behind vmcnt(0) %4d  exp %3d  barriers %2d  instrs %5d" % (
        dem[:70], v, sp, nm, stalled_l, stalled_v, sum(1 for l in lines if l.startswith("v_exp_f32")), sum(1 for l in lines if l.startswith("s_barrier")), len(lines)))
