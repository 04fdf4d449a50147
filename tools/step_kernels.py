#!/usr/bin/env python3
"""Per-STEP kernel table of the training step from a rocprofv3 `--kernel-trace --output-format csv` trace: the steps of the trace are cut at the
optimizer's fused-SGD launches, the last N whole steps (steady state: past the warm-up steps that time the conv variants) are averaged.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py --train --steps 6 --warmup 8 --no-cpu-baseline
    python tools/step_kernels.py OUT/.../t_kernel_trace.csv 4 [out.md]
"""
import csv
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from summarize_profile import demangle          # noqa: E402


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search(r"[Ff]used[_]?[Ss]gd|FusedSgd|sgd_update_kernel", r[2])]
    # one optimizer step may be several launches (parameter groups): merge marks that are close together
    cuts = []
    for i in marks:
        if not cuts or rows[i][0] - rows[cuts[-1]][0] > 5_000_000:
            cuts.append(i)
    if len(cuts) < n + 1:
        raise SystemExit("only %d optimizer steps in the trace" % len(cuts))
    # the trailing steps of `bench.py --train` are its by-kind timing pass (eager launches between event pairs, no step tape, no lanes): the steady-state steps
    # are the ones on the most queues — take the last n of those
    pairs = list(zip(cuts[:-1], cuts[1:]))
    nq = [len({r[3] for r in rows[a:b]}) for a, b in pairs]
    pairs = [pr for pr, q in zip(pairs, nq) if q == max(nq)][-n:]
    agg, span = {}, 0.0
    for a, b in pairs:
        span += (rows[b][0] - rows[a][0]) / 1e6
        for s, e, name, q in rows[a:b]:
            d = agg.setdefault(demangle(name), [0, 0.0])
            d[0] += 1; d[1] += (e - s) / 1e3
    lines = ["# kernels of one training step (average of the last %d steps of the trace; step = %.3f ms wall between optimizer launches)" % (n, span / n), "",
             "| kernel | launches / step | avg us | ms / step |", "|---|---|---|---|"]
    tot = 0.0
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| `%s` | %.1f | %.2f | %.3f |" % (k[:150], c / n, us / c, us / n / 1e3))
        tot += us / n / 1e3
    lines.insert(2, "Sum of kernel durations: %.3f ms / step over %.0f launches (both streams; the weight-gradient stream overlaps the main one)." % (tot, sum(c for c, _ in agg.values()) / n))
    lines.insert(3, "")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out[:6000])


if __name__ == "__main__":
    main()
