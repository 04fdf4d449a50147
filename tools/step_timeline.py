#!/usr/bin/env python3
"""Timeline of the training step from a rocprofv3 `--kernel-trace --output-format csv` trace: per HIP queue (the main stream, the weight-gradient stream, the
lanes) the busy time, first start and last end inside a step (steps cut at the fused-SGD launches as in tools/step_kernels.py), the time no queue runs
anything, and what the tail of a step looks like: how long after the main stream's last backward kernel the weight-gradient queue still runs before the
optimizer can start.  Says whether the step is bound by the sum of the work or by what one queue waits for.

    python tools/step_timeline.py OUT/.../t_kernel_trace.csv 4 [out.md]
"""
import csv
import re
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    rows = []
    for r in csv.DictReader(open(path)):
        q = r.get("Queue_Id", "") or r.get("Stream_Id", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search(r"[Ff]used[_]?[Ss]gd|FusedSgd|sgd_update_kernel", r[2])]
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
    lines = ["# timeline of the training step per HIP queue (last %d steps of the trace; times in ms from the step's optimizer launch)" % n, ""]
    for a, b in pairs:
        t0, t1 = rows[a][0], rows[b][0]
        seg = rows[a:b]
        per = {}
        for s, e, name, q in seg:
            per.setdefault(q, []).append((s, e, name))
        lines.append("## step of %.3f ms, %d launches; some queue busy %.3f ms, all idle %.3f ms" % ((t1 - t0) / 1e6, len(seg), union([(s, e) for s, e, _, _ in seg]) / 1e6,
                                                                                          ((t1 - t0) - union([(s, min(e, t1)) for s, e, _, _ in seg])) / 1e6))
        lines += ["", "| queue | launches | busy ms | sum of durations ms | first start | last end | largest gaps inside (ms @ ms, kernel after the gap) |", "|---|---|---|---|---|---|---|"]
        for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
            ks.sort()
            gaps = sorted(((ks[i + 1][0] - max(k[1] for k in ks[:i + 1][-8:]), ks[i + 1][0], ks[i + 1][2]) for i in range(len(ks) - 1)), reverse=True)[:3]
            gtxt = "; ".join("%.3f @ %.2f %s" % (g / 1e6, (at - t0) / 1e6, re.sub(r"\(anonymous namespace\)::|void ", "", nm)[:40]) for g, at, nm in gaps if g > 0)
            lines.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %s |" % (q, len(ks), union([(s, e) for s, e, _ in ks]) / 1e6, sum(e - s for s, e, _ in ks) / 1e6,
                                                                         (ks[0][0] - t0) / 1e6, (max(k[1] for k in ks) - t0) / 1e6, gtxt))
        lines.append("")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out[:8000])


if __name__ == "__main__":
    main()
