#!/usr/bin/env python3
"""Copy the summaries of one tools/profile_round.sh chain from gpurun_out/<dir>/ into profiles/<prefix>_* (tracked).

    python tools/collect_profiles.py gpurun_out/r2prof_b round2 [gpurun_out/pmc_a]
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prefix = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "profiles")


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, "%s_%s" % (prefix, b)))
        print("copied", b)


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


cp("train_timeline.md", "train_timeline.md"); cp("train_launches_isolated.md", "train_launches_isolated.md"); cp("first_kernels.md", "first_kernels.md"); cp("train_step_kernels.md", "train_step_kernels.md"); cp("train_n_rccl1.json", "train_n_rccl1.json"); cp("train_host_profile.txt", "train_host_profile.txt")
cp("bench.json", "bench.json"); cp("per_op.txt", "per_op.txt"); cp("tune.json", "tune.json"); cp("train_n.json", "train_n.json")
cp("train_s.json", "train_s.json"); cp("train_m.json", "train_m.json"); cp("latency_m.json", "latency_m.json"); cp("bench_s.json", "bench_s.json"); cp("bench_m.json", "bench_m.json"); cp("bench_inflight1.json", "bench_inflight1.json"); cp("pmc_calibration.json", "pmc_calibration.json"); cp("tune_s.json", "tune_s.json"); cp("tune_m.json", "tune_m.json")
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    d, f = os.path.split(st[0])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), d, f[:-len("_kernel_stats.csv")], os.path.join(P, prefix), os.path.join(src, "stats_bench.json")])
st = glob.glob(os.path.join(src, "train_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    d, f = os.path.split(st[0])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), d, f[:-len("_kernel_stats.csv")], os.path.join(P, prefix + "_train_n")])
    os.remove(os.path.join(P, prefix + "_train_n_kernel_stats.csv"))
if os.path.isdir(os.path.join(src, "pmc_FETCH_SIZE")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), src, os.path.join(P, prefix + "_pmc_traffic.json")], stdout=subprocess.DEVNULL)
    print("wrote", prefix + "_pmc_traffic.json")
if os.path.isdir(os.path.join(src, "trainpmc", "pmc_FETCH_SIZE")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(src, "trainpmc"), os.path.join(P, prefix + "_train_pmc_traffic.json")], stdout=subprocess.DEVNULL)
    print("wrote", prefix + "_train_pmc_traffic.json")
if len(sys.argv) > 3 and os.path.exists(os.path.join(sys.argv[3], "summary.txt")):
    shutil.copy(os.path.join(sys.argv[3], "summary.txt"), os.path.join(P, prefix + "_sq_counters.txt"))
    print("copied sq counters")
