#!/bin/bash
# One measurement chain on the GPU box (run through gpurun): bench line + per-op table, rocprofv3 kernel stats of the same command, the two PMC
# passes for HBM traffic, the training step.  Everything lands under gpurun_out/$1/; tools/collect_profiles.py copies the summaries to profiles/.
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r2prof'
set -u
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
TUNE=$OUT/tune.json
rm -f $TUNE
python bench.py --per-op --tune-file $TUNE --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/per_op.txt
echo "bench rc=$?"; tail -c 400 $OUT/bench.json | head -c 400; echo
python bench.py --tune-file $TUNE --inflight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg > $OUT/bench_inflight1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --tune-file $TUNE --inflight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg > $OUT/stats_bench.json 2> $OUT/stats.err
echo "rocprof stats rc=$?"
python tools/first_kernels.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/first_kernels.md > /dev/null 2>&1; echo "first kernels rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --tune-file $TUNE --inflight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-train-leg > /dev/null 2> $OUT/pmc_$c.err
  echo "pmc $c rc=$?"
done
python bench.py --train --scale n --batch 32 --steps 20 --warmup 8 > $OUT/train_n.json 2> $OUT/train_n.err; echo "train n rc=$?"
python bench.py --train --scale n --batch 32 --steps 20 --warmup 8 --rccl1 --no-cpu-baseline > $OUT/train_n_rccl1.json 2> $OUT/train_n_rccl1.err; echo "train n rccl1 rc=$?"
python tools/train_host_profile.py 5 > $OUT/train_host_profile.txt 2>&1; echo "train host profile rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o t -- python bench.py --train --scale n --batch 32 --steps 6 --warmup 8 --no-cpu-baseline > /dev/null 2> $OUT/train_stats.err
echo "train rocprof rc=$?"
python tools/step_kernels.py $(find $OUT/train_stats -name "*kernel_trace.csv" | head -1) 4 $OUT/train_step_kernels.md > /dev/null 2>&1; echo "train step table rc=$?"
python tools/step_timeline.py $(find $OUT/train_stats -name "*kernel_trace.csv" | head -1) 3 $OUT/train_timeline.md > /dev/null 2>&1; echo "train timeline rc=$?"
python tools/tape_times.py n 32 10 > $OUT/train_launches_isolated.md 2> /dev/null; echo "isolated launches rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/trainpmc/pmc_$c -o p -- python bench.py --train --scale n --batch 32 --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trainpmc_$c.err
  echo "train pmc $c rc=$?"
done
python bench.py --train --scale s --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_s.json 2>/dev/null; echo "train s rc=$?"
python bench.py --train --scale m --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_m.json 2>/dev/null; echo "train m rc=$?"
python bench.py --latency --scale m --tune-file $TUNE > $OUT/latency_m.json 2>/dev/null; echo "latency m rc=$?"
for s in s m; do python bench.py --scale $s --steps 30 --warmup 10 --no-cpu-baseline --no-train-leg --tune-file $OUT/tune_$s.json > $OUT/bench_$s.json 2>/dev/null; echo "bench $s rc=$?"; done
# PMC calibration (tools/pmc_calibrate.py): known byte counts per access shape, one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal/pmc_$c -o p -- python tools/pmc_calibrate.py run > /dev/null 2> $OUT/cal_$c.err
  echo "pmc calibration $c rc=$?"
done
python tools/pmc_calibrate.py reduce $OUT/cal $OUT/pmc_calibration.json > $OUT/pmc_calibration.txt 2>&1; cat $OUT/pmc_calibration.txt
# keep the merge under the 64 MiB limit: traces are big, the stats are what gets committed
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
