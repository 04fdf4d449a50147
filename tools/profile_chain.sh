set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1; mkdir -p $O
cd $R
timeout 600 python bench.py --per-op --tune-file $O/tune.json > $O/bench0.json 2> $O/per_op0.txt
cp $O/tune.json profiles/round1_tune.json
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inflight 1 --steps 20 --warmup 5 --no-cpu-baseline --tune-file profiles/round1_tune.json > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -o p -- python bench.py --inflight 1 --steps 3 --warmup 2 --no-cpu-baseline --tune-file profiles/round1_tune.json > $O/pmcf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -o p -- python bench.py --inflight 1 --steps 3 --warmup 2 --no-cpu-baseline --tune-file profiles/round1_tune.json > $O/pmcw.log 2>&1
python tools/pmc_traffic.py $O $O/pmc_traffic.json > $O/pmc.log 2>&1
cp $O/pmc_traffic.json profiles/round1_pmc_traffic.json
timeout 600 python bench.py --per-op --tune-file profiles/round1_tune.json > $O/bench.json 2> $O/per_op.txt
tail -1 $O/bench.json | cut -c1-1500
for s in s m; do timeout 300 python bench.py --scale $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done
timeout 300 python bench.py --latency --scale m --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
timeout 300 python bench.py --latency --scale n --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
ls $O/stats | head
