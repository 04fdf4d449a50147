#!/bin/bash
# A/B of the staged-store form of dwconv_p2 (tile_k + 128, csrc/dwconv_p2.hip) on the GPU box: its parity tests, then the per-op bench with the tuner's staged
# candidates off / on (MAF_DW_STAGE=0 / 1; every run times its tiles at start-up).
#   gpurun --timeout 1200 -- 'bash tools/dw_stage_ab.sh out_dir'
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dwconv" 2>&1 | tail -6) > $OUT/tests.log
echo "== tests: $(tail -1 $OUT/tests.log)"
for rep in 1 2; do
  for v in 0 1; do
    MAF_DW_STAGE=$v python bench.py --per-op --steps 60 --warmup 20 --no-cpu-baseline --no-train-leg --no-extra-legs --tune-file none > $OUT/bench_${v}_$rep.json 2> $OUT/per_op_${v}_$rep.txt
    python - $OUT/bench_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("stage=%s   value %.0f  ms/step %.4f  forward_only %s" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("forward_only", {}).get("ms_per_step")))
PY
    grep -E "dwconv_p2_kernel" $OUT/per_op_${v}_$rep.txt | grep -v GROUP | sed 's/^/      /'
    grep -E "GROUP dwconv_p2" $OUT/per_op_${v}_$rep.txt | sed -E 's/.*total +([0-9.]+) ms.*/\1/' | awk '{s+=$1} END {printf "      dwconv_p2 family total %.4f ms\n", s}'
  done
done
