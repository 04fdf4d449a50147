#!/bin/bash
# A/B of experiment builds (csrc/Makefile `make var VAR=<name> ...` -> maf-yolo_amd/libmafyolo_<name>.so) on the GPU box: the parity tests named in $TESTS and
# a short per-op bench for the product library and every variant given.
#   gpurun --timeout 1200 -- 'bash tools/ab_lib.sh out_dir hip pk ps'
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
TESTS=${TESTS:-"tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fused_parity.py"}
KEXPR=${KEXPR:-"bottleneck or closing_conv"}
for v in "$@"; do
  export MAF_HIP_LIB=$PWD/maf-yolo_amd/libmafyolo_$v.so
  (timeout 600 python -m pytest $TESTS -m gpu -q -k "$KEXPR" 2>&1 | tail -4) > $OUT/tests_$v.log
  echo "== $v tests: $(tail -1 $OUT/tests_$v.log)"
  for rep in 1 2; do
    python bench.py --per-op --steps 60 --warmup 20 --no-cpu-baseline --no-train-leg --no-extra-legs ${BENCH_ARGS:-} > $OUT/bench_${v}_$rep.json 2> $OUT/per_op_${v}_$rep.txt
    python - $OUT/bench_${v}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.0f  ms/step %.4f  forward_only %s" % (d["value"], d["ms_per_step"], d.get("forward_only", {}).get("ms_per_step")))
PY
    grep -E "${ROWS:-bottleneck_kernel}" $OUT/per_op_${v}_$rep.txt | grep -v GROUP | awk '{printf "      %-36s %s %s %s %s %s\n", $1, $2, $3, $4, $5, $6}' | head -12
  done
done
