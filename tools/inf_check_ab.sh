#!/bin/bash
# A/B of the one-launch inf check (solver.GradScaler, MAF_INF_CHECK_NATIVE=0/1) on the GPU box: its parity test, the train-step tests that go through build_optimizer, then
# the n / s / m train legs three times each side.
#   gpurun --timeout 1500 -- 'bash tools/inf_check_ab.sh out_dir'
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_tape.py tests/test_gpu_exchange.py -m gpu -q -k "sgd or inf_check or optimizer or scaler or tape or exchange_train or two_ranks" 2>&1 | tail -6) > $OUT/tests.log
echo "== tests: $(tail -1 $OUT/tests.log)"
for rep in 1 2 3; do
  for v in 0 1; do
    MAF_INF_CHECK_NATIVE=$v python bench.py --train --scale n --batch 32 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/train_n_${v}_$rep.json 2> /dev/null
    python - $OUT/train_n_${v}_$rep.json $v n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("native=%s  %s  ms/step %.3f" % (sys.argv[2], sys.argv[3], d["ms_per_step"]))
PY
  done
done
for s in s m; do
  for v in 0 1; do
    b=32; [ $s = m ] && b=16
    MAF_INF_CHECK_NATIVE=$v python bench.py --train --scale $s --batch $b --steps 12 --warmup 4 --no-cpu-baseline > $OUT/train_${s}_$v.json 2> /dev/null
    python - $OUT/train_${s}_$v.json $v $s <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("native=%s  %s  ms/step %.3f" % (sys.argv[2], sys.argv[3], d["ms_per_step"]))
PY
  done
done
