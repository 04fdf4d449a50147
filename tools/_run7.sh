python -m pytest tests/test_gpu_kernels.py -x -q -k "register_resident or lds_resident" 2>&1 | tail -12
python tools/tune_probe.py 2>&1 | grep -E "block|conv2 .*M=.*(96|192)->" > gpurun_out/tune_probe4.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --tune-file gpurun_out/r3_tune_b.json --per-op > gpurun_out/r3_b3.json 2> gpurun_out/r3_b3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_b3.json'))
print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['frac'], d['roofline']['kernel'])
PY
