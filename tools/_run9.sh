python -m pytest tests/test_gpu_kernels.py -x -q -k "register_resident or stream or conv1x1" 2>&1 | tail -8
python tools/tune_probe.py 2>&1 | grep -E "block|conv1 |conv2 .*M=|stem|cv" | cut -c1-330 > gpurun_out/tune_probe7.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --tune-file gpurun_out/r3_tune_e.json --per-op > gpurun_out/r3_b6.json 2> gpurun_out/r3_b6.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_b6.json'))
print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['frac'], d['roofline']['kernel'])
PY
