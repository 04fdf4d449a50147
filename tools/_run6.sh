python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fused_parity.py tests/test_gpu_model.py -x -q 2>&1 | tail -15
python tools/tune_probe.py > gpurun_out/tune_probe3.txt 2>&1
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune-file gpurun_out/r3_tune_a.json --per-op > gpurun_out/r3_b2.json 2> gpurun_out/r3_b2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_b2.json'))
print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['frac'], d['roofline']['kernel'], d['train']['ms_per_step'] if d.get('train') else None)
PY
