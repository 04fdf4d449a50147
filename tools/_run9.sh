python -m pytest tests/test_gpu_kernels.py -x -q -k "register_resident or dot2 or dwconv" 2>&1 | tail -8
python tools/tune_probe.py 2>&1 | grep -E "block|conv2 .*(k=|M=.*(96|192)->)|cls_reg" | cut -c1-330 > gpurun_out/tune_probe6.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --tune-file gpurun_out/r3_tune_d.json --per-op > gpurun_out/r3_b5.json 2> gpurun_out/r3_b5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_b5.json'))
print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['frac'], d['roofline']['kernel'])
PY
