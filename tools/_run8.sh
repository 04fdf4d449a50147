python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -6
python tools/tune_probe.py 2>&1 | grep -E "block|conv2 .*M=.*(96|192)->|20.conv1|9.cv2|26.conv1" | cut -c1-420 > gpurun_out/tune_probe5.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --tune-file gpurun_out/r3_tune_c.json --per-op > gpurun_out/r3_b4.json 2> gpurun_out/r3_b4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_b4.json'))
print(d['value'], d['ms_per_step'], d['forward_only'], d['roofline']['frac'], d['roofline']['kernel'])
PY
