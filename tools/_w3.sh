python tools/conv_wgrad_bench.py 2>&1 | grep -E "per step"
for gx in 1536 2048; do echo GX=$gx; MAF_WGRAD_GX=$gx python tools/conv_wgrad_bench.py 2>&1 | grep -E "^\| (160|640|320) \| [0-9]+ \| [0-9]+ \| 1 |^\| 80 \| 48 \| 48"; done
for i in 1 2; do python bench.py --train --steps 30 --warmup 10 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"; done
