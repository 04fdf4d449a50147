#!/bin/bash
for i in 1 2 3; do
for f in 1 0; do
  echo "MAF_CAT_FREE=$f"
  MAF_CAT_FREE=$f python bench.py --train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('value'))"
done; done
