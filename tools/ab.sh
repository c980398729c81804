#!/bin/bash
# A/B one environment variable over bench.py in ONE box: tools/ab.sh VAR v1 v2 ...   (extra bench flags in $AB_FLAGS)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode --no-train-mode $AB_FLAGS 2>&1 | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['loss'], d['grad_norm'])"
done
