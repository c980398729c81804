#!/bin/bash
# tools/ab2.sh "ENV1=a ENV2=b" "ENV1=c" ...: one bench line per environment setting, same box
for e in "$@"; do
  env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode --no-train-mode $AB_FLAGS 2>&1 | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d['loss'], d['grad_norm'])"
done
