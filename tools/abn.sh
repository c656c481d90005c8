#!/bin/bash
# N-way A/B on ONE box: alternate any number of environments (each argument is "VAR=value" or "VAR1=a VAR2=b"), 3 rounds.
# usage: STEPS=60 tools/abn.sh "ENV_A=1" "ENV_B=0" "ENV_C=2" ...
steps="${STEPS:-60}"
for i in 1 2 3; do
  for e in "$@"; do
    r=$(env $e python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$e  ms/step, images/s: $r"
  done
done
