#!/bin/bash
# A/B on ONE box (box-to-box spread is larger than most single changes): alternate two environments, 3 runs each.
# usage: tools/ab.sh "ENV_A=1" "ENV_B=0" [steps]
a="$1"; b="$2"; steps="${3:-60}"
for i in 1 2 3; do
  for e in "$a" "$b"; do
    r=$(env $e python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$e  ms/step, images/s: $r"
  done
done
