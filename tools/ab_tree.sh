#!/bin/bash
# A/B of the working tree against another full tree (e.g. an export of HEAD under tools/_head_tree) on ONE box, alternating runs.
# usage: tools/ab_tree.sh <other tree> [steps]
other="$1"; steps="${2:-40}"
here="$(pwd)"
for i in 1 2 3; do
  for t in "$here" "$other"; do
    r=$(cd "$t" && python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$t  ms/step, images/s: $r"
  done
done
