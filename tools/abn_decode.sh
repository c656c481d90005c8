#!/bin/bash
# N-way A/B of the beam-5 decode bench (BASELINE configs[4]) on ONE box: each argument is an environment ("VAR=value ...").
for i in 1 2 3; do
  for e in "$@"; do
    r=$(env $e python bench.py --decode --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'))")
    echo "$e  captions/s: $r"
  done
done
