#!/bin/bash
# A/B of the beam-5 decode (BASELINE configs[4]) between the working tree and another full tree on ONE box.
other="$1"
here="$(pwd)"
for i in 1 2 3; do
  for t in "$here" "$other"; do
    r=$(cd "$t" && python bench.py --decode --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('p50_ms', d.get('extra', {})))")
    echo "$t  captions/s, p50: $r"
  done
done
