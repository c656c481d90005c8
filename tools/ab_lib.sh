#!/bin/bash
# A/B of the in-tree libcapmi.so against another build of the library (same C ABI) on ONE box, alternating runs.
# usage: tools/ab_lib.sh <other .so> [steps]     (CAPMI_LIB selects the library: myimagecaptioningmodel_amd/_lib.py)
other="$1"; steps="${2:-40}"
for i in 1 2 3; do
  for lib in "" "$other"; do
    r=$(CAPMI_LIB="$lib" python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "${lib:-in-tree}  ms/step, images/s: $r"
  done
done
