#!/bin/bash
# Kernel-trace tables (per shape, per queue) of bench.py under two environments, on ONE box.
# usage: tools/prof_ab.sh <outdir> "ENV_A=1" "ENV_B=0"
set -uo pipefail
out="$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for e in "$@"; do
  i=$((i+1))
  d="$out/trace_$i"
  rm -rf "$d"
  export $e
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$out/bench_$i.json" 2> "$out/bench_$i.err"
  unset ${e%%=*}
  python3 tools/trace_by_shape.py "$d" 13 > "$out/by_shape_$i.txt"
  python3 tools/trace_by_queue.py "$d" > "$out/by_queue_$i.txt"
  echo "$e" > "$out/env_$i.txt"
  rm -rf "$d"
done
