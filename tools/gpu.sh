#!/bin/bash
# Local helper (container side): submit one gpurun call, waiting for a free slot when the pod is busy.
# Retries ONLY on exit code 3 ("no box or slot free right now, nothing charged"); any other outcome is final.
# usage: tools/gpu.sh <timeout_s> '<command>'
t="$1"; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 90
done
exit 3
