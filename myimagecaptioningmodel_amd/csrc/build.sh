#!/bin/bash
# Builds libcapmi.so (gfx950 only) next to the Python package.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/../libcapmi.so"
objdir="$here/build"
mkdir -p "$objdir"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$here $*"
pids=()
for f in capi plan igemm bn_ops encoder_ops decoder_ops optim; do
    src="$here/$f.hip"; obj="$objdir/$f.o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$here/common.h" -nt "$obj" ] || [ "$here/bn_merge.h" -nt "$obj" ] || [ "$root/include/capmi.h" -nt "$obj" ]; then
        hipcc $flags -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$objdir"/capi.o "$objdir"/plan.o "$objdir"/igemm.o "$objdir"/bn_ops.o "$objdir"/encoder_ops.o "$objdir"/decoder_ops.o "$objdir"/optim.o -ldl
echo "built $out"
