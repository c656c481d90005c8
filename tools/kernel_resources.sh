#!/bin/bash
# Lists VGPR / spill / LDS usage of every kernel in a .hip file (device assembly of gfx950).
# usage: tools/kernel_resources.sh myimagecaptioningmodel_amd/csrc/igemm.hip
src="$1"; tmp="$(mktemp -d)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$(dirname "$0")/../include" -I"$(dirname "$src")" -S --cuda-device-only "$src" -o "$tmp/k.s" 2>/dev/null
awk '/^  - \.agpr_count|^  - \.args/ {lds="";name="";v="";sp="";ag=""}
     /\.agpr_count:/ {ag=$NF} /\.group_segment_fixed_size:/ {lds=$NF} /^[ \t]+\.name:/ {name=$NF} /\.vgpr_count:/ {v=$NF}
     /\.vgpr_spill_count:/ {sp=$NF; printf "%-70s vgpr %4s agpr %4s spill %3s lds %6s\n", name, v, ag, sp, lds}' "$tmp/k.s"
rm -rf "$tmp"
