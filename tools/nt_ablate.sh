#!/bin/bash
# builds tools/nt_ablate_{0,1,2,4} (run here, cross-compiles); on the GPU box: for a in 0 1 2 4; do tools/nt_ablate_$a; done
set -euo pipefail
cd "$(dirname "$0")/.."
for a in 0 1 2 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DCAPMI_NT_ABL=$a -Iinclude -Imyimagecaptioningmodel_amd/csrc tools/nt_ablate.hip myimagecaptioningmodel_amd/csrc/capi.hip -o tools/nt_ablate_$a &
done
wait
ls -la tools/nt_ablate_*
