#!/bin/bash
# Regenerates the profiles/ set on the GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh r01        -> gpurun_out/prof_r01/*, copied into profiles/ by hand afterwards
# Four separate rocprofv3 runs of bench.py: kernel statistics (with the roofline passes), a kernel trace without
# them (timeline / per-shape / per-queue tables), and the two PMC passes (counters on their own, kernel trace only).
set -euo pipefail
tag="${1:-r01}"
export CAPMI_HEAD="${2:-}"      # head of the tree being profiled (the GPU box has no .git): tools/refresh_profiles.sh r02 $(git rev-parse --short=12 HEAD)
out="gpurun_out/prof_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
cp "$(ls $out/stats/*/*kernel_stats.csv | head -1)" "$out/${tag}_bench_kernel_stats.csv"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > "$out/bench_trace.json" 2> "$out/trace.err"
python3 tools/trace_by_shape.py "$out/trace" 13 > "$out/${tag}_kernel_time_by_shape.txt"      # 10 timed + 3 warm-up steps: per-step columns
python3 tools/trace_overlap.py "$out/trace" > "$out/${tag}_timeline_two_streams.txt"
python3 tools/trace_by_queue.py "$out/trace" > "$out/${tag}_kernel_time_by_queue.txt"
python3 tools/trace_step.py "$out/trace" > "$out/${tag}_step_timeline.txt"
timeout -k 10 300 python3 tools/bench_layers.py 2> /dev/null > "$out/${tag}_layers_alone.txt"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --no-roofline > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --no-roofline > "$out/pmc_write.json" 2> "$out/pmc_write.err"
python3 tools/pmc_summary.py "$out/pmc_fetch" "$out/pmc_write" "$out/${tag}_pmc_traffic.json" "3 steps, two lanes"
# MFMA busy per kernel, normalisation validated on the register-only MFMA loop of tools/peaks (tools/pmc_mfma.py)
hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o tools/peaks
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_cal" -- tools/peaks > "$out/peaks.txt" 2> "$out/pmc_cal.err"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --no-roofline > "$out/pmc_mfma.json" 2> "$out/pmc_mfma.err"
python3 tools/pmc_mfma.py "$out/pmc_cal" "$out/pmc_mfma" "$out/${tag}_pmc_mfma_busy.txt" 3
rm -rf "$out/pmc_cal" "$out/pmc_mfma"
# BASELINE configs[4]: kernel table of the beam-5 decode at batch 128
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/dec" -- python3 bench.py --decode --steps 10 --warmup 3 > "$out/decode_under_rocprof.json" 2> "$out/dec.err"
python3 tools/trace_by_shape.py "$out/dec" 13 > "$out/${tag}_decode_beam5_by_shape.txt"
rm -rf "$out/dec"
# operand-path batch norm (capmi_igemm_nt_bnact), levels 0 / 1 / 2: step time (3 alternating runs each) and the per-queue kernel tables
# (SKIP_INBN=1 leaves that evidence file as recorded)
if [ "${SKIP_INBN:-0}" != "1" ]; then
{
  echo "# CAPMI_INBN A/B on one box: ms per step, images/s (bench.py --steps 40 --warmup 8, three alternating runs)"
  for i in 1 2 3; do for e in CAPMI_INBN=0 CAPMI_INBN=1 CAPMI_INBN=2; do
    r=$(env $e python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$e  $r"
  done; done
} > "$out/${tag}_inbn_ab.txt"
bash tools/prof_ab.sh "$out/inbn" CAPMI_INBN=0 CAPMI_INBN=1 CAPMI_INBN=2
for i in 1 2 3; do { echo; echo "# $(cat $out/inbn/env_$i.txt): kernel time per queue (rocprofv3 --kernel-trace of bench.py --steps 10 --warmup 3)"; head -26 "$out/inbn/by_queue_$i.txt"; echo "# ... per shape (the convolutions that carry the operand path, and bn_apply):"; grep -E "halo3|inbn|bn_apply|igemm_nt_glds_kernel<(64, 128|128, 128)" "$out/inbn/by_shape_$i.txt" | head -24; } >> "$out/${tag}_inbn_ab.txt"; done
rm -rf "$out/inbn"
fi
rm -rf "$out/stats" "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
ls -la "$out"
