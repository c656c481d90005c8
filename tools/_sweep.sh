set -e
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/record.txt
CAPMI_TEST_RECORD=gpurun_out/r4/record.txt timeout -k 10 1100 python -m pytest tests -q -m gpu -x -k "round4 or bf16_gradient or batch_norm_backward_sums" 2>&1 | tail -25
cat gpurun_out/r4/record.txt
for i in 1 2; do for e in "CAPMI_BNSUM=1 CAPMI_STAT_APPLY=1" "CAPMI_BNSUM=1 CAPMI_STAT_APPLY=0" "CAPMI_BNSUM=0 CAPMI_STAT_APPLY=0"; do
  r=$(env $e python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$e  $r"
done; done
