"""Aggregates a rocprofv3 --kernel-trace CSV by (kernel, grid size): launches, mean us, total ms.
usage: python tools/trace_by_shape.py <dir with *kernel_trace.csv> [steps]   (steps: divide totals into per-step ms)"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')))].append(
        int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
print('total %.3f ms/step' % (tot / steps / 1e6))
for (name, grid, wg), v in rows[:70]:
    print('%-60s grid %8s wg %4s  n/step %6.1f  mean %8.1f us  %7.3f ms/step' % (name, grid, wg, len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / steps / 1e6))
