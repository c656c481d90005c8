"""One train step of a rocprofv3 --kernel-trace CSV as a launch-by-launch timeline (steps delimited by the loss kernel):
start offset, queue, duration, gap to the previous launch on the same queue, kernel (+ grid).
usage: trace_step.py <dir> [step index from the end, default 2]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', ''))
        for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if 'xent_fwd' in r[3]]
# a step runs from the first encoder kernel (s2d_stem) before a loss kernel to the one before the next
stems = [i for i, r in enumerate(rows) if 's2d_stem_kernel' in r[3]]
lo = [i for i in stems if i < marks[-back]][-1]
hi = [i for i in stems if i > marks[-back]]
hi = hi[0] if hi else len(rows)
sel = rows[lo:hi]
t0 = sel[0][0]
last = {}
qs = sorted({r[2] for r in sel})
print('step: %d launches, %.3f ms from first start to last end' % (len(sel), (max(r[1] for r in sel) - t0) / 1e6))
for s, e, q, n, g, w in sel:
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)[:58]
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print('%9.1f us  q%d  %7.1f us  gap %6.1f  %-58s wgs %6d' % ((s - t0) / 1e3, qs.index(q), (e - s) / 1e3, gap, n, int(g) // max(1, int(w or 1))))
