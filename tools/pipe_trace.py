"""Timeline of the LAST decode_pipelined call in a rocprofv3 --kernel-trace CSV (tools/decode_pipe.py one ...): per-queue busy
time, the time no kernel runs at all, and the longest holes with the kernels either side.  usage: pipe_trace.py <dir> [batches]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if 's2d_stem' in r[3]]          # first kernel of every encoder
sel = rows[marks[-nb]:]
t0, t1 = sel[0][0], max(r[1] for r in sel)
print('%d batches: %.3f ms per batch, %d launches per batch' % (nb, (t1 - t0) / nb / 1e6, len(sel) // nb))
qs = sorted(set(r[2] for r in sel))
for q in qs:
    rq = [r for r in sel if r[2] == q]
    print('  queue %s: %d launches, busy %.3f ms per batch, mean %.1f us' % (q, len(rq), sum(r[1] - r[0] for r in rq) / nb / 1e6, sum(r[1] - r[0] for r in rq) / len(rq) / 1e3))
cover, holes, end, last = 0, [], sel[0][0], sel[0]
for r in sel:
    if r[0] > end:
        holes.append((r[0] - end, last, r))
    if r[1] > end:
        cover += r[1] - max(end, r[0])
        end, last = r[1], r
print('  some kernel runs %.3f ms per batch; nothing runs %.3f ms per batch in %d holes (%d of them > 10 us: %.3f ms per batch)'
      % (cover / nb / 1e6, sum(h[0] for h in holes) / nb / 1e6, len(holes), sum(1 for h in holes if h[0] > 10000), sum(h[0] for h in holes if h[0] > 10000) / nb / 1e6))
holes.sort(key=lambda h: -h[0])
for h in holes[:12]:
    print('    %8.1f us after %-50s (queue %s) before %-50s (queue %s)' % (h[0] / 1e3, h[1][3][:50], h[1][2], h[2][3][:50], h[2][2]))
