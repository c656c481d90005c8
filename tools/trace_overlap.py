"""Timeline summary of a rocprofv3 --kernel-trace CSV: per-queue busy time, union busy time and wall span
of the last three train steps (delimited by the loss kernel).  usage: trace_overlap.py <dir> """
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:40]) for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if 'xent_fwd' in r[3]]      # once per train step
lo, hi = adam[-4], adam[-1]            # three whole steps
sel = rows[lo:hi]
span = sel[-1][1] - sel[0][0]
busy = collections.defaultdict(int)
for s, e, q, n in sel:
    busy[q] += e - s
ev = sorted([(s, 1) for s, e, q, n in sel] + [(e, -1) for s, e, q, n in sel])
union = both = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: union += t - last
    if depth >= 2: both += t - last
    depth += d; last = t
print('3 steps: span %.3f ms/step, union busy %.3f, >=2 kernels in flight %.3f ms/step' % (span / 3e6, union / 3e6, both / 3e6))
for q, b in busy.items():
    print('queue', q, '%.3f ms/step' % (b / 3e6))
# largest idle gaps (no kernel in flight) inside the selected steps
iv = sorted((s, e, n) for s, e, q, n in sel)
gaps, cur_end, prev = [], iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        cur_end, prev = e, n
gaps.sort(reverse=True)
print('idle total %.3f ms/step in %d gaps/step; largest:' % (sum(g[0] for g in gaps) / 3e6, len(gaps) // 3))
for g in gaps[:12]:
    print('  %7.1f us  after %-40s before %s' % (g[0] / 1e3, g[1], g[2]))
import collections as C
by = C.Counter()
for g in gaps:
    by[(g[1][:24], g[2][:24])] += g[0]
for k, v in by.most_common(10):
    print('  %7.1f us/step  %s -> %s' % (v / 3e3, k[0], k[1]))
