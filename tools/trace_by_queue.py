"""Per-queue kernel time by kernel family, and the main queue's phases, from a rocprofv3 --kernel-trace CSV (last three
train steps, delimited by the loss kernel).  usage: trace_by_queue.py <dir>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if 'xent_fwd' in r[3]]
sel = rows[marks[-4]:marks[-1]]


def family(n):
    n = re.sub(r'^void ', '', n)
    m = re.match(r'(_Z\d+)?([A-Za-z_0-9]+)', n)
    fam = m.group(2)
    t = re.search(r'<(\d+), (\d+)', n)
    if t and 'igemm' in fam:
        fam += '<%s,%s>' % t.groups()
    return fam


by = collections.defaultdict(collections.Counter)
cnt = collections.defaultdict(collections.Counter)
for s, e, q, n in sel:
    by[q][family(n)] += e - s
    cnt[q][family(n)] += 1
for q in by:
    tot = sum(by[q].values())
    print('queue %s: %.3f ms/step busy, %d launches/step' % (q, tot / 3e6, sum(cnt[q].values()) // 3))
    for k, v in by[q].most_common(22):
        print('   %-36s %7.3f ms/step  %5.1f launches  mean %6.1f us' % (k, v / 3e6, cnt[q][k] / 3, v / cnt[q][k] / 1e3))
