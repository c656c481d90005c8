#!/usr/bin/env python
"""Per-kernel LDS bank-conflict rate (and an MFMA busy column whose normalisation over XCDs is unverified: do not quote
it) from two rocprofv3 counter passes
(--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ; --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; kernel trace only).
usage: python tools/pmc_rates.py <lds_dir> <mfma_dir> <out.txt>"""
import collections
import csv
import glob
import os
import sys


def load(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r['Kernel_Name'], r.get('Dispatch_Id'))
            if key not in seen:
                seen.add(key)
                n[r['Kernel_Name']] += 1
    return acc, n


def short(k):
    k = k.replace('void ', '')
    return k[:64]


lds, n1 = load(sys.argv[1])
mf, n2 = load(sys.argv[2])
rows = []
for k in sorted(set(lds) | set(mf)):
    idx, bc = lds[k].get('SQ_LDS_IDX_ACTIVE', 0.0), lds[k].get('SQ_LDS_BANK_CONFLICT', 0.0)
    busy, act = mf[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), mf[k].get('GRBM_GUI_ACTIVE', 0.0)
    rows.append((act, k, n2.get(k, n1.get(k, 0)), (bc / idx if idx else 0.0), (busy / (act * 1024) if act else 0.0)))   # 256 CUs x 4 SIMDs
rows.sort(reverse=True)
with open(sys.argv[3], 'w') as f:
    f.write('%-66s %8s %14s %12s\n' % ('kernel', 'launches', 'LDS conflict', 'MFMA busy'))
    f.write('%-66s %8s %14s %12s\n' % ('', '', 'cycles/active', 'of SIMD time'))
    for act, k, n, c, m in rows[:28]:
        f.write('%-66s %8d %13.1f%% %11.1f%%\n' % (short(k), n, 100 * c, 100 * m))
print(open(sys.argv[3]).read())
