#!/usr/bin/env python
"""Per-kernel MFMA busy fraction from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, kernel
trace only), with the normalisation VALIDATED on a kernel whose answer is known.

    python tools/pmc_mfma.py <calibration_dir> <bench_dir> <out.txt> [steps]

Normalisation (MI355X_MICROARCH.md, per-instruction table + DVFS note): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles
in which a SIMD's MFMA pipe is busy, summed over all 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is reported as the SUM
over the 8 XCDs of the cycles the dispatch was active.  So
    MFMA busy = BUSY / (GUI_ACTIVE / 8 * 1024) = BUSY / (128 * GUI_ACTIVE).
calibration_dir: the same counters over tools/peaks, whose `mfma` kernel is a register-only chain of
v_mfma_f32_16x16x32_bf16 on every SIMD -- it must read ~100 % (1 wave per SIMD: the chain is back-to-back, 16 cycles per
MFMA); the table is only written if it does (95-105 %), otherwise the script fails.  (Round 1's tools/pmc_rates.py divided
by 1024 * GUI_ACTIVE and read 8x low.)"""
import collections
import csv
import glob
import os
import sys

K = 128.0


def load(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    grids = collections.defaultdict(set)
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r['Kernel_Name'], r.get('Dispatch_Id'))
            if key not in seen:
                seen.add(key)
                n[r['Kernel_Name']] += 1
    return acc, n


cal, ncal = load(sys.argv[1])
rows = [(k, v) for k, v in cal.items() if k.startswith('mfma') or 'mfma(' in k]
if not rows:
    sys.exit('no mfma kernel in the calibration pass: %s' % list(cal)[:5])
k, v = rows[0]
cal_frac = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (K * v['GRBM_GUI_ACTIVE'])
if not 0.95 <= cal_frac <= 1.05:
    sys.exit('calibration failed: the register-only MFMA loop reads %.1f %% busy with K = %g' % (100 * cal_frac, K))
acc, n = load(sys.argv[2])
steps = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
out = []
tot_busy = tot_act = 0.0
for name, c in acc.items():
    busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('GRBM_GUI_ACTIVE', 0.0)
    tot_busy += busy
    tot_act += act
    out.append((act, name.replace('void ', '')[:72], n[name], busy / (K * act) if act else 0.0))
out.sort(reverse=True)
with open(sys.argv[3], 'w') as f:
    f.write('MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)   [1024 SIMDs; GUI_ACTIVE is summed over 8 XCDs]\n')
    f.write('calibration: tools/peaks `%s` (register-only v_mfma_f32_16x16x32_bf16 chains on every SIMD), %d launches: %.1f %% busy\n'
            % (k[:40], ncal[k], 100 * cal_frac))
    f.write('all kernels of the run together (active-cycle weighted): %.1f %% MFMA busy\n\n' % (100 * tot_busy / (K * tot_act) if tot_act else 0.0))
    f.write('%-74s %9s %12s %10s\n' % ('kernel', 'launches', 'active share', 'MFMA busy'))
    for act, name, cnt, frac in out[:32]:
        f.write('%-74s %9.1f %11.1f%% %9.1f%%\n' % (name, cnt / steps, 100 * act / tot_act, 100 * frac))
print(open(sys.argv[3]).read())
