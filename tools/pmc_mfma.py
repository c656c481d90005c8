#!/usr/bin/env python
"""Per-kernel MFMA busy fraction from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, kernel
trace only), with the normalisation VALIDATED on a kernel whose answer is known.

    python tools/pmc_mfma.py <calibration_dir> <bench_dir> <out.txt> [steps]

Normalisation (MI355X_MICROARCH.md, per-instruction table + DVFS note): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles
in which a SIMD's MFMA pipe is busy, summed over all 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is reported as the SUM
over the 8 XCDs of the cycles the dispatch was active.  So
    MFMA busy = BUSY / (GUI_ACTIVE / 8 * 1024) = BUSY / (128 * GUI_ACTIVE).
calibration_dir: the same counters over tools/peaks, whose `mfma` kernel issues a KNOWN number of
v_mfma_f32_16x16x32_bf16 (16 cycles of its SIMD's matrix pipe each): the counter must equal 16 x that number, the implied
clock must be plausible, and busy fraction x peak at that clock must reproduce the launch's measured TFLOP/s; otherwise the
script fails and no table is written.  (The loop itself is NOT 100 % busy: four dependent chains per wave reach 42 % at one
wave per SIMD, 69 % at two, 73 % at four.  Round 1's tools/pmc_rates.py divided by 1024 * GUI_ACTIVE and read 8x low.)"""
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


# ---- calibration on tools/peaks' `mfma` kernel: every launch issues exactly grid/256 workgroups x 4 waves x 20000 iterations
# x 4 v_mfma_f32_16x16x32_bf16, each of which holds its SIMD's matrix pipe for 16 cycles (MI355X_MICROARCH.md)
ITERS, CHAINS, CYC = 20000, 4, 16.0
cal_rows = collections.defaultdict(dict)
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        if r['Kernel_Name'].startswith('mfma') or 'mfma(' in r['Kernel_Name']:
            d = cal_rows[r['Dispatch_Id']]
            d[r['Counter_Name']] = float(r['Counter_Value'])
            d['grid'], d['dur'] = int(r['Grid_Size']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])
if not cal_rows:
    sys.exit('no mfma kernel in the calibration pass')
cal_lines = []
for wgs in sorted({d['grid'] // 256 for d in cal_rows.values()}):
    ds = [d for d in cal_rows.values() if d['grid'] // 256 == wgs]
    expect = CYC * wgs * 4 * ITERS * CHAINS
    ratio = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in ds) / (expect * len(ds))
    clock = sum(d['GRBM_GUI_ACTIVE'] / 8.0 / d['dur'] for d in ds) / len(ds)
    frac = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (K * d['GRBM_GUI_ACTIVE']) for d in ds) / len(ds)
    tflops = sum(2.0 * 16 * 16 * 32 * wgs * 4 * ITERS * CHAINS / (d['dur'] * 1e-9) for d in ds) / len(ds) / 1e12
    if not (0.99 <= ratio <= 1.01 and 1.2 <= clock <= 2.7):
        sys.exit('calibration failed at %d workgroups: busy / (16 x MFMAs issued) = %.4f, implied clock %.2f GHz' % (wgs, ratio, clock))
    # consistency of the whole normalisation: busy fraction x peak-at-this-clock must reproduce the measured rate
    pred = frac * 2500.0 * clock / 2.4
    cal_lines.append('  %4d workgroups (%d wave(s) per SIMD): busy / (16 x MFMAs issued) = %.4f, clock %.2f GHz, MFMA busy %.1f %% -> predicts %.0f TFLOP/s, '
                     'measured %.0f' % (wgs, wgs // 256, ratio, clock, 100 * frac, pred, tflops))
    if abs(pred - tflops) > 0.05 * tflops:
        sys.exit('calibration failed: busy fraction predicts %.0f TFLOP/s, the launch ran at %.0f' % (pred, tflops))
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
    f.write('calibration on tools/peaks (register-only v_mfma_f32_16x16x32_bf16 chains; the counter must equal 16 cycles x MFMAs issued,\n'
            'GRBM_GUI_ACTIVE / 8 / duration must be a plausible clock, and busy x 2.5 PFLOP/s x clock / 2.4 GHz must reproduce the measured rate):\n')
    f.write('\n'.join(cal_lines) + '\n')
    f.write('all kernels of the run together (active-cycle weighted): %.1f %% MFMA busy\n\n' % (100 * tot_busy / (K * tot_act) if tot_act else 0.0))
    f.write('%-74s %9s %12s %10s\n' % ('kernel', 'launches', 'active share', 'MFMA busy'))
    for act, name, cnt, frac in out[:32]:
        f.write('%-74s %9.1f %11.1f%% %9.1f%%\n' % (name, cnt / steps, 100 * act / tot_act, 100 * frac))
print(open(sys.argv[3]).read())
