#!/usr/bin/env python
"""Fold two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same
command) into profiles/rNN_pmc_traffic.json: HBM bytes per launch for every kernel symbol.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [note]

Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are reported in KiB; on gfx950
FETCH_SIZE counts 64 B for each 128-B request, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == counter:
                a = acc[r['Kernel_Name']]
                a[0] += float(r['Counter_Value'])
                a[1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ''
    fetch, write = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
    kernels = {}
    for name in fetch:
        fs, fn = fetch[name]
        ws, wn = write.get(name, (0.0, 0))
        f = fs / fn * 1024.0 * 2.0
        w = ws / wn * 1024.0 if wn else 0.0
        kernels[name] = dict(launches=fn, fetch_bytes_per_launch=f, write_bytes_per_launch=w, hbm_bytes_per_launch=f + w)
    import subprocess
    try:
        head = subprocess.check_output(['git', 'rev-parse', '--short=12', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None          # (the GPU box has no .git: refresh_profiles.sh passes the head through CAPMI_HEAD)
    head = os.environ.get('CAPMI_HEAD') or head
    json.dump(dict(head=head, note='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches); FETCH_SIZE doubled '
                        'per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); bytes per launch averaged over all '
                        'launches of the kernel symbol. ' + note, kernels=kernels), open(out, 'w'), indent=1)
    print('wrote', out, len(kernels), 'kernels')


if __name__ == '__main__':
    main()
