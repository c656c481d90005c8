"""Static screen of the device assembly for the LDS write-after-read hazard of lesson 33: an LDS-DMA issue
(`global_load_lds` / `buffer_load ... lds`) that follows an `s_barrier` while fragment reads (`ds_read*`) issued BEFORE that
barrier have not been retired by an `s_waitcnt lgkmcnt(0)` yet.  Straight-line approximation per kernel (branches ignored;
scalar loads, which share lgkmcnt, are ignored), so a hit is a place to read, not a verdict.
Usage: python tools/lds_war_audit.py myimagecaptioningmodel_amd/csrc/igemm.hip"""
import re
import subprocess
import sys
import tempfile
import os

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'k.s')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(root, 'include'), '-I' + os.path.dirname(src),
                    '-S', '--cuda-device-only', src, '-o', out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split('\n')
name, pending, crossed, hits = None, 0, False, {}
for ln, line in enumerate(text, 1):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name, pending, crossed = m.group(1), 0, False
        continue
    t = line.strip()
    if not name or not t or t.startswith(('.', ';')):
        continue
    op = t.split()[0]
    if op.startswith('ds_read') or op.startswith('ds_load'):
        if not crossed:
            pending += 1
    elif op == 's_waitcnt' and 'lgkmcnt(0)' in t:
        pending, crossed = 0, False
    elif op == 's_barrier':
        crossed = pending > 0
    elif (op.startswith('global_load_lds') or (op.startswith('buffer_load') and ' lds' in t)) and crossed:
        hits.setdefault(name, []).append(ln)
    elif op == 's_endpgm':
        name = None
for k, v in hits.items():
    print('%-90s %d LDS-DMA issues behind a barrier with unretired reads (first at asm line %d)' % (k[:90], len(v), v[0]))
print('%d kernels flagged' % len(hits))
