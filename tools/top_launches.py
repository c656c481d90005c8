"""Ranks the individual launches of one train step (bench config) by HIP-event time."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg, profiling
from myimagecaptioningmodel_amd.model import CaptionEngine
from myimagecaptioningmodel_amd._lib import ConvGeom

B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
image, cap = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
for _ in range(2):
    eng.train_step(image, cap)
prog = eng._train[B]
rows = []
st = eng._stream()
for plan, tag in ((prog['fwd'], 'F'), (prog['bwd'], 'B')):
    evs = []
    for rep in range(3):
        cur = torch.cuda.current_stream()
        es = []
        for fn, name, args in plan.launches():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur); fn(*[x.value if type(x).__name__ == "PtrSlot" else x for x in args], st); b.record(cur)
            es.append((a, b))
        torch.cuda.synchronize()
        evs.append([a.elapsed_time(b) * 1e3 for a, b in es])
    for i, (fn, name, args) in enumerate(plan.launches()):
        us = min(e[i] for e in evs)
        desc = ''
        for x in args:
            if isinstance(x, ConvGeom):
                desc = 'M=%d K=%d (%dx%d k%d s%d up%d)' % (x.B * x.Ho * x.Wo, x.kh * x.kw * x.Cin, x.Hi, x.Wi, x.kh, x.sd, x.up)
        label, fl, by = profiling.describe(name, args)
        N = args[4] if 'igemm' in name else ''
        rows.append((us, tag, i, label, desc, N, fl, by))
tot = sum(r[0] for r in rows)
print('total %.1f us over %d launches' % (tot, len(rows)))
agg = collections.defaultdict(float)
for r in rows:
    agg[r[3]] += r[0]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
    print('%-40s %8.1f us' % (k, v))
print()
for r in sorted(rows, key=lambda r: -r[0])[:60]:
    print('%7.1f us %s%-4d %-34s %-40s N=%-6s %7.1f TF/s %7.0f GB/s' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6] / r[0] / 1e6 if r[6] else 0, r[7] / r[0] / 1e3 if r[7] else 0))
