"""Per-launch HIP-event times of the DECODER part of the train plans (forward and backward), in plan order."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg, profiling
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
image, cap = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
for _ in range(2):
    eng.train_step(image, cap)
prog = eng._train[B]
st = eng._stream()
cur = torch.cuda.current_stream()
for tag, plan in (('F', prog['fwd_parts'][1]), ('B', prog['bwd'])):
    calls = plan.launches()
    if tag == 'B':
        calls = calls[:prog['n_dec']]
    best = None
    for rep in range(3):
        es = []
        for fn, name, args in calls:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur); fn(*[x.value if type(x).__name__ == "PtrSlot" else x for x in args], st); b.record(cur)
            es.append((a, b))
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) * 1e3 for a, b in es]
        best = t if best is None else [min(x, y) for x, y in zip(best, t)]
    print(tag, 'total %.1f us over %d launches' % (sum(best), len(best)))
    i = 0
    while i < len(calls):          # collapse runs of the same entry point
        j = i
        while j + 1 < len(calls) and calls[j + 1][1] == calls[i][1]:
            j += 1
        print('  %-34s x%-3d %8.1f us' % (calls[i][1], j - i + 1, sum(best[i:j + 1])))
        i = j + 1
