"""Host-side cost of one train step: time to ENQUEUE a step (no device sync) vs time to run it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=True)
image, cap = bench.synthetic_batch(B, cfg, 1234)
image, cap = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
for _ in range(5):
    eng.train_step(image, cap)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step(image, cap)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step   end-to-end %.2f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
