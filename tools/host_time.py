"""Host-side cost of one train step: time to ENQUEUE a step vs time to run it.
  enqueue (idle device): sync, then time one train_step() call until it returns -- the pure host cost of the launch table
  enqueue (steady state): 20 back-to-back calls without a sync (includes any back-pressure of a full HIP queue)
CAPMI_PY_PLAN=1 gives the round-1 host path (one ctypes call per launch) for comparison."""
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
one = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(image, cap)
    one.append(time.perf_counter() - t0)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step(image, cap)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
prog = eng._train[B]
rows = sum(len(p) for p in (prog['fwd'], prog['bwd_opt']))
print('plan rows %d   enqueue on an idle device %.2f ms/step (median of 10)   steady-state enqueue %.2f ms/step   end-to-end %.2f ms/step   [%s]'
      % (rows, sorted(one)[5] * 1e3, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, 'CAPMI_PY_PLAN=1' if os.environ.get('CAPMI_PY_PLAN') == '1' else 'capmi_plan_run'))
