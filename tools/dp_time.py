"""Host / device time of the data-parallel step on a one-rank RCCL group (CAPMI_FORCE_DP=1): native (capmi_allreduce_bucket
rows in one three-lane plan) or torch.distributed per segment (CAPMI_NATIVE_COMM=0)."""
import os, sys, time
os.environ.setdefault('CAPMI_FORCE_DP', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg, dp
from myimagecaptioningmodel_amd.model import CaptionEngine
pg, rank, world, local = dp.init_process_group_from_env()
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=True, process_group=pg)
tr = dp.OverlappedTrainer(eng)
image, cap = bench.synthetic_batch(B, cfg, 1234)
image, cap = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
for _ in range(5):
    tr.train_step(image, cap)
torch.cuda.synchronize()
one = []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(image, cap)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    one.append((t1 - t0, time.perf_counter() - t0))
print('native' if tr.native_comm is not None else 'torch.distributed', ' enqueue %.2f ms, step on an idle device %.2f ms (medians of 8)' % (
    sorted(x[0] for x in one)[4] * 1e3, sorted(x[1] for x in one)[4] * 1e3))
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(image, cap)
torch.cuda.synchronize()
print('back to back: %.2f ms/step' % ((time.perf_counter() - t0) / n * 1e3))
tr.check_sync()         # a timed-out grid barrier anywhere above voids the numbers
import torch.distributed as dist
dist.destroy_process_group()
