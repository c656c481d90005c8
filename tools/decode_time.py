import os, sys, time
import torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import bench
from myimagecaptioningmodel_amd import default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=True)
image, cap = bench.synthetic_batch(B, cfg, 1234)
image = torch.as_tensor(image).cuda()
for _ in range(3): eng.decode(image)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ids = eng.decode(image)
torch.cuda.synchronize(); print('greedy decode B=64, T=20: %.2f ms/batch' % ((time.perf_counter() - t0) / 10 * 1e3), 'graph' if eng._eval[B].get('graph') else 'eager')
