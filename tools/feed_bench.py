"""Train throughput with the inputs coming from HOST memory through the pinned, double-buffered feeder
(PCIe-inclusive rate; bench.py's `value` keeps its inputs resident in HBM).  usage: python tools/feed_bench.py [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg
from myimagecaptioningmodel_amd.feeder import DeviceFeeder
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=True)
image, cap = bench.synthetic_batch(B, cfg, 1234)
for dt in (np.float32, np.float16):
    host = [(image.astype(dt), cap)] * (steps + 10)          # reader items, pre-stacked form
    feeder = DeviceFeeder(iter(host), device='cuda:0', depth=2)
    for k, (img_d, cap_d) in enumerate(feeder):
        if k == 10:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.train_step(img_d, cap_d)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print('host-fed (%s pixels over PCIe): %.1f images/s, %.3f ms/step' % (np.dtype(dt).name, B * steps / dt_s, dt_s / steps * 1e3))
