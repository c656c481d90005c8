"""Debug aid: which encoder tensors does the backward plan modify (it must modify none of raw / act)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import arch, default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
eng.forward_loss(image, cap)
torch.cuda.synchronize()
prog = eng._train[B]
enc = prog['enc']
snap_raw = {k: v.clone() for k, v in enc.raw.items()}
snap_act = {k: v.clone() for k, v in enc.act.items()}
prog['bwd'].run(eng._stream())
torch.cuda.synchronize()
names = {op.dst: op.name for op in enc.enc.ops if isinstance(op, arch.ConvBN)}
for k, v in enc.raw.items():
    n = int((v != snap_raw[k]).sum())
    if n:
        idx = (v != snap_raw[k]).nonzero()
        print('raw', names.get(k, k), tuple(v.shape), 'changed elements', n, 'first', idx[0].tolist(), 'last', idx[-1].tolist())
for k, v in enc.act.items():
    n = int((v != snap_act[k]).sum())
    if n:
        idx = (v != snap_act[k]).nonzero()
        print('act', names.get(k, k), tuple(v.shape), 'changed elements', n, 'first', idx[0].tolist(), 'last', idx[-1].tolist())
print('done')
