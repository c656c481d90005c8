"""Debug aid: in-model conv outputs of the 3x3 stride-1 layers against torch conv2d on the engine's own input tensors."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import arch, default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
mode = sys.argv[1] if len(sys.argv) > 1 else 'fb'
if mode == 'f':
    eng.forward_loss(image, cap)
else:
    eng.forward_backward(image, cap)
torch.cuda.synchronize()
enc = eng._train[B]['enc']
params = eng.export_reference_params()
for op in enc.enc.ops:
    if not isinstance(op, arch.ConvBN) or op.k != 3 or op.stride != 1:
        continue
    x = enc.act[op.src].float()
    w = torch.as_tensor(params[op.name + '_weights']).cuda().to(torch.bfloat16).float()     # [co, ci, 3, 3]
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    raw = enc.raw[op.dst].float()
    d = (raw - ref).abs()
    bad = d > 0.03 * ref.abs().max()
    print('%-18s rel L2 %.2e  bad %d' % (op.name, float((raw - ref).norm() / ref.norm()), int(bad.sum())), end='')
    if bad.any():
        idx = bad.nonzero()
        H = x.shape[1]
        pix = idx[:, 0] * H * H + idx[:, 1] * H + idx[:, 2]
        print('  pixels %d  b %s h %s w %s  m%%64 %s' % (len(set(pix.tolist())), sorted(set(idx[:, 0].tolist()))[:6], sorted(set(idx[:, 1].tolist()))[:8],
              sorted(set(idx[:, 2].tolist()))[:8], sorted(set((pix % 64).tolist()))[:10]), end='')
    print()
