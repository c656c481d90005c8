"""Debug aid: for the pixels the in-model halo conv gets wrong, which (tap, channel chunk) contribution is missing / garbage?"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import _lib, arch, default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
for attempt in range(4):
    eng.forward_loss(image, cap)
    torch.cuda.synchronize()
    enc = eng._train[B]['enc']
    op = [o for o in enc.enc.ops if isinstance(o, arch.ConvBN) and o.name == 'res2_1_branch2b'][0]
    x, w = enc.act[op.src].float(), eng.W(op.name + '_weights').float().view(64, 3, 3, 64)
    raw = enc.raw[op.dst].float()
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    d = raw - ref
    badpix = (d.abs().amax(-1) > 0.02 * ref.abs().max()).nonzero()
    print('attempt', attempt, 'bad pixels', len(badpix))
    if len(badpix) == 0:
        continue
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))                       # [B, H+2, W+2, C]
    for (b, h, ww) in badpix[:6].tolist():
        m = (b * 56 + h) * 56 + ww
        diff = d[b, h, ww]                                   # [64] over output channels
        best = None
        for tap in range(9):
            r, q = tap // 3, tap % 3
            for cc in range(2):
                contrib = w[:, r, q, cc * 32:(cc + 1) * 32] @ xp[b, h + r, ww + q, cc * 32:(cc + 1) * 32]
                res = float((diff + contrib).norm() / (diff.norm() + 1e-9))     # diff == -contrib: that k-step is MISSING
                if best is None or res < best[0]:
                    best = (res, tap, cc)
        print('  pixel m=%d (tile row %d, b %d h %d w %d): |diff| %.3f ; best single missing (tap, chunk) = (%d, %d) leaves %.2f of it' % (
            m, m % 64, b, h, ww, float(diff.norm()), best[1], best[2], best[0]))
    break
