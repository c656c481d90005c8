"""Debug aid: the 3x3 / stride-1 halo-staged conv kernel against torch conv2d (f32 on bf16-rounded operands); prints where it differs."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib
dev = 'cuda:0'
for (B, H, Cin, Cout) in [(64, 56, 64, 64), (8, 56, 64, 64), (64, 28, 128, 128), (64, 14, 256, 256), (64, 7, 512, 512), (3, 9, 32, 48)]:
    torch.manual_seed(0)
    x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, 3, 3, Cin), device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    y = torch.zeros((B, H, H, Cout), device=dev, dtype=torch.bfloat16)
    g = _lib.ConvGeom(B, H, H, Cin, H, H, 3, 3, 1, 1, 1, Cin)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('capmi_igemm_nt', x.data_ptr(), w.data_ptr(), y.data_ptr(), g, Cout, 9 * Cin, Cout, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16, st)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    d = (y.float() - ref).abs()
    bad = d > 0.05 * ref.abs().max()
    print('B %d %dx%d %d->%d: rel L2 %.2e  max abs %.3g (ref max %.3g)  bad elements %d of %d' % (
        B, H, H, Cin, Cout, float((y.float() - ref).norm() / ref.norm()), float(d.max()), float(ref.abs().max()), int(bad.sum()), bad.numel()))
    if bad.any():
        idx = bad.nonzero()
        print('  first bad (b,h,w,c):', idx[:8].tolist())
        pix = (idx[:, 0] * H * H + idx[:, 1] * H + idx[:, 2])
        print('  bad pixels m mod 64:', sorted(set((pix % 64).tolist()))[:20], ' distinct pixels', len(set(pix.tolist())), ' h values', sorted(set(idx[:, 1].tolist()))[:12], ' w values', sorted(set(idx[:, 2].tolist()))[:12])
