"""Debug aid: the halo conv on the ENGINE's own tensors, repeated -- does the address / layout of the in-model operands matter?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import _lib, arch, default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine
B = 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
image, cap = bench.synthetic_batch(B, cfg, 1234)
eng.forward_loss(image, cap)
torch.cuda.synchronize()
enc = eng._train[B]['enc']
op = [o for o in enc.enc.ops if isinstance(o, arch.ConvBN) and o.name == 'res2_1_branch2b'][0]
x, w = enc.act[op.src], eng.W(op.name + '_weights')
print('x ptr %% 4096 = %d, w ptr %% 4096 = %d, w offset in low (elements) = %d' % (x.data_ptr() % 4096, w.data_ptr() % 4096, (w.data_ptr() - eng.low.data_ptr()) // 2))
g = enc._conv_geom(op)
st = torch.cuda.current_stream().cuda_stream
def rep(xx, ww, stats):
    outs = []
    for r in range(6):
        y = torch.zeros_like(enc.raw[op.dst])
        _lib.call('capmi_igemm_nt', xx.data_ptr(), ww.data_ptr(), y.data_ptr(), g, 64, 576, 64, None, None, 0, None, 0, stats, 0, 0, 0, _lib.BF16, st)
        torch.cuda.synchronize()
        outs.append(y)
    return [int((o != outs[0]).sum()) for o in outs[1:]], int((outs[0] != enc.raw[op.dst]).sum())
print('engine x, engine w     :', rep(x, w, enc.bn[op.dst]['stats'].data_ptr()))
print('engine x, copied w     :', rep(x, w.clone(), None))
print('copied x, engine w     :', rep(x.clone(), w, None))
wpad = torch.zeros(w.numel() + 8, dtype=w.dtype, device='cuda:0')
wpad[8:].copy_(w.reshape(-1))
print('engine x, w at +16 B   :', rep(x, wpad[8:], None))
