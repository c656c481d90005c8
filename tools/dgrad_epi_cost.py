"""Cost of the data-gradient epilogues: a 1x1 / 3x3 data gradient [M x Cout] . [Cin x Cout]^T with the ReLU mask as bits
(capmi_igemm_nt, class 6) against the same launch with the batch-norm backward sums (capmi_igemm_nt_bnsum, class 7), with and
without an addend, on the shapes ResNet-50 has at batch 64 (cold operands: a ring of buffers).
    python tools/dgrad_epi_cost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib

dev, bf = 'cuda:0', torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def run(B, H, C, Co, k, addend, ring=5, iters=10):
    M = B * H * H
    pad = (k - 1) // 2
    K = k * k * Co
    dys = [torch.randn((M, Co), device=dev).to(bf) for _ in range(ring)]
    wT = (torch.randn((C, K), device=dev) / K ** 0.5).to(bf)
    outs = [torch.randn((M, C), device=dev).to(bf) for _ in range(ring)]
    raws = [torch.randn((M, C), device=dev).to(bf) for _ in range(ring)]
    bits = [torch.randint(0, 255, (M * C // 8,), device=dev, dtype=torch.uint8) for _ in range(ring)]
    mu, inv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    acc, red = torch.zeros((4, 2 * C), device=dev), torch.zeros(2 * C, device=dev)
    g = _lib.ConvGeom(B, H, H, Co, H, H, k, k, 1, 1, k - 1 - pad, Co)
    R = _lib.lib().capmi_igemm_nt_bnsum_part_rows(g, C, _lib.BF16)
    parts = torch.zeros((((M + max(R, 1) - 1) // max(R, 1)) * 2 * C,), device=dev)
    dact = _lib.ACT_RELU | _lib.DACT_BITMASK

    def a6(i):
        o = outs[i % ring]
        return ('capmi_igemm_nt', p(dys[i % ring]), p(wT), p(o), g, C, K, C, None, p(o) if addend else None, C, p(bits[i % ring]), C, None, 0, dact, 0, _lib.BF16)

    def a4(i):
        o = outs[i % ring]
        return ('capmi_igemm_nt', p(dys[i % ring]), p(wT), p(o), g, C, K, C, None, p(o) if addend else None, C, None, 0, None, 0, 0, 0, _lib.BF16)

    def a7(i):
        o = outs[i % ring]
        return ('capmi_igemm_nt_bnsum', p(dys[i % ring]), p(wT), p(o), g, C, K, C, p(o) if addend else None, C, p(bits[i % ring]), C, dact,
                p(raws[i % ring]), p(mu), p(inv), p(acc), p(parts), p(red), _lib.BF16)
    res = []
    for mk in (a4, a6, a7):
        if mk is a7 and R == 0:
            res.append(float('nan'))
            continue
        for i in range(3):
            _lib.call(*mk(i), st())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            _lib.call(*mk(i), st())
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    sym = _lib.probe_kernel(*a7(0))[0] if R else '-'
    return res, sym


for B, H, C, Co, k in ((64, 56, 64, 256, 1), (64, 56, 256, 64, 1), (64, 56, 64, 64, 3), (64, 28, 128, 512, 1), (64, 28, 512, 128, 1), (64, 28, 128, 128, 3),
                       (64, 14, 256, 1024, 1), (64, 14, 1024, 256, 1), (64, 14, 256, 256, 3), (64, 7, 512, 2048, 1), (64, 7, 2048, 512, 1), (64, 7, 512, 512, 3)):
    for addend in (False, True):
        (t4, t6, t7), sym = run(B, H, C, Co, k, addend)
        print('%2dx%-2d %4d <- %4d %dx%d %s  plain %6.1f us   mask %6.1f us (+%4.1f)   mask + sums %6.1f us  (+%4.1f)   %s' % (
            H, H, C, Co, k, k, 'addend' if addend else '      ', t4, t6, t6 - t4, t7, t7 - t6, sym.replace('void ', '').replace('(IGemmArgs)', '')), flush=True)
