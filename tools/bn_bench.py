"""Isolated bandwidth of the batch-norm kernels on ResNet-50 tensor shapes (batch 64): GB/s of algorithmic traffic.
usage: python tools/bn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib  # noqa: E402

dev = 'cuda:0'
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, iters=20):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        fn(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (hw, C) in ((56 * 56, 256), (56 * 56, 64), (28 * 28, 512), (28 * 28, 128), (14 * 14, 1024), (14 * 14, 256), (7 * 7, 2048), (7 * 7, 512)):
    M = 64 * hw
    x, dy, y, dx = (torch.randn(M, C, device=dev).to(torch.bfloat16) for _ in range(4))
    mean, invstd, scale, off, a = (torch.rand(C, device=dev) + 0.5 for _ in range(5))
    red = torch.zeros(2 * C, device=dev)
    ws = torch.zeros(_lib.lib().capmi_bn_bwd_ws_floats(M, C, _lib.BF16), device=dev)
    nbytes = M * C * 2
    t_apply = timeit(lambda st: _lib.call('capmi_bn_apply', p(x), p(mean), p(a), p(off), None, p(y), M, C, 1, _lib.BF16, st))
    t_red = timeit(lambda st: _lib.call('capmi_bn_bwd_reduce', p(dy), p(x), p(y), p(mean), p(invstd), p(ws), p(red), M, C, 0, _lib.BF16, st))
    t_bapp = timeit(lambda st: _lib.call('capmi_bn_bwd_apply', p(dy), p(x), p(y), p(mean), p(invstd), p(scale), p(red), p(dx), 0, None, 0, M, C, 0, _lib.BF16, st))
    print('M=%7d C=%4d (%5.1f MB): apply %6.1f us %5.0f GB/s | bwd reduce (+final) %6.1f us %5.0f GB/s | bwd apply %6.1f us %5.0f GB/s' % (
        M, C, nbytes / 1e6, t_apply, 2 * nbytes / t_apply / 1e3, t_red, 2 * nbytes / t_red / 1e3, t_bapp, 3 * nbytes / t_bapp / 1e3))
