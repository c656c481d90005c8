"""Weight-gradient launches of the 56 x 56 ResNet layers (one 64-wide operand) timed alone with HIP events.
The split count is read from the environment once per process (CAPMI_TN_SLOTS), so run it once per setting.  Usage: python tools/wgrad_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib  # noqa: E402

dev = 'cuda:0'
B = int(os.environ.get('B', 64))
shapes = [(56, 64, 64, 1), (56, 256, 64, 1), (56, 64, 256, 1), (56, 64, 64, 3), (56, 256, 128, 1), (28, 128, 128, 3), ('stem', 16, 64, 4)]
ws = _lib.wgrad_workspace(dev)
for hw, cin, cout, k in shapes:
    if hw == 'stem':        # the 7x7 / stride-2 stem as the engine runs it: a 4x4 / stride-1 conv on the 2x2 space-to-depth image (115 x 115 x 16 -> 112 x 112)
        hw, hi = 112, 115
        x = torch.randn((B, hi, hi, cin), device=dev).to(torch.bfloat16)
        g = _lib.ConvGeom(B, hi, hi, cin, hw, hw, k, k, 1, 1, 0, cin)
    else:
        x = torch.randn((B, hw, hw, cin), device=dev).to(torch.bfloat16)
        g = _lib.ConvGeom(B, hw, hw, cin, hw, hw, k, k, 1, 1, k // 2, cin)
    M, K = B * hw * hw, k * k * cin
    dy = torch.randn((B, hw, hw, cout), device=dev).to(torch.bfloat16)
    dw = torch.zeros((cout, K), device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.call('capmi_igemm_tn_wgrad', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), g, cout, cout, K, ws.data_ptr(), _lib.WGRAD_WS_BYTES, _lib.BF16, st)
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    gb = M * (cin + cout) * 2 / 1e9
    print('%dx%d cin %4d cout %4d k %d: %7.1f us  %6.0f GB/s algorithmic  %6.1f TFLOP/s' % (hw, hw, cin, cout, k, us, gb / (us * 1e-6), 2.0 * M * cout * K / us / 1e6))
