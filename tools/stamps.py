"""Diagnostic: per-workgroup s_memtime stamps of the tiled NT kernel (libcapmi_stamps.so, built with
-DCAPMI_STAMPS; never the product library).  Prints where a workgroup's lifetime goes."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib
here = os.path.dirname(os.path.abspath(__file__))
_lib.LIB_PATH = os.path.join(here, 'libcapmi_stamps.so')
L = _lib.lib()
L.capmi_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = 'cuda:0'; bf = torch.bfloat16; code = _lib.BF16
def p(t): return None if t is None else t.data_ptr()
for (h, cin, cout, k) in [(56, 64, 256, 1), (56, 256, 64, 1), (28, 128, 128, 3), (7, 512, 512, 3)]:
    B = 64; M = B * h * h; K = k * k * cin
    x = torch.randn((B, h, h, cin), device=dev).to(bf); w = (torch.randn((cout, K), device=dev) / K ** .5).to(bf)
    y = torch.zeros((B, h, h, cout), device=dev, dtype=bf)
    g = _lib.ConvGeom(B, h, h, cin, h, h, k, k, 1, 1, k // 2, cin)
    nblk = 200000
    stamps = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: L.capmi_igemm_nt(p(x), p(w), p(y), g, cout, K, cout, None, None, 0, None, 0, None, 0, 0, 0, code, st)
    L.capmi_debug_set_stamp_buffer(None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    L.capmi_debug_set_stamp_buffer(p(stamps))
    f(); torch.cuda.synchronize()
    L.capmi_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy()
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    d = np.diff(s[:, :7], axis=1).astype(np.float64)
    names = ['issue loads', 'wait+LDS store', 'k-loop', 'stats/bias', 'store loop', 'drain stores']
    print('%dx%d %d->%d k%d: %d workgroups, kernel span %.1f us (100 MHz ticks -> us = /100)' % (h, h, cin, cout, k, len(s), (s[:, 6].max() - t0) / 100.0))
    for i, n in enumerate(names):
        print('   %-16s median %7.2f us   p90 %7.2f us' % (n, np.median(d[:, i]) / 100.0, np.percentile(d[:, i], 90) / 100.0))
    life = (s[:, 6] - s[:, 0]) / 100.0
    print('   workgroup lifetime median %.2f us  p90 %.2f us; start-time spread: %.1f us' % (np.median(life), np.percentile(life, 90), (s[:, 0].max() - t0) / 100.0))
