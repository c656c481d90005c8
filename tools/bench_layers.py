"""Per-layer microbenchmark of the MFMA kernels on the ResNet-50 (B=64, 224x224) conv shapes:
forward, data-gradient and weight-gradient launches timed with HIP events; prints TFLOP/s and the
algorithmic GB/s of each so the weak shapes are visible.  Usage: python tools/bench_layers.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib, arch  # noqa: E402
from myimagecaptioningmodel_amd.encoder import dgrad_class_offsets  # noqa: E402

B, S = int(os.environ.get('B', 64)), 224
dev = 'cuda:0'
bf = torch.bfloat16
code = _lib.BF16


def p(t):
    return None if t is None else t.data_ptr()


def timeit(fn, iters=10):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        fn(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


enc = arch.resnet(50)
shape = {0: (S, S, 3)}
seen = set()
rows = []
for op in enc.ops:
    if isinstance(op, arch.ConvBN):
        h, w, _ = shape[op.src]
        ho, wo = (h + 2 * op.pad - op.k) // op.stride + 1, (w + 2 * op.pad - op.k) // op.stride + 1
        shape[op.dst] = (ho, wo, op.cout)
        key = (h, w, op.cin, op.cout, op.k, op.stride)
        if op.src == 0 or key in seen:
            continue
        seen.add(key)
        rows.append((op.name, h, w, op.cin, op.cout, op.k, op.stride, op.pad, ho, wo))
    elif isinstance(op, arch.Add):
        shape[op.dst] = shape[op.a]
    else:
        h, w, c = shape[op.src]
        shape[op.dst] = ((h - 1) // 2 + 1, (w - 1) // 2 + 1, c)

print('%-22s %-28s %9s %9s %9s | %9s %9s | %9s %9s' % ('layer', 'shape', 'fwd us', 'TF/s', 'GB/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
tot = [0.0, 0.0, 0.0]
for name, h, w, cin, cout, k, s, pad, ho, wo in rows:
    M, K = B * ho * wo, k * k * cin
    x = torch.randn((B, h, w, cin), device=dev).to(bf)
    wt = (torch.randn((cout, K), device=dev) / K ** 0.5).to(bf)
    wtT = (torch.randn((cin, k * k * cout), device=dev) / K ** 0.5).to(bf)
    y = torch.zeros((B, ho, wo, cout), device=dev, dtype=bf)
    dx = torch.zeros((B, h, w, cin), device=dev, dtype=bf)
    dw = torch.zeros((cout, K), device=dev, dtype=torch.float32)
    pr = _lib.lib().capmi_igemm_nt_stats_part_rows(M, cout, K, code)
    ws = torch.zeros(((M + pr - 1) // pr + 32, cout, 2), device=dev, dtype=torch.float32)
    g = _lib.ConvGeom(B, h, w, cin, ho, wo, k, k, s, 1, pad, cin)
    gd = _lib.ConvGeom(B, ho, wo, cout, h, w, k, k, 1, s, k - 1 - pad, cout)
    f = lambda st: _lib.lib().capmi_igemm_nt(p(x), p(wt), p(y), g, cout, K, cout, None, None, 0, None, 0, p(ws), 0, 0, 0, code, st)
    if s == 1:
        d = lambda st: _lib.lib().capmi_igemm_nt(p(y), p(wtT), p(dx), gd, cin, k * k * cout, cin, None, None, 0, None, 0, None, 0, 0, 0, code, st)
    else:
        # what the engine launches (encoder.plan_backward): one dense GEMM per output-parity class over the compact pixel grid,
        # the classes as ONE grouped launch (+ a zero fill when the classes do not cover every pixel: 1x1 / stride 2)
        classes = dgrad_class_offsets(k, s, pad)
        calls = (_lib.NtCall * len(classes))()
        wcls = []
        for cl, ((ph, pw), (d0h, d0w, nkh, nkw)) in zip(calls, classes.items()):
            hc, wc = (h - ph + s - 1) // s, (w - pw + s - 1) // s
            wk = (torch.randn((cin, nkh * nkw * cout), device=dev) / K ** 0.5).to(bf)
            wcls.append(wk)
            cl.x, cl.w, cl.y = p(y), p(wk), p(dx)
            cl.g = _lib.ConvGeom(B, ho, wo, cout, hc, wc, nkh, nkw, 1, 1, -d0h, cout, s, ph, pw, h, w)
            cl.N, cl.ldw, cl.ldy = cin, nkh * nkw * cout, cin
        covered = len(classes) == s * s

        def d(st, calls=calls, n=len(classes), covered=covered):
            if not covered:
                _lib.lib().capmi_fill_f32(p(dx), 0.0, dx.numel() // 2, st)
            return _lib.lib().capmi_igemm_nt_group(calls, n, code, st)
    wsb = _lib.wgrad_workspace(dev)
    wg = lambda st: _lib.lib().capmi_igemm_tn_wgrad(p(x), p(y), p(dw), g, cout, cout, K, p(wsb), _lib.WGRAD_WS_BYTES, code, st)
    tf, td, tw = timeit(f), timeit(d), timeit(wg)
    fl = 2.0 * M * cout * K
    by = (B * h * w * cin + M * cout) * 2
    n = sum(1 for o in enc.ops if isinstance(o, arch.ConvBN) and (o.cin, o.cout, o.k, o.stride) == (cin, cout, k, s)
            and shape[o.src][0] == h)
    tot[0] += tf * n; tot[1] += td * n; tot[2] += tw * n
    print('%-22s %-28s %9.1f %9.1f %9.0f | %9.1f %9.1f | %9.1f %9.1f  x%d' % (
        name, '%dx%d %d->%d k%d s%d' % (h, w, cin, cout, k, s), tf, fl / tf / 1e6, by / tf / 1e3, td, fl / td / 1e6, tw, fl / tw / 1e6, n))
print('sum over the net (us): fwd %.0f dgrad %.0f wgrad %.0f' % tuple(tot))
