"""Per-workgroup fixed cost of the NT kernels: time of a 1x1 forward convolution [M x K] . [N x K]^T as a function of K at fixed M, N
(cold operands: a ring of buffers larger than the Infinity Cache), with and without the statistics epilogue.  time(K) = a + b K:
`a` is what a launch pays whatever its reduction length (prologue, first-stage latency, epilogue, stores).
    python tools/nt_fixed_cost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib

dev, bf = 'cuda:0', torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def run(M, N, K, stats, ring=6, iters=12):
    H = W = int((M // 64) ** 0.5)
    assert 64 * H * W == M
    xs = [torch.randn((M, K), device=dev).to(bf) for _ in range(ring)]
    w = (torch.randn((N, K), device=dev) / K ** 0.5).to(bf)
    ys = [torch.zeros((M, N), device=dev, dtype=bf) for _ in range(ring)]
    rows = torch.zeros((4, 2 * N), device=dev)
    shift = torch.zeros((N,), device=dev)
    parts = torch.zeros(((M // 64 + 64) * N * 2,), device=dev)
    g = _lib.ConvGeom(64, H, W, K, H, W, 1, 1, 1, 1, 0, K)
    def one(i):
        if stats:
            _lib.call('capmi_igemm_nt_stat', p(xs[i % ring]), p(w), p(ys[i % ring]), g, N, K, N, p(parts), p(rows), p(shift), _lib.BF16, st())
        else:
            _lib.call('capmi_igemm_nt', p(xs[i % ring]), p(w), p(ys[i % ring]), g, N, K, N, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16, st())
    for i in range(3):
        one(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        one(i)
    b.record()
    torch.cuda.synchronize()
    sym = _lib.probe_kernel('capmi_igemm_nt', p(xs[0]), p(w), p(ys[0]), g, N, K, N, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16)
    return a.elapsed_time(b) / iters * 1e3, sym


for M, N in ((200704, 256), (50176, 512), (12544, 1024), (3136, 2048), (50176, 128)):
    print('M = %d, N = %d' % (M, N))
    for K in (32, 64, 128, 256, 512, 1024):
        t0, sym = run(M, N, K, False)
        t1, _ = run(M, N, K, True)
        by = (M * K + N * K + M * N) * 2
        print('   K %5d  plain %7.1f us (%5.2f TB/s, %6.1f TF/s)   + statistics %7.1f us   %s grid %d' % (
            K, t0, by / t0 / 1e6, 2.0 * M * N * K / t0 / 1e6, t1, sym[0].replace('void ', '').replace('(IGemmArgs)', ''), sym[1]))
