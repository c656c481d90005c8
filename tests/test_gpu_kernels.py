"""Kernel-level parity (MI355X only): every libcapmi.so kernel family, called through the C ABI,
against the NumPy oracle ops on the same seeded inputs.  f32 runs at reference precision
(tight tolerance); bf16 compares against the oracle evaluated on bf16-rounded inputs with a
tolerance set by bf16's 8-bit mantissa."""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
TOL = {'f32': dict(rtol=2e-4, atol=2e-4), 'bf16': dict(rtol=3e-2, atol=3e-2)}


def _env():
    from myimagecaptioningmodel_amd import _lib
    tdt = {'f32': torch.float32, 'bf16': torch.bfloat16}
    code = {'f32': _lib.F32, 'bf16': _lib.BF16}
    return _lib, tdt, code


def stream():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []      # device tensors passed as raw pointers must outlive the (asynchronous) launch


def dev(a, dt):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(device=DEV, dtype=dt)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def rnd(a, dtype):
    """Round a float64 array through the storage dtype (what the kernel actually reads)."""
    if dtype == 'bf16':
        return torch.as_tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    return a.astype(np.float32).astype(np.float64)


def host(t):
    torch.cuda.synchronize()
    return t.detach().to(torch.float64).cpu().numpy()


def p(t):
    return None if t is None else t.data_ptr()


def check(got, want, dtype, scale=None, name=''):
    tol = TOL[dtype]
    s = np.abs(want).max() if scale is None else scale
    err = np.abs(got - want).max()
    assert err <= tol['atol'] * max(1.0, s) , '%s: max err %g (scale %g)' % (name, err, s)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (200, 72, 40), (130, 136, 104), (1216, 1000, 64), (37, 8, 16), (300, 260, 512),
                                   (64, 2048, 512), (64, 512, 2048), (37, 50, 256), (4, 136, 384),
                                   (24600, 260, 520), (49152, 136, 512),      # these two run the big LDS-DMA pipeline tiles (bf16)
                                   (300, 260, 2048), (700, 40, 1088), (3136, 512, 1024)])   # two k-groups per workgroup (bf16)
def test_gemm_nt_epilogues(dtype, M, N, K):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(M + N + K)
    a = rnd(rng.standard_normal((M, K)), dtype)
    w = rnd(rng.standard_normal((N, K)) / np.sqrt(K), dtype)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rnd(rng.standard_normal((M, N)), dtype)
    ysaved = rnd(np.tanh(rng.standard_normal((M, N))), dtype)
    A, W = dev(a, tdt[dtype]), dev(w, tdt[dtype])
    # plain + bias + tanh
    Y = torch.zeros((M, N), dtype=tdt[dtype], device=DEV)
    g = _lib.gemm_geom(M, K)
    _lib.call('capmi_igemm_nt', p(A), p(W), p(Y), g, N, K, N, p(dev(bias, torch.float32)), None, 0, None, 0, None,
              _lib.ACT_TANH, 0, 0, code[dtype], stream())
    check(host(Y), np.tanh(a @ w.T + bias), dtype, name='bias+tanh')
    # addend + dact + f32 output + stats
    Y32 = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    pr = _lib.lib().capmi_igemm_nt_stats_part_rows(M, N, K, code[dtype])
    nparts = (M + pr - 1) // pr
    stats = torch.full((nparts * N * 2,), float('nan'), dtype=torch.float32, device=DEV)
    _lib.call('capmi_igemm_nt', p(A), p(W), p(Y32), g, N, K, N, None, p(dev(add, tdt[dtype])), N, p(dev(ysaved, tdt[dtype])), N,
              None, 0, _lib.ACT_TANH, 1, code[dtype], stream())
    pre = a @ w.T + add
    check(host(Y32), pre * (1 - ysaved ** 2), dtype, name='addend+dact')
    # in-place accumulate (y == addend), as the recurrent gate GEMM and the data-gradient sums use it
    Yacc = dev(add, tdt[dtype]).clone()
    _lib.call('capmi_igemm_nt', p(A), p(W), p(Yacc), g, N, K, N, None, p(Yacc), N, None, 0, None, 0, 0, 0, code[dtype], stream())
    check(host(Yacc), pre, dtype, name='in-place addend')
    _lib.call('capmi_igemm_nt', p(A), p(W), p(Y32), g, N, K, N, p(dev(bias, torch.float32)), None, 0, None, 0,
              p(stats), 0, 0, 1, code[dtype], stream())
    pre = a @ w.T + bias
    # fused statistics: exact (mean, M2) per part of `pr` rows; merged here like bn_finalize does
    st = host(stats).reshape(nparts, N, 2)
    for pi in range(nparts):
        blk = pre[pi * pr:(pi + 1) * pr]
        check(st[pi, :, 0], blk.mean(0), dtype, name='part mean')
        check(st[pi, :, 1], ((blk - blk.mean(0)) ** 2).sum(0), dtype, scale=(blk ** 2).sum(0).max(), name='part M2')


@pytest.mark.parametrize('M,N,K', [(1216, 512, 10000), (1856, 1024, 20000), (640, 512, 4096), (300, 96, 8200), (1216, 512, 1024)])
def test_split_k_product_equals_the_plain_one(M, N, K):
    """capmi_igemm_nt_splitk (the tied projection's data gradient, model_adaAttention_aic.py:25 backward: [T*B][V] x [V][E])
    against capmi_igemm_nt on the same operands and against the oracle's matmul: same product, f32 slabs added in a fixed
    order and rounded to bf16 once; bit-reproducible; shapes outside the split fall through to the plain kernel."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(M + K)
    x = rnd(rng.standard_normal((M, K)), 'bf16')
    w = rnd(rng.standard_normal((N, K)) / np.sqrt(K), 'bf16')
    X, Wt = dev(x, torch.bfloat16), dev(w, torch.bfloat16)
    need = _lib.lib().capmi_igemm_nt_splitk_ws_bytes(M, N, K, _lib.BF16)
    assert (need > 0) == (K >= 4096 and ((M + 127) // 128) * ((N + 127) // 128) <= 128 and need % (M * N * 4) == 0)
    ws = torch.zeros(max(need, 16) // 4, dtype=torch.float32, device=DEV)
    Y1, Y2, Y3 = (torch.zeros((M, N), dtype=torch.bfloat16, device=DEV) for _ in range(3))
    _KEEP.extend([ws, Y1, Y2, Y3])
    _lib.call('capmi_igemm_nt_splitk', p(X), p(Wt), p(Y1), M, K, K, N, K, N, p(ws), ws.numel() * 4, _lib.BF16, stream())
    _lib.call('capmi_igemm_nt_splitk', p(X), p(Wt), p(Y3), M, K, K, N, K, N, p(ws), ws.numel() * 4, _lib.BF16, stream())
    g = _lib.gemm_geom(M, K)
    _lib.call('capmi_igemm_nt', p(X), p(Wt), p(Y2), g, N, K, N, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16, stream())
    torch.cuda.synchronize()
    assert torch.equal(Y1, Y3)
    want = x @ w.T
    check(host(Y1), want, 'bf16', name='split-K product')
    assert float((Y1.float() - Y2.float()).abs().max()) <= 2 ** -7 * max(1.0, float(Y2.float().abs().max()))
    if need == 0:
        assert torch.equal(Y1, Y2)
    # no workspace: the plain kernel
    Y4 = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    _lib.call('capmi_igemm_nt_splitk', p(X), p(Wt), p(Y4), M, K, K, N, K, N, None, 0, _lib.BF16, stream())
    torch.cuda.synchronize()
    assert torch.equal(Y4, Y2)


def _nhwc(x):
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,C,H,W,Co,k,s,pad', [(2, 16, 9, 9, 24, 3, 1, 1), (3, 8, 12, 10, 40, 3, 2, 1), (2, 32, 8, 8, 16, 1, 1, 0), (4, 128, 13, 11, 72, 3, 1, 1),
                                                (2, 16, 8, 8, 32, 1, 2, 0), (1, 64, 14, 14, 64, 3, 1, 1),
                                                (48, 64, 33, 31, 136, 3, 1, 1), (64, 64, 57, 57, 128, 3, 2, 1)])   # LDS-DMA pipeline kernel (bf16)
def test_conv_fwd_dgrad_wgrad(dtype, B, C, H, W, Co, k, s, pad):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(B * C + Co + k + s)
    x = rnd(rng.standard_normal((B, C, H, W)), dtype)
    w = rnd(rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k), dtype)
    y = O.conv2d_fwd(x, w, s, pad)
    Ho, Wo = y.shape[2], y.shape[3]
    dy = rnd(rng.standard_normal(y.shape), dtype)
    dx, dw = O.conv2d_bwd(dy, x, w, s, pad)
    X = dev(_nhwc(x), tdt[dtype])
    Wk = dev(w.transpose(0, 2, 3, 1), tdt[dtype])                    # [Co][kh][kw][C]
    Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
    g = _lib.ConvGeom(B, H, W, C, Ho, Wo, k, k, s, 1, pad, C)
    _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, k * k * C, Co, None, None, 0, None, 0, None, 0, 0, 0, code[dtype], stream())
    check(host(Y), _nhwc(y), dtype, name='conv fwd')
    # weight gradient
    DY = dev(_nhwc(dy), tdt[dtype])
    DW = torch.zeros((Co, k, k, C), dtype=torch.float32, device=DEV)
    _lib.call('capmi_igemm_tn_wgrad', p(X), p(DY), p(DW), g, Co, Co, k * k * C, p(_lib.wgrad_workspace(DEV)), _lib.WGRAD_WS_BYTES, code[dtype], stream())
    check(host(DW), dw.transpose(0, 2, 3, 1), dtype, name='conv wgrad')
    # data gradient through the flipped/transposed weight form
    WT = torch.zeros((C, k, k, Co), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_weight_dgrad_form', p(dev(w.transpose(0, 2, 3, 1), torch.float32)), p(WT), Co, k, k, C, Co, code[dtype], stream())
    DX = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    gd = _lib.ConvGeom(B, Ho, Wo, Co, H, W, k, k, 1, s, k - 1 - pad, Co)
    _lib.call('capmi_igemm_nt', p(DY), p(WT), p(DX), gd, C, k * k * Co, C, None, None, 0, None, 0, None, 0, 0, 0, code[dtype], stream())
    check(host(DX), _nhwc(dx), dtype, name='conv dgrad')
    # accumulate form: dX += ...
    _lib.call('capmi_igemm_nt', p(DY), p(WT), p(DX), gd, C, k * k * Co, C, None, p(DX), C, None, 0, None, 0, 0, 0, code[dtype], stream())
    check(host(DX), 2 * _nhwc(dx), dtype, name='conv dgrad accumulate')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,C,H,W,Co,k,pad', [(2, 16, 12, 10, 24, 3, 1), (3, 8, 9, 11, 16, 3, 1), (2, 32, 8, 8, 16, 1, 0), (2, 16, 7, 9, 8, 1, 0),
                                              (2, 136, 12, 10, 72, 3, 1), (3, 72, 21, 19, 40, 3, 1)])
@pytest.mark.parametrize('grouped', [False, True])
def test_strided_dgrad_by_parity_classes(dtype, B, C, H, W, Co, k, pad, grouped):
    """Data gradient of a stride-2 conv as dense GEMMs per output-parity class, rows scattered to the
    strided pixels (what encoder.plan_backward launches) == the oracle's conv2d_bwd."""
    _lib, tdt, code = _env()
    from myimagecaptioningmodel_amd.encoder import dgrad_classes, dgrad_class_offsets
    s = 2
    rng = np.random.RandomState(B * C + Co + k)
    x = rnd(rng.standard_normal((B, C, H, W)), dtype)
    w = rnd(rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k), dtype)
    y = O.conv2d_fwd(x, w, s, pad)
    Ho, Wo = y.shape[2:]
    dy = rnd(rng.standard_normal(y.shape), dtype)
    dx, _ = O.conv2d_bwd(dy, x, w, s, pad)
    DY = dev(_nhwc(dy), tdt[dtype])
    base = rnd(rng.standard_normal((B, H, W, C)), dtype)          # pre-existing gradient: accumulate form
    DX = dev(base, tdt[dtype]).clone()
    offs = dgrad_class_offsets(k, s, pad)
    classes = list(dgrad_classes(k, s, pad))
    calls = (_lib.NtCall * len(classes))()
    for ci, (ph, pw, rmap, qmap) in enumerate(classes):
        d0h, d0w, nkh, nkw = offs[(ph, pw)]
        wc = np.zeros((C, nkh, nkw, Co))
        for a_, r in enumerate(rmap):
            for b_, q in enumerate(qmap):
                wc[:, a_, b_, :] = w[:, :, r, q].T
        hc, wcc = (H - ph + s - 1) // s, (W - pw + s - 1) // s
        gd = _lib.ConvGeom(B, Ho, Wo, Co, hc, wcc, nkh, nkw, 1, 1, -d0h, Co, s, ph, pw, H, W)
        WC = dev(wc, tdt[dtype])
        if grouped:
            c = calls[ci]
            c.x, c.w, c.y, c.g = p(DY), p(WC), p(DX), gd
            c.N, c.ldw, c.ldy, c.addend, c.ld_addend, c.ysaved, c.ld_saved, c.dact = C, nkh * nkw * Co, C, p(DX), C, None, 0, 0
        else:
            _lib.call('capmi_igemm_nt', p(DY), p(WC), p(DX), gd, C, nkh * nkw * Co, C, None, p(DX), C, None, 0, None, 0, 0, 0,
                      code[dtype], stream())
    if grouped:         # one call for all classes (a single launch when the shape qualifies)
        _lib.call('capmi_igemm_nt_group', calls, len(classes), code[dtype], stream())
    check(host(DX), base + _nhwc(dx), dtype, name='strided dgrad (accumulate)')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,C,H,W,Co,k,s,pad', [(2, 16, 12, 10, 24, 3, 2, 1), (2, 136, 9, 11, 16, 1, 1, 0), (3, 24, 20, 20, 8, 3, 1, 1),
                                                (2, 32, 8, 8, 16, 1, 2, 0)])
def test_dgrad_with_fused_bn_backward_sums(dtype, B, C, H, W, Co, k, s, pad):
    """capmi_igemm_nt_bnred: the data gradient (+ addend, ReLU mask of the completed tensor) and, from the
    same epilogue, the batch-norm backward sums of TWO layers that read that gradient, finished by
    capmi_bn_bwd_reduce_final == the separate capmi_bn_bwd_reduce semantics (sum dz, sum dz*xhat)."""
    _lib, tdt, code = _env()
    from myimagecaptioningmodel_amd.encoder import dgrad_classes, dgrad_class_offsets
    rng = np.random.RandomState(B * C + Co + k + s)
    x = rnd(rng.standard_normal((B, C, H, W)), dtype)
    w = rnd(rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k), dtype)
    y = O.conv2d_fwd(x, w, s, pad)
    Ho, Wo = y.shape[2:]
    dy = rnd(rng.standard_normal(y.shape), dtype)
    dx, _ = O.conv2d_bwd(dy, x, w, s, pad)
    covered = not (k == 1 and s == 2)
    base = rnd(rng.standard_normal((B, H, W, C)), dtype)
    ysaved = rnd(np.maximum(rng.standard_normal((B, H, W, C)), 0.0), dtype)
    want_dz = (base + _nhwc(dx)) * (ysaved > 0)
    raws = [rnd(rng.standard_normal((B, H, W, C)), dtype) for _ in range(2)]
    means = [rng.standard_normal(C) * 0.3 for _ in range(2)]
    invs = [np.abs(rng.standard_normal(C)) + 0.5 for _ in range(2)]
    DY = dev(_nhwc(dy), tdt[dtype])
    DX = dev(base, tdt[dtype]).clone()
    YS = dev(ysaved, tdt[dtype])
    RX = [dev(r, tdt[dtype]) for r in raws]
    MU = [dev(m, torch.float32) for m in means]
    IS = [dev(v, torch.float32) for v in invs]
    launches = []
    if s == 1:
        launches.append((_lib.ConvGeom(B, Ho, Wo, Co, H, W, k, k, 1, 1, k - 1 - pad, Co), np.flip(w, (2, 3)).transpose(1, 2, 3, 0)))
    else:
        offs = dgrad_class_offsets(k, s, pad)
        for (ph, pw, rmap, qmap) in dgrad_classes(k, s, pad):
            d0h, d0w, nkh, nkw = offs[(ph, pw)]
            wc = np.zeros((C, nkh, nkw, Co))
            for a_, r in enumerate(rmap):
                for b_, q in enumerate(qmap):
                    wc[:, a_, b_, :] = w[:, :, r, q].T
            hc, wcc = (H - ph + s - 1) // s, (W - pw + s - 1) // s
            launches.append((_lib.ConvGeom(B, Ho, Wo, Co, hc, wcc, nkh, nkw, 1, 1, -d0h, Co, s, ph, pw, H, W), wc))
    rows = [(_lib.lib().capmi_igemm_nt_bnred_part_rows(g, C, code[dtype])) for g, _ in launches]
    assert min(rows) > 0
    parts = [(g.B * g.Ho * g.Wo + r - 1) // r for (g, _), r in zip(launches, rows)]
    WS = [torch.full((sum(parts) * 2 * C,), float('nan'), device=DEV) for _ in range(2)]
    _KEEP.extend(WS)
    off = 0
    for (g, wc), np_ in zip(launches, parts):
        _lib.call('capmi_igemm_nt_bnred', p(DY), p(dev(np.ascontiguousarray(wc), tdt[dtype])), p(DX), g, C, g.kh * g.kw * Co, C,
                  p(DX), C, p(YS), C, _lib.ACT_RELU, 2,
                  p(RX[0]), p(MU[0]), p(IS[0]), WS[0].data_ptr() + off * 2 * C * 4,
                  p(RX[1]), p(MU[1]), p(IS[1]), WS[1].data_ptr() + off * 2 * C * 4, code[dtype], stream())
        off += np_
    got_dz = host(DX)
    if covered:
        check(got_dz, want_dz, dtype, name='masked dgrad')
        sel = np.ones((B, H, W, 1), bool)
    else:           # only the class pixels are rewritten (and summed); the others keep `base`
        sel = np.zeros((B, H, W, 1), bool)
        sel[:, ::2, ::2] = True
        check(got_dz, np.where(sel, want_dz, base), dtype, name='masked dgrad (class pixels)')
    dzr = got_dz * sel                 # sums are defined on the values the kernel stored
    for q in range(2):
        red = torch.zeros(2 * C, device=DEV)
        _KEEP.append(red)
        _lib.call('capmi_bn_bwd_reduce_final', p(WS[q]), sum(parts), C, p(red), stream())
        want = np.concatenate([dzr.sum((0, 1, 2)), (dzr * (raws[q] - means[q].astype(np.float32)) * invs[q].astype(np.float32)).sum((0, 1, 2))])
        check(host(red), want, 'f32' if dtype == 'f32' else 'bf16', scale=np.abs(want).max(), name='fused BN sums %d' % q)
        assert np.abs(host(red) - want).max() <= 2e-3 * np.abs(want).max() + 1e-3, 'fused BN sums %d' % q


@pytest.mark.parametrize('B,C,H,W,Co,k,addend', [
    (4, 128, 56, 56, 64, 1, False),      # 98 x 1 tiles of 128 x 128 would under-fill: 64 x 128 tiles, K = 64
    (16, 256, 56, 56, 64, 1, True),      # 128 x 128 LDS-DMA tiles (>= 384 tiles), addend (the shortcut gradient)
    (8, 64, 28, 28, 256, 1, False),      # 64 x 64 tiles
    (8, 128, 28, 28, 128, 3, False),     # halo-staged 3 x 3, 64 x 128 or 128 x 128
    (16, 64, 56, 56, 64, 3, False),      # halo-staged 3 x 3, 64-column tiles
    (8, 512, 14, 14, 2048, 1, True),     # deep K on a small grid: k-groups
    (4, 512, 7, 7, 512, 3, False),       # 7 x 7 3 x 3: k-group kernel with select-based taps
])
def test_dgrad_with_batch_norm_backward_sums_in_its_epilogue(B, C, H, W, Co, k, addend):
    """capmi_igemm_nt_bnsum (EPI 7): the data gradient of a convolution with the completed tensor's ReLU mask as bits, bit-identical
    to capmi_igemm_nt with the same mask, and the batch-norm backward sums of that tensor's layer -- sum dz, sum dz * xhat over the
    STORED values -- in the four accumulator rows (default) or, deterministic mode, added to red in a fixed order."""
    _lib, tdt, code = _env()
    L = _lib.lib()
    pad = (k - 1) // 2
    rng = np.random.RandomState(B + C + Co + k)
    M = B * H * W
    dy = rnd(rng.standard_normal((M, Co)), 'bf16')                                  # gradient w.r.t. the conv output [B,H,W,Co]
    wT = rnd(rng.standard_normal((C, k, k, Co)) / np.sqrt(Co * k * k), 'bf16')      # data-gradient form [Cin][kh][kw][Cout]
    base = rnd(rng.standard_normal((M, C)), 'bf16') if addend else None
    mask = rng.uniform(size=(M, C)) > 0.4
    bits = np.packbits(mask.reshape(-1), bitorder='little')
    raw = rnd(rng.standard_normal((M, C)) * 1.5 + 0.3, 'bf16')
    mean = (rng.standard_normal(C) * 0.3).astype(np.float32)
    inv = (np.abs(rng.standard_normal(C)) + 0.5).astype(np.float32)
    DY, WT = dev(dy, torch.bfloat16), dev(wT, torch.bfloat16)
    BITS = torch.as_tensor(bits).to(DEV)
    RAW, MU, IS = dev(raw, torch.bfloat16), dev(mean, torch.float32), dev(inv, torch.float32)
    _KEEP.append(BITS)
    g = _lib.ConvGeom(B, H, W, Co, H, W, k, k, 1, 1, k - 1 - pad, Co)
    K = k * k * Co
    R = L.capmi_igemm_nt_bnsum_part_rows(g, C, _lib.BF16)
    assert R in (64, 128), R
    dact = _lib.ACT_RELU | _lib.DACT_BITMASK

    def fresh():
        t = dev(base, torch.bfloat16).clone() if addend else torch.zeros((M, C), dtype=torch.bfloat16, device=DEV)
        _KEEP.append(t)
        return t
    ref = fresh()
    _lib.call('capmi_igemm_nt', p(DY), p(WT), p(ref), g, C, K, C, None, p(ref) if addend else None, C, p(BITS), C, None, 0, dact, 0, _lib.BF16, stream())
    dz = host(ref)
    want = np.concatenate([dz.sum(0), (dz * (raw - mean.astype(np.float64)) * inv.astype(np.float64)).sum(0)])
    scale = np.abs(want).max()
    nparts = (M + R - 1) // R
    PARTS = torch.full((nparts * 2 * C,), float('nan'), device=DEV)
    ACC = torch.zeros((4, 2 * C), device=DEV)
    RED = torch.zeros(2 * C, device=DEV)
    _KEEP.extend([PARTS, ACC, RED])
    out = fresh()
    sym = _lib.probe_kernel('capmi_igemm_nt_bnsum', p(DY), p(WT), p(out), g, C, K, C, p(out) if addend else None, C, p(BITS), C, dact,
                            p(RAW), p(MU), p(IS), p(ACC), p(PARTS), p(RED), _lib.BF16)[0]
    assert ', 7>' in sym, sym                     # the sums epilogue class, whatever the kernel family
    _lib.call('capmi_igemm_nt_bnsum', p(DY), p(WT), p(out), g, C, K, C, p(out) if addend else None, C, p(BITS), C, dact,
              p(RAW), p(MU), p(IS), p(ACC), p(PARTS), p(RED), _lib.BF16, stream())
    assert torch.equal(out, ref), 'the data gradient itself must not change'
    got = host(ACC).sum(0)
    assert np.abs(got - want).max() <= 2e-4 * scale + 1e-3, (np.abs(got - want).max(), scale)
    assert float(host(RED).max()) == 0.0           # default mode leaves red to capmi_bn_bwd_apply_spread
    prev = _lib.set_deterministic(True)
    try:
        reds = []
        for _ in range(2):
            out2, red2 = fresh(), torch.zeros(2 * C, device=DEV)
            _KEEP.append(red2)
            _lib.call('capmi_igemm_nt_bnsum', p(DY), p(WT), p(out2), g, C, K, C, p(out2) if addend else None, C, p(BITS), C, dact,
                      p(RAW), p(MU), p(IS), p(ACC), p(PARTS), p(red2), _lib.BF16, stream())
            assert torch.equal(out2, ref)
            reds.append(host(red2))
        np.testing.assert_array_equal(reds[0], reds[1])
        assert np.abs(reds[0] - want).max() <= 2e-4 * scale + 1e-3
    finally:
        _lib.set_deterministic(prev)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(1216, 40, 48), (64, 264, 136), (5000, 16, 24), (333, 1000, 32)])
def test_fc_wgrad_and_colsum(dtype, M, N, K):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(M + N)
    x = rnd(rng.standard_normal((M, K)), dtype)
    ldy = (N + 7) // 8 * 8
    dy = np.zeros((M, ldy))
    dy[:, :N] = rnd(rng.standard_normal((M, N)), dtype)
    X, DY = dev(x, tdt[dtype]), dev(dy, tdt[dtype])
    DW = torch.zeros((N, K), dtype=torch.float32, device=DEV)
    _lib.call('capmi_igemm_tn_wgrad', p(X), p(DY), p(DW), _lib.gemm_geom(M, K), N, ldy, K, p(_lib.wgrad_workspace(DEV)), _lib.WGRAD_WS_BYTES, code[dtype], stream())
    want = dy[:, :N].T @ x
    check(host(DW), want, dtype, name='fc wgrad')
    DB = torch.zeros(N, dtype=torch.float32, device=DEV)
    _lib.call('capmi_colsum', p(DY), M, N, ldy, p(DB), code[dtype], stream())
    check(host(DB), dy[:, :N].sum(0), dtype, scale=np.abs(dy).sum(0).max(), name='colsum')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('stride', [1, 2])
def test_depthwise(dtype, stride):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(stride)
    B, C, H, W = 2, 24, 9, 11
    x = rnd(rng.standard_normal((B, C, H, W)), dtype)
    w = rnd(rng.standard_normal((C, 1, 3, 3)), dtype)
    y = O.conv2d_fwd(x, w, stride, 1, groups=C)
    Ho, Wo = y.shape[2:]
    dy = rnd(rng.standard_normal(y.shape), dtype)
    dx, dw = O.conv2d_bwd(dy, x, w, stride, 1, groups=C)
    X, Wk, DY = dev(_nhwc(x), tdt[dtype]), dev(w[:, 0].transpose(1, 2, 0), tdt[dtype]), dev(_nhwc(dy), tdt[dtype])
    Y = torch.zeros((B, Ho, Wo, C), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_dwconv3x3_fwd', p(X), p(Wk), p(Y), B, H, W, C, stride, Ho, Wo, code[dtype], stream())
    check(host(Y), _nhwc(y), dtype, name='dw fwd')
    DX = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_dwconv3x3_bwd_data', p(DY), p(Wk), p(DX), B, H, W, C, stride, Ho, Wo, 0, code[dtype], stream())
    check(host(DX), _nhwc(dx), dtype, name='dw dgrad')
    DW = torch.zeros((3, 3, C), dtype=torch.float32, device=DEV)
    _lib.call('capmi_dwconv3x3_bwd_weight', p(X), p(DY), p(DW), B, H, W, C, stride, Ho, Wo, code[dtype], stream())
    check(host(DW), dw[:, 0].transpose(1, 2, 0), dtype, name='dw wgrad')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_maxpool_and_stem_im2col(dtype):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(3)
    for B, C, H, W in [(2, 16, 9, 12), (3, 64, 14, 14), (1, 8, 7, 5), (2, 32, 16, 11)]:      # odd / even extents: ragged last window, ragged 2 x 2 block
        x = rnd(rng.standard_normal((B, C, H, W)), dtype)
        y, idx = O.maxpool3x3s2_fwd(x)
        Ho, Wo = y.shape[2:]
        dy = rnd(rng.standard_normal(y.shape), dtype)
        dx = O.maxpool3x3s2_bwd(dy, idx, x.shape)
        X, DY = dev(_nhwc(x), tdt[dtype]), dev(_nhwc(dy), tdt[dtype])
        Y = torch.zeros((B, Ho, Wo, C), dtype=tdt[dtype], device=DEV)
        IDX = torch.zeros((B, Ho, Wo, C), dtype=torch.uint8, device=DEV)
        _lib.call('capmi_maxpool3x3s2_fwd', p(X), p(Y), p(IDX), B, H, W, C, Ho, Wo, code[dtype], stream())
        np.testing.assert_array_equal(host(Y), _nhwc(y))
        np.testing.assert_array_equal(host(IDX), _nhwc(idx))
        DX = torch.full((B, H, W, C), float('nan'), dtype=tdt[dtype], device=DEV)             # every pixel is written
        _lib.call('capmi_maxpool3x3s2_bwd', p(DY), p(IDX), p(DX), B, H, W, C, Ho, Wo, code[dtype], stream())
        check(host(DX), _nhwc(dx), dtype, name='maxpool bwd %dx%d' % (H, W))
    # stem im2col + GEMM == conv on the NCHW feed
    img = rng.uniform(0, 1, (2, 3, 20, 20)).astype(np.float32)
    k, s, pad, Kpad = 7, 2, 3, 160
    w = rnd(rng.standard_normal((8, 3, k, k)) * 0.1, dtype)
    yc = O.conv2d_fwd(rnd(img.astype(np.float64), dtype), w, s, pad)
    Ho, Wo = yc.shape[2:]
    col = torch.zeros((2 * Ho * Wo, Kpad), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_im2col_stem', p(dev(img, torch.float32)), p(col), 2, 3, 20, 20, k, s, pad, Ho, Wo, Kpad, code[dtype], stream())
    wk = np.zeros((8, Kpad)); wk[:, :k * k * 3] = w.transpose(0, 2, 3, 1).reshape(8, -1)
    Yc = torch.zeros((2 * Ho * Wo, 8), dtype=torch.float32, device=DEV)
    _lib.call('capmi_igemm_nt', p(col), p(dev(wk, tdt[dtype])), p(Yc), _lib.gemm_geom(2 * Ho * Wo, Kpad), 8, Kpad, 8, None, None, 0,
              None, 0, None, 0, 0, 1, code[dtype], stream())
    check(host(Yc).reshape(2, Ho, Wo, 8), _nhwc(yc), dtype, name='stem conv')
    # the same conv without a patch matrix: space-to-depth feed + stride-1 implicit GEMM, and its filter gradient
    from myimagecaptioningmodel_amd.params import stem_s2d, to_kernel, to_reference
    kt, Cs = stem_s2d(k, 3)
    Hb, Wb = Ho + (k - 1) // 2, Wo + (k - 1) // 2
    S2D = torch.full((2, Hb, Wb, Cs), 7.0, dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_s2d_stem', p(dev(img, torch.float32)), p(S2D), 2, 3, 20, 20, pad, Hb, Wb, Cs, code[dtype], stream())
    g = _lib.ConvGeom(2, Hb, Wb, Cs, Ho, Wo, kt, kt, 1, 1, 0, Cs)
    ws = to_kernel(w, 'stem')
    assert ws.shape == (8, kt, kt, Cs) and np.array_equal(to_reference(ws, 'stem', w.shape), w)
    Ys = torch.zeros((2 * Ho * Wo, 8), dtype=torch.float32, device=DEV)
    _lib.call('capmi_igemm_nt', p(S2D), p(dev(ws, tdt[dtype])), p(Ys), g, 8, kt * kt * Cs, 8, None, None, 0, None, 0, None, 0, 0, 1,
              code[dtype], stream())
    check(host(Ys).reshape(2, Ho, Wo, 8), _nhwc(yc), dtype, name='space-to-depth stem conv')
    dyc = rnd(rng.standard_normal(yc.shape), dtype)
    _, dw_ref = O.conv2d_bwd(dyc, rnd(img.astype(np.float64), dtype), w, s, pad)
    DW = torch.zeros((8, kt, kt, Cs), dtype=torch.float32, device=DEV)
    WS = torch.zeros(_lib.WGRAD_WS_BYTES // 4, dtype=torch.float32, device=DEV)
    _KEEP.extend([DW, WS])
    _lib.call('capmi_igemm_tn_wgrad', p(S2D), p(dev(_nhwc(dyc), tdt[dtype])), p(DW), g, 8, 8, kt * kt * Cs, p(WS), _lib.WGRAD_WS_BYTES,
              code[dtype], stream())
    _lib.call('capmi_s2d_stem_mask_grad', p(DW), 8, 3, k, Cs, stream())
    got = host(DW)
    check(to_reference(got, 'stem', w.shape), dw_ref, dtype, scale=np.abs(dw_ref).max(), name='stem filter gradient')
    assert np.count_nonzero(got) <= 8 * k * k * 3            # every structural-zero slot is zero


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('act,res', [('relu6', False), (None, True), ('relu', True), ('relu', False)])
def test_batch_norm_chain(dtype, act, res):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(7)
    B, C, H, W = 3, 24, 5, 7
    M = B * H * W
    x = rnd(rng.standard_normal((B, C, H, W)) * 2 + 0.5, dtype)
    r = rnd(rng.standard_normal((B, C, H, W)), dtype) if res else None
    scale, offset = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.2
    rm, rv = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    y, saved, (nm, nv) = O.batch_norm_fwd(x, scale, offset, rm, rv)
    pre = y + (r if res else 0)
    out = O.relu6(pre) if act == 'relu6' else O.relu(pre) if act == 'relu' else pre
    dout = rnd(rng.standard_normal(x.shape), dtype)
    dz = O.relu6_bwd(dout, pre) if act == 'relu6' else O.relu_bwd(dout, out) if act == 'relu' else dout
    dx, dscale, doffset = O.batch_norm_bwd(dz, saved, scale)
    f32 = torch.float32
    X = dev(_nhwc(x), tdt[dtype])
    pr = _lib.lib().capmi_bn_stats_part_rows(M, C, code[dtype])
    stats = torch.full((((M + pr - 1) // pr + 64) * C * 2,), float('nan'), dtype=f32, device=DEV)
    _lib.call('capmi_bn_stats', p(X), M, C, p(stats), code[dtype], stream())
    SC, OF, RM, RV = dev(scale, f32), dev(offset, f32), dev(rm, f32), dev(rv, f32)
    mean, invstd, ca = (torch.zeros(C, dtype=f32, device=DEV) for _ in range(3))
    _lib.call('capmi_bn_finalize', p(stats), pr, M, C, p(SC), p(RM), p(RV), 0.9, 1e-5, p(mean), p(invstd), p(ca), 1, stream())
    check(host(RM), nm, 'f32', name='running mean')
    check(host(RV), nv, 'f32', name='running var')
    R = dev(_nhwc(r), tdt[dtype]) if res else None
    Y = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    ac = _lib.ACT_CODES[act]
    _lib.call('capmi_bn_apply', p(X), p(mean), p(ca), p(OF), p(R), p(Y), M, C, ac, code[dtype], stream())
    check(host(Y), _nhwc(out), dtype, name='bn apply')
    # the one-launch form (statistics merge inside the apply kernel) gives the same output, saved statistics and running stats
    RM2, RV2 = dev(rm, f32), dev(rv, f32)
    mean2, invstd2 = (torch.zeros(C, dtype=f32, device=DEV) for _ in range(2))
    Y2 = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    _KEEP.extend([mean2, invstd2, Y2])
    _lib.call('capmi_bn_finalize_apply', p(stats), pr, M, C, p(SC), p(OF), p(RM2), p(RV2), 0.9, 1e-5, p(mean2), p(invstd2), 1,
              p(X), p(R), p(Y2), ac, code[dtype], stream())
    torch.cuda.synchronize()
    assert torch.allclose(mean2, mean, rtol=1e-6, atol=1e-7) and torch.allclose(invstd2, invstd, rtol=1e-6, atol=0)
    assert torch.allclose(RM2, RM, rtol=1e-6, atol=1e-7) and torch.allclose(RV2, RV, rtol=1e-6, atol=1e-7)
    check(host(Y2), _nhwc(out), dtype, name='bn finalize+apply')
    # backward uses the stored (rounded) output for the activation mask, as the engine does
    Yexact = dev(_nhwc(out), tdt[dtype])
    DY = dev(_nhwc(dout), tdt[dtype])
    red = torch.zeros(2 * C, dtype=f32, device=DEV)
    bws = torch.zeros(_lib.lib().capmi_bn_bwd_ws_floats(M, C, code[dtype]), dtype=f32, device=DEV)
    _lib.call('capmi_bn_bwd_reduce', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(bws), p(red), M, C, ac, code[dtype], stream())
    rr = host(red)
    check(rr[:C], doffset, dtype, scale=np.abs(dz).sum((0, 2, 3)).max(), name='d offset')
    check(rr[C:], dscale, dtype, scale=np.abs(dz).sum((0, 2, 3)).max(), name='d scale')
    DX = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    DR = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV) if res else None
    _lib.call('capmi_bn_bwd_apply', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(SC), p(red), p(DX), 0, p(DR), 0, M, C, ac, code[dtype], stream())
    check(host(DX), _nhwc(dx), dtype, name='bn dx')
    if res:
        check(host(DR), _nhwc(dz), dtype, name='d residual')
    # the form without the second-stage launch: block totals added into eight accumulator rows, summed in the apply prologue
    acc8 = torch.zeros(16 * C, dtype=f32, device=DEV)
    red2 = torch.zeros(2 * C, dtype=f32, device=DEV)
    DX2 = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    DR2 = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV) if res else None
    _KEEP.extend([acc8, red2, DX2, DR2])
    _lib.call('capmi_bn_bwd_reduce_spread', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(bws), p(red2), p(acc8), M, C, ac, code[dtype], stream())
    _lib.call('capmi_bn_bwd_apply_spread', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(SC), p(red2), p(acc8), p(DX2), 0, p(DR2), 0, M, C, ac,
              code[dtype], stream())
    torch.cuda.synchronize()
    sc_ = float(np.abs(dz).sum((0, 2, 3)).max())
    assert torch.allclose(red2, red, rtol=1e-5, atol=1e-5 * sc_), float((red2 - red).abs().max())       # same partial sums, another f32 order
    assert torch.allclose(acc8.view(8, 2 * C).sum(0), red, rtol=1e-5, atol=1e-5 * sc_)
    check(host(DX2), _nhwc(dx), dtype, name='bn dx (spread)')
    assert float((DX2.float() - DX.float()).abs().max()) <= (1e-4 if dtype == 'f32' else 2 ** -6) * max(1.0, float(DX.float().abs().max()))
    if res:
        assert torch.equal(DR2, DR)
    # deterministic mode: the pair IS the two-stage form (bit for bit) and leaves the accumulator rows alone
    prev = _lib.set_deterministic(True)
    try:
        acc8.zero_()
        red3 = torch.zeros(2 * C, dtype=f32, device=DEV)
        DX3 = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
        _KEEP.extend([red3, DX3])
        _lib.call('capmi_bn_bwd_reduce_spread', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(bws), p(red3), p(acc8), M, C, ac, code[dtype], stream())
        _lib.call('capmi_bn_bwd_apply_spread', p(DY), p(X), p(Yexact), p(mean), p(invstd), p(SC), p(red3), p(acc8), p(DX3), 0, None, 0, M, C, ac,
                  code[dtype], stream())
        torch.cuda.synchronize()
        assert torch.equal(red3, red) and torch.equal(DX3, DX) and not bool(acc8.any())
    finally:
        _lib.set_deterministic(prev)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_batch_norm_backward_gathered_from_a_max_pool_gradient(dtype):
    """capmi_bn_bwd_reduce_pool / capmi_bn_bwd_apply_pool == capmi_maxpool3x3s2_bwd + the batch-norm backward pair on the
    materialised gradient (oracle chain: ops.maxpool3x3s2_bwd -> ops.batch_norm_bwd behind a relu), for even / odd extents."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(11)
    f32 = torch.float32
    for B, C, H, W in [(2, 64, 16, 16), (3, 32, 9, 13), (1, 16, 7, 8), (2, 64, 12, 11)]:
        M = B * H * W
        x = rnd(rng.standard_normal((B, C, H, W)) * 1.5 + 0.3, dtype)
        scale = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32)
        offset = (0.2 * rng.standard_normal(C)).astype(np.float32)
        x64 = x.astype(np.float64)
        out, saved, _ = O.batch_norm_fwd(x64, scale.astype(np.float64), offset.astype(np.float64), np.zeros(C), np.ones(C))
        yact = rnd(np.maximum(out, 0.0), dtype)
        pooled, idx = O.maxpool3x3s2_fwd(yact)
        Ho, Wo = pooled.shape[2:]
        dpool = rnd(rng.standard_normal(pooled.shape), dtype)
        mean = dev(x64.mean((0, 2, 3)), f32)
        invstd = dev(saved[1], f32)
        X, Y = dev(_nhwc(x), tdt[dtype]), dev(_nhwc(yact), tdt[dtype])
        DP = dev(_nhwc(dpool), tdt[dtype])
        IDX = dev(_nhwc(idx), torch.uint8)
        SC = dev(scale, f32)
        ac = _lib.ACT_CODES['relu']
        # reference: three launches on the materialised gradient
        DYP = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_maxpool3x3s2_bwd', p(DP), p(IDX), p(DYP), B, H, W, C, Ho, Wo, code[dtype], stream())
        red = torch.zeros(2 * C, dtype=f32, device=DEV)
        bws = torch.zeros(_lib.lib().capmi_bn_bwd_ws_floats(M, C, code[dtype]), dtype=f32, device=DEV)
        DX = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_bn_bwd_reduce', p(DYP), p(X), p(Y), p(mean), p(invstd), p(bws), p(red), M, C, ac, code[dtype], stream())
        _lib.call('capmi_bn_bwd_apply', p(DYP), p(X), p(Y), p(mean), p(invstd), p(SC), p(red), p(DX), 0, None, 0, M, C, ac, code[dtype], stream())
        # oracle chain on the same operands
        dyp = O.maxpool3x3s2_bwd(dpool, idx, x.shape)
        dz = dyp.astype(np.float64) * (yact > 0)
        dx_o, _, _ = O.batch_norm_bwd(dz, saved, scale.astype(np.float64))
        check(host(DX), _nhwc(dx_o), dtype, name='pool + bn dx (three launches)')
        # the gathered pair
        acc8 = torch.zeros(16 * C, dtype=f32, device=DEV)
        red2 = torch.zeros(2 * C, dtype=f32, device=DEV)
        scratch = torch.full((B, H, W, C), float('nan'), dtype=tdt[dtype], device=DEV)
        DX2 = torch.full((B, H, W, C), float('nan'), dtype=tdt[dtype], device=DEV)
        _KEEP.extend([DYP, red, bws, DX, acc8, red2, scratch, DX2])
        args = (B, H, W, C, Ho, Wo, ac, code[dtype], stream())
        _lib.call('capmi_bn_bwd_reduce_pool', p(DP), p(IDX), p(X), p(Y), p(mean), p(invstd), p(bws), p(red2), p(acc8), p(scratch), *args)
        _lib.call('capmi_bn_bwd_apply_pool', p(DP), p(IDX), p(X), p(Y), p(mean), p(invstd), p(SC), p(red2), p(acc8), p(scratch), p(DX2), *args)
        torch.cuda.synchronize()
        sc_ = float(np.abs(dz).sum((0, 2, 3)).max())
        assert torch.allclose(red2, red, rtol=1e-5, atol=1e-5 * sc_), float((red2 - red).abs().max())
        assert bool(torch.isnan(scratch.float()).all())           # the pool's input gradient is never written
        check(host(DX2), _nhwc(dx_o), dtype, name='pool + bn dx (gathered)')
        assert float((DX2.float() - DX.float()).abs().max()) <= (1e-4 if dtype == 'f32' else 2 ** -6) * max(1.0, float(DX.float().abs().max()))
        # deterministic mode: the pair IS the three-launch path, bit for bit
        prev = _lib.set_deterministic(True)
        try:
            red3 = torch.zeros(2 * C, dtype=f32, device=DEV)
            DX3 = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
            acc8.zero_()
            _KEEP.extend([red3, DX3])
            _lib.call('capmi_bn_bwd_reduce_pool', p(DP), p(IDX), p(X), p(Y), p(mean), p(invstd), p(bws), p(red3), p(acc8), p(scratch), *args)
            _lib.call('capmi_bn_bwd_apply_pool', p(DP), p(IDX), p(X), p(Y), p(mean), p(invstd), p(SC), p(red3), p(acc8), p(scratch), p(DX3), *args)
            torch.cuda.synchronize()
            assert torch.equal(red3, red) and torch.equal(DX3, DX) and torch.equal(scratch, DYP) and not bool(acc8.any())
        finally:
            _lib.set_deterministic(prev)


def test_batch_norm_statistics_no_cancellation():
    """mean >> std: a single-pass E[x^2]-E[x]^2 in f32 loses the variance entirely here."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(0)
    M, C = 5000, 32
    x = (100.0 + 1e-2 * rng.standard_normal((M, C))).astype(np.float32)
    X = dev(x, torch.float32)
    f32 = torch.float32
    pr = _lib.lib().capmi_bn_stats_part_rows(M, C, _lib.F32)
    ws = torch.zeros((((M + pr - 1) // pr + 64) * C * 2,), dtype=f32, device=DEV)
    _lib.call('capmi_bn_stats', p(X), M, C, p(ws), _lib.F32, stream())
    ones, zeros = torch.ones(C, dtype=f32, device=DEV), torch.zeros(C, dtype=f32, device=DEV)
    mean, invstd, ca = (torch.zeros(C, dtype=f32, device=DEV) for _ in range(3))
    _lib.call('capmi_bn_finalize', p(ws), pr, M, C, p(ones), None, None, 0.9, 1e-5, p(mean), p(invstd), p(ca), 0, stream())
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(host(mean), x64.mean(0), rtol=1e-6)
    np.testing.assert_allclose(host(invstd), 1 / np.sqrt(x64.var(0) + 1e-5), rtol=1e-4)
    Yn = torch.zeros((M, C), dtype=f32, device=DEV)
    _lib.call('capmi_bn_apply', p(X), p(mean), p(ca), p(zeros), None, p(Yn), M, C, 0, _lib.F32, stream())
    want = (x64 - x64.mean(0)) / np.sqrt(x64.var(0) + 1e-5)
    assert np.abs(host(Yn) - want).max() < 2e-3          # x itself carries only ~4 digits below the mean
    # the same through the GEMM epilogue: x = A . I
    eye = torch.eye(C, dtype=f32, device=DEV)
    Y = torch.zeros((M, C), dtype=f32, device=DEV)
    pr2 = _lib.lib().capmi_igemm_nt_stats_part_rows(M, C, C, _lib.F32)
    ws2 = torch.zeros((((M + pr2 - 1) // pr2 + 64) * C * 2,), dtype=f32, device=DEV)
    _lib.call('capmi_igemm_nt', p(X), p(eye), p(Y), _lib.gemm_geom(M, C), C, C, C, None, None, 0, None, 0, p(ws2), 0, 0, 0, _lib.F32, stream())
    _lib.call('capmi_bn_finalize', p(ws2), pr2, M, C, p(ones), None, None, 0.9, 1e-5, p(mean), p(invstd), p(ca), 0, stream())
    np.testing.assert_allclose(host(invstd), 1 / np.sqrt(x64.var(0) + 1e-5), rtol=1e-4)


@pytest.mark.parametrize('M,part_rows,C', [(65 * 64, 64, 24), (12544, 128, 1024), (50176, 128, 136), (200704 + 37, 64, 64), (802816, 256, 64),
                                           (3136 * 64, 64, 2048), (64 * 64, 64, 64)])
def test_bn_finalize_merges_many_parts_in_one_launch(M, part_rows, C):
    """capmi_bn_finalize over more than 64 statistic parts: the merge level and the finalize are ONE launch whose
    last-arriving workgroup finalizes (bn_merge_finalize_kernel).  Against an f64 Chan merge of the same parts; four
    launches back to back without a synchronisation in between (the arrival counters reset themselves, every set is
    reused), and the running statistics must have moved exactly once per launch (a second 'last' workgroup would move
    them twice)."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(M % 1000 + C)
    nparts = (M + part_rows - 1) // part_rows
    rows = np.minimum(part_rows, M - np.arange(nparts) * part_rows).astype(np.float64)
    pmean = rng.standard_normal((nparts, C)) * 0.3 + rng.standard_normal(C) * 2
    pm2 = rng.uniform(0.5, 1.5, (nparts, C)) * rows[:, None]
    pmean32, pm232 = pmean.astype(np.float32), pm2.astype(np.float32)
    n = rows[:, None]
    mean = (pmean32.astype(np.float64) * n).sum(0) / M
    m2 = (pm232.astype(np.float64) + n * (pmean32.astype(np.float64) - mean) ** 2).sum(0)
    var = m2 / M
    f32 = torch.float32
    ws = torch.full((nparts + 64, C, 2), float('nan'), dtype=f32, device=DEV)
    ws[:nparts] = dev(np.stack([pmean32, pm232], -1), f32)
    scale = rng.uniform(0.5, 1.5, C)
    rm0, rv0 = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    SC, RM, RV = dev(scale, f32), dev(rm0, f32), dev(rv0, f32)
    outs = [[torch.zeros(C, dtype=f32, device=DEV) for _ in range(3)] for _ in range(4)]
    for o in outs:
        _lib.call('capmi_bn_finalize', p(ws), part_rows, M, C, p(SC), p(RM), p(RV), 0.9, 1e-5, p(o[0]), p(o[1]), p(o[2]), 1, stream())
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_allclose(host(o[0]), mean, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(host(o[1]), 1 / np.sqrt(var + 1e-5), rtol=2e-6)
        np.testing.assert_allclose(host(o[2]), scale / np.sqrt(var + 1e-5), rtol=2e-6)
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])          # fixed fold order: bit-reproducible
    want_rm, want_rv = rm0.copy(), rv0.copy()
    for _ in range(4):
        want_rm, want_rv = want_rm * 0.9 + mean * 0.1, want_rv * 0.9 + var * 0.1
    np.testing.assert_allclose(host(RM), want_rm, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(host(RV), want_rv, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,C,H,W,Co,k,s,pad,act,res', [(2, 16, 9, 9, 24, 3, 1, 1, 'relu', False), (3, 32, 8, 8, 16, 1, 1, 0, None, True),
                                                         (2, 64, 14, 14, 64, 3, 1, 1, 'relu', True), (2, 16, 8, 8, 32, 1, 2, 0, 'relu6', False),
                                                         (8, 256, 14, 14, 1024, 1, 1, 0, 'relu', True)])
def test_conv_with_inference_batch_norm_in_the_epilogue(dtype, B, C, H, W, Co, k, s, pad, act, res):
    """capmi_igemm_nt_bn: conv -> batch_norm(is_test) -> (+ residual) -> activation in one launch, against the oracle's
    conv + running-statistics normalisation, and against capmi_igemm_nt + capmi_bn_apply (f32: the same formula on the
    same accumulator; bf16: one rounding fewer)."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(B * C + Co + k)
    x = rnd(rng.standard_normal((B, C, H, W)), dtype)
    w = rnd(rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k), dtype)
    y = O.conv2d_fwd(x, w, s, pad)
    Ho, Wo = y.shape[2], y.shape[3]
    scale, offset = rng.uniform(0.5, 1.5, Co), rng.standard_normal(Co) * 0.2
    rm, rv = rng.standard_normal(Co) * 0.3, rng.uniform(0.5, 2, Co)
    r = rnd(rng.standard_normal(y.shape), dtype) if res else None
    a = scale / np.sqrt(rv + 1e-5)
    pre = a[None, :, None, None] * (y - rm[None, :, None, None]) + offset[None, :, None, None] + (r if res else 0)
    want = O.relu6(pre) if act == 'relu6' else O.relu(pre) if act == 'relu' else pre
    f32 = torch.float32
    X = dev(_nhwc(x), tdt[dtype])
    Wk = dev(w.transpose(0, 2, 3, 1), tdt[dtype])
    SC, OF, RM, RV = dev(scale, f32), dev(offset, f32), dev(rm, f32), dev(rv, f32)
    mean, ca = torch.zeros(Co, dtype=f32, device=DEV), torch.zeros(Co, dtype=f32, device=DEV)
    _lib.call('capmi_bn_inference_coef', p(SC), p(RM), p(RV), 1e-5, p(mean), p(ca), Co, stream())
    R = dev(_nhwc(r), tdt[dtype]) if res else None
    Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
    g = _lib.ConvGeom(B, H, W, C, Ho, Wo, k, k, s, 1, pad, C)
    ac = _lib.ACT_CODES[act]
    _lib.call('capmi_igemm_nt_bn', p(X), p(Wk), p(Y), g, Co, k * k * C, Co, p(mean), p(ca), p(OF), p(R), Co, ac, code[dtype], stream())
    check(host(Y), _nhwc(want), dtype, name='conv + inference bn')
    RAW = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
    Y2 = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_igemm_nt', p(X), p(Wk), p(RAW), g, Co, k * k * C, Co, None, None, 0, None, 0, None, 0, 0, 0, code[dtype], stream())
    _lib.call('capmi_bn_apply', p(RAW), p(mean), p(ca), p(OF), p(R), p(Y2), B * Ho * Wo, Co, ac, code[dtype], stream())
    torch.cuda.synchronize()
    d = (Y.float() - Y2.float()).abs().max()
    tol = 1e-5 if dtype == 'f32' else 2 ** -6
    assert float(d) <= tol * max(1.0, float(Y2.float().abs().max())), float(d)


@pytest.mark.parametrize('act,res', [('relu', False), ('relu', True), ('relu6', False)])
@pytest.mark.parametrize('B,C,H,W,Cn,k', [(4, 64, 56, 56, 256, 1),       # 128 x 128 tiles
                                            (8, 256, 14, 14, 64, 1),       # 64 x 64 tiles
                                            (8, 256, 56, 56, 64, 1),       # 128 x 64 tiles (tall grid)
                                            (8, 128, 28, 28, 128, 3),      # halo kernel
                                            (16, 512, 7, 7, 512, 3),       # k-group kernel
                                            (3, 40, 9, 11, 72, 3),         # ragged rows, N = 72
                                            (64, 1024, 14, 14, 256, 1)])   # 128 x 128 k-group tiles
def test_activation_bit_mask_in_the_data_gradient_epilogue(act, res, B, C, H, W, Cn, k):
    """capmi_bn_apply_mask writes y (== capmi_bn_apply, bit for bit) and one bit per element, act'(y) == 1 at the STORED output;
    a data-gradient launch that takes the bit mask (dact | CAPMI_DACT_BITMASK) must equal the launch that reads y itself, bit
    for bit (conv2d_grad under a ReLU: the mask of MobileNetV2.py:112-121's activation), with and without an addend."""
    _lib, tdt, code = _env()
    dtype = 'bf16'
    rng = np.random.RandomState(C + Cn + k + B)
    f32 = torch.float32
    M = B * H * W
    raw = dev(rng.standard_normal((B, H, W, Cn)) * 1.3 + 0.2, tdt[dtype])
    mean, ca = dev(rng.standard_normal(Cn) * 0.2, f32), dev(rng.uniform(0.5, 1.5, Cn), f32)
    off = dev(rng.standard_normal(Cn) * (2.0 if act == 'relu6' else 0.3) + (3.0 if act == 'relu6' else 0.0), f32)
    R = dev(rng.standard_normal((B, H, W, Cn)), tdt[dtype]) if res else None
    ac = _lib.ACT_CODES[act]
    Y0 = torch.zeros((B, H, W, Cn), dtype=tdt[dtype], device=DEV)
    Y1 = torch.zeros((B, H, W, Cn), dtype=tdt[dtype], device=DEV)
    bits = torch.zeros(M * Cn // 8, dtype=torch.uint8, device=DEV)
    _lib.call('capmi_bn_apply', p(raw), p(mean), p(ca), p(off), p(R), p(Y0), M, Cn, ac, code[dtype], stream())
    _lib.call('capmi_bn_apply_mask', p(raw), p(mean), p(ca), p(off), p(R), p(Y1), p(bits), M, Cn, ac, code[dtype], stream())
    torch.cuda.synchronize()
    assert torch.equal(Y0, Y1)
    yf = Y1.float().reshape(M, Cn)
    want = (yf > 0) if act == 'relu' else ((yf > 0) & (yf < 6))
    assert 0.05 < float(want.float().mean()) < 0.95          # both values of the bit occur
    w8 = (want.reshape(M, Cn // 8, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits.reshape(M, Cn // 8), w8)
    pad = 1 if k == 3 else 0
    dy = dev(rng.standard_normal((B, H, W, C)), tdt[dtype])                         # gradient of the conv's output (C channels)
    wt = dev(rng.standard_normal((Cn, k, k, C)) / np.sqrt(C * k * k), tdt[dtype])   # data-gradient weight form [Cn][kh][kw][C]
    gd = _lib.ConvGeom(B, H, W, C, H, W, k, k, 1, 1, k - 1 - pad, C)
    K = k * k * C
    for with_addend in (False, True):
        A = dev(rng.standard_normal((B, H, W, Cn)), tdt[dtype]) if with_addend else None
        D0 = torch.full((B, H, W, Cn), float('nan'), dtype=tdt[dtype], device=DEV)
        D1 = torch.full((B, H, W, Cn), float('nan'), dtype=tdt[dtype], device=DEV)
        _KEEP.extend([D0, D1, Y0, Y1, bits])
        _lib.call('capmi_igemm_nt', p(dy), p(wt), p(D0), gd, Cn, K, Cn, None, p(A), Cn, p(Y1), Cn, None, 0, ac, 0, code[dtype], stream())
        _lib.call('capmi_igemm_nt', p(dy), p(wt), p(D1), gd, Cn, K, Cn, None, p(A), Cn, p(bits), Cn, None, 0, ac | _lib.DACT_BITMASK, 0, code[dtype], stream())
        torch.cuda.synchronize()
        assert torch.equal(D0, D1), int((D0 != D1).sum())
        assert bool(((D0.float().reshape(M, Cn) == 0) | want).all())             # masked elements are exactly zero


@pytest.mark.parametrize('B,C,H,W,Co,k,s', [(4, 64, 56, 56, 256, 1, 1),       # 128 x 128 tiles, K = 64
                                             (4, 256, 56, 56, 64, 1, 1),       # 128 x 64 tiles (tall grid, narrow output)
                                             (8, 128, 28, 28, 128, 3, 1),      # halo kernel
                                             (64, 1024, 14, 14, 256, 1, 1),    # k-group kernel
                                             (8, 256, 28, 28, 512, 1, 2),      # strided 1 x 1 (projection shortcut)
                                             (16, 512, 7, 7, 512, 3, 1)])      # 7 x 7: 64 x 64 k-group tiles
def test_epilogue_classes_equal_the_general_epilogue(B, C, H, W, Co, k, s):
    """The NT kernels carry their epilogue as a class chosen per launch (training convolution forward: stores + statistics;
    data gradient: addend + ReLU mask; decoder fc: bias / tanh; inference: batch norm + ReLU; f32 logits) -- the same arithmetic
    with the paths the launch cannot take compiled out (DESIGN.md lesson 54).  Every class must reproduce the general epilogue
    (capmi_set_general_epilogue) BIT FOR BIT."""
    _lib, tdt, code = _env()
    dtype = 'bf16'
    rng = np.random.RandomState(B + C + Co + k + s)
    f32 = torch.float32
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    M, K = B * Ho * Wo, k * k * C
    X = dev(rng.standard_normal((B, H, W, C)) + 0.3, tdt[dtype])
    Wk = dev(rng.standard_normal((Co, k, k, C)) / np.sqrt(K), tdt[dtype])
    g = _lib.ConvGeom(B, H, W, C, Ho, Wo, k, k, s, 1, pad, C)
    pr = _lib.lib().capmi_igemm_nt_stats_part_rows(M, Co, K, code[dtype])
    nparts = (M + pr - 1) // pr
    bias = dev(rng.standard_normal(Co) * 0.2, f32)
    mean, ca = dev(rng.standard_normal(Co) * 0.1, f32), dev(rng.uniform(0.5, 1.5, Co), f32)
    RES = dev(rng.standard_normal((B, Ho, Wo, Co)), tdt[dtype])
    YS = dev(rng.standard_normal((B, Ho, Wo, Co)), tdt[dtype])          # a saved activation output (mask / tanh source)
    relu, tanh = _lib.ACT_CODES['relu'], _lib.ACT_CODES['tanh']

    def run():
        outs = []
        # 1: training convolution, forward form (statistics)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        st = torch.zeros((nparts + 64, Co, 2), dtype=f32, device=DEV)
        _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, None, None, 0, None, 0, p(st), 0, 0, 0, code[dtype], stream())
        outs += [Y, st[:nparts]]
        # 4: data-gradient form (addend + ReLU mask), on the same geometry
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, None, p(RES), Co, p(YS), Co, None, 0, relu, 0, code[dtype], stream())
        outs.append(Y)
        # 2: fc form (bias + tanh; tanh derivative with an addend)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, p(bias), None, 0, None, 0, None, tanh, 0, 0, code[dtype], stream())
        outs.append(Y)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, None, p(RES), Co, p(YS), Co, None, 0, tanh, 0, code[dtype], stream())
        outs.append(Y)
        # 3: inference form (batch norm on the accumulator + residual + ReLU)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_igemm_nt_bn', p(X), p(Wk), p(Y), g, Co, K, Co, p(mean), p(ca), p(bias), p(RES), Co, relu, code[dtype], stream())
        outs.append(Y)
        # 5: f32 output with a bias (logits)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=f32, device=DEV)
        _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, p(bias), None, 0, None, 0, None, 0, 0, 1, code[dtype], stream())
        outs.append(Y)
        torch.cuda.synchronize()
        _KEEP.extend(outs)
        return outs

    prev = _lib.set_general_epilogue(False)
    try:
        cls = run()
        _lib.set_general_epilogue(True)
        gen = run()
    finally:
        _lib.set_general_epilogue(prev)
    names = ['conv forward', 'its statistics', 'data gradient (addend, relu mask)', 'fc (bias, tanh)', 'fc gradient (addend, tanh derivative)',
             'inference conv + batch norm', 'f32 logits']
    for n_, a_, b_ in zip(names, cls, gen):
        assert torch.equal(a_, b_), (n_, int((a_ != b_).sum()))
    assert float(cls[0].float().abs().max()) > 0 and float(cls[2].float().abs().max()) > 0


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('B,C,H,W,Co,k,s', [(64, 1024, 14, 14, 256, 1, 1),     # k-group kernel, 128 x 128 tiles, 98 parts: two levels
                                             (64, 256, 14, 14, 256, 3, 1),      # halo kernel, 196 workgroups, two levels
                                             (64, 256, 14, 14, 1024, 1, 1),     # 64 x 128 tiles, 1568 workgroups: the two-call form (grid too large)
                                             (64, 512, 7, 7, 512, 3, 1),        # 7 x 7: 64 x 64 k-group tiles, 49 parts: ONE level
                                             (64, 2048, 7, 7, 512, 1, 1),
                                             (16, 512, 7, 7, 2048, 1, 1),       # 13 parts
                                             (64, 1024, 14, 14, 2048, 1, 2),    # strided projection
                                             (5, 96, 10, 7, 40, 1, 1),          # ragged M and N, one 64 x 64 column tile
                                             (9, 32, 21, 19, 200, 3, 1),        # ragged N over two 128-wide column tiles, 57 parts
                                             (33, 64, 14, 14, 72, 1, 1)])       # 102 parts of 64 rows, N = 72: two levels, ragged column tile
def test_conv_statistics_finalize_in_one_launch(dtype, B, C, H, W, Co, k, s):
    """capmi_igemm_nt_bnfin (conv2d -> batch_norm statistics of MobileNetV2.py:88-121, train mode): the last-arriving
    workgroup of the convolution merges and finalizes.  Must equal capmi_igemm_nt + capmi_bn_finalize BIT FOR BIT -- output,
    saved mean / invstd / coef_a and running statistics -- on three launches back to back (the arrival counters reset
    themselves; a second 'last' workgroup would move the running statistics twice)."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(B + C + Co + k)
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = rnd(rng.standard_normal((B, H, W, C)) + 0.5, dtype)
    w = rnd(rng.standard_normal((Co, k, k, C)) / np.sqrt(C * k * k), dtype)
    f32 = torch.float32
    X, Wk = dev(x, tdt[dtype]), dev(w, tdt[dtype])
    g = _lib.ConvGeom(B, H, W, C, Ho, Wo, k, k, s, 1, pad, C)
    M, K = B * Ho * Wo, k * k * C
    pr = _lib.lib().capmi_igemm_nt_stats_part_rows(M, Co, K, code[dtype])
    nparts = (M + pr - 1) // pr
    SC = dev(rng.uniform(0.5, 1.5, Co), f32)
    rm0, rv0 = rng.standard_normal(Co), rng.uniform(0.5, 2, Co)

    def run(fused):
        st = torch.zeros((nparts + 64, Co, 2), dtype=f32, device=DEV)
        Y = torch.zeros((B, Ho, Wo, Co), dtype=tdt[dtype], device=DEV)
        RM, RV = dev(rm0, f32), dev(rv0, f32)
        outs = []
        for _ in range(3):
            o = [torch.zeros(Co, dtype=f32, device=DEV) for _ in range(3)]
            outs.append(o)
            if fused:
                _lib.call('capmi_igemm_nt_bnfin', p(X), p(Wk), p(Y), g, Co, K, Co, p(st), p(SC), p(RM), p(RV), 0.9, 1e-5, p(o[0]), p(o[1]), p(o[2]), 1,
                          code[dtype], stream())
            else:
                _lib.call('capmi_igemm_nt', p(X), p(Wk), p(Y), g, Co, K, Co, None, None, 0, None, 0, p(st), 0, 0, 0, code[dtype], stream())
                _lib.call('capmi_bn_finalize', p(st), pr, M, Co, p(SC), p(RM), p(RV), 0.9, 1e-5, p(o[0]), p(o[1]), p(o[2]), 1, stream())
        torch.cuda.synchronize()
        _KEEP.extend([st, Y, RM, RV] + [t for o in outs for t in o])
        return Y, st, RM, RV, outs

    Yf, stf, RMf, RVf, of = run(True)
    Yr, str_, RMr, RVr, orf = run(False)
    assert torch.equal(Yf, Yr) and torch.equal(stf[:nparts], str_[:nparts])
    for a_, b_ in zip(of, orf):
        for q in range(3):
            assert torch.equal(a_[q], b_[q]), ('launch output %d differs' % q, float((a_[q] - b_[q]).abs().max()))
    assert torch.equal(RMf, RMr) and torch.equal(RVf, RVr)
    # and against the plain statistics of the stored output
    y64 = host(Yr).astype(np.float64).reshape(M, Co)
    np.testing.assert_allclose(host(of[0][0]), y64.mean(0), rtol=0, atol=(2e-2 if dtype == 'bf16' else 1e-4))


@pytest.mark.parametrize('B,C,H,W,Co,k,act', [(4, 64, 56, 56, 64, 3, 'relu'),      # halo kernel, 64 x 64 tiles, two channel chunks
                                               (8, 128, 28, 28, 128, 3, 'relu'),    # halo kernel, 128-wide tiles
                                               (3, 32, 9, 11, 48, 3, 'relu6'),      # ragged rows, one chunk, relu6
                                               (2, 256, 14, 14, 256, 3, 'relu'),
                                               (4, 64, 56, 56, 256, 1, 'relu'),     # 1x1, K = 64 (two k-steps), 64 x 128 tiles
                                               (16, 512, 7, 7, 2048, 1, 'relu'),    # 1x1, K = 512: the whole coefficient table
                                               (64, 512, 7, 7, 2048, 1, 'relu'),    # the same at batch 64: 128 x 128 tiles
                                               (5, 96, 10, 7, 40, 1, 'relu6'),      # 1x1, ragged M / N, 64 x 64 tiles
                                               (8, 256, 14, 14, 1024, 1, 'relu')])
def test_conv_with_batch_norm_in_the_operand_path(B, C, H, W, Co, k, act):
    """capmi_igemm_nt_bnact (MobileNetV2.py:88-121: conv -> batch_norm -> relu -> conv): the consumer convolution reads the
    producer's RAW output and applies its train-mode batch norm + activation in the A-operand path.  Against (i) the
    oracle's batch_norm_fwd + activation + conv2d_fwd on the same raw tensor, and (ii) capmi_bn_apply followed by
    capmi_igemm_nt -- the materialised path it replaces -- BIT FOR BIT, output and fused batch statistics alike."""
    _lib, tdt, code = _env()
    dtype = 'bf16'
    rng = np.random.RandomState(B * C + Co + k)
    raw = rnd(rng.standard_normal((B, C, H, W)) * rng.uniform(0.5, 2, (1, C, 1, 1)) + rng.standard_normal((1, C, 1, 1)), dtype)
    w = rnd(rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k), dtype)
    scale, offset = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3
    pad = 1 if k == 3 else 0
    f32 = torch.float32
    g = _lib.ConvGeom(B, H, W, C, H, W, k, k, 1, 1, pad, C)
    kind = _lib.lib().capmi_igemm_nt_bnact_supported(g, Co, code[dtype])
    assert kind == (1 if k == 3 else 2), kind
    # the producer's statistics through the product path: bn_stats -> bn_finalize (mean, coef_a = scale * invstd)
    M = B * H * W
    RAW = dev(_nhwc(raw), tdt[dtype])
    pr = _lib.lib().capmi_bn_stats_part_rows(M, C, code[dtype])
    ws = torch.zeros(((M + pr - 1) // pr + 64, C, 2), dtype=f32, device=DEV)
    SC, OF = dev(scale, f32), dev(offset, f32)
    rm, rv = torch.zeros(C, dtype=f32, device=DEV), torch.ones(C, dtype=f32, device=DEV)
    mean, invstd, ca = (torch.zeros(C, dtype=f32, device=DEV) for _ in range(3))
    _lib.call('capmi_bn_stats', p(RAW), M, C, p(ws), code[dtype], stream())
    _lib.call('capmi_bn_finalize', p(ws), pr, M, C, p(SC), p(rm), p(rv), 0.9, 1e-5, p(mean), p(invstd), p(ca), 1, stream())
    Wk = dev(w.transpose(0, 2, 3, 1), tdt[dtype])
    K = k * k * C
    pr2 = _lib.lib().capmi_igemm_nt_stats_part_rows(M, Co, K, code[dtype])
    np2 = (M + pr2 - 1) // pr2
    st_f = torch.zeros((np2 + 64, Co, 2), dtype=f32, device=DEV)
    st_m = torch.zeros((np2 + 64, Co, 2), dtype=f32, device=DEV)
    Yf = torch.zeros((B, H, W, Co), dtype=tdt[dtype], device=DEV)
    Ym = torch.zeros((B, H, W, Co), dtype=tdt[dtype], device=DEV)
    ACTT = torch.zeros((B, H, W, C), dtype=tdt[dtype], device=DEV)
    ac = _lib.ACT_CODES[act]
    _lib.call('capmi_igemm_nt_bnact', p(RAW), p(Wk), p(Yf), g, Co, K, Co, p(mean), p(ca), p(OF), ac, p(st_f), code[dtype], stream())
    _lib.call('capmi_bn_apply', p(RAW), p(mean), p(ca), p(OF), None, p(ACTT), M, C, ac, code[dtype], stream())
    _lib.call('capmi_igemm_nt', p(ACTT), p(Wk), p(Ym), g, Co, K, Co, None, None, 0, None, 0, p(st_m), 0, 0, 0, code[dtype], stream())
    torch.cuda.synchronize()
    assert torch.equal(Yf, Ym), 'fused operand path differs from bn_apply + conv in %d elements' % int((Yf != Ym).sum())
    assert torch.equal(st_f[:np2], st_m[:np2])
    # oracle: train-mode batch norm of the raw tensor -> activation -> (rounded to the storage type, as the product stores it) -> conv
    yb, _, _ = O.batch_norm_fwd(raw, scale, offset, np.zeros(C), np.ones(C))
    a_ = rnd(O.relu6(yb) if act == 'relu6' else O.relu(yb), dtype)
    want = O.conv2d_fwd(a_, w, 1, pad)
    got = host(Yf)
    err = np.linalg.norm(got - _nhwc(want)) / np.linalg.norm(want)
    assert err <= 1.5e-2, err        # bf16 output rounding (2^-9 relative per element) + the activations' own rounding flips
    check(got, _nhwc(want), dtype, name='conv on bn(raw)')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_lstm_cell_sentinel_embedding(dtype):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(11)
    B, H, E, V = 5, 32, 16, 50
    gates = rnd(rng.standard_normal((B, 4 * H)), dtype)
    c_prev = rnd(rng.standard_normal((B, H)), dtype)
    i, f, o, g = (O.sigmoid(gates[:, :H]), O.sigmoid(gates[:, H:2 * H]), O.sigmoid(gates[:, 2 * H:3 * H]), np.tanh(gates[:, 3 * H:]))
    c = f * c_prev + i * g
    h = o * np.tanh(c)
    G, CP = dev(gates, tdt[dtype]), dev(c_prev, tdt[dtype])
    Hn, Cn = (torch.zeros((B, H), dtype=tdt[dtype], device=DEV) for _ in range(2))
    _lib.call('capmi_lstm_cell_fwd', p(G), p(CP), p(Hn), p(Cn), B, H, code[dtype], stream())
    check(host(Hn), h, dtype, name='lstm h')
    check(host(Cn), c, dtype, name='lstm c')
    dh, dc = rnd(rng.standard_normal((B, H)), dtype), rnd(rng.standard_normal((B, H)), dtype)
    cr = rnd(c, dtype)
    tc = np.tanh(cr)
    dct = dc + dh * o * (1 - tc ** 2)
    want = np.concatenate([dct * g * i * (1 - i), dct * c_prev * f * (1 - f), dh * tc * o * (1 - o), dct * i * (1 - g * g)], 1)
    DG = torch.zeros((B, 4 * H), dtype=tdt[dtype], device=DEV)
    old = rnd(rng.standard_normal((B, H)), dtype)
    DCP = dev(old, tdt[dtype])
    _lib.call('capmi_lstm_cell_bwd', p(G), p(CP), p(dev(cr, tdt[dtype])), p(dev(dh, tdt[dtype])), p(dev(dc, tdt[dtype])), p(DG), p(DCP), 1,
              B, H, code[dtype], stream())
    check(host(DG), want, dtype, name='lstm dgates')
    check(host(DCP), old + dct * f, dtype, name='lstm dc_prev (accumulate)')
    # sentinel
    sgpre = rnd(rng.standard_normal((B, H)), dtype)
    S = torch.zeros((B, H), dtype=tdt[dtype], device=DEV)
    _lib.call('capmi_sentinel_fwd', p(dev(sgpre, tdt[dtype])), p(dev(cr, tdt[dtype])), p(S), B * H, code[dtype], stream())
    check(host(S), O.sigmoid(sgpre) * tc, dtype, name='sentinel')
    ds = rnd(rng.standard_normal((B, H)), dtype)
    D1, D2 = (torch.zeros((B, H), dtype=tdt[dtype], device=DEV) for _ in range(2))
    _lib.call('capmi_sentinel_bwd', p(dev(ds, tdt[dtype])), p(dev(sgpre, tdt[dtype])), p(dev(cr, tdt[dtype])), p(D1), p(D2), B * H, code[dtype], stream())
    sg = O.sigmoid(sgpre)
    check(host(D1), ds * tc * sg * (1 - sg), dtype, name='sentinel dsg')
    check(host(D2), ds * sg * (1 - tc ** 2), dtype, name='sentinel dc')
    # embedding with padding rows, strided output
    table = rnd(rng.standard_normal((V, E)), dtype)
    ids = rng.randint(0, V, size=13).astype(np.int64)
    ids[[2, 7]] = 0
    out = torch.full((13, E + 8), 7.0, dtype=tdt[dtype], device=DEV)
    IDS = torch.as_tensor(ids, device=DEV)
    _lib.call('capmi_embedding_fwd', p(IDS), p(dev(table, tdt[dtype])), p(out), 13, E, V, E + 8, 0, code[dtype], stream())
    check(host(out)[:, :E], O.embedding_fwd(ids, table, 0), dtype, name='embedding fwd')
    assert np.all(host(out)[:, E:] == 7.0)
    dout = rnd(rng.standard_normal((13, E + 8)), dtype)
    DT = torch.zeros((V, E), dtype=torch.float32, device=DEV)
    _lib.call('capmi_embedding_bwd', p(IDS), p(dev(dout, tdt[dtype])), p(DT), 13, E, V, E + 8, 0, code[dtype], stream())
    check(host(DT), O.embedding_bwd(dout[:, :E], ids, (V, E), 0), dtype, name='embedding bwd')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('B,H,E', [(64, 256, 128), (5, 384, 40)])
def test_fused_lstm_steps_equal_product_plus_cell(dtype, B, H, E):
    """capmi_lstm_step_{fwd,bwd} (recurrent product + cell in one launch) against the two-launch form they replace
    (capmi_igemm_nt + capmi_lstm_cell_*): same arithmetic and rounding points (the stored gate pre-activations are
    bit-identical; the cell may differ by FMA contraction, i.e. an ulp); the two-launch form is pinned against the
    oracle by test_lstm_cell_sentinel_embedding and the GEMM tests."""
    _lib, tdt, code = _env()
    assert _lib.lib().capmi_lstm_step_supported(B, H, code[dtype]) == 1
    rng = np.random.RandomState(B + H)
    ld = E + 2 * H
    lw = dev(rng.standard_normal((4 * H, ld)) / np.sqrt(H), tdt[dtype])           # lstm_w kernel layout [4H][E+H | H]
    es = lw.element_size()
    wh = lw.data_ptr() + (E + H) * es
    hp, cp = dev(rng.standard_normal((B, H)) * 0.5, tdt[dtype]), dev(rng.standard_normal((B, H)) * 0.5, tdt[dtype])
    gin = dev(rng.standard_normal((B, 4 * H)), tdt[dtype])
    # forward
    g1, g2 = gin.clone(), gin.clone()
    h1, c1, h2, c2 = (torch.zeros((B, H), dtype=tdt[dtype], device=DEV) for _ in range(4))
    _KEEP.extend([g1, g2, h1, c1, h2, c2])
    _lib.call('capmi_igemm_nt', p(hp), wh, p(g1), _lib.gemm_geom(B, H), 4 * H, ld, 4 * H, None, p(g1), 4 * H, None, 0, None, 0, 0, 0, code[dtype], stream())
    _lib.call('capmi_lstm_cell_fwd', p(g1), p(cp), p(h1), p(c1), B, H, code[dtype], stream())
    _lib.call('capmi_lstm_step_fwd', p(hp), wh, ld, p(g2), p(cp), p(h2), p(c2), B, H, code[dtype], stream())
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == 'f32' else dict(rtol=8e-3, atol=1e-3)
    close = lambda x, y: torch.allclose(x.float(), y.float(), **tol)
    assert torch.equal(g1, g2) and close(c1, c2) and close(h1, h2)
    # backward: dh_{t-1} += dG_t . Wh, then the cell backward of step t-1
    whT = dev(rng.standard_normal((H, 4 * H)) / np.sqrt(H), tdt[dtype])
    dgt = dev(rng.standard_normal((B, 4 * H)) * 0.1, tdt[dtype])
    dh0 = dev(rng.standard_normal((B, H)) * 0.1, tdt[dtype])
    dcin = dev(rng.standard_normal((B, H)) * 0.1, tdt[dtype])
    dcp0 = dev(rng.standard_normal((B, H)) * 0.1, tdt[dtype])
    for cprev, acc in ((cp, 1), (None, 0)):
        dh1 = dh0.clone()
        dg1, dg2 = (torch.zeros((B, 4 * H), dtype=tdt[dtype], device=DEV) for _ in range(2))
        dcp1, dcp2 = dcp0.clone(), dcp0.clone()
        _KEEP.extend([dh1, dg1, dg2, dcp1, dcp2])
        _lib.call('capmi_igemm_nt', p(dgt), p(whT), p(dh1), _lib.gemm_geom(B, 4 * H), H, 4 * H, H, None, p(dh1), H, None, 0, None, 0, 0, 0,
                  code[dtype], stream())
        _lib.call('capmi_lstm_cell_bwd', p(g1), p(cprev), p(c1), p(dh1), p(dcin), p(dg1), p(dcp1), acc, B, H, code[dtype], stream())
        _lib.call('capmi_lstm_step_bwd', p(dgt), p(whT), 4 * H, p(dh0), p(g1), p(cprev), p(c1), p(dcin), p(dg2), p(dcp2), acc, B, H,
                  code[dtype], stream())
        torch.cuda.synchronize()
        assert close(dg1, dg2) and close(dcp1, dcp2)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('R,E,H,V,gather', [(640, 512, 512, 10000, True), (37, 16, 32, 50, True), (64, 256, 1024, 1000, False)])
def test_fused_decode_step_kernels_equal_the_launches_they_replace(dtype, R, E, H, V, gather):
    """capmi_decode_prep = capmi_embedding_fwd + capmi_gather_rows of h into the [embedding | g | h] operand (:84-88);
    capmi_lstm_cell_sentinel_fwd = capmi_gather_rows of c + capmi_lstm_cell_fwd + capmi_sentinel_fwd (:87-92); the grouped
    p_hid || sent_emb launch (capmi_igemm_nt_group with bias / activation, 64 x 64 tiles) = two capmi_igemm_nt calls --
    all bit for bit."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(R + H)
    T_ = tdt[dtype]
    ids = rng.randint(0, V, R).astype(np.int64)
    ids[::7] = 0                                                       # padding rows
    IDS = dev(ids, torch.int64)
    table = dev(rng.standard_normal((V, E)), T_)
    hsrc, csrc = dev(rng.standard_normal((R, H)), T_), dev(rng.standard_normal((R, H)), T_)
    rows = dev(rng.randint(0, R, R).astype(np.int32), torch.int32) if gather else None
    ld = E + 2 * H
    XH = torch.full((R, ld), 7.0, dtype=T_, device=DEV)
    _lib.call('capmi_decode_prep', p(IDS), p(table), p(hsrc), p(rows), p(XH), R, E, H, V, ld, E + H, 0, code[dtype], stream())
    X0 = torch.zeros((R, E), dtype=T_, device=DEV)
    _lib.call('capmi_embedding_fwd', p(IDS), p(table), p(X0), R, E, V, E, 0, code[dtype], stream())
    Hg = torch.zeros((R, H), dtype=T_, device=DEV)
    Cg = torch.zeros((R, H), dtype=T_, device=DEV)
    if gather:
        _lib.call('capmi_gather_rows', p(hsrc), p(rows), p(Hg), R, H, code[dtype], stream())
        _lib.call('capmi_gather_rows', p(csrc), p(rows), p(Cg), R, H, code[dtype], stream())
    else:
        Hg.copy_(hsrc); Cg.copy_(csrc)
    torch.cuda.synchronize()
    assert torch.equal(XH[:, :E], X0) and torch.equal(XH[:, E + H:], Hg) and bool((XH[:, E:E + H] == 7.0).all())
    assert not bool(XH[::7, :E].any())
    GS = dev(rng.standard_normal((R, 5 * H)), T_)
    h1, c1, s1 = (torch.zeros((R, H), dtype=T_, device=DEV) for _ in range(3))
    _lib.call('capmi_lstm_cell_sentinel_fwd', p(GS), 5 * H, p(csrc), p(rows), p(h1), p(c1), p(s1), R, H, code[dtype], stream())
    G4 = GS[:, :4 * H].contiguous()
    SG = GS[:, 4 * H:].contiguous()
    h2, c2, s2 = (torch.zeros((R, H), dtype=T_, device=DEV) for _ in range(3))
    _lib.call('capmi_lstm_cell_fwd', p(G4), p(Cg), p(h2), p(c2), R, H, code[dtype], stream())
    _lib.call('capmi_sentinel_fwd', p(SG), p(c2), p(s2), R * H, code[dtype], stream())
    torch.cuda.synchronize()
    assert torch.equal(h1, h2) and torch.equal(c1, c2) and torch.equal(s1, s2)
    # grouped projections with bias / activation
    w7, w9 = dev(rng.standard_normal((H, H)) / np.sqrt(H), T_), dev(rng.standard_normal((H, H)) / np.sqrt(H), T_)
    b7, b9 = dev(rng.standard_normal(H), torch.float32), dev(rng.standard_normal(H), torch.float32)
    P1, S1 = torch.zeros((R, H), dtype=T_, device=DEV), torch.zeros((R, H), dtype=T_, device=DEV)
    P2, S2 = torch.zeros((R, H), dtype=T_, device=DEV), torch.zeros((R, H), dtype=T_, device=DEV)
    calls = (_lib.NtCall * 2)()
    for cl, (xin, w_, b_, out, act) in zip(calls, ((h1, w7, b7, P1, _lib.ACT_TANH), (s1, w9, b9, S1, _lib.ACT_NONE))):
        cl.x, cl.w, cl.y, cl.g = p(xin), p(w_), p(out), _lib.gemm_geom(R, H)
        cl.N, cl.ldw, cl.ldy = H, H, H
        cl.bias, cl.act = p(b_), act
    _lib.call('capmi_igemm_nt_group', calls, 2, code[dtype], stream())
    g = _lib.gemm_geom(R, H)
    _lib.call('capmi_igemm_nt', p(h1), p(w7), p(P2), g, H, H, H, p(b7), None, 0, None, 0, None, _lib.ACT_TANH, 0, 0, code[dtype], stream())
    _lib.call('capmi_igemm_nt', p(s1), p(w9), p(S2), g, H, H, H, p(b9), None, 0, None, 0, None, _lib.ACT_NONE, 0, 0, code[dtype], stream())
    torch.cuda.synchronize()
    assert torch.equal(P1, P2) and torch.equal(S1, S2)
    assert float(P1.float().abs().max()) <= 1.0 and float(P1.float().abs().max()) > 0.5


def _attn_ref(Ve, Vt, q, se, s, pp, w10, b10, T, B, K, H, slots):
    """Oracle of one attention call, rows time-major; returns out, alpha."""
    M = T * B
    out = np.zeros((M, H)); alpha = np.ones((M, K + 1))
    for m in range(M):
        b = m % B
        ctx_all = np.concatenate([Vt[b], s[m][None]], 0)
        if slots:
            z = np.tanh(np.concatenate([Ve[b], se[m][None]], 0) + q[m][None])
            e = z @ w10 + b10
            a = np.exp(e - e.max()); a /= a.sum()
            alpha[m] = a
        out[m] = (ctx_all * alpha[m][:, None]).mean(0) + pp[m]
    return out, alpha


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('slots', [0, 1])
def test_attention_fwd_bwd(dtype, slots):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(5 + slots)
    T, B, K, H = 3, 4, 9, 64
    M = T * B
    mk = lambda *s: rnd(rng.standard_normal(s) * 0.7, dtype)
    Ve, Vt = mk(B, K, H), mk(B, K, H)
    q, se, s, pp, dout = mk(M, H), mk(M, H), mk(M, H), mk(M, H), mk(M, H)
    w10, b10 = mk(H), np.array([0.3])
    out, alpha = _attn_ref(Ve, Vt, q, se, s, pp, w10, b10[0], T, B, K, H, slots)
    t = lambda a: dev(a, tdt[dtype])
    f32 = torch.float32
    dVe_, dVt_, dq_, dse_, ds_, out_ = (torch.zeros(sh, dtype=tdt[dtype], device=DEV) for sh in
                                        [(B, K, H), (B, K, H), (M, H), (M, H), (M, H), (M, H)])
    alpha_, de_ = torch.zeros((M, K + 1), dtype=f32, device=DEV), torch.zeros((M, K + 1), dtype=f32, device=DEV)
    tVe, tVt, tq, tse, ts, tp, tw, tb = t(Ve), t(Vt), t(q), t(se), t(s), t(pp), t(w10), dev(b10, f32)
    _lib.call('capmi_ada_attention_fwd', p(tVe), p(tVt), p(tq), p(tse), p(ts), p(tp), p(tw), p(tb), p(out_), p(alpha_), T, B, K, H, slots,
              code[dtype], stream())
    check(host(out_), out, dtype, name='attention out')
    if slots:
        check(host(alpha_), alpha, dtype, name='alpha')
    # backward reference via torch autograd (float64)
    tt = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    aVe, aVt, aq, ase, as_, aw, ab = tt(Ve), tt(Vt), tt(q), tt(se), tt(s), tt(w10), tt(b10)
    bidx = torch.arange(M) % B
    ctx_all = torch.cat([aVt[bidx], as_[:, None]], 1)
    if slots:
        z = torch.tanh(torch.cat([aVe[bidx], ase[:, None]], 1) + aq[:, None])
        al = torch.softmax(z @ aw + ab, dim=1)
    else:
        al = torch.ones(M, K + 1, dtype=torch.float64)
    o = (ctx_all * al[:, :, None]).mean(1)
    (o * torch.tensor(dout)).sum().backward()
    dw10_, db10_ = torch.zeros(H, dtype=f32, device=DEV), torch.zeros(1, dtype=f32, device=DEV)
    _lib.call('capmi_ada_attention_bwd', p(tVe), p(tVt), p(tq), p(tse), p(ts), p(tw), p(alpha_), p(t(dout)), p(ds_), p(dVt_), p(dVe_), p(dq_),
              p(dse_), p(dw10_), p(db10_), p(de_), T, B, K, H, slots, code[dtype], stream())
    check(host(ds_), as_.grad.numpy(), dtype, name='ds')
    check(host(dVt_), aVt.grad.numpy(), dtype, name='dVt')
    if slots:
        check(host(dVe_), aVe.grad.numpy(), dtype, name='dVe')
        check(host(dq_), aq.grad.numpy(), dtype, name='dq')
        check(host(dse_), ase.grad.numpy(), dtype, name='dse')
        check(host(dw10_), aw.grad.numpy(), dtype, name='dw10')
        check(host(db10_), ab.grad.numpy(), dtype, scale=1.0, name='db10')


@pytest.mark.parametrize('B,beam,V', [(3, 4, 37), (70, 5, 1000), (2, 8, 9)])
def test_beam_step_gather_backtrack(B, beam, V):
    """capmi_beam_step / capmi_gather_rows / capmi_beam_backtrack against NumPy, with tied logits and tied totals:
    ties go to the lower beam index, then the lower token id (stable descending sort of the flat [beam*V] totals)."""
    _lib, tdt, code = _env()
    rng = np.random.RandomState(B + beam + V)
    ld = V if V == 37 else (V + 7) // 8 * 8          # 37: an unpadded, odd row pitch (the scalar kernels); else the 16-byte-load kernels
    Ti = 3
    logits = np.round(rng.standard_normal((Ti, beam * B, ld)) * 2) / 2          # coarse values: many exact ties
    score = np.zeros((beam, B), np.float32)
    score[1:] = rng.standard_normal((beam - 1, B)).astype(np.float32)
    score[min(2, beam - 1)] = score[1] if beam > 2 else score[min(2, beam - 1)]   # two hypotheses with equal scores
    f32, i32 = torch.float32, torch.int32
    S = [dev(score, f32), torch.zeros((beam, B), dtype=f32, device=DEV)]
    cv, ci = torch.zeros((beam * B, beam), dtype=f32, device=DEV), torch.zeros((beam * B, beam), dtype=i32, device=DEV)
    lse, rows = torch.zeros(beam * B, dtype=f32, device=DEV), torch.zeros(beam * B, dtype=i32, device=DEV)
    par, tok = torch.zeros((Ti, beam, B), dtype=i32, device=DEV), torch.zeros((Ti, beam, B), dtype=i32, device=DEV)
    ids = torch.zeros(beam * B, dtype=torch.int64, device=DEV)
    _KEEP.extend(S + [cv, ci, lse, rows, par, tok, ids])
    sc = score.astype(np.float64)
    want_par, want_tok = [], []
    for t in range(Ti):
        L = dev(logits[t], f32)
        _lib.call('capmi_beam_step', p(L), V, ld, B, beam, p(S[t % 2]), p(S[(t + 1) % 2]), p(cv), p(ci), p(lse),
                  par.data_ptr() + t * beam * B * 4, tok.data_ptr() + t * beam * B * 4, p(ids), p(rows), stream())
        lg = logits[t][:, :V].astype(np.float32).astype(np.float64).reshape(beam, B, V)
        m = lg.max(-1, keepdims=True)
        logp = lg - (m + np.log(np.exp(lg - m).sum(-1, keepdims=True)))
        tot = (sc[:, :, None] + logp).astype(np.float32).transpose(1, 0, 2).reshape(B, beam * V)
        best = np.argsort(-tot, axis=1, kind='stable')[:, :beam]
        sc = np.take_along_axis(tot, best, 1).T.astype(np.float64)
        want_par.append((best // V).T)
        want_tok.append((best % V).T)
        got_rows, got_ids = host(rows).astype(np.int64), host(ids).astype(np.int64)
        np.testing.assert_array_equal(got_ids.reshape(beam, B), want_tok[-1])
        np.testing.assert_array_equal(got_rows.reshape(beam, B), want_par[-1] * B + np.arange(B)[None, :])
        np.testing.assert_allclose(host(S[(t + 1) % 2]), sc, rtol=0, atol=1e-4)
    out = torch.zeros((B, Ti), dtype=f32, device=DEV)
    _lib.call('capmi_beam_backtrack', p(tok), p(par), p(out), Ti, B, beam, stream())
    want = np.zeros((B, Ti))
    j = np.zeros(B, np.int64)
    for t in reversed(range(Ti)):
        want[:, t] = want_tok[t][j, np.arange(B)]
        j = want_par[t][j, np.arange(B)]
    np.testing.assert_array_equal(host(out), want)
    for dtype in ('f32', 'bf16'):
        src = dev(rng.standard_normal((beam * B, 64)), tdt[dtype])
        dst = torch.zeros_like(src)
        _lib.call('capmi_gather_rows', p(src), p(rows), p(dst), beam * B, 64, code[dtype], stream())
        torch.cuda.synchronize()
        assert torch.equal(dst, src[rows.long()])


@pytest.mark.parametrize('V', [50, 1000, 12295])
def test_softmax_xent_argmax(V):
    _lib, tdt, code = _env()
    rng = np.random.RandomState(V)
    M, ld = 37, (V if V == 50 else (V + 7) // 8 * 8)      # 50: row pitch not a multiple of 4 (the scalar kernels)
    logits = np.zeros((M, ld), np.float32)
    logits[:, :V] = rng.standard_normal((M, V)).astype(np.float32) * 3
    tgt = rng.randint(1, V, size=M).astype(np.int64)
    tgt[::5] = 0
    ce, sm = O.softmax_with_cross_entropy_fwd(logits[:, :V].astype(np.float64), tgt)
    mask = (tgt != 0)
    want_loss = (ce[:, 0] * mask).sum() / mask.sum()
    f32 = torch.float32
    L, TG = dev(logits, f32), torch.as_tensor(tgt, device=DEV)
    rl, lse, loss, cnt = (torch.zeros(n, dtype=f32, device=DEV) for n in (M, M, 1, 1))
    _lib.call('capmi_softmax_xent_fwd', p(L), p(TG), p(rl), p(lse), M, V, ld, 0, stream())
    _lib.call('capmi_xent_finalize', p(rl), p(TG), p(loss), p(cnt), M, 0, stream())
    assert abs(host(loss)[0] - want_loss) < 1e-5 * max(1, abs(want_loss))
    assert host(cnt)[0] == mask.sum()
    for dtype in ('f32', 'bf16'):
        D = torch.full((M, ld), 3.0, dtype=tdt[dtype], device=DEV)
        _lib.call('capmi_softmax_xent_bwd', p(L), p(TG), p(lse), p(cnt), p(D), M, V, ld, ld, 0, code[dtype], stream())
        want = O.softmax_with_cross_entropy_bwd((mask / mask.sum())[:, None], sm, tgt)
        got = host(D)
        check(got[:, :V], want, dtype, scale=1.0 / mask.sum(), name='dlogits')
        assert np.all(got[:, V:] == 0)
    # argmax with ties -> lowest index
    logits[3, :V] = 0.0
    logits[4, 7] = logits[4, 19] = 99.0
    L = dev(logits, f32)
    ids = torch.zeros(M, dtype=torch.int64, device=DEV)
    idf = torch.zeros((M, 6), dtype=f32, device=DEV)
    _lib.call('capmi_argmax', p(L), p(ids), p(idf) + 8, 6, M, V, ld, stream())
    want = logits[:, :V].argmax(1)
    np.testing.assert_array_equal(host(ids).astype(np.int64), want)
    np.testing.assert_array_equal(host(idf)[:, 2].astype(np.int64), want)
    assert want[3] == 0 and want[4] == 7


def test_adam_paddle_form_and_shadows():
    _lib, tdt, code = _env()
    rng = np.random.RandomState(1)
    n = 1003
    pv, g, m, v = (rng.standard_normal(n).astype(np.float32) for _ in range(4))
    v = np.abs(v)
    f32 = torch.float32
    P, G, Mm, Vv = dev(pv, f32), dev(g, f32), dev(m, f32), dev(v, f32)
    from myimagecaptioningmodel_amd.optim import adam_lr_t
    _lib.call('capmi_adam', p(P), p(G), p(Mm), p(Vv), n, adam_lr_t(1e-3, 3), 0.9, 0.999, 1e-8, 0.0, 0.5, stream())
    # oracle: step counter 3, gradient pre-scaled by 1/2 (two ranks)
    po, mo, vo = O.adam_update(pv.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64), v.astype(np.float64), 1e-3, 3)
    np.testing.assert_allclose(host(P), po, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(Mm), mo, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(host(Vv), vo, rtol=1e-5, atol=1e-7)
    # clip
    P2, M2, V2 = dev(pv, f32), dev(m, f32), dev(v, f32)
    _lib.call('capmi_adam', p(P2), p(G), p(M2), p(V2), n, adam_lr_t(1e-3, 1), 0.9, 0.999, 1e-8, 0.1, 1.0, stream())
    po, _, _ = O.adam_update(pv.astype(np.float64), g.astype(np.float64), m.astype(np.float64), v.astype(np.float64), 1e-3, 1, clip=0.1)
    np.testing.assert_allclose(host(P2), po, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ persistent recurrence (capmi_lstm_seq_{fwd,bwd})
def _seq_buffers(rng, T, B, H, E, tdt_):
    ld = E + 2 * H
    lw = dev(rng.standard_normal((4 * H, ld)) / np.sqrt(H), tdt_)             # lstm_w kernel layout [4H][E+H | H]
    whT = lw[:, E + H:].t().contiguous()                                        # [H][4H]: the data-gradient form of the recurrent part
    _KEEP.append(whT)
    gin = dev(rng.standard_normal((T, B, 4 * H)) * 0.7, tdt_)                  # input part of the gates, every step
    return ld, lw, whT, gin


@pytest.mark.parametrize('dtype,B,H,T', [('f32', 64, 256, 7), ('f32', 5, 256, 3), ('bf16', 64, 512, 19), ('bf16', 33, 384, 6), ('bf16', 64, 256, 1), ('bf16', 64, 1024, 5), ('bf16', 17, 768, 4)])
def test_persistent_lstm_sequence_against_oracle_and_per_step_launches(dtype, B, H, T):
    """The whole recurrence of a layer in one launch per direction (grid barrier between steps) against
    (i) oracle.ops.lstm_unit_fwd / lstm_unit_bwd (model_adaAttention_aic.py:87-88), step by step on the same inputs, and
    (ii) the per-step launches it replaces (capmi_lstm_cell_* for the boundary step, capmi_lstm_step_* for the others):
    forward (stored gate pre-activations, h, c) bit-identical in bf16 -- same k-split, MFMA order and rounding points --,
    backward equal up to FMA contraction in the cell.
    The barrier's timeout word must stay clear."""
    _lib, tdt, code = _env()
    L = _lib.lib()
    assert L.capmi_lstm_seq_supported(B, H, T, code[dtype]) == 1
    assert L.capmi_lstm_seq_supported(B, 640, T, code[dtype]) == 0 and L.capmi_lstm_seq_supported(65, H, T, code[dtype]) == 0
    rng = np.random.RandomState(B + H + T)
    E = 24
    ld, lw, whT, gin = _seq_buffers(rng, T, B, H, E, tdt[dtype])
    es = lw.element_size()
    wh = lw.data_ptr() + (E + H) * es
    z = lambda *s: torch.zeros(s, dtype=tdt[dtype], device=DEV)
    sync = torch.zeros((2, 4), dtype=torch.int32, device=DEV)
    # ---- forward: persistent
    g_seq, h_seq, c_seq = gin.clone(), z(T + 1, B, H), z(T + 1, B, H)
    _KEEP.extend([g_seq, h_seq, c_seq, sync])
    _lib.call('capmi_lstm_seq_fwd', p(h_seq), wh, ld, p(g_seq), p(c_seq), B, H, T, sync.data_ptr(), code[dtype], stream())
    # ---- forward: per-step launches (the default plan of round 1)
    g_ref, h_ref, c_ref = gin.clone(), z(T + 1, B, H), z(T + 1, B, H)
    _KEEP.extend([g_ref, h_ref, c_ref])
    for t in range(T):
        if t == 0:
            _lib.call('capmi_lstm_cell_fwd', p(g_ref[0]), p(c_ref[0]), p(h_ref[1]), p(c_ref[1]), B, H, code[dtype], stream())
        else:
            _lib.call('capmi_lstm_step_fwd', p(h_ref[t]), wh, ld, p(g_ref[t]), p(c_ref[t]), p(h_ref[t + 1]), p(c_ref[t + 1]), B, H, code[dtype], stream())
    torch.cuda.synchronize()
    assert not sync[:, 1].any(), sync
    assert int(sync[0, 0]) == (T - 1) * (H // 8)                    # one arrival per workgroup and barrier
    # bf16: bit-identical; f32: the two code paths may contract a*b + c*d differently (one ulp), which the recurrence carries on
    same = (lambda x, y: torch.equal(x, y)) if dtype == 'bf16' else (lambda x, y: torch.allclose(x, y, rtol=2e-5, atol=2e-6))
    assert same(g_seq, g_ref) and same(h_seq, h_ref) and same(c_seq, c_ref), [float((x.float() - y.float()).abs().max()) for x, y in ((g_seq, g_ref), (h_seq, h_ref), (c_seq, c_ref))]
    # ---- forward: oracle, step by step (f64 on the storage-rounded inputs)
    W = rnd(host(lw), dtype).T                                        # reference layout [(E+H)+H, 4H]; x part multiplies zeros here
    hh, cc = np.zeros((B, H)), np.zeros((B, H))
    x0 = np.zeros((B, E + H))
    caches = []
    tol = dict(f32=2e-5, bf16=3e-2)[dtype]
    for t in range(T):
        hh, cc, cache = O.lstm_unit_fwd(x0, hh, cc, W, np.zeros(4 * H))
        # the kernel adds the precomputed input part: fold it in by recomputing the cell from the full pre-activation
        pre = cache[0] @ W + host(gin[t])
        i_, f_, o_, g_ = O.sigmoid(pre[:, :H]), O.sigmoid(pre[:, H:2 * H]), O.sigmoid(pre[:, 2 * H:3 * H]), np.tanh(pre[:, 3 * H:])
        c_prev = cache[6]
        cc = f_ * c_prev + i_ * g_
        hh = o_ * np.tanh(cc)
        caches.append((cache[0], i_, f_, o_, g_, np.tanh(cc), c_prev))
        assert np.abs(host(g_seq[t]) - pre).max() <= tol * max(1.0, np.abs(pre).max()), ('gates', t)
        assert np.abs(host(c_seq[t + 1]) - cc).max() <= tol * max(1.0, np.abs(cc).max()), ('c', t)
        assert np.abs(host(h_seq[t + 1]) - hh).max() <= tol, ('h', t)
        if dtype == 'bf16':       # follow the device's rounded state, so that rounding does not accumulate into the comparison
            hh, cc = host(h_seq[t + 1]), host(c_seq[t + 1])
    # ---- backward
    for top in (1, 0):
        dh = dev(rng.standard_normal((T + 1, B, H)) * 0.1, tdt[dtype])
        dc0 = dev(rng.standard_normal((T + 1, B, H)) * 0.1, tdt[dtype]) if top else z(T + 1, B, H)
        dc_seq, dg_seq = dc0.clone(), z(T, B, 4 * H)
        sync.zero_()
        _KEEP.extend([dc_seq, dg_seq])
        _lib.call('capmi_lstm_seq_bwd', p(g_seq), p(c_seq), p(whT), 4 * H, p(dh), p(dc_seq), p(dg_seq), top, B, H, T,
                  sync.data_ptr() + 16, code[dtype], stream())
        # per-step launches: capmi_lstm_cell_bwd for the last step, capmi_lstm_step_bwd (product + cell of step t-1) for the rest
        dc_ref, dg_ref = dc0.clone(), z(T, B, 4 * H)
        _KEEP.extend([dc_ref, dg_ref])
        t = T - 1
        _lib.call('capmi_lstm_cell_bwd', p(g_ref[t]), p(c_ref[t]), p(c_ref[t + 1]), p(dh[t + 1]), p(dc_ref[t + 1]) if top else None, p(dg_ref[t]),
                  p(dc_ref[t]) if t > 0 else None, top, B, H, code[dtype], stream())
        for t in range(T - 1, 0, -1):
            _lib.call('capmi_lstm_step_bwd', p(dg_ref[t]), p(whT), 4 * H, p(dh[t]), p(g_ref[t - 1]), p(c_ref[t - 1]), p(c_ref[t]), p(dc_ref[t]),
                      p(dg_ref[t - 1]), p(dc_ref[t - 1]) if t > 1 else None, top, B, H, code[dtype], stream())
        torch.cuda.synchronize()
        assert not sync[:, 1].any(), sync
        assert int(sync[1, 0]) == (T - 1) * (H // 32) * ((B + 15) // 16)
        # backward: the cell's sums of products may be contracted into FMAs differently in the two code paths (one f32 ulp,
        # now and then one bf16 ulp after rounding) -- the tolerance test_fused_lstm_steps_equal_product_plus_cell uses
        btol = dict(rtol=1e-5, atol=1e-6) if dtype == 'f32' else dict(rtol=8e-3, atol=1e-3)
        assert torch.allclose(dg_seq.float(), dg_ref.float(), **btol), float((dg_seq.float() - dg_ref.float()).abs().max())
        assert torch.allclose(dc_seq[1:].float(), dc_ref[1:].float(), **btol)
        assert dtype == 'f32' or float((dg_seq != dg_ref).float().mean()) < 0.02      # ... and, after rounding to bf16, rare
        # oracle BPTT with the caches of the forward pass above (f32 only: tight)
        if dtype == 'f32':
            dh_next = np.zeros((B, H)); dc_next = np.zeros((B, H))
            for t in reversed(range(T)):
                dht = host(dh[t + 1]) + dh_next
                dct = (host(dc0[t + 1]) if top else 0.0) + dc_next
                _dx, dh_next, dc_next, _dw, _db = O.lstm_unit_bwd(dht, dct, caches[t], W)
                xin, i_, f_, o_, g_, tc, c_prev = caches[t]
                do = dht * tc
                dc_tot = dct + dht * o_ * (1 - tc * tc)
                want = np.concatenate([dc_tot * g_ * i_ * (1 - i_), dc_tot * c_prev * f_ * (1 - f_), do * o_ * (1 - o_), dc_tot * i_ * (1 - g_ * g_)], -1)
                got = host(dg_seq[t])
                assert np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max()), ('dG', t, np.abs(got - want).max())
