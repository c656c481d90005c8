"""Reproducers of the halo-staged 3x3 kernel's LDS write-after-read (DESIGN.md lesson 29), one sub-command each:
    alone    the 3x3 / stride-1 halo-staged conv kernel against torch conv2d (f32 on bf16-rounded operands); prints where it differs.
    inmodel  in-model conv outputs of the 3x3 stride-1 layers against torch conv2d on the engine's own input tensors.
    touched  which encoder tensors does the backward plan modify (it must modify none of raw / act)?
    fwdonly  forward tensors of forward_backward() (no sync between the two plans) against those of a forward-only run.
    repeat   is the halo-staged conv deterministic when run alone / with statistics / next to a kernel on another stream?
    engine   the halo conv on the ENGINE's own tensors, repeated -- does the address / layout of the in-model operands matter?
    taps     for the pixels the in-model halo conv gets wrong, which (tap, channel chunk) contribution is missing / garbage?
usage: python tools/halo_check.py <sub-command>   (on the GPU box; every one prints what it finds)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mode_alone():
    """the 3x3 / stride-1 halo-staged conv kernel against torch conv2d (f32 on bf16-rounded operands); prints where it differs."""
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

def mode_inmodel():
    """in-model conv outputs of the 3x3 stride-1 layers against torch conv2d on the engine's own input tensors."""
    import os, sys
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from myimagecaptioningmodel_amd import arch, default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    B = 64
    cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
    image, cap = bench.synthetic_batch(B, cfg, 1234)
    mode = sys.argv[1] if len(sys.argv) > 1 else 'fb'
    if mode == 'f':
        eng.forward_loss(image, cap)
    else:
        eng.forward_backward(image, cap)
    torch.cuda.synchronize()
    enc = eng._train[B]['enc']
    params = eng.export_reference_params()
    for op in enc.enc.ops:
        if not isinstance(op, arch.ConvBN) or op.k != 3 or op.stride != 1:
            continue
        x = enc.act[op.src].float()
        w = torch.as_tensor(params[op.name + '_weights']).cuda().to(torch.bfloat16).float()     # [co, ci, 3, 3]
        ref = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
        raw = enc.raw[op.dst].float()
        d = (raw - ref).abs()
        bad = d > 0.03 * ref.abs().max()
        print('%-18s rel L2 %.2e  bad %d' % (op.name, float((raw - ref).norm() / ref.norm()), int(bad.sum())), end='')
        if bad.any():
            idx = bad.nonzero()
            H = x.shape[1]
            pix = idx[:, 0] * H * H + idx[:, 1] * H + idx[:, 2]
            print('  pixels %d  b %s h %s w %s  m%%64 %s' % (len(set(pix.tolist())), sorted(set(idx[:, 0].tolist()))[:6], sorted(set(idx[:, 1].tolist()))[:8],
                  sorted(set(idx[:, 2].tolist()))[:8], sorted(set((pix % 64).tolist()))[:10]), end='')
        print()

def mode_touched():
    """which encoder tensors does the backward plan modify (it must modify none of raw / act)?"""
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from myimagecaptioningmodel_amd import arch, default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    B = 64
    cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
    image, cap = bench.synthetic_batch(B, cfg, 1234)
    eng.forward_loss(image, cap)
    torch.cuda.synchronize()
    prog = eng._train[B]
    enc = prog['enc']
    snap_raw = {k: v.clone() for k, v in enc.raw.items()}
    snap_act = {k: v.clone() for k, v in enc.act.items()}
    prog['bwd'].run(eng._stream())
    torch.cuda.synchronize()
    names = {op.dst: op.name for op in enc.enc.ops if isinstance(op, arch.ConvBN)}
    for k, v in enc.raw.items():
        n = int((v != snap_raw[k]).sum())
        if n:
            idx = (v != snap_raw[k]).nonzero()
            print('raw', names.get(k, k), tuple(v.shape), 'changed elements', n, 'first', idx[0].tolist(), 'last', idx[-1].tolist())
    for k, v in enc.act.items():
        n = int((v != snap_act[k]).sum())
        if n:
            idx = (v != snap_act[k]).nonzero()
            print('act', names.get(k, k), tuple(v.shape), 'changed elements', n, 'first', idx[0].tolist(), 'last', idx[-1].tolist())
    print('done')

def mode_fwdonly():
    """forward tensors of forward_backward() (no sync between the two plans) against those of a forward-only run."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from myimagecaptioningmodel_amd import arch, default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    B = 64
    cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
    image, cap = bench.synthetic_batch(B, cfg, 1234)
    params = eng.export_reference_params()
    eng.forward_loss(image, cap)
    torch.cuda.synchronize()
    enc = eng._train[B]['enc']
    snap_raw = {k: v.clone() for k, v in enc.raw.items()}
    snap_act = {k: v.clone() for k, v in enc.act.items()}
    names = {op.dst: op.name for op in enc.enc.ops if isinstance(op, arch.ConvBN)}
    for rep in range(3):
        eng.load_reference_params(params)        # (running statistics back to their start: the forward pass is deterministic)
        mode = sys.argv[1] if len(sys.argv) > 1 else 'fb'
        if mode == 'fb':
            eng.forward_backward(image, cap)
        else:
            eng.forward_loss(image, cap)
        torch.cuda.synchronize()
        out = []
        for k, v in enc.raw.items():
            n = int((v != snap_raw[k]).sum())
            if n:
                out.append('raw %s %d' % (names.get(k, k), n))
        for k, v in enc.act.items():
            n = int((v != snap_act[k]).sum())
            if n:
                out.append('act %s %d' % (names.get(k, k), n))
        print('rep', rep, mode, 'changed:', out[:12], '... total tensors', len(out))

def mode_repeat():
    """is the halo-staged conv deterministic when run alone / with statistics / next to a kernel on another stream?"""
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from myimagecaptioningmodel_amd import _lib
    dev = 'cuda:0'
    B, H, Cin, Cout = 64, 56, 64, 64
    torch.manual_seed(0)
    x = torch.relu(torch.randn((B, H, H, Cin), device=dev)).to(torch.bfloat16)
    w = (torch.randn((Cout, 3, 3, Cin), device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    g = _lib.ConvGeom(B, H, H, Cin, H, H, 3, 3, 1, 1, 1, Cin)
    M = B * H * H
    pr = _lib.lib().capmi_igemm_nt_stats_part_rows(M, Cout, 9 * Cin, _lib.BF16)
    stats = torch.zeros(((M + pr - 1) // pr + 64, Cout, 2), device=dev)
    side = torch.cuda.Stream()
    big = torch.randn((64, 56, 56, 256), device=dev).to(torch.bfloat16)
    def run(with_stats, concurrent):
        outs = []
        for r in range(6):
            y = torch.zeros((B, H, H, Cout), device=dev, dtype=torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream
            if concurrent:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        big2 = big * 1.0001
            _lib.call('capmi_igemm_nt', x.data_ptr(), w.data_ptr(), y.data_ptr(), g, Cout, 9 * Cin, Cout, None, None, 0, None, 0,
                      stats.data_ptr() if with_stats else None, 0, 0, 0, _lib.BF16, st)
            torch.cuda.synchronize()
            outs.append(y)
        return [int((o != outs[0]).sum()) for o in outs[1:]]
    for ws in (False, True):
        for cc in (False, True):
            print('stats', ws, 'concurrent', cc, 'elements differing from run 0:', run(ws, cc))

    # ---- the in-model neighbourhood: input freshly written by the kernel in front (bn_apply), other LDS-DMA kernels before it
    xr = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    mean = torch.zeros(Cin, device=dev); ca = torch.ones(Cin, device=dev); off = torch.zeros(Cin, device=dev)
    xa = torch.zeros_like(xr)
    w1 = (torch.randn((256, Cin), device=dev) / 8).to(torch.bfloat16)
    y1 = torch.zeros((B, H, H, 256), device=dev, dtype=torch.bfloat16)
    g1 = _lib.ConvGeom(B, H, H, Cin, H, H, 1, 1, 1, 1, 0, Cin)
    def run2(pre_gemm, pre_apply):
        outs = []
        for r in range(6):
            y = torch.zeros((B, H, H, Cout), device=dev, dtype=torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream
            if pre_gemm:
                _lib.call('capmi_igemm_nt', xr.data_ptr(), w1.data_ptr(), y1.data_ptr(), g1, 256, Cin, 256, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16, st)
            if pre_apply:
                xa.zero_()
                _lib.call('capmi_bn_apply', xr.data_ptr(), mean.data_ptr(), ca.data_ptr(), off.data_ptr(), None, xa.data_ptr(), M, Cin, 1, _lib.BF16, st)
            _lib.call('capmi_igemm_nt', (xa if pre_apply else x).data_ptr(), w.data_ptr(), y.data_ptr(), g, Cout, 9 * Cin, Cout, None, None, 0, None, 0,
                      stats.data_ptr(), 0, 0, 0, _lib.BF16, st)
            torch.cuda.synchronize()
            outs.append(y)
        return [int((o != outs[0]).sum()) for o in outs[1:]]
    for pg in (False, True):
        for pa in (False, True):
            print('1x1 GEMM in front', pg, ' input written by bn_apply in front', pa, ' elements differing from run 0:', run2(pg, pa))

def mode_engine():
    """the halo conv on the ENGINE's own tensors, repeated -- does the address / layout of the in-model operands matter?"""
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

def mode_taps():
    """for the pixels the in-model halo conv gets wrong, which (tap, channel chunk) contribution is missing / garbage?"""
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


MODES = {'alone': mode_alone, 'inmodel': mode_inmodel, 'touched': mode_touched, 'fwdonly': mode_fwdonly, 'repeat': mode_repeat, 'engine': mode_engine, 'taps': mode_taps}

if __name__ == '__main__':
    if len(sys.argv) != 2 or sys.argv[1] not in MODES:
        raise SystemExit(__doc__)
    MODES[sys.argv[1]]()
