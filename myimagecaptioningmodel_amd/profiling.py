"""Per-launch timing of a launch plan with HIP events, plus the algorithmic work model
(FLOPs / HBM bytes per launch) used for the roofline line of bench.py.

`time_plan` replays a Plan eagerly on the current stream and brackets every launch with a pair
of events recorded on THAT stream, then groups by kernel label.  Labels follow the kernel
symbol the C ABI entry dispatches to (tile shape included for the MFMA kernels) so they can be
matched against `rocprofv3 --kernel-trace --stats` rows.
"""
import os
import ctypes

import torch

from ._lib import BF16

# Peaks used as roofline denominators (/opt/skills/guides/MI355X_MICROARCH.md, chip-level table)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec; ~6300 measured achievable
PEAK_MFMA_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}


def _geom(arg):
    g = arg
    if isinstance(arg, ctypes._Pointer):
        g = arg.contents
    return g


def _nt_tile(M, N, K, bf16):
    """Mirror of nt_cfg() in csrc/igemm.hip (labels only)."""
    cd = lambda x, y: -(-x // y)
    wide = N > 64
    if bf16:
        if wide:
            big = K >= int(os.environ.get('CAPMI_NT_BIGK', '64')) and cd(M, 128) * cd(N, 128) >= int(os.environ.get('CAPMI_NT_BIGTILES', '384'))
            if not big and K >= 1024 and 160 <= cd(M, 128) * cd(N, 128) <= 256:
                return (128, 128, True)
            if not big and cd(M, 64) * cd(N, 128) < 256:
                return (64, 64, True)
            return (128 if big else 64, 128, True)
        if N >= 32:
            tall64 = int(os.environ.get('CAPMI_NT_TALL64', '1024'))
            return (128, 64, True) if (tall64 > 0 and cd(M, 128) >= tall64) else (64, 64, True)
        return (128 if cd(M, 128) >= 512 else 64, 64, False)
    tall = cd(M, 128) * cd(N, 64) >= 256
    return (64, 128, False) if wide else ((128 if tall else 64), 64, False)


def describe(name, args):
    """-> (label, flops, algorithmic_bytes) of one recorded launch."""
    if name == 'capmi_igemm_nt':
        g = _geom(args[3])
        N, code, out_f32 = args[4], args[16], args[15]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        bm, bn, glds = _nt_tile(M, N, K, code == BF16)
        plain = g.kh == 1 and g.kw == 1 and g.sd == 1 and g.up == 1 and g.pad == 0 and g.Hi == 1 and g.Wi == 1
        if plain and not args[12] and M <= 64 and K % 128 == 0 and K >= 256:
            return 'igemm_nt_skinny_kernel<%s>' % ('bf16' if code == BF16 else 'f32'), 2.0 * M * N * K, (M * K + N * K + M * N) * es
        # each input pixel / weight read once, output written once (im2col re-reads are on-chip)
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * (4 if out_f32 else es)
        if args[8]:
            nbytes += M * N * es
        if args[10]:
            nbytes += M * N * es
        if glds:
            return 'igemm_nt_glds_kernel<%d,%d>' % (bm, bn), 2.0 * M * N * K, nbytes
        return 'igemm_nt_kernel<%s,%d,%d>' % ('bf16' if code == BF16 else 'f32', bm, bn), 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_bnred':
        # (x, w, y, g, N, ldw, ldy, addend, ld_addend, ysaved, ld_saved, dact, nred, 8 target fields, dtype)
        g = _geom(args[3])
        N, code, nred = args[4], args[21], args[12]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        bm, bn, glds = _nt_tile(M, N, K, code == BF16)
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * es * (1 + nred + (1 if args[7] else 0) + (1 if args[9] else 0))
        if glds:
            return 'igemm_nt_glds_kernel<%d,%d>' % (bm, bn), 2.0 * M * N * K, nbytes
        return 'igemm_nt_kernel<%s,%d,%d>' % ('bf16' if code == BF16 else 'f32', bm, bn), 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_group':
        calls, n, code = args[0], args[1], args[2]
        es = 2 if code == BF16 else 4
        flops = nbytes = 0.0
        for c in calls[:n]:
            g = c.g
            M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
            flops += 2.0 * M * c.N * K
            nbytes += c.N * K * es + M * c.N * es * (1 + (1 if c.addend else 0) + (1 if c.ysaved else 0))
        g = calls[0].g
        nbytes += g.B * g.Hi * g.Wi * g.Cin * es          # the shared input is read once
        return 'igemm_nt_glds_group_kernel', flops, nbytes
    if name == 'capmi_bn_bwd_reduce_final':
        return 'bn_bwd_reduce_final_kernel', 0.0, args[1] * 2 * args[2] * 4
    if name == 'capmi_igemm_tn_wgrad':
        g = _geom(args[3])
        N, code = args[4], args[9]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        big = N >= 128 and K >= 128 and code == BF16
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + M * N * es + N * K * 4
        if big:
            return 'igemm_tn_glds_kernel', 2.0 * M * N * K, nbytes
        return 'igemm_tn_kernel<%s,64,64>' % ('bf16' if code == BF16 else 'f32'), 2.0 * M * N * K, nbytes
    es_of = lambda code: 2 if code == BF16 else 4
    if name == 'capmi_s2d_stem':
        B, C, H, W, Hb, Wb, Cs, code = args[2], args[3], args[4], args[5], args[7], args[8], args[9], args[10]
        return 's2d_stem_kernel', 0.0, B * C * H * W * 4 + B * Hb * Wb * Cs * es_of(code)
    if name == 'capmi_bn_apply':
        M, C, code = args[6], args[7], args[9]
        return 'bn_apply_kernel', 0.0, M * C * es_of(code) * (3 if args[4] else 2)
    if name == 'capmi_bn_finalize_apply':
        M, C, code = args[2], args[3], args[17]
        return 'bn_finalize_apply_kernel', 0.0, M * C * es_of(code) * (3 if args[14] else 2)
    if name == 'capmi_bn_stats':
        return 'bn_stats_kernel', 0.0, args[1] * args[2] * es_of(args[4])
    if name == 'capmi_bn_bwd_reduce':
        M, C, act, code = args[7], args[8], args[9], args[10]
        return 'bn_bwd_reduce_kernel', 0.0, M * C * es_of(code) * (3 if act else 2)
    if name == 'capmi_bn_bwd_apply':
        M, C, act, code = args[11], args[12], args[13], args[14]
        n = 3 + (1 if act else 0) + (1 if args[9] else 0)
        return 'bn_bwd_apply_kernel', 0.0, M * C * es_of(code) * n
    return name.replace('capmi_', '') + '_kernel', 0.0, 0.0


def time_plan(plan, stream_ptr, repeats=1):
    """Returns {label: dict(ms, launches, flops, bytes)} for one (or `repeats`) eager replays."""
    stats = {}
    cur = torch.cuda.current_stream()
    assert cur.cuda_stream == stream_ptr
    for _ in range(repeats):
        evs = []
        for fn, name, args in plan.launches():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
            rc = fn(*[x.value if hasattr(x, 'value') and type(x).__name__ == 'PtrSlot' else x for x in args], stream_ptr)
            b.record(cur)
            if rc != 0:
                raise RuntimeError('%s failed during timing' % name)
            evs.append((name, args, a, b))
        torch.cuda.synchronize()
        for name, args, a, b in evs:
            label, fl, by = describe(name, args)
            s = stats.setdefault(label, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            s['ms'] += a.elapsed_time(b)
            s['launches'] += 1
            s['flops'] += fl
            s['bytes'] += by
    return stats
