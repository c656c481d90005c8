"""Per-launch timing of a launch plan with HIP events, plus the algorithmic work model
(FLOPs / HBM bytes per launch) used for the roofline line of bench.py.

`time_step` runs the plans of a train step on their lanes (two HIP streams) with a pair of timing events around every launch,
recorded on the stream the launch goes to: the IN-MODEL duration of every kernel, which is what `rocprofv3 --kernel-trace
--stats` of the same command reports.  `time_plan` is the single-stream variant (every kernel with the chip to itself; tools).
GEMM launches are labelled with the EXACT symbol of the kernel the library dispatches to -- asked of the library
(`kernel_symbol`), not mirrored here -- so a label is a row of the rocprof summary verbatim.
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


_SYMBOLS = {}


def kernel_symbol(name, args):
    """The kernel a capmi_igemm_* launch runs, as rocprofv3 prints it -- asked of the library's own dispatch code
    (capmi_kernel_probe_begin / _end, include/capmi.h), memoised per argument shape (pointers do not choose kernels... except
    by being NULL, which the key keeps)."""
    from . import _lib

    def sig(a):
        if isinstance(a, ctypes._Pointer):
            a = a.contents
        if isinstance(a, ctypes.Structure):
            return tuple(sig(getattr(a, f)) for f, _ in a._fields_)
        if isinstance(a, ctypes.Array):
            return tuple(sig(x) for x in a)
        if isinstance(a, _lib.PtrSlot) or (isinstance(a, int) and abs(a) >= (1 << 32)):
            return 'ptr'
        return a
    key = (name, _lib.lib().capmi_deterministic(), _lib.lib().capmi_general_epilogue()) + tuple(sig(a) for a in args)
    hit = _SYMBOLS.get(key)
    if hit is None:
        hit = _SYMBOLS[key] = _lib.probe_kernel(name, *args)
    return hit


def describe(name, args):
    """-> (label, flops, algorithmic_bytes) of one recorded launch."""
    if name == 'capmi_igemm_nt':
        g = _geom(args[3])
        N, code, out_f32 = args[4], args[16], args[15]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        sym = kernel_symbol(name, args)[0]
        if 'skinny' in sym:
            return sym, 2.0 * M * N * K, (M * K + N * K + M * N) * es
        # each input pixel / weight read once, output written once (im2col re-reads are on-chip)
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * (4 if out_f32 else es)
        if args[8]:
            nbytes += M * N * es
        if args[10]:        # the activation mask: the saved output, or one bit per element (CAPMI_DACT_BITMASK)
            nbytes += M * N * es if not (args[14] & 0x100) else M * N // 8
        return sym, 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_bnred':
        # (x, w, y, g, N, ldw, ldy, addend, ld_addend, ysaved, ld_saved, dact, nred, 8 target fields, dtype)
        g = _geom(args[3])
        N, code, nred = args[4], args[21], args[12]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * es * (1 + nred + (1 if args[7] else 0) + (1 if args[9] else 0))
        return kernel_symbol(name, args)[0], 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_bnsum':
        # (x, w, y, g, N, ldw, ldy, addend, ld_addend, maskbits, ld_saved, dact, raw, mean, invstd, acc_rows, parts_ws, red, dtype)
        g = _geom(args[3])
        N, code = args[4], args[18]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * es * (2 + (1 if args[7] else 0)) + M * N // 8
        return kernel_symbol(name, args)[0], 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_group':
        calls, n, code = args[0], args[1], args[2]
        es = 2 if code == BF16 else 4
        flops = nbytes = 0.0
        for c in calls[:n]:
            g = c.g
            M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
            flops += 2.0 * M * c.N * K
            nbytes += c.N * K * es + M * c.N * es * (1 + (1 if c.addend else 0)) + (0 if not c.ysaved else (M * c.N // 8 if c.dact & 0x100 else M * c.N * es))
        g = calls[0].g
        nbytes += g.B * g.Hi * g.Wi * g.Cin * es          # the shared input is read once
        return kernel_symbol(name, args)[0], flops, nbytes
    if name == 'capmi_bn_bwd_reduce_final':
        return 'bn_bwd_reduce_final_kernel', 0.0, args[1] * 2 * args[2] * 4
    if name == 'capmi_igemm_tn_wgrad':
        g = _geom(args[3])
        N, code = args[4], args[9]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + M * N * es + N * K * 4
        return kernel_symbol(name, args)[0], 2.0 * M * N * K, nbytes
    if name in ('capmi_igemm_nt_bn', 'capmi_igemm_nt_bnact', 'capmi_igemm_nt_bnfin', 'capmi_igemm_nt_stat'):
        # (x, w, y, g, N, ldw, ldy, ...): the convolution's own operands; epilogue vectors are noise next to them
        g = _geom(args[3])
        N, code = args[4], args[-1]
        M, K = g.B * g.Ho * g.Wo, g.kh * g.kw * g.Cin
        es = 2 if code == BF16 else 4
        nbytes = g.B * g.Hi * g.Wi * g.Cin * es + N * K * es + M * N * es
        if name == 'capmi_igemm_nt_bn' and args[10]:
            nbytes += M * N * es                            # residual
        return kernel_symbol(name, args)[0], 2.0 * M * N * K, nbytes
    if name == 'capmi_igemm_nt_splitk':
        M, K, N, code = args[3], args[4], args[6], args[11]
        es = 2 if code == BF16 else 4
        return kernel_symbol(name, args)[0], 2.0 * M * N * K, (M * K + N * K + M * N) * es
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
    if name == 'capmi_bn_stat_apply':
        M, C, code = args[5], args[6], args[21]
        return 'bn_stat_apply_kernel', 0.0, M * C * es_of(code) * (3 if args[17] else 2) + (M * C // 8 if args[19] else 0)
    if name == 'capmi_bn_stat_apply_pool':
        # (x, parts, part_rows, rows, shift, B, Hi, Wi, C, Ho, Wo, ...): the conv output read (each element by the windows that cover it: once
        # from memory), the pooled tensor + its argmax map written
        B_, Hi, Wi, C, Ho, Wo, code = args[5], args[6], args[7], args[8], args[9], args[10], args[25]
        return 'bn_stat_apply_pool_kernel', 0.0, B_ * Hi * Wi * C * es_of(code) + B_ * Ho * Wo * C * (es_of(code) + 1)
    if name == 'capmi_bn_apply_mask':
        M, C, code = args[7], args[8], args[10]
        return 'bn_apply_kernel', 0.0, M * C * es_of(code) * (3 if args[4] else 2) + M * C // 8
    if name == 'capmi_bn_bwd_reduce':
        M, C, act, code = args[7], args[8], args[9], args[10]
        return 'bn_bwd_reduce_kernel', 0.0, M * C * es_of(code) * (3 if act else 2)
    if name == 'capmi_bn_bwd_reduce_spread':
        M, C, act, code = args[8], args[9], args[10], args[11]
        return 'bn_bwd_reduce_kernel', 0.0, M * C * es_of(code) * (3 if act else 2)
    if name == 'capmi_bn_bwd_apply':
        M, C, act, code = args[11], args[12], args[13], args[14]
        n = 3 + (1 if act else 0) + (1 if args[9] else 0)
        return 'bn_bwd_apply_kernel', 0.0, M * C * es_of(code) * n
    if name == 'capmi_bn_bwd_apply_spread':
        M, C, act, code = args[12], args[13], args[14], args[15]
        n = 3 + (1 if act else 0) + (1 if args[10] else 0)
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


def time_label_alone(plans, label, stream_ptr, overhead_ms=0.0):
    """Average duration of the launches labelled `label` (describe()) when each runs ALONE: the device is idle before every one of
    them, the launch sits between two timing events on the caller's stream.  Next to time_step's in-model figure this says how much
    of a kernel's time in the step is its own and how much is what running beside the other lane costs it.
    -> (average ms, launches)"""
    cur = torch.cuda.current_stream()
    assert cur.cuda_stream == stream_ptr
    total, n = 0.0, 0
    for plan in plans:
        for fn, name, args in plan.launches():
            if not name.startswith(('capmi_igemm_', 'capmi_lstm_')) or describe(name, args)[0] != label:
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(cur)
            rc = fn(*[x.value if hasattr(x, 'value') and type(x).__name__ == 'PtrSlot' else x for x in args], stream_ptr)
            b.record(cur)
            torch.cuda.synchronize()
            if rc != 0:
                raise RuntimeError('%s failed during timing' % name)
            total += max(a.elapsed_time(b) - overhead_ms, 0.0005)
            n += 1
    return (total / n if n else 0.0), n


def event_pair_overhead_ms(stream_ptr, n=64):
    """What a PAIR of timing events around one launch adds to the interval they measure (the two barrier packets the command
    processor handles on either side of the kernel): the same n small fills timed once with a pair around each and once with
    one pair around all of them, on an idle device.  Subtracted from every per-launch interval of time_step -- without it a
    10 us kernel reads 16-18 us and a 38 us kernel 45 us against rocprofv3's dispatch-to-end durations."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    buf = torch.zeros(1 << 20, dtype=torch.float32, device='cuda')

    def ev():
        e = ctypes.c_void_p()
        if L.capmi_event_create_timed(ctypes.byref(e)) != 0:
            raise _lib.CapmiError(_lib.last_error())
        return e

    def elapsed(a, b):
        ms = ctypes.c_float(0.0)
        if L.capmi_event_elapsed_ms(a, b, ctypes.byref(ms)) != 0:
            raise _lib.CapmiError(_lib.last_error())
        return float(ms.value)
    fill = lambda: L.capmi_fill_f32(buf.data_ptr(), 0.0, buf.numel(), stream_ptr)
    for _ in range(8):
        fill()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    L.capmi_event_record(a, stream_ptr)
    for _ in range(n):
        fill()
    L.capmi_event_record(b, stream_ptr)
    torch.cuda.synchronize()
    together = elapsed(a, b)
    pairs = []
    for _ in range(n):
        x, y = ev(), ev()
        L.capmi_event_record(x, stream_ptr)
        fill()
        L.capmi_event_record(y, stream_ptr)
        pairs.append((x, y))
    torch.cuda.synchronize()
    each = sum(elapsed(x, y) for x, y in pairs)
    for e in [a, b] + [z for p in pairs for z in p]:
        L.capmi_event_destroy(e)
    return max(0.0, (each - together) / n)


def time_step(plans, stream_ptr, repeats=2, overhead_ms=0.0):
    """IN-MODEL per-kernel time of the plans of one train step, run on their lanes (_lib.Plan.run_timed), grouped by label:
    {label: dict(ms, launches, flops, bytes, lanes)} summed over `repeats` runs, and the per-lane busy time."""
    stats, lane_ms = {}, {}
    for _ in range(repeats):
        for plan in plans:
            for name, args, lane, ms in plan.run_timed(stream_ptr):
                label, fl, by = describe(name, args)
                s = stats.setdefault(label, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, lanes=set()))
                ms = max(ms - overhead_ms, 0.0005)
                s['ms'] += ms
                s['launches'] += 1
                s['flops'] += fl
                s['bytes'] += by
                s['lanes'].add(lane)
                lane_ms[lane] = lane_ms.get(lane, 0.0) + ms
    return stats, lane_ms
