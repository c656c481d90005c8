"""Encoder executor: turns an op list (arch.py) into launch plans of libcapmi kernels.

Forward of one `conv -> batch_norm -> relu6` unit (the reference's `conv_bn_layer`,
/root/reference/ImageCaptioning/model/MobileNetV2.py:88-121) is
    implicit-GEMM conv (MFMA) with batch-norm statistics fused into its epilogue
    -> bn_finalize (per-channel mean / invstd / affine coefficients, running-stat update)
    -> bn_apply (normalise + activation, with the shortcut add of :123-124 folded in).
Activations are NHWC in HBM, so the encoder output [B, S/32, S/32, C] already IS the
[B, K, C] matrix `_img2feature` builds with reshape+transpose
(model_adaAttention_aic.py:193-195).  Backward mirrors it on two lanes (HIP streams): the main lane runs
bn_bwd_reduce -> bn_bwd_apply -> data-gradient GEMM (which applies the ReLU mask of the tensor whose gradient it
completes); the weight-gradient GEMM of the layer, and the whole backward of a projection shortcut, run on the side
lane between device-scope events.  The stem reads a 2x2 space-to-depth copy of the feed (capmi_s2d_stem) and is an
ordinary stride-1 implicit GEMM; strided data gradients are one dense GEMM per output-parity class.
"""
import os

import torch

from . import arch
from ._lib import ACT_CODES, DACT_BITMASK, WGRAD_WS_BYTES, ConvGeom, NtCall, PtrSlot, lib, wgrad_workspace
from .params import stem_s2d

BN_MOMENTUM = 0.9      # fluid.layers.batch_norm defaults; MobileNetV2.py:112-117 overrides neither
BN_EPS = 1e-5


def _p(t):
    if isinstance(t, PtrSlot):          # a pointer the plan re-reads before every run (the caller's feed tensor)
        return t
    return None if t is None else t.data_ptr()


def dgrad_classes(k, stride, pad):
    """Output-parity classes of the data gradient of a k x k / stride-s / pad-p convolution.
    Input pixel hi = s*i + ph receives tap r from output row ho = (hi + p - r)/s only when
    s | (hi + p - r): per class (ph, pw) the taps form a DENSE small convolution over the compact
    [ceil((H-ph)/s)] grid.  Returns [(ph, pw, rmap, qmap)]: class tap r' (offset ho - i = r' - lo)
    reads source tap rmap[r']; classes without taps (no gradient reaches them) are omitted."""
    def taps(ph):
        # valid source taps r with (ph + pad - r) % stride == 0, as offsets d = (ph + pad - r)/stride
        offs = sorted(((ph + pad - r) // stride, r) for r in range(k) if (ph + pad - r) % stride == 0)
        return offs          # [(d, r)] ascending d: ho = i + d
    out = []
    for ph in range(stride):
        th = taps(ph)
        if not th:
            continue
        for pw in range(stride):
            tw = taps(pw)
            if not tw:
                continue
            out.append((ph, pw, [r for _, r in th], [q for _, q in tw], th[0][0], tw[0][0]))
    return [(ph, pw, rm, qm) for ph, pw, rm, qm, _, _ in out]


def dgrad_class_offsets(k, stride, pad):
    """First row/col offset d0 (ho = i + d0 + r') of each class, keyed (ph, pw)."""
    res = {}
    for ph in range(stride):
        oh = sorted((ph + pad - r) // stride for r in range(k) if (ph + pad - r) % stride == 0)
        for pw in range(stride):
            ow = sorted((pw + pad - q) // stride for q in range(k) if (pw + pad - q) % stride == 0)
            if oh and ow:
                res[(ph, pw)] = (oh[0], ow[0], len(oh), len(ow))
    return res


class EncoderRunner:
    def __init__(self, store, B, S, dtype_code, torch_dtype, need_backward):
        self.store, self.enc = store, store.enc
        self.B, self.S = B, S
        self.code, self.tdt = dtype_code, torch_dtype
        self.dev = store.device
        self.need_backward = need_backward
        self.fa_max_rows = int(os.environ.get('CAPMI_BN_FA_MAXM', '0'))     # layers up to this many rows: capmi_bn_finalize_apply (0: never)
        self.fuse_bn_reduce = os.environ.get('CAPMI_BNRED', '0') == '1'     # BN backward sums from the dgrad epilogue (capmi_igemm_nt_bnred): measured slower than the streaming reduce
        enc = self.enc
        # ---- fold `add` ops into the bn_apply of the conv that produces their second operand
        consumers = {}
        for op in enc.ops:
            srcs = (op.src,) if not isinstance(op, arch.Add) else (op.a, op.b)
            for s in srcs:
                consumers[s] = consumers.get(s, 0) + 1
        self._consumers = consumers
        producer = {op.dst: op for op in enc.ops}
        self.fused_add = {}     # conv dst tensor id -> Add op folded into it
        self.skipped = set()
        order = {id(op): i for i, op in enumerate(enc.ops)}
        for op in enc.ops:
            if isinstance(op, arch.Add):
                pb = producer.get(op.b)
                pa = producer.get(op.a)
                a_ready = pa is None or pb is None or order[id(pa)] < order[id(pb)]   # operand a exists when b's bn_apply runs
                if isinstance(pb, arch.ConvBN) and pb.act is None and consumers.get(op.b, 0) == 1 and a_ready:
                    self.fused_add[pb.dst] = op
                    self.skipped.add(id(op))
        # ---- shapes and buffers
        self.shape = {0: (S, S, 3)}
        for op in enc.ops:
            if isinstance(op, arch.ConvBN):
                h, w, _ = self.shape[op.src]
                ho = (h + 2 * op.pad - op.k) // op.stride + 1
                wo = (w + 2 * op.pad - op.k) // op.stride + 1
                self.shape[op.dst] = (ho, wo, op.cout)
            elif isinstance(op, arch.Add):
                self.shape[op.dst] = self.shape[op.a]
            else:
                h, w, c = self.shape[op.src]
                self.shape[op.dst] = ((h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, c)
        z = lambda shape, dt=None: torch.zeros(shape, dtype=dt or self.tdt, device=self.dev)
        self.act, self.grad, self.raw, self.bn = {}, {}, {}, {}
        # saved batch means of every layer in ONE buffer (views below), and a copy of it taken at the head of every forward plan: the
        # shift of the one-pass statistics (capmi_igemm_nt_stat / capmi_bn_stat_apply) -- last step's mean, read by kernels that
        # run while this step's mean is being written
        nmean = sum((op.cout + 7) // 8 * 8 for op in enc.ops if isinstance(op, arch.ConvBN))
        self.mean_all, self.shift_all, moff = z((max(nmean, 8),), torch.float32), z((max(nmean, 8),), torch.float32), 0
        self.pool_idx = {}
        max_elems = 0
        for op in enc.ops:
            if id(op) in self.skipped:
                continue
            h, w, c = self.shape[op.dst]
            out_id = self.fused_add[op.dst].dst if isinstance(op, arch.ConvBN) and op.dst in self.fused_add else op.dst
            self.act[out_id] = z((B, h, w, c))
            if need_backward:
                self.grad[out_id] = z((B, h, w, c))
            if isinstance(op, arch.ConvBN):
                self.raw[op.dst] = z((B, h, w, c))
                # statistics workspace: (mean, M2) of every `part_rows`-row block, written by the conv
                # epilogue (dense convs) or by bn_stats (depthwise); no zeroing needed
                M = B * h * w
                if op.groups > 1:
                    part_rows = lib().capmi_bn_stats_part_rows(M, c, dtype_code)
                else:
                    kk = self.kpad_of(op) if op.src == 0 else op.k * op.k * op.cin
                    part_rows = lib().capmi_igemm_nt_stats_part_rows(M, c, kk, dtype_code)
                nparts = (M + part_rows - 1) // part_rows
                self.bn[op.dst] = dict(stats=z((nparts + 64, c, 2), torch.float32), part_rows=part_rows, mean=self.mean_all[moff:moff + c],
                                       shift=self.shift_all[moff:moff + c], invstd=z((c,), torch.float32), a=z((c,), torch.float32))
                moff += (c + 7) // 8 * 8
                max_elems = max(max_elems, B * h * w * c)
            elif isinstance(op, arch.MaxPool):
                self.pool_idx[op.dst] = z((B, h, w, c), torch.uint8)
        # grad w.r.t. a conv's raw output: a ring of scratch buffers, so that the weight gradient of layer L
        # (side lane) can still read its buffer while the main lane already produces the ones of the next layers
        self.draws = [z((max_elems,)) for _ in range(int(os.environ.get('CAPMI_RING', '6')))] if need_backward else None
        self.draw = self.draws[0] if need_backward else None
        self.overlap_wgrad = True
        # lane of a projection shortcut's backward: 1 = the weight-gradient lane (default), 3 = a lane of its own (measured equal
        # on one GPU: 7.64-7.71 against 7.69-7.70 ms per step -- tools/lane_gaps.py had shown the main lane waiting for it under
        # instrumentation only)
        self.shortcut_lane = int(os.environ.get('CAPMI_SHORTCUT_LANE', '1'))
        self.overlap_forward = os.environ.get('CAPMI_FWD_SIDE', '1') != '0'     # projection shortcuts of the forward pass on the side lane
        self.draw_side = z((max_elems,)) if need_backward else None     # raw-output gradient of a projection shortcut (side lane)
        ws = 0
        for op in enc.ops:
            if isinstance(op, arch.ConvBN):
                h, w, c = self.shape[op.dst]
                ws = max(ws, lib().capmi_bn_bwd_ws_floats(B * h * w, c, dtype_code))
        # capmi_bn_bwd_reduce_spread: eight accumulator rows [8][2C] per batch-norm layer, one buffer for all of them (zeroed by
        # ONE fill per step, model._compile_train); CAPMI_BN_SPREAD=0 keeps the two-stage reduction
        self.bn_spread = need_backward and os.environ.get('CAPMI_BN_SPREAD', '1') != '0'
        # max-pool backward gathered inside the producer's batch-norm backward (capmi_bn_bwd_reduce_pool); needs the accumulator rows
        self.pool_fuse = self.bn_spread and os.environ.get('CAPMI_POOL_FUSE', '1') != '0'
        # ... and forward: capmi_bn_stat_apply + the pool as one launch that never writes the activated tensor (capmi_bn_stat_apply_pool); the
        # backward pair then forms the activation's derivative from the conv output (capmi_bn_bwd_*_pool_x).  Decided here, once, for both passes.
        self.pool_fwd = {}                  # tensor id (the conv + bn + relu output) -> the MaxPool that is its only reader
        if self.pool_fuse and need_backward and self.code == 1 and os.environ.get('CAPMI_POOL_FWD_FUSE', '1') != '0' and os.environ.get('CAPMI_STAT_APPLY', '1') != '0':
            prod_of = {o.dst: o for o in enc.ops if isinstance(o, arch.ConvBN)}
            for o in enc.ops:
                if isinstance(o, arch.MaxPool) and o.src in prod_of and id(prod_of[o.src]) not in self.skipped:
                    pr, c = prod_of[o.src], self.shape[o.src][2]
                    users = sum(1 for u in enc.ops if id(u) not in self.skipped and ((o.src in (u.a, u.b)) if isinstance(u, arch.Add) else u.src == o.src))
                    if (pr.act in ('relu', 'relu6') and users == 1 and pr.groups == 1 and c % 8 == 0 and 256 % (c // 8) == 0 and o.src != enc.out
                            and o.src not in self.fused_add and B * self.shape[o.src][0] * self.shape[o.src][1] > self.fa_max_rows):
                        self.pool_fwd[o.src] = o
        # convolution + batch-norm statistics + finalize through ONE entry point (capmi_igemm_nt_bnfin: the last-arriving workgroup
        # finalizes when the library is built with -DCAPMI_FIN=1).  Measured slower than the dependent launch it removes (DESIGN.md
        # lesson 48): off by default.
        self.bnfin = os.environ.get('CAPMI_BNFIN', '0') != '0'
        # batch-norm BACKWARD sums in the epilogue of the data gradient that completes the layer's output gradient
        # (capmi_igemm_nt_bnsum, igemm.hip EPI 7): the layer's capmi_bn_bwd_reduce_spread launch -- two tensor reads on the
        # dependency chain -- is gone wherever that data gradient is a dense launch with its mask as bits.  CAPMI_BNSUM=0: off.
        self.bnsum = os.environ.get('CAPMI_BNSUM', '1') != '0'
        # forward statistics as sums the convolution's epilogue adds into four accumulator rows per layer (capmi_igemm_nt_stat,
        # igemm.hip EPI 8) + mean / invstd formed in the prologue of the apply launch (capmi_bn_stat_apply): the merge + finalize
        # launch behind every convolution of the forward chain is gone.  bf16 only; deterministic mode switches both entry points
        # back to the exact parts inside the library.  CAPMI_STAT_APPLY=0: conv -> capmi_bn_finalize -> capmi_bn_apply.
        self.stat_apply = dtype_code == 1 and os.environ.get('CAPMI_STAT_APPLY', '1') != '0'
        # activation-derivative bit masks (capmi_bn_apply_mask): one bit per element of every ReLU / ReLU6 tensor, written next to the
        # activated tensor in the forward pass; the data-gradient epilogue that masks the tensor's gradient reads them instead of the
        # tensor, and reads them in FRONT of its main loop (igemm.hip EPI 6, nt_prefetch_mask: a byte per row and lane): a mask-only
        # data gradient's epilogue then issues no load at all, -0.03 ms per step.  bf16 only; CAPMI_MASKBITS=0: the saved output itself.
        self.maskbits = {}
        if (need_backward and dtype_code == 1 and os.environ.get('CAPMI_MASKBITS', '1') != '0' and not self.fuse_bn_reduce
                and os.environ.get('CAPMI_INBN', '0') == '0'):
            for op in enc.ops:
                if isinstance(op, arch.ConvBN) and id(op) not in self.skipped:
                    fa = self.fused_add.get(op.dst)
                    out_id, act = (fa.dst, fa.act) if fa is not None else (op.dst, op.act)
                    h, w, c = self.shape[out_id]
                    # (layers whose finalize rides inside the apply launch -- capmi_bn_finalize_apply, CAPMI_BN_FA_MAXM -- write no bits:
                    # their data gradients keep reading the saved output)
                    # (... and a tensor whose only reader is a max pool is never written at all: capmi_bn_stat_apply_pool)
                    if (act in ('relu', 'relu6') and c % 8 == 0 and c >= 32 and out_id != enc.out and B * h * w > self.fa_max_rows
                            and out_id not in self.pool_fwd):
                        self.maskbits[out_id] = z((B * h * w * c // 8,), torch.uint8)
        self.bn_acc, off = {}, 0
        for op in enc.ops:
            if isinstance(op, arch.ConvBN) and id(op) not in self.skipped:
                self.bn_acc[op.dst] = off
                off += 16 * self.shape[op.dst][2]
        self.bn_acc_all = z((max(off, 4),), torch.float32) if self.bn_spread else None
        self.fwd_rows, off = {}, 0          # accumulator rows [4][2C] of the forward statistics (capmi_igemm_nt_stat), zeroed at the head of every forward plan
        for op in enc.ops:
            if isinstance(op, arch.ConvBN) and id(op) not in self.skipped:
                self.fwd_rows[op.dst] = off
                off += 8 * self.shape[op.dst][2]
        self.fwd_rows_all = z((max(off, 4),), torch.float32) if self.stat_apply else None
        self.bwd_ws = z((ws,), torch.float32) if need_backward else None  # partial sums of bn_bwd_reduce (scratch)
        self.bwd_ws_side = z((ws,), torch.float32) if need_backward else None
        # partial sums written by data-gradient epilogues (capmi_igemm_nt_bnred): <= one part per 64 rows (+ class tails)
        red_floats = max([(((B * self.shape[op.dst][0] * self.shape[op.dst][1] + 63) // 64) + 8) * 2 * self.shape[op.dst][2]
                          for op in enc.ops if isinstance(op, arch.ConvBN)] or [0])
        self.red_ws = [z((red_floats,), torch.float32), z((red_floats,), torch.float32)] if need_backward else None
        stem = enc.ops[0]
        h, w, _ = self.shape[stem.dst]
        assert stem.stride == 2 and stem.src == 0, 'the stem is a stride-2 convolution on the image feed'
        # space-to-depth image of the padded feed: the stem becomes a stride-1 implicit GEMM on it (capmi_s2d_stem)
        self.stem_kt, self.stem_cs = stem_s2d(stem.k, stem.cin)
        self.stem_hb, self.stem_wb = h + (stem.k - 1) // 2, w + (stem.k - 1) // 2
        self.s2d = z((B, self.stem_hb, self.stem_wb, self.stem_cs))
        self.out_id = enc.out
        # ---- batch norm in the consumer's operand path (capmi_igemm_nt_bnact): a conv whose input tensor is produced by a
        # conv -> batch_norm -> relu unit and read by nobody else multiplies the producer's RAW output, normalised and
        # activated in its A-operand path -- the producer's bn_apply leaves the forward chain (MobileNetV2.py:88-121: the
        # conv1 -> conv2 -> conv3 links of a bottleneck / inverted-residual unit).  The backward pass still reads the
        # activated tensor (weight-gradient operand, ReLU mask): it is written on the side lane, under the decoder.
        self.inbn = {}                   # id(consumer op) -> producer op
        self.inbn_tensors = set()        # tensors whose main-lane bn_apply is gone
        # CAPMI_INBN: 0 (default) = off; 1 = 1x1 consumers (the LDS-DMA kernel transforms its A fragments: one VALU pass per
        # staged element); 2 = 3x3 consumers on the halo-staged kernel as well (the halo tile is transformed in LDS: 1.9-2.8x
        # the elements of the output tile, in a loop with no MFMA shadow to hide ~50 VALU instructions per piece in).
        # Measured on the bench workload, A/B on one box (DESIGN.md lesson 39, profiles/r03_inbn_ab.txt): level 1 9.08 ms per
        # step against 9.06 without, level 2 9.26 -- bit-identical results, no gain: the bn_apply launches it takes off the
        # forward chain (0.19 ms) come back as slower convolutions.  Kept as a tested alternate, off by default.
        inbn_level = int(os.environ.get('CAPMI_INBN', '0'))
        inbn_max_rows = int(os.environ.get('CAPMI_INBN_MAXM', str(1 << 30)))
        if dtype_code == 1 and inbn_level > 0:
            for op in enc.ops:
                if not isinstance(op, arch.ConvBN) or id(op) in self.skipped or op.groups != 1 or op.src == 0:
                    continue
                P = producer.get(op.src)
                if not isinstance(P, arch.ConvBN) or P.act not in ('relu', 'relu6') or P.dst in self.fused_add or consumers.get(op.src, 0) != 1:
                    continue
                if op.src == self.out_id:
                    continue
                kind = lib().capmi_igemm_nt_bnact_supported(self._conv_geom(op), op.cout, dtype_code)
                ph, pw, _ = self.shape[P.dst]
                if B * ph * pw > inbn_max_rows:      # experiment knob: only layers of at most this many pixels
                    continue
                if kind == 2 or (kind == 1 and inbn_level >= 2):
                    self.inbn[id(op)] = P
                    self.inbn_tensors.add(P.dst)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def kpad_of(op):
        kt, cs = stem_s2d(op.k, op.cin)
        return kt * kt * cs

    def _stem_geom(self, op):
        ho, wo, _ = self.shape[op.dst]
        return ConvGeom(self.B, self.stem_hb, self.stem_wb, self.stem_cs, ho, wo, self.stem_kt, self.stem_kt, 1, 1, 0, self.stem_cs)

    def _conv_geom(self, op):
        hi, wi, _ = self.shape[op.src]
        ho, wo, _ = self.shape[op.dst]
        return ConvGeom(self.B, hi, wi, op.cin, ho, wo, op.k, op.k, op.stride, 1, op.pad, op.cin)

    def _dgrad_geom(self, op):
        hi, wi, _ = self.shape[op.src]
        ho, wo, _ = self.shape[op.dst]
        # "input" = dY [B,Ho,Wo,Cout]; output pixels = forward input pixels; flipped taps
        return ConvGeom(self.B, ho, wo, op.cout, hi, wi, op.k, op.k, 1, op.stride, op.k - 1 - op.pad, op.cout)

    def out_tensor(self):
        return self.act[self.out_id]

    def out_grad(self):
        return self.grad[self.out_id]

    # ------------------------------------------------------------------ forward plan
    def plan_forward(self, plan, image, weights, update_running=True, is_test=False):
        """image: f32 NCHW [B,3,S,S] device tensor (the reference feed), or a _lib.PtrSlot holding its address (re-read
        before every run: the train step then reads the caller's tensor in place).  weights(name) -> tensor
        the kernels read (f32 master or bf16 shadow).  is_test: normalise with the running statistics (the exported
        inference model, infer.py) instead of the batch statistics (every in-training graph, quirks Q3/Q4)."""
        st, B, code = self.store, self.B, self.code
        # a projection shortcut (conv + BN whose output only feeds a fused add) is independent of the
        # branch2a..2c chain of its block: side lane, joined before the bn_apply that adds it
        side_ops, side_out = set(), set()
        if self.overlap_forward:
            producer = {o.dst: o for o in self.enc.ops}
            for o in self.enc.ops:
                fa = self.fused_add.get(o.dst) if isinstance(o, arch.ConvBN) else None
                p = producer.get(fa.a) if fa is not None else None
                if isinstance(p, arch.ConvBN) and p.act is None and p.dst not in self.fused_add and self._consumers.get(p.dst, 0) == 1 \
                        and p.src != 0 and p.groups == 1:
                    side_ops.add(id(p))
        if is_test:
            # inference coefficients (mean = running mean, a = scale / sqrt(running variance + eps)) of EVERY layer in one launch
            import numpy as np
            convs = [o for o in self.enc.ops if isinstance(o, arch.ConvBN) and id(o) not in self.skipped]
            table = np.zeros((len(convs), 6), dtype=np.int64)
            for i, o in enumerate(convs):
                c = self.shape[o.dst][2]
                table[i] = (_p(st.view(o.name + '_bn_scale')), _p(st.state[o.name + '_bn_mean']), _p(st.state[o.name + '_bn_variance']),
                            _p(self.bn[o.dst]['mean']), _p(self.bn[o.dst]['a']), c)
            self.coef_jobs = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(self.dev)
            plan.add('capmi_bn_inference_coef_batched', _p(self.coef_jobs), len(convs), int(table[:, 5].max()), BN_EPS)
        deferred = []               # bn_apply launches moved off the forward chain (operand-path batch norm)
        pooled_here = set()         # MaxPool ops whose output came out of their producer's apply launch (capmi_bn_stat_apply_pool)
        use_sa = self.stat_apply and not is_test and not self.bnfin
        if use_sa:
            plan.add('capmi_cast', _p(self.mean_all), _p(self.shift_all), self.mean_all.numel(), 0)      # (dtype code 0 = f32: a plain copy)
            plan.add('capmi_fill_f32', _p(self.fwd_rows_all), 0.0, self.fwd_rows_all.numel())
        for op in self.enc.ops:
            if id(op) in self.skipped:
                continue
            if isinstance(op, arch.ConvBN):
                ln = 1 if id(op) in side_ops else 0
                if ln:
                    plan.record(('fin', op.name), 0)
                    plan.wait(('fin', op.name), 1)
                ho, wo, c = self.shape[op.dst]
                M = B * ho * wo
                bn = self.bn[op.dst]
                raw = self.raw[op.dst]
                w = weights(op.name + '_weights')
                if is_test and op.groups == 1:
                    # exported inference model: conv -> batch_norm(is_test) -> (add) -> activation as ONE launch -- the
                    # normalisation, the residual and the activation sit in the GEMM epilogue (capmi_igemm_nt_bn)
                    fa = self.fused_add.get(op.dst)
                    if fa is not None and fa.a in side_out:
                        plan.wait(('fout', fa.a), 0)
                    res, out, act = (_p(self.act[fa.a]), self.act[fa.dst], fa.act) if fa is not None else (None, self.act[op.dst], op.act)
                    if op.src == 0:
                        plan.add('capmi_s2d_stem', _p(image), _p(self.s2d), B, op.cin, self.S, self.S, op.pad, self.stem_hb, self.stem_wb,
                                 self.stem_cs, code)
                        xin, g, K = self.s2d, self._stem_geom(op), self.kpad_of(op)
                    else:
                        xin, g, K = self.act[op.src], self._conv_geom(op), op.k * op.k * op.cin
                    plan.add('capmi_igemm_nt_bn', _p(xin), _p(w), _p(out), g, c, K, c, _p(bn['mean']), _p(bn['a']),
                             _p(st.view(op.name + '_bn_offset')), res, c, ACT_CODES[act], code, lane=(0 if fa is not None else ln))
                    if ln and fa is None:
                        plan.record(('fout', op.dst), 1)
                        side_out.add(op.dst)
                    continue
                finalized = False
                # statistics as accumulator rows + finalize inside the apply launch (capmi_igemm_nt_stat / capmi_bn_stat_apply)
                sa_geom = (self._stem_geom(op) if op.src == 0 else self._conv_geom(op)) if op.groups == 1 else None
                sa = (use_sa and op.groups == 1 and id(op) not in self.inbn and op.dst not in self.inbn_tensors and M > self.fa_max_rows
                      and lib().capmi_igemm_nt_stat_supported(sa_geom, c, code) == 1)
                sa_rows = self.fwd_rows_all.data_ptr() + 4 * self.fwd_rows[op.dst] if sa else None
                if op.src == 0:        # stem: space-to-depth of the NCHW feed, then an ordinary stride-1 implicit GEMM
                    plan.add('capmi_s2d_stem', _p(image), _p(self.s2d), B, op.cin, self.S, self.S, op.pad, self.stem_hb, self.stem_wb,
                             self.stem_cs, code)
                    if sa:
                        plan.add('capmi_igemm_nt_stat', _p(self.s2d), _p(w), _p(raw), sa_geom, c, self.kpad_of(op), c, _p(bn['stats']), sa_rows, _p(bn['shift']), code)
                    else:
                        plan.add('capmi_igemm_nt', _p(self.s2d), _p(w), _p(raw), self._stem_geom(op), c, self.kpad_of(op), c, None, None, 0, None, 0,
                                 None if is_test else _p(bn['stats']), 0, 0, 0, code)
                elif op.groups > 1:
                    hi, wi, _ = self.shape[op.src]
                    plan.add('capmi_dwconv3x3_fwd', _p(self.act[op.src]), _p(w), _p(raw), B, hi, wi, c, op.stride, ho, wo, code)
                    if not is_test:
                        plan.add('capmi_bn_stats', _p(raw), M, c, _p(bn['stats']), code)
                elif id(op) in self.inbn and not is_test:
                    # the input is the producer's RAW output; its batch norm + activation ride in the A-operand path
                    P = self.inbn[id(op)]
                    pb = self.bn[P.dst]
                    plan.add('capmi_igemm_nt_bnact', _p(self.raw[P.dst]), _p(w), _p(raw), self._conv_geom(op), c, op.k * op.k * op.cin, c,
                             _p(pb['mean']), _p(pb['a']), _p(st.view(P.name + '_bn_offset')), ACT_CODES[P.act], _p(bn['stats']), code, lane=ln)
                else:
                    g = self._conv_geom(op)
                    K = op.k * op.k * op.cin
                    if not is_test and self.bnfin and M > self.fa_max_rows:
                        # convolution + statistics + finalize as one launch (the last-arriving workgroup finalizes; large grids
                        # run the two calls inside the entry point)
                        plan.add('capmi_igemm_nt_bnfin', _p(self.act[op.src]), _p(w), _p(raw), g, c, K, c, _p(bn['stats']),
                                 _p(st.view(op.name + '_bn_scale')), _p(st.state[op.name + '_bn_mean']), _p(st.state[op.name + '_bn_variance']),
                                 BN_MOMENTUM, BN_EPS, _p(bn['mean']), _p(bn['invstd']), _p(bn['a']), 1 if update_running else 0, code, lane=ln)
                        finalized = True
                    elif sa:
                        plan.add('capmi_igemm_nt_stat', _p(self.act[op.src]), _p(w), _p(raw), g, c, K, c, _p(bn['stats']), sa_rows, _p(bn['shift']), code, lane=ln)
                    else:
                        plan.add('capmi_igemm_nt', _p(self.act[op.src]), _p(w), _p(raw), g, c, K, c, None, None, 0, None, 0,
                                 None if is_test else _p(bn['stats']), 0, 0, 0, code, lane=ln)
                if not is_test and not sa:
                    # small layers (stage 4 / 5 of a ResNet at batch 64): finalize inside the apply launch -- one dependent
                    # ~5 us kernel less on the forward chain; on big layers every one of thousands of apply workgroups would
                    # redo the merge (measured slower, capmi.h)
                    fa_fused = M <= self.fa_max_rows
                    if not fa_fused and not finalized:
                        plan.add('capmi_bn_finalize', _p(bn['stats']), bn['part_rows'], M, c, _p(st.view(op.name + '_bn_scale')),
                                 _p(st.state[op.name + '_bn_mean']), _p(st.state[op.name + '_bn_variance']), BN_MOMENTUM, BN_EPS,
                                 _p(bn['mean']), _p(bn['invstd']), _p(bn['a']), 1 if update_running else 0, lane=ln)
                offset = _p(st.view(op.name + '_bn_offset'))
                fa = self.fused_add.get(op.dst)
                fused_here = (not is_test) and M <= self.fa_max_rows

                def apply(res, out, act, lane, out_id=None):
                    bits = self.maskbits.get(out_id) if (out_id is not None and not is_test) else None
                    pool = self.pool_fwd.get(out_id) if (sa and res is None and not is_test and lane == 0) else None
                    if pool is not None:
                        # conv -> bn -> relu -> max pool: the activated tensor has one reader; it is never written (act[out_id] is the
                        # deterministic mode's scratch), the pool's output and argmax map come out of this launch
                        pho, pwo, _ = self.shape[pool.dst]
                        plan.add('capmi_bn_stat_apply_pool', _p(raw), _p(bn['stats']), bn['part_rows'], sa_rows, _p(bn['shift']), B, ho, wo, c, pho, pwo,
                                 _p(st.view(op.name + '_bn_scale')), offset, _p(st.state[op.name + '_bn_mean']), _p(st.state[op.name + '_bn_variance']), BN_MOMENTUM,
                                 BN_EPS, _p(bn['mean']), _p(bn['invstd']), _p(bn['a']), 1 if update_running else 0, out, _p(self.act[pool.dst]),
                                 _p(self.pool_idx[pool.dst]), act, code, lane=lane)
                        pooled_here.add(id(pool))
                    elif sa:
                        plan.add('capmi_bn_stat_apply', _p(raw), _p(bn['stats']), bn['part_rows'], sa_rows, _p(bn['shift']), M, c, _p(st.view(op.name + '_bn_scale')), offset,
                                 _p(st.state[op.name + '_bn_mean']), _p(st.state[op.name + '_bn_variance']), BN_MOMENTUM, BN_EPS, _p(bn['mean']),
                                 _p(bn['invstd']), _p(bn['a']), 1 if update_running else 0, res, out, _p(bits), act, code, lane=lane)
                    elif bits is not None and not fused_here:
                        plan.add('capmi_bn_apply_mask', _p(raw), _p(bn['mean']), _p(bn['a']), offset, res, out, _p(bits), M, c, act, code, lane=lane)
                    elif fused_here:
                        plan.add('capmi_bn_finalize_apply', _p(bn['stats']), bn['part_rows'], M, c, _p(st.view(op.name + '_bn_scale')), offset,
                                 _p(st.state[op.name + '_bn_mean']), _p(st.state[op.name + '_bn_variance']), BN_MOMENTUM, BN_EPS,
                                 _p(bn['mean']), _p(bn['invstd']), 1 if update_running else 0, _p(raw), res, out, act, code, lane=lane)
                    else:
                        plan.add('capmi_bn_apply', _p(raw), _p(bn['mean']), _p(bn['a']), offset, res, out, M, c, act, code, lane=lane)
                if fa is None and op.dst in self.inbn_tensors and not is_test:
                    # consumed through capmi_igemm_nt_bnact: no bn_apply on the forward chain; the backward pass's copy of
                    # the activated tensor is written on the side lane after the encoder (below)
                    deferred.append((raw, bn, offset, self.act[op.dst], M, c, ACT_CODES[op.act]))
                elif fa is None:
                    apply(None, _p(self.act[op.dst]), ACT_CODES[op.act], ln, op.dst)
                    if ln:
                        plan.record(('fout', op.dst), 1)
                        side_out.add(op.dst)
                else:
                    if fa.a in side_out:
                        plan.wait(('fout', fa.a), 0)
                    apply(_p(self.act[fa.a]), _p(self.act[fa.dst]), ACT_CODES[fa.act], 0, fa.dst)
            elif isinstance(op, arch.Add):
                n = self.act[op.dst].numel()
                plan.add('capmi_add_act', _p(self.act[op.a]), _p(self.act[op.b]), _p(self.act[op.dst]), n, ACT_CODES[op.act], code)
            elif id(op) not in pooled_here:
                hi, wi, c = self.shape[op.src]
                ho, wo, _ = self.shape[op.dst]
                plan.add('capmi_maxpool3x3s2_fwd', _p(self.act[op.src]), _p(self.act[op.dst]), _p(self.pool_idx[op.dst]),
                         B, hi, wi, c, ho, wo, code)
        if deferred and self.need_backward:
            # the activated tensors the backward pass reads (weight-gradient operand, ReLU mask of the data gradient): written on
            # the side lane behind the encoder's last statistics, i.e. under the decoder's forward / backward pass
            lane = 1 if self.overlap_forward else 0
            if lane:
                plan.record(('enc', 'statistics final'), 0)
                plan.wait(('enc', 'statistics final'), 1)
            for raw_t, bn_t, off_t, act_t, M_t, c_t, code_t in deferred:
                plan.add('capmi_bn_apply', _p(raw_t), _p(bn_t['mean']), _p(bn_t['a']), off_t, None, _p(act_t), M_t, c_t, code_t, code, lane=lane)
            if lane:
                plan.record(('enc', 'deferred applies'), 1)

    # ------------------------------------------------------------------ backward plan
    def _backward_order(self):
        """Reversed op order, except that a projection shortcut (a ConvBN whose output only feeds a
        fused add) runs its backward just BEFORE the conv that owns the add: its strided 1x1 data
        gradient (which leaves pixels untouched) then is the first writer of the block input's
        gradient and the dense 1x1 of branch2a the last one, which can apply the ReLU mask."""
        producer = {op.dst: op for op in self.enc.ops}
        order, early = [], set()
        for op in reversed(self.enc.ops):
            if id(op) in self.skipped or id(op) in early:
                continue
            fa = self.fused_add.get(op.dst) if isinstance(op, arch.ConvBN) else None
            if fa is not None:
                p = producer.get(fa.a)
                if isinstance(p, arch.ConvBN) and p.act is None and p.dst not in self.fused_add and self._consumers.get(p.dst, 0) == 1:
                    order.append(p)
                    early.add(id(p))
            order.append(op)
        return order, early

    def plan_backward(self, plan, weights, weights_bwd, segments=None):
        """Expects d(loss)/d(encoder output) in self.out_grad().  weights(name) -> forward filter,
        weights_bwd(name) -> its data-gradient form.  `segments`, if given, is a list that receives (plan_index, param_name)
        marks after each op's parameter gradients are final (used to place all-reduce buckets).

        Gradient buffers hold d(loss)/d(PRE-activation) wherever the last writer is a dense data-gradient
        GEMM ("premasked"): its epilogue applies the activation mask of the tensor it completes, so the
        batch-norm backward kernels of the producer read two tensors instead of three, and the shortcut
        gradient of a fused residual add is the block-output gradient itself -- never copied (`pending`)."""
        assert self.need_backward
        st, B, code = self.store, self.B, self.code
        NONE = ACT_CODES[None]
        producer = {op.dst: op for op in self.enc.ops}
        tensor_act = {}                 # activation applied where the tensor is produced
        for op in self.enc.ops:
            if id(op) in self.skipped:
                continue
            if isinstance(op, arch.ConvBN):
                fa = self.fused_add.get(op.dst)
                tensor_act[fa.dst if fa else op.dst] = fa.act if fa else op.act
            elif isinstance(op, arch.Add):
                tensor_act[op.dst] = op.act
        order, early = self._backward_order()
        last_writer = {}
        for op in order:
            if isinstance(op, arch.Add):
                last_writer[op.a] = last_writer[op.b] = op
            elif op.src != 0:
                last_writer[op.src] = op
        written = {self.out_id}
        premasked = set()
        pending = {}                    # tensor -> buffer that holds its first gradient contribution (alias, not a copy)
        n_out = self.grad[self.out_id].numel()
        if self.inbn_tensors:       # (a no-op when the forward plan was a launch of its own: plans end with their lanes joined)
            plan.wait(('enc', 'deferred applies'), 0)
        if tensor_act.get(self.out_id) is not None:     # the decoder hands over d/d(post-activation): mask it once, in place
            plan.add('capmi_act_bwd', _p(self.grad[self.out_id]), _p(self.act[self.out_id]), _p(self.grad[self.out_id]), 0, n_out,
                     ACT_CODES[tensor_act[self.out_id]], code)
        premasked.add(self.out_id)

        def materialize(t):
            """First contribution of `t` is an alias and the next writer cannot take an addend: copy it."""
            if t in pending:
                buf = pending.pop(t)
                plan.add('capmi_act_bwd', _p(buf), _p(buf), _p(self.grad[t]), 0, self.grad[t].numel(), NONE, code)
                written.add(t)

        canonical = [op for op in reversed(self.enc.ops) if id(op) not in self.skipped]
        marks_due = list(canonical)
        done = set()

        def flush_marks():
            while marks_due and id(marks_due[0]) in done:
                op = marks_due.pop(0)
                if segments is not None and isinstance(op, arch.ConvBN):
                    segments.append((len(plan), op.name))

        def grad_buf(t):
            return self.grad[t] if t in written else pending[t]

        def ensure_premasked(t):
            """Make grad[t] the pre-activation gradient with one in-place pass (only where the last writer could not)."""
            if t not in premasked and tensor_act.get(t) is not None:
                materialize(t)
                plan.add('capmi_act_bwd', _p(self.grad[t]), _p(self.act[t]), _p(self.grad[t]), 0, self.grad[t].numel(),
                         ACT_CODES[tensor_act[t]], code)
            premasked.add(t)

        shortcut_done = set()           # outputs of projection shortcuts whose backward already ran on the aliased gradient
        ring_pos, ring_user = [0], [None] * len(self.draws)
        side_draw_user = [None]         # the projection shortcut whose weight gradient (lane 1) read draw_side last
        side_written = {}               # tensor -> event key: its gradient's first writer ran on the side lane
        conv_of_out = {}                # tensor id -> the ConvBN whose (fused-add) output it is
        for o in self.enc.ops:
            if isinstance(o, arch.ConvBN) and id(o) not in self.skipped:
                f = self.fused_add.get(o.dst)
                conv_of_out[f.dst if f else o.dst] = o
        reduced = {}                    # id(ConvBN) -> (workspace, parts): BN backward sums already taken by a dgrad epilogue
        summed = set()                  # id(ConvBN): sums already in the layer's accumulator rows (capmi_igemm_nt_bnsum)
        pool_fused = {}                 # tensor id -> the MaxPool whose backward its producer's batch-norm backward carries

        def consumers_of(t):
            return sum(1 for o in self.enc.ops if id(o) not in self.skipped
                       and (t in (o.a, o.b) if isinstance(o, arch.Add) else o.src == t))
        ws_busy = [None, None]

        def bnred_targets(t):
            """BN layers that read grad[t] as their dy: the producer of t and, for a block output, the projection shortcut."""
            X = conv_of_out.get(t)
            if X is None:
                return []
            out = [X]
            f = self.fused_add.get(X.dst)
            if f is not None:
                p2 = producer.get(f.a)
                if p2 is not None and id(p2) in early:
                    out.append(p2)
            return out
        for pos, op in enumerate(order):
            if isinstance(op, arch.ConvBN):
                if id(op) in early:     # projection shortcut: its output gradient IS the (masked) block-output gradient
                    nxt = order[pos + 1]
                    blk_out = self.fused_add[nxt.dst].dst
                    ensure_premasked(blk_out)
                    pending[op.dst] = grad_buf(blk_out)
                    premasked.add(op.dst)
                    shortcut_done.add(op.dst)
                # a projection shortcut's whole backward (BN backward, both gradients) is independent of the
                # branch2c..2a chain: it runs on the side lane, between two events
                # ... on a lane of ITS OWN (3): on the weight-gradient lane it queued behind whatever weight gradients that lane
                # still owed, and the main lane then waited for it at the block's first convolution -- 40-200 us per stage
                # (tools/lane_gaps.py: `wait pdone/...` binding).  Only its weight gradient stays on lane 1.
                ln = self.shortcut_lane if (id(op) in early and self.overlap_wgrad) else 0
                if ln:
                    plan.record(('blk', op.name), 0)
                    plan.wait(('blk', op.name), ln)
                    if ln != 1 and side_draw_user[0] is not None:        # draw_side is still the operand of the previous shortcut's weight gradient
                        plan.wait(('wgrad', side_draw_user[0]), ln)
                ho, wo, c = self.shape[op.dst]
                M = B * ho * wo
                bn = self.bn[op.dst]
                raw = self.raw[op.dst]
                # ring slot of this layer's raw-output gradient; its previous user's weight gradient (side lane) must be done
                wl = 1 if self.overlap_wgrad else 0       # lane of the weight-gradient launches
                if ln:
                    draw = self.draw_side
                else:
                    slot = ring_pos[0] % len(self.draws)
                    ring_pos[0] += 1
                    draw = self.draws[slot]
                    if wl and ring_user[slot] is not None:
                        plan.wait(('wgrad', ring_user[slot]), 0)
                    ring_user[slot] = op.name
                fa = self.fused_add.get(op.dst)
                out_id = fa.dst if fa else op.dst
                t_act = fa.act if fa else op.act
                act = NONE if (out_id in premasked or t_act is None) else ACT_CODES[t_act]
                dy = grad_buf(out_id)
                y = self.act[out_id]
                red = st.gview(op.name + '_bn_offset')        # [d offset | d scale] adjacent in the flat buffer
                pool = None
                assert c % 8 == 0 and st.entries[op.name + '_bn_scale'].offset == st.entries[op.name + '_bn_offset'].offset + c
                if id(op) in reduced:
                    slot, parts = reduced.pop(id(op))
                    assert act == NONE
                    plan.add('capmi_bn_bwd_reduce_final', _p(self.red_ws[slot]), parts, c, _p(red))
                    ws_busy[slot] = None
                    spread = None
                elif out_id in pool_fused:
                    # the layer feeds a max pool and nothing else: both kernels gather the pool's input gradient from the pooled
                    # one and the argmax map (capmi_bn_bwd_reduce_pool) -- it is never written
                    pool = pool_fused.pop(out_id)
                    spread = None
                    acc = self.bn_acc_all.data_ptr() + 4 * self.bn_acc[op.dst]
                    pho, pwo, _ = self.shape[pool.dst]
                    pargs = (B, ho, wo, c, pho, pwo, act, code)
                    if out_id in self.pool_fwd and self.stat_apply:
                        # the forward pass never wrote y (capmi_bn_stat_apply_pool): the activation's derivative from the conv output
                        cx = (_p(bn['a']), _p(st.view(op.name + '_bn_offset')))
                        plan.add('capmi_bn_bwd_reduce_pool_x', _p(self.grad[pool.dst]), _p(self.pool_idx[pool.dst]), _p(raw), _p(y), *cx, _p(bn['mean']),
                                 _p(bn['invstd']), _p(self.bwd_ws), _p(red), acc, _p(self.grad[out_id]), *pargs)
                        plan.add('capmi_bn_bwd_apply_pool_x', _p(self.grad[pool.dst]), _p(self.pool_idx[pool.dst]), _p(raw), _p(y), *cx, _p(bn['mean']),
                                 _p(bn['invstd']), _p(st.view(op.name + '_bn_scale')), _p(red), acc, _p(self.grad[out_id]), _p(draw), *pargs)
                    else:
                        plan.add('capmi_bn_bwd_reduce_pool', _p(self.grad[pool.dst]), _p(self.pool_idx[pool.dst]), _p(raw), _p(y), _p(bn['mean']),
                                 _p(bn['invstd']), _p(self.bwd_ws), _p(red), acc, _p(self.grad[out_id]), *pargs)
                        plan.add('capmi_bn_bwd_apply_pool', _p(self.grad[pool.dst]), _p(self.pool_idx[pool.dst]), _p(raw), _p(y), _p(bn['mean']),
                                 _p(bn['invstd']), _p(st.view(op.name + '_bn_scale')), _p(red), acc, _p(self.grad[out_id]), _p(draw), *pargs)
                elif id(op) in summed:
                    # the data gradient that completed dy took the sums in its epilogue: accumulator rows (or, deterministic, red) are final
                    assert act == NONE and not ln
                    summed.discard(id(op))
                    spread = self.bn_acc_all.data_ptr() + 4 * self.bn_acc[op.dst]
                elif self.bn_spread:
                    spread = self.bn_acc_all.data_ptr() + 4 * self.bn_acc[op.dst]
                    plan.add('capmi_bn_bwd_reduce_spread', _p(dy), _p(raw), _p(y), _p(bn['mean']), _p(bn['invstd']),
                             _p(self.bwd_ws_side if ln else self.bwd_ws), _p(red), spread, M, c, act, code, lane=ln)
                else:
                    spread = None
                    plan.add('capmi_bn_bwd_reduce', _p(dy), _p(raw), _p(y), _p(bn['mean']), _p(bn['invstd']), _p(self.bwd_ws_side if ln else self.bwd_ws), _p(red), M, c, act, code, lane=ln)
                dres, dres_acc = None, 0
                if fa is not None and fa.a in shortcut_done:
                    pending.pop(fa.a, None)
                elif fa is not None and fa.a != 0:
                    if act == NONE and fa.a not in written and fa.a not in pending:
                        pending[fa.a] = dy              # shortcut gradient == (masked) block-output gradient
                        if tensor_act.get(fa.a) is None:
                            premasked.add(fa.a)
                    else:
                        materialize(fa.a)
                        dres = self.grad[fa.a]
                        dres_acc = 1 if fa.a in written else 0
                        written.add(fa.a)
                if pool is not None:
                    pass
                elif spread is not None:
                    plan.add('capmi_bn_bwd_apply_spread', _p(dy), _p(raw), _p(y), _p(bn['mean']), _p(bn['invstd']),
                             _p(st.view(op.name + '_bn_scale')), _p(red), spread, _p(draw), 0, _p(dres), dres_acc, M, c, act, code, lane=ln)
                else:
                    plan.add('capmi_bn_bwd_apply', _p(dy), _p(raw), _p(y), _p(bn['mean']), _p(bn['invstd']),
                             _p(st.view(op.name + '_bn_scale')), _p(red), _p(draw), 0, _p(dres), dres_acc, M, c, act, code, lane=ln)
                if wl and ln != wl:
                    plan.record(('dz', op.name), ln)
                    plan.wait(('dz', op.name), wl)
                dwt = st.gview(op.name + '_weights')
                if op.src == 0:
                    plan.add('capmi_igemm_tn_wgrad', _p(self.s2d), _p(draw), _p(dwt), self._stem_geom(op), c, c, self.kpad_of(op),
                             _p(wgrad_workspace(self.dev, wl)), WGRAD_WS_BYTES, code, lane=wl)
                    plan.add('capmi_s2d_stem_mask_grad', _p(dwt), c, op.cin, op.k, self.stem_cs, lane=wl)
                    if wl and not ln:
                        plan.record(('wgrad', op.name), 1)
                elif op.groups > 1:
                    hi, wi, _ = self.shape[op.src]
                    plan.add('capmi_dwconv3x3_bwd_weight', _p(self.act[op.src]), _p(draw), _p(dwt), B, hi, wi, c, op.stride, ho, wo, code, lane=wl)
                    if wl and not ln:
                        plan.record(('wgrad', op.name), 1)
                    materialize(op.src)
                    acc = 1 if op.src in written else 0
                    plan.add('capmi_dwconv3x3_bwd_data', _p(draw), _p(weights(op.name + '_weights')), _p(self.grad[op.src]),
                             B, hi, wi, c, op.stride, ho, wo, acc, code)
                    written.add(op.src)
                else:
                    g = self._conv_geom(op)
                    K = op.k * op.k * op.cin
                    plan.add('capmi_igemm_tn_wgrad', _p(self.act[op.src]), _p(draw), _p(dwt), g, c, c, K, _p(wgrad_workspace(self.dev, wl)), WGRAD_WS_BYTES, code, lane=wl)
                    if wl and not ln:
                        plan.record(('wgrad', op.name), 1)
                    elif wl and ln != wl:
                        plan.record(('wgrad', op.name), wl)
                        side_draw_user[0] = op.name
                    t = op.src
                    dx = self.grad[t]
                    src_act = tensor_act.get(t)
                    is_last = last_writer.get(t) is op
                    classes = None if op.stride == 1 else dgrad_class_offsets(op.k, op.stride, op.pad)
                    covered = classes is None or len(classes) == op.stride * op.stride
                    if not covered:
                        materialize(t)          # pixels outside the classes keep their earlier contribution
                    if not ln and t in side_written:     # the side lane wrote grad[t] first: wait for it
                        plan.wait(side_written.pop(t), 0)
                    addend = dx if t in written else pending.pop(t, None)
                    mask = is_last and covered and src_act is not None
                    ysaved, dact = (_p(self.act[t]), ACT_CODES[src_act]) if mask else (None, 0)
                    if mask and t in self.maskbits:         # the bit mask of the forward pass instead of the tensor itself
                        ysaved, dact = _p(self.maskbits[t]), dact | DACT_BITMASK
                    # this launch completes grad[t] in its final (pre-activation) form: take the BN backward
                    # sums of the layers that consume it in the same epilogue
                    targets = bnred_targets(t) if (self.fuse_bn_reduce and is_last and covered and (mask or src_act is None)) else []
                    launches = []       # (geometry, weight key, Kd)
                    if op.stride == 1:
                        launches.append((self._dgrad_geom(op), op.name + '_weights', op.k * op.k * op.cout))
                    else:
                        hi, wi, _ = self.shape[op.src]
                        for (ph, pw), (d0h, d0w, nkh, nkw) in classes.items():
                            hc, wc = (hi - ph + op.stride - 1) // op.stride, (wi - pw + op.stride - 1) // op.stride
                            if d0h != d0w:
                                raise NotImplementedError('asymmetric parity classes')
                            # ho = i + d0h + r'  <=>  hn = i*1 - pad' + r' with pad' = -d0h
                            launches.append((ConvGeom(B, ho, wo, op.cout, hc, wc, nkh, nkw, 1, 1, -d0h, op.cout, op.stride, ph, pw, hi, wi),
                                             (op.name + '_weights', ph, pw), nkh * nkw * op.cout))
                    part_rows = [lib().capmi_igemm_nt_bnred_part_rows(gd, op.cin, code) for gd, _, _ in launches] if targets else []
                    if targets and (min(part_rows) <= 0 or any(ws_busy[q] is not None for q in range(len(targets)))):
                        targets = []
                    if targets:
                        part_off = 0
                        for (gd, wkey, Kd), pr in zip(launches, part_rows):
                            tg = []
                            for q in range(2):
                                if q < len(targets):
                                    X = targets[q]
                                    bq = self.bn[X.dst]
                                    tg += [_p(self.raw[X.dst]), _p(bq['mean']), _p(bq['invstd']),
                                           self.red_ws[q].data_ptr() + part_off * 2 * op.cin * 4]
                                else:
                                    tg += [None, None, None, None]
                            plan.add('capmi_igemm_nt_bnred', _p(draw), _p(weights_bwd(wkey)), _p(dx), gd, op.cin, Kd, op.cin,
                                     _p(addend), op.cin, ysaved, op.cin, dact, len(targets), *tg, code)
                            part_off += (gd.B * gd.Ho * gd.Wo + pr - 1) // pr
                        assert part_off * 2 * op.cin <= self.red_ws[0].numel()
                        for q, X in enumerate(targets):
                            reduced[id(X)] = (q, part_off)
                            ws_busy[q] = id(X)
                    elif op.stride == 1:
                        gd, wkey, Kd = launches[0]
                        X = conv_of_out.get(t)          # the layer whose output gradient this launch completes
                        if (self.bnsum and self.bn_spread and not ln and mask and t in self.maskbits and X is not None and id(X) not in early
                                and X.dst in self.bn_acc and not (self.pool_fuse and t in pool_fused)
                                and lib().capmi_igemm_nt_bnsum_part_rows(gd, op.cin, code) > 0):
                            bq = self.bn[X.dst]
                            plan.add('capmi_igemm_nt_bnsum', _p(draw), _p(weights_bwd(wkey)), _p(dx), gd, op.cin, Kd, op.cin, _p(addend), op.cin,
                                     ysaved, op.cin, dact, _p(self.raw[X.dst]), _p(bq['mean']), _p(bq['invstd']),
                                     self.bn_acc_all.data_ptr() + 4 * self.bn_acc[X.dst], _p(self.red_ws[0]),
                                     _p(st.gview(X.name + '_bn_offset')), code)
                            summed.add(id(X))
                        else:
                            plan.add('capmi_igemm_nt', _p(draw), _p(weights_bwd(wkey)), _p(dx), gd, op.cin, Kd, op.cin,
                                     None, _p(addend), op.cin, ysaved, op.cin, None, 0, dact, 0, code, lane=ln)
                    else:
                        # strided conv: one dense GEMM per output-parity class over the compact grid,
                        # rows scattered to pixels (s*i+ph, s*j+pw) -- no MFMA on structural zeros; the
                        # classes go out as one grouped launch
                        if not covered and addend is None:      # some pixels get no gradient: start from zero
                            plan.add('capmi_fill_f32', _p(dx), 0.0, dx.numel() * dx.element_size() // 4, lane=ln)
                            addend = dx
                        calls = (NtCall * len(launches))()
                        for cl, (gd, wkey, Kd) in zip(calls, launches):
                            cl.x, cl.w, cl.y, cl.g = _p(draw), _p(weights_bwd(wkey)), _p(dx), gd
                            cl.N, cl.ldw, cl.ldy = op.cin, Kd, op.cin
                            cl.addend, cl.ld_addend = _p(addend), op.cin
                            cl.ysaved, cl.ld_saved, cl.dact = ysaved, op.cin, dact
                        plan.add('capmi_igemm_nt_group', calls, len(launches), code, lane=ln)
                    written.add(t)
                    if ln:
                        plan.record(('pdone', op.name), ln)
                        side_written[t] = ('pdone', op.name)
                    if is_last and (mask or src_act is None):
                        premasked.add(t)
            elif isinstance(op, arch.Add):
                n = self.act[op.dst].numel()
                act = NONE if (op.dst in premasked or op.act is None) else ACT_CODES[op.act]
                src = self.grad[op.dst] if op.dst in written else pending[op.dst]
                for t in (op.a, op.b):
                    materialize(t)
                    plan.add('capmi_act_bwd', _p(src), _p(self.act[op.dst]), _p(self.grad[t]), 1 if t in written else 0, n, act, code)
                    written.add(t)
            else:
                hi, wi, c = self.shape[op.src]
                ho, wo, _ = self.shape[op.dst]
                assert op.src not in written and op.src not in pending
                materialize(op.dst)
                prod = conv_of_out.get(op.src)
                if (self.pool_fuse and prod is not None and prod.dst == op.src and id(prod) not in early and id(prod) not in reduced
                        and tensor_act.get(op.src) in ('relu', 'relu6') and op.dst in written and consumers_of(op.src) == 1
                        and 256 % (c // (8 if code == 1 else 4)) == 0):
                    pool_fused[op.src] = op          # the producer's batch-norm backward gathers from grad[op.dst] itself
                    written.add(op.src)
                    done.add(id(op))
                    flush_marks()
                    continue
                assert op.src not in self.pool_fwd, 'the forward pass never wrote this activated tensor: its backward must take the gathered pair'
                plan.add('capmi_maxpool3x3s2_bwd', _p(self.grad[op.dst]), _p(self.pool_idx[op.dst]), _p(self.grad[op.src]), B, hi, wi, c, ho, wo, code)
                written.add(op.src)
            done.add(id(op))
            flush_marks()
