"""`ImageCaptionModel` facade + execution engine: the drop-in for the reference's model/ package
and the `train_exe.run(...)` step of train.py.

Reference protocol being mirrored (/root/reference/ImageCaptioning/):
  model/model_adaAttention_aic.py:138-212  ImageCaptionModel.build_input / build_network
  train.py:34-58                           training_net() / eval_net()
  train.py:121-127,139,163                 ParallelExecutor(...).run(feed=..., fetch_list=[...])
Feeds: `image` float32 [B,3,S,S] (NCHW), `caption` int64 [B,L].  Fetches: loss float32 [1],
lr float32 [1]; eval `caption` ids float32 [B,Ti] (quirk Q2).  Errors: ValueError on a bad
mode, AssertionError on NaN loss (train.py:140-141) are raised by the callers in train_loop.py.

All arithmetic runs in libcapmi.so (hand-written gfx950 kernels) through static launch plans: the train
step is launched on two HIP streams (main chain + side lane, _lib.Plan), the decode plan replays from a
hipGraph; torch only owns device memory, the current stream, graph capture and torch.distributed.
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, Plan
from .decoder import DecoderRunner, vocab_ld
from .encoder import EncoderRunner
from .optim import ADAM_BETA1, ADAM_BETA2, ADAM_EPS, LRSchedule, adam_lr_t
from .params import ParamStore

_DTYPES = {'f32': (F32, torch.float32), 'bf16': (BF16, torch.bfloat16)}


def _p(t):
    return None if t is None else t.data_ptr()


class Var:
    """Stand-in for a fluid Variable handle: only its name matters (fetch/feed lookup)."""

    def __init__(self, name, shape, dtype):
        self.name, self.shape, self.dtype = name, shape, dtype

    def __repr__(self):
        return 'Var(%s, %s, %s)' % (self.name, self.shape, self.dtype)


class CaptionEngine:
    """Owns parameters, weight shadows, static buffers and launch plans for one device."""

    def __init__(self, cfg, device='cuda:0', use_graph=True, process_group=None):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.CapmiError('CaptionEngine needs a HIP device (got %s); there is no CPU path' % device)
        _lib.lib()                                   # fail loudly if the extension is missing
        self.code, self.tdt = _DTYPES[cfg.get('dtype', 'f32')]
        self.slots = cfg.get('attention', 'singleton') == 'slots'
        if cfg.get('attention', 'singleton') not in ('singleton', 'slots'):
            raise ValueError('不支持{}'.format(cfg['attention']))
        self.store = ParamStore(self.cfg, self.device)
        self.store.init_reference(seed=cfg.get('seed') or 0)
        self.use_graph = use_graph
        self.graph_decoder_forward = False
        self.fuse_optimizer = True      # single rank: Adam + shadow refresh inside the backward plan (side lane)
        self.overlap_lanes = True       # laned plans (side-stream weight gradients) run eagerly on two HIP streams
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.lr_schedule = LRSchedule(cfg.get('lr_decay_strategy'), cfg.get('learning_rate', 5e-5),
                                      cfg.get('sample_count', 0), cfg.get('batch_size', 1), cfg.get('decay_epoch', 0),
                                      cfg.get('warmup_epoch', 3), cfg.get('max_epoch', 10))
        self.step_count = 0
        self._build_shadows()
        # the side lane's HIP stream now, before anything else in the process creates streams (hardware-queue assignment
        # follows creation order: see dp.OverlappedTrainer)
        with torch.cuda.device(self.device):
            Plan._lane_streams({1, 3})
        self._train = {}      # batch size -> compiled train program
        self._eval = {}
        self.shadows_dirty = True

    # ------------------------------------------------------------------ weight shadows
    def _build_shadows(self):
        """bf16 mode: `low` = bf16 copy of the flat parameter buffer (same offsets).  Both modes:
        `wT` = data-gradient forms ([C][kh][kw][N], taps flipped) of every GEMM weight."""
        st = self.store
        self.low = torch.zeros(st.size, dtype=self.tdt, device=self.device) if self.code == BF16 else None
        self.wT_entries = {}
        off = 0
        V, E = self.cfg['vocab'], self.cfg['embed']
        strides = {op.name + '_weights': op.stride for op in st.enc.ops if hasattr(op, 'stride')}
        for name, e in st.entries.items():
            if e.kind == 'conv' and strides.get(name, 1) > 1:
                # strided conv: one weight form per output-parity class of its data gradient (see
                # encoder.dgrad_classes); key = (name, ph, pw)
                from .encoder import dgrad_classes
                n, kh, kw, c = e.kshape
                for (ph, pw, rmap, qmap) in dgrad_classes(kh, strides[name], (kh - 1) // 2):
                    size = c * len(rmap) * len(qmap) * n
                    self.wT_entries[(name, ph, pw)] = (off, (n, kh, kw, c, n), rmap, qmap)
                    off += (size + 7) // 8 * 8
                continue
            if e.kind == 'conv':
                n, kh, kw, c = e.kshape
                spec = (n, kh, kw, c, n)
            elif e.kind == 'fc_w':
                n, k = e.kshape
                spec = (n, 1, 1, k, n)
            elif name == 'word_embedding':
                spec = (V, 1, 1, E, vocab_ld(V))
            else:
                continue
            size = spec[3] * spec[1] * spec[2] * spec[4]
            self.wT_entries[name] = (off, spec, [spec[1] - 1 - r for r in range(spec[1])], [spec[2] - 1 - q for q in range(spec[2])])
            off += (size + 7) // 8 * 8
        self.wT = torch.zeros(off, dtype=self.tdt, device=self.device)
        self.shadow_plan = Plan()
        if self.code == BF16:
            self.shadow_plan.add('capmi_cast', _p(st.flat), _p(self.low), st.size, self.code)
        # one launch for every data-gradient form: job table = one entry per run of 2 64x64 tiles
        jobs = []
        for key, (o, (n, kh, kw, c, ldt), rmap, qmap) in self.wT_entries.items():
            name = key if isinstance(key, str) else key[0]
            total = len(rmap) * len(qmap) * ((c + 63) // 64) * ((ldt + 63) // 64)
            for first in range(0, total, 2):
                jobs.append((st.entries[name].offset, o, n, kh, kw, c, ldt, first, len(rmap), len(qmap),
                             (list(rmap) + [0] * 4)[:4], (list(qmap) + [0] * 4)[:4]))
        table = np.zeros(len(jobs), dtype=np.dtype([('src', '<i8'), ('dst', '<i8'), ('N', '<i4'), ('kh', '<i4'), ('kw', '<i4'),
                                                    ('C', '<i4'), ('ldt', '<i4'), ('first', '<i4'), ('okh', '<i4'), ('okw', '<i4'),
                                                    ('rmap', 'i1', (4,)), ('qmap', 'i1', (4,))]))
        assert table.dtype.itemsize == 56
        for i, j in enumerate(jobs):
            table[i] = j
        self.dgrad_jobs = torch.from_numpy(table.view(np.uint8)).to(self.device)
        self.dgrad_job_src = [j[0] for j in jobs]           # flat offsets of the jobs' source weights (non-decreasing)
        self.shadow_plan.add('capmi_weight_dgrad_form_batched', _p(st.flat), _p(self.wT), _p(self.dgrad_jobs), len(jobs), self.code)

    def W(self, name):
        return self.store.view(name) if self.low is None else self.store.view(name, self.low)

    def WT(self, key):
        """Data-gradient form of weight `key` (a name, or (name, ph, pw) for a parity class of a strided conv)."""
        o, (n, kh, kw, c, ldt), rmap, qmap = self.wT_entries[key]
        return self.wT[o:o + c * len(rmap) * len(qmap) * ldt]

    def refresh_shadows(self):
        self.shadow_plan.run(self._stream())
        self.shadows_dirty = False
        self.shadow_version = getattr(self, 'shadow_version', 0) + 1

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------ parameters in / out
    def load_reference_params(self, params):
        self.store.load_reference(params)
        self.shadows_dirty = True

    def export_reference_params(self):
        torch.cuda.synchronize(self.device)
        return self.store.export_reference()

    def export_reference_grads(self):
        torch.cuda.synchronize(self.device)
        return self.store.export_reference_grads()

    # ------------------------------------------------------------------ program construction
    def _compile_train(self, B):
        cfg, S = self.cfg, self.cfg['image_size']
        need_enc_bwd = bool(cfg['encoder_trainable'])
        enc = EncoderRunner(self.store, B, S, self.code, self.tdt, True)
        K = enc.shape[enc.out_id][0] * enc.shape[enc.out_id][1]
        T = cfg['sentence_length'] - 1                                         # model_adaAttention_aic.py:66
        dec = DecoderRunner(self.store, B, K, T, self.code, self.tdt, self.slots, True)
        image = torch.zeros((B, 3, S, S), dtype=torch.float32, device=self.device)
        # the forward pass as two plans: the encoder part has lanes (two HIP streams, launched eagerly), the
        # decoder part (2T small recurrent launches) has none and replays from a hipGraph -- keeps the host
        # time per step well below the device time
        fwd_enc, fwd_dec, fwd, bwd = Plan(), Plan(), Plan(), Plan()
        # the feed's address is an argument slot the plan re-reads before every run: a float32 device tensor of the caller is read
        # in place by the first kernel (38.5 MB less to copy in front of every step at cfg 2), anything else is staged in `image`
        image_slot = _lib.PtrSlot(image.data_ptr())
        enc.plan_forward(fwd_enc, image_slot, self.W)
        # Where the encoder forward has a side lane anyway (projection shortcuts), the gradient buffer is zeroed there, under
        # the forward pass, instead of in front of the backward pass on the critical chain; the forward plan's final join
        # orders it before the first gradient write.  (A lane-less forward plan stays lane-less: it replays from a hipGraph.)
        zero_in_fwd = fwd_enc.has_lanes
        if zero_in_fwd:
            head = Plan()
            # the data-gradient weight forms of the parameters the last step's optimizer wrote: only the backward pass reads them,
            # so they are rebuilt HERE, on the side lane under the forward pass -- the backward pass is bound by HBM bytes
            # (16.8 GB in ~4.2 ms), the forward pass is not (DESIGN.md lesson 60)
            if os.environ.get('CAPMI_FORMS_IN_FWD', '1') != '0' and need_enc_bwd:
                head.add('capmi_weight_dgrad_form_batched', _p(self.store.flat), _p(self.wT), _p(self.dgrad_jobs), len(self.dgrad_job_src), self.code, lane=1)
            head.add('capmi_fill_f32', _p(self.store.grad), 0.0, self.store.size, lane=1)
            if enc.bn_acc_all is not None:          # accumulator rows of capmi_bn_bwd_reduce_spread
                head.add('capmi_fill_f32', _p(enc.bn_acc_all), 0.0, enc.bn_acc_all.numel(), lane=1)
            head.record(('zero', 'gradient buffers'), 1)
            head.extend(fwd_enc)
            fwd_enc = head
        dec.plan_forward(fwd_dec, enc.out_tensor(), self.W)
        fwd.extend(fwd_enc)
        fwd.extend(fwd_dec)
        if zero_in_fwd:     # one launch table for the whole step (dp._fuse_step): the first gradient write follows the side lane's fills
            bwd.wait(('zero', 'gradient buffers'), 0)      # (dropped when forward and backward are launched as two tables: those end joined)
        if not zero_in_fwd:
            bwd.add('capmi_fill_f32', _p(self.store.grad), 0.0, self.store.size)
            if enc.bn_acc_all is not None:
                bwd.add('capmi_fill_f32', _p(enc.bn_acc_all), 0.0, enc.bn_acc_all.numel())
        dec.plan_backward(bwd, enc.out_tensor(), enc.out_grad(), self.W, self.WT)
        marks = []
        n_dec = len(bwd)
        if need_enc_bwd:
            enc.plan_backward(bwd, self.W, self.WT, marks)
        prog = dict(B=B, enc=enc, dec=dec, image=image, image_slot=image_slot, fwd=fwd, fwd_parts=[fwd_enc, fwd_dec], bwd=bwd, graph=None, marks=marks, n_dec=n_dec,
                    forms_in_fwd=any(c[1] == 'capmi_weight_dgrad_form_batched' for c in fwd_enc.calls))
        return prog

    def _compile_eval(self, B, beam=1, is_test=False, scored=False):
        cfg, S = self.cfg, self.cfg['image_size']
        enc = EncoderRunner(self.store, B, S, self.code, self.tdt, False)
        enc.overlap_forward = False         # one lane: the whole decode (encoder + Ti sequential steps) replays from a hipGraph
        K = enc.shape[enc.out_id][0] * enc.shape[enc.out_id][1]
        Ti = cfg['infer_max_length']
        dec = DecoderRunner(self.store, B, K, max(1, beam), self.code, self.tdt, self.slots, False)
        image = torch.zeros((B, 3, S, S), dtype=torch.float32, device=self.device)
        out = torch.zeros((B, Ti), dtype=torch.float32, device=self.device)
        plan_enc, plan = Plan(), Plan()             # two plans: decode_pipelined runs them on two streams
        # in-training eval graph: batch statistics AND running-stat update (quirk Q3); is_test: the exported
        # inference model (infer.py) -- running statistics, nothing updated
        enc.plan_forward(plan_enc, image, self.W, update_running=not is_test, is_test=is_test)
        if beam <= 1 and not scored:
            es = dec.Hbuf.element_size()
            for hb, cb in zip(dec.Hbufs, dec.Cbufs):                                 # zero state of every LSTM layer (:63)
                plan.add('capmi_fill_f32', _p(hb), 0.0, B * dec.H * es // 4)
                plan.add('capmi_fill_f32', _p(cb), 0.0, B * dec.H * es // 4)
            dec.plan_greedy(plan, enc.out_tensor(), self.W, out, Ti)
        else:
            dec.plan_beam(plan, enc.out_tensor(), self.W, out, Ti, max(1, beam))
        return dict(B=B, enc=enc, dec=dec, image=image, out=out, plan_enc=plan_enc, plan_dec=plan, graph=None, beam=beam, scored=scored)

    def _run_captured(self, prog, key, plans):
        """Replays `plans` from a hipGraph captured on first use (static shapes and pointers).

        A plan with lanes (weight gradients on a side stream) is launched eagerly instead: hipGraph
        replays its branches one after another, two HIP streams really overlap them (measured: +7 %),
        and the host keeps ahead of the device either way."""
        if self.use_graph and self.overlap_lanes and any(p.has_lanes for p in plans) and os.environ.get('CAPMI_LANES', '1') != '0':
            for i, p in enumerate(plans):
                if p.has_lanes:
                    p.run(self._stream())
                else:
                    self._run_captured(prog, '%s/%d' % (key, i), [p])
            return
        if not self.use_graph:
            for p in plans:
                p.run(self._stream())
            return
        if prog.get(key) is None:
            # warm-up outside capture so lazy module loading does not happen inside it
            for p in plans:
                p.run(self._stream())
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            try:
                # thread_local: other threads (e.g. the RCCL watchdog) may touch the runtime during capture
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    for p in plans:
                        p.run(self._stream(), side=False)
                prog[key] = g
            except Exception as e:                    # stay on the HIP path, just without the graph
                import sys
                print('capmi: hipGraph capture failed (%s); replaying the launch plan eagerly' % e, file=sys.stderr)
                torch.cuda.synchronize(self.device)
                prog[key] = False
            return            # the warm-up run already produced this call's results
        if prog[key] is False:
            for p in plans:
                p.run(self._stream())
            return
        prog[key].replay()

    # ------------------------------------------------------------------ feeds
    @staticmethod
    def _as_tensor(x, dtype, device):
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        return t.to(device=device, dtype=dtype, non_blocking=True)

    def _feed_train(self, prog, image, caption):
        cfg = self.cfg
        B, dec = prog['B'], prog['dec']
        img = self._as_tensor(image, torch.float32, self.device)
        cap = self._as_tensor(caption, torch.int64, self.device)
        if tuple(img.shape) != tuple(prog['image'].shape):
            raise ValueError('image feed must be %s, got %s' % (tuple(prog['image'].shape), tuple(img.shape)))
        if tuple(cap.shape) != (B, cfg['sentence_length']):
            raise ValueError('caption feed must be %s, got %s' % ((B, cfg['sentence_length']), tuple(cap.shape)))
        # Laned plans are launched eagerly (capmi_plan_run patches the slot); a plan replayed from a hipGraph has its pointers
        # frozen at capture and reads the staging tensor
        # (the plan that holds the image read: the encoder part when the decoder forward is replayed from its own hipGraph)
        img_plan = prog['fwd_parts'][0] if self.graph_decoder_forward else prog['fwd']
        eager = (not self.use_graph) or (img_plan.has_lanes and self.overlap_lanes and os.environ.get('CAPMI_LANES', '1') != '0')
        if eager and img.is_contiguous() and os.environ.get('CAPMI_FEED_COPY', '0') != '1':
            prog['image_slot'].value = img.data_ptr()       # the caller's own float32 device tensor (or the device copy just made of a host
            prog['image_ref'] = img                         # feed): read in place; kept alive until the next feed -- the caller must not rewrite it before the step ran
        else:
            prog['image'].copy_(img)
            prog['image_slot'].value = prog['image'].data_ptr()
            prog['image_ref'] = None
        if not cap.is_contiguous():
            cap = cap.contiguous()
        prog['caption_ref'] = cap
        # source = caption[:, :-1] (:164) and target = caption[:, 1:] (:163), time-major (:60), in one launch
        _lib.call('capmi_caption_feed', cap.data_ptr(), dec.ids.data_ptr(), dec.tgt.data_ptr(), B, cfg['sentence_length'], self._stream())

    # ------------------------------------------------------------------ public steps
    def forward_backward(self, image, caption):
        """One forward + backward on this rank's batch; gradients land in store.grad.  Returns the
        device loss tensor (float32 [1], named 'loss' in the reference, :182)."""
        B = int(image.shape[0])
        prog = self._train.get(B)
        if prog is None:
            prog = self._train[B] = self._compile_train(B)
        if self.shadows_dirty:
            self.refresh_shadows()
        self._feed_train(prog, image, caption)
        # decoder forward from a hipGraph costs ~0.1 ms of replay gaps per step and saves ~0.8 ms of host time;
        # the host (7-8 ms of enqueue per 11.4 ms step) is not the limit, so everything laned goes out eagerly
        fwd = prog['fwd_parts'] if self.graph_decoder_forward else [prog['fwd']]
        self._run_captured(prog, 'graph', fwd + [prog['bwd']])
        return prog['dec'].loss

    def forward_loss(self, image, caption):
        """Forward only (no gradient, running statistics still updated as in the train graph)."""
        B = int(image.shape[0])
        prog = self._train.get(B)
        if prog is None:
            prog = self._train[B] = self._compile_train(B)
        if self.shadows_dirty:
            self.refresh_shadows()
        self._feed_train(prog, image, caption)
        prog['fwd'].run(self._stream())
        return prog['dec'].loss

    def optimizer_step(self):
        """Paddle-form Adam over the trainable slice of the flat buffers (IC/train.py:26-31,45)."""
        cfg = self.cfg
        lr = self.lr_schedule.value(self.step_count)
        self.step_count += 1
        clip = float(cfg['gradient_clip']) if cfg.get('gradient_clip') else 0.0
        st = self.store
        _lib.call('capmi_adam', _p(st.flat), _p(st.grad), _p(st.adam_m), _p(st.adam_v), st.trainable_size,
                  adam_lr_t(lr, self.step_count), ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, 1.0 / self.world, self._stream())
        self.shadows_dirty = True
        return lr

    def grad16(self):
        """bf16 staging copy of the trainable slice of the gradient buffer (same offsets): the payload of the data-parallel
        all-reduce when the buckets travel in bf16 (dp.OverlappedTrainer, capmi_allreduce_bucket_bf16)."""
        if getattr(self, '_grad16', None) is None:
            self._grad16 = torch.zeros((self.store.trainable_size + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
        return self._grad16

    def optimizer_range(self, b, e, lr_t, stream, tail=False, g16=None):
        """Adam + shadow refresh of flat range [b, e) on `stream` (a raw HIP stream): what optimizer_step +
        refresh_shadows do for the whole buffer, for one all-reduce bucket.  tail=True also refreshes the
        shadows of everything behind e (non-trainable state).  g16: read the gradients from this bf16 buffer."""
        st, cfg = self.store, self.cfg
        clip = float(cfg['gradient_clip']) if cfg.get('gradient_clip') else 0.0
        if e > b and g16 is not None:
            _lib.call('capmi_adam_g16', st.flat.data_ptr() + b * 4, g16.data_ptr() + b * 2, st.adam_m.data_ptr() + b * 4,
                      st.adam_v.data_ptr() + b * 4, e - b, lr_t, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, 1.0 / self.world, stream)
        elif e > b:
            _lib.call('capmi_adam', st.flat.data_ptr() + b * 4, st.grad.data_ptr() + b * 4, st.adam_m.data_ptr() + b * 4,
                      st.adam_v.data_ptr() + b * 4, e - b, lr_t, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, 1.0 / self.world, stream)
        hi = st.size if tail else e
        if self.low is not None and hi > b:
            _lib.call('capmi_cast', st.flat.data_ptr() + b * 4, self.low.data_ptr() + b * self.low.element_size(), hi - b, self.code, stream)
        j0 = next((i for i, s in enumerate(self.dgrad_job_src) if s >= b), len(self.dgrad_job_src))
        j1 = next((i for i, s in enumerate(self.dgrad_job_src) if s >= hi), len(self.dgrad_job_src))
        if j1 > j0:
            _lib.call('capmi_weight_dgrad_form_batched', _p(st.flat), _p(self.wT), self.dgrad_jobs.data_ptr() + j0 * 56, j1 - j0, self.code, stream)

    def allreduce_grads(self):
        """Sum of the per-rank gradients (ParallelExecutor's AllReduce strategy, IC/train.py:121-124);
        the 1/N CoeffNumDevice scale is applied inside the Adam kernel."""
        if self.world > 1:
            torch.distributed.all_reduce(self.store.grad[:self.store.trainable_size], group=self.pg)

    def plan_adam(self, plan, b, e, lrt, lane, grad_scale=1.0, g16=None, shadow=False):
        """Adam over flat range [b, e) as a plan entry; lrt: a ctypes.c_float re-read before every run; g16: gradients
        from this bf16 buffer (same offsets) instead of the f32 gradient buffer; shadow: write the bf16 weight shadow of the
        range in the same pass (capmi_adam_shadow; returns True when it did, so the caller can drop the range's capmi_cast)."""
        st, cfg = self.store, self.cfg
        clip = float(cfg['gradient_clip']) if cfg.get('gradient_clip') else 0.0
        if e > b and shadow and g16 is None and self.low is not None and self.code == BF16:
            plan.add('capmi_adam_shadow', st.flat.data_ptr() + b * 4, st.grad.data_ptr() + b * 4, st.adam_m.data_ptr() + b * 4,
                     st.adam_v.data_ptr() + b * 4, self.low.data_ptr() + b * 2, e - b, lrt, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, grad_scale, lane=lane)
            return True
        if e > b and g16 is not None:
            plan.add('capmi_adam_g16', st.flat.data_ptr() + b * 4, g16.data_ptr() + b * 2, st.adam_m.data_ptr() + b * 4,
                     st.adam_v.data_ptr() + b * 4, e - b, lrt, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, grad_scale, lane=lane)
        elif e > b:
            plan.add('capmi_adam', st.flat.data_ptr() + b * 4, st.grad.data_ptr() + b * 4, st.adam_m.data_ptr() + b * 4,
                     st.adam_v.data_ptr() + b * 4, e - b, lrt, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, clip, grad_scale, lane=lane)

    def plan_shadow(self, plan, b, e, lane, cast=True, forms=True):
        """Refresh of the bf16 shadow (cast) and of the data-gradient weight forms (forms) whose source lies in flat range [b, e)."""
        st = self.store
        if not forms:
            if cast and self.low is not None and e > b:
                plan.add('capmi_cast', st.flat.data_ptr() + b * 4, self.low.data_ptr() + b * self.low.element_size(), e - b, self.code, lane=lane)
            return
        if cast and self.low is not None and e > b:
            plan.add('capmi_cast', st.flat.data_ptr() + b * 4, self.low.data_ptr() + b * self.low.element_size(), e - b, self.code, lane=lane)
        j0 = next((i for i, s in enumerate(self.dgrad_job_src) if s >= b), len(self.dgrad_job_src))
        j1 = next((i for i, s in enumerate(self.dgrad_job_src) if s >= e), len(self.dgrad_job_src))
        if j1 > j0:
            plan.add('capmi_weight_dgrad_form_batched', _p(st.flat), _p(self.wT), self.dgrad_jobs.data_ptr() + j0 * 56, j1 - j0, self.code, lane=lane)

    def _build_fused_bwd(self, prog):
        """Backward plan with the optimizer inside (single-rank training): once ~90 % of the parameters have
        their final gradients, Adam + the shadow refresh of that range run on the side lane under the rest
        of the backward pass; only the small remainder is updated after the last weight gradient."""
        import ctypes
        st, bwd, cfg = self.store, prog['bwd'], self.cfg
        total = st.trainable_size
        # cut points (fractions of the flat parameter vector; the LAST one is the classic 90 % cut): the optimizer of the range
        # between two cuts goes out as soon as its gradients are final.  One cut = one big update under the bandwidth-bound
        # stage-3 / stage-2 backward; several = the decoder's and the deep stages' parameters are updated under the
        # latency-bound stage-5 / stage-4 backward instead (CAPMI_OPT_CUTS, lesson 49).
        fracs = [float(f) for f in os.environ.get('CAPMI_OPT_CUTS', '0.9').split(',') if f]
        cuts = []                       # (plan index, flat offset), increasing
        for frac in fracs:
            for idx, opname in prog['marks']:
                e = st.entries[opname + '_bn_scale']
                off = e.offset + (int(e.kshape[0]) + 7) // 8 * 8
                if off >= frac * total and off < total:
                    if not cuts or (idx > cuts[-1][0] and off > cuts[-1][1]):
                        cuts.append((idx, off))
                    break
        cut_idx, cut = cuts[-1] if cuts else (None, total)
        lrt = ctypes.c_float(0.0)

        forms_here = not prog.get('forms_in_fwd', False)      # the data-gradient weight forms are (re)built under the NEXT forward pass

        wrote_shadow = [False]

        def optimizer_range(plan, b, e, lane):
            wrote_shadow[0] = bool(self.plan_adam(plan, b, e, lrt, lane, shadow=True))

        def shadow_range(plan, b, e, lane):
            # (always called right behind optimizer_range of the same range: the bf16 shadow came out of capmi_adam_shadow then;
            # parameters behind the trainable range never change, their shadow dates from refresh_shadows)
            self.plan_shadow(plan, b, min(e, total) if wrote_shadow[0] else e, lane, cast=not wrote_shadow[0], forms=forms_here)

        fused = Plan()

        def piece(i0, i1):
            q = Plan()
            q.calls, q._keep, q.has_lanes = bwd.calls[i0:i1], bwd._keep[i0:i1], bwd.has_lanes
            return q

        if cut_idx is not None and bwd.has_lanes:
            at, lo = 0, 0
            for n, (idx, off) in enumerate(cuts):
                fused.extend(piece(at, idx))
                fused.record(('opt', 'head', n), 0)
                fused.wait(('opt', 'head', n), 1)
                if any(getattr(fn, 'lane', 0) == 3 for fn, _, _ in bwd.calls[:idx] if fn is not None):
                    fused.record(('opt', 'head3', n), 3)        # (batch-norm gradients of a projection shortcut are written on lane 3)
                    fused.wait(('opt', 'head3', n), 1)
                optimizer_range(fused, lo, off, 1)
                shadow_range(fused, lo, off, 1)
                at, lo = idx, off
            fused.extend(piece(at, None))
        else:
            cut = 0
            fused.extend(piece(0, None))
        if bwd.has_lanes:
            fused.record(('opt', 'tail'), 1)            # the last weight gradient (side lane) is done
            fused.wait(('opt', 'tail'), 0)
        optimizer_range(fused, cut, total, 0)
        shadow_range(fused, cut, st.size, 0)
        prog['bwd_opt'], prog['lrt'] = fused, lrt
        return fused

    def train_step(self, image, caption):
        """feed -> fwd -> bwd -> (all-reduce) -> Adam; returns (loss tensor [1], lr float)."""
        lanes_on = self.overlap_lanes and os.environ.get('CAPMI_LANES', '1') != '0'
        # (a captured graph would freeze this step's learning rate: the fused plan is only ever launched eagerly)
        if self.world == 1 and self.pg is None and self.fuse_optimizer and (lanes_on or not self.use_graph):
            B = int(image.shape[0])
            prog = self._train.get(B)
            if prog is None:
                prog = self._train[B] = self._compile_train(B)
            if self.shadows_dirty:
                self.refresh_shadows()
            if 'bwd_opt' not in prog:
                self._build_fused_bwd(prog)
            self._feed_train(prog, image, caption)
            lr = self.lr_schedule.value(self.step_count)
            self.step_count += 1
            prog['lrt'].value = adam_lr_t(lr, self.step_count)
            fwd = prog['fwd_parts'] if self.graph_decoder_forward else [prog['fwd']]
            self._run_captured(prog, 'graph_opt', fwd + [prog['bwd_opt']])
            return prog['dec'].loss, lr
        loss = self.forward_backward(image, caption)
        self.allreduce_grads()
        lr = self.optimizer_step()
        self.refresh_shadows()
        return loss, lr

    def _eval_prog(self, B, beam, is_test, scored=False, slot=0):
        key = (B, int(beam), bool(is_test)) + (('scored',) if scored else ()) + (('slot', int(slot)) if slot else ())
        prog = self._eval.get(key)
        if prog is None:
            prog = self._eval[key] = self._compile_eval(B, int(beam), bool(is_test), scored)
        return prog

    def _eval_feed(self, prog, image):
        img = self._as_tensor(image, torch.float32, self.device)
        if tuple(img.shape) != tuple(prog['image'].shape):
            raise ValueError('image feed must be %s, got %s' % (tuple(prog['image'].shape), tuple(img.shape)))
        prog['image'].copy_(img)

    def _eval_init(self, prog):
        dec, B, beam = prog['dec'], prog['B'], prog['beam']
        dec.refresh_stacked(self.W, (getattr(self, 'shadow_version', 0), self.step_count, id(self.store.flat)))
        dec.ids[:max(1, beam) * B].fill_(self.cfg['start_idx'])                   # :56-58
        if beam > 1 or prog['scored']:
            dec.beam_score[0].fill_(-1e30)
            dec.beam_score[0][0].zero_()

    def decode(self, image, beam=1, is_test=False, scored=False):
        """Decode of the eval graph: float32 ids [B, infer_max_length] (quirk Q2).  beam = 1: the reference's greedy
        loop (:119-123); beam > 1: beam search (build-defined, see DecoderRunner.plan_beam), best hypothesis
        returned.  is_test: batch norm on the running statistics, as in the exported inference model (infer.py);
        the default is the in-training eval graph (batch statistics, running stats updated: quirk Q3).  scored: run a
        beam of one through the beam-search plan too, so that decode_scores() has the greedy caption's log-probability."""
        scored = bool(scored) and beam <= 1
        prog = self._eval_prog(int(image.shape[0]), beam, is_test, scored)
        if self.shadows_dirty:
            self.refresh_shadows()
        self._eval_feed(prog, image)
        self._eval_init(prog)
        self._run_captured(prog, 'graph', [prog['plan_enc'], prog['plan_dec']])
        return prog['out']

    def decode_pipelined(self, batches, beam=1, depth=4, decoders=3, graph=False):
        """The infer.py path (`is_test` batch norm) over a sequence of image batches as a two-stage pipeline: the encoder of
        a later batch -- long kernels that fill the chip -- runs on one HIP stream under the decoder steps of `decoders`
        earlier batches on streams of their own (higher priority: 11 short dependent launches per word on a few dozen
        workgroups each; beside an encoder each of them waits for room, which is why more than one decoder is in flight).
        `depth` copies of the program (activation buffers, decoder state) take the batches in turn; the encoder of batch
        i + depth waits for the decoder of batch i.  Same ids as decode(): the kernels and their order within a batch are
        the same, only their neighbours on the chip differ.  Returns one [B, infer_max_length] float32 tensor per batch
        (copies: a slot's output buffer is reused `depth` batches later); the caller's stream has waited for all of them on
        return.  graph: walk the launch plans (default: one foreign call per stage, the host stays ahead) or replay captured
        graphs (hipGraph launches behind cross-stream waits run late on this runtime: slower whenever fewer than three
        decoders hide it).  is_test only: the in-training eval graph updates the running statistics (quirk Q3), which orders
        its batches.  Measured at BASELINE configs[4] (batch 128, beam 5, 64 batches; tools/decode_pipe.py): one batch at a
        time 5.8 ms; plan walks: 3 copies / 2 decoders 4.46-4.50, 4 / 3 4.30-4.32 (29 700 captions/s), 6 / 3 4.32-4.36; graph
        replay: 3 / 2 5.9, 3 / 3 5.2, 4 / 3 4.42.  A fourth decoder stream is a fifth stream on four hardware queues and
        shares one with the encoder (6.6 ms) -- hence at most three."""
        decoders = min(3, max(1, int(decoders)))
        depth = max(decoders, int(depth))
        cur = torch.cuda.current_stream(self.device)
        if self.shadows_dirty:
            self.refresh_shadows()
        if getattr(self, '_pipe_streams', None) is None:
            self._pipe_streams = [torch.cuda.Stream(device=self.device)]
        prio = int(os.environ.get('CAPMI_PIPE_PRIORITY', '-1'))
        while len(self._pipe_streams) < 1 + decoders:
            self._pipe_streams.append(torch.cuda.Stream(device=self.device, priority=prio))
        E, Ds = self._pipe_streams[0], self._pipe_streams[1:1 + decoders]
        for st in [E] + Ds:
            st.wait_stream(cur)
        use_graph = bool(graph) and self.use_graph

        def run(prog, key, plan, st):
            if use_graph:
                self._run_captured(prog, key, [plan])
            else:
                plan.run(st.cuda_stream)
        dec_done = [None] * depth
        outs = []
        for i, image in enumerate(batches):
            prog = self._eval_prog(int(image.shape[0]), beam, True, slot=1 + i % depth)
            D = Ds[i % decoders]
            with torch.cuda.stream(E):
                if dec_done[i % depth] is not None:
                    E.wait_event(dec_done[i % depth])
                self._eval_feed(prog, image)
                run(prog, 'graph_enc', prog['plan_enc'], E)
                enc_done = torch.cuda.Event()
                enc_done.record(E)
            if torch.is_tensor(image) and image.is_cuda:
                image.record_stream(E)
            with torch.cuda.stream(D):
                D.wait_event(enc_done)
                self._eval_init(prog)
                run(prog, 'graph_dec', prog['plan_dec'], D)
                outs.append(prog['out'].clone())
                outs[-1].record_stream(cur)             # allocated on the decoder's stream, read on the caller's
                dec_done[i % depth] = torch.cuda.Event()
                dec_done[i % depth].record(D)
        for st in [E] + Ds:
            cur.wait_stream(st)
        return outs

    def check_sync(self):
        """Raises CapmiError if a grid barrier inside a persistent kernel gave up waiting in any step since the last call
        (the sticky word of every launch slot is read back: synchronises the device).  train_loop.train and Executor.run
        call it every step; callers that drive train_step / OverlappedTrainer.train_step themselves (bench.py, tools/) call
        it once after their timed loop -- a timed-out barrier means the losses, weights and rates of that loop are invalid."""
        for prog in self._train.values():
            prog['dec'].check_sync()

    def decode_scores(self, B, beam, is_test=False):
        """Scores (sum of log-probabilities) of the best hypothesis of the last beam decode of this shape (beam = 1: of the
        last decode(..., scored=True))."""
        key = (B, int(beam), bool(is_test)) + (('scored',) if beam <= 1 else ())
        return self._eval[key]['dec'].beam_final_score[0]


class ImageCaptionModel:
    """Same construction protocol as the reference class (model_adaAttention_aic.py:138-203)."""

    def __init__(self, cfg=None, engine=None, **engine_kw):
        from .config_compat import default_cfg
        self.cfg = default_cfg() if cfg is None else cfg
        self._engine = engine
        self._engine_kw = engine_kw

    @property
    def engine(self):
        if self._engine is None:
            self._engine = CaptionEngine(self.cfg, **self._engine_kw)
        return self._engine

    def build_input(self, mode='train'):
        if mode not in ['train', 'eval']:
            raise ValueError('不支持{}'.format(mode))                         # :144-145
        S = self.cfg['image_size']
        img = Var('image', [-1, 3, S, S], 'float32')                          # :146
        if mode == 'train':
            caption = Var('caption', [-1, self.cfg['sentence_length']], 'int64')   # :149
            return {'img': img, 'caption': caption}, [img, caption]
        return {'img': img}, [img]

    def build_network(self, mode='train', **kwargs):
        if mode not in ['train', 'eval']:
            raise ValueError('不支持{}'.format(mode))                         # :154-155
        if mode == 'train':
            return Var('loss', [1], 'float32')                                # :182
        return Var('caption', [-1, self.cfg['infer_max_length']], 'float32')  # :185-189 (float ids, Q2)

    @staticmethod
    def first_init(places):
        pass                                                                  # :201-203


class Executor:
    """`fluid.ParallelExecutor`-shaped driver: run(feed=..., fetch_list=[...]) (train.py:139,163)."""

    def __init__(self, model, lr_var=None):
        self.model = model
        self.lr_var = lr_var or Var('learning_rate', [1], 'float32')

    def run(self, feed, fetch_list):
        names = [f.name if isinstance(f, Var) else f for f in fetch_list]
        eng = self.model.engine
        out = []
        if 'loss' in names:
            loss, lr = eng.train_step(feed['image'], feed['caption'])
            eng.check_sync()               # (the loss fetch below synchronises anyway, as train_exe.run does: train.py:139)
            res = {'loss': loss.detach().cpu().numpy().astype(np.float32),
                   self.lr_var.name: np.array([lr], dtype=np.float32)}
        elif 'caption' in names:
            res = {'caption': eng.decode(feed['image']).detach().cpu().numpy()}
        else:
            raise ValueError('unknown fetch %r' % (names,))
        for n in names:
            out.append(res[n])
        return out
