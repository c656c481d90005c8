"""Data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

Semantics are ParallelExecutor's defaults (/root/reference/ImageCaptioning/train.py:121-124,
the reference's only parallel mechanism): every rank runs the
same program on its own slice of the batch, normalises its loss by its OWN mask count, the
per-parameter gradients are SUMMED across ranks (ReduceStrategy.AllReduce) after being scaled
by 1/N (GradientScaleStrategy.CoeffNumDevice -- applied here inside the Adam kernel), batch-norm
statistics stay per rank.

The flat gradient buffer is ordered in backward-completion order (params.py), so a bucket is a
contiguous slice.  `GradBuckets` cuts it at parameter boundaries into ~bucket_bytes pieces;
`OverlappedTrainer` replays backward in segments and launches each bucket's all-reduce on a
side stream as soon as its segment has been enqueued -- and Adam + the weight-shadow refresh of the bucket
right behind it on the same stream -- so xGMI traffic and the optimizer hide under the remaining encoder
backward.  xGMI is point-to-point (7 links per GPU), ring all-reduce time is set by one
link, hence few large buckets (default 32 MiB) rather than per-tensor calls.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK from the launcher."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('CAPMI_FORCE_DP', '0') in ('', '0'):
        return None, 0, 1, 0
    # CAPMI_FORCE_DP=1: a one-rank group, to run the N > 1 code path (RCCL calls, bucket streams) on a one-GPU box
    os.environ.setdefault('MASTER_PORT', '29533')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        # 'nccl' is RCCL on ROCm; CAPMI_DIST_BACKEND=gloo rehearses the N > 1 path where RCCL cannot run (two ranks on one GPU)
        backend = os.environ.get('CAPMI_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if local >= ndev:
            if backend != 'gloo':       # RCCL refuses two ranks on one device, late and obscurely: say it here
                raise RuntimeError('rank %d (LOCAL_RANK %d) has no GPU of its own: %d device(s) visible, WORLD_SIZE %d; several ranks '
                                   'may share a GPU only for a gloo rehearsal (CAPMI_DIST_BACKEND=gloo)' % (rank, local, ndev, world))
            local %= ndev               # gloo rehearsal on a box with fewer GPUs than ranks
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist.group.WORLD, rank, world, local


class GradBuckets:
    """Contiguous [begin, end) element ranges of the flat gradient buffer, cut at `cut_points`
    (element offsets where a backward segment ends) so that each bucket is >= bucket_bytes.

    The LAST bucket's all-reduce is the only one nothing can hide (backward has ended): it is cut down to the
    smallest tail of >= tail_bytes that the cut points allow (the early encoder layers, whose gradients come last,
    hold few parameters), instead of whatever remainder the greedy pass leaves (up to 2 x bucket_bytes)."""

    def __init__(self, total, cut_points, bucket_bytes=32 << 20, elem_bytes=4, tail_bytes=2 << 20):
        cuts = sorted(set(c for c in cut_points if 0 < c < total)) + [total]
        self.ranges = []
        begin = 0
        for c in cuts:
            if (c - begin) * elem_bytes >= bucket_bytes or c == total:
                if c > begin:
                    self.ranges.append((begin, c))
                begin = c
        if self.ranges and tail_bytes:
            b, e = self.ranges[-1]
            inner = [c for c in cuts if b < c < e and (e - c) * elem_bytes >= tail_bytes]
            if inner and (e - b) * elem_bytes > 2 * tail_bytes:
                c = inner[-1]                      # the latest cut that still leaves tail_bytes
                self.ranges[-1:] = [(b, c), (c, e)]

    def __iter__(self):
        return iter(self.ranges)

    def __len__(self):
        return len(self.ranges)


def bucket_cut_points(store):
    """Flat offsets at which everything below is final once the backward pass has got that far: the decoder's parameters first,
    then the encoder's layers from last to first (params.ParamStore lays the flat buffer out in exactly that order; a layer's
    group ends with its batch-norm scale).  The data-parallel step may only cut a bucket at one of these."""
    import numpy as np
    cuts = [store.decoder_size]
    for name, e in store.entries.items():
        if name.endswith('_bn_scale') and e.offset >= store.decoder_size:
            cuts.append(e.offset + (int(np.prod(e.kshape)) + 7) // 8 * 8)
    return [c for c in cuts if c <= store.trainable_size]


def bucket_plan(store, bucket_bytes=32 << 20, elem_bytes=4):
    """The gradient buckets of a model (GradBuckets over bucket_cut_points): what OverlappedTrainer exchanges per step,
    whatever the device -- the plan depends on the parameter layout only."""
    return GradBuckets(store.trainable_size, bucket_cut_points(store), bucket_bytes, elem_bytes)


def allreduce_flat(flat, buckets, group=None, async_op=False):
    """Sum-all-reduce `flat` bucket by bucket.  Returns the work handles when async."""
    works = []
    for b, e in buckets:
        w = dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


class NativeComm:
    """The RCCL communicator behind capmi_allreduce_bucket (C ABI, include/capmi.h): one per process, on the current
    device.  Rank 0's unique id travels through the torch.distributed group that launched the job (host channel only);
    after that the gradient buckets never pass through torch.distributed.  `ok` is False -- with the reason in `why` --
    when the library could not set the communicator up or its self-test (a sum of ones must equal the world size on
    every rank) failed; the caller then keeps the torch.distributed all-reduce (the same RCCL underneath)."""

    def __init__(self, pg, rank, world, device):
        import ctypes
        from . import _lib
        self.ok, self.why, self.comm = False, '', ctypes.c_void_p()
        L = _lib.lib()
        # The id and the per-rank verdicts travel through the rendezvous STORE (host TCP), not through a collective of the
        # group: torch.distributed then never has to build a communicator of its own just to carry 128 bytes.
        store = dist.distributed_c10d._get_default_store()
        NativeComm._serial = getattr(NativeComm, '_serial', 0) + 1
        NativeComm._made_on = getattr(NativeComm, '_made_on', set()) | {str(device)}
        key = 'capmi/comm/%d/' % NativeComm._serial
        if rank == 0:
            buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
            if L.capmi_comm_unique_id(buf) != 0:
                self.why = _lib.last_error()
                store.set(key + 'id', b'')
            else:
                store.set(key + 'id', bytes(bytearray(buf)))
        raw = bytes(store.get(key + 'id'))
        if len(raw) != _lib.COMM_ID_BYTES:
            self.why = self.why or 'rank 0 could not create the RCCL unique id'
            return
        rc = L.capmi_comm_init(ctypes.byref(self.comm), world, rank, raw)
        if rc != 0:
            self.why = _lib.last_error()
        # Every rank publishes the verdict of its capmi_comm_init BEFORE any collective runs on the new communicator, and the
        # self-test all-reduce below is entered only when EVERY rank reports success: a rank whose set-up failed would
        # otherwise leave the others waiting inside ncclAllReduce for ever (no time-out there).
        store.set(key + 'init/%d' % rank, b'1' if rc == 0 else b'0')
        if not all(bytes(store.get(key + 'init/%d' % r)) == b'1' for r in range(world)):
            self.why = self.why or 'another rank failed in capmi_comm_init'
            if rc == 0:
                self.close()
            return
        # self-test on the current stream: a sum of ones must equal the world size on every rank
        t = torch.ones(1024, dtype=torch.float32, device=device)
        rc = L.capmi_allreduce_bucket(self.comm, t.data_ptr(), t.numel(), torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
        if rc != 0:
            self.why = _lib.last_error()
        elif not bool((t == float(world)).all()):
            rc, self.why = 1, 'self-test: sum of ones over %d ranks gave %r' % (world, float(t[0]))
        store.set(key + 'ok/%d' % rank, b'1' if rc == 0 else b'0')          # all ranks take the same path
        self.ok = all(bytes(store.get(key + 'ok/%d' % r)) == b'1' for r in range(world))
        if not self.ok and not self.why:
            self.why = 'another rank failed the self-test of its communicator'

    @staticmethod
    def _serial_of(device):
        """True when this process has already created a communicator on `device` (lane creation must come first)."""
        return str(device) in getattr(NativeComm, '_made_on', set())

    def count(self):
        """(ranks, this rank) as RCCL reports them for the communicator (ncclCommCount / ncclCommUserRank)."""
        import ctypes
        from . import _lib
        n, r = ctypes.c_int(0), ctypes.c_int(0)
        if not self.comm or _lib.lib().capmi_comm_count(self.comm, ctypes.byref(n), ctypes.byref(r)) != 0:
            return None
        return n.value, r.value

    def close(self):
        from . import _lib
        if self.comm:
            _lib.lib().capmi_comm_destroy(self.comm)
            self.comm = None
        self.ok = False


class OverlappedTrainer:
    """Train step with the gradient all-reduce overlapped with backward (one instance per rank).

    Backward is cut where a bucket of the flat gradient buffer becomes final (after the decoder,
    then after encoder layers from last to first); each segment is its own hipGraph.  After a
    segment is enqueued, an event is recorded on the compute stream and the bucket's all-reduce is
    issued on a side stream behind that event, followed on the same stream by Adam and the shadow refresh of
    that bucket's parameters (`CaptionEngine.optimizer_range`): the optimizer, too, runs under the remaining
    backward pass; the next forward waits for the side stream."""

    def __init__(self, engine, bucket_bytes=32 << 20, bucket_dtype=None):
        self.eng = engine
        self.bucket_bytes = bucket_bytes
        self.active = engine.world > 1 or (engine.pg is not None and os.environ.get('CAPMI_FORCE_DP', '0') not in ('', '0'))
        # Payload type of the gradient exchange.  'f32' (the default of every engine): the reference's precision -- ParallelExecutor
        # all-reduces f32 gradients (train.py:121-124).  'bf16' (opt-in: bucket_dtype='bf16' / CAPMI_BUCKET_DTYPE=bf16; SURVEY.md
        # section 8(e) budgets the exchange in bf16: 73 MB at BASELINE cfg 2 instead of 146): the bucket's f32 gradients are cast
        # into a bf16 staging buffer by the bucket's producer lane, the ring sums in bf16 (every hop rounds), Adam widens them
        # again (capmi_adam_g16).  It stays opt-in until a run on more than one GPU has shown loss parity with the f32 exchange:
        # the one-rank and gloo tests cannot (round-3 advisor finding).
        self.bucket_dtype = bucket_dtype or os.environ.get('CAPMI_BUCKET_DTYPE') or 'f32'
        if self.bucket_dtype not in ('f32', 'bf16'):
            raise ValueError('bucket_dtype must be f32 or bf16, got %r' % (self.bucket_dtype,))
        # CAPMI_COMM_PRIORITY=-1 puts the bucket stream (all-reduce + the bucket's optimizer) above the backward kernels;
        # on one rank that costs 0.6 % (the optimizer then pre-empts the critical lane), untested on several GPUs
        prio = int(os.environ.get('CAPMI_COMM_PRIORITY', '0'))
        self.comm_stream = torch.cuda.Stream(device=engine.device, priority=prio) if self.active else None
        self._progs = {}
        # backend nccl (= RCCL): the collective goes through the C ABI (capmi_allreduce_bucket) as rows of ONE launch table per
        # step; any other backend (the gloo rehearsals) keeps torch.distributed and the per-segment replay below
        self.native_comm = None
        if self.active and dist.get_backend(engine.pg) == 'nccl' and os.environ.get('CAPMI_NATIVE_COMM', '1') != '0':
            # The plan lanes' HIP streams FIRST.  ncclCommInitRank creates streams of its own; lanes created after it came out
            # multiplexed onto the hardware queues in a way that made every kernel of the three-lane step run ~2x slower
            # (one rank: 24-25 ms per step, with or without the all-reduce rows; 9.2 ms with the lanes created first -- the
            # single-rank fused step takes 9.27 on the same box)
            from ._lib import Plan
            with torch.cuda.device(engine.device):
                if NativeComm._serial_of(engine.device) and 2 not in Plan._side.get(engine.device.index or 0, {}).get('streams', {}):
                    # a communicator of this process already exists on the device and lane 2 does not: the lane would be created
                    # BEHIND ncclCommInitRank -- the slow hardware-queue assignment described above.  Refuse it loudly.
                    raise RuntimeError('dp.OverlappedTrainer: an RCCL communicator was created on %s before the plan lanes; build the '
                                       'trainer (or call _lib.Plan._lane_streams({1, 2})) before any NativeComm' % engine.device)
                Plan._lane_streams({1, 2, 3})
            nc = NativeComm(engine.pg, dist.get_rank(engine.pg), dist.get_world_size(engine.pg), engine.device)
            if nc.ok and os.environ.get('CAPMI_NATIVE_COMM') == '2':       # timing experiment: communicator created, not used
                self._unused_comm = nc
            elif nc.ok:
                self.native_comm = nc
            else:
                import sys
                print('capmi: RCCL communicator behind the C ABI unavailable (%s); gradient buckets go through torch.distributed'
                      % nc.why, file=sys.stderr)

    def _prepare(self, B):
        from ._lib import Plan
        eng = self.eng
        prog = eng._train.get(B)
        if prog is None:
            prog = eng._train[B] = eng._compile_train(B)
        st, bwd = eng.store, prog['bwd']
        # (plan index, flat offset) pairs at which everything below the offset is final
        cuts = [(prog['n_dec'], st.decoder_size)]
        for call_idx, opname in prog['marks']:
            e = st.entries[opname + '_bn_scale']                 # last entry of the op's group
            cuts.append((call_idx, e.offset + (int(e.kshape[0]) + 7) // 8 * 8))
        total = st.trainable_size
        cuts = [(ci, off) for ci, off in cuts if off <= total]
        assert all(a[1] <= b[1] and a[0] <= b[0] for a, b in zip(cuts, cuts[1:])), 'gradient order != backward order'
        buckets = GradBuckets(total, [off for _, off in cuts], self.bucket_bytes)
        assert buckets.ranges == bucket_plan(st, self.bucket_bytes).ranges, 'the backward plan marks and the parameter layout disagree on the cut points'
        index_of = {off: ci for ci, off in cuts}
        segs, start = [], 0
        for (b, e) in buckets:
            stop = len(bwd) if e == total else index_of[e]
            sub = Plan()
            sub.calls, sub._keep, sub.has_lanes = bwd.calls[start:stop], bwd._keep[start:stop], bwd.has_lanes
            segs.append((sub, (b, e), {}))
            start = stop
        assert start == len(bwd)
        P = dict(prog=prog, segs=segs, fwd_graph={})
        if self.native_comm is not None:
            P['step'], P['lrt'] = self._fuse_step(prog, segs)
        return P

    def _fuse_step(self, prog, segs):
        """The whole data-parallel step as ONE launch plan on three lanes: forward, then per bucket the backward segment
        (lanes 0/1) followed -- on lane 2, behind events of both -- by capmi_allreduce_bucket, Adam (1/N folded in) and the
        shadow refresh of that bucket's parameters.  Nothing on lanes 0/1 ever waits for lane 2 inside a step: the
        all-reduces and the optimizer hide under the remaining backward pass; the plan's final join makes the next
        forward wait for the refreshed weights."""
        import ctypes
        from ._lib import Plan
        eng = self.eng
        st = eng.store
        lrt = ctypes.c_float(0.0)
        total = st.trainable_size
        g16 = eng.grad16() if self.bucket_dtype == 'bf16' else None
        from ._lib import BF16
        step = Plan()
        step.extend(prog['fwd'])
        for i, (sub, (b, e), _) in enumerate(segs):
            step.extend(sub)
            step.record(('bucket', i, 0), 0)
            step.wait(('bucket', i, 0), 2)
            if sub.has_lanes:                       # the segment's weight gradients (lane 1) are part of the bucket
                step.record(('bucket', i, 1), 1)
                step.wait(('bucket', i, 1), 2)
                if any(getattr(fn, 'lane', 0) == 3 for fn, _, _ in sub.calls if fn is not None):      # ... and a projection shortcut's batch-norm gradients (lane 3)
                    step.record(('bucket', i, 3), 3)
                    step.wait(('bucket', i, 3), 2)
            if g16 is not None:                     # bf16 payload: cast -> all-reduce -> Adam reads the bf16 sum
                step.add('capmi_cast', st.grad.data_ptr() + b * 4, g16.data_ptr() + b * 2, e - b, BF16, lane=2)
                step.add('capmi_allreduce_bucket_bf16', self.native_comm.comm, g16.data_ptr() + b * 2, e - b, lane=2)
            else:
                step.add('capmi_allreduce_bucket', self.native_comm.comm, st.grad.data_ptr() + b * 4, e - b, lane=2)
            wrote = eng.plan_adam(step, b, e, lrt, 2, grad_scale=1.0 / eng.world, g16=g16, shadow=True)
            # (the weight forms of the data gradients are rebuilt under the next forward pass when the forward plan carries them)
            eng.plan_shadow(step, b, e if wrote else (st.size if e == total else e), 2, cast=not wrote, forms=not prog.get('forms_in_fwd', False))
        return step, lrt

    def describe(self, B):
        """What the data-parallel step of batch B exchanges, for the bench record: ranks as RCCL counts them, bucket sizes in MB
        (of the payload type), payload type, path."""
        P = self._progs.get(B)
        es = 2 if self.bucket_dtype == 'bf16' else 4
        d = dict(bucket_dtype=self.bucket_dtype, path='capmi_allreduce_bucket (RCCL, one launch table per step)' if self.native_comm is not None
                 else 'torch.distributed all_reduce per bucket')
        if P is not None:
            d['buckets_mb'] = [round((e - b) * es / 1e6, 2) for _, (b, e), _ in P['segs']]
        if self.native_comm is not None:
            c = self.native_comm.count()
            d['rccl_nranks'], d['rccl_rank'] = (c if c is not None else (None, None))
        return d

    def exposed_allreduce_ms(self, image, caption, repeats=5):
        """How long the step's LAST bucket (the only all-reduce nothing can hide: the backward pass has ended) keeps the step
        waiting: milliseconds between the end of lanes 0 / 1 and the end of lane 2 (communication + the bucket's optimizer), from
        timing events recorded at the tails of the lanes of `repeats` steps.  Native (C ABI) path only."""
        import ctypes
        from . import _lib
        if self.native_comm is None:
            return None
        L = _lib.lib()
        B = int(image.shape[0])
        eng = self.eng
        out = []
        for _ in range(repeats):
            self.train_step(image, caption)
            # the plan ends with its lanes joined; re-run with events at the tails: lane 2's last row is the tail bucket's shadow refresh
            P = self._progs[B]
            side = _lib.Plan._side[torch.cuda.current_device()]
            evs = {}
            for lane, s in [(0, eng._stream())] + [(l, st.value) for l, st in side['streams'].items()]:
                e = ctypes.c_void_p()
                L.capmi_event_create_timed(ctypes.byref(e))
                evs[lane] = (e, s)
            start = ctypes.c_void_p()
            L.capmi_event_create_timed(ctypes.byref(start))
            torch.cuda.synchronize()
            L.capmi_event_record(start, eng._stream())
            from .optim import adam_lr_t
            lr = eng.lr_schedule.value(eng.step_count)
            eng.step_count += 1
            P['lrt'].value = adam_lr_t(lr, eng.step_count)
            eng._feed_train(P['prog'], image, caption)
            P['step'].run(eng._stream(), tail_events={l: e for l, (e, _) in evs.items()})
            torch.cuda.synchronize()
            ms = ctypes.c_float(0.0)
            ends = {}
            for lane, (e, _) in evs.items():
                if L.capmi_event_elapsed_ms(start, e, ctypes.byref(ms)) == 0:
                    ends[lane] = float(ms.value)
            if 2 in ends:
                out.append(max(0.0, ends[2] - max(ends.get(0, 0.0), ends.get(1, 0.0))))
            for e, _ in list(evs.values()) + [(start, None)]:
                L.capmi_event_destroy(e)
        out.sort()
        return out[len(out) // 2] if out else None

    def check_sync(self):
        """CaptionEngine.check_sync for the engine this trainer drives: call it after a loop of train_step calls (the
        step itself never synchronises)."""
        self.eng.check_sync()

    def train_step(self, image, caption):
        eng = self.eng
        if not self.active:
            return eng.train_step(image, caption)
        B = int(image.shape[0])
        P = self._progs.get(B)
        if P is None:
            P = self._progs[B] = self._prepare(B)
        prog = P['prog']
        if eng.shadows_dirty:
            eng.refresh_shadows()
        eng._feed_train(prog, image, caption)
        from .optim import adam_lr_t
        if 'step' in P:         # RCCL through the C ABI: one launch table per step
            lr = eng.lr_schedule.value(eng.step_count)
            eng.step_count += 1
            P['lrt'].value = adam_lr_t(lr, eng.step_count)
            P['step'].run(eng._stream())
            eng.shadows_dirty = False
            return prog['dec'].loss, lr
        cur = torch.cuda.current_stream(eng.device)
        eng._run_captured(P['fwd_graph'], 'g', prog['fwd_parts'] if eng.graph_decoder_forward else [prog['fwd']])
        grad = eng.store.grad
        lr = eng.lr_schedule.value(eng.step_count)
        eng.step_count += 1
        lr_t = adam_lr_t(lr, eng.step_count)
        total = eng.store.trainable_size
        for sub, (b, e), holder in P['segs']:
            eng._run_captured(holder, 'g', [sub])
            ev = holder.get('event')
            if ev is None:
                ev = holder['event'] = torch.cuda.Event()      # one event per segment, re-recorded every step
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                g16 = None
                if self.bucket_dtype == 'bf16':
                    from . import _lib
                    g16 = eng.grad16()
                    _lib.call('capmi_cast', grad.data_ptr() + b * 4, g16.data_ptr() + b * 2, e - b, _lib.BF16, self.comm_stream.cuda_stream)
                    dist.all_reduce(g16[b:e], op=dist.ReduceOp.SUM, group=eng.pg)
                else:
                    dist.all_reduce(grad[b:e], op=dist.ReduceOp.SUM, group=eng.pg)
                # the bucket's parameters are final for this step: Adam + shadow refresh right behind its
                # all-reduce, on the communication stream, under the rest of the backward pass
                eng.optimizer_range(b, e, lr_t, self.comm_stream.cuda_stream, tail=(e == total), g16=g16)
        cur.wait_stream(self.comm_stream)
        eng.shadows_dirty = False
        return prog['dec'].loss, lr
