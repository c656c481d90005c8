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
    if torch.cuda.is_available():
        local %= torch.cuda.device_count()          # rehearsal on a box with fewer GPUs than ranks
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        # 'nccl' is RCCL on ROCm; CAPMI_DIST_BACKEND=gloo rehearses the N > 1 path where RCCL cannot run (two ranks on one GPU)
        backend = os.environ.get('CAPMI_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
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


def allreduce_flat(flat, buckets, group=None, async_op=False):
    """Sum-all-reduce `flat` bucket by bucket.  Returns the work handles when async."""
    works = []
    for b, e in buckets:
        w = dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


class OverlappedTrainer:
    """Train step with the gradient all-reduce overlapped with backward (one instance per rank).

    Backward is cut where a bucket of the flat gradient buffer becomes final (after the decoder,
    then after encoder layers from last to first); each segment is its own hipGraph.  After a
    segment is enqueued, an event is recorded on the compute stream and the bucket's all-reduce is
    issued on a side stream behind that event, followed on the same stream by Adam and the shadow refresh of
    that bucket's parameters (`CaptionEngine.optimizer_range`): the optimizer, too, runs under the remaining
    backward pass; the next forward waits for the side stream."""

    def __init__(self, engine, bucket_bytes=32 << 20):
        self.eng = engine
        self.bucket_bytes = bucket_bytes
        self.active = engine.world > 1 or (engine.pg is not None and os.environ.get('CAPMI_FORCE_DP', '0') not in ('', '0'))
        # CAPMI_COMM_PRIORITY=-1 puts the bucket stream (all-reduce + the bucket's optimizer) above the backward kernels;
        # on one rank that costs 0.6 % (the optimizer then pre-empts the critical lane), untested on several GPUs
        prio = int(os.environ.get('CAPMI_COMM_PRIORITY', '0'))
        self.comm_stream = torch.cuda.Stream(device=engine.device, priority=prio) if self.active else None
        self._progs = {}

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
        index_of = {off: ci for ci, off in cuts}
        segs, start = [], 0
        for (b, e) in buckets:
            stop = len(bwd) if e == total else index_of[e]
            sub = Plan()
            sub.calls, sub._keep, sub.has_lanes = bwd.calls[start:stop], bwd._keep[start:stop], bwd.has_lanes
            segs.append((sub, (b, e), {}))
            start = stop
        assert start == len(bwd)
        return dict(prog=prog, segs=segs, fwd_graph={})

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
        cur = torch.cuda.current_stream(eng.device)
        eng._run_captured(P['fwd_graph'], 'g', prog['fwd_parts'] if eng.graph_decoder_forward else [prog['fwd']])
        grad = eng.store.grad
        lr = eng.lr_schedule.value(eng.step_count)
        eng.step_count += 1
        lr_t = adam_lr_t(lr, eng.step_count)
        total = eng.store.trainable_size
        for sub, (b, e), holder in P['segs']:
            eng._run_captured(holder, 'g', [sub])
            ev = torch.cuda.Event()
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(grad[b:e], op=dist.ReduceOp.SUM, group=eng.pg)
                # the bucket's parameters are final for this step: Adam + shadow refresh right behind its
                # all-reduce, on the communication stream, under the rest of the backward pass
                eng.optimizer_range(b, e, lr_t, self.comm_stream.cuda_stream, tail=(e == total))
        cur.wait_stream(self.comm_stream)
        eng.shadows_dirty = False
        return prog['dec'].loss, lr
