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
side stream as soon as its segment has been enqueued, so xGMI traffic hides under the remaining
encoder backward.  xGMI is point-to-point (7 links per GPU), ring all-reduce time is set by one
link, hence few large buckets (default 32 MiB) rather than per-tensor calls.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK from the launcher."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None, 0, 1, 0
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'     # 'nccl' is RCCL on ROCm
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist.group.WORLD, rank, world, local


class GradBuckets:
    """Contiguous [begin, end) element ranges of the flat gradient buffer, cut at `cut_points`
    (element offsets where a backward segment ends) so that each bucket is >= bucket_bytes."""

    def __init__(self, total, cut_points, bucket_bytes=32 << 20, elem_bytes=4):
        cuts = sorted(set(c for c in cut_points if 0 < c < total)) + [total]
        self.ranges = []
        begin = 0
        for c in cuts:
            if (c - begin) * elem_bytes >= bucket_bytes or c == total:
                if c > begin:
                    self.ranges.append((begin, c))
                begin = c

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
    """Train step with gradient all-reduce overlapped with backward (one rank)."""

    def __init__(self, engine, bucket_bytes=32 << 20):
        self.eng = engine
        self.bucket_bytes = bucket_bytes
        self.comm_stream = torch.cuda.Stream(device=engine.device) if engine.world > 1 else None
        self._progs = {}

    def _prepare(self, B):
        eng = self.eng
        prog = eng._train.get(B)
        if prog is None:
            prog = eng._train[B] = eng._compile_train(B)
        st = eng.store
        # segment boundaries: after the decoder, then after each encoder op (its params are final)
        cut_calls, cut_offsets = [prog['n_dec']], [st.decoder_size]
        for call_idx, opname in prog['marks']:
            e = st.entries[opname + '_bn_scale']
            cut_calls.append(call_idx)
            cut_offsets.append(e.offset + (int(e.kshape[0]) + 7) // 8 * 8)
        buckets = GradBuckets(st.trainable_size, cut_offsets, self.bucket_bytes)
        # map each bucket end to the call index at which it is complete
        segs, start = [], 0
        for (b, e) in buckets:
            idx = max(ci for ci, off in zip(cut_calls, cut_offsets) if off <= e) if e < st.trainable_size else len(prog['bwd'])
            if e == st.trainable_size:
                idx = len(prog['bwd'])
            sub = _lib_plan_slice(prog['bwd'], start, idx)
            segs.append((sub, (b, e)))
            start = idx
        return dict(prog=prog, segs=segs, graphs=[None] * (len(segs) + 1))

    def train_step(self, image, caption):
        eng = self.eng
        if eng.world <= 1:
            return eng.train_step(image, caption)
        B = int(image.shape[0])
        P = self._progs.get(B)
        if P is None:
            P = self._progs[B] = self._prepare(B)
        prog = P['prog']
        if eng.shadows_dirty:
            eng.refresh_shadows()
        eng._feed_train(prog, image, caption)
        cur = torch.cuda.current_stream(eng.device)
        holder = {}
        eng._run_captured(holder_graph(P, 0), 'g', [prog['fwd']])
        st = eng.store
        for i, (sub, (b, e)) in enumerate(P['segs']):
            eng._run_captured(holder_graph(P, i + 1), 'g', [sub])
            ev = torch.cuda.Event()
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(st.grad[b:e], op=dist.ReduceOp.SUM, group=eng.pg)
        cur.wait_stream(self.comm_stream)
        lr = eng.optimizer_step()
        eng.refresh_shadows()
        return prog['dec'].loss, lr


def holder_graph(P, i):
    """Per-segment dict that _run_captured stores its hipGraph in."""
    if P['graphs'][i] is None:
        P['graphs'][i] = {}
    return P['graphs'][i]


def _lib_plan_slice(plan, start, stop):
    from ._lib import Plan
    sub = Plan()
    sub.calls = plan.calls[start:stop]
    sub._keep = plan._keep[start:stop]
    return sub
