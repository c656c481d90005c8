"""Rehearsal of the N > 1 path on the one-GPU box: two ranks share cuda:0 and exchange gradients
over gloo (RCCL refuses two ranks on one device).  Exercises dp.OverlappedTrainer end to end --
segmented backward graphs, side-stream all-reduce of contiguous buckets, 1/N scale in Adam -- and
checks the result against a single-process emulation of ParallelExecutor's semantics (quirk Q9)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(encoder='mobilenetv2', image_size=64, hidden=32, embed=16, vocab=50, sentence_length=6, infer_max_length=6,
          attention='slots', dtype='f32', learning_rate=1e-3, batch_size=4)


def _batch():
    sys.path.insert(0, ROOT)
    from tests.conftest import make_caption
    rng = np.random.RandomState(3)
    return rng.uniform(0, 1, (4, 3, 64, 64)).astype(np.float32), make_caption(rng, 4, 6, 50)


def _worker(rank, world, port, q, payload='f32'):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      CAPMI_DETERMINISTIC='1')
    import torch.distributed as dist
    from myimagecaptioningmodel_amd import default_cfg, dp
    from myimagecaptioningmodel_amd.model import CaptionEngine
    pg, r, w, _ = dp.init_process_group_from_env(backend='gloo')
    eng = CaptionEngine(default_cfg(**KW), device='cuda:0', use_graph=True, process_group=pg)
    trainer = dp.OverlappedTrainer(eng, bucket_bytes=64 << 10, bucket_dtype=payload)          # small buckets: several segments
    assert trainer.bucket_dtype == payload
    image, cap = _batch()
    lo, hi = rank * 2, rank * 2 + 2
    losses = []
    for _ in range(3):       # step 1 captures the segment graphs, steps 2-3 replay them
        loss, lr = trainer.train_step(image[lo:hi], cap[lo:hi])
        losses.append(float(loss.cpu()[0]))
    nseg = len(trainer._progs[2]['segs'])
    p = eng.export_reference_params()
    q.put((rank, losses, nseg, {k: p[k] for k in ('lstm_w', 'fc_0.w_0', 'conv9_weights', 'conv1_1_weights', 'conv9_bn_mean')}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('payload', ['f32', 'bf16'])
def test_two_ranks_overlapped_allreduce_equals_emulation(deterministic, payload):
    """payload: what travels through the all-reduce -- the f32 gradients (the reference's precision) or their bf16 cast
    (capmi_cast -> bf16 sum -> capmi_adam_g16).  The emulation applies the same cast, so both are bit-exact."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, payload)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    assert res[0][2] >= 3                                               # really segmented
    # both ranks hold identical parameters after the synchronised steps (BN running stats stay per rank)
    for k in ('lstm_w', 'fc_0.w_0', 'conv9_weights', 'conv1_1_weights'):
        np.testing.assert_array_equal(res[0][3][k], res[1][3][k], err_msg=k)
    assert np.abs(res[0][3]['conv9_bn_mean'] - res[1][3]['conv9_bn_mean']).max() > 0
    # single-process emulation: per-shard forward/backward (own mask count), mean gradient, Adam
    from myimagecaptioningmodel_amd import default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    image, cap = _batch()
    engs = [CaptionEngine(default_cfg(**KW), device='cuda:0', use_graph=False) for _ in range(2)]
    losses0 = []
    for _ in range(3):
        ls = [float(e.forward_backward(image[2 * i:2 * i + 2], cap[2 * i:2 * i + 2]).cpu()[0]) for i, e in enumerate(engs)]
        losses0.append(ls[0])
        if payload == 'bf16':
            g = (engs[0].store.grad.bfloat16() + engs[1].store.grad.bfloat16()).float() * 0.5
        else:
            g = (engs[0].store.grad + engs[1].store.grad) / 2
        for e in engs:
            e.store.grad.copy_(g)
            e.optimizer_step()
            e.refresh_shadows()
    # rank 0 reports its local loss (train.py:142).  Deterministic mode on both sides (no f32 atomics), a two-rank sum is
    # commutative, and (g0 + g1) / 2 == (g0 + g1) * 0.5 exactly: the overlapped, bucketed, segmented path must land on the
    # emulation's bits -- a wrong bucket boundary or a missed event wait cannot hide inside a tolerance
    assert res[0][1] == losses0, (res[0][1], losses0)
    p = engs[0].export_reference_params()
    for k in ('lstm_w', 'fc_0.w_0', 'conv9_weights', 'conv1_1_weights'):
        np.testing.assert_array_equal(p[k], res[0][3][k], err_msg=k)


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', CAPMI_FORCE_DP='1',
                      CAPMI_DETERMINISTIC='1')
    import torch.distributed as dist
    from myimagecaptioningmodel_amd import default_cfg, dp
    from myimagecaptioningmodel_amd.model import CaptionEngine
    pg, r, w, _ = dp.init_process_group_from_env()                      # backend nccl = RCCL
    assert dist.get_backend(pg) == 'nccl' and w == 1
    eng = CaptionEngine(default_cfg(**KW), device='cuda:0', use_graph=True, process_group=pg)
    trainer = dp.OverlappedTrainer(eng, bucket_bytes=64 << 10)
    assert trainer.active
    # the collective must be the C ABI's (capmi_comm_init + capmi_allreduce_bucket rows in the three-lane table): a failed
    # communicator set-up fails the test instead of passing through torch.distributed
    assert trainer.native_comm is not None and trainer.native_comm.ok, getattr(trainer.native_comm, 'why', 'no native communicator')
    image, cap = _batch()
    losses = [float(trainer.train_step(image, cap)[0].cpu()[0]) for _ in range(3)]
    trainer.check_sync()
    assert 'step' in trainer._progs[4]                                  # the fused three-lane plan ran, not the per-segment replay
    p = eng.export_reference_params()
    q.put((losses, len(trainer._progs[4]['segs']), p))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_bucketed_path(deterministic):
    """The N > 1 code path with real RCCL calls on the bucket stream (one-rank group: the sum is the identity), against
    the single-process step: the same losses and EVERY parameter bit for bit (deterministic mode on both sides)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    proc = ctx.Process(target=_rccl_worker, args=(port, q))
    proc.start()
    losses, nseg, params = q.get(timeout=900)
    proc.join(timeout=900)
    assert proc.exitcode == 0 and nseg >= 3
    from myimagecaptioningmodel_amd import default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    eng = CaptionEngine(default_cfg(**KW), device='cuda:0', use_graph=True)
    image, cap = _batch()
    want = [float(eng.train_step(image, cap)[0].cpu()[0]) for _ in range(3)]
    assert losses == want, (losses, want)
    p = eng.export_reference_params()
    for k in p:
        np.testing.assert_array_equal(p[k], params[k], err_msg=k)


# ------------------------------------------------------------------ BASELINE configs[2]: ResNet-50 bf16, 32 images per rank
def _cfg2_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', CAPMI_FORCE_DP='1',
                      CAPMI_DETERMINISTIC='1')
    import torch.distributed as dist
    import bench
    from myimagecaptioningmodel_amd import default_cfg, dp
    from myimagecaptioningmodel_amd.model import CaptionEngine
    pg, r, w, _ = dp.init_process_group_from_env()                      # backend nccl = RCCL
    assert dist.get_backend(pg) == 'nccl'
    cfg = default_cfg(batch_size=32, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=True, process_group=pg)
    assert dp.OverlappedTrainer(eng).bucket_dtype == 'f32'             # what bench.py --gpus N runs: the reference's f32 exchange (bf16 buckets are opt-in)
    trainer = dp.OverlappedTrainer(eng, bucket_dtype='f32')             # default 32 MiB buckets; f32 payload: comparable bit for bit
    assert trainer.active
    assert trainer.native_comm is not None and trainer.native_comm.ok, getattr(trainer.native_comm, 'why', 'no native communicator')
    image, cap = bench.synthetic_batch(32, cfg, 1234)
    image_d, cap_d = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
    losses = [float(trainer.train_step(image_d, cap_d)[0].cpu()[0]) for _ in range(3)]
    trainer.check_sync()
    p = eng.export_reference_params()
    segs = trainer._progs[32]['segs']
    # the same three steps with the bf16 payload (capmi_cast -> capmi_allreduce_bucket_bf16 -> capmi_adam_g16) on a fresh engine
    eng16 = CaptionEngine(cfg, device='cuda:0', use_graph=True, process_group=pg)
    tr16 = dp.OverlappedTrainer(eng16, bucket_dtype='bf16')
    assert tr16.native_comm is not None and tr16.native_comm.ok
    losses16 = [float(tr16.train_step(image_d, cap_d)[0].cpu()[0]) for _ in range(3)]
    tr16.check_sync()
    names16 = [r[1] for r in tr16._progs[32]['step'].calls if r[0] is not None]
    p16 = eng16.export_reference_params()
    q.put((losses, [(b, e) for _, (b, e), _ in segs], eng.store.trainable_size, 'step' in trainer._progs[32], p,
           losses16, names16.count('capmi_allreduce_bucket_bf16'), names16.count('capmi_adam_g16'),
           {k: p16[k] for k in ('lstm_w', 'fc_11.w_0', 'word_embedding')}))
    dist.barrier()
    dist.destroy_process_group()


def test_configs2_resnet50_bf16_through_the_bucketed_rccl_path(deterministic):
    """BASELINE configs[2]'s per-rank program (ResNet-50 + 512-d decoder, bf16, 32 images per rank) driven by
    dp.OverlappedTrainer on a one-rank RCCL group -- the three-lane launch table, each bucket's capmi_allreduce_bucket + Adam
    + shadow refresh on the communication lane -- against the fused single-rank step of the same engine code: the buckets
    tile the trainable range, and (deterministic mode on both sides, one-rank sum = identity) every loss and EVERY
    parameter, encoder included, after three steps is bit-identical."""
    import bench
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    proc = ctx.Process(target=_cfg2_worker, args=(port, q))
    proc.start()
    losses, ranges, total, native, params, losses16, n_ar16, n_adam16, params16 = q.get(timeout=1200)
    proc.join(timeout=600)
    assert proc.exitcode == 0
    assert len(ranges) >= 4 and ranges[0][0] == 0 and ranges[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    from myimagecaptioningmodel_amd import default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    cfg = default_cfg(batch_size=32, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=True)
    p0 = eng.export_reference_params()
    image, cap = bench.synthetic_batch(32, cfg, 1234)
    image_d, cap_d = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
    want = [float(eng.train_step(image_d, cap_d)[0].cpu()[0]) for _ in range(3)]
    print('configs[2] losses: bucketed RCCL path', losses, 'fused single-rank', want, 'three-lane table', native)
    assert native, 'the data-parallel step did not run as the three-lane launch table with capmi_allreduce_bucket rows'
    assert losses == want, (losses, want)
    p = eng.export_reference_params()
    assert any(np.abs(p[k] - p0[k]).max() > 0 for k in ('lstm_w', 'res_conv1_weights', 'res5_3_branch2c_weights'))
    for k in params:
        np.testing.assert_array_equal(p[k], params[k], err_msg=k)
    # bf16 payload: every bucket went through the bf16 all-reduce and the bf16-gradient Adam; the first loss is the same
    # forward pass, later steps stay within the bf16 bound and the decoder's parameters move with the f32-payload run
    assert n_ar16 == len(ranges) and n_adam16 == len(ranges)
    assert losses16[0] == want[0]
    np.testing.assert_allclose(losses16, want, rtol=0, atol=5e-2)
    for k in params16:
        moved = np.linalg.norm(p[k] - p0[k])
        assert np.linalg.norm(params16[k] - p[k]) <= 0.1 * moved, (k, np.linalg.norm(params16[k] - p[k]) / moved)


# ------------------------------------------------------------------ train.py:121-139,172: the loop drives the data-parallel step
def _loop_worker(rank, world, port, q, root):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      CAPMI_DETERMINISTIC='1')
    import torch.distributed as dist
    from myimagecaptioningmodel_amd import ckpt, default_cfg, dp, train_loop
    from myimagecaptioningmodel_amd.model import CaptionEngine
    from tests.conftest import make_caption
    pg, r, w, _ = dp.init_process_group_from_env(backend='gloo')
    eng = CaptionEngine(default_cfg(**dict(KW, sample_count=8, batch_size=4)), device='cuda:0', use_graph=False, process_group=pg)
    trainer = dp.OverlappedTrainer(eng, bucket_bytes=64 << 10)
    rng = np.random.RandomState(11)
    data = {ep: [(rng.uniform(0, 1, (4, 3, 64, 64)).astype(np.float32), make_caption(rng, 4, 6, 50)) for _ in range(2)] for ep in (1, 2)}

    def batches(epoch):
        for image, cap in data[epoch]:
            yield dict(image=image[2 * rank:2 * rank + 2], caption=cap[2 * rank:2 * rank + 2])      # this rank's shard (quirk Q8)
    cp, lp = os.path.join(root, 'ckpt'), os.path.join(root, 'log')
    scores = {1: 0.25, 2: 0.125}
    conf = train_loop.train(eng, batches, 2, cp, lp, log_every_n_step=1, trainer=trainer, eval_score=lambda ep: scores[ep])
    p = eng.export_reference_params()
    # a fresh engine on every rank resumes from what rank 0 wrote (epoch 2 is re-run: the JSON holds the epoch last STARTED)
    eng2 = CaptionEngine(default_cfg(**dict(KW, sample_count=8, batch_size=4, seed=5)), device='cuda:0', use_graph=False, process_group=pg)
    ckpt.load_persistables(eng2, os.path.join(cp, 'checkpoint'))
    p2 = eng2.export_reference_params()
    q.put((rank, conf, {k: p[k] for k in ('lstm_w', 'conv1_1_weights')}, all(np.array_equal(p[k], p2[k]) for k in p if k in eng.store.entries),
           eng.step_count, eng2.step_count))
    dist.barrier()
    dist.destroy_process_group()


def test_train_loop_drives_the_overlapped_trainer_and_rank_zero_saves(tmp_path):
    """train.py:121-139: the executor that runs the loop IS the data-parallel one; :172 one process saves.  Two ranks
    (gloo, sharing the card) run train_loop.train(..., trainer=OverlappedTrainer): both end on the same parameters, only
    rank 0 wrote the checkpoint / resume JSON / log / best-score copy (train.py:85-91), and every rank can load them."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    roots = [str(tmp_path / ('rank%d' % r)) for r in range(2)]
    shared = str(tmp_path / 'shared')
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, shared)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    for k in ('lstm_w', 'conv1_1_weights'):
        np.testing.assert_array_equal(res[0][2][k], res[1][2][k], err_msg=k)
    assert res[0][3] and res[1][3]                                      # the checkpoint rank 0 wrote restores rank r's masters bit for bit
    assert res[0][4] == res[1][4] == 4 and res[0][5] == res[1][5] == 4  # four Adam steps; the step counter travels with the checkpoint
    import json
    conf = json.loads(open(os.path.join(shared, 'log', 'config')).read())
    assert conf['epoch'] == 2 and conf['best_bleu'] == 0.25 and res[0][1]['best_bleu'] == 0.25
    assert os.path.isfile(os.path.join(shared, 'ckpt', 'checkpoint', 'lstm_w'))
    assert os.path.isfile(os.path.join(shared, 'ckpt', 'checkpoint_best_bleu', 'lstm_w'))
    log = open(os.path.join(shared, 'log', 'log.txt')).read()
    assert log.count('Epoch 1') == 1 and log.count('Step 1 ') == 2      # one writer: every line once
