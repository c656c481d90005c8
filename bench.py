#!/usr/bin/env python
"""bench.py -- train images/sec of the captioning hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: ResNet-50 encoder + 512-d attention-LSTM decoder,
vocab 10 000, 224x224, seq_len 20, batch 64 per GPU, bf16 storage / f32 accumulate.  One step =
feed (device-resident synthetic batch) -> forward -> backward -> gradient all-reduce (N > 1) ->
Paddle-form Adam -> weight-shadow refresh.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOAD = dict(encoder='resnet50', image_size=224, hidden=512, embed=512, vocab=10000, sentence_length=20,
                infer_max_length=20, attention='slots', dtype='bf16', learning_rate=5e-5, encoder_trainable=True)
PER_GPU_BATCH = 64
FLOPS_PER_IMAGE = {1: 26.18e9, 3: 149.9e9}      # 6 x forward GEMM-class MACs (cfg 2/3: 4.363 GMAC; cfg 4: 24.98 GMAC)
# BASELINE configs[3] (a parity / stress case, not the bench line): ResNet-101 + 2-layer 1024-d LSTM decoder, 384x384,
# seq_len 30, vocab 20k, 64 images per GPU (512 over 8).  E = H = 1024 and Ti = L are this build's reading of the
# unspecified sizes (SURVEY.md section 8, table of configs)
WORKLOAD_CFG3 = dict(encoder='resnet101', image_size=384, hidden=1024, embed=1024, vocab=20000, sentence_length=30,
                     infer_max_length=30, attention='slots', dtype='bf16', learning_rate=5e-5, encoder_trainable=True, rnn_layer=2)


def synthetic_batch(B, cfg, seed):
    """uniform [0,1) images; captions <start>=2, U{L/2..L-2} content tokens in
    [4,V), <stop>=3, <pad>=0 (mirrors ai_challenge_tokenizer.py:81-86)."""
    rng = np.random.RandomState(seed)
    S, L, V = cfg['image_size'], cfg['sentence_length'], cfg['vocab']
    image = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    cap = np.zeros((B, L), np.int64)
    for b in range(B):
        n = rng.randint(L // 2, L - 1)
        cap[b, 0] = 2
        cap[b, 1:1 + n] = rng.randint(4, V, size=n)
        cap[b, 1 + n] = 3
    return image, cap


def pmc_traffic(symbol):
    """HBM bytes per launch of the kernel `symbol` (the exact rocprofv3 row) from the newest committed PMC pass
    (profiles/rNN_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be read inside this process, so the value comes from a
    SEPARATE profile pass of the same command; returns (MB per launch or None, source) -- source names the file and the tree it
    was recorded on, and the value is None when the symbol has no row there or the kernels changed since."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json'))
    rnd = lambda f: int(re.match(r'r(\d+)_', os.path.basename(f)).group(1))
    files = sorted((f for f in files if re.match(r'r\d+_', os.path.basename(f))), key=rnd)
    if not files:
        return None, None
    with open(files[-1]) as fh:
        doc = json.load(fh)
    source = dict(file=os.path.relpath(files[-1], ROOT), recorded_on=doc.get('head'), pass_='separate rocprofv3 --pmc runs of bench.py')
    row = doc['kernels'].get(symbol)
    if row is None:
        source['missing'] = 'no row for this symbol'
        return None, source
    # stale profile: the kernels changed since the pass was recorded (checkable only where the git history is present --
    # on the GPU box the snapshot has none, there the recorded head is all that can be reported)
    if doc.get('head') and os.path.isdir(os.path.join(ROOT, '.git')):
        import subprocess
        rc = subprocess.call(['git', '-C', ROOT, 'diff', '--quiet', doc['head'], 'HEAD', '--', 'myimagecaptioningmodel_amd/csrc'],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if rc != 0:
            source['stale'] = 'csrc/ differs from the tree the pass was recorded on'
            return None, source
    return round(row['hbm_bytes_per_launch'] / 1e6, 3), source


def rocprof_row(symbol):
    """The row of `symbol` in the newest committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/rNN_bench_kernel_stats.csv): (average launch duration in us, calls, source) or (None, None, source)."""
    import csv
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')) if re.match(r'r\d+_', os.path.basename(f))]
    if not files:
        return None, None, None
    f = max(files, key=lambda x: int(re.match(r'r(\d+)_', os.path.basename(x)).group(1)))
    src = dict(file=os.path.relpath(f, ROOT))
    pmc = f.replace('_bench_kernel_stats.csv', '_pmc_traffic.json')
    if os.path.exists(pmc):
        try:
            src['recorded_on'] = json.load(open(pmc)).get('head')
        except Exception:
            pass
    for r in csv.DictReader(open(f)):
        if r['Name'] == symbol:
            return float(r['AverageNs']) / 1e3, int(r['Calls']), src
    src['missing'] = 'no row for this symbol'
    return None, None, src


def roofline_of(symbol, s, steps):
    """Both roofline fractions of one kernel symbol from its in-model time: against HBM (algorithmic bytes) and -- for a GEMM --
    against the dense bf16 MFMA peak (algorithmic flops).  `bound` names the roof the kernel's intensity puts it under (the
    bf16 ridge is 2.5 PF / 8 TB/s = 312 flop/B) and `frac` is the fraction against THAT roof; `frac_hbm` / `frac_mfma` are
    always both there."""
    from myimagecaptioningmodel_amd import profiling
    sec = s['ms'] * 1e-3
    gbps = s['bytes'] / sec / 1e9
    tflops = s['flops'] / sec / 1e12
    r = dict(kernel=symbol, frac_hbm=round(gbps / profiling.PEAK_HBM_GBPS, 4),
             frac_mfma=round(tflops / profiling.PEAK_MFMA_TFLOPS['bf16'], 4) if s['flops'] > 0 else None,
             achieved_gbps=round(gbps, 1), achieved_tflops=round(tflops, 2) if s['flops'] > 0 else None,
             avg_launch_us=round(s['ms'] / s['launches'] * 1e3, 2), launches_per_step=round(s['launches'] / steps, 1),
             ms_per_step=round(s['ms'] / steps, 3), lanes=sorted(s['lanes']),
             algorithmic_per_launch={'GFLOP': round(s['flops'] / s['launches'] / 1e9, 3), 'MB': round(s['bytes'] / s['launches'] / 1e6, 3)})
    if s['flops'] > 0 and s['flops'] / max(s['bytes'], 1) > 312:
        r.update(bound='mfma', achieved=r['achieved_tflops'], peak=profiling.PEAK_MFMA_TFLOPS['bf16'], unit='TFLOP/s', frac=r['frac_mfma'])
    else:
        r.update(bound='hbm', achieved=r['achieved_gbps'], peak=profiling.PEAK_HBM_GBPS, unit='GB/s', frac=r['frac_hbm'])
    return r


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box gives one GPU's
    job a share of the host -- spinning more threads than the quota allows stalls the whole process every scheduler
    period, DESIGN.md lesson 18).  When no quota can be read the share is taken to be 16 cores (what the pool grants a
    one-GPU job); CAPMI_CPU_CORES overrides."""
    if os.environ.get('CAPMI_CPU_CORES'):
        return max(1, int(os.environ['CAPMI_CPU_CORES']))
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    rel = ''
    try:
        for line in open('/proc/self/cgroup'):
            parts = line.strip().split(':', 2)
            if len(parts) == 3 and (parts[1] == '' or 'cpu' in parts[1].split(',')):
                rel = parts[2]
    except Exception:
        pass
    cands = [('/sys/fs/cgroup' + rel + '/cpu.max', 2), ('/sys/fs/cgroup/cpu.max', 2),
             ('/sys/fs/cgroup/cpu' + rel + '/cpu.cfs_quota_us', 1), ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', 1)]
    for path, ver in cands:
        try:
            txt = open(path).read().split()
            if ver == 2:
                q = None if txt[0] == 'max' else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                q = None if q <= 0 else q / float(open(os.path.join(os.path.dirname(path), 'cpu.cfs_period_us')).read())
            if q:
                quota = q
            break
        except Exception:
            continue
    if quota is None:
        return max(1, min(n, 16))
    return max(1, min(n, int(quota)))


def progress(msg):
    """One line on stderr per phase: a long run stays visibly alive (stdout carries only the JSON line)."""
    sys.stderr.write('bench: %s\n' % msg)
    sys.stderr.flush()


def _cpu_train_steps(ocfg, B, warm, steps, seed, deadline):
    """fwd + bwd (torch.autograd) + Paddle-form Adam of the reference graph on the host cores: tests/torch_ref.py, the
    torch restatement that pins the NumPy oracle (same graph, same initialisers).  Returns (seconds per step, steps)."""
    import torch
    from oracle import model as om
    from tests import torch_ref
    params = om.init_params(ocfg, seed=0, dtype=np.float32)
    p = {k: torch.tensor(v, requires_grad=om.is_trainable(k, ocfg)) for k, v in params.items()}
    train = [v for v in p.values() if v.requires_grad]
    m = [torch.zeros_like(v) for v in train]
    v2 = [torch.zeros_like(v) for v in train]
    image, cap = synthetic_batch(B, ocfg, seed)
    image, cap = torch.tensor(image), torch.tensor(cap)
    times = []
    for it in range(warm + steps):
        t0 = time.perf_counter()
        loss, _ = torch_ref.forward_loss(ocfg, p, image, cap)
        grads = torch.autograd.grad(loss, train, allow_unused=True)
        t = it + 1
        lr_t = 5e-5 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)            # IC/train.py:26-31: Paddle-1.8 adam op
        with torch.no_grad():
            for w, g, mm, vv in zip(train, grads, m, v2):
                if g is None:
                    continue
                mm.mul_(0.9).add_(g, alpha=0.1)
                vv.mul_(0.999).addcmul_(g, g, value=0.001)
                w.sub_(lr_t * mm / (vv.sqrt() + 1e-8))
        dt = time.perf_counter() - t0
        if it >= warm:
            times.append(dt)
        if time.perf_counter() > deadline and len(times) >= 1:
            break
    return (sum(times) / len(times), len(times)) if times else (dt, 0)


def cpu_baseline(cfg, budget_s=25.0, probe_batch=8, max_batch=64, label='bench workload (ResNet-50, 224x224, V 10000, L 20)'):
    """'build's CPU restatement of the reference graph, not PaddlePaddle' (Paddle 1.8 is not
    installable here) = the torch-CPU build of the identical graph (tests/torch_ref.py), whole train steps (fwd + bwd +
    Paddle-form Adam) in fp32 on the box's host cores, 2 warm-up + >= 5 timed steps: BASELINE cfg 1 (repo-default
    MobileNetV2 model, 64x64, V 1000, L 10, B 4) and the bench workload (ResNet-50, 224x224, V 10000, L 20) at the
    largest batch <= 64 whose 7 steps fit the budget.  `value` is the bench workload's rate."""
    import torch
    from oracle import model as om
    cores = host_cores()
    torch.set_num_threads(cores)
    t_start = time.perf_counter()
    progress('cpu baseline on %d core(s): cfg 1' % cores)
    ocfg1 = om.default_cfg(encoder='mobilenetv2', image_size=64, hidden=1024, embed=256, vocab=1000, sentence_length=10,
                           infer_max_length=10, attention='singleton')
    s1, n1 = _cpu_train_steps(ocfg1, 4, 2, 5, 1234, t_start + 0.3 * budget_s)
    ocfg2 = om.default_cfg(**{k: cfg[k] for k in ('encoder', 'image_size', 'hidden', 'embed', 'vocab', 'sentence_length',
                                                  'infer_max_length', 'attention', 'rnn_layer') if k in cfg})
    # probe at a small batch, then the largest power-of-two batch <= max_batch whose 2 + 5 steps fit what is left of the budget
    progress('cpu baseline: cfg 1 %.3f s/step; probing the %s at batch %d' % (s1, label.split(' (')[0], probe_batch))
    probe, _ = _cpu_train_steps(ocfg2, probe_batch, 1, 1, 1234, time.perf_counter() + 60.0)
    left = budget_s - (time.perf_counter() - t_start)
    B = max_batch
    while B > probe_batch and 7 * probe * (B / float(probe_batch)) * 0.8 > left:        # (x0.8: larger batches run the cores more efficiently)
        B //= 2
    progress('cpu baseline: %.2f s/step at batch %d; timing batch %d' % (probe, probe_batch, B))
    s2, n2 = _cpu_train_steps(ocfg2, B, 2 if B > probe_batch else 1, 5, 1234, time.perf_counter() + max(left, 5.0) * 1.5)
    return dict(value=round(B / s2, 3), unit='images/sec', cores=int(cores), kind='port',
                sample='torch-CPU fp32 restatement of the reference graph (tests/torch_ref.py: fwd + autograd bwd + Paddle-form Adam), '
                       'NOT PaddlePaddle; %s at batch %d: %d timed steps after warm-up, '
                       '%.2f s/step; %.1f s of CPU work in all' % (label, B, n2, s2, time.perf_counter() - t_start),
                cfg1=dict(value=round(4 / s1, 2), unit='images/sec', sample='BASELINE cfg 1 (repo-default MobileNetV2 + 1024/256 decoder, 64x64, '
                          'V 1000, L 10, batch 4, singleton attention): %d timed steps after 2 warm-up, %.3f s/step' % (n1, s1)))


def extras(eng, cfg, B, image_d, cap_d, dev, steps=10):
    """Extra lines next to the headline (same workload, same batch, same weights): the reference-faithful attention
    (quirk Q1, softmax over a size-1 axis: 'singleton'), the reference precision (f32 end to end), and the loss gap of the
    bf16 engine against the f32 engine on the first step."""
    import torch
    from myimagecaptioningmodel_amd.model import CaptionEngine
    res = {}

    def rate(e, n):
        for _ in range(3):
            e.train_step(image_d, cap_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            e.train_step(image_d, cap_d)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e.check_sync()
        return round(B * n / dt, 1)
    first = {}
    for key, over in (('bf16_slots', {}), ('bf16_singleton_attention', dict(attention='singleton')), ('f32_slots', dict(dtype='f32'))):
        progress('extra: ' + key)
        e = CaptionEngine(dict(cfg, **over), device=dev, use_graph=False)      # every engine starts from the same seeded initialisation
        first[key] = float(e.forward_loss(image_d, cap_d).cpu()[0])
        if key != 'bf16_slots':
            res[key + '_images_per_sec'] = rate(e, steps if key != 'f32_slots' else max(3, steps // 2))
        del e
        torch.cuda.empty_cache()
    # the reference-precision engine against ITS roof: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), 157.3 TFLOP/s dense
    from myimagecaptioningmodel_amd import profiling
    res['f32_slots_model_mfma_frac'] = round(res['f32_slots_images_per_sec'] * FLOPS_PER_IMAGE[1] / (profiling.PEAK_MFMA_TFLOPS['f32'] * 1e12), 4)
    res['first_step_loss'] = {k: round(v, 5) for k, v in first.items()}
    res['loss_gap_bf16_vs_f32_engine'] = round(abs(first['bf16_slots'] - first['f32_slots']), 5)
    res['note'] = ('singleton = the reference graph as written (alpha == 1); f32 = reference precision (exact-f32 MFMA kernels); losses and the '
                   'gap are taken at the seeded initialisation, engine against engine at full size -- against the CPU oracle the tests '
                   'hold bf16 to 5e-2 and f32 to 1e-3')
    return res


def decode_measure(B, beam, steps, warmup, use_graph=True, pipelined=0):
    """BASELINE configs[4]: the infer.py path (/root/reference/ImageCaptioning/infer.py:26-36 runs the saved GREEDY graph on
    one image; beam search is this build's extension) at batch B, 224x224, ResNet-50 + 512-d decoder, bf16, `is_test`
    batch norm as in the exported inference model.  One step = one batch decoded (encoder + Ti = 20 decoder steps +
    backtrack), replayed from a hipGraph.  Returns captions/sec and the p50 / p90 latency of a batch."""
    import torch
    from myimagecaptioningmodel_amd import default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    cfg = default_cfg(batch_size=B, sample_count=0, **WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=use_graph)
    image, _ = synthetic_batch(B, cfg, 1234)
    image_d = torch.as_tensor(image).to('cuda:0')
    for _ in range(max(2, warmup)):          # call 1 warms up + captures, later calls replay
        ids = eng.decode(image_d, beam=beam, is_test=True)
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        ids = eng.decode(image_d, beam=beam, is_test=True)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    ids = ids.cpu().numpy()
    assert ids.shape == (B, cfg['infer_max_length']) and ((ids >= 0) & (ids < cfg['vocab'])).all()
    lat.sort()
    res = dict(captions_per_sec=round(B * steps / dt, 1), ms_per_batch=round(dt / steps * 1e3, 3),
               p50_latency_ms=round(lat[len(lat) // 2] * 1e3, 3), p90_latency_ms=round(lat[int(len(lat) * 0.9)] * 1e3, 3),
               batch=B, beam=beam, steps=steps)
    if pipelined:
        # the serving form of the same path: `pipelined` batches in flight on as many HIP streams (CaptionEngine.decode_pipelined);
        # a batch's latency is then no longer the reciprocal of the rate, so both are reported
        n = max(64, 4 * steps)          # long enough that filling and draining the pipeline (one batch time each) is a few per cent
        feeds = [image_d] * n
        one = eng.decode(image_d, beam=beam, is_test=True).clone()
        for _ in range(2):
            outs = eng.decode_pipelined(feeds[:8], beam=beam, depth=pipelined + 1, decoders=pipelined)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = eng.decode_pipelined(feeds, beam=beam, depth=pipelined + 1, decoders=pipelined)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        assert all(bool((o == one).all()) for o in outs), 'pipelined decode changed the ids'
        res['pipelined'] = dict(decoders_in_flight=pipelined, program_copies=pipelined + 1, batches=n, captions_per_sec=round(B * n / dtp, 1), ms_per_batch=round(dtp / n * 1e3, 3),
                                ids_equal_to_one_at_a_time=True)
    del eng
    torch.cuda.empty_cache()
    return res


def decode_bench(args):
    """`--decode`: BASELINE configs[4] as its own JSON line."""
    B, beam = args.batch or 128, args.beam
    m = decode_measure(B, beam, args.steps, args.warmup, use_graph=not args.no_graph, pipelined=0 if args.no_graph else args.in_flight)
    out = {'metric': 'decode captions/sec (224x224, beam=%d, batch %d)' % (beam, B), 'value': m['captions_per_sec'], 'unit': 'captions/sec',
           'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': m['ms_per_batch'],
           'p50_latency_ms': m['p50_latency_ms'], 'p90_latency_ms': m['p90_latency_ms'], 'pipelined': m.get('pipelined'),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': 'BASELINE configs[4]: infer.py path, ResNet-50 (build-defined) + 512-d decoder, vocab 10000, 224x224, '
                                  'Ti 20, beam %d (build-defined; beam 1 = the reference greedy loop), is_test batch norm, random-init weights' % beam,
                      'batch': B, 'beam': beam, 'launch': 'hipGraph replay of a single-lane plan' if not args.no_graph else 'capmi_plan_run, eager'}}
    print(json.dumps(out), flush=True)


def config3_measure(dev, steps=4, warmup=2):
    """BASELINE configs[3] at its full per-GPU size (ResNet-101 + 2-layer 1024-d LSTM decoder, 384x384, L 30, V 20 000, 64
    images) as an `extra` of the default run: ms/step, images/s and the whole-step MFMA fraction with SURVEY.md 8(d)'s
    149.9 GFLOP per image."""
    import torch
    from myimagecaptioningmodel_amd import default_cfg, profiling
    from myimagecaptioningmodel_amd.model import CaptionEngine
    B = PER_GPU_BATCH
    cfg = default_cfg(batch_size=B, sample_count=0, **WORKLOAD_CFG3)
    eng = CaptionEngine(cfg, device=dev, use_graph=True)
    image, cap = synthetic_batch(B, cfg, 1234)
    image_d, cap_d = torch.as_tensor(image).to(dev), torch.as_tensor(cap).to(dev)
    for _ in range(warmup):
        loss, _ = eng.train_step(image_d, cap_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = eng.train_step(image_d, cap_d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_sync()
    final = float(loss.cpu()[0])
    assert final == final, 'configs[3]: loss is NaN'
    rate = B * steps / dt
    del eng
    torch.cuda.empty_cache()
    return dict(images_per_sec=round(rate, 1), ms_per_step=round(dt / steps * 1e3, 3), steps=steps, warmup=warmup, per_gpu_batch=B,
                model_mfma_frac=round(rate * FLOPS_PER_IMAGE[3] / (profiling.PEAK_MFMA_TFLOPS['bf16'] * 1e12), 4),
                final_loss=round(final, 4),
                workload='BASELINE configs[3]: ResNet-101 (build-defined) + 2-layer 1024-d LSTM decoder, 384x384, seq_len 30, vocab 20000, 64 images per GPU, bf16')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: the BASELINE config: 64 train, 128 decode)')
    ap.add_argument('--decode', action='store_true', help='BASELINE configs[4]: beam-search decode captions/sec + p50 latency (extra mode)')
    ap.add_argument('--beam', type=int, default=5)
    ap.add_argument('--in-flight', type=int, default=3, help='--decode: also time the serving form (CaptionEngine.decode_pipelined) with this many decoders in flight beside an encoder (0: skip)')
    ap.add_argument('--config', type=int, default=1, choices=[1, 3], help='1 (default): the bench workload, BASELINE configs[1]; 3: BASELINE configs[3] '
                    '(ResNet-101 + 2-layer 1024-d LSTM, 384x384, L = 30, V = 20 000) as its own JSON line -- an extra mode like --decode')
    ap.add_argument('--no-extras', action='store_true', help='skip the extra lines (singleton attention, f32, loss gap)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--protocol', action='store_true', help="SURVEY.md 8(d)'s protocol: 20 warm-up + 100 timed steps, three runs, the MEDIAN run "
                    'is the value (every run is listed under `protocol`); implies --no-extras')
    args = ap.parse_args()
    if args.protocol:
        args.steps, args.warmup, args.no_extras = 100, 20, True

    # stdout carries exactly ONE line (the JSON result): libraries that write banners to fd 1 (RCCL prints its version
    # block there when the first communicator is created) are diverted to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.decode:
        os.dup2(real_stdout, 1)
        return decode_bench(args)
    import torch
    import torch.distributed as dist
    from myimagecaptioningmodel_amd import default_cfg, dp, profiling
    from myimagecaptioningmodel_amd.model import CaptionEngine

    pg, rank, world, local = dp.init_process_group_from_env()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    dev = 'cuda:%d' % local
    torch.cuda.set_device(local)
    B = args.batch or PER_GPU_BATCH
    workload = WORKLOAD_CFG3 if args.config == 3 else WORKLOAD
    # (--config 3 is an extra mode: its own metric line, with its own roofline and CPU baseline; the `extra` block -- the
    # bench workload's variants -- belongs to the default line only)
    cfg = default_cfg(batch_size=B * world, sample_count=0, **workload)
    eng = CaptionEngine(cfg, device=dev, use_graph=not args.no_graph, process_group=pg)
    trainer = dp.OverlappedTrainer(eng) if pg is not None else None      # (CAPMI_FORCE_DP=1: the N > 1 path on one rank)
    image, cap = synthetic_batch(B, cfg, 1234 + rank)
    image_d = torch.as_tensor(image).to(dev)
    cap_d = torch.as_tensor(cap).to(dev)

    def step():
        if trainer is not None:
            return trainer.train_step(image_d, cap_d)
        return eng.train_step(image_d, cap_d)

    progress('engine built; warm-up')
    for _ in range(max(1, args.warmup)):      # >= 1: builds plans, captures the hipGraph
        loss, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    def timed_run():
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_, _ = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, loss_
    dt, loss = timed_run()
    runs = [dt]
    if args.protocol:           # two more runs of the same length; the median run is reported
        for _ in range(2):
            if world > 1:
                dist.barrier()
            d2, loss = timed_run()
            runs.append(d2)
        dt = sorted(runs)[1]
    final_loss = float(loss.cpu()[0])
    assert final_loss == final_loss, 'loss is NaN'           # the check of train.py:140-141
    eng.check_sync()            # a grid barrier of the persistent recurrence that timed out anywhere in the loop voids the number

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            'metric': ('train images/sec (224x224, seq_len=20, vocab~10k)' if args.config == 1 else
                       'train images/sec (384x384, seq_len=30, vocab 20k, ResNet-101 + 2-layer 1024-d LSTM)'),
            'value': round(B * world * args.steps / dt, 2),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]: ResNet-50 (build-defined) + 512-d adaptive-attention LSTM decoder, '
                                    'vocab 10000, 224x224, seq_len 20, E=H=512, attention=slots, random-init weights' if args.config == 1 else
                                    'BASELINE configs[3]: ResNet-101 (build-defined) + 2-layer 1024-d adaptive-attention LSTM decoder, '
                                    'vocab 20000, 384x384, seq_len 30, E=H=1024, attention=slots, random-init weights'),
                       'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       # what ran in the timed region: laned plans are enqueued eagerly through capmi_plan_run (one foreign
                       # call per plan) on HIP streams; no hipGraph replays there (hipGraph serialises parallel branches)
                       'hipgraph': False,
                       'launch': ('capmi_plan_run: forward + backward + Adam as launch tables on 2 HIP streams' if trainer is None or not getattr(trainer, 'active', False)
                                  else ('capmi_plan_run: one table per step on 3 HIP streams, gradient buckets through capmi_allreduce_bucket (RCCL)'
                                        if trainer.native_comm is not None else 'per-bucket backward segments + torch.distributed all-reduce on a side stream')),
                       'precision_note': 'bf16 storage / f32 accumulate: the bf16 engine is held to |loss - oracle| <= 5e-2 (tests); north_star\'s 1e-3 is met by the f32 engine'},
            'final_loss': round(final_loss, 4),
        }
        if args.protocol:
            out['protocol'] = dict(rule='20 warm-up + 100 timed steps, 3 runs, median run reported (SURVEY.md 8(d))',
                                   ms_per_step_runs=[round(r / args.steps * 1e3, 3) for r in runs])
        if trainer is not None and getattr(trainer, 'active', False):
            # what the all-reduce really spanned, as RCCL itself counts it, and what the step exchanged
            out['config']['data_parallel'] = trainer.describe(B)
    # ---- roofline of the dominant kernel: HIP-event timing of every launch of the step ON ITS LANE (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_roofline:
        progress('%.2f ms/step; in-model HIP-event pass (two lanes) for the roofline line' % (dt / args.steps * 1e3))
        prog = eng._train[B]
        if 'bwd_opt' not in prog:
            eng._build_fused_bwd(prog)
        R = 3
        over = profiling.event_pair_overhead_ms(eng._stream())
        stats, lane_ms = profiling.time_step([prog['fwd'], prog['bwd_opt']], eng._stream(), repeats=R, overhead_ms=over)
        top = sorted(stats.items(), key=lambda kv: -kv[1]['ms'])
        # the dominant kernel = the SINGLE kernel symbol with the most in-model time (a row of rocprofv3 --kernel-trace --stats)
        name, s = top[0]
        roof = roofline_of(name, s, R)
        traffic, traffic_src = pmc_traffic(name)
        roof.update(traffic=traffic, traffic_source=traffic_src, traffic_unit='MB per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)',
                    share_of_kernel_time=round(s['ms'] / sum(v['ms'] for v in stats.values()), 3),
                    timing='HIP events around every launch on the stream it is launched on, the step running on its two lanes (%d runs); '
                           'minus the interval an event pair adds by itself, calibrated on an idle device' % R,
                    event_pair_overhead_us=round(over * 1e3, 2))
        # cross-check against the committed rocprofv3 summary of the same command: the same algorithmic bytes / flops over ITS
        # average duration.  HIP events on the low-priority side lane measure from the moment a kernel is ELIGIBLE (its
        # predecessor on the lane has ended) to its end, rocprofv3 from the moment it is dispatched: for side-lane kernels
        # (the weight gradients) the event figure is the larger by the dispatch wait behind the main lane's kernels; for
        # main-lane kernels the two agree within a few per cent (see roofline_main_lane_gemm).
        avg_us, calls, src = rocprof_row(name)
        if avg_us:
            per = roof['algorithmic_per_launch']
            roof['rocprof_check'] = dict(avg_launch_us=round(avg_us, 2), calls=calls,
                                         frac_hbm=round(per['MB'] * 1e6 / (avg_us * 1e-6) / 1e9 / profiling.PEAK_HBM_GBPS, 4),
                                         frac_mfma=round(per['GFLOP'] * 1e9 / (avg_us * 1e-6) / 1e12 / profiling.PEAK_MFMA_TFLOPS['bf16'], 4), source=src)
        elif src:
            roof['rocprof_check'] = dict(source=src)
        # ... and the same launches ALONE (idle device before each): what the kernel needs by itself against what it gets in the
        # step -- the dominant symbol is a weight gradient on the lowest-priority lane under a bandwidth-bound backward pass
        alone_ms, alone_n = profiling.time_label_alone([prog['fwd'], prog['bwd_opt']], name, eng._stream(), overhead_ms=over)
        if alone_n:
            per = roof['algorithmic_per_launch']
            roof['alone'] = dict(avg_launch_us=round(alone_ms * 1e3, 2), launches=alone_n,
                                 frac_hbm=round(per['MB'] * 1e6 / (alone_ms * 1e-3) / 1e9 / profiling.PEAK_HBM_GBPS, 4),
                                 frac_mfma=round(per['GFLOP'] * 1e9 / (alone_ms * 1e-3) / 1e12 / profiling.PEAK_MFMA_TFLOPS['bf16'], 4),
                                 note='the same launches one at a time on an idle device (operands as the step left them)')
        out['roofline'] = roof
        gemm = [(k, v) for k, v in top if v['flops'] > 0]
        # the largest MAIN-lane GEMM symbol next to it (the dominant symbol is a side-lane weight gradient)
        main_gemm = next(((k, v) for k, v in gemm if 0 in v['lanes'] and k != name), None)
        if main_gemm is not None:
            out['roofline_main_lane_gemm'] = roofline_of(main_gemm[0], main_gemm[1], R)
            avg_us, calls, src = rocprof_row(main_gemm[0])
            if avg_us:
                out['roofline_main_lane_gemm']['rocprof_check'] = dict(avg_launch_us=round(avg_us, 2), calls=calls, source=src)
        g_fl, g_ms = sum(v['flops'] for _, v in gemm), sum(v['ms'] for _, v in gemm)
        out['gemm_mfma_frac'] = round(g_fl / (g_ms * 1e-3) / 1e12 / profiling.PEAK_MFMA_TFLOPS['bf16'], 4)
        out['gemm_summary'] = dict(gflop_per_step=round(g_fl / R / 1e9, 1), gemm_kernel_ms_per_step=round(g_ms / R, 3),
                                   note='sum of the algorithmic flops of every MFMA kernel of the step / the sum of their in-model durations / 2.5 PFLOP/s')
        out['kernel_breakdown_ms_per_step'] = {k: round(v['ms'] / R, 3) for k, v in top[:10]}
        out['lane_busy_ms_per_step'] = {str(k): round(v / R, 3) for k, v in sorted(lane_ms.items())}
        flops_img = FLOPS_PER_IMAGE[args.config]          # 6 x forward GEMM-class MACs
        out['model_mfma_frac'] = round(out['value'] * flops_img / (world * profiling.PEAK_MFMA_TFLOPS['bf16'] * 1e12), 4)
    if trainer is not None and getattr(trainer, 'active', False) and trainer.native_comm is not None:
        exposed = trainer.exposed_allreduce_ms(image_d, cap_d)         # every rank runs the extra steps (collectives inside)
        if rank == 0 and exposed is not None:
            out['config']['data_parallel']['allreduce_exposed_ms'] = round(exposed, 3)
            out['config']['data_parallel']['allreduce_exposed_note'] = ('median over 5 extra steps of (end of the communication lane) - (end of the compute '
                                                                       'lanes): the tail bucket\'s all-reduce + its Adam + shadow refresh, which nothing can hide')
    if rank == 0 and world == 1 and not args.no_extras and args.config == 1:
        out['extra'] = extras(eng, cfg, B, image_d, cap_d, dev)
        # BASELINE configs[3] and configs[4] in the record the driver writes (the headline above is untouched: its engine,
        # timed region, workload and dtype are as before; these run after it)
        del eng
        torch.cuda.empty_cache()
        progress('extra: config3 (ResNet-101 + 2-layer 1024-d LSTM, 384x384)')
        out['extra']['config3'] = config3_measure(dev)
        progress('extra: decode_beam5 (batch 128)')
        out['extra']['decode_beam5'] = decode_measure(128, 5, 10, 2, pipelined=3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.config == 3:
            out['cpu_baseline'] = cpu_baseline(cfg, budget_s=40.0, probe_batch=1, max_batch=4,
                                               label='BASELINE configs[3] workload (ResNet-101 + 2-layer 1024-d LSTM, 384x384, V 20000, L 30)')
        else:
            out['cpu_baseline'] = cpu_baseline(cfg)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if pg is not None:
        dist.barrier()
        if trainer is not None and trainer.native_comm is not None:
            torch.cuda.synchronize()
            trainer.native_comm.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
