"""CPU tests of the host layer: C-ABI library + header agreement, parameter layouts, LR schedules,
config compatibility, checkpoint files, and that the product refuses to run without a GPU."""
import ctypes
import importlib
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'capmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return re.findall(r'\b(?:int|long long|const char\*)\s+(capmi_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S)


def test_library_exports_every_header_symbol_with_matching_signature():
    from myimagecaptioningmodel_amd import _lib
    L = _lib.lib()                                 # loads libcapmi.so; raises if missing
    assert L.capmi_version() == 1
    decls = _header_decls()
    assert len(decls) >= 40
    for name, args in decls:
        assert hasattr(L, name), name              # exported by the shared library
        if name in ('capmi_version', 'capmi_last_error'):
            continue
        if name in ('capmi_igemm_tn_ws_bytes', 'capmi_igemm_nt_splitk_ws_bytes'):
            continue
        if name.startswith('capmi_plan_'):        # bound by hand in _lib.lib() (struct pointer / string results); used below
            assert getattr(L, name).argtypes is not None or name == 'capmi_plan_entry_count'
            continue
        sig = next((d[name] for d in (_lib.SIGNATURES, _lib.QUERIES, _lib.SYNC, _lib.COMM) if name in d), None)
        assert sig is not None, 'no ctypes signature for %s' % name
        arglist = [a.strip() for a in args.split(',') if a.strip() and a.strip() != 'void']
        assert len(sig) == len(arglist), (name, len(sig), len(arglist))
        for a, t in zip(arglist, sig):
            if 'capmi_conv_geom' in a:
                exp = _lib._g
            elif 'capmi_igemm_nt_call' in a:
                exp = ctypes.POINTER(_lib.NtCall)
            elif '*' in a:
                exp = ctypes.c_void_p
            elif a.startswith('int64_t'):
                exp = ctypes.c_int64
            elif a.startswith('long long'):
                exp = ctypes.c_longlong
            elif a.startswith('float'):
                exp = ctypes.c_float
            else:
                exp = ctypes.c_int
            assert t is exp, (name, a)
    # and nothing bound that the header does not declare
    declared = {n for n, _ in decls}
    assert set(_lib.SIGNATURES) | set(_lib.QUERIES) | set(_lib.SYNC) | set(_lib.COMM) <= declared
    # the plan entry table (capmi_plan_run) covers every stream-taking entry point, with the arity the binding has
    assert set(_lib.PLAN_ENTRIES) == set(_lib.SIGNATURES)
    assert ctypes.sizeof(_lib.Launch) == 16 + 8 * _lib.PLAN_MAX_ARGS
    assert max(len(v) for v in _lib.SIGNATURES.values()) - 1 <= _lib.PLAN_MAX_ARGS


def test_plan_table_packs_arguments_and_reports_bad_rows():
    """capmi_plan_run without a GPU: the packed table of a recorded plan (slots: pointers, two's-complement ints, float
    bits, host struct addresses; patch points for ctypes floats) and the runner's argument checks."""
    from myimagecaptioningmodel_amd import _lib
    L = _lib.lib()
    p = _lib.Plan()
    g = _lib.gemm_geom(8, 16)
    lrt = ctypes.c_float(0.25)
    p.add('capmi_igemm_nt', 4096, 8192, 12288, g, 32, 16, 32, None, None, 0, None, 0, None, 0, 0, 0, _lib.F32)
    p.add('capmi_adam', 64, 128, 192, 256, 1000, lrt, 0.9, 0.999, 1e-8, 0.0, 1.0)
    p.add('capmi_embedding_fwd', 64, 128, 192, 4, 8, 50, 24, -1, _lib.F32)
    c = p._compile(False)
    t = c['table']
    assert c['n'] == 3 and [t[i].kind for i in range(3)] == [0, 0, 0] and [t[i].lane for i in range(3)] == [0, 0, 0]
    assert t[0].entry == _lib.PLAN_ENTRIES['capmi_igemm_nt'] and t[0].nargs == 17
    assert t[0].args[0] == 4096 and t[0].args[3] == ctypes.addressof(g) and t[0].args[7] == 0
    assert t[1].args[5] == _lib._float_bits(0.25) and t[1].args[6] == _lib._float_bits(0.9)
    assert c['patches'] == [(1, 5, lrt)]
    assert t[2].args[7] == 0xFFFFFFFFFFFFFFFF                      # padding_idx = -1, two's complement
    streams = (ctypes.c_void_p * 3)(None, None, None)
    bad = (_lib.Launch * 1)()
    bad[0].kind, bad[0].lane, bad[0].entry, bad[0].nargs = 0, 0, _lib.PLAN_ENTRIES['capmi_adam'], 3
    assert L.capmi_plan_run(bad, 1, streams, 3) != 0 and 'capmi_adam takes 11' in _lib.last_error()
    bad[0].lane = 5
    assert L.capmi_plan_run(bad, 1, streams, 3) != 0 and 'lane 5' in _lib.last_error()
    bad[0].lane, bad[0].kind = 0, 9
    assert L.capmi_plan_run(bad, 1, streams, 3) != 0 and 'kind 9' in _lib.last_error()
    assert L.capmi_plan_run(bad, 0, streams, 3) == 0


def test_argument_errors_are_reported_not_thrown():
    from myimagecaptioningmodel_amd import _lib
    L = _lib.lib()
    rc = L.capmi_adam(None, None, None, None, 10, 0.1, 0.9, 0.999, 1e-8, 0.0, 1.0, None)
    assert rc != 0 and b'null pointer' in L.capmi_last_error()
    g = _lib.gemm_geom(4, 12)                      # K=12 is not a multiple of the 8-element bf16 vector
    rc = L.capmi_igemm_nt(1, 1, 1, g, 8, 12, 8, None, None, 0, None, 0, None, 0, 0, 0, _lib.BF16, None)
    assert rc != 0 and b'multiples' in L.capmi_last_error()
    assert L.capmi_igemm_nt_stats_part_rows(200704, 64, 64, _lib.BF16) in (64, 128)   # one part per workgroup row block


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    from myimagecaptioningmodel_amd import CapmiError, default_cfg
    from myimagecaptioningmodel_amd.model import CaptionEngine
    with pytest.raises(CapmiError):
        CaptionEngine(default_cfg(), device='cpu')
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            CaptionEngine(default_cfg(image_size=64, hidden=32, embed=16, vocab=50), device='cuda:0')


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'myimagecaptioningmodel_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), fn


# ------------------------------------------------------------------ parameter store
def test_param_store_names_layouts_and_round_trip():
    from myimagecaptioningmodel_amd import default_cfg, params as P
    from oracle import model as om
    cfg = default_cfg(image_size=64, hidden=32, embed=16, vocab=50, sentence_length=6)
    st = P.ParamStore(cfg, 'cpu')
    ocfg = om.default_cfg(image_size=64, hidden=32, embed=16, vocab=50, sentence_length=6)
    shapes = om.param_shapes(ocfg)
    assert set(st.names()) == set(shapes)                       # same variable names as the reference checkpoint
    for n, e in st.entries.items():
        assert tuple(shapes[n]) == e.ref_shape, n
        assert e.offset % P.ALIGN == 0
    # BN offset sits directly before the scale (bn_bwd_reduce writes [d offset | d scale] in place)
    for n, e in st.entries.items():
        if n.endswith('_bn_offset'):
            assert st.entries[n[:-6] + 'scale'].offset == e.offset + e.kshape[0]
    # decoder first, encoder in reverse layer order (= backward completion order)
    names = list(st.entries)
    assert names[0] == 'fc_0.w_0' and names.index('conv9_weights') < names.index('conv1_1_weights')
    rng = np.random.RandomState(0)
    ref = {n: rng.standard_normal(s).astype(np.float32) for n, s in shapes.items()}
    st.load_reference(ref)
    back = st.export_reference()
    for n in ref:
        np.testing.assert_array_equal(back[n], ref[n], err_msg=n)
    # kernel layouts
    w = ref['conv2_1_expand_weights']
    np.testing.assert_array_equal(st.view('conv2_1_expand_weights').numpy(), w.transpose(0, 2, 3, 1))
    np.testing.assert_array_equal(st.view('conv2_1_dwise_weights').numpy(), ref['conv2_1_dwise_weights'][:, 0].transpose(1, 2, 0))
    np.testing.assert_array_equal(st.view('fc_5.w_0').numpy(), ref['fc_5.w_0'].T)
    np.testing.assert_array_equal(st.view('lstm_w').numpy(), ref['lstm_w'].T)
    stem = st.view('conv1_1_weights').numpy()                   # space-to-depth form [Cout][kt][kt][Cs] (capmi_s2d_stem)
    wref = ref['conv1_1_weights']
    assert stem.shape == (32, 2, 2, 16) and np.count_nonzero(stem) <= wref.size
    for r in range(3):
        for q in range(3):
            sub = (r % 2) * 2 + (q % 2)
            np.testing.assert_array_equal(stem[:, r // 2, q // 2, sub * 3:sub * 3 + 3], wref[:, :, r, q])
    assert np.all(stem[:, :, :, 12:] == 0) and np.all(stem[:, 1, :, 6:12] == 0) and np.all(stem[:, :, 1, 3:6] == 0)


def test_frozen_encoder_is_excluded_from_the_optimizer_range():
    from myimagecaptioningmodel_amd import default_cfg, params as P
    st = P.ParamStore(default_cfg(image_size=64, hidden=32, embed=16, vocab=50, encoder_trainable=False), 'cpu')
    assert st.trainable_size == st.decoder_size < st.size
    assert not st.entries['conv9_weights'].trainable and st.entries['lstm_w'].trainable


def test_encoder_topologies():
    from myimagecaptioningmodel_amd import arch
    from oracle import arch as oarch
    m = arch.mobilenet_v2()
    convs = [o for o in m.ops if isinstance(o, arch.ConvBN)]
    assert len(convs) == 53 and m.channels == 1280            # conv1_1 + 17 units x 3 + conv9 (53 conv-bn)
    assert sum(1 for o in convs if o.groups > 1) == 17
    onames = {o[1] for o in oarch.mobilenet_v2_ops()[0] if o[0] == 'conv_bn'}
    assert {o.name for o in convs} == onames
    r = arch.resnet(50)
    rconvs = [o for o in r.ops if isinstance(o, arch.ConvBN)]
    assert len(rconvs) == 53 and r.channels == 2048
    assert {o.name for o in rconvs} == {o[1] for o in oarch.resnet_ops(50)[0] if o[0] == 'conv_bn'}
    with pytest.raises(ValueError):
        arch.encoder('vgg')


# ------------------------------------------------------------------ LR schedules (tools/util.py:20-119)
def test_lr_schedules():
    import math
    from myimagecaptioningmodel_amd.optim import LRSchedule, adam_lr_t
    with pytest.raises(ValueError):
        LRSchedule('linear', 1e-3, 1000, 10)
    s = LRSchedule(None, 5e-5, 944996, 128)
    assert s.step_each_epoch == 7383 and s.value(0) == s.value(10 ** 6) == 5e-5
    c = LRSchedule('cosine_decay', 1e-3, 1000, 10, decay_epoch=10)
    assert c.value(0) == pytest.approx(1e-3) and c.value(100 * 5) == pytest.approx(1e-3 * 0.5 * (math.cos(math.pi * 5 / 10) + 1))
    w = LRSchedule('cosine_decay_warmup', 1e-3, 1000, 10, warmup_epoch=3, max_epoch=10)
    assert w.value(0) == pytest.approx(1e-5)                            # epoch 0 of warm-up: start_lr
    assert w.value(99) == pytest.approx(1e-5 + (1e-3 - 1e-5) / 3 * 1)   # counter starts at 1: step 99 -> epoch 1
    assert w.value(100 * 3) == pytest.approx(1e-3)                      # first post-warm-up epoch
    r = LRSchedule('cosine_decay_restart', 1e-3, 1000, 10, decay_epoch=2)
    assert r.value(0) == pytest.approx(1e-3) and r.value(100 * 2) == pytest.approx(1e-3)     # restart at epoch 2
    assert r.value(100 * 1) == pytest.approx(1e-3 * 0.5 * (math.cos(math.pi * 0.5) + 1))
    rw = LRSchedule('cosine_decay_restart_warmup', 1e-3, 1000, 10, decay_epoch=2, warmup_epoch=2)
    vals = [rw.value(i) for i in range(400)]
    assert vals[0] == pytest.approx(1e-5) and vals[99] == pytest.approx(1e-5 + (1e-3 - 1e-5) * 0.5)
    assert vals[199] == pytest.approx(1e-3)
    assert adam_lr_t(1e-3, 1) == pytest.approx(1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9))


# ------------------------------------------------------------------ config compatibility
def _reference_style_config():
    """A module of dicts with the reference's schema (config.py:2-73), values = the repo defaults."""
    m = types.ModuleType('config')
    m.data = {'ImageShape': [224, 224], 'ImageMean': [0, 0, 0], 'ImageStd': [1, 1, 1], 'start_idx': 2, 'stop_idx': 3,
              'padding_idx': 0, 'PretrainedMobileNetPath': None, 'sample_count': 944996}
    m.train = {'seed': None, 'learning_rate': 0.00005, 'lr_decay_strategy': None, 'decay_epoch': 0, 'warmup_epoch': 3,
               'gradient_clip': False, 'batch_size': 128, 'max_epoch': 10, 'log_every_n_step': 150}
    m.model = {'encoder': {'encoder_trainable': True, 'encoder_dim': 49, 'encoder_channel': 1280},
               'decoder': {'vocab_size': 12295, 'embedding_size': 256, 'sentence_length': 35, 'hidden_dim': 1024,
                           'infer_max_length': 35}}
    m.dc, m.md = m.data, m.model
    return m


def test_reference_config_module_is_read_unchanged():
    from myimagecaptioningmodel_amd import default_cfg, from_reference_config
    cfg = from_reference_config(_reference_style_config())
    d = default_cfg()
    for k in ('image_size', 'hidden', 'embed', 'vocab', 'sentence_length', 'infer_max_length', 'start_idx', 'stop_idx',
              'padding_idx', 'encoder_trainable', 'learning_rate', 'batch_size', 'sample_count'):
        assert cfg[k] == d[k], k
    assert cfg['attention'] == 'singleton' and cfg['dtype'] == 'f32' and cfg['encoder'] == 'mobilenetv2'
    assert from_reference_config(_reference_style_config(), encoder='resnet50', dtype='bf16')['encoder'] == 'resnet50'


def test_facade_modes_without_gpu():
    from myimagecaptioningmodel_amd import ImageCaptionModel, default_cfg
    m = ImageCaptionModel(default_cfg())
    inputs, feeds = m.build_input('train')
    assert feeds[0].shape == [-1, 3, 224, 224] and feeds[1].shape == [-1, 35] and feeds[1].dtype == 'int64'
    assert m.build_network('train', **inputs).name == 'loss'
    assert m.build_network('eval').dtype == 'float32'          # float ids, quirk Q2
    for bad in ('test', 'infer'):
        with pytest.raises(ValueError):
            m.build_input(bad)
        with pytest.raises(ValueError):
            m.build_network(bad)


# ------------------------------------------------------------------ checkpoint files
def test_lod_tensor_round_trip_and_layout(tmp_path):
    from myimagecaptioningmodel_amd import ckpt
    rng = np.random.RandomState(0)
    for arr in (rng.standard_normal((3, 4, 5)).astype(np.float32), rng.standard_normal(7), np.arange(6, dtype=np.int64).reshape(2, 3),
                np.array([0.729], np.float32)):
        p = str(tmp_path / 'v')
        ckpt.write_lod_tensor(p, arr)
        back = ckpt.read_lod_tensor(p)
        assert back.dtype == arr.dtype and back.shape == arr.shape
        np.testing.assert_array_equal(back, arr)
    # byte layout: u32 0 | u64 0 | u32 0 | i32 desc_len | desc | data
    ckpt.write_lod_tensor(str(tmp_path / 'w'), np.zeros((2, 3), np.float32))
    raw = open(str(tmp_path / 'w'), 'rb').read()
    assert raw[:16] == b'\x00' * 16 and raw[16:20] == (6).to_bytes(4, 'little')
    assert raw[20:26] == b'\x08\x05\x10\x02\x10\x03' and len(raw) == 26 + 24
    conf = ckpt.load_resume_state(str(tmp_path / 'log'))
    assert conf == {'epoch': 1, 'best_bleu': 0, 'best_meteor': 0, 'train_encoder': True}
    conf['epoch'] = 4
    ckpt.save_resume_state(str(tmp_path / 'log'), conf)
    assert ckpt.load_resume_state(str(tmp_path / 'log'))['epoch'] == 4
    open(str(tmp_path / 'conv9_weights'), 'wb').close()
    assert ckpt.predicate_existing(str(tmp_path), ['conv9_weights', 'missing']) == ['conv9_weights']


def test_plan_lanes_keep_a_valid_sequential_order():
    """Plan lanes: side-lane launches and record/wait marks are extra entries; `launches()` (what every
    single-stream consumer replays) is the plain launch sequence, and slices keep their marks."""
    from myimagecaptioningmodel_amd._lib import Plan
    p = Plan()
    p.add('capmi_fill_f32', 1, 0.0, 4)
    p.record(('dz', 'a'), 0)
    p.wait(('dz', 'a'), 1)
    p.add('capmi_fill_f32', 2, 0.0, 4, lane=1)
    p.record(('wgrad', 'a'), 1)
    p.wait(('wgrad', 'a'), 0)
    p.add('capmi_fill_f32', 3, 0.0, 4)
    assert p.has_lanes and len(p) == 7 and len(p._keep) == 7
    names = [(n, a[0]) for _, n, a in p.launches()]
    assert names == [('capmi_fill_f32', 1), ('capmi_fill_f32', 2), ('capmi_fill_f32', 3)]
    assert [getattr(fn, 'lane', 0) for fn, _, _ in p.launches()] == [0, 1, 0]
    q = Plan()
    q.extend(p)
    assert q.has_lanes and len(q.launches()) == 3


def test_device_feeder_keeps_the_reader_contract():
    """reader.py:41-76: fixed sample order, batches of (image, caption) samples, a short last batch, fp16-stored
    pixels widened to float32 -- served through the double-buffered feeder (CPU device here: same logic, no pinning)."""
    import torch
    from myimagecaptioningmodel_amd.feeder import DeviceFeeder, collate
    rng = np.random.RandomState(0)
    samples = [(rng.rand(3, 8, 8).astype(np.float16), np.arange(5) + i) for i in range(11)]

    def batches(bs):
        for i in range(0, len(samples), bs):
            yield samples[i:i + bs]

    for depth in (1, 2, 4):
        # a handed-out batch aliases its slot's buffers: valid until the next batch is requested
        out = [(i.clone(), c.clone()) for i, c in DeviceFeeder(batches(4), device='cpu', depth=depth)]
        assert [tuple(o[0].shape) for o in out] == [(4, 3, 8, 8), (4, 3, 8, 8), (3, 3, 8, 8)]
        assert all(o[0].dtype == torch.float32 and o[1].dtype == torch.int64 for o in out)
        assert [o[1][:, 0].tolist() for o in out] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]
        for i, o in enumerate(out):
            want = np.stack([s[0] for s in samples[4 * i:4 * i + 4]]).astype(np.float32)
            np.testing.assert_array_equal(o[0].numpy(), want)
    img, cap = collate((np.zeros((2, 3, 4, 4), np.float32), np.ones((2, 6), np.int32)))     # pre-stacked form
    assert img.shape == (2, 3, 4, 4) and cap.dtype == np.int64

    def broken():
        yield samples[:4]
        raise IOError('store went away')

    f = DeviceFeeder(broken(), device='cpu', depth=2)
    assert tuple(next(f)[0].shape) == (4, 3, 8, 8)
    with pytest.raises(IOError):                 # a reader failure surfaces at its position, not as a silent end
        next(f)
    with pytest.raises(StopIteration):
        next(f)
    f = DeviceFeeder(batches(4), device='cpu', depth=1)
    next(f)
    f.close()
    f.worker.join(timeout=5)
    assert not f.worker.is_alive()


# ------------------------------------------------------------------ LR schedule state + checkpoint resume (host logic, CPU)
def _stateful_reference_lr(strategy, base_lr, spe, steps, decay_epoch, warmup_epoch, max_epoch):
    """Run-by-run simulation of the reference's graph ops (tools/util.py:47-119) with their persistable state:
    `@LR_DECAY_COUNTER@` incremented inside every run, `cur_epoch` += 1 when counter % spe == 0."""
    import math
    begin = 1 if strategy in ('cosine_decay_warmup', 'cosine_decay_restart_warmup') else 0
    counter, cur_epoch_var, out = begin - 1, 0.0, []

    def restart(cf, t_mul=2.0):
        i = math.floor(math.log(1.0 - cf * (1.0 - t_mul)) / math.log(t_mul))
        return (cf - (1.0 - t_mul ** i) / (1.0 - t_mul)) / t_mul ** i
    for _ in range(steps):
        counter += 1
        g = float(counter)
        if strategy == 'cosine_decay':
            lr = base_lr * 0.5 * (math.cos(math.floor(g / spe) * math.pi / decay_epoch) + 1)
        elif strategy == 'cosine_decay_warmup':
            ce = math.floor(g / spe)
            lr = 1e-5 + (base_lr - 1e-5) / warmup_epoch * ce if ce < warmup_epoch else \
                0.5 * base_lr * (math.cos((ce - warmup_epoch) * math.pi / float(max_epoch - warmup_epoch)) + 1)
        elif strategy == 'cosine_decay_restart':
            lr = base_lr * 0.5 * (math.cos(math.pi * restart(math.floor(g / spe) / decay_epoch)) + 1)
        else:
            if counter % spe <= 0:
                cur_epoch_var += 1
            lr = 1e-5 + (base_lr - 1e-5) * (cur_epoch_var / float(warmup_epoch)) if cur_epoch_var < warmup_epoch else \
                base_lr * 0.5 * (math.cos(math.pi * restart((cur_epoch_var - warmup_epoch) / decay_epoch)) + 1)
        out.append(lr)
    return out, counter, cur_epoch_var


@pytest.mark.parametrize('strategy', ['cosine_decay', 'cosine_decay_warmup', 'cosine_decay_restart', 'cosine_decay_restart_warmup'])
def test_lr_value_is_a_pure_function_of_the_step(strategy):
    from myimagecaptioningmodel_amd.optim import LRSchedule
    spe, steps = 7, 80
    s = LRSchedule(strategy, 1e-3, 70, 10, decay_epoch=3, warmup_epoch=2, max_epoch=10)
    assert s.step_each_epoch == spe
    want, counter, cur_epoch = _stateful_reference_lr(strategy, 1e-3, spe, steps, 3, 2, 10)
    got_fwd = [s.value(i) for i in range(steps)]
    got_rev = [s.value(i) for i in reversed(range(steps))][::-1]        # any order, any number of calls
    assert got_fwd == got_rev == [s.value(i) for i in range(steps)]
    np.testing.assert_allclose(got_fwd, want, rtol=1e-12)
    assert s.counter_after(steps) == counter and s.steps_from_counter(counter) == steps
    if strategy == 'cosine_decay_restart_warmup':
        assert s.cur_epoch_after(steps) == cur_epoch


class _HostEngine:
    """What ckpt.save_persistables / load_persistables touch of an engine, over a CPU ParamStore."""

    def __init__(self, cfg, seed):
        from myimagecaptioningmodel_amd.optim import LRSchedule
        from myimagecaptioningmodel_amd.params import ParamStore
        self.cfg = cfg
        self.store = ParamStore(cfg, 'cpu')
        self.store.init_reference(seed)
        self.step_count = 0
        self.lr_schedule = LRSchedule(cfg.get('lr_decay_strategy'), 1e-3, cfg.get('sample_count', 0), cfg.get('batch_size', 1),
                                      cfg.get('decay_epoch', 0), cfg.get('warmup_epoch', 3), cfg.get('max_epoch', 10))

    def export_reference_params(self):
        return self.store.export_reference()

    def load_reference_params(self, params):
        self.store.load_reference(params)


@pytest.mark.parametrize('steps', [0, 5, 1000, 22149])
@pytest.mark.parametrize('strategy', [None, 'cosine_decay_restart_warmup'])
def test_checkpoint_round_trip_restores_step_moments_and_counters(tmp_path, steps, strategy):
    """the step count used to be recovered from float32 0.9^(t+1), which is 0 from t ~ 980."""
    import torch
    from myimagecaptioningmodel_amd import ckpt
    from myimagecaptioningmodel_amd.config_compat import default_cfg
    cfg = default_cfg(encoder='mobilenetv2', image_size=64, hidden=16, embed=8, vocab=20, sentence_length=5, infer_max_length=5,
                      lr_decay_strategy=strategy, decay_epoch=2, warmup_epoch=1, sample_count=70, batch_size=10)
    a, b = _HostEngine(cfg, 0), _HostEngine(cfg, 1)
    g = torch.Generator().manual_seed(3)
    a.store.adam_m.copy_(torch.randn(a.store.size, generator=g))
    a.store.adam_v.copy_(torch.rand(a.store.size, generator=g))
    for t in a.store.state.values():
        t.copy_(torch.rand(t.shape, generator=g) + 0.5)
    a.step_count = steps
    d = str(tmp_path / 'checkpoint')
    ckpt.save_persistables(a, d)
    assert os.path.isfile(os.path.join(d, '@LR_DECAY_COUNTER@')) and os.path.isfile(os.path.join(d, 'lstm_w_moment1_0'))
    assert os.path.isfile(os.path.join(d, 'cur_epoch')) == (strategy is not None)
    ckpt.load_persistables(b, d)
    assert b.step_count == steps
    for buf in ('flat', 'adam_m', 'adam_v'):       # compared in the reference layout (the flat buffers have alignment gaps)
        x, y = a.store.export_reference(getattr(a.store, buf)), b.store.export_reference(getattr(b.store, buf))
        assert all(np.array_equal(x[k], y[k]) for k in a.store.entries), buf
    for k in a.store.state:
        assert torch.equal(a.store.state[k], b.store.state[k])
    assert b.lr_schedule.value(b.step_count) == a.lr_schedule.value(steps)
    # a reference checkpoint trained without a decay strategy has no counter file: beta2_pow_acc in float64
    os.remove(os.path.join(d, '@LR_DECAY_COUNTER@'))
    c = _HostEngine(cfg, 2)
    ckpt.load_persistables(c, d)
    assert abs(c.step_count - steps) <= max(0, int(steps * 6e-5)) and (steps > 1700 or c.step_count == steps)
    # params-only export (train.py:78-79) and the pretrained-encoder predicate load (train.py:97-99)
    ckpt.save_params(a, str(tmp_path / 'params'))
    assert not os.path.exists(str(tmp_path / 'params' / 'lstm_w_moment1_0'))
    e = _HostEngine(cfg, 4)
    enc_only = str(tmp_path / 'pretrained')
    os.makedirs(enc_only)
    for n in ('conv1_1_weights', 'conv9_weights'):
        os.link(str(tmp_path / 'params' / n), os.path.join(enc_only, n))
    assert sorted(ckpt.load_vars_existing(e, enc_only)) == ['conv1_1_weights', 'conv9_weights']
    assert torch.equal(e.store.view('conv9_weights'), a.store.view('conv9_weights'))
    assert not torch.equal(e.store.view('lstm_w'), a.store.view('lstm_w'))


def test_no_lds_dma_refill_over_unretired_fragment_reads():
    """Static screen of the gfx950 assembly of the GEMM kernels (tools/lds_war_audit.py, DESIGN.md lesson 29): no LDS-DMA
    issue may follow a barrier while ds_reads issued before that barrier are not yet retired by lgkmcnt(0) -- the
    write-after-read on LDS that made the halo-staged 3x3 kernel irreproducible inside the two-lane step."""
    import shutil
    import subprocess
    import sys
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not on PATH')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'lds_war_audit.py'),
                          os.path.join(root, 'myimagecaptioningmodel_amd', 'csrc', 'igemm.hip')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith('0 kernels flagged'), out.stdout[-2000:]


@pytest.mark.parametrize('which', ['configs2', 'configs3'])
def test_gradient_bucket_plan_of_the_data_parallel_configs_at_world_eight(which):
    """BASELINE configs[2] (ResNet-50 + 512-d decoder, 256 images over 8 GPUs) and configs[3] (ResNet-101 + 2-layer 1024-d
    decoder, 512 over 8): the bucket plan OverlappedTrainer exchanges per step -- it depends on the parameter layout only, so
    it is checked here without a GPU.  Buckets tile the trainable range in backward-completion order (decoder first), are cut
    at layer boundaries only, every one but the tail is >= 32 MiB, the tail (the only all-reduce nothing can hide) is a few
    MB, the payload is what SURVEY.md 8(e) budgets, and the 1/N of ParallelExecutor's CoeffNumDevice (train.py:121-124) is the
    Adam kernel's gradient scale at world 8."""
    import torch
    import bench
    from myimagecaptioningmodel_amd import default_cfg, dp
    from myimagecaptioningmodel_amd.params import ParamStore
    workload, per_gpu = (bench.WORKLOAD, 32) if which == 'configs2' else (bench.WORKLOAD_CFG3, 64)
    world = 8
    cfg = default_cfg(batch_size=per_gpu * world, sample_count=0, **workload)
    st = ParamStore(cfg, torch.device('cpu'))
    plan = dp.bucket_plan(st)
    cuts = dp.bucket_cut_points(st)
    r = plan.ranges
    assert r[0][0] == 0 and r[-1][1] == st.trainable_size == st.size
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert all(e in cuts or e == st.trainable_size for _, e in r)
    assert r[0][1] >= st.decoder_size                                  # the decoder's gradients are final first: they open the exchange
    mb = [(e - b) * 4 / 2 ** 20 for b, e in r]
    assert all(m >= 32.0 for m in mb[:-2]) and mb[-1] <= 4.5 and mb[-1] >= 2.0, mb
    total_mb = st.trainable_size * 4 / 1e6
    assert (140 < total_mb < 150) if which == 'configs2' else (390 < total_mb < 410), total_mb      # f32 payload: 146 MB / 398 MB per step
    assert len(r) == (5 if which == 'configs2' else 7), (len(r), mb)
    # 1/N: Adam on the SUM of eight identical per-rank gradients with grad_scale 1/8 == Adam on one of them
    from oracle import ops
    p0 = np.linspace(-1, 1, 64)
    g = np.cos(np.arange(64.0))
    a, _, _ = ops.adam_update(p0, (8 * g) * (1.0 / world), np.zeros(64), np.zeros(64), 1e-3, 1)
    b, _, _ = ops.adam_update(p0, g, np.zeros(64), np.zeros(64), 1e-3, 1)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-15)
