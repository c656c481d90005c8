"""End-to-end parity on the MI355X: the engine (HIP kernels through the C ABI) against the
NumPy oracle on the same seeded inputs and parameters -- loss, logits, every parameter
gradient, Paddle-form Adam updates, BN running statistics and greedy token ids.

Tolerances (stated per BASELINE.json's north_star: loss within 1e-3, argmax ids bit-exact):
  f32 engine vs f64 oracle: |loss| <= 1e-4; all gradients together <= max(2e-3, 10 x the noise floor) relative L2, where
  the noise floor is the SAME oracle run in f32 against itself in f64 (measured 2e-3 .. 3e-2: batch norm at random
  initialisation amplifies f32 rounding); each tensor <= max(5 %, 20 x its own f32 noise) -- the test
  prints the worst tensor and the worst error-to-noise ratio (observed: 0.6 .. 14, i.e. the kernels sit AT the f32 noise
  of the model, which is what bounds a per-tensor claim, not the kernels); greedy ids bit-exact;
  bf16 engine: |loss| <= 5e-2, gradient direction cos >= 0.97.
"""
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from tests.conftest import make_caption

pytestmark = pytest.mark.gpu


def _cfgs(encoder, attention, dtype, S=64, H=32, E=16, V=50, L=6):
    ocfg = om.default_cfg(encoder=encoder, image_size=S, hidden=H, embed=E, vocab=V, sentence_length=L,
                          infer_max_length=L, attention=attention)
    from myimagecaptioningmodel_amd import default_cfg
    ecfg = default_cfg(encoder=encoder, image_size=S, hidden=H, embed=E, vocab=V, sentence_length=L,
                       infer_max_length=L, attention=attention, dtype=dtype, learning_rate=1e-4, batch_size=4)
    return ocfg, ecfg


def _data(ocfg, B, seed):
    rng = np.random.RandomState(seed)
    params = om.init_params(ocfg, seed=seed, dtype=np.float64)
    for k in params:      # move BN affine / biases off their init point so their gradients are exercised
        if k.endswith('_bn_scale') or k.endswith('_bn_offset') or k.endswith('.b_0') or k in ('lstm_b', 'out_fc_bias'):
            params[k] = params[k] + 0.1 * rng.standard_normal(params[k].shape)
    S = ocfg['image_size']
    image = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    caption = make_caption(rng, B, ocfg['sentence_length'], ocfg['vocab'])
    return params, image, caption


def _engine(ecfg, params, use_graph=False):
    from myimagecaptioningmodel_amd.model import CaptionEngine
    eng = CaptionEngine(ecfg, device='cuda:0', use_graph=use_graph)
    eng.load_reference_params(params)
    return eng


def _f32_oracle_noise(ocfg, params, image, caption):
    """|f32 NumPy evaluation - f64 evaluation| of the same graph: the rounding noise floor of this
    input (batch norm over a few dozen samples amplifies f32 rounding by orders of magnitude)."""
    o32 = om.OracleModel(ocfg, {k: v.astype(np.float32) for k, v in params.items()})
    l32, _ = o32.forward_train(image.astype(np.float32), caption, update_stats=False)
    return float(l32), o32.backward()


@pytest.mark.parametrize('encoder,attention,S', [('mobilenetv2', 'singleton', 96), ('mobilenetv2', 'slots', 96), ('resnet50', 'slots', 64)])
def test_f32_train_step_matches_oracle(encoder, attention, S):
    ocfg, ecfg = _cfgs(encoder, attention, 'f32', S=S)
    B = 6
    params, image, caption = _data(ocfg, B, seed=3)
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    eng = _engine(ecfg, params)
    for step in range(2):
        l32, g32 = _f32_oracle_noise(ocfg, oracle.p, image, caption)
        loss_o, logits_o = oracle.forward_train(image.astype(np.float64), caption)
        grads_o = oracle.backward()
        loss_e = eng.forward_backward(image, caption)
        torch.cuda.synchronize()
        le = float(loss_e.cpu()[0])
        assert abs(le - loss_o) <= max(1e-4, 5 * abs(l32 - loss_o)), (step, le, loss_o, l32)
        assert abs(le - loss_o) <= 1e-3                      # BASELINE.json north_star bound
        # logits [M, Vld] time-major f32 -> [B,T,V]
        dec = eng._train[B]['dec']
        lg = dec.logits.cpu().numpy()[:, :ocfg['vocab']].reshape(dec.T, B, -1).transpose(1, 0, 2)
        assert np.abs(lg - logits_o).max() <= 2e-3 * max(1.0, np.abs(logits_o).max())
        grads_e = eng.export_reference_grads()
        # Relative L2 errors.  (A max-abs metric is brittle here: a ReLU mask flips when a
        # pre-activation lies within f32 rounding of 0, which moves single gradient elements by
        # O(1/pixels) -- two runs of the SAME engine differ that way through atomic ordering.)
        #  * all gradients together: <= 10x the f32-NumPy noise floor of this very input (>= 2e-3)
        #  * each tensor: <= 5 % or 20x its own noise floor, with an absolute floor for tensors whose
        #    true gradient is ~0 (e.g. a BN offset that feeds straight into the next batch norm)
        gscale = max(np.abs(g).max() for g in grads_o.values())
        tot_err = np.sqrt(sum(np.sum((grads_e[n] - g) ** 2) for n, g in grads_o.items()))
        tot_noise = np.sqrt(sum(np.sum((g32[n].astype(np.float64) - g) ** 2) for n, g in grads_o.items()))
        tot = np.sqrt(sum(np.sum(g ** 2) for g in grads_o.values()))
        print('step %d total relative-L2 gradient error %.2e (f32 NumPy noise floor %.2e)' % (step, tot_err / tot, tot_noise / tot))
        assert tot_err / tot <= max(2e-3, 10 * tot_noise / tot), (step, tot_err / tot, tot_noise / tot)
        worst, ratio = (0.0, None), (0.0, None)
        for name, go in grads_o.items():
            ge = grads_e[name]
            floor = 1e-6 * gscale * np.sqrt(go.size)
            nrm = np.linalg.norm(go)
            err = np.linalg.norm(ge - go)
            noise = np.linalg.norm(g32[name].astype(np.float64) - go)
            assert err <= max(5e-2 * nrm, 20 * noise, floor), (step, name, err / (nrm + 1e-30), noise / (nrm + 1e-30))
            if nrm > floor:
                worst = max(worst, (err / nrm, name))
                ratio = max(ratio, (err / max(noise, 2e-3 * nrm, floor), name))
        print('step %d worst relative-L2 gradient error %.2e (%s); worst error / max(f32 noise of that tensor, 2e-3) = %.2f (%s)' % (
            step, worst[0], worst[1], ratio[0], ratio[1]))
        # the bound on the printed figure: no tensor further from the f64 oracle than 20 x what the f32 evaluation of the
        # same graph in NumPy is (observed 0.6 .. 14 -- the kernels sit at the f32 noise of this input, not above it);
        # recorded next to the test so that a drift shows up as a number, not only as a pass
        assert ratio[0] <= 20.0, (step, ratio)
        rec = os.environ.get('CAPMI_TEST_RECORD')
        if rec:
            with open(rec, 'a') as fh:
                fh.write('test_f32_train_step_matches_oracle step %d: worst error-to-noise ratio %.3f (%s), worst relative L2 %.3e (%s)\n'
                         % (step, ratio[0], ratio[1], worst[0], worst[1]))
        if attention == 'singleton':      # quirk Q1: exactly zero gradient, parameters never move
            for n in ('fc_3', 'fc_8', 'fc_9', 'fc_10'):
                assert np.all(grads_e[n + '.w_0'] == 0) and np.all(grads_e[n + '.b_0'] == 0)
        # Adam on IDENTICAL gradients (the oracle's), so the optimizer arithmetic is what is compared
        eng.store.load_reference({})       # no-op; gradients are overwritten below
        for name, e in eng.store.entries.items():
            from myimagecaptioningmodel_amd.params import to_kernel
            eng.store.gview(name).copy_(torch.from_numpy(to_kernel(grads_o[name].astype(np.float32), e.kind)))
        oracle.adam_step(grads_o, lr=1e-4)
        eng.optimizer_step()
        pe = eng.export_reference_params()
        for name, po in oracle.p.items():
            err = np.abs(pe[name] - po).max()
            tol = 1e-3 * max(1.0, np.abs(po).max()) if name.endswith(('_mean', '_variance')) else 2e-5 * max(1.0, np.abs(po).max())
            assert err <= tol, (step, name, err)
        # re-sync so the second step compares the two implementations, not accumulated drift
        eng.load_reference_params({k: v for k, v in oracle.p.items()})


@pytest.mark.parametrize('attention', ['singleton', 'slots'])
def test_f32_greedy_ids_bit_exact(attention):
    ocfg, ecfg = _cfgs('mobilenetv2', attention, 'f32', L=8)
    B = 4
    params, image, _ = _data(ocfg, B, seed=5)
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    ids_o, logits_o = oracle.greedy_decode(image.astype(np.float64))
    eng = _engine(ecfg, params)
    ids_e = eng.decode(image).cpu().numpy()
    assert ids_e.dtype == np.float32 and ids_e.shape == (B, ocfg['infer_max_length'])      # quirk Q2
    # guard against near-ties in the oracle itself: only then may the comparison be skipped
    top2 = np.sort(logits_o, axis=-1)[..., -2:]
    assert (top2[..., 1] - top2[..., 0]).min() > 1e-3, 'test inputs have a near-tie; change the seed'
    np.testing.assert_array_equal(ids_e, ids_o)
    # running statistics were updated by the eval graph too (quirk Q3)
    pe = eng.export_reference_params()
    assert np.abs(pe['conv9_bn_mean'] - oracle.p['conv9_bn_mean']).max() < 1e-3


def test_graph_replay_equals_eager(deterministic):
    """hipGraph replay of the captured fwd+bwd launch sequence reproduces the eager launches -- bit for bit in deterministic
    mode (every f32 atomic accumulation has a fixed-order twin there, include/capmi.h), like the other schedule tests."""
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', 'f32', S=96)
    B = 6
    params, image, caption = _data(ocfg, B, seed=9)
    e1, e2 = _engine(ecfg, params, use_graph=False), _engine(ecfg, params, use_graph=True)
    for it in range(3):       # call 1 warms up + captures, calls 2-3 replay the hipGraph
        l1 = float(e1.forward_backward(image, caption).cpu()[0])
        l2 = float(e2.forward_backward(image, caption).cpu()[0])
        assert l1 == l2, (it, l1, l2)
        g1, g2 = e1.export_reference_grads(), e2.export_reference_grads()
        for k in g1:
            np.testing.assert_array_equal(g1[k], g2[k], err_msg='%s (call %d)' % (k, it))
    # the forward plan replays from a hipGraph (a laned backward plan is launched on two streams instead)
    assert any(str(k).startswith('graph') and v for k, v in e2._train[B].items())
    # full steps through the graph path stay finite and reduce the loss
    first = float(e2.train_step(image, caption)[0].cpu()[0])
    for _ in range(20):
        last = float(e2.train_step(image, caption)[0].cpu()[0])
    assert np.isfinite(last) and last < first


@pytest.mark.parametrize('encoder,S', [('mobilenetv2', 128), ('resnet50', 128)])
def test_bf16_train_step_close_to_oracle(encoder, S):
    ocfg, ecfg = _cfgs(encoder, 'slots', 'bf16', S=S, H=64, E=32, V=100)
    B = 8
    params, image, caption = _data(ocfg, B, seed=4)
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    loss_o, _ = oracle.forward_train(image.astype(np.float64), caption)
    grads_o = oracle.backward()
    eng = _engine(ecfg, params)
    loss_e = float(eng.forward_backward(image, caption).cpu()[0])
    print('bf16 loss %.5f oracle %.5f' % (loss_e, loss_o))
    assert abs(loss_e - loss_o) <= 5e-2, (loss_e, loss_o)
    grads_e = eng.export_reference_grads()
    for name in ('lstm_w', 'word_embedding', 'fc_0.w_0', 'fc_11.w_0', 'fc_7.w_0', 'fc_12.w_0'):
        a, b = grads_e[name].ravel(), grads_o[name].ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        print('bf16 grad cosine %s %.4f' % (name, cos))
        # fc_0's gradient is formed from the encoder OUTPUT (50+ bf16 layers of rounding noise)
        assert cos >= (0.90 if name == 'fc_0.w_0' else 0.97), (name, cos)


def test_reference_shaped_facade_and_errors():
    from myimagecaptioningmodel_amd import ImageCaptionModel, Executor
    ocfg, ecfg = _cfgs('mobilenetv2', 'singleton', 'f32')
    model = ImageCaptionModel(ecfg, device='cuda:0', use_graph=False)
    with pytest.raises(ValueError):
        model.build_input('test')                      # model_adaAttention_aic.py:144-145
    with pytest.raises(ValueError):
        model.build_network('infer')                   # :154-155
    inputs, feed_list = model.build_input('train')
    assert [v.name for v in feed_list] == ['image', 'caption'] and feed_list[1].dtype == 'int64'
    loss_var = model.build_network('train', **inputs)
    assert loss_var.name == 'loss'
    exe = Executor(model)
    params, image, caption = _data(ocfg, 4, seed=2)
    step_loss, lr = exe.run(feed={'image': image, 'caption': caption}, fetch_list=[loss_var, exe.lr_var])
    assert step_loss.shape == (1,) and step_loss.dtype == np.float32 and lr.shape == (1,) and abs(lr[0] - 1e-4) < 1e-9
    assert not np.isnan(step_loss).any()               # the check train.py:140-141 performs
    cap_var = model.build_network('eval')
    ids = exe.run(feed={'image': image}, fetch_list=[cap_var])[0]
    assert ids.dtype == np.float32 and ids.shape == (4, ecfg['infer_max_length'])
    with pytest.raises(ValueError):
        exe.run(feed={'image': image[:, :, :32]}, fetch_list=[cap_var])


@pytest.mark.parametrize('encoder', ['resnet50', 'mobilenetv2'])
def test_weight_shadows_are_exact_transposes(encoder):
    """The one-launch shadow refresh (capmi_cast + capmi_weight_dgrad_form_batched) against NumPy:
    wT[c][r'][q'][n] = W[n][rmap[r']][qmap[q']][c], bit-exact after the same f32 -> bf16 rounding."""
    ocfg, ecfg = _cfgs(encoder, 'slots', 'bf16', H=48, E=40, V=77)
    params, _, _ = _data(ocfg, 2, 5)
    eng = _engine(ecfg, params)
    eng.refresh_shadows()
    torch.cuda.synchronize()
    st = eng.store
    checked = 0
    for key, (off, (n, kh, kw, c, ldt), rmap, qmap) in eng.wT_entries.items():
        name = key if isinstance(key, str) else key[0]
        e = st.entries[name]
        w = st.flat[e.offset:e.offset + n * kh * kw * c].view(n, kh, kw, c)
        want = torch.zeros(c, len(rmap), len(qmap), ldt, dtype=torch.float32, device=w.device)
        sel = w[:, list(rmap)][:, :, list(qmap)]                       # [n][r'][q'][c]
        want[..., :n] = sel.permute(3, 1, 2, 0)
        got = eng.wT[off:off + want.numel()].view_as(want)
        assert torch.equal(got, want.to(got.dtype)), key
        checked += 1
    assert checked >= 20


def test_side_stream_weight_gradients_match_single_stream(monkeypatch):
    """The laned backward plan (weight gradients on a second HIP stream, ring of raw-gradient buffers)
    against the same plan replayed on one stream: identical up to the order of float atomics."""
    ocfg, ecfg = _cfgs('resnet50', 'slots', 'f32', S=64)
    params, image, caption = _data(ocfg, 4, 11)
    grads = []
    for lanes in ('0', '1'):
        monkeypatch.setenv('CAPMI_LANES', lanes)
        eng = _engine(ecfg, params)
        assert eng._compile_train is not None
        eng.forward_backward(torch.as_tensor(image).cuda(), torch.as_tensor(caption).cuda())
        torch.cuda.synchronize()
        if lanes == '1':
            assert eng._train[4]['bwd'].has_lanes
        grads.append(eng.store.grad[:eng.store.trainable_size].double().cpu().numpy().copy())
    scale = np.abs(grads[0]).max()
    assert np.abs(grads[0] - grads[1]).max() <= 1e-5 * scale


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_train_step_with_fused_lstm_steps(dtype, monkeypatch, deterministic):
    """hidden = 256 takes the fused recurrence kernels (capmi_lstm_step_*): f32 against the oracle at the tolerances
    of the two-launch path; both dtypes against the same engine with the fusion switched off (deterministic mode: two default-mode
    runs of ONE bf16 plan already differ by the order of their batch-norm atomics, amplified by the model at random initialisation)."""
    from myimagecaptioningmodel_amd import _lib
    from myimagecaptioningmodel_amd.decoder import DecoderRunner
    monkeypatch.setenv('CAPMI_LSTM_FUSE', '2')          # both directions (the default fuses the forward step only)
    monkeypatch.setenv('CAPMI_LSTM_SEQ', '0')           # (the persistent whole-sequence kernels have their own tests: test_gpu_round2)
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', dtype, S=64, H=256, E=64, V=60, L=5)
    params, image, caption = _data(ocfg, 4, 21)
    eng = _engine(ecfg, params)
    assert _lib.lib().capmi_lstm_step_supported(4, 256, eng.code) == 1
    le = float(eng.forward_backward(image, caption).cpu()[0])
    assert any(n == 'capmi_lstm_step_fwd' for _, n, _ in eng._train[4]['fwd'].launches())
    assert any(n == 'capmi_lstm_step_bwd' for _, n, _ in eng._train[4]['bwd'].launches())
    ge = eng.export_reference_grads()
    monkeypatch.setattr(DecoderRunner, 'fuse_lstm', False)
    ref = _engine(ecfg, params)
    lr = float(ref.forward_backward(image, caption).cpu()[0])
    assert not any(n == 'capmi_lstm_step_fwd' for _, n, _ in ref._train[4]['fwd'].launches())
    gr = ref.export_reference_grads()
    a = np.concatenate([ge[k].ravel() for k in sorted(ge)])
    b = np.concatenate([gr[k].ravel() for k in sorted(gr)])
    tol = 1e-5 if dtype == 'f32' else 2e-2
    assert abs(le - lr) <= tol and np.linalg.norm(a - b) <= (1e-4 if dtype == 'f32' else 5e-2) * np.linalg.norm(b)
    if dtype == 'f32':
        o = om.OracleModel(ocfg, params)
        lo, _ = o.forward_train(image, caption, update_stats=False)
        go = o.backward()
        assert abs(le - lo) <= 1e-4
        num = np.sqrt(sum(np.sum((ge[k] - go[k]) ** 2) for k in go))
        den = np.sqrt(sum(np.sum(go[k] ** 2) for k in go))
        assert num <= 2e-3 * den, num / den


def test_baseline_config0_repo_default_model():
    """BASELINE.json configs[0], the reference's own CPU-runnable case: repo-default encoder + decoder (MobileNetV2,
    hidden 1024, embed 256, singleton attention, f32) on 64x64 images, vocab 1000, seq_len 10, batch 4 -- loss
    within 1e-3 of the oracle, gradients in relative L2, greedy ids bit-exact (north_star tolerances)."""
    ocfg = om.default_cfg(encoder='mobilenetv2', image_size=64, hidden=1024, embed=256, vocab=1000, sentence_length=10,
                          infer_max_length=10, attention='singleton')
    from myimagecaptioningmodel_amd import default_cfg
    ecfg = default_cfg(encoder='mobilenetv2', image_size=64, hidden=1024, embed=256, vocab=1000, sentence_length=10,
                       infer_max_length=10, attention='singleton', dtype='f32', learning_rate=1e-4, batch_size=4)
    params, image, caption = _data(ocfg, 4, 7)
    o = om.OracleModel(ocfg, params)
    lo, _ = o.forward_train(image.astype(np.float64), caption, update_stats=False)
    go = o.backward()
    eng = _engine(ecfg, params)
    le = float(eng.forward_backward(image, caption).cpu()[0])
    assert abs(le - lo) <= 1e-3, (le, lo)
    ge = eng.export_reference_grads()
    # batch 4 at 64x64 puts 16 samples under the last batch norms: calibrate against the f32 NumPy evaluation of the
    # same graph (see test_f32_train_step_matches_oracle)
    _, g32 = _f32_oracle_noise(ocfg, params, image, caption)
    num = np.sqrt(sum(np.sum((ge[k] - go[k]) ** 2) for k in go))
    noise = np.sqrt(sum(np.sum((g32[k].astype(np.float64) - go[k]) ** 2) for k in go))
    den = np.sqrt(sum(np.sum(go[k] ** 2) for k in go))
    print('config[0] gradient relative-L2 error %.2e (f32 NumPy noise floor %.2e)' % (num / den, noise / den))
    assert num <= max(5e-3 * den, 10 * noise), (num / den, noise / den)
    ids_o, logits_o = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()}).greedy_decode(image.astype(np.float64))
    eng2 = _engine(ecfg, params)                 # fresh running statistics, as the oracle's
    ids_e = eng2.decode(image).cpu().numpy()
    top2 = np.sort(logits_o, axis=-1)[..., -2:]
    assert (top2[..., 1] - top2[..., 0]).min() > 1e-3, 'test inputs have a near-tie; change the seed'
    assert ids_e.dtype == np.float32
    np.testing.assert_array_equal(ids_e, ids_o)


def test_resnet101_encoder_small():
    """The deeper build-defined encoder of BASELINE.json configs[3] (ResNet-101) at a size the oracle finishes in
    seconds: loss and all gradients against the oracle."""
    ocfg, ecfg = _cfgs('resnet101', 'slots', 'f32', S=64, H=48, E=24, V=80, L=6)
    params, image, caption = _data(ocfg, 2, 13)
    o = om.OracleModel(ocfg, params)
    lo, _ = o.forward_train(image.astype(np.float64), caption, update_stats=False)
    go = o.backward()
    eng = _engine(ecfg, params)
    le = float(eng.forward_backward(image, caption).cpu()[0])
    assert abs(le - lo) <= 1e-3, (le, lo)
    ge = eng.export_reference_grads()
    _, g32 = _f32_oracle_noise(ocfg, params, image, caption)
    num = np.sqrt(sum(np.sum((ge[k] - go[k]) ** 2) for k in go))
    noise = np.sqrt(sum(np.sum((g32[k].astype(np.float64) - go[k]) ** 2) for k in go))
    den = np.sqrt(sum(np.sum(go[k] ** 2) for k in go))
    print('resnet101 gradient relative-L2 error %.2e (f32 NumPy noise floor %.2e)' % (num / den, noise / den))
    assert num <= max(5e-3 * den, 10 * noise), (num / den, noise / den)      # 104 BN layers over 2 x (2 x 2) pixels


@pytest.mark.parametrize('attention', ['singleton', 'slots'])
def test_beam_search_decode(attention):
    """Beam search (build-defined extension, BASELINE cfg 5) against the oracle's beam_decode: beam = 1 is the greedy
    loop bit for bit; beam = 3 and 5 return the oracle's best hypothesis (ids bit-exact where the oracle has no
    near-tie between the kept and the first dropped candidate) with its score."""
    ocfg, ecfg = _cfgs('mobilenetv2', attention, 'f32', L=7)
    B = 5
    params, image, _ = _data(ocfg, B, seed=17)
    eng = _engine(ecfg, params)
    greedy = eng.decode(image).cpu().numpy().copy()
    eng1 = _engine(ecfg, params)
    np.testing.assert_array_equal(eng1.decode(image, beam=1).cpu().numpy(), greedy)
    for beam in (3, 5):
        ids_o, score_o, gaps = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()}).beam_decode(image.astype(np.float64), beam)
        e = _engine(ecfg, params)
        ids_e = e.decode(image, beam=beam).cpu().numpy()
        score_e = e.decode_scores(B, beam).cpu().numpy()
        assert ids_e.dtype == np.float32 and ids_e.shape == ids_o.shape
        safe = gaps.min(axis=0) > 1e-3               # images whose every step separates kept from dropped candidates
        assert safe.sum() >= B - 1, 'test inputs have near-ties; change the seed'
        np.testing.assert_array_equal(ids_e[safe], ids_o[safe])
        np.testing.assert_allclose(score_e[safe], score_o[safe], rtol=0, atol=2e-3)
        # a wider beam never finds a worse best hypothesis than greedy
        greedy_o, lg = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()}).greedy_decode(image.astype(np.float64))
        lg = lg.astype(np.float64)
        logp = lg - (lg.max(-1, keepdims=True) + np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1, keepdims=True)))
        gscore = np.take_along_axis(logp, greedy_o.astype(np.int64)[..., None], -1)[..., 0].sum(1)
        assert np.all(score_o >= gscore - 1e-9)


def test_inference_mode_decode_uses_running_statistics():
    """is_test decode (the exported inference model of infer.py: batch norm on the running statistics, nothing
    updated) against the oracle, and the host-side caption filter of evaluate.py:14-25."""
    from myimagecaptioningmodel_amd.decode import ids_to_tokens
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', 'f32', L=8)
    B = 4
    params, image, _ = _data(ocfg, B, seed=5)
    rng = np.random.RandomState(1)
    for k in params:                       # running statistics away from their init, so that they matter
        if k.endswith('_bn_mean'):
            params[k] = params[k] + 0.05 * rng.standard_normal(params[k].shape)
        if k.endswith('_bn_variance'):
            params[k] = params[k] * (1.0 + 0.3 * rng.uniform(size=params[k].shape))
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    ids_o, logits_o = oracle.greedy_decode(image.astype(np.float64), is_test=True)
    top2 = np.sort(logits_o, axis=-1)[..., -2:]
    assert (top2[..., 1] - top2[..., 0]).min() > 1e-3, 'test inputs have a near-tie; change the seed'
    eng = _engine(ecfg, params)
    ids_e = eng.decode(image, is_test=True).cpu().numpy()
    np.testing.assert_array_equal(ids_e, ids_o)
    pe = eng.export_reference_params()
    np.testing.assert_array_equal(pe['conv9_bn_mean'], params['conv9_bn_mean'].astype(np.float32))      # nothing updated
    # differs from the batch-statistics graph on the same input (otherwise the test would prove nothing)
    ids_b, _ = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()}).greedy_decode(image.astype(np.float64))
    assert not np.array_equal(ids_b, ids_o)
    assert ids_to_tokens(np.array([5., 0., 7., 3., 9.], np.float32)) == [5, 7]
    assert ids_to_tokens(ids_e[0]) == om.ids_to_tokens(ids_o[0])


def test_device_feeder_on_gpu_feeds_identical_batches():
    """Pinned double-buffered H2D feeder: device batches equal the host data in order (including the short last batch),
    and a train step fed through it gives the loss of feeding the arrays directly."""
    from myimagecaptioningmodel_amd.feeder import DeviceFeeder
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', 'f32')
    params, image, caption = _data(ocfg, 6, 23)
    samples = [(image[i].astype(np.float16), caption[i]) for i in range(6)] * 3      # 18 samples -> batches 4,4,4,4,2

    def batches():
        for i in range(0, len(samples), 4):
            yield samples[i:i + 4]

    eng, ref = _engine(ecfg, params), _engine(ecfg, params)
    n = 0
    for k, (img_d, cap_d) in enumerate(DeviceFeeder(batches(), device='cuda:0', depth=2)):
        host_img = np.stack([s[0] for s in samples[4 * k:4 * k + 4]]).astype(np.float32)
        host_cap = np.stack([s[1] for s in samples[4 * k:4 * k + 4]])
        assert img_d.is_cuda and tuple(img_d.shape) == host_img.shape
        l1 = float(eng.train_step(img_d, cap_d)[0].cpu()[0])
        l2 = float(ref.train_step(host_img, host_cap)[0].cpu()[0])
        # same inputs, same weights: equal up to the f32 atomics' summation order -- which Adam amplifies step by step
        # and batch norm over the 2-image last batch (8 values per channel in the last layers) amplifies again; the
        # feeder's own contract is the exact equality of the fed tensors, asserted below
        assert abs(l1 - l2) <= (1e-6 if k == 0 else 2e-2) * max(1.0, abs(l2)), (k, l1, l2)
        np.testing.assert_array_equal(img_d.cpu().numpy(), host_img)
        n += 1
    assert n == 5
