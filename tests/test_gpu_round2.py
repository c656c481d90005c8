"""Round-2 parity cases on the MI355X (all through the C ABI): the stacked (2-layer) LSTM decoder of
BASELINE configs[3], the frozen-encoder graph (quirk Q4, MobileNetV2.py:27-29 + config.py:50), and the
checkpoint directory round trip of train.py:68-107 through a live engine.

Tolerances as in tests/test_gpu_model.py: f32 engine vs f64 oracle loss <= 1e-3 (north_star), greedy / beam ids
bit-exact, gradients in relative L2 against the f32-NumPy noise floor of the same input.
"""
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from tests.test_gpu_model import _cfgs, _data, _engine, _f32_oracle_noise

pytestmark = pytest.mark.gpu


def _grad_check(grads_e, grads_o, g32, tag):
    gscale = max(np.abs(g).max() for g in grads_o.values())
    tot_err = np.sqrt(sum(np.sum((grads_e[n] - g) ** 2) for n, g in grads_o.items()))
    tot_noise = np.sqrt(sum(np.sum((g32[n].astype(np.float64) - g) ** 2) for n, g in grads_o.items()))
    tot = np.sqrt(sum(np.sum(g ** 2) for g in grads_o.values()))
    print('%s total relative-L2 gradient error %.2e (f32 NumPy noise floor %.2e)' % (tag, tot_err / tot, tot_noise / tot))
    assert tot_err / tot <= max(2e-3, 10 * tot_noise / tot), (tag, tot_err / tot, tot_noise / tot)
    for name, go in grads_o.items():
        floor = 1e-6 * gscale * np.sqrt(go.size)
        nrm, err = np.linalg.norm(go), np.linalg.norm(grads_e[name] - go)
        noise = np.linalg.norm(g32[name].astype(np.float64) - go)
        assert err <= max(5e-2 * nrm, 20 * noise, floor), (tag, name, err / (nrm + 1e-30), noise / (nrm + 1e-30))


# ------------------------------------------------------------------ row X2: stacked LSTM decoder
@pytest.mark.parametrize('attention', ['singleton', 'slots'])
def test_two_layer_lstm_train_step_and_decode_match_oracle(attention):
    """rnn_layer = 2 (build-defined: layer 1 = lstm_unit on layer 0's new hidden state; sentinel / attention / output
    head read the top layer): loss, logits, every gradient incl. lstm_w_l1 / lstm_b_l1, Adam, greedy and beam ids."""
    ocfg, ecfg = _cfgs('mobilenetv2', attention, 'f32', S=64, L=8)
    ocfg['rnn_layer'] = ecfg['rnn_layer'] = 2
    B = 5
    params, image, caption = _data(ocfg, B, seed={'singleton': 12, 'slots': 13}[attention])     # seeds without near-ties in the oracle's decode
    assert 'lstm_w_l1' in params
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    eng = _engine(ecfg, params)
    l32, g32 = _f32_oracle_noise(ocfg, params, image, caption)
    loss_o, logits_o = oracle.forward_train(image.astype(np.float64), caption, update_stats=False)
    grads_o = oracle.backward()
    loss_e = float(eng.forward_backward(image, caption).cpu()[0])
    assert abs(loss_e - loss_o) <= 1e-3 and abs(loss_e - loss_o) <= max(1e-4, 5 * abs(l32 - loss_o)), (loss_e, loss_o, l32)
    dec = eng._train[B]['dec']
    lg = dec.logits.cpu().numpy()[:, :ocfg['vocab']].reshape(dec.T, B, -1).transpose(1, 0, 2)
    assert np.abs(lg - logits_o).max() <= 2e-3 * max(1.0, np.abs(logits_o).max())
    grads_e = eng.export_reference_grads()
    _grad_check(grads_e, grads_o, g32, '2-layer ' + attention)
    for n in ('lstm_w_l1', 'lstm_b_l1', 'lstm_w', 'lstm_b'):
        assert np.abs(grads_e[n]).max() > 0
        assert np.linalg.norm(grads_e[n] - grads_o[n]) <= 2e-3 * np.linalg.norm(grads_o[n]) + 20 * np.linalg.norm(g32[n] - grads_o[n]), n
    # the fused train step (Adam inside the backward plan) moves the new parameters like the oracle's Adam
    eng2 = _engine(ecfg, params)
    eng2.train_step(image, caption)
    oracle2 = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    oracle2.forward_train(image.astype(np.float64), caption)
    oracle2.adam_step(oracle2.backward(), lr=1e-4)
    pe = eng2.export_reference_params()
    for n in ('lstm_w_l1', 'lstm_b_l1', 'lstm_w', 'fc_7.w_0'):
        # |update| = lr exactly where |g| >> eps; the sign is what can differ at f32-noise gradients: compare in L2
        du_e, du_o = pe[n] - params[n], oracle2.p[n] - params[n]
        assert np.linalg.norm(du_e - du_o) <= 0.05 * np.linalg.norm(du_o), n
    # eval graph: greedy ids bit-exact (Q2 float32, Q5 no early stop), beam = 1 == greedy, beam = 3 == the oracle's beam
    o = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    ids_o, lo = o.greedy_decode(image.astype(np.float64))
    top2 = np.sort(lo, axis=-1)[..., -2:]
    assert (top2[..., 1] - top2[..., 0]).min() > 1e-3, 'test inputs have a near-tie; change the seed'
    ids_e = _engine(ecfg, params).decode(image).cpu().numpy()
    assert ids_e.dtype == np.float32
    np.testing.assert_array_equal(ids_e, ids_o)
    np.testing.assert_array_equal(_engine(ecfg, params).decode(image, beam=1).cpu().numpy(), ids_o)
    ids_b, score_b, gaps = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()}).beam_decode(image.astype(np.float64), 3)
    e3 = _engine(ecfg, params)
    ids_e3 = e3.decode(image, beam=3).cpu().numpy()
    safe = gaps.min(axis=0) > 1e-3
    assert safe.sum() >= B - 1, 'test inputs have near-ties; change the seed'
    np.testing.assert_array_equal(ids_e3[safe], ids_b[safe])
    np.testing.assert_allclose(e3.decode_scores(B, 3).cpu().numpy()[safe], score_b[safe], rtol=0, atol=2e-3)


def test_two_layer_lstm_bf16_fused_recurrence():
    """H = 256, B = 16: the one-launch-per-step recurrence (capmi_lstm_step_fwd) carries both layers in bf16."""
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', 'bf16', S=64, H=256, E=64, V=120, L=7)
    ocfg['rnn_layer'] = ecfg['rnn_layer'] = 2
    B = 16
    params, image, caption = _data(ocfg, B, seed=6)
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    loss_o, _ = oracle.forward_train(image.astype(np.float64), caption)
    grads_o = oracle.backward()
    eng = _engine(ecfg, params)
    from myimagecaptioningmodel_amd import _lib
    assert _lib.lib().capmi_lstm_step_supported(B, 256, _lib.BF16)
    loss_e = float(eng.forward_backward(image, caption).cpu()[0])
    assert abs(loss_e - loss_o) <= 5e-2, (loss_e, loss_o)
    ge = eng.export_reference_grads()
    for name in ('lstm_w', 'lstm_w_l1', 'lstm_b_l1', 'word_embedding', 'fc_7.w_0'):
        a, b = ge[name].ravel(), grads_o[name].ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        print('bf16 2-layer grad cosine %s %.4f' % (name, cos))
        assert cos >= 0.97, (name, cos)


# ------------------------------------------------------------------ frozen encoder (quirk Q4)
@pytest.mark.parametrize('encoder,S', [('mobilenetv2', 96), ('resnet50', 64)])
def test_frozen_encoder_on_gpu(encoder, S):
    """encoder_trainable = False (MobileNetV2.py:27-29: only ParamAttr(trainable=False) changes): the encoder still
    normalises with BATCH statistics and still updates its running statistics, its parameters are bit-unchanged by a
    train step, the loss and the decoder gradients equal the trainable run's, and Adam moves the decoder only."""
    ocfg, ecfg = _cfgs(encoder, 'slots', 'f32', S=S)
    B = 6
    params, image, caption = _data(ocfg, B, seed=8)
    e_train = _engine(dict(ecfg, encoder_trainable=True), params)
    e_froz = _engine(dict(ecfg, encoder_trainable=False), params)
    assert e_froz.store.trainable_size == e_froz.store.decoder_size < e_froz.store.size
    lt = float(e_train.forward_backward(image, caption).cpu()[0])
    lf = float(e_froz.forward_backward(image, caption).cpu()[0])
    assert abs(lt - lf) <= 1e-6                                        # same forward launches
    gt, gf = e_train.export_reference_grads(), e_froz.export_reference_grads()
    ocfg_f = dict(ocfg, encoder_trainable=False)
    oracle = om.OracleModel(ocfg_f, {k: v.copy() for k, v in params.items()})
    loss_o, _ = oracle.forward_train(image.astype(np.float64), caption)
    grads_o = oracle.backward()
    assert abs(lf - loss_o) <= 1e-3
    enc_names = [n for n in gf if not om.is_trainable(n, ocfg_f)]
    assert enc_names and all(np.all(gf[n] == 0) for n in enc_names)    # no encoder backward was launched
    gscale = max(np.abs(g).max() for g in grads_o.values())
    for n in gf:
        if n in enc_names:
            continue
        # decoder gradients: the two engines run the same decoder launches (atomics order aside) and match the oracle
        floor = 1e-6 * gscale * np.sqrt(gf[n].size)
        assert np.linalg.norm(gf[n] - gt[n]) <= 1e-4 * np.linalg.norm(gt[n]) + floor, n
        assert np.linalg.norm(gf[n] - grads_o[n]) <= 2e-3 * np.linalg.norm(grads_o[n]) + floor, n
    # one full train step: encoder parameters bit-unchanged, running statistics updated (Q4), decoder moved
    e2 = _engine(dict(ecfg, encoder_trainable=False), params)
    before = e2.export_reference_params()
    loss, lr = e2.train_step(image, caption)
    torch.cuda.synchronize()
    after = e2.export_reference_params()
    first_conv = 'conv1_1' if encoder == 'mobilenetv2' else 'res_conv1'
    for n in after:
        if n.endswith('_bn_mean') or n.endswith('_bn_variance'):
            continue
        if not om.is_trainable(n, ocfg_f):
            assert np.array_equal(before[n], after[n]), n
    assert not np.array_equal(before[first_conv + '_bn_mean'], after[first_conv + '_bn_mean'])
    assert not np.array_equal(before['lstm_w'], after['lstm_w'])
    oracle.adam_step(grads_o, lr=1e-4)
    du_e, du_o = after['fc_11.w_0'] - before['fc_11.w_0'], oracle.p['fc_11.w_0'] - params['fc_11.w_0']
    assert np.linalg.norm(du_e - du_o) <= 0.05 * np.linalg.norm(du_o)
    # and the weight shadows of the (frozen) encoder are still valid for the next step
    l2 = float(e2.train_step(image, caption)[0].cpu()[0])
    assert np.isfinite(l2) and l2 < float(loss.cpu()[0]) + 1e-3


# ------------------------------------------------------------------ §8(f1): checkpoint directory through a live engine
@pytest.mark.parametrize('strategy,dtype', [(None, 'f32'), ('cosine_decay_restart_warmup', 'f32'), ('cosine_decay_warmup', 'bf16')])
def test_checkpoint_resume_equals_uninterrupted_run(tmp_path, strategy, dtype, deterministic):
    """train 2 steps -> save_persistables -> fresh engine -> load_persistables -> step 3: same loss, parameters, Adam
    moments, running statistics, step counter and learning rate as the uninterrupted run (train.py:68-107) -- BIT FOR BIT
    (deterministic mode: no f32 atomics, so the two engines' step 3 cannot differ by summation order)."""
    from myimagecaptioningmodel_amd import ckpt
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', dtype, S=64)
    ecfg.update(lr_decay_strategy=strategy, decay_epoch=2, warmup_epoch=1, max_epoch=4, sample_count=8, batch_size=4, learning_rate=1e-3)
    B = 4
    params, image, caption = _data(ocfg, B, seed=21)
    batches = [_data(ocfg, B, seed=30 + i)[1:] for i in range(3)]
    from myimagecaptioningmodel_amd.optim import LRSchedule
    sched = LRSchedule(strategy, 1e-3, 8, 4, decay_epoch=2, warmup_epoch=1, max_epoch=4)
    b = _engine(ecfg, params)
    lrs = [b.train_step(img, cap)[1] for img, cap in batches[:2]]
    assert lrs == [sched.value(0), sched.value(1)]
    d = str(tmp_path / 'checkpoint')
    ckpt.save_persistables(b, d)
    assert int(ckpt.read_lod_tensor(os.path.join(d, '@LR_DECAY_COUNTER@'))[0]) == b.lr_schedule.counter_after(2)
    c = _engine(ecfg, om.init_params(ocfg, seed=99, dtype=np.float64))          # different weights: everything must come from disk
    ckpt.load_persistables(c, d)
    assert c.step_count == 2
    before = c.export_reference_params()
    pb0 = b.export_reference_params()
    assert all(np.array_equal(before[n], pb0[n]) for n in before)               # f32 masters + running statistics: bit-exact
    for buf in ('adam_m', 'adam_v'):
        xb, xc = b.store.export_reference(getattr(b.store, buf)), c.store.export_reference(getattr(c.store, buf))
        assert all(np.array_equal(xb[n], xc[n]) for n in xb), buf
    # step 3 on both: the engine that never stopped and the resumed one start from identical bits
    loss_b, lr_b = b.train_step(*batches[2])
    loss_c, lr_c = c.train_step(*batches[2])
    torch.cuda.synchronize()
    assert lr_c == lr_b == sched.value(2) and (strategy is None or lr_b != lrs[0])
    assert float(loss_c.cpu()[0]) == float(loss_b.cpu()[0])
    gb, gc = b.export_reference_grads(), c.export_reference_grads()
    for n in gb:
        np.testing.assert_array_equal(gb[n], gc[n], err_msg='gradient ' + n)
    pb, pc = b.export_reference_params(), c.export_reference_params()
    for n in pb:
        np.testing.assert_array_equal(pb[n], pc[n], err_msg=n)
    for buf in ('adam_m', 'adam_v'):
        xb, xc = b.store.export_reference(getattr(b.store, buf)), c.store.export_reference(getattr(c.store, buf))
        for n in xb:
            np.testing.assert_array_equal(xb[n], xc[n], err_msg='%s %s' % (buf, n))
    assert b.step_count == c.step_count == 3


def test_train_loop_crash_in_epoch_resumes_from_last_checkpoint(tmp_path, deterministic):
    """train.py:133-134,172 + logger.py:42: the epoch is written at the START of an epoch and the checkpoint at its END, so
    a crash inside epoch 2 restarts epoch 2 from the end-of-epoch-1 state; a crash inside epoch 1 starts from scratch."""
    from myimagecaptioningmodel_amd import ckpt, train_loop
    ocfg, ecfg = _cfgs('mobilenetv2', 'singleton', 'f32', S=64)
    ecfg.update(lr_decay_strategy='cosine_decay', decay_epoch=3, sample_count=8, batch_size=4, learning_rate=1e-3)
    params, _, _ = _data(ocfg, 4, seed=1)
    data = {ep: [dict(zip(('image', 'caption'), _data(ocfg, 4, seed=100 * ep + i)[1:])) for i in range(2)] for ep in (1, 2, 3)}
    cp, lp = str(tmp_path / 'ckpt'), str(tmp_path / 'log')

    class Crash(RuntimeError):
        pass

    def batches(crash_at):
        def gen(epoch):
            for i, d in enumerate(data[epoch]):
                if (epoch, i) == crash_at:
                    raise Crash()
                yield d
        return gen
    # uninterrupted: 3 epochs
    ref = _engine(ecfg, params)
    train_loop.train(ref, batches(None), 3, str(tmp_path / 'ckpt_ref'), str(tmp_path / 'log_ref'), log_every_n_step=1)
    # crash inside epoch 1 -> restart from scratch (is_first_init stays True: epoch == 1)
    e1 = _engine(ecfg, params)
    with pytest.raises(Crash):
        train_loop.train(e1, batches((1, 1)), 3, cp, lp)
    assert ckpt.load_resume_state(lp)['epoch'] == 1 and not os.path.exists(os.path.join(cp, 'checkpoint'))
    # second process: crashes inside epoch 2 (after one step of it)
    e2 = _engine(ecfg, params)
    with pytest.raises(Crash):
        train_loop.train(e2, batches((2, 1)), 3, cp, lp)
    assert ckpt.load_resume_state(lp)['epoch'] == 2 and os.path.isfile(os.path.join(cp, 'checkpoint', 'lstm_w'))
    # third process: resumes epoch 2 from the end-of-epoch-1 checkpoint and finishes
    e3 = _engine(ecfg, om.init_params(ocfg, seed=5, dtype=np.float64))
    conf = train_loop.train(e3, batches(None), 3, cp, lp, checkpoint_backup_every_n_epoch=3, export_params=True)
    assert conf['epoch'] == 3 and e3.step_count == ref.step_count == 6
    assert os.path.isfile(os.path.join(cp, 'checkpoint3', 'lstm_w')) and os.path.isfile(os.path.join(cp, 'params', 'lstm_w'))
    assert not os.path.exists(os.path.join(cp, 'params', 'lstm_w_moment1_0'))
    # six Adam steps in three "processes" against six in one: in deterministic mode (no f32 atomics) every step is a
    # pure function of the checkpointed state, so the resumed run lands on the uninterrupted run's bits
    pr, p3 = ref.export_reference_params(), e3.export_reference_params()
    p0 = {k: np.asarray(v, np.float32) for k, v in params.items()}
    assert any(np.abs(pr[n] - p0[n]).max() > 0 for n in pr if n in ref.store.entries)
    for n in pr:
        np.testing.assert_array_equal(pr[n], p3[n], err_msg=n)
    assert ref.lr_schedule.value(ref.step_count) == e3.lr_schedule.value(e3.step_count)
    log = open(os.path.join(lp, 'log.txt')).read()
    assert log.count('Epoch 2') == 2 and 'Epoch loss' in log


# ------------------------------------------------------------------ C-side plan runner (capmi_plan_run)
@pytest.mark.parametrize('encoder,dtype', [('mobilenetv2', 'f32'), ('resnet50', 'bf16')])
def test_plan_runner_equals_the_per_launch_host_walk(encoder, dtype, monkeypatch, deterministic):
    """One capmi_plan_run call per plan against the round-1 host path (one foreign call per launch, CAPMI_PY_PLAN=1), both
    on two lanes and in the single-stream order: losses, logits, batch / running statistics, EVERY gradient and every
    updated parameter are bit-identical -- the same entry points get the same arguments in the same order, and in
    deterministic mode (no f32 atomics) no result depends on how the two lanes interleave."""
    ocfg, ecfg = _cfgs(encoder, 'slots', dtype, S=96 if encoder == 'mobilenetv2' else 64)
    ecfg['learning_rate'] = 1e-3
    B = 6
    params, image, caption = _data(ocfg, B, seed=2)
    out = {}
    for mode in ('c', 'py', 'c1', 'py1'):
        monkeypatch.setenv('CAPMI_PY_PLAN', '1' if mode.startswith('py') else '0')
        monkeypatch.setenv('CAPMI_LANES', '0' if mode.endswith('1') else '1')
        eng = _engine(ecfg, params)
        ids = eng.decode(image, is_test=True).cpu().numpy().copy()   # a single-lane plan on the untouched parameters (nothing updated)
        losses = []
        for _ in range(2):
            loss, lr = eng.train_step(image, caption)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        prog = eng._train[B]
        out[mode] = dict(loss=[float(l.cpu()[0]) for l in losses], logits=prog['dec'].logits.clone(),
                         stats={k: v['mean'].clone() for k, v in prog['enc'].bn.items()},
                         params=eng.export_reference_params(), grads=eng.export_reference_grads(), lr=lr)
        out[mode]['ids'] = ids
    ref = out['py']
    assert ref['loss'][1] < ref['loss'][0]
    for mode in ('c', 'c1', 'py1'):
        o = out[mode]
        assert o['lr'] == ref['lr'] and o['loss'] == ref['loss'], (mode, o['loss'], ref['loss'])
        np.testing.assert_array_equal(o['ids'], ref['ids'])
        for n, g in ref['grads'].items():
            np.testing.assert_array_equal(o['grads'][n], g, err_msg='%s gradient %s' % (mode, n))
        for n, v in ref['params'].items():
            np.testing.assert_array_equal(o['params'][n], v, err_msg='%s %s' % (mode, n))
    # first step's forward pass, same lane mode: bit for bit (re-run one step on fresh engines and compare device buffers)
    fw = {}
    for mode in ('c', 'py'):
        monkeypatch.setenv('CAPMI_PY_PLAN', '1' if mode == 'py' else '0')
        monkeypatch.setenv('CAPMI_LANES', '1')
        eng = _engine(ecfg, params)
        eng.forward_backward(image, caption)
        torch.cuda.synchronize()
        prog = eng._train[B]
        fw[mode] = (prog['dec'].logits.clone(), [v['mean'].clone() for v in prog['enc'].bn.values()], [v['invstd'].clone() for v in prog['enc'].bn.values()],
                    prog['enc'].out_tensor().clone())
    assert torch.equal(fw['c'][0], fw['py'][0]) and torch.equal(fw['c'][3], fw['py'][3])
    assert all(torch.equal(a, b) for a, b in zip(fw['c'][1], fw['py'][1])) and all(torch.equal(a, b) for a, b in zip(fw['c'][2], fw['py'][2]))


# ------------------------------------------------------------------ persistent LSTM recurrence inside the engine
@pytest.mark.parametrize('layers', [1, 2])
def test_engine_with_persistent_recurrence_equals_per_step_plan(layers, monkeypatch, deterministic):
    """H = 256, B = 16 bf16: the train step whose LSTM layers each run as ONE forward and ONE backward launch
    (capmi_lstm_seq_*, grid barriers) against the same step with per-step launches (CAPMI_LSTM_SEQ=0, fused backward
    steps CAPMI_LSTM_FUSE=2 -- the arithmetic the persistent kernels mirror): forward bit-identical, gradients equal up
    to the FMA contraction in the cell (deterministic mode: no atomic-order noise on top); and against the oracle within the
    bf16 bounds of test_gpu_model."""
    ocfg, ecfg = _cfgs('mobilenetv2', 'slots', 'bf16', S=64, H=256, E=64, V=120, L=7)
    ocfg['rnn_layer'] = ecfg['rnn_layer'] = layers
    B = 16
    params, image, caption = _data(ocfg, B, seed=6)
    res = {}
    for mode, env in (('seq', {'CAPMI_LSTM_SEQ': '1'}), ('steps', {'CAPMI_LSTM_SEQ': '0', 'CAPMI_LSTM_FUSE': '2'})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = _engine(ecfg, params)
        loss = float(eng.forward_backward(image, caption).cpu()[0])
        dec = eng._train[B]['dec']
        assert dec.use_seq == (mode == 'seq')
        eng.check_sync()
        res[mode] = dict(loss=loss, logits=dec.logits.clone(), H=[h.clone() for h in dec.Hbufs], G=[g.clone() for g in dec.Gs],
                         dG=[g.clone() for g in dec.dGs], grads=eng.export_reference_grads(), n=len(eng._train[B]['fwd']) + len(eng._train[B]['bwd']))
        monkeypatch.delenv('CAPMI_LSTM_FUSE', raising=False)
    a, b = res['seq'], res['steps']
    assert a['n'] < b['n'] - 2 * layers * (ocfg['sentence_length'] - 3)            # the per-step launches are gone
    assert a['loss'] == b['loss'] and torch.equal(a['logits'], b['logits'])
    assert all(torch.equal(x, y) for x, y in zip(a['H'], b['H'])) and all(torch.equal(x, y) for x, y in zip(a['G'], b['G']))
    assert all(torch.allclose(x.float(), y.float(), rtol=8e-3, atol=1e-3) for x, y in zip(a['dG'], b['dG']))     # up to FMA contraction in the cell
    for n, g in b['grads'].items():
        assert np.linalg.norm(a['grads'][n] - g) <= 1e-3 * np.linalg.norm(g) + 1e-7, n
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    loss_o, _ = oracle.forward_train(image.astype(np.float64), caption)
    grads_o = oracle.backward()
    assert abs(a['loss'] - loss_o) <= 5e-2
    for name in ('lstm_w', 'lstm_b', 'word_embedding', 'fc_7.w_0') + (('lstm_w_l1',) if layers == 2 else ()):
        x, y = a['grads'][name].ravel(), grads_o[name].ravel()
        assert float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30)) >= 0.97, name


# ------------------------------------------------------------------ batch norm in the consumer's operand path, inside the engine
def test_relu_masks_as_bits_in_the_engine_are_bit_identical(monkeypatch, deterministic):
    """CAPMI_MASKBITS=1 (default: capmi_bn_apply_mask writes one bit per element of every ReLU tensor in the forward pass, the data
    gradient that masks the tensor's gradient reads the bits in front of its main loop -- igemm.hip EPI 6) against CAPMI_MASKBITS=0
    (the epilogue reads the saved output itself): the same loss, every activation gradient and every parameter gradient BIT FOR BIT
    (deterministic mode), and the bits equal y > 0 of the stored tensors (conv2d_grad under relu, MobileNetV2.py:112-121)."""
    monkeypatch.setenv('CAPMI_BNSUM', '0')        # (the sums epilogue needs the mask bits: with it the two plans would add their batch-norm sums in different orders)
    ocfg, ecfg = _cfgs('resnet50', 'slots', 'bf16', S=128)
    B = 8
    params, image, caption = _data(ocfg, B, seed=4)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('CAPMI_MASKBITS', mode)
        eng = _engine(ecfg, params)
        loss = float(eng.forward_backward(image, caption).cpu()[0])
        torch.cuda.synchronize()
        prog = eng._train[B]
        enc = prog['enc']
        out[mode] = dict(loss=loss, grads=eng.export_reference_grads(), g={k: v.clone() for k, v in enc.grad.items()},
                         act={k: v.clone() for k, v in enc.act.items()}, bits={k: v.clone() for k, v in enc.maskbits.items()},
                         calls=['capmi_bn_apply_mask' if (c[1] == 'capmi_bn_stat_apply' and c[2][19]) else c[1]      # (capmi_bn_stat_apply with a mask
                                for c in prog['fwd'].calls if c[0] is not None])                                      #  pointer = finalize + capmi_bn_apply_mask)
    a, b = out['0'], out['1']
    assert len(a['bits']) == 0 and len(b['bits']) >= 40 and b['calls'].count('capmi_bn_apply_mask') == len(b['bits'])
    assert a['loss'] == b['loss']
    for k in a['g']:
        assert torch.equal(a['g'][k], b['g'][k]), ('activation gradient', k)
    for n, g in a['grads'].items():
        np.testing.assert_array_equal(b['grads'][n], g, err_msg=n)
    for k, bits in b['bits'].items():
        y = b['act'][k].float().reshape(-1, 8)
        want = ((y > 0).to(torch.int32) << torch.arange(8, device=y.device, dtype=torch.int32)).sum(-1).to(torch.uint8)
        assert torch.equal(want, bits), ('mask bits of tensor', k)


@pytest.mark.parametrize('level', [1, 2])
def test_operand_path_batch_norm_in_the_engine_is_bit_identical(level, monkeypatch, deterministic):
    """CAPMI_INBN=1 / 2 (capmi_igemm_nt_bnact on the 1x1 / also the halo-staged 3x3 consumers, the producers' bn_apply moved
    to the side lane) against the default plan: the same bits everywhere -- loss, logits, every conv output and activated
    tensor after the step, every gradient (deterministic mode) -- on a ResNet-50 whose res2 / res3 layers are large enough
    for both kernel families (MobileNetV2.py:88-121: the unit chain this fuses)."""
    monkeypatch.setenv('CAPMI_BNSUM', '0')        # (the sums epilogue needs the mask bits: with it the two plans would add their batch-norm sums in different orders)
    ocfg, ecfg = _cfgs('resnet50', 'slots', 'bf16', S=128)
    B = 8
    params, image, caption = _data(ocfg, B, seed=4)
    out = {}
    for mode in (0, level):
        monkeypatch.setenv('CAPMI_INBN', str(mode))
        eng = _engine(ecfg, params)
        loss = float(eng.forward_backward(image, caption).cpu()[0])
        torch.cuda.synchronize()
        prog = eng._train[B]
        enc = prog['enc']
        out[mode] = dict(loss=loss, logits=prog['dec'].logits.clone(), raw={k: v.clone() for k, v in enc.raw.items()},
                         act={k: v.clone() for k, v in enc.act.items()}, grads=eng.export_reference_grads(), n_fused=len(enc.inbn),
                         calls=['capmi_bn_apply_mask' if (c[1] == 'capmi_bn_stat_apply' and c[2][19]) else c[1]      # (capmi_bn_stat_apply with a mask
                                for c in prog['fwd'].calls if c[0] is not None])                                      #  pointer = finalize + capmi_bn_apply_mask)
    a, b = out[0], out[level]
    assert a['n_fused'] == 0 and b['n_fused'] >= (10 if level == 1 else 16), b['n_fused']
    assert b['calls'].count('capmi_igemm_nt_bnact') == b['n_fused'] and a['calls'].count('capmi_igemm_nt_bnact') == 0
    assert a['loss'] == b['loss'] and torch.equal(a['logits'], b['logits'])
    for k in a['raw']:
        assert torch.equal(a['raw'][k], b['raw'][k]), ('conv output', k)
    for k in a['act']:
        assert torch.equal(a['act'][k], b['act'][k]), ('activated tensor', k)
    for n, g in a['grads'].items():
        np.testing.assert_array_equal(b['grads'][n], g, err_msg=n)
