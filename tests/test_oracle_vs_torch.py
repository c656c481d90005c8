"""Pins the NumPy oracle (PARITY UNPINNED by the reference -- it has no tests): fp64 agreement
with an independent torch.autograd build of the same graph, finite differences, and the
analytic known-answer tests listed in SURVEY.md section 8(c)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import ops
from tests.conftest import make_caption
from tests import torch_ref


def _setup(cfg, B=3, seed=0):
    rng = np.random.RandomState(seed)
    params = om.init_params(cfg, seed=seed, dtype=np.float64)
    # perturb BN affine + biases so that their gradients are exercised away from the init point
    for k in params:
        if k.endswith('_bn_scale') or k.endswith('_bn_offset') or k.endswith('.b_0') or k in ('lstm_b', 'out_fc_bias'):
            params[k] = params[k] + 0.1 * rng.standard_normal(params[k].shape)
    S = cfg['image_size']
    image = rng.uniform(0, 1, (B, 3, S, S))
    caption = make_caption(rng, B, cfg['sentence_length'], cfg['vocab'])
    return params, image, caption


@pytest.mark.parametrize('attention', ['singleton', 'slots'])
@pytest.mark.parametrize('encoder', ['mobilenetv2', 'resnet50'])
def test_oracle_matches_torch_autograd(tiny_cfg, attention, encoder):
    cfg = dict(tiny_cfg, attention=attention, encoder=encoder)
    params, image, caption = _setup(cfg)
    m = om.OracleModel(cfg, {k: v.copy() for k, v in params.items()})
    loss, logits = m.forward_train(image, caption)
    grads = m.backward()

    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=om.is_trainable(k, cfg)) for k, v in params.items()}
    tl, tlogits = torch_ref.forward_loss(cfg, tp, torch.tensor(image), torch.tensor(caption))
    tl.backward()
    assert abs(float(tl.detach()) - float(loss)) <= 1e-10 * max(1.0, abs(float(tl.detach())))
    np.testing.assert_allclose(logits, tlogits.detach().numpy(), rtol=1e-9, atol=1e-9)
    for k, g in grads.items():
        tg = tp[k].grad
        tg = np.zeros_like(g) if tg is None else tg.numpy()
        scale = max(1e-12, np.abs(tg).max())
        assert np.abs(g - tg).max() <= 1e-8 * scale + 1e-12, k


@pytest.mark.parametrize('attention', ['singleton', 'slots'])
def test_two_layer_lstm_oracle_matches_torch_autograd(tiny_cfg, attention):
    """BASELINE configs[3]'s stacked decoder (build-defined: the reference stores rnn_layer and never reads it)."""
    cfg = dict(tiny_cfg, attention=attention, rnn_layer=2)
    params, image, caption = _setup(cfg, seed=4)
    assert params['lstm_w_l1'].shape == (2 * cfg['hidden'], 4 * cfg['hidden'])
    m = om.OracleModel(cfg, {k: v.copy() for k, v in params.items()})
    loss, logits = m.forward_train(image, caption)
    grads = m.backward()
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=om.is_trainable(k, cfg)) for k, v in params.items()}
    tl, tlogits = torch_ref.forward_loss(cfg, tp, torch.tensor(image), torch.tensor(caption))
    tl.backward()
    assert abs(float(tl.detach()) - float(loss)) <= 1e-10 * max(1.0, abs(float(tl.detach())))
    np.testing.assert_allclose(logits, tlogits.detach().numpy(), rtol=1e-9, atol=1e-9)
    for k, g in grads.items():
        tg = tp[k].grad
        tg = np.zeros_like(g) if tg is None else tg.numpy()
        assert np.abs(g - tg).max() <= 1e-8 * max(1e-12, np.abs(tg).max()) + 1e-12, k
    assert np.abs(grads['lstm_w_l1']).max() > 0 and np.abs(grads['lstm_b_l1']).max() > 0
    # the stacked decode loops: beam = 1 is the greedy loop, and one layer of the stack with an identity-free check:
    ids, _ = m.greedy_decode(image, update_stats=False)
    ids1, score, _ = m.beam_decode(image, 1, update_stats=False)
    assert np.array_equal(ids, ids1) and ids.shape == (image.shape[0], cfg['infer_max_length'])
    ids3, score3, _ = m.beam_decode(image, 3, update_stats=False)
    assert np.all(score3 >= score - 1e-12)


def test_finite_differences(tiny_cfg):
    cfg = dict(tiny_cfg, attention='slots')
    params, image, caption = _setup(cfg, B=2, seed=1)
    m = om.OracleModel(cfg, {k: v.copy() for k, v in params.items()})
    m.forward_train(image, caption, update_stats=False)
    grads = m.backward()
    rng = np.random.RandomState(5)
    for name in ['lstm_w', 'word_embedding', 'fc_10.w_0', 'fc_3.w_0', 'conv9_weights', 'conv4_2_dwise_weights',
                 'conv1_1_bn_scale', 'fc_0.b_0']:
        for _ in range(2):
            idx = tuple(rng.randint(0, s) for s in params[name].shape)
            eps = 1e-6
            vals = []
            for sgn in (+1, -1):
                pp = {k: v.copy() for k, v in params.items()}
                pp[name][idx] += sgn * eps
                vals.append(om.OracleModel(cfg, pp).forward_train(image, caption, update_stats=False)[0])
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - grads[name][idx]) <= 1e-5 * max(1.0, abs(fd)) + 1e-7, (name, idx, fd, grads[name][idx])


def test_q1_singleton_alpha_is_one_and_dead_params_get_zero_grad(tiny_cfg):
    """Quirk Q1: softmax over a size-1 axis -> alpha == 1 and fc_3/fc_8/fc_9/fc_10 get exactly 0."""
    cfg = dict(tiny_cfg, attention='singleton')
    params, image, caption = _setup(cfg)
    m = om.OracleModel(cfg, params)
    m.forward_train(image, caption)
    for c in m._saved['steps']:
        assert np.all(c['alpha'] == 1.0)
    g = m.backward()
    for name in ('fc_3', 'fc_8', 'fc_9', 'fc_10'):
        assert np.all(g[name + '.w_0'] == 0) and np.all(g[name + '.b_0'] == 0)
    assert np.abs(g['fc_2.w_0']).max() > 0


def test_zero_embedding_gives_ln_v_and_greedy_emits_zero(tiny_cfg):
    cfg = tiny_cfg
    params, image, caption = _setup(cfg)
    params['word_embedding'][:] = 0
    params['out_fc_bias'][:] = 0
    m = om.OracleModel(cfg, params)
    loss, _ = m.forward_train(image, caption)
    assert abs(loss - np.log(cfg['vocab'])) < 1e-12
    ids, _ = m.greedy_decode(image)
    assert ids.dtype == np.float32 and ids.shape == (3, cfg['infer_max_length'])   # quirk Q2
    assert np.all(ids == 0)                                                         # lowest-index tie


def test_mask_normalisation_hand_count(tiny_cfg):
    cfg = tiny_cfg
    params, image, caption = _setup(cfg)
    m = om.OracleModel(cfg, params)
    loss, logits = m.forward_train(image, caption)
    tot, cnt = 0.0, 0
    for b in range(caption.shape[0]):
        for s in range(cfg['sentence_length'] - 1):
            tgt = caption[b, s + 1]
            if tgt == 0:
                continue
            row = logits[b, s]
            tot += np.log(np.exp(row - row.max()).sum()) + row.max() - row[tgt]
            cnt += 1
    assert abs(loss - tot / cnt) < 1e-12
    assert logits.shape[1] == cfg['sentence_length'] - 1


def test_lstm_zero_weights_known_answer():
    H = 4
    x = np.ones((2, 3)); h = np.zeros((2, H)); c = np.full((2, H), 0.8)
    w = np.zeros((3 + H, 4 * H)); b = np.zeros(4 * H)
    h1, c1, _ = ops.lstm_unit_fwd(x, h, c, w, b)
    np.testing.assert_allclose(c1, 0.5 * c)
    np.testing.assert_allclose(h1, 0.5 * np.tanh(c1))


def test_bn_constant_channel_outputs_offset():
    x = np.full((2, 3, 4, 4), 2.5)
    y, _, _ = ops.batch_norm_fwd(x, np.array([1., 2., 3.]), np.array([.1, .2, .3]), np.zeros(3), np.ones(3))
    np.testing.assert_allclose(y, np.broadcast_to(np.array([.1, .2, .3])[None, :, None, None], x.shape), atol=1e-12)


def test_context_of_identical_slots_is_that_slot(tiny_cfg):
    cfg = dict(tiny_cfg, attention='slots')
    params, image, caption = _setup(cfg)
    m = om.OracleModel(cfg, params)
    B, K, H = 2, 4, cfg['hidden']
    v = np.random.RandomState(0).standard_normal((B, 1, H))
    # alpha sums to 1 over slots and context uses reduce_mean -> context = slot / (K+1) * 1
    alpha = m._alpha(np.random.RandomState(1).standard_normal((B, K + 1, 1)))
    np.testing.assert_allclose(alpha.sum(1), 1.0)
    ctx = (np.repeat(v, K + 1, 1) * alpha).mean(1)
    np.testing.assert_allclose(ctx, v[:, 0] / (K + 1))


def test_adam_paddle_form_differs_from_torch_eps_placement():
    rng = np.random.RandomState(0)
    p = rng.standard_normal(5); g = rng.standard_normal(5) * 1e-6
    p1, m1, v1 = ops.adam_update(p, g, np.zeros(5), np.zeros(5), 1e-3, 1)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p1, p - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8), rtol=1e-12)


def test_greedy_first_token_is_start_and_no_early_stop(tiny_cfg):
    cfg = tiny_cfg
    params, image, caption = _setup(cfg)
    m = om.OracleModel(cfg, params)
    ids, logits = m.greedy_decode(image)
    assert logits.shape[1] == cfg['infer_max_length']
    # step 0 is fed <start>=2: recompute its logits through the train graph with caption [2, ...]
    cap = np.zeros_like(caption); cap[:, 0] = 2; cap[:, 1] = 5
    _, tl = om.OracleModel(cfg, params).forward_train(image, cap, update_stats=False)
    np.testing.assert_allclose(tl[:, 0], logits[:, 0], rtol=1e-9, atol=1e-9)


def test_beam_search_restatement_properties():
    """Beam search is a build-defined extension (the reference only has the greedy loop): its oracle is pinned by
    properties -- beam = 1 is the greedy loop exactly; the best hypothesis' score is the sum of its tokens'
    log-probabilities under teacher forcing; a wider beam never scores worse; the caption filter follows
    evaluate.py:14-25."""
    ocfg = om.default_cfg(encoder='mobilenetv2', image_size=64, hidden=24, embed=12, vocab=40, sentence_length=6,
                          infer_max_length=6, attention='slots')
    params = om.init_params(ocfg, seed=4, dtype=np.float64)
    rng = np.random.RandomState(2)
    image = rng.uniform(0, 1, (3, 3, 64, 64))
    fresh = lambda: om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    greedy, lg = fresh().greedy_decode(image)
    b1, s1, _ = fresh().beam_decode(image, 1)
    np.testing.assert_array_equal(b1, greedy)
    prev = s1
    for beam in (2, 4):
        ids, score, gaps = fresh().beam_decode(image, beam)
        assert np.all(score >= prev - 1e-9) and gaps.shape == (6, 3) and np.all(gaps >= 0)
        prev = score
        # re-score the returned hypothesis by teacher forcing through the step function
        m = fresh()
        t, _ = m._encoder_fwd(image, is_test=False, update_stats=False)
        A, V0, Amean, g = m._bridge_fwd(t[m.enc_out])
        Vt = np.tanh(om.ops.fc_fwd(V0, m.p[om.FC_IMG_FEAT + '.w_0'], m.p[om.FC_IMG_FEAT + '.b_0']))
        Ve = om.ops.fc_fwd(V0, m.p[om.FC_IMG_FEAT_EMB + '.w_0'], m.p[om.FC_IMG_FEAT_EMB + '.b_0'])
        hid = np.zeros((3, 24)); cell = np.zeros((3, 24)); word = np.full((3,), ocfg['start_idx'], np.int64)
        tot = np.zeros(3)
        for s in range(6):
            hid, cell, logits, _ = m._step_fwd(word, g, hid, cell, Vt, Ve)
            mx = logits.max(-1, keepdims=True)
            logp = logits - (mx + np.log(np.exp(logits - mx).sum(-1, keepdims=True)))
            word = ids[:, s].astype(np.int64)
            tot += logp[np.arange(3), word]
        np.testing.assert_allclose(tot, score, rtol=0, atol=1e-9)
    assert om.ids_to_tokens(np.array([2., 9., 0., 4., 3., 8.])) == [2, 9, 4]
    assert om.ids_to_tokens(np.array([3., 1.])) == []
