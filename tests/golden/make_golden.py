"""Generates tests/golden/*.npz from the NumPy oracle in THIS container (python tests/golden/make_golden.py).

The reference has no golden vectors and cannot run here (PaddlePaddle 1.8 is not installable), so
these fixtures pin the ORACLE against regressions and give the GPU tests a stored target: seeded
inputs are regenerated from the recorded seeds; stored are loss, a logits slice, greedy ids, a few
gradient norms, and parameter values after one Paddle-form Adam step.  f64 evaluation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as om          # noqa: E402
from tests.conftest import make_caption  # noqa: E402

CASES = {
    # BASELINE configs[0] plumbing shape (64x64, L=10, B=4) with reduced H/E/V so the fixture stays small
    'cfg1_mobilenetv2_singleton': dict(encoder='mobilenetv2', image_size=64, hidden=32, embed=16, vocab=100,
                                       sentence_length=10, infer_max_length=10, attention='singleton', B=4, seed=11),
    'cfg1_mobilenetv2_slots': dict(encoder='mobilenetv2', image_size=64, hidden=32, embed=16, vocab=100,
                                   sentence_length=10, infer_max_length=10, attention='slots', B=4, seed=24),
    'resnet50_slots': dict(encoder='resnet50', image_size=64, hidden=32, embed=16, vocab=100,
                           sentence_length=8, infer_max_length=8, attention='slots', B=4, seed=13),
}
GRAD_NAMES = ['lstm_w', 'word_embedding', 'fc_0.w_0', 'fc_2.w_0', 'fc_11.w_0', 'out_fc_bias']


def inputs(case):
    c = dict(case)
    B, seed = c.pop('B'), c.pop('seed')
    cfg = om.default_cfg(**c)
    rng = np.random.RandomState(seed)
    params = om.init_params(cfg, seed=seed, dtype=np.float64)
    image = rng.uniform(0, 1, (B, 3, cfg['image_size'], cfg['image_size'])).astype(np.float32)
    caption = make_caption(rng, B, cfg['sentence_length'], cfg['vocab'])
    return cfg, params, image, caption


def evaluate(case):
    cfg, params, image, caption = inputs(case)
    m = om.OracleModel(cfg, {k: v.copy() for k, v in params.items()})
    loss, logits = m.forward_train(image.astype(np.float64), caption, update_stats=False)
    grads = m.backward()
    out = dict(loss=np.float64(loss), logits_slice=logits[:2, :3, :16].copy(), caption=caption)
    for n in GRAD_NAMES:
        out['gradnorm_' + n] = np.float64(np.linalg.norm(grads[n]))
    first_conv = [k for k in grads if k.endswith('_weights')][0]
    out['gradnorm_first_conv'] = np.float64(np.linalg.norm(grads[first_conv]))
    m.adam_step(grads, lr=1e-3)
    out['adam_lstm_b'] = m.p['lstm_b'][:32].copy()
    out['adam_out_fc_bias'] = m.p['out_fc_bias'][:32].copy()
    m2 = om.OracleModel(cfg, {k: v.copy() for k, v in params.items()})
    ids, dlogits = m2.greedy_decode(image.astype(np.float64), update_stats=False)
    out['greedy_ids'] = ids
    top2 = np.sort(dlogits, axis=-1)[..., -2:]
    out['greedy_margin'] = np.float64((top2[..., 1] - top2[..., 0]).min())
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    for name, case in CASES.items():
        out = evaluate(case)
        np.savez_compressed(os.path.join(here, name + '.npz'), **out)
        print(name, 'loss %.6f' % out['loss'], 'greedy margin %.3e' % out['greedy_margin'])
