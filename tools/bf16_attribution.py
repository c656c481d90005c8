"""Which bf16 STORAGE POINT of the encoder costs the end-to-end gradient its direction?  CPU only (torch f64 autograd,
tests/torch_ref.py with one storage point at a time rounded to bf16 the way the bf16 engine stores it), so it runs in the
build container.  For each input regime it prints cos(gradient with the rounding, f64 gradient) of the conv weight
gradients (first layer, one per stage, last layer) and the worst over all conv / batch-norm tensors.

    python tools/bf16_attribution.py [resnet50|mobilenetv2] [S] [B] > profiles/r04_bf16_attribution.txt
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model as om
from tests import torch_ref
from tests.conftest import make_caption, structured_images

torch.set_num_threads(os.cpu_count() or 8)


def grads(cfg, params, image, caption, rounding):
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('_mean', '_variance'))) for k, v in params.items()}
    loss, _ = torch_ref.forward_loss(cfg, p, torch.tensor(image, dtype=torch.float64), torch.tensor(caption), rounding=rounding)
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in p.items() if v.grad is not None}


def cos(a, b):
    a, b = a.ravel(), b.ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def main():
    encoder = sys.argv[1] if len(sys.argv) > 1 else 'resnet50'
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cfg = om.default_cfg(encoder=encoder, image_size=S, hidden=64, embed=32, vocab=100, sentence_length=6, infer_max_length=6, attention='slots')
    rng = np.random.RandomState(4)
    params = om.init_params(cfg, seed=4, dtype=np.float64)
    for k in params:
        if k.endswith('_bn_scale') or k.endswith('_bn_offset') or k.endswith('.b_0') or k in ('lstm_b', 'out_fc_bias'):
            params[k] = params[k] + 0.1 * rng.standard_normal(params[k].shape)
    caption = make_caption(rng, B, cfg['sentence_length'], cfg['vocab'])
    regimes = {'noise U[0,1)': rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32),
               'structured': structured_images(B, S, seed=4)}
    enc_names = [n for n in params if n.endswith('_weights')]
    pick = [enc_names[0]] + [n for n in enc_names if n.endswith('_1_branch2a_weights') or n.endswith('_1_expand_weights')][:6] + [enc_names[-1]]
    points = [('w',), ('img',), ('raw',), ('act',), ('dy',), ('dz',), ('feat_grad',), ('w', 'img', 'raw', 'act'), ('dy', 'dz'),
              ('w', 'img', 'raw', 'act', 'dy', 'dz')]
    for rname, image in regimes.items():
        l0, g0 = grads(cfg, params, image, caption, ())
        print('== %s, %s %dx%d batch %d, random initialisation: loss %.6f' % (rname, encoder, S, S, B, l0))
        print('%-28s %s   worst(all encoder tensors)' % ('rounded to bf16', ' '.join('%-10s' % n.replace('_weights', '')[-10:] for n in pick)))
        for pt in points:
            l1, g1 = grads(cfg, params, image, caption, pt)
            enc_t = [n for n in g0 if n.endswith(('_weights', '_bn_scale', '_bn_offset'))]
            worst = min((cos(g1[n], g0[n]), n) for n in enc_t)
            print('%-28s %s   %.4f (%s)' % ('+'.join(pt), ' '.join('%-10.4f' % cos(g1[n], g0[n]) for n in pick), worst[0], worst[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
