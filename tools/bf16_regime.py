"""Where is an end-to-end bf16 encoder gradient comparable with the reference arithmetic?  (GPU box.)

The f32 engine trains `steps` Adam steps on structured images (tests/conftest.structured_images); at the resulting weights the
bf16 engine's and the f32 engine's gradients of one more batch are compared per tensor (cosine), next to the f32 engine's own
noise (two batch orders of the same batch -- the summation order moves, nothing else) and, with --oracle, to the torch f64
build of the graph (tests/torch_ref.py, pinned against the NumPy oracle by tests/test_oracle_vs_torch.py).

    python tools/bf16_regime.py resnet50 128 32 20 1e-3 [--oracle]
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model as om
from tests.conftest import make_caption, structured_images
from myimagecaptioningmodel_amd import default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine


def cos(a, b):
    a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-300))


def regime(encoder, S, B, steps, lr, oracle=False, emulate=False, heldout=False, nbatches=4, noise_images=False, H=64, E=32, V=100, L=6, quiet=False):
    kw = dict(encoder=encoder, image_size=S, hidden=H, embed=E, vocab=V, sentence_length=L, infer_max_length=L, attention='slots')
    ocfg = om.default_cfg(**kw)
    rng = np.random.RandomState(4)
    params = om.init_params(ocfg, seed=4, dtype=np.float64)
    caps = [make_caption(rng, B, L, V) for _ in range(nbatches)]
    if noise_images:
        imgs = [rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32) for _ in range(nbatches)]
    else:
        imgs = [structured_images(B, S, seed=10 + i) for i in range(nbatches)]
    e32 = CaptionEngine(default_cfg(dtype='f32', learning_rate=lr, batch_size=B, **kw), device='cuda:0', use_graph=False)
    e32.load_reference_params(params)
    losses = []
    for s in range(steps):
        loss, _ = e32.train_step(imgs[s % nbatches], caps[s % nbatches])
        losses.append(float(loss.cpu()[0]))
    trained = e32.export_reference_params()
    if heldout:         # the gradient of a batch the steps never saw: a large, well-conditioned signal in an over-fitted state
        imgs = [structured_images(B, S, seed=99)] + imgs
        caps = [make_caption(np.random.RandomState(99), B, L, V)] + caps
    e32.load_reference_params(trained)
    e32.forward_backward(imgs[0], caps[0])
    g32 = e32.export_reference_grads()
    perm = np.random.RandomState(0).permutation(B)
    e32.forward_backward(imgs[0][perm], caps[0][perm])
    g32p = e32.export_reference_grads()
    e16 = CaptionEngine(default_cfg(dtype='bf16', learning_rate=lr, batch_size=B, **kw), device='cuda:0', use_graph=False)
    e16.load_reference_params(trained)
    l16 = float(e16.forward_backward(imgs[0], caps[0]).cpu()[0])
    g16 = e16.export_reference_grads()
    l32 = float(e32.forward_backward(imgs[0], caps[0]).cpu()[0])
    enc_t = [n for n in g32 if n.endswith(('_weights', '_bn_scale', '_bn_offset'))]
    out = dict(losses=losses, l16=l16, l32=l32)
    for cls in ('_weights', '_bn_scale', '_bn_offset'):
        cs = sorted((cos(g16[n], g32[n]), n) for n in enc_t if n.endswith(cls))
        out['cos' + cls] = cs
        out['f32perm' + cls] = max((rel(g32p[n], g32[n]), n) for n in enc_t if n.endswith(cls))
    dec_t = [n for n in g32 if n not in enc_t and np.linalg.norm(g32[n]) > 0]
    out['cos_decoder'] = sorted((cos(g16[n], g32[n]), n) for n in dec_t)
    if oracle:
        from tests import torch_ref
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
        p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('_mean', '_variance'))) for k, v in trained.items()}
        t0 = time.time()
        lo, _ = torch_ref.forward_loss(ocfg, p, torch.tensor(imgs[0], dtype=torch.float64), torch.tensor(caps[0]))
        lo.backward()
        go = {k: v.grad.numpy() for k, v in p.items() if v.grad is not None}
        out['oracle_s'] = time.time() - t0
        out['loss_o'] = float(lo.detach())
        out['f32_vs_f64'] = max((rel(g32[n], go[n]), n) for n in enc_t)
        out['cos16_vs_f64'] = sorted((cos(g16[n], go[n]), n) for n in enc_t)
    if emulate:
        # the same graph in f64 on the CPU with the bf16 engine's STORAGE rounded (tests/torch_ref.py): what bf16 storage alone does
        from tests import torch_ref
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))

        def tg(rounding):
            p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('_mean', '_variance'))) for k, v in trained.items()}
            lo, _ = torch_ref.forward_loss(ocfg, p, torch.tensor(imgs[0], dtype=torch.float64), torch.tensor(caps[0]), rounding=rounding)
            lo.backward()
            return {k: v.grad.numpy() for k, v in p.items() if v.grad is not None}
        g0 = tg(())
        gmax = max(np.linalg.norm(g0[n]) for n in enc_t)
        live = [n for n in enc_t if np.linalg.norm(g0[n]) > 1e-4 * gmax]
        out['emu'] = {}
        for pt in (('w',), ('img',), ('raw',), ('act',), ('dy', 'dz'), ('w', 'img', 'raw', 'act', 'dy', 'dz')):
            g1 = tg(pt)
            cs = sorted((cos(g1[n], g0[n]), n) for n in live)
            out['emu']['+'.join(pt)] = cs
        out['eng16_vs_f64_live'] = sorted((cos(g16[n], g0[n]), n) for n in live)
        out['eng32_vs_f64_live'] = sorted((cos(g32[n], g0[n]), n) for n in live)
    if not quiet:
        print('== %s%s %dx%d batch %d, %d f32 Adam steps at lr %g on %s images: loss %.4f -> %.4f; at the trained weights f32 %.5f bf16 %.5f'
              % (encoder, ' (held-out batch)' if heldout else '', S, S, B, steps, lr, 'noise' if noise_images else 'structured', losses[0] if losses else float('nan'),
                 losses[-1] if losses else float('nan'), l32, l16))
        for cls in ('_weights', '_bn_scale', '_bn_offset'):
            cs = out['cos' + cls]
            print('  cos(bf16 engine, f32 engine) %-11s min %.4f (%s)  5th-lowest %.4f  median %.4f   | f32 engine, batch permuted: worst rel L2 %.2e (%s)'
                  % (cls, cs[0][0], cs[0][1], cs[min(4, len(cs) - 1)][0], cs[len(cs) // 2][0], out['f32perm' + cls][0], out['f32perm' + cls][1]))
        print('  cos decoder tensors: min %.4f (%s)' % out['cos_decoder'][0])
        if oracle:
            print('  torch f64 graph (%.0f s): loss %.6f; f32 engine worst rel L2 %.2e (%s); cos(bf16 engine, f64) min %.4f (%s) median %.4f'
                  % (out['oracle_s'], out['loss_o'], out['f32_vs_f64'][0], out['f32_vs_f64'][1], out['cos16_vs_f64'][0][0], out['cos16_vs_f64'][0][1],
                     out['cos16_vs_f64'][len(out['cos16_vs_f64']) // 2][0]))
        if emulate:
            print('  f64 graph with bf16 STORAGE emulated (live tensors: |g| > 1e-4 of the largest): min / 5th / median cosine against plain f64')
            for k, cs in out['emu'].items():
                print('    %-24s %.4f (%s) %.4f %.4f' % (k, cs[0][0], cs[0][1], cs[min(4, len(cs) - 1)][0], cs[len(cs) // 2][0]))
            for k in ('eng16_vs_f64_live', 'eng32_vs_f64_live'):
                cs = out[k]
                print('    %-24s %.4f (%s) %.4f %.4f' % (k, cs[0][0], cs[0][1], cs[min(4, len(cs) - 1)][0], cs[len(cs) // 2][0]))
        sys.stdout.flush()
    return out


if __name__ == '__main__':
    a = sys.argv[1:]
    oracle = '--oracle' in a
    emulate = '--emulate' in a
    heldout = '--heldout' in a
    if '--deterministic' in a:
        from myimagecaptioningmodel_amd import _lib
        _lib.set_deterministic(True)
    noise = '--noise' in a
    a = [x for x in a if not x.startswith('--')]
    regime(a[0], int(a[1]), int(a[2]), int(a[3]), float(a[4]), oracle=oracle, emulate=emulate, heldout=heldout, noise_images=noise)
