"""A regime in which an end-to-end ENCODER gradient can be compared across precisions.

At random initialisation the gradient that reaches the encoder is a small residue of a large common-mode term (the loss
barely depends on the image yet), and its direction turns over under a 2^-9 perturbation of ANY forward tensor: rounding only
the image, or only the filters, or only the conv outputs to bf16 in an otherwise f64 evaluation of the graph already takes
the cosine of the conv gradients to 0.2-0.5 (profiles/r04_bf16_attribution.txt; the gradient buffers themselves can be
rounded at no cost: 0.9999).  After a few hundred optimizer steps on images that differ from each other the encoder carries
real signal and the same comparison reads >= 0.95 on every tensor.  This module builds that state on the GPU: structured
images (tests/conftest.structured_images), `steps` Adam steps of the f32 engine from the reference initialisation."""
import numpy as np

from oracle import model as om
from tests.conftest import make_caption, structured_images


def model_cfgs(encoder, S, B, lr, dtype, H=64, E=32, V=100, L=6):
    from myimagecaptioningmodel_amd import default_cfg
    kw = dict(encoder=encoder, image_size=S, hidden=H, embed=E, vocab=V, sentence_length=L, infer_max_length=L, attention='slots')
    return om.default_cfg(**kw), default_cfg(dtype=dtype, learning_rate=lr, batch_size=B, **kw)


def batches(ocfg, B, n, seed=4, noise=False):
    rng = np.random.RandomState(seed)
    S, L, V = ocfg['image_size'], ocfg['sentence_length'], ocfg['vocab']
    caps = [make_caption(rng, B, L, V) for _ in range(n)]
    if noise:
        imgs = [rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32) for _ in range(n)]
    else:
        imgs = [structured_images(B, S, seed=10 + i) for i in range(n)]
    return imgs, caps


def heldout_batch(ocfg, B, seed=99):
    """A batch the training steps never saw: in the (over-fitted) trained state its loss is large and its gradient a strong,
    well-conditioned signal -- the gradient of a TRAINING batch there is again a small residue (the model has fitted it)."""
    return structured_images(B, ocfg['image_size'], seed=seed), make_caption(np.random.RandomState(seed), B, ocfg['sentence_length'], ocfg['vocab'])


def trained_params(encoder, S, B, steps, lr, nbatches=4, seed=4):
    """(oracle cfg, images, captions, parameters after `steps` f32 Adam steps, the losses of those steps)."""
    from myimagecaptioningmodel_amd.model import CaptionEngine
    ocfg, ecfg = model_cfgs(encoder, S, B, lr, 'f32')
    imgs, caps = batches(ocfg, B, nbatches, seed)
    eng = CaptionEngine(ecfg, device='cuda:0', use_graph=False)
    eng.load_reference_params(om.init_params(ocfg, seed=seed, dtype=np.float64))
    losses = []
    for s in range(steps):
        losses.append(float(eng.train_step(imgs[s % nbatches], caps[s % nbatches])[0].cpu()[0]))
    eng.check_sync()
    return ocfg, imgs, caps, eng.export_reference_params(), losses


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300))
