import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_caption(rng, B, L, V, min_len=None):
    """Synthetic captions mirroring IC/preprocess/ai_challenge_tokenizer.py:81-86:
    <start>=2, content tokens in [4, V), <stop>=3, right-padded with <pad>=0."""
    cap = np.zeros((B, L), np.int64)
    lo = max(1, L // 2 - 1) if min_len is None else min_len
    for b in range(B):
        n = rng.randint(lo, L - 1)          # content tokens, leaves room for <start>/<stop>
        n = min(n, L - 2)
        cap[b, 0] = 2
        cap[b, 1:1 + n] = rng.randint(4, V, size=n)
        cap[b, 1 + n] = 3
    return cap


def structured_images(B, S, seed=0):
    """Images that differ from each other the way photographs do and U[0,1) noise does not: per image a few low-frequency
    waves per colour channel with its own orientation, phase, mean and contrast, plus an edge and a little texture, in [0, 1].
    Used where an end-to-end gradient has to be compared across precisions: over noise images every batch-norm input is the
    same distribution per image, the gradient that reaches the encoder is almost entirely common-mode, and batch-norm backward
    keeps only the small residue of it (profiles/r04_bf16_attribution.txt)."""
    rng = np.random.RandomState(1000 + seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, S), np.linspace(0, 1, S), indexing='ij')
    img = np.zeros((B, 3, S, S), np.float64)
    for b in range(B):
        mean, contrast = rng.uniform(0.25, 0.75), rng.uniform(0.1, 0.45)
        for c in range(3):
            f = np.zeros((S, S))
            for _ in range(3):
                kx, ky = rng.uniform(-3, 3, 2)
                f += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (kx * xx + ky * yy) + rng.uniform(0, 2 * np.pi))
            ex, ey, e0 = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.2, 0.8)
            f += 1.5 * ((ex * (xx - e0) + ey * (yy - e0)) > 0)              # an edge
            f += 0.15 * rng.standard_normal((S, S))                       # texture
            f = (f - f.mean()) / (f.std() + 1e-9)
            img[b, c] = mean + rng.uniform(-0.1, 0.1) + contrast * 0.5 * f
    return np.clip(img, 0.0, 1.0).astype(np.float32)


@pytest.fixture
def tiny_cfg():
    from oracle.model import default_cfg
    return default_cfg(encoder='mobilenetv2', image_size=64, hidden=32, embed=16, vocab=50,
                       sentence_length=6, infer_max_length=6)


@pytest.fixture
def deterministic():
    """capmi_set_deterministic(1) for the test (include/capmi.h): fixed-order reductions in place of every f32 atomic
    accumulation, so that launch-schedule variants of one computation can be compared with array_equal."""
    from myimagecaptioningmodel_amd import _lib
    prev = _lib.set_deterministic(True)
    yield
    _lib.set_deterministic(prev)
