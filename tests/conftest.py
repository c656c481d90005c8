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
