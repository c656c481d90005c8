"""NumPy restatement of the PaddlePaddle-1.8 ops the reference hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function follows the op semantics
of the Paddle layer named in its docstring and cites the reference call site
(`IC/` = /root/reference/ImageCaptioning/).  Paddle is not installable here, so each
semantic is "unverified against Paddle" unless a test pins it analytically.

All functions are dtype-preserving: feed float64 arrays for the high-precision oracle,
float32 for the reference's own precision.  Layouts are the reference's: NCHW activations,
OIHW conv filters, fc weights [in, out], embedding table [V, E].
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


# ----------------------------------------------------------------------------- activations
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def relu6(x):
    """fluid.layers.relu6 (IC/model/MobileNetV2.py:119): min(max(x,0),6)."""
    return np.minimum(np.maximum(x, 0), 6)


def relu6_bwd(dy, x):
    return dy * ((x > 0) & (x < 6))


def relu(x):
    return np.maximum(x, 0)


def relu_bwd(dy, y):
    return dy * (y > 0)


# ----------------------------------------------------------------------------- conv2d
def _windows(x, kh, kw, stride, pad):
    """[B,C,H,W] -> view [B,C,Ho,Wo,kh,kw] of the zero-padded input."""
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))
    return win[:, :, ::stride, ::stride]


def conv2d_fwd(x, w, stride=1, pad=0, groups=1):
    """fluid.layers.conv2d, bias_attr=False (IC/model/MobileNetV2.py:99-109).

    x [B,C,H,W], w [O, C/groups, kh, kw].  groups is 1 (dense) or C (depthwise, the
    `use_cudnn=False` call at MobileNetV2.py:155-164).  Cross-correlation, zero padding.
    """
    O, Cg, kh, kw = w.shape
    B, C, H, W = x.shape
    if groups == 1:
        assert Cg == C
        if kh == 1 and kw == 1 and pad == 0:
            xs = x[:, :, ::stride, ::stride]
            return np.einsum('bchw,oc->bohw', xs, w[:, :, 0, 0], optimize=True)
        win = _windows(x, kh, kw, stride, pad)
        return np.einsum('bchwij,ocij->bohw', win, w, optimize=True)
    assert groups == C and Cg == 1 and O == C, "only dense or depthwise convs are on the path"
    win = _windows(x, kh, kw, stride, pad)
    return np.einsum('bchwij,cij->bchw', win, w[:, 0], optimize=True)


def conv2d_bwd(dy, x, w, stride=1, pad=0, groups=1, need_dx=True):
    """Gradient of conv2d_fwd w.r.t. x and w (what Paddle's conv2d_grad computes)."""
    O, Cg, kh, kw = w.shape
    B, C, H, W = x.shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    win = _windows(x, kh, kw, stride, pad)
    if groups == 1:
        dw = np.einsum('bohw,bchwij->ocij', dy, win, optimize=True)
    else:
        dw = np.einsum('bchw,bchwij->cij', dy, win, optimize=True)[:, None]
    if not need_dx:
        return None, dw
    dxp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            if groups == 1:
                contrib = np.einsum('bohw,oc->bchw', dy, w[:, :, i, j], optimize=True)
            else:
                contrib = dy * w[None, :, 0, i, j, None, None]
            dxp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride] += contrib
    dx = dxp[:, :, pad:pad + H, pad:pad + W] if pad else dxp
    return dx, dw


# ----------------------------------------------------------------------------- pooling (ResNet extension only)
def maxpool3x3s2_fwd(x):
    """3x3 stride-2 pad-1 max pool of the ResNet stem (build-defined extension; the
    reference has no pooling: MobileNetV2(use_pooling=False), model_adaAttention_aic.py:141).
    Padding is -inf; ties pick the first tap in (row, col) order."""
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    win = sliding_window_view(xp, (3, 3), axis=(2, 3))[:, :, ::2, ::2]
    flat = win.reshape(win.shape[:4] + (9,))
    idx = flat.argmax(-1)
    return np.take_along_axis(flat, idx[..., None], -1)[..., 0], idx


def maxpool3x3s2_bwd(dy, idx, xshape):
    B, C, H, W = xshape
    Ho, Wo = dy.shape[2], dy.shape[3]
    dxp = np.zeros((B, C, H + 2, W + 2), dtype=dy.dtype)
    for t in range(9):
        i, j = divmod(t, 3)
        dxp[:, :, i:i + 2 * Ho:2, j:j + 2 * Wo:2] += dy * (idx == t)
    return dxp[:, :, 1:1 + H, 1:1 + W]


# ----------------------------------------------------------------------------- batch norm
BN_MOMENTUM = 0.9   # fluid.layers.batch_norm defaults (MobileNetV2.py:112-117 passes neither)
BN_EPS = 1e-5


def batch_norm_fwd(x, scale, offset, run_mean, run_var, is_test=False):
    """fluid.layers.batch_norm (IC/model/MobileNetV2.py:112-117), NCHW.

    Train mode (the only mode the reference's train AND in-training eval graphs use, quirk
    Q3/Q4): normalise with the batch mean and the BIASED batch variance; running stats
    <- momentum*running + (1-momentum)*batch (biased variance; unverified against Paddle).
    Returns y, (xhat, invstd), (new_run_mean, new_run_var).
    """
    if is_test:
        mean, var = run_mean, run_var
    else:
        mean = x.mean(axis=(0, 2, 3))
        var = x.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    y = xhat * scale[None, :, None, None] + offset[None, :, None, None]
    if is_test:
        return y, (xhat, invstd), (run_mean, run_var)
    m = x.dtype.type(BN_MOMENTUM)
    new_mean = run_mean * m + mean * (1 - m)
    new_var = run_var * m + var * (1 - m)
    return y, (xhat, invstd), (new_mean, new_var)


def batch_norm_bwd(dy, saved, scale):
    """Train-mode BN backward: returns dx, dscale, doffset."""
    xhat, invstd = saved
    n = dy.shape[0] * dy.shape[2] * dy.shape[3]
    doffset = dy.sum(axis=(0, 2, 3))
    dscale = (dy * xhat).sum(axis=(0, 2, 3))
    dx = (scale * invstd)[None, :, None, None] / n * (
        n * dy - doffset[None, :, None, None] - xhat * dscale[None, :, None, None])
    return dx, dscale, doffset


# ----------------------------------------------------------------------------- fc
def fc_fwd(x, w, b):
    """layers.fc without activation: mul(x flattened to 2-D on the last axis, W[in,out]) + b.
    Covers num_flatten_dims=1 on [B,in] and num_flatten_dims=2 on [B,K,in]
    (IC/model/model_adaAttention_aic.py:24,52,53,89,90,99,102,104,107,115,196,198)."""
    y = x @ w
    return y if b is None else y + b


def fc_bwd(dy, x, w):
    """Returns dx, dw, db for y = x @ w + b (leading axes flattened for dw/db)."""
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    return dy @ w.T, x2.T @ dy2, dy2.sum(0)


# ----------------------------------------------------------------------------- embedding
def embedding_fwd(ids, table, padding_idx):
    """fluid.embedding(padding_idx=0) (IC/model/model_adaAttention_aic.py:28-32): row gather;
    ids equal to padding_idx yield an all-zero row."""
    out = table[ids]
    out = np.where((ids == padding_idx)[..., None], np.zeros((), table.dtype), out)
    return out


def embedding_bwd(dout, ids, table_shape, padding_idx):
    """Scatter-add; rows looked up through padding_idx receive no gradient."""
    dtab = np.zeros(table_shape, dtype=dout.dtype)
    keep = ids != padding_idx
    np.add.at(dtab, ids[keep], dout[keep])
    return dtab


# ----------------------------------------------------------------------------- lstm_unit
def lstm_unit_fwd(x_t, h_prev, c_prev, w, b, forget_bias=0.0):
    """layers.lstm_unit (IC/model/model_adaAttention_aic.py:87-88): Paddle composite
    concat([x_t, h_prev]) -> fc(4H) -> lstm_unit op.  Gate blocks in order i, f, o, g
    (each H wide; unverified against Paddle), forget_bias default 0:
        c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(g);  h = sigmoid(o) * tanh(c)
    Returns h, c and the cache for backward."""
    H = h_prev.shape[-1]
    xin = np.concatenate([x_t, h_prev], axis=-1)
    gates = xin @ w + b
    i = sigmoid(gates[:, 0:H])
    f = sigmoid(gates[:, H:2 * H] + forget_bias)
    o = sigmoid(gates[:, 2 * H:3 * H])
    g = np.tanh(gates[:, 3 * H:4 * H])
    c = f * c_prev + i * g
    tc = np.tanh(c)
    h = o * tc
    return h, c, (xin, i, f, o, g, tc, c_prev)


def lstm_unit_bwd(dh, dc, cache, w):
    """Returns dx_t, dh_prev, dc_prev, dw, db."""
    xin, i, f, o, g, tc, c_prev = cache
    H = dh.shape[-1]
    do = dh * tc
    dc_tot = dc + dh * o * (1 - tc * tc)
    di = dc_tot * g
    df = dc_tot * c_prev
    dg = dc_tot * i
    dgates = np.concatenate([di * i * (1 - i), df * f * (1 - f), do * o * (1 - o), dg * (1 - g * g)], axis=-1)
    dxin = dgates @ w.T
    dw = xin.T @ dgates
    db = dgates.sum(0)
    nx = xin.shape[-1] - H
    return dxin[:, :nx], dxin[:, nx:], dc_tot * f, dw, db


# ----------------------------------------------------------------------------- loss
def softmax_with_cross_entropy_fwd(logits, label):
    """layers.softmax_with_cross_entropy, hard label, last axis
    (IC/model/model_adaAttention_aic.py:205-212): loss = logsumexp(logits) - logits[label].
    Returns loss[..., 1] (Paddle keeps the trailing axis) and the softmax."""
    m = logits.max(-1, keepdims=True)
    e = np.exp(logits - m)
    s = e.sum(-1, keepdims=True)
    lse = np.log(s) + m
    picked = np.take_along_axis(logits, label[..., None], -1)
    return lse - picked, e / s


def softmax_with_cross_entropy_bwd(dloss, softmax, label):
    d = softmax.copy()
    np.put_along_axis(d, label[..., None], np.take_along_axis(d, label[..., None], -1) - 1, -1)
    return d * dloss


def argmax_lowest(x):
    """layers.argmax(axis=-1) (model_adaAttention_aic.py:120); ties -> lowest index."""
    return x.argmax(-1)


# ----------------------------------------------------------------------------- optimizer
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # fluid.optimizer.Adam defaults (IC/train.py:31)


def adam_update(p, g, m, v, lr, step, clip=None):
    """One Paddle-1.8 `adam` op on one tensor (IC/train.py:26-31,45), `step` counted from 1:
        lr_t = lr * sqrt(1 - b2^step) / (1 - b1^step)
        m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= lr_t * m / (sqrt(v) + eps)
    (epsilon is NOT bias-corrected -- differs from torch.optim.Adam; unverified against
    Paddle).  `clip` = GradientClipByValue bound (IC/train.py:42-43), None = off (default).
    All in the dtype of p.  Returns new p, m, v."""
    t = p.dtype.type
    if clip:
        g = np.clip(g, -clip, clip)
    b1, b2, eps = t(ADAM_B1), t(ADAM_B2), t(ADAM_EPS)
    lr_t = t(lr) * np.sqrt(t(1) - b2 ** t(step)) / (t(1) - b1 ** t(step))
    m = b1 * m + (t(1) - b1) * g
    v = b2 * v + (t(1) - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p.astype(t), m.astype(t), v.astype(t)
