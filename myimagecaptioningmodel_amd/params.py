"""Parameter store: every persistable variable of the reference train program in one flat f32
device buffer (plus flat gradient / Adam-moment buffers of the same layout).

Names and reference layouts follow the reference checkpoint (SURVEY.md section 5;
/root/reference/ImageCaptioning/model/MobileNetV2.py:108,111-117,
model_adaAttention_aic.py:16-23,29-32,87-88 and the `fc_<n>` auto-names in creation order).
Inside the flat buffer each tensor is stored in the KERNEL layout the gfx950 kernels read:

    conv filter  [Cout, Cin, kh, kw]  ->  [Cout, kh, kw, Cin]   (stem: space-to-depth form [Cout, kt, kt, Cs])
    depthwise    [C, 1, 3, 3]         ->  [3, 3, C]
    fc weight    [in, out]            ->  [out, in]
    lstm_w       [(E+H)+H, 4H]        ->  [4H, (E+H)+H]
    fc_10.w_0    [H, 1]               ->  [H]
    everything else unchanged

`to_kernel` / `to_reference` convert single tensors (pure NumPy, used by checkpoint I/O and the
tests).  Buffer order is decoder first, then the encoder from its LAST layer to its first, i.e.
the order in which backward finishes the gradients, so all-reduce buckets are contiguous slices.
For every batch norm the offset is stored directly before the scale: the backward reduction
kernel writes [sum dz | sum dz*xhat] = [d offset | d scale] straight into the gradient buffer.
"""
from collections import OrderedDict, namedtuple

import numpy as np
import torch

from . import arch

Entry = namedtuple('Entry', 'name offset kshape ref_shape kind trainable')

ALIGN = 8          # elements; keeps f32 views 32-byte and bf16 shadows 16-byte aligned

# fc auto-names, creation order under unique_name.guard (IC/train.py:38): see SURVEY.md section 5
FC = dict(img_embed='fc_0', img_global='fc_1', img_feat='fc_2', img_feat_emb='fc_3',
          p_word='fc_5', p_hidden='fc_6', p_hid='fc_7', hid_emb='fc_8', sent_emb='fc_9',
          alpha='fc_10', out='fc_11', proj='fc_12')


def stem_s2d(k, cin):
    """(taps per axis, channels) of the space-to-depth form of the k x k / stride-2 stem (capmi_s2d_stem)."""
    return (k + 1) // 2, (4 * cin + 7) // 8 * 8


def decoder_param_specs(C, H, E, V, rnn_layer=1):
    """(name, ref_shape, kind) for the bridge + decoder, reference creation order."""
    specs = []

    def fc(key, i, o):
        specs.append((FC[key] + '.w_0', (i, o), 'fc_w' if o > 1 else 'fc_col'))
        specs.append((FC[key] + '.b_0', (o,), 'vec'))

    fc('img_embed', C, H)       # model_adaAttention_aic.py:196
    fc('img_global', C, H)      # :198
    fc('img_feat', H, H)        # :52
    fc('img_feat_emb', H, H)    # :53
    specs.append(('lstm_w', (E + H + H, 4 * H), 'fc_w'))     # :87-88 (fc_4 renamed by param_attr)
    specs.append(('lstm_b', (4 * H,), 'vec'))
    for l in range(1, rnn_layer):     # build-defined stacked layers (BASELINE configs[3]); the reference never reads rnn_layer (:42,46,174)
        specs.append(('lstm_w_l%d' % l, (H + H, 4 * H), 'fc_w'))
        specs.append(('lstm_b_l%d' % l, (4 * H,), 'vec'))
    fc('p_word', E + H, H)      # :89
    fc('p_hidden', H, H)        # :90
    fc('p_hid', H, H)           # :99
    fc('hid_emb', H, H)         # :102
    fc('sent_emb', H, H)        # :104
    fc('alpha', H, 1)           # :107
    fc('out', H, H)             # :115
    fc('proj', H, E)            # :24
    specs.append(('word_embedding', (V, E), 'same'))          # :16-19 / :29-32 (tied)
    specs.append(('out_fc_bias', (V,), 'vec'))                # :20-23
    return specs


def encoder_param_specs(enc):
    """Per ConvBN op, in op order: filter, then BN offset, BN scale (trainable) -- and the
    running mean/variance (state)."""
    params, state = [], []
    for op in enc.ops:
        if not isinstance(op, arch.ConvBN):
            continue
        kind = 'dwconv' if op.groups > 1 else ('stem' if op.cin < 8 else 'conv')
        params.append([(op.name + '_weights', (op.cout, op.cin // op.groups, op.k, op.k), kind),
                       (op.name + '_bn_offset', (op.cout,), 'vec'),
                       (op.name + '_bn_scale', (op.cout,), 'vec')])
        state.append([(op.name + '_bn_mean', (op.cout,)), (op.name + '_bn_variance', (op.cout,))])
    return params, state


def kernel_shape(ref_shape, kind):
    if kind == 'conv':
        o, c, kh, kw = ref_shape
        return (o, kh, kw, c)
    if kind == 'stem':
        o, c, kh, kw = ref_shape
        kt, cs = stem_s2d(kh, c)
        return (o, kt, kt, cs)
    if kind == 'dwconv':
        c, _, kh, kw = ref_shape
        return (kh, kw, c)
    if kind == 'fc_w':
        return (ref_shape[1], ref_shape[0])
    if kind == 'fc_col':
        return (ref_shape[0],)
    return tuple(ref_shape)


def to_kernel(arr, kind):
    """reference layout -> kernel layout (NumPy)."""
    if kind == 'conv':
        return np.ascontiguousarray(arr.transpose(0, 2, 3, 1))
    if kind == 'stem':      # Ws[n][r'][q'][(ph*2+pw)*C + c] = W[n][c][2r'+ph][2q'+pw]
        o, c, kh, kw = arr.shape
        kt, cs = stem_s2d(kh, c)
        out = np.zeros((o, kt, kt, cs), arr.dtype)
        for r in range(kh):
            for q in range(kw):
                sub = (r % 2) * 2 + (q % 2)
                out[:, r // 2, q // 2, sub * c:(sub + 1) * c] = arr[:, :, r, q]
        return out
    if kind == 'dwconv':
        return np.ascontiguousarray(arr[:, 0].transpose(1, 2, 0))
    if kind == 'fc_w':
        return np.ascontiguousarray(arr.T)
    if kind == 'fc_col':
        return np.ascontiguousarray(arr[:, 0])
    return np.ascontiguousarray(arr)


def to_reference(arr, kind, ref_shape):
    """kernel layout -> reference layout (NumPy)."""
    if kind == 'conv':
        return np.ascontiguousarray(arr.transpose(0, 3, 1, 2))
    if kind == 'stem':
        o, c, kh, kw = ref_shape
        out = np.zeros(ref_shape, arr.dtype)
        for r in range(kh):
            for q in range(kw):
                sub = (r % 2) * 2 + (q % 2)
                out[:, :, r, q] = arr[:, r // 2, q // 2, sub * c:(sub + 1) * c]
        return out
    if kind == 'dwconv':
        return np.ascontiguousarray(arr.transpose(2, 0, 1)[:, None])
    if kind == 'fc_w':
        return np.ascontiguousarray(arr.T)
    if kind == 'fc_col':
        return np.ascontiguousarray(arr[:, None])
    return np.ascontiguousarray(arr)


class ParamStore:
    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.enc = arch.encoder(cfg['encoder'])
        H, E, V = cfg['hidden'], cfg['embed'], cfg['vocab']
        self.entries = OrderedDict()
        off = 0

        def add(name, ref_shape, kind, trainable):
            nonlocal off
            ks = kernel_shape(ref_shape, kind)
            self.entries[name] = Entry(name, off, ks, tuple(ref_shape), kind, trainable)
            off += (int(np.prod(ks)) + ALIGN - 1) // ALIGN * ALIGN

        for name, shp, kind in decoder_param_specs(self.enc.channels, H, E, V, int(cfg.get('rnn_layer', 1))):
            add(name, shp, kind, True)
        self.decoder_size = off
        enc_params, enc_state = encoder_param_specs(self.enc)
        for group in reversed(enc_params):
            for name, shp, kind in group:
                add(name, shp, kind, bool(cfg['encoder_trainable']))     # MobileNetV2.py:27-29
        self.size = off
        self.trainable_size = self.size if cfg['encoder_trainable'] else self.decoder_size
        self.flat = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.state = OrderedDict()
        for group in enc_state:
            for name, shp in group:
                self.state[name] = (torch.ones if name.endswith('variance') else torch.zeros)(
                    shp, dtype=torch.float32, device=self.device)

    # ---------------------------------------------------------------- views
    def view(self, name, buf=None):
        e = self.entries[name]
        buf = self.flat if buf is None else buf
        return buf[e.offset:e.offset + int(np.prod(e.kshape))].view(e.kshape)

    def gview(self, name):
        return self.view(name, self.grad)

    def names(self):
        return list(self.entries) + list(self.state)

    # ---------------------------------------------------------------- reference-layout I/O
    def load_reference(self, params):
        """params: name -> ndarray in the REFERENCE layout (missing names are left untouched)."""
        for name, e in self.entries.items():
            if name in params:
                a = np.asarray(params[name], dtype=np.float32)
                if a.shape != e.ref_shape:
                    raise ValueError('%s: expected shape %s, got %s' % (name, e.ref_shape, a.shape))
                self.view(name).copy_(torch.from_numpy(to_kernel(a, e.kind)))
        for name, t in self.state.items():
            if name in params:
                t.copy_(torch.from_numpy(np.asarray(params[name], dtype=np.float32)))

    def export_reference(self, buf=None):
        out = OrderedDict()
        for name, e in self.entries.items():
            out[name] = to_reference(self.view(name, buf).detach().cpu().numpy(), e.kind, e.ref_shape)
        if buf is None:
            for name, t in self.state.items():
                out[name] = t.detach().cpu().numpy().copy()
        return out

    def export_reference_grads(self):
        return self.export_reference(self.grad)

    # ---------------------------------------------------------------- initialisers (quirk Q7)
    def init_reference(self, seed=0):
        """Reference initialisers (Paddle defaults): conv N(0, sqrt(2/(k*k*Cin))),
        BN scale 1 / offset 0 / mean 0 / variance 1, fc + lstm_w Xavier-uniform, biases 0,
        `word_embedding` U(-1, 1)."""
        rng = np.random.RandomState(seed)
        vals = {}
        for name, e in self.entries.items():
            shp = e.ref_shape
            if e.kind in ('conv', 'stem', 'dwconv'):
                v = rng.normal(0.0, np.sqrt(2.0 / (shp[1] * shp[2] * shp[3])), shp)
            elif name.endswith('_bn_scale'):
                v = np.ones(shp)
            elif name == 'word_embedding':
                v = rng.uniform(-1.0, 1.0, shp)
            elif len(shp) == 2:
                lim = np.sqrt(6.0 / (shp[0] + shp[1]))
                v = rng.uniform(-lim, lim, shp)
            else:
                v = np.zeros(shp)
            vals[name] = v.astype(np.float32)
        self.load_reference(vals)
        for name, t in self.state.items():
            t.fill_(1.0 if name.endswith('variance') else 0.0)
