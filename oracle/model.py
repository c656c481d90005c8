"""NumPy restatement of the reference model, forward + hand-derived backward + Adam + greedy
decode -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; PARITY UNPINNED).

Follows, line by line:
  IC/model/model_adaAttention_aic.py:15-37   weight_tying_fc / embedding_function / zero state
  IC/model/model_adaAttention_aic.py:50-135  Decoder.call (per-timestep While loop, both modes)
  IC/model/model_adaAttention_aic.py:161-212 training_network / eval_network / _img2feature / loss
  IC/model/MobileNetV2.py:31-209             encoder (through oracle/arch.py)
  IC/train.py:26-47                          Adam wiring, optional GradientClipByValue
Parameters live in a dict keyed by the reference's variable names in the reference's layouts
(conv OIHW, fc [in,out], `word_embedding` [V,E]).
"""
import numpy as np

from . import ops
from .arch import encoder_ops

# fc auto-names in creation order under unique_name.guard (IC/train.py:38)
FC_IMG_EMBED, FC_IMG_GLOBAL, FC_IMG_FEAT, FC_IMG_FEAT_EMB = 'fc_0', 'fc_1', 'fc_2', 'fc_3'
FC_P_WORD, FC_P_HIDDEN, FC_P_HID, FC_HID_EMB = 'fc_5', 'fc_6', 'fc_7', 'fc_8'
FC_SENT_EMB, FC_ALPHA, FC_OUT, FC_PROJ = 'fc_9', 'fc_10', 'fc_11', 'fc_12'


def default_cfg(**kw):
    """Repo-default hyper-parameters (IC/config.py:15,22-24,50,55-60)."""
    cfg = dict(encoder='mobilenetv2', image_size=224, hidden=1024, embed=256, vocab=12295,
               sentence_length=35, infer_max_length=35, start_idx=2, stop_idx=3, padding_idx=0,
               encoder_trainable=True, attention='singleton', rnn_layer=1)
    cfg.update(kw)
    return cfg


def lstm_names(layer):
    """(weight, bias) variable names of LSTM layer `layer` (0 = the reference's only layer, :87-88)."""
    return ('lstm_w', 'lstm_b') if layer == 0 else ('lstm_w_l%d' % layer, 'lstm_b_l%d' % layer)


def param_shapes(cfg):
    """name -> shape for every persistable variable of the train program (reference names)."""
    enc, _, C = encoder_ops(cfg['encoder'])
    H, E, V = cfg['hidden'], cfg['embed'], cfg['vocab']
    shapes = {}
    for op in enc:
        if op[0] == 'conv_bn':
            _, name, _, _, cin, cout, k, _, _, groups, _ = op
            shapes[name + '_weights'] = (cout, cin // groups, k, k)
            for s in ('scale', 'offset', 'mean', 'variance'):
                shapes['%s_bn_%s' % (name, s)] = (cout,)
    for name, (i, o) in {FC_IMG_EMBED: (C, H), FC_IMG_GLOBAL: (C, H), FC_IMG_FEAT: (H, H),
                         FC_IMG_FEAT_EMB: (H, H), FC_P_WORD: (E + H, H), FC_P_HIDDEN: (H, H),
                         FC_P_HID: (H, H), FC_HID_EMB: (H, H), FC_SENT_EMB: (H, H),
                         FC_ALPHA: (H, 1), FC_OUT: (H, H), FC_PROJ: (H, E)}.items():
        shapes[name + '.w_0'] = (i, o)
        shapes[name + '.b_0'] = (o,)
    shapes['lstm_w'] = (E + H + H, 4 * H)      # model_adaAttention_aic.py:87-88
    shapes['lstm_b'] = (4 * H,)
    # BUILD-DEFINED extension (BASELINE configs[3]; the reference stores `rnn_layer` and never reads it, :42,46,174):
    # layer l >= 1 is another lstm_unit whose input is the hidden state of layer l-1 at the same step
    for l in range(1, cfg.get('rnn_layer', 1)):
        w, b = lstm_names(l)
        shapes[w] = (H + H, 4 * H)
        shapes[b] = (4 * H,)
    shapes['word_embedding'] = (V, E)          # :16-19, shared with :29-32
    shapes['out_fc_bias'] = (V,)               # :20-23
    return shapes


def is_trainable(name, cfg):
    if name.endswith('_bn_mean') or name.endswith('_bn_variance'):
        return False
    if not cfg['encoder_trainable'] and (name.endswith('_weights') or '_bn_' in name):
        return False                            # MobileNetV2.py:27-29 (quirk Q4)
    return True


def init_params(cfg, seed=0, dtype=np.float64):
    """Reference initialisers (quirk Q7; Paddle defaults from memory, unverified): conv
    N(0, sqrt(2/(k*k*Cin))), BN scale 1 / offset 0 / mean 0 / variance 1, fc Xavier-uniform,
    biases 0, `word_embedding` U(-1,1), `lstm_w` Xavier-uniform."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith('_weights'):
            fan = shp[1] * shp[2] * shp[3]
            v = rng.normal(0.0, np.sqrt(2.0 / fan), shp)
        elif name.endswith('_bn_scale') or name.endswith('_bn_variance'):
            v = np.ones(shp)
        elif name == 'word_embedding':
            v = rng.uniform(-1.0, 1.0, shp)
        elif len(shp) == 2:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            v = rng.uniform(-lim, lim, shp)
        else:
            v = np.zeros(shp)
        out[name] = v.astype(dtype)
    return out


class OracleModel:
    def __init__(self, cfg, params):
        self.cfg = cfg
        self.p = params
        self.enc_ops, self.enc_out, self.C = encoder_ops(cfg['encoder'])
        self.dtype = params['lstm_w'].dtype
        self.adam_m = {}
        self.adam_v = {}
        self.adam_step_count = 0

    # ------------------------------------------------------------------ encoder
    def _encoder_fwd(self, img, is_test=False, update_stats=True):
        p = self.p
        t = {0: img}
        cache = {}
        for op in self.enc_ops:
            if op[0] == 'conv_bn':
                _, name, src, dst, cin, cout, k, stride, pad, groups, act = op
                conv = ops.conv2d_fwd(t[src], p[name + '_weights'], stride, pad, groups)
                y, saved, new_stats = ops.batch_norm_fwd(
                    conv, p[name + '_bn_scale'], p[name + '_bn_offset'],
                    p[name + '_bn_mean'], p[name + '_bn_variance'], is_test)
                if update_stats and not is_test:
                    p[name + '_bn_mean'], p[name + '_bn_variance'] = new_stats
                cache[dst] = (saved, y)
                t[dst] = ops.relu6(y) if act == 'relu6' else ops.relu(y) if act == 'relu' else y
            elif op[0] == 'add':
                _, a, b, dst, act = op
                s = t[a] + t[b]                                   # MobileNetV2.py:123-124
                cache[dst] = s
                t[dst] = ops.relu(s) if act == 'relu' else s
            else:
                _, src, dst = op
                t[dst], idx = ops.maxpool3x3s2_fwd(t[src])
                cache[dst] = idx
        return t, cache

    def _encoder_bwd(self, dout, t, cache, grads):
        d = {self.enc_out: dout}
        p = self.p

        def acc(i, g):
            d[i] = g if i not in d else d[i] + g

        for op in reversed(self.enc_ops):
            if op[0] == 'conv_bn':
                _, name, src, dst, cin, cout, k, stride, pad, groups, act = op
                saved, y = cache[dst]
                dy = d.pop(dst)
                if act == 'relu6':
                    dy = ops.relu6_bwd(dy, y)
                elif act == 'relu':
                    dy = ops.relu_bwd(dy, y)
                dconv, dscale, doffset = ops.batch_norm_bwd(dy, saved, p[name + '_bn_scale'])
                dx, dw = ops.conv2d_bwd(dconv, t[src], p[name + '_weights'], stride, pad, groups,
                                        need_dx=(src != 0))
                grads[name + '_weights'] = dw
                grads[name + '_bn_scale'] = dscale
                grads[name + '_bn_offset'] = doffset
                if src != 0:
                    acc(src, dx)
            elif op[0] == 'add':
                _, a, b, dst, act = op
                dy = d.pop(dst)
                if act == 'relu':
                    dy = ops.relu_bwd(dy, cache[dst])
                acc(a, dy)
                acc(b, dy)
            else:
                _, src, dst = op
                acc(src, ops.maxpool3x3s2_bwd(d.pop(dst), cache[dst], t[src].shape))

    # ------------------------------------------------------------------ bridge (:191-199)
    def _bridge_fwd(self, feat):
        p = self.p
        B, C = feat.shape[0], feat.shape[1]
        A = feat.reshape(B, C, -1).transpose(0, 2, 1)                       # [B,K,C] :194-195
        V0 = ops.relu(ops.fc_fwd(A, p[FC_IMG_EMBED + '.w_0'], p[FC_IMG_EMBED + '.b_0']))   # :196
        Amean = A.mean(1)                                                   # :197
        g = ops.relu(ops.fc_fwd(Amean, p[FC_IMG_GLOBAL + '.w_0'], p[FC_IMG_GLOBAL + '.b_0']))  # :198
        return A, V0, Amean, g

    def _alpha(self, e):
        """`layers.fc(z, size=1, num_flatten_dims=2, act='softmax')` (:107): fc appends a
        softmax over the LAST axis, which has size 1 -> alpha == 1 (quirk Q1, 'singleton').
        'slots' = softmax over the K+1 slots, the intended adaptive attention (extension)."""
        if self.cfg['attention'] == 'singleton':
            m = e.max(-1, keepdims=True)
            ex = np.exp(e - m)
            return ex / ex.sum(-1, keepdims=True)
        m = e.max(1, keepdims=True)
        ex = np.exp(e - m)
        return ex / ex.sum(1, keepdims=True)

    def _step_fwd(self, word, g, hid, cell, Vt, Ve):
        """One body of the While loop, model_adaAttention_aic.py:84-117.

        rnn_layer = 1 (the reference): hid / cell are [B,H] arrays.  rnn_layer = L > 1 (build-defined stacked
        extension): hid / cell are [L,B,H]; layer 0 is the reference's lstm_unit on [emb ; g], layer l its own
        lstm_unit on the NEW hidden state of layer l-1; everything after the LSTM (sentinel gate :89-92,
        attention, output head) reads the TOP layer -- its previous hidden state, new hidden state, new cell."""
        p, cfg = self.p, self.cfg
        emb = ops.embedding_fwd(word, p['word_embedding'], cfg['padding_idx'])       # :84
        xt = np.concatenate([emb, g], axis=-1)                                        # :86
        nl = cfg.get('rnn_layer', 1)
        if nl == 1:
            h, c, lcache = ops.lstm_unit_fwd(xt, hid, cell, p['lstm_w'], p['lstm_b'])     # :87-88
            lcaches, h_state, c_state = [lcache], h, c
        else:
            hs, cs, lcaches, xin = [], [], [], xt
            for l in range(nl):
                wn, bn = lstm_names(l)
                hl, cl, lc = ops.lstm_unit_fwd(xin, hid[l], cell[l], p[wn], p[bn])
                hs.append(hl), cs.append(cl), lcaches.append(lc)
                xin = hl
            h_state, c_state = np.stack(hs), np.stack(cs)
            hid, h, c = hid[nl - 1], hs[-1], cs[-1]
        sg = ops.sigmoid(ops.fc_fwd(xt, p[FC_P_WORD + '.w_0'], p[FC_P_WORD + '.b_0'])
                         + ops.fc_fwd(hid, p[FC_P_HIDDEN + '.w_0'], p[FC_P_HIDDEN + '.b_0']))  # :89-91
        tc = np.tanh(c)
        sentinel = sg * tc                                                            # :92
        p_hid = np.tanh(ops.fc_fwd(h, p[FC_P_HID + '.w_0'], p[FC_P_HID + '.b_0']))    # :99
        hid_emb = ops.fc_fwd(p_hid, p[FC_HID_EMB + '.w_0'], p[FC_HID_EMB + '.b_0'])   # :102
        sent_emb = ops.fc_fwd(sentinel, p[FC_SENT_EMB + '.w_0'], p[FC_SENT_EMB + '.b_0'])  # :104
        feat_emb = np.concatenate([Ve, sent_emb[:, None, :]], axis=1)                 # :105
        z = np.tanh(feat_emb + hid_emb[:, None, :])                                   # :103,106
        e = ops.fc_fwd(z, p[FC_ALPHA + '.w_0'], p[FC_ALPHA + '.b_0'])                 # :107
        alpha = self._alpha(e)
        ctx_all = np.concatenate([Vt, sentinel[:, None, :]], axis=1)                  # :111
        context = (ctx_all * alpha).mean(1)                                           # :112-113
        ctxp = context + p_hid
        out = np.tanh(ops.fc_fwd(ctxp, p[FC_OUT + '.w_0'], p[FC_OUT + '.b_0']))       # :115
        proj = ops.fc_fwd(out, p[FC_PROJ + '.w_0'], p[FC_PROJ + '.b_0'])              # :24
        logits = proj @ p['word_embedding'].T + p['out_fc_bias']                      # :25
        cache = dict(word=word, xt=xt, hid_prev=hid, lcaches=lcaches, sg=sg, tc=tc, sentinel=sentinel,
                     h=h, p_hid=p_hid, z=z, alpha=alpha, ctx_all=ctx_all, ctxp=ctxp, out=out, proj=proj)
        return h_state, c_state, logits, cache

    def _zero_state(self, B):
        """create_zero_state (:35-37, :63); one (h, c) pair per layer."""
        nl, H = self.cfg.get('rnn_layer', 1), self.cfg['hidden']
        shape = (B, H) if nl == 1 else (nl, B, H)
        return np.zeros(shape, self.dtype), np.zeros(shape, self.dtype)

    # ------------------------------------------------------------------ train forward (:161-183)
    def forward_train(self, image, caption, update_stats=True):
        p, cfg = self.p, self.cfg
        dt = self.dtype
        image = image.astype(dt)
        target = caption[:, 1:]                                                       # :163
        source = caption[:, :-1]                                                      # :164
        mask = (target != cfg['padding_idx']).astype(dt)                              # :165-168
        scale_factor = mask.sum()                                                     # :169
        t, ecache = self._encoder_fwd(image, update_stats=update_stats)
        A, V0, Amean, g = self._bridge_fwd(t[self.enc_out])
        Vt = np.tanh(ops.fc_fwd(V0, p[FC_IMG_FEAT + '.w_0'], p[FC_IMG_FEAT + '.b_0']))       # :52
        Ve = ops.fc_fwd(V0, p[FC_IMG_FEAT_EMB + '.w_0'], p[FC_IMG_FEAT_EMB + '.b_0'])        # :53
        B = image.shape[0]
        hid, cell = self._zero_state(B)                                               # :63
        T = cfg['sentence_length'] - 1                                                # :66
        steps, logits = [], []
        for s in range(T):                                                            # :75-127
            hid, cell, lg, sc = self._step_fwd(source[:, s], g, hid, cell, Vt, Ve)
            steps.append(sc)
            logits.append(lg)
        logits = np.stack(logits, axis=1)                                             # :129-130 [B,T,V]
        ce, sm = ops.softmax_with_cross_entropy_fwd(logits, target)                   # :180, :205-212
        loss = (ce[..., 0] * mask).sum() / scale_factor                               # :181-182
        self._saved = dict(t=t, ecache=ecache, A=A, V0=V0, Amean=Amean, g=g, Vt=Vt, Ve=Ve, steps=steps,
                           sm=sm, target=target, mask=mask, scale=scale_factor, T=T)
        return loss, logits

    # ------------------------------------------------------------------ backward
    def backward(self):
        """d(loss)/d(every parameter), hand-derived; checked against torch.autograd and finite
        differences in tests/test_oracle_vs_torch.py."""
        p, cfg, S = self.p, self.cfg, self._saved
        dt = self.dtype
        H, E = cfg['hidden'], cfg['embed']
        grads = {k: np.zeros_like(v) for k, v in p.items()
                 if not (k.endswith('_bn_mean') or k.endswith('_bn_variance'))}
        Vt, Ve, g = S['Vt'], S['Ve'], S['g']
        K = Vt.shape[1]
        dloss = (S['mask'] / S['scale'])[..., None]
        dlogits_all = ops.softmax_with_cross_entropy_bwd(dloss, S['sm'], S['target'])   # [B,T,V]
        dVt = np.zeros_like(Vt)
        dVe = np.zeros_like(Ve)
        dg = np.zeros_like(g)
        nl = cfg.get('rnn_layer', 1)
        dh_next = [np.zeros_like(g) for _ in range(nl)]       # per layer: d loss / d h_{t-1}, d c_{t-1} from step t
        dc_next = [np.zeros_like(g) for _ in range(nl)]

        def fcb(name, dy, x):
            dx, dw, db = ops.fc_bwd(dy, x, p[name + '.w_0'])
            grads[name + '.w_0'] += dw
            grads[name + '.b_0'] += db
            return dx

        for s in reversed(range(S['T'])):
            c = S['steps'][s]
            dlogits = dlogits_all[:, s]
            grads['out_fc_bias'] += dlogits.sum(0)
            grads['word_embedding'] += dlogits.T @ c['proj']            # dense (tied) part, quirk Q6
            dproj = dlogits @ p['word_embedding']
            dout = fcb(FC_PROJ, dproj, c['out'])
            dctxp = fcb(FC_OUT, dout * (1 - c['out'] ** 2), c['ctxp'])
            dp_hid = dctxp.copy()
            dca = dctxp[:, None, :] / (K + 1)                           # reduce_mean over K+1 slots
            dctx_all = dca * c['alpha']
            dalpha = (dca * c['ctx_all']).sum(-1, keepdims=True)        # [B,K+1,1]
            dVt += dctx_all[:, :K]
            dsent = dctx_all[:, K].copy()
            a = c['alpha']
            if cfg['attention'] == 'singleton':
                de = a * (dalpha - (a * dalpha).sum(-1, keepdims=True))  # == 0 exactly (Q1)
            else:
                de = a * (dalpha - (a * dalpha).sum(1, keepdims=True))
            dz = fcb(FC_ALPHA, de, c['z'])
            dzpre = dz * (1 - c['z'] ** 2)
            dVe += dzpre[:, :K]
            dsent += fcb(FC_SENT_EMB, dzpre[:, K], c['sentinel'])
            dp_hid += fcb(FC_HID_EMB, dzpre.sum(1), c['p_hid'])
            dh = fcb(FC_P_HID, dp_hid * (1 - c['p_hid'] ** 2), c['h']) + dh_next[nl - 1]
            dsg = dsent * c['tc']
            dc = dsent * c['sg'] * (1 - c['tc'] ** 2) + dc_next[nl - 1]
            dsgpre = dsg * c['sg'] * (1 - c['sg'])
            dxt = fcb(FC_P_WORD, dsgpre, c['xt'])
            dhid_prev = fcb(FC_P_HIDDEN, dsgpre, c['hid_prev'])
            for l in reversed(range(nl)):                       # top layer first; layer l's input gradient is d h of layer l-1
                wn, bn = lstm_names(l)
                dxin, dh_prev_l, dc_prev_l, dlw, dlb = ops.lstm_unit_bwd(dh, dc, c['lcaches'][l], p[wn])
                grads[wn] += dlw
                grads[bn] += dlb
                dh_next[l] = dh_prev_l + (dhid_prev if l == nl - 1 else 0.0)
                dc_next[l] = dc_prev_l
                if l > 0:
                    dh, dc = dxin + dh_next[l - 1], dc_next[l - 1]      # (still step t+1's values: overwritten below)
                else:
                    dxt = dxt + dxin
            grads['word_embedding'] += ops.embedding_bwd(dxt[:, :E], c['word'],
                                                         p['word_embedding'].shape, cfg['padding_idx'])
            dg += dxt[:, E:]

        # pre-loop projections (:52-53)
        dV0 = fcb(FC_IMG_FEAT, dVt * (1 - Vt ** 2), S['V0']) + fcb(FC_IMG_FEAT_EMB, dVe, S['V0'])
        # bridge (:191-199)
        dA = fcb(FC_IMG_EMBED, ops.relu_bwd(dV0, S['V0']), S['A'])
        dAmean = fcb(FC_IMG_GLOBAL, ops.relu_bwd(dg, g), S['Amean'])
        dA = dA + dAmean[:, None, :] / K
        feat = S['t'][self.enc_out]
        dfeat = dA.transpose(0, 2, 1).reshape(feat.shape)
        if cfg['encoder_trainable']:
            self._encoder_bwd(dfeat, S['t'], S['ecache'], grads)
        self._dfeat = dfeat
        return {k: v.astype(dt) for k, v in grads.items()}

    # ------------------------------------------------------------------ optimizer (IC/train.py:26-47)
    def adam_step(self, grads, lr, clip=None):
        self.adam_step_count += 1
        for name, gval in grads.items():
            if not is_trainable(name, self.cfg):
                continue
            m = self.adam_m.get(name, np.zeros_like(self.p[name]))
            v = self.adam_v.get(name, np.zeros_like(self.p[name]))
            self.p[name], self.adam_m[name], self.adam_v[name] = ops.adam_update(
                self.p[name], gval, m, v, lr, self.adam_step_count, clip)

    # ------------------------------------------------------------------ eval graph (:185-189 + eval branches)
    def greedy_decode(self, image, is_test=False, update_stats=True):
        """Greedy decode: first fed token start_idx (:56-58), `infer_max_length` steps with no
        early stop (:66-68, quirk Q5), argmax feedback (:119-121), ids returned as FLOAT32
        [B,Ti] (:122-123,132-133, quirk Q2).  BN uses batch statistics unless is_test (Q3)."""
        p, cfg = self.p, self.cfg
        dt = self.dtype
        t, _ = self._encoder_fwd(image.astype(dt), is_test=is_test, update_stats=update_stats)
        A, V0, Amean, g = self._bridge_fwd(t[self.enc_out])
        Vt = np.tanh(ops.fc_fwd(V0, p[FC_IMG_FEAT + '.w_0'], p[FC_IMG_FEAT + '.b_0']))
        Ve = ops.fc_fwd(V0, p[FC_IMG_FEAT_EMB + '.w_0'], p[FC_IMG_FEAT_EMB + '.b_0'])
        B = image.shape[0]
        hid, cell = self._zero_state(B)
        word = np.full((B,), cfg['start_idx'], np.int64)
        out, all_logits = [], []
        for _ in range(cfg['infer_max_length']):
            hid, cell, logits, _c = self._step_fwd(word, g, hid, cell, Vt, Ve)
            word = ops.argmax_lowest(logits).astype(np.int64)
            out.append(word.astype(np.float32))
            all_logits.append(logits)
        return np.stack(out, axis=1), np.stack(all_logits, axis=1)

    def beam_decode(self, image, beam, is_test=False, update_stats=True):
        """Beam-search decode (BUILD-DEFINED extension, BASELINE cfg 5; the reference only has the greedy
        loop).  Semantics chosen so that beam = 1 IS greedy_decode: fixed infer_max_length steps, no early
        stop and no special casing of <stop> (quirk Q5 carried over; the caller truncates with ids_to_tokens),
        score = sum of log-softmax probabilities, no length normalisation.  Every step keeps the `beam` best
        of the beam x V continuations per image; ties go to the lower beam index, then the lower token id.
        Hypothesis 0 starts with score 0, the others with -1e30 (all beams hold <start>).  Returns
        (ids float32 [B, Ti] of the best final hypothesis, its score [B], per-step gap [Ti, B] between the
        beam-th and (beam+1)-th candidate -- the tests' near-tie guard)."""
        p, cfg = self.p, self.cfg
        dt = self.dtype
        t, _ = self._encoder_fwd(image.astype(dt), is_test=is_test, update_stats=update_stats)
        A, V0, Amean, g = self._bridge_fwd(t[self.enc_out])
        Vt = np.tanh(ops.fc_fwd(V0, p[FC_IMG_FEAT + '.w_0'], p[FC_IMG_FEAT + '.b_0']))
        Ve = ops.fc_fwd(V0, p[FC_IMG_FEAT_EMB + '.w_0'], p[FC_IMG_FEAT_EMB + '.b_0'])
        B, Ti, V = image.shape[0], cfg['infer_max_length'], cfg['vocab']
        rep = lambda x: np.concatenate([x] * beam, axis=0)                # rows k*B + b (beam-major)
        g_, Vt_, Ve_ = rep(g), rep(Vt), rep(Ve)
        hid, cell = self._zero_state(beam * B)
        word = np.full((beam * B,), cfg['start_idx'], np.int64)
        score = np.full((beam, B), -1e30, np.float64)
        score[0] = 0.0
        tokens, parents, gaps = [], [], []
        for _ in range(Ti):
            hid, cell, logits, _c = self._step_fwd(word, g_, hid, cell, Vt_, Ve_)
            lg = logits.astype(np.float64).reshape(beam, B, V)
            m = lg.max(-1, keepdims=True)
            logp = lg - (m + np.log(np.exp(lg - m).sum(-1, keepdims=True)))
            tot = (score[:, :, None] + logp).transpose(1, 0, 2).reshape(B, beam * V)      # flat index k*V + v
            order = np.argsort(-tot, axis=1, kind='stable')[:, :beam + 1]
            best = order[:, :beam]
            gaps.append(np.take_along_axis(tot, order[:, beam - 1:beam], 1)[:, 0] - np.take_along_axis(tot, order[:, beam:beam + 1], 1)[:, 0])
            par, tok = best // V, best % V                                                 # [B, beam]
            score = np.take_along_axis(tot, best, 1).T.copy()                              # [beam, B]
            src = (par.T * B + np.arange(B)[None, :]).reshape(-1)                          # row of the parent state
            hid, cell = hid[..., src, :], cell[..., src, :]
            word = tok.T.reshape(-1).astype(np.int64)
            tokens.append(tok.T.copy())
            parents.append(par.T.copy())
        out = np.zeros((B, Ti), np.float32)
        j = np.zeros(B, np.int64)                                                          # best final hypothesis: rank 0
        for s in reversed(range(Ti)):
            out[:, s] = tokens[s][j, np.arange(B)]
            j = parents[s][j, np.arange(B)]
        return out, score[0].copy(), np.stack(gaps)


def ids_to_tokens(ids, stop_idx=3, pad_idx=0):
    """Host-side filter of evaluate.py:15-25 / infer.py: a caption is the ids up to (not including) the first
    <stop>, with <pad> ids skipped.  ids: one row of the float32 [B, Ti] decode output."""
    out = []
    for v in np.rint(np.asarray(ids)).astype(np.int64).tolist():      # float32 ids are rounded first (evaluate.py:30-33)
        if v == stop_idx:
            break
        if v != pad_idx:
            out.append(v)
    return out
