"""Independent torch (CPU, autograd) build of the reference graph, used ONLY to pin the NumPy
oracle (tests/test_oracle_vs_torch.py).  It shares the op list in oracle/arch.py but none of
the oracle's arithmetic: convs are F.conv2d, gradients come from torch.autograd."""
import torch
import torch.nn.functional as F

from oracle.arch import encoder_ops
from oracle import model as om


def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundValue(torch.autograd.Function):
    """y = bf16(x) in the forward pass, identity in the backward pass: a tensor STORED in bf16."""
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """Identity in the forward pass, bf16(g) in the backward pass: a GRADIENT buffer stored in bf16."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


STORAGE_POINTS = ('w', 'img', 'raw', 'act', 'dy', 'dz', 'feat_grad')


def forward_loss(cfg, params, image, caption, rounding=()):
    """params: dict name -> torch tensor (requires_grad where wanted), reference layouts.
    rounding: the ENCODER storage points rounded to bf16 the way the bf16 engine stores them (tools/bf16_attribution.py,
    tests/test_bf16_attribution.py) -- 'w' conv filters (the bf16 shadow), 'img' the feed, 'raw' conv outputs (the batch
    statistics still come from the unrounded accumulators, as in the conv epilogue), 'act' the activated tensors, 'dy' the
    gradient buffers of the activated tensors, 'dz' the gradients w.r.t. the conv outputs (the weight-gradient / data-gradient
    operand), 'feat_grad' only the gradient the decoder hands to the encoder output.  Empty: the plain graph."""
    p = params
    rv = lambda key, x: _RoundValue.apply(x) if key in rounding else x
    rg = lambda key, x: _RoundGrad.apply(x) if key in rounding else x
    enc, enc_out, C = encoder_ops(cfg['encoder'])
    uses = {}
    for op in enc:
        for s_ in ((op[2],) if op[0] == 'conv_bn' else (op[1], op[2]) if op[0] == 'add' else (op[1],)):
            uses[s_] = uses.get(s_, 0) + 1
    # a linear conv + batch norm whose only reader is a residual add is never stored (the add rides in its bn_apply)
    unstored = {op[2] for op in enc if op[0] == 'add' and uses.get(op[2], 0) == 1}
    t = {0: rv('img', image)}
    for op in enc:
        if op[0] == 'conv_bn':
            _, name, src, dst, cin, cout, k, stride, pad, groups, act = op
            y = F.conv2d(t[src], rv('w', p[name + '_weights']), None, stride, pad, 1, groups)
            y = rg('dz', y)
            mean = y.mean(dim=(0, 2, 3), keepdim=True)
            var = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            if 'raw' in rounding:       # the stored copy is rounded, the statistics are those of the f32 accumulators
                y = y + (_bf16(y) - y).detach()
            y = (y - mean) / torch.sqrt(var + 1e-5)
            y = y * p[name + '_bn_scale'][None, :, None, None] + p[name + '_bn_offset'][None, :, None, None]
            if act == 'relu6':
                y = torch.clamp(y, 0, 6)
            elif act == 'relu':
                y = torch.relu(y)
            t[dst] = y if (act is None and dst in unstored) else rg('dy', rv('act', y))
        elif op[0] == 'add':
            _, a, b, dst, act = op
            y = t[a] + t[b]
            t[dst] = rg('dy', rv('act', torch.relu(y) if act == 'relu' else y))
        else:
            _, src, dst = op
            t[dst] = F.max_pool2d(t[src], 3, 2, 1)
    feat = rg('feat_grad', t[enc_out])
    B = feat.shape[0]
    A = feat.reshape(B, C, -1).permute(0, 2, 1)

    def fc(name, x):
        return x @ p[name + '.w_0'] + p[name + '.b_0']

    V0 = torch.relu(fc(om.FC_IMG_EMBED, A))
    g = torch.relu(fc(om.FC_IMG_GLOBAL, A.mean(1)))
    Vt = torch.tanh(fc(om.FC_IMG_FEAT, V0))
    Ve = fc(om.FC_IMG_FEAT_EMB, V0)
    H = cfg['hidden']
    nl = cfg.get('rnn_layer', 1)
    hids = [torch.zeros(B, H, dtype=feat.dtype) for _ in range(nl)]
    cells = [torch.zeros(B, H, dtype=feat.dtype) for _ in range(nl)]
    target = caption[:, 1:]
    source = caption[:, :-1]
    mask = (target != cfg['padding_idx']).to(feat.dtype)
    logits = []
    emb_table = p['word_embedding']
    for s in range(cfg['sentence_length'] - 1):
        w = source[:, s]
        emb = emb_table[w] * (w != cfg['padding_idx']).to(feat.dtype)[:, None]
        xt = torch.cat([emb, g], -1)
        hid = hids[-1]                      # the sentinel gate reads the top layer's previous hidden state
        xin = xt
        for l in range(nl):                 # stacked lstm_unit layers (build-defined for l >= 1)
            wn, bn = om.lstm_names(l)
            gates = torch.cat([xin, hids[l]], -1) @ p[wn] + p[bn]
            i, f, o, gg = gates.split(H, dim=-1)
            c = torch.sigmoid(f) * cells[l] + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hids[l], cells[l] = h, c
            xin = h
        sg = torch.sigmoid(fc(om.FC_P_WORD, xt) + fc(om.FC_P_HIDDEN, hid))
        sentinel = sg * torch.tanh(c)
        p_hid = torch.tanh(fc(om.FC_P_HID, h))
        hid_emb = fc(om.FC_HID_EMB, p_hid)
        sent_emb = fc(om.FC_SENT_EMB, sentinel)
        z = torch.tanh(torch.cat([Ve, sent_emb[:, None]], 1) + hid_emb[:, None])
        e = fc(om.FC_ALPHA, z)
        alpha = torch.softmax(e, dim=-1 if cfg['attention'] == 'singleton' else 1)
        ctx = (torch.cat([Vt, sentinel[:, None]], 1) * alpha).mean(1)
        out = torch.tanh(fc(om.FC_OUT, ctx + p_hid))
        logits.append(fc(om.FC_PROJ, out) @ emb_table.t() + p['out_fc_bias'])
    logits = torch.stack(logits, 1)
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1), reduction='none')
    loss = (ce.reshape(target.shape) * mask).sum() / mask.sum()
    return loss, logits
