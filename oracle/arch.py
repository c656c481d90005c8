"""Encoder topologies as a tiny op list -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

An encoder is a list of ops over integer tensor ids (id 0 = the input image):
    ('conv_bn', name, src, dst, cin, cout, k, stride, pad, groups, act)   act in {None,'relu','relu6'}
    ('add',     a, b, dst, act)                                          dst = act(a + b)
    ('maxpool', src, dst)                                                3x3 s2 p1
`name` is the reference variable-name stem: conv filter `<name>_weights`, BN
`<name>_bn_{scale,offset,mean,variance}` (IC/model/MobileNetV2.py:108,111-117).
"""


def mobilenet_v2_ops():
    """IC/model/MobileNetV2.py:31-86 (`net`), :128-181 (`inverted_residual_unit`),
    :183-209 (`invresi_blocks`); scale=1.0, use_pooling=False
    (model_adaAttention_aic.py:141).  Output channels 1280, spatial S/32."""
    ops = []
    nid = [0]

    def new():
        nid[0] += 1
        return nid[0]

    def conv_bn(name, src, cin, cout, k, stride, pad, groups, act):
        dst = new()
        ops.append(('conv_bn', name, src, dst, cin, cout, k, stride, pad, groups, act))
        return dst

    cur = conv_bn('conv1_1', 0, 3, 32, 3, 2, 1, 1, 'relu6')          # :49-56
    in_c = 32
    settings = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]       # :37-45
    i = 1
    for t, c, n, s in settings:
        i += 1
        for j in range(n):
            name = 'conv%d_%d' % (i, j + 1)
            stride = s if j == 0 else 1
            cin = in_c if j == 0 else c
            cexp = int(round(cin * t))                                # :138
            e = conv_bn(name + '_expand', cur, cin, cexp, 1, 1, 0, 1, 'relu6')        # :141-149
            d = conv_bn(name + '_dwise', e, cexp, cexp, 3, stride, 1, cexp, 'relu6')  # :155-164
            l = conv_bn(name + '_linear', d, cexp, c, 1, 1, 0, 1, None)               # :168-176
            if j > 0:                                                 # ifshortcut, :177-179,198-208
                dst = new()
                ops.append(('add', cur, l, dst, None))
                cur = dst
            else:
                cur = l
        in_c = c
    cur = conv_bn('conv9', cur, in_c, 1280, 1, 1, 0, 1, 'relu6')    # :74-81
    return ops, cur, 1280


def resnet_ops(depth=50):
    """ResNet-50/101 v1.5 bottleneck encoder WITHOUT pooling/fc head -- a BUILD-DEFINED
    EXTENSION for BASELINE.json configs 2-5; the reference only has MobileNetV2, so there
    are no reference numerics for it.  conv->BN->ReLU, stride on the 3x3, projection
    shortcut on the first block of each stage.  Output channels 2048, spatial S/32."""
    blocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}[depth]
    ops = []
    nid = [0]

    def new():
        nid[0] += 1
        return nid[0]

    def conv_bn(name, src, cin, cout, k, stride, pad, act):
        dst = new()
        ops.append(('conv_bn', name, src, dst, cin, cout, k, stride, pad, 1, act))
        return dst

    cur = conv_bn('res_conv1', 0, 3, 64, 7, 2, 3, 'relu')
    dst = new()
    ops.append(('maxpool', cur, dst))
    cur = dst
    cin = 64
    for si, nb in enumerate(blocks):
        w = 64 << si
        for b in range(nb):
            name = 'res%d_%d' % (si + 2, b + 1)
            stride = 2 if (b == 0 and si > 0) else 1
            a = conv_bn(name + '_branch2a', cur, cin, w, 1, 1, 0, 'relu')
            bb = conv_bn(name + '_branch2b', a, w, w, 3, stride, 1, 'relu')
            c = conv_bn(name + '_branch2c', bb, w, 4 * w, 1, 1, 0, None)
            if b == 0:
                sc = conv_bn(name + '_branch1', cur, cin, 4 * w, 1, stride, 0, None)
            else:
                sc = cur
            dst = new()
            ops.append(('add', sc, c, dst, 'relu'))
            cur = dst
            cin = 4 * w
    return ops, cur, 2048


def encoder_ops(kind):
    if kind == 'mobilenetv2':
        return mobilenet_v2_ops()
    if kind == 'resnet50':
        return resnet_ops(50)
    if kind == 'resnet101':
        return resnet_ops(101)
    raise ValueError('unknown encoder %r' % (kind,))
