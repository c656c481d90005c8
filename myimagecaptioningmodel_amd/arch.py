"""Encoder topologies as flat op lists (host-side mirror of the reference encoder definition).

`mobilenet_v2()` restates the layer table of the reference encoder
(/root/reference/ImageCaptioning/model/MobileNetV2.py:31-86 `net`, :128-181
`inverted_residual_unit`, :183-209 `invresi_blocks`) with the reference's variable names;
`resnet(depth)` is the build-defined extension BASELINE.json's throughput configs ask for.
"""
from collections import namedtuple

# dst = act(bn(conv(src)))            name = reference variable stem (`<name>_weights`, `<name>_bn_*`)
ConvBN = namedtuple('ConvBN', 'name src dst cin cout k stride pad groups act')
# dst = act(a + b)                    MobileNetV2.py:123-124 shortcut / ResNet block output
Add = namedtuple('Add', 'a b dst act')
# dst = maxpool3x3/s2/p1(src)         ResNet stem only
MaxPool = namedtuple('MaxPool', 'src dst')

Encoder = namedtuple('Encoder', 'ops out channels reduction')   # reduction: S -> S/reduction


class _Builder:
    def __init__(self):
        self.ops = []
        self.n = 0

    def tensor(self):
        self.n += 1
        return self.n

    def conv(self, name, src, cin, cout, k, stride, pad, groups, act):
        dst = self.tensor()
        self.ops.append(ConvBN(name, src, dst, cin, cout, k, stride, pad, groups, act))
        return dst

    def add(self, a, b, act):
        dst = self.tensor()
        self.ops.append(Add(a, b, dst, act))
        return dst


def mobilenet_v2():
    b = _Builder()
    x = b.conv('conv1_1', 0, 3, 32, 3, 2, 1, 1, 'relu6')                      # MobileNetV2.py:49-56
    cin = 32
    table = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]                     # :37-45 (t, c, n, s)
    for stage, (t, c, n, s) in enumerate(table, start=2):
        for unit in range(1, n + 1):
            name = 'conv%d_%d' % (stage, unit)
            width = int(round(cin * t))                                         # :138
            e = b.conv(name + '_expand', x, cin, width, 1, 1, 0, 1, 'relu6')   # :141-149
            d = b.conv(name + '_dwise', e, width, width, 3, s if unit == 1 else 1, 1, width, 'relu6')  # :155-164
            y = b.conv(name + '_linear', d, width, c, 1, 1, 0, 1, None)        # :168-176
            x = b.add(x, y, None) if unit > 1 else y                            # :177-179, :198-208
            cin = c
    x = b.conv('conv9', x, cin, 1280, 1, 1, 0, 1, 'relu6')                      # :74-81
    return Encoder(b.ops, x, 1280, 32)


def resnet(depth):
    """ResNet-v1.5 bottleneck trunk (no pooling/fc head): BUILD-DEFINED, not in the reference."""
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    b = _Builder()
    x = b.conv('res_conv1', 0, 3, 64, 7, 2, 3, 1, 'relu')
    pooled = b.tensor()
    b.ops.append(MaxPool(x, pooled))
    x, cin = pooled, 64
    for si, nb in enumerate(blocks):
        w = 64 << si
        for bi in range(nb):
            name = 'res%d_%d' % (si + 2, bi + 1)
            stride = 2 if (bi == 0 and si > 0) else 1
            # the projection shortcut is emitted FIRST: the add is folded into branch2c's bn_apply
            # (encoder.py), which must find the shortcut already computed -- and, in backward,
            # must have written the shortcut's gradient before branch1's backward reads it
            sc = b.conv(name + '_branch1', x, cin, 4 * w, 1, stride, 0, 1, None) if bi == 0 else x
            a = b.conv(name + '_branch2a', x, cin, w, 1, 1, 0, 1, 'relu')
            m = b.conv(name + '_branch2b', a, w, w, 3, stride, 1, 1, 'relu')
            c = b.conv(name + '_branch2c', m, w, 4 * w, 1, 1, 0, 1, None)
            x = b.add(sc, c, 'relu')
            cin = 4 * w
    return Encoder(b.ops, x, 2048, 32)


def encoder(kind):
    if kind == 'mobilenetv2':
        return mobilenet_v2()
    if kind == 'resnet50':
        return resnet(50)
    if kind == 'resnet101':
        return resnet(101)
    raise ValueError('不支持{}'.format(kind))
