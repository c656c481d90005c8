"""bf16 engine against the f64 oracle, per ENCODER tensor: cosine and norm ratio of every conv / batch-norm gradient of one
train step (ResNet-50 and MobileNetV2 at 128x128, batch 8 -- the sizes of test_bf16_train_step_close_to_oracle), next to the
f32 engine's.  Shows how far down the encoder an end-to-end bf16 gradient comparison is meaningful at random initialisation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model as om
from tests.test_gpu_model import _cfgs, _data, _engine
for encoder in ('resnet50', 'mobilenetv2'):
    ocfg, ecfg = _cfgs(encoder, 'slots', 'bf16', S=128, H=64, E=32, V=100)
    B = 8
    params, image, caption = _data(ocfg, B, seed=4)
    oracle = om.OracleModel(ocfg, {k: v.copy() for k, v in params.items()})
    oracle.forward_train(image.astype(np.float64), caption)
    go = oracle.backward()
    res = {}
    for dt in ('bf16', 'f32'):
        e = dict(ecfg, dtype=dt)
        eng = _engine(e, params)
        eng.forward_backward(image, caption)
        res[dt] = eng.export_reference_grads()
    names = [n for n in go if n.endswith('_weights')]
    print(encoder, 'conv weight gradients, first layer -> last layer: cos(bf16, oracle)  |bf16|/|oracle|   cos(f32, oracle)')
    for n in names:
        a, b, c = res['bf16'][n].ravel(), go[n].ravel(), res['f32'][n].ravel()
        cos = lambda x, y: float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30))
        print('  %-34s %.4f  %.3f   %.6f' % (n, cos(a, b), np.linalg.norm(a) / (np.linalg.norm(b) + 1e-30), cos(c, b)))
