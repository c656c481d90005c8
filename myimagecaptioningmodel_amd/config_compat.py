"""Reads an UNCHANGED reference-style config (the module of dicts in
/root/reference/ImageCaptioning/config.py:2-73) into the flat hyper-parameter dict the engine
uses.  Keys consumed by the hot path: data.{start_idx,stop_idx,padding_idx,ImageShape,
sample_count}, train.{seed,learning_rate,lr_decay_strategy,decay_epoch,warmup_epoch,
gradient_clip,batch_size,max_epoch}, model.encoder.encoder_trainable,
model.decoder.{vocab_size,embedding_size,sentence_length,hidden_dim,infer_max_length}.
`encoder_dim` / `encoder_channel` (config.py:51-52) are declared by the reference but read
nowhere; they are ignored here too.
"""


def _get(obj, name):
    return obj[name] if isinstance(obj, dict) else getattr(obj, name)


def from_reference_config(config, **overrides):
    """`config`: a module (or dict) exposing `data`, `train`, `model` dicts like the reference's
    config.py.  Returns the engine cfg.  Build-defined extensions (encoder kind, attention mode,
    compute dtype) are not part of the reference schema and come in through **overrides."""
    data, train, model = _get(config, 'data'), _get(config, 'train'), _get(config, 'model')
    dec, enc = model['decoder'], model['encoder']
    shape = data.get('ImageShape', [224, 224])
    if shape[0] != shape[1]:
        raise ValueError('ImageShape must be square, got %r' % (shape,))
    cfg = dict(
        encoder='mobilenetv2',                       # the only encoder the reference has
        image_size=int(shape[0]),
        hidden=int(dec['hidden_dim']), embed=int(dec['embedding_size']), vocab=int(dec['vocab_size']),
        sentence_length=int(dec['sentence_length']), infer_max_length=int(dec['infer_max_length']),
        start_idx=int(data['start_idx']), stop_idx=int(data['stop_idx']), padding_idx=int(data['padding_idx']),
        encoder_trainable=bool(enc['encoder_trainable']),
        attention='singleton',                       # reference-faithful quirk Q1
        rnn_layer=1,                                 # model_adaAttention_aic.py:174 builds Decoder(..., rnn_layer=1)
        dtype='f32',                                 # the reference computes in fp32 throughout
        learning_rate=float(train['learning_rate']), lr_decay_strategy=train['lr_decay_strategy'],
        decay_epoch=train['decay_epoch'], warmup_epoch=train['warmup_epoch'], max_epoch=int(train['max_epoch']),
        gradient_clip=train['gradient_clip'], batch_size=int(train['batch_size']),
        sample_count=int(data.get('sample_count', 0)), seed=train.get('seed'),
    )
    cfg.update(overrides)
    return cfg


def default_cfg(**overrides):
    """Repo-default hyper-parameters (config.py:15,22-24,31-43,50,55-60) without needing the module."""
    cfg = dict(encoder='mobilenetv2', image_size=224, hidden=1024, embed=256, vocab=12295, sentence_length=35,
               infer_max_length=35, start_idx=2, stop_idx=3, padding_idx=0, encoder_trainable=True,
               attention='singleton', rnn_layer=1, dtype='f32', learning_rate=5e-5, lr_decay_strategy=None, decay_epoch=0,
               warmup_epoch=3, max_epoch=10, gradient_clip=False, batch_size=128, sample_count=944996, seed=None)
    cfg.update(overrides)
    return cfg
