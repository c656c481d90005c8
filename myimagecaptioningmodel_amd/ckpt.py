"""Checkpoint directory layout of the reference: one file per persistable variable, file name =
variable name, each file a Paddle-1.x LoDTensor stream; plus the logger's resume JSON.

Mirrors /root/reference/ImageCaptioning/train.py:68-107 (`save_model` / `load_model`:
`fluid.io.save_persistables(exe, <ckpt>/checkpoint, train_prog)`, `load_persistables`) and
tools/logger.py:24-45 (`<log_path>/config` = {"epoch","best_bleu","best_meteor","train_encoder"}).

LoDTensor stream (Paddle 1.x `SerializeToStream`, restated from memory -- PaddlePaddle is not
installable here, so this format is UNVERIFIED against a real Paddle-written file; tests are
round-trip only):
    u32  lod version (0)
    u64  number of LoD levels (0 for every variable of this model)
    u32  tensor version (0)
    i32  size of the TensorDesc protobuf
    TensorDesc { required data_type (field 1, varint: INT64 = 3, FP32 = 5, FP64 = 6);
                 repeated int64 dims (field 2, one varint entry per dim, not packed) }
    raw little-endian data
Variables written: every parameter (reference names and layouts), BN running mean/variance, the
Adam accumulators `<param>_moment1_0`, `<param>_moment2_0`, `<param>_beta1_pow_acc_0`,
`<param>_beta2_pow_acc_0` (Paddle's naming convention, from memory), the global step counter
`@LR_DECAY_COUNTER@` (int64 [1], tools/util.py:47-51) and, for `cosine_decay_restart_warmup`, the
persistable `cur_epoch` (float32 [1], tools/util.py:94-95).

The number of steps taken is what Adam's bias correction and every LR schedule hang on.  It is read
back from `@LR_DECAY_COUNTER@` (exact).  The reference writes that variable only when a decay strategy
is configured; this writer always writes it (an extra file is ignored by `load_persistables`, which
loads the program's own variables).  A checkpoint without it (a reference checkpoint trained with
lr_decay_strategy=None) falls back to `beta2_pow_acc` = 0.999^(t+1) evaluated in float64 -- float32
0.9^(t+1) underflows near t = 980, 0.999^(t+1) stays a normal float32 up to t ~ 87 000 with a relative
rounding of 6e-8, i.e. the recovered t is exact below ~1 700 steps and within t * 6e-5 beyond.
"""
import json
import os
import struct

import numpy as np

from .optim import ADAM_BETA1, ADAM_BETA2

LR_COUNTER = '@LR_DECAY_COUNTER@'

_DTYPE_CODE = {np.dtype('int64'): 3, np.dtype('float32'): 5, np.dtype('float64'): 6}
_CODE_DTYPE = {v: k for k, v in _DTYPE_CODE.items()}


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def write_lod_tensor(path, arr):
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in _DTYPE_CODE:
        raise ValueError('unsupported dtype %s' % arr.dtype)
    desc = b'\x08' + _varint(_DTYPE_CODE[arr.dtype])
    for d in arr.shape:
        desc += b'\x10' + _varint(int(d))
    with open(path, 'wb') as f:
        f.write(struct.pack('<I', 0))
        f.write(struct.pack('<Q', 0))
        f.write(struct.pack('<I', 0))
        f.write(struct.pack('<i', len(desc)))
        f.write(desc)
        f.write(arr.astype(arr.dtype.newbyteorder('<'), copy=False).tobytes())


def read_lod_tensor(path):
    with open(path, 'rb') as f:
        buf = f.read()
    pos = 0
    (ver,) = struct.unpack_from('<I', buf, pos); pos += 4
    (levels,) = struct.unpack_from('<Q', buf, pos); pos += 8
    for _ in range(levels):                      # LoD levels: u64 byte size + data (unused by this model)
        (nbytes,) = struct.unpack_from('<Q', buf, pos); pos += 8 + nbytes
    (tver,) = struct.unpack_from('<I', buf, pos); pos += 4
    (dlen,) = struct.unpack_from('<i', buf, pos); pos += 4
    if ver != 0 or tver != 0:
        raise ValueError('%s: unsupported LoDTensor version %d/%d' % (path, ver, tver))
    end = pos + dlen
    dtype, dims = None, []
    while pos < end:
        tag, pos = _read_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _read_varint(buf, pos)
            if field == 1:
                dtype = _CODE_DTYPE[val]
            elif field == 2:
                dims.append(val)
        elif wire == 2:                          # packed dims
            ln, pos = _read_varint(buf, pos)
            stop = pos + ln
            while pos < stop:
                val, pos = _read_varint(buf, pos)
                dims.append(val)
        else:
            raise ValueError('%s: unexpected wire type %d' % (path, wire))
    n = int(np.prod(dims)) if dims else 1
    data = np.frombuffer(buf, dtype=dtype.newbyteorder('<'), count=n, offset=end)
    return data.reshape(dims).astype(dtype)


# ---------------------------------------------------------------------------- whole checkpoints
def save_persistables(engine, dirname):
    """`fluid.io.save_persistables(exe, dirname, train_prog)` (train.py:73): one file per variable."""
    os.makedirs(dirname, exist_ok=True)
    params = engine.export_reference_params()
    st = engine.store
    m = st.export_reference(st.adam_m)
    v = st.export_reference(st.adam_v)
    t = engine.step_count
    sched = engine.lr_schedule
    write_lod_tensor(os.path.join(dirname, LR_COUNTER), np.array([sched.counter_after(t)], np.int64))
    if sched.strategy == 'cosine_decay_restart_warmup':
        write_lod_tensor(os.path.join(dirname, 'cur_epoch'), np.array([sched.cur_epoch_after(t)], np.float32))
    for name, arr in params.items():
        write_lod_tensor(os.path.join(dirname, name), arr.astype(np.float32))
        if name in st.entries and st.entries[name].trainable:
            write_lod_tensor(os.path.join(dirname, name + '_moment1_0'), m[name].astype(np.float32))
            write_lod_tensor(os.path.join(dirname, name + '_moment2_0'), v[name].astype(np.float32))
            write_lod_tensor(os.path.join(dirname, name + '_beta1_pow_acc_0'), np.array([ADAM_BETA1 ** (t + 1)], np.float32))
            write_lod_tensor(os.path.join(dirname, name + '_beta2_pow_acc_0'), np.array([ADAM_BETA2 ** (t + 1)], np.float32))


def load_persistables(engine, dirname, strict=True):
    """`fluid.io.load_persistables` (train.py:103-104).  Restores parameters, BN running statistics,
    Adam moments and the number of steps taken (module docstring: `@LR_DECAY_COUNTER@`, else beta2_pow_acc)."""
    import torch
    from .params import to_kernel
    st = engine.store
    params, m, v = {}, {}, {}
    step, pow2 = None, None
    cpath = os.path.join(dirname, LR_COUNTER)
    if os.path.isfile(cpath):
        step = engine.lr_schedule.steps_from_counter(int(read_lod_tensor(cpath).ravel()[0]))
    for name in st.names():
        path = os.path.join(dirname, name)
        if not os.path.isfile(path):
            if strict:
                raise FileNotFoundError(path)
            continue
        params[name] = read_lod_tensor(path)
        for suffix, dst in (('_moment1_0', m), ('_moment2_0', v)):
            p2 = path + suffix
            if os.path.isfile(p2):
                dst[name] = read_lod_tensor(p2)
        p3 = path + '_beta2_pow_acc_0'
        if step is None and pow2 is None and os.path.isfile(p3):
            pow2 = float(read_lod_tensor(p3).ravel()[0])
    if step is None and pow2 is not None:
        if not (0.0 < pow2 < 1.0) or not np.isfinite(pow2):
            raise ValueError('%s: cannot recover the step count (no %s file and beta2_pow_acc = %r)' % (dirname, LR_COUNTER, pow2))
        step = int(round(np.log(np.float64(pow2)) / np.log(np.float64(ADAM_BETA2)))) - 1
    engine.load_reference_params(params)
    for src, buf in ((m, st.adam_m), (v, st.adam_v)):
        for name, arr in src.items():
            e = st.entries[name]
            st.view(name, buf).copy_(torch.from_numpy(to_kernel(arr.astype(np.float32), e.kind)))
    if step is not None:
        engine.step_count = max(0, step)
    cur = os.path.join(dirname, 'cur_epoch')
    if os.path.isfile(cur) and engine.lr_schedule.strategy == 'cosine_decay_restart_warmup':
        have, want = float(read_lod_tensor(cur).ravel()[0]), engine.lr_schedule.cur_epoch_after(engine.step_count)
        if have != want:       # derived from the counter here; a mismatch means batch_size / sample_count changed
            raise ValueError('%s: cur_epoch = %g but the step counter implies %g (step_each_epoch changed?)' % (dirname, have, want))


def save_params(engine, dirname):
    """`fluid.io.save_params` (train.py:78-79): the parameters only (no optimizer state, no BN running statistics
    -- in Paddle those are persistable non-Parameter variables)."""
    os.makedirs(dirname, exist_ok=True)
    params = engine.export_reference_params()
    for name in engine.store.entries:
        write_lod_tensor(os.path.join(dirname, name), params[name].astype(np.float32))


def load_vars_existing(engine, dirname):
    """`fluid.io.load_vars(exe, p, prog, predicate=util.get_predicate(p))` (train.py:97-99, util.py:122-138): loads
    every PARAMETER whose file exists under `dirname` (a pretrained-encoder directory), silently skipping the rest.
    Returns the names loaded."""
    names = predicate_existing(dirname, list(engine.store.entries))
    engine.load_reference_params({n: read_lod_tensor(os.path.join(dirname, n)) for n in names})
    return names


def predicate_existing(dirname, names):
    """`util.get_predicate` (tools/util.py:122-138): the variables whose file exists under dirname."""
    return [n for n in names if os.path.isfile(os.path.normpath(os.path.join(dirname, n)))]


# ---------------------------------------------------------------------------- logger resume state
def load_resume_state(log_path, train_encoder=True):
    """tools/logger.py:24-45: creates `<log_path>/config` on first use; is_first_init = (epoch == 1)."""
    os.makedirs(log_path, exist_ok=True)
    path = os.path.join(log_path, 'config')
    if not os.path.exists(path):
        conf = {'epoch': 1, 'best_bleu': 0, 'best_meteor': 0, 'train_encoder': train_encoder}
        save_resume_state(log_path, conf)
    else:
        with open(path, encoding='utf-8') as f:
            conf = json.loads(f.read())
    return conf


def save_resume_state(log_path, conf):
    with open(os.path.join(log_path, 'config'), 'w', encoding='utf-8') as f:
        f.write(json.dumps(conf))
