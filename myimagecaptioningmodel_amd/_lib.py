"""ctypes binding of libcapmi.so (the C ABI declared in include/capmi.h).

The library is the product path: if it is missing or a symbol is absent this module raises --
there is no CPU/eager fallback anywhere in the package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CAPMI_LIB') or os.path.join(_HERE, 'libcapmi.so')

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
DACT_BITMASK = 0x100      # capmi.h CAPMI_DACT_BITMASK
ACT_CODES = {None: ACT_NONE, 'relu': ACT_RELU, 'relu6': ACT_RELU6, 'tanh': ACT_TANH, 'sigmoid': ACT_SIGMOID}


class ConvGeom(ctypes.Structure):
    """capmi_conv_geom (include/capmi.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ('B', 'Hi', 'Wi', 'Cin', 'Ho', 'Wo', 'kh', 'kw', 'sd', 'up', 'pad', 'ldx', 'os', 'oh0', 'ow0', 'Hof', 'Wof')]


class NtCall(ctypes.Structure):
    """capmi_igemm_nt_call (include/capmi.h)."""
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p), ('g', ConvGeom),
                ('N', ctypes.c_int), ('ldw', ctypes.c_int), ('ldy', ctypes.c_int),
                ('addend', ctypes.c_void_p), ('ld_addend', ctypes.c_int),
                ('ysaved', ctypes.c_void_p), ('ld_saved', ctypes.c_int), ('dact', ctypes.c_int),
                ('bias', ctypes.c_void_p), ('act', ctypes.c_int)]


def gemm_geom(rows, K, ldx=None):
    """Geometry of a plain row-major GEMM A[rows][K] (row stride ldx)."""
    return ConvGeom(rows, 1, 1, K, 1, 1, 1, 1, 1, 1, 0, K if ldx is None else ldx)


_p, _i, _f, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
_g = ctypes.POINTER(ConvGeom)

# name -> argument ctypes (the trailing `void* stream` included).  Must list every symbol of capmi.h.
SIGNATURES = {
    'capmi_igemm_nt': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    'capmi_igemm_nt_bn': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_igemm_nt_bnact': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p],
    'capmi_igemm_nt_bnfin': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _i, _p],
    'capmi_igemm_nt_bnred': [_p, _p, _p, _g, _i, _i, _i, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'capmi_igemm_nt_stat': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _p, _i, _p],
    'capmi_bn_stat_apply': [_p, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p],
    'capmi_bn_stat_apply_pool': [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p],
    'capmi_igemm_nt_bnsum': [_p, _p, _p, _g, _i, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    'capmi_igemm_nt_group': [ctypes.POINTER(NtCall), _i, _i, _p],
    'capmi_igemm_nt_splitk': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, ctypes.c_longlong, _i, _p],
    'capmi_igemm_tn_wgrad': [_p, _p, _p, _g, _i, _i, _i, _p, ctypes.c_longlong, _i, _p],
    'capmi_colsum': [_p, _i, _i, _i, _p, _i, _p],
    'capmi_im2col_stem': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_s2d_stem': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_s2d_stem_mask_grad': [_p, _i, _i, _i, _i, _p],
    'capmi_dwconv3x3_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_dwconv3x3_bwd_data': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_dwconv3x3_bwd_weight': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_maxpool3x3s2_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_maxpool3x3s2_bwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bn_stats': [_p, _i, _i, _p, _i, _p],
    'capmi_bn_finalize': [_p, _i, _i, _i, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p],
    'capmi_bn_apply': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_bn_apply_mask': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_bn_inference_coef': [_p, _p, _p, _f, _p, _p, _i, _p],
    'capmi_bn_inference_coef_batched': [_p, _i, _i, _f, _p],
    'capmi_bn_finalize_apply': [_p, _i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _i, _p, _p, _p, _i, _i, _p],
    'capmi_bn_bwd_reduce': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_reduce_final': [_p, _i, _i, _p, _p],
    'capmi_bn_bwd_apply': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_reduce_spread': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_apply_spread': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_reduce_pool': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_reduce_pool_x': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_apply_pool': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_apply_pool_x': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_add_act': [_p, _p, _p, _l, _i, _i, _p],
    'capmi_act_bwd': [_p, _p, _p, _i, _l, _i, _i, _p],
    'capmi_mean_rows': [_p, _p, _i, _i, _i, _i, _p],
    'capmi_mean_rows_bwd': [_p, _p, _i, _i, _i, _i, _p],
    'capmi_embedding_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_caption_feed': [_p, _p, _p, _i, _i, _p],
    'capmi_embedding_bwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bcast_rows': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bcast_rows_bwd': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_lstm_cell_fwd': [_p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_decode_prep': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'capmi_lstm_cell_sentinel_fwd': [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_lstm_cell_bwd': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_lstm_step_fwd': [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_lstm_step_bwd': [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_lstm_seq_fwd': [_p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p],
    'capmi_lstm_seq_bwd': [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p],
    'capmi_beam_step': [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'capmi_gather_rows': [_p, _p, _p, _i, _i, _i, _p],
    'capmi_beam_backtrack': [_p, _p, _p, _i, _i, _i, _p],
    'capmi_sentinel_fwd': [_p, _p, _p, _l, _i, _p],
    'capmi_sentinel_bwd': [_p, _p, _p, _p, _p, _l, _i, _p],
    'capmi_ada_attention_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_ada_attention_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                _i, _i, _i, _i, _i, _i, _p],
    'capmi_softmax_xent_fwd': [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_xent_finalize': [_p, _p, _p, _p, _i, _i, _p],
    'capmi_softmax_xent_bwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_argmax': [_p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_adam': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p],
    'capmi_adam_shadow': [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p],
    'capmi_adam_g16': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p],
    'capmi_cast': [_p, _p, _l, _i, _p],
    'capmi_weight_dgrad_form': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_weight_dgrad_form_batched': [_p, _p, _p, _i, _i, _p],
    'capmi_fill_f32': [_p, _f, _l, _p],
    'capmi_allreduce_bucket': [_p, _p, _l, _p],
    'capmi_allreduce_bucket_bf16': [_p, _p, _l, _p],
}


class CapmiError(RuntimeError):
    pass


# queries without a stream argument: name -> argument ctypes (return the part size, > 0)
QUERIES = {
    'capmi_igemm_nt_stats_part_rows': [_i, _i, _i, _i],
    'capmi_igemm_nt_bnred_part_rows': [_g, _i, _i],
    'capmi_igemm_nt_bnsum_part_rows': [_g, _i, _i],
    'capmi_igemm_nt_stat_supported': [_g, _i, _i],
    'capmi_igemm_nt_bnact_supported': [_g, _i, _i],
    'capmi_bn_stats_part_rows': [_i, _i, _i],
    'capmi_lstm_step_supported': [_i, _i, _i],
    'capmi_lstm_seq_supported': [_i, _i, _i, _i],
    'capmi_bn_bwd_ws_floats': [_i, _i, _i],
    'capmi_deterministic': [],
    'capmi_set_deterministic': [_i],
    'capmi_general_epilogue': [],
    'capmi_set_general_epilogue': [_i],
    'capmi_kernel_probe_begin': [],
    'capmi_kernel_probe_end': [_p, _i, _p, _p, _p],
}

# lane synchronisation (no stream-last convention): name -> argument ctypes
SYNC = {
    'capmi_event_create': [_p],          # void** event
    'capmi_event_create_timed': [_p],
    'capmi_event_elapsed_ms': [_p, _p, _p],
    'capmi_event_destroy': [_p],
    'capmi_event_record': [_p, _p],
    'capmi_stream_wait_event': [_p, _p],
    'capmi_stream_create': [_p, _i],     # void** stream, priority
}

# communicator management (no stream): name -> argument ctypes
COMM = {
    'capmi_comm_unique_id': [_p],
    'capmi_comm_init': [_p, _i, _i, _p],      # void** comm, nranks, rank, id
    'capmi_comm_count': [_p, _p, _p],
    'capmi_comm_destroy': [_p],
}
COMM_ID_BYTES = 128

NLANES = 4                # lanes of a plan: 0 dependency chain, 1 weight gradients, 2 communication + optimizer, 3 projection-shortcut backward
PLAN_MAX_ARGS = 32
PLAN_LAUNCH, PLAN_RECORD, PLAN_WAIT = 0, 1, 2


class Launch(ctypes.Structure):
    """capmi_launch (include/capmi.h): one row of a packed launch table."""
    _fields_ = [('kind', ctypes.c_int32), ('lane', ctypes.c_int32), ('entry', ctypes.c_int32), ('nargs', ctypes.c_int32),
                ('args', ctypes.c_uint64 * PLAN_MAX_ARGS)]


PLAN_ENTRIES = {}       # entry-point name -> index in the library's plan entry table (filled by lib())

_lib = None


def lib():
    """Loads libcapmi.so once; raises CapmiError when the extension is missing."""
    global _lib
    if _lib is None:
        # torch first: libcapmi.so must bind to the HIP runtime torch brings (one runtime per process)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise CapmiError('libcapmi.so not found at %s -- run __graft_entry__.build() '
                             '(myimagecaptioningmodel_amd/csrc/build.sh); there is no fallback path' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.capmi_version.restype = ctypes.c_int
        L.capmi_last_error.restype = ctypes.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = ctypes.c_int
        for name, args in QUERIES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        for name, args in SYNC.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        for name, args in COMM.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        L.capmi_igemm_tn_ws_bytes.argtypes = [_i, _i, _i, _i]
        L.capmi_igemm_tn_ws_bytes.restype = ctypes.c_longlong
        L.capmi_igemm_nt_splitk_ws_bytes.argtypes = [_i, _i, _i, _i]
        L.capmi_igemm_nt_splitk_ws_bytes.restype = ctypes.c_longlong
        L.capmi_plan_entry_count.restype = ctypes.c_int
        L.capmi_plan_entry_name.argtypes = [_i]
        L.capmi_plan_entry_name.restype = ctypes.c_char_p
        L.capmi_plan_entry_nargs.argtypes = [_i]
        L.capmi_plan_entry_nargs.restype = ctypes.c_int
        L.capmi_plan_run.argtypes = [ctypes.POINTER(Launch), _i, ctypes.POINTER(ctypes.c_void_p), _i]
        L.capmi_plan_run.restype = ctypes.c_int
        for i in range(L.capmi_plan_entry_count()):
            name = L.capmi_plan_entry_name(i).decode()
            if L.capmi_plan_entry_nargs(i) != len(SIGNATURES[name]) - 1:
                raise CapmiError('%s: the plan entry table and the binding disagree on its arity' % name)
            PLAN_ENTRIES[name] = i
        _lib = L
    return _lib


def last_error():
    return lib().capmi_last_error().decode()


def set_deterministic(on):
    """capmi_set_deterministic (include/capmi.h): fixed-order reductions in place of every f32 atomic accumulation, so that
    runs of one launch sequence are bit-identical whatever the lanes' timing.  Returns the previous setting."""
    L = lib()
    prev = bool(L.capmi_deterministic())
    L.capmi_set_deterministic(1 if on else 0)
    return prev


def set_general_epilogue(on):
    """capmi_set_general_epilogue (include/capmi.h): every NT launch on the general epilogue instantiation instead of its
    epilogue class (verification: the classes must be bit-identical to it).  Returns the previous setting."""
    L = lib()
    prev = bool(L.capmi_general_epilogue())
    L.capmi_set_general_epilogue(1 if on else 0)
    return prev


def call(name, *args):
    """Calls one entry point (the stream is the last argument) and raises on a non-zero return."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise CapmiError('%s failed (%d): %s' % (name, rc, last_error()))


def probe_kernel(name, *args):
    """capmi_kernel_probe_begin / _end around one capmi_igemm_* (or capmi_lstm_*) call: nothing is launched; returns
    (symbol as rocprofv3 prints it, workgroups, block size, kernels the call launches)."""
    if not name.startswith(('capmi_igemm_', 'capmi_lstm_')):
        raise CapmiError('probe_kernel: %s is not a probed entry point' % name)
    L = lib()
    buf = ctypes.create_string_buffer(512)
    grid, block, n = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    L.capmi_kernel_probe_begin()
    rc = getattr(L, name)(*[a.value if isinstance(a, PtrSlot) else a for a in args], None)
    rc2 = L.capmi_kernel_probe_end(buf, 512, ctypes.byref(grid), ctypes.byref(block), ctypes.byref(n))
    if rc != 0 or rc2 != 0:
        raise CapmiError('probe of %s failed: %s' % (name, last_error()))
    return buf.value.decode(), grid.value, block.value, n.value


WGRAD_WS_BYTES = 32 << 20      # >= any capmi_igemm_tn_ws_bytes() result (partial slabs are capped at 24 MiB)
_wgrad_ws = {}


def wgrad_workspace(device, lane=1):
    """One f32 scratch buffer per (device, lane) for the weight-gradient partial slabs: launches on one stream run
    in order and can share it; launches on different lanes never do."""
    import torch
    key = (str(device), int(lane))
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.zeros(WGRAD_WS_BYTES // 4, dtype=torch.float32, device=device)
    return _wgrad_ws[key]


class _SideCall:
    """A launch that may run on another stream than the main lane; calling it is calling the entry point."""

    def __init__(self, fn, lane):
        self.fn, self.lane = fn, lane

    def __call__(self, *args):
        return self.fn(*args)


def _float_bits(x):
    import struct
    return struct.unpack('<I', struct.pack('<f', float(x)))[0]


class PtrSlot:
    """A device pointer argument that is re-read before every run of a plan (like the ctypes.c_float of the Adam step size):
    the feed tensor of a step can then be the caller's own device tensor instead of a copy of it."""

    def __init__(self, value=0):
        self.value = int(value)


def _patch_bits(src):
    return _float_bits(src.value) if isinstance(src, ctypes.c_float) else int(src.value) & 0xFFFFFFFFFFFFFFFF


def _slot(value, ctype):
    """One argument as the 8-byte slot capmi_plan_run expects; (bits, patch source or None)."""
    if ctype is _f:
        if isinstance(value, ctypes.c_float):          # re-read before every run (the per-step Adam step size)
            return _float_bits(value.value), value
        return _float_bits(value), None
    if value is None:
        return 0, None
    if isinstance(value, PtrSlot):
        return int(value.value) & 0xFFFFFFFFFFFFFFFF, value
    if isinstance(value, (ctypes.Structure, ctypes.Array)):
        return ctypes.addressof(value), None            # host struct kept alive by Plan._keep
    if isinstance(value, ctypes.c_void_p):
        return value.value or 0, None
    return int(value) & 0xFFFFFFFFFFFFFFFF, None


_SKIP = set(filter(None, os.environ.get('CAPMI_SKIP', '').split(',')))


class Plan:
    """A recorded launch sequence: (function, name, args-without-stream).  `run(stream)` replays it through
    capmi_plan_run -- the whole sequence in ONE foreign call -- on up to three HIP streams; static shapes make a
    single-lane sequence capturable in a hipGraph.

    Lanes: a call added with lane=1 (weight gradients and other work off the dependency chain) or lane=2
    (communication + optimizer of the data-parallel step) may run on its own stream, ordered against the others only by
    the `record(key, lane)` / `wait(key, lane)` entries around it.  The recorded order is always a valid sequential
    order, so replaying everything on one stream (side=False, or any consumer that just iterates `calls`) gives the
    same results.  CAPMI_PY_PLAN=1 walks the table from Python, one foreign call per launch (the round-1 host path;
    kept for the equivalence test and for debugging a single launch)."""

    _side = {}          # device index -> {'streams': {lane: stream}, 'fork': event, 'join': {lane: event}}
    _timed_events = []  # pool of timing events (run_timed)

    def __init__(self):
        self.calls = []
        self._keep = []      # keeps ConvGeom structs / tensors alive
        self.has_lanes = False
        self._compiled = {}

    def add(self, name, *args, lane=0):
        if name in _SKIP:           # CAPMI_SKIP=<entry>,<entry>: timing experiments only (results are wrong)
            return
        fn = getattr(lib(), name)
        if lane:
            fn = _SideCall(fn, lane)
            self.has_lanes = True
        self.calls.append((fn, name, args))
        self._keep.append(args)

    def record(self, key, lane):
        """Mark a point of `lane` that another lane can wait for."""
        self.calls.append((None, 'record', (key, lane)))
        self._keep.append(None)

    def wait(self, key, lane):
        """`lane` does not proceed before the point recorded under `key` (no-op if it was recorded in an
        earlier run: runs end with the lanes joined)."""
        self.calls.append((None, 'wait', (key, lane)))
        self._keep.append(None)

    def extend(self, other):
        self.calls.extend(other.calls)
        self._keep.extend(other._keep)
        self.has_lanes = self.has_lanes or other.has_lanes

    def launches(self):
        """The kernel launches only: (fn, name, args), synchronisation entries skipped."""
        return [c for c in self.calls if c[0] is not None]

    # ------------------------------------------------------------------ lanes of a device
    @staticmethod
    def _lane_streams(lanes_needed):
        import torch
        L = lib()
        dev = torch.cuda.current_device()
        side = Plan._side.setdefault(dev, dict(streams={}, fork=None, join={}))
        if side['fork'] is None:
            ev = ctypes.c_void_p()
            if L.capmi_event_create(ctypes.byref(ev)):
                raise CapmiError('lanes: %s' % last_error())
            side['fork'] = ev
        for lane in lanes_needed:
            if lane and lane not in side['streams']:
                # lane 1 (weight gradients): lowest priority, it fills the main lane's gaps; lane 2 (all-reduce +
                # optimizer): CAPMI_COMM_PRIORITY (default 0)
                s, ev = ctypes.c_void_p(), ctypes.c_void_p()
                prio = (int(os.environ.get('CAPMI_SIDE_PRIORITY', '-1')) if lane == 1 else int(os.environ.get('CAPMI_COMM_PRIORITY', '0')) if lane == 2
                        else int(os.environ.get('CAPMI_SHORTCUT_PRIORITY', '0')))
                if L.capmi_stream_create(ctypes.byref(s), prio) or L.capmi_event_create(ctypes.byref(ev)):
                    raise CapmiError('lane %d: %s' % (lane, last_error()))
                side['streams'][lane], side['join'][lane] = s, ev
        return side

    def _compile(self, lanes):
        """The packed capmi_launch table of this plan: (rows, n, patches, lanes used).  lanes=False drops the
        synchronisation entries and maps every call to lane 0."""
        L = lib()
        rows, patches, events, used = [], [], {}, {0}
        side = Plan._lane_streams({getattr(fn, 'lane', 0) for fn, _, _ in self.calls if fn is not None}) if lanes else None

        def sync_row(kind, ev, lane):
            r = Launch()
            r.kind, r.lane, r.entry, r.nargs = kind, lane, -1, 1
            r.args[0] = ev.value
            return r
        body = []
        for fn, name, args in self.calls:
            if fn is None:
                key, lane = args
                if not lanes:
                    continue
                if name == 'record':
                    ev = ctypes.c_void_p()
                    if L.capmi_event_create(ctypes.byref(ev)) != 0:
                        raise CapmiError('capmi_event_create: %s' % last_error())
                    events[key] = ev
                    body.append(sync_row(PLAN_RECORD, ev, lane))
                elif key in events:             # waits on keys recorded in an earlier run are dropped
                    body.append(sync_row(PLAN_WAIT, events[key], lane))
                used.add(lane)
                continue
            types = SIGNATURES[name]
            if len(args) != len(types) - 1:
                raise CapmiError('%s: %d arguments recorded, the entry point takes %d (+ stream)' % (name, len(args), len(types) - 1))
            r = Launch()
            r.kind, r.entry, r.nargs = PLAN_LAUNCH, PLAN_ENTRIES[name], len(args)
            r.lane = getattr(fn, 'lane', 0) if lanes else 0
            used.add(r.lane)
            for i, (a, t) in enumerate(zip(args, types)):
                r.args[i], src = _slot(a, t)
                if src is not None:
                    patches.append((len(body), i, src))
            body.append(r)
        if lanes:       # fork: every other lane starts behind the main lane's current position; join: the main lane waits for them
            head = [sync_row(PLAN_RECORD, side['fork'], 0)] + [sync_row(PLAN_WAIT, side['fork'], l) for l in sorted(used - {0})]
            tail = []
            for l in sorted(used - {0}):
                tail += [sync_row(PLAN_RECORD, side['join'][l], l), sync_row(PLAN_WAIT, side['join'][l], 0)]
            patches = [(r + len(head), i, src) for r, i, src in patches]
            body = head + body + tail
        table = (Launch * max(1, len(body)))(*body)
        return dict(table=table, n=len(body), patches=patches, events=events, used=sorted(used), side=side,
                    tail_len=2 * len(used - {0}) if lanes else 0)

    def run(self, stream, side=True, tail_events=None):
        """tail_events: {lane: event of capmi_event_create_timed} recorded at the END of each lane's work, in front of the final
        join (measurements only: the table goes out as two calls)."""
        lanes = bool(side and self.has_lanes and os.environ.get('CAPMI_LANES', '1') != '0')
        if os.environ.get('CAPMI_PY_PLAN', '0') == '1' and tail_events is None:
            return self._run_py(stream, lanes)
        key = (lanes, len(self.calls))
        c = self._compiled.get(key)
        if c is None:
            c = self._compiled[key] = self._compile(lanes)
        for row, slot, src in c['patches']:
            c['table'][row].args[slot] = _patch_bits(src)
        streams = (ctypes.c_void_p * NLANES)(*([stream] * NLANES))
        if lanes:
            for l, s in c['side']['streams'].items():
                streams[l] = s.value
        if tail_events is not None and lanes:
            k = c['n'] - c['tail_len']
            if lib().capmi_plan_run(c['table'], k, streams, NLANES) != 0:
                raise CapmiError('plan: %s' % last_error())
            for lane, ev in tail_events.items():
                lib().capmi_event_record(ev, streams[lane])
            tail = (Launch * max(1, c['tail_len']))(*[c['table'][i] for i in range(k, c['n'])])
            if c['tail_len'] and lib().capmi_plan_run(tail, c['tail_len'], streams, NLANES) != 0:
                raise CapmiError('plan: %s' % last_error())
            return
        if lib().capmi_plan_run(c['table'], c['n'], streams, NLANES) != 0:
            raise CapmiError('plan: %s' % last_error())

    def _run_py(self, stream, lanes):
        """The same sequence, one foreign call per launch."""
        L = lib()
        if not lanes:
            for fn, name, args in self.calls:
                if fn is not None and fn(*[a.value if isinstance(a, PtrSlot) else a for a in args], stream) != 0:
                    raise CapmiError('%s failed: %s' % (name, last_error()))
            return
        c = self._compiled.get(('py', len(self.calls)))
        if c is None:
            c = self._compiled[('py', len(self.calls))] = self._compile(True)
        ptrs = {0: stream}
        ptrs.update({l: s.value for l, s in c['side']['streams'].items()})
        others = [l for l in c['used'] if l]
        L.capmi_event_record(c['side']['fork'], stream)
        for l in others:
            L.capmi_stream_wait_event(ptrs[l], c['side']['fork'])
        events = c['events']
        seen = set()
        for fn, name, args in self.calls:
            if fn is None:
                key, lane = args
                if name == 'record':
                    seen.add(key)
                    rc = L.capmi_event_record(events[key], ptrs[lane])
                elif key in seen:
                    rc = L.capmi_stream_wait_event(ptrs[lane], events[key])
                else:
                    rc = 0
            else:
                rc = getattr(fn, 'fn', fn)(*[a.value if isinstance(a, PtrSlot) else a for a in args], ptrs[getattr(fn, 'lane', 0)])
            if rc != 0:
                raise CapmiError('%s failed: %s' % (name, last_error()))
        for l in others:
            L.capmi_event_record(c['side']['join'][l], ptrs[l])
            L.capmi_stream_wait_event(stream, c['side']['join'][l])

    def run_timed(self, stream):
        """One eager run of the plan ON ITS LANES with a pair of HIP timing events around every launch, recorded on the
        stream the launch goes to (`torch.cuda.Event` records on a torch stream object: the lanes are wrapped as
        ExternalStream).  Returns [(name, args, lane, milliseconds)] in launch order -- the IN-MODEL duration of every kernel
        (side-lane kernels beside the main lane's and the other way round), which is what `rocprofv3 --kernel-trace` reports
        for the same step; a single-stream replay times every kernel with the chip to itself.  The event records cost the
        queues a few microseconds per launch: the step as a whole runs ~10 % slower under it, the kernels themselves do not."""
        import torch
        L = lib()
        lanes = bool(self.has_lanes and os.environ.get('CAPMI_LANES', '1') != '0')
        c = self._compiled.get(('py', len(self.calls)))
        if c is None:
            c = self._compiled[('py', len(self.calls))] = self._compile(True)
        ptrs = {0: stream}
        if lanes:
            ptrs.update({l: s.value for l, s in c['side']['streams'].items()})
        pool = Plan._timed_events

        def timed_event():
            if pool:
                return pool.pop()
            ev = ctypes.c_void_p()
            if L.capmi_event_create_timed(ctypes.byref(ev)) != 0:
                raise CapmiError('capmi_event_create_timed: %s' % last_error())
            return ev
        others = [l for l in c['used'] if l] if lanes else []
        L.capmi_event_record(c['side']['fork'], stream)
        for l in others:
            L.capmi_stream_wait_event(ptrs[l], c['side']['fork'])
        events, seen, timed = c['events'], set(), []
        for fn, name, args in self.calls:
            if fn is None:
                key, lane = args
                if not lanes:
                    continue
                if name == 'record':
                    seen.add(key)
                    rc = L.capmi_event_record(events[key], ptrs[lane])
                else:
                    rc = L.capmi_stream_wait_event(ptrs[lane], events[key]) if key in seen else 0
            else:
                lane = getattr(fn, 'lane', 0) if lanes else 0
                a, b = timed_event(), timed_event()
                L.capmi_event_record(a, ptrs[lane])
                rc = getattr(fn, 'fn', fn)(*[x.value if isinstance(x, PtrSlot) else x for x in args], ptrs[lane])
                L.capmi_event_record(b, ptrs[lane])
                timed.append((name, args, lane, a, b))
            if rc != 0:
                raise CapmiError('%s failed: %s' % (name, last_error()))
        for l in others:
            L.capmi_event_record(c['side']['join'][l], ptrs[l])
            L.capmi_stream_wait_event(stream, c['side']['join'][l])
        torch.cuda.synchronize()
        out, ms = [], ctypes.c_float(0.0)
        for name, args, lane, a, b in timed:
            if L.capmi_event_elapsed_ms(a, b, ctypes.byref(ms)) != 0:
                raise CapmiError('capmi_event_elapsed_ms: %s' % last_error())
            out.append((name, args, lane, float(ms.value)))
            pool.extend((a, b))
        return out

    def __len__(self):
        return len(self.calls)
