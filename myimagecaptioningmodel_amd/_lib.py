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
                ('ysaved', ctypes.c_void_p), ('ld_saved', ctypes.c_int), ('dact', ctypes.c_int)]


def gemm_geom(rows, K, ldx=None):
    """Geometry of a plain row-major GEMM A[rows][K] (row stride ldx)."""
    return ConvGeom(rows, 1, 1, K, 1, 1, 1, 1, 1, 1, 0, K if ldx is None else ldx)


_p, _i, _f, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
_g = ctypes.POINTER(ConvGeom)

# name -> argument ctypes (the trailing `void* stream` included).  Must list every symbol of capmi.h.
SIGNATURES = {
    'capmi_igemm_nt': [_p, _p, _p, _g, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    'capmi_igemm_nt_bnred': [_p, _p, _p, _g, _i, _i, _i, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'capmi_igemm_nt_group': [ctypes.POINTER(NtCall), _i, _i, _p],
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
    'capmi_bn_inference_coef': [_p, _p, _p, _f, _p, _p, _i, _p],
    'capmi_bn_finalize_apply': [_p, _i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _i, _p, _p, _p, _i, _i, _p],
    'capmi_bn_bwd_reduce': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_bn_bwd_reduce_final': [_p, _i, _i, _p, _p],
    'capmi_bn_bwd_apply': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'capmi_add_act': [_p, _p, _p, _l, _i, _i, _p],
    'capmi_act_bwd': [_p, _p, _p, _i, _l, _i, _i, _p],
    'capmi_mean_rows': [_p, _p, _i, _i, _i, _i, _p],
    'capmi_mean_rows_bwd': [_p, _p, _i, _i, _i, _i, _p],
    'capmi_embedding_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_embedding_bwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bcast_rows': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_bcast_rows_bwd': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_lstm_cell_fwd': [_p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_lstm_cell_bwd': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'capmi_lstm_step_fwd': [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    'capmi_lstm_step_bwd': [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
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
    'capmi_cast': [_p, _p, _l, _i, _p],
    'capmi_weight_dgrad_form': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'capmi_weight_dgrad_form_batched': [_p, _p, _p, _i, _i, _p],
    'capmi_fill_f32': [_p, _f, _l, _p],
}


class CapmiError(RuntimeError):
    pass


# queries without a stream argument: name -> argument ctypes (return the part size, > 0)
QUERIES = {
    'capmi_igemm_nt_stats_part_rows': [_i, _i, _i, _i],
    'capmi_igemm_nt_bnred_part_rows': [_g, _i, _i],
    'capmi_bn_stats_part_rows': [_i, _i, _i],
    'capmi_lstm_step_supported': [_i, _i, _i],
    'capmi_bn_bwd_ws_floats': [_i, _i, _i],
}

# lane synchronisation (no stream-last convention): name -> argument ctypes
SYNC = {
    'capmi_event_create': [_p],          # void** event
    'capmi_event_destroy': [_p],
    'capmi_event_record': [_p, _p],
    'capmi_stream_wait_event': [_p, _p],
    'capmi_stream_create': [_p, _i],     # void** stream, priority
}

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
        L.capmi_igemm_tn_ws_bytes.argtypes = [_i, _i, _i, _i]
        L.capmi_igemm_tn_ws_bytes.restype = ctypes.c_longlong
        _lib = L
    return _lib


def last_error():
    return lib().capmi_last_error().decode()


def call(name, *args):
    """Calls one entry point (the stream is the last argument) and raises on a non-zero return."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise CapmiError('%s failed (%d): %s' % (name, rc, last_error()))


WGRAD_WS_BYTES = 32 << 20      # >= any capmi_igemm_tn_ws_bytes() result (partial slabs are capped at 24 MiB)
_wgrad_ws = {}


def wgrad_workspace(device):
    """One shared f32 scratch buffer per device for the weight-gradient partial slabs (launches on
    one stream run in order, so they can share it)."""
    import torch
    key = str(device)
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.zeros(WGRAD_WS_BYTES // 4, dtype=torch.float32, device=device)
    return _wgrad_ws[key]


class _SideCall:
    """A launch that may run on the plan's side stream (lane 1); calling it is calling the entry point."""
    lane = 1

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        return self.fn(*args)


class Plan:
    """A recorded launch sequence: (function, name, args-without-stream).  `run(stream)` replays it on
    a HIP stream; static shapes make the whole sequence capturable in a hipGraph.

    Lanes: a call added with lane=1 may run on a side stream, ordered against the main lane only by
    the `record(key, lane)` / `wait(key, lane)` entries around it.  The recorded order is always a valid
    sequential order, so replaying everything on one stream (side=False, or any consumer that just
    iterates `calls`) gives the same results."""

    _side = {}          # device index -> torch side stream

    def __init__(self):
        self.calls = []
        self._keep = []      # keeps ConvGeom structs / tensors alive
        self.has_lanes = False

    def add(self, name, *args, lane=0):
        fn = getattr(lib(), name)
        if lane:
            fn = _SideCall(fn)
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

    def run(self, stream, side=True):
        lanes = side and self.has_lanes and os.environ.get('CAPMI_LANES', '1') != '0'
        if not lanes:
            for fn, name, args in self.calls:
                if fn is not None and fn(*args, stream) != 0:
                    raise CapmiError('%s failed: %s' % (name, last_error()))
            return
        import torch
        L = lib()
        dev = torch.cuda.current_device()
        side = Plan._side.get(dev)
        if side is None:            # (stream, fork event, join event): lowest priority, it fills the main lane's gaps
            s, e0, e1 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            prio = int(os.environ.get('CAPMI_SIDE_PRIORITY', '-1'))
            if L.capmi_stream_create(ctypes.byref(s), prio) or L.capmi_event_create(ctypes.byref(e0)) or L.capmi_event_create(ctypes.byref(e1)):
                raise CapmiError('side lane: %s' % last_error())
            side = Plan._side[dev] = (s, e0, e1)
        s1, fork_ev, join_ev = side
        prog = getattr(self, '_prog', None)
        if prog is None or self._prog_key != (len(self.calls), stream, s1.value):
            # compiled form: ('launch', raw entry point, args + stream) / ('record' | 'wait', event, stream); waits on
            # keys recorded in an earlier run are dropped; stream pointers are baked in (the key checks them)
            prog, events = [], {}
            ptrs = (stream, s1)
            for fn, name, args in self.calls:
                if fn is None:
                    key, lane = args
                    if name == 'record':
                        ev = ctypes.c_void_p()
                        if L.capmi_event_create(ctypes.byref(ev)) != 0:
                            raise CapmiError('capmi_event_create: %s' % last_error())
                        events[key] = ev
                        prog.append((1, L.capmi_event_record, (ev, ptrs[lane])))
                    elif key in events:
                        prog.append((2, L.capmi_stream_wait_event, (ptrs[lane], events[key])))
                else:
                    lane = getattr(fn, 'lane', 0)
                    prog.append((0, getattr(fn, 'fn', fn), tuple(args) + (ptrs[lane],)))
            self._prog, self._prog_key = prog, (len(self.calls), stream, s1.value)
        L.capmi_event_record(fork_ev, stream)             # fork
        L.capmi_stream_wait_event(s1, fork_ev)
        for kind, fn, args in prog:
            if fn(*args) != 0:
                raise CapmiError('launch failed: %s' % last_error())
        L.capmi_event_record(join_ev, s1)                 # join
        L.capmi_stream_wait_event(stream, join_ev)

    def __len__(self):
        return len(self.calls)
