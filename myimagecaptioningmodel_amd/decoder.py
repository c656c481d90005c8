"""Decoder executor: bridge + adaptive-attention LSTM decoder + tied vocabulary projection +
masked cross-entropy, forward and backward, plus the greedy decode loop -- as launch plans.

Semantics follow /root/reference/ImageCaptioning/model/model_adaAttention_aic.py:
  `_img2feature` :191-199, `Decoder.call` :50-135, `weight_tying_fc` :15-25,
  `training_network` :161-183, `loss` :205-212, `eval_network` :185-189.

MI355X-first restructuring (same arithmetic, different schedule): in train mode the only
recurrence is the LSTM (x_t depends on the given caption and the image only), so every
projection except h_{t-1}.W_h runs ONCE over all T steps as an M = T*B row GEMM; rows are
time-major (m = t*B + b) so a timestep is a contiguous row block.  Per step only the
[B,H]x[H,4H] recurrent GEMM (accumulated onto the precomputed input gates in its epilogue)
and the pointwise cell remain.  BPTT mirrors it.

`rnn_layer` > 1 (build-defined stacked decoder of BASELINE configs[3]; the reference stores the argument and never
reads it, :42,46,174): layer l >= 1 is another lstm_unit (`lstm_w_l<l>` [H+H, 4H], `lstm_b_l<l>`) fed with the new
hidden state of layer l-1; sentinel, attention and the output head read the TOP layer.  In train mode the layers run
one after another over all T steps: layer l's input gates are ONE M = T*B row GEMM over layer l-1's hidden states.
"""
import os

import torch

from ._lib import lib, ACT_NONE, ACT_RELU, ACT_TANH, WGRAD_WS_BYTES, NtCall, gemm_geom, wgrad_workspace
from .params import FC


def _p(t):
    return None if t is None else t.data_ptr()


def vocab_ld(V):
    return (V + 7) // 8 * 8


class DecoderRunner:
    fuse_lstm = True        # recurrent product + cell in one launch per step where capmi_lstm_step_supported
    seq_lstm = True         # the whole recurrence of a layer in one launch per direction where capmi_lstm_seq_supported

    def __init__(self, store, B, K, T, dtype_code, torch_dtype, slots, need_backward=True):
        self.overlap_wgrad = True       # parameter-gradient launches of the backward plan go to the side lane
        self._sync_at = 0
        cfg = store.cfg
        self.store = store
        self.B, self.K, self.T = B, K, T
        self.C, self.H, self.E, self.V = store.enc.channels, cfg['hidden'], cfg['embed'], cfg['vocab']
        self.Vld = vocab_ld(self.V)
        self.code, self.tdt = dtype_code, torch_dtype
        self.slots = 1 if slots else 0
        self.pad = cfg['padding_idx']
        self.L = int(cfg.get('rnn_layer', 1))
        # grid-barrier state of the persistent recurrence kernels: 16 bytes per launch slot (layer x direction).  Word 0 =
        # arrival counter, word 1 = a barrier gave up in the launch in flight (both zeroed by the library per launch),
        # word 2 = STICKY copy of word 1 that only check_sync clears: a time-out in any earlier step stays visible
        self.seq_sync = torch.zeros((2 * self.L, 4), dtype=torch.int32, device=store.device)
        self.use_seq = (self.seq_lstm and os.environ.get('CAPMI_LSTM_SEQ', '1') != '0'
                        and bool(lib().capmi_lstm_seq_supported(B, cfg['hidden'], T, dtype_code)))
        dev = store.device
        B, K, T, C, H, E, M = self.B, self.K, self.T, self.C, self.H, self.E, T * B
        z = lambda shape, dt=None: torch.zeros(shape, dtype=dt or self.tdt, device=dev)
        f32, i64 = torch.float32, torch.int64
        self.ids = z((M,), i64)            # source words, time-major
        self.tgt = z((M,), i64)
        self.V0, self.Amean, self.g = z((B * K, H)), z((B, C)), z((B, H))
        self.Vt = z((B * K, H))
        self.Ve = z((B * K, H)) if slots else None
        self.X = z((M, E + H))
        # per LSTM layer: gate pre-activations and (T+1) row blocks of h / c (block 0 = the zero state, :63)
        self.Gs = [z((M, 4 * H)) for _ in range(self.L)]
        self.Hbufs = [z(((T + 1) * B, H)) for _ in range(self.L)]
        self.Cbufs = [z(((T + 1) * B, H)) for _ in range(self.L)]
        self.G, self.Hbuf, self.Cbuf = self.Gs[0], self.Hbufs[-1], self.Cbufs[-1]     # Hbuf / Cbuf: the TOP layer (what the loop body reads after the LSTM)
        self.SGpre, self.S, self.P = z((M, H)), z((M, H)), z((M, H))
        self.Q = z((M, H)) if slots else None
        self.SE = z((M, H)) if slots else None
        self.alpha = z((M, K + 1), f32)
        self.CTXP, self.O, self.R = z((M, H)), z((M, H)), z((M, E))
        self.logits = z((M, self.Vld), f32)
        self.row_loss, self.row_lse = z((M,), f32), z((M,), f32)
        self.loss, self.count = z((1,), f32), z((1,), f32)
        if need_backward:
            self.dlogits = z((M, self.Vld))
            self.dR, self.dOpre, self.dCTXP = z((M, E)), z((M, H)), z((M, H))
            self.dS, self.dPpre = z((M, H)), z((M, H))
            self.dVt = z((B * K, H))
            self.dVe = z((B * K, H)) if slots else None
            self.dQ = z((M, H)) if slots else None
            self.dSE = z((M, H)) if slots else None
            self.de = z((M, K + 1), f32)
            self.dHbufs = [z(((T + 1) * B, H)) for _ in range(self.L)]
            self.dCbufs = [z(((T + 1) * B, H)) for _ in range(self.L)]
            self.dGs = [z((M, 4 * H)) for _ in range(self.L)]
            self.dHbuf, self.dCbuf, self.dG = self.dHbufs[-1], self.dCbufs[-1], self.dGs[0]
            self.dSGpre, self.dX = z((M, H)), z((M, E + H))
            self.dg, self.dV0, self.dAmean = z((B, H)), z((B * K, H)), z((B, C))

    def check_sync(self):
        """Raises if a grid barrier of the persistent recurrence kernels timed out in ANY launch since the last call (the
        sticky word of each launch slot; reading it synchronises the device).  The flag is cleared when it is reported."""
        if bool(self.seq_sync[:, 2].any().item()):
            from ._lib import CapmiError
            words = self.seq_sync.cpu().tolist()
            self.seq_sync[:, 2].zero_()
            raise CapmiError('capmi_lstm_seq: a grid barrier gave up waiting (sync words %s); the recurrence results of at least one '
                             'step since the last check are invalid' % words)

    # ------------------------------------------------------------------ tiny helpers
    def _gemm(self, plan, x, rows, K, w, N, y, ldw=None, ldx=None, ldy=None, bias=None, addend=None, ld_add=0,
              act=ACT_NONE, dact=ACT_NONE, ysaved=None, ld_saved=0, out_f32=0):
        """y[rows][N] = epilogue(x[rows][K] . w[N][K]^T); pointers may be ints (pre-offset)."""
        g = gemm_geom(rows, K, ldx)
        plan.add('capmi_igemm_nt', x, w, y, g, N, K if ldw is None else ldw, N if ldy is None else ldy, bias, addend,
                 ld_add, ysaved, ld_saved, None, act, dact, out_f32, self.code)

    def _side_ready(self, plan):
        """Parameter-gradient launches run on the side lane (they feed nothing downstream): before one is
        recorded, let lane 1 see everything the main lane has produced so far.  One event per burst."""
        if not self.overlap_wgrad:
            return 0
        tail = plan.calls[self._sync_at:]
        if self._sync_at == 0 or any(fn is not None and not getattr(fn, 'lane', 0) for fn, _, _ in tail):
            key = ('dec', len(plan.calls))
            plan.record(key, 0)
            plan.wait(key, 1)
            self._sync_at = len(plan.calls)
        return 1

    def _wgrad(self, plan, x, rows, K, dy, N, dw, ldx=None, ldy=None, lddw=None):
        """dw[N][K] += dy[rows][N]^T . x[rows][K]"""
        g = gemm_geom(rows, K, ldx)
        lane = self._side_ready(plan)
        plan.add('capmi_igemm_tn_wgrad', x, dy, dw, g, N, N if ldy is None else ldy, K if lddw is None else lddw,
                 _p(wgrad_workspace(self.store.device, lane)), WGRAD_WS_BYTES, self.code, lane=lane)

    def _colsum(self, plan, a, rows, N, out, lda=None):
        plan.add('capmi_colsum', a, rows, N, N if lda is None else lda, out, self.code, lane=self._side_ready(plan))

    def _fc(self, key):
        return FC[key] + '.w_0', FC[key] + '.b_0'

    def _lstm(self, l):
        """(weight name, bias name, input width) of LSTM layer l; kernel layout [4H][in | H], row stride in + H."""
        if l == 0:
            return 'lstm_w', 'lstm_b', self.E + self.H
        return 'lstm_w_l%d' % l, 'lstm_b_l%d' % l, self.H

    # ------------------------------------------------------------------ shared forward pieces
    def _plan_bridge(self, plan, A, W):
        """`_img2feature` :191-199 and the pre-loop projections :52-53.  A = encoder output [B*K, C]."""
        st, B, K, C, H = self.store, self.B, self.K, self.C, self.H
        w0, b0 = self._fc('img_embed')
        w1, b1 = self._fc('img_global')
        w2, b2 = self._fc('img_feat')
        w3, b3 = self._fc('img_feat_emb')
        self._gemm(plan, _p(A), B * K, C, _p(W(w0)), H, _p(self.V0), bias=_p(st.view(b0)), act=ACT_RELU)          # :196
        plan.add('capmi_mean_rows', _p(A), _p(self.Amean), B, K, C, self.code)                                    # :197
        self._gemm(plan, _p(self.Amean), B, C, _p(W(w1)), H, _p(self.g), bias=_p(st.view(b1)), act=ACT_RELU)      # :198
        self._gemm(plan, _p(self.V0), B * K, H, _p(W(w2)), H, _p(self.Vt), bias=_p(st.view(b2)), act=ACT_TANH)    # :52
        if self.slots:
            self._gemm(plan, _p(self.V0), B * K, H, _p(W(w3)), H, _p(self.Ve), bias=_p(st.view(b3)))              # :53

    def _plan_post_lstm(self, plan, W, rows, X, Hprev, Hcur, Ccur, row0=0, have_s=False):
        """Everything of one loop body after the lstm_unit (:89-117) for `rows` rows starting at
        row `row0` of the time-major buffers; X/Hprev/Hcur/Ccur are pre-offset pointers.  have_s: the sentinel S is
        already in place (the fused decode step, capmi_lstm_cell_sentinel_fwd) and the two projections that only need h
        and S -- p_hid :99 and sent_emb :104 -- go out as ONE grouped launch."""
        st, H, E, V = self.store, self.H, self.E, self.V
        es = self.SGpre.element_size()
        o = row0 * H * es
        SG, S, P, CTXP, O = (_p(t) + o for t in (self.SGpre, self.S, self.P, self.CTXP, self.O))
        R = _p(self.R) + row0 * E * es
        w5, b5 = self._fc('p_word')
        w6, b6 = self._fc('p_hidden')
        w7, b7 = self._fc('p_hid')
        w8, b8 = self._fc('hid_emb')
        w9, b9 = self._fc('sent_emb')
        w10, b10 = self._fc('alpha')
        w11, b11 = self._fc('out')
        w12, b12 = self._fc('proj')
        if not have_s:
            self._gemm(plan, X, rows, E + H, _p(W(w5)), H, SG, bias=_p(st.view(b5)))                              # :89
            self._gemm(plan, Hprev, rows, H, _p(W(w6)), H, SG, bias=_p(st.view(b6)), addend=SG, ld_add=H)        # :90-91
            plan.add('capmi_sentinel_fwd', SG, Ccur, S, rows * H, self.code)                                      # :91-92
        Tn = rows // self.B
        Q = SE = None
        if self.slots:
            Q, SE = _p(self.Q) + o, _p(self.SE) + o
        if have_s and self.slots:
            calls = (NtCall * 2)()
            for cl, (xin, wn_, bn_, out, act) in zip(calls, ((Hcur, w7, b7, P, ACT_TANH), (S, w9, b9, SE, ACT_NONE))):   # :99, :104
                cl.x, cl.w, cl.y, cl.g = xin, _p(W(wn_)), out, gemm_geom(rows, H)
                cl.N, cl.ldw, cl.ldy = H, H, H
                cl.bias, cl.act = _p(st.view(bn_)), act
            plan.add('capmi_igemm_nt_group', calls, 2, self.code)
            self._gemm(plan, P, rows, H, _p(W(w8)), H, Q, bias=_p(st.view(b8)))                                   # :102
        else:
            self._gemm(plan, Hcur, rows, H, _p(W(w7)), H, P, bias=_p(st.view(b7)), act=ACT_TANH)                 # :99
            if self.slots:
                self._gemm(plan, P, rows, H, _p(W(w8)), H, Q, bias=_p(st.view(b8)))                               # :102
                self._gemm(plan, S, rows, H, _p(W(w9)), H, SE, bias=_p(st.view(b9)))                              # :104
        alpha = _p(self.alpha) + row0 * (self.K + 1) * 4
        plan.add('capmi_ada_attention_fwd', _p(self.Ve), _p(self.Vt), Q, SE, S, P, _p(W(w10)) if self.slots else None,
                 _p(st.view(b10)), CTXP, alpha, Tn, self.B, self.K, H, self.slots, self.code)                     # :103-113
        self._gemm(plan, CTXP, rows, H, _p(W(w11)), H, O, bias=_p(st.view(b11)), act=ACT_TANH)                   # :115
        self._gemm(plan, O, rows, H, _p(W(w12)), E, R, bias=_p(st.view(b12)))                                    # :24
        logits = _p(self.logits) + row0 * self.Vld * 4
        self._gemm(plan, R, rows, E, _p(W('word_embedding')), V, logits, ldy=self.Vld,
                   bias=_p(st.view('out_fc_bias')), out_f32=1)                                                    # :25

    # ------------------------------------------------------------------ train forward
    def plan_forward(self, plan, A, W):
        st, B, K, T, H, E = self.store, self.B, self.K, self.T, self.H, self.E
        M, code = T * B, self.code
        es = self.X.element_size()
        self._plan_bridge(plan, A, W)
        # x_t = [embedding(w_t) ; g]  (:84-86), all steps at once
        plan.add('capmi_embedding_fwd', _p(self.ids), _p(W('word_embedding')), _p(self.X), M, E, self.V, E + H, self.pad, code)
        plan.add('capmi_bcast_rows', _p(self.g), _p(self.X), T, B, H, E + H, E, code)
        # per layer: input part of the gates for every step in one GEMM (G = in . Wx^T + b; kernel layout [4H][in | H]),
        # then the recurrence: only h_{t-1} . Wh^T and the cell are sequential
        off1 = B * H * es
        for l in range(self.L):
            wn, bn, kin = self._lstm(l)
            lw, ldl = W(wn), kin + H
            G, Hb, Cb = self.Gs[l], self.Hbufs[l], self.Cbufs[l]
            xin = _p(self.X) if l == 0 else _p(self.Hbufs[l - 1]) + off1
            self._gemm(plan, xin, M, kin, _p(lw), 4 * H, _p(G), ldw=ldl, bias=_p(st.view(bn)))
            wh = _p(lw) + kin * es
            if self.use_seq:      # all T steps of the layer in one launch (grid barrier between steps)
                plan.add('capmi_lstm_seq_fwd', _p(Hb), wh, ldl, _p(G), _p(Cb), B, H, T, self.seq_sync.data_ptr() + 32 * l, code)
                continue
            fused = self.fuse_lstm and os.environ.get('CAPMI_LSTM_FUSE', '1') != '0' and bool(lib().capmi_lstm_step_supported(B, H, code))   # one launch per step
            for t in range(T):                                                                                    # :75-127
                Gt = _p(G) + t * B * 4 * H * es
                if t > 0 and fused:
                    plan.add('capmi_lstm_step_fwd', _p(Hb) + t * B * H * es, wh, ldl, Gt, _p(Cb) + t * B * H * es,
                             _p(Hb) + (t + 1) * B * H * es, _p(Cb) + (t + 1) * B * H * es, B, H, code)
                    continue
                if t > 0:     # h_{-1} = 0 (:63): nothing to add at t = 0
                    self._gemm(plan, _p(Hb) + t * B * H * es, B, H, wh, 4 * H, Gt, ldw=ldl, addend=Gt, ld_add=4 * H)
                plan.add('capmi_lstm_cell_fwd', Gt, _p(Cb) + t * B * H * es, _p(Hb) + (t + 1) * B * H * es,
                         _p(Cb) + (t + 1) * B * H * es, B, H, code)                                                # :87-88
        off1 = B * H * es
        self._plan_post_lstm(plan, W, M, _p(self.X), _p(self.Hbuf), _p(self.Hbuf) + off1, _p(self.Cbuf) + off1)
        plan.add('capmi_softmax_xent_fwd', _p(self.logits), _p(self.tgt), _p(self.row_loss), _p(self.row_lse), M, self.V,
                 self.Vld, self.pad)                                                                              # :205-212
        plan.add('capmi_xent_finalize', _p(self.row_loss), _p(self.tgt), _p(self.loss), _p(self.count), M, self.pad)  # :165-182

    # ------------------------------------------------------------------ train backward
    def plan_backward(self, plan, A, dA, W, WT):
        """W(name): forward weights, WT(name): data-gradient forms.  Writes d(loss)/dA into dA and
        accumulates every decoder parameter gradient into the store's flat gradient buffer."""
        st, B, K, T, C, H, E, V, Vld = self.store, self.B, self.K, self.T, self.C, self.H, self.E, self.V, self.Vld
        M, code = T * B, self.code
        self._sync_at = 0
        es = self.X.element_size()
        g_ = lambda n: _p(st.gview(n))
        w0, b0 = self._fc('img_embed')
        w1, b1 = self._fc('img_global')
        w2, b2 = self._fc('img_feat')
        w3, b3 = self._fc('img_feat_emb')
        w5, b5 = self._fc('p_word')
        w6, b6 = self._fc('p_hidden')
        w7, b7 = self._fc('p_hid')
        w8, b8 = self._fc('hid_emb')
        w9, b9 = self._fc('sent_emb')
        w10, b10 = self._fc('alpha')
        w11, b11 = self._fc('out')
        w12, b12 = self._fc('proj')
        # loss -> logits
        plan.add('capmi_softmax_xent_bwd', _p(self.logits), _p(self.tgt), _p(self.row_lse), _p(self.count), _p(self.dlogits),
                 M, V, Vld, Vld, self.pad, code)
        # tied projection (:25): d bias, d Emb (dense part, quirk Q6), d proj
        self._colsum(plan, _p(self.dlogits), M, V, g_('out_fc_bias'), lda=Vld)
        self._wgrad(plan, _p(self.R), M, E, _p(self.dlogits), V, g_('word_embedding'), ldy=Vld)
        if self.overlap_wgrad:
            plan.record(('dec', 'tied projection'), 1)      # embedding_bwd adds into the same rows (ordered below)
        # [T*B][V] x [V][E]: a long reduction into 40 output tiles -- split over workgroups (f32 slabs in the main lane's scratch)
        if os.environ.get('CAPMI_SPLITK', '1') == '0':
            self._gemm(plan, _p(self.dlogits), M, Vld, _p(WT('word_embedding')), E, _p(self.dR), ldw=Vld)
        else:
            plan.add('capmi_igemm_nt_splitk', _p(self.dlogits), _p(WT('word_embedding')), _p(self.dR), M, Vld, Vld, E, Vld, E,
                     _p(wgrad_workspace(self.store.device, 0)), WGRAD_WS_BYTES, code)
        # fc_12 (:24) and fc_11 + tanh (:115)
        self._wgrad(plan, _p(self.O), M, H, _p(self.dR), E, g_(w12))
        self._colsum(plan, _p(self.dR), M, E, g_(b12))
        self._gemm(plan, _p(self.dR), M, E, _p(WT(w12)), H, _p(self.dOpre), dact=ACT_TANH, ysaved=_p(self.O), ld_saved=H)
        self._wgrad(plan, _p(self.CTXP), M, H, _p(self.dOpre), H, g_(w11))
        self._colsum(plan, _p(self.dOpre), M, H, g_(b11))
        self._gemm(plan, _p(self.dOpre), M, H, _p(WT(w11)), H, _p(self.dCTXP))
        # attention (:97-113)
        plan.add('capmi_ada_attention_bwd', _p(self.Ve), _p(self.Vt), _p(self.Q), _p(self.SE), _p(self.S),
                 _p(W(w10)) if self.slots else None, _p(self.alpha), _p(self.dCTXP), _p(self.dS), _p(self.dVt), _p(self.dVe),
                 _p(self.dQ), _p(self.dSE), g_(w10), g_(b10), _p(self.de), T, B, K, H, self.slots, code)
        if self.slots:
            self._wgrad(plan, _p(self.S), M, H, _p(self.dSE), H, g_(w9))
            self._colsum(plan, _p(self.dSE), M, H, g_(b9))
            self._gemm(plan, _p(self.dSE), M, H, _p(WT(w9)), H, _p(self.dS), addend=_p(self.dS), ld_add=H)
            self._wgrad(plan, _p(self.P), M, H, _p(self.dQ), H, g_(w8))
            self._colsum(plan, _p(self.dQ), M, H, g_(b8))
            self._gemm(plan, _p(self.dQ), M, H, _p(WT(w8)), H, _p(self.dPpre), addend=_p(self.dCTXP), ld_add=H,
                       dact=ACT_TANH, ysaved=_p(self.P), ld_saved=H)
        else:
            plan.add('capmi_act_bwd', _p(self.dCTXP), _p(self.P), _p(self.dPpre), 0, M * H, ACT_TANH, code)
        # fc_7 (:99): p_hid = tanh(fc(h))
        off1 = B * H * es
        Hall, Hprev = _p(self.Hbuf) + off1, _p(self.Hbuf)
        dHall, dCall = _p(self.dHbuf) + off1, _p(self.dCbuf) + off1
        self._wgrad(plan, Hall, M, H, _p(self.dPpre), H, g_(w7))
        self._colsum(plan, _p(self.dPpre), M, H, g_(b7))
        self._gemm(plan, _p(self.dPpre), M, H, _p(WT(w7)), H, dHall)
        # sentinel (:89-92)
        plan.add('capmi_sentinel_bwd', _p(self.dS), _p(self.SGpre), _p(self.Cbuf) + off1, _p(self.dSGpre), dCall, M * H, code)
        self._wgrad(plan, _p(self.X), M, E + H, _p(self.dSGpre), H, g_(w5))
        self._wgrad(plan, Hprev, M, H, _p(self.dSGpre), H, g_(w6))
        self._colsum(plan, _p(self.dSGpre), M, H, g_(b5))
        self._colsum(plan, _p(self.dSGpre), M, H, g_(b6))
        self._gemm(plan, _p(self.dSGpre), M, H, _p(WT(w5)), E + H, _p(self.dX))
        # d h_{t-1} through fc_6 lands one row block earlier in dHbuf (block 0 = h_{-1}, unused)
        self._gemm(plan, _p(self.dSGpre), M, H, _p(WT(w6)), H, _p(self.dHbuf), addend=_p(self.dHbuf), ld_add=H)
        # BPTT through the lstm_unit layers (:87-88), top layer first.  The top layer's dHbuf / dCbuf hold what the loop
        # body's readers left there (above); a lower layer's d h_t is the input gradient of the layer above it.
        for l in reversed(range(self.L)):
            wn, bn, kin = self._lstm(l)
            lwT = WT(wn)                             # [kin + H][4H]
            whT = _p(lwT) + kin * 4 * H * es
            G, dG, Cb, dHb, dCb = self.Gs[l], self.dGs[l], self.Cbufs[l], self.dHbufs[l], self.dCbufs[l]
            top = l == self.L - 1
            # measured at cfg 2: the fused forward step wins (10.8 us vs 15.7 + 5.2), the fused backward step (16 workgroups
            # carrying the whole cell backward) loses to product + cell (18-29 us vs 7.9 + 6.0): forward only by default
            fused = self.fuse_lstm and os.environ.get('CAPMI_LSTM_FUSE', '1') == '2' and bool(lib().capmi_lstm_step_supported(B, H, code))
            if self.use_seq:      # BPTT of the layer in one launch
                plan.add('capmi_lstm_seq_bwd', _p(G), _p(Cb), whT, 4 * H, _p(dHb), _p(dCb), _p(dG), 1 if top else 0, B, H, T,
                         self.seq_sync.data_ptr() + 32 * l + 16, code)
            for t in reversed(range(T if not self.use_seq else 0)):
                blk = lambda buf, i: _p(buf) + i * B * H * es
                Gt = _p(G) + t * B * 4 * H * es
                dGt = _p(dG) + t * B * 4 * H * es
                # incoming d c_t: the sentinel's share + step t+1's (top layer, accumulated in place); step t+1's only below it
                dc_in = blk(dCb, t + 1) if (top or t < T - 1) else None
                acc = 1 if top else 0
                if not fused or t == T - 1:
                    plan.add('capmi_lstm_cell_bwd', Gt, blk(Cb, t), blk(Cb, t + 1), blk(dHb, t + 1),
                             dc_in, dGt, blk(dCb, t) if t > 0 else None, acc, B, H, code)
                if t > 0 and fused:
                    # dh_{t-1} += dG_t . Wh and the cell backward of step t-1 in one launch
                    plan.add('capmi_lstm_step_bwd', dGt, whT, 4 * H, blk(dHb, t), Gt - B * 4 * H * es, blk(Cb, t - 1), blk(Cb, t),
                             blk(dCb, t), dGt - B * 4 * H * es, blk(dCb, t - 1) if t > 1 else None, acc, B, H, code)
                elif t > 0:
                    self._gemm(plan, dGt, B, 4 * H, whT, H, blk(dHb, t), addend=blk(dHb, t), ld_add=H)
            ldl = kin + H
            xin = _p(self.X) if l == 0 else _p(self.Hbufs[l - 1]) + off1
            self._wgrad(plan, xin, M, kin, _p(dG), 4 * H, g_(wn), lddw=ldl)
            self._wgrad(plan, _p(self.Hbufs[l]), M, H, _p(dG), 4 * H, g_(wn) + kin * 4, lddw=ldl)
            self._colsum(plan, _p(dG), M, 4 * H, g_(bn))
            if l == 0:
                self._gemm(plan, _p(dG), M, 4 * H, _p(lwT), kin, _p(self.dX), addend=_p(self.dX), ld_add=kin)
            else:       # d h of layer l-1 for every step (blocks 1..T of its dHbuf): written, then its own BPTT adds into it
                self._gemm(plan, _p(dG), M, 4 * H, _p(lwT), H, _p(self.dHbufs[l - 1]) + off1)
        # x_t = [emb ; g] (:84-86)
        if self.overlap_wgrad:
            plan.wait(('dec', 'tied projection'), 0)
        plan.add('capmi_embedding_bwd', _p(self.ids), _p(self.dX), g_('word_embedding'), M, E, V, E + H, self.pad, code)
        plan.add('capmi_bcast_rows_bwd', _p(self.dX), _p(self.dg), T, B, H, E + H, E, code)
        # pre-loop projections (:52-53)
        plan.add('capmi_act_bwd', _p(self.dVt), _p(self.Vt), _p(self.dVt), 0, B * K * H, ACT_TANH, code)
        self._wgrad(plan, _p(self.V0), B * K, H, _p(self.dVt), H, g_(w2))
        self._colsum(plan, _p(self.dVt), B * K, H, g_(b2))
        if self.slots:
            self._wgrad(plan, _p(self.V0), B * K, H, _p(self.dVe), H, g_(w3))
            self._colsum(plan, _p(self.dVe), B * K, H, g_(b3))
            self._gemm(plan, _p(self.dVe), B * K, H, _p(WT(w3)), H, _p(self.dV0))
            self._gemm(plan, _p(self.dVt), B * K, H, _p(WT(w2)), H, _p(self.dV0), addend=_p(self.dV0), ld_add=H,
                       dact=ACT_RELU, ysaved=_p(self.V0), ld_saved=H)
        else:
            self._gemm(plan, _p(self.dVt), B * K, H, _p(WT(w2)), H, _p(self.dV0), dact=ACT_RELU, ysaved=_p(self.V0), ld_saved=H)
        # bridge (:191-199)
        self._wgrad(plan, _p(A), B * K, C, _p(self.dV0), H, g_(w0))
        self._colsum(plan, _p(self.dV0), B * K, H, g_(b0))
        self._gemm(plan, _p(self.dV0), B * K, H, _p(WT(w0)), C, _p(dA))
        plan.add('capmi_act_bwd', _p(self.dg), _p(self.g), _p(self.dg), 0, B * H, ACT_RELU, code)
        self._wgrad(plan, _p(self.Amean), B, C, _p(self.dg), H, g_(w1))
        self._colsum(plan, _p(self.dg), B, H, g_(b1))
        self._gemm(plan, _p(self.dg), B, H, _p(WT(w1)), C, _p(self.dAmean))
        plan.add('capmi_mean_rows_bwd', _p(self.dAmean), _p(dA), B, K, C, code)

    # ------------------------------------------------------------------ fused decode step (one LSTM layer)
    fuse_decode = True      # CAPMI_DECODE_FUSE=0: the per-projection step (round 2; the only form for rnn_layer > 1)

    def _decode_fused(self):
        return self.fuse_decode and self.L == 1 and os.environ.get('CAPMI_DECODE_FUSE', '1') != '0'

    def build_stacked(self, R):
        """Buffers of the fused decode step: xh = [embedding | global feature | h_prev] rows, gs = [i | f | o | g | sentinel
        gate] pre-activations, and the stacked weight / bias of the ONE GEMM between them: rows 0..4H-1 = lstm_w (already
        [in | H] wide, the lstm_unit's fc over the concatenated input, :87-88), rows 4H..5H-1 = [fc_5 | fc_6] (:89-90)."""
        H, E = self.H, self.E
        dev = self.X.device
        self.XH = torch.zeros((R, E + 2 * H), dtype=self.tdt, device=dev)
        self.GS = torch.zeros((R, 5 * H), dtype=self.tdt, device=dev)
        self.Wstack = torch.zeros((5 * H, E + 2 * H), dtype=self.tdt, device=dev)
        self.bstack = torch.zeros((5 * H,), dtype=torch.float32, device=dev)
        self.stack_version = None

    def refresh_stacked(self, W, version):
        """Copies the current weights into the stacked GEMM operand (device-to-device slices; called by the engine whenever
        the weight shadows changed since the last decode)."""
        if getattr(self, 'Wstack', None) is None or self.stack_version == version:
            return
        st, H, E = self.store, self.H, self.E
        w5, b5 = self._fc('p_word')
        w6, b6 = self._fc('p_hidden')
        self.Wstack[:4 * H].copy_(W('lstm_w').reshape(4 * H, E + 2 * H))
        self.Wstack[4 * H:, :E + H].copy_(W(w5).reshape(H, E + H))
        self.Wstack[4 * H:, E + H:].copy_(W(w6).reshape(H, H))
        self.bstack[:4 * H].copy_(st.view('lstm_b').reshape(-1))
        self.bstack[4 * H:].copy_(st.view(b5).reshape(-1) + st.view(b6).reshape(-1))
        self.stack_version = version

    def _plan_fused_step(self, plan, W, R, rows, h_src, c_src, h_dst, c_dst):
        """One decode step in 11 launches (19 before): state plumbing (embedding + the survivors' h into the GEMM operand),
        ONE GEMM for the lstm_unit's gates and the sentinel gate, cell + sentinel, p_hid || sent_emb grouped, hid_emb,
        attention, out, proj, vocabulary -- then the caller's argmax / beam step.  rows: device int32 row map of the
        surviving hypotheses (None: identity)."""
        H, E = self.H, self.E
        code = self.code
        plan.add('capmi_decode_prep', _p(self.ids), _p(W('word_embedding')), h_src, rows, _p(self.XH), R, E, H, self.V, E + 2 * H, E + H,
                 self.pad, code)                                                                                  # :84-86 (+ beam gather)
        plan.add('capmi_igemm_nt', _p(self.XH), _p(self.Wstack), _p(self.GS), gemm_geom(R, E + 2 * H), 5 * H, E + 2 * H, 5 * H,
                 _p(self.bstack), None, 0, None, 0, None, ACT_NONE, ACT_NONE, 0, code)                            # :87-91
        plan.add('capmi_lstm_cell_sentinel_fwd', _p(self.GS), 5 * H, c_src, rows, h_dst, c_dst, _p(self.S), R, H, code)   # :87-88, :91-92
        self._plan_post_lstm(plan, W, R, _p(self.XH), None, h_dst, c_dst, have_s=True)

    # ------------------------------------------------------------------ greedy decode (eval graph)
    def plan_greedy(self, plan, A, W, out_ids_f32, Ti):
        """`eval_network` :185-189 + the eval branches of Decoder.call: first fed token start_idx
        (:56-58, the caller fills self.ids[:B]), Ti = infer_max_length steps, no early stop (Q5),
        argmax feedback (:119-121), ids written as float32 [B, Ti] (Q2)."""
        st, B, H, E = self.store, self.B, self.H, self.E
        assert self.T >= 1
        code = self.code
        es = self.X.element_size()
        self._plan_bridge(plan, A, W)
        if self._decode_fused():
            self.build_stacked(B)
            plan.add('capmi_bcast_rows', _p(self.g), _p(self.XH), 1, B, H, E + 2 * H, E, code)
            for t in range(Ti):
                cur, nxt = (t % 2), ((t + 1) % 2)
                hp, cp = _p(self.Hbufs[0]) + cur * B * H * es, _p(self.Cbufs[0]) + cur * B * H * es
                hn, cn = _p(self.Hbufs[0]) + nxt * B * H * es, _p(self.Cbufs[0]) + nxt * B * H * es
                self._plan_fused_step(plan, W, B, None, hp, cp, hn, cn)
                plan.add('capmi_argmax', _p(self.logits), _p(self.ids), _p(out_ids_f32) + t * 4, Ti, B, self.V, self.Vld)  # :120-123
            return
        plan.add('capmi_bcast_rows', _p(self.g), _p(self.X), 1, B, H, E + H, E, code)
        # per layer two (h, c) row blocks used alternately: block 0 starts as the zero state (:63)
        for t in range(Ti):
            cur, nxt = (t % 2), ((t + 1) % 2)
            plan.add('capmi_embedding_fwd', _p(self.ids), _p(W('word_embedding')), _p(self.X), B, E, self.V, E + H, self.pad, code)
            xin = _p(self.X)
            for l in range(self.L):
                wn, bn, kin = self._lstm(l)
                lw, ldl = W(wn), kin + H
                hp, cp = _p(self.Hbufs[l]) + cur * B * H * es, _p(self.Cbufs[l]) + cur * B * H * es
                hn, cn = _p(self.Hbufs[l]) + nxt * B * H * es, _p(self.Cbufs[l]) + nxt * B * H * es
                self._gemm(plan, xin, B, kin, _p(lw), 4 * H, _p(self.G), ldw=ldl, bias=_p(st.view(bn)))
                self._gemm(plan, hp, B, H, _p(lw) + kin * es, 4 * H, _p(self.G), ldw=ldl, addend=_p(self.G), ld_add=4 * H)
                plan.add('capmi_lstm_cell_fwd', _p(self.G), cp, hn, cn, B, H, code)
                xin = hn
            self._plan_post_lstm(plan, W, B, _p(self.X), hp, hn, cn)       # top layer: previous h, new h, new c
            plan.add('capmi_argmax', _p(self.logits), _p(self.ids), _p(out_ids_f32) + t * 4, Ti, B, self.V, self.Vld)  # :120-123

    def plan_beam(self, plan, A, W, out_ids_f32, Ti, beam):
        """Beam-search decode (build-defined extension, oracle/model.py beam_decode): the greedy loop on beam*B rows
        (row k*B + b = hypothesis k of image b; this runner is built with T = beam so its per-step buffers hold
        them), plus per step capmi_beam_step (keep the `beam` best continuations per image) and a gather of the
        surviving hypotheses' (h, c).  The caller fills self.ids[:beam*B] with start_idx and self.beam_score[0]."""
        st, B, H, E = self.store, self.B, self.H, self.E
        assert self.T >= beam and 1 <= beam <= 8
        code = self.code
        es = self.X.element_size()
        R = beam * B
        dev = self.X.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        # per layer: [0] = state fed to the step, [1] = state it produces
        self.beam_h = [[z((R, H), self.tdt) for _ in range(2)] for _ in range(self.L)]
        self.beam_c = [[z((R, H), self.tdt) for _ in range(2)] for _ in range(self.L)]
        self.beam_score = [z((beam, B), torch.float32) for _ in range(2)]
        self.beam_cand_val, self.beam_cand_idx = z((R, beam), torch.float32), z((R, beam), torch.int32)
        self.beam_lse, self.beam_rows = z((R,), torch.float32), z((R,), torch.int32)
        self.beam_parents, self.beam_tokens = z((Ti, beam, B), torch.int32), z((Ti, beam, B), torch.int32)
        self._plan_bridge(plan, A, W)
        if self._decode_fused():
            # the step's new (h, c) land in buffer 1 - t % 2 ... the survivors are picked from it by the NEXT step's prep / cell
            # kernels through beam_rows: no gather launches, two (h, c) buffers used alternately
            self.build_stacked(R)
            self.beam_ident = torch.arange(R, dtype=torch.int32, device=dev)
            plan.add('capmi_bcast_rows', _p(self.g), _p(self.XH), beam, B, H, E + 2 * H, E, code)
            plan.add('capmi_fill_f32', _p(self.beam_h[0][0]), 0.0, R * H * es // 4)         # h_{-1} = c_{-1} = 0 (:63)
            plan.add('capmi_fill_f32', _p(self.beam_c[0][0]), 0.0, R * H * es // 4)
            for t in range(Ti):
                sin, sout = self.beam_score[t % 2], self.beam_score[(t + 1) % 2]
                src, dst = t % 2, (t + 1) % 2
                rows = _p(self.beam_rows) if t > 0 else None                                # step 0: every row starts from the zero state
                self._plan_fused_step(plan, W, R, rows, _p(self.beam_h[0][src]), _p(self.beam_c[0][src]),
                                      _p(self.beam_h[0][dst]), _p(self.beam_c[0][dst]))
                off = t * beam * B * 4
                plan.add('capmi_beam_step', _p(self.logits), self.V, self.Vld, B, beam, _p(sin), _p(sout), _p(self.beam_cand_val),
                         _p(self.beam_cand_idx), _p(self.beam_lse), _p(self.beam_parents) + off, _p(self.beam_tokens) + off,
                         _p(self.ids), _p(self.beam_rows))
            plan.add('capmi_beam_backtrack', _p(self.beam_tokens), _p(self.beam_parents), _p(out_ids_f32), Ti, B, beam)
            self.beam_final_score = self.beam_score[Ti % 2]
            return
        plan.add('capmi_bcast_rows', _p(self.g), _p(self.X), beam, B, H, E + H, E, code)
        for l in range(self.L):
            plan.add('capmi_fill_f32', _p(self.beam_h[l][0]), 0.0, R * H * es // 4)         # h_{-1} = c_{-1} = 0 (:63)
            plan.add('capmi_fill_f32', _p(self.beam_c[l][0]), 0.0, R * H * es // 4)
        for t in range(Ti):
            sin, sout = self.beam_score[t % 2], self.beam_score[(t + 1) % 2]
            plan.add('capmi_embedding_fwd', _p(self.ids), _p(W('word_embedding')), _p(self.X), R, E, self.V, E + H, self.pad, code)
            xin = _p(self.X)
            for l in range(self.L):
                wn, bn, kin = self._lstm(l)
                lw, ldl = W(wn), kin + H
                hp, cp, hn, cn = _p(self.beam_h[l][0]), _p(self.beam_c[l][0]), _p(self.beam_h[l][1]), _p(self.beam_c[l][1])
                self._gemm(plan, xin, R, kin, _p(lw), 4 * H, _p(self.G), ldw=ldl, bias=_p(st.view(bn)))
                self._gemm(plan, hp, R, H, _p(lw) + kin * es, 4 * H, _p(self.G), ldw=ldl, addend=_p(self.G), ld_add=4 * H)
                plan.add('capmi_lstm_cell_fwd', _p(self.G), cp, hn, cn, R, H, code)
                xin = hn
            self._plan_post_lstm(plan, W, R, _p(self.X), hp, hn, cn)
            off = t * beam * B * 4
            plan.add('capmi_beam_step', _p(self.logits), self.V, self.Vld, B, beam, _p(sin), _p(sout), _p(self.beam_cand_val),
                     _p(self.beam_cand_idx), _p(self.beam_lse), _p(self.beam_parents) + off, _p(self.beam_tokens) + off,
                     _p(self.ids), _p(self.beam_rows))
            for l in range(self.L):       # survivors' state (every layer) feeds the next step
                plan.add('capmi_gather_rows', _p(self.beam_h[l][1]), _p(self.beam_rows), _p(self.beam_h[l][0]), R, H, code)
                plan.add('capmi_gather_rows', _p(self.beam_c[l][1]), _p(self.beam_rows), _p(self.beam_c[l][0]), R, H, code)
        plan.add('capmi_beam_backtrack', _p(self.beam_tokens), _p(self.beam_parents), _p(out_ids_f32), Ti, B, beam)
        self.beam_final_score = self.beam_score[Ti % 2]
