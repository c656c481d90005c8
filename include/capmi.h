/* capmi.h -- C ABI of libcapmi.so: the MI355X (gfx950) kernels behind the captioning hot path.
 *
 * The reference (Chgtaxihe/MyImageCaptioningModel) has no FFI of its own: every op below is a
 * PaddlePaddle-1.8 `fluid.layers.*` call that Paddle lowers to its CUDA/cuDNN kernels.  Each
 * entry point names the reference call site(s) it replaces (IC/ = ImageCaptioning/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors); nothing is
 *     allocated or freed here; every call only enqueues work on `stream` (a hipStream_t) and
 *     returns: 0 = ok, non-zero = error (text via capmi_last_error()).  Never throws.
 *   - `dtype` selects the activation/weight storage type: CAPMI_F32 (reference precision, exact
 *     f32 MFMA) or CAPMI_BF16 (bf16 storage, f32 accumulate).  Statistics, losses, gradients of
 *     parameters and optimizer state are always f32.
 *   - encoder activations are NHWC ([B,H,W,C], C fastest); GEMM weights are [N][K] with K
 *     fastest, K ordered (kh, kw, cin) for convolutions.
 *   - optional pointers may be NULL where stated.
 */
#ifndef CAPMI_H
#define CAPMI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPMI_ABI_VERSION 1
enum { CAPMI_F32 = 0, CAPMI_BF16 = 1 };
enum { CAPMI_ACT_NONE = 0, CAPMI_ACT_RELU = 1, CAPMI_ACT_RELU6 = 2, CAPMI_ACT_TANH = 3, CAPMI_ACT_SIGMOID = 4 };
/* `dact` of capmi_igemm_nt / capmi_igemm_nt_group may carry this flag (dact = CAPMI_ACT_RELU | CAPMI_DACT_BITMASK): ysaved then is
 * the activation-derivative BIT MASK capmi_bn_apply_mask wrote ([rows][ld_saved / 8] bytes, ld_saved = channels per row) instead
 * of the saved output itself -- 1/16 of the bytes in the epilogue that masks a ReLU tensor's gradient.  bf16 convolution data
 * gradients only (relu / relu6; N, ldy, ld_saved, ld_addend multiples of 8; no bias / activation / f32 output). */
#define CAPMI_DACT_BITMASK 0x100

int capmi_version(void);
const char* capmi_last_error(void);

/* Deterministic mode (verification; default off, or CAPMI_DETERMINISTIC=1 in the environment).  Paddle's executor gives
 * no run-to-run guarantee either (IC/train.py:121-124 runs cuDNN / atomics underneath); this switch exists so that the
 * build can PROVE schedule-independence: with it on, every f32 atomic accumulation of the library -- weight-gradient
 * split-K on sub-tile outputs (capmi_igemm_tn_wgrad), the embedding scatter (capmi_embedding_bwd), bias column sums
 * (capmi_colsum, capmi_dwconv3x3_bwd_weight) and the attention's d fc_10 (capmi_ada_attention_bwd) -- becomes a
 * fixed-order reduction, and two runs of one launch sequence are bit-identical whatever the lanes' relative timing.
 * Slower (single-pass column sums, slab reductions for every split weight gradient); capmi_ada_attention_bwd then uses
 * a library-owned scratch buffer per device, so its deterministic launches must not overlap on one device. */
int capmi_deterministic(void);
int capmi_set_deterministic(int on);

/* Verification switch (default off, or CAPMI_NT_GENERAL=1): every capmi_igemm_nt* launch runs the general epilogue instantiation
 * instead of the epilogue class the library would pick for it (training convolution forward / data gradient, decoder fc,
 * inference convolution, f32 logits -- the same arithmetic with the paths that launch cannot take compiled out, which is what keeps
 * the kernels within the instruction cache).  Results are bit-identical either way; tests hold them to it. */
int capmi_general_epilogue(void);
int capmi_set_general_epilogue(int on);

/* Measurement aid: WHICH kernel would a capmi_igemm_* call launch?  Between capmi_kernel_probe_begin() and
 * capmi_kernel_probe_end() on one host thread, every capmi_igemm_* / capmi_lstm_* entry point called on that thread goes
 * through its normal argument checks and kernel selection but RECORDS the kernel instead of launching it (nothing is enqueued,
 * no memory is touched).  _end closes the probe and returns the symbol of the first kernel of the call as rocprofv3 prints it
 * (demangled where the C++ ABI demangler can; the mangled name otherwise), its grid (workgroups) and block size, and the
 * number of kernels the call would have launched.  bench.py labels every GEMM launch of a step this way, so its `roofline.kernel`
 * is a row of `rocprofv3 --kernel-trace --stats` verbatim and cannot drift from the dispatch code. */
int capmi_kernel_probe_begin(void);
int capmi_kernel_probe_end(char* symbol, int symbol_len, int* grid, int* block, int* launches);

/* Alternates.  Four entry points are NOT on the default launch plans: each is a fused form that measured slower than
 * the launches it replaces on the ResNet-50 workload, is kept because it is the faster form at other sizes or the
 * per-step fallback of a fused kernel, and is covered by the same parity tests as the default path:
 *   capmi_igemm_nt_bnred     (BN-backward sums from the data-gradient epilogue; CAPMI_BNRED=1)
 *   capmi_bn_finalize_apply  (statistics merge inside the apply kernel)
 *   capmi_im2col_stem        (patch-matrix stem; the default is capmi_s2d_stem)
 *   capmi_lstm_step_bwd      (one BPTT step per launch; CAPMI_LSTM_FUSE=2.  Shapes outside capmi_lstm_seq_supported
 *                             run capmi_lstm_cell_bwd + a skinny product per step.) */

/* Lane synchronisation: device-scope events ordering two HIP streams of one device (the launch plan's
 * main lane and the side lane that runs weight gradients).  No timing, no system-scope fence: a default
 * hipEventRecord writes the L2 back for the host's benefit, ~6 us of idle queue each time. */
int capmi_event_create(void** event);
/* The same with timing enabled (still no system-scope fence), and the milliseconds between two of them once both have
 * completed: the per-launch measurements of bench.py's roofline pass, recorded on the lane a kernel is launched on. */
int capmi_event_create_timed(void** event);
int capmi_event_elapsed_ms(void* start, void* stop, float* ms);
int capmi_event_destroy(void* event);
int capmi_event_record(void* event, void* stream);
int capmi_stream_wait_event(void* stream, void* event);
/* Side-lane stream of the current device: priority < 0 lowest available, > 0 highest, 0 default. */
int capmi_stream_create(void** stream, int priority);

/* Geometry of one implicit-GEMM convolution pass over an NHWC tensor.
 * Output pixel (b,ho,wo), tap (r,q) reads input pixel hn = ho*sd - pad + r (same for w); with
 * `up` > 1 (data-gradient of a strided conv) the tap is valid only if hn % up == 0 and reads
 * hn/up.  A plain GEMM is Hi=Wi=Ho=Wo=kh=kw=sd=up=1, pad=0, B = rows. */
typedef struct {
    int B, Hi, Wi, Cin;   /* input tensor [B,Hi,Wi,*], Cin channels used               */
    int Ho, Wo;           /* output spatial size                                       */
    int kh, kw, sd, up, pad;
    int ldx;              /* elements between consecutive input pixels (>= Cin)        */
    /* Output scatter (0/1 = dense): GEMM row (b,i,j) of the [B,Ho,Wo] grid is stored at pixel
     * (b, i*os + oh0, j*os + ow0) of a [B,Hof,Wof] tensor -- one parity class of a strided
     * convolution's data gradient.  addend / ysaved rows follow the same mapping. */
    int os, oh0, ow0, Hof, Wof;
} capmi_conv_geom;

/* Y[m][n] = epilogue( sum_k A(m,k) * W[n][k] ),  m=(b,ho,wo), k=(r,q,c), A gathered by `g`.
 * Replaces: fluid.layers.conv2d dense calls (IC/model/MobileNetV2.py:99-109) forward and their
 * data gradients; layers.fc / mul (IC/model/model_adaAttention_aic.py:24,52,53,89,90,99,102,
 * 104,107,115,196,198), the lstm_unit gate fc (:87-88) and the tied vocabulary projection
 * matmul(transpose_y=True) (:25), forward and data-gradient.
 * Epilogue, in order: (+ bias[n]) (+ addend[m][n]) -> optional fused batch-norm statistics: for
 * every block of capmi_igemm_nt_stats_part_rows(M,N,K,dtype) consecutive rows and every column the
 * exact (mean, sum (v-mean)^2) of the f32 accumulators, stored to stats[part][N][2] (plain
 * stores: deterministic, cancellation-free; merged by capmi_bn_finalize)
 * -> act -> (* act'(ysaved[m][n]) when dact != NONE: ysaved holds the forward OUTPUT of that
 * activation) -> store as `dtype`, or f32 when out_f32.
 * Requires Cin % (16/sizeof(elem)) == 0 and ldx, ldw likewise; N, M arbitrary. */
int capmi_igemm_nt_stats_part_rows(int M, int N, int K, int dtype);
int capmi_igemm_nt(const void* x, const void* w, void* y, const capmi_conv_geom* g,
                   int N, int ldw, int ldy,
                   const float* bias, const void* addend, int ld_addend,
                   const void* ysaved, int ld_saved, float* stats,
                   int act, int dact, int out_f32, int dtype, void* stream);

/* Convolution on the RAW output of the producing convolution, with that producer's batch norm + activation applied in
 * the A-operand path:  y = conv( act(in_coef_a * (x_raw - in_mean) + in_offset) )  -- the formula and rounding points of
 * capmi_bn_apply, so the result equals capmi_bn_apply followed by capmi_igemm_nt bit for bit, but the normalised tensor
 * is never written to or read from memory for this consumer.  Replaces the `conv2d(batch_norm(conv2d(..)))` links inside
 * a unit chain (IC/model/MobileNetV2.py:88-121 conv_bn_layer, :128-181 / the ResNet bottleneck: conv1 -> conv2 -> conv3).
 * bf16 only; two kernel families carry it -- the halo-staged 3x3 / stride 1 / pad 1 kernel (rows of <= 56 pixels) and the
 * LDS-DMA kernel on 1x1 convolutions -- with Cin <= 512, Cin % 32 == 0, ldx == Cin: ask capmi_igemm_nt_bnact_supported
 * (1 / 2 = which family, 0 = materialise the tensor and call capmi_igemm_nt).  in_act: CAPMI_ACT_RELU or _RELU6.
 * stats as capmi_igemm_nt (same part height).  The vectors are f32 [Cin], 16-byte aligned. */
int capmi_igemm_nt_bnact_supported(const capmi_conv_geom* g, int N, int dtype);
int capmi_igemm_nt_bnact(const void* x_raw, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                         const float* in_mean, const float* in_coef_a, const float* in_offset, int in_act,
                         float* stats, int dtype, void* stream);

/* Convolution + batch-norm statistics + capmi_bn_finalize in ONE launch: conv2d -> batch_norm (train mode) of
 * MobileNetV2.py:88-121 conv_bn_layer up to the saved mean / invstd / coef_a and the running statistics; follow with
 * capmi_bn_apply.  = capmi_igemm_nt(x, w, y, g, N, ldw, ldy, NULL, NULL, 0, NULL, 0, stats, 0, 0, 0, dtype) followed by
 * capmi_bn_finalize(stats, capmi_igemm_nt_stats_part_rows(M, N, K, dtype), M, N, ...), bit for bit: where the convolution's
 * grid is small (bf16, <= 1024 workgroups, stream not capturing) every workgroup stores its statistics part write-through
 * and counts itself in; the LAST one to arrive merges and finalizes (same f64 Chan fold, same fixed order), so the dependent
 * ~10 us finalize launch behind the convolution disappears from the forward chain.  Elsewhere it IS the two calls.
 * stats: the statistics workspace of the two-call form.  Library-owned arrival counters: the contract of capmi_bn_finalize
 * (at most 127 later fused launches in flight next to an unfinished one on a device). */
int capmi_igemm_nt_bnfin(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy, float* stats,
                         const float* scale, float* run_mean, float* run_var, float momentum, float eps, float* saved_mean,
                         float* saved_invstd, float* coef_a, int update_running, int dtype, void* stream);

/* Plain product y[M][N] = x[M][K] . w[N][K]^T with a LONG reduction and a SMALL output (the tied projection's data
 * gradient, dR = dlogits . Emb: model_adaAttention_aic.py:25 backward, [T*B][V] x [V][E] -- 40 output tiles for 256 CUs):
 * K is split over workgroups into f32 slabs in ws (capmi_igemm_nt_splitk_ws_bytes(M, N, K, dtype) bytes; 0 = this shape
 * is not split) and a second launch adds the slabs in a fixed order (deterministic).  No bias / activation / statistics;
 * without enough workspace, or for shapes the split does not pay for, it is capmi_igemm_nt. */
long long capmi_igemm_nt_splitk_ws_bytes(int M, int N, int K, int dtype);
int capmi_igemm_nt_splitk(const void* x, const void* w, void* y, int M, int K, int ldx, int N, int ldw, int ldy,
                          float* ws, long long ws_bytes, int dtype, void* stream);

/* Several independent capmi_igemm_nt products with disjoint outputs (the output-parity classes of a
 * strided convolution's data gradient; the p_hid / sent_emb projections of a decode step,
 * model_adaAttention_aic.py:99,104: small GEMMs that under-fill the chip one at a time), issued
 * together.  Semantics = the calls one after another (statistics off, output in `dtype`; bias may be NULL, act
 * CAPMI_ACT_NONE); eligible groups (bf16, every call on the same LDS-DMA tile shape) run as ONE launch, anything else
 * falls back to per-call launches. */
typedef struct capmi_igemm_nt_call {
    const void* x; const void* w; void* y;
    capmi_conv_geom g;
    int N, ldw, ldy;
    const void* addend; int ld_addend;
    const void* ysaved; int ld_saved; int dact;
    const float* bias; int act;
} capmi_igemm_nt_call;
int capmi_igemm_nt_group(const capmi_igemm_nt_call* calls, int count, int dtype, void* stream);

/* Convolution + INFERENCE batch norm + residual + activation in ONE launch -- the exported inference model's
 * conv2d -> batch_norm(is_test=True) -> (elementwise_add) -> relu/relu6 chain (MobileNetV2.py:99-124 under infer.py:27-31):
 * y = act(coef_a[n] * (conv - mean[n]) + offset[n] (+ res)), capmi_bn_apply's formula applied to the f32 accumulator in
 * the epilogue; the conv output itself never goes to memory.  mean / coef_a from capmi_bn_inference_coef. */
int capmi_igemm_nt_bn(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                      const float* mean, const float* coef_a, const float* offset, const void* res, int ld_res,
                      int act, int dtype, void* stream);

/* Data-gradient GEMM whose OUTPUT completes the gradient of a batch-normalised tensor: capmi_igemm_nt
 * (bias, act, statistics off; output dense, ldy == N) plus, in the same epilogue, the first stage of
 * capmi_bn_bwd_reduce for the `nred` (1 or 2) layers that take this output as their dy (the layer
 * that produced the tensor; a projection shortcut that shares the gradient).  Target q: rx_q = that
 * layer's conv output [rows][N] (indexed like y, scattered rows included), its saved mean / invstd,
 * and ws_q[part][2][N] f32: part p = rows [p*R, (p+1)*R) of this GEMM with
 * R = capmi_igemm_nt_bnred_part_rows(g, N, dtype) (0: not available for this shape -- use
 * capmi_bn_bwd_reduce); ws_q[p][0][n] = sum dz, ws_q[p][1][n] = sum dz*(x-mean)*invstd with dz the
 * value stored to y.  Plain stores, one producer per element: deterministic.  Finish with
 * capmi_bn_bwd_reduce_final.  Replaces one full read of dy and of the conv output per BN layer
 * (fluid batch_norm_grad, model_adaAttention_aic.py:193-195 backward). */
int capmi_igemm_nt_bnred_part_rows(const capmi_conv_geom* g, int N, int dtype);
int capmi_igemm_nt_bnred(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                         const void* addend, int ld_addend, const void* ysaved, int ld_saved, int dact,
                         int nred, const void* rx0, const float* mean0, const float* invstd0, float* ws0,
                         const void* rx1, const float* mean1, const float* invstd1, float* ws1,
                         int dtype, void* stream);

/* The data gradient of a convolution (capmi_igemm_nt with the ReLU mask of the completed tensor as bits: dact carries
 * CAPMI_DACT_BITMASK, `maskbits` from capmi_bn_apply_mask) that ALSO takes the batch-norm backward sums of the layer whose
 * output gradient it completes -- fluid's batch_norm_grad reductions (IC/model/MobileNetV2.py:112-119 backward) inside the
 * conv2d_grad of the consumer.  raw = that layer's conv output [M][N] (the rows of y), mean / invstd its saved statistics:
 * per channel sum_m dz and sum_m dz * (raw - mean) * invstd, dz = the value stored to y (masked, rounded to bf16 -- what
 * capmi_bn_bwd_apply_spread reads back).  Default: the sums are ADDED (f32 atomics) to acc_rows[4][2N] -- the accumulator rows
 * capmi_bn_bwd_apply_spread consumes, zeroed by the caller once per step -- and the layer's capmi_bn_bwd_reduce_spread launch
 * (two tensor reads on the dependency chain) is not needed.  Deterministic mode (capmi_set_deterministic): per-tile parts into
 * parts_ws (ceil(M / R) * 2N floats, R = capmi_igemm_nt_bnsum_part_rows) + the fixed-order second stage into red
 * ([d offset | d scale], as capmi_bn_bwd_reduce leaves it).  R = 0: this shape has no kernel with the sums epilogue (the
 * register-staged kernel, grouped / strided data gradients) -- launch capmi_igemm_nt + capmi_bn_bwd_reduce_spread.  bf16 only. */
int capmi_igemm_nt_bnsum_part_rows(const capmi_conv_geom* g, int N, int dtype);
int capmi_igemm_nt_bnsum(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                         const void* addend, int ld_addend, const void* maskbits, int ld_saved, int dact,
                         const void* raw, const float* mean, const float* invstd, float* acc_rows, float* parts_ws,
                         float* red, int dtype, void* stream);

/* dW[n][k] += sum_m dY[m][n] * A(m,k): weight gradient (f32 output; the caller zeroes dW once per
 * step).  The pixel axis is split over workgroups; each split stores its partial tile to the f32
 * workspace `ws` (capmi_igemm_tn_ws_bytes(M,N,K,dtype) bytes; 0 = not needed) and a second kernel
 * sums the splits in fixed order (deterministic); outputs smaller than a 128x128 tile use f32
 * atomics instead.  Replaces conv2d_grad's filter
 * gradient and mul_grad's weight gradient for the same call sites as capmi_igemm_nt.
 * dY is [M][ldy] in `dtype`. */
long long capmi_igemm_tn_ws_bytes(int M, int N, int K, int dtype);
int capmi_igemm_tn_wgrad(const void* x, const void* dy, float* dw, const capmi_conv_geom* g,
                         int N, int ldy, int lddw, float* ws, long long ws_bytes, int dtype, void* stream);

/* out[n] += sum_m a[m][n] (f32 atomic accumulate): bias gradients (elementwise_add_grad). */
int capmi_colsum(const void* a, int M, int N, int lda, float* out, int dtype, void* stream);

/* Stem im2col: NCHW f32 image (the reference feed, IC/reader.py:45-47) -> [B*Ho*Wo][Kpad]
 * patch matrix in `dtype`, k=(r,q,c), zero padded to Kpad. */
int capmi_im2col_stem(const float* img, void* out, int B, int C, int H, int W, int k, int stride,
                      int pad, int Ho, int Wo, int Kpad, int dtype, void* stream);
/* The stem without a patch matrix: 2x2 space-to-depth of the zero-padded NCHW f32 feed,
 * out[b][bh][bw][(ph*2+pw)*C + c] = img[b][c][2bh+ph-pad][2bw+pw-pad] (0 outside, channels 4C..Cs zero),
 * on which a k x k / stride-2 / pad convolution is a ceil(k/2)^2 / stride-1 / unpadded one with the filter
 * Ws[n][r'][q'][(ph*2+pw)*C + c] = W[n][c][2r'+ph][2q'+pw] (0 where 2r'+ph >= k or 2q'+pw >= k):
 * run it with capmi_igemm_nt / capmi_igemm_tn_wgrad like any other conv (Hb = Ho + (k-1)/2).
 * capmi_s2d_stem_mask_grad zeroes the filter-gradient slots of those structural zeros.
 * Replaces conv2d on the image feed (MobileNetV2.py:28-36 / the build-defined ResNet stem). */
int capmi_s2d_stem(const float* img, void* out, int B, int C, int H, int W, int pad, int Hb, int Wb, int Cs,
                   int dtype, void* stream);
int capmi_s2d_stem_mask_grad(float* dw, int Cout, int C, int k, int Cs, void* stream);

/* Depthwise 3x3 (fluid.layers.conv2d groups=C use_cudnn=False, IC/model/MobileNetV2.py:155-164).
 * w is [3][3][C] f32/bf16 as `dtype`; dw is f32 [3][3][C] (atomic accumulate). */
int capmi_dwconv3x3_fwd(const void* x, const void* w, void* y, int B, int Hi, int Wi, int C,
                        int stride, int Ho, int Wo, int dtype, void* stream);
int capmi_dwconv3x3_bwd_data(const void* dy, const void* w, void* dx, int B, int Hi, int Wi, int C,
                             int stride, int Ho, int Wo, int accumulate, int dtype, void* stream);
int capmi_dwconv3x3_bwd_weight(const void* x, const void* dy, float* dw, int B, int Hi, int Wi, int C,
                               int stride, int Ho, int Wo, int dtype, void* stream);

/* 3x3 stride-2 pad-1 max pool, NHWC (ResNet stem; build-defined extension). idx: uint8 tap. */
int capmi_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int B, int Hi, int Wi, int C,
                           int Ho, int Wo, int dtype, void* stream);
int capmi_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int Hi, int Wi, int C,
                           int Ho, int Wo, int dtype, void* stream);

/* Batch norm, train mode (fluid.layers.batch_norm, IC/model/MobileNetV2.py:112-117) over
 * x [M][C] (M = B*H*W), fused with relu/relu6 (:119) and the residual add (:123-124).
 *   bn_stats   : ws[part][C][2] = exact (mean, sum (x-mean)^2) of every block of
 *                capmi_bn_stats_part_rows(M,C,dtype) rows (two passes over the block, no atomics)
 *   bn_finalize: merges the parts (Chan's formula, f64; two levels when there are many -- ws must
 *                have room for 64 extra parts: [ceil(M/part_rows) + 64][C][2]) into mean / biased
 *                variance over M rows;
 *                writes saved_mean, saved_invstd, coef_a = scale*invstd, and updates the running
 *                stats with momentum (run = m*run + (1-m)*batch)
 *                (more than 64 parts: a merge level first -- in the same launch, its last-arriving workgroup
 *                finalizes; library-owned arrival counters, nothing for the caller to provide or zero.
 *                CONCURRENCY CONTRACT: the counters come from a per-device pool of 128 sets used round-robin in
 *                host call order, so at most 127 LATER capmi_bn_finalize launches of this fused form may be in
 *                flight on a device next to one that has not finished -- on any number of streams.  A caller that
 *                cannot bound that sets CAPMI_BN_FUSE_MERGE=0, or captures the launch in a hipGraph: both take the
 *                two-kernel merge -> finalize path, which has no shared state.)
 *   bn_apply   : y = act(coef_a*(x - mean) + offset (+ res))   (mean subtracted first: a*x + b
 *                with b = offset - a*mean would cancel when |mean| >> std)
 *   bn_bwd_reduce: with dz = dy * act'(y): red[0..C) += sum dz, red[C..2C) += sum dz*xhat; two
 *                stages through the partial-sum workspace ws (capmi_bn_bwd_ws_floats(M,C,dtype)
 *                floats) -- no atomics, deterministic
 *   bn_bwd_apply : dx (+)= scale*invstd*(dz - red0/M - xhat*red1/M); optional dres (+)= dz;
 *                  dscale/doffset = red1/red0 are read by the optimizer straight from `red`. */
int capmi_bn_stats_part_rows(int M, int C, int dtype);
int capmi_bn_stats(const void* x, int M, int C, float* ws, int dtype, void* stream);
int capmi_bn_finalize(float* ws, int part_rows, int M, int C, const float* scale,
                      float* run_mean, float* run_var, float momentum, float eps,
                      float* saved_mean, float* saved_invstd, float* coef_a,
                      int update_running, void* stream);
int capmi_bn_apply(const void* x, const float* saved_mean, const float* coef_a, const float* offset,
                   const void* res, void* y, int M, int C, int act, int dtype, void* stream);
/* capmi_bn_apply + the activation-derivative bit mask of the STORED output: mask[m][c >> 3] bit (c & 7) = 1 where
 * act'(y[m][c]) = 1 (relu: y > 0; relu6: 0 < y < 6) -- batch_norm -> relu of MobileNetV2.py:112-121 with what its backward
 * needs of the output kept at one bit per element.  bf16, C % 8 == 0, act relu / relu6; mask: M * C / 8 bytes.
 * Consumer: dact | CAPMI_DACT_BITMASK (above). */
int capmi_bn_apply_mask(const void* x, const float* saved_mean, const float* coef_a, const float* offset, const void* res,
                        void* y, uint8_t* mask, int M, int C, int act, int dtype, void* stream);
/* capmi_bn_bwd_reduce / capmi_bn_bwd_apply without the dependent second-stage launch between them (~50 per train step, ~9 us
 * each on the critical chain): the reduction ADDS its block totals (f32 atomics; a wave instruction covers 256 contiguous
 * bytes, at most 1/8 of the grid per address) into eight accumulator rows acc8[8][2C] -- zeroed by the caller once per
 * step, 16-byte aligned -- and the apply kernel sums the rows in its prologue and adds the result to red
 * ([d offset | d scale]).  The summation order inside a row is not fixed: in deterministic mode (capmi_deterministic) both
 * calls run the two-stage form through ws / red instead, which is why they take both sets of buffers. */
int capmi_bn_bwd_reduce_spread(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                               float* ws, float* red, float* acc8, int M, int C, int act, int dtype, void* stream);
int capmi_bn_bwd_apply_spread(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                              const float* scale, float* red, const float* acc8, void* dx, int dx_accumulate, void* dres,
                              int dres_accumulate, int M, int C, int act, int dtype, void* stream);
/* The same pair for a layer whose activated output feeds a 3x3 / stride-2 max pool and nothing else (the ResNet stem:
 * conv -> batch_norm -> relu -> pool2d, MobileNetV2.py:88-121 conv_bn_layer + the pool of the build-defined ResNet encoders):
 * capmi_maxpool3x3s2_bwd + capmi_bn_bwd_reduce_spread + capmi_bn_bwd_apply_spread WITHOUT materialising the pool's input
 * gradient -- both kernels gather it from dpool [B,Ho,Wo,C] (the gradient of the pool's OUTPUT, a quarter of the size) and
 * the forward pass's uint8 argmax map idx; x / y [B,Hi,Wi,C] are the layer's conv output and activated output.  act must
 * be CAPMI_ACT_RELU or CAPMI_ACT_RELU6.  dy_scratch [B,Hi,Wi,C] is written only in deterministic mode, where the pair runs
 * the three-launch path (pool backward into dy_scratch, two-stage sums through ws / red). */
int capmi_bn_bwd_reduce_pool(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* saved_mean,
                             const float* saved_invstd, float* ws, float* red, float* acc8, void* dy_scratch, int B, int Hi, int Wi,
                             int C, int Ho, int Wo, int act, int dtype, void* stream);
int capmi_bn_bwd_apply_pool(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* saved_mean,
                            const float* saved_invstd, const float* scale, float* red, const float* acc8, const void* dy_scratch,
                            void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream);
/* The stem of the ResNet encoders as the forward pass runs it on a bf16 engine: capmi_bn_stat_apply + capmi_maxpool3x3s2_fwd as ONE
 * launch that never writes the activated tensor [B,Hi,Wi,C] -- it has one reader, the pool, and the backward pass needs only the sign
 * of its elements.  x: the conv output; statistics as capmi_bn_stat_apply takes them (stat_rows / shift; parts in deterministic mode);
 * pooled [B,Ho,Wo,C] and idx (uint8 argmax map) as capmi_maxpool3x3s2_fwd writes them -- bit for bit the two-launch result for the
 * same statistics (every element is normalised, activated and rounded to bf16 before it is compared; first maximum wins).
 * y_scratch [B,Hi,Wi,C]: written in deterministic mode only (capmi_bn_finalize + capmi_bn_apply + capmi_maxpool3x3s2_fwd).
 * The backward pair for a layer run this way: capmi_bn_bwd_reduce_pool_x / capmi_bn_bwd_apply_pool_x = the _pool entry points with
 * the activation's derivative formed from x, coef_a, saved_mean and offset (y = y_scratch: read in deterministic mode only). */
int capmi_bn_stat_apply_pool(const void* x, float* parts, int part_rows, const float* stat_rows, const float* shift, int B, int Hi, int Wi, int C,
                             int Ho, int Wo, const float* scale, const float* offset, float* run_mean, float* run_var, float momentum, float eps,
                             float* saved_mean, float* saved_invstd, float* coef_a, int update_running, void* y_scratch, void* pooled, uint8_t* idx,
                             int act, int dtype, void* stream);
int capmi_bn_bwd_reduce_pool_x(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                               const float* saved_mean, const float* saved_invstd, float* ws, float* red, float* acc8, void* dy_scratch, int B,
                               int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream);
int capmi_bn_bwd_apply_pool_x(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                              const float* saved_mean, const float* saved_invstd, const float* scale, float* red, const float* acc8,
                              const void* dy_scratch, void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream);
/* Inference mode (fluid batch_norm is_test=True, the exported model of infer.py): mean = running mean,
 * coef_a = scale / sqrt(running variance + eps); follow with capmi_bn_apply. */
int capmi_bn_inference_coef(const float* scale, const float* run_mean, const float* run_var, float eps, float* mean,
                            float* coef_a, int C, void* stream);
/* The same for every batch-norm layer of a model in one launch.  jobs: device array of
 *   struct { const float* scale, *run_mean, *run_var; float* mean, *coef_a; int64 C; }   (48 bytes per layer);
 * max_c = the largest C among them. */
int capmi_bn_inference_coef_batched(const void* jobs, int njobs, int max_c, float eps, void* stream);
/* capmi_bn_finalize + capmi_bn_apply in ONE launch (plus the merge launch for > 64 parts): every apply workgroup
 * merges the statistic groups of its own channels (f64, Chan) before normalising its rows; saved mean / invstd and
 * the running statistics are written by the first workgroup row.  Same results as the two calls.  Measured
 * SLOWER in the model (-6 %: the per-workgroup merge delays every workgroup's first load by more than the
 * saved launch); the engine uses capmi_bn_finalize + capmi_bn_apply. */
int capmi_bn_finalize_apply(float* ws, int part_rows, int M, int C, const float* scale, const float* offset, float* run_mean,
                            float* run_var, float momentum, float eps, float* saved_mean, float* saved_invstd,
                            int update_running, const void* x, const void* res, void* y, int act, int dtype, void* stream);
/* conv2d -> batch_norm (train mode) of MobileNetV2.py:99-117 WITHOUT the merge + finalize launch between the convolution and
 * the normalisation (bf16).  capmi_igemm_nt_stat is capmi_igemm_nt whose epilogue ADDS the per-column sums of d = v - shift[n] and
 * d^2 over its f32 accumulators (f32 atomics) to stat_rows[4][2N], zeroed by the caller once per step; `shift` [N] is any
 * per-channel estimate of the mean -- the engine hands over the previous step's batch mean (a copy: NOT the saved_mean buffer
 * of this step), which keeps the one-pass variance free of cancellation.  capmi_bn_stat_apply forms mean = shift + E[d],
 * var = E[d^2] - E[d]^2 (clamped at 0), invstd and coef_a from the rows in every workgroup's prologue, applies
 * y = act(coef_a * (x - mean) + offset (+ res)) -- capmi_bn_apply's formula; with `mask` also capmi_bn_apply_mask's bits --
 * and its first row block writes saved_mean / saved_invstd / coef_a and updates the running statistics (momentum as
 * capmi_bn_finalize).  Deterministic mode (capmi_set_deterministic) and nothing else switches BOTH entry points to the exact
 * path: (mean, M2) parts of part_rows = capmi_igemm_nt_stats_part_rows rows into `parts` (the capmi_bn_finalize workspace),
 * then capmi_bn_finalize + capmi_bn_apply[_mask].  capmi_igemm_nt_stat_supported = 0: the shape has no kernel with the sums
 * epilogue (narrow or ragged outputs, f32): use capmi_igemm_nt + capmi_bn_finalize + capmi_bn_apply. */
int capmi_igemm_nt_stat_supported(const capmi_conv_geom* g, int N, int dtype);
int capmi_igemm_nt_stat(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                        float* parts, float* stat_rows, const float* shift, int dtype, void* stream);
int capmi_bn_stat_apply(const void* x, float* parts, int part_rows, const float* stat_rows, const float* shift, int M, int C, const float* scale,
                        const float* offset, float* run_mean, float* run_var, float momentum, float eps, float* saved_mean,
                        float* saved_invstd, float* coef_a, int update_running, const void* res, void* y, uint8_t* mask, int act,
                        int dtype, void* stream);
int capmi_bn_bwd_ws_floats(int M, int C, int dtype);
int capmi_bn_bwd_reduce(const void* dy, const void* x, const void* y, const float* saved_mean,
                        const float* saved_invstd, float* ws, float* red, int M, int C, int act,
                        int dtype, void* stream);
/* Second stage alone, for partial sums from capmi_igemm_nt_bnred: red[0..C) += sum_p ws[p][0][c],
 * red[C..2C) += sum_p ws[p][1][c]. */
int capmi_bn_bwd_reduce_final(const float* ws, int nparts, int C, float* red, void* stream);
int capmi_bn_bwd_apply(const void* dy, const void* x, const void* y, const float* saved_mean,
                       const float* saved_invstd, const float* scale, const float* red,
                       void* dx, int dx_accumulate, void* dres, int dres_accumulate,
                       int M, int C, int act, int dtype, void* stream);

/* Elementwise helpers. add_act: y = act(a + b) (MobileNetV2 shortcut, :123-124);
 * add_act_bwd: d (+)= dy * act'(y).  mean_rows: out[b][c] = mean_k x[b][k][c] (reduce_mean,
 * model_adaAttention_aic.py:197) and its broadcast gradient. */
int capmi_add_act(const void* a, const void* b, void* y, int64_t n, int act, int dtype, void* stream);
int capmi_act_bwd(const void* dy, const void* y, void* dx, int accumulate, int64_t n, int act, int dtype, void* stream);
int capmi_mean_rows(const void* x, void* out, int B, int K, int C, int dtype, void* stream);
int capmi_mean_rows_bwd(const void* dout, void* dx, int B, int K, int C, int dtype, void* stream);

/* The caption feed of a train step (reader.py:45-47: int64 [B][L], <start> first, right-padded with 0) as the decoder reads it:
 * ids[t*B + b] = caption[b][t] (the source words caption[:, :-1], model_adaAttention_aic.py:164, time-major :60) and
 * tgt[t*B + b] = caption[b][t + 1] (the targets caption[:, 1:], :163), t = 0 .. L-2.  One launch. */
int capmi_caption_feed(const int64_t* caption, int64_t* ids, int64_t* tgt, int B, int L, void* stream);

/* fluid.embedding(padding_idx) (model_adaAttention_aic.py:28-32): out[m][0..E) = table[ids[m]]
 * (zeros for padding_idx), written with row stride ldo.  bwd: dtable[ids[m]] += dout[m] (f32
 * atomics; padding rows skipped). */
int capmi_embedding_fwd(const int64_t* ids, const void* table, void* out, int M, int E, int V,
                        int ldo, int padding_idx, int dtype, void* stream);
int capmi_embedding_bwd(const int64_t* ids, const void* dout, float* dtable, int M, int E, int V,
                        int ldo, int padding_idx, int dtype, void* stream);
/* dst[t][b][col0 + j] = src[b][j] for t in [0,T): the concat([word_emb, global_img_feat]) of
 * model_adaAttention_aic.py:86 batched over time; bwd: dsrc[b][j] = sum_t ddst[t][b][col0+j]. */
int capmi_bcast_rows(const void* src, void* dst, int T, int B, int H, int ldd, int col0, int dtype, void* stream);
int capmi_bcast_rows_bwd(const void* ddst, void* dsrc, int T, int B, int H, int ldd, int col0, int dtype, void* stream);

/* lstm_unit op (model_adaAttention_aic.py:87-88) on pre-computed gate pre-activations
 * gates[B][4H] (blocks i,f,o,g; forget_bias 0): c = s(f)c' + s(i)tanh(g); h = s(o)tanh(c). */
int capmi_lstm_cell_fwd(const void* gates, const void* c_prev, void* h, void* c, int B, int H,
                        int dtype, void* stream);
/* Backward of the cell: dh = total gradient on h_t, dc_in = gradient on c_t from later steps and
 * the sentinel (NULL = 0); c_prev NULL = zero state.  Writes dgates [B][4H]; dc_prev (optional)
 * receives sigmoid(f)*dc_total, added to its old contents when dc_prev_accumulate. */
int capmi_lstm_cell_bwd(const void* gates, const void* c_prev, const void* c, const void* dh,
                        const void* dc_in, void* dgates, void* dc_prev, int dc_prev_accumulate,
                        int B, int H, int dtype, void* stream);
/* One time step of the recurrence in ONE launch (B <= 64, H % 128 == 0, H >= 256: capmi_lstm_step_supported):
 *   fwd: gates[b][g*H+u] (in: the input part x_t.Wx^T + b, out: the full pre-activations, same precision as
 *        capmi_igemm_nt would store) += h_prev[b][:] . wh[g*H+u][:]   (wh: [4H] rows of stride ldw, the recurrent
 *        columns of lstm_w), then the lstm_unit cell (gate order i,f,o,g) -> c, h.   c_prev NULL = zeros.
 *   bwd: dh = dh_in + dgates_t . whT^T (whT: [H] rows of stride ldwT, reduction over the 4H gates), then the cell
 *        backward of the PREVIOUS step: gates_prev, c_prev (its incoming cell state, NULL = zeros), c (its outgoing
 *        one), dc_in -> dgates_prev, dc_prev (+= when dc_prev_accumulate; NULL = not needed).
 * Same arithmetic and rounding points as capmi_igemm_nt + capmi_lstm_cell_{fwd,bwd} (model_adaAttention_aic.py:87-88). */
int capmi_lstm_step_supported(int B, int H, int dtype);
int capmi_lstm_step_fwd(const void* h_prev, const void* wh, int ldw, void* gates, const void* c_prev, void* h, void* c,
                        int B, int H, int dtype, void* stream);
int capmi_lstm_step_bwd(const void* dgates_t, const void* whT, int ldwT, const void* dh_in, const void* gates_prev,
                        const void* c_prev, const void* c, const void* dc_in, void* dgates_prev, void* dc_prev,
                        int dc_prev_accumulate, int B, int H, int dtype, void* stream);
/* The whole recurrence of ONE LSTM layer in one launch per direction (the While loop of model_adaAttention_aic.py:75-127
 * reduced to its only sequential part, `lstm_unit` :87-88): the grid of the fused step kernels stays resident for all
 * T steps, every wave keeps its slice of the recurrent weights in registers, and the steps are separated by a grid
 * barrier (write-through hand-off + one agent-scope arrival counter) instead of 2T kernel boundaries.  Results equal
 * the per-step launches up to FMA contraction in the cell (same k-split, MFMA order and rounding points).
 *   fwd: hbuf / cbuf hold T+1 row blocks [B][H]; block 0 is the initial state (zeros, :63), blocks 1..T are written;
 *        gates [T][B][4H] in: x_t.Wx^T + b, out: the full pre-activations.  wh: the recurrent columns of lstm_w,
 *        [4H] rows of stride ldw.
 *   bwd: gates / cbuf as fwd left them; dhbuf blocks 1..T = d loss / d h_t from everything except the recurrence (read
 *        only); dcbuf blocks 1..T = d loss / d c_t from outside when dc_outside != 0 (the cell's own d c_{t-1} is then
 *        added into block t-1), scratch otherwise; whT: [H] rows of stride ldwT (reduction over the 4H gates);
 *        writes dgates [T][B][4H].
 * sync: 16 bytes of device memory owned by the caller, zeroed by the call; word 1 != 0 after the launch means a grid
 * barrier gave up waiting (bounded spin: the launch always drains) and the results are invalid.
 * capmi_lstm_seq_supported: B <= 64 and H in {256, 384, 512, 768, 1024} (bf16) / H = 256 (f32). */
int capmi_lstm_seq_supported(int B, int H, int T, int dtype);
int capmi_lstm_seq_fwd(void* hbuf, const void* wh, int ldw, void* gates, void* cbuf, int B, int H, int T, void* sync,
                       int dtype, void* stream);
int capmi_lstm_seq_bwd(const void* gates, const void* cbuf, const void* whT, int ldwT, const void* dhbuf, void* dcbuf,
                       void* dgates, int dc_outside, int B, int H, int T, void* sync, int dtype, void* stream);
/* visual sentinel (:91-92): s = sigmoid(sgpre) * tanh(c); bwd -> dsgpre, dc. */
int capmi_sentinel_fwd(const void* sgpre, const void* c, void* s, int64_t n, int dtype, void* stream);
int capmi_sentinel_bwd(const void* ds, const void* sgpre, const void* c, void* dsgpre, void* dc,
                       int64_t n, int dtype, void* stream);

/* Adaptive attention over the K image slots + the sentinel slot (model_adaAttention_aic.py:
 * 97-113).  Rows are time-major: row m = t*B + b reads image b.
 *   slots = 0 ('singleton', reference-faithful quirk Q1: softmax over the size-1 axis, alpha==1)
 *           ctx = (sum_k Vt[b,k,:] + s) / (K+1)
 *   slots = 1 (intended attention): z = tanh([Ve[b]; se] + q); e = z.w10 + b10;
 *           alpha = softmax over the K+1 slots; ctx = mean_k(alpha_k * [Vt[b]; s])
 * out[m] = ctx + p[m] (the fc input of :115).  alpha [M][K+1] f32 is saved for backward. */
int capmi_ada_attention_fwd(const void* Ve, const void* Vt, const void* q, const void* se,
                            const void* s, const void* p, const void* w10, const float* b10,
                            void* out, float* alpha, int T, int B, int K, int H, int slots,
                            int dtype, void* stream);
/* Given dout [M][H]: ds (written), dVt/dVe [B][K][H] (written), dq, dse [M][H] (written; slots
 * only), dw10 [H] / db10 [1] (f32 atomic accumulate; slots only).  de [M][K+1] f32 workspace. */
int capmi_ada_attention_bwd(const void* Ve, const void* Vt, const void* q, const void* se,
                            const void* s, const void* w10, const float* alpha, const void* dout,
                            void* ds, void* dVt, void* dVe, void* dq, void* dse,
                            float* dw10, float* db10, float* de,
                            int T, int B, int K, int H, int slots, int dtype, void* stream);

/* softmax_with_cross_entropy + mask/sum/div (model_adaAttention_aic.py:165-182,205-212) over
 * f32 logits [M][V]: row_loss[m] = (lse - logit[target]) * (target != padding_idx);
 * loss_out[0] = sum row_loss / count(mask) computed by capmi_xent_finalize; bwd writes
 * dlogits = (softmax - onehot) * mask / count in `dtype`, row stride ldd >= V (columns V..ldd
 * are zero-filled so the padded matrix can feed the MFMA kernels); ld = logits row stride. */
int capmi_softmax_xent_fwd(const float* logits, const int64_t* target, float* row_loss,
                           float* row_lse, int M, int V, int ld, int padding_idx, void* stream);
int capmi_xent_finalize(const float* row_loss, const int64_t* target, float* loss_out,
                        float* count_out, int M, int padding_idx, void* stream);
int capmi_softmax_xent_bwd(const float* logits, const int64_t* target, const float* row_lse,
                           const float* count, void* dlogits, int M, int V, int ld, int ldd,
                           int padding_idx, int dtype, void* stream);
/* layers.argmax (:120): lowest index on ties; ids_out int64 [M], also f32 copy (quirk Q2). */
int capmi_argmax(const float* logits, int64_t* ids_out, float* ids_f32, int ld_f32, int M, int V, int ld, void* stream);

/* One decode step's state plumbing in two launches instead of five (eval branches of Decoder.call,
 * model_adaAttention_aic.py:84-92 under :119-123; beam search: this build's extension).
 * capmi_decode_prep: row r of xh [R][ldx] (= [embedding | global feature | h_prev]) gets the embedding of ids[r] in columns
 *   0..E-1 (zero row for padding_idx / out-of-range ids, :28-32) and h_src[rows ? rows[r] : r] in columns h_col..h_col+H-1
 *   -- the survivors' hidden state of a beam step, or the previous step's own (rows NULL).  The columns in between (the
 *   global image feature, :86) are written once before the loop and left alone.  With the LSTM's and the sentinel gate's
 *   weights stacked [4H + H][E + 2H], ONE capmi_igemm_nt on xh then gives the lstm_unit's gate pre-activations and the
 *   sentinel gate's (:87-91: fc over the concatenated input, exactly as the reference's lstm_unit concatenates).
 * capmi_lstm_cell_sentinel_fwd: gs [R][ld_gs] = [i | f | o | g | sentinel gate] pre-activations; c_prev = c_src[rows ?
 *   rows[r] : r]; writes c, h (lstm_unit, forget_bias 0, :87-88) and s = sigmoid(sentinel gate) * tanh(c) (:91-92). */
int capmi_decode_prep(const int64_t* ids, const void* table, const void* h_src, const int* rows, void* xh, int R, int E, int H,
                      int V, int ldx, int h_col, int padding_idx, int dtype, void* stream);
int capmi_lstm_cell_sentinel_fwd(const void* gs, int ld_gs, const void* c_src, const int* rows, void* h, void* c, void* s,
                                 int R, int H, int dtype, void* stream);

/* Beam-search decode (BUILD-DEFINED extension of the eval graph, BASELINE cfg 5; infer.py only has the greedy loop).
 * Rows are beam-major: row k*B + b holds hypothesis k of image b.  One step: capmi_beam_step takes the f32 logits
 * [beam*B][ld] of the current hypotheses and their scores [beam][B] (sum of log-softmax probabilities) and keeps, per
 * image, the `beam` best of the beam x V continuations (ties: lower beam index, then lower token id): new scores,
 * parents / tokens [beam][B] of this step, the next input ids and the rows to gather the LSTM state from
 * (capmi_gather_rows).  cand_val / cand_idx [beam*B][beam] and lse [beam*B] are scratch.  No early stop and no
 * special casing of <stop> (quirk Q5 carried over): beam = 1 is exactly the greedy loop.  capmi_beam_backtrack walks
 * the parents [Ti][beam][B] back from the best final hypothesis and writes float32 ids [B][Ti] (quirk Q2). beam <= 8. */
int capmi_beam_step(const float* logits, int V, int ld, int B, int beam, const float* score_in, float* score_out,
                    float* cand_val, int* cand_idx, float* lse, int* parents, int* tokens, int64_t* next_ids,
                    int* gather_rows, void* stream);
int capmi_gather_rows(const void* src, const int* rows, void* dst, int n, int H, int dtype, void* stream);
int capmi_beam_backtrack(const int* tokens, const int* parents, float* out_ids_f32, int Ti, int B, int beam, void* stream);

/* fluid.optimizer.Adam, Paddle-1.8 form, over one flat f32 range (IC/train.py:26-31,45) with
 * optional GradientClipByValue (:42-43; clip <= 0 disables):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr_t * m / (sqrt(v) + eps).   lr_t is computed on the host and passed in.
 * grad_scale multiplies g first (1/N of ParallelExecutor's CoeffNumDevice, IC/train.py:121-124). */
int capmi_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1,
               float b2, float eps, float clip, float grad_scale, void* stream);
/* capmi_adam followed by capmi_cast of the same range to bf16, as one pass: low16[i] = bf16(updated p[i]) -- the weight shadow
 * the bf16 kernels read -- written from the registers of the update (IC/train.py:26-31,45: the adam op; the shadow is this
 * build's bf16 storage of the parameters). */
int capmi_adam_shadow(float* p, const float* g, float* m, float* v, void* low16, int64_t n, float lr_t, float b1, float b2,
                      float eps, float clip, float grad_scale, void* stream);
/* capmi_adam with the gradient read from a bf16 buffer (widened to f32 on load): the consumer of a bucket that was
 * all-reduced in bf16 (capmi_allreduce_bucket_bf16).  Moments and master weights stay f32. */
int capmi_adam_g16(float* p, const void* g16, float* m, float* v, int64_t n, float lr_t, float b1,
                   float b2, float eps, float clip, float grad_scale, void* stream);
/* Weight shadows for the MFMA kernels: capmi_cast = elementwise f32 -> `dtype`;
 * capmi_weight_dgrad_form: W [N][kh][kw][C] f32 -> the data-gradient operand [C][kh][kw][ldt]
 * (taps flipped; columns N..ldt zero; ldt > N only for 1x1 weights). */
int capmi_cast(const float* src, void* dst, int64_t n, int dtype, void* stream);
int capmi_weight_dgrad_form(const float* w, void* wt, int N, int kh, int kw, int C, int ldt, int dtype, void* stream);
/* The same for every GEMM weight of a model in one launch.  jobs: device array of
 *   struct { int64 src_off, dst_off; int32 N, kh, kw, C, ldt, first, okh, okw; int8 rmap[4], qmap[4]; }
 * (56 bytes, one per run of 2 tiles of one weight's output -- a tile is 64 n x 64 c of one tap,
 * tile index = (tap * ceil(C/64) + ct) * ceil(ldt/64) + nt, `first` = the run's first tile;
 * offsets in elements into `flat` (f32) / `shadow` (`dtype`)).  Output is [C][okh][okw][ldt] with tap (r',q') taken from source tap
 * (rmap[r'], qmap[q']): the plain data-gradient form is okh=kh, rmap[r'] = kh-1-r'; the parity
 * classes of a strided convolution use a subset of the taps. */
int capmi_weight_dgrad_form_batched(const float* flat, void* shadow, const void* jobs, int njobs, int dtype, void* stream);
int capmi_fill_f32(float* p, float value, int64_t n, void* stream);

/* Gradient all-reduce of the data-parallel step.  Replaces the implicit NCCL all-reduce inside
 * `fluid.ParallelExecutor(use_cuda=True, loss_name='loss', ...)` (IC/train.py:121-124, ReduceStrategy.AllReduce:
 * per-parameter gradients SUMMED over the devices; the 1/N of GradientScaleStrategy.CoeffNumDevice is the
 * grad_scale of capmi_adam).  One communicator per process (one process per GPU), RCCL over xGMI:
 *   rank 0 calls capmi_comm_unique_id and hands the CAPMI_COMM_ID_BYTES bytes to every rank by any host channel
 *   (the Python host uses torch.distributed's store); every rank then calls capmi_comm_init on its current device.
 * capmi_allreduce_bucket: in-place f32 sum of buf[0..n) -- a contiguous bucket of the flat gradient buffer -- enqueued
 * on `stream`; like every entry point it only enqueues.  The library uses the librccl already mapped into the process
 * (the one behind torch.distributed's "nccl" backend), never a second copy. */
#define CAPMI_COMM_ID_BYTES 128
int capmi_comm_unique_id(void* id_out);
int capmi_comm_init(void** comm, int nranks, int rank, const void* id);
/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank): bench.py prints it next to a multi-GPU
 * number, so that the record proves how many ranks the all-reduce really spanned. */
int capmi_comm_count(void* comm, int* nranks, int* rank);
int capmi_comm_destroy(void* comm);
int capmi_allreduce_bucket(void* comm, float* buf, int64_t n, void* stream);
/* The same on a bf16 copy of the bucket (capmi_cast of the f32 gradients into a staging buffer; capmi_adam_g16 reads the sum
 * back): half the xGMI bytes -- SURVEY.md section 8(e) budgets the exchange in bf16 (73 MB at BASELINE cfg 2).  The ring
 * rounds every partial sum to bf16; the f32 form above is the reference-precision path. */
int capmi_allreduce_bucket_bf16(void* comm, void* buf16, int64_t n, void* stream);

/* Launch plans.  A train step is a fixed sequence of ~650 of the entry points above on a few HIP streams ("lanes":
 * 0 = the dependency chain, 1 = weight gradients and other off-chain work, 2 = communication + optimizer).  The host
 * packs it once into a table of capmi_launch rows and replays it with ONE call per step -- the role
 * `train_exe.run(feed, fetch_list)` (IC/train.py:139) plays over Paddle's op list.
 *   kind CAPMI_PLAN_LAUNCH: entry `entry` (index into capmi_plan_entry_name) with `nargs` argument slots -- the entry
 *     point's arguments in order WITHOUT its trailing stream, 8 bytes each: pointers (device pointers, or host pointers
 *     to capmi_conv_geom / capmi_igemm_nt_call that the caller keeps alive), integers (two's complement), floats (IEEE
 *     bits in the low 4 bytes) -- enqueued on streams[lane];
 *   kind CAPMI_PLAN_RECORD / CAPMI_PLAN_WAIT: args[0] = an event of capmi_event_create, recorded on / awaited by
 *     streams[lane].
 * Returns 0, or the failing row's error (capmi_last_error names the row).  The host may patch argument slots between
 * runs (the per-step Adam step size). */
enum { CAPMI_PLAN_LAUNCH = 0, CAPMI_PLAN_RECORD = 1, CAPMI_PLAN_WAIT = 2 };
#define CAPMI_PLAN_MAX_ARGS 32
typedef struct {
    int32_t kind, lane, entry, nargs;
    uint64_t args[CAPMI_PLAN_MAX_ARGS];
} capmi_launch;
int capmi_plan_entry_count(void);
const char* capmi_plan_entry_name(int i);
int capmi_plan_entry_nargs(int i);
int capmi_plan_run(const capmi_launch* table, int n, void* const* streams, int nstreams);

#ifdef __cplusplus
}
#endif
#endif /* CAPMI_H */
