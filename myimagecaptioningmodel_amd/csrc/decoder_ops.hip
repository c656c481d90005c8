// Decoder-side kernels for gfx950: embedding gather/scatter, LSTM cell, visual sentinel, adaptive
// attention (LDS-staged, wave reductions), masked softmax cross-entropy, argmax, column sums.
// Rows of every [M][*] operand are time-major (m = t*B + b).
#include "common.h"

// ------------------------------------------------------------------ embedding
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table, T* out,
                                                            int M, int cpr, int V, int ldo, int padding_idx) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * cpr) return;
    int m = e / cpr, cc = e % cpr;
    int64_t id = ids[m];
    Vec<T> v = (id == padding_idx || id < 0 || id >= V) ? vzero<T>() : vload<T>(table + (id * cpr + cc) * VEC);
    vstore<T>(out + (int64_t)m * ldo + cc * VEC, v);
}
extern "C" int capmi_embedding_fwd(const int64_t* ids, const void* table, void* out, int M, int E, int V, int ldo,
                                   int padding_idx, int dtype, void* stream) {
    CAPMI_CHECK(ids && table && out, "capmi_embedding_fwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_embedding_fwd", {
        CAPMI_CHECK(E % Vec<T>::N == 0 && ldo % Vec<T>::N == 0, "capmi_embedding_fwd: E/ldo not multiples of the vector width");
        int cpr = E / Vec<T>::N;
        hipLaunchKernelGGL(embedding_fwd_kernel<T>, dim3(cdiv((int64_t)M * cpr, 256)), dim3(256), 0, (hipStream_t)stream, ids, (const T*)table, (T*)out, M, cpr, V, ldo, padding_idx);
    });
    CAPMI_LAUNCH_CHECK("capmi_embedding_fwd");
    return 0;
}
// caption [B][L] (the reader's feed, reader.py:45-47) -> time-major source words ids[t*B + b] = caption[b][t] (caption[:, :-1],
// model_adaAttention_aic.py:164,60) and targets tgt[t*B + b] = caption[b][t + 1] (caption[:, 1:], :163), t < L - 1: one launch
// instead of two slice + transpose + copy chains of the host framework in front of every step
__global__ __launch_bounds__(256) void caption_feed_kernel(const int64_t* __restrict__ caption, int64_t* ids, int64_t* tgt, int B, int L) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (L - 1) * B) return;
    const int t = e / B, b = e - t * B;
    ids[e] = caption[(int64_t)b * L + t];
    tgt[e] = caption[(int64_t)b * L + t + 1];
}
extern "C" int capmi_caption_feed(const int64_t* caption, int64_t* ids, int64_t* tgt, int B, int L, void* stream) {
    CAPMI_CHECK(caption && ids && tgt && B >= 1 && L >= 2, "capmi_caption_feed: bad arguments");
    hipLaunchKernelGGL(caption_feed_kernel, dim3(cdiv((int64_t)(L - 1) * B, 256)), dim3(256), 0, (hipStream_t)stream, caption, ids, tgt, B, L);
    CAPMI_LAUNCH_CHECK("capmi_caption_feed");
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout, float* dtable,
                                                            int M, int E, int V, int ldo, int padding_idx) {
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * E) return;
    int m = e / E, j = e % E;
    int64_t id = ids[m];
    if (id == padding_idx || id < 0 || id >= V) return;
    atomicAdd(&dtable[id * E + j], to_f32(dout[(int64_t)m * ldo + j]));
}
// Deterministic form (capmi_deterministic): workgroup m owns table row ids[m] iff m is the FIRST row with that id; the
// owner adds the rows m, m' > m, ... with the same id in increasing row order and is the only writer of its table row.
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_det_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout, float* dtable,
                                                                int M, int E, int V, int ldo, int padding_idx) {
    const int m = blockIdx.x;
    const int64_t id = ids[m];
    if (id == padding_idx || id < 0 || id >= V) return;
    for (int p = 0; p < m; ++p)
        if (ids[p] == id) return;               // an earlier row owns this id (uniform branch: ids are read by every lane alike)
    for (int j = threadIdx.x; j < E; j += 256) {
        float acc = to_f32(dout[(int64_t)m * ldo + j]);
        for (int p = m + 1; p < M; ++p)
            if (ids[p] == id) acc += to_f32(dout[(int64_t)p * ldo + j]);
        dtable[id * E + j] += acc;
    }
}
extern "C" int capmi_embedding_bwd(const int64_t* ids, const void* dout, float* dtable, int M, int E, int V, int ldo,
                                   int padding_idx, int dtype, void* stream) {
    CAPMI_CHECK(ids && dout && dtable, "capmi_embedding_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_embedding_bwd", {
        if (capmi_deterministic())
            hipLaunchKernelGGL(embedding_bwd_det_kernel<T>, dim3(M), dim3(256), 0, (hipStream_t)stream, ids, (const T*)dout, dtable, M, E, V, ldo, padding_idx);
        else
            hipLaunchKernelGGL(embedding_bwd_kernel<T>, dim3(cdiv((int64_t)M * E, 256)), dim3(256), 0, (hipStream_t)stream, ids, (const T*)dout, dtable, M, E, V, ldo, padding_idx);
    });
    CAPMI_LAUNCH_CHECK("capmi_embedding_bwd");
    return 0;
}

// ------------------------------------------------------------------ broadcast of the global image feature over time
template <typename T>
__global__ __launch_bounds__(256) void bcast_rows_kernel(const T* __restrict__ src, T* dst, int T_, int B, int cpr, int ldd, int col0) {
    constexpr int VEC = Vec<T>::N;
    int64_t n = (int64_t)T_ * B * cpr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        int cc = (int)(e % cpr);
        int64_t row = e / cpr;
        int b = (int)(row % B);
        vstore<T>(dst + row * ldd + col0 + cc * VEC, vload<T>(src + ((int64_t)b * cpr + cc) * VEC));
    }
}
extern "C" int capmi_bcast_rows(const void* src, void* dst, int T_, int B, int H, int ldd, int col0, int dtype, void* stream) {
    CAPMI_CHECK(src && dst, "capmi_bcast_rows: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_bcast_rows", {
        CAPMI_CHECK(H % Vec<T>::N == 0 && ldd % Vec<T>::N == 0 && col0 % Vec<T>::N == 0, "capmi_bcast_rows: misaligned");
        int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(bcast_rows_kernel<T>, dim3(ew_grid((int64_t)T_ * B * cpr)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, T_, B, cpr, ldd, col0);
    });
    CAPMI_LAUNCH_CHECK("capmi_bcast_rows");
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void bcast_rows_bwd_kernel(const T* __restrict__ ddst, T* dsrc, int T_, int B, int cpr, int ldd, int col0) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int t = 0; t < T_; ++t) {
        Vec<T> g = vload<T>(ddst + ((int64_t)t * B + b) * ldd + col0 + cc * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += g.get(v);
    }
    Vec<T> ov;
#pragma unroll
    for (int v = 0; v < VEC; ++v) ov.set(v, acc[v]);
    vstore<T>(dsrc + (int64_t)e * VEC, ov);
}
extern "C" int capmi_bcast_rows_bwd(const void* ddst, void* dsrc, int T_, int B, int H, int ldd, int col0, int dtype, void* stream) {
    CAPMI_CHECK(ddst && dsrc, "capmi_bcast_rows_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_bcast_rows_bwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0 && ldd % Vec<T>::N == 0 && col0 % Vec<T>::N == 0, "capmi_bcast_rows_bwd: misaligned");
        int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(bcast_rows_bwd_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)ddst, (T*)dsrc, T_, B, cpr, ldd, col0);
    });
    CAPMI_LAUNCH_CHECK("capmi_bcast_rows_bwd");
    return 0;
}

// ------------------------------------------------------------------ LSTM cell (gate blocks i, f, o, g)
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const T* __restrict__ gates, const T* __restrict__ c_prev, T* h, T* c, int B, int cpr) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    const int H = cpr * VEC;
    const T* gr = gates + (int64_t)b * 4 * H + cc * VEC;
    Vec<T> gi = vload<T>(gr), gf = vload<T>(gr + H), go = vload<T>(gr + 2 * H), gg = vload<T>(gr + 3 * H), cp, hv, cv;
    if (c_prev) cp = vload<T>(c_prev + (int64_t)e * VEC);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float i_ = sigmoidf_(gi.get(v)), f_ = sigmoidf_(gf.get(v)), o_ = sigmoidf_(go.get(v)), g_ = tanhf_(gg.get(v));
        float cn = f_ * (c_prev ? cp.get(v) : 0.f) + i_ * g_;
        cv.set(v, cn);
        hv.set(v, o_ * tanhf_(cn));
    }
    vstore<T>(h + (int64_t)e * VEC, hv);
    vstore<T>(c + (int64_t)e * VEC, cv);
}
extern "C" int capmi_lstm_cell_fwd(const void* gates, const void* c_prev, void* h, void* c, int B, int H, int dtype, void* stream) {
    CAPMI_CHECK(gates && h && c, "capmi_lstm_cell_fwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_lstm_cell_fwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0, "capmi_lstm_cell_fwd: H not a multiple of the vector width");
        int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(lstm_cell_fwd_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)gates, (const T*)c_prev, (T*)h, (T*)c, B, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_lstm_cell_fwd");
    return 0;
}
// ------------------------------------------------------------------ decode step: state plumbing (capmi.h)
template <typename T>
__global__ __launch_bounds__(256) void decode_prep_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table, const T* __restrict__ h_src,
                                                          const int* __restrict__ rows, T* xh, int R, int ecpr, int hcpr, int V, int ldx, int h_col,
                                                          int padding_idx) {
    constexpr int VEC = Vec<T>::N;
    const int per = ecpr + hcpr;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * per) return;
    const int r = e / per, cc = e % per;
    if (cc < ecpr) {
        const int64_t id = ids[r];
        Vec<T> v = (id == padding_idx || id < 0 || id >= V) ? vzero<T>() : vload<T>(table + (id * ecpr + cc) * VEC);
        vstore<T>(xh + (int64_t)r * ldx + cc * VEC, v);
    } else {
        const int hc = cc - ecpr;
        const int src = rows ? rows[r] : r;
        vstore<T>(xh + (int64_t)r * ldx + h_col + hc * VEC, vload<T>(h_src + ((int64_t)src * hcpr + hc) * VEC));
    }
}
extern "C" int capmi_decode_prep(const int64_t* ids, const void* table, const void* h_src, const int* rows, void* xh, int R, int E, int H,
                                 int V, int ldx, int h_col, int padding_idx, int dtype, void* stream) {
    CAPMI_CHECK(ids && table && h_src && xh, "capmi_decode_prep: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_decode_prep", {
        constexpr int VEC = Vec<T>::N;
        CAPMI_CHECK(E % VEC == 0 && H % VEC == 0 && ldx % VEC == 0 && h_col % VEC == 0 && h_col >= E && h_col + H <= ldx,
                    "capmi_decode_prep: E=%d H=%d ldx=%d h_col=%d must be multiples of %d with E <= h_col <= ldx - H", E, H, ldx, h_col, VEC);
        const int ecpr = E / VEC, hcpr = H / VEC;
        hipLaunchKernelGGL(decode_prep_kernel<T>, dim3(cdiv((int64_t)R * (ecpr + hcpr), 256)), dim3(256), 0, (hipStream_t)stream, ids, (const T*)table,
                           (const T*)h_src, rows, (T*)xh, R, ecpr, hcpr, V, ldx, h_col, padding_idx);
    });
    CAPMI_LAUNCH_CHECK("capmi_decode_prep");
    return 0;
}
// the arithmetic of lstm_cell_fwd_kernel followed by sentinel_fwd_kernel's on the NEW cell state (both round h / c / s to T
// exactly where the two kernels do: s reads the stored c)
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_sentinel_fwd_kernel(const T* __restrict__ gs, int ld_gs, const T* __restrict__ c_src,
                                                                     const int* __restrict__ rows, T* h, T* c, T* s, int R, int cpr) {
    constexpr int VEC = Vec<T>::N;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * cpr) return;
    const int r = e / cpr, cc = e % cpr;
    const int H = cpr * VEC;
    const T* gr = gs + (int64_t)r * ld_gs + cc * VEC;
    Vec<T> gi = vload<T>(gr), gf = vload<T>(gr + H), go = vload<T>(gr + 2 * H), gg = vload<T>(gr + 3 * H), sg = vload<T>(gr + 4 * H);
    const int src = rows ? rows[r] : r;
    Vec<T> cp = vload<T>(c_src + ((int64_t)src * cpr + cc) * VEC), hv, cv, sv;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float i_ = sigmoidf_(gi.get(v)), f_ = sigmoidf_(gf.get(v)), o_ = sigmoidf_(go.get(v)), g_ = tanhf_(gg.get(v));
        float cn = f_ * cp.get(v) + i_ * g_;
        cv.set(v, cn);
        hv.set(v, o_ * tanhf_(cn));
        sv.set(v, sigmoidf_(sg.get(v)) * tanhf_(cv.get(v)));
    }
    vstore<T>(h + (int64_t)e * VEC, hv);
    vstore<T>(c + (int64_t)e * VEC, cv);
    vstore<T>(s + (int64_t)e * VEC, sv);
}
extern "C" int capmi_lstm_cell_sentinel_fwd(const void* gs, int ld_gs, const void* c_src, const int* rows, void* h, void* c, void* s,
                                            int R, int H, int dtype, void* stream) {
    CAPMI_CHECK(gs && c_src && h && c && s, "capmi_lstm_cell_sentinel_fwd: null pointer");
    CAPMI_CHECK(c_src != c || !rows, "capmi_lstm_cell_sentinel_fwd: a gathered cell state cannot be updated in place");
    CAPMI_DISPATCH(dtype, "capmi_lstm_cell_sentinel_fwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0 && ld_gs % Vec<T>::N == 0 && ld_gs >= 5 * H, "capmi_lstm_cell_sentinel_fwd: H / ld_gs not multiples of the vector width (ld_gs >= 5H)");
        const int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(lstm_cell_sentinel_fwd_kernel<T>, dim3(cdiv((int64_t)R * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)gs, ld_gs,
                           (const T*)c_src, rows, (T*)h, (T*)c, (T*)s, R, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_lstm_cell_sentinel_fwd");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const T* __restrict__ gates, const T* __restrict__ c_prev, const T* __restrict__ c,
                                                            const T* __restrict__ dh, const T* __restrict__ dc_in, T* dgates, T* dc_prev,
                                                            int dc_prev_acc, int B, int cpr) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    const int H = cpr * VEC;
    const int64_t go_ = (int64_t)b * 4 * H + cc * VEC;
    Vec<T> gi = vload<T>(gates + go_), gf = vload<T>(gates + go_ + H), go = vload<T>(gates + go_ + 2 * H), gg = vload<T>(gates + go_ + 3 * H);
    Vec<T> cv = vload<T>(c + (int64_t)e * VEC), dhv = vload<T>(dh + (int64_t)e * VEC), cp, dci, dcp_old;
    if (c_prev) cp = vload<T>(c_prev + (int64_t)e * VEC);
    if (dc_in) dci = vload<T>(dc_in + (int64_t)e * VEC);
    if (dc_prev && dc_prev_acc) dcp_old = vload<T>(dc_prev + (int64_t)e * VEC);
    Vec<T> di, df, do_, dg, dcp;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float i_ = sigmoidf_(gi.get(v)), f_ = sigmoidf_(gf.get(v)), o_ = sigmoidf_(go.get(v)), g_ = tanhf_(gg.get(v));
        float tc = tanhf_(cv.get(v));
        float dhh = dhv.get(v);
        float dct = (dc_in ? dci.get(v) : 0.f) + dhh * o_ * (1.f - tc * tc);
        float cpv = c_prev ? cp.get(v) : 0.f;
        di.set(v, dct * g_ * i_ * (1.f - i_));
        df.set(v, dct * cpv * f_ * (1.f - f_));
        do_.set(v, dhh * tc * o_ * (1.f - o_));
        dg.set(v, dct * i_ * (1.f - g_ * g_));
        float d = dct * f_;
        if (dc_prev && dc_prev_acc) d += dcp_old.get(v);
        dcp.set(v, d);
    }
    vstore<T>(dgates + go_, di);
    vstore<T>(dgates + go_ + H, df);
    vstore<T>(dgates + go_ + 2 * H, do_);
    vstore<T>(dgates + go_ + 3 * H, dg);
    if (dc_prev) vstore<T>(dc_prev + (int64_t)e * VEC, dcp);
}
extern "C" int capmi_lstm_cell_bwd(const void* gates, const void* c_prev, const void* c, const void* dh, const void* dc_in,
                                   void* dgates, void* dc_prev, int dc_prev_accumulate, int B, int H, int dtype, void* stream) {
    CAPMI_CHECK(gates && c && dh && dgates, "capmi_lstm_cell_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_lstm_cell_bwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0, "capmi_lstm_cell_bwd: H not a multiple of the vector width");
        int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(lstm_cell_bwd_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)gates, (const T*)c_prev,
                           (const T*)c, (const T*)dh, (const T*)dc_in, (T*)dgates, (T*)dc_prev, dc_prev_accumulate, B, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_lstm_cell_bwd");
    return 0;
}

// ------------------------------------------------------------------ visual sentinel
template <typename T>
__global__ __launch_bounds__(256) void sentinel_fwd_kernel(const T* __restrict__ sgpre, const T* __restrict__ c, T* s, int64_t nchunks) {
    constexpr int VEC = Vec<T>::N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        Vec<T> a = vload<T>(sgpre + e * VEC), cv = vload<T>(c + e * VEC), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, sigmoidf_(a.get(v)) * tanhf_(cv.get(v)));
        vstore<T>(s + e * VEC, ov);
    }
}
extern "C" int capmi_sentinel_fwd(const void* sgpre, const void* c, void* s, int64_t n, int dtype, void* stream) {
    CAPMI_CHECK(sgpre && c && s, "capmi_sentinel_fwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_sentinel_fwd", {
        CAPMI_CHECK(n % Vec<T>::N == 0, "capmi_sentinel_fwd: n not a multiple of the vector width");
        hipLaunchKernelGGL(sentinel_fwd_kernel<T>, dim3(ew_grid(n / Vec<T>::N)), dim3(256), 0, (hipStream_t)stream, (const T*)sgpre, (const T*)c, (T*)s, n / Vec<T>::N);
    });
    CAPMI_LAUNCH_CHECK("capmi_sentinel_fwd");
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void sentinel_bwd_kernel(const T* __restrict__ ds, const T* __restrict__ sgpre, const T* __restrict__ c,
                                                           T* dsgpre, T* dc, int64_t nchunks) {
    constexpr int VEC = Vec<T>::N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        Vec<T> d = vload<T>(ds + e * VEC), a = vload<T>(sgpre + e * VEC), cv = vload<T>(c + e * VEC), o1, o2;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float sg = sigmoidf_(a.get(v)), tc = tanhf_(cv.get(v)), dd = d.get(v);
            o1.set(v, dd * tc * sg * (1.f - sg));
            o2.set(v, dd * sg * (1.f - tc * tc));
        }
        vstore<T>(dsgpre + e * VEC, o1);
        vstore<T>(dc + e * VEC, o2);
    }
}
extern "C" int capmi_sentinel_bwd(const void* ds, const void* sgpre, const void* c, void* dsgpre, void* dc, int64_t n, int dtype, void* stream) {
    CAPMI_CHECK(ds && sgpre && c && dsgpre && dc, "capmi_sentinel_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_sentinel_bwd", {
        CAPMI_CHECK(n % Vec<T>::N == 0, "capmi_sentinel_bwd: n not a multiple of the vector width");
        hipLaunchKernelGGL(sentinel_bwd_kernel<T>, dim3(ew_grid(n / Vec<T>::N)), dim3(256), 0, (hipStream_t)stream, (const T*)ds, (const T*)sgpre, (const T*)c, (T*)dsgpre, (T*)dc, n / Vec<T>::N);
    });
    CAPMI_LAUNCH_CHECK("capmi_sentinel_bwd");
    return 0;
}

// ------------------------------------------------------------------ adaptive attention
// singleton mode (quirk Q1): alpha == 1, ctx = (sum_k Vt[b,k] + s)/(K+1); thread per (b, h-chunk).
template <typename T>
__global__ __launch_bounds__(256) void attn_singleton_fwd_kernel(const T* __restrict__ Vt, const T* __restrict__ s, const T* __restrict__ p,
                                                                 T* out, int T_, int B, int K, int cpr) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    float vs[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) vs[v] = 0.f;
    for (int k = 0; k < K; ++k) {
        Vec<T> x = vload<T>(Vt + (((int64_t)b * K + k) * cpr + cc) * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) vs[v] += x.get(v);
    }
    const float inv = 1.f / (float)(K + 1);
    for (int t = 0; t < T_; ++t) {
        int64_t o = (((int64_t)t * B + b) * cpr + cc) * VEC;
        Vec<T> sv = vload<T>(s + o), pv = vload<T>(p + o), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, (vs[v] + sv.get(v)) * inv + pv.get(v));
        vstore<T>(out + o, ov);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void attn_singleton_bwd_kernel(const T* __restrict__ dout, T* ds, T* dVt, int T_, int B, int K, int cpr) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    const float inv = 1.f / (float)(K + 1);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int t = 0; t < T_; ++t) {
        int64_t o = (((int64_t)t * B + b) * cpr + cc) * VEC;
        Vec<T> d = vload<T>(dout + o), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { float g = d.get(v) * inv; acc[v] += g; ov.set(v, g); }
        vstore<T>(ds + o, ov);
    }
    Vec<T> ov;
#pragma unroll
    for (int v = 0; v < VEC; ++v) ov.set(v, acc[v]);
    for (int k = 0; k < K; ++k) vstore<T>(dVt + (((int64_t)b * K + k) * cpr + cc) * VEC, ov);
}

// slots mode forward: one workgroup per row m = (t, b).  Phase 1: one wave per slot, e_k by a wave
// reduction; softmax over the K+1 slots; phase 3: thread (chunk, slot group) accumulates its slots'
// alpha_k * ctx_k in registers, the four slot groups meet in LDS (no atomics).
template <typename T>
__global__ __launch_bounds__(256) void attn_slots_fwd_kernel(const T* __restrict__ Ve, const T* __restrict__ Vt, const T* __restrict__ q,
                                                             const T* __restrict__ se, const T* __restrict__ s, const T* __restrict__ p,
                                                             const T* __restrict__ w10, const float* __restrict__ b10, T* out, float* alpha,
                                                             int B, int K, int H) {
    constexpr int VEC = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][H] ctx partials | [K+1] e | [16] scratch
    float* ctx = smem;
    float* ev = smem + 4 * H;
    float* scratch = ev + (K + 1);
    const int m = blockIdx.x, b = m % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpr = H / VEC;
    // e_k = sum_h tanh(feat_emb[k,h] + q[h]) * w10[h] + b10  -- one wave per slot, two slots in flight
    for (int k0 = wave; k0 <= K; k0 += 8) {
        const int k1 = k0 + 4;
        const T* row0 = k0 < K ? Ve + ((int64_t)b * K + k0) * H : se + (int64_t)m * H;
        const T* row1 = k1 < K ? Ve + ((int64_t)b * K + k1) * H : se + (int64_t)m * H;
        float part0 = 0.f, part1 = 0.f;
        for (int cc = lane; cc < cpr; cc += 64) {
            Vec<T> qq = vload<T>(q + (int64_t)m * H + cc * VEC), ww = vload<T>(w10 + cc * VEC);
            Vec<T> a0 = vload<T>(row0 + cc * VEC), a1 = k1 <= K ? vload<T>(row1 + cc * VEC) : vzero<T>();
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                part0 += tanh_for<T>(a0.get(v) + qq.get(v)) * ww.get(v);
                part1 += tanh_for<T>(a1.get(v) + qq.get(v)) * ww.get(v);
            }
        }
        part0 = wave_sum(part0);
        part1 = wave_sum(part1);
        if (lane == 0) {
            ev[k0] = part0 + b10[0];
            if (k1 <= K) ev[k1] = part1 + b10[0];
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = tid; k <= K; k += 256) mx = fmaxf(mx, ev[k]);
    mx = block_max(mx, scratch);
    float sum = 0.f;
    for (int k = tid; k <= K; k += 256) sum += expf(ev[k] - mx);
    sum = block_sum(sum, scratch);
    __syncthreads();
    for (int k = tid; k <= K; k += 256) {
        float a = expf(ev[k] - mx) / sum;
        ev[k] = a;
        alpha[(int64_t)m * (K + 1) + k] = a;
    }
    __syncthreads();
    // ctx[h] = sum_k alpha_k * ctx_all[k,h]: slot group `wave` takes k = wave, wave+4, ...
    for (int cc = lane; cc < cpr; cc += 64) {
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll 4
        for (int k = wave; k <= K; k += 4) {
            const T* row = k < K ? Vt + ((int64_t)b * K + k) * H : s + (int64_t)m * H;
            Vec<T> x = vload<T>(row + cc * VEC);
            const float a = ev[k];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += a * x.get(v);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) ctx[wave * H + cc * VEC + v] = acc[v];
    }
    __syncthreads();
    const float inv = 1.f / (float)(K + 1);
    for (int cc = tid; cc < cpr; cc += 256) {
        Vec<T> pv = vload<T>(p + (int64_t)m * H + cc * VEC), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int h = cc * VEC + v;
            ov.set(v, (ctx[h] + ctx[H + h] + ctx[2 * H + h] + ctx[3 * H + h]) * inv + pv.get(v));
        }
        vstore<T>(out + (int64_t)m * H + cc * VEC, ov);
    }
}

extern "C" int capmi_ada_attention_fwd(const void* Ve, const void* Vt, const void* q, const void* se, const void* s, const void* p,
                                       const void* w10, const float* b10, void* out, float* alpha, int T_, int B, int K, int H,
                                       int slots, int dtype, void* stream) {
    CAPMI_CHECK(Vt && s && p && out, "capmi_ada_attention_fwd: null pointer");
    CAPMI_CHECK(!slots || (Ve && q && se && w10 && b10 && alpha), "capmi_ada_attention_fwd: slots mode needs Ve,q,se,w10,b10,alpha");
    CAPMI_DISPATCH(dtype, "capmi_ada_attention_fwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0, "capmi_ada_attention_fwd: H not a multiple of the vector width");
        int cpr = H / Vec<T>::N;
        if (!slots) {
            hipLaunchKernelGGL(attn_singleton_fwd_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)Vt, (const T*)s, (const T*)p, (T*)out, T_, B, K, cpr);
        } else {
            size_t sh = (size_t)(4 * H + K + 1 + 16) * sizeof(float);
            hipLaunchKernelGGL(attn_slots_fwd_kernel<T>, dim3(T_ * B), dim3(256), sh, (hipStream_t)stream, (const T*)Ve, (const T*)Vt, (const T*)q, (const T*)se,
                               (const T*)s, (const T*)p, (const T*)w10, b10, (T*)out, alpha, B, K, H);
        }
    });
    CAPMI_LAUNCH_CHECK("capmi_ada_attention_fwd");
    return 0;
}

// slots backward, phase 1 (workgroup per row m): dalpha, de, ds, db10.
template <typename T>
__global__ __launch_bounds__(256) void attn_slots_bwd1_kernel(const T* __restrict__ Vt, const T* __restrict__ s, const float* __restrict__ alpha,
                                                              const T* __restrict__ dout, T* ds, float* de, float* db10, float* det_db, int B, int K, int H) {
    constexpr int VEC = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [K+1] dalpha | [16] scratch
    float* da = smem;
    float* scratch = smem + (K + 1);
    const int m = blockIdx.x, b = m % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpr = H / VEC;
    const float inv = 1.f / (float)(K + 1);
    for (int k = wave; k <= K; k += 4) {
        const T* row = k < K ? Vt + ((int64_t)b * K + k) * H : s + (int64_t)m * H;
        float part = 0.f;
        for (int cc = lane; cc < cpr; cc += 64) {
            Vec<T> x = vload<T>(row + cc * VEC), d = vload<T>(dout + (int64_t)m * H + cc * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) part += x.get(v) * d.get(v);
        }
        part = wave_sum(part);
        if (lane == 0) da[k] = part * inv;
    }
    __syncthreads();
    float dot = 0.f;
    for (int k = tid; k <= K; k += 256) dot += alpha[(int64_t)m * (K + 1) + k] * da[k];
    dot = block_sum(dot, scratch);
    float dsum = 0.f;
    for (int k = tid; k <= K; k += 256) {
        float a = alpha[(int64_t)m * (K + 1) + k];
        float d = a * (da[k] - dot);
        de[(int64_t)m * (K + 1) + k] = d;
        dsum += d;
    }
    dsum = block_sum(dsum, scratch);
    if (tid == 0) {
        if (det_db) det_db[m] = dsum;           // deterministic mode: per-row partial, summed in row order by attn_det_finish_kernel
        else atomicAdd(db10, dsum);
    }
    const float aK = alpha[(int64_t)m * (K + 1) + K] * inv;
    for (int cc = tid; cc < cpr; cc += 256) {
        Vec<T> d = vload<T>(dout + (int64_t)m * H + cc * VEC), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, d.get(v) * aK);
        vstore<T>(ds + (int64_t)m * H + cc * VEC, ov);
    }
}

// slots backward, phase 2: workgroup per (b, tile of TH = 8 16-byte chunks of h).  Threads = 8 chunk columns x
// 32 slot groups; dVt/dVe accumulate over t in registers; dq (a sum over the slots) meets by wave shuffles
// + one LDS hand-off per step (double-buffered: one barrier per step); the next step's operands are
// loaded before the current step is computed.
template <typename T, int MAXK>
__global__ __launch_bounds__(256) void attn_slots_bwd2_kernel(const T* __restrict__ Ve, const T* __restrict__ q, const T* __restrict__ se,
                                                              const T* __restrict__ w10, const float* __restrict__ alpha, const float* __restrict__ de,
                                                              const T* __restrict__ dout, T* dVt, T* dVe, T* dq, T* dse, float* dw10, float* det_dw,
                                                              int T_, int B, int K, int H) {
    constexpr int VEC = Vec<T>::N;
    constexpr int TH = 8, KG = 256 / TH;
    __shared__ float sdq[2][4][TH * VEC], sdw[4][TH * VEC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cc = tid % TH, kg = tid / TH;
    const int tiles = (H / VEC + TH - 1) / TH;
    const int b = blockIdx.x / tiles, chunk = (blockIdx.x % tiles) * TH + cc;
    const bool cok = chunk * VEC < H;
    const float inv = 1.f / (float)(K + 1);
    float aVt[MAXK][VEC], aVe[MAXK][VEC], ve[MAXK][VEC], ww[VEC], dwl[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { ww[v] = 0.f; dwl[v] = 0.f; }
    if (cok) {
        Vec<T> w = vload<T>(w10 + chunk * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) ww[v] = w.get(v);
    }
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        int k = kg + j * KG;
        Vec<T> x = (cok && k < K) ? vload<T>(Ve + ((int64_t)b * K + k) * H + chunk * VEC) : vzero<T>();
#pragma unroll
        for (int v = 0; v < VEC; ++v) { ve[j][v] = x.get(v); aVt[j][v] = 0.f; aVe[j][v] = 0.f; }
    }
    // operands of one step: this thread's slots' alpha / de, the row's dout / q chunk, the sentinel's de and se chunk
    struct Step { float a[MAXK], d[MAXK], dK; Vec<T> dv, qv, sv; };
    auto load_step = [&](int t) {
        Step s;
        const int64_t m = (int64_t)t * B + b;
#pragma unroll
        for (int j = 0; j < MAXK; ++j) {
            const int k = kg + j * KG;
            s.a[j] = k < K ? alpha[m * (K + 1) + k] * inv : 0.f;
            s.d[j] = k < K ? de[m * (K + 1) + k] : 0.f;
        }
        s.dK = de[m * (K + 1) + K];
        s.dv = cok ? vload<T>(dout + m * H + chunk * VEC) : vzero<T>();
        s.qv = cok ? vload<T>(q + m * H + chunk * VEC) : vzero<T>();
        s.sv = (cok && kg == 0) ? vload<T>(se + m * H + chunk * VEC) : vzero<T>();
        return s;
    };
    Step cur = load_step(0);
    for (int t = 0; t < T_; ++t) {
        const int64_t m = (int64_t)t * B + b;
        Step nxt = cur;
        if (t + 1 < T_) nxt = load_step(t + 1);
        float dql[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) dql[v] = 0.f;
#pragma unroll
        for (int j = 0; j < MAXK; ++j) {
            const float a = cur.a[j], dd = cur.d[j];       // zero beyond K: contributes nothing
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float z = tanh_for<T>(ve[j][v] + cur.qv.get(v));
                float dz = dd * ww[v] * (1.f - z * z);
                aVe[j][v] += dz;
                dql[v] += dz;
                dwl[v] += dd * z;
                aVt[j][v] += cur.dv.get(v) * a;
            }
        }
        if (kg == 0 && cok) {      // sentinel slot K
            Vec<T> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float z = tanh_for<T>(cur.sv.get(v) + cur.qv.get(v));
                float dz = cur.dK * ww[v] * (1.f - z * z);
                o.set(v, dz);
                dql[v] += dz;
                dwl[v] += cur.dK * z;
            }
            vstore<T>(dse + m * H + chunk * VEC, o);
        }
        // dq: sum over the 32 slot groups = lanes with equal (lane & 7), then over the 4 waves
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float x = dql[v];
            x += __shfl_xor(x, 8);
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            dql[v] = x;
        }
        if (lane < TH) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) sdq[t & 1][wave][lane * VEC + v] = dql[v];
        }
        __syncthreads();
        if (kg == 0 && cok) {
            Vec<T> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                o.set(v, sdq[t & 1][0][cc * VEC + v] + sdq[t & 1][1][cc * VEC + v] + sdq[t & 1][2][cc * VEC + v] + sdq[t & 1][3][cc * VEC + v]);
            vstore<T>(dq + m * H + chunk * VEC, o);
        }
        cur = nxt;
    }
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        int k = kg + j * KG;
        if (cok && k < K) {
            Vec<T> o1, o2;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { o1.set(v, aVt[j][v]); o2.set(v, aVe[j][v]); }
            vstore<T>(dVt + ((int64_t)b * K + k) * H + chunk * VEC, o1);
            vstore<T>(dVe + ((int64_t)b * K + k) * H + chunk * VEC, o2);
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float x = dwl[v];
        x += __shfl_xor(x, 8);
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        dwl[v] = x;
    }
    if (lane < TH) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) sdw[wave][lane * VEC + v] = dwl[v];
    }
    __syncthreads();
    if (kg == 0 && cok) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float part = sdw[0][cc * VEC + v] + sdw[1][cc * VEC + v] + sdw[2][cc * VEC + v] + sdw[3][cc * VEC + v];
            if (det_dw) det_dw[(int64_t)b * H + chunk * VEC + v] = part;      // deterministic mode: per-image partial
            else atomicAdd(&dw10[chunk * VEC + v], part);
        }
    }
}

// Deterministic mode (capmi_deterministic): d b10 += sum_m det_db[m] (row order), d w10[h] += sum_b det_dw[b][h] (image order).
constexpr int ATTN_DET_FLOATS = 1 << 19;
__device__ float attn_det_scratch[ATTN_DET_FLOATS];
__global__ __launch_bounds__(256) void attn_det_finish_kernel(const float* __restrict__ det_db, int M, const float* __restrict__ det_dw, int B, int H,
                                                              float* db10, float* dw10) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h < H) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += det_dw[(int64_t)b * H + h];
        dw10[h] += acc;
    }
    if (h == 0) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += det_db[m];
        *db10 += acc;
    }
}

extern "C" int capmi_ada_attention_bwd(const void* Ve, const void* Vt, const void* q, const void* se, const void* s, const void* w10,
                                       const float* alpha, const void* dout, void* ds, void* dVt, void* dVe, void* dq, void* dse,
                                       float* dw10, float* db10, float* de, int T_, int B, int K, int H, int slots, int dtype,
                                       void* stream) {
    CAPMI_CHECK(dout && ds && dVt, "capmi_ada_attention_bwd: null pointer");
    CAPMI_CHECK(!slots || (Ve && Vt && q && se && s && w10 && alpha && dVe && dq && dse && dw10 && db10 && de),
                "capmi_ada_attention_bwd: slots mode needs every operand");
    CAPMI_DISPATCH(dtype, "capmi_ada_attention_bwd", {
        CAPMI_CHECK(H % Vec<T>::N == 0, "capmi_ada_attention_bwd: H not a multiple of the vector width");
        int cpr = H / Vec<T>::N;
        if (!slots) {
            hipLaunchKernelGGL(attn_singleton_bwd_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)ds, (T*)dVt, T_, B, K, cpr);
        } else {
            constexpr int TH = 8;
            CAPMI_CHECK(K <= 8 * (256 / TH), "capmi_ada_attention_bwd: K=%d above the supported %d", K, 8 * (256 / TH));
            size_t sh = (size_t)(K + 1 + 16) * sizeof(float);
            float* det_db = nullptr;
            float* det_dw = nullptr;
            if (capmi_deterministic()) {        // library-owned scratch of the current device: [T*B] d b10 partials | [B][H] d w10 partials
                CAPMI_CHECK((long long)T_ * B + (long long)B * H <= ATTN_DET_FLOATS, "capmi_ada_attention_bwd: T*B + B*H above the deterministic scratch (%d floats)", ATTN_DET_FLOATS);
                CAPMI_CHECK(hipGetSymbolAddress((void**)&det_db, HIP_SYMBOL(attn_det_scratch)) == hipSuccess && det_db, "capmi_ada_attention_bwd: no deterministic scratch");
                det_dw = det_db + (size_t)T_ * B;
            }
            hipLaunchKernelGGL(attn_slots_bwd1_kernel<T>, dim3(T_ * B), dim3(256), sh, (hipStream_t)stream, (const T*)Vt, (const T*)s, alpha, (const T*)dout, (T*)ds, de, db10, det_db, B, K, H);
            int tiles = cdiv(cpr, TH);
            if (K <= 2 * (256 / TH))
                hipLaunchKernelGGL((attn_slots_bwd2_kernel<T, 2>), dim3(B * tiles), dim3(256), 0, (hipStream_t)stream, (const T*)Ve, (const T*)q, (const T*)se,
                                   (const T*)w10, alpha, de, (const T*)dout, (T*)dVt, (T*)dVe, (T*)dq, (T*)dse, dw10, det_dw, T_, B, K, H);
            else
                hipLaunchKernelGGL((attn_slots_bwd2_kernel<T, 8>), dim3(B * tiles), dim3(256), 0, (hipStream_t)stream, (const T*)Ve, (const T*)q, (const T*)se,
                                   (const T*)w10, alpha, de, (const T*)dout, (T*)dVt, (T*)dVe, (T*)dq, (T*)dse, dw10, det_dw, T_, B, K, H);
            if (det_db)
                hipLaunchKernelGGL(attn_det_finish_kernel, dim3(cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, det_db, T_ * B, det_dw, B, H, db10, dw10);
        }
    });
    CAPMI_LAUNCH_CHECK("capmi_ada_attention_bwd");
    return 0;
}

// ------------------------------------------------------------------ masked softmax cross-entropy over f32 logits
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* row_loss,
                                                       float* row_lse, int V, int ld, int padding_idx) {
    __shared__ float scratch[16];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
    mx = block_max(mx, scratch);
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(row[i] - mx);
    sum = block_sum(sum, scratch);
    if (tid == 0) {
        float lse = logf(sum) + mx;
        int64_t t = target[m];
        row_lse[m] = lse;
        row_loss[m] = (t != padding_idx && t >= 0 && t < V) ? lse - row[t] : 0.f;
    }
}
// The same with the row held in registers (V <= 1024 * NV columns, ld a multiple of 4, 16-byte aligned rows): ONE pass over
// the logits as 16-byte loads, all in flight before the first comparison -- the two-pass scalar form above read every row
// twice, four bytes per lane (28 us for 1216 x 10 000; the 49 MB take ~12 us).  Same max-then-sum arithmetic.
template <int NV>
__global__ __launch_bounds__(256) void xent_fwd_reg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* row_loss,
                                                           float* row_lse, int V, int ld, int padding_idx) {
    __shared__ float scratch[16];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    f32x4 r[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + tid) * 4;
        r[j] = i < V ? *reinterpret_cast<const f32x4*>(row + i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e >= V) r[j][e] = -INFINITY;            // columns V .. ld-1 of the padded row
            mx = fmaxf(mx, r[j][e]);
        }
    }
    mx = block_max(mx, scratch);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += expf(r[j][e] - mx);      // exp(-inf) = 0 for the padding
    sum = block_sum(sum, scratch);
    if (tid == 0) {
        float lse = logf(sum) + mx;
        int64_t t = target[m];
        row_lse[m] = lse;
        row_loss[m] = (t != padding_idx && t >= 0 && t < V) ? lse - row[t] : 0.f;
    }
}
extern "C" int capmi_softmax_xent_fwd(const float* logits, const int64_t* target, float* row_loss, float* row_lse, int M, int V,
                                      int ld, int padding_idx, void* stream) {
    CAPMI_CHECK(logits && target && row_loss && row_lse, "capmi_softmax_xent_fwd: null pointer");
    if (M <= 0) return 0;
    const bool vec = ld % 4 == 0 && ((uintptr_t)logits & 15) == 0;
    if (vec && V <= 1024 * 10) hipLaunchKernelGGL(xent_fwd_reg_kernel<10>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, row_loss, row_lse, V, ld, padding_idx);
    else if (vec && V <= 1024 * 20) hipLaunchKernelGGL(xent_fwd_reg_kernel<20>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, row_loss, row_lse, V, ld, padding_idx);
    else hipLaunchKernelGGL(xent_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, row_loss, row_lse, V, ld, padding_idx);
    CAPMI_LAUNCH_CHECK("capmi_softmax_xent_fwd");
    return 0;
}
// loss = sum(row_loss) / count(target != pad); single workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(1024) void xent_finalize_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ target, float* loss_out,
                                                             float* count_out, int M, int padding_idx) {
    __shared__ float scratch[16];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < M; i += 1024) { s += row_loss[i]; c += target[i] != padding_idx ? 1.f : 0.f; }
    s = block_sum(s, scratch);
    c = block_sum(c, scratch);
    if (threadIdx.x == 0) { loss_out[0] = s / c; count_out[0] = c; }
}
extern "C" int capmi_xent_finalize(const float* row_loss, const int64_t* target, float* loss_out, float* count_out, int M,
                                   int padding_idx, void* stream) {
    CAPMI_CHECK(row_loss && target && loss_out && count_out, "capmi_xent_finalize: null pointer");
    hipLaunchKernelGGL(xent_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, target, loss_out, count_out, M, padding_idx);
    CAPMI_LAUNCH_CHECK("capmi_xent_finalize");
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ row_lse,
                                                       const float* __restrict__ count, T* dlogits, int V, int ld, int ldd, int padding_idx) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    T* drow = dlogits + (int64_t)m * ldd;
    const int64_t t = target[m];
    const bool live = t != padding_idx && t >= 0 && t < V;
    const float scale = live ? 1.f / count[0] : 0.f;
    const float lse = row_lse[m];
    for (int i = tid; i < ldd; i += 256) {
        float g = 0.f;
        if (i < V && live) g = (expf(row[i] - lse) - (i == t ? 1.f : 0.f)) * scale;
        drow[i] = from_f32<T>(g);
    }
}
// 4 columns per lane (16-byte loads; 8-byte bf16 / 16-byte f32 stores) when ld, ldd are multiples of 4 and the rows 16-byte aligned
template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_vec_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ row_lse,
                                                           const float* __restrict__ count, T* dlogits, int V, int ld, int ldd, int padding_idx) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    T* drow = dlogits + (int64_t)m * ldd;
    const int64_t t = target[m];
    const bool live = t != padding_idx && t >= 0 && t < V;
    const float scale = live ? 1.f / count[0] : 0.f;
    const float lse = row_lse[m];
    for (int i = tid * 4; i < ldd; i += 1024) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (live && i < V) {
            const f32x4 x = i + 4 <= ld ? *reinterpret_cast<const f32x4*>(row + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = i + e < V ? (expf(x[e] - lse) - (i + e == t ? 1.f : 0.f)) * scale : 0.f;
        }
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>(drow + i) = g;
        } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            bf16x4_t o = {(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
            *reinterpret_cast<bf16x4_t*>(drow + i) = o;
        }
    }
}
extern "C" int capmi_softmax_xent_bwd(const float* logits, const int64_t* target, const float* row_lse, const float* count,
                                      void* dlogits, int M, int V, int ld, int ldd, int padding_idx, int dtype, void* stream) {
    CAPMI_CHECK(logits && target && row_lse && count && dlogits, "capmi_softmax_xent_bwd: null pointer");
    if (M <= 0) return 0;
    CAPMI_DISPATCH(dtype, "capmi_softmax_xent_bwd", {
        if (ld % 4 == 0 && ldd % 4 == 0 && ldd <= ld && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0)
            hipLaunchKernelGGL(xent_bwd_vec_kernel<T>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, row_lse, count, (T*)dlogits, V, ld, ldd, padding_idx);
        else
            hipLaunchKernelGGL(xent_bwd_kernel<T>, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, row_lse, count, (T*)dlogits, V, ld, ldd, padding_idx);
    });
    CAPMI_LAUNCH_CHECK("capmi_softmax_xent_bwd");
    return 0;
}

// ------------------------------------------------------------------ argmax (lowest index on ties)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int64_t* ids, float* ids_f32, int ld_f32, int V, int ld) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (ld % 4 == 0 && ((uintptr_t)logits & 15) == 0) {         // 16-byte loads, four columns per lane (ascending index inside a lane)
        for (int i = tid * 4; i < V; i += 1024) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < V && f[e] > best) { best = f[e]; bi = i + e; }
        }
    } else {
        for (int i = tid; i < V; i += 256) {
            float f = row[i];
            if (f > best || (f == best && i < bi)) { best = f; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;
        ids[m] = bi;
        if (ids_f32) ids_f32[(int64_t)m * ld_f32] = (float)bi;
    }
}
extern "C" int capmi_argmax(const float* logits, int64_t* ids_out, float* ids_f32, int ld_f32, int M, int V, int ld, void* stream) {
    CAPMI_CHECK(logits && ids_out, "capmi_argmax: null pointer");
    if (M <= 0) return 0;
    hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ids_out, ids_f32, ld_f32, V, ld);
    CAPMI_LAUNCH_CHECK("capmi_argmax");
    return 0;
}

// ------------------------------------------------------------------ beam search (build-defined extension of the decode path)
// Semantics = oracle/model.py beam_decode: rows are beam-major (row k*B + b), score = sum of log-softmax
// probabilities, ties to the lower beam index then the lower token id, no special casing of <stop>.
// Step 1 (one workgroup per row): log-sum-exp of the row and its `beam` largest logits (value desc, index asc).
__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ logits, int V, int ld, int beam, float* cand_val,
                                                        int* cand_idx, float* lse) {
    __shared__ float scratch[16];
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int chosen[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)m * ld;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
    mx = block_max(mx, scratch);
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(row[i] - mx);
    sum = block_sum(sum, scratch);
    if (tid == 0) lse[m] = mx + logf(sum);
    for (int r = 0; r < beam; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            bool taken = false;
            for (int q = 0; q < r; ++q) taken |= chosen[q] == i;
            const float f = row[i];
            if (!taken && (f > best || (f == best && i < bi))) { best = f; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
            if (bi == 0x7fffffff) { bi = 0; best = -INFINITY; }      // V < beam: fewer candidates than ranks
            chosen[r] = bi;
            cand_val[(int64_t)m * beam + r] = best;
            cand_idx[(int64_t)m * beam + r] = bi;
        }
        __syncthreads();
    }
}
// Step 1 with the row in registers (V <= 1024 * NV, ld a multiple of 4, 16-byte aligned rows): ONE pass over the logits
// (16-byte loads, all in flight at once), then max, sum and the `beam` selection rounds run on registers -- a chosen
// element is overwritten with -inf by the lane that holds it.  (The scalar form above walks the row 2 + beam times:
// 113 us per step for 640 rows x 10 000, a fifth of the whole beam-5 decode.)  Same order: value desc, index asc.
template <int NV>
__global__ __launch_bounds__(256) void beam_topk_reg_kernel(const float* __restrict__ logits, int V, int ld, int beam, float* cand_val,
                                                            int* cand_idx, float* lse) {
    // ONE barrier per selection round (the first form took nineteen for beam 5: two each for the max and the sum, three per
    // round): every lane keeps the best of ITS values; a round folds the lane bests over the wave with shuffles, the four wave
    // winners meet in LDS (two buffers, alternating: a round's writes cannot overtake the reads of the round before last, which
    // all happened before the barrier in between), EVERY thread picks the winner from the four -- no thread-0 section, no
    // broadcast -- and only the lane that held it rescans its values.  The row maximum is round 0's winner; the sum of
    // exponentials rides in round 1's exchange (its own exchange when beam == 1).  Same results bit for bit: same order (value
    // desc, index asc), same per-lane / per-wave / wave-order summation as block_sum.
    __shared__ float sv[2][4];
    __shared__ int si[2][4];
    __shared__ float ssum[16];
    const int m = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const float* row = logits + (int64_t)m * ld;
    f32x4 r[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + tid) * 4;
        r[j] = i < V ? *reinterpret_cast<const f32x4*>(row + i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((j * 256 + tid) * 4 + e >= V) r[j][e] = -INFINITY;      // columns V .. ld-1 of the padded row
    float best;
    int bi;
    auto lane_best = [&]() {
        best = -INFINITY;
        bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                               // ascending index inside the lane: strict > keeps the lower one
                const float f = r[j][e];
                if (f > best) { best = f; bi = (j * 256 + tid) * 4 + e; }
            }
    };
    lane_best();
    float wsum = 0.f;
    for (int q = 0; q < beam; ++q) {
        float wb = best;
        int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(wb, o, 64);
            const int oi = __shfl_xor(wi, o, 64);
            if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
        }
        if ((tid & 63) == 0) {
            sv[q & 1][wave] = wb;
            si[q & 1][wave] = wi;
            if (q == 1) ssum[wave] = wsum;
        }
        __syncthreads();
        float gb = sv[q & 1][0];
        int gi = si[q & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = sv[q & 1][w];
            const int oi = si[q & 1][w];
            if (ob > gb || (ob == gb && oi < gi)) { gb = ob; gi = oi; }
        }
        if (q == 0) {           // gb is the row maximum: this lane's share of the sum, over the values as loaded
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += expf(r[j][e] - gb);
            wsum = wave_sum(sum);
            if (tid == 0) ssum[8] = gb;
        }
        if (q == 1 && tid == 0) lse[m] = ssum[8] + logf(ssum[0] + ssum[1] + ssum[2] + ssum[3]);
        if (gi == 0x7fffffff) { gi = 0; gb = -INFINITY; }               // V < beam: fewer candidates than ranks
        if (tid == 0) {
            cand_val[(int64_t)m * beam + q] = gb;
            cand_idx[(int64_t)m * beam + q] = gi;
        }
        if (gb != -INFINITY && ((gi >> 2) & 255) == tid) {              // the lane that holds it takes it out of the running
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((j * 256 + tid) * 4 + e == gi) r[j][e] = -INFINITY;
            lane_best();
        }
    }
    if (beam < 2) {             // no second round carried the sum
        __syncthreads();
        if ((tid & 63) == 0) ssum[wave] = wsum;
        __syncthreads();
        if (tid == 0) lse[m] = ssum[8] + logf(ssum[0] + ssum[1] + ssum[2] + ssum[3]);
    }
}
// Step 2: the `beam` best of the beam x beam candidates by score[k] + logit - lse[k]; writes the new scores, (parent,
// token) of step t, the next input ids and the state-gather rows.  Eight lanes per image, lane k owns source hypothesis k:
// its candidates arrive sorted (value desc, token asc), so the selection is a `beam`-round merge of up to 8 sorted lists --
// every lane offers the head of its list, three xor-shuffles pick the winner (total desc, then the lower k: the order
// the one-thread-per-image form visited them in), the winner advances.  (That form took 21-28 us per step: one wave,
// 5 x 64 dependent compares per lane.)
__global__ __launch_bounds__(256) void beam_select_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                          const float* __restrict__ lse, const float* __restrict__ score_in, int B, int beam,
                                                          float* score_out, int* parents, int* tokens, int64_t* next_ids, int* gather_rows) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = t >> 3, k = t & 7;
    const bool live = b < B && k < beam;
    const int row = live ? k * B + b : 0;
    float tot[8];
    int tok[8];
    const float base = live ? score_in[row] - lse[row] : 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool ok = live && r < beam;
        const int64_t o = (int64_t)row * beam + (r < beam ? r : 0);
        tot[r] = ok ? base + cand_val[o] : -INFINITY;
        tok[r] = ok ? cand_idx[o] : 0x7fffffff;
    }
    int head = 0;                                       // next unused candidate of this lane's list
    for (int j = 0; j < beam; ++j) {
        float hv = -INFINITY;
        int ht = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r == head) { hv = tot[r]; ht = tok[r]; }
        int key = (live && head < beam) ? k : 8 + k;    // exhausted / idle lanes lose against any live one
        float bv = hv;
        int bk = key, bt = ht;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int ok2 = __shfl_xor(bk, o, 64), ot = __shfl_xor(bt, o, 64);
            const bool mine_dead = bk >= 8, other_dead = ok2 >= 8;
            const bool take = mine_dead != other_dead ? mine_dead : (ov > bv || (ov == bv && ok2 < bk));
            if (take) { bv = ov; bk = ok2; bt = ot; }
        }
        if (bk == k && live) ++head;                    // this lane's head was taken
        if (k == 0 && b < B) {
            const int o = j * B + b;
            score_out[o] = bv;
            parents[o] = bk & 7;
            tokens[o] = bt;
            next_ids[o] = bt;
            gather_rows[o] = (bk & 7) * B + b;
        }
    }
}
extern "C" int capmi_beam_step(const float* logits, int V, int ld, int B, int beam, const float* score_in, float* score_out,
                               float* cand_val, int* cand_idx, float* lse, int* parents, int* tokens, int64_t* next_ids,
                               int* gather_rows, void* stream) {
    CAPMI_CHECK(logits && score_in && score_out && cand_val && cand_idx && lse && parents && tokens && next_ids && gather_rows,
                "capmi_beam_step: null pointer");
    CAPMI_CHECK(beam >= 1 && beam <= 8 && B >= 1 && V >= 1, "capmi_beam_step: beam=%d (1..8), B=%d, V=%d", beam, B, V);
    const bool vec = ld % 4 == 0 && ((uintptr_t)logits & 15) == 0;
    if (vec && V <= 1024 * 10) hipLaunchKernelGGL(beam_topk_reg_kernel<10>, dim3(beam * B), dim3(256), 0, (hipStream_t)stream, logits, V, ld, beam, cand_val, cand_idx, lse);
    else if (vec && V <= 1024 * 20) hipLaunchKernelGGL(beam_topk_reg_kernel<20>, dim3(beam * B), dim3(256), 0, (hipStream_t)stream, logits, V, ld, beam, cand_val, cand_idx, lse);
    else hipLaunchKernelGGL(beam_topk_kernel, dim3(beam * B), dim3(256), 0, (hipStream_t)stream, logits, V, ld, beam, cand_val, cand_idx, lse);
    hipLaunchKernelGGL(beam_select_kernel, dim3(cdiv(B, 32)), dim3(256), 0, (hipStream_t)stream, cand_val, cand_idx, lse, score_in, B, beam,
                       score_out, parents, tokens, next_ids, gather_rows);
    CAPMI_LAUNCH_CHECK("capmi_beam_step");
    return 0;
}
// dst[r][:] = src[rows[r]][:]  (hidden / cell state of the surviving hypotheses)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ rows, T* dst, int n, int cpr) {
    constexpr int VEC = Vec<T>::N;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * cpr) return;
    const int r = e / cpr, cc = e % cpr;
    vstore<T>(dst + ((int64_t)r * cpr + cc) * VEC, vload<T>(src + ((int64_t)rows[r] * cpr + cc) * VEC));
}
extern "C" int capmi_gather_rows(const void* src, const int* rows, void* dst, int n, int H, int dtype, void* stream) {
    CAPMI_CHECK(src && rows && dst, "capmi_gather_rows: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_gather_rows", {
        CAPMI_CHECK(H % Vec<T>::N == 0, "capmi_gather_rows: H not a multiple of the vector width");
        const int cpr = H / Vec<T>::N;
        hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(cdiv((int64_t)n * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)src, rows, (T*)dst, n, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_gather_rows");
    return 0;
}
// ids[b][t] (float32, quirk Q2) of the best final hypothesis (rank 0), walking the parents back from the last step
__global__ __launch_bounds__(64) void beam_backtrack_kernel(const int* __restrict__ tokens, const int* __restrict__ parents, float* out, int Ti, int B, int beam) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    int j = 0;
    for (int t = Ti - 1; t >= 0; --t) {
        const int o = (t * beam + j) * B + b;
        out[(int64_t)b * Ti + t] = (float)tokens[o];
        j = parents[o];
    }
}
extern "C" int capmi_beam_backtrack(const int* tokens, const int* parents, float* out_ids_f32, int Ti, int B, int beam, void* stream) {
    CAPMI_CHECK(tokens && parents && out_ids_f32, "capmi_beam_backtrack: null pointer");
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, tokens, parents, out_ids_f32, Ti, B, beam);
    CAPMI_LAUNCH_CHECK("capmi_beam_backtrack");
    return 0;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// DET (capmi_deterministic): one row block per column block, row lanes folded in lane order, a single writer per column.
template <typename T, bool DET>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ a, int M, int N, int lda, float* out, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float s1[256 * VEC];
    const int tid = threadIdx.x;
    for (int i = tid; i < L.cpc * VEC; i += 256) s1[i] = 0.f;
    __syncthreads();
    const int cc = tid % L.cpc, rr = tid / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    const bool active = rr < L.rp && chunk * VEC < N;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (active) {
        const int m_begin = blockIdx.x * L.rows_per_block;
        const int m_end = min(M, m_begin + L.rows_per_block);
        for (int m = m_begin + rr; m < m_end; m += L.rp) {
            Vec<T> x = vload<T>(a + (int64_t)m * lda + chunk * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += x.get(v);
        }
        if constexpr (!DET) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) atomicAdd(&s1[cc * VEC + v], acc[v]);
        }
    }
    if constexpr (DET) {
        for (int r = 0; r < L.rp; ++r) {
            if (active && rr == r) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) s1[cc * VEC + v] += acc[v];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = tid; i < L.cpc * VEC; i += 256) {
        int c = blockIdx.y * L.cpc * VEC + i;
        if (c < N) {
            if constexpr (DET) out[c] += s1[i];
            else atomicAdd(&out[c], s1[i]);
        }
    }
}
extern "C" int capmi_colsum(const void* a, int M, int N, int lda, float* out, int dtype, void* stream) {
    CAPMI_CHECK(a && out, "capmi_colsum: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_colsum", {
        constexpr int VEC = Vec<T>::N;
        CAPMI_CHECK(lda % VEC == 0 && lda >= (N + VEC - 1) / VEC * VEC, "capmi_colsum: lda=%d must be a multiple of %d covering N=%d", lda, VEC, N);
        int gx, gy;
        ColLayout L = col_layout(M, (N + VEC - 1) / VEC * VEC, VEC, &gx, &gy);
        if (capmi_deterministic()) {
            L.rows_per_block = M;
            hipLaunchKernelGGL((colsum_kernel<T, true>), dim3(1, gy), dim3(256), 0, (hipStream_t)stream, (const T*)a, M, N, lda, out, L);
        } else {
            hipLaunchKernelGGL((colsum_kernel<T, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)a, M, N, lda, out, L);
        }
    });
    CAPMI_LAUNCH_CHECK("capmi_colsum");
    return 0;
}
