// HBM-bound encoder kernels for gfx950: depthwise 3x3, 3x3/s2 max pool, stem im2col and the
// elementwise glue (batch norm lives in bn_ops.hip).  All operate on NHWC tensors viewed as
// [M = B*H*W][C]; every global access is a 16-byte chunk (8 bf16 / 4 f32) of consecutive
// channels, so a 64-lane wave touches 1 KiB of contiguous memory per instruction.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------ elementwise glue
template <typename T>
__global__ __launch_bounds__(256) void add_act_kernel(const T* __restrict__ a, const T* __restrict__ b, T* y, int64_t nchunks, int act) {
    constexpr int VEC = Vec<T>::N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        Vec<T> av = vload<T>(a + e * VEC), bv = vload<T>(b + e * VEC), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, apply_act(av.get(v) + bv.get(v), act));
        vstore<T>(y + e * VEC, ov);
    }
}
extern "C" int capmi_add_act(const void* a, const void* b, void* y, int64_t n, int act, int dtype, void* stream) {
    CAPMI_CHECK(a && b && y, "capmi_add_act: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_add_act", {
        CAPMI_CHECK(n % Vec<T>::N == 0, "capmi_add_act: n not a multiple of the vector width");
        int64_t nc = n / Vec<T>::N;
        hipLaunchKernelGGL(add_act_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)y, nc, act);
    });
    CAPMI_LAUNCH_CHECK("capmi_add_act");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* dx, int acc, int64_t nchunks, int act) {
    constexpr int VEC = Vec<T>::N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        Vec<T> dv = vload<T>(dy + e * VEC), yv, ov, old;
        if (act) yv = vload<T>(y + e * VEC);
        if (acc) old = vload<T>(dx + e * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float g = dv.get(v);
            if (act) g *= act_grad_from_out(yv.get(v), act);
            if (acc) g += old.get(v);
            ov.set(v, g);
        }
        vstore<T>(dx + e * VEC, ov);
    }
}
extern "C" int capmi_act_bwd(const void* dy, const void* y, void* dx, int accumulate, int64_t n, int act, int dtype, void* stream) {
    CAPMI_CHECK(dy && dx && (!act || y), "capmi_act_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_act_bwd", {
        CAPMI_CHECK(n % Vec<T>::N == 0, "capmi_act_bwd: n not a multiple of the vector width");
        int64_t nc = n / Vec<T>::N;
        hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)y, (T*)dx, accumulate, nc, act);
    });
    CAPMI_LAUNCH_CHECK("capmi_act_bwd");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void mean_rows_kernel(const T* __restrict__ x, T* out, int B, int K, int cpr) {
    constexpr int VEC = Vec<T>::N;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * cpr) return;
    int b = e / cpr, cc = e % cpr;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int k = 0; k < K; ++k) {
        Vec<T> xv = vload<T>(x + ((int64_t)(b * K + k) * cpr + cc) * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += xv.get(v);
    }
    Vec<T> ov;
#pragma unroll
    for (int v = 0; v < VEC; ++v) ov.set(v, acc[v] / (float)K);
    vstore<T>(out + (int64_t)e * VEC, ov);
}
extern "C" int capmi_mean_rows(const void* x, void* out, int B, int K, int C, int dtype, void* stream) {
    CAPMI_CHECK(x && out, "capmi_mean_rows: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_mean_rows", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_mean_rows: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        hipLaunchKernelGGL(mean_rows_kernel<T>, dim3(cdiv((int64_t)B * cpr, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, B, K, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_mean_rows");
    return 0;
}
// dx[b][k][:] += dout[b][:] / K
template <typename T>
__global__ __launch_bounds__(256) void mean_rows_bwd_kernel(const T* __restrict__ dout, T* dx, int64_t nchunks, int K, int cpr) {
    constexpr int VEC = Vec<T>::N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        int cc = (int)(e % cpr);
        int64_t b = e / cpr / K;
        Vec<T> g = vload<T>(dout + (b * cpr + cc) * VEC), old = vload<T>(dx + e * VEC), ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, old.get(v) + g.get(v) / (float)K);
        vstore<T>(dx + e * VEC, ov);
    }
}
extern "C" int capmi_mean_rows_bwd(const void* dout, void* dx, int B, int K, int C, int dtype, void* stream) {
    CAPMI_CHECK(dout && dx, "capmi_mean_rows_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_mean_rows_bwd", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_mean_rows_bwd: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        int64_t n = (int64_t)B * K * cpr;
        hipLaunchKernelGGL(mean_rows_bwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)dx, n, K, cpr);
    });
    CAPMI_LAUNCH_CHECK("capmi_mean_rows_bwd");
    return 0;
}

// ------------------------------------------------------------------ stem im2col (NCHW f32 feed -> patch matrix)
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ img, T* out, int B, int C, int H, int W,
                                                          int ks, int stride, int pad, int Ho, int Wo, int Kpad) {
    constexpr int VEC = Vec<T>::N;
    const int cpr = Kpad / VEC;
    const int64_t nchunks = (int64_t)B * Ho * Wo * cpr;
    const int K = ks * ks * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        int kc = (int)(e % cpr);
        int64_t m = e / cpr;
        int wo = (int)(m % Wo);
        int ho = (int)((m / Wo) % Ho);
        int b = (int)(m / ((int64_t)Wo * Ho));
        Vec<T> ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            int k = kc * VEC + v;
            float f = 0.f;
            if (k < K) {
                int c = k % C, tap = k / C;
                int r = tap / ks, q = tap % ks;
                int hi = ho * stride - pad + r, wi = wo * stride - pad + q;
                if (hi >= 0 && hi < H && wi >= 0 && wi < W) f = img[(((int64_t)b * C + c) * H + hi) * W + wi];
            }
            ov.set(v, f);
        }
        vstore<T>(out + e * VEC, ov);
    }
}
extern "C" int capmi_im2col_stem(const float* img, void* out, int B, int C, int H, int W, int k, int stride, int pad,
                                 int Ho, int Wo, int Kpad, int dtype, void* stream) {
    CAPMI_CHECK(img && out, "capmi_im2col_stem: null pointer");
    CAPMI_CHECK(Kpad >= k * k * C, "capmi_im2col_stem: Kpad too small");
    CAPMI_DISPATCH(dtype, "capmi_im2col_stem", {
        CAPMI_CHECK(Kpad % Vec<T>::N == 0, "capmi_im2col_stem: Kpad not a multiple of the vector width");
        int64_t n = (int64_t)B * Ho * Wo * (Kpad / Vec<T>::N);
        hipLaunchKernelGGL(im2col_stem_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, img, (T*)out, B, C, H, W, k, stride, pad, Ho, Wo, Kpad);
    });
    CAPMI_LAUNCH_CHECK("capmi_im2col_stem");
    return 0;
}

// ------------------------------------------------------------------ stem as space-to-depth (no patch matrix)
// A k x k / stride-2 convolution on the 3-channel feed equals a ceil(k/2) x ceil(k/2) / stride-1 convolution
// on the 2x2 space-to-depth image of the zero-padded feed: tap r = 2r' + ph of input row 2ho - pad + r is
// tap r' of block row ho + r', sub-row ph.  The blocks are written NHWC with channel (ph*2 + pw)*C + c,
// padded to Cs channels: [B][Hb][Wb][Cs], 27 MB at B = 64 instead of a 257 MB im2col matrix that the
// forward GEMM and the weight gradient would each read again; the stem then is an ordinary implicit GEMM.
template <typename T>
__global__ __launch_bounds__(256) void s2d_stem_kernel(const float* __restrict__ img, T* out, int B, int C, int H, int W, int pad,
                                                       int Hb, int Wb, int Cs) {
    const int64_t n = (int64_t)B * Hb * Wb;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int bw = (int)(e % Wb), bh = (int)((e / Wb) % Hb);
        const int64_t b = e / ((int64_t)Wb * Hb);
        T* o = out + e * Cs;
        constexpr int VEC = Vec<T>::N;
        for (int c0 = 0; c0 < Cs; c0 += VEC) {
            Vec<T> ov;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int ch = c0 + v;
                float f = 0.f;
                if (ch < 4 * C) {
                    const int c = ch % C, sub = ch / C, ph = sub >> 1, pw = sub & 1;
                    const int h = 2 * bh + ph - pad, w = 2 * bw + pw - pad;
                    if (h >= 0 && h < H && w >= 0 && w < W) f = img[((b * C + c) * H + h) * W + w];
                }
                ov.set(v, f);
            }
            vstore<T>(o + c0, ov);
        }
    }
}
extern "C" int capmi_s2d_stem(const float* img, void* out, int B, int C, int H, int W, int pad, int Hb, int Wb, int Cs,
                              int dtype, void* stream) {
    CAPMI_CHECK(img && out, "capmi_s2d_stem: null pointer");
    CAPMI_CHECK(Cs >= 4 * C && Cs % 8 == 0, "capmi_s2d_stem: Cs=%d must be a multiple of 8 and >= 4*C", Cs);
    CAPMI_DISPATCH(dtype, "capmi_s2d_stem", {
        hipLaunchKernelGGL(s2d_stem_kernel<T>, dim3(ew_grid((int64_t)B * Hb * Wb)), dim3(256), 0, (hipStream_t)stream, img, (T*)out, B, C, H, W,
                           pad, Hb, Wb, Cs);
    });
    CAPMI_LAUNCH_CHECK("capmi_s2d_stem");
    return 0;
}
// Filter-gradient slots of the space-to-depth form that no filter tap maps to (r = 2r'+ph >= k, the padding
// channels) pick up data products: zero them so that Adam leaves the structural zeros of the filter alone.
__global__ __launch_bounds__(256) void s2d_stem_mask_kernel(float* dw, int Cout, int C, int k, int kt, int Cs) {
    const int per = kt * kt * Cs, n = Cout * per;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int ch = e % Cs, q2 = (e / Cs) % kt, r2 = (e / (Cs * kt)) % kt;
        const int sub = ch / C, ph = sub >> 1, pw = sub & 1;
        if (ch >= 4 * C || 2 * r2 + ph >= k || 2 * q2 + pw >= k) dw[e] = 0.f;
    }
}
extern "C" int capmi_s2d_stem_mask_grad(float* dw, int Cout, int C, int k, int Cs, void* stream) {
    CAPMI_CHECK(dw, "capmi_s2d_stem_mask_grad: null pointer");
    const int kt = (k + 1) / 2;
    hipLaunchKernelGGL(s2d_stem_mask_kernel, dim3(cdiv(Cout * kt * kt * Cs, 256)), dim3(256), 0, (hipStream_t)stream, dw, Cout, C, k, kt, Cs);
    CAPMI_LAUNCH_CHECK("capmi_s2d_stem_mask_grad");
    return 0;
}

// ------------------------------------------------------------------ depthwise 3x3
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, T* y, int B, int Hi, int Wi,
                                                         int cpr, int stride, int Ho, int Wo) {
    constexpr int VEC = Vec<T>::N;
    const int64_t nchunks = (int64_t)B * Ho * Wo * cpr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        int cc = (int)(e % cpr);
        int64_t m = e / cpr;
        int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
        int64_t b = m / ((int64_t)Wo * Ho);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int hi = ho * stride - 1 + r;
            if (hi < 0 || hi >= Hi) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int wi = wo * stride - 1 + q;
                if (wi < 0 || wi >= Wi) continue;
                Vec<T> xv = vload<T>(x + (((b * Hi + hi) * Wi + wi) * cpr + cc) * VEC);
                Vec<T> wv = vload<T>(w + ((int64_t)(r * 3 + q) * cpr + cc) * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += xv.get(v) * wv.get(v);
            }
        }
        Vec<T> ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, acc[v]);
        vstore<T>(y + e * VEC, ov);
    }
}
extern "C" int capmi_dwconv3x3_fwd(const void* x, const void* w, void* y, int B, int Hi, int Wi, int C, int stride, int Ho,
                                   int Wo, int dtype, void* stream) {
    CAPMI_CHECK(x && w && y, "capmi_dwconv3x3_fwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_dwconv3x3_fwd", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_dwconv3x3_fwd: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        int64_t n = (int64_t)B * Ho * Wo * cpr;
        hipLaunchKernelGGL(dwconv_fwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (T*)y, B, Hi, Wi, cpr, stride, Ho, Wo);
    });
    CAPMI_LAUNCH_CHECK("capmi_dwconv3x3_fwd");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* dx, int B, int Hi, int Wi,
                                                              int cpr, int stride, int Ho, int Wo, int acc_flag) {
    constexpr int VEC = Vec<T>::N;
    const int64_t nchunks = (int64_t)B * Hi * Wi * cpr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        int cc = (int)(e % cpr);
        int64_t m = e / cpr;
        int wi = (int)(m % Wi), hi = (int)((m / Wi) % Hi);
        int64_t b = m / ((int64_t)Wi * Hi);
        float acc[VEC];
        if (acc_flag) {
            Vec<T> old = vload<T>(dx + e * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = old.get(v);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int hn = hi + 1 - r;
            if (hn < 0 || hn % stride) continue;
            int ho = hn / stride;
            if (ho >= Ho) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int wn = wi + 1 - q;
                if (wn < 0 || wn % stride) continue;
                int wo = wn / stride;
                if (wo >= Wo) continue;
                Vec<T> dv = vload<T>(dy + (((b * Ho + ho) * Wo + wo) * cpr + cc) * VEC);
                Vec<T> wv = vload<T>(w + ((int64_t)(r * 3 + q) * cpr + cc) * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += dv.get(v) * wv.get(v);
            }
        }
        Vec<T> ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, acc[v]);
        vstore<T>(dx + e * VEC, ov);
    }
}
extern "C" int capmi_dwconv3x3_bwd_data(const void* dy, const void* w, void* dx, int B, int Hi, int Wi, int C, int stride,
                                        int Ho, int Wo, int accumulate, int dtype, void* stream) {
    CAPMI_CHECK(dy && w && dx, "capmi_dwconv3x3_bwd_data: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_dwconv3x3_bwd_data", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_dwconv3x3_bwd_data: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        int64_t n = (int64_t)B * Hi * Wi * cpr;
        hipLaunchKernelGGL(dwconv_bwd_data_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)w, (T*)dx, B, Hi, Wi, cpr, stride, Ho, Wo, accumulate);
    });
    CAPMI_LAUNCH_CHECK("capmi_dwconv3x3_bwd_data");
    return 0;
}

// dw[r][q][c] += sum over output pixels dy[p][c] * x[p*stride + tap][c]
// DET (capmi_deterministic): ONE row block per column block (gridDim.x == 1), the row lanes' partials are folded into LDS
// in lane order and the block total is added by its only writer -- no atomics, fixed summation order.
template <typename T, bool DET>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* dw, int B, int Hi, int Wi,
                                                                int C, int stride, int Ho, int Wo, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float sacc[9 * 64 * VEC];      // col_layout(max_cpc = 64)
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * L.cpc * VEC; i += 256) sacc[i] = 0.f;
    __syncthreads();
    const int cpr = C / VEC;
    const int cc = tid % L.cpc, rr = tid / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    const bool active = rr < L.rp && chunk < cpr;
    float acc[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[t][v] = 0.f;
    if (active) {
        const int M = B * Ho * Wo;
        const int m_begin = blockIdx.x * L.rows_per_block;
        const int m_end = min(M, m_begin + L.rows_per_block);
        for (int m = m_begin + rr; m < m_end; m += L.rp) {
            int wo = m % Wo, ho = (m / Wo) % Ho;
            int64_t b = m / (Wo * Ho);
            Vec<T> dv = vload<T>(dy + ((int64_t)m * cpr + chunk) * VEC);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int hi = ho * stride - 1 + r;
                if (hi < 0 || hi >= Hi) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    int wi = wo * stride - 1 + q;
                    if (wi < 0 || wi >= Wi) continue;
                    Vec<T> xv = vload<T>(x + (((b * Hi + hi) * Wi + wi) * cpr + chunk) * VEC);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[r * 3 + q][v] += dv.get(v) * xv.get(v);
                }
            }
        }
        if constexpr (!DET) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) atomicAdd(&sacc[(t * L.cpc + cc) * VEC + v], acc[t][v]);
        }
    }
    if constexpr (DET) {
        for (int r = 0; r < L.rp; ++r) {
            if (active && rr == r) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sacc[(t * L.cpc + cc) * VEC + v] += acc[t][v];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = tid; i < 9 * L.cpc * VEC; i += 256) {
        int t = i / (L.cpc * VEC), j = i % (L.cpc * VEC);
        int c = blockIdx.y * L.cpc * VEC + j;
        if (c < C) {
            if constexpr (DET) dw[t * C + c] += sacc[i];
            else atomicAdd(&dw[t * C + c], sacc[i]);
        }
    }
}
extern "C" int capmi_dwconv3x3_bwd_weight(const void* x, const void* dy, float* dw, int B, int Hi, int Wi, int C, int stride,
                                          int Ho, int Wo, int dtype, void* stream) {
    CAPMI_CHECK(x && dy && dw, "capmi_dwconv3x3_bwd_weight: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_dwconv3x3_bwd_weight", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_dwconv3x3_bwd_weight: C not a multiple of the vector width");
        int gx, gy;
        ColLayout L = col_layout(B * Ho * Wo, C, Vec<T>::N, &gx, &gy, 64);
        if (capmi_deterministic()) {
            L.rows_per_block = B * Ho * Wo;
            hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, true>), dim3(1, gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, B, Hi, Wi, C, stride, Ho, Wo, L);
        } else {
            hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, B, Hi, Wi, C, stride, Ho, Wo, L);
        }
    });
    CAPMI_LAUNCH_CHECK("capmi_dwconv3x3_bwd_weight");
    return 0;
}

// ------------------------------------------------------------------ 3x3 stride-2 pad-1 max pool
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* y, uint8_t* idx, int B, int Hi, int Wi, int cpr, int Ho, int Wo) {
    constexpr int VEC = Vec<T>::N;
    const int64_t nchunks = (int64_t)B * Ho * Wo * cpr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nchunks; e += (int64_t)gridDim.x * 256) {
        int cc = (int)(e % cpr);
        int64_t m = e / cpr;
        int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
        int64_t b = m / ((int64_t)Wo * Ho);
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { best[v] = -INFINITY; bi[v] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int hi = ho * 2 - 1 + r;
            if (hi < 0 || hi >= Hi) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int wi = wo * 2 - 1 + q;
                if (wi < 0 || wi >= Wi) continue;
                Vec<T> xv = vload<T>(x + (((b * Hi + hi) * Wi + wi) * cpr + cc) * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float f = xv.get(v);
                    if (f > best[v]) { best[v] = f; bi[v] = r * 3 + q; }   // first maximum wins ties
                }
            }
        }
        Vec<T> ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { ov.set(v, best[v]); idx[e * VEC + v] = (uint8_t)bi[v]; }
        vstore<T>(y + e * VEC, ov);
    }
}
extern "C" int capmi_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                      int dtype, void* stream) {
    CAPMI_CHECK(x && y && idx, "capmi_maxpool3x3s2_fwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_maxpool3x3s2_fwd", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_maxpool3x3s2_fwd: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        int64_t n = (int64_t)B * Ho * Wo * cpr;
        hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, idx, B, Hi, Wi, cpr, Ho, Wo);
    });
    CAPMI_LAUNCH_CHECK("capmi_maxpool3x3s2_fwd");
    return 0;
}
// One thread per 2 x 2 block of input pixels (rows 2i, 2i+1; columns 2j, 2j+1) and channel chunk: with k = 3, stride 2,
// pad 1 the block's gradients come from exactly the four windows (i..i+1, j..j+1) -- pixel (even, even) is the centre of
// window (i, j), an odd row / column sits on the edge of two windows -- so the four dy vectors and their index bytes are
// loaded ONCE, up front and branch-free (clamped addresses, masked use), for the nine tap contributions they carry.  (One
// thread per input pixel loaded every window again for each of its up to four pixels, behind data-dependent branches:
// 1.4 TB/s on the 112 x 112 stem output.)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* dx, int B, int Hi, int Wi, int cpr, int Ho, int Wo) {
    constexpr int VEC = Vec<T>::N;
    typedef typename std::conditional<VEC == 8, uint64_t, uint32_t>::type IdxT;
    static_assert(sizeof(IdxT) == VEC, "one index byte per vector element");
    const int Wb = (Wi + 1) >> 1, Hb = (Hi + 1) >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Wb * cpr) return;
    const int j = t / cpr, cc = t - j * cpr;
    const int b = blockIdx.y / Hb, i = blockIdx.y - b * Hb;
    Vec<T> dv[2][2];
    IdxT iv[2][2];
    bool ok[2][2];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int ho = i + di, wo = j + dj;
            ok[di][dj] = ho < Ho && wo < Wo;
            const int64_t o = ((((int64_t)b * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1)) * cpr + cc) * VEC;
            dv[di][dj] = vload<T>(dy + o);
            iv[di][dj] = *reinterpret_cast<const IdxT*>(idx + o);
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int hi = 2 * i + a, wi = 2 * j + c;
            if (hi >= Hi || wi >= Wi) continue;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
            for (int di = 0; di <= a; ++di)
#pragma unroll
                for (int dj = 0; dj <= c; ++dj) {
                    const int r = a ? 2 - 2 * di : 1, q = c ? 2 - 2 * dj : 1;      // hi = 2 ho - 1 + r
                    if (!ok[di][dj]) continue;
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if ((int)((iv[di][dj] >> (8 * v)) & 0xff) == r * 3 + q) acc[v] += dv[di][dj].get(v);
                }
            Vec<T> ov;
#pragma unroll
            for (int v = 0; v < VEC; ++v) ov.set(v, acc[v]);
            vstore<T>(dx + ((((int64_t)b * Hi + hi) * Wi + wi) * cpr + cc) * VEC, ov);
        }
}
extern "C" int capmi_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                      int dtype, void* stream) {
    CAPMI_CHECK(dy && idx && dx, "capmi_maxpool3x3s2_bwd: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_maxpool3x3s2_bwd", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_maxpool3x3s2_bwd: C not a multiple of the vector width");
        int cpr = C / Vec<T>::N;
        const int Hb = (Hi + 1) / 2, Wb = (Wi + 1) / 2;
        CAPMI_CHECK((int64_t)B * Hb <= 65535 && (int64_t)Wi * cpr < (1ll << 30), "capmi_maxpool3x3s2_bwd: B*ceil(Hi/2)=%lld exceeds the grid", (long long)B * Hb);
        hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(cdiv(Wb * cpr, 256), B * Hb), dim3(256), 0, (hipStream_t)stream, (const T*)dy, idx, (T*)dx, B, Hi, Wi, cpr, Ho, Wo);
    });
    CAPMI_LAUNCH_CHECK("capmi_maxpool3x3s2_bwd");
    return 0;
}
