// Optimizer + weight-shadow kernels: Paddle-1.8 Adam over one flat f32 range, f32->bf16 casts,
// and the data-gradient weight form ([N][kh][kw][C] -> [C][kh][kw][N], taps flipped).
#include "common.h"

// G = float: the flat f32 gradient buffer; G = bf16: a bucket that went through the all-reduce in bf16 (capmi_adam_g16).
template <typename G> __device__ __forceinline__ f32x4 adam_load4(const G* g, int64_t i);
template <> __device__ __forceinline__ f32x4 adam_load4<float>(const float* g, int64_t i) { return reinterpret_cast<const f32x4*>(g)[i]; }
template <> __device__ __forceinline__ f32x4 adam_load4<bf16>(const bf16* g, int64_t i) {
    const bf16x4 v = reinterpret_cast<const bf16x4*>(g)[i];
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// Grid of the bulk updates (>= 2^21 16-byte chunks: the optimizer / shadow refresh of most of the model, which the engine runs
// on the side lane UNDER the backward pass): experiment knob CAPMI_BULK_WGS caps it so that the main lane's kernels find free
// slots next to it.
static inline int bulk_grid(int64_t chunks, int64_t elems) {
    static const int cap = getenv("CAPMI_BULK_WGS") ? atoi(getenv("CAPMI_BULK_WGS")) : 0;
    const int g = ew_grid(chunks);
    return (cap > 0 && elems >= (1ll << 23) && g > cap) ? cap : g;
}
// LOW: also store the updated parameter rounded to bf16 into `low` (the weight shadow the kernels read: capmi_cast of the same
// range right behind the update would read all of p again)
template <typename G, bool LOW = false>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const G* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_t, float b1, float b2, float eps, float clip, float gscale, bf16* __restrict__ low = nullptr) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = adam_load4<G>(g, i);
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = gv[k] * gscale;
            if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
            mv[k] = b1 * mv[k] + (1.f - b1) * gg;
            vv[k] = b2 * vv[k] + (1.f - b2) * gg * gg;
            pv[k] = pv[k] - lr_t * (mv[k] / (sqrtf(vv[k]) + eps));
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if constexpr (LOW) reinterpret_cast<bf16x4*>(low)[i] = bf16x4{(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = n4 * 4 + threadIdx.x;
        float gg = to_f32(g[i]) * gscale;
        if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
        float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        p[i] = p[i] - lr_t * (mm / (sqrtf(vv) + eps));
        if constexpr (LOW) low[i] = (bf16)p[i];
    }
}
extern "C" int capmi_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps,
                          float clip, float grad_scale, void* stream) {
    CAPMI_CHECK(p && g && m && v, "capmi_adam: null pointer");
    CAPMI_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "capmi_adam: buffers must be 16-byte aligned");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_kernel<float>, dim3(bulk_grid(n / 4 + 1, n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, b1, b2, eps, clip, grad_scale);
    CAPMI_LAUNCH_CHECK("capmi_adam");
    return 0;
}
/* capmi_adam + capmi_cast(p -> low16, bf16) of the same range in one pass: the bf16 weight shadow is written from the registers that
 * hold the updated parameter (the separate cast read every parameter again: 4 B each, under the bandwidth-bound backward pass). */
extern "C" int capmi_adam_shadow(float* p, const float* g, float* m, float* v, void* low16, int64_t n, float lr_t, float b1, float b2, float eps,
                                 float clip, float grad_scale, void* stream) {
    CAPMI_CHECK(p && g && m && v && low16, "capmi_adam_shadow: null pointer");
    CAPMI_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && (uintptr_t)low16 % 8 == 0, "capmi_adam_shadow: buffers must be 16-byte (shadow: 8-byte) aligned");
    if (n <= 0) return 0;
    hipLaunchKernelGGL((adam_kernel<float, true>), dim3(bulk_grid(n / 4 + 1, n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, b1, b2, eps, clip, grad_scale, (bf16*)low16);
    CAPMI_LAUNCH_CHECK("capmi_adam_shadow");
    return 0;
}
/* The same update from a bf16 gradient bucket (the data-parallel step's all-reduce payload, capmi_allreduce_bucket_bf16):
 * the gradient is widened to f32 on load, everything else -- moments, master weights, arithmetic -- is capmi_adam. */
extern "C" int capmi_adam_g16(float* p, const void* g16, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps,
                              float clip, float grad_scale, void* stream) {
    CAPMI_CHECK(p && g16 && m && v, "capmi_adam_g16: null pointer");
    CAPMI_CHECK(((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && (uintptr_t)g16 % 8 == 0, "capmi_adam_g16: buffers must be 16-byte (gradient: 8-byte) aligned");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_kernel<bf16>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, (const bf16*)g16, m, v, n, lr_t, b1, b2, eps, clip, grad_scale);
    CAPMI_LAUNCH_CHECK("capmi_adam_g16");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = from_f32<T>(src[i]);
}
// 8 elements per thread and iteration: two 16-byte loads, one 16-byte (bf16) store; the tail goes through the scalar form
__global__ __launch_bounds__(256) void cast8_kernel(const float* __restrict__ src, bf16* dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = (bf16)a[k]; o[4 + k] = (bf16)b[k]; }
        reinterpret_cast<bf16x8*>(dst)[i] = o;
    }
}
extern "C" int capmi_cast(const float* src, void* dst, int64_t n, int dtype, void* stream) {
    CAPMI_CHECK(src && dst, "capmi_cast: null pointer");
    if (n <= 0) return 0;
    if (dtype == CAPMI_BF16 && n >= 4096 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0) {
        const int64_t n8 = n / 8;
        hipLaunchKernelGGL(cast8_kernel, dim3(bulk_grid(n8, n)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n8);
        if (n > n8 * 8) hipLaunchKernelGGL(cast_kernel<bf16>, dim3(1), dim3(256), 0, (hipStream_t)stream, src + n8 * 8, (bf16*)dst + n8 * 8, n - n8 * 8);
        CAPMI_LAUNCH_CHECK("capmi_cast");
        return 0;
    }
    CAPMI_DISPATCH(dtype, "capmi_cast", {
        hipLaunchKernelGGL(cast_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, n);
    });
    CAPMI_LAUNCH_CHECK("capmi_cast");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void dgrad_form_kernel(const float* __restrict__ w, T* wt, int N, int kh, int kw, int C, int ldt) {
    const int64_t total = (int64_t)C * kh * kw * ldt;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int n = (int)(i % ldt);
        int64_t rest = i / ldt;
        int q = (int)(rest % kw);
        int r = (int)((rest / kw) % kh);
        int c = (int)(rest / ((int64_t)kw * kh));
        float f = 0.f;
        if (n < N) f = w[(((int64_t)n * kh + (kh - 1 - r)) * kw + (kw - 1 - q)) * C + c];
        wt[i] = from_f32<T>(f);
    }
}
extern "C" int capmi_weight_dgrad_form(const float* w, void* wt, int N, int kh, int kw, int C, int ldt, int dtype, void* stream) {
    CAPMI_CHECK(w && wt, "capmi_weight_dgrad_form: null pointer");
    CAPMI_CHECK(ldt >= N && (ldt == N || kh * kw == 1), "capmi_weight_dgrad_form: padded rows only for 1x1 weights");
    CAPMI_DISPATCH(dtype, "capmi_weight_dgrad_form", {
        hipLaunchKernelGGL(dgrad_form_kernel<T>, dim3(ew_grid((int64_t)C * kh * kw * ldt)), dim3(256), 0, (hipStream_t)stream, w, (T*)wt, N, kh, kw, C, ldt);
    });
    CAPMI_LAUNCH_CHECK("capmi_weight_dgrad_form");
    return 0;
}

// All data-gradient weight forms of a model in ONE launch: a job table in device memory, one job per
// (weight, run of 2 tiles); a tile is 64 output channels x 64 input channels of one tap, transposed
// through LDS so that both the f32 reads (along c) and the low-precision writes (along n) are coalesced.
struct DgradJob {
    long long src_off, dst_off;     // element offsets into the f32 master / the shadow buffer
    int N, kh, kw, C, ldt;
    int first;                      // first tile of this job; tile = (tap * ctiles + ct) * ntiles + nt
    int okh, okw;                   // taps of the OUTPUT form
    signed char rmap[4], qmap[4];   // output tap -> source tap
};
template <typename T>
__global__ __launch_bounds__(256) void dgrad_form_batched_kernel(const float* __restrict__ flat, T* shadow, const DgradJob* __restrict__ jobs) {
    __shared__ float tile[64][65];
    const DgradJob j = jobs[blockIdx.x];
    const int ntiles = (j.ldt + 63) / 64, ctiles = (j.C + 63) / 64;
    const int total = j.okh * j.okw * ctiles * ntiles;
    const float* w = flat + j.src_off;
    T* wt = shadow + j.dst_off;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int last = min(total, j.first + 2);
    for (int t = j.first; t < last; ++t) {
        const int nt = t % ntiles, ct = (t / ntiles) % ctiles, tap = t / (ntiles * ctiles);
        const int q = tap % j.okw, r = tap / j.okw;
        const int n0 = nt * 64, c0 = ct * 64;
        const long long tap_off = ((long long)j.rmap[r] * j.kw + j.qmap[q]) * j.C;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int n = n0 + ty + 4 * i, c = c0 + tx;
            tile[ty + 4 * i][tx] = (n < j.N && c < j.C) ? w[(long long)n * j.kh * j.kw * j.C + tap_off + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty + 4 * i, n = n0 + tx;
            if (c < j.C && n < j.ldt) wt[(((long long)c * j.okh + r) * j.okw + q) * j.ldt + n] = from_f32<T>(tile[tx][ty + 4 * i]);
        }
        __syncthreads();
    }
}
extern "C" int capmi_weight_dgrad_form_batched(const float* flat, void* shadow, const void* jobs, int njobs, int dtype, void* stream) {
    CAPMI_CHECK(flat && shadow && jobs, "capmi_weight_dgrad_form_batched: null pointer");
    if (njobs <= 0) return 0;
    CAPMI_DISPATCH(dtype, "capmi_weight_dgrad_form_batched", {
        hipLaunchKernelGGL(dgrad_form_batched_kernel<T>, dim3(njobs), dim3(256), 0, (hipStream_t)stream, flat, (T*)shadow, (const DgradJob*)jobs);
    });
    CAPMI_LAUNCH_CHECK("capmi_weight_dgrad_form_batched");
    return 0;
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float value, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = value;
}
extern "C" int capmi_fill_f32(float* p, float value, int64_t n, void* stream) {
    CAPMI_CHECK(p, "capmi_fill_f32: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, value, n);
    CAPMI_LAUNCH_CHECK("capmi_fill_f32");
    return 0;
}
