// Batch norm for gfx950 (train mode; fluid.layers.batch_norm, IC/model/MobileNetV2.py:112-117) on
// NHWC tensors viewed as [M = B*H*W][C], fused with relu/relu6 and the shortcut add.
//
// Statistics travel as "parts": for each block of `part_rows` consecutive rows and each channel
// the exact (mean, M2 = sum (x-mean)^2) of that block -- ws[part][C][2] f32.  A part is written
// with plain stores by exactly one producer (the conv epilogue in igemm.hip, or bn_stats here):
// deterministic, no atomics.  bn_finalize merges parts in f64 with Chan's formula, so there is no
// E[x^2]-E[x]^2 cancellation anywhere.  The elementwise kernels keep ONE channel chunk per thread
// (thread -> (cc, rr), ColLayout): coefficients live in registers, the row loop is pure 16-byte
// traffic, and every formula subtracts the mean BEFORE scaling (as the reference's op does).
#include "common.h"

// ------------------------------------------------------------------ statistics (standalone producer)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int M, int C, float* ws, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float part[256 * VEC];
    __shared__ float smean[256 * VEC];
    const int tid = threadIdx.x;
    const int cc = tid % L.cpc, rr = tid / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    const bool active = rr < L.rp && chunk * VEC < C;
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
    const float inv_n = 1.f / (float)(m_end - m_begin);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (active)
        for (int m = m_begin + rr; m < m_end; m += L.rp) {
            Vec<T> xv = vload<T>(x + (int64_t)m * C + chunk * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += xv.get(v);
        }
    block_col_reduce<VEC>(part, acc, cc, rr, L, active);
    if (active && rr == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) smean[cc * VEC + v] = acc[v] * inv_n;
    }
    __syncthreads();
    float mean[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { mean[v] = active ? smean[cc * VEC + v] : 0.f; acc[v] = 0.f; }
    if (active)      // second pass over the block's rows (just read: L2-resident)
        for (int m = m_begin + rr; m < m_end; m += L.rp) {
            Vec<T> xv = vload<T>(x + (int64_t)m * C + chunk * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) { float d = xv.get(v) - mean[v]; acc[v] += d * d; }
        }
    block_col_reduce<VEC>(part, acc, cc, rr, L, active);
    if (active && rr == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float* w = ws + ((int64_t)blockIdx.x * C + chunk * VEC + v) * 2;
            w[0] = mean[v];
            w[1] = acc[v];
        }
    }
}

extern "C" int capmi_bn_stats_part_rows(int M, int C, int dtype) {
    int gx, gy;
    return col_layout(M, C, dtype == CAPMI_F32 ? 4 : 8, &gx, &gy).rows_per_block;
}

extern "C" int capmi_bn_stats(const void* x, int M, int C, float* ws, int dtype, void* stream) {
    CAPMI_CHECK(x && ws, "capmi_bn_stats: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_bn_stats", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_stats: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = col_layout(M, C, Vec<T>::N, &gx, &gy);
        hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, M, C, ws, L);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_stats");
    return 0;
}

// ------------------------------------------------------------------ finalize (Chan merge, f64)
// Merge of parts [p0, p1) for 64 channels per workgroup: 4 thread groups stride over the parts.
__device__ __forceinline__ void merge_parts(const float* __restrict__ ws, int part_rows, int M, int C, int c, int p0, int p1,
                                            double (*red)[64], double* mean_out, double* m2_out) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double s = 0.0, cnt = 0.0;
    if (c < C)
        for (int p = p0 + ty; p < p1; p += 4) {
            int n = min(part_rows, M - p * part_rows);
            s += (double)n * (double)ws[((int64_t)p * C + c) * 2];
            cnt += n;
        }
    red[ty][tx] = s;
    red[4 + ty][tx] = cnt;
    __syncthreads();
    const double ntot = red[4][tx] + red[5][tx] + red[6][tx] + red[7][tx];
    const double mean = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (ntot > 0 ? ntot : 1.0);
    __syncthreads();
    double m2 = 0.0;
    if (c < C)
        for (int p = p0 + ty; p < p1; p += 4) {
            int n = min(part_rows, M - p * part_rows);
            const float* w = ws + ((int64_t)p * C + c) * 2;
            double d = (double)w[0] - mean;
            m2 += (double)w[1] + (double)n * d * d;
        }
    red[ty][tx] = m2;
    __syncthreads();
    *mean_out = mean;
    *m2_out = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// level 1 (only when there are many parts): groups of k parts -> one merged part each
__global__ __launch_bounds__(256) void bn_merge_kernel(const float* __restrict__ ws, int part_rows, int M, int C, int k, float* out) {
    __shared__ double red[8][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int nparts = (M + part_rows - 1) / part_rows;
    const int p0 = blockIdx.y * k, p1 = min(nparts, p0 + k);
    double mean, m2;
    merge_parts(ws, part_rows, M, C, c, p0, p1, red, &mean, &m2);
    if ((threadIdx.x >> 6) == 0 && c < C) {
        float* w = out + ((int64_t)blockIdx.y * C + c) * 2;
        w[0] = (float)mean;
        w[1] = (float)m2;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ ws, int part_rows, int M, int C, const float* scale,
                                                          float* run_mean, float* run_var, float momentum, float eps,
                                                          float* saved_mean, float* saved_invstd, float* coef_a, int update_running) {
    __shared__ double red[8][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int nparts = (M + part_rows - 1) / part_rows;
    double mean, m2;
    merge_parts(ws, part_rows, M, C, c, 0, nparts, red, &mean, &m2);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    const double var = m2 / (double)M;      // biased
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    saved_mean[c] = (float)mean;
    saved_invstd[c] = invstd;
    coef_a[c] = scale[c] * invstd;
    if (update_running) {
        run_mean[c] = run_mean[c] * momentum + (float)mean * (1.f - momentum);
        run_var[c] = run_var[c] * momentum + (float)var * (1.f - momentum);
    }
}

#define CAPMI_BN_MERGE_GROUPS 32     // merged groups (64 for C <= 128: few channel blocks, so more row groups); ws has room for 64 extra parts (capmi.h)

extern "C" int capmi_bn_finalize(float* ws, int part_rows, int M, int C, const float* scale, float* run_mean, float* run_var,
                                 float momentum, float eps, float* saved_mean, float* saved_invstd, float* coef_a,
                                 int update_running, void* stream) {
    CAPMI_CHECK(ws && scale && saved_mean && saved_invstd && coef_a, "capmi_bn_finalize: null pointer");
    CAPMI_CHECK(part_rows > 0 && M > 0, "capmi_bn_finalize: bad part_rows/M");
    CAPMI_CHECK(!update_running || (run_mean && run_var), "capmi_bn_finalize: running stats missing");
    const int nparts = cdiv(M, part_rows);
    const float* src = ws;
    int rows = part_rows;
    if (nparts > 2 * CAPMI_BN_MERGE_GROUPS) {
        const int k = cdiv(nparts, C <= 128 ? 2 * CAPMI_BN_MERGE_GROUPS : CAPMI_BN_MERGE_GROUPS);
        float* merged = ws + (int64_t)nparts * C * 2;
        hipLaunchKernelGGL(bn_merge_kernel, dim3(cdiv(C, 64), cdiv(nparts, k)), dim3(256), 0, (hipStream_t)stream, ws, part_rows, M, C, k, merged);
        src = merged;
        rows = part_rows * k;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, src, rows, M, C, scale, run_mean,
                       run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running);
    CAPMI_LAUNCH_CHECK("capmi_bn_finalize");
    return 0;
}

// ------------------------------------------------------------------ finalize + apply in one launch
// The forward critical path per layer was conv -> (merge) -> finalize -> apply: the finalize kernel is ~6 us of
// pure latency.  Here every apply workgroup merges the (<= 64) statistic groups of ITS channels itself
// (the lanes that share a channel chunk split the groups, Chan's formula in f64 through LDS) and goes straight
// on to normalise its rows; workgroup row 0 also stores the saved mean / invstd and updates the running stats.
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ ws, int part_rows, int nparts, int M, int C,
                                                                const float* __restrict__ scale, const float* __restrict__ offset,
                                                                float* run_mean, float* run_var, float momentum, float eps,
                                                                float* saved_mean, float* saved_invstd, int update_running,
                                                                const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                                int act, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    __shared__ double part[256 * VEC];
    const int cc = threadIdx.x % L.cpc, rr = threadIdx.x / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    const bool active = rr < L.rp && chunk * VEC < C;
    // pass 1: weighted mean of the group means
    double s[VEC], cnt = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.0;
    if (active)
        for (int p = rr; p < nparts; p += L.rp) {
            const double n = (double)min(part_rows, M - p * part_rows);
            const float* w = ws + ((int64_t)p * C + chunk * VEC) * 2;
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] += n * (double)w[2 * v];
            cnt += n;
        }
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[(rr * L.cpc + cc) * VEC + v] = s[v];
    __syncthreads();
    double mean[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        double t = 0.0;
        for (int r2 = 0; r2 < L.rp; ++r2) t += part[(r2 * L.cpc + cc) * VEC + v];
        mean[v] = t / (double)M;
    }
    __syncthreads();
    // pass 2: M2 = sum_p M2_p + n_p (mean_p - mean)^2
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.0;
    if (active)
        for (int p = rr; p < nparts; p += L.rp) {
            const double n = (double)min(part_rows, M - p * part_rows);
            const float* w = ws + ((int64_t)p * C + chunk * VEC) * 2;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double d = (double)w[2 * v] - mean[v];
                s[v] += (double)w[2 * v + 1] + n * d * d;
            }
        }
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[(rr * L.cpc + cc) * VEC + v] = s[v];
    __syncthreads();
    if (!active) return;
    float a[VEC], bo[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        double m2 = 0.0;
        for (int r2 = 0; r2 < L.rp; ++r2) m2 += part[(r2 * L.cpc + cc) * VEC + v];
        const int c = chunk * VEC + v;
        const double var = m2 / (double)M;      // biased
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        mu[v] = (float)mean[v];
        a[v] = scale[c] * invstd;
        bo[v] = offset[c];
        if (blockIdx.x == 0 && rr == 0) {
            saved_mean[c] = mu[v];
            saved_invstd[c] = invstd;
            if (update_running) {
                run_mean[c] = run_mean[c] * momentum + mu[v] * (1.f - momentum);
                run_var[c] = run_var[c] * momentum + (float)var * (1.f - momentum);
            }
        }
    }
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
#pragma unroll 8
    for (int m = m_begin + rr; m < m_end; m += L.rp) {
        const int64_t off = (int64_t)m * C + chunk * VEC;
        Vec<T> xv = vload<T>(x + off), rv, ov;
        if (res) rv = vload<T>(res + off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float f = a[v] * (xv.get(v) - mu[v]) + bo[v];
            if (res) f += rv.get(v);
            ov.set(v, apply_act(f, act));
        }
        vstore<T>(y + off, ov);
    }
}

extern "C" int capmi_bn_finalize_apply(float* ws, int part_rows, int M, int C, const float* scale, const float* offset, float* run_mean,
                                       float* run_var, float momentum, float eps, float* saved_mean, float* saved_invstd,
                                       int update_running, const void* x, const void* res, void* y, int act, int dtype, void* stream) {
    CAPMI_CHECK(ws && scale && offset && saved_mean && saved_invstd && x && y, "capmi_bn_finalize_apply: null pointer");
    CAPMI_CHECK(part_rows > 0 && M > 0, "capmi_bn_finalize_apply: bad part_rows/M");
    CAPMI_CHECK(!update_running || (run_mean && run_var), "capmi_bn_finalize_apply: running stats missing");
    int nparts = cdiv(M, part_rows);
    const float* src = ws;
    int rows = part_rows;
    if (nparts > 2 * CAPMI_BN_MERGE_GROUPS) {
        const int k = cdiv(nparts, C <= 128 ? 2 * CAPMI_BN_MERGE_GROUPS : CAPMI_BN_MERGE_GROUPS);
        float* merged = ws + (int64_t)nparts * C * 2;
        hipLaunchKernelGGL(bn_merge_kernel, dim3(cdiv(C, 64), cdiv(nparts, k)), dim3(256), 0, (hipStream_t)stream, ws, part_rows, M, C, k, merged);
        src = merged;
        rows = part_rows * k;
        nparts = cdiv(M, rows);
    }
    CAPMI_DISPATCH(dtype, "capmi_bn_finalize_apply", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_finalize_apply: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = ew_layout(M, C, Vec<T>::N, &gx, &gy, 64);      // <= 64 chunk columns: >= 4 lanes share a chunk's groups
        hipLaunchKernelGGL(bn_finalize_apply_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, src, rows, nparts, M, C, scale, offset,
                           run_mean, run_var, momentum, eps, saved_mean, saved_invstd, update_running, (const T*)x, (const T*)res, (T*)y, act, L);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_finalize_apply");
    return 0;
}

// ------------------------------------------------------------------ inference-mode coefficients (is_test)
// The exported inference model normalises with the RUNNING statistics (fluid batch_norm is_test=True, infer.py /
// save_inference_model): mean = running mean, a = scale / sqrt(running variance + eps); then capmi_bn_apply.
__global__ __launch_bounds__(256) void bn_inference_coef_kernel(const float* __restrict__ scale, const float* __restrict__ run_mean,
                                                                const float* __restrict__ run_var, float eps, float* mean, float* coef_a, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    coef_a[c] = scale[c] / sqrtf(run_var[c] + eps);
}
extern "C" int capmi_bn_inference_coef(const float* scale, const float* run_mean, const float* run_var, float eps, float* mean,
                                       float* coef_a, int C, void* stream) {
    CAPMI_CHECK(scale && run_mean && run_var && mean && coef_a, "capmi_bn_inference_coef: null pointer");
    hipLaunchKernelGGL(bn_inference_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, scale, run_mean, run_var, eps, mean, coef_a, C);
    CAPMI_LAUNCH_CHECK("capmi_bn_inference_coef");
    return 0;
}

// ------------------------------------------------------------------ apply: y = act(a*(x - mean) + offset (+ res))
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ ca,
                                                       const float* __restrict__ offset, const T* __restrict__ res, T* __restrict__ y,
                                                       int M, int C, int act, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    const int cc = threadIdx.x % L.cpc, rr = threadIdx.x / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    if (rr >= L.rp || chunk * VEC >= C) return;
    float a[VEC], b[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v += 4) {
        f32x4 av = *reinterpret_cast<const f32x4*>(ca + chunk * VEC + v), bv = *reinterpret_cast<const f32x4*>(offset + chunk * VEC + v);
        f32x4 mv = *reinterpret_cast<const f32x4*>(mean + chunk * VEC + v);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[v + k] = av[k]; b[v + k] = bv[k]; mu[v + k] = mv[k]; }
    }
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
#pragma unroll 8
    for (int m = m_begin + rr; m < m_end; m += L.rp) {
        const int64_t off = (int64_t)m * C + chunk * VEC;
        Vec<T> xv = vload<T>(x + off), rv, ov;
        if (res) rv = vload<T>(res + off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float f = a[v] * (xv.get(v) - mu[v]) + b[v];
            if (res) f += rv.get(v);
            ov.set(v, apply_act(f, act));
        }
        vstore<T>(y + off, ov);
    }
}

extern "C" int capmi_bn_apply(const void* x, const float* saved_mean, const float* coef_a, const float* offset, const void* res,
                              void* y, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(x && saved_mean && coef_a && offset && y, "capmi_bn_apply: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_bn_apply", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_apply: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = ew_layout(M, C, Vec<T>::N, &gx, &gy);
        hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, saved_mean, coef_a, offset,
                           (const T*)res, (T*)y, M, C, act, L);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_apply");
    return 0;
}

// ------------------------------------------------------------------ backward
// Stage 1: every workgroup reduces its row block to partial sums ws[block][2C] (plain stores).
// Stage 2: red[0..C) += sum dz, red[C..2C) += sum dz*xhat over the partials (fixed order).
// No global atomics: float atomics from every workgroup to the same cache line serialise at the
// memory side (~17 ns each) and dominated this kernel; this form is also deterministic.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float* ws,
                                                            int M, int C, int act, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float part[256 * VEC];
    const int tid = threadIdx.x;
    const int cc = tid % L.cpc, rr = tid / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    const bool active = rr < L.rp && chunk * VEC < C;
    float a1[VEC], a2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { a1[v] = 0.f; a2[v] = 0.f; }
    if (active) {
        float mu[VEC], is[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { mu[v] = mean[chunk * VEC + v]; is[v] = invstd[chunk * VEC + v]; }
        const int m_begin = blockIdx.x * L.rows_per_block;
        const int m_end = min(M, m_begin + L.rows_per_block);
#pragma unroll 8
        for (int m = m_begin + rr; m < m_end; m += L.rp) {
            const int64_t off = (int64_t)m * C + chunk * VEC;
            Vec<T> dv = vload<T>(dy + off), xv = vload<T>(x + off), yv;
            if (act) yv = vload<T>(y + off);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float dz = dv.get(v);
                if (act) dz *= act_grad_from_out(yv.get(v), act);
                a1[v] += dz;
                a2[v] += dz * (xv.get(v) - mu[v]) * is[v];
            }
        }
    }
    block_col_reduce<VEC>(part, a1, cc, rr, L, active);
    block_col_reduce<VEC>(part, a2, cc, rr, L, active);
    if (active && rr == 0) {
        float* w = ws + (int64_t)blockIdx.x * 2 * C + chunk * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { w[v] = a1[v]; w[C + v] = a2[v]; }
    }
}

// one workgroup per 16 entries of the [2C] vector; 16 thread groups stride over the partial rows
// with 8 independent loads in flight each (the partials are L2-resident: this is latency, not bytes)
template <int LANES>      // 16 columns x LANES part lanes per block
__global__ __launch_bounds__(16 * LANES) void bn_bwd_reduce_final_kernel(const float* __restrict__ ws, int nblocks, int n2c, float* red) {
    __shared__ float s[LANES][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + tx;
    float acc = 0.f;
    if (j < n2c) {
        int b = ty;
        for (; b + 7 * LANES < nblocks; b += 8 * LANES) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = ws[(int64_t)(b + u * LANES) * n2c + j];
            acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; b < nblocks; b += LANES) acc += ws[(int64_t)b * n2c + j];
    }
    s[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && j < n2c) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < LANES; ++r) tot += s[r][tx];
        red[j] += tot;
    }
}
static void launch_bwd_reduce_final(const float* ws, int nparts, int C, float* red, hipStream_t st) {
    if (nparts > 512) hipLaunchKernelGGL(bn_bwd_reduce_final_kernel<64>, dim3(cdiv(2 * C, 16)), dim3(1024), 0, st, ws, nparts, 2 * C, red);
    else hipLaunchKernelGGL(bn_bwd_reduce_final_kernel<16>, dim3(cdiv(2 * C, 16)), dim3(256), 0, st, ws, nparts, 2 * C, red);
}
/* Second stage alone: red[0..C) += sum_p ws[p][0][c], red[C..2C) += sum_p ws[p][1][c] -- for partial sums
 * produced by capmi_igemm_nt_bnred's epilogue. */
extern "C" int capmi_bn_bwd_reduce_final(const float* ws, int nparts, int C, float* red, void* stream) {
    CAPMI_CHECK(ws && red && nparts > 0 && C > 0, "capmi_bn_bwd_reduce_final: bad arguments");
    launch_bwd_reduce_final(ws, nparts, C, red, (hipStream_t)stream);
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_reduce_final");
    return 0;
}

static ColLayout bwd_reduce_layout(int M, int C, int vec, int* gx, int* gy) {
    ColLayout L = ew_layout(M, C, vec, gx, gy);
    // fewer, deeper workgroups: 2 per CU keep the partial workspace and the second stage small
    int target = 512 / *gy;
    if (target < 1) target = 1;
    int rpb = cdiv(M, target);
    rpb = cdiv(rpb, L.rp) * L.rp;
    if (rpb > L.rows_per_block) L.rows_per_block = rpb;
    *gx = cdiv(M, L.rows_per_block);
    return L;
}

extern "C" int capmi_bn_bwd_ws_floats(int M, int C, int dtype) {
    int gx, gy;
    bwd_reduce_layout(M, C, dtype == CAPMI_F32 ? 4 : 8, &gx, &gy);
    return gx * 2 * C;
}

extern "C" int capmi_bn_bwd_reduce(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                                   float* ws, float* red, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(dy && x && saved_mean && saved_invstd && ws && red, "capmi_bn_bwd_reduce: null pointer");
    CAPMI_CHECK(!act || y, "capmi_bn_bwd_reduce: activation mask needs y");
    int gx = 0, gy = 0;
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_reduce", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_bwd_reduce: C=%d not a multiple of %d", C, Vec<T>::N);
        ColLayout L = bwd_reduce_layout(M, C, Vec<T>::N, &gx, &gy);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)x,
                           (const T*)y, saved_mean, saved_invstd, ws, M, C, act, L);
    });
    launch_bwd_reduce_final(ws, gx, C, red, (hipStream_t)stream);
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_reduce");
    return 0;
}

// dx (+)= k1 * ((dz - m0) - (x - mu)*c2),  k1 = s*is, m0 = red0/M, c2 = is*red1/M;  dres (+)= dz.
// Both differences are formed BEFORE scaling: a spatially uniform dz (e.g. the reference's
// singleton attention) makes dz - mean(dz) cancel almost completely.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ red, T* dx, int dx_acc,
                                                           T* dres, int dres_acc, int M, int C, float inv_m, int act, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    const int cc = threadIdx.x % L.cpc, rr = threadIdx.x / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    if (rr >= L.rp || chunk * VEC >= C) return;
    float k1[VEC], c2[VEC], m0[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = chunk * VEC + v;
        const float is = invstd[c], s = scale[c];
        mu[v] = mean[c];
        k1[v] = s * is;
        c2[v] = is * red[C + c] * inv_m;
        m0[v] = red[c] * inv_m;
    }
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
#pragma unroll 2
    for (int m = m_begin + rr; m < m_end; m += L.rp) {
        const int64_t off = (int64_t)m * C + chunk * VEC;
        Vec<T> dv = vload<T>(dy + off), xv = vload<T>(x + off), yv, ov, rv, dxo, dro;
        if (act) yv = vload<T>(y + off);
        if (dx_acc) dxo = vload<T>(dx + off);
        if (dres && dres_acc) dro = vload<T>(dres + off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float dz = dv.get(v);
            if (act) dz *= act_grad_from_out(yv.get(v), act);
            float g = k1[v] * ((dz - m0[v]) - (xv.get(v) - mu[v]) * c2[v]);
            if (dx_acc) g += dxo.get(v);
            ov.set(v, g);
            if (dres) rv.set(v, dres_acc ? dz + dro.get(v) : dz);
        }
        vstore<T>(dx + off, ov);
        if (dres) vstore<T>(dres + off, rv);
    }
}

extern "C" int capmi_bn_bwd_apply(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                                  const float* scale, const float* red, void* dx, int dx_accumulate, void* dres, int dres_accumulate,
                                  int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(dy && x && saved_mean && saved_invstd && scale && red && dx, "capmi_bn_bwd_apply: null pointer");
    CAPMI_CHECK(!act || y, "capmi_bn_bwd_apply: activation mask needs y");
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_apply", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_bwd_apply: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = ew_layout(M, C, Vec<T>::N, &gx, &gy);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)y,
                           saved_mean, saved_invstd, scale, red, (T*)dx, dx_accumulate, (T*)dres, dres_accumulate, M, C,
                           1.f / (float)M, act, L);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_apply");
    return 0;
}
