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
#include <atomic>
#include <type_traits>
#include "common.h"
#include "bn_merge.h"

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
// Merge of parts [p0, p1) for 64 channels per workgroup: 4 thread groups stride over the parts.  ONE pass: every thread
// folds its parts with Chan's update in f64 (loads in batches of four, independent of the arithmetic), the four
// partial results meet in LDS and are folded in a fixed order.  (The first form took two passes -- weighted mean,
// then M2 around it -- i.e. twice the dependent load rounds in kernels that are nothing but latency.)
// (chan_fold, merge_parts, bn_finalize_body: bn_merge.h -- shared with the convolution epilogue's last-arriver finalize)
// level 1 (only when there are many parts): groups of k parts -> one merged part each
__global__ __launch_bounds__(256) void bn_merge_kernel(const float* __restrict__ ws, int part_rows, int M, int C, int k, float* out) {
    __shared__ double red[12][64];
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
    __shared__ double red[12][64];
    bn_finalize_body<false>(ws, part_rows, M, C, blockIdx.x, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, red);
}

// Merge level and finalize in ONE launch (the merge -> finalize pair was two ~6 us dependent kernels on the forward chain
// of 41 of the 53 layers): every workgroup merges its group of parts as bn_merge_kernel does, stores the result
// write-through, drains the store and adds to the arrival counter of its 64-channel block; the workgroup that arrives
// LAST (it has seen every other group's add, and each add followed that group's drained store) resets the counter and
// finalizes the block from the merged parts, loaded past L1 / the XCD-local L2.  Same arithmetic and the same fixed fold
// order as the two kernels; nobody waits for anybody, so the launch drains whatever the arrival order.
__global__ __launch_bounds__(256) void bn_merge_finalize_kernel(const float* __restrict__ ws, int part_rows, int M, int C, int k, float* merged,
                                                                unsigned* counters, const float* scale, float* run_mean, float* run_var,
                                                                float momentum, float eps, float* saved_mean, float* saved_invstd,
                                                                float* coef_a, int update_running) {
    __shared__ double red[12][64];
    __shared__ unsigned s_last;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int nparts = (M + part_rows - 1) / part_rows;
    const int p0 = blockIdx.y * k, p1 = min(nparts, p0 + k);
    double mean, m2;
    merge_parts<false>(ws, part_rows, M, C, c, p0, p1, red, &mean, &m2);
    if ((threadIdx.x >> 6) == 0 && c < C) {
        const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, (float)mean) |
                                     ((unsigned long long)__builtin_bit_cast(unsigned, (float)m2) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(merged + ((int64_t)blockIdx.y * C + c) * 2), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores have left the chip's caches
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == gridDim.y - 1 ? 1u : 0u;
        if (old == gridDim.y - 1) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    bn_finalize_body<true>(merged, part_rows * k, M, C, blockIdx.x, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a,
                           update_running, red);
}

// Arrival counters of bn_merge_finalize_kernel: a pool of 128 sets of 32 per DEVICE (one counter per 64-channel block,
// C <= 2048), handed out round-robin per launch and zero again when a launch ends.  Contract (capmi.h): two launches share
// a set only when they are 128 capmi_bn_finalize calls apart in host order on that device, so the caller must not keep
// more than 127 later fused launches in flight next to an unfinished one -- the engine joins its lanes at the end of every
// step (< 60 such launches per step).  The symbol's address differs per device: it is looked up once per device.
__device__ unsigned bn_arrival_counters[128 * 32];
static unsigned* next_arrival_counters() {
    constexpr int MAXDEV = 64;
    static std::atomic<unsigned> next[MAXDEV];
    static std::atomic<unsigned*> base[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    unsigned* b = base[dev].load(std::memory_order_acquire);
    if (!b) {
        if (hipGetSymbolAddress((void**)&b, HIP_SYMBOL(bn_arrival_counters)) != hipSuccess || !b) return nullptr;
        base[dev].store(b, std::memory_order_release);
    }
    return b + (next[dev].fetch_add(1) % 128u) * 32u;
}


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
        static const int fuse = getenv("CAPMI_BN_FUSE_MERGE") ? atoi(getenv("CAPMI_BN_FUSE_MERGE")) : 1;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);       // a captured launch would freeze its counter set into the graph
        unsigned* counters = (fuse && C <= 2048 && cap == hipStreamCaptureStatusNone) ? next_arrival_counters() : nullptr;
        if (counters) {
            hipLaunchKernelGGL(bn_merge_finalize_kernel, dim3(cdiv(C, 64), cdiv(nparts, k)), dim3(256), 0, (hipStream_t)stream, ws, part_rows, M, C, k,
                               merged, counters, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running);
            CAPMI_LAUNCH_CHECK("capmi_bn_finalize");
            return 0;
        }
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
// every batch-norm layer of a model in ONE launch (the decode graph had one ~5 us launch per layer in front of its conv)
struct BnCoefJob { const float* scale; const float* run_mean; const float* run_var; float* mean; float* coef_a; long long C; };
static_assert(sizeof(BnCoefJob) == 48, "capmi.h documents 48-byte jobs");
__global__ __launch_bounds__(256) void bn_inference_coef_batched_kernel(const BnCoefJob* __restrict__ jobs, float eps) {
    const BnCoefJob j = jobs[blockIdx.y];
    for (int c = blockIdx.x * 256 + threadIdx.x; c < (int)j.C; c += gridDim.x * 256) {
        j.mean[c] = j.run_mean[c];
        j.coef_a[c] = j.scale[c] / sqrtf(j.run_var[c] + eps);
    }
}
extern "C" int capmi_bn_inference_coef_batched(const void* jobs, int njobs, int max_c, float eps, void* stream) {
    CAPMI_CHECK(jobs && njobs >= 1 && njobs <= 65535 && max_c >= 1, "capmi_bn_inference_coef_batched: bad arguments");
    hipLaunchKernelGGL(bn_inference_coef_batched_kernel, dim3(cdiv(max_c, 256) < 8 ? cdiv(max_c, 256) : 8, njobs), dim3(256), 0, (hipStream_t)stream,
                       (const BnCoefJob*)jobs, eps);
    CAPMI_LAUNCH_CHECK("capmi_bn_inference_coef_batched");
    return 0;
}
extern "C" int capmi_bn_inference_coef(const float* scale, const float* run_mean, const float* run_var, float eps, float* mean,
                                       float* coef_a, int C, void* stream) {
    CAPMI_CHECK(scale && run_mean && run_var && mean && coef_a, "capmi_bn_inference_coef: null pointer");
    hipLaunchKernelGGL(bn_inference_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, scale, run_mean, run_var, eps, mean, coef_a, C);
    CAPMI_LAUNCH_CHECK("capmi_bn_inference_coef");
    return 0;
}

// ------------------------------------------------------------------ apply: y = act(a*(x - mean) + offset (+ res))
// ACT >= 0: compile-time activation (none / relu / relu6, what the encoders use); ACT < 0: the run-time code `act`.
// U rows are loaded as one batch before any arithmetic (see bn_bwd_reduce_kernel: with the run-time switch in the loop
// the compiler kept one or two loads in flight per thread).
// MASK (capmi_bn_apply_mask, bf16): next to y the kernel stores one bit per element -- "the activation's derivative at this
// output is 1", from the ROUNDED value it stores -- as a byte per 8 channels ([M][C / 8]); the data-gradient epilogue that
// masks this tensor's gradient reads the bits (CAPMI_DACT_BITMASK, igemm.hip EPI 6) instead of y itself: 1/16 of the bytes.
template <typename T, int ACT, bool RES, bool MASK = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ ca,
                                                       const float* __restrict__ offset, const T* __restrict__ res, T* __restrict__ y,
                                                       uint8_t* __restrict__ mask, int M, int C, int act, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    static_assert(!MASK || (VEC == 8 && ACT > 0), "the bit mask is a byte per thread and row: bf16, relu / relu6");
    constexpr int U = 4;
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
    const int64_t step = (int64_t)L.rp * C;
    auto one = [&](const Vec<T>& xv, const Vec<T>& rv, int64_t o) {
        Vec<T> ov;
        unsigned bits = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float f = a[v] * (xv.get(v) - mu[v]) + b[v];
            if (RES) f += rv.get(v);
            ov.set(v, apply_act(f, ACT >= 0 ? ACT : act));
            if (MASK) bits |= (act_grad_from_out(ov.get(v), ACT) != 0.f ? 1u : 0u) << v;
        }
        if (MASK) mask[o >> 3] = (uint8_t)bits;
        return ov;
    };
    int m = m_begin + rr;
    int64_t off = (int64_t)m * C + (int64_t)chunk * VEC;
    for (; m + (U - 1) * L.rp < m_end; m += U * L.rp, off += U * step) {
        Vec<T> xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = vload<T>(x + off + u * step);
            if (RES) rv[u] = vload<T>(res + off + u * step);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the batch: the scheduler otherwise sinks every load to its use (one in flight)
#pragma unroll
        for (int u = 0; u < U; ++u) vstore<T>(y + off + u * step, one(xv[u], rv[u], off + u * step));
    }
    for (; m < m_end; m += L.rp, off += step) {
        Vec<T> xv = vload<T>(x + off), rv;
        if (RES) rv = vload<T>(res + off);
        vstore<T>(y + off, one(xv, rv, off));
    }
}

template <typename T, bool RES>
static void bn_apply_launch(int act, dim3 grid, hipStream_t st, const T* x, const float* mean, const float* ca, const float* offset, const T* res,
                            T* y, int M, int C, const ColLayout& L) {
    switch (act) {
        case CAPMI_ACT_NONE: hipLaunchKernelGGL((bn_apply_kernel<T, CAPMI_ACT_NONE, RES>), grid, dim3(256), 0, st, x, mean, ca, offset, res, y, nullptr, M, C, act, L); break;
        case CAPMI_ACT_RELU: hipLaunchKernelGGL((bn_apply_kernel<T, CAPMI_ACT_RELU, RES>), grid, dim3(256), 0, st, x, mean, ca, offset, res, y, nullptr, M, C, act, L); break;
        case CAPMI_ACT_RELU6: hipLaunchKernelGGL((bn_apply_kernel<T, CAPMI_ACT_RELU6, RES>), grid, dim3(256), 0, st, x, mean, ca, offset, res, y, nullptr, M, C, act, L); break;
        default: hipLaunchKernelGGL((bn_apply_kernel<T, -1, RES>), grid, dim3(256), 0, st, x, mean, ca, offset, res, y, nullptr, M, C, act, L); break;
    }
}

extern "C" int capmi_bn_apply(const void* x, const float* saved_mean, const float* coef_a, const float* offset, const void* res,
                              void* y, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(x && saved_mean && coef_a && offset && y, "capmi_bn_apply: null pointer");
    CAPMI_DISPATCH(dtype, "capmi_bn_apply", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_apply: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = ew_layout(M, C, Vec<T>::N, &gx, &gy);
        if (res) bn_apply_launch<T, true>(act, dim3(gx, gy), (hipStream_t)stream, (const T*)x, saved_mean, coef_a, offset, (const T*)res, (T*)y, M, C, L);
        else bn_apply_launch<T, false>(act, dim3(gx, gy), (hipStream_t)stream, (const T*)x, saved_mean, coef_a, offset, (const T*)res, (T*)y, M, C, L);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_apply");
    return 0;
}
/* capmi_bn_apply that also writes the activation-derivative bit mask of its output (bf16, relu / relu6, C % 8 == 0): see capmi.h. */
extern "C" int capmi_bn_apply_mask(const void* x, const float* saved_mean, const float* coef_a, const float* offset, const void* res,
                                   void* y, uint8_t* mask, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(x && saved_mean && coef_a && offset && y && mask, "capmi_bn_apply_mask: null pointer");
    CAPMI_CHECK(dtype == CAPMI_BF16 && C % 8 == 0 && (act == CAPMI_ACT_RELU || act == CAPMI_ACT_RELU6),
                "capmi_bn_apply_mask: bf16 tensors with C %% 8 == 0 and a relu / relu6 activation only (C=%d act=%d dtype=%d)", C, act, dtype);
    typedef bf16 T;
    int gx, gy;
    ColLayout L = ew_layout(M, C, 8, &gx, &gy);
    const dim3 grid(gx, gy);
    hipStream_t st = (hipStream_t)stream;
#define CAPMI_BN_APPLY_MASK(ACT_, RES_)                                                                                                            \
    hipLaunchKernelGGL((bn_apply_kernel<T, ACT_, RES_, true>), grid, dim3(256), 0, st, (const T*)x, saved_mean, coef_a, offset, (const T*)res, (T*)y, \
                       mask, M, C, act, L)
    if (act == CAPMI_ACT_RELU) { if (res) CAPMI_BN_APPLY_MASK(CAPMI_ACT_RELU, true); else CAPMI_BN_APPLY_MASK(CAPMI_ACT_RELU, false); }
    else { if (res) CAPMI_BN_APPLY_MASK(CAPMI_ACT_RELU6, true); else CAPMI_BN_APPLY_MASK(CAPMI_ACT_RELU6, false); }
#undef CAPMI_BN_APPLY_MASK
    CAPMI_LAUNCH_CHECK("capmi_bn_apply_mask");
    return 0;
}

// ------------------------------------------------------------------ finalize inside the apply launch, from accumulator rows
// capmi_bn_stat_apply: the statistics arrive as four rows of per-channel (sum (v - s), sum (v - s)^2) -- s = `shift`, the batch mean
// of the previous step -- that the convolution's epilogue added
// up with f32 atomics (capmi_igemm_nt_stat, igemm.hip EPI 8).  Every workgroup forms mean / invstd / coef_a of ITS channels from
// the 4 x 2 numbers per channel in its prologue -- the row loads are issued next to the first batch of tensor loads and
// consumed behind it (as capmi_bn_bwd_apply_spread does: a prologue that waited for them first cost every workgroup one more
// memory round trip) -- and the first row block also writes saved mean / invstd / coef_a and the running statistics (one
// writer per channel).  The merge + finalize launch (capmi_bn_finalize: 41 x ~10 us on the forward chain at cfg 2) is gone;
// unlike capmi_bn_finalize_apply (lesson 30) no workgroup merges parts.  One-pass variance in f32 around the shift (var =
// E[d^2] - E[d]^2, d = v - s, clamped at 0): relative error ~1e-7 x (1 + (mean - s)^2 / var), i.e. that of the exact merge from
// the second step on; the f32 engine and deterministic mode keep the exact two-level merge.
template <int ACT, bool RES, bool MASK>
__global__ __launch_bounds__(256) void bn_stat_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ rows, const float* __restrict__ shift, float inv_m, const float* __restrict__ scale,
                                                            const float* __restrict__ offset, float* run_mean, float* run_var, float momentum, float eps,
                                                            float* saved_mean, float* saved_invstd, float* coef_a, int update_running,
                                                            const bf16* __restrict__ res, bf16* __restrict__ y, uint8_t* __restrict__ mask, int M, int C, ColLayout L) {
    typedef bf16 T;
    constexpr int VEC = 8, U = 4;
    const int cc = threadIdx.x % L.cpc, rr = threadIdx.x / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    if (rr >= L.rp || chunk * VEC >= C) return;
    f32x4 t0[4][2], t1[4][2], scv[2], ofv[2], rmv[2], rvv[2], shv[2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            t0[j][q] = *reinterpret_cast<const f32x4*>(rows + (int64_t)j * 2 * C + chunk * VEC + 4 * q);
            t1[j][q] = *reinterpret_cast<const f32x4*>(rows + (int64_t)j * 2 * C + C + chunk * VEC + 4 * q);
        }
    const bool writer = blockIdx.x == 0 && rr == 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        scv[q] = *reinterpret_cast<const f32x4*>(scale + chunk * VEC + 4 * q);
        ofv[q] = *reinterpret_cast<const f32x4*>(offset + chunk * VEC + 4 * q);
        shv[q] = *reinterpret_cast<const f32x4*>(shift + chunk * VEC + 4 * q);
        if (writer && update_running) {
            rmv[q] = *reinterpret_cast<const f32x4*>(run_mean + chunk * VEC + 4 * q);
            rvv[q] = *reinterpret_cast<const f32x4*>(run_var + chunk * VEC + 4 * q);
        }
    }
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
    const int64_t step = (int64_t)L.rp * C;
    int m = m_begin + rr;
    int64_t off = (int64_t)m * C + (int64_t)chunk * VEC;
    Vec<T> xv[U], rv[U];
    auto load = [&](int64_t o0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = vload<T>(x + o0 + u * step);
            if (RES) rv[u] = vload<T>(res + o0 + u * step);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool first = m + (U - 1) * L.rp < m_end;
    if (first) load(off);
    float a[VEC], b[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0 += t0[j][v / 4][v % 4]; s1 += t1[j][v / 4][v % 4]; }
        const float dm = s0 * inv_m;                                 // mean - shift
        const float mean = shv[v / 4][v % 4] + dm;
        const float var = fmaxf(s1 * inv_m - dm * dm, 0.f);          // biased
        const float invstd = 1.f / sqrtf(var + eps);
        mu[v] = mean;
        a[v] = scv[v / 4][v % 4] * invstd;
        b[v] = ofv[v / 4][v % 4];
        if (writer) {
            const int c = chunk * VEC + v;
            saved_mean[c] = mean;
            saved_invstd[c] = invstd;
            coef_a[c] = a[v];
            if (update_running) {
                run_mean[c] = rmv[v / 4][v % 4] * momentum + mean * (1.f - momentum);
                run_var[c] = rvv[v / 4][v % 4] * momentum + var * (1.f - momentum);
            }
        }
    }
    auto one = [&](const Vec<T>& xi, const Vec<T>& ri, int64_t o) {
        Vec<T> ov;
        unsigned bits = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float f = a[v] * (xi.get(v) - mu[v]) + b[v];
            if (RES) f += ri.get(v);
            ov.set(v, apply_act(f, ACT));
            if (MASK) bits |= (act_grad_from_out(ov.get(v), ACT) != 0.f ? 1u : 0u) << v;
        }
        if (MASK) mask[o >> 3] = (uint8_t)bits;
        return ov;
    };
    if (first) {
#pragma unroll
        for (int u = 0; u < U; ++u) vstore<T>(y + off + u * step, one(xv[u], rv[u], off + u * step));
        m += U * L.rp;
        off += U * step;
    }
    for (; m + (U - 1) * L.rp < m_end; m += U * L.rp, off += U * step) {
        load(off);
#pragma unroll
        for (int u = 0; u < U; ++u) vstore<T>(y + off + u * step, one(xv[u], rv[u], off + u * step));
    }
    for (; m < m_end; m += L.rp, off += step) {
        Vec<T> xi = vload<T>(x + off), ri;
        if (RES) ri = vload<T>(res + off);
        vstore<T>(y + off, one(xi, ri, off));
    }
}

/* capmi_bn_finalize + capmi_bn_apply (or capmi_bn_apply_mask when `mask` is given) behind capmi_igemm_nt_stat, as ONE launch:
 * see capmi.h.  Deterministic mode: the two (three) launches on the exact parts. */
extern "C" int capmi_bn_stat_apply(const void* x, float* parts, int part_rows, const float* stat_rows, const float* shift, int M, int C, const float* scale,
                                   const float* offset, float* run_mean, float* run_var, float momentum, float eps, float* saved_mean,
                                   float* saved_invstd, float* coef_a, int update_running, const void* res, void* y, uint8_t* mask, int act,
                                   int dtype, void* stream) {
    CAPMI_CHECK(x && parts && stat_rows && shift && scale && offset && saved_mean && saved_invstd && coef_a && y, "capmi_bn_stat_apply: null pointer");
    CAPMI_CHECK(shift != saved_mean, "capmi_bn_stat_apply: the shift must not be the buffer the new mean is written to (other workgroups still read it)");
    CAPMI_CHECK(!update_running || (run_mean && run_var), "capmi_bn_stat_apply: running stats missing");
    CAPMI_CHECK(dtype == CAPMI_BF16 && C % 8 == 0, "capmi_bn_stat_apply: bf16 tensors with C %% 8 == 0 only (C=%d dtype=%d)", C, dtype);
    CAPMI_CHECK(act == CAPMI_ACT_NONE || act == CAPMI_ACT_RELU || act == CAPMI_ACT_RELU6, "capmi_bn_stat_apply: activation %d (none / relu / relu6)", act);
    CAPMI_CHECK(!mask || act != CAPMI_ACT_NONE, "capmi_bn_stat_apply: the bit mask is for relu / relu6 outputs");
    CAPMI_CHECK(((uintptr_t)stat_rows | (uintptr_t)shift | (uintptr_t)scale | (uintptr_t)offset | (uintptr_t)run_mean | (uintptr_t)run_var) % 16 == 0,
                "capmi_bn_stat_apply: per-channel vectors must be 16-byte aligned");
    if (capmi_deterministic()) {
        if (capmi_bn_finalize(parts, part_rows, M, C, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, stream)) return 1;
        if (mask) return capmi_bn_apply_mask(x, saved_mean, coef_a, offset, res, y, mask, M, C, act, dtype, stream);
        return capmi_bn_apply(x, saved_mean, coef_a, offset, res, y, M, C, act, dtype, stream);
    }
    int gx, gy;
    ColLayout L = ew_layout(M, C, 8, &gx, &gy);
    const dim3 grid(gx, gy);
    hipStream_t st = (hipStream_t)stream;
    const float inv_m = 1.f / (float)M;
#define CAPMI_BN_SA(ACT_, RES_, MASK_)                                                                                                          \
    hipLaunchKernelGGL((bn_stat_apply_kernel<ACT_, RES_, MASK_>), grid, dim3(256), 0, st, (const bf16*)x, stat_rows, shift, inv_m, scale, offset, run_mean, \
                       run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, (const bf16*)res, (bf16*)y, mask, M, C, L)
#define CAPMI_BN_SA_ACT(ACT_)                                                                              \
    do {                                                                                                   \
        if (mask) { if (res) CAPMI_BN_SA(ACT_, true, true); else CAPMI_BN_SA(ACT_, false, true); }         \
        else { if (res) CAPMI_BN_SA(ACT_, true, false); else CAPMI_BN_SA(ACT_, false, false); }            \
    } while (0)
    if (act == CAPMI_ACT_RELU) CAPMI_BN_SA_ACT(CAPMI_ACT_RELU);
    else if (act == CAPMI_ACT_RELU6) CAPMI_BN_SA_ACT(CAPMI_ACT_RELU6);
    else { if (res) CAPMI_BN_SA(CAPMI_ACT_NONE, true, false); else CAPMI_BN_SA(CAPMI_ACT_NONE, false, false); }
#undef CAPMI_BN_SA_ACT
#undef CAPMI_BN_SA
    CAPMI_LAUNCH_CHECK("capmi_bn_stat_apply");
    return 0;
}

// the bf16 (T) value the apply launch would have stored for conv output x: shared by capmi_bn_stat_apply_pool and its backward pair
template <typename T, int ACT>
__device__ __forceinline__ float pool_act_out(float x, float coef_a, float mean, float offset) {
    return to_f32(from_f32<T>(apply_act(__builtin_fmaf(coef_a, x - mean, offset), ACT)));
}

// ------------------------------------------------------------------ finalize + apply + 3x3 / stride-2 max pool in one launch (the stem)
// conv -> batch_norm -> relu -> pool2d(max, 3, 2, 1): the activated 112 x 112 tensor has ONE reader, the pool, and the backward pass
// needs of it only the sign of every element (capmi_bn_bwd_*_pool_x form that from the conv output and the saved coefficients).  So
// it is never written: a thread owns one pooled pixel x 8 channels, loads the nine conv-output vectors of its window up front (clamped
// addresses, masked use), normalises / activates / rounds each to the bf16 the apply launch would have stored (pool_act_out) and keeps
// the first maximum -- the values and the argmax map of capmi_bn_stat_apply + capmi_maxpool3x3s2_fwd.  Per step at cfg 2: 103 MB
// written + 103 MB read less on the forward chain, 2 x 103 MB of reads less in the backward pair.  Prologue as bn_stat_apply_kernel's.
template <int ACT>
__global__ __launch_bounds__(256) void bn_stat_apply_pool_kernel(const bf16* __restrict__ x, const float* __restrict__ rows, const float* __restrict__ shift, float inv_m,
                                                                 const float* __restrict__ scale, const float* __restrict__ offset, float* run_mean, float* run_var,
                                                                 float momentum, float eps, float* saved_mean, float* saved_invstd, float* coef_a, int update_running,
                                                                 bf16* __restrict__ pooled, uint8_t* __restrict__ idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                                 int rows_per_block) {
    typedef bf16 T;
    constexpr int VEC = 8;
    const int cpr = C / VEC;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int j = t / cpr, cc = t - j * cpr;
    if (j >= Wo) return;
    f32x4 t0[4][2], t1[4][2], scv[2], ofv[2], rmv[2], rvv[2], shv[2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            t0[r][q] = *reinterpret_cast<const f32x4*>(rows + (int64_t)r * 2 * C + cc * VEC + 4 * q);
            t1[r][q] = *reinterpret_cast<const f32x4*>(rows + (int64_t)r * 2 * C + C + cc * VEC + 4 * q);
        }
    const bool writer = blockIdx.y == 0 && j == 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        scv[q] = *reinterpret_cast<const f32x4*>(scale + cc * VEC + 4 * q);
        ofv[q] = *reinterpret_cast<const f32x4*>(offset + cc * VEC + 4 * q);
        shv[q] = *reinterpret_cast<const f32x4*>(shift + cc * VEC + 4 * q);
        if (writer && update_running) {
            rmv[q] = *reinterpret_cast<const f32x4*>(run_mean + cc * VEC + 4 * q);
            rvv[q] = *reinterpret_cast<const f32x4*>(run_var + cc * VEC + 4 * q);
        }
    }
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(B * Ho, r_begin + rows_per_block);
    Vec<T> win[3][3];
    bool ok[3][3];
    auto load_row = [&](int r) {
        const int b = r / Ho, ho = r - b * Ho;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int hi = 2 * ho - 1 + rr, wi = 2 * j - 1 + q;
                ok[rr][q] = hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
                const int hc = min(max(hi, 0), Hi - 1), wc = min(max(wi, 0), Wi - 1);
                win[rr][q] = vload<T>(x + ((((int64_t)b * Hi + hc) * Wi + wc) * cpr + cc) * VEC);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (r_begin < r_end) load_row(r_begin);
    float a[VEC], bo[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 += t0[r][v / 4][v % 4]; s1 += t1[r][v / 4][v % 4]; }
        const float dm = s0 * inv_m;                                 // mean - shift
        const float mean = shv[v / 4][v % 4] + dm;
        const float var = fmaxf(s1 * inv_m - dm * dm, 0.f);          // biased
        const float invstd = 1.f / sqrtf(var + eps);
        mu[v] = mean;
        a[v] = scv[v / 4][v % 4] * invstd;
        bo[v] = ofv[v / 4][v % 4];
        if (writer) {
            const int c = cc * VEC + v;
            saved_mean[c] = mean;
            saved_invstd[c] = invstd;
            coef_a[c] = a[v];
            if (update_running) {
                run_mean[c] = rmv[v / 4][v % 4] * momentum + mean * (1.f - momentum);
                run_var[c] = rvv[v / 4][v % 4] * momentum + var * (1.f - momentum);
            }
        }
    }
    for (int r = r_begin; r < r_end; ++r) {
        if (r != r_begin) load_row(r);
        float best[VEC];
        unsigned long long bi = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) best[v] = -INFINITY;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (!ok[rr][q]) continue;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float f = pool_act_out<T, ACT>(win[rr][q].get(v), a[v], mu[v], bo[v]);
                    if (f > best[v]) {                                   // first maximum wins ties (capmi_maxpool3x3s2_fwd)
                        best[v] = f;
                        bi = (bi & ~(0xffull << (8 * v))) | ((unsigned long long)(rr * 3 + q) << (8 * v));
                    }
                }
            }
        Vec<T> ov;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.set(v, best[v]);
        const int64_t o = ((int64_t)r * Wo + j) * C + cc * VEC;
        vstore<T>(pooled + o, ov);
        *reinterpret_cast<unsigned long long*>(idx + o) = bi;
    }
}

extern "C" int capmi_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo, int dtype, void* stream);

/* capmi_bn_stat_apply + capmi_maxpool3x3s2_fwd as ONE launch that never writes the activated tensor (see above; capmi.h).
 * y_scratch [B,Hi,Wi,C] is written in deterministic mode only, where the entry point IS the exact path: capmi_bn_finalize on the
 * parts, capmi_bn_apply into y_scratch, capmi_maxpool3x3s2_fwd. */
extern "C" int capmi_bn_stat_apply_pool(const void* x, float* parts, int part_rows, const float* stat_rows, const float* shift, int B, int Hi, int Wi, int C,
                                        int Ho, int Wo, const float* scale, const float* offset, float* run_mean, float* run_var, float momentum, float eps,
                                        float* saved_mean, float* saved_invstd, float* coef_a, int update_running, void* y_scratch, void* pooled, uint8_t* idx,
                                        int act, int dtype, void* stream) {
    CAPMI_CHECK(x && parts && stat_rows && shift && scale && offset && saved_mean && saved_invstd && coef_a && y_scratch && pooled && idx, "capmi_bn_stat_apply_pool: null pointer");
    CAPMI_CHECK(shift != saved_mean, "capmi_bn_stat_apply_pool: the shift must not be the buffer the new mean is written to");
    CAPMI_CHECK(!update_running || (run_mean && run_var), "capmi_bn_stat_apply_pool: running stats missing");
    CAPMI_CHECK(dtype == CAPMI_BF16 && C % 8 == 0 && 256 % (C / 8) == 0, "capmi_bn_stat_apply_pool: bf16 tensors, C / 8 a divisor of 256 (C=%d dtype=%d)", C, dtype);
    CAPMI_CHECK(act == CAPMI_ACT_RELU || act == CAPMI_ACT_RELU6, "capmi_bn_stat_apply_pool: the pooled layer's activation must be relu or relu6");
    CAPMI_CHECK(Ho == (Hi + 1) / 2 && Wo == (Wi + 1) / 2, "capmi_bn_stat_apply_pool: 3x3 / stride 2 / pad 1 pooling of %dx%d gives %dx%d", Hi, Wi, (Hi + 1) / 2, (Wi + 1) / 2);
    CAPMI_CHECK(((uintptr_t)stat_rows | (uintptr_t)shift | (uintptr_t)scale | (uintptr_t)offset | (uintptr_t)run_mean | (uintptr_t)run_var) % 16 == 0,
                "capmi_bn_stat_apply_pool: per-channel vectors must be 16-byte aligned");
    const int M = B * Hi * Wi;
    if (capmi_deterministic()) {
        if (capmi_bn_finalize(parts, part_rows, M, C, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, stream)) return 1;
        if (capmi_bn_apply(x, saved_mean, coef_a, offset, nullptr, y_scratch, M, C, act, dtype, stream)) return 1;
        return capmi_maxpool3x3s2_fwd(y_scratch, pooled, idx, B, Hi, Wi, C, Ho, Wo, dtype, stream);
    }
    const int cpr = C / 8;
    const int gx = cdiv(Wo * cpr, 256);
    int rpb = cdiv((int64_t)B * Ho * gx, 4096);            // ~16 workgroups per CU, a few pooled rows each
    if (rpb < 1) rpb = 1;
    const dim3 grid(gx, cdiv(B * Ho, rpb));
    const float inv_m = 1.f / (float)M;
    if (act == CAPMI_ACT_RELU)
        hipLaunchKernelGGL((bn_stat_apply_pool_kernel<CAPMI_ACT_RELU>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, stat_rows, shift, inv_m, scale, offset, run_mean,
                           run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, (bf16*)pooled, idx, B, Hi, Wi, C, Ho, Wo, rpb);
    else
        hipLaunchKernelGGL((bn_stat_apply_pool_kernel<CAPMI_ACT_RELU6>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, stat_rows, shift, inv_m, scale, offset, run_mean,
                           run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, (bf16*)pooled, idx, B, Hi, Wi, C, Ho, Wo, rpb);
    CAPMI_LAUNCH_CHECK("capmi_bn_stat_apply_pool");
    return 0;
}

// ------------------------------------------------------------------ backward
// Stage 1: every workgroup reduces its row block to partial sums ws[block][2C] (plain stores).
// Stage 2: red[0..C) += sum dz, red[C..2C) += sum dz*xhat over the partials (fixed order).
// No global atomics: float atomics from every workgroup to the same cache line serialise at the
// memory side (~17 ns each) and dominated this kernel; this form is also deterministic.
// ACT is a template parameter and the loads of U rows are issued as one batch before any arithmetic: with a run-time
// activation switch the loop compiled to a branch per element and 2-3 loads in flight per thread -- 2.8 TB/s on
// cold tensors in the model (5.7 alone on MALL-warm ones), the largest kernel family on the main lane.
// SPREAD: no second stage.  The block totals go out as f32 atomics into BN_SPREAD_ROWS accumulator rows acc[rows][2C] (row =
// workgroup index mod rows: a quarter of the grid adds to any one address, and a wave instruction covers 256 contiguous
// bytes -- the memory-side atomic units' full rate, MI355X_MICROARCH.md); bn_bwd_apply sums the rows in its prologue.
// The dependent ~9 us second-stage launch behind every one of the ~50 reductions of a step is gone; the price is the
// summation order of a row (not fixed): capmi_deterministic() keeps the two-stage form.
constexpr int BN_SPREAD_ROWS = 4;       // (capmi.h documents room for 8)
template <typename T, int ACT, bool SPREAD = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float* ws,
                                                            int M, int C, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    constexpr int U = 4;                            // rows per batch; two batches (U * (2 or 3) 16-byte loads each) in flight
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
        const int64_t col = (int64_t)chunk * VEC, step = (int64_t)L.rp * C;
        auto add_row = [&](const Vec<T>& dv, const Vec<T>& xv, const Vec<T>& yv) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float dz = dv.get(v);
                if (ACT) dz *= act_grad_from_out(yv.get(v), ACT);
                a1[v] += dz;
                a2[v] += dz * (xv.get(v) - mu[v]) * is[v];
            }
        };
        // two register buffers: the loads of batch k + 1 are in flight while batch k is summed (2 workgroups per CU leave
        // few other waves to hide them behind)
        int m = m_begin + rr;
        int64_t off = (int64_t)m * C + col;
        Vec<T> dA[U], xA[U], yA[U], dB[U], xB[U], yB[U];
        auto load = [&](Vec<T> (&dv)[U], Vec<T> (&xv)[U], Vec<T> (&yv)[U], int64_t o) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                dv[u] = vload<T>(dy + o + u * step);
                xv[u] = vload<T>(x + o + u * step);
                if (ACT) yv[u] = vload<T>(y + o + u * step);
            }
        };
        auto sum = [&](Vec<T> (&dv)[U], Vec<T> (&xv)[U], Vec<T> (&yv)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) add_row(dv[u], xv[u], yv[u]);
        };
        const int nb = m < m_end ? (m_end - m + L.rp - 1) / L.rp / U : 0;       // full batches of U rows
        if (nb > 0) load(dA, xA, yA, off);
        for (int b = 0; b < nb; b += 2) {
            if (b + 1 < nb) load(dB, xB, yB, off + (int64_t)(b + 1) * U * step);
            sum(dA, xA, yA);
            if (b + 1 < nb) {
                if (b + 2 < nb) load(dA, xA, yA, off + (int64_t)(b + 2) * U * step);
                sum(dB, xB, yB);
            }
        }
        m += nb * U * L.rp;
        off += (int64_t)nb * U * step;
        for (; m < m_end; m += L.rp, off += step) {
            Vec<T> dv = vload<T>(dy + off), xv = vload<T>(x + off), yv;
            if (ACT) yv = vload<T>(y + off);
            add_row(dv, xv, yv);
        }
    }
    block_col_reduce<VEC>(part, a1, cc, rr, L, active);
    block_col_reduce<VEC>(part, a2, cc, rr, L, active);
    if constexpr (SPREAD) {
        // block totals -> LDS -> one atomic per thread and entry, 64 lanes on 64 consecutive floats
        __syncthreads();
        const int ncol = L.cpc * VEC;                       // columns of this column block (<= 256 * VEC floats of `part` = 2 x ncol for cpc <= 128)
        float* tot = part;
        if (active && rr == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) tot[cc * VEC + v] = a1[v];
        }
        __syncthreads();
        float* row = ws + (int64_t)(blockIdx.x & (BN_SPREAD_ROWS - 1)) * 2 * C + (int64_t)blockIdx.y * ncol;
        for (int i = tid; i < ncol; i += 256)
            if (blockIdx.y * ncol + i < C) __hip_atomic_fetch_add(row + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (active && rr == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) tot[cc * VEC + v] = a2[v];
        }
        __syncthreads();
        for (int i = tid; i < ncol; i += 256)
            if (blockIdx.y * ncol + i < C) __hip_atomic_fetch_add(row + C + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (active && rr == 0) {
            float* w = ws + (int64_t)blockIdx.x * 2 * C + chunk * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { w[v] = a1[v]; w[C + v] = a2[v]; }
        }
    }
}

// one workgroup per 16 entries of the [2C] vector; 16 thread groups stride over the partial rows
// with 8 independent loads in flight each (the partials are L2-resident: this is latency, not bytes)
template <int LANES>      // 16 columns x LANES part lanes per block
__global__ __launch_bounds__(16 * LANES) void bn_bwd_reduce_final_kernel(const float* __restrict__ ws, int nblocks, int n2c, float* red) {
    __shared__ float s[LANES][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + tx;
    const float before = (ty == 0 && j < n2c) ? red[j] : 0.f;      // fetched now, added at the end: off the dependent chain
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
        red[j] = before + tot;
    }
}
static void launch_bwd_reduce_final(const float* ws, int nparts, int C, float* red, hipStream_t st) {
    if (nparts >= 128) hipLaunchKernelGGL(bn_bwd_reduce_final_kernel<64>, dim3(cdiv(2 * C, 16)), dim3(1024), 0, st, ws, nparts, 2 * C, red);
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

template <typename T, bool SPREAD = false>
static int bwd_reduce_launch(int act, int gx, int gy, hipStream_t st, const T* dy, const T* x, const T* y, const float* mean, const float* invstd,
                             float* ws, int M, int C, const ColLayout& L) {
    switch (act) {
        case CAPMI_ACT_NONE: hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, CAPMI_ACT_NONE, SPREAD>), dim3(gx, gy), dim3(256), 0, st, dy, x, y, mean, invstd, ws, M, C, L); return 0;
        case CAPMI_ACT_RELU: hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, CAPMI_ACT_RELU, SPREAD>), dim3(gx, gy), dim3(256), 0, st, dy, x, y, mean, invstd, ws, M, C, L); return 0;
        case CAPMI_ACT_RELU6: hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, CAPMI_ACT_RELU6, SPREAD>), dim3(gx, gy), dim3(256), 0, st, dy, x, y, mean, invstd, ws, M, C, L); return 0;
    }
    capmi_set_error("capmi_bn_bwd_reduce: unsupported activation %d", act);
    return 1;
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
        if (bwd_reduce_launch<T>(act, gx, gy, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)y, saved_mean, saved_invstd, ws, M, C, L)) return 1;
    });
    launch_bwd_reduce_final(ws, gx, C, red, (hipStream_t)stream);
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_reduce");
    return 0;
}

/* capmi_bn_bwd_reduce without its second stage: the block totals are ADDED (f32 atomics) into acc8[8][2C], which the caller
 * zeroes once per step and hands to capmi_bn_bwd_apply_spread -- that kernel sums the eight rows in its prologue and adds
 * the result to `red` ([d offset | d scale]).  In deterministic mode (capmi.h) the pair falls back to the two-stage form
 * through ws / red, so both entry points take both sets of buffers. */
extern "C" int capmi_bn_bwd_reduce_spread(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                                          float* ws, float* red, float* acc8, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(acc8, "capmi_bn_bwd_reduce_spread: null accumulator");
    if (capmi_deterministic()) return capmi_bn_bwd_reduce(dy, x, y, saved_mean, saved_invstd, ws, red, M, C, act, dtype, stream);
    CAPMI_CHECK(dy && x && saved_mean && saved_invstd, "capmi_bn_bwd_reduce_spread: null pointer");
    CAPMI_CHECK(!act || y, "capmi_bn_bwd_reduce_spread: activation mask needs y");
    int gx = 0, gy = 0;
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_reduce_spread", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_bwd_reduce_spread: C=%d not a multiple of %d", C, Vec<T>::N);
        ColLayout L = bwd_reduce_layout(M, C, Vec<T>::N, &gx, &gy);
        if (bwd_reduce_launch<T, true>(act, gx, gy, (hipStream_t)stream, (const T*)dy, (const T*)x, (const T*)y, saved_mean, saved_invstd, acc8, M, C, L)) return 1;
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_reduce_spread");
    return 0;
}

// dx (+)= k1 * ((dz - m0) - (x - mu)*c2),  k1 = s*is, m0 = red0/M, c2 = is*red1/M;  dres (+)= dz.
// Both differences are formed BEFORE scaling: a spatially uniform dz (e.g. the reference's
// singleton attention) makes dz - mean(dz) cancel almost completely.
// ACT: compile-time activation of the layer's output (none / relu / relu6); DRES: 0 = no residual gradient, 1 = store
// dz, 2 = accumulate dz; DXACC: dx accumulates.  Batched loads as in bn_bwd_reduce_kernel.
// SPREAD (capmi_bn_bwd_apply_spread): the sums are the BN_SPREAD_ROWS accumulator rows of capmi_bn_bwd_reduce_spread.  Their
// loads are issued next to the first batch of row loads and added up (in row order) only behind it, like the other
// per-channel constants: a prologue that waited for them first cost every workgroup one more memory round trip (+4 us per
// launch, measured).  The first row block also adds the sums to `red` (one writer per channel).
template <typename T, int ACT, int DRES, bool DXACC, bool SPREAD = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ red, const float* __restrict__ acc,
                                                           float* red_out, T* dx, T* dres, int M, int C, float inv_m, ColLayout L) {
    constexpr int VEC = Vec<T>::N;
    constexpr int U = (ACT ? 1 : 0) + (DRES == 2 ? 1 : 0) + (DXACC ? 1 : 0) >= 2 ? 2 : 4;      // register budget
    const int cc = threadIdx.x % L.cpc, rr = threadIdx.x / L.cpc;
    const int chunk = blockIdx.y * L.cpc + cc;
    if (rr >= L.rp || chunk * VEC >= C) return;
    // raw per-channel scalars now, the constants formed from them only after the first batch of row loads is in flight:
    // formed here, the row loads would wait one memory latency for them at the start of every workgroup
    float k1[VEC], c2[VEC], m0[VEC], mu[VEC];
    float is_[VEC], sc_[VEC], r0_[VEC], r1_[VEC];
    f32x4 t0[SPREAD ? BN_SPREAD_ROWS : 1][VEC / 4], t1[SPREAD ? BN_SPREAD_ROWS : 1][VEC / 4];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = chunk * VEC + v;
        is_[v] = invstd[c];
        sc_[v] = scale[c];
        mu[v] = mean[c];
        if constexpr (!SPREAD) {
            r0_[v] = red[c];
            r1_[v] = red[C + c];
        }
    }
    if constexpr (SPREAD) {
#pragma unroll
        for (int j = 0; j < BN_SPREAD_ROWS; ++j)
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                t0[j][q] = *reinterpret_cast<const f32x4*>(acc + (int64_t)j * 2 * C + chunk * VEC + 4 * q);
                t1[j][q] = *reinterpret_cast<const f32x4*>(acc + (int64_t)j * 2 * C + C + chunk * VEC + 4 * q);
            }
    }
    const int m_begin = blockIdx.x * L.rows_per_block;
    const int m_end = min(M, m_begin + L.rows_per_block);
    const int64_t step = (int64_t)L.rp * C;
    auto one = [&](int64_t off, const Vec<T>& dv, const Vec<T>& xv, const Vec<T>& yv, const Vec<T>& dxo, const Vec<T>& dro) {
        Vec<T> ov, rv;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float dz = dv.get(v);
            if (ACT) dz *= act_grad_from_out(yv.get(v), ACT);
            float g = k1[v] * ((dz - m0[v]) - (xv.get(v) - mu[v]) * c2[v]);
            if (DXACC) g += dxo.get(v);
            ov.set(v, g);
            if (DRES) rv.set(v, DRES == 2 ? dz + dro.get(v) : dz);
        }
        vstore<T>(dx + off, ov);
        if (DRES) vstore<T>(dres + off, rv);
    };
    int m = m_begin + rr;
    int64_t off = (int64_t)m * C + (int64_t)chunk * VEC;
    Vec<T> dv[U], xv[U], yv[U], dxo[U], dro[U];
    auto load = [&](int64_t o0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t o = o0 + u * step;
            dv[u] = vload<T>(dy + o);
            xv[u] = vload<T>(x + o);
            if (ACT) yv[u] = vload<T>(y + o);
            if (DXACC) dxo[u] = vload<T>(dx + o);
            if (DRES == 2) dro[u] = vload<T>(dres + o);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the batch: the scheduler otherwise sinks every load to its use
    };
    const bool first = m + (U - 1) * L.rp < m_end;
    if (first) load(off);
    if constexpr (SPREAD) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < BN_SPREAD_ROWS; ++j) { s0 += t0[j][v / 4][v % 4]; s1 += t1[j][v / 4][v % 4]; }
            r0_[v] = s0;
            r1_[v] = s1;
        }
        if (blockIdx.x == 0 && rr == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int c = chunk * VEC + v;
                red_out[c] += r0_[v];
                red_out[C + c] += r1_[v];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        k1[v] = sc_[v] * is_[v];
        c2[v] = is_[v] * r1_[v] * inv_m;
        m0[v] = r0_[v] * inv_m;
    }
    if (first) {
#pragma unroll
        for (int u = 0; u < U; ++u) one(off + u * step, dv[u], xv[u], yv[u], dxo[u], dro[u]);
        m += U * L.rp;
        off += U * step;
    }
    for (; m + (U - 1) * L.rp < m_end; m += U * L.rp, off += U * step) {
        load(off);
#pragma unroll
        for (int u = 0; u < U; ++u) one(off + u * step, dv[u], xv[u], yv[u], dxo[u], dro[u]);
    }
    for (; m < m_end; m += L.rp, off += step) {
        Vec<T> dv = vload<T>(dy + off), xv = vload<T>(x + off), yv, dxo, dro;
        if (ACT) yv = vload<T>(y + off);
        if (DXACC) dxo = vload<T>(dx + off);
        if (DRES == 2) dro = vload<T>(dres + off);
        one(off, dv, xv, yv, dxo, dro);
    }
}

template <typename T, int ACT, int DRES>
static void bn_bwd_apply_launch2(bool dxacc, dim3 grid, hipStream_t st, const T* dy, const T* x, const T* y, const float* mean, const float* invstd,
                                 const float* scale, float* red, const float* acc, T* dx, T* dres, int M, int C, const ColLayout& L) {
    const float inv_m = 1.f / (float)M;
    if (acc) {
        if (dxacc) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, ACT, DRES, true, true>), grid, dim3(256), 0, st, dy, x, y, mean, invstd, scale, nullptr, acc, red, dx, dres, M, C, inv_m, L);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, ACT, DRES, false, true>), grid, dim3(256), 0, st, dy, x, y, mean, invstd, scale, nullptr, acc, red, dx, dres, M, C, inv_m, L);
    } else {
        if (dxacc) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, ACT, DRES, true, false>), grid, dim3(256), 0, st, dy, x, y, mean, invstd, scale, red, nullptr, nullptr, dx, dres, M, C, inv_m, L);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, ACT, DRES, false, false>), grid, dim3(256), 0, st, dy, x, y, mean, invstd, scale, red, nullptr, nullptr, dx, dres, M, C, inv_m, L);
    }
}
template <typename T, int ACT>
static void bn_bwd_apply_launch1(int dres_mode, bool dxacc, dim3 grid, hipStream_t st, const T* dy, const T* x, const T* y, const float* mean,
                                 const float* invstd, const float* scale, float* red, const float* acc8, T* dx, T* dres, int M, int C, const ColLayout& L) {
    if (dres_mode == 0) bn_bwd_apply_launch2<T, ACT, 0>(dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L);
    else if (dres_mode == 1) bn_bwd_apply_launch2<T, ACT, 1>(dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L);
    else bn_bwd_apply_launch2<T, ACT, 2>(dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L);
}
template <typename T>
static int bn_bwd_apply_launch(int act, int dres_mode, bool dxacc, dim3 grid, hipStream_t st, const T* dy, const T* x, const T* y, const float* mean,
                               const float* invstd, const float* scale, float* red, const float* acc8, T* dx, T* dres, int M, int C, const ColLayout& L) {
    switch (act) {
        case CAPMI_ACT_NONE: bn_bwd_apply_launch1<T, CAPMI_ACT_NONE>(dres_mode, dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L); return 0;
        case CAPMI_ACT_RELU: bn_bwd_apply_launch1<T, CAPMI_ACT_RELU>(dres_mode, dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L); return 0;
        case CAPMI_ACT_RELU6: bn_bwd_apply_launch1<T, CAPMI_ACT_RELU6>(dres_mode, dxacc, grid, st, dy, x, y, mean, invstd, scale, red, acc8, dx, dres, M, C, L); return 0;
    }
    capmi_set_error("capmi_bn_bwd_apply: unsupported activation %d", act);
    return 1;
}

static int bn_bwd_apply_impl(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                             const float* scale, float* red, const float* acc8, void* dx, int dx_accumulate, void* dres, int dres_accumulate,
                             int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(dy && x && saved_mean && saved_invstd && scale && red && dx, "capmi_bn_bwd_apply: null pointer");
    CAPMI_CHECK(!act || y, "capmi_bn_bwd_apply: activation mask needs y");
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_apply", {
        CAPMI_CHECK(C % Vec<T>::N == 0, "capmi_bn_bwd_apply: C=%d not a multiple of %d", C, Vec<T>::N);
        int gx, gy;
        ColLayout L = ew_layout(M, C, Vec<T>::N, &gx, &gy);
        if (bn_bwd_apply_launch<T>(act, dres ? (dres_accumulate ? 2 : 1) : 0, dx_accumulate != 0, dim3(gx, gy), (hipStream_t)stream, (const T*)dy,
                                   (const T*)x, (const T*)y, saved_mean, saved_invstd, scale, red, acc8, (T*)dx, (T*)dres, M, C, L)) return 1;
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_apply");
    return 0;
}
extern "C" int capmi_bn_bwd_apply(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                                  const float* scale, const float* red, void* dx, int dx_accumulate, void* dres, int dres_accumulate,
                                  int M, int C, int act, int dtype, void* stream) {
    return bn_bwd_apply_impl(dy, x, y, saved_mean, saved_invstd, scale, const_cast<float*>(red), nullptr, dx, dx_accumulate, dres, dres_accumulate,
                             M, C, act, dtype, stream);
}
/* The consumer of capmi_bn_bwd_reduce_spread: the channel sums are the eight rows of acc8 (summed in row order in the
 * kernel's prologue); the first row block adds them to red ([d offset | d scale]).  Deterministic mode: capmi_bn_bwd_apply. */
extern "C" int capmi_bn_bwd_apply_spread(const void* dy, const void* x, const void* y, const float* saved_mean, const float* saved_invstd,
                                         const float* scale, float* red, const float* acc8, void* dx, int dx_accumulate, void* dres,
                                         int dres_accumulate, int M, int C, int act, int dtype, void* stream) {
    CAPMI_CHECK(acc8 && ((uintptr_t)acc8 % 16 == 0) && C % 4 == 0, "capmi_bn_bwd_apply_spread: acc8 must be 16-byte aligned (and C a multiple of 4)");
    return bn_bwd_apply_impl(dy, x, y, saved_mean, saved_invstd, scale, red, capmi_deterministic() ? nullptr : acc8, dx, dx_accumulate, dres,
                             dres_accumulate, M, C, act, dtype, stream);
}

// ------------------------------------------------------------------ batch-norm backward straight from a max-pool's gradient
// The stem: conv -> batch_norm -> relu -> 3x3 / stride-2 max pool (the ResNet stem; MobileNetV2.py:88-121 + the pool of the
// build-defined ResNet encoders).  Its backward used to be three streaming launches on the step's serial tail --
// capmi_maxpool3x3s2_bwd writes the 103 MB gradient of the pool's input, capmi_bn_bwd_reduce and capmi_bn_bwd_apply read it
// back.  Here both batch-norm kernels GATHER that gradient from the pooled one (a quarter of the size) and the uint8 argmax
// map: a thread owns a 2 x 2 block of input pixels x VEC channels, loads the (up to) four windows that cover it once, and the
// pool's input gradient is never materialised.  dz = gathered dy * act'(y), rounded to T where the three-kernel path stores it.
template <typename T> struct PoolWin {
    static constexpr int VEC = Vec<T>::N;
    typedef typename std::conditional<VEC == 8, uint64_t, uint32_t>::type IdxT;
    Vec<T> dv[2][2];
    IdxT iv[2][2];
    bool ok[2][2];
    __device__ __forceinline__ void load(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, int b, int i, int j, int cc, int cpr, int Ho, int Wo) {
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int ho = i + di, wo = j + dj;
                ok[di][dj] = ho < Ho && wo < Wo;
                const int64_t o = ((((int64_t)b * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1)) * cpr + cc) * VEC;
                dv[di][dj] = vload<T>(dpool + o);
                iv[di][dj] = *reinterpret_cast<const IdxT*>(idx + o);
            }
    }
    // gradient of input pixel (2i + a, 2j + c): the windows whose argmax is this pixel (the arithmetic of maxpool_bwd_kernel)
    __device__ __forceinline__ void gather(int a, int c, float (&acc)[VEC]) const {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                if (di > a || dj > c || !ok[di][dj]) continue;
                const int r = a ? 2 - 2 * di : 1, q = c ? 2 - 2 * dj : 1;      // hi = 2 ho - 1 + r
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if ((int)((iv[di][dj] >> (8 * v)) & 0xff) == r * 3 + q) acc[v] += dv[di][dj].get(v);
            }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = to_f32(from_f32<T>(acc[v]));     // the value capmi_maxpool3x3s2_bwd would have stored
    }
};

// FROMX (capmi_bn_stat_apply_pool's backward): the activated tensor was never stored -- the activation's derivative comes from the
// conv output and the layer's saved coefficients (pool_act_out: the very expression the forward kernel rounded and compared)
template <typename T, int ACT, bool FROMX = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                                 const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* acc_rows, int B, int Hi, int Wi, int C, int Ho, int Wo, int rows_per_block,
                                                                 const float* __restrict__ coef_a = nullptr, const float* __restrict__ offset = nullptr) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float part[256 * VEC];
    const int cpr = C / VEC, Wb = (Wi + 1) >> 1, Hb = (Hi + 1) >> 1;
    const int tid = threadIdx.x, t = blockIdx.x * 256 + tid;
    const int j = t / cpr, cc = t - j * cpr;
    const bool active = j < Wb;
    ColLayout L;
    L.cpc = cpr; L.rp = 256 / cpr; L.rows_per_block = 0;
    float a1[VEC], a2[VEC], mu[VEC], is[VEC], ca[VEC], of[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        a1[v] = 0.f; a2[v] = 0.f; mu[v] = mean[cc * VEC + v]; is[v] = invstd[cc * VEC + v];
        ca[v] = FROMX ? coef_a[cc * VEC + v] : 0.f;
        of[v] = FROMX ? offset[cc * VEC + v] : 0.f;
    }
    const int r_end = min(B * Hb, (int)(blockIdx.y + 1) * rows_per_block);
    if (active)
        for (int r = blockIdx.y * rows_per_block; r < r_end; ++r) {
            const int b = r / Hb, i = r - b * Hb;
            PoolWin<T> w;
            w.load(dpool, idx, b, i, j, cc, cpr, Ho, Wo);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int hi = 2 * i + a, wi = 2 * j + c;
                    if (hi >= Hi || wi >= Wi) continue;
                    const int64_t o = ((((int64_t)b * Hi + hi) * Wi + wi) * cpr + cc) * VEC;
                    Vec<T> xv = vload<T>(x + o), yv;
                    if constexpr (!FROMX) yv = vload<T>(y + o);
                    float g[VEC];
                    w.gather(a, c, g);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float yo = FROMX ? pool_act_out<T, ACT>(xv.get(v), ca[v], mu[v], of[v]) : yv.get(v);
                        const float dz = g[v] * act_grad_from_out(yo, ACT);
                        a1[v] += dz;
                        a2[v] += dz * (xv.get(v) - mu[v]) * is[v];
                    }
                }
        }
    block_col_reduce<VEC>(part, a1, tid % cpr, tid / cpr, L, true);
    block_col_reduce<VEC>(part, a2, tid % cpr, tid / cpr, L, true);
    // block totals -> one f32 atomic per column into the accumulator row of this block (capmi_bn_bwd_reduce_spread's layout)
    __syncthreads();
    if (tid / cpr == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { part[(tid % cpr) * VEC + v] = a1[v]; part[C + (tid % cpr) * VEC + v] = a2[v]; }
    }
    __syncthreads();
    float* row = acc_rows + (int64_t)((blockIdx.y * gridDim.x + blockIdx.x) & (BN_SPREAD_ROWS - 1)) * 2 * C;
    for (int e = tid; e < 2 * C; e += 256) __hip_atomic_fetch_add(row + e, part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup = rows_per_block rows of 2 x 2 blocks: the per-channel constants (16-byte loads, the accumulator rows among
// them) are formed once per thread behind the first row's loads, as in bn_bwd_apply_kernel.
template <typename T, int ACT, bool FROMX = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                                const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ scale, const float* __restrict__ acc_rows, float* red_out, T* dx,
                                                                int B, int Hi, int Wi, int C, int Ho, int Wo, int rows_per_block, float inv_m,
                                                                const float* __restrict__ coef_a = nullptr, const float* __restrict__ offset = nullptr) {
    constexpr int VEC = Vec<T>::N;
    const int cpr = C / VEC, Wb = (Wi + 1) >> 1, Hb = (Hi + 1) >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int j = t / cpr, cc = t - j * cpr;
    if (j >= Wb) return;
    f32x4 isv[VEC / 4], scv[VEC / 4], muv[VEC / 4], t0[BN_SPREAD_ROWS][VEC / 4], t1[BN_SPREAD_ROWS][VEC / 4];
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
        const int ch = cc * VEC + 4 * q;
        isv[q] = *reinterpret_cast<const f32x4*>(invstd + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) scv[q][e] = scale[ch + e];       // a view into the flat parameter vector: any 4-byte alignment
        muv[q] = *reinterpret_cast<const f32x4*>(mean + ch);
#pragma unroll
        for (int r = 0; r < BN_SPREAD_ROWS; ++r) {
            t0[r][q] = *reinterpret_cast<const f32x4*>(acc_rows + (int64_t)r * 2 * C + ch);
            t1[r][q] = *reinterpret_cast<const f32x4*>(acc_rows + (int64_t)r * 2 * C + C + ch);
        }
    }
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(B * Hb, r_begin + rows_per_block);
    PoolWin<T> w;
    Vec<T> xv[2][2], yv[2][2];
    auto load_row = [&](int r) {
        const int b = r / Hb, i = r - b * Hb;
        w.load(dpool, idx, b, i, j, cc, cpr, Ho, Wo);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int hi = min(2 * i + a, Hi - 1), wi = min(2 * j + c, Wi - 1);      // clamped: the store below is guarded
                const int64_t o = ((((int64_t)b * Hi + hi) * Wi + wi) * cpr + cc) * VEC;
                xv[a][c] = vload<T>(x + o);
                if constexpr (!FROMX) yv[a][c] = vload<T>(y + o);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    float k1[VEC], c2[VEC], m0[VEC], mu[VEC], ca[VEC], of[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        ca[v] = FROMX ? coef_a[cc * VEC + v] : 0.f;
        of[v] = FROMX ? offset[cc * VEC + v] : 0.f;
    }
    auto store_row = [&](int r) {
        const int b = r / Hb, i = r - b * Hb;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int hi = 2 * i + a, wi = 2 * j + c;
                if (hi >= Hi || wi >= Wi) continue;
                const int64_t o = ((((int64_t)b * Hi + hi) * Wi + wi) * cpr + cc) * VEC;
                Vec<T> ov;
                float g[VEC];
                w.gather(a, c, g);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float yo = FROMX ? pool_act_out<T, ACT>(xv[a][c].get(v), ca[v], muv[v / 4][v % 4], of[v]) : yv[a][c].get(v);
                    const float dz = g[v] * act_grad_from_out(yo, ACT);
                    ov.set(v, k1[v] * ((dz - m0[v]) - (xv[a][c].get(v) - mu[v]) * c2[v]));
                }
                vstore<T>(dx + o, ov);
            }
    };
    if (r_begin < r_end) load_row(r_begin);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int q = 0; q < BN_SPREAD_ROWS; ++q) { r0 += t0[q][v / 4][v % 4]; r1 += t1[q][v / 4][v % 4]; }
        const float is_ = isv[v / 4][v % 4];
        mu[v] = muv[v / 4][v % 4];
        k1[v] = scv[v / 4][v % 4] * is_;
        c2[v] = is_ * r1 * inv_m;
        m0[v] = r0 * inv_m;
        if (blockIdx.y == 0 && j == 0) { red_out[cc * VEC + v] += r0; red_out[C + cc * VEC + v] += r1; }      // one writer per channel
    }
    for (int r = r_begin; r < r_end; ++r) {
        if (r != r_begin) load_row(r);
        store_row(r);
    }
}

extern "C" int capmi_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo, int dtype, void* stream);

/* capmi_maxpool3x3s2_bwd + capmi_bn_bwd_reduce_spread / capmi_bn_bwd_apply_spread of the layer that feeds the pool, without
 * materialising the pool's input gradient (see above).  dpool [B,Ho,Wo,C]: gradient of the pool's output; idx: the forward
 * pass's argmax map; x / y: the layer's conv output and activated output [B,Hi,Wi,C]; acc: the layer's accumulator rows
 * (zeroed once per step).  dy_scratch [B,Hi,Wi,C] is written only in deterministic mode, where the pair IS the three-launch
 * path (capmi_maxpool3x3s2_bwd, then the two-stage reduction through ws / red). */
static int bn_bwd_reduce_pool_impl(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                                   const float* saved_mean, const float* saved_invstd, float* ws, float* red, float* acc, void* dy_scratch, int B, int Hi,
                                   int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream) {
    CAPMI_CHECK(dpool && idx && x && y && saved_mean && saved_invstd && acc && dy_scratch, "capmi_bn_bwd_reduce_pool: null pointer");
    CAPMI_CHECK(act == CAPMI_ACT_RELU || act == CAPMI_ACT_RELU6, "capmi_bn_bwd_reduce_pool: the pooled layer's activation must be relu or relu6");
    const int M = B * Hi * Wi;
    if (capmi_deterministic()) {
        if (capmi_maxpool3x3s2_bwd(dpool, idx, dy_scratch, B, Hi, Wi, C, Ho, Wo, dtype, stream)) return 1;
        return capmi_bn_bwd_reduce(dy_scratch, x, y, saved_mean, saved_invstd, ws, red, M, C, act, dtype, stream);
    }
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_reduce_pool", {
        constexpr int VEC = Vec<T>::N;
        CAPMI_CHECK(C % VEC == 0 && 256 % (C / VEC) == 0 && 2 * C <= 256 * VEC, "capmi_bn_bwd_reduce_pool: C=%d unsupported (C / %d must divide 256)", C, VEC);
        const int cpr = C / VEC, Wb = (Wi + 1) / 2, Hb = (Hi + 1) / 2;
        const int gx = cdiv(Wb * cpr, 256);
        static const int wgs = getenv("CAPMI_POOL_WGS") ? atoi(getenv("CAPMI_POOL_WGS")) : 1024;      // experiment knob
        int rpb = cdiv((int64_t)B * Hb * gx, wgs);         // ~4 workgroups per CU (256 adds per accumulator address)
        if (rpb < 1) rpb = 1;
        const dim3 grid(gx, cdiv(B * Hb, rpb));
        if (coef_a) {       // the activated tensor was never written (capmi_bn_stat_apply_pool): its sign comes from the conv output
            if (act == CAPMI_ACT_RELU) hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T, CAPMI_ACT_RELU, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, acc, B, Hi, Wi, C, Ho, Wo, rpb, coef_a, offset);
            else hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T, CAPMI_ACT_RELU6, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, acc, B, Hi, Wi, C, Ho, Wo, rpb, coef_a, offset);
        } else if (act == CAPMI_ACT_RELU) hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T, CAPMI_ACT_RELU>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, acc, B, Hi, Wi, C, Ho, Wo, rpb);
        else hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T, CAPMI_ACT_RELU6>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, acc, B, Hi, Wi, C, Ho, Wo, rpb);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_reduce_pool");
    return 0;
}
extern "C" int capmi_bn_bwd_reduce_pool(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* saved_mean,
                                        const float* saved_invstd, float* ws, float* red, float* acc, void* dy_scratch, int B, int Hi, int Wi, int C,
                                        int Ho, int Wo, int act, int dtype, void* stream) {
    return bn_bwd_reduce_pool_impl(dpool, idx, x, y, nullptr, nullptr, saved_mean, saved_invstd, ws, red, acc, dy_scratch, B, Hi, Wi, C, Ho, Wo, act, dtype, stream);
}
// The pair behind capmi_bn_stat_apply_pool: y is the scratch tensor that entry point fills in deterministic mode only (read here in
// that mode only); outside it the activation's derivative is formed from x, coef_a, saved_mean and offset (capmi.h).
extern "C" int capmi_bn_bwd_reduce_pool_x(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                                          const float* saved_mean, const float* saved_invstd, float* ws, float* red, float* acc, void* dy_scratch, int B,
                                          int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream) {
    CAPMI_CHECK(coef_a && offset, "capmi_bn_bwd_reduce_pool_x: null pointer");
    return bn_bwd_reduce_pool_impl(dpool, idx, x, y, coef_a, offset, saved_mean, saved_invstd, ws, red, acc, dy_scratch, B, Hi, Wi, C, Ho, Wo, act, dtype, stream);
}
static int bn_bwd_apply_pool_impl(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                                  const float* saved_mean, const float* saved_invstd, const float* scale, float* red, const float* acc, const void* dy_scratch,
                                  void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream) {
    CAPMI_CHECK(dpool && idx && x && y && saved_mean && saved_invstd && scale && red && acc && dy_scratch && dx, "capmi_bn_bwd_apply_pool: null pointer");
    CAPMI_CHECK(act == CAPMI_ACT_RELU || act == CAPMI_ACT_RELU6, "capmi_bn_bwd_apply_pool: the pooled layer's activation must be relu or relu6");
    const int M = B * Hi * Wi;
    if (capmi_deterministic())      // dy_scratch holds the pool's input gradient (capmi_bn_bwd_reduce_pool wrote it)
        return capmi_bn_bwd_apply(dy_scratch, x, y, saved_mean, saved_invstd, scale, red, dx, 0, nullptr, 0, M, C, act, dtype, stream);
    CAPMI_DISPATCH(dtype, "capmi_bn_bwd_apply_pool", {
        constexpr int VEC = Vec<T>::N;
        CAPMI_CHECK(C % VEC == 0 && 256 % (C / VEC) == 0, "capmi_bn_bwd_apply_pool: C=%d unsupported", C);
        CAPMI_CHECK(((uintptr_t)acc | (uintptr_t)saved_mean | (uintptr_t)saved_invstd) % 16 == 0, "capmi_bn_bwd_apply_pool: acc8 / saved_mean / saved_invstd must be 16-byte aligned");
        const int cpr = C / VEC, Wb = (Wi + 1) / 2, Hb = (Hi + 1) / 2;
        const int gx = cdiv(Wb * cpr, 256);
        int rpb = cdiv((int64_t)B * Hb * gx, 4096);          // ~16 workgroups per CU, a few rows each
        if (rpb < 1) rpb = 1;
        const dim3 grid(gx, cdiv(B * Hb, rpb));
        if (coef_a) {
            if (act == CAPMI_ACT_RELU) hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T, CAPMI_ACT_RELU, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, scale, acc, red, (T*)dx, B, Hi, Wi, C, Ho, Wo, rpb, 1.f / (float)M, coef_a, offset);
            else hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T, CAPMI_ACT_RELU6, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, scale, acc, red, (T*)dx, B, Hi, Wi, C, Ho, Wo, rpb, 1.f / (float)M, coef_a, offset);
        } else if (act == CAPMI_ACT_RELU) hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T, CAPMI_ACT_RELU>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, scale, acc, red, (T*)dx, B, Hi, Wi, C, Ho, Wo, rpb, 1.f / (float)M);
        else hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T, CAPMI_ACT_RELU6>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dpool, idx, (const T*)x, (const T*)y, saved_mean, saved_invstd, scale, acc, red, (T*)dx, B, Hi, Wi, C, Ho, Wo, rpb, 1.f / (float)M);
    });
    CAPMI_LAUNCH_CHECK("capmi_bn_bwd_apply_pool");
    return 0;
}
extern "C" int capmi_bn_bwd_apply_pool(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* saved_mean,
                                       const float* saved_invstd, const float* scale, float* red, const float* acc, const void* dy_scratch, void* dx,
                                       int B, int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream) {
    return bn_bwd_apply_pool_impl(dpool, idx, x, y, nullptr, nullptr, saved_mean, saved_invstd, scale, red, acc, dy_scratch, dx, B, Hi, Wi, C, Ho, Wo, act, dtype, stream);
}
extern "C" int capmi_bn_bwd_apply_pool_x(const void* dpool, const uint8_t* idx, const void* x, const void* y, const float* coef_a, const float* offset,
                                         const float* saved_mean, const float* saved_invstd, const float* scale, float* red, const float* acc,
                                         const void* dy_scratch, void* dx, int B, int Hi, int Wi, int C, int Ho, int Wo, int act, int dtype, void* stream) {
    CAPMI_CHECK(coef_a && offset, "capmi_bn_bwd_apply_pool_x: null pointer");
    return bn_bwd_apply_pool_impl(dpool, idx, x, y, coef_a, offset, saved_mean, saved_invstd, scale, red, acc, dy_scratch, dx, B, Hi, Wi, C, Ho, Wo, act, dtype, stream);
}
