// Chan merge of batch-norm statistic parts (mean, M2) in f64 and the finalize step: shared by bn_ops.hip (capmi_bn_finalize)
// and the convolution epilogue whose last-arriving workgroup finalizes (igemm.hip, capmi_igemm_nt_bnfin).  Same arithmetic
// and fold order wherever it runs: the two paths are bit-identical.
#pragma once
#include "common.h"

#define CAPMI_BN_MERGE_GROUPS 32     // merged groups (64 for C <= 128: few channel blocks, so more row groups); ws has room for 64 extra parts (capmi.h)

__device__ __forceinline__ void chan_fold(double& n, double& mean, double& m2, double nb, double mb, double qb) {
    if (nb <= 0.0) return;
    const double tot = n + nb, d = mb - mean;
    mean += d * (nb / tot);
    m2 += qb + d * d * (n * nb / tot);
    n = tot;
}
template <bool SC1 = false>       // SC1: the parts were written by other workgroups of THIS launch (write-through): load past L1 / the local L2
__device__ __forceinline__ void merge_parts(const float* __restrict__ ws, int part_rows, int M, int C, int c, int p0, int p1,
                                            double (*red)[64], double* mean_out, double* m2_out) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    if (c < C) {
        int p = p0 + ty;
        for (; p + 12 < p1; p += 16) {
            float mu[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* w = ws + ((int64_t)(p + 4 * u) * C + c) * 2;
                if (SC1) {
                    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mu[u] = __builtin_bit_cast(float, (unsigned)v);
                    q[u] = __builtin_bit_cast(float, (unsigned)(v >> 32));
                } else {
                    mu[u] = w[0];
                    q[u] = w[1];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) chan_fold(n, mean, m2, (double)min(part_rows, M - (p + 4 * u) * part_rows), (double)mu[u], (double)q[u]);
        }
        for (; p < p1; p += 4) {
            const float* w = ws + ((int64_t)p * C + c) * 2;
            float mu1, q1;
            if (SC1) {
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mu1 = __builtin_bit_cast(float, (unsigned)v);
                q1 = __builtin_bit_cast(float, (unsigned)(v >> 32));
            } else {
                mu1 = w[0];
                q1 = w[1];
            }
            chan_fold(n, mean, m2, (double)min(part_rows, M - p * part_rows), (double)mu1, (double)q1);
        }
    }
    __syncthreads();                 // a second call may follow a first one's reads of red
    red[ty][tx] = mean;
    red[4 + ty][tx] = m2;
    red[8 + ty][tx] = n;
    __syncthreads();
    double tn = 0.0, tmean = 0.0, tm2 = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) chan_fold(tn, tmean, tm2, red[8 + g][tx], red[g][tx], red[4 + g][tx]);
    *mean_out = tmean;
    *m2_out = tm2;
}

template <bool SC1>
__device__ __forceinline__ void bn_finalize_body(const float* __restrict__ ws, int part_rows, int M, int C, int cblock, const float* scale,
                                                 float* run_mean, float* run_var, float momentum, float eps,
                                                 float* saved_mean, float* saved_invstd, float* coef_a, int update_running, double (*red)[64]) {
    const int c = cblock * 64 + (threadIdx.x & 63);
    const int nparts = (M + part_rows - 1) / part_rows;
    // the per-channel scalars are fetched before the merge, not after it (one memory latency less in a kernel that is
    // nothing but latency)
    const bool writer = (threadIdx.x >> 6) == 0 && c < C;
    const float sc = writer ? scale[c] : 0.f;
    const float rm = writer && update_running ? run_mean[c] : 0.f, rv = writer && update_running ? run_var[c] : 0.f;
    double mean, m2;
    merge_parts<SC1>(ws, part_rows, M, C, c, 0, nparts, red, &mean, &m2);
    if (!writer) return;
    const double var = m2 / (double)M;      // biased
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    saved_mean[c] = (float)mean;
    saved_invstd[c] = invstd;
    coef_a[c] = sc * invstd;
    if (update_running) {
        run_mean[c] = rm * momentum + (float)mean * (1.f - momentum);
        run_var[c] = rv * momentum + (float)var * (1.f - momentum);
    }
}

