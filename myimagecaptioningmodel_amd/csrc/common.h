// Shared device helpers for the gfx950 kernels of libcapmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "capmi.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CAPMI_WAVE 64

extern "C" void capmi_set_error(const char* fmt, ...);

#define CAPMI_CHECK(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            capmi_set_error(__VA_ARGS__);     \
            return 1;                         \
        }                                     \
    } while (0)

#define CAPMI_LAUNCH_CHECK(name)                                                       \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            capmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

// dtype dispatch: DT(float) / DT(bf16)
#define CAPMI_DISPATCH(dtype, NAME, ...)                              \
    do {                                                              \
        if ((dtype) == CAPMI_F32) {                                   \
            typedef float T;                                          \
            __VA_ARGS__;                                              \
        } else if ((dtype) == CAPMI_BF16) {                           \
            typedef bf16 T;                                           \
            __VA_ARGS__;                                              \
        } else {                                                      \
            capmi_set_error("%s: bad dtype %d", NAME, (int)(dtype));  \
            return 1;                                                 \
        }                                                             \
    } while (0)

// ------------------------------------------------------------------ 16-byte vectors of T
template <typename T> struct Vec;            // VEC elements of T in 16 bytes
template <> struct Vec<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <typename T> __device__ __forceinline__ Vec<T> vzero() {
    Vec<T> r;
#pragma unroll
    for (int i = 0; i < Vec<T>::N; ++i) r.set(i, 0.f);
    return r;
}
template <typename T> __device__ __forceinline__ Vec<T> vload(const T* p) {
    Vec<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void vstore(T* p, const Vec<T>& x) {
    *reinterpret_cast<decltype(x.v)*>(p) = x.v;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }   // ocml: ~1 ulp, keeps f32 parity
// tanh for kernels whose result is rounded to bf16 anyway: 1 - 2/(1 + e^{2x}) on the hardware exp / rcp
// (absolute error ~1e-7; exact limits +-1).  The f32 instantiations keep the ocml function.
template <typename T> __device__ __forceinline__ float tanh_for(float x) { return tanhf(x); }
template <typename T> __device__ __forceinline__ float sigmoid_for(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float sigmoid_for<bf16>(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
template <> __device__ __forceinline__ float tanh_for<bf16>(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case CAPMI_ACT_RELU: return fmaxf(x, 0.f);
        case CAPMI_ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
        case CAPMI_ACT_TANH: return tanhf_(x);
        case CAPMI_ACT_SIGMOID: return sigmoidf_(x);
        default: return x;
    }
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case CAPMI_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case CAPMI_ACT_RELU6: return (y > 0.f && y < 6.f) ? 1.f : 0.f;
        case CAPMI_ACT_TANH: return 1.f - y * y;
        case CAPMI_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ------------------------------------------------------------------ LDS-only workgroup barrier
// __syncthreads() is a full workgroup fence: it also waits (s_waitcnt vmcnt(0)) until every global
// store the wave has issued is complete.  Where only LDS data is handed between waves, wait for the
// LDS counter alone so global stores stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Sum of x over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane.
// VALU only (v_permlane32_swap / v_permlane16_swap, gfx950): no LDS crossbar, no lgkmcnt wait.
__device__ __forceinline__ float row4_sum(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // a[0] = lower half everywhere, a[1] = upper half
    float y = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    unsigned w = __builtin_bit_cast(unsigned, y);
    auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);     // b[0] = even rows everywhere, b[1] = odd rows
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// ------------------------------------------------------------------ reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
    v = wave_max(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = smem[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
    return r;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Per-channel reductions use a (column-chunk, row) thread layout: thread -> (cc = tid % cpc,
// rr = tid / cpc), loops over rows rr, rr+RP, ...; partial sums meet in LDS (ds_add_f32), then
// one global f32 atomic per channel per workgroup.
struct ColLayout {
    int cpc;        // 16-byte chunks per row handled by this block (<= 256)
    int rp;         // rows per pass
    int rows_per_block;
};
static ColLayout col_layout(int M, int C, int vec, int* grid_x, int* grid_y, int max_cpc = 256) {
    ColLayout L;
    int chunks = C / vec;
    L.cpc = chunks < max_cpc ? chunks : max_cpc;
    *grid_y = cdiv(chunks, L.cpc);
    L.rp = 256 / L.cpc;
    int target_blocks = 2048 / *grid_y;
    if (target_blocks < 1) target_blocks = 1;
    int rpb = cdiv(M, target_blocks);
    rpb = cdiv(rpb, L.rp) * L.rp;
    if (rpb < L.rp * 4) rpb = L.rp * 4;
    L.rows_per_block = rpb;
    *grid_x = cdiv(M, rpb);
    return L;
}


// Sum the per-thread partials acc[VEC] of the (cc, rr) layout over rr.  `part` holds 256*VEC floats.
// After the call threads with rr == 0 hold the block totals of their channel chunk in acc[].
// Plain LDS stores + a strided re-read: no same-address atomic contention, fixed summation order.
template <int VEC>
__device__ __forceinline__ void block_col_reduce(float* part, float (&acc)[VEC], int cc, int rr, const ColLayout& L, bool active) {
    __syncthreads();
    if (active) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[(rr * L.cpc + cc) * VEC + v] = acc[v];
    }
    __syncthreads();
    if (active && rr == 0) {
        for (int r = 1; r < L.rp; ++r) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += part[(r * L.cpc + cc) * VEC + v];
        }
    }
}

// rows-per-thread-heavy variant of col_layout for the streaming (elementwise / reduce) kernels
static ColLayout ew_layout(int M, int C, int vec, int* grid_x, int* grid_y, int max_cpc = 256) {
    ColLayout L;
    int chunks = C / vec;
    L.cpc = chunks < max_cpc ? chunks : max_cpc;
    *grid_y = cdiv(chunks, L.cpc);
    L.rp = 256 / L.cpc;
    int target_blocks = 1024 / *grid_y;                 // ~4 workgroups per CU
    if (target_blocks < 1) target_blocks = 1;
    int rpb = cdiv(M, target_blocks);
    rpb = cdiv(rpb, L.rp) * L.rp;
    if (rpb < L.rp * 8) rpb = L.rp * 8;                 // >= 8 rows per thread amortise the setup
    L.rows_per_block = rpb;
    *grid_x = cdiv(M, rpb);
    return L;
}

// grid for grid-stride elementwise kernels of n 16-byte chunks
static inline int ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
