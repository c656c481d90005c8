// Microbenchmark of the two denominators bench.py's roofline uses, measured on the box itself (SURVEY.md 8d asks
// for the vendor peaks to be backed by a measurement): HBM streaming bandwidth (read, copy, read+read->write, the
// shapes of the BN kernels) over buffers far larger than MALL + L2, and the issue-bound bf16 MFMA rate
// (register-only v_mfma_f32_16x16x32_bf16 chains, 4 independent accumulators per wave, 8 waves per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o tools/peaks ; run: tools/peaks
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ a, size_t n, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= a[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}
__global__ __launch_bounds__(256) void cp(const u32x4* __restrict__ a, u32x4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ __launch_bounds__(256) void rrw(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i] ^ b[i];
}
__global__ __launch_bounds__(256) void mfma(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    f32x4 s = c0 + c1 + c2 + c3;
    if (s.x == 12345.f) out[threadIdx.x] = s.y + s.z + s.w;
}

template <typename F> static float time_ms(F launch, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;          // 2 GiB per buffer >> 256 MB MALL
    u32x4 *a, *b, *o;
    uint32_t* sink;
    if (hipMalloc(&a, bytes) || hipMalloc(&b, bytes) || hipMalloc(&o, bytes) || hipMalloc(&sink, 4096)) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(o, 0, bytes);
    int grids[] = {2048, 8192, 32768};
    for (int g : grids) {
        float t1 = time_ms([&] { hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, a, n, sink); }, 5);
        float t2 = time_ms([&] { hipLaunchKernelGGL(cp, dim3(g), dim3(256), 0, 0, a, o, n); }, 5);
        float t3 = time_ms([&] { hipLaunchKernelGGL(rrw, dim3(g), dim3(256), 0, 0, a, b, o, n); }, 5);
        printf("hbm grid %6d: read %.0f GB/s   copy %.0f GB/s (read+write)   2 reads + 1 write %.0f GB/s\n", g, bytes / t1 / 1e6,
               2.0 * bytes / t2 / 1e6, 3.0 * bytes / t3 / 1e6);
    }
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        int g = 256 * wg_per_cu;
        float t = time_ms([&] { hipLaunchKernelGGL(mfma, dim3(g), dim3(256), 0, 0, (float*)sink, iters); }, 3);
        double flop = (double)g * 4 /*waves*/ * iters * 4 /*chains*/ * 2.0 * 16 * 16 * 32;
        printf("mfma bf16 16x16x32, %d workgroup(s) of 4 waves per CU: %.0f TFLOP/s\n", wg_per_cu, flop / t / 1e9);
    }
    return 0;
}
