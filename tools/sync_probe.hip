// What does a lane hand-off cost the RECORDING stream?  A chain of short kernels on one stream with, between consecutive
// kernels: nothing / hipEventRecord (no timing, no system fence: the plan runner's events) / hipStreamWriteValue32.
// Prints microseconds per link.  hipcc --offload-arch=gfx950 -O3 tools/sync_probe.hip -o tools/sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int n) {
    float x = p[threadIdx.x];
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 24));
    uint32_t* flag; CK(hipMalloc(&flag, 256));
    CK(hipMemset(flag, 0, 256));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int links = 200, reps = 5;
    std::vector<hipEvent_t> ev(links);
    for (auto& evi : ev) CK(hipEventCreateWithFlags(&evi, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int mode = 0; mode < 6; ++mode) {
        double best = 1e9;
        for (int r = 0; r < reps; ++r) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < links; ++i) {
                if (mode >= 4) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), 0, a, nullptr, ev[i], 0, buf, 2000);      // the event rides on the dispatch packet's completion signal
                else hipLaunchKernelGGL(spin, dim3(512), dim3(256), 0, a, buf, 2000);
                if (mode == 5) { CK(hipStreamWaitEvent(b, ev[i], 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, buf + (1 << 20), 500); }
                if (mode == 1) CK(hipEventRecord(ev[i], a));
                if (mode == 2) CK(hipStreamWriteValue32(a, flag, (uint32_t)(i + 1), 0));
                if (mode == 3) { CK(hipEventRecord(ev[i], a)); CK(hipStreamWaitEvent(b, ev[i], 0)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, buf + (1 << 20), 500); }
            }
            CK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / links;
            if (us < best) best = us;
        }
        const char* names[] = {"kernel chain alone", "+ hipEventRecord per link", "+ hipStreamWriteValue32 per link", "+ record, other stream waits and runs a small kernel",
                               "hipExtLaunchKernelGGL with a stop event per link", "stop event per link, other stream waits and runs a small kernel"};
        printf("%-55s %.2f us per link\n", names[mode], best);
    }
    return 0;
}
