// Ablation of the LDS-DMA forward / data-gradient GEMM (igemm_nt_glds_*) on ResNet-50 layer shapes at batch 64: the
// library source is compiled four times with -DCAPMI_NT_ABL=0/1/2/4 (all / no MFMA / DMA only / no DMA) and each
// binary times the same capmi_igemm_nt calls, so the product dispatch (tile, k-groups, addressing mode) is what runs.
// Build + run: see tools/nt_ablate.sh
#include "../myimagecaptioningmodel_amd/csrc/igemm.hip"
#include <stdio.h>
#include <stdlib.h>

int main() {
    const int B = 64;
    const int shapes[][5] = {{56, 64, 256, 1, 0}, {56, 256, 64, 1, 0}, {56, 64, 64, 3, 1}, {56, 256, 128, 1, 0}, {28, 128, 128, 3, 1}, {28, 512, 128, 1, 0},
                             {28, 128, 512, 1, 0}, {14, 256, 256, 3, 1}, {14, 1024, 256, 1, 0}, {14, 256, 1024, 1, 0}, {7, 512, 512, 3, 1}, {7, 2048, 512, 1, 0}};
    void *x, *w, *y;
    hipMalloc(&x, (size_t)B * 56 * 56 * 256 * 2 + 4096); hipMalloc(&w, (size_t)4 << 20 << 2); hipMalloc(&y, (size_t)B * 56 * 56 * 256 * 2 + 4096);
    hipMemset(x, 0, (size_t)B * 56 * 56 * 256 * 2); hipMemset(w, 0, (size_t)4 << 20 << 2);
    float* stats;
    hipMalloc(&stats, (size_t)64 << 20);
    if (getenv("RANDOM")) {       // RANDOM=1: random bf16 operands instead of zeros (switching power, same addresses)
        size_t n = (size_t)B * 56 * 56 * 256;
        unsigned short* h = (unsigned short*)malloc(n * 2);
        unsigned r = 1;
        for (size_t i = 0; i < n; ++i) { r = r * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00 + ((r >> 16) & 0x3ff) + ((r >> 31) << 15)); }
        hipMemcpy(x, h, n * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, h, (size_t)8 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("ABL=%d\n", CAPMI_NT_ABL);
    for (auto& s : shapes) {
        const int H = s[0], Cin = s[1], Cout = s[2], k = s[3], pad = s[4];
        capmi_conv_geom g;
        memset(&g, 0, sizeof g);
        g.B = B; g.Hi = H; g.Wi = H; g.Cin = Cin; g.ldx = Cin; g.kh = g.kw = k; g.sd = 1; g.pad = pad; g.up = 1; g.Ho = H; g.Wo = H;
        const int K = k * k * Cin;
        float* st = getenv("STATS") ? stats : nullptr;       // STATS=1: with the fused batch-norm statistics epilogue (forward convs)
        auto call = [&]() { return capmi_igemm_nt(x, w, y, &g, Cout, K, Cout, nullptr, nullptr, 0, nullptr, 0, st, 0, 0, 0, CAPMI_BF16, nullptr); };
        for (int i = 0; i < 3; ++i) if (call()) { printf("call failed: %s\n", capmi_last_error()); return 1; }
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) call();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3, flop = 2.0 * B * H * H * Cout * K;
        printf("%2dx%-2d %4d->%-4d k%d  %7.1f us  %6.0f TFLOP/s\n", H, H, Cin, Cout, k, us, flop / us / 1e6);
    }
    return 0;
}
