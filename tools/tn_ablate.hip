// Ablation of the LDS-DMA weight-gradient kernel on one conv shape: which of its three streams (DMA into LDS,
// transposing LDS reads, MFMA) sets the time.  Includes the library source so the product kernel itself is measured.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imyimagecaptioningmodel_amd/csrc tools/tn_ablate.hip \
//        myimagecaptioningmodel_amd/csrc/capi.hip -o tools/tn_ablate ; run: tools/tn_ablate [Hi Cin Cout k]
#include "../myimagecaptioningmodel_amd/csrc/igemm.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

template <int ABL, int NST = 2, int KS = 1, bool PIPE = false> static float run(WGradArgs a, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((igemm_tn_glds_kernel<ABL, NST, KS, PIPE>), dim3(grid), dim3(512), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((igemm_tn_glds_kernel<ABL, NST, KS, PIPE>), dim3(grid), dim3(512), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int B = 64;
    const int Hi = argc > 1 ? atoi(argv[1]) : 28, Cin = argc > 2 ? atoi(argv[2]) : 128, Cout = argc > 3 ? atoi(argv[3]) : 128;
    const int k = argc > 4 ? atoi(argv[4]) : 3;
    capmi_conv_geom g;
    memset(&g, 0, sizeof g);
    g.B = B; g.Hi = Hi; g.Wi = Hi; g.Cin = Cin; g.ldx = Cin; g.kh = g.kw = k; g.sd = 1; g.pad = k / 2; g.up = 1; g.Ho = Hi; g.Wo = Hi;
    WGradArgs a;
    memset(&a, 0, sizeof a);
    a.M = B * Hi * Hi; a.N = Cout; a.K = k * k * Cin; a.ldy = Cout; a.lddw = a.K; a.g = g;
    a.fd_hw = fast_div(g.Ho * g.Wo); a.fd_w = fast_div(g.Wo);
    a.linear = k == 1;
    void *x, *dy; float *dw, *slab;
    hipMalloc(&x, (size_t)a.M * Cin * 2 + 4096); hipMalloc(&dy, (size_t)a.M * Cout * 2 + 4096);
    {   // small integers in bf16: products and sums are exact in f32, so both wave layouts must agree bit for bit
        size_t nx = (size_t)a.M * Cin, ny = (size_t)a.M * Cout;
        unsigned short *hx = (unsigned short*)malloc(nx * 2), *hy = (unsigned short*)malloc(ny * 2);
        const unsigned short vals[4] = {0x0000, 0x3f80, 0xbf80, 0x4000};       // 0, 1, -1, 2
        unsigned r = 12345;
        for (size_t i = 0; i < nx; ++i) { r = r * 1664525u + 1013904223u; hx[i] = vals[(r >> 20) & 3]; }
        for (size_t i = 0; i < ny; ++i) { r = r * 1664525u + 1013904223u; hy[i] = vals[(r >> 20) & 3]; }
        hipMemcpy(x, hx, nx * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hy, ny * 2, hipMemcpyHostToDevice);
        free(hx); free(hy);
    }
    hipMalloc(&dw, (size_t)a.N * a.K * 4);
    int per;
    const int tiles = cdiv(a.N, 128) * cdiv(a.K, 128);
    const int splits = tn_splits(a.M, a.N, a.K, 128, 128, &per);
    a.m_per_split = per; a.splits = splits; a.Np = cdiv(a.N, 128) * 128; a.Kp = cdiv(a.K, 128) * 128;
    hipMalloc(&slab, (size_t)splits * a.Np * a.Kp * 4);
    a.x = x; a.dy = dy; a.dw = dw; a.slab = splits > 1 ? slab : nullptr;
    const int grid = tiles * splits;
    const double staged = (double)grid * ((per + 63) / 64) * 32768.0, flop = 2.0 * a.M * a.N * a.K;
    printf("%dx%d %d->%d k%d: M=%d tiles=%d splits=%d rows/split=%d grid=%d, staged %.0f MB, %.2f GFLOP\n", Hi, Hi, Cin, Cout, k, a.M, tiles,
           splits, per, grid, staged / 1e6, flop / 1e9);
    const float t0 = run<0>(a, grid, 20), t1 = run<1>(a, grid, 20), t2 = run<2>(a, grid, 20), t4 = run<4>(a, grid, 20), t6 = run<6>(a, grid, 20);
    printf("full kernel          %7.1f us  %6.0f TFLOP/s  staged %5.1f GB/s per CU\n", t0, flop / t0 / 1e6, staged / t0 / 1e3 / 256);
    printf("no MFMA              %7.1f us\n", t1);
    printf("DMA only             %7.1f us  staged %5.1f GB/s per CU\n", t2, staged / t2 / 1e3 / 256);
    printf("no DMA (reads+MFMA)  %7.1f us  %6.0f TFLOP/s\n", t4, flop / t4 / 1e6);
    printf("barriers + epilogue  %7.1f us\n", t6);
    const float u0 = run<0, 3>(a, grid, 20), u2 = run<2, 3>(a, grid, 20);
    printf("3-stage ring: full   %7.1f us  %6.0f TFLOP/s;  DMA only %7.1f us  staged %5.1f GB/s per CU\n", u0, flop / u0 / 1e6, u2, staged / u2 / 1e3 / 256);
    {
        const size_t ne = (size_t)splits * a.Np * a.Kp;
        float *ha = (float*)malloc(ne * 4), *hb = (float*)malloc(ne * 4);
        hipMemset(slab, 0, ne * 4);
        hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 2, 1>), dim3(grid), dim3(512), 0, 0, a);
        hipMemcpy(ha, slab, ne * 4, hipMemcpyDeviceToHost);
        for (int nst = 2; nst <= 3; ++nst) {
            hipMemset(slab, 0, ne * 4);
            if (nst == 2) hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 2, 2>), dim3(grid), dim3(512), 0, 0, a);
            else hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 3, 2>), dim3(grid), dim3(512), 0, 0, a);
            hipMemcpy(hb, slab, ne * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; double sum = 0;
            for (size_t i = 0; i < ne; ++i) { bad += ha[i] != hb[i]; sum += fabs((double)ha[i]); }
            printf("k-step groups, %d stages: %zu of %zu slab entries differ from the 2 x 4 layout (mean |entry| %.2f)\n", nst, bad, ne, sum / ne);
        }
        free(ha); free(hb);
    }
    {
        const size_t ne = (size_t)splits * a.Np * a.Kp;
        float *ha = (float*)malloc(ne * 4), *hb = (float*)malloc(ne * 4);
        hipMemset(slab, 0, ne * 4);
        hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 2, 1>), dim3(grid), dim3(512), 0, 0, a);
        hipMemcpy(ha, slab, ne * 4, hipMemcpyDeviceToHost);
        for (int v = 0; v < 4; ++v) {
            hipMemset(slab, 0, ne * 4);
            if (v == 0) hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 2, 1, true>), dim3(grid), dim3(512), 0, 0, a);
            if (v == 1) hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 3, 1, true>), dim3(grid), dim3(512), 0, 0, a);
            if (v == 2) hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 2, 2, true>), dim3(grid), dim3(512), 0, 0, a);
            if (v == 3) hipLaunchKernelGGL((igemm_tn_glds_kernel<0, 3, 2, true>), dim3(grid), dim3(512), 0, 0, a);
            hipMemcpy(hb, slab, ne * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < ne; ++i) bad += ha[i] != hb[i];
            printf("pipelined variant %d: %zu of %zu slab entries differ\n", v, bad, ne);
        }
        free(ha); free(hb);
    }
    const float p0 = run<0, 2, 1, true>(a, grid, 20), p1 = run<0, 3, 1, true>(a, grid, 20), p2 = run<0, 2, 2, true>(a, grid, 20), p3 = run<0, 3, 2, true>(a, grid, 20);
    printf("pipelined: 2x4 waves, 2 stages %6.1f us (%4.0f TFLOP/s) | 2x4, 3 stages %6.1f (%4.0f) | k-step groups, 2 stages %6.1f (%4.0f) | k-step groups, 3 stages %6.1f (%4.0f)\n",
           p0, flop / p0 / 1e6, p1, flop / p1 / 1e6, p2, flop / p2 / 1e6, p3, flop / p3 / 1e6);
    const float w0 = run<0, 2, 2>(a, grid, 20), w4 = run<4, 2, 2>(a, grid, 20), w1 = run<1, 2, 2>(a, grid, 20);
    printf("k-step groups, 2 stages: full %7.1f us  %6.0f TFLOP/s;  no DMA %7.1f us;  no MFMA %7.1f us\n", w0, flop / w0 / 1e6, w4, w1);
    const float y0 = run<0, 3, 2>(a, grid, 20), y4 = run<4, 3, 2>(a, grid, 20);
    printf("k-step groups, 3 stages: full %7.1f us  %6.0f TFLOP/s;  no DMA %7.1f us\n", y0, flop / y0 / 1e6, y4);
    const float v0 = run<0, 4>(a, grid, 20), v2 = run<2, 4>(a, grid, 20);
    printf("4-stage ring: full   %7.1f us  %6.0f TFLOP/s;  DMA only %7.1f us  staged %5.1f GB/s per CU\n", v0, flop / v0 / 1e6, v2, staged / v2 / 1e3 / 256);
    return 0;
}
