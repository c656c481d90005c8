// Microbenchmark: write bandwidth of the GEMM-epilogue store shape.  Each workgroup writes a
// [128 rows][seg bytes] tile into rows of `row_bytes`; seg == row_bytes is the fully contiguous case.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void wr(char* out, int rows, int row_bytes, int seg, int tiles_n, int mode) {
    int tile = blockIdx.x;
    int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * seg;
    int cpr = seg / 16;                       // 16-byte chunks per tile row
    u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (int idx = threadIdx.x; idx < 128 * cpr; idx += 256) {
        int r, c;
        if (mode == 0) { r = idx / cpr; c = idx % cpr; }          // row-major over the tile (GEMM epilogue)
        else { c = idx / 128; r = idx % 128; }                     // column-major: lanes walk down rows
        if (m0 + r < rows) {
            char* p = out + (int64_t)(m0 + r) * row_bytes + n0 + c * 16;
            if (mode == 2) __builtin_nontemporal_store(v, (u32x4*)p);
            else *(u32x4*)p = v;
        }
    }
}

int main() {
    const int rows = 200704;
    char* buf;
    hipMalloc(&buf, (size_t)rows * 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int row_bytes_list[] = {512, 1024, 2048, 4096};
    for (int rb : row_bytes_list)
        for (int seg = 128; seg <= rb; seg *= 2)
            for (int mode = 0; mode < 3; ++mode) {
                int tiles_n = rb / seg, tiles = (rows / 128) * tiles_n;
                for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(wr, dim3(tiles), dim3(256), 0, 0, buf, rows, rb, seg, tiles_n, mode == 1 ? 1 : 0 + (mode == 2 ? 2 : 0));
                hipEventRecord(a);
                for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(wr, dim3(tiles), dim3(256), 0, 0, buf, rows, rb, seg, tiles_n, mode == 1 ? 1 : (mode == 2 ? 2 : 0));
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                double gb = (double)rows * rb / 1e9;
                printf("row %4d B  seg %4d B  mode %d  %8.1f us  %7.1f GB/s\n", rb, seg, mode, ms / 10 * 1e3, gb / (ms / 10 * 1e-3));
            }
    return 0;
}
