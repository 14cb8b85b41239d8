// What the band tiler can hope for: HBM write rates of its store patterns, nothing else in the kernel.
//   hipcc --offload-arch=gfx950 -O3 write_rate.hip -o build/write_rate && build/write_rate
//   (a) grid-stride 16-byte stores over one 2.4 GB buffer            -- the ceiling
//   (b) wave-per-row: a wave writes 8 KB of float64 + 4 KB of float32 of "its" row (rows 8 apart per
//       workgroup, as stage_tile_kernel does), 1 KB per store instruction
//   (c) as (b), float32 rows only
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill16(d2* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const d2 v = {1.0, 2.0};
    for (; i < n; i += stride) p[i] = v;
}

template <bool F64, bool F32>
__global__ __launch_bounds__(512) void rows(double* b64, float* b32, int n_rows, int ld)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const d2 v = {1.0, 2.0};
    const f4 w = {1.f, 2.f, 3.f, 4.f};
    for (int g = blockIdx.x; g * 128 < n_rows; g += gridDim.x)
        for (int r = g * 128 + wv; r < min(n_rows, (g + 1) * 128); r += 8) {
            if (F64)
                for (int x = 2 * lane; x < ld; x += 128) *reinterpret_cast<d2*>(b64 + (size_t)r * ld + x) = v;
            if (F32)
                for (int x = 4 * lane; x < ld; x += 256) *reinterpret_cast<f4*>(b32 + (size_t)r * ld + x) = w;
        }
}

int main()
{
    const int n_rows = 200000, ld = 1088;
    double* b64;
    float* b32;
    hipMalloc(&b64, (size_t)n_rows * ld * 8);
    hipMalloc(&b32, (size_t)n_rows * ld * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int k = 0; k < 3; ++k) launch();
        hipEventRecord(e0, 0);
        for (int k = 0; k < 10; ++k) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.3f ms  %.0f GB/s\n", name, ms / 10, bytes / (ms / 10) / 1e6);
    };
    const double b8 = (double)n_rows * ld * 8, b4 = (double)n_rows * ld * 4;
    run("grid-stride 16 B stores, 1.74 GB", b8, [&] { hipLaunchKernelGGL(fill16, dim3(256 * 16), dim3(256), 0, 0, (d2*)b64, (size_t)n_rows * ld / 2); });
    for (int per_cu : {2, 3, 4, 8}) {
        char nm[64];
        snprintf(nm, 64, "rows f64+f32, %d workgroups / CU", per_cu);
        run(nm, b8 + b4, [&] { hipLaunchKernelGGL((rows<true, true>), dim3(256 * per_cu), dim3(512), 0, 0, b64, b32, n_rows, ld); });
    }
    run("rows f64 only, 4 / CU", b8, [&] { hipLaunchKernelGGL((rows<true, false>), dim3(1024), dim3(512), 0, 0, b64, b32, n_rows, ld); });
    run("rows f32 only, 4 / CU", b4, [&] { hipLaunchKernelGGL((rows<false, true>), dim3(1024), dim3(512), 0, 0, b64, b32, n_rows, ld); });
    return 0;
}
