// What an (almost) empty launch costs as a function of the workgroup's LDS request and register count:
// 512 workgroups of 256 threads (2 per CU), each returning at once.  tools/ubench/launch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int REGS>
__global__ __launch_bounds__(256) void k_empty(float* out, int n)
{
    extern __shared__ char smem[];
    float acc[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc[i] = out[(threadIdx.x + i) & 1023];
    if (n == 12345) {                      // never: keeps the registers and the LDS alive
        float s = 0;
#pragma unroll
        for (int i = 0; i < REGS; ++i) s += acc[i] * smem[i];
        out[threadIdx.x] = s;
    }
}
template <int REGS>
static void run(const char* name, size_t lds, int grid)
{
    float* d;
    hipMalloc(&d, 1 << 20);
    hipFuncSetAttribute((const void*)k_empty<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty<REGS>, dim3(grid), dim3(256), lds, 0, d, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty<REGS>, dim3(grid), dim3(256), lds, 0, d, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %4d lds %6zu B: %.2f us per launch\n", name, grid, lds, ms * 1000 / 200);
    hipFree(d);
}
int main()
{
    run<4>("4 values/thread", 0, 512);
    run<4>("4 values/thread", 16 * 1024, 512);
    run<4>("4 values/thread", 79 * 1024, 512);
    run<4>("4 values/thread", 79 * 1024, 256);
    run<4>("4 values/thread", 160 * 1024, 256);
    run<200>("200 values/thread", 0, 512);
    run<200>("200 values/thread", 79 * 1024, 512);
    run<200>("200 values/thread", 79 * 1024, 4096);
    return 0;
}
