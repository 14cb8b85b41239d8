// Micro-benchmark: issue rate of the two float16 MFMA shapes the two-pass kernel could use on gfx950 -- v_mfma_f32_16x16x32_f16 (k = 32)
// and the older v_mfma_f32_16x16x16_f16 (k = 16: all a second Toeplitz pass of a template of up to 33 columns needs).  One wave per
// SIMD (256 threads per CU-resident workgroup), 8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    h8 a8, b8;
    h4 a4, b4;
    for (int e = 0; e < 8; ++e) {
        a8[e] = (_Float16)(threadIdx.x * 0.001f + e);
        b8[e] = (_Float16)(1.0f + e * 0.01f);
    }
    for (int e = 0; e < 4; ++e) {
        a4[e] = a8[e];
        b4[e] = b8[e];
    }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int kdim)
{
    int dev = 0, cus = 0, mhz = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev);
    float* out;
    hipMalloc(&out, sizeof(float) * cus * 256);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<cus, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<cus, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 8;                       // per wave
    const double flop = n_mfma * 4 * cus * 2.0 * 16 * 16 * kdim;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  %6.1f ns per MFMA and wave (= %.1f cycles at the reported %d MHz)\n", name, ms, flop / (ms * 1e-3) / 1e12,
           ms * 1e6 / n_mfma, ms * 1e-3 / n_mfma * mhz * 1e3, mhz / 1000);
    hipFree(out);
}

int main()
{
    run<0>("v_mfma_f32_16x16x32_f16", 32);
    run<1>("v_mfma_f32_16x16x16_f16", 16);
    return 0;
}
