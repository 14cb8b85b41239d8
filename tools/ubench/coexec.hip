// Micro-benchmark: do MFMA waves and packed-FP32-FMA waves co-execute on one SIMD at full rate?
// One 512-thread workgroup per CU = 2 waves per SIMD; waves 0-3 run role A, waves 4-7 role B.
//   hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x

// ROLE: 0 idle, 1 = v_mfma_f32_16x16x32_bf16, 2 = v_mfma_f32_16x16x4_f32, 3 = v_pk_fma_f32 (sgpr weight),
//       4 = v_fma_f32 vvv
template <int ROLE>
__device__ __forceinline__ float work(int iters, float s)
{
    float r = 0.f;
    if (ROLE == 1) {
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(1.0f + i); }
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                 c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                 c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
                 c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
                 c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0);
                 c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);)
        }
        f4 c = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        r = c.x + c.y + c.z + c.w;
    } else if (ROLE == 2) {
        float a = threadIdx.x, b = 1.5f;
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                 c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                 c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                 c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
                 c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
                 c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
                 c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);)
        }
        f4 c = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        r = c.x + c.y + c.z + c.w;
    } else if (ROLE == 3) {
        float a0 = threadIdx.x;
        f2 p0 = {a0, a0 + 1}, p1 = {a0 + 2, a0 + 3}, p2 = {a0 + 4, a0 + 5}, p3 = {a0 + 6, a0 + 7}, q = {1.0001f, 0.9999f},
           sv = {s, s};
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n"
                              "v_pk_fma_f32 %3, %4, %5, %3\n v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n"
                              "v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                              : "v"(q), "s"(sv));)
        }
        r = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    } else if (ROLE == 4) {
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        float b0 = 1.0001f, b1 = 0.9999f;
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0), "v"(b1));)
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    return r;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* out, int* simd, int iters_a, int iters_b, float s)
{
    const int wave = threadIdx.x >> 6;
    float r;
    if (wave < 4) r = work<RA>(iters_a, s);
    else r = work<RB>(iters_b, s);
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        simd[wave] = (hw >> 4) & 3;
    }
}

template <int RA, int RB>
float run(int ia, int ib)
{
    float* out; int* simd;
    hipMalloc(&out, sizeof(float) * 256 * 512);
    hipMalloc(&simd, 8 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<RA, RB><<<256, 512>>>(out, simd, 8, 8, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<RA, RB><<<256, 512>>>(out, simd, ia, ib, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int h[8];
    hipMemcpy(h, simd, sizeof(h), hipMemcpyDeviceToHost);
    static bool shown = false;
    if (!shown) { printf("wave->SIMD: %d %d %d %d | %d %d %d %d\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]); shown = true; }
    hipFree(out); hipFree(simd);
    return ms;
}

int main()
{
    // iteration counts: 64 instructions per iteration for every role
    const int n = 4000;
    // flops per wave-instruction: bf16 16x16x32 = 16384, f32 16x16x4 = 2048, pk_fma = 256, fma = 128
    auto tf = [&](double flop_per_instr, int iters, float ms) { return 256.0 * 4 * iters * 64 * flop_per_instr / (ms * 1e-3) / 1e12; };
    float a, b, c;
    a = run<1, 0>(n, 0); b = run<0, 3>(0, 4 * n); c = run<1, 3>(n, 4 * n);
    printf("bf16 mfma alone %.3f ms (%.0f TF) | pk_fma alone %.3f ms (%.1f TF) | both %.3f ms  (sum if serial %.3f)\n", a, tf(16384, n, a), b,
           tf(256, 4 * n, b), c, a + b);
    a = run<2, 0>(n, 0); b = run<0, 3>(0, 8 * n); c = run<2, 3>(n, 8 * n);
    printf("f32 mfma alone %.3f ms (%.0f TF) | pk_fma alone %.3f ms (%.1f TF) | both %.3f ms  (sum if serial %.3f)\n", a, tf(2048, n, a), b,
           tf(256, 8 * n, b), c, a + b);
    a = run<1, 0>(n, 0); b = run<0, 4>(0, 4 * n); c = run<1, 4>(n, 4 * n);
    printf("bf16 mfma alone %.3f ms (%.0f TF) | v_fma alone %.3f ms (%.1f TF) | both %.3f ms  (sum if serial %.3f)\n", a, tf(16384, n, a), b,
           tf(128, 4 * n, b), c, a + b);
    a = run<1, 1>(n, n);
    printf("bf16 mfma on both waves %.3f ms (%.0f TF)\n", a, 2 * tf(16384, n, a));
    a = run<3, 3>(4 * n, 4 * n);
    printf("pk_fma on both waves %.3f ms (%.1f TF)\n", a, 2 * tf(256, 4 * n, a));
    return 0;
}
