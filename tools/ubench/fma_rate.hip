// Micro-benchmark: issue rate of the FP32 FMA forms the hot kernel can use on gfx950.
//   hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate && ./fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1}, sv = {s, s};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // v_fma_f32 vgpr,vgpr,vgpr : 8 independent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (MODE == 1) {  // v_fmac_f32 with an SGPR multiplier (acc += v * s)
            REP16(asm volatile("v_fmac_f32 %0, %9, %8\n v_fmac_f32 %1, %9, %8\n v_fmac_f32 %2, %9, %8\n v_fmac_f32 %3, %9, %8\n"
                               "v_fmac_f32 %4, %9, %8\n v_fmac_f32 %5, %9, %8\n v_fmac_f32 %6, %9, %8\n v_fmac_f32 %7, %9, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(s));)
        } else if (MODE == 2) {  // v_pk_fma_f32 vgpr pairs: 4 independent chains of 2
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (MODE == 3) {  // v_pk_fma_f32 with an SGPR pair multiplier
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "s"(sv));)
        } else if (MODE == 4) {  // v_fma_f32 VOP3 with SGPR: acc = v * s + acc
            REP16(asm volatile("v_fma_f32 %0, %9, %8, %0\n v_fma_f32 %1, %9, %8, %1\n v_fma_f32 %2, %9, %8, %2\n v_fma_f32 %3, %9, %8, %3\n"
                               "v_fma_f32 %4, %9, %8, %4\n v_fma_f32 %5, %9, %8, %5\n v_fma_f32 %6, %9, %8, %6\n v_fma_f32 %7, %9, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(s));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(int blocks, int iters, int flop_per_iter_per_lane)
{
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 16, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    double flop = (double)blocks * 256 * iters * flop_per_iter_per_lane;
    return flop / (ms * 1e-3) / 1e12;
}

int main()
{
    const int iters = 2000;
    for (int wpc : {1, 2, 4, 8}) {  // workgroups (4 waves) per CU
        int blocks = 256 * wpc;
        // per iteration per lane: 16 reps x 8 FMA x 2 flop (modes 0,1,4); 16 x 4 pk x 4 flop (modes 2,3)
        printf("waves/SIMD=%d  v_fma vvv %.1f TF | v_fmac sgpr %.1f TF | v_pk_fma vvv %.1f TF | v_pk_fma sgpr %.1f TF | v_fma vop3 sgpr %.1f TF\n",
               wpc, run<0>(blocks, iters, 256), run<1>(blocks, iters, 256), run<2>(blocks, iters, 256),
               run<3>(blocks, iters, 256), run<4>(blocks, iters, 256));
    }
    return 0;
}
