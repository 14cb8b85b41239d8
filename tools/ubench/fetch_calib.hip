// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the streaming kernel's row
// staging: every lane issues 4 separate 4-byte global loads of 4 consecutive floats (a wave reads
// 1 KiB contiguous).  The buffer (2 GiB) is far beyond the 256 MiB Infinity Cache and every byte is
// read exactly once, so the true HBM read volume is known.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void read4x4(const float* __restrict__ p, float* out, size_t n_vec4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n_vec4; i += stride) {
        const volatile float* q = p + 4 * i;
        acc += q[0];
        acc += q[1];
        acc += q[2];
        acc += q[3];
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ __launch_bounds__(256) void read16(const float4* __restrict__ p, float* out, size_t n_vec4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n_vec4; i += stride) {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    float *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const size_t n4 = bytes / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read4x4, dim3(256 * 16), dim3(256), 0, 0, buf, out, n4);
        hipLaunchKernelGGL(read16, dim3(256 * 16), dim3(256), 0, 0, (const float4*)buf, out, n4);
    }
    hipDeviceSynchronize();
    printf("each launch reads %zu bytes = %.1f KiB\n", bytes, bytes / 1024.0);
    return 0;
}
