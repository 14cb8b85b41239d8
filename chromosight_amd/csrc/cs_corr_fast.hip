// cs_corr_fast.hip -- explicit instantiations of the LDS-tiled Pearson kernel for one
// compile-time template size.  Compiled once per size with -DCS_K=<odd K> so the sizes
// build in parallel (each fully unrolled kernel takes tens of seconds to compile).
#include "cs_corr_tile.h"
#include "cs_launch.h"

#ifndef CS_K
#error "compile with -DCS_K=<odd kernel size>"
#endif
#ifndef CS_RH
#define CS_RH 8
#endif

namespace cs {

template <typename TC>
static int launch_fast(const CorrArgs<TC>& A, hipStream_t stream)
{
    using G = TileGeom<CS_K, CS_RH>;
    constexpr size_t smem = corr_tile_smem_bytes<CS_K, CS_RH, TC>();
    dim3 grid(A.tiles_x, A.tiles_y), block(G::NTHREADS);
    if (A.mask_mode != 0) {
        auto kern = corr_tile_kernel<TC, CS_K, CS_RH, true>;
        if (smem > 48 * 1024)
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, grid, block, smem, stream, A);
    } else {
        auto kern = corr_tile_kernel<TC, CS_K, CS_RH, false>;
        if (smem > 48 * 1024)
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, grid, block, smem, stream, A);
    }
    return (int)hipGetLastError();
}

#define CS_CAT_(a, b) a##b
#define CS_CAT(a, b) CS_CAT_(a, b)

int CS_CAT(launch_corr_fast_f32_k, CS_K)(const CorrArgs<float>& A, hipStream_t s) { return launch_fast<float>(A, s); }
int CS_CAT(launch_corr_fast_f64_k, CS_K)(const CorrArgs<double>& A, hipStream_t s) { return launch_fast<double>(A, s); }
void CS_CAT(corr_fast_tile_k, CS_K)(int* tw, int* th)
{
    *tw = TileGeom<CS_K, CS_RH>::TW;
    *th = TileGeom<CS_K, CS_RH>::TH;
}

}  // namespace cs
