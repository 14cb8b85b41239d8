// cs_launch.h -- host-visible launchers of the device kernels (internal to the library).
#pragma once
#include "cs_device.h"

namespace cs {

// fast, fully unrolled square-template kernels (cs_corr_fast.hip, one object per size)
#define CS_DECL_FAST(K)                                                        \
    int launch_corr_fast_f32_k##K(const CorrArgs<float>& A, hipStream_t s);    \
    int launch_corr_fast_f64_k##K(const CorrArgs<double>& A, hipStream_t s);   \
    void corr_fast_tile_k##K(int ms, int ns, int band_w, int n_cu, int* tw, int* th);
CS_DECL_FAST(7)
CS_DECL_FAST(9)
CS_DECL_FAST(11)
CS_DECL_FAST(13)
CS_DECL_FAST(15)
CS_DECL_FAST(17)
#undef CS_DECL_FAST

// generic runtime-size kernel (cs_corr_generic.hip)
int launch_corr_generic_f32(const CorrArgs<float>& A, hipStream_t s);
int launch_corr_generic_f64(const CorrArgs<double>& A, hipStream_t s);
void corr_generic_tile(int km, int kn, int* tw, int* th);

}  // namespace cs
