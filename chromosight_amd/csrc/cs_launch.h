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
// separable evaluation of exactly rank-1 templates (cs_corr_sep.hip): u at A.w + 3 km kn, v behind it
int launch_corr_sep_f32(const CorrArgs<float>& A, hipStream_t s);
bool corr_sep_fits(int km, int kn, bool masked);
void corr_sep_tile(int* tw, int* th);

// matrix-core kernel for templates up to 17 x 17, float32 class (cs_corr_mfma.hip).  The weight sets
// arrive as ready-made B fragments: frag[set][s][head | tail][lane] = 8 float16 values
// W_set[s][8 (lane >> 4) + e - (lane & 15)] * 2^ew (0 outside 0 .. kn-1); unscale[set] = 2^-ew.
struct MfmaWeights {
    const uint4* frag;
    float unscale[3];
    // Rim tables of the mask weight sets Wa (set 0) and Wb (set 1), float32, fixed strides whatever K is: what the masked
    // tile kernel needs to form, per pixel, the correction of a window that reaches below the main diagonal or beyond
    // max_dist (reference preprocessing.py:404-498 frame_missing_mask: the sub-diagonals are flagged, pixels beyond
    // max_dist never are) -- the arithmetic of cs_mask_prep.hip mask_edge_fix without its 16-byte record per rim pixel:
    //   PW[set][ki][m] = sum_{kj < m} W[ki][kj]   at kRimPW + (set * 17 + ki) * 18 + m
    //   QW[set][kj][m] = sum_{ki < m} W[ki][kj]   at kRimQW + (set * 17 + kj) * 18 + m
    //   W[set][ki][kj]                            at kRimW  + set * 289 + ki * 17 + kj
    //   base[x][D], x = {count, Wa, Wb}: the whole lower triangle of diagonal D < K - 1   at kRimBase + x * 17 + D
    const float* rim;
};
constexpr int kRimPW = 0, kRimQW = 2 * 17 * 18, kRimW = 4 * 17 * 18, kRimBase = 4 * 17 * 18 + 2 * 289;
constexpr int kRimFloats = kRimBase + 3 * 17;
int launch_corr_mfma_f32(CorrArgs<float>& A, const MfmaWeights& E, hipStream_t s, int* dense_path);
// Several matrices in one persistent launch of the masked candidate instance (corr_mfma_blocks_kernel).  h_table: page-locked,
// mfma_blocks_table_bytes(n) bytes, argument block b (CorrArgs::defer_args of launch_corr_mfma_f32) at
// h_table + mfma_blocks_arg_offset(n) + b * mfma_blocks_arg_bytes(); d_table: as many device bytes.  The table is
// completed (tile ranges), uploaded and the kernel launched on `s`; both buffers stay untouched until `s` has drained.
size_t mfma_blocks_arg_bytes();
size_t mfma_blocks_arg_offset(int n_blocks);
size_t mfma_blocks_table_bytes(int n_blocks);
int mfma_blocks_table_finish(void* h_table, int n_blocks);      // fills the table's tile ranges (for an upload made by the caller)
int launch_corr_mfma_blocks(void* h_table, void* d_table, int n_blocks, int rsym, int n_cu, hipStream_t s, hipStream_t upload, bool do_upload, bool launch,
                            unsigned* started = nullptr, unsigned epoch = 0);
// one block of such a table as its own persistent launch (its mask tables were prepared earlier)
int launch_corr_mfma_prepared(const void* h_arg, int rsym, int n_cu, int grid_cap, hipStream_t s);

// matrix-core kernel for templates of up to 33 x 33 (cs_corr_wide.hip): two k = 32 Toeplitz passes per template row.
// frag[set][s][pass][head | tail][lane] = 8 float16 values W_set[s][32 pass + 8 (lane >> 4) + e - (lane & 15)] * 2^ew
// (0 outside 0 .. kn-1), unscale[set] = 2^-ew; sums: row sums of Wa (33 slots), row sums of Wb, column sums of Wa,
// column sums of Wb -- the 1-D tables of the factorised per-bin mask; plane_only: never factorise (test switch).
struct MfmaWideWeights {
    const uint4* frag;
    float unscale[3];
    const float* sums;
    int plane_only;
    int plane_only_staging;      // ... and stage every tile pixel by pixel with the general predicate (test switch)
    int one_launch;              // per-bin masks: 1 in ONE launch, 2 always in two (test switches; default: two on wide bands, inner tiles first)
    int tile_mode;               // set by the launcher: 0 every tile, 1 inner tiles only (small LDS image), 2 all but the inner tiles
};
bool corr_mfma_wide_fits(int km, int kn);
int launch_corr_mfma_wide_f32(CorrArgs<float>& A, const MfmaWideWeights& E, hipStream_t s);

}  // namespace cs
