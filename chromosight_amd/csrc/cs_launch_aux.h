// cs_launch_aux.h -- launchers of the helper kernels in cs_aux.hip (internal).
#pragma once
#include "cs_device.h"

namespace cs {

struct CsrView {
    int n_rows, n_cols;
    long long nnz;
    const long long* __restrict__ indptr;
    const int* __restrict__ indices;
    const void* __restrict__ data;
    int is_f64;
};

int launch_distance_law(const CsrView& M, const uint8_t* det, int n_diags, double* d_sum,
                        long long* d_cnt, int n_cu, hipStream_t stream);
int launch_detrend_csr(const CsrView& M, const double* law, int n_law, double max_val, void* out,
                       int n_cu, hipStream_t stream);
int launch_csr_to_band(const CsrView& M, const double* law, int n_law, double max_val,
                       const MatView& band, int band_dtype, int n_cu, hipStream_t stream);
int launch_compact_ge(const MatView& corr, int corr_is_f64, int ms, int ns, double threshold,
                      int lo_diag, int hi_diag, int* rows, int* cols, double* vals, long long cap,
                      long long* count, int n_cu, hipStream_t stream);
int launch_rescore_f64(const CorrArgs<double>& A, const int* rows, const int* cols, long long n_px,
                       double* out_corr, double* out_nobs, hipStream_t stream);

// tables of the factorised per-bin mask sums (cs_mask_prep.hip)
template <typename TC>
int launch_mask_tables(const uint8_t* rr, const uint8_t* cc, int ms, int ns, int K, const TC* w, TC* rowtab, TC* coltab,
                       unsigned* rbits, unsigned* cbits, hipStream_t stream);
template <typename TC>
int launch_mask_edge_fix(const unsigned* rbits, const unsigned* cbits, int ms, int ns, int K, int md, int hi_d0, int hi_w,
                         const TC* w, TC* fix_lo, TC* fix_hi, hipStream_t stream);
template <typename TC>
int launch_mask_frame_fix(const unsigned* rbits, const unsigned* cbits, int ms, int ns, int K, int sym_upper, int max_dist,
                          const TC* w, int top, int bot0, int width, int x_band, int x_lo, int side, TC* fix_rows, TC* fix_cols,
                          hipStream_t stream);

}  // namespace cs
