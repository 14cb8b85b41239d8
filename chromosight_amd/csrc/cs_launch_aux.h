// cs_launch_aux.h -- launchers of the helper kernels in cs_aux.hip (internal).
#pragma once
#include "cs_device.h"

namespace cs {

// CSR matrix or a view on a genome-wide pixel table (include/chromosight_hip.h cs_csr)
struct CsrView {
    int n_rows, n_cols;
    long long nnz;
    const long long* __restrict__ indptr;    // row begin offsets
    const long long* __restrict__ row_end;   // row end offsets (plain CSR: indptr + 1)
    const int* __restrict__ indices;
    const void* __restrict__ data;
    int is_f64;
    int col0;                                // stored column - col0 = column of the block
    const double* __restrict__ row_w;        // balancing weights or nullptr
    const double* __restrict__ col_w;
};

// value of stored entry k at (row, col) of the view: balanced on the fly when weights are given
// (count * w[bin1] * w[bin2], what cooler's matrix(balance=True) returns; NaN for unweighted bins)
template <typename TV>
__device__ __forceinline__ double csr_value(const CsrView& M, const TV* __restrict__ data, long long k, int row, int col)
{
    double v = (double)data[k];
    if (M.row_w) v = (v * M.row_w[row]) * M.col_w[col];
    return v;
}

// one record of cs_detect_foci / cs_quantify_pixels (layout of cs_focus in the C ABI)
struct FocusRec {
    int bin1, bin2, inside, n_zero, n_missing, focus_size;
    double score, n_obs;
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
// n_ptr (optional, device): the number of listed pixels actually valid; n_px then only sizes the launch
int launch_rescore_f64(const CorrArgs<double>& A, const int* rows, const int* cols, long long n_px,
                       double* out_corr, double* out_nobs, hipStream_t stream, const long long* n_ptr = nullptr);
int launch_csr_band_extent(const CsrView& M, int lo_diag, int hi_diag, long long* begin, long long* end,
                           hipStream_t stream);
int launch_law_finish(const double* sum, const long long* cnt, int n, double* law, hipStream_t stream);
int launch_narrow_rows(const double* src, long long ld_src, float* dst, long long ld_dst, int rows, int cols, int n_cu,
                       hipStream_t stream);

// median of the stored values of a view (cs_foci.hip)
int csr_median(const CsrView& M, int n_cu, hipStream_t stream, void* (*grow)(void*, size_t), void* user, double* h_median);

// split blocks: candidates of a row window / foci of a merged candidate list (cs_foci.hip)
size_t keep_scratch_bytes(long long n_cand);
int enqueue_keep(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n_cand, double pearson,
                 void* scratch, int** rows_out, int** cols_out, double** vals_out, int** n_kept_out, hipStream_t stream);
size_t label_scratch_bytes(long long n);
int enqueue_label(const int* d_rows, const int* d_cols, const double* d_vals, long long n, int ns, int min_size,
                  int diag_only, void* scratch, int** f_rows_out, int** f_cols_out, int** f_size_out,
                  long long* d_n_foci, hipStream_t stream);

// 1-D patterns: all pixels of a band of a few diagonals as the candidate list (cs_foci.hip)
long long narrow_band_pixels(int rb, int re, int ns, int lo, int w);      // rows rb <= i < re
int enqueue_enumerate_band(int rb, int re, int ns, int lo, int w, long long n, int* d_rows, int* d_cols, hipStream_t stream,
                           int* row_major);      // *row_major = 1: the list is already in row-major order

// device-side foci (cs_foci.hip)
size_t foci_scratch_bytes(long long n_cand);
int enqueue_foci(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n_cand, double pearson,
                 int min_size, int diag_only, int inter, void* scratch, FocusRec** d_rec_out, double* d_windows,
                 long long win_cap, long long* d_n_foci, hipStream_t stream, int presorted, FocusRec* rec_target,
                 long long rec_cap, long long* n_out);
int enqueue_quantify(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n, int inter,
                     double* d_score, double* d_nobs, FocusRec* d_rec, double* d_windows, hipStream_t stream);

// tables of the factorised per-bin mask sums (cs_mask_prep.hip), one launch
template <typename TC>
struct MaskPrepArgs {
    const uint8_t* rr;
    const uint8_t* cc;
    int ms, ns, K, sym_upper, max_dist;
    const TC* w;
    TC* rowtab;
    TC* coltab;
    int edge, hi_d0, hi_w;          // edge corrections (band outputs ending near max_dist)
    TC* fix_lo;
    TC* fix_hi;
    int top, bot0, width, x_band, x_lo, side;   // frame corrections
    TC* fix_rows;
    TC* fix_cols;
    int b_tab, b_edge;              // block ranges (filled by the launcher)
};
template <typename TC>
int launch_mask_prep(MaskPrepArgs<TC> P, hipStream_t stream);

}  // namespace cs
