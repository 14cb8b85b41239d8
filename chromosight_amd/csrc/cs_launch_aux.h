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

#ifdef __HIPCC__
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// float64 coefficient (and present-pixel count) of one pixel by one wave: _normxcorr2_sparse per pixel
// (detection.py:917-1131).  `A` may live in registers (kernel argument) or in global memory (a table of
// per-sub-matrix arguments: the wave's loads are uniform).
__device__ __forceinline__ void rescore_pixel(const CorrArgs<double>& A, int oi, int oj, int lane, double& r, double& nobs)
{
    r = 0.0;
    nobs = A.ks.n;
    const bool inside = (oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns);
    if (inside && !pixel_forced_zero(A, oi, oj)) {       // wave-uniform
        const int km = A.km, kn = A.kn, kk = km * kn;
        const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
        const bool masked = A.mask_mode != 0;
        double cs_ = 0, s1 = 0, s2 = 0, nm = 0, ka = 0, kb = 0;
        for (int e = lane; e < kk; e += 64) {
            const int ki = e / kn, kj = e - ki * kn;
            const int p = oi - kh + ki, q = oj - kw + kj;
            const double v = load_signal(A, p, q);
            cs_ = fma(v, A.w[e], cs_);
            s1 += v;
            s2 = fma(v, v, s2);
            if (masked && missing_pred(A, p, q)) {
                nm += 1.0;
                ka += A.w[kk + e];
                kb += A.w[2 * kk + e];
            }
        }
        cs_ = wave_sum(cs_);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (masked) {
            nm = wave_sum(nm);
            ka = wave_sum(ka);
            kb = wave_sum(kb);
        }
        r = pearson_from_sums<double>(cs_, s1, s2, nm, ka, kb, A.ks, masked, &nobs);
    }
}
#endif

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

// 1-D patterns of many sub-matrices in one launch chain (cs_foci.hip)
constexpr long long kFociSmallMax = 1 << 16;       // candidates of one sub-matrix a single workgroup labels
size_t narrow_batch_scratch_bytes(int n_blocks, long long n_total);
int enqueue_foci_narrow_batch(const CorrArgs<double>* h_tab, const long long* h_seg, const int* h_lo_w, int n_blocks, double pearson,
                              int min_size, int diag_only, int inter, void* scratch, FocusRec* rec, long long rec_cap,
                              double* windows, long long win_cap, long long* h_counts, hipStream_t stream);

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
