// cs_device.h -- device-side parameter blocks and helpers shared by all kernels.
//
// gfx950 / CDNA4 only (wave64, 160 KiB LDS per CU).  No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cs {

constexpr int kWave = 64;

// One matrix operand as the kernels see it (mirrors cs_matrix of the C ABI).
struct MatView {
    void* ptr;
    long long ld;
    int layout;   // 0 dense row-major, 1 diagonal band, 2 diagonal band recomputed on demand (ptr -> LazyBand, below)
    int band_lo;  // first stored diagonal offset (band)
    int band_w;   // number of stored diagonals (band)
    long long row0;  // matrix row stored at ptr: the buffer holds a window of rows (0 = the whole matrix)
    int pad;      // band: the slots band_w .. ld - 1 of every row (>= 4) and every slot outside the matrix are zero (CS_LAYOUT_BAND_PADDED)
    int counts;   // band of RAW COUNTS (CS_LAYOUT_BAND_COUNTS, float32, zero-padded like `pad`): the reader balances and detrends
                  // what it fetches (CountsHeader in the kCountsHeaderBytes in front of ptr) -- the masked tile kernel only
};

// CS_LAYOUT_BAND_COUNTS: the staging pass of a genome (cs_stage.hip) writes the band of a block ONCE, in the same pass over the
// pixel table that reduces its distance law, as the raw counts (exact in float32: integers below 2^24); balancing
// (contacts_map.py:531-540) and the detrend by the law (preprocessing.py:296-302) are a function of a pixel's count, its row and
// column weights and its diagonal, and are applied by whoever reads a pixel -- the tile kernel while it splits a landed tile into
// float16 planes, the float64 kernels pixel by pixel (LazyBand::counts) -- with the operations of stage_detrend_rcp, in its
// order: bit for bit the values the detrended bands held.  No second pass over the pixel table, no detrended band in HBM.
// The header sits in the 128 bytes in front of the band (written by stage_finish_kernel).
struct CountsHeader {
    const double* weight;      // the genome's ICE weights (NaN: undetectable bin)
    const double* law;         // the block's distance law, n_diags values
    const double* rlaw;        // 1 / law, n_diags values with one more slot on either side (rlaw[-1], rlaw[n_diags]: zeros)
    long long row0;            // first genome bin of the block
    double max_val;
    int n, n_diags;
    const float* weight32;     // the block's n weights and the n_diags + 2 reciprocals rounded to float32 (rlaw32[-1 .. n_diags]): what
    const float* rlaw32;       // the float32 tile kernel multiplies a landed tile's counts with
};
// layout of a counts block's d_law buffer (include/chromosight_hip.h CS_COUNTS_LAW_BYTES): law[n_diags], rlaw[-1 .. n_diags],
// rlaw32[-1 .. n_diags] (padded to 8 bytes), weight32[n]
__host__ __device__ inline long long counts_law_bytes(int n, int n_diags)
{
    return 8ll * (2 * n_diags + 2) + 4ll * ((n_diags + 2 + 1) / 2 * 2) + 4ll * n;
}
constexpr int kCountsHeaderBytes = 128;
static_assert(sizeof(CountsHeader) <= kCountsHeaderBytes, "CS_COUNTS_HEADER_BYTES of include/chromosight_hip.h");

// element offset of (p, q), or -1 if the pixel is not stored
__device__ __forceinline__ long long mat_offset(const MatView& m, int p, int q)
{
    if (m.layout == 0) return ((long long)p - m.row0) * m.ld + q;
    int d = q - p - m.band_lo;
    if (d < 0 || d >= m.band_w) return -1;
    return ((long long)p - m.row0) * m.ld + d;
}

// layout 2 (CS_LAYOUT_BAND_LAZY): the float64 band of a staged intra block WITHOUT its storage.  A genome's float64 bands
// are twice the bytes of the float32 bands the tile kernel reads, and the float64 consumers -- the exact evaluation of
// the candidates, the windows of the records -- touch 1e-3 of them; writing them was a third of the staging pass.  The
// pixel (p, q) of the band is a pure function of the genome's pixel table, the block's distance law and the ICE weights
// (cs_stage.hip stage_tile_kernel: stage_value below), so MatView::ptr points to this descriptor (device memory,
// written by cs_stage_blocks) and load_signal recomputes the value: from `near` for the first near_w diagonals (stored:
// the runs of a 1-D pattern live there), else by a binary search of the pixel table's row.  band_lo / band_w of the
// view still bound the diagonals that read as non-zero (a narrower view of the same block shares the descriptor).
struct LazyBand {
    const long long* indptr;       // the genome's pixel table (CSR over all bins, columns >= row)
    const int* indices;
    const void* data;              // float32 or float64 counts
    const double* weight;          // ICE weights (NaN: undetectable bin)
    const double* law;             // the block's distance law, n_diags values
    const double* near_;           // float64 band of the diagonals 0 .. near_w - 1, row pitch near_ld (or null)
    long long row0;                // first genome bin of the block
    long long near_ld;
    double max_val;
    int n, n_diags, near_w, data_is_f64;
    const float* counts;           // (or null) the block's band of raw counts, every kept diagonal, row pitch counts_ld: the stored
    long long counts_ld;           //  diagonals are then ALL of them (near_w = n_diags, near_ null) and nothing is searched
};

// the staged value of one stored pixel (contacts_map.py:531-540, preprocessing.py:296-302): balance, detrend by the law of
// its diagonal (0: inf / NaN), >= max_val -> 1, NaN -> 0.
// The staging pass of a genome (cs_stage.hip stage_tile_kernel) multiplies by the RECIPROCAL of the law, taken once per
// diagonal and workgroup, instead of dividing every stored pixel by it: the float64 division sequence (11 instructions, one of
// them the quarter-rate reciprocal) was a quarter of what that kernel issues per stored pixel, and the kernel is bound by
// issue, not by bytes.  v * (1 / y) differs from the reference's v / y (preprocessing.py:298) by at most one unit in the last
// place of a float64 -- 1e-16 relative against tolerances of 1e-11 and wider everywhere downstream -- and keeps its special
// cases: law 0 -> 1 / 0 = inf -> v * inf = inf (>= max_val -> 1) or NaN for v = 0 (-> 0), as inf and NaN of the division.
// Whatever recomputes a staged pixel (the lazily evaluated bands) uses THIS function, so that the two stay bit-identical.
// The cap is a DISCRETE decision (>= max_val -> 1): where the product lands within rounding of max_val -- unbalanced integer
// counts over a rational law can make v / y exactly 10 while v * (1 / y) rounds to 10 - ulp -- it is made with the reference's
// own quotient (`y_at`: address of the law's value, read only then: once in ~ 1e15 pixels of real data).
__device__ __forceinline__ double stage_detrend_rcp(double v, double inv_y, double max_val, const double* y_at)
{
    double out = v * inv_y;
    if (max_val > 0.0) {
        if (fabs(out - max_val) <= max_val * 8.9e-16) out = v / *y_at;      // (4 ulp; inf and NaN never come here)
        if (out >= max_val) out = 1.0;                     // :301-302
    }
    return out != out ? 0.0 : out;                         // NaN -> 0 (contacts_map.py:539-540)
}

// the staged value of slot (p, d) of a band of counts: what the detrended band held there (0 for an empty slot: a stored count
// of 0 is 0 / law = 0, or NaN -> 0 on an empty diagonal)
// (`law`: the block's d_law buffer as cs_stage_blocks lays it out for a band of counts -- n_diags values of the law, then their
// reciprocals with one slot on either side: the reciprocal the staging's finish pass took is READ, not taken again per pixel
// -- the same IEEE quotient, and a float64 division is ~ 30 instructions)
__device__ __forceinline__ double counts_value(float c, const double* __restrict__ weight, const double* __restrict__ law, long long row0,
                                               int p, int d, double max_val, int n_diags)
{
    if (c == 0.0f) return 0.0;
    return stage_detrend_rcp(((double)c * weight[row0 + p]) * weight[row0 + p + d], law[n_diags + 1 + d], max_val, law + d);
}

// the same value with every load issued at once (the count, both weights, the reciprocal): the wave-per-window and tile-staging
// kernels wait for round trips, not for bytes, and the early exit above makes the weights' loads wait for the count's.  A count
// of 0 gives 0 (or NaN -> 0) through the arithmetic itself.
__device__ __forceinline__ double counts_value_flat(const float* __restrict__ cnt_at, const double* __restrict__ weight,
                                                    const double* __restrict__ law, long long row0, int p, int d, double max_val, int n_diags)
{
    const float c = *cnt_at;
    const double wr = weight[row0 + p], wc = weight[row0 + p + d], rl = law[n_diags + 1 + d];
    return stage_detrend_rcp(((double)c * wr) * wc, rl, max_val, law + d);
}

// slot (p, d), d < near_w, of the diagonals a lazily evaluated band keeps in memory
__device__ __forceinline__ double lazy_near_value(const LazyBand& L, int p, int d)
{
    if (L.counts) return counts_value(L.counts[(long long)p * L.counts_ld + d], L.weight, L.law, L.row0, p, d, L.max_val, L.n_diags);
    return L.near_[(long long)p * L.near_ld + d];
}

__device__ __forceinline__ double lazy_stored_value(const LazyBand& L, long long k, int p, int q)
{
    const double x = L.data_is_f64 ? reinterpret_cast<const double*>(L.data)[k] : (double)reinterpret_cast<const float*>(L.data)[k];
    const int d = q - p;
    const int dc = min(d, L.n_diags - 1);
    return stage_detrend_rcp((x * L.weight[L.row0 + p]) * L.weight[L.row0 + q], 1.0 / (d < L.n_diags ? L.law[d] : 0.0), L.max_val, L.law + dc);
}

// pixel (p, q) of the block, 0 <= p <= q < n, on a diagonal the view keeps
__device__ __forceinline__ double lazy_load(const MatView& m, int p, int q)
{
    const LazyBand& L = *reinterpret_cast<const LazyBand*>(m.ptr);
    const int d = q - p;
    if (d - m.band_lo < 0 || d - m.band_lo >= m.band_w || d < 0 || d >= L.n_diags) return 0.0;
    if (d < L.near_w) return lazy_near_value(L, p, d);
    const long long r = L.row0 + p;
    long long lo = L.indptr[r], hi = L.indptr[r + 1];
    const long long end = hi;
    const int c = (int)(L.row0 + q);
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (L.indices[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= end || L.indices[lo] != c) return 0.0;
    return lazy_stored_value(L, lo, p, q);
}

template <typename T>
struct KernelStats {
    T n;        // km * kn
    T inv_n;
    T kmean;    // mean of the exact template
    T kstd;     // population std of the exact template (numpy .std())
    T kvar;     // k2mean - kmean^2
    T ksum;
    T k2sum;
    T thr;      // 1e-4
    T eps;      // 1e-10
    T cut;      // int((1 - missing_tol) * n) as a float
    // derived constants of the lean float32 epilogues (filled by cs_api.cpp build_args)
    T thr_n;    // thr * n: |sum| below it <=> |mean| below the reference's 1e-4 zeroing threshold
    T nkvar;    // n * kvar = sum of the squared centred template
    T eps2;     // eps^2
    T den2_min; // eps^2 * n^2: (s2 n - s1^2) kvar below it <=> unmasked denominator below eps
    int zk_possible;    // a mask-weighted template sum can fall under thr (some |K'| or K'^2 entry is tiny)
    int snap_possible;  // the template variance over the present pixels can degenerate (few distinct values)
    // cand_cmin > 0: candidate mode of the float32 kernels (cs_detect_foci / cs_candidates, see cand_screen_* below): a
    // pixel keeps its value only if it is below cand_thr on a window conditioned at least cand_cmin, else it stores 2.0
    T cand_cmin;
    T cand_thr;
};

// Arguments of the tile kernels.  Weights live in a small device buffer:
//   w[0 .. kk)      conv weights minus kmean            (sum S * (K' - kmean))
//   w[kk .. 2kk)    mask weights Wa   (float64: K'; float32: K' - kmean)
//   w[2kk .. 3kk)   mask weights Wb   (float64: K'^2; float32: the centred form), see cs_api.cpp
template <typename TC>
struct CorrArgs {
    MatView sig;
    MatView out;
    MatView nobs;            // ptr == nullptr -> not written; float elements
    int sig_is_f64;
    int out_is_f64;
    int ms, ns;
    int km, kn;
    int full, sym_upper, max_dist;
    int mask_mode;           // 0 none, 1 bins, 2 explicit
    const uint8_t* miss_row;
    const uint8_t* miss_col;
    MatView mask;            // explicit mask (uint8), same geometry as sig
    const TC* __restrict__ w;
    KernelStats<TC> ks;
    int xcorr_only;          // 1: out = thresholded sum S*w[0..kk) (plain xcorr2)
    // tile grid
    int tiles_x;             // tiles per row-block
    int tiles_y;
    int tile_w, tile_h;
    int out_lo, out_hi;      // only pixels with out_lo <= j - i <= out_hi are produced
    int row_begin, row_end;  // only output rows row_begin <= i < row_end are produced (a row window of the
                             // map: API calls pipelined over PCIe, one block split over several GPUs);
                             // the signal buffer must hold the rows row_begin - kh .. row_end + kh - 1
    int n_cu;                // compute units of the device (launch shaping)
    int grid_cap;            // > 0: at most this many persistent workgroups (tile kernel launches that run side by side)
    int w_sym;               // all three weight sets are symmetric under a vertical flip (row s == row km-1-s)
    int w_rank1;             // the template is exactly u v^T: u (km values) and v (kn) follow the three weight sets in `w`
    // factorised per-bin mask sums of the streaming kernel (cs_mask_prep.hip); reg_mode = 1:
    // strips whose windows stay inside the matrix use the tables, the others the general path
    int reg_mode;
    // candidate mode (ks.cand_cmin > 0) with a sink: the masked matrix-core tile kernel appends the coordinates of its
    // candidate pixels (keys cand_tag + row * ns + col) to this list instead of writing a map; `out.ptr` may then be null
    unsigned long long* cand_keys;
    unsigned long long* cand_count;
    long long cand_cap;
    unsigned long long cand_tag;
    // non-null (candidate mode, masked tile kernel only): do not launch -- write the kernel's argument block here
    // (mfma_blocks_arg_bytes() bytes, host memory) for launch_corr_mfma_blocks; defer_rsym: which instance it needs
    void* defer_args;
    int* defer_rsym;
    int cand_dlo, cand_dhi;  // only pixels on these diagonals are candidates (diag_trim of the coefficient map)
    int fix_on, fix_hi_w, fix_hi_d0;
    const TC* rowtab;
    const TC* coltab;
    const TC* fix_lo;
    const TC* fix_hi;
    // frame corrections: rows < fix_top and rows >= fix_bot0 (fix_width entries each, indexed by
    // column, or by diagonal - fix_xlo when fix_xband), first / last fix_side columns of every row
    const TC* fix_rows;
    const TC* fix_cols;
    int fix_top, fix_bot0, fix_width, fix_xband, fix_xlo, fix_side;
    int rim_in_kernel;       // edge mode without fix_lo / fix_hi records: the masked tile kernel forms them (MfmaWeights::rim)
};

// ---------------------------------------------------------------------------------------
// The framed missing predicate (reference preprocessing.py:535-633 make_missing_mask and
// :404-498 frame_missing_mask), evaluated analytically for matrix coordinates (p, q) that
// may lie in the virtual frame (p in [-(km-1), ms+km-1), q likewise).
// ---------------------------------------------------------------------------------------
template <typename TC>
__device__ __forceinline__ bool missing_pred(const CorrArgs<TC>& A, int p, int q)
{
    if (A.mask_mode == 0) return false;
    const bool in_r = (p >= 0) & (p < A.ms);
    const bool in_c = (q >= 0) & (q < A.ns);
    const int d = q - p;
    const bool have_md = A.max_dist >= 0;
    bool m = false;
    if (in_r & in_c) {
        if (A.mask_mode == 1) {
            m = (A.miss_row[p] | A.miss_col[q]) != 0;
            if (A.sym_upper) {
                const int md = have_md ? A.max_dist : min(A.ms, A.ns);
                m = m & (d >= 0) & (d <= md);
            }
        } else {
            long long off = mat_offset(A.mask, p, q);
            m = (off >= 0) ? (((const uint8_t*)A.mask.ptr)[off] != 0) : false;
            if (A.full && A.sym_upper && have_md) {
                // frame_missing_mask trims the mask to diagonals 0 .. max_dist + max(k)
                const int lim = A.max_dist + max(A.km, A.kn);
                m = m & (d >= 0) & (d <= lim);
            }
        }
        if (!A.full) return m;
    } else {
        if (!A.full) return false;  // no frame in 'valid' mode
        if (A.sym_upper && have_md) {
            if (q >= A.ns) {
                m = p >= A.ms - A.max_dist - 2;          // right margin, last max_dist+km+1 framed rows
            } else if (p < 0) {
                m = (q < 0) ? true : (q < A.max_dist + A.kn);  // top-left corner / top margin
            } else {
                m = false;                                // left and bottom margins are not flagged
            }
        } else {
            m = true;                                     // all four margins
        }
    }
    if (A.sym_upper) {
        const int off = d + (A.kn - A.km);                // diagonal offset in framed coordinates
        const int big_k = max(A.km, A.kn);
        m = m | ((off <= -1) & (off >= -big_k));
    }
    return m;
}

// missing_pred with the mask values already loaded (rflag / cflag: per-bin flags of row p / column q,
// mval: the explicit map's byte; each is ignored where (p, q) lies outside the matrix or the stored
// band, so callers may load them from a clamped address without branching).
template <typename TC>
__device__ __forceinline__ bool missing_from_flags(const CorrArgs<TC>& A, int p, int q, bool rflag, bool cflag, bool mval,
                                                   bool stored)
{
    if (A.mask_mode == 0) return false;
    const bool in_r = (p >= 0) & (p < A.ms);
    const bool in_c = (q >= 0) & (q < A.ns);
    const int d = q - p;
    const bool have_md = A.max_dist >= 0;
    bool m;
    if (in_r & in_c) {
        if (A.mask_mode == 1) {
            m = rflag | cflag;
            if (A.sym_upper) {
                const int md = have_md ? A.max_dist : min(A.ms, A.ns);
                m = m & (d >= 0) & (d <= md);
            }
        } else {
            m = stored & mval;
            if (A.full && A.sym_upper && have_md) {
                const int lim = A.max_dist + max(A.km, A.kn);
                m = m & (d >= 0) & (d <= lim);
            }
        }
        if (!A.full) return m;
    } else {
        if (!A.full) return false;
        if (A.sym_upper && have_md) {
            if (q >= A.ns) m = p >= A.ms - A.max_dist - 2;
            else if (p < 0) m = (q < 0) ? true : (q < A.max_dist + A.kn);
            else m = false;
        } else {
            m = true;
        }
    }
    if (A.sym_upper) {
        const int off = d + (A.kn - A.km);
        const int big_k = max(A.km, A.kn);
        m = m | ((off <= -1) & (off >= -big_k));
    }
    return m;
}

// load one signal pixel as TC (0 outside the matrix / outside the stored band)
template <typename TC>
__device__ __forceinline__ TC load_signal(const CorrArgs<TC>& A, int p, int q)
{
    if ((p < 0) | (p >= A.ms) | (q < 0) | (q >= A.ns)) return TC(0);
    if (A.sig.layout == 2) return (TC)lazy_load(A.sig, p, q);
    long long off = mat_offset(A.sig, p, q);
    if (off < 0) return TC(0);
    if (A.sig_is_f64) return (TC)(((const double*)A.sig.ptr)[off]);
    return (TC)(((const float*)A.sig.ptr)[off]);
}

__device__ __forceinline__ float cs_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double cs_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float cs_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double cs_abs(double x) { return fabs(x); }
__device__ __forceinline__ float cs_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double cs_fma(double a, double b, double c) { return fma(a, b, c); }

// ---------------------------------------------------------------------------------------
// Per-pixel epilogue: window sums -> Pearson coefficient, following SURVEY.md 8(a2)
// (reference detection.py:1000-1107).  Inputs:
//   cs  = sum S * (K' - kmean)      s1 = sum S        s2 = sum S^2
//   nm  = number of missing pixels in the window (0 when no mask)
//   ka, kb = mask-weighted template sums; their definition depends on the precision, see the
//            masked branch below and build_args() in cs_api.cpp
// `mask_branch` selects the reference's missing-mask code path (different but equivalent
// ordering of the denominator); n_obs receives the number of present pixels.
// ---------------------------------------------------------------------------------------
template <typename TC>
__device__ __forceinline__ TC pearson_from_sums(TC cs, TC s1, TC s2, TC nm, TC ka, TC kb,
                                                const KernelStats<TC>& K, bool mask_branch,
                                                TC* n_obs)
{
#include "cs_pearson_body.inc"
}

// The float64 instance, without floating-point contraction: the compiler's choice of which a * b + c it fuses depends on the
// code the function is inlined into (a compile-time-true mask_branch, a neighbouring common subexpression), and the same
// window scored by two kernels -- a block by its own entry or in a batch, the general or the compile-time-size
// wave-per-window function -- then differed by a few ulps.  Unfused, every kernel evaluates the reference's own operations
// (numpy fuses nothing either), and a window has ONE float64 coefficient whichever kernel scores it.
template <>
__device__ __forceinline__ double pearson_from_sums<double>(double cs, double s1, double s2, double nm, double ka, double kb,
                                                            const KernelStats<double>& K, bool mask_branch, double* n_obs)
{
#pragma clang fp contract(off)
    using TC = double;
#include "cs_pearson_body.inc"
}

// Unmasked branch only (reference detection.py:1000-1018, 1213-1220), used by the streaming
// float32 kernel: same thresholds, hardware 1-ulp sqrt / reciprocal instead of the IEEE sequences.
__device__ __forceinline__ float pearson_nomask_f32(float cs, float s1, float s2, const KernelStats<float>& K)
{
    const float m1 = s1 * K.inv_n;
    const float m2 = s2 * K.inv_n;
    const float c = fmaf(K.kmean, s1, cs) * K.inv_n;
    const bool z1 = fabsf(m1) < K.thr;
    const bool z2 = fabsf(m2) < K.thr;
    const bool zc = fabsf(c) < K.thr;
    const float m1z = z1 ? 0.0f : m1;
    const float m2z = z2 ? 0.0f : m2;
    const float num = (z1 | zc) ? ((zc ? 0.0f : c) - m1z * K.kmean) : cs * K.inv_n;
    const float var = fmaf(-m1z, m1z, m2z);
    const float den = __builtin_amdgcn_sqrtf(var) * K.kstd;      // NaN for var < 0
    float r = num * __builtin_amdgcn_rcpf(den);
    r = (fabsf(den) < K.eps) ? 0.0f : r;
    r = (fabsf(r) <= 3.0e38f) ? r : 0.0f;                           // NaN / inf -> 0
    return fminf(fmaxf(r, -1.0f), 1.0f);
}

// Lean float32 epilogues of the streaming kernels.  Away from the reference's zeroing thresholds the
// whole normalisation collapses into one reciprocal square root:
//   no mask:  r = (cs / n) / (sqrt(s2/n - (s1/n)^2) kstd)       = cs * rsq((s2 n - s1^2) kvar)
//   masked:   r = ((cs + s1 ka/np) / np) / sqrt(var_w kvar_w)   = (cs np + s1 ka) * rsq(A B),
//             A = s2 np - s1^2,  B = (n kvar - kb) np - ka^2,   np = n - nm
// (centred mask sums ka, kb as in pearson_from_sums<float>).  A pixel whose window mean, mean square
// or raw correlation is under the 1e-4 threshold -- or, with a mask, whose mask-weighted template
// sums are -- takes the literal per-pixel function instead; such pixels are rare (near-empty
// windows), so the branch is almost never entered by any lane of a wave.
__device__ __forceinline__ float pearson_nomask_f32(float cs, float s1, float s2, const KernelStats<float>& K);
__device__ __forceinline__ float pearson_masked_f32(float cs, float s1, float s2, float nm, float ka, float kb,
                                                    const KernelStats<float>& K);

// ---- candidate mode (KernelStats::cand_cmin > 0) ----------------------------------------------------------------
// cs_detect_foci thresholds a float32 map and re-evaluates the survivors in float64, so the float32 pass must never
// lose a pixel whose exact coefficient reaches the threshold.  A margin below the threshold alone cannot promise that:
// the float32 error of r = cs / sqrt(A B) grows like 1 / conditioning, where the conditioning of a window is
// A / (np s2) -- its variance relative to its mean square -- and that of the template over the present pixels
// B / (np X).  With sums accurate to gamma (gamma <= n 2^-24 for an n-term float32 chain; the float16-pair operands of
// the matrix-core kernels stay inside that, cs_corr_mfma.hip cand_range_guard) the first-order bound is
// |r32 - r64| <= 2 gamma / min(condA, condB).  In candidate mode a pixel therefore keeps its float32 value only when
//   * both conditionings are at least cand_cmin = 8 n 2^-24 / margin, i.e. the error is below margin / 4, AND
//   * that value is below cand_thr = threshold - margin, AND
//   * none of its sums is within 0.1 % of one of the reference's zeroing thresholds (the exact sums may fall on the
//     other side: detection.py:1004-1018, 1051-1064), its denominator is clear of the eps cut (:1088-1101) and its
//     template variance was not snapped to zero;
// every other pixel stores the sentinel 2.0 and becomes a candidate (the compaction keeps values >= cand_thr).  Windows
// without signal (s2 == 0) and windows with too few present pixels (np < cut, exact integer arithmetic) are exactly 0
// in both precisions and stay 0.  No extra transcendental, a handful of compares per pixel.
constexpr float kCandGuard = 1.001f;

__device__ __forceinline__ float cand_screen_nomask(float r, float cs, float s1, float s2, float A, float den2,
                                                    const KernelStats<float>& K)
{
    const float g = K.thr_n * kCandGuard;
    const bool keep = (int)(r < K.cand_thr) & (int)(A > s2 * K.n * K.cand_cmin) & (int)(den2 >= 16.0f * K.den2_min) &
                      (int)(fabsf(s1) >= g) & (int)(s2 >= g) & (int)(fabsf(fmaf(K.kmean, s1, cs)) >= g);
    const float out = keep ? r : 2.0f;
    return (s2 > 0.0f) ? out : 0.0f;
}

__device__ __forceinline__ float cand_screen_masked(float r, float cs, float s1, float s2, float nm, float ka, float kb, float np,
                                                    float A, float B, float den2, const KernelStats<float>& K)
{
    const float c = np * K.cand_cmin;
    const float np2 = np * np;
    const float g = K.thr_n * kCandGuard;
    bool keep = (int)(r < K.cand_thr) & (int)(A > s2 * c) & (int)(B > (K.nkvar - kb) * c) & (int)(den2 >= 16.0f * K.eps2 * np2 * np2) &
                (int)(fabsf(s1) >= g) & (int)(s2 >= g) & (int)(fabsf(fmaf(K.kmean, s1, cs)) >= g);
    if (K.zk_possible) {
        const float km_ = fmaf(K.kmean, nm, ka);
        const float k2m = kb + 2.0f * K.kmean * ka + K.kmean * K.kmean * nm;
        keep &= (bool)((int)(nm < 0.5f) | ((int)(fabsf(km_) >= K.thr * kCandGuard) & (int)(fabsf(k2m) >= K.thr * kCandGuard)));
    }
    const float out = keep ? r : 2.0f;
    return ((int)(s2 > 0.0f) & (int)(np >= K.cut)) ? out : 0.0f;
}

// branch-free cores: coefficient by the one-rsq formula plus "this pixel is near a zeroing threshold".
// CAND: candidate mode decided at run time from K.cand_cmin (-1, default), compiled out (0) or compiled in (1) -- the masked
// matrix-core tile kernel sits at its register budget and instantiates both forms
template <int CAND = -1>
__device__ __forceinline__ float pearson_nomask_core(float cs, float s1, float s2, const KernelStats<float>& K, bool& rare)
{
    const float A = fmaf(s2, K.n, -s1 * s1);
    const float den2 = A * K.kvar;
    float r = cs * __builtin_amdgcn_rsqf(den2);
    r = (den2 >= K.den2_min) ? r : 0.0f;                 // denominator under eps, NaN -> 0
    r = __builtin_amdgcn_fmed3f(r, -1.0f, 1.0f);
    rare = !((int)(fabsf(s1) >= K.thr_n) & (int)(s2 >= K.thr_n) & (int)(fabsf(fmaf(K.kmean, s1, cs)) >= K.thr_n));
    if (CAND > 0 || (CAND < 0 && K.cand_cmin > 0.0f)) {  // wave-uniform
        r = cand_screen_nomask(r, cs, s1, s2, A, den2, K);
        rare = false;                                    // near-threshold pixels already carry the sentinel
    }
    return r;
}

template <int CAND = -1>
__device__ __forceinline__ float pearson_masked_core(float cs, float s1, float s2, float nm, float ka, float kb,
                                                     const KernelStats<float>& K, bool& rare)
{
    const float np = K.n - nm;
    const float num = fmaf(cs, np, s1 * ka);
    const float A = fmaf(s2, np, -s1 * s1);
    float B = fmaf(K.nkvar - kb, np, -ka * ka);
    const float np2 = np * np;
    if (K.snap_possible) B = (B < 1e-5f * K.kvar * np2) ? 0.0f : B;    // wave-uniform branch
    const float den2 = A * B;
    float r = num * __builtin_amdgcn_rsqf(den2);
    r = ((den2 >= K.eps2 * np2 * np2) & (np >= K.cut)) ? r : 0.0f;    // eps, missing_tol cut, NaN -> 0
    r = __builtin_amdgcn_fmed3f(r, -1.0f, 1.0f);
    bool normal = (int)(fabsf(s1) >= K.thr_n) & (int)(s2 >= K.thr_n) & (int)(fabsf(fmaf(K.kmean, s1, cs)) >= K.thr_n);
    if (K.zk_possible) {
        const float km_ = fmaf(K.kmean, nm, ka);
        const float k2m = kb + 2.0f * K.kmean * ka + K.kmean * K.kmean * nm;
        normal &= (bool)((int)(nm < 0.5f) | ((int)(fabsf(km_) >= K.thr) & (int)(fabsf(k2m) >= K.thr)));     // (no short circuits: no branches)
    }
    rare = !normal;
    if (CAND > 0 || (CAND < 0 && K.cand_cmin > 0.0f)) {  // wave-uniform
        r = cand_screen_masked(r, cs, s1, s2, nm, ka, kb, np, A, B, den2, K);
        rare = false;
    }
    return r;
}

// the same screen for the kernels that evaluate pearson_from_sums<float> directly (runtime-size and separable)
__device__ __forceinline__ float cand_upper_from_sums(float r, float cs, float s1, float s2, float nm, float ka, float kb,
                                                      const KernelStats<float>& K, bool masked)
{
    if (!(K.cand_cmin > 0.0f)) return r;
    bool rare;
    return masked ? pearson_masked_core(cs, s1, s2, nm, ka, kb, K, rare) : pearson_nomask_core(cs, s1, s2, K, rare);
}

__device__ __forceinline__ float pearson_nomask_lean(float cs, float s1, float s2, const KernelStats<float>& K)
{
    bool rare;
    float r = pearson_nomask_core(cs, s1, s2, K, rare);
    if (rare) r = pearson_nomask_f32(cs, s1, s2, K);
    return r;
}

__device__ __forceinline__ float pearson_masked_lean(float cs, float s1, float s2, float nm, float ka, float kb,
                                                     const KernelStats<float>& K)
{
    bool rare;
    float r = pearson_masked_core(cs, s1, s2, nm, ka, kb, K, rare);
    if (rare) r = pearson_masked_f32(cs, s1, s2, nm, ka, kb, K);
    return r;
}

// Masked branch for the float32 streaming kernels: the arithmetic of pearson_from_sums<float>
// (same thresholds, same centred mask sums, same snap of a degenerate template variance) with
// selects instead of branches for windows without missing pixels, hardware 1-ulp reciprocal /
// square root instead of the IEEE division and sqrt sequences (three of them per pixel made the
// epilogue as expensive as the FMAs on banded maps), and the generic function only for the rare
// windows whose mask-weighted template sums fall under the 1e-4 threshold.
__device__ __forceinline__ float pearson_masked_f32(float cs, float s1, float s2, float nm, float ka, float kb,
                                                    const KernelStats<float>& K)
{
    const float m1 = s1 * K.inv_n;
    const float m2 = s2 * K.inv_n;
    const float c = fmaf(K.kmean, s1, cs) * K.inv_n;
    const bool z1 = fabsf(m1) < K.thr;
    const bool z2 = fabsf(m2) < K.thr;
    const bool zc = fabsf(c) < K.thr;
    const float m1z = z1 ? 0.0f : m1;
    const float m2z = z2 ? 0.0f : m2;
    const float cz = zc ? 0.0f : c;
    const bool clean = nm < 0.5f;                      // window without missing pixels
    const float np = K.n - nm;
    const float inv_np = clean ? K.inv_n : __builtin_amdgcn_rcpf(np);
    const float km_ = fmaf(K.kmean, nm, ka);
    const float k2m = kb + 2.0f * K.kmean * ka + K.kmean * K.kmean * nm;
    const bool rare = !clean && ((fabsf(km_) < K.thr) || (fabsf(k2m) < K.thr));
    const float a = clean ? 0.0f : ka * inv_np;        // kmean - kmw
    float kvw = fmaf(-a, a, (K.n * K.kvar - kb) * inv_np);
    kvw = (kvw < 1e-5f * K.kvar) ? 0.0f : kvw;
    kvw = clean ? K.kvar : kvw;
    const float ratio = clean ? 1.0f : K.n * inv_np;
    const float m1w = m1z * ratio;
    const float m2w = m2z * ratio;
    float den = __builtin_amdgcn_sqrtf(fmaf(-m1w, m1w, m2w) * kvw);   // NaN for a negative product
    den = (np < K.cut) ? 0.0f : den;
    const float kmw = K.kmean - a;
    const float num = (z1 | zc) ? (cz - m1z * kmw) * ratio : fmaf(s1, a, cs) * inv_np;
    float r = num * __builtin_amdgcn_rcpf(den);
    r = (fabsf(den) < K.eps) ? 0.0f : r;
    r = (fabsf(r) <= 3.0e38f) ? r : 0.0f;               // NaN / inf -> 0
    r = fminf(fmaxf(r, -1.0f), 1.0f);
    if (rare) {
        float nobs;
        r = pearson_from_sums<float>(cs, s1, s2, nm, ka, kb, K, true, &nobs);
    }
    return r;
}

// does output pixel (i, j) exist, and is its value forced to 0?
//   valid-mode margins (detection.py:720-722, 797-801) and sym_upper triu (:1098-1099)
template <typename TC>
__device__ __forceinline__ bool pixel_forced_zero(const CorrArgs<TC>& A, int i, int j)
{
    const int kh = (A.km - 1) / 2, kw = (A.kn - 1) / 2;
    bool z = false;
    // window rows [i-kh, i-kh+km-1] must lie inside the signal (also right for even sizes)
    if (!A.full) z = (i < kh) | (i > A.ms - A.km + kh) | (j < kw) | (j > A.ns - A.kn + kw);
    // triu is applied in framed coordinates when full (frame = (km-1, kn-1))
    if (A.sym_upper) z = z | ((j - i) + (A.full ? (A.kn - A.km) : 0) < 0);
    return z;
}

template <typename TC>
__device__ __forceinline__ void store_pixel(const CorrArgs<TC>& A, int i, int j, TC r, TC nobs)
{
    long long off = mat_offset(A.out, i, j);
    if (off >= 0) {
        if (A.out_is_f64) ((double*)A.out.ptr)[off] = (double)r;
        else ((float*)A.out.ptr)[off] = (float)r;
    }
    if (A.nobs.ptr) {
        long long o2 = mat_offset(A.nobs, i, j);
        if (o2 >= 0) ((float*)A.nobs.ptr)[o2] = (float)nobs;
    }
}

// tile origin of a block.  Dense: plain 2-D tiling.  Band: for row-block by, the x tiles start
// at the tile column containing (i0 + out_lo), clamped to >= 0.
template <typename TC>
__device__ __forceinline__ bool tile_origin(const CorrArgs<TC>& A, int bx, int by, int* i0, int* j0)
{
    *i0 = A.row_begin + by * A.tile_h;
    int jbase = 0;
    if (A.out.layout == 1) {
        int jmin = *i0 + A.out_lo;
        if (jmin < 0) jmin = 0;
        jbase = (jmin / A.tile_w) * A.tile_w;
    }
    *j0 = jbase + bx * A.tile_w;
    if (*i0 >= A.row_end || *j0 >= A.ns) return false;
    if (A.out.layout == 1) {
        // tile intersects the diagonal range [out_lo, out_hi]?
        const int dmax = (*j0 + A.tile_w - 1) - *i0;
        const int dmin = *j0 - (*i0 + A.tile_h - 1);
        if (dmax < A.out_lo || dmin > A.out_hi) return false;
    }
    return true;
}

}  // namespace cs
