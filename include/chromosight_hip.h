/*
 * chromosight_hip.h -- C ABI of the MI355X (gfx950) implementation of chromosight's
 * sliding-window Pearson-correlation hot path.
 *
 * The reference (koszullab/chromosight, pure Python) has no FFI for this path; the
 * entry points below are what a binding for the bodies of the following reference
 * functions would call (all paths relative to /root/reference/chromosight/utils/):
 *
 *   cs_xcorr2_*            detection.py:595  xcorr2  (-> :627 _xcorr2_sparse, :726 _xcorr2_dense)
 *   cs_normxcorr2_*        detection.py:807  normxcorr2 (-> :917 _normxcorr2_sparse, :1134 _normxcorr2_dense)
 *                          incl. the missing-pixel predicate of preprocessing.py:535 make_missing_mask
 *                          and :404 frame_missing_mask, evaluated analytically on the device
 *   cs_distance_law_csr    preprocessing.py:129 distance_law (per-diagonal sum / count reduction)
 *   cs_csr_to_band         preprocessing.py:256 detrend (divide by the law, >= max_val -> 1) fused with
 *                          preprocessing.py:93 diag_trim and the CSR -> dense diagonal-band tiler
 *   cs_detrend_csr         preprocessing.py:256 detrend on the stored values of a CSR/COO matrix
 *   cs_compact_*           the thresholding step of detection.py:387 pick_foci (score >= pearson)
 *   cs_rescore_f64         detection.py:917 evaluated in float64 at a list of pixels
 *                          (quantify mode, detection.py:277, and exact re-scoring of candidates)
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative
 *     cs_status; cs_last_error(ctx) gives a message for the last failure on that context.
 *   - pointers named d_* are DEVICE pointers (HBM); h_* are HOST pointers.  The caller owns
 *     all buffers; the library retains no pointer after a call returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream unless documented otherwise; one in-flight call per
 *     context (re-entrant across contexts).
 *   - matrices: "dense" = row-major, leading dimension `ld` in elements;
 *     "band" = diagonal-offset layout: element (i, j) of an n x n matrix is stored at
 *     d_band[i * ld + (j - i - lo)] for lo <= j - i < lo + width.
 *   - dtype codes: CS_F32 = 0, CS_F64 = 1.
 */
#ifndef CHROMOSIGHT_HIP_H
#define CHROMOSIGHT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_ctx cs_ctx;

typedef enum {
    CS_OK = 0,
    CS_ERR_INVALID = -1,   /* bad argument (the Python shim raises ValueError) */
    CS_ERR_HIP = -2,       /* HIP runtime error */
    CS_ERR_UNSUPPORTED = -3,
    CS_ERR_OVERFLOW = -4,  /* an output buffer was too small; see the call's doc */
    CS_ERR_RANGE = -5      /* cs_normxcorr2_host, cs_normxcorr2 with cs_ctx_set_range_check: the map holds a non-finite pixel,
                              or magnitudes beyond what the float32 kernels square without overflow (1e15): evaluate it in
                              float64 (the Python shim does) */
} cs_status;

enum { CS_F32 = 0, CS_F64 = 1, CS_U8 = 2 /* masks only */ };
enum { CS_LAYOUT_DENSE = 0, CS_LAYOUT_BAND = 1, CS_LAYOUT_BAND_LAZY = 2 /* see cs_stage_block */,
       /* CS_LAYOUT_BAND with a promise: ld >= band_w + 4, the slots band_w .. ld - 1 of every row are ZERO and so is every stored
        * slot whose column lies outside the matrix.  What cs_stage_blocks writes (rows assembled in zeroed pieces of ld slots).
        * The masked tile kernel then fetches the tiles on the rim of the band like the inner ones: a 16-byte piece that reaches
        * beyond a row's stored diagonals reads zeros where a plain band makes it clamp, shift and mask (10 % of the kernel on a
        * 234-diagonal band).  Every other consumer reads it as CS_LAYOUT_BAND. */
       CS_LAYOUT_BAND_PADDED = 3,
       /* A float32 band of RAW COUNTS, zero-padded like CS_LAYOUT_BAND_PADDED, with a header in the CS_COUNTS_HEADER_BYTES in front
        * of d_ptr (weights, distance law and its reciprocals, first genome bin, cap): what cs_stage_blocks writes for a block with
        * band32_counts = 1 -- in the SAME pass over the pixel table that reduces the distance law.  Balancing
        * (contacts_map.py:531-540) and the detrend (preprocessing.py:296-302) are applied by the reader to the pixels it
        * fetches.  The float64 kernels (exact re-scoring, window statistics: through the block's CS_LAYOUT_BAND_LAZY
        * descriptor) use the operations of the staging pass in their order: bit for bit the detrended band.  The masked float32
        * tile kernel behind cs_normxcorr2 / cs_detect_foci* multiplies float32 copies of the weights and reciprocals while it
        * splits a landed tile: within 6 units in the last place of the float32 band (and the float64 expression where a
        * product comes within 2e-6 of the cap, a discrete decision), so its maps and candidate screens differ from those on
        * a CS_LAYOUT_BAND_PADDED band by float32 rounding -- inside cs_foci_params.rescore_margin, which is what decides a
        * candidate (tests/test_gpu_counts_band.py).  band_lo 0, row0 0, CS_F32.  Every other consumer refuses the layout
        * (CS_ERR_INVALID / CS_ERR_UNSUPPORTED). */
       CS_LAYOUT_BAND_COUNTS = 4,
       /* the first band_w diagonals of a wider band of counts (same d_ptr, ld and header): no promise about the slots behind them */
       CS_LAYOUT_BAND_COUNTS_VIEW = 5 };
#define CS_COUNTS_HEADER_BYTES 128
#define CS_COUNTS_LAW_BYTES(n, n_diags) (8ll * (2 * (int64_t)(n_diags) + 2) + 4ll * (((int64_t)(n_diags) + 3) / 2 * 2) + 4ll * (int64_t)(n))
enum { CS_MASK_NONE = 0, CS_MASK_BINS = 1, CS_MASK_EXPLICIT = 2 };

/* ---- context ----------------------------------------------------------------------- */
int cs_ctx_create(int device, cs_ctx** out);
void cs_ctx_destroy(cs_ctx* ctx);
const char* cs_last_error(const cs_ctx* ctx);
/* library / device facts: returns the device's compute-unit count, 0 if ctx is NULL */
int cs_device_cu_count(const cs_ctx* ctx);
const char* cs_version(void);
/* which correlation kernel served the last cs_normxcorr2 / cs_xcorr2 (or the map stage of
 * cs_detect_foci / cs_candidates) on this context: diagnostics, and what the tests use to make sure
 * the intended native path ran */
enum { CS_KERNEL_NONE = 0, CS_KERNEL_GENERIC = 1, CS_KERNEL_STREAM = 2, CS_KERNEL_MFMA = 3, CS_KERNEL_MFMA_DENSE = 4,
       CS_KERNEL_MFMA_REG = 5, CS_KERNEL_SEPARABLE = 6, CS_KERNEL_MFMA_WIDE = 7 };
int cs_last_kernel(const cs_ctx* ctx);
/* Range guard of the device entries (off by default).  The reference sums every window on its own
 * (detection.py:1002-1018) and zeroes exactly the windows that hold a non-finite pixel (:1088-1101); the device kernels
 * keep running box sums and square in float32, so a DEVICE-RESIDENT map handed to cs_normxcorr2 must be finite and, for
 * CS_F32 arithmetic, below 1e15 in magnitude -- PRECONDITION of cs_normxcorr2 (maps prepared by cs_stage_blocks are:
 * detrended, capped, NaN -> 0; cs_normxcorr2_host and the Python surface check their host maps themselves).  With the
 * guard on, cs_normxcorr2 first reduces the rows of `signal` that the call reads (one pass, one stream
 * synchronisation) and refuses a violating map with CS_ERR_RANGE before anything is computed. */
int cs_ctx_set_range_check(cs_ctx* ctx, int32_t on);

/* ---- device memory helpers (so a ctypes caller needs nothing but this library) ------ */
int cs_malloc(cs_ctx* ctx, size_t bytes, void** d_ptr);
int cs_free(cs_ctx* ctx, void* d_ptr);
int cs_memcpy_h2d(cs_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream);
int cs_memcpy_d2h(cs_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream);
int cs_memset(cs_ctx* ctx, void* d_dst, int value, size_t bytes, void* stream);
int cs_stream_sync(cs_ctx* ctx, void* stream);
/* streams and events (hipStream_t / hipEvent_t as void*), for callers without a HIP binding */
int cs_stream_create(cs_ctx* ctx, void** stream);
/* high = 1: a stream whose kernels are dispatched ahead of those of ordinary streams (short latency-bound launch chains --
 * labelling, statistics of a 1-D pattern -- beside persistent tile kernels that would otherwise hold every slot) */
int cs_stream_create_priority(cs_ctx* ctx, int32_t high, void** stream);
int cs_stream_destroy(cs_ctx* ctx, void* stream);
int cs_event_create(cs_ctx* ctx, void** event);
int cs_event_destroy(cs_ctx* ctx, void* event);
int cs_event_record(cs_ctx* ctx, void* event, void* stream);
/* work enqueued on `stream` after this call starts only when `event` (recorded on any stream of the same GPU, by any
 * context) has fired: ordering between streams without a host synchronisation */
int cs_stream_wait_event(cs_ctx* ctx, void* stream, void* event);
/* Work enqueued on `stream` after this call starts only when the tile workgroups of the cs_detect_foci_blocks call on `tiles_ctx`
 * that carries the same EPOCH (cs_foci_params.reserved >> 8 of its entry 0: 1 .. 2^23 - 1, growing from call to call) are resident
 * -- the last workgroup of its tile launch stores the epoch in a word of the context when it starts, one sleeping wave on `stream`
 * waits until the word has reached it -- or after `timeout_us` (at most 5000; the word is a scheduling hint, not a lock: a tile
 * launch that never comes only costs the time-out, and a word left by an earlier launch never lets a later wait through).  Why:
 * a launch chain that waits for the same event as a persistent launch of another stream races it for the workgroup slots, and
 * the persistent workgroups it displaces start late and -- their tile ranges being static -- finish late (a rank's step took 0.6
 * or 0.75 ms depending on which chain won).  Behind this call the chain runs in what the tile workgroups leave, every time
 * (chromosight_amd/plan.py: the 1-D pattern's chain beside the 2-D pattern's tile kernels).  Both contexts on the same GPU. */
int cs_stream_wait_tiles(cs_ctx* ctx, void* stream, cs_ctx* tiles_ctx, int32_t epoch, int32_t timeout_us);
/* synchronises on `stop`, then returns the elapsed milliseconds between the two events */
int cs_event_elapsed_ms(cs_ctx* ctx, void* start, void* stop, float* ms);

/* ---- description of one matrix operand --------------------------------------------- */
typedef struct {
    void* d_ptr;       /* device pointer                                             */
    int32_t dtype;     /* CS_F32 / CS_F64                                            */
    int32_t layout;    /* CS_LAYOUT_DENSE / CS_LAYOUT_BAND (_PADDED)                 */
    int64_t ld;        /* leading dimension in elements                              */
    int32_t band_lo;   /* band layout: first stored diagonal offset (j - i)          */
    int32_t band_w;    /* band layout: number of stored diagonals                    */
    int64_t row0;      /* matrix row stored at d_ptr: the buffer may hold a window of
                          rows of the matrix (0 = from the first row)                */
} cs_matrix;

/* ---- pattern kernel (template) operand --------------------------------------------- */
/* All arrays are HOST pointers to km*kn float64 values, row-major.
 *   h_kernel      the template K; its statistics (sum, mean, std) are the ones used in the
 *                 normalisation (detection.py:1002-1003, 1023-1026)
 *   h_kernel_conv the kernel actually correlated with the signal and with the missing mask;
 *                 NULL = h_kernel.  With --tsvd this is the truncated-SVD reconstruction
 *                 U.V (detection.py:618-619, preprocessing.py:810-847)
 *   h_kernel_sq   the kernel correlated with the mask for the sum-of-squares term; NULL =
 *                 h_kernel squared.  With --tsvd: the reconstruction of K**2 (detection.py:1043)
 */
typedef struct {
    int32_t km, kn;
    const double* h_kernel;
    const double* h_kernel_conv;
    const double* h_kernel_sq;
} cs_kernel;

/* ---- normalised cross-correlation --------------------------------------------------- */
typedef struct {
    int32_t ms, ns;          /* signal shape                                            */
    int32_t full;            /* 1 = 'full' mode: virtual zero frame of (km-1, kn-1)      */
    int32_t sym_upper;       /* 1 = keep j >= i only (detection.py:1098-1099)            */
    int32_t max_dist;        /* -1 = None                                               */
    int32_t mask_mode;       /* CS_MASK_*                                               */
    const uint8_t* d_miss_row; /* CS_MASK_BINS: ms bytes, 1 = bin not detectable         */
    const uint8_t* d_miss_col; /* CS_MASK_BINS: ns bytes                                 */
    const uint8_t* d_mask;   /* CS_MASK_EXPLICIT: missing mask, same layout/ld as signal */
    int32_t min_present;     /* int((1 - missing_tol) * km * kn), detection.py:1069-1072 */
    int32_t compute_dtype;   /* CS_F32 or CS_F64 arithmetic                             */
    double xcorr_threshold;  /* 1e-4: xcorr2's zeroing threshold (detection.py:595,716)  */
    double denom_eps;        /* 1e-10 (detection.py:1088)                                */
    int32_t row_begin, row_end; /* produce the output rows row_begin <= i < row_end only (0, 0 = all;
                                not with CS_MASK_EXPLICIT).
                                The geometry (frame, masks, matrix edges) stays that of the ms x ns
                                matrix; `signal` must hold the rows row_begin - (km-1)/2 ..
                                row_end + (km-1)/2 - 1 that exist (cs_matrix.row0 says where its buffer
                                starts).  Row windows let one map be pipelined over PCIe in slabs and
                                one sub-matrix be split over several GPUs.         */
} cs_normxcorr2_params;

/* Coefficient map.  `signal`, `out_corr` (and `out_nobs` when d_ptr != NULL) share the
 * logical shape ms x ns; layouts may differ (e.g. band in, band out with another range).
 * out_nobs (CS_F32 only) receives the number of present pixels of each window.
 * Pixels outside the stored band of `out_corr` are not written.
 * The call is asynchronous on `stream`.  The template weights and, for CS_MASK_BINS, the per-bin
 * mask tables live in buffers owned by `ctx`, so one context serves one call in flight: use one
 * context per stream / host thread (contexts are independent and cheap).
 * PRECONDITION on device-resident maps (finite; below 1e15 for CS_F32): see cs_ctx_set_range_check. */
int cs_normxcorr2(cs_ctx* ctx, void* stream, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* params, const cs_matrix* out_corr,
                  const cs_matrix* out_nobs);

/* The same map call for a caller whose map lives in HOST memory (what chromosight's Python surface
 * hands over: detection.py:807 takes and returns numpy / scipy containers): float32 or float64 ms x ns
 * map in (sig_dtype, row pitch ld_in elements; float64 rows are rounded to float32 on the device),
 * float32 or float64 coefficient map out (ld_out), no mask, float32 arithmetic.  The map crosses PCIe in row slabs: upload of slab k + 1, the kernel on the row window of
 * slab k (params->row_begin / row_end are set by the call) and the download of slab k - 1 overlap on three
 * streams; the float32 result is widened to float64 on host threads while later slabs are still in
 * flight (the link moves 4 bytes per pixel each way instead of 4 + 8).  Synchronous.  Device staging
 * buffers, the pinned bounce buffer, streams and events belong to the context (grow-only). */
int cs_normxcorr2_host(cs_ctx* ctx, const void* h_signal, int32_t sig_dtype, int64_t ld_in, const cs_kernel* kernel,
                       const cs_normxcorr2_params* params, void* h_out, int32_t out_dtype, int64_t ld_out);

/* Plain cross-correlation: centre-aligned, zero on the (k-1)/2 margins, |v| < threshold -> 0
 * (detection.py:716-722, 797-803).  `h_weights` = km*kn float64 host values. */
int cs_xcorr2(cs_ctx* ctx, void* stream, const cs_matrix* signal, int32_t ms, int32_t ns,
              const double* h_weights, int32_t km, int32_t kn, double threshold,
              int32_t compute_dtype, const cs_matrix* out);

/* Float64 evaluation of the coefficient at `n_px` pixels (d_rows/d_cols int32 device arrays).
 * d_out_corr / d_out_nobs: n_px float64 each (d_out_nobs may be NULL). */
int cs_rescore_f64(cs_ctx* ctx, void* stream, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* params, const int32_t* d_rows,
                   const int32_t* d_cols, int64_t n_px, double* d_out_corr, double* d_out_nobs);

/* ---- thresholded compaction of a coefficient map ------------------------------------ */
/* Appends (row, col, value) of every stored pixel of `corr` (float32 or float64) with
 * value >= threshold, and lo_diag <= col - row <= hi_diag, to the output arrays (capacity
 * `cap`).  *d_count (device int64, caller-zeroed) receives the total number of matches; if it
 * exceeds cap only the first cap are stored and the caller must retry with a larger buffer. */
int cs_compact_ge(cs_ctx* ctx, void* stream, const cs_matrix* corr, int32_t ms, int32_t ns,
                  double threshold, int32_t lo_diag, int32_t hi_diag, int32_t* d_rows,
                  int32_t* d_cols, double* d_vals, int64_t cap, int64_t* d_count);

/* ---- CSR side: distance law, detrend, band tiler ------------------------------------- */
typedef struct {
    int32_t n_rows, n_cols;
    int64_t nnz;
    const int64_t* d_indptr;  /* n_rows + 1 */
    const int32_t* d_indices; /* nnz        */
    const void* d_data;       /* nnz values of `dtype` */
    int32_t dtype;
    /* ---- optional view extensions; all zero / NULL = a plain CSR matrix -------------------------
     * A whole-genome pixel table sorted by (bin1, bin2) (what a .cool stores, contacts_map.py:527
     * create_mat reads it through cooler) IS a CSR of the genome's upper triangle.  One
     * sub-matrix is then a view: d_indptr offset to its first row, col0 = its first column bin,
     * per-row end offsets that stop at its last column / last kept diagonal (cs_csr_band_extent),
     * and the ICE weights of its bins, so that balancing (count * w[bin1] * w[bin2]) happens on the
     * fly and no per-block matrix is ever materialised. */
    int32_t col0;                /* stored column index - col0 = column inside the block          */
    const int64_t* d_row_end;    /* n_rows end offsets; NULL = d_indptr + 1                       */
    const double* d_row_weight;  /* n_rows balancing weights (NaN = bin not detectable) or NULL   */
    const double* d_col_weight;  /* n_cols weights (used only with d_row_weight)                  */
} cs_csr;

/* Per-diagonal sums and counts of the strictly positive stored values whose row and column
 * bins are both detectable, for diagonals 0 <= d < n_diags (upper triangle, col >= row).
 * d_sum / d_cnt: n_diags float64 / int64, overwritten.  The host finishes with
 * law[d] = sum/cnt (NaN when cnt == 0), preprocessing.py:173-188.
 * d_detectable: n_rows bytes (1 = detectable) or NULL (all). */
int cs_distance_law_csr(cs_ctx* ctx, void* stream, const cs_csr* mat, const uint8_t* d_detectable,
                        int32_t n_diags, double* d_sum, int64_t* d_cnt);

/* In-place-capable detrend of the stored values: out[k] = data[k] / law[|row-col|] then
 * (max_val > 0) values >= max_val -> 1 (preprocessing.py:298-309).  d_law: n_law float64
 * (NaN already replaced by 0 by the caller); diagonals >= n_law use law = 0.  d_out has the
 * CSR's dtype and nnz entries. */
int cs_detrend_csr(cs_ctx* ctx, void* stream, const cs_csr* mat, const double* d_law,
                   int32_t n_law, double max_val, void* d_out);

/* CSR -> diagonal band (or dense, by `band->layout`).  Zero-fills the n_rows x ld output then
 * scatters the stored pixels that fall inside it (band: band_lo <= col - row < band_lo + band_w).
 * With d_law != NULL the detrend above is fused into the scatter; NaN results are stored as 0
 * (contacts_map.py:539-540).  Output dtype CS_F32 / CS_F64, or CS_U8 to write a 0/1 mask of the
 * non-zero stored values.  Duplicated entries are not supported (canonical CSR). */
int cs_csr_to_band(cs_ctx* ctx, void* stream, const cs_csr* mat, const double* d_law,
                   int32_t n_law, double max_val, const cs_matrix* band);

/* Per-row entry ranges of the stored pixels with lo_diag <= (col - col0) - row <= hi_diag and
 * 0 <= col - col0 < n_cols (columns sorted within a row): d_begin / d_end, n_rows int64 each, usable
 * as d_indptr / d_row_end of a view on the same arrays.  This is diag_trim (preprocessing.py:93) and
 * the sub-matrix slicing of contacts_map.py:531 without copying a pixel. */
int cs_csr_band_extent(cs_ctx* ctx, void* stream, const cs_csr* mat, int32_t lo_diag, int32_t hi_diag,
                       int64_t* d_begin, int64_t* d_end);

/* law[d] = sum[d] / cnt[d], 0 where cnt[d] == 0 (the NaN of an empty diagonal, replaced by 0 as
 * preprocessing.py:296 does before dividing).  Keeps the distance law on the device between
 * cs_distance_law_csr and cs_csr_to_band. */
int cs_distance_law_finish(cs_ctx* ctx, void* stream, const double* d_sum, const int64_t* d_cnt,
                           int32_t n_diags, double* d_law);

/* ContactMap.create_mat of MANY intra-chromosomal blocks of one genome with three launches (contacts_map.py:527-548,
 * 603-638; preprocessing.py:129-197, 256-310): balance, slice, trim to the diagonals 0 .. keep, per-block distance law,
 * detrend, >= max_val -> 1, NaN -> 0, and the band (or dense map) written once in float64 and / or float32.
 * `genome`: the pixel table of the whole genome as ONE CSR (what a .cool stores: upper triangle sorted by bin1, bin2;
 * n_rows = n_cols = bins, col0 = 0, d_row_end NULL, d_row_weight = d_col_weight = the ICE weights, NaN = undetectable;
 * every stored column must be >= its row).  blocks[b]: rows and columns row0 .. row0 + n - 1; layout CS_LAYOUT_BAND
 * (band_lo 0, band_w = min(keep, n - 1) + 1 diagonals) or CS_LAYOUT_DENSE (n columns); ld >= width, the padding is
 * zeroed; d_band64 / d_band32 (either may be NULL) n * ld elements; d_law min(n, keep + 1) float64 (the law, 0 for an
 * empty diagonal).  Laws longer than 4096 diagonals: CS_ERR_UNSUPPORTED (use the per-block entry points). */
typedef struct {
    int64_t row0;
    int32_t n;
    int32_t keep;
    int32_t layout;
    int32_t band_w;
    int64_t ld;
    void* d_band64;
    void* d_band32;
    double* d_law;
    /* A lazily evaluated float64 band (layout CS_LAYOUT_BAND only; all zero: the plain float64 band above).  The float64
     * consumers of a detect step -- the exact evaluation of the candidates, the windows of the records -- read 1e-3 of a
     * band that costs twice the bytes of the float32 one; with f64_diags > 0 only the diagonals 0 .. f64_diags - 1 are
     * written to d_band64 (n rows of pitch ld64 >= f64_diags, even), and d_lazy (CS_LAZY_BAND_BYTES of device memory)
     * receives a descriptor through which the kernels behind cs_detect_foci_blocks / cs_detect_foci_batch(_templates)
     * recompute any other pixel from the pixel table, the weights and d_law -- the same operations in the same order as the
     * staging pass, bit for bit.  The block is then handed to those entries as a cs_matrix {d_ptr = d_lazy, CS_F64,
     * CS_LAYOUT_BAND_LAZY, ld (unused), band_lo 0, band_w <= min(keep, n - 1) + 1, row0 0}; `genome`'s arrays, d_law and
     * d_band64 must outlive it.  Every other entry refuses the layout (CS_ERR_INVALID). */
    int64_t ld64;
    int32_t f64_diags;
    /* 1: d_band32 receives the block's RAW COUNTS (CS_LAYOUT_BAND_COUNTS above) instead of detrended values, written by the pass
     * that reduces the distance law; the detrend / tiler pass then has nothing to do for this block (none of the blocks: it is not
     * launched).  The caller vouches that every stored count is exact in float32 (integers below 2^24: what a .cool holds) and
     * that no count and no weight is negative (NaN weights: undetectable bins, as everywhere).  Needs: layout CS_LAYOUT_BAND, ld a multiple of 4 and >= band_w + 4, d_band32 16-byte aligned with
     * CS_COUNTS_HEADER_BYTES writable bytes in front of it, d_band64 NULL, f64_diags 0, and d_law of CS_COUNTS_LAW_BYTES(n, n_diags)
     * bytes (the law, then its reciprocals with one slot on either side, then float32 copies of the reciprocals and of the
     * block's weights).  d_lazy (optional) then receives a descriptor that evaluates EVERY pixel of the float64 band from the
     * counts: nothing is searched in the pixel table. */
    int32_t band32_counts;
    void* d_lazy;
} cs_stage_block;
#define CS_LAZY_BAND_BYTES 128
int cs_stage_blocks(cs_ctx* ctx, void* stream, const cs_csr* genome, const cs_stage_block* blocks, int32_t n_blocks,
                    double max_val);

/* Median of the stored values of a CSR view, NaN counted as 0 (np.nanmedian after the NaN -> 0 of
 * contacts_map.py:598-601 preprocess_inter_matrix, which divides an inter-chromosomal block by it).
 * Balanced on the fly like every view.  *h_median = NaN for an empty view.  Synchronous. */
int cs_csr_median(cs_ctx* ctx, void* stream, const cs_csr* mat, double* h_median);
/* The same for n views with two synchronisations in all (sizes, results) instead of two per view: every inter-chromosomal
 * sub-matrix that holds a position of a `quantify --inter` run is scaled by its own median (cli/chromosight.py:229-260 one
 * task per sub-matrix).  h_medians[i] = NaN for an empty view.  Synchronous. */
int cs_csr_median_many(cs_ctx* ctx, void* stream, const cs_csr* mats, int32_t n, double* h_medians);

/* ---- device-side foci: detection.py:387 pick_foci + the statistics of :18 validate_patterns ---- */
typedef struct {
    double pearson;         /* candidate threshold: coefficient >= pearson and != 0 (detection.py:417-421) */
    double rescore_margin;  /* float32 maps: every pixel whose float32 coefficient is >= pearson - margin is
                               re-evaluated in float64 before thresholding -- and so is every pixel whose
                               window is too ill-conditioned for the float32 error to stay below margin / 4
                               (variance of the window relative to its mean square, or of the template over
                               the present pixels, below 8 km kn 2^-24 / margin) or whose sums sit next to one
                               of the reference's zeroing thresholds: foci never depend on float32 rounding */
    int32_t min_size;       /* foci of fewer pixels are dropped (2; detection.py:557 filter_foci)        */
    int32_t diag_only;      /* 1-D patterns: bin1 = bin2 after picking (detection.py:311-315)            */
    int32_t lo_diag, hi_diag; /* scanned diagonals (diag_trim of the coefficient map, :269-270)          */
    int32_t inter;          /* 1: inter-chromosomal block (no NaN sub-diagonals in the windows, :301)    */
    int32_t want_windows;   /* 1: also return the km x kn window of every record                         */
    int32_t exclusive;      /* cs_detect_foci_blocks, entry 0: 1 = nothing else is queued on the device beside
                               this call -- the tiles of all blocks go out as ONE persistent launch (a table of
                               per-block arguments); 0 = one launch per block over several streams, which leaves
                               gaps for the launch chains of other templates running side by side (calls with up
                               to four blocks take the single launch either way)                                */
    int32_t reserved;       /* bit 0: asynchronous form (cs_detect_foci_batch_finish), bit 1: prepare form (cs_detect_foci_blocks),
                               bits 8 ..: tile epoch (cs_stream_wait_tiles); 0 for a plain call */
} cs_foci_params;

typedef struct {
    int32_t bin1, bin2;     /* matrix coordinates of the pattern                                         */
    int32_t inside;         /* 1: its window lies inside the (zero padded) map, strict bounds (:99-104)  */
    int32_t n_zero;         /* zero pixels of the window (missing ones excluded)                          */
    int32_t n_missing;      /* non-finite pixels: missing bins, NaN sub-diagonals                         */
    int32_t focus_size;     /* pixels of the focus (0 in quantify mode)                                   */
    double score;           /* float64 coefficient at (bin1, bin2), untrimmed map                         */
    double n_obs;           /* present pixels of that correlation window                                  */
    double pval;            /* two-sided p-value of `score` under n_obs observations (Fisher z: reference
                               detection.py:332-336 + stats.py:43-81), formed by the kernel that writes the
                               record: cs_accept_records takes it as it is (flags bit 1) instead of spending
                               an atanh, a square root and an erfc per record on the host                   */
} cs_focus;

/* One sub-matrix x one template, `detect` mode, entirely on the device: coefficient map (float32 or
 * float64 arithmetic, context-owned scratch) -> thresholded candidates -> float64 re-scoring ->
 * 4-connected foci (union-find over the sorted candidate list) -> size filter -> first row-major
 * maximum of every focus -> window statistics.  Records come back in the reference's order (foci
 * numbered by the row-major position of their first pixel).  h_foci: `cap` records; h_windows:
 * cap * km * kn float64 (NaN = missing) or NULL.  *n_foci receives the number of foci; if it exceeds
 * cap the call returns CS_ERR_OVERFLOW and the caller retries with larger buffers.  Synchronous.
 * `signal` must stay float64 for bit-identical scores (float32 signals are scored as stored). */
int cs_detect_foci(cs_ctx* ctx, void* stream, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci,
                   int64_t cap, int64_t* n_foci, double* h_windows);

/* 1-D patterns (borders, hairpins: at most 4 scanned diagonals from the main one up) of MANY banded
 * sub-matrices with one launch chain: per sub-matrix such a scan is a few thousand float64 evaluations,
 * i.e. a fixed ~0.1 ms of tiny kernels and a synchronisation, 23 times per template on a human genome.
 * signals / params / foci: arrays of n_blocks (whole blocks, no row windows; the detection parameters
 * pearson, min_size, diag_only, inter, want_windows must agree).  Records (and windows) of all blocks, block
 * after block, go to h_foci / h_windows, which must be page-locked (cs_host_alloc) -- the last kernel
 * writes them; h_n_foci[b] = foci of block b.  CS_ERR_UNSUPPORTED when a block does not qualify (the
 * caller then uses cs_detect_foci per block), CS_ERR_OVERFLOW when cap is too small (counts are set). */
int cs_detect_foci_batch(cs_ctx* ctx, void* stream, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                         const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci, int64_t cap,
                         int64_t* h_n_foci, double* h_windows);

/* cs_detect_foci_batch for SEVERAL templates of one size (n_kernels <= 4: the three borders templates of the reference's
 * borders.json) in the same launch chain: sub-matrix b under template t is handled like block t * n_blocks + b of a
 * batch -- records and windows template by template, h_n_foci[t * n_blocks + b].  What it replaces: one launch chain and
 * one synchronisation per template (reference: the loop over kernels in cli/chromosight.py:731). */
int cs_detect_foci_batch_templates(cs_ctx* ctx, void* stream, int32_t n_blocks, const cs_matrix* signals, int32_t n_kernels,
                                   const cs_kernel* kernels, const cs_normxcorr2_params* params, const cs_foci_params* foci,
                                   cs_focus* h_foci, int64_t cap, int64_t* h_n_foci, double* h_windows);
/* Asynchronous form: with foci[0].reserved & 1, cs_detect_foci_batch_templates (and cs_detect_foci_batch) return as soon as
 * the whole chain is enqueued on `stream` -- records and windows go to the caller's page-locked buffers when it runs -- and
 * cs_detect_foci_batch_finish(ctx, stream, h_n_foci) waits for it and fills the per-(template, block) counts
 * (CS_ERR_OVERFLOW as in the synchronous call).  One call may be pending per context; the caller keeps `stream` and the
 * buffers alive in between.  This is how a 1-D pattern's short, latency-bound kernels are put on the device BEFORE the
 * persistent tile kernels of a 2-D pattern and finished after them (parallel.detect_patterns), instead of racing them from
 * another host thread. */
int cs_detect_foci_batch_finish(cs_ctx* ctx, void* stream, int64_t* h_n_foci);

/* The same for ANY pattern: blocks whose scan is a band of diagonals (loops, stripes) run the masked matrix-core tile
 * kernel in candidate mode -- every block's kernel appends its candidate pixels to one list -- followed by one sort, one
 * float64 re-scoring, one labelling workgroup per block and one statistics pass for all blocks; batches of 1-D patterns
 * are passed on to cs_detect_foci_batch.  signals: the float64 maps (exact scores, windows); signals_f32 (may be NULL,
 * entries may have a NULL d_ptr): the same maps in float32 with the same layout, what the tile kernel stages (a float64
 * map is rounded into context scratch first when it has no float32 twin).  Per-bin masks, float32 arithmetic, whole
 * blocks; CS_ERR_UNSUPPORTED when a block is not served by the tile kernel (template size, layout): the caller then
 * uses cs_detect_foci block by block.  Outputs as cs_detect_foci_batch.
 * PRECONDITION on the flags: params[b].miss_row / miss_col must be COMPLETE in device memory when the call is made,
 * independently of `stream` -- the mask tables that depend only on them and on the template are built on the context's side
 * streams as soon as the call is entered, beside whatever `stream` is still running (the staging of the maps); only the
 * maps themselves are ordered behind `stream`.  A caller that produces the flags on `stream` synchronises it (or waits for
 * an event of it) first.  While an asynchronous batch is pending on the context (cs_detect_foci_batch_finish not yet
 * called) every foci / quantify entry on it answers CS_ERR_INVALID: they share its count words, block tables and pool.
 * PREPARE form (foci[0].reserved & 2): everything of the call that does not depend on the maps -- the blocks' mask tables, the
 * zeroed candidate counters, the argument tables of the tile launch and of the chain behind it -- is enqueued on the context's
 * side streams and the call returns CS_OK at once (nothing is written to the outputs); the SAME call without the flag, made
 * next on this context with the same arguments, finds that work done (anything else in between, or other arguments, and it
 * simply does everything itself).  A caller that stages the maps on `stream` makes the prepare form BEFORE the staging: the
 * tile launch behind the staging then waits for side streams that finished long ago -- a stream that reaches a wait for an
 * event that has not fired yet loses 20-40 us to it. */
int cs_detect_foci_blocks(cs_ctx* ctx, void* stream, int32_t n_blocks, const cs_matrix* signals, const cs_matrix* signals_f32,
                          const cs_kernel* kernel, const cs_normxcorr2_params* params, const cs_foci_params* foci,
                          cs_focus* h_foci, int64_t cap, int64_t* h_n_foci, double* h_windows);

/* `quantify` mode (detection.py:277, 297-298): score, n_obs and window statistics at n given pixels
 * (host int32 arrays).  h_out: n records in input order; h_windows: n * km * kn float64 or NULL. */
int cs_quantify_pixels(cs_ctx* ctx, void* stream, const cs_matrix* signal, const cs_kernel* kernel,
                       const cs_normxcorr2_params* params, const cs_foci_params* foci,
                       const int32_t* h_rows, const int32_t* h_cols, int64_t n, cs_focus* h_out,
                       double* h_windows);

/* The same over the sub-matrices of a genome in ONE launch chain (`quantify` scores one sub-matrix per pool task,
 * cli/chromosight.py:229-260, 396-410: hundreds of tasks of a few positions each): position t is pixel (h_rows[t], h_cols[t])
 * of sub-matrix h_blk[t]; signals / params / foci hold n_blocks entries (float64 arithmetic; foci[b].inter says whether
 * sub-matrix b is inter-chromosomal; want_windows from entry 0).  h_out: n records in input order, h_windows: n * km * kn
 * float64 or NULL.  Synchronous. */
int cs_quantify_blocks(cs_ctx* ctx, void* stream, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                       const cs_normxcorr2_params* params, const cs_foci_params* foci, const int32_t* h_blk,
                       const int32_t* h_rows, const int32_t* h_cols, int64_t n, cs_focus* h_out, double* h_windows);

/* One sub-matrix split over several GPUs by row windows: pick_foci (detection.py:387-592)
 * needs the thresholded pixels of the whole map, so the two halves of cs_detect_foci are exposed.
 *
 * cs_candidates: the pixels of the rows params->row_begin <= i < row_end of the coefficient map that
 * pass the exact (float64) threshold foci->pearson inside [lo_diag, hi_diag], in row-major order: int32
 * coordinates and float64 coefficients into host arrays of `cap` entries; CS_ERR_OVERFLOW with *n set if
 * cap is too small.  `signal` holds the window's rows plus the (km-1)/2 halo rows on either side.
 *
 * cs_label_foci: the 4-connected foci of a candidate list (the concatenation of every window's
 * candidates, any order): coordinates of each focus of >= min_size pixels at its maximum (row = col when
 * diag_only) and its size, in row-major order of the maxima's first pixels, exactly the foci cs_detect_foci
 * finds on the whole map.  Their records follow from cs_quantify_pixels on the window that owns the row. */
int cs_candidates(cs_ctx* ctx, void* stream, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* params, const cs_foci_params* foci, int32_t* h_rows,
                  int32_t* h_cols, double* h_vals, int64_t cap, int64_t* n);
int cs_label_foci(cs_ctx* ctx, void* stream, int32_t ms, int32_t ns, const int32_t* h_rows,
                  const int32_t* h_cols, const double* h_vals, int64_t n, int32_t min_size, int32_t diag_only,
                  int32_t* h_foci_rows, int32_t* h_foci_cols, int32_t* h_foci_size, int64_t cap, int64_t* n_foci);

/* ---- a genome step as ONE native call -------------------------------------------------------------------------------
 * The entries above are what a detect step is made of -- cs_stage_blocks, an event, cs_detect_foci_blocks for the 2-D
 * pattern, cs_detect_foci_batch_templates for a 1-D pattern's templates on another context, cs_accept_records on each
 * result -- and on a rank's share of a genome the interpreter between them costs as much as their kernels (reference:
 * cli/chromosight.py:738-755, the same calls per sub-matrix task).  cs_run_calls runs a list of such calls natively: the
 * calls of lane 0 on the calling thread, in order, the calls of lane k > 0 in order on a worker thread of the library
 * (kept between calls), `after` >= 0 making a call wait until call number `after` of the list has returned (how a lane
 * waits for another lane's event record).  The argument slots are the entry's own arguments in order: pointers in p[],
 * integers in i[], doubles in d[] (chromosight_amd/plan.py builds them from the arguments of a step that ran the usual way
 * and replays them on the same buffers).  rc of every call is filled in; a lane stops at its first failing call; the
 * return value is the first non-zero rc (0: every call succeeded). */
enum { CS_CALL_STAGE_BLOCKS = 1, CS_CALL_EVENT_RECORD = 2, CS_CALL_STREAM_WAIT_EVENT = 3, CS_CALL_DETECT_FOCI_BLOCKS = 4,
       CS_CALL_DETECT_FOCI_BATCH_TEMPLATES = 5, CS_CALL_ACCEPT_RECORDS = 6, CS_CALL_DETECT_FOCI_BATCH_FINISH = 7,
       CS_CALL_STREAM_WAIT_TILES = 9 /* cs_stream_wait_tiles(p[0], p[1], p[2], i[0], i[1]) */ };
typedef struct {
    int32_t fn;        /* CS_CALL_* */
    int32_t lane;      /* 0: the calling thread */
    int32_t after;     /* index of a call that must have returned first, or -1 */
    int32_t rc;        /* out */
    void* p[12];
    int64_t i[6];
    double d[2];
} cs_call;
int cs_run_calls(cs_call* calls, int32_t n_calls);

/* ---- the exchange step of the sharded path on RCCL (one process per GPU) --------------------------------
 * The path shards over independent sub-matrices like the reference's Pool.imap (cli/chromosight.py:748-752); ranks only
 * exchange pattern records (variable-length lists of fixed-size float64 records) and small float64 vectors (the pileup of
 * an iterated template, :791; the per-diagonal sums of a sub-matrix split over ranks).  Host arrays in and out -- the
 * records are host data on both sides -- device staging and the collectives (ncclAllGather / ncclAllReduce over xGMI)
 * inside.  librccl is loaded on first use; CS_ERR_UNSUPPORTED when it is absent.
 *
 * cs_comm_unique_id: 128 bytes to be created on ONE rank and handed to the others by whatever launched them (the Python
 * side broadcasts them over the torch.distributed store).  cs_comm_create: collective over all ranks.
 * cs_comm_allgather_rows: every rank passes n_rows rows of `width` float64; h_out receives the rows of rank 0, 1, ...
 * in rank order (h_counts[r] rows of rank r), identical on every rank; CS_ERR_OVERFLOW (counts set) when the rows do not
 * fit the SMALLEST cap_rows any rank passed -- the capacities travel with the counts, so every rank gets the same
 * answer and all of them call again (with room for the sum of the counts): the collectives of a communicator stay
 * matched however unevenly the rows are spread.  cs_comm_allgather_rows_once: the same result from ONE collective -- a rank's
 * block is slot_rows + 1 rows, the first of which carries its count; CS_ERR_OVERFLOW (counts set, on every rank alike) when a
 * list is longer than the slot or the total exceeds cap_rows: call again with a slot for the longest list.  For exchanges
 * that repeat with similar counts (the steps of a sharded run).  cs_comm_allreduce_f64: element-wise sum over the ranks, in
 * place. */
typedef struct cs_comm cs_comm;
int cs_comm_available(void);            /* CS_OK when the RCCL of the library's HIP runtime loads (a local check: no rank talks) */
int cs_comm_unique_id(void* out128);
int cs_comm_create(int device, int rank, int world, const void* unique_id128, cs_comm** out);
void cs_comm_destroy(cs_comm* comm);
const char* cs_comm_last_error(const cs_comm* comm);
int cs_comm_rank(const cs_comm* comm);
int cs_comm_world(const cs_comm* comm);
int cs_comm_allgather_rows(cs_comm* comm, const double* h_rows, int64_t n_rows, int32_t width, double* h_out,
                           int64_t cap_rows, int64_t* h_counts);
int cs_comm_allgather_rows_once(cs_comm* comm, const double* h_rows, int64_t n_rows, int32_t width, int64_t slot_rows,
                                double* h_out, int64_t cap_rows, int64_t* h_counts);
int cs_comm_allreduce_f64(cs_comm* comm, double* h_values, int64_t n);

/* Greedy neighbour suppression of detection.py:348 remove_neighbours, on the host, in O(n) with a
 * grid of win x win cells instead of the reference's O(n^2) scan: patterns are visited in `h_order`
 * (indices by decreasing score, as the caller sorted them); a pattern is kept iff no pattern kept
 * before it lies closer than `win` bins on both axes.  h_keep: n bytes, 1 = kept. */
int cs_remove_neighbours(const int64_t* h_bin1, const int64_t* h_bin2, const int64_t* h_order, int64_t n,
                         int64_t win, uint8_t* h_keep);

/* The acceptance rules and p-values that follow a detection call, on its HOST records (reference
 * detection.py:121-141 validate_patterns' window rules; :269-270 coefficient read on the trimmed map; :332-336 +
 * stats.py:43-81 corr_to_pval on the untrimmed one): plain C++ over a few thousand records, here because the numpy
 * version cost more than the device work that produced them.  h_rec: the records of n_blocks sub-matrices one after
 * the other (h_counts[b] each; sub-matrix b has h_rows[b] x h_cols[b] bins and was scanned up to diagonal
 * h_max_dist[b], < 0 or NULL: no limit).  inter: no diagonal rules.  full = 0: every window has km * kn
 * observations.  h_table: 4 doubles (bin1, bin2, score, p-value) per record -- flags bit 0 set (detect mode): accepted
 * records only, packed; clear (quantify mode): every record, rejected ones with a NaN score.  flags bit 1: the records
 * carry their p-values (cs_focus.pval: records written by this library's device entries do; records assembled by the
 * caller need not) -- they are copied instead of computed.  h_ok: one byte per input record; h_kept[b]: accepted records
 * of sub-matrix b. */
int cs_accept_records(const cs_focus* h_rec, int64_t n_blocks, const int64_t* h_counts, const int32_t* h_rows,
                      const int32_t* h_cols, const int32_t* h_max_dist, int32_t inter, int32_t km, int32_t kn,
                      double missing_tol, double zero_tol, int32_t full, int32_t flags, double* h_table, uint8_t* h_ok,
                      int64_t* h_kept);

/* ---- pinned host memory (PCIe side of the boundary: page-locked buffers copy at link speed) ---- */
int cs_host_alloc(cs_ctx* ctx, size_t bytes, void** h_ptr);
int cs_host_free(cs_ctx* ctx, void* h_ptr);

#ifdef __cplusplus
}
#endif
#endif /* CHROMOSIGHT_HIP_H */
