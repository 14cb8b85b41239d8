// cs_aux.hip -- HBM-bound helper kernels either side of the correlation:
//   * per-diagonal distance-law reduction over CSR      (reference preprocessing.py:129-197)
//   * detrend of CSR values                             (reference preprocessing.py:256-310)
//   * CSR -> diagonal-band tiler with fused detrend     (+ diag_trim, preprocessing.py:93-126)
//   * threshold compaction of a coefficient map         (first step of pick_foci, detection.py:417-421)
//   * float64 re-scoring of a list of pixels            (detection.py:917-1131 evaluated per pixel)
#include <algorithm>

#include "cs_device.h"
#include "cs_launch_aux.h"

namespace cs {

// ------------------------------------------------------------------------------------------
// distance law: sum / count of the strictly positive pixels of each diagonal whose two bins are
// detectable.  One wave per CSR row, lanes stride over the row's stored entries (coalesced);
// per-block partial sums live in LDS (ds_add_f64), flushed with one global atomic per diagonal.
// ------------------------------------------------------------------------------------------
// Blocks of 16 waves, one block per CU: every block ends with one global atomic per diagonal, and the
// atomics of all blocks on one diagonal serialise in L2 -- with 8 four-wave blocks per CU that tail
// was most of the kernel (91 us for the 10 M pixels of C3, 1 TB/s).
constexpr int kLawThreads = 1024;

template <typename TV, bool USE_LDS>
__global__ __launch_bounds__(kLawThreads) void distance_law_kernel(CsrView M, const uint8_t* __restrict__ det,
                                                           int n_diags, double* __restrict__ g_sum,
                                                           unsigned long long* __restrict__ g_cnt,
                                                           int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* l_sum = reinterpret_cast<double*>(smem_raw);
    unsigned int* l_cnt = reinterpret_cast<unsigned int*>(smem_raw + sizeof(double) * (size_t)n_diags);
    const int tid = threadIdx.x;
    if (USE_LDS) {
        for (int d = tid; d < n_diags; d += kLawThreads) {
            l_sum[d] = 0.0;
            l_cnt[d] = 0u;
        }
        __syncthreads();
    }
    const int lane = tid & 63, wv = tid >> 6;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(r_begin + rows_per_block, M.n_rows);
    const TV* __restrict__ data = reinterpret_cast<const TV*>(M.data);
    for (int row = r_begin + wv; row < r_end; row += kLawThreads / 64) {
        if (det && !det[row]) continue;
        const long long b = M.indptr[row], e = M.row_end[row];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = M.indices[k] - M.col0;
            const int d = col - row;
            if (d < 0 || d >= n_diags || col >= M.n_cols) continue;
            if (det && !det[col]) continue;
            const double v = csr_value(M, data, k, row, col);
            if (!(v > 0.0)) continue;  // also drops NaN (preprocessing.py:188)
            if (USE_LDS) {
                atomicAdd(&l_sum[d], v);
                atomicAdd(&l_cnt[d], 1u);
            } else {
                atomicAdd(&g_sum[d], v);
                atomicAdd(&g_cnt[d], 1ull);
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int d = tid; d < n_diags; d += kLawThreads) {
            if (l_cnt[d]) {
                atomicAdd(&g_sum[d], l_sum[d]);
                atomicAdd(&g_cnt[d], (unsigned long long)l_cnt[d]);
            }
        }
    }
}

int launch_distance_law(const CsrView& M, const uint8_t* det, int n_diags, double* d_sum,
                        long long* d_cnt, int n_cu, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(double) * (size_t)n_diags, stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d_cnt, 0, sizeof(long long) * (size_t)n_diags, stream);
    if (e != hipSuccess) return (int)e;
    if (M.n_rows == 0 || M.nnz == 0 || n_diags == 0) return 0;
    const size_t smem = (sizeof(double) + sizeof(unsigned int)) * (size_t)n_diags;
    const bool use_lds = smem <= 64 * 1024;
    int blocks = min((M.n_rows + 15) / 16, n_cu);
    if (blocks < 1) blocks = 1;
    const int rows_per_block = (M.n_rows + blocks - 1) / blocks;
    blocks = (M.n_rows + rows_per_block - 1) / rows_per_block;
    auto cnt = reinterpret_cast<unsigned long long*>(d_cnt);
#define CS_DL(TV, L)                                                                             \
    hipLaunchKernelGGL((distance_law_kernel<TV, L>), dim3(blocks), dim3(kLawThreads), (L) ? smem : 0, \
                       stream, M, det, n_diags, d_sum, cnt, rows_per_block)
    if (M.is_f64) {
        if (use_lds) {
            if (smem > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)distance_law_kernel<double, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            CS_DL(double, true);
        } else {
            CS_DL(double, false);
        }
    } else {
        if (use_lds) {
            if (smem > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)distance_law_kernel<float, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            CS_DL(float, true);
        } else {
            CS_DL(float, false);
        }
    }
#undef CS_DL
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// detrend value: v / law[|d|], then >= max_val -> 1 (NaN stays NaN; the band writer turns it
// into 0 as contacts_map.py:539-540 does after trimming)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double detrend_value(double v, int d, const double* __restrict__ law,
                                                int n_law, double max_val)
{
    const int ad = d < 0 ? -d : d;
    const double y = (ad < n_law) ? law[ad] : 0.0;
    double out = v / y;
    if (max_val > 0.0 && out >= max_val) out = 1.0;
    return out;
}

template <typename TV>
__global__ __launch_bounds__(256) void detrend_csr_kernel(CsrView M, const double* __restrict__ law,
                                                          int n_law, double max_val, TV* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TV* __restrict__ data = reinterpret_cast<const TV*>(M.data);
    for (int row = blockIdx.x * 4 + wv; row < M.n_rows; row += gridDim.x * 4) {
        const long long b = M.indptr[row], e = M.row_end[row];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = M.indices[k] - M.col0;
            out[k] = (TV)detrend_value(csr_value(M, data, k, row, col), col - row, law, n_law, max_val);
        }
    }
}

int launch_detrend_csr(const CsrView& M, const double* law, int n_law, double max_val, void* out,
                       int n_cu, hipStream_t stream)
{
    if (M.n_rows == 0 || M.nnz == 0) return 0;
    int blocks = min((M.n_rows + 3) / 4, n_cu * 16);
    if (M.is_f64)
        hipLaunchKernelGGL(detrend_csr_kernel<double>, dim3(blocks), dim3(256), 0, stream, M, law, n_law,
                           max_val, (double*)out);
    else
        hipLaunchKernelGGL(detrend_csr_kernel<float>, dim3(blocks), dim3(256), 0, stream, M, law, n_law,
                           max_val, (float*)out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// CSR -> band (or dense) scatter: one wave per row; a row's stored entries are contiguous in the
// band, so both the CSR reads and the band writes are coalesced runs.  TB = uint8_t writes a
// 0/1 mask (explicit missing masks).
// ------------------------------------------------------------------------------------------
template <typename TV, typename TB>
__global__ __launch_bounds__(256) void csr_to_band_kernel(CsrView M, const double* __restrict__ law,
                                                          int n_law, double max_val, MatView band)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TV* __restrict__ data = reinterpret_cast<const TV*>(M.data);
    TB* __restrict__ dst = reinterpret_cast<TB*>(band.ptr);
    for (int row = blockIdx.x * 4 + wv; row < M.n_rows; row += gridDim.x * 4) {
        const long long b = M.indptr[row], e = M.row_end[row];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = M.indices[k] - M.col0;
            if (col < 0 || col >= M.n_cols) continue;
            const long long off = mat_offset(band, row, col);
            if (off < 0) continue;
            double v = csr_value(M, data, k, row, col);
            if (law) v = detrend_value(v, col - row, law, n_law, max_val);
            if (v != v) v = 0.0;
            if constexpr (sizeof(TB) == 1) dst[off] = (v != 0.0) ? 1 : 0;
            else dst[off] = (TB)v;
        }
    }
}

// Band outputs: every element of the band is written exactly once, with no zero-fill pass before the
// scatter (for a 200 000-bin block with 1018 float64 diagonals that pass alone moved 1.6 GB).  The
// columns of a CSR row are sorted, so the lane that holds stored pixel k writes its value at
// x_k = col_k - row - lo and the zeros of the gap up to the next stored pixel (up to the row pitch after
// the last one; the lane of the first pixel also writes the leading gap).  Near the diagonal rows are
// dense and the gaps are empty: the stores of a wave are one contiguous run.
template <typename TV, typename TB>
__global__ __launch_bounds__(256) void csr_to_band_rows_kernel(CsrView M, const double* __restrict__ law, int n_law,
                                                               double max_val, MatView band)
{
    // a row is a chain of dependent global round trips (row pointers -> columns / values -> store), so
    // every wave works on R rows at once to keep R chains in flight
    constexpr int R = 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ld = (int)band.ld, W = band.band_w;
    const TV* __restrict__ data = reinterpret_cast<const TV*>(M.data);
    TB* __restrict__ dst = reinterpret_cast<TB*>(band.ptr);
    const int stride = gridDim.x * 4;
    auto slot = [&](long long k, int row) {      // band slot of stored pixel k, clamped to [-1, W]
        const int col = M.indices[k] - M.col0;
        long long x = (long long)col - row - band.band_lo;
        if (col < 0) x = -1;
        if (col >= M.n_cols) x = W;
        return (int)(x < -1 ? -1 : (x > W ? W : x));
    };
    for (int row0 = blockIdx.x * 4 + wv; row0 < M.n_rows; row0 += R * stride) {
        long long b[R], e[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r * stride;
            const bool live = row < M.n_rows;
            b[r] = live ? M.indptr[row] : 0;
            e[r] = live ? M.row_end[row] : 0;
        }
        long long len = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            len = max(len, e[r] - b[r]);
            const int row = row0 + r * stride;
            if (row < M.n_rows && e[r] == b[r])                        // empty row: all zeros
                for (int x = lane; x < ld; x += 64) dst[(size_t)row * ld + x] = TB(0);
        }
        for (long long o = lane; o < len; o += 64) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long k = b[r] + o;
                if (k >= e[r]) continue;
                const int row = row0 + r * stride;
                TB* out = dst + (size_t)row * ld;
                const int x = slot(k, row);
                const int nx = (k + 1 < e[r]) ? slot(k + 1, row) : ld;
                if (x >= 0 && x < W) {
                    const int col = M.indices[k] - M.col0;
                    double v = csr_value(M, data, k, row, col);
                    if (law) v = detrend_value(v, col - row, law, n_law, max_val);
                    if (v != v) v = 0.0;
                    if constexpr (sizeof(TB) == 1) out[x] = (v != 0.0) ? 1 : 0;
                    else out[x] = (TB)v;
                }
                const int z1 = (k + 1 < e[r]) ? min(nx, W) : ld;      // zeros up to the next stored slot / the pitch
                for (int z = (x >= W ? W : x + 1); z < z1; ++z) out[z] = TB(0);
                if (k == b[r])
                    for (int z = 0; z < min(x, W); ++z) out[z] = TB(0);   // leading gap
            }
        }
    }
}

int launch_csr_to_band(const CsrView& M, const double* law, int n_law, double max_val,
                       const MatView& band, int band_dtype, int n_cu, hipStream_t stream)
{
    const size_t esz = band_dtype == 1 ? 8 : (band_dtype == 2 ? 1 : 4);
    if (band.layout == 1 && M.n_rows > 0) {
        const size_t smem = 0;
        int blocks = std::max(1, std::min((M.n_rows + 7) / 8, n_cu * 8));
#define CS_C2R(TV, TB)                                                                                                  \
    {                                                                                                                   \
        if (smem > 48 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)csr_to_band_rows_kernel<TV, TB>,                                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
        hipLaunchKernelGGL((csr_to_band_rows_kernel<TV, TB>), dim3(blocks), dim3(256), smem, stream, M, law, n_law,       \
                           max_val, band);                                                                              \
    }
        if (M.is_f64) {
            if (band_dtype == 1) CS_C2R(double, double)
            else if (band_dtype == 2) CS_C2R(double, uint8_t)
            else CS_C2R(double, float)
        } else {
            if (band_dtype == 1) CS_C2R(float, double)
            else if (band_dtype == 2) CS_C2R(float, uint8_t)
            else CS_C2R(float, float)
        }
#undef CS_C2R
        return (int)hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(band.ptr, 0, esz * (size_t)band.ld * (size_t)M.n_rows, stream);
    if (e != hipSuccess) return (int)e;
    if (M.n_rows == 0 || M.nnz == 0) return 0;
    int blocks = min((M.n_rows + 3) / 4, n_cu * 16);
#define CS_C2B(TV, TB) \
    hipLaunchKernelGGL((csr_to_band_kernel<TV, TB>), dim3(blocks), dim3(256), 0, stream, M, law, n_law, max_val, band)
    if (M.is_f64) {
        if (band_dtype == 1) CS_C2B(double, double);
        else if (band_dtype == 2) CS_C2B(double, uint8_t);
        else CS_C2B(double, float);
    } else {
        if (band_dtype == 1) CS_C2B(float, double);
        else if (band_dtype == 2) CS_C2B(float, uint8_t);
        else CS_C2B(float, float);
    }
#undef CS_C2B
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// thresholded compaction: append (row, col, value) of every stored pixel >= threshold
// ------------------------------------------------------------------------------------------
template <typename TV>
__global__ __launch_bounds__(256) void compact_ge_kernel(MatView corr, int ms, int ns, double threshold,
                                                         int lo_diag, int hi_diag, int* __restrict__ rows,
                                                         int* __restrict__ cols, double* __restrict__ vals,
                                                         long long cap, unsigned long long* __restrict__ count)
{
    // the buffer holds the rows corr.row0 .. ms - 1 of the map
    const int width = (corr.layout == 1) ? corr.band_w : ns;
    const long long total = (long long)(ms - corr.row0) * width;
    const TV* __restrict__ src = reinterpret_cast<const TV*>(corr.ptr);
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // every lane runs the same number of iterations so that the wave-level ballot is well defined
    const long long n_iter = (total + stride - 1) / stride;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long it = 0; it < n_iter; ++it, idx += stride) {
        bool hit = false;
        int i = 0, j = 0;
        double v = 0.0;
        if (idx < total) {
            const long long li = idx / width;
            i = (int)(li + corr.row0);
            const int x = (int)(idx - li * width);
            j = (corr.layout == 1) ? (i + corr.band_lo + x) : x;
            if (j >= 0 && j < ns) {
                const int d = j - i;
                v = (double)src[li * corr.ld + x];
                hit = (d >= lo_diag) && (d <= hi_diag) && (v >= threshold) && (v != 0.0);
            }
        }
        const unsigned long long ballot = __ballot(hit);
        if (ballot) {
            const int n_hit = __popcll(ballot);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned long long)n_hit);
            base = __shfl(base, 0);
            if (hit) {
                const unsigned long long pos = base + __popcll(ballot & ((1ull << lane) - 1ull));
                if ((long long)pos < cap) {
                    rows[pos] = i;
                    cols[pos] = j;
                    vals[pos] = v;
                }
            }
        }
    }
}

int launch_compact_ge(const MatView& corr, int corr_is_f64, int ms, int ns, double threshold,
                      int lo_diag, int hi_diag, int* rows, int* cols, double* vals, long long cap,
                      long long* count, int n_cu, hipStream_t stream)
{
    const int width = (corr.layout == 1) ? corr.band_w : ns;
    const long long total = (long long)(ms - corr.row0) * width;
    if (total <= 0) return 0;
    long long want = (total + 255) / 256;
    int blocks = (int)(want < (long long)n_cu * 8 ? want : (long long)n_cu * 8);
    auto cnt = reinterpret_cast<unsigned long long*>(count);
    if (corr_is_f64)
        hipLaunchKernelGGL(compact_ge_kernel<double>, dim3(blocks), dim3(256), 0, stream, corr, ms, ns,
                           threshold, lo_diag, hi_diag, rows, cols, vals, cap, cnt);
    else
        hipLaunchKernelGGL(compact_ge_kernel<float>, dim3(blocks), dim3(256), 0, stream, corr, ms, ns,
                           threshold, lo_diag, hi_diag, rows, cols, vals, cap, cnt);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// float64 evaluation of the coefficient at a list of pixels: one wave per pixel, the lanes stride
// over the window (read straight from HBM/L2 -- candidate lists are tiny compared with the map),
// fixed-order butterfly reduction.  Lists are short (tens to thousands of pixels), so one lane per
// pixel left the chip empty and a call cost the latency of 289 dependent loads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rescore_f64_kernel(const CorrArgs<double> A, const int* __restrict__ rows,
                                                          const int* __restrict__ cols, long long n_px,
                                                          double* __restrict__ out_corr,
                                                          double* __restrict__ out_nobs,
                                                          const long long* __restrict__ n_ptr)
{
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_px || (n_ptr && t >= *n_ptr)) return;
    double r, nobs;
    rescore_pixel(A, rows[t], cols[t], lane, r, nobs);
    if (lane == 0) {
        out_corr[t] = r;
        if (out_nobs) out_nobs[t] = nobs;
    }
}

// The same from a list of pixel KEYS (row * ns + col, what a tile kernel's candidate sink appends): decoded here (rows[], cols[]
// written for the host), the device's count copied next to them (count_copy, so that ONE download carries count, pixels and
// scores) and the counter of the NEXT call cleared (zero_next: two counters alternate, no memset in the chain).
__global__ __launch_bounds__(256) void rescore_f64_keys_kernel(const CorrArgs<double> A, const long long* __restrict__ keys, int ns,
                                                               long long n_px, int* __restrict__ rows, int* __restrict__ cols,
                                                               double* __restrict__ out_corr, const long long* __restrict__ n_ptr,
                                                               long long* __restrict__ count_copy, long long* __restrict__ zero_next)
{
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n = *n_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *count_copy = n;
        if (zero_next) *zero_next = 0;
    }
    if (t >= n_px || t >= n) return;
    const long long k = keys[t];
    const int i = (int)(k / ns), j = (int)(k - (k / ns) * ns);
    double r, nobs;
    rescore_pixel(A, i, j, lane, r, nobs);
    if (lane == 0) {
        rows[t] = i;
        cols[t] = j;
        out_corr[t] = r;
    }
}

__global__ __launch_bounds__(256) void rescore_f64_run_kernel(const CorrArgs<double> A, const int* __restrict__ rows,
                                                              const int* __restrict__ cols, long long n_px,
                                                              double* __restrict__ out_corr, double* __restrict__ out_nobs,
                                                              const long long* __restrict__ n_ptr)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_px || (n_ptr && t >= *n_ptr)) return;
    double r, nobs;
    rescore_pixel_lane(A, rows[t], cols[t], r, nobs);
    out_corr[t] = r;
    if (out_nobs) out_nobs[t] = nobs;
}

int launch_rescore_f64(const CorrArgs<double>& A, const int* rows, const int* cols, long long n_px,
                       double* out_corr, double* out_nobs, hipStream_t stream, const long long* n_ptr, int run)
{
    if (n_px == 0) return 0;
    if (run) {
        hipLaunchKernelGGL(rescore_f64_run_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, stream, A, rows, cols, n_px,
                           out_corr, out_nobs, n_ptr);
        return (int)hipGetLastError();
    }
    const long long blocks = (n_px + 3) / 4;              // 4 waves = 4 pixels per block
    hipLaunchKernelGGL(rescore_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, A, rows, cols, n_px, out_corr,
                       out_nobs, n_ptr);
    return (int)hipGetLastError();
}

int launch_rescore_f64_keys(const CorrArgs<double>& A, const long long* keys, int ns, long long n_px, int* rows, int* cols,
                            double* out_corr, const long long* n_ptr, long long* count_copy, long long* zero_next, hipStream_t stream)
{
    if (n_px <= 0 || !n_ptr || !count_copy) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rescore_f64_keys_kernel, dim3((unsigned)((n_px + 3) / 4)), dim3(256), 0, stream, A, keys, ns, n_px, rows, cols,
                       out_corr, n_ptr, count_copy, zero_next);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// view helpers: per-row entry range of a diagonal band of a (sub-)matrix, and the finished law
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_band_extent_kernel(CsrView M, int lo_diag, int hi_diag,
                                                              long long* __restrict__ begin, long long* __restrict__ end)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M.n_rows) return;
    const long long b = M.indptr[row], e = M.row_end[row];
    // first entry with column >= c_lo, first entry with column > c_hi (columns sorted in a row)
    long long c_lo = (long long)row + lo_diag, c_hi = (long long)row + hi_diag;
    if (c_lo < 0) c_lo = 0;
    if (c_hi > M.n_cols - 1) c_hi = M.n_cols - 1;
    c_lo += M.col0;
    c_hi += M.col0;
    long long lo = b, hi = e;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (M.indices[mid] < c_lo) lo = mid + 1;
        else hi = mid;
    }
    const long long first = lo;
    hi = e;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (M.indices[mid] <= c_hi) lo = mid + 1;
        else hi = mid;
    }
    begin[row] = first;
    end[row] = lo < first ? first : lo;
}

int launch_csr_band_extent(const CsrView& M, int lo_diag, int hi_diag, long long* begin, long long* end,
                           hipStream_t stream)
{
    if (M.n_rows == 0) return 0;
    hipLaunchKernelGGL(csr_band_extent_kernel, dim3((M.n_rows + 255) / 256), dim3(256), 0, stream, M, lo_diag, hi_diag,
                       begin, end);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void law_finish_kernel(const double* __restrict__ sum, const long long* __restrict__ cnt,
                                                         int n, double* __restrict__ law)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < n) law[d] = cnt[d] > 0 ? sum[d] / (double)cnt[d] : 0.0;
}

// float64 rows -> float32 rows (the matrix-core kernel stages float32 pixels by LDS-DMA; the float32
// arithmetic class rounds every pixel to float32 first in any case): 12 B per pixel, HBM-bound
__global__ __launch_bounds__(256) void narrow_rows_kernel(const double* __restrict__ src, long long ld_src,
                                                          float* __restrict__ dst, long long ld_dst, int rows, int cols)
{
    const long long total = (long long)rows * cols;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
        dst[(long long)r * ld_dst + c] = (float)src[(long long)r * ld_src + c];
    }
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void peak_rows_kernel(const T* __restrict__ src, long long ld, int rows, int cols,
                                                        unsigned* __restrict__ out)
{
    const long long n = (long long)rows * cols;
    unsigned best = 0u;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const long long r = t / cols, c = t - r * cols;
        best = max(best, __float_as_uint(fabsf((float)src[r * ld + c])));      // (float64 beyond FLT_MAX -> inf: out of range too)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}
}  // namespace

int launch_peak_rows(const void* src, int src_is_f64, long long ld_src, int rows, int cols, int n_cu, unsigned* d_peak_bits,
                     hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return 0;
    const long long want = ((long long)rows * cols + 255) / 256;
    const int blocks = (int)std::min<long long>(want, (long long)n_cu * 8);
    if (src_is_f64)
        hipLaunchKernelGGL(peak_rows_kernel<double>, dim3(blocks), dim3(256), 0, stream, (const double*)src, ld_src, rows, cols, d_peak_bits);
    else
        hipLaunchKernelGGL(peak_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)src, ld_src, rows, cols, d_peak_bits);
    return (int)hipGetLastError();
}

int launch_narrow_rows(const double* src, long long ld_src, float* dst, long long ld_dst, int rows, int cols, int n_cu,
                       hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return 0;
    const long long want = ((long long)rows * cols + 255) / 256;
    const int blocks = (int)std::min<long long>(want, (long long)n_cu * 16);
    hipLaunchKernelGGL(narrow_rows_kernel, dim3(blocks), dim3(256), 0, stream, src, ld_src, dst, ld_dst, rows, cols);
    return (int)hipGetLastError();
}

int launch_law_finish(const double* sum, const long long* cnt, int n, double* law, hipStream_t stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(law_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, cnt, n, law);
    return (int)hipGetLastError();
}

}  // namespace cs
