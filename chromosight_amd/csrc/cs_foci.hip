// cs_foci.hip -- the callers of the correlation on the device: focus picking and the window
// statistics of the validation step, so that `detect` returns a few records per sub-matrix instead
// of a coefficient map.
//
//   reference                                              here
//   detection.py:387-456  pick_foci                        run_detect_foci (host orchestration below)
//   detection.py:459-554  label_foci (4-way adjacency)     link_kernel: union-find over the sorted
//                                                          candidate list (right neighbour = next
//                                                          entry, lower neighbour = binary search)
//   detection.py:557-592  filter_foci                      focus_stats_kernel + flag_roots_kernel
//   detection.py:438-453  per-focus arg-max (Python loop)  focus_stats / focus_argbest kernels
//   detection.py:18-155   validate_patterns (Python loop)  window_stats_kernel, one wave per pattern
//
// Candidate lists are small next to the map (10^3..10^6 pixels), so these kernels are latency
// bound; what matters is that nothing per-candidate runs on the host and that the stages need no
// host round trip between them (counts are read from device memory by the next kernel).
#include <hipcub/hipcub.hpp>

#include <climits>
#include <cstdlib>

#include "cs_device.h"
#include "cs_launch_aux.h"

namespace cs {

#include "cs_foci_kernels.h"     // the kernels (anonymous namespace): this translation unit only

// ================================================================================================
// host side
// ================================================================================================
int launch_decode_keys(const long long* keys, long long n, int ns, int* rows, int* cols, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(decode_keys_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, stream, keys, n, ns, rows, cols);
    return (int)hipGetLastError();
}

size_t foci_scratch_bytes(long long n_cand)
{
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)n_cand);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n_cand);
    const size_t tmp = std::max(sort_tmp, scan_tmp);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n = (size_t)std::max<long long>(n_cand, 1);
    // keys, keys sorted, keys kept (8 B) + vals, vals kept, best_val (8 B) + rows, cols, flag, pos, parent,
    // size, best_idx, out rows / cols / size (4 B) + scores, nobs of the foci (8 B) + records + counters
    return al(tmp) + 3 * al(8 * n) + 3 * al(8 * n) + 10 * al(4 * n) + 2 * al(8 * n) + al(sizeof(FocusRec) * n) + 1024;
}

namespace {
struct Bump {
    char* p;
    template <typename T>
    T* take(size_t n)
    {
        T* out = reinterpret_cast<T*>(p);
        p += (n * sizeof(T) + 255) & ~(size_t)255;
        return out;
    }
};
}  // namespace

// Candidate pixels (rows, cols; n_cand of them, any order) -> foci records.  `scratch` holds
// foci_scratch_bytes(n_cand) bytes.  Everything is enqueued on `stream`; the host reads *h_n_foci
// (pinned) after the stream has drained.  Returns a hipError_t as int.
// radix passes that a row-major key (row * ns + col) needs: the keys of a 16 000 x 16 000 block have 28
// significant bits, and every 8 bits less is one pass (three launches) less
static inline int key_bits(long long ms, long long ns)
{
    unsigned long long top = (unsigned long long)(ms > 0 ? ms : 1) * (unsigned long long)(ns > 0 ? ns : 1);
    int bits = 1;
    while (bits < 63 && (1ull << bits) <= top) ++bits;
    return bits;
}

int enqueue_foci(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n_cand, double pearson,
                 int min_size, int diag_only, int inter, void* scratch, FocusRec** d_rec_out, double* d_windows,
                 long long win_cap, long long* d_n_foci, hipStream_t stream, int presorted, FocusRec* rec_target,
                 long long rec_cap, long long* n_out)
{
    const int ns = A64.ns;
    Bump b{(char*)scratch};
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)n_cand);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n_cand);
    size_t tmp_bytes = std::max(sort_tmp, scan_tmp);
    void* tmp = b.take<char>(tmp_bytes);
    const size_t n = (size_t)n_cand;
    long long* keys = b.take<long long>(n);
    long long* keys_s = b.take<long long>(n);
    long long* keys_k = b.take<long long>(n);
    double* vals = b.take<double>(n);
    double* vals_k = b.take<double>(n);
    unsigned long long* best_val = b.take<unsigned long long>(n);
    int* rows = b.take<int>(n);
    int* cols = b.take<int>(n);
    int* flag = b.take<int>(n);
    int* pos = b.take<int>(n);
    int* parent = b.take<int>(n);
    int* size = b.take<int>(n);
    int* best_idx = b.take<int>(n);
    int* f_rows = b.take<int>(n);
    int* f_cols = b.take<int>(n);
    int* f_size = b.take<int>(n);
    double* f_score = b.take<double>(n);
    double* f_nobs = b.take<double>(n);
    FocusRec* rec = b.take<FocusRec>(n);
    int* n_kept = b.take<int>(64);
    if (rec_target) rec = rec_target;          // caller's (page-locked host) records, rec_cap of them
    else rec_cap = (long long)n;
    *d_rec_out = rec;
    const unsigned g = blocks_for(n_cand);

    // row-major order of the candidates (a list that already has it only needs its keys)
    hipError_t e = hipSuccess;
    const int* s_rows = rows;
    const int* s_cols = cols;
    if (presorted) {
        hipLaunchKernelGGL(make_keys_kernel, dim3(g), dim3(kThreads), 0, stream, d_rows, d_cols, n_cand, ns, keys_s);
        s_rows = d_rows;
        s_cols = d_cols;
    } else {
        hipLaunchKernelGGL(make_keys_kernel, dim3(g), dim3(kThreads), 0, stream, d_rows, d_cols, n_cand, ns, keys);
        e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, keys_s, (int)n_cand, 0, key_bits(A64.ms, ns), stream);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(decode_keys_kernel, dim3(g), dim3(kThreads), 0, stream, keys_s, n_cand, ns, rows, cols);
    }
    // exact coefficients, exact threshold
    int rc = launch_rescore_f64(A64, s_rows, s_cols, n_cand, vals, nullptr, stream, nullptr,
                                presorted && A64.km * A64.kn <= 17 * 17 && !getenv("CHROMOSIGHT_HIP_NO_RUN_RESCORE"));
    if (rc) return rc;
    if (n_cand <= kSmallMax) {
        hipLaunchKernelGGL(foci_small_kernel, dim3(1), dim3(kSmallThreads), 0, stream, keys_s, vals, n_cand, pearson, ns, min_size,
                           diag_only, flag, pos, keys_k, vals_k, parent, size, best_val, best_idx, f_rows, f_cols, f_size, n_kept,
                           d_n_foci);
    } else {
    hipLaunchKernelGGL(flag_keep_kernel, dim3(g), dim3(kThreads), 0, stream, vals, n_cand, pearson, flag);
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, pos, (int)n_cand, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(scatter_keep_kernel, dim3(g), dim3(kThreads), 0, stream, keys_s, vals, flag, pos, n_cand, keys_k,
                       vals_k, n_kept);
    // 4-connected foci of the kept pixels
    hipLaunchKernelGGL(init_focus_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, parent, size, best_val, best_idx, n_cand);
    hipLaunchKernelGGL(link_kernel, dim3(g), dim3(kThreads), 0, stream, keys_k, n_kept, ns, parent);
    hipLaunchKernelGGL(flatten_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, parent);
    hipLaunchKernelGGL(focus_stats_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, parent, vals_k, size, best_val);
    hipLaunchKernelGGL(focus_argbest_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, parent, vals_k, best_val, best_idx);
    hipLaunchKernelGGL(flag_roots_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, parent, size, min_size, flag, n_cand);
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, pos, (int)n_cand, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(emit_foci_kernel, dim3(g), dim3(kThreads), 0, stream, n_kept, flag, pos, best_idx, size, keys_k, ns,
                       diag_only, f_rows, f_cols, f_size, n_cand, d_n_foci);
    }
    // score / n_obs at the final coordinates (they move for 1-D patterns), window statistics.
    // The number of foci stays on the device: launched for the worst case (n_cand / min_size foci),
    // surplus waves exit on the device-side count.
    const long long max_foci = std::max<long long>(1, n_cand / std::max(min_size, 1));
    rc = launch_rescore_f64(A64, f_rows, f_cols, max_foci, f_score, f_nobs, stream, d_n_foci);
    if (rc) return rc;
    hipLaunchKernelGGL(window_stats_kernel, dim3((unsigned)((max_foci + 3) / 4)), dim3(kThreads), 0, stream, A64, inter,
                       f_rows, f_cols, f_size, f_score, f_nobs, d_n_foci, 0ll, rec, d_windows, win_cap, rec_cap, n_out);
    return (int)hipGetLastError();
}

// ---- one sub-matrix split over several GPUs: candidates per row window, labelled
//      together.  Part A = candidates of a row window that pass the exact threshold, sorted; part B =
//      foci of a (merged) candidate list with known float64 values.
namespace {
__global__ void store_int_kernel(int* p, int v) { *p = v; }
}  // namespace

size_t keep_scratch_bytes(long long n_cand)
{
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)n_cand);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n_cand);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n = (size_t)std::max<long long>(n_cand, 1);
    return al(std::max(sort_tmp, scan_tmp)) + 3 * al(8 * n) + 2 * al(8 * n) + 4 * al(4 * n) + 1024;
}

// candidates (any order) -> sorted keys / float64 values of those >= pearson; *n_kept on the device
int enqueue_keep(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n_cand, double pearson,
                 void* scratch, int** rows_out, int** cols_out, double** vals_out, int** n_kept_out, hipStream_t stream)
{
    const int ns = A64.ns;
    Bump b{(char*)scratch};
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)n_cand);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n_cand);
    size_t tmp_bytes = std::max(sort_tmp, scan_tmp);
    void* tmp = b.take<char>(tmp_bytes);
    const size_t n = (size_t)n_cand;
    long long* keys = b.take<long long>(n);
    long long* keys_s = b.take<long long>(n);
    long long* keys_k = b.take<long long>(n);
    double* vals = b.take<double>(n);
    double* vals_k = b.take<double>(n);
    int* rows = b.take<int>(n);
    int* cols = b.take<int>(n);
    int* flag = b.take<int>(n);
    int* pos = b.take<int>(n);
    int* n_kept = b.take<int>(64);
    *rows_out = rows;
    *cols_out = cols;
    *vals_out = vals_k;
    *n_kept_out = n_kept;
    const unsigned g = blocks_for(n_cand);
    hipLaunchKernelGGL(store_int_kernel, dim3(1), dim3(1), 0, stream, n_kept, 0);
    hipLaunchKernelGGL(make_keys_kernel, dim3(g), dim3(kThreads), 0, stream, d_rows, d_cols, n_cand, ns, keys);
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, keys_s, (int)n_cand, 0, key_bits(A64.ms, ns), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(decode_keys_kernel, dim3(g), dim3(kThreads), 0, stream, keys_s, n_cand, ns, rows, cols);
    int rc = launch_rescore_f64(A64, rows, cols, n_cand, vals, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(flag_keep_kernel, dim3(g), dim3(kThreads), 0, stream, vals, n_cand, pearson, flag);
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, pos, (int)n_cand, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(scatter_keep_kernel, dim3(g), dim3(kThreads), 0, stream, keys_s, vals, flag, pos, n_cand, keys_k,
                       vals_k, n_kept);
    // coordinates of the kept pixels (the entries beyond *n_kept are scratch)
    hipLaunchKernelGGL(decode_keys_kernel, dim3(g), dim3(kThreads), 0, stream, keys_k, n_cand, ns, rows, cols);
    return (int)hipGetLastError();
}

size_t label_scratch_bytes(long long n)
{
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr,
                                             (const double*)nullptr, (double*)nullptr, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t m = (size_t)std::max<long long>(n, 1);
    return al(std::max(sort_tmp, scan_tmp)) + 2 * al(8 * m) + 3 * al(8 * m) + 10 * al(4 * m) + 1024;
}

// candidates with their float64 values (device arrays, any order) -> foci in the reference's order:
// (row, col) of the maximum of every focus of >= min_size pixels and its size; *d_n_foci on the device
int enqueue_label(const int* d_rows, const int* d_cols, const double* d_vals, long long n, int ns, int min_size,
                  int diag_only, void* scratch, int** f_rows_out, int** f_cols_out, int** f_size_out,
                  long long* d_n_foci, hipStream_t stream)
{
    Bump b{(char*)scratch};
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr,
                                             (const double*)nullptr, (double*)nullptr, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n);
    size_t tmp_bytes = std::max(sort_tmp, scan_tmp);
    void* tmp = b.take<char>(tmp_bytes);
    const size_t m = (size_t)n;
    long long* keys = b.take<long long>(m);
    long long* keys_s = b.take<long long>(m);
    double* vals_s = b.take<double>(m);
    unsigned long long* best_val = b.take<unsigned long long>(m);
    double* spare = b.take<double>(m);
    (void)spare;
    int* flag = b.take<int>(m);
    int* pos = b.take<int>(m);
    int* parent = b.take<int>(m);
    int* size = b.take<int>(m);
    int* best_idx = b.take<int>(m);
    int* f_rows = b.take<int>(m);
    int* f_cols = b.take<int>(m);
    int* f_size = b.take<int>(m);
    int* n_dev = b.take<int>(64);
    *f_rows_out = f_rows;
    *f_cols_out = f_cols;
    *f_size_out = f_size;
    const unsigned g = blocks_for(n);
    hipLaunchKernelGGL(store_int_kernel, dim3(1), dim3(1), 0, stream, n_dev, (int)n);
    hipLaunchKernelGGL(make_keys_kernel, dim3(g), dim3(kThreads), 0, stream, d_rows, d_cols, n, ns, keys);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_s, d_vals, vals_s, (int)n, 0, 64, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(init_focus_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, parent, size, best_val, best_idx, n);
    hipLaunchKernelGGL(link_kernel, dim3(g), dim3(kThreads), 0, stream, keys_s, n_dev, ns, parent);
    hipLaunchKernelGGL(flatten_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, parent);
    hipLaunchKernelGGL(focus_stats_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, parent, vals_s, size, best_val);
    hipLaunchKernelGGL(focus_argbest_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, parent, vals_s, best_val, best_idx);
    hipLaunchKernelGGL(flag_roots_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, parent, size, min_size, flag, n);
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, pos, (int)n, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(emit_foci_kernel, dim3(g), dim3(kThreads), 0, stream, n_dev, flag, pos, best_idx, size, keys_s, ns,
                       diag_only, f_rows, f_cols, f_size, n, d_n_foci);
    return (int)hipGetLastError();
}

// ---- 1-D patterns: every pixel of a band of at most 4 diagonals is a candidate ---------------------
namespace {
// row-major enumeration of the band lo .. lo + w - 1 (lo >= 0) over the rows rb .. re - 1: rows whose w
// pixels all exist come first (w per row), the last rows lose one pixel each
__global__ __launch_bounds__(kThreads) void enumerate_band_rowmajor_kernel(int rb, int re, int ns, int lo, int w, long long n,
                                                                           int* __restrict__ rows, int* __restrict__ cols)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    const int last_full = min(re, ns - lo - w + 1);              // rows < last_full have all w pixels
    const long long full = (long long)max(last_full - rb, 0) * w;
    int row, x;
    if (t < full) {
        row = rb + (int)(t / w);
        x = (int)(t - (long long)(row - rb) * w);
    } else {
        long long rest = t - full;
        row = max(last_full, rb);
        for (;;) {                                               // at most w - 1 short rows
            const int cnt = max(0, min(w, ns - row - lo));
            if (rest < cnt) break;
            rest -= cnt;
            ++row;
        }
        x = (int)rest;
    }
    rows[t] = row;
    cols[t] = row + lo + x;
}

__global__ __launch_bounds__(kThreads) void enumerate_band_kernel(int rb, int re, int ns, int lo, int w, long long n,
                                                                  int* __restrict__ rows, int* __restrict__ cols)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    // diagonal-major enumeration: diagonal d = lo + x holds the rows max(rb, -d) .. min(re, ns - d) - 1
    long long rest = t;
    for (int x = 0; x < w; ++x) {
        const int d = lo + x;
        const int r0 = max(rb, -d);
        const long long cnt = (long long)min(re, ns - d) - r0;
        if (cnt <= 0) continue;
        if (rest < cnt) {
            rows[t] = r0 + (int)rest;
            cols[t] = r0 + (int)rest + d;
            return;
        }
        rest -= cnt;
    }
}
}  // namespace

long long narrow_band_pixels(int rb, int re, int ns, int lo, int w)
{
    long long n = 0;
    for (int x = 0; x < w; ++x) {
        const int d = lo + x;
        const long long cnt = (long long)std::min(re, ns - d) - std::max(rb, -d);
        if (cnt > 0) n += cnt;
    }
    return n;
}

int enqueue_enumerate_band(int rb, int re, int ns, int lo, int w, long long n, int* d_rows, int* d_cols, hipStream_t stream,
                           int* row_major)
{
    *row_major = 0;
    if (n <= 0) return 0;
    if (lo >= 0) {
        // the list comes out in the row-major order the foci stages need: no sort
        *row_major = 1;
        hipLaunchKernelGGL(enumerate_band_rowmajor_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, stream, rb, re, ns, lo, w, n,
                           d_rows, d_cols);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(enumerate_band_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, stream, rb, re, ns, lo, w, n, d_rows, d_cols);
    return (int)hipGetLastError();
}

// ---- host side of the batched 1-D pattern chain ----------------------------------------------------
size_t narrow_batch_scratch_bytes(int n_blocks, long long n_total)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n = (size_t)std::max<long long>(n_total, 1), nb = (size_t)n_blocks + 1;
    return al(sizeof(CorrArgs<double>) * nb) + al(8 * nb) + al(8 * nb) + 4 * al(8 * n) + 5 * al(8 * n) + 16 * al(4 * n) +
           4 * al(8 * nb) + 4096;
}

// h_tab / h_seg / h_lo_w: host arrays of n_blocks (+1 for seg); results: records (and windows) block after
// block into rec / windows (device-visible, capacities rec_cap / win_cap), h_counts[0] = total,
// h_counts[1 + b] = foci of block b (device-visible page-locked memory)
int enqueue_foci_narrow_batch(const CorrArgs<double>* h_tab, const long long* h_seg, const int* h_lo_w, int n_blocks, double pearson,
                              int min_size, int diag_only, int inter, void* scratch, FocusRec* rec, long long rec_cap,
                              double* windows, long long win_cap, long long* h_counts, hipStream_t stream)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const long long n_total = h_seg[n_blocks];
    const size_t n = (size_t)std::max<long long>(n_total, 1), nb = (size_t)n_blocks + 1;
    Bump b{(char*)scratch};
    (void)al;
    CorrArgs<double>* tab = b.take<CorrArgs<double>>(nb);
    long long* seg = b.take<long long>(nb);
    int2* lo_w = reinterpret_cast<int2*>(b.take<long long>(nb));
    long long* keys = b.take<long long>(n);
    long long* keys_k = b.take<long long>(n);
    double* vals = b.take<double>(n);
    double* vals_k = b.take<double>(n);
    unsigned long long* best_val = b.take<unsigned long long>(n);
    double* f_score = b.take<double>(n);
    double* f_nobs = b.take<double>(n);
    long long* spare64 = b.take<long long>(n);
    (void)spare64;
    int* rows = b.take<int>(n);
    int* cols = b.take<int>(n);
    int* blk = b.take<int>(n);
    int* flag = b.take<int>(n);
    int* pos = b.take<int>(n);
    int* parent = b.take<int>(n);
    int* size = b.take<int>(n);
    int* best_idx = b.take<int>(n);
    int* s_rows = b.take<int>(n);
    int* s_cols = b.take<int>(n);
    int* s_size = b.take<int>(n);
    int* f_rows = b.take<int>(n);
    int* f_cols = b.take<int>(n);
    int* f_size = b.take<int>(n);
    int* f_blk = b.take<int>(n);
    int* n_kept = b.take<int>(nb);
    long long* n_foci_blk = b.take<long long>(nb);
    long long* f_off = b.take<long long>(nb);
    long long* d_total = b.take<long long>(8);
    hipError_t e = hipMemcpyAsync(tab, h_tab, sizeof(CorrArgs<double>) * (size_t)n_blocks, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(seg, h_seg, 8 * nb, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(lo_w, h_lo_w, 8 * (size_t)n_blocks, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    if (n_total > 0) {
        const unsigned g = blocks_for(n_total);
        hipLaunchKernelGGL(narrow_enumerate_batch_kernel, dim3(g), dim3(kThreads), 0, stream, tab, seg, lo_w, n_blocks, rows, cols, keys,
                           blk);
        // templates beyond 17 x 17 (81 x 81 centromeres) do not fit the run kernel's LDS tile: a lane would walk
        // thousands of window pixels straight from memory, so they keep one wave per pixel
        const bool big_template = 3LL * h_tab[0].km * h_tab[0].kn > kRunWeights;
        if (getenv("CHROMOSIGHT_HIP_NO_RUN_RESCORE") || big_template)
            launch_rescore_batch(h_tab, n_blocks, n_total, stream, tab, blk, rows, cols, n_total, (const long long*)nullptr, vals,
                                 (double*)nullptr);
        else {
            // tile of a workgroup's 256 entries: 256 / w rows (+ halo) x (w + 32) diagonals, w = narrowest scanned band
            int w_min = 1 << 30, w_max = 1;
            for (int k = 0; k < n_blocks; ++k) {
                w_min = std::min(w_min, std::max(h_lo_w[2 * k + 1], 1));
                w_max = std::max(w_max, h_lo_w[2 * k + 1]);
            }
            int tile_cap = getenv("CHROMOSIGHT_HIP_RUN_NO_LDS") ? 0 : (256 / w_min + 1 + 17) * (w_max + 33);
            if ((size_t)tile_cap * 8 > 48 * 1024) tile_cap = 48 * 1024 / 8;        // wider scans: the direct route where needed
            const size_t smem = (size_t)tile_cap * 8 + kRunWeights * 8 + 512 + 1024;
            const bool no_run17 = getenv("CHROMOSIGHT_HIP_NO_RUN17") != nullptr;
            bool only17 = fast_windows_on() && !no_run17 && w_max <= 2 && tile_cap > 0;
            for (int k = 0; only17 && k < n_blocks; ++k) {
                const CorrArgs<double>& A = h_tab[k];
                only17 = A.km == 17 && A.kn == 17 && A.mask_mode == 1 && A.sym_upper && A.full && A.max_dist >= 0;
            }
            // Several templates on the same sub-matrices (borders: three; the virtual blocks k nb .. k nb + nb - 1 are template k's):
            // the list holds every pixel once per template, in the same order -- ONE pass over the first template's entries
            // evaluates all of them (rescore_run17_multi).  CHROMOSIGHT_HIP_TEMPLATE_FUSION=0 | 1: never | always.
            // Only where the separate passes would not all be resident at once (three workgroups of this kernel per CU): a short
            // list -- a rank's eighth of a genome: 600 workgroups -- is bound by ONE wave's latency, which grows with the templates
            // a lane evaluates (measured: 80 us separate, 100 us fused; the whole genome: 515 against 286 us).
            static int n_cu_cached = 0;
            if (!n_cu_cached) {
                int dev_id = 0, n_cu_dev = 0;
                if (hipGetDevice(&dev_id) == hipSuccess && hipDeviceGetAttribute(&n_cu_dev, hipDeviceAttributeMultiprocessorCount, dev_id) == hipSuccess && n_cu_dev > 0)
                    n_cu_cached = n_cu_dev;
                else
                    n_cu_cached = 256;
            }
            const char* fusion = getenv("CHROMOSIGHT_HIP_TEMPLATE_FUSION");       // 1: always, 0: never
            const bool worth_it = fusion ? atoi(fusion) != 0 : (n_total + 255) / 256 > 3LL * n_cu_cached;
            int n_fuse = 1;
            if (only17 && worth_it) {
                for (int T = 3; T >= 2 && n_fuse == 1; --T) {       // (instances for two and three templates)
                    if (n_blocks % T) continue;
                    const int nbk = n_blocks / T;
                    bool same = true;
                    for (int k = 1; k < T && same; ++k)
                        for (int q = 0; q < nbk && same; ++q) {
                            const int o = k * nbk + q;
                            CorrArgs<double> other = h_tab[o];
                            other.w = h_tab[q].w;                // what a template brings: weights, their statistics and shape flags
                            other.ks = h_tab[q].ks;
                            other.w_sym = h_tab[q].w_sym;
                            other.w_rank1 = h_tab[q].w_rank1;
                            same = std::memcmp(&other, &h_tab[q], sizeof(other)) == 0 && h_lo_w[2 * o] == h_lo_w[2 * q] &&
                                   h_lo_w[2 * o + 1] == h_lo_w[2 * q + 1] && h_seg[o + 1] - h_seg[o] == h_seg[q + 1] - h_seg[q];
                        }
                    if (same) n_fuse = T;
                }
                if (getenv("CHROMOSIGHT_HIP_DEBUG")) fprintf(stderr, "[chromosight_hip] run re-scoring: %d sub-matrices, %d templates fused\n", n_blocks, n_fuse);
            }
            const long long n_launch = n_total / n_fuse;
            const dim3 run_grid((unsigned)((n_launch + 255) / 256));
            if (only17)
                hipLaunchKernelGGL(rescore_run_batch_kernel<true>, run_grid, dim3(256), smem, stream, tab, blk, rows, cols, n_launch, vals,
                                   tile_cap, 0, true, n_fuse, n_blocks / n_fuse, n_launch);
            else
                hipLaunchKernelGGL(rescore_run_batch_kernel<false>, run_grid, dim3(256), smem, stream, tab, blk, rows, cols, n_total, vals,
                                   tile_cap, no_run17 ? 1 : 0, fast_windows_on(), 1, 0, 0LL);
        }
    }
    // every sub-matrix scans the diagonals 0 and 1 (borders, hairpins: max_dist = 0 in the config): the foci are runs
    bool path = n_total > 0 && min_size >= 1 && !getenv("CHROMOSIGHT_HIP_NO_PATH_FOCI");
    for (int k = 0; k < n_blocks && path; ++k) path = h_lo_w[2 * k] == 0 && h_lo_w[2 * k + 1] == 2;
    size_t scan_tmp = 0;
    if (path) {
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, (int)n_total);
        path = scan_tmp <= 8 * n;                              // (its scratch: the list of kept keys, unused on this route)
    }
    if (path) {
        const unsigned g = blocks_for(std::max<long long>(n_total, n_blocks + 1));
        hipLaunchKernelGGL(path_runs_kernel, dim3(blocks_for(n_total)), dim3(kThreads), 0, stream, seg, blk, vals, n_total, pearson, min_size,
                           flag, size, best_idx);
        hipError_t es = hipcub::DeviceScan::ExclusiveSum(keys_k, scan_tmp, flag, pos, (int)n_total, stream);
        if (es != hipSuccess) return (int)es;
        hipLaunchKernelGGL(path_emit_kernel, dim3(g), dim3(kThreads), 0, stream, tab, seg, n_blocks, n_total, blk, keys, flag, pos, size,
                           best_idx, diag_only, f_rows, f_cols, f_size, f_blk, f_off, d_total, h_counts);
    } else {
        const size_t lds_small = foci_small_lds_bytes();
        hipLaunchKernelGGL(foci_small_batch_kernel, dim3(n_blocks), dim3(kSmallThreads), lds_small, stream, tab, seg, keys, vals, pearson, min_size,
                           diag_only, flag, pos, keys_k, vals_k, parent, size, best_val, best_idx, s_rows, s_cols, s_size, n_kept, n_foci_blk,
                           lds_small ? 1 : 0);
        hipLaunchKernelGGL(gather_foci_batch_kernel, dim3(1), dim3(kSmallThreads), 0, stream, seg, n_foci_blk, n_blocks, s_rows, s_cols, s_size,
                           f_rows, f_cols, f_size, f_blk, f_off, d_total, h_counts);
    }
    if (n_total > 0) {
        const long long max_foci = std::max<long long>(1, n_total / std::max(min_size, 1));
        launch_focus_records(h_tab, n_blocks, max_foci, stream, tab, inter, f_blk, f_rows, f_cols, f_size, f_score, f_nobs, d_total, rec,
                             windows, win_cap, rec_cap);
    }
    return (int)hipGetLastError();
}

// ---- 2-D patterns of MANY sub-matrices: the candidates arrive as one unsorted list of composite keys
//      (block << kKeyShift) + row * ns + col, appended by the tile kernels of all blocks ------------------------------
namespace {
// One sorted list of composite keys: threads [0, n) split their key into (block, row, column, the block's own row-major key),
// threads [n, n + n_blocks] find the first key of block t - n (the segments)
__global__ __launch_bounds__(kThreads) void keyed_split_segments_kernel(const CorrArgs<double>* __restrict__ tab, const long long* __restrict__ keys,
                                                                        long long n, int n_blocks, int shift, int* __restrict__ rows,
                                                                        int* __restrict__ cols, int* __restrict__ blk,
                                                                        long long* __restrict__ local, long long* __restrict__ seg)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < n) {
        const long long key = keys[t];
        const int b = (int)(key >> shift);
        const long long rem = key - ((long long)b << shift);
        const int ns = tab[b].ns;
        rows[t] = (int)(rem / ns);
        cols[t] = (int)(rem - (rem / ns) * ns);
        blk[t] = b;
        local[t] = rem;
    } else if (t - n <= n_blocks) {
        const long long want = (t - n) << shift;
        long long lo = 0, hi = n;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        seg[t - n] = lo;
    }
}
}  // namespace

namespace {
// segmented candidate lists (block b's keys at regions[base[b] ..), seg = prefix of their lengths): entry t of the compact
// numbering -> its key, decoded like keyed_split_kernel; one wave-uniform search over the few blocks per thread
__global__ __launch_bounds__(kThreads) void segmented_unpack_kernel(const CorrArgs<double>* __restrict__ tab, const long long* __restrict__ regions,
                                                                    const long long* __restrict__ base, const long long* __restrict__ seg,
                                                                    int n_blocks, long long n, int shift, int* __restrict__ rows,
                                                                    int* __restrict__ cols, int* __restrict__ blk,
                                                                    long long* __restrict__ local, long long* __restrict__ keys_c,
                                                                    const long long* __restrict__ n_ptr)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n || (n_ptr && t >= *n_ptr)) return;            // (n_ptr: the length is known on the device only; n bounds the launch)
    int lo = 0, hi = n_blocks - 1;                            // the block with seg[b] <= t < seg[b + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    const long long key = regions[base[lo] + (t - seg[lo])];
    const int b = (int)(key >> shift);
    const long long rem = key - ((long long)b << shift);
    const int ns = tab[b].ns;
    rows[t] = (int)(rem / ns);
    cols[t] = (int)(rem - (rem / ns) * ns);
    blk[t] = b;
    local[t] = rem;
    keys_c[t] = key;
}

// The segments of the compact numbering from the blocks' own candidate counters, on the device (the chain behind the tile kernels
// is enqueued before the host knows them): seg[b + 1] - seg[b] = min(count[b], cap[b]); status[0] = seg[n_blocks],
// status[1] = flags -- 1: a list outgrew its room (the host repeats the call with more), 2: a list is too long for the labelling
// workgroup's LDS arrays or its sub-matrix too large for 32-bit keys, 4: more candidates than the launches were sized for
// (2, 4: the host runs the chain again, paced by the counts); the regions' starts go from the host's table to the device's.
// One wave.
__global__ __launch_bounds__(64) void segments_from_counts_kernel(const CorrArgs<double>* __restrict__ tab, const long long* __restrict__ counts,
                                                                  const long long* __restrict__ h_base, const long long* __restrict__ cap,
                                                                  int n_blocks, long long bound, long long* __restrict__ seg,
                                                                  long long* __restrict__ base_d, long long* __restrict__ h_out)
{
    // h_base / cap / h_out: page-locked host memory the device reads and writes directly (a few words: no copy engine, no
    // launch gap in front of or behind the chain)
    const int lane = threadIdx.x;
    long long run = 0;
    int flags = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + lane;
        long long n = 0;
        if (b < n_blocks) {
            const long long c = counts[b];
            n = min(c, cap[b]);
            base_d[b] = h_base[b];
            h_out[b] = c;
            if (c > cap[b]) flags |= 1;
            if (n > kSmallLds || (unsigned long long)tab[b].ms * (unsigned long long)tab[b].ns > 0xffffffffull) flags |= 2;
        }
        long long incl = n;                                   // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (b < n_blocks) seg[b] = run + incl - n;
        run += __shfl(incl, 63, 64);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) flags |= __shfl_xor(flags, off, 64);
    if (run > bound) flags |= 4;                              // (run and flags are wave-uniform here)
    if (lane == 0) {
        seg[n_blocks] = run;
        h_out[60] = run;
        h_out[61] = flags;
    }
    // Any flag sends the caller to the host-paced chain (cs_api_foci.cpp cs_detect_foci_blocks), whose results replace whatever this
    // chain writes: with empty segments the kernels behind this one do nothing -- they used to label lists of which only the first
    // `bound` entries were filled and to write records built from that into the caller's page-locked buffers first
    if (flags != 0)
        for (int b = lane; b <= n_blocks; b += 64) seg[b] = 0;
}
}  // namespace

// the chain with device-side counts needs the labelling workgroups' LDS route (unsorted lists)
bool keyed_batch_deferred_available() { return foci_small_lds_bytes() > 0; }

// The argument table of enqueue_foci_keyed_batch(..., n_total, scratch, ...) uploaded ahead of it on another stream (ordered
// before the chain by the caller): where the chain's first allocation puts it.
int upload_keyed_batch_table(const CorrArgs<double>* h_tab, int n_blocks, long long n_total, void* scratch, hipStream_t stream)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)std::max<long long>(n_total, 1));
    CorrArgs<double>* tab = reinterpret_cast<CorrArgs<double>*>((char*)scratch + al(al(sort_tmp)));
    return (int)hipMemcpyAsync(tab, h_tab, sizeof(CorrArgs<double>) * (size_t)n_blocks, hipMemcpyHostToDevice, stream);
}

size_t keyed_batch_scratch_bytes(int n_blocks, long long n_total)
{
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)std::max<long long>(n_total, 1));
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n = (size_t)std::max<long long>(n_total, 1), nb = (size_t)n_blocks + 1;
    return al(sort_tmp) + al(sizeof(CorrArgs<double>) * nb) + 9 * al(8 * n) + 15 * al(4 * n) + 7 * al(8 * nb) + 4096;
}

// d_keys: n_total composite keys in any order (device).  h_tab: the float64 argument blocks of the sub-matrices.
// Records (and windows) block after block into rec / windows (device-visible page-locked memory), h_counts[0] = total,
// h_counts[1 + b] = foci of block b.  Same chain as enqueue_foci_narrow_batch after its enumeration.
int enqueue_foci_keyed_batch(const CorrArgs<double>* h_tab, int n_blocks, const long long* d_keys, long long n_total, int shift,
                             double pearson, int min_size, int diag_only, int inter, void* scratch, FocusRec* rec, long long rec_cap,
                             double* windows, long long win_cap, long long* h_counts, hipStream_t stream,
                             const long long* h_base, const long long* h_seg, const DeferredSegments* deferred)
{
    // h_base / h_seg (or null): SEGMENTED lists -- block b's keys, unsorted, at d_keys[h_base[b] ..), h_seg[b + 1] - h_seg[b] of
    // them (n_total = h_seg[n_blocks]); both arrays must outlive the call's copies (the caller synchronises).
    // deferred (or null): segmented lists whose LENGTHS are still being written by the tile kernels in front of this chain on
    // `stream` -- n_total is then the room of d_keys (every array of the chain is laid out for it), the segments are formed on
    // the device from deferred->d_counts, the launches are sized for deferred->bound candidates, and the blocks' counts, the
    // total and the status flags (segments_from_counts_kernel) are written to deferred->h_counts_out by the chain itself.
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n = (size_t)std::max<long long>(n_total, 1), nb = (size_t)n_blocks + 1;
    Bump b{(char*)scratch};
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const long long*)nullptr, (long long*)nullptr, (int)n);
    void* tmp = b.take<char>(al(sort_tmp));
    CorrArgs<double>* tab = b.take<CorrArgs<double>>(nb);
    long long* keys_s = b.take<long long>(n);
    long long* local = b.take<long long>(n);
    long long* keys_k = b.take<long long>(n);
    double* vals = b.take<double>(n);
    double* vals_k = b.take<double>(n);
    unsigned long long* best_val = b.take<unsigned long long>(n);
    double* f_score = b.take<double>(n);
    double* f_nobs = b.take<double>(n);
    long long* spare = b.take<long long>(n);
    (void)spare;
    int* rows = b.take<int>(n);
    int* cols = b.take<int>(n);
    int* blk = b.take<int>(n);
    int* flag = b.take<int>(n);
    int* pos = b.take<int>(n);
    int* parent = b.take<int>(n);
    int* size = b.take<int>(n);
    int* best_idx = b.take<int>(n);
    int* s_rows = b.take<int>(n);
    int* s_cols = b.take<int>(n);
    int* s_size = b.take<int>(n);
    int* f_rows = b.take<int>(n);
    int* f_cols = b.take<int>(n);
    int* f_size = b.take<int>(n);
    int* f_blk = b.take<int>(n);
    long long* seg = b.take<long long>(2 * nb);            // (segmented lists: the regions' starts right behind the segments, one upload)
    long long* base_d = seg + nb;
    int* n_kept = reinterpret_cast<int*>(b.take<long long>(nb));
    long long* n_foci_blk = b.take<long long>(nb);
    long long* f_off = b.take<long long>(nb);
    long long* d_total = b.take<long long>(8);
    hipError_t e = hipSuccess;
    if (!(deferred && deferred->tab_uploaded)) e = hipMemcpyAsync(tab, h_tab, sizeof(CorrArgs<double>) * (size_t)n_blocks, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    const size_t lds_small = foci_small_lds_bytes();
    if (deferred) {
        if (lds_small == 0) return (int)hipErrorInvalidValue;          // (the unsorted lists need the labelling workgroups' LDS route)
        const long long bound = std::max<long long>(1, std::min<long long>(deferred->bound, n_total));
        hipLaunchKernelGGL(segments_from_counts_kernel, dim3(1), dim3(64), 0, stream, tab, deferred->d_counts, deferred->h_base, deferred->h_cap,
                           n_blocks, bound, seg, base_d, deferred->h_counts_out);
        const long long* n_dev = seg + n_blocks;
        hipLaunchKernelGGL(segmented_unpack_kernel, dim3(blocks_for(bound)), dim3(kThreads), 0, stream, tab, d_keys, base_d, seg, n_blocks, bound,
                           shift, rows, cols, blk, local, keys_k, n_dev);
        launch_rescore_batch(h_tab, n_blocks, bound, stream, tab, blk, rows, cols, bound, n_dev, vals, (double*)nullptr);
        hipLaunchKernelGGL(foci_small_batch_kernel, dim3(n_blocks), dim3(kSmallThreads), lds_small, stream, tab, seg, local, vals, pearson,
                           min_size, diag_only, flag, pos, keys_k, vals_k, parent, size, best_val, best_idx, s_rows, s_cols, s_size, n_kept,
                           n_foci_blk, 1, 1);
        hipLaunchKernelGGL(gather_foci_batch_kernel, dim3(1), dim3(kSmallThreads), 0, stream, seg, n_foci_blk, n_blocks, s_rows, s_cols, s_size,
                           f_rows, f_cols, f_size, f_blk, f_off, d_total, h_counts);
        launch_focus_records(h_tab, n_blocks, std::max<long long>(1, bound / std::max(min_size, 1)), stream, tab, inter, f_blk, f_rows, f_cols,
                             f_size, f_score, f_nobs, d_total, rec, windows, win_cap, rec_cap);
        return (int)hipGetLastError();
    }
    int unsorted = 0;
    if (h_base && h_seg && n_total > 0) {
        // every block's list short enough for the labelling workgroup's LDS arrays (which sort it there)?  then no device-wide
        // sort, no segments, no split: one unpack launch
        bool all_small = lds_small > 0;
        for (int k = 0; k < n_blocks && all_small; ++k)
            all_small = h_seg[k + 1] - h_seg[k] <= kSmallLds &&
                        (unsigned long long)h_tab[k].ms * (unsigned long long)h_tab[k].ns <= 0xffffffffull;
        // (h_seg: n_blocks + 1 segment starts, then -- h_base == h_seg + n_blocks + 1 -- the n_blocks region starts: one copy)
        if (h_base == h_seg + nb) e = hipMemcpyAsync(seg, h_seg, 8 * (nb + (size_t)n_blocks), hipMemcpyHostToDevice, stream);
        else {
            e = hipMemcpyAsync(seg, h_seg, 8 * nb, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(base_d, h_base, 8 * (size_t)n_blocks, hipMemcpyHostToDevice, stream);
        }
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(segmented_unpack_kernel, dim3(blocks_for(n_total)), dim3(kThreads), 0, stream, tab, d_keys, base_d, seg, n_blocks,
                           n_total, shift, rows, cols, blk, local, keys_k, (const long long*)nullptr);
        if (all_small) unsorted = 1;
        else d_keys = keys_k;                       // the compact list through the sorted route below
    }
    int blk_bits = 1;
    while ((1 << blk_bits) < n_blocks) ++blk_bits;
    if (n_total > 0 && !unsorted) {
        e = hipcub::DeviceRadixSort::SortKeys(tmp, sort_tmp, d_keys, keys_s, (int)n_total, 0, shift + blk_bits, stream);
        if (e != hipSuccess) return (int)e;
    }
    if (unsorted) {
        launch_rescore_batch(h_tab, n_blocks, n_total, stream, tab, blk, rows, cols, n_total, (const long long*)nullptr, vals, (double*)nullptr);
    } else {
        hipLaunchKernelGGL(keyed_split_segments_kernel, dim3(blocks_for(n_total + n_blocks + 1)), dim3(kThreads), 0, stream, tab, keys_s,
                           n_total, n_blocks, shift, rows, cols, blk, local, seg);
        if (n_total > 0)
            launch_rescore_batch(h_tab, n_blocks, n_total, stream, tab, blk, rows, cols, n_total, (const long long*)nullptr, vals, (double*)nullptr);
    }
    hipLaunchKernelGGL(foci_small_batch_kernel, dim3(n_blocks), dim3(kSmallThreads), lds_small, stream, tab, seg, local, vals, pearson, min_size,
                       diag_only, flag, pos, keys_k, vals_k, parent, size, best_val, best_idx, s_rows, s_cols, s_size, n_kept, n_foci_blk,
                       lds_small ? 1 : 0, unsorted);
    hipLaunchKernelGGL(gather_foci_batch_kernel, dim3(1), dim3(kSmallThreads), 0, stream, seg, n_foci_blk, n_blocks, s_rows, s_cols, s_size,
                       f_rows, f_cols, f_size, f_blk, f_off, d_total, h_counts);
    if (n_total > 0) {
        const long long max_foci = std::max<long long>(1, n_total / std::max(min_size, 1));
        launch_focus_records(h_tab, n_blocks, max_foci, stream, tab, inter, f_blk, f_rows, f_cols, f_size, f_score, f_nobs, d_total, rec,
                             windows, win_cap, rec_cap);
    }
    return (int)hipGetLastError();
}

// ---- median of the stored values of a CSR view (inter-chromosomal blocks are scaled by it) -------
namespace {
__global__ __launch_bounds__(kThreads) void row_len_kernel(CsrView M, int* __restrict__ len)
{
    const int row = blockIdx.x * kThreads + threadIdx.x;
    if (row < M.n_rows) len[row] = (int)(M.row_end[row] - M.indptr[row]);
}

template <typename TV>
__global__ __launch_bounds__(kThreads) void gather_values_kernel(CsrView M, const int* __restrict__ off,
                                                                 double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TV* __restrict__ data = reinterpret_cast<const TV*>(M.data);
    for (int row = blockIdx.x * 4 + wv; row < M.n_rows; row += gridDim.x * 4) {
        const long long b = M.indptr[row], e = M.row_end[row];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = M.indices[k] - M.col0;
            double v = csr_value(M, data, k, row, col);
            if (v != v) v = 0.0;                               // NaN -> 0 (contacts_map.py:599)
            out[off[row] + (k - b)] = v;
        }
    }
}
}  // namespace

// Median of the values of the view (host result; synchronous).  `grow(user, bytes)` returns device
// scratch of at least `bytes` (it may move the block: the first pass is then simply repeated).
int csr_median(const CsrView& M, int n_cu, hipStream_t stream, void* (*grow)(void*, size_t), void* user, double* h_median)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    *h_median = __builtin_nan("");
    if (M.n_rows <= 0) return 0;
    size_t scan_tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int*)nullptr, (int*)nullptr, M.n_rows);
    const size_t head = 2 * al(4 * ((size_t)M.n_rows + 1));
    long long n_vals = -1;
    size_t have = 0;
    char* base = nullptr;
    for (int pass = 0; pass < 3; ++pass) {
        size_t sort_tmp = 0;
        if (n_vals > 0)
            (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const double*)nullptr, (double*)nullptr, (int)n_vals);
        const size_t tmp_bytes = al(std::max(scan_tmp, sort_tmp));
        const size_t need = head + tmp_bytes + (n_vals > 0 ? 2 * al(8 * (size_t)n_vals) : 0);
        char* p = (char*)grow(user, need);
        if (!p) return (int)hipErrorOutOfMemory;
        const bool moved = p != base || need > have;
        base = p;
        have = need;
        int* len = (int*)base;
        int* off = (int*)(base + al(4 * ((size_t)M.n_rows + 1)));
        void* tmp = base + head;
        if (n_vals < 0 || moved) {
            hipLaunchKernelGGL(row_len_kernel, dim3(blocks_for(M.n_rows)), dim3(kThreads), 0, stream, M, len);
            size_t tb = tmp_bytes;
            hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, len, off, M.n_rows, stream);
            if (e != hipSuccess) return (int)e;
        }
        if (n_vals < 0) {
            int last[2] = {0, 0};
            hipError_t e = hipMemcpyAsync(&last[0], off + (M.n_rows - 1), 4, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&last[1], len + (M.n_rows - 1), 4, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return (int)e;
            n_vals = (long long)last[0] + last[1];
            if (n_vals <= 0) return 0;
            continue;                                          // size the second pass
        }
        double* vals = (double*)(base + head + tmp_bytes);
        double* sorted = vals + al(8 * (size_t)n_vals) / 8;
        const int blocks = std::max(1, std::min((M.n_rows + 3) / 4, n_cu * 16));
        if (M.is_f64) hipLaunchKernelGGL(gather_values_kernel<double>, dim3(blocks), dim3(kThreads), 0, stream, M, off, vals);
        else hipLaunchKernelGGL(gather_values_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, M, off, vals);
        size_t tb = tmp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, tb, vals, sorted, (int)n_vals, 0, 64, stream);
        if (e != hipSuccess) return (int)e;
        double mid[2];
        e = hipMemcpyAsync(&mid[0], sorted + (n_vals - 1) / 2, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&mid[1], sorted + n_vals / 2, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return (int)e;
        *h_median = 0.5 * (mid[0] + mid[1]);                   // numpy: mean of the two middle values
        return 0;
    }
    return 0;
}

// The medians of SEVERAL views with two synchronisations in all (sizes, results) instead of two per view: a `quantify --inter`
// run stages every inter-chromosomal sub-matrix that holds a position (contacts_map.py:598-601), and seven medians one after
// the other were 1.1 of its 5.9 ms.  Same kernels, same values as csr_median.
int csr_median_many(const CsrView* views, int n, int n_cu, hipStream_t stream, void* (*grow)(void*, size_t), void* user, double* h_medians)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (n <= 0) return 0;
    std::vector<size_t> head_off((size_t)n);
    std::vector<long long> n_vals((size_t)n, 0);
    size_t heads = 0, scan_max = 0;
    for (int i = 0; i < n; ++i) {
        h_medians[i] = __builtin_nan("");
        head_off[(size_t)i] = heads;
        const int rows = std::max(views[i].n_rows, 0);
        heads += 2 * al(4 * ((size_t)rows + 1));
        if (rows > 0) {
            size_t t = 0;
            (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int*)nullptr, (int*)nullptr, rows);
            scan_max = std::max(scan_max, t);
        }
    }
    auto first_pass = [&](char* base, size_t tmp_bytes) -> hipError_t {
        for (int i = 0; i < n; ++i) {
            const CsrView& M = views[i];
            if (M.n_rows <= 0) continue;
            int* len = (int*)(base + head_off[(size_t)i]);
            int* off = (int*)(base + head_off[(size_t)i] + al(4 * ((size_t)M.n_rows + 1)));
            hipLaunchKernelGGL(row_len_kernel, dim3(blocks_for(M.n_rows)), dim3(kThreads), 0, stream, M, len);
            size_t tb = tmp_bytes;
            hipError_t e = hipcub::DeviceScan::ExclusiveSum(base + heads, tb, len, off, M.n_rows, stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    size_t tmp_bytes = al(scan_max);
    char* base = (char*)grow(user, heads + tmp_bytes);
    if (!base) return (int)hipErrorOutOfMemory;
    hipError_t e = first_pass(base, tmp_bytes);
    if (e != hipSuccess) return (int)e;
    std::vector<int> last(2 * (size_t)n, 0);
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        const CsrView& M = views[i];
        if (M.n_rows <= 0) continue;
        int* len = (int*)(base + head_off[(size_t)i]);
        int* off = (int*)(base + head_off[(size_t)i] + al(4 * ((size_t)M.n_rows + 1)));
        e = hipMemcpyAsync(&last[2 * (size_t)i], off + (M.n_rows - 1), 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&last[2 * (size_t)i + 1], len + (M.n_rows - 1), 4, hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return (int)e;
    size_t sort_max = 0, vals_bytes = 0;
    std::vector<size_t> vals_off((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        n_vals[(size_t)i] = (long long)last[2 * (size_t)i] + last[2 * (size_t)i + 1];
        if (n_vals[(size_t)i] <= 0) continue;
        size_t t = 0;
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, t, (const double*)nullptr, (double*)nullptr, (int)n_vals[(size_t)i]);
        sort_max = std::max(sort_max, t);
        vals_off[(size_t)i] = vals_bytes;
        vals_bytes += 2 * al(8 * (size_t)n_vals[(size_t)i]);
    }
    if (vals_bytes == 0) return 0;
    const size_t tmp2 = al(std::max(scan_max, sort_max));
    char* base2 = (char*)grow(user, heads + tmp2 + vals_bytes);
    if (!base2) return (int)hipErrorOutOfMemory;
    if (base2 != base) {                         // the block moved: the row offsets are rebuilt (their sizes are known now)
        base = base2;
        e = first_pass(base, tmp2);
        if (e != hipSuccess) return (int)e;
    }
    std::vector<double> mid(2 * (size_t)n, 0.0);
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        const CsrView& M = views[i];
        const long long nv = n_vals[(size_t)i];
        if (nv <= 0) continue;
        int* off = (int*)(base + head_off[(size_t)i] + al(4 * ((size_t)M.n_rows + 1)));
        double* vals = (double*)(base + heads + tmp2 + vals_off[(size_t)i]);
        double* sorted = vals + al(8 * (size_t)nv) / 8;
        const int blocks = std::max(1, std::min((M.n_rows + 3) / 4, n_cu * 16));
        if (M.is_f64) hipLaunchKernelGGL(gather_values_kernel<double>, dim3(blocks), dim3(kThreads), 0, stream, M, off, vals);
        else hipLaunchKernelGGL(gather_values_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, M, off, vals);
        size_t tb = tmp2;
        e = hipcub::DeviceRadixSort::SortKeys(base + heads, tb, vals, sorted, (int)nv, 0, 64, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&mid[2 * (size_t)i], sorted + (nv - 1) / 2, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&mid[2 * (size_t)i + 1], sorted + nv / 2, 8, hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < n; ++i)
        if (n_vals[(size_t)i] > 0) h_medians[i] = 0.5 * (mid[2 * (size_t)i] + mid[2 * (size_t)i + 1]);     // numpy: mean of the two middle values
    return 0;
}

// quantify mode, several sub-matrices in one chain: d_tab / d_inter (n_blocks entries, already on the device), the pixel list
// (d_blk, d_rows, d_cols) -> float64 scores, records and windows (device arrays of n entries)
int enqueue_quantify_batch(const CorrArgs<double>* d_tab, const int* d_inter, const int* d_blk, const int* d_rows, const int* d_cols,
                           long long n, double* d_score, double* d_nobs, FocusRec* d_rec, double* d_windows, hipStream_t stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(rescore_batch_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_tab, d_blk, d_rows, d_cols, n,
                       (const long long*)nullptr, d_score, d_nobs, fast_windows_on());
    hipLaunchKernelGGL(quantify_stats_batch_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kThreads), 0, stream, d_tab, d_inter, d_blk, d_rows,
                       d_cols, d_score, d_nobs, n, d_rec, d_windows);
    return (int)hipGetLastError();
}

// quantify mode: records at n given pixels (device arrays)
int enqueue_quantify(const CorrArgs<double>& A64, const int* d_rows, const int* d_cols, long long n, int inter,
                     double* d_score, double* d_nobs, FocusRec* d_rec, double* d_windows, hipStream_t stream)
{
    if (n == 0) return 0;
    int rc = launch_rescore_f64(A64, d_rows, d_cols, n, d_score, d_nobs, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(window_stats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kThreads), 0, stream, A64, inter, d_rows,
                       d_cols, (const int*)nullptr, d_score, d_nobs, (const long long*)nullptr, n, d_rec, d_windows, n, n,
                       (long long*)nullptr);
    return (int)hipGetLastError();
}

}  // namespace cs
