// cs_foci_kernels.h -- the device kernels of cs_foci.hip (focus picking, exact re-scoring, window statistics, records), included
// by that file alone: one translation unit, kept in two files of a readable size.  Host orchestration: cs_foci.hip.
#pragma once

namespace {

constexpr int kThreads = 256;

inline unsigned blocks_for(long long n) { return (unsigned)((n + kThreads - 1) / kThreads); }

// ---- candidate keys ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void make_keys_kernel(const int* __restrict__ rows, const int* __restrict__ cols,
                                                             long long n, int ns, long long* __restrict__ keys)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < n) keys[t] = (long long)rows[t] * ns + cols[t];
}

__global__ __launch_bounds__(kThreads) void decode_keys_kernel(const long long* __restrict__ keys, long long n, int ns,
                                                               int* __restrict__ rows, int* __restrict__ cols)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < n) {
        const long long k = keys[t];
        rows[t] = (int)(k / ns);
        cols[t] = (int)(k - (k / ns) * ns);
    }
}

// candidates that pass the exact threshold: value >= pearson and != 0 (detection.py:417-421)
__global__ __launch_bounds__(kThreads) void flag_keep_kernel(const double* __restrict__ vals, long long n, double pearson,
                                                             int* __restrict__ flag)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < n) flag[t] = (vals[t] >= pearson && vals[t] != 0.0) ? 1 : 0;
}

// stable compaction by the exclusive scan of the flags; the last thread publishes the count
__global__ __launch_bounds__(kThreads) void scatter_keep_kernel(const long long* __restrict__ keys,
                                                                const double* __restrict__ vals,
                                                                const int* __restrict__ flag, const int* __restrict__ pos,
                                                                long long n, long long* __restrict__ keys2,
                                                                double* __restrict__ vals2, int* __restrict__ n_kept)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    if (flag[t]) {
        keys2[pos[t]] = keys[t];
        vals2[pos[t]] = vals[t];
    }
    if (t == n - 1) *n_kept = pos[t] + flag[t];
}

// ---- union-find over the candidate list (indices = row-major rank of the pixel) -----------------
__device__ __forceinline__ int uf_load(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

__device__ __forceinline__ int uf_find(int* parent, int x)
{
    int p = uf_load(parent + x);
    while (p != x) {
        const int gp = uf_load(parent + p);
        if (gp != p) __atomic_store_n(parent + x, gp, __ATOMIC_RELAXED);   // path halving (benign race)
        x = p;
        p = gp;
    }
    return x;
}

// the larger root is hooked under the smaller one, so the root of a focus is its first pixel in
// row-major order -- the reference numbers foci in that order (label_foci)
__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(parent + a, a, b) == a) return;
    }
}

__global__ __launch_bounds__(kThreads) void init_focus_kernel(const int* __restrict__ n_ptr, int* __restrict__ parent,
                                                              int* __restrict__ size, unsigned long long* __restrict__ best_val,
                                                              int* __restrict__ best_idx, long long cap)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < cap && t < *n_ptr) {
        parent[t] = (int)t;
        size[t] = 0;
        best_val[t] = 0ull;
        best_idx[t] = INT_MAX;
    }
}

__global__ __launch_bounds__(kThreads) void link_kernel(const long long* __restrict__ keys, const int* __restrict__ n_ptr,
                                                        int ns, int* __restrict__ parent)
{
    const int n = *n_ptr;
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    const long long key = keys[t];
    const int col = (int)(key % ns);
    // right neighbour: the next candidate, if it is the next pixel of the same row
    if (t + 1 < n && col + 1 < ns && keys[t + 1] == key + 1) uf_union(parent, (int)t, (int)t + 1);
    // lower neighbour: the candidate with key + ns, if any
    long long lo = t + 1, hi = n;
    const long long want = key + ns;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    if (lo < n && keys[lo] == want) uf_union(parent, (int)t, (int)lo);
}

__global__ __launch_bounds__(kThreads) void flatten_kernel(const int* __restrict__ n_ptr, int* __restrict__ parent)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t < *n_ptr) parent[t] = uf_find(parent, (int)t);
}

// order-preserving map of a double onto unsigned integers (for atomicMax)
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(kThreads) void focus_stats_kernel(const int* __restrict__ n_ptr, const int* __restrict__ parent,
                                                               const double* __restrict__ vals, int* __restrict__ size,
                                                               unsigned long long* __restrict__ best_val)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= *n_ptr) return;
    const int r = parent[t];
    atomicAdd(size + r, 1);
    atomicMax(best_val + r, order_key(vals[t]));
}

// first pixel (row-major) holding the focus' maximum: np.argmax over the focus' pixels in
// coordinate order (detection.py:446-449)
__global__ __launch_bounds__(kThreads) void focus_argbest_kernel(const int* __restrict__ n_ptr, const int* __restrict__ parent,
                                                                 const double* __restrict__ vals,
                                                                 const unsigned long long* __restrict__ best_val,
                                                                 int* __restrict__ best_idx)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= *n_ptr) return;
    const int r = parent[t];
    if (order_key(vals[t]) == best_val[r]) atomicMin(best_idx + r, (int)t);
}

__global__ __launch_bounds__(kThreads) void flag_roots_kernel(const int* __restrict__ n_ptr, const int* __restrict__ parent,
                                                              const int* __restrict__ size, int min_size, int* __restrict__ flag,
                                                              long long cap)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= cap) return;
    flag[t] = (t < *n_ptr && parent[t] == (int)t && size[t] >= min_size) ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void emit_foci_kernel(const int* __restrict__ n_ptr, const int* __restrict__ flag,
                                                             const int* __restrict__ pos, const int* __restrict__ best_idx,
                                                             const int* __restrict__ size, const long long* __restrict__ keys,
                                                             int ns, int diag_only, int* __restrict__ out_rows,
                                                             int* __restrict__ out_cols, int* __restrict__ out_size,
                                                             long long cap, long long* __restrict__ n_foci)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= cap) return;
    const int n = *n_ptr;
    if (t < n && flag[t]) {
        const long long key = keys[best_idx[t]];
        int row = (int)(key / ns);
        const int col = (int)(key - (long long)row * ns);
        if (diag_only) row = col + (diag_only >> 1);   // odd code: 1-D pattern, row offset in the upper bits
        out_rows[pos[t]] = row;
        out_cols[pos[t]] = col;
        out_size[pos[t]] = size[t];
    }
    if (t == cap - 1) *n_foci = pos[t] + flag[t];     // flags are 0 beyond n, so this is the total
}

// ---- short candidate lists: threshold, compaction, labelling, per-focus maximum in ONE launch -----
// A detect call on one sub-matrix is a chain of ~35 dependent launches that process a few thousand
// candidates; each of those kernels lasts 4-5 us however little it does, so 23 sub-matrices x 4 templates
// were bound by the launch chain (borders: 4 ms per template for 1.2 Mpixel).  One workgroup of 1024
// threads walks the same phases over the same global arrays with a workgroup barrier between them.
constexpr int kSmallThreads = 1024;
constexpr long long kSmallMax = 1 << 16;          // candidates a single workgroup takes (64 per thread)

// exclusive scan of flag[0..n) into pos[0..n), total returned to every thread; chunk per thread
__device__ __forceinline__ int block_scan_flags(const int* __restrict__ flag, int* __restrict__ pos, long long n,
                                                int* __restrict__ lds_part)
{
    const int tid = threadIdx.x;
    const long long chunk = (n + kSmallThreads - 1) / kSmallThreads;
    const long long b = min(n, (long long)tid * chunk), e = min(n, b + chunk);
    int cnt = 0;
    for (long long t = b; t < e; ++t) cnt += flag[t];
    lds_part[tid] = cnt;
    __syncthreads();
    // Hillis-Steele over the 1024 partial counts
    for (int off = 1; off < kSmallThreads; off <<= 1) {
        const int add = tid >= off ? lds_part[tid - off] : 0;
        __syncthreads();
        lds_part[tid] += add;
        __syncthreads();
    }
    const int total = lds_part[kSmallThreads - 1];
    int run = lds_part[tid] - cnt;
    for (long long t = b; t < e; ++t) {
        pos[t] = run;
        run += flag[t];
    }
    __syncthreads();
    return total;
}

__device__ __forceinline__ void foci_small_body(
    const long long* __restrict__ keys_s, const double* __restrict__ vals, long long n_cand, double pearson, int ns,
    int min_size, int diag_only, int* __restrict__ flag, int* __restrict__ pos, long long* __restrict__ keys_k,
    double* __restrict__ vals_k, int* __restrict__ parent, int* __restrict__ size, unsigned long long* __restrict__ best_val,
    int* __restrict__ best_idx, int* __restrict__ out_rows, int* __restrict__ out_cols, int* __restrict__ out_size,
    int* __restrict__ n_kept_out, long long* __restrict__ n_foci, int* __restrict__ part)
{
    const int tid = threadIdx.x;
    // (1) exact threshold (detection.py:417-421), stable compaction
    for (long long t = tid; t < n_cand; t += kSmallThreads) flag[t] = (vals[t] >= pearson && vals[t] != 0.0) ? 1 : 0;
    __syncthreads();
    const int n = block_scan_flags(flag, pos, n_cand, part);
    for (long long t = tid; t < n_cand; t += kSmallThreads)
        if (flag[t]) {
            keys_k[pos[t]] = keys_s[t];
            vals_k[pos[t]] = vals[t];
        }
    if (tid == 0) *n_kept_out = n;
    // (2) 4-connected foci of the kept pixels: union-find, root = first pixel in row-major order
    for (int t = tid; t < n; t += kSmallThreads) {
        parent[t] = t;
        size[t] = 0;
        best_val[t] = 0ull;
        best_idx[t] = INT_MAX;
    }
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) {
        const long long key = keys_k[t];
        const int col = (int)(key % ns);
        if (t + 1 < n && col + 1 < ns && keys_k[t + 1] == key + 1) uf_union(parent, t, t + 1);
        // lower neighbour (key + ns): the list is sorted, and in both regimes it sits -- or is seen to be absent -- within a
        // few entries: a narrow scan (2 diagonals) keeps it 1-2 entries ahead, a sparse list jumps past it at once.  The
        // binary search (15 dependent loads) is only the fallback.
        const long long want = key + ns;
        long long lo = t + 1, hi = n;
        bool decided = false;
#pragma unroll
        for (int s = 1; s <= 4 && !decided; ++s) {
            if (t + s >= n) {
                decided = true;
                lo = n;
            } else {
                const long long k = keys_k[t + s];
                if (k >= want) {
                    decided = true;
                    lo = t + s;
                }
            }
        }
        if (!decided) {
            lo = t + 5;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (keys_k[mid] < want) lo = mid + 1;
                else hi = mid;
            }
        }
        if (lo < n && keys_k[lo] == want) uf_union(parent, t, (int)lo);
    }
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) parent[t] = uf_find(parent, t);
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) {
        const int r = parent[t];
        atomicAdd(size + r, 1);
        atomicMax(best_val + r, order_key(vals_k[t]));
    }
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) {
        const int r = parent[t];
        if (order_key(vals_k[t]) == best_val[r]) atomicMin(best_idx + r, t);
    }
    __syncthreads();
    // (3) foci of at least min_size pixels, in the order of their first pixels
    for (int t = tid; t < n; t += kSmallThreads) flag[t] = (parent[t] == t && size[t] >= min_size) ? 1 : 0;
    __syncthreads();
    const int n_out = block_scan_flags(flag, pos, n, part);
    for (int t = tid; t < n; t += kSmallThreads)
        if (flag[t]) {
            const long long key = keys_k[best_idx[t]];
            int row = (int)(key / ns);
            const int col = (int)(key - (long long)row * ns);
            if (diag_only) row = col + (diag_only >> 1);   // odd code: 1-D pattern, row offset in the upper bits
            out_rows[pos[t]] = row;
            out_cols[pos[t]] = col;
            out_size[pos[t]] = size[t];
        }
    if (tid == 0) *n_foci = n_out;
}

__global__ __launch_bounds__(kSmallThreads) void foci_small_kernel(
    const long long* __restrict__ keys_s, const double* __restrict__ vals, long long n_cand, double pearson, int ns,
    int min_size, int diag_only, int* __restrict__ flag, int* __restrict__ pos, long long* __restrict__ keys_k,
    double* __restrict__ vals_k, int* __restrict__ parent, int* __restrict__ size, unsigned long long* __restrict__ best_val,
    int* __restrict__ best_idx, int* __restrict__ out_rows, int* __restrict__ out_cols, int* __restrict__ out_size,
    int* __restrict__ n_kept_out, long long* __restrict__ n_foci)
{
    __shared__ int part[kSmallThreads];
    foci_small_body(keys_s, vals, n_cand, pearson, ns, min_size, diag_only, flag, pos, keys_k, vals_k, parent, size, best_val,
                    best_idx, out_rows, out_cols, out_size, n_kept_out, n_foci, part);
}

// ---- 1-D patterns of MANY sub-matrices with one launch chain (cs_detect_foci_batch) ----------------
// A borders template scans 2 diagonals of every chromosome: per sub-matrix that is ~0.1 ms of tiny kernels
// and a host synchronisation, and the calls of different host threads hardly overlap on the GPU.  Here the
// candidate lists of all sub-matrices are one array with segment offsets; every kernel finds its
// sub-matrix's arguments in a device table.
__device__ __forceinline__ int segment_of(const long long* __restrict__ seg, int n_seg, long long t)
{
    int lo = 0, hi = n_seg - 1;                  // seg[lo] <= t < seg[hi + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads) void narrow_enumerate_batch_kernel(const CorrArgs<double>* __restrict__ tab,
                                                                         const long long* __restrict__ seg,
                                                                         const int2* __restrict__ lo_w, int n_blocks,
                                                                         int* __restrict__ rows, int* __restrict__ cols,
                                                                         long long* __restrict__ keys, int* __restrict__ blk)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= seg[n_blocks]) return;
    const int b = segment_of(seg, n_blocks, t);
    const int rb = tab[b].row_begin, re = tab[b].row_end, ns = tab[b].ns, lo = lo_w[b].x, w = lo_w[b].y;
    const long long local = t - seg[b];
    const int last_full = min(re, ns - lo - w + 1);
    const long long full = (long long)max(last_full - rb, 0) * w;
    int row, x;
    if (local < full) {
        row = rb + (int)(local / w);
        x = (int)(local - (long long)(row - rb) * w);
    } else {
        long long rest = local - full;
        row = max(last_full, rb);
        for (;;) {
            const int cnt = max(0, min(w, ns - row - lo));
            if (rest < cnt) break;
            rest -= cnt;
            ++row;
        }
        x = (int)rest;
    }
    rows[t] = row;
    cols[t] = row + lo + x;
    keys[t] = (long long)row * ns + (row + lo + x);
    blk[t] = b;
}

// CHROMOSIGHT_HIP_NO_FAST_WINDOWS=1: the wave-per-window kernels keep the general functions (rescore_pixel, lazy_gather_window)
// where the compile-time-size ones of cs_launch_aux.h apply -- the two are the same sums in the same order
static bool fast_windows_on()
{
    return std::getenv("CHROMOSIGHT_HIP_NO_FAST_WINDOWS") == nullptr;     // (read per launch: the switch test flips it in-process)
}

__global__ __launch_bounds__(256) void rescore_batch_kernel(const CorrArgs<double>* __restrict__ tab, const int* __restrict__ blk,
                                                            const int* __restrict__ rows, const int* __restrict__ cols,
                                                            long long n_px, const long long* __restrict__ n_ptr,
                                                            double* __restrict__ out_corr, double* __restrict__ out_nobs,
                                                            bool fast_windows)
{
    __shared__ double lazy_win[4][kLazyWinMax];             // (lazily evaluated float64 bands: the wave's window)
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_px || (n_ptr && t >= *n_ptr)) return;
    double r, nobs;
    const CorrArgs<double>& A = tab[blk[t]];
    const int oi = rows[t], oj = cols[t];
    const bool inside = (oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns);
    if (inside && fast_windows && window_fast_applies<17>(A)) {                  // wave-uniform
        // the detection configuration with a 17 x 17 template (cs_launch_aux.h rescore_pixel_sq: same sums, ~ 1/2 of the
        // instructions); CHROMOSIGHT_HIP_NO_FAST_WINDOWS=1 keeps the general functions (tested bit-equal)
        double* win = nullptr;
        if (A.sig.layout == 2) {
            win = lazy_win[threadIdx.x >> 6];
            lazy_gather_window_sq<17>(A, oi - 8, oj - 8, lane, win);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        rescore_pixel_sq<17>(A, oi, oj, lane, r, nobs, win);
    } else if (A.sig.layout == 2 && lazy_window_fits(A.km, A.kn) && inside) {     // wave-uniform
        double* win = lazy_win[threadIdx.x >> 6];
        lazy_gather_window(A, oi - (A.km - 1) / 2, oj - (A.kn - 1) / 2, lane, win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        rescore_pixel(A, oi, oj, lane, r, nobs, win);
    } else {
        rescore_pixel(A, oi, oj, lane, r, nobs);
    }
    if (lane == 0) {
        out_corr[t] = r;
        if (out_nobs) out_nobs[t] = nobs;
    }
}

// rescore_batch_kernel when EVERY sub-matrix of the table is in the detection configuration with a 17 x 17 template on a lazily
// evaluated band (checked on the host, launch_rescore_batch below) -- what the 2-D chain of a genome step runs.  Nothing but the
// compile-time-size functions is compiled in: the general kernel's 92 vector registers (its runtime-size paths) hold five waves
// per SIMD, and a window's evaluation is a chain of dependent round trips (row search, entries, values) that only other waves hide.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void rescore_batch_lazy17_kernel(const CorrArgs<double>* __restrict__ tab, const int* __restrict__ blk,
                                                                   const int* __restrict__ rows, const int* __restrict__ cols,
                                                                   long long n_px, const long long* __restrict__ n_ptr,
                                                                   double* __restrict__ out_corr, double* __restrict__ out_nobs)
{
    __shared__ double lazy_win[4][kLazyWinMax];
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_px || (n_ptr && t >= *n_ptr)) return;
    const CorrArgs<double>& A = tab[blk[t]];
    const int oi = rows[t], oj = cols[t];
    double r = 0.0, nobs = A.ks.n;
    if ((oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns)) {                      // wave-uniform
        double* win = lazy_win[threadIdx.x >> 6];
        lazy_gather_window_sq<17>(A, oi - 8, oj - 8, lane, win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        rescore_pixel_sq<17>(A, oi, oj, lane, r, nobs, win);
    }
    if (lane == 0) {
        out_corr[t] = r;
        if (out_nobs) out_nobs[t] = nobs;
    }
}

// h_tab: the host copy of the argument table (or nullptr: unknown -> the general kernel)
static void launch_rescore_batch(const CorrArgs<double>* h_tab, int n_blocks, long long n_waves, hipStream_t stream,
                                 const CorrArgs<double>* tab, const int* blk, const int* rows, const int* cols, long long n_px,
                                 const long long* n_ptr, double* out_corr, double* out_nobs)
{
    const bool fast = fast_windows_on();
    bool lazy17 = fast && h_tab != nullptr && n_blocks > 0;
    for (int b = 0; lazy17 && b < n_blocks; ++b) {
        const CorrArgs<double>& A = h_tab[b];
        lazy17 = A.km == 17 && A.kn == 17 && A.mask_mode == 1 && A.sym_upper && A.full && A.max_dist >= 0 && A.sig.layout == 2;
    }
    const dim3 grid((unsigned)((n_waves + 3) / 4));
    if (lazy17)
        hipLaunchKernelGGL(rescore_batch_lazy17_kernel, grid, dim3(256), 0, stream, tab, blk, rows, cols, n_px, n_ptr, out_corr, out_nobs);
    else
        hipLaunchKernelGGL(rescore_batch_kernel, grid, dim3(256), 0, stream, tab, blk, rows, cols, n_px, n_ptr, out_corr, out_nobs, fast);
}

// lane-per-pixel version for the enumerated diagonals (rescore_pixel_lane).  A workgroup's 256 list entries are
// normally a run of 256 / w rows on w neighbouring diagonals of one sub-matrix: the rows and diagonals their
// windows reach are staged in LDS once (coalesced) and every lane walks its window there -- straight from memory
// the 64 lanes of a wave read 64 different rows per instruction and 24 waves per CU thrash L1 (measured 485 us for
// 400 000 pixels; the wave-per-pixel kernel 714).  Workgroups that straddle two sub-matrices or whose entries are
// not such a run take the direct route, a wave that straddles sub-matrices one after the other.
constexpr int kRunWeights = 3 * 17 * 17;               // the three weight sets of a template of up to 289 entries
constexpr int kRunStageMax = 24;                       // pieces of 8 rows x 8 diagonals a wave stages at once (run17: 90 pieces per tile)

// dynamic LDS: tile (tile_cap doubles), then the weights, then the row / column flags
// ONLY17: nothing but the run17 route (and the direct one as its fall-back) is compiled in -- what the 1-D patterns of a genome
// take.  The general instance carries every route (runtime sizes, 15 x 15, edge and interior forms) and with them 241 vector
// registers: two workgroups per CU where the LDS would hold three, and the kernel is bound by its workgroups' latency.
template <bool ONLY17>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ONLY17 ? 3 : 1, ONLY17 ? 3 : 8))) void rescore_run_batch_kernel(const CorrArgs<double>* __restrict__ tab, const int* __restrict__ blk,
                                                                const int* __restrict__ rows, const int* __restrict__ cols,
                                                                long long n_px, double* __restrict__ out_corr, int tile_cap,
                                                                int no_run17, bool fast_windows, int n_fuse, int tab_step,
                                                                long long out_stride)
{
    // n_fuse > 1 (ONLY17 instance): the list holds the same pixels once per template -- entries t + k out_stride, argument blocks
    // tab[b + k tab_step], k < n_fuse -- and the launch covers the first template's entries: a workgroup stages its tile once and
    // every lane evaluates its pixel under all the templates (rescore_run17_multi)
    const bool RUN_NO_FAST = no_run17 != 0;
    extern __shared__ __attribute__((aligned(16))) double run_smem[];
    double* const tile = run_smem;
    double* const wl = run_smem + tile_cap;
    unsigned char* const rfl = reinterpret_cast<unsigned char*>(wl + kRunWeights);       // 512 row flags, 1024 column flags
    unsigned char* const cfl = rfl + 512;
    __shared__ int red[8];
    __shared__ int part[4][6];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long t = (long long)blockIdx.x * blockDim.x + tid;
    const bool valid = t < n_px;
    const int b = valid ? blk[t] : -1;
    const int oi = valid ? rows[t] : 0, oj = valid ? cols[t] : 0;
    // extent (rows, diagonals, sub-matrices) of the entries selected by `sel`: wave reductions, then 4 partial results each
    auto extent = [&](bool sel) {
        int v_lo[3] = {sel ? oi : 0x7fffffff, sel ? oj - oi : 0x7fffffff, sel ? b : 0x7fffffff};
        int v_hi[3] = {sel ? oi : -0x7fffffff, sel ? oj - oi : -0x7fffffff, sel ? b : -0x7fffffff};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                v_lo[k] = min(v_lo[k], __shfl_xor(v_lo[k], o));
                v_hi[k] = max(v_hi[k], __shfl_xor(v_hi[k], o));
            }
        }
        __syncthreads();                                      // (the previous round's readers of red / part are done)
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                part[tid >> 6][2 * k] = v_lo[k];
                part[tid >> 6][2 * k + 1] = v_hi[k];
            }
        }
        __syncthreads();
        if (tid < 6) {
            int v = part[0][tid];
            for (int w = 1; w < 4; ++w) v = (tid & 1) ? max(v, part[w][tid]) : min(v, part[w][tid]);
            red[tid] = v;
        }
        __syncthreads();
    };
    extent(valid);
    const int b_lo = __builtin_amdgcn_readfirstlane(red[4]), b_hi = __builtin_amdgcn_readfirstlane(red[5]);   // (uniform: the
                                                              // sub-matrix's table entry is read through the scalar unit)
    double r = 0.0, nobs = 0.0;
    // A workgroup's 256 entries normally belong to one sub-matrix; the few that straddle two (or more, tiny ones) take them
    // one after the other, each with its own tile -- left to the direct route, one such workgroup (289 dependent global
    // loads per pixel) lasted as long as the rest of the launch together.
    for (int bb = b_lo; bb <= b_hi && b_lo >= 0; ++bb) {
        const bool mine = valid && b == bb;
        if (b_lo != b_hi) extent(mine);
        const int i_lo = red[0], i_hi = red[1], d_lo = red[2], d_hi = red[3];
        if (i_hi < i_lo) continue;                            // no entry of this sub-matrix here (uniform)
        const CorrArgs<double>& A = tab[bb];
        const int km = A.km, kn = A.kn, kh = (km - 1) / 2, kw = (kn - 1) / 2;
        const int P0 = i_lo - kh, RN = (i_hi - i_lo) + km;
        const int D0 = d_lo - kw - (km - 1 - kh), DN = (d_hi - d_lo) + (kn - 1) + (km - 1) + 1;
        const int C0 = P0 + D0, CN = RN + DN;                  // columns the tile can reach
        if (tile_cap > 0 && (long long)RN * DN <= tile_cap && RN <= 512 && CN <= 1024 && 3 * km * kn <= kRunWeights) {
            if constexpr (!ONLY17)
                for (int idx = tid; idx < 3 * km * kn; idx += 256) wl[idx] = A.w[idx];         // (run17: weights through the scalar unit)
            const bool bins = A.mask_mode == 1;
            for (int idx = tid; idx < RN; idx += 256) {
                const int p = P0 + idx;
                rfl[idx] = (bins && p >= 0 && p < A.ms) ? A.miss_row[p] : 0;
            }
            for (int idx = tid; idx < CN; idx += 256) {
                const int q = C0 + idx;
                cfl[idx] = (bins && q >= 0 && q < A.ns) ? A.miss_col[q] : 0;
            }
            // detection configuration, square template: the branch-free forms (every window of the run inside the matrix,
            // or not)
            const bool lean = bins && A.sym_upper && A.full && A.max_dist >= 0 && km == kn;
            const bool interior = P0 >= 0 && P0 + RN <= A.ms && C0 >= 0 && C0 + CN <= A.ns;
            // runs on one or two diagonals under a 17 x 17 template (1-D patterns): the transposed tile of rescore_run17
            const bool run17 = lean && kn == 17 && d_hi - d_lo <= 1 && RN <= kRunRP && DN * kRunRP <= tile_cap && !RUN_NO_FAST;
            // (the signal through a SigReader: the argument block's and the lazily evaluated band's fields are read once, not
            // behind every LDS store of the loops below; CHROMOSIGHT_HIP_NO_FAST_WINDOWS=1 keeps load_signal)
            const SigReader S(A);
            if (run17) {
                // 8 rows x 8 diagonals per wave and step: 64-byte pieces of band rows in, 8 consecutive doubles of 8 tile
                // rows out
                const int n_dc = (DN + 7) >> 3, n_ch = ((RN + 7) >> 3) * n_dc;
                if (fast_windows && n_ch <= 4 * kRunStageMax) {
                    // every piece of the wave requested before the first one is stored (a tile is 90 pieces, 23 per wave: in
                    // batches of four the wave waited six times for a round trip)
                    double got[kRunStageMax];
#pragma unroll
                    for (int k = 0; k < kRunStageMax; ++k) {
                        const int ch = (tid >> 6) + 4 * k;
                        const int cr = ch / n_dc, cd = ch - cr * n_dc;
                        const int rr = 8 * cr + (lane >> 3), dd = 8 * cd + (lane & 7);
                        const int p = P0 + rr;
                        got[k] = (ch < n_ch && rr < RN && dd < DN) ? S.at(p, p + D0 + dd) : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < kRunStageMax; ++k) {
                        const int ch = (tid >> 6) + 4 * k;
                        const int cr = ch / n_dc, cd = ch - cr * n_dc;
                        const int rr = 8 * cr + (lane >> 3), dd = 8 * cd + (lane & 7);
                        if (ch < n_ch && rr < RN && dd < DN) tile[dd * kRunRP + rr] = got[k];
                    }
                } else if (ONLY17 || fast_windows) {
#pragma unroll 4
                    for (int ch = tid >> 6; ch < n_ch; ch += 4) {
                        const int cr = ch / n_dc, cd = ch - cr * n_dc;
                        const int rr = 8 * cr + (lane >> 3), dd = 8 * cd + (lane & 7);
                        const int p = P0 + rr;
                        if (rr < RN && dd < DN) tile[dd * kRunRP + rr] = S.at(p, p + D0 + dd);
                    }
                } else if constexpr (!ONLY17) {
                    for (int ch = tid >> 6; ch < n_ch; ch += 4) {
                        const int cr = ch / n_dc, cd = ch - cr * n_dc;
                        const int rr = 8 * cr + (lane >> 3), dd = 8 * cd + (lane & 7);
                        const int p = P0 + rr;
                        if (rr < RN && dd < DN) tile[dd * kRunRP + rr] = load_signal(A, p, p + D0 + dd);
                    }
                }
            } else if constexpr (ONLY17) {
                // (not a run the transposed tile holds: the direct route below)
            } else if (fast_windows) {
#pragma unroll 4
                for (int idx = tid; idx < RN * DN; idx += 256) {
                    const int rr = idx / DN, dd = idx - rr * DN;
                    const int p = P0 + rr;
                    tile[idx] = S.at(p, p + D0 + dd);
                }
            } else {
                for (int idx = tid; idx < RN * DN; idx += 256) {
                    const int rr = idx / DN, dd = idx - rr * DN;
                    const int p = P0 + rr;
                    tile[idx] = load_signal(A, p, p + D0 + dd);
                }
            }
            __syncthreads();
            if (run17) {
                if (mine) {
#define CS_RUN17_MULTI(T_)                                                                                        \
    {                                                                                                             \
        double rk[T_];                                                                                            \
        if (interior) rescore_run17_multi<false, T_>(A, tab_step, tile, rfl, cfl, P0, C0, D0, oi, oj, rk);        \
        else rescore_run17_multi<true, T_>(A, tab_step, tile, rfl, cfl, P0, C0, D0, oi, oj, rk);                  \
        r = rk[0];                                                                                                \
        for (int k = 1; k < T_; ++k) out_corr[t + k * out_stride] = rk[k];                                        \
    }
                    if (ONLY17 && n_fuse == 3) CS_RUN17_MULTI(3)
                    else if (ONLY17 && n_fuse == 2) CS_RUN17_MULTI(2)
                    else if (interior) rescore_run17<false>(A, tile, rfl, cfl, P0, C0, D0, oi, oj, r, nobs);
                    else rescore_run17<true>(A, tile, rfl, cfl, P0, C0, D0, oi, oj, r, nobs);
#undef CS_RUN17_MULTI
                }
            } else if constexpr (ONLY17) {
                if (mine) {
                    rescore_pixel_lane(A, oi, oj, r, nobs);
                    for (int k = 1; k < n_fuse; ++k) {
                        double rr, nn;
                        rescore_pixel_lane((&A)[k * tab_step], oi, oj, rr, nn);
                        out_corr[t + k * out_stride] = rr;
                    }
                }
            } else if (mine && lean && interior && kn == 17) rescore_pixel_lane_lds_interior<17>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
            else if (mine && lean && interior && kn == 15) rescore_pixel_lane_lds_interior<15>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
            else if (mine && lean && kn == 17) rescore_pixel_lane_lds_interior<17, true>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
            else if (mine && lean && kn == 15) rescore_pixel_lane_lds_interior<15, true>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
            else if (mine) {
                if (kn == 17) rescore_pixel_lane_lds<17>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
                else if (kn == 15) rescore_pixel_lane_lds<15>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
                else rescore_pixel_lane_lds<0>(A, tile, wl, rfl, cfl, P0, C0, D0, DN, oi, oj, r, nobs);
            }
            __syncthreads();                                  // the tile is reused by the next sub-matrix
        } else if (mine) {
            rescore_pixel_lane(A, oi, oj, r, nobs);
            for (int k = 1; k < n_fuse; ++k) {
                double rr, nn;
                rescore_pixel_lane((&A)[k * tab_step], oi, oj, rr, nn);
                out_corr[t + k * out_stride] = rr;
            }
        }
    }
    if (valid) out_corr[t] = r;
}

// The same phases with the working arrays in LDS, for lists of up to kSmallLds candidates of a sub-matrix whose keys fit
// 32 bits -- the blocks of a genome scan: ~ 4 000 candidates each.  foci_small_body walks global arrays with a workgroup
// barrier between its dozen phases and a 10-step scan (20 barriers) twice: ~ 95 us however short the list, all of it on the
// critical path behind the tile kernels.  Here the keys, the union-find forest, the sizes and the arg-maxima live in LDS
// (18 bytes per candidate), compaction and emission take their positions from a shuffle scan of per-thread counts (two
// barriers), and global memory is read for the values and written for the foci only.  Same algorithm, same order: the
// root of a focus is its first pixel in row-major order, ties of the maximum go to the first pixel.
constexpr int kSmallLds = 8192;
constexpr size_t kSmallLdsBytes = (size_t)kSmallLds * (4 + 4 + 4 + 4 + 2);

// exclusive prefix of `cnt` over the workgroup's 1024 threads and the total; part: 32 ints
__device__ __forceinline__ int block_scan_counts(int cnt, int* __restrict__ part, int& total)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    __syncthreads();                                      // (readers of the previous scan's partials are done)
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        int v = lane < 16 ? part[lane] : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int up = __shfl_up(v, off);
            if (lane >= off) v += up;
        }
        if (lane < 16) part[16 + lane] = v;
    }
    __syncthreads();
    total = part[31];
    return (wv ? part[16 + wv - 1] : 0) + incl - cnt;
}

// (key, source) pairs of the kept pixels into ascending key order, in LDS: bitonic network over the next power of two (the
// padding holds the largest key).  For lists that arrive UNSORTED -- the candidates of a block as the tile kernel appended them,
// cs_detect_foci_blocks' segmented lists -- instead of a device-wide sort of all blocks' candidates in front of the chain
// (five launches and ~ 40 us of a rank's step for ~ 4 000 candidates per block, which one workgroup sorts in ~ 5 us).
__device__ __forceinline__ void lds_sort_pairs(unsigned* __restrict__ key, unsigned short* __restrict__ src, int n)
{
    // The all-ascending form of the network: the first stage of a merge pairs i with its mirror image in the block of k
    // (i ^ (k - 1)), the later ones i with i ^ j, every exchange puts the smaller key first.  Elements beyond n would be
    // +infinity and never move, so pairs that reach beyond n are skipped and nothing is padded: a list of 2 400 kept pixels
    // costs 0.6 of the 4 096 its power of two would.
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool flip = j == (k >> 1);
            for (int q = threadIdx.x; q < (P >> 1); q += kSmallThreads) {      // pair q: i has bit j clear
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int l = flip ? (i ^ (k - 1)) : (i | j);
                if (l < n) {
                    const unsigned a = key[i], b = key[l];
                    if (a > b) {
                        key[i] = b;
                        key[l] = a;
                        const unsigned short sa = src[i];
                        src[i] = src[l];
                        src[l] = sa;
                    }
                }
            }
            // pairs at distance <= 64 (blocks of k <= 128 in a flip stage) stay inside the 128 elements a wave's 64 consecutive
            // pairs span: between such stages only the wave's own LDS operations must be ordered (they execute in order); the
            // workgroup meets where the distance grows again (j == 1: the next level starts with a long one) or is still long
            if (j > 64 || j == 1) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
}

__device__ __forceinline__ void foci_small_lds_body(
    const long long* __restrict__ keys_s, const double* __restrict__ vals, int n_cand, double pearson, unsigned ns, int min_size,
    int diag_only, unsigned long long* __restrict__ best_val, int* __restrict__ out_rows, int* __restrict__ out_cols,
    int* __restrict__ out_size, int* __restrict__ n_kept_out, long long* __restrict__ n_foci, int* __restrict__ part, char* lds,
    int unsorted = 0)
{
    unsigned* s_key = reinterpret_cast<unsigned*>(lds);
    int* s_parent = reinterpret_cast<int*>(lds + 4 * (size_t)kSmallLds);
    int* s_size = reinterpret_cast<int*>(lds + 8 * (size_t)kSmallLds);
    int* s_best = reinterpret_cast<int*>(lds + 12 * (size_t)kSmallLds);
    unsigned short* s_src = reinterpret_cast<unsigned short*>(lds + 16 * (size_t)kSmallLds);
    const int tid = threadIdx.x;
    // (1) exact threshold (detection.py:417-421), stable compaction: thread tid owns the candidates [b, e)
    const int chunk = (n_cand + kSmallThreads - 1) / kSmallThreads;
    const int b = min(n_cand, tid * chunk), e = min(n_cand, b + chunk);
    unsigned keep_bits = 0u;                              // (chunk <= 8)
    for (int t = b; t < e; ++t) {
        const double v = vals[t];
        keep_bits |= ((v >= pearson && v != 0.0) ? 1u : 0u) << (t - b);
    }
    int n = 0;
    int run = block_scan_counts(__builtin_popcount(keep_bits), part, n);
    for (int t = b; t < e; ++t)
        if ((keep_bits >> (t - b)) & 1u) {
            s_key[run] = (unsigned)keys_s[t];
            s_src[run] = (unsigned short)t;
            s_parent[run] = run;
            s_size[run] = 0;
            s_best[run] = INT_MAX;
            best_val[run] = 0ull;
            ++run;
        }
    if (tid == 0) *n_kept_out = n;
    __syncthreads();
    if (unsorted) lds_sort_pairs(s_key, s_src, n);        // (wave-uniform; the other arrays are position-indexed: untouched)
    // (2) 4-connected foci of the kept pixels: union-find, root = first pixel in row-major order
    for (int t = tid; t < n; t += kSmallThreads) {
        const unsigned key = s_key[t];
        const unsigned col = key % ns;
        if (t + 1 < n && col + 1 < ns && s_key[t + 1] == key + 1u) uf_union(s_parent, t, t + 1);
        const unsigned long long want = (unsigned long long)key + ns;       // lower neighbour: the list is sorted
        int lo = t + 1, hi = n;
        bool decided = false;
#pragma unroll
        for (int s = 1; s <= 4 && !decided; ++s) {
            if (t + s >= n) {
                decided = true;
                lo = n;
            } else if (s_key[t + s] >= want) {
                decided = true;
                lo = t + s;
            }
        }
        if (!decided) {
            lo = t + 5;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_key[mid] < want) lo = mid + 1;
                else hi = mid;
            }
        }
        if (lo < n && s_key[lo] == want) uf_union(s_parent, t, lo);
    }
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) s_parent[t] = uf_find(s_parent, t);
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) {
        const int r = s_parent[t];
        atomicAdd(s_size + r, 1);
        atomicMax(best_val + r, order_key(vals[s_src[t]]));
    }
    __syncthreads();
    for (int t = tid; t < n; t += kSmallThreads) {
        const int r = s_parent[t];
        if (order_key(vals[s_src[t]]) == best_val[r]) atomicMin(s_best + r, t);
    }
    __syncthreads();
    // (3) foci of at least min_size pixels, in the order of their first pixels: thread tid owns the kept pixels [b2, e2)
    const int chunk2 = (n + kSmallThreads - 1) / kSmallThreads;
    const int b2 = min(n, tid * chunk2), e2 = min(n, b2 + chunk2);
    int cnt = 0;
    for (int t = b2; t < e2; ++t) cnt += (s_parent[t] == t && s_size[t] >= min_size) ? 1 : 0;
    int n_out = 0;
    int at = block_scan_counts(cnt, part, n_out);
    for (int t = b2; t < e2; ++t)
        if (s_parent[t] == t && s_size[t] >= min_size) {
            const unsigned key = s_key[s_best[t]];
            int row = (int)(key / ns);
            const int col = (int)(key - (unsigned)row * ns);
            if (diag_only) row = col + (diag_only >> 1);   // odd code: 1-D pattern, row offset in the upper bits
            out_rows[at] = row;
            out_cols[at] = col;
            out_size[at] = s_size[t];
            ++at;
        }
    if (tid == 0) *n_foci = n_out;
}

__global__ __launch_bounds__(kSmallThreads) void foci_small_batch_kernel(
    const CorrArgs<double>* __restrict__ tab, const long long* __restrict__ seg, const long long* __restrict__ keys_s,
    const double* __restrict__ vals, double pearson, int min_size, int diag_only, int* __restrict__ flag, int* __restrict__ pos,
    long long* __restrict__ keys_k, double* __restrict__ vals_k, int* __restrict__ parent, int* __restrict__ size,
    unsigned long long* __restrict__ best_val, int* __restrict__ best_idx, int* __restrict__ out_rows, int* __restrict__ out_cols,
    int* __restrict__ out_size, int* __restrict__ n_kept, long long* __restrict__ n_foci_blk, int lds_ok, int unsorted = 0)
{
    __shared__ int part[kSmallThreads];
    extern __shared__ __attribute__((aligned(16))) char small_lds[];     // kSmallLdsBytes, or nothing (lds_ok = 0)
    const int b = blockIdx.x;
    const long long o = seg[b], n_cand = seg[b + 1] - o;
    if (lds_ok && n_cand <= kSmallLds && (unsigned long long)tab[b].ms * (unsigned long long)tab[b].ns <= 0xffffffffull) {
        foci_small_lds_body(keys_s + o, vals + o, (int)n_cand, pearson, (unsigned)tab[b].ns, min_size, diag_only, best_val + o,
                            out_rows + o, out_cols + o, out_size + o, n_kept + b, n_foci_blk + b, part, small_lds, unsorted);
        return;
    }
    if (unsorted) {                 // (the host sends unsorted lists only where every block takes the route above)
        if (threadIdx.x == 0) {
            n_kept[b] = 0;
            n_foci_blk[b] = 0;
        }
        return;
    }
    foci_small_body(keys_s + o, vals + o, n_cand, pearson, tab[b].ns, min_size, diag_only, flag + o, pos + o, keys_k + o, vals_k + o,
                    parent + o, size + o, best_val + o, best_idx + o, out_rows + o, out_cols + o, out_size + o, n_kept + b,
                    n_foci_blk + b, part);
}

// dynamic LDS of foci_small_batch_kernel (0: CHROMOSIGHT_HIP_NO_LDS_FOCI, every list through the global arrays)
static size_t foci_small_lds_bytes()
{
    static int state = 0;                                  // 1: the kernel may ask for that much, 2: it may not
    if (state == 0) {
        state = hipFuncSetAttribute((const void*)foci_small_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLdsBytes) == hipSuccess ? 1 : 2;
        (void)hipGetLastError();
    }
    return (state == 1 && !getenv("CHROMOSIGHT_HIP_NO_LDS_FOCI")) ? kSmallLdsBytes : 0;
}

// ---- foci of a 1-D pattern without the labelling workgroup --------------------------------------------------------------
// A scan of the diagonals 0 and 1 lists (i, i), (i, i + 1), (i + 1, i + 1), ...: every 4-neighbour pair among the listed
// pixels is a pair of CONSECUTIVE entries, so the 4-connected foci of the thresholded list (label_foci, detection.py:459-554)
// are its maximal runs -- no union-find, and nothing that needs one workgroup per sub-matrix (foci_small_batch_kernel: 1024
// threads walk 17 000 entries through a dozen barrier-separated phases, 160-180 us however many sub-matrices there are;
// the gather after it copies block by block in ONE workgroup: 40 us).  Here: one thread per entry flags the start of a run
// of at least min_size pixels and walks it (size, first maximum in row-major order = path order), a device-wide exclusive
// scan gives every focus its place in the list of ALL sub-matrices, one more pass writes the foci and the per-block counts.
__global__ __launch_bounds__(kThreads) void path_runs_kernel(const long long* __restrict__ seg, const int* __restrict__ blk,
                                                             const double* __restrict__ vals, long long n_total, double pearson,
                                                             int min_size, int* __restrict__ flag, int* __restrict__ size,
                                                             int* __restrict__ best_idx)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_total) return;
    const int b = blk[t];
    const long long s0 = seg[b], s1 = seg[b + 1];
    auto above = [&](long long u) {
        const double v = vals[u];
        return v >= pearson && v != 0.0;                       // detection.py:417-421
    };
    int f = 0;
    if (above(t) && (t == s0 || !above(t - 1))) {
        int cnt = 0;
        unsigned long long best = 0ull;
        long long best_u = t;
        for (long long u = t; u < s1 && above(u); ++u) {
            ++cnt;
            const unsigned long long k = order_key(vals[u]);
            if (cnt == 1 || k > best) {                        // the first maximum in row-major order (pick_foci, :438-453)
                best = k;
                best_u = u;
            }
        }
        if (cnt >= min_size) {
            f = 1;
            size[t] = cnt;
            best_idx[t] = (int)best_u;
        }
    }
    flag[t] = f;
}

__global__ __launch_bounds__(kThreads) void path_emit_kernel(const CorrArgs<double>* __restrict__ tab, const long long* __restrict__ seg,
                                                             int n_blocks, long long n_total, const int* __restrict__ blk,
                                                             const long long* __restrict__ keys, const int* __restrict__ flag,
                                                             const int* __restrict__ pos, const int* __restrict__ size,
                                                             const int* __restrict__ best_idx, int diag_only, int* __restrict__ f_rows,
                                                             int* __restrict__ f_cols, int* __restrict__ f_size, int* __restrict__ f_blk,
                                                             long long* __restrict__ f_off, long long* __restrict__ d_total,
                                                             long long* __restrict__ h_counts)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    auto before = [&](long long x) { return x < n_total ? (long long)pos[x] : (long long)pos[n_total - 1] + flag[n_total - 1]; };
    if (t <= n_blocks) {                                       // per-block counts and the total (also to page-locked host memory)
        const long long here = before(seg[t]);
        f_off[t] = here;
        if (t < n_blocks) {
            if (h_counts) h_counts[1 + t] = before(seg[t + 1]) - here;
        } else {
            *d_total = here;
            if (h_counts) h_counts[0] = here;
        }
    }
    if (t >= n_total || !flag[t]) return;
    const int b = blk[t], p = pos[t];
    const int ns = tab[b].ns;
    const long long key = keys[best_idx[t]];
    int row = (int)(key / ns);
    const int col = (int)(key - (long long)row * ns);
    if (diag_only) row = col + (diag_only >> 1);               // odd code: 1-D pattern, row offset in the upper bits
    f_rows[p] = row;
    f_cols[p] = col;
    f_size[p] = size[t];
    f_blk[p] = b;
}

// the foci of all sub-matrices, block after block: offsets from the per-block counts (one workgroup),
// the per-block counts and the total also go to (page-locked) host memory
__global__ __launch_bounds__(kSmallThreads) void gather_foci_batch_kernel(const long long* __restrict__ seg,
                                                                         const long long* __restrict__ n_foci_blk, int n_blocks,
                                                                         const int* __restrict__ seg_rows, const int* __restrict__ seg_cols,
                                                                         const int* __restrict__ seg_size, int* __restrict__ f_rows,
                                                                         int* __restrict__ f_cols, int* __restrict__ f_size,
                                                                         int* __restrict__ f_blk, long long* __restrict__ f_off,
                                                                         long long* __restrict__ d_total, long long* __restrict__ h_counts)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        long long acc = 0;
        for (int b = 0; b < n_blocks; ++b) {
            f_off[b] = acc;
            acc += n_foci_blk[b];
            if (h_counts) h_counts[1 + b] = n_foci_blk[b];
        }
        f_off[n_blocks] = acc;
        *d_total = acc;
        if (h_counts) h_counts[0] = acc;
    }
    __syncthreads();
    for (int b = 0; b < n_blocks; ++b) {
        const long long src = seg[b], dst = f_off[b], cnt = n_foci_blk[b];
        for (long long t = tid; t < cnt; t += kSmallThreads) {
            f_rows[dst + t] = seg_rows[src + t];
            f_cols[dst + t] = seg_cols[src + t];
            f_size[dst + t] = seg_size[src + t];
            f_blk[dst + t] = b;
        }
    }
}


// ---- window statistics of validate_patterns, one wave per pattern --------------------------------
// The map pattern_detector validates on (detection.py:287-310) is never built: the contact map framed
// by (kw rows, kh columns) of zeros when full, NaN on the max(km, kn) first sub-diagonals of intra
// maps, NaN on every row / column that is not a detectable bin; coordinates shifted by (kh, kw).
// window statistics of one pattern by one wave (validate_patterns, detection.py:18-155); rec_out / win_out
// may be nullptr (beyond the caller's capacity) or page-locked host memory
// The p-value of a record (reference detection.py:332-336 on the untrimmed map, stats.py:43-81 corr_to_pval: Fisher z, two-sided
// normal tail with the case split of scipy.special.ndtr) -- the arithmetic of cs_accept_records (cs_api_entries.cpp two_sided_tail), on
// the lane that writes the record: three transcendental functions per record were most of what the host spent on a record.
__device__ __forceinline__ double focus_pval(double score, double nobs, bool full, double tot)
{
    double n_obs = full ? nobs : tot;
    if (n_obs == 0) n_obs = tot;
    if (score == 0) return 1.0;                                   // 10 ** 0 where the coefficient is exactly 0
    const double a = fabs(atanh(score) * sqrt(n_obs - 3.0));
    const double x = -a * 0.70710678118654752440, z = fabs(x);
    double y;
    if (z < 0.70710678118654752440) y = 0.5 + 0.5 * erf(x);
    else {
        y = 0.5 * erfc(z);
        if (x > 0) y = 1.0 - y;
    }
    return 2.0 * y;
}

__device__ __forceinline__ void window_stats_pattern(const CorrArgs<double>& A, int inter, int row, int col, int fsize, double score,
                                                     double nobs, FocusRec* rec_out, double* win_out, int lane,
                                                     double* lazy_win = nullptr, bool fast_windows = false,
                                                     int have_p0 = INT_MIN, int have_q0 = INT_MIN)
{
    const int km = A.km, kn = A.kn, kk = km * kn;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
    const int half_h = km / 2 + 1, half_w = kn / 2 + 1;
    const int pad_r = A.full ? kw : 0, pad_c = A.full ? kh : 0;     // zero_pad_sparse(mat, kh, kw)
    const int sh_r = A.full ? kh : 0, sh_c = A.full ? kw : 0;       // coords += (kh, kw)
    const int H = A.ms + 2 * pad_r, W = A.ns + 2 * pad_c;
    const int big_k = inter ? 0 : max(km, kn);
    const int p1 = row + sh_r, p2 = col + sh_c;
    const int high = p1 - half_h + 1, low = p1 + half_h;
    const int left = p2 - half_w + 1, right = p2 + half_w;
    const bool inside = (high >= 0) & (low < H) & (left >= 0) & (right < W);   // strict upper bounds (:99-104)
    int n_zero = 0, n_miss = 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // a lazily evaluated band: the window's pixels gathered by the wave first (cs_launch_aux.h lazy_gather_window)
    const bool gathered = lazy_win && inside && A.sig.layout == 2 && lazy_window_fits(km, kn);          // wave-uniform
    // (have_p0, have_q0: lazy_win already holds the window with this top left pixel -- the exact evaluation of the same record
    // gathered it a moment ago, focus_records_lazy17_kernel)
    const bool have = gathered && have_p0 == high - pad_r && have_q0 == left - pad_c;
    if (gathered && !have) {
        if (fast_windows && km == 17 && kn == 17) lazy_gather_window_sq<17>(A, high - pad_r, left - pad_c, lane, lazy_win);
        else lazy_gather_window(A, high - pad_r, left - pad_c, lane, lazy_win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (int e = lane; e < kk; e += 64) {
        double v = nan;
        if (inside) {
            const int a = e / kn, b = e - a * kn;
            const int rr = high + a, cc = left + b;
            const int src_r = rr - pad_r, src_c = cc - pad_c;
            v = gathered ? lazy_win[e] : load_signal(A, src_r, src_c);       // 0 outside the matrix / stored band
            const int d = cc - rr;
            bool miss = (d <= -1) & (d >= -big_k);
            // framed row rr is detectable iff rr - sh_r is a detectable bin
            const int br = rr - sh_r, bc = cc - sh_c;
            miss |= (br < 0) | (br >= A.ms) | (bc < 0) | (bc >= A.ns);
            if (!miss && A.miss_row) miss = (A.miss_row[br] != 0) | (A.miss_col[bc] != 0);
            if (miss) v = nan;
            const bool fin = (v - v) == 0.0;                     // finite
            n_zero += (fin && v == 0.0) ? 1 : 0;
            n_miss += fin ? 0 : 1;
        }
        if (win_out) win_out[e] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_zero += __shfl_xor(n_zero, off, 64);
        n_miss += __shfl_xor(n_miss, off, 64);
    }
    if (lane == 0) {
        FocusRec r;
        r.bin1 = row;
        r.bin2 = col;
        r.inside = inside ? 1 : 0;
        r.n_zero = n_zero;
        r.n_missing = n_miss;
        r.focus_size = fsize;
        r.score = score;
        r.n_obs = nobs;
        r.pval = focus_pval(score, nobs, A.full != 0, (double)kk);
        if (rec_out) *rec_out = r;
    }
}

__global__ __launch_bounds__(kThreads) void window_stats_kernel(const CorrArgs<double> A, int inter,
                                                                const int* __restrict__ rows, const int* __restrict__ cols,
                                                                const int* __restrict__ focus_size,
                                                                const double* __restrict__ score,
                                                                const double* __restrict__ nobs,
                                                                const long long* __restrict__ n_ptr, long long n_fixed,
                                                                FocusRec* __restrict__ rec, double* __restrict__ windows,
                                                                long long win_cap, long long rec_cap, long long* __restrict__ n_out)
{
    // rec / windows / n_out may be page-locked HOST memory (results written over the link, no copy call):
    // nothing beyond the caller's capacities is touched, and the count is published even when it is 0
    const long long n = n_ptr ? *n_ptr : n_fixed;
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (kThreads >> 6) + (threadIdx.x >> 6);
    if (t == 0 && lane == 0 && n_out) *n_out = n;
    if (t >= n) return;
    const int kk = A.km * A.kn;
    window_stats_pattern(A, inter, rows[t], cols[t], focus_size ? focus_size[t] : 0, score[t], nobs[t],
                         t < rec_cap ? rec + t : nullptr, (windows && t < win_cap) ? windows + t * kk : nullptr, lane);
}

__global__ __launch_bounds__(kThreads) void window_stats_batch_kernel(const CorrArgs<double>* __restrict__ tab, int inter,
                                                                      const int* __restrict__ blk, const int* __restrict__ rows,
                                                                      const int* __restrict__ cols, const int* __restrict__ focus_size,
                                                                      const double* __restrict__ score, const double* __restrict__ nobs,
                                                                      const long long* __restrict__ n_ptr, FocusRec* __restrict__ rec,
                                                                      double* __restrict__ windows, long long win_cap, long long rec_cap,
                                                                      bool fast_windows)
{
    const long long n = *n_ptr;
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (kThreads >> 6) + (threadIdx.x >> 6);
    if (t >= n) return;
    __shared__ double lazy_win[kThreads >> 6][kLazyWinMax];
    const CorrArgs<double>& A = tab[blk[t]];
    const int kk = A.km * A.kn;
    window_stats_pattern(A, inter, rows[t], cols[t], focus_size[t], score[t], nobs[t], t < rec_cap ? rec + t : nullptr,
                         (windows && t < win_cap) ? windows + t * kk : nullptr, lane, lazy_win[threadIdx.x >> 6], fast_windows);
}

// The records of the foci in ONE pass (detect mode, every sub-matrix in the detection configuration with a 17 x 17 template on a
// lazily evaluated band): exact coefficient AND window statistics from one gathered window -- they read the same 17 x 17
// pixels -- instead of rescore_batch_kernel followed by window_stats_batch_kernel, each with its own gather.  Grid-stride over
// the foci: their number is known on the device only, and the two kernels used to be launched with one wave per POSSIBLE
// focus (candidates / min_size: 600 000 waves for the 1-D patterns of a genome, 56 000 of them with work -- the empty
// workgroups alone cost 2 x 140 us at the end of a genome step).
__global__ __launch_bounds__(256) void focus_records_lazy17_kernel(
    const CorrArgs<double>* __restrict__ tab, int inter, const int* __restrict__ blk, const int* __restrict__ rows, const int* __restrict__ cols,
    const int* __restrict__ focus_size, const long long* __restrict__ n_ptr, FocusRec* __restrict__ rec, double* __restrict__ windows,
    long long win_cap, long long rec_cap)
{
    __shared__ double lazy_win[4][kLazyWinMax];
    const long long n = *n_ptr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long stride = (long long)gridDim.x * 4;
    for (long long t = (long long)blockIdx.x * 4 + wv; t < n; t += stride) {
        const CorrArgs<double>& A = tab[blk[t]];
        const int oi = rows[t], oj = cols[t];
        double r = 0.0, nobs = A.ks.n;
        double* win = lazy_win[wv];
        const bool inside = (oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns);                  // wave-uniform
        if (inside) {
            lazy_gather_window_sq<17>(A, oi - 8, oj - 8, lane, win);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            rescore_pixel_sq<17>(A, oi, oj, lane, r, nobs, win);
        }
        window_stats_pattern(A, inter, oi, oj, focus_size[t], r, nobs, t < rec_cap ? rec + t : nullptr,
                             (windows && t < win_cap) ? windows + t * (17 * 17) : nullptr, lane, win, true, inside ? oi - 8 : INT_MIN, inside ? oj - 8 : INT_MIN);
        // (the next record's gather overwrites the window: every lane is done reading it)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// exact coefficients + records of the foci of a batch (f_* arrays, *d_total of them, at most max_foci)
static void launch_focus_records(const CorrArgs<double>* h_tab, int n_blocks, long long max_foci, hipStream_t stream,
                                 const CorrArgs<double>* tab, int inter, const int* f_blk, const int* f_rows, const int* f_cols,
                                 const int* f_size, double* f_score, double* f_nobs, const long long* d_total, FocusRec* rec,
                                 double* windows, long long win_cap, long long rec_cap)
{
    bool lazy17 = fast_windows_on() && h_tab != nullptr && n_blocks > 0;
    for (int b = 0; lazy17 && b < n_blocks; ++b) {
        const CorrArgs<double>& A = h_tab[b];
        lazy17 = A.km == 17 && A.kn == 17 && A.mask_mode == 1 && A.sym_upper && A.full && A.max_dist >= 0 && A.sig.layout == 2;
    }
    if (lazy17) {
        const unsigned grid = (unsigned)std::min<long long>((max_foci + 3) / 4, 4096);
        hipLaunchKernelGGL(focus_records_lazy17_kernel, dim3(grid), dim3(256), 0, stream, tab, inter, f_blk, f_rows, f_cols, f_size, d_total, rec,
                           windows, win_cap, rec_cap);
        return;
    }
    launch_rescore_batch(h_tab, n_blocks, max_foci, stream, tab, f_blk, f_rows, f_cols, max_foci, d_total, f_score, f_nobs);
    hipLaunchKernelGGL(window_stats_batch_kernel, dim3((unsigned)((max_foci + 3) / 4)), dim3(kThreads), 0, stream, tab, inter, f_blk, f_rows,
                       f_cols, f_size, f_score, f_nobs, d_total, rec, windows, win_cap, rec_cap, fast_windows_on());
}

// quantify mode over several sub-matrices (cs_quantify_blocks): entry t is pixel (rows[t], cols[t]) of sub-matrix blk[t];
// inter-chromosomal sub-matrices (no NaN sub-diagonals in the windows) are flagged per sub-matrix
__global__ __launch_bounds__(kThreads) void quantify_stats_batch_kernel(const CorrArgs<double>* __restrict__ tab, const int* __restrict__ blk_inter,
                                                                        const int* __restrict__ blk, const int* __restrict__ rows,
                                                                        const int* __restrict__ cols, const double* __restrict__ score,
                                                                        const double* __restrict__ nobs, long long n,
                                                                        FocusRec* __restrict__ rec, double* __restrict__ windows)
{
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * (kThreads >> 6) + (threadIdx.x >> 6);
    if (t >= n) return;
    const int b = blk[t];
    const CorrArgs<double>& A = tab[b];
    const int kk = A.km * A.kn;
    window_stats_pattern(A, blk_inter[b], rows[t], cols[t], 0, score[t], nobs[t], rec + t, windows ? windows + t * kk : nullptr, lane);
}

}  // namespace
