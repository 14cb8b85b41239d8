// cs_launch_aux.h -- launchers of the helper kernels in cs_aux.hip (internal).
#pragma once
#include <vector>
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
    double score, n_obs, pval;
};

#ifdef __HIPCC__
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The km x kn window with the top left pixel (p0, q0) of a lazily evaluated band (cs_device.h LazyBand), gathered by ONE WAVE into `win`
// (km * kn doubles of LDS): what load_signal returns for every pixel of the window, without a binary search per pixel.
// The stored diagonals come from the near band; beyond them lane ki finds the first entry of row ki of the window in the
// pixel table (one search per ROW, all rows at once), and the at most kn entries behind it are read side by side,
// detrended and dropped into their window slots.
constexpr int kLazyWinMax = 17 * 17;

// Templates lazy_gather_window serves: the window fits the wave's LDS slots AND has at most one row per lane (lane ki searches
// row ki of the window; a 70 x 4 template passes the area test and would leave its rows 64.. unread beyond the near band).
// Everything else reads through load_signal pixel by pixel.
__device__ __forceinline__ bool lazy_window_fits(int km, int kn) { return km * kn <= kLazyWinMax && km <= 64 && kn <= 64; }

__device__ __forceinline__ void lazy_gather_window(const CorrArgs<double>& A, int p0, int q0, int lane, double* win)
{
    const LazyBand& L = *reinterpret_cast<const LazyBand*>(A.sig.ptr);
    const int km = A.km, kn = A.kn, kk = km * kn;
    const int d_lo = max(A.sig.band_lo, 0), d_end = min(A.sig.band_lo + A.sig.band_w, L.n_diags);      // kept diagonals
    const int near_end = min(L.near_w, d_end);
    for (int e = lane; e < kk; e += 64) {
        const int ki = e / kn, kj = e - ki * kn;
        const int p = p0 + ki, q = q0 + kj, d = q - p;
        double v = 0.0;
        if ((p >= 0) & (p < A.ms) & (q >= 0) & (q < A.ns) & (d >= d_lo) & (d < near_end)) v = lazy_near_value(L, p, d);
        win[e] = v;
    }
    // (the LDS operations of a wave execute in order: the slots are zero before anything is dropped into them)
    long long pos = 0, end = 0;
    int c_hi = -1;
    if (lane < km) {
        const int p = p0 + lane;
        if (p >= 0 && p < A.ms) {
            const int c_lo = max(q0, p + max(near_end, d_lo));
            const int hi = min(min(q0 + kn - 1, p + d_end - 1), A.ns - 1);
            if (c_lo <= hi) {
                const long long r = L.row0 + p;
                long long lo = L.indptr[r];
                end = L.indptr[r + 1];
                long long up = end;
                const int target = (int)(L.row0 + c_lo);
                while (lo < up) {
                    const long long mid = (lo + up) >> 1;
                    if (L.indices[mid] < target) lo = mid + 1;
                    else up = mid;
                }
                pos = lo;
                c_hi = hi;
            }
        }
    }
    if (!__builtin_amdgcn_ballot_w64(c_hi >= 0)) return;
    for (int t0 = 0; t0 < kk; t0 += 64) {
        const int t = min(t0 + lane, kk - 1);
        const int ki = t / kn, s = t - ki * kn;
        const long long ps = __shfl(pos, ki), en = __shfl(end, ki);
        const int ch = __shfl(c_hi, ki);
        const long long idx = ps + s;
        if (t0 + lane < kk && ch >= 0 && idx < en) {
            const int c = L.indices[idx] - (int)L.row0;
            if (c <= ch) {
                const int p = p0 + ki;
                win[ki * kn + (c - q0)] = lazy_stored_value(L, idx, p, c);
            }
        }
    }
}

// float64 coefficient (and present-pixel count) of one pixel by one wave: _normxcorr2_sparse per pixel
// (detection.py:917-1131).  `A` may live in registers (kernel argument) or in global memory (a table of
// per-sub-matrix arguments: the wave's loads are uniform).
// `win` (optional): the window's km x kn pixels already gathered by the wave (lazy_gather_window) -- the same values
// load_signal returns, in the same places of the same sums.
__device__ __forceinline__ void rescore_pixel(const CorrArgs<double>& A, int oi, int oj, int lane, double& r, double& nobs,
                                              const double* win = nullptr)
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
            const double v = win ? win[e] : load_signal(A, p, q);
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

// ---- the wave-per-window evaluation for the configuration nearly every window of a detect run has --------------------
// rescore_pixel / lazy_gather_window compile to ~1 800 instructions per window for a 17 x 17 template (runtime sizes: an integer
// division per pixel; a branch, two flag loads and three weight loads inside it per pixel for the predicate; 64-bit lane
// exchanges), and the float64 re-scoring of a genome's ~10^5 candidate windows is bound by exactly that -- instruction issue at
// full occupancy, not latency, not bytes (313 us exposed at the end of a 23-block genome step, profiles/r04b_genome_timeline.txt).
// The same operations on the same operands in the same order -- per-lane sums over e = lane, lane + 64, ..., then the
// butterfly -- for per-bin masks, sym_upper, full, max_dist given and a SQUARE template of compile-time size K:
//   * the flags of the window's K rows and K columns are loaded once (lane k: row k, lane K + k: column k) and become two
//     bit words by a ballot; the predicate is missing_from_flags written as selects (x + 0.0 == x: the mask sums are
//     bit-identical to the branching form's);
//   * the three mask sums are reduced only when some lane saw a missing pixel (else they are exactly 0);
//   * the lazily evaluated band's gather keeps its positions relative to the window's first row (32-bit exchanges).
template <int K>
__device__ __forceinline__ bool window_fast_applies(const CorrArgs<double>& A)
{
    return (A.km == K) & (A.kn == K) & (A.mask_mode == 1) & (A.sym_upper != 0) & (A.full != 0) & (A.max_dist >= 0);
}

// a wave-uniform pointer as a scalar pointer to GLOBAL memory: pointers read from a descriptor in memory are generic
// (flat_load: every wait also waits for the LDS) and, kept in vector registers, cost two lanes-wide additions per address
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* uniform_global(const T* p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const __attribute__((address_space(1))) T*>(((unsigned long long)hi << 32) | lo);
}

// load_signal for a wave-uniform argument block with every field read ONCE into scalars: `A` lives in a device table, a lazily
// evaluated band behind one more pointer, and a loop  tile[i] = load_signal(A, p, q)  re-reads those fields behind every LDS
// store (the compiler must assume they changed) -- three dependent round trips per element, which is what the tile staging
// of rescore_run_batch_kernel spent its time on (~ 100 of its 144 us on a rank's share of a genome).  Same values as
// load_signal; the pixels of a lazily evaluated band beyond its stored diagonals still go through lazy_load.
struct SigReader {
    const __attribute__((address_space(1))) double* p64;
    const __attribute__((address_space(1))) float* p32;
    const MatView* view;
    long long ld, row0;
    int ms, ns, layout, band_lo, band_w, is_f64, near_w, n_diags;
    // a lazily evaluated band whose stored diagonals are a band of counts (cs_device.h LazyBand::counts)
    const __attribute__((address_space(1))) float* cnt;
    const __attribute__((address_space(1))) double* cnt_weight;
    const __attribute__((address_space(1))) double* cnt_law;
    long long cnt_row0;
    double cnt_max;

    __device__ __forceinline__ explicit SigReader(const CorrArgs<double>& A)
    {
        view = &A.sig;
        ms = __builtin_amdgcn_readfirstlane(A.ms);
        ns = __builtin_amdgcn_readfirstlane(A.ns);
        layout = __builtin_amdgcn_readfirstlane(A.sig.layout);
        band_lo = __builtin_amdgcn_readfirstlane(A.sig.band_lo);
        band_w = __builtin_amdgcn_readfirstlane(A.sig.band_w);
        is_f64 = __builtin_amdgcn_readfirstlane(A.sig_is_f64);
        near_w = n_diags = 0;
        cnt = nullptr;
        cnt_weight = cnt_law = nullptr;
        cnt_row0 = 0;
        cnt_max = 0.0;
        const void* ptr = A.sig.ptr;
        ld = A.sig.ld;
        row0 = A.sig.row0;
        if (layout == 2) {
            const LazyBand* L = reinterpret_cast<const LazyBand*>(ptr);
            cnt = uniform_global(L->counts);
            cnt_weight = uniform_global(L->weight);
            cnt_law = uniform_global(L->law);
            cnt_row0 = __builtin_amdgcn_readfirstlane((int)L->row0);
            cnt_max = L->max_val;
            ptr = L->near_;
            ld = L->counts ? L->counts_ld : L->near_ld;
            row0 = 0;
            near_w = __builtin_amdgcn_readfirstlane(L->near_w);
            n_diags = __builtin_amdgcn_readfirstlane(L->n_diags);
            is_f64 = 1;
        }
        ld = ((long long)__builtin_amdgcn_readfirstlane((int)(ld >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ld);
        row0 = ((long long)__builtin_amdgcn_readfirstlane((int)(row0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)row0);
        p64 = uniform_global(reinterpret_cast<const double*>(ptr));
        p32 = uniform_global(reinterpret_cast<const float*>(ptr));
    }

    __device__ __forceinline__ double at(int p, int q) const
    {
        if ((p < 0) | (p >= ms) | (q < 0) | (q >= ns)) return 0.0;
        const int d = q - p;
        if (layout == 2) {
            if ((d - band_lo < 0) | (d - band_lo >= band_w) | (d < 0) | (d >= n_diags)) return 0.0;
            if (d < near_w) {
                if (cnt) return counts_value_flat((const float*)cnt + ((long long)p * ld + d), (const double*)cnt_weight, (const double*)cnt_law, cnt_row0, p, d, cnt_max, n_diags);
                return p64[(long long)p * ld + d];
            }
            return lazy_load(*view, p, q);
        }
        long long off;
        if (layout == 0) off = ((long long)p - row0) * ld + q;
        else {
            const int dd = d - band_lo;
            if ((dd < 0) | (dd >= band_w)) return 0.0;
            off = ((long long)p - row0) * ld + dd;
        }
        return is_f64 ? p64[off] : (double)p32[off];
    }
};

template <int K>
__device__ __forceinline__ void lazy_gather_window_sq(const CorrArgs<double>& A, int p0, int q0, int lane, double* win)
{
    // every field of the descriptor and of the arguments is read ONCE, before the first LDS store: behind a store to `win` the
    // compiler must assume the descriptor changed and re-read it (the general function pays six dependent round trips per
    // 64 window slots for that)
    const LazyBand* Lp = reinterpret_cast<const LazyBand*>(A.sig.ptr);
    const auto* indptr = uniform_global(Lp->indptr);
    const auto* indices = uniform_global(Lp->indices);
    const auto* data32 = uniform_global(reinterpret_cast<const float*>(Lp->data));
    const auto* data64 = uniform_global(reinterpret_cast<const double*>(Lp->data));
    const auto* weight = uniform_global(Lp->weight);
    const auto* law = uniform_global(Lp->law);
    const auto* near_ = uniform_global(Lp->near_);
    const auto* counts = uniform_global(Lp->counts);
    const long long row0 = __builtin_amdgcn_readfirstlane((int)Lp->row0);         // (genome bins: < 2^31)
    const long long near_ld = __builtin_amdgcn_readfirstlane((int)(Lp->counts ? Lp->counts_ld : Lp->near_ld));
    const double max_val = Lp->max_val;
    const int n_diags = __builtin_amdgcn_readfirstlane(Lp->n_diags), near_w = __builtin_amdgcn_readfirstlane(Lp->near_w);
    const bool is_f64 = __builtin_amdgcn_readfirstlane(Lp->data_is_f64) != 0;
    const int ms = __builtin_amdgcn_readfirstlane(A.ms), ns = __builtin_amdgcn_readfirstlane(A.ns);
    const int band_lo = __builtin_amdgcn_readfirstlane(A.sig.band_lo), band_w = __builtin_amdgcn_readfirstlane(A.sig.band_w);
    constexpr int kk = K * K;
    const int d_lo = max(band_lo, 0), d_end = min(band_lo + band_w, n_diags);      // kept diagonals
    const int near_end = min(near_w, d_end);
    // (wave-uniform: does the window reach the stored diagonals at all?  its smallest diagonal is q0 - p0 - (K - 1))
    const bool near_hit = q0 - p0 - (K - 1) < near_end;
    // the rows' first entries at or behind their first wanted column: one search per ROW, all rows at once, positions
    // relative to the window's first row (32-bit lane exchanges below)
    const long long base = indptr[row0 + min(max(p0, 0), ms - 1)];                // uniform; every position below is >= base
    int pos = 0, end = 0, c_hi = -1;
    if (lane < K) {
        const int p = p0 + lane;
        if (p >= 0 && p < ms) {
            const int c_lo = max(q0, p + max(near_end, d_lo));
            const int hi = min(min(q0 + K - 1, p + d_end - 1), ns - 1);
            if (c_lo <= hi) {
                const long long r = row0 + p;
                int lo = (int)(indptr[r] - base);
                end = (int)(indptr[r + 1] - base);
                int up = end;
                const int target = (int)row0 + c_lo;
                const auto* ind = indices + base;
                while (lo < up) {
                    const int mid = (lo + up) >> 1;
                    if (ind[mid] < target) lo = mid + 1;
                    else up = mid;
                }
                pos = lo;
                c_hi = hi;
            }
        }
    }
    const bool any_row = __builtin_amdgcn_ballot_w64(c_hi >= 0) != 0;
    constexpr int kRounds = (kk + 63) / 64;
#pragma unroll
    for (int e = lane, i = 0; i < kRounds; e += 64, ++i) {
        if (e < kk) {
            double v = 0.0;
            if (near_hit) {
                const int ki = e / K, kj = e - ki * K;
                const int p = p0 + ki, q = q0 + kj, d = q - p;
                if ((p >= 0) & (p < ms) & (q >= 0) & (q < ns) & (d >= d_lo) & (d < near_end)) {
                    if (counts) v = counts_value_flat((const float*)counts + ((long long)p * near_ld + d), (const double*)weight, (const double*)law, row0, p, d, max_val, n_diags);
                    else v = near_[(long long)p * near_ld + d];
                }
            }
            win[e] = v;
        }
    }
    if (!any_row) return;
    // (the LDS operations of a wave execute in order: the slots are zero before anything is dropped into them; the global
    // loads below are through address-space-1 pointers held in scalars, so nothing is re-read behind the LDS stores)
#pragma unroll
    for (int i = 0; i < kRounds; ++i) {
        const int t = min(64 * i + lane, kk - 1);
        const int ki = t / K, s = t - ki * K;
        const int ps = __shfl(pos, ki), en = __shfl(end, ki), ch = __shfl(c_hi, ki);
        const int rel = ps + s;
        if (64 * i + lane < kk && ch >= 0 && rel < en) {
            const long long idx = base + rel;
            const int c = indices[idx] - (int)row0;
            if (c <= ch) {
                // lazy_stored_value / stage_detrend_rcp (cs_device.h), on the hoisted fields
                const int p = p0 + ki, d = c - p;
                const double x = is_f64 ? data64[idx] : (double)data32[idx];
                const double y = d < n_diags ? law[d] : 0.0;
                win[ki * K + (c - q0)] = stage_detrend_rcp((x * weight[row0 + p]) * weight[row0 + c], 1.0 / y, max_val, (const double*)(law + min(d, n_diags - 1)));
            }
        }
    }
}

// rescore_pixel for window_fast_applies<K>(A) and (oi, oj) inside the matrix; `win`: the gathered window (lazily evaluated band) or
// nullptr (stored band / dense map: load_signal)
template <int K>
__device__ __forceinline__ void rescore_pixel_sq(const CorrArgs<double>& A, int oi, int oj, int lane, double& r, double& nobs,
                                                 const double* win)
{
    r = 0.0;
    nobs = A.ks.n;
    if (oj - oi < 0) return;                                   // pixel_forced_zero: sym_upper, km == kn
    constexpr int kh = (K - 1) / 2, kk = K * K;
    const int p0 = oi - kh, q0 = oj - kh;
    const int ms = __builtin_amdgcn_readfirstlane(A.ms), ns = __builtin_amdgcn_readfirstlane(A.ns);
    const int md = __builtin_amdgcn_readfirstlane(A.max_dist);
    // flags of the window's rows and columns, one load per lane
    unsigned rbits, cbits;
    {
        const auto* miss_row = uniform_global(A.miss_row);
        const auto* miss_col = uniform_global(A.miss_col);
        const bool is_row = lane < K;
        const int idx = is_row ? p0 + lane : q0 + (lane - K);
        const bool in = is_row ? (idx >= 0) & (idx < ms) : (lane < 2 * K) & (idx >= 0) & (idx < ns);
        const unsigned char fr = miss_row[(in & is_row) ? idx : 0], fc = miss_col[(in & !is_row) ? idx : 0];
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(in & ((is_row ? fr : fc) != 0));
        rbits = (unsigned)bal & ((1u << K) - 1u);
        cbits = (unsigned)(bal >> K) & ((1u << K) - 1u);
    }
    const auto* w = uniform_global(A.w);
    double cs_ = 0, s1 = 0, s2 = 0, nm = 0, ka = 0, kb = 0;
    bool any_missing = false;
#pragma unroll
    for (int e = lane; e < kk; e += 64) {
        const int ki = e / K, kj = e - ki * K;
        const int p = p0 + ki, q = q0 + kj, d = q - p;
        const double v = win ? win[e] : load_signal(A, p, q);
        cs_ = fma(v, w[e], cs_);
        s1 += v;
        s2 = fma(v, v, s2);
        // missing_from_flags for per-bin masks, sym_upper, full, max_dist >= 0, km == kn
        const bool in = (p >= 0) & (p < ms) & (q >= 0) & (q < ns);
        const bool flagged = (((rbits >> ki) | (cbits >> kj)) & 1u) != 0;
        const bool m_in = flagged & (d >= 0) & (d <= md);
        const bool m_out = (q >= ns) ? (p >= ms - md - 2) : ((p < 0) & ((q < 0) | (q < md + K)));
        const bool m = (in ? m_in : m_out) | ((d <= -1) & (d >= -K));
        if (__builtin_amdgcn_ballot_w64(m)) {                  // (uniform; half of a genome's windows have no missing pixel at all)
            any_missing = true;
            nm += m ? 1.0 : 0.0;
            ka += m ? w[kk + e] : 0.0;
            kb += m ? w[2 * kk + e] : 0.0;
        }
    }
    cs_ = wave_sum(cs_);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (any_missing) {
        nm = wave_sum(nm);
        ka = wave_sum(ka);
        kb = wave_sum(kb);
    }
    r = pearson_from_sums<double>(cs_, s1, s2, nm, ka, kb, A.ks, true, &nobs);
}

// per-bin flags of the rows p0 .. p0 + km - 1 and the columns q0 .. q0 + kn - 1 of a window as bit masks (bins outside
// the matrix: 0); for templates of up to 32 x 32 -- larger ones evaluate missing_pred per pixel
__device__ __forceinline__ void window_flag_bits(const CorrArgs<double>& A, int p0, int q0, unsigned& rbits, unsigned& cbits)
{
    rbits = cbits = 0u;
    if (A.mask_mode != 1 || A.km > 32 || A.kn > 32) return;
    for (int k = 0; k < A.km; ++k) {
        const int p = p0 + k;
        if (p >= 0 && p < A.ms && A.miss_row[p]) rbits |= 1u << k;
    }
    for (int k = 0; k < A.kn; ++k) {
        const int q = q0 + k;
        if (q >= 0 && q < A.ns && A.miss_col[q]) cbits |= 1u << k;
    }
}

// The same coefficient by ONE LANE (it walks the whole window): for lists that are runs of neighbouring pixels --
// the 2-4 scanned diagonals of a 1-D pattern, hundreds of thousands of pixels whose windows overlap in 16 of 17
// rows, so the loads hit L1 -- a wave then scores 64 pixels instead of one (no reductions, 1/9 of the instructions).
__device__ __forceinline__ void rescore_pixel_lane(const CorrArgs<double>& A, int oi, int oj, double& r, double& nobs)
{
    r = 0.0;
    nobs = A.ks.n;
    const bool inside = (oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns);
    if (inside && !pixel_forced_zero(A, oi, oj)) {
        const int km = A.km, kn = A.kn, kk = km * kn;
        const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
        const bool masked = A.mask_mode != 0;
        double cs_ = 0, s1 = 0, s2 = 0, nm = 0, ka = 0, kb = 0;
        const bool bins = A.mask_mode == 1 && km <= 32 && kn <= 32;
        unsigned rbits = 0, cbits = 0;
        window_flag_bits(A, oi - kh, oj - kw, rbits, cbits);
        int e = 0;
        for (int ki = 0; ki < km; ++ki) {
            const int p = oi - kh + ki;
            for (int kj = 0; kj < kn; ++kj, ++e) {
                const int q = oj - kw + kj;
                const double v = load_signal(A, p, q);
                cs_ = fma(v, A.w[e], cs_);
                s1 += v;
                s2 = fma(v, v, s2);
                if (masked && (bins ? missing_from_flags(A, p, q, (rbits >> ki) & 1u, (cbits >> kj) & 1u, false, true)
                                    : missing_pred(A, p, q))) {
                    nm += 1.0;
                    ka += A.w[kk + e];
                    kb += A.w[2 * kk + e];
                }
            }
        }
        r = pearson_from_sums<double>(cs_, s1, s2, nm, ka, kb, A.ks, masked, &nobs);
    }
}

// rescore_pixel_lane with everything the window loop touches staged in LDS by the workgroup:
//   tile[(p - P0) * DN + (q - p - D0)] = load_signal(p, q) for P0 <= p < P0 + RN, D0 <= q - p < D0 + DN,
//   wl = the three weight sets (copy of A.w), rfl[p - P0] / cfl[q - C0] = per-bin flags (0 outside the matrix).
// A global load per window pixel, each awaited before the next, is what made the direct version slow (and the flags
// alone cost 34 such round trips per pixel).  Same operations in the same order as rescore_pixel_lane: identical
// results.  The predicate is missing_from_flags written without branches for the detection configuration
// (per-bin masks, sym_upper, full, max_dist given); other configurations call the general function.
// The same for the configuration nearly every call has -- per-bin masks, sym_upper, full, max_dist given, a SQUARE
// template of compile-time size K, and every window of the workgroup inside the matrix -- without a branch: the
// predicate collapses to  d < 0 ? d >= -K : (row flag | column flag) & d <= max_dist  and the mask sums take 0.0 where
// it is false (x + 0.0 == x: the sums are bit-identical to the branching form's).  The general function compiles to
// ~215 instructions per window pixel (135 of them scalar branch bookkeeping around `if (missing)`): 363 us for the 400 000
// pixels of a borders template on the 23-block genome; this one to ~25.
// EDGE: windows that leave the matrix (the first and last runs of every sub-matrix) -- the frame rules of
// missing_from_flags as selects; still no branch.
template <int K, bool EDGE = false>
__device__ __forceinline__ void rescore_pixel_lane_lds_interior(const CorrArgs<double>& A, const double* tile, const double* wl,
                                                                const unsigned char* rfl, const unsigned char* cfl, int P0, int C0,
                                                                int D0, int DN, int oi, int oj, double& r, double& nobs)
{
    r = 0.0;
    nobs = A.ks.n;
    if (EDGE && !((oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns))) return;
    if (pixel_forced_zero(A, oi, oj)) return;
    constexpr int kh = (K - 1) / 2, kk = K * K;
    const int md = A.max_dist, ms = A.ms, ns = A.ns;
    double cs_ = 0, s1 = 0, s2 = 0, nm = 0, ka = 0, kb = 0;
    const int q0 = oj - kh;
    const unsigned char* cf = cfl + (q0 - C0);
    for (int ki = 0; ki < K; ++ki) {
        const int p = oi - kh + ki;
        const int d0 = q0 - p;                                       // diagonal of the window row's first pixel
        const double* row = tile + (p - P0) * DN + (d0 - D0);
        const double* w = wl + ki * K;
        const bool rf = rfl[p - P0] != 0;
#pragma unroll
        for (int kj = 0; kj < K; ++kj) {
            const double v = row[kj];
            cs_ = fma(v, w[kj], cs_);
            s1 += v;
            s2 = fma(v, v, s2);
            const int d = d0 + kj;
            bool m = d < 0 ? d >= -K : ((rf | (cf[kj] != 0)) & (d <= md));
            if constexpr (EDGE) {
                const int q = q0 + kj;
                const bool inside = ((unsigned)p < (unsigned)ms) & ((unsigned)q < (unsigned)ns);
                const bool m_out = (q >= ns) ? (p >= ms - md - 2) : ((p < 0) & ((q < 0) | (q < md + K)));
                m = (inside ? ((rf | (cf[kj] != 0)) & (d >= 0) & (d <= md)) : m_out) | ((d < 0) & (d >= -K));
            }
            nm += m ? 1.0 : 0.0;
            ka += m ? w[kk + kj] : 0.0;
            kb += m ? w[2 * kk + kj] : 0.0;
        }
    }
    r = pearson_from_sums<double>(cs_, s1, s2, nm, ka, kb, A.ks, true, &nobs);
}

// The 17 x 17 walk of a run on at most two neighbouring diagonals (borders: every template, every sub-matrix), written
// for the instruction-issue floor: the interior walk above is ~25 instructions per window pixel -- five LDS reads (pixel,
// three weights, a column flag), the predicate from compares, six selects -- and the kernel around it was the second
// largest of a genome step (0.8 ms of device time on the 23-block genome).  Here
//   * the tile is staged TRANSPOSED (tileT[(d - D0) * kRunRP + (p - P0)]): the lanes of a wave are consecutive rows of
//     one or two diagonals, so an LDS read takes consecutive doubles; kRunRP = 146 rows (256 entries on two diagonals
//     span 129 rows when they start on the second one, + 16 of the template) = 18 (mod 32) doubles puts the second
//     diagonal's lanes on the other 32 banks but for two doubles;
//   * the weights come through the scalar unit (the same address for every lane: s_load from the template's table);
//   * the predicate of a window row is ONE 17-bit word per lane, built from ranges once per row (the EDGE rules of
//     missing_from_flags included), nm is its population count (exact: the sums of 1.0 it replaces are integers),
//     and ka / kb take fma(w, m ? 1.0 : 0.0, .) -- w * 1.0 is exact and x + 0.0 == x, so the sums are bit for bit those
//     of `ka += m ? w : 0.0`;
// 7 vector instructions per window pixel (3 sums, 2 for the multiplier, 2 mask sums) and one LDS read.  Same operations
// on the same values in the same order as rescore_pixel_lane: identical results.
constexpr int kRunRP = 146;

__device__ __forceinline__ unsigned run_range_bits(int a, int b)     // bits k with a <= k < b, clipped to 0 .. 17
{
    const int ac = min(max(a, 0), 17), bc = min(max(b, 0), 17);
    return bc > ac ? ((1u << bc) - 1u) & ~((1u << ac) - 1u) : 0u;
}

template <bool EDGE>
__device__ __forceinline__ void rescore_run17(const CorrArgs<double>& A, const double* tileT, const unsigned char* rfl,
                                              const unsigned char* cfl, int P0, int C0, int D0, int oi, int oj, double& r,
                                              double& nobs)
{
    constexpr int K = 17, kh = 8, kk = K * K;
    constexpr unsigned all = (1u << K) - 1u;
    r = 0.0;
    nobs = A.ks.n;
    if (EDGE && !((oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns))) return;
    if (pixel_forced_zero(A, oi, oj)) return;
    typedef const __attribute__((address_space(4))) double DblC;
    DblC* const w = (DblC*)(unsigned long long)A.w;
    const int ms = A.ms, ns = A.ns, md = min(A.max_dist, 1 << 20);
    const int q0 = oj - kh, p0 = oi - kh;
    unsigned cbits = 0u, qin = all, qge = 0u, qlt = 0u;
#pragma unroll
    for (int kj = 0; kj < K; ++kj) {
        cbits |= (cfl[q0 - C0 + kj] != 0 ? 1u : 0u) << kj;
        if constexpr (EDGE) {
            const int q = q0 + kj;
            qin &= ~(((unsigned)q < (unsigned)ns ? 0u : 1u) << kj);
            qge |= (q >= ns ? 1u : 0u) << kj;
            qlt |= (((q < 0) | (q < md + K)) ? 1u : 0u) << kj;
        }
    }
    double cs_ = 0, s1 = 0, s2 = 0, ka = 0, kb = 0;
    int nm_i = 0;
    const double* row = tileT + (q0 - p0 - D0) * kRunRP + (p0 - P0);         // window row 0, kj = 0; one row on: 1 - kRunRP
#pragma unroll 1
    for (int ki = 0; ki < K; ++ki, row -= kRunRP - 1) {
        const int p = p0 + ki;
        const int dl = q0 - p;                                               // diagonal of the row's first pixel
        const unsigned rc = rfl[p - P0] != 0 ? all : cbits;
        const unsigned below = run_range_bits(-K - dl, -dl);                 // -K <= d < 0
        const unsigned band = run_range_bits(-dl, md - dl + 1);              // 0 <= d <= max_dist
        unsigned M = below | (rc & band);
        if constexpr (EDGE) {
            const unsigned in_b = ((unsigned)p < (unsigned)ms) ? qin : 0u;
            const unsigned out_b = (qge & (p >= ms - md - 2 ? all : 0u)) | (~qge & (p < 0 ? qlt : 0u));
            M = (in_b & rc & band) | (~in_b & out_b & all) | below;
        }
        nm_i += __builtin_popcount(M);
        DblC* const wr = w + ki * K;
#pragma unroll
        for (int kj = 0; kj < K; ++kj) {
            const double v = row[kj * kRunRP];
            cs_ = fma(v, wr[kj], cs_);
            s1 += v;
            s2 = fma(v, v, s2);
            const int sel = ((int)(M << (31 - kj))) >> 31;                   // bit kj of M as 0 / -1
            const double mf = __hiloint2double(sel & 0x3FF00000, 0);        // 0.0 / 1.0
            ka = fma(wr[kk + kj], mf, ka);
            kb = fma(wr[2 * kk + kj], mf, kb);
        }
    }
    r = pearson_from_sums<double>(cs_, s1, s2, (double)nm_i, ka, kb, A.ks, true, &nobs);
}

// The same walk for T templates on the SAME pixels (the three templates of borders: detect_foci_batch_templates lists the pixels
// of the scanned diagonals once per template, (&A)[k * tab_step] is template k's argument block -- the same matrix, another set of
// weights and statistics).  The window's 289 pixels are read from LDS once, their squares and sums and the 17-bit predicate words
// are formed once; per template remain the three multiply-adds of a window pixel (and the scalar loads of its weights): 13
// vector instructions per window pixel for three templates instead of 21, a third of the LDS reads and of the tile staging.
// Every accumulator sees the operations of rescore_run17 in the same order: identical results.
template <bool EDGE, int T>
__device__ __forceinline__ void rescore_run17_multi(const CorrArgs<double>& A, int tab_step, const double* tileT, const unsigned char* rfl,
                                                    const unsigned char* cfl, int P0, int C0, int D0, int oi, int oj, double (&r)[T])
{
    constexpr int K = 17, kh = 8, kk = K * K;
    constexpr unsigned all = (1u << K) - 1u;
#pragma unroll
    for (int k = 0; k < T; ++k) r[k] = 0.0;
    if (EDGE && !((oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns))) return;
    if (pixel_forced_zero(A, oi, oj)) return;
    typedef const __attribute__((address_space(4))) double DblC;
    DblC* w[T];
#pragma unroll
    for (int k = 0; k < T; ++k) w[k] = (DblC*)(unsigned long long)(&A)[k * tab_step].w;
    const int ms = A.ms, ns = A.ns, md = min(A.max_dist, 1 << 20);
    const int q0 = oj - kh, p0 = oi - kh;
    unsigned cbits = 0u, qin = all, qge = 0u, qlt = 0u;
#pragma unroll
    for (int kj = 0; kj < K; ++kj) {
        cbits |= (cfl[q0 - C0 + kj] != 0 ? 1u : 0u) << kj;
        if constexpr (EDGE) {
            const int q = q0 + kj;
            qin &= ~(((unsigned)q < (unsigned)ns ? 0u : 1u) << kj);
            qge |= (q >= ns ? 1u : 0u) << kj;
            qlt |= (((q < 0) | (q < md + K)) ? 1u : 0u) << kj;
        }
    }
    double cs_[T], ka[T], kb[T], s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < T; ++k) cs_[k] = ka[k] = kb[k] = 0.0;
    int nm_i = 0;
    const double* row = tileT + (q0 - p0 - D0) * kRunRP + (p0 - P0);
#pragma unroll 1
    for (int ki = 0; ki < K; ++ki, row -= kRunRP - 1) {
        const int p = p0 + ki;
        const int dl = q0 - p;
        const unsigned rc = rfl[p - P0] != 0 ? all : cbits;
        const unsigned below = run_range_bits(-K - dl, -dl);
        const unsigned band = run_range_bits(-dl, md - dl + 1);
        unsigned M = below | (rc & band);
        if constexpr (EDGE) {
            const unsigned in_b = ((unsigned)p < (unsigned)ms) ? qin : 0u;
            const unsigned out_b = (qge & (p >= ms - md - 2 ? all : 0u)) | (~qge & (p < 0 ? qlt : 0u));
            M = (in_b & rc & band) | (~in_b & out_b & all) | below;
        }
        nm_i += __builtin_popcount(M);
        double v[K], mf[K];
#pragma unroll
        for (int kj = 0; kj < K; ++kj) {
            v[kj] = row[kj * kRunRP];
            s1 += v[kj];
            s2 = fma(v[kj], v[kj], s2);
            const int sel = ((int)(M << (31 - kj))) >> 31;
            mf[kj] = __hiloint2double(sel & 0x3FF00000, 0);
        }
#pragma unroll
        for (int k = 0; k < T; ++k) {
            DblC* const wr = w[k] + ki * K;
#pragma unroll
            for (int kj = 0; kj < K; ++kj) {
                cs_[k] = fma(v[kj], wr[kj], cs_[k]);
                ka[k] = fma(wr[kk + kj], mf[kj], ka[k]);
                kb[k] = fma(wr[2 * kk + kj], mf[kj], kb[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < T; ++k) {
        double nobs;
        r[k] = pearson_from_sums<double>(cs_[k], s1, s2, (double)nm_i, ka[k], kb[k], (&A)[k * tab_step].ks, true, &nobs);
    }
}

// KN > 0: template width known at compile time (17: every built-in 2-D / 1-D template but the 15 x 15 hairpin) -- the
// row loop is unrolled, so its 17 LDS reads are issued together instead of one dependent read per multiply-add (the
// runtime-size loop ran at a third of the instruction-issue floor).
template <int KN = 0>
__device__ __forceinline__ void rescore_pixel_lane_lds(const CorrArgs<double>& A, const double* tile, const double* wl,
                                                       const unsigned char* rfl, const unsigned char* cfl, int P0, int C0, int D0,
                                                       int DN, int oi, int oj, double& r, double& nobs)
{
    r = 0.0;
    nobs = A.ks.n;
    const bool inside = (oi >= 0) & (oi < A.ms) & (oj >= 0) & (oj < A.ns);
    if (inside && !pixel_forced_zero(A, oi, oj)) {
        const int km = A.km, kn = KN ? KN : A.kn, kk = km * kn;
        const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
        const bool masked = A.mask_mode != 0;
        const bool bins = A.mask_mode == 1;
        const bool lean = bins && A.sym_upper && A.full && A.max_dist >= 0;
        const int ms = A.ms, ns = A.ns, md = A.max_dist, big_k = max(km, kn), skew = kn - km;
        double cs_ = 0, s1 = 0, s2 = 0, nm = 0, ka = 0, kb = 0;
        int e = 0;
        for (int ki = 0; ki < km; ++ki) {
            const int p = oi - kh + ki;
            const double* row = tile + (p - P0) * DN + ((oj - kw) - p - D0);
            const bool rf = bins && rfl[p - P0] != 0;
            const bool in_r = (p >= 0) & (p < ms);
            auto step = [&](int kj) {
                const int q = oj - kw + kj;
                const double v = row[kj];
                cs_ = fma(v, wl[e], cs_);
                s1 += v;
                s2 = fma(v, v, s2);
                if (masked) {
                    bool m;
                    if (lean) {
                        const bool cf = cfl[q - C0] != 0;
                        const int d = q - p;
                        const bool in_c = (q >= 0) & (q < ns);
                        const bool m_in = (rf | cf) & (d >= 0) & (d <= md);
                        const bool m_out = (q >= ns) ? (p >= ms - md - 2) : ((p < 0) & ((q < 0) | (q < md + kn)));
                        m = (in_r & in_c) ? m_in : m_out;
                        const int off = d + skew;
                        m = m | ((off <= -1) & (off >= -big_k));
                    } else if (bins) {
                        m = missing_from_flags(A, p, q, rf, cfl[q - C0] != 0, false, true);
                    } else {
                        m = missing_pred(A, p, q);
                    }
                    if (m) {
                        nm += 1.0;
                        ka += wl[kk + e];
                        kb += wl[2 * kk + e];
                    }
                }
                ++e;
            };
            if constexpr (KN > 0) {
#pragma unroll
                for (int kj = 0; kj < KN; ++kj) step(kj);
            } else {
                for (int kj = 0; kj < kn; ++kj) step(kj);
            }
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
// ---- cs_stage.hip: all intra blocks of a genome staged by three launches ------------------------------------
struct StageBlock {
    long long row0;          // first genome bin of the block (its rows and its columns start there)
    int n;                   // bins
    int keep;                // last kept diagonal (diag_trim)
    int n_diags;             // entries of the distance law: min(n, keep + 1)
    int dense;               // 1: dense rows (slot = column), 0: diagonal band from diagonal 0 (slot = diagonal)
    int width;               // slots that can hold a pixel: n (dense) or n_diags (band)
    int group0, n_groups;    // filled by enqueue_stage_blocks
    long long ld;            // row pitch in elements (slots width .. ld - 1 are zero padding)
    double* band64;          // outputs, either may be null
    float* band32;
    double* law;             // n_diags values
    long long ld64;          // row pitch of band64 (= ld unless only its first w64 diagonals are written)
    int w64;                 // band layout: band64 receives the slots 0 .. w64 - 1 only (0: all of them)
    int counts;              // band32 receives the block's RAW COUNTS, written by the law pass (cs_device.h CountsHeader in the 128
                             // bytes in front of it; `law` then holds n_diags values + the n_diags + 2 of their reciprocals); the
                             // tiler has nothing to write for such a block
    LazyBand* lazy;          // descriptor of the block's lazily evaluated float64 band, written by stage_finish_kernel (or null)
};
// what a LazyBand points to besides its block (the genome's pixel table)
struct LazySource {
    const long long* indptr;
    const int* indices;
    const void* data;
    const double* weight;
    double max_val;
    int data_is_f64;
};
struct StageGroup {
    int block, row_begin, row_end;
};
size_t stage_scratch_bytes(int n_blocks, int n_groups, int pitch, long long n_rows);
size_t stage_table_bytes(int n_blocks, int n_groups);        // page-locked host bytes for h_tables
int enqueue_stage_blocks(const long long* indptr, const int* indices, const void* data, int data_is_f64, const double* weight,
                         long long n_rows, StageBlock* h_blocks, int n_blocks, double max_val, int rows_per_group, int n_cu,
                         void* scratch, void* h_tables, hipStream_t stream, std::vector<char>* uploaded = nullptr);

// 2-D patterns of many sub-matrices: candidates as composite keys (block << shift) + row * ns + col
size_t keyed_batch_scratch_bytes(int n_blocks, long long n_total);
// (segmented lists whose lengths are only known on the device when the chain is enqueued: see enqueue_foci_keyed_batch)
struct DeferredSegments {
    const long long* d_counts = nullptr;     // device: the blocks' candidate counters
    const long long* h_base = nullptr;       // page-locked host memory: the regions' starts in d_keys
    const long long* h_cap = nullptr;        // page-locked host memory: the regions' rooms
    long long bound = 0;                      // candidates the launches are sized for
    long long* h_counts_out = nullptr;        // page-locked, 64 entries: [b] counts, [60] total, [61] status flags
    bool tab_uploaded = false;                // upload_keyed_batch_table ran on a stream ordered before this chain
};
bool keyed_batch_deferred_available();
int upload_keyed_batch_table(const CorrArgs<double>* h_tab, int n_blocks, long long n_total, void* scratch, hipStream_t stream);
int enqueue_foci_keyed_batch(const CorrArgs<double>* h_tab, int n_blocks, const long long* d_keys, long long n_total, int shift,
                             double pearson, int min_size, int diag_only, int inter, void* scratch, FocusRec* rec, long long rec_cap,
                             double* windows, long long win_cap, long long* h_counts, hipStream_t stream,
                             const long long* h_base = nullptr, const long long* h_seg = nullptr, const DeferredSegments* deferred = nullptr);

// row-major keys (row * ns + col) -> coordinates
int launch_decode_keys(const long long* keys, long long n, int ns, int* rows, int* cols, hipStream_t stream);

int launch_compact_ge(const MatView& corr, int corr_is_f64, int ms, int ns, double threshold,
                      int lo_diag, int hi_diag, int* rows, int* cols, double* vals, long long cap,
                      long long* count, int n_cu, hipStream_t stream);
// n_ptr (optional, device): the number of listed pixels actually valid; n_px then only sizes the launch
// run != 0: the list is a run of neighbouring pixels (enumerated diagonals): one lane per pixel
int launch_rescore_f64(const CorrArgs<double>& A, const int* rows, const int* cols, long long n_px,
                       double* out_corr, double* out_nobs, hipStream_t stream, const long long* n_ptr = nullptr, int run = 0);
// the same from pixel keys (row * ns + col): rows[], cols[] and the scores of the first min(n_px, *n_ptr) keys; *count_copy = *n_ptr;
// *zero_next = 0 when given (the counter the next call will use)
int launch_rescore_f64_keys(const CorrArgs<double>& A, const long long* keys, int ns, long long n_px, int* rows, int* cols,
                            double* out_corr, const long long* n_ptr, long long* count_copy, long long* zero_next, hipStream_t stream);
int launch_csr_band_extent(const CsrView& M, int lo_diag, int hi_diag, long long* begin, long long* end,
                           hipStream_t stream);
int launch_law_finish(const double* sum, const long long* cnt, int n, double* law, hipStream_t stream);
// largest |pixel| of rows x cols values (as the bits of a float: a NaN or an infinity compares above every finite value),
// combined into *d_peak_bits with an atomic maximum (zero it first)
int launch_peak_rows(const void* src, int src_is_f64, long long ld_src, int rows, int cols, int n_cu, unsigned* d_peak_bits,
                     hipStream_t stream);
int launch_narrow_rows(const double* src, long long ld_src, float* dst, long long ld_dst, int rows, int cols, int n_cu,
                       hipStream_t stream);

// median of the stored values of a view (cs_foci.hip)
int csr_median(const CsrView& M, int n_cu, hipStream_t stream, void* (*grow)(void*, size_t), void* user, double* h_median);
int csr_median_many(const CsrView* views, int n, int n_cu, hipStream_t stream, void* (*grow)(void*, size_t), void* user, double* h_medians);

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
int enqueue_quantify_batch(const CorrArgs<double>* d_tab, const int* d_inter, const int* d_blk, const int* d_rows, const int* d_cols,
                           long long n, double* d_score, double* d_nobs, FocusRec* d_rec, double* d_windows, hipStream_t stream);
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
    int skip_edge;                  // edge mode, but the records of the edge diagonals are not built (the masked tile kernel
                                    // forms those corrections itself from the rim tables of the weights, cs_launch.h)
    TC* fix_lo;
    TC* fix_hi;
    int top, bot0, width, x_band, x_lo, side;   // frame corrections
    TC* fix_rows;
    TC* fix_cols;
    int b_tab, b_edge, b_rows;      // block ranges (filled by the launcher); b_tab = b_rows + blocks of the column table
    // a call on a row window (a rank's share of a row-split block): only the entries its tiles read -- rows r_lo .. r_hi - 1 of the
    // row table, columns c_lo .. c_hi - 1 of the column table (0, 0: all), and the frame rows the window reaches
    int r_lo, r_hi, c_lo, c_hi;
    int skip_top, skip_bot;         // frame corrections of the first rows / of the rows >= bot0 are not wanted
};
template <typename TC>
int launch_mask_prep(MaskPrepArgs<TC> P, hipStream_t stream);
// the same for several matrices in ONE launch (the blocks of a multi-block tile launch): mask_prep_blocks fills a matrix's workgroup
// ranges and returns their number; launch_mask_prep_batch copies the table (first[], arguments) through h_tab (page-locked) to d_tab
// (mask_prep_table_bytes each) and launches
template <typename TC>
int mask_prep_blocks(MaskPrepArgs<TC>& P);
size_t mask_prep_table_bytes(int n);
// lead_bytes: bytes in front of h_tab / d_tab (the same offset in both buffers) that travel in the same copy; zero / zero_bytes: 8-byte
// words the kernel clears (a memset where nothing is launched)
int launch_mask_prep_batch(const MaskPrepArgs<float>* args, const int* n_groups, int n, void* h_tab, void* d_tab, hipStream_t stream,
                           size_t lead_bytes = 0, void* zero = nullptr, size_t zero_bytes = 0);

}  // namespace cs
