// cs_mask_prep.hip -- tables for the factorised mask sums of the streaming kernel (MODE 2 of
// cs_corr_stream.h).
//
// With a per-bin missing mask (reference preprocessing.py:535 make_missing_mask: a pixel is
// missing when its row bin or its column bin is undetectable) the mask-weighted template sums of
// the masked Pearson branch (detection.py:1020-1060) factorise for every window that lies inside
// the matrix:
//     sum_{ki,kj} (r[i+ki] | c[j+kj]) W[ki][kj] = RA[i] + CA[j] - sum_kj c[j+kj] U_i[kj]
//     U_i[kj] = sum_ki r[i+ki] W[ki][kj],  RA[i] = sum_kj U_i[kj],  CA[j] = sum_kj c[j+kj] colsum_W[kj]
// (window indices are centred: i + ki means i - (K-1)/2 + ki).  `mask_rowcol_tables_kernel`
// builds the per-row and per-column tables for the two weight sets of the masked branch.
//
// In sym_upper mode the reference restricts that mask to diagonals 0..max_dist and flags the K
// sub-diagonals -K..-1 (preprocessing.py:404 frame_missing_mask; cs_device.h missing_pred).
// `mask_edge_fix_kernel` evaluates, for the output diagonals whose windows leave 0..max_dist,
// the exact difference  sum_window (missing_true - (r|c)) * {1, Wa, Wb}  once per pixel.
#include <algorithm>
#include <cstring>

#include <cstdlib>
#include "cs_device.h"
#include "cs_launch_aux.h"

namespace cs {

namespace {

constexpr int kMaxK = 17;
constexpr int kRowTabStride = 4;   // {nr, RA, RB, window row flags}; keep in sync with cs_corr_stream.h

// flags of the K bins around `centre` as a bit mask (bit k = flag of bin centre - KH + k, 0 outside 0..n-1)
__device__ __forceinline__ unsigned window_bits(const uint8_t* __restrict__ flags, int centre, int n, int K)
{
    const int KH = (K - 1) / 2;
    unsigned bits = 0;
    for (int k = 0; k < K; ++k) {
        const int p = centre - KH + k;
        if (p >= 0 && p < n && flags[p] != 0) bits |= 1u << k;
    }
    return bits;
}

// Row table: {number of flagged rows of the window, their Wa row sums, their Wb row sums, the flags as a number};
// column table: {number of flagged columns, their Wa column sums, their Wb column sums}.  Blocks [0, b_rows) take the
// rows, the rest the columns, so a wave works on ONE flag vector: every lane loads the flag of bin (wave base - KH +
// lane), the first 2 KH lanes a second one 64 bins further, and two ballots give every lane its window (2 byte loads
// per lane instead of K dependent ones); the sums come from K row / column sums of the weights held in LDS instead
// of 2 K loads per flagged bin.
template <typename TC>
__device__ __forceinline__ void mask_rowcol_tables(int block, int b_rows, const uint8_t* __restrict__ rr,
                                                   const uint8_t* __restrict__ cc, int ms, int ns, int K,
                                                   const TC* __restrict__ w, TC* __restrict__ rowtab, TC* __restrict__ coltab,
                                                   int r_lo, int r_hi, int c_lo, int c_hi)
{
    __shared__ TC sums[2][kMaxK];               // of Wa / Wb along the other axis
    const bool rows = block < b_rows;
    const int KH = (K - 1) / 2, kk = K * K;
    if (threadIdx.x < 2 * K) {
        const int set = threadIdx.x / K, k = threadIdx.x - set * K;
        const TC* ws = w + (1 + set) * kk;
        TC acc = TC(0);
        for (int t = 0; t < K; ++t) acc += rows ? ws[k * K + t] : ws[t * K + k];
        sums[set][k] = acc;
    }
    __syncthreads();
    const int x = (rows ? r_lo + block * (int)blockDim.x : c_lo + (block - b_rows) * (int)blockDim.x) + threadIdx.x;
    const int n = rows ? ms : ns;
    const uint8_t* __restrict__ flags = rows ? rr : cc;
    const int lane = threadIdx.x & 63;
    const int p0 = x - KH, p1 = x - KH + 64;
    const bool f0 = p0 >= 0 && p0 < n && flags[min(max(p0, 0), n - 1)] != 0;
    const bool f1 = lane < 2 * KH && p1 >= 0 && p1 < n && flags[min(max(p1, 0), n - 1)] != 0;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(f0), m1 = __builtin_amdgcn_ballot_w64(f1);
    const unsigned long long win = (m0 >> lane) | (lane ? (m1 << (64 - lane)) : 0ull);
    const unsigned bits = (unsigned)win & ((K >= 32) ? 0xffffffffu : ((1u << K) - 1u));
    if (x >= (rows ? r_hi : c_hi)) return;
    TC sa = TC(0), sb = TC(0);
    for (unsigned b = bits; b; b &= b - 1) {
        const int k = __ffs(b) - 1;
        sa += sums[0][k];
        sb += sums[1][k];
    }
    if (rows) {
        TC* row = rowtab + (size_t)x * kRowTabStride;
        row[0] = (TC)__popc(bits);
        row[1] = sa;
        row[2] = sb;
        row[3] = (TC)bits;   // flagged rows of the window as a number (17 bits: exact in float32)
    } else {
        coltab[x] = (TC)__popc(bits);
        coltab[(size_t)ns + x] = sa;
        coltab[2 * (size_t)ns + x] = sb;
    }
}

// one thread per (row i, edge slot e): e < K-1 -> diagonal e (lower edge), else diagonal
// hi_d0 + (e - (K-1)) (upper edge).  Only pixels whose window lies inside the matrix (the others
// are covered by the frame tables below).
//
// Window pixel (ki, kj) lies on diagonal d = D + kj - ki.  Row ki of the window has its first
// L = clamp(ki - D, 0, K) pixels below the main diagonal (flagged stripes in the reference, r|c in
// the regular model) and its pixels kj >= H = clamp(md - D + ki + 1, 0, K) beyond max_dist (never
// missing in the reference).  With prefix sums of the weights along rows (PW) and columns (QW)
// the correction is a per-diagonal constant minus one term per flagged row / column of the
// window (0.7 flagged bins per window at 2 % missing bins), instead of a loop over the triangle.
// one correction record {d n_missing, d ka, d kb, 0} as ONE 16-byte (float) / 32-byte (double) store: four scalar
// stores per thread wrote 4 of every 16 bytes of a line at a time (the 100 MB of rim records of a 200 000-row band
// took 100 us).  The tables start at multiples of 64 elements (cs_api.cpp prepare_regular_mask).
template <typename TC>
__device__ __forceinline__ void store_record(TC* dst, TC a, TC b, TC c)
{
    typedef TC v4 __attribute__((ext_vector_type(4)));
    v4 r;
    r[0] = a;
    r[1] = b;
    r[2] = c;
    r[3] = TC(0);
    *reinterpret_cast<v4*>(dst) = r;
}

template <typename TC>
__device__ __forceinline__ void mask_edge_fix(int block, int n_blocks, const uint8_t* __restrict__ rr,
                                              const uint8_t* __restrict__ cc, int ms, int ns, int K, int md, int hi_d0, int hi_w,
                                              const TC* __restrict__ w, TC* __restrict__ fix_lo, TC* __restrict__ fix_hi)
{
    constexpr int P = kMaxK + 1;
    // [set 0 = Wa, 1 = Wb]; PW[ki][m] = sum_{kj < m} W[ki][kj], QW[kj][m] = sum_{ki < m} W[ki][kj]
    __shared__ TC w_s[2][kMaxK * kMaxK], pw_s[2][kMaxK * P], qw_s[2][kMaxK * P], base_s[3][kMaxK];
    const int kk = K * K;
    for (int t = threadIdx.x; t < 2 * kk; t += blockDim.x) w_s[t / kk][t % kk] = w[kk + t];
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * K * (K + 1); t += blockDim.x) {
        const int set = t / (K * (K + 1)), r = (t / (K + 1)) % K, m = t % (K + 1);
        TC sp = TC(0), sq = TC(0);
        for (int x = 0; x < m; ++x) {
            sp += w_s[set][r * K + x];
            sq += w_s[set][x * K + r];
        }
        pw_s[set][r * P + m] = sp;
        qw_s[set][r * P + m] = sq;
    }
    __syncthreads();
    // flag-free value of the lower triangle per diagonal D < K-1: all of it is flagged stripes
    for (int t = threadIdx.x; t < K - 1; t += blockDim.x) {
        TC n = TC(0), a = TC(0), b = TC(0);
        for (int ki = 0; ki < K; ++ki) {
            const int L = min(K, max(0, ki - t));
            n += (TC)L;
            a += pw_s[0][ki * P + L];
            b += pw_s[1][ki * P + L];
        }
        base_s[0][t] = n;
        base_s[1][t] = a;
        base_s[2][t] = b;
    }
    __syncthreads();

    const int per_row = (K - 1) + hi_w;
    const int KH = (K - 1) / 2;
    // Half a wave per row when its records fit (K - 1 <= 16 lower, hi_w <= 16 upper): the flags of the row window and
    // of the two 32-column stretches the records' column windows lie in are then loaded once per half wave -- one byte
    // per lane, a ballot -- instead of 2 x 17 dependent byte loads per record (3 loads instead of 34; the records were
    // 0.08 of the 0.11 ms of this kernel on a 200 000-row band).
    const bool coop = (K - 1) <= 16 && hi_w <= 16;
    const int slots = coop ? 32 : per_row;
    // grid-stride: the prefix tables above are built once per block, so a block handles many pixels
    for (long long id = (long long)block * blockDim.x + threadIdx.x; id < (long long)ms * slots;
         id += (long long)n_blocks * blockDim.x) {
    const int i = (int)(id / slots);
    const int e = (int)(id - (long long)i * slots);
    unsigned co_r = 0, co_lo = 0, co_hi = 0;
    if (coop) {
        // lane l of the half wave: row flag i - KH + l, column flags i - KH + l and i + hi_d0 - KH + l
        const int sh = (threadIdx.x & 32);
        const int pr = i - KH + e, ql = i - KH + e, qh = i + hi_d0 - KH + e;
        const bool fr = e < K && pr >= 0 && pr < ms && rr[min(max(pr, 0), ms - 1)] != 0;
        const bool fl = ql >= 0 && ql < ns && cc[min(max(ql, 0), ns - 1)] != 0;
        const bool fh = qh >= 0 && qh < ns && cc[min(max(qh, 0), ns - 1)] != 0;
        co_r = (unsigned)(__builtin_amdgcn_ballot_w64(fr) >> sh);
        co_lo = (unsigned)(__builtin_amdgcn_ballot_w64(fl) >> sh);
        co_hi = (unsigned)(__builtin_amdgcn_ballot_w64(fh) >> sh);
        if (e >= per_row) continue;
    }
    int D;
    TC* dst;
    if (e < K - 1) {
        D = e;
        dst = fix_lo + ((size_t)i * (K - 1) + e) * 4;
    } else {
        D = hi_d0 + (e - (K - 1));
        dst = fix_hi + ((size_t)i * hi_w + (e - (K - 1))) * 4;
    }
    const int j = i + D;
    TC fn = TC(0), fa = TC(0), fb = TC(0);
    if (i >= KH && i + KH < ms && j >= KH && j + KH < ns) {
        if (D >= 0 && D < K - 1) {
            fn = base_s[0][D];
            fa = base_s[1][D];
            fb = base_s[2][D];
        }
        const unsigned kmask = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
        const unsigned rbits = coop ? (co_r & kmask) : window_bits(rr, i, ms, K);
        const unsigned cbits = coop ? (((e < K - 1) ? (co_lo >> e) : (co_hi >> (e - (K - 1)))) & kmask) : window_bits(cc, j, ns, K);
        // flagged rows: every pixel of the row that lies in either triangle
        for (unsigned rb = rbits; rb; rb &= rb - 1) {
            const int ki = __ffs(rb) - 1;
            const int L = min(K, max(0, ki - D));
            const int H = min(K, max(0, md - D + ki + 1));
            fn -= (TC)(L + (K - H));
            fa -= pw_s[0][ki * P + L] + (pw_s[0][ki * P + K] - pw_s[0][ki * P + H]);
            fb -= pw_s[1][ki * P + L] + (pw_s[1][ki * P + K] - pw_s[1][ki * P + H]);
        }
        // flagged columns: the pixels of the column in either triangle whose row is not flagged.
        // lower triangle: kj < ki - D  <=>  ki >= kj + D + 1; upper: ki < kj - (md - D)
        for (unsigned cb = cbits; cb; cb &= cb - 1) {
            const int kj = __ffs(cb) - 1;
            const int lo_k = min(K, max(0, kj + D + 1));
            const int hi_k = min(K, max(0, kj - (md - D)));
            TC n = (TC)((K - lo_k) + hi_k);
            TC a = (qw_s[0][kj * P + K] - qw_s[0][kj * P + lo_k]) + qw_s[0][kj * P + hi_k];
            TC b = (qw_s[1][kj * P + K] - qw_s[1][kj * P + lo_k]) + qw_s[1][kj * P + hi_k];
            for (unsigned rb = rbits; rb; rb &= rb - 1) {
                const int ki = __ffs(rb) - 1;
                if (ki >= lo_k || ki < hi_k) {
                    n -= TC(1);
                    a -= w_s[0][ki * K + kj];
                    b -= w_s[1][ki * K + kj];
                }
            }
            fn -= n;
            fa -= a;
            fb -= b;
        }
    }
    store_record(dst, fn, fa, fb);
    }
}

// The framed per-bin predicate of cs_device.h missing_pred (mask_mode 1, full, square template)
// with the two bin flags passed in.
__device__ __forceinline__ bool missing_bins(int ms, int ns, int K, int sym_upper, int max_dist, int p, int q, bool rflag,
                                             bool cflag)
{
    const bool in_r = (p >= 0) & (p < ms), in_c = (q >= 0) & (q < ns);
    const int d = q - p;
    const bool have_md = max_dist >= 0;
    bool m;
    if (in_r & in_c) {
        m = rflag | cflag;
        if (sym_upper) {
            const int md = have_md ? max_dist : min(ms, ns);
            m = m & (d >= 0) & (d <= md);
        }
    } else if (sym_upper && have_md) {
        if (q >= ns) m = p >= ms - max_dist - 2;
        else if (p < 0) m = (q < 0) ? true : (q < max_dist + K);
        else m = false;
    } else {
        m = true;
    }
    if (sym_upper) m = m | ((d <= -1) & (d >= -K));
    return m;
}

// Frame tables: complete correction  sum_window (missing - (r|c)) * {1, Wa, Wb}  for the pixels
// whose window leaves the matrix (flags count as 0 outside the matrix in the regular model).
//   rows table: the first `top` output rows and the rows >= bot0, `width` entries each
//               (x = j for dense outputs, x = (j - i) - x_lo for band outputs)
//   cols table: (dense outputs) the first and last `side` columns of every row
// One thread per (pixel, window row): 15 pixels x 17 rows per 256-thread block, partial sums over
// the window row in registers, then a fixed-order sum over the rows through LDS.
constexpr int kFramePixPerBlock = 256 / kMaxK;

template <typename TC>
__device__ __forceinline__ void mask_frame_fix(int block, const uint8_t* __restrict__ rr, const uint8_t* __restrict__ cc, int ms,
                                               int ns, int K, int sym_upper, int max_dist, const TC* __restrict__ w, int top,
                                               int bot0, int width, int x_band, int x_lo, int side, int lazy,
                                               TC* __restrict__ fix_rows, TC* __restrict__ fix_cols, int skip_top, int skip_bot)
{
    __shared__ TC wa_s[kMaxK * kMaxK], wb_s[kMaxK * kMaxK];
    __shared__ TC part[kFramePixPerBlock][kMaxK][3];
    const int kk = K * K;
    for (int t = threadIdx.x; t < kk; t += blockDim.x) {
        wa_s[t] = w[kk + t];
        wb_s[t] = w[2 * kk + t];
    }
    __syncthreads();
    const int KHf = (K - 1) / 2;
    // lazy (band outputs with edge tables): below the top rows only the pixels whose window leaves the
    // matrix are looked up in this table (cs_corr_stream.h fix_fetch), i.e. the last KH rows and, on the
    // rows before them, the last KH columns; the other entries are never read and are not computed
    const int full_bot = lazy ? min(KHf, ms - bot0) : ms - bot0;      // trailing rows computed entirely
    const int right_rows = lazy ? (ms - bot0) - full_bot : 0;         // rows with only their last KH columns
    const long long n_top_px = (long long)top * width, n_bot_px = (long long)full_bot * width;
    const long long n_right_px = (long long)right_rows * KHf;
    const long long n_row_px = n_top_px + n_bot_px + n_right_px;
    const long long n_col_px = fix_cols ? (long long)ms * 2 * side : 0;
    const int slot = threadIdx.x / kMaxK, ki = threadIdx.x % kMaxK;
    // (a row window: the pixels of the top rows come first, then those of the bottom rows -- either part may be left out)
    const long long id = (skip_top ? n_top_px : 0) + (long long)block * kFramePixPerBlock + slot;
    bool live = slot < kFramePixPerBlock && id < (skip_bot ? n_top_px : n_row_px + n_col_px);
    int i = 0, j = -1;
    TC* dst = nullptr;
    if (live) {
        if (id < n_top_px + n_bot_px) {
            const int r = (int)(id / width), x = (int)(id - (long long)r * width);
            i = r < top ? r : (ms - full_bot) + (r - top);
            j = x_band ? i + x_lo + x : x;
            dst = fix_rows + ((long long)(i < top ? i : top + (i - bot0)) * width + x) * 4;
        } else if (id < n_row_px) {
            const long long c = id - n_top_px - n_bot_px;
            i = bot0 + (int)(c / KHf);
            j = ns - KHf + (int)(c % KHf);
            const int x = x_band ? j - i - x_lo : j;
            live = x >= 0 && x < width;
            dst = fix_rows + ((long long)(top + (i - bot0)) * width + (live ? x : 0)) * 4;
        } else {
            const long long c = id - n_row_px;
            i = (int)(c / (2 * side));
            const int x = (int)(c - (long long)i * 2 * side);
            j = x < side ? x : ns - 2 * side + x;
            dst = fix_cols + c * 4;
        }
    }
    const int KH = (K - 1) / 2;
    TC fn = TC(0), fa = TC(0), fb = TC(0);
    if (live && ki < K && j >= 0 && j < ns) {
        const unsigned cbits = window_bits(cc, j, ns, K);
        const int p = i - KH + ki;
        const bool rflag = p >= 0 && p < ms && rr[p] != 0;
        for (int kj = 0; kj < K; ++kj) {
            const bool cflag = (cbits >> kj) & 1u;
            const int f = (missing_bins(ms, ns, K, sym_upper, max_dist, p, j - KH + kj, rflag, cflag) ? 1 : 0) -
                          ((rflag | cflag) ? 1 : 0);
            if (f != 0) {
                fn += (TC)f;
                fa += (TC)f * wa_s[ki * K + kj];
                fb += (TC)f * wb_s[ki * K + kj];
            }
        }
    }
    if (slot < kFramePixPerBlock) {
        part[slot][ki][0] = fn;
        part[slot][ki][1] = fa;
        part[slot][ki][2] = fb;
    }
    __syncthreads();
    if (live && ki == 0) {
        TC n = TC(0), a = TC(0), b = TC(0);
        for (int r = 0; r < K; ++r) {
            n += part[slot][r][0];
            a += part[slot][r][1];
            b += part[slot][r][2];
        }
        store_record(dst, n, a, b);
    }
}

// All tables of one call in ONE launch: the three parts read only the flag vectors and the weights,
// so their blocks are independent and run side by side (three dependent launches used to cost 60 us
// of a 270 us call on a 50 000-bin band).  Blocks [0, b_tab) build the row / column tables,
// [b_tab, b_tab + b_edge) the edge corrections, the rest the frame corrections.
template <typename TC>
__global__ __launch_bounds__(256) void mask_prep_kernel(MaskPrepArgs<TC> P)
{
    const int b = blockIdx.x;
    if (b < P.b_tab) {
        mask_rowcol_tables<TC>(b, P.b_rows, P.rr, P.cc, P.ms, P.ns, P.K, P.w, P.rowtab, P.coltab, P.r_lo, P.r_hi, P.c_lo, P.c_hi);
    } else if (b < P.b_tab + P.b_edge) {
        mask_edge_fix<TC>(b - P.b_tab, P.b_edge, P.rr, P.cc, P.ms, P.ns, P.K, P.max_dist, P.hi_d0, P.hi_w, P.w, P.fix_lo,
                          P.fix_hi);
    } else {
        mask_frame_fix<TC>(b - P.b_tab - P.b_edge, P.rr, P.cc, P.ms, P.ns, P.K, P.sym_upper, P.max_dist, P.w, P.top, P.bot0,
                           P.width, P.x_band, P.x_lo, P.side, P.edge, P.fix_rows, P.fix_cols, P.skip_top, P.skip_bot);
    }
}

// The tables of SEVERAL matrices (the blocks of a genome) in one launch: tab[k] = the arguments of matrix k, its workgroups
// are first[k] .. first[k + 1].  One launch per block is 12 us alone and 30-90 us beside the staging kernels of a genome step (profiles/r05_genome_timeline.txt: 23 of
// them in a row outlast the staging and hold the tile launch back); one launch for all of them does not.
template <typename TC>
__global__ __launch_bounds__(256) void mask_prep_batch_kernel(const MaskPrepArgs<TC>* __restrict__ tab, const int* __restrict__ first, int n,
                                                              unsigned long long* __restrict__ zero, int zero_words)
{
    // (the caller's counters, cleared here instead of by a memset of their own in front of this launch: one operation less on the
    // lane the tile launch waits for)
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < zero_words; k += 256) zero[k] = 0ull;
    int lo = 0, hi = n - 1;
    const int g = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= g) lo = mid;
        else hi = mid - 1;
    }
    const MaskPrepArgs<TC> P = tab[lo];
    const int b = g - first[lo];
    if (b < P.b_tab) {
        mask_rowcol_tables<TC>(b, P.b_rows, P.rr, P.cc, P.ms, P.ns, P.K, P.w, P.rowtab, P.coltab, P.r_lo, P.r_hi, P.c_lo, P.c_hi);
    } else if (b < P.b_tab + P.b_edge) {
        mask_edge_fix<TC>(b - P.b_tab, P.b_edge, P.rr, P.cc, P.ms, P.ns, P.K, P.max_dist, P.hi_d0, P.hi_w, P.w, P.fix_lo,
                          P.fix_hi);
    } else {
        mask_frame_fix<TC>(b - P.b_tab - P.b_edge, P.rr, P.cc, P.ms, P.ns, P.K, P.sym_upper, P.max_dist, P.w, P.top, P.bot0,
                           P.width, P.x_band, P.x_lo, P.side, P.edge, P.fix_rows, P.fix_cols, P.skip_top, P.skip_bot);
    }
}

}  // namespace

// workgroup ranges of one matrix (b_rows, b_tab, b_edge filled in); returns the number of workgroups, -1: template too large
template <typename TC>
int mask_prep_blocks(MaskPrepArgs<TC>& P)
{
    if (P.K > kMaxK) return -1;
    if (P.r_hi <= P.r_lo || P.c_hi <= P.c_lo) {           // (no window: every row and column)
        P.r_lo = P.c_lo = 0;
        P.r_hi = P.ms;
        P.c_hi = P.ns;
    }
    if (P.fix_cols) P.skip_top = P.skip_bot = 0;          // (dense outputs: the column table of the frame follows the rows')
    P.b_rows = (P.r_hi - P.r_lo + 255) / 256;
    P.b_tab = P.b_rows + (P.c_hi - P.c_lo + 255) / 256;
    P.b_edge = 0;
    if (P.edge && !P.skip_edge) {
        const long long n = (long long)P.ms * ((P.K - 1) + P.hi_w);
        P.b_edge = (int)std::min<long long>((n + 255) / 256, 4096);
    }
    const int KH = (P.K - 1) / 2;
    const int full_bot = P.edge ? std::min(KH, P.ms - P.bot0) : P.ms - P.bot0;
    const long long n_top = (long long)P.top * P.width;
    const long long n_frame = (P.skip_top ? 0 : n_top) +
                              (P.skip_bot ? 0 : (long long)full_bot * P.width + (P.edge ? (long long)((P.ms - P.bot0) - full_bot) * KH : 0)) +
                              (P.fix_cols ? (long long)P.ms * 2 * P.side : 0);
    const long long b_frame = n_frame > 0 ? (n_frame + kFramePixPerBlock - 1) / kFramePixPerBlock : 0;
    const long long total = (long long)P.b_tab + P.b_edge + b_frame;
    return total > 0x3fffffffLL ? -1 : (int)total;
}

template <typename TC>
int launch_mask_prep(MaskPrepArgs<TC> P, hipStream_t stream)
{
    const int n = mask_prep_blocks(P);
    if (n < 0) return -1;
    hipLaunchKernelGGL(mask_prep_kernel<TC>, dim3((unsigned)n), dim3(256), 0, stream, P);
    return (int)hipGetLastError();
}

size_t mask_prep_table_bytes(int n) { return (((size_t)(n + 1) * sizeof(int) + 255) & ~(size_t)255) + (size_t)n * sizeof(MaskPrepArgs<float>); }

int launch_mask_prep_batch(const MaskPrepArgs<float>* args, const int* n_groups, int n, void* h_tab, void* d_tab, hipStream_t stream,
                           size_t lead_bytes, void* zero, size_t zero_bytes)
{
    // lead_bytes: what lies in front of the table in the same two buffers (the tile launch's argument table) travels with it in the
    // ONE copy; zero: counters the kernel clears.  Without a launch (no table, no workgroup) both are done here.
    auto no_launch = [&]() {
        hipError_t e0 = hipSuccess;
        if (lead_bytes) e0 = hipMemcpyAsync((char*)d_tab - lead_bytes, (const char*)h_tab - lead_bytes, lead_bytes, hipMemcpyHostToDevice, stream);
        if (e0 == hipSuccess && zero && zero_bytes) e0 = hipMemsetAsync(zero, 0, zero_bytes, stream);
        return (int)e0;
    };
    if (n <= 0) return no_launch();
    const size_t off = ((size_t)(n + 1) * sizeof(int) + 255) & ~(size_t)255;
    int* first = reinterpret_cast<int*>(h_tab);
    long long total = 0;
    for (int k = 0; k < n; ++k) {
        first[k] = (int)total;
        total += n_groups[k];
        if (total > 0x7fffffffLL) return -3;
    }
    first[n] = (int)total;
    if (total == 0) return no_launch();
    std::memcpy((char*)h_tab + off, args, (size_t)n * sizeof(MaskPrepArgs<float>));
    hipError_t e = hipMemcpyAsync((char*)d_tab - lead_bytes, (const char*)h_tab - lead_bytes, lead_bytes + mask_prep_table_bytes(n), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mask_prep_batch_kernel<float>, dim3((unsigned)total), dim3(256), 0, stream,
                       reinterpret_cast<const MaskPrepArgs<float>*>((const char*)d_tab + off), reinterpret_cast<const int*>(d_tab), n,
                       reinterpret_cast<unsigned long long*>(zero), (int)(zero ? zero_bytes / 8 : 0));
    return (int)hipGetLastError();
}

template int launch_mask_prep<float>(MaskPrepArgs<float>, hipStream_t);
template int launch_mask_prep<double>(MaskPrepArgs<double>, hipStream_t);

template int mask_prep_blocks<float>(MaskPrepArgs<float>&);
template int mask_prep_blocks<double>(MaskPrepArgs<double>&);

}  // namespace cs
