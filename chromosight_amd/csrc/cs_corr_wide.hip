// cs_corr_wide.hip -- sliding-window correlation on the matrix cores for templates of up to 33 x 33.
//
// The float32 class of cs_normxcorr2 / cs_xcorr2 for the templates `--win-size` makes (reference
// cli/chromosight.py:365-370, 689-695 -> preprocessing.py:731-807 resize_kernel: every template of a pattern is
// zoomed to win-size x win-size) and for the 19 x 19 ... 33 x 33 templates of API users: any km, kn <= 33, any
// container (dense / band, float32 / float64), per-bin or explicit masks, n_obs, row windows, candidate mode.
// Same recasting as cs_corr_mfma.hip (v_mfma_f32_16x16x32_f16, float16 head / tail pairs after a power-of-two
// scale per tile, float32 accumulation), with what a wider template changes:
//
//   * a 16-column output tile of a template row of up to 33 weights reads 16 + 32 = 48 staged columns: TWO k = 32
//     Toeplitz passes per template row (B_p[k][n] = W[s][32 p + k - n]); the A blocks of the second pass of column
//     tile c are those of the first pass of tile c + 2, so a wave reads 6 head blocks (not 8) per template row.  Of the
//     second pass only k = 0 .. 15 ever meet a weight (t <= 32): its idle half carries the signal's TAILS of the same
//     columns against a second copy of the weight heads, so heads x heads + tails x heads is one MFMA -- 5 per
//     template row and column tile instead of 6 (tools/ubench/mfma_rate.hip: the k = 16 MFMA shape issues no faster);
//   * 64 x 64 output pixels stage (64 + 32)^2 pixels: float16 planes of 96 rows x 112 halfs (pitch 224 B: the 4 x
//     16-lane groups of a ds_read_b128 hit 16 distinct 16-byte slots; columns 96 .. 111 are zero and are what the
//     last block of a row reads beyond the staged pixels).  Staging is row-regular: a staged row is one run of
//     consecutive slots of its stored row (dense: columns; band: diagonals shifted by the row), fetched as 16-byte
//     pieces at 4-byte alignment by a 10 x 24 thread grid; a piece that straddles the end of the stored run is fetched
//     from the clamped start and shifted in registers; tiles on the frame of the matrix under a mask, and explicit
//     masks, take a pixel-by-pixel form with the full predicate;
//   * 33 rows x 2 passes x {head, tail} x 1 KiB of B fragments (x 3 weight sets) do not fit the LDS next to the
//     planes: a wave loads the four fragments of a template row straight from global memory (the image is built
//     once per template by the host, cs_api.cpp ensure_wfrag_wide: 132 KB per set, L2-resident), one row ahead;
//   * box sums (sum x, sum x^2, missing pixels): horizontal all-ones Toeplitz pass over the wave's 48 input rows,
//     and the accumulator layout of that pass (a lane holds rows 4 g + v of ONE column) IS the B-operand layout of
//     the vertical pass when the contraction index is labelled (g, e) -> row 16 rb + 4 g + e: no LDS transposition.
//     The squares are formed as float16 pairs in packed float16 arithmetic (2.5 instructions per pixel, 6 through
//     float32): the kernel is bound by the instructions it issues, like its 17 x 17 siblings;
//   * masks.  General form: the missing predicate of every staged pixel (cs_device.h missing_from_flags: per-bin
//     flags or an explicit map, frames, diagonal limits -- preprocessing.py:404-498, 535-633) is a 0/1 plane, and the
//     mask-weighted template sums are two more correlations of that plane (blocks whose 16 x 16 sub-blocks hold no
//     flagged pixel are skipped).  Per-bin masks on a tile whose staged pixels all lie inside the matrix and the
//     diagonals 0 .. max_dist ("inner": all but the rim of a band) factorise:
//         missing(p, q) = r_p | c_q   =>   sum_missing W = sum_{s in R_i} rowsum_W[s] + sum_{t in C_j} colsum_W[t]
//                                                          - sum_{s in R_i, t in C_j} W[s][t]
//     two 1-D tables per tile (built from the 96 + 96 flags, held as bit words, and the template's row / column sums)
//     minus the flagged-row x flagged-column term, which a wave evaluates where it is needed (it meets one flagged row
//     per tile on average, a window half a flagged column): no plane, no mask MFMAs.  On wide bands the inner tiles run
//     in a launch of their own with a 45 KB LDS image -- three workgroups per CU -- and the rim in a second one;
//   * a wave whose 16 rows x 64 columns hold no produced pixel (the corners of a band's outer tiles) stops after
//     staging; tiles all of whose pixels are produced store without per-pixel predicates.
//
// One workgroup (4 waves) = one 64 x 64 output tile; wave w owns rows 16 w .. 16 w + 15 and four 16-column tiles.
// LDS 66 KB (masked, with the plane) / 45 KB (inner tiles) / 43 KB (no mask): two or three workgroups per CU; tiles
// are dealt to the 8 XCDs in contiguous ranges.  Measured (profiles/r06_template_kernels.txt): 21 x 21 dense 4096^2
// 0.165 ms (102 Gpixel/s), C4' 200 000 x 1001 masked 3.26 ms (61 Gpixel/s), 33 x 33 4.08 ms (49) -- the runtime-size
// kernel these calls took before: 0.84 ms and 18.7 ms at 21 x 21.
#include "cs_device.h"
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "cs_launch.h"

namespace cs {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte global access at 4-byte alignment
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef __fp16 hv2 __attribute__((ext_vector_type(2)));

constexpr int WD_T = 64;                      // output tile edge
constexpr int WD_R = 96;                      // staged rows / columns (tile + 32)
constexpr int WD_P = 112;                     // halfs per staged row
constexpr int WD_PLANE = WD_R * WD_P * 2;     // bytes of one float16 plane
constexpr int WD_PER_THREAD = (WD_R * WD_R) / 256;      // 36 staged pixels per thread
constexpr int WD_TAB = 6 * 64 * 4;            // per-tile mask tables: Tn, Ta, Tb (rows), Un, Ua, Ub (columns)
constexpr int WD_FLAGS = 192;                 // flags of the 96 staged rows, then of the 96 staged columns
constexpr int WD_SMEM_PLAIN = 2 * WD_PLANE + 64;
constexpr int WD_SMEM_MASKED = 3 * WD_PLANE + WD_TAB + WD_FLAGS + 64;
constexpr int WD_SMEM_INNER = 2 * WD_PLANE + WD_TAB + WD_FLAGS + 64;       // tile_mode 1
static_assert(3 * WD_SMEM_INNER <= 160 * 1024, "three workgroups per CU");
static_assert(WD_R * WD_R == 256 * WD_PER_THREAD, "staging loop");
static_assert(WD_SMEM_MASKED <= 80 * 1024, "two workgroups per CU");

#ifdef CS_WD_PROFILE
// Per-phase cycle stamps of every wave (tools/prof_wide_sections.py; `make prof`): the differences stay in registers and are
// added to one of 64 copies of the device counters when the wave ends (atomics on nine words from 16 000 waves, issued between
// the phases, queued in front of the fragment loads and were what the first version measured).  [15]: wave-tiles.
__device__ unsigned long long cs_wd_prof[64 * 16];
#define WD_STAMP(k)                                                        \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        tdelta_[k] += now_ - tprev_;                                        \
        tprev_ = now_;                                                     \
    } while (0)
#else
#define WD_STAMP(k)
#endif

__device__ __forceinline__ f4 mfma16(const h8& a, const h8& b, const f4& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ h8 as_h8(const uint4& v) { return __builtin_bit_cast(h8, v); }

// the head / tail float16 pair of four float32 sums as the LOWER half of a B operand (slots e = 0 .. 3; 4 .. 7 zero)
__device__ __forceinline__ void split_low(const f4& v, h8& hi, h8& lo)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)(v[e] - (float)h);
        hi[e + 4] = (_Float16)0.0f;
        lo[e + 4] = (_Float16)0.0f;
    }
}

// candidate mode: windows far below the tile's scale are outside the error model of the float16 pairs
// (cs_corr_mfma.hip cand_range_guard)
__device__ __forceinline__ float wide_range_guard(float r, float s2, float unscale, const KernelStats<float>& K)
{
    if (K.cand_cmin > 0.0f) r = ((int)(s2 > 0.0f) & (int)(s2 < K.n * (unscale * unscale) * 0.0625f)) ? 2.0f : r;
    return r;
}

}  // namespace

template <bool MASKED, bool TWO>
__global__ __launch_bounds__(256, 2) void corr_mfma_wide_kernel(const CorrArgs<float> A, const MfmaWideWeights E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* xh = reinterpret_cast<_Float16*>(smem);
    _Float16* xl = reinterpret_cast<_Float16*>(smem + WD_PLANE);
    _Float16* xm = reinterpret_cast<_Float16*>(smem + 2 * WD_PLANE);                     // MASKED only
    // (E.tile_mode 1: a launch that works on inner tiles only -- no plane, LDS for three workgroups per CU)
    const int n_planes = (MASKED && E.tile_mode != 1) ? 3 : 2;
    float* tab = reinterpret_cast<float*>(smem + n_planes * WD_PLANE);                          // MASKED only
    unsigned char* flg = reinterpret_cast<unsigned char*>(smem + n_planes * WD_PLANE + WD_TAB);  // MASKED only
    unsigned* red = reinterpret_cast<unsigned*>(smem + (MASKED ? n_planes * WD_PLANE + WD_TAB + WD_FLAGS : 2 * WD_PLANE));

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const bool band_out = A.out.layout == 1;
    // ---- tile of this workgroup: XCD x (workgroups x, x + 8, ...) takes the x-th eighth of the row-major tile list
    // (persistent workgroups were tried: inlined, the tile loop held everything that does not depend on the tile index in
    // registers -- 256 + 168 spilled; as a called function the arguments went through scratch; and the cost of one workgroup
    // per tile is small: the per-phase cycle stamps account for the whole call at the clock the kernel runs at)
    const int n_tiles = A.tiles_x * A.tiles_y;
    const int per = (int)gridDim.x >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= n_tiles) return;
    {
    const int by = t / A.tiles_x;
    const int bx = t - by * A.tiles_x;
    const int I0 = A.row_begin + by * WD_T;
    if (I0 >= A.row_end) return;
    const int J0 = band_out ? I0 + A.out_lo + bx * WD_T : bx * WD_T;
    if (J0 >= A.ns || J0 + WD_T <= 0) return;
    if (J0 + WD_T - 1 - I0 < A.out_lo || J0 - (I0 + WD_T - 1) > A.out_hi) return;   // no produced diagonal
#ifdef CS_WD_PROFILE
    unsigned long long tdelta_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    const int km = A.km, kn = A.kn;
    const int RH = WD_T - 1 + km, RW = WD_T - 1 + kn;            // staged rows / columns the tile's windows reach
    const int P0 = I0 - (km - 1) / 2, Q0 = J0 - (kn - 1) / 2;
    // rows no window of the row range [row_begin, row_end) reaches are not part of the input contract
    const int p_lo = A.row_begin - (km - 1) / 2, p_hi = A.row_end + (km - 1) - (km - 1) / 2;

    // ---- inner tile of a per-bin mask: every staged pixel inside the matrix, the row window and the diagonals
    //      0 .. max_dist, where the missing predicate is r_p | c_q and nothing else (cs_device.h missing_from_flags)
    bool inner = false;
    if constexpr (MASKED) {
        inner = A.mask_mode == 1 && !E.plane_only && P0 >= 0 && P0 >= p_lo && P0 + RH <= A.ms && P0 + RH <= p_hi && Q0 >= 0 &&
                Q0 + RW <= A.ns;
        if (inner && A.sym_upper) {
            const int md = A.max_dist >= 0 ? A.max_dist : min(A.ms, A.ns);
            const int dmin = Q0 - (P0 + RH - 1), dmax = Q0 + RW - 1 - P0;
            inner = dmin >= 0 && dmax <= md && dmin + (kn - km) >= 0;
        }
    }

    if constexpr (MASKED) {
        // the two launches of a per-bin mask (launch_corr_mfma_wide_f32): inner tiles with the small LDS image, then the rest
        if ((E.tile_mode == 1 && !inner) || (E.tile_mode == 2 && inner)) return;
    }
    if (tid < 4) red[tid] = 0u;
    if constexpr (MASKED) {
        if (A.mask_mode == 1) {
            if (tid < WD_R) flg[tid] = A.miss_row[min(max(P0 + tid, 0), A.ms - 1)];
            else if (tid < 2 * WD_R) flg[tid] = A.miss_col[min(max(Q0 + tid - WD_R, 0), A.ns - 1)];
        }
    }

    // ---- stage 96 x 96 pixels.  Two forms, one per workgroup:
    //   * row-regular (`fast`): a staged row is one run of consecutive slots of its stored row (dense: columns; band: diagonals
    //     shifted by the row), so a thread fetches 9 pieces of 4 pixels with 16-byte loads at 4-byte alignment; a piece that
    //     straddles the end of the stored run is fetched from the clamped start and shifted in registers.  Masks: none, or
    //     per-bin flags on a tile whose staged pixels all lie inside the matrix (no frame: the predicate is the flags, the
    //     diagonal limits and the sub-diagonal stripes -- cs_device.h missing_from_flags for in-matrix pixels).
    //   * general: one pixel at a time from clamped addresses, the missing predicate in full (frames, explicit maps).
    constexpr int FPIECES = 10;             // 16-byte pieces per thread of the row-regular form
    const bool band_in = A.sig.layout == 1;
    const int W_in = band_in ? A.sig.band_w : A.ns;       // slots a stored row holds
    const bool frame_free = P0 >= 0 && P0 >= p_lo && P0 + RH <= A.ms && P0 + RH <= p_hi && Q0 >= 0 && Q0 + RW <= A.ns;
    const bool fast = !E.plane_only_staging && W_in >= 4 && (!MASKED || (A.mask_mode == 1 && frame_free));
    float xv[4 * FPIECES];           // (the general form fills 36 slots; the ok bits of the others stay 0)
#pragma unroll
    for (int k = WD_PER_THREAD; k < 4 * FPIECES; ++k) xv[k] = 0.0f;
    unsigned long long ok_bits = 0, miss_bits = 0, mval_bits = 0;
    // row-regular form: thread (tr, tc) of 10 x 24 fetches the 4 pixels from column 4 tc of the rows tr, tr + 10, ..., tr + 90
    // (16 threads idle): the address advances by a constant from piece to piece
    const int tr = tid / 24, tc4 = (tid - tr * 24) * 4;
    const bool stager = tid < 240;
    if (fast) {
        const int row_lo = max(0, p_lo), row_hi = min(A.ms, p_hi);
        const int lo_in = band_in ? A.sig.band_lo : 0;
        // every one of the 96 x 96 pixels is stored (rows and columns beyond the 63 + km x 63 + kn the windows reach hold real
        // pixels then: they meet zero weights only)
        const bool whole = P0 >= row_lo && P0 + WD_R <= row_hi && Q0 >= 0 && Q0 + WD_R <= A.ns &&
                           (!band_in || (Q0 - (P0 + WD_R - 1) - lo_in >= 0 && Q0 + WD_R - 1 - P0 - lo_in < W_in));
        const long long step = 10 * A.sig.ld - (band_in ? 10 : 0);
        const long long at0 = ((long long)(P0 + tr) - A.sig.row0) * A.sig.ld + (band_in ? (Q0 + tc4) - (P0 + tr) - lo_in : Q0 + tc4);
#pragma unroll
        for (int k = 0; k < FPIECES; ++k) {
            const int r = tr + 10 * k;
            const bool live = stager && (k < 9 || r < WD_R);
            long long at = at0 + k * step;
            int sh = 0;
            bool oks[4] = {live, live, live, live};
            if (!whole) {
                const int p = P0 + r, q0 = Q0 + tc4;
                const bool rowok = live & (r < RH) & (p >= row_lo) & (p < row_hi);
                const int pc = min(max(p, row_lo), row_hi - 1);
                const int s0 = band_in ? q0 - p - lo_in : q0;
                const int s0c = min(max(s0, 0), W_in - 4);
                at = ((long long)pc - A.sig.row0) * A.sig.ld + s0c;
                sh = s0 - s0c;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    oks[e] = rowok & (tc4 + e < RW) & (s0 + e >= 0) & (s0 + e < W_in) & (q0 + e >= 0) & (q0 + e < A.ns);
            } else if (!live) {
                at = 0;
            }
            float v0, v1, v2, v3;
            if (A.sig_is_f64) {
                const double* src = reinterpret_cast<const double*>(A.sig.ptr) + at;
                const d2u lo2 = *reinterpret_cast<const d2u*>(src), hi2 = *reinterpret_cast<const d2u*>(src + 2);
                v0 = (float)lo2[0];
                v1 = (float)lo2[1];
                v2 = (float)hi2[0];
                v3 = (float)hi2[1];
            } else {
                const f4u v = *reinterpret_cast<const f4u*>(reinterpret_cast<const float*>(A.sig.ptr) + at);
                v0 = v[0];
                v1 = v[1];
                v2 = v[2];
                v3 = v[3];
            }
            if (sh != 0) {                      // (a piece on the rim of the stored run: element e sits at position sh + e of the fetch)
                const float w0 = v0, w1 = v1, w2 = v2, w3 = v3;
                v0 = sh == 1 ? w1 : sh == 2 ? w2 : w3;              // sh in 1 .. 3 (positions beyond 3 are not stored: zeroed below)
                v1 = sh == 1 ? w2 : sh == -1 ? w0 : w3;
                v2 = sh == 1 ? w3 : sh == -1 ? w1 : w0;             // sh == -2: w0
                v3 = sh == -1 ? w2 : sh == -2 ? w1 : w0;            // sh == -3: w0
            }
            const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (oks[e]) ok_bits |= 1ull << (4 * k + e);
                xv[4 * k + e] = vv[e];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < WD_PER_THREAD; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx / WD_R, c = idx - r * WD_R;
            const int p = P0 + r, q = Q0 + c;
            const bool inside = (r < RH) & (c < RW) & (p >= 0) & (p < A.ms) & (q >= 0) & (q < A.ns) & (p >= p_lo) & (p < p_hi);
            const long long off = inside ? mat_offset(A.sig, p, q) : -1;
            if (off >= 0) ok_bits |= 1ull << k;
            const long long o = off >= 0 ? off : 0;            // element 0 of the buffer always exists
            xv[k] = A.sig_is_f64 ? (float)reinterpret_cast<const double*>(A.sig.ptr)[o] : reinterpret_cast<const float*>(A.sig.ptr)[o];
            if constexpr (MASKED) {
                if (A.mask_mode == 2) {
                    const long long om = inside ? mat_offset(A.mask, p, q) : -1;
                    const unsigned char mv = reinterpret_cast<const unsigned char*>(A.mask.ptr)[om >= 0 ? om : 0];
                    if (om >= 0 && mv != 0) mval_bits |= 1ull << k;
                }
            }
        }
    }
    WD_STAMP(0);          // addresses, loads issued
    __syncthreads();      // flags and the zeroed words are in LDS

    // flags of the 96 staged rows / columns as bit words (every wave forms its own copy: wave-uniform scalars)
    unsigned long long rbits_lo = 0, cbits_lo = 0;
    unsigned rbits_hi = 0, cbits_hi = 0;
    if constexpr (MASKED) {
        if (inner) {
            rbits_lo = __ballot(flg[lane] != 0);
            rbits_hi = (unsigned)__ballot(lane < 32 && flg[64 + (lane & 31)] != 0);
            cbits_lo = __ballot(flg[WD_R + lane] != 0);
            cbits_hi = (unsigned)__ballot(lane < 32 && flg[WD_R + 64 + (lane & 31)] != 0);
        }
    }
    // bits i .. i + len - 1 (len <= 33, i <= 63) of a 96-bit word
    auto window_bits = [](unsigned long long lo, unsigned hi, int i, int len) -> unsigned long long {
        unsigned long long w = lo >> i;
        if (i) w |= (unsigned long long)hi << (64 - i);
        return w & ((1ull << len) - 1ull);
    };
    const bool has_cross = MASKED && inner && (rbits_lo | rbits_hi) != 0 && (cbits_lo | cbits_hi) != 0;
    unsigned long long cross_bits = 0;     // what the mask plane holds on a general tile: the missing predicate
    unsigned occ0 = 0, occ1 = 0;           // 16 x 16 blocks of the plane that hold a 1: bit 6 (r >> 4) + (c >> 4), 18 per word
    if constexpr (MASKED) {
        if (fast && inner) {
            // inner tile: missing = r_p | c_q on the pixels the windows reach, four at a time from the flag words (no plane)
#pragma unroll
            for (int k = 0; k < FPIECES; ++k) {
                const int r = min(tr + 10 * k, WD_R - 1), c4 = min(tc4, WD_R - 4);
                const unsigned fr = (unsigned)(r < 64 ? rbits_lo >> r : (unsigned long long)(rbits_hi >> (r - 64))) & 1u;
                const unsigned nib = (unsigned)(c4 < 64 ? cbits_lo >> c4 : (unsigned long long)(cbits_hi >> (c4 - 64))) & 0xfu;
                const unsigned need4 = r < RH ? (1u << min(max(RW - c4, 0), 4)) - 1u : 0u;
                const unsigned miss4 = (fr ? 0xfu : nib) & need4;
                miss_bits |= (unsigned long long)miss4 << (4 * k);
            }
        } else if (fast) {
            const int md = A.max_dist >= 0 ? A.max_dist : min(A.ms, A.ns);
            const int big_k = max(km, kn);
            const bool stripes_on = A.sym_upper && A.full;
#pragma unroll
            for (int k = 0; k < FPIECES; ++k) {
                const int r = min(tr + 10 * k, WD_R - 1), c4 = min(tc4, WD_R - 4);      // (idle threads, pieces beyond row 95: ok bits 0)
                const int d0 = (Q0 + c4) - (P0 + r);
                const bool fr = flg[r] != 0;
                const unsigned fc4 = *reinterpret_cast<const unsigned*>(flg + WD_R + c4);
                unsigned pl = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool fc = ((fc4 >> (8 * e)) & 0xffu) != 0;
                    const bool needed = (r < RH) & (c4 + e < RW);
                    bool miss, plane;
                    if (inner) {             // (no plane: the flagged-row x flagged-column term comes from the V table below)
                        miss = needed & (fr | fc);
                        plane = false;
                    } else {
                        const int d = d0 + e, off = d + (kn - km);
                        const bool in_d = !A.sym_upper | ((d >= 0) & (d <= md));
                        miss = needed & (((fr | fc) & in_d) | (stripes_on & (off <= -1) & (off >= -big_k)));
                        plane = miss;
                    }
                    if (miss) miss_bits |= 1ull << (4 * k + e);
                    if (plane) {
                        cross_bits |= 1ull << (4 * k + e);
                        pl = 1u;
                    }
                }
                if (pl) {
                    const int b = 6 * (r >> 4) + (c4 >> 4);
                    if (b < 18) occ0 |= 1u << b;
                    else occ1 |= 1u << (b - 18);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < WD_PER_THREAD; ++k) {
                const int idx = tid + 256 * k;
                const int r = idx / WD_R, c = idx - r * WD_R;
                const int p = P0 + r, q = Q0 + c;
                bool fr = false, fc = false;
                if (A.mask_mode == 1) {
                    fr = flg[r] != 0;
                    fc = flg[WD_R + c] != 0;
                }
                bool miss, plane;
                if (inner) {
                    miss = fr | fc;
                    plane = false;
                } else {
                    const bool needed = (r < RH) & (c < RW) & (p >= p_lo) & (p < p_hi);
                    // (mval: the explicit map's byte where the map stores the pixel, so `stored` has nothing left to say)
                    miss = needed && missing_from_flags(A, p, q, fr, fc, (bool)((mval_bits >> k) & 1ull), true);
                    plane = miss;
                }
                if (miss) miss_bits |= 1ull << k;
                if (plane) {
                    cross_bits |= 1ull << k;
                    const int b = 6 * (r >> 4) + (c >> 4);
                    if (b < 18) occ0 |= 1u << b;
                    else occ1 |= 1u << (b - 18);
                }
            }
        }
    }
    WD_STAMP(1);          // barrier, mask bits
    float amax = 0.0f;
#pragma unroll
    for (int k = 0; k < 4 * FPIECES; ++k) {
        // the reference requires 0 at missing pixels (check_missing_mask); enforce it
        const float x = (((ok_bits & ~miss_bits) >> k) & 1ull) ? xv[k] : 0.0f;
        xv[k] = x;
        amax = fmaxf(amax, fabsf(x));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) atomicMax(&red[0], __float_as_uint(amax));
    if constexpr (MASKED) {
        if (occ0) atomicOr(&red[1], occ0);
        if (occ1) atomicOr(&red[2], occ1);
    }
    WD_STAMP(2);          // wait for the loads, maximum
    __syncthreads();
    int ex = 0;
    {
        const int e = (int)((red[0] >> 23) & 0xffu);
        if (e != 0 && e != 255) ex = 6 - (e - 127);
        ex = max(-100, min(100, ex));
    }
    const float scale = __uint_as_float((unsigned)(ex + 127) << 23);
    const float unscale = __uint_as_float((unsigned)(127 - ex) << 23);
    unsigned long long occ = 0;
    if constexpr (MASKED) {
        occ = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)red[1]) |
              ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)red[2]) << 18);
    }
    if (fast) {
        // heads by truncation, tails exact differences: head + tail carries 21-22 bits either way (cs_corr_mfma_body.inc)
#pragma unroll
        for (int k = 0; k < FPIECES; ++k) {
            const int r = tr + 10 * k;
            if (!stager || (k == 9 && r >= WD_R)) continue;
            const int o = r * WD_P + tc4;
            const float a0 = xv[4 * k] * scale, a1 = xv[4 * k + 1] * scale, a2 = xv[4 * k + 2] * scale, a3 = xv[4 * k + 3] * scale;
            const hv2 h01 = __builtin_amdgcn_cvt_pkrtz(a0, a1), h23 = __builtin_amdgcn_cvt_pkrtz(a2, a3);
            const hv2 t01 = __builtin_amdgcn_cvt_pkrtz(a0 - (float)h01[0], a1 - (float)h01[1]);
            const hv2 t23 = __builtin_amdgcn_cvt_pkrtz(a2 - (float)h23[0], a3 - (float)h23[1]);
            uint2 hw, tw;
            hw.x = __builtin_bit_cast(unsigned, h01);
            hw.y = __builtin_bit_cast(unsigned, h23);
            tw.x = __builtin_bit_cast(unsigned, t01);
            tw.y = __builtin_bit_cast(unsigned, t23);
            *reinterpret_cast<uint2*>(xh + o) = hw;
            *reinterpret_cast<uint2*>(xl + o) = tw;
            if (MASKED && !inner) {
                const unsigned m4 = (unsigned)(cross_bits >> (4 * k)) & 0xfu;
                uint2 mw;                                        // 0x3c00 = 1.0 in float16
                mw.x = ((m4 & 1u) ? 0x3c00u : 0u) | ((m4 & 2u) ? 0x3c000000u : 0u);
                mw.y = ((m4 & 4u) ? 0x3c00u : 0u) | ((m4 & 8u) ? 0x3c000000u : 0u);
                *reinterpret_cast<uint2*>(xm + o) = mw;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < WD_PER_THREAD; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx / WD_R, c = idx - r * WD_R;
            const int o = r * WD_P + c;
            const float xs = xv[k] * scale;
            const _Float16 h = (_Float16)xs;
            xh[o] = h;
            xl[o] = (_Float16)(xs - (float)h);
            if (MASKED && !inner) xm[o] = ((cross_bits >> k) & 1ull) ? (_Float16)1.0f : (_Float16)0.0f;
        }
    }
    if (tid < WD_R) {      // columns 96 .. 111 of every row: zeros
        const uint4 z = {0u, 0u, 0u, 0u};
        uint4* ph = reinterpret_cast<uint4*>(xh + tid * WD_P + WD_R);
        uint4* pl = reinterpret_cast<uint4*>(xl + tid * WD_P + WD_R);
        ph[0] = z;
        ph[1] = z;
        pl[0] = z;
        pl[1] = z;
        if (MASKED && !inner) {
            uint4* pm = reinterpret_cast<uint4*>(xm + tid * WD_P + WD_R);
            pm[0] = z;
            pm[1] = z;
        }
    }
    if constexpr (MASKED) {
        // per-tile tables of the factorised form (zeros on a general tile: the plane carries the whole predicate):
        //   tab[0..63] nr_i, [64..] RA_i, [128..] RB_i (flagged rows of row i's window: count, their Wa / Wb row sums),
        //   tab[192..] nc_j, CA_j, CB_j likewise for the columns
        if (tid < 128) {
            float cnt = 0.0f, sa = 0.0f, sb = 0.0f;
            if (inner) {
                const bool rows = tid < 64;
                const int i = tid & 63;
                const float* wa = E.sums + (rows ? 0 : 2 * 33);
                const float* wb = wa + 33;
                for (unsigned long long w = rows ? window_bits(rbits_lo, rbits_hi, i, km) : window_bits(cbits_lo, cbits_hi, i, kn); w; w &= w - 1ull) {
                    const int t = __builtin_ctzll(w);
                    cnt += 1.0f;
                    sa += wa[t];
                    sb += wb[t];
                }
            }
            const int base = tid < 64 ? 0 : 3 * 64;
            tab[base + (tid & 63)] = cnt;
            tab[base + 64 + (tid & 63)] = sa;
            tab[base + 128 + (tid & 63)] = sb;
        }
    }
    WD_STAMP(3);          // barrier, split, plane writes, tables
    __syncthreads();
    WD_STAMP(4);          // barrier

    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int wr0 = 16 * wv;                  // first staged row of the wave's windows
    // the 16 x 16 pixel blocks of this wave that hold a produced pixel (rows below row_end, columns inside the matrix, diagonals
    // out_lo .. out_hi): a wave without one -- the corners of a band's outer tiles -- has nothing left to do (no barrier follows)
    unsigned cmask = 0;
    {
        const int i_lo = I0 + wr0, i_hi = min(I0 + wr0 + 15, A.row_end - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j_lo = max(J0 + 16 * c, 0), j_hi = min(J0 + 16 * c + 15, A.ns - 1);
            if (i_lo <= i_hi && j_lo <= j_hi && j_hi - i_lo >= A.out_lo && j_lo - i_hi <= A.out_hi) cmask |= 1u << c;
        }
    }
    if (cmask != 0u) {
    constexpr int NCB = TWO ? 6 : 4;          // 16-column steps at which a wave's A blocks (16 rows x 32 columns) start
    const bool any_mask = MASKED && !inner && occ != 0ull;

    // ---- all-ones Toeplitz operands of the horizontal pass: B_p[k][n] = 1 for 0 <= 32 p + k - n < kn
    h8 ones_b0, ones_b1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t0 = 8 * g + e - n;
        ones_b0[e] = (t0 >= 0 && t0 < kn) ? (_Float16)1.0f : (_Float16)0.0f;
        ones_b1[e] = (t0 + 32 >= 0 && t0 + 32 < kn) ? (_Float16)1.0f : (_Float16)0.0f;
    }

    // ---- box sums.  Horizontal pass per 16-row block rb of the wave's 48 input rows; its accumulators (lane (n, g):
    //      rows 16 rb + 4 g + v of column n) go straight back in as the B operand of the vertical pass, slots e = 0 .. 3
    //      of k = 8 g + e labelled as row 16 rb + 4 g + e: A[m][8 g + e] = 1 for 0 <= 16 rb + 4 g + e - m < km
    f4 S1[4], S2[4], NM[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) S1[c] = S2[c] = NM[c] = zero4;
    const int n_rb = km > 17 ? 3 : 2;
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
        if (rb < n_rb) {
            f4 a1[4], a2[4], am_[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) a1[c] = a2[c] = am_[c] = zero4;
            const int rowoff = (wr0 + 16 * rb + n) * WD_P + 8 * g;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const h8 ah = *reinterpret_cast<const h8*>(xh + rowoff + 16 * cb);
                const h8 al = *reinterpret_cast<const h8*>(xl + rowoff + 16 * cb);
                // squares as float16 pairs in packed float16 arithmetic (2.5 instructions per pixel; through float32: 6):
                // x^2 / 32 = (xh^2 + 2 xh xl) / 32 up to xl^2 (2^-22 of it) -- qh the rounded product of the heads, then its EXACT
                // remainder by one fused multiply-add (the product of two 11-bit heads has 22 bits: head + remainder hold them
                // all), then the cross product on top of the remainder.  2^-5: a 33-sum of squares stays below 65504.
                const h8 ts = ah * (_Float16)0.03125f;
                const h8 qh = ts * ah;
                h8 ql = __builtin_elementwise_fma(ts, ah, -qh);
                ql = __builtin_elementwise_fma(ts + ts, al, ql);
                if (cb < 4) {
                    a1[cb] = mfma16(ah, ones_b0, a1[cb]);
                    a1[cb] = mfma16(al, ones_b0, a1[cb]);
                    a2[cb] = mfma16(qh, ones_b0, a2[cb]);
                    a2[cb] = mfma16(ql, ones_b0, a2[cb]);
                }
                if (TWO && cb >= 2) {
                    a1[cb - 2] = mfma16(ah, ones_b1, a1[cb - 2]);
                    a1[cb - 2] = mfma16(al, ones_b1, a1[cb - 2]);
                    a2[cb - 2] = mfma16(qh, ones_b1, a2[cb - 2]);
                    a2[cb - 2] = mfma16(ql, ones_b1, a2[cb - 2]);
                }
                if constexpr (MASKED) {
                    if (any_mask) {
                        const h8 am = *reinterpret_cast<const h8*>(xm + rowoff + 16 * cb);
                        if (cb < 4) am_[cb] = mfma16(am, ones_b0, am_[cb]);
                        if (TWO && cb >= 2) am_[cb - 2] = mfma16(am, ones_b1, am_[cb - 2]);
                    }
                }
            }
            h8 ones_a;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * rb + 4 * g + e - n;
                ones_a[e] = (e < 4 && d >= 0 && d < km) ? (_Float16)1.0f : (_Float16)0.0f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                h8 bh, bl;
                split_low(a1[c], bh, bl);
                S1[c] = mfma16(ones_a, bh, S1[c]);
                S1[c] = mfma16(ones_a, bl, S1[c]);
                split_low(a2[c], bh, bl);
                S2[c] = mfma16(ones_a, bh, S2[c]);
                S2[c] = mfma16(ones_a, bl, S2[c]);
                if constexpr (MASKED) {
                    if (any_mask) {
                        split_low(am_[c], bh, bl);          // (counts of up to 33: exact in the head)
                        NM[c] = mfma16(ones_a, bh, NM[c]);
                    }
                }
            }
        }
    }

    WD_STAMP(5);          // box sums
    // ---- cross term: per template row the wave's 6 (4) A blocks against the row's Toeplitz fragments, loaded from the
    //      image in global memory one row ahead
    f4 accM[4], accC[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accM[c] = accC[c] = zero4;
    {
        const uint4* F = E.frag + lane;                   // [set 0][s][pass][head | tail][lane]
        uint4 nf[4];
#pragma unroll
        for (int q = 0; q < (TWO ? 4 : 2); ++q) nf[q] = F[q * 64];
        for (int s = 0; s < km; ++s) {
            uint4 cf[4];
#pragma unroll
            for (int q = 0; q < (TWO ? 4 : 2); ++q) cf[q] = nf[q];
            {
                const uint4* Fn = F + (size_t)min(s + 1, km - 1) * 256;
#pragma unroll
                for (int q = 0; q < (TWO ? 4 : 2); ++q) nf[q] = Fn[q * 64];
            }
            const int rowoff = (wr0 + s + n) * WD_P + 8 * g;
            h8 ah[NCB], al[4];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) ah[cb] = *reinterpret_cast<const h8*>(xh + rowoff + 16 * cb);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) al[cb] = *reinterpret_cast<const h8*>(xl + rowoff + 16 * cb);
            const h8 bh0 = as_h8(cf[0]), bl0 = as_h8(cf[1]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                accM[c] = mfma16(ah[c], bh0, accM[c]);
                accC[c] = mfma16(ah[c], bl0, accC[c]);
                accC[c] = mfma16(al[c], bh0, accC[c]);
            }
            if constexpr (TWO) {
                // second pass: the weights of a row end at t = 32, so only k = 0 .. 15 of its Toeplitz operand are ever non-zero --
                // the other half of the contraction carries the TAILS of the same 16 columns against a second copy of the weight
                // heads (the fragment image repeats them at k = 16 .. 31, cs_api.cpp ensure_wfrag_wide): lanes of k groups 2 and 3
                // read the tail plane, and heads x heads + tails x heads is ONE MFMA (5 per template row and column tile, not 6)
                const h8 bh1 = as_h8(cf[2]), bl1 = as_h8(cf[3]);
                const _Float16* mixed = (g < 2 ? xh : xl) + (wr0 + s + n) * WD_P + 8 * (g & 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const h8 am2 = *reinterpret_cast<const h8*>(mixed + 16 * (c + 2));
                    accM[c] = mfma16(am2, bh1, accM[c]);
                    accC[c] = mfma16(ah[c + 2], bl1, accC[c]);
                }
            }
        }
    }

    WD_STAMP(6);          // cross term
    // ---- mask-weighted template sums: the plane against the Wa and the Wb fragments (the plane is exact in float16:
    //      two MFMAs per block and set); blocks whose 16 x 16 sub-blocks hold no flagged pixel are skipped
    f4 KA[4], KB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) KA[c] = KB[c] = zero4;
    if constexpr (MASKED) {
        if (any_mask) {
#pragma unroll
            for (int set = 1; set <= 2; ++set) {
                const uint4* F = E.frag + (size_t)set * km * 256 + lane;
                uint4 nf[4];
#pragma unroll
                for (int q = 0; q < (TWO ? 4 : 2); ++q) nf[q] = F[q * 64];
                for (int s = 0; s < km; ++s) {
                    uint4 cf[4];
#pragma unroll
                    for (int q = 0; q < (TWO ? 4 : 2); ++q) cf[q] = nf[q];
                    {
                        const uint4* Fn = F + (size_t)min(s + 1, km - 1) * 256;
#pragma unroll
                        for (int q = 0; q < (TWO ? 4 : 2); ++q) nf[q] = Fn[q * 64];
                    }
                    // occupied 16-column blocks among the (one or two) 16-row blocks the rows wr0 + s .. wr0 + s + 15 touch
                    const int r0 = wr0 + s;
                    const unsigned ro = (unsigned)(((occ >> (6 * (r0 >> 4))) | (occ >> (6 * ((r0 + 15) >> 4)))) & 0x3full);
                    if (ro == 0u) continue;
                    const int rowoff = (r0 + n) * WD_P + 8 * g;
                    const h8 wh0 = as_h8(cf[0]), wl0 = as_h8(cf[1]);
                    const h8 wh1 = as_h8(cf[TWO ? 2 : 0]), wl1 = as_h8(cf[TWO ? 3 : 1]);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (ro & (3u << cb)) {             // (block cb covers the 16-column blocks cb and cb + 1; block 6 is the zero pad)
                            const h8 am = *reinterpret_cast<const h8*>(xm + rowoff + 16 * cb);
                            if (cb < 4) {
                                if (set == 1) {
                                    KA[cb] = mfma16(am, wh0, KA[cb]);
                                    KA[cb] = mfma16(am, wl0, KA[cb]);
                                } else {
                                    KB[cb] = mfma16(am, wh0, KB[cb]);
                                    KB[cb] = mfma16(am, wl0, KB[cb]);
                                }
                            }
                            if (TWO && cb >= 2) {
                                if (set == 1) {
                                    KA[cb - 2] = mfma16(am, wh1, KA[cb - 2]);
                                    KA[cb - 2] = mfma16(am, wl1, KA[cb - 2]);
                                } else {
                                    KB[cb - 2] = mfma16(am, wh1, KB[cb - 2]);
                                    KB[cb - 2] = mfma16(am, wl1, KB[cb - 2]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if constexpr (MASKED) {
        if (any_mask) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                KA[c] *= E.unscale[1];
                KB[c] *= E.unscale[2];
            }
        }
        if (has_cross) {
            // flagged rows among the wave's input rows wr0 .. wr0 + 14 + km: row p of them lies in the windows of the wave's rows
            // p - s, 0 <= s < km, and takes sum_{t in C_j} W[s][t] (C_j: the flagged columns of column j's window) off the row +
            // column terms of those pixels.  Evaluated where it is needed (a wave meets one flagged row per tile on average, a
            // window half a flagged column): no table, no LDS -- what lets a launch of inner tiles run three workgroups per CU.
            unsigned long long wrows = rbits_lo >> wr0;
            if (wr0) wrows |= (unsigned long long)rbits_hi << (64 - wr0);
            wrows &= (1ull << (15 + km)) - 1ull;
            if (wrows) {
                unsigned long long wcol[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) wcol[c] = window_bits(cbits_lo, cbits_hi, 16 * c + n, kn);
                const int kk = km * kn;
                for (; wrows; wrows &= wrows - 1ull) {
                    const int pr = __builtin_ctzll(wrows);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int sr = pr - (4 * g + v);
                        if ((unsigned)sr < (unsigned)km) {
                            const float* wa = A.w + kk + sr * kn;
                            const float* wb = wa + kk;
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                for (unsigned long long w = wcol[c]; w; w &= w - 1ull) {
                                    const int t = __builtin_ctzll(w);
                                    KA[c][v] -= wa[t];
                                    KB[c][v] -= wb[t];
                                }
                        }
                    }
                }
            }
        }
    }
    WD_STAMP(7);          // mask sums
    // ---- epilogue: lane = column n of tile c, rows 4 g + v
    const float u_cs = unscale * E.unscale[0];
    const float u_s2 = 32.0f * unscale;
    const float kn_f = (float)kn, km_f = (float)km;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
    // a tile all of whose 64 x 64 pixels are produced and none forced to zero (cs_device.h pixel_forced_zero): no per-pixel
    // predicates, addresses by increments
    const int dmin_t = J0 - (I0 + WD_T - 1), dmax_t = J0 + WD_T - 1 - I0;
    const bool same_nobs = !A.nobs.ptr || (A.nobs.layout == A.out.layout && A.nobs.ld == A.out.ld && A.nobs.band_lo == A.out.band_lo &&
                                           A.nobs.band_w == A.out.band_w && A.nobs.row0 == A.out.row0);
    const bool plain = I0 + WD_T <= A.row_end && J0 >= 0 && J0 + WD_T <= A.ns && dmin_t >= A.out_lo && dmax_t <= A.out_hi && same_nobs &&
                       (A.full || (I0 >= kh && I0 + WD_T - 1 <= A.ms - km + kh && J0 >= kw && J0 + WD_T - 1 <= A.ns - kn + kw)) &&
                       (!A.sym_upper || dmin_t + (A.full ? kn - km : 0) >= 0) &&
                       (!band_out || (dmin_t >= A.out.band_lo && dmax_t < A.out.band_lo + A.out.band_w));
    const int i_lane = I0 + wr0 + 4 * g, j_lane = J0 + n;
    const long long o_lane = ((long long)i_lane - A.out.row0) * A.out.ld + (band_out ? j_lane - i_lane - A.out.band_lo : j_lane);
    const long long o_row = band_out ? A.out.ld - 1 : A.out.ld;
    // the window sums of pixel (c, v) -> coefficient (cs_device.h: the one-rsq form, the literal function next to a zeroing threshold)
    auto coefficient = [&](int c, int v, float& nobs) -> float {
        const int il = wr0 + 4 * g + v, jl = 16 * c + n;
        const float cs = (accM[c][v] + accC[c][v]) * u_cs;
        const float s1 = S1[c][v] * unscale;
        const float s2 = (S2[c][v] * u_s2) * unscale;
        nobs = A.ks.n;
        float r;
        if (A.xcorr_only) {
            r = (fabsf(cs) < A.ks.thr) ? 0.0f : cs;
        } else if constexpr (MASKED) {
            // missing pixels of the window: nr rows x kn + nc columns x km - nr nc on both (exact small integers), or the plane's
            const float nr = tab[il], nc = tab[3 * 64 + jl];
            const float nm = nr * kn_f + nc * (km_f - nr) + NM[c][v];
            const float ka = tab[64 + il] + tab[4 * 64 + jl] + KA[c][v];
            const float kb = tab[128 + il] + tab[5 * 64 + jl] + KB[c][v];
            r = wide_range_guard(pearson_masked_lean(cs, s1, s2, nm, ka, kb, A.ks), s2, unscale, A.ks);
            nobs = A.ks.n - nm;
        } else {
            r = wide_range_guard(pearson_nomask_lean(cs, s1, s2, A.ks), s2, unscale, A.ks);
        }
        return r;
    };
    // candidate sink (cs_device.h CorrArgs::cand_keys: cs_detect_foci / cs_candidates): no map leaves the kernel, only the keys
    // tag + row * ns + col of the pixels that carry a candidate value (>= cand_thr: the screen's sentinel included), appended to
    // the caller's list -- 1e-4 of the pixels, one atomic each; the counter runs on beyond the capacity so that the caller learns
    // how much room a second call needs
    const bool sinking = A.cand_keys != nullptr;
    auto sink = [&](int i, int j, float r) {
        if (r >= A.ks.cand_thr) {
            const unsigned long long pos = atomicAdd(A.cand_count, 1ull);
            if (pos < (unsigned long long)A.cand_cap)
                A.cand_keys[pos] = A.cand_tag + (unsigned long long)i * (unsigned long long)A.ns + (unsigned long long)j;
        }
    };
    if (plain) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float nobs;
                const float r = coefficient(c, v, nobs);
                if (sinking) {
                    sink(i_lane + v, j_lane + 16 * c, r);
                    continue;
                }
                const long long o = o_lane + v * o_row + 16 * c;
                if (A.out_is_f64) reinterpret_cast<double*>(A.out.ptr)[o] = (double)r;
                else reinterpret_cast<float*>(A.out.ptr)[o] = r;
                if (A.nobs.ptr) reinterpret_cast<float*>(A.nobs.ptr)[o] = nobs;
            }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!((cmask >> c) & 1u)) continue;
            const int j = J0 + 16 * c + n;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = I0 + wr0 + 4 * g + v;
                if (i >= A.row_end || j < 0 || j >= A.ns) continue;
                const int d = j - i;
                if (d < A.out_lo || d > A.out_hi) continue;
                float nobs;
                const float r = coefficient(c, v, nobs);
                if (sinking) {
                    if (!pixel_forced_zero(A, i, j)) sink(i, j, r);
                    continue;
                }
                store_pixel(A, i, j, pixel_forced_zero(A, i, j) ? 0.0f : r, nobs);
            }
        }
    }
    }      // cmask
    WD_STAMP(8);          // epilogue
#ifdef CS_WD_PROFILE
    if (lane == 0) {
        unsigned long long* dst = cs_wd_prof + (blockIdx.x & 63) * 16;
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(dst + k, tdelta_[k]);
        atomicAdd(dst + 15, 1ull);
    }
#endif
    }
}

#ifdef CS_WD_PROFILE
extern "C" int cs_debug_wide_profile(unsigned long long* out)
{
    static unsigned long long all[64 * 16];
    hipError_t e = hipMemcpyFromSymbol(all, HIP_SYMBOL(cs_wd_prof), sizeof(all));
    if (e != hipSuccess) return (int)e;
    for (int k = 0; k < 16; ++k) {
        out[k] = 0;
        for (int c = 0; c < 64; ++c) out[k] += all[c * 16 + k];
    }
    for (auto& v : all) v = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(cs_wd_prof), all, sizeof(all));
}
#endif

// The 160 KB dynamic-LDS ceiling is a per-function, per-device attribute: set it the first time a kernel is
// launched on a device.
static hipError_t wide_allow_big_lds(const void* fn)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& d : done)
        if (d.first == fn && d.second == dev) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.emplace_back(fn, dev);
    return e;
}

bool corr_mfma_wide_fits(int km, int kn) { return km >= 1 && kn >= 1 && km <= 33 && kn <= 33; }

int launch_corr_mfma_wide_f32(CorrArgs<float>& A, const MfmaWideWeights& E, hipStream_t stream)
{
    if (!corr_mfma_wide_fits(A.km, A.kn)) return -3;
    if (A.sig.counts || A.sig.layout == 2) return -6;      // (bands of counts / lazily evaluated bands: other readers)
    if (!A.out.ptr && !(A.cand_keys && A.cand_count && A.ks.cand_cmin > 0.0f)) return -5;      // a map, or a candidate sink
    if (A.defer_args) return -5;                           // (argument tables of the multi-block launch: the 17 x 17 tile kernel only)
    A.tile_w = A.tile_h = WD_T;
    A.tiles_y = (A.row_end - A.row_begin + WD_T - 1) / WD_T;
    if (A.out.layout == 1) {
        A.out_lo = A.out.band_lo;
        A.out_hi = A.out.band_lo + A.out.band_w - 1;
        A.tiles_x = (A.out.band_w + WD_T - 1 + WD_T - 1) / WD_T;
    } else {
        A.out_lo = -(1 << 30);
        A.out_hi = (1 << 30);
        A.tiles_x = (A.ns + WD_T - 1) / WD_T;
    }
    if (A.cand_keys && !A.out.ptr) {
        // candidate sink: only the scanned diagonals (the map path trims in the compaction)
        A.out_lo = std::max(A.out_lo, A.cand_dlo);
        A.out_hi = std::min(A.out_hi, A.cand_dhi);
    } else {
        A.cand_keys = nullptr;                             // (a map was asked for: the kernel writes it)
    }
    const long long blocks = (long long)A.tiles_x * A.tiles_y;
    if (blocks <= 0) return 0;
    if (blocks > 0x7ffffff0LL) return -3;
    const bool masked = A.mask_mode != 0;
    const bool two = A.kn > 17;
    typedef void (*kern_t)(const CorrArgs<float>, const MfmaWideWeights);
    const kern_t k = masked ? (two ? corr_mfma_wide_kernel<true, true> : corr_mfma_wide_kernel<true, false>)
                            : (two ? corr_mfma_wide_kernel<false, true> : corr_mfma_wide_kernel<false, false>);
    hipError_t e = wide_allow_big_lds((const void*)k);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)((blocks + 7) / 8 * 8);
    // (a narrow band has few inner tiles and two short launches cost more than the third workgroup per CU buys: 234 diagonals
    // 0.426 ms in one launch, 0.470 in two; 1001 diagonals 3.74 against 3.38)
    if (masked && A.mask_mode == 1 && !E.plane_only && !E.plane_only_staging && E.one_launch != 1 && ((A.tiles_x >= 8 && blocks >= 2048) || E.one_launch == 2)) {
        // per-bin masks: the inner tiles (no plane: 45 KB of LDS, three workgroups per CU) in a launch of their own, then the
        // tiles on the rim of the band / the frame of the matrix with the plane (66 KB, two per CU); a tile that is not the
        // launch's returns at once.  C4' 21 x 21: 3.74 -> see profiles/r06_template_kernels.txt
        MfmaWideWeights E1 = E, E2 = E;
        E1.tile_mode = 1;
        E2.tile_mode = 2;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), WD_SMEM_INNER, stream, A, E1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), WD_SMEM_MASKED, stream, A, E2);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), masked ? WD_SMEM_MASKED : WD_SMEM_PLAIN, stream, A, E);
    return (int)hipGetLastError();
}

}  // namespace cs
