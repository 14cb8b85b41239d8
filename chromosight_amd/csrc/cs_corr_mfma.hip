// cs_corr_mfma.hip -- sliding-window correlation on the matrix cores (gfx950 v_mfma_f32_16x16x32_f16).
//
// The float32 class of cs_normxcorr2 / cs_xcorr2 for templates of up to 17 x 17 (any km, kn <= 17).
// The packed-FMA streaming kernel (cs_corr_stream.h) is bound by instruction issue, not by the VALU
// (DESIGN.md 7); one MFMA does the work of 128 v_pk_fma_f32 for one issue slot, so the window sums
// are recast as small GEMMs:
//
//   * cross term  sum_{s,t} W[s][t] x[i+s][j+t]: for every template row s, a 16-row x 32-column block
//     of the staged signal (A operand) times the 32 x 16 Toeplitz matrix of W[s][.] (B operand,
//     B[k][n] = W[s][k - n]) gives the 16 x 16 output tile's contribution of that row; the 17 rows
//     accumulate in the MFMA accumulator.  17 of the 32 k are useful (53 %).
//   * float32 accuracy from float16 operands: x and W are split into a float16 head and tail
//     (x = xh + xl exactly to 22 bits after a power-of-two scale that puts the tile's largest |x| in
//     [64, 128)); xh*Wh + xh*Wl + xl*Wh with exact products and float32 accumulation differs from the
//     float32 product sum by 2^-22 relative -- three MFMAs per (template row, tile), 2.5 PFLOP/s / 3 is
//     still 5x the 157 TFLOP/s of the FP32 vector pipe.
//   * box sums (sum x, sum x^2, number of missing pixels) are separable: a horizontal pass with the
//     all-ones Toeplitz matrix over the wave's 32 input rows, the result split again and transposed
//     through LDS, then a vertical pass with the all-ones Toeplitz matrix as the A operand.  Both land
//     in the accumulator layout of the cross term (lane = column, 4 consecutive rows per lane).
//   * missing masks (per-bin flags or an explicit map, frame rules of preprocessing.py:404-498
//     included) are staged as a 0/1 plane; sum_missing Wa / Wb are two more correlations of that plane
//     (exact operand, two MFMAs per row and tile each).  No correction tables.
//
// One workgroup (4 waves) = one 64 x 64 output tile; wave w owns rows 16 w .. 16 w + 15 and four
// 16-column tiles.  LDS: 80 x 80 staged pixels as three float16 planes (head, tail, mask) + one weight
// set as ready-made B fragments (17 rows x {head, tail} x 1 KiB, built on the host, cs_api.cpp) =
// 73 KB, two workgroups per CU.  The epilogue is that of the other kernels (cs_device.h).
#include "cs_device.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "cs_launch.h"

namespace cs {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float cand_range_guard(float r, float s2, float unscale, const KernelStats<float>& K);
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int MF_T = 64;                      // output tile edge
constexpr int MF_R = 80;                      // staged rows / columns (tile + 16)
constexpr int MF_PLANE = MF_R * MF_R * 2;     // bytes of one float16 plane
constexpr int MF_WSET = 17 * 2 * 1024;        // bytes of one weight set's fragments
constexpr int MF_SCR_PITCH = 40;              // halfs per column of the transposed scratch (32 rows + pad)
constexpr int MF_SCR_PLANE = 16 * MF_SCR_PITCH * 2;
constexpr int MF_SMEM = 3 * MF_PLANE + MF_WSET + 64;
constexpr int MF_PER_THREAD = (MF_R * MF_R) / 256;   // 25 staged pixels per thread

static_assert(5 * MF_SCR_PLANE * 4 <= MF_WSET, "scratch aliases the weight region");

#ifdef CS_MF_PROFILE
__device__ unsigned long long cs_mf_prof[16];
#define MF_STAMP(k)                                                        \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        if (tid == 0) atomicAdd(&cs_mf_prof[k], now_ - tprev_);            \
        tprev_ = now_;                                                     \
    } while (0)
#else
#define MF_STAMP(k)
#endif

__device__ __forceinline__ f4 mfma16(const h8& a, const h8& b, const f4& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <bool MASKED>
__global__ __launch_bounds__(256, 2) void corr_mfma_kernel(const CorrArgs<float> A, const MfmaWeights E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* xh = reinterpret_cast<_Float16*>(smem);
    _Float16* xl = reinterpret_cast<_Float16*>(smem + MF_PLANE);
    _Float16* xm = reinterpret_cast<_Float16*>(smem + 2 * MF_PLANE);
    char* wreg = smem + 3 * MF_PLANE;
    unsigned* red = reinterpret_cast<unsigned*>(wreg + MF_WSET);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---- tile of this workgroup
    const int by = blockIdx.x / A.tiles_x;
    const int bx = blockIdx.x - by * A.tiles_x;
    const int I0 = A.row_begin + by * MF_T;
    if (I0 >= A.row_end) return;
    const bool band_out = A.out.layout == 1;
    const int J0 = band_out ? I0 + A.out_lo + bx * MF_T : bx * MF_T;
    if (J0 >= A.ns || J0 + MF_T <= 0) return;
    if (J0 + MF_T - 1 - I0 < A.out_lo || J0 - (I0 + MF_T - 1) > A.out_hi) return;   // no produced diagonal
#ifdef CS_MF_PROFILE
    unsigned long long tprev_ = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(&cs_mf_prof[15], 1ull);
#endif
    const int km = A.km, kn = A.kn;
    const int P0 = I0 - (km - 1) / 2, Q0 = J0 - (kn - 1) / 2;
    // rows no window of the row range [row_begin, row_end) reaches are not part of the input contract
    // (the signal buffer may be a slab that ends there)
    const int p_lo = A.row_begin - (km - 1) / 2, p_hi = A.row_end + (km - 1) - (km - 1) / 2;

    // ---- stage 80 x 80 pixels: scale by a power of two, split into float16 head / tail, mask plane.
    // All loads are unconditional from clamped addresses (a load inside a bounds branch is waited for
    // before the next one is issued: 25 serialised L2 round trips per thread).
    float xv[MF_PER_THREAD];
    unsigned miss_bits = 0, ok_bits = 0;
    long long offs[MF_PER_THREAD];
#pragma unroll
    for (int k = 0; k < MF_PER_THREAD; ++k) {
        const int idx = tid + 256 * k;
        const int r = idx / MF_R, c = idx - r * MF_R;
        const int p = P0 + r, q = Q0 + c;
        const bool inside = (p >= 0) & (p < A.ms) & (q >= 0) & (q < A.ns) & (p >= p_lo) & (p < p_hi);
        const long long off = inside ? mat_offset(A.sig, p, q) : -1;
        if (off >= 0) ok_bits |= 1u << k;
        offs[k] = off >= 0 ? off : 0;     // element 0 of the buffer always exists
    }
    if (A.sig_is_f64) {
        const double* src = reinterpret_cast<const double*>(A.sig.ptr);
#pragma unroll
        for (int k = 0; k < MF_PER_THREAD; ++k) xv[k] = (float)src[offs[k]];
    } else {
        const float* src = reinterpret_cast<const float*>(A.sig.ptr);
#pragma unroll
        for (int k = 0; k < MF_PER_THREAD; ++k) xv[k] = src[offs[k]];
    }
    if constexpr (MASKED) {
        unsigned char fr[MF_PER_THREAD], fc[MF_PER_THREAD];
        if (A.mask_mode == 1) {
#pragma unroll
            for (int k = 0; k < MF_PER_THREAD; ++k) {
                const int idx = tid + 256 * k;
                const int r = idx / MF_R, c = idx - r * MF_R;
                fr[k] = A.miss_row[min(max(P0 + r, 0), A.ms - 1)];
                fc[k] = A.miss_col[min(max(Q0 + c, 0), A.ns - 1)];
            }
        } else {
            const unsigned char* mp = reinterpret_cast<const unsigned char*>(A.mask.ptr);
#pragma unroll
            for (int k = 0; k < MF_PER_THREAD; ++k) {
                fr[k] = mp[offs[k]];
                fc[k] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < MF_PER_THREAD; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx / MF_R, c = idx - r * MF_R;
            const int p = P0 + r, q = Q0 + c;
            const bool stored = (ok_bits >> k) & 1u;
            const bool needed = (p >= p_lo) & (p < p_hi);
            if (needed && missing_from_flags(A, p, q, fr[k] != 0, fc[k] != 0, fr[k] != 0, stored)) miss_bits |= 1u << k;
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int k = 0; k < MF_PER_THREAD; ++k) {
        // the reference requires 0 at missing pixels (check_missing_mask); enforce it
        const float x = (((ok_bits & ~miss_bits) >> k) & 1u) ? xv[k] : 0.0f;
        xv[k] = x;
        amax = fmaxf(amax, fabsf(x));
    }
    MF_STAMP(0);      // global loads of the staged pixels
    if (tid == 0) *red = 0u;
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) atomicMax(red, __float_as_uint(amax));
    __syncthreads();
    int ex = 0;
    {
        const int e = (int)((*red >> 23) & 0xffu);
        if (e != 0 && e != 255) ex = 6 - (e - 127);
        ex = max(-100, min(100, ex));
    }
    const float scale = __uint_as_float((unsigned)(ex + 127) << 23);
    const float unscale = __uint_as_float((unsigned)(127 - ex) << 23);
#pragma unroll
    for (int k = 0; k < MF_PER_THREAD; ++k) {
        const int idx = tid + 256 * k;
        const float xs = xv[k] * scale;
        const _Float16 h = (_Float16)xs;
        xh[idx] = h;
        xl[idx] = (_Float16)(xs - (float)h);
        if constexpr (MASKED) xm[idx] = (miss_bits >> k) & 1u ? (_Float16)1.0f : (_Float16)0.0f;
    }
    __syncthreads();

    MF_STAMP(1);      // scale, split, LDS planes
    // ---- all-ones Toeplitz operands: B[k][n] = 1 for 0 <= k - n < kn, A[m][k] = 1 for 0 <= k - m < km
    h8 ones_b, ones_a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = 8 * g + e - n;
        ones_b[e] = (t >= 0 && t < kn) ? (_Float16)1.0f : (_Float16)0.0f;
        ones_a[e] = (t >= 0 && t < km) ? (_Float16)1.0f : (_Float16)0.0f;
    }
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int wr0 = 16 * wv;                  // first staged row of the wave's windows

    // ---- box sums: horizontal pass over the wave's 32 input rows, transpose through LDS, vertical pass
    f4 S1[4], S2[4], NM[4];
    char* scr = wreg + wv * (5 * MF_SCR_PLANE);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int off = (wr0 + 16 * rb + n) * MF_R + 16 * c + 8 * g;
            const h8 ah = *reinterpret_cast<const h8*>(xh + off);
            const h8 al = *reinterpret_cast<const h8*>(xl + off);
            f4 h1 = mfma16(ah, ones_b, zero4);
            h1 = mfma16(al, ones_b, h1);
            h8 qh, ql;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (float)ah[e] + (float)al[e];
                const float q = x * x * 0.03125f;            // 2^-5: the 17-sum of squares stays below 65504
                qh[e] = (_Float16)q;
                ql[e] = (_Float16)(q - (float)qh[e]);
            }
            f4 h2 = mfma16(qh, ones_b, zero4);
            h2 = mfma16(ql, ones_b, h2);
            // accumulator layout: h[v] = H[row 16 rb + 4 g + v][col n] -> transposed scratch [col][row]
            h4 t1h, t1l, t2h, t2l;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                t1h[v] = (_Float16)h1[v];
                t1l[v] = (_Float16)(h1[v] - (float)t1h[v]);
                t2h[v] = (_Float16)h2[v];
                t2l[v] = (_Float16)(h2[v] - (float)t2h[v]);
            }
            const int so = (n * MF_SCR_PITCH + 16 * rb + 4 * g) * 2;
            *reinterpret_cast<h4*>(scr + 0 * MF_SCR_PLANE + so) = t1h;
            *reinterpret_cast<h4*>(scr + 1 * MF_SCR_PLANE + so) = t1l;
            *reinterpret_cast<h4*>(scr + 2 * MF_SCR_PLANE + so) = t2h;
            *reinterpret_cast<h4*>(scr + 3 * MF_SCR_PLANE + so) = t2l;
            if constexpr (MASKED) {
                const h8 am = *reinterpret_cast<const h8*>(xm + off);
                const f4 hm = mfma16(am, ones_b, zero4);
                h4 tm;
#pragma unroll
                for (int v = 0; v < 4; ++v) tm[v] = (_Float16)hm[v];
                *reinterpret_cast<h4*>(scr + 4 * MF_SCR_PLANE + so) = tm;
            }
        }
        __syncthreads();
        {
            const int ro = (n * MF_SCR_PITCH + 8 * g) * 2;
            const h8 b1h = *reinterpret_cast<const h8*>(scr + 0 * MF_SCR_PLANE + ro);
            const h8 b1l = *reinterpret_cast<const h8*>(scr + 1 * MF_SCR_PLANE + ro);
            const h8 b2h = *reinterpret_cast<const h8*>(scr + 2 * MF_SCR_PLANE + ro);
            const h8 b2l = *reinterpret_cast<const h8*>(scr + 3 * MF_SCR_PLANE + ro);
            S1[c] = mfma16(ones_a, b1h, zero4);
            S1[c] = mfma16(ones_a, b1l, S1[c]);
            S2[c] = mfma16(ones_a, b2h, zero4);
            S2[c] = mfma16(ones_a, b2l, S2[c]);
            NM[c] = zero4;
            if constexpr (MASKED) {
                const h8 bm = *reinterpret_cast<const h8*>(scr + 4 * MF_SCR_PLANE + ro);
                NM[c] = mfma16(ones_a, bm, zero4);
            }
        }
        __syncthreads();
    }

    MF_STAMP(2);      // box sums
    // ---- correlation passes: one weight set at a time in LDS
    auto load_set = [&](int set) {
        const uint4* src = E.frag + (size_t)set * km * 128;
        uint4* dst = reinterpret_cast<uint4*>(wreg);
        for (int idx = tid; idx < km * 128; idx += 256) dst[idx] = src[idx];
    };
    f4 accM[4], accC[4], KA[4], KB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accM[c] = accC[c] = KA[c] = KB[c] = zero4;

    load_set(0);
    __syncthreads();
    MF_STAMP(3);      // weight fragments -> LDS
    for (int s = 0; s < km; ++s) {
        const h8 bh = reinterpret_cast<const h8*>(wreg)[(2 * s + 0) * 64 + lane];
        const h8 bl = reinterpret_cast<const h8*>(wreg)[(2 * s + 1) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int off = (wr0 + s + n) * MF_R + 16 * c + 8 * g;
            const h8 ah = *reinterpret_cast<const h8*>(xh + off);
            const h8 al = *reinterpret_cast<const h8*>(xl + off);
            accM[c] = mfma16(ah, bh, accM[c]);
            accC[c] = mfma16(ah, bl, accC[c]);
            accC[c] = mfma16(al, bh, accC[c]);
        }
    }
    MF_STAMP(4);      // cross term
    if constexpr (MASKED) {
        for (int set = 1; set <= 2; ++set) {
            __syncthreads();
            load_set(set);
            __syncthreads();
            for (int s = 0; s < km; ++s) {
                const h8 bh = reinterpret_cast<const h8*>(wreg)[(2 * s + 0) * 64 + lane];
                const h8 bl = reinterpret_cast<const h8*>(wreg)[(2 * s + 1) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int off = (wr0 + s + n) * MF_R + 16 * c + 8 * g;
                    const h8 am = *reinterpret_cast<const h8*>(xm + off);
                    if (set == 1) {
                        KA[c] = mfma16(am, bh, KA[c]);
                        KA[c] = mfma16(am, bl, KA[c]);
                    } else {
                        KB[c] = mfma16(am, bh, KB[c]);
                        KB[c] = mfma16(am, bl, KB[c]);
                    }
                }
            }
        }
    }

    MF_STAMP(5);      // mask-weighted template sums
    // ---- epilogue: lane = column n of tile c, rows 4 g + v
    const float u_cs = unscale * E.unscale[0];
    const float u_s2 = 32.0f * unscale;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = J0 + 16 * c + n;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = I0 + wr0 + 4 * g + v;
            if (i >= A.row_end || j < 0 || j >= A.ns) continue;
            const int d = j - i;
            if (d < A.out_lo || d > A.out_hi) continue;
            const float cs = (accM[c][v] + accC[c][v]) * u_cs;
            const float s1 = S1[c][v] * unscale;
            const float s2 = (S2[c][v] * u_s2) * unscale;
            float r, nobs = A.ks.n;
            if (pixel_forced_zero(A, i, j)) {
                r = 0.0f;
            } else if (A.xcorr_only) {
                r = (fabsf(cs) < A.ks.thr) ? 0.0f : cs;
            } else if constexpr (MASKED) {
                const float nm = NM[c][v];
                const float ka = KA[c][v] * E.unscale[1], kb = KB[c][v] * E.unscale[2];
                r = cand_range_guard(pearson_masked_lean(cs, s1, s2, nm, ka, kb, A.ks), s2, unscale, A.ks);
                nobs = A.ks.n - nm;
            } else {
                r = cand_range_guard(pearson_nomask_lean(cs, s1, s2, A.ks), s2, unscale, A.ks);
            }
            store_pixel(A, i, j, r, nobs);
        }
    }
    MF_STAMP(6);      // epilogue
}

#ifdef CS_MF_PROFILE
extern "C" int cs_debug_mfma_profile(unsigned long long* out)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(cs_mf_prof), sizeof(cs_mf_prof));
    if (e != hipSuccess) return (int)e;
    unsigned long long zero[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(cs_mf_prof), zero, sizeof(zero));
}
#endif

// ------------------------------------------------------------------------------------------------
// Unmasked dense float32 maps (the headline configuration): persistent workgroups; the next tile is
// fetched by LDS-DMA while the current one is in the MFMA phase; weight heads in registers, tails in
// LDS; squares staged next to the signal; no workgroup barrier between the box sums and the stores.
// ------------------------------------------------------------------------------------------------
// Correction records of cs_mask_prep.hip (pixels whose window leaves the matrix or the diagonals 0 .. max_dist).
// Only tiles on the rim of the band need them: the kernel reads this part of its arguments through a pointer the
// compiler cannot see through (fix_args), so the 20 scalar registers load inside that branch instead of being
// held -- spilled -- across the whole tile loop.
struct MfmaFixArgs {
    int fix_on, fix_hi_w, fix_hi_d0;
    const float* fix_lo;
    const float* fix_hi;
    const float* fix_rows;
    const float* fix_cols;
    int fix_top, fix_bot0, fix_width, fix_xband, fix_xlo, fix_side;
};

struct MfmaDenseArgs {
    const float* sig;
    void* out;               // float32, or float64 when out_is_f64 (the Python surface returns float64 maps)
    int out_is_f64;
    long long ld_in, ld_out, row0_in, row0_out;
    int ms, ns, km, kn;
    int row_begin, row_end;
    int full, sym_upper;
    int xcorr_only;          // plain cross-correlation: out = thresholded sum S*w (no box sums, no normalisation)
    int tiles_x, n_tiles;
    const uint4* frag;
    float w_unscale;
    float wa_unscale, wb_unscale;    // REG: power-of-two scales of the Wa / Wb sets (float16 range)
    int xcd_order;           // 1: contiguous tile ranges per XCD
    int dbg;                 // diagnostics (CHROMOSIGHT_HIP_MFMA_DBG): skip 1 stores, 2 prefetch, 4 box sums, 8 cross term
    KernelStats<float> ks;
    // ---- REG instances: per-bin missing masks through the factorised tables of cs_mask_prep.hip, any
    //      layout (band / dense) on either side, optional n_obs map
    int band_in, lo_in, bw_in;       // input band: first stored diagonal, stored diagonals
    int band_out, lo_out;            // output band: first stored diagonal
    int out_lo, out_hi;              // produced diagonals
    float* nobs;                     // same geometry as `out`, or nullptr
    const float* w;                  // float32 weight sets (K*K each): centred template, Wa, Wb
    const uint8_t* miss_col;
    const float* rowtab;             // [row][4]: nr, RA, RB, flags of the window rows
    const float* coltab;             // [3][ns]: ncol, CA, CB
    int fix_on, fix_hi_w, fix_hi_d0, fix_top, fix_bot0, fix_any_side;     // what decides whether a tile has records
    MfmaFixArgs fx;          // read through fix_args() only
};

constexpr int MFD_ROWS_PER_THREAD = 14;       // staging: 240 threads = 40 column pairs x 6 row groups
constexpr int MFD_WL = 4 * MF_PLANE;          // weight tails
constexpr int MFD_SCR = MFD_WL + 17 * 1024;   // per wave: head / tail plane of the transposed sums
constexpr int MFD_RED = MFD_SCR + 4 * 2 * MF_SCR_PLANE;
constexpr int MFD_SMEM = MFD_RED + 64;
// REG: column terms (ncol, CA, CB of the tile's 64 columns) and the column flags of its 80 staged columns,
// double-buffered (a tile's epilogue runs while the next tile's are already landing)
constexpr int MFD_COL = MFD_RED + 64;
constexpr int MFD_CFL = MFD_COL + 2 * 3 * 64 * 4;
constexpr int MFD_CFB = MFD_CFL + 2 * 16;              // (CFL: the 80 column flags as 3 mask words per slot) raw flag bytes as they arrive
constexpr int MFD_SMEM_REG = MFD_CFB + 2 * 128;
static_assert(MFD_SMEM_REG <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v;
    v[0] = a;
    v[1] = b;
    return __builtin_bit_cast(unsigned, v);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): with
// global stores or LDS-DMA transfers in flight it would stall every wave for a full memory round trip.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// maximum of a non-negative value over the wave, valid in lane 63 (DPP: no LDS round trips)
__device__ __forceinline__ float wave_max_nonneg(float v)
{
#define CS_DPP_MAX(ctrl, rmask)                                                                                  \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true)))
    CS_DPP_MAX(0x111, 0xf);      // row_shr:1
    CS_DPP_MAX(0x112, 0xf);      // row_shr:2
    CS_DPP_MAX(0x114, 0xf);      // row_shr:4
    CS_DPP_MAX(0x118, 0xf);      // row_shr:8   -> lane 15 of every row holds the row's maximum
    CS_DPP_MAX(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    CS_DPP_MAX(0x143, 0xc);      // row_bcast:31 into rows 2 and 3
#undef CS_DPP_MAX
    return v;
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave execute in order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Candidate mode (cs_device.h: cand_screen_*): the float16 pairs carry 21-22 bits of every pixel only down to 2^-10 of
// the tile's largest value (the tails leave the normal float16 range below that) and the squares down to 2^-6 of it,
// so a window whose mean square is below 2^-18 of the largest square of its tile is outside the error model of the
// bound and takes the sentinel.  `unscale` = 2^-ex with the tile's largest |x| in [64, 128) / 2^ex.
__device__ __forceinline__ float cand_range_guard(float r, float s2, float unscale, const KernelStats<float>& K)
{
    if (K.cand_cmin > 0.0f) r = ((int)(s2 > 0.0f) & (int)(s2 < K.n * (unscale * unscale) * 0.0625f)) ? 2.0f : r;
    return r;
}

// Out-of-line pieces of the REG epilogue: 16 inlined copies of each made the kernel 100 KB of code.
__device__ __attribute__((noinline)) float masked_coefficient_rare(float cs, float s1, float s2, float nm, float ka, float kb,
                                                                   const KernelStats<float>& K)
{
    return pearson_masked_f32(cs, s1, s2, nm, ka, kb, K);
}

__device__ __forceinline__ float masked_coefficient(float cs, float s1, float s2, float nm, float ka, float kb,
                                                    const KernelStats<float>& K)
{
    bool rare;
    float r = pearson_masked_core(cs, s1, s2, nm, ka, kb, K, rare);
    if (rare) r = masked_coefficient_rare(cs, s1, s2, nm, ka, kb, K);
    return r;
}

// correction {d n_missing, d ka, d kb} of a pixel whose window leaves the matrix or the diagonals 0..max_dist
// (tables of cs_mask_prep.hip; same selection order as cs_corr_stream.h fix_fetch), or nullptr.  Select
// chains instead of nested branches: 16 copies of this are inlined into the epilogue.
typedef const __attribute__((address_space(4))) MfmaFixArgs* FixArgsPtr;

__device__ __forceinline__ FixArgsPtr fix_args()
{
    unsigned long long kv = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kv));                  // opaque: loads through it are neither hoisted nor speculated
    return (FixArgsPtr)(kv + offsetof(MfmaDenseArgs, fx));
}

// the epilogue's statistics block through the same kind of pointer: 16 scalar words that only the emit phase reads
typedef const __attribute__((address_space(4))) KernelStats<float>* StatsPtr;

__device__ __forceinline__ StatsPtr stats_args()
{
    unsigned long long kv = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kv));
    return (StatsPtr)(kv + offsetof(MfmaDenseArgs, ks));
}

__device__ __forceinline__ const float* mask_fix_record(const MfmaDenseArgs& A, FixArgsPtr F, int i, int j)
{
    const int K = A.km, KH = (A.km - 1) / 2;
    const int d = j - i;
    const int fix_on = F->fix_on, fix_side = F->fix_side, fix_top = F->fix_top, fix_bot0 = F->fix_bot0, fix_hi_d0 = F->fix_hi_d0,
              fix_hi_w = F->fix_hi_w;
    const float* fix_cols = F->fix_cols;
    const bool in_range = (j >= 0) & (j < A.ns) & (d >= A.out_lo) & (d <= A.out_hi) & (i < A.row_end);
    const int x = F->fix_xband ? d - F->fix_xlo : j;
    const bool top = i < fix_top;
    const bool bot = (i >= fix_bot0) & (!fix_on | (i + KH >= A.ms) | (j + KH >= A.ns));
    const bool side = (fix_cols != nullptr) & ((j < fix_side) | (j >= A.ns - fix_side));
    const bool lo = (fix_on != 0) & (d >= 0) & (d < K - 1);
    const bool hi = (fix_on != 0) & (d >= fix_hi_d0) & (d - fix_hi_d0 < fix_hi_w);
    const long long o_rows = (long long)(top ? i : fix_top + i - fix_bot0) * F->fix_width + x;
    const long long o_side = (long long)i * 2 * fix_side + (j < fix_side ? j : j - (A.ns - 2 * fix_side));
    const long long o_lo = (long long)i * (K - 1) + d;
    const long long o_hi = (long long)i * fix_hi_w + (d - fix_hi_d0);
    const float* base = (top | bot) ? F->fix_rows : side ? fix_cols : lo ? F->fix_lo : F->fix_hi;
    const long long off = (top | bot) ? o_rows : side ? o_side : lo ? o_lo : o_hi;
    return (in_range & (top | bot | side | lo | hi)) ? base + 4 * off : nullptr;
}

// RSYM (REG only): 17 x 17 template whose rows mirror (row s == row 16 - s): 9 head fragments in registers
// instead of 17 -- the masked epilogue needs the 32 registers.
// CAND (REG instances only): candidate mode of cs_detect_foci compiled in (cs_device.h cand_screen_*); the plain REG
// instances compile it out -- the masked tile sits at 256 registers and seven more spilled ones cost 8 % of C4'.
// The unmasked instances decide at run time (they have room).
template <bool VEC4, bool REG, bool RSYM = false, bool CAND = false>
__global__ __launch_bounds__(256, 2) void corr_mfma_dense_kernel(const MfmaDenseArgs A)
{

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const pl_xh = smem;
    char* const pl_xl = smem + MF_PLANE;
    char* const pl_qh = smem + 2 * MF_PLANE;
    char* const pl_ql = smem + 3 * MF_PLANE;
    float* const raw = reinterpret_cast<float*>(smem + 2 * MF_PLANE);      // next tile's pixels: aliases the squares
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    char* const scr = smem + MFD_SCR + wv * (2 * MF_SCR_PLANE);
    unsigned* const red = reinterpret_cast<unsigned*>(smem + MFD_RED);
    const int km = A.km, kn = A.kn;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;

    h8 ones_b, ones_a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = 8 * g + e - n;
        ones_b[e] = (t >= 0 && t < kn) ? (_Float16)1.0f : (_Float16)0.0f;
        ones_a[e] = (t >= 0 && t < km) ? (_Float16)1.0f : (_Float16)0.0f;
    }
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int wr0 = 16 * wv;
    const int c2 = tid % 40, rg = tid / 40;         // staging: column pair, row group (threads >= 240 idle)
    const int c2_ = c2, rg_ = rg;
    const bool stager = tid < 240;
    // input rows that windows of [row_begin, row_end) reach and that exist
    const int p_min = max(0, A.row_begin - kh), p_max = min(A.ms, A.row_end + (km - 1) - kh) - 1;

    auto tile_origin = [&](int tile, int& I0, int& J0) {
        const int by = tile / A.tiles_x;
        I0 = A.row_begin + by * MF_T;
        J0 = (tile - by * A.tiles_x) * MF_T;
        if (REG && A.band_out) J0 += I0 + A.out_lo;       // the strip of tiles follows the band
    };
    // LDS-DMA of one tile's 80 x 80 pixels (clamped addresses; the reader masks what lies outside).
    // The image is row-major; a wave-wide transfer moves 64 consecutive 4-byte (VEC4: 16-byte) pieces of
    // it, wave w issues the transfers w, w + 4, ...  A lane's piece advances by 256 pieces per step.
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    constexpr int kPiecesPerRow = VEC4 ? MF_R / 4 : MF_R;
    constexpr int kTransfers = VEC4 ? 25 : 100;
    auto fetch = [&](int tile, int slot) {
        int I0, J0;
        tile_origin(tile, I0, J0);
        const int P0 = I0 - kh, Q0 = J0 - kw;
        // The per-lane constants of the transfers are recomputed from an opaque copy of the lane index: hoisted out
        // of the tile loop they are spilled, and a reload here waits (vmcnt) for the previous tile's stores.
        int lane_f = lane;
        if constexpr (REG) asm volatile("" : "+v"(lane_f));         // (the dense instances have registers to spare)
        const int e0 = 64 * wv_u + lane_f;
        int r = e0 / kPiecesPerRow, c = e0 - r * kPiecesPerRow;
        if (REG && !(A.dbg & 16384)) {
            // column terms of the tile's 64 columns (waves 0..2) and the flags of its 80 staged columns (wave 3)
            if (wv_u < 3) {
                const float* src = A.coltab + (size_t)wv_u * A.ns + min(max(J0 + lane_f, 0), A.ns - 1);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(
                    (__attribute__((address_space(3))) char*)(smem) + MFD_COL + (slot * 3 + wv_u) * 256), 4, 0, 0);
            } else {
                // the 80 flag bytes of columns Q0 .. Q0 + 79 as the <= 22 aligned dwords that hold them (lanes
                // beyond are harmless repeats; addresses clamped to the dwords that overlap the array)
                const long long base = (long long)(uintptr_t)A.miss_col;
                const long long first = ((base + Q0) >> 2) << 2;
                long long addr = first + 4 * min(lane_f, 23);
                addr = min(max(addr, (base >> 2) << 2), ((base + A.ns - 1) >> 2) << 2);
                if (lane_f < 24)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>((uintptr_t)addr),
                                                     (__attribute__((address_space(3))) void*)(
                                                         (__attribute__((address_space(3))) char*)(smem) + MFD_CFB + slot * 128), 4, 0, 0);
            }
        }
        const __attribute__((address_space(3))) char* dst =
            (const __attribute__((address_space(3))) char*)(raw) + (VEC4 ? 1024 : 256) * wv_u;
#pragma unroll 1
        for (int i = wv_u; i < kTransfers; i += 4) {
            const int p = min(max(P0 + r, p_min), p_max);
            const float* row = A.sig + ((long long)p - A.row0_in) * A.ld_in;
            if constexpr (VEC4) {
                // REG: the piece's first stored index (band: diagonal index q - p - lo), clamped into the row's
                // stored range -- the reader undoes the shift (16-byte transfers need no 16-byte alignment)
                int q = Q0 + 4 * c;
                if (REG && A.band_in) q = min(max(q - p - A.lo_in, 0), A.bw_in - 4);
                else q = min(max(q, 0), A.ns - 4);
                __builtin_amdgcn_global_load_lds(row + q, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                dst += 4096;
                c += 256 % kPiecesPerRow;
                r += 256 / kPiecesPerRow;
            } else {
                int q = min(max(Q0 + c, 0), A.ns - 1);
                if (REG && A.band_in) q = min(max(q - p - A.lo_in, 0), A.bw_in - 1);      // stored diagonal index
                __builtin_amdgcn_global_load_lds(row + q, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
                dst += 1024;
                c += 256 % kPiecesPerRow;
                r += 256 / kPiecesPerRow;
            }
            if (c >= kPiecesPerRow) {
                c -= kPiecesPerRow;
                r += 1;
            }
        }
    };

    // epilogue of one tile (lane = row n of the wave's 16, columns 16 c + 4 g + v), run one iteration late:
    // its stores then have the whole next tile to retire before the `vmcnt(0)` that awaits the DMA
    auto emit = [&](int I0, int J0, float unscale, const f4 (&acc)[4], const f4 (&S1)[4], const f4 (&S2)[4],
                    const f4& hdr, int slot) {
        const float u_cs = unscale * A.w_unscale;
        const float u_s2 = 32.0f * unscale;
        if constexpr (REG) {
            // ---- per-bin masks: the factorised mask sums of cs_mask_prep.hip (see cs_corr_stream.h MODE 2)
            //   n_missing = K nr[i] + (K - nr[i]) ncol[j],  sum_missing Wa = RA[i] + CA[j] - sum_kj c[j+kj] U_i[kj]
            // plus the precomputed corrections of the pixels whose window leaves the matrix / 0..max_dist
            const int K = km;
            const int i = I0 + wr0 + n;
            KernelStats<float> KS;                                // loaded here, not held across the tile loop
            {
                const StatsPtr sp = stats_args();
                KS.n = sp->n; KS.inv_n = sp->inv_n; KS.kmean = sp->kmean; KS.kstd = sp->kstd; KS.kvar = sp->kvar;
                KS.ksum = sp->ksum; KS.k2sum = sp->k2sum; KS.thr = sp->thr; KS.eps = sp->eps; KS.cut = sp->cut;
                KS.thr_n = sp->thr_n; KS.nkvar = sp->nkvar; KS.eps2 = sp->eps2; KS.den2_min = sp->den2_min;
                KS.zk_possible = sp->zk_possible; KS.snap_possible = sp->snap_possible;
                KS.cand_cmin = CAND ? sp->cand_cmin : 0.0f;
                KS.cand_thr = CAND ? sp->cand_thr : 0.0f;
            }
            const float* colb = reinterpret_cast<const float*>(smem + MFD_COL) + slot * 3 * 64;
            const float nr = hdr[0], ra = hdr[1], rb = hdr[2];
            // The cross term sum_kj c[j + kj] U_i[kj], U_i[kj] = sum over the flagged rows ki of row i's window of
            // W[ki][kj], exists only where a flagged row AND a flagged column reach the window.  On the matrix
            // cores: A = Hankel matrix of the column flags (exact in float16), B = U (head + tail), per column tile.
            const unsigned* cfm = reinterpret_cast<const unsigned*>(smem + MFD_CFL) + slot * 4;
            const unsigned m0 = __builtin_amdgcn_readfirstlane(cfm[0]), m1 = __builtin_amdgcn_readfirstlane(cfm[1]),
                           m2 = __builtin_amdgcn_readfirstlane(cfm[2]);
            const bool tile_flags = (m0 | m1 | m2) != 0;
            const unsigned row_bits = (tile_flags && !(A.dbg & 128)) ? (unsigned)hdr[3] : 0u;
            const bool cross_on = __builtin_amdgcn_ballot_w64(row_bits != 0) != 0;
            h8 ua_h, ua_t, ub_h, ub_t;
            if (cross_on) {
                float ua[8], ub[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ua[e] = ub[e] = 0.0f;
                unsigned bits = row_bits;
                while (__builtin_amdgcn_ballot_w64(bits != 0)) {
                    if (bits) {
                        const int ki = __builtin_ctz(bits);
                        bits &= bits - 1;
                        const float* wa = A.w + K * K + ki * K;
                        const float* wb = A.w + 2 * K * K + ki * K;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int kj = 8 * g + e;
                            const float in = kj < K ? 1.0f : 0.0f;
                            ua[e] = fmaf(wa[min(kj, K - 1)], in, ua[e]);
                            ub[e] = fmaf(wb[min(kj, K - 1)], in, ub[e]);
                        }
                    }
                }
                const float sa = __builtin_amdgcn_rcpf(A.wa_unscale), sb = __builtin_amdgcn_rcpf(A.wb_unscale);   // powers of two
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = ua[e] * sa, b = ub[e] * sb;
                    ua_h[e] = (_Float16)a;
                    ua_t[e] = (_Float16)(a - (float)ua_h[e]);
                    ub_h[e] = (_Float16)b;
                    ub_t[e] = (_Float16)(b - (float)ub_h[e]);
                }
            }
            const unsigned kmask8 = (((1u << K) - 1u) >> (8 * g)) & 0xffu;      // template columns 8 g .. 8 g + 7 that exist
            const int dmin = J0 - (I0 + MF_T - 1), dmax = J0 + MF_T - 1 - I0;
            const bool needs_fix = !(A.dbg & 256) && ((I0 < A.fix_top) | (I0 + MF_T - 1 >= A.fix_bot0) | (A.fix_any_side != 0) |
                                   (A.fix_on && ((dmin < K - 1 && dmax >= 0) |
                                                 (dmax >= A.fix_hi_d0 && dmin < A.fix_hi_d0 + A.fix_hi_w))));
            // every pixel of the tile is produced: rows below row_end, columns inside the matrix, diagonals inside the band
            const bool plain_out = I0 + MF_T <= A.row_end && J0 >= 0 && J0 + MF_T <= A.ns && dmin >= A.out_lo && dmax <= A.out_hi;
            float* tb = reinterpret_cast<float*>(scr);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f4 rv[2], nmv[2], kav[2], kbv[2];
                unsigned rare_bits = 0u;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * half + cc;
                    __builtin_amdgcn_sched_barrier(0);       // one column tile at a time: keeps the live set small
                    // cross term of this column tile
                    f4 xa = zero4, xb = zero4;
#ifndef CS_X_NOCROSS
                    if (cross_on) {
                        // lane (m = n, k group g) of the Hankel operand: flags of staged columns 16 c + n + 8 g + e
                        const int sh = 16 * c + n + 8 * g;                       // 0 .. 87
                        const unsigned lo = sh < 32 ? m0 : sh < 64 ? m1 : m2;
                        const unsigned hi = sh < 32 ? m1 : sh < 64 ? m2 : 0u;
                        const unsigned w8 = (unsigned)((((unsigned long long)hi << 32) | lo) >> (sh & 31)) & kmask8;
                        typedef unsigned u4 __attribute__((ext_vector_type(4)));
                        u4 fw;
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr)
                            fw[pr] = ((w8 >> (2 * pr)) & 1u) * 0x3C00u + ((w8 >> (2 * pr + 1)) & 1u) * 0x3C000000u;   // 1.0 in float16
                        const h8 ff = __builtin_bit_cast(h8, fw);
                        xa = mfma16(ff, ua_h, xa);
                        xa = mfma16(ff, ua_t, xa);
                        xb = mfma16(ff, ub_h, xb);
                        xb = mfma16(ff, ub_t, xb);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            xa[v] *= A.wa_unscale;
                            xb[v] *= A.wb_unscale;
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    const f4 ncol = *reinterpret_cast<const f4*>(colb + 16 * c + 4 * g);
                    const f4 ca = *reinterpret_cast<const f4*>(colb + 64 + 16 * c + 4 * g);
                    const f4 cb = *reinterpret_cast<const f4*>(colb + 128 + 16 * c + 4 * g);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        nmv[cc][v] = (float)K * nr + ((float)K - nr) * ncol[v];
                        kav[cc][v] = ra + ca[v] - xa[v];
                        kbv[cc][v] = rb + cb[v] - xb[v];
                    }
                }
#ifndef CS_X_NOFIX
                if (needs_fix) {
                    // pixels whose window leaves the matrix or the diagonals 0 .. max_dist (tiles on the rim of the band)
                    const FixArgsPtr F = fix_args();
                    const int iw0 = I0 + wr0;                 // the wave's 16 rows
                    const bool rows_hit = (iw0 < A.fix_top) | (iw0 + 15 >= A.fix_bot0) | (A.fix_any_side != 0);
                    // rolled over the half's 8 pixels two at a time: two record loads in flight (one at a time, each awaited
                    // before the next lookup, made a rim tile 45 % slower than an inner one; four at a time spill)
#pragma unroll 1
                    for (int k = 0; k < 4; ++k) {
                        // the wave's pixels of this step: columns j0 + 4 g + {0, 1}, diagonals [j0 - iw0 - 15, j0 + 13 - iw0]
                        const int cc = k >> 1, v0 = 2 * (k & 1);
                        const int j0 = J0 + 16 * (2 * half + cc) + v0;
                        const int dlo = j0 - iw0 - 15, dhi = j0 + 13 - iw0;
                        const bool diag_hit = A.fix_on && ((dlo < K - 1 && dhi >= 0) | (dhi >= A.fix_hi_d0 && dlo < A.fix_hi_d0 + A.fix_hi_w));
                        if (!(rows_hit | diag_hit)) continue;
                        const float* fa = mask_fix_record(A, F, i, j0 + 4 * g);
                        const float* fb = mask_fix_record(A, F, i, j0 + 4 * g + 1);
                        if (!__builtin_amdgcn_ballot_w64((fa != nullptr) | (fb != nullptr))) continue;
                        f4 ra = zero4, rb = zero4;
                        if (fa) ra = *reinterpret_cast<const f4*>(fa);        // 16-byte records, 16-byte aligned tables
                        if (fb) rb = *reinterpret_cast<const f4*>(fb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (e == k) {
                                nmv[e >> 1][2 * (e & 1)] += ra[0];
                                kav[e >> 1][2 * (e & 1)] += ra[1];
                                kbv[e >> 1][2 * (e & 1)] += ra[2];
                                nmv[e >> 1][2 * (e & 1) + 1] += rb[0];
                                kav[e >> 1][2 * (e & 1) + 1] += rb[1];
                                kbv[e >> 1][2 * (e & 1) + 1] += rb[2];
                            }
                        }
                    }
                }
#endif
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * half + cc;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int j = J0 + 16 * c + 4 * g + v;
                        const int d = j - i;
                        const float nm = nmv[cc][v], ka = kav[cc][v], kb = kbv[cc][v];
                        const float cs = acc[c][v] * u_cs;
                        const float s1 = S1[c][v] * unscale;
                        const float s2 = (S2[c][v] * u_s2) * unscale;
                        bool rare;
                        float val = pearson_masked_core<CAND ? 1 : 0>(cs, s1, s2, nm, ka, kb, KS, rare);
                        if constexpr (CAND) val = cand_range_guard(val, s2, unscale, KS);
                        if (A.dbg & 4096) {
                            val = cs + s1 + s2 + nm + ka + kb;
                            rare = false;
                        }
                        const bool forced = A.sym_upper && d + (kn - km) < 0;          // full mode: triu in framed coordinates
                        if (forced) val = 0.0f;
                        if (rare && !forced) rare_bits |= 1u << (4 * cc + v);
                        rv[cc][v] = val;
                    }
                }
                // windows whose sums fall under the 1e-4 thresholds (rare): ONE inlined copy of the exact function
                // per half instead of one per pixel (16 copies: 100 KB of code; a call pins the epilogue to the stack)
#pragma unroll 1
                for (int k = 0; k < (CAND ? 0 : 8); ++k) {      // (candidate mode: such pixels carry the sentinel)
                    if (!__builtin_amdgcn_ballot_w64((rare_bits >> k) != 0)) break;
                    if ((rare_bits >> k) & 1u) {
                        float cs = 0.0f, s1 = 0.0f, s2 = 0.0f, nm = 0.0f, ka = 0.0f, kb = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int c = 2 * half + (e >> 2), v = e & 3;
                            if (e == k) {
                                cs = acc[c][v] * u_cs;
                                s1 = S1[c][v] * unscale;
                                s2 = (S2[c][v] * u_s2) * unscale;
                                nm = nmv[e >> 2][v];
                                ka = kav[e >> 2][v];
                                kb = kbv[e >> 2][v];
                            }
                        }
                        const float val = pearson_masked_f32(cs, s1, s2, nm, ka, kb, KS);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e == k) rv[e >> 2][e & 3] = val;
                    }
                }
                if constexpr (CAND) {
                    // candidate mode: no map leaves the kernel, only the coordinates of the pixels that carry the
                    // sentinel (1e-4 of them), appended to the caller's list -- one atomic per wave and column that
                    // has any.  A.out = the list (keys tag + row * ns + col), A.nobs = its counter, A.ld_out = its
                    // capacity, A.row0_out = the tag (cs_api.cpp find_candidates / cs_detect_foci_blocks).
                    unsigned long long* keys = reinterpret_cast<unsigned long long*>(A.out);
                    unsigned long long* counter = reinterpret_cast<unsigned long long*>(A.nobs);
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int j = J0 + 16 * (2 * half + cc) + 4 * g + v;
                            const int d = j - i;
                            const bool hit = (rv[cc][v] >= KS.cand_thr) & (i < A.row_end) & (j >= 0) & (j < A.ns) &
                                             (d >= A.out_lo) & (d <= A.out_hi);
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                            if (m) {                                          // wave-uniform, rare
                                const int first = __builtin_ctzll(m);
                                unsigned long long base = 0;
                                if (lane == first) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(m));
                                const unsigned lo = __builtin_amdgcn_readlane((unsigned)base, first);
                                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(base >> 32), first);
                                const unsigned long long pos = (((unsigned long long)hi << 32) | lo) +
                                                               (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                                if (hit && pos < (unsigned long long)A.ld_out)
                                    keys[pos] = (unsigned long long)A.row0_out + (unsigned long long)i * (unsigned long long)A.ns + (unsigned long long)j;
                            }
                        }
                    }
                    continue;
                }
                // 16 rows x 32 columns through the wave's scratch, out as 2 rows x 32 consecutive floats per
                // instruction (band rows are shifted against each other: no wider aligned store exists)
                const int pass_n = (A.dbg & 8192) ? (int)(rv[0][0] == 123.456f) : (A.nobs ? 2 : 1);
                for (int pass = 0; pass < pass_n; ++pass) {
                    wave_lds_sync();
                    f4 t0 = rv[0], t1 = rv[1];
                    if (pass) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            t0[v] = KS.n - nmv[0][v];
                            t1[v] = KS.n - nmv[1][v];
                        }
                    }
                    *reinterpret_cast<f4*>(tb + n * 36 + 4 * g) = t0;
                    *reinterpret_cast<f4*>(tb + n * 36 + 16 + 4 * g) = t1;
                    wave_lds_sync();
                    const int xx = lane & 31;
                    int oi = I0 + wr0 + (lane >> 5);
                    const int oj = J0 + 32 * half + xx;
                    // two rows further down: 2 ld, and on a banded output two diagonals to the left
                    long long idx = ((long long)oi - A.row0_out) * A.ld_out + (A.band_out ? oj - oi - A.lo_out : oj);
                    const long long step = 2 * (long long)A.ld_out - (A.band_out ? 2 : 0);
                    const bool col_ok = oj >= 0 && oj < A.ns;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const float val = tb[(2 * it + (lane >> 5)) * 36 + xx];
                        bool ok = !(A.dbg & 1);
                        if (!plain_out) {
                            const int d = oj - oi;
                            ok = ok && oi < A.row_end && col_ok && d >= A.out_lo && d <= A.out_hi;
                        }
                        if (ok) {
                            if (pass == 1) A.nobs[idx] = val;
                            else if (A.out_is_f64) reinterpret_cast<double*>(A.out)[idx] = (double)val;
                            else reinterpret_cast<float*>(A.out)[idx] = val;
                        }
                        oi += 2;
                        idx += step;
                    }
                }
            }
            return;
        }
        const bool plain = I0 + MF_T <= A.row_end && J0 + MF_T <= A.ns &&
                           (A.full || (I0 >= kh && I0 + MF_T - 1 <= A.ms - km + kh && J0 >= kw && J0 + MF_T - 1 <= A.ns - kn + kw)) &&
                           (!A.sym_upper || J0 - (I0 + MF_T - 1) + (A.full ? kn - km : 0) >= 0);
        const int i = I0 + wr0 + n;
        const long long o_idx = ((long long)i - A.row0_out) * A.ld_out + (J0 + 4 * g);
        f4 rv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float cs = acc[c][v] * u_cs;
                const float s1 = S1[c][v] * unscale;
                const float s2 = (S2[c][v] * u_s2) * unscale;
                rv[c][v] = A.xcorr_only ? (fabsf(cs) < A.ks.thr ? 0.0f : cs)            // detection.py:716-722
                                        : (A.dbg & 32) ? cs + s1 + s2
                                                       : cand_range_guard(pearson_nomask_lean(cs, s1, s2, A.ks), s2, unscale, A.ks);
            }
        }
        if (A.dbg & 1) {
            if (rv[0][0] == 123.456f) reinterpret_cast<float*>(A.out)[o_idx] = rv[1][1] + rv[2][2] + rv[3][3];
        } else if (plain && VEC4) {
            // A lane holds 4 consecutive columns of ONE row per column tile: stored directly, every
            // instruction would touch 16 rows with 64 bytes each (measured: 4x the cost of the same bytes
            // at consecutive addresses).  Two column tiles at a time go through the wave's scratch
            // (16 rows x 32 columns, pitch 36 floats) and leave as 8 rows x 128 contiguous bytes.
            float* tb = reinterpret_cast<float*>(scr);
            const long long r_idx = ((long long)(I0 + wr0 + (lane >> 3)) - A.row0_out) * A.ld_out + J0 + 4 * (lane & 7);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                wave_lds_sync();
                *reinterpret_cast<f4*>(tb + n * 36 + 4 * g) = rv[2 * half];
                *reinterpret_cast<f4*>(tb + n * 36 + 16 + 4 * g) = rv[2 * half + 1];
                wave_lds_sync();
                const f4 lo = *reinterpret_cast<const f4*>(tb + (lane >> 3) * 36 + 4 * (lane & 7));
                const f4 hi = *reinterpret_cast<const f4*>(tb + ((lane >> 3) + 8) * 36 + 4 * (lane & 7));
                if (A.out_is_f64) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    double* od = reinterpret_cast<double*>(A.out) + r_idx + 32 * half;
                    d2 a, b;
                    a[0] = lo[0]; a[1] = lo[1]; b[0] = lo[2]; b[1] = lo[3];
                    *reinterpret_cast<d2*>(od) = a;
                    *reinterpret_cast<d2*>(od + 2) = b;
                    a[0] = hi[0]; a[1] = hi[1]; b[0] = hi[2]; b[1] = hi[3];
                    *reinterpret_cast<d2*>(od + 8 * A.ld_out) = a;
                    *reinterpret_cast<d2*>(od + 8 * A.ld_out + 2) = b;
                } else {
                    float* of = reinterpret_cast<float*>(A.out) + r_idx + 32 * half;
                    *reinterpret_cast<f4*>(of) = lo;
                    *reinterpret_cast<f4*>(of + 8 * A.ld_out) = hi;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int j = J0 + 16 * c + 4 * g + v;
                    bool z = false;
                    if (!A.full) z = (i < kh) | (i > A.ms - km + kh) | (j < kw) | (j > A.ns - kn + kw);
                    if (A.sym_upper) z = z | ((j - i) + (A.full ? (kn - km) : 0) < 0);
                    if (i < A.row_end && j < A.ns) {
                        const float val = z ? 0.0f : rv[c][v];
                        if (A.out_is_f64) reinterpret_cast<double*>(A.out)[o_idx + 16 * c + v] = (double)val;
                        else reinterpret_cast<float*>(A.out)[o_idx + 16 * c + v] = val;
                    }
                }
            }
        }
    };

    f4 p_acc[4], p_S1[4], p_S2[4], p_hdr = zero4;
    int p_I0 = 0, p_J0 = 0, p_slot = 0, slot = 0;
    float p_unscale = 0.0f;
    bool pending = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) p_acc[c] = p_S1[c] = p_S2[c] = zero4;

    // Tile sequence of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs (own L2 each): XCD x
    // takes the tiles [x n / 8, (x + 1) n / 8) and its workgroups walk that range side by side, so the 16
    // halo rows / columns neighbouring tiles share are re-read from the XCD's own L2.
    int tile, tile_end, tile_step;
    if (A.xcd_order && gridDim.x % 8 == 0) {
        const int x = blockIdx.x & 7, per = (A.n_tiles + 7) / 8;
        tile = x * per + (blockIdx.x >> 3);
        tile_end = min(A.n_tiles, (x + 1) * per);
        tile_step = gridDim.x >> 3;
    } else {
        tile = blockIdx.x;
        tile_end = A.n_tiles;
        tile_step = gridDim.x;
    }
    if (A.dbg & 2048) return;                  // diagnostics: launch + dispatch only
    if (tile < tile_end) fetch(tile, 0);       // the first tile is on its way while the weights are loaded

    // ---- weights: heads of all 17 template rows in registers (rows >= km are zero), tails in LDS
    constexpr int kHeads = RSYM ? 9 : 17;
    h8 wh[kHeads];
    {
        const h8* frag = reinterpret_cast<const h8*>(A.frag);
        h8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
        for (int s = 0; s < kHeads; ++s) {
            const int sc = min(s, km - 1);
            const h8 a = frag[(2 * sc + 0) * 64 + lane];
            wh[s] = s < km ? a : z;
        }
        h8 tails[5];                                   // 17 x 64 fragments = 4.25 per thread: loads first
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = min(tid + 256 * k, 17 * 64 - 1);
            tails[k] = frag[(2 * min(idx >> 6, km - 1) + 1) * 64 + (idx & 63)];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = tid + 256 * k;
            if (idx < 17 * 64) reinterpret_cast<h8*>(smem + MFD_WL)[idx] = (idx >> 6) < km ? tails[k] : z;
        }
    }
    // REG: one emit site (its epilogue is 4 K instructions) -- the loop runs one pass beyond the last tile, in which
    // only the pending tile is emitted.  The dense instances emit the last tile after the loop.
    for (;; tile += tile_step) {
        const bool have = REG ? tile < tile_end : true;              // uniform over the workgroup
        if constexpr (REG) {
            if (!have && !pending) break;
        } else {
            if (tile >= tile_end) break;
        }
        int I0 = 0, J0 = 0;
        if (have) tile_origin(tile, I0, J0);
        const int P0 = I0 - kh, Q0 = J0 - kw;
        // every staged pixel exists (rows, columns and, for a banded input, stored diagonals): the transfers were not
        // clamped and the reader needs no masks
        bool inside = P0 >= p_min && P0 + MF_R - 1 <= p_max && Q0 >= 0 && Q0 + MF_R <= A.ns;
        if (REG && A.band_in) inside = inside && Q0 - (P0 + MF_R - 1) - A.lo_in >= 0 && Q0 + MF_R - 1 - P0 - A.lo_in <= A.bw_in - 1;
        if (REG && (A.dbg & 512)) inside = true;
        if (A.dbg & 1024) {                    // diagnostics: prologue only
            if (!have) break;
            continue;
        }
        float unscale = 0.0f;
#ifdef CS_MF_PROFILE
        unsigned long long tprev_ = __builtin_readcyclecounter();
        if (tid == 0 && have) atomicAdd(&cs_mf_prof[15], 1ull);
#endif
        if (have) {

            // ---- the tile's pixels have landed in `raw`: read this thread's 14 x 2, find the scale
            __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's DMA transfers (and long-retired stores)
            lds_barrier();                           // everyone's transfers; the previous tile's plane readers are done
            float xa[MFD_ROWS_PER_THREAD], xb[MFD_ROWS_PER_THREAD];
            float amax = 0.0f;
            if (inside) {
#pragma unroll
                for (int k = 0; k < MFD_ROWS_PER_THREAD; ++k) {
                    const int r = rg + 6 * k;
                    float2 v = make_float2(0.0f, 0.0f);
                    if (stager && r < MF_R) v = *reinterpret_cast<const float2*>(raw + r * MF_R + 2 * c2);
                    xa[k] = v.x;
                    xb[k] = v.y;
                    amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
                }
            } else {
                int c2 = c2_, rg = rg_;               // (opaque copies: see fetch)
                if constexpr (REG) asm volatile("" : "+v"(c2), "+v"(rg));
#pragma unroll
                for (int k = 0; k < MFD_ROWS_PER_THREAD; ++k) {
                    const int r = rg + 6 * k;
                    float a = 0.0f, b = 0.0f;
                    if (REG && VEC4) {
                        if (stager && r < MF_R) {
                            // pieces of 4 whose start was clamped into the stored range of their row: element o of the
                            // piece holds stored index idx_c + o
                            const int p = P0 + r, q = Q0 + 2 * c2, cs = (2 * c2) & ~3;
                            const int w = A.band_in ? A.bw_in : A.ns;
                            const int off = A.band_in ? p + A.lo_in : 0;
                            const int idx_s = Q0 + cs - off;
                            const int idx_c = min(max(idx_s, 0), w - 4);
                            const int oa = (2 * c2 - cs) + (idx_s - idx_c), ob = oa + 1;
                            const int idx_a = q - off;
                            const bool rok = (p >= p_min) & (p <= p_max);
                            const bool oka = rok & (q >= 0) & (q < A.ns) & (idx_a >= 0) & (idx_a < w) & (oa >= 0) & (oa < 4);
                            const bool okb = rok & (q + 1 >= 0) & (q + 1 < A.ns) & (idx_a + 1 >= 0) & (idx_a + 1 < w) & (ob >= 0) & (ob < 4);
                            const float* piece = raw + r * MF_R + cs;
                            a = oka ? piece[min(max(oa, 0), 3)] : 0.0f;
                            b = okb ? piece[min(max(ob, 0), 3)] : 0.0f;
                        }
                    } else if (stager && r < MF_R) {
                        const float2 v = *reinterpret_cast<const float2*>(raw + r * MF_R + 2 * c2);
                        const int p = P0 + r, q = Q0 + 2 * c2;
                        const bool rok = (p >= p_min) & (p <= p_max);
                        bool oka = rok & (q >= 0) & (q < A.ns), okb = rok & (q + 1 >= 0) & (q + 1 < A.ns);
                        if (REG && A.band_in) {               // outside the stored diagonals: zero
                            const int dd = q - p - A.lo_in;
                            oka &= (dd >= 0) & (dd < A.bw_in);
                            okb &= (dd + 1 >= 0) & (dd + 1 < A.bw_in);
                        }
                        a = oka ? v.x : 0.0f;
                        b = okb ? v.y : 0.0f;
                    }
                    xa[k] = a;
                    xb[k] = b;
                    amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
                }
            }
            amax = wave_max_nonneg(amax);
            if (lane == 63) red[wv] = __float_as_uint(amax);
            if constexpr (REG) {
                // the tile's column flags as bit masks (bit t = staged column Q0 + t), zero outside the matrix (the
                // transfer clamped its addresses)
                if (wv < 2) {
                    const int q = Q0 + tid;
                    const long long start = (long long)(uintptr_t)A.miss_col + Q0;
                    const int skew = (int)(start - ((start >> 2) << 2));           // bytes before column Q0 in the first dword
                    const uint8_t fb = reinterpret_cast<const uint8_t*>(smem + MFD_CFB)[slot * 128 + skew + min(tid, 79)];
                    const bool flagged = tid < 80 && q >= 0 && q < A.ns && fb;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(flagged);
                    unsigned* cfm = reinterpret_cast<unsigned*>(smem + MFD_CFL) + slot * 4;
                    if (lane == 0) {
                        cfm[2 * wv] = (unsigned)m;
                        if (wv == 0) cfm[1] = (unsigned)(m >> 32);
                    }
                }
            }
            MF_STAMP(0);
            lds_barrier();                           // everyone has read `raw`: the squares may overwrite it
            int ex = 0;
            {
                const uint4 m4 = *reinterpret_cast<const uint4*>(red);
                const unsigned mx = max(max(m4.x, m4.y), max(m4.z, m4.w));     // non-negative floats order as integers
                const int e = (int)((mx >> 23) & 0xffu);
                if (e != 0 && e != 255) ex = 6 - (e - 127);
                ex = max(-100, min(100, ex));
            }
            const float scale = __uint_as_float((unsigned)(ex + 127) << 23);
            const float qscale = __uint_as_float((unsigned)(ex + 127 - 5) << 23);   // 2^-5: 17-sums of squares < 65504
            unscale = __uint_as_float((unsigned)(127 - ex) << 23);
            typedef __fp16 hv2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < MFD_ROWS_PER_THREAD; ++k) {
                const int r = rg + 6 * k;
                if (stager && r < MF_R && !(A.dbg & 16)) {
                    // heads by truncation, tails exact differences: head + tail carries 21-22 bits either way
                    const float a = xa[k] * scale, b = xb[k] * scale;
                    const hv2 hh = __builtin_amdgcn_cvt_pkrtz(a, b);
                    const hv2 tt = __builtin_amdgcn_cvt_pkrtz(a - (float)hh[0], b - (float)hh[1]);
                    const float qa = (xa[k] * qscale) * a, qb = (xb[k] * qscale) * b;
                    const hv2 qh = __builtin_amdgcn_cvt_pkrtz(qa, qb);
                    const hv2 qt = __builtin_amdgcn_cvt_pkrtz(qa - (float)qh[0], qb - (float)qh[1]);
                    const int o = (r * MF_R + 2 * c2) * 2;
                    *reinterpret_cast<hv2*>(pl_xh + o) = hh;
                    *reinterpret_cast<hv2*>(pl_xl + o) = tt;
                    *reinterpret_cast<hv2*>(pl_qh + o) = qh;
                    *reinterpret_cast<hv2*>(pl_ql + o) = qt;
                }
            }
            MF_STAMP(1);
        } else if (REG) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            lds_barrier();
        }
        // ---- the previous tile's coefficients and stores
        if (pending) emit(p_I0, p_J0, p_unscale, p_acc, p_S1, p_S2, p_hdr, p_slot);
        if (!have) break;
        MF_STAMP(6);
        lds_barrier();

        // ---- box sums: horizontal pass over the wave's 32 input rows (all-ones Toeplitz as B), the
        //      partial sums split again and transposed through the wave's scratch, vertical pass
        //      (all-ones Toeplitz as B again: transposed tile).  8 steps (4 column tiles x {x, x^2}); the
        //      fragments of step t + 1 are loaded before the scratch round trip of step t.
        f4 S1[4], S2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) S1[c] = S2[c] = zero4;
        if (!(A.dbg & 4) && !A.xcorr_only) {
            auto hfrag = [&](int t, h8 (&f)[4]) {
                const int c = t >> 1;
                const char* ph = (t & 1) ? pl_qh : pl_xh;
                const char* pt = (t & 1) ? pl_ql : pl_xl;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int off = ((wr0 + 16 * rb + n) * MF_R + 16 * c + 8 * g) * 2;
                    f[2 * rb] = *reinterpret_cast<const h8*>(ph + off);
                    f[2 * rb + 1] = *reinterpret_cast<const h8*>(pt + off);
                }
            };
            h8 cur[4];
            hfrag(0, cur);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                h8 nxt[4];
                if (t + 1 < 8) hfrag(t + 1, nxt);
                h4 th[2], tl[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    f4 h = mfma16(cur[2 * rb], ones_b, zero4);
                    h = mfma16(cur[2 * rb + 1], ones_b, h);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        th[rb][v] = (_Float16)h[v];
                        tl[rb][v] = (_Float16)(h[v] - (float)th[rb][v]);
                    }
                }
                wave_lds_sync();                 // the previous step's scratch reads are issued
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int so = (n * MF_SCR_PITCH + 16 * rb + 4 * g) * 2;
                    *reinterpret_cast<h4*>(scr + so) = th[rb];
                    *reinterpret_cast<h4*>(scr + MF_SCR_PLANE + so) = tl[rb];
                }
                wave_lds_sync();
                const int ro = (n * MF_SCR_PITCH + 8 * g) * 2;
                const h8 bh = *reinterpret_cast<const h8*>(scr + ro);
                const h8 bl = *reinterpret_cast<const h8*>(scr + MF_SCR_PLANE + ro);
                f4 sacc = mfma16(bh, ones_a, zero4);
                sacc = mfma16(bl, ones_a, sacc);
                if (t & 1) S2[t >> 1] = sacc;
                else S1[t >> 1] = sacc;
                if (t + 1 < 8) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
                }
            }
        }
        MF_STAMP(2);
        lds_barrier();                           // all waves are done with the squares
        // ---- next tile's pixels -> `raw` while this tile's correlation runs
        if (tile + tile_step < tile_end && !(A.dbg & 2)) fetch(tile + tile_step, slot ^ 1);

        MF_STAMP(3);
        // ---- cross term: 17 template rows x 4 column tiles, fragments of row s + 1 in flight during row s
        f4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = zero4;
        h8 ah[4], al[4], bl;
        const int fo = ((wr0 + n) * MF_R + 8 * g) * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ah[c] = *reinterpret_cast<const h8*>(pl_xh + fo + 32 * c);
            al[c] = *reinterpret_cast<const h8*>(pl_xl + fo + 32 * c);
        }
        bl = reinterpret_cast<const h8*>(smem + MFD_WL)[lane];
        if (!(A.dbg & 8))
#pragma unroll
        for (int s = 0; s < 17; ++s) {
            h8 nh[4], nl[4], nb;
            if (s + 1 < 17) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    nh[c] = *reinterpret_cast<const h8*>(pl_xh + fo + (s + 1) * MF_R * 2 + 32 * c);
                    nl[c] = *reinterpret_cast<const h8*>(pl_xl + fo + (s + 1) * MF_R * 2 + 32 * c);
                }
                nb = reinterpret_cast<const h8*>(smem + MFD_WL)[(s + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma16(wh[RSYM && s > 8 ? 16 - s : s], ah[c], acc[c]);     // weights as A: transposed tile
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma16(bl, ah[c], acc[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma16(wh[RSYM && s > 8 ? 16 - s : s], al[c], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < 17) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ah[c] = nh[c];
                    al[c] = nl[c];
                }
                bl = nb;
            }
        }
        MF_STAMP(4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            p_acc[c] = acc[c];
            p_S1[c] = S1[c];
            p_S2[c] = S2[c];
        }
        p_I0 = I0;
        p_J0 = J0;
        p_unscale = unscale;
        p_slot = slot;
        if constexpr (REG) p_hdr = *reinterpret_cast<const f4*>(A.rowtab + 4 * (size_t)min(I0 + wr0 + n, A.ms - 1));
        slot ^= 1;
        pending = true;
    }
    if constexpr (!REG) {
        if (pending) emit(p_I0, p_J0, p_unscale, p_acc, p_S1, p_S2, p_hdr, p_slot);
    }
}

// The 160 KB dynamic-LDS ceiling is a per-function, per-device attribute: set it the first time a kernel is
// launched on a device (the call costs microseconds on every launch otherwise).
static hipError_t allow_big_lds(const void* fn)
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

int launch_corr_mfma_f32(CorrArgs<float>& A, const MfmaWeights& E, hipStream_t stream, int* dense_path)
{
    *dense_path = 0;
    A.tile_w = A.tile_h = MF_T;
    A.tiles_y = (A.row_end - A.row_begin + MF_T - 1) / MF_T;
    if (A.out.layout == 1) {
        A.out_lo = A.out.band_lo;
        A.out_hi = A.out.band_lo + A.out.band_w - 1;
        A.tiles_x = (A.out.band_w + MF_T - 1 + MF_T - 1) / MF_T;
    } else {
        A.out_lo = -(1 << 30);
        A.out_hi = (1 << 30);
        A.tiles_x = (A.ns + MF_T - 1) / MF_T;
    }
    const long long blocks = (long long)A.tiles_x * A.tiles_y;
    if (blocks <= 0) return 0;
    if (blocks > 0x7fffffffLL) return -3;
    const bool masked = A.mask_mode != 0;
    const bool dense_f32 = !masked && A.sig.layout == 0 && A.out.layout == 0 && !A.sig_is_f64 && !A.nobs.ptr && A.ms > 0 && A.ns > 0;
    // per-bin masks with the factorised tables in place (cs_api.cpp prepare_regular_mask), square template
    const bool reg_f32 = A.mask_mode == 1 && A.reg_mode == 1 && A.km == A.kn && !A.sig_is_f64 && !A.xcorr_only && A.full &&
                         A.ms > 0 && A.ns > 0 &&
                         (!A.nobs.ptr || (A.nobs.layout == A.out.layout && A.nobs.ld == A.out.ld && A.nobs.band_lo == A.out.band_lo &&
                                          A.nobs.band_w == A.out.band_w && A.nobs.row0 == A.out.row0));
    if ((dense_f32 || reg_f32) && !getenv("CHROMOSIGHT_HIP_MFMA_V1")) {
        *dense_path = 1;
        MfmaDenseArgs D;
        D.sig = reinterpret_cast<const float*>(A.sig.ptr);
        D.out = A.out.ptr;
        D.out_is_f64 = A.out_is_f64;
        D.ld_in = A.sig.ld;
        D.ld_out = A.out.ld;
        D.row0_in = A.sig.row0;
        D.row0_out = A.out.row0;
        D.ms = A.ms;
        D.ns = A.ns;
        D.km = A.km;
        D.kn = A.kn;
        D.row_begin = A.row_begin;
        D.row_end = A.row_end;
        D.full = A.full;
        D.sym_upper = A.sym_upper;
        D.xcorr_only = A.xcorr_only;
        D.tiles_x = A.tiles_x;
        D.n_tiles = (int)blocks;
        D.frag = E.frag;
        D.w_unscale = E.unscale[0];
        D.wa_unscale = E.unscale[1];
        D.wb_unscale = E.unscale[2];
        D.dbg = getenv("CHROMOSIGHT_HIP_MFMA_DBG") ? atoi(getenv("CHROMOSIGHT_HIP_MFMA_DBG")) : 0;
        D.xcd_order = getenv("CHROMOSIGHT_HIP_NO_XCD") ? 0 : 1;
        D.ks = A.ks;
        D.band_in = A.sig.layout == 1;
        D.lo_in = A.sig.band_lo;
        D.bw_in = A.sig.band_w;
        D.band_out = A.out.layout == 1;
        D.lo_out = A.out.band_lo;
        D.out_lo = A.out_lo;
        D.out_hi = A.out_hi;
        D.nobs = reinterpret_cast<float*>(A.nobs.ptr);
        D.w = A.w;
        D.miss_col = A.miss_col;
        D.rowtab = A.rowtab;
        D.coltab = A.coltab;
        D.fix_on = A.fix_on;
        D.fix_hi_w = A.fix_hi_w;
        D.fix_hi_d0 = A.fix_hi_d0;
        D.fix_top = A.fix_top;
        D.fix_bot0 = A.fix_bot0;
        D.fix_any_side = A.fix_cols != nullptr;
        D.fx.fix_on = A.fix_on;
        D.fx.fix_hi_w = A.fix_hi_w;
        D.fx.fix_hi_d0 = A.fix_hi_d0;
        D.fx.fix_lo = A.fix_lo;
        D.fx.fix_hi = A.fix_hi;
        D.fx.fix_rows = A.fix_rows;
        D.fx.fix_cols = A.fix_cols;
        D.fx.fix_top = A.fix_top;
        D.fx.fix_bot0 = A.fix_bot0;
        D.fx.fix_width = A.fix_width;
        D.fx.fix_xband = A.fix_xband;
        D.fx.fix_xlo = A.fix_xlo;
        D.fx.fix_side = A.fix_side;
        if (reg_f32) {
            *dense_path = 2;
            // 16-byte transfers need rows that store at least 4 values (no alignment needed); narrower maps take the
            // streaming kernel
            if ((D.band_in ? D.bw_in : D.ns) < 4) {
                *dense_path = 0;
                return -4;
            }
            *dense_path = 2;
            const bool rsym = A.w_sym && A.km == 17 && A.kn == 17 && !getenv("CHROMOSIGHT_HIP_MFMA_NORSYM");
            // candidate mode: the CAND instances append candidate coordinates to the caller's list and write no map
            const bool cand = A.ks.cand_cmin > 0.0f;
            if (cand) {
                if (!A.cand_keys || !A.cand_count) return -5;
                D.out = A.cand_keys;
                D.nobs = reinterpret_cast<float*>(A.cand_count);
                D.ld_out = A.cand_cap;
                D.row0_out = (long long)A.cand_tag;
            }
            typedef void (*reg_kernel_t)(const MfmaDenseArgs);
            const reg_kernel_t kr = rsym ? (cand ? corr_mfma_dense_kernel<true, true, true, true> : corr_mfma_dense_kernel<true, true, true, false>)
                                         : (cand ? corr_mfma_dense_kernel<true, true, false, true> : corr_mfma_dense_kernel<true, true, false, false>);
            hipError_t e3 = allow_big_lds((const void*)kr);
            if (e3 != hipSuccess) return (int)e3;
            const int per_cu_r = getenv("CHROMOSIGHT_HIP_MFMA_GRID") ? atoi(getenv("CHROMOSIGHT_HIP_MFMA_GRID")) : 2;
            const int grid_r = (int)std::min<long long>(blocks, (long long)per_cu_r * A.n_cu);
            hipLaunchKernelGGL(kr, dim3((unsigned)grid_r), dim3(256), MFD_SMEM_REG, stream, D);
            return (int)hipGetLastError();
        }
        if (!A.out.ptr) return -5;
        // 16-byte pieces: the tile's first staged column (64 bx - kw) and the row length must be multiples
        // of 4 so that no piece straddles the matrix edge; the transfers themselves need only 4-byte alignment
        // (checked bit for bit against 4-byte transfers at every misalignment), the 16-byte stores aligned rows
        const bool vec4 = ((uintptr_t)D.out % 32 == 0) && D.ld_out % 4 == 0 &&
                          ((A.kn - 1) / 2) % 4 == 0 && D.ns % 4 == 0 && D.ns >= 4 && !getenv("CHROMOSIGHT_HIP_MFMA_NOVEC");
        const void* kd = vec4 ? (const void*)corr_mfma_dense_kernel<true, false> : (const void*)corr_mfma_dense_kernel<false, false>;
        hipError_t e2 = allow_big_lds(kd);
        if (e2 != hipSuccess) return (int)e2;
        const int per_cu = getenv("CHROMOSIGHT_HIP_MFMA_GRID") ? atoi(getenv("CHROMOSIGHT_HIP_MFMA_GRID")) : 2;
        const int grid = (int)std::min<long long>(blocks, (long long)per_cu * A.n_cu);
        if (vec4) hipLaunchKernelGGL((corr_mfma_dense_kernel<true, false>), dim3((unsigned)grid), dim3(256), MFD_SMEM, stream, D);
        else hipLaunchKernelGGL((corr_mfma_dense_kernel<false, false>), dim3((unsigned)grid), dim3(256), MFD_SMEM, stream, D);
        return (int)hipGetLastError();
    }
    if (!A.out.ptr) return -5;                   // (a candidate sink without a map: only the masked tile kernel serves that)
    const void* kern = masked ? (const void*)corr_mfma_kernel<true> : (const void*)corr_mfma_kernel<false>;
    hipError_t e = allow_big_lds(kern);
    if (e != hipSuccess) return (int)e;
    if (masked) hipLaunchKernelGGL(corr_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    else hipLaunchKernelGGL(corr_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    return (int)hipGetLastError();
}

}  // namespace cs
