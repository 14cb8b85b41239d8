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
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "cs_launch.h"

namespace cs {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float cand_range_guard(float r, float s2, float unscale, const KernelStats<float>& K);
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte global access at 4-byte alignment (band rows)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

constexpr int MF_T = 64;                      // output tile edge
constexpr int MF_R = 80;                      // staged rows / columns (tile + 16)
constexpr int MF_PLANE = MF_R * MF_R * 2;     // bytes of one float16 plane
constexpr int MF_WSET = 17 * 2 * 1024;        // bytes of one weight set's fragments
constexpr int MF_SCR_PITCH = 40;              // halfs per column of the transposed scratch (32 rows + pad)
constexpr int MF_SCR_PLANE = 16 * MF_SCR_PITCH * 2;
constexpr int MF_SMEM = 3 * MF_PLANE + MF_WSET + 64;
constexpr int MF_PER_THREAD = (MF_R * MF_R) / 256;   // 25 staged pixels per thread

static_assert(5 * MF_SCR_PLANE * 4 <= MF_WSET, "scratch aliases the weight region");

// Ablation switches of the persistent kernels (CHROMOSIGHT_HIP_MFMA_DBG: skip a phase, results garbage) exist only in
// builds with -DCS_MF_DEBUG: in the product build the word would be one more live scalar and a branch per phase.
#ifdef CS_MF_DEBUG
#define MFD_DBG(bit) (A.dbg & (bit))
#else
#define MFD_DBG(bit) (false)
#endif

#ifdef CS_MF_PROFILE
// Per-phase cycle stamps of thread 0 of every workgroup.  The persistent kernels accumulate in LDS (prof_lds, 16 words
// behind the kernel's own LDS) and add their totals to the device counters once, when the workgroup ends: one global
// atomic per stamp and tile from 512 workgroups on the same six words cost more than the phases they measured.
__device__ unsigned long long cs_mf_prof[16];
#define MF_STAMP(k)                                                        \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        if (tid == 0) MF_PROF_ADD(k, now_ - tprev_);                       \
        tprev_ = now_;                                                     \
    } while (0)
#define MF_PROF_ADD(k, v) atomicAdd(&cs_mf_prof[k], (unsigned long long)(v))
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
    const float* rim;        // rim tables of the weights (cs_launch.h MfmaWeights::rim)
    int max_dist;
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
    int by_cut;              // REG band outputs: tile rows from this one on are cut at the matrix's last column (see tile_origin)
    const uint4* frag;
    float w_unscale;
    float wa_unscale, wb_unscale;    // REG: power-of-two scales of the Wa / Wb sets (float16 range)
    int xcd_order;           // bit 0: contiguous tile ranges per XCD; bits 1-2 (dense instances): column skew per tile row
    int dbg;                 // diagnostics (CHROMOSIGHT_HIP_MFMA_DBG): skip 1 stores, 2 prefetch, 4 box sums, 8 cross term
    KernelStats<float> ks;
    // ---- REG instances: per-bin missing masks through the factorised tables of cs_mask_prep.hip, any
    //      layout (band / dense) on either side, optional n_obs map
    int band_in, lo_in, bw_in;       // input band: first stored diagonal, stored diagonals
    int pad_in;                      // ... zero-padded (CS_LAYOUT_BAND_PADDED): >= 4 zero slots behind every row's stored diagonals
    int cnt_in;                      // ... of raw counts (CS_LAYOUT_BAND_COUNTS): detrended as a landed tile is split (cs_device.h CountsHeader)
    int band_out, lo_out;            // output band: first stored diagonal
    int out_lo, out_hi;              // produced diagonals
    float* nobs;                     // same geometry as `out`, or nullptr
    const float* w;                  // float32 weight sets (K*K each): centred template, Wa, Wb
    const uint8_t* miss_col;
    const float* rowtab;             // [row][4]: nr, RA, RB, flags of the window rows
    const float* coltab;             // [3][ns]: ncol, CA, CB
    int fix_on, fix_hi_w, fix_hi_d0, fix_top, fix_bot0, fix_any_side;     // what decides whether a tile has records
    int rim_in_kernel;       // the corrections of the edge diagonals are formed in the epilogue (no fix_lo / fix_hi records)
    MfmaFixArgs fx;          // read through fix_args() only
};

constexpr int MFD_ROWS_PER_THREAD = 14;       // staging: 240 threads = 40 column pairs x 6 row groups
constexpr int MFD_WL = 4 * MF_PLANE;          // weight tails
constexpr int MFD_SCR = MFD_WL + 18 * 1024;   // (17 tail rows; RSYM: 9 tail rows + 9 head rows) then, per wave: scratch of the stores
constexpr int MFD_RED = MFD_SCR + 4 * 2 * MF_SCR_PLANE;
constexpr int MFD_SMEM = MFD_RED + 64;
// REG: column terms (ncol, CA, CB of the tile's 64 columns) and the column flags of its 80 staged columns,
// double-buffered (a tile's epilogue runs while the next tile's are already landing)
constexpr int MFD_COL = MFD_RED + 64;
constexpr int MFD_CFL = MFD_COL + 2 * 3 * 64 * 4;
constexpr int MFD_CFB = MFD_CFL + 2 * 16;              // (CFL: the 80 column flags as 3 mask words per slot) raw flag bytes as they arrive
constexpr int MFD_SMEM_REG = MFD_CFB + 2 * 128;
static_assert(MFD_SMEM_REG <= 80 * 1024, "two workgroups per CU");
#ifdef CS_MF_PROFILE
constexpr int MFD_PROF = MFD_SMEM_REG;             // 16 x 8 bytes of per-workgroup phase counters
constexpr int MFD_LAUNCH_EXTRA = 128;
#else
constexpr int MFD_LAUNCH_EXTRA = 0;
#endif

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

// ... and the whole argument block for the epilogue of the masked instances
typedef const __attribute__((address_space(4))) MfmaDenseArgs* ArgsPtr;

__device__ __forceinline__ ArgsPtr args_here()
{
    unsigned long long kv = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kv));
    return (ArgsPtr)kv;
}

// the header in front of a band of counts (cs_device.h), through the constant address space: scalar loads
typedef const __attribute__((address_space(4))) CountsHeader* CountsHdrPtr;

__device__ __forceinline__ CountsHdrPtr counts_header(const float* band)
{
    return (CountsHdrPtr)((unsigned long long)band - (unsigned long long)kCountsHeaderBytes);
}

__device__ __forceinline__ ArgsPtr args_at(unsigned long long entry)
{
    asm volatile("" : "+s"(entry));
    return (ArgsPtr)entry;
}

// the same two blocks of an entry of a device table (corr_mfma_blocks_kernel)
__device__ __forceinline__ FixArgsPtr fix_args_at(unsigned long long entry)
{
    asm volatile("" : "+s"(entry));
    return (FixArgsPtr)(entry + offsetof(MfmaDenseArgs, fx));
}

__device__ __forceinline__ StatsPtr stats_args_at(unsigned long long entry)
{
    asm volatile("" : "+s"(entry));
    return (StatsPtr)(entry + offsetof(MfmaDenseArgs, ks));
}

template <typename AT>
__device__ __forceinline__ const float* mask_fix_record(AT& A, FixArgsPtr F, int i, int j)
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
    const bool edge_rec = (fix_on != 0) & (A.rim_in_kernel == 0);
    const bool lo = edge_rec & (d >= 0) & (d < K - 1);
    const bool hi = edge_rec & (d >= fix_hi_d0) & (d - fix_hi_d0 < fix_hi_w);
    const long long o_rows = (long long)(top ? i : fix_top + i - fix_bot0) * F->fix_width + x;
    const long long o_side = (long long)i * 2 * fix_side + (j < fix_side ? j : j - (A.ns - 2 * fix_side));
    const long long o_lo = (long long)i * (K - 1) + d;
    const long long o_hi = (long long)i * fix_hi_w + (d - fix_hi_d0);
    const float* base = (top | bot) ? F->fix_rows : side ? fix_cols : lo ? F->fix_lo : F->fix_hi;
    const long long off = (top | bot) ? o_rows : side ? o_side : lo ? o_lo : o_hi;
    return (in_range & (top | bot | side | lo | hi)) ? base + 4 * off : nullptr;
}

// The correction {d n_missing, d ka, d kb} of a pixel on an edge diagonal, formed from the rim tables (cs_launch.h
// MfmaWeights::rim) instead of read from a record of cs_mask_prep.hip mask_edge_fix -- the same arithmetic.  The window
// of pixel (i, i + D) reaches below the main diagonal (D < K - 1: those pixels are flagged stripes in the reference,
// row | column flags in the regular model) or beyond max_dist (never missing in the reference): row ki of the window has
// its first L = clamp(ki - D) pixels below the diagonal and its pixels from H = clamp(md - D + ki + 1) on beyond max_dist.
// rbits / cbits: flags of the window's rows / columns.  Half of the windows have none and cost three loads.
__device__ __forceinline__ void rim_correction(const float* __restrict__ T, int K, int D, int md, unsigned rbits, unsigned cbits,
                                               float& fn, float& fa, float& fb)
{
    fn = fa = fb = 0.0f;
    if (D < K - 1) {
        fn = T[kRimBase + D];
        fa = T[kRimBase + 17 + D];
        fb = T[kRimBase + 34 + D];
    }
    for (unsigned rb = rbits; rb; rb &= rb - 1) {
        const int ki = __builtin_ctz(rb);
        const int L = min(K, max(0, ki - D));
        const int H = min(K, max(0, md - D + ki + 1));
        const float* pa = T + kRimPW + ki * 18;
        const float* pb = pa + 17 * 18;
        fn -= (float)(L + (K - H));
        fa -= pa[L] + (pa[K] - pa[H]);
        fb -= pb[L] + (pb[K] - pb[H]);
    }
    for (unsigned cb = cbits; cb; cb &= cb - 1) {
        const int kj = __builtin_ctz(cb);
        const int lo_k = min(K, max(0, kj + D + 1));
        const int hi_k = min(K, max(0, kj - (md - D)));
        const float* qa = T + kRimQW + kj * 18;
        const float* qb = qa + 17 * 18;
        float nn = (float)((K - lo_k) + hi_k);
        float a = (qa[K] - qa[lo_k]) + qa[hi_k];
        float b = (qb[K] - qb[lo_k]) + qb[hi_k];
        for (unsigned rb = rbits; rb; rb &= rb - 1) {
            const int ki = __builtin_ctz(rb);
            if (ki >= lo_k || ki < hi_k) {
                nn -= 1.0f;
                a -= T[kRimW + ki * 17 + kj];
                b -= T[kRimW + 289 + ki * 17 + kj];
            }
        }
        fn -= nn;
        fa -= a;
        fb -= b;
    }
}

// RSYM (REG only): 17 x 17 template whose rows mirror (row s == row 16 - s): 9 head fragments in registers
// instead of 17 -- the masked epilogue needs the 32 registers.
// CAND (REG instances only): candidate mode of cs_detect_foci compiled in (cs_device.h cand_screen_*); the plain REG
// instances compile it out -- the masked tile sits at 256 registers and seven more spilled ones cost 8 % of C4'.
// The unmasked instances decide at run time (they have room).
template <bool VEC4, bool REG, bool RSYM = false, bool CAND = false>
__global__ __launch_bounds__(256, 2) void corr_mfma_dense_kernel(const MfmaDenseArgs A)
{
    constexpr bool TABLE = false;
    constexpr unsigned long long table_entry = 0;
    constexpr int table_tile = 0, table_tile_end = 0, table_tile_step = 0;
    (void)table_entry; (void)table_tile; (void)table_tile_end; (void)table_tile_step;
#define MFD_NOMASK_KS A.ks
#include "cs_corr_mfma_body.inc"
#undef MFD_NOMASK_KS
}

// Several matrices (the banded blocks of a genome) in ONE persistent launch of the masked candidate instance: the tiles of
// all blocks form one list (first[b] .. first[b + 1] are block b's), XCD x takes the x-th eighth of it and its workgroups
// walk that range side by side; a workgroup drains its pipeline only where its range crosses into the next block (a few
// times per launch instead of once per block and launch).  The argument blocks live in a device table.
struct MfmaBlocksArgs {
    const MfmaDenseArgs* args;
    const int* first;
    int n_blocks, n_tiles;
    unsigned* started;       // (or null) set to `epoch` by the LAST workgroup when it starts: workgroups are dispatched in order, so
    unsigned epoch;          // every workgroup of the launch is resident then (cs_stream_wait_tiles)
};

typedef const __attribute__((address_space(4))) MfmaDenseArgs MfmaDenseArgsC;

template <bool VEC4, bool REG, bool RSYM, bool CAND, bool TABLE_ = true>
__device__ __forceinline__ void mfma_dense_tiles_table(MfmaDenseArgsC& A, const unsigned long long table_entry, const int table_tile,
                                                       const int table_tile_end, const int table_tile_step)
{
    constexpr bool TABLE = TABLE_;
    const KernelStats<float> no_ks{};       // (the unmasked epilogue is never part of a table instance: REG)
#define MFD_NOMASK_KS no_ks
#include "cs_corr_mfma_body.inc"
#undef MFD_NOMASK_KS
}

template <bool RSYM>
__global__ __launch_bounds__(256, 2) void corr_mfma_blocks_kernel(const MfmaBlocksArgs T)
{
    typedef const __attribute__((address_space(4))) int IntC;
    if (T.started && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(T.started, T.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int per = (T.n_tiles + 7) / 8;
    const int lo_x = x * per, hi_x = min(T.n_tiles, (x + 1) * per);
    IntC* first = (IntC*)(unsigned long long)T.first;
    bool ran = false;
    for (int b = 0; b < T.n_blocks; ++b) {
        const int s0 = first[b], s1 = first[b + 1];
        const int lo = max(lo_x, s0), hi = min(hi_x, s1);
        if (lo >= hi) continue;
        int g = lo_x + j;                              // this workgroup's sequence: lo_x + j + k step
        if (g < lo) g += (lo - g + step - 1) / step * step;
        if (g >= hi) continue;
        if (ran) lds_barrier();                        // the previous block's last epilogue has read its LDS tables
        ran = true;
        const unsigned long long entry = (unsigned long long)(T.args + b);
        mfma_dense_tiles_table<true, true, RSYM, true>(*(MfmaDenseArgsC*)entry, entry, g - s0, hi - s0, step);
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

// Column skew per tile row (0 .. 3) of a dense map's tile order (cs_corr_mfma_body.inc tile_origin): the smallest one under
// which a workgroup, stepping `step` tiles at a time through the row-major tile list, does not come back to the column of
// earlier tiles during its first eight steps (most distinct columns).  Measured (profiles/r05_dense_sizes.txt): a straight walk down one column --
// tiles_x = 64 with 64 workgroups per XCD -- costs 13 % of the map; 65 columns are fine as they are and would be hurt by a
// skew of 1 (64 + 65 = 129 = 2 x 65 - 1: the skewed walk is the straight one).
static int dense_tile_skew(int tiles_x, long long n_tiles, int step)
{
    if (tiles_x <= 2 || step <= 0) return 0;
    int best = 0, best_score = -1;
    for (int s = 0; s < 4; ++s) {
        int cols[8], n = 0, score = 0;
        long long t = 0;
        for (int k = 0; k < 8 && t < n_tiles; ++k, t += step) {
            const long long by = t / tiles_x;
            cols[n++] = (int)((t - by * tiles_x + by * s) % tiles_x);
        }
        for (int a = 0; a < n; ++a) {                       // distinct columns among the first eight tiles
            bool seen = false;
            for (int b = 0; b < a; ++b) seen |= cols[b] == cols[a];
            score += !seen;
        }
        if (score > best_score) {
            best_score = score;
            best = s;
        }
    }
    return best;
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
    if (A.cand_keys) {
        // candidate sink: only the scanned diagonals (diag_trim of the coefficient map) -- also for a dense layout, whose
        // range was just reset above (a 1-D pattern on a chromosome short enough to be staged dense: pixels of every
        // diagonal became candidates and joined the foci of the scanned ones)
        A.out_lo = std::max(A.out_lo, A.cand_dlo);
        A.out_hi = std::min(A.out_hi, A.cand_dhi);
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
    if (dense_f32 || reg_f32) {
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
        D.by_cut = A.tiles_y;
        if (reg_f32 && A.out.layout == 1) {
            // The strip of tiles follows the band: J0 = I0 + out_lo + 64 bx.  On the last rows of a matrix the band leaves it
            // (columns >= ns): tiles wholly beyond the last column are not enumerated -- the last 16 tile rows of a 1001-diagonal
            // band hold 120 of them, 23 blocks of a genome 2 760 of 53 000 tiles, each a full pass over zeros.
            auto valid = [&](int by) {
                const long long left = (long long)A.ns - ((long long)A.row_begin + (long long)by * MF_T + A.out_lo);
                const long long v = left <= 0 ? 0 : (left + MF_T - 1) / MF_T;
                return (int)std::min<long long>(v, A.tiles_x);
            };
            int cut = A.tiles_y;
            long long n = 0;
            while (cut > 0 && valid(cut - 1) < A.tiles_x) n += valid(--cut);
            D.by_cut = cut;
            D.n_tiles = (int)((long long)cut * A.tiles_x + n);
        }
        D.frag = E.frag;
        D.w_unscale = E.unscale[0];
        D.wa_unscale = E.unscale[1];
        D.wb_unscale = E.unscale[2];
#ifdef CS_MF_DEBUG
        D.dbg = getenv("CHROMOSIGHT_HIP_MFMA_DBG") ? atoi(getenv("CHROMOSIGHT_HIP_MFMA_DBG")) : 0;
#else
        D.dbg = 0;
#endif
        D.xcd_order = 1;
        D.ks = A.ks;
        D.band_in = A.sig.layout == 1;
        D.lo_in = A.sig.band_lo;
        D.bw_in = A.sig.band_w;
        D.pad_in = (A.sig.layout == 1 && A.sig.pad && A.sig.ld >= (long long)A.sig.band_w + 4) ? 1 : 0;
        D.cnt_in = (A.sig.layout == 1 && A.sig.counts) ? 1 : 0;
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
        D.fx.rim = E.rim;
        D.fx.max_dist = A.max_dist;
        D.rim_in_kernel = A.rim_in_kernel;
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
            if (A.defer_args) {
                if (!cand) return -5;
                std::memcpy(A.defer_args, &D, sizeof(D));
                if (A.defer_rsym) *A.defer_rsym = rsym ? 1 : 0;
                return 0;
            }
            typedef void (*reg_kernel_t)(const MfmaDenseArgs);
            const reg_kernel_t kr = rsym ? (cand ? corr_mfma_dense_kernel<true, true, true, true> : corr_mfma_dense_kernel<true, true, true, false>)
                                         : (cand ? corr_mfma_dense_kernel<true, true, false, true> : corr_mfma_dense_kernel<true, true, false, false>);
            hipError_t e3 = allow_big_lds((const void*)kr);
            if (e3 != hipSuccess) return (int)e3;
            const int per_cu_r = 2;                // (LDS: two workgroups per CU)
            if (D.n_tiles <= 0) return 0;          // (every tile row beyond the last column)
            int grid_r = (int)std::min<long long>(D.n_tiles, (long long)per_cu_r * A.n_cu);
            if (A.grid_cap > 0 && grid_r > A.grid_cap) grid_r = std::max(8, A.grid_cap & ~7);     // (multiples of 8: XCD-contiguous ranges)
            hipLaunchKernelGGL(kr, dim3((unsigned)grid_r), dim3(256), MFD_SMEM_REG + MFD_LAUNCH_EXTRA, stream, D);
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
        const int per_cu = 2;
        const int grid = (int)std::min<long long>(blocks, (long long)per_cu * A.n_cu);
        D.xcd_order |= dense_tile_skew(A.tiles_x, blocks, (D.xcd_order & 1) && grid % 8 == 0 ? grid / 8 : grid) << 1;
        if (vec4) hipLaunchKernelGGL((corr_mfma_dense_kernel<true, false>), dim3((unsigned)grid), dim3(256), (MFD_LAUNCH_EXTRA ? MFD_SMEM_REG + MFD_LAUNCH_EXTRA : MFD_SMEM), stream, D);
        else hipLaunchKernelGGL((corr_mfma_dense_kernel<false, false>), dim3((unsigned)grid), dim3(256), (MFD_LAUNCH_EXTRA ? MFD_SMEM_REG + MFD_LAUNCH_EXTRA : MFD_SMEM), stream, D);
        return (int)hipGetLastError();
    }
    if (A.sig.counts) return -6;                 // (a band of counts: only the masked tile kernel detrends what it reads)
    if (!A.out.ptr) return -5;                   // (a candidate sink without a map: only the masked tile kernel serves that)
    const void* kern = masked ? (const void*)corr_mfma_kernel<true> : (const void*)corr_mfma_kernel<false>;
    hipError_t e = allow_big_lds(kern);
    if (e != hipSuccess) return (int)e;
    if (masked) hipLaunchKernelGGL(corr_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    else hipLaunchKernelGGL(corr_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    return (int)hipGetLastError();
}

size_t mfma_blocks_arg_bytes() { return sizeof(MfmaDenseArgs); }
size_t mfma_blocks_arg_offset(int n_blocks) { return ((size_t)(n_blocks + 1) * sizeof(int) + 255) & ~(size_t)255; }
size_t mfma_blocks_table_bytes(int n_blocks) { return mfma_blocks_arg_offset(n_blocks) + (size_t)n_blocks * sizeof(MfmaDenseArgs); }

int launch_corr_mfma_prepared(const void* h_arg, int rsym, int n_cu, int grid_cap, hipStream_t stream)
{
    MfmaDenseArgs D;
    std::memcpy(&D, h_arg, sizeof(D));
    if (D.n_tiles <= 0) return 0;
    typedef void (*reg_kernel_t)(const MfmaDenseArgs);
    const reg_kernel_t kr = rsym ? corr_mfma_dense_kernel<true, true, true, true> : corr_mfma_dense_kernel<true, true, false, true>;
    hipError_t e = allow_big_lds((const void*)kr);
    if (e != hipSuccess) return (int)e;
    const int per_cu = 2;
    int grid = (int)std::min<long long>(D.n_tiles, (long long)per_cu * n_cu);
    if (grid_cap > 0 && grid > grid_cap) grid = std::max(8, grid_cap & ~7);
    hipLaunchKernelGGL(kr, dim3((unsigned)grid), dim3(256), MFD_SMEM_REG + MFD_LAUNCH_EXTRA, stream, D);
    return (int)hipGetLastError();
}

// upload: the stream the argument table travels on (nullptr: `stream`, right before the launch).  A caller whose stream is still
// busy with other work -- the staging of the maps -- hands a side stream that is ordered before the launch by an event of its own:
// the copy is then done when the stream gets to the tile kernel instead of sitting between the staging and it.
// (do_upload / launch: a caller that uploads on a side stream makes two calls -- upload only, then launch only; the caller's own
// stream may be the null stream, so "no upload" is a flag and not a null handle)
// the table's tile ranges (first[]) from its argument blocks: what an upload by somebody else needs in place (-3: too many tiles)
int mfma_blocks_table_finish(void* h_table, int n_blocks)
{
    int* first = reinterpret_cast<int*>(h_table);
    const MfmaDenseArgs* args = reinterpret_cast<const MfmaDenseArgs*>((char*)h_table + mfma_blocks_arg_offset(n_blocks));
    long long total = 0;
    for (int b = 0; b < n_blocks; ++b) {
        first[b] = (int)total;
        total += args[b].n_tiles;
        if (total > 0x7fffffffLL) return -3;
    }
    first[n_blocks] = (int)total;
    return 0;
}

int launch_corr_mfma_blocks(void* h_table, void* d_table, int n_blocks, int rsym, int n_cu, hipStream_t stream, hipStream_t upload,
                            bool do_upload, bool launch, unsigned* started, unsigned epoch)
{
    if (n_blocks <= 0) return 0;
    int* first = reinterpret_cast<int*>(h_table);
    const MfmaDenseArgs* args = reinterpret_cast<const MfmaDenseArgs*>((char*)h_table + mfma_blocks_arg_offset(n_blocks));
    long long total = 0;
    for (int b = 0; b < n_blocks; ++b) {
        first[b] = (int)total;
        total += args[b].n_tiles;
        if (total > 0x7fffffffLL) return -3;
    }
    first[n_blocks] = (int)total;
    if (total == 0) return 0;
    hipError_t e = hipSuccess;
    if (do_upload) e = hipMemcpyAsync(d_table, h_table, mfma_blocks_table_bytes(n_blocks), hipMemcpyHostToDevice, upload);
    if (e != hipSuccess) return (int)e;
    if (!launch) return 0;
    MfmaBlocksArgs T;
    T.first = reinterpret_cast<const int*>(d_table);
    T.args = reinterpret_cast<const MfmaDenseArgs*>((char*)d_table + mfma_blocks_arg_offset(n_blocks));
    T.n_blocks = n_blocks;
    T.n_tiles = (int)total;
    T.started = started;
    T.epoch = epoch;
    const void* kr = rsym ? (const void*)corr_mfma_blocks_kernel<true> : (const void*)corr_mfma_blocks_kernel<false>;
    e = allow_big_lds(kr);
    if (e != hipSuccess) return (int)e;
    const int per_cu = 2;
    const int grid = std::max(8, (int)std::min<long long>((total + 7) / 8 * 8, (long long)per_cu * n_cu) & ~7);
    if (rsym) hipLaunchKernelGGL(corr_mfma_blocks_kernel<true>, dim3((unsigned)grid), dim3(256), MFD_SMEM_REG + MFD_LAUNCH_EXTRA, stream, T);
    else hipLaunchKernelGGL(corr_mfma_blocks_kernel<false>, dim3((unsigned)grid), dim3(256), MFD_SMEM_REG + MFD_LAUNCH_EXTRA, stream, T);
    return (int)hipGetLastError();
}

}  // namespace cs
