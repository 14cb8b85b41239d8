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
#include "cs_launch.h"

namespace cs {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
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
                r = pearson_masked_lean(cs, s1, s2, nm, ka, kb, A.ks);
                nobs = A.ks.n - nm;
            } else {
                r = pearson_nomask_lean(cs, s1, s2, A.ks);
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

int launch_corr_mfma_f32(CorrArgs<float>& A, const MfmaWeights& E, hipStream_t stream)
{
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
    const void* kern = masked ? (const void*)corr_mfma_kernel<true> : (const void*)corr_mfma_kernel<false>;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (masked) hipLaunchKernelGGL(corr_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    else hipLaunchKernelGGL(corr_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), MF_SMEM, stream, A, E);
    return (int)hipGetLastError();
}

}  // namespace cs
