// cs_corr_fast.hip -- explicit instantiations of the streaming Pearson kernel
// (cs_corr_stream.h) for one compile-time template size.  Compiled once per size with
// -DCS_K=<odd K> so that the sizes build in parallel.
#include <cstdio>
#include <cstdlib>

#include "cs_corr_stream.h"
#include "cs_launch.h"

#ifndef CS_K
#error "compile with -DCS_K=<odd kernel size>"
#endif

namespace cs {

// Two-height tiling of a dense launch that is a single generation of two waves per SIMD (see
// launch_fast): heights 5/4 h and 3/4 h for the first / second half of the row blocks.  Applies only
// if the re-tiled launch still fits that single generation.
static bool split_heights(int ms, long long strips_x, int h, long long n_simd, int* h1, int* h2, int* pairs)
{
    if (n_simd <= 0 || h < 32) return false;
    const long long waves = strips_x * ((ms + h - 1) / h);
    if (waves <= n_simd || waves > 2 * n_simd) return false;
    *h1 = (h * 5 / 4 + 1) & ~1;
    *h2 = 2 * h - *h1;
    *pairs = (ms + *h1 + *h2 - 1) / (*h1 + *h2);
    return 2LL * *pairs * strips_x <= 2 * n_simd;
}

template <typename TC>
static int launch_fast(const CorrArgs<TC>& A, hipStream_t stream)
{
    using G = StreamGeom<CS_K>;
    StreamArgs<TC> S;
    S.sig = A.sig.ptr;
    S.out = A.out.ptr;
    S.nobs = (float*)A.nobs.ptr;
    S.xcorr_only = A.xcorr_only;
    S.w = (unsigned long long)(uintptr_t)A.w;
    S.ld_in = A.sig.ld;
    S.ld_out = A.out.ld;
    S.sig_is_f64 = A.sig_is_f64;
    S.out_is_f64 = A.out_is_f64;
    S.band_in = A.sig.layout == 1;
    S.lo_in = A.sig.band_lo;
    S.bw_in = A.sig.band_w;
    S.band_out = A.out.layout == 1;
    S.out_lo = A.out_lo;
    S.out_hi = A.out_hi;
    S.lo_out = A.out.band_lo;
    S.ms = A.ms;
    S.ns = A.ns;
    S.full = A.full;
    S.sym_upper = A.sym_upper;
    S.strip_h = A.tile_h;
    S.strips_x = A.tiles_x;
    S.strips_y = A.tiles_y;
    S.split_sy = 0;
    S.row_begin = A.row_begin;
    S.row_end = A.row_end;
    S.row0_in = A.sig.row0;
    S.row0_out = A.out.row0;
    S.xcd_remap = S.band_out ? 1 : 0;
    S.strip_h2 = A.tile_h;
    // Single generation of two waves per SIMD (e.g. dense 4096^2 on 256 CUs): the arbiter favours
    // the wave that started first.  Measured with per-wave clocks on C2 (uniform 64-row strips):
    // on every SIMD the first wave finished after 214k cycles, the second after 290k, alone for the
    // last quarter at ~60 % of the two-wave VALU rate.  Workgroups are placed in launch order, so
    // giving the first half of the row blocks 5/4 and the second half 3/4 of the height makes each
    // pair finish together: C2 110 -> 116 Gpixel/s (the reverse split loses 6 %, as it must).
    {
        const long long n_simd = (long long)(A.n_cu > 0 ? A.n_cu : 0) * 4;
        const char* e = getenv("CHROMOSIGHT_HIP_SPLIT");   // "h1,h2" forces a split, "0" disables it
        int h1 = 0, h2 = 0, auto_pairs = 0;
        if (e && sscanf(e, "%d,%d", &h1, &h2) == 2) {
        } else if (!e && !S.band_out && !split_heights(A.row_end - A.row_begin, A.tiles_x, A.tile_h, n_simd, &h1, &h2, &auto_pairs)) {
            h1 = h2 = 0;
        }
        if (h1 > 0 && h2 > 0 && !S.band_out) {
            const int pairs = (A.row_end - A.row_begin + h1 + h2 - 1) / (h1 + h2);
            S.strip_h = h1;
            S.strip_h2 = h2;
            S.split_sy = pairs;
            S.strips_y = 2 * pairs;
        }
    }
    S.ks = A.ks;
    S.mask_mode = A.mask_mode;
    S.max_dist = A.max_dist;
    S.miss_row = A.miss_row;
    S.miss_col = A.miss_col;
    S.mask = (const uint8_t*)A.mask.ptr;
    S.fix_rows = A.fix_rows;
    S.fix_cols = A.fix_cols;
    S.fix_top = A.fix_top;
    S.fix_bot0 = A.fix_bot0;
    S.fix_width = A.fix_width;
    S.fix_xband = A.fix_xband;
    S.fix_xlo = A.fix_xlo;
    S.fix_side = A.fix_side;
    S.fix_on = A.fix_on;
    S.fix_hi_w = A.fix_hi_w;
    S.fix_hi_d0 = A.fix_hi_d0;
    S.rowtab = A.rowtab;
    S.coltab = A.coltab;
    S.fix_lo = A.fix_lo;
    S.fix_hi = A.fix_hi;
    const int n_waves = S.strips_x * S.strips_y;
    const int blocks = (n_waves + G::NWAVES - 1) / G::NWAVES;
    // A launch of at most two workgroups per CU is a single generation of waves: whatever the
    // dispatcher packs three-deep on one CU leaves another CU short and sets the kernel time.
    // Asking for more LDS than a third (half) of the 160 KB caps the residency at 2 (1) workgroups
    // per CU, which spreads such launches evenly.
    auto launch = [&](auto kern, size_t smem) -> int {
        size_t smem_req = smem;
        if (A.n_cu > 0) {
            const int per_cu = (blocks + A.n_cu - 1) / A.n_cu;
            if (per_cu <= 2) {
                const size_t cap = (size_t)160 * 1024 / (per_cu + 1) + 1024;
                if (cap > smem_req) smem_req = cap;
            }
        }
        if (smem_req > 48 * 1024) {
            // process-wide property of the kernel: always the hardware maximum, so that launches from
            // several host threads (pipeline._Workers) cannot undercut each other
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::NWAVES * kWave), smem_req, stream, S);
        return (int)hipGetLastError();
    };
    constexpr size_t smem0 = corr_stream_smem_bytes<CS_K, TC, 0>(), smem1 = corr_stream_smem_bytes<CS_K, TC, 1>(),
                     smem2 = corr_stream_smem_bytes<CS_K, TC, 2>();
    // vertically symmetric templates (float32 kernels only): folded template rows, see steps2_rec
    if constexpr (sizeof(TC) == 4) {
        if (A.w_sym) {
            // (the general masked kernel has no registers left for the shared row products)
            // short strips: the variant that skips the row products of the halo rows (steps2_rec)
            const bool skip = S.strip_h <= 96;
            if (A.mask_mode != 0 && A.reg_mode)
                return skip ? launch(corr_stream_kernel<TC, CS_K, 2, true, true>, smem2)
                            : launch(corr_stream_kernel<TC, CS_K, 2, true, false>, smem2);
            if (A.mask_mode == 0)
                return skip ? launch(corr_stream_kernel<TC, CS_K, 0, true, true>, smem0)
                            : launch(corr_stream_kernel<TC, CS_K, 0, true, false>, smem0);
        }
    }
    // per-bin mask: factorised mask sums (tables built by cs_api.cpp prepare_regular_mask)
    if (A.mask_mode != 0 && A.reg_mode) return launch(corr_stream_kernel<TC, CS_K, 2, false>, smem2);
    if (A.mask_mode != 0) return launch(corr_stream_kernel<TC, CS_K, 1, false>, smem1);
    return launch(corr_stream_kernel<TC, CS_K, 0, false>, smem0);
}

#if defined(CS_PROFILE) && CS_K == 17
// diagnostics build only: read and clear the section timers of cs_corr_stream.h
extern "C" int cs_debug_profile(unsigned long long* out)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(cs_prof), sizeof(cs_prof));
    if (e != hipSuccess) return (int)e;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(cs_prof), zero, sizeof(zero));
}
#endif

#define CS_CAT_(a, b) a##b
#define CS_CAT(a, b) CS_CAT_(a, b)

int CS_CAT(launch_corr_fast_f32_k, CS_K)(const CorrArgs<float>& A, hipStream_t s) { return launch_fast<float>(A, s); }
int CS_CAT(launch_corr_fast_f64_k, CS_K)(const CorrArgs<double>& A, hipStream_t s) { return launch_fast<double>(A, s); }

// strip geometry: 128 columns per wave, strip height chosen by a small cost model fitted to
// measurements on MI355X (height sweeps in profiles/r01_strip_height_sweep.txt):
//   time per SIMD ~ c * (2 * floor(w / 2) / e2 + (w odd ? 1 / 0.55 : 0))  for w <= 3,  c * w beyond;  c = h + 0.7 (K-1)
// w = strips per SIMD = ceil(strips / (4 n_cu)).  A strip stages h + K-1 rows (the K-1 warm-up rows
// emit nothing and cost ~0.7 of a row).  Two waves are resident per SIMD (registers, LDS) and share
// its VALU; a wave that is alone reaches ~55 % of the rate of a pair, so an odd w pays for one lone
// round (6144^2: 150-row strips, w = 2, beat 96-row strips, w = 3, by 8 %).  e2 = 1 when the
// two-height tiling of launch_fast applies (dense, single generation), else 0.95.  For band outputs
// the number of 128-column strips per row block jumps whenever band_w + h crosses a multiple of 128,
// which is what makes e.g. h = 20 better than h = 32 for a 234-diagonal band.
// band_w = number of output diagonals (0 for dense outputs).
void CS_CAT(corr_fast_tile_k, CS_K)(int ms, int ns, int band_w, int n_cu, int* tw, int* th)
{
    *tw = StreamGeom<CS_K>::TW;
    const long long max_sx = (ns + *tw - 1) / *tw;
    const long long n_simd = (long long)(n_cu > 0 ? n_cu : 256) * 4;
    int h = 32;
    double best = 1e300;
    for (int hh = 8; hh <= 256; hh += 2) {
        long long sx = max_sx;
        if (band_w > 0) {
            sx = (band_w - 1 + hh + *tw - 1) / *tw;   // same span as fill_grid (cs_api.cpp)
            if (sx > max_sx) sx = max_sx;
            if (sx < 1) sx = 1;
        }
        const long long waves = sx * ((ms + hh - 1) / hh);
        const long long w = (waves + n_simd - 1) / n_simd;
        int h1, h2, pairs;
        const double e2 = (band_w == 0 && split_heights(ms, sx, hh, n_simd, &h1, &h2, &pairs)) ? 1.0 : 0.95;
        // from four strips per SIMD on, finished waves are replaced continuously and the parity of w
        // stops mattering
        const double rounds = w >= 4 ? (double)w : 2.0 * (double)(w / 2) / e2 + ((w & 1) ? 1.0 / 0.55 : 0.0);
        const double t = (hh + 0.7 * (CS_K - 1)) * rounds;
        if (t <= best) {   // ties: the taller strip (fewer halo rows)
            best = t;
            h = hh;
        }
    }
    if (const char* e = getenv("CHROMOSIGHT_HIP_STRIP_H")) {
        const int v = atoi(e);
        if (v >= 1) h = v;
    }
    *th = h;
}

}  // namespace cs
