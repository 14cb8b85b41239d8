// cs_corr_sep.hip -- separable evaluation for templates that are exactly an outer product, K'[a][b] = u[a] v[b].
//
// The reference has a factorised branch for --tsvd (detection.py:648-665, preprocessing.py:810-847); of the built-in
// templates the two 31 x 31 stripes templates are rank 1 as they stand -- and too large for the unrolled / matrix-core
// kernels, so the runtime-size kernel used to evaluate their 961 products (and, with a mask, the 2 x 961 mask-weighted
// template sums) per pixel.  With K' = u v^T every window sum the coefficient needs separates:
//     sum S K'              = sum_a u[a]    (sum_b v[b]   S[i+a][j+b])
//     sum S, sum S^2, sum M = sum_a          (sum_b  ...           )            (box sums)
//     sum M K', sum M K'^2  = sum_a u[a]^k  (sum_b v[b]^k M[i+a][j+b]),  k = 1, 2
// and the centred weight sets of build_args (cs_api.cpp: Wc = Wa = K' - mean, Wb = (K' - mean)^2) follow from these
// and the box sums.  A horizontal pass over the staged rows (6 sums per staged pixel, 3 without a mask) leaves its
// results in LDS, a vertical pass over km rows finishes them: 2 * 6 * K multiply-adds per pixel instead of 6 * K^2.
// Staging, mask predicate, epilogue and stores are those of the runtime-size kernel (cs_corr_generic.hip), so
// edges, layouts and mask modes behave identically; only the order of the float32 additions differs.
#include "cs_device.h"
#include "cs_launch.h"

namespace cs {

constexpr int SEP_TW = 64;   // output columns per block (= lanes per wave)
constexpr int SEP_RG = 4;    // output rows per lane
constexpr int SEP_NW = 8;    // waves per block
constexpr int SEP_TH = SEP_RG * SEP_NW;

static size_t sep_smem(int km, int kn, bool masked)
{
    const size_t LH = SEP_TH + km - 1, LWP = (size_t)((SEP_TW + kn - 1 + 3) & ~3);
    const size_t nq = 3;
    (void)masked;
    return 4 * LH * LWP + ((LH * LWP + 15) & ~(size_t)15) + 4 * nq * LH * SEP_TW + 4 * (size_t)(km + kn) + (LH + LWP) + 64;
}

template <bool MASKED>
__global__ __launch_bounds__(512) void corr_sep_kernel(const CorrArgs<float> A)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int km = A.km, kn = A.kn, kk = km * kn;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
    const int LH = SEP_TH + km - 1;
    const int LW = SEP_TW + kn - 1;
    const int LWP = (LW + 3) & ~3;
    constexpr int NQ = 3;                   // planes of intermediate sums (signal, then mask)
    float* sS = reinterpret_cast<float*>(smem_raw);
    uint8_t* sM = reinterpret_cast<uint8_t*>(sS + (size_t)LH * LWP);
    float* H = reinterpret_cast<float*>(sM + (((size_t)LH * LWP + 15) & ~(size_t)15));      // [NQ][LH][SEP_TW]
    float* su = H + (size_t)NQ * LH * SEP_TW;
    float* sv = su + km;

    int i0, j0;
    if (!tile_origin(A, blockIdx.x, blockIdx.y, &i0, &j0)) return;
    const int tid = threadIdx.x;

    for (int idx = tid; idx < km + kn; idx += 512) su[idx] = A.w[3 * kk + idx];             // u, then v
    // per-bin flags of the staged rows and columns first (two dependent byte loads per staged pixel made the staging
    // the longest phase of a tile: one workgroup per CU, nothing to overlap it with)
    const bool bins = MASKED && A.mask_mode == 1;
    uint8_t* rfl = reinterpret_cast<uint8_t*>(sv + kn);
    uint8_t* cfl = rfl + LH;
    if (bins) {
        for (int idx = tid; idx < LH + LW; idx += 512) {
            const bool is_row = idx < LH;
            const int x = is_row ? i0 - kh + idx : j0 - kw + (idx - LH);
            const int n = is_row ? A.ms : A.ns;
            const uint8_t* src = is_row ? A.miss_row : A.miss_col;
            rfl[idx] = (x >= 0 && x < n) ? src[x] : 0;
        }
        __syncthreads();
    }
#pragma unroll 4
    for (int idx = tid; idx < LH * LWP; idx += 512) {
        const int tr = idx / LWP;
        const int tc = idx - tr * LWP;
        const int p = i0 - kh + tr;
        const int q = j0 - kw + tc;
        sS[idx] = load_signal(A, p, q);
        if (MASKED) {
            const bool m = bins ? missing_from_flags(A, p, q, rfl[tr] != 0, cfl[min(tc, LW - 1)] != 0, false, true) : missing_pred(A, p, q);
            sM[idx] = (tc < LW && m) ? 1 : 0;
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tr0 = wv * SEP_RG;            // vertical pass: wave wv owns output rows SEP_RG wv .. SEP_RG wv + SEP_RG - 1
    float a1[SEP_RG], s1[SEP_RG], s2[SEP_RG], nm[SEP_RG], b1[SEP_RG], b2[SEP_RG];
#pragma unroll
    for (int i = 0; i < SEP_RG; ++i) a1[i] = s1[i] = s2[i] = nm[i] = b1[i] = b2[i] = 0.0f;
    // Two rounds through the same three LDS planes -- the signal sums, then the mask sums: 78 KB instead of 125 KB
    // at 31 x 31, i.e. two workgroups per CU.
    float* H0 = H;
    float* H1 = H + (size_t)LH * SEP_TW;
    float* H2 = H + 2 * (size_t)LH * SEP_TW;
    // ---- signal: horizontal pass over every staged row (64 output columns), vertical pass over km rows
    // two staged rows per step share the read of v[b] (and double the independent accumulation chains)
    for (int r = wv; r < LH; r += 2 * SEP_NW) {
        const int r2 = min(r + SEP_NW, LH - 1);
        const float* row = sS + r * LWP + lane;
        const float* row2 = sS + r2 * LWP + lane;
        float hv = 0.0f, h1 = 0.0f, h2 = 0.0f, kv = 0.0f, k1 = 0.0f, k2 = 0.0f;
#pragma unroll 4
        for (int b = 0; b < kn; ++b) {
            const float x = row[b], y = row2[b], vb = sv[b];
            hv = fmaf(x, vb, hv);
            h1 += x;
            h2 = fmaf(x, x, h2);
            kv = fmaf(y, vb, kv);
            k1 += y;
            k2 = fmaf(y, y, k2);
        }
        H0[r * SEP_TW + lane] = hv;
        H1[r * SEP_TW + lane] = h1;
        H2[r * SEP_TW + lane] = h2;
        if (r + SEP_NW < LH) {
            H0[r2 * SEP_TW + lane] = kv;
            H1[r2 * SEP_TW + lane] = k1;
            H2[r2 * SEP_TW + lane] = k2;
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int a = 0; a < km; ++a) {
        const float ua = su[a];
#pragma unroll
        for (int i = 0; i < SEP_RG; ++i) {
            const int rr = tr0 + i + a;
            a1[i] = fmaf(ua, H0[rr * SEP_TW + lane], a1[i]);
            s1[i] += H1[rr * SEP_TW + lane];
            s2[i] += H2[rr * SEP_TW + lane];
        }
    }
    if (MASKED) {
        __syncthreads();
        // ---- mask: sum M, sum v M, sum v^2 M per staged row; then with 1, u, u^2 down the rows
        for (int r = wv; r < LH; r += 2 * SEP_NW) {
            const int r2 = min(r + SEP_NW, LH - 1);
            const uint8_t* mrow = sM + r * LWP + lane;
            const uint8_t* mrow2 = sM + r2 * LWP + lane;
            float g1 = 0.0f, gv = 0.0f, gw = 0.0f, q1 = 0.0f, qv = 0.0f, qw = 0.0f;
#pragma unroll 4
            for (int b = 0; b < kn; ++b) {
                const float m = (float)mrow[b], m2 = (float)mrow2[b], vb = sv[b];
                const float t = m * vb, t2 = m2 * vb;
                g1 += m;
                gv += t;
                gw = fmaf(t, vb, gw);
                q1 += m2;
                qv += t2;
                qw = fmaf(t2, vb, qw);
            }
            H0[r * SEP_TW + lane] = g1;
            H1[r * SEP_TW + lane] = gv;
            H2[r * SEP_TW + lane] = gw;
            if (r + SEP_NW < LH) {
                H0[r2 * SEP_TW + lane] = q1;
                H1[r2 * SEP_TW + lane] = qv;
                H2[r2 * SEP_TW + lane] = qw;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int a = 0; a < km; ++a) {
            const float ua = su[a], ua2 = ua * ua;
#pragma unroll
            for (int i = 0; i < SEP_RG; ++i) {
                const int rr = tr0 + i + a;
                nm[i] += H0[rr * SEP_TW + lane];
                b1[i] = fmaf(ua, H1[rr * SEP_TW + lane], b1[i]);
                b2[i] = fmaf(ua2, H2[rr * SEP_TW + lane], b2[i]);
            }
        }
    }

    const float mu = A.ks.kmean;
#pragma unroll
    for (int i = 0; i < SEP_RG; ++i) {
        const int oi = i0 + tr0 + i;
        const int oj = j0 + lane;
        if (oi >= A.row_end || oj >= A.ns) continue;
        const int d = oj - oi;
        if (d < A.out_lo || d > A.out_hi) continue;
        float r, nobs = A.ks.n;
        if (pixel_forced_zero(A, oi, oj)) {
            r = 0.0f;
        } else {
            // the centred sums the epilogue expects: Wc = Wa = K' - mean, Wb = (K' - mean)^2
            const float cs_ = fmaf(-mu, s1[i], a1[i]);
            const float ka = fmaf(-mu, nm[i], b1[i]);
            const float kb = fmaf(mu * mu, nm[i], fmaf(-2.0f * mu, b1[i], b2[i]));
            r = pearson_from_sums<float>(cs_, s1[i], s2[i], nm[i], ka, kb, A.ks, MASKED, &nobs);
            r = cand_upper_from_sums(r, cs_, s1[i], s2[i], nm[i], ka, kb, A.ks, MASKED);
        }
        store_pixel(A, oi, oj, r, nobs);
    }
}

// 0 on success, -3: the template needs more LDS than a CU has (the caller falls back to the runtime-size kernel)
int launch_corr_sep_f32(const CorrArgs<float>& A, hipStream_t stream)
{
    const bool masked = A.mask_mode != 0;
    const size_t smem = sep_smem(A.km, A.kn, masked);
    if (smem > 160 * 1024) return -3;
    const void* kern = masked ? (const void*)corr_sep_kernel<true> : (const void*)corr_sep_kernel<false>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(A.tiles_x, A.tiles_y), block(512);
    if (masked) hipLaunchKernelGGL(corr_sep_kernel<true>, grid, block, smem, stream, A);
    else hipLaunchKernelGGL(corr_sep_kernel<false>, grid, block, smem, stream, A);
    return (int)hipGetLastError();
}

bool corr_sep_fits(int km, int kn, bool masked) { return sep_smem(km, kn, masked) <= 160 * 1024; }

void corr_sep_tile(int* tw, int* th)
{
    *tw = SEP_TW;
    *th = SEP_TH;
}

}  // namespace cs
