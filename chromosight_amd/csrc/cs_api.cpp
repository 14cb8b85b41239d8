// cs_api.cpp -- the C ABI (include/chromosight_hip.h), part 1: contexts, memory, streams and events; argument validation, template
// statistics, weight upload and kernel dispatch (build_args, launch_corr) for every entry.  The entries themselves:
// cs_api_entries.cpp (correlation, staging, host-side passes, call lists) and cs_api_foci.cpp (device foci, quantify); shared
// declarations: cs_api_internal.h.  No torch types, no retained pointers.
#include "cs_api_internal.h"

namespace csapi {

// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); a genome step runs five to seven streams
// side by side and an RCCL communicator adds its own: chains meant to overlap then share a queue (a rank's share of 8: 0.94 ms
// instead of 0.57 once ncclCommInitRank has run, tools/rccl_probe.py).  The runtime reads the variable when it initialises, so it
// is set when the library is loaded -- without overriding the caller's choice, and without effect where HIP is already up
// (such a host sets it itself).
static const int kHwQueuesDefault = (setenv("GPU_MAX_HW_QUEUES", "8", 0), 0);


// cs_stream_wait_tiles: one wave that sleeps until the word has reached `epoch` (the tile workgroups of the launch that carries this
// epoch are resident) and gives up after `ticks` of the constant-rate counter (100 MHz) -- the word is a scheduling hint, never a lock
__global__ void cs_wait_tiles_kernel(const unsigned* word, unsigned epoch, long long ticks)
{
    // (epochs only grow: a word left by an earlier launch never lets a later wait through; the difference is taken modulo 2^32)
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0 && wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
}

int fail(cs_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

cs::MatView view_of(const cs_matrix* m)
{
    cs::MatView v;
    v.ptr = m ? m->d_ptr : nullptr;
    v.ld = m ? m->ld : 0;
    v.layout = m ? ((m->layout == CS_LAYOUT_BAND_PADDED || m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW) ? CS_LAYOUT_BAND : m->layout) : 0;
    v.band_lo = m ? m->band_lo : 0;
    v.band_w = m ? m->band_w : 0;
    v.row0 = m ? m->row0 : 0;
    v.pad = (m && (m->layout == CS_LAYOUT_BAND_PADDED || m->layout == CS_LAYOUT_BAND_COUNTS)) ? 1 : 0;
    v.counts = (m && (m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW)) ? 1 : 0;
    return v;
}

AllowLazy::AllowLazy(cs_ctx* c_) : c(c_), was(c_ ? c_->allow_lazy : false), was_counts(c_ ? c_->allow_counts : false)
{
    if (c) c->allow_lazy = c->allow_counts = true;
}
AllowLazy::~AllowLazy()
{
    if (c) {
        c->allow_lazy = was;
        c->allow_counts = was_counts;
    }
}


struct HostStats {
    double n, kmean, kstd, kvar, ksum, k2sum;
};

// statistics of the exact template, float64, in the reference's order of operations
// (detection.py:1002-1003: kernel.mean(), kernel.std(); :1023-1026: sums and means)
HostStats template_stats(const double* k, int kk)
{
    HostStats s;
    s.n = (double)kk;
    double sum = 0, sum2 = 0;
    for (int t = 0; t < kk; ++t) {
        sum += k[t];
        sum2 += k[t] * k[t];
    }
    s.ksum = sum;
    s.k2sum = sum2;
    s.kmean = sum / kk;
    s.kvar = sum2 / kk - s.kmean * s.kmean;
    double dev = 0;
    for (int t = 0; t < kk; ++t) dev += (k[t] - s.kmean) * (k[t] - s.kmean);
    s.kstd = std::sqrt(dev / kk);
    return s;
}

template <typename TC>
int upload_weights(cs_ctx* ctx, hipStream_t stream, const std::vector<double>& w64)
{
    constexpr int slot = sizeof(TC) == 8 ? 1 : 0;
    std::vector<TC> w(w64.size());
    for (size_t t = 0; t < w64.size(); ++t) w[t] = (TC)w64[t];
    const size_t bytes = w.size() * sizeof(TC);
    if (ctx->w_cached[slot].size() == bytes && std::memcmp(ctx->w_cached[slot].data(), w.data(), bytes) == 0) return CS_OK;
    // a set used before?  swap it back in; otherwise the current set is parked in the slot used longest ago and that
    // slot's buffer takes the upload (work queued by earlier calls has drained: every entry point ends synchronised)
    {
        auto swap_in = [&](cs_ctx::ParkedWeights& pk) {
            std::swap(ctx->d_w[slot], pk.d);
            std::swap(ctx->d_w_bytes[slot], pk.bytes);
            ctx->w_cached[slot].swap(pk.host);
            pk.stamp = ++ctx->w_clock;
        };
        cs_ctx::ParkedWeights* oldest = &ctx->w_parked[slot][0];
        for (auto& pk : ctx->w_parked[slot]) {
            if (pk.d && pk.host.size() == bytes && std::memcmp(pk.host.data(), w.data(), bytes) == 0) {
                swap_in(pk);
                return CS_OK;
            }
            if (pk.stamp < oldest->stamp) oldest = &pk;
        }
        if (ctx->d_w[slot]) swap_in(*oldest);
    }
    // + 64 bytes of slack: the fast kernels' scalar row loads over-read (cs_corr_stream.h WRow)
    if (bytes + 64 > ctx->d_w_bytes[slot]) {
        if (ctx->d_w[slot]) {
            CS_HIP(ctx, hipDeviceSynchronize());   // a queued kernel may still read the old buffer
            CS_HIP(ctx, hipFree(ctx->d_w[slot]));
        }
        ctx->d_w[slot] = nullptr;
        ctx->d_w_bytes[slot] = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_w[slot], bytes + 64));
        CS_HIP(ctx, hipMemsetAsync(ctx->d_w[slot], 0, bytes + 64, stream));   // same stream as the upload below
        ctx->d_w_bytes[slot] = bytes + 64;
        ctx->w_cached[slot].clear();
    }
    if (ctx->w_cached[slot].size() == bytes && std::memcmp(ctx->w_cached[slot].data(), w.data(), bytes) == 0)
        return CS_OK;
    CS_HIP(ctx, hipMemcpyAsync(ctx->d_w[slot], w.data(), bytes, hipMemcpyHostToDevice, stream));
    ++ctx->uploads;
    // `w` is pageable and dies here: on a non-default stream the runtime may still be reading it after
    // the call returns, so wait (uploads only happen when the template changes)
    CS_HIP(ctx, hipStreamSynchronize(stream));
    ctx->w_cached[slot].assign((unsigned char*)w.data(), (unsigned char*)w.data() + bytes);
    return CS_OK;
}

// ---- float16 head / tail fragments of the weight sets for the matrix-core kernel -----------------
uint16_t f32_to_f16_bits(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t be = (x >> 23) & 0xffu;
    uint32_t m = x & 0x7fffffu;
    if (be == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
    const int e = (int)be - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t half = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;     // may carry into the exponent: still right
    return (uint16_t)(sign | half);
}

float f16_bits_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    float out;
    if (e == 0) {
        out = std::ldexp((float)m, -24);
    } else if (e == 31) {
        uint32_t x = 0x7f800000u | (m << 13);
        std::memcpy(&out, &x, 4);
    } else {
        out = std::ldexp((float)(m | 0x400u), e - 25);
    }
    uint32_t x;
    std::memcpy(&x, &out, 4);
    x |= sign;
    std::memcpy(&out, &x, 4);
    return out;
}

// Build (or reuse) the fragment image of the float32 weights currently in ctx->d_w[0] (layout in
// cs_launch.h MfmaWeights).  Each set is scaled by the power of two that puts its largest magnitude in
// [64, 128), so that heads and tails stay in float16's normal range.
int ensure_wfrag(cs_ctx* ctx, hipStream_t stream, int km, int kn, cs::MfmaWeights* E)
{
    const std::vector<unsigned char>& key = ctx->w_cached[0];
    const int kk = km * kn;
    const size_t n_floats = key.size() / 4;
    if (kk <= 0 || n_floats < (size_t)kk) return fail(ctx, CS_ERR_INVALID, "weights missing for the matrix-core kernel");
    if (km > 17 || kn > 17) return fail(ctx, CS_ERR_INVALID, "the matrix-core weight image holds templates of up to 17 x 17");
    const int nsets = (int)std::min<size_t>(3, n_floats / kk);
    constexpr size_t kImage = 3 * 17 * 2 * 1024;
    if (!ctx->d_wfrag) CS_HIP(ctx, hipMalloc(&ctx->d_wfrag, kImage));
    if (!(ctx->wfrag_km == km && ctx->wfrag_kn == kn && ctx->wfrag_key == key)) {
        const float* w = reinterpret_cast<const float*>(key.data());
        std::vector<uint16_t> img(kImage / 2, 0);
        for (int set = 0; set < nsets; ++set) {
            float amax = 0.0f;
            for (int t = 0; t < kk; ++t) amax = std::max(amax, std::fabs(w[set * kk + t]));
            int ew = 0;
            if (amax > 0.0f && std::isfinite(amax)) {
                int e2;
                (void)std::frexp(amax, &e2);          // amax = f * 2^e2, f in [0.5, 1)
                ew = 7 - e2;                           // amax * 2^ew in [64, 128)
            }
            ew = std::max(-100, std::min(100, ew));
            ctx->wfrag_unscale[set] = std::ldexp(1.0f, -ew);
            for (int s = 0; s < km; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int t = 8 * (lane >> 4) + e - (lane & 15);
                        if (t < 0 || t >= kn) continue;
                        const float v = std::ldexp(w[set * kk + s * kn + t], ew);
                        const uint16_t hb = f32_to_f16_bits(v);
                        const uint16_t lb = f32_to_f16_bits(v - f16_bits_to_f32(hb));
                        const size_t base = ((size_t)set * km + s) * 2 * 64 * 8;
                        img[base + (size_t)lane * 8 + e] = hb;
                        img[base + 64 * 8 + (size_t)lane * 8 + e] = lb;
                    }
        }
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_wfrag, img.data(), kImage, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        // rim tables of the two mask weight sets (cs_launch.h MfmaWeights::rim), square templates only
        std::vector<float> rim(cs::kRimFloats, 0.0f);
        if (nsets == 3 && km == kn) {
            const int K = km;
            for (int set = 0; set < 2; ++set) {
                const float* ws = w + (1 + set) * kk;
                for (int r = 0; r < K; ++r) {
                    double sp = 0.0, sq = 0.0;
                    for (int m = 0; m <= K; ++m) {
                        rim[cs::kRimPW + (set * 17 + r) * 18 + m] = (float)sp;
                        rim[cs::kRimQW + (set * 17 + r) * 18 + m] = (float)sq;
                        if (m < K) {
                            sp += ws[r * K + m];
                            sq += ws[m * K + r];
                        }
                    }
                    for (int c = 0; c < K; ++c) rim[cs::kRimW + set * 289 + r * 17 + c] = ws[r * K + c];
                }
            }
            for (int D = 0; D < K - 1; ++D) {          // the whole lower triangle of diagonal D: row ki has its first ki - D pixels in it
                double n = 0.0, a = 0.0, b = 0.0;
                for (int ki = 0; ki < K; ++ki) {
                    const int L = std::min(K, std::max(0, ki - D));
                    n += L;
                    for (int kj = 0; kj < L; ++kj) {
                        a += w[kk + ki * K + kj];
                        b += w[2 * kk + ki * K + kj];
                    }
                }
                rim[cs::kRimBase + 0 * 17 + D] = (float)n;
                rim[cs::kRimBase + 1 * 17 + D] = (float)a;
                rim[cs::kRimBase + 2 * 17 + D] = (float)b;
            }
        }
        if (!ctx->d_rim) CS_HIP(ctx, hipMalloc(&ctx->d_rim, sizeof(float) * cs::kRimFloats));
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_rim, rim.data(), sizeof(float) * cs::kRimFloats, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        CS_HIP(ctx, hipStreamSynchronize(stream));    // pageable sources die here
        ctx->wfrag_key = key;
        ctx->wfrag_km = km;
        ctx->wfrag_kn = kn;
    }
    E->frag = reinterpret_cast<const uint4*>(ctx->d_wfrag);
    E->rim = reinterpret_cast<const float*>(ctx->d_rim);
    for (int set = 0; set < 3; ++set) E->unscale[set] = ctx->wfrag_unscale[set];
    return CS_OK;
}

// The image of the two-pass kernel (cs_corr_wide.hip; layout in cs_launch.h MfmaWideWeights): three sets x km rows x
// two passes x {head, tail} x 1 KiB, then the 4 x 33 row / column sums of the two mask weight sets (float64 sums of the
// float32 weights the device holds, rounded once).
int ensure_wfrag_wide(cs_ctx* ctx, hipStream_t stream, int km, int kn, cs::MfmaWideWeights* E)
{
    const std::vector<unsigned char>& key = ctx->w_cached[0];
    const int kk = km * kn;
    const size_t n_floats = key.size() / 4;
    if (kk <= 0 || n_floats < (size_t)kk) return fail(ctx, CS_ERR_INVALID, "weights missing for the matrix-core kernel");
    if (!cs::corr_mfma_wide_fits(km, kn)) return fail(ctx, CS_ERR_INVALID, "the two-pass matrix-core kernel holds templates of up to 33 x 33");
    const int nsets = (int)std::min<size_t>(3, n_floats / kk);
    const size_t frag_halfs = (size_t)3 * km * 2 * 2 * 512;
    const size_t sums_off = frag_halfs * 2;                    // bytes (a multiple of 16)
    const size_t bytes = sums_off + 4 * 33 * sizeof(float);
    if (bytes > ctx->d_wfrag_wide_bytes) {
        if (ctx->d_wfrag_wide) {
            CS_HIP(ctx, hipDeviceSynchronize());   // a queued kernel may still read the old image
            CS_HIP(ctx, hipFree(ctx->d_wfrag_wide));
        }
        ctx->d_wfrag_wide = nullptr;
        ctx->d_wfrag_wide_bytes = 0;
        ctx->wfrag_wide_key.clear();
        CS_HIP(ctx, hipMalloc(&ctx->d_wfrag_wide, bytes));
        ctx->d_wfrag_wide_bytes = bytes;
    }
    if (!(ctx->wfrag_wide_km == km && ctx->wfrag_wide_kn == kn && ctx->wfrag_wide_key == key)) {
        const float* w = reinterpret_cast<const float*>(key.data());
        std::vector<unsigned char> img(bytes, 0);
        uint16_t* halfs = reinterpret_cast<uint16_t*>(img.data());
        float* sums = reinterpret_cast<float*>(img.data() + sums_off);
        for (int set = 0; set < nsets; ++set) {
            float amax = 0.0f;
            for (int t = 0; t < kk; ++t) amax = std::max(amax, std::fabs(w[set * kk + t]));
            int ew = 0;
            if (amax > 0.0f && std::isfinite(amax)) {
                int e2;
                (void)std::frexp(amax, &e2);          // amax = f * 2^e2, f in [0.5, 1)
                ew = 7 - e2;                           // amax * 2^ew in [64, 128)
            }
            ew = std::max(-100, std::min(100, ew));
            ctx->wfrag_wide_unscale[set] = std::ldexp(1.0f, -ew);
            for (int s = 0; s < km; ++s)
                for (int pass = 0; pass < 2; ++pass)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int k = 8 * (lane >> 4) + e;
                            const size_t base = ((((size_t)set * km + s) * 2 + pass) * 2) * 512;
                            const int t = 32 * pass + k - (lane & 15);
                            if (t >= 0 && t < kn) {
                                const float v = std::ldexp(w[set * kk + s * kn + t], ew);
                                const uint16_t hb = f32_to_f16_bits(v);
                                const uint16_t lb = f32_to_f16_bits(v - f16_bits_to_f32(hb));
                                halfs[base + (size_t)lane * 8 + e] = hb;
                                halfs[base + 512 + (size_t)lane * 8 + e] = lb;
                            }
                            // the second pass of the template itself (set 0): k = 16 .. 31 meet no weight (t >= 33), and the kernel
                            // sends the signal's TAILS of the columns of k - 16 through them: a second copy of the heads
                            const int t2 = 32 + (k - 16) - (lane & 15);
                            if (set == 0 && pass == 1 && k >= 16 && t2 >= 0 && t2 < kn)
                                halfs[base + (size_t)lane * 8 + e] = f32_to_f16_bits(std::ldexp(w[s * kn + t2], ew));
                        }
        }
        if (nsets == 3)
            for (int set = 0; set < 2; ++set) {
                const float* ws = w + (1 + set) * kk;
                for (int s = 0; s < km; ++s) {
                    double acc = 0.0;
                    for (int t = 0; t < kn; ++t) acc += ws[s * kn + t];
                    sums[set * 33 + s] = (float)acc;
                }
                for (int t = 0; t < kn; ++t) {
                    double acc = 0.0;
                    for (int s = 0; s < km; ++s) acc += ws[s * kn + t];
                    sums[(2 + set) * 33 + t] = (float)acc;
                }
            }
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_wfrag_wide, img.data(), bytes, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        CS_HIP(ctx, hipStreamSynchronize(stream));    // the pageable source dies here
        ctx->wfrag_wide_key = key;
        ctx->wfrag_wide_km = km;
        ctx->wfrag_wide_kn = kn;
    }
    E->frag = reinterpret_cast<const uint4*>(ctx->d_wfrag_wide);
    E->sums = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ctx->d_wfrag_wide) + sums_off);
    for (int set = 0; set < 3; ++set) E->unscale[set] = ctx->wfrag_wide_unscale[set];
    E->plane_only = std::getenv("CHROMOSIGHT_HIP_WIDE_PLANE") ? 1 : 0;
    E->plane_only_staging = std::getenv("CHROMOSIGHT_HIP_WIDE_SLOW") ? 1 : 0;
    E->one_launch = std::getenv("CHROMOSIGHT_HIP_WIDE_ONE_LAUNCH") ? 1 : std::getenv("CHROMOSIGHT_HIP_WIDE_TWO_LAUNCHES") ? 2 : 0;
    E->tile_mode = 0;
    return CS_OK;
}

bool fast_available(int km, int kn, int* K);

// Which float32 calls take the two-pass matrix-core kernel (cs_corr_wide.hip): templates with a side of 18 .. 33 -- what
// `--win-size` makes (cli/chromosight.py:365-370) and the 19 x 19 .. 33 x 33 templates of API users -- in every container
// the runtime-size kernel served (bands and dense maps, float32 and float64, any mask, n_obs, plain cross-correlations).
// CHROMOSIGHT_HIP_NO_MFMA=1 / CHROMOSIGHT_HIP_NO_WIDE=1: never (the runtime-size kernel: the in-library cross-check).
bool mfma_wide_wanted(const cs::CorrArgs<float>& A)
{
    // (templates of up to 17 x 17 have their own instances; CHROMOSIGHT_HIP_WIDE_ALL=1 sends them here too: a measurement switch)
    if (!cs::corr_mfma_wide_fits(A.km, A.kn) || (A.km <= 17 && A.kn <= 17 && !std::getenv("CHROMOSIGHT_HIP_WIDE_ALL"))) return false;
    if (A.sig.counts || A.sig.layout == CS_LAYOUT_BAND_LAZY) return false;
    if (!A.out.ptr && !(A.cand_keys && A.cand_count && !A.defer_args && A.ks.cand_cmin > 0.0f)) return false;     // a map, or a candidate sink
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA") || std::getenv("CHROMOSIGHT_HIP_NO_WIDE")) return false;
    return true;
}

// Which float32 calls go to the matrix cores (cs_corr_mfma.hip).  Default: unmasked dense float32 maps
// (cs_normxcorr2 without a mask: the API / benchmark configuration) whenever the template is large
// (>= 13 x 13 entries: every template costs a full 17-row pass there, so small ones are cheaper on the
// packed-FMA kernel -- measured 4096^2: 17x17 0.115 vs 0.185 ms (no symmetry), 13x13 0.124 vs 0.127,
// 11x11 0.129 vs 0.108), the map is small (the strips of the streaming kernel under-fill the chip:
// 2048^2 11x11 0.035 vs 0.042 ms, 1024^2 9x9 0.014 vs 0.019) or the template has no streaming kernel
// (rectangular sizes).  CHROMOSIGHT_HIP_MFMA=1: every call with a template of up to 17 x 17
// (masked / banded maps run the general, slower, matrix-core kernel -- a test switch);
// CHROMOSIGHT_HIP_NO_MFMA=1: never.  Read per call so that tests can flip them.
// per-bin masks (detect / quantify configuration): opt-in for now (CHROMOSIGHT_HIP_MFMA_REG=1)
bool mfma_reg_wanted(const cs::CorrArgs<float>& A)
{
    if (A.mask_mode != CS_MASK_BINS || !A.full || A.km != A.kn || A.km > 17 || A.km < 3 || !(A.km & 1)) return false;
    if (std::getenv("CHROMOSIGHT_HIP_WIDE_ALL")) return false;
    if (A.xcorr_only) return false;            // (float64 containers are narrowed row by row first, see launch_corr)
    if ((A.sig.layout == CS_LAYOUT_BAND ? A.sig.band_w : A.ns) < 4) return false;      // 16-byte staging pieces
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA")) return false;
    // default: templates of 15 x 15 and 17 x 17 -- measured on the 234- and 1001-diagonal bands (tools/time_templates.py,
    // profiles/*_template_kernels.txt): 17 x 17 mirrored rows 2.01 vs 2.56 ms, 17 x 17 general 2.27 vs 3.01, 15 x 15 2.24 vs
    // 2.72; at 13 x 13 it wins only on the wide band (2.29 vs 2.51, 0.251 vs 0.235 on the narrow one), below that the
    // streaming kernel does (the tile kernel always walks 17 template rows).  CHROMOSIGHT_HIP_MFMA_REG=1 sends every
    // compatible call here, =0 none
    const char* e = std::getenv("CHROMOSIGHT_HIP_MFMA_REG");
    if (e && e[0] == '1') return true;
    if (e && e[0] == '0') return false;
    const char* general = std::getenv("CHROMOSIGHT_HIP_MFMA");
    if (general && general[0] == '1') return false;      // the general matrix-core kernel was asked for by name
    return A.km >= 15;
}

bool mfma_wanted(const cs::CorrArgs<float>& A)
{
    if (A.km < 1 || A.kn < 1 || A.km > 17 || A.kn > 17) return false;
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA") || std::getenv("CHROMOSIGHT_HIP_WIDE_ALL")) return false;
    const char* e = std::getenv("CHROMOSIGHT_HIP_MFMA");
    if (e && e[0] == '1') return true;
    const bool dense_f32 = A.mask_mode == 0 && A.sig.layout == 0 && A.out.layout == 0 && !A.nobs.ptr;
    int K = 0;
    const long long px = (long long)(A.row_end - A.row_begin) * A.ns;
    return dense_f32 && (A.km * A.kn >= 169 || px <= 6000000 || !fast_available(A.km, A.kn, &K));
}

bool fast_available(int km, int kn, int* K)
{
    (void)km;
    (void)kn;
    (void)K;
#ifdef CS_HAVE_FAST
    if (km != kn) return false;
    switch (km) {
        case 7: case 9: case 11: case 13: case 15: case 17:
            *K = km;
            return true;
        default:
            return false;
    }
#else
    return false;
#endif
}

// aligned_x: the generic kernel starts the x tiles of a row block at a multiple of the tile width,
// the streaming kernel at the block's first in-band column
template <typename TC>
void fill_grid(cs::CorrArgs<TC>& A, int tw, int th, bool aligned_x = true)
{
    A.tile_w = tw;
    A.tile_h = th;
    A.tiles_y = (A.row_end - A.row_begin + th - 1) / th;
    if (A.out.layout == CS_LAYOUT_BAND) {
        A.out_lo = A.out.band_lo;
        A.out_hi = A.out.band_lo + A.out.band_w - 1;
        const long long span = (long long)(A.out_hi - A.out_lo) + th + tw - 1;
        A.tiles_x = aligned_x ? (int)(span / tw) + 2 : (int)(span / tw);
        const int max_x = (A.ns + tw - 1) / tw;
        if (A.tiles_x > max_x) A.tiles_x = max_x;
    } else {
        A.out_lo = -(1 << 30);
        A.out_hi = (1 << 30);
        A.tiles_x = (A.ns + tw - 1) / tw;
    }
}

// Decide whether the factorised per-bin mask path applies and, if so, build its tables in the
// context's scratch buffer (cs_mask_prep.hip).  K = template size served by a streaming kernel.
template <typename TC>
int prepare_regular_mask(cs_ctx* ctx, cs::CorrArgs<TC>& A, int K, hipStream_t stream, bool rim_in_kernel = false, bool tile_reader = false)
{
    A.reg_mode = 0;
    A.fix_on = 0;
    A.rim_in_kernel = 0;
    if (std::getenv("CHROMOSIGHT_HIP_DEBUG"))
        fprintf(stderr, "[chromosight_hip] mask_mode=%d full=%d sym_upper=%d ms=%d ns=%d out_layout=%d out_lo=%d out_hi=%d max_dist=%d\n",
                A.mask_mode, A.full, A.sym_upper, A.ms, A.ns, A.out.layout, A.out_lo, A.out_hi, A.max_dist);
    if (A.mask_mode != CS_MASK_BINS || !A.full || std::getenv("CHROMOSIGHT_HIP_NO_REGULAR_MASK")) return CS_OK;
    if (A.ms < 2 * K || A.ns < 2 * K) return CS_OK;
    const int KH = (K - 1) / 2;
    const bool band_out = A.out.layout == CS_LAYOUT_BAND;
    int hi_d0 = 0, hi_w = 0, bot0, width, side = 0;
    bool edge_tables = false;
    if (A.sym_upper && !band_out) {
        // small dense maps (API users, short chromosomes): one correction per pixel, all from the
        // general predicate
        if ((long long)A.ms * A.ns > (1 << 22)) return CS_OK;
        bot0 = KH;
        width = A.ns;
    } else if (A.sym_upper) {
        // band outputs that end near max_dist: the edge tables cover the diagonals whose windows
        // leave 0..max_dist
        if (A.max_dist < 0 || A.out_lo < 0) return CS_OK;
        edge_tables = true;
        hi_d0 = A.max_dist - K + 2;
        hi_w = A.out_hi - hi_d0 + 1;
        if (hi_w < 0) hi_w = 0;
        if (hi_w > 64) return CS_OK;
        // rows whose in-band pixels reach the right or the bottom frame
        bot0 = std::max(KH, std::min(A.ms - KH, A.ns - KH - A.out_hi));
        width = A.out.band_w;
    } else {
        if (band_out) return CS_OK;
        bot0 = A.ms - KH;
        width = A.ns;
        side = KH;
    }
    const int top = KH;
    auto align = [](size_t n) { return (n + 63) & ~(size_t)63; };   // elements
    const size_t n_row = align((size_t)A.ms * 4 + 64), n_col = align(3 * (size_t)A.ns);
    // (rim_in_kernel: the masked matrix-core tile kernel forms the corrections of the edge diagonals itself)
    // ... when the two edge ranges hold at most 16 diagonals each and lie more than the 79 diagonals of a wave's
    // 16 rows x 64 columns apart (its epilogue handles one range per wave); narrow bands keep the records
    const bool edge_records = edge_tables && !(rim_in_kernel && K - 1 <= 16 && hi_w <= 16 && hi_d0 >= 96);
    const size_t n_lo = edge_records ? align((size_t)A.ms * (K - 1) * 4) : 0;
    const size_t n_hi = edge_records ? align((size_t)A.ms * hi_w * 4 + 4) : 0;
    const size_t n_frows = align((size_t)(top + A.ms - bot0) * width * 4);
    const size_t n_fcols = side ? align((size_t)A.ms * 2 * side * 4) : 0;
    const size_t bytes = (n_row + n_col + n_lo + n_hi + n_frows + n_fcols) * sizeof(TC);
    if (bytes > ctx->d_ws_bytes) {
        if (ctx->d_ws) CS_HIP(ctx, hipFree(ctx->d_ws));
        ctx->d_ws = nullptr;
        ctx->d_ws_bytes = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_ws, bytes));
        ctx->d_ws_bytes = bytes;
    }
    TC* rowtab = reinterpret_cast<TC*>(ctx->d_ws);
    TC* coltab = rowtab + n_row;
    TC* fix_lo = coltab + n_col;
    TC* fix_hi = fix_lo + n_lo;
    TC* fix_rows = fix_hi + n_hi;
    TC* fix_cols = side ? fix_rows + n_frows : nullptr;
    cs::MaskPrepArgs<TC> P;
    std::memset(&P, 0, sizeof(P));
    P.rr = A.miss_row;
    P.cc = A.miss_col;
    P.ms = A.ms;
    P.ns = A.ns;
    P.K = K;
    P.sym_upper = A.sym_upper;
    P.max_dist = A.max_dist;
    P.w = A.w;
    P.rowtab = rowtab;
    P.coltab = coltab;
    P.edge = edge_tables ? 1 : 0;
    P.skip_edge = edge_records ? 0 : 1;
    P.hi_d0 = hi_d0;
    P.hi_w = hi_w;
    P.fix_lo = fix_lo;
    P.fix_hi = fix_hi;
    P.top = top;
    P.bot0 = bot0;
    P.width = width;
    P.x_band = band_out ? 1 : 0;
    P.x_lo = A.out_lo;
    P.side = side;
    P.fix_rows = fix_rows;
    P.fix_cols = fix_cols;
    if (tile_reader && edge_tables && !edge_records && (A.row_begin > 0 || A.row_end < A.ms) && !std::getenv("CHROMOSIGHT_HIP_FULL_MASK_TABLES")) {
        // A row window read by the 64 x 64 tiles of the matrix-core kernel (a rank's share of a row-split block): the table entries
        // of its rows and of the columns its strip of tiles reaches (J0 + lane, J0 <= I0 + out_lo + 64 (tiles_x - 1)), and the frame
        // rows only when the window touches them -- the tables of all 200 000 bins cost 19 us in front of an eighth's 220 us of tiles
        P.r_lo = std::max(0, A.row_begin);
        P.r_hi = std::min(A.ms, A.row_end + 64);
        P.c_lo = std::max(0, A.row_begin + A.out_lo - 64);
        P.c_hi = (int)std::min<long long>(A.ns, (long long)A.row_end + A.out_hi + 320);
        P.skip_top = P.r_lo >= top ? 1 : 0;
        P.skip_bot = P.r_hi <= bot0 ? 1 : 0;
    }
    bool collected = false;
    if constexpr (std::is_same<TC, float>::value) {
        if (ctx->prep_collect) {             // (cs_detect_foci_blocks, one tile launch: the tables of all blocks from ONE launch)
            const int n_wg = cs::mask_prep_blocks<float>(P);
            if (n_wg < 0) return fail(ctx, CS_ERR_HIP, "mask tables: template too large");
            ctx->prep_collect->push_back(P);
            ctx->prep_groups.push_back(n_wg);
            collected = true;
        }
    }
    if (!collected && !ctx->skip_prep_launch) {       // (skip: after the call's own prepare form the tables are in place, only A is wanted)
        int rc = cs::launch_mask_prep<TC>(P, stream);
        if (rc != 0) return fail(ctx, CS_ERR_HIP, "mask table kernel failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (edge_tables) A.fix_on = 1;
    A.rim_in_kernel = (edge_tables && !edge_records) ? 1 : 0;
    A.reg_mode = 1;
    A.rowtab = rowtab;
    A.coltab = coltab;
    A.fix_lo = fix_lo;
    A.fix_hi = fix_hi;
    A.fix_hi_w = hi_w;
    A.fix_hi_d0 = hi_d0;
    A.fix_rows = fix_rows;
    A.fix_cols = fix_cols;
    A.fix_top = top;
    A.fix_bot0 = bot0;
    A.fix_width = width;
    A.fix_xband = band_out ? 1 : 0;
    A.fix_xlo = band_out ? A.out_lo : 0;
    A.fix_side = side;
    return CS_OK;
}

template <typename TC>
int launch_corr(cs_ctx* ctx, cs::CorrArgs<TC>& A, hipStream_t stream, bool allow_fast);

// internal status of launch_corr<float>: a candidate sink was given without a map, and the kernel that would serve the
// call writes maps (nothing was launched that matters: the caller allocates the map and calls again)

// the streaming kernels write n_obs next to the coefficient (same index), so both maps must share
// one geometry; plain cross-correlations run their unmasked instance
template <typename TC>
bool fast_compatible(const cs::CorrArgs<TC>& A)
{
    if (A.nobs.ptr && (A.nobs.layout != A.out.layout || A.nobs.ld != A.out.ld || A.nobs.band_lo != A.out.band_lo ||
                       A.nobs.band_w != A.out.band_w || A.nobs.row0 != A.out.row0))
        return false;
    if (A.xcorr_only && (A.mask_mode != 0 || A.full || A.sym_upper)) return false;
    return true;
}

template <>
int launch_corr<float>(cs_ctx* ctx, cs::CorrArgs<float>& A, hipStream_t stream, bool allow_fast)
{
    int K = 0, tw, th, rc;
    A.n_cu = ctx->n_cu;
    A.grid_cap = ctx->grid_cap;
    A.reg_mode = 0;
    // per-bin masks on the matrix cores: the factorised mask tables + the persistent tile kernel
    // (candidate mode without a sink -- the map fallback of find_candidates -- runs on the kernels that decide the screen
    // at run time: the tile kernel's candidate instance has no map output)
    if (allow_fast && mfma_reg_wanted(A) && !(A.ks.cand_cmin > 0.0f && !A.cand_keys)) {
        if (A.out.layout == CS_LAYOUT_BAND) {
            A.out_lo = A.out.band_lo;
            A.out_hi = A.out.band_lo + A.out.band_w - 1;
        } else {
            A.out_lo = -(1 << 30);
            A.out_hi = (1 << 30);
        }
        if (A.cand_keys) {                   // candidate sink: only the scanned diagonals (the map path trims in the compaction)
            A.out_lo = std::max(A.out_lo, A.cand_dlo);
            A.out_hi = std::min(A.out_hi, A.cand_dhi);
        }
        // (the mirrored-row instance of the tile kernel forms the corrections of the edge diagonals itself: no records)
        const bool rim_in_kernel = A.w_sym && A.km == 17 && A.kn == 17 && !std::getenv("CHROMOSIGHT_HIP_MFMA_NORSYM");
        rc = prepare_regular_mask<float>(ctx, A, A.km, stream, rim_in_kernel, true);
        if (rc != CS_OK) return rc;
        if (A.reg_mode == 1 && A.sig_is_f64) {
            // float64 container (the pipeline keeps the detrended band in float64 for the exact re-scoring of the
            // candidates), float32 arithmetic: the rows the windows reach are rounded into context scratch -- what
            // the streaming kernel does pixel by pixel -- 12 B of traffic per pixel for a kernel 1.3x faster
            const int kh = (A.km - 1) / 2;
            const int p_lo = std::max(0, A.row_begin - kh), p_hi = std::min(A.ms, A.row_end + (A.km - 1) - kh);
            const int width = A.sig.layout == CS_LAYOUT_BAND ? A.sig.band_w : A.ns;
            const long long ld = ((long long)width + 15) / 16 * 16;
            rc = ensure_scratch(ctx, &ctx->d_narrow, &ctx->d_narrow_bytes, (size_t)(p_hi - p_lo) * (size_t)ld * 4);
            if (rc != CS_OK) return rc;
            const double* src = reinterpret_cast<const double*>(A.sig.ptr) + ((long long)p_lo - A.sig.row0) * A.sig.ld;
            rc = cs::launch_narrow_rows(src, A.sig.ld, reinterpret_cast<float*>(ctx->d_narrow), ld, p_hi - p_lo, width, ctx->n_cu, stream);
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "narrowing kernel failed: %s", hipGetErrorString((hipError_t)rc));
            A.sig.ptr = ctx->d_narrow;
            A.sig.ld = ld;
            A.sig.row0 = p_lo;
            A.sig_is_f64 = 0;
        }
        if (A.reg_mode == 1) {
            cs::MfmaWeights E;
            rc = ensure_wfrag(ctx, stream, A.km, A.kn, &E);
            if (rc != CS_OK) return rc;
            int path = 0;
            rc = cs::launch_corr_mfma_f32(A, E, stream, &path);
            ctx->last_kernel = path == 2 ? CS_KERNEL_MFMA_REG : CS_KERNEL_MFMA;
            if (rc == -5) return CS_NEED_MAP;
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            if (path == 2 && A.cand_keys) ctx->cand_fused = true;
            return CS_OK;
        }
    }
    if (A.sig.counts) return fail(ctx, CS_ERR_UNSUPPORTED, "a band of counts (CS_LAYOUT_BAND_COUNTS) is read by the masked float32 tile kernel only: per-bin masks, full mode, odd square template of up to 17");
    if (!A.out.ptr && allow_fast && mfma_wide_wanted(A) && !A.w_rank1) {
        // candidate sink without a map, template side 18 .. 33: the two-pass kernel appends the candidates itself
        cs::MfmaWideWeights E;
        rc = ensure_wfrag_wide(ctx, stream, A.km, A.kn, &E);
        if (rc != CS_OK) return rc;
        ctx->last_kernel = CS_KERNEL_MFMA_WIDE;
        rc = cs::launch_corr_mfma_wide_f32(A, E, stream);
        if (rc == -5) return CS_NEED_MAP;
        if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        ctx->cand_fused = true;
        return CS_OK;
    }
    if (!A.out.ptr) return CS_NEED_MAP;      // every other kernel writes a map
    if (allow_fast && mfma_wanted(A)) {
        if (A.sig_is_f64 && A.mask_mode == 0 && A.sig.layout == 0 && A.out.layout == 0 && !A.nobs.ptr) {
            // float64 container, float32 arithmetic: narrow the rows the windows reach into context scratch
            // (what the kernels do pixel by pixel anyway) so that the persistent tile kernel can stage them
            const int kh = (A.km - 1) / 2;
            const int p_lo = std::max(0, A.row_begin - kh), p_hi = std::min(A.ms, A.row_end + (A.km - 1) - kh);
            const long long ld = ((long long)A.ns + 15) / 16 * 16;
            const size_t bytes = (size_t)(p_hi - p_lo) * (size_t)ld * 4;
            rc = ensure_scratch(ctx, &ctx->d_narrow, &ctx->d_narrow_bytes, bytes);
            if (rc != CS_OK) return rc;
            const double* src = reinterpret_cast<const double*>(A.sig.ptr) + ((long long)p_lo - A.sig.row0) * A.sig.ld;
            rc = cs::launch_narrow_rows(src, A.sig.ld, reinterpret_cast<float*>(ctx->d_narrow), ld, p_hi - p_lo, A.ns, ctx->n_cu, stream);
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "narrowing kernel failed: %s", hipGetErrorString((hipError_t)rc));
            A.sig.ptr = ctx->d_narrow;
            A.sig.ld = ld;
            A.sig.row0 = p_lo;
            A.sig_is_f64 = 0;
        }
        cs::MfmaWeights E;
        rc = ensure_wfrag(ctx, stream, A.km, A.kn, &E);
        if (rc != CS_OK) return rc;
        int dense_path = 0;
        rc = cs::launch_corr_mfma_f32(A, E, stream, &dense_path);
        ctx->last_kernel = dense_path == 1 ? CS_KERNEL_MFMA_DENSE : dense_path == 2 ? CS_KERNEL_MFMA_REG : CS_KERNEL_MFMA;
    } else if (allow_fast && fast_compatible(A) && fast_available(A.km, A.kn, &K) && !std::getenv("CHROMOSIGHT_HIP_WIDE_ALL")) {
        ctx->last_kernel = CS_KERNEL_STREAM;
#ifdef CS_HAVE_FAST
#define CS_CASE(KK)                          \
    case KK:                                 \
        cs::corr_fast_tile_k##KK(A.row_end - A.row_begin, A.ns, A.out.layout == 1 ? A.out.band_w : 0, ctx->n_cu, &tw, &th);  \
        fill_grid(A, tw, th, false);         \
        rc = prepare_regular_mask<float>(ctx, A, KK, stream); \
        if (rc != CS_OK) return rc;          \
        rc = cs::launch_corr_fast_f32_k##KK(A, stream); \
        break;
        switch (K) {
            CS_CASE(7) CS_CASE(9) CS_CASE(11) CS_CASE(13) CS_CASE(15) CS_CASE(17)
            default: rc = -1;
        }
#undef CS_CASE
#else
        rc = -1;
#endif
    } else if (allow_fast && A.w_rank1 && !A.xcorr_only && !std::getenv("CHROMOSIGHT_HIP_NO_SEPARABLE") &&
               cs::corr_sep_fits(A.km, A.kn, A.mask_mode != 0) &&
               !(mfma_wide_wanted(A) && A.out.layout == CS_LAYOUT_BAND && A.out.band_w >= 512 && !std::getenv("CHROMOSIGHT_HIP_SEPARABLE_FIRST"))) {
        // (an outer product that also fits the two-pass matrix-core kernel: that one on wide bands -- measured on the
        // 31 x 31 stripes, profiles/r06_template_kernels.txt: 1001 diagonals 5.40 vs 6.92 ms, 234 diagonals 0.577 vs 0.517)
        // templates without an unrolled instance that are an outer product (31 x 31 stripes): separable sums
        ctx->last_kernel = CS_KERNEL_SEPARABLE;
        cs::corr_sep_tile(&tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_sep_f32(A, stream);
    } else if (allow_fast && mfma_wide_wanted(A)) {
        cs::MfmaWideWeights E;
        rc = ensure_wfrag_wide(ctx, stream, A.km, A.kn, &E);
        if (rc != CS_OK) return rc;
        ctx->last_kernel = CS_KERNEL_MFMA_WIDE;
        rc = cs::launch_corr_mfma_wide_f32(A, E, stream);
    } else {
        ctx->last_kernel = CS_KERNEL_GENERIC;
        cs::corr_generic_tile(A.km, A.kn, &tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_generic_f32(A, stream);
    }
    if (rc == -3) return fail(ctx, CS_ERR_UNSUPPORTED, "template %dx%d needs more than 160 KiB of LDS", A.km, A.kn);
    if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

template <>
int launch_corr<double>(cs_ctx* ctx, cs::CorrArgs<double>& A, hipStream_t stream, bool allow_fast)
{
    int K = 0, tw, th, rc;
    if (A.sig.counts) return fail(ctx, CS_ERR_UNSUPPORTED, "a band of counts (CS_LAYOUT_BAND_COUNTS) is read by the masked float32 tile kernel only");
    A.n_cu = ctx->n_cu;
    A.grid_cap = ctx->grid_cap;
    ctx->last_kernel = CS_KERNEL_GENERIC;
    if (allow_fast && fast_compatible(A) && fast_available(A.km, A.kn, &K)) {
        ctx->last_kernel = CS_KERNEL_STREAM;
#ifdef CS_HAVE_FAST
#define CS_CASE(KK)                          \
    case KK:                                 \
        cs::corr_fast_tile_k##KK(A.row_end - A.row_begin, A.ns, A.out.layout == 1 ? A.out.band_w : 0, ctx->n_cu, &tw, &th);  \
        fill_grid(A, tw, th, false);         \
        rc = prepare_regular_mask<double>(ctx, A, KK, stream); \
        if (rc != CS_OK) return rc;          \
        rc = cs::launch_corr_fast_f64_k##KK(A, stream); \
        break;
        switch (K) {
            CS_CASE(7) CS_CASE(9) CS_CASE(11) CS_CASE(13) CS_CASE(15) CS_CASE(17)
            default: rc = -1;
        }
#undef CS_CASE
#else
        rc = -1;
#endif
    } else {
        cs::corr_generic_tile(A.km, A.kn, &tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_generic_f64(A, stream);
    }
    if (rc == -3) return fail(ctx, CS_ERR_UNSUPPORTED, "template %dx%d needs more than 160 KiB of LDS", A.km, A.kn);
    if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int check_matrix(cs_ctx* ctx, const cs_matrix* m, const char* what, int ns)
{
    if (!m || !m->d_ptr) return fail(ctx, CS_ERR_INVALID, "%s: null matrix", what);
    if (m->dtype != CS_F32 && m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "%s: bad dtype", what);
    if (m->layout == CS_LAYOUT_DENSE) {
        if (m->ld < ns) return fail(ctx, CS_ERR_INVALID, "%s: ld < number of columns", what);
    } else if (m->layout == CS_LAYOUT_BAND || m->layout == CS_LAYOUT_BAND_PADDED) {
        if (m->band_w <= 0 || m->ld < m->band_w) return fail(ctx, CS_ERR_INVALID, "%s: bad band geometry", what);
        if (m->layout == CS_LAYOUT_BAND_PADDED && m->ld < (int64_t)m->band_w + 4) return fail(ctx, CS_ERR_INVALID, "%s: a padded band keeps 4 zero slots per row", what);
    } else if ((m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW) && ctx->allow_counts) {
        // a band of raw counts (cs_stage_block.band32_counts): only the masked tile kernel reads it (launch_corr<float>)
        if (m->band_w <= 0 || m->band_lo != 0 || m->row0 != 0 || m->dtype != CS_F32 || m->ld < (int64_t)m->band_w + (m->layout == CS_LAYOUT_BAND_COUNTS ? 4 : 0))
            return fail(ctx, CS_ERR_INVALID, "%s: bad band of counts", what);
    } else if (m->layout == CS_LAYOUT_BAND_LAZY && ctx->allow_lazy) {
        // a lazily evaluated float64 band (cs_stage_block): only the float64 kernels behind the batched foci entries read it
        if (m->band_w <= 0 || m->band_lo != 0 || m->row0 != 0 || m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "%s: bad lazy band", what);
    } else {
        return fail(ctx, CS_ERR_INVALID, "%s: bad layout", what);
    }
    return CS_OK;
}

// Build the device argument block shared by cs_normxcorr2 and cs_rescore_f64.
template <typename TC>
int build_args(cs_ctx* ctx, hipStream_t stream, const cs_matrix* signal, const cs_kernel* kernel,
               const cs_normxcorr2_params* p, cs::CorrArgs<TC>* out)
{
    if (!signal || !kernel || !p) return fail(ctx, CS_ERR_INVALID, "null argument");
    const int km = kernel->km, kn = kernel->kn, kk = km * kn;
    if (km <= 0 || kn <= 0 || !kernel->h_kernel) return fail(ctx, CS_ERR_INVALID, "bad template");
    if (!(km & 1) || !(kn & 1))
        return fail(ctx, CS_ERR_INVALID, "template dimensions must be odd (reference preprocessing.py:774-775)");
    if (p->ms <= 0 || p->ns <= 0) return fail(ctx, CS_ERR_INVALID, "empty signal");
    int rc = check_matrix(ctx, signal, "signal", p->ns);
    if (rc) return rc;
    if (p->mask_mode != CS_MASK_NONE) {
        // detection.py:880-881
        if (std::min(km, kn) >= std::max(p->ms, p->ns))
            return fail(ctx, CS_ERR_INVALID, "cannot have kernel bigger than signal");
        if (p->mask_mode == CS_MASK_BINS && (!p->d_miss_row || !p->d_miss_col))
            return fail(ctx, CS_ERR_INVALID, "missing-bin vectors are null");
        if (p->mask_mode == CS_MASK_EXPLICIT && !p->d_mask)
            return fail(ctx, CS_ERR_INVALID, "explicit mask is null");
    }
    const double* kconv = kernel->h_kernel_conv ? kernel->h_kernel_conv : kernel->h_kernel;
    const bool want_sym = !std::getenv("CHROMOSIGHT_HIP_NO_SYMMETRY");
    cs_ctx::TemplateCache& tc = ctx->tcache[sizeof(TC) == 8 ? 1 : 0];
    {
        std::vector<double> key;
        key.reserve(6 + 3 * (size_t)kk);
        key.push_back(km);
        key.push_back(kn);
        key.push_back(kernel->h_kernel_conv ? 1 : 0);
        key.push_back(kernel->h_kernel_sq ? 1 : 0);
        key.push_back(want_sym ? 1 : 0);
        key.push_back(p->xcorr_threshold);
        key.insert(key.end(), kernel->h_kernel, kernel->h_kernel + kk);
        if (kernel->h_kernel_conv) key.insert(key.end(), kernel->h_kernel_conv, kernel->h_kernel_conv + kk);
        if (kernel->h_kernel_sq) key.insert(key.end(), kernel->h_kernel_sq, kernel->h_kernel_sq + kk);
        const bool hit = key.size() == tc.key.size() && std::memcmp(key.data(), tc.key.data(), key.size() * sizeof(double)) == 0;
        if (!hit) {
            const HostStats st0 = template_stats(kernel->h_kernel, kk);
            if (!(st0.kstd > 0)) return fail(ctx, CS_ERR_INVALID, "Cannot have flat kernel.");  // detection.py:887-888
            tc.stats[0] = st0.n; tc.stats[1] = st0.kmean; tc.stats[2] = st0.kstd;
            tc.stats[3] = st0.kvar; tc.stats[4] = st0.ksum; tc.stats[5] = st0.k2sum;
            std::vector<double>& w0 = tc.w;
            w0.assign(3 * (size_t)kk, 0.0);
            for (int t = 0; t < kk; ++t) {
                const double kc = kconv[t];
                const double k2 = kernel->h_kernel_sq ? kernel->h_kernel_sq[t] : kc * kc;
                w0[t] = kc - st0.kmean;
                if (sizeof(TC) == 8) {
                    // float64 kernels use the reference's literal (un-centred) mask sums
                    w0[kk + t] = kc;
                    w0[2 * kk + t] = k2;
                } else {
                    w0[kk + t] = kc - st0.kmean;
                    // chosen so that kb + 2*kmean*ka + kmean^2*nm == sum over missing pixels of k2
                    w0[2 * kk + t] = k2 - 2.0 * st0.kmean * kc + st0.kmean * st0.kmean;
                }
            }
            // vertical symmetry of the template (loops, stripes): lets the streaming kernel fold template
            // rows.  Rounding-level asymmetries (truncated-SVD reconstructions) are symmetrised for the
            // float32 kernels, whose own rounding is 1e5 times larger; float64 needs exact symmetry.
            bool sym0 = want_sym;
            double wmax = 0;
            for (size_t t = 0; t < w0.size(); ++t) wmax = std::max(wmax, std::fabs(w0[t]));
            const double tol = sizeof(TC) == 8 ? 0.0 : 1e-12 * wmax;
            for (int set = 0; set < 3 && sym0; ++set)
                for (int r = 0; r < km / 2 && sym0; ++r)
                    for (int c = 0; c < kn; ++c)
                        if (std::fabs(w0[set * kk + r * kn + c] - w0[set * kk + (km - 1 - r) * kn + c]) > tol) {
                            sym0 = false;
                            break;
                        }
            if (sym0 && sizeof(TC) == 4)
                for (int set = 0; set < 3; ++set)
                    for (int r = 0; r < km / 2; ++r)
                        for (int c = 0; c < kn; ++c) w0[set * kk + (km - 1 - r) * kn + c] = w0[set * kk + r * kn + c];
            tc.sym = sym0;
            // Exactly rank-1 templates (the 31 x 31 stripes templates): u and v for the separable kernel
            // (cs_corr_sep.hip).  Pivot on the largest entry; the outer product must reproduce the template to
            // rounding (1e-12 relative) -- truncated-SVD templates carry their own squares and stay on the full kernels.
            tc.rank1 = false;
            if (sizeof(TC) == 4 && !kernel->h_kernel_sq) {
                int pa = 0, pb = 0;
                double kmax = 0;
                for (int a = 0; a < km; ++a)
                    for (int b = 0; b < kn; ++b)
                        if (std::fabs(kconv[a * kn + b]) > kmax) {
                            kmax = std::fabs(kconv[a * kn + b]);
                            pa = a;
                            pb = b;
                        }
                if (kmax > 0) {
                    std::vector<double> u(km), v(kn);
                    for (int b = 0; b < kn; ++b) v[b] = kconv[pa * kn + b];
                    for (int a = 0; a < km; ++a) u[a] = kconv[a * kn + pb] / kconv[pa * kn + pb];
                    double worst = 0;
                    for (int a = 0; a < km; ++a)
                        for (int b = 0; b < kn; ++b) worst = std::max(worst, std::fabs(u[a] * v[b] - kconv[a * kn + b]));
                    if (worst <= 1e-12 * kmax) {
                        tc.rank1 = true;
                        w0.insert(w0.end(), u.begin(), u.end());
                        w0.insert(w0.end(), v.begin(), v.end());
                    }
                }
            }
            // can sum_missing K' or sum_missing K'^2 of a non-empty set fall under the zeroing threshold?
            // not if every entry alone exceeds it and all have one sign (the built-in templates: >= 0.5)
            double lo = 1e300, lo2 = 1e300;
            bool pos = true, pos2 = true;
            std::vector<double> distinct;
            for (int t = 0; t < kk; ++t) {
                const double kc = kconv[t];
                const double k2 = kernel->h_kernel_sq ? kernel->h_kernel_sq[t] : kc * kc;
                lo = std::min(lo, std::fabs(kc));
                lo2 = std::min(lo2, std::fabs(k2));
                pos = pos && kc > 0;
                pos2 = pos2 && k2 > 0;
                bool seen = false;
                for (double d : distinct) seen = seen || std::fabs(d - kc) <= 1e-9 * std::max(1.0, std::fabs(d));
                if (!seen && distinct.size() < 64) distinct.push_back(kc);
            }
            tc.zk_possible = !(pos && pos2 && lo > 2 * p->xcorr_threshold && lo2 > 2 * p->xcorr_threshold);
            // piecewise-constant templates (borders, hairpins: 2 levels) have windows whose present pixels
            // are all equal; smooth templates never do, but with up to 75 % of a window missing a handful of
            // levels can still coincide, so only clearly many-valued templates skip the snap
            tc.snap_possible = distinct.size() < 64;
            tc.key.swap(key);
        }
    }
    HostStats st;
    st.n = tc.stats[0]; st.kmean = tc.stats[1]; st.kstd = tc.stats[2];
    st.kvar = tc.stats[3]; st.ksum = tc.stats[4]; st.k2sum = tc.stats[5];
    const std::vector<double>& w = tc.w;
    const bool sym = tc.sym;
    rc = upload_weights<TC>(ctx, stream, w);
    if (rc) return rc;

    cs::CorrArgs<TC> A;
    std::memset(&A, 0, sizeof(A));
    A.sig = view_of(signal);
    A.sig_is_f64 = signal->dtype == CS_F64;
    A.ms = p->ms;
    A.ns = p->ns;
    A.km = km;
    A.kn = kn;
    A.full = p->full ? 1 : 0;
    A.sym_upper = p->sym_upper ? 1 : 0;
    A.max_dist = p->max_dist;
    A.mask_mode = p->mask_mode;
    A.miss_row = p->d_miss_row;
    A.miss_col = p->d_miss_col;
    A.mask = A.sig;
    A.mask.ptr = (void*)p->d_mask;
    A.w = reinterpret_cast<const TC*>(ctx->d_w[sizeof(TC) == 8 ? 1 : 0]);
    A.ks.n = (TC)st.n;
    A.ks.inv_n = (TC)(1.0 / st.n);
    A.ks.kmean = (TC)st.kmean;
    A.ks.kstd = (TC)st.kstd;
    A.ks.kvar = (TC)st.kvar;
    A.ks.ksum = (TC)st.ksum;
    A.ks.k2sum = (TC)st.k2sum;
    A.ks.thr = (TC)p->xcorr_threshold;
    A.ks.eps = (TC)p->denom_eps;
    A.ks.cut = (TC)p->min_present;
    A.ks.thr_n = (TC)(p->xcorr_threshold * st.n);
    A.ks.nkvar = (TC)(st.n * st.kvar);
    A.ks.eps2 = (TC)(p->denom_eps * p->denom_eps);
    A.ks.den2_min = (TC)(p->denom_eps * p->denom_eps * st.n * st.n);
    A.ks.zk_possible = tc.zk_possible;
    A.ks.snap_possible = tc.snap_possible;
    A.ks.cand_cmin = (TC)0;              // candidate mode: corr_candidates_f32 below
    A.ks.cand_thr = (TC)0;
    A.xcorr_only = 0;
    A.w_sym = sym ? 1 : 0;
    A.w_rank1 = tc.rank1 ? 1 : 0;
    A.row_begin = 0;
    A.row_end = p->ms;
    if (p->row_end > p->row_begin) {
        if (p->row_begin < 0 || p->row_end > p->ms) return fail(ctx, CS_ERR_INVALID, "row window outside the matrix");
        if ((p->row_begin != 0 || p->row_end != p->ms) && p->mask_mode == CS_MASK_EXPLICIT)
            return fail(ctx, CS_ERR_UNSUPPORTED, "row windows need per-bin masks or none");
        A.row_begin = p->row_begin;
        A.row_end = p->row_end;
    }
    *out = A;
    return CS_OK;
}


// grow-only device scratch; growing waits for the device, since queued work may still use the old block
int ensure_scratch(cs_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (need <= *have) return CS_OK;
    if (*buf) {
        CS_HIP(ctx, hipDeviceSynchronize());
        CS_HIP(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    const size_t want = need + need / 4;
    CS_HIP(ctx, hipMalloc(buf, want));
    *have = want;
    return CS_OK;
}

template int upload_weights<float>(cs_ctx*, hipStream_t, const std::vector<double>&);
template int upload_weights<double>(cs_ctx*, hipStream_t, const std::vector<double>&);
template int build_args<float>(cs_ctx*, hipStream_t, const cs_matrix*, const cs_kernel*, const cs_normxcorr2_params*, cs::CorrArgs<float>*);
template int build_args<double>(cs_ctx*, hipStream_t, const cs_matrix*, const cs_kernel*, const cs_normxcorr2_params*, cs::CorrArgs<double>*);

}  // namespace csapi

using namespace csapi;


// ============================================================================================
extern "C" {

const char* cs_version(void) { return "chromosight_hip 0.1 (gfx950)"; }

int cs_last_kernel(const cs_ctx* ctx) { return ctx ? ctx->last_kernel : 0; }

int cs_ctx_set_range_check(cs_ctx* ctx, int32_t on)
{
    if (!ctx) return CS_ERR_INVALID;
    ctx->range_check = on ? 1 : 0;
    return CS_OK;
}

int cs_ctx_create(int device, cs_ctx** out)
{
    if (!out) return CS_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return CS_ERR_HIP;
    if (device < 0 || device >= count) return CS_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return CS_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CS_ERR_HIP;
    cs_ctx* ctx = new cs_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    if (hipMalloc((void**)&ctx->d_tiles_started, 256) != hipSuccess || hipMemset(ctx->d_tiles_started, 0, 256) != hipSuccess) {
        delete ctx;
        return CS_ERR_HIP;
    }
    *out = ctx;
    return CS_OK;
}

void cs_ctx_destroy(cs_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int slot = 0; slot < 2; ++slot) {
        if (ctx->d_w[slot]) (void)hipFree(ctx->d_w[slot]);
        for (auto& pk : ctx->w_parked[slot])
            if (pk.d) (void)hipFree(pk.d);
    }
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_tiles_started) (void)hipFree(ctx->d_tiles_started);
    if (ctx->d_wfrag) (void)hipFree(ctx->d_wfrag);
    if (ctx->d_rim) (void)hipFree(ctx->d_rim);
    if (ctx->d_wfrag_wide) (void)hipFree(ctx->d_wfrag_wide);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->d_cand_cnt) (void)hipFree(ctx->d_cand_cnt);
    if (ctx->d_map) (void)hipFree(ctx->d_map);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    for (int k = 0; k < 2; ++k) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->ev_stage[k]) (void)hipEventDestroy(ctx->ev_stage[k]);
    }
    for (int k = 0; k < kBlkLanes - 1; ++k) {
        if (ctx->ws_alt[k]) (void)hipFree(ctx->ws_alt[k]);
        if (ctx->s_blk[k]) (void)hipStreamDestroy(ctx->s_blk[k]);
    }
    for (int k = 0; k < kBlkLanes; ++k)
        if (ctx->ev_blk[k]) (void)hipEventDestroy(ctx->ev_blk[k]);
    for (void* w : ctx->ws_tab)
        if (w) (void)hipFree(w);
    if (ctx->h_tab) (void)hipHostFree(ctx->h_tab);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_pool) (void)hipFree(ctx->d_pool);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->h_cand_counts) (void)hipHostFree(ctx->h_cand_counts);
    if (ctx->d_counts_peak) (void)hipFree(ctx->d_counts_peak);
    if (ctx->h_peak) (void)hipHostFree(ctx->h_peak);
    if (ctx->d_narrow) (void)hipFree(ctx->d_narrow);
    if (ctx->h_blk_counts) (void)hipHostFree(ctx->h_blk_counts);
    if (ctx->d_host_in) (void)hipFree(ctx->d_host_in);
    if (ctx->d_host_out) (void)hipFree(ctx->d_host_out);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    for (hipStream_t st : {ctx->s_up, ctx->s_run, ctx->s_down})
        if (st) (void)hipStreamDestroy(st);
    for (auto* v : {&ctx->ev_up, &ctx->ev_run, &ctx->ev_down})
        for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    delete ctx;
}

const char* cs_last_error(const cs_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int cs_device_cu_count(const cs_ctx* ctx) { return ctx ? ctx->n_cu : 0; }

int cs_malloc(cs_ctx* ctx, size_t bytes, void** d_ptr)
{
    if (!d_ptr) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return CS_OK;
}

int cs_free(cs_ctx* ctx, void* d_ptr)
{
    CS_ENTER(ctx);
    if (d_ptr) CS_HIP(ctx, hipFree(d_ptr));
    return CS_OK;
}

int cs_memcpy_h2d(cs_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) {
        CS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        // the caller may reuse h_src as soon as this returns (pageable sources are not always staged
        // before an asynchronous copy returns)
        CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    }
    return CS_OK;
}

int cs_memcpy_d2h(cs_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) {
        CS_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    }
    return CS_OK;
}

int cs_memset(cs_ctx* ctx, void* d_dst, int value, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) CS_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
    return CS_OK;
}

int cs_stream_sync(cs_ctx* ctx, void* stream)
{
    CS_ENTER(ctx);
    CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CS_OK;
}

int cs_stream_create(cs_ctx* ctx, void** stream)
{
    if (!stream) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    hipStream_t s;
    CS_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return CS_OK;
}

int cs_stream_create_priority(cs_ctx* ctx, int32_t high, void** stream)
{
    if (!stream) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    int least = 0, greatest = 0;                     // numerically lower = served first
    CS_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    CS_HIP(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    *stream = (void*)s;
    return CS_OK;
}

int cs_stream_destroy(cs_ctx* ctx, void* stream)
{
    CS_ENTER(ctx);
    if (stream) CS_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return CS_OK;
}

int cs_event_create(cs_ctx* ctx, void** event)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    hipEvent_t e;
    CS_HIP(ctx, hipEventCreate(&e));
    *event = (void*)e;
    return CS_OK;
}

int cs_event_destroy(cs_ctx* ctx, void* event)
{
    CS_ENTER(ctx);
    if (event) CS_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return CS_OK;
}

int cs_event_record(cs_ctx* ctx, void* event, void* stream)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return CS_OK;
}

int cs_stream_wait_event(cs_ctx* ctx, void* stream, void* event)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return CS_OK;
}

int cs_stream_wait_tiles(cs_ctx* ctx, void* stream, cs_ctx* tiles_ctx, int32_t epoch, int32_t timeout_us)
{
    CS_ENTER(ctx);
    if (!tiles_ctx || !tiles_ctx->d_tiles_started || tiles_ctx->device != ctx->device) return fail(ctx, CS_ERR_INVALID, "no tile context on this device");
    if (epoch <= 0 || epoch >= (1 << 23)) return fail(ctx, CS_ERR_INVALID, "tile epochs are 1 .. 2^23 - 1");
    hipLaunchKernelGGL(cs_wait_tiles_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)tiles_ctx->d_tiles_started,
                       (unsigned)epoch, (long long)std::max(1, std::min(timeout_us, 5000)) * 100);
    CS_HIP(ctx, hipGetLastError());
    return CS_OK;
}

int cs_event_elapsed_ms(cs_ctx* ctx, void* start, void* stop, float* ms)
{
    if (!start || !stop || !ms) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    CS_HIP(ctx, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return CS_OK;
}

}  // extern "C"
