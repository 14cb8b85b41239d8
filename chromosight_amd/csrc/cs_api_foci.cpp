// cs_api_foci.cpp -- the C ABI (include/chromosight_hip.h): device foci (cs_detect_foci and its batched / multi-block forms,
// cs_candidates, cs_label_foci) and the quantify entries.  Context, weights and kernel dispatch: cs_api.cpp (cs_api_internal.h).
#include "cs_api_internal.h"

using namespace csapi;

extern "C" {

static int check_foci_args(cs_ctx* ctx, const cs_matrix* signal, const cs_kernel* kernel, const cs_normxcorr2_params* p,
                           const cs_foci_params* fp)
{
    if (!signal || !kernel || !p || !fp) return fail(ctx, CS_ERR_INVALID, "null argument");
    // (every foci entry shares d_pool, the candidate counters and the argument tables with a pending prepare form of
    // cs_detect_foci_blocks: whatever runs in between invalidates it -- that entry notes the flag before its own checks)
    ctx->prep_pending = false;
    // (every foci entry shares the context's count words, block tables and record pool with an asynchronous batch)
    if (ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "an asynchronous batch is pending on this context: cs_detect_foci_batch_finish first");
    if (p->mask_mode == CS_MASK_EXPLICIT) return fail(ctx, CS_ERR_UNSUPPORTED, "device foci need per-bin masks or none");
    if (signal->layout == CS_LAYOUT_BAND_COUNTS || signal->layout == CS_LAYOUT_BAND_COUNTS_VIEW)
        return fail(ctx, CS_ERR_INVALID, "a band of counts is the float32 twin of a block (signals_f32), not its signal: hand the block's CS_LAYOUT_BAND_LAZY descriptor");
    if (fp->min_size < 1) return fail(ctx, CS_ERR_INVALID, "min_size must be >= 1");
    if (!ctx->h_counts) CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_counts, 64, hipHostMallocDefault));
    return CS_OK;
}

namespace {
// Where the masked matrix-core tile kernel appends its candidates (keys tag + row * ns + col); see CorrArgs::cand_keys.
struct CandSink {
    unsigned long long* keys;
    unsigned long long* count;       // zeroed by the caller
    long long cap;
    unsigned long long tag;
    int lo_diag, hi_diag;            // scanned diagonals
    void* defer_args = nullptr;      // see CorrArgs::defer_args: prepare the tile kernel's launch, do not launch
    int* defer_rsym = nullptr;
};

// float32 correlation in candidate mode (cs_device.h cand_screen_*: margin + conditioning screen, sentinel 2.0).  With a
// sink and the masked tile kernel the candidates are appended to the sink and no map is written (ctx->cand_fused);
// otherwise the map goes to `out` -- CS_NEED_MAP when that has no storage, before anything that matters was launched.
int corr_candidates_f32(cs_ctx* ctx, hipStream_t stream, const cs_matrix* signal, const cs_kernel* kernel,
                        const cs_normxcorr2_params* p, const cs_matrix* out, double margin, double thr, const CandSink* sink)
{
    cs::CorrArgs<float> A;
    int rc = build_args<float>(ctx, stream, signal, kernel, p, &A);
    if (rc) return rc;
    // windows conditioned at least 8 n 2^-24 / margin have a float32 error below margin / 4 (2 gamma / conditioning)
    A.ks.cand_cmin = (float)std::min(0.5, 8.0 * (double)A.ks.n * 0x1p-24 / margin);
    A.ks.cand_thr = (float)thr;
    A.out = view_of(out);
    A.out_is_f64 = 0;
    A.nobs = cs::MatView{nullptr, 0, 0, 0, 0, 0};
    if (sink) {
        A.cand_keys = sink->keys;
        A.cand_count = sink->count;
        A.cand_cap = sink->cap;
        A.cand_tag = sink->tag;
        A.cand_dlo = sink->lo_diag;
        A.cand_dhi = sink->hi_diag;
        A.defer_args = sink->defer_args;
        A.defer_rsym = sink->defer_rsym;
    } else if (!out || !out->d_ptr) {
        return fail(ctx, CS_ERR_INVALID, "candidate mode needs a sink or a map");
    }
    ctx->cand_fused = false;
    return launch_corr<float>(ctx, A, stream, getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr);
}

// 1-D patterns (cs_foci_params.diag_only): the reference forces bin1 = bin2 AFTER shifting the coordinates by (kh, kw)
// into a map padded by (kw, kh) (detection.py:287-315, preprocessing.py:636-676), so with a non-square template in full
// mode the row ends up kw - kh away from the column.  The kernels take an odd code whose upper bits hold that offset.
inline int diag_code(const cs_foci_params* fp, const cs_kernel* kernel, const cs_normxcorr2_params* p)
{
    if (!fp->diag_only) return 0;
    const int shift = p->full ? (kernel->kn - 1) / 2 - (kernel->km - 1) / 2 : 0;
    return shift * 2 + 1;
}

// Candidate stage shared by cs_detect_foci and cs_candidates: the coefficient map of the row window in
// context scratch, thresholded compaction (or, for 1-D patterns, the enumeration of the few scanned
// diagonals).  The pool is laid out as rows | cols | vals | counters | windows | tail.
struct CandPlan {
    size_t off_cols = 0, off_vals = 0, off_cnt = 0, off_win = 0, off_tail = 0;
    long long n_cand = 0;
    int row_major = 0;       // the candidate list is already sorted row-major (1-D patterns)
};

int find_candidates(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                    const cs_normxcorr2_params* p, const cs_foci_params* fp, size_t win_bytes,
                    size_t (*tail_bytes)(long long), CandPlan* P)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    int rb = 0, re = p->ms;
    if (p->row_end > p->row_begin) {
        if (p->row_begin < 0 || p->row_end > p->ms) return fail(ctx, CS_ERR_INVALID, "row window outside the matrix");
        rb = p->row_begin;
        re = p->row_end;
    }
    // ---- coefficient map in context-owned scratch, with the signal's layout
    const bool f64 = p->compute_dtype == CS_F64;
    cs_matrix map;
    map.d_ptr = nullptr;
    map.dtype = f64 ? CS_F64 : CS_F32;
    map.layout = signal->layout == CS_LAYOUT_BAND_PADDED ? CS_LAYOUT_BAND : signal->layout;
    map.row0 = rb;
    if (signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) {
        if (fp->hi_diag < fp->lo_diag) return fail(ctx, CS_ERR_INVALID, "empty diagonal range");
        map.band_lo = fp->lo_diag;
        map.band_w = fp->hi_diag - fp->lo_diag + 1;
        map.ld = ((int64_t)map.band_w + 63) / 64 * 64;
    } else {
        map.band_lo = map.band_w = 0;
        map.ld = ((int64_t)p->ns + 15) / 16 * 16;
    }
    // 1-D patterns (borders, hairpins: max_dist = 0 in the config, 2 scanned diagonals): a streamed
    // 128-column strip would compute 64 columns for every one it keeps.  Every pixel of the few diagonals
    // is a candidate instead and goes straight to the float64 evaluation (one wave per pixel).
    const bool narrow = (signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) && map.band_w <= 4;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto layout = [&](size_t c_cap) {
        P->off_cols = al(4 * c_cap);
        P->off_vals = P->off_cols + al(4 * c_cap);
        P->off_cnt = P->off_vals + al(8 * c_cap);
        P->off_win = P->off_cnt + 256;
        P->off_tail = P->off_win + al(win_bytes);
        return ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, P->off_tail + tail_bytes((long long)c_cap));
    };
    P->n_cand = 0;
    if (narrow) {
        const long long n_cand = cs::narrow_band_pixels(rb, re, p->ns, map.band_lo, map.band_w);
        if (n_cand > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", n_cand);
        if (n_cand > 0) {
            rc = layout((size_t)n_cand);
            if (rc) return rc;
            char* pool = (char*)ctx->d_pool;
            rc = cs::enqueue_enumerate_band(rb, re, p->ns, map.band_lo, map.band_w, n_cand, (int*)pool,
                                            (int*)(pool + P->off_cols), stream, &P->row_major);
            if (rc) return fail(ctx, CS_ERR_HIP, "enumerate launch failed: %s", hipGetErrorString((hipError_t)rc));
        }
        P->n_cand = n_cand;
        return CS_OK;
    }
    const size_t map_bytes = (size_t)(re - rb) * (size_t)map.ld * (f64 ? 8 : 4);
    // float32 maps are computed in candidate mode (cs_device.h cand_screen_*): a pixel keeps its value only if that is
    // below pearson - margin on a window conditioned well enough for the float32 error to stay under margin / 4; every
    // other pixel holds 2.0 and is re-evaluated.  float64 maps hold the coefficient.
    const double margin = std::max(fp->rescore_margin, 1e-4);
    const double thr = f64 ? fp->pearson : fp->pearson - margin;
    size_t c_cap = std::max<size_t>(1 << 16, (size_t)(re - rb) * (size_t)((signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) ? map.band_w : p->ns) / 256);
    // ---- float32, masked tile kernel: the kernel appends the candidates itself (no map, no compaction pass)
    bool fused_ok = !f64;
    while (fused_ok) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        rc = layout(c_cap);
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        long long* d_cnt = (long long*)(pool + P->off_cnt);
        CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, stream));
        CandSink sink{(unsigned long long*)(pool + P->off_vals), (unsigned long long*)d_cnt, (long long)c_cap, 0ull,
                      fp->lo_diag, fp->hi_diag};
        rc = corr_candidates_f32(ctx, stream, signal, kernel, p, &map, margin, thr, &sink);
        if (rc == CS_NEED_MAP) {
            fused_ok = false;
            break;
        }
        if (rc) return rc;
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        P->n_cand = ctx->h_counts[0];
        if ((size_t)P->n_cand <= c_cap) {
            if (P->n_cand > 0) {
                rc = cs::launch_decode_keys((const long long*)(pool + P->off_vals), P->n_cand, p->ns, (int*)pool,
                                            (int*)(pool + P->off_cols), stream);
                if (rc) return fail(ctx, CS_ERR_HIP, "key decoding failed: %s", hipGetErrorString((hipError_t)rc));
            }
            return CS_OK;
        }
        c_cap = (size_t)P->n_cand + (size_t)P->n_cand / 8;      // the list overflowed: once more with room for all
    }
    // ---- coefficient map in context scratch, then thresholded compaction
    rc = ensure_scratch(ctx, &ctx->d_map, &ctx->d_map_bytes, map_bytes);
    if (rc) return rc;
    map.d_ptr = ctx->d_map;
    if (f64) rc = cs_normxcorr2(ctx, stream_, signal, kernel, p, &map, nullptr);
    else rc = corr_candidates_f32(ctx, stream, signal, kernel, p, &map, margin, thr, nullptr);
    if (rc) return rc;
    while (true) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        rc = layout(c_cap);
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        long long* d_cnt = (long long*)(pool + P->off_cnt);
        CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, stream));
        rc = cs::launch_compact_ge(view_of(&map), f64, re, p->ns, thr, fp->lo_diag, fp->hi_diag, (int*)pool,
                                   (int*)(pool + P->off_cols), (double*)(pool + P->off_vals), (long long)c_cap, d_cnt,
                                   ctx->n_cu, stream);
        if (rc) return fail(ctx, CS_ERR_HIP, "compact launch failed: %s", hipGetErrorString((hipError_t)rc));
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        P->n_cand = ctx->h_counts[0];
        if ((size_t)P->n_cand <= c_cap) break;
        c_cap = (size_t)P->n_cand + (size_t)P->n_cand / 8;
    }
    return CS_OK;
}
}  // namespace

int cs_detect_foci(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* p, const cs_foci_params* fp, cs_focus* h_foci, int64_t cap,
                   int64_t* n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    static_assert(sizeof(cs_focus) == sizeof(cs::FocusRec), "record layouts must agree");
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (!n_foci || cap < 0 || (cap > 0 && !h_foci)) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    if (p->row_end > p->row_begin && (p->row_begin != 0 || p->row_end != p->ms))
        return fail(ctx, CS_ERR_INVALID, "foci of a row window: use cs_candidates + cs_label_foci");
    *n_foci = 0;
    const int kk = kernel->km * kernel->kn;
    const size_t win_pat = fp->want_windows ? (size_t)std::max<int64_t>(cap, 1) : 0;
    CandPlan P;
    rc = find_candidates(ctx, stream_, signal, kernel, p, fp, 8 * win_pat * kk, cs::foci_scratch_bytes, &P);
    if (rc) return rc;
    const long long n_cand = P.n_cand;
    if (n_cand == 0) return CS_OK;
    // ---- foci
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    long long* d_cnt = (long long*)(pool + P.off_cnt);
    double* d_win = fp->want_windows ? (double*)(pool + P.off_win) : nullptr;
    cs::FocusRec* d_rec = nullptr;
    // Page-locked output buffers (what cs_host_alloc hands out) are written by the last kernel itself: one
    // stream synchronisation per call instead of a count round trip plus two copies.
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec_direct = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win_direct = (fp->want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    const bool direct = rec_direct && (!fp->want_windows || !h_windows || win_direct);
    if (direct) {
        ctx->h_counts[1] = -1;
        rc = cs::enqueue_foci(A64, (const int*)pool, (const int*)(pool + P.off_cols), n_cand, fp->pearson, fp->min_size,
                              diag_code(fp, kernel, p), fp->inter, pool + P.off_tail, &d_rec, win_direct, win_direct ? (long long)cap : 0,
                              d_cnt + 1, stream, P.row_major, rec_direct, (long long)cap, ctx->h_counts + 1);
        if (rc) return fail(ctx, CS_ERR_HIP, "foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        const long long n = ctx->h_counts[1];
        if (n < 0) return fail(ctx, CS_ERR_HIP, "foci kernels did not report a count");
        *n_foci = n;
        if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", n, (long long)cap);
        return CS_OK;
    }
    rc = cs::enqueue_foci(A64, (const int*)pool, (const int*)(pool + P.off_cols), n_cand, fp->pearson, fp->min_size,
                          diag_code(fp, kernel, p), fp->inter, pool + P.off_tail, &d_rec, d_win, (long long)win_pat, d_cnt + 1, stream,
                          P.row_major, nullptr, 0, nullptr);
    if (rc) return fail(ctx, CS_ERR_HIP, "foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts + 1, d_cnt + 1, 8, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long n = ctx->h_counts[1];
    *n_foci = n;
    if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", n, (long long)cap);
    if (n > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_foci, d_rec, sizeof(cs_focus) * (size_t)n, hipMemcpyDeviceToHost, stream));
        if (d_win && h_windows)
            CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * (size_t)n * kk, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_detect_foci_batch_templates(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, int32_t n_kernels,
                                   const cs_kernel* kernels, const cs_normxcorr2_params* params, const cs_foci_params* foci,
                                   cs_focus* h_foci, int64_t cap, int64_t* h_n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    AllowLazy allow_lazy(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernels || !params || !foci || !h_n_foci || cap < 0 || (cap > 0 && !h_foci))
        return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    // (the weight sets of all templates stay resident side by side: the current one + the parked ones of upload_weights)
    if (n_kernels < 1 || n_kernels > 4) return fail(ctx, CS_ERR_UNSUPPORTED, "1 to 4 templates per batch");
    for (int t = 1; t < n_kernels; ++t)
        if (kernels[t].km != kernels[0].km || kernels[t].kn != kernels[0].kn)
            return fail(ctx, CS_ERR_INVALID, "the templates of a batch share their size");
    // virtual block v = t * n_blocks + b: sub-matrix b under template t
    const int n_virtual = n_blocks * n_kernels;
    // (the host tables live in the context: an asynchronous call -- foci[0].reserved & 1 -- returns while their copies may
    // still be in flight; cs_detect_foci_batch_finish ends the call)
    if (ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "an asynchronous batch is pending on this context: cs_detect_foci_batch_finish first");
    std::vector<cs::CorrArgs<double>>& tab = ctx->nb_tab;
    std::vector<long long>& seg = ctx->nb_seg;
    std::vector<int>& lo_w = ctx->nb_lo_w;
    tab.assign((size_t)n_virtual, cs::CorrArgs<double>{});
    seg.assign((size_t)n_virtual + 1, 0);
    lo_w.assign(2 * (size_t)n_virtual, 0);
    for (int t = 0; t < n_kernels; ++t)
        for (int b = 0; b < n_blocks; ++b) {
            const int v = t * n_blocks + b;
            const cs_normxcorr2_params* p = params + b;
            const cs_foci_params* fp = foci + b;
            int rc = check_foci_args(ctx, signals + b, kernels + t, p, fp);
            if (rc) return rc;
            const int w = fp->hi_diag - fp->lo_diag + 1;
            if (!is_band(signals[b].layout) || w < 1 || w > 4 || fp->lo_diag < 0 || (p->row_end > p->row_begin))
                return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry takes 1-D patterns (<= 4 scanned diagonals from 0 up) of whole banded blocks");
            if (fp->pearson != foci[0].pearson || fp->min_size != foci[0].min_size || fp->diag_only != foci[0].diag_only ||
                fp->inter != foci[0].inter || fp->want_windows != foci[0].want_windows)
                return fail(ctx, CS_ERR_INVALID, "the blocks of a batch share the detection parameters");
            const long long n_b = cs::narrow_band_pixels(0, p->ms, p->ns, fp->lo_diag, w);
            if (n_b > cs::kFociSmallMax) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d has %lld candidate pixels", b, n_b);
            seg[v + 1] = seg[v] + n_b;
            lo_w[2 * v] = fp->lo_diag;
            lo_w[2 * v + 1] = w;
            rc = build_args<double>(ctx, stream, signals + b, kernels + t, p, &tab[v]);      // (uploads template t once: b == 0)
            if (rc) return rc;
        }
    const long long n_total = seg[n_virtual];
    if (n_total > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", n_total);
    // results straight into page-locked caller buffers (cs_host_alloc); anything else goes through cs_detect_foci
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win = (foci[0].want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    if ((cap > 0 && !rec) || (foci[0].want_windows && h_windows && !win))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry writes into page-locked buffers (cs_host_alloc)");
    // per-block counts through a page-locked array owned by the context
    const size_t cnt_bytes = 8 * ((size_t)n_virtual + 2);
    if (cnt_bytes > ctx->h_blk_bytes) {
        if (ctx->h_blk_counts) CS_HIP(ctx, hipHostFree(ctx->h_blk_counts));
        ctx->h_blk_counts = nullptr;
        ctx->h_blk_bytes = 0;
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_blk_counts, 2 * cnt_bytes, hipHostMallocDefault));
        ctx->h_blk_bytes = 2 * cnt_bytes;
    }
    ctx->h_blk_counts[0] = -1;
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, cs::narrow_batch_scratch_bytes(n_virtual, n_total));
    if (rc) return rc;
    rc = cs::enqueue_foci_narrow_batch(tab.data(), seg.data(), lo_w.data(), n_virtual, foci[0].pearson, foci[0].min_size,
                                       diag_code(foci, kernels, params), foci[0].inter, ctx->d_pool, rec, (long long)cap, win,
                                       win ? (long long)cap : 0, ctx->h_blk_counts, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    ctx->nb_pending = n_virtual;
    ctx->nb_cap = cap;
    if (foci[0].reserved & 1) return CS_OK;          // asynchronous: everything is enqueued, cs_detect_foci_batch_finish waits
    return cs_detect_foci_batch_finish(ctx, stream_, h_n_foci);
}

int cs_detect_foci_batch_finish(cs_ctx* ctx, void* stream_, int64_t* h_n_foci)
{
    CS_ENTER(ctx);
    if (!h_n_foci) return fail(ctx, CS_ERR_INVALID, "null counts");
    if (!ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "no batch is pending on this context");
    const int n_virtual = ctx->nb_pending;
    ctx->nb_pending = 0;
    CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream_));            // also: the host tables were consumed
    const long long total = ctx->h_blk_counts[0];
    if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
    for (int v = 0; v < n_virtual; ++v) h_n_foci[v] = ctx->h_blk_counts[1 + v];
    if (total > ctx->nb_cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)ctx->nb_cap);
    return CS_OK;
}

int cs_detect_foci_batch(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                         const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci, int64_t cap,
                         int64_t* h_n_foci, double* h_windows)
{
    return cs_detect_foci_batch_templates(ctx, stream_, n_blocks, signals, 1, kernel, params, foci, h_foci, cap, h_n_foci, h_windows);
}

// 2-D patterns (loops, stripes: a band of scanned diagonals) of MANY sub-matrices with one launch chain: the masked
// matrix-core tile kernel of every block appends its candidates to one list (composite keys block | row | col), then
// ONE sort, ONE float64 re-scoring, one labelling workgroup per block, one statistics pass -- instead of ~20 small
// launches and three synchronisations per block (23 blocks of a human genome: the loops pass was bound by them).
int cs_detect_foci_blocks(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_matrix* signals_f32,
                          const cs_kernel* kernel, const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci,
                          int64_t cap, int64_t* h_n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    AllowLazy allow_lazy(ctx);
    Laps laps("detect_foci_blocks");
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernel || !params || !foci || !h_n_foci || cap < 0 || (cap > 0 && !h_foci))
        return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    const bool prepare_only = (foci[0].reserved & 2) != 0;
    const unsigned tile_epoch = (unsigned)foci[0].reserved >> 8;          // (0: nobody waits for this call's tile launch)
    const bool was_pending = ctx->prep_pending;
    ctx->prep_pending = false;                     // (whatever a prepare form left is used by the very next call or not at all)
    constexpr int kKeyShift = 40;                 // row * ns + col < 2^40: sub-matrices of up to 2^20 bins
    bool all_narrow = true;
    long long pixels = 0;
    std::vector<long long> block_pixels((size_t)std::max(n_blocks, 1), 0);
    for (int b = 0; b < n_blocks; ++b) {
        const cs_normxcorr2_params* p = params + b;
        const cs_foci_params* fp = foci + b;
        int rc = check_foci_args(ctx, signals + b, kernel, p, fp);
        if (rc) return rc;
        if (fp->pearson != foci[0].pearson || fp->min_size != foci[0].min_size || fp->diag_only != foci[0].diag_only ||
            fp->inter != foci[0].inter || fp->want_windows != foci[0].want_windows || fp->rescore_margin != foci[0].rescore_margin)
            return fail(ctx, CS_ERR_INVALID, "the blocks of a batch share the detection parameters");
        if (p->row_end > p->row_begin) return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry takes whole blocks");
        if (fp->hi_diag < fp->lo_diag) return fail(ctx, CS_ERR_INVALID, "empty diagonal range");
        const int w = fp->hi_diag - fp->lo_diag + 1;
        all_narrow = all_narrow && is_band(signals[b].layout) && w <= 4 && fp->lo_diag >= 0;
        if (signals[b].layout == CS_LAYOUT_BAND_LAZY && !(signals_f32 && signals_f32[b].d_ptr) && !(is_band(signals[b].layout) && w <= 4 && fp->lo_diag >= 0))
            return fail(ctx, CS_ERR_INVALID, "block %d: a lazily evaluated float64 band needs its float32 twin for the tile kernel", b);
        if ((long long)p->ms * p->ns >= (1ll << kKeyShift)) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d is too large for the batch keys", b);
        pixels += (long long)p->ms * std::min<long long>(w, p->ns);
        block_pixels[(size_t)b] = (long long)p->ms * std::min<long long>(w, p->ns);
    }
    if (all_narrow)
        return prepare_only ? CS_OK : cs_detect_foci_batch(ctx, stream_, n_blocks, signals, kernel, params, foci, h_foci, cap, h_n_foci, h_windows);
    if (params[0].compute_dtype != CS_F32) return fail(ctx, CS_ERR_UNSUPPORTED, "the 2-D batch runs the float32 tile kernel");
    if (kernel->km != kernel->kn || kernel->km > 17 || kernel->km < 3 || !(kernel->km & 1))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the masked tile kernel takes odd square templates of 3 .. 17 (caller: block by block)");
    // results straight into page-locked caller buffers (cs_host_alloc)
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win = (foci[0].want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    if ((cap > 0 && !rec) || (foci[0].want_windows && h_windows && !win))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry writes into page-locked buffers (cs_host_alloc)");
    const size_t cnt_bytes = 8 * ((size_t)n_blocks + 2);
    if (cnt_bytes > ctx->h_blk_bytes) {
        if (ctx->h_blk_counts) CS_HIP(ctx, hipHostFree(ctx->h_blk_counts));
        ctx->h_blk_counts = nullptr;
        ctx->h_blk_bytes = 0;
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_blk_counts, 2 * cnt_bytes, hipHostMallocDefault));
        ctx->h_blk_bytes = 2 * cnt_bytes;
    }
    laps.lap("checks, pinned views");
    std::vector<cs::CorrArgs<double>> tab;       // the float64 argument blocks of the chain behind the tile kernels
    // ---- candidates of every block into one list
    const double margin = std::max(foci[0].rescore_margin, 1e-4);
    const double thr = foci[0].pearson - margin;
    size_t c_cap = std::max<size_t>(1 << 16, (size_t)(pixels / 256));
    // (tests of the retry and fall-back paths: a first room for the candidates / a bound for the deferred chain's launches that
    // this call's lists outgrow)
    long long seg_min = 2048;
    if (const char* t = std::getenv("CHROMOSIGHT_HIP_TEST_CAND_CAP")) {
        c_cap = (size_t)std::max(1, atoi(t));
        seg_min = 64;
    }
    const long long test_bound = std::getenv("CHROMOSIGHT_HIP_TEST_DEFER_BOUND") ? atoll(std::getenv("CHROMOSIGHT_HIP_TEST_DEFER_BOUND")) : 0;
    // what the previous call on this context saw: when it scanned the same layout (a run's steps, an iterated template's
    // passes) its candidate count sizes the LAUNCHES of the chain that is enqueued before this call's count is known (below);
    // nothing but a size is carried over
    const bool same_layout = ctx->cand_hint > 0 && ctx->cand_hint_pixels == (long long)pixels && ctx->cand_hint_blocks == n_blocks;
    const long long hint = same_layout ? ctx->cand_hint : 0;
    struct HintUpdate {                     // every successful exit records what this call saw
        cs_ctx* c;
        long long* n;
        long long px;
        int nb;
        bool on;
        bool paced;
        ~HintUpdate() { if (on) { c->cand_hint = *n; c->cand_hint_pixels = px; c->cand_hint_blocks = nb; c->cand_hint_paced = paced; } }
    };
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    long long n_total = 0;
    HintUpdate hint_update{ctx, &n_total, (long long)pixels, n_blocks, !prepare_only, same_layout && ctx->cand_hint_paced};
    // SEGMENTED candidate lists: every block appends to a region of its own with a counter of its own (room in proportion to
    // its pixels), so the chain behind the tile kernels knows the blocks' segments from n_blocks counts instead of sorting
    // all candidates by block and position -- the labelling workgroup of a block sorts its own few thousand in LDS (cs_foci.hip
    // lds_sort_pairs): no device-wide sort, no segments / split kernels.  CHROMOSIGHT_HIP_NO_SEGMENTED=1: one list, sorted on the device.
    const bool segmented_ok = n_blocks <= 60 && !std::getenv("CHROMOSIGHT_HIP_NO_SEGMENTED");
    if (segmented_ok && !ctx->h_cand_counts)
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_cand_counts, 256 * sizeof(long long), hipHostMallocDefault));
    // (seg_tab: the n_blocks + 1 segment starts of the compact numbering, then the n_blocks region starts: one upload)
    std::vector<long long> seg_tab(2 * (size_t)n_blocks + 2, 0), seg_cap((size_t)n_blocks, 0);
    long long* const seg_off = seg_tab.data();
    long long* const seg_base = seg_tab.data() + n_blocks + 1;
    constexpr size_t kCntBytes = 8 * 64;           // the one list's counter (and a spare word) + up to 60 blocks' own
    size_t off_tail_now = 0;                        // where the chain's scratch starts in the current layout
    bool segmented = false;
    bool pass_again = false;                        // a list outgrew its room: this pass does everything itself, whatever was prepared
    while (true) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        size_t list_cap = c_cap;
        if (segmented_ok) {
            list_cap = 0;
            for (int b = 0; b < n_blocks; ++b) {
                seg_cap[(size_t)b] = std::max<long long>(seg_min, (long long)((double)c_cap * (double)block_pixels[(size_t)b] / (double)std::max<long long>(pixels, 1)) + 1);
                seg_base[(size_t)b] = (long long)list_cap;
                list_cap += (size_t)seg_cap[(size_t)b];
            }
        }
        const size_t off_cnt = al(8 * list_cap), off_tail = off_cnt + 1024;
        off_tail_now = off_tail;
        segmented = segmented_ok;
        int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, off_tail + cs::keyed_batch_scratch_bytes(n_blocks, (long long)list_cap));
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        unsigned long long* d_cnt = (unsigned long long*)(pool + off_cnt);       // [0] the one list's counter; [2 + b] block b's
        // (the counter is zeroed right before the tile kernels go out: on a side lane when the lanes carry the call's
        // preparations, see `early_upload` below)
        // side streams only when no block needs the (single) narrowing scratch
        bool twins = signals_f32 != nullptr;
        for (int b = 0; b < n_blocks && twins; ++b) twins = signals_f32[b].d_ptr != nullptr;
        // Lanes: the blocks' persistent launches run side by side on n_lanes streams -- a launch then walks n_lanes times as
        // many tiles per workgroup (pipeline fill / drain and the rounding to whole tiles per workgroup are paid per launch),
        // and no launch waits for slots another one holds.
        constexpr int lanes_env = 3;       // (2 .. 6 lanes measured on the 23-block genome, tools/c4_mode_sweep.sh: flat between 3 and 6)
        const int n_lanes = (twins && n_blocks > 1) ? std::min(std::min(std::max(lanes_env, 1), kBlkLanes), n_blocks) : 1;
        // one persistent launch for the tiles of all blocks (the lanes then only carry the blocks' mask tables): when the caller
        // says nothing else is queued beside it, or for a few blocks (a rank's share of a genome on 8 GPUs: 3 blocks -- one
        // launch beats three that fight for the slots: 1.40 -> 1.29 ms per rank)
        const bool table = n_lanes > 1 && (foci[0].exclusive != 0 || n_blocks <= 4);
        // either way the blocks' launches are PREPARED first (mask tables of every block on the lanes, argument blocks in a host
        // table) and the tile kernels go out afterwards: a mask-table launch queued behind persistent tile kernels would wait
        // for their workgroups to drain, and its lane's next tile kernel with it
        const bool prepared = n_lanes > 1;
        if (!prepared) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));      // (tile kernels go out block by block below)
        int table_rsym = -1;
        if (prepared) {
            const size_t need = al(cs::mfma_blocks_table_bytes(n_blocks)) + cs::mask_prep_table_bytes(n_blocks);
            if (need > ctx->tab_bytes) {
                CS_HIP(ctx, hipDeviceSynchronize());
                if (ctx->h_tab) CS_HIP(ctx, hipHostFree(ctx->h_tab));
                if (ctx->d_tab) CS_HIP(ctx, hipFree(ctx->d_tab));
                ctx->h_tab = ctx->d_tab = nullptr;
                ctx->tab_bytes = 0;
                CS_HIP(ctx, hipHostMalloc(&ctx->h_tab, 2 * need, hipHostMallocDefault));
                CS_HIP(ctx, hipMalloc(&ctx->d_tab, 2 * need));
                ctx->tab_bytes = 2 * need;
            }
            if (ctx->ws_tab.size() < (size_t)n_blocks) {
                ctx->ws_tab.resize((size_t)n_blocks, nullptr);
                ctx->ws_tab_bytes.resize((size_t)n_blocks, 0);
            }
        }
        // (a layout whose lists went to the host-paced chain last time -- a block with more candidates than the labelling
        // workgroup's LDS arrays hold -- goes there directly)
        const bool deferred = segmented && cs::keyed_batch_deferred_available() && !(same_layout && ctx->cand_hint_paced) &&
                              !std::getenv("CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN");
        if (tab.empty()) {
            // the float64 argument blocks of the chain behind the tile kernels: built (and, for the chain that is enqueued ahead
            // of the counts, uploaded) while the caller's stream is still staging the maps
            tab.resize((size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) {
                int rc2 = build_args<double>(ctx, stream, signals + b, kernel, params + b, &tab[b]);
                if (rc2) return rc2;
            }
        }
        bool early_tables = false;
        // blocks to lanes: largest first onto the least loaded lane
        std::vector<int> lane_of((size_t)n_blocks, 0);
        if (n_lanes > 1) {
            std::vector<int> order((size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) order[b] = b;
            auto cost = [&](int b) { return (long long)params[b].ms * (foci[b].hi_diag - foci[b].lo_diag + 1); };
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
            long long load[kBlkLanes] = {};
            for (int b : order) {
                int best = 0;
                for (int k = 1; k < n_lanes; ++k)
                    if (load[k] < load[best]) best = k;
                lane_of[b] = best;
                load[best] += cost(b);
            }
        }
        if (n_lanes > 1) {
            for (int k = 0; k < n_lanes - 1; ++k)
                if (!ctx->s_blk[k]) {
                    // (the lanes carry the short mask-table launches beside the caller's staging kernels, which fill every
                    // wave slot of the chip: served first, they are done when the staging is)
                    int lo_p = 0, hi_p = 0;
                    CS_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
                    CS_HIP(ctx, hipStreamCreateWithPriority(&ctx->s_blk[k], hipStreamNonBlocking, hi_p));
                }
            for (int k = 0; k < kBlkLanes; ++k)
                if (!ctx->ev_blk[k]) CS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_blk[k], hipEventDisableTiming));
            // template weights / matrix-core fragments are uploaded on the caller's stream (once per template): before the
            // side streams are released
            const long long uploads_before = ctx->uploads;
            {
                cs::CorrArgs<float> A0;
                int rc0 = build_args<float>(ctx, stream, signals_f32, kernel, params, &A0);
                if (rc0) return rc0;
                cs::MfmaWeights E0;
                rc0 = ensure_wfrag(ctx, stream, A0.km, A0.kn, &E0);
                if (rc0) return rc0;
            }
            // The mask tables depend on the bins' flags and the template only -- not on the maps, which the caller's stream
            // may still be staging (a genome step enqueues this call right behind cs_stage_blocks): with the launches
            // prepared first, every block's tables are built on the SIDE lanes at once, beside whatever the caller's stream is
            // doing, and the lanes wait for that stream only when this call uploaded the template.  (The tables' scratch is
            // free: the previous call's tile kernels were synchronised before it returned.)
            early_tables = prepared && !std::getenv("CHROMOSIGHT_HIP_NO_EARLY_TABLES");
            // (without early tables the lanes' only link to this stream is the event below: the counter is zeroed before it)
            if (prepared && !early_tables) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));
            if (!early_tables || ctx->uploads != uploads_before) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[kBlkLanes - 1], stream));  // the counter is zero, earlier work is done
                for (int k = 0; k < n_lanes - 1; ++k) CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[kBlkLanes - 1], 0));
            }
        }
        // The PREPARE form (foci[0].reserved & 2): everything of this call that does not depend on the maps -- the blocks' mask
        // tables, the zeroed counters, the tile kernels' argument table and the chain's -- is enqueued on the side lanes NOW,
        // and the call returns; the same call without the flag, made next on this context, finds it done.  A genome step makes
        // the prepare form BEFORE cs_stage_blocks: the lanes' work is then long finished when the staging is, and the tile
        // launch behind the staging waits for events that have fired -- a wait for an event that fires later costs the queue
        // 20-40 us (profiles/r05_rank_share_timeline.txt: staging done at 76 us, lanes at 93, tile kernel at 134).
        const bool can_split = table && prepared && early_tables && deferred;
        unsigned long long key = 1469598103934665603ull;
        {
            auto mix = [&](const void* p, size_t n) {
                const unsigned char* q = (const unsigned char*)p;
                for (size_t k = 0; k < n; ++k) key = (key ^ q[k]) * 1099511628211ull;
            };
            mix(params, sizeof(cs_normxcorr2_params) * (size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) {
                cs_foci_params f = foci[b];
                f.reserved = 0;
                mix(&f, sizeof(f));
            }
            mix(signals, sizeof(cs_matrix) * (size_t)n_blocks);
            if (signals_f32) mix(signals_f32, sizeof(cs_matrix) * (size_t)n_blocks);
            mix(kernel, sizeof(cs_kernel));
            const long long extra[4] = {ctx->uploads, (long long)c_cap, (long long)n_blocks, (long long)(uintptr_t)ctx->d_pool};
            mix(extra, sizeof(extra));
        }
        if (prepare_only && !can_split) return CS_OK;
        const bool reuse = !prepare_only && was_pending && can_split && key == ctx->prep_key && !pass_again;
        struct SkipLaunch {
            cs_ctx* c;
            SkipLaunch(cs_ctx* c_, bool on) : c(c_) { c->skip_prep_launch = on; }
            ~SkipLaunch() { c->skip_prep_launch = false; }
        } skip_launch(ctx, reuse);
        // (one persistent tile launch: ONE side lane carries all of it -- every event the launch waits for costs the caller's
        // queue ~ 7 us between the staging and the tile kernel, and the lane's work is off the critical path)
        const int side_lanes = (table && early_tables) ? 1 : n_lanes - 1;
        laps.lap("pool, weights, events");
        // One tile launch + the lanes busy with the mask tables while the caller's stream is still staging the maps: the zeroed
        // counter (first) and the argument table (behind the tables of its lane) travel on the LAST side lane, which carries the
        // fewest tables -- ordered before the launch by the lanes' events below -- instead of sitting between the staging and
        // the tile kernel
        const bool early_upload = table && early_tables && n_lanes > 1;
        // (one mask-table launch for all blocks: it clears the counters and its upload carries the tile launch's table too -- below)
        const bool prep_carries = early_upload && table && early_tables && side_lanes == 1;
        if (early_upload && !reuse && !prep_carries) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, ctx->s_blk[side_lanes - 1]));
        // (largest blocks first on every lane: the short ones fill the end)
        std::vector<int> launch_order((size_t)n_blocks);
        for (int b = 0; b < n_blocks; ++b) launch_order[b] = b;
        if (n_lanes > 1)
            std::stable_sort(launch_order.begin(), launch_order.end(), [&](int x, int y) {
                return (long long)params[x].ms * (foci[x].hi_diag - foci[x].lo_diag + 1) > (long long)params[y].ms * (foci[y].hi_diag - foci[y].lo_diag + 1);
            });
        // one tile launch: the mask tables of all blocks from ONE launch on the side lane (collected in the loop, launched
        // behind it) -- a launch per block takes 30-90 us each beside a genome's staging kernels
        const bool one_prep = table && early_tables;
        std::vector<cs::MaskPrepArgs<float>> prep_list;
        struct PrepCollect {
            cs_ctx* c;
            PrepCollect(cs_ctx* c_, std::vector<cs::MaskPrepArgs<float>>* v) : c(c_)
            {
                c->prep_collect = v;
                c->prep_groups.clear();
            }
            ~PrepCollect() { c->prep_collect = nullptr; }
        } prep_collect(ctx, one_prep ? &prep_list : nullptr);
        int next_side = 0;
        for (int b : launch_order) {
            // (early tables: on the side lanes only, dealt round-robin in launch order -- largest first)
            const int lane = early_tables ? 1 + (next_side++) % side_lanes : lane_of[b];
            hipStream_t stream = lane == 0 ? (hipStream_t)stream_ : ctx->s_blk[lane - 1];
            struct WsSwap {                   // the side lanes build their mask tables in their own scratch
                cs_ctx* c;
                int k;
                WsSwap(cs_ctx* c_, int k_) : c(c_), k(k_) { swap(); }
                ~WsSwap() { swap(); }
                void swap()
                {
                    if (k < 0) return;
                    std::swap(c->d_ws, c->ws_alt[k]);
                    std::swap(c->d_ws_bytes, c->ws_alt_bytes[k]);
                }
            } ws_swap(ctx, prepared ? -1 : lane - 1);
            struct TabSwap {                  // table mode: every block keeps its own mask tables until the one launch is done
                cs_ctx* c;
                int b;
                TabSwap(cs_ctx* c_, int b_) : c(c_), b(b_) { swap(); }
                ~TabSwap() { swap(); }
                void swap()
                {
                    if (b < 0) return;
                    std::swap(c->d_ws, c->ws_tab[(size_t)b]);
                    std::swap(c->d_ws_bytes, c->ws_tab_bytes[(size_t)b]);
                }
            } tab_swap(ctx, prepared ? b : -1);
            const cs_matrix* sig = (signals_f32 && signals_f32[b].d_ptr) ? signals_f32 + b : signals + b;
            cs_matrix map;                        // geometry of the (virtual) coefficient map: the scanned diagonals
            map.d_ptr = nullptr;
            map.dtype = CS_F32;
            map.layout = is_band(signals[b].layout) ? CS_LAYOUT_BAND : signals[b].layout;
            map.row0 = 0;
            if (map.layout == CS_LAYOUT_BAND) {
                map.band_lo = foci[b].lo_diag;
                map.band_w = foci[b].hi_diag - foci[b].lo_diag + 1;
                map.ld = ((int64_t)map.band_w + 63) / 64 * 64;
            } else {
                map.band_lo = map.band_w = 0;
                map.ld = ((int64_t)params[b].ns + 15) / 16 * 16;
            }
            CandSink sink{(unsigned long long*)pool, d_cnt, (long long)c_cap, (unsigned long long)b << kKeyShift, foci[b].lo_diag,
                          foci[b].hi_diag};
            if (segmented) {
                sink.keys = (unsigned long long*)pool + seg_base[(size_t)b];
                sink.count = d_cnt + 2 + b;
                sink.cap = seg_cap[(size_t)b];
            }
            int rsym = 0;
            if (prepared) {
                sink.defer_args = (char*)ctx->h_tab + cs::mfma_blocks_arg_offset(n_blocks) + (size_t)b * cs::mfma_blocks_arg_bytes();
                sink.defer_rsym = &rsym;
            }
            rc = corr_candidates_f32(ctx, stream, sig, kernel, params + b, &map, margin, thr, &sink);
            if (!rc && prepared) {
                if (!ctx->cand_fused) rc = CS_NEED_MAP;             // another kernel than the masked tile kernel took the block
                else if (table_rsym >= 0 && table_rsym != rsym) rc = fail(ctx, CS_ERR_UNSUPPORTED, "blocks need different tile kernels");
                table_rsym = rsym;
            }
            if (rc) {
                if (n_lanes > 1) (void)hipDeviceSynchronize();      // nothing of this call may still be running on a side stream
                if (rc == CS_NEED_MAP) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d is not served by the masked tile kernel", b);
                return rc;
            }
        }
        if (one_prep) {
            ctx->prep_collect = nullptr;
            if (!reuse) {
                const size_t tile_tab_bytes = al(cs::mfma_blocks_table_bytes(n_blocks));
                int rcp = 0;
                if (prep_carries) {
                    // the lane the tile launch waits for: ONE upload (tile table + mask-table arguments, neighbours in h_tab / d_tab)
                    // and ONE kernel that also clears the counters, instead of memset, upload, kernel, upload
                    rcp = cs::mfma_blocks_table_finish(ctx->h_tab, n_blocks);
                    if (rcp) return fail(ctx, CS_ERR_OVERFLOW, "too many tiles for one launch");
                    rcp = cs::launch_mask_prep_batch(prep_list.data(), ctx->prep_groups.data(), (int)prep_list.size(), (char*)ctx->h_tab + tile_tab_bytes,
                                                     (char*)ctx->d_tab + tile_tab_bytes, ctx->s_blk[0], tile_tab_bytes, d_cnt, kCntBytes);
                } else {
                    rcp = cs::launch_mask_prep_batch(prep_list.data(), ctx->prep_groups.data(), (int)prep_list.size(), (char*)ctx->h_tab + tile_tab_bytes,
                                                     (char*)ctx->d_tab + tile_tab_bytes, ctx->s_blk[0]);
                }
                if (rcp) return fail(ctx, CS_ERR_HIP, "mask table kernel failed: %s", hipGetErrorString((hipError_t)rcp));
            }
        }
        laps.lap("mask tables + arguments");
        bool tab_uploaded = false;
        if (early_upload) {
            // (reuse: both tables were uploaded by the prepare form; the host-side table is still filled in -- the launch reads
            // its block count and tile ranges from it)
            rc = cs::launch_corr_mfma_blocks(ctx->h_tab, ctx->d_tab, n_blocks, table_rsym, ctx->n_cu, stream, ctx->s_blk[side_lanes - 1], !reuse && !prep_carries, false);
            if (rc) return fail(ctx, CS_ERR_HIP, "tile kernel table upload failed: %s", hipGetErrorString((hipError_t)rc));
            if (deferred) {
                if (!reuse) rc = cs::upload_keyed_batch_table(tab.data(), n_blocks, (long long)list_cap, pool + off_tail, ctx->s_blk[side_lanes - 1]);
                if (rc) return fail(ctx, CS_ERR_HIP, "argument table upload failed: %s", hipGetErrorString((hipError_t)rc));
                tab_uploaded = true;
            }
            if (prepare_only) {
                ctx->prep_tab_keep.swap(tab);           // (the source of the asynchronous upload above outlives this call)
                ctx->prep_key = key;
                ctx->prep_pending = true;
                return CS_OK;
            }
        } else if (prepared && early_tables) {
            CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));      // (before the event the lanes' tile kernels wait for, below)
        }
        if (early_tables) {
            // the tile kernels read the maps and the zeroed counter: behind the caller's stream.  A block whose tables were
            // built on another lane than the one that launches its tiles: every lane waits for every lane's tables.
            if (!table) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[kBlkLanes - 1], stream));
                for (int k = 0; k < n_lanes - 1; ++k) CS_HIP(ctx, hipEventRecord(ctx->ev_blk[k], ctx->s_blk[k]));
                for (int k = 0; k < n_lanes - 1; ++k) {
                    CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[kBlkLanes - 1], 0));
                    for (int j = 0; j < n_lanes - 1; ++j)
                        if (j != k) CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[j], 0));
                    CS_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_blk[k], 0));
                }
            }
        }
        if (prepared && !table) {
            // one persistent launch per block, each on the lane that built its mask tables
            for (int b : launch_order) {
                hipStream_t s_lane = lane_of[b] == 0 ? stream : ctx->s_blk[lane_of[b] - 1];
                const void* arg = (const char*)ctx->h_tab + cs::mfma_blocks_arg_offset(n_blocks) + (size_t)b * cs::mfma_blocks_arg_bytes();
                rc = cs::launch_corr_mfma_prepared(arg, table_rsym, ctx->n_cu, 0, s_lane);
                if (rc) {
                    (void)hipDeviceSynchronize();
                    return fail(ctx, CS_ERR_HIP, "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
                }
            }
        }
        if (n_lanes > 1) {
            for (int k = 0; k < (table && early_tables ? side_lanes : n_lanes - 1); ++k) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[k], ctx->s_blk[k]));
                CS_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_blk[k], 0));
            }
        }
        if (table) {
            rc = cs::launch_corr_mfma_blocks(ctx->h_tab, ctx->d_tab, n_blocks, table_rsym, ctx->n_cu, stream, stream, !early_upload, true,
                                             tile_epoch ? ctx->d_tiles_started : nullptr, tile_epoch);
            if (rc) return fail(ctx, CS_ERR_HIP, "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        } else if (tile_epoch) {
            // (the per-block launches carry no start word: whoever waits for this epoch -- cs_stream_wait_tiles -- is let go here)
            CS_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->d_tiles_started, (int)tile_epoch, 1, stream));
        }
        laps.lap("tile kernels launched");
        // ---- The chain behind the tile kernels, enqueued BEFORE they have finished (segmented lists): the blocks' candidate
        // counts stay on the device -- one tiny kernel turns them into the segments of the compact numbering --, the launches
        // are sized for a bound (the previous call's count on the same layout + 50 %; without one, a share of the lists'
        // room), and the host reads counts, status and records after ONE synchronisation.  What the round trip in the middle
        // cost a rank's share of a genome: the wake-up, ~ 30 us of enqueueing with the device idle, of a 160 us tail
        // (profiles/r05_rank_share_timeline.txt).  A list that outgrew its room sends the call round again as before; lists too
        // long for the labelling workgroups' LDS arrays, or more candidates than the bound, take the host-paced chain below
        // on the same lists.  CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN=1: always the host-paced chain.
        bool counts_known = false;
        if (deferred) {
            long long bound = std::min<long long>((long long)list_cap, hint > 0 ? hint + hint / 2 + 4096
                                                                               : std::max<long long>(16384, (long long)list_cap / 8));
            if (test_bound > 0) bound = std::min(bound, test_bound);
            cs::DeferredSegments D;
            D.d_counts = (const long long*)(d_cnt + 2);
            for (int b = 0; b < n_blocks; ++b) {                 // (the device reads the two tables where they are: page-locked)
                ctx->h_cand_counts[64 + b] = seg_base[(size_t)b];
                ctx->h_cand_counts[128 + b] = seg_cap[(size_t)b];
            }
            D.h_base = ctx->h_cand_counts + 64;
            D.h_cap = ctx->h_cand_counts + 128;
            D.bound = bound;
            D.tab_uploaded = tab_uploaded;
            D.h_counts_out = ctx->h_cand_counts;             // [0, n_blocks): the blocks' counts; [60], [61]: total, status flags
            ctx->h_blk_counts[0] = -1;
            ctx->h_cand_counts[61] = -1;
            int rc1 = cs::enqueue_foci_keyed_batch(tab.data(), n_blocks, (const long long*)pool, (long long)list_cap, kKeyShift, foci[0].pearson,
                                                   foci[0].min_size, diag_code(foci, kernel, params), foci[0].inter, pool + off_tail, rec,
                                                   (long long)cap, win, win ? (long long)cap : 0, ctx->h_blk_counts, stream, nullptr, nullptr, &D);
            if (rc1) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc1));
            laps.lap("foci chain enqueued");
            CS_HIP(ctx, hipStreamSynchronize(stream));
            laps.lap("wait: records");
            counts_known = true;
        } else if (segmented) {
            CS_HIP(ctx, hipMemcpyAsync(ctx->h_cand_counts, d_cnt + 2, 8 * (size_t)n_blocks, hipMemcpyDeviceToHost, stream));
        } else {
            CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        }
        if (!counts_known) {
            CS_HIP(ctx, hipStreamSynchronize(stream));
            laps.lap("wait: candidates");
        }
        if (segmented) {
            // the blocks' own counts: segments of the compact numbering; a block that outgrew its room sends the call round again
            bool fits = true;
            long long sum = 0, worst = 1;
            for (int b = 0; b < n_blocks; ++b) {
                const long long nb_ = ctx->h_cand_counts[b];
                seg_off[(size_t)b] = sum;
                sum += std::min(nb_, seg_cap[(size_t)b]);
                if (nb_ > seg_cap[(size_t)b]) {
                    fits = false;
                    // (room in proportion to the pixels: the total that would have given this block enough)
                    worst = std::max(worst, (long long)((double)nb_ * (double)std::max<long long>(pixels, 1) / (double)std::max<long long>(block_pixels[(size_t)b], 1)) + 1);
                }
            }
            seg_off[(size_t)n_blocks] = sum;
            n_total = sum;
            if (!fits) {
                c_cap = (size_t)worst + (size_t)worst / 8;
                pass_again = true;
                continue;
            }
            if (counts_known && ctx->h_cand_counts[61] == 0) {
                // the chain ran on this call's lists as they are: done
                const long long total = ctx->h_blk_counts[0];
                if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
                for (int b = 0; b < n_blocks; ++b) h_n_foci[b] = ctx->h_blk_counts[1 + b];
                if (total > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)cap);
                return CS_OK;
            }
            if (counts_known) hint_update.paced = (ctx->h_cand_counts[61] & 2) != 0;
            break;
        }
        n_total = ctx->h_counts[0];
        if ((size_t)n_total <= c_cap) break;
        c_cap = (size_t)n_total + (size_t)n_total / 8;
        pass_again = true;
    }
    // ---- exact scores, foci, statistics: one chain for all blocks (its argument table was built while the tile kernels ran),
    // sized by the counts the host has read
    char* pool = (char*)ctx->d_pool;
    const size_t off_tail = off_tail_now;
    ctx->h_blk_counts[0] = -1;
    int rc = cs::enqueue_foci_keyed_batch(tab.data(), n_blocks, (const long long*)pool, n_total, kKeyShift, foci[0].pearson,
                                          foci[0].min_size, diag_code(foci, kernel, params), foci[0].inter, pool + off_tail, rec,
                                          (long long)cap, win, win ? (long long)cap : 0, ctx->h_blk_counts, stream,
                                          segmented ? seg_base : nullptr, segmented ? seg_off : nullptr, nullptr);
    if (rc) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    laps.lap("foci chain enqueued");
    CS_HIP(ctx, hipStreamSynchronize(stream));            // also: the host table above was consumed
    laps.lap("wait: records");
    const long long total = ctx->h_blk_counts[0];
    if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
    for (int b = 0; b < n_blocks; ++b) h_n_foci[b] = ctx->h_blk_counts[1 + b];
    if (total > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)cap);
    return CS_OK;
}

int cs_candidates(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* p, const cs_foci_params* fp, int32_t* h_rows, int32_t* h_cols,
                  double* h_vals, int64_t cap, int64_t* n_out)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (!n_out || cap < 0 || (cap > 0 && (!h_rows || !h_cols || !h_vals))) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    *n_out = 0;
    // ---- One synchronisation for the usual case (a 2-D pattern in float32 arithmetic on a kernel with a candidate sink, a few hundred
    // candidates): tile kernel -> the first n_first keys decoded and re-scored in float64, bounded by the DEVICE's count, written with the count
    // into a mapped host block -> threshold and order on the host.  The general flow below reads the count first (to size
    // what follows), sorts and compacts on the device and synchronises three times: 0.14 ms that a rank's share of a row-split block
    // waited for behind a 0.23 ms tile kernel (profiles/r06_c4p_split_shares.txt).  A list longer than the key list's capacity, or a
    // kernel without a sink, takes the general flow.
    const bool band_sig = signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED;
    if (p->compute_dtype == CS_F32 && !(band_sig && fp->hi_diag - fp->lo_diag + 1 <= 4) && fp->hi_diag >= fp->lo_diag &&
        !std::getenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP")) {
        constexpr long long kHeads = 32768;                      // room of the host block and of the heads' device arrays
        // how many keys the ONE launch covers: the previous call's count on this context and a quarter (a whole 200 000-bin block
        // has ~ 15 000 candidates before the float64 scores thin them out; its second call needs no second round), at least 8192
        const long long n_first = std::min<long long>(kHeads, std::max<long long>(8192, ctx->cand_prev + ctx->cand_prev / 4 + 64));
        int rb = 0, re = p->ms;
        if (p->row_end > p->row_begin) {
            if (p->row_begin < 0 || p->row_end > p->ms) return fail(ctx, CS_ERR_INVALID, "row window outside the matrix");
            rb = p->row_begin;
            re = p->row_end;
        }
        cs_matrix map;                                          // the output geometry of the sink (no storage)
        map.d_ptr = nullptr;
        map.dtype = CS_F32;
        map.layout = band_sig ? CS_LAYOUT_BAND : signal->layout;
        map.row0 = rb;
        map.band_lo = band_sig ? fp->lo_diag : 0;
        map.band_w = band_sig ? fp->hi_diag - fp->lo_diag + 1 : 0;
        map.ld = band_sig ? ((int64_t)map.band_w + 63) / 64 * 64 : ((int64_t)p->ns + 15) / 16 * 16;
        const bool plain_layout = band_sig || signal->layout == CS_LAYOUT_DENSE;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t c_cap = std::max<size_t>(1 << 16, (size_t)(re - rb) * (size_t)(band_sig ? map.band_w : p->ns) / 256);
        // device scratch: keys[c_cap] | count | (heads: in the host block) | a second round's rows, cols, vals
        // ('count' = the copy of the device's counter that travels with the heads; the counters themselves: ctx->d_cand_cnt)
        const size_t off_cnt = al(8 * c_cap), off_rows = off_cnt + 256, off_cols = off_rows + al(4 * kHeads), off_vals = off_cols + al(4 * kHeads);
        const size_t rest = c_cap;                             // (a second round holds at most c_cap - 8192)
        const size_t off_rows2 = off_vals + al(8 * kHeads), off_cols2 = off_rows2 + al(4 * rest), off_vals2 = off_cols2 + al(4 * rest);
        const size_t h_bytes = 256 + 4 * kHeads + 4 * kHeads + 8 * kHeads;
        rc = plain_layout ? ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, off_vals2 + al(8 * rest)) : CS_ERR_UNSUPPORTED;
        if (rc == CS_OK && !ctx->h_small) {
            if (hipHostMalloc(&ctx->h_small, h_bytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                ctx->h_small = nullptr;
                rc = CS_ERR_HIP;
            }
        }
        if (rc == CS_OK && !ctx->d_cand_cnt) {
            if (hipMalloc((void**)&ctx->d_cand_cnt, 512) != hipSuccess) {
                (void)hipGetLastError();
                ctx->d_cand_cnt = nullptr;
                rc = CS_ERR_HIP;
            }
            ctx->cand_cnt_clean = false;
        }
        if (rc == CS_OK) {
            char* pool = (char*)ctx->d_pool;
            unsigned long long* d_keys = (unsigned long long*)pool;
            if (!ctx->cand_cnt_clean) CS_HIP(ctx, hipMemsetAsync(ctx->d_cand_cnt, 0, 512, stream));
            ctx->cand_cnt_clean = false;                        // (until this call's kernels are known to have run)
            long long* d_cnt = ctx->d_cand_cnt + 32 * ctx->cand_cnt_phase;
            long long* d_cnt_next = ctx->d_cand_cnt + 32 * (ctx->cand_cnt_phase ^ 1);
            const double margin = std::max(fp->rescore_margin, 1e-4);
            CandSink sink{d_keys, (unsigned long long*)d_cnt, (long long)c_cap, 0ull, fp->lo_diag, fp->hi_diag};
            rc = corr_candidates_f32(ctx, stream, signal, kernel, p, &map, margin, fp->pearson - margin, &sink);
            if (rc == CS_OK) {
                int* h_rows_small = (int*)((char*)ctx->h_small + 256);
                int* h_cols_small = (int*)((char*)ctx->h_small + 256 + (off_cols - off_rows));
                double* h_vals_small = (double*)((char*)ctx->h_small + 256 + (off_vals - off_rows));
                cs::CorrArgs<double> A64;
                rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
                if (rc) return rc;
                // keys -> pixels and float64 scores of the first min(n_first, count) of them; the count lands next to them and the
                // other counter is cleared for the next call
                // -- written by the kernel straight into the page-locked host block (h_small is mapped and coherent: a few hundred
                // 4- and 8-byte stores over the link instead of a 9 us wait for the copy engine and an 8 us copy)
                rc = cs::launch_rescore_f64_keys(A64, (const long long*)d_keys, p->ns, n_first, h_rows_small, h_cols_small, h_vals_small, d_cnt,
                                                 (long long*)ctx->h_small, d_cnt_next, stream);
                if (rc) return fail(ctx, CS_ERR_HIP, "candidate kernels failed: %s", hipGetErrorString((hipError_t)rc));
                char* h = (char*)ctx->h_small;
                CS_HIP(ctx, hipStreamSynchronize(stream));
                ctx->cand_cnt_phase ^= 1;
                ctx->cand_cnt_clean = true;
                const long long n_cand = *reinterpret_cast<const long long*>(h);
                ctx->cand_prev = n_cand;
                if (n_cand <= (long long)c_cap) {
                    const int32_t* rows = reinterpret_cast<const int32_t*>(h + 256);
                    const int32_t* cols = reinterpret_cast<const int32_t*>(h + 256 + (off_cols - off_rows));
                    const double* vals = reinterpret_cast<const double*>(h + 256 + (off_vals - off_rows));
                    // a list longer than the heads (a whole 200 000-bin block: ~ 15 000 candidates before the float64 scores thin them
                    // out): the rest is decoded and re-scored from the SAME key list -- the tile kernel does not run again -- and
                    // arrives with a second synchronisation
                    std::vector<int32_t> rows2, cols2;
                    std::vector<double> vals2;
                    const long long m2 = std::max(0ll, n_cand - n_first);
                    if (m2 > 0) {
                        int* d_r2 = (int*)(pool + off_rows2);
                        int* d_c2 = (int*)(pool + off_cols2);
                        double* d_v2 = (double*)(pool + off_vals2);
                        rc = cs::launch_decode_keys((const long long*)d_keys + n_first, m2, p->ns, d_r2, d_c2, stream);
                        if (rc) return fail(ctx, CS_ERR_HIP, "key decoding failed: %s", hipGetErrorString((hipError_t)rc));
                        rc = cs::launch_rescore_f64(A64, d_r2, d_c2, m2, d_v2, nullptr, stream);
                        if (rc) return fail(ctx, CS_ERR_HIP, "candidate kernels failed: %s", hipGetErrorString((hipError_t)rc));
                        rows2.resize((size_t)m2);
                        cols2.resize((size_t)m2);
                        vals2.resize((size_t)m2);
                        CS_HIP(ctx, hipMemcpyAsync(rows2.data(), d_r2, 4 * (size_t)m2, hipMemcpyDeviceToHost, stream));
                        CS_HIP(ctx, hipMemcpyAsync(cols2.data(), d_c2, 4 * (size_t)m2, hipMemcpyDeviceToHost, stream));
                        CS_HIP(ctx, hipMemcpyAsync(vals2.data(), d_v2, 8 * (size_t)m2, hipMemcpyDeviceToHost, stream));
                        CS_HIP(ctx, hipStreamSynchronize(stream));
                    }
                    auto row_of = [&](uint32_t t) { return t < (uint32_t)n_first ? rows[t] : rows2[t - (uint32_t)n_first]; };
                    auto col_of = [&](uint32_t t) { return t < (uint32_t)n_first ? cols[t] : cols2[t - (uint32_t)n_first]; };
                    auto val_of = [&](uint32_t t) { return t < (uint32_t)n_first ? vals[t] : vals2[t - (uint32_t)n_first]; };
                    std::vector<uint32_t> keep;
                    keep.reserve((size_t)n_cand);
                    for (long long t = 0; t < n_cand; ++t)
                        if (val_of((uint32_t)t) >= fp->pearson && val_of((uint32_t)t) != 0.0) keep.push_back((uint32_t)t);      // (flag_keep_kernel's rule)
                    std::sort(keep.begin(), keep.end(),
                              [&](uint32_t a, uint32_t b) { return row_of(a) != row_of(b) ? row_of(a) < row_of(b) : col_of(a) < col_of(b); });
                    *n_out = (int64_t)keep.size();
                    if ((int64_t)keep.size() > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld candidates, room for %lld", (long long)keep.size(), (long long)cap);
                    for (size_t t = 0; t < keep.size(); ++t) {
                        h_rows[t] = row_of(keep[t]);
                        h_cols[t] = col_of(keep[t]);
                        h_vals[t] = val_of(keep[t]);
                    }
                    return CS_OK;
                }
                // (more candidates than the key list holds: the general flow, which sizes its list by the count)
            } else if (rc != CS_NEED_MAP) {
                return rc;
            }
        }
    }
    CandPlan P;
    rc = find_candidates(ctx, stream_, signal, kernel, p, fp, 0, cs::keep_scratch_bytes, &P);
    if (rc) return rc;
    if (P.n_cand == 0) return CS_OK;
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    constexpr long long kSmallList = 16384;
    if (P.n_cand <= kSmallList && !std::getenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP")) {
        // A short candidate list (a 2-D pattern: 1e-5 of the scanned pixels) is re-scored as it came -- unsorted -- downloaded with ONE
        // synchronisation and thresholded and ordered on the host: the device form below costs a dozen launches (keys, a three-pass
        // sort, decode, flags, scan, scatter, decode), a count read-back and a second synchronisation, which is what a rank's share of a
        // row-split block waited for once its tile kernel was done (profiles/r06_c4p_split_shares.txt: 0.14 ms of a 0.40 ms step).
        const size_t m = (size_t)P.n_cand;
        double* d_vals = reinterpret_cast<double*>(pool + P.off_tail);
        rc = cs::launch_rescore_f64(A64, (const int*)pool, (const int*)(pool + P.off_cols), P.n_cand, d_vals, nullptr, stream);
        if (rc) return fail(ctx, CS_ERR_HIP, "candidate kernels failed: %s", hipGetErrorString((hipError_t)rc));
        std::vector<int32_t> rows(m), cols(m);
        std::vector<double> vals(m);
        CS_HIP(ctx, hipMemcpyAsync(rows.data(), pool, 4 * m, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(cols.data(), pool + P.off_cols, 4 * m, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(vals.data(), d_vals, 8 * m, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        std::vector<uint32_t> keep;
        keep.reserve(m);
        for (size_t t = 0; t < m; ++t)
            if (vals[t] >= fp->pearson && vals[t] != 0.0) keep.push_back((uint32_t)t);          // (flag_keep_kernel's rule)
        std::sort(keep.begin(), keep.end(), [&](uint32_t a, uint32_t b) { return rows[a] != rows[b] ? rows[a] < rows[b] : cols[a] < cols[b]; });
        *n_out = (int64_t)keep.size();
        if ((int64_t)keep.size() > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld candidates, room for %lld", (long long)keep.size(), (long long)cap);
        for (size_t t = 0; t < keep.size(); ++t) {
            h_rows[t] = rows[keep[t]];
            h_cols[t] = cols[keep[t]];
            h_vals[t] = vals[keep[t]];
        }
        return CS_OK;
    }
    int *d_rows = nullptr, *d_cols = nullptr, *d_n = nullptr;
    double* d_vals = nullptr;
    rc = cs::enqueue_keep(A64, (const int*)pool, (const int*)(pool + P.off_cols), P.n_cand, fp->pearson, pool + P.off_tail,
                          &d_rows, &d_cols, &d_vals, &d_n, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "candidate kernels failed: %s", hipGetErrorString((hipError_t)rc));
    int* h_n = reinterpret_cast<int*>(ctx->h_counts + 2);
    CS_HIP(ctx, hipMemcpyAsync(h_n, d_n, 4, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long n = *h_n;
    *n_out = n;
    if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld candidates, room for %lld", n, (long long)cap);
    if (n > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_rows, d_rows, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_cols, d_cols, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_vals, d_vals, 8 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_label_foci(cs_ctx* ctx, void* stream_, int32_t ms, int32_t ns, const int32_t* h_rows, const int32_t* h_cols,
                  const double* h_vals, int64_t n, int32_t min_size, int32_t diag_only, int32_t* h_foci_rows,
                  int32_t* h_foci_cols, int32_t* h_foci_size, int64_t cap, int64_t* n_foci)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_foci || n < 0 || ms <= 0 || ns <= 0 || min_size < 1 || (n > 0 && (!h_rows || !h_cols || !h_vals)))
        return fail(ctx, CS_ERR_INVALID, "bad candidate list");
    if (cap < 0 || (cap > 0 && (!h_foci_rows || !h_foci_cols || !h_foci_size))) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    if (n > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", (long long)n);
    *n_foci = 0;
    if (n == 0) return CS_OK;
    if (!ctx->h_counts) CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_counts, 64, hipHostMallocDefault));
    for (int64_t t = 0; t < n; ++t)
        if (h_rows[t] < 0 || h_rows[t] >= ms || h_cols[t] < 0 || h_cols[t] >= ns)
            return fail(ctx, CS_ERR_INVALID, "candidate %lld outside the matrix", (long long)t);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n;
    const size_t off_cols = al(4 * un), off_vals = off_cols + al(4 * un), off_cnt = off_vals + al(8 * un),
                 off_tail = off_cnt + 256;
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, off_tail + cs::label_scratch_bytes(n));
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    CS_HIP(ctx, hipMemcpyAsync(pool, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_vals, h_vals, 8 * un, hipMemcpyHostToDevice, stream));
    long long* d_cnt = (long long*)(pool + off_cnt);
    int *f_rows = nullptr, *f_cols = nullptr, *f_size = nullptr;
    rc = cs::enqueue_label((const int*)pool, (const int*)(pool + off_cols), (const double*)(pool + off_vals), n, ns, min_size,
                           diag_only, pool + off_tail, &f_rows, &f_cols, &f_size, d_cnt, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "labelling kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts + 1, d_cnt, 8, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long k = ctx->h_counts[1];
    *n_foci = k;
    if (k > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", k, (long long)cap);
    if (k > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_foci_rows, f_rows, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_foci_cols, f_cols, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_foci_size, f_size, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_quantify_pixels(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                       const cs_normxcorr2_params* p, const cs_foci_params* fp, const int32_t* h_rows,
                       const int32_t* h_cols, int64_t n, cs_focus* h_out, double* h_windows)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!h_rows || !h_cols || !h_out))) return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    if (n == 0) return CS_OK;
    const int kk = kernel->km * kernel->kn;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n;
    const size_t off_cols = al(4 * un), off_score = off_cols + al(4 * un), off_nobs = off_score + al(8 * un),
                 off_rec = off_nobs + al(8 * un), off_win = off_rec + al(sizeof(cs::FocusRec) * un),
                 total = off_win + (fp->want_windows ? al(8 * un * kk) : 0);
    rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, total);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    CS_HIP(ctx, hipMemcpyAsync(pool, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    double* d_win = fp->want_windows ? (double*)(pool + off_win) : nullptr;
    rc = cs::enqueue_quantify(A64, (const int*)pool, (const int*)(pool + off_cols), n, fp->inter, (double*)(pool + off_score),
                              (double*)(pool + off_nobs), (cs::FocusRec*)(pool + off_rec), d_win, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "quantify kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(h_out, pool + off_rec, sizeof(cs_focus) * un, hipMemcpyDeviceToHost, stream));
    if (d_win && h_windows) CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * un * kk, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    return CS_OK;
}

// quantify mode over the sub-matrices of a genome in ONE launch chain (one call per template instead of one per
// sub-matrix and template: cli/chromosight.py:229-260 scores one sub-matrix per task)
int cs_quantify_blocks(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                       const cs_normxcorr2_params* params, const cs_foci_params* foci, const int32_t* h_blk, const int32_t* h_rows,
                       const int32_t* h_cols, int64_t n, cs_focus* h_out, double* h_windows)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernel || !params || !foci) return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    if (n < 0 || (n > 0 && (!h_blk || !h_rows || !h_cols || !h_out))) return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    for (int b = 0; b < n_blocks; ++b) {
        int rc = check_foci_args(ctx, signals + b, kernel, params + b, foci + b);
        if (rc) return rc;
        if (params[b].compute_dtype != CS_F64) return fail(ctx, CS_ERR_UNSUPPORTED, "quantify scores in float64");
    }
    for (int64_t t = 0; t < n; ++t)
        if (h_blk[t] < 0 || h_blk[t] >= n_blocks) return fail(ctx, CS_ERR_INVALID, "pixel %lld names sub-matrix %d of %d", (long long)t, h_blk[t], n_blocks);
    if (n == 0) return CS_OK;
    const bool want_windows = foci[0].want_windows && h_windows;
    const int kk = kernel->km * kernel->kn;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n, nb = (size_t)n_blocks;
    const size_t off_rows = al(4 * un), off_cols = off_rows + al(4 * un), off_inter = off_cols + al(4 * un),
                 off_tab = off_inter + al(4 * nb), off_score = off_tab + al(sizeof(cs::CorrArgs<double>) * nb),
                 off_nobs = off_score + al(8 * un), off_rec = off_nobs + al(8 * un), off_win = off_rec + al(sizeof(cs::FocusRec) * un),
                 total = off_win + (want_windows ? al(8 * un * kk) : 0);
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, total);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    std::vector<cs::CorrArgs<double>> tab(nb);
    std::vector<int> inter(nb);
    for (int b = 0; b < n_blocks; ++b) {
        rc = build_args<double>(ctx, stream, signals + b, kernel, params + b, &tab[(size_t)b]);
        if (rc) return rc;
        inter[(size_t)b] = foci[b].inter;
    }
    CS_HIP(ctx, hipMemcpyAsync(pool, h_blk, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_rows, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_inter, inter.data(), 4 * nb, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_tab, tab.data(), sizeof(cs::CorrArgs<double>) * nb, hipMemcpyHostToDevice, stream));
    double* d_win = want_windows ? (double*)(pool + off_win) : nullptr;
    rc = cs::enqueue_quantify_batch((const cs::CorrArgs<double>*)(pool + off_tab), (const int*)(pool + off_inter), (const int*)pool,
                                    (const int*)(pool + off_rows), (const int*)(pool + off_cols), n, (double*)(pool + off_score),
                                    (double*)(pool + off_nobs), (cs::FocusRec*)(pool + off_rec), d_win, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "quantify kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(h_out, pool + off_rec, sizeof(cs_focus) * un, hipMemcpyDeviceToHost, stream));
    if (d_win) CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * un * kk, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));       // (the pageable tables above were consumed)
    return CS_OK;
}

}  // extern "C"
