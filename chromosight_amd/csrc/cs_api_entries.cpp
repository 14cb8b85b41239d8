// cs_api_entries.cpp -- the C ABI (include/chromosight_hip.h): the correlation entries (cs_normxcorr2, its host-pipelined form,
// cs_xcorr2, exact re-scoring, compaction), the pixel-table / staging entries, the host-side passes (neighbour removal,
// acceptance rules) and cs_run_calls.  Context, weights and kernel dispatch: cs_api.cpp (cs_api_internal.h).
#include "cs_api_internal.h"

using namespace csapi;

extern "C" {

// --------------------------------------------------------------------------------------------
int cs_normxcorr2(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* p, const cs_matrix* out_corr, const cs_matrix* out_nobs)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(ctx, CS_ERR_INVALID, "null params");
    int rc = check_matrix(ctx, out_corr, "out_corr", p->ns);
    AllowCounts allow_counts(ctx);                   // (the signal only: the outputs were just checked without it)
    if (rc) return rc;
    const bool want_nobs = out_nobs && out_nobs->d_ptr;
    if (want_nobs) {
        rc = check_matrix(ctx, out_nobs, "out_nobs", p->ns);
        if (rc) return rc;
        if (out_nobs->dtype != CS_F32) return fail(ctx, CS_ERR_INVALID, "out_nobs must be float32");
    }
    const bool allow_fast = getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr;
    if (ctx->range_check && signal && signal->d_ptr && kernel) {
        // the guard of cs_ctx_set_range_check: largest |pixel| of the rows this call reads (the reduction of
        // cs_normxcorr2_host's slabs), then CS_ERR_RANGE for a non-finite pixel or, in float32, a magnitude beyond 1e15
        const int kh = (kernel->km - 1) / 2;
        const int rb = (p->row_begin == 0 && p->row_end == 0) ? 0 : p->row_begin, re = (p->row_begin == 0 && p->row_end == 0) ? p->ms : p->row_end;
        const int p_lo = std::max(0, rb - kh), p_hi = std::min(p->ms, re + (kernel->km - 1) - kh);
        const int width = is_band(signal->layout) ? signal->band_w : p->ns;
        if (!ctx->d_counts_peak) {
            CS_HIP(ctx, hipMalloc(&ctx->d_counts_peak, 256));
            CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_peak, 256, hipHostMallocDefault));
        }
        CS_HIP(ctx, hipMemsetAsync(ctx->d_counts_peak, 0, 4, stream));
        const size_t esz = signal->dtype == CS_F64 ? 8 : 4;
        const char* src = (const char*)signal->d_ptr + ((long long)p_lo - signal->row0) * signal->ld * (long long)esz;
        if (p_hi > p_lo && cs::launch_peak_rows(src, signal->dtype == CS_F64, signal->ld, p_hi - p_lo, width, ctx->n_cu,
                                                reinterpret_cast<unsigned*>(ctx->d_counts_peak), stream) != 0)
            return fail(ctx, CS_ERR_HIP, "range reduction failed to launch");
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_peak, ctx->d_counts_peak, 4, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        const float limit = p->compute_dtype == CS_F32 ? 1e15f : 3.4e38f;
        unsigned limit_bits;
        std::memcpy(&limit_bits, &limit, 4);
        if (*ctx->h_peak > limit_bits)
            return fail(ctx, CS_ERR_RANGE, p->compute_dtype == CS_F32 ? "the map holds non-finite pixels or magnitudes beyond 1e15: float64 path"
                                                                      : "the map holds non-finite pixels");
    }
    if (p->compute_dtype == CS_F64) {
        cs::CorrArgs<double> A;
        rc = build_args<double>(ctx, stream, signal, kernel, p, &A);
        if (rc) return rc;
        A.out = view_of(out_corr);
        A.out_is_f64 = out_corr->dtype == CS_F64;
        A.nobs = want_nobs ? view_of(out_nobs) : cs::MatView{nullptr, 0, 0, 0, 0, 0};
        return launch_corr<double>(ctx, A, stream, allow_fast);
    } else if (p->compute_dtype == CS_F32) {
        cs::CorrArgs<float> A;
        rc = build_args<float>(ctx, stream, signal, kernel, p, &A);
        if (rc) return rc;
        A.out = view_of(out_corr);
        A.out_is_f64 = out_corr->dtype == CS_F64;
        A.nobs = want_nobs ? view_of(out_nobs) : cs::MatView{nullptr, 0, 0, 0, 0, 0};
        return launch_corr<float>(ctx, A, stream, allow_fast);
    }
    return fail(ctx, CS_ERR_INVALID, "bad compute dtype");
}

// Host map in, host map out, pipelined over PCIe in row slabs (see the header).
int cs_normxcorr2_host(cs_ctx* ctx, const void* h_signal, int32_t sig_dtype, int64_t ld_in, const cs_kernel* kernel,
                       const cs_normxcorr2_params* p, void* h_out, int32_t out_dtype, int64_t ld_out)
{
    CS_ENTER(ctx);
    if (!p || !h_signal || !h_out || !kernel) return fail(ctx, CS_ERR_INVALID, "null argument");
    if (p->mask_mode != CS_MASK_NONE) return fail(ctx, CS_ERR_UNSUPPORTED, "cs_normxcorr2_host takes unmasked maps");
    if (p->compute_dtype != CS_F32) return fail(ctx, CS_ERR_UNSUPPORTED, "cs_normxcorr2_host computes in float32");
    if (out_dtype != CS_F32 && out_dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad output dtype");
    if (sig_dtype != CS_F32 && sig_dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad signal dtype");
    const size_t esz = sig_dtype == CS_F64 ? 8 : 4;          // float64 maps are narrowed on the device, slab by slab
    const int ms = p->ms, ns = p->ns;
    if (ms <= 0 || ns <= 0 || ld_in < ns || ld_out < ns) return fail(ctx, CS_ERR_INVALID, "bad geometry");
    const int km = kernel->km;
    const int kh = (km - 1) / 2, kt = km - 1 - kh;          // rows a window reaches above / below its pixel
    const int64_t ld = ((int64_t)ns + 15) / 16 * 16;
    const size_t map_bytes = (size_t)ms * (size_t)ld * 8;     // sized for either input type
    if (map_bytes > ctx->d_host_bytes) {
        CS_HIP(ctx, hipDeviceSynchronize());
        if (ctx->d_host_in) CS_HIP(ctx, hipFree(ctx->d_host_in));
        if (ctx->d_host_out) CS_HIP(ctx, hipFree(ctx->d_host_out));
        ctx->d_host_in = ctx->d_host_out = nullptr;
        ctx->d_host_bytes = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_host_in, map_bytes));
        CS_HIP(ctx, hipMalloc(&ctx->d_host_out, map_bytes));
        ctx->d_host_bytes = map_bytes;
    }
    const size_t bounce_bytes = (size_t)ms * (size_t)ns * 4;
    if (bounce_bytes > ctx->h_bounce_bytes) {
        if (ctx->h_bounce) CS_HIP(ctx, hipHostFree(ctx->h_bounce));
        ctx->h_bounce = nullptr;
        ctx->h_bounce_bytes = 0;
        CS_HIP(ctx, hipHostMalloc(&ctx->h_bounce, bounce_bytes, hipHostMallocDefault));
        ctx->h_bounce_bytes = bounce_bytes;
    }
    if (!ctx->s_up) {
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_run, hipStreamNonBlocking));
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
    }
    // slabs of ~1/12 of the map, whole 64-row tiles
    int rows = std::max(64, ((ms + 11) / 12 + 63) / 64 * 64);
    const int n_slabs = (ms + rows - 1) / rows;
    while ((int)ctx->ev_up.size() < n_slabs) {
        hipEvent_t a, b, c;
        CS_HIP(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
        CS_HIP(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
        CS_HIP(ctx, hipEventCreateWithFlags(&c, hipEventDisableTiming));
        ctx->ev_up.push_back(a);
        ctx->ev_run.push_back(b);
        ctx->ev_down.push_back(c);
    }
    cs_matrix m_in{ctx->d_host_in, sig_dtype, CS_LAYOUT_DENSE, ld, 0, 0, 0};
    cs_matrix m_out{ctx->d_host_out, CS_F32, CS_LAYOUT_DENSE, ld, 0, 0, 0};
    // warm the template upload (it synchronises) before the pipeline starts
    {
        cs_normxcorr2_params p0 = *p;
        p0.row_begin = 0;
        p0.row_end = 0;
        cs::CorrArgs<float> A;
        int rc0 = build_args<float>(ctx, ctx->s_run, &m_in, kernel, &p0, &A);
        if (rc0) return rc0;
    }
    // ---- host side of the drain: workers convert / copy each slab out of the bounce buffer as soon as
    //      its download has completed
    const int n_workers = (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> workers;
    std::vector<int> worker_rc(n_workers, 0);
    // slabs whose download has been ENQUEUED (an event that was never recorded, or still carries the
    // previous call's record, would let hipEventSynchronize return at once); -1 = give up
    std::atomic<int> enqueued{0};
    const float* bounce = reinterpret_cast<const float*>(ctx->h_bounce);
    const int device = ctx->device;
    for (int w = 0; w < n_workers; ++w) {
        workers.emplace_back([=, &worker_rc, &enqueued]() {
            (void)hipSetDevice(device);
            for (int k = 0; k < n_slabs; ++k) {
                int seen;
                while ((seen = enqueued.load(std::memory_order_acquire)) <= k && seen >= 0) std::this_thread::yield();
                if (seen < 0) return;
                if (hipEventSynchronize(ctx->ev_down[k]) != hipSuccess) {
                    worker_rc[w] = 1;
                    return;
                }
                const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
                const int span = r1 - r0, lo = r0 + (int)((long long)span * w / n_workers),
                          hi = r0 + (int)((long long)span * (w + 1) / n_workers);
                for (int r = lo; r < hi; ++r) {
                    const float* src = bounce + (size_t)r * ns;
                    if (out_dtype == CS_F64) {
                        double* dst = reinterpret_cast<double*>(h_out) + (size_t)r * ld_out;
                        for (int c = 0; c < ns; ++c) dst[c] = (double)src[c];
                    } else {
                        std::memcpy(reinterpret_cast<float*>(h_out) + (size_t)r * ld_out, src, (size_t)ns * 4);
                    }
                }
            }
        });
    }
    // ---- enqueue: upload slab k + 1, kernel of slab k (its windows reach into slab k + 1), download slab k
    int rc = CS_OK;
    auto upload = [&](int k) -> hipError_t {
        const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
        hipError_t e;
        const char* src = reinterpret_cast<const char*>(h_signal) + (size_t)r0 * ld_in * esz;
        if (ld_in == ns && ld == ns)
            e = hipMemcpyAsync((char*)ctx->d_host_in + (size_t)r0 * ld * esz, src, (size_t)(r1 - r0) * ns * esz,
                               hipMemcpyHostToDevice, ctx->s_up);
        else
            e = hipMemcpy2DAsync((char*)ctx->d_host_in + (size_t)r0 * ld * esz, (size_t)ld * esz, src, (size_t)ld_in * esz,
                                 (size_t)ns * esz, (size_t)(r1 - r0), hipMemcpyHostToDevice, ctx->s_up);
        if (e != hipSuccess) return e;
        return hipEventRecord(ctx->ev_up[k], ctx->s_up);
    };
    // the windows of slab k's last row reach kt rows down: with slabs of `rows` rows that is `ahead` slabs (1 unless the
    // template is taller than two slabs), all of which must have landed before the kernel of slab k starts (uploads are
    // issued in order on one stream, so waiting for the furthest one covers the others)
    (void)kh;
    if (!ctx->d_counts_peak) {
        CS_HIP(ctx, hipMalloc(&ctx->d_counts_peak, 256));
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_peak, 256, hipHostMallocDefault));
    }
    CS_HIP(ctx, hipMemsetAsync(ctx->d_counts_peak, 0, 4, ctx->s_run));
    const int ahead = std::max(1, (kt + rows - 1) / rows);
    int uploaded = -1;
    hipError_t he = hipSuccess;
    for (int k = 0; k < n_slabs && he == hipSuccess && rc == CS_OK; ++k) {
        const int need = std::min(k + ahead, n_slabs - 1);
        while (uploaded < need && he == hipSuccess) he = upload(++uploaded);
        if (he != hipSuccess) break;
        const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
        he = hipStreamWaitEvent(ctx->s_run, ctx->ev_up[need], 0);
        if (he != hipSuccess) break;
        cs_normxcorr2_params pk = *p;
        pk.row_begin = r0;
        pk.row_end = r1;
        rc = cs_normxcorr2(ctx, ctx->s_run, &m_in, kernel, &pk, &m_out, nullptr);
        if (rc != CS_OK) break;
        // the slab's largest |pixel| on the side (16 us for the whole 4096^2 map): see CS_ERR_RANGE
        if (cs::launch_peak_rows((const char*)ctx->d_host_in + (size_t)r0 * ld * esz, sig_dtype == CS_F64, ld, r1 - r0, ns, ctx->n_cu,
                                 reinterpret_cast<unsigned*>(ctx->d_counts_peak), ctx->s_run) != 0) {
            he = hipErrorLaunchFailure;
            break;
        }
        he = hipEventRecord(ctx->ev_run[k], ctx->s_run);
        if (he != hipSuccess) break;
        he = hipStreamWaitEvent(ctx->s_down, ctx->ev_run[k], 0);
        if (he != hipSuccess) break;
        if (ld == ns)
            he = hipMemcpyAsync((char*)ctx->h_bounce + (size_t)r0 * ns * 4, (char*)ctx->d_host_out + (size_t)r0 * ld * 4,
                                (size_t)(r1 - r0) * ns * 4, hipMemcpyDeviceToHost, ctx->s_down);
        else
            he = hipMemcpy2DAsync((char*)ctx->h_bounce + (size_t)r0 * ns * 4, (size_t)ns * 4,
                                  (char*)ctx->d_host_out + (size_t)r0 * ld * 4, (size_t)ld * 4, (size_t)ns * 4,
                                  (size_t)(r1 - r0), hipMemcpyDeviceToHost, ctx->s_down);
        if (he != hipSuccess) break;
        he = hipEventRecord(ctx->ev_down[k], ctx->s_down);
        if (he == hipSuccess) enqueued.store(k + 1, std::memory_order_release);
    }
    unsigned peak_bits = 0u;
    if (he == hipSuccess && rc == CS_OK) {
        he = hipMemcpyAsync(ctx->h_peak, ctx->d_counts_peak, 4, hipMemcpyDeviceToHost, ctx->s_run);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->s_run);
        if (he == hipSuccess) peak_bits = *ctx->h_peak;
    }
    if (he != hipSuccess || rc != CS_OK) enqueued.store(-1, std::memory_order_release);     // release the workers
    for (auto& t : workers) t.join();
    if (rc != CS_OK) return rc;
    if (he != hipSuccess) return fail(ctx, CS_ERR_HIP, "pipelined call failed: %s", hipGetErrorString(he));
    for (int w = 0; w < n_workers; ++w)
        if (worker_rc[w]) return fail(ctx, CS_ERR_HIP, "download wait failed");
    {
        const float limit = 1e15f;
        unsigned limit_bits;
        std::memcpy(&limit_bits, &limit, 4);
        if (p->compute_dtype == CS_F32 && peak_bits > limit_bits)
            return fail(ctx, CS_ERR_RANGE, "the map holds non-finite pixels or magnitudes beyond 1e15: float64 path");
    }
    return CS_OK;
}

int cs_xcorr2(cs_ctx* ctx, void* stream_, const cs_matrix* signal, int32_t ms, int32_t ns,
              const double* h_weights, int32_t km, int32_t kn, double threshold, int32_t compute_dtype,
              const cs_matrix* out)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!h_weights || km <= 0 || kn <= 0) return fail(ctx, CS_ERR_INVALID, "bad weights");
    if (ms < km || ns < kn) return fail(ctx, CS_ERR_INVALID, "signal smaller than kernel");
    int rc = check_matrix(ctx, signal, "signal", ns);
    if (rc) return rc;
    rc = check_matrix(ctx, out, "out", ns);
    if (rc) return rc;
    const int kk = km * kn;
    std::vector<double> w(3 * (size_t)kk, 0.0);
    for (int t = 0; t < kk; ++t) w[t] = h_weights[t];
    // exactly vertically symmetric weights: the folded chain of the streaming kernel applies
    bool sym = !std::getenv("CHROMOSIGHT_HIP_NO_SYMMETRY");
    for (int r = 0; r < km / 2 && sym; ++r)
        for (int c = 0; c < kn; ++c)
            if (w[r * kn + c] != w[(km - 1 - r) * kn + c]) {
                sym = false;
                break;
            }
#define CS_XC(TC)                                                         \
    {                                                                     \
        rc = upload_weights<TC>(ctx, stream, w);                          \
        if (rc) return rc;                                                \
        cs::CorrArgs<TC> A;                                               \
        std::memset(&A, 0, sizeof(A));                                    \
        A.sig = view_of(signal);                                          \
        A.sig_is_f64 = signal->dtype == CS_F64;                           \
        A.out = view_of(out);                                             \
        A.out_is_f64 = out->dtype == CS_F64;                              \
        A.ms = ms; A.ns = ns; A.km = km; A.kn = kn;                       \
        A.row_begin = 0; A.row_end = ms;                                  \
        A.max_dist = -1;                                                  \
        A.w = reinterpret_cast<const TC*>(ctx->d_w[sizeof(TC) == 8 ? 1 : 0]); \
        A.ks.n = (TC)kk; A.ks.thr = (TC)threshold;                        \
        A.xcorr_only = 1;                                                 \
        A.w_sym = sym ? 1 : 0;                                            \
        return launch_corr<TC>(ctx, A, stream, getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr); \
    }
    if (compute_dtype == CS_F64) CS_XC(double)
    if (compute_dtype == CS_F32) CS_XC(float)
#undef CS_XC
    return fail(ctx, CS_ERR_INVALID, "bad compute dtype");
}

int cs_rescore_f64(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* p, const int32_t* d_rows, const int32_t* d_cols,
                   int64_t n_px, double* d_out_corr, double* d_out_nobs)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_px < 0 || (n_px > 0 && (!d_rows || !d_cols || !d_out_corr)))
        return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    cs::CorrArgs<double> A;
    int rc = build_args<double>(ctx, stream, signal, kernel, p, &A);
    if (rc) return rc;
    rc = cs::launch_rescore_f64(A, d_rows, d_cols, n_px, d_out_corr, d_out_nobs, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "rescore launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_compact_ge(cs_ctx* ctx, void* stream_, const cs_matrix* corr, int32_t ms, int32_t ns,
                  double threshold, int32_t lo_diag, int32_t hi_diag, int32_t* d_rows, int32_t* d_cols,
                  double* d_vals, int64_t cap, int64_t* d_count)
{
    CS_ENTER(ctx);
    int rc = check_matrix(ctx, corr, "corr", ns);
    if (rc) return rc;
    if (!d_rows || !d_cols || !d_vals || !d_count || cap < 0) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    rc = cs::launch_compact_ge(view_of(corr), corr->dtype == CS_F64, ms, ns, threshold, lo_diag, hi_diag, d_rows,
                               d_cols, d_vals, cap, (long long*)d_count, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "compact launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

static int csr_view(cs_ctx* ctx, const cs_csr* m, cs::CsrView* v)
{
    if (!m) return fail(ctx, CS_ERR_INVALID, "null csr");
    if (m->n_rows < 0 || m->n_cols < 0 || m->nnz < 0) return fail(ctx, CS_ERR_INVALID, "bad csr shape");
    if (m->dtype != CS_F32 && m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad csr dtype");
    if (!m->d_indptr || (m->nnz > 0 && (!m->d_indices || !m->d_data))) return fail(ctx, CS_ERR_INVALID, "null csr arrays");
    v->n_rows = m->n_rows;
    v->n_cols = m->n_cols;
    v->nnz = m->nnz;
    v->indptr = (const long long*)m->d_indptr;
    v->row_end = m->d_row_end ? (const long long*)m->d_row_end : (const long long*)m->d_indptr + 1;
    v->col0 = m->col0;
    v->row_w = m->d_row_weight;
    v->col_w = m->d_row_weight ? m->d_col_weight : nullptr;
    if (m->d_row_weight && !m->d_col_weight) return fail(ctx, CS_ERR_INVALID, "row weights without column weights");
    v->indices = m->d_indices;
    v->data = m->d_data;
    v->is_f64 = m->dtype == CS_F64;
    return CS_OK;
}

int cs_distance_law_csr(cs_ctx* ctx, void* stream_, const cs_csr* mat, const uint8_t* d_detectable,
                        int32_t n_diags, double* d_sum, int64_t* d_cnt)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (n_diags < 0 || (n_diags > 0 && (!d_sum || !d_cnt))) return fail(ctx, CS_ERR_INVALID, "bad law buffers");
    rc = cs::launch_distance_law(v, d_detectable, n_diags, d_sum, (long long*)d_cnt, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "distance law launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_detrend_csr(cs_ctx* ctx, void* stream_, const cs_csr* mat, const double* d_law, int32_t n_law,
                   double max_val, void* d_out)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (!d_law || n_law < 0 || (v.nnz > 0 && !d_out)) return fail(ctx, CS_ERR_INVALID, "bad detrend buffers");
    rc = cs::launch_detrend_csr(v, d_law, n_law, max_val, d_out, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "detrend launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_csr_to_band(cs_ctx* ctx, void* stream_, const cs_csr* mat, const double* d_law, int32_t n_law,
                   double max_val, const cs_matrix* band)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (!band || !band->d_ptr) return fail(ctx, CS_ERR_INVALID, "null output matrix");
    if (band->dtype != CS_F32 && band->dtype != CS_F64 && band->dtype != CS_U8)
        return fail(ctx, CS_ERR_INVALID, "bad output dtype");
    if (band->layout == CS_LAYOUT_DENSE ? band->ld < v.n_cols : (band->band_w <= 0 || band->ld < band->band_w))
        return fail(ctx, CS_ERR_INVALID, "bad output geometry");
    rc = cs::launch_csr_to_band(v, d_law, n_law, max_val, view_of(band), band->dtype, ctx->n_cu,
                                (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "csr_to_band launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}


int cs_csr_band_extent(cs_ctx* ctx, void* stream_, const cs_csr* mat, int32_t lo_diag, int32_t hi_diag,
                       int64_t* d_begin, int64_t* d_end)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (v.n_rows > 0 && (!d_begin || !d_end)) return fail(ctx, CS_ERR_INVALID, "null extent buffers");
    rc = cs::launch_csr_band_extent(v, lo_diag, hi_diag, (long long*)d_begin, (long long*)d_end, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "band extent launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_distance_law_finish(cs_ctx* ctx, void* stream_, const double* d_sum, const int64_t* d_cnt, int32_t n_diags,
                           double* d_law)
{
    CS_ENTER(ctx);
    if (n_diags < 0 || (n_diags > 0 && (!d_sum || !d_cnt || !d_law))) return fail(ctx, CS_ERR_INVALID, "bad law buffers");
    int rc = cs::launch_law_finish(d_sum, (const long long*)d_cnt, n_diags, d_law, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "law finish launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_remove_neighbours(const int64_t* h_bin1, const int64_t* h_bin2, const int64_t* h_order, int64_t n, int64_t win,
                         uint8_t* h_keep)
{
    if (n < 0 || win < 1 || (n > 0 && (!h_bin1 || !h_bin2 || !h_order || !h_keep))) return CS_ERR_INVALID;
    // kept patterns bucketed by (bin1 / win, bin2 / win): a neighbour closer than win on both axes lies in one of the 3 x 3
    // surrounding cells -- and a cell holds at most ONE kept pattern (two patterns of one cell are closer than win on both axes),
    // so the grid is a flat open-addressing table of (cell, pattern): nine probes of a few nanoseconds per pattern (the
    // node-based map of vectors this replaces took 140 ns per pattern: 2.3 of the 4.8 ms of a borders table of the C4 genome)
    size_t cap = 16;
    while (cap < 2 * (size_t)n + 2) cap <<= 1;
    std::vector<uint64_t> keys(cap, ~0ull);
    std::vector<int64_t> vals(cap);
    auto cell = [](int64_t a, int64_t b) { return ((uint64_t)(a + (1ll << 30)) << 32) | (uint64_t)(uint32_t)(b + (1ll << 30)); };
    auto slot_of = [&](uint64_t key) {
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (cap - 1);
        while (keys[h] != ~0ull && keys[h] != key) h = (h + 1) & (cap - 1);
        return h;
    };
    for (int64_t t = 0; t < n; ++t) h_keep[t] = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t i = h_order[t];
        if (i < 0 || i >= n) return CS_ERR_INVALID;
        const int64_t b1 = h_bin1[i], b2 = h_bin2[i];
        const int64_t c1 = b1 >= 0 ? b1 / win : -((-b1 + win - 1) / win), c2 = b2 >= 0 ? b2 / win : -((-b2 + win - 1) / win);
        bool close = false;
        for (int64_t d1 = -1; d1 <= 1 && !close; ++d1)
            for (int64_t d2 = -1; d2 <= 1 && !close; ++d2) {
                const size_t h = slot_of(cell(c1 + d1, c2 + d2));
                if (keys[h] == ~0ull) continue;
                const int64_t j = vals[h];
                const int64_t e1 = h_bin1[j] - b1, e2 = h_bin2[j] - b2;
                close = (e1 < 0 ? -e1 : e1) < win && (e2 < 0 ? -e2 : e2) < win;
            }
        if (!close) {
            h_keep[i] = 1;
            const size_t h = slot_of(cell(c1, c2));          // (empty: a kept pattern of this cell would have been close)
            keys[h] = cell(c1, c2);
            vals[h] = i;
        }
    }
    return CS_OK;
}

// 2 * Phi(-a), a >= 0 or NaN: the two-sided tail of stats.py:43-81 with the case split of the normal distribution
// function the reference calls (scipy.special.ndtr)
static double two_sided_tail(double a)
{
    const double x = -a * M_SQRT1_2, z = std::fabs(x);
    double y;
    if (z < M_SQRT1_2) y = 0.5 + 0.5 * std::erf(x);
    else {
        y = 0.5 * std::erfc(z);
        if (x > 0) y = 1.0 - y;
    }
    return 2.0 * y;
}

int cs_accept_records(const cs_focus* h_rec, int64_t n_blocks, const int64_t* h_counts, const int32_t* h_rows,
                      const int32_t* h_cols, const int32_t* h_max_dist, int32_t inter, int32_t km, int32_t kn,
                      double missing_tol, double zero_tol, int32_t full, int32_t flags, double* h_table, uint8_t* h_ok,
                      int64_t* h_kept)
{
    const bool compact = (flags & 1) != 0, have_p = (flags & 2) != 0;
    if (n_blocks < 0 || km < 1 || kn < 1 || (n_blocks > 0 && (!h_counts || !h_rows || !h_cols || !h_kept))) return CS_ERR_INVALID;
    const double tot = (double)km * (double)kn;
    int64_t n = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (h_counts[b] < 0) return CS_ERR_INVALID;
        n += h_counts[b];
    }
    if (n > 0 && (!h_rec || !h_table || !h_ok)) return CS_ERR_INVALID;
    if (compact && have_p) {
        // Records that carry their p-values, accepted rows packed: ONE pass on the calling thread -- a record is two quotients, three
        // compares and, when it passes, a 32-byte row written at the packed position.  The two-pass form below wakes pool workers
        // (for the transcendental functions of records WITHOUT p-values) and then moves every accepted row: 122 us for the 18 600
        // records of a genome's 1-D pattern at the END of a step's critical path, 94 us this way (5 ns a record on the GPU box's
        // host; tried and dropped: counting on the pool's threads first -- 96 us --, integer limits instead of the quotients --
        // their table costs a short list more than it saves a long one).
        int64_t at = 0, out = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            const int64_t ms = h_rows[b], ns = h_cols[b];
            const bool limited = !inter && h_max_dist && h_max_dist[b] >= 0;
            const int64_t md = limited ? (int64_t)h_max_dist[b] : 0;
            int64_t kept = 0;
            for (int64_t t = at; t < at + h_counts[b]; ++t) {
                const cs_focus& f = h_rec[t];
                const double undetected = (double)f.n_missing / tot;
                const double zero = (double)f.n_zero / (tot - (double)f.n_missing);   // 0 / 0 -> NaN -> rejected
                const bool ok = f.inside != 0 && undetected < missing_tol && zero < zero_tol;
                h_ok[t] = ok ? 1 : 0;
                if (!ok) continue;
                const int64_t r = f.bin1, c = f.bin2;
                bool in_band = r >= 0 && r < ms && c >= 0 && c < ns;
                if (!inter) in_band = in_band && c - r >= 0 && (!limited || c - r <= md);
                double* row = h_table + 4 * out;
                row[0] = (double)r;
                row[1] = (double)c;
                row[2] = in_band ? f.score : 0.0;
                row[3] = f.pval;
                ++out;
                ++kept;
            }
            h_kept[b] = kept;
            at += h_counts[b];
        }
        return CS_OK;
    }
    // pass 1, record by record (a few transcendental functions each: threads beyond a couple of thousand records):
    // the row of every record at its own slot
    auto rows_piece = [&](int64_t b, int64_t at, int64_t cnt) {
        {
            const int64_t ms = h_rows[b], ns = h_cols[b];
            const bool limited = !inter && h_max_dist && h_max_dist[b] >= 0;
            for (int64_t t = at; t < at + cnt; ++t) {
                const cs_focus& f = h_rec[t];
                const int64_t r = f.bin1, c = f.bin2;
                // coefficient on the trimmed map (detection.py:269-270) ...
                bool in_band = r >= 0 && r < ms && c >= 0 && c < ns;
                if (!inter) in_band = in_band && c - r >= 0 && (!limited || c - r <= (int64_t)h_max_dist[b]);
                // acceptance rules of validate_patterns (:121-141) on the window statistics
                const double undetected = (double)f.n_missing / tot;
                const double zero = (double)f.n_zero / (tot - (double)f.n_missing);   // 0 / 0 -> NaN -> rejected
                const bool ok = f.inside != 0 && undetected < missing_tol && zero < zero_tol;
                h_ok[t] = ok ? 1 : 0;
                if (!ok && compact) continue;
                // ... p-value on the untrimmed one (:332-336), Fisher z (stats.py:43-81)
                double n_obs = full ? f.n_obs : tot;
                if (n_obs == 0) n_obs = tot;
                double pval = 1.0;                                // 10 ** 0 where the coefficient is exactly 0
                if (have_p) {
                    pval = f.pval;                                // formed by the kernel that wrote the record (cs_foci.hip focus_pval)
                } else if (f.score != 0) {
                    const double zz = std::atanh(f.score) * std::sqrt(n_obs - 3.0);
                    // (the reference forms 10 ** log10(p): p again within two units in the last place -- the round trip through
                    // log10 and pow was 40 % of a record's arithmetic, and the p-values are pinned to 1e-12)
                    pval = two_sided_tail(std::fabs(zz));
                }
                double* row = h_table + 4 * t;
                row[0] = (double)r;
                row[1] = (double)c;
                row[2] = ok ? (in_band ? f.score : 0.0) : std::numeric_limits<double>::quiet_NaN();
                row[3] = pval;
            }
        }
    };
    // tasks: a block's records, long blocks in pieces of 512 (waking a sleeping worker costs the caller ~ 4 us: 16 threads at most --
    // with 64 the 56 000 records of a genome's 1-D pattern took 266 us, with 32 and pieces of 1024 180 us); the pool's workers and this thread take them from a counter
    // (records that carry their p-values cost a few nanoseconds each: pieces of 4096, so that a share's few thousand stay on the
    // calling thread)
    const int64_t piece_n = have_p ? 4096 : 512;
    struct Piece { int64_t b, at, n; };
    std::vector<Piece> pieces;
    {
        int64_t at = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            for (int64_t o = 0; o < h_counts[b]; o += piece_n) pieces.push_back({b, at + o, std::min<int64_t>(piece_n, h_counts[b] - o)});
            at += h_counts[b];
        }
    }
    static const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
    const int max_threads = (int)std::min<int64_t>(std::min(16, std::max(1, cores / 2)), n / piece_n);
    HostPool::get().run((int)pieces.size(), max_threads, [&](int t) {
        const Piece& pc = pieces[(size_t)t];
        rows_piece(pc.b, pc.at, pc.n);
    });
    // pass 2: counts, and in compact mode the accepted rows packed to the front (in place: a row never moves backwards)
    int64_t at = 0, out = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        int64_t kept = 0;
        for (int64_t t = at; t < at + h_counts[b]; ++t) {
            if (!h_ok[t]) continue;
            ++kept;
            if (compact) {
                if (out != t) std::memcpy(h_table + 4 * out, h_table + 4 * t, 4 * sizeof(double));
                ++out;
            }
        }
        h_kept[b] = kept;
        at += h_counts[b];
    }
    return CS_OK;
}


int cs_csr_median(cs_ctx* ctx, void* stream_, const cs_csr* mat, double* h_median)
{
    CS_ENTER(ctx);
    if (!h_median) return fail(ctx, CS_ERR_INVALID, "null output");
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    auto grow = [](void* user, size_t bytes) -> void* {
        cs_ctx* c = (cs_ctx*)user;
        return ensure_scratch(c, &c->d_pool, &c->d_pool_bytes, bytes) == CS_OK ? c->d_pool : nullptr;
    };
    rc = cs::csr_median(v, ctx->n_cu, (hipStream_t)stream_, grow, ctx, h_median);
    if (rc) return fail(ctx, CS_ERR_HIP, "median failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_csr_median_many(cs_ctx* ctx, void* stream_, const cs_csr* mats, int32_t n, double* h_medians)
{
    CS_ENTER(ctx);
    if (n < 0 || (n > 0 && (!mats || !h_medians))) return fail(ctx, CS_ERR_INVALID, "bad view list");
    std::vector<cs::CsrView> views((size_t)n);
    for (int i = 0; i < n; ++i) {
        int rc = csr_view(ctx, mats + i, &views[(size_t)i]);
        if (rc) return rc;
    }
    auto grow = [](void* user, size_t bytes) -> void* {
        cs_ctx* c = (cs_ctx*)user;
        return ensure_scratch(c, &c->d_pool, &c->d_pool_bytes, bytes) == CS_OK ? c->d_pool : nullptr;
    };
    int rc = cs::csr_median_many(views.data(), n, ctx->n_cu, (hipStream_t)stream_, grow, ctx, h_medians);
    if (rc) return fail(ctx, CS_ERR_HIP, "medians failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_stage_blocks(cs_ctx* ctx, void* stream_, const cs_csr* genome, const cs_stage_block* blocks, int32_t n_blocks, double max_val)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, genome, &v);
    if (rc) return rc;
    if (n_blocks < 0 || (n_blocks > 0 && !blocks)) return fail(ctx, CS_ERR_INVALID, "bad block table");
    if (n_blocks == 0) return CS_OK;
    if (genome->d_row_end || genome->col0 != 0 || !v.row_w || v.row_w != v.col_w || v.n_rows != v.n_cols)
        return fail(ctx, CS_ERR_INVALID, "cs_stage_blocks takes the whole-genome pixel table (square, one weight vector, plain row pointers)");
    // rows per group: enough groups to keep every CU's waves on different rows (a wave walks its rows one after the other,
    // each a chain of dependent loads), 64 to 128 rows (measured: 50 000-bin block 0.353 -> 0.335 ms per C3 step with 64,
    // the 200 000-bin genome 1.59 -> 1.47 ms with 128; a group costs an LDS clear, a flush and a slot for the finish pass)
    long long total_rows = 0;
    for (int b = 0; b < n_blocks; ++b) total_rows += std::max(blocks[b].n, 0);
    const int kRowsPerGroup = (int)std::min<long long>(128, std::max<long long>(64, (total_rows / (4LL * ctx->n_cu) + 7) / 8 * 8));
    std::vector<cs::StageBlock> tab((size_t)n_blocks);
    int n_groups = 0, pitch = 1;
    for (int b = 0; b < n_blocks; ++b) {
        const cs_stage_block& s = blocks[b];
        cs::StageBlock& B = tab[b];
        if (s.n <= 0 || s.row0 < 0 || s.row0 + s.n > v.n_rows || s.keep < 0) return fail(ctx, CS_ERR_INVALID, "block %d outside the genome", b);
        if (s.layout != CS_LAYOUT_BAND && s.layout != CS_LAYOUT_DENSE) return fail(ctx, CS_ERR_INVALID, "block %d: bad layout", b);
        B.row0 = s.row0;
        B.n = s.n;
        B.keep = s.keep;
        B.n_diags = (int)std::min<int64_t>(s.n, (int64_t)s.keep + 1);
        B.dense = s.layout == CS_LAYOUT_DENSE;
        B.width = B.dense ? s.n : B.n_diags;
        if (!B.dense && s.band_w != B.n_diags) return fail(ctx, CS_ERR_INVALID, "block %d: band_w must be min(keep, n - 1) + 1", b);
        if (s.ld < B.width) return fail(ctx, CS_ERR_INVALID, "block %d: ld < stored slots", b);
        if (!s.d_law || (!s.d_band64 && !s.d_band32)) return fail(ctx, CS_ERR_INVALID, "block %d: null outputs", b);
        B.ld = s.ld;
        B.band64 = (double*)s.d_band64;
        B.band32 = (float*)s.d_band32;
        B.law = s.d_law;
        B.ld64 = s.ld;
        B.w64 = 0;
        B.counts = 0;
        B.lazy = nullptr;
        if (s.band32_counts) {
            // CS_LAYOUT_BAND_COUNTS: d_band32 receives the raw counts (the caller vouches that they are exact in float32)
            if (B.dense || !s.d_band32 || (s.ld & 3) || s.ld < (int64_t)B.width + 4 || ((uintptr_t)s.d_band32 & 15) || s.d_band64 || s.f64_diags > 0)
                return fail(ctx, CS_ERR_INVALID, "block %d: a band of counts needs the band layout, a 16-byte aligned d_band32 behind its header, ld a multiple of 4 and >= band_w + 4, and no float64 band", b);
            B.counts = 1;
            B.lazy = reinterpret_cast<cs::LazyBand*>(s.d_lazy);          // (or null: float32 consumers only)
        } else if (s.f64_diags > 0 || s.d_lazy) {
            if (B.dense || !s.d_lazy || !s.d_band64 || s.f64_diags <= 0 || s.ld64 < s.f64_diags || (s.ld64 & 1))
                return fail(ctx, CS_ERR_INVALID, "block %d: a lazy float64 band needs the band layout, d_lazy, d_band64 and an even ld64 >= f64_diags > 0", b);
            B.ld64 = s.ld64;
            B.w64 = std::min<int>(s.f64_diags, B.width);
            B.lazy = reinterpret_cast<cs::LazyBand*>(s.d_lazy);
        }
        B.group0 = B.n_groups = 0;
        n_groups += (s.n + kRowsPerGroup - 1) / kRowsPerGroup;
        pitch = std::max(pitch, B.n_diags);
    }
    if (pitch > 4096) return fail(ctx, CS_ERR_UNSUPPORTED, "distance laws of more than 4096 diagonals: stage block by block");
    pitch = (pitch + 63) / 64 * 64;
    const size_t stage_had = ctx->d_stage_bytes;
    rc = ensure_scratch(ctx, &ctx->d_stage, &ctx->d_stage_bytes, cs::stage_scratch_bytes(n_blocks, n_groups, pitch, v.n_rows));
    if (rc) return rc;
    if (ctx->d_stage_bytes != stage_had) ctx->stage_uploaded.clear();       // a fresh allocation holds no tables
    // the block / group tables go through one of two page-locked slots; a slot is reused two calls later, after the
    // event recorded behind its copy has fired (normally long ago): no synchronisation on the way in
    const int slot = ctx->stage_slot ^= 1;
    const size_t tbytes = cs::stage_table_bytes(n_blocks, n_groups);
    if (!ctx->ev_stage[slot]) CS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stage[slot], hipEventDisableTiming));
    else CS_HIP(ctx, hipEventSynchronize(ctx->ev_stage[slot]));
    if (tbytes > ctx->h_stage_bytes[slot]) {
        if (ctx->h_stage[slot]) CS_HIP(ctx, hipHostFree(ctx->h_stage[slot]));
        ctx->h_stage[slot] = nullptr;
        ctx->h_stage_bytes[slot] = 0;
        CS_HIP(ctx, hipHostMalloc(&ctx->h_stage[slot], 2 * tbytes, hipHostMallocDefault));
        ctx->h_stage_bytes[slot] = 2 * tbytes;
    }
    rc = cs::enqueue_stage_blocks(v.indptr, v.indices, v.data, v.is_f64, v.row_w, v.n_rows, tab.data(), n_blocks, max_val,
                                  kRowsPerGroup, ctx->n_cu, ctx->d_stage, ctx->h_stage[slot], (hipStream_t)stream_, &ctx->stage_uploaded);
    if (rc) return fail(ctx, CS_ERR_HIP, "staging kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipEventRecord(ctx->ev_stage[slot], (hipStream_t)stream_));
    return CS_OK;
}

int cs_host_alloc(cs_ctx* ctx, size_t bytes, void** h_ptr)
{
    if (!h_ptr) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return CS_OK;
}

int cs_host_free(cs_ctx* ctx, void* h_ptr)
{
    CS_ENTER(ctx);
    if (h_ptr) CS_HIP(ctx, hipHostFree(h_ptr));
    return CS_OK;
}

}  // extern "C"

// ---- cs_run_calls: a list of the library's own calls, natively (see the header) ---------------------------------------
namespace {
struct CallRun {
    cs_call* calls = nullptr;
    int n = 0;
    std::atomic<int>* done = nullptr;      // per call: 1 once it has returned (or was skipped)
    bool timing = false;
    std::chrono::steady_clock::time_point t0;
};

int dispatch_call(cs_call& c)
{
    void** p = c.p;
    const int64_t* i = c.i;
    switch (c.fn) {
        case CS_CALL_STAGE_BLOCKS:
            return cs_stage_blocks((cs_ctx*)p[0], p[1], (const cs_csr*)p[2], (const cs_stage_block*)p[3], (int32_t)i[0], c.d[0]);
        case CS_CALL_EVENT_RECORD:
            return cs_event_record((cs_ctx*)p[0], p[1], p[2]);
        case CS_CALL_STREAM_WAIT_EVENT:
            return cs_stream_wait_event((cs_ctx*)p[0], p[1], p[2]);
        case CS_CALL_DETECT_FOCI_BLOCKS:
            return cs_detect_foci_blocks((cs_ctx*)p[0], p[1], (int32_t)i[0], (const cs_matrix*)p[2], (const cs_matrix*)p[3],
                                         (const cs_kernel*)p[4], (const cs_normxcorr2_params*)p[5], (const cs_foci_params*)p[6],
                                         (cs_focus*)p[7], i[1], (int64_t*)p[8], (double*)p[9]);
        case CS_CALL_DETECT_FOCI_BATCH_TEMPLATES:
            return cs_detect_foci_batch_templates((cs_ctx*)p[0], p[1], (int32_t)i[0], (const cs_matrix*)p[2], (int32_t)i[1],
                                                  (const cs_kernel*)p[3], (const cs_normxcorr2_params*)p[4],
                                                  (const cs_foci_params*)p[5], (cs_focus*)p[6], i[2], (int64_t*)p[7], (double*)p[8]);
        case CS_CALL_ACCEPT_RECORDS:
            return cs_accept_records((const cs_focus*)p[0], i[0], (const int64_t*)p[1], (const int32_t*)p[2], (const int32_t*)p[3],
                                     (const int32_t*)p[4], (int32_t)i[1], (int32_t)i[2], (int32_t)i[3], c.d[0], c.d[1], (int32_t)i[4],
                                     (int32_t)i[5], (double*)p[5], (uint8_t*)p[6], (int64_t*)p[7]);
        case CS_CALL_DETECT_FOCI_BATCH_FINISH:
            return cs_detect_foci_batch_finish((cs_ctx*)p[0], p[1], (int64_t*)p[2]);
        case CS_CALL_STREAM_WAIT_TILES:
            return cs_stream_wait_tiles((cs_ctx*)p[0], p[1], (cs_ctx*)p[2], (int32_t)i[0], (int32_t)i[1]);
        default:
            return CS_ERR_INVALID;
    }
}

void run_lane(const CallRun& R, int lane)
{
    bool failed = false;
    for (int k = 0; k < R.n; ++k) {
        cs_call& c = R.calls[k];
        if (c.lane != lane) continue;
        if (!failed && c.after >= 0 && c.after < R.n) {
            int spins = 0;
            while (R.done[c.after].load(std::memory_order_acquire) == 0)
                if (++spins > 2000) std::this_thread::yield();
            if (R.calls[c.after].rc != 0) failed = true;            // what it waited for did not happen
        }
        const auto t_begin = std::chrono::steady_clock::now();
        if (failed && c.fn == CS_CALL_DETECT_FOCI_BATCH_FINISH) {
            // a lane that failed between the asynchronous batch and its finish must not leave the context "pending" (every later
            // foci call on it would be refused): end the batch whatever it holds; "nothing pending" is as good
            (void)dispatch_call(c);
        }
        c.rc = failed ? CS_ERR_INVALID : dispatch_call(c);
        if (R.timing)       // CHROMOSIGHT_HIP_TIMING: the host timeline of the list (lane, entry, begin and end since the list began)
            fprintf(stderr, "[timing] run_calls: lane %d call %2d fn %d  %7.1f -> %7.1f us\n", lane, k, c.fn,
                    std::chrono::duration<double, std::micro>(t_begin - R.t0).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - R.t0).count());
        if (c.rc != 0) failed = true;
        R.done[k].store(1, std::memory_order_release);
    }
}

// worker threads of the extra lanes: kept between calls (starting a thread costs more than a lane's host work), spinning
// briefly after a job -- the next step of a loop is usually microseconds away -- before they sleep
struct LaneWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0};             // 0 idle, 1 job posted, 2 job done
    const CallRun* job = nullptr;
    int lane = 0;
    bool quit = false;
    void loop()
    {
        for (;;) {
            int spins = 0;
            while (state.load(std::memory_order_acquire) != 1) {
                if (++spins < 20000) continue;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 1 || quit; });
                if (quit) return;
            }
            run_lane(*job, lane);
            state.store(2, std::memory_order_release);
        }
    }
};

std::mutex g_run_mu;                       // one cs_run_calls at a time (the workers are shared)
// (never destroyed: the detached workers may be waiting on their condition variables when the process exits)
std::vector<LaneWorker*>& lane_workers()
{
    static std::vector<LaneWorker*>* v = new std::vector<LaneWorker*>();
    return *v;
}
}  // namespace

extern "C" int cs_run_calls(cs_call* calls, int32_t n_calls)
{
    if (n_calls < 0 || (n_calls > 0 && !calls)) return CS_ERR_INVALID;
    if (n_calls == 0) return CS_OK;
    int lanes = 1;
    for (int k = 0; k < n_calls; ++k) {
        if (calls[k].lane < 0 || calls[k].lane > 7 || calls[k].after >= k) return CS_ERR_INVALID;     // (waits only look back)
        lanes = std::max(lanes, calls[k].lane + 1);
        calls[k].rc = 0;
    }
    std::lock_guard<std::mutex> lock(g_run_mu);
    std::vector<std::atomic<int>> done((size_t)n_calls);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    CallRun R;
    R.calls = calls;
    R.n = n_calls;
    R.done = done.data();
    R.timing = std::getenv("CHROMOSIGHT_HIP_TIMING") != nullptr;
    R.t0 = std::chrono::steady_clock::now();
    std::vector<LaneWorker*>& g_workers = lane_workers();
    while ((int)g_workers.size() < lanes - 1) {
        LaneWorker* w = new LaneWorker;
        g_workers.push_back(w);
        w->th = std::thread([w] { w->loop(); });
        w->th.detach();
    }
    for (int l = 1; l < lanes; ++l) {
        LaneWorker* w = g_workers[(size_t)l - 1];
        w->job = &R;
        w->lane = l;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->state.store(1, std::memory_order_release);
        }
        w->cv.notify_one();
    }
    run_lane(R, 0);
    for (int l = 1; l < lanes; ++l) {
        LaneWorker* w = g_workers[(size_t)l - 1];
        int spins = 0;
        while (w->state.load(std::memory_order_acquire) != 2)
            if (++spins > 2000) std::this_thread::yield();
        w->state.store(0, std::memory_order_release);
    }
    for (int k = 0; k < n_calls; ++k)
        if (calls[k].rc != 0) return calls[k].rc;
    return CS_OK;
}
