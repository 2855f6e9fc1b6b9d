// api.cpp — context, error text, profiling brackets and the host-buffer
// conveniences of the C ABI declared in include/tcmi.h.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "tcmi_internal.h"

static thread_local std::string g_err;

void *tcmi_ctx_pinned(tcmi_ctx *ctx, size_t bytes)
{
    if (ctx->h_pin_cap < bytes) {
        if (ctx->h_pin) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; ctx->h_pin_cap = 0; }
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void **)&ctx->h_pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->h_pin = nullptr; return nullptr; }
        ctx->h_pin_cap = want;
    }
    return ctx->h_pin;
}

int tcmi_fail(tcmi_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}

extern "C" {

int tcmi_abi_version(void) { return TCMI_ABI_VERSION; }

const char *tcmi_last_error(const tcmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int tcmi_device_count(int *out_count)
{
    if (!out_count) return tcmi_fail(nullptr, TCMI_E_ARG, "out_count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    (void)hipGetLastError();
    *out_count = n;
    return TCMI_OK;
}

static int ctx_create(int device, bool own, void *stream, tcmi_ctx **out);

int tcmi_ctx_create(int device, tcmi_ctx **out) { return ctx_create(device, true, nullptr, out); }

// stream may be NULL: the device's default (null) stream, e.g. torch's default current stream
int tcmi_ctx_create_on_stream(int device, void *stream, tcmi_ctx **out) { return ctx_create(device, false, stream, out); }

} // extern "C"

static int ctx_create(int device, bool own, void *stream, tcmi_ctx **out)
{
    if (!out) return tcmi_fail(nullptr, TCMI_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return tcmi_fail(nullptr, TCMI_E_NODEVICE,
                         "no HIP device available (%s); libtcmi has no CPU path",
                         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n)
        return tcmi_fail(nullptr, TCMI_E_ARG, "device %d out of range (%d devices)", device, n);
    if (hipSetDevice(device) != hipSuccess)
        return tcmi_fail(nullptr, TCMI_E_NODEVICE, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return tcmi_fail(nullptr, TCMI_E_NODEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return tcmi_fail(nullptr, TCMI_E_NODEVICE, "device %d is %s; libtcmi is built for gfx950 only",
                         device, prop.gcnArchName);
    tcmi_ctx *c = new tcmi_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // TCMI_STREAM_SPLIT (A/B): 1 = two streams by priority (inflate low, pack + tally + call high); 2 = by compute units
    // (TCMI_CU_HI of them, default 48, for the high stream, the rest for the inflate); 3 = inflate on its share of the compute units,
    // the rest at high priority on all of them
    const char *sv = std::getenv("TCMI_STREAM_SPLIT");
    const int split = sv ? std::atoi(sv) : 0;
    if (!own) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else if (split >= 1 && split <= 3) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const char *hv = std::getenv("TCMI_CU_HI");
        const int n_hi = std::max(1, std::min(c->n_cu - 1, hv ? std::atoi(hv) : 48));
        uint32_t m_lo[16] = {0}, m_hi[16] = {0};
        for (int i = 0; i < c->n_cu && i < 512; ++i) (i < c->n_cu - n_hi ? m_lo : m_hi)[i >> 5] |= 1u << (i & 31);
        const uint32_t words = (uint32_t)((c->n_cu + 31) / 32);
        hipError_t e1, e2;
        if (split == 1) {
            e1 = hipStreamCreateWithPriority(&c->stream_lo, hipStreamNonBlocking, least);
            e2 = hipStreamCreateWithPriority(&c->stream_hi, hipStreamNonBlocking, greatest);
        } else if (split == 2) {
            e1 = hipExtStreamCreateWithCUMask(&c->stream_lo, words, m_lo);
            e2 = hipExtStreamCreateWithCUMask(&c->stream_hi, words, m_hi);
        } else {
            e1 = hipExtStreamCreateWithCUMask(&c->stream_lo, words, m_lo);
            e2 = hipStreamCreateWithPriority(&c->stream_hi, hipStreamNonBlocking, greatest);
        }
        if (e1 != hipSuccess || e2 != hipSuccess || hipEventCreateWithFlags(&c->ev_split, hipEventDisableTiming) != hipSuccess) {
            const hipError_t e = e1 != hipSuccess ? e1 : e2;
            delete c;
            return tcmi_fail(nullptr, TCMI_E_HIP, "TCMI_STREAM_SPLIT=%d: the streams could not be made (%s)", split, hipGetErrorString(e));
        }
        c->stream = c->stream_lo;
    } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return tcmi_fail(nullptr, TCMI_E_HIP, "hipStreamCreate failed");
    }
    if (hipEventCreateWithFlags(&c->step_done, hipEventDisableTiming) != hipSuccess) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return tcmi_fail(nullptr, TCMI_E_HIP, "hipEventCreate failed");
    }
    const char *v = std::getenv("TCMI_TALLY_VARIANT");
    if (v) c->tally_variant = std::atoi(v);
    v = std::getenv("TCMI_ROUNDS_PER_WG");
    if (v) c->rounds_per_wg = std::atoi(v);
    {
        const unsigned hc = std::thread::hardware_concurrency();
        c->host_threads = hc ? (int)std::min(hc, 16u) : 4;
    }
    v = std::getenv("TCMI_HOST_THREADS");
    if (v && std::atoi(v) >= 1) c->host_threads = std::atoi(v);
    if (std::getenv("TCMI_NO_FUSED")) c->one_sync = 0;          // (A/B measurements)
    v = std::getenv("TCMI_CHUNK_STAGES");
    if (v && std::atoi(v) >= 1 && std::atoi(v) <= TCMI_F_MAXSTAGE) c->chunk_stages = std::atoi(v);

    *out = c;
    return TCMI_OK;
}

extern "C" {

static void free_ws(tcmi_ctx *c)
{
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_plain) (void)hipFree(c->d_plain);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    c->d_counts = nullptr;
    c->d_plain = c->d_alt = c->d_flags = nullptr;
    c->h_rec = nullptr;
    c->h_counts = nullptr;
    c->ws_L = c->ws_ld = 0;
}

int tcmi_ctx_destroy(tcmi_ctx *c)
{
    if (!c) return TCMI_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (tcmi_ctx *h : c->helpers) (void)tcmi_ctx_destroy(h);
    c->helpers.clear();
    for (auto &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    free_ws(c);
    tcmi_upload_scratch_free(c->upload_scratch);
    c->upload_scratch = nullptr;
    tcmi_dev_arena_free(c->dev_arena);
    c->dev_arena = nullptr;
    for (auto &b : c->blob_pool) (void)hipFree(b.p);
    c->blob_pool.clear();
    if (c->tok_dev) (void)hipFree(c->tok_dev);
    if (c->tok_host) (void)hipHostFree(c->tok_host);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->step_done) (void)hipEventDestroy(c->step_done);
    if (c->ev_after_sym) (void)hipEventDestroy(c->ev_after_sym);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipEvent_t e : c->ev_piece) (void)hipEventDestroy(e);
    if (c->stream_hi) {
        (void)hipStreamSynchronize(c->stream_hi);
        (void)hipStreamDestroy(c->stream_hi);
        (void)hipEventDestroy(c->ev_split);
        c->stream = c->stream_lo;
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TCMI_OK;
}

int tcmi_ctx_sync(tcmi_ctx *c)
{
    if (!c) return tcmi_fail(nullptr, TCMI_E_ARG, "ctx is NULL");
    TCMI_HIP(c, hipStreamSynchronize(c->stream));
    if (c->stream_hi) TCMI_HIP(c, hipStreamSynchronize(c->stream_hi));
    return TCMI_OK;
}

void *tcmi_ctx_stream(tcmi_ctx *c) { return c ? (void *)c->stream : nullptr; }

int tcmi_ctx_set_option(tcmi_ctx *c, const char *key, int value)
{
    if (!c || !key) return tcmi_fail(c, TCMI_E_ARG, "null argument");
    if (!std::strcmp(key, "tally_variant")) c->tally_variant = value;
    else if (!std::strcmp(key, "rounds_per_wg")) c->rounds_per_wg = value;
    else if (!std::strcmp(key, "host_threads")) c->host_threads = value < 1 ? 1 : value;
    else if (!std::strcmp(key, "chunk_stages")) c->chunk_stages = value < 0 ? 0 : value > TCMI_F_MAXSTAGE ? TCMI_F_MAXSTAGE : value;
    else if (!std::strcmp(key, "defer_call")) c->defer_call = value != 0;
    else if (!std::strcmp(key, "wg_per_cu")) c->wg_per_cu = value < 1 ? 1 : value;
    else if (!std::strcmp(key, "stage_cap")) c->stage_cap = value < 0 ? 0 : value;
    else if (!std::strcmp(key, "balance_chunks")) c->balance_chunks = value != 0;
    else if (!std::strcmp(key, "project_reads")) c->project_reads = value != 0;
    else if (!std::strcmp(key, "device_pack")) c->device_pack = value != 0;
    else if (!std::strcmp(key, "verify_crc")) c->verify_crc = value != 0;
    else if (!std::strcmp(key, "one_sync")) c->one_sync = value != 0;
    else if (!std::strcmp(key, "mid_wait")) c->mid_wait = value != 0;
    else if (!std::strcmp(key, "prefix_kernels")) c->prefix_kernels = value;
    else if (!std::strcmp(key, "decode_token_mb")) c->decode_token_mb = value > 0 ? value : 4096;
    else if (!std::strcmp(key, "profile_every")) c->prof_every = value < 1 ? 1 : value;
    else if (!std::strcmp(key, "sym_scratch_div")) c->sym_scratch_div = value < 1 ? 1 : value > 64 ? 64 : value;
    else if (!std::strcmp(key, "h2d_pieces")) c->h2d_pieces = value < -1 ? -1 : value > 16 ? 16 : value;
    else if (!std::strcmp(key, "split_sub")) c->split_sub = value < 0 ? 0 : value > 8 ? 8 : value;

    else return tcmi_fail(c, TCMI_E_ARG, "unknown option %s", key);
    return TCMI_OK;
}

int tcmi_ctx_stat(tcmi_ctx *c, const char *key, int64_t *value)
{
    if (!c || !key || !value) return tcmi_fail(c, TCMI_E_ARG, "null argument");
    if (!std::strcmp(key, "one_sync_taken")) { *value = c->stat_one_sync_taken; for (const tcmi_ctx *h : c->helpers) *value += h->stat_one_sync_taken; }
    else if (!std::strcmp(key, "one_sync_declined")) *value = c->stat_one_sync_declined;
    else if (!std::strcmp(key, "decode_batched")) { *value = c->stat_decode_batched; for (const tcmi_ctx *h : c->helpers) *value += h->stat_decode_batched; }
    else if (!std::strcmp(key, "split_sub_taken")) *value = c->stat_split_sub;
    else if (!std::strcmp(key, "h2d_piped")) { *value = c->stat_h2d_piped; for (const tcmi_ctx *h : c->helpers) *value += h->stat_h2d_piped; }
    else if (!std::strcmp(key, "one_sync_last_decline_flags")) *value = c->stat_last_decline;
    else return tcmi_fail(c, TCMI_E_ARG, "unknown statistic %s", key);
    return TCMI_OK;
}

// ---- profiling -------------------------------------------------------------------------
int tcmi_profile_enable(tcmi_ctx *c, int on)
{
    if (!c) return tcmi_fail(nullptr, TCMI_E_ARG, "ctx is NULL");
    c->prof = on != 0;
    return TCMI_OK;
}

static int drain(tcmi_ctx *c)
{
    for (auto &p : c->pending) {
        TCMI_HIP(c, hipEventSynchronize(p.b));
        float ms = 0.f;
        TCMI_HIP(c, hipEventElapsedTime(&ms, p.a, p.b));
        c->prof_ms[p.k] += ms;
        c->prof_n[p.k] += 1;
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
    return TCMI_OK;
}

int tcmi_profile_reset(tcmi_ctx *c)
{
    if (!c) return tcmi_fail(nullptr, TCMI_E_ARG, "ctx is NULL");
    int rc = drain(c);
    for (int k = 0; k < TCMI_K_NKERNELS; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return rc;
}

int tcmi_profile_get(tcmi_ctx *c, int k, double *total_ms, int64_t *launches)
{
    if (!c || k < 0 || k >= TCMI_K_NKERNELS) return tcmi_fail(c, TCMI_E_ARG, "bad kernel id");
    int rc = drain(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return TCMI_OK;
}

} // extern "C"

static hipEvent_t get_event(tcmi_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// recycle the events of launches that have finished, so a long profiled run does not keep
// creating events (hipEventCreate is far more expensive than hipEventRecord)
static void reap(tcmi_ctx *c)
{
    size_t done = 0;
    while (done < c->pending.size() && hipEventQuery(c->pending[done].b) == hipSuccess) {
        auto &p = c->pending[done];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->prof_ms[p.k] += ms; c->prof_n[p.k] += 1; }
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
        ++done;
    }
    (void)hipGetLastError();                                 // hipErrorNotReady from the query is expected
    if (done) c->pending.erase(c->pending.begin(), c->pending.begin() + (long)done);
}

void tcmi_prof_begin(tcmi_ctx *c, int k)
{
    c->prof_open = false;
    if (!c->prof || c->prof_mute) return;
    c->prof_open = true;
    if (c->pending.size() >= 24) reap(c);
    tcmi_ctx::Pending p{k, get_event(c), get_event(c)};
    (void)hipEventRecord(p.a, c->stream);
    c->pending.push_back(p);
}

void tcmi_prof_end(tcmi_ctx *c, int k)
{
    if (!c->prof || !c->prof_open || c->pending.empty()) return;
    c->prof_open = false;
    (void)k;
    (void)hipEventRecord(c->pending.back().b, c->stream);
}

extern "C" {

// ---- tally -----------------------------------------------------------------------------------
int tcmi_tally_dev(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, void *d_counts, int zero)
{
    if (!ctx || !rs || !d_counts) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0 || ld < L) return tcmi_fail(ctx, TCMI_E_ARG, "need 0 < L <= ld (L=%lld ld=%lld)", (long long)L, (long long)ld);
    if (L > INT32_MAX - 1024) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "L too large");
    if (rs->device != ctx->device) return tcmi_fail(ctx, TCMI_E_ARG, "read set lives on device %d, context on %d", rs->device, ctx->device);
    if (rs->max_end > L)
        return tcmi_fail(ctx, TCMI_E_ARG, "L=%lld is smaller than the read extent %lld (use tcmi_reads_extent)",
                         (long long)L, (long long)rs->max_end);
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    if (!rs->parts.empty()) {                                   // a read set of sub-ranges: every part adds its counts from its own context
        if (zero) {
            TCMI_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)ld * TCMI_NCOL * 4, ctx->stream));
            TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        for (const tcmi_readset::Part &p : rs->parts) {
            const int rc = tcmi_tally_dev(p.cx, p.rs, L, ld, d_counts, 0);
            if (rc) return tcmi_fail(ctx, rc, "%s", p.cx->err.c_str());
            TCMI_HIP(ctx, hipStreamSynchronize(p.cx->stream));
        }
        return TCMI_OK;
    }
    if (zero) {
        tcmi_prof_begin(ctx, TCMI_K_ZERO);
        TCMI_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)ld * TCMI_NCOL * 4, ctx->stream));
        tcmi_prof_end(ctx, TCMI_K_ZERO);
    }
    if (rs->n_piled == 0) return TCMI_OK;
    return tcmi_launch_tally(ctx, rs, L, ld, (int32_t *)d_counts);
}

static int ensure_ws(tcmi_ctx *ctx, int64_t L)
{
    if (ctx->ws_L >= L && ctx->d_counts) return TCMI_OK;
    free_ws(ctx);
    ctx->counts_clean = false;
    int64_t ld = tcmi_round_up(L, 256);
    TCMI_HIP(ctx, hipMalloc((void **)&ctx->d_counts, (size_t)ld * TCMI_NCOL * 4 + 256));
    TCMI_HIP(ctx, hipMalloc((void **)&ctx->d_plain, (size_t)ld * 3));
    ctx->d_alt = ctx->d_plain + ld;
    ctx->d_flags = ctx->d_plain + 2 * ld;
    TCMI_HIP(ctx, hipHostMalloc((void **)&ctx->h_rec, (size_t)ld * 3, hipHostMallocDefault));
    TCMI_HIP(ctx, hipHostMalloc((void **)&ctx->h_counts, (size_t)ld * TCMI_NCOL * 4 + 256, hipHostMallocDefault));
    ctx->ws_L = L;
    ctx->ws_ld = ld;
    return TCMI_OK;
}

int tcmi_counts_download(tcmi_ctx *ctx, const void *d_counts, int64_t L, int64_t ld, int32_t *counts)
{
    if (!ctx || !d_counts || !counts || L <= 0 || ld < L) return tcmi_fail(ctx, TCMI_E_ARG, "bad argument");
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> planes((size_t)ld * TCMI_NCOL);
    TCMI_HIP(ctx, hipMemcpyAsync(planes.data(), d_counts, planes.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < TCMI_NCOL; ++c) {
        const int32_t *src = planes.data() + (size_t)c * ld;
        for (int64_t p = 0; p < L; ++p) counts[p * TCMI_NCOL + c] = src[p];
    }
    return TCMI_OK;
}

int tcmi_counts_upload(tcmi_ctx *ctx, const int32_t *counts, int64_t L, int64_t ld, void *d_counts)
{
    if (!ctx || !d_counts || !counts || L <= 0 || ld < L) return tcmi_fail(ctx, TCMI_E_ARG, "bad argument");
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> planes((size_t)ld * TCMI_NCOL, 0);
    for (int c = 0; c < TCMI_NCOL; ++c)
        for (int64_t p = 0; p < L; ++p) planes[(size_t)c * ld + p] = counts[p * TCMI_NCOL + c];
    TCMI_HIP(ctx, hipMemcpyAsync(d_counts, planes.data(), planes.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TCMI_OK;
}

int tcmi_tally(tcmi_ctx *ctx, const tcmi_reads *reads, int64_t L, int32_t *counts)
{
    if (!ctx || !counts) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0) return tcmi_fail(ctx, TCMI_E_ARG, "L must be positive");
    tcmi_readset *rs = nullptr;
    int rc = tcmi_readset_upload(ctx, reads, &rs);
    if (rc) return rc;
    if (rs->max_end > L) {
        tcmi_readset_free(ctx, rs);
        return tcmi_fail(ctx, TCMI_E_ARG, "L=%lld is smaller than the read extent %lld (use tcmi_reads_extent)",
                         (long long)L, (long long)rs->max_end);
    }
    rc = ensure_ws(ctx, L);
    ctx->counts_clean = false;
    if (!rc) rc = tcmi_tally_dev(ctx, rs, L, ctx->ws_ld, ctx->d_counts, 1);
    if (!rc) rc = tcmi_counts_download(ctx, ctx->d_counts, L, ctx->ws_ld, counts);
    tcmi_readset_free(ctx, rs);
    return rc;
}

// ---- call ------------------------------------------------------------------------------------
int tcmi_call_dev(tcmi_ctx *ctx, const void *d_counts, int64_t L, int64_t ld, int32_t mincov, int include_ambig,
                  void *d_plain, void *d_alt, void *d_flags, void *d_events, void *d_event_counts)
{
    if (!ctx || !d_counts || !d_plain || !d_alt || !d_flags) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0 || ld < L) return tcmi_fail(ctx, TCMI_E_ARG, "need 0 < L <= ld");
    if ((d_events == nullptr) != (d_event_counts == nullptr))
        return tcmi_fail(ctx, TCMI_E_ARG, "d_events and d_event_counts must both be given or both NULL");
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    return tcmi_launch_call(ctx, (int32_t *)const_cast<void *>(d_counts), L, ld, mincov, include_ambig, 0, (uint8_t *)d_plain,
                            (uint8_t *)d_alt, (uint8_t *)d_flags, (int32_t *)d_events, (int32_t *)d_event_counts);
}

int tcmi_call(tcmi_ctx *ctx, const int32_t *counts, int64_t L, int32_t mincov, int include_ambig, uint8_t *plain,
              uint8_t *alt, uint8_t *flags, int32_t *event_idx, int64_t *n_events)
{
    if (!ctx || !counts || !plain || !alt || !flags) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0) return tcmi_fail(ctx, TCMI_E_ARG, "L must be positive");
    int rc = ensure_ws(ctx, L);
    if (rc) return rc;
    const int64_t ld = ctx->ws_ld;
    ctx->counts_clean = false;
    rc = tcmi_counts_upload(ctx, counts, L, ld, ctx->d_counts);
    if (rc) return rc;
    int32_t *d_ev = nullptr, *d_evc = nullptr;
    const int64_t nblk = (L + 255) / 256;
    if (event_idx) {
        TCMI_HIP(ctx, hipMalloc((void **)&d_ev, (size_t)nblk * 256 * 4));
        TCMI_HIP(ctx, hipMalloc((void **)&d_evc, (size_t)nblk * 4));
    }
    rc = tcmi_call_dev(ctx, ctx->d_counts, L, ld, mincov, include_ambig, ctx->d_plain, ctx->d_alt, ctx->d_flags, d_ev, d_evc);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(ctx->h_rec, ctx->d_plain, (size_t)ld * 3, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tcmi_fail(ctx, TCMI_E_HIP, "record download failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
        std::memcpy(plain, ctx->h_rec, (size_t)L);
        std::memcpy(alt, ctx->h_rec + ld, (size_t)L);
        std::memcpy(flags, ctx->h_rec + 2 * ld, (size_t)L);
    }
    if (!rc && event_idx) {
        std::vector<int32_t> ev((size_t)nblk * 256), evc((size_t)nblk);
        hipError_t e = hipMemcpy(ev.data(), d_ev, ev.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(evc.data(), d_evc, evc.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = tcmi_fail(ctx, TCMI_E_HIP, "event download failed: %s", hipGetErrorString(e));
        int64_t n = 0;
        for (int64_t b = 0; !rc && b < nblk; ++b)
            for (int32_t k = 0; k < evc[(size_t)b]; ++k) event_idx[n++] = ev[(size_t)b * 256 + k];
        if (n_events) *n_events = n;
    }
    if (d_ev) (void)hipFree(d_ev);
    if (d_evc) (void)hipFree(d_evc);
    return rc;
}

// The launches of one step on ctx->stream: [memset] tally, call.  When the counts are not wanted on the host the
// call kernel zeroes them behind itself and the next step into this workspace needs no memset.  The call records
// (3 bytes per position) go straight to the pinned host buffer: the kernel's own stores cross PCIe, which saves a
// separate copy and one launch boundary per step.
static int enqueue_step(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig, int want_counts,
                        bool memset_first)
{
    const int64_t ld = ctx->ws_ld;
    if (!rs->parts.empty()) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "a read set of sub-ranges (tcmi_split_step) takes no step of its own: tcmi_tally_dev + tcmi_call_dev");
    if (memset_first) TCMI_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, (size_t)ld * TCMI_NCOL * 4, ctx->stream));
    int rc = tcmi_tally_dev(ctx, rs, L, ld, ctx->d_counts, 0);
    if (rc) return rc;
    rc = tcmi_launch_call(ctx, ctx->d_counts, L, ld, mincov, include_ambig, want_counts ? 0 : 1, ctx->h_rec, ctx->h_rec + ld,
                          ctx->h_rec + 2 * ld, nullptr, nullptr);
    if (rc) return rc;
    if (want_counts)
        TCMI_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, (size_t)ld * TCMI_NCOL * 4, hipMemcpyDeviceToHost, ctx->stream));
    return TCMI_OK;
}

} // extern "C"  (the next three functions are internal C++ linkage: tcmi_internal.h)

// ---- ride-along call (used by the pipeline) ----------------------------------------------------------------------
// A step without its call kernel: the tally is launched now; the call of this matrix is carried by the tally launch
// of the NEXT step on the stream (`prev` of that call), or launched on its own by tcmi_step_flush.  One launch per
// step instead of two, and the call's PCIe stores overlap with the next tally's streaming.
static int launch_pending_call(tcmi_ctx *ctx)
{
    const int64_t ld = ctx->ws_ld;
    int rc = tcmi_launch_call(ctx, ctx->d_counts, ctx->pend_L, ld, ctx->pend_mincov, ctx->pend_amb, 1, ctx->h_rec, ctx->h_rec + ld,
                              ctx->h_rec + 2 * ld, nullptr, nullptr);
    if (rc) return rc;
    TCMI_HIP(ctx, hipEventRecord(ctx->step_done, ctx->stream));
    ctx->call_pending = false;
    ctx->counts_clean = true;
    return TCMI_OK;
}

int tcmi_step_flush(tcmi_ctx *ctx)
{
    if (!ctx) return tcmi_fail(ctx, TCMI_E_ARG, "ctx is NULL");
    if (!ctx->call_pending) return TCMI_OK;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    return launch_pending_call(ctx);
}

int tcmi_step_begin_deferred(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig, tcmi_ctx *prev)
{
    if (!ctx || !rs) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0 || rs->max_end > L) return tcmi_fail(ctx, TCMI_E_ARG, "L=%lld does not cover the reads (extent %lld)", (long long)L, (long long)rs->max_end);
    if (rs->device != ctx->device) return tcmi_fail(ctx, TCMI_E_ARG, "read set lives on device %d, context on %d", rs->device, ctx->device);
    if (ctx->step_L > 0 || ctx->call_pending) return tcmi_fail(ctx, TCMI_E_ARG, "tcmi_step_begin: the previous step of this context has not been ended");
    int rc = ensure_ws(ctx, L);
    if (rc) return rc;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    if (prev && (prev == ctx || !prev->call_pending)) prev = nullptr;
    if (prev && prev->stream != ctx->stream) {                 // no stream order between the two: the call goes out on its own
        rc = launch_pending_call(prev);
        if (rc) return rc;
        prev = nullptr;
    }
    if (!ctx->counts_clean) TCMI_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, (size_t)ctx->ws_ld * TCMI_NCOL * 4, ctx->stream));
    ctx->counts_clean = false;
    const bool sampled = ctx->prof && (ctx->step_tick++ % ctx->prof_every) == 0;
    tcmi_ride ride = {};
    if (prev) {
        ride.counts = prev->d_counts; ride.ld = prev->ws_ld; ride.L = prev->pend_L; ride.mincov = prev->pend_mincov;
        ride.amb = prev->pend_amb; ride.plain = prev->h_rec; ride.alt = prev->h_rec + prev->ws_ld; ride.flags = prev->h_rec + 2 * prev->ws_ld;
        ctx->ride = &ride;
    }
    ctx->prof_mute = !sampled;
    rc = tcmi_tally_dev(ctx, rs, L, ctx->ws_ld, ctx->d_counts, 0);
    ctx->prof_mute = false;
    ctx->ride = nullptr;
    if (rc) return rc;
    if (prev) {
        if (ride.taken) {
            TCMI_HIP(ctx, hipEventRecord(prev->step_done, ctx->stream));
            prev->call_pending = false;
            prev->counts_clean = true;
        } else {                                               // this step had no fast-kernel launch to carry it
            rc = launch_pending_call(prev);
            if (rc) return rc;
        }
    }
    ctx->call_pending = true;
    ctx->pend_L = L; ctx->pend_mincov = mincov; ctx->pend_amb = include_ambig;
    ctx->step_L = L;
    ctx->step_counts = false;
    return TCMI_OK;
}

extern "C" {

int tcmi_step_begin(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig, int want_counts)
{
    if (!ctx || !rs) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0 || rs->max_end > L) return tcmi_fail(ctx, TCMI_E_ARG, "L=%lld does not cover the reads (extent %lld)", (long long)L, (long long)rs->max_end);
    if (rs->device != ctx->device) return tcmi_fail(ctx, TCMI_E_ARG, "read set lives on device %d, context on %d", rs->device, ctx->device);
    if (ctx->step_L > 0) return tcmi_fail(ctx, TCMI_E_ARG, "tcmi_step_begin: the previous step of this context has not been ended");
    int rc = ensure_ws(ctx, L);
    if (rc) return rc;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    const bool memset_first = !ctx->counts_clean;
    ctx->counts_clean = false;
    // with profiling on, every prof_every-th step has its kernels bracketed by events
    const bool sampled = ctx->prof && (ctx->step_tick++ % ctx->prof_every) == 0;
    ctx->prof_mute = !sampled;
    rc = enqueue_step(ctx, rs, L, mincov, include_ambig, want_counts, memset_first);
    ctx->prof_mute = false;
    if (rc) return rc;
    ctx->counts_clean = !want_counts;
    TCMI_HIP(ctx, hipEventRecord(ctx->step_done, ctx->stream));
    ctx->step_L = L;
    ctx->step_counts = want_counts != 0;
    return TCMI_OK;
}

int tcmi_step_end(tcmi_ctx *ctx, const uint8_t **plain, const uint8_t **alt, const uint8_t **flags,
                  const int32_t **counts_planes, int64_t *ld_out)
{
    if (!ctx) return tcmi_fail(ctx, TCMI_E_ARG, "ctx is NULL");
    if (ctx->step_L <= 0) return tcmi_fail(ctx, TCMI_E_ARG, "tcmi_step_end without tcmi_step_begin");
    if (ctx->call_pending) {                                  // nobody carried this step's call: launch it now
        const int rc = tcmi_step_flush(ctx);
        if (rc) return rc;
    }
    TCMI_HIP(ctx, hipEventSynchronize(ctx->step_done));      // only this step: the stream may be shared
    const int64_t ld = ctx->ws_ld;
    if (plain) *plain = ctx->h_rec;
    if (alt) *alt = ctx->h_rec + ld;
    if (flags) *flags = ctx->h_rec + 2 * ld;
    if (counts_planes) *counts_planes = ctx->step_counts ? ctx->h_counts : nullptr;
    if (ld_out) *ld_out = ld;
    ctx->step_L = 0;
    return TCMI_OK;
}

// The joining rule of block ranges (include/tcmi.h, tcmi_split_step; distributed.check_range_anchors is the same rule in Python):
// ranges in rank order, each {first_block, n_blocks, first, next}; a range in the middle of the file starts at the first offset its
// first block finds plausible for a record, and only the range in front can vouch for it: its last record ends there.
static const char *ranges_join(const int64_t (*rg)[4], int world, int64_t inflated, int *who, int64_t *at, int64_t *want)
{
    bool have = false, any_before = false;
    int64_t expect = -1;
    for (int r = 0; r < world; ++r) {
        const int64_t fb = rg[r][0], nb = rg[r][1], first = rg[r][2], nxt = rg[r][3];
        if (nb <= 0) continue;
        if (fb != 0 && first >= 0) {
            *who = r; *at = first; *want = expect;
            if (have && first != expect) return "starts a record where the range in front does not end its last one";
            if (!have && any_before) return "starts a record, but no range in front says where its last record ends";
        }
        any_before = true;
        if (nxt >= 0) { expect = nxt; have = true; }
    }
    if (have && inflated >= 0 && expect != inflated) { *who = world - 1; *at = expect; *want = inflated; return "the last alignment record does not end with the file's stream"; }
    return nullptr;
}

// A rank's block range [first, first + cnt) as K sub-ranges decoded, packed and tallied SIDE BY SIDE, each on a context of its own (sub-range 0
// on the caller's, the others on helper contexts the caller's context owns: a stream and an arena each) with a host thread of its own
// — the inflate of one sub-range runs under the pack of another, as the many-file runner's contexts overlap files; the tallies add
// into the one matrix (the tally kernel's adds are atomic).  The sub-ranges join like ranks' ranges do: each starts where the one in
// front ends its last record (ranges_join).  -> a read set that is the sum of its parts; TCMI_E_UNSUPPORTED: sub-ranges that cannot
// vouch for each other (a sub-range without a record start, a chain that does not join) — the caller decodes the range in one piece.
static int split_sub_ranges(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first, int64_t cnt, int K, int64_t L, int64_t ld, void *d_counts,
                            size_t n_words, tcmi_readset **out)
{
    *out = nullptr;
    while ((int)ctx->helpers.size() < K - 1) {
        tcmi_ctx *h = nullptr;
        const int rc = tcmi_ctx_create(ctx->device, &h);
        if (rc) return tcmi_fail(ctx, rc, "a helper context for the sub-ranges could not be made");
        ctx->helpers.push_back(h);
    }
    for (tcmi_ctx *h : ctx->helpers) {                          // (the caller's decoder options)
        h->verify_crc = ctx->verify_crc; h->decode_token_mb = ctx->decode_token_mb; h->one_sync = ctx->one_sync; h->mid_wait = ctx->mid_wait;
        h->prefix_kernels = ctx->prefix_kernels; h->h2d_pieces = ctx->h2d_pieces; h->sym_scratch_div = ctx->sym_scratch_div; h->prof = false;
    }
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));            // (the matrix is zero before anybody adds to it)
    std::vector<tcmi_readset *> rs((size_t)K, nullptr);
    std::vector<int> rcs((size_t)K, TCMI_OK);
    std::vector<int64_t> fb((size_t)K + 1);
    for (int k = 0; k <= K; ++k) fb[(size_t)k] = first + cnt * k / K;
    // skewed starts: sub-range k's inflate waits (on the device) for the bgzf_symbols of sub-range k - 1; its thread waits (on the host)
    // until that launch and its event are queued.  A sub-range that fails before its launch lets the next one go all the same.
    static const bool skew = !(std::getenv("TCMI_SPLIT_SKEW") && std::atoi(std::getenv("TCMI_SPLIT_SKEW")) == 0);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> queued((size_t)K, 0);
    auto release = [&](int k) { { std::lock_guard<std::mutex> lk(mu); queued[(size_t)k] = 1; } cv.notify_all(); };
    auto cx_of = [&](int k) { return k == 0 ? ctx : ctx->helpers[(size_t)k - 1]; };
    if (skew)
        for (int k = 0; k < K; ++k)
            if (!cx_of(k)->ev_after_sym && hipEventCreateWithFlags(&cx_of(k)->ev_after_sym, hipEventDisableTiming) != hipSuccess)
                return tcmi_fail(ctx, TCMI_E_HIP, "hipEventCreate failed");
    auto work = [&](int k) {
        tcmi_ctx *cx = cx_of(k);
        if (skew) {
            if (k > 0) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return queued[(size_t)k - 1] != 0; });
                cx->ev_before_sym = cx_of(k - 1)->ev_after_sym;
            }
            cx->after_sym = [&release, k] { release(k); };
        }
        int rc = tcmi_readset_from_bamfile_blocks(cx, f, fb[(size_t)k], fb[(size_t)k + 1] - fb[(size_t)k], &rs[(size_t)k], nullptr);
        if (skew) { cx->after_sym = nullptr; cx->ev_before_sym = nullptr; release(k); }
        if (rc == TCMI_OK && rs[(size_t)k]->max_end > L) rc = tcmi_fail(cx, TCMI_E_ARG, "L=%lld is smaller than the reads' extent %lld", (long long)L, (long long)rs[(size_t)k]->max_end);
        if (rc == TCMI_OK && rs[(size_t)k]->n_piled) rc = tcmi_tally_dev(cx, rs[(size_t)k], L, ld, d_counts, 0);
        if (rc == TCMI_OK && hipStreamSynchronize(cx->stream) != hipSuccess) rc = tcmi_fail(cx, TCMI_E_HIP, "hipStreamSynchronize failed");
        rcs[(size_t)k] = rc;
    };
    {
        // (a thread that cannot be started — std::system_error must not cross the C ABI — has its sub-range done by this one, after its own;
        //  in order: a sub-range waits for the one in front to be queued)
        std::vector<std::thread> th;
        std::vector<int> inline_k;
        for (int k = 1; k < K; ++k) {
            try { th.emplace_back(work, k); }
            catch (const std::exception &) { inline_k.push_back(k); }
        }
        work(0);
        for (int k : inline_k) work(k);                             // (ascending: each waits only for the one in front of it to be queued)
        for (std::thread &t : th) t.join();
    }
    auto drop = [&]() { for (int k = 0; k < K; ++k) if (rs[(size_t)k]) tcmi_readset_free(k == 0 ? ctx : ctx->helpers[(size_t)k - 1], rs[(size_t)k]); };
    for (int k = 0; k < K; ++k)
        if (rcs[(size_t)k]) {
            const int rc = rcs[(size_t)k];
            const std::string why = (k == 0 ? ctx : ctx->helpers[(size_t)k - 1])->err;
            drop();
            (void)n_words;
            return tcmi_fail(ctx, rc == TCMI_E_UNSUPPORTED ? TCMI_E_UNSUPPORTED : rc, "sub-range %d of %d: %s", k, K, why.c_str());
        }
    // the joins: a sub-range behind the first vouches for nothing by itself
    std::vector<int64_t> rg((size_t)K * 4);
    bool plain = true;
    for (int k = 0; k < K; ++k) {
        rg[4 * (size_t)k] = fb[(size_t)k]; rg[4 * (size_t)k + 1] = fb[(size_t)k + 1] - fb[(size_t)k];
        rg[4 * (size_t)k + 2] = rs[(size_t)k]->range_first; rg[4 * (size_t)k + 3] = rs[(size_t)k]->range_next;
        if ((k > 0 && rs[(size_t)k]->range_first < 0) || rs[(size_t)k]->range_next < 0) plain = false;      // (a sub-range that found no record start, or lies inside one record)
    }
    int who = 0; int64_t at = 0, want = 0;
    const char *why = plain ? ranges_join(reinterpret_cast<const int64_t (*)[4]>(rg.data()), K, -1, &who, &at, &want) : "a sub-range holds no record start";
    if (why) {
        drop();
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "the sub-ranges of blocks [%lld, %lld) do not join (%s): the range in one piece", (long long)first, (long long)(first + cnt), why);
    }
    tcmi_readset *sum = new tcmi_readset();
    sum->device = ctx->device; sum->packed_on_device = 2;
    sum->range_first = rs[0]->range_first; sum->range_next = rs[(size_t)K - 1]->range_next;
    for (int k = 0; k < K; ++k) {
        const tcmi_readset *r = rs[(size_t)k];
        sum->n_reads += r->n_reads; sum->n_piled += r->n_piled; sum->alg_bytes += r->alg_bytes; sum->dev_bytes += r->dev_bytes;
        sum->max_end = std::max(sum->max_end, r->max_end); sum->max_len = std::max(sum->max_len, r->max_len); sum->s_reads += r->s_reads;
        sum->f_reads += r->f_reads;
        sum->parts.push_back({k == 0 ? ctx : ctx->helpers[(size_t)k - 1], rs[(size_t)k]});
    }
    *out = sum;
    return TCMI_OK;
}

// One rank's part of a step of ONE BAM file shared by several GPUs (include/tcmi.h): decode + pack + tally its block range, the
// caller's reduce hook, and on rank 0 the call kernel.  Behind the matrix: six words per rank (its range and anchors, in its own
// slot, zeros in the others' — the sum hands rank 0 the table) and one word that counts the ranks that failed.  Every rank enters
// the exchange exactly once, whatever happened to it before — a device that cannot be set, a workspace that cannot be allocated,
// a range that is refused: its share is then zeros + one failure.
int tcmi_split_step(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, int64_t L, int64_t ld, void *d_counts,
                    int32_t mincov, int include_ambig, tcmi_reduce_fn reduce, void *user, int rank, int world, tcmi_readset **rs_out,
                    const uint8_t **plain, const uint8_t **alt, const uint8_t **flags)
{
    if (!ctx || !f || !d_counts || !reduce || !rs_out) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (L <= 0 || ld < L) return tcmi_fail(ctx, TCMI_E_ARG, "need 0 < L <= ld");
    if (world < 1 || rank < 0 || rank >= world) return tcmi_fail(ctx, TCMI_E_ARG, "need 0 <= rank < world");
    *rs_out = nullptr;
    const bool is_root = rank == 0;
    const size_t n_mat = (size_t)ld * TCMI_NCOL, n_words = n_mat + (size_t)TCMI_SPLIT_TAIL_WORDS(world);
    int32_t *const d_all = static_cast<int32_t *>(d_counts);
    tcmi_readset *rs = nullptr;
    int own = TCMI_OK;
    std::string own_err;
    auto note = [&](int code) { own = code; own_err = ctx->err; };
    // (all of these are argument checks of the caller's on every rank alike — the collective has not been entered — or failures of THIS
    // rank, which must not keep it from the exchange)
    if (hipSetDevice(ctx->device) != hipSuccess) note(tcmi_fail(ctx, TCMI_E_HIP, "hipSetDevice(%d) failed", ctx->device));
    if (!own) { const int rc0 = ensure_ws(ctx, L); if (rc0) note(rc0); }
    if (!own && hipMemsetAsync(d_counts, 0, n_words * 4, ctx->stream) != hipSuccess) note(tcmi_fail(ctx, TCMI_E_HIP, "hipMemsetAsync of the count matrix failed"));
    int64_t all = 0, inflated = 0;
    (void)tcmi_bamfile_info(f, nullptr, &inflated, &all, nullptr, nullptr, nullptr);
    bool tallied = false;
    if (!own) {                                                 // sub-ranges side by side where the range is large enough (else, or if they do not work out: in one piece)
        const int64_t cnt0 = std::max<int64_t>(0, n_blocks < 0 ? all - first_block : std::min<int64_t>(n_blocks, all - first_block));
        static const int sub_env = std::getenv("TCMI_SPLIT_SUB") ? std::atoi(std::getenv("TCMI_SPLIT_SUB")) : 0;
        int K = sub_env > 0 ? sub_env : ctx->split_sub;
        // (auto: a true RANGE of a larger file gains ~ 10 % from three sub-ranges — 2.65 - 2.83 -> 2.42 - 2.54 ms at 4 M reads, 3.9 -> 3.5 - 3.6 ms
        //  at 6.25 M; the WHOLE file as one rank's range — world 1 — is decoded without the range's re-based block table and loses 3 - 8 % to them:
        //  profiles/r06i_split_ab.log, r06j_*)
        const bool whole = first_block == 0 && cnt0 == all;
        if (K <= 0) K = whole ? 1 : cnt0 >= 6144 ? 3 : cnt0 >= 4096 ? 2 : 1;
        K = (int)std::min<int64_t>(std::min(K, 8), std::max<int64_t>(1, cnt0 / 64));
        if (K > 1 && !ctx->stream_hi) {
            const int rc0 = split_sub_ranges(ctx, f, first_block, cnt0, K, L, ld, d_counts, n_words, &rs);
            // (whatever kept the sub-ranges from working out — they do not join, a helper context could not be made, a sub-range was
            //  refused — the range in one piece has the last word: it words the refusal, or simply works)
            if (rc0 == TCMI_OK) { tallied = true; ++ctx->stat_split_sub; }
            else if (hipMemsetAsync(d_counts, 0, n_words * 4, ctx->stream) != hipSuccess) note(tcmi_fail(ctx, TCMI_E_HIP, "hipMemsetAsync of the count matrix failed"));
        }
    }
    if (!own && !tallied) { const int rc0 = tcmi_readset_from_bamfile_blocks(ctx, f, first_block, n_blocks, &rs, nullptr); if (rc0) note(rc0); }
    if (!own && rs->max_end > L) note(tcmi_fail(ctx, TCMI_E_ARG, "L=%lld is smaller than the reads' extent %lld", (long long)L, (long long)rs->max_end));
    if (!own && !tallied && rs->n_piled) { const int rc0 = tcmi_tally_dev(ctx, rs, L, ld, d_counts, 0); if (rc0) note(rc0); }
    if (own) {                                                  // this rank's share is zeros + one failure
        static const int32_t one = 1;
        (void)hipMemsetAsync(d_counts, 0, n_words * 4, ctx->stream);
        (void)hipMemcpyAsync(d_all + (n_words - 1), &one, 4, hipMemcpyHostToDevice, ctx->stream);
    } else {
        const int64_t cnt = std::max<int64_t>(0, n_blocks < 0 ? all - first_block : std::min<int64_t>(n_blocks, all - first_block));
        const int64_t v[3] = {rs->range_first, rs->range_next, 0};
        int32_t *t = ctx->split_tail;
        t[0] = (int32_t)first_block; t[1] = (int32_t)cnt;
        std::memcpy(t + 2, &v[0], 8); std::memcpy(t + 4, &v[1], 8);
        (void)hipMemcpyAsync(d_all + n_mat + (size_t)6 * rank, t, 24, hipMemcpyHostToDevice, ctx->stream);
    }
    const int rrc = reduce(user, d_counts, (int64_t)n_words, (void *)ctx->stream);
    if (rrc) { if (rs) tcmi_readset_free(ctx, rs); return tcmi_fail(ctx, TCMI_E_HIP, "the reduce hook failed (%d)", rrc); }
    int rc = TCMI_OK;
    std::vector<int32_t> tail((size_t)TCMI_SPLIT_TAIL_WORDS(world), 0);
    if (is_root && !own) {
        rc = tcmi_launch_call(ctx, d_all, L, ld, mincov, include_ambig, 0, ctx->h_rec, ctx->h_rec + ctx->ws_ld, ctx->h_rec + 2 * ctx->ws_ld, nullptr, nullptr);
        if (!rc && hipMemcpyAsync(tail.data(), d_all + n_mat, tail.size() * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = tcmi_fail(ctx, TCMI_E_HIP, "copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc && !own) rc = tcmi_fail(ctx, TCMI_E_HIP, "hipStreamSynchronize failed");
    ctx->counts_clean = false;
    if (own) { if (rs) tcmi_readset_free(ctx, rs); return tcmi_fail(ctx, own, "%s", own_err.c_str()); }
    if (rc) { tcmi_readset_free(ctx, rs); return rc; }
    const int32_t failed = tail.back();
    if (is_root && failed) { tcmi_readset_free(ctx, rs); return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%d rank(s) could not decode their range of %s on the device", (int)failed, tcmi_bamfile_path(f)); }
    if (is_root) {
        std::vector<int64_t> rg((size_t)world * 4);
        for (int r = 0; r < world; ++r) {
            const int32_t *t = tail.data() + (size_t)6 * r;
            rg[4 * r] = t[0]; rg[4 * r + 1] = t[1];
            std::memcpy(&rg[4 * r + 2], t + 2, 8); std::memcpy(&rg[4 * r + 3], t + 4, 8);
        }
        int who = 0; int64_t at = 0, want = 0;
        const char *why = ranges_join(reinterpret_cast<const int64_t (*)[4]>(rg.data()), world, inflated, &who, &at, &want);
        if (why) {
            tcmi_readset_free(ctx, rs);
            return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: the ranks' block ranges do not join into one chain of alignment records (rank %d %s: offset %lld, expected %lld): host reader",
                             tcmi_bamfile_path(f), who, why, (long long)at, (long long)want);
        }
    }
    *rs_out = rs;
    if (plain) *plain = ctx->h_rec;
    if (alt) *alt = ctx->h_rec + ctx->ws_ld;
    if (flags) *flags = ctx->h_rec + 2 * ctx->ws_ld;
    return TCMI_OK;
}

int tcmi_step(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig,
              const uint8_t **plain, const uint8_t **alt, const uint8_t **flags, const int32_t **counts_planes,
              int64_t *ld_out)
{
    const int rc = tcmi_step_begin(ctx, rs, L, mincov, include_ambig, counts_planes != nullptr);
    if (rc) return rc;
    return tcmi_step_end(ctx, plain, alt, flags, counts_planes, ld_out);
}

} // extern "C"
