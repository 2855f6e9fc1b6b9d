// readset.cpp — HOST: selects the reads that pile up, splits them into the
// ALIGNED and GENERAL device sets (layout: tcmi_internal.h) and copies them into HBM.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <memory>

#include "tcmi_internal.h"

namespace {

inline bool consumes_ref(unsigned op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
inline bool is_match(unsigned op) { return op == 0 || op == 7 || op == 8; }

int64_t ref_span(const uint32_t *cg, int64_t n)
{
    int64_t s = 0;
    for (int64_t k = 0; k < n; ++k)
        if (consumes_ref(cg[k] & 0xF)) s += cg[k] >> 4;
    return s;
}

// Reads on a second or later reference: the reference implementation piles them up keyed by position only, so they
// collide with the first reference's columns and its BuildIndex fails (indexing.py:137-151).  They never
// pile up here, and an upload that meets a mapped one fails with TCMI_E_UNSUPPORTED instead of tallying it onto
// reference 0's coordinates.
inline bool piles_up(const tcmi_reads *r, int64_t i, int64_t *span)
{
    if (r->flag[i] & 0x4) return false;
    if (r->tid && r->tid[i] != 0) return false;
    if (r->pos[i] < 0) return false;
    *span = ref_span(r->cigar + r->cigar_off[i], (int64_t)(r->cigar_off[i + 1] - r->cigar_off[i]));
    return *span > 0;
}

int check_reads(tcmi_ctx *ctx, const tcmi_reads *r)
{
    if (!r) return tcmi_fail(ctx, TCMI_E_ARG, "reads is NULL");
    if (r->n_reads < 0) return tcmi_fail(ctx, TCMI_E_ARG, "n_reads < 0");
    if (r->n_reads > 0 && (!r->pos || !r->flag || !r->l_qseq || !r->cigar_off || !r->seq_off))
        return tcmi_fail(ctx, TCMI_E_ARG, "reads has NULL arrays");
    return TCMI_OK;
}

// htslib resolve_cigar2's peek at the last reference base of op k: is an insertion reported there?
bool ins_after(const uint32_t *cg, int64_t n, int64_t k)
{
    if (k + 1 >= n) return false;
    const unsigned op2 = cg[k + 1] & 0xF;
    int64_t tot = 0;
    if (op2 == 1) {
        tot = cg[k + 1] >> 4;
        for (int64_t j = k + 2; j < n; ++j) {
            const unsigned o = cg[j] & 0xF;
            if (o == 1) tot += cg[j] >> 4;
            else if (o != 6) break;
        }
    } else if (op2 == 6 && k + 2 < n) {
        for (int64_t j = k + 2; j < n; ++j) {
            const unsigned o = cg[j] & 0xF;
            if (o == 1) tot += cg[j] >> 4;
            else if (consumes_ref(o)) break;
        }
    }
    return tot > 0;
}

// [H]*[S]* (M|=|X)+ [S]*[H]*  ->  query offset of the first aligned base, aligned length
bool aligned_shape(const uint32_t *cg, int64_t n, int64_t *y0, int64_t *len)
{
    int64_t k = 0, clip = 0, m = 0;
    while (k < n && (cg[k] & 0xF) == 5) ++k;
    while (k < n && (cg[k] & 0xF) == 4) { clip += cg[k] >> 4; ++k; }
    if (k == n || !is_match(cg[k] & 0xF)) return false;
    while (k < n && is_match(cg[k] & 0xF)) { m += cg[k] >> 4; ++k; }
    while (k < n && (cg[k] & 0xF) == 4) ++k;
    while (k < n && (cg[k] & 0xF) == 5) ++k;
    if (k != n || m <= 0 || m > TCMI_F_MAXSPAN) return false;
    *y0 = clip;
    *len = m;
    return true;
}

// BAM byte (two 4-bit codes, first base in the high nibble) -> two one-hot class nibbles in
// linear order (first base in the low nibble); codes other than A/C/G/T become 0.
struct SwapLut {
    uint8_t t[256];
    SwapLut()
    {
        auto oh = [](unsigned c) -> unsigned { return (c == 1 || c == 2 || c == 4 || c == 8) ? c : 0; };
        for (unsigned b = 0; b < 256; ++b) t[b] = (uint8_t)(oh(b >> 4) | (oh(b & 15) << 4));
    }
};
const SwapLut kSwap;

struct Up {
    tcmi_ctx *ctx;
    tcmi_readset *rs;
    int operator()(void **d, const void *h, size_t bytes)
    {
        // +256 B slack so 16-byte vector loads around the last elements stay inside the allocation
        hipError_t e = hipMalloc(d, bytes + 256);
        if (e != hipSuccess) return tcmi_fail(ctx, TCMI_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        e = hipMemsetAsync((char *)*d + bytes, 0, 256, ctx->stream);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return tcmi_fail(ctx, TCMI_E_HIP, "upload failed: %s", hipGetErrorString(e));
        rs->dev_bytes += (int64_t)bytes;
        return TCMI_OK;
    }
};

// one entry of the aligned set: read i of BAM r (positions shifted by off); for a projected read the piece
// [seg, seg + len) of its reference span
struct Sel { const tcmi_reads *r; int64_t i, off, y0, len, seg; bool projected; };
struct GSel { const tcmi_reads *r; int64_t i, off; };
struct Part {                       // what one classification thread found in its slice of a BAM
    std::vector<Sel> fsel; std::vector<GSel> gsel;
    int64_t g_cig = 0, g_seqw = 0, alg = 0, max_end = 0; bool any_cut = false;
    int err = TCMI_OK; char msg[160] = {0};
};

} // namespace

// Host buffers of tcmi_readset_upload, kept per context between calls: an upload of 1 M reads walks through
// ~200 MB of them, and a third of its time used to go into page faults of fresh allocations and their release.
struct tcmi_upload_scratch {
    std::vector<Sel> fsel;
    std::vector<GSel> gsel;
    std::vector<Part> parts;
    std::vector<uint32_t> f_lenoff, f_event, f_covrun;
    uint32_t *f_seq = nullptr;
    size_t f_seq_cap = 0;
    ~tcmi_upload_scratch() { delete[] f_seq; }
    size_t bytes() const
    {
        size_t b = fsel.capacity() * sizeof(Sel) + gsel.capacity() * sizeof(GSel) + f_lenoff.capacity() * 4 +
                   f_event.capacity() * 4 + f_covrun.capacity() * 4 + f_seq_cap * 4;
        for (const Part &p : parts) b += p.fsel.capacity() * sizeof(Sel) + p.gsel.capacity() * sizeof(GSel);
        return b;
    }
};

void tcmi_upload_scratch_free(tcmi_upload_scratch *s) { delete s; }

extern "C" {

int tcmi_reads_extent(const tcmi_reads *r, int64_t ref_len, int64_t *out_L)
{
    int rc = check_reads(nullptr, r);
    if (rc) return rc;
    if (!out_L) return tcmi_fail(nullptr, TCMI_E_ARG, "out_L is NULL");
    int64_t L = ref_len > 0 ? ref_len : 0;
    for (int64_t i = 0; i < r->n_reads; ++i) {
        int64_t span;
        if (!piles_up(r, i, &span)) continue;
        if (r->pos[i] + span > L) L = r->pos[i] + span;
    }
    *out_L = L;
    return TCMI_OK;
}

int tcmi_readset_free(tcmi_ctx *ctx, tcmi_readset *rs)
{
    if (!rs) return TCMI_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (tcmi_readset::Part &p : rs->parts) (void)tcmi_readset_free(p.cx, p.rs);      // (a read set of sub-ranges: each part with its own context)
    rs->parts.clear();
    if (rs->d_blob) {                                            // device-packed: one allocation holds the aligned set
        if (ctx && ctx->device == rs->device && ctx->blob_pool.size() < 4 && rs->blob_bytes)
            ctx->blob_pool.push_back({rs->d_blob, rs->blob_bytes});   // (stream order: the next user's kernels queue behind this one's)
        else
            (void)hipFree(rs->d_blob);
        rs->d_flenoff = nullptr; rs->d_fseq = nullptr; rs->d_fevent = nullptr; rs->d_fchunk = nullptr; rs->d_fcovrun = nullptr;
    }
    void *ptrs[] = {rs->d_flenoff, rs->d_fseq, rs->d_fevent, rs->d_fchunk, rs->d_fcovrun, rs->d_pos, rs->d_meta,
                    rs->d_lseq, rs->d_cigar, rs->d_seq, rs->d_round_cig, rs->d_round_seq};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete rs;
    return TCMI_OK;
}

// One or several BAMs into one read set; BAM b's positions are shifted by b * stride, so that the
// kernels see one long coordinate axis and a single launch tallies the whole batch.
static int upload_impl(tcmi_ctx *ctx, const tcmi_reads *const *batch, int32_t n_batch, int64_t stride, tcmi_readset **out)
{
    if (!ctx || !out || !batch || n_batch < 1) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    *out = nullptr;
    int rc = TCMI_OK;
    int64_t n_reads_in = 0;
    for (int32_t b = 0; b < n_batch; ++b) {
        rc = check_reads(ctx, batch[b]);
        if (rc) return rc;
        n_reads_in += batch[b]->n_reads;
    }
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    const bool use_fast = ctx->tally_variant != 1;
    const bool timing = std::getenv("TCMI_UPLOAD_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    static std::atomic<uint64_t> next_uid{1};

    // ---- default: the BAM-native arrays go to the device as they are and HIP kernels pack them (pack_device.hip) ----
    if (ctx->device_pack && use_fast && ctx->project_reads && n_batch == 1) {
        const tcmi_reads *r = batch[0];
        bool multi_ref = false;
        tcmi_readset *rs = new tcmi_readset();
        rs->uid = next_uid.fetch_add(1);
        rs->n_reads = r->n_reads;
        rs->device = ctx->device;
        uint32_t why = 0;
        rc = tcmi_upload_and_pack_on_device(ctx, r, rs, &why);
        if (rc == TCMI_OK) {
            if (timing) std::fprintf(stderr, "[tcmi upload] device pack: %.1f ms (%.1f MB on the device)\n", ms(t0, now()), rs->dev_bytes / 1e6);
            *out = rs;
            return TCMI_OK;
        }
        tcmi_readset_free(ctx, rs);
        if (rc != TCMI_E_UNSUPPORTED) return rc;
        (void)multi_ref;
        if (timing) std::fprintf(stderr, "[tcmi upload] device pack declined (flags 0x%x): host packer\n", why);
        rc = TCMI_OK;
    }

    // pass 1: select, classify, size
    // one entry of the aligned set: read i of BAM r (positions shifted by off); for a projected read the piece
    // [seg, seg + len) of its reference span (long reads are cut into pieces of <= TCMI_F_SEG positions)
    if (!ctx->upload_scratch) ctx->upload_scratch = new tcmi_upload_scratch();
    tcmi_upload_scratch &SC = *ctx->upload_scratch;
    struct Trim {                           // big batches do not keep their gigabytes around
        tcmi_ctx *c;
        ~Trim() { if (c->upload_scratch && c->upload_scratch->bytes() > ((size_t)768 << 20)) { tcmi_upload_scratch_free(c->upload_scratch); c->upload_scratch = nullptr; } }
    } trim{ctx};
    std::vector<Sel> &fsel = SC.fsel;       // aligned set (len > 0)
    std::vector<GSel> &gsel = SC.gsel;      // general set
    fsel.clear();
    gsel.clear();
    int64_t g_cig = 0, g_seqw = 0, alg = 0, max_end = 0;
    bool any_cut = false;
    // every BAM's reads in `host_threads` contiguous slices, each into its own lists, joined in order afterwards
    const int n_cls = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ctx->host_threads, 64, n_reads_in / 65536 + 1}));
    std::vector<Part> &parts = SC.parts;
    parts.resize((size_t)n_batch * (size_t)n_cls);
    for (Part &P : parts) { P.fsel.clear(); P.gsel.clear(); P.g_cig = P.g_seqw = P.alg = P.max_end = 0; P.any_cut = false; P.err = TCMI_OK; }
    auto classify = [&](int32_t bi, int t) {
        Part &P = parts[(size_t)bi * (size_t)n_cls + (size_t)t];
        const tcmi_reads *r = batch[bi];
        const int64_t off = (int64_t)bi * stride;
        const int64_t i0 = r->n_reads * t / n_cls, i1 = r->n_reads * (t + 1) / n_cls;
        auto fail = [&](int code, const char *fmt, long long x, long long y, long long z) {
            P.err = code;
            std::snprintf(P.msg, sizeof P.msg, fmt, x, y, z);
        };
        for (int64_t i = i0; i < i1; ++i) {
            int64_t span;
            if (r->tid && r->tid[i] > 0 && !(r->flag[i] & 0x4))
                return fail(TCMI_E_UNSUPPORTED, "read %lld is mapped to reference %lld: only single-reference alignments are supported "
                                                "(the reference implementation keys columns by position only and fails on these)%.0lld", i, r->tid[i], 0);
            if (!piles_up(r, i, &span)) continue;
            const uint32_t *cg = r->cigar + r->cigar_off[i];
            const int64_t nc = (int64_t)(r->cigar_off[i + 1] - r->cigar_off[i]);
            if (nc > 65535) return fail(TCMI_E_UNSUPPORTED, "read %lld has %lld CIGAR ops (> 65535)%.0lld", i, nc, 0);
            const int64_t lq = r->l_qseq[i];

            if (lq < 0) return fail(TCMI_E_ARG, "read %lld has negative l_qseq%.0lld%.0lld", i, 0, 0);
            const int64_t nbytes = (int64_t)(r->seq_off[i + 1] - r->seq_off[i]);
            if (nbytes < (lq + 1) / 2) return fail(TCMI_E_ARG, "read %lld: seq bytes %lld < ceil(l_qseq/2)%.0lld", i, nbytes, 0);
            if (n_batch > 1 && r->pos[i] + span > stride)
                return fail(TCMI_E_ARG, "read %lld of a batched BAM ends at %lld, beyond the batch stride %lld", i, r->pos[i] + span, stride);
            if (span > INT32_MAX || off + r->pos[i] + span > INT32_MAX - 4096)
                return fail(TCMI_E_UNSUPPORTED, "read %lld ends beyond 2^31%.0lld%.0lld", i, 0, 0);
            P.alg += 12 + 4 * nc + (lq + 1) / 2;
            if (off + r->pos[i] + span > P.max_end) P.max_end = off + r->pos[i] + span;
            int64_t y0, len;
            if (use_fast && off + r->pos[i] + span < TCMI_F_EVPOS && aligned_shape(cg, nc, &y0, &len)) P.fsel.push_back({r, i, off, y0, len, 0, false});
            else if (use_fast && ctx->project_reads && off + r->pos[i] + span < TCMI_F_EVPOS) {
                // any CIGAR, projected onto the reference; a long read in pieces (the count matrix is a sum over
                // positions, so cutting a read changes nothing)
                for (int64_t seg = 0; seg < span; seg += TCMI_F_SEG)
                    P.fsel.push_back({r, i, off, 0, std::min<int64_t>(TCMI_F_SEG, span - seg), seg, true});
                if (span > TCMI_F_SEG) P.any_cut = true;
            }
            else { P.gsel.push_back({r, i, off}); P.g_cig += nc; P.g_seqw += (lq + 7) / 8; }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_cls; ++t)
            th.emplace_back([&, t] { for (int32_t bi = 0; bi < n_batch; ++bi) classify(bi, t); });
        for (int32_t bi = 0; bi < n_batch; ++bi) classify(bi, 0);
        for (auto &x : th) x.join();
    }
    {
        size_t nfs = 0, ngs = 0;
        for (const Part &P : parts) {
            if (P.err) return tcmi_fail(ctx, P.err, "%s", P.msg);
            nfs += P.fsel.size(); ngs += P.gsel.size();
        }
        fsel.reserve(nfs); gsel.reserve(ngs);
        for (Part &P : parts) {
            fsel.insert(fsel.end(), P.fsel.begin(), P.fsel.end());
            gsel.insert(gsel.end(), P.gsel.begin(), P.gsel.end());
            g_cig += P.g_cig; g_seqw += P.g_seqw; alg += P.alg; max_end = std::max(max_end, P.max_end); any_cut |= P.any_cut;
        }
    }

    if (any_cut)                                // pieces of long reads start further right than the reads that follow them
        std::stable_sort(fsel.begin(), fsel.end(), [](const Sel &a, const Sel &b) {
            return a.r->pos[a.i] + a.off + a.seg < b.r->pos[b.i] + b.off + b.seg;
        });
    const auto t1 = now();
    // ---- aligned set: chunks, stages, padded one-hot bases, "other" positions ----------------
    const int64_t nf = (int64_t)fsel.size();
    // Layout of the base stream (tcmi_internal.h): {lo, hi} plane pairs.
    const int64_t prefix = 2;                                                    // zero words in front of a chunk's first read
    auto read_words = [&](int64_t len) -> int64_t {                              // words of one read, trailing zero pair included
        return 2 * ((len + 31) / 32) + 2;
    };
    // chunk_stages = 0: long chunks (up to 8 stages: the spread / reduce epilogue is paid once per chunk),
    // but capped so that the launch has k * (4 workgroups per CU) chunks — with 2 315 chunks on 1 024 slots the third
    // round of workgroups ran a quarter full.
    const int n_stages = ctx->chunk_stages > 0 ? std::min(ctx->chunk_stages, TCMI_F_MAXSTAGE) : TCMI_F_MAXSTAGE;
    int64_t balanced_cap = INT64_MAX;
    if (ctx->chunk_stages == 0 && ctx->balance_chunks && nf > 0) {
        const int64_t slots = (int64_t)ctx->n_cu * ctx->wg_per_cu, longest = (int64_t)TCMI_F_MAXSTAGE * 400;   // ~ 400 reads per stage at 5 000x / 150 bp
        const int64_t k = (nf + slots * longest - 1) / (slots * longest);
        balanced_cap = std::max<int64_t>(64, (nf + k * slots - 1) / (k * slots));
    }
    std::vector<uint32_t> &f_event = SC.f_event;      // position | TCMI_F_EV_* : tokens that are not plain A/C/G/T bases
    std::vector<uint32_t> &f_lenoff = SC.f_lenoff;
    std::vector<uint32_t> &f_covrun = SC.f_covrun;    // format 2: coverage runs (tcmi_fast_chunk::run0 / n_runs)
    f_lenoff.resize((size_t)nf);
    f_event.clear();
    f_covrun.clear();
    std::vector<tcmi_fast_chunk> chunks;
    uint32_t *f_seq = nullptr;                  // (SC.f_seq) not zero-filled: every packing thread clears its own chunks
    size_t f_seq_n = 0;
    {
        int64_t c_read0 = 0, c_lo = 0, c_hi = 0, c_maxnw = 0, c_n = 0;
        // Stage size for a window of `words` grid words and reads of <= maxnw grid words: lanes own 32 positions, the
        // kernel splits a stage over S = 256 / ceil(window / 32) depth slices and its inner loop takes bodies of 8
        // reads per lane and one of 4: a stage of S * 4 * m reads wastes none.  Fill the stage buffer.
        auto stage_reads = [&](int64_t words, int64_t maxnw) -> int64_t {
            const int64_t S = TCMI_F_BLOCK / std::max<int64_t>(2, (words * 8 + 31) / 32)   /* (the kernel keeps at least two lane groups) */;
            int64_t cap = std::min<int64_t>(TCMI_P_SUB, (TCMI_F_SEQCAP - 16 - prefix) / read_words(maxnw * 8));
            if (ctx->stage_cap > 0) cap = std::min<int64_t>(cap, std::max<int64_t>(ctx->stage_cap, S * 4));   // (experiments)
            int64_t sub = S * 4 * std::max<int64_t>(1, cap / (S * 4));
            if (sub > cap) sub = std::max<int64_t>(S, cap / S * S);
            return sub;
        };
        auto chunk_reads = [&](int64_t sub, int64_t words) -> int64_t {          // whole stages, <= 2^planes - 1 reads per lane
            const int64_t S = TCMI_F_BLOCK / std::max<int64_t>(2, (words * 8 + 31) / 32)   /* (the kernel keeps at least two lane groups) */;
            const int64_t whole = std::max<int64_t>(sub, std::min<int64_t>(((1 << TCMI_P_NPL) - 1) * S, n_stages * sub) / sub * sub);
            return std::min(whole, balanced_cap);
        };
        auto close = [&](int64_t next_read) {
            if (c_n == 0) return;
            tcmi_fast_chunk c;
            std::memset(&c, 0, sizeof c);
            c.read0 = c_read0;
            c.n_reads = (int32_t)c_n;
            c.P0 = (int32_t)c_lo;
            c.Wn = (int32_t)((c_hi - c_lo + 7) / 8);
            c.sub_reads = (int32_t)stage_reads(c.Wn, c_maxnw);
            chunks.push_back(c);
            c_read0 = next_read;
            c_n = 0;
        };
        for (int64_t j = 0; j < nf; ++j) {
            const int64_t p = fsel[(size_t)j].r->pos[fsel[(size_t)j].i] + fsel[(size_t)j].off + fsel[(size_t)j].seg, e = p + fsel[(size_t)j].len;
            const int64_t lo = p & ~(int64_t)7, nw = (fsel[(size_t)j].len + 7) / 8;
            if (c_n > 0) {
                const int64_t nlo = std::min(c_lo, lo), nhi = std::max(c_hi, e), nmax = std::max(c_maxnw, nw);
                const int64_t words = (nhi - nlo + 7) / 8;
                if (words > TCMI_F_MAXW || c_n >= chunk_reads(stage_reads(words, nmax), words)) close(j);
                else { c_lo = nlo; c_hi = nhi; c_maxnw = nmax; }
            }
            if (c_n == 0) { c_read0 = j; c_lo = lo; c_hi = e; c_maxnw = nw; }
            ++c_n;
        }
        close(nf);
        // the base stream, chunk by chunk: [pad] read [pad] read [pad] ... each chunk 16-byte aligned.
        // Sizes first (sequential, cheap), then the chunks are packed by `n_threads` host threads.
        {
            size_t total = 0;
            for (auto &c : chunks) {
                total = (total + 3) & ~(size_t)3;
                c.word0 = (int64_t)total;
                total += (size_t)prefix;
                for (int64_t j = c.read0; j < c.read0 + c.n_reads; ++j) total += (size_t)read_words(fsel[(size_t)j].len);
            }
            total = (total + 3) & ~(size_t)3;
            if (SC.f_seq_cap < total + 4) {
                delete[] SC.f_seq;
                SC.f_seq = nullptr;
                SC.f_seq_cap = 0;
                SC.f_seq = new uint32_t[total + 4 + total / 16];
                SC.f_seq_cap = total + 4 + total / 16;
            }
            f_seq = SC.f_seq;
            f_seq_n = total;
        }
        const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ctx->host_threads, (int64_t)chunks.size(), 64}));
        std::vector<std::vector<uint32_t>> ev_parts((size_t)n_threads);
        std::vector<std::vector<uint32_t>> run_parts((size_t)n_threads);   // format 2: coverage runs, chunk by chunk
        std::atomic<bool> pack_overflow{false};
        auto pack_range = [&](int t) {
            std::vector<uint32_t> &ev = ev_parts[(size_t)t];
            std::vector<uint32_t> &runs = run_parts[(size_t)t];
            std::vector<uint32_t> scratch;                                        // format 2: a read's nibbles before they become planes
            const size_t c0 = chunks.size() * (size_t)t / (size_t)n_threads, c1 = chunks.size() * (size_t)(t + 1) / (size_t)n_threads;
            for (size_t ci = c0; ci < c1; ++ci) {
                tcmi_fast_chunk &c = chunks[ci];
                const size_t c_end = ci + 1 < chunks.size() ? (size_t)chunks[ci + 1].word0 : f_seq_n;
                std::memset(&f_seq[(size_t)c.word0], 0, (c_end - (size_t)c.word0) * 4);      // pads and alignment gaps stay zero
                size_t cursor = (size_t)c.word0 + (size_t)prefix;
                size_t stage_begin = (size_t)c.word0;
                c.run0 = (int64_t)runs.size();                   // (made global below, once the threads' parts are joined)
                uint32_t run_key = 0xFFFFFFFFu;
            for (int64_t j = c.read0; j < c.read0 + c.n_reads; ++j) {
                const Sel &s = fsel[(size_t)j];
                const tcmi_reads *r = s.r;
                const int64_t rpos = r->pos[s.i] + s.off + s.seg;      // reference position of the entry's first base
                const uint8_t *src = r->seq + r->seq_off[s.i];
                const int64_t lq = r->l_qseq[s.i];
                const int64_t nw = (s.len + 7) / 8;
                const size_t base = cursor;
                // ONE packed word per read — position relative to the window | len << 10 | pair offset from
                // the stage's first word << 20 (a stage starts on the zero pair in front of its first read)
                {
                    // (10 + 10 + 12 bits: the chunker keeps windows <= 768 positions, entries <= 600 positions and
                    // stages <= 6144 words; checked, not assumed)
                    if (rpos - c.P0 > 1023 || s.len > 1023 || (base - stage_begin) / 2 > 4095) pack_overflow.store(true);
                    f_lenoff[(size_t)j] = (uint32_t)(rpos - c.P0) | ((uint32_t)s.len << 10) | ((uint32_t)((base - stage_begin) / 2) << 20);
                    // coverage: reads of equal (position, length) follow each other in a sorted BAM — one run word per
                    // group instead of per-read bookkeeping in the kernel
                    const uint32_t key = (uint32_t)(rpos - c.P0) | ((uint32_t)s.len << 10);
                    if (key == run_key && (runs.back() >> 20) < 4095u) runs.back() += 1u << 20;
                    else { runs.push_back(key | (1u << 20)); run_key = key; }
                }
                cursor += (size_t)read_words(s.len);
                scratch.assign((size_t)nw + 1, 0u);
                uint8_t *dst = reinterpret_cast<uint8_t *>(scratch.data());
                const int64_t have = std::max<int64_t>(0, std::min(s.len, lq - s.y0));   // bases present in SEQ
                if (s.projected) {
                    // Walk the CIGAR once: matched bases land on their reference offset, D / N leave zero
                    // nibbles (coverage only), and the tokens that are not plain bases become events
                    // (SURVEY §8-P6): X for a deleted base whose token is exactly "*", I on the last
                    // reference base before an insertion (also "*+..": I but not X).
                    const uint32_t *cg = r->cigar + r->cigar_off[s.i];
                    const int64_t nc = (int64_t)(r->cigar_off[s.i + 1] - r->cigar_off[s.i]);
                    // x = offset in the read's reference span, relative to this piece [0, s.len)
                    int64_t x = -s.seg, y = 0;
                    for (int64_t k = 0; k < nc && x < s.len; ++k) {
                        const unsigned op = cg[k] & 0xF;
                        const int64_t len = cg[k] >> 4;
                        if (consumes_ref(op)) {
                            const bool ins = len > 0 && ins_after(cg, nc, k);
                            const int64_t t0 = std::max<int64_t>(0, -x), t1 = std::min(len, s.len - x);   // part inside the piece
                            if (is_match(op)) {
                                for (int64_t t = t0; t < t1; ++t) {
                                    const int64_t q = y + t;
                                    if (q >= lq) break;
                                    const unsigned code = (q & 1) ? (src[q >> 1] & 15u) : (src[q >> 1] >> 4);
                                    const unsigned oh = (code == 1 || code == 2 || code == 4 || code == 8) ? code : 0;
                                    dst[(x + t) >> 1] |= (uint8_t)(oh << (((x + t) & 1) * 4));
                                }
                            } else if (op == 2) {
                                for (int64_t t = t0; t < std::min(t1, ins ? len - 1 : len); ++t)
                                    ev.push_back((uint32_t)(rpos + x + t) | TCMI_F_EV_X);
                            }
                            if (ins && x + len - 1 >= 0 && x + len - 1 < s.len) ev.push_back((uint32_t)(rpos + x + len - 1) | TCMI_F_EV_I);
                            x += len;
                        }
                        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) y += len;
                    }
                } else if ((s.y0 & 1) == 0) {
                    const uint8_t *b = src + (s.y0 >> 1);
                    const int64_t full = have >> 1;
                    for (int64_t k = 0; k < full; ++k) dst[k] = kSwap.t[b[k]];
                    if (have & 1) dst[full] = (uint8_t)(kSwap.t[b[full]] & 0x0F);
                } else {
                    for (int64_t k = 0; k < have; ++k) {
                        const int64_t q = s.y0 + k;
                        const unsigned code = (q & 1) ? (src[q >> 1] & 15u) : (src[q >> 1] >> 4);
                        const unsigned oh = (code == 1 || code == 2 || code == 4 || code == 8) ? code : 0;
                        dst[k >> 1] |= (uint8_t)(oh << ((k & 1) * 4));
                    }
                }
                // bases that are no A/C/G/T: rare, found a word at a time
                const uint32_t *w = scratch.data();
                for (int64_t k = 0; k < nw; ++k) {
                    const uint32_t v = w[k];
                    uint32_t nz = (v | (v >> 1) | (v >> 2) | (v >> 3)) & 0x11111111u;   // 1 per non-zero nibble
                    const int64_t in_read = std::min<int64_t>(8, s.len - 8 * k);
                    const uint32_t want = in_read >= 8 ? 0x11111111u : (0x11111111u >> (4 * (8 - in_read)));
                    uint32_t miss = want & ~nz;
                    while (miss) {
                        const int bit = __builtin_ctz(miss);
                        ev.push_back((uint32_t)(rpos + 8 * k + bit / 4) | TCMI_F_EV_OTHER);
                        miss &= miss - 1;
                    }
                }
                {
                    // one-hot nibbles -> codes A=0 C=1 G=2 T=3 (class-less = 0) as {lo plane, hi plane} per 32 bases
                    auto squeeze = [](uint32_t x) -> uint32_t {                 // bits 0,4,..,28 -> bits 0..7
                        x = (x | (x >> 3)) & 0x03030303u;
                        x = (x | (x >> 6)) & 0x000F000Fu;
                        return (x | (x >> 12)) & 0xFFu;
                    };
                    uint32_t *out = &f_seq[base];
                    for (int64_t q = 0; q < (s.len + 31) / 32; ++q) {
                        uint32_t lo = 0, hi = 0;
                        for (int64_t k = 0; k < 4 && 4 * q + k < nw; ++k) {
                            const uint32_t v = w[4 * q + k];
                            lo |= squeeze(((v >> 1) | (v >> 3)) & 0x11111111u) << (8 * k);   // C or T
                            hi |= squeeze(((v >> 2) | (v >> 3)) & 0x11111111u) << (8 * k);   // G or T
                        }
                        out[2 * q] = lo;
                        out[2 * q + 1] = hi;
                    }
                }
                if ((j - c.read0 + 1) % c.sub_reads == 0 || j + 1 == c.read0 + c.n_reads) {
                    c.stage_end[(j - c.read0) / c.sub_reads] = (int32_t)(cursor - (size_t)c.word0);
                    stage_begin = cursor - 2;                        // the next stage starts on this read's zero pair
                }
                if (j + 1 == c.read0 + c.n_reads) c.n_runs = (int32_t)((int64_t)runs.size() - c.run0);
            }
            }
        };
        if (n_threads == 1) pack_range(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; ++t) th.emplace_back(pack_range, t);
            for (auto &x : th) x.join();
        }
        if (pack_overflow.load())
            return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "internal: a packed read header field overflowed (window / length / stage offset)");
        for (auto &part : ev_parts) f_event.insert(f_event.end(), part.begin(), part.end());
        for (int t = 0; t < n_threads; ++t) {                    // thread t packed the chunks [c0, c1): shift their run offsets
            const size_t c0 = chunks.size() * (size_t)t / (size_t)n_threads, c1 = chunks.size() * (size_t)(t + 1) / (size_t)n_threads;
            for (size_t ci = c0; ci < c1; ++ci) chunks[ci].run0 += (int64_t)f_covrun.size();
            f_covrun.insert(f_covrun.end(), run_parts[(size_t)t].begin(), run_parts[(size_t)t].end());
        }
    }

    const auto t2 = now();
    // ---- general set: rounds, raw codes --------------------------------------------------
    const int64_t ng = (int64_t)gsel.size();
    const int64_t n_rounds = (ng + TCMI_ROUND - 1) / TCMI_ROUND;
    std::vector<int32_t> h_pos((size_t)ng), h_lseq((size_t)ng);
    std::vector<uint32_t> h_meta((size_t)ng), h_cig((size_t)g_cig + 1), h_seq((size_t)g_seqw + 1);
    std::vector<int64_t> h_rc((size_t)n_rounds + 1), h_rs((size_t)n_rounds + 1);
    int64_t co = 0, so = 0;
    for (int64_t j = 0; j < ng; ++j) {
        const tcmi_reads *r = gsel[(size_t)j].r;
        const int64_t i = gsel[(size_t)j].i;
        if (j % TCMI_ROUND == 0) { h_rc[(size_t)(j / TCMI_ROUND)] = co; h_rs[(size_t)(j / TCMI_ROUND)] = so; }
        const int64_t nc = (int64_t)(r->cigar_off[i + 1] - r->cigar_off[i]);
        const int64_t lq = r->l_qseq[i];
        h_pos[(size_t)j] = (int32_t)(r->pos[i] + gsel[(size_t)j].off);
        h_lseq[(size_t)j] = (int32_t)lq;
        h_meta[(size_t)j] = ((uint32_t)r->flag[i] << 16) | (uint32_t)nc;
        std::memcpy(&h_cig[(size_t)co], r->cigar + r->cigar_off[i], (size_t)nc * 4);
        co += nc;
        const uint8_t *s = r->seq + r->seq_off[i];
        const int64_t nw = (lq + 7) / 8, nb = (lq + 1) / 2;
        uint8_t *dst = reinterpret_cast<uint8_t *>(&h_seq[(size_t)so]);
        for (int64_t b = 0; b < nb; ++b) dst[b] = (uint8_t)((s[b] << 4) | (s[b] >> 4));   // linear nibble order
        if (lq & 1) dst[nb - 1] &= 0x0F;                                                   // pad nibble = 0
        for (int64_t b = nb; b < nw * 4; ++b) dst[b] = 0;
        so += nw;
    }
    h_rc[(size_t)n_rounds] = co;
    h_rs[(size_t)n_rounds] = so;

    const auto t3 = now();
    tcmi_readset *rs = new tcmi_readset();
    rs->uid = next_uid.fetch_add(1);
    rs->n_reads = n_reads_in; rs->n_piled = nf + ng; rs->alg_bytes = alg; rs->max_end = max_end; rs->device = ctx->device;
    rs->f_reads = nf; rs->f_chunks = (int64_t)chunks.size(); rs->f_words = (int64_t)f_seq_n;
    rs->f_events = (int64_t)f_event.size();
    rs->g_reads = ng; rs->n_rounds = n_rounds; rs->n_cigar = g_cig; rs->n_seqw = g_seqw;
    Up up{ctx, rs};
    if (!rc && nf) {
        rc = up((void **)&rs->d_flenoff, f_lenoff.data(), (size_t)nf * 4);
        if (!rc && !f_event.empty()) rc = up((void **)&rs->d_fevent, f_event.data(), f_event.size() * 4);
        if (!rc) rc = up((void **)&rs->d_fseq, f_seq, f_seq_n * 4);
        if (!rc) rc = up((void **)&rs->d_fchunk, chunks.data(), chunks.size() * sizeof(tcmi_fast_chunk));
        if (!rc && !f_covrun.empty()) rc = up((void **)&rs->d_fcovrun, f_covrun.data(), f_covrun.size() * 4);
    }
    if (!rc && ng) {
        rc = up((void **)&rs->d_pos, h_pos.data(), (size_t)ng * 4);
        if (!rc) rc = up((void **)&rs->d_meta, h_meta.data(), (size_t)ng * 4);
        if (!rc) rc = up((void **)&rs->d_lseq, h_lseq.data(), (size_t)ng * 4);
        if (!rc) rc = up((void **)&rs->d_cigar, h_cig.data(), (size_t)g_cig * 4);
        if (!rc) rc = up((void **)&rs->d_seq, h_seq.data(), (size_t)g_seqw * 4);
        if (!rc) rc = up((void **)&rs->d_round_cig, h_rc.data(), (size_t)(n_rounds + 1) * 8);
        if (!rc) rc = up((void **)&rs->d_round_seq, h_rs.data(), (size_t)(n_rounds + 1) * 8);
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = tcmi_fail(ctx, TCMI_E_HIP, "sync after upload failed");
    if (rc) { tcmi_readset_free(ctx, rs); return rc; }
    if (timing)
        std::fprintf(stderr, "[tcmi upload] classify %.1f ms, pack aligned %.1f ms (%d threads), pack general %.1f ms, H2D %.1f ms (%.1f MB)\n",
                     ms(t0, t1), ms(t1, t2), ctx->host_threads, ms(t2, t3), ms(t3, now()), rs->dev_bytes / 1e6);
    *out = rs;
    return TCMI_OK;
}

int tcmi_readset_upload(tcmi_ctx *ctx, const tcmi_reads *r, tcmi_readset **out)
{
    return upload_impl(ctx, &r, 1, 0, out);
}

int tcmi_readset_upload_batch(tcmi_ctx *ctx, const tcmi_reads *const *reads, int32_t n, int64_t stride, tcmi_readset **out)
{
    if (n < 1 || n > 4096 || stride <= 0 || stride % 256) return tcmi_fail(ctx, TCMI_E_ARG, "need 1 <= n <= 4096 and a positive stride that is a multiple of 256");
    return upload_impl(ctx, reads, n, stride, out);
}

int tcmi_readset_info(const tcmi_readset *rs, int64_t *n_reads, int64_t *n_piled, int64_t *alg, int64_t *dev,
                      int64_t *max_end)
{
    if (!rs) return tcmi_fail(nullptr, TCMI_E_ARG, "readset is NULL");
    if (n_reads) *n_reads = rs->n_reads;
    if (n_piled) *n_piled = rs->n_piled;
    if (alg) *alg = rs->alg_bytes;
    if (dev) *dev = rs->dev_bytes;
    if (max_end) *max_end = rs->max_end;
    return TCMI_OK;
}

int tcmi_readset_origin(const tcmi_readset *rs, int32_t *packed_on_device)
{
    if (!rs || !packed_on_device) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    *packed_on_device = rs->packed_on_device;
    return TCMI_OK;
}

int tcmi_readset_sets(const tcmi_readset *rs, int64_t *aligned_reads, int64_t *aligned_chunks, int64_t *general_reads)
{
    if (!rs) return tcmi_fail(nullptr, TCMI_E_ARG, "readset is NULL");
    if (aligned_reads) *aligned_reads = rs->f_reads;
    if (aligned_chunks) *aligned_chunks = rs->f_chunks;
    if (general_reads) *general_reads = rs->g_reads;
    return TCMI_OK;
}

} // extern "C"
