// pipeline.cpp — HOST: many BAMs through the hot path with the GPU and the host cores both busy.
//
// BASELINE.json configs[3] ("independent BAMs, one per stream, no collective") as a native batch
// runner: the calling thread enqueues step i+1 (memset + tally + call + D2H of the call records)
// behind step i on one HIP stream, using `n_slots` workspaces, while `n_walkers` host threads turn
// finished records into consensus sequences: insert candidates -> modal tokens (Events.py:5-82),
// then the sequential walk of Sequences.BuildConsensus (Sequences.py:179-322, consensus_walk.cpp).
// No Python in the loop, so no GIL between the walkers.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "tcmi_internal.h"


struct tcmi_pipeline {
    int device = 0;
    std::vector<tcmi_ctx *> slots;
    std::vector<int> slot_busy;                     // 1 from step_begin until the walk has consumed the records
    std::vector<int64_t> orf_start, orf_end;
    std::vector<uint8_t> orf_plus;
    // walker pool
    struct Job { int slot; int64_t item; int b; };
    std::vector<int> slot_jobs;                     // walk jobs still open per slot (a batched item has several)
    int batch = 1;
    int64_t pos_stride = 0;
    std::deque<Job> jobs;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> workers;
    bool stop = false;
    int64_t jobs_open = 0;
    // the run in progress
    int64_t L = 0;
    const tcmi_reads *const *host_reads = nullptr;
    char *out = nullptr;
    int64_t stride = 0;
    int64_t *out_len = nullptr;
    int32_t *status = nullptr;
    std::string err;
};

namespace {

// Events.py:75-80 on the modal token: re.search("(\d)([a-zA-Z]+)")
bool parse_token(const char *t, int64_t n, int *size_digit, const char **bases, int64_t *n_bases)
{
    for (int64_t i = 0; i + 1 < n; ++i)
        if (std::isdigit((unsigned char)t[i]) && std::isalpha((unsigned char)t[i + 1])) {
            int64_t j = i + 1;
            while (j < n && std::isalpha((unsigned char)t[j])) ++j;
            *size_digit = t[i] - '0';
            *bases = t + i + 1;
            *n_bases = j - i - 1;
            return true;
        }
    return false;
}

// call records of one BAM -> consensus: insert candidates -> modal tokens (Events.py:5-82), then the sequential walk
// modal tokens of the candidate columns resolved beforehand (on the device: tcmi_readset_modal_tokens)
struct PreTokens { std::vector<int64_t> cand, off, cnt; std::vector<char> toks; bool valid = false; };

int walk_records(const uint8_t *plain, const uint8_t *alt, const uint8_t *flags, int64_t L, const tcmi_reads *reads, bool reads_known,
                 const std::vector<int64_t> &orf_start, const std::vector<int64_t> &orf_end, const std::vector<uint8_t> &orf_plus,
                 char *out, int64_t cap, int64_t *out_len, long long item, const PreTokens *pre = nullptr)
{
    // insert candidates (Events.py:29-36 evaluated by the call kernel) -> accepted inserts
    std::vector<int64_t> cand;
    for (int64_t i = 0; i < L; ++i)
        if (flags[i] & TCMI_F_INSCAND) cand.push_back(i + 1);
    std::vector<int64_t> ins_pos, ins_off(1, 0);
    std::vector<int32_t> ins_shift;
    std::string ins_seq;
    const bool use_pre = pre && pre->valid && pre->cand == cand;
    if (!cand.empty() && !reads_known && !use_pre)
        return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED,
                         "item %lld has %zu insert candidates but no host reads were given to resolve their tokens", item, cand.size());
    if (!cand.empty()) {
        std::vector<int64_t> off(cand.size() + 1), cnt(cand.size());
        std::vector<char> toks(1 << 16);
        int32_t tok_status = 0;
        int rc = TCMI_OK;
        if (use_pre) { off = pre->off; cnt = pre->cnt; toks = pre->toks; }
        else for (;;) {
            // pysam's defaults for AlignmentFile.pileup() (Events.py:66 passes none): SURVEY §8-Q8
            rc = tcmi_modal_tokens(reads, (int32_t)cand.size(), cand.data(), 13, 0x4 | 0x100 | 0x200 | 0x400, 1, 8000, 1,
                                   toks.data(), (int64_t)toks.size(), off.data(), cnt.data(), &tok_status);
            if (rc == TCMI_E_ARG && toks.size() < ((size_t)1 << 30) && std::strstr(tcmi_last_error(nullptr), "token buffer too small")) {
                toks.resize(toks.size() * 16);
                continue;
            }
            break;
        }
        if (rc) return rc;
        if (tok_status & TCMI_TOKENS_OVERLAP_UNKNOWN)
            return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED,
                             "item %lld: overlapping mates with a deletion on an insert-candidate column: pysam's overlap quality tweak there is not modelled", item);
        for (size_t k = 0; k < cand.size(); ++k) {
            if (cnt[k] == 0) continue;
            int digit;
            const char *b;
            int64_t nb;
            if (!parse_token(toks.data() + off[k], off[k + 1] - off[k], &digit, &b, &nb)) continue;
            ins_pos.push_back(cand[k]);
            ins_shift.push_back(digit);
            ins_seq.append(b, (size_t)nb);
            ins_off.push_back((int64_t)ins_seq.size());
        }
    }
    const int32_t n_orf = (int32_t)orf_start.size();
    std::vector<int64_t> ns((size_t)n_orf), ne((size_t)n_orf);
    int64_t err_pos = 0;
    return tcmi_consensus_walk(plain, alt, flags, L, n_orf, orf_start.data(), orf_end.data(), orf_plus.data(),
                               (int32_t)ins_pos.size(), ins_pos.data(), ins_shift.data(), ins_seq.c_str(), ins_off.data(), 1,
                               out, cap, out_len, ns.data(), ne.data(), &err_pos);
}

int walk_item(tcmi_pipeline *p, int slot, int64_t item_in, int b)
{
    const int64_t item = item_in * p->batch + b;               // output index
    const int64_t shift = (int64_t)b * p->pos_stride;           // BAM b's slice of every record plane
    tcmi_ctx *c = p->slots[(size_t)slot];
    const int64_t L = p->L, ld = c->ws_ld;
    const tcmi_reads *reads = p->host_reads ? p->host_reads[item] : nullptr;
    return walk_records(c->h_rec + shift, c->h_rec + ld + shift, c->h_rec + 2 * ld + shift, L, reads, reads != nullptr, p->orf_start,
                        p->orf_end, p->orf_plus, p->out + item * p->stride, p->stride, &p->out_len[item], (long long)item);
}

void worker_main(tcmi_pipeline *p)
{
    for (;;) {
        tcmi_pipeline::Job job;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->stop || !p->jobs.empty(); });
            if (p->jobs.empty()) return;
            job = p->jobs.front();
            p->jobs.pop_front();
        }
        const int rc = walk_item(p, job.slot, job.item, job.b);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (p->status) p->status[job.item * p->batch + job.b] = rc;
            if (--p->slot_jobs[(size_t)job.slot] == 0) p->slot_busy[(size_t)job.slot] = 0;
            --p->jobs_open;
        }
        p->cv_done.notify_all();
    }
}

} // namespace

extern "C" {

int tcmi_pipeline_create(int device, int n_slots, int n_walkers, tcmi_pipeline **out)
{
    if (!out || n_slots < 1 || n_slots > 64 || n_walkers < 1 || n_walkers > 256)
        return tcmi_fail(nullptr, TCMI_E_ARG, "need 1..64 slots and 1..256 walkers");
    *out = nullptr;
    tcmi_pipeline *p = new tcmi_pipeline();
    p->device = device;
    for (int s = 0; s < n_slots; ++s) {
        tcmi_ctx *c = nullptr;
        // one stream for all workspaces: slot 0 owns it
        const int rc = s == 0 ? tcmi_ctx_create(device, &c) : tcmi_ctx_create_on_stream(device, tcmi_ctx_stream(p->slots[0]), &c);
        if (rc) {
            for (size_t k = p->slots.size(); k-- > 0;) tcmi_ctx_destroy(p->slots[k]);
            delete p;
            return rc;
        }
        p->slots.push_back(c);
    }
    p->slot_busy.assign((size_t)n_slots, 0);
    p->slot_jobs.assign((size_t)n_slots, 0);
    for (int w = 0; w < n_walkers; ++w) p->workers.emplace_back(worker_main, p);
    *out = p;
    return TCMI_OK;
}

int tcmi_pipeline_destroy(tcmi_pipeline *p)
{
    if (!p) return TCMI_OK;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_job.notify_all();
    for (auto &t : p->workers) t.join();
    for (size_t k = p->slots.size(); k-- > 0;) tcmi_ctx_destroy(p->slots[k]);    // slot 0 owns the stream: last
    delete p;
    return TCMI_OK;
}

int tcmi_pipeline_set_orfs(tcmi_pipeline *p, int32_t n_orf, const int64_t *start, const int64_t *end, const uint8_t *is_plus)
{
    if (!p || n_orf < 0 || (n_orf > 0 && (!start || !end || !is_plus))) return tcmi_fail(nullptr, TCMI_E_ARG, "bad argument");
    p->orf_start.assign(start, start + n_orf);
    p->orf_end.assign(end, end + n_orf);
    p->orf_plus.assign(is_plus, is_plus + n_orf);
    return TCMI_OK;
}

tcmi_ctx *tcmi_pipeline_ctx(tcmi_pipeline *p, int slot)
{
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? p->slots[(size_t)slot] : nullptr;
}

int tcmi_pipeline_run_batched(tcmi_pipeline *p, int64_t n_items, const tcmi_readset *const *readsets, int32_t batch,
                              int64_t pos_stride, const tcmi_reads *const *host_reads, int64_t L, int32_t mincov,
                              int include_ambig, char *out_cons, int64_t stride, int64_t *out_len, int32_t *status)
{
    if (!p || n_items < 0 || (n_items > 0 && (!readsets || !out_cons || !out_len || !status)) || stride < L + 1)
        return tcmi_fail(nullptr, TCMI_E_ARG, "bad argument (stride must be >= L + 1 + inserted bases)");
    if (batch < 1 || (batch > 1 && (pos_stride < L || pos_stride % 256)))
        return tcmi_fail(nullptr, TCMI_E_ARG, "bad batch / position stride");
    p->batch = batch;
    p->pos_stride = batch > 1 ? pos_stride : 0;
    const int64_t L_gpu = batch > 1 ? (int64_t)batch * pos_stride : L;   // positions one step covers
    p->L = L; p->host_reads = host_reads; p->out = out_cons; p->stride = stride; p->out_len = out_len; p->status = status;
    const int n_slots = (int)p->slots.size();
    int64_t submitted = 0;
    int first_err = TCMI_OK;
    // Ride-along call (option defer_call of slot 0): the call of step k is carried by the tally launch of step k + 1,
    // so a step is ONE launch; the last step of the queue (or one nobody followed) launches its call on its own.
    const bool defer = p->slots[0]->defer_call != 0;
    int pending_slot = -1;
    for (int64_t waited = 0; waited < n_items; ++waited) {
        // keep the stream fed: queue every step whose workspace is free
        while (submitted < n_items && submitted - waited < n_slots) {
            const int slot = (int)(submitted % n_slots);
            {
                std::unique_lock<std::mutex> lk(p->mu);
                if (p->slot_busy[(size_t)slot]) {
                    if (submitted > waited) break;               // something is already in flight: come back later
                    p->cv_done.wait(lk, [&] { return !p->slot_busy[(size_t)slot]; });
                }
                p->slot_busy[(size_t)slot] = 1;
            }
            const int rc = defer ? tcmi_step_begin_deferred(p->slots[(size_t)slot], readsets[submitted], L_gpu, mincov, include_ambig,
                                                            pending_slot >= 0 ? p->slots[(size_t)pending_slot] : nullptr)
                                 : tcmi_step_begin(p->slots[(size_t)slot], readsets[submitted], L_gpu, mincov, include_ambig, 0);
            if (defer && !rc) pending_slot = slot;
            if (rc) {
                for (int b = 0; b < batch; ++b) { status[submitted * batch + b] = rc; out_len[submitted * batch + b] = 0; }
                if (!first_err) { first_err = rc; p->err = tcmi_last_error(p->slots[(size_t)slot]); }
                std::lock_guard<std::mutex> lk(p->mu);
                p->slot_busy[(size_t)slot] = 2;                  // nothing queued for this item
            }
            ++submitted;
        }
        const int slot = (int)(waited % n_slots);
        bool queued;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            queued = p->slot_busy[(size_t)slot] == 1;
            if (!queued) p->slot_busy[(size_t)slot] = 0;
        }
        if (!queued) continue;
        if (slot == pending_slot) pending_slot = -1;             // nobody followed it: tcmi_step_end launches its call
        const int rc = tcmi_step_end(p->slots[(size_t)slot], nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc) {
            for (int b = 0; b < batch; ++b) { status[waited * batch + b] = rc; out_len[waited * batch + b] = 0; }
            if (!first_err) { first_err = rc; p->err = tcmi_last_error(p->slots[(size_t)slot]); }
            std::lock_guard<std::mutex> lk(p->mu);
            p->slot_busy[(size_t)slot] = 0;
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->slot_jobs[(size_t)slot] = batch;
            for (int b = 0; b < batch; ++b) p->jobs.push_back({slot, waited, b});
            p->jobs_open += batch;
        }
        p->cv_job.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return p->jobs_open == 0; });
    }
    if (first_err) return tcmi_fail(nullptr, first_err, "%s", p->err.c_str());
    for (int64_t i = 0; i < n_items * batch; ++i)
        if (status[i]) return tcmi_fail(nullptr, status[i], "item %lld: the consensus walk failed (status %d)", (long long)i, status[i]);
    return TCMI_OK;
}

int tcmi_pipeline_run(tcmi_pipeline *p, int64_t n_items, const tcmi_readset *const *readsets,
                      const tcmi_reads *const *host_reads, int64_t L, int32_t mincov, int include_ambig,
                      char *out_cons, int64_t stride, int64_t *out_len, int32_t *status)
{
    return tcmi_pipeline_run_batched(p, n_items, readsets, 1, 0, host_reads, L, mincov, include_ambig, out_cons, stride,
                                     out_len, status);
}

} // extern "C"


// ---- BAM FILES -> FASTA text: the whole command line of the reference (TrueConsense.py:212-264) for many inputs -------------
// Three stages on their own threads, consecutive BAMs overlapping:
//   read   HOST: file bytes into pinned memory + BGZF block table + BAM header (tcmi_bamfile_read)
//   gpu    one thread per context (stream + device arena): H2D of the compressed bytes, HIP inflate / record chain / pack,
//          tally + call, call records into the item's buffer; a file the device decoder declines is decoded by the host
//          reader (tcmi_bam_load) and packed from its flat arrays
//   walk   HOST: insert tokens when a candidate exists (decodes the BAM on the host then: the decoded reads never left the
//          device), the sequential consensus walk, ">name mincov=N\n<consensus>\n" (Outputs.py:182-183)
struct tcmi_filerunner {
    int device = 0, n_readers = 2, n_walkers = 2, host_threads = 8;
    std::vector<tcmi_ctx *> ctxs;
    std::vector<int64_t> orf_start, orf_end;
    std::vector<uint8_t> orf_plus;
};

namespace {

struct FileItem {
    tcmi_bamfile *file = nullptr;       // after the read stage
    std::vector<uint8_t> rec;           // plain | alt | flags, L each, after the gpu stage
    int64_t L = 0;
    int state = 0;                      // 0 nothing, 1 read, 2 records ready, 3 done
    int rc = TCMI_OK;
    std::string err;
    bool on_device = false;
    PreTokens pre;                      // insert tokens resolved on the device while the decoded stream was still resident
};

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace

extern "C" {

int tcmi_filerunner_create(int device, int n_readers, int n_gpu, int n_walkers, int host_decode_threads, tcmi_filerunner **out)
{
    if (!out || n_readers < 1 || n_readers > 64 || n_gpu < 1 || n_gpu > 16 || n_walkers < 1 || n_walkers > 64)
        return tcmi_fail(nullptr, TCMI_E_ARG, "need 1..64 readers, 1..16 gpu contexts, 1..64 walkers");
    *out = nullptr;
    tcmi_filerunner *r = new tcmi_filerunner();
    r->device = device; r->n_readers = n_readers; r->n_walkers = n_walkers; r->host_threads = host_decode_threads > 0 ? host_decode_threads : 8;
    for (int k = 0; k < n_gpu; ++k) {
        tcmi_ctx *c = nullptr;
        const int rc = tcmi_ctx_create(device, &c);
        if (rc) {
            for (auto *x : r->ctxs) tcmi_ctx_destroy(x);
            delete r;
            return rc;
        }
        r->ctxs.push_back(c);
    }
    *out = r;
    return TCMI_OK;
}

int tcmi_filerunner_destroy(tcmi_filerunner *r)
{
    if (!r) return TCMI_OK;
    for (auto *c : r->ctxs) tcmi_ctx_destroy(c);
    delete r;
    return TCMI_OK;
}

int tcmi_filerunner_set_orfs(tcmi_filerunner *r, int32_t n_orf, const int64_t *start, const int64_t *end, const uint8_t *is_plus)
{
    if (!r || n_orf < 0 || (n_orf > 0 && (!start || !end || !is_plus))) return tcmi_fail(nullptr, TCMI_E_ARG, "bad argument");
    r->orf_start.assign(start, start + n_orf);
    r->orf_end.assign(end, end + n_orf);
    r->orf_plus.assign(is_plus, is_plus + n_orf);
    return TCMI_OK;
}

tcmi_ctx *tcmi_filerunner_ctx(tcmi_filerunner *r, int k) { return (r && k >= 0 && k < (int)r->ctxs.size()) ? r->ctxs[(size_t)k] : nullptr; }

// out_text: n * stride bytes, text i at out_text + i * stride, out_len[i] bytes (stride >= ref_len + inserted bases + name + 32)
// stage_seconds[4]: busy seconds summed over the items: read, upload (decode + pack), step, walk
// decoded_on[2]: items decoded on the device / by the host reader
int tcmi_filerunner_run(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, int64_t ref_len,
                        int32_t mincov, int include_ambig, int device_decode, char *out_text, int64_t stride, int64_t *out_len,
                        int32_t *status, double *stage_seconds, int64_t *decoded_on)
{
    if (!r || n < 0 || (n > 0 && (!paths || !out_text || !out_len || !status))) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    std::vector<FileItem> items((size_t)n);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int64_t> next_read{0}, next_gpu{0}, next_walk{0};
    int64_t n_done = 0;                                         // (under mu) items finished: bounds how far the readers run ahead
    const int64_t window = (int64_t)r->n_readers + (int64_t)r->ctxs.size() + r->n_walkers + 2;
    double sec[4] = {0, 0, 0, 0};
    int64_t on[2] = {0, 0};

    auto reader = [&]() {
        for (;;) {
            const int64_t i = next_read.fetch_add(1);
            if (i >= n) return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return i < n_done + window; });
            }
            const auto t0 = std::chrono::steady_clock::now();
            FileItem &it = items[(size_t)i];
            if (device_decode) {
                it.rc = tcmi_bamfile_read_threads(paths[i], r->n_readers > 1 ? 1 : 0, &it.file);
                if (it.rc) it.err = tcmi_last_error(nullptr);
            }
            const double dt = seconds_since(t0);
            {
                std::lock_guard<std::mutex> lk(mu);
                it.state = 1;
                sec[0] += dt;
            }
            cv.notify_all();
        }
    };
    auto gpu = [&](tcmi_ctx *ctx) {
        for (;;) {
            const int64_t i = next_gpu.fetch_add(1);
            if (i >= n) return;
            FileItem &it = items[(size_t)i];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return it.state >= 1; });
            }
            double t_up = 0, t_step = 0;
            if (!it.rc) {
                const auto t0 = std::chrono::steady_clock::now();
                tcmi_readset *rs = nullptr;
                tcmi_bam *hb = nullptr;
                int rc = TCMI_E_UNSUPPORTED;
                if (it.file) {
                    rc = tcmi_readset_from_bamfile(ctx, it.file, &rs, nullptr);
                    it.on_device = rc == TCMI_OK;
                }
                std::string host_err;
                if (rc == TCMI_E_UNSUPPORTED) {                  // the host reader takes it
                    rc = tcmi_bam_load(paths[i], r->host_threads, &hb);
                    if (rc) host_err = tcmi_last_error(nullptr); // (its words: it failed without a context)
                    if (!rc) {
                        tcmi_reads reads;
                        tcmi_bam_reads(hb, &reads);
                        rc = tcmi_readset_upload(ctx, &reads, &rs);
                    }
                }
                t_up = seconds_since(t0);
                const auto t1 = std::chrono::steady_clock::now();
                if (!rc) {
                    int64_t max_end = 0;
                    tcmi_readset_info(rs, nullptr, nullptr, nullptr, nullptr, &max_end);
                    const int64_t L = std::max<int64_t>({ref_len, max_end, 1});
                    const uint8_t *pl, *al, *fl;
                    int64_t ld = 0;
                    rc = tcmi_step(ctx, rs, L, mincov, include_ambig, &pl, &al, &fl, nullptr, &ld);
                    if (!rc) {
                        it.L = L;
                        it.rec.resize((size_t)L * 3);
                        std::memcpy(it.rec.data(), pl, (size_t)L);
                        std::memcpy(it.rec.data() + L, al, (size_t)L);
                        std::memcpy(it.rec.data() + 2 * L, fl, (size_t)L);
                        // insert candidates of a BAM the device decoded: their tokens are voted on now, from the stream that is
                        // still resident in this context's arena (Events.py:47-82 without the reads ever reaching the host)
                        if (it.on_device) {
                            for (int64_t k = 0; k < L; ++k)
                                if (fl[k] & TCMI_F_INSCAND) it.pre.cand.push_back(k + 1);
                            if (!it.pre.cand.empty()) {
                                const size_t nc = it.pre.cand.size();
                                it.pre.off.assign(nc + 1, 0);
                                it.pre.cnt.assign(nc, 0);
                                it.pre.toks.assign(1 << 16, 0);
                                int32_t st = 0;
                                int rt;
                                for (;;) {
                                    rt = tcmi_readset_modal_tokens(ctx, rs, (int32_t)nc, it.pre.cand.data(), 13, 0x4 | 0x100 | 0x200 | 0x400, 1, 8000, 1,
                                                                   it.pre.toks.data(), (int64_t)it.pre.toks.size(), it.pre.off.data(), it.pre.cnt.data(), &st);
                                    if (rt == TCMI_E_ARG && it.pre.toks.size() < ((size_t)1 << 30) && std::strstr(tcmi_last_error(ctx), "token buffer too small")) {
                                        it.pre.toks.resize(it.pre.toks.size() * 16);
                                        continue;
                                    }
                                    break;
                                }
                                // (anything the device path declines — a long insertion, overlapping mates with a deletion on the
                                // column — is left to the walker's host sweep, which words the refusal if it is one)
                                it.pre.valid = rt == TCMI_OK && !(st & TCMI_TOKENS_OVERLAP_UNKNOWN);
                            }
                        }
                    }
                }
                t_step = seconds_since(t1);
                if (rc) { it.rc = rc; it.err = host_err.empty() ? tcmi_last_error(ctx) : host_err; }
                if (rs) tcmi_readset_free(ctx, rs);
                if (hb) tcmi_bam_free(hb);
            }
            if (it.file) { tcmi_bamfile_free(it.file); it.file = nullptr; }
            {
                std::lock_guard<std::mutex> lk(mu);
                it.state = 2;
                sec[1] += t_up; sec[2] += t_step;
                if (!it.rc) on[it.on_device ? 0 : 1] += 1;
            }
            cv.notify_all();
        }
    };
    auto walker = [&]() {
        for (;;) {
            const int64_t i = next_walk.fetch_add(1);
            if (i >= n) return;
            FileItem &it = items[(size_t)i];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return it.state >= 2; });
            }
            const auto t0 = std::chrono::steady_clock::now();
            out_len[i] = 0;
            if (!it.rc) {
                const int64_t L = it.L;
                const uint8_t *pl = it.rec.data(), *al = pl + L, *fl = pl + 2 * L;
                bool cand = false;
                for (int64_t k = 0; k < L && !cand; ++k) cand = (fl[k] & TCMI_F_INSCAND) != 0;
                if (it.pre.valid) cand = false;                  // already resolved on the device
                tcmi_bam *hb = nullptr;
                tcmi_reads reads;
                int rc = TCMI_OK;
                if (cand) {                                      // insert tokens need the reads on the host
                    rc = tcmi_bam_load(paths[i], r->host_threads, &hb);
                    if (!rc) tcmi_bam_reads(hb, &reads);
                }
                char *dst = out_text + i * stride;
                const int head = std::snprintf(dst, (size_t)stride, ">%s mincov=%d\n", names && names[i] ? names[i] : "sample", (int)mincov);
                int64_t len = 0;
                if (!rc && (head < 0 || head + 2 >= stride)) rc = tcmi_fail(nullptr, TCMI_E_ARG, "output stride too small");
                if (!rc) rc = walk_records(pl, al, fl, L, cand ? &reads : nullptr, true, r->orf_start, r->orf_end, r->orf_plus, dst + head,
                                           stride - head - 1, &len, (long long)i, &it.pre);
                if (!rc) { dst[head + len] = '\n'; out_len[i] = head + len + 1; }
                else { it.rc = rc; it.err = tcmi_last_error(nullptr); }
                if (hb) tcmi_bam_free(hb);
                std::vector<uint8_t>().swap(it.rec);
            }
            status[i] = it.rc;
            const double dt = seconds_since(t0);
            {
                std::lock_guard<std::mutex> lk(mu);
                it.state = 3;
                ++n_done;
                sec[3] += dt;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < r->n_readers; ++k) th.emplace_back(reader);
    for (auto *c : r->ctxs) th.emplace_back(gpu, c);
    for (int k = 0; k < r->n_walkers; ++k) th.emplace_back(walker);
    for (auto &t : th) t.join();
    if (stage_seconds) for (int k = 0; k < 4; ++k) stage_seconds[k] = sec[k];
    if (decoded_on) { decoded_on[0] = on[0]; decoded_on[1] = on[1]; }
    for (int64_t i = 0; i < n; ++i)
        if (items[(size_t)i].rc) return tcmi_fail(nullptr, items[(size_t)i].rc, "%s: %s", paths[i], items[(size_t)i].err.c_str());
    return TCMI_OK;
}

} // extern "C"
