// pipeline.cpp — HOST: many BAMs through the hot path with the GPU and the host cores both busy.
//
// BASELINE.json configs[3] ("independent BAMs, one per stream, no collective") as a native batch
// runner: the calling thread enqueues step i+1 (memset + tally + call + D2H of the call records)
// behind step i on one HIP stream, using `n_slots` workspaces, while `n_walkers` host threads turn
// finished records into consensus sequences: insert candidates -> modal tokens (Events.py:5-82),
// then the sequential walk of Sequences.BuildConsensus (Sequences.py:179-322, consensus_walk.cpp).
// No Python in the loop, so no GIL between the walkers.
#include <cctype>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "tcmi_internal.h"

extern "C" int tcmi_modal_tokens(const tcmi_reads *r, int32_t n_pos, const int64_t *positions, int32_t min_base_quality,
                                 uint32_t flag_filter, int ignore_orphans, int64_t max_depth, char *tokens,
                                 int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens, int32_t *depth_exceeded);

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

int walk_item(tcmi_pipeline *p, int slot, int64_t item_in, int b)
{
    const int64_t item = item_in * p->batch + b;               // output index
    const int64_t shift = (int64_t)b * p->pos_stride;           // BAM b's slice of every record plane
    tcmi_ctx *c = p->slots[(size_t)slot];
    const int64_t L = p->L, ld = c->ws_ld;
    const uint8_t *plain = c->h_rec + shift, *alt = c->h_rec + ld + shift, *flags = c->h_rec + 2 * ld + shift;
    // insert candidates (Events.py:29-36 evaluated by the call kernel) -> accepted inserts
    std::vector<int64_t> cand;
    for (int64_t i = 0; i < L; ++i)
        if (flags[i] & TCMI_F_INSCAND) cand.push_back(i + 1);
    std::vector<int64_t> ins_pos, ins_off(1, 0);
    std::vector<int32_t> ins_shift;
    std::string ins_seq;
    const tcmi_reads *reads = p->host_reads ? p->host_reads[item] : nullptr;
    if (!cand.empty() && !reads)
        return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED,
                         "item %lld has %zu insert candidates but no host reads were given to resolve their tokens",
                         (long long)item, cand.size());
    if (!cand.empty()) {
        std::vector<int64_t> off(cand.size() + 1), cnt(cand.size());
        std::vector<char> toks(1 << 16);
        int32_t deep = 0;
        int rc;
        for (;;) {
            rc = tcmi_modal_tokens(reads, (int32_t)cand.size(), cand.data(), 13, 0x4 | 0x100 | 0x200 | 0x400, 1, 8000,
                                   toks.data(), (int64_t)toks.size(), off.data(), cnt.data(), &deep);
            if (rc == TCMI_E_ARG && toks.size() < ((size_t)1 << 30) && std::strstr(tcmi_last_error(nullptr), "token buffer too small")) {
                toks.resize(toks.size() * 16);
                continue;
            }
            break;
        }
        if (rc) return rc;
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
    const int32_t n_orf = (int32_t)p->orf_start.size();
    std::vector<int64_t> ns((size_t)n_orf), ne((size_t)n_orf);
    int64_t err_pos = 0;
    return tcmi_consensus_walk(plain, alt, flags, L, n_orf, p->orf_start.data(), p->orf_end.data(), p->orf_plus.data(),
                               (int32_t)ins_pos.size(), ins_pos.data(), ins_shift.data(), ins_seq.c_str(), ins_off.data(), 1,
                               p->out + item * p->stride, p->stride, &p->out_len[item], ns.data(), ne.data(), &err_pos);
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
