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
// what the other writers (VCF, corrected GFF) take from the walk besides the consensus
struct WalkExtra {
    std::vector<int64_t> new_start, new_end;        // corrected GFF coordinates per row (of the walk WITH inserts: Outputs.py:95-98)
    std::vector<int64_t> ins_pos, ins_off;          // accepted inserts: 1-based positions, bases at ins_seq[ins_off[k] .. ins_off[k + 1])
    std::string ins_seq;
    std::string cons_noinsert;                      // the second walk (includeINS = False)
};

int walk_records(const uint8_t *plain, const uint8_t *alt, const uint8_t *flags, int64_t L, const tcmi_reads *reads, bool reads_known,
                 const std::vector<int64_t> &orf_start, const std::vector<int64_t> &orf_end, const std::vector<uint8_t> &orf_plus,
                 char *out, int64_t cap, int64_t *out_len, long long item, const PreTokens *pre = nullptr, WalkExtra *extra = nullptr)
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
    int rc = tcmi_consensus_walk(plain, alt, flags, L, n_orf, orf_start.data(), orf_end.data(), orf_plus.data(),
                                 (int32_t)ins_pos.size(), ins_pos.data(), ins_shift.data(), ins_seq.c_str(), ins_off.data(), 1,
                                 out, cap, out_len, ns.data(), ne.data(), &err_pos);
    if (rc || !extra) return rc;
    // the insert-free consensus the VCF is made from: a second walk (Outputs.py:98), whose ORF corrections are discarded
    extra->cons_noinsert.resize((size_t)L + 1);
    int64_t len2 = 0;
    std::vector<int64_t> ns2((size_t)n_orf), ne2((size_t)n_orf);
    rc = tcmi_consensus_walk(plain, alt, flags, L, n_orf, orf_start.data(), orf_end.data(), orf_plus.data(),
                             (int32_t)ins_pos.size(), ins_pos.data(), ins_shift.data(), ins_seq.c_str(), ins_off.data(), 0,
                             &extra->cons_noinsert[0], (int64_t)extra->cons_noinsert.size(), &len2, ns2.data(), ne2.data(), &err_pos);
    if (rc) return rc;
    extra->cons_noinsert.resize((size_t)len2);
    extra->new_start = ns; extra->new_end = ne;
    extra->ins_pos = ins_pos; extra->ins_off = ins_off; extra->ins_seq = ins_seq;
    return TCMI_OK;
}

// ---- the other three outputs of the command line, as text (Outputs.py:13-71, 104-180; Coverage.py:1-16) ----------------------------
void put_int(std::string &o, long long v)
{
    char b[24];
    int n = 0;
    const bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) o.push_back('-');
    while (n) o.push_back(b[--n]);
}

// Coverage.BuildCoverage: "{pos}\t{coverage}\n" for every position of the index
void coverage_tsv_text(const int32_t *cov, int64_t L, std::string &o)
{
    o.clear();
    o.reserve((size_t)L * 12);
    for (int64_t i = 0; i < L; ++i) { put_int(o, i + 1); o.push_back('\t'); put_int(o, cov[i]); o.push_back('\n'); }
}

// Outputs.WriteGFF: the header text verbatim, then one line per row — seqid = the sample's name (TrueConsense.py:240), start / end as
// the walk corrected them, the other columns as the caller folded them (Outputs.py:32-58)
void gff_text(const std::string &head, const std::vector<std::string> &row_cols, const char *name, const std::vector<int64_t> &ns,
              const std::vector<int64_t> &ne, std::string &o)
{
    o = head;
    for (size_t k = 0; k < ns.size(); ++k) {
        const std::string *c = &row_cols[6 * k];            // source, type, score, strand, phase, attributes
        o += name; o.push_back('\t'); o += c[0]; o.push_back('\t'); o += c[1]; o.push_back('\t');
        put_int(o, ns[k]); o.push_back('\t'); put_int(o, ne[k]); o.push_back('\t');
        o += c[2]; o.push_back('\t'); o += c[3]; o.push_back('\t'); o += c[4]; o.push_back('\t'); o += c[5]; o.push_back('\n');
    }
}

// Outputs.py:115-180 as one left-to-right scan (the Python twin: trueconsense_amd/Outputs.py vcf_text).  The quirks are upstream's
// (SURVEY §8-Q10): raw reference characters against the upper-cased insert-free consensus, DP from the FOLLOWING position,
// POS of indel records one less than their index + 1, an insert matched by 0-based index == its 1-based position, Python's
// index -1 (the last element) for a deletion at the very first position.  -> TCMI_E_KEYERROR where upstream raises (a deletion
// run that reaches the end of the consensus: IndexError; a coverage look-up beyond the index: KeyError).
int vcf_text(const std::string &head, const std::string &ref_id, const std::string &ref, const std::string &cons_in, const int32_t *cov,
             int64_t L, int32_t mincov, const WalkExtra &x, std::string &o)
{
    std::string cons = cons_in;
    for (char &c : cons) c = (char)std::toupper((unsigned char)c);
    const int64_t n = (int64_t)ref.size(), nc = (int64_t)cons.size();
    o = head;
    auto coverage = [&](int64_t pos1, long long *out) { if (pos1 < 1 || pos1 > L) return false; *out = cov[pos1 - 1]; return true; };
    size_t next_ins = 0;
    int64_t i = 0;
    while (i < n) {
        if (i >= nc) return tcmi_fail(nullptr, TCMI_E_KEYERROR, "VCF: the consensus ends at %lld, the reference at %lld (IndexError upstream)", (long long)nc, (long long)n);
        const char here = cons[(size_t)i];
        int64_t step = 1;
        long long dp = 0;
        if (here == '-') {
            int64_t j = i;
            while (j < nc && cons[(size_t)j] == '-') ++j;
            if (j >= nc) return tcmi_fail(nullptr, TCMI_E_KEYERROR, "VCF: a deletion runs to the end of the consensus (IndexError upstream)");
            if (!coverage(i + 1, &dp)) return tcmi_fail(nullptr, TCMI_E_KEYERROR, "VCF: no coverage for position %lld (KeyError upstream)", (long long)(i + 1));
            o += ref_id; o.push_back('\t'); put_int(o, i); o += "\t.\t";
            o.push_back(ref[(size_t)((i - 1 + n) % n)]);
            o.append(ref, (size_t)i, (size_t)(std::min(j, n) - i));         // (a Python slice: cut at the reference's end)
            o.push_back('\t'); o.push_back(cons[(size_t)((i - 1 + nc) % nc)]);
            o += "\t.\tPASS\tDP="; put_int(o, dp); o += ";INDEL\n";
            step = j - i;
        } else if (here != ref[(size_t)i]) {
            if (!coverage(std::max<int64_t>(i, 1) + 1, &dp)) return tcmi_fail(nullptr, TCMI_E_KEYERROR, "VCF: no coverage for position %lld (KeyError upstream)", (long long)(std::max<int64_t>(i, 1) + 1));
            o += ref_id; o.push_back('\t'); put_int(o, i + 1); o += "\t.\t"; o.push_back(ref[(size_t)i]); o.push_back('\t'); o.push_back(here);
            o += "\t.\tPASS\tDP="; put_int(o, dp); o.push_back('\n');
        }
        while (next_ins < x.ins_pos.size() && x.ins_pos[next_ins] < i) ++next_ins;
        if (next_ins < x.ins_pos.size() && x.ins_pos[next_ins] == i) {
            if (!coverage(i + 1, &dp)) return tcmi_fail(nullptr, TCMI_E_KEYERROR, "VCF: no coverage for position %lld (KeyError upstream)", (long long)(i + 1));
            if (dp > mincov) {
                o += ref_id; o.push_back('\t'); put_int(o, i); o += "\t.\t"; o.push_back(ref[(size_t)i]); o.push_back('\t'); o.push_back(here);
                o.append(x.ins_seq, (size_t)x.ins_off[next_ins], (size_t)(x.ins_off[next_ins + 1] - x.ins_off[next_ins]));
                o += "\t.\tPASS\tDP="; put_int(o, dp); o += ";INDEL\n";
            }
        }
        i += step;
    }
    return TCMI_OK;
}

int write_file(const char *path, const std::string &text)
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return tcmi_fail(nullptr, TCMI_E_IO, "cannot write %s", path);
    const size_t w = std::fwrite(text.data(), 1, text.size(), fp);
    if (std::fclose(fp) != 0 || w != text.size()) return tcmi_fail(nullptr, TCMI_E_IO, "short write to %s", path);
    return TCMI_OK;
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
    // what the VCF / GFF writers of tcmi_filerunner_run_files need besides the walk (tcmi_filerunner_set_outputs)
    std::string ref_id, ref_seq, vcf_head, gff_head;
    std::vector<std::string> gff_cols;              // per GFF row: source, type, score, strand, phase, attributes
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
    std::vector<int32_t> cov;           // files mode: the coverage column (VCF, coverage TSV)
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
} // extern "C"

namespace {
struct OutFiles { const char *const *fasta, *const *vcf, *const *gff, *const *doc; };     // per item; an entry may be NULL (not wanted)

int filerunner_core(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, int64_t ref_len,
                    int32_t mincov, int include_ambig, int device_decode, char *out_text, int64_t stride, int64_t *out_len,
                    int32_t *status, double *stage_seconds, int64_t *decoded_on, const OutFiles *files, tcmi_bamfile *const *resident = nullptr)
{
    if (!r || n < 0 || (n > 0 && ((!paths && !resident) || !status || (!files && (!out_text || !out_len))))) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    std::vector<const char *> own_paths;
    if (resident) {                                             // files already read (and, with tcmi_bamfile_to_device, already in HBM)
        for (int64_t i = 0; i < n; ++i) {
            if (!resident[i]) return tcmi_fail(nullptr, TCMI_E_ARG, "null file");
            own_paths.push_back(tcmi_bamfile_path(resident[i]));
        }
        paths = own_paths.data();
    }
    std::vector<int64_t> len_store;
    if (!out_len) { len_store.assign((size_t)n, 0); out_len = len_store.data(); }
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
            if (resident) it.file = resident[i];
            else if (device_decode) {
                // (several readers: two copy threads a file — with one, a reader needs 1.0 - 1.4 ms for a 7.7 MB file and three readers run
                //  close to the GPU's 0.42 ms per file: on a slower host the file leg fell to 55 - 66 M positions/s where two threads gave
                //  63 - 69 M, three turns each, profiles/r06l_read_threads_ab.log)
                it.rc = tcmi_bamfile_read_threads(paths[i], r->n_readers > 1 ? 2 : 0, &it.file);
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
                const bool want_cov = files && ((files->vcf && files->vcf[i]) || (files->doc && files->doc[i]));
                const uint8_t *pl = nullptr, *al = nullptr, *fl = nullptr;
                const int32_t *planes = nullptr;
                int64_t ld = 0, L = 0;
                bool stepped = false;                            // the device path queues decode, pack, tally and call back to back: ONE wait per file
                if (it.file) {
                    rc = tcmi_bamfile_step(ctx, it.file, ref_len, mincov, include_ambig, &rs, &L, &pl, &al, &fl, want_cov ? &planes : nullptr, &ld);
                    it.on_device = rc == TCMI_OK;
                    stepped = rc == TCMI_OK;
                    if (rc == TCMI_E_NOMEM) rc = TCMI_E_UNSUPPORTED;     // (a file whose decode does not fit the device's memory: the host reader streams it)
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
                    if (!stepped) {
                        int64_t max_end = 0;
                        tcmi_readset_info(rs, nullptr, nullptr, nullptr, nullptr, &max_end);
                        L = std::max<int64_t>({ref_len, max_end, 1});
                        rc = tcmi_step(ctx, rs, L, mincov, include_ambig, &pl, &al, &fl, want_cov ? &planes : nullptr, &ld);
                    }
                    if (!rc) {
                        it.L = L;
                        if (want_cov) it.cov.assign(planes + (size_t)TCMI_COV * ld, planes + (size_t)TCMI_COV * ld + L);
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
                                // (anything the device path declines — megabytes of long insertions, an overlap it cannot resolve —
                                // is left to the walker's host sweep, which words the refusal if it is one)
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
            if (it.file && !resident) tcmi_bamfile_free(it.file);
            it.file = nullptr;
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
                const char *nm = names && names[i] ? names[i] : "sample";
                std::string own;                                 // files mode: the FASTA text lives here
                int64_t room = stride;
                char *dst = out_text ? out_text + i * stride : nullptr;
                if (files) { room = L + (int64_t)std::strlen(nm) + (1 << 16); own.resize((size_t)room); dst = &own[0]; }
                const int head = std::snprintf(dst, (size_t)room, ">%s mincov=%d\n", nm, (int)mincov);
                int64_t len = 0;
                WalkExtra extra;
                const bool want_extra = files && ((files->vcf && files->vcf[i]) || (files->gff && files->gff[i]));
                if (!rc && (head < 0 || head + 2 >= room)) rc = tcmi_fail(nullptr, TCMI_E_ARG, "output stride too small");
                if (!rc) rc = walk_records(pl, al, fl, L, cand ? &reads : nullptr, true, r->orf_start, r->orf_end, r->orf_plus, dst + head,
                                           room - head - 1, &len, (long long)i, &it.pre, want_extra ? &extra : nullptr);
                if (!rc) { dst[head + len] = '\n'; out_len[i] = head + len + 1; }
                if (!rc && files) {
                    // the four writers of Outputs.WriteOutputs / Coverage.BuildCoverage, in the reference's order: coverage TSV
                    // (TrueConsense.py:243-245), corrected GFF, VCF, FASTA
                    std::string text;
                    if (files->doc && files->doc[i]) { coverage_tsv_text(it.cov.data(), L, text); rc = write_file(files->doc[i], text); }
                    if (!rc && files->gff && files->gff[i]) { gff_text(r->gff_head, r->gff_cols, nm, extra.new_start, extra.new_end, text); rc = write_file(files->gff[i], text); }
                    if (!rc && files->vcf && files->vcf[i]) {
                        rc = vcf_text(r->vcf_head, r->ref_id, r->ref_seq, extra.cons_noinsert, it.cov.data(), L, mincov, extra, text);
                        if (!rc) rc = write_file(files->vcf[i], text);
                    }
                    if (!rc && files->fasta && files->fasta[i]) { own.resize((size_t)out_len[i]); rc = write_file(files->fasta[i], own); }
                }
                if (rc) { it.rc = rc; it.err = tcmi_last_error(nullptr); }
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
} // namespace

extern "C" {

int tcmi_filerunner_run(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, int64_t ref_len,
                        int32_t mincov, int include_ambig, int device_decode, char *out_text, int64_t stride, int64_t *out_len,
                        int32_t *status, double *stage_seconds, int64_t *decoded_on)
{
    return filerunner_core(r, n, paths, names, ref_len, mincov, include_ambig, device_decode, out_text, stride, out_len, status, stage_seconds,
                           decoded_on, nullptr);
}

// ... of files that were read before (tcmi_bamfile_read) and stay the caller's: the read stage has nothing to do.  With their bytes
// in HBM (tcmi_bamfile_to_device) the GPU stage starts from device memory — no PCIe copy inside the run.
int tcmi_filerunner_run_resident(tcmi_filerunner *r, int64_t n, tcmi_bamfile *const *files, const char *const *names, int64_t ref_len,
                                 int32_t mincov, int include_ambig, char *out_text, int64_t stride, int64_t *out_len, int32_t *status,
                                 double *stage_seconds, int64_t *decoded_on)
{
    if (!files && n > 0) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    return filerunner_core(r, n, nullptr, names, ref_len, mincov, include_ambig, 1, out_text, stride, out_len, status, stage_seconds, decoded_on,
                           nullptr, files);
}

// What the VCF and GFF writers need besides a sample's walk: the reference (first FASTA record: id and sequence, Outputs.py:108-113),
// the complete VCF header text (Outputs.py:115-127: date, command line and reference path are the caller's), the GFF header text
// and, per GFF row, its columns other than seqid / start / end: source, type, score, strand, phase, attributes (6 strings per row,
// the attributes folded as Outputs.py:32-58 folds them).
int tcmi_filerunner_set_outputs(tcmi_filerunner *r, const char *ref_id, const char *ref_seq, const char *vcf_head, const char *gff_head,
                                int32_t n_rows, const char *const *row_cols)
{
    if (!r || !ref_id || !ref_seq || !vcf_head || !gff_head || n_rows < 0 || (n_rows > 0 && !row_cols)) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    if ((size_t)n_rows != r->orf_start.size()) return tcmi_fail(nullptr, TCMI_E_ARG, "%d GFF rows, %zu ORFs set", (int)n_rows, r->orf_start.size());
    r->ref_id = ref_id; r->ref_seq = ref_seq; r->vcf_head = vcf_head; r->gff_head = gff_head;
    r->gff_cols.clear();
    for (int32_t k = 0; k < 6 * n_rows; ++k) r->gff_cols.emplace_back(row_cols[k] ? row_cols[k] : "");
    return TCMI_OK;
}

// BAM files -> the command line's four output FILES per sample (TrueConsense.py:212-264 with -o, -vcf, -ogff, -doc), same stages
// and overlap as tcmi_filerunner_run; the walker threads also write the VCF, the corrected GFF and the coverage TSV (no Python
// per sample).  Any of vcf / gff / doc, or single entries of them, may be NULL.  tcmi_filerunner_set_outputs first.
int tcmi_filerunner_run_files(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, const char *const *fasta,
                              const char *const *vcf, const char *const *gff, const char *const *doc, int64_t ref_len, int32_t mincov,
                              int include_ambig, int device_decode, int32_t *status, double *stage_seconds, int64_t *decoded_on)
{
    if (!r || !fasta) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    if ((vcf || gff) && r->gff_cols.size() != 6 * r->orf_start.size()) return tcmi_fail(nullptr, TCMI_E_ARG, "tcmi_filerunner_set_outputs first");
    const OutFiles f = {fasta, vcf, gff, doc};
    return filerunner_core(r, n, paths, names, ref_len, mincov, include_ambig, device_decode, nullptr, 0, nullptr, status, stage_seconds, decoded_on, &f);
}

} // extern "C"
