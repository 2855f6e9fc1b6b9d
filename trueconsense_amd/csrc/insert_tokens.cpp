// insert_tokens.cpp — HOST: the token multiset behind Events.ExtractInserts
// (TrueConsense/Events.py:47-82), for all candidate positions in one sweep over the reads.
//
// The reference opens a fresh region pileup per candidate with pysam's DEFAULT arguments
// (Events.py:66): stepper "samtools" (drop UNMAP|SECONDARY|QCFAIL|DUP and orphans),
// min_base_quality 13, no reference attached; it upper-cases every token and takes the modal
// one with first-seen tie-break (collections.Counter.most_common, Events.py:71-74).
// Token text follows pysam's PileupColumn.get_query_sequences(add_indels=True).
// The two remaining default arguments are modelled after htslib 1.21 (the one pysam 0.23.3 bundles; PARITY UNPINNED —
// neither is installed here):
//   max_depth = 8000     bam_plp_push drops a read that starts on the iterator's current column while the buffer holds
//                        maxcnt nodes (its sentinel included).  A region pileup is fed only the reads that overlap the
//                        column, all of them still alive: the 8 001st and later reads that share their start with the read
//                        before them are dropped.
//   ignore_overlaps      overlap_push / tweak_overlap_quality: of two properly paired mates that both cover the column, the
//                        one pushed first keeps the base (quality = sum, at most 200, when the bases agree; 0.8 x the higher
//                        one when they differ) and the other's quality becomes 0, so its token fails min_base_quality.
//                        htslib tweaks every reference position where BOTH mates have a matched base, in both quality arrays,
//                        when the second mate arrives.  For a column that is evaluated where it matters: on the column
//                        itself when both mates have a base there; and — a mate whose token is a deletion / ref-skip is
//                        tested on the quality of its NEXT query base (htslib's qpos) — on that base's reference position
//                        when it is a matched base, for which the other mate is probed there (tcmi_probe).
#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "tcmi_internal.h"

namespace {

const char NT16[] = "=ACMGRSVTWYHKDBN";

inline bool consumes_ref(unsigned op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
inline bool is_match(unsigned op) { return op == 0 || op == 7 || op == 8; }
inline bool consumes_query(unsigned op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }

// p->indel of htslib's resolve_cigar2 at the last base of op k
int64_t indel_after(const uint32_t *cg, int64_t n, int64_t k)
{
    if (k + 1 >= n) return 0;
    const unsigned op = cg[k] & 0xF, op2 = cg[k + 1] & 0xF;
    int64_t tot = 0;
    if (op2 == 2 && op != 2) {
        tot = -(int64_t)(cg[k + 1] >> 4);
        for (int64_t j = k + 2; j < n && (cg[j] & 0xF) == 2; ++j) tot -= cg[j] >> 4;
    } else if (op2 == 1) {
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
    return tot;
}

// One read on one candidate column, before the filters that depend on the other reads of the column.
struct Entry {
    uint64_t key;               // packed token (see below), or 0: the text is long_tokens[tok] (insertions of more than 12 bases)
    int32_t tok = -1;           // (an index, not a string: an entry per read and column is made a few million times per file)
    uint64_t name_hash;         // 0: unnamed
    int64_t end;                // end of the read on the reference (exclusive)
    int32_t pos, tid, mtid, mpos, isize, l_qseq;
    uint16_t flag;
    uint8_t qual;               // base quality pysam tests: at the column's query position, or — deletion / ref-skip tokens — of the next base
    uint8_t base;               // 4-bit code of that base
    bool on_base;               // the token's first character is a base of this read sitting on the column
    int64_t idx = -1;           // the read (for probes): index in the caller's arrays / compacted index on the device
    int32_t qref = -1;          // !on_base: reference position of the matched base whose quality is tested, or -1: that base is
                                // not a matched one (inserted, clipped, beyond SEQ) and no overlap tweak can reach it
};

// Tokens are PACKED into 64 bits where they fit (a base, a deletion of any length, an insertion of <= 12 bases): no string
// is built per read; the text is produced once, for the modal token.
//   bit 63 set | bits 0-7 first character | bits 8-9 kind (0 plain, 1 insertion, 2 deletion)
//   insertion: bits 10-13 length, bit 14 reverse strand (only if an inserted base is '=': its text depends on the
//              strand), bits 15-62 the inserted bases as BAM 4-bit codes;  deletion: bits 10-41 length
// Distinct tokens of one column in first-seen order.  A column holds a handful of distinct tokens and one of
// them nearly always repeats, so a move-to-front probe of a short list beats hashing every token.
struct Column {
    struct Seen { uint64_t key; std::string tok; int64_t count; };   // key == 0: `tok` holds the text (long insertions)
    std::vector<Seen> seen;
    size_t last = 0;       // entry the previous token matched
    int64_t n = 0;
    std::unordered_map<std::string, size_t> index;                 // only past 64 distinct tokens
    std::unordered_map<uint64_t, size_t> kindex;

    void grow_index()
    {
        if (seen.size() != 65) return;                             // many distinct tokens (noisy inserts): hash them from here on
        for (size_t e = 0; e < seen.size(); ++e) {
            if (seen[e].key) kindex.emplace(seen[e].key, e);
            else index.emplace(seen[e].tok, e);
        }
    }
    void add_key(uint64_t key)
    {
        ++n;
        if (last < seen.size() && seen[last].key == key) { ++seen[last].count; return; }
        if (seen.size() <= 64) {
            for (size_t e = 0; e < seen.size(); ++e)
                if (seen[e].key == key) { ++seen[e].count; last = e; return; }
        } else {
            auto it = kindex.find(key);
            if (it != kindex.end()) { ++seen[it->second].count; last = it->second; return; }
        }
        seen.push_back(Seen{key, std::string(), 1});
        last = seen.size() - 1;
        if (seen.size() > 65) kindex.emplace(key, last);
        grow_index();
    }
    void add_text(const std::string &t)
    {
        ++n;
        if (last < seen.size() && !seen[last].key && seen[last].tok == t) { ++seen[last].count; return; }
        if (seen.size() <= 64) {
            for (size_t e = 0; e < seen.size(); ++e)
                if (!seen[e].key && seen[e].tok == t) { ++seen[e].count; last = e; return; }
        } else {
            auto it = index.find(t);
            if (it != index.end()) { ++seen[it->second].count; last = it->second; return; }
        }
        seen.push_back(Seen{0, t, 1});
        last = seen.size() - 1;
        if (seen.size() > 65) index.emplace(t, last);
        grow_index();
    }
};

// htslib tweak_overlap_quality at ONE reference position where both mates have a matched base: `first` is the mate that
// arrived first.  Agreeing bases: first += second (at most 200), second = 0; differing: the higher one x 0.8, the other 0.
inline void tweak_pair(uint8_t base_first, uint8_t &q_first, uint8_t base_second, uint8_t &q_second)
{
    if (base_first == base_second) {
        const int q = (int)q_first + (int)q_second;
        q_first = (uint8_t)(q > 200 ? 200 : q);
        q_second = 0;
    } else if (q_first >= q_second) {
        q_first = (uint8_t)(0.8 * q_first);
        q_second = 0;
    } else {
        q_second = (uint8_t)(0.8 * q_second);
        q_first = 0;
    }
}

// The column's entries in file order -> the tokens pysam's default pileup would yield, counted.
//   status bit 0: max_depth dropped reads (modelled);  bit 1: a pair of overlapping mates whose quality tweak could not be
//   evaluated (no prober given: the caller refuses rather than guesses)
int finalize_column(std::vector<Entry> &es, const std::vector<std::string> &long_tokens, int32_t min_base_quality, int64_t max_depth,
                    int ignore_overlaps, Column &C, int32_t *status, const tcmi_prober *prober)
{
    // admission: htslib bam_plp_push
    std::vector<uint8_t> in(es.size(), 1);
    if (max_depth > 0) {
        int64_t live = 0;
        int64_t engine_pos = INT64_MIN;
        for (size_t i = 0; i < es.size(); ++i) {
            if ((int64_t)es[i].pos == engine_pos && live + 1 > max_depth) { in[i] = 0; *status |= 1; continue; }
            ++live;
            engine_pos = es[i].pos;
        }
    }
    // overlapping mates: htslib overlap_push + tweak_overlap_quality
    if (ignore_overlaps) {
        struct Later { size_t x, y; bool x_first; };           // x: the mate tested on its next base; y: the mate to probe there
        std::vector<Later> later;
        std::unordered_map<uint64_t, size_t> waiting;
        for (size_t i = 0; i < es.size(); ++i) {
            Entry &b = es[i];
            if (!in[i] || !b.name_hash) continue;
            if ((b.flag & 0x8) || !(b.flag & 0x2)) continue;                         // mate unmapped / not a proper pair
            if ((b.mtid >= 0 && b.tid != b.mtid) ||
                (std::llabs((long long)b.isize) >= 2ll * b.l_qseq && (int64_t)b.mpos >= b.end)) continue;
            auto it = waiting.find(b.name_hash);
            if (it == waiting.end()) {
                if (b.mpos >= b.pos || ((b.flag & 0x1) && b.mpos == -1)) waiting.emplace(b.name_hash, i);
                continue;
            }
            const size_t ia = it->second;
            Entry &a = es[ia];
            waiting.erase(it);
            if (a.on_base && b.on_base) {
                tweak_pair(a.base, a.qual, b.base, b.qual);
                continue;
            }
            // A mate without a base on the column leaves the other's quality on the column alone; its own token is tested on
            // its next query base, which the tweak reaches if that base is a matched one and the other mate has a matched base
            // on the same reference position.
            if (!a.on_base && a.qref >= 0) later.push_back(Later{ia, i, true});
            if (!b.on_base && b.qref >= 0) later.push_back(Later{i, ia, false});
        }
        if (!later.empty()) {
            if (!prober || !*prober) *status |= 2;
            else {
                std::vector<tcmi_probe_req> req(later.size());
                std::vector<tcmi_probe_res> res(later.size());
                for (size_t t = 0; t < later.size(); ++t) { req[t].idx = es[later[t].y].idx; req[t].ref = es[later[t].x].qref; }
                const int rc = (*prober)(req, res);
                if (rc) return rc;
                for (size_t t = 0; t < later.size(); ++t) {
                    if (!res[t].matched) continue;
                    Entry &x = es[later[t].x];
                    uint8_t qy = res[t].qual;                  // (the other mate's quality THERE: its token on this column does not use it)
                    if (later[t].x_first) tweak_pair(x.base, x.qual, res[t].base, qy);
                    else tweak_pair(res[t].base, qy, x.base, x.qual);
                }
            }
        }
    }
    for (size_t i = 0; i < es.size(); ++i) {
        if (!in[i] || (int)es[i].qual < min_base_quality) continue;                   // pileup_base_qual_skip
        if (es[i].key) C.add_key(es[i].key);
        else C.add_text(long_tokens[(size_t)es[i].tok]);
    }
    return TCMI_OK;
}

uint64_t fnv1a(const char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
    return h ? h : 1;
}

inline void append_number(std::string &s, int64_t v)
{
    char d[24];
    int n = 0;
    do { d[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(d[--n]);
}

std::string text_of(uint64_t key)
{
    std::string t(1, (char)(key & 0xFF));
    const unsigned kind = (unsigned)(key >> 8) & 3u;
    if (kind == 1) {
        const int64_t len = (int64_t)(key >> 10) & 15;
        const bool rev = (key >> 14) & 1;
        t.push_back('+');
        append_number(t, len);
        for (int64_t i = 0; i < len; ++i) {
            const char ch = NT16[(key >> (15 + 4 * i)) & 15];
            t.push_back(ch == '=' ? (rev ? ',' : '.') : ch);
        }
    } else if (kind == 2) {
        const int64_t len = (int64_t)((key >> 10) & 0xFFFFFFFFull);
        t.push_back('-');
        append_number(t, len);
        t.append((size_t)len, 'N');
    }
    return t;
}

// every column's entries -> its modal token (first-seen tie-break) and its token count
int emit_modal(std::vector<std::vector<Entry>> &entries, const std::vector<std::string> &long_tokens, int32_t min_base_quality, int64_t max_depth,
               int ignore_overlaps, const tcmi_prober *prober, char *tokens,
               int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens, int32_t *status)
{
    const size_t n_pos = entries.size();
    int64_t off = 0;
    for (size_t k = 0; k < n_pos; ++k) {
        Column col;
        const int frc = finalize_column(entries[k], long_tokens, min_base_quality, max_depth, ignore_overlaps, col, status, prober);
        if (frc) return frc;
        token_off[k] = off;
        n_tokens[k] = col.n;
        const Column::Seen *best = nullptr;
        int64_t bc = 0;
        for (auto &e : col.seen)                                 // first-seen order: ties go to the earliest
            if (e.count > bc) {
                best = &e; bc = e.count;
            }
        if (best) {
            const std::string text = best->key ? text_of(best->key) : best->tok;
            if (off + (int64_t)text.size() > tokens_cap) return tcmi_fail(nullptr, TCMI_E_ARG, "token buffer too small");
            std::memcpy(tokens + off, text.data(), text.size());
            off += (int64_t)text.size();
        }
    }
    token_off[n_pos] = off;
    return TCMI_OK;
}

} // namespace

// Entries produced on the device (pack_device.hip, ins_entries_kernel) -> the same finalisation as the host sweep.
int tcmi_modal_from_dev_entries(int32_t n_pos, const tcmi_dev_entry *ents, const int64_t *ent_off, const int32_t *ent_cnt,
                                int32_t min_base_quality, int64_t max_depth, int ignore_overlaps, const tcmi_prober *prober, char *tokens,
                                int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens, int32_t *status_flags, const uint8_t *long_text,
                                size_t long_bytes)
{
    std::vector<std::vector<Entry>> entries((size_t)n_pos);
    std::vector<std::string> long_tokens;                      // the text of insertions of more than 12 bases (as the host sweep builds it)
    std::string tok;
    int32_t status = 0;
    for (int32_t k = 0; k < n_pos; ++k) {
        // the entries of a column arrive in file order, one slot per read that could reach it; key 0: no token there
        std::vector<const tcmi_dev_entry *> order;
        order.reserve((size_t)ent_cnt[k]);
        for (int32_t t = 0; t < ent_cnt[k]; ++t)
            if (ents[ent_off[k] + t].key) order.push_back(ents + ent_off[k] + t);
        auto &E = entries[(size_t)k];
        E.reserve(order.size());
        for (const tcmi_dev_entry *d : order) {
            Entry e;
            e.key = d->key;
            if (d->bits & 0x40) {
                const uint64_t at = (d->key >> 8) & 0xFFFFFFFFull, n = (d->key >> 40) & 0x7FFFFFull;
                if ((d->bits & 0x80) || !long_text || at + n > long_bytes)
                    return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED, "the insertions of more than 12 bases on the candidate columns exceed the device's text buffer: host sweep");
                const bool rev = (d->flag & 0x10) != 0;
                tok.clear();
                tok.push_back((char)(d->key & 0xFF));
                tok.push_back('+');
                append_number(tok, (int64_t)n);
                for (uint64_t t = 0; t < n; ++t) {
                    const char ch = NT16[long_text[at + t] & 15u];
                    tok.push_back(ch == '=' ? (rev ? ',' : '.') : ch);
                }
                e.key = 0;
                e.tok = (int32_t)long_tokens.size();
                long_tokens.push_back(tok);
            } e.name_hash = d->name_hash; e.end = d->end; e.pos = d->pos; e.tid = 0;
            e.mtid = (d->bits & 0x20) ? 1 : 0;                 // only "same reference or not" matters
            e.mpos = d->mpos; e.isize = d->isize; e.l_qseq = d->l_qseq; e.flag = d->flag; e.qual = d->qual;
            e.base = d->bits & 0xF; e.on_base = (d->bits & 0x10) != 0;
            e.idx = d->j; e.qref = d->qref;
            E.push_back(std::move(e));
        }
    }
    const int rc = emit_modal(entries, long_tokens, min_base_quality, max_depth, ignore_overlaps, prober, tokens, tokens_cap, token_off, n_tokens, &status);
    if (status_flags) *status_flags = status;
    return rc;
}

extern "C" int tcmi_modal_tokens(const tcmi_reads *r, int32_t n_pos, const int64_t *positions /* 1-based, ascending */,
                                 int32_t min_base_quality, uint32_t flag_filter, int ignore_orphans, int64_t max_depth,
                                 int ignore_overlaps, char *tokens, int64_t tokens_cap, int64_t *token_off /* [n_pos+1] */,
                                 int64_t *n_tokens /* [n_pos] */, int32_t *status_flags)
{
    if (!r || n_pos < 0 || (n_pos > 0 && (!positions || !tokens || !token_off || !n_tokens)))
        return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    for (int32_t k = 1; k < n_pos; ++k)
        if (positions[k] <= positions[k - 1]) return tcmi_fail(nullptr, TCMI_E_ARG, "positions must ascend");
    int32_t status = 0;
    std::vector<std::vector<Entry>> entries((size_t)n_pos);
    // Which reads to visit: all of them, or — when the caller promises sorted reads and a span bound
    // (tcmi_reads.sorted_max_span) — only those that can reach one of the candidate columns.
    std::vector<std::pair<int64_t, int64_t>> ranges;           // [first, last) read indices, ascending, disjoint
    const bool windowed = r->sorted_max_span > 0 && (!r->qual || r->qual_off) && n_pos > 0;
    if (windowed) {
        int64_t n_placed = r->n_reads;                         // unplaced reads (tid < 0 / pos < 0) sit at the end
        {
            int64_t lo = 0, hi = r->n_reads;
            while (lo < hi) {
                const int64_t mid = (lo + hi) / 2;
                if ((r->tid && r->tid[mid] < 0) || r->pos[mid] < 0) hi = mid; else lo = mid + 1;
            }
            n_placed = lo;
        }
        const int32_t *pb = r->pos, *pe = r->pos + n_placed;
        for (int32_t k = 0; k < n_pos; ++k) {
            const int64_t col = positions[k] - 1;
            const int64_t a = std::lower_bound(pb, pe, (int32_t)std::max<int64_t>(col - r->sorted_max_span + 1, INT32_MIN / 2),
                                               [](int32_t v, int32_t key) { return v < key; }) - pb;
            const int64_t b = std::upper_bound(pb, pe, (int32_t)std::min<int64_t>(col, INT32_MAX)) - pb;
            if (a >= b) continue;
            if (!ranges.empty() && a <= ranges.back().second) ranges.back().second = std::max(ranges.back().second, b);
            else ranges.emplace_back(a, b);
        }
    } else {
        ranges.emplace_back(0, r->n_reads);
    }
    int64_t qoff = 0;
    std::string tok;
    std::vector<std::string> long_tokens;
    for (const auto &range : ranges)
    for (int64_t i = range.first; i < range.second; ++i) {
        if (windowed && r->qual) qoff = (int64_t)r->qual_off[i];
        const int64_t lq = r->l_qseq[i];
        const int64_t my_qoff = qoff;
        qoff += lq;
        const unsigned fl = r->flag[i];
        if ((fl & 0x4) || (r->tid && r->tid[i] != 0) || r->pos[i] < 0) continue;   // Events.py:64: references[0] only
        if (fl & flag_filter) continue;
        if (ignore_orphans && (fl & 0x1) && !(fl & 0x2)) continue;
        const uint32_t *cg = r->cigar + r->cigar_off[i];
        const int64_t nc = (int64_t)(r->cigar_off[i + 1] - r->cigar_off[i]);
        int64_t span = 0;
        for (int64_t k = 0; k < nc; ++k)
            if (consumes_ref(cg[k] & 0xF)) span += cg[k] >> 4;
        if (span == 0) continue;
        const int64_t beg = r->pos[i], end = beg + span;      // 0-based half-open
        // candidate columns c = position-1 in [beg, end)
        const int64_t *lo = std::lower_bound(positions, positions + n_pos, beg + 1);
        for (const int64_t *pp = lo; pp < positions + n_pos && *pp - 1 < end; ++pp) {
            const int64_t col = *pp - 1;
            std::vector<Entry> &E = entries[(size_t)(pp - positions)];
            // locate the op covering col
            int64_t x = beg, y = 0;
            for (int64_t k = 0; k < nc; ++k) {
                const unsigned op = cg[k] & 0xF;
                const int64_t len = cg[k] >> 4;
                if (consumes_ref(op)) {
                    if (col < x + len) {
                        const bool rev = fl & 0x10;
                        const uint8_t *s = r->seq + r->seq_off[i];
                        auto base = [&](int64_t q) -> char {
                            if (q >= lq) return 'N';
                            const unsigned nib = (q & 1) ? (s[q >> 1] & 0xF) : (s[q >> 1] >> 4);
                            const char ch = NT16[nib];
                            return ch == '=' ? (rev ? ',' : '.') : ch;       // upper-cased text
                        };
                        const int64_t qpos = is_match(op) ? y + (col - x) : y;
                        int q = 255;
                        if (r->qual) q = qpos < lq ? r->qual[my_qoff + qpos] : 0;
                        Entry e;
                        e.key = 0;
                        e.qual = (uint8_t)q;
                        e.on_base = is_match(op);
                        e.base = (uint8_t)(qpos < lq ? ((qpos & 1) ? (s[qpos >> 1] & 0xF) : (s[qpos >> 1] >> 4)) : 15);
                        e.idx = i;
                        e.qref = -1;
                        if (!e.on_base && qpos < lq) {          // the next query base: a matched one? then on which reference position
                            int64_t xr = x + len;
                            for (int64_t k2 = k + 1; k2 < nc; ++k2) {
                                const unsigned o2 = cg[k2] & 0xF;
                                if (is_match(o2)) { if ((cg[k2] >> 4) > 0 && xr <= INT32_MAX) e.qref = (int32_t)xr; break; }
                                if (consumes_query(o2) && (cg[k2] >> 4) > 0) break;      // inserted / clipped: no reference position
                                if (consumes_ref(o2)) xr += cg[k2] >> 4;
                            }
                        }
                        e.pos = r->pos[i]; e.end = end; e.flag = (uint16_t)fl; e.l_qseq = (int32_t)lq;
                        e.tid = r->tid ? r->tid[i] : 0;
                        e.mtid = r->next_tid ? r->next_tid[i] : -1;
                        e.mpos = r->next_pos ? r->next_pos[i] : -1;
                        e.isize = r->tlen ? r->tlen[i] : 0;
                        e.name_hash = (r->names && r->name_off) ? fnv1a(r->names + r->name_off[i], (size_t)(r->name_off[i + 1] - r->name_off[i])) : 0;
                        const char first = is_match(op) ? base(qpos) : (op == 3 ? (rev ? '<' : '>') : '*');
                        const int64_t indel = col == x + len - 1 ? indel_after(cg, nc, k) : 0;
                        if (indel <= 12 && -indel <= 0xFFFFFFFFll) {
                            uint64_t key = (1ull << 63) | (uint8_t)first;
                            if (indel > 0) {
                                key |= (1ull << 8) | ((uint64_t)indel << 10);
                                bool any_eq = false;
                                for (int64_t t = 1; t <= indel; ++t) {
                                    const int64_t q2 = qpos + t;
                                    // a base beyond SEQ reads as 'N' (code 15)
                                    const unsigned nib = q2 >= lq ? 15u : ((q2 & 1) ? (s[q2 >> 1] & 0xFu) : (unsigned)(s[q2 >> 1] >> 4));
                                    any_eq |= nib == 0;
                                    key |= (uint64_t)nib << (15 + 4 * (t - 1));
                                }
                                if (any_eq && rev) key |= 1ull << 14;
                            } else if (indel < 0) {
                                key |= (2ull << 8) | ((uint64_t)(-indel) << 10);
                            }
                            e.key = key;
                        } else {
                            tok.clear();
                            tok.push_back(first);
                            tok.push_back('+');
                            append_number(tok, indel);
                            for (int64_t t = 1; t <= indel; ++t) tok.push_back(base(qpos + t));
                            e.tok = (int32_t)long_tokens.size();
                            long_tokens.push_back(tok);
                        }
                        E.push_back(std::move(e));
                        break;
                    }
                    x += len;
                }
                if (consumes_query(op)) y += len;
            }
        }
    }
    // the other mate of a pair, probed on one reference position: matched base there? which, with what quality
    const tcmi_prober prober = [&](const std::vector<tcmi_probe_req> &req, std::vector<tcmi_probe_res> &res) {
        std::vector<int64_t> qo;
        if (r->qual && !r->qual_off) { qo.resize((size_t)r->n_reads + 1, 0); for (int64_t i = 0; i < r->n_reads; ++i) qo[(size_t)i + 1] = qo[(size_t)i] + r->l_qseq[i]; }
        for (size_t t = 0; t < req.size(); ++t) {
            res[t] = tcmi_probe_res{0, 15, 0};
            const int64_t i = req[t].idx;
            if (i < 0 || i >= r->n_reads) continue;
            const uint32_t *cg = r->cigar + r->cigar_off[i];
            const int64_t nc = (int64_t)(r->cigar_off[i + 1] - r->cigar_off[i]), lq = r->l_qseq[i];
            int64_t x = r->pos[i], y = 0;
            for (int64_t k = 0; k < nc; ++k) {
                const unsigned op = cg[k] & 0xF;
                const int64_t len = cg[k] >> 4;
                if (consumes_ref(op)) {
                    if (req[t].ref < x + len) {
                        if (is_match(op) && req[t].ref >= x) {
                            const int64_t q = y + (req[t].ref - x);
                            if (q < lq) {
                                const uint8_t *sq = r->seq + r->seq_off[i];
                                res[t].matched = 1;
                                res[t].base = (uint8_t)((q & 1) ? (sq[q >> 1] & 0xF) : (sq[q >> 1] >> 4));
                                res[t].qual = r->qual ? r->qual[(r->qual_off ? (int64_t)r->qual_off[i] : qo[(size_t)i]) + q] : 255;
                            }
                        }
                        break;
                    }
                    x += len;
                }
                if (consumes_query(op)) y += len;
            }
        }
        return (int)TCMI_OK;
    };
    const int rc = emit_modal(entries, long_tokens, min_base_quality, max_depth, ignore_overlaps, &prober, tokens, tokens_cap, token_off, n_tokens, &status);
    if (rc) return rc;
    if (status_flags) *status_flags = status;
    return TCMI_OK;
}
