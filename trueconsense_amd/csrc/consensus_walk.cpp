// consensus_walk.cpp — sequential part of stage B, on the HOST, over the GPU's call records.
//
// Restates the loop of Sequences.BuildConsensus (TrueConsense/Sequences.py:179-322) together
// with ORFs.in_orf (ORFs.py:1-26), SolveTripletLength (:45-77), CorrectStartPositions (:80-108)
// and CorrectGFF (:111-192).  The reference re-joins the whole consensus and re-splits it into
// codons for every position inside an ORF (O(sum ORF_len^2), 95 % of its stage-B time,
// SURVEY §3.3); here each active ORF carries an incremental scanner instead, so the walk is
// O(L + inserted bases) with identical results, quirks included:
//   * ORF membership is the half-open range(start, end) over ALL GFF rows, on the evolving
//     coordinates; only '+' rows get their end corrected;
//   * an accepted insert at p shifts the START of every ORF with start > p by int(size_str)
//     (last digit only), also in the include_ins=0 run; ends are never shifted;
//   * the tail string is indexed by `start-1` into the joined consensus (inserted characters
//     included), '-' characters are counted as `gaps` and stripped before codon splitting,
//     only upper-case TAG/TAA/TGA are stops;
//   * end := newend only when p == newend; with the last appended element "-" end := original.
#include <algorithm>
#include <cstring>
#include <vector>

#include "tcmi_internal.h"

static bool g_fast_runs = true;      // tests switch the run fast path off to compare both walks

extern "C" void tcmi_walk_set_fast_runs(int on) { g_fast_runs = on != 0; }

namespace {

struct Orf {
    int64_t start, end, orig_end;
    bool plus;
    // incremental scanner over J[start-1:] (valid once `scanning`)
    bool scanning = false;
    int64_t consumed = 0;      // index into J of the next unread character
    int64_t n_bare = 0;        // characters other than '-'
    int64_t gaps = 0;          // '-' characters
    int64_t it_stop = 0;       // 1-based codon index of the first stop, 0 = none yet
    char c0 = 0, c1 = 0;       // partial codon

    void feed(const char *J, int64_t len)
    {
        for (; consumed < len; ++consumed) {
            const char ch = J[consumed];
            if (ch == '-') { ++gaps; continue; }
            const int ph = (int)(n_bare % 3);
            ++n_bare;
            if (ph == 0) c0 = ch;
            else if (ph == 1) c1 = ch;
            else if (!it_stop && c0 == 'T' &&
                     ((c1 == 'A' && (ch == 'G' || ch == 'A')) || (c1 == 'G' && ch == 'A')))
                it_stop = n_bare / 3;
        }
    }
    // the same, but returns right after the character that completes the first stop codon
    void feed_until_stop(const char *J, int64_t len)
    {
        while (consumed < len) {
            const char ch = J[consumed++];
            if (ch == '-') { ++gaps; continue; }
            const int ph = (int)(n_bare % 3);
            ++n_bare;
            if (ph == 0) c0 = ch;
            else if (ph == 1) c1 = ch;
            else if (c0 == 'T' && ((c1 == 'A' && (ch == 'G' || ch == 'A')) || (c1 == 'G' && ch == 'A'))) {
                it_stop = n_bare / 3;
                return;
            }
        }
    }
};

inline bool triplet_ok(int64_t n_up, int64_t n_min)       // ORFs.py:45-77
{
    return (n_up % 3 == 0) ? (n_min % 3 == 0) : ((n_min + n_up) % 3 == 0);
}

} // namespace

extern "C" int tcmi_consensus_walk(const uint8_t *plain, const uint8_t *alt, const uint8_t *flags, int64_t L,
                                   int32_t n_orf, const int64_t *orf_start, const int64_t *orf_end,
                                   const uint8_t *orf_is_plus, int32_t n_ins, const int64_t *ins_pos,
                                   const int32_t *ins_shift, const char *ins_seq, const int64_t *ins_off,
                                   int include_ins, char *J, int64_t cap, int64_t *out_len, int64_t *new_start,
                                   int64_t *new_end, int64_t *err_pos)
{
    if (!plain || !alt || !flags || L < 0 || !J || !out_len) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    if (n_orf < 0 || (n_orf > 0 && (!orf_start || !orf_end || !orf_is_plus || !new_start || !new_end)))
        return tcmi_fail(nullptr, TCMI_E_ARG, "bad ORF arrays");
    if (n_ins < 0 || (n_ins > 0 && (!ins_pos || !ins_shift || !ins_seq || !ins_off)))
        return tcmi_fail(nullptr, TCMI_E_ARG, "bad insert arrays");
    for (int32_t k = 1; k < n_ins; ++k)
        if (ins_pos[k] <= ins_pos[k - 1]) return tcmi_fail(nullptr, TCMI_E_ARG, "insert positions must ascend");
    if (err_pos) *err_pos = 0;

    std::vector<Orf> orfs((size_t)n_orf);
    for (int32_t k = 0; k < n_orf; ++k) {
        if (orf_start[k] < 1 && orf_start[k] < orf_end[k])
            return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED, "GFF row %d starts at %lld (< 1)", k, (long long)orf_start[k]);
        orfs[(size_t)k].start = orf_start[k];
        orfs[(size_t)k].end = orfs[(size_t)k].orig_end = orf_end[k];
        orfs[(size_t)k].plus = orf_is_plus[k] != 0;
    }
    // pending = rows not yet reached, ordered by start (shifts preserve this order: they are
    // applied to every row with start > p alike); live = rows with start <= p < end
    std::vector<int32_t> pending((size_t)n_orf), live;
    for (int32_t k = 0; k < n_orf; ++k) pending[(size_t)k] = k;
    for (size_t a = 1; a < pending.size(); ++a)              // insertion sort, descending start (pop from back)
        for (size_t b = a; b > 0 && orfs[(size_t)pending[b - 1]].start < orfs[(size_t)pending[b]].start; --b)
            std::swap(pending[b - 1], pending[b]);

    int64_t len = 0;
    int64_t skip_end = 0;          // positions < skip_end are in `dskips` (always one contiguous group)
    int32_t next_ins = 0;

    auto run_after = [&](int64_t p, int64_t *n) -> bool {     // Sequences.py:44-53; false = KeyError(L+1)
        int64_t q = p + 1;
        while (q <= L && (flags[q - 1] & TCMI_F_PRIMX)) ++q;
        if (q > L) return false;
        *n = q - p - 1;
        return true;
    };
    auto key_error = [&]() {
        if (err_pos) *err_pos = L + 1;
        return tcmi_fail(nullptr, TCMI_E_KEYERROR, "KeyError: %lld (a deletion walk ran past the last position)",
                         (long long)(L + 1));
    };

    for (int64_t p = 1; p <= L; ++p) {
        const unsigned f = flags[p - 1];
        while (!pending.empty() && orfs[(size_t)pending.back()].start <= p) {
            const int32_t k = pending.back();
            pending.pop_back();
            if (p < orfs[(size_t)k].end) live.push_back(k);
        }
        for (size_t a = 0; a < live.size();)                  // drop rows that ended
            if (p >= orfs[(size_t)live[a]].end) { live[a] = live.back(); live.pop_back(); } else ++a;
        while (next_ins < n_ins && ins_pos[next_ins] < p) ++next_ins;
        const bool ins_here = next_ins < n_ins && ins_pos[next_ins] == p;

        // ---- fast path: a run of positions where nothing sequential happens (primary is a base, no
        //      minority deletion, no pending skip, no insert, no row starting or ending): the characters
        //      are the call kernel's, and each live '+' row only has to be scanned for its first stop.
        if (g_fast_runs && p >= skip_end && !(f & (TCMI_F_PRIMX | TCMI_F_MINDEL)) && !ins_here) {
            int64_t lim = L + 1;
            if (!pending.empty()) lim = std::min(lim, orfs[(size_t)pending.back()].start);
            for (int32_t k : live) lim = std::min(lim, orfs[(size_t)k].end);
            if (next_ins < n_ins) lim = std::min(lim, ins_pos[next_ins]);
            int64_t q = p + 1;
            while (q < lim && !(flags[q - 1] & (TCMI_F_PRIMX | TCMI_F_MINDEL))) ++q;
            const int64_t n = q - p, len0 = len;
            if (len + n > cap) return tcmi_fail(nullptr, TCMI_E_ARG, "consensus buffer too small");
            std::memcpy(J + len, plain + (p - 1), (size_t)n);
            len += n;
            for (int32_t k : live) {
                Orf &o = orfs[(size_t)k];
                if (!o.plus) continue;
                if (!o.scanning) { o.scanning = true; o.consumed = o.start - 1; }
                // position t of the run has appended J[.. len0 + (t - p) + 1)
                int64_t t_stop = p;                                    // first position at which it_stop is known
                if (!o.it_stop) {
                    o.feed(J, len0 + 1);                              // catch up to position p (may cross older '-')
                    if (!o.it_stop) o.feed_until_stop(J, len);         // then one character per position
                    t_stop = o.it_stop ? std::max<int64_t>(p, p + (o.consumed - len0) - 1) : q;
                }
                if (o.it_stop) {
                    // from t_stop on the check of ORFs.py:183-189 sees a constant newend (no '-' is added in a run)
                    const int64_t gaps_at = o.gaps;                   // '-' never occurs inside the run
                    const int64_t newend = o.start + 3 * o.it_stop + gaps_at - 1;
                    if (newend >= t_stop && newend < q) o.end = newend;
                    o.feed(J, len);                                    // keep the scanner in step with J
                }
            }
            p = q - 1;
            continue;
        }

        if (len + 1 > cap) return tcmi_fail(nullptr, TCMI_E_ARG, "consensus buffer too small");
        bool last_is_dash = false;
        bool spliceable = false;
        if (p < skip_end) {                                   // Sequences.py:184-189
            J[len++] = '-';
            last_is_dash = true;
        } else if (f & TCMI_F_LOWCOV) {                       // Sequences.py:191-197
            J[len++] = 'N';
        } else {
            spliceable = true;
            char ch;
            if (!(f & TCMI_F_PRIMX)) {                        // Sequences.py:210-275
                int64_t group = 0;
                if (f & TCMI_F_MINDEL) {
                    int64_t n_up;
                    if (!run_after(p, &n_up)) return key_error();
                    if (n_up > 0) {
                        if (triplet_ok(n_up, 1)) group = 1 + n_up;
                    } else {
                        const unsigned f1 = flags[p];        // position p+1 exists: run_after(p) returned
                        if (f1 & TCMI_F_COVZERO)
                            return tcmi_fail(nullptr, TCMI_E_ZERODIV, "ZeroDivisionError: coverage 0 at position %lld",
                                             (long long)(p + 1));
                        if (f1 & TCMI_F_MINDEL) {
                            int64_t n_up2;
                            if (!run_after(p + 1, &n_up2)) return key_error();
                            if (n_up2 > 0 && triplet_ok(n_up2, 2)) group = 2 + n_up2;
                        }
                    }
                }
                if (group) { ch = '-'; skip_end = p + group; } else ch = (char)plain[p - 1];
            } else if (!live.empty()) {                       // Sequences.py:277-306 (inside an ORF)
                int64_t n_up;
                if (!run_after(p, &n_up)) return key_error();
                if (n_up >= 2) { ch = '-'; skip_end = p + 1 + n_up; } else ch = (char)alt[p - 1];
            } else {
                ch = '-';                                     // Sequences.py:307-308
            }
            J[len++] = ch;
            last_is_dash = ch == '-';
        }
        if (spliceable && include_ins && ins_here && (f & TCMI_F_COVGT)) {   // Sequences.py:310-316
            const int64_t n = ins_off[next_ins + 1] - ins_off[next_ins];
            if (len + n > cap) return tcmi_fail(nullptr, TCMI_E_ARG, "consensus buffer too small");
            std::memcpy(J + len, ins_seq + ins_off[next_ins], (size_t)n);
            len += n;
            if (n > 0) last_is_dash = false;                  // cons[-1] is the insert string
        }

        // ---- CorrectGFF (ORFs.py:111-192) ----
        if (ins_here && (f & TCMI_F_COVGT)) {                 // ORFs.py:141-145 -> :80-108
            for (int32_t k : pending) orfs[(size_t)k].start += ins_shift[next_ins];
        }
        for (int32_t k : live) {
            Orf &o = orfs[(size_t)k];
            if (!o.plus) continue;
            if (!o.scanning) { o.scanning = true; o.consumed = o.start - 1; }
            o.feed(J, len);
            if (last_is_dash) { o.end = o.orig_end; continue; }
            int64_t newend;
            if (o.it_stop) newend = o.start + 3 * o.it_stop + o.gaps - 1;
            else newend = o.start + 3 * ((o.n_bare + 2) / 3) + o.gaps;
            if (p == newend) o.end = newend;
        }
    }
    *out_len = len;
    for (int32_t k = 0; k < n_orf; ++k) { new_start[k] = orfs[(size_t)k].start; new_end[k] = orfs[(size_t)k].end; }
    return TCMI_OK;
}
