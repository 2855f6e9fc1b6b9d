/*
 * tally_oracle.c — scalar C restatement of stage A (pileup tally) and of the
 * position-local part of stage B (base call record).  TEST INFRASTRUCTURE ONLY:
 * loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
 * product never links or calls it.
 *
 * Read-major formulation (one pass over the reads, CIGAR walk per read), written
 * independently of the column-major Python emulator in tc_oracle.py so the two can
 * check each other.  Semantics: SURVEY.md §8-P (reference call indexing.py:100,
 * token rule indexing.py:102-132) and §8-Q1..Q7 (Sequences.py:119-165,
 * Ambig.py:18-228, Events.py:29-36, 85-106).
 *
 * PARITY UNPINNED for BAM record -> pileup token (that step is pysam/htslib in the
 * reference and cannot run here); pinned from the token level down by tests/golden.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { COV = 0, CA = 1, CT = 2, CC = 3, CG = 4, CX = 5, CI = 6 };
enum { OP_M = 0, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X };

static int consumes_ref(unsigned op) { return op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X; }
static int is_match(unsigned op) { return op == OP_M || op == OP_EQ || op == OP_X; }

/* 4-bit code -> count column; 0 = none of A/T/C/G ("=ACMGRSVTWYHKDBN": A=1 C=2 G=4 T=8) */
static const int8_t NIB_COL[16] = {0, CA, CC, 0, CG, 0, 0, 0, CT, 0, 0, 0, 0, 0, 0, 0};

/* htslib resolve_cigar2 peek: is an insertion reported on the last base of op k? */
static int ins_after(const uint32_t *cg, int64_t n, int64_t k)
{
    if (k + 1 >= n) return 0;
    unsigned op2 = cg[k + 1] & 0xF;
    int64_t tot = 0;
    if (op2 == OP_I) {
        tot = cg[k + 1] >> 4;
        for (int64_t j = k + 2; j < n; ++j) {
            unsigned o = cg[j] & 0xF;
            if (o == OP_I) tot += cg[j] >> 4;
            else if (o != OP_P) break;
        }
    } else if (op2 == OP_P && k + 2 < n) {
        for (int64_t j = k + 2; j < n; ++j) {
            unsigned o = cg[j] & 0xF;
            if (o == OP_I) tot += cg[j] >> 4;
            else if (consumes_ref(o)) break;
        }
    }
    return tot > 0;
}

static int64_t ref_span(const uint32_t *cg, int64_t n)
{
    int64_t s = 0;
    for (int64_t k = 0; k < n; ++k)
        if (consumes_ref(cg[k] & 0xF)) s += cg[k] >> 4;
    return s;
}

/* max(ref_len, max end of any piled-up read) — indexing.py:137-151 keeps extra columns */
int64_t orc_extent(int64_t n_reads, const int32_t *pos, const uint16_t *flag, const uint64_t *cigar_off,
                   const uint32_t *cigar, const int32_t *tid, int64_t ref_len)
{
    int64_t L = ref_len;
    for (int64_t i = 0; i < n_reads; ++i) {
        if ((flag[i] & 0x4) || (tid && tid[i] < 0)) continue;
        int64_t span = ref_span(cigar + cigar_off[i], (int64_t)(cigar_off[i + 1] - cigar_off[i]));
        if (span > 0 && pos[i] + span > L) L = pos[i] + span;
    }
    return L;
}

/* counts: [L][7] int32, zeroed by the caller (accumulates). Returns piled-up read count. */
int64_t orc_tally(int64_t n_reads, const int32_t *pos, const uint16_t *flag, const int32_t *l_qseq,
                  const uint64_t *cigar_off, const uint32_t *cigar, const uint64_t *seq_off,
                  const uint8_t *seq, const int32_t *tid, int64_t L, int32_t *counts)
{
    int64_t piled = 0;
    for (int64_t i = 0; i < n_reads; ++i) {
        if ((flag[i] & 0x4) || (tid && tid[i] < 0)) continue;          /* §8-P4 */
        const uint32_t *cg = cigar + cigar_off[i];
        int64_t n = (int64_t)(cigar_off[i + 1] - cigar_off[i]);
        if (ref_span(cg, n) == 0) continue;
        ++piled;
        const uint8_t *s = seq + seq_off[i];
        int64_t lq = l_qseq[i];
        int64_t x = pos[i], y = 0;
        for (int64_t k = 0; k < n; ++k) {
            unsigned op = cg[k] & 0xF;
            int64_t len = cg[k] >> 4;
            if (consumes_ref(op)) {
                int ins = len > 0 ? ins_after(cg, n, k) : 0;
                for (int64_t j = 0; j < len; ++j) {
                    int64_t p = x + j;
                    if (p < 0 || p >= L) continue;
                    int32_t *row = counts + 7 * p;
                    row[COV] += 1;                                      /* every token */
                    int last = (j == len - 1);
                    if (is_match(op)) {
                        int64_t q = y + j;
                        unsigned nib = 15;                              /* past SEQ -> 'N' */
                        if (q < lq) nib = (q & 1) ? (s[q >> 1] & 0xF) : (s[q >> 1] >> 4);
                        int col = NIB_COL[nib];
                        if (col) row[col] += 1;
                    } else if (op == OP_D) {
                        if (!(last && ins)) row[CX] += 1;               /* token exactly "*" */
                    }
                    if (last && ins) row[CI] += 1;                      /* '+' in token */
                }
                x += len;
            }
            if (op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X) y += len;
        }
    }
    return piled;
}

/* ---- call record ------------------------------------------------------------- */
#define F_LOWCOV 1
#define F_PRIMX 2
#define F_MINDEL 4
#define F_INSCAND 8
#define F_COVGT 16
#define F_COVZERO 32
#define F_AMBIG 64

/* letter order of Python's tuple sort on (count, letter): A < C < G < T < X */
static const char LET[5] = {'A', 'C', 'G', 'T', 'X'};

static char iupac(unsigned mask)
{   /* bit0 A, bit1 C, bit2 G, bit3 T  (Ambig.py tables) */
    switch (mask) {
    case 0x3: return 'M'; case 0x5: return 'R'; case 0x9: return 'W';
    case 0x6: return 'S'; case 0xA: return 'Y'; case 0xC: return 'K';
    case 0x7: return 'V'; case 0xB: return 'H'; case 0xD: return 'D'; case 0xE: return 'B';
    }
    return '?';
}

static double dabs(double v) { return v < 0 ? -v : v; }

void orc_call(const int32_t *counts, int64_t L, int32_t mincov, int include_ambig,
              uint8_t *plain, uint8_t *alt, uint8_t *flags)
{
    for (int64_t p = 0; p < L; ++p) {
        const int32_t *r = counts + 7 * p;
        int64_t cov = r[COV];
        int64_t c[5] = {r[CA], r[CC], r[CG], r[CT], r[CX]};    /* letter-rank order */
        int ord[5] = {0, 1, 2, 3, 4};
        /* descending by (count, letter rank): insertion sort */
        for (int a = 1; a < 5; ++a)
            for (int b = a; b > 0; --b) {
                int u = ord[b - 1], v = ord[b];
                if (c[v] > c[u] || (c[v] == c[u] && v > u)) { ord[b - 1] = v; ord[b] = u; } else break;
            }
        unsigned f = 0;
        if (cov < mincov) f |= F_LOWCOV;
        if (ord[0] == 4) f |= F_PRIMX;
        if (cov > mincov) f |= F_COVGT;
        if (cov == 0) f |= F_COVZERO;
        else if (((double)r[CX] / (double)cov) * 100.0 >= 15.0) f |= F_MINDEL;
        if (cov >= mincov && cov != 0 && r[CI] != 0 && ((double)r[CI] / (double)cov) * 100.0 > 55.0)
            f |= F_INSCAND;
        char amb = 0;
        if (cov != 0 && ord[0] != 4 && ord[1] != 4) {
            double p1 = ((double)c[ord[0]] / (double)cov) * 100.0;
            double p2 = ((double)c[ord[1]] / (double)cov) * 100.0;
            double p3 = ((double)c[ord[2]] / (double)cov) * 100.0;
            double p4 = ((double)c[ord[3]] / (double)cov) * 100.0;
            if (dabs(p1 - p2) <= 10.0) {
                if (dabs(p1 - p3) <= 10.0 && dabs(p2 - p3) <= 10.0) {
                    if (dabs(p1 - p4) <= 10.0 && dabs(p2 - p4) <= 10.0 && dabs(p3 - p4) <= 10.0) amb = 'N';
                    else if (ord[2] == 4) amb = 'N';
                    else amb = iupac((1u << ord[0]) | (1u << ord[1]) | (1u << ord[2]));
                } else amb = iupac((1u << ord[0]) | (1u << ord[1]));
            }
        }
        if (amb) f |= F_AMBIG;
        char c1 = LET[ord[0]], c2 = LET[ord[1]];
        if (c[ord[0]] < mincov) c1 = (char)(c1 | 0x20);
        if (c[ord[1]] < mincov) c2 = (char)(c2 | 0x20);
        plain[p] = (f & F_LOWCOV) ? 'N' : ((include_ambig && amb) ? amb : c1);
        alt[p] = c2;
        flags[p] = (uint8_t)f;
    }
}
