"""
tc_oracle.py — CPU restatement of the TrueConsense hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (trueconsense_amd/) never does: it fails loudly when the HIP
library is missing instead of falling back to anything in oracle/.

Every function cites the reference lines it restates (paths relative to the upstream
repository, RIVM-bioinformatics/TrueConsense v0.5.2).

Parity status
-------------
* pinned by golden vectors generated from the imported reference
  (tests/golden/make_golden.py): token tally (indexing.py:102-132), ranking
  (Sequences.py:119-165), ambiguity (Ambig.py:18-228), MinorityDel / ListInserts /
  ExtractInserts post-pileup logic (Events.py), BuildConsensus incl. the ORF logic
  (Sequences.py:168-322, ORFs.py:1-192), WriteOutputs text (Outputs.py:74-183).
* PARITY UNPINNED: BAM records -> pileup tokens.  That step lives in pysam 0.23.3 /
  htslib (pyproject.toml:27 of the reference), which is neither vendored nor
  installable here.  `pileup_columns` restates htslib's published pileup algorithm
  (bam_plp_push / resolve_cigar2 in htslib sam.c, PileupColumn.get_query_sequences in
  pysam libcalignedsegment.pyx) from the specification in SURVEY.md §8-P; it is
  cross-checked against an independent read-major implementation (oracle/tally_oracle.c).
"""
from __future__ import annotations

import re
from collections import Counter

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"          # SAM spec §4.2.3
COLS = ("coverage", "A", "T", "C", "G", "X", "I")   # indexing.py:134

# CIGAR op codes (SAM spec §4.2): M I D N S H P = X
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
_REF_OPS = (OP_M, OP_D, OP_N, OP_EQ, OP_X)
_QRY_OPS = (OP_M, OP_I, OP_S, OP_EQ, OP_X)
_MATCH_OPS = (OP_M, OP_EQ, OP_X)

FLAG_PAIRED, FLAG_PROPER, FLAG_UNMAP, FLAG_REV = 0x1, 0x2, 0x4, 0x10
FLAG_SECONDARY, FLAG_QCFAIL, FLAG_DUP = 0x100, 0x200, 0x400
DEFAULT_FILTER = FLAG_UNMAP | FLAG_SECONDARY | FLAG_QCFAIL | FLAG_DUP   # pysam default flag_filter

F_LOWCOV, F_PRIMX, F_MINDEL, F_INSCAND, F_COVGT, F_COVZERO, F_AMBIG = 1, 2, 4, 8, 16, 32, 64


# --------------------------------------------------------------------------- reads
def read_cigar(reads, i):
    a, b = int(reads["cigar_off"][i]), int(reads["cigar_off"][i + 1])
    c = reads["cigar"][a:b]
    return [(int(x) & 0xF, int(x) >> 4) for x in c]


def read_base(reads, i, q):
    """4-bit code of query base q of read i (BAM packing: high nibble first)."""
    byte = int(reads["seq"][int(reads["seq_off"][i]) + (q >> 1)])
    return (byte >> 4) if (q & 1) == 0 else (byte & 0xF)


def ref_length(cigar):
    return sum(l for op, l in cigar if op in _REF_OPS)


def read_piles_up(reads, i):
    """unmapped / tid<0 reads never enter; reads whose CIGAR consumes no
    reference are treated as not piling up (htslib behaviour for them is undefined)."""
    if int(reads["flag"][i]) & FLAG_UNMAP:
        return False
    tid = reads.get("tid")
    if tid is not None and int(tid[i]) < 0:
        return False
    return ref_length(read_cigar(reads, i)) > 0


def _indel_after(cigar, k):
    """htslib resolve_cigar2 peek at the last reference base of op k: returns p->indel."""
    n = len(cigar)
    if k + 1 >= n:
        return 0
    op, _ = cigar[k]
    op2, l2 = cigar[k + 1]
    if op2 == OP_D and op != OP_D:
        tot = -l2
        for j in range(k + 2, n):
            o, l = cigar[j]
            if o == OP_D:
                tot -= l
            else:
                break
        return tot
    if op2 == OP_I:
        tot = l2
        for j in range(k + 2, n):
            o, l = cigar[j]
            if o == OP_I:
                tot += l
            elif o != OP_P:
                break
        return tot
    if op2 == OP_P and k + 2 < n:
        tot = 0
        for j in range(k + 2, n):
            o, l = cigar[j]
            if o == OP_I:
                tot += l
            elif o in _REF_OPS:
                break
        return tot if tot > 0 else 0
    return 0


def read_tokens(reads, i, with_qual=False, qual_override=None):
    """Yield (column, token[, qual_at_qpos]) for read i in column order — the pileup
    entries htslib builds for this read and the text pysam's
    get_query_sequences(add_indels=True) prints for them."""
    cigar = read_cigar(reads, i)
    rev = bool(int(reads["flag"][i]) & FLAG_REV)
    lq = int(reads["l_qseq"][i])
    x = int(reads["pos"][i])
    y = 0
    qual = None
    if with_qual and reads.get("qual") is not None:
        qoff = int(reads["qual_off"][i]) if "qual_off" in reads else None
        qual = (reads["qual"], qoff)

    def case(ch):
        if ch == "=":
            return "," if rev else "."
        return ch.lower() if rev else ch.upper()

    def qual_at(q):
        if qual_override is not None:
            return int(qual_override[q]) if q < lq else 0
        if qual is None or qual[1] is None:
            return 255
        return int(qual[0][qual[1] + q]) if q < lq else 0

    started = False
    for k, (op, l) in enumerate(cigar):
        if op in _REF_OPS:
            started = True
            for j in range(l):
                col = x + j
                if op in _MATCH_OPS:
                    qpos = y + j
                    tok = case(NT16[read_base(reads, i, qpos)]) if qpos < lq else case("N")
                else:
                    qpos = y
                    tok = ("<" if rev else ">") if op == OP_N else "*"
                if j == l - 1:
                    indel = _indel_after(cigar, k)
                    if indel > 0:
                        tok += "+%d" % indel
                        for t in range(1, indel + 1):
                            qq = qpos + t
                            tok += case(NT16[read_base(reads, i, qq)]) if qq < lq else case("N")
                    elif indel < 0:
                        tok += "-%d" % (-indel) + case("N") * (-indel)
                if with_qual:
                    yield col, tok, qual_at(qpos)
                else:
                    yield col, tok
            x += l
        if op in _QRY_OPS:
            y += l
        _ = started


def pileup_columns(reads, flag_filter=0, ignore_orphans=False, min_base_quality=0,
                   only_column=None):
    """Column-major pileup: {0-based column: [token, ...]} with tokens in file order.
    Defaults = the stage-A call of indexing.py:100 (stepper "nofilter", BQ 0)."""
    cols = {}
    n = int(reads["n_reads"])
    need_q = min_base_quality > 0
    for i in range(n):
        if not read_piles_up(reads, i):
            continue
        f = int(reads["flag"][i])
        if f & flag_filter:
            continue
        if ignore_orphans and (f & FLAG_PAIRED) and not (f & FLAG_PROPER):
            continue
        for ent in read_tokens(reads, i, with_qual=need_q):
            col, tok = ent[0], ent[1]
            if only_column is not None and col != only_column:
                continue
            if need_q and ent[2] < min_base_quality:
                continue
            cols.setdefault(col, []).append(tok)
    return cols


# --------------------------------------------------------------------------- stage A
def tally_tokens(tokens):
    """indexing.py:102-132 — one column's tokens -> (coverage, A, T, C, G, X, I)."""
    n = {"a": 0, "t": 0, "c": 0, "g": 0}
    x = ins = 0
    for tok in tokens:
        if tok == "*":
            x += 1
        else:
            k = tok[0].lower()
            if k in n:
                n[k] += 1
        ins += "+" in tok
    return (len(tokens), n["a"], n["t"], n["c"], n["g"], x, ins)


def tally_matrix(reads, ref_len):
    """indexing.py:137-151 — rows for positions 1..max(ref_len, last covered column)."""
    cols = pileup_columns(reads)
    L = max([ref_len] + [c + 1 for c in cols])
    out = np.zeros((L, 7), dtype=np.int64)
    for c, toks in cols.items():
        if c >= 0:
            out[c] = tally_tokens(toks)
    return out


# --------------------------------------------------------------------------- stage B, local
_LETTER_RANK = {"A": 0, "C": 1, "G": 2, "T": 3, "X": 4}   # Python tuple sort on (count, letter)


def ranked(row):
    """Sequences.py:119-165 — [(nuc, count)] best first; ties go to the larger letter."""
    cov, a, t, c, g, x, ins = (int(v) for v in row)
    items = [("A", a), ("T", t), ("C", c), ("G", g), ("X", x)]
    items.sort(key=lambda kv: (kv[1], _LETTER_RANK[kv[0]]), reverse=True)
    return items


_IUPAC = {frozenset("AC"): "M", frozenset("AG"): "R", frozenset("AT"): "W",
          frozenset("CG"): "S", frozenset("CT"): "Y", frozenset("GT"): "K",
          frozenset("ACG"): "V", frozenset("ACT"): "H", frozenset("AGT"): "D",
          frozenset("CGT"): "B"}


def ambiguity(rank, cov):
    """Ambig.py:179-228 (thresholds :156-171, percentages :123-126, fp64)."""
    if cov == 0:
        return False, None
    (n1, c1), (n2, c2), (n3, c3), (n4, c4) = rank[:4]
    if n1 == "X" or n2 == "X":
        return False, None
    p1, p2, p3, p4 = ((c / cov) * 100 for c in (c1, c2, c3, c4))
    if not abs(p1 - p2) <= 10:
        return False, None
    if abs(p1 - p3) <= 10 and abs(p2 - p3) <= 10:
        if abs(p1 - p4) <= 10 and abs(p2 - p4) <= 10 and abs(p3 - p4) <= 10:
            return True, "N"
        if "X" in (n1, n2, n3):
            return True, "N"
        return True, _IUPAC[frozenset((n1, n2, n3))]
    return True, _IUPAC[frozenset((n1, n2))]


def minority_del(row):
    """Events.py:85-106."""
    cov, x = int(row[0]), int(row[5])
    return (x / cov) * 100 >= 15


def insert_candidate(row, mincov):
    """Events.py:29-36."""
    cov, ins = int(row[0]), int(row[6])
    if cov < mincov or cov == 0 or ins == 0:
        return False
    return (ins / cov) * 100 > 55


def call_record(row, mincov, include_ambig):
    """(plain, alt, flags) for one position — what the call kernel emits (include/tcmi.h)."""
    cov = int(row[0])
    rk = ranked(row)
    flags = 0
    if cov < mincov:
        flags |= F_LOWCOV
    if rk[0][0] == "X":
        flags |= F_PRIMX
    if cov == 0:
        flags |= F_COVZERO
    elif minority_del(row):
        flags |= F_MINDEL
    if insert_candidate(row, mincov):
        flags |= F_INSCAND
    if cov > mincov:
        flags |= F_COVGT
    amb, ch = ambiguity(rk, cov)
    if amb:
        flags |= F_AMBIG

    def cased(nuc, cnt):
        return nuc.lower() if cnt < mincov else nuc.upper()

    if flags & F_LOWCOV:
        plain = "N"
    elif include_ambig and amb:
        plain = ch
    else:
        plain = cased(*rk[0])
    alt = cased(*rk[1])
    return plain, alt, flags


def call_records(counts, mincov, include_ambig):
    L = len(counts)
    plain = np.empty(L, np.uint8)
    alt = np.empty(L, np.uint8)
    flags = np.empty(L, np.uint8)
    for i in range(L):
        p, a, f = call_record(counts[i], mincov, include_ambig)
        plain[i], alt[i], flags[i] = ord(p), ord(a), f
    return plain, alt, flags


# --------------------------------------------------------------------------- inserts
_TOKEN_RE = re.compile(r"(\d)([a-zA-Z]+)")


def modal_insert(tokens):
    """Events.py:68-82 — modal upper-cased token (first-seen tie-break) -> (bases, size_str)."""
    if not tokens:
        return None, None
    top = Counter(t.upper() for t in tokens).most_common(1)[0][0]
    m = _TOKEN_RE.search(top)
    if not m:
        return None, None
    return m.group(2), m.group(1)


def list_inserts(counts, mincov, tokens_at):
    """Events.py:5-44.  tokens_at(pos1) -> token list of the default-argument region pileup."""
    found = {}
    for i in range(len(counts)):
        if insert_candidate(counts[i], mincov):
            bases, size = modal_insert(tokens_at(i + 1))
            if bases is None or size is None:
                continue
            found[i + 1] = {size: bases}
    return (True, found) if found else (False, None)


def _match_positions(reads, i):
    """{reference position: query index} over the M / = / X bases of read i (what htslib's cigar_iref2iseq_* walk)."""
    out = {}
    x, y = int(reads["pos"][i]), 0
    for op, l in read_cigar(reads, i):
        if op in _MATCH_OPS:
            for j in range(l):
                out[x + j] = y + j
        if op in _REF_OPS:
            x += l
        if op in _QRY_OPS:
            y += l
    return out


def _read_name(reads, i):
    if reads.get("names") is None or reads.get("name_off") is None:
        return None
    a, b = int(reads["name_off"][i]), int(reads["name_off"][i + 1])
    return bytes(bytearray(reads["names"][a:b]))


def region_tokens(reads, pos1, min_base_quality=13, flag_filter=DEFAULT_FILTER,
                  ignore_orphans=True, max_depth=8000, ignore_overlaps=True):
    """Tokens pysam's default-argument region pileup yields for column pos1-1 (Events.py:63-67; PARITY
    UNPINNED: restated from htslib 1.21's bam_plp_push / overlap_push / tweak_overlap_quality and pysam's
    pileup_base_qual_skip, neither installed here).
      * fed to the engine: the reads of the region fetch (those overlapping the column) that pass the samtools stepper
        (flag filter, orphans), in file order;
      * max_depth: a read that starts where the read before it started is dropped while the buffer holds max_depth nodes
        (the list's sentinel counts as one);
      * ignore_overlaps: when the second of two properly paired mates arrives, every reference position where both have a
        matched base is tweaked in BOTH quality arrays (agreeing bases: first += second, at most 200, second = 0; differing:
        the higher one x 0.8, the other 0) — applied here to the whole overlap, as htslib does;
      * then every entry whose quality (at its query position; for deletion / ref-skip entries that of the next base) is
        below min_base_quality is skipped."""
    c = pos1 - 1
    n = int(reads["n_reads"])
    fed = []
    for i in range(n):
        if not read_piles_up(reads, i):
            continue
        f = int(reads["flag"][i])
        if f & flag_filter:
            continue
        if ignore_orphans and (f & FLAG_PAIRED) and not (f & FLAG_PROPER):
            continue
        p = int(reads["pos"][i])
        if p <= c < p + ref_length(read_cigar(reads, i)):
            fed.append(i)
    admitted, engine = [], None
    for i in fed:
        p = int(reads["pos"][i])
        if max_depth and p == engine and len(admitted) + 1 > max_depth:
            continue
        admitted.append(i)
        engine = p
    quals = {}
    have_q = reads.get("qual") is not None and "qual_off" in reads

    def q_of(i):
        if i not in quals:
            o, l = int(reads["qual_off"][i]), int(reads["l_qseq"][i])
            quals[i] = [int(v) for v in reads["qual"][o:o + l]]
        return quals[i]

    if ignore_overlaps and have_q and reads.get("names") is not None:
        waiting = {}
        for i in admitted:
            f = int(reads["flag"][i])
            if (f & 0x8) or not (f & FLAG_PROPER):
                continue
            p, l = int(reads["pos"][i]), int(reads["l_qseq"][i])
            end = p + ref_length(read_cigar(reads, i))
            mtid = int(reads["next_tid"][i]) if reads.get("next_tid") is not None else -1
            mpos = int(reads["next_pos"][i]) if reads.get("next_pos") is not None else -1
            isize = int(reads["tlen"][i]) if reads.get("tlen") is not None else 0
            tid = int(reads["tid"][i]) if reads.get("tid") is not None else 0
            if (mtid >= 0 and tid != mtid) or (abs(isize) >= 2 * l and mpos >= end):
                continue
            name = _read_name(reads, i)
            if name not in waiting:
                if mpos >= p or ((f & FLAG_PAIRED) and mpos == -1):
                    waiting[name] = i
                continue
            a = waiting.pop(name)
            ma, mb = _match_positions(reads, a), _match_positions(reads, i)
            qa, qb = q_of(a), q_of(i)
            for rp in sorted(set(ma) & set(mb)):
                ia, ib = ma[rp], mb[rp]
                if ia >= len(qa) or ib >= len(qb):
                    continue
                if read_base(reads, a, ia) == read_base(reads, i, ib):
                    qa[ia] = min(200, qa[ia] + qb[ib])
                    qb[ib] = 0
                elif qa[ia] >= qb[ib]:
                    qa[ia] = int(0.8 * qa[ia])
                    qb[ib] = 0
                else:
                    qb[ib] = int(0.8 * qb[ib])
                    qa[ia] = 0
    out = []
    for i in admitted:
        for col, tok, q in read_tokens(reads, i, with_qual=True, qual_override=quals.get(i)):
            if col == c:
                if q >= min_base_quality:
                    out.append(tok)
                break
    return out


# --------------------------------------------------------------------------- stage B, walk
def _in_orf(p, orfs):
    """ORFs.py:1-26 — half-open range over every GFF row."""
    return any(o["start"] <= p < o["end"] for o in orfs)


def _triplet_ok(n_up, n_min):
    """ORFs.py:45-77."""
    return (n_min % 3 == 0) if n_up % 3 == 0 else ((n_min + n_up) % 3 == 0)


def _correct_gff(orig, cur, cons, p, inserts, mincov, cov):
    """ORFs.py:111-192 (literal: re-joins the consensus for every active ORF)."""
    if inserts is not None and p in inserts and cov > mincov:
        shift = int(next(iter(inserts[p])))
        for o in cur:
            if o["start"] > p:                     # ORFs.py:80-108
                o["start"] += shift
    joined = None
    for o, o0 in zip(cur, orig):
        if not (o["start"] <= p < o["end"]) or o["strand"] != "+":
            continue
        if joined is None:
            joined = "".join(cons)
        tail = joined[o["start"] - 1:]
        gaps = tail.count("-")
        if cons[-1] == "-":
            o["end"] = o0["end"]
            continue
        bare = tail.replace("-", "")
        it, hit = 0, False
        for s in range(0, len(bare), 3):
            it += 1
            if bare[s:s + 3] in ("TAG", "TAA", "TGA"):
                hit = True
                break
        newend = o["start"] + 3 * it + gaps - 1 + (0 if hit else 1)
        if p == newend:
            o["end"] = newend


def build_consensus(mincov, counts, orfs, include_ambig, inserts, include_ins):
    """Sequences.py:168-322.  counts: [L,7]; orfs: [{'start','end','strand'}, ...];
    inserts: {pos1: {size_str: bases}} or None (result of list_inserts).
    Returns (consensus, corrected orfs).  Raises KeyError like the reference when a
    deletion walk runs past the last position (Sequences.py:47)."""
    L = len(counts)
    rk = [None] * (L + 2)

    def rank(p):
        if p < 1 or p > L:
            raise KeyError(p)
        if rk[p] is None:
            rk[p] = ranked(counts[p - 1])
        return rk[p]

    def run_after(p):                              # Sequences.py:44-53
        q = p + 1
        while rank(q)[0][0] == "X":
            q += 1
        return q - p - 1

    cur = [dict(o) for o in orfs]
    orig = [dict(o) for o in orfs]
    cons = []
    skip = set()
    for p in range(1, L + 1):
        row = counts[p - 1]
        cov = int(row[0])
        inside = _in_orf(p, cur)
        if p in skip:
            cons.append("-")
            _correct_gff(orig, cur, cons, p, inserts, mincov, cov)
            continue
        if cov < mincov:
            cons.append("N")
            _correct_gff(orig, cur, cons, p, inserts, mincov, cov)
            continue
        r = rank(p)
        amb, amb_ch = ambiguity(r, cov)

        def plain_char(nuc, cnt):
            if include_ambig and amb:
                return amb_ch
            return nuc.lower() if cnt < mincov else nuc.upper()

        if r[0][0] != "X":
            group = None
            if minority_del(row):
                n_up = run_after(p)
                if n_up:
                    if _triplet_ok(n_up, 1):
                        group = range(p, p + 1 + n_up)
                elif minority_del(counts[p]) if p < L else rank(p + 1):
                    n_up2 = run_after(p + 1)
                    if n_up2 and _triplet_ok(n_up2, 2):
                        group = range(p, p + 2 + n_up2)
            if group is not None:
                cons.append("-")
                skip.update(group)
            else:
                cons.append(plain_char(*r[0]))
        else:
            if inside:
                n_up = run_after(p)
                if n_up >= 2:
                    cons.append("-")
                    skip.update(range(p, p + 1 + n_up))
                else:
                    cons.append(plain_char(*r[1]))
            else:
                cons.append("-")
        if include_ins and cov > mincov and inserts and p in inserts:
            for size in inserts[p]:
                cons.append(str(inserts[p][size]))
        _correct_gff(orig, cur, cons, p, inserts, mincov, cov)
    return "".join(cons), cur


# --------------------------------------------------------------------------- writers
def fasta_text(name, mincov, consensus):
    """Outputs.py:182-183."""
    return ">%s mincov=%s\n%s\n" % (name, mincov, consensus)


def coverage_tsv(counts):
    """Coverage.py:1-16."""
    return "".join("%d\t%d\n" % (i + 1, int(counts[i][0])) for i in range(len(counts)))


VCF_HEADER = ("##fileformat=VCFv4.3\n##fileDate={today}\n##source='TrueConsense {argv}'\n"
              "##reference='{ref}'\n##contig=<ID={refid}>\n"
              '##INFO=<ID=DP,Number=1,Type=Integer,Description="Read Depth">\n'
              '##INFO=<ID=INDEL,Number=0,Type=Flag,Description="Indicates that the variant is an INDEL.">\n'
              "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")


def vcf_records(refid, refseq, cons_noins, counts, mincov, inserts):
    """Outputs.py:131-180 — the record lines (header handled by the caller)."""
    ref = list(refseq)
    seq = list(cons_noins.upper())
    cov = lambda p1: int(counts[p1 - 1][0])        # noqa: E731  (KeyError -> IndexError)
    out = []
    skipped = set()
    for i in range(len(ref)):
        if i in skipped:
            continue
        if ref[i] != seq[i]:
            if seq[i] == "-":
                b = i
                gone = []
                while seq[b] == "-":
                    gone.append(ref[b])
                    skipped.add(b)
                    b += 1
                out.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d;INDEL\n" % (
                    refid, i, ref[i - 1] + "".join(gone), seq[i - 1], cov(i + 1)))
            else:
                p = 1 if i < 2 else i
                out.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d\n" % (
                    refid, i + 1, ref[i], seq[i], cov(p + 1)))
        if inserts:
            for lp in inserts:
                if i == lp:
                    c = cov(i + 1)
                    if c > mincov:
                        for size in inserts[lp]:
                            out.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%d;INDEL\n" % (
                                refid, i, ref[i], seq[i] + str(inserts[lp][size]), c))
    return "".join(out)
