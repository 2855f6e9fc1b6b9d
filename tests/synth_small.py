"""
Small seeded inputs shared by tests/golden/make_golden.py (which feeds them to the real
reference) and the parity tests (which feed them to the oracle and to the HIP path).
Test infrastructure; nothing here is imported by the product.
"""
from __future__ import annotations

import re

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
_CIG_RE = re.compile(r"(\d+)([MIDNSHP=X])")
_OPS = "MIDNSHP=X"
STOPS = ("TAG", "TAA", "TGA")


# ----------------------------------------------------------------------------- reads
def parse_cigar(text):
    return [(_OPS.index(op), int(n)) for n, op in _CIG_RE.findall(text)]


def reads_from_spec(spec):
    """spec['reads'] = [{'pos','flag','cigar','seq','qual'(opt int or list),'tid'(opt),'name','mtid','mpos','tlen'(opt)}]
    -> dict of flat arrays in the tcmi_reads layout (include/tcmi.h)."""
    rs = spec["reads"]
    n = len(rs)
    pos = np.zeros(n, np.int32)
    flag = np.zeros(n, np.uint16)
    lq = np.zeros(n, np.int32)
    tid = np.zeros(n, np.int32)
    cig_off = np.zeros(n + 1, np.uint64)
    seq_off = np.zeros(n + 1, np.uint64)
    qual_off = np.zeros(n + 1, np.uint64)
    cig, seq, qual = [], bytearray(), bytearray()
    for i, r in enumerate(rs):
        pos[i], flag[i], tid[i] = r["pos"], r["flag"], r.get("tid", 0)
        ops = parse_cigar(r["cigar"])
        cig.extend((l << 4) | op for op, l in ops)
        cig_off[i + 1] = len(cig)
        s = r["seq"]
        s = "" if s == "*" else s
        lq[i] = len(s)
        codes = [NT16.index(c) for c in s.upper()]
        if len(codes) & 1:
            codes.append(0)
        seq.extend((codes[k] << 4) | codes[k + 1] for k in range(0, len(codes), 2))
        seq_off[i + 1] = len(seq)
        q = r.get("qual", 30)
        qual.extend([q] * len(s) if isinstance(q, int) else q)
        qual_off[i + 1] = len(qual)
    out = {"n_reads": n, "pos": pos, "flag": flag, "l_qseq": lq, "tid": tid,
           "cigar_off": cig_off, "cigar": np.array(cig, np.uint32),
           "seq_off": seq_off, "seq": np.frombuffer(bytes(seq), np.uint8).copy(),
           "qual_off": qual_off, "qual": np.frombuffer(bytes(qual), np.uint8).copy()}
    if any("name" in r for r in rs):                # mate fields and names (optional in tcmi_reads)
        names = [str(r.get("name", "r%d" % i)).encode() for i, r in enumerate(rs)]
        out["name_off"] = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.uint64)
        out["names"] = np.frombuffer(b"".join(names) or b"\0", np.uint8).copy()
        out["next_tid"] = np.array([r.get("mtid", -1) for r in rs], np.int32)
        out["next_pos"] = np.array([r.get("mpos", -1) for r in rs], np.int32)
        out["tlen"] = np.array([r.get("tlen", 0) for r in rs], np.int32)
    return out


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[int(k)] for k in rng.integers(0, len(alphabet), n))


def _scrub_stops(seq, start, end):
    """make [start, end] (1-based inclusive) free of in-frame stops, ending in TAA."""
    s = list(seq)
    for p in range(start - 1, end - 3, 3):
        if "".join(s[p:p + 3]) in STOPS:
            s[p] = "C"
    if end - start + 1 >= 6:
        s[end - 3:end] = list("TAA")
    return "".join(s)


def _mutate(rng, s, rate):
    out = list(s)
    for k in range(len(out)):
        if rng.random() < rate:
            out[k] = "ACGT"[int(rng.integers(0, 4))]
    return "".join(out)


def read_specs(rng):
    """Small read sets covering the CIGAR / flag / SEQ edge cases of SURVEY §8-P."""
    specs = []
    for case in range(14):
        L = int(rng.integers(70, 160))
        ref = _rand_seq(rng, L)
        o1s = int(rng.integers(4, 12))
        o1e = o1s + 3 * int(rng.integers(8, (L - o1s) // 3 - 2)) - 1
        ref = _scrub_stops(ref, o1s, o1e)
        orfs = [{"start": o1s, "end": o1e, "strand": "+"}]
        if case % 3 == 1 and o1e + 20 < L:
            orfs.append({"start": o1e - 3, "end": L - 2, "strand": "-" if case % 2 else "+"})
        mincov = [5, 10, 3, 8][case % 4]
        depth = int(rng.integers(12, 40))
        reads = []
        # planted sites
        del_at = int(rng.integers(o1s + 6, o1s + 20))
        del_len = [3, 1, 2, 6, 3][case % 5]
        del_frac = [0.9, 0.2, 0.95, 0.6, 0.17][case % 5]
        ins_at = int(rng.integers(o1s + 24, max(o1s + 26, min(L - 30, o1e - 4))))
        ins_seq = ["G", "GG", "ACG", "ACGTACGTACGT", "T"][case % 5]
        ins_frac = [0.9, 0.6, 0.56, 0.7, 0.5][case % 5]
        snp_at = int(rng.integers(2, L - 2))
        n_reads = depth * L // 30
        for _ in range(n_reads):
            rl = int(rng.integers(18, 46))
            p = int(rng.integers(-3, L - 10))
            p = max(0, p)
            rl = min(rl, L - p + (2 if rng.random() < 0.05 else 0))   # a few run past the end
            if rl < 4:
                continue
            frag = (ref + "ACGT")[p:p + rl]
            frag = _mutate(rng, frag, 0.01)
            flag = 16 if rng.random() < 0.5 else 0
            cigar = "%dM" % rl
            seq = frag
            q = 30
            lo, hi = p + 1, p + rl                       # 1-based span
            if lo < snp_at <= hi and rng.random() < 0.5:
                k = snp_at - lo
                seq = seq[:k] + ("T" if seq[k] != "T" else "A") + seq[k + 1:]
            r = rng.random()
            if lo + 2 < del_at and del_at + del_len + 2 < hi and r < del_frac:
                a = del_at - lo
                cigar = "%dM%dD%dM" % (a, del_len, rl - a - del_len)
                seq = seq[:a] + seq[a + del_len:]
            elif lo + 2 < ins_at < hi - 2 and rng.random() < ins_frac:
                a = ins_at - lo + 1
                cigar = "%dM%dI%dM" % (a, len(ins_seq), rl - a)
                seq = seq[:a] + ins_seq + seq[a:]
            elif r > 0.97:
                a = max(2, rl // 3)
                cigar = "%dM5N%dM" % (a, rl - a)
            elif r > 0.94:
                cigar = "3S%dM2S" % rl
                seq = "AAA" + seq + "CC"
            elif r > 0.92:
                cigar = "2H%dM" % rl
            elif r > 0.90:
                a = max(2, rl // 2)
                cigar = "%dM1P1I%dM" % (a, rl - a)
                seq = seq[:a] + "C" + seq[a:]
            elif r > 0.88:
                cigar = "%d=%dX" % (rl - 3, 3)
            elif r > 0.87:
                a = max(2, rl // 2)
                cigar = "%dM2D1I%dM" % (a, max(1, rl - a - 2))     # D directly followed by I
                seq = seq[:a] + "G" + seq[a + 2:a + 2 + max(1, rl - a - 2)]
            elif r > 0.86:
                cigar = "1I%dM2I" % rl                              # leading / trailing insertion
                seq = "T" + seq + "GA"
            elif r > 0.85:
                seq = "*"                                            # SEQ absent
            elif r > 0.84:
                seq = seq[:2] + "N" + seq[3:5] + "R=" + seq[7:]
            fl = flag
            rr = rng.random()
            if rr > 0.985:
                fl |= 4                                              # unmapped: never piles up
            elif rr > 0.97:
                fl |= 0x100                                          # secondary: counted in stage A
            elif rr > 0.955:
                fl |= 0x400
            elif rr > 0.94:
                fl |= 0x1                                            # paired, not proper (orphan)
            elif rr > 0.92:
                fl |= 0x3
            if rng.random() < 0.08:
                q = int(rng.integers(0, 20))
            reads.append({"pos": p, "flag": fl, "cigar": cigar, "seq": seq, "qual": q})
        if case == 0:
            reads.append({"pos": 5, "flag": 0, "cigar": "*", "seq": "ACGT", "qual": 30})  # no CIGAR
            reads.append({"pos": 6, "flag": 0, "cigar": "4S", "seq": "ACGT", "qual": 30})
            reads.append({"pos": 7, "flag": 0, "cigar": "10M", "seq": "ACGTACGTAC", "qual": 30,
                          "tid": -1})
        reads.sort(key=lambda r: r["pos"])
        specs.append({"name": "reads%02d" % case, "ref": ref, "orfs": orfs, "mincov": mincov,
                      "reads": reads})
    return specs


# ----------------------------------------------------------------------------- count matrices
_COL = {"A": 1, "T": 2, "C": 3, "G": 4, "X": 5, "I": 6}


def counts_from_seq(seq, cov=100):
    m = np.zeros((len(seq), 7), np.int64)
    m[:, 0] = cov
    for i, c in enumerate(seq):
        m[i, _COL[c]] = cov
    return m


def _set(m, pos1, cov=None, **kv):
    r = m[pos1 - 1]
    r[1:] = 0
    for k, v in kv.items():
        r[_COL[k]] = v
    if cov is not None:
        r[0] = cov


TOY38 = "ACGATGAAACCCGGGTTTAAACCCGGGTAAACGTACGT"
TOY40 = "ACGATGAAACCCGGGTTTAAACCCGGGTAAACGTACGTAC"


def _toy_cases():
    """SURVEY Appendix A, as count matrices."""
    orf = [{"start": 4, "end": 30, "strand": "+"}]
    two = orf + [{"start": 32, "end": 40, "strand": "+"}]
    ins12 = ["C+12ACGTACGTACGT"] * 55 + ["c+12acgtacgtacgt"] * 5 + ["C"] * 40

    def case(name, seq, orfs, edits, region=None, mincov=30):
        m = counts_from_seq(seq)
        for e in edits:
            e(m)
        return {"name": name, "counts": m, "orfs": [dict(o) for o in orfs],
                "region": region or {}, "mincov": mincov}

    def ed(pos, cov=None, **kv):
        return lambda m: _set(m, pos, cov, **kv)

    def raw(pos, col, val):
        def f(m):
            m[pos - 1, col] = val
        return f

    out = [
        case("toy_plain", TOY38, orf, []),
        case("toy_del3_and_single", TOY38, orf, [ed(7, A=10, X=90), ed(8, A=10, X=90), ed(9, A=10, X=90),
                                                  ed(12, C=20, X=80)]),
        case("toy_mindel_accept", TOY38, orf, [ed(7, A=80, X=20), ed(8, A=10, X=90), ed(9, A=10, X=90)]),
        case("toy_mindel_reject", TOY38, orf, [ed(7, A=80, X=20), ed(8, A=10, X=90), ed(9, A=10, X=90),
                                                ed(10, C=10, X=90)]),
        case("toy_mindel_pair", TOY38, orf, [ed(7, A=80, X=20), ed(8, A=80, X=20), ed(9, A=10, X=90)]),
        case("toy_mindel_pair_reject", TOY38, orf, [ed(7, A=80, X=20), ed(8, A=80, X=20), ed(9, A=10, X=90),
                                                     ed(10, C=5, X=95)]),
        case("toy_single_del_inorf", TOY38, orf, [ed(7, A=5, X=95)]),
        case("toy_single_del_outside", TOY38, orf, [ed(2, C=5, X=95), ed(34, T=40, X=60)]),
        case("toy_lowercase", TOY38, orf, [ed(6, G=25, A=24, T=24, C=27 - 27, cov=100)]),
        case("toy_end_keyerror", TOY38, orf, [ed(38, T=80, X=20)]),
        case("toy_end_xrun_inorf", TOY38, [{"start": 4, "end": 40, "strand": "+"}],
             [ed(37, X=90, G=10), ed(38, X=90, T=10)]),
        case("toy_end_xrun_outside", TOY38, orf, [ed(37, X=90, G=10), ed(38, X=90, T=10)]),
        case("toy_insert", TOY38, orf, [raw(10, 6, 60)], {10: ins12}),
        case("toy_insert_cov_eq_mincov", TOY38, orf, [ed(10, cov=30, C=30), raw(10, 6, 30)], {10: ins12}),
        case("toy_premature_stop", TOY38[:12] + "TAA" + TOY38[15:], orf, []),
        case("toy_premature_stop_lower", TOY38[:12] + "TAA" + TOY38[15:], orf,
             [ed(13, T=25, cov=100)]),
        case("toy_stop_lost", TOY38[:28] + "C" + TOY38[29:], orf, []),
        case("toy_insert_before_orfs_1", TOY40, two, [raw(2, 6, 60)], {2: ["C+1T"] * 60 + ["C"] * 40}),
        case("toy_insert_before_orfs_12", TOY40, two, [raw(2, 6, 60)], {2: ins12}),
        case("toy_two_runs_diverge", TOY40, orf, [raw(2, 6, 60), ed(12, C=40, X=60)],
             {2: ["C+1T"] * 60 + ["C"] * 40}),
        case("toy_del_modal_token", TOY40, orf, [raw(9, 6, 70)], {9: ["A-2NN"] * 3 + ["A+1T"] * 2}),
        case("toy_insert_no_digits", TOY40, orf, [raw(9, 6, 70)], {9: ["A", "A", "C"]}),
        case("toy_insert_empty_pileup", TOY40, orf, [raw(9, 6, 70)], {}),
        case("toy_zero_cov_gap", TOY40, orf, [ed(k, cov=0) for k in range(18, 23)], mincov=0),
        case("toy_zero_cov_gap_mindel", TOY40, orf, [ed(17, T=80, X=20)] + [ed(k, cov=0) for k in range(18, 21)],
             mincov=1),
        case("toy_minus_strand", TOY40, [{"start": 4, "end": 30, "strand": "-"}],
             [raw(2, 6, 60), ed(12, C=40, X=60)], {2: ["C+1T"] * 60 + ["C"] * 40}),
        case("toy_writers", TOY40, orf,
             [ed(1, G=100), ed(5, C=100), ed(20, cov=10, A=10), ed(7, A=5, X=95), ed(8, A=5, X=95),
              ed(9, A=5, X=95), ed(12, cov=101, C=101), raw(12, 6, 70), ed(35, A=50, G=50)],
             {12: ["C+2GG"] * 70 + ["C"] * 31}),
    ]
    return out


def consensus_specs(rng):
    specs = _toy_cases()
    for k in range(150):
        L = int(rng.integers(45, 150))
        seq = _rand_seq(rng, L)
        n_orf = int(rng.integers(1, 4))
        orfs = []
        for _ in range(n_orf):
            s = int(rng.integers(2, L - 25))
            e = min(L - int(rng.integers(0, 6)), s + 3 * int(rng.integers(5, 30)) - 1)
            if rng.random() < 0.7:
                seq = _scrub_stops(seq, s, e)
            orfs.append({"start": s, "end": e, "strand": "+" if rng.random() < 0.8 else "-"})
        mincov = int(rng.choice([1, 10, 30, 30, 50]))
        cov = int(rng.choice([40, 100, 100, 333]))
        m = counts_from_seq(seq, cov)
        region = {}
        n_events = int(rng.integers(1, 9))
        for _ in range(n_events):
            p = int(rng.integers(1, L + 1))
            kind = int(rng.integers(0, 12))
            base = seq[p - 1]
            others = [b for b in "ATCG" if b != base]
            if kind == 0:                                   # deletion run
                n = int(rng.integers(1, 8))
                fx = float(rng.choice([0.95, 0.7, 0.55]))
                for q in range(p, min(L, p + n - 1) + 1):
                    x = int(cov * fx)
                    _set(m, q, cov, **{seq[q - 1]: cov - x, "X": x})
            elif kind == 1:                                 # minority deletion before a run
                n = int(rng.integers(1, 6))
                x = int(cov * float(rng.choice([0.15, 0.2, 0.3, 0.14])))
                _set(m, p, cov, **{base: cov - x, "X": x})
                if rng.random() < 0.4 and p + 1 <= L:
                    _set(m, p + 1, cov, **{seq[p]: cov - x, "X": x})
                    p += 1
                for q in range(p + 1, min(L, p + n) + 1):
                    _set(m, q, cov, **{seq[q - 1]: cov // 10, "X": cov - cov // 10})
            elif kind == 2:                                 # two-way ambiguity
                a = int(cov * float(rng.choice([0.5, 0.55, 0.52, 0.45])))
                _set(m, p, cov, **{base: a, others[0]: cov - a})
            elif kind == 3:                                 # three / four-way
                if rng.random() < 0.5:
                    a, b = cov // 3, cov // 3
                    _set(m, p, cov, **{base: cov - a - b, others[0]: a, others[1]: b})
                else:
                    q4 = cov // 4
                    _set(m, p, cov, **{base: cov - 3 * q4, others[0]: q4, others[1]: q4, others[2]: q4})
            elif kind == 4:                                 # low / zero coverage stretch
                n = int(rng.integers(1, 6))
                c = int(rng.choice([0, 0, max(0, mincov - 1), mincov]))
                for q in range(p, min(L, p + n - 1) + 1):
                    _set(m, q, c, **{seq[q - 1]: c})
            elif kind == 5:                                 # lower-case primary
                a = max(1, min(cov, mincov) - 1)
                rest = cov - a
                _set(m, p, cov, **{base: a, others[0]: min(a - 1, rest) if a > 1 else 0})
            elif kind == 6:                                 # SNP
                _set(m, p, cov, **{others[1]: cov})
            elif kind in (7, 8):                            # insert
                frac = float(rng.choice([0.9, 0.6, 0.56, 0.55, 0.5]))
                n_ins = int(round(cov * frac))
                m[p - 1, 6] = n_ins
                ins = _rand_seq(rng, int(rng.choice([1, 1, 2, 3, 4, 11, 12])))
                region[p] = ["%s+%d%s" % (base, len(ins), ins)] * n_ins + [base] * (cov - n_ins)
                if rng.random() < 0.15:
                    m[p - 1, 0] = mincov                    # cov == mincov: called, not spliced
            elif kind == 9:                                 # premature stop inside the first ORF
                o = orfs[0]
                if o["end"] - o["start"] > 12:
                    c0 = o["start"] + 3 * int(rng.integers(1, (o["end"] - o["start"]) // 3 - 1))
                    for j, ch in enumerate(rng.choice(STOPS)):
                        _set(m, c0 + j, cov, **{ch: cov})
            elif kind == 10:                                # X primary with strong secondary
                x = int(cov * 0.6)
                _set(m, p, cov, **{base: cov - x, "X": x})
            else:                                           # X / base exact tie
                _set(m, p, cov, **{base: cov // 2, "X": cov // 2})
        specs.append({"name": "rand%03d" % k, "counts": m, "orfs": orfs, "region": region,
                      "mincov": mincov})
    return specs
