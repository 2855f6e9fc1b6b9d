"""Seeded synthetic inputs of the BASELINE.json shapes: a 29 903-bp reference with
the SARS-CoV-2 CDS layout and coordinate-sorted 150-bp reads, produced directly as the flat
read arrays of the tcmi_reads layout (and optionally written out as a BAM by io.bamwriter).
There is no network for real data; everything here is random with a fixed seed.
"""
from __future__ import annotations

import numpy as np

L_SARS2 = 29903
# CDS coordinates of MN908947.3 (1-based inclusive), all '+' strand
SARS2_CDS = [(266, 13483), (21563, 25384), (25393, 26220), (26245, 26472), (26523, 27191), (27202, 27387),
             (27394, 27759), (27756, 27887), (27894, 28259), (28274, 29533), (29558, 29674)]
_CODE = np.array([1, 2, 4, 8], np.uint8)          # BAM 4-bit codes of A C G T
_LETTER = np.frombuffer(b"ACGT", np.uint8)
_STOPS = {"TAG", "TAA", "TGA"}


def make_reference(seed=20251121, L=L_SARS2, cds=SARS2_CDS):
    """-> (reference str, GFF rows [{'start','end','strand'}]); CDSs are free of in-frame stops
    and end in TAA."""
    rng = np.random.default_rng(seed)
    s = _LETTER[rng.integers(0, 4, L)].copy()
    orfs = []
    for a, b in cds:
        if b > L:
            continue
        for p in range(a - 1, b - 3, 3):
            if bytes(s[p:p + 3]).decode() in _STOPS:
                s[p] = ord("C")
        s[b - 3:b] = np.frombuffer(b"TAA", np.uint8)
        orfs.append({"start": a, "end": b, "strand": "+"})
    return bytes(s).decode(), orfs


def gff_text(orfs, seqid="MN908947.3"):
    lines = ["##gff-version 3\n"]
    body = ["%s\tsynthetic\tCDS\t%d\t%d\t.\t%s\t0\tID=cds%d;Name=orf%d\n" % (seqid, o["start"], o["end"], o["strand"], k, k)
            for k, o in enumerate(orfs)]
    return "".join(lines), "".join(body)


def make_reads(ref, n_reads, read_len=150, seed=1, sub_rate=0.005, planted=True, indel_sites=None, start_range=None):
    """Coordinate-sorted reads, CIGAR `<read_len>M` except carriers of `indel_sites`.

    indel_sites: list of (pos1, kind, payload, fraction): kind 'I' with payload = inserted bases
    placed after 1-based reference position pos1; kind 'D' with payload = number of deleted
    bases starting at pos1 + 1.  `fraction` of the reads covering pos1 carry it (drawn from those spanning
    the site with >= 5 bases either side; a read may carry several sites).
    start_range: (lo, hi) restricts the 0-based read starts to [lo, hi) — one genome tile of a BAM that is
    split over several GPUs.
    Returns the dict of flat arrays (tcmi_reads layout) with constant quality 30.
    """
    rng = np.random.default_rng(seed)
    L = len(ref)
    refc = np.zeros(L + 64, np.uint8)
    lut = np.zeros(256, np.uint8)
    for ch, code in zip(b"ACGT", _CODE):
        lut[ch] = code
    refc[:L] = lut[np.frombuffer(ref.encode(), np.uint8)]
    s_lo, s_hi = (0, L - read_len + 1) if start_range is None else (max(0, start_range[0]), min(L - read_len + 1, start_range[1]))
    starts = np.sort(rng.integers(s_lo, s_hi, n_reads)).astype(np.int32)
    flag = (rng.integers(0, 2, n_reads) * 16).astype(np.uint16)
    nb = (read_len + 1) // 2
    seq = np.empty((n_reads, nb), np.uint8)
    planted_cols = {}
    if planted:                                     # columns with fixed splits to exercise Ambig.py
        for k, split in enumerate(([50, 50], [55, 45], [45, 35, 20], [34, 33, 33], [25, 25, 25, 25])):
            planted_cols[1000 + 1500 * k] = np.array(split) / 100.0
    chunk = 65536
    ar = np.arange(read_len, dtype=np.int64)
    for c0 in range(0, n_reads, chunk):
        st = starts[c0:c0 + chunk].astype(np.int64)
        codes = refc[st[:, None] + ar[None, :]]
        sub = rng.random(codes.shape) < sub_rate
        codes[sub] = _CODE[rng.integers(0, 4, int(sub.sum()))]
        for col, probs in planted_cols.items():
            off = col - st
            hit = np.nonzero((off >= 0) & (off < read_len))[0]
            if len(hit):
                base0 = int(np.log2(refc[col]))
                pick = rng.choice(len(probs), len(hit), p=probs)
                codes[hit, off[hit]] = _CODE[(base0 + pick) % 4]
        if read_len & 1:
            codes = np.concatenate([codes, np.zeros((len(st), 1), np.uint8)], axis=1)
        seq[c0:c0 + chunk] = (codes[:, 0::2] << 4) | codes[:, 1::2]
    n_cig = np.ones(n_reads, np.int64)
    cig_first = np.full(n_reads, (read_len << 4) | 0, np.uint32)
    extra = {}                                       # read index -> list of cigar words
    max_span = read_len
    if indel_sites:
        events = {}                                  # read index -> [(pos1, kind, payload)] in site order
        for pos1, kind, payload, frac in sorted(indel_sites, key=lambda t: t[0]):
            lo = np.searchsorted(starts, pos1 - read_len + 5, side="left")
            hi = np.searchsorted(starts, pos1 - 5, side="right")
            n_cover = int(np.searchsorted(starts, pos1 - 1, side="right") - np.searchsorted(starts, pos1 - read_len, side="left"))
            want = min(int(hi - lo), int(round(frac * n_cover)))      # `fraction` of the reads covering pos1
            for i in (lo + rng.choice(int(hi - lo), want, replace=False)).tolist() if want > 0 else []:
                events.setdefault(int(i), []).append((pos1, kind, payload))
        for i, evs in events.items():
            st = int(starts[i])
            cur, used, words, parts = st, 0, [], []  # reference cursor (0-based), read bases used so far
            for pos1, kind, payload in evs:
                m = pos1 - cur                       # matched bases up to and including pos1
                k = len(payload) if kind == "I" else int(payload)
                left = read_len - used - m - (k if kind == "I" else 0)
                if m < 5 or left < 5 or cur + m + (k if kind == "D" else 0) + left > L:
                    continue                         # the event does not fit this read (an earlier one moved it)
                parts.append(refc[cur:cur + m])
                words.append((m << 4) | 0)
                cur += m
                used += m
                if kind == "I":
                    parts.append(lut[np.frombuffer(payload.encode(), np.uint8)])
                    words.append((k << 4) | 1)
                    used += k
                else:
                    words.append((k << 4) | 2)
                    cur += k
            if not words:
                continue
            rest = read_len - used
            parts.append(refc[cur:cur + rest])
            words.append((rest << 4) | 0)
            max_span = max(max_span, cur + rest - st)
            codes = np.concatenate(parts).copy()
            sub = rng.random(read_len) < sub_rate
            codes[sub] = _CODE[rng.integers(0, 4, int(sub.sum()))]
            if read_len & 1:
                codes = np.concatenate([codes, np.zeros(1, np.uint8)])
            seq[i] = (codes[0::2] << 4) | codes[1::2]
            extra[i] = words
            n_cig[i] = len(words)
    cigar_off = np.zeros(n_reads + 1, np.uint64)
    cigar_off[1:] = np.cumsum(n_cig)
    cigar = np.empty(int(cigar_off[-1]), np.uint32)
    cigar[cigar_off[:-1].astype(np.int64)] = cig_first
    for i, words in extra.items():
        o = int(cigar_off[i])
        cigar[o:o + len(words)] = words
    seq_off = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(nb))
    return {"n_reads": n_reads, "pos": starts, "flag": flag, "l_qseq": np.full(n_reads, read_len, np.int32),
            "tid": np.zeros(n_reads, np.int32), "cigar_off": cigar_off, "cigar": cigar, "seq_off": seq_off,
            "seq": seq.reshape(-1), "qual": np.full(n_reads * read_len, 30, np.uint8),
            "qual_off": np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len),
            # reads are sorted by pos and none spans more reference than this (see tcmi_reads)
            "sorted_max_span": int(max_span)}


def default_indel_sites(orfs, seed=7):
    """cfg 3 of BASELINE.json: 1-3 bp (and one 12 bp) insertions / deletions at CDS boundaries
    with carrier fractions {10,15,20,50,55,56,60,90} %."""
    rng = np.random.default_rng(seed)
    fracs = [0.10, 0.15, 0.20, 0.50, 0.55, 0.56, 0.60, 0.90]
    order_i = [0, 7, 5, 2, 6, 4, 1, 3]            # insertions and deletions each run through all eight fractions,
    order_d = [7, 1, 3, 6, 0, 5, 2, 4]            # starting with a clear minority and a clear majority
    sites, k = [], 0
    for o in orfs:
        for edge in (o["start"] + 5, o["end"] - 7):
            if k % 2 == 0:
                f = fracs[order_i[(k // 2) % 8]]
                n = 12 if k == 4 else 1 + (k // 2) % 3
                payload = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
                sites.append((edge, "I", payload, f))
            else:
                f = fracs[order_d[(k // 2) % 8]]
                sites.append((edge, "D", 1 + (k // 2) % 3, f))
            k += 1
    return sites
