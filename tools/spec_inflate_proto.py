"""Prototype (CPU, pure Python) of the speculative 64-lane decode of one deflate block, the scheme of inflate_device.hip's
bgzf_symbols kernel: lane c starts decoding at bit s_c = start + c * chunk (lane 0 at the true start), notes the bit position of
every symbol it starts in a window of W bits behind s_c, and stops when a symbol of its own starts on a position that the
lane in front of it has noted (from there on the two decode the same).  Prints how many lock-step rounds pass A (find the
merge points) and pass B (decode the ranges for real) take, per block, for a few files.  A design tool, not product code.
usage: python tools/spec_inflate_proto.py [headline|hard] [n_reads]"""
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))

CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def table(lens):
    """code lengths -> {(len, code): sym} and a 15-bit peek table (bits LSB first) -> (sym, len)"""
    lens = list(lens)
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    code = 0
    nxt = [0] * 16
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    peek = {}
    for s, l in enumerate(lens):
        if l:
            c = nxt[l]
            nxt[l] += 1
            rev = int(format(c, "0%db" % l)[::-1], 2)
            peek[(l, rev)] = s
    return peek


class Stream:
    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.n = len(data) * 8

    def bits(self, pos, n):
        return (self.v >> pos) & ((1 << n) - 1)


def dec(st, pos, peek):
    """one Huffman symbol at pos -> (sym, newpos) or (None, pos)"""
    w = st.bits(pos, 15)
    for l in range(1, 16):
        s = peek.get((l, w & ((1 << l) - 1)))
        if s is not None:
            return s, pos + l
    return None, pos


def header(st, pos):
    """dynamic / fixed block header at pos -> (final, type, ll peek, d peek, pos of the first symbol)"""
    final = st.bits(pos, 1)
    typ = st.bits(pos + 1, 2)
    pos += 3
    if typ == 1:
        ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
        return final, typ, table(ll), table([5] * 30), pos
    assert typ == 2, typ
    nlen = st.bits(pos, 5) + 257
    ndist = st.bits(pos + 5, 5) + 1
    ncode = st.bits(pos + 10, 4) + 4
    pos += 14
    cl = [0] * 19
    for i in range(ncode):
        cl[CL_ORDER[i]] = st.bits(pos, 3)
        pos += 3
    clp = table(cl)
    lens = []
    while len(lens) < nlen + ndist:
        s, pos = dec(st, pos, clp)
        if s < 16:
            lens.append(s)
        elif s == 16:
            lens += [lens[-1]] * (3 + st.bits(pos, 2)); pos += 2
        elif s == 17:
            lens += [0] * (3 + st.bits(pos, 3)); pos += 3
        else:
            lens += [0] * (11 + st.bits(pos, 7)); pos += 7
    return final, typ, table(lens[:nlen]), table(lens[nlen:nlen + ndist]), pos


def symbol(st, pos, llp, dp, end):
    """one literal / match / end-of-block at pos -> (kind, newpos); kind 0 literal, 1 match, 2 eob, -1 invalid"""
    s, p = dec(st, pos, llp)
    if s is None or p > end:
        return -1, pos
    if s < 256:
        return 0, p
    if s == 256:
        return 2, p
    if s > 285:
        return -1, pos
    p += LEXT[s - 257]
    d, p = dec(st, p, dp)
    if d is None or d > 29:
        return -1, pos
    p += DEXT[d]
    if p > end:
        return -1, pos
    return 1, p


def simulate(st, start, end, llp, dp, lanes=64, W=256):
    """-> (true symbols, rounds of pass A, rounds of pass B, alive lanes)"""
    # the truth, for the check
    truth = []
    p = start
    while True:
        truth.append(p)
        k, p = symbol(st, p, llp, dp, end)
        assert k >= 0
        if k == 2:
            break
    true_set = set(truth)
    chunk = max(1, -(-(end - start) // lanes))
    s = [start + c * chunk for c in range(lanes)]
    pos = list(s)
    state = ["run" if s[c] < end else "idle" for c in range(lanes)]    # run / merged / eob / dead / idle
    marks = [set() for _ in range(lanes)]
    tgt = [c + 1 for c in range(lanes)]
    mpos = [None] * lanes
    total = [0] * lanes
    rounds_a = 0
    while any(x == "run" for x in state):
        rounds_a += 1
        # every running lane notes its position ...
        for c in range(lanes):
            if state[c] == "run" and pos[c] - s[c] < W:
                marks[c].add(pos[c])
        # ... then looks it up in the window of the lane in front, then decodes one symbol
        for c in range(lanes):
            if state[c] != "run":
                continue
            p = pos[c]
            while tgt[c] < lanes and p >= s[tgt[c]] + W:
                tgt[c] += 1
            t = tgt[c]
            if t < lanes and state[t] != "idle" and p >= s[t] and p in marks[t]:
                state[c] = "merged"
                mpos[c] = p
                continue
            k, np_ = symbol(st, p, llp, dp, end)
            if k < 0:
                state[c] = "dead"
                continue
            total[c] += 1
            pos[c] = np_
            if k == 2:
                state[c] = "eob"
    # the chain of lanes that hold the truth
    alive = []
    c, P = 0, start
    while True:
        before = sum(1 for m in marks[c] if m < P)
        # (symbols lane c decoded before P: all of them noted, P lies inside its window)
        if state[c] == "merged":
            cnt = total[c] - before
            alive.append((c, P, cnt))
            P = mpos[c]
            c = tgt[c]
        else:
            assert state[c] == "eob", (c, state[c])
            alive.append((c, P, total[c] - before))
            break
    n = sum(a[2] for a in alive)
    assert n == len(truth), (n, len(truth))
    for c, P, cnt in alive:
        assert P in true_set
    rounds_b = max(a[2] for a in alive)
    return len(truth), rounds_a, rounds_b, len(alive)


def blocks_of(path):
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        yield raw[off + 12 + xlen: off + bsize - 8]
        off += bsize


def main():
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    kind = sys.argv[1] if len(sys.argv) > 1 else "headline"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ref, orfs = sy.make_reference()
    reads = sy.make_reads(ref, n, seed=3)
    d = tempfile.mkdtemp()
    p = os.path.join(d, "x.bam")
    if kind == "hard":
        rng = np.random.default_rng(1)
        reads["qual"] = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=len(reads["qual"]), p=[0.02, 0.05, 0.13, 0.80])
        names = [("A00123:45:HXXXXX:%d:%d:%d:%d" % (rng.integers(1, 5), rng.integers(1101, 2679), rng.integers(1000, 33000), rng.integers(1000, 37000))).encode() for _ in range(n)]
        reads["name_off"] = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.uint64)
        reads["names"] = np.frombuffer(b"".join(names), np.uint8).copy()
        bamwriter.write_bam(p, reads, "MN908947.3", len(ref), level=6)
    elif kind == "random":
        rng = np.random.default_rng(1)
        reads["qual"] = rng.integers(0, 42, len(reads["qual"])).astype(np.uint8)
        bamwriter.write_bam(p, reads, "MN908947.3", len(ref), level=6)
    else:
        # (the coverage of the 1M-read bench file matters for the matches: scale the genome down instead of the reads up)
        bamwriter.write_bam_fast(p, np.sort(reads["pos"] // 50), reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref), level=6)
    for W in (256, 512):
        tot = []
        for k, body in enumerate(blocks_of(p)):
            if k < 1 or len(body) < 200:
                continue
            st = Stream(body + b"\0" * 8)
            final, typ, llp, dp, pos = header(st, 0)
            syms, ra, rb, al = simulate(st, pos, len(body) * 8, llp, dp, 64, W)
            tot.append((len(body), syms, ra, rb, al))
            if len(tot) >= 12:
                break
        a = np.array(tot)
        print(kind, "W", W, "blocks", len(tot), "bytes %.0f symbols %.0f | rounds A %.0f (max %d) B %.0f (max %d) alive %.1f | serial/parallel %.1f" %
              (a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 2].max(), a[:, 3].mean(), a[:, 3].max(), a[:, 4].mean(),
               a[:, 1].mean() / (a[:, 2].mean() + a[:, 3].mean())))


if __name__ == "__main__":
    main()
