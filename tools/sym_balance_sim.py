"""Design tool (CPU, pure Python): how many lock-step rounds bgzf_symbols' pass A takes per BGZF block under different ways of dealing the
block's bits to its lanes — (cur) a chunk of bits per lane, as the kernel does it; (static k) k sub-chunks per lane, c, c + lanes, ..;
(queue S) S sub-chunks, a lane that has met its neighbour takes the next unclaimed one — on real blocks of the bench's file kinds.
A round = every running lane decodes one symbol; a lane that crosses into a new stretch of 2^shift bits notes the position and,
every fourth round, looks it up in the notes of the sub-chunk in front.  python tools/sym_balance_sim.py [headline|hard|real] [blocks]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spec_inflate_proto import Stream, blocks_of, header, symbol     # noqa: E402


def next_table(st, start, end, llp, dp):
    """nxt[p - start] = where the symbol that starts at bit p ends (-1: no symbol / runs over the end; -2: end-of-block code)"""
    n = end - start
    nxt = np.full(n + 64, -1, np.int64)
    global SHORT_LIT
    SHORT_LIT = np.zeros(n + 64, bool)           # a literal whose code is in the 9-bit root table
    for p in range(start, end):
        k, q = symbol(st, p, llp, dp, end)
        if k == 2:
            nxt[p - start] = -2
        elif k >= 0:
            nxt[p - start] = q - start
            SHORT_LIT[p - start] = k == 0 and q - p <= 9
    return nxt


def run_exact(nxt, n_bits, lanes, W=256, check_every=1, S=None):
    """every symbol start within W bits of a lane's own start is noted (a bitmap); the lane behind stops at the first of its symbols that
    starts on a noted position.  -> (rounds, decoded, true, per-lane symbols)"""
    S = S or lanes
    chunk = max(1, -(-n_bits // S))
    starts = [min(c * chunk, n_bits) for c in range(S)]
    marks = [set() for _ in range(S)]
    pos = list(starts)
    run_ = [s < n_bits for s in starts]
    tot = [0] * S
    rounds = decoded = 0
    while any(run_):
        rounds += 1
        for c in range(S):
            if run_[c] and pos[c] - starts[c] < W:
                marks[c].add(pos[c])
        for c in range(S):
            if not run_[c]:
                continue
            p = pos[c]
            if rounds % check_every == 0 or check_every == 1:
                t = c + 1
                while t < S and p >= starts[t] + W:
                    t += 1
                if t < S and p >= starts[t] and p in marks[t]:
                    run_[c] = False
                    continue
            q = nxt[p] if p < n_bits else -1
            decoded += 1; tot[c] += 1
            if q < 0:
                run_[c] = False
                continue
            pos[c] = int(q)
    t, p = 0, 0
    while p >= 0 and p < n_bits:
        q = nxt[p]; t += 1
        if q < 0:
            break
        p = int(q)
    return rounds, decoded, t


SHORT_LIT = None


def run(nxt, n_bits, lanes, subs_per_lane=1, queue=0, check_every=4, shift_bias=0, pair=0):
    """-> (rounds, total symbols decoded by all lanes, true symbols)"""
    S = queue if queue else lanes * subs_per_lane
    chunk = max(1, -(-n_bits // S))
    shift = max(6, int(np.floor(np.log2(max(chunk, 2)))) - 1 - shift_bias)      # stretches of about chunk / 4 .. chunk / 2 bits, >= 64
    starts = [min(j * chunk, n_bits) for j in range(S)]
    notes = [dict() for _ in range(S)]           # per sub-chunk: stretch -> first symbol start in it
    owner_done = [False] * S                     # the sub-chunk's decoder has stopped (merged / eob / dead)
    claimed = [False] * S
    # lane state
    cur = [-1] * lanes
    pos = [0] * lanes
    kprev = [-1] * lanes
    cross = [None] * lanes
    run_ = [False] * lanes
    nxt_static = [0] * lanes
    next_q = 0
    p_before = [0] * lanes

    def take(c, j):
        cur[c] = j; pos[c] = starts[j]; kprev[c] = -1; cross[c] = None; run_[c] = starts[j] < n_bits; claimed[j] = True
        if not run_[c]:
            owner_done[j] = True
    for c in range(lanes):
        if queue:
            if next_q < S:
                take(c, next_q); next_q += 1
        else:
            take(c, c); nxt_static[c] = 1
    rounds = decoded = 0
    while any(run_):
        rounds += 1
        finished = []
        for c in range(lanes):
            if not run_[c]:
                continue
            j = cur[c]
            p = pos[c]
            k = p >> shift
            if k != kprev[c]:
                kprev[c] = k
                notes[j].setdefault(k, p)
                cross[c] = p
        for c in range(lanes):
            if not run_[c]:
                continue
            j = cur[c]
            if rounds % check_every == check_every - 1 and cross[c] is not None:
                p0 = cross[c]
                cross[c] = None
                # the sub-chunk in front whose range p0 lies in or behind (claimed ones only: an unclaimed one has no notes)
                t = j + 1
                while t < S and owner_done[t] and t + 1 < S and p0 >= starts[t + 1] and claimed[t + 1]:
                    t += 1
                if t < S and claimed[t] and notes[t].get(p0 >> shift) == p0:
                    run_[c] = False; owner_done[j] = True; finished.append(c)
                    continue
            q = nxt[pos[c]] if pos[c] < n_bits else -1
            decoded += 1
            p_before[c] = pos[c]
            if q < 0:
                run_[c] = False; owner_done[j] = True; finished.append(c)
                continue
            pos[c] = int(q)
            # literals in pairs (and triples): a lane whose symbol was a literal with a short code takes the next one too, in the same round, if
            # that is such a literal as well and starts in the same stretch (meeting points stay at round starts)
            first = p_here = int(q)
            for _ in range(pair):
                if not SHORT_LIT[pos[c] - 0] or not SHORT_LIT[p_before[c]]:
                    break
                q2 = nxt[pos[c]]
                if q2 < 0 or (pos[c] >> shift) != (p_before[c] >> shift) or pos[c] >= n_bits:
                    break
                p_before[c] = pos[c]
                pos[c] = int(q2)
            # crossing into a sub-chunk nobody has claimed yet: it is this lane's now
            if queue:
                while next_q < S and pos[c] >= starts[next_q] and next_q == max(jj for jj in range(S) if claimed[jj]) + 1 and cur[c] == next_q - 1:
                    claimed[next_q] = True; owner_done[next_q] = True; notes[next_q] = notes[j]; cur[c] = next_q; j = next_q; next_q += 1
        for c in finished:
            if queue:
                if next_q < S:
                    take(c, next_q); next_q += 1
            elif nxt_static[c] < subs_per_lane:
                take(c, c + lanes * nxt_static[c]); nxt_static[c] += 1
    # true symbols
    t, p = 0, 0
    while p >= 0 and p < n_bits:
        q = nxt[p]; t += 1
        if q < 0:
            break
        p = int(q)
    return rounds, decoded, t


def main():
    from bench import like_real_data
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    kind = sys.argv[1] if len(sys.argv) > 1 else "headline"
    nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ref, orfs = sy.make_reference()
    n = 200_000
    reads = sy.make_reads(ref, n, seed=3, start_range=(0, 6000))      # (the bench file's coverage: 1 M reads over 29 903 positions)
    d = tempfile.mkdtemp()
    p = os.path.join(d, "x.bam")
    if kind == "headline":
        bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref), level=6)
    else:
        qual, names = like_real_data(np, kind, n, seed=1)
        bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref), level=6, qual=qual, names=names)
    lanes = 32 if kind == "headline" else 64
    res = {}
    done = 0
    for k, body in enumerate(blocks_of(p)):
        if k < 3 or len(body) < 500:
            continue
        st = Stream(body + b"\0" * 16)
        final, typ, llp, dp, pos = header(st, 0)
        end = len(body) * 8
        nxt = next_table(st, pos, end, llp, dp)
        nb = end - pos
        for name, kw in (("cur", dict()), ("cur check1", dict(check_every=1)), ("cur bias1", dict(shift_bias=1)), ("pair2", dict(pair=1)), ("pair3", dict(pair=2)), ("pair4", dict(pair=3)), ("pair2 bias1", dict(pair=1, shift_bias=1)), ("static2", dict(subs_per_lane=2)),
                         ("queue2x", dict(queue=2 * lanes))):
            r = run(nxt, nb, lanes, **kw)
            res.setdefault(name, []).append(r)
        for name, kw in (("exact W256", dict(W=256)), ("exact W512", dict(W=512)), ("exact W256 every 2", dict(W=256, check_every=2)), ("exact W128", dict(W=128))):
            res.setdefault(name, []).append(run_exact(nxt, nb, lanes, **kw))
        done += 1
        if done >= nblk:
            break
    print(kind, "lanes", lanes, "blocks", done, "(deflate streams: only the first of each BGZF block)")
    for name, rs in res.items():
        a = np.array(rs, float)
        print("%-12s rounds mean %.0f max %.0f | decoded / true %.2f | true symbols %.0f | ideal rounds %.0f" %
              (name, a[:, 0].mean(), a[:, 0].max(), a[:, 1].sum() / a[:, 2].sum(), a[:, 2].mean(), a[:, 2].mean() / lanes))


if __name__ == "__main__":
    main()
