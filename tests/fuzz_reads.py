"""Random reads with arbitrary CIGARs (every op of SAM spec §4.2: M I D N S H P = X), odd SEQ
content (N, IUPAC codes, '=', SEQ '*'), flags and placements — for differential tests between
the column-major Python emulator, the read-major C oracle and the HIP kernels."""
from tests import synth_small as ss

_BASES = "ACGT"
_ODD = "NRYKM="


def random_cigar(rng, long_reads=False):
    """A structurally valid CIGAR: [H][S] core [S][H], core = ops over M I D N P = X."""
    n_core = int(rng.integers(1, 9))
    core = []
    last = None
    for _ in range(n_core):
        op = "MMMM=XIDDNP"[int(rng.integers(0, 11))]
        if op == last and op in "MIDNP=X" and rng.random() < 0.7:
            continue
        ln = int(rng.integers(1, 400 if long_reads and op in "MN" else 12 if op in "IDP" else 40))
        core.append((ln, op))
        last = op
    if not core:
        core = [(int(rng.integers(1, 30)), "M")]
    pre, post = [], []
    if rng.random() < 0.2:
        pre.append((int(rng.integers(1, 5)), "H"))
    if rng.random() < 0.3:
        pre.append((int(rng.integers(1, 9)), "S"))
    if rng.random() < 0.3:
        post.append((int(rng.integers(1, 9)), "S"))
    if rng.random() < 0.2:
        post.append((int(rng.integers(1, 5)), "H"))
    return pre + core + post


def random_specs(rng, n, L, long_reads=False):
    """-> list of read dicts (synth_small.reads_from_spec's input), unsorted."""
    reads = []
    for _ in range(n):
        cig = random_cigar(rng, long_reads)
        qlen = sum(l for l, op in cig if op in "MIS=X")
        r = rng.random()
        if r < 0.03:
            seq = "*"
        elif r < 0.06 and qlen > 3:
            seq = "".join(_BASES[int(k)] for k in rng.integers(0, 4, qlen - int(rng.integers(1, 3))))   # SEQ shorter than the CIGAR says
        else:
            seq = "".join((_ODD[int(rng.integers(0, len(_ODD)))] if rng.random() < 0.03 else _BASES[int(rng.integers(0, 4))])
                          for _ in range(qlen))
            if not seq:
                seq = "*"
        span = sum(l for l, op in cig if op in "MDN=X")
        pos = int(rng.integers(0, max(1, L - min(span, L // 2))))
        flag = int(rng.choice([0, 16, 0, 16, 4, 0x100, 0x400, 1, 3, 0x10 | 0x200]))
        reads.append({"pos": pos, "flag": flag, "cigar": "".join("%d%s" % t for t in cig), "seq": seq,
                      "qual": int(rng.integers(0, 41)), "tid": -1 if rng.random() < 0.01 else 0})
    return reads


def random_reads(rng, n, L, long_reads=False, sort=True):
    reads = random_specs(rng, n, L, long_reads)
    if sort:
        reads.sort(key=lambda r: r["pos"])
    return ss.reads_from_spec({"reads": reads})
