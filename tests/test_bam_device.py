"""GPU: the device BAM decoder (bam_device.hip: BGZF inflate, record chain) against zlib and a plain Python record
walk, and the whole file -> device -> counts path against the oracle.  Byte / index work: bit-exact."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import c_oracle
from tests import fuzz_reads as fz
from tests import synth_small as ss
from trueconsense_amd import _ffi, _state, engine
from trueconsense_amd import synthetic as sy
from trueconsense_amd.io import bamwriter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return _state.default_context()


def host_stream_and_records(path):
    """zlib (gzip.decompress walks every member) + a Python walk of the block_size chain."""
    raw = gzip.decompress(open(path, "rb").read())
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, o)[0]
        o += 4 + l_name + 4
    rec = []
    while o < len(raw):
        rec.append(o)
        o += 4 + struct.unpack_from("<i", raw, o)[0]
    return np.frombuffer(raw, np.uint8), np.array(rec, np.uint64)


def check_decode(ctx, path):
    want_stream, want_rec = host_stream_and_records(path)
    d = engine.DeviceBam(path)
    assert d.inflated_bytes == len(want_stream)
    stream, rec = d.decode_to_host(ctx)
    assert np.array_equal(stream, want_stream), np.argwhere(stream != want_stream)[:5]
    assert np.array_equal(rec, want_rec)
    return d


def check_counts(ctx, path, L):
    reads = c_oracle.read_bam(path)
    want = c_oracle.tally(reads, L)
    d = engine.DeviceBam(path)
    rs = ctx.upload_bamfile(d)
    assert rs.n_reads == reads["n_reads"]
    got = ctx.step(rs, L, 30, True)[3]
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    rs.free()
    d.close()


def test_device_inflate_and_record_index_match_zlib(ctx, tmp_path):
    ref, orfs = sy.make_reference()
    n = 60_000
    reads = sy.make_reads(ref, n, seed=3)
    for level in (6, 1, 9, 0):                         # level 0: stored blocks
        p = str(tmp_path / ("fast%d.bam" % level))
        bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref), level=level)
        check_decode(ctx, p).close()
    # indel carriers, qualities and names of varying length; every CIGAR op, odd SEQ content
    rng = np.random.default_rng(5)
    indel = sy.make_reads(ref, 30_000, seed=4, indel_sites=sy.default_indel_sites(orfs))
    indel["qual"] = rng.integers(0, 42, len(indel["qual"])).astype(np.uint8)      # incompressible-ish: long literal runs
    p = str(tmp_path / "indel.bam")
    bamwriter.write_bam(p, indel, "MN908947.3", len(ref), level=6)
    check_decode(ctx, p).close()
    fuzz = fz.random_reads(rng, 5000, 3000)
    p = str(tmp_path / "fuzz.bam")
    bamwriter.write_bam(p, fuzz, "ref", 3000, level=4)
    check_decode(ctx, p).close()
    # matches that reach back 10 - 30 KB (beyond the kernel's LDS ring): groups of reads with random qualities, repeated
    grp = sy.make_reads(ref, 40, seed=6)
    k = 50
    far = {"n_reads": 40 * k, "pos": np.tile(grp["pos"], k), "flag": np.tile(grp["flag"], k), "l_qseq": np.tile(grp["l_qseq"], k),
           "tid": np.zeros(40 * k, np.int32), "cigar_off": np.arange(40 * k + 1, dtype=np.uint64), "cigar": np.tile(grp["cigar"], k),
           "seq_off": np.arange(40 * k + 1, dtype=np.uint64) * np.uint64(75), "seq": np.tile(grp["seq"], k),
           "qual": np.tile(rng.integers(0, 42, 40 * 150).astype(np.uint8), k)}
    for level in (6, 9):
        p = str(tmp_path / ("far%d.bam" % level))
        bamwriter.write_bam(p, far, "MN908947.3", len(ref), level=level)
        check_decode(ctx, p).close()
    # a file that compresses like real data (6 : 1): distinct names, qualities from four bins — thousands of short matches per block,
    # the ones bgzf_copy hands to teams of lanes, a fifth of them further back than its LDS ring
    nh = 80_000
    hard = sy.make_reads(ref, nh, seed=12)
    qual = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=(nh, 150), p=[0.02, 0.05, 0.13, 0.80])
    names = rng.integers(48, 58, (nh, 27)).astype(np.uint8)
    names[:, :10] = np.frombuffer(b"A00123:45:", np.uint8)
    for level in (6, 1):
        p = str(tmp_path / ("hard%d.bam" % level))
        bamwriter.write_bam_fast(p, hard["pos"], hard["flag"], hard["seq"].reshape(nh, -1), 150, "MN908947.3", len(ref), level=level, qual=qual, names=names)
        d = check_decode(ctx, p)
        assert d.inflated_bytes < 12 * d.file_bytes              # (the ratio below which the host picks the team variant)
        d.close()
        check_counts(ctx, p, len(ref))
    # ... and one that compresses like real data with unbinned qualities (2.5 : 1): most matches are 3 - 8 bytes long and come from
    # anywhere in the 32 KB window — the far ones among them are finished in the batch's set-up (bgzf_copy<true, true>)
    q = rng.choice(np.arange(2, 42, dtype=np.uint8), size=(nh, 150), p=(lambda w: w / w.sum())(np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004))
    for level in (6, 1, 9):
        p = str(tmp_path / ("real%d.bam" % level))
        bamwriter.write_bam_fast(p, hard["pos"], hard["flag"], hard["seq"].reshape(nh, -1), 150, "MN908947.3", len(ref), level=level, qual=q, names=names)
        d = check_decode(ctx, p)
        assert d.inflated_bytes < 4 * d.file_bytes
        d.close()
        check_counts(ctx, p, len(ref))
    # no reads at all; one read
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and k not in ("cigar_off", "seq_off", "qual_off") else v) for k, v in reads.items()}
    empty.update(n_reads=0, cigar_off=np.zeros(1, np.uint64), seq_off=np.zeros(1, np.uint64), qual_off=np.zeros(1, np.uint64))
    p = str(tmp_path / "empty.bam")
    bamwriter.write_bam(p, empty, "MN908947.3", len(ref))
    check_decode(ctx, p).close()


def test_device_inflate_on_other_deflate_flavours(ctx, tmp_path):
    """Streams zlib's default settings never write: fixed-Huffman blocks only, run-length matches (distance 1, length 258:
    source overlaps destination), Huffman-only (no matches at all), small hash tables, and several deflate blocks — dynamic,
    and the empty stored ones of a full flush — inside one BGZF block.  Byte for byte against zlib."""
    import zlib
    ref, _ = sy.make_reference()
    rng = np.random.default_rng(11)
    n = 20_000
    reads = sy.make_reads(ref, n, seed=9)
    seq = reads["seq"].reshape(n, -1).copy()
    seq[::3, 10:60] = 0x11                                          # homopolymer stretches: long runs of one byte
    reads["seq"] = seq.reshape(-1)
    q = rng.integers(0, 42, n * 150).astype(np.uint8)
    q.reshape(n, 150)[::2, :] = 30                                  # constant qualities in every other read: runs of 150
    reads["qual"] = q
    try:
        for k, (strategy, mem, every, level) in enumerate(((zlib.Z_FIXED, 8, 0, 6), (zlib.Z_RLE, 8, 0, 6), (zlib.Z_HUFFMAN_ONLY, 8, 0, 6),
                                                         (zlib.Z_FILTERED, 1, 0, 9), (zlib.Z_DEFAULT_STRATEGY, 8, 5000, 6),
                                                         (zlib.Z_DEFAULT_STRATEGY, 9, 777, 1), (zlib.Z_FIXED, 8, 300, 6))):
            bamwriter.DEFLATE.update(strategy=strategy, mem_level=mem, flush_every=every)
            p = str(tmp_path / ("flavour%d.bam" % k))
            bamwriter.write_bam(p, reads, "MN908947.3", len(ref), level=level)
            check_decode(ctx, p).close()
            check_counts(ctx, p, len(ref))
    finally:
        bamwriter.DEFLATE.update(strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, flush_every=0)


def test_two_literals_in_one_token_and_pass_b_cut_the_stream_alike(ctx, tmp_path):
    """bgzf_symbols<1, *> (payloads beyond 4 KB) puts two literals into one token where the second code is a root-table literal in the same
    stretch of bits — in the hand-scheduled rounds of pass A, and in the C++ decoder that pass B runs over a block whose lanes overflowed
    their scratch: the two must cut the stream into the same tokens (a lane's `before` counts tokens).  Option "sym_scratch_div" makes lanes
    overflow: Huffman-only streams (nothing but literals), files of short tokens at 6 : 1 and 2.5 : 1, fixed-Huffman blocks, whole payloads
    and windows — byte for byte against zlib, counts against the oracle."""
    import zlib
    ref, _ = sy.make_reference()
    rng = np.random.default_rng(17)
    n = 30_000
    reads = sy.make_reads(ref, n, seed=19)
    q_real = rng.choice(np.arange(2, 42, dtype=np.uint8), size=(n, 150), p=(lambda w: w / w.sum())(np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004))
    q_hard = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=(n, 150), p=[0.02, 0.05, 0.13, 0.80])
    names = rng.integers(48, 58, (n, 27)).astype(np.uint8)
    files = []
    for tag, q in (("real", q_real), ("hard", q_hard)):
        p = str(tmp_path / (tag + ".bam"))
        bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref), level=6, qual=q, names=names)
        files.append(p)
    r2 = dict(reads)
    r2["qual"] = q_real.reshape(-1)
    try:
        for k, (strategy, level) in enumerate(((zlib.Z_HUFFMAN_ONLY, 6), (zlib.Z_FIXED, 6))):
            bamwriter.DEFLATE.update(strategy=strategy, mem_level=8, flush_every=0)
            p = str(tmp_path / ("lit%d.bam" % k))
            bamwriter.write_bam(p, r2, "MN908947.3", len(ref), level=level)
            files.append(p)
    finally:
        bamwriter.DEFLATE.update(strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, flush_every=0)
    try:
        for div in (1, 4, 16, 64):
            ctx.set_option("sym_scratch_div", div)
            for p in files:
                check_decode(ctx, p).close()
                check_counts(ctx, p, len(ref))
    finally:
        ctx.set_option("sym_scratch_div", 1)


@pytest.mark.timeout(300)
def test_damaged_files_end_in_an_error_or_the_right_answer(ctx, tmp_path):
    """Random damage to the compressed bytes (1 - 3 byte flips per trial, 60 trials, dynamic and fixed-Huffman streams): the
    device decoder must refuse the file (deflate stream, ISIZE, CRC-32, record chain) — or, where the damage hit a byte
    nothing depends on (a gzip MTIME, say), give exactly the counts of the intact file.  Never hang, never a wrong answer."""
    import zlib
    ref, _ = sy.make_reference(L=6000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 30_000, seed=13)
    rng = np.random.default_rng(17)
    reads["qual"] = rng.integers(0, 42, len(reads["qual"])).astype(np.uint8)
    L = len(ref)
    refused = same = 0
    try:
        for flavour, (strategy, level) in enumerate(((zlib.Z_DEFAULT_STRATEGY, 6), (zlib.Z_FIXED, 6), (zlib.Z_DEFAULT_STRATEGY, 1))):
            bamwriter.DEFLATE.update(strategy=strategy)
            p = str(tmp_path / ("intact%d.bam" % flavour))
            bamwriter.write_bam(p, reads, "r", L, level=level)
            d = engine.DeviceBam(p)
            rs = ctx.upload_bamfile(d)
            want = ctx.step(rs, L, 30, True)[3].copy()
            rs.free(); d.close()
            raw = open(p, "rb").read()
            for trial in range(20):
                bad = bytearray(raw)
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, len(bad) - 28))] ^= int(rng.integers(1, 256))      # (the EOF block stays)
                q = str(tmp_path / "damaged.bam")
                open(q, "wb").write(bytes(bad))
                try:
                    d = engine.DeviceBam(q)                 # (the host side may refuse already: block sizes, magic)
                except _ffi.TcmiError:
                    refused += 1
                    continue
                try:
                    rs = ctx.upload_bamfile(d)
                except _ffi.TcmiError as e:
                    assert e.code in (_ffi.E_FORMAT, _ffi.E_UNSUPPORTED), e
                    refused += 1
                else:
                    got = ctx.step(rs, max(L, rs.max_end), 30, True)[3]
                    assert got.shape == want.shape and np.array_equal(got, want), (flavour, trial)
                    same += 1
                    rs.free()
                d.close()
    finally:
        bamwriter.DEFLATE.update(strategy=zlib.Z_DEFAULT_STRATEGY)
    assert refused >= 50 and refused + same == 60


def test_file_to_counts_all_on_device(ctx, tmp_path):
    ref, orfs = sy.make_reference()
    L = len(ref)
    n = 200_000
    reads = sy.make_reads(ref, n, seed=8)
    p = str(tmp_path / "a.bam")
    bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)
    check_counts(ctx, p, L)
    indel = sy.make_reads(ref, 50_000, seed=9, indel_sites=sy.default_indel_sites(orfs))
    tail = 300                                         # unmapped reads at the end of a sorted BAM
    indel["flag"][-tail:] |= 4
    indel["tid"][-tail:] = -1
    indel["pos"][-tail:] = -1
    p = str(tmp_path / "b.bam")
    bamwriter.write_bam(p, indel, "MN908947.3", L, level=6)
    check_counts(ctx, p, L)
    rng = np.random.default_rng(17)
    fuzz = fz.random_reads(rng, 8000, 3000)
    p = str(tmp_path / "c.bam")
    bamwriter.write_bam(p, fuzz, "ref", 3000, level=6)
    check_counts(ctx, p, int(engine.reads_extent(fuzz, 3000)))


def test_records_that_straddle_bgzf_blocks_are_decoded_on_the_device(ctx, tmp_path):
    """Writers other than htslib (htsjdk: Picard, GATK) fill every BGZF block to the brim: records — and their block_size fields —
    run from one block into the next.  Every block finds the first record start in its own bytes, the host checks that the
    chain closes (bam_device.hip); stream, record index and counts as from a file cut on record boundaries."""
    ref, _ = sy.make_reference(L=5000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 4000, seed=1)
    p = str(tmp_path / "split.bam")
    bamwriter.write_bam(p, reads, "r", len(ref), split_records=True)      # records straddle BGZF blocks
    check_decode(ctx, p).close()
    check_counts(ctx, p, len(ref))
    # names of different lengths, indel carriers, small blocks: starts and size fields fall on every offset of a block's end
    rng = np.random.default_rng(23)
    fuzz = fz.random_reads(rng, 6000, 3000)
    for block in (0xFF00, 4099, 1021):
        p = str(tmp_path / ("split%d.bam" % block))
        bamwriter.write_bam(p, fuzz, "ref", 3000, level=6, block=block, split_records=True)
        check_decode(ctx, p).close()
        check_counts(ctx, p, int(engine.reads_extent(fuzz, 3000)))
    # the file runner takes the device for both kinds
    runner = engine.FileRunner(ctx, [{"start": 10, "end": 600, "strand": "+"}], 30)
    p = str(tmp_path / "split.bam")
    p2 = str(tmp_path / "whole.bam")
    bamwriter.write_bam(p2, reads, "r", len(ref))
    a, b = runner.run([p, p2], names=["S", "S"], ref_len=len(ref))
    assert a == b and len(a.split("\n")[1]) == len(ref)
    assert runner.decoded_on == {"device": 2, "host": 0}


def test_long_reads_stay_on_the_device(ctx, tmp_path):
    """Reads that span more reference positions than a chunk of the packed set holds (long-read platforms: 2 kb amplicon reads
    among short ones, every CIGAR operation in them) no longer send the whole file to the host reader: they are walked in
    the inflated stream by tally_stream_kernel.  Counts against the oracle; the file runner reports a device decode."""
    ref, _ = sy.make_reference(L=12000, cds=[(10, 600)])
    L = len(ref)
    rng = np.random.default_rng(31)
    specs = fz.random_specs(rng, 40_000, L) + fz.random_specs(rng, 800, L, long_reads=True)    # 2 %: up to a few kb each, indels / skips / clips
    specs.sort(key=lambda r: r["pos"])
    both = ss.reads_from_spec({"reads": specs})
    p = str(tmp_path / "mixed.bam")
    bamwriter.write_bam(p, both, "r", L, level=6)
    Lx = int(max(L, engine.reads_extent(both, L)))
    d = engine.DeviceBam(p)
    rs = ctx.upload_bamfile(d)
    want = c_oracle.tally(both, Lx)
    got = ctx.step(rs, Lx, 30, True)[3]
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    rs.free()
    d.close()
    runner = engine.FileRunner(ctx, [{"start": 10, "end": 600, "strand": "+"}], 30)
    runner.run([p], names=["S"], ref_len=L)
    assert runner.decoded_on == {"device": 1, "host": 0}


def test_a_damaged_stream_is_an_error(ctx, tmp_path):
    ref, _ = sy.make_reference(L=5000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 4000, seed=1)
    p2 = str(tmp_path / "whole.bam")
    bamwriter.write_bam(p2, reads, "r", len(ref))
    # a damaged deflate stream is an error, not a hang and not a wrong answer
    raw = bytearray(open(p2, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    p3 = str(tmp_path / "bad.bam")
    open(p3, "wb").write(bytes(raw))
    d = engine.DeviceBam(p3)
    try:
        rs = ctx.upload_bamfile(d)
    except _ffi.TcmiError as err:
        assert err.code in (_ffi.E_FORMAT, _ffi.E_UNSUPPORTED)
    else:                                               # (the flipped bit sat in a field nothing depends on, e.g. a gzip MTIME)
        rs.free()
    d.close()


def test_crc32_of_every_block_is_checked_on_the_device(ctx, tmp_path):
    """A payload byte changed inside a STORED deflate block: the stream inflates, ISIZE holds, the record chain holds — only
    the block's CRC-32 can tell (htslib checks it on every block; bgzf_copy does here, while it flushes the block's bytes)."""
    ref, _ = sy.make_reference(L=5000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 6000, seed=2)
    p = str(tmp_path / "stored.bam")
    bamwriter.write_bam(p, reads, "r", len(ref), level=0)
    check_decode(ctx, p).close()                        # (all CRCs of the intact file agree, lengths that are not multiples of 16 included)
    raw = bytearray(open(p, "rb").read())
    offs, o = [], 0
    while o < len(raw):
        offs.append(o)
        o += struct.unpack_from("<H", raw, o + 16)[0] + 1
    assert len(offs) > 4
    b = offs[2]
    assert raw[b + 18] & 6 == 0                         # BTYPE 00: stored
    at = b + 18 + 5 + 36 + 20                           # inside the first record's name / CIGAR / SEQ bytes
    raw[at] ^= 0x10
    p2 = str(tmp_path / "flipped.bam")
    open(p2, "wb").write(bytes(raw))
    d = engine.DeviceBam(p2)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.upload_bamfile(d)
    assert e.value.code == _ffi.E_FORMAT and "CRC32" in str(e.value) and "block 2" in str(e.value)
    ctx.set_option("verify_crc", 0)                     # without the check the damaged byte goes through unnoticed
    try:
        ctx.upload_bamfile(d).free()
    finally:
        ctx.set_option("verify_crc", 1)
    d.close()
    with pytest.raises(_ffi.TcmiError):                 # the host reader says the same
        engine.BamFile(p2, threads=2)
    # ... and a damaged CRC field over intact data
    raw[at] ^= 0x10
    end = offs[3]
    raw[end - 8] ^= 1
    open(p2, "wb").write(bytes(raw))
    d = engine.DeviceBam(p2)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.upload_bamfile(d)
    assert e.value.code == _ffi.E_FORMAT and "CRC32" in str(e.value)
    d.close()


def test_crc_in_the_flush_blocks_of_every_shape(ctx, tmp_path):
    """The CRC-32 is taken from bgzf_copy's LDS ring while a block's 2 KiB segments are flushed: a block's first byte lies
    at any offset of its first 16-byte row (the blocks' outputs follow each other without gaps), its length is anything — under a segment,
    a whole number of them, one byte more or less — and its end is cut into 32-byte pieces from the END.  Stored blocks of random sizes
    (level 0: what one flipped payload byte changes is the CRC alone), filled to the brim so that records straddle them: every intact
    file decodes (all CRCs agree), and a flipped byte in a random block — first / middle / last bytes of it — is a CRC32 error of that block."""
    rng = np.random.default_rng(11)
    ref, _ = sy.make_reference(L=4000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 5000, seed=9)
    n_flips = 0
    for block in (700, 2047, 2048, 2049, 4096 + 17, 6000, 65280, int(rng.integers(800, 9000)), int(rng.integers(9000, 60000))):      # (blocks shorter than a record: the record search may not close — host reader — another test)
        p = str(tmp_path / ("b%d.bam" % block))
        bamwriter.write_bam(p, reads, "r", len(ref), level=0, block=block, split_records=True)
        check_decode(ctx, p).close()
        raw = bytearray(open(p, "rb").read())
        offs, o = [], 0
        while o < len(raw):
            offs.append(o)
            o += struct.unpack_from("<H", raw, o + 16)[0] + 1
        body = [k for k in range(1, len(offs) - 1) if struct.unpack_from("<I", raw, offs[k + 1] - 4)[0] >= 128]      # (not the header's, not the end-of-file marker)
        for k in (body[0], body[len(body) // 2], body[-1]):
            ulen = struct.unpack_from("<I", raw, offs[k + 1] - 4)[0]
            assert raw[offs[k] + 18] & 6 == 0                    # BTYPE 00: stored
            for rel in (0, ulen // 2, ulen - 1):
                at = offs[k] + 18 + 5 + rel
                raw[at] ^= 0x40
                p2 = str(tmp_path / "flip.bam")
                open(p2, "wb").write(bytes(raw))
                d = engine.DeviceBam(p2)
                with pytest.raises(_ffi.TcmiError) as e:
                    ctx.upload_bamfile(d)
                # (a flipped length / name byte can also break the record chain or a record's fields: refused either way, and never silently)
                assert e.value.code in (_ffi.E_FORMAT, _ffi.E_UNSUPPORTED), (block, k, rel, str(e.value))
                if "CRC32" in str(e.value):
                    assert ("block %d" % k) in str(e.value), (block, k, rel, str(e.value))
                    n_flips += 1
                d.close()
                raw[at] ^= 0x40
    assert n_flips >= 40


def test_long_insertions_on_a_candidate_column_stay_on_the_device(ctx, tmp_path):
    """An insertion of more than 12 bases does not fit the device entry's 64-bit token key (round 2 sent such a file to a host
    decode): the kernel leaves its bases in a text buffer, the host builds the token text from there.  Against the host sweep and the
    pileup emulator (Events.py:63-80: the modal token of the column, in upper case)."""
    from tests import synth_small as ss
    from oracle import tc_oracle as orc
    rng = np.random.default_rng(8)
    ins_a = "ACGTTGCAAGGCTTAACCGGT"                       # 21 bases
    ins_b = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 57))
    reads = []
    for k in range(900):
        kind = k % 3
        if kind == 0:
            reads.append({"pos": 100 + k % 7, "flag": 16 if k % 2 else 0, "cigar": "%dM21I%dM" % (20 - k % 7, 20 + k % 7), "qual": 30, "name": "a%d" % k,
                          "seq": "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 20 - k % 7)) + ins_a + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 20 + k % 7))})
        elif kind == 1:
            reads.append({"pos": 150 + k % 5, "flag": 0, "cigar": "%dM57I%dM" % (30 - k % 5, 10 + k % 5), "qual": 25, "name": "b%d" % k,
                          "seq": "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30 - k % 5)) + (ins_b if k % 9 else ins_b[:-1] + "N") + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 10 + k % 5))})
        else:
            reads.append({"pos": 95 + k % 11, "flag": 0, "cigar": "90M", "qual": 30, "name": "c%d" % k, "seq": "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 90))})
    reads.sort(key=lambda r: r["pos"])
    rd = ss.reads_from_spec({"reads": reads})
    p = str(tmp_path / "longins.bam")
    bamwriter.write_bam(p, rd, "r", 400)
    d = engine.DeviceBam(p)
    rs = ctx.upload_bamfile(d)
    assert rs.packed_on_device
    cols = [119, 120, 179, 180]                            # (1-based: the last base in front of each insertion, and the one behind)
    got = ctx.readset_modal_tokens(rs, cols)
    assert got == engine.modal_tokens(engine.BamFile(p), cols)
    for c in cols:
        want = orc.region_tokens(rd, c)
        assert got[c][1] == len(want) and got[c][0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0], c
    assert got[120][0].endswith("+21" + ins_a) and got[180][0].endswith("+57" + ins_b)
    rs.free(); d.close()


def test_insert_tokens_resolved_on_the_device(ctx, tmp_path):
    """Events.ExtractInserts (Events.py:47-82) for a BAM the device decoded: the HIP kernel's entries + the host vote against
    the host sweep over the host-decoded file and against the oracle — indel sites of the configs[2] kind, a column deeper than
    max_depth = 8000, and overlapping mates."""
    from tests import synth_small as ss
    from oracle import tc_oracle as orc
    ref, orfs = sy.make_reference()
    L = len(ref)
    reads = sy.make_reads(ref, 120_000, seed=21, indel_sites=sy.default_indel_sites(orfs))
    rng = np.random.default_rng(3)
    reads["qual"] = rng.integers(5, 41, len(reads["qual"])).astype(np.uint8)          # the BQ >= 13 filter has something to do
    p = str(tmp_path / "ind.bam")
    bamwriter.write_bam(p, reads, "MN908947.3", L, level=1)
    d = engine.DeviceBam(p)
    rs = ctx.upload_bamfile(d)
    plain, alt, flags, counts = ctx.step(rs, L, 30, True)
    cand = (np.nonzero(flags & _ffi.F_INSCAND)[0] + 1).tolist()
    sites = sorted(s[0] for s in sy.default_indel_sites(orfs))
    cols = sorted(set(cand + sites))
    assert len(cand) >= 3
    got = ctx.readset_modal_tokens(rs, cols)
    host = engine.modal_tokens(engine.BamFile(p), cols)
    assert got == host
    reads["sorted_max_span"] = int(reads["sorted_max_span"])
    for c in cols:
        lo, hi = int(np.searchsorted(reads["pos"], c - 1 - 200, "left")), int(np.searchsorted(reads["pos"], c - 1, "right"))
        sub = {"n_reads": hi - lo, "pos": reads["pos"][lo:hi], "flag": reads["flag"][lo:hi], "l_qseq": reads["l_qseq"][lo:hi], "tid": reads["tid"][lo:hi],
               "cigar_off": (reads["cigar_off"][lo:hi + 1] - reads["cigar_off"][lo]).astype(np.uint64), "cigar": reads["cigar"][int(reads["cigar_off"][lo]):int(reads["cigar_off"][hi])],
               "seq_off": (reads["seq_off"][lo:hi + 1] - reads["seq_off"][lo]).astype(np.uint64), "seq": reads["seq"][int(reads["seq_off"][lo]):int(reads["seq_off"][hi])],
               "qual_off": (reads["qual_off"][lo:hi + 1] - reads["qual_off"][lo]).astype(np.uint64), "qual": reads["qual"][int(reads["qual_off"][lo]):int(reads["qual_off"][hi])]}
        want = orc.region_tokens(sub, c)
        assert got[c][1] == len(want), c
        if want:
            assert got[c][0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0], c
    # after another upload on the context the stream is gone: refused, and the caller takes the host sweep
    rs2 = ctx.upload_bamfile(d)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.readset_modal_tokens(rs, cols)
    assert e.value.code == _ffi.E_UNSUPPORTED
    assert ctx.readset_modal_tokens(rs2, cols) == host
    rs.free(); rs2.free(); d.close()
    # ---- deeper than max_depth, and overlapping mates with names ----
    deep = [{"pos": 100, "flag": 0, "cigar": "10M2I10M" if k < 4200 else "20M", "seq": "ACGTACGTAC" + ("GG" if k < 4200 else "") + "ACGTACGTAC",
             "qual": 30, "name": "d%d" % k} for k in range(9500)]
    pairs = []
    for k in range(300):
        start, mate = 200 + int(rng.integers(0, 8)), 208 + int(rng.integers(0, 8))
        b1 = "ACGT"[int(rng.integers(0, 4))]
        b2 = b1 if rng.random() < 0.6 else "ACGT"[int(rng.integers(0, 4))]
        s1 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30))
        s2 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30))
        i1, i2 = 220 - start, 220 - mate
        s1, s2 = s1[:i1] + b1 + s1[i1 + 1:], s2[:i2] + b2 + s2[i2 + 1:]
        pairs.append({"pos": start, "flag": 99, "cigar": "%dM1I%dM" % (i1 + 1, 28 - i1), "seq": s1, "qual": [int(rng.integers(5, 41))] * 30,
                      "name": "p%d" % k, "mtid": 0, "mpos": mate, "tlen": mate + 30 - start})
        pairs.append({"pos": mate, "flag": 147, "cigar": "30M", "seq": s2, "qual": [int(rng.integers(5, 41))] * 30, "name": "p%d" % k, "mtid": 0,
                      "mpos": start, "tlen": -(mate + 30 - start)})
    allr = sorted(deep + pairs, key=lambda r: r["pos"])
    rd = ss.reads_from_spec({"reads": allr})
    p2 = str(tmp_path / "deep.bam")
    bamwriter.write_bam(p2, rd, "r", 400)
    d2 = engine.DeviceBam(p2)
    rs = ctx.upload_bamfile(d2)
    got = ctx.readset_modal_tokens(rs, [110, 221])
    hb = engine.BamFile(p2)
    assert got == engine.modal_tokens(hb, [110, 221])
    assert got[110] == ("C+2GG", 8000)
    want = orc.region_tokens(rd, 221)
    assert got[221][1] == len(want) and got[221][0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0]
    assert got[221][1] < len(orc.region_tokens(rd, 221, ignore_overlaps=False))
    rs.free(); d2.close()
    # ---- a mate with a deletion / ref-skip ON the column: its token is tested on its next query base, where the overlap tweak
    #      reaches it if the other mate has a matched base there (ins_probe_kernel looks) ----
    dels = []
    for k in range(500):
        start = 300 + int(rng.integers(0, 6))
        mate = start + int(rng.integers(0, 8))
        s1 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 40))
        s2 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 40))
        if rng.random() < 0.5:
            s2 = s1[mate - start:] + s2[:mate - start]

        def gap_cigar(read_start):
            gap = int(rng.integers(1, 5))
            a = 319 - read_start - int(rng.integers(0, gap))
            kind = "N" if rng.random() < 0.2 else "D"
            if rng.random() < 0.2:
                return "%dM%d%s2I%dM" % (a, gap, kind, 40 - a - 2)
            return "%dM%d%s%dM" % (a, gap, kind, 40 - a)
        which = rng.random()
        dels.append({"pos": start, "flag": 99, "cigar": gap_cigar(start) if which < 0.7 else "40M", "seq": s1, "qual": [int(x) for x in rng.integers(4, 30, 40)],
                     "name": "g%d" % k, "mtid": 0, "mpos": mate, "tlen": mate + 44 - start})
        dels.append({"pos": mate, "flag": 147, "cigar": gap_cigar(mate) if which > 0.5 else "40M", "seq": s2, "qual": [int(x) for x in rng.integers(4, 30, 40)],
                     "name": "g%d" % k, "mtid": 0, "mpos": start, "tlen": -(mate + 44 - start)})
    dels.sort(key=lambda r: r["pos"])
    rd = ss.reads_from_spec({"reads": dels})
    p3 = str(tmp_path / "gaps.bam")
    bamwriter.write_bam(p3, rd, "r", 500)
    d3 = engine.DeviceBam(p3)
    rs = ctx.upload_bamfile(d3)
    cols = [319, 320, 321, 322]
    got = ctx.readset_modal_tokens(rs, cols)
    assert got == engine.modal_tokens(engine.BamFile(p3), cols)
    changed = False
    for c in cols:
        want = orc.region_tokens(rd, c)
        assert got[c][1] == len(want) and got[c][0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0], c
        changed |= len(want) != len(orc.region_tokens(rd, c, ignore_overlaps=False))
    assert changed
    rs.free(); d3.close()


def test_copy_loop_on_matches_of_every_length_and_alignment(ctx, tmp_path):
    """bgzf_copy's plain-match loop copies up to 64 bytes a byte a lane and longer matches an aligned destination dword a lane
    (the source funnel-shifted, the first and the last dword merged under a byte mask, two more dwords beyond 256 bytes).  Here
    every read's qualities are random except for a stretch of 3 .. 300 bytes taken from an earlier read's at random offsets: matches
    of every length at every pair of alignments, from the read before (in the LDS ring), from twelve reads back (beyond it: fetched
    from the flushed output, parked next to the ring up to 64 bytes, 64 bytes a load beyond) and from three reads back.
    Byte for byte against zlib; with and without teams (the variant follows the file's compression ratio: both kinds of file)."""
    ref, _ = sy.make_reference()
    rng = np.random.default_rng(21)
    n, lq = 6000, 400
    base = sy.make_reads(ref, n, read_len=lq, seed=22)
    common = rng.integers(0, 42, lq).astype(np.uint8)
    for k, (filler, level) in enumerate((("random", 9), ("random", 6), ("common", 9))):
        # ("common": every read's qualities are one pattern apart from the stretch — 18 : 1, the host picks bgzf_copy<false>; random
        #  filler: 3 : 1, bgzf_copy<true> with its teams)
        q = np.empty((n, lq), np.uint8)
        q[0] = common
        for i in range(1, n):
            row = common.copy() if filler == "common" else rng.integers(0, 42, lq).astype(np.uint8)
            L = 3 + (i * 7) % 298
            back = 12 if i % 5 == 0 and i >= 12 else 3 if i % 7 == 0 and i >= 3 else 1
            a, b = int(rng.integers(0, lq - L + 1)), int(rng.integers(0, lq - L + 1))
            row[b:b + L] = q[i - back][a:a + L]
            q[i] = row
        reads = dict(base)
        reads["qual"] = q.reshape(-1)
        p = str(tmp_path / ("lengths%d.bam" % k))
        bamwriter.write_bam(p, reads, "MN908947.3", len(ref), level=level)
        d = check_decode(ctx, p)
        assert (d.inflated_bytes >= 12 * d.file_bytes) == (filler == "common")      # (the ratio at which the host changes the variant)
        d.close()

@pytest.mark.timeout(600)
@pytest.mark.parametrize("window", [2048, 4100])
def test_windowed_symbol_decoder_at_small_windows(window, tmp_path):
    """bgzf_symbols<1, true> stages a block's payload a window at a time (8 KB when a payload exceeds 16 KB).  Here the window is
    forced small (the launcher reads TCMI_SYM_WINDOW once per process, hence a child process), so that every file of the
    decoder tests above with payloads over 4 KB crosses many window ends: in the middle of a symbol, of a dynamic header,
    of a stored block, of the last symbols before an end-of-block code.  Byte for byte against zlib, counts against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import pathlib, sys\n"
            "from tests import test_bam_device as t\n"
            "from trueconsense_amd import _state\n"
            "ctx = _state.default_context()\n"
            "for k, f in enumerate((t.test_device_inflate_and_record_index_match_zlib, t.test_device_inflate_on_other_deflate_flavours)):\n"
            "    d = pathlib.Path(sys.argv[1]) / str(k); d.mkdir()\n"
            "    f(ctx, d)\n"
            "print('windowed ok')\n")
    env = dict(os.environ, TCMI_SYM_WINDOW=str(window), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], cwd=root, env=env, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0 and "windowed ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_files_beyond_the_token_scratch_are_decoded_in_batches_of_blocks(tmp_path):
    """"decode_token_mb": the decoder's token scratch.  A block in flight needs 3 - 10 times its inflated bytes of it; a file that
    needs more than the scratch holds goes through bgzf_symbols + bgzf_copy a batch of blocks at a time (a large real file: tens of
    GB otherwise) — same stream, same records, same counts, in both packer paths, whole and as block ranges."""
    c = engine.Context(0)
    ref, orfs = sy.make_reference()
    L = len(ref)
    rng = np.random.default_rng(41)
    files = []
    n = 60_000
    r = sy.make_reads(ref, n, seed=42)
    p = str(tmp_path / "plain.bam")
    bamwriter.write_bam_fast(p, r["pos"], r["flag"], r["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)            # pairs of blocks per workgroup
    files.append(p)
    q = rng.choice(np.arange(2, 42, dtype=np.uint8), size=(n, 150), p=(lambda w: w / w.sum())(np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004))
    names = rng.integers(48, 58, (n, 27)).astype(np.uint8)
    p = str(tmp_path / "real.bam")
    bamwriter.write_bam_fast(p, r["pos"], r["flag"], r["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6, qual=q, names=names)   # windowed decoder, tokens moved
    files.append(p)
    indel = sy.make_reads(ref, 20_000, seed=43, indel_sites=sy.default_indel_sites(orfs))
    p = str(tmp_path / "brim.bam")
    bamwriter.write_bam(p, indel, "MN908947.3", L, level=6, split_records=True)                                          # records across blocks — and across batches
    files.append(p)
    try:
        for path in files:
            for mb in (1, 3):
                c.set_option("decode_token_mb", mb)
                before = c.stat("decode_batched")
                check_decode(c, path).close()
                assert c.stat("decode_batched") > before, "the file fits %d MiB of tokens: no batches" % mb
                for one_sync in (1, 0):
                    c.set_option("one_sync", one_sync)
                    check_counts(c, path, L)
                    d = engine.DeviceBam(path)
                    nb = d.n_blocks
                    acc = None
                    for a, k in ((0, nb // 2), (nb // 2, nb - nb // 2)):
                        rs = c.upload_bamfile(d, blocks=(a, k))
                        got = c.step(rs, L, 30, True)[3].astype(np.int64)
                        acc = got if acc is None else acc + got
                        rs.free()
                    d.close()
                    assert np.array_equal(acc, c_oracle.tally(c_oracle.read_bam(path), L))
                c.set_option("one_sync", 1)
    finally:
        c.close()
