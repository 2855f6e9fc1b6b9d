"""GPU: the device BAM decoder (bam_device.hip: BGZF inflate, record chain) against zlib and a plain Python record
walk, and the whole file -> device -> counts path against the oracle.  Byte / index work: bit-exact."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import c_oracle
from tests import fuzz_reads as fz
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
    # no reads at all; one read
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and k not in ("cigar_off", "seq_off", "qual_off") else v) for k, v in reads.items()}
    empty.update(n_reads=0, cigar_off=np.zeros(1, np.uint64), seq_off=np.zeros(1, np.uint64), qual_off=np.zeros(1, np.uint64))
    p = str(tmp_path / "empty.bam")
    bamwriter.write_bam(p, empty, "MN908947.3", len(ref))
    check_decode(ctx, p).close()


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


def test_files_the_device_decoder_leaves_to_the_host(ctx, tmp_path):
    ref, _ = sy.make_reference(L=5000, cds=[(10, 600)])
    reads = sy.make_reads(ref, 4000, seed=1)
    p = str(tmp_path / "split.bam")
    bamwriter.write_bam(p, reads, "r", len(ref), split_records=True)      # records straddle BGZF blocks
    d = engine.DeviceBam(p)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.upload_bamfile(d)
    assert e.value.code == _ffi.E_UNSUPPORTED and "straddles" in str(e.value)
    d.close()
    # ... and the file runner then takes the host reader for it, with the same result
    runner = engine.FileRunner(ctx, [{"start": 10, "end": 600, "strand": "+"}], 30)
    p2 = str(tmp_path / "whole.bam")
    bamwriter.write_bam(p2, reads, "r", len(ref))
    a, b = runner.run([p, p2], names=["S", "S"], ref_len=len(ref))
    assert a == b and len(a.split("\n")[1]) == len(ref)
    assert runner.decoded_on == {"device": 1, "host": 1}
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
    else:                                               # the flipped bit may decode to a valid stream of the right length:
        rs.free()                                       # then only the CRC (checked by the host reader) can tell
    d.close()
