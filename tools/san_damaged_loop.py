#!/usr/bin/env python3
"""Randomly damaged BAM files through the HOST reader (tcmi_bam_load), for the sanitizer build (tools/san_check.sh): the reader must
refuse or decode every one of them without a report from AddressSanitizer / UBSan.  No GPU.

    python tools/san_damaged_loop.py [N=400] [seed=11]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                                   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    from trueconsense_amd import _ffi, engine
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    ref, _ = sy.make_reference()
    reads = sy.make_reads(ref[:3000], 2000, seed=seed, indel_sites=None)
    rng = np.random.default_rng(seed)
    refused = decoded = 0
    with tempfile.TemporaryDirectory() as tmp:
        good = os.path.join(tmp, "g.bam")
        bamwriter.write_bam(good, reads, "refid", 3000, block=4000)
        raw = open(good, "rb").read()
        p = os.path.join(tmp, "d.bam")
        for trial in range(n):
            data = bytearray(raw)
            kind = trial % 4
            if kind == 0:
                data = data[:int(rng.integers(1, len(data)))]
            elif kind == 1:
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(0, len(data)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                a = int(rng.integers(0, len(data) - 64))
                data[a:a + int(rng.integers(1, 64))] = bytes(int(rng.integers(0, 256)) for _ in range(1))
            else:
                a = int(rng.integers(0, len(data) - 8))
                data[a:a + 4] = int(rng.integers(0, 1 << 32)).to_bytes(4, "little")
            open(p, "wb").write(bytes(data))
            try:
                bam = engine.BamFile(p, threads=2)
                arr = bam.arrays()
                assert len(arr["pos"]) == bam.n_reads
                engine.reads_extent(bam, 3000)
                bam.close()
                decoded += 1
            except _ffi.TcmiError:
                refused += 1
    print("san_damaged_loop: %d files, %d refused, %d decoded (harmless damage), library %s" % (n, refused, decoded, _ffi.LIB_PATH))


if __name__ == "__main__":
    main()
