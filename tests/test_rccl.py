"""RCCL itself, executed: the reduce of BASELINE configs[4] ("an RCCL all-reduce over xGMI of the per-tile count matrices", the pass it
replaces: the single pile-up of indexing.py:96-100) through the C hook of include/tcmi_rccl.h — ncclReduce queued on the context's
stream by tcmi_split_step.  The box has one GPU, so the communicator has one rank; what this pins is that the hook library loads next
to torch's HIP runtime, that a communicator comes up, that the collective runs on the stream behind the tally with the range table
and the failure word behind the matrix, and that the result is the oracle chain's FASTA."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_split_step_reduces_through_rccl_on_a_one_rank_communicator(tmp_path):
    import torch
    from tests.test_distributed import _consensus_case, _oracle_fasta
    from trueconsense_amd import _ffi
    from trueconsense_amd import distributed as td
    from trueconsense_amd.io import bamwriter
    ref, orfs, reads = _consensus_case()
    L = len(ref)
    rows = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs]
    want, ins = _oracle_fasta(reads, orfs, L, 30)
    assert len(ins) >= 4
    path = str(tmp_path / "one.bam")
    bamwriter.write_bam(path, reads, "r", L, level=6, block=3000, split_records=True)
    torch.cuda.set_device(0)
    comm, user = td.rccl_communicator(0, 1)
    r = _ffi.rccl_lib()
    try:
        w, k, dev = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        assert r.tcmi_rccl_comm_info(comm, C.byref(w), C.byref(k), C.byref(dev)) == 0 and (w.value, k.value, dev.value) == (1, 0, 0)
        # the hook alone: an int32 buffer summed "over the ranks" in place on a side stream
        t = torch.arange(1000, dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        assert r.tcmi_rccl_reduce(C.cast(C.pointer(user), C.c_void_p), C.c_void_p(t.data_ptr()), 1000, C.c_void_p(s.cuda_stream)) == 0
        s.synchronize()
        assert t.cpu().tolist() == list(range(1000))
        # the product's function for configs[4] with that hook, all the way to the FASTA (+ counts and tokens)
        tm = {}
        text, counts, toks = td.consensus_split_bamfile(path, L, rows, 30, True, "S", 0, 1, device=0, rccl_user=user, return_parts=True, timings=tm)
        from oracle import c_oracle
        assert text == want
        assert np.array_equal(counts, c_oracle.tally(reads, L))
        assert tm["step"] > 0 and sum(1 for v in toks.values() if v) >= 4
    finally:
        assert r.tcmi_rccl_comm_destroy(comm) == 0


def test_cli_gpus_2_reduces_through_the_rccl_hook_on_two_gpus(tmp_path, monkeypatch):
    """The hook at world > 1 — `TrueConsense -i one.bam --gpus 2` on a node with two GPUs: the split workers make a two-rank RCCL
    communicator (distributed.split_reduce_hook) and tcmi_split_step queues ncclReduce on each rank's stream; the four outputs equal
    the single-GPU command line's.  Skips on the one-GPU boxes (RCCL wants a GPU per rank)."""
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    ref, orfs = sy.make_reference(L=4000, cds=[(100, 1900), (2100, 3900)])
    sites = [(1900, "I", "A", 0.9), (1990, "I", "GT", 0.7), (2010, "D", 3, 0.6), (3000, "I", "TT", 0.95)]
    reads = sy.make_reads(ref, 6000, seed=91, indel_sites=sites)
    monkeypatch.chdir(tmp_path)
    bamwriter.write_bam("one.bam", reads, "MN", len(ref), block=3000, split_records=True)
    open("r.fa", "w").write(">MN x\n" + ref + "\n")
    head, body = sy.gff_text(orfs, seqid="MN")
    open("g.gff", "w").write(head + body)
    env = dict(os.environ, PYTHONPATH=ROOT, TCMI_SPLIT_VERBOSE="1")
    for k in ("TCMI_SPLIT_ONE_GPU", "TCMI_SPLIT_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2"])):
        argv = [sys.executable, "-m", "trueconsense_amd.TrueConsense", "-i", "one.bam", "-ref", "r.fa", "-gff", "g.gff", "-cov", "30", "-name", "S",
                "-o", tag + ".fa", "-vcf", tag + ".vcf", "-ogff", tag + ".gff", "-doc", tag + ".tsv"] + extra
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        if extra:
            assert "2-rank communicator" in r.stderr, r.stderr[-1500:]
        vcf = open(tag + ".vcf").read().split("\n")
        outs[tag] = (open(tag + ".fa").read(), open(tag + ".gff").read(), open(tag + ".tsv").read(), vcf[:1] + vcf[3:])
    assert outs["one"] == outs["two"]
