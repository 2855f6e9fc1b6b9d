"""Worker of `TrueConsense -i x.bam ... --gpus N` (BASELINE configs[4]: ONE BAM file over N GPUs of a node): one process per GPU,
started by TrueConsense.main BEFORE anything touches a GPU, with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in its
environment.  Every rank decodes, packs and tallies its contiguous range of the file's BGZF blocks; one reduce of the count matrix
(RCCL under "nccl"); rank 0 calls, votes on the insert candidates' entries the ranks send it, and writes the command line's outputs
(TrueConsense.py:212-264) exactly as the single-GPU command line does — through Outputs.WriteOutputs / Coverage.BuildCoverage.

    TCMI_SPLIT_BACKEND=gloo TCMI_SPLIT_ONE_GPU=1   rehearsal: all ranks on GPU 0, the exchange over gloo (RCCL wants a GPU per rank)
"""
from __future__ import annotations

import os
import sys


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    one_gpu = os.environ.get("TCMI_SPLIT_ONE_GPU") == "1"
    device = 0 if one_gpu else local
    os.environ["TCMI_DEVICE"] = str(device)
    import torch
    import torch.distributed as dist
    from . import _state
    from . import distributed as td
    from .Coverage import BuildCoverage
    from .indexing import Gffindex
    from .io import fasta
    from .Outputs import WriteOutputs
    from .TrueConsense import GetArgs
    a = GetArgs([x for x in argv])
    backend = os.environ.get("TCMI_SPLIT_BACKEND", "nccl")
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    rc = 0
    comm = None
    try:
        # the step's one exchange: the C hook over an RCCL communicator of this job's own (what tests/test_rccl.py and bench.py --split-bam
        # --from-file run); a rehearsal of several ranks on one GPU (gloo) keeps the process group's reduce
        comm, user, hook_kind = td.split_reduce_hook(rank, world, rccl=backend == "nccl" and (world == 1 or not one_gpu))
        if os.environ.get("TCMI_SPLIT_VERBOSE") == "1" and rank == 0:
            print("TrueConsense --gpus %d: the count matrices are summed by %s" % (world, hook_kind), file=sys.stderr)
        IndexGff = Gffindex(a.features)
        gffdict = IndexGff.index_dict(seqid=a.samplename)
        rows = [{"start": int(r["start"]), "end": int(r["end"]), "strand": r.get("strand")} for r in gffdict.values()]
        _, refseq = fasta.read_first_record(a.reference)
        parts = td.consensus_split_bamfile(a.input, len(refseq), rows, a.coverage_level, a.noambiguity is False, a.samplename, rank, world,
                                           device=device, return_parts=True, rccl_user=user)
        if rank == 0:
            _, counts, toks = parts
            index = _state.IndexDict(counts)
            if a.depth_of_coverage is not None:
                BuildCoverage(index, a.depth_of_coverage)
            WriteOutputs(a.coverage_level, index, gffdict, td._Tokens(toks), a.noambiguity is False, a.variants, a.samplename, a.reference,
                         a.output_gff, IndexGff.header, a.output)
    except Exception as e:                                           # noqa: BLE001 — every rank must reach the group's teardown
        print("TrueConsense --gpus (rank %d): %s" % (rank, e), file=sys.stderr)
        rc = 1
    finally:
        try:
            td.split_reduce_hook_close(comm)
        except Exception:                                            # noqa: BLE001
            pass
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
