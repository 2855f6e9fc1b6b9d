"""Stage B — same surface as TrueConsense/Sequences.py:168-322 (BuildConsensus).

Per position, ranking / ambiguity / thresholds / case run in the HIP call kernel
(csrc/call.hip); the sequential part (deletion runs, triplet rule, ORF membership on the
evolving GFF, insert splice, GFF correction — Sequences.py:179-322 with ORFs.py:1-192) is the
native O(L) host walk tcmi_consensus_walk over those records.
"""
from __future__ import annotations

import copy

import numpy as np

from . import _state
from .engine import consensus_walk
from .Events import inserts_from_flags


def _orf_arrays(GFFdict):
    keys = list(GFFdict.keys())
    start = np.array([int(GFFdict[k]["start"]) for k in keys], np.int64)
    end = np.array([int(GFFdict[k]["end"]) for k in keys], np.int64)
    plus = np.array([GFFdict[k].get("strand") == "+" for k in keys], np.uint8)
    return keys, start, end, plus


def _insert_arrays(insertpositions):
    if not insertpositions:
        return [], [], []
    pos = sorted(insertpositions)
    shift, seqs = [], []
    for p in pos:
        size, bases = next(iter(insertpositions[p].items()))
        shift.append(int(size))                  # ORFs.py:141-145: int(size_str), last digit only
        seqs.append(str(bases))
    return pos, shift, seqs


def consensus_from_records(plain, alt, flags, GFFdict, insertpositions, includeINS):
    """call records + GFF rows + accepted inserts -> (consensus, newGffdict)."""
    keys, start, end, plus = _orf_arrays(GFFdict)
    pos, shift, seqs = _insert_arrays(insertpositions)
    cons, ns, ne = consensus_walk(plain, alt, flags, start, end, plus, pos, shift, seqs, includeINS)
    newGffdict = copy.deepcopy(GFFdict)
    for k, s, e in zip(keys, ns.tolist(), ne.tolist()):
        newGffdict[k]["start"], newGffdict[k]["end"] = s, e
    return cons, newGffdict


def BuildConsensus(mincov, iDict, GFFdict, IncludeAmbig, bam, includeINS):
    """Sequences.py:168-322."""
    counts = _state.counts_of(iDict)
    if len(counts) == 0:
        return "", copy.deepcopy(GFFdict)
    plain, alt, flags = _state.default_context().call(counts, mincov, IncludeAmbig)
    _, insertpositions = inserts_from_flags(flags, bam)
    return consensus_from_records(plain, alt, flags, GFFdict, insertpositions, includeINS)
