"""Process-wide default device context and the count-matrix carrier."""
from __future__ import annotations

import os

import numpy as np

from ._ffi import COLS
from .engine import Context

_ctx = None


def default_context():
    """One Context per process: device TCMI_DEVICE, else LOCAL_RANK, else 0."""
    global _ctx
    if _ctx is None:
        dev = int(os.environ.get("TCMI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ctx = Context(dev)
    return _ctx


def set_default_context(ctx):
    global _ctx
    _ctx = ctx


class IndexDict(dict):
    """{pos: {"coverage","A","T","C","G","X","I"}} exactly as IndexDF.to_dict("index") gives
    (TrueConsense.py:237), carrying the int32 [L,7] matrix it was built from so stage B does not
    have to re-assemble it."""

    def __init__(self, counts):
        counts = np.ascontiguousarray(counts, np.int32)
        super().__init__((i + 1, dict(zip(COLS, row))) for i, row in enumerate(counts.tolist()))
        self.counts = counts


def counts_of(iDict):
    """Any iDict -> int32 [L,7]; keys must be 1..L (what BuildIndex produces)."""
    if isinstance(iDict, IndexDict):
        return iDict.counts
    L = len(iDict)
    out = np.empty((L, 7), np.int32)
    for i in range(L):
        row = iDict[i + 1]                      # KeyError like the reference on a gapped index
        out[i] = [row[c] for c in COLS]
    return out
