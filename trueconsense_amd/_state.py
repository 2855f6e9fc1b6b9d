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
    """{pos: {"coverage","A","T","C","G","X","I"}} as IndexDF.to_dict("index") gives it (TrueConsense.py:237), carrying the
    int32 [L,7] matrix it stands for so that stage B does not have to re-assemble it.  The 30 000 row dictionaries are made
    when somebody asks for them (building them all up front was 26 of the command line's 33 ms); the accessors of this
    package (Coverage, Events, Outputs) read the matrix."""

    def __init__(self, counts):
        super().__init__()
        self.counts = np.ascontiguousarray(counts, np.int32)
        self._full = False

    def _row(self, pos):
        return dict(zip(COLS, self.counts[pos - 1].tolist()))

    def _fill(self):
        if not self._full:
            for i, row in enumerate(self.counts.tolist()):
                if not dict.__contains__(self, i + 1):
                    dict.__setitem__(self, i + 1, dict(zip(COLS, row)))
            self._full = True

    def __missing__(self, pos):                 # dict.__getitem__ calls this for a key it does not hold yet
        if isinstance(pos, (int, np.integer)) and 1 <= pos <= len(self.counts):
            row = self._row(int(pos))
            dict.__setitem__(self, int(pos), row)
            return row
        raise KeyError(pos)

    def get(self, pos, default=None):
        try:
            return self[pos]
        except KeyError:
            return default

    def coverage(self, pos):
        """iDict[pos]["coverage"] without the row dictionary."""
        if not 1 <= pos <= len(self.counts):
            raise KeyError(pos)
        return int(self.counts[pos - 1, 0])

    def __len__(self):
        return len(self.counts)

    def __contains__(self, pos):
        return isinstance(pos, (int, np.integer)) and 1 <= pos <= len(self.counts)

    def __iter__(self):
        return iter(range(1, len(self.counts) + 1))

    def keys(self):
        return range(1, len(self.counts) + 1)

    def values(self):                           # (in key order, like the dictionary this stands for — not in the order of first use)
        return [self[k] for k in range(1, len(self.counts) + 1)]

    def items(self):
        return [(k, self[k]) for k in range(1, len(self.counts) + 1)]

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        return "IndexDict(%d positions)" % len(self.counts)


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
