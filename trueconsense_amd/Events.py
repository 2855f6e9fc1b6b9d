"""Indel events — same surface as TrueConsense/Events.py.

The per-position threshold tests ((I/cov)*100 > 55, Events.py:29-36; (X/cov)*100 >= 15,
Events.py:85-106) are evaluated by the HIP call kernel (fp64, no contraction); the modal insert
token of each candidate (Events.py:47-82) comes from one native host sweep over the reads under
pysam's default pileup filters.
"""
from __future__ import annotations

import re

import numpy as np

from . import _ffi, _state
from .engine import BamFile, LazyBam, modal_tokens

_TOKEN = re.compile(r"(\d)([a-zA-Z]+)")          # Events.py:75


def _parse_token(tok):
    """Events.py:75-80 on the modal token -> (bases, size_str) or (None, None)."""
    if not tok:
        return None, None
    m = _TOKEN.search(tok)
    if not m:
        return None, None
    return m.group(2), m.group(1)


def _tokens_at(bam, positions):
    """{pos: modal token or None} for 1-based positions."""
    if not positions:
        return {}
    if isinstance(bam, LazyBam):
        bam = bam.get()
    if isinstance(bam, BamFile) or isinstance(bam, dict):
        return {p: t for p, (t, _) in modal_tokens(bam, positions).items()}
    if hasattr(bam, "modal_token"):               # test double: bam.modal_token(pos) -> token or None
        return {p: bam.modal_token(p) for p in positions}
    raise TypeError("bam must be a trueconsense_amd BamFile (or a dict of flat read arrays)")


def ExtractInserts(bam, position):
    """Events.py:47-82."""
    return _parse_token(_tokens_at(bam, [position]).get(position))


def candidates_from_flags(flags):
    """1-based positions whose call record carries the insert-candidate bit."""
    return (np.nonzero(np.asarray(flags) & _ffi.F_INSCAND)[0] + 1).tolist()


def inserts_from_flags(flags, bam):
    """-> (hasinserts, {pos: {size_str: bases}} or None), Events.py:37-44."""
    cand = candidates_from_flags(flags)
    toks = _tokens_at(bam, cand)
    positions = {}
    for k in cand:
        bases, size = _parse_token(toks.get(k))
        if bases is None or size is None:
            continue
        positions[k] = {size: bases}
    if not positions:
        return False, None
    return True, positions


def ListInserts(iDict, mincov, bam):
    """Events.py:5-44."""
    counts = _state.counts_of(iDict)
    if len(counts) == 0:
        return False, None
    _, _, flags = _state.default_context().call(counts, mincov, True)
    return inserts_from_flags(flags, bam)


def MinorityDel(iDict, position):
    """Events.py:85-106 — raises ZeroDivisionError at zero coverage like the reference."""
    row = iDict[position]
    counts = np.array([[row[c] for c in _ffi.COLS]], np.int32)
    _, _, flags = _state.default_context().call(counts, 0, True)
    if flags[0] & _ffi.F_COVZERO:
        raise ZeroDivisionError("division by zero")
    return bool(flags[0] & _ffi.F_MINDEL)
