"""Minimal GFF3 reader giving what the reference takes from AminoExtract's reader
(indexing.py:22-36; TrueConsense.py:238-241): an object with `.header.raw_text` (the leading
comment lines, verbatim) and `.df`, a pandas DataFrame with the nine GFF3 columns plus one
column per attribute key (AminoExtract 0.4.1 splits the attributes column that way; its exact
column set is not pinned here — SURVEY §8-c — but stage B only reads start / end / strand and
the writer folds every extra column back into `attributes`, Outputs.py:32-58)."""
from __future__ import annotations

import pandas as pd

GFF3_COLUMNS = ["seqid", "source", "type", "start", "end", "score", "strand", "phase", "attributes"]


class GFFHeader:
    def __init__(self, raw_text):
        self.raw_text = raw_text


class GFFDataFrame:
    def __init__(self, header, df):
        self.header, self.df = header, df


def read_gff(path):
    head, rows = [], []
    in_head = True
    with open(path) as fh:
        for line in fh:
            if line.startswith("#"):
                if in_head:
                    head.append(line)
                continue
            if not line.strip():
                continue
            in_head = False
            f = line.rstrip("\n").split("\t")
            if len(f) != 9:
                raise ValueError("%s: expected 9 tab-separated GFF3 columns, got %d" % (path, len(f)))
            row = dict(zip(GFF3_COLUMNS, f))
            row["start"], row["end"] = int(f[3]), int(f[4])
            for item in f[8].split(";"):
                if "=" in item:
                    k, v = item.split("=", 1)
                    row[k] = v
            rows.append(row)
    cols = list(GFF3_COLUMNS)
    for r in rows:
        for k in r:
            if k not in cols:
                cols.append(k)
    df = pd.DataFrame(rows, columns=cols)
    return GFFDataFrame(GFFHeader("".join(head)), df)
