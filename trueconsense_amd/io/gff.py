"""Minimal GFF3 reader giving what the reference takes from AminoExtract's reader
(indexing.py:22-36; TrueConsense.py:238-241): an object with `.header.raw_text` (the leading
comment lines, verbatim) and `.df`, a pandas DataFrame with the nine GFF3 columns plus one
column per attribute key (AminoExtract 0.4.1 splits the attributes column that way; its exact
column set is not pinned here — SURVEY §8-c — but stage B only reads start / end / strand and
the writer folds every extra column back into `attributes`, Outputs.py:32-58)."""
from __future__ import annotations

GFF3_COLUMNS = ["seqid", "source", "type", "start", "end", "score", "strand", "phase", "attributes"]


class GFFHeader:
    def __init__(self, raw_text):
        self.raw_text = raw_text


class GFFDataFrame:
    """`.header`, and `.df` — the DataFrame, made when somebody asks for it: importing pandas costs the single-sample command line
    more than its GPU work (0.6 s of a 0.85-s process), and the command line itself only needs `index_dict`."""

    def __init__(self, header, rows, cols):
        self.header, self.rows, self.columns = header, rows, cols
        self._df = None

    @property
    def df(self):
        if self._df is None:
            import pandas as pd
            self._df = pd.DataFrame(self.rows, columns=self.columns)
        return self._df

    @df.setter
    def df(self, value):
        self._df = value

    def index_dict(self, seqid=None):
        """What `df["seqid"] = seqid; df.to_dict("index")` gives (TrueConsense.py:238-241) — {row number: {column: value}} with every
        column in every row, a missing attribute as NaN (pandas' fill: it prints as "nan", as upstream's writer prints it) — without
        the DataFrame.  (A DataFrame somebody has asked for, and possibly edited, has the last word.)"""
        if self._df is not None:
            d = self._df.copy()
            if seqid is not None:
                d["seqid"] = seqid
            return d.to_dict("index")
        out = {}
        for i, r in enumerate(self.rows):
            row = {c: r.get(c, float("nan")) for c in self.columns}
            if seqid is not None:
                row["seqid"] = seqid
            out[i] = row
        return out


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
    return GFFDataFrame(GFFHeader("".join(head)), rows, cols)
