"""Coverage accessors — same surface as TrueConsense/Coverage.py:1-34."""


def BuildCoverage(iDict, output):
    """Coverage.py:1-16 — "{pos}\\t{coverage}\\n" for positions 1..len(iDict)."""
    counts = getattr(iDict, "counts", None)
    if counts is not None:                      # the matrix behind an IndexDict: no per-position dictionaries
        cov = counts[:, 0].tolist()
        text = "".join("%d\t%d\n" % (i + 1, c) for i, c in enumerate(cov))
    else:
        text = "".join("%d\t%d\n" % (i + 1, iDict[i + 1].get("coverage")) for i in range(len(iDict)))
    with open(output, "w") as outfile:
        outfile.write(text)


def GetCoverage(iDict, position):
    """Coverage.py:19-34."""
    cov = getattr(iDict, "coverage", None)
    return cov(position) if cov is not None else iDict[position].get("coverage")
