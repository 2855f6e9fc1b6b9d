"""Coverage accessors — same surface as TrueConsense/Coverage.py:1-34."""


def BuildCoverage(iDict, output):
    """Coverage.py:1-16 — "{pos}\\t{coverage}\\n" for positions 1..len(iDict)."""
    with open(output, "w") as outfile:
        outfile.write("".join("%d\t%d\n" % (i + 1, iDict[i + 1].get("coverage")) for i in range(len(iDict))))


def GetCoverage(iDict, position):
    """Coverage.py:19-34."""
    return iDict[position].get("coverage")
