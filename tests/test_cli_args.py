"""Command-line surface (TrueConsense.py:25-220 of the reference): flags, required-ness and exit codes.
No GPU: only argument handling runs."""
import pytest

from trueconsense_amd import TrueConsense as cli


def _files(tmp_path):
    p = {}
    for name in ("x.bam", "r.fa", "r.fasta", "f.gff", "o.csv.gz", "bad.txt"):
        (tmp_path / name).write_text("x")
        p[name] = str(tmp_path / name)
    return p


def test_all_flags_parse(tmp_path):
    f = _files(tmp_path)
    a = cli.GetArgs(["-i", f["x.bam"], "-ref", f["r.fa"], "-gff", f["f.gff"], "-cov", "30", "-name", "S", "-o", "out.fa",
                     "-vcf", "v.vcf", "-doc", "d.tsv", "-ogff", "g.gff", "-t", "3", "-noambig", "--index-override", f["o.csv.gz"]])
    assert (a.input, a.reference, a.features, a.coverage_level, a.samplename, a.output) == \
        (f["x.bam"], f["r.fa"], f["f.gff"], 30, "S", "out.fa")
    assert (a.variants, a.depth_of_coverage, a.output_gff, a.threads, a.noambiguity, a.index_override) == \
        ("v.vcf", "d.tsv", "g.gff", 3, True, f["o.csv.gz"])
    b = cli.GetArgs(["--input", f["x.bam"], "--reference", f["r.fasta"], "--features", f["f.gff"], "--coverage-level", "5",
                     "--samplename", "S", "--output", "o.fa"])
    assert b.noambiguity is False and b.variants is None and b.coverage_level == 5


def test_exit_codes_match_the_reference(tmp_path, capsys, monkeypatch):
    f = _files(tmp_path)
    base = ["-ref", f["r.fa"], "-gff", f["f.gff"], "-cov", "30", "-name", "S", "-o", "out.fa"]
    with pytest.raises(SystemExit) as e:                      # missing BAM: exit -1 (TrueConsense.py:34-35)
        cli.GetArgs(["-i", str(tmp_path / "nope.bam")] + base)
    assert e.value.code == -1 and "is not a file" in capsys.readouterr().out
    with pytest.raises(SystemExit) as e:                      # other missing files: exit 1
        cli.GetArgs(["-i", f["x.bam"], "-ref", str(tmp_path / "nope.fa"), "-gff", f["f.gff"], "-cov", "30", "-name", "S", "-o", "o"])
    assert e.value.code == 1
    for argv in (["-i", f["bad.txt"]] + base,                 # wrong extensions: parser.error -> 2
                 ["-i", f["x.bam"], "-ref", f["bad.txt"], "-gff", f["f.gff"], "-cov", "30", "-name", "S", "-o", "o"],
                 ["-i", f["x.bam"], "-ref", f["r.fa"], "-gff", f["bad.txt"], "-cov", "30", "-name", "S", "-o", "o"],
                 ["-i", f["x.bam"]] + base + ["--index-override", f["bad.txt"]],
                 ["-i", f["x.bam"], "-ref", f["r.fa"], "-gff", f["f.gff"], "-name", "S", "-o", "o"]):   # -cov is required
        with pytest.raises(SystemExit) as e:
            cli.GetArgs(argv)
        assert e.value.code == 2
    monkeypatch.setattr("sys.argv", ["TrueConsense"])
    with pytest.raises(SystemExit) as e:                      # no arguments at all: message + exit 1 (:216-220)
        cli.main([])
    assert e.value.code == 1


def test_gpus_children_get_the_parsed_arguments_not_the_typed_ones(tmp_path, monkeypatch):
    """--gpus N deals the work to N child processes.  argparse takes abbreviations (`--gpu`, `--bat`, `--dev`: the reference's parser
    does too, TrueConsense.py:25-30), so a child's command line is built from the parsed namespace: it never carries the parent's
    --gpus / --batch / --device again, and always an explicit `--gpus 1` — a child can never deal itself out once more."""
    f = _files(tmp_path)
    man = tmp_path / "m.tsv"
    man.write_text("\n".join("%s\tS%d\t%s" % (f["x.bam"], k, tmp_path / ("o%d.fa" % k)) for k in range(5)) + "\n")
    seen = {}

    def fake_spawn(cmds, envs):
        seen["cmds"], seen["envs"] = cmds, envs
        seen["rows"] = [open(c[4]).read().count("\n") for c in cmds if c[3] == "--batch"]   # (the shards live as long as the children)
        return 0
    monkeypatch.setattr(cli, "_spawn", fake_spawn)
    cli.main(["--gpu", "2", "--bat", str(man), "--dev", "5", "-ref", f["r.fa"], "-gff", f["f.gff"], "-cov", "12", "-noambig", "-t", "3"])
    assert len(seen["cmds"]) == 2
    for k, cmd in enumerate(seen["cmds"]):
        tail = cmd[3:]
        assert tail[0] == "--batch" and tail[2:4] == ["--device", str(k)]
        rest = tail[4:]
        assert rest == ["-ref", f["r.fa"], "-gff", f["f.gff"], "-cov", "12", "-t", "3", "-noambig", "--gpus", "1"], rest
        child = cli.GetArgs(tail)                                # what the child itself will understand
        assert child.gpus == 1 and child.batch == tail[1] and child.device == k and child.coverage_level == 12 and child.noambiguity
        assert seen["rows"] == [3, 2]
    # ONE file over N GPUs: the split workers get the single-sample flags, again with --gpus 1
    cli.main(["--gpu=3", "--inp", f["x.bam"], "-ref", f["r.fa"], "-gff", f["f.gff"], "-cov", "30", "--samplen", "S", "-o", "o.fa", "-vcf", "v.vcf"])
    assert len(seen["cmds"]) == 3 and [e["RANK"] for e in seen["envs"]] == ["0", "1", "2"]
    for cmd in seen["cmds"]:
        assert cmd[1:3] == ["-m", "trueconsense_amd.split_main"]
        child = cli.GetArgs(cmd[3:])
        assert child.gpus == 1 and child.input == f["x.bam"] and child.samplename == "S" and child.variants == "v.vcf" and child.batch is None


def test_spawn_ends_the_others_when_one_child_fails(monkeypatch):
    """A child that dies early would leave the others waiting in a collective: the rest are terminated after a short grace."""
    import sys
    import time
    monkeypatch.setenv("TCMI_SPAWN_GRACE", "0.5")
    t0 = time.monotonic()
    rc = cli._spawn([[sys.executable, "-c", "import sys; sys.exit(3)"], [sys.executable, "-c", "import time; time.sleep(120)"]], [None, None])
    assert rc >= 3 and time.monotonic() - t0 < 30
    assert cli._spawn([[sys.executable, "-c", "pass"]] * 2, [None, None]) == 0


def test_console_scripts_of_the_reference_are_declared():
    """pyproject.toml declares `trueconsense` and `TrueConsense` (the reference's pyproject.toml:38-40) and both resolve to a callable."""
    import importlib
    import os
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as fh:
        scripts = tomli.load(fh)["project"]["scripts"]
    assert set(scripts) == {"trueconsense", "TrueConsense"}
    for target in scripts.values():
        mod, fn = target.split(":")
        assert callable(getattr(importlib.import_module(mod), fn))


def test_gff_index_dict_without_pandas_is_what_the_dataframe_gives(tmp_path):
    """The command line takes its GFF rows from `Gffindex(...).index_dict(seqid)` — what upstream gets from `df["seqid"] = name;
    df.to_dict("index")` (TrueConsense.py:238-241) — without importing pandas (0.6 s of the single-sample process).  Same keys in the
    same order, same values, a missing attribute NaN either way, and the same GFF text line from the writer."""
    import math
    import subprocess
    import sys
    from trueconsense_amd import Outputs
    from trueconsense_amd.indexing import Gffindex
    texts = ["##gff-version 3\n#!x y\nS\tsrc\tCDS\t10\t90\t.\t+\t0\tID=a;Name=orfA\nS\tsrc\tCDS\t100\t190\t0.5\t-\t2\tID=b;product=p q;Note=n\n"
             "S\tsrc\tgene\t5\t300\t.\t+\t.\tName=g\n",
             "##gff-version 3\n",
             "S\tx\tCDS\t1\t9\t.\t+\t0\tID=o0\n"]
    for k, text in enumerate(texts):
        p = tmp_path / ("g%d.gff" % k)
        p.write_text(text)
        lean = Gffindex(str(p)).index_dict(seqid="NAME")
        g = Gffindex(str(p))
        df = g.df
        df["seqid"] = "NAME"
        want = df.to_dict("index")
        assert list(lean) == list(want)
        for i in want:
            assert list(lean[i]) == list(want[i])
            for c in want[i]:
                a, b = lean[i][c], want[i][c]
                assert (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b)) or (a == b and type(a) is type(b)), (i, c, a, b)
            assert Outputs._gff_line(lean[i]) == Outputs._gff_line(want[i])
        assert g.index_dict(seqid="OTHER")[0]["seqid"] == "OTHER" if want else True      # (a DataFrame that was asked for has the last word)
    r = subprocess.run([sys.executable, "-c", "import sys, trueconsense_amd.TrueConsense, trueconsense_amd.split_main; print('pandas' in sys.modules)"],
                       capture_output=True, text=True, cwd=str(tmp_path.parent), env=dict(__import__("os").environ, PYTHONPATH=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
    assert r.stdout.strip() == "False", r.stdout + r.stderr
