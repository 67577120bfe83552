"""Host-side periphery (key/tag readers, sanitiser, merge, writers, CLI) against fixtures
captured from the real reference (tests/golden/make_periphery_golden.py).  CPU only, except
the last test, which runs the CLI with the real GPU engine."""
import base64
import contextlib
import copy
import io
import json
import os

import pytest

from conftest import load_golden

PERI = load_golden("periphery.json")
CLI = load_golden("cli_config1.json")


def run_in(tmp_path, files, binary, fn, args, kwargs):
    old = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        for name, text in files.items():
            if name in binary:
                open(name, "wb").write(base64.b64decode(text))
            else:
                with open(name, "w", newline="") as fh:
                    fh.write(text)
        before = set(os.listdir("."))
        out = io.StringIO()
        rec = {}
        try:
            with contextlib.redirect_stdout(out):
                rec["result"] = fn(*args, **kwargs)
        except Exception as e:
            rec["raises"] = type(e).__name__
            rec["message"] = str(e)
        rec["stdout"] = out.getvalue()
        rec["written_b64"] = {n: base64.b64encode(open(n, "rb").read()).decode()
                              for n in sorted(set(os.listdir(".")) - before)}
        return rec
    finally:
        os.chdir(old)


def tuples_to_lists(x):
    if isinstance(x, (list, tuple)):
        return [tuples_to_lists(v) for v in x]
    if isinstance(x, dict):
        return {k: tuples_to_lists(v) for k, v in x.items()}
    return x


@pytest.mark.parametrize("e", PERI, ids=lambda e: e["func"])
def test_periphery_matches_reference(e, tmp_path):
    from tagdigger_amd import tagdigger_fun as tf
    got = run_in(tmp_path, e["files"], e["binary_files"], getattr(tf, e["func"]),
                 copy.deepcopy(e["args"]), copy.deepcopy(e["kwargs"]))
    if "raises" in e:
        assert got.get("raises") == e["raises"], got
    else:
        assert "raises" not in got, got
        assert tuples_to_lists(got["result"]) == e["result"]
    assert got["stdout"] == e["stdout"]
    assert got["written_b64"] == e["written_b64"]


def _stage_cli(tmp_path):
    d = str(tmp_path)
    open(os.path.join(d, "key.csv"), "w").write(CLI["key_csv"])
    open(os.path.join(d, "tags.csv"), "w").write(CLI["tags_csv"])
    open(os.path.join(d, "lib.fq.gz"), "wb").write(base64.b64decode(CLI["fastq_gz_b64"]))
    return d


def _run_cli(tmp_path):
    from tagdigger_amd import tagdigger_script
    d = _stage_cli(tmp_path)
    out = io.StringIO()
    old = os.getcwd()
    try:
        with contextlib.redirect_stdout(out):
            tagdigger_script.main(CLI["argv"] + ["-w", d])
    finally:
        os.chdir(old)
    return d, out.getvalue()


def test_cli_plumbing_with_oracle_counts(tmp_path, monkeypatch):
    """Everything around the hot path, byte for byte, with the CPU oracle standing in for the GPU."""
    from oracle import c_oracle
    from tagdigger_amd import tagdigger_fun as tf
    import numpy as np

    def oracle_counts(f, b, t, cutsite="TGCAG", device=0, as_array=False):
        m = c_oracle.find_tags_fastq(f, b, t, cutsite=cutsite)
        return np.array(m, dtype=np.uint64) if as_array else m
    monkeypatch.setattr(tf, "find_tags_fastq", oracle_counts)
    d, stdout = _run_cli(tmp_path)
    assert open(os.path.join(d, "counts.csv"), "rb").read() == base64.b64decode(CLI["counts_csv_b64"])
    assert open(os.path.join(d, "geno.csv"), "rb").read() == base64.b64decode(CLI["geno_csv_b64"])
    # the reference also prints progress lines from inside find_tags_fastq; everything else is identical
    ref_lines = [ln for ln in CLI["stdout"].splitlines() if not ln.startswith("Reads: ")]
    assert stdout.splitlines() == ref_lines


def test_cli_flag_rules():
    from tagdigger_amd import tagdigger_script as s
    with pytest.raises(Exception, match="Need either restriction enzyme"):
        s.main(["--MergedTags", "t", "-b", "k", "-o", "o"])
    with pytest.raises(Exception, match="do not match"):
        s.main(["-e", "PstI", "-c", "CATGG", "--MergedTags", "t", "-b", "k", "-o", "o"])
    with pytest.raises(Exception, match="unexpected characters"):
        s.main(["-c", "TGXAG", "--MergedTags", "t", "-b", "k", "-o", "o"])
    with pytest.raises(Exception, match="Exactly one tag format"):
        s.main(["-e", "PstI", "-b", "k", "-o", "o"])
    with pytest.raises(Exception, match="Exactly one tag format"):
        s.main(["-e", "PstI", "--MergedTags", "t", "--RowTags", "r", "-b", "k", "-o", "o"])
    with pytest.raises(Exception, match="all three files for Stacks"):
        s.main(["-e", "PstI", "--StacksTags", "t", "-b", "k", "-o", "o"])


@pytest.mark.gpu
def test_cli_end_to_end_gpu(tmp_path):
    """The drop-in command line on an MI355X: counts.csv and geno.csv byte-identical to the reference's."""
    d, stdout = _run_cli(tmp_path)
    assert open(os.path.join(d, "counts.csv"), "rb").read() == base64.b64decode(CLI["counts_csv_b64"])
    assert open(os.path.join(d, "geno.csv"), "rb").read() == base64.b64decode(CLI["geno_csv_b64"])
    # ... and its stdout line for line (a 3 000-read library: the reference prints no progress line here either;
    # tests/test_progress.py compares those on longer inputs)
    assert stdout.splitlines() == CLI["stdout"].splitlines()


@pytest.mark.gpu
def test_cli_several_devices(tmp_path):
    """--td-devices: one process per listed GPU (the same GPU twice here: the rehearsal form, over gloo), libraries
    dealt over the ranks, K3 + one all-reduce of the samples x tags device matrix, rank 0 writes: same bytes."""
    from tagdigger_amd import tagdigger_script
    d = _stage_cli(tmp_path)
    old = os.getcwd()
    try:
        tagdigger_script.main(CLI["argv"] + ["-w", d, "--td-devices", "0,0"])
    finally:
        os.chdir(old)
    assert open(os.path.join(d, "counts.csv"), "rb").read() == base64.b64decode(CLI["counts_csv_b64"])
    assert open(os.path.join(d, "geno.csv"), "rb").read() == base64.b64decode(CLI["geno_csv_b64"])
