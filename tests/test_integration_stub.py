"""INTEGRATION.md section 2 shows the ctypes stub a reference maintainer would paste into tagdigger_fun.py.  This runs
that very text (extracted from the document) in a fresh interpreter -- no tagdigger_amd package, no torch: only ctypes,
libtagdig.so and ROCm's HIP runtime -- on an input of tests/golden/progress.json, and compares what it returns and
prints with what the real reference returned and printed."""
import gzip
import json
import os
import re
import subprocess
import sys

import pytest

import test_progress as tp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_the_documented_stub_is_a_drop_in(tmp_path):
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"## 2\..*?```python\n(.*?)```", doc, re.S).group(1)
    assert "def find_tags_fastq(" in code and "td_get_progress" in code
    code = code.replace('C.CDLL("libtagdig.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "tagdigger_amd", "libtagdig.so"))
    case = [c for c in tp.GOLD if c["file"].endswith("gz")][0]           # gzip by name, maxreads inside the file
    raw = tp.case_bytes(case)
    with open(tmp_path / case["file"], "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=1))
    with open(tmp_path / "case.json", "w") as fh:
        json.dump({"barcodes": case["barcodes"], "tags": case["tags"], "kwargs": case["kwargs"], "file": case["file"]}, fh)
    script = tmp_path / "stub_run.py"
    script.write_text(
        "import json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle.tagdigger_oracle import enumerate_cut_sites, combine_barcode_and_cutsite   # (the two primitives the set-up calls)\n"
        % ROOT + code +
        "\ncase = json.load(open('case.json'))\n"
        "res = find_tags_fastq(case['file'], case['barcodes'], case['tags'], **case['kwargs'])\n"
        "print('RESULT ' + json.dumps(res))\n")
    out = subprocess.run([sys.executable, str(script)], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    result = [json.loads(ln[7:]) for ln in lines if ln.startswith("RESULT ")]
    assert result and result[0] == case["counts"]
    assert [ln for ln in lines if not ln.startswith("RESULT ")] == case["stdout"]
