#!/usr/bin/env python3
"""Generate tests/golden/splitter.json from the REAL reference (build container only; data only,
never reference code): findAdapterSeq values, build_adapter_tree messages, and barcodeSplitter
end to end (input FASTQ bytes -> output files + stdout).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_splitter_golden.py [/root/reference]
"""
import base64
import contextlib
import gzip
import io
import json
import os
import random
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import tagdigger_fun as ref  # noqa: E402

ADAPTERS = {k: [list(x) for x in v] for k, v in ref.adapters.items()}
ADAPTERS["repeat"] = [["CCG^G", "CCGCCGCCGAT"], ["CTGCA^G", "[barcode]CTGCACTGCAAGAT"]]     # beginnings that overlap
ADAPTERS["short"] = [["AT^CG", "GGTT"], ["G^C", "[barcode]AC"]]


def adapter_of(name):
    return [tuple(x) for x in ADAPTERS[name]]


def find_cases():
    out = []
    g = "AT" * 10
    table = [("PstI-MspI-Hall", ["AACG", "TTGACC"], 0, "TGCAG", [
        "AACGTGCAG" + g, "AACGTGCAG" + g[:8] + "CCGG" + g, "AACGTGCAG" + g[:8] + "CTGCAG" + g,
        "AACGTGCAG" + g[:4] + "CTGCAG" + g[:3] + "CCGG", "AACGTGCAG" + g + "CCGCTCAG", "AACGTGCAG" + g + "CCG",
        "AACGTGCAG" + g + "CCGC", "AACGTGCAG" + g + "CTGCACGTTAGA", "AACGTGCAGCCGG", "AACGTGCAGCTGCAG", "AACGTGCAG",
        "AACGTGCAG" + g + "CCGN", "AACGTGCAG" + g + "NCGCTCAG", "AACGTGCAGCCG" + "CTCAGGCATCACTCGATTCCTCCGTCGTATGCCGTCTTCTGCTTG",
        "AACGTGCAGCCG" + "CTCAGGCATCACTCGATTCCTCCGTCGTATGCCGTCTTCTGCTTGA"]),
        ("PstI-MspI-Hall", ["AACG", "TTGACC"], 1, "TGCAG", [
            "TTGACCTGCAG" + g + "CTGCAGGTCAAAGATCGG", "TTGACCTGCAG" + g + "CTGCAGGTCAA", "TTGACCTGCAG" + g + "CTGCACGTT"]),
        ("repeat", ["AC", "GGT"], 0, "TGCAG", [
            "ACTGCAG" + g + "CCGCCG", "ACTGCAG" + g + "CCGC", "ACTGCAG" + g + "CCGCCGCCGA", "ACTGCAG" + g + "CTGCAGTCTGCACTG",
            "ACTGCAG" + g + "CTGCAGT", "ACTGCAG" + g + "CCGCCGC"]),
        ("short", ["A"], 0, "C", ["ACGGGGATG", "ACGGGGATGG", "ACTTTTGA", "ACTTATCGTT", "ACTTGCTT", "ACGC", "AC"]),
    ]
    for name, barcodes, bi, cutsite, reads in table:
        ad = adapter_of(name)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trees = ref.build_adapter_tree(ad, barcodes)
        site0, site1 = ad[0][0].replace("^", ""), ad[1][0].replace("^", "")
        start = len(barcodes[bi]) + len(cutsite)
        out.append({"adapter": name, "barcodes": barcodes, "barcode": bi, "cutsite": cutsite, "stdout": buf.getvalue(),
                    "reads": reads, "values": [ref.findAdapterSeq(r, trees[bi], site0, site1, start) for r in reads]})
    return out


def run_splitter(fastq, barcodes, cutsite, adapter_name, maxreads=None, gz=False):
    with tempfile.TemporaryDirectory() as d:
        old = os.getcwd()
        os.chdir(d)
        try:
            name = "in.fq.gz" if gz else "in.fq"
            with open(name, "wb") as fh:
                fh.write(gzip.compress(fastq, mtime=0) if gz else fastq)
            outs = ["out%d.fq" % i for i in range(len(barcodes))]
            kw = {"cutsite": cutsite, "adapter": adapter_of(adapter_name)}
            if maxreads is not None:
                kw["maxreads"] = maxreads
            buf = io.StringIO()
            rec = {}
            try:
                with contextlib.redirect_stdout(buf):
                    ref.barcodeSplitter(name, barcodes, outs, **kw)
            except Exception as e:
                rec["raises"] = type(e).__name__
                rec["message"] = str(e)
            rec["stdout"] = buf.getvalue()
            rec["outputs_b64"] = [base64.b64encode(open(o, "rb").read()).decode() if os.path.exists(o) else None for o in outs]
        finally:
            os.chdir(old)
    return {"fastq_b64": base64.b64encode(fastq).decode(), "gz": gz, "barcodes": barcodes, "cutsite": cutsite,
            "adapter": adapter_name, "maxreads": maxreads, **rec}


def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def random_fastq(rng, barcodes, cutsite, adapter_name, nreads, style):
    ad = adapter_of(adapter_name)
    site0, site1 = ad[0][0].replace("^", ""), ad[1][0].replace("^", "")
    recs = []
    for i in range(nreads):
        kind = rng.randrange(10)
        bc = rng.choice(barcodes)
        body = rnd_seq(rng, rng.randrange(5, 60))
        if kind == 0:
            seq = rnd_seq(rng, rng.randrange(0, 50))                       # no barcode (mostly)
        elif kind == 1:
            seq = bc + cutsite + body + site0 + rnd_seq(rng, rng.randrange(0, 12))
        elif kind == 2:
            seq = bc + cutsite + body + site1 + rnd_seq(rng, rng.randrange(0, 12))
        elif kind == 3:
            full = ad[0][0].replace("^", "")[:ad[0][0].find("^")] + ad[0][1]
            seq = bc + cutsite + body + full[:rng.randrange(1, len(full) + 1)]
        elif kind == 4:
            full = ad[1][0][:ad[1][0].find("^")] + ad[1][1].replace("[barcode]", ref.reverseComplement(bc))
            seq = bc + cutsite + body + full[:rng.randrange(1, len(full) + 1)]
        elif kind == 5:
            seq = bc + cutsite + body + "N" + rnd_seq(rng, 5)
        elif kind == 6:
            seq = (bc + cutsite + body).lower()
        elif kind == 7:
            seq = bc + cutsite[:rng.randrange(0, len(cutsite) + 1)]        # truncated
        else:
            seq = bc + cutsite + body
        qual = "".join(chr(33 + rng.randrange(40)) for _ in seq)
        if style == "ragged" and rng.random() < 0.2:
            qual = qual[:rng.randrange(0, len(qual) + 1)]                  # quality shorter than the sequence
        head = "@r%d %s" % (i, rnd_seq(rng, 3, "xyz:/"))
        plus = "+" if rng.random() < 0.7 else "+" + head[1:]
        pad = (lambda s: s)
        if style == "blanks":
            pad = lambda s: rng.choice(["", " ", "\t"]) + s + rng.choice(["", " ", "\t "])
        recs.append((pad(head), pad(seq), pad(plus), pad(qual)))
    nl = {"lf": "\n", "crlf": "\r\n", "cr": "\r"}.get(style, "\n")
    text = "".join(nl.join(r) + nl for r in recs)
    if style == "nofinal":
        text = text.rstrip("\n")
    if style == "partial":
        text += "@tail\nACGT\n"
    return text.encode("ascii")


def main():
    rng = random.Random(20260)
    cases = []
    bcs = ["AACG", "TTGACC", "CGT", "GATTACAG"]
    for style in ("lf", "crlf", "cr", "blanks", "ragged", "nofinal", "partial"):
        cases.append(run_splitter(random_fastq(rng, bcs, "TGCAG", "PstI-MspI-Hall", 60, style), bcs, "TGCAG", "PstI-MspI-Hall"))
    cases.append(run_splitter(random_fastq(rng, bcs, "TGCAT", "NsiI-MspI-Clark", 80, "lf"), bcs, "TGCAT", "NsiI-MspI-Clark", gz=True))
    cases.append(run_splitter(random_fastq(rng, bcs, "TGCAG", "PstI-MspI-Poland", 80, "lf"), bcs, "TGCAG", "PstI-MspI-Poland", maxreads=37))
    cases.append(run_splitter(random_fastq(rng, bcs, "TGCAG", "PstI-MspI-Hall", 10, "lf"), bcs, "TGCAG", "PstI-MspI-Hall", maxreads=0.5))
    cases.append(run_splitter(random_fastq(rng, ["AC", "GGT"], "TGCAG", "repeat", 120, "lf"), ["AC", "GGT"], "TGCAG", "repeat"))
    cases.append(run_splitter(random_fastq(rng, ["A", "CC"], "C", "short", 120, "lf"), ["A", "CC"], "C", "short"))
    cases.append(run_splitter(b"", bcs, "TGCAG", "PstI-MspI-Hall"))
    cases.append(run_splitter(b"@a\nAACGTGCAGTTTT\n+\nIIIIIIIIIIIII", bcs, "TGCAG", "PstI-MspI-Hall"))
    cases.append(run_splitter(b"\n@a\nAACGTGCAGTTTT\n+\nIIIIIIIIIIIII\n@b\nTTGACCTGCAGAAAAAA\n+\nJJJJJJJJJJJJJJJJJ\n", bcs, "TGCAG", "PstI-MspI-Hall"))
    cases.append(run_splitter(b"@a\nAACGTGCAGTT\n+\nIIIIIIIIIII\n", ["AACG", "AACGT"], "TGCAG", "PstI-MspI-Hall"))     # overlapping barcode+site
    cases.append(run_splitter(b"@a\nAACGTGCAGTT\n+\nIIIIIIIIIII\n", ["AACG", "AXG"], "TGCAG", "PstI-MspI-Hall"))       # bad barcode
    cases.append(run_splitter(b"@a\nAACGTGCAGTT\n+\nIIIIIIIIIII\n", ["AACG"], "TGCWG", "PstI-MspI-Hall"))              # bad cut site
    # ---- the reference's barcode_splitter_script.py end to end (key file with Input/Barcode/Output columns)
    import subprocess
    fq = random_fastq(rng, bcs, "TGCAG", "PstI-MspI-Hall", 150, "lf")
    key = "Input File,Barcode,Output File\n" + "".join("lane1.fq,%s,s%d.fq\n" % (b, i) for i, b in enumerate(bcs))
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "lane1.fq"), "wb").write(fq)
        open(os.path.join(d, "key.csv"), "w").write(key)
        argv = ["-b", "key.csv", "-a", "PstI-MspI-Hall"]
        p = subprocess.run([sys.executable, os.path.join(REF, "barcode_splitter_script.py")] + argv, cwd=d, capture_output=True,
                           text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert p.returncode == 0, p.stderr
        cli = {"argv": argv, "key_csv": key, "fastq_b64": base64.b64encode(fq).decode(), "stdout": p.stdout,
               "outputs_b64": {"s%d.fq" % i: base64.b64encode(open(os.path.join(d, "s%d.fq" % i), "rb").read()).decode()
                               for i in range(len(bcs))}}
    with open(os.path.join(HERE, "splitter.json"), "w") as fh:
        json.dump({"adapters": ADAPTERS, "find": find_cases(), "split": cases, "cli": cli}, fh, separators=(",", ":"))
        fh.write("\n")
    print("wrote splitter.json:", len(cases), "split cases,", os.path.getsize(os.path.join(HERE, "splitter.json")), "bytes")


if __name__ == "__main__":
    main()
