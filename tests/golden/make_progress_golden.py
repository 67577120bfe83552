#!/usr/bin/env python3
"""Generate tests/golden/progress.json from the REAL reference (build container only): what
find_tags_fastq PRINTS while it reads (tagdigger_fun.py:268-271: the file name after every
1 000 000 reads, its three counters after every 50 000) together with the matrix it returns,
on inputs long enough to print something.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_progress_golden.py [/root/reference]

The inputs are not stored: they come from seeded generators the tests share -- the synthetic
stream of include/td_synth_spec.h (oracle/synth_ref.c on the CPU) and tests/helpers.dirty_fastq --
and the fixture holds their SHA-256, so a test that regenerates them knows it reads the same
bytes the reference read.  Only DATA is written (parameters, observed stdout, observed counts).
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import random
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
import tagdigger_fun as ref  # noqa: E402  (the reference's, from its checkout)
import helpers  # noqa: E402


def cases():
    """name -> (file name, bytes, barcodes, tags, kwargs, how the test regenerates the bytes)"""
    from tagdigger_amd.synth import SynthConfig
    shape = dict(nbar=8, nmarkers=50, seed=1234, cutsite="TGCAG", bclen=(4, 8))
    a = SynthConfig(nreads=1_050_000, **shape)
    yield ("synthetic, 1.05 M reads, plain", "lib_a.fq", helpers.synth_host_bytes(a, 0, a.nreads).tobytes(), a.barcodes, a.tags,
           dict(cutsite="TGCAG"), dict(kind="synth", config=dict(shape, nreads=a.nreads)))
    b = SynthConfig(nreads=160_000, **dict(shape, seed=77))
    yield ("synthetic, 160 k reads, gzip, maxreads 120 000", "lib_b.fq.gz", helpers.synth_host_bytes(b, 0, b.nreads).tobytes(),
           b.barcodes, b.tags, dict(cutsite="TGCAG", maxreads=120000), dict(kind="synth", config=dict(shape, seed=77, nreads=b.nreads)))
    rnd = random.Random(20260)
    barcodes, tags, cutsites = helpers.small_index(rnd, "TGCAG", nbar=12, ntag=60)
    raw = helpers.dirty_fastq(rnd, barcodes, tags, cutsites, 130_000, nl_choices=("\n", "\r\n"), long_lines=True)
    yield ("irregular lines, 130 k records, LF and CRLF", "lib_c.fq", raw, barcodes, tags, dict(cutsite="TGCAG"),
           dict(kind="dirty", seed=20260, nbar=12, ntag=60, nrec=130_000, nl_choices=["\n", "\r\n"], long_lines=True))
    rnd = random.Random(20261)
    barcodes, tags, cutsites = helpers.small_index(rnd, "CWGC", nbar=10, ntag=40)
    raw = helpers.dirty_fastq(rnd, barcodes, tags, cutsites, 110_000, nl_choices=("\n",), permanent_shifts=True)
    yield ("irregular lines, lost line phase, CWGC, 110 k records", "lib_d.fq", raw, barcodes, tags, dict(cutsite="CWGC"),
           dict(kind="dirty", seed=20261, cutsite="CWGC", nbar=10, ntag=40, nrec=110_000, nl_choices=["\n"], permanent_shifts=True))


def splitter_case():
    """barcodeSplitter (:1286-1368) prints its own progress lines (:1357-1360): 130 k reads of the synthetic stream with
    adapter read-through on a fifth of them; the output files are recorded by their SHA-256."""
    from tagdigger_amd.synth import SynthConfig
    shape = dict(nbar=8, nmarkers=50, seed=909, cutsite="TGCAG", bclen=(4, 8), adapter_pct=20)
    cfg = SynthConfig(nreads=130_000, **shape)
    raw = helpers.synth_host_bytes(cfg, 0, cfg.nreads).tobytes()
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "lib_s.fq"), "wb") as fh:
            fh.write(raw)
        outs = ["split_%d.fq" % k for k in range(len(cfg.barcodes))]
        old = os.getcwd()
        os.chdir(d)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ref.barcodeSplitter("lib_s.fq", list(cfg.barcodes), outs, cutsite="TGCAG", adapter=ref.adapters["PstI-MspI-Hall"],
                                    maxreads=120000)
            sums = [hashlib.sha256(open(o, "rb").read()).hexdigest() for o in outs]
        finally:
            os.chdir(old)
    lines = buf.getvalue().splitlines()
    print("%-55s %8d bytes  %4d lines printed, last: %s" % ("splitter", len(raw), len(lines), lines[-1]))
    return dict(name="splitter, 130 k reads, adapter read-through, maxreads 120 000", file="lib_s.fq", sha256=hashlib.sha256(raw).hexdigest(),
                recipe=dict(kind="synth", config=dict(shape, nreads=cfg.nreads)), splitter=dict(adapter="PstI-MspI-Hall", maxreads=120000),
                kwargs=dict(cutsite="TGCAG"), barcodes=list(cfg.barcodes), tags=list(cfg.tags), stdout=lines, outputs_sha256=sums)


def main():
    out = []
    for name, fname, raw, barcodes, tags, kw, recipe in cases():
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, fname), "wb") as fh:
                fh.write(gzip.compress(raw, compresslevel=1) if fname.endswith("gz") else raw)
            old = os.getcwd()
            os.chdir(d)                                   # (the reference prints the name it was given)
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    counts = ref.find_tags_fastq(fname, list(barcodes), list(tags), **kw)
            finally:
                os.chdir(old)
        lines = buf.getvalue().splitlines()
        print("%-55s %8d bytes  %4d lines printed, last: %s" % (name, len(raw), len(lines), lines[-1] if lines else ""))
        out.append(dict(name=name, file=fname, sha256=hashlib.sha256(raw).hexdigest(), recipe=recipe, kwargs=kw,
                        barcodes=list(barcodes), tags=list(tags), stdout=lines, counts=counts))
    out.append(splitter_case())
    path = os.path.join(HERE, "progress.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
        fh.write("\n")
    print("wrote progress.json", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
