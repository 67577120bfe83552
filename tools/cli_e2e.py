#!/usr/bin/env python3
"""End-to-end cost of the drop-in command line at a BASELINE config's scale, stage by stage
(VERDICT r01 item 6): key file + Merged tag table + one FASTQ library of N synthetic reads on disk ->
`python -m tagdigger_amd.tagdigger_script ... --td-timing` -> counts.csv, geno.csv.
The stages are the reference script's (tagdigger_script.py:80-135): tag reader, sanitizeTags, key file,
find_tags_fastq per library (index build, the counting itself, matrix to the host), combineReadCounts,
writeCounts, writeDiploidGeno.

  usage: tools/cli_e2e.py [--config 2] [--reads 50000000] [--gz] [--keep]
"""
import argparse
import csv
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--gz", action="store_true", help="write the library gzip-compressed (level 1)")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import tagdigger_amd
    from tagdigger_amd import tagdigger_script
    from tagdigger_amd.synth import CONFIGS, SynthConfig, merged_rows
    base = dict(CONFIGS[a.config])
    base.pop("triallelic_pct", None)
    base.pop("adapter_pct", None)
    if a.reads:
        base["nreads"] = a.reads
    cfg = SynthConfig(**base)
    work = tempfile.mkdtemp(prefix="td_cli_", dir=os.environ.get("TMPDIR"))
    t0 = time.perf_counter()
    # inputs: the library (generated in HBM, copied out, written), the key file, the Merged tag table
    eng = tagdigger_amd.Engine(0)
    lib = "lib.fq.gz" if a.gz else "lib.fq"
    with open(os.path.join(work, lib), "wb") as fh:
        step = 8_000_000
        for first in range(0, cfg.nreads, step):
            n = min(step, cfg.nreads - first)
            d = eng.dev_alloc(n * cfg.record_bytes)
            cfg.fill_device(eng, d, first, n)
            piece = eng.d2h(d, n * cfg.record_bytes)
            eng.dev_free(d)
            if a.gz:
                import gzip
                piece = gzip.compress(piece, compresslevel=1)      # (a multi-member file: the reference reads it through)
            fh.write(piece)
    eng.close()
    with open(os.path.join(work, "key.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["File", "Barcode", "Sample"])
        for k, b in enumerate(cfg.barcodes):
            w.writerow([lib, b, "S%d" % (k % max(1, len(cfg.barcodes) * 3 // 4))])      # a quarter of the samples carry two barcodes
    rng = np.random.default_rng(1)
    with open(os.path.join(work, "tags.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Marker name", "Tag sequence"])
        w.writerows(merged_rows(cfg.tags, len(cfg.tags) // 2, rng, 0))
    size = os.path.getsize(os.path.join(work, lib))
    print("inputs ready in %.1f s: %d reads, %s = %.2f GB, %d barcodes, %d tags" % (
        time.perf_counter() - t0, cfg.nreads, lib, size / 1e9, len(cfg.barcodes), len(cfg.tags)), file=sys.stderr)
    old = os.getcwd()
    t0 = time.perf_counter()
    try:
        tagdigger_script.main(["-c", cfg.cutsite, "--MergedTags", "tags.csv", "-b", "key.csv", "-o", "counts.csv", "-g", "geno.csv",
                               "-w", work, "--td-timing"])
    finally:
        os.chdir(old)
    wall = time.perf_counter() - t0
    print("command wall time %.2f s  (%.1f M reads/s, %.2f GB/s of FASTQ on disk); counts.csv %.1f MB, geno.csv %.1f MB" % (
        wall, cfg.nreads / wall / 1e6, size / wall / 1e9, os.path.getsize(os.path.join(work, "counts.csv")) / 1e6,
        os.path.getsize(os.path.join(work, "geno.csv")) / 1e6), file=sys.stderr)
    # the matrix the command wrote against the generator's own expectation (the product's generator: no checker code
    # in this tool), folded into samples the way the key file says
    try:
        eng = tagdigger_amd.Engine(0)
        cells = len(cfg.barcodes) * len(cfg.tags)
        dw = eng.dev_alloc(cells * 4)
        eng.h2d(dw, bytes(cells * 4))
        cfg.expected_device(eng, dw, 0, cfg.nreads)
        exp = np.frombuffer(eng.d2h(dw, cells * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
        eng.dev_free(dw)
        eng.close()
        rows = [k % max(1, len(cfg.barcodes) * 3 // 4) for k in range(len(cfg.barcodes))]
        tot = np.zeros((max(rows) + 1, exp.shape[1]), dtype=np.int64)
        np.add.at(tot, rows, exp.astype(np.int64))
        got = np.loadtxt(os.path.join(work, "counts.csv"), delimiter=",", skiprows=1, usecols=range(1, exp.shape[1] + 1), dtype=np.int64)
        print("counts.csv equals the generator's expected matrix: %s (%d hits)" % (bool((got == tot).all()), int(exp.sum())), file=sys.stderr)
    except Exception as e:
        print("check skipped: %s" % e, file=sys.stderr)
    if not a.keep:
        import shutil
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
