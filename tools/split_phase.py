#!/usr/bin/env python3
"""Diagnostic: where does a tile of k_split2 spend its time?  Phase-stamp build (libtagdig_prof.so), thread 0 of
every workgroup; shares only."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TAGDIG_LIB", os.path.join(ROOT, "tagdigger_amd", "libtagdig_prof.so"))
sys.path.insert(0, ROOT)
import tagdigger_amd
from tagdigger_amd import tagdigger_fun as tf
from tagdigger_amd.synth import SynthConfig
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3, adapter_pct=20)
ad = tf.adapters["PstI-MspI-Hall"]
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
with contextlib.redirect_stdout(io.StringIO()):
    ends = tf._adapter_ends(ad, cfg.barcodes)
eng.set_splitter(cfg.barcodes, cfg.cutsite, "CCGG", "CTGCAG", ends)
eng.split_device(d, nb)
eng.reset()
eng.split_device(d, nb)
c = list(eng.debug_counters()[:20]); c[11] = 0
names = ["loop head", "A: loads + wait, raw + masks -> LDS", "B: scan, list", "barrier 1", "C + D remainder (thread 0)", "end barrier",
         "  D: line selection", "  D: terminator + strip", "  D: pack + barcode walk", "  D: entries + site search", "  D: adapter search"]
tot = float(sum(c)) or 1.0
ntiles = (nb + 24 * 1024 - 1) // (24 * 1024)
for n, v in zip(names, c):
    print("  %-40s %6.2f %%   %8.0f cycles/tile" % (n, 100.0 * v / tot, v / ntiles))
print("  %-40s            %8.0f cycles/tile" % ("total", tot / ntiles))
eng.dev_free(d); eng.close()
