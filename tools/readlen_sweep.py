#!/usr/bin/env python3
"""Kernel throughput against read length (canonical synthetic stream, 96 barcodes, 10 k tags):
short reads put more than 256 wanted lines into a 32 KB tile and take the general path of phase D."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig

eng = tagdigger_amd.Engine(0)
for kv in os.environ.get('TD_OPTS', '').split():      # e.g. TD_OPTS='kernel=1 table_load_pct=25'
    k, v = kv.split('='); eng.set_option(k, int(v))
NBAR, NMARK = int(os.environ.get('TD_SWEEP_NBAR', '96')), int(os.environ.get('TD_SWEEP_NMARK', '5000'))     # (index shape: config 2's by default)
LENS = [int(x) for x in os.environ.get('TD_SWEEP_LENS', '36 50 75 100 150 250').split()]
print("index: %d barcodes x %d tags; options: %s" % (NBAR, 2 * NMARK, os.environ.get('TD_OPTS', '(default)')))
for read_len, body in ((36, 20), (50, 30), (75, 45), (100, 59), (150, 59), (250, 59)):
    if read_len not in LENS:
        continue
    nreads = int(4e9 // (2 * read_len + 19))
    cfg = SynthConfig(nreads=nreads, nbar=NBAR, nmarkers=NMARK, seed=2, read_len=read_len, body=body)
    nb = cfg.nbytes()
    d = eng.dev_alloc(nb)
    cfg.fill_device(eng, d, 0, nreads)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.count_device(d, nb); eng.sync()
    fast = eng.counts_numpy().copy()
    # the exact in-flight kernel on the same bytes (the generator's expected matrix ignores the
    # one-in-a-billion random read that happens to match a short tag)
    eng.reset(); eng.set_option("fastpath", 0); eng.count_device(d, nb); eng.sync()
    ok = bool((eng.counts_numpy() == fast).all())
    eng.set_option("fastpath", 1)
    eng.reset(); eng.set_option("timing", 1)
    for _ in range(3):
        eng.count_device(d, nb)
    eng.sync()
    ms, n = eng.kernel_time_ms()
    eng.set_option("timing", 0)
    print("read length %3d (%3d B/record): %6.2f Gread/s  %5.2f TB/s  %5.1f %% of 8 TB/s  equals the exact kernel: %s  fix-ups %d"
          % (read_len, cfg.record_bytes, nreads / ms / 1e6, nb / ms / 1e9, nb / ms / 1e9 / 8 * 100, ok, eng.debug_counters()[11]))
    eng.dev_free(d)
eng.close()
