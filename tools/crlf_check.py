#!/usr/bin/env python3
"""Throughput on files that leave the short forms of phase A: CRLF line ends (every chunk takes the
general terminator / packing forms) and lower-case bases.  Result compared with the LF original."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
for kv in os.environ.get('TD_OPTS', '').split():      # e.g. TD_OPTS='kernel=1 table_load_pct=25'
    k, v = kv.split('='); eng.set_option(k, int(v))
nb = cfg.nbytes()
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
eng.count_device(d, nb); eng.sync()
want = eng.counts_numpy().copy()
host = eng.d2h(d, nb)
eng.dev_free(d)
for name, data in (("LF", host), ("CRLF", host.replace(b"\n", b"\r\n")), ("lower case", host.lower())):
    dd = eng.dev_alloc(len(data)); eng.h2d(dd, data)
    eng.reset(); eng.count_device(dd, len(data)); eng.sync()
    ok = bool((eng.counts_numpy() == want).all())
    eng.reset(); eng.set_option("timing", 1)
    for _ in range(3):
        eng.count_device(dd, len(data))
    eng.sync()
    ms, _ = eng.kernel_time_ms(); eng.set_option("timing", 0)
    print("%-10s %6.2f Gread/s  %5.2f TB/s  same matrix: %s  fix-ups %d" % (name, reads / ms / 1e6, len(data) / ms / 1e9, ok, eng.debug_counters()[11]))
    eng.dev_free(dd)
