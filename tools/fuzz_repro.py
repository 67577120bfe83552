#!/usr/bin/env python3
"""One case of tests/test_gpu_parity.py::test_fuzz_campaign by its seed, through chosen option sets, several times each:
where the matrix differs from the C oracle's and by how much."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tagdigger_amd
from helpers import fuzz_case
from oracle import c_oracle

seed = int(sys.argv[1])
reps = int(os.environ.get("TD_REPS", "5"))
barcodes, tags, cutsite, nl, data = fuzz_case(seed)
ost = {}
want = c_oracle.COracle(barcodes, tags, cutsite).count_bytes(data, stats=ost)
print("seed %d cutsite %r nl %r: %d barcodes %d tags %d bytes, oracle %r" % (seed, cutsite, nl, len(barcodes), len(tags), len(data), ost))
eng = tagdigger_amd.Engine(0)
eng.set_index(barcodes, tags, cutsite)
for opts in sys.argv[2:] or ["kernel=4"]:
    for kv in opts.split(","):
        k, v = kv.split("="); eng.set_option(k, int(v))
    bad = 0
    for r in range(reps):
        eng.reset(); eng.count_bytes(data)
        got = eng.counts_numpy(); st = eng.stats()
        if not (got == want).all() or (st["reads"], st["barcut"], st["tag"]) != (ost["reads"], ost["barcut"], ost["tag"]):
            bad += 1
            if bad == 1:
                d = np.argwhere(got != want)
                print("  %s rep %d: %d cells differ; stats %r; first: %r" % (opts, r, len(d), (st["reads"], st["barcut"], st["tag"]),
                      [(tuple(x), int(got[tuple(x)]), int(want[tuple(x)])) for x in d[:6]]), "debug", eng.debug_counters()[:16])
    print("%-40s %d of %d runs differ" % (opts, bad, reps))
eng.close()
