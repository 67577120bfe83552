#!/usr/bin/env python3
"""Per-chunk timing of the chunk-parallel gzip decoder on one file (TAGDIG_INFLATE_STATS=2)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tagdigger_amd
from tagdigger_amd import _binding as B
from tagdigger_amd.synth import SynthConfig
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb); cfg.fill_device(eng, d, 0, reads); host = eng.d2h(d, nb); eng.dev_free(d)
import gzip
gz = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gzdebug.fq.gz")
with gzip.open(gz, "wb", compresslevel=1) as fh:
    fh.write(host)
L = B.load()
buf = np.zeros(nb + 16, dtype=np.uint8)
os.environ["TAGDIG_INFLATE_STATS"] = "2"
for th in sys.argv[2:] or ["16"]:
    os.environ["TAGDIG_INFLATE_THREADS"] = th
    for rep in range(2):
        n = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = L.td_gunzip_file(gz.encode(), buf.ctypes.data_as(C.c_void_p), nb + 16, 32 << 20, C.byref(n))
        dt = time.perf_counter() - t0
        print("threads %s: %.2f GB/s" % (th, nb / dt / 1e9), file=sys.stderr, flush=True)
os.remove(gz)
