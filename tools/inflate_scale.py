#!/usr/bin/env python3
"""Host inflate against the number of threads: one ordinary gzip stream (chunk-parallel decoder) and one BGZF file
(member-parallel), td_gunzip_file alone (into host memory) and td_count_file end to end (inflate -> pinned ->
H2D -> count), TAGDIG_INFLATE_THREADS = 8 .. 64.   usage: inflate_scale.py [reads] [gzip level]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import tagdigger_amd
from tagdigger_amd import _binding as B
from tagdigger_amd.synth import SynthConfig
from compress_formats import bgzf_bytes

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb); cfg.fill_device(eng, d, 0, reads); host = eng.d2h(d, nb); eng.dev_free(d)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
eng.count_bytes(host); want = eng.counts_numpy().copy()
tmp = os.environ.get("TMPDIR", "/tmp")
plain = os.path.join(tmp, "infl.fq"); open(plain, "wb").write(host)
t0 = time.perf_counter(); subprocess.check_call("gzip -%d -c %s > %s.gz" % (level, plain, plain), shell=True); tg = time.perf_counter() - t0
t0 = time.perf_counter(); open(plain + ".bgzf.gz", "wb").write(bgzf_bytes(host, level=level, threads=32)); tb = time.perf_counter() - t0
print("%d reads, %.2f GB; gzip -%d %.2f GB (%.0f s), BGZF %.2f GB (%.0f s); cores %d" % (reads, nb / 1e9, level, os.path.getsize(plain + ".gz") / 1e9, tg,
      os.path.getsize(plain + ".bgzf.gz") / 1e9, tb, len(os.sched_getaffinity(0))), flush=True)
L = B.load()
buf = np.zeros(nb + 16, dtype=np.uint8)
ref = np.frombuffer(host, dtype=np.uint8)
for name, path in (("gzip", plain + ".gz"), ("BGZF", plain + ".bgzf.gz")):
    for th in (8, 16, 24, 32, 48, 64):
        os.environ["TAGDIG_INFLATE_THREADS"] = str(th)
        best = 1e9
        for _ in range(2):
            n = C.c_uint64(0)
            t0 = time.perf_counter()
            rc = L.td_gunzip_file(path.encode(), buf.ctypes.data_as(C.c_void_p), nb + 16, 32 << 20, C.byref(n))
            best = min(best, time.perf_counter() - t0)
            assert rc == 0 and n.value == nb, (rc, n.value)
        assert (buf[:nb] == ref).all()
        bestc = 1e9
        for _ in range(2):
            eng.reset(); t0 = time.perf_counter(); eng.count_file(path); eng.sync(); bestc = min(bestc, time.perf_counter() - t0)
        assert (eng.counts_numpy() == want).all()
        print("%-5s %2d threads: inflate alone %6.2f GB/s   count_file %6.2f GB/s = %6.1f Mreads/s" % (name, th, nb / best / 1e9, nb / bestc / 1e9, reads / bestc / 1e6), flush=True)
for f in (plain, plain + ".gz", plain + ".bgzf.gz"):
    os.remove(f)
