#!/usr/bin/env python3
"""Single-stream gzip on the GPU box: the sequential decoder (fast_inflate.hpp) against the chunk-parallel
one (par_inflate.hpp) over thread counts and chunk sizes -- td_gunzip_file alone (inflate into host memory)
and td_count_file end to end (inflate -> pinned staging -> copy -> count)."""
import ctypes as C, os, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tagdigger_amd
from tagdigger_amd import _binding as B
from tagdigger_amd.synth import SynthConfig

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
host = eng.d2h(d, nb)
eng.dev_free(d)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
eng.count_bytes(host)
want = eng.counts_numpy().copy()
tmp = os.environ.get("TMPDIR", "/tmp")
plain = os.path.join(tmp, "gzsweep.fq")
open(plain, "wb").write(host)
t0 = time.perf_counter()
subprocess.check_call("gzip -%d -c %s > %s.gz" % (level, plain, plain), shell=True)
gz = plain + ".gz"
print("%d reads, %.2f GB -> %.2f GB gzip -%d (%.0f s); host threads available: %d" % (
    reads, nb / 1e9, os.path.getsize(gz) / 1e9, level, time.perf_counter() - t0, len(os.sched_getaffinity(0))), flush=True)
L = B.load()
buf = np.zeros(nb + 16, dtype=np.uint8)
ref = np.frombuffer(host, dtype=np.uint8)

def run(label, env):
    for k in ("TAGDIG_PAR_INFLATE", "TAGDIG_INFLATE_THREADS", "TAGDIG_INFLATE_CHUNK", "TAGDIG_ZLIB", "TAGDIG_INFLATE_OVERSUB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    best = 1e9
    for _ in range(2):
        n = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = L.td_gunzip_file(gz.encode(), buf.ctypes.data_as(C.c_void_p), nb + 16, 32 << 20, C.byref(n))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0 and n.value == nb, (rc, n.value)
    assert (buf[:nb] == ref).all()
    eng.reset(); t0 = time.perf_counter(); eng.count_file(gz); eng.sync(); dt = time.perf_counter() - t0
    assert (eng.counts_numpy() == want).all()
    print("%-34s inflate %6.2f GB/s   count_file %6.2f GB/s  %6.2f Mreads/s" % (label, nb / best / 1e9, nb / dt / 1e9, reads / dt / 1e6), flush=True)

run("sequential (fast_inflate)", {"TAGDIG_PAR_INFLATE": "0"})
run("default", {})
os.environ["TAGDIG_INFLATE_STATS"] = "1"
for th in (8, 16):
    run("parallel %2d threads" % th, {"TAGDIG_INFLATE_THREADS": str(th)})
run("default again", {})
os.remove(plain); os.remove(gz)
