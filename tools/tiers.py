#!/usr/bin/env python3
"""Throughput tiers T2/T3 of SURVEY section 8d (never the bench `value`, which is T1 = HBM-resident):
T2 host buffer -> pinned staging -> hipMemcpyAsync overlapped with counting (td_count_host);
T3 end to end from a file, plain, gzip (par_inflate.hpp chunk-parallel, fast_inflate.hpp on one host thread, zlib)
and BGZF (member-parallel)."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
host = eng.d2h(d, nb)
eng.dev_free(d)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
d_want = eng.dev_alloc(len(cfg.barcodes) * len(cfg.tags) * 4)
eng.h2d(d_want, bytes(len(cfg.barcodes) * len(cfg.tags) * 4))
cfg.expected_device(eng, d_want, 0, reads)            # the matrix the generator's own choices imply
want = np.frombuffer(eng.d2h(d_want, len(cfg.barcodes) * len(cfg.tags) * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
eng.dev_free(d_want)

def check():
    assert (eng.counts_numpy() == want).all()

eng.count_bytes(host); check(); eng.reset()
t0 = time.perf_counter(); eng.count_bytes(host); eng.sync(); dt = time.perf_counter() - t0; check()
print("T2 host buffer   : %6.2f Mreads/s  %6.2f GB/s  (%.2f GB, includes one host->pinned memcpy)" % (reads / dt / 1e6, nb / dt / 1e9, nb / 1e9))
tmp = os.environ.get("TMPDIR", "/tmp")
plain = os.path.join(tmp, "tiers_lib.fq")
open(plain, "wb").write(host)
eng.reset(); t0 = time.perf_counter(); eng.count_file(plain); eng.sync(); dt = time.perf_counter() - t0; check()
print("T3 plain file    : %6.2f Mreads/s  %6.2f GB/s  (page cache warm)" % (reads / dt / 1e6, nb / dt / 1e9))
gzp = plain + ".gz"
t0 = time.perf_counter()
with gzip.open(gzp, "wb", compresslevel=1) as fh:
    fh.write(host)
tz = time.perf_counter() - t0
eng.reset(); t0 = time.perf_counter(); eng.count_file(gzp); eng.sync(); dt = time.perf_counter() - t0; check()
print("T3 gzip file     : %6.2f Mreads/s  %6.2f GB/s uncompressed (chunk-parallel DEFLATE decoder, %s threads, CRC-32 checked; %.2f GB gz, made in %.1f s)" % (
    reads / dt / 1e6, nb / dt / 1e9, os.environ.get("TAGDIG_INFLATE_THREADS", "default"), os.path.getsize(gzp) / 1e9, tz))
os.environ["TAGDIG_PAR_INFLATE"] = "0"
eng.reset(); t0 = time.perf_counter(); eng.count_file(gzp); eng.sync(); dt = time.perf_counter() - t0; check()
print("T3 gzip, 1 thread: %6.2f Mreads/s  %6.2f GB/s uncompressed (TAGDIG_PAR_INFLATE=0: the sequential decoder)" % (reads / dt / 1e6, nb / dt / 1e9))
os.environ["TAGDIG_ZLIB"] = "1"
eng.reset(); t0 = time.perf_counter(); eng.count_file(gzp); eng.sync(); dt = time.perf_counter() - t0; check()
print("T3 gzip, zlib    : %6.2f Mreads/s  %6.2f GB/s uncompressed (TAGDIG_ZLIB=1)" % (reads / dt / 1e6, nb / dt / 1e9))
del os.environ["TAGDIG_PAR_INFLATE"], os.environ["TAGDIG_ZLIB"]
def bgzf_bytes(data, block=0xFF00, level=6):
    import struct, zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                   + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)
bz = plain + ".bgzf.gz"
part = bytes(host[: (len(host) // 4 // cfg.record_bytes) * cfg.record_bytes])      # (python-side compression is slow: a quarter)
t0 = time.perf_counter(); open(bz, "wb").write(bgzf_bytes(part, level=1)); tz = time.perf_counter() - t0
eng.reset(); t0 = time.perf_counter(); eng.count_file(bz); eng.sync(); dt = time.perf_counter() - t0
print("T3 BGZF file     : %6.2f Mreads/s  %6.2f GB/s uncompressed (member-parallel inflate, %s threads; %.2f GB gz, made in %.1f s)" % (
    len(part) / cfg.record_bytes / dt / 1e6, len(part) / dt / 1e9, os.environ.get("TAGDIG_INFLATE_THREADS", "default"),
    os.path.getsize(bz) / 1e9, tz))
os.remove(plain); os.remove(gzp); os.remove(bz)
