#!/usr/bin/env python3
"""BASELINE configs[2] as written, once: ONE gzip file (one member, one DEFLATE stream) of the canonical synthetic
stream -- 200 M reads x 384 barcodes x 100 k tags, 43.8 GB of FASTQ -- counted end to end by
find_tags_fastq's file path (gzip.open of the reference, tagdigger_fun.py:240-243 -> td_count_file: decoded on the device in
segments of 1 GiB, csrc/gz_gpu.hpp; then once more with round 3's path -- DEFLATE decoded into symbols by the host's threads as a
pipeline, markers -> bytes and the CRC-32 on the GPU), the whole matrix
checked against the generator's expectation.  TAGDIG_INFLATE_STATS=1 prints where the time went.

  tools/config3_gzip_e2e.py [reads] [dir]      (default 200 000 000, $TMPDIR or /tmp)

The file is written slice by slice (the stream is generated on the GPU 8 M reads at a time, compressed on a thread
pool the way pigz does -- level 1, pieces of 1 MiB closed by full flushes -- and appended: one header, one trailer
with the CRC-32 of everything); neither the 43.8 GB nor the ~8 GB ever sit in host memory."""
import os
import struct
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig
from compress_formats import _crc32_combine

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
where = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("TMPDIR", "/tmp")
threads = int(os.environ.get("TD_E2E_COMPRESS_THREADS", "16"))
cfg = SynthConfig.from_id(3, nreads=reads)
eng = tagdigger_amd.Engine(0)
path = os.path.join(where, "config3_lib.fq.gz")
SLICE, PIECE = 8_000_000, 1 << 20
rb = cfg.record_bytes
t0 = time.perf_counter()
crc, total = zlib.crc32(b""), 0
d = eng.dev_alloc(SLICE * rb)
with open(path, "wb") as fh, ThreadPoolExecutor(threads) as ex:
    fh.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff")
    for first in range(0, reads, SLICE):
        n = min(SLICE, reads - first)
        cfg.fill_device(eng, d, first, n)
        view = memoryview(eng.d2h(d, n * rb))
        starts = list(range(0, len(view), PIECE))
        last_slice = first + n >= reads

        def part(i):
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = co.compress(view[i:i + PIECE])
            fin = last_slice and i == starts[-1]
            return body + co.flush(zlib.Z_FINISH if fin else zlib.Z_FULL_FLUSH), zlib.crc32(view[i:i + PIECE])
        for (body, c), i in zip(ex.map(part, starts), starts):
            fh.write(body)
            crc = _crc32_combine(crc, c, min(PIECE, len(view) - i))
        total += len(view)
        print("  written %d M reads (%.1f s)" % ((first + n) // 1_000_000, time.perf_counter() - t0), flush=True)
    fh.write(struct.pack("<II", crc & 0xFFFFFFFF, total & 0xFFFFFFFF))
eng.dev_free(d)
gz_bytes = os.path.getsize(path)
print("file: %s, %.2f GB of FASTQ in %.2f GB (one member), written in %.1f s" % (path, total / 1e9, gz_bytes / 1e9, time.perf_counter() - t0), flush=True)

# the generator's expected matrix (device-built; nothing is parsed)
cells = len(cfg.barcodes) * len(cfg.tags)
dw = eng.dev_alloc(cells * 4)
eng.h2d(dw, bytes(cells * 4))
hits = cfg.expected_device(eng, dw, 0, reads)
want = np.frombuffer(eng.d2h(dw, cells * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
eng.dev_free(dw)

eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
for attempt in ("first call (page cache warm from writing; pinned pieces allocated)", "second call", "third call, Huffman decoding on the host's threads (gpu_huffman 0: round 3's path)"):
    eng.set_option("gpu_huffman", 0 if attempt.startswith("third") else 1)
    eng.reset()
    t1 = time.perf_counter()
    eng.count_file(path)
    eng.sync()
    dt = time.perf_counter() - t1
    ok = bool((eng.counts_numpy() == want).all())
    st = eng.stats()
    print("%s: %.2f s = %.1f M reads/s = %.2f GB/s of FASTQ (%.2f GB/s compressed); reads %d, tag hits %d (expected %d); bit-exact %s; "
          "decoded on the device: %s; inflate threads %s" % (attempt, dt, reads / dt / 1e6, total / dt / 1e9, gz_bytes / dt / 1e9, st["reads"], st["tag"], hits, ok,
                                  bool(eng.last_gz_route() == 1), os.environ.get("TAGDIG_INFLATE_THREADS", "default (host cores, at most 16)")), flush=True)
os.remove(path)
eng.close()
