#!/usr/bin/env python3
"""The ordinary-gzip tier by itself (what bench.py's tiers.T3_gzip_file_gpu_inflate times): a one-member gzip file of the
canonical synthetic stream through td_count_file, decoded on the device (csrc/gz_gpu.hpp) -- for `rocprofv3 --kernel-trace
--stats -- python3 tools/gz_tier.py` (the kernels' durations) and with TAGDIG_INFLATE_STATS=1 (the stages' wall times).

  tools/gz_tier.py [reads] [calls] [level]        (default 16 000 000 reads, 3 calls after a warm one, zlib level 1)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig
from compress_formats import gzip_one_member

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = SynthConfig.from_id(3, nreads=reads)
eng = tagdigger_amd.Engine(0)
nb = reads * cfg.record_bytes
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
host = np.frombuffer(eng.d2h(d, nb), dtype=np.uint8)
eng.dev_free(d)
cells = len(cfg.barcodes) * len(cfg.tags)
dw = eng.dev_alloc(cells * 4)
eng.h2d(dw, bytes(cells * 4))
cfg.expected_device(eng, dw, 0, reads)
want = np.frombuffer(eng.d2h(dw, cells * 4), dtype=np.uint32).reshape(len(cfg.barcodes), len(cfg.tags))
eng.dev_free(dw)
eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
for kv in os.environ.get('TD_OPTS', '').split():      # e.g. TD_OPTS='gz_gpu_seg_kb=170000 gz_gpu_terr_kb=32'
    k, v = kv.split('='); eng.set_option(k, int(v))
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR")) as tmp:
    path = os.path.join(tmp, "tier.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip_one_member(host, level=level, threads=16))
    print("%d reads, %.1f MB of FASTQ in %.1f MB of gzip (one member, level %d)" % (reads, nb / 1e6, os.path.getsize(path) / 1e6, level), flush=True)
    eng.count_file(path)                                   # (warm: buffers)
    for k in range(calls):
        eng.reset()
        t0 = time.perf_counter()
        eng.count_file(path)
        eng.sync()
        dt = time.perf_counter() - t0
        ok = bool((eng.counts_numpy() == want).all())
        print("call %d: %.1f ms = %.1f M reads/s = %.2f GB/s of FASTQ; bit-exact %s; decoded on the device: %s"
              % (k, dt * 1e3, reads / dt / 1e6, nb / dt / 1e9, ok, eng.last_gz_route() == 1), flush=True)
eng.close()
