#!/usr/bin/env python3
"""Throughput of the barcode splitter on the canonical synthetic stream:
decisions only (k_count_lines + k_scan_tiles + k_split on a buffer resident in HBM) and end to end
from a file (td_split_file: read, H2D, decide, D2H, host writes the clipped records).  (The CPU
restatement of the same loop is timed by tests/test_splitter.py::test_splitter_cpu_restatement_rate.)"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tagdigger_amd
from tagdigger_amd import tagdigger_fun as tf
from tagdigger_amd.synth import SynthConfig

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
kern = int(sys.argv[2]) if len(sys.argv) > 2 else 2        # 2: k_split2 (tile in LDS), 1: k_split
files = len(sys.argv) <= 3 or sys.argv[3] != "nofile"
cfg = SynthConfig(nreads=reads, nbar=384, nmarkers=50_000, seed=3, adapter_pct=20)     # (config 5's read-through on a fifth of the reads)
ad = tf.adapters["PstI-MspI-Hall"]
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
d = eng.dev_alloc(nb)
cfg.fill_device(eng, d, 0, reads)
with contextlib.redirect_stdout(io.StringIO()):
    ends = tf._adapter_ends(ad, cfg.barcodes)
eng.set_splitter(cfg.barcodes, cfg.cutsite, "CCGG", "CTGCAG", ends)
eng.set_option("split_kernel", kern)
if os.environ.get("TD_ABLATE"):
    eng.set_option("debug_ablate", int(os.environ["TD_ABLATE"]))      # (timing only: 64 no site search, 128 no adapter search)
eng.split_device(d, nb)                                   # warm
t0 = time.perf_counter(); res, terms = eng.split_device(d, nb); dt = time.perf_counter() - t0
eng.set_option("timing", 0)
hit = int((res[:reads, 0] >= 0).sum()); clip = int(((res[:reads, 0] >= 0) & (res[:reads, 1] != 999)).sum())
print("decisions, HBM-resident : %7.1f Mreads/s  %6.1f GB/s  (%d reads, %d with barcode+site, %d clipped; includes the D2H of 8 B per read)"
      % (reads / dt / 1e6, nb / dt / 1e9, reads, hit, clip))
if not files:
    eng.dev_free(d); eng.close(); sys.exit(0)
tmp = os.environ.get("TMPDIR", "/tmp")
src = os.path.join(tmp, "split_in.fq")
n_file = min(reads, 8_000_000)
open(src, "wb").write(eng.d2h(d, n_file * cfg.record_bytes))
eng.dev_free(d)
outs = [os.path.join(tmp, "split_out_%03d.fq" % i) for i in range(len(cfg.barcodes))]
eng.split_file(src, outs)                                 # (first call: pinned staging buffers are allocated and kept)
for o in outs: os.remove(o)                               # (fresh output files, as in real use: truncating 1.3 GB of cached pages is not the splitter's work)
t0 = time.perf_counter(); st = eng.split_file(src, outs); dt = time.perf_counter() - t0
outbytes = sum(os.path.getsize(o) for o in outs)
print("file -> %d files        : %7.1f Mreads/s  %6.2f GB/s in, %.2f GB written  (reads %d, barcode+site %d, clipped %d)"
      % (len(outs), n_file / dt / 1e6, n_file * cfg.record_bytes / dt / 1e9, outbytes / 1e9, *st))
for o in outs: os.remove(o)
os.remove(src)
