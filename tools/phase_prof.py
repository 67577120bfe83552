#!/usr/bin/env python3
"""Diagnostic: where does a tile's time go?  Runs the phase-stamp build
(libtagdig_prof.so, `make -C tagdigger_amd/csrc prof`) over a synthetic library
and prints each phase's share of workgroup time.  Shares only -- the stamped
build's run time is never quoted as a result."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TAGDIG_LIB", os.path.join(ROOT, "tagdigger_amd", "libtagdig_prof.so"))
sys.path.insert(0, ROOT)

NAMES_EXACT = ["ticket", "load+masks+LDS store", "barrier after load", "block scan", "look-back",
         "emission", "phase 2 (match)", "loop tail",
         "  p2: fetch+convert", "  p2: barcode", "  p2: tag probes", "-"]


NAMES = ["loop head", "A: wait bytes + masks + pack", "barrier A", "B: scan + local votes", "C: vote",
         "  pending: commit (atomic issued)", "D: commit + loop (rest)", "end barrier",
         "  D: packed chunks + align", "  D: barcode walk", "  D: tag words", "(fix-up queue length)",
         "  D: wanted-line selection", "  D: commit issued", "  D: hash + bucket loads issued", "  D: hot part exit",
         "  pending: wait for bucket", "  pending: compares", "-", "-"]


NAMES2 = ["loop head", "A: wait bytes, raw + masks -> LDS (own quarter)", "B: masks, scan, vote, list", "  pending: wait bucket + compares",
          "  next loads issued, pending commit", "barrier 1", "C: phase", "D: remainder (thread 0)", "end barrier",
          "  D: list + pieces from LDS", "  D: pack + nvalid", "  D: barcode walk", "  D: tag words, hash, bucket loads", "-", "-", "-", "-", "-", "-", "-"]


NAMES4 = ["producer: waiting for the slot", "producer: A (wait bytes, raw + masks -> LDS, next loads)", "producer: B (scan, vote, lists)",
          "producer: closing a tile", "producer: loop", "consumer: pending lines", "consumer: waiting for lines", "consumer: matching",
          "consumer: loop", "-", "-", "-", "-", "-", "-", "-", "-", "-", "-", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--barcodes", type=int, default=384)
    ap.add_argument("--markers", type=int, default=50_000)
    ap.add_argument("--tile-kb", type=int, default=16)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--prescan", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=2)
    ap.add_argument("--tile-kb2", type=int, default=24)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import tagdigger_amd
    from tagdigger_amd.synth import SynthConfig
    cfg = SynthConfig(nreads=a.reads, nbar=a.barcodes, nmarkers=a.markers, seed=3)
    eng = tagdigger_amd.Engine(0)
    d = eng.dev_alloc(cfg.nbytes())
    cfg.fill_device(eng, d, 0, cfg.nreads)
    eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
    eng.set_option("tile_kb", a.tile_kb)
    eng.set_option("prescan", a.prescan)
    eng.set_option("kernel", a.kernel)
    if a.kernel == 2:
        eng.set_option("tile_kb2", a.tile_kb2)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v, 0))
    if a.blocks_per_cu:
        eng.set_option("blocks_per_cu", a.blocks_per_cu)
    eng.count_device(d, cfg.nbytes())       # warm
    eng.reset()
    eng.set_option("timing", 1)
    eng.count_device(d, cfg.nbytes())
    ms, _ = eng.kernel_time_ms()
    c = list(eng.debug_counters()[:20]); c[11] = 0
    tot = float(sum(c[:20])) or 1.0
    tkb = 24 if a.kernel == 4 else a.tile_kb2 if a.kernel == 2 else a.tile_kb
    ntiles = (cfg.nbytes() + tkb * 1024 - 1) // (tkb * 1024)
    print("kernel=%d tile_kb=%d blocks_per_cu=%s prescan=%d  kernel %.2f ms (stamped build)  tiles=%d" % (
        a.kernel, tkb, a.blocks_per_cu or "auto", a.prescan, ms, ntiles))
    for n, v in zip(NAMES4 if a.kernel == 4 else NAMES2 if a.kernel == 2 else NAMES, c):
        print("  %-24s %6.2f %%   %8.0f cycles/tile" % (n, 100.0 * v / tot, v / ntiles))
    print("  %-24s            %8.0f cycles/tile" % ("total", tot / ntiles))
    eng.dev_free(d)
    eng.close()


if __name__ == "__main__":
    main()
