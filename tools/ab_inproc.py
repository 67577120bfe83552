#!/usr/bin/env python3
"""A/B of two (or more) builds of libtagdig IN ONE PROCESS, passes alternating over the same resident buffer:
clock drift of the box (the same binary moves by +-4 % from run to run) hits every arm alike.

  tools/ab_inproc.py libtagdig_base.so libtagdig.so [--reads N] [--rounds R] [--config 3] [--opt k=v ...]

Prints per arm: mean / min kernel ms (HIP events around the count kernels) and whether the matrix is bit-exact
against the generator's expectation.  B.check reports errors through the first library's td_last_error (messages
only)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--progress", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--arm", action="append", default=[], help="k=v[,k=v...] options of arm i (repeat per library, in order)")
    a = ap.parse_args()
    import torch
    from tagdigger_amd import _binding as B
    from tagdigger_amd import engine as E
    from tagdigger_amd.synth import CONFIGS, SynthConfig
    # every build exports the same names: load them RTLD_LOCAL, or the second one's internal calls (and kernel stubs)
    # would bind to the first one's
    orig_cdll = C.CDLL
    C.CDLL = lambda path, mode=0, **kw: orig_cdll(path, mode=(C.RTLD_GLOBAL if "amdhip" in str(path) else C.RTLD_LOCAL), **kw)
    engines = []
    for lib in a.libs:
        B._lib = None
        B.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, "tagdigger_amd", lib)
        engines.append(E.Engine(0))                       # (keeps its own library handle)
    base = dict(CONFIGS[a.config])
    base["nreads"] = a.reads
    if a.read_len:
        base["read_len"] = a.read_len
    cfg = SynthConfig(**base)
    dev = torch.device("cuda", 0)
    nbytes = cfg.nreads * cfg.record_bytes
    fastq = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cfg.fill_device(engines[0], fastq.data_ptr(), 0, cfg.nreads)
    counts = torch.zeros(len(cfg.barcodes) * len(cfg.tags), dtype=torch.int32, device=dev)
    want = torch.zeros_like(counts)
    cfg.expected_device(engines[0], want.data_ptr(), 0, cfg.nreads)
    stream = torch.cuda.current_stream().cuda_stream
    exact = []
    for n, eng in enumerate(engines):
        eng.set_option("progress", a.progress)
        for kv in a.opt + (a.arm[n].split(",") if n < len(a.arm) and a.arm[n] else []):
            k, v = kv.split("=")
            eng.set_option(k, int(v, 0))
        eng.bind_counts(counts.data_ptr())
        eng.set_index(cfg.barcodes, cfg.tags, cfg.cutsite)
        counts.zero_()
        eng.count_device(fastq.data_ptr(), nbytes, stream=stream)
        torch.cuda.synchronize()
        exact.append(bool(torch.equal(counts, want)))
        eng.set_option("timing", 1)
        eng.kernel_times_ms()
    times = [[] for _ in engines]
    for r in range(a.rounds):
        for k, eng in enumerate(engines):
            counts.zero_()
            eng.reset()
            eng.count_device(fastq.data_ptr(), nbytes, stream=stream)
            torch.cuda.synchronize()
            times[k] += eng.kernel_times_ms()
    scale = 200_000_000 / cfg.nreads
    for n, (lib, t, ok) in enumerate(zip(a.libs, times, exact)):
        lib = lib + (" [" + a.arm[n] + "]" if n < len(a.arm) and a.arm[n] else "")
        t = sorted(t)
        print("%-28s mean %.3f  median %.3f  min %.3f ms (x%.1f = %.2f ms per 200 M reads)  bit-exact %s" % (
            lib, sum(t) / len(t), t[len(t) // 2], t[0], scale, t[len(t) // 2] * scale, ok))
    for eng in engines:
        eng.bind_counts(0)
        eng.close()


if __name__ == "__main__":
    main()
