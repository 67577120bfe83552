#!/usr/bin/env python3
"""Diagnostic: streaming rates of the building blocks on a synthetic library in HBM
(td_count_lines_device = loads + terminator masks + block reduce; torch copy as the memcpy yardstick)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tagdigger_amd
from tagdigger_amd.synth import SynthConfig

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
cfg = SynthConfig(nreads=reads, nbar=96, nmarkers=5000, seed=2)
eng = tagdigger_amd.Engine(0)
nb = cfg.nbytes()
buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
cfg.fill_device(eng, buf.data_ptr(), 0, reads)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    n = eng.count_lines_device(buf.data_ptr(), nb, stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    n = eng.count_lines_device(buf.data_ptr(), nb, stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("count_lines: %d terminators, %.3f ms  -> %.2f TB/s (includes scan kernel + D2H of the total)" % (n, dt * 1e3, nb / dt / 1e12))
dst = torch.empty_like(buf)
for _ in range(2):
    dst.copy_(buf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    dst.copy_(buf)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("torch copy : %.3f ms -> read %.2f TB/s (+ same written)" % (dt * 1e3, nb / dt / 1e12))
s = torch.zeros(1, device="cuda", dtype=torch.int64)
v = buf.view(torch.int64)
for _ in range(2):
    s = v.sum()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    s = v.sum()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("torch sum  : %.3f ms -> read %.2f TB/s" % (dt * 1e3, nb / dt / 1e12))
