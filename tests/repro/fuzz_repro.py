#!/usr/bin/env python3
"""Re-create one case of tests/test_gpu_parity.py::test_fuzz_campaign by seed and find the shortest
prefix (in lines) on which the GPU and the C oracle disagree.  usage: fuzz_repro.py SEED [opt=value ...]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tagdigger_amd
from helpers import dirty_fastq, small_index
from oracle import c_oracle

seed = int(sys.argv[1])
opts = [a.split("=") for a in sys.argv[2:]]
rnd = random.Random(seed)
cuts = ["TGCAG", "CWGC", "", "RCATGY", "TGCAT", "CATGG", "GWC"]
cutsite = rnd.choice(cuts)
nl = rnd.choice([("\n",), ("\r\n",), ("\r",), ("\n", "\r\n", "\r")])
taglens = rnd.choice([(8, 30), (20, 70), (60, 130), (30, 64)])
barcodes, tags, cutsites = small_index(rnd, cutsite, nbar=rnd.randint(1, 24), ntag=rnd.randint(1, 120), taglens=taglens)
data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=rnd.randint(1, 3000), nl_choices=nl,
                   long_lines=rnd.random() < 0.3, permanent_shifts=rnd.random() < 0.3)
print("cutsite %r nl %r taglens %r nbar %d ntag %d bytes %d" % (cutsite, nl, taglens, len(barcodes), len(tags), len(data)))
eng = tagdigger_amd.Engine(0)
eng.set_index(barcodes, tags, cutsite)
for k, v in opts:
    eng.set_option(k, int(v))
ora = c_oracle.COracle(barcodes, tags, cutsite)

def differs(buf):
    eng.reset()
    eng.count_bytes(buf)
    got = eng.counts_numpy()
    return not (got == ora.count_bytes(buf)).all()

# line ends
ends = [i + 1 for i, c in enumerate(data) if c in (10, 13) and not (c == 13 and data[i + 1:i + 2] == b"\n")]
print("whole file differs:", differs(data), "lines", len(ends))
lo, hi = 0, len(ends) - 1          # smallest k with differs(data[:ends[k]])
if differs(data[:ends[hi]]):
    while lo < hi:
        mid = (lo + hi) // 2
        if differs(data[:ends[mid]]): hi = mid
        else: lo = mid + 1
    cut = ends[lo]
    print("shortest failing prefix: %d lines, %d bytes (mod 24576: %d, mod 32768: %d)" % (lo + 1, cut, cut % 24576, cut % 32768))
    start = ends[max(0, lo - 5)]
    print("last lines:", data[start:cut])
    buf = data[:cut]
    eng.reset(); eng.count_bytes(buf); got = eng.counts_numpy(); want = ora.count_bytes(buf)
    for r, c in zip(*np.nonzero(got != want)):
        print("cell", r, c, "gpu", got[r, c], "oracle", want[r, c], "barcode", barcodes[r], "tag", tags[c])
    print("stats", eng.stats(), "fixups", eng.debug_counters()[11])
