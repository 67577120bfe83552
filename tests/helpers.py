"""Shared helpers for tests and bench: host reference of the synthetic stream
(via the oracle's C library) and a dirty-FASTQ fuzzer."""
import ctypes as C
import random

import numpy as np

from oracle import c_oracle


# Every way the count can be computed on the device (td_set_option names): the free-running path with its
# producer / consumer main pass (k_fast4, 24 KiB tiles; the default), its second-generation main pass (k_fast2, 24 and 32 KiB tiles) and its first (k_fast, 32 and 16 KiB tiles), and
# the exact look-back kernel.  All give the same counts.
KERNEL_MODES = [
    dict(fastpath=1, kernel=2, tile_kb2=24),
    dict(fastpath=1, kernel=2, tile_kb2=32),
    dict(fastpath=1, kernel=2, tile_kb2=16),
    dict(fastpath=1, kernel=1, tile_kb=32),
    dict(fastpath=1, kernel=1, tile_kb=16),
    dict(fastpath=0, tile_kb=32),
    dict(fastpath=1, kernel=4),
]
DEFAULT_MODE = dict(fastpath=1, kernel=4, tile_kb2=0, tile_kb=32)


def apply_mode(eng, mode):
    for k, v in mode.items():
        eng.set_option(k, v)


def mode_id(mode):
    return ",".join("%s=%s" % kv for kv in mode.items())


def synth_params(cfg):
    return cfg.params(c_oracle.SynthParams)


def synth_host_bytes(cfg, first_read, nreads):
    """Records [first_read, first_read+nreads) from oracle/synth_ref.c (numpy uint8)."""
    out = np.empty(nreads * cfg.record_bytes, dtype=np.uint8)
    P = synth_params(cfg)
    bl = (C.c_char * len(cfg.bar_len)).from_buffer_copy(cfg.bar_len)
    tl = (C.c_char * len(cfg.tag_len)).from_buffer_copy(cfg.tag_len)
    c_oracle.lib().synth_fill_host(C.byref(P), first_read, nreads, cfg.bar_tab, bl, cfg.cut_tab,
                                   cfg.tag_tab, tl, out.ctypes.data)
    return out


def synth_expected(cfg, first_read, nreads):
    """Count matrix implied by the generator's own choices (no FASTQ parsing)."""
    counts = np.zeros((len(cfg.barcodes), len(cfg.tags)), dtype=np.uint64)
    P = synth_params(cfg)
    hits = c_oracle.lib().synth_expected(C.byref(P), first_read, nreads, counts.ctypes.data)
    return counts, hits


def dirty_fastq(rnd, barcodes, tags, cutsites, nrec, nl_choices=("\n",), long_lines=False,
                permanent_shifts=False):
    """FASTQ-ish bytes with every irregularity the reference tolerates."""
    out = []
    for ri in range(nrec):
        u = rnd.random()
        b = rnd.choice(barcodes)
        cs = rnd.choice(cutsites)
        t = rnd.choice(tags)
        carries = len(cs) > 0 and t[:len(cs)] in cutsites
        if u < 0.6:
            seq = b + (t if carries else cs + t) + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 30)))
        elif u < 0.75:
            seq = b + cs + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 80)))
        elif u < 0.85:
            seq = "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 120)))
        else:
            seq = b + (t if carries else cs + t)
            if seq:
                p = rnd.randrange(len(seq))
                seq = seq[:p] + rnd.choice("Nn.-*RYX\x00~`[{@") + seq[p + 1:]
        if rnd.random() < 0.2:
            seq = seq.lower()
        if rnd.random() < 0.08:
            seq = rnd.choice([" ", "\t", "  ", "\x0b\x0c", "\x1c\x1d\x1e\x1f "]) + seq + rnd.choice(["", " ", "\t "])
        if long_lines and rnd.random() < 0.02:
            seq = " " * rnd.randint(100, 700) + seq
        if rnd.random() < 0.05:
            seq = seq[:rnd.randint(0, len(seq))]
        hdr = "@r%d" % ri + ("" if rnd.random() < 0.8 else " " + "x" * rnd.randint(0, 40))
        qual = "I" * (len(seq) if rnd.random() < 0.9 else rnd.randint(0, 5))
        nl = rnd.choice(nl_choices)
        if rnd.random() < 0.03:
            nl = rnd.choice(["\n", "\r\n", "\r"])
        out.append(hdr + nl + seq + nl + "+" + nl + qual + nl)
        r = rnd.random()
        if r < 0.01:
            out.append(rnd.choice(["\n\n\n\n", "\r\n\r\n\n\r", "\r\r\r\r"]))   # four blank lines: phase kept
        elif r < 0.012:
            shift = rnd.choice(["\n", "\n\n", "\r", "\r\n\n"])                   # phase lost ...
            out.append(shift)
            pending = 4 - (shift.count("\n") + shift.count("\r") - shift.count("\r\n"))
            out.append("\r" * pending if not (permanent_shifts and rnd.random() < 0.3) else "")                 # ... and usually restored
    return "".join(out).encode("latin-1")


def small_index(rnd, cutsite, nbar=10, ntag=40, taglens=(20, 70)):
    from oracle.tagdigger_oracle import enumerate_cut_sites
    cutsites = enumerate_cut_sites(cutsite)
    barcodes, full = [], []
    while len(barcodes) < nbar:
        b = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(3, 10)))
        mine = [b + c for c in cutsites]
        if any(a.startswith(o) or o.startswith(a) for a in mine for o in full):
            continue
        barcodes.append(b)
        full += mine
    tags = []
    while len(tags) < ntag:
        t = rnd.choice(cutsites) + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(*taglens)))
        if any(t.startswith(o) or o.startswith(t) for o in tags):
            continue
        tags.append(t)
    return barcodes, tags, cutsites


import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools"))
from compress_formats import bgzf_bytes, gzip_one_member  # noqa: E402,F401  (tools/compress_formats.py: shared with the bench's tiers)


FUZZ_CUTS = ["TGCAG", "CWGC", "", "RCATGY", "TGCAT", "CATGG", "GWC"]


def fuzz_case(seed):
    """Case number `seed` of the randomised campaign (tests/test_gpu_parity.py::test_fuzz_campaign, tools/fuzz_repro.py):
    (barcodes, tags, cutsite, newline styles, bytes)."""
    import random
    rnd = random.Random(seed)
    cutsite = rnd.choice(FUZZ_CUTS)
    nl = rnd.choice([("\n",), ("\r\n",), ("\r",), ("\n", "\r\n", "\r")])
    taglens = rnd.choice([(8, 30), (20, 70), (60, 130), (30, 64)])
    barcodes, tags, cutsites = small_index(rnd, cutsite, nbar=rnd.randint(1, 24), ntag=rnd.randint(1, 120), taglens=taglens)
    data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=rnd.randint(1, 3000), nl_choices=nl,
                       long_lines=rnd.random() < 0.3, permanent_shifts=rnd.random() < 0.3)
    return barcodes, tags, cutsite, nl, data
