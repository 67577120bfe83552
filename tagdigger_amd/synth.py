"""Canonical synthetic inputs of BASELINE.json's configs.

Bench / test utility, not on the counting path.  Barcode and tag sets are
drawn here (numpy, seeded); the FASTQ bytes themselves are a counter-based
function of (seed, read index) defined in include/td_synth_spec.h and written
straight into HBM by libtagdig's td_synth_fill_device, so any shard of the
stream can be produced on any GPU without moving data.
"""
import ctypes as C

import numpy as np

from . import _binding as B
from .engine import enumerate_cut_sites

BAR_STRIDE = 16
CUT_STRIDE = 16

# nreads / barcodes / markers (tags = 2 x markers) / seed / cut site / barcode length range
CONFIGS = {
    1: dict(nreads=100_000, nbar=8, nmarkers=50, seed=1234, cutsite="TGCAG", bclen=(4, 8)),
    2: dict(nreads=50_000_000, nbar=96, nmarkers=5_000, seed=2, cutsite="TGCAG", bclen=(4, 8)),
    3: dict(nreads=200_000_000, nbar=384, nmarkers=50_000, seed=3, cutsite="TGCAG", bclen=(4, 8)),
    4: dict(nreads=200_000_000, nbar=384, nmarkers=250_000, seed=40, cutsite="TGCAG", bclen=(4, 8)),
    5: dict(nreads=1_000_000_000, nbar=384, nmarkers=50_000, seed=5, cutsite="CWGC", bclen=(4, 10)),
}


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def make_barcodes(rng, n, lo, hi, cutsites):
    """n barcodes of length lo..hi such that no barcode+site is a prefix of another."""
    out, full = [], []
    guard = 0
    while len(out) < n:
        guard += 1
        if guard > 200000:
            raise RuntimeError("could not draw a prefix-free barcode set")
        cand = _rand_seq(rng, int(rng.integers(lo, hi + 1)))
        mine = [cand + c for c in cutsites]
        if any(a.startswith(b) or b.startswith(a) for a in mine for b in full):
            continue
        out.append(cand)
        full.extend(mine)
    return out


def make_tags(rng, nmarkers, cutsites, body=59):
    """Marker-major biallelic tags: site + random body, partner differs at one base after the site."""
    cl = len(cutsites[0])
    L = cl + body
    sites = np.array([np.frombuffer(c.encode(), dtype=np.uint8) for c in cutsites])
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    tags, seen = [], set()
    while len(tags) < 2 * nmarkers:
        need = nmarkers - len(tags) // 2
        arr = np.empty((need, L), dtype=np.uint8)
        arr[:, :cl] = sites[rng.integers(0, len(cutsites), need)]
        arr[:, cl:] = letters[rng.integers(0, 4, (need, body))]
        pos = rng.integers(cl, L, need)
        shift = rng.integers(1, 4, need)
        alt = arr.copy()
        idx = np.arange(need)
        cur = np.searchsorted(letters, alt[idx, pos])
        alt[idx, pos] = letters[(cur + shift) % 4]
        for a, b in zip(arr, alt):
            sa, sb = a.tobytes(), b.tobytes()
            if sa in seen or sb in seen:
                continue
            seen.add(sa)
            seen.add(sb)
            tags.append(sa.decode())
            tags.append(sb.decode())
    return tags


class SynthConfig:
    def __init__(self, nreads, nbar, nmarkers, seed, cutsite="TGCAG", bclen=(4, 8), read_len=100, body=59):
        self.nreads, self.seed, self.cutsite, self.read_len = nreads, seed, cutsite, read_len
        self.cutsites = enumerate_cut_sites(cutsite)
        rng = np.random.default_rng(seed)
        self.barcodes = make_barcodes(rng, nbar, bclen[0], bclen[1], self.cutsites)
        self.tags = make_tags(rng, nmarkers, self.cutsites, body)
        assert max(len(b) for b in self.barcodes) + max(len(t) for t in self.tags) <= read_len
        self.tag_stride = max(len(t) for t in self.tags)
        self.record_bytes = 2 * read_len + 19
        # flat tables for the C-ABI
        self.bar_tab = b"".join(b.encode().ljust(BAR_STRIDE, b"\0") for b in self.barcodes)
        self.bar_len = bytes(len(b) for b in self.barcodes)
        self.cut_tab = b"".join(c.encode().ljust(CUT_STRIDE, b"\0") for c in self.cutsites)
        self.tag_tab = b"".join(t.encode().ljust(self.tag_stride, b"\0") for t in self.tags)
        self.tag_len = np.array([len(t) for t in self.tags], dtype=np.uint16).tobytes()

    @classmethod
    def from_id(cls, cid, nreads=None):
        c = dict(CONFIGS[cid])
        if nreads is not None:
            c["nreads"] = nreads
        return cls(**c)

    def params(self):
        return B.SynthParams(self.seed, len(self.barcodes), len(self.tags), len(self.cutsites),
                             self.read_len, len(self.cutsites[0]), self.tag_stride)

    def nbytes(self, nreads=None):
        return (self.nreads if nreads is None else nreads) * self.record_bytes

    def fill_device(self, engine, d_ptr, first_read, nreads, stream=0):
        """Write records [first_read, first_read+nreads) at d_ptr (device memory)."""
        P = self.params()
        bl = (C.c_char * len(self.bar_len)).from_buffer_copy(self.bar_len)
        tl = (C.c_char * len(self.tag_len)).from_buffer_copy(self.tag_len)
        B.check(engine._L.td_synth_fill_device(engine._h, C.byref(P), first_read, nreads, self.bar_tab, bl,
                                               self.cut_tab, self.tag_tab, tl, C.c_void_p(d_ptr),
                                               C.c_void_p(stream) if stream else None))

    def expected_device(self, engine, d_counts, first_read, nreads, stream=0):
        """Add the matrix the generator's own choices imply (uint32 [barcodes][tags] at d_counts, device
        memory) -- no FASTQ is parsed; returns the number of hits."""
        P = self.params()
        hits = C.c_uint64(0)
        B.check(engine._L.td_synth_expected_device(engine._h, C.byref(P), first_read, nreads, C.c_void_p(d_counts),
                                                   C.byref(hits), C.c_void_p(stream) if stream else None))
        return hits.value
